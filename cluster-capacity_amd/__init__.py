"""cluster-capacity_amd: MI355X-native batched placement engine behind the
kubernetes-sigs/cluster-capacity simulator API (hot path only).

The directory name carries a hyphen (it mirrors the reference's name), so it is imported under
the module name ``cluster_capacity_amd`` via ``__graft_entry__.load_package()``.
"""
from . import model, report  # noqa: F401

__all__ = ["model", "report"]
