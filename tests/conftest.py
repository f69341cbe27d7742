import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import __graft_entry__ as ge  # noqa: E402

ge.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    """One probe per session: does ccsim_create find a HIP device?  (No torch import: it takes minutes on a fresh box.)"""
    try:
        from cluster_capacity_amd import capi

        e = capi.Engine(device=0)
        e.close()
        return True
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `pytest tests` on a host without a HIP device: skip the gpu-marked tests instead of failing them one by one.
    # With an explicit `-m gpu` nothing is skipped: on a GPU box a missing device / library must fail loudly.
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items or "gpu" in (config.getoption("-m") or "").replace("not gpu", ""):
        return
    if not _have_gpu():
        skip = pytest.mark.skip(reason="no HIP device visible (run with -m gpu on the GPU box)")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def ccref():
    """The CPU oracle (test infrastructure)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ccref_py

    ccref_py.build()
    return ccref_py
