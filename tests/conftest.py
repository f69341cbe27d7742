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


# Run order of the files (with the driver's `-x`, a plumbing test that sorts early must not be able to take
# the kernel parity evidence down with it).  Kernel parity against the oracle first, through the C ABI in-process; then the
# configs / goldens; the in-process RCCL communicator; last whatever starts child processes (C demo, the native host, its
# one-rank RCCL path on a cold box).  Files not named keep their alphabetical order between the two groups.
_ORDER_FIRST = ["test_gpu_parity.py", "test_persist.py", "test_sampling.py", "test_spread.py", "test_ipa.py", "test_coupled.py", "test_multi.py",
                "test_ports_images.py", "test_golden.py", "test_baseline_configs.py", "test_kernel_resources.py"]
_ORDER_LAST = ["test_dist_mailbox.py", "test_dist_rccl.py", "test_abi.py", "test_preemption.py", "test_ingest_cli.py", "test_host_robustness.py", "test_native_host.py"]


def _file_rank(item) -> int:
    name = os.path.basename(str(item.fspath))
    if name in _ORDER_FIRST:
        return _ORDER_FIRST.index(name)
    if name in _ORDER_LAST:
        return 1000 + _ORDER_LAST.index(name)
    return 500


def pytest_collection_modifyitems(config, items):
    items.sort(key=_file_rank)  # (stable: the order inside a file, and among unnamed files, is pytest's)
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
