"""Build recipe for libccsim.so (HIP, gfx950 only).  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
SOURCES = ["ccsim_engine.hip"]
DEPS = ["ccsim_kernels.h", "ccsim_level.h", "ccsim_persist.h", "ccsim_multi.h", "ccsim_coupled.h", "ccsim_sampled.h", "ccsim_sampled_zone.h", "ccsim_search_full.h", os.path.join(ROOT, "include", "ccsim.h")]
# -ffp-contract=off: the fp64 score arithmetic must match Go (no FMA fusion); no fast-math anywhere.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-Wno-pass-failed"]


def lib_path() -> str:
    return os.path.join(CSRC, "libccsim.so")


def source_sha16() -> str:
    """Hash of everything libccsim.so is built from (sources, the ABI header, the flags).  The binary itself embeds the path of
    the build directory, so the same sources built elsewhere give another file hash: measurement files are stamped with both."""
    import hashlib

    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in SOURCES] + [d if os.path.isabs(d) else os.path.join(CSRC, d) for d in DEPS]:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read() + b"\0")
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def _stale(out: str) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [d if os.path.isabs(d) else os.path.join(CSRC, d) for d in DEPS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


class _BuildLock:
    """One builder at a time per artefact, and nobody loads a file that is being written: processes that find the artefact stale
    together (pytest-xdist workers on a fresh checkout) queue on the lock, the first one builds into a temporary name and renames
    it into place, the others find it fresh.  (A worker once dlopen()ed a library another worker's compiler was still writing.)"""

    def __init__(self, out: str):
        self.path = out + ".lock"

    def __enter__(self):
        import fcntl

        os.makedirs(os.path.dirname(self.path), exist_ok=True)
        self.f = open(self.path, "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl

        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def _run_into(cmd_without_out: list, out: str, verbose: bool) -> None:
    tmp = "%s.tmp.%d" % (out, os.getpid())
    cmd = cmd_without_out + ["-o", tmp]
    if verbose:
        print(" ".join(cmd_without_out + ["-o", out]))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build_all(force: bool = False, verbose: bool = False) -> str:
    out = lib_path()
    if force or _stale(out):
        with _BuildLock(out):
            if force or _stale(out):
                hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
                extra = os.environ.get("CCSIM_EXTRA_FLAGS", "").split()
                _run_into([hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES], out, verbose)
    return out


HOST = os.path.join(HERE, "host")
HOST_SOURCES = ["main.cpp"]
HOST_DEPS = sorted(f for f in os.listdir(HOST) if f.endswith(".hpp"))  # every header of the host: a stale binary is a silent lie


def host_path() -> str:
    san = os.environ.get("CCHOST_SANITIZE")  # e.g. "address,undefined" or "thread": a second binary, for tests/ under the sanitizers
    return os.path.join(HERE, "bin", "cluster-capacity-native" + ("-" + san.replace(",", "-") if san else ""))


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The native (C++) host above the C ABI: ingest, CLI, report.  Loads libccsim.so at run time (dlopen)."""
    out = host_path()
    deps = [os.path.join(HOST, f) for f in HOST_SOURCES + HOST_DEPS] + [os.path.join(ROOT, "include", "ccsim.h")]
    stale = lambda: not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)
    if force or stale():
        with _BuildLock(out):
            if force or stale():
                san = os.environ.get("CCHOST_SANITIZE")
                opt = ["-O1", "-g", "-fno-omit-frame-pointer", "-fno-sanitize-recover=all", "-fsanitize=" + san] if san else ["-O2"]
                _run_into([os.environ.get("CXX", "g++")] + opt + ["-std=c++17", "-Wall", "-Wextra"] +
                          [os.path.join(HOST, f) for f in HOST_SOURCES] + ["-ldl", "-pthread"], out, verbose)
    return out


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
