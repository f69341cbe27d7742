"""CPU-side checks of the drop-in boundary: libccsim.so builds for gfx950, loads, and exports every
symbol include/ccsim.h declares (no compute calls: there is no GPU here)."""
import os
import re

from cluster_capacity_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ccsim.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ccsim_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build_all()
    lib = capi.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert n in capi.SYMBOLS, f"{n} declared in ccsim.h but not bound in capi.py"
        assert getattr(lib, n) is not None
    assert set(capi.SYMBOLS) == set(names)
    assert lib.ccsim_abi_version() == capi.ABI_VERSION == 3


def test_struct_layouts_match_header_sizes():
    # sizes computed by the C compiler for the same header must equal the ctypes mirrors
    import subprocess, tempfile, ctypes
    prog = r'''
    #include <stdio.h>
    #include "ccsim.h"
    int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ccsim_config), sizeof(ccsim_nodes), sizeof(ccsim_requirement),
      sizeof(ccsim_term), sizeof(ccsim_pod), sizeof(ccsim_profile), sizeof(ccsim_report), sizeof(ccsim_cycle),
      sizeof(ccsim_spread_constraint), sizeof(ccsim_ipa));return 0;}
    '''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        sizes = list(map(int, subprocess.check_output([os.path.join(d, "t")]).split()))
    mirrors = [capi.CConfig, capi.CNodes, capi.CReq, capi.CTerm, capi.CPod, capi.CProfile, capi.CReport, capi.CCycle, capi.CSpread, capi.CIpa]
    assert sizes == [ctypes.sizeof(m) for m in mirrors]


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(capi.CcsimError):
        capi.Engine(device=0)


def _build_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "ccsim_demo")
    csrc = os.path.join(ROOT, "cluster-capacity_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "ccsim_demo.c"), "-L", csrc, "-lccsim",
                           f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_c_demo_compiles_against_the_header(tmp_path):
    """include/ccsim.h is usable from plain C (what cgo sees) and links against libccsim.so."""
    build.build_all()
    _build_demo(tmp_path)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_c_demo_reproduces_readme_answer(tmp_path):
    """The reference's README demo through the C ABI from a C program, no Python in the loop: 52 = 13 x 4."""
    import subprocess
    out = subprocess.run([_build_demo(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "The cluster can schedule 52 instance(s)" in out.stdout and "4 Insufficient cpu" in out.stdout
    assert out.stdout.count("13 instance(s)") == 4
