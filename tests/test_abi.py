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
    assert lib.ccsim_abi_version() == capi.ABI_VERSION == 5


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

from helpers import SUBPROC_TIMEOUT  # noqa: E402


@pytest.mark.gpu
def test_c_demo_reproduces_readme_answer(tmp_path):
    """The reference's README demo through the C ABI from a C program, no Python in the loop: 52 = 13 x 4."""
    import subprocess
    out = subprocess.run([_build_demo(tmp_path)], capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "The cluster can schedule 52 instance(s)" in out.stdout and "4 Insufficient cpu" in out.stdout
    assert out.stdout.count("13 instance(s)") == 4


def _header_structs():
    """struct name -> member names, read out of include/ccsim.h."""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ccsim.h")).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(ccsim_\w+);", text, re.S):
        members = set()
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                for part in decl.split(","):
                    m = re.search(r"(\w+)\s*(?:\[[^\]]*\])*\s*$", part.strip())
                    if m:
                        members.add(m.group(1))
        out[name] = members
    return out


def test_integration_md_go_binding_names_what_the_header_declares():
    """INTEGRATION.md's cgo binding cannot be compiled here (no Go toolchain): at least every C identifier, and every struct member it
    touches through a `var c C.ccsim_x` / `c := C.ccsim_x{...}` value, must exist in include/ccsim.h."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    header = open(os.path.join(ROOT, "include", "ccsim.h")).read()
    structs = _header_structs()
    assert {"ccsim_config", "ccsim_nodes", "ccsim_pod", "ccsim_profile", "ccsim_report"} <= structs.keys()
    blocks = re.findall(r"```go\n(.*?)```", md, re.S)
    assert blocks
    checked = 0
    for ident in set(re.findall(r"\bC\.((?:ccsim|CCSIM)_\w+)", "\n".join(blocks))):
        assert re.search(r"\b%s\b" % ident, header), ident
    for block in blocks:
        for func in re.split(r"\nfunc ", block):
            for var, struct in re.findall(r"\b(\w+) :?= C\.(ccsim_\w+)\{", func) + re.findall(r"\bvar (\w+) C\.(ccsim_\w+)\b", func):
                if struct not in structs:
                    continue
                used = set(re.findall(r"\b%s\.(\w+)" % var, func))
                lit = re.search(r"\b%s :?= C\.%s\{(.*?)\}\n" % (var, struct), func, re.S)
                if lit:
                    used |= set(re.findall(r"(\w+):", lit.group(1)))
                for member in used:
                    assert member in structs[struct], (struct, member)
                    checked += 1
    assert checked > 40


def test_cached_report_block_keeps_its_arrays_across_other_runs(tmp_path):
    """capi.Engine hands a caller that reuses buffers ONE report block (ccsim_report) run after run.  A run WITH a log in between builds
    its own block and arrays: the cached block must still point at live arrays afterwards (its per-spec array used to be dropped), and
    the result must be read from the arrays the library wrote.  Driven against tests/abi_recorder.c in a process of its own."""
    import subprocess
    import sys
    import textwrap
    here = os.path.dirname(os.path.abspath(__file__))
    rec = tmp_path / "libabi_recorder.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", str(rec), os.path.join(here, "abi_recorder.c")])
    code = textwrap.dedent("""
        import ctypes, gc, sys
        sys.path.insert(0, %r)
        import __graft_entry__ as ge
        ge.load_package()
        import numpy as np
        from cluster_capacity_amd import capi, synth
        nodes, pod, prof = synth.make_config("C4", n_nodes=2048)
        e = capi.Engine(device=0)
        e.load(nodes, pod, prof)
        e._pin_per_node = (0, np.zeros(2048, np.int32))  # (stands in for the page-locked array: no GPU here)
        rep1, pn1, _, ht1 = e._report(False, 0, True)
        spec1 = e._per_spec
        addr = ctypes.addressof(rep1.per_spec_count.contents)
        assert addr == spec1.ctypes.data
        e._report(True, 16, False)  # a run with a log: its own block, its own arrays
        assert e._per_spec is not spec1
        del spec1
        gc.collect()
        rep2, pn2, log2, ht2 = e._report(False, 0, True)
        assert rep2 is rep1 and pn2 is pn1 and ht2 is ht1 and log2 is None
        assert ctypes.addressof(rep2.per_spec_count.contents) == addr == e._per_spec.ctypes.data  # the block's array is alive and is the one read
        assert ctypes.addressof(rep2.hist_taintset.contents) == ht2.ctypes.data
        print("ok")
    """) % os.path.dirname(here)
    env = dict(os.environ, CCSIM_LIB=str(rec), CCSIM_RECORD=str(tmp_path / "rec.json"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]
