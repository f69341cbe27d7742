"""Build hygiene of the gfx950 kernels (no GPU needed: hipcc cross-compiles and reports per-kernel resources).

A per-lane scratch frame in a streaming kernel is pure overhead (every lane first copies its frame to memory):
the node-scan kernels must keep everything in registers; only the one-block decision kernels may hold their
400-byte DevState working copy in scratch."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import kernel_resources as KR  # noqa: E402


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_streaming_kernels_use_no_scratch_and_do_not_spill():
    # read off the library the suite runs with (build_all rebuilds it only when a source is newer): the AMDGPU metadata of its gfx950
    # code object holds what hipcc's -Rpass-analysis=kernel-resource-usage prints (KR.collect(): a second 70-second compile)
    from cluster_capacity_amd import build as B
    rows = KR.collect_from_library(B.build_all())
    scans = [k for k in rows if k.startswith(("k_scan<", "k_level_score<", "k_level_commit<"))]
    assert len(scans) >= 30  # every NX / coupled / narrow / sampled variant was instantiated
    for k in scans + ["k_level_final", "k_level_decide", "k_smp_prefix", "k_hist", "k_static", "k_rows_build", "k_rows_flush"]:
        # the throughput shapes (no extended resources: NX = 0) keep everything in registers; the NX > 0 variants also carry the
        # general resource-list scoring (dynamic_score_gen, runtime-indexed lists): a few dozen bytes of frame are tolerated there
        # (the widest variant, 9 extra columns, sits at 80 B since the static word also carries the ImageLocality score)
        limit = 0 if ("<0," in k or "<" not in k) else (96 if "<9," in k else 64)
        assert int(rows[k]["ScratchSize"]) <= limit, (k, rows[k])
        assert rows[k]["VGPRs Spill"] == "0", (k, rows[k])  # (SGPRs may spill into VGPR lanes: no memory traffic)
    persist = [f"k_level_persist<{k},{mb}>" for k in (1, 2, 4, 8) for mb in (0, 1)]  # (local grid reduce, mailbox form)
    for k in persist + ["k_multi_scan", "k_multi_commit_par"]:
        assert rows[k]["VGPRs Spill"] == "0", (k, rows[k])
    for k in persist:  # 512-thread workgroups, one per CU: 2 waves per SIMD is all the kernel is launched with
        assert int(rows[k]["ScratchSize"]) == 0 and int(rows[k]["Occupancy"]) >= 2, (k, rows[k])
    # (k_multi_commit_par's launch also carries the in-order commit as its wave 0 -- its MState working copy is the frame)
    assert rows["k_multi_scan"]["ScratchSize"] == "0" and int(rows["k_multi_commit_par"]["ScratchSize"]) <= 192
    assert rows["k_multi_refresh"]["ScratchSize"] == "0" and rows["k_multi_refresh"]["VGPRs Spill"] == "0"
    for k in ("k_final", "k_decide"):  # one working copy of DevState, nothing else (not the 1.7 KB argument block)
        # (k_decide, round 6: + the eight assumed spread minima it keeps to see whether the decision moved them -- the sampled search on shards
        # goes back to its counting pass then; a one-thread kernel, the frame is not on any throughput path)
        assert int(rows[k]["ScratchSize"]) <= (768 if k == "k_final" else 832), (k, rows[k])
        assert rows[k]["VGPRs Spill"] == "0"
    # the throughput kernels keep >= 4 waves per SIMD for the common shapes (no extended resources)
    for k in ("k_scan<0,0,1,0>", "k_scan<0,0,0,0>", "k_level_score<0,1>", "k_level_score<0,0>", "k_level_commit<0,0>"):
        assert int(rows[k]["Occupancy"]) >= 4, (k, rows[k])
    # the narrow commit kernel carries the run-down skip's fp64 coefficients (run_down_safe_skip): 3 waves per SIMD; it is a sparse,
    # latency-bound pass (a few per cent of the nodes per level), and capping it at 128 VGPRs spills
    assert int(rows["k_level_commit<0,1>"]["Occupancy"]) >= 3, rows["k_level_commit<0,1>"]
    for k in [k for k in rows if k.startswith(("k_cw_scan<", "k_cw_decide", "k_scan_fused<"))] + ["k_cw_top", "k_cw_merge", "k_final_fused"]:
        assert rows[k]["VGPRs Spill"] == "0" and int(rows[k]["ScratchSize"]) == 0, (k, rows[k])  # round 3's kernels: everything in registers


def test_kernel_name_demangling():
    assert KR.demangle_kernel("_ZN5ccsim7k_finalENS_8ScanArgsE") == "k_final"
    assert KR.demangle_kernel("_ZN5ccsim6k_scanILi0ELb0ELb1ELi2EEEvNS_8ScanArgsE") == "k_scan<0,0,1,2>"
