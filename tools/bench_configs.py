#!/usr/bin/env python
"""Secondary measurements for BASELINE.md section 3 (NOT the driver's bench line): the BASELINE.json configs C1-C3 and a
C5-shaped coupled-plugin case, each timed on the GPU engine (both modes where valid) and on the CPU oracle (1 thread
and OpenMP), with the parity gate applied first.  Prints a markdown table."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
import numpy as np  # noqa: E402
import ccref_py  # noqa: E402
import helpers as H  # noqa: E402
from cluster_capacity_amd import capi, model as M, synth  # noqa: E402


def timed(f, reps=3):
    best, out = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = f()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, out


def gpu(nodes, pod, prof, limit, mode):
    eng = capi.Engine(device=0)
    eng.load(nodes, pod, prof)

    def run():
        eng.reset_state()
        return eng.run(max_limit=limit, mode=mode, want_log=False)

    run()
    dt, r = timed(run)
    eng.close()
    return dt, r


def row(name, nodes, pod, prof, limit, cpu_limit, modes):
    cells = []
    threads = min(16, os.cpu_count() or 1)
    ref_small = ccref_py.run(prof, nodes, pod, max_limit=cpu_limit, want_log=False)
    t1, _ = timed(lambda: ccref_py.run(prof, nodes, pod, max_limit=cpu_limit, threads=1, want_log=False), reps=1)
    tn, _ = timed(lambda: ccref_py.run(prof, nodes, pod, max_limit=cpu_limit, threads=threads, want_log=False), reps=1)
    cells.append(f"oracle 1T {ref_small.placed / t1:,.0f}/s; OMP x{threads} {ref_small.placed / tn:,.0f}/s (first {ref_small.placed} cycles)")
    for mode in modes:
        # parity gate on the bounded prefix, then the full run
        eng = capi.Engine(device=0)
        eng.load(nodes, pod, prof)
        chk = eng.run(max_limit=cpu_limit, mode=mode, want_log=False)
        eng.close()
        assert chk.placed == ref_small.placed and np.array_equal(chk.per_node_count, ref_small.per_node_count), (name, mode)
        dt, r = gpu(nodes, pod, prof, limit, mode)
        cells.append(f"GPU {mode}: {r.placed:,} placements in {dt * 1e3:.1f} ms = {r.placed / dt:,.0f}/s ({r.scans} passes)")
    print(f"| {name} | {nodes.n:,} | " + " | ".join(cells) + " |")


def main():
    print("| config | nodes | CPU oracle | GPU | GPU |\n|---|---|---|---|---|")
    r = ccref_py.run(M.Profile.default(), H.readme_nodes(4), H.examples_pod())
    print(f"| C1 README demo | 4 | oracle: {r.placed} = {r.per_node_count.tolist()} | – | – |")
    n, p, f = synth.make_config("C2", n_nodes=10_000)
    row("C2 Fit-only", n, p, f, 0, 2000, ["sequential", "batched"])
    n, p, f = synth.make_config("C3", n_nodes=100_000)
    row("C3 default plugins", n, p, f, 0, 400, ["batched"])
    row("C3 default plugins, 4096 cycles", n, p, f, 4096, 400, ["sequential"])
    n, p, f = synth.make_config("C3", n_nodes=100_000)
    n.label_cols.append(np.arange(1, n.n + 1, dtype=np.int32))
    p.spread = [synth.zone_spread(n.n, max_skew=2)]
    p.ipa = M.InterPodAffinity(key_cols=[2], key_ndom=[n.n], anti_keys=[0], anti_self=[True], anti_existing=[None])
    row("C5-shaped (ONE spec): zone DoNotSchedule spread + hostname anti-affinity, 2048 cycles", n, p, f, 2048, 200, ["sequential"])
    c5(100_000, 1024, 200_000)


def c5(n_nodes, n_specs, limit):
    """BASELINE config 5 proper: n_specs genpod-shaped pod specs cycled round-robin (ccsim_set_pods), parity gate on the
    oracle's first cycles, algorithmic bytes per placement = N x B_node / pods sharing a pass."""
    nodes, pods, prof = synth.make_c5(n_nodes, n_specs)
    threads = min(16, os.cpu_count() or 1)
    cycles = 300
    t0 = time.perf_counter()
    ref = ccref_py.run_multi(prof, nodes, pods, max_limit=cycles, threads=threads)
    t_cpu = time.perf_counter() - t0
    eng = capi.Engine(device=0)
    eng.load(nodes, pods, prof)
    head = eng.run(max_limit=cycles, log_cap=cycles)
    assert np.array_equal(head.log, ref.log) and np.array_equal(head.per_spec_count, ref.per_spec_count), "C5 parity gate"

    def run():
        eng.reset_state()
        return eng.run(max_limit=limit, want_log=False, log_cap=0)

    run()
    dt, r = timed(run)
    eng.close()
    pods_per_pass = r.placed / max(1, r.scans)
    print(f"| C5: {n_specs} genpod-shaped specs round-robin (zone DoNotSchedule spread + hostname anti-affinity to their own label), "
          f"{limit:,} placements | {n_nodes:,} | oracle OMP x{threads} {ref.placed / t_cpu:,.0f}/s (first {cycles} cycles) | "
          f"GPU windows of <= 128 pods: {r.placed:,} placements in {dt * 1e3:.1f} ms = {r.placed / dt:,.0f}/s ({r.scans} passes, "
          f"{pods_per_pass:.1f} pods per pass, {r.pass_launches} passes ended early by the exact validation) | "
          f"algorithmic bytes per placement = N x 92 B / pods per pass = {n_nodes * 92 / pods_per_pass / 1e3:,.0f} KB "
          f"(one cycle per pass: {n_nodes * 92 / 1e6:.1f} MB) |")


if __name__ == "__main__":
    main()
