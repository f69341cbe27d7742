#!/bin/bash
# round 5: the sampled search on resident block summaries (csrc/ccsim_sampled.h): parity (tests/test_sampling.py), throughput at 1M nodes A/B
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/${1:-r5e}
mkdir -p $O
timeout 900 python -m pytest tests/test_sampling.py -m gpu -x -q -n 4 > $O/tests.txt 2>&1; tail -8 $O/tests.txt
for sb in 1 0; do
CCSIM_SB=$sb timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_mode_b_1M.txt
import os, sys, time, dataclasses
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import __graft_entry__ as ge; ge.load_package()
import numpy as np, ccref_py
from cluster_capacity_amd import capi, synth
for n, lim in ((1_000_000, 20000), (100_000, 20000)):
    nodes, pod, prof = synth.make_config("C4", n_nodes=n)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=0)
    ref = ccref_py.run(prof, nodes, pod, max_limit=2000, threads=16)
    e = capi.Engine(device=0); e.load(nodes, pod, prof)
    head = e.run(max_limit=2000, mode="sequential", log_cap=2000)
    assert np.array_equal(head.log, ref.log) and head.evaluated_total == ref.evaluated_total
    best = None
    for rep in range(3):
        e.reset_state(); t0 = time.perf_counter(); r = e.run(max_limit=lim, mode="sequential", want_log=False, log_cap=0); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print(f"CCSIM_SB={os.environ.get('CCSIM_SB')} {n} nodes, adaptive sampling: {r.placed} cycles in {best*1e3:.1f} ms -> {r.placed/best:.3e} placements/s, {best*1e6/r.placed:.2f} us/cycle, "
          f"{r.evaluated_total/r.placed:.0f} nodes visited per cycle, launches {r.pass_launches}, kernel {r.kernel_ns/1e6:.1f} ms", flush=True)
    e.close()
PY
done
