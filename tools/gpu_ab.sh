#!/bin/bash
# A/B: sequential + batched throughput of the in-tree library vs cluster-capacity_amd/csrc/libccsim_head.so
exec < /dev/null
cd /root/repo
[ -f cluster-capacity_amd/csrc/libccsim_head.so ] || { echo "build the baseline library first: cluster-capacity_amd/csrc/libccsim_head.so"; exit 1; }
mkdir -p gpurun_out
run() {
timeout 200 python - <<'PY'
import time
import __graft_entry__ as ge; ge.load_package()
from cluster_capacity_amd import capi, synth
n,p,f = synth.make_config("C4", n_nodes=1_000_000)
e = capi.Engine(device=0); e.load(n,p,f)
for mode in ("batched","sequential"):
    ns,b = e.time_scan(300, mode=mode); print(mode, "full pass: %.2f us/launch" % (ns/300/1e3))
for rep in range(3):
    e.reset_state(); e.run(max_limit=2048, mode="sequential", want_log=False); e.reset_state()
    t=time.perf_counter(); r=e.run(max_limit=2048, mode="sequential", want_log=False); dt=time.perf_counter()-t
    print("sequential: %.0f placements/s  (kernel %.1f ms of %.1f ms)" % (r.placed/dt, r.kernel_ns/1e6, dt*1e3))
for rep in range(2):
    e.reset_state()
    t=time.perf_counter(); r=e.run(max_limit=0, mode="batched", want_log=False); dt=time.perf_counter()-t
    print("batched: %.3e placements/s (%.1f ms)" % (r.placed/dt, dt*1e3))
PY
}
echo "== current"; run
cp cluster-capacity_amd/csrc/libccsim.so /tmp/cur.so; cp cluster-capacity_amd/csrc/libccsim_head.so cluster-capacity_amd/csrc/libccsim.so
echo "== head"; run
cp /tmp/cur.so cluster-capacity_amd/csrc/libccsim.so
echo "== current C5-shaped"; timeout 120 python tools/c5_shaped.py 2>&1 | grep C5-shaped
cp cluster-capacity_amd/csrc/libccsim.so /tmp/cur.so; cp cluster-capacity_amd/csrc/libccsim_head.so cluster-capacity_amd/csrc/libccsim.so
echo "== head C5-shaped"; timeout 120 python tools/c5_shaped.py 2>&1 | grep C5-shaped
cp /tmp/cur.so cluster-capacity_amd/csrc/libccsim.so
