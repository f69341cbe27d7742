#!/bin/bash
# per-kernel time of one batched 1M-node run (rocprofv3 --kernel-trace --stats); bounded, no stdin reads
exec < /dev/null
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/run1.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge; ge.load_package()
from cluster_capacity_amd import capi, synth
n,p,f = synth.make_config("C4", n_nodes=1_000_000)
e = capi.Engine(device=0); e.load(n,p,f)
for rep in range(2):
    e.reset_state(); r = e.run(max_limit=0, mode="batched", want_log=False)
print(r.placed, r.scans, r.kernel_ns/1e6)
PY
rm -rf /root/repo/gpurun_out/prof_ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_ks -o ks -- python /tmp/run1.py 2>&1 | grep -v amdgpu.ids | tail -2
f=$(find /root/repo/gpurun_out/prof_ks -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-170 "$f" | head -10
