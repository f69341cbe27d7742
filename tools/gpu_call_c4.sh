#!/bin/bash
# the C4 headline pieces of tools/gpu_round_profile.sh alone (a second box: the kernel's time varies a few per cent between boxes)
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r04b
mkdir -p $O
python - > $O/lib_hash.txt <<'PY'
import hashlib, sys
sys.path.insert(0, ".")
import __graft_entry__ as ge
ge.load_package()
from cluster_capacity_amd import build as b
print("libccsim.so sha256[:16]", hashlib.sha256(open(b.lib_path(), "rb").read()).hexdigest()[:16], "| sources + flags sha256[:16]", b.source_sha16())
PY
timeout 400 python bench.py > $O/bench_1M.json 2> $O/bench_1M.err; tail -c 300 $O/bench_1M.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/bench.py --no-cpu --no-variants --seq-rounds 0 > $O/bench_1M_under_rocprofv3.json 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_1M_kernel_stats.csv && cut -c1-200 $O/bench_1M_kernel_stats.csv | head -4
rm -rf $O/ks
cd /root/repo
timeout 120 python tools/persist_prof.py 1000000 8 1024 2>&1 | grep -v amdgpu.ids | tee $O/persist_phase_profile.txt | cut -c1-330
timeout 120 python tools/step_breakdown.py 2>&1 | grep -v amdgpu.ids | tee $O/step_breakdown.txt | cut -c1-250
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 | tee $O/clocks.txt
