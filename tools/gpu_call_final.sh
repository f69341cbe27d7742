#!/bin/bash
# the headline evidence of tools/gpu_round_profile.sh on the FINAL library (after the coupled window went to 2048 cycles): PMC traffic,
# the bench line, its kernel stats, config 5, the coupled windows, smoke -- plus the parity files closest to the change
exec < /dev/null
cd /root/repo
R=r04
O=/root/repo/gpurun_out/$R
mkdir -p $O
python - > $O/lib_hash.txt <<'PY'
import hashlib, sys
sys.path.insert(0, ".")
import __graft_entry__ as ge
ge.load_package()
from cluster_capacity_amd import build as b
print("libccsim.so sha256[:16]", hashlib.sha256(open(b.lib_path(), "rb").read()).hexdigest()[:16], "| sources + flags sha256[:16]", b.source_sha16())
PY
cat $O/lib_hash.txt
bash tools/gpu_pmc.sh $R "FETCH_SIZE" "WRITE_SIZE" 2>&1 | tee $O/pmc_summary_final.txt | tail -8
cp gpurun_out/pmc_traffic_$R.json $O/pmc_traffic.json 2>/dev/null && cp gpurun_out/pmc_traffic_$R.json profiles/$R/pmc_traffic.json
rm -rf gpurun_out/pmc_${R}_1 gpurun_out/pmc_${R}_2
cd /root/repo
timeout 400 python bench.py > $O/bench_1M.json 2> $O/bench_1M.err; tail -c 200 $O/bench_1M.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/bench.py --no-cpu --no-variants --seq-rounds 0 > $O/bench_1M_under_rocprofv3.json 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_1M_kernel_stats.csv && cut -c1-200 $O/bench_1M_kernel_stats.csv | head -4
rm -rf $O/ks
cd /root/repo
timeout 300 python tools/bench_c5.py 100000 1024 200000 64 2>&1 | grep -v amdgpu.ids | tee $O/bench_c5.txt | grep "window=" | cut -c1-200
CCSIM_BENCH_SKIP_SEQ=1 timeout 200 python tools/bench_coupled.py 1000000 50000 2048,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_1M_64zones.txt | grep windowed | cut -c1-200
timeout 200 python tools/bench_coupled.py 100000 50000 2048,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_100k.txt | grep "windowed\|one pass" | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt | tail -3
( cat $O/lib_hash.txt; echo "python -m pytest tests/test_coupled.py tests/test_baseline_configs.py tests/test_multi.py tests/test_spread.py tests/test_ipa.py tests/test_bench_line.py -m gpu -q -n 4"; timeout 400 python -m pytest tests/test_coupled.py tests/test_baseline_configs.py tests/test_multi.py tests/test_spread.py tests/test_ipa.py tests/test_bench_line.py -m gpu -q -n 4 --timeout 300 2>&1 | grep -v amdgpu.ids | tail -4 ) | tee $O/gpu_tests_final_library_subset.txt | tail -3
