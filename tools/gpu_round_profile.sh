#!/bin/bash
# round-end evidence: the bench line, rocprofv3 kernel stats of the same command, PMC traffic (separate passes), the
# barrier micro-benchmark.  Everything lands in gpurun_out/r02/ ; copy what is to be judged into profiles/r02/.
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r02
mkdir -p $O
timeout 200 python -m pytest tests/test_multi.py tests/test_golden.py tests/test_ports_images.py tests/test_dist_rccl.py -m gpu -q -rf 2>&1 | tail -5 | tee $O/quick_tests.txt
# PMC traffic first (separate rocprofv3 passes, --kernel-trace only): bench.py accepts profiles/r02/pmc_traffic.json only if it
# was collected with the libccsim.so it runs
bash tools/gpu_pmc.sh r02 "FETCH_SIZE" "WRITE_SIZE" 2>&1 | tail -12
cp gpurun_out/pmc_traffic_r02.json $O/pmc_traffic.json 2>/dev/null && cp gpurun_out/pmc_traffic_r02.json profiles/r02/pmc_traffic.json
rm -rf gpurun_out/pmc_r02_1 gpurun_out/pmc_r02_2
cd /root/repo
timeout 400 python bench.py > $O/bench_1M.json 2> $O/bench_1M.err; tail -c 400 $O/bench_1M.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/bench.py --no-cpu --no-variants --seq-rounds 0 > $O/bench_1M_under_rocprofv3.json 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_1M_kernel_stats.csv && cut -c1-200 $O/bench_1M_kernel_stats.csv | head -12
cd /root/repo
rm -rf $O/ks
# config 5 (1024 pod specs): throughput line + per-kernel time
timeout 300 python tools/bench_c5.py 100000 1024 200000 64 2>&1 | grep -v amdgpu.ids | tee $O/bench_c5.txt | cut -c1-300
bash tools/gpu_c5_prof.sh > /dev/null 2>&1; cp gpurun_out/c5_kernel_stats.csv $O/c5_kernel_stats.csv 2>/dev/null
timeout 120 python tools/persist_prof.py 1000000 8 1,16,64 2>&1 | grep -v amdgpu.ids | tee $O/persist_phase_profile.txt | cut -c1-250
CCSIM_FORCE_DIST=1 timeout 120 python bench.py --no-variants --no-cpu --seq-rounds 0 --steps 3 2>/dev/null > $O/bench_dist_world1.json; cut -c1-200 $O/bench_dist_world1.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
