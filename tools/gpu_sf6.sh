#!/bin/bash
# the full search on resident summaries (k_sf_cycles): parity tests of the sequential mode, then throughput vs the one-pass-per-cycle form
#   tools/gpu_sf6.sh [out dir under gpurun_out, default r06/sf]
exec < /dev/null
O=/root/repo/gpurun_out/${1:-r06/sf}
mkdir -p $O
cd /root/repo
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest ${SF_TESTS:-tests/test_gpu_parity.py tests/test_full_search.py} -m gpu -q -x --timeout 600 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/tests.txt
fi
for sf in 1 0; do
  CCSIM_SF=$sf MB_PCT=100 MB_GATE=${MB_GATE:-1200} MB_LIMIT=${MB_LIMIT:-30000} timeout 600 python tools/bench_mode_b.py 1000000 100000 2>&1 | grep -v amdgpu.ids | sed "s/^CCSIM_SB=1/CCSIM_SF=$sf/" | tee -a $O/bench_full_search.txt | cut -c1-400
done
CCSIM_SB_PROF=1 CCSIM_SF=1 MB_PCT=100 MB_GATE=300 MB_LIMIT=${MB_LIMIT:-30000} timeout 600 python tools/bench_mode_b.py 1000000 2>&1 | grep -v amdgpu.ids | sed "s/^CCSIM_SB=1/CCSIM_SF=1 CCSIM_SB_PROF=1/" | tee $O/bench_full_search_prof.txt | cut -c1-900
# the end of a sampled run (fewer feasible nodes than the search keeps: every node visited): handed over to k_sf_cycles vs the lap kernel's one-stretch laps
for ho in 1 0; do
  CCSIM_SB_HANDOVER=$ho MB_PCT=0 MB_GATE=1000 MB_LIMIT=0 timeout 900 python tools/bench_mode_b.py 100000 2>&1 | grep -v amdgpu.ids | sed "s/^CCSIM_SB=1/whole run to Unschedulable, CCSIM_SB_HANDOVER=$ho/" | tee -a $O/bench_mode_b_whole_run.txt | cut -c1-500
done
# rocprofv3 kernel stats of the same command (the sequential mode at 1M nodes: k_sb_build once or twice, then k_sf_cycles)
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/ks && CCSIM_SF=1 MB_PCT=100 MB_GATE=200 MB_LIMIT=30000 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/tools/bench_mode_b.py 1000000 > $O/bench_full_search_under_rocprofv3.txt 2> $O/ks.err
  f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/full_search_1M_kernel_stats.csv && cut -c1-160 $O/full_search_1M_kernel_stats.csv | head -5; rm -rf $O/ks )
# the SchedulePod seam: microseconds per ccsim_schedule_one call, resident forms vs node passes
timeout 300 python tools/bench_seam.py 1000000 2>&1 | grep -v amdgpu.ids | tee $O/bench_seam.txt | cut -c1-250
CCSIM_SF=0 CCSIM_SB=0 SEAM_CALLS=1000 timeout 300 python tools/bench_seam.py 1000000 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_seam.txt | cut -c1-250
