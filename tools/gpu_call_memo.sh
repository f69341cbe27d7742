#!/bin/bash
# one GPU call: the multi-spec parity tests on the new library + config 5's throughput line, kernel stats and SQ counters of the scan
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r04
mkdir -p $O
( timeout 700 python -m pytest tests/test_multi.py tests/test_baseline_configs.py::test_c5_100k_nodes_1024_specs_prefix_vs_oracle -m gpu -q -x --timeout 500 2>&1 | grep -v amdgpu.ids | tail -5 ) | tee $O/gpu_multi_memo_tests.txt
timeout 300 python tools/bench_c5.py 100000 1024 200000 64 2>&1 | grep -v amdgpu.ids | tee $O/bench_c5.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks; CCSIM_MULTI_MEMO_MB=65536 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/tools/bench_c5.py 100000 1024 100000 64 > /dev/null 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c5_kernel_stats.csv && cut -c1-200 $O/c5_kernel_stats.csv | head -6; rm -rf $O/ks
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rm -rf $O/pmc_c5_$i
  CCSIM_MULTI_MEMO_MB=65536 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_c5_$i -o p -- python /root/repo/tools/bench_c5.py 100000 1024 20000 64 > /dev/null 2> $O/pmc_c5.err
  f=$(find $O/pmc_c5_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 /root/repo/tools/pmc_summary.py "$f" | grep multi; else echo "no counter csv ($set)"; tail -3 $O/pmc_c5.err; fi
  rm -rf $O/pmc_c5_$i
done 2>&1 | tee $O/c5_pmc_summary.txt
