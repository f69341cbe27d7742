#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, --kernel-trace only) of the coupled windowed mode's kernels at 1M nodes /
# 64 zones and of config 5's kernels; kernel stats of the coupled run with 4096-cycle windows on the same library.
exec < /dev/null
O=/root/repo/gpurun_out/${1:-r05pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $O/p
  CCSIM_BENCH_SKIP_SEQ=1 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p -o p -- python /root/repo/tools/bench_coupled.py 1000000 50000 4096,64 > /dev/null 2> $O/p.err
  f=$(find $O/p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { echo "== coupled 1M / 64 zones, $set"; python3 /root/repo/tools/pmc_summary.py "$f" | grep "k_cw"; }
  rm -rf $O/p
  CCSIM_MULTI_MEMO_MB=65536 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p -o p -- python /root/repo/tools/bench_c5.py 100000 1024 50000 128 > /dev/null 2> $O/p.err
  f=$(find $O/p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { echo "== config 5 (100k x 1024, memo on), $set"; python3 /root/repo/tools/pmc_summary.py "$f" | grep "k_multi"; }
  rm -rf $O/p
done 2>&1 | tee $O/pmc_cw_c5.txt
rm -rf $O/ks
CCSIM_BENCH_SKIP_SEQ=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/tools/bench_coupled.py 1000000 200000 4096,64 > $O/bench_coupled_1M_w4096.txt 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cw_1M_64zones_kernel_stats_w4096.csv && cut -c1-170 $O/cw_1M_64zones_kernel_stats_w4096.csv | head -10
grep -v amdgpu.ids $O/bench_coupled_1M_w4096.txt | tail -3 | cut -c1-300
rm -rf $O/ks
