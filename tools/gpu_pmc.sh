#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counters set 1>" ["<counters set 2>" ...]   -- one rocprofv3 run per set
# (PMC runs use --kernel-trace only; never sys/hip/hsa trace domains).  Bounded, no stdin reads.
# The workload is one timed step of bench.py plus its full-pass train (k_level_score x 200) and 64 sequential cycles (k_scan_fused),
# so that every kernel the bench line quotes has counters.
tag=$1; shift
exec < /dev/null
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/gpurun_out/pmc_${tag}_$i -o p -- \
     python /root/repo/bench.py --steps 1 --warmup 0 --seq-rounds 64 --no-cpu --no-variants $BENCH_EXTRA > /root/repo/gpurun_out/pmc_${tag}_$i.json 2> /root/repo/gpurun_out/pmc_${tag}_$i.err
  echo "set $i ($set): rc=$?"
  f=$(find /root/repo/gpurun_out/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 /root/repo/tools/pmc_summary.py "$f"; else echo "no counter csv"; tail -3 /root/repo/gpurun_out/pmc_${tag}_$i.err; fi
done
f1=$(find /root/repo/gpurun_out/pmc_${tag}_1 -name "*counter_collection.csv" | head -1)
f2=$(find /root/repo/gpurun_out/pmc_${tag}_2 -name "*counter_collection.csv" | head -1)
if [ -n "$f1" ] && [ -n "$f2" ]; then python3 /root/repo/tools/pmc_to_json.py "$f1" "$f2" > /root/repo/gpurun_out/pmc_traffic_${tag}.json; fi
