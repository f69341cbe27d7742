#!/bin/bash
# round 6: mode B (the reference's default percentageOfNodesToScore) at 1M nodes under rocprofv3 -- kernel stats of k_sb_laps / k_sb_build, and
# PMC passes (separate runs, --kernel-trace only): HBM traffic (FETCH_SIZE, WRITE_SIZE) and the issue mix (SQ_*) of the one workgroup.
exec < /dev/null
O=/root/repo/gpurun_out/${1:-r06mb}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks
MB_LIMIT=100000 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/tools/bench_mode_b.py 1000000 > $O/bench_mode_b_under_rocprofv3.txt 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/mode_b_1M_kernel_stats.csv && cut -c1-170 $O/mode_b_1M_kernel_stats.csv | head -8
grep -v amdgpu.ids $O/bench_mode_b_under_rocprofv3.txt | tail -2 | cut -c1-400
rm -rf $O/ks
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
  rm -rf $O/p
  MB_LIMIT=100000 MB_GATE=100 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p -o p -- python /root/repo/tools/bench_mode_b.py 1000000 > /dev/null 2> $O/p.err
  f=$(find $O/p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { echo "== mode B 1M, $set"; python3 /root/repo/tools/pmc_summary.py "$f" | grep "k_sb"; } || tail -3 $O/p.err
  rm -rf $O/p
done 2>&1 | tee $O/pmc_mode_b.txt
