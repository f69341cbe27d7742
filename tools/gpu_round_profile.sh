#!/bin/bash
# round-end evidence on the FINAL library: the whole -m gpu suite, the bench line, rocprofv3 kernel stats of the same command,
# PMC traffic (separate passes), the sequential cycle, the coupled windowed mode, config 5, the sharded protocol at one rank,
# smoke().  Everything lands in gpurun_out/$R/ ; copy what is to be judged into profiles/$R/.
#   tools/gpu_round_profile.sh [round tag, default r04] [skip-suite]
exec < /dev/null
R=${1:-r04}
mkdir -p /root/repo/profiles/$R
cd /root/repo
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
if [ "$2" != "skip-suite" ]; then
  ( cat $O/lib_hash.txt; echo "python -m pytest tests -m gpu -q --timeout 900"; \
    timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=8 2>&1 | grep -v "amdgpu.ids" | tail -25 ) | tee $O/gpu_tests.txt | tail -6
fi
# PMC traffic (separate rocprofv3 passes, --kernel-trace only): bench.py accepts profiles/$R/pmc_traffic.json only if it was
# collected with the sources it runs
bash tools/gpu_pmc.sh $R "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" 2>&1 | tee $O/pmc_summary.txt | tail -24
cp gpurun_out/pmc_traffic_$R.json $O/pmc_traffic.json 2>/dev/null && cp gpurun_out/pmc_traffic_$R.json profiles/$R/pmc_traffic.json
rm -rf gpurun_out/pmc_${R}_1 gpurun_out/pmc_${R}_2 gpurun_out/pmc_${R}_3 gpurun_out/pmc_${R}_4
cd /root/repo
timeout 400 python bench.py > $O/bench_1M.json 2> $O/bench_1M.err; tail -c 300 $O/bench_1M.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/bench.py --no-cpu --no-variants --seq-rounds 0 > $O/bench_1M_under_rocprofv3.json 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_1M_kernel_stats.csv && cut -c1-200 $O/bench_1M_kernel_stats.csv | head -8
rm -rf $O/ks
# the sequential cycle (one dispatch: k_scan_fused)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/bench.py --mode sequential --no-cpu --no-variants --no-roofline --steps 3 > $O/bench_seq_1M.json 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/seq_1M_kernel_stats.csv && cut -c1-200 $O/seq_1M_kernel_stats.csv | head -5
rm -rf $O/ks
cd /root/repo
# config 5 (1024 pod specs): throughput line
timeout 300 python tools/bench_c5.py 100000 1024 200000 128,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_c5.txt | cut -c1-300
( cd /tmp && rm -rf $O/ks && CCSIM_MULTI_MEMO_MB=65536 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/tools/bench_c5.py 100000 1024 100000 128 > /dev/null 2> $O/ks.err
  f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c5_kernel_stats.csv && cut -c1-200 $O/c5_kernel_stats.csv | head -8; rm -rf $O/ks )
( cd /tmp && for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    rm -rf $O/pmc_c5; CCSIM_MULTI_MEMO_MB=65536 timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_c5 -o p -- python /root/repo/tools/bench_c5.py 100000 1024 20000 128 > /dev/null 2> $O/pmc_c5.err
    f=$(find $O/pmc_c5 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 /root/repo/tools/pmc_summary.py "$f" | grep multi; rm -rf $O/pmc_c5
  done ) 2>&1 | tee $O/c5_pmc_summary.txt | cut -c1-250
timeout 120 python tools/persist_prof.py 1000000 8 1024 2>&1 | grep -v amdgpu.ids | tee $O/persist_phase_profile.txt | cut -c1-330
CCSIM_PERSIST_SPEC=0 timeout 120 python tools/persist_prof.py 1000000 8 1024 2>&1 | grep -v amdgpu.ids | sed "s/^/CCSIM_PERSIST_SPEC=0 (round 3 form: stop above the event, that level ordered): /" | tee -a $O/persist_phase_profile.txt | cut -c1-200
timeout 120 python tools/persist_prof.py 1000000 8 64,192,384,1024,4096 2>&1 | grep -v amdgpu.ids | cut -c1-330 > $O/persist_batch_sweep.txt
timeout 120 python tools/step_breakdown.py 2>&1 | grep -v amdgpu.ids | tee $O/step_breakdown.txt | cut -c1-250
CCSIM_BENCH_SKIP_SEQ=1 timeout 200 python tools/bench_coupled.py 1000000 50000 1024,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_1M_64zones.txt | cut -c1-300
timeout 200 python tools/bench_coupled.py 100000 50000 1024,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_100k.txt | cut -c1-300
CCSIM_FORCE_DIST=1 CCSIM_DIST_DEBUG=1 timeout 120 python bench.py --no-variants --seq-rounds 0 --steps 3 2>$O/bench_dist_world1.err > $O/bench_dist_world1_mailbox.json; cut -c1-200 $O/bench_dist_world1_mailbox.json; grep "ccsim dist" $O/bench_dist_world1.err | tail -3
CCSIM_FORCE_DIST=1 CCSIM_DIST_MAILBOX=0 timeout 120 python bench.py --no-variants --seq-rounds 0 --steps 3 2>/dev/null > $O/bench_dist_world1_rccl_passes.json; cut -c1-200 $O/bench_dist_world1_rccl_passes.json
( cat $O/lib_hash.txt; timeout 600 python -m pytest tests/test_dist_mailbox.py -m gpu -q -s 2>&1 | grep -E "^\[mailbox\]|two processes|passed|failed" ) | tee $O/mailbox_forms_taken.txt | tail -8
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
# round 6: mode B (the reference's default percentage) -- an uncoupled template a lap of the ring at a time, a template with a hard zone constraint resident
bash tools/gpu_mode_b_prof.sh $R/mode_b > /dev/null 2>&1; cp $O/mode_b/mode_b_1M_kernel_stats.csv $O/mode_b/pmc_mode_b.txt $O/mode_b/bench_mode_b_under_rocprofv3.txt $O/ 2>/dev/null; head -3 $O/mode_b_1M_kernel_stats.csv | cut -c1-160
SKIP_TESTS=1 bash tools/gpu_sb6.sh $R/sb > /dev/null 2>&1; cp $O/sb/bench_mode_b.txt $O/bench_mode_b_laps_vs_cycle_at_a_time.txt; cp $O/sb/bench_mode_b_prof.txt $O/mode_b_laps_phase_profile.txt; grep "CCSIM_SB=1" $O/bench_mode_b_laps_vs_cycle_at_a_time.txt | cut -c1-140
SKIP_TESTS=1 bash tools/gpu_sz6.sh $R/sz > /dev/null 2>&1; cp $O/sz/bench_mode_b_zone.txt $O/bench_mode_b_zone.txt; cut -c1-150 $O/bench_mode_b_zone.txt
# round 6: the full search on the block summaries by one wave (the sequential mode, the SchedulePod seam), the end of a sampled run handed over to it
SKIP_TESTS=1 bash tools/gpu_sf6.sh $R/sf > /dev/null 2>&1; cp $O/sf/bench_full_search.txt $O/sf/bench_full_search_prof.txt $O/sf/bench_mode_b_whole_run.txt $O/sf/full_search_1M_kernel_stats.csv $O/sf/bench_seam.txt $O/ 2>/dev/null; cut -c1-170 $O/bench_full_search.txt; cut -c1-200 $O/bench_mode_b_whole_run.txt
