#!/bin/bash
exec < /dev/null
cd /root/repo
for g in "$@"; do
  echo "scan grid $g"
  CCSIM_SCAN_GRID=$g bash tools/gpu_prof.sh sg$g --steps 2 --warmup 1 --no-cpu 2>&1 | grep -E "k_level_score|k_scan|k_final|k_level_final|value" | cut -c1-150
done
