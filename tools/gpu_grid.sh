#!/bin/bash
exec < /dev/null
cd /root/repo
for g in "$@"; do
  echo "grid $g"; CCSIM_LEVEL_GRID=$g bash tools/gpu_prof.sh g$g --steps 2 --warmup 1 --no-cpu --seq-rounds 0 2>&1 | grep -E "k_level|value" | cut -c1-160
done
