#!/bin/bash
# usage: tools/gpu_variant.sh "<extra hipcc flags>" <tag> : rebuild on the box with flags, run the batched bench (no cpu)
exec < /dev/null
cd /root/repo
CCSIM_EXTRA_FLAGS="$1" timeout 200 python cluster-capacity_amd/build.py > /dev/null 2>&1
bash tools/gpu_prof.sh $2 --steps 2 --warmup 1 --no-cpu --seq-rounds 0 2>&1 | grep -E "k_level|value" | cut -c1-200
