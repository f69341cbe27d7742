#!/bin/bash
# world_size-1 RCCL self-test of the sharded path (one GPU box): torchrun -> make_torch_runner -> all_gather per pass
exec < /dev/null
cd /root/repo
export CCSIM_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu --seq-rounds 256 > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err
echo "rc=$?"; tail -3 gpurun_out/bench_dist1.err | cut -c1-300; cut -c1-900 gpurun_out/bench_dist1.json
