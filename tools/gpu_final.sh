#!/bin/bash
# end-of-round sanity: smoke() + the PCIe-inclusive rate (host buffers -> HBM -> one full simulation) for DESIGN.md section 6
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pcie.log
import time
import __graft_entry__ as ge; ge.load_package()
import torch
from cluster_capacity_amd import capi, synth
n,p,f = synth.make_config("C4", n_nodes=1_000_000)
e0 = capi.Engine(device=0); e0.load(n,p,f); e0.run(max_limit=0, mode="batched", want_log=False); e0.close()  # warm: module load, allocator
for rep in range(3):
    torch.cuda.synchronize()
    t0=time.perf_counter()
    e = capi.Engine(device=0); e.load(n,p,f)          # ccsim_create + load_nodes (host -> HBM) + set_profile + set_pod
    torch.cuda.synchronize(); t1=time.perf_counter()
    r = e.run(max_limit=0, mode="batched", want_log=False)
    t2=time.perf_counter(); e.close()
    print("load %.1f ms, run %.1f ms, %d placements -> %.3e placements/s incl. the host->HBM copy (%.3e excl.)" % ((t1-t0)*1e3, (t2-t1)*1e3, r.placed, r.placed/(t2-t0), r.placed/(t2-t1)))
PY
