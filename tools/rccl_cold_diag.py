"""Cold-box diagnosis of the native host's sharded path: time the plain run and the --force-sharded run
(one RCCL rank) of cluster-capacity-native, first call on a fresh box, with the library's own timing lines
(CCSIM_DIST_DEBUG=1) and RCCL's (NCCL_DEBUG=INFO).  Usage (GPU box): python tools/rccl_cold_diag.py > gpurun_out/rccl_cold.txt"""
import json, os, subprocess, sys, tempfile, time, pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT), sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge

ge.load_package()
import test_native_host as T
from cluster_capacity_amd import build as B

native = B.host_path()
tmp = pathlib.Path(tempfile.mkdtemp())
nodes, pods, pod, exclude = T.CASES["readme"]()
podspec, snaps = T._write(tmp, "json", nodes, pods, pod)
args = [native, "--podspec", podspec, "--snapshot", snaps[0], "-o", "json"]
for label, extra, env in (("plain (cold)", [], {}), ("sharded (cold RCCL)", ["--force-sharded"], {"NCCL_DEBUG": "INFO"}),
                          ("sharded (warm)", ["--force-sharded"], {}), ("plain (warm)", [], {})):
    t0 = time.time()
    try:
        p = subprocess.run(args + extra, capture_output=True, text=True, timeout=900, env=dict(os.environ, CCSIM_DIST_DEBUG="1", **env))
        rc, err = p.returncode, p.stderr
    except subprocess.TimeoutExpired as ex:
        rc, err = "TIMEOUT", (ex.stderr or b"").decode(errors="replace") if isinstance(ex.stderr, bytes) else str(ex.stderr)
    print(f"== {label}: rc={rc} wall={time.time() - t0:.2f}s")
    print("\n".join(l for l in err.splitlines() if "[ccsim" in l or "NCCL" in l or "error" in l.lower())[-3000:])
    sys.stdout.flush()
