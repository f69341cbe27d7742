"""The one JSON line `bench.py` prints is a contract with the driver: its keys, the metric BASELINE.json names, the arithmetic that ties
`value`, `ms_per_step` and the workload together, the `roofline` and `cpu_baseline` objects.  Checked on the lines committed under
profiles/ (CPU) and on a short run of bench.py itself (GPU)."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOP = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]


def _check_line(d, full):
    for k in TOP:
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"].split(";")[0].strip() and d["unit"] == "placements/s"
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert isinstance(d["n_gpus"], int) and d["n_gpus"] >= 1 and d["steps"] >= 1 and d["ms_per_step"] > 0
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg
    # value = whole-job placements per second: the placements of one step over the time of one step
    assert d["value"] == pytest.approx(cfg["placements_per_step"] / (d["ms_per_step"] * 1e-3), rel=0.02)
    if "roofline" in d and d["roofline"]:
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, k
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-6) and 0 < r["frac"] < 1
        if r["traffic"] is not None:  # PMC bytes of one launch over the launch's duration = the achieved rate
            assert r["achieved"] == pytest.approx(r["traffic"] / (r["us_per_launch"] * 1e-6) / 1e9, rel=0.02)
        assert r["us_per_launch"] * 1e-3 <= d["ms_per_step"] * 1.05  # the dominant kernel fits inside the step it is part of
    if full:
        c = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
        assert c["log_equals_engine_prefix"] is True  # (the oracle's log is the engine's: the baseline ran the same simulation)
        assert "error" not in (d.get("secondary") or {}), d["secondary"]
        for name, sec in (d.get("secondary") or {}).items():  # config 5 / the coupled template, timed in the same process, checked first
            assert sec["unit"] == "placements/s" and sec["value"] > 0 and sec["placements"] > 0, name
            assert any(k.endswith("_equal_oracle") and v is True for k, v in sec.items()), name


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[34]", "**", "bench_1M.json"), recursive=True)))
def test_committed_bench_lines_keep_the_contract(path):
    _check_line(json.load(open(path)), full=True)


def test_a_committed_pmc_file_was_collected_with_the_sources_in_the_tree():
    """bench.py accepts a profiles/<round>/pmc_traffic.json only when it was collected with the sources it runs: some committed file must
    be stamped with the hash of the tree's csrc/ + include/, or the driver's line says `traffic: null`."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    from cluster_capacity_amd import build as b
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_traffic.json")), reverse=True)  # (bench.py takes the newest that matches)
    have = {os.path.relpath(f, ROOT): json.load(open(f)).get("src_sha16") for f in files}
    assert b.source_sha16() in have.values(), (f"no profiles/rNN/pmc_traffic.json was collected with the tree's sources ({b.source_sha16()}; have {have}): "
                                                "re-run tools/gpu_round_profile.sh <round> skip-suite on the GPU and commit profiles/<round>/pmc_traffic.json")


def _recorder(tmp_path):
    rec = tmp_path / "libabi_recorder.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", str(rec), os.path.join(ROOT, "tests", "abi_recorder.c")])
    return dict(os.environ, CCSIM_LIB=str(rec), CCSIM_RECORD=str(tmp_path / "rec.json"), CCSIM_RECORD_PER_DEVICE="1", OMP_NUM_THREADS="1")


def _strip_launcher_env(env):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("gpus", [2, 3])
def test_bench_gpus_n_without_a_launcher_starts_n_ranks(tmp_path, gpus):
    """`python bench.py --gpus N` parsed the flag and ran on one GPU.  Now it launches itself as N ranks (torch.distributed.run
    on 127.0.0.1) and the line says how many ranks took part -- both by the job (`n_gpus`) and by the communicator the library built
    (`rccl_ranks_seen`, ccsim_dist_comm_size).  On the CPU: tests/abi_recorder.c stands in for libccsim.so and gloo for RCCL; every rank's
    record must show ITS shard of the one snapshot and the communicator of N ranks."""
    env = _strip_launcher_env(_recorder(tmp_path))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--nodes", "2000", "--no-cpu", "--no-roofline", "--seq-rounds", "0",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    _check_line(d, full=False)
    assert d["n_gpus"] == gpus and d["rccl_ranks_seen"] == gpus and d["scaling"] == "strong" and "invalid" in d
    per = -(-2000 // gpus)
    for g in range(gpus):
        r = json.load(open(f"{tmp_path}/rec.json.{g}"))
        assert r["dist_comm_init"] == {"n_ranks": gpus, "rank": g, "id_ok": 1}
        assert r["nodes"]["n_global"] == 2000 and r["nodes"]["global_offset"] == g * per and r["nodes"]["n_nodes"] == min(2000, g * per + per) - g * per


def test_bench_refuses_a_gpus_flag_that_disagrees_with_the_launcher(tmp_path):
    env = _recorder(tmp_path)
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nodes", "2000", "--no-cpu"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=env)
    assert out.returncode == 2 and "--gpus 2" in out.stderr and not out.stdout.strip()


def test_bench_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    env = _strip_launcher_env(dict(os.environ))
    env.pop("CCSIM_LIB", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nodes", "2000"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 3 and "no GPU" in out.stderr and not out.stdout.strip()


@pytest.mark.gpu
def test_bench_gpus_1_is_a_plain_one_gpu_run():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-variants", "--seq-rounds", "0",
                          "--nodes", "100000"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=_strip_launcher_env(dict(os.environ)))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][0])
    _check_line(d, full=False)
    assert d["n_gpus"] == 1 and d["rccl_ranks_seen"] is None and "invalid" not in d


@pytest.mark.gpu
def test_bench_prints_one_json_line_on_stdout():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--no-variants", "--seq-rounds", "0",
                          "--nodes", "100000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    _check_line(d, full=False)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
