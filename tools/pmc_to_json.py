#!/usr/bin/env python
"""rocprofv3 PMC passes (one counter_collection.csv per counter set) -> per-kernel HBM bytes per launch (JSON).

    python tools/pmc_to_json.py <fetch.csv> <write.csv> > pmc_traffic.json

hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md
prescribes for gfx950 (128-byte requests are tallied as 64 B); the factor is calibrated for wide coalesced reads, so the
figure of kernels dominated by sparse accesses (k_level_commit) is indicative."""
import collections
import csv
import json
import os
import re
import sys


def averages(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0, 0.0]))
    for r in csv.DictReader(open(path)):
        m = re.search(r"ccsim::(k_\w+)", r["Kernel_Name"])
        if not m:
            continue
        a = acc[m.group(1)][r["Counter_Name"]]
        v = float(r["Counter_Value"])
        a[0] += v
        a[1] += 1
        a[2] = max(a[2], v)
    return acc


def _source_sha16():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge

    ge.load_package()
    from cluster_capacity_amd import build as b

    return b.source_sha16()


def main():
    fetch, write = averages(sys.argv[1]), averages(sys.argv[2])
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, {}).get("FETCH_SIZE", [0.0, 0, 0.0])
        w = write.get(k, {}).get("WRITE_SIZE", [0.0, 0, 0.0])
        if not f[1] or not w[1]:
            continue
        fk, wk = f[0] / f[1], w[0] / w[1]
        out[k] = {"launches": f[1], "FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1),
                  "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
                  "hbm_bytes_largest_launch": int((2 * f[2] + w[2]) * 1024)}
    import hashlib, os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cluster-capacity_amd", "csrc", "libccsim.so")
    json.dump({
        "lib_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],  # bench.py refuses traffic collected with another build
        "src_sha16": _source_sha16(),  # ... where "build" means the sources + flags (the binary embeds its build directory)
        "workload": "bench.py C4 1,000,000 nodes, 1 GPU, batched mode: one timed step, the full-pass train (200 x k_level_score), 64 sequential cycles",
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_pmc.sh); hbm_bytes_per_launch = "
                "(2*FETCH_SIZE + WRITE_SIZE) KB, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); averages over ALL "
                "launches of the run incl. the no-op launches (passes after the done flag is set, score-only graph heads); "
                "hbm_bytes_largest_launch = the same from the per-counter maxima: for kernels that mostly no-op (k_level_score, "
                "k_rows_flush: ~3 real launches per run) it is the traffic of a real launch",
        "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
