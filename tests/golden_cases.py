"""Seeded inputs of the golden vectors (tests/golden/cases.json) and the result summary they store."""
import hashlib

import numpy as np

import helpers as H
from cluster_capacity_amd import model as M, report as R, synth

CASES = {
    "ka1_unlimited": ("ka1", 0), "ka1_limit6": ("ka1", 6), "ka2_readme4": ("readme", 4, 0), "readme2_limit5": ("readme", 2, 5),
    "c2_n1000": ("synth", "C2", 1000, 11, 0), "c3_n1000": ("synth", "C3", 1000, 12, 0), "c3_n4096_l700": ("synth", "C3", 4096, 13, 700),
    "c3_n1": ("synth", "C3", 1, 14, 0), "c3_n513_l50": ("synth", "C3", 513, 15, 50),
    **{f"random_{s}": ("random", s) for s in range(8)},
    **{f"random_ports_images_{s}": ("random_pi", s) for s in range(4)},
}


def build(name):
    spec = CASES[name]
    if spec[0] == "ka1":
        return H.test_prediction_nodes(), H.test_prediction_pod(), M.Profile.default(), spec[1]
    if spec[0] == "readme":
        return H.readme_nodes(spec[1]), H.examples_pod(), M.Profile.default(), spec[2]
    if spec[0] == "synth":
        nodes, pod, prof = synth.make_config(spec[1], n_nodes=spec[2], seed=spec[3])
        return nodes, pod, prof, spec[4]
    rng = np.random.default_rng((1000 if spec[0] == "random" else 2000) + spec[1])
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 1200)))
    if spec[0] == "random_pi":
        nodes, pod, prof = H.with_ports_and_images(rng, nodes, pod, prof)
    return nodes, pod, prof, int(rng.choice([0, 0, 37, 500]))


def summarize(r, n_nodes, limit):
    log = np.asarray(r.log, dtype=np.int32)
    return {
        "placed": int(r.placed), "stop": int(r.stop),
        "per_node_count_sha256": hashlib.sha256(np.asarray(r.per_node_count, np.int32).tobytes()).hexdigest(),
        "per_node_count_head": [int(x) for x in r.per_node_count[:16]],
        "log_sha256": hashlib.sha256(log.tobytes()).hexdigest(), "log_head": [int(x) for x in log[:16]],
        # (the slots up to NodePorts, as the file was made; the volume plugins' slots added by ABI 4 follow them and no golden case has a volume)
        "hist": [int(x) for x in r.hist[: M.R_NODEPORTS + 1]] if r.stop == M.STOP_UNSCHEDULABLE else None,
        "hist_taintset": [int(x) for x in r.hist_taintset] if r.stop == M.STOP_UNSCHEDULABLE else None,
        "stop_reason": R.stop_reason(r, n_nodes, limit),
    }
