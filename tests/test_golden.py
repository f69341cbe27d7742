"""Golden vectors (tests/golden/cases.json, made by tests/golden/make_golden.py from the oracle):
CPU: the oracle still reproduces them.  GPU: the HIP path (both modes, through the C ABI) reproduces them."""
import json
import os

import pytest

import golden_cases
from cluster_capacity_amd import capi

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cases.json")))


def test_golden_file_covers_every_case():
    assert set(GOLD) == set(golden_cases.CASES)


@pytest.mark.parametrize("name", sorted(golden_cases.CASES))
def test_oracle_reproduces_golden(ccref, name):
    nodes, pod, prof, limit = golden_cases.build(name)
    assert golden_cases.summarize(ccref.run(prof, nodes, pod, max_limit=limit), nodes.n, limit) == GOLD[name]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sequential", "batched"])
@pytest.mark.parametrize("name", sorted(golden_cases.CASES))
def test_hip_reproduces_golden(name, mode):
    nodes, pod, prof, limit = golden_cases.build(name)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    got = e.run(max_limit=limit, mode=mode, log_cap=max(1, GOLD[name]["placed"]))
    if got.hist_taintset is not None:
        got.hist_taintset = got.hist_taintset[: len(pod.taint_filter_ok)]
    assert golden_cases.summarize(got, nodes.n, limit) == GOLD[name]
