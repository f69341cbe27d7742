"""The resident form of the sampled search (tests/sampled_resident_model.py = the argument of csrc/ccsim_sampled.h) against the oracle's
literal visiting loop on the CPU: same placements, same nodes visited (the start index trajectory), with blocks small enough that every
mode of the stretch's end occurs -- inside the start block on either side of the start index, in a whole block, nowhere."""
import dataclasses

import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import model as M, synth
from sampled_resident_model import ResidentSampledModel


@pytest.mark.parametrize("block", [16, 64, 256])
@pytest.mark.parametrize("seed", range(12))
def test_resident_sampled_search_vs_oracle(ccref, seed, block):
    rng = np.random.default_rng(9100 + seed)
    nodes, pod, prof = H.with_ports_and_images(rng, *H.random_case(rng, int(rng.integers(100, 600))))
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=int(rng.choice([0, 10, 35, 70, 99])))
    limit = int(rng.choice([0, 0, 150]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    model = ResidentSampledModel(prof, nodes.copy(), pod, block=block)
    log, stop, visited, starts = model.run(limit)
    assert log == ref.log.tolist(), (seed, block)
    assert (stop == "Unschedulable") == (ref.stop == M.STOP_UNSCHEDULABLE)
    assert visited == ref.evaluated_total  # the same nodes were visited, cycle by cycle
    assert model.builds <= 2 + len(log)  # (a rebuild only when the kept nodes' maxima moved)


def test_c3_shape_adaptive_default_every_mode_of_the_stretch_end(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=600, seed=5)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=0)  # adaptive: 50 - 600/125 = 46 % -> K = 276
    ref = ccref.run(prof, nodes, pod, max_limit=0)
    model = ResidentSampledModel(prof, nodes.copy(), pod, block=256, check=False)  # (K + 1 feasible nodes fit behind a start index: mode 1 occurs)
    log, stop, visited, starts = model.run(0)
    assert log == ref.log.tolist() and visited == ref.evaluated_total and stop == "Unschedulable"
