"""The lap-parallel form of the sampled search (tests/sampled_lap_model.py = the argument of k_sb_laps, csrc/ccsim_sampled.h) against the
oracle's literal visiting loop on the CPU: same placements, same nodes visited, over several laps of the ring and wraps, with blocks
small enough that stretch boundaries fall into whole blocks, into the start block before the start index, and onto a block's first
feasible node."""
import dataclasses

import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import model as M, synth
from sampled_lap_model import LapSampledModel
from sharded_sampled_model import num_feasible_nodes_to_find


def _block_for(prof, n, want):
    k = num_feasible_nodes_to_find(prof.percentage_of_nodes_to_score, n)
    b = want
    while b > k:
        b //= 2
    return b


@pytest.mark.parametrize("block", [16, 64, 256])
@pytest.mark.parametrize("seed", range(16))
def test_lap_sampled_search_vs_oracle(ccref, seed, block):
    rng = np.random.default_rng(9300 + seed)
    nodes, pod, prof = H.with_ports_and_images(rng, *H.random_case(rng, int(rng.integers(120, 2500))))
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=int(rng.choice([0, 5, 10, 35, 70, 99])))
    if num_feasible_nodes_to_find(prof.percentage_of_nodes_to_score, nodes.n) >= nodes.n:
        pytest.skip("every node is scored: not a sampled search")
    limit = int(rng.choice([0, 0, 150, 1000]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    model = LapSampledModel(prof, nodes.copy(), pod, block=_block_for(prof, nodes.n, block), max_stretches=int(rng.choice([63, 63, 3, 1])), slow_floor=int(rng.choice([0, 0, 1 << 16])))
    log, stop, visited, _ = model.run(limit)
    assert log == ref.log.tolist(), (seed, block)
    assert (stop == "Unschedulable") == (ref.stop == M.STOP_UNSCHEDULABLE)
    assert visited == ref.evaluated_total  # the same nodes were visited, cycle by cycle


def test_c3_shape_many_laps(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=2000, seed=5)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=5)  # K = 100: 19 stretches per lap while every node is feasible
    ref = ccref.run(prof, nodes, pod, max_limit=6000)
    model = LapSampledModel(prof, nodes.copy(), pod, block=64, check=False)
    log, stop, visited, _ = model.run(6000)
    assert log == ref.log.tolist() and visited == ref.evaluated_total and stop == "LimitReached"
    assert max(model.stretches_per_lap) >= 15 and model.laps < len(log) // 8  # (F - 1) // K stretches per lap, few rebuilds
    assert model.slow_stretches > 0


def test_c3_shape_to_the_end_the_degenerate_lap(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=400, seed=6)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=25)  # K = 100
    ref = ccref.run(prof, nodes, pod, max_limit=0)
    model = LapSampledModel(prof, nodes.copy(), pod, block=64, check=False)
    log, stop, visited, _ = model.run(0)
    assert log == ref.log.tolist() and visited == ref.evaluated_total and stop == "Unschedulable"
    assert model.stretches_per_lap[-1] <= 1  # the run ended on one-stretch laps (F <= K: every node visited)
