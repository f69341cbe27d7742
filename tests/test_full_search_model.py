"""The full search on resident summaries with a held winner and streaks of wins (tests/full_search_model.py = the argument of k_sf_cycles,
csrc/ccsim_search_full.h) against the oracle's literal loop on the CPU: same placements, same stop, same nodes visited; blocks and groups
small enough that winners change block and group all the time, streak widths from 1 (a cycle at a time) to 64."""
import dataclasses

import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import model as M, synth
from full_search_model import FullSearchModel


@pytest.mark.parametrize("block,group,lanes", [(4, 2, 64), (16, 8, 5), (64, 64, 64), (8, 4, 1)])
@pytest.mark.parametrize("seed", range(12))
def test_full_search_model_vs_oracle(ccref, seed, block, group, lanes):
    rng = np.random.default_rng(7700 + seed)
    nodes, pod, prof = H.with_ports_and_images(rng, *H.random_case(rng, int(rng.integers(1, 900))))
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=100)
    limit = int(rng.choice([0, 0, 90, 700]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    model = FullSearchModel(prof, nodes.copy(), pod, block=block, group=group, lanes=lanes)
    log, stop, visited = model.run(limit)
    assert log == ref.log.tolist(), (seed, block, group, lanes)
    assert (stop == "Unschedulable") == (ref.stop == M.STOP_UNSCHEDULABLE)
    assert visited == ref.evaluated_total


def test_c4_shape_streaks(ccref):
    # BASELINE's C4 shape: the emptiest node of a cluster wins until its score has come down to the next one's -- several cycles per
    # evaluation; nodes fill up and leave (the maxima they held are recomputed), to the end of the run
    nodes, pod, prof = synth.make_config("C4", n_nodes=300, seed=3)
    ref = ccref.run(prof, nodes, pod, max_limit=0)
    model = FullSearchModel(prof, nodes.copy(), pod, block=16, group=4, lanes=64, check=False)
    log, stop, visited = model.run(0)
    assert log == ref.log.tolist() and stop == "Unschedulable" and visited == ref.evaluated_total
    assert model.evaluations < 0.7 * len(log) and model.node_changes <= model.evaluations
    assert model.gone > 0 and model.block_changes > 10


def test_one_node_and_no_feasible_node(ccref):
    nodes, pod, prof = H.readme_nodes(1), H.examples_pod(), M.Profile.default()
    ref = ccref.run(prof, nodes, pod, max_limit=0)
    log, stop, visited = FullSearchModel(prof, nodes.copy(), pod, block=4, group=2).run(0)
    assert log == ref.log.tolist() and stop == "Unschedulable" and visited == ref.evaluated_total
    big = dataclasses.replace(pod, req=np.array([10**9, 0, 0]))
    ref = ccref.run(prof, nodes, big, max_limit=0)
    log, stop, visited = FullSearchModel(prof, nodes.copy(), big, block=4, group=2).run(0)
    assert log == [] and ref.placed == 0 and stop == "Unschedulable" and visited == ref.evaluated_total


@pytest.mark.parametrize("seed", range(10))
def test_sampled_search_hands_over_to_the_full_search(ccref, seed):
    """percentageOfNodesToScore < 100 to the end of the run: once fewer feasible nodes are left than the search wants to keep every node is
    visited (schedule_one.go:538: processed = N, the start index stays) -- the lap model runs until then, this model takes over on the
    same node state with the visiting order (and the tie-break) starting at that index."""
    from sampled_lap_model import LapSampledModel
    rng = np.random.default_rng(8800 + seed)
    nodes, pod, prof = H.with_ports_and_images(rng, *H.random_case(rng, int(rng.integers(300, 1500))))
    if seed % 2:  # equal nodes: ties between the two parts of the ring decide
        nodes, pod, prof = synth.make_config("C3", n_nodes=int(rng.integers(300, 900)), seed=seed)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=int(rng.choice([20, 35, 60])))
    ref = ccref.run(prof, nodes, pod, max_limit=0)
    lap = LapSampledModel(prof, nodes.copy(), pod, block=16, check=False)
    log, visited = [], 0
    while lap.Ftotal > lap.K:
        out, flag = lap.lap(0)
        assert flag is not None
        for g, v in out:
            log.append(g), (visited := visited + v)
    assert lap.Ftotal > 0 and log == ref.log[:len(log)].tolist()
    full = FullSearchModel(prof, nodes, pod, block=16, group=4, lanes=64, start=lap.start, node_model=lap.m, assumed=(lap.mt_a, lap.ma_a))
    log2, stop, visited2 = full.run(0)
    assert log + log2 == ref.log.tolist(), seed
    assert stop == "Unschedulable" and visited + visited2 == ref.evaluated_total
    assert lap.start > 0 or seed % 2 == 0
