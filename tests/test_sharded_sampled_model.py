"""The sampled search on node-range shards (tests/sharded_sampled_model.py: three fixed-size exchanges per cycle) against the oracle's
literal visiting loop: same placements, same stop, whatever the number of shards -- including the start index trajectory that the
cancelling (K+1)-th feasible node drives."""
import dataclasses

import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import model as M, synth
from sharded_sampled_model import ShardedSampledModel


@pytest.mark.parametrize("ranks", [1, 2, 3, 5])
@pytest.mark.parametrize("seed", range(14))
def test_sharded_sampled_search_vs_oracle(ccref, seed, ranks):
    rng = np.random.default_rng(6400 + seed)
    nodes, pod, prof = H.with_ports_and_images(rng, *H.random_case(rng, int(rng.integers(100, 700))))
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=int(rng.choice([0, 10, 35, 70, 99])))
    limit = int(rng.choice([0, 0, 120]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    model = ShardedSampledModel(prof, nodes.copy(), pod, ranks)
    log, stop, starts = model.run(limit)
    assert log == ref.log.tolist(), (seed, ranks)
    assert (stop == "Unschedulable") == (ref.stop == M.STOP_UNSCHEDULABLE)
    assert model.exchanges <= 3 * (len(log) + 1)


def test_c3_shape_with_the_adaptive_default(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=3000, seed=5)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=0)  # adaptive: 50 - 3000/125 = 26 % -> K = 780
    ref = ccref.run(prof, nodes, pod, max_limit=400)
    log, stop, starts = ShardedSampledModel(prof, nodes.copy(), pod, 8).run(400)
    assert log == ref.log.tolist() and len(set(starts)) > 100  # the start index really rotates
