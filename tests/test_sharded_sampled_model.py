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


@pytest.mark.parametrize("ranks", [1, 2, 3, 5])
@pytest.mark.parametrize("seed", range(16))
def test_sharded_sampled_search_of_a_coupled_template_vs_oracle(ccref, seed, ranks):
    """Round 6: the same protocol with topology-coupled plugins (tests/sharded_sampled_model.py::ShardedSampledCoupledModel: replicated
    PreFilter tables, PreScore facts over the selected nodes only) -- the cases of
    tests/test_sampling.py::test_sampled_search_on_shards_with_topology_coupled_plugins, which holds the HIP engine against the same oracle."""
    from sharded_sampled_model import ShardedSampledCoupledModel
    rng = np.random.default_rng(8800 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(100, 1200)))
    kind = seed % 4
    if kind in (0, 1, 3):
        pod.spread = H.random_spread(rng, nodes, n_constraints=int(rng.integers(1, 3)))
        if kind == 3:
            for k in pod.spread:
                k.hard = bool(rng.integers(0, 2))
            pod.spread[-1].hard = False
    if kind in (1, 2) or (kind == 3 and seed % 8 == 3):
        pod.ipa = H.random_ipa(rng, nodes)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=int(rng.choice([0, 10, 35, 70])))
    limit = int(rng.choice([60, 150]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    e_nodes, e_pod = M.relax_soft(nodes, pod)
    model = ShardedSampledCoupledModel(prof, e_nodes.copy(), e_pod, ccref.go_log, ranks)
    log, stop, visited = model.run(limit)
    assert log == ref.log.tolist(), (seed, ranks)
    assert (stop == "Unschedulable") == (ref.stop == M.STOP_UNSCHEDULABLE)
    assert sum(visited) == ref.evaluated_total, (sum(visited), ref.evaluated_total)
    assert model.exchanges <= 4 * (len(log) + 1)
