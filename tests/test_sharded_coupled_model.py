"""One coupled template on node-range shards (tests/sharded_coupled_model.py: replicated domain tables + three small exchanges per
cycle) against the oracle: same placements and stop for 1 .. 4 shards."""
import numpy as np
import pytest

from cluster_capacity_amd import model as M
from sharded_coupled_model import ShardedCoupledModel
from test_coupled_model import coupled_case


@pytest.mark.parametrize("ranks", [1, 2, 4])
@pytest.mark.parametrize("seed", range(30))
def test_sharded_coupled_scores_vs_oracle(ccref, seed, ranks):
    rng = np.random.default_rng(7100 + seed)
    nodes, pod, prof = coupled_case(rng, int(rng.integers(12, 160)), roomy=bool(seed % 3 == 0))
    ref = ccref.run(prof, nodes, pod, max_limit=600)
    model = ShardedCoupledModel(prof, nodes.copy(), pod, ccref.go_log, ranks)
    log, stop = model.run(600)
    assert log == ref.log.tolist(), (seed, ranks)
    assert (stop == "Unschedulable") == (ref.stop == M.STOP_UNSCHEDULABLE)
    assert model.exchanges <= 3 * (len(log) + 1)
