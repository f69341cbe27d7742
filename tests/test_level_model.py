"""The exactness argument of CCSIM_MODE_BATCHED, checked on the CPU: the Python restatement of the
level algorithm (tests/level_model.py, mirroring ccsim_level.h) must reproduce the sequential oracle's
placement sequence -- same log, same per-node counts, same stop -- on the cases the GPU tests use."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import model as M, synth
from level_model import LevelModel


def _check(ccref, nodes, pod, prof, limit):
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    got = LevelModel(prof, nodes, pod).run(limit)
    assert got["placed"] == ref.placed and got["stop"] == ref.stop
    assert np.array_equal(got["per_node_count"], ref.per_node_count)
    assert np.array_equal(got["log"], ref.log)
    return got


@pytest.mark.parametrize("seed", range(12))
def test_level_model_random_plugin_mix(ccref, seed):
    rng = np.random.default_rng(seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 400)))
    _check(ccref, nodes, pod, prof, int(rng.choice([0, 0, 37, 500])))


@pytest.mark.parametrize("cfg,n,limit", [("C2", 300, 0), ("C3", 300, 0), ("C3", 513, 700), ("C3", 1, 0)])
def test_level_model_synthetic(ccref, cfg, n, limit):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=1234 + n)
    got = _check(ccref, nodes, pod, prof, limit)
    assert got["levels"] <= got["placed"] + 1  # a level never holds less than one placement


def test_level_model_known_answers(ccref):
    _check(ccref, H.test_prediction_nodes(), H.test_prediction_pod(), M.Profile.default(), 0)
    _check(ccref, H.test_prediction_nodes(), H.test_prediction_pod(), M.Profile.default(), 6)
    _check(ccref, H.readme_nodes(4), H.examples_pod(), M.Profile.default(), 0)


def _check_incremental(ccref, nodes, pod, prof, limit):
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    got = LevelModel(prof, nodes, pod).run_incremental(limit)
    assert got["placed"] == ref.placed and got["stop"] == ref.stop
    assert np.array_equal(got["per_node_count"], ref.per_node_count)
    assert np.array_equal(got["log"], ref.log)
    return got


@pytest.mark.parametrize("seed", range(12))
def test_incremental_score_cache_random_plugin_mix(ccref, seed):
    """The score cache argument: after a level is committed only its own nodes have new scores, the feasible and holder
    counts can be kept by subtraction, and a full pass is needed only when a maximum loses its last feasible holder."""
    rng = np.random.default_rng(seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 400)))
    got = _check_incremental(ccref, nodes, pod, prof, int(rng.choice([0, 0, 37, 500])))
    assert got["full_passes"] <= 2 + 2 * 6  # first pass (+ its rescan), then at most one pair per distinct maximum


@pytest.mark.parametrize("cfg,n,limit", [("C2", 300, 0), ("C3", 300, 0), ("C3", 513, 700), ("C3", 1, 0)])
def test_incremental_score_cache_synthetic(ccref, cfg, n, limit):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=1234 + n)
    got = _check_incremental(ccref, nodes, pod, prof, limit)
    assert got["full_passes"] < got["levels"] or got["levels"] <= 2


def _check_persistent(ccref, nodes, pod, prof, limit, batch, spec=True):
    ref = ccref.run(prof, nodes, pod, max_limit=limit, want_log=False)
    got = LevelModel(prof, nodes, pod).run_persistent(limit, batch, spec=spec)
    assert got["placed"] == ref.placed and got["stop"] == ref.stop
    assert np.array_equal(got["per_node_count"], ref.per_node_count)
    return got


@pytest.mark.parametrize("spec", [True, False])
@pytest.mark.parametrize("batch", [1, 4, 64, 1024])
@pytest.mark.parametrize("seed", range(12))
def test_persistent_level_batches_random_plugin_mix(ccref, seed, batch, spec):
    """The persistent kernel's argument (ccsim_persist.h): several score levels per sync committed blindly + validation +
    roll-back give the sequential oracle's totals, per-node counts and stop -- incl. limits falling inside a batch and
    normalization maxima losing their last feasible holder inside one."""
    rng = np.random.default_rng(seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 400)))
    _check_persistent(ccref, nodes, pod, prof, int(rng.choice([0, 0, 37, 500])), batch, spec)


@pytest.mark.parametrize("seed", range(40))
def test_persistent_batches_that_end_at_the_event(ccref, seed):
    """Round 4: a blind batch ends exactly where a normalization maximum loses its last feasible holder (level + node predicted by
    the re-score, per-node threshold, validated by the holders that filled up).  Few holders of the maxima, so that the events fall
    inside batches; with and without a limit."""
    rng = np.random.default_rng(900 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(20, 500)))
    limit = int(rng.choice([0, 0, 0, 211]))
    got = _check_persistent(ccref, nodes, pod, prof, limit, int(rng.choice([16, 64, 1024])))
    old = _check_persistent(ccref, nodes, pod, prof, limit, 64, spec=False)
    assert np.array_equal(got["per_node_count"], old["per_node_count"])


def test_batches_that_end_at_the_event_save_syncs(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=1000, seed=7)
    new = _check_persistent(ccref, nodes, pod, prof, 0, 1024)
    old = _check_persistent(ccref, nodes, pod, prof, 0, 1024, spec=False)
    assert new["spec_ok"] >= 1 and new["syncs"] < old["syncs"], (new["syncs"], old["syncs"], new["spec_ok"])


def test_persistent_level_batches_save_syncs_and_roll_back(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=513, seed=1747)
    one = _check_persistent(ccref, nodes, pod, prof, 0, 1)
    many = _check_persistent(ccref, nodes, pod, prof, 0, 64)
    assert many["syncs"] < one["syncs"] / 4
    limited = _check_persistent(ccref, nodes, pod, prof, 700, 64)  # the limit falls inside a batch: rolled back, halved, ...
    assert limited["rollbacks"] >= 1 and limited["placed"] == 700
