"""The exactness argument of the several-pod-specs path (csrc/ccsim_multi.h), checked on the CPU: the Python restatement of the
window algorithm (tests/window_model.py: scan against S0, candidate lists with hidden-node bounds, in-order commit or
assign + verify) must reproduce the oracle's round-robin loop -- same log, same stop, same failing spec -- whatever the window
size, the tile size and the commit variant."""
import numpy as np
import pytest

from cluster_capacity_amd import model as M
from test_multi import random_multi_case
from window_model import WindowModel


@pytest.mark.parametrize("parallel", [False, True])
@pytest.mark.parametrize("window,tile", [(1, 16), (5, 4), (64, 16), (64, 2)])
@pytest.mark.parametrize("seed", range(10))
def test_window_model_vs_oracle(ccref, seed, window, tile, parallel):
    rng = np.random.default_rng(600 + seed)
    nodes, pods, prof = random_multi_case(rng, int(rng.integers(20, 200)), int(rng.integers(2, 24)))
    limit = int(rng.choice([0, 0, 37]))
    ref = ccref.run_multi(prof, nodes, pods, max_limit=limit)
    got = WindowModel(prof, nodes, pods, tile=tile, topk=8 if tile > 2 else 3, window=window).run(limit, parallel=parallel)
    assert got["placed"] == ref.placed and got["stop"] == ref.stop and got["stop_spec"] == ref.stop_spec
    assert np.array_equal(got["log"], ref.log)


def test_windows_end_early_and_still_match(ccref):
    """Small tiles and a short candidate list force the bounds to end windows early (hidden nodes, exhausted lists): the result
    still equals the oracle's, only the number of windows grows."""
    rng = np.random.default_rng(4242)
    nodes, pods, prof = random_multi_case(rng, 60, 12)
    ref = ccref.run_multi(prof, nodes, pods)
    wide = WindowModel(prof, nodes, pods, tile=16, topk=8, window=64).run(0)
    tight = WindowModel(prof, nodes, pods, tile=2, topk=2, window=64).run(0)
    for got in (wide, tight):
        assert got["placed"] == ref.placed and np.array_equal(got["log"], ref.log)
    assert tight["windows"] >= wide["windows"]


@pytest.mark.parametrize("parallel", [False, True])
@pytest.mark.parametrize("window,tile", [(1, 16), (5, 4), (64, 16)])
@pytest.mark.parametrize("seed", range(8))
def test_memo_masks_and_assumed_maxima_vs_oracle(ccref, seed, window, tile, parallel):
    """Round 4's bookkeeping (score memo + refresh, per-domain spread masks, assumed maxima + repair; MemoWindowModel) reproduces the
    oracle's loop, and its invariants -- every memo word a scan reads equals its recomputation, the masks equal the table-derived
    verdicts -- hold at every use (asserted inside the model)."""
    from window_model import MemoWindowModel
    rng = np.random.default_rng(700 + seed)
    nodes, pods, prof = random_multi_case(rng, int(rng.integers(20, 160)), int(rng.integers(2, 20)))
    limit = int(rng.choice([0, 0, 41]))
    ref = ccref.run_multi(prof, nodes, pods, max_limit=limit)
    got = MemoWindowModel(prof, nodes, pods, tile=tile, topk=8, window=window).run(limit, parallel=parallel)
    assert got["placed"] == ref.placed and got["stop"] == ref.stop and got["stop_spec"] == ref.stop_spec
    assert np.array_equal(got["log"], ref.log)
    st = got["stats"]
    if ref.placed > 3 * len(pods):
        assert st["memo_scans"] > 0 and st["words_checked"] > 0  # (the rows were read, not only filled)


@pytest.mark.parametrize("parallel", [False, True])
@pytest.mark.parametrize("window,tile", [(1, 16), (5, 4), (64, 16)])
@pytest.mark.parametrize("seed", range(12))
def test_flag_words_report_maxima_inexactly_and_still_reproduce_the_oracle(ccref, seed, window, tile, parallel):
    """Round 5's 16-bit memo word (csrc/ccsim_multi.h): a scan that reads its row knows only whether a feasible node holds / exceeds the
    assumed normalization maxima.  A differing maximum is reported as assumed +- 1; the self-correction (window ends, general scan, exact
    value) must leave the placement sequence the oracle's, with at most two zero-progress windows in a row (asserted inside the model)."""
    from window_model import MemoWindowModel
    rng = np.random.default_rng(7700 + seed)
    nodes, pods, prof = random_multi_case(rng, int(rng.integers(20, 160)), int(rng.integers(2, 20)))
    limit = int(rng.choice([0, 0, 41]))
    ref = ccref.run_multi(prof, nodes, pods, max_limit=limit)
    got = MemoWindowModel(prof, nodes, pods, tile=tile, topk=8, window=window, flags=True).run(limit, parallel=parallel)
    assert got["placed"] == ref.placed and got["stop"] == ref.stop and got["stop_spec"] == ref.stop_spec
    assert np.array_equal(got["log"], ref.log)


def test_flag_words_do_report_inexactly_somewhere(ccref):
    """(the cases above exercise the inexact report: summed over a few of them it happens)"""
    from window_model import MemoWindowModel
    inexact = 0
    for seed in range(12):
        rng = np.random.default_rng(7700 + seed)
        nodes, pods, prof = random_multi_case(rng, int(rng.integers(20, 160)), int(rng.integers(2, 20)))
        inexact += MemoWindowModel(prof, nodes, pods, window=5, tile=4, flags=True).run(0)["stats"].get("inexact", 0)
    assert inexact > 0


def test_memo_without_refresh_is_caught(ccref):
    """The model's invariant has teeth: without the refresh of the touched nodes a scan meets a stale word."""
    from window_model import MemoWindowModel
    rng = np.random.default_rng(4243)
    nodes, pods, prof = random_multi_case(rng, 80, 6)
    with pytest.raises(AssertionError, match="stale memo word"):
        MemoWindowModel(prof, nodes, pods, window=6, refresh=False).run(0)
