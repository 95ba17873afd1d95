"""A seeded, bounded slice of the randomised differential soak (tests/fuzz_cases.py) inside the
driver-run GPU suite: every algorithm / min_match / cap / exclusion choice, upserts between index
build and match, top-k + merge, long single queries and the scene path against the oracle - and, since
round 4, what round 3 built: corpora of 2-3 sub-indexes, background rebuilds + swaps under matches in
flight (a second thread upserts while this one matches), batches that hold queries of more than 4095
timestamps.  The counters of both slices together must show that those cases ran."""
import pytest

from tests import fuzz_cases

pytestmark = pytest.mark.gpu
_totals = {}


@pytest.mark.parametrize("seed", [12345, 777])
def test_fuzz_slice_against_oracle(seed):
    stats = fuzz_cases.run(seconds=28.0, seed=seed, max_cases=60)
    assert stats["match_cases"] >= 5 and stats["scene_cases"] >= 5, stats
    for k, v in stats.items():
        _totals[k] = _totals.get(k, 0) + v
    print(stats)


def test_fuzz_slices_reached_the_round3_mechanisms():
    if not _totals:
        pytest.skip("the slices did not run")
    assert _totals["multi_sub_cases"] >= 1 and _totals["rebuilds_during_cases"] >= 1, _totals
    assert _totals["concurrent_match_calls"] >= 2 and _totals["long_batch_cases"] >= 1, _totals
