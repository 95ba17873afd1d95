"""A seeded, bounded slice of the randomised differential soak (tests/fuzz_cases.py) inside the
driver-run GPU suite: every algorithm / min_match / cap / exclusion choice, upserts between index
build and match, top-k + merge, long single queries and the scene path against the oracle."""
import pytest

from tests import fuzz_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [12345, 777])
def test_fuzz_slice_against_oracle(seed):
    stats = fuzz_cases.run(seconds=25.0, seed=seed, max_cases=60)
    assert stats["match_cases"] >= 5 and stats["scene_cases"] >= 5, stats
