"""Differential fuzzing of every engine, forced scan plan and positions path against the oracle's brute
force (tests/fuzz_gpu.py): pattern sets with long shared prefixes / suffixes, duplicates and text-cut
patterns over random alphabets, lengths and sizes.  `python tests/fuzz_gpu.py 80 <seed>` runs more."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_random_sets_agree_with_the_oracle(seed):
    import fuzz_gpu
    assert fuzz_gpu.run(25, seed, verbose=False) > 200
