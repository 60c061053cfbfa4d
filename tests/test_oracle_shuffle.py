"""oracle/shuffle.py: the reference's two shuffling algorithms restated independently must agree with each other."""
import random

import pytest

from oracle import shuffle


@pytest.mark.parametrize("n", [1, 2, 3, 7, 100, 255, 256, 257, 513, 1000])
@pytest.mark.parametrize("rounds", [10, 90])
def test_list_form_equals_per_index_form(n, rounds):
    r = random.Random(n * 1000 + rounds)
    seed = r.randbytes(32)
    inp = [r.randrange(1 << 40) for _ in range(n)]
    out = shuffle.compute_shuffled_indices(inp, seed, rounds)
    assert sorted(out) == sorted(inp)
    assert out == [inp[shuffle.compute_shuffled_index(i, n, seed, rounds)] for i in range(n)]
    assert list(shuffle.shuffled_indices_numpy(n, seed, rounds)) == [shuffle.compute_shuffled_index(i, n, seed, rounds) for i in range(n)]


def test_empty_list():
    assert shuffle.compute_shuffled_indices([], bytes(32)) == []
