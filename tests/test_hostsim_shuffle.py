"""The shuffling lane programs (csrc/shuffle.h) on the CPU simulator against oracle/shuffle.py."""
import ctypes
import random

import pytest

from oracle import shuffle
from tests import _hostsim as hs


def sim_shuffle(inp, seed, rounds):
    L = hs.lib()
    n = len(inp)
    a = (ctypes.c_uint64 * max(n, 1))(*inp)
    o = (ctypes.c_uint64 * max(n, 1))()
    L.hs_shuffle.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p]
    L.hs_shuffle(a, n, seed, rounds, o)
    return list(o[:n])


@pytest.mark.parametrize("n", [1, 2, 3, 100, 255, 256, 257, 1000, 5000])
@pytest.mark.parametrize("rounds", [10, 90])
def test_shuffle_lane_programs(n, rounds):
    r = random.Random(n + rounds)
    seed = r.randbytes(32)
    inp = [r.randrange(1 << 63) for _ in range(n)]
    if n <= 1000:
        assert sim_shuffle(inp, seed, rounds) == shuffle.compute_shuffled_indices(inp, seed, rounds)
    else:
        p = shuffle.shuffled_indices_numpy(n, seed, rounds)
        assert sim_shuffle(inp, seed, rounds) == [inp[int(j)] for j in p]
