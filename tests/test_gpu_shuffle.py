"""GPU parity for the swap-or-not shuffling (ecgpu_compute_shuffled_indices) against oracle/shuffle.py."""
import random
import time

import pytest

from oracle import shuffle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from ethereum_consensus_amd import _lib, shuffling
    L = _lib.load(build_if_missing=False)
    assert L.ecgpu_init(-1) == 0, L.ecgpu_last_error()
    return shuffling


@pytest.mark.parametrize("n", [0, 1, 2, 3, 100, 255, 256, 257, 1000])
def test_small_lists_vs_the_reference_list_algorithm(gpu, n):
    r = random.Random(n)
    for rounds in (10, 90):
        seed = r.randbytes(32)
        inp = [r.randrange(1 << 63) for _ in range(n)]
        assert gpu.compute_shuffled_indices(inp, seed, rounds) == shuffle.compute_shuffled_indices(inp, seed, rounds)


def test_full_registry(gpu):
    """2^20 validator indices, 90 rounds: the per-index definition vectorised with numpy"""
    n = 1 << 20
    seed = bytes(range(32))
    t0 = time.time()
    got = gpu.compute_shuffled_indices(range(n), seed, 90)
    dt = time.time() - t0
    want = shuffle.shuffled_indices_numpy(n, seed, 90)
    assert got == [int(x) for x in want]
    assert sorted(got[:1000] + got[-1000:]) != list(range(2000))  # it did shuffle
    print(f"2^20 indices shuffled through the host entry in {dt * 1e3:.1f} ms (ctypes marshalling included)")
