"""Host-side mirror of the reference's committee shuffling (phase0/helpers.rs:249-360), backed by the HIP kernels."""
from __future__ import annotations

import ctypes
from typing import Sequence

from . import _lib

SHUFFLE_ROUND_COUNT_MAINNET, SHUFFLE_ROUND_COUNT_MINIMAL = 90, 10  # phase0/presets/{mainnet,minimal}.rs


def compute_shuffled_indices(indices: Sequence[int], seed: bytes, shuffle_round_count: int = SHUFFLE_ROUND_COUNT_MAINNET) -> list:
    """phase0/helpers.rs:287-360: the whole-list swap-or-not shuffle."""
    if len(seed) != 32:
        raise ValueError("seed is a Bytes32")
    L = _lib.load()
    n = len(indices)
    inp = (ctypes.c_uint64 * max(n, 1))(*indices)
    out = (ctypes.c_uint64 * max(n, 1))()
    _lib.check(L.ecgpu_compute_shuffled_indices(inp, n, ctypes.create_string_buffer(seed, 32), shuffle_round_count, out),
               "ecgpu_compute_shuffled_indices")
    return list(out[:n])
