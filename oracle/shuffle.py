"""CPU restatement of the reference's swap-or-not shuffling -- TEST INFRASTRUCTURE, not product.

compute_shuffled_index follows /root/reference/ethereum-consensus/src/phase0/helpers.rs:249-282 line by line;
compute_shuffled_indices follows the whole-list algorithm of :287-360 (rounds in reverse, mirrored swaps around the
pivot).  tests/test_oracle_shuffle.py checks that the two agree (out[i] = in[shuffled_index(i)]), which is the
property the GPU kernels rely on; there is no fixed vector for shuffling in the reference tree (parity pinned on
the agreement of two independently restated reference algorithms + the consensus-spec definition they implement).
`shuffled_indices_numpy` is the per-index form vectorised over all indices, for sizes the loops cannot reach."""
import hashlib


def _hash(b: bytes) -> bytes:
    return hashlib.sha256(b).digest()


def compute_shuffled_index(index: int, index_count: int, seed: bytes, rounds: int = 90) -> int:
    assert index < index_count
    for r in range(rounds):
        pivot = int.from_bytes(_hash(seed + bytes([r]))[:8], "little") % index_count
        flip = (pivot + index_count - index) % index_count
        position = max(index, flip)
        source = _hash(seed + bytes([r]) + (position // 256).to_bytes(4, "little"))
        byte = source[(position % 256) // 8]
        if (byte >> (position % 8)) % 2:
            index = flip
    return index


def compute_shuffled_indices(indices, seed: bytes, rounds: int = 90):
    inp = list(indices)
    n = len(inp)
    if n == 0:
        return inp
    for r in range(rounds - 1, -1, -1):
        pivot = int.from_bytes(_hash(seed + bytes([r]))[:8], "little") % n
        src_in = seed + bytes([r])
        source = _hash(src_in + (pivot >> 8).to_bytes(4, "little"))
        byte_source = source[(pivot & 0xFF) >> 3]
        mirror = (pivot + 1) >> 1
        for i in range(mirror):
            j = pivot - i
            if j & 0xFF == 0xFF:
                source = _hash(src_in + (j >> 8).to_bytes(4, "little"))
            if j & 0x07 == 0x07:
                byte_source = source[(j & 0xFF) >> 3]
            if (byte_source >> (j & 0x07)) & 1:
                inp[i], inp[j] = inp[j], inp[i]
        end = n - 1
        source = _hash(src_in + (end >> 8).to_bytes(4, "little"))
        byte_source = source[(end & 0xFF) >> 3]
        mirror = (pivot + n + 1) >> 1
        for k, i in enumerate(range(pivot + 1, mirror)):
            j = end - k
            if j & 0xFF == 0xFF:
                source = _hash(src_in + (j >> 8).to_bytes(4, "little"))
            if j & 0x07 == 0x07:
                byte_source = source[(j & 0xFF) >> 3]
            if (byte_source >> (j & 0x07)) & 1:
                inp[i], inp[j] = inp[j], inp[i]
    return inp


def shuffled_indices_numpy(n: int, seed: bytes, rounds: int = 90):
    """permutation p with p[i] = compute_shuffled_index(i, n, seed), all i at once"""
    import numpy as np
    idx = np.arange(n, dtype=np.uint64)
    nb = (n + 255) // 256
    for r in range(rounds):
        pivot = int.from_bytes(_hash(seed + bytes([r]))[:8], "little") % n
        table = np.frombuffer(b"".join(_hash(seed + bytes([r]) + p.to_bytes(4, "little")) for p in range(nb)), dtype=np.uint8)
        flip = (np.uint64(pivot + n) - idx) % np.uint64(n)
        pos = np.maximum(idx, flip)
        byte = table[(pos // np.uint64(256)) * np.uint64(32) + (pos % np.uint64(256)) // np.uint64(8)]
        bit = (byte >> (pos % np.uint64(8)).astype(np.uint8)) & 1
        idx = np.where(bit == 1, flip, idx)
    return idx
