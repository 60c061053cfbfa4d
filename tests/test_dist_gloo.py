"""N > 1 control flow of the sharded paths on CPU: world size 2, gloo backend (the GPU path uses the
same code with backend nccl = RCCL).  Each rank fabricates the statuses / root its shard would
produce; the test checks the shard arithmetic and that every rank ends with the full, ordered result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ethereum_consensus_amd import shard


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 65536, 1000003):
        for w in (1, 2, 3, 8):
            got = [shard.shard_range(n, r, w) for r in range(w)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[i][1] == got[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1


def _status_of(i: int) -> int:
    return 5 if i % 64 == 0 else 0


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard.shard_range(n_total, rank, world)
        local = torch.tensor([_status_of(i) for i in range(lo, hi)], dtype=torch.uint8)
        full = shard.all_gather_ragged(dist, local, n_total, world)
        ok = full.tolist() == [_status_of(i) for i in range(n_total)]
        root = torch.full((32,), rank + 1, dtype=torch.uint8)
        roots = shard.all_gather_bytes(dist, root, world)
        ok = ok and roots.tolist() == [r + 1 for r in range(world) for _ in range(32)]
        dist.barrier()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4096, 1001])
def test_world2_gloo_all_gather(n_total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


# ---- one big list sharded over the ranks: aligned subtrees, all-gather of sub-roots, redundant top ------
def test_subtree_width_and_ranges():
    for n in (0, 1, 5, 64, 1000, 1 << 20, (1 << 20) + 1):
        for w in (1, 2, 4, 8):
            width = shard.subtree_width(n, w)
            assert width & (width - 1) == 0 and width * w >= n
            if width > 1:
                assert (width // 2) * w < n
            got = [shard.subtree_range(n, r, w) for r in range(w)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[i][1] == got[i + 1][0] for i in range(w - 1))
            assert all(lo % width == 0 or lo == n for lo, _ in got)


def _oracle_merkleize(data: bytes, limit_chunks: int = 0, mix_in_length=None) -> bytes:
    from oracle import ssz as O
    root = O.merkleize_bytes(data, limit_chunks or None)
    return O.mix_in_length(root, mix_in_length) if mix_in_length is not None else root


def _oracle_top(sub_roots: bytes, width: int, limit: int, mix_in_length=None) -> bytes:
    from oracle import ssz as O
    root = O.merkleize_subtree_roots([sub_roots[i:i + 32] for i in range(0, len(sub_roots), 32)], width, limit)
    return O.mix_in_length(root, mix_in_length) if mix_in_length is not None else root


def _validator_roots(lo: int, hi: int) -> bytes:
    import hashlib
    return b"".join(hashlib.sha256(b"validator-root" + i.to_bytes(8, "little")).digest() for i in range(lo, hi))


def _list_worker(rank, world, port, n_total, limit, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard.subtree_range(n_total, rank, world)
        mine = _validator_roots(lo, hi)  # this rank's leaves (element roots), the others never leave their rank
        root = shard.sharded_list_root(dist, n_total, limit, lambda w: _oracle_merkleize(mine, w), _oracle_top,
                                       mix_in_length=n_total)
        dist.barrier()
        q.put((rank, root))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,limit", [(1000, 1 << 40), (1024, 1 << 12), (1, 1 << 10), (0, 1 << 10), (1025, 1 << 40)])
def test_world2_gloo_sharded_list_root_equals_the_unsharded_root(n_total, limit):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_list_worker, args=(r, 2, port, n_total, limit, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    want = _oracle_merkleize(_validator_roots(0, n_total), limit, n_total)
    assert res[0] == want and res[1] == want
