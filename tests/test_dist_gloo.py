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


# ---- bench.py's N-rank control flow (finish, self-check gather, epoch sharding, strong-scaled K = 1 batch) -----------------
# bench.py is device-agnostic above its `DEV` switch: here it runs on CPU tensors under gloo with a stub library whose entry
# points have the C ABI's argument order and are answered by the C++ oracle (oracle/cbls.py).  What stays untested without
# N GPUs is the RCCL transport itself.
class _StubLib:
    """the subset of libecgpu.so's entries bench.run_epoch / bench.run_bls call, over host memory (data_ptr of CPU tensors)"""

    def __init__(self):
        self.regs = {}

    @staticmethod
    def _rd(ptr, n):
        import ctypes
        return ctypes.string_at(ptr, n) if n else b""

    @staticmethod
    def _wr(ptr, data):
        import ctypes
        ctypes.memmove(ptr, bytes(data), len(data))

    def ecgpu_last_error(self):
        return b""

    def ecgpu_sk_to_pk_batch_dev(self, d_sk, n, d_pk, stream):
        from oracle import cbls
        sk = self._rd(d_sk, 32 * n)
        self._wr(d_pk, b"".join(cbls.sk_to_pk(int.from_bytes(sk[32 * i:32 * i + 32], "big")) for i in range(n)))
        return 0

    def ecgpu_sign_batch_dev(self, d_sk, stride, d_msg, n, d_sig, stream):
        from oracle import cbls
        sk, msg = self._rd(d_sk, 32 * n if stride else 32), self._rd(d_msg, 32 * n)
        self._wr(d_sig, b"".join(cbls.sign(int.from_bytes(sk[stride * i:stride * i + 32], "big"), msg[32 * i:32 * i + 32]) for i in range(n)))
        return 0

    def ecgpu_registry_create(self, n, ref):
        h = len(self.regs) + 1
        self.regs[h] = bytearray(48 * n)
        ref._obj.value = h
        return 0

    def ecgpu_registry_set_dev(self, reg, first, d_keys, n, stream):
        self.regs[reg.value][48 * first:48 * (first + n)] = self._rd(d_keys, 48 * n)
        return 0

    def ecgpu_registry_destroy(self, reg):
        self.regs.pop(reg.value, None)

    def _fav(self, key_of, d_off, n_keys, d_msg, d_sig, n, eth, d_st):
        import struct
        from oracle import cbls
        off = list(struct.unpack(f"<{n + 1}I", self._rd(d_off, 4 * (n + 1)))) if d_off else list(range(n + 1))
        msg, sig = self._rd(d_msg, 32 * n), self._rd(d_sig, 96 * n)
        st = bytes(cbls.fast_aggregate_verify([key_of(j) for j in range(off[i], off[i + 1])], msg[32 * i:32 * i + 32], sig[96 * i:96 * i + 96],
                                              bool(eth)) & 0xFF for i in range(n))
        self._wr(d_st, st)
        return 0

    def ecgpu_fast_aggregate_verify_batch_dev(self, d_pk, d_off, n_pks, d_msg, d_sig, n, eth, d_st, stream):
        pk = self._rd(d_pk, 48 * n_pks)
        return self._fav(lambda j: pk[48 * j:48 * j + 48], d_off, n_pks, d_msg, d_sig, n, eth, d_st)

    def ecgpu_fast_aggregate_verify_indexed_batch_dev(self, reg, d_idx, d_off, n_keys, d_msg, d_sig, n, eth, d_st, stream):
        import struct
        keys = self.regs[reg.value]
        idx = struct.unpack(f"<{n_keys}I", self._rd(d_idx, 4 * n_keys))
        return self._fav(lambda j: bytes(keys[48 * idx[j]:48 * idx[j] + 48]), d_off, n_keys, d_msg, d_sig, n, eth, d_st)

    def ecgpu_prof_filter(self, tag):
        return 0

    def ecgpu_prof_enable(self, on):
        return 0

    def ecgpu_prof_read(self, tag, ms, nl):
        ms._obj.value, nl._obj.value = 1.0, 1
        return 1

    def ecgpu_bls_tower(self):
        return 1

    def ecgpu_bls_last_pairing_path(self):
        return 1


def _bench_worker(rank, world, port, q):
    import argparse
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        bench.DEV = "cpu"
        args = argparse.Namespace(steps=1, warmup=1, tuples=10, scaling="strong", even_shards=True)
        res = {}
        # finish(): the slowest rank's wall time sets the step time; value = all ranks' units over it
        line = bench.finish(dict(dt=0.5 * (rank + 1), units_per_step=100, metric="m", unit="u", dtype="u32", config={}, roofline={},
                                 check={"r": rank}), args, world, dist, torch)
        res["finish"] = (line["n_gpus"], round(line["ms_per_step"], 6), round(line["value"], 6), line["scaling"])
        # the per-rank self-check gather: every rank's pair in rank order
        res["selfcheck"] = bench.gather_selfcheck([1.0 + rank, float(1 + rank % 2)], world, dist, torch)
        # configs[3] in miniature: 7 aggregates of 3 keys over `world` ranks (ragged), both key paths, statuses all-gathered
        L = _StubLib()
        e = bench.run_epoch(args, L, torch, dist, rank, world, n_total=7, k=3, n_reg=32, sk_period=8)
        res["epoch"] = (e["check"]["statuses_match_construction"], e["scaling"], e["config"]["aggregates_per_gpu"], e["units_per_step"] * world)
        # north_star's strong-scaled K = 1 batch in miniature: 10 tuples in all, rank g verifies shard_range(10, g, world)
        b = bench.run_bls(args, L, torch, dist, rank, world)
        fb = bench.finish(b, args, world, dist, torch)
        res["bls"] = (b["check"]["statuses_match_construction"], b["config"]["tuples_this_rank"], fb["scaling"], round(fb["value"] * fb["ms_per_step"] / 1e3, 6))
        dist.barrier()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_n_rank_control_flow_under_gloo(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        out = res[r]
        # max over ranks of dt = 0.5 * world seconds for ONE step; value = 100 units x world ranks over that time
        assert out["finish"] == (world, round(500.0 * world, 6), round(100 * world / (0.5 * world), 6), "weak")
        assert out["selfcheck"] == [[1.0 + k, float(1 + k % 2)] for k in range(world)]
        per = -(-7 // world)
        assert out["epoch"] == (True, "strong", per, 7 * 3)
        lo, hi = shard.shard_range(10, r, world)
        assert out["bls"][:3] == (True, hi - lo, "strong")
        assert abs(out["bls"][3] - 10) < 1e-3  # value x time of one step = the whole batch, whatever the rank count


# ---- speed-weighted shards of the strong-scaled BLS batch ---------------------------
class _SlowRankStub(_StubLib):
    """the stub library with a verify call that takes `ms_per_tuple` of wall time per tuple on top of the oracle's: one rank of
    the group is made 1.8 x slower, the way a GPU with slow instruction fetch is"""

    def __init__(self, ms_per_tuple):
        super().__init__()
        self.ms_per_tuple = ms_per_tuple

    def ecgpu_fast_aggregate_verify_batch_dev(self, d_pk, d_off, n_pks, d_msg, d_sig, n, eth, d_st, stream):
        import time
        time.sleep(n * self.ms_per_tuple * 1e-3)
        return super().ecgpu_fast_aggregate_verify_batch_dev(d_pk, d_off, n_pks, d_msg, d_sig, n, eth, d_st, stream)


def _weighted_worker(rank, world, port, q, total, slow_rank):
    import argparse
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        bench.DEV = "cpu"
        bench.BLS_SHARD_GRANULE = 4
        L = _SlowRankStub(40.0 * (1.8 if rank == slow_rank else 1.0))
        args = argparse.Namespace(steps=2, warmup=1, tuples=total, scaling="strong", even_shards=False)
        b = bench.run_bls(args, L, torch, dist, rank, world)
        line = bench.finish(b, args, world, dist, torch)
        even = argparse.Namespace(steps=2, warmup=1, tuples=total, scaling="strong", even_shards=True)
        e = bench.finish(bench.run_bls(even, L, torch, dist, rank, world), even, world, dist, torch)
        dist.barrier()
        q.put((rank, dict(ok=b["check"]["statuses_match_construction"], n=b["config"]["tuples_this_rank"], weights=line["weights"],
                          ms=line["ms_per_step"], ms_even=e["ms_per_step"], n_even=e["config"]["tuples_this_rank"] if "config" in e else None)))
    finally:
        dist.destroy_process_group()


def test_speed_weighted_shards_with_one_slow_rank_under_gloo():
    """world 4, rank 2 verifies 1.8 x slower: the strong-scaled batch is cut in proportion to the measured speeds (all-gathered:
    every rank computes the same cuts), the ragged gather still puts every shard's statuses where they belong, and a step takes
    within 10 % of the weighted optimum -- where the even split waits for the slow rank."""
    world, total, slow = 4, 96, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_weighted_worker, args=(r, world, port, q, total, slow)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    w = res[0]["weights"]
    assert all(res[r]["ok"] for r in range(world))          # own shard AND the gathered whole equal construction on every rank
    assert all(res[r]["weights"] == w for r in range(world))  # the same cuts everywhere
    assert w["weights"] is not None and min(w["weights"]) == w["weights"][slow]
    sizes = [res[r]["n"] for r in range(world)]
    assert sum(sizes) == total and sizes[slow] == min(sizes) and sizes[slow] < total // world
    bounds = shard.weighted_bounds(total, w["speeds_tuples_per_s"], 4)
    assert sizes == [bounds[r + 1] - bounds[r] for r in range(world)]
    optimum_ms = total / sum(w["speeds_tuples_per_s"]) * 1e3
    assert res[0]["ms"] <= 1.10 * optimum_ms + 15.0, (res[0]["ms"], optimum_ms)  # (+ the gather's few milliseconds under gloo)
    assert res[0]["ms_even"] >= 1.25 * res[0]["ms"]           # the even split waits for the slow rank


def test_weighted_bounds_are_monotonic_cover_the_range_and_follow_the_weights():
    for total, weights, g in ((1 << 20, [1] * 8, 1024), (1 << 20, [1, 1, 1, 0.55, 1, 1, 1, 1], 1024), (10, [3, 1], 1), (7, [0, 0, 0], 1),
                              (1000, [1, 2, 3, 4], 16), (5, [1, 1, 1, 1, 1, 1, 1, 1], 1), (131072, [1, 1e-9], 65536)):
        b = shard.weighted_bounds(total, weights, g)
        assert b[0] == 0 and b[-1] == total and all(x <= y for x, y in zip(b, b[1:])) and len(b) == len(weights) + 1
        assert all(x % g == 0 for x in b[1:-1])
        if sum(weights) > 0 and total >= 100 * g:
            for r, wt in enumerate(weights):
                assert abs((b[r + 1] - b[r]) - total * wt / sum(weights)) <= g
        for r in range(len(weights)):
            assert shard.shard_range(total, r, len(weights), weights, g) == (b[r], b[r + 1])
    assert shard.balanced_enough([1.0, 1.1, 1.05]) and not shard.balanced_enough([1.0, 1.8, 1.0])


# ---- ONE BeaconState over N ranks (bench.py --workload merkle --scaling strong) --------------------------
# The same stub technique: the three state entries of the C ABI answered by oracle/ssz.py over host memory, so that
# bench.run_merkle_sharded's own N-rank flow -- phase A per rank, the all-gather of 5 x 32 bytes, phase B on every rank, the
# root compared with the unsharded root -- runs under gloo at world sizes 2, 3 and 4.
class _StateStub:
    LISTS = (("validators", 121, 1), ("balances", 8, 4), ("previous_epoch_participation", 1, 32),
             ("current_epoch_participation", 1, 32), ("inactivity_scores", 8, 4))

    def __init__(self, n_validators):
        from ethereum_consensus_amd import synthetic as S
        from oracle import ssz as O
        self.f = S.state_fields(n_validators, "mainnet", seed=1)
        self.enc = S.serialize_state(self.f)
        self.n = n_validators
        self.O = O
        self.last = 0
        from tests._statevalue import oracle_state_value
        t = O.BeaconStateDeneb(O.MAINNET)
        self.names = [k for k, _ in t.fields]
        light = dict(self.f)
        light["validators"] = self.f["validators"][:0]
        for k in ("balances", "previous_epoch_participation", "current_epoch_participation", "inactivity_scores"):
            light[k] = self.f[k][:0]
        self.small_roots = t.field_roots(oracle_state_value(light))
        vt = dict(t.fields)["validators"].elem
        full = oracle_state_value(self.f)
        self.vroots = [vt.htr(v) for v in full["validators"]]

    def _leaves(self, name):
        """(chunks or element roots, limit in leaves) of one of the five lists"""
        O = self.O
        if name == "validators":
            return self.vroots, O.MAINNET.VALIDATOR_REGISTRY_LIMIT
        data = self.f[name].tobytes()
        per = 32 // (8 if name in ("balances", "inactivity_scores") else 1)
        chunks = [data[i:i + 32].ljust(32, b"\0") for i in range(0, len(data), 32)]
        return chunks, O.MAINNET.VALIDATOR_REGISTRY_LIMIT // per

    def ecgpu_last_error(self):
        return b""

    def ecgpu_beacon_state_fixed_size(self, fork, preset):
        return 2736629 + 4  # unused by the stub beyond slicing the host copy; any value <= len(enc)

    def ecgpu_beacon_state_shard_lists(self):
        return 5

    def ecgpu_last_hash64_count(self):
        return self.last

    def _check_state(self, d_ssz, n_bytes):
        import ctypes
        assert n_bytes == len(self.enc) and ctypes.string_at(d_ssz, 64) == self.enc[:64]

    def ecgpu_htr_beacon_state_dev(self, fork, d_ssz, n_bytes, h_fixed, preset, d_root, stream):
        import ctypes
        self._check_state(d_ssz, n_bytes)
        O = self.O
        roots = list(self.small_roots)
        for name, _, _ in self.LISTS:
            leaves, limit = self._leaves(name)
            roots[self.names.index(name)] = O.mix_in_length(O.merkleize_chunks(leaves, limit), self.n)
        ctypes.memmove(d_root, O.merkleize_chunks(roots, len(roots)), 32)
        self.last = 1000
        return 0

    def ecgpu_beacon_state_shard_subroots_dev(self, fork, d_ssz, n_bytes, h_fixed, preset, rank, world, d_sub, d_keep, stream):
        import ctypes
        self._check_state(d_ssz, n_bytes)
        out = b""
        for name, _, _ in self.LISTS:
            leaves, _ = self._leaves(name)
            w = shard.subtree_width(len(leaves), world)
            lo, hi = shard.subtree_range(len(leaves), rank, world)
            out += self.O.merkleize_chunks(leaves[lo:hi], w)
        ctypes.memmove(d_sub, out, 160)
        self.last = 100
        return 0

    def ecgpu_htr_beacon_state_sharded_dev(self, fork, d_ssz, n_bytes, h_fixed, preset, d_all, world, d_keep, d_root, stream):
        import ctypes
        self._check_state(d_ssz, n_bytes)
        O = self.O
        allb = ctypes.string_at(d_all, 160 * world)
        roots = list(self.small_roots)
        for k, (name, _, _) in enumerate(self.LISTS):
            leaves, limit = self._leaves(name)
            w = shard.subtree_width(len(leaves), world)
            n_sub = -(-len(leaves) // w)
            subs = [allb[160 * r + 32 * k:160 * r + 32 * k + 32] for r in range(n_sub)]
            roots[self.names.index(name)] = O.mix_in_length(O.merkleize_subtree_roots(subs, w, limit), self.n)
        ctypes.memmove(d_root, O.merkleize_chunks(roots, len(roots)), 32)
        self.last = 50
        return 0


def _merkle_worker(rank, world, port, n_validators, q):
    import argparse
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        bench.DEV = "cpu"
        args = argparse.Namespace(steps=1, warmup=1, validators=n_validators, scaling="strong")
        L = _StateStub(n_validators)
        r = bench.run_merkle_sharded(args, L, torch, dist, rank, world)
        line = bench.finish(r, args, world, dist, torch)
        dist.barrier()
        q.put((rank, (r["check"]["equals_unsharded_root"], r["check"]["root"], line["scaling"], line["n_gpus"],
                      round(line["value"] * line["ms_per_step"] / 1e3))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_validators", [(2, 300), (4, 300), (3, 77), (4, 2)])
def test_one_state_sharded_over_the_ranks_under_gloo(world, n_validators):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_merkle_worker, args=(r, world, port, n_validators, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    roots = {res[r][1] for r in range(world)}
    assert len(roots) == 1  # every rank ends with the same root ...
    for r in range(world):
        ok, _, scaling, n_gpus, units = res[r]
        assert ok is True and scaling == "strong" and n_gpus == world  # ... which is the unsharded root
        assert units == 1000  # value x step time = the whole state's hash64, whatever the rank count
