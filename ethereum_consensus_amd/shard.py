"""Multi-GPU sharding of the two hot paths: independent index ranges per rank, and the
path's only collective -- an all-gather of per-shard verify statuses / Merkle roots.  One process per
GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
from __future__ import annotations


def weighted_bounds(n_total: int, weights, granule: int = 1) -> list[int]:
    """Cut points b[0] = 0 <= b[1] <= ... <= b[world] = n_total of contiguous shards whose sizes are proportional to `weights`
    (a rank's measured speed: units per second), each cut rounded to a multiple of `granule`.  Deterministic in its
    arguments: every rank computes the same cuts from the same all-gathered weights.  One slow GPU in a node (a box whose
    instruction fetch is slow runs the BLS batch 1.8 x slower) otherwise sets the step time for all."""
    world = len(weights)
    w = [max(float(x), 0.0) for x in weights]
    tot = sum(w)
    if tot <= 0:
        w, tot = [1.0] * world, float(world)
    g = max(1, int(granule))
    b, acc = [0], 0.0
    for r in range(world - 1):
        acc += w[r]
        cut = int(round(n_total * acc / tot / g)) * g
        b.append(min(max(cut, b[-1]), n_total))
    b.append(n_total)
    return b


def shard_range(n_total: int, rank: int, world: int, weights=None, granule: int = 1) -> tuple[int, int]:
    """Contiguous [lo, hi) of rank `rank`.  Without weights: balanced, the first n_total % world ranks get one extra.  With
    per-rank weights (len == world): proportional to them (weighted_bounds)."""
    if weights is not None:
        if len(weights) != world:
            raise ValueError("one weight per rank")
        b = weighted_bounds(n_total, weights, granule)
        return b[rank], b[rank + 1]
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_bytes(dist, local, world: int, force: bool = False):
    """All-gather equally sized uint8 tensors; returns the concatenation in rank order.  force: go through the collective even
    with one rank (a single-rank process group is legal: the GPU tests exercise the RCCL call that way on a one-GPU box)."""
    import torch
    if world == 1 and not force:
        return local
    out = torch.empty(local.numel() * world, dtype=torch.uint8, device=local.device)
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        return torch.cat(parts)
    dist.all_gather_into_tensor(out, local)
    return out


def all_gather_ragged(dist, local, n_total: int, world: int, force: bool = False, weights=None, granule: int = 1):
    """All-gather shards produced by shard_range (balanced: sizes differ by at most one; weighted: by whatever the weights
    say): pad to the widest shard, gather, trim."""
    import torch
    if world == 1 and not force:
        return local
    sizes = [hi - lo for lo, hi in (shard_range(n_total, r, world, weights, granule) for r in range(world))]
    width = max(max(sizes), 1)
    padded = torch.zeros(width, dtype=torch.uint8, device=local.device)
    padded[:local.numel()] = local
    g = all_gather_bytes(dist, padded, world, force)
    return torch.cat([g[r * width:r * width + sizes[r]] for r in range(world)])


def gather_speeds(dist, mine: float, world: int, device="cpu"):
    """every rank's measured speed (units per second), in rank order, on every rank"""
    import torch
    if world == 1 or dist is None:
        return [float(mine)]
    t = torch.tensor([float(mine)], dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def balanced_enough(speeds, tolerance: float = 1.15) -> bool:
    """True when the fastest rank is within `tolerance` of the slowest: an even split is kept (re-sharding costs a second
    workload generation and, on a homogeneous node, would only follow timer noise)"""
    pos = [s for s in speeds if s > 0]
    return not pos or max(pos) <= tolerance * min(pos)


def subtree_width(n_total: int, world: int) -> int:
    """Leaves per rank when one big list is sharded: the smallest power of two W with W * world >=
    n_total, so that every rank owns one aligned subtree (ranks past the end own an all-zero subtree)."""
    per = max(1, -(-n_total // world))
    return 1 << (per - 1).bit_length()


def subtree_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """[lo, hi) of the leaves rank `rank` owns under subtree_width."""
    w = subtree_width(n_total, world)
    return min(rank * w, n_total), min((rank + 1) * w, n_total)


def sharded_list_root(dist, n_total: int, limit: int, sub_root, top, mix_in_length=None) -> bytes:
    """Root of one list whose leaves are sharded by subtree_range.  `sub_root(width) -> 32 bytes` is this rank's aligned
    subtree (no length mix-in); `top(sub_roots, width, limit, mix_in_length)` climbs the top log2(world) levels and the
    zero-hash ladder to the list limit and mixes the length in -- redundantly on every rank, after the path's only
    exchange: an all-gather of the 32-byte sub-roots."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    w = subtree_width(n_total, world)
    if limit % w:
        raise ValueError("list limit is not a multiple of the shard width")
    mine = sub_root(w)
    if world > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.frombuffer(bytearray(mine), dtype=torch.uint8).to(dev)
        roots = bytes(all_gather_bytes(dist, t, world).cpu().numpy())
    else:
        roots = mine
    return top(roots, w, limit, mix_in_length)
