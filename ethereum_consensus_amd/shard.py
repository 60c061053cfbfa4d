"""Multi-GPU sharding of the two hot paths: independent index ranges per rank, and the
path's only collective -- an all-gather of per-shard verify statuses / Merkle roots.  One process per
GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced [lo, hi) of rank `rank`; the first n_total % world ranks get one extra."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_bytes(dist, local, world: int):
    """All-gather equally sized uint8 tensors; returns the concatenation in rank order."""
    import torch
    if world == 1:
        return local
    out = torch.empty(local.numel() * world, dtype=torch.uint8, device=local.device)
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        return torch.cat(parts)
    dist.all_gather_into_tensor(out, local)
    return out


def all_gather_ragged(dist, local, n_total: int, world: int):
    """All-gather shards produced by shard_range (sizes differ by at most one): pad, gather, trim."""
    import torch
    if world == 1:
        return local
    width = (n_total + world - 1) // world
    padded = torch.zeros(width, dtype=torch.uint8, device=local.device)
    padded[:local.numel()] = local
    g = all_gather_bytes(dist, padded, world)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(g[r * width:r * width + (hi - lo)])
    return torch.cat(parts)
