#!/usr/bin/env python
"""What fails first on a real N-GPU node, as a command of its own (bench.py runs the same check before it times anything):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/multi_gpu_preflight.py

Every rank: torch.cuda.set_device(LOCAL_RANK), an nccl (= RCCL) process group bound to that device, ecgpu_init(LOCAL_RANK),
the library's thread device == torch's current device == LOCAL_RANK, one kernel of the library, a 1-byte
all_gather_into_tensor on device tensors, then the sharded Merkle path and a strong-scaled K = 1 BLS batch in miniature
(1 024 validators / 64 tuples), results asserted.  Rank 0 prints one JSON line.  Works with N = 1."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import bench
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local), rank=rank, world_size=world)
    from ethereum_consensus_amd import _lib
    L = _lib.load(build_if_missing=False)
    rc = L.ecgpu_init(local)
    if rc != 0:
        raise RuntimeError(f"ecgpu_init -> {rc}: {L.ecgpu_last_error()}")
    bench.FORCE_DIST = True
    out = bench.multi_gpu_preflight(L, torch, dist, rank, world, local)
    args = argparse.Namespace(steps=1, warmup=1, tuples=64, scaling="strong", validators=1024)
    m = bench.run_merkle_sharded(args, L, torch, dist, rank, world)
    out["sharded_state_root_equals_unsharded"] = m["check"]["equals_unsharded_root"]
    b = bench.run_bls(args, L, torch, dist, rank, world)
    out["strong_bls_statuses_match"] = b["check"]["statuses_match_construction"]
    ok = torch.tensor([int(out["sharded_state_root_equals_unsharded"] and out["strong_bls_statuses_match"])], device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    out["all_ranks_ok"] = bool(ok.item())
    dist.destroy_process_group()
    if rank == 0:
        bench.flush_c_stdio()
        print(json.dumps(out), flush=True)
    sys.exit(0 if out["all_ranks_ok"] else 1)


if __name__ == "__main__":
    main()
