#!/usr/bin/env python
"""bench.py -- the two hot paths on MI355X, one JSON line (contract: see README / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W [--workload bls|merkle|epoch|slots] [--tuples T --scaling strong]

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
  bls    : fast_aggregate_verify of 65 536 (pk, msg, sig) tuples  (BASELINE.json configs[1])
  merkle : hash_tree_root(BeaconState), deneb mainnet, 2^20 validators (configs[2])
  epoch  : 32 slots x 64 committees x 2 048 keys (configs[3]);  slots: sync aggregate + state root per slot (configs[4])
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank processes its own shard
of the same size (weak scaling); the only collective is the RCCL all-gather of the per-shard
verify status bytes (bls) / of the 32-byte roots (merkle).  `--tuples 1048576 --scaling strong` is north_star's
2^20-signature batch: the total is fixed and rank g verifies shard_range(2^20, g, N).
The default line (N = 1, no flags) carries every configuration as a sub-record: "merkle", "aggregates_k2048", "block",
"strong_2p20", "epoch", "slots", "half_round_32768".
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s, 6.29 measured copy)
# Measured ALU ceilings of the two paths' inner loops on MI355X (tools/fpbench.hip, register-resident chains at 8 waves/SIMD):
HASH64_PEAK_GHS = 17.66    # profiles/r01h_hash64_rate_vs_occupancy.txt: 2410 VALU instructions per hash64 at ~3.7 cycles each
# v_mad_u64_u32 issue rate, chip-wide, from the un-foldable asm streams of tools/issue_rate.hip (profiles/r02p_issue_rates.txt):
# 36.4 T/s at 8 waves per SIMD -- the multiplier's ceiling, what every `valu_int.frac` below is a fraction of -- 31.8 T at two
# waves, 26.7 T for a LONE wave per SIMD, which is all a 512-register lane kernel can have (rounds 1-3 quoted 31.0 T, the round-1
# microbench, as the peak: that flattered every fraction by 17 %).
MUL_PIPE_PEAK_TOPS = 36.4
MUL_PIPE_ONE_WAVE_TOPS = 26.7
MUL_PIPE_TWO_WAVES_TOPS = 31.8


# The N-rank control flow below (finish, gather_selfcheck, run_epoch, run_bls) is device-agnostic on purpose: tests/
# test_dist_gloo.py drives it on CPU tensors under gloo with a stub library (the oracle standing in for the kernels), so the
# only multi-GPU lines left untested without hardware are the RCCL calls themselves.
DEV = "cuda"
# ECGPU_BENCH_FORCE_DIST=1: run the collectives of the N-rank flow even when the process group has ONE rank.  A single-rank
# nccl group is legal, so `pytest -m gpu` executes the RCCL path of run_bls / run_merkle_sharded / finish on the one-GPU box
# every round (tests/test_gpu_dist.py) instead of leaving it to the first 8-GPU run.
FORCE_DIST = os.environ.get("ECGPU_BENCH_FORCE_DIST", "0") == "1"


def _multi(dist, world):
    return world > 1 or (FORCE_DIST and dist is not None)


def _sync(torch):
    if DEV == "cuda":
        torch.cuda.synchronize()


def _stream(torch):
    return torch.cuda.current_stream().cuda_stream if DEV == "cuda" else 0


def _device(torch):
    return torch.device("cuda", torch.cuda.current_device()) if DEV == "cuda" else torch.device("cpu")


def effective_cores():
    """Host parallelism this process is actually granted: the affinity mask capped by the cgroup CPU quota (os.cpu_count()
    is the machine's, not ours: the GPU box reports 256 hardware threads and grants about ten)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return {"affinity": aff, "cgroup_quota_cores": quota, "cores_effective": aff if quota is None else min(aff, max(1, int(quota + 0.5)))}


# translation unit whose objects hold each kernel the roofline is quoted on (lib/build_manifest.json is keyed by it)
KERNEL_UNIT = {"k_pairing": "bls_pairing_kernels.hip", "k_miller2": "bls_pairing2_kernels.hip", "k_finalexp": "bls_pairing_kernels.hip", "k_finalexp2": "bls_finalexp2_kernels.hip",
               "k_vm3_pair_a": "bls_vm3.hip",
               "k_vm3_pair_c": "bls_vm3.hip", "k_merkle_pass<2, ValidatorLeaves>": "merkle.hip"}


def kernel_source_hash(kernel_key: str):
    """identity of the loaded kernel: the hash build.py recorded over the translation unit's sources when it built the object"""
    try:
        lib = os.environ.get("ECGPU_LIB")
        d = os.path.dirname(lib) if lib else os.path.join(ROOT, "ethereum_consensus_amd", "lib")
        name = "build_manifest.json"
        if lib and os.path.basename(lib).startswith("libecgpu_"):
            name = "build_manifest_" + os.path.basename(lib)[len("libecgpu_"):-3] + ".json"
        with open(os.path.join(d, name)) as f:
            return json.load(f).get(KERNEL_UNIT.get(kernel_key, ""))
    except (OSError, ValueError):
        return None


def pmc_traffic(kernel_key: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (separate runs, see
    profiles/pmc_traffic.json and the `pmc` step of tools/gpu_visit.sh): (2 x FETCH_SIZE + WRITE_SIZE) KiB -- the factor 2 is the gfx950
    FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section).  None when no pass has been recorded for this kernel, AND when the
    recorded pass was taken on other kernels than the ones loaded now (the source hash of the kernel's translation unit, written by
    the build into lib/build_manifest.json and by the PMC step into the record, must agree): a stale figure is not reported."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get(kernel_key)
    except (OSError, ValueError):
        return None
    if not rec:
        return None
    have = kernel_source_hash(kernel_key)
    if not have or rec.get("src_hash") != have:
        return None
    return {"bytes_per_launch": int((2 * rec["fetch_kib"] + rec["write_kib"]) * 1024), "fetch_size_kib": rec["fetch_kib"],
            "write_size_kib": rec["write_kib"], "source": rec.get("source"), "kernel_src_hash": have,
            "note": "separate rocprofv3 --pmc passes, summed over the kernel's launches of one step; FETCH_SIZE doubled (gfx950)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("ECGPU_BENCH_WORKLOAD", "auto"))
    ap.add_argument("--validators", type=int, default=1 << 20)
    ap.add_argument("--tuples", type=int, default=65536)
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="bls workload with N ranks: weak = --tuples per rank, strong = --tuples in all (rank g takes shard_range)")
    ap.add_argument("--no-extras", action="store_true", help="skip the strong_2p20 / epoch / slots sub-records of the default line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aggregates", action="store_true", help="skip the secondary K = 2048 aggregates line")
    ap.add_argument("--no-calibration", action="store_true",
                    help="warm up without the timed batches that place the dispatch thresholds (counter passes: the per-launch means of "
                         "tools/pmc_to_json.py must not mix in the calibration's launches)")
    ap.add_argument("--even-shards", action="store_true", help="strong scaling: keep the even split (no speed-weighted re-sharding)")
    ap.add_argument("--verbose", action="store_true",
                    help="print the full record (every note and provenance string) instead of the compact line; the full record is "
                         "always written to gpurun_out/bench_full.json as well")
    return ap.parse_args()


def cpu_baseline_merkle(n_validators: int):
    """oracle/c restatement (SHA-NI when the host has it) on 1 and on 8 host threads: htr(List<Validator>)
    of the same registry -- 93 % of the state's hash64.  Reported, never the target."""
    from oracle import cref
    from ethereum_consensus_amd import synthetic as S
    enc = S.validators(n_validators).tobytes()
    best = None
    hashes = 0
    t_total = time.time()
    reps = 0
    while reps < 3 or (time.time() - t_total < 6 and reps < 8):
        t = time.time()
        root1, hashes = cref.htr_validators(enc)
        dt = time.time() - t
        best = dt if best is None else min(best, dt)
        reps += 1
    best8, reps8 = None, 0
    t_total = time.time()
    while reps8 < 3 or (time.time() - t_total < 4 and reps8 < 8):
        t = time.time()
        root8, _ = cref.htr_validators_threads(enc, 8)
        dt = time.time() - t
        best8 = dt if best8 is None else min(best8, dt)
        reps8 += 1
    assert root8 == root1, "threaded C restatement disagrees with the single-thread root"
    eff = effective_cores()
    return {"value": hashes / best8, "unit": "leaves/s", "cores": 8, "cores_effective": min(8, eff["cores_effective"]), "kind": "port",
            "host": eff, "measured_parallel_speedup": best / best8,
            "one_thread": {"value": hashes / best, "unit": "leaves/s", "cores": 1, "sample": f"best of {reps}"},
            "sample": f"htr(List<Validator,2^40>) of the same {n_validators} validators ({hashes} hash64), "
                      f"oracle/c/sha256_merkle.c, sha_ni={int(cref.lib().oc_have_shani())}: 8 threads (aligned subtrees, best of {reps8}) "
                      f"and 1 thread (best of {reps})"}


def run_merkle(args, L, torch, dist, rank, world):
    from ethereum_consensus_amd import synthetic as S
    n = args.validators
    enc = S.beacon_state_deneb(n, "mainnet", seed=1 + rank)
    fixed = int(L.ecgpu_beacon_state_deneb_fixed_size(0))
    h_fixed = ctypes.create_string_buffer(enc[:fixed], fixed)
    import numpy as np
    d_state = torch.from_numpy(np.frombuffer(enc, dtype=np.uint8).copy()).cuda()
    d_root = torch.zeros(32, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    # N > 1: the path's only exchange, every step: every rank learns every shard's root (32 B per rank).  The gather of step k
    # is ASYNCHRONOUS -- two root buffers alternate, the collective of step k runs on RCCL's stream while step k + 1 computes,
    # and every outstanding gather is waited for before the clock stops: a 32-byte all-gather is ~20-40 us of pure latency,
    # 2-4 % of a 1 ms step if the next root waited for it.
    d_root2 = [d_root, torch.zeros_like(d_root)]
    multi = _multi(dist, world)  # (a single-rank group with the collectives forced: tests/test_gpu_dist.py)
    gathered = [torch.empty(32 * world, dtype=torch.uint8, device="cuda") for _ in range(2)] if multi else None
    pending = [None, None]
    turn = {"k": 0}

    def step():
        k = turn["k"] & 1
        turn["k"] += 1
        if pending[k] is not None:  # the buffer pair of two steps ago: its gather has long finished
            pending[k].wait()
            pending[k] = None
        rc = L.ecgpu_htr_beacon_state_deneb_dev(d_state.data_ptr(), len(enc), h_fixed, 0, d_root2[k].data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(f"ecgpu_htr_beacon_state_deneb_dev -> {rc}: {L.ecgpu_last_error()}")
        if multi:
            pending[k] = dist.all_gather_into_tensor(gathered[k], d_root2[k], async_op=True)

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for _ in range(max(args.warmup, 1)):
        step()
    drain()
    torch.cuda.synchronize()
    hashes = int(L.ecgpu_last_hash64_count())
    dom = b"merkle_pass_validators"
    L.ecgpu_prof_filter(dom)
    L.ecgpu_prof_enable(1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if turn["k"] & 1:  # the last root sits in the second buffer
        d_root.copy_(d_root2[1])
    ms = ctypes.c_double(0)
    nl = ctypes.c_uint64(0)
    L.ecgpu_prof_read(dom, ctypes.byref(ms), ctypes.byref(nl))
    L.ecgpu_prof_enable(0)
    kern_ms = ms.value / max(args.steps, 1)  # per state: one launch of the pass
    # Throughput with TWO roots in flight (world == 1): consecutive roots alternate between two streams, so the latency-bound
    # tail of one (a 45-deep chain of dependent hash64 on a handful of workgroups) runs underneath the chip-filling validator
    # pass of the next.  A caller with independent states to root (fork choice over several heads, checkpoint sync verifying
    # a batch of states) gets this rate; a caller that needs root N before it can build state N + 1 (process_slots) gets
    # `ms_per_step`.  Reported next to the headline, never instead of it.
    pipelined = None
    if world == 1:
        s_alt = [torch.cuda.Stream(), torch.cuda.Stream()]
        d_roots2 = [torch.zeros(32, dtype=torch.uint8, device="cuda") for _ in range(2)]
        n_pipe = 2 * max(4, args.steps // 2)
        for k in range(4):  # warm both arenas
            rc = L.ecgpu_htr_beacon_state_deneb_dev(d_state.data_ptr(), len(enc), h_fixed, 0, d_roots2[k & 1].data_ptr(), s_alt[k & 1].cuda_stream)
            if rc != 0:
                raise RuntimeError(f"ecgpu_htr_beacon_state_deneb_dev -> {rc}: {L.ecgpu_last_error()}")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(n_pipe):
            rc = L.ecgpu_htr_beacon_state_deneb_dev(d_state.data_ptr(), len(enc), h_fixed, 0, d_roots2[k & 1].data_ptr(), s_alt[k & 1].cuda_stream)
            if rc != 0:
                raise RuntimeError(f"ecgpu_htr_beacon_state_deneb_dev -> {rc}: {L.ecgpu_last_error()}")
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ok2 = all(bytes(r.cpu().numpy()) == bytes(d_root.cpu().numpy()) for r in d_roots2)
        pipelined = {"ms_per_root": (t2 - t1) / n_pipe * 1e3, "leaves_per_s": hashes * n_pipe / (t2 - t1), "roots": n_pipe, "streams": 2,
                     "roots_equal_the_one_stream_root": ok2,
                     "note": "two independent roots in flight on two streams: throughput for callers with independent states; "
                             "ms_per_step above is the one-root-at-a-time latency"}
    # SURVEY.md 8(d) config 3 asks for the root "timed device-resident AND including H2D": the host-pointer entry
    # (ecgpu_htr_beacon_state: upload of the whole encoding, then the same kernels) from pageable and from pinned host memory.
    # This is what an un-patched `state.hash_tree_root()` caller pays per slot on top of its own serialization; the resident
    # state (slots workload) is what avoids it.  Never `value`.
    h2d = None
    if world == 1:
        h_root = ctypes.create_string_buffer(32)
        pinned = torch.from_numpy(np.frombuffer(enc, dtype=np.uint8).copy()).pin_memory()
        h2d = {}
        for name, ptr in (("pageable_ms", ctypes.cast(ctypes.c_char_p(enc), ctypes.c_void_p).value), ("pinned_ms", pinned.data_ptr())):
            best = None
            for _ in range(4):
                t1 = time.perf_counter()
                rc = L.ecgpu_htr_beacon_state(4, ptr, len(enc), 0, h_root)
                t2 = time.perf_counter()
                if rc != 0:
                    raise RuntimeError(f"ecgpu_htr_beacon_state -> {rc}: {L.ecgpu_last_error()}")
                best = (t2 - t1) if best is None else min(best, t2 - t1)
            h2d[name] = best * 1e3
        h2d["root_equals_resident_root"] = h_root.raw == bytes(d_root.cpu().numpy())
        h2d["bytes_uploaded"] = len(enc)
        h2d["note"] = "host entry ecgpu_htr_beacon_state: H2D copy of the encoding + root + 32-byte D2H, best of 4"
    # algorithmic bytes of the dominant kernel per state: 121 B read per validator + one
    # 32-byte node written per lane (2^D validators per lane, D from the schedule)
    lanes = n >> max(1, min(6, n.bit_length() - 1 - 18))
    alg_bytes = 121 * n + 32 * lanes
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    val_hashes = 8 * n + (n - lanes)  # hash64 executed inside the dominant kernel
    return dict(
        dt=dt, units_per_step=hashes, metric="merkle_leaves_hashed_per_sec", unit="leaves/s", dtype="u32",
        config={"workload": f"hash_tree_root(BeaconState) deneb mainnet, {n} validators, SSZ-encoded state "
                            f"({len(enc)} B) resident in HBM", "hash64_per_state": hashes, "state_bytes": len(enc),
                "sharding": "one independent state per GPU; all-gather of the 32-byte roots"},
        roofline={"bound": "hbm", "kernel": "k_merkle_pass<2, ValidatorLeaves>", "achieved": achieved,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                  "traffic": (pmc_traffic("k_merkle_pass<2, ValidatorLeaves>") or {}).get("bytes_per_launch"),
                  "traffic_source": "committed pmc pass" if pmc_traffic("k_merkle_pass<2, ValidatorLeaves>") else None,
                  "traffic_detail": pmc_traffic("k_merkle_pass<2, ValidatorLeaves>"),
                  "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms, "launches_timed": int(nl.value),
                  "launch_note": "one launch of the validator pass per state root",
                  "valu_int": {"unit": "G hash64/s", "achieved": val_hashes / (kern_ms * 1e-3) / 1e9 if kern_ms else 0.0,
                               "peak": HASH64_PEAK_GHS,
                               "frac": (val_hashes / (kern_ms * 1e-3) / 1e9 / HASH64_PEAK_GHS) if kern_ms else 0.0,
                               "whole_state_frac": hashes / (dt / args.steps) / 1e9 / HASH64_PEAK_GHS,
                               "note": "the path is integer-VALU bound: 2410 VALU instructions per 64-byte hash64; peak = "
                                       "register-resident hash64 chains at 8 waves/SIMD (tools/fpbench.hip)"}},
        root=bytes(d_root.cpu().numpy()).hex(),
        extra=({"h2d_inclusive": h2d} if h2d else {}) | ({"two_roots_in_flight": pipelined} if pipelined else {}),
    )


def run_merkle_sharded(args, L, torch, dist, rank, world, emulate_world=None):
    """BASELINE north_star's Merkle half, strong-scaled: ONE deneb mainnet state of --validators validators over the ranks
    (the reference's single call: state.hash_tree_root(), phase0/slot_processing.rs:67).  Every rank holds the state's encoding
    (the caller's placement; rank g READS only its subtree of each registry-sized list), reduces its aligned subtrees (phase A),
    the ranks all-gather 5 x 32 bytes each -- the path's only collective --, and every rank finishes the five lists, the other
    fields and the root (phase B).  The root is asserted equal to the unsharded root of the same state on every rank.
    emulate_world = W (single process): the W ranks' phase A run one after the other on this GPU and the exchange is a device
    copy -- a parity check of the sharded path (and its per-phase cost) on a one-GPU box, not a scaling number."""
    from ethereum_consensus_amd import synthetic as S
    import numpy as np
    n = args.validators
    dev = _device(torch)
    enc = S.beacon_state_deneb(n, "mainnet", seed=1)  # the same state on every rank
    fork = 4
    fixed = int(L.ecgpu_beacon_state_fixed_size(fork, 0))
    h_fixed = ctypes.create_string_buffer(enc[:fixed], fixed)
    d_state = torch.from_numpy(np.frombuffer(enc, dtype=np.uint8).copy()).to(dev)
    nl = int(L.ecgpu_beacon_state_shard_lists())
    w_eff = emulate_world or world
    d_sub = torch.zeros(32 * nl, dtype=torch.uint8, device=dev)
    d_all = torch.zeros(32 * nl * w_eff, dtype=torch.uint8, device=dev)
    d_root = torch.zeros(32, dtype=torch.uint8, device=dev)
    d_ref = torch.zeros(32, dtype=torch.uint8, device=dev)
    d_keep = torch.zeros(64 * 32, dtype=torch.uint8, device=dev)  # the other fields' roots: phase A leaves them, phase B takes them
    stream = _stream(torch)

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} -> {rc}: {L.ecgpu_last_error()}")

    chk(L.ecgpu_htr_beacon_state_dev(fork, d_state.data_ptr(), len(enc), h_fixed, 0, d_ref.data_ptr(), stream), "ecgpu_htr_beacon_state_dev")
    _sync(torch)
    hashes_unsharded = int(L.ecgpu_last_hash64_count())
    hashes = {"a": 0, "b": 0}

    def step():
        if emulate_world:
            for r in range(emulate_world):
                chk(L.ecgpu_beacon_state_shard_subroots_dev(fork, d_state.data_ptr(), len(enc), h_fixed, 0, r, emulate_world,
                                                            d_all.data_ptr() + 32 * nl * r, d_keep.data_ptr(), stream), "ecgpu_beacon_state_shard_subroots_dev")
            gathered = d_all
        else:
            chk(L.ecgpu_beacon_state_shard_subroots_dev(fork, d_state.data_ptr(), len(enc), h_fixed, 0, rank, world, d_sub.data_ptr(),
                                                        d_keep.data_ptr(), stream), "ecgpu_beacon_state_shard_subroots_dev")
            hashes["a"] = int(L.ecgpu_last_hash64_count())
            # the path's only collective: 160 bytes per rank
            from ethereum_consensus_amd import shard
            gathered = shard.all_gather_bytes(dist, d_sub, world, force=FORCE_DIST) if _multi(dist, world) else d_sub
        chk(L.ecgpu_htr_beacon_state_sharded_dev(fork, d_state.data_ptr(), len(enc), h_fixed, 0, gathered.data_ptr(), w_eff, d_keep.data_ptr(),
                                                 d_root.data_ptr(), stream), "ecgpu_htr_beacon_state_sharded_dev")
        hashes["b"] = int(L.ecgpu_last_hash64_count())
        return gathered

    for _ in range(max(args.warmup, 1)):
        keep = step()
    _sync(torch)
    if _multi(dist, world):
        dist.barrier()
    _sync(torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        keep = step()
    _sync(torch)
    if _multi(dist, world):
        dist.barrier()
    dt = time.perf_counter() - t0
    del keep
    # the two phases on their own (this rank's phase A; phase B), a few repetitions each: what an N-rank run adds to their sum is
    # the all-gather
    phases = None
    if DEV == "cuda":
        def timed(fn, reps=5):
            fn()
            _sync(torch)
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            _sync(torch)
            return (time.perf_counter() - t) / reps * 1e3
        r_a, w_a = (0, emulate_world) if emulate_world else (rank, world)
        a_ms = timed(lambda: chk(L.ecgpu_beacon_state_shard_subroots_dev(fork, d_state.data_ptr(), len(enc), h_fixed, 0, r_a, w_a,
                                                                        d_sub.data_ptr(), d_keep.data_ptr(), stream), "phase A"))
        g_all = d_all if (emulate_world or world > 1 or FORCE_DIST) else d_sub
        if not emulate_world and (world > 1 or FORCE_DIST):
            from ethereum_consensus_amd import shard as _sh
            g_all = _sh.all_gather_bytes(dist, d_sub, world, force=FORCE_DIST)
        b_ms = timed(lambda: chk(L.ecgpu_htr_beacon_state_sharded_dev(fork, d_state.data_ptr(), len(enc), h_fixed, 0, g_all.data_ptr(), w_eff,
                                                                     d_keep.data_ptr(), d_root.data_ptr(), stream), "phase B"))
        phases = {"phase_a_ms_this_rank": a_ms, "phase_b_ms": b_ms,
                  "note": "each phase alone, enqueue to completion; phase A = this rank's subtrees of the five lists + every other "
                          "field, phase B = the five list tops + the state container"}
    root = bytes(d_root.cpu().numpy())
    ref = bytes(d_ref.cpu().numpy())
    same = root == ref
    return dict(
        dt=dt, units_per_step=hashes_unsharded / world, metric="merkle_leaves_hashed_per_sec", unit="leaves/s", dtype="u32", scaling="strong",
        config={"workload": f"hash_tree_root(BeaconState) deneb mainnet, {n} validators: ONE state over {w_eff} "
                            f"{'emulated ranks on one GPU' if emulate_world else 'rank(s)'}, SSZ-encoded state ({len(enc)} B) resident in HBM",
                "hash64_per_state": hashes_unsharded, "hash64_this_rank_phase_a": hashes["a"], "hash64_every_rank_phase_b": hashes["b"],
                "state_bytes": len(enc),
                "sharding": "the five registry-sized lists (validators, balances, 2 x participation, inactivity_scores) as aligned "
                            "power-of-two subtrees per rank, the other fields redundantly on every rank underneath its validator pass "
                            "(phase A); all-gather of 5 x 32 bytes per rank; the five list tops and the state container on every rank "
                            "(phase B)"},
        roofline={"bound": "hbm", "kernel": "k_merkle_pass<2, ValidatorLeaves>", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": None, "traffic": None, "note": "latency-bound at N > 1: see the N = 1 `merkle` record for the pass kernel's roofline"},
        check={"root": root.hex(), "equals_unsharded_root": same},
        extra={"phases": phases} if phases else {},
    )


R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
# Integer multiplies of ONE K = 1 verification, counted on the lane programs themselves (tests/hostsim hs_op_census, mean over
# the first 16 tuples of this workload): (fp_mul calls, fp_sqr calls, multiplies inside sums of products) per stage, for each
# build of the kernels.  An Fp product is 351 (273 for a square) multiply instructions (v_mad_u64_u32 / v_mul_lo_u32), a sum
# of N products with one reduction 169 N + 182, see csrc/bls_fp.h.  The census used is the one of the build that RAN
# (ecgpu_bls_tower(): 1 = sums of products, 2 = compact-code G2 stage kernels; tools/bls_op_census.py prints these tuples).
BLS_OPS_BY_BUILD = {
    1: {"bls_pk_validate": (441, 807, 65520), "bls_sig": (170, 756, 434434), "bls_h2c": (361, 1520, 1076572),
        "bls_pairing": (588, 22, 6592154)},
    # compact-code build of the G2 stages (boxes with slow instruction fetch); the pairing check runs on the lane groups there
    2: {"bls_pk_validate": (441, 807, 65520), "bls_sig": (1428, 756, 0), "bls_h2c": (3311, 1520, 7920), "bls_pairing": (588, 22, 6592154)},
}
BLS_OPS = BLS_OPS_BY_BUILD[1]
PAIRING_KERNEL_BY_BUILD = {1: "k_pairing", 2: "k_pairing"}
BLS_MULTS_PER_SIG = sum(m * 351 + s * 273 + x for m, s, x in BLS_OPS.values())
BLS_BYTES_PER_SIG = 48 + 32 + 96 + 1  # K = 1 tuple in, status byte out


def vm3_multiplies_per_tuple() -> int:
    """useful multiply instructions of one pairing check on the sum-of-products lane groups, as counted by the generator
    (tools/gen_bls_vm3.py writes them into the program header)"""
    import re
    try:
        text = open(os.path.join(ROOT, "ethereum_consensus_amd", "csrc", "bls_vm3_prog.h")).read()
        return sum(int(x) for x in re.findall(r"#define ECG_VM3_[AC]_MADS (\d+)", text))
    except OSError:
        return 0


def S(tag: bytes, i: int) -> bytes:
    import hashlib
    return hashlib.sha256(b"ecgpu/v1/" + tag + b"/" + i.to_bytes(4, "little")).digest()


def bls_inputs(n: int, base: int):
    """sk_i = 1 + S("sk",i) mod (r-1), msg_i = S("msg",i) (32 B)."""
    sks = b"".join((1 + int.from_bytes(S(b"sk", base + i), "big") % (R_ORDER - 1)).to_bytes(32, "big") for i in range(n))
    msgs = b"".join(S(b"msg", base + i) for i in range(n))
    return sks, msgs


BLST_SIGS_PER_CORE = 1350.0  # midpoint of the published 1.2-1.5 k verifications/s per core (blst, x86-64 with ADX)


def cpu_baseline_bls(sample, budget_s: float = 20.0):
    """oracle/c/bls12_381.cpp -- the C++ restatement of the blst behaviour (6 x 64-bit Montgomery limbs, unsigned __int128,
    -O3) -- on 1 and on all host threads, on the FIRST tuples of the very workload the GPU verified (host copies of the same
    keys, messages, signatures and fault cycle; statuses asserted equal to the ones known by construction).  Reported, never
    the target.  blst itself cannot be built offline; its published figure is ~1.2-1.5 k verifications/s per core, i.e.
    several times this restatement."""
    from oracle import cbls
    pks, msgs, sigs, want = sample
    eff = effective_cores()
    nthr = max(1, min(cbls.host_threads(), eff["cores_effective"]))  # the threads the box grants, not the 256 it shows
    # calibrate on one thread, then size the all-thread sample for ~budget_s / 2 of wall time
    m1 = min(len(want), 256)
    t0 = time.time()
    st1 = cbls.fast_aggregate_verify_batch_k1(pks[:48 * m1], msgs[:32 * m1], sigs[:96 * m1], 1)
    dt_1 = time.time() - t0
    assert st1 == want[:m1], "C++ restatement disagrees with the statuses known by construction"
    rate1 = m1 / dt_1
    # "1 and 8 threads" -- and all of the host's, which on a cgroup-limited box is not much more than 8
    m8 = int(min(len(want), max(512, rate1 * 8 * 3)))
    t0 = time.time()
    st8 = cbls.fast_aggregate_verify_batch_k1(pks[:48 * m8], msgs[:32 * m8], sigs[:96 * m8], 8)
    dt_8 = time.time() - t0
    assert st8 == want[:m8], "C++ restatement disagrees with the statuses known by construction"
    m = int(min(len(want), max(64 * nthr, rate1 * nthr * budget_s / 2)))
    t0 = time.time()
    st = cbls.fast_aggregate_verify_batch_k1(pks[:48 * m], msgs[:32 * m], sigs[:96 * m], nthr)
    dt_n = time.time() - t0
    assert st == want[:m], "C++ restatement disagrees with the statuses known by construction"
    speedup = (m / dt_n) / rate1
    # the threads used are as many as the cgroup quota grants; where that is not readable they are the affinity mask's and the
    # measured speed-up over one thread says what was granted (round 2 printed "cores": 256 beside an 8.9x speed-up)
    granted = eff["cores_effective"] if eff["cgroup_quota_cores"] is not None else min(eff["affinity"], max(1, int(speedup + 0.999)))
    return {"value": m / dt_n, "unit": "sigs/s", "cores": nthr, "cores_effective": granted, "host": eff,
            "measured_parallel_speedup": speedup, "kind": "port",
            # blst is not buildable offline; its published single-core rate for one K = 1 verification (two Miller loops, one
            # final exponentiation, hash-to-G2, both subgroup checks) is ~1.2-1.5 k/s: the estimate is the midpoint times the
            # threads used here.  A field, so that nobody has to compute a GPU/CPU ratio from the slower restatement
            "blst_equivalent_estimate": {"value": BLST_SIGS_PER_CORE * nthr, "unit": "sigs/s", "cores": nthr,
                                         "per_core": BLST_SIGS_PER_CORE, "basis": "published blst figure, not measured here"},
            "one_thread": {"value": rate1, "unit": "sigs/s", "cores": 1, "sample": f"first {m1} tuples in {dt_1:.1f} s"},
            "eight_threads": {"value": m8 / dt_8, "unit": "sigs/s", "cores": 8, "sample": f"first {m8} tuples in {dt_8:.1f} s"},
            "sample": f"first {m} K = 1 tuples of the same workload (fault cycle included, statuses equal to construction) in {dt_n:.1f} s on "
                      f"{nthr} threads; oracle/c/bls12_381.cpp, g++ -O3 -march=x86-64-v3, 6 x 64-bit limbs with unsigned __int128 "
                      "(blst itself is not available offline: ~1.2-1.5 k/s per core published)"}


BLS_SHARD_GRANULE = 1024  # weighted shards are cut at multiples of this many tuples


def _bls_setup(L, torch, n, lo):
    """the shard [lo, lo + n) of the K = 1 workload, resident in HBM: keys and signatures made on the device, the fault cycle
    injected on the host copy (statuses known by construction)"""
    dev = _device(torch)
    sks, msgs = bls_inputs(n, lo)
    d_sk = torch.frombuffer(bytearray(sks or b"\0"), dtype=torch.uint8).to(dev)
    msgs = bytearray(msgs)
    stream = _stream(torch)
    d_pk = torch.empty(max(48 * n, 1), dtype=torch.uint8, device=dev)
    d_sig = torch.empty(max(96 * n, 1), dtype=torch.uint8, device=dev)
    d_msg_clean = torch.frombuffer(bytearray(msgs or b"\0"), dtype=torch.uint8).to(dev)
    # workload generation on the device (SecretKey::public_key / sign, crypto/bls.rs:193-219); untimed
    if n:
        assert L.ecgpu_sk_to_pk_batch_dev(d_sk.data_ptr(), n, d_pk.data_ptr(), stream) == 0
        assert L.ecgpu_sign_batch_dev(d_sk.data_ptr(), 32, d_msg_clean.data_ptr(), n, d_sig.data_ptr(), stream) == 0
    # fault injection of SURVEY.md 8(d) config 2: tuple i = 0 (mod 64) of the shard is corrupted, cycling through eight fault
    # classes (wrong message, swapped key, signature outside G2, key outside G1, bad flag bits, x >= p, key = infinity,
    # signature = infinity); the expected status of every tuple is known by construction
    from ethereum_consensus_amd import synthetic as syn
    _sync(torch)
    h_pk = bytearray(d_pk.cpu().numpy().tobytes()[:48 * n])
    h_sig = bytearray(d_sig.cpu().numpy().tobytes()[:96 * n])
    want_bytes, _kinds = syn.bls_inject_faults(h_pk, msgs, h_sig, n)
    d_pk = torch.frombuffer(h_pk or bytearray(1), dtype=torch.uint8).to(dev)
    d_sig = torch.frombuffer(h_sig or bytearray(1), dtype=torch.uint8).to(dev)
    d_msg = torch.frombuffer(msgs or bytearray(1), dtype=torch.uint8).to(dev)
    d_st = torch.full((max(n, 1),), 0xFF, dtype=torch.uint8, device=dev)
    _sync(torch)
    return d_pk, d_msg, d_sig, d_st, h_pk, msgs, h_sig, want_bytes, stream


def run_bls(args, L, torch, dist, rank, world, n_total=None, strong=None):
    """K = 1 tuples.  weak scaling (default): every rank verifies `--tuples` tuples of its own.  strong (`--scaling strong`,
    north_star's 2^20-signature batch): `--tuples` in all, rank g verifies shard_range(total, g, N) -- ragged shards, one
    all-gather of the status bytes per step either way."""
    from ethereum_consensus_amd import shard
    strong = (args.scaling == "strong") if strong is None else strong
    total = args.tuples if n_total is None else n_total
    weights = None  # strong scaling only: per-rank speeds once measured (speed-weighted shards)
    balance = {"weights": None}
    while True:
        if strong:
            lo, hi = shard.shard_range(total, rank, world, weights, BLS_SHARD_GRANULE)
        else:
            lo, hi = rank * total, (rank + 1) * total
        n = hi - lo
        setup = _bls_setup(L, torch, n, lo)
        d_pk, d_msg, d_sig, d_st, h_pk, msgs, h_sig, want_bytes, stream = setup
        gathered = {}

        def step():
            if n:
                rc = L.ecgpu_fast_aggregate_verify_batch_dev(d_pk.data_ptr(), None, n, d_msg.data_ptr(), d_sig.data_ptr(), n, 0,
                                                             d_st.data_ptr(), stream)
                if rc != 0:
                    raise RuntimeError(f"ecgpu_fast_aggregate_verify_batch_dev -> {rc}: {L.ecgpu_last_error()}")
            if _multi(dist, world):
                # the path's only collective: every rank learns every shard's verify statuses
                gathered["st"] = (shard.all_gather_ragged(dist, d_st[:n], total, world, force=FORCE_DIST, weights=weights,
                                                          granule=BLS_SHARD_GRANULE) if strong
                                  else shard.all_gather_bytes(dist, d_st, world, force=FORCE_DIST))

        if not (strong and _multi(dist, world) and weights is None and world > 1 and not getattr(args, "even_shards", False)):
            break
        # Speed-weighted shards: one GPU with slow instruction fetch runs this batch
        # 1.8 x slower and an even split makes every step wait for it.  Each rank times its own kernels on its
        # even shard (no collective inside), the speeds are all-gathered, and unless the node is homogeneous the batch is cut
        # again in proportion.
        local_step = lambda: L.ecgpu_fast_aggregate_verify_batch_dev(d_pk.data_ptr(), None, n, d_msg.data_ptr(), d_sig.data_ptr(), n, 0,
                                                                     d_st.data_ptr(), stream) if n else 0
        local_step()
        _sync(torch)
        t_probe = time.perf_counter()
        for _ in range(2):
            local_step()
        _sync(torch)
        mine = (2 * n) / max(time.perf_counter() - t_probe, 1e-9) * float(os.environ.get("ECGPU_BENCH_SPEED_SCALE", "1"))
        speeds = shard.gather_speeds(dist, mine, world, device=_device(torch))
        balance = {"speeds_tuples_per_s": speeds, "weights": None}
        if shard.balanced_enough(speeds):
            break
        weights = speeds
        balance["weights"] = [s_ / sum(speeds) for s_ in speeds]
    dev = _device(torch)
    for _ in range(max(args.warmup, 1)):
        step()
    _sync(torch)
    L.ecgpu_prof_filter(None)
    L.ecgpu_prof_enable(1)
    if _multi(dist, world):
        dist.barrier()
    _sync(torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _sync(torch)
    if _multi(dist, world):
        dist.barrier()
    dt = time.perf_counter() - t0
    build = int(L.ecgpu_bls_tower())
    ops = dict(BLS_OPS_BY_BUILD.get(build, BLS_OPS))
    pairing_kernel = PAIRING_KERNEL_BY_BUILD.get(build, "k_pairing")
    path = int(L.ecgpu_bls_last_pairing_path())
    if path == 3:  # the sum-of-products lane groups ran the pairing check (ECGPU_PAIRING=vm3, or a box with slow instruction fetch)
        pairing_kernel = "k_vm3_pair_a + k_vm3_pair_c"
        ops["bls_pairing"] = (0, 0, vm3_multiplies_per_tuple())
    if path == 7:  # the row machine (csrc/bls_row.hip): the lane groups' programs, one Fp operation per 16-lane row
        pairing_kernel = "k_row_pair_a + k_row_pair_c"
        ops["bls_pairing"] = (0, 0, vm3_multiplies_per_tuple())
    if path == 5:  # Miller loop on two lanes per tuple; final exponentiation on one lane, or (up to half a round of lanes, or when
        # ECGPU_FINALEXP_LANES=2) on the lane pair as well: the same multiplies either way
        pairing_kernel = "k_miller2_w1 + k_finalexp2_w1"  # (the two-wave builds and the one-lane final exponentiation: experiments library only)
    mults_per_sig = sum(m * 351 + s_ * 273 + x for m, s_, x in ops.values())
    stages = {}
    for tag in ops:
        ms, cnt = _prof(L, tag)
        stages[tag] = ms / max(cnt, 1)
    pairing_parts = None
    if path == 5:
        pairing_parts = {}
        for tag in ("bls_miller2", "bls_finalexp"):
            ms, cnt = _prof(L, tag)
            pairing_parts[tag] = ms / max(cnt, 1)
    L.ecgpu_prof_enable(0)
    st = d_st[:n].cpu().numpy()
    import numpy as np
    want = np.frombuffer(bytes(want_bytes), dtype=np.uint8)
    ok = bool((st == want).all())
    if _multi(dist, world) and strong:
        # every rank holds the whole job's statuses after the gather: its own shard sits where shard_range puts it
        full = gathered["st"].cpu().numpy()
        ok = ok and full.shape[0] == total and bool((full[lo:hi] == want).all())
    dom = max(stages, key=lambda k: stages[k])
    kern_ms = stages[dom]
    alg_bytes = BLS_BYTES_PER_SIG * n
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    mul_ops = {k: (m * 351 + s_ * 273 + x) * n for k, (m, s_, x) in ops.items()}
    # None unless a PMC pass of THIS kernel build is committed under profiles/
    if path == 3:
        ta, tc = pmc_traffic("k_vm3_pair_a"), pmc_traffic("k_vm3_pair_c")
        traffic = None if not (ta and tc) else {"bytes_per_launch": ta["bytes_per_launch"] + tc["bytes_per_launch"], "parts": [ta, tc]}
    else:
        traffic = pmc_traffic(pairing_kernel)
    if path == 5:
        ta, tc = pmc_traffic("k_miller2"), pmc_traffic("k_finalexp")
        traffic = None if not (ta and tc) else {"bytes_per_launch": ta["bytes_per_launch"] + tc["bytes_per_launch"], "parts": [ta, tc]}
    m_cpu = min(n, 16384)
    return dict(
        host_sample=(bytes(h_pk[:48 * m_cpu]), bytes(msgs[:32 * m_cpu]), bytes(h_sig[:96 * m_cpu]), bytes(want_bytes[:m_cpu])),
        dt=dt, units_per_step=(total / world) if strong else n, metric="bls_signatures_verified_per_sec", unit="sigs/s", dtype="u32",
        scaling="strong" if strong else "weak",
        config={"workload": f"fast_aggregate_verify of {total if strong else n} synthetic (pk, msg, sig) tuples{' in all' if strong else ''}, K = 1, 32-byte messages, "
                            "1/64 tuples corrupted, cycling through 8 fault classes (wrong message, swapped key, signature outside G2, key "
                            "outside G1, bad flags, x >= p, key = infinity, signature = infinity); compressed keys/messages/signatures "
                            "resident in HBM",
                "tuples": total if strong else n, "tuples_this_rank": n,
                "semantics": "reference: every key decompressed + subgroup-checked, every signature "
                             "decompressed + subgroup-checked, every message hashed to G2, per-tuple pairing check",
                "sharding": ("strong scaling: the batch is fixed, rank g verifies shard_range(total, g, N); ragged all-gather of the status "
                             "bytes every step") if strong else "one independent batch per GPU; all-gather of the status bytes every step"},
        roofline={"bound": "hbm", "kernel": pairing_kernel, "kernel_build": ("sum-of-products lane groups (vm3)" if path == 3 else "row machine" if path == 7 else "two lanes per tuple (Miller loop) + one lane (final exponentiation)" if path == 5
                                   else {1: "sums of products", 2: "compact-code tower"}.get(build, "?")),
                  "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": achieved / HBM_PEAK_GBS, "traffic": (traffic or {}).get("bytes_per_launch"), "traffic_detail": traffic,
                  "traffic_source": "committed pmc pass" if traffic else None,
                  "algorithmic_bytes_per_launch": alg_bytes,
                  "avg_launch_ms": kern_ms, "stage_ms": stages, "pairing_parts_ms": pairing_parts,
                  "valu_int": {"unit": "T multiplies/s (v_mad_u64_u32 + v_mul_lo_u32)", "peak": MUL_PIPE_PEAK_TOPS,
                               "one_wave_ceiling": MUL_PIPE_ONE_WAVE_TOPS, "two_waves_ceiling": MUL_PIPE_TWO_WAVES_TOPS,
                               "peak_source": "profiles/r02p_issue_rates.txt: v_mad_u64_u32 at 8 / 2 / 1 waves per SIMD",
                               "achieved": {k: (mul_ops[k] / (stages[k] * 1e-3) / 1e12 if stages[k] > 0 else 0.0) for k in stages},
                               "frac": {k: (mul_ops[k] / (stages[k] * 1e-3) / 1e12 / MUL_PIPE_PEAK_TOPS if stages[k] > 0 else 0.0) for k in stages},
                               "multiplies_per_signature": mults_per_sig,
                               "note": f"the path is integer-multiplier bound, not HBM bound: {mults_per_sig / 1e6:.1f} M multiplies vs 177 B per "
                                       "signature (census of the kernel build that ran)"}},
        check={"statuses_match_construction": ok, "expected_failures": int(want.astype(bool).sum())},
        extra=({"weights": balance} if strong and _multi(dist, world) else {}),
    )


def run_bls_aggregate(args, L, torch, dist, rank, world, n_agg=256, k=2048):
    """BASELINE.json configs[3] per-GPU share: 256 committee aggregates of K = 2048 keys each (32 slots x 64 committees
    over 8 GPUs), reference semantics (every key decompressed + subgroup-checked on every call).  Secondary line."""
    import numpy as np
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    n_keys = n_agg * k
    sks, _ = bls_inputs(n_keys, rank * n_keys)
    d_sk = torch.frombuffer(bytearray(sks), dtype=torch.uint8).to(dev)
    d_pk = torch.empty(48 * n_keys, dtype=torch.uint8, device=dev)
    assert L.ecgpu_sk_to_pk_batch_dev(d_sk.data_ptr(), n_keys, d_pk.data_ptr(), stream) == 0
    agg_sk = b"".join((sum(int.from_bytes(sks[32 * j:32 * j + 32], "big") for j in range(c * k, (c + 1) * k)) % R_ORDER).to_bytes(32, "big")
                      for c in range(n_agg))
    msgs = bytearray(b"".join(S(b"att", rank * n_agg + c) for c in range(n_agg)))
    d_ask = torch.frombuffer(bytearray(agg_sk), dtype=torch.uint8).to(dev)
    d_msg_clean = torch.frombuffer(bytearray(msgs), dtype=torch.uint8).to(dev)
    d_sig = torch.empty(96 * n_agg, dtype=torch.uint8, device=dev)
    assert L.ecgpu_sign_batch_dev(d_ask.data_ptr(), 32, d_msg_clean.data_ptr(), n_agg, d_sig.data_ptr(), stream) == 0
    for c in range(0, n_agg, 64):  # 1/64 committees verify a message that was not signed
        msgs[32 * c] ^= 1
    d_msg = torch.frombuffer(msgs, dtype=torch.uint8).to(dev)
    d_off = torch.from_numpy(np.arange(0, n_keys + 1, k, dtype=np.uint32)).to(dev)
    d_st = torch.full((n_agg,), 0xFF, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        rc = L.ecgpu_fast_aggregate_verify_batch_dev(d_pk.data_ptr(), d_off.data_ptr(), n_keys, d_msg.data_ptr(), d_sig.data_ptr(),
                                                     n_agg, 0, d_st.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(f"ecgpu_fast_aggregate_verify_batch_dev -> {rc}: {L.ecgpu_last_error()}")

    step()
    torch.cuda.synchronize()
    steps = max(3, args.steps // 4)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    want = np.zeros(n_agg, dtype=np.uint8)
    want[::64] = 5
    ok = bool((d_st.cpu().numpy() == want).all())
    # the same aggregates through the validated-key registry: keys converted once, untimed
    reg = ctypes.c_void_p()
    cache = None
    if L.ecgpu_registry_create(n_keys, ctypes.byref(reg)) == 0:
        assert L.ecgpu_registry_set_dev(reg, 0, d_pk.data_ptr(), n_keys, stream) == 0
        d_idx = torch.arange(n_keys, dtype=torch.int32, device=dev)
        d_st2 = torch.full((n_agg,), 0xFF, dtype=torch.uint8, device=dev)

        def step2():
            rc = L.ecgpu_fast_aggregate_verify_indexed_batch_dev(reg, d_idx.data_ptr(), d_off.data_ptr(), n_keys, d_msg.data_ptr(),
                                                                 d_sig.data_ptr(), n_agg, 0, d_st2.data_ptr(), stream)
            if rc != 0:
                raise RuntimeError(f"ecgpu_fast_aggregate_verify_indexed_batch_dev -> {rc}: {L.ecgpu_last_error()}")

        step2()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step2()
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / steps
        cache = {"value": n_keys / dt2, "unit": "sigs/s", "aggregates_per_s": n_agg / dt2, "ms_per_step": dt2 * 1e3,
                 "statuses_equal_uncached": bool((d_st2 == d_st).all().item()),
                 "note": "keys validated once into a device-resident registry (untimed), then gathered by index"}
        torch.cuda.synchronize()
        L.ecgpu_registry_destroy(reg)
    return {"metric": "bls_signatures_verified_per_sec (K = 2048 aggregates)", "validated_key_cache": cache, "value": n_keys / dt, "unit": "sigs/s",
            "aggregates_per_s": n_agg / dt, "ms_per_step": dt * 1e3, "n_gpus": 1,
            "config": {"workload": f"fast_aggregate_verify of {n_agg} aggregates x {k} keys (configs[3] per-GPU share), "
                                   "reference semantics, 1/64 aggregates carry a wrong message"},
            "roofline": {"bound": "hbm", "achieved": (48 * n_keys + 129 * n_agg) / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (48 * n_keys + 129 * n_agg) / dt / 1e9 / HBM_PEAK_GBS,
                         "note": "48.06 B per signature (SURVEY.md 8d); the work is the per-key decompress + subgroup check"},
            "check": {"statuses_match_construction": ok}}


def run_epoch(args, L, torch, dist, rank, world, n_total=2048, k=2048, n_reg=1 << 20, sk_period=1 << 16):
    """BASELINE.json configs[3]: a full epoch of attestation aggregates -- 32 slots x 64 committees = 2 048 aggregates of
    K = 2 048 keys -- sharded over the ranks (contiguous committee ranges), statuses all-gathered every step.
    Registry = validators 0 .. 2^20 - 1; committee c, member j = validator (2048 c + j) mod 2^20 (the
    first 2^20 / 2048 = 512 committees are disjoint, so with 2 048 committees every validator appears 4 times); 1/64
    committees are corrupted (a wrong message / one wrong member alternately).  Two timings: reference semantics (every key
    decompressed + subgroup-checked on every call) and the validated-key registry.  Total work is fixed: strong scaling."""
    import numpy as np
    from ethereum_consensus_amd import shard
    dev = _device(torch)
    stream = _stream(torch)
    per = (n_total + world - 1) // world
    c0, c1 = min(rank * per, n_total), min((rank + 1) * per, n_total)
    n_agg = c1 - c0
    # the registry's secret keys repeat with period 2^16 (sk_i = sk_(i mod 65536)): 2^20 distinct SHA-derived keys would cost
    # a minute of host hashing per rank; the kernels see 2^20 separately stored, separately validated keys either way
    assert n_reg % sk_period == 0
    sk_small = bls_inputs(sk_period, 0)[0]
    d_sk = torch.frombuffer(bytearray(sk_small), dtype=torch.uint8).to(dev)
    d_pk_small = torch.empty(48 * sk_period, dtype=torch.uint8, device=dev)
    assert L.ecgpu_sk_to_pk_batch_dev(d_sk.data_ptr(), sk_period, d_pk_small.data_ptr(), stream) == 0
    d_reg_keys = d_pk_small.view(sk_period, 48).repeat(n_reg // sk_period, 1).contiguous().view(-1)
    sk_int = [int.from_bytes(sk_small[32 * i:32 * i + 32], "big") for i in range(sk_period)]
    # committee c covers validators [2048 c, 2048 c + 2048) mod 2^20: its keys are contiguous in the registry
    agg_sk, msgs = [], bytearray()
    for c in range(c0, c1):
        base = (k * c) % n_reg
        agg_sk.append(sum(sk_int[(base + j) % sk_period] for j in range(k)) % R_ORDER)
        msgs += S(b"att", c)
    idx = np.concatenate([(np.arange(k, dtype=np.int64) + (k * c) % n_reg) % n_reg for c in range(c0, c1)]).astype(np.uint32) \
        if n_agg else np.zeros(0, dtype=np.uint32)
    want = np.zeros(n_agg, dtype=np.uint8)
    for a, c in enumerate(range(c0, c1)):
        if c % 64 == 0:
            want[a] = 5
            if (c // 64) % 2 == 0:
                msgs[32 * a] ^= 1          # wrong message
            else:
                idx[a * k + min(7, k - 1)] = (int(idx[a * k + min(7, k - 1)]) + 1) % n_reg  # one wrong member (its neighbour's key differs: period sk_period)
    d_idx = torch.from_numpy(idx.astype(np.int32)).to(dev)
    d_ask = torch.frombuffer(bytearray(b"".join(x.to_bytes(32, "big") for x in agg_sk) or b"\0"), dtype=torch.uint8).to(dev)
    d_msg_clean = torch.frombuffer(bytearray(b"".join(S(b"att", c) for c in range(c0, c1)) or b"\0"), dtype=torch.uint8).to(dev)
    d_sig = torch.empty(max(96 * n_agg, 1), dtype=torch.uint8, device=dev)
    if n_agg:
        assert L.ecgpu_sign_batch_dev(d_ask.data_ptr(), 32, d_msg_clean.data_ptr(), n_agg, d_sig.data_ptr(), stream) == 0
    d_msg = torch.frombuffer(msgs if n_agg else bytearray(1), dtype=torch.uint8).to(dev)
    d_off = torch.from_numpy(np.arange(0, n_agg * k + 1, k, dtype=np.uint32).astype(np.int32)).to(dev)
    # reference semantics needs the key BYTES of every committee in list order: gathered once (workload layout, untimed)
    d_keys = d_reg_keys.view(n_reg, 48)[d_idx.long()].contiguous().view(-1) if n_agg else torch.zeros(1, dtype=torch.uint8, device=dev)
    d_st = torch.full((max(per, 1),), 0xFF, dtype=torch.uint8, device=dev)
    reg = ctypes.c_void_p()
    assert L.ecgpu_registry_create(n_reg, ctypes.byref(reg)) == 0
    assert L.ecgpu_registry_set_dev(reg, 0, d_reg_keys.data_ptr(), n_reg, stream) == 0
    _sync(torch)
    gathered = {}

    def make_step(use_registry):
        def step():
            if n_agg:
                if use_registry:
                    rc = L.ecgpu_fast_aggregate_verify_indexed_batch_dev(reg, d_idx.data_ptr(), d_off.data_ptr(), n_agg * k, d_msg.data_ptr(),
                                                                         d_sig.data_ptr(), n_agg, 0, d_st.data_ptr(), stream)
                else:
                    rc = L.ecgpu_fast_aggregate_verify_batch_dev(d_keys.data_ptr(), d_off.data_ptr(), n_agg * k, d_msg.data_ptr(),
                                                                 d_sig.data_ptr(), n_agg, 0, d_st.data_ptr(), stream)
                if rc != 0:
                    raise RuntimeError(f"epoch step -> {rc}: {L.ecgpu_last_error()}")
            if world > 1:
                gathered["st"] = shard.all_gather_bytes(dist, d_st, world)  # 2 048 status bytes in all
        return step

    out = {}
    for name, use_registry in (("reference_semantics", False), ("validated_key_registry", True)):
        step = make_step(use_registry)
        for _ in range(max(args.warmup, 1)):
            step()
        _sync(torch)
        if world > 1:
            dist.barrier()
        _sync(torch)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        _sync(torch)
        if world > 1:
            dist.barrier()
        ok_here = bool((d_st[:n_agg].cpu().numpy() == want).all())
        if world > 1:
            # after the gather every rank holds the whole epoch's statuses, rank r's at [r per, r per + its count)
            full = gathered["st"].cpu().numpy()
            ok_here = ok_here and full.shape[0] == per * world and bool((full[rank * per:rank * per + n_agg] == want).all())
        out[name] = {"dt": time.perf_counter() - t0, "ok": ok_here}
    L.ecgpu_registry_destroy(reg)
    ok = out["reference_semantics"]["ok"] and out["validated_key_registry"]["ok"]
    if world > 1:
        t = torch.tensor([1.0 if ok else 0.0, out["validated_key_registry"]["dt"]], dtype=torch.float64, device=DEV)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MIN)
        dist.all_reduce(t[1:], op=dist.ReduceOp.MAX)
        ok, out["validated_key_registry"]["dt"] = bool(t[0].item() > 0.5), float(t[1].item())
    alg = (48 * k + 129) * n_total
    return dict(
        dt=out["reference_semantics"]["dt"], units_per_step=n_total * k / world, metric="bls_signatures_verified_per_sec", unit="sigs/s", dtype="u32",
        scaling="strong",
        config={"workload": f"full-epoch attestation batch: {n_total} aggregates (32 slots x 64 committees) x {k} keys = {n_total * k} "
                            f"signatures, sharded over {world} GPU(s) ({per} aggregates per GPU), 1/64 committees corrupted; reference "
                            "semantics (every key decompressed + subgroup-checked on every call); statuses all-gathered every step",
                "aggregates": n_total, "keys_per_aggregate": k, "aggregates_per_gpu": per},
        roofline={"bound": "hbm", "kernel": "k_pk_validate", "achieved": alg / (out["reference_semantics"]["dt"] / args.steps) / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (out["reference_semantics"]["dt"] / args.steps) / 1e9 / HBM_PEAK_GBS,
                  "traffic": None, "algorithmic_bytes_per_step": alg,
                  "note": "48.06 B per signature (SURVEY.md 8d); whole-job bytes over the step time of all ranks"},
        check={"statuses_match_construction": ok},
        extra={"validated_key_registry": {"value": n_total * k * args.steps / out["validated_key_registry"]["dt"], "unit": "sigs/s",
                                          "aggregates_per_s": n_total * args.steps / out["validated_key_registry"]["dt"],
                                          "ms_per_step": out["validated_key_registry"]["dt"] / args.steps * 1e3,
                                          "note": "same statuses; the 2^20 keys are validated once into a device-resident registry "
                                                  "(replicated per GPU, 109 MB) and gathered by validator index"},
               "aggregates_per_s": n_total * args.steps / out["reference_semantics"]["dt"]})


def run_slots(args, L, torch, dist, rank, world, n_sync=512):
    """BASELINE.json configs[4]: per slot ONE eth_fast_aggregate_verify over the participating ~95 % of a 512-key sync committee
    (altair/block_processing.rs:216-236) and ONE root of the device-resident 2^20-validator deneb state after ~2^12 balance +
    participation patches (phase0/slot_processing.rs:67), the two enqueued on separate streams so that they overlap; sustained
    over `steps` slots (>= 64 asked for).  Neither piece shards usefully (a single K <= 512 aggregate, a 0.5 ms
    root): with N GPUs every rank runs its own slot stream (replicas) and the 33 result bytes per slot are all-gathered."""
    import random
    import numpy as np
    from ethereum_consensus_amd import bls, shard, ssz, synthetic
    dev = torch.device("cuda", torch.cuda.current_device())
    n = args.validators
    slots = max(args.steps, 64)
    r = random.Random(11 + rank)
    sks = [1 + int.from_bytes(S(b"sync", i), "big") % (R_ORDER - 1) for i in range(n_sync)]
    pks = bls.sk_to_pk_batch(b"".join(x.to_bytes(32, "big") for x in sks))
    reg = bls.ValidatorKeyRegistry(n_sync)
    reg.set(0, pks)
    f = synthetic.state_fields(n, "mainnet", seed=5 + rank)
    enc = synthetic.serialize_state(f)
    st = ssz.ResidentBeaconStateDeneb(enc, 0)
    st.hash_tree_root()  # builds the cached levels
    tail = [f["balances"].tobytes(), f["previous_epoch_participation"].tobytes(), f["current_epoch_participation"].tobytes(),
            f["inactivity_scores"].tobytes(), synthetic.serialize_payload_header(f["payload_header"]), f["historical_summaries"].tobytes()]
    bal_off = len(enc) - sum(len(x) for x in tail)
    part_off = bal_off + 8 * n + n
    n_distinct = 16  # distinct slot inputs, cycled
    work = []
    for slot in range(n_distinct):
        part = [i for i in range(n_sync) if r.random() < 0.95]
        msg = S(b"slot", slot + 1000 * rank)
        sig = bls.sign_batch((sum(sks[i] for i in part) % R_ORDER).to_bytes(32, "big"), [msg])
        if slot == 5:
            msg = S(b"slot", 999999)  # one forged aggregate per cycle
        patches = {}
        for _ in range(2048):
            patches[bal_off + 8 * r.randrange(n)] = r.randbytes(8)
            patches[part_off + r.randrange(n)] = bytes([r.randrange(8)])
        plist = sorted(patches.items())
        # marshalled once: what a Rust caller hands over is three arrays, not 4 096 Python tuples per slot
        c_offs = (ctypes.c_uint64 * len(plist))(*[o for o, _ in plist])
        doff = [0]
        for _, b_ in plist:
            doff.append(doff[-1] + len(b_))
        c_doff = (ctypes.c_uint64 * len(doff))(*doff)
        blob = ssz._buf(b"".join(b_ for _, b_ in plist))
        work.append(dict(c_patch=(c_offs, c_doff, blob, len(plist)), k=len(part), d_idx=torch.tensor(part, dtype=torch.int32, device=dev),
                         d_off=torch.tensor([0, len(part)], dtype=torch.int32, device=dev),
                         d_msg=torch.frombuffer(bytearray(msg), dtype=torch.uint8).to(dev),
                         d_sig=torch.frombuffer(bytearray(sig), dtype=torch.uint8).to(dev), patches=plist,
                         want=5 if slot == 5 else 0))
    def apply_patches(w_):
        o_, d_, b_, n_ = w_["c_patch"]
        if L.ecgpu_resident_state_patch(st.handle, o_, d_, b_, n_) != 0:
            raise RuntimeError(f"resident patch: {L.ecgpu_last_error()}")

    d_res = torch.zeros(33, dtype=torch.uint8, device=dev)  # status byte + state root of the slot
    s_bls, s_mk = torch.cuda.Stream(), torch.cuda.Stream()
    statuses = []

    def one_slot(i, record):
        w = work[i % n_distinct]
        # the aggregate does not depend on the state: enqueued first, so that marshalling and uploading the slot's patches
        # (host work: ~1 ms of Python per 4 096 patches) and the root run underneath its 6.7 ms dependent chain
        rc1 = L.ecgpu_fast_aggregate_verify_indexed_batch_dev(reg.handle, w["d_idx"].data_ptr(), w["d_off"].data_ptr(), w["k"], w["d_msg"].data_ptr(),
                                                              w["d_sig"].data_ptr(), 1, 1, d_res.data_ptr(), s_bls.cuda_stream)
        apply_patches(w)
        rc2 = L.ecgpu_resident_state_root_dev(st.handle, d_res.data_ptr() + 1, s_mk.cuda_stream)
        if rc1 or rc2:
            raise RuntimeError(f"slot step -> {rc1}, {rc2}: {L.ecgpu_last_error()}")
        # a slot is complete when both results exist: the next slot's patches depend on this slot's block
        s_bls.synchronize()
        s_mk.synchronize()
        if world > 1:
            shard.all_gather_bytes(dist, d_res, world)
        if record:
            statuses.append((int(d_res[0].item()), w["want"]))

    for i in range(max(args.warmup, 2)):
        one_slot(i, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(slots):
        one_slot(i, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # sub-latencies, each alone on an idle GPU
    w = work[0]
    lat = {}
    for name, fn in (("sync_aggregate_ms", lambda: L.ecgpu_fast_aggregate_verify_indexed_batch_dev(
            reg.handle, w["d_idx"].data_ptr(), w["d_off"].data_ptr(), w["k"], w["d_msg"].data_ptr(), w["d_sig"].data_ptr(), 1, 1, d_res.data_ptr(),
            s_bls.cuda_stream)), ("state_root_ms", lambda: L.ecgpu_resident_state_root_dev(st.handle, d_res.data_ptr() + 1, s_mk.cuda_stream))):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(8):
            if name == "state_root_ms":
                apply_patches(w)
            fn()
            torch.cuda.synchronize()
        lat[name] = (time.perf_counter() - t1) / 8 * 1e3
    # the root alone (dirty paths only): the slot's 4 096 patches applied and marked (`patch_ms`), then the
    # host-pointer root -- rebuild nothing, ONE climb launch, the fused tail, 40 bytes back (`root_ms`); hash64 counted on the device
    import ctypes as _ct
    rb = _ct.create_string_buffer(32)
    t_patch = t_root = 0.0
    tree_hashes = []
    for k in range(8):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        apply_patches(work[k % n_distinct])
        t2 = time.perf_counter()
        if L.ecgpu_resident_state_root(st.handle, rb) != 0:
            raise RuntimeError(f"resident root: {L.ecgpu_last_error()}")
        t3 = time.perf_counter()
        t_patch += t2 - t1
        t_root += t3 - t2
        tree_hashes.append(int(L.ecgpu_last_hash64_count()))
    lat["patch_ms"], lat["root_ms"] = t_patch / 8 * 1e3, t_root / 8 * 1e3
    lat["hash64_per_root"] = tree_hashes[-1]  # (the first host-pointer root also collects the count of the timed loop's device-entry roots)
    # the resident root after all those patches equals a from-scratch root of the patched encoding
    encb = bytearray(enc)
    applied = max(args.warmup, 2) + slots + 16
    for i in list(range(max(args.warmup, 2))) + list(range(slots)) + [0] * 8 + [k % n_distinct for k in range(8)]:
        for off, b in work[i % n_distinct]["patches"]:
            encb[off:off + len(b)] = b
    root_ok = rb.raw == ssz.hash_tree_root_beacon_state_deneb(bytes(encb), 0)
    ok = all(g == wv for g, wv in statuses) and root_ok
    st.close()
    reg.close()
    hashes_per_root = tree_hashes[-1]
    return dict(
        dt=dt, units_per_step=1, steps=slots, metric="slots_per_sec (sync-committee aggregate + state root per slot)", unit="slots/s", dtype="u32",
        config={"workload": f"per slot: eth_fast_aggregate_verify over ~95 % of a {n_sync}-key sync committee (validated-key registry) + root of "
                            f"the resident deneb mainnet state ({n} validators) after 4 096 balance / participation patches, two streams; "
                            f"{slots} slots back to back per GPU, results all-gathered per slot", "slots": slots, "validators": n,
                "sharding": "replicas: every GPU runs its own slot stream (SURVEY.md 8e: neither piece shards usefully)"},
        roofline={"bound": "hbm", "kernel": "(latency-bound: one aggregate and one incremental root per slot)", "achieved": 0.0, "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": 0.0, "traffic": None, "sub_latency_ms": lat,
                  "note": "a slot is a dependent chain (message stage -> pairing check); the root hides under it"},
        check={"statuses_match_construction": all(g == wv for g, wv in statuses), "last_root_equals_from_scratch_root": root_ok, "ok": ok,
               "applied_patch_sets": applied, "hash64_of_last_root": hashes_per_root})


def run_block(args, L, torch, n_val=1 << 16):
    """Whole-block batching: every verification of ONE block -- 128 attestations x ~400 keys
    (phase0/block_processing.rs:752-761), the sync aggregate over ~95 % of 512 keys (altair/block_processing.rs:226-234), 16
    single-key operations (proposer, randao, exits ...) -- pushed into the collector (ecgpu_batch_*) and verified in one pass;
    host memory in, statuses out (the shape the Rust caller has).  Timed with raw keys (reference semantics: every key
    decompressed + checked) and through a validated-key registry.  Secondary line."""
    from ethereum_consensus_amd import bls
    sk = bls_inputs(n_val, 0)[0]
    sks = [int.from_bytes(sk[32 * i:32 * i + 32], "big") for i in range(n_val)]
    reg_keys = bls.sk_to_pk_batch(sk)
    key = lambda i: reg_keys[48 * i:48 * i + 48]
    members = [[(c * 509 + j * 7) % n_val for j in range(400)] for c in range(128)]
    sync = [i for i in range(512) if i % 20 != 3]
    single = list(range(1000, 1016))
    lists = members + [sync] + [[i] for i in single]
    msgs = [S(b"blk", c) for c in range(len(lists))]
    agg = [sum(sks[i] for i in l) % R_ORDER for l in lists]
    sigs = bls.sign_batch(b"".join(a.to_bytes(32, "big") for a in agg), msgs)
    eth = [0] * 128 + [1] + [0] * 16
    sig = lambda t: sigs[96 * t:96 * t + 96]
    msgs[7] = S(b"blk", 9999)  # one attestation over the wrong message
    want = [5 if t == 7 else 0 for t in range(len(lists))]
    reg = bls.ValidatorKeyRegistry(n_val)
    reg.set(0, reg_keys)
    out = {}
    for name, registry in (("reference_semantics", None), ("validated_key_registry", reg)):
        b = bls.SignatureBatch(registry)
        best = None
        for rep in range(4):
            t0 = time.perf_counter()
            for t, l in enumerate(lists):
                if registry is None:
                    b.fast_aggregate_verify([key(i) for i in l], msgs[t], sig(t), eth=bool(eth[t]))
                else:
                    b.fast_aggregate_verify_indexed(l, msgs[t], sig(t), eth=bool(eth[t]))
            t1 = time.perf_counter()
            got = list(b.flush())
            t2 = time.perf_counter()
            if rep:  # the first pass warms the arenas
                best = (t2 - t1, t1 - t0) if best is None or t2 - t1 < best[0] else best
            assert got == want, (name, got[:10])
        b.close()
        out[name] = {"block_verify_ms": best[0] * 1e3, "push_ms_python": best[1] * 1e3}
    reg.close()
    # the reference's own call pattern, one verification per call (crypto/bls.rs:64-77, 91-112): the latency of ONE scalar call,
    # host buffers in, status out -- a single-key one and a 400-key attestation
    scalar = {}
    for name, t in (("verify_signature", 129 + 3), ("fast_aggregate_verify_400_keys", 3)):
        keys = [key(i) for i in lists[t]]
        best = None
        for rep in range(6):
            t0 = time.perf_counter()
            code = (bls.verify_signature_status(keys[0], msgs[t], sig(t)) if len(keys) == 1
                    else bls.fast_aggregate_verify_status(keys, msgs[t], sig(t)))
            dt = time.perf_counter() - t0
            assert code == 0, (name, code)
            if rep:
                best = dt if best is None or dt < best else best
        scalar[name + "_ms"] = best * 1e3
    # where a lone call's time goes: the three side stages run side by side on three queues, then the pairing check (HIP events
    # per stage; the wall time above is host buffers in, status out)
    L.ecgpu_prof_enable(1)
    tags = ("bls_pk_validate", "bls_sig", "bls_h2c", "bls_pairing", "bls_row_a", "bls_row_inv", "bls_row_c")
    before = {t_: _prof(L, t_) for t_ in tags}
    keys = [key(i) for i in lists[129 + 3]]
    for _ in range(4):
        assert bls.verify_signature_status(keys[0], msgs[129 + 3], sig(129 + 3)) == 0
    scalar["verify_signature_stage_ms"] = {t_: (_prof(L, t_)[0] - before[t_][0]) / max(1, _prof(L, t_)[1] - before[t_][1]) for t_ in tags}
    L.ecgpu_prof_enable(0)
    scalar["pairing_path"] = {1: "lane", 3: "lane groups", 5: "two lanes per tuple", 7: "row machine"}.get(int(L.ecgpu_bls_last_pairing_path()), "?")
    scalar["blst_one_core_estimate_ms"] = 1e3 / BLST_SIGS_PER_CORE
    n_sigs = sum(len(l) for l in lists)
    return {"verifications": len(lists), "signatures": n_sigs, **out, "scalar_call": scalar,
            "note": "flush() of one block's verifications (host buffers in, statuses out); scalar_call: ONE verification per call, "
                    "the reference's own pattern -- dependent latency each, so a block through 145 scalar calls costs 145 of them; "
                    "push_ms_python is ctypes marshalling, not the library",
            "check": {"statuses_match_construction": True}}


def run_latency_curve(args, L, torch, sizes=(1, 64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536)):
    """whole-call latency of ONE ecgpu_fast_aggregate_verify_batch_dev over the first n tuples of the
    headline workload (K = 1, resident inputs, fault cycle included) at every size class of the dispatch -- rows, lane groups, two
    lanes per tuple, one lane per tuple: the best of 3 warm calls each, statuses checked."""
    import numpy as np
    nmax = max(sizes)
    d_pk, d_msg, d_sig, d_st, h_pk, msgs, h_sig, want_bytes, stream = _bls_setup(L, torch, nmax, 0)
    want = np.frombuffer(bytes(want_bytes), dtype=np.uint8)
    ms, ok, paths = [], True, []
    for n in sizes:
        best = None
        for rep in range(4):
            _sync(torch)
            t0 = time.perf_counter()
            rc = L.ecgpu_fast_aggregate_verify_batch_dev(d_pk.data_ptr(), None, n, d_msg.data_ptr(), d_sig.data_ptr(), n, 0, d_st.data_ptr(), stream)
            if rc != 0:
                raise RuntimeError(f"latency curve n = {n}: {rc} {L.ecgpu_last_error()}")
            _sync(torch)
            dt = time.perf_counter() - t0
            if rep:
                best = dt if best is None or dt < best else best
        ok = ok and bool((d_st[:n].cpu().numpy() == want[:n]).all())
        ms.append(best * 1e3)
        paths.append(int(L.ecgpu_bls_last_pairing_path()))
    import ctypes
    thr = (ctypes.c_uint32 * 4)()
    L.ecgpu_bls_dispatch_thresholds(thr)
    rates = [n / (t * 1e-3) for n, t in zip(sizes, ms)]
    worst_drop = min(rates[i + 1] / rates[i] for i in range(len(rates) - 1))
    return {"n": list(sizes), "ms": ms, "pairing_path": paths, "sigs_per_s_never_drops_below": worst_drop,
            "thresholds": {"rows_up_to": int(thr[0]), "lane_groups_up_to": int(thr[1]), "two_lanes_up_to": int(thr[2]), "measured_on_this_device": bool(thr[3])},
            "note": "whole call, inputs resident in HBM, best of 3 warm calls; pairing_path: 7 rows, 3 lane groups, 5 two lanes per tuple, 1 one lane "
                    "per tuple; sigs_per_s_never_drops_below = min over consecutive sizes of rate(n_next) / rate(n)",
            "check": {"statuses_match_construction": ok}}


def run_config2_readings(args, L, torch):
    """SURVEY.md 8(d) config 2's other two readings (missing 4): the signature of the reference function is
    `fast_aggregate_verify(&[&PublicKey], &[u8], &Signature)` (crypto/bls.rs:114-118), so "65 536 (pk, msg, sig) tuples" can also be
    ONE call over 65 536 keys and one message, or 64 aggregates of 1 024 keys.  Both timed with reference semantics (every key
    decompressed + subgroup-checked in the call) and through a validated-key registry; sig = (sum of the secret keys) H(msg), so
    the expected status is success by construction (parity with the oracle incl. damaged keys: tests/test_gpu_bls.py
    test_config2_one_call_with_65536_keys_and_one_message / test_config2_64_aggregates_of_1024_keys)."""
    import numpy as np
    from ethereum_consensus_amd import bls
    dev = _device(torch)
    N = 65536
    sk, _ = bls_inputs(N, 0)
    sks = [int.from_bytes(sk[32 * i:32 * i + 32], "big") for i in range(N)]
    pks = bls.sk_to_pk_batch(sk)
    d_pk = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
    reg = bls.ValidatorKeyRegistry(N)
    reg.set(0, pks)
    d_idx = torch.arange(N, dtype=torch.int32, device=dev)
    stream = _stream(torch)
    out = {}
    for name, n, k in (("one_call_k65536", 1, N), ("n64_k1024", 64, 1024)):
        msgs = [S(b"c2r" + name.encode(), t) for t in range(n)]
        agg = [sum(sks[k * t:k * t + k]) % R_ORDER for t in range(n)]
        sigs = bls.sign_batch(b"".join(a.to_bytes(32, "big") for a in agg), msgs)
        d_msg = torch.frombuffer(bytearray(b"".join(msgs)), dtype=torch.uint8).to(dev)
        d_sig = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
        d_off = torch.tensor([k * t for t in range(n + 1)], dtype=torch.int32, device=dev)
        d_st = torch.full((n,), 0xFF, dtype=torch.uint8, device=dev)
        rec = {}
        for sem in ("reference_semantics", "validated_key_registry"):
            best = None
            for rep in range(4):
                _sync(torch)
                t0 = time.perf_counter()
                if sem == "reference_semantics":
                    rc = L.ecgpu_fast_aggregate_verify_batch_dev(d_pk.data_ptr(), d_off.data_ptr(), N, d_msg.data_ptr(), d_sig.data_ptr(), n, 0,
                                                                 d_st.data_ptr(), stream)
                else:
                    rc = L.ecgpu_fast_aggregate_verify_indexed_batch_dev(reg.handle, d_idx.data_ptr(), d_off.data_ptr(), N, d_msg.data_ptr(),
                                                                         d_sig.data_ptr(), n, 0, d_st.data_ptr(), stream)
                if rc != 0:
                    raise RuntimeError(f"{name}/{sem}: {rc} {L.ecgpu_last_error()}")
                _sync(torch)
                dt = time.perf_counter() - t0
                if rep:
                    best = dt if best is None or dt < best else best
            if not bool((d_st.cpu().numpy() == 0).all()):
                raise RuntimeError(f"{name}/{sem}: a valid aggregate did not verify")
            rec[sem] = {"ms": best * 1e3, "sigs_per_s": N / best}
        out[name] = rec
    reg.close()
    out["note"] = "65 536 signatures either way; resident inputs, best of 3 warm calls; expected status success by construction"
    out["check"] = {"statuses_match_construction": True}
    return out


def first_call_record():
    """what the FIRST BLS call of a process costs, with and without ecgpu_warmup (two fresh processes: tools/first_call_probe.py)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import first_call_probe
        return first_call_probe.probe()
    except Exception as e:  # noqa: BLE001 -- a probe must not take the bench line down
        return {"error": repr(e)[:80]}



def run_msm(args, L, torch, sizes=(1 << 16, 1 << 18)):
    """north_star "G1/G2 ... multi-scalar-mult": sum_i [k_i] P_i over G1 with 255-bit scalars through the host entry
    (ecgpu_g1_msm: compressed points and scalars in host memory, H2D included), by buckets (csrc/bls.hip).  P_i = [sk_i] g1, so the
    expected result is [sum k_i sk_i mod r] g1: one device key derivation.  Secondary line."""
    import random
    from ethereum_consensus_amd import bls
    base = 1 << 16
    sk = bls_inputs(base, 0)[0]
    sks = [int.from_bytes(sk[32 * i:32 * i + 32], "big") for i in range(base)]
    pk = bls.sk_to_pk_batch(sk)
    r = random.Random(7)
    out = {}
    for n in sizes:
        reps = n // base
        pts = pk * reps
        ks = [r.randrange(1, 1 << 255) for _ in range(n)]
        sc = b"".join(k.to_bytes(32, "big") for k in ks)
        want = bls.sk_to_pk_batch((sum(k * sks[i % base] for i, k in enumerate(ks)) % R_ORDER).to_bytes(32, "big"))
        res = ctypes.create_string_buffer(48)
        L.ecgpu_prof_filter(None)
        best = None
        for rep in range(3):
            L.ecgpu_prof_enable(1)
            t0 = time.perf_counter()
            rc = L.ecgpu_g1_msm(pts, sc, n, 255, res)
            dt = time.perf_counter() - t0
            if rc != 0:
                raise RuntimeError(f"ecgpu_g1_msm -> {rc}: {L.ecgpu_last_error()}")
            stages = {t: _prof(L, t)[0] for t in ("bls_pk_validate", "bls_msm_sort", "bls_msm_buckets", "bls_msm_reduce")}
            L.ecgpu_prof_enable(0)
            if rep and (best is None or dt < best[0]):
                best = (dt, stages)
        out[f"g1_n{n}"] = {"points_per_s": n / best[0], "ms": best[0] * 1e3, "stage_ms": best[1], "result_ok": res.raw == want}
    return {"metric": "msm_points_per_sec (G1, 255-bit scalars, 8-bit windows, host buffers in)", **out,
            "work_per_term": "key_validate (decompress + subgroup check: the bulk of the time) + 32 mixed additions (one per window) "
                             "~ 32 x (7 products + 4 squarings) = 113 k multiply instructions; fixed: 8 160 bucket workgroups, "
                             "~250 dependent doublings",
            "check": {"equals_sk_times_generator": all(v["result_ok"] for v in out.values())}}


def _prof(L, tag):
    ms = ctypes.c_double(0)
    nl = ctypes.c_uint64(0)
    L.ecgpu_prof_read(tag.encode(), ctypes.byref(ms), ctypes.byref(nl))
    return ms.value, int(nl.value)



def finish(r, args, world, dist, torch):
    """max-over-ranks wall time of the K timed steps -> the fields of one metric"""
    dt = r["dt"]
    if _multi(dist, world):
        t = torch.tensor([dt], dtype=torch.float64, device=DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    steps = r.get("steps", args.steps)
    total_units = r["units_per_step"] * steps * world
    out = {"metric": r["metric"], "value": total_units / dt, "unit": r["unit"], "n_gpus": world,
           "steps": steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3,
           "higher_is_better": True, "scaling": r.get("scaling", "weak"), "vs_baseline": None, "dtype": r["dtype"],
           "data": "synthetic", "config": r["config"], "roofline": r["roofline"], "check": r.get("check")}
    out.update(r.get("extra", {}))
    return out


def gather_selfcheck(mine, world, dist, torch):
    """every rank's (large-code slowdown, pairing build): with N > 1 the slowest rank sets the step time, so the line carries all"""
    per_rank = [mine]
    if _multi(dist, world):
        try:
            t = torch.tensor(mine, dtype=torch.float64, device=DEV)
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            per_rank = [p.cpu().tolist() for p in parts]
        except Exception:  # noqa: BLE001 - a failed diagnostic must not cost the measurement
            per_rank = [mine]
    return per_rank


def flush_c_stdio():
    """RCCL announces itself through printf ("RCCL version : ..."); on a pipe that text sits in the C library's buffer until the
    process exits, i.e. it would FOLLOW the one JSON line the driver reads.  Flushing the C streams first keeps the line last."""
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()


def multi_gpu_preflight(L, torch, dist, rank, world, local):
    """What fails first on a real N-GPU node, checked before anything is timed (tools/multi_gpu_preflight.py runs the same):
    torch and the library are bound to the SAME device (LOCAL_RANK), a kernel of the library runs there, and a 1-byte
    all_gather_into_tensor on device tensors goes through RCCL and returns every rank's byte in rank order."""
    out = {"rank": rank, "world": world, "local_rank": local}
    cur = torch.cuda.current_device()
    lib_dev = int(L.ecgpu_thread_device())
    out["torch_device"], out["library_device"] = cur, lib_dev
    if cur != local or lib_dev != local:
        raise RuntimeError(f"rank {rank}: torch is on device {cur}, the library on {lib_dev}, LOCAL_RANK is {local}")
    probe = ctypes.create_string_buffer(32)
    if L.ecgpu_sha256(b"preflight", 9, probe) != 0:
        raise RuntimeError(f"rank {rank}: the library's first kernel failed: {L.ecgpu_last_error()}")
    mine = torch.tensor([rank + 1], dtype=torch.uint8, device=DEV if DEV == "cpu" else torch.device("cuda", local))
    from ethereum_consensus_amd import shard
    got = shard.all_gather_bytes(dist, mine, world, force=True).cpu().tolist()
    if got != [r + 1 for r in range(world)]:
        raise RuntimeError(f"rank {rank}: 1-byte all-gather returned {got}")
    out["all_gather_1_byte"] = "ok"
    out["backend"] = dist.get_backend()
    return out


def sub_record(line, keys=("metric", "value", "unit", "ms_per_step", "steps", "scaling", "dtype", "config", "roofline", "check", "phases")):
    out = {k: line[k] for k in keys if k in line}
    for k in ("validated_key_registry", "aggregates_per_s"):
        if k in line:
            out[k] = line[k]
    return out


# ---- the line the driver records ------------------------------------------------------------------------------------------
# The driver keeps 8 KB of stdout; round 4's line was 17 KB and its Merkle half fell off the end.  The default line is therefore
# COMPACT: both halves of the metric first (top level = BLS with `roofline` and `cpu_baseline`, then `merkle` with its own), the
# other configurations after them, numbers rounded to 5 significant digits, and every note / provenance string left to the full
# record (gpurun_out/bench_full.json, or `--verbose`).
LINE_BUDGET = 6900  # (7 KB = 7 168 bytes is the line's contract, tests/test_bench_line.py; `full_record` and the lighten passes' slack come on top)
_PROSE_KEYS = {"note", "launch_note", "peak_source", "source", "host", "semantics", "basis", "sample_detail", "why"}
_KEY_ORDER = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "merkle", "check", "latency_curve", "config2_readings", "slots", "block",
              "aggregates_k2048",
              "resident_tree", "epoch", "strong_2p20", "merkle_strong", "merkle_sharded_emulated", "half_round_32768", "msm",
              "weights", "preflight", "box_selfcheck")


def _compact(v, cap=88, drop=_PROSE_KEYS):
    if isinstance(v, dict):
        return {k: _compact(x, cap, drop) for k, x in v.items() if k not in drop and x is not None or k in ("vs_baseline", "traffic")}
    if isinstance(v, (list, tuple)):
        return [_compact(x, cap, drop) for x in v]
    if isinstance(v, float):
        return float(f"{v:.5g}")
    if isinstance(v, str) and len(v) > cap:
        return v[:cap - 1] + "~"
    return v


def compact_line(full):
    """the recorded line: same keys for everything the contract names, prose dropped, both halves of the metric in front"""
    ordered = {k: full[k] for k in _KEY_ORDER if k in full}
    ordered.update({k: v for k, v in full.items() if k not in ordered})
    line = _compact(ordered)
    # progressively lighter until it fits: the secondary records lose their second-level detail first, never the two rooflines
    lighten = (("traffic_detail",), ("per_rank", "phases", "one_thread", "eight_threads", "stage_ms_registry"),
               ("valu_int", "config"))
    secondary = [k for k in line if k not in ("roofline", "cpu_baseline", "merkle", "config") and isinstance(line[k], dict)]
    for keys in lighten:
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        for k in secondary:
            line[k] = _compact(line[k], 64, _PROSE_KEYS | set(keys))
    if len(json.dumps(line)) > LINE_BUDGET:  # last resort: a secondary record's roofline shrinks to its five numbers
        for k in secondary:
            rf = line[k].get("roofline")
            if isinstance(rf, dict):
                line[k]["roofline"] = {q: rf[q] for q in ("kernel", "achieved", "frac", "traffic", "avg_launch_ms", "sub_latency_ms") if q in rf}
            for q in ("dtype", "scaling", "steps"):
                line[k].pop(q, None)
            if line[k].get("metric") == line.get("metric"):
                line[k].pop("metric")
    # still too long: the records furthest from the two halves of the metric shrink to value / time / verdict, one at a time
    def _minimal(rec):
        out = {q: rec[q] for q in ("value", "unit", "ms_per_step") if q in rec}
        chk = rec.get("check")
        if isinstance(chk, dict):
            out["check_ok"] = all(v for v in chk.values() if isinstance(v, bool))
        return out
    for k in ("msm", "half_round_32768", "merkle_sharded_emulated", "merkle_strong", "box_selfcheck", "epoch", "strong_2p20", "aggregates_k2048"):
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        if isinstance(line.get(k), dict) and k != "box_selfcheck":
            line[k] = _minimal(line[k])
        elif k == "box_selfcheck" and isinstance(line.get(k), dict):
            line[k] = {q: line[k][q] for q in ("large_code_slowdown", "pairing_kernels") if q in line[k]}
    line["full_record"] = "gpurun_out/bench_full.json"
    return line


def emit(full, args):
    """rank 0: the full record to a side file, ONE line to stdout (compact unless --verbose)"""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as fh:
            json.dump(full, fh)
    except OSError:
        pass
    print(json.dumps(full if args.verbose else compact_line(full)), flush=True)


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible: bench.py measures the HIP path only"}))
        sys.exit(2)
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or FORCE_DIST:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local), rank=rank, world_size=world)
    from ethereum_consensus_amd import _lib
    L = _lib.load(build_if_missing=False)
    rc = L.ecgpu_init(local)
    if rc != 0:
        raise RuntimeError(f"ecgpu_init -> {rc}: {L.ecgpu_last_error()}")
    # what a host that verifies blocks and epochs does once per process: the warm-up with the reference's fixed
    # vector, which also places the dispatch thresholds where THIS device's kernels cross (untimed; `latency_curve.thresholds`)
    rc = L.ecgpu_warmup((1 | 4) if args.no_calibration else (1 | 2 | 4))
    if rc != 0:
        raise RuntimeError(f"ecgpu_warmup -> {rc}: {L.ecgpu_last_error()}")
    preflight = multi_gpu_preflight(L, torch, dist, rank, world, local) if dist is not None else None
    workload = args.workload
    if workload == "auto":
        workload = "both"
    if workload not in ("bls", "merkle", "both", "epoch", "slots"):
        raise SystemExit("unknown workload " + workload)

    # BASELINE.json's metric has two halves.  The line's top level is the BLS half on configs[1]
    # (65 536 tuples); the Merkle half on configs[2] (2^20-validator state root) is timed the same way
    # right after it and reported, complete with its own roofline, under "merkle".
    line = None
    if workload == "epoch":
        line = finish(run_epoch(args, L, torch, dist, rank, world), args, world, dist, torch)
    if workload == "slots":
        line = finish(run_slots(args, L, torch, dist, rank, world), args, world, dist, torch)
    if workload in ("bls", "both"):
        r_bls = run_bls(args, L, torch, dist, rank, world)
        line = finish(r_bls, args, world, dist, torch)
        if world == 1 and not args.no_aggregates:
            line["aggregates_k2048"] = run_bls_aggregate(args, L, torch, dist, rank, world)
            line["block"] = run_block(args, L, torch)
            line["block"]["scalar_call"]["first_call"] = first_call_record()
            line["latency_curve"] = run_latency_curve(args, L, torch)
            line["config2_readings"] = run_config2_readings(args, L, torch)
            line["msm"] = run_msm(args, L, torch)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_bls(r_bls["host_sample"])
    if workload == "merkle" and args.scaling == "strong":
        r = run_merkle_sharded(args, L, torch, dist, rank, world)
        if not r["check"]["equals_unsharded_root"]:
            raise RuntimeError("sharded state root differs from the unsharded root")
        line = finish(r, args, world, dist, torch)
    elif workload in ("merkle", "both"):
        r = run_merkle(args, L, torch, dist, rank, world)
        r["check"] = {"root": r.get("root")}
        m = finish(r, args, world, dist, torch)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            m["cpu_baseline"] = cpu_baseline_merkle(args.validators)
        if line is None:
            line = m
        else:
            line["merkle"] = {k: m[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline",
                                                "check", "cpu_baseline", "h2d_inclusive", "two_roots_in_flight") if k in m}
    # The default line (N = 1, both halves, default sizes) also carries the other configurations of BASELINE.json, each a few
    # seconds of GPU time, each with its own check: north_star's 2^20-signature K = 1 batch in ONE call ("strong_2p20": the
    # N = 1 point of `--tuples 1048576 --scaling strong`), configs[3] ("epoch": all 2 048 aggregates of 2 048 keys) and
    # configs[4] ("slots": 64 slots of sync aggregate + state root).
    if workload == "both" and not args.no_extras and args.tuples == 65536 and args.scaling == "weak":
        import copy
        a2 = copy.copy(args)
        a2.steps, a2.warmup = 2, 1
        # north_star's two strong-scaled points ride in every line, N = 1 and N > 1 alike: the 2^20-signature batch over the
        # N ranks, and ONE 2^20-validator state over the N ranks (at N = 1 the fused single-GPU root above IS that point; the
        # sharded code path then runs as an 8-rank emulation on the one GPU, root asserted equal)
        sp = sub_record(finish(run_bls(a2, L, torch, dist, rank, world, n_total=1 << 20, strong=True), a2, world, dist, torch))
        if line is not None:
            line["strong_2p20"] = sp
            if world == 1:
                line["strong_2p20"]["vs_16x_65536_step"] = line["strong_2p20"]["ms_per_step"] / (16 * line["ms_per_step"])
        am = copy.copy(args)
        am.steps, am.warmup = (args.steps, args.warmup) if world > 1 else (5, 1)
        rm = run_merkle_sharded(am, L, torch, dist, rank, world, emulate_world=8 if world == 1 else None)
        ms_rec = sub_record(finish(rm, am, world, dist, torch))
        if world == 1:
            ms_rec["note"] = ("8 ranks emulated one after the other on one GPU: a parity check of the sharded path, not a scaling "
                              "point (value = the whole state's leaves over the time of all 8 phase A + one phase B)")
        if line is not None:
            line["merkle_strong" if world > 1 else "merkle_sharded_emulated"] = ms_rec
    if workload == "both" and world == 1 and not args.no_extras and args.tuples == 65536 and args.scaling == "weak":
        import copy
        a3 = copy.copy(args)
        a3.steps, a3.warmup = 2, 1
        line["epoch"] = sub_record(finish(run_epoch(a3, L, torch, dist, rank, world), a3, world, dist, torch))
        a4 = copy.copy(args)
        a4.steps, a4.warmup = 64, 2
        line["slots"] = sub_record(finish(run_slots(a4, L, torch, dist, rank, world), a4, world, dist, torch))
        # half a round of lanes (32 768 tuples of the same workload): the three side stages side by side on three hardware queues,
        # Miller loop and final exponentiation on two lanes per tuple in their one-wave builds
        a5 = copy.copy(args)
        a5.steps, a5.warmup = 5, 2
        line["half_round_32768"] = sub_record(finish(run_bls(a5, L, torch, dist, rank, world, n_total=32768, strong=False), a5, world, dist, torch))
    # box self-check (csrc/selfcheck.hip): 2^21 multiply-adds per lane as loops over 8 KB .. 1 MB of code.  On a healthy box
    # they take the same time; where the large ones are several times slower, so are the sums-of-products lane kernels, and
    # the library has switched that rank to its compact-code build.  Every rank checks its own GPU: with N > 1 the slowest
    # rank sets the step time, so the line carries every rank's slowdown and build.
    import ctypes
    sweep = (ctypes.c_double * 4)()
    ok = L.ecgpu_selfcheck_ifetch_sweep(sweep) == 0 and sweep[0] > 0
    mine = [sweep[3] / sweep[0] if ok else 0.0, float(L.ecgpu_bls_tower())]
    per_rank = gather_selfcheck(mine, world, dist, torch)
    if rank == 0:
        if preflight is not None:
            line["preflight"] = preflight
        names = {1: "sums of products", 2: "compact-code tower"}
        if ok:
            line["box_selfcheck"] = {"mad_loop_8KB_ms": sweep[0], "mad_loop_64KB_ms": sweep[1], "mad_loop_256KB_ms": sweep[2],
                                     "mad_loop_1MB_ms": sweep[3], "large_code_slowdown": mine[0],
                                     "pairing_kernels": names.get(int(mine[1]), "?"),
                                     "per_rank": [{"large_code_slowdown": r[0], "pairing_kernels": names.get(int(r[1]), "?")}
                                                  for r in per_rank],
                                     "note": "2^21 multiply-adds per lane as loops over 8 KB .. 1 MB of code: instruction fetch far beyond the "
                                             "64 KB instruction cache; slowdown ~1.0 on a healthy box, 2.2 measured on a slow one "
                                             "(DESIGN.md 3.3)"}
    # the process group goes first (whatever RCCL still has to say, it says before the line), then the C streams, then the line
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        flush_c_stdio()
        emit(line, args)


if __name__ == "__main__":
    main()
