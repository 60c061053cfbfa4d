"""The RCCL code path on a one-GPU box.  A single-rank `nccl` process group is legal, so the N-rank flow of bench.py --
process-group initialisation bound to LOCAL_RANK, the device-binding preflight, all_gather_into_tensor on device tensors,
the all-reduce of the step time, the strong-scaled K = 1 batch and ONE state sharded over the ranks -- executes on hardware in
every `pytest -m gpu` run instead of for the first time on an 8-GPU node.  Each case is a subprocess:
a process group belongs to a process."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    from ethereum_consensus_amd import _lib
    L = _lib.load(build_if_missing=False)
    assert L.ecgpu_init(0) == 0, "no gfx950 device: the GPU tests need one"
    return L


def _json_line(stdout):
    """the JSON line of the run (the last line that is one: RCCL's printf banner may trail it on some builds)"""
    for ln in reversed(stdout.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError("no JSON line in: " + stdout[-1500:])


def _env():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "ECGPU_BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    return env


@pytest.mark.gpu
def test_preflight_under_a_single_rank_nccl_group(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multi_gpu_preflight.py")], cwd=ROOT, env=_env(), capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["backend"] == "nccl" and line["torch_device"] == line["library_device"] == 0
    assert line["all_gather_1_byte"] == "ok" and line["sharded_state_root_equals_unsharded"] and line["strong_bls_statuses_match"]
    assert line["all_ranks_ok"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("argv,check", [
    (["--workload", "merkle", "--scaling", "strong", "--validators", "70001", "--steps", "2", "--warmup", "1"], "equals_unsharded_root"),
    (["--workload", "bls", "--scaling", "strong", "--tuples", "4097", "--steps", "1", "--warmup", "1", "--no-aggregates"],
     "statuses_match_construction"),
    # one state per rank, the roots all-gathered asynchronously every step (two buffers alternating)
    (["--workload", "merkle", "--validators", "70001", "--steps", "5", "--warmup", "2"], "root"),
])
def test_bench_strong_modes_through_rccl_with_one_rank(gpu, argv, check):
    """bench.py's own main(): nccl process group of one rank, collectives forced (ECGPU_BENCH_FORCE_DIST), preflight in the line"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + argv, cwd=ROOT, env=_env(),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 1
    if check == "root":
        assert line["scaling"] == "weak" and len(line["check"]["root"]) == 64 and line["h2d_inclusive"]["root_equals_resident_root"] is True
    else:
        assert line["scaling"] == "strong" and line["check"][check] is True, line["check"]
    assert line["preflight"]["backend"] == "nccl" and line["preflight"]["all_gather_1_byte"] == "ok"
