"""The C-ABI library loads, exports every symbol include/ecgpu.h declares, and has no CPU fallback."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ecgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ecgpu_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for want in ["ecgpu_verify", "ecgpu_fast_aggregate_verify", "ecgpu_aggregate_verify", "ecgpu_aggregate_sigs",
                 "ecgpu_aggregate_pks", "ecgpu_fast_aggregate_verify_batch", "ecgpu_merkleize", "ecgpu_htr_validators",
                 "ecgpu_htr_beacon_state_deneb", "ecgpu_sha256", "ecgpu_is_valid_merkle_branch"]:
        assert want in names


def test_library_exports_every_declared_symbol():
    from ethereum_consensus_amd import _lib
    L = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), f"libecgpu.so does not export {name}"
    assert b"gfx950" in L.ecgpu_version()


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ethereum_consensus_amd import _lib, ssz
    L = _lib.load()
    assert L.ecgpu_device_count() == 0
    assert L.ecgpu_init(-1) == -1  # ECGPU_ERR_NO_DEVICE
    with pytest.raises(_lib.EcgpuError):
        ssz.merkleize(b"\x01" * 64)
    with pytest.raises(_lib.EcgpuError):
        ssz.hash(b"abc")


def test_product_does_not_import_the_oracle():
    """oracle/ and tests/hostsim are checkers: nothing in the package may load them
    (build.py only knows how to *compile* them for the test-suite)."""
    pkg = os.path.join(ROOT, "ethereum_consensus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".h", ".hip", ".cpp")) or f == "build.py":
                continue
            text = open(os.path.join(dirpath, f), errors="ignore").read()
            for needle in ("import oracle", "from oracle", "liboracle", "libhostsim", "oracle/_build", "oracle.cref"):
                assert needle not in text, (f, needle)


def test_no_long_branch_clobbers_a_return_address():
    """Static guard against an LLVM AMDGPU branch-relaxation bug (csrc/common.h ECG_LONG_BRANCH_GUARD): in a device function
    without calls a branch longer than 128 KB is expanded through s[30:31] -- the function's own return address -- and the
    kernel never terminates.  Every device object of the product build is disassembled and checked."""
    import glob
    import sys
    from ethereum_consensus_amd import _lib
    _lib.load()  # builds the library if needed
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_census
    objs = sorted(glob.glob(os.path.join(ROOT, "ethereum_consensus_amd", "lib", "obj", "*.o")))
    if not objs:
        pytest.skip("no object files next to the library (built elsewhere)")
    bad = [(os.path.basename(o), fn, addr) for o in objs for fn, addr in isa_census.long_branch_clobbers(o)]
    assert not bad, bad


def _build_c_caller():
    from ethereum_consensus_amd import _lib
    _lib.load()  # builds the library if needed
    src = os.path.join(ROOT, "tests", "cabi", "abi_smoke.c")
    exe = os.path.join(ROOT, "tests", "cabi", "abi_smoke")
    libdir = os.path.dirname(_lib.LIB_PATH)
    import subprocess
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + libdir, "-lecgpu",
                    "-Wl,-rpath," + libdir], check=True)
    return exe


def test_plain_c_caller_without_a_gpu():
    """include/ecgpu.h compiles as C99 and a C program linked against libecgpu.so gets ECGPU_ERR_NO_DEVICE everywhere."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = _build_c_caller()
    r = subprocess.run([exe, "nogpu"], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_plain_c_caller_on_the_gpu():
    import subprocess
    from oracle import bls12_381 as B, ssz as ossz
    from tests import _blscases as C
    exe = _build_c_caller()
    root = ossz.BeaconBlockHeader.htr(ossz.BeaconBlockHeader.default()).hex()
    pk = B.sk_to_pk(C.CAN_SIGN_SK).hex()
    r = subprocess.run([exe, "gpu", root, pk, C.CAN_SIGN_SIG.hex(), C.CAN_SIGN_MSG.decode()], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


def test_ctypes_signatures_follow_the_header():
    """ethereum_consensus_amd/_lib.py states every entry's argument and return types by hand: a c_uint32 where the header says
    uint64_t still "works" for small values.  Every declared signature is compared with the header's prototype: arity, integer
    width and signedness, pointer-ness."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_rust_bindings import header_prototypes, c_type
    from ethereum_consensus_amd import _lib
    L = _lib.load()
    protos = header_prototypes()
    scalar = {"i32": (ctypes.c_int, ctypes.c_int32), "u32": (ctypes.c_uint, ctypes.c_uint32), "u64": (ctypes.c_uint64, ctypes.c_ulong),
              "i64": (ctypes.c_int64, ctypes.c_long), "usize": (ctypes.c_size_t,), "f64": (ctypes.c_double,)}

    def agrees(ct, want):
        base, depth, _ = want
        if depth > 0:
            return ct is not None and (ct in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(ct, "contents") or hasattr(ct, "_type_") and
                                       isinstance(getattr(ct, "_type_"), type))
        if base == "void":
            return ct is None
        return any(ct is c or (ct is not None and ctypes.sizeof(ct) == ctypes.sizeof(c) and ct._type_ == c._type_) for c in scalar[base])

    checked = 0
    for name, (c_ret, c_params) in protos.items():
        fn = getattr(L, name)
        if fn.argtypes is None:
            continue  # not bound from Python (the Rust-only or C-only entries)
        assert len(fn.argtypes) == len(c_params), (name, fn.argtypes, c_params)
        for ct, c in zip(fn.argtypes, c_params):
            assert agrees(ct, c_type(c)), (name, c, ct)
        assert agrees(fn.restype, c_type(c_ret, is_param=False)), (name, c_ret, fn.restype)
        checked += 1
    assert checked >= 80, checked
