"""ctypes binding of libecgpu.so (include/ecgpu.h).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ECGPU_LIB: development override (kernel-variant experiments); the product path is the in-tree build
LIB_PATH = os.environ.get("ECGPU_LIB") or os.path.join(_HERE, "lib", "libecgpu.so")

u8p = ctypes.c_void_p
_lib = None


class EcgpuError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: ecgpu error {code}" + (f" ({detail})" if detail else ""))


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load the HIP library.  Raises if it is absent and cannot be built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise FileNotFoundError(LIB_PATH + " is missing: run `python -m ethereum_consensus_amd.build`")
        from . import build
        build.build_lib()
    # torch bundles its own libamdhip64.so.7; load it first so that the whole process shares ONE
    # HIP runtime (device pointers and streams are then interchangeable with torch's).
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for the library itself
        pass
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_u64, c_u32, c_size = ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_size_t
    sig = {
        "ecgpu_init": (c_int, [c_int]),
        "ecgpu_device_count": (c_int, []),
        "ecgpu_warmup": (c_int, [ctypes.c_uint]),
        "ecgpu_bind_thread": (c_int, [c_int]),
        "ecgpu_thread_device": (c_int, []),
        "ecgpu_fast_aggregate_verify_batch_multi": (c_int, [ctypes.c_void_p, c_u32, u8p, ctypes.c_void_p, u8p, u8p, c_u32, c_int, u8p]),
        "ecgpu_htr_validators_multi": (c_int, [ctypes.c_void_p, c_u32, u8p, c_u64, c_u64, u8p]),
        "ecgpu_version": (ctypes.c_char_p, []),
        "ecgpu_last_error": (ctypes.c_char_p, []),
        "ecgpu_sha256": (c_int, [u8p, c_size, u8p]),
        "ecgpu_sha256_batch": (c_int, [u8p, c_size, c_u64, u8p]),
        "ecgpu_merkleize": (c_int, [u8p, c_u64, c_u64, c_int, c_u64, u8p]),
        "ecgpu_merkleize_dev": (c_int, [u8p, c_u64, c_u64, c_int, c_u64, u8p, ctypes.c_void_p]),
        "ecgpu_htr_validators": (c_int, [u8p, c_u64, c_u64, u8p]),
        "ecgpu_htr_validators_dev": (c_int, [u8p, c_u64, c_u64, u8p, ctypes.c_void_p]),
        "ecgpu_validators_subtree_root": (c_int, [u8p, c_u64, c_u64, u8p]),
        "ecgpu_validators_subtree_root_dev": (c_int, [u8p, c_u64, c_u64, u8p, ctypes.c_void_p]),
        "ecgpu_merkleize_subtree_roots": (c_int, [u8p, c_u32, c_u64, c_u64, c_int, c_u64, u8p]),
        "ecgpu_merkleize_subtree_roots_dev": (c_int, [u8p, c_u32, c_u64, c_u64, c_int, c_u64, u8p, ctypes.c_void_p]),
        "ecgpu_htr_beacon_block_header": (c_int, [u8p, u8p]),
        "ecgpu_signing_root": (c_int, [u8p, u8p, u8p]),
        "ecgpu_is_valid_merkle_branch": (c_int, [u8p, u8p, c_u32, c_u64, u8p]),
        "ecgpu_htr_beacon_state_deneb": (c_int, [u8p, c_u64, c_int, u8p]),
        "ecgpu_htr_beacon_state_deneb_dev": (c_int, [u8p, c_u64, u8p, c_int, u8p, ctypes.c_void_p]),
        "ecgpu_beacon_state_deneb_fixed_size": (c_u64, [c_int]),
        "ecgpu_htr_beacon_state": (c_int, [c_int, u8p, c_u64, c_int, u8p]),
        "ecgpu_htr_beacon_state_dev": (c_int, [c_int, u8p, c_u64, u8p, c_int, u8p, ctypes.c_void_p]),
        "ecgpu_beacon_state_fixed_size": (c_u64, [c_int, c_int]),
        "ecgpu_htr_beacon_state_dev_checked": (c_int, [c_int, u8p, c_u64, u8p, c_int, u8p, ctypes.c_void_p, ctypes.c_void_p]),
        "ecgpu_beacon_state_shard_subroots_dev": (c_int, [c_int, u8p, c_u64, u8p, c_int, c_u32, c_u32, u8p, u8p, ctypes.c_void_p]),
        "ecgpu_htr_beacon_state_sharded_dev": (c_int, [c_int, u8p, c_u64, u8p, c_int, u8p, c_u32, u8p, u8p, ctypes.c_void_p]),
        "ecgpu_beacon_state_shard_lists": (c_u32, []),
        "ecgpu_resident_state_create_fork": (c_int, [c_int, c_int, u8p, c_u64, ctypes.POINTER(ctypes.c_void_p)]),
        "ecgpu_last_hash64_count": (c_u64, []),
        "ecgpu_resident_state_create": (c_int, [c_int, u8p, c_u64, ctypes.POINTER(ctypes.c_void_p)]),
        "ecgpu_resident_state_destroy": (None, [ctypes.c_void_p]),
        "ecgpu_resident_state_patch": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, u8p, c_u32]),
        "ecgpu_resident_state_append": (c_int, [ctypes.c_void_p, c_int, u8p, c_u64]),
        "ecgpu_resident_state_truncate": (c_int, [ctypes.c_void_p, c_int, c_u64]),
        "ecgpu_resident_state_replace": (c_int, [ctypes.c_void_p, c_int, u8p, c_u64]),
        "ecgpu_resident_state_size": (c_u64, [ctypes.c_void_p]),
        "ecgpu_resident_state_patch_field": (c_int, [ctypes.c_void_p, c_u32, c_u64, u8p, c_u64]),
        "ecgpu_resident_state_patch_elements": (c_int, [ctypes.c_void_p, c_u32, c_u64, u8p, c_u64]),
        "ecgpu_resident_state_push": (c_int, [ctypes.c_void_p, c_u32, u8p, c_u64]),
        "ecgpu_resident_state_truncate_field": (c_int, [ctypes.c_void_p, c_u32, c_u64]),
        "ecgpu_resident_state_set_field": (c_int, [ctypes.c_void_p, c_u32, u8p, c_u64]),
        "ecgpu_resident_state_add_validator": (c_int, [ctypes.c_void_p, u8p, c_u64]),
        "ecgpu_resident_state_rotate_participation": (c_int, [ctypes.c_void_p]),
        "ecgpu_resident_state_flush": (c_int, [ctypes.c_void_p]),
        "ecgpu_resident_state_field_size": (ctypes.c_int64, [ctypes.c_void_p, c_u32]),
        "ecgpu_resident_state_root": (c_int, [ctypes.c_void_p, u8p]),
        "ecgpu_resident_state_root_dev": (c_int, [ctypes.c_void_p, u8p, ctypes.c_void_p]),
        "ecgpu_compute_shuffled_indices": (c_int, [ctypes.c_void_p, c_u64, u8p, c_u32, ctypes.c_void_p]),
        "ecgpu_compute_shuffled_indices_dev": (c_int, [ctypes.c_void_p, c_u64, u8p, c_u32, ctypes.c_void_p, ctypes.c_void_p]),
        "ecgpu_htr_ssz": (c_int, [ctypes.c_void_p, c_u32, ctypes.c_void_p, c_u32, c_u32, u8p, c_u64, u8p]),
        "ecgpu_ssz_generalized_index": (c_int, [ctypes.c_void_p, c_u32, ctypes.c_void_p, c_u32, c_u32, ctypes.c_void_p, c_u32,
                                                ctypes.POINTER(c_u64)]),
        "ecgpu_ssz_prove": (c_int, [ctypes.c_void_p, c_u32, ctypes.c_void_p, c_u32, c_u32, u8p, c_u64, ctypes.c_void_p, c_u32, u8p, u8p,
                                    c_u32, ctypes.POINTER(c_u32), ctypes.POINTER(c_u64), u8p]),
        "ecgpu_merkle_proof": (c_int, [u8p, c_u64, c_u64, c_u64, u8p]),
        "ecgpu_beacon_state_field_roots": (c_int, [c_int, u8p, c_u64, c_int, u8p, c_u32, ctypes.POINTER(c_u32), u8p]),
        "ecgpu_verify": (c_int, [u8p, u8p, c_size, u8p]),
        "ecgpu_fast_aggregate_verify": (c_int, [u8p, c_u32, u8p, c_size, u8p, c_int]),
        "ecgpu_aggregate_verify": (c_int, [u8p, c_u32, u8p, ctypes.c_void_p, c_u32, u8p]),
        "ecgpu_aggregate_sigs": (c_int, [u8p, c_u32, u8p]),
        "ecgpu_aggregate_pks": (c_int, [u8p, c_u32, u8p]),
        "ecgpu_g1_msm": (c_int, [u8p, u8p, c_u32, c_u32, u8p]),
        "ecgpu_g2_msm": (c_int, [u8p, u8p, c_u32, c_u32, u8p]),
        "ecgpu_fast_aggregate_verify_batch": (c_int, [u8p, ctypes.c_void_p, u8p, u8p, c_u32, c_int, u8p]),
        "ecgpu_fast_aggregate_verify_batch_dev": (c_int, [u8p, ctypes.c_void_p, c_u32, u8p, u8p, c_u32, c_int, u8p,
                                                          ctypes.c_void_p]),
        "ecgpu_registry_create": (c_int, [c_u64, ctypes.POINTER(ctypes.c_void_p)]),
        "ecgpu_registry_destroy": (None, [ctypes.c_void_p]),
        "ecgpu_registry_set": (c_int, [ctypes.c_void_p, c_u64, u8p, c_u64]),
        "ecgpu_registry_set_dev": (c_int, [ctypes.c_void_p, c_u64, u8p, c_u64, ctypes.c_void_p]),
        "ecgpu_fast_aggregate_verify_indexed_batch": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, u8p, u8p, c_u32, c_int,
                                                              u8p]),
        "ecgpu_fast_aggregate_verify_indexed_batch_dev": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_u32, u8p, u8p,
                                                                  c_u32, c_int, u8p, ctypes.c_void_p]),
        "ecgpu_batch_create": (c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
        "ecgpu_batch_destroy": (None, [ctypes.c_void_p]),
        "ecgpu_batch_push": (ctypes.c_int64, [ctypes.c_void_p, u8p, c_u32, u8p, c_size, u8p, c_int]),
        "ecgpu_batch_push_indexed": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, c_u32, u8p, c_size, u8p, c_int]),
        "ecgpu_batch_len": (c_u32, [ctypes.c_void_p]),
        "ecgpu_batch_flush": (c_int, [ctypes.c_void_p, u8p, c_u32]),
        "ecgpu_sk_to_pk_batch": (c_int, [u8p, c_u32, u8p]),
        "ecgpu_sign_batch": (c_int, [u8p, u8p, ctypes.c_void_p, c_u32, u8p]),
        "ecgpu_sk_to_pk_batch_dev": (c_int, [u8p, c_u32, u8p, ctypes.c_void_p]),
        "ecgpu_sign_batch_dev": (c_int, [u8p, c_u32, u8p, c_u32, u8p, ctypes.c_void_p]),
        "ecgpu_prof_enable": (c_int, [c_int]),
        "ecgpu_prof_filter": (c_int, [ctypes.c_char_p]),
        "ecgpu_prof_read": (c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_u64)]),
        "ecgpu_selfcheck_ifetch": (c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
        "ecgpu_selfcheck_ifetch_sweep": (c_int, [ctypes.POINTER(ctypes.c_double)]),
        "ecgpu_bls_tower": (c_int, []),
        "ecgpu_bls_last_pairing_path": (c_int, []),
        "ecgpu_bls_dispatch_thresholds": (c_int, [ctypes.c_void_p]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise ImportError("libecgpu.so does not export: " + ", ".join(missing))
    _lib = L
    return L


ABI_SYMBOLS = None  # filled lazily by tests from include/ecgpu.h


def check(code: int, where: str) -> int:
    """Negative codes are backend faults -> raise; 0..7 / EMPTY_AGGREGATE are returned."""
    if code < 0 and code != -100:
        raise EcgpuError(code, where, (load().ecgpu_last_error() or b"").decode())
    return code


def prof_read(tag: str | None = None):
    L = load()
    ms = ctypes.c_double(0)
    n = ctypes.c_uint64(0)
    L.ecgpu_prof_read(tag.encode() if tag else None, ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value
