"""Host-side mirror of the reference's Merkleization entry points, backed by the HIP kernels.

Names follow the reference / ssz_rs: `hash_tree_root` of the hot-path types
(/root/reference/ethereum-consensus/src/phase0/slot_processing.rs:67,75,
phase0/state_transition.rs:60, signing.rs:14-22), `is_valid_merkle_branch`
(phase0/block_processing.rs:433).  Inputs are SSZ encodings (bytes); results are 32-byte roots.
Every hash runs on the GPU through libecgpu.so; a missing library or device raises.
"""
from __future__ import annotations

import ctypes

from . import _lib

MAINNET, MINIMAL = 0, 1
VALIDATOR_REGISTRY_LIMIT = 1 << 40


class MerkleizationError(RuntimeError):
    """ssz_rs `MerkleizationError` analogue (e.g. input exceeds the type's limit)."""


def _buf(b: bytes):
    return ctypes.create_string_buffer(bytes(b), len(b)) if len(b) else ctypes.create_string_buffer(1)


def _root(call, *args) -> bytes:
    out = ctypes.create_string_buffer(32)
    rc = call(*args, out)
    if rc == -3:
        raise MerkleizationError((_lib.load().ecgpu_last_error() or b"bad argument").decode())
    _lib.check(rc, call.__name__)
    return out.raw


def hash(data: bytes) -> bytes:  # noqa: A001 - the reference's name (crypto/bls.rs:12)
    """crypto::hash: SHA-256."""
    L = _lib.load()
    return _root(L.ecgpu_sha256, _buf(data), len(data))


def merkleize(data: bytes, limit_chunks: int = 0, mix_in_length: int | None = None) -> bytes:
    """ssz_rs `merkleize(pack(data), limit)` [+ `mix_in_length`]."""
    L = _lib.load()
    return _root(L.ecgpu_merkleize, _buf(data), len(data), limit_chunks, 0 if mix_in_length is None else 1,
                 mix_in_length or 0)


def hash_tree_root_validators(ssz121: bytes, limit: int = VALIDATOR_REGISTRY_LIMIT) -> bytes:
    """hash_tree_root(List<Validator, limit>) from packed 121-byte records."""
    if len(ssz121) % 121:
        raise MerkleizationError("validator encoding is not a multiple of 121 bytes")
    L = _lib.load()
    return _root(L.ecgpu_htr_validators, _buf(ssz121), len(ssz121) // 121, limit)


def hash_tree_root_validators_multi(devices, ssz121: bytes, limit: int = VALIDATOR_REGISTRY_LIMIT) -> bytes:
    """hash_tree_root(List<Validator, limit>) with the registry sharded over several GPUs of this process."""
    if len(ssz121) % 121:
        raise MerkleizationError("validator encoding is not a multiple of 121 bytes")
    L = _lib.load()
    devs = (ctypes.c_int * len(devices))(*devices)
    return _root(L.ecgpu_htr_validators_multi, devs, len(devices), _buf(ssz121), len(ssz121) // 121, limit)


def validators_subtree_root(ssz121: bytes, width: int) -> bytes:
    """One shard's share of a sharded `List<Validator, N>`: root of the aligned subtree of `width`
    validators (a power of two), no length mix-in."""
    if len(ssz121) % 121:
        raise MerkleizationError("validator encoding is not a multiple of 121 bytes")
    L = _lib.load()
    return _root(L.ecgpu_validators_subtree_root, _buf(ssz121), len(ssz121) // 121, width)


def merkleize_subtree_roots(sub_roots: bytes, width: int, limit: int, mix_in_length: int | None = None) -> bytes:
    """Top of a sharded list: sub-roots of aligned `width`-leaf subtrees -> root of the `limit`-leaf tree [+ length]."""
    L = _lib.load()
    return _root(L.ecgpu_merkleize_subtree_roots, _buf(sub_roots), len(sub_roots) // 32, width, limit,
                 0 if mix_in_length is None else 1, mix_in_length or 0)


def hash_tree_root_validators_sharded(dist, ssz121_local: bytes, n_total: int, limit: int = VALIDATOR_REGISTRY_LIMIT) -> bytes:
    """hash_tree_root(List<Validator, limit>) of a registry sharded over the ranks of `dist` (torch.distributed, one
    process per GPU): rank r holds validators [r W, (r + 1) W) with W = shard.subtree_width(n_total, world).  One
    all-gather of 32-byte sub-roots, then every rank finishes the top of the tree."""
    from . import shard
    return shard.sharded_list_root(dist, n_total, limit, lambda w: validators_subtree_root(ssz121_local, w),
                                   merkleize_subtree_roots, mix_in_length=n_total)


def merkleize_sharded(dist, data_local: bytes, n_chunks_total: int, limit_chunks: int, mix_in_length: int | None = None) -> bytes:
    """`merkleize(pack(data), limit_chunks)` [+ mix_in_length] of a packed basic list (balances, participation,
    inactivity scores) whose chunks are sharded like `hash_tree_root_validators_sharded` shards validators."""
    from . import shard
    return shard.sharded_list_root(dist, n_chunks_total, limit_chunks, lambda w: merkleize(data_local, w),
                                   merkleize_subtree_roots, mix_in_length=mix_in_length)


def hash_tree_root_beacon_block_header(ssz112: bytes) -> bytes:
    if len(ssz112) != 112:
        raise MerkleizationError("BeaconBlockHeader encoding must be 112 bytes")
    L = _lib.load()
    return _root(L.ecgpu_htr_beacon_block_header, _buf(ssz112))


def compute_signing_root(object_root: bytes, domain: bytes) -> bytes:
    """signing.rs:14-22 with the object's root already computed."""
    L = _lib.load()
    return _root(L.ecgpu_signing_root, _buf(_node32(object_root, "object root")), _buf(_node32(domain, "domain")))


def hash_tree_root_beacon_state_deneb(ssz: bytes, preset: int = MAINNET) -> bytes:
    L = _lib.load()
    return _root(L.ecgpu_htr_beacon_state_deneb, _buf(ssz), len(ssz), preset)


FORKS = {"phase0": 0, "altair": 1, "bellatrix": 2, "capella": 3, "deneb": 4, "electra": 5}


def hash_tree_root_beacon_state(fork, ssz: bytes, preset: int = MAINNET) -> bytes:
    """`BeaconState::hash_tree_root` of any fork up to deneb (phase0/beacon_state.rs:50, altair/beacon_state.rs:13,
    bellatrix/beacon_state.rs:13, capella/beacon_state.rs:13, deneb/beacon_state.rs:13) from the state's SSZ serialization;
    called per slot (phase0/slot_processing.rs:67) and per block (phase0/state_transition.rs:60)."""
    L = _lib.load()
    return _root(L.ecgpu_htr_beacon_state, FORKS[fork] if isinstance(fork, str) else int(fork), _buf(ssz), len(ssz), preset)


class ResidentBeaconStateDeneb:
    """A BeaconState (deneb unless `fork` says phase0 / altair / bellatrix / capella / electra) kept in HBM: uploaded once, then patched in place with the bytes a block changed and
    re-Merkleized on the device (the reference re-hashes the host-resident state every slot,
    phase0/slot_processing.rs:67)."""

    def __init__(self, encoding: bytes, preset: int = MAINNET, fork="deneb"):
        self._h = None
        self._L = _lib.load()
        h = ctypes.c_void_p()
        self._fork = FORKS[fork] if isinstance(fork, str) else int(fork)
        rc = self._L.ecgpu_resident_state_create_fork(self._fork, preset, _buf(encoding),
                                                      len(encoding), ctypes.byref(h))
        if rc == -3:
            raise MerkleizationError((self._L.ecgpu_last_error() or b"bad state").decode())
        _lib.check(rc, "ecgpu_resident_state_create")
        self._h = h

    def close(self):
        if self._h:
            self._L.ecgpu_resident_state_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._h

    def patch(self, patches) -> None:
        """patches: iterable of (offset into the SSZ encoding, replacement bytes)"""
        patches = list(patches)
        if not patches:
            return
        offs = (ctypes.c_uint64 * len(patches))(*[o for o, _ in patches])
        doff = [0]
        for _, b in patches:
            doff.append(doff[-1] + len(b))
        rc = self._L.ecgpu_resident_state_patch(self._h, offs, (ctypes.c_uint64 * len(doff))(*doff), _buf(b"".join(b for _, b in patches)),
                                                len(patches))
        if rc == -3:
            raise MerkleizationError((self._L.ecgpu_last_error() or b"bad patch").decode())
        _lib.check(rc, "ecgpu_resident_state_patch")

    VALIDATORS, BALANCES, PREVIOUS_EPOCH_PARTICIPATION, CURRENT_EPOCH_PARTICIPATION, INACTIVITY_SCORES = 2, 3, 4, 5, 6
    HISTORICAL_ROOTS, ETH1_DATA_VOTES, HISTORICAL_SUMMARIES = 0, 1, 8
    PENDING_BALANCE_DEPOSITS, PENDING_PARTIAL_WITHDRAWALS, PENDING_CONSOLIDATIONS = 9, 10, 11  # electra

    def append(self, field: int, data: bytes) -> None:
        """whole elements appended to a variable-length list of the state (a new validator = five appends)"""
        rc = self._L.ecgpu_resident_state_append(self._h, field, _buf(data), len(data))
        if rc == -3:
            raise MerkleizationError((self._L.ecgpu_last_error() or b"bad append").decode())
        _lib.check(rc, "ecgpu_resident_state_append")

    PREVIOUS_EPOCH_ATTESTATIONS, CURRENT_EPOCH_ATTESTATIONS = 4, 5  # phase0 (in place of the participation lists): replace() only

    def replace(self, field: int, data: bytes) -> None:
        """a variable-length list exchanged for a new serialization of it (phase0: the two PendingAttestation lists change
        this way -- process_attestation, phase0/block_processing.rs:160-189, and the rotation at the epoch boundary)"""
        rc = self._L.ecgpu_resident_state_replace(self._h, field, _buf(data), len(data))
        if rc == -3:
            raise MerkleizationError((self._L.ecgpu_last_error() or b"bad replace").decode())
        _lib.check(rc, "ecgpu_resident_state_replace")

    def truncate(self, field: int, new_n_bytes: int) -> None:
        rc = self._L.ecgpu_resident_state_truncate(self._h, field, new_n_bytes)
        if rc == -3:
            raise MerkleizationError((self._L.ecgpu_last_error() or b"bad truncate").decode())
        _lib.check(rc, "ecgpu_resident_state_truncate")

    def add_validator(self, validator121: bytes, balance: int) -> None:
        """add_validator_to_registry (phase0/block_processing.rs:317-349; altair and later: + participation flags, inactivity
        score): queued, applied at the next root -- one call of the C ABI, the fork's list of pushes is the library's"""
        if len(validator121) != 121:
            raise MerkleizationError("a Validator record is 121 bytes")
        self._field_rc(self._L.ecgpu_resident_state_add_validator(self._h, _buf(validator121), int(balance)), "add_validator")

    # ---- field-addressed changes (include/ecgpu.h ECGPU_BS_*: the field's POSITION in the fork's BeaconState container) ----
    FIELD_POSITIONS = {name: i for i, name in enumerate((
        "genesis_time", "genesis_validators_root", "slot", "fork", "latest_block_header", "block_roots", "state_roots",
        "historical_roots", "eth1_data", "eth1_data_votes", "eth1_deposit_index", "validators", "balances", "randao_mixes",
        "slashings", "previous_epoch_participation", "current_epoch_participation", "justification_bits",
        "previous_justified_checkpoint", "current_justified_checkpoint", "finalized_checkpoint", "inactivity_scores",
        "current_sync_committee", "next_sync_committee", "latest_execution_payload_header", "next_withdrawal_index",
        "next_withdrawal_validator_index", "historical_summaries", "deposit_receipts_start_index", "deposit_balance_to_consume",
        "exit_balance_to_consume", "earliest_exit_epoch", "consolidation_balance_to_consume", "earliest_consolidation_epoch",
        "pending_balance_deposits", "pending_partial_withdrawals", "pending_consolidations"))}
    FIELD_POSITIONS["previous_epoch_attestations"] = 15  # phase0
    FIELD_POSITIONS["current_epoch_attestations"] = 16

    def _pos(self, field) -> int:
        return self.FIELD_POSITIONS[field] if isinstance(field, str) else int(field)

    def _field_rc(self, rc: int, what: str) -> None:
        if rc == -3:
            raise MerkleizationError((self._L.ecgpu_last_error() or what.encode()).decode())
        _lib.check(rc, "ecgpu_resident_state_" + what)

    def patch_field(self, field, offset_in_field: int, data: bytes) -> None:
        """bytes of one field overwritten, addressed INSIDE the field (`state.slot = ..`, one byte of a validator record ...)"""
        self._field_rc(self._L.ecgpu_resident_state_patch_field(self._h, self._pos(field), offset_in_field, _buf(data), len(data)), "patch_field")

    def patch_elements(self, field, first_index: int, data: bytes) -> None:
        """elements first_index .. of a list / vector overwritten (`state.balances[i] = ..`, `state.block_roots[slot % N] = ..`)"""
        self._field_rc(self._L.ecgpu_resident_state_patch_elements(self._h, self._pos(field), first_index, _buf(data), len(data)), "patch_elements")

    def push(self, field, data: bytes) -> None:
        """`state.<list>.push(..)`: whole elements appended"""
        self._field_rc(self._L.ecgpu_resident_state_push(self._h, self._pos(field), _buf(data), len(data)), "push")

    def truncate_field(self, field, new_n_bytes: int) -> None:
        self._field_rc(self._L.ecgpu_resident_state_truncate_field(self._h, self._pos(field), new_n_bytes), "truncate_field")

    def set_field(self, field, data: bytes) -> None:
        """the whole field exchanged for a new serialization of it"""
        self._field_rc(self._L.ecgpu_resident_state_set_field(self._h, self._pos(field), _buf(data), len(data)), "set_field")

    def rotate_participation(self) -> None:
        """process_participation_flag_updates: previous = current, current = zeros (on the device)"""
        self._field_rc(self._L.ecgpu_resident_state_rotate_participation(self._h), "rotate_participation")

    def flush(self) -> None:
        self._field_rc(self._L.ecgpu_resident_state_flush(self._h), "flush")

    def field_size(self, field) -> int:
        n = int(self._L.ecgpu_resident_state_field_size(self._h, self._pos(field)))
        if n < 0:
            raise MerkleizationError("no such field in this fork")
        return n

    def __len__(self):
        return int(self._L.ecgpu_resident_state_size(self._h))

    def hash_tree_root(self) -> bytes:
        return _root(self._L.ecgpu_resident_state_root, self._h)


_compiled = {}


def hash_tree_root(ssz_type, encoding: bytes) -> bytes:
    """`T::hash_tree_root` for any SSZ type described with ssz_types (what #[derive(SimpleSerialize)] generates,
    e.g. deneb BeaconBlock: deneb/beacon_block.rs:12-91) from the value's SSZ serialization."""
    from . import ssz_types
    L = _lib.load()
    if ssz_type not in _compiled:
        _compiled[ssz_type] = ssz_types.compile(ssz_type)
    arr, farr, nf, root_idx = _compiled[ssz_type]
    out = ctypes.create_string_buffer(32)
    _lib.check(L.ecgpu_htr_ssz(arr, len(arr), farr, nf, root_idx, _buf(encoding), len(encoding), out), "ecgpu_htr_ssz")
    return out.raw


def _node32(b, what: str) -> bytes:
    b = bytes(b)
    if len(b) != 32:
        raise MerkleizationError(f"{what} must be a 32-byte node, got {len(b)} bytes")
    return b


def is_valid_merkle_branch(leaf: bytes, branch, depth: int, index: int, root: bytes) -> bool:
    """ssz_rs `is_valid_merkle_branch` (phase0/block_processing.rs:433, deneb/blob_sidecar.rs:62): a branch shorter than
    `depth` is not a proof (False); nodes are `Node`s, i.e. exactly 32 bytes each."""
    L = _lib.load()
    leaf, root = _node32(leaf, "leaf"), _node32(root, "root")
    if depth < 0 or depth > 64:
        raise MerkleizationError(f"depth {depth} out of range")
    branch = list(branch)
    if len(branch) < depth:
        return False
    b = b"".join(_node32(x, "branch node") for x in branch[:depth])
    rc = L.ecgpu_is_valid_merkle_branch(_buf(leaf), _buf(b), depth, index, _buf(root))
    _lib.check(rc, "ecgpu_is_valid_merkle_branch")
    return rc == 0


# ---- proofs (ssz_rs `Prove` / `GeneralizedIndexable`; spec-tests/runners/light_client.rs:42-69, deneb/blob_sidecar.rs:47-64) ----
LENGTH = "__len__"


def _path_positions(ssz_type, path):
    """field names -> field positions, element indices as they are, LENGTH -> ECGPU_SSZ_PATH_LENGTH"""
    from . import ssz_types as T
    out, t = [], ssz_type
    for p in path:
        if p == LENGTH:
            out.append(0xFFFFFFFFFFFFFFFF)
            t = None
            continue
        if t is not None and t[0] == T.CONTAINER:
            i = t[2].index(p) if isinstance(p, str) else int(p)
            out.append(i)
            t = t[1][i]
        else:
            out.append(int(p))
            t = t[1] if t is not None and t[0] in (T.VECTOR, T.LIST) and t[1][0] != T.UINT else None
    return out


def generalized_index(ssz_type, path) -> int:
    """`T::generalized_index(path)` (deneb/beacon_block.rs:139-154, deneb/blob_sidecar.rs:56-57)"""
    from . import ssz_types
    L = _lib.load()
    if ssz_type not in _compiled:
        _compiled[ssz_type] = ssz_types.compile(ssz_type)
    arr, farr, nf, root_idx = _compiled[ssz_type]
    pos = _path_positions(ssz_type, path)
    parr = (ctypes.c_uint64 * max(len(pos), 1))(*pos)
    g = ctypes.c_uint64(0)
    rc = L.ecgpu_ssz_generalized_index(arr, len(arr), farr, nf, root_idx, parr, len(pos), ctypes.byref(g))
    if rc == -3:
        raise MerkleizationError((L.ecgpu_last_error() or b"bad path").decode())
    _lib.check(rc, "ecgpu_ssz_generalized_index")
    return g.value


def prove(ssz_type, encoding: bytes, path):
    """`value.prove(path)` from the value's serialization -> (leaf, branch [bottom-up], generalized index, witness root)"""
    from . import ssz_types
    L = _lib.load()
    if ssz_type not in _compiled:
        _compiled[ssz_type] = ssz_types.compile(ssz_type)
    arr, farr, nf, root_idx = _compiled[ssz_type]
    pos = _path_positions(ssz_type, path)
    parr = (ctypes.c_uint64 * max(len(pos), 1))(*pos)
    leaf, root = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    branch = ctypes.create_string_buffer(32 * 128)
    depth, g = ctypes.c_uint32(0), ctypes.c_uint64(0)
    rc = L.ecgpu_ssz_prove(arr, len(arr), farr, nf, root_idx, _buf(encoding), len(encoding), parr, len(pos), leaf, branch, 128,
                           ctypes.byref(depth), ctypes.byref(g), root)
    if rc == -3:
        raise MerkleizationError((L.ecgpu_last_error() or b"bad path or encoding").decode())
    _lib.check(rc, "ecgpu_ssz_prove")
    return leaf.raw, [branch.raw[32 * i:32 * i + 32] for i in range(depth.value)], g.value, root.raw


def merkle_proof(chunks: bytes, limit_chunks: int, index: int):
    """the branch of chunk `index` in merkleize(chunks, limit_chunks), bottom-up"""
    L = _lib.load()
    if len(chunks) % 32:
        raise MerkleizationError("chunks must be a whole number of 32-byte chunks")
    n = len(chunks) // 32
    # the C entry's rule (csrc/ssz_proof.hip): limit_chunks == 0 means "no limit beyond the data": the tree of the chunks themselves
    eff = limit_chunks or max(n, 1)
    if n > eff or index < 0:
        raise MerkleizationError("bad proof request")
    depth = (eff - 1).bit_length()
    if index >= 1 << depth:
        raise MerkleizationError("bad proof request")
    out = ctypes.create_string_buffer(max(32 * depth, 1))
    rc = L.ecgpu_merkle_proof(_buf(chunks), n, limit_chunks, index, out)
    if rc == -3:
        raise MerkleizationError("bad proof request")
    _lib.check(rc, "ecgpu_merkle_proof")
    return [out.raw[32 * i:32 * i + 32] for i in range(depth)]


def beacon_state_field_roots(fork, ssz: bytes, preset: int = MAINNET):
    """(roots of the state's fields, state root): one state root's worth of work"""
    L = _lib.load()
    roots = ctypes.create_string_buffer(32 * 32)
    n = ctypes.c_uint32(0)
    root = ctypes.create_string_buffer(32)
    rc = L.ecgpu_beacon_state_field_roots(FORKS[fork] if isinstance(fork, str) else int(fork), _buf(ssz), len(ssz), preset, roots, 32,
                                          ctypes.byref(n), root)
    if rc == -3:
        raise MerkleizationError((L.ecgpu_last_error() or b"bad state").decode())
    _lib.check(rc, "ecgpu_beacon_state_field_roots")
    return [roots.raw[32 * i:32 * i + 32] for i in range(n.value)], root.raw


def prove_beacon_state_field(fork, ssz: bytes, preset: int, field_position: int):
    """light-client style branch of one field of a BeaconState (spec-tests/runners/light_client.rs:32-40: current / next sync
    committee; finalized_checkpoint, continued below the field by `prove` on the field's own encoding):
    -> (leaf = the field's root, branch of depth 5, generalized index 32 + position, state root)"""
    roots, root = beacon_state_field_roots(fork, ssz, preset)
    return roots[field_position], merkle_proof(b"".join(roots), 32, field_position), 32 + field_position, root


def last_hash64_count() -> int:
    return int(_lib.load().ecgpu_last_hash64_count())
