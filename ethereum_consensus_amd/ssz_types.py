"""SSZ type descriptions for `ssz.hash_tree_root(type, encoding)` (C ABI: ecgpu_htr_ssz, include/ecgpu.h) and the
deneb block types of the reference, field for field:

    phase0/operations.rs:13-140   Checkpoint, AttestationData, IndexedAttestation, Attestation, Eth1Data, DepositData,
                                  ProposerSlashing, AttesterSlashing, Deposit, VoluntaryExit, SignedVoluntaryExit
    phase0/beacon_block.rs:83-100 BeaconBlockHeader, SignedBeaconBlockHeader
    altair/sync.rs:9-12           SyncAggregate
    capella/withdrawal.rs:9-17, capella/bls_to_execution_change.rs:9-23
    deneb/execution_payload.rs:13-46, bellatrix/execution_payload.rs:8 (Transaction = ByteList)
    deneb/beacon_block.rs:12-91   BeaconBlockBody, BeaconBlock
    limits: phase0/presets/mainnet.rs:7,32-36, altair/presets/mainnet.rs:19, bellatrix/presets/mainnet.rs:21-24,
            capella/presets/mainnet.rs:18-19, deneb/presets/mainnet.rs:20 (and the minimal presets next to them)

A type is an immutable tuple tree; `compile()` flattens it into the arrays of the C ABI (children before parents,
structurally equal types shared)."""
from __future__ import annotations

import ctypes

UINT, BYTEVECTOR, BYTELIST, VECTOR, LIST, BITVECTOR, BITLIST, CONTAINER = range(8)


def uint(bits: int):
    return (UINT, bits // 8)


boolean = (UINT, 1)
uint8, uint64, uint256 = uint(8), uint(64), uint(256)


def bytevector(n: int):
    return (BYTEVECTOR, n)


def bytelist(limit: int):
    return (BYTELIST, limit)


def vector(elem, n: int):
    return (VECTOR, elem, n)


def list_(elem, limit: int):
    return (LIST, elem, limit)


def bitvector(n: int):
    return (BITVECTOR, n)


def bitlist(limit: int):
    return (BITLIST, limit)


def container(*fields):
    """fields: (name, type) pairs; names are documentation only"""
    return (CONTAINER, tuple(t for _, t in fields), tuple(n for n, _ in fields))


class SszTypeC(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("elem", ctypes.c_uint32), ("param", ctypes.c_uint64),
                ("n_fields", ctypes.c_uint32), ("first_field", ctypes.c_uint32)]


def compile(t):  # noqa: A001
    """-> (ctypes array of SszTypeC, ctypes uint32 array of field refs, index of `t`)"""
    types, fields, memo = [], [], {}

    def go(x):
        key = x[:2] if x[0] == CONTAINER else x
        if key in memo:
            return memo[key]
        k = x[0]
        if k in (VECTOR, LIST):
            e = go(x[1])
            types.append((k, e, x[2], 0, 0))
        elif k == CONTAINER:
            refs = [go(f) for f in x[1]]
            types.append((k, 0, 0, len(refs), len(fields)))
            fields.extend(refs)
        else:
            types.append((k, 0, x[1], 0, 0))
        memo[key] = len(types) - 1
        return memo[key]

    root = go(t)
    arr = (SszTypeC * len(types))(*[SszTypeC(*v) for v in types])
    farr = (ctypes.c_uint32 * max(len(fields), 1))(*fields)
    return arr, farr, len(fields), root


# ---- the reference's containers ---------------------------------------------------------------------------------
Root = Bytes32 = Hash32 = bytevector(32)
BlsPublicKey, BlsSignature, KzgCommitment = bytevector(48), bytevector(96), bytevector(48)
ExecutionAddress, Version = bytevector(20), bytevector(4)

Checkpoint = container(("epoch", uint64), ("root", Root))
AttestationData = container(("slot", uint64), ("index", uint64), ("beacon_block_root", Root), ("source", Checkpoint),
                            ("target", Checkpoint))
Eth1Data = container(("deposit_root", Root), ("deposit_count", uint64), ("block_hash", Hash32))
BeaconBlockHeader = container(("slot", uint64), ("proposer_index", uint64), ("parent_root", Root), ("state_root", Root),
                              ("body_root", Root))
SignedBeaconBlockHeader = container(("message", BeaconBlockHeader), ("signature", BlsSignature))
ProposerSlashing = container(("signed_header_1", SignedBeaconBlockHeader), ("signed_header_2", SignedBeaconBlockHeader))
DepositData = container(("public_key", BlsPublicKey), ("withdrawal_credentials", Bytes32), ("amount", uint64),
                        ("signature", BlsSignature))
Deposit = container(("proof", vector(Root, 33)), ("data", DepositData))
VoluntaryExit = container(("epoch", uint64), ("validator_index", uint64))
SignedVoluntaryExit = container(("message", VoluntaryExit), ("signature", BlsSignature))
Withdrawal = container(("index", uint64), ("validator_index", uint64), ("address", ExecutionAddress), ("amount", uint64))
BlsToExecutionChange = container(("validator_index", uint64), ("from_bls_public_key", BlsPublicKey),
                                 ("to_execution_address", ExecutionAddress))
SignedBlsToExecutionChange = container(("message", BlsToExecutionChange), ("signature", BlsSignature))
SigningData = container(("object_root", Root), ("domain", Bytes32))

MAINNET = dict(MAX_PROPOSER_SLASHINGS=16, MAX_VALIDATORS_PER_COMMITTEE=2048, MAX_ATTESTER_SLASHINGS=2, MAX_ATTESTATIONS=128,
               MAX_DEPOSITS=16, MAX_VOLUNTARY_EXITS=16, SYNC_COMMITTEE_SIZE=512, BYTES_PER_LOGS_BLOOM=256, MAX_EXTRA_DATA_BYTES=32,
               MAX_BYTES_PER_TRANSACTION=1 << 30, MAX_TRANSACTIONS_PER_PAYLOAD=1 << 20, MAX_WITHDRAWALS_PER_PAYLOAD=16,
               MAX_BLS_TO_EXECUTION_CHANGES=16, MAX_BLOB_COMMITMENTS_PER_BLOCK=4096)
MINIMAL = dict(MAINNET, SYNC_COMMITTEE_SIZE=32, MAX_WITHDRAWALS_PER_PAYLOAD=4, MAX_BLOB_COMMITMENTS_PER_BLOCK=16)


def IndexedAttestation(p):
    return container(("attesting_indices", list_(uint64, p["MAX_VALIDATORS_PER_COMMITTEE"])), ("data", AttestationData),
                     ("signature", BlsSignature))


def Attestation(p):
    return container(("aggregation_bits", bitlist(p["MAX_VALIDATORS_PER_COMMITTEE"])), ("data", AttestationData),
                     ("signature", BlsSignature))


def AttesterSlashing(p):
    return container(("attestation_1", IndexedAttestation(p)), ("attestation_2", IndexedAttestation(p)))


def SyncAggregate(p):
    return container(("sync_committee_bits", bitvector(p["SYNC_COMMITTEE_SIZE"])), ("sync_committee_signature", BlsSignature))


def ExecutionPayloadDeneb(p):
    return container(
        ("parent_hash", Hash32), ("fee_recipient", ExecutionAddress), ("state_root", Bytes32), ("receipts_root", Bytes32),
        ("logs_bloom", bytevector(p["BYTES_PER_LOGS_BLOOM"])), ("prev_randao", Bytes32), ("block_number", uint64),
        ("gas_limit", uint64), ("gas_used", uint64), ("timestamp", uint64), ("extra_data", bytelist(p["MAX_EXTRA_DATA_BYTES"])),
        ("base_fee_per_gas", uint256), ("block_hash", Hash32),
        ("transactions", list_(bytelist(p["MAX_BYTES_PER_TRANSACTION"]), p["MAX_TRANSACTIONS_PER_PAYLOAD"])),
        ("withdrawals", list_(Withdrawal, p["MAX_WITHDRAWALS_PER_PAYLOAD"])), ("blob_gas_used", uint64), ("excess_blob_gas", uint64))


def BeaconBlockBodyDeneb(p):
    return container(
        ("randao_reveal", BlsSignature), ("eth1_data", Eth1Data), ("graffiti", Bytes32),
        ("proposer_slashings", list_(ProposerSlashing, p["MAX_PROPOSER_SLASHINGS"])),
        ("attester_slashings", list_(AttesterSlashing(p), p["MAX_ATTESTER_SLASHINGS"])),
        ("attestations", list_(Attestation(p), p["MAX_ATTESTATIONS"])), ("deposits", list_(Deposit, p["MAX_DEPOSITS"])),
        ("voluntary_exits", list_(SignedVoluntaryExit, p["MAX_VOLUNTARY_EXITS"])), ("sync_aggregate", SyncAggregate(p)),
        ("execution_payload", ExecutionPayloadDeneb(p)),
        ("bls_to_execution_changes", list_(SignedBlsToExecutionChange, p["MAX_BLS_TO_EXECUTION_CHANGES"])),
        ("blob_kzg_commitments", list_(KzgCommitment, p["MAX_BLOB_COMMITMENTS_PER_BLOCK"])))


def BeaconBlockDeneb(p):
    return container(("slot", uint64), ("proposer_index", uint64), ("parent_root", Root), ("state_root", Root),
                     ("body", BeaconBlockBodyDeneb(p)))


# ---- electra (the alpha the reference implements: electra/beacon_block.rs:17-63, electra/operations.rs:10-50,
# electra/execution_payload.rs:13-45, electra/beacon_state.rs:16-25,62-68, electra/presets/{mainnet,minimal}.rs:13-17).
# MAX_VALIDATORS_PER_SLOT is a const parameter the reference never binds to a number; the specification's value is
# MAX_VALIDATORS_PER_COMMITTEE * MAX_COMMITTEES_PER_SLOT (phase0/presets/*.rs:5).
ELECTRA_MAINNET = dict(MAINNET, MAX_COMMITTEES_PER_SLOT=64, MAX_VALIDATORS_PER_SLOT=2048 * 64, MAX_ATTESTER_SLASHINGS_ELECTRA=1,
                       MAX_ATTESTATIONS_ELECTRA=8, MAX_CONSOLIDATIONS=1, MAX_DEPOSIT_RECEIPTS_PER_PAYLOAD=8192,
                       MAX_WITHDRAWAL_REQUESTS_PER_PAYLOAD=16)
ELECTRA_MINIMAL = dict(MINIMAL, MAX_COMMITTEES_PER_SLOT=4, MAX_VALIDATORS_PER_SLOT=2048 * 4, MAX_ATTESTER_SLASHINGS_ELECTRA=1,
                       MAX_ATTESTATIONS_ELECTRA=8, MAX_CONSOLIDATIONS=1, MAX_DEPOSIT_RECEIPTS_PER_PAYLOAD=4,
                       MAX_WITHDRAWAL_REQUESTS_PER_PAYLOAD=2)
DepositReceipt = container(("public_key", BlsPublicKey), ("withdrawal_credentials", Bytes32), ("amount", uint64),
                           ("signature", BlsSignature), ("index", uint64))
ExecutionLayerWithdrawalRequest = container(("source_address", ExecutionAddress), ("validator_public_key", BlsPublicKey),
                                            ("amount", uint64))
Consolidation = container(("source_index", uint64), ("target_index", uint64), ("epoch", uint64))
SignedConsolidation = container(("message", Consolidation), ("signature", BlsSignature))


def IndexedAttestationElectra(p):
    return container(("attesting_indices", list_(uint64, p["MAX_VALIDATORS_PER_SLOT"])), ("data", AttestationData),
                     ("signature", BlsSignature))


def AttestationElectra(p):
    return container(("aggregation_bits", bitlist(p["MAX_VALIDATORS_PER_SLOT"])), ("data", AttestationData),
                     ("committee_bits", bitvector(p["MAX_COMMITTEES_PER_SLOT"])), ("signature", BlsSignature))


def ExecutionPayloadElectra(p):
    return container(
        ("parent_hash", Hash32), ("fee_recipient", ExecutionAddress), ("state_root", Bytes32), ("receipts_root", Bytes32),
        ("logs_bloom", bytevector(p["BYTES_PER_LOGS_BLOOM"])), ("prev_randao", Bytes32), ("block_number", uint64),
        ("gas_limit", uint64), ("gas_used", uint64), ("timestamp", uint64), ("extra_data", bytelist(p["MAX_EXTRA_DATA_BYTES"])),
        ("base_fee_per_gas", uint256), ("block_hash", Hash32),
        ("transactions", list_(bytelist(p["MAX_BYTES_PER_TRANSACTION"]), p["MAX_TRANSACTIONS_PER_PAYLOAD"])),
        ("withdrawals", list_(Withdrawal, p["MAX_WITHDRAWALS_PER_PAYLOAD"])), ("blob_gas_used", uint64), ("excess_blob_gas", uint64),
        ("deposit_receipts", list_(DepositReceipt, p["MAX_DEPOSIT_RECEIPTS_PER_PAYLOAD"])),
        ("withdrawal_requests", list_(ExecutionLayerWithdrawalRequest, p["MAX_WITHDRAWAL_REQUESTS_PER_PAYLOAD"])))


def BeaconBlockBodyElectra(p):
    ix = IndexedAttestationElectra(p)
    return container(
        ("randao_reveal", BlsSignature), ("eth1_data", Eth1Data), ("graffiti", Bytes32),
        ("proposer_slashings", list_(ProposerSlashing, p["MAX_PROPOSER_SLASHINGS"])),
        ("attester_slashings", list_(container(("attestation_1", ix), ("attestation_2", ix)), p["MAX_ATTESTER_SLASHINGS_ELECTRA"])),
        ("attestations", list_(AttestationElectra(p), p["MAX_ATTESTATIONS_ELECTRA"])), ("deposits", list_(Deposit, p["MAX_DEPOSITS"])),
        ("voluntary_exits", list_(SignedVoluntaryExit, p["MAX_VOLUNTARY_EXITS"])), ("sync_aggregate", SyncAggregate(p)),
        ("execution_payload", ExecutionPayloadElectra(p)),
        ("bls_to_execution_changes", list_(SignedBlsToExecutionChange, p["MAX_BLS_TO_EXECUTION_CHANGES"])),
        ("blob_kzg_commitments", list_(KzgCommitment, p["MAX_BLOB_COMMITMENTS_PER_BLOCK"])),
        ("consolidations", list_(SignedConsolidation, p["MAX_CONSOLIDATIONS"])))


def BeaconBlockElectra(p):
    return container(("slot", uint64), ("proposer_index", uint64), ("parent_root", Root), ("state_root", Root),
                     ("body", BeaconBlockBodyElectra(p)))
