//! One declaration per host-pointer entry point of include/ecgpu.h (same order as the header).  Not declared: the `_dev`
//! variants (device pointers + a HIP stream: for hosts that own HBM buffers, i.e. not a Rust beacon node today), the key /
//! signature generators of the test harness (`ecgpu_sk_to_pk_batch`, `ecgpu_sign_batch`) and the profiling hooks.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const ECGPU_SUCCESS: c_int = 0;
pub const ECGPU_IN_VERIFY: c_int = 0x40;
pub const ECGPU_EMPTY_AGGREGATE: c_int = -100;
pub const ECGPU_ERR_NO_DEVICE: c_int = -1;
pub const ECGPU_ERR_HIP: c_int = -2;
pub const ECGPU_ERR_BAD_ARG: c_int = -3;
pub const ECGPU_ERR_OOM: c_int = -4;
// fork ids of the BeaconState entry points ({phase0,altair,bellatrix,capella,deneb}/beacon_state.rs)
pub const ECGPU_FORK_PHASE0: c_int = 0;
pub const ECGPU_FORK_ALTAIR: c_int = 1;
pub const ECGPU_FORK_BELLATRIX: c_int = 2;
pub const ECGPU_FORK_CAPELLA: c_int = 3;
pub const ECGPU_FORK_DENEB: c_int = 4;
pub const ECGPU_FORK_ELECTRA: c_int = 5;
// variable-length fields of a resident state (ecgpu_resident_state_append / _truncate)
pub const ECGPU_STATE_HISTORICAL_ROOTS: c_int = 0;
pub const ECGPU_STATE_ETH1_DATA_VOTES: c_int = 1;
pub const ECGPU_STATE_VALIDATORS: c_int = 2;
pub const ECGPU_STATE_BALANCES: c_int = 3;
pub const ECGPU_STATE_PREVIOUS_EPOCH_PARTICIPATION: c_int = 4;
pub const ECGPU_STATE_CURRENT_EPOCH_PARTICIPATION: c_int = 5;
pub const ECGPU_STATE_INACTIVITY_SCORES: c_int = 6;
pub const ECGPU_STATE_HISTORICAL_SUMMARIES: c_int = 8;
pub const ECGPU_STATE_PENDING_BALANCE_DEPOSITS: c_int = 9; // electra
pub const ECGPU_STATE_PENDING_PARTIAL_WITHDRAWALS: c_int = 10;
pub const ECGPU_STATE_PENDING_CONSOLIDATIONS: c_int = 11;
// field positions in the fork's BeaconState container (include/ecgpu.h ECGPU_BS_*): the coordinates of the field-addressed entries
pub const ECGPU_BS_SLOT: u32 = 2;
pub const ECGPU_BS_BLOCK_ROOTS: u32 = 5;
pub const ECGPU_BS_STATE_ROOTS: u32 = 6;
pub const ECGPU_BS_HISTORICAL_ROOTS: u32 = 7;
pub const ECGPU_BS_ETH1_DATA_VOTES: u32 = 9;
pub const ECGPU_BS_VALIDATORS: u32 = 11;
pub const ECGPU_BS_BALANCES: u32 = 12;
pub const ECGPU_BS_RANDAO_MIXES: u32 = 13;
pub const ECGPU_BS_SLASHINGS: u32 = 14;
pub const ECGPU_BS_PREVIOUS_EPOCH_PARTICIPATION: u32 = 15;
pub const ECGPU_BS_CURRENT_EPOCH_PARTICIPATION: u32 = 16;
pub const ECGPU_BS_INACTIVITY_SCORES: u32 = 21;
pub const ECGPU_BS_LATEST_EXECUTION_PAYLOAD_HEADER: u32 = 24;
pub const ECGPU_BS_HISTORICAL_SUMMARIES: u32 = 27;
pub const ECGPU_STATE_PREVIOUS_EPOCH_ATTESTATIONS: c_int = 4; // phase0, in place of the participation lists: replace only
pub const ECGPU_STATE_CURRENT_EPOCH_ATTESTATIONS: c_int = 5;

#[repr(C)]
pub struct ecgpu_registry_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ecgpu_batch_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ecgpu_resident_state_t {
    _private: [u8; 0],
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ecgpu_ssz_type {
    pub kind: u32,
    pub elem: u32,
    pub param: u64,
    pub n_fields: u32,
    pub first_field: u32,
}
pub type ecgpu_stream_t = *mut c_void;

extern "C" {
    pub fn ecgpu_init(device: c_int) -> c_int;
    pub fn ecgpu_device_count() -> c_int;
    pub fn ecgpu_last_error() -> *const c_char;
    pub fn ecgpu_version() -> *const c_char;
    pub fn ecgpu_bind_thread(device: c_int) -> c_int;
    pub fn ecgpu_selfcheck_ifetch(ms_small_loop: *mut f64, ms_large_loop: *mut f64) -> c_int;
    pub fn ecgpu_bls_tower() -> c_int;
    pub fn ecgpu_bls_last_pairing_path() -> c_int;

    pub fn ecgpu_sha256(data: *const u8, len: usize, out: *mut u8) -> c_int;
    pub fn ecgpu_merkleize(data: *const u8, n_bytes: u64, limit_chunks: u64, mix_in_len: c_int, len: u64, root: *mut u8) -> c_int;
    pub fn ecgpu_htr_validators(ssz121: *const u8, n: u64, limit: u64, root: *mut u8) -> c_int;
    pub fn ecgpu_htr_beacon_block_header(ssz112: *const u8, root: *mut u8) -> c_int;
    pub fn ecgpu_signing_root(object_root: *const u8, domain: *const u8, root: *mut u8) -> c_int;
    pub fn ecgpu_is_valid_merkle_branch(leaf: *const u8, branch: *const u8, depth: u32, index: u64, root: *const u8) -> c_int;
    pub fn ecgpu_htr_beacon_state(fork: c_int, ssz: *const u8, n_bytes: u64, preset: c_int, root: *mut u8) -> c_int;
    pub fn ecgpu_htr_ssz(types: *const ecgpu_ssz_type, n_types: u32, fields: *const u32, n_field_refs: u32, root_type: u32,
                         ssz: *const u8, n_bytes: u64, root: *mut u8) -> c_int;
    pub fn ecgpu_resident_state_create_fork(fork: c_int, preset: c_int, ssz: *const u8, n_bytes: u64,
                                            out: *mut *mut ecgpu_resident_state_t) -> c_int;
    pub fn ecgpu_resident_state_destroy(st: *mut ecgpu_resident_state_t);
    pub fn ecgpu_resident_state_patch(st: *mut ecgpu_resident_state_t, offsets: *const u64, data_off: *const u64, data: *const u8,
                                      n: u32) -> c_int;
    pub fn ecgpu_resident_state_root(st: *mut ecgpu_resident_state_t, root: *mut u8) -> c_int;
    pub fn ecgpu_sha256_batch(data: *const u8, len: usize, n: u64, out: *mut u8) -> c_int;
    // device-resident forms (round 4): the checked root, ONE state over several ranks, the thread's device
    pub fn ecgpu_htr_beacon_state_dev_checked(fork: c_int, d_ssz: *const u8, n_bytes: u64, h_fixed: *const u8, preset: c_int,
                                              d_root: *mut u8, d_status: *mut i32, stream: *mut core::ffi::c_void) -> c_int;
    pub fn ecgpu_beacon_state_shard_subroots_dev(fork: c_int, d_ssz: *const u8, n_bytes: u64, h_fixed: *const u8, preset: c_int,
                                                 rank: u32, world: u32, d_subroots: *mut u8, d_field_roots: *mut u8,
                                                 stream: *mut core::ffi::c_void) -> c_int;
    pub fn ecgpu_htr_beacon_state_sharded_dev(fork: c_int, d_ssz: *const u8, n_bytes: u64, h_fixed: *const u8, preset: c_int,
                                              d_all_subroots: *const u8, world: u32, d_field_roots: *const u8, d_root: *mut u8,
                                              stream: *mut core::ffi::c_void) -> c_int;
    pub fn ecgpu_beacon_state_shard_lists() -> u32;
    pub fn ecgpu_thread_device() -> c_int;
    pub fn ecgpu_validators_subtree_root(ssz121: *const u8, n: u64, width: u64, root: *mut u8) -> c_int;
    pub fn ecgpu_merkleize_subtree_roots(sub_roots: *const u8, n_sub: u32, width: u64, limit: u64, mix_in_len: c_int, len: u64,
                                         root: *mut u8) -> c_int;
    pub fn ecgpu_htr_validators_multi(devices: *const c_int, n_devices: u32, ssz121: *const u8, n: u64, limit: u64,
                                      root: *mut u8) -> c_int;
    pub fn ecgpu_beacon_state_fixed_size(fork: c_int, preset: c_int) -> u64;
    pub fn ecgpu_resident_state_append(st: *mut ecgpu_resident_state_t, field: c_int, data: *const u8, n_bytes: u64) -> c_int;
    pub fn ecgpu_resident_state_truncate(st: *mut ecgpu_resident_state_t, field: c_int, new_n_bytes: u64) -> c_int;
    pub fn ecgpu_resident_state_replace(st: *mut ecgpu_resident_state_t, field: c_int, data: *const u8, n_bytes: u64) -> c_int;
    pub fn ecgpu_resident_state_size(st: *const ecgpu_resident_state_t) -> u64;
    // field-addressed changes (round 6): queued in program order, applied at the next root
    pub fn ecgpu_resident_state_patch_field(st: *mut ecgpu_resident_state_t, field: u32, offset_in_field: u64, data: *const u8, n_bytes: u64) -> c_int;
    pub fn ecgpu_resident_state_patch_elements(st: *mut ecgpu_resident_state_t, field: u32, first_index: u64, data: *const u8, n_bytes: u64) -> c_int;
    pub fn ecgpu_resident_state_push(st: *mut ecgpu_resident_state_t, field: u32, data: *const u8, n_bytes: u64) -> c_int;
    pub fn ecgpu_resident_state_truncate_field(st: *mut ecgpu_resident_state_t, field: u32, new_n_bytes: u64) -> c_int;
    pub fn ecgpu_resident_state_set_field(st: *mut ecgpu_resident_state_t, field: u32, data: *const u8, n_bytes: u64) -> c_int;
    pub fn ecgpu_resident_state_add_validator(st: *mut ecgpu_resident_state_t, validator121: *const u8, balance: u64) -> c_int;
    pub fn ecgpu_resident_state_rotate_participation(st: *mut ecgpu_resident_state_t) -> c_int;
    pub fn ecgpu_resident_state_flush(st: *mut ecgpu_resident_state_t) -> c_int;
    pub fn ecgpu_resident_state_field_size(st: *mut ecgpu_resident_state_t, field: u32) -> i64;
    pub fn ecgpu_warmup(flags: u32) -> c_int;
    pub fn ecgpu_ssz_generalized_index(types: *const ecgpu_ssz_type, n_types: u32, fields: *const u32, n_field_refs: u32,
                                       root_type: u32, path: *const u64, path_len: u32, gindex: *mut u64) -> c_int;
    pub fn ecgpu_ssz_prove(types: *const ecgpu_ssz_type, n_types: u32, fields: *const u32, n_field_refs: u32, root_type: u32,
                           ssz: *const u8, n_bytes: u64, path: *const u64, path_len: u32, leaf: *mut u8, branch: *mut u8,
                           max_depth: u32, depth: *mut u32, gindex: *mut u64, root: *mut u8) -> c_int;
    pub fn ecgpu_merkle_proof(chunks: *const u8, n_chunks: u64, limit_chunks: u64, index: u64, branch: *mut u8) -> c_int;
    pub fn ecgpu_beacon_state_field_roots(fork: c_int, ssz: *const u8, n_bytes: u64, preset: c_int, roots: *mut u8, capacity: u32,
                                          n_fields: *mut u32, root: *mut u8) -> c_int;
    pub fn ecgpu_compute_shuffled_indices(indices: *const u64, n: u64, seed: *const u8, rounds: u32, out: *mut u64) -> c_int;

    pub fn ecgpu_verify(pk: *const u8, msg: *const u8, msg_len: usize, sig: *const u8) -> c_int;
    pub fn ecgpu_fast_aggregate_verify(pks48: *const u8, k: u32, msg: *const u8, msg_len: usize, sig: *const u8,
                                       eth_variant: c_int) -> c_int;
    pub fn ecgpu_aggregate_verify(pks48: *const u8, n_pks: u32, msgs: *const u8, msg_off: *const u64, n_msgs: u32,
                                  sig: *const u8) -> c_int;
    pub fn ecgpu_aggregate_sigs(sigs96: *const u8, n: u32, out: *mut u8) -> c_int;
    pub fn ecgpu_aggregate_pks(pks48: *const u8, n: u32, out: *mut u8) -> c_int;
    pub fn ecgpu_fast_aggregate_verify_batch(pks48: *const u8, pk_off: *const u32, msgs32: *const u8, sigs96: *const u8, n: u32,
                                             eth_variant: c_int, status_out: *mut u8) -> c_int;

    pub fn ecgpu_fast_aggregate_verify_batch_multi(devices: *const c_int, n_devices: u32, pks48: *const u8, pk_off: *const u32,
                                                   msgs32: *const u8, sigs96: *const u8, n: u32, eth_variant: c_int,
                                                   status_out: *mut u8) -> c_int;
    pub fn ecgpu_g1_msm(pks48: *const u8, scalars32: *const u8, n: u32, scalar_bits: u32, out48: *mut u8) -> c_int;
    pub fn ecgpu_g2_msm(sigs96: *const u8, scalars32: *const u8, n: u32, scalar_bits: u32, out96: *mut u8) -> c_int;

    pub fn ecgpu_registry_create(capacity: u64, out: *mut *mut ecgpu_registry_t) -> c_int;
    pub fn ecgpu_registry_destroy(reg: *mut ecgpu_registry_t);
    pub fn ecgpu_registry_set(reg: *mut ecgpu_registry_t, first_index: u64, pks48: *const u8, n: u64) -> c_int;
    pub fn ecgpu_fast_aggregate_verify_indexed_batch(reg: *const ecgpu_registry_t, indices: *const u32, idx_off: *const u32,
                                                     msgs32: *const u8, sigs96: *const u8, n: u32, eth_variant: c_int,
                                                     status_out: *mut u8) -> c_int;

    pub fn ecgpu_batch_create(reg: *const ecgpu_registry_t, out: *mut *mut ecgpu_batch_t) -> c_int;
    pub fn ecgpu_batch_destroy(b: *mut ecgpu_batch_t);
    pub fn ecgpu_batch_push(b: *mut ecgpu_batch_t, pks48: *const u8, k: u32, msg: *const u8, msg_len: usize, sig: *const u8,
                            eth_variant: c_int) -> i64;
    pub fn ecgpu_batch_push_indexed(b: *mut ecgpu_batch_t, indices: *const u32, k: u32, msg: *const u8, msg_len: usize,
                                    sig: *const u8, eth_variant: c_int) -> i64;
    pub fn ecgpu_batch_len(b: *const ecgpu_batch_t) -> u32;
    pub fn ecgpu_batch_flush(b: *mut ecgpu_batch_t, status_out: *mut u8, capacity: u32) -> c_int;
}
