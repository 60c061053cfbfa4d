//! One declaration per entry point of include/ecgpu.h that the shim uses (same order as the header).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const ECGPU_SUCCESS: c_int = 0;
pub const ECGPU_IN_VERIFY: c_int = 0x40;
pub const ECGPU_EMPTY_AGGREGATE: c_int = -100;

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
