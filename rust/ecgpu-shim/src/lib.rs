//! `ethereum_consensus::crypto::bls` on the MI355X backend: the same free functions, argument types reduced to the byte
//! arrays the reference's `PublicKey(ByteVector<48>)` / `Signature(ByteVector<96>)` wrap (crypto/bls.rs:239,290), the same
//! `Error` variants for the same inputs (crypto/bls.rs:27-62), and the same evaluation order (keys left to right, then the
//! signature, then blst's verify).  `rust/patches/ethereum-consensus-gpu-feature.patch` shows the ten-line glue that makes
//! `crypto/bls.rs` call these behind `--features gpu`; no caller changes.
//!
//! There is NO CPU fallback: a backend fault (no gfx950 device, HIP error) is a panic, exactly like an allocation failure
//! inside blst would be.
pub mod sys;

use std::ffi::CStr;
use std::os::raw::c_int;
use thiserror::Error;

pub const BLS_PUBLIC_KEY_BYTES_LEN: usize = 48;
pub const BLS_SIGNATURE_BYTES_LEN: usize = 96;
pub type PublicKeyBytes = [u8; BLS_PUBLIC_KEY_BYTES_LEN];
pub type SignatureBytes = [u8; BLS_SIGNATURE_BYTES_LEN];
pub type Bytes32 = [u8; 32];

/// crypto/bls.rs:44-62
#[derive(Debug, Error, PartialEq, Eq)]
#[error("{0}")]
pub struct BLSTError(pub String);

impl BLSTError {
    /// `impl From<BLST_ERROR> for BLSTError` (crypto/bls.rs:48-62) over the numeric code
    fn from_code(code: c_int) -> Self {
        let inner = match code {
            1 => "bad encoding",
            2 => "point not on curve",
            3 => "point not in group",
            4 => "aggregation type mismatch",
            5 => "verification failed",
            6 => "public key is infinity",
            7 => "bad scalar input",
            _ => unreachable!("do not create a BLSTError from a success"),
        };
        Self(inner.to_string())
    }
}

/// crypto/bls.rs:27-42 (the variants these functions can produce)
#[derive(Debug, Error, PartialEq, Eq)]
pub enum Error {
    #[error("inputs required for aggregation but none were provided")]
    EmptyAggregate,
    #[error("blst error: {0}")]
    BLST(#[from] BLSTError),
    #[error("invalid signature")]
    InvalidSignature,
}

fn backend_fault(rc: c_int) -> ! {
    let msg = unsafe {
        let p = sys::ecgpu_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    panic!("ecgpu backend fault {rc}: {msg}");
}

/// Status of a verify function -> the reference's `Result` (include/ecgpu.h, "Error identity"): 1, 2, 3, 6 were raised while
/// CONVERTING a key or the signature (`TryFrom`, crypto/bls.rs:69-70,100-105,119-125) -> `Error::BLST`; everything else that
/// is not success came out of blst's verify call -> `Error::InvalidSignature` (crypto/bls.rs:72-76,107-111,127-131); that
/// includes 0x43 / 0x46, the group / infinity conditions found INSIDE verify.
pub fn verify_status_to_result(rc: c_int) -> Result<(), Error> {
    match rc {
        0 => Ok(()),
        1 | 2 | 3 | 6 => Err(BLSTError::from_code(rc).into()),
        rc if rc < 0 => backend_fault(rc),
        _ => Err(Error::InvalidSignature),
    }
}

/// aggregate / eth_aggregate_public_keys: every BLST_ERROR is `Error::BLST` there (crypto/bls.rs:92,147)
fn aggregate_status_to_result(rc: c_int) -> Result<(), Error> {
    match rc {
        0 => Ok(()),
        sys::ECGPU_EMPTY_AGGREGATE => Err(Error::EmptyAggregate),
        rc if rc < 0 => backend_fault(rc),
        rc => Err(BLSTError::from_code(rc & 7).into()),
    }
}

fn concat48(keys: &[&PublicKeyBytes]) -> Vec<u8> {
    let mut out = Vec::with_capacity(48 * keys.len());
    for k in keys {
        out.extend_from_slice(&k[..]);
    }
    out
}

/// crypto/bls.rs:12-20
pub fn hash<D: AsRef<[u8]>>(data: D) -> Bytes32 {
    let d = data.as_ref();
    let mut out = [0u8; 32];
    let rc = unsafe { sys::ecgpu_sha256(d.as_ptr(), d.len(), out.as_mut_ptr()) };
    if rc != 0 {
        backend_fault(rc);
    }
    out
}

/// crypto/bls.rs:64-77
pub fn verify_signature(public_key: &PublicKeyBytes, msg: &[u8], signature: &SignatureBytes) -> Result<(), Error> {
    verify_status_to_result(unsafe { sys::ecgpu_verify(public_key.as_ptr(), msg.as_ptr(), msg.len(), signature.as_ptr()) })
}

/// crypto/bls.rs:79-93
pub fn aggregate(signatures: &[SignatureBytes]) -> Result<SignatureBytes, Error> {
    if signatures.is_empty() {
        return Err(Error::EmptyAggregate);
    }
    let mut flat = Vec::with_capacity(96 * signatures.len());
    for s in signatures {
        flat.extend_from_slice(&s[..]);
    }
    let mut out = [0u8; 96];
    aggregate_status_to_result(unsafe { sys::ecgpu_aggregate_sigs(flat.as_ptr(), signatures.len() as u32, out.as_mut_ptr()) })?;
    Ok(out)
}

/// crypto/bls.rs:95-112
pub fn aggregate_verify(public_keys: &[PublicKeyBytes], msgs: &[&[u8]], signature: &SignatureBytes) -> Result<(), Error> {
    let mut pks = Vec::with_capacity(48 * public_keys.len());
    for k in public_keys {
        pks.extend_from_slice(&k[..]);
    }
    let mut flat = Vec::new();
    let mut off = Vec::with_capacity(msgs.len() + 1);
    off.push(0u64);
    for m in msgs {
        flat.extend_from_slice(m);
        off.push(flat.len() as u64);
    }
    verify_status_to_result(unsafe {
        sys::ecgpu_aggregate_verify(pks.as_ptr(), public_keys.len() as u32, flat.as_ptr(), off.as_ptr(), msgs.len() as u32,
                                    signature.as_ptr())
    })
}

/// crypto/bls.rs:114-132
pub fn fast_aggregate_verify(public_keys: &[&PublicKeyBytes], msg: &[u8], signature: &SignatureBytes) -> Result<(), Error> {
    let pks = concat48(public_keys);
    verify_status_to_result(unsafe {
        sys::ecgpu_fast_aggregate_verify(pks.as_ptr(), public_keys.len() as u32, msg.as_ptr(), msg.len(), signature.as_ptr(), 0)
    })
}

/// crypto/bls.rs:135-148
pub fn eth_aggregate_public_keys(public_keys: &[PublicKeyBytes]) -> Result<PublicKeyBytes, Error> {
    if public_keys.is_empty() {
        return Err(Error::EmptyAggregate);
    }
    let mut flat = Vec::with_capacity(48 * public_keys.len());
    for k in public_keys {
        flat.extend_from_slice(&k[..]);
    }
    let mut out = [0u8; 48];
    aggregate_status_to_result(unsafe { sys::ecgpu_aggregate_pks(flat.as_ptr(), public_keys.len() as u32, out.as_mut_ptr()) })?;
    Ok(out)
}

/// crypto/bls.rs:150-160 (the `public_keys.is_empty() && signature.is_infinity()` rule is inside the backend: eth_variant = 1)
pub fn eth_fast_aggregate_verify(public_keys: &[&PublicKeyBytes], message: &[u8], signature: &SignatureBytes) -> Result<(), Error> {
    let pks = concat48(public_keys);
    verify_status_to_result(unsafe {
        sys::ecgpu_fast_aggregate_verify(pks.as_ptr(), public_keys.len() as u32, message.as_ptr(), message.len(),
                                         signature.as_ptr(), 1)
    })
}

/// Whole-block batching: defer every verification of a block, verify all of them in one pass of the
/// GPU pipeline, get back per call exactly the `Result` the scalar function returns.  `process_block` pushes where it used
/// to verify (phase0/state_transition.rs:56, phase0/block_processing.rs:649,752-761, altair/block_processing.rs:226-234)
/// and checks the results once, after the last operation.
pub struct SignatureBatch {
    raw: *mut sys::ecgpu_batch_t,
}
// the handle is internally locked (include/ecgpu.h)
unsafe impl Send for SignatureBatch {}
unsafe impl Sync for SignatureBatch {}

impl SignatureBatch {
    pub fn new() -> Self {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sys::ecgpu_batch_create(std::ptr::null(), &mut raw) };
        if rc != 0 {
            backend_fault(rc);
        }
        Self { raw }
    }
    fn pushed(rc: i64) -> usize {
        if rc < 0 {
            backend_fault(rc as c_int);
        }
        rc as usize
    }
    /// position of the deferred `verify_signature` call in the batch
    pub fn verify_signature(&self, public_key: &PublicKeyBytes, msg: &[u8], signature: &SignatureBytes) -> usize {
        Self::pushed(unsafe { sys::ecgpu_batch_push(self.raw, public_key.as_ptr(), 1, msg.as_ptr(), msg.len(), signature.as_ptr(), 0) })
    }
    pub fn fast_aggregate_verify(&self, public_keys: &[&PublicKeyBytes], msg: &[u8], signature: &SignatureBytes) -> usize {
        let pks = concat48(public_keys);
        Self::pushed(unsafe {
            sys::ecgpu_batch_push(self.raw, pks.as_ptr(), public_keys.len() as u32, msg.as_ptr(), msg.len(), signature.as_ptr(), 0)
        })
    }
    pub fn eth_fast_aggregate_verify(&self, public_keys: &[&PublicKeyBytes], msg: &[u8], signature: &SignatureBytes) -> usize {
        let pks = concat48(public_keys);
        Self::pushed(unsafe {
            sys::ecgpu_batch_push(self.raw, pks.as_ptr(), public_keys.len() as u32, msg.as_ptr(), msg.len(), signature.as_ptr(), 1)
        })
    }
    pub fn len(&self) -> usize {
        unsafe { sys::ecgpu_batch_len(self.raw) as usize }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
    /// verifies and empties the batch; `results[p]` is what the scalar call pushed at position p would have returned
    pub fn flush(&self) -> Vec<Result<(), Error>> {
        let n = self.len();
        let mut st = vec![0u8; n.max(1)];
        let rc = unsafe { sys::ecgpu_batch_flush(self.raw, st.as_mut_ptr(), n as u32) };
        if rc != 0 {
            backend_fault(rc);
        }
        st[..n].iter().map(|&s| verify_status_to_result(s as c_int)).collect()
    }
}
impl Default for SignatureBatch {
    fn default() -> Self {
        Self::new()
    }
}
impl Drop for SignatureBatch {
    fn drop(&mut self) {
        unsafe { sys::ecgpu_batch_destroy(self.raw) }
    }
}

/// Merkleization entry points for the `ssz_rs` fork (rust/patches/ssz-rs-ecgpu.patch): `merkleize` / `mix_in_length`
/// (what every derived `HashTreeRoot` bottoms out in) and the whole-object shortcuts.
pub mod merkle {
    use super::{backend_fault, sys, Bytes32};

    #[derive(Debug, PartialEq, Eq)]
    pub enum MerkleizationError {
        /// ssz_rs `MerkleizationError::InputExceedsLimit`
        InputExceedsLimit(usize),
        /// a malformed serialization handed to a whole-object entry point
        InvalidEncoding,
    }

    fn finish(rc: i32, root: Bytes32, limit: usize) -> Result<Bytes32, MerkleizationError> {
        match rc {
            0 => Ok(root),
            -3 => Err(if limit != 0 { MerkleizationError::InputExceedsLimit(limit) } else { MerkleizationError::InvalidEncoding }),
            rc => backend_fault(rc),
        }
    }
    /// `merkleize(chunks, limit)`; `mix_in_length = Some(len)` for lists
    pub fn merkleize(packed: &[u8], limit_chunks: Option<usize>, mix_in_length: Option<usize>) -> Result<Bytes32, MerkleizationError> {
        let mut root = [0u8; 32];
        let rc = unsafe {
            sys::ecgpu_merkleize(packed.as_ptr(), packed.len() as u64, limit_chunks.unwrap_or(0) as u64, mix_in_length.is_some() as i32,
                                 mix_in_length.unwrap_or(0) as u64, root.as_mut_ptr())
        };
        finish(rc, root, limit_chunks.unwrap_or(0))
    }
    /// `List<Validator, LIMIT>::hash_tree_root` from the packed 121-byte records (phase0/validator.rs:10-26)
    pub fn validators_root(ssz121: &[u8], limit: usize) -> Result<Bytes32, MerkleizationError> {
        let mut root = [0u8; 32];
        let rc = unsafe { sys::ecgpu_htr_validators(ssz121.as_ptr(), (ssz121.len() / 121) as u64, limit as u64, root.as_mut_ptr()) };
        finish(rc, root, limit)
    }
    /// `BeaconState::hash_tree_root` of fork 0..=4 (phase0..deneb) from its serialization; preset 0 = mainnet, 1 = minimal
    pub fn beacon_state_root(fork: i32, preset: i32, ssz: &[u8]) -> Result<Bytes32, MerkleizationError> {
        let mut root = [0u8; 32];
        let rc = unsafe { sys::ecgpu_htr_beacon_state(fork, ssz.as_ptr(), ssz.len() as u64, preset, root.as_mut_ptr()) };
        finish(rc, root, 0)
    }
    /// any derived container from its serialization and its type table (the derive macro emits the table)
    pub fn hash_tree_root(types: &[sys::ecgpu_ssz_type], fields: &[u32], root_type: u32, ssz: &[u8]) -> Result<Bytes32, MerkleizationError> {
        let mut root = [0u8; 32];
        let rc = unsafe {
            sys::ecgpu_htr_ssz(types.as_ptr(), types.len() as u32, fields.as_ptr(), fields.len() as u32, root_type, ssz.as_ptr(),
                               ssz.len() as u64, root.as_mut_ptr())
        };
        finish(rc, root, 0)
    }
    /// ssz_rs `is_valid_merkle_branch` (phase0/block_processing.rs:433, deneb/blob_sidecar.rs:62)
    pub fn is_valid_merkle_branch(leaf: &Bytes32, branch: &[Bytes32], depth: usize, index: usize, root: &Bytes32) -> bool {
        if branch.len() < depth || depth > 64 {
            return false;
        }
        let flat: Vec<u8> = branch[..depth].iter().flat_map(|n| n.iter().copied()).collect();
        match unsafe { sys::ecgpu_is_valid_merkle_branch(leaf.as_ptr(), flat.as_ptr(), depth as u32, index as u64, root.as_ptr()) } {
            0 => true,
            rc if rc < 0 => backend_fault(rc),
            _ => false,
        }
    }
}

#[cfg(test)]
mod tests {
    //! run on a box with an MI355X: `ECGPU_LIB_DIR=... cargo test`
    use super::*;

    // crypto/bls.rs:530-544 `test_can_sign`: pk of sk 0x4009..3d50 (derived, pinned in tests/test_oracle_bls.py)
    const PK: &str = "a3843eddcff557c1d9cc39b165688a8211979cef3679ef7c79751023dce64396f9ae6b86fa7b1fa15b9041d71dde7614";
    const SIG: &str = "a01e49276730e4752eef31b0570c8707de501398dac70dd144438cd1bd05fb9b9bb3e1a9ceef0a68cc08904362cafa3f1005e5b699a41847fff6f5552260468846de5bdbf94a9aedeb29bc6cdb2c1d34922d9e9af4c0593a69ae978a90b5aba6";

    fn unhex<const N: usize>(s: &str) -> [u8; N] {
        let mut out = [0u8; N];
        for i in 0..N {
            out[i] = u8::from_str_radix(&s[2 * i..2 * i + 2], 16).unwrap();
        }
        out
    }

    #[test]
    fn reference_vector_and_error_variants() {
        let pk: PublicKeyBytes = unhex(PK);
        let sig: SignatureBytes = unhex(SIG);
        let msg = b"blst is such a blast";
        assert_eq!(verify_signature(&pk, msg, &sig), Ok(()));
        assert_eq!(verify_signature(&pk, b"another message", &sig), Err(Error::InvalidSignature));
        assert_eq!(fast_aggregate_verify(&[&pk], msg, &sig), Ok(()));
        assert_eq!(fast_aggregate_verify(&[], msg, &sig), Err(Error::InvalidSignature)); // AGGR_TYPE_MISMATCH inside verify
        let mut inf_sig = [0u8; 96];
        inf_sig[0] = 0xc0;
        assert_eq!(eth_fast_aggregate_verify(&[], msg, &inf_sig), Ok(()));
        assert_eq!(verify_signature(&[0u8; 48], msg, &sig), Err(Error::BLST(BLSTError("bad encoding".into()))));
        let mut inf_pk = [0u8; 48];
        inf_pk[0] = 0xc0;
        assert_eq!(verify_signature(&inf_pk, msg, &sig), Err(Error::BLST(BLSTError("public key is infinity".into()))));
        assert_eq!(aggregate(&[]), Err(Error::EmptyAggregate));
        let batch = SignatureBatch::new();
        batch.verify_signature(&pk, msg, &sig);
        batch.verify_signature(&pk, b"x", &sig);
        assert_eq!(batch.flush(), vec![Ok(()), Err(Error::InvalidSignature)]);
    }
}
