//! Times the unmodified reference (`ethereum_consensus::crypto::fast_aggregate_verify` over blst,
//! `hash_tree_root` over ssz_rs + sha2) on the fixtures of `tools/make_reference_fixtures.py`: the same
//! tuples / the same state `bench.py` feeds the MI355X backend.  Single thread, like the reference's callers.
//!
//!   reference_rs bls   <tuples_N.bin>     records of 48 (public key) + 32 (message) + 96 (signature) bytes;
//!                                         every 64th record carries a message that was not signed
//!   reference_rs state <state.ssz>        SSZ encoding of a deneb mainnet BeaconState
use ethereum_consensus::crypto::{fast_aggregate_verify, PublicKey, Signature};
use ethereum_consensus::deneb::mainnet::BeaconState;
use ssz_rs::prelude::*;
use std::{env, fs, time::Instant};

fn bls(path: &str) {
    let data = fs::read(path).expect("fixture file");
    const REC: usize = 48 + 32 + 96;
    assert!(data.len() % REC == 0, "not a whole number of records");
    let n = data.len() / REC;
    // decoding the byte strings is not timed: the reference's types hold bytes and convert to blst points inside verify
    let recs: Vec<(PublicKey, [u8; 32], Signature)> = data
        .chunks_exact(REC)
        .map(|r| {
            let pk = PublicKey::try_from(&r[..48]).expect("48-byte key");
            let mut msg = [0u8; 32];
            msg.copy_from_slice(&r[48..80]);
            let sig = Signature::try_from(&r[80..]).expect("96-byte signature");
            (pk, msg, sig)
        })
        .collect();
    let start = Instant::now();
    let mut failures = 0usize;
    for (pk, msg, sig) in &recs {
        if fast_aggregate_verify(&[pk], msg.as_ref(), sig).is_err() {
            failures += 1;
        }
    }
    let dt = start.elapsed().as_secs_f64();
    println!(
        "{{\"metric\": \"bls_signatures_verified_per_sec\", \"value\": {:.1}, \"unit\": \"sigs/s\", \"tuples\": {}, \"failures\": {}, \"expected_failures\": {}, \"seconds\": {:.3}, \"cores\": 1, \"kind\": \"reference\"}}",
        n as f64 / dt,
        n,
        failures,
        (n + 63) / 64,
        dt
    );
}

fn state(path: &str) {
    let data = fs::read(path).expect("fixture file");
    let mut st = <BeaconState as ssz_rs::Deserialize>::deserialize(&data).expect("deneb mainnet BeaconState");
    let n = st.validators.len();
    let reps = 3;
    let mut best = f64::MAX;
    let mut root = Node::default();
    for _ in 0..reps {
        let start = Instant::now();
        root = st.hash_tree_root().expect("merkleization");
        best = best.min(start.elapsed().as_secs_f64());
    }
    println!(
        "{{\"metric\": \"state_roots_per_sec\", \"value\": {:.4}, \"seconds_per_root\": {:.4}, \"validators\": {}, \"root\": \"0x{}\", \"cores\": 1, \"kind\": \"reference\"}}",
        1.0 / best,
        best,
        n,
        hex::encode(root.as_ref() as &[u8])
    );
}

fn main() {
    let args: Vec<String> = env::args().collect();
    match (args.get(1).map(String::as_str), args.get(2)) {
        (Some("bls"), Some(p)) => bls(p),
        (Some("state"), Some(p)) => state(p),
        _ => eprintln!("usage: reference_rs bls <tuples.bin> | state <state.ssz>"),
    }
}
