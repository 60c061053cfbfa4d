/* ecgpu.h -- C ABI of the MI355X (gfx950) batch-crypto backend for ralexstokes/ethereum_consensus.
 *
 * The reference has no FFI for these paths: `crypto::bls` calls `blst::min_pk` directly
 * (/root/reference/ethereum-consensus/src/crypto/bls.rs:4) and Merkleization is the
 * `ssz_rs::HashTreeRoot` trait (ssz/mod.rs:4-7).  This header is the boundary a maintainer binds
 * with `extern "C"` in a `gpu` cargo feature of `crypto/bls.rs` and in a `[patch]`ed ssz_rs
 * (INTEGRATION.md shows both stubs).  Each entry point names the reference interface it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; all buffers caller-owned; nothing allocated by the library is
 *     handed to the caller; thread-safe (per-thread HIP stream + workspace).
 *   - return value: 0..7 = blst BLST_ERROR numbering (so the Rust shim can rebuild
 *     `BLSTError` strings, crypto/bls.rs:48-62), 0x43 / 0x46 = the two codes that blst's verify
 *     call (not a conversion) produced, see ECGPU_IN_VERIFY; negative = backend fault.  There is NO CPU
 *     fallback: without a usable gfx950 device every call returns ECGPU_ERR_NO_DEVICE.
 *   - "host" entry points take host memory and copy; "_dev" entry points take device pointers
 *     (already resident in HBM) plus the HIP stream to enqueue on (NULL = the library's per-thread
 *     stream) and are asynchronous: results are valid after the stream is synchronized.
 */
#ifndef ECGPU_H
#define ECGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#include "ecgpu_status.h" /* the status codes: BLST_ERROR numbering, ECGPU_IN_VERIFY, backend faults */

typedef void* ecgpu_stream_t; /* a hipStream_t */

/* ---- lifecycle ------------------------------------------------------------------------------ */
/* Bind the calling process to HIP device `device` (-1: current device) and build the device
 * tables (zero-hash ladder, BLS constants).  Idempotent. */
int ecgpu_init(int device);
int ecgpu_device_count(void);
/* Several GPUs in one process (a Rust host has no torch.distributed): bind the calling host thread to `device`; every later
 * call of this thread runs there (tables, streams and workspaces exist per device; a registry / resident state / batch
 * belongs to the device of the thread that created it).  The *_multi entries below do this on threads of their own. */
int ecgpu_bind_thread(int device);
/* the device the calling thread's calls run on (ecgpu_init / ecgpu_bind_thread), or ECGPU_ERR_NO_DEVICE before ecgpu_init: a
 * multi-process host checks it against its own LOCAL_RANK before the first collective (bench.py multi_gpu_preflight) */
int ecgpu_thread_device(void);
/* Warm-up.  The first call of a process pays for the box self-check, the upload of the small-batch programs, the calling
 * thread's stream sets and arenas and the code objects of the kernels it launches: ~45 ms against ~2 ms for a warm
 * verify_signature.  ecgpu_warmup makes real calls with the reference's fixed vector (crypto/bls.rs:530-544) on the calling
 * thread -- ECGPU_WARM_BLS: one verify_signature (the small-batch path: what a block's scalar calls use); ECGPU_WARM_BLS_BATCHES:
 * also every larger dispatch class, timed -- the crossovers between the kernel sets are placed where THIS device puts them
 * (ecgpu_bls_dispatch_thresholds; ~0.3 s in all; for hosts that verify blocks and epochs); ECGPU_WARM_MERKLE: one header root.
 * flags == 0 means BLS | MERKLE.  Process-wide state is warm for every thread afterwards; the per-thread part (streams, arenas:
 * ~1 ms) is paid by each thread's own first call.  Returns 0, or a negative code -- ECGPU_ERR_HIP when the fixed vector does
 * not verify (a broken build or device). */
#define ECGPU_WARM_BLS 1u
#define ECGPU_WARM_BLS_BATCHES 2u
#define ECGPU_WARM_MERKLE 4u
int ecgpu_warmup(unsigned flags);
const char* ecgpu_version(void);
const char* ecgpu_last_error(void); /* thread-local description of the last negative return */

/* ---- SHA-256 / SSZ Merkleization ------------------------------------------------------------ */
/* crypto::hash (crypto/bls.rs:12-20): SHA-256 of an arbitrary byte string. */
int ecgpu_sha256(const uint8_t* data, size_t len, uint8_t out[32]);
/* n independent SHA-256 of equal-length messages (msg i at data + i*len) -> out + 32*i. */
int ecgpu_sha256_batch(const uint8_t* data, size_t len, uint64_t n, uint8_t* out);

/* ssz_rs `merkleize(chunks, limit)` [+ `mix_in_length`]:
 * `data` = packed bytes (`pack`: zero-padded to 32-byte chunks), limit_chunks = chunk limit of the
 * SSZ type (0 = tight: next_pow2(n_chunks)), mix_in_len != 0 adds mix_in_length(root, len). */
int ecgpu_merkleize(const uint8_t* data, uint64_t n_bytes, uint64_t limit_chunks, int mix_in_len,
                    uint64_t len, uint8_t root[32]);
int ecgpu_merkleize_dev(const uint8_t* d_data, uint64_t n_bytes, uint64_t limit_chunks,
                        int mix_in_len, uint64_t len, uint8_t* d_root, ecgpu_stream_t stream);

/* hash_tree_root(List<Validator, limit>) from n packed 121-byte SSZ Validator records
 * (phase0/validator.rs:10-26; `validators` field, phase0/beacon_state.rs:74). */
int ecgpu_htr_validators(const uint8_t* ssz121, uint64_t n, uint64_t limit, uint8_t root[32]);
int ecgpu_htr_validators_dev(const uint8_t* d_ssz121, uint64_t n, uint64_t limit, uint8_t* d_root,
                             ecgpu_stream_t stream);
/* Multi-GPU sharding of one big list: rank g owns the aligned subtree of `width` (a power of two)
 * leaves starting at g * width.  ecgpu_validators_subtree_root is that subtree's root for validator records --
 * `merkleize(validator roots, width)` without the length mix-in; packed basic lists (balances, participation) use
 * ecgpu_merkleize[_dev] with limit_chunks = width and mix_in_len = 0.  The ranks all-gather their 32-byte sub-roots and
 * each finishes with ecgpu_merkleize_subtree_roots: n_sub (<= 512) nodes entering at level log2(width) -- odd tails pair
 * with the zero hashes of that level upwards --, climbed to the `limit`-leaf root, then mix_in_length(len). */
int ecgpu_validators_subtree_root(const uint8_t* ssz121, uint64_t n, uint64_t width, uint8_t root[32]);
int ecgpu_validators_subtree_root_dev(const uint8_t* d_ssz121, uint64_t n, uint64_t width, uint8_t* d_root,
                                      ecgpu_stream_t stream);
int ecgpu_merkleize_subtree_roots(const uint8_t* sub_roots, uint32_t n_sub, uint64_t width, uint64_t limit, int mix_in_len,
                                  uint64_t len, uint8_t root[32]);
int ecgpu_merkleize_subtree_roots_dev(const uint8_t* d_sub_roots, uint32_t n_sub, uint64_t width, uint64_t limit,
                                      int mix_in_len, uint64_t len, uint8_t* d_root, ecgpu_stream_t stream);

/* hash_tree_root(List<Validator, limit>) with the registry sharded over `n_devices` GPUs of this process: aligned subtrees
 * on one host thread per device, sub-roots exchanged through host memory, top of the tree on devices[0]. */
int ecgpu_htr_validators_multi(const int* devices, uint32_t n_devices, const uint8_t* ssz121, uint64_t n, uint64_t limit,
                               uint8_t root[32]);

/* hash_tree_root(BeaconBlockHeader) from its 112-byte SSZ encoding (phase0/beacon_block.rs:83-91;
 * called at phase0/slot_processing.rs:75, block_processing.rs:579). */
int ecgpu_htr_beacon_block_header(const uint8_t ssz112[112], uint8_t root[32]);

/* compute_signing_root (signing.rs:14-22): htr(SigningData{object_root, domain}). */
int ecgpu_signing_root(const uint8_t object_root[32], const uint8_t domain[32], uint8_t root[32]);

/* ssz_rs `is_valid_merkle_branch` (used at phase0/block_processing.rs:433,
 * deneb/blob_sidecar.rs:62).  Returns 0 (valid) or ECGPU_VERIFY_FAIL. */
int ecgpu_is_valid_merkle_branch(const uint8_t leaf[32], const uint8_t* branch, uint32_t depth,
                                 uint64_t index, const uint8_t root[32]);

/* hash_tree_root(BeaconState) for the deneb fork (deneb/beacon_state.rs:13-64) from the SSZ
 * serialization of the state; called per slot (phase0/slot_processing.rs:67) and per block
 * (phase0/state_transition.rs:60).  preset: 0 = mainnet, 1 = minimal. */
#define ECGPU_PRESET_MAINNET 0
#define ECGPU_PRESET_MINIMAL 1
int ecgpu_htr_beacon_state_deneb(const uint8_t* ssz, uint64_t n_bytes, int preset, uint8_t root[32]);
/* device-resident state bytes; `h_fixed` = host copy of the fixed-size part of the encoding
 * (the first ecgpu_beacon_state_deneb_fixed_size(preset) bytes: offsets and small fields).  Asynchronous: the root appears
 * at d_root in stream order.  Everything the host entries reject is rejected here at the call too (ECGPU_ERR_BAD_ARG), with one
 * exception: the `extra_data` offset word INSIDE the payload header (bellatrix+) lives in the variable part, which the
 * host never sees -- it is compared on the device, and an encoding that fails the comparison gets the root 0xFF x 32 (no
 * SHA-256 output a caller will ever meet) instead of an error code. */
int ecgpu_htr_beacon_state_deneb_dev(const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed,
                                     int preset, uint8_t* d_root, ecgpu_stream_t stream);
uint64_t ecgpu_beacon_state_deneb_fixed_size(int preset);
/* The same for every fork the reference defines up to deneb: phase0/beacon_state.rs:50-88 (21 fields),
 * altair/beacon_state.rs:13-55 (24), bellatrix/beacon_state.rs:13-58 (25), capella/beacon_state.rs:13-64 (28),
 * deneb/beacon_state.rs:13-64 (28).  The host-pointer entry takes any fork.  phase0 states hold two lists of variable-size
 * elements (PendingAttestation) whose offset tables live in the encoding itself: ecgpu_htr_beacon_state_dev(phase0) copies those
 * two lists -- the tail of the encoding, KBs to ~1 MB -- back to the host once (ONE synchronisation of the stream; every other
 * fork is fully asynchronous) and plans them there; the checked / sharded forms start at altair.  A RESIDENT phase0 state
 * (round 5) never copies them back: the two lists are rooted where the host hands them over (at creation, and in
 * ecgpu_resident_state_replace), and the state root takes the two nodes as they are. */
#define ECGPU_FORK_PHASE0 0
#define ECGPU_FORK_ALTAIR 1
#define ECGPU_FORK_BELLATRIX 2
#define ECGPU_FORK_CAPELLA 3
#define ECGPU_FORK_DENEB 4
/* electra as this revision of the reference defines it (electra/beacon_state.rs:73-145: 37 fields, three lists of pending
 * operations; electra/execution_payload.rs:54-84: a 19-field payload header): the host-pointer entry, the _dev entries and
 * the sharded form and, since round 5, resident states. */
#define ECGPU_FORK_ELECTRA 5
int ecgpu_htr_beacon_state(int fork, const uint8_t* ssz, uint64_t n_bytes, int preset, uint8_t root[32]);
int ecgpu_htr_beacon_state_dev(int fork, const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset,
                               uint8_t* d_root, ecgpu_stream_t stream);
uint64_t ecgpu_beacon_state_fixed_size(int fork, int preset);
/* ecgpu_htr_beacon_state_dev with the device-side check made visible: *d_status (device memory, written in stream order next
 * to the root) = 0, or ECGPU_ERR_BAD_ARG when the payload header's extra_data offset word -- the one part of the encoding the
 * host never sees -- is not what the reference's deserializer accepts.  The root is still poisoned (0xFF x 32) in that case. */
int ecgpu_htr_beacon_state_dev_checked(int fork, const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset,
                                       uint8_t* d_root, int32_t* d_status, ecgpu_stream_t stream);

/* ONE BeaconState over several GPUs (the reference's single call is `state.hash_tree_root()`,
 * phase0/slot_processing.rs:67).  The five registry-sized lists -- validators, balances, previous / current epoch
 * participation, inactivity_scores, 99 % of the hash64 -- are cut into aligned power-of-two subtrees, one per rank
 * (width = the smallest power of two with width * world >= leaves); everything else is computed by every rank.
 *   phase A  ecgpu_beacon_state_shard_subroots_dev: rank `rank` of `world` reduces its subtree of each list from the
 *            device-resident encoding (only its own byte ranges of the five lists are read) -> d_subroots: 5 x 32 bytes.
 *            d_field_roots (64 x 32 bytes, may be NULL): the rank ALSO computes every other field of the state in the same
 *            launches -- underneath its validator pass -- and leaves the field roots there for phase B;
 *   exchange all-gather the 160 bytes over the ranks (rank-major: d_all[rank][list]) -- the path's only collective;
 *   phase B  ecgpu_htr_beacon_state_sharded_dev: finishes the five lists from the gathered nodes (zero ladder to the list
 *            limit, length mix-in) and hashes the state container.  d_field_roots = what phase A left (then that is ALL it
 *            hashes: ~30 dependent hash64 between the exchange and the root), or NULL: it computes the remaining fields
 *            itself.  world == 1 gives the plain root.
 * fork >= altair, like the other device-resident forms. */
int ecgpu_beacon_state_shard_subroots_dev(int fork, const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset,
                                          uint32_t rank, uint32_t world, uint8_t* d_subroots, uint8_t* d_field_roots,
                                          ecgpu_stream_t stream);
int ecgpu_htr_beacon_state_sharded_dev(int fork, const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset,
                                       const uint8_t* d_all_subroots, uint32_t world, const uint8_t* d_field_roots,
                                       uint8_t* d_root, ecgpu_stream_t stream);
uint32_t ecgpu_beacon_state_shard_lists(void); /* 5: nodes per rank in the exchange */
/* number of hash64 the last state root of this thread performed (work accounting for benches) */
uint64_t ecgpu_last_hash64_count(void);

/* Device-resident state (BASELINE configs[4]: "state root after mutating ~2^12 balances + participation bytes per
 * slot").  The reference keeps the state in host memory and re-Merkleizes all of it every slot
 * (phase0/slot_processing.rs:67); shipping 148 MB of serialization over PCIe per slot would cost more than hashing
 * it.  A resident state is uploaded once; afterwards only the bytes a block changed travel: `patch` overwrites byte
 * ranges of the encoding in place (same total length: field values, balances, participation flags, roots ...),
 * `root` re-Merkleizes on the device; lists change length through `append` / `truncate` below.
 * What stays on the device (csrc/state_tree.h): next to the encoding, EVERY interior node of the tree
 * of every big field (the registry, balances, participation, inactivity scores, the root vectors, randao mixes ...).  A
 * patch / append marks the level-0 entries its bytes belong to; `root` re-hashes the paths above the marked entries and
 * nothing else (one climb launch), then one finishing job per field and the small fields: ~28 k hash64 for a slot's 4 096
 * balance + 4 096 participation writes on a 2^20-validator state instead of 10.1 M.  A field rewritten wholesale (an epoch's
 * balances) or whose tree height changed is rebuilt level by level at the next root.
 * Threads, streams, ordering.  One resident state must not be used from two host threads AT ONCE; it may be handed from one
 * thread to another.  patch / append / truncate / replace and the field-addressed entries below run on the calling thread's
 * own stream; root_dev runs on the caller's stream.  Each of them first waits (hipStreamWaitEvent, no host block) for the last
 * change and the last root enqueued on any OTHER stream, so the effects of the calls on one state are always those of the
 * order in which the host issued them.  patch returns once its bytes are staged (asynchronous up to 1 MB per call: the
 * caller's buffers are free on return); append / truncate / replace synchronise their stream before they return. */
typedef struct ecgpu_resident_state ecgpu_resident_state_t;
int ecgpu_resident_state_create(int preset, const uint8_t* ssz, uint64_t n_bytes, ecgpu_resident_state_t** out); /* deneb */
int ecgpu_resident_state_create_fork(int fork, int preset, const uint8_t* ssz, uint64_t n_bytes, ecgpu_resident_state_t** out);
void ecgpu_resident_state_destroy(ecgpu_resident_state_t* st);
/* n patches: bytes data[data_off[i] .. data_off[i+1]) replace the encoding at offsets[i]; the patches of one call must
 * not overlap (they are applied concurrently) */
int ecgpu_resident_state_patch(ecgpu_resident_state_t* st, const uint64_t* offsets, const uint64_t* data_off,
                               const uint8_t* data, uint32_t n);
/* Lists that change length without re-creating the state: `append` adds whole elements at the end of a variable-length list
 * (add_validator_to_registry, phase0/block_processing.rs:317-349, is five appends: a 121-byte Validator, an 8-byte balance,
 * two participation bytes, an 8-byte inactivity score; historical_summaries grows once per period), `truncate` keeps the
 * first new_n_bytes of it (the eth1_data_votes reset).  The bytes behind the list move on the device, the SSZ offsets
 * of later fields are rewritten, patch offsets refer to the NEW encoding afterwards (ecgpu_resident_state_size). */
#define ECGPU_STATE_HISTORICAL_ROOTS 0
#define ECGPU_STATE_ETH1_DATA_VOTES 1
#define ECGPU_STATE_VALIDATORS 2
#define ECGPU_STATE_BALANCES 3
#define ECGPU_STATE_PREVIOUS_EPOCH_PARTICIPATION 4
#define ECGPU_STATE_CURRENT_EPOCH_PARTICIPATION 5
#define ECGPU_STATE_INACTIVITY_SCORES 6
#define ECGPU_STATE_HISTORICAL_SUMMARIES 8
#define ECGPU_STATE_PENDING_BALANCE_DEPOSITS 9      /* electra: 16-byte records */
#define ECGPU_STATE_PENDING_PARTIAL_WITHDRAWALS 10  /* electra: 24-byte records */
#define ECGPU_STATE_PENDING_CONSOLIDATIONS 11       /* electra: 16-byte records */
#define ECGPU_STATE_PREVIOUS_EPOCH_ATTESTATIONS 4   /* phase0 (in place of the participation lists): ecgpu_resident_state_replace only */
#define ECGPU_STATE_CURRENT_EPOCH_ATTESTATIONS 5
int ecgpu_resident_state_append(ecgpu_resident_state_t* st, int field, const uint8_t* data, uint64_t n_bytes);
int ecgpu_resident_state_truncate(ecgpu_resident_state_t* st, int field, uint64_t new_n_bytes);
/* `replace` exchanges a variable-length list for a new serialization of it, whatever the two lengths.  It is how the two
 * lists of variable-size elements of a phase0 state change (phase0/beacon_state.rs:80-81: process_attestation pushes a
 * PendingAttestation onto current_epoch_attestations, phase0/block_processing.rs:160-189; the epoch boundary moves current to
 * previous and empties it, phase0/epoch_processing.rs process_participation_record_updates): their offset tables are part of
 * the encoding, so there is no in-place append.  The list is rooted from `data` inside the call (a malformed serialization is
 * refused with ECGPU_ERR_BAD_ARG before anything moves); patches may not reach into these two lists.  Lists of fixed-size
 * elements can be replaced too (their cached tree is rebuilt at the next root). */
int ecgpu_resident_state_replace(ecgpu_resident_state_t* st, int field, const uint8_t* data, uint64_t n_bytes);
uint64_t ecgpu_resident_state_size(const ecgpu_resident_state_t* st);

/* Field-addressed changes (round 6; csrc/state_fields.h).  The reference's state transition names what it changes by field and
 * index -- `state.balances[index]` (increase_balance / decrease_balance, phase0/helpers.rs:979-1030), `state.validators.push`
 * (add_validator_to_registry, phase0/block_processing.rs:317-349, altair/block_processing.rs:192-213),
 * `state.current_epoch_participation[index]` (altair/block_processing.rs:98-170), `state.eth1_data_votes.push` / `.clear()`
 * (phase0/block_processing.rs:689-700), `state.state_roots[slot % N]`, `state.slot` (phase0/slot_processing.rs:58-86) -- never
 * by a byte offset.  These entries take the same coordinates: `field` = POSITION of the field in the fork's BeaconState
 * container (ECGPU_BS_*: phase0/beacon_state.rs:50-88 ... electra/beacon_state.rs:73-145; the same number is the field's chunk
 * in the container tree), plus a byte offset / element index INSIDE the field.  The library resolves them against the
 * encoding as it is when the bytes are applied, i.e. after every length change issued before -- a balance written after a
 * deposit of the same block lands behind the new validator record, without the caller recomputing anything.
 * Semantics: as if every call were applied at once, in program order (also relative to the byte-addressed entries above).
 * Implementation: writes and pushes are queued on the host and travel in ONE block at the next root / flush / byte-addressed
 * call (where two queued writes cover the same byte the later one wins; a write may target elements that are still queued);
 * truncate_field, set_field on a variable-size field and rotate_participation flush the queue and run at once.  An entry that
 * returns ECGPU_ERR_BAD_ARG (no such field in the fork, outside the field, past a list limit, not whole elements) has changed
 * nothing. */
#define ECGPU_BS_GENESIS_TIME 0
#define ECGPU_BS_GENESIS_VALIDATORS_ROOT 1
#define ECGPU_BS_SLOT 2
#define ECGPU_BS_FORK 3
#define ECGPU_BS_LATEST_BLOCK_HEADER 4
#define ECGPU_BS_BLOCK_ROOTS 5
#define ECGPU_BS_STATE_ROOTS 6
#define ECGPU_BS_HISTORICAL_ROOTS 7
#define ECGPU_BS_ETH1_DATA 8
#define ECGPU_BS_ETH1_DATA_VOTES 9
#define ECGPU_BS_ETH1_DEPOSIT_INDEX 10
#define ECGPU_BS_VALIDATORS 11
#define ECGPU_BS_BALANCES 12
#define ECGPU_BS_RANDAO_MIXES 13
#define ECGPU_BS_SLASHINGS 14
#define ECGPU_BS_PREVIOUS_EPOCH_PARTICIPATION 15 /* phase0: previous_epoch_attestations (set_field only) */
#define ECGPU_BS_CURRENT_EPOCH_PARTICIPATION 16  /* phase0: current_epoch_attestations (set_field only) */
#define ECGPU_BS_JUSTIFICATION_BITS 17
#define ECGPU_BS_PREVIOUS_JUSTIFIED_CHECKPOINT 18
#define ECGPU_BS_CURRENT_JUSTIFIED_CHECKPOINT 19
#define ECGPU_BS_FINALIZED_CHECKPOINT 20
#define ECGPU_BS_INACTIVITY_SCORES 21              /* altair+ */
#define ECGPU_BS_CURRENT_SYNC_COMMITTEE 22
#define ECGPU_BS_NEXT_SYNC_COMMITTEE 23
#define ECGPU_BS_LATEST_EXECUTION_PAYLOAD_HEADER 24 /* bellatrix+ */
#define ECGPU_BS_NEXT_WITHDRAWAL_INDEX 25           /* capella+ */
#define ECGPU_BS_NEXT_WITHDRAWAL_VALIDATOR_INDEX 26
#define ECGPU_BS_HISTORICAL_SUMMARIES 27
#define ECGPU_BS_DEPOSIT_RECEIPTS_START_INDEX 28    /* electra: 28 .. 33 are its six uint64 fields, in struct order */
#define ECGPU_BS_PENDING_BALANCE_DEPOSITS 34
#define ECGPU_BS_PENDING_PARTIAL_WITHDRAWALS 35
#define ECGPU_BS_PENDING_CONSOLIDATIONS 36
/* bytes [offset_in_field, offset_in_field + n_bytes) of the field's serialization are overwritten (any field except phase0's
 * two attestation lists; inside the payload header everything but its extra_data offset word) */
int ecgpu_resident_state_patch_field(ecgpu_resident_state_t* st, uint32_t field, uint64_t offset_in_field, const uint8_t* data,
                                     uint64_t n_bytes);
/* the same with the offset given as an element index: elements first_index .. of a list / vector (121-byte validators, 8-byte
 * balances / scores / slashings, 1-byte participation flags, 32-byte roots, 72-byte eth1 votes, 64-byte summaries ...) */
int ecgpu_resident_state_patch_elements(ecgpu_resident_state_t* st, uint32_t field, uint64_t first_index, const uint8_t* data,
                                        uint64_t n_bytes);
/* whole elements appended to a list of fixed-size elements (`list.push`) */
int ecgpu_resident_state_push(ecgpu_resident_state_t* st, uint32_t field, const uint8_t* data, uint64_t n_bytes);
/* the list keeps its first new_n_bytes (`list.clear()` = 0: process_eth1_data_reset) */
int ecgpu_resident_state_truncate_field(ecgpu_resident_state_t* st, uint32_t field, uint64_t new_n_bytes);
/* the whole field exchanged for `data`: a fixed-size field keeps its size; a list may change length; the payload header may
 * arrive with another extra_data; phase0's attestation lists change this way only */
int ecgpu_resident_state_set_field(ecgpu_resident_state_t* st, uint32_t field, const uint8_t* data, uint64_t n_bytes);
/* add_validator_to_registry: the 121-byte record and the balance are pushed; from altair on also a zero flag on both
 * participation lists and a zero inactivity score */
int ecgpu_resident_state_add_validator(ecgpu_resident_state_t* st, const uint8_t validator121[121], uint64_t balance);
/* process_participation_flag_updates (altair/epoch_processing.rs): previous_epoch_participation = current, current = zeros --
 * on the device, nothing travels */
int ecgpu_resident_state_rotate_participation(ecgpu_resident_state_t* st);
/* apply what is queued now (root does it anyway) */
int ecgpu_resident_state_flush(ecgpu_resident_state_t* st);
/* byte length of the field as program order has left it (queued pushes count), or a negative error */
int64_t ecgpu_resident_state_field_size(ecgpu_resident_state_t* st, uint32_t field);

int ecgpu_resident_state_root(ecgpu_resident_state_t* st, uint8_t root[32]);
/* asynchronous form: root written to device memory on `stream` */
int ecgpu_resident_state_root_dev(ecgpu_resident_state_t* st, uint8_t* d_root, ecgpu_stream_t stream);

/* Generic SSZ hash_tree_root driven by a type description: what `#[derive(SimpleSerialize)]` generates for every
 * container of the reference (`ssz_rs::HashTreeRoot`, ssz/mod.rs:4-7), e.g. deneb `BeaconBlock` /
 * `BeaconBlockBody` (deneb/beacon_block.rs:12-91; called via compute_signing_root, signing.rs:14-22, and at
 * phase0/block_processing.rs:591).  `types[i]` describes one SSZ type; composite types name their element /
 * field types by index.  The encoding is walked on the host (offsets only), every hash64 runs on the GPU. */
enum {
    ECGPU_SSZ_UINT = 0,       /* param = size in bytes (1, 2, 4, 8, 16, 32); `bool` is UINT 1 */
    ECGPU_SSZ_BYTEVECTOR = 1, /* param = length in bytes */
    ECGPU_SSZ_BYTELIST = 2,   /* param = limit in bytes */
    ECGPU_SSZ_VECTOR = 3,     /* elem = element type, param = length */
    ECGPU_SSZ_LIST = 4,       /* elem = element type, param = limit */
    ECGPU_SSZ_BITVECTOR = 5,  /* param = length in bits */
    ECGPU_SSZ_BITLIST = 6,    /* param = limit in bits */
    ECGPU_SSZ_CONTAINER = 7   /* fields[first_field .. first_field + n_fields) = field type indices */
};
typedef struct {
    uint32_t kind;
    uint32_t elem;
    uint64_t param;
    uint32_t n_fields;
    uint32_t first_field;
} ecgpu_ssz_type;
int ecgpu_htr_ssz(const ecgpu_ssz_type* types, uint32_t n_types, const uint32_t* fields, uint32_t n_field_refs,
                  uint32_t root_type, const uint8_t* ssz, uint64_t n_bytes, uint8_t root[32]);

/* Merkle proofs and generalized indices (second half): ssz_rs `GeneralizedIndexable::generalized_index`
 * and `Prove::prove` for any described type, as exercised at spec-tests/runners/light_client.rs:42-69,
 * deneb/blob_sidecar.rs:47-64 and deneb/beacon_block.rs:139-154.  A path element is a field POSITION (containers) or an
 * element index (vectors / lists; for packed basic sequences the proof ends at the chunk holding the element);
 * ECGPU_SSZ_PATH_LENGTH selects a list's length node.  prove: leaf, branch (bottom-up, 32 bytes per node, at most
 * max_depth nodes; *depth = nodes written), the generalized index and the witness root; every hash64 runs on the GPU.
 * Verify with ecgpu_is_valid_merkle_branch(leaf, branch, depth, gindex - (1 << depth), root). */
#define ECGPU_SSZ_PATH_LENGTH 0xffffffffffffffffull
int ecgpu_ssz_generalized_index(const ecgpu_ssz_type* types, uint32_t n_types, const uint32_t* fields, uint32_t n_field_refs,
                                uint32_t root_type, const uint64_t* path, uint32_t path_len, uint64_t* gindex);
int ecgpu_ssz_prove(const ecgpu_ssz_type* types, uint32_t n_types, const uint32_t* fields, uint32_t n_field_refs, uint32_t root_type,
                    const uint8_t* ssz, uint64_t n_bytes, const uint64_t* path, uint32_t path_len, uint8_t leaf[32], uint8_t* branch,
                    uint32_t max_depth, uint32_t* depth, uint64_t* gindex, uint8_t root[32]);
/* the branch of chunk `index` in merkleize(chunks, limit_chunks): ceil_log2(limit_chunks) sibling nodes, bottom-up */
int ecgpu_merkle_proof(const uint8_t* chunks, uint64_t n_chunks, uint64_t limit_chunks, uint64_t index, uint8_t* branch);
/* the roots of the fields of a BeaconState (the chunks of its container tree) next to its root: with ecgpu_merkle_proof
 * they give the light-client branches (current / next sync committee, finalized_checkpoint -> root) of a 2^20-validator
 * state at the cost of one state root.  roots: 32 * capacity bytes, *n_fields = 21 / 24 / 25 / 28 / 28 / 37 by fork. */
int ecgpu_beacon_state_field_roots(int fork, const uint8_t* ssz, uint64_t n_bytes, int preset, uint8_t* roots, uint32_t capacity,
                                   uint32_t* n_fields, uint8_t root[32]);

/* Swap-or-not shuffling: `compute_shuffled_indices(indices, seed, context)`
 * (phase0/helpers.rs:287-360; out[i] = indices[compute_shuffled_index(i, n, seed)], :249-282), the SHA-256 consumer
 * behind every committee computation.  ValidatorIndex = usize -> uint64_t.  indices == NULL: the permutation itself.
 * rounds = context.shuffle_round_count (90 on mainnet, 10 on minimal). */
int ecgpu_compute_shuffled_indices(const uint64_t* indices, uint64_t n, const uint8_t seed[32], uint32_t rounds,
                                   uint64_t* out);
int ecgpu_compute_shuffled_indices_dev(const uint64_t* d_indices, uint64_t n, const uint8_t seed[32], uint32_t rounds,
                                       uint64_t* d_out, ecgpu_stream_t stream);

/* ---- BLS12-381 (min_pk: 48-byte G1 public keys, 96-byte G2 signatures) ---------------------- */
/* crypto::verify_signature (crypto/bls.rs:64-77) */
int ecgpu_verify(const uint8_t pk[48], const uint8_t* msg, size_t msg_len, const uint8_t sig[96]);
/* crypto::fast_aggregate_verify (:114-132); eth_variant != 0 = eth_fast_aggregate_verify (:150-160) */
int ecgpu_fast_aggregate_verify(const uint8_t* pks48, uint32_t k, const uint8_t* msg, size_t msg_len,
                                const uint8_t sig[96], int eth_variant);
/* crypto::aggregate_verify (:95-112); messages concatenated, msg i = msgs[msg_off[i] .. msg_off[i+1]) */
int ecgpu_aggregate_verify(const uint8_t* pks48, uint32_t n_pks, const uint8_t* msgs,
                           const uint64_t* msg_off, uint32_t n_msgs, const uint8_t sig[96]);
/* crypto::aggregate (:79-93) and crypto::eth_aggregate_public_keys (:135-148) */
int ecgpu_aggregate_sigs(const uint8_t* sigs96, uint32_t n, uint8_t out[96]);
int ecgpu_aggregate_pks(const uint8_t* pks48, uint32_t n, uint8_t out[48]);

/* Multi-scalar multiplication sum_i [k_i] P_i over G1 / G2 (the north_star's "G1/G2 addition and multi-scalar-mult"; what a
 * random-coefficient batch check -- blst's verify_multiple_aggregate_signatures -- is made of).  Points arrive compressed and
 * are validated like the reference validates them (G1: key_validate, crypto/bls.rs:279-285; G2: from_bytes + the group
 * check of aggregate, :86-90): a bad point returns its BLST_ERROR.  Scalars: 32 big-endian bytes each, of which the low
 * `scalar_bits` (1..256) are used -- 64 for batch-check coefficients.  n == 0 -> ECGPU_EMPTY_AGGREGATE. */
int ecgpu_g1_msm(const uint8_t* pks48, const uint8_t* scalars32, uint32_t n, uint32_t scalar_bits, uint8_t out48[48]);
int ecgpu_g2_msm(const uint8_t* sigs96, const uint8_t* scalars32, uint32_t n, uint32_t scalar_bits, uint8_t out96[96]);

/* Batch entry (one call per block / per epoch instead of one call per signature):
 * n independent fast_aggregate_verify over 32-byte messages (every in-crate caller signs a
 * 32-byte signing root, signing.rs:14-22).  Tuple i uses public keys pk_off[i]..pk_off[i+1] of
 * pks48 (pk_off == NULL: exactly one key per tuple), message msgs32 + 32 i, signature
 * sigs96 + 96 i; status_out[i] receives the BLST_ERROR the scalar call would have returned. */
int ecgpu_fast_aggregate_verify_batch(const uint8_t* pks48, const uint32_t* pk_off,
                                      const uint8_t* msgs32, const uint8_t* sigs96, uint32_t n,
                                      int eth_variant, uint8_t* status_out);
int ecgpu_fast_aggregate_verify_batch_dev(const uint8_t* d_pks48, const uint32_t* d_pk_off,
                                          uint32_t n_pks_total, const uint8_t* d_msgs32,
                                          const uint8_t* d_sigs96, uint32_t n, int eth_variant,
                                          uint8_t* d_status_out, ecgpu_stream_t stream);

/* The same batch sharded over `n_devices` GPUs of this process (contiguous tuple ranges): one host thread per
 * device, every shard writes its statuses into status_out -- the all-gather of the verify booleans is the shared buffer.
 * devices may repeat (two shards on one GPU). */
int ecgpu_fast_aggregate_verify_batch_multi(const int* devices, uint32_t n_devices, const uint8_t* pks48, const uint32_t* pk_off,
                                            const uint8_t* msgs32, const uint8_t* sigs96, uint32_t n, int eth_variant,
                                            uint8_t* status_out);

/* Validated-key registry.  The reference decompresses and subgroup-checks every public key
 * on every call (`TryFrom<&PublicKey>`, crypto/bls.rs:279-285, reached from :122 for each key gathered out of
 * `state.validators` at phase0/helpers.rs:123-131): > 95 % of the work of a 2 048-key committee.  A registry keeps
 * the RESULT of that conversion per validator index on the device -- the affine point, or the BLST_ERROR the
 * conversion would raise -- so that an indexed verify returns exactly what the reference call over the same keys
 * returns.  `set` is the invalidation hook (`add_validator_to_registry`, phase0/block_processing.rs:317-349);
 * it must not run concurrently with a verify on the same registry.  Handles are process-wide. */
typedef struct ecgpu_registry ecgpu_registry_t;
int ecgpu_registry_create(uint64_t capacity, ecgpu_registry_t** out);
void ecgpu_registry_destroy(ecgpu_registry_t* reg);
/* keys first_index .. first_index + n: decompress + validate on the GPU, store point or error */
int ecgpu_registry_set(ecgpu_registry_t* reg, uint64_t first_index, const uint8_t* pks48, uint64_t n);
int ecgpu_registry_set_dev(ecgpu_registry_t* reg, uint64_t first_index, const uint8_t* d_pks48, uint64_t n,
                           ecgpu_stream_t stream);
/* ecgpu_fast_aggregate_verify_batch with tuple i's keys = registry[indices[idx_off[i] .. idx_off[i+1])] */
int ecgpu_fast_aggregate_verify_indexed_batch(const ecgpu_registry_t* reg, const uint32_t* indices,
                                              const uint32_t* idx_off, const uint8_t* msgs32,
                                              const uint8_t* sigs96, uint32_t n, int eth_variant,
                                              uint8_t* status_out);
int ecgpu_fast_aggregate_verify_indexed_batch_dev(const ecgpu_registry_t* reg, const uint32_t* d_indices,
                                                  const uint32_t* d_idx_off, uint32_t n_indices_total,
                                                  const uint8_t* d_msgs32, const uint8_t* d_sigs96, uint32_t n,
                                                  int eth_variant, uint8_t* d_status_out, ecgpu_stream_t stream);

/* Whole-block batching.  A block carries ~100-200 independent verifications that the reference
 * makes one after the other: proposer signature (phase0/state_transition.rs:56), randao reveal
 * (phase0/block_processing.rs:649), slashings / exits / deposits / BLS changes (signing.rs:40 callers), one
 * fast_aggregate_verify per attestation (phase0/block_processing.rs:752-761 -> phase0/helpers.rs:140) and the sync
 * aggregate (altair/block_processing.rs:226-234).  On a GPU a scalar call is ~6 ms of dependent latency; a collector
 * queues them (host memory only, any thread) and `flush` verifies everything queued in ONE pass of the batch pipeline.
 * status_out[p] = exactly what the scalar call pushed at position p would have returned (verify_signature = one key;
 * eth_variant != 0 = eth_fast_aggregate_verify).  `reg` (may be NULL) enables push_indexed: keys named by validator
 * index in a validated-key registry.  push returns the position (>= 0) or a negative error. */
typedef struct ecgpu_batch ecgpu_batch_t;
int ecgpu_batch_create(const ecgpu_registry_t* reg, ecgpu_batch_t** out);
void ecgpu_batch_destroy(ecgpu_batch_t* b);
int64_t ecgpu_batch_push(ecgpu_batch_t* b, const uint8_t* pks48, uint32_t k, const uint8_t* msg, size_t msg_len,
                         const uint8_t sig[96], int eth_variant);
int64_t ecgpu_batch_push_indexed(ecgpu_batch_t* b, const uint32_t* indices, uint32_t k, const uint8_t* msg, size_t msg_len,
                                 const uint8_t sig[96], int eth_variant);
uint32_t ecgpu_batch_len(const ecgpu_batch_t* b);
/* verifies and empties the batch; capacity = entries available in status_out (>= ecgpu_batch_len) */
int ecgpu_batch_flush(ecgpu_batch_t* b, uint8_t* status_out, uint32_t capacity);

/* SecretKey side, used to generate workloads and test vectors on the device:
 * SecretKey::public_key (crypto/bls.rs:193-197) and SecretKey::sign (:213-219).  sk = 32 big-endian
 * bytes, taken as given (pass sk < r).  msg_off == NULL: 32-byte messages at msgs + 32 i.
 * sk_stride (dev variant) = byte distance between consecutive secret keys (0: one key signs all). */
int ecgpu_sk_to_pk_batch(const uint8_t* sks32, uint32_t n, uint8_t* pks48);
int ecgpu_sign_batch(const uint8_t* sks32, const uint8_t* msgs, const uint64_t* msg_off, uint32_t n,
                     uint8_t* sigs96);
int ecgpu_sk_to_pk_batch_dev(const uint8_t* d_sks32, uint32_t n, uint8_t* d_pks48, ecgpu_stream_t stream);
int ecgpu_sign_batch_dev(const uint8_t* d_sks32, uint32_t sk_stride, const uint8_t* d_msgs32, uint32_t n,
                         uint8_t* d_sigs96, ecgpu_stream_t stream);

/* ---- measurement helpers (bench.py) --------------------------------------------------------- */
/* HIP-event timing of the dominant kernel of the last *_dev call on this thread:
 * returns the number of launches recorded and writes their total duration. */
int ecgpu_prof_enable(int on);
int ecgpu_prof_filter(const char* kernel_tag); /* NULL/"" = time every tagged kernel */
int ecgpu_prof_read(const char* kernel_tag, double* total_ms, uint64_t* launches);
/* Box self-check: the same 2^21 multiply-adds per lane as a loop over 8 KB of code and as a loop over 1 MB of code
 * (one wave per SIMD, like the BLS lane kernels).  The two take the same time on a healthy box. */
int ecgpu_selfcheck_ifetch(double* ms_small_loop, double* ms_large_loop);
/* the same work over loops of 8 KB, 64 KB, 256 KB and 1 MB of code */
int ecgpu_selfcheck_ifetch_sweep(double ms[4]);
/* Which build of the G2 stage kernels (signature decoding, hash-to-curve) this process uses: 1 = sums of products (fastest
 * on a healthy box), 2 = the compact-code tower (faster where the self-check above reports a slowdown beyond 1.5; the pairing
 * check then runs on the lane groups at every batch size).  Decided once per process, at the first BLS call or here; the
 * environment variable ECGPU_TOWER=sums|calls overrides the self-check. */
int ecgpu_bls_tower(void);
/* Which kernels ran the pairing check of the calling thread's last verification (for a ragged batch: of its full rounds):
 *   0 = none yet
 *   1 = one lane per tuple (k_pairing: Miller loop + final exponentiation in one kernel)
 *   3 = 16 / 12-lane groups over Fp registers in LDS with sums of products (bls_vm3.hip)
 *   5 = two lanes per tuple (k_miller2 + k_finalexp2 / k_finalexp: bls_pair2.h, bls_finalexp2.h)
 *   7 = one Fp operation across a 16-lane row, limb per lane (bls_row.hip: the latency path of small batches)
 * (2 was round 2's Fp2 lane groups, 4 and 6 never existed.)  ECGPU_PAIRING=auto|lane|vm3|split|row forces one path at every
 * size; auto (default) chooses by batch size: rows up to ECGPU_ROW_MAX tuples, lane groups up to ECGPU_VM_MAX, two lanes per
 * tuple up to ECGPU_SPLIT_MAX, the lane kernel above. */
int ecgpu_bls_last_pairing_path(void);
/* The batch sizes at which auto mode changes kernels on the calling thread's device: out[0] = rows up to, out[1] = lane groups up
 * to, out[2] = two lanes per tuple up to (then one lane per tuple); out[3] = 1 when out[0] / out[1] were MEASURED on this device
 * (ecgpu_warmup with ECGPU_WARM_BLS_BATCHES times the kernel sets against each other on the reference's fixed vector and places
 * the two crossovers), 0 for the built-in defaults.  ECGPU_ROW_MAX / ECGPU_VM_MAX / ECGPU_SPLIT_MAX override. */
int ecgpu_bls_dispatch_thresholds(uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* ECGPU_H */
