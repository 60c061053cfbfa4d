/* A C translation unit that uses libecgpu.so exactly as a foreign caller would: only include/ecgpu.h, plain C.
 * Exit code 0 = behaved as expected.  argv[1] = "nogpu": expect ECGPU_ERR_NO_DEVICE from every entry point;
 * argv[1] = "gpu": the reference's fixed signature (crypto/bls.rs:530-544) verifies, a forged message does not,
 * and hash_tree_root(BeaconBlockHeader::default()) has the expected root (hex in argv[2]). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ecgpu.h"

static int hex2bin(const char* h, unsigned char* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        unsigned v;
        if (sscanf(h + 2 * i, "%2x", &v) != 1) return -1;
        out[i] = (unsigned char)v;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    unsigned char root[32], hdr[112];
    memset(hdr, 0, sizeof hdr);
    if (!strcmp(argv[1], "nogpu")) {
        if (ecgpu_init(-1) != ECGPU_ERR_NO_DEVICE) return 1;
        if (ecgpu_htr_beacon_block_header(hdr, root) != ECGPU_ERR_NO_DEVICE) return 2;
        unsigned char pk[48] = {0}, sig[96] = {0};
        if (ecgpu_verify(pk, (const unsigned char*)"x", 1, sig) != ECGPU_ERR_NO_DEVICE) return 3;
        if (ecgpu_last_error() == NULL) return 4;
        if (ecgpu_warmup(0) != ECGPU_ERR_NO_DEVICE) return 5;
        uint32_t thr0[4];
        if (ecgpu_bls_dispatch_thresholds(thr0) != ECGPU_ERR_NO_DEVICE) return 6;
        printf("no device: every entry point refused, as designed (%s)\n", ecgpu_last_error());
        return 0;
    }
    if (argc < 6) return 64;
    unsigned char want_root[32], pk[48], sig[96];
    if (hex2bin(argv[2], want_root, 32) || hex2bin(argv[3], pk, 48) || hex2bin(argv[4], sig, 96)) return 65;
    const char* msg = argv[5];
    if (ecgpu_init(-1) != ECGPU_SUCCESS) return 10;
    /* the warm-up a host makes once per process (real calls with the reference's fixed vector) and the dispatch thresholds */
    if (ecgpu_warmup(ECGPU_WARM_BLS | ECGPU_WARM_MERKLE) != ECGPU_SUCCESS) return 30;
    uint32_t thr[4];
    if (ecgpu_bls_dispatch_thresholds(thr) != ECGPU_SUCCESS || thr[0] == 0 || thr[1] < thr[0] || thr[2] < thr[1]) return 31;
    if (ecgpu_htr_beacon_block_header(hdr, root) != ECGPU_SUCCESS || memcmp(root, want_root, 32)) return 11;
    if (ecgpu_verify(pk, (const unsigned char*)msg, strlen(msg), sig) != ECGPU_SUCCESS) return 12;
    if (ecgpu_verify(pk, (const unsigned char*)"forged", 6, sig) != ECGPU_VERIFY_FAIL) return 13;
    unsigned char st[2];
    unsigned char pks[96], msgs[64], sigs[192];
    memcpy(pks, pk, 48), memcpy(pks + 48, pk, 48);
    memset(msgs, 7, 64);
    memcpy(sigs, sig, 96), memcpy(sigs + 96, sig, 96);
    if (ecgpu_fast_aggregate_verify_batch(pks, NULL, msgs, sigs, 2, 0, st) != ECGPU_SUCCESS) return 14;
    if (st[0] != ECGPU_VERIFY_FAIL || st[1] != ECGPU_VERIFY_FAIL) return 15; /* wrong 32-byte messages */
    /* whole-block collector: two deferred verifications, one pass, the scalar entries' statuses in push order */
    ecgpu_batch_t* b = NULL;
    if (ecgpu_batch_create(NULL, &b) != ECGPU_SUCCESS || !b) return 16;
    if (ecgpu_batch_push(b, pk, 1, (const unsigned char*)msg, strlen(msg), sig, 0) != 0) return 17;
    if (ecgpu_batch_push(b, pk, 1, (const unsigned char*)"forged", 6, sig, 0) != 1) return 18;
    if (ecgpu_batch_len(b) != 2 || ecgpu_batch_flush(b, st, 2) != ECGPU_SUCCESS) return 19;
    if (st[0] != ECGPU_SUCCESS || st[1] != ECGPU_VERIFY_FAIL || ecgpu_batch_len(b) != 0) return 20;
    ecgpu_batch_destroy(b);
    /* the same batch split over a device list from one process (both entries name device 0 here) */
    int devs[2] = {0, 0};
    if (ecgpu_fast_aggregate_verify_batch_multi(devs, 2, pks, NULL, msgs, sigs, 2, 0, st) != ECGPU_SUCCESS) return 21;
    if (st[0] != ECGPU_VERIFY_FAIL || st[1] != ECGPU_VERIFY_FAIL) return 22;
    /* a Merkle branch of a 4-chunk tree, checked by the library's own is_valid_merkle_branch */
    unsigned char chunks[128], branch[64], r4[32];
    for (int i = 0; i < 128; i++) chunks[i] = (unsigned char)(i * 7 + 1);
    if (ecgpu_merkleize(chunks, 128, 4, 0, 0, r4) != ECGPU_SUCCESS) return 23;
    if (ecgpu_merkle_proof(chunks, 4, 4, 2, branch) != ECGPU_SUCCESS) return 24;
    if (ecgpu_is_valid_merkle_branch(chunks + 64, branch, 2, 2, r4) != ECGPU_SUCCESS) return 25;
    printf("C caller ok: %s\n", ecgpu_version());
    return 0;
}
