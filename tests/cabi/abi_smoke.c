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
        printf("no device: every entry point refused, as designed (%s)\n", ecgpu_last_error());
        return 0;
    }
    if (argc < 6) return 64;
    unsigned char want_root[32], pk[48], sig[96];
    if (hex2bin(argv[2], want_root, 32) || hex2bin(argv[3], pk, 48) || hex2bin(argv[4], sig, 96)) return 65;
    const char* msg = argv[5];
    if (ecgpu_init(-1) != ECGPU_SUCCESS) return 10;
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
    printf("C caller ok: %s\n", ecgpu_version());
    return 0;
}
