/* What a slot's 4 096 balance + 4 096 participation writes cost through the field-addressed entries (one C call per element, the
 * way the Rust hooks make them) against ONE byte-addressed ecgpu_resident_state_patch of the same bytes: host time to hand the
 * writes over, and wall time to the root.   usage: field_write_probe <state.ssz> <n_validators>   (tools/field_write_probe.sh) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ecgpu.h"

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

int main(int argc, char** argv) {
    if (argc < 3) return 64;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 65;
    fseek(f, 0, SEEK_END);
    long n_bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* enc = malloc(n_bytes);
    if (fread(enc, 1, n_bytes, f) != (size_t)n_bytes) return 66;
    const uint64_t n = strtoull(argv[2], NULL, 10);
    if (ecgpu_init(0) || ecgpu_warmup(ECGPU_WARM_MERKLE)) return 1;
    ecgpu_resident_state_t* st = NULL;
    if (ecgpu_resident_state_create(ECGPU_PRESET_MAINNET, enc, n_bytes, &st)) return 2;
    unsigned char root[32], root2[32];
    if (ecgpu_resident_state_root(st, root)) return 3;
    /* where the lists lie, for the byte-addressed form: fixed part, historical_roots, eth1 votes, validators, balances, prev, cur */
    const uint64_t fixed = ecgpu_beacon_state_deneb_fixed_size(ECGPU_PRESET_MAINNET);
    const uint32_t bal_field = ECGPU_BS_BALANCES, cur_field = ECGPU_BS_CURRENT_EPOCH_PARTICIPATION;
    enum { W = 4096 };
    static uint64_t idx_b[W], idx_p[W];
    uint64_t seed = 12345;
    double t_field_host = 0, t_field_root = 0, t_byte_host = 0, t_byte_root = 0;
    const int slots = 24;
    /* byte offsets of the two lists: field sizes before them */
    long long sz[40];
    for (unsigned k = 0; k < 28; k++) sz[k] = ecgpu_resident_state_field_size(st, k);
    /* variable parts in encoding order after the fixed part: 7, 9, 11, 12, 15, 16, ... */
    const uint64_t bal_off = fixed + sz[7] + sz[9] + sz[11], cur_off = bal_off + sz[12] + sz[15];
    for (int slot = 0; slot < 2 * slots; slot++) {
        for (int i = 0; i < W; i++) {
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            idx_b[i] = (seed >> 20) % n;
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            idx_p[i] = (seed >> 20) % n;
        }
        /* (duplicates among 4 096 random indices of 2^20 are rare but legal for the field-addressed form; the byte-addressed call
         *  must not overlap: make the indices distinct by construction) */
        for (int i = 0; i < W; i++) idx_b[i] = (idx_b[i] / W) * W + i < n ? (idx_b[i] / W) * W + i : i, idx_p[i] = (idx_p[i] / W) * W + i < n ? (idx_p[i] / W) * W + i : i;
        uint64_t v = 32000000000ull + slot;
        unsigned char flag = (unsigned char)(1 + slot % 7);
        const int field_form = slot & 1;
        double t0 = now_ms();
        if (field_form) {
            for (int i = 0; i < W; i++)
                if (ecgpu_resident_state_patch_elements(st, bal_field, idx_b[i], (const unsigned char*)&v, 8)) return 4;
            for (int i = 0; i < W; i++)
                if (ecgpu_resident_state_patch_elements(st, cur_field, idx_p[i], &flag, 1)) return 5;
            if (ecgpu_resident_state_flush(st)) return 6;
        } else {
            static uint64_t offs[2 * W], doff[2 * W + 1];
            static unsigned char data[9 * W];
            uint64_t at = 0;
            for (int i = 0; i < W; i++) {
                offs[i] = bal_off + 8 * idx_b[i], doff[i] = at;
                memcpy(data + at, &v, 8), at += 8;
            }
            for (int i = 0; i < W; i++) {
                offs[W + i] = cur_off + idx_p[i], doff[W + i] = at;
                data[at++] = flag;
            }
            doff[2 * W] = at;
            if (ecgpu_resident_state_patch(st, offs, doff, data, 2 * W)) return 7;
        }
        double t1 = now_ms();
        if (ecgpu_resident_state_root(st, field_form ? root : root2)) return 8;
        double t2 = now_ms();
        if (slot >= 4) {
            if (field_form) t_field_host += t1 - t0, t_field_root += t2 - t0;
            else t_byte_host += t1 - t0, t_byte_root += t2 - t0;
        }
    }
    const int cnt = slots - 2;
    printf("{\"writes_per_slot\": %d, \"field_addressed\": {\"hand_over_ms\": %.3f, \"to_root_ms\": %.3f}, \"byte_addressed\": {\"hand_over_ms\": %.3f, \"to_root_ms\": %.3f}}\n",
           2 * W, t_field_host / cnt, t_field_root / cnt, t_byte_host / cnt, t_byte_root / cnt);
    ecgpu_resident_state_destroy(st);
    return 0;
}
