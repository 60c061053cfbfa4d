// Lane programs of the BLS verification pipeline (one lane = one public key / one signature /
// one message / one pairing check) and the status algebra that reproduces the error ORDER of the
// reference wrappers (/root/reference/ethereum-consensus/src/crypto/bls.rs):
//   fast_aggregate_verify :114-132  keys left to right (first failing key wins) -> signature
//       decode -> empty key list (AGGR_TYPE_MISMATCH) -> signature group check -> aggregate key
//       at infinity -> pairing equation.  The first two are conversions (Error::BLST); what follows happens inside
//       blst's verify call (Error::InvalidSignature): its POINT_NOT_IN_GROUP / PK_IS_INFINITY carry ECGPU_IN_VERIFY.
//   eth_fast_aggregate_verify :150-160  (no keys AND sig == 0xc0 00..00) -> Ok, else as above
//   aggregate_verify :95-112, aggregate :79-93, eth_aggregate_public_keys :135-148
// The kernels in bls.hip run these stage by stage over a whole batch; tests/hostsim runs the same
// functions lane by lane on the CPU.
#pragma once
#include "bls_h2c.h"
#include "bls_pairing.h"

namespace ecg {

// stage 1: one lane per public key
ECG_HD u8 stage_pk_validate(A1& out, const u8* pk48) { return (u8)g1_key_validate(out, pk48); }

// stage 2 (serial form): sum of validated keys lo..hi; first failing status wins
ECG_HD u8 stage_pk_aggregate_serial(A1& agg, const A1* pts, const u8* st, u32 lo, u32 hi) {
    J1 acc;
    jac_set_inf(acc);
    for (u32 i = lo; i < hi; i++) {
        if (st[i]) {
            agg.inf = 1;
            agg.x = fp_zero();
            agg.y = fp_zero();
            return st[i];
        }
        jac_add_aff(acc, acc, pts[i].x, pts[i].y);
    }
    jac_to_aff(agg, acc);
    return 0;
}

// stage 3: one lane per signature: decode (on-curve) and group check, reported separately because
// other errors rank between them.
ECG_HD void stage_sig(A2& out, u8& st_decode, u8& st_group, const u8* sig96) {
    st_group = 0;
    st_decode = (u8)g2_decompress(out, sig96);
    if (st_decode) return;
    if (!g2_in_subgroup(out)) st_group = ECGPU_POINT_NOT_IN_GROUP;
}

ECG_HD bool sig_is_infinity_bytes(const u8* sig96) { return sig96[0] == 0xc0 && bytes_all_zero(sig96, 1, 96); }

// stage 5: e(agg_pk, H) * e(-g1, sig) == 1
ECG_HD u8 stage_pairing(const A1& agg, const A2& h, const A2& sig) {
    A1 ng;
    ng.x = blsc::G1_X;
    ng.y = blsc::G1_NEG_Y;
    ng.inf = 0;
    return pairing_product2_is_one(agg, h, ng, sig) ? ECGPU_SUCCESS : ECGPU_VERIFY_FAIL;
}

// status algebra for one fast_aggregate_verify tuple once every stage has reported
ECG_HD u8 combine_fav_status(u32 k, bool eth_variant, bool sig_inf_bytes, u8 st_pk, u8 st_sig_decode, u8 st_sig_group,
                             bool agg_inf, u8 st_pairing) {
    if (eth_variant && k == 0 && sig_inf_bytes) return ECGPU_SUCCESS;
    if (st_pk) return st_pk;
    if (st_sig_decode) return st_sig_decode;
    if (k == 0) return ECGPU_AGGR_TYPE_MISMATCH;
    if (st_sig_group) return ECGPU_IN_VERIFY | st_sig_group;  // found by verify's group check: Error::InvalidSignature
    if (agg_inf) return ECGPU_VERIFY_PK_IS_INFINITY;          // keys sum to infinity inside verify: Error::InvalidSignature
    return st_pairing;
}

// the whole tuple on one lane (hostsim, and the reference for the staged kernels)
ECG_HD u8 fav_tuple_serial(const u8* pks48, u32 k, const u8* msg, size_t msg_len, const u8* sig96, bool eth_variant) {
    if (eth_variant && k == 0 && sig_is_infinity_bytes(sig96)) return ECGPU_SUCCESS;
    J1 acc;
    jac_set_inf(acc);
    for (u32 i = 0; i < k; i++) {
        A1 p;
        u8 st = stage_pk_validate(p, pks48 + 48 * (size_t)i);
        if (st) return st;
        jac_add_aff(acc, acc, p.x, p.y);
    }
    A2 sig;
    u8 sd, sg;
    stage_sig(sig, sd, sg, sig96);
    if (sd) return sd;
    if (k == 0) return ECGPU_AGGR_TYPE_MISMATCH;
    if (sg) return ECGPU_IN_VERIFY | sg;
    A1 agg;
    jac_to_aff(agg, acc);
    if (agg.inf) return ECGPU_VERIFY_PK_IS_INFINITY;
    A2 h;
    hash_to_g2(h, msg, msg_len);
    return stage_pairing(agg, h, sig);
}

}  // namespace ecg
