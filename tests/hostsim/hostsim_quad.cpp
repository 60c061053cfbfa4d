// Host simulator, 64 lanes per lane vector: the two row PAIRS of a wave in lock step (csrc/bls_rowcurve.h jac_dbl_quad: a G2 doubling
// scheduled over both pairs).  A translation unit of its own because the lane-vector width is a compile-time constant of
// bls_row.h; everything else in the simulator runs 32 lanes (one row pair).  Test infrastructure.
#define ECG_ROW_SIM 64
// (the inline functions of the headers have external linkage and their lane-vector types another size here: a namespace of this
// translation unit's own keeps the linker from merging them with the 32-lane ones of hostsim_bls.cpp)
#include <cstdint>
#include <cstring>
#include <vector>
#define ecg ecg_quad
#include "bls_rowcurve.h"

using namespace ecg;

namespace ecg {
unsigned long long g_ecg_column_overflows = 0;
}

static void q_out_fp(const Fp& a, u8* b) { raw_to_be48(fp_to_raw(a), b); }
static Fp q_in_fp(const u8* b) { return fp_from_raw(raw_from_be48(b, false)); }

extern "C" {

// the end of the message stage with one message per wave (k_h2c_finish_quad): the two maps by the one-lane routines, then the
// addition, the cofactor clearing with its doublings over both row pairs, the affine conversion.  xy: x.c0 x.c1 y.c0 y.c1
// big-endian canonical; *inf = the point's flag, or -1 if an accumulator overflowed.
void hs_hash_to_g2_quad(const u8* msg, u64 len, u8* xy, int* inf) {
    g_ecg_column_overflows = 0;
    J2 q0, q1;
    hash_to_g2_map(q0, msg, (size_t)len, 0);
    hash_to_g2_map(q1, msg, (size_t)len, 1);
    std::vector<u32> tab(16 * ROW_REG_DW, 0);
    A2 h;
    std::memset(&h, 0, sizeof(h));
    r_hash_to_g2_finish_quad(&h, &q0, &q1, tab.data());
    q_out_fp(h.x.c0, xy), q_out_fp(h.x.c1, xy + 48), q_out_fp(h.y.c0, xy + 96), q_out_fp(h.y.c1, xy + 144);
    *inf = g_ecg_column_overflows ? -1 : (int)h.inf;
}

// one doubling over both pairs against the generic one: Jacobian point in (x.c0 x.c1 y.c0 y.c1 z.c0 z.c1, 288 B), out the same
// returns 0, or 1 if the two pairs do not hold the same result, or -1 on an accumulator overflow
int hs_g2_dbl_quad(const u8* in288, u8* out288, u8* ref288) {
    g_ecg_column_overflows = 0;
    J2 p;
    Fp* c[6] = {&p.x.c0, &p.x.c1, &p.y.c0, &p.y.c1, &p.z.c0, &p.z.c1};
    for (int k = 0; k < 6; k++) *c[k] = q_in_fp(in288 + 48 * k);
    const RP2* tag = nullptr;
    const PJ2 a{f_load2(tag, &p.x), f_load2(tag, &p.y), f_load2(tag, &p.z)};
    PJ2 d, e;
    jac_dbl_quad(d, a);
    jac_dbl_inl(e, a);
    J2 o, w;
    f_store2(&o.x, d.x), f_store2(&o.y, d.y), f_store2(&o.z, d.z);
    f_store2(&w.x, e.x), f_store2(&w.y, e.y), f_store2(&w.z, e.z);
    Fp* oc[6] = {&o.x.c0, &o.x.c1, &o.y.c0, &o.y.c1, &o.z.c0, &o.z.c1};
    Fp* wc[6] = {&w.x.c0, &w.x.c1, &w.y.c0, &w.y.c1, &w.z.c0, &w.z.c1};
    for (int k = 0; k < 6; k++) q_out_fp(*oc[k], out288 + 48 * k), q_out_fp(*wc[k], ref288 + 48 * k);
    const RowK K = row_k();
    int differ = 0;
    const RP2* coords[3] = {&d.x, &d.y, &d.z};
    for (int k = 0; k < 3; k++) {
        const RFp cn = rfp_canon(RFp{coords[k]->v}, K);
        for (int l = 0; l < 32; l++) differ |= cn.v.v[l] != cn.v.v[32 + l];
    }
    return g_ecg_column_overflows ? -1 : differ;
}

// one addition over both pairs against the generic one: two Jacobian points in (288 B each), out / ref 288 B; returns 0, 1 if the
// pairs differ, -1 on an accumulator overflow
int hs_g2_add_quad(const u8* p288, const u8* q288, u8* out288, u8* ref288) {
    g_ecg_column_overflows = 0;
    J2 pj, qj;
    Fp* pc[6] = {&pj.x.c0, &pj.x.c1, &pj.y.c0, &pj.y.c1, &pj.z.c0, &pj.z.c1};
    Fp* qc[6] = {&qj.x.c0, &qj.x.c1, &qj.y.c0, &qj.y.c1, &qj.z.c0, &qj.z.c1};
    for (int k = 0; k < 6; k++) *pc[k] = q_in_fp(p288 + 48 * k), *qc[k] = q_in_fp(q288 + 48 * k);
    const RP2* tag = nullptr;
    const PJ2 a{f_load2(tag, &pj.x), f_load2(tag, &pj.y), f_load2(tag, &pj.z)}, b{f_load2(tag, &qj.x), f_load2(tag, &qj.y), f_load2(tag, &qj.z)};
    PJ2 d, e;
    jac_add_quad(d, a, b);
    jac_add_inl(e, a, b);
    J2 o, w;
    f_store2(&o.x, d.x), f_store2(&o.y, d.y), f_store2(&o.z, d.z);
    f_store2(&w.x, e.x), f_store2(&w.y, e.y), f_store2(&w.z, e.z);
    Fp* oc[6] = {&o.x.c0, &o.x.c1, &o.y.c0, &o.y.c1, &o.z.c0, &o.z.c1};
    Fp* wc[6] = {&w.x.c0, &w.x.c1, &w.y.c0, &w.y.c1, &w.z.c0, &w.z.c1};
    for (int k = 0; k < 6; k++) q_out_fp(*oc[k], out288 + 48 * k), q_out_fp(*wc[k], ref288 + 48 * k);
    const RowK K = row_k();
    int differ = 0;
    const RP2* coords[3] = {&d.x, &d.y, &d.z};
    for (int k = 0; k < 3; k++) {
        const RFp cn = rfp_canon(RFp{coords[k]->v}, K);
        for (int l = 0; l < 32; l++) differ |= cn.v.v[l] != cn.v.v[32 + l];
    }
    return g_ecg_column_overflows ? -1 : differ;
}

}  // extern "C"
