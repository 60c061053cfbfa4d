// tests/hostsim (BLS half): CPU-side execution of the very same ECG_HD lane programs the gfx950
// kernels run -- TEST INFRASTRUCTURE ONLY (see hostsim.cpp).  Values cross this boundary as
// big-endian canonical integers so that the tests can compare with oracle/bls12_381.py directly.
#include <cstring>
#include <array>
#include <vector>

#define ECG_COUNT_OPS 1
#include "ecgpu.h"
#include "bls_verify.h"
#include <thread>

#include "bls_vm3.h"
#include "bls_row.h"
#include "bls_rowcurve.h"
#include "bls_vm3_prog.h"
#include "bls_pair2.h"
#include "bls_g2_pair2.h"
#include "bls_finalexp2.h"
#include <atomic>

using namespace ecg;

namespace ecg {
unsigned long long g_ecg_fp_mul_count = 0, g_ecg_fp_sqr_count = 0, g_ecg_fp_mad_count = 0, g_ecg_column_overflows = 0;
}

// ---- the two lanes of a pair (bls_pair2.h) as two host threads in lock step: an exchange is a rendezvous ---------------------
namespace ecg {
thread_local PairChannel* t_pair_channel = nullptr;
thread_local u32 t_pair_lane = 0;
static std::atomic<int> g_pair_arrivals[2];
Fp h_xch_host(const Fp& e) {
    PairChannel* ch = t_pair_channel;
    const u32 me = t_pair_lane;
    ch->slot[me] = e;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    // sense-reversing barrier over two counters: publish, wait for the partner, read, acknowledge, wait for the partner's read
    g_pair_arrivals[0].fetch_add(1, std::memory_order_acq_rel);
    while (g_pair_arrivals[0].load(std::memory_order_acquire) % 2 != 0) std::this_thread::yield();
    const Fp r = ch->slot[1 - me];
    g_pair_arrivals[1].fetch_add(1, std::memory_order_acq_rel);
    while (g_pair_arrivals[1].load(std::memory_order_acquire) % 2 != 0) std::this_thread::yield();
    return r;
}
}  // namespace ecg

static Fp in_fp(const u8* b) { return fp_from_raw(raw_from_be48(b, false)); }
static void out_fp(const Fp& a, u8* b) { raw_to_be48(fp_to_raw(a), b); }
static Fp2 in_fp2(const u8* b) { return Fp2{in_fp(b), in_fp(b + 48)}; }
static void out_fp2(const Fp2& a, u8* b) {
    out_fp(a.c0, b);
    out_fp(a.c1, b + 48);
}

// Sum of n (1..8) products over RAW limb vectors (13 u32 each, no range reduction: the test chooses lazy operands up to
// the documented bounds and worst-case limb patterns): out = 13 limbs of sum a_k b_k / R mod p (almost reduced).
// Returns the number of 64-bit column overflows the host arithmetic detected since the last call (must be 0).
template <int N>
static void sumprod_raw(const u32* a, const u32* b, u32* out) {
    Fp x[N], y[N];
    for (int k = 0; k < N; k++)
        for (int i = 0; i < 13; i++) {
            x[k].l[i] = a[13 * k + i];
            y[k].l[i] = b[13 * k + i];
        }
    const Fp r = fp_sumprod<N>(x, y);
    for (int i = 0; i < 13; i++) out[i] = r.l[i];
}
extern "C" {

// op: 0 mul 1 add 2 sub 3 neg 4 inv 5 sqrt(returns 1 if square) 6 sqr 7 lex_largest 8 dbl 9 is_zero 10 eq
int hs_fp_op(int op, const u8* a, const u8* b, u8* out) {
    Fp x = in_fp(a), y = b ? in_fp(b) : fp_zero(), r = fp_zero();
    int rc = 0;
    switch (op) {
        case 0: r = fp_mul(x, y); break;
        case 1: r = fp_add(x, y); break;
        case 2: r = fp_sub(x, y); break;
        case 3: r = fp_neg(x); break;
        case 4: r = fp_inv(x); break;
        case 5: rc = fp_sqrt(x, r) ? 1 : 0; break;
        case 6: r = fp_sqr(x); break;
        case 7: rc = fp_lex_largest(x) ? 1 : 0; break;
        case 8: r = fp_dbl(x); break;
        case 9: rc = fp_is_zero(x) ? 1 : 0; break;
        case 10: rc = fp_eq(x, y) ? 1 : 0; break;
        default: return -1;
    }
    out_fp(r, out);
    return rc;
}

// chained stress: r = a; repeat n times r = r*b + a - b (keeps values in the lazy [0,2p) range busy)
void hs_fp_chain(const u8* a, const u8* b, int n, u8* out) {
    Fp x = in_fp(a), y = in_fp(b), r = x;
    for (int i = 0; i < n; i++) r = fp_sub(fp_add(fp_mul(r, y), x), y);
    out_fp(r, out);
}

// op: 0 mul 1 sqr 2 inv 3 sqrt(rc=1 if square) 4 sgn0 5 lex_largest 6 mul_xi 7 add 8 sub 9 neg 10 conj
int hs_fp2_op(int op, const u8* a, const u8* b, u8* out) {
    Fp2 x = in_fp2(a), y = b ? in_fp2(b) : fp2_zero(), r = fp2_zero();
    int rc = 0;
    switch (op) {
        case 0: r = fp2_mul(x, y); break;
        case 1: r = fp2_sqr(x); break;
        case 2: r = fp2_inv(x); break;
        case 3: rc = fp2_sqrt(x, r) ? 1 : 0; break;
        case 4: rc = (int)fp2_sgn0(x); break;
        case 5: rc = fp2_lex_largest(x) ? 1 : 0; break;
        case 6: r = fp2_mul_xi(x); break;
        case 7: r = fp2_add(x, y); break;
        case 8: r = fp2_sub(x, y); break;
        case 9: r = fp2_neg(x); break;
        case 10: r = fp2_conj(x); break;
        default: return -1;
    }
    out_fp2(r, out);
    return rc;
}

// ---- points ----------------------------------------------------------------------------------
// affine points cross as x||y big-endian canonical (96 B for G1, 192 B for G2: x.c0 x.c1 y.c0 y.c1)
static void out_a1(const A1& p, u8* b) {
    out_fp(p.x, b);
    out_fp(p.y, b + 48);
}
static void out_a2(const A2& p, u8* b) {
    out_fp2(p.x, b);
    out_fp2(p.y, b + 96);
}
static A1 in_a1(const u8* b, int inf) {
    A1 p;
    p.x = in_fp(b);
    p.y = in_fp(b + 48);
    p.inf = (u32)inf;
    return p;
}
static A2 in_a2(const u8* b, int inf) {
    A2 p;
    p.x = in_fp2(b);
    p.y = in_fp2(b + 96);
    p.inf = (u32)inf;
    return p;
}

int hs_g1_decompress(const u8* b48, u8* xy, int* inf) {
    A1 p;
    int st = g1_decompress(p, b48);
    out_a1(p, xy);
    *inf = (int)p.inf;
    return st;
}
int hs_g1_key_validate(const u8* b48, u8* xy) {
    A1 p;
    int st = g1_key_validate(p, b48);
    out_a1(p, xy);
    return st;
}
int hs_g1_in_subgroup(const u8* xy) { return g1_in_subgroup(in_a1(xy, 0)) ? 1 : 0; }
void hs_g1_compress(const u8* xy, int inf, u8* out48) { g1_compress(out48, in_a1(xy, inf)); }
// sum of n affine points (inf flags in `infs`), result affine
void hs_g1_sum(const u8* xys, const int* infs, int n, u8* xy, int* inf) {
    J1 acc;
    jac_set_inf(acc);
    for (int i = 0; i < n; i++) {
        A1 p = in_a1(xys + 96 * i, infs[i]);
        J1 q;
        jac_from_aff(q, p);
        if (i & 1) jac_add(acc, acc, q);
        else if (!p.inf) jac_add_aff(acc, acc, p.x, p.y);
    }
    A1 r;
    jac_to_aff(r, acc);
    out_a1(r, xy);
    *inf = (int)r.inf;
}
void hs_g1_mul(const u8* xy, const u8* k32be, u8* out, int* inf) {
    u32 k[8];
    for (int i = 0; i < 8; i++) k[i] = ((u32)k32be[4 * (7 - i)] << 24) | ((u32)k32be[4 * (7 - i) + 1] << 16) | ((u32)k32be[4 * (7 - i) + 2] << 8) | k32be[4 * (7 - i) + 3];
    J1 P, R;
    jac_from_aff(P, in_a1(xy, 0));
    jac_mul_scalar(R, P, k, 8);
    A1 r;
    jac_to_aff(r, R);
    out_a1(r, out);
    *inf = (int)r.inf;
}

int hs_g2_decompress(const u8* b96, u8* xy, int* inf) {
    A2 p;
    int st = g2_decompress(p, b96);
    out_a2(p, xy);
    *inf = (int)p.inf;
    return st;
}
int hs_g2_in_subgroup(const u8* xy) { return g2_in_subgroup(in_a2(xy, 0)) ? 1 : 0; }
void hs_g2_compress(const u8* xy, int inf, u8* out96) { g2_compress(out96, in_a2(xy, inf)); }
void hs_g2_sum(const u8* xys, const int* infs, int n, u8* xy, int* inf) {
    J2 acc;
    jac_set_inf(acc);
    for (int i = 0; i < n; i++) {
        A2 p = in_a2(xys + 192 * i, infs[i]);
        J2 q;
        jac_from_aff(q, p);
        if (i & 1) jac_add(acc, acc, q);
        else if (!p.inf) jac_add_aff(acc, acc, p.x, p.y);
    }
    A2 r;
    jac_to_aff(r, acc);
    out_a2(r, xy);
    *inf = (int)r.inf;
}
void hs_g2_mul(const u8* xy, const u8* k32be, u8* out, int* inf) {
    u32 k[8];
    for (int i = 0; i < 8; i++) k[i] = ((u32)k32be[4 * (7 - i)] << 24) | ((u32)k32be[4 * (7 - i) + 1] << 16) | ((u32)k32be[4 * (7 - i) + 2] << 8) | k32be[4 * (7 - i) + 3];
    J2 P, R;
    jac_from_aff(P, in_a2(xy, 0));
    jac_mul_scalar(R, P, k, 8);
    A2 r;
    jac_to_aff(r, R);
    out_a2(r, out);
    *inf = (int)r.inf;
}
void hs_xmd(const u8* msg, u64 len, u8* out256) { xmd_expand_256(out256, msg, (size_t)len); }
// split != 0: the two-lanes-per-message form of the small-batch message stage (hash_to_g2_map x 2, hash_to_g2_finish)
void hs_hash_to_g2_split(const u8* msg, u64 len, u8* xy, int* inf) {
    J2 q0, q1;
    hash_to_g2_map(q0, msg, (size_t)len, 0);
    hash_to_g2_map(q1, msg, (size_t)len, 1);
    A2 h;
    hash_to_g2_finish(h, q0, q1);
    out_a2(h, xy);
    *inf = (int)h.inf;
}
// the two-lanes-per-message end of the message stage (bls_g2_pair2.h, k_h2c_finish2): the two maps on one lane each, then the
// addition, the cofactor clearing and the affine conversion on a lane PAIR -- two host threads in lock step
void hs_hash_to_g2_pair2(const u8* msg, u64 len, u8* xy, int* inf) {
    J2 q0, q1;
    hash_to_g2_map(q0, msg, (size_t)len, 0);
    hash_to_g2_map(q1, msg, (size_t)len, 1);
    PairChannel ch;
    std::memset(&ch, 0, sizeof(ch));
    g_pair_arrivals[0] = 0;
    g_pair_arrivals[1] = 0;
    A2 h;
    std::memset(&h, 0, sizeof(h));
    auto lane = [&](u32 s) {
        t_pair_channel = &ch;
        t_pair_lane = s;
        h_hash_to_g2_finish(&h, q0, q1);
    };
    std::thread t1(lane, 1u);
    lane(0u);
    t1.join();
    out_a2(h, xy);
    *inf = (int)h.inf;
}
// the end of the message stage on a ROW (bls_rowcurve.h, k_h2c_finish_row): the two maps on one lane each, then the addition,
// the cofactor clearing and the affine conversion with one point per 16-lane row -- the same routines on the host's lane vectors
// which: 0 = only the end on a row (the two maps by the one-lane routines), 1 = the SSWU maps on rows as well (k_h2c_map_row)
void hs_hash_to_g2_row(const u8* msg, u64 len, u8* xy, int* inf, int which) {
    g_ecg_column_overflows = 0;
    J2 q0, q1;
    std::vector<u32> tab(16 * ROW_REG_DW, 0);
    if (which & 1) {
        r_hash_to_g2_map(&q0, msg, (size_t)len, 0, tab.data());
        r_hash_to_g2_map(&q1, msg, (size_t)len, 1, tab.data());
    } else {
        hash_to_g2_map(q0, msg, (size_t)len, 0);
        hash_to_g2_map(q1, msg, (size_t)len, 1);
    }
    A2 h;
    std::memset(&h, 0, sizeof(h));
    if (which & 2) r_hash_to_g2_finish<RP2>(&h, &q0, &q1, tab.data());  // a row PAIR per message (k_h2c_finish_row)
    else r_hash_to_g2_finish<RFp2>(&h, &q0, &q1, tab.data());
    out_a2(h, xy);
    *inf = g_ecg_column_overflows ? -1 : (int)h.inf;
}
// the psi subgroup check of a decoded signature on a row (k_sig_group_row): 1 in G2, 0 not; affine point canonical big-endian
int hs_g2_in_subgroup_row(const u8* xy, int pair) {
    g_ecg_column_overflows = 0;
    const A2 q = in_a2(xy, 0);
    const int r = (pair ? r_g2_in_subgroup<RP2>(&q) : r_g2_in_subgroup<RFp2>(&q)) ? 1 : 0;
    return g_ecg_column_overflows ? -1 : r;
}
// Signature::try_from + the group check on a row pair (k_sig_row): returns st_dec | st_grp << 8, the point as the kernel stores it
int hs_sig_row(const u8* b96, u8* xy, int* inf) {
    g_ecg_column_overflows = 0;
    std::vector<u32> tab(16 * ROW_REG_DW, 0);
    A2 p;
    p.x = fp2_zero(), p.y = fp2_zero(), p.inf = 7;
    u8 sd = 0xee, sg = 0xee;
    r_sig_decode_and_group(&p, &sd, &sg, b96, tab.data());
    out_a2(p, xy);
    *inf = (int)p.inf;
    return g_ecg_column_overflows ? -1 : (int)sd | ((int)sg << 8);
}
// key_validate on a row (k_pk_row): returns the status, the point as the kernel stores it
int hs_pk_row(const u8* b48, u8* xy, int* inf) {
    g_ecg_column_overflows = 0;
    std::vector<u32> tab(16 * ROW_REG_DW, 0);
    A1 p;
    p.x = fp_zero(), p.y = fp_zero(), p.inf = 7;
    u8 st = 0xee;
    r_pk_validate(&p, &st, b48, tab.data());
    out_a1(p, xy);
    *inf = (int)p.inf;
    return g_ecg_column_overflows ? -1 : (int)st;
}
// the subgroup check of a decoded public key on a row (k_pk_group_row): 1 in G1, 0 not
int hs_g1_in_subgroup_row(const u8* xy) {
    g_ecg_column_overflows = 0;
    const A1 p = in_a1(xy, 0);
    const int r = r_g1_in_subgroup(&p) ? 1 : 0;
    return g_ecg_column_overflows ? -1 : r;
}
// Fp2 square root / signs on a row against the one-lane routines: out = root (canonical big-endian c0 | c1); returns
// is_square | sgn0(a) << 1 | lex_largest(a) << 2
int hs_rowfield_sqrt(const u8* a96, u8* out96) {
    g_ecg_column_overflows = 0;
    const Fp2 a = in_fp2(a96);
    const RowK K = row_k();
    std::vector<u32> tab(16 * ROW_REG_DW, 0);
    const RFp2 ra = rfp2_load(&a);
    RFp2 r;
    f_set_zero(r);
    const bool sq = rfp2_sqrt(ra, r, tab.data(), K);
    Fp2 o;
    rfp_store(&o.c0, r.c0);
    rfp_store(&o.c1, r.c1);
    out_fp2(o, out96);
    if (g_ecg_column_overflows) return -1;
    return (sq ? 1 : 0) | (int)(rfp2_sgn0(ra, K) << 1) | (rfp2_lex_largest(ra, K) ? 4 : 0);
}

// row field operations on raw limbs (13 x u32 in, 13 out): 0 add 1 sub 2 neg 3 mul 4 sqr 5 canon 6 sub_dbl 7 pow_pm3d4 8 is_zero 9 eq
int hs_rowfield_op(int op, const u32* a, const u32* b, u32* out) {
    g_ecg_column_overflows = 0;
    const RowK K = row_k();
    Fp fa, fb;
    for (int i = 0; i < 13; i++) fa.l[i] = a[i], fb.l[i] = b ? b[i] : 0;
    const RFp x = rfp_load(&fa), y = rfp_load(&fb);
    RFp r = rfp_zero();
    int rc = 0;
    std::vector<u32> tab(16 * ROW_REG_DW, 0);
    switch (op) {
        case 0: r = rfp_add(x, y, K); break;
        case 1: r = rfp_sub(x, y, K); break;
        case 2: r = rfp_neg(x, K); break;
        case 3: r = rfp_mul(x, y, K); break;
        case 4: r = rfp_sqr(x, K); break;
        case 5: r = rfp_canon(x, K); break;
        case 6: r = rfp_sub_dbl(x, y, K); break;
        case 7: r = rfp_pow_pm3d4(x, tab.data(), K); break;
        case 8: rc = rfp_is_zero(x, K) ? 1 : 0; break;
        case 9: rc = rfp_eq(x, y, K) ? 1 : 0; break;
        default: return -2;
    }
    for (int i = 0; i < 16; i++) {
        out[i] = r.v.v[i];
        if (r.v.v[16 + i] != r.v.v[i]) return -3;  // both simulated rows ran the same computation
    }
    return g_ecg_column_overflows ? -1 : rc;
}

void hs_hash_to_g2(const u8* msg, u64 len, u8* xy, int* inf) {
    A2 h;
    hash_to_g2(h, msg, (size_t)len);
    out_a2(h, xy);
    *inf = (int)h.inf;
}

// ---- Fp12: 12 Fp coefficients in the order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1 -------------
static void out_fp12(const Fp12& a, u8* b) {
    const Fp2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
    for (int i = 0; i < 6; i++) out_fp2(*c[i], b + 96 * i);
}
static Fp12 in_fp12(const u8* b) {
    Fp12 a;
    Fp2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
    for (int i = 0; i < 6; i++) *c[i] = in_fp2(b + 96 * i);
    return a;
}
// op: 0 mul 1 sqr 2 inv 3 frob 4 conj 5 cyclotomic_sqr 6 final_exp 7 cyc_pow_x
void hs_fp12_op(int op, const u8* a, const u8* b, u8* out) {
    Fp12 x = in_fp12(a), y, r;
    if (b) y = in_fp12(b);
    switch (op) {
        case 0: fp12_mul(r, x, y); break;
        case 1: fp12_sqr(r, x); break;
        case 2: fp12_inv(r, x); break;
        case 3: fp12_frob(r, x); break;
        case 4: fp12_conj(r, x); break;
        case 5: fp12_cyclotomic_sqr(r, x); break;
        case 6: final_exponentiation(r, x); break;
        default: fp12_cyc_pow_x(r, x); break;
    }
    out_fp12(r, out);
}
// final_exponentiation(miller(P0,Q0) * miller(P1,Q1)); n = 1 or 2 pairs
void hs_pairing(int n, const u8* p_xy, const int* p_inf, const u8* q_xy, const int* q_inf, u8* out) {
    MillerPair pr[2];
    for (int k = 0; k < n; k++) miller_pair_init(pr[k], in_a1(p_xy + 96 * k, p_inf[k]), in_a2(q_xy + 192 * k, q_inf[k]));
    Fp12 f, e;
    miller_loop(f, pr, n);
    final_exponentiation(e, f);
    out_fp12(e, out);
}

// The same two-pair Miller loop on TWO lanes (bls_pair2.h h_miller_loop: what k_miller2 runs), then the one-lane final
// exponentiation (k_finalexp): out = the 12 coefficients like hs_pairing.  miller_only != 0: the Miller value itself.
void hs_pairing_split(const u8* p_xy, const int* p_inf, const u8* q_xy, const int* q_inf, int miller_only, u8* out) {
    PairChannel ch;
    std::memset(&ch, 0, sizeof(ch));
    g_pair_arrivals[0] = 0;
    g_pair_arrivals[1] = 0;
    Fp12 f;
    auto lane = [&](u32 s) {
        t_pair_channel = &ch;
        t_pair_lane = s;
        MillerPairH pr[2];
        for (int k = 0; k < 2; k++) miller_pair_h_init(pr[k], in_a1(p_xy + 96 * k, p_inf[k]), in_a2(q_xy + 192 * k, q_inf[k]));
        H12 h;
        h_miller_loop(h, pr);
        h12_store(&f, h);
    };
    std::thread t1(lane, 1u);
    lane(0u);
    t1.join();
    if (miller_only == 1) {
        out_fp12(f, out);
        return;
    }
    Fp12 e;
    if (miller_only == 2) {
        // ... and the final exponentiation on the lane pair as well (bls_finalexp2.h: what k_finalexp2 runs); out[576] = the
        // pair's verdict "is one" (must be the same on both lanes)
        int verdict[2] = {-1, -1};
        auto lane2 = [&](u32 s) {
            t_pair_channel = &ch;
            t_pair_lane = s;
            H12 h, r;
            h12_load(h, &f);
            h_final_exponentiation(r, h);
            verdict[s] = h12_is_one(r) ? 1 : 0;
            h12_store(&e, r);
        };
        g_pair_arrivals[0] = 0;
        g_pair_arrivals[1] = 0;
        std::thread t2(lane2, 1u);
        lane2(0u);
        t2.join();
        out[576] = (u8)(verdict[0] == verdict[1] ? verdict[0] : 0xee);
    } else {
        final_exponentiation(e, f);
    }
    out_fp12(e, out);
}
// the one-lane Miller value, for comparison with the split one
void hs_miller(const u8* p_xy, const int* p_inf, const u8* q_xy, const int* q_inf, u8* out) {
    MillerPair pr[2];
    for (int k = 0; k < 2; k++) miller_pair_init(pr[k], in_a1(p_xy + 96 * k, p_inf[k]), in_a2(q_xy + 192 * k, q_inf[k]));
    Fp12 f;
    miller_loop(f, pr, 2);
    out_fp12(f, out);
}

// Multiplier census of the stages of one K = 1 verification (valid inputs): out[3*s] = fp_mul calls, out[3*s+1] = fp_sqr
// calls, out[3*s+2] = multiply instructions inside sums of products, for s = pk_validate, sig (decode + group check),
// hash_to_g2, pairing.
void hs_op_census(const u8* pk48, const u8* msg, u64 msg_len, const u8* sig96, u64* out) {
    auto snap = [&](int s) {
        out[3 * s] = g_ecg_fp_mul_count;
        out[3 * s + 1] = g_ecg_fp_sqr_count;
        out[3 * s + 2] = g_ecg_fp_mad_count;
        g_ecg_fp_mul_count = g_ecg_fp_sqr_count = g_ecg_fp_mad_count = 0;
    };
    g_ecg_fp_mul_count = g_ecg_fp_sqr_count = g_ecg_fp_mad_count = 0;
    A1 p;
    stage_pk_validate(p, pk48);
    snap(0);
    A2 sg;
    u8 sd, sgr;
    stage_sig(sg, sd, sgr, sig96);
    snap(1);
    A2 h;
    hash_to_g2(h, msg, (size_t)msg_len);
    snap(2);
    stage_pairing(p, h, sg);
    snap(3);
}

// The sum-of-products lane-group programs (tools/gen_bls_vm3.py) executed with the kernel's lock-step semantics on one tuple:
// part A, the Fp inversion, part C.  Inputs / outputs canonical big-endian like hs_pairing; Fp12 coefficients arrive in w-power order.
int hs_vm3_pairing(const u8* p_xy, const u8* h_xy, const u8* s_xy, u8* out576) {
    std::vector<u32> RA((size_t)ECG_VM3_A_NREG * 13, 0), RC((size_t)ECG_VM3_C_NREG * 13, 0), KA(64 * 13, 0), KC(64 * 13, 0);
    for (int c = 0; c < ECG_VM3_A_NCONST; c++)
        for (int i = 0; i < 13; i++) KA[(size_t)(ECG_VM3_A_CONST_REG[c] - VM3_CONST_BASE) * 13 + i] = ECG_VM3_A_CONST_VAL[c * 13 + i];
    for (int c = 0; c < ECG_VM3_C_NCONST; c++)
        for (int i = 0; i < 13; i++) KC[(size_t)(ECG_VM3_C_CONST_REG[c] - VM3_CONST_BASE) * 13 + i] = ECG_VM3_C_CONST_VAL[c * 13 + i];
    const Vm3Regs A{RA.data(), KA.data()}, Cr{RC.data(), KC.data()};
    A1 p = in_a1(p_xy, 0);
    A2 h = in_a2(h_xy, 0), sg = in_a2(s_xy, 0);
    const Fp in[10] = {p.x, p.y, h.x.c0, h.x.c1, h.y.c0, h.y.c1, sg.x.c0, sg.x.c1, sg.y.c0, sg.y.c1};
    for (int k = 0; k < 10; k++) vm3_store(A, ECG_VM3_A_IN[k], in[k]);
    vm3_run_serial(ECG_VM3_A_PROG, ECG_VM3_A_HDR, ECG_VM3_A_ROUNDS, ECG_VM3_A_LANES, A);
    for (int k = 0; k < 12; k++) vm3_store(Cr, ECG_VM3_C_IN[k], vm3_load(A, ECG_VM3_A_OUT[k]));
    vm3_store(Cr, ECG_VM3_C_IN[12], fp_inv(vm3_load(A, ECG_VM3_A_OUT[12])));
    vm3_store(Cr, ECG_VM3_C_IN[13], fp_zero());
    vm3_run_serial(ECG_VM3_C_PROG, ECG_VM3_C_HDR, ECG_VM3_C_ROUNDS, ECG_VM3_C_LANES, Cr);
    // w-power order g0..g5 = c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2
    Fp12 e;
    Fp2* c[6] = {&e.c0.c0, &e.c1.c0, &e.c0.c1, &e.c1.c1, &e.c0.c2, &e.c1.c2};
    for (int k = 0; k < 6; k++) *c[k] = Fp2{vm3_load(Cr, ECG_VM3_C_OUT[2 * k]), vm3_load(Cr, ECG_VM3_C_OUT[2 * k + 1])};
    out_fp12(e, out576);
    return fp12_is_one(e) ? 1 : 0;
}

// The same programs on the ROW machine (csrc/bls_row.h: one Fp operation across a 16-lane row, limb per lane), with the kernel's
// schedule: every row pair of the tuple computes from the register file as it was before the round (k_row_pair_*: up to the first
// barrier), then every row pair writes (up to the second).  A row pair = the 32 host lanes of rv32.
static void row_run_host(const unsigned int* prog, const unsigned int* hdr, u32 rounds, u32 slots, const RowFile& F) {
    rv32 p_limb;
    for (int l = 0; l < ROW_SIM; l++) p_limb.v[l] = (l & 15) < 13 ? blsc::P[l & 15] : 0u;
    std::vector<RowResult> res(slots / 2);
    std::vector<std::array<rv32, 8>> ws(slots / 2);
    for (u32 r = 0; r < rounds; r++) {
        const u32 n = hdr[r] & 255, nder = (hdr[r] >> 8) & 255;
        for (u32 pr = 0; pr < slots / 2; pr++) {
            rv32 w[8];
            for (int q = 0; q < 8; q++)
                for (int l = 0; l < ROW_SIM; l++) w[q].v[l] = prog[((size_t)r * slots + 2 * pr + (l >> 4)) * VM3_DESC_DW + q];
            res[pr] = row_round_compute(F, n, nder, w, p_limb);
            for (int q = 0; q < 8; q++) ws[pr][q] = w[q];
        }
        for (u32 pr = 0; pr < slots / 2; pr++) {
            rv32 w[8];
            for (int q = 0; q < 8; q++) w[q] = ws[pr][q];
            row_round_store(F, n, nder, w, res[pr]);
        }
    }
}
static void row_file_init(std::vector<u32>& lds, u32 nreg, u32 nconst, const unsigned int* const_reg, const unsigned int* const_val) {
    lds.assign((size_t)(nreg + 64) * ROW_REG_DW, 0xdeadbeefu);  // (unwritten registers are poison: a program never reads them)
    for (u32 i = 0; i < ROW_REG_DW; i++) lds[i] = 0;
    for (u32 c = 0; c < nconst; c++)
        for (u32 k = 0; k < ROW_REG_DW; k++) lds[(size_t)(nreg + const_reg[c] - VM3_CONST_BASE) * ROW_REG_DW + k] = k < 13 ? const_val[c * 13 + k] : 0u;
}
static void row_put_host(std::vector<u32>& lds, u32 reg, const Fp& x) {
    for (u32 k = 0; k < ROW_REG_DW; k++) lds[(size_t)reg * ROW_REG_DW + k] = k < 13 ? x.l[k] : 0u;
}
int hs_row_pairing(const u8* p_xy, const u8* h_xy, const u8* s_xy, u8* out576) {
    g_ecg_column_overflows = 0;
    std::vector<u32> LA, LC;
    row_file_init(LA, ECG_VM3_A_NREG, ECG_VM3_A_NCONST, ECG_VM3_A_CONST_REG, ECG_VM3_A_CONST_VAL);
    row_file_init(LC, ECG_VM3_C_NREG, ECG_VM3_C_NCONST, ECG_VM3_C_CONST_REG, ECG_VM3_C_CONST_VAL);
    A1 p = in_a1(p_xy, 0);
    A2 h = in_a2(h_xy, 0), sg = in_a2(s_xy, 0);
    const Fp in[10] = {p.x, p.y, h.x.c0, h.x.c1, h.y.c0, h.y.c1, sg.x.c0, sg.x.c1, sg.y.c0, sg.y.c1};
    for (int k = 0; k < 10; k++) row_put_host(LA, ECG_VM3_A_IN[k], in[k]);
    row_run_host(ECG_VM3_A_PROG, ECG_VM3_A_HDR, ECG_VM3_A_ROUNDS, ECG_VM3_A_LANES, RowFile{LA.data(), ECG_VM3_A_NREG});
    for (int k = 0; k < 12; k++) row_put_host(LC, ECG_VM3_C_IN[k], row_image_to_fp(LA.data() + (size_t)ECG_VM3_A_OUT[k] * ROW_REG_DW));
    row_put_host(LC, ECG_VM3_C_IN[12], fp_inv(row_image_to_fp(LA.data() + (size_t)ECG_VM3_A_OUT[12] * ROW_REG_DW)));
    row_put_host(LC, ECG_VM3_C_IN[13], fp_zero());
    row_run_host(ECG_VM3_C_PROG, ECG_VM3_C_HDR, ECG_VM3_C_ROUNDS, ECG_VM3_C_LANES, RowFile{LC.data(), ECG_VM3_C_NREG});
    Fp12 e;
    Fp2* c[6] = {&e.c0.c0, &e.c1.c0, &e.c0.c1, &e.c1.c1, &e.c0.c2, &e.c1.c2};
    for (int k = 0; k < 6; k++)
        *c[k] = Fp2{row_image_to_fp(LC.data() + (size_t)ECG_VM3_C_OUT[2 * k] * ROW_REG_DW),
                    row_image_to_fp(LC.data() + (size_t)ECG_VM3_C_OUT[2 * k + 1] * ROW_REG_DW)};
    out_fp12(e, out576);
    if (g_ecg_column_overflows) return -1;  // a lane's 64-bit accumulator overflowed
    return fp12_is_one(e) ? 1 : 0;
}

// one sum of products on the row machine against the one-lane routine: raw limbs in, limbs (<= 2^30) out
u64 hs_row_sumprod_raw(int n, const u32* a, const u32* b, u32* out) {
    g_ecg_column_overflows = 0;
    rv32 av[7], bv[7][13], p_limb;
    for (int l = 0; l < ROW_SIM; l++) p_limb.v[l] = (l & 15) < 13 ? blsc::P[l & 15] : 0u;
    for (int k = 0; k < 7; k++) {
        for (int l = 0; l < ROW_SIM; l++) av[k].v[l] = (k < n && (l & 15) < 13) ? a[13 * k + (l & 15)] : 0u;
        for (int i = 0; i < 13; i++) bv[k][i] = rv_splat(k < n ? b[13 * k + i] : 0u);
    }
    const rv32 r = row_sumprod<7>(av, bv, p_limb);
    for (int i = 0; i < 16; i++) out[i] = r.v[i];
    for (int i = 0; i < 16; i++)
        if (r.v[16 + i] != r.v[i]) return ~0ull;  // both rows of the pair ran the same sum
    return g_ecg_column_overflows;
}

u64 hs_sumprod_raw(int n, const u32* a, const u32* b, u32* out) {
    g_ecg_column_overflows = 0;
    switch (n) {
        case 1: sumprod_raw<1>(a, b, out); break;
        case 2: sumprod_raw<2>(a, b, out); break;
        case 3: sumprod_raw<3>(a, b, out); break;
        case 4: sumprod_raw<4>(a, b, out); break;
        case 5: sumprod_raw<5>(a, b, out); break;
        case 6: sumprod_raw<6>(a, b, out); break;
        case 7: sumprod_raw<7>(a, b, out); break;
        case 8: sumprod_raw<8>(a, b, out); break;
        default: return ~0ull;
    }
    return g_ecg_column_overflows;
}
// the linear helpers of round 3 on raw limbs (13 x 30 bits, the top limb carrying the excess of a lazy value)
// op: 0 fp_sub_dbl(a, b)  1 fp_gs_lin<+1,2,4,4>  2 fp_gs_lin<-1,2,4,4>  3 fp_gs_lin<+1,4,4,4>  4 fp_reduce_below<12,2>(a)
//     5 fp_reduce_below<20,4>(a)  6 fp_reduce_below<14,4>(a)  7 fp_cond_sub(a, 2p)
int hs_fp_lin_raw(int op, const u32* a, const u32* b, u32* out) {
    Fp x, y, r;
    for (int i = 0; i < 13; i++) {
        x.l[i] = a[i];
        y.l[i] = b ? b[i] : 0;
    }
    switch (op) {
        case 0: r = fp_sub_dbl(x, y); break;
        case 1: r = fp_gs_lin<+1, 2, 4, 4>(x, y); break;
        case 2: r = fp_gs_lin<-1, 2, 4, 4>(x, y); break;
        case 3: r = fp_gs_lin<+1, 4, 4, 4>(x, y); break;
        case 4: r = fp_reduce_below<12, 2>(x); break;
        case 5: r = fp_reduce_below<20, 4>(x); break;
        case 6: r = fp_reduce_below<14, 4>(x); break;
        case 7: r = fp_cond_sub(x, blsc::P2); break;
        default: return -1;
    }
    for (int i = 0; i < 13; i++) out[i] = r.l[i];
    return 0;
}
u64 hs_column_overflows() { return g_ecg_column_overflows; }

void hs_census_reset() { g_ecg_fp_mul_count = g_ecg_fp_sqr_count = g_ecg_fp_mad_count = 0; }
void hs_census_read(u64* out) {
    out[0] = g_ecg_fp_mul_count;
    out[1] = g_ecg_fp_sqr_count;
    out[2] = g_ecg_fp_mad_count;
}

int hs_fast_aggregate_verify(const u8* pks48, u32 k, const u8* msg, u64 msg_len, const u8* sig96, int eth) {
    return fav_tuple_serial(pks48, k, msg, (size_t)msg_len, sig96, eth != 0);
}

// n independent K = 1 tuples with 32-byte messages on `threads` host threads (bench.py's "device lane programs on the
// host cores" line; the independent oracle is oracle/bls12_381.py)
void hs_fav_batch_k1(const u8* pks48, const u8* msgs32, const u8* sigs96, u32 n, int threads, u8* status) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([=] {
            for (u32 i = (u32)t; i < n; i += (u32)threads)
                status[i] = fav_tuple_serial(pks48 + 48 * (size_t)i, 1, msgs32 + 32 * (size_t)i, 32, sigs96 + 96 * (size_t)i, false);
        });
    for (auto& x : th) x.join();
}

}  // extern "C"
