// Lane-group ("sum-of-products VM") pairing kernels for gfx950: 16 (Miller loops) / 12 (final exponentiation) lanes share one pairing check; the tower
// arithmetic is a generated straight-line program over an LDS-resident Fp register file in which every operation is a
// sum of products with one Montgomery reduction and every linear step rides along as a derived output of its producer
// (tools/gen_bls_vm3.py, csrc/bls_vm3.h).  This is the e(pk, H(m)) == e(g1, sig) check of
// /root/reference/ethereum-consensus/src/crypto/bls.rs:71,126 for every tuple of a batch.
//
// Why this shape: it is the north_star layout -- Fp12 state in LDS, field operations on LDS-resident
// operands, carries / partner values over cross-lane moves -- with a hot loop of ~40 KB (two compiled sums, one
// interpreter loop) that fits the instruction cache: unlike the lane kernels it does not care whether the box prefetches
// megabytes of straight-line code, it has no private-segment traffic at all, and a batch of a few thousand tuples fills the
// chip (a tuple is 16 lanes wide, so its latency is a fraction of the one-lane chain).
// Own translation unit: the sums are register-allocated per TU.
#include "bls_verify.h"
#include "bls_vm3.h"
#include "bls_vm_host.h"
#ifndef ECG_VM3_PROG_HEADER
#define ECG_VM3_PROG_HEADER "bls_vm3_prog.h"
#endif
#include ECG_VM3_PROG_HEADER

namespace ecg {

// lanes per tuple, per part: the Miller loops keep 8 lane pairs busy, the final exponentiation is Fp12 arithmetic (6 pairs), so
// part C runs 12-lane groups, five tuples to a wave (tools/gen_bls_vm3.py --lanes-c; 23.9 -> 22.3 ms per 65 536 tuples)
constexpr int VM3_G_A = ECG_VM3_A_LANES, VM3_G_C = ECG_VM3_C_LANES;
constexpr int VM3_TPW_A = 64 / VM3_G_A, VM3_TPW_C = 64 / VM3_G_C;  // tuples per wave (= per workgroup)
constexpr u32 XFER3_REGS = 16;          // per tuple: f (12 Fp, w-power order), d, -, 1/d, -
constexpr u32 XFER3_STRIDE = XFER3_REGS * VM3_REG_DW;

static Vm3Desc g_vm3_a_dev[MAX_DEVICES], g_vm3_c_dev[MAX_DEVICES];  // program tables live in each device's memory
#define g_vm3_a g_vm3_a_dev[current_device()]
#define g_vm3_c g_vm3_c_dev[current_device()]

static int upload3(const unsigned int* h, size_t n, const u32** d) {
    u32* p = nullptr;
    ECG_HIP_CHECK(hipMalloc((void**)&p, (n ? n : 1) * 4));
    if (n) ECG_HIP_CHECK(hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice));
    *d = p;
    return ECGPU_SUCCESS;
}

#define VM3_FILL(D, T)                                                                                                       \
    do {                                                                                                                     \
        int rc_;                                                                                                             \
        if ((rc_ = upload3(ECG_VM3_##T##_PROG, (size_t)ECG_VM3_##T##_ROUNDS * ECG_VM3_##T##_LANES * VM3_DESC_DW, &D.prog))) return rc_; \
        if ((rc_ = upload3(ECG_VM3_##T##_HDR, ECG_VM3_##T##_ROUNDS, &D.hdr))) return rc_;                                    \
        if ((rc_ = upload3(ECG_VM3_##T##_CONST_REG, ECG_VM3_##T##_NCONST, &D.const_reg))) return rc_;                        \
        if ((rc_ = upload3(ECG_VM3_##T##_CONST_VAL, (size_t)ECG_VM3_##T##_NCONST * 13, &D.const_val))) return rc_;           \
        D.rounds = ECG_VM3_##T##_ROUNDS;                                                                                     \
        D.nreg = ECG_VM3_##T##_NREG;                                                                                         \
        D.nconst = ECG_VM3_##T##_NCONST;                                                                                     \
        D.nin = ECG_VM3_##T##_NIN;                                                                                           \
        D.nout = ECG_VM3_##T##_NOUT;                                                                                         \
        for (int i = 0; i < ECG_VM3_##T##_NIN; i++) D.in_reg[i] = ECG_VM3_##T##_IN[i];                                       \
        for (int i = 0; i < ECG_VM3_##T##_NOUT; i++) D.out_reg[i] = ECG_VM3_##T##_OUT[i];                                    \
    } while (0)

const Vm3Desc& vm3_program(int part) { return part == 0 ? g_vm3_a : g_vm3_c; }

int init_vm3_tables() {
    static_assert(ECG_VM3_A_NIN == 10 && ECG_VM3_A_NOUT == 14 && ECG_VM3_C_NIN == 14 && ECG_VM3_C_NOUT == 12, "program interface");
    static_assert(VM3_G_A <= 64 && VM3_G_A % 2 == 0 && VM3_G_C <= 64 && VM3_G_C % 2 == 0, "results travel in lane pairs; lanes beyond TPW groups of a wave idle");
    static_assert(ECG_VM3_CONST_BASE == VM3_CONST_BASE, "generator and kernel agree on where the constants start");
    static_assert(ECG_VM3_A_LANES == VM3_SLOTS_A && ECG_VM3_C_LANES == VM3_SLOTS_C, "the row machine (bls_row.hip) runs the same programs");
    static_assert(XFER3_STRIDE == VM3_XFER_STRIDE, "one transfer layout for both machines");
    VM3_FILL(g_vm3_a, A);
    VM3_FILL(g_vm3_c, C);
    return ECGPU_SUCCESS;
}

ECG_D u32 dpp_partner(u32 v) {  // the value of lane ^ 1 (quad_perm [1, 0, 3, 2])
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
}

template <int N>
ECG_D Fp vm3_round_sum(const Vm3Regs& R, const uint4 w01) {
    const u32 w[4] = {w01.x, w01.y, w01.z, w01.w};
    return vm3_sum<N>(R, w);
}

// one lane group (VM3_G lanes) through a whole program; R = the tuple's register file in LDS
template <int VM3_G>
ECG_D void vm3_run(const Vm3Desc& d, const Vm3Regs& R, u32 slot, bool idle = false) {
    // `idle`: a lane beyond the last whole group of the wave (64 is not a multiple of every group size): it runs the rounds
    // with all-zero descriptors -- operands ZERO, no stores
    const uint4* pp = (const uint4*)d.prog + (size_t)slot * 2;
    uint4 w01 = pp[0], w23 = pp[1];  // (loaded unconditionally, then blanked: a select between the two SOURCES becomes a flat load)
    if (idle) w01 = w23 = make_uint4(0, 0, 0, 0);
    u32 h = d.hdr[0];
    for (u32 r = 0; r < d.rounds; r++) {
        const u32 rn = (r + 1 < d.rounds) ? r + 1 : r;
        uint4 n01 = pp[(size_t)rn * VM3_G * 2], n23 = pp[(size_t)rn * VM3_G * 2 + 1];  // next round's descriptor, in flight
        if (idle) n01 = n23 = make_uint4(0, 0, 0, 0);
        const u32 hn = d.hdr[rn];
        const u32 hu = (u32)__builtin_amdgcn_readfirstlane((int)h);
        const u32 n = hu & 255, nder = (hu >> 8) & 255;
        Fp own;
        if (n == 0)  // wave-uniform
            own = vm3_load(R, (w01.x >> 8) & 255);
        else if (n <= 3)
            own = vm3_round_sum<3>(R, w01);
        else if (n <= 4)
            own = vm3_round_sum<4>(R, w01);
        else
            own = vm3_round_sum<7>(R, w01);
        // every operand of the round has been read: results may overwrite registers whose last reader was this round
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        asm volatile("" ::: "memory");
        const u32 dst = w01.x & 255;
        if (n && dst) vm3_store(R, dst, own);
        if (nder) {
            Fp par;
#pragma unroll
            for (int i = 0; i < 13; i++) par.l[i] = dpp_partner(own.l[i]);
            const u32 dw[4] = {w23.x, w23.y, w23.z, w23.w};
            for (u32 k = 0; k < nder; k++) {  // wave-uniform trip count
                const u32 x = dw[k];
                const Fp v = vm3_derive(own, par, (int)(int8_t)(x >> 8), (int)(int8_t)(x >> 16), x >> 24);
                if (x & 255) vm3_store(R, x & 255, v);
            }
        }
        __syncthreads();  // one wave per workgroup: orders this round's LDS writes before the next reads
        w01 = n01;
        w23 = n23;
        h = hn;
    }
}

// register 0 of every tuple = ZERO; the constants once per workgroup, behind the tuples' slices
template <int VM3_G>
ECG_D Vm3Regs vm3_setup(const Vm3Desc& d, u32* lds, u32 lane) {
    constexpr int VM3_TPW = 64 / VM3_G;
    const u32 slot = lane % VM3_G, tl = lane / VM3_G < VM3_TPW ? lane / VM3_G : VM3_TPW - 1;  // idle lanes look at the last slice
    u32* own = lds + tl * d.nreg * VM3_REG_DW;
    u32* consts = lds + VM3_TPW * d.nreg * VM3_REG_DW;
    for (u32 i = slot; i < VM3_REG_DW; i += VM3_G) own[i] = 0;
    for (u32 i = lane; i < d.nconst * VM3_REG_DW; i += 64) {
        const u32 c = i / VM3_REG_DW, k = i % VM3_REG_DW;
        consts[(d.const_reg[c] - VM3_CONST_BASE) * VM3_REG_DW + k] = d.const_val[i];
    }
    return Vm3Regs{own, consts};
}

// part A: Miller loops of e(agg, H) e(-g1, sig) -> f (12 Fp) and the Fp norm d to invert
__global__ void __launch_bounds__(64) k_vm3_pair_a(Vm3Desc d, const A1* agg, const A2* hpts, const A2* sigpts, u32 n, u32* xfer) {
    extern __shared__ u32 vm3_lds[];
    constexpr int VM3_G = VM3_G_A, VM3_TPW = VM3_TPW_A;
    const u32 lane = threadIdx.x, slot = lane % VM3_G, tl = lane / VM3_G;
    const bool idle = tl >= VM3_TPW;
    const u32 tuple = idle ? n : blockIdx.x * VM3_TPW + tl;
    const u32 tc = tuple < n ? tuple : n - 1;
    const Vm3Regs R = vm3_setup<VM3_G>(d, vm3_lds, lane);
    for (u32 k = slot; k < 10 && !idle; k += VM3_G) {
        // inputs in the generator's order: PXY = (x, y) of the aggregate key, then HX, HY, SX, SY (c0, c1 each)
        const u32* w = k == 0   ? agg[tc].x.l
                       : k == 1 ? agg[tc].y.l
                       : k < 4  ? (k == 2 ? hpts[tc].x.c0.l : hpts[tc].x.c1.l)
                       : k < 6  ? (k == 4 ? hpts[tc].y.c0.l : hpts[tc].y.c1.l)
                       : k < 8  ? (k == 6 ? sigpts[tc].x.c0.l : sigpts[tc].x.c1.l)
                                : (k == 8 ? sigpts[tc].y.c0.l : sigpts[tc].y.c1.l);
        for (u32 i = 0; i < VM3_REG_DW; i++) R.own[d.in_reg[k] * VM3_REG_DW + i] = w[i];
    }
    __syncthreads();
    vm3_run<VM3_G>(d, R, slot, idle);
    if (tuple < n)
        for (u32 k = slot; k < 13; k += VM3_G) {
            u32* o = xfer + (size_t)tuple * XFER3_STRIDE + k * VM3_REG_DW;
            for (u32 i = 0; i < VM3_REG_DW; i++) o[i] = R.own[d.out_reg[k] * VM3_REG_DW + i];
        }
}

// the one sequential chain of the pairing check: d -> 1/d, one lane per tuple, register resident
__global__ void __launch_bounds__(BLS_BLOCK) k_vm3_inv(u32* xfer, u32 n) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    Fp* base = (Fp*)(xfer + (size_t)i * XFER3_STRIDE);
    Fp dv = base[12];
    base[14] = fp_inv(dv);
    base[15] = fp_zero();
}

// part C: final exponentiation, == 1 test and the status algebra of fast_aggregate_verify
__global__ void __launch_bounds__(64) k_vm3_pair_c(Vm3Desc d, const u32* xfer, const A1* agg, const u8* st_pk, const u32* pk_off,
                                                   const A2* hpts, const A2* sigpts, const u8* st_dec, const u8* st_grp, const u8* sigs96,
                                                   u32 n, int eth_variant, u8* status_out) {
    extern __shared__ u32 vm3_lds[];
    constexpr int VM3_G = VM3_G_C, VM3_TPW = VM3_TPW_C;
    __shared__ u32 not_one[VM3_TPW];
    const u32 lane = threadIdx.x, slot = lane % VM3_G, tl = lane / VM3_G;
    const bool idle = tl >= VM3_TPW;
    const u32 tuple = idle ? n : blockIdx.x * VM3_TPW + tl;
    const u32 tc = tuple < n ? tuple : n - 1;
    const Vm3Regs R = vm3_setup<VM3_G>(d, vm3_lds, lane);
    for (u32 k = slot; k < 14 && !idle; k += VM3_G) {
        const u32* w = xfer + (size_t)tc * XFER3_STRIDE + (k < 12 ? k : k + 2) * VM3_REG_DW;
        for (u32 i = 0; i < VM3_REG_DW; i++) R.own[d.in_reg[k] * VM3_REG_DW + i] = w[i];
    }
    if (slot == 0 && !idle) not_one[tl] = 0;
    __syncthreads();
    vm3_run<VM3_G>(d, R, slot, idle);
    for (u32 k = slot; k < 12 && !idle; k += VM3_G) {
        const Fp v = vm3_load(R, d.out_reg[k]);
        const bool ok = k == 0 ? fp_eq(v, fp_one()) : fp_is_zero(v);
        if (!ok) not_one[tl] = 1;
    }
    __syncthreads();
    if (slot == 0 && tuple < n) {
        const u32 k = pk_off ? pk_off[tuple + 1] - pk_off[tuple] : 1;
        const bool sig_inf_bytes = sig_is_infinity_bytes(sigs96 + 96 * (size_t)tuple);
        const bool agg_inf = agg[tuple].inf != 0;
        u8 st = combine_fav_status(k, eth_variant != 0, sig_inf_bytes, st_pk[tuple], st_dec[tuple], st_grp[tuple], agg_inf, 0xff);
        if (st == 0xff) {
            // A pair with a point at infinity contributes 1 to the product, and e(P, Q) != 1 for non-zero P in G1, Q in G2
            // (the pairing is non-degenerate on the order-r subgroups; the key was validated, H(m) is cofactor-cleared, the
            // signature passed its group check): the infinity cases are decided without evaluating anything.
            const bool s_inf = sigpts[tuple].inf != 0, h_inf = hpts[tuple].inf != 0;
            if (s_inf || h_inf)
                st = (s_inf && h_inf) ? ECGPU_SUCCESS : ECGPU_VERIFY_FAIL;
            else
                st = not_one[tl] ? ECGPU_VERIFY_FAIL : ECGPU_SUCCESS;
        }
        status_out[tuple] = st;
    }
}

size_t vm3_xfer_bytes(u32 n) { return (size_t)n * XFER3_STRIDE * 4 + 256; }

void vm3_launch_inv(hipStream_t s, u32* xfer, u32 n) {
    hipLaunchKernelGGL(k_vm3_inv, dim3((n + BLS_BLOCK - 1) / BLS_BLOCK), dim3(BLS_BLOCK), 0, s, xfer, n);
}

int vm3_pairing_launch(hipStream_t s, const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts,
                       const u8* st_dec, const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* d_status, u32* xfer) {
    static_assert(sizeof(Fp) == 13 * 4, "register images are read straight from the staged points");
    const dim3 vgrid_a((n + VM3_TPW_A - 1) / VM3_TPW_A), vgrid_c((n + VM3_TPW_C - 1) / VM3_TPW_C);
    // per workgroup: the tuples' register slices + one copy of the constants
    const size_t lds_a = ((size_t)VM3_TPW_A * g_vm3_a.nreg + g_vm3_a.nconst) * VM3_REG_DW * 4,
                 lds_c = ((size_t)VM3_TPW_C * g_vm3_c.nreg + g_vm3_c.nconst) * VM3_REG_DW * 4;
    static bool attr_set[MAX_DEVICES] = {};
    if (!attr_set[current_device()]) {
        ECG_HIP_CHECK(hipFuncSetAttribute((const void*)k_vm3_pair_a, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
        ECG_HIP_CHECK(hipFuncSetAttribute((const void*)k_vm3_pair_c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c + 64));
        attr_set[current_device()] = true;
    }
    {
        ProfScope pa("bls_vm3_a", s);
        hipLaunchKernelGGL(k_vm3_pair_a, vgrid_a, dim3(64), lds_a, s, g_vm3_a, agg, hpts, sigpts, n, xfer);
    }
    {
        ProfScope pi("bls_vm3_inv", s);
        hipLaunchKernelGGL(k_vm3_inv, dim3((n + BLS_BLOCK - 1) / BLS_BLOCK), dim3(BLS_BLOCK), 0, s, xfer, n);
    }
    {
        ProfScope pc("bls_vm3_c", s);
        hipLaunchKernelGGL(k_vm3_pair_c, vgrid_c, dim3(64), lds_c, s, g_vm3_c, (const u32*)xfer, agg, st_pk, pk_off, hpts, sigpts, st_dec,
                           st_grp, sigs96, n, eth_variant, d_status);
    }
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

}  // namespace ecg
