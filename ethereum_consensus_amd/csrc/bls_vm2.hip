// Lane-group ("Fp2 VM") pairing kernels for gfx950: ECG_VM2_LANES lanes share one pairing check; the
// tower arithmetic is a generated straight-line program over an LDS-resident Fp2 register file
// (tools/gen_bls_vm2.py, csrc/bls_vm2.h).  This is the e(pk, H(m)) == e(g1, sig) check of
// /root/reference/ethereum-consensus/src/crypto/bls.rs:71,126 for every tuple of a batch.
//
// Why this shape: an Fp12 is 624 B, so a lane-per-tuple pairing lives in scratch and is
// HBM-bound; with the state in LDS (a few KB per tuple) a CU holds only tens of tuples, so each
// tuple must feed several lanes.  Fp2 is the unit of work: 3 Montgomery products + 5 additions per
// lane per round keep the LDS round trip at a few % of a round, so one wave per SIMD suffices.
// Own translation unit: the out-of-line fp_mul body is register-allocated per TU.
#include "bls_verify.h"
#include "bls_vm2.h"
#include "bls_vm_host.h"
#ifndef ECG_VM2_PROG_HEADER
#define ECG_VM2_PROG_HEADER "bls_vm2_prog.h"  // tools/build_vm2_variant.sh substitutes other generator settings
#endif
#include ECG_VM2_PROG_HEADER

namespace ecg {

constexpr int VM2_G = ECG_VM2_LANES;
constexpr int VM2_TPW = 64 / VM2_G;     // tuples per wave (= per workgroup)
constexpr u32 XFER2_REGS = 8;           // per tuple: 6 coefficients of f, (d, 0), (1/d, -)
constexpr u32 XFER2_STRIDE = XFER2_REGS * VM2_REG_DW;

struct Vm2Desc {
    const u32* prog;       // rounds x LANES slot words
    const u32* cls;        // rounds class words
    const u32* const_reg;  // nconst register numbers
    const u32* const_val;  // nconst x 26 limbs (Montgomery)
    u32 rounds, nreg, nconst, nin, nout;
    u32 in_reg[8], out_reg[8];
};
static Vm2Desc g_vm2_a_dev[MAX_DEVICES], g_vm2_c_dev[MAX_DEVICES];  // program tables live in each device's memory
#define g_vm2_a g_vm2_a_dev[current_device()]
#define g_vm2_c g_vm2_c_dev[current_device()]

static int upload_u32(const unsigned int* h, size_t n, const u32** d) {
    u32* p = nullptr;
    ECG_HIP_CHECK(hipMalloc((void**)&p, (n ? n : 1) * 4));
    if (n) ECG_HIP_CHECK(hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice));
    *d = p;
    return ECGPU_SUCCESS;
}
static int upload_u8_as_u32(const unsigned char* h, size_t n, const u32** d) {
    std::vector<unsigned int> w(n);
    for (size_t i = 0; i < n; i++) w[i] = h[i];
    return upload_u32(w.data(), n, d);
}

#define VM2_FILL(D, T)                                                                                         \
    do {                                                                                                       \
        int rc_;                                                                                               \
        if ((rc_ = upload_u32(ECG_VM2_##T##_PROG, (size_t)ECG_VM2_##T##_ROUNDS * ECG_VM2_LANES, &D.prog))) return rc_; \
        if ((rc_ = upload_u8_as_u32(ECG_VM2_##T##_CLS, ECG_VM2_##T##_ROUNDS, &D.cls))) return rc_;             \
        if ((rc_ = upload_u32(ECG_VM2_##T##_CONST_REG, ECG_VM2_##T##_NCONST, &D.const_reg))) return rc_;       \
        if ((rc_ = upload_u32(ECG_VM2_##T##_CONST_VAL, (size_t)ECG_VM2_##T##_NCONST * 26, &D.const_val))) return rc_; \
        D.rounds = ECG_VM2_##T##_ROUNDS;                                                                       \
        D.nreg = ECG_VM2_##T##_NREG;                                                                           \
        D.nconst = ECG_VM2_##T##_NCONST;                                                                       \
        D.nin = ECG_VM2_##T##_NIN;                                                                             \
        D.nout = ECG_VM2_##T##_NOUT;                                                                           \
        for (int i = 0; i < ECG_VM2_##T##_NIN; i++) D.in_reg[i] = ECG_VM2_##T##_IN[i];                         \
        for (int i = 0; i < ECG_VM2_##T##_NOUT; i++) D.out_reg[i] = ECG_VM2_##T##_OUT[i];                      \
    } while (0)

int init_vm2_tables() {
    static_assert(ECG_VM2_A_NIN == 5 && ECG_VM2_A_NOUT == 7 && ECG_VM2_C_NIN == 7 && ECG_VM2_C_NOUT == 6, "program interface");
    VM2_FILL(g_vm2_a, A);
    VM2_FILL(g_vm2_c, C);
    return ECGPU_SUCCESS;
}

ECG_D void vm2_load_consts(const Vm2Desc& d, u32* R, u32 slot) {
    for (u32 c = slot; c < d.nconst; c += VM2_G)
        for (u32 i = 0; i < VM2_REG_DW; i++) R[d.const_reg[c] * VM2_REG_DW + i] = d.const_val[c * VM2_REG_DW + i];
}

ECG_D void vm2_run(const Vm2Desc& d, u32* R, u32 slot) {
    const u32* pp = d.prog + slot;
    u32 ins = pp[0];
    u32 cls = d.cls[0];
    for (u32 r = 0; r < d.rounds; r++) {
        const u32 rn = (r + 1 < d.rounds) ? r + 1 : r;
        const u32 nxt = pp[(size_t)rn * VM2_G];
        const u32 cls_n = d.cls[rn];
        Fp2 out;
        u32 dst;
        if (vm2_slot(ins, (u32)__builtin_amdgcn_readfirstlane((int)cls), R, out, dst)) vm2_store(R, dst, out);
        __syncthreads();  // one wave per workgroup: orders this round's LDS writes before the next reads
        ins = nxt;
        cls = cls_n;
    }
}

// part A: Miller loops of e(agg, H) e(-g1, sig) -> f (6 Fp2) and the Fp norm d to invert
__global__ void __launch_bounds__(64) k_vm2_pair_a(Vm2Desc d, const A1* agg, const A2* hpts, const A2* sigpts, u32 n, u32* xfer) {
    extern __shared__ u32 vm2_lds[];
    const u32 lane = threadIdx.x, slot = lane % VM2_G, tl = lane / VM2_G;
    const u32 tuple = blockIdx.x * VM2_TPW + tl;
    const u32 tc = tuple < n ? tuple : n - 1;
    u32* R = vm2_lds + tl * d.nreg * VM2_REG_DW;
    vm2_load_consts(d, R, slot);
    for (u32 k = slot; k < 5; k += VM2_G) {
        // PXY = (x, y) of the aggregate key; HX, HY, SX, SY = affine G2 coordinates
        const u32* w = k == 0   ? agg[tc].x.l
                       : k == 1 ? hpts[tc].x.c0.l
                       : k == 2 ? hpts[tc].y.c0.l
                       : k == 3 ? sigpts[tc].x.c0.l
                                : sigpts[tc].y.c0.l;
        for (u32 i = 0; i < VM2_REG_DW; i++) R[d.in_reg[k] * VM2_REG_DW + i] = w[i];
    }
    __syncthreads();
    vm2_run(d, R, slot);
    if (tuple < n)
        for (u32 k = slot; k < 7; k += VM2_G) {
            u32* o = xfer + (size_t)tuple * XFER2_STRIDE + k * VM2_REG_DW;
            for (u32 i = 0; i < VM2_REG_DW; i++) o[i] = R[d.out_reg[k] * VM2_REG_DW + i];
        }
}

// the one sequential chain of the pairing check: d -> 1/d, one lane per tuple, register resident
__global__ void __launch_bounds__(BLS_BLOCK) k_vm2_inv(u32* xfer, u32 n) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    Fp* base = (Fp*)(xfer + (size_t)i * XFER2_STRIDE);
    Fp dv = base[12];
    base[14] = fp_inv(dv);
}

// part C: final exponentiation, == 1 test and the status algebra of fast_aggregate_verify
__global__ void __launch_bounds__(64) k_vm2_pair_c(Vm2Desc d, const u32* xfer, const A1* agg, const u8* st_pk, const u32* pk_off,
                                                   const A2* hpts, const A2* sigpts, const u8* st_dec, const u8* st_grp, const u8* sigs96,
                                                   u32 n, int eth_variant, u8* status_out) {
    extern __shared__ u32 vm2_lds[];
    __shared__ u32 not_one[VM2_TPW];
    const u32 lane = threadIdx.x, slot = lane % VM2_G, tl = lane / VM2_G;
    const u32 tuple = blockIdx.x * VM2_TPW + tl;
    const u32 tc = tuple < n ? tuple : n - 1;
    u32* R = vm2_lds + tl * d.nreg * VM2_REG_DW;
    vm2_load_consts(d, R, slot);
    for (u32 k = slot; k < 7; k += VM2_G) {
        const u32* w = xfer + (size_t)tc * XFER2_STRIDE + (k < 6 ? k : 7) * VM2_REG_DW;
        for (u32 i = 0; i < VM2_REG_DW; i++) R[d.in_reg[k] * VM2_REG_DW + i] = w[i];
    }
    if (slot == 0) not_one[tl] = 0;
    __syncthreads();
    vm2_run(d, R, slot);
    for (u32 k = slot; k < 6; k += VM2_G) {
        Fp2 v = vm2_load(R, d.out_reg[k]);
        bool ok = (k == 0 ? fp_eq(v.c0, fp_one()) : fp_is_zero(v.c0)) && fp_is_zero(v.c1);
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

size_t vm2_xfer_bytes(u32 n) { return (size_t)n * XFER2_STRIDE * 4 + 256; }

int vm2_pairing_launch(hipStream_t s, const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts,
                       const u8* st_dec, const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* d_status, u32* xfer) {
    static_assert(sizeof(A1) >= 26 * 4 && sizeof(Fp2) == 26 * 4, "register images are read straight from the staged points");
    const dim3 vgrid((n + VM2_TPW - 1) / VM2_TPW);
    {
        ProfScope pa("bls_vm_a", s);
        hipLaunchKernelGGL(k_vm2_pair_a, vgrid, dim3(64), (size_t)VM2_TPW * g_vm2_a.nreg * VM2_REG_DW * 4, s, g_vm2_a, agg, hpts, sigpts, n,
                           xfer);
    }
    {
        ProfScope pi("bls_vm_inv", s);
        hipLaunchKernelGGL(k_vm2_inv, dim3((n + BLS_BLOCK - 1) / BLS_BLOCK), dim3(BLS_BLOCK), 0, s, xfer, n);
    }
    {
        ProfScope pc("bls_vm_c", s);
        hipLaunchKernelGGL(k_vm2_pair_c, vgrid, dim3(64), (size_t)VM2_TPW * g_vm2_c.nreg * VM2_REG_DW * 4, s, g_vm2_c, (const u32*)xfer, agg,
                           st_pk, pk_off, hpts, sigpts, st_dec, st_grp, sigs96, n, eth_variant, d_status);
    }
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

}  // namespace ecg
