// Lane-group ("field VM") pairing kernels for gfx950: ECG_VM_LANES lanes share one pairing check, the
// tower arithmetic is a generated straight-line program over an LDS-resident Fp register file
// (tools/gen_bls_vm.py, csrc/bls_vm.h).  Replaces the scratch-bound one-lane-per-tuple k_pairing for the
// e(pk, H(m)) == e(g1, sig) check of /root/reference/ethereum-consensus/src/crypto/bls.rs:71,126.
// Own translation unit on purpose: the out-of-line fp_mul / fp_sqr bodies are register-allocated per
// TU, and these kernels want them small (83 VGPRs, no scratch) while the lane kernels of bls.hip
// trade registers for fewer private-segment round trips.
#include "bls_verify.h"
#include "bls_vm.h"
#include "bls_vm_host.h"
#include "bls_vm_prog.h"

namespace ecg {

constexpr int BLS_BLOCK = 64;

// ---- lane-group pairing programs (tools/gen_bls_vm.py) in device memory ---------------------------------
struct VmDesc {
    const u32* prog;       // rounds x ECG_VM_LANES slot words
    const u32* const_reg;  // nconst register numbers
    const u32* const_val;  // nconst x 13 limbs (Montgomery)
    u32 rounds, nreg, nconst, nin, nout;
    u32 in_reg[16], out_reg[16];
};
static VmDesc g_vm_a, g_vm_c;

static int upload_u32(const unsigned int* h, size_t n, const u32** d) {
    u32* p = nullptr;
    ECG_HIP_CHECK(hipMalloc((void**)&p, n * 4));
    ECG_HIP_CHECK(hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice));
    *d = p;
    return ECGPU_SUCCESS;
}

int init_vm_tables() {
    static_assert(ECG_VM_A_NIN <= 16 && ECG_VM_A_NOUT <= 16 && ECG_VM_C_NIN <= 16 && ECG_VM_C_NOUT <= 16, "descriptor arrays");
    int rc;
    if ((rc = upload_u32(ECG_VM_A_PROG, (size_t)ECG_VM_A_ROUNDS * ECG_VM_LANES, &g_vm_a.prog))) return rc;
    if ((rc = upload_u32(ECG_VM_A_CONST_REG, ECG_VM_A_NCONST, &g_vm_a.const_reg))) return rc;
    if ((rc = upload_u32(ECG_VM_A_CONST_VAL, ECG_VM_A_NCONST * 13, &g_vm_a.const_val))) return rc;
    g_vm_a.rounds = ECG_VM_A_ROUNDS;
    g_vm_a.nreg = ECG_VM_A_NREG;
    g_vm_a.nconst = ECG_VM_A_NCONST;
    g_vm_a.nin = ECG_VM_A_NIN;
    g_vm_a.nout = ECG_VM_A_NOUT;
    for (int i = 0; i < ECG_VM_A_NIN; i++) g_vm_a.in_reg[i] = ECG_VM_A_IN[i];
    for (int i = 0; i < ECG_VM_A_NOUT; i++) g_vm_a.out_reg[i] = ECG_VM_A_OUT[i];
    if ((rc = upload_u32(ECG_VM_C_PROG, (size_t)ECG_VM_C_ROUNDS * ECG_VM_LANES, &g_vm_c.prog))) return rc;
    if ((rc = upload_u32(ECG_VM_C_CONST_REG, ECG_VM_C_NCONST, &g_vm_c.const_reg))) return rc;
    if ((rc = upload_u32(ECG_VM_C_CONST_VAL, ECG_VM_C_NCONST * 13, &g_vm_c.const_val))) return rc;
    g_vm_c.rounds = ECG_VM_C_ROUNDS;
    g_vm_c.nreg = ECG_VM_C_NREG;
    g_vm_c.nconst = ECG_VM_C_NCONST;
    g_vm_c.nin = ECG_VM_C_NIN;
    g_vm_c.nout = ECG_VM_C_NOUT;
    for (int i = 0; i < ECG_VM_C_NIN; i++) g_vm_c.in_reg[i] = ECG_VM_C_IN[i];
    for (int i = 0; i < ECG_VM_C_NOUT; i++) g_vm_c.out_reg[i] = ECG_VM_C_OUT[i];
    return ECGPU_SUCCESS;
}

// ---- lane-group pairing kernels: ECG_VM_LANES lanes per tuple, register file in LDS --------------------
constexpr int VM_TPW = 64 / ECG_VM_LANES;  // tuples per wave (= per workgroup)
constexpr u32 XFER_STRIDE = 14;            // per tuple: 12 coefficients of f, d, 1/d

ECG_D void vm_load_consts(const VmDesc& d, u32* R, u32 slot) {
    for (u32 c = slot; c < d.nconst; c += ECG_VM_LANES)
        for (int i = 0; i < 13; i++) R[d.const_reg[c] * 13 + i] = d.const_val[c * 13 + i];
}
ECG_D void vm_run(const VmDesc& d, u32* R, u32 slot) {
    const u32* pp = d.prog + slot;
    u32 ins = pp[0];
    for (u32 r = 0; r < d.rounds; r++) {
        const u32 nxt = (r + 1 < d.rounds) ? pp[(size_t)(r + 1) * ECG_VM_LANES] : 0u;
        Fp out;
        u32 dst;
        if (vm_slot(ins, R, out, dst)) vm_store(R, dst, out);
        __syncthreads();  // one wave per workgroup: orders this round's LDS writes before the next reads
        ins = nxt;
    }
}

// part A: Miller loops of e(agg, H) e(-g1, sig) -> f (12 Fp) and d = the Fp norm to invert
__global__ void __launch_bounds__(64) k_vm_pair_a(VmDesc d, const A1* agg, const A2* hpts, const A2* sigpts, u32 n, u32* xfer) {
    extern __shared__ u32 vm_lds[];
    const u32 lane = threadIdx.x, slot = lane % ECG_VM_LANES, tl = lane / ECG_VM_LANES;
    const u32 tuple = blockIdx.x * VM_TPW + tl;
    const u32 tc = tuple < n ? tuple : n - 1;
    u32* R = vm_lds + tl * d.nreg * 13;
    vm_load_consts(d, R, slot);
    if (slot < 10) {
        const Fp* src = slot < 2 ? (&agg[tc].x + slot) : slot < 6 ? (&hpts[tc].x.c0 + (slot - 2)) : (&sigpts[tc].x.c0 + (slot - 6));
        const u32* w = src->l;
        for (int i = 0; i < 13; i++) R[d.in_reg[slot] * 13 + i] = w[i];
    }
    __syncthreads();
    vm_run(d, R, slot);
    if (slot < 13 && tuple < n) {
        u32* o = xfer + ((size_t)tuple * XFER_STRIDE + slot) * 13;
        for (int i = 0; i < 13; i++) o[i] = R[d.out_reg[slot] * 13 + i];
    }
}

// the one sequential chain of the pairing check: d -> 1/d, one lane per tuple, register resident
__global__ void __launch_bounds__(BLS_BLOCK) k_vm_inv(u32* xfer, u32 n) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    Fp* base = (Fp*)(xfer + (size_t)i * XFER_STRIDE * 13);
    Fp dv = base[12];
    base[13] = fp_inv(dv);
}

// part C: final exponentiation, == 1 test and the status algebra of fast_aggregate_verify
__global__ void __launch_bounds__(64) k_vm_pair_c(VmDesc d, const u32* xfer, const A1* agg, const u8* st_pk, const u32* pk_off,
                                                  const A2* hpts, const A2* sigpts, const u8* st_dec, const u8* st_grp, const u8* sigs96,
                                                  u32 n, int eth_variant, u8* status_out) {
    extern __shared__ u32 vm_lds[];
    __shared__ u32 not_one[VM_TPW];
    const u32 lane = threadIdx.x, slot = lane % ECG_VM_LANES, tl = lane / ECG_VM_LANES;
    const u32 tuple = blockIdx.x * VM_TPW + tl;
    const u32 tc = tuple < n ? tuple : n - 1;
    u32* R = vm_lds + tl * d.nreg * 13;
    vm_load_consts(d, R, slot);
    if (slot < 13) {
        const u32* w = xfer + ((size_t)tc * XFER_STRIDE + (slot < 12 ? slot : 13)) * 13;
        for (int i = 0; i < 13; i++) R[d.in_reg[slot] * 13 + i] = w[i];
    }
    if (slot == 0) not_one[tl] = 0;
    __syncthreads();
    vm_run(d, R, slot);
    if (slot < 12) {
        Fp v = vm_load(R, d.out_reg[slot]);
        bool ok = slot == 0 ? fp_eq(v, fp_one()) : fp_is_zero(v);
        if (!ok) not_one[tl] = 1;
    }
    __syncthreads();
    if (slot == 0 && tuple < n) {
        const u32 k = pk_off ? pk_off[tuple + 1] - pk_off[tuple] : 1;
        const bool sig_inf_bytes = sig_is_infinity_bytes(sigs96 + 96 * (size_t)tuple);
        const bool agg_inf = agg[tuple].inf != 0;
        u8 st = combine_fav_status(k, eth_variant != 0, sig_inf_bytes, st_pk[tuple], st_dec[tuple], st_grp[tuple], agg_inf, 0xff);
        if (st == 0xff) {
            if (sigpts[tuple].inf || hpts[tuple].inf)
                st = VM_NEEDS_LANE_PATH;
            else
                st = not_one[tl] ? ECGPU_VERIFY_FAIL : ECGPU_SUCCESS;
        }
        status_out[tuple] = st;
    }
}


size_t vm_xfer_bytes(u32 n) { return (size_t)n * XFER_STRIDE * 52 + 256; }

int vm_pairing_launch(hipStream_t s, const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts,
                      const u8* st_dec, const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* d_status, u32* xfer) {
    const dim3 vgrid((n + VM_TPW - 1) / VM_TPW);
    {
        ProfScope pa("bls_vm_a", s);
        hipLaunchKernelGGL(k_vm_pair_a, vgrid, dim3(64), (size_t)VM_TPW * g_vm_a.nreg * 52, s, g_vm_a, agg, hpts, sigpts, n, xfer);
    }
    {
        ProfScope pi("bls_vm_inv", s);
        hipLaunchKernelGGL(k_vm_inv, dim3((n + BLS_BLOCK - 1) / BLS_BLOCK), dim3(BLS_BLOCK), 0, s, xfer, n);
    }
    {
        ProfScope pc("bls_vm_c", s);
        hipLaunchKernelGGL(k_vm_pair_c, vgrid, dim3(64), (size_t)VM_TPW * g_vm_c.nreg * 52, s, g_vm_c, (const u32*)xfer, agg, st_pk, pk_off,
                           hpts, sigpts, st_dec, st_grp, sigs96, n, eth_variant, d_status);
    }
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

}  // namespace ecg
