// The pairing check of a SMALL batch on the row machine (csrc/bls_row.h): one workgroup per tuple, one Fp operation per 16-lane
// row, the generated programs of the lane groups (tools/gen_bls_vm3.py) executed round by round with two barriers per round.
// e(pk, H(m)) == e(g1, sig) of /root/reference/ethereum-consensus/src/crypto/bls.rs:71,126 -- the reference's callers verify ONE
// signature per call (crypto/bls.rs:64-77), so the latency of a lone check is the product's latency.
#include "bls_row.h"
#include "bls_verify.h"
#include "bls_vm_host.h"

#include <atomic>
#include <mutex>
#include <vector>

namespace ecg {

// rows per tuple = lane slots of the program (16: Miller loops, 12: final exponentiation)
template <int G>
__device__ __forceinline__ void row_run(const Vm3Desc& d, const RowFile& F, u32 slot, u32 p_limb) {
    const uint4* pp = (const uint4*)d.prog + (size_t)slot * 2;
    uint4 w01 = pp[0], w23 = pp[1];
    u32 h = d.hdr[0];
    for (u32 r = 0; r < d.rounds; r++) {
        const u32 rn = (r + 1 < d.rounds) ? r + 1 : r;
        const uint4 n01 = pp[(size_t)rn * G * 2], n23 = pp[(size_t)rn * G * 2 + 1];  // next round's descriptor, in flight
        const u32 hn = d.hdr[rn];
        const u32 hu = (u32)__builtin_amdgcn_readfirstlane((int)h);
        const u32 n = hu & 255, nder = (hu >> 8) & 255;
        const u32 w[8] = {w01.x, w01.y, w01.z, w01.w, w23.x, w23.y, w23.z, w23.w};
        const RowResult res = row_round_compute(F, n, nder, w, p_limb);
        // every row of the tuple has read its operands: results may overwrite registers read in this round.  (A register allocation
        // without reuse inside a round would make this barrier unnecessary; a build without it runs part A in 0.499 ms against
        // 0.503 -- profiles/r05g_*: the barriers are not what a round costs)
        __syncthreads();
        row_round_store(F, n, nder, w, res);
        __syncthreads();  // ... and are visible to the next round
        w01 = n01;
        w23 = n23;
        h = hn;
    }
}

// register 0 = ZERO, the program's constants behind the registers, every image padded with zero words
__device__ __forceinline__ RowFile row_setup(const Vm3Desc& d, u32* lds) {
    const u32 t = threadIdx.x, nt = blockDim.x;
    for (u32 i = t; i < ROW_REG_DW; i += nt) lds[i] = 0;
    for (u32 i = t; i < d.nconst * ROW_REG_DW; i += nt) {
        const u32 c = i / ROW_REG_DW, k = i % ROW_REG_DW;
        lds[(d.nreg + (d.const_reg[c] - VM3_CONST_BASE)) * ROW_REG_DW + k] = k < VM3_REG_DW ? d.const_val[c * VM3_REG_DW + k] : 0u;
    }
    return RowFile{lds, d.nreg};
}
__device__ __forceinline__ void row_put(const RowFile& F, u32 reg, const u32* limbs13, u32 lane) {  // one row writes one register
    F.lds[reg * ROW_REG_DW + lane] = lane < VM3_REG_DW ? limbs13[lane] : 0u;
}

// part A: Miller loops of e(agg, H) e(-g1, sig) -> f (12 Fp) and the Fp norm d to invert
__global__ void __launch_bounds__(256) k_row_pair_a(Vm3Desc d, const A1* agg, const A2* hpts, const A2* sigpts, u32* xfer) {
    extern __shared__ __attribute__((aligned(16))) u32 row_lds[];
    const u32 slot = threadIdx.x >> 4, lane = threadIdx.x & 15, tuple = blockIdx.x;
    const RowFile F = row_setup(d, row_lds);
    if (slot < 10) {
        // inputs in the generator's order: PXY = (x, y) of the aggregate key, then HX, HY, SX, SY (c0, c1 each)
        const u32 k = slot;
        const u32* w = k == 0   ? agg[tuple].x.l
                       : k == 1 ? agg[tuple].y.l
                       : k < 4  ? (k == 2 ? hpts[tuple].x.c0.l : hpts[tuple].x.c1.l)
                       : k < 6  ? (k == 4 ? hpts[tuple].y.c0.l : hpts[tuple].y.c1.l)
                       : k < 8  ? (k == 6 ? sigpts[tuple].x.c0.l : sigpts[tuple].x.c1.l)
                                : (k == 8 ? sigpts[tuple].y.c0.l : sigpts[tuple].y.c1.l);
        row_put(F, d.in_reg[k], w, lane);
    }
    __syncthreads();
    row_run<VM3_SLOTS_A>(d, F, slot, lane < VM3_REG_DW ? blsc::P[lane] : 0u);
    if (threadIdx.x < 13) {  // exact 30-bit limbs for the inversion kernel and for part C
        const Fp v = row_image_to_fp(F.lds + d.out_reg[threadIdx.x] * ROW_REG_DW);
        u32* o = xfer + (size_t)tuple * VM3_XFER_STRIDE + threadIdx.x * VM3_REG_DW;
        for (u32 i = 0; i < VM3_REG_DW; i++) o[i] = v.l[i];
    }
}

// part C: final exponentiation, == 1 test and the status algebra of fast_aggregate_verify
__global__ void __launch_bounds__(192) k_row_pair_c(Vm3Desc d, const u32* xfer, const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts,
                                                    const A2* sigpts, const u8* st_dec, const u8* st_grp, const u8* sigs96, int eth_variant,
                                                    u8* status_out) {
    extern __shared__ __attribute__((aligned(16))) u32 row_lds[];
    __shared__ u32 not_one;
    const u32 slot = threadIdx.x >> 4, lane = threadIdx.x & 15, tuple = blockIdx.x;
    const RowFile F = row_setup(d, row_lds);
    for (u32 k = slot; k < 14; k += VM3_SLOTS_C) {
        const u32* w = xfer + (size_t)tuple * VM3_XFER_STRIDE + (k < 12 ? k : k + 2) * VM3_REG_DW;
        row_put(F, d.in_reg[k], w, lane);
    }
    if (threadIdx.x == 0) not_one = 0;
    __syncthreads();
    row_run<VM3_SLOTS_C>(d, F, slot, lane < VM3_REG_DW ? blsc::P[lane] : 0u);
    if (threadIdx.x < 12) {
        const Fp v = row_image_to_fp(F.lds + d.out_reg[threadIdx.x] * ROW_REG_DW);
        const bool ok = threadIdx.x == 0 ? fp_eq(v, fp_one()) : fp_is_zero(v);
        if (!ok) not_one = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 k = pk_off ? pk_off[tuple + 1] - pk_off[tuple] : 1;
        const bool sig_inf_bytes = sig_is_infinity_bytes(sigs96 + 96 * (size_t)tuple);
        const bool agg_inf = agg[tuple].inf != 0;
        u8 st = combine_fav_status(k, eth_variant != 0, sig_inf_bytes, st_pk[tuple], st_dec[tuple], st_grp[tuple], agg_inf, 0xff);
        if (st == 0xff) {
            // (a pair with a point at infinity contributes 1, and a single non-degenerate pair cannot be 1: bls_vm3.hip)
            const bool s_inf = sigpts[tuple].inf != 0, h_inf = hpts[tuple].inf != 0;
            if (s_inf || h_inf)
                st = (s_inf && h_inf) ? ECGPU_SUCCESS : ECGPU_VERIFY_FAIL;
            else
                st = not_one ? ECGPU_VERIFY_FAIL : ECGPU_SUCCESS;
        }
        status_out[tuple] = st;
    }
}

// The row machine's copy of a program: register numbers are positions in ITS register file (the program's constants, numbers
// from VM3_CONST_BASE up, sit behind the nreg registers of the tuple), so an operand's image starts at dword 16 x its byte --
// the interpreter does not spend five instructions per operand telling constants from registers.
static Vm3Desc g_row_prog[MAX_DEVICES][2];
// (advisor, round 5: the flag is read outside the mutex by threads making their first small-batch call at the same time -- an
// acquire load pairs with the release store behind the descriptor writes)
static std::atomic<bool> g_row_prog_ready[MAX_DEVICES] = {};
int row_programs() {
    const int dev = current_device();
    if (g_row_prog_ready[dev].load(std::memory_order_acquire)) return ECGPU_SUCCESS;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_row_prog_ready[dev].load(std::memory_order_acquire)) return ECGPU_SUCCESS;
    for (int part = 0; part < 2; part++) {
        Vm3Desc d = vm3_program(part);
        const u32 slots = part == 0 ? VM3_SLOTS_A : VM3_SLOTS_C;
        const size_t n_dw = (size_t)d.rounds * slots * VM3_DESC_DW;
        if (d.nreg + d.nconst > 255) {
            set_last_error("row machine: register file does not fit a byte");
            return ECGPU_ERR_BAD_ARG;
        }
        std::vector<u32> h(n_dw);
        ECG_HIP_CHECK(hipMemcpy(h.data(), d.prog, n_dw * 4, hipMemcpyDeviceToHost));
        auto remap = [&](u32 r) { return r >= VM3_CONST_BASE ? d.nreg + (r - VM3_CONST_BASE) : r; };
        auto remap_word = [&](u32 w, u32 from_byte) {
            u32 o = w;
            for (u32 b = from_byte; b < 4; b++) o = (o & ~(255u << (8 * b))) | (remap((w >> (8 * b)) & 255u) << (8 * b));
            return o;
        };
        for (size_t i = 0; i < n_dw; i += VM3_DESC_DW) {
            for (int q = 0; q < 4; q++) h[i + q] = remap_word(h[i + q], 0);  // dst, a0 .. a6, b0 .. b6 (+ one unused byte)
            // (derived outputs name own registers only: byte 0 of w4 .. w7 is never a constant)
        }
        u32* dp = nullptr;
        ECG_HIP_CHECK(hipMalloc((void**)&dp, n_dw * 4));
        ECG_HIP_CHECK(hipMemcpy(dp, h.data(), n_dw * 4, hipMemcpyHostToDevice));
        d.prog = dp;
        g_row_prog[dev][part] = d;
    }
    g_row_prog_ready[dev].store(true, std::memory_order_release);
    return ECGPU_SUCCESS;
}

int row_pairing_launch(hipStream_t s, const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts, const u8* st_dec,
                       const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* d_status, u32* xfer) {
    if (!n) return ECGPU_SUCCESS;
    int rc = row_programs();
    if (rc) return rc;
    const Vm3Desc &pa = g_row_prog[current_device()][0], &pc = g_row_prog[current_device()][1];
    const size_t lds_a = (size_t)(pa.nreg + pa.nconst) * ROW_REG_DW * 4, lds_c = (size_t)(pc.nreg + pc.nconst) * ROW_REG_DW * 4;
    {
        ProfScope p("bls_row_a", s);
        hipLaunchKernelGGL(k_row_pair_a, dim3(n), dim3(16 * VM3_SLOTS_A), lds_a, s, pa, agg, hpts, sigpts, xfer);
    }
    {
        ProfScope p("bls_row_inv", s);
        vm3_launch_inv(s, xfer, n);
    }
    {
        ProfScope p("bls_row_c", s);
        hipLaunchKernelGGL(k_row_pair_c, dim3(n), dim3(16 * VM3_SLOTS_C), lds_c, s, pc, (const u32*)xfer, agg, st_pk, pk_off, hpts, sigpts,
                           st_dec, st_grp, sigs96, eth_variant, d_status);
    }
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

}  // namespace ecg
