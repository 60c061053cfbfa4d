// Development probe: sustained rate of the field primitives the BLS kernels are built from, as a
// function of resident waves per SIMD.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iethereum_consensus_amd/csrc tools/fpbench.hip -o tools/fpbench
#include <cstdio>
#include <vector>

#include "bls_pairing.h"
#include "sha256.h"
using namespace ecg;

constexpr int ITERS = 2000;

// candidate: Fp2 square as ONE out-of-line call with both Montgomery products in one basic block (ILP x2)
typedef u32 fp2_vec26 __attribute__((ext_vector_type(26)));
static __device__ __attribute__((noinline)) fp2_vec26 fp2_sqr_dual_call(fp2_vec26 v) {
    Fp a0, a1;
    for (int i = 0; i < 13; i++) { a0.l[i] = v[i]; a1.l[i] = v[13 + i]; }
    const Fp t0 = fp_mul_body(fp_add_lazy(a0, a1), fp_sub_lazy(a0, a1));
    const Fp t1 = fp_mul_body(a0, a1);
    const Fp d = fp_add(t1, t1);
    fp2_vec26 o;
    for (int i = 0; i < 13; i++) { o[i] = t0.l[i]; o[13 + i] = d.l[i]; }
    return o;
}
// candidate: Fp2 product as ONE call, second operand handed over through a lane-private LDS slot, three products
// in one basic block
__shared__ u32 g_fp2_arg[26 * 64];
static __device__ __attribute__((noinline)) fp2_vec26 fp2_mul_lds_call(fp2_vec26 v) {
    Fp a0, a1, b0, b1;
    for (int i = 0; i < 13; i++) { a0.l[i] = v[i]; a1.l[i] = v[13 + i]; }
    for (int i = 0; i < 13; i++) { b0.l[i] = g_fp2_arg[i * 64 + threadIdx.x]; b1.l[i] = g_fp2_arg[(13 + i) * 64 + threadIdx.x]; }
    const Fp t0 = fp_mul_body(a0, b0);
    const Fp t1 = fp_mul_body(a1, b1);
    const Fp t2 = fp_mul_body(fp_add_lazy(a0, a1), fp_add_lazy(b0, b1));
    const Fp c0 = fp_sub(t0, t1), c1 = fp_sub(fp_sub(t2, t0), t1);
    fp2_vec26 o;
    for (int i = 0; i < 13; i++) { o[i] = c0.l[i]; o[13 + i] = c1.l[i]; }
    return o;
}

template <int OP>
__global__ void __launch_bounds__(64) k_bench(const Fp* in, Fp* out) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    Fp x = in[t & 63], y = in[(t + 7) & 63];
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) x = fp_mul(x, y);                    // out-of-line call
        if (OP == 1) x = fp_mul_body(x, y);               // inlined body
        if (OP == 2) x = fp_sqr(x);
        if (OP == 3) { x = fp_add(x, y); y = fp_sub(y, x); }   // 2 linear ops
        if (OP == 5) {                                     // Fp2 square, current form: 2 calls + inline linear ops
            Fp2 r = fp2_sqr(Fp2{x, y});
            x = r.c0;
            y = r.c1;
        }
        if (OP == 6) {                                     // Fp2 square, one dual-product call
            fp2_vec26 v;
            for (int k = 0; k < 13; k++) { v[k] = x.l[k]; v[13 + k] = y.l[k]; }
            v = fp2_sqr_dual_call(v);
            for (int k = 0; k < 13; k++) { x.l[k] = v[k]; y.l[k] = v[13 + k]; }
        }
        if (OP == 7) {                                     // Fp2 product, current form: 3 calls + inline linear ops
            Fp2 r = fp2_mul(Fp2{x, y}, Fp2{y, x});
            x = r.c0;
            y = r.c1;
        }
        if (OP == 8) {                                     // Fp2 product, one triple-product call, b through LDS
            fp2_vec26 v;
            for (int k = 0; k < 13; k++) { v[k] = x.l[k]; v[13 + k] = y.l[k]; }
            for (int k = 0; k < 13; k++) { g_fp2_arg[k * 64 + threadIdx.x] = y.l[k]; g_fp2_arg[(13 + k) * 64 + threadIdx.x] = x.l[k]; }
            v = fp2_mul_lds_call(v);
            for (int k = 0; k < 13; k++) { x.l[k] = v[k]; y.l[k] = v[13 + k]; }
        }
        if (OP == 4) {                                     // Fp2 Karatsuba product on (x, y) * (y, x)
            Fp t0 = fp_mul(x, y), t1 = fp_mul(y, x), t2 = fp_mul(fp_add(x, y), fp_add(y, x));
            x = fp_sub(t0, t1);
            y = fp_sub(fp_sub(t2, t0), t1);
        }
    }
    out[t] = fp_add(x, y);
}


// the round-1 forms, kept here for comparison: Karatsuba over reduced Fp products (out-of-line calls)
static __device__ __forceinline__ Fp2 fp2_mul_karatsuba(const Fp2& a, const Fp2& b) {
    Fp t0 = fp_mul(a.c0, b.c0);
    Fp t1 = fp_mul(a.c1, b.c1);
    Fp t2 = fp_mul(fp_add_lazy(a.c0, a.c1), fp_add_lazy(b.c0, b.c1));
    return Fp2{fp_sub(t0, t1), fp_sub(fp_sub(t2, t0), t1)};
}
static __device__ __forceinline__ void fp6_mul_karatsuba(Fp6& r, const Fp2& a0, const Fp2& a1, const Fp2& a2, const Fp2& b0, const Fp2& b1,
                                                         const Fp2& b2) {
    Fp2 t0 = fp2_mul_karatsuba(a0, b0);
    Fp2 t1 = fp2_mul_karatsuba(a1, b1);
    Fp2 t2 = fp2_mul_karatsuba(a2, b2);
    Fp2 m12 = fp2_mul_karatsuba(fp2_add_lazy(a1, a2), fp2_add_lazy(b1, b2));
    Fp2 m01 = fp2_mul_karatsuba(fp2_add_lazy(a0, a1), fp2_add_lazy(b0, b1));
    Fp2 m02 = fp2_mul_karatsuba(fp2_add_lazy(a0, a2), fp2_add_lazy(b0, b2));
    r.c0 = fp2_add(t0, fp2_mul_xi(fp2_sub(fp2_sub(m12, t1), t2)));
    r.c1 = fp2_add(fp2_sub(fp2_sub(m01, t0), t1), fp2_mul_xi(t2));
    r.c2 = fp2_add(fp2_sub(fp2_sub(m02, t0), t2), t1);
}
// Fp6 product chain: OP 0 = Karatsuba over out-of-line Fp products (round-1 form), 1 = schoolbook with lazy reduction
template <int OP>
__global__ void __launch_bounds__(64) k_bench6(const Fp* in, Fp* out) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    Fp6 x, y;
    Fp* xs = (Fp*)&x;
    Fp* ys = (Fp*)&y;
    for (int k = 0; k < 6; k++) {
        xs[k] = in[(t + k) & 63];
        ys[k] = in[(t + 11 * k + 5) & 63];
    }
    for (int i = 0; i < ITERS / 4; i++) {
        if (OP == 0) fp6_mul_karatsuba(x, x.c0, x.c1, x.c2, y.c0, y.c1, y.c2);
        if (OP == 1) fp6_mul_lazy<2, 2>(x, x.c0, x.c1, x.c2, y.c0, y.c1, y.c2);
        if (OP == 2) {  // Fp2 product by two sums of two products
            x.c0 = fp2_mul(x.c0, y.c0);
        }
        if (OP == 3) {
            x.c0 = fp2_mul_karatsuba(x.c0, y.c0);
        }
        if (OP == 4) {  // one product through the sum-of-products body (instruction-level multiply-adds)
            const Fp a[1] = {x.c0.c0}, b[1] = {y.c0.c0};
            x.c0.c0 = fp_sumprod<1>(a, b);
        }
    }
    Fp acc = xs[0];
    for (int k = 1; k < 6; k++) acc = fp_add(acc, xs[k]);
    out[t] = acc;
}
template <int OP>
void run6(const char* name, double mults_per_iter, const Fp* d_in, Fp* d_out) {
    for (int wps : {1, 2}) {
        const int blocks = 256 * 4 * wps;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(k_bench6<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_bench6<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double us_per_op = ms * 1e3 / (ITERS / 4) / wps;
        printf("%-34s waves/SIMD %d  %8.3f ms  %7.3f us SIMD-time per wave-iteration  (%.0f cycles @2.4GHz)  %6.2f T mult/s\n", name, wps, ms,
               us_per_op, us_per_op * 2400, mults_per_iter * (ITERS / 4) * blocks * 64.0 / (ms * 1e-3) / 1e12);
    }
}

// G2 doubling chain: OP 0 = through the out-of-line jac_dbl<Fp2> (references to private memory, as the kernels call it),
// 1 = the same formulas inlined into the loop
template <int OP>
__global__ void __launch_bounds__(64) k_bench_dbl(const Fp* in, Fp* out) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    J2 p;
    Fp* ps = (Fp*)&p;
    for (int k = 0; k < 6; k++) ps[k] = in[(t + 3 * k) & 63];
    for (int i = 0; i < ITERS / 8; i++) {
        if (OP == 0) jac_dbl(p, p);
        if (OP == 1) {
            Fp2 A = fp2_sqr(p.x), B = fp2_sqr(p.y), C = fp2_sqr(B);
            Fp2 D = fp2_dbl(fp2_sub(fp2_sub(fp2_sqr(fp2_add(p.x, B)), A), C));
            Fp2 E = fp2_add(fp2_dbl(A), A), Fq = fp2_sqr(E);
            Fp2 Z3 = fp2_dbl(fp2_mul(p.y, p.z)), X3 = fp2_sub(Fq, fp2_dbl(D));
            Fp2 C8 = fp2_dbl(fp2_dbl(fp2_dbl(C)));
            p.y = fp2_sub(fp2_mul(E, fp2_sub(D, X3)), C8);
            p.x = X3;
            p.z = Z3;
        }
    }
    Fp acc = ps[0];
    for (int k = 1; k < 6; k++) acc = fp_add(acc, ps[k]);
    out[t] = acc;
}
template <int OP>
void run_dbl(const char* name, const Fp* d_in, Fp* d_out) {
    for (int wps : {1, 2}) {
        const int blocks = 256 * 4 * wps;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(k_bench_dbl<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_bench_dbl<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double us_per_op = ms * 1e3 / (ITERS / 8) / wps;
        printf("%-34s waves/SIMD %d  %8.3f ms  %7.3f us SIMD-time per wave-iteration  (%.0f cycles @2.4GHz)\n", name, wps, ms, us_per_op,
               us_per_op * 2400);
    }
}

// Fp12-level operations of the pairing on a register-resident accumulator (everything inlined into the loop)
template <int OP>
__global__ void __launch_bounds__(64) k_bench12(const Fp* in, Fp* out) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    Fp12 f, g;
    Fp* fs = (Fp*)&f;
    Fp* gs = (Fp*)&g;
    for (int k = 0; k < 12; k++) {
        fs[k] = in[(t + k) & 63];
        gs[k] = in[(t + 5 * k + 1) & 63];
    }
    MillerPair m;
    m.px = in[t & 63];
    m.py = in[(t + 1) & 63];
    m.qx = g.c0.c0;
    m.qy = g.c0.c1;
    m.t.x = g.c0.c2;
    m.t.y = g.c1.c0;
    m.t.z = g.c1.c1;
    m.active = 1;
    m.npx = fp_neg_lazy<2>(m.px); m.n3px = fp_add_lazy(fp_add_lazy(m.npx, m.npx), m.npx);
    slot_store_point(0, m.t);
    for (int i = 0; i < ITERS / 16; i++) {
        if (OP == 0) fp12_sqr(f, f);
        if (OP == 1) fp12_mul(f, f, g);
        if (OP == 2) fp12_mul_by_line<2>(f, g.c0.c0, g.c0.c1, g.c0.c2);
        if (OP == 3) fp12_cyclotomic_sqr(f, f);
        if (OP == 4) miller_dbl_step(f, m, 0);  // running point in the lane slots (LDS)
        if (OP == 5) {  // fp12_mul with everything inlined, operands in registers
            Fp6 t0, t1, mm;
            fp6_mul(t0, f.c0, g.c0);
            fp6_mul(t1, f.c1, g.c1);
            fp6_mul_sums(mm, f.c0, f.c1, g.c0, g.c1);
            fp12_karatsuba_combine(f.c0, f.c1, mm, t0, t1);
        }
        if (OP == 6) fp12_cyclotomic_sqr_inl(f, f);
    }
    Fp acc = fs[0];
    for (int k = 1; k < 12; k++) acc = fp_add(acc, fs[k]);
    acc = fp_add(acc, m.t.x.c0);
    out[t] = acc;
}
template <int OP>
void run12(const char* name, double mults, const Fp* d_in, Fp* d_out) {
    for (int wps : {1}) {
        const int blocks = 256 * 4 * wps;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(k_bench12<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_bench12<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double us_per_op = ms * 1e3 / (ITERS / 16) / wps;
        printf("%-34s waves/SIMD %d  %8.3f ms  %7.3f us SIMD-time per wave-iteration  (%.0f cycles @2.4GHz)  %6.2f T mult/s\n", name, wps, ms,
               us_per_op, us_per_op * 2400, mults * (ITERS / 16) * blocks * 64.0 / (ms * 1e-3) / 1e12);
    }
}

// hash64 chain in registers: the unit of the Merkle roofline
__global__ void __launch_bounds__(64) k_hash(const Fp* in, Fp* out) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    Node a, b;
    for (int i = 0; i < 8; i++) {
        a.w[i] = in[t & 63].l[i] + t;
        b.w[i] = in[(t + 5) & 63].l[i] ^ t;
    }
    for (int i = 0; i < ITERS / 2; i++) {
        a = hash64(a, b);
        b = hash64(b, a);
    }
    for (int i = 0; i < 8; i++) out[t].l[i] = a.w[i] ^ b.w[i];
}
void run_hash(const Fp* d_in, Fp* d_out) {
    for (int wps : {1, 2, 4, 5, 8}) {
        const int blocks = 256 * 4 * wps;
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k_hash, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_hash, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double hashes = (double)ITERS * blocks * 64.0;
        printf("hash64 chain           waves/SIMD %d  %8.3f ms  %7.2f G hash64/s  %7.3f us per sequential hash64  (%.0f cycles SIMD-time per wave-hash)\n", wps,
               ms, hashes / (ms * 1e-3) / 1e9, ms * 1e3 / ITERS, ms * 1e3 / ITERS / wps * 2400);
    }
}

template <int OP>
void run(const char* name, double mults_per_iter, double lin_per_iter, const Fp* d_in, Fp* d_out) {
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * 4 * wps;  // one wave per block
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double wave_ops = (double)ITERS;  // per wave
        const double us_per_op = ms * 1e3 / wave_ops / wps;  // SIMD time per wave-level iteration
        printf("%-22s waves/SIMD %d  %8.3f ms  %7.3f us SIMD-time per wave-iteration  (%.0f cycles @2.4GHz)", name, wps, ms, us_per_op,
               us_per_op * 2400);
        if (mults_per_iter > 0) printf("  %6.2f T mult/s", mults_per_iter * ITERS * blocks * 64.0 / (ms * 1e-3) / 1e12);
        if (lin_per_iter > 0) printf("  %.0f cycles per linear op", us_per_op * 2400 / lin_per_iter);
        printf("\n");
    }
}

int main() {
    std::vector<Fp> h(64);
    for (int i = 0; i < 64; i++)
        for (int k = 0; k < 13; k++) h[i].l[k] = (0x12345u * (i + 3) + 0x9e3779u * (k + 1)) & (k == 12 ? 0xfffff : FP_MASK);
    Fp *d_in, *d_out;
    hipMalloc(&d_in, 64 * sizeof(Fp));
    hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(Fp));
    hipMemcpy(d_in, h.data(), 64 * sizeof(Fp), hipMemcpyHostToDevice);
    run_hash(d_in, d_out);
    run<0>("fp_mul (call)", 351, 0, d_in, d_out);
    run<1>("fp_mul (inline)", 351, 0, d_in, d_out);
    run<2>("fp_sqr (call)", 273, 0, d_in, d_out);
    run<3>("fp_add + fp_sub", 0, 2, d_in, d_out);
    run<4>("fp2 product", 1053, 0, d_in, d_out);
    run<5>("fp2_sqr (2 calls)", 702, 0, d_in, d_out);
    run<6>("fp2_sqr (dual call)", 702, 0, d_in, d_out);
    run<7>("fp2_mul (3 calls)", 1053, 0, d_in, d_out);
    run<8>("fp2_mul (LDS-arg call)", 1053, 0, d_in, d_out);
    run12<0>("fp12_sqr (registers)", 2 * (36 * 169 + 6 * 195), d_in, d_out);
    run12<1>("fp12_mul (registers / call)", 3 * (36 * 169 + 6 * 195), d_in, d_out);
    run12<5>("fp12_mul (registers, inlined)", 3 * (36 * 169 + 6 * 195), d_in, d_out);
    run12<6>("fp12_cyclotomic_sqr (inlined)", 3 * (10 * 169 + 4 * 195), d_in, d_out);
    run12<2>("fp12_mul_by_line (registers)", 72 * 169 + 12 * 195, d_in, d_out);
    run12<3>("fp12_cyclotomic_sqr (call)", 3 * (10 * 169 + 4 * 195), d_in, d_out);
    run12<4>("miller_dbl_step incl. line mul", 0, d_in, d_out);
    run_dbl<0>("G2 doubling (out-of-line jac_dbl)", d_in, d_out);
    run_dbl<1>("G2 doubling (inline formulas)", d_in, d_out);
    run6<4>("fp_mul as fp_sumprod<1> (inline)", 364, d_in, d_out);
    run6<3>("fp2_mul Karatsuba (3 calls)", 1053 + 39, d_in, d_out);
    run6<2>("fp2_mul 2 sums of 2 products", 1040 + 26, d_in, d_out);
    run6<0>("fp6_mul Karatsuba (18 calls)", 18 * 364, d_in, d_out);
    run6<1>("fp6_mul schoolbook lazy (6 sums)", 36 * 169 + 6 * 195, d_in, d_out);
    return 0;
}
