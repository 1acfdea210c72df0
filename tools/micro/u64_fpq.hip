// u64_fpq.hip — round 4, VERDICT item 6: price a u64-engine butterfly whose quotient comes from fp64 on split operands.
//   T = y * w mod q (lazy), y = yh * 2^32 + yl any 64-bit value below 2^63, w < q < 2^60 a twiddle with precomputed
//   w32 = w * 2^32 mod q, c0 = fl(w / q), c32 = fl(w32 / q):
//     y * w  ==  yh * w32 + yl * w  (mod q),   both products below 2^92
//     Q  = floor(fma(yh, c32, yl * c0))         the quotient of their sum by q, exact to +-1 (|error| < 2^-19), below 2^33
//     T  = (yh * w32 + yl * w - Q * q + q) mod 2^64   in [0, 3q)   -- the "+ q" rides the accumulator's initial value
//   low 64 bits through two v_mad_u64_u32 chains (low words with carries into the pair, high words: only their low 32 bits count),
//   7 multiplier instructions, against the Shoup form's 10 (mulhi64 + two low products).  The 33rd bit of Q costs a compare, a
//   select and an add.  Compared in the same harness as tools/micro/u64_bfly.hip (8 butterflies per iteration in registers, two
//   waves per SIMD on every CU): ns per butterfly per wave per SIMD slot, and a host check that the priced code is a correct multiply.
// Build: hipcc --offload-arch=gfx950 -O3 -o u64_fpq u64_fpq.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../reference-seal-backend_amd/csrc/modarith.h"
using namespace he355;

struct TwF { u64 w, w32; double c0, c32; };

__device__ __forceinline__ u64 mul_fpq(u64 y, const TwF &t, u64 q, u64 nq) // y * w mod q in [0, 3q)
{
    const u32 yl = (u32)y, yh = (u32)(y >> 32);
    const double Qd = __builtin_fma((double)yh, t.c32, (double)yl * t.c0);
    const double Qh_d = Qd * 2.3283064365386963e-10;                 // 2^-32
    const u32 Qh = (u32)Qh_d;                                        // 0 or 1
    const u32 Ql = (u32)__builtin_fma(-(double)Qh, 4294967296.0, Qd); // floor(Qd) - Qh * 2^32
    // low words: acc = q + yh * w32_lo + yl * w_lo + Ql * nq_lo   (64-bit, wraps)
    u64 acc = q;
    acc += (u64)yh * (u32)t.w32;
    acc += (u64)yl * (u32)t.w;
    acc += (u64)Ql * (u32)nq;
    // high words: only the low 32 bits of their sum reach bit 32..63
    u32 hi = yh * (u32)(t.w32 >> 32) + yl * (u32)(t.w >> 32) + Ql * (u32)(nq >> 32) + (Qh ? (u32)nq : 0u);
    return acc + ((u64)hi << 32);
}

// lazy Cooley-Tukey butterfly on the wide range: X' = X + T, Y' = X - T + 3q
__device__ __forceinline__ void bfly_fpq(u64 &X, u64 &Y, const TwF &t, u64 q, u64 nq, u64 q3)
{
    const u64 T = mul_fpq(Y, t, q, nq);
    Y = X - T + q3;
    X = X + T;
}

__global__ void __launch_bounds__(256, 2) kb_fpq(u64 *x, const TwF *w, u64 q, int n)
{
    const u64 nq = 0 - q, q3 = 3 * q;
    u64 X[8], Y[8];
    TwF t[4];
    for (int i = 0; i < 8; ++i) { X[i] = x[threadIdx.x + 256 * i]; Y[i] = x[threadIdx.x + 256 * (i + 8)]; }
    for (int i = 0; i < 4; ++i) t[i] = w[threadIdx.x * 4 + i];
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            bfly_fpq(X[i], Y[i], t[i / 2], q, nq, q3);
            bfly_fpq(X[i + 1], Y[i + 1], t[(i / 2 + 1) & 3], q, nq, q3);
            const u64 a = X[i], b = Y[i + 1];
            X[i] = Y[i]; Y[i + 1] = X[i + 1]; Y[i] = a; X[i + 1] = b;
        }
        // keep the values inside the 64-bit lazy window (as kb's reduce-free variant does not: this one masks to 2^62 so that the
        // priced multiply always sees legal inputs; one v_and per value, the same for every variant that would adopt it)
#pragma unroll
        for (int i = 0; i < 8; ++i) { X[i] &= 0x3FFFFFFFFFFFFFFFull; Y[i] &= 0x3FFFFFFFFFFFFFFFull; }
    }
    for (int i = 0; i < 8; ++i) { x[threadIdx.x + 256 * i] = X[i]; x[threadIdx.x + 256 * (i + 8)] = Y[i]; }
}
// the product alone on random inputs, for the host check
__global__ void k_check(const u64 *y, const TwF *w, u64 *out, u64 q, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mul_fpq(y[i], w[i], q, 0 - q);
}
// reference harness of u64_bfly.hip (Shoup form, wide lazy), same masking
__global__ void __launch_bounds__(256, 2) kb_shoup(u64 *x, const Tw16 *w, u64 q, int n)
{
    ArU64 ar; ar.q = q; ar.two_q = 2 * q;
    u64 X[8], Y[8];
    Tw16 t[4];
    for (int i = 0; i < 8; ++i) { X[i] = x[threadIdx.x + 256 * i]; Y[i] = x[threadIdx.x + 256 * (i + 8)]; }
    for (int i = 0; i < 4; ++i) t[i] = w[threadIdx.x * 4 + i];
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            u64 A[2] = {X[i], X[i + 1]}, B[2] = {Y[i], Y[i + 1]};
            Tw16 W[2] = {t[i / 2], t[(i / 2 + 1) & 3]};
            ar.bfly_fwd_lazy_g<2>(A, B, W);
            X[i] = B[0]; X[i + 1] = A[1]; Y[i] = A[0]; Y[i + 1] = B[1];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { X[i] &= 0x3FFFFFFFFFFFFFFFull; Y[i] &= 0x3FFFFFFFFFFFFFFFull; }
    }
    for (int i = 0; i < 8; ++i) { x[threadIdx.x + 256 * i] = X[i]; x[threadIdx.x + 256 * (i + 8)] = Y[i]; }
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const u64 q = 0xFFFFFFFFFFC0001ull;
    // ---- correctness of the priced multiply on 1 M random pairs (inputs below 2^63) ----
    const int nc = 1 << 20;
    u64 *hy = (u64 *)malloc(nc * 8), *ho = (u64 *)malloc(nc * 8);
    TwF *hw = (TwF *)malloc(nc * sizeof(TwF));
    u64 s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < nc; ++i) {
        hy[i] = i < 16 ? (i & 1 ? 0x7FFFFFFFFFFFFFFFull : (u64)i) : rnd() >> 1;
        const u64 w = i < 8 ? q - 1 - i : rnd() % q;
        hw[i].w = w;
        hw[i].w32 = (u64)(((unsigned __int128)w << 32) % q);
        hw[i].c0 = (double)w / (double)q;
        hw[i].c32 = (double)hw[i].w32 / (double)q;
    }
    u64 *dy, *dout; TwF *dw;
    (void)hipMalloc(&dy, nc * 8); (void)hipMalloc(&dout, nc * 8); (void)hipMalloc(&dw, nc * sizeof(TwF));
    (void)hipMemcpy(dy, hy, nc * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw, nc * sizeof(TwF), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(nc / 256), dim3(256), 0, 0, dy, dw, dout, q, nc);
    (void)hipMemcpy(ho, dout, nc * 8, hipMemcpyDeviceToHost);
    int bad = 0; u64 maxv = 0;
    for (int i = 0; i < nc; ++i) {
        const u64 want = (u64)(((unsigned __int128)hy[i] * hw[i].w) % q);
        if (ho[i] % q != want || ho[i] >= 3 * q) { if (bad++ < 4) printf("MISMATCH i=%d y=%llx w=%llx got=%llx want=%llx\n", i, (unsigned long long)hy[i], (unsigned long long)hw[i].w, (unsigned long long)ho[i], (unsigned long long)want); }
        if (ho[i] > maxv) maxv = ho[i];
    }
    printf("fp64-quotient multiply: %d mismatches in %d products; largest result / q = %.3f\n", bad, nc, (double)maxv / (double)q);
    // ---- the price ----
    u64 *x; Tw16 *w16; TwF *wf;
    (void)hipMalloc(&x, 256 * 16 * 8); (void)hipMalloc(&w16, 256 * 4 * 16); (void)hipMalloc(&wf, 256 * 4 * sizeof(TwF));
    (void)hipMemset(x, 0x1a, 256 * 16 * 8); (void)hipMemset(w16, 0x07, 256 * 4 * 16);
    (void)hipMemcpy(wf, hw + 16, 256 * 4 * sizeof(TwF), hipMemcpyHostToDevice);
    const int n = 100000, blocks = p.multiProcessorCount * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which)
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(kb_shoup, dim3(blocks), dim3(256), 0, 0, x, w16, q, n);
            else hipLaunchKernelGGL(kb_fpq, dim3(blocks), dim3(256), 0, 0, x, wf, q, n);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.2f ns per butterfly per wave per SIMD-slot (%.1f ms)\n", which ? "fp64-quotient butterfly (7 multiplies + 5 fp64)" : "Shoup wide-lazy butterfly (product)      ", ms * 1e6 / ((double)n * 8 * 2), ms);
        }
    return bad != 0;
}
