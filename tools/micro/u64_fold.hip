// u64_fold.hip — round 5, VERDICT item 2: price the u64 engine's multiply for primes q = 2^60 - c (c < 2^24: every 60-bit prime the
// reference's parameter rule produces, seal_context.cpp:79-82,107-110) as a FOLD reduction (2^60 == c mod q) instead of Shoup's quotient.
//   fold128 : P = x * w as 128 bits (4 multiplier ops), hi = P >> 60, r = lo60 + (hi * c folded once more)    -- 7 multiplier ops, no table
//   fold2c  : w2 = w * 2^32 mod q beside w (a 16-byte entry as today's {w, Shoup quotient}); S = x0 * w + x1 * w2 < 2^93,
//             r = lo60(S) + (S >> 61) * 2c + bit60(S) * c  < 2^60 + 2^57                                      -- 6 multiplier ops
//   mac128  : key multiply-accumulate with the reduction DEFERRED: acc128 += x * key (4 multiplier ops + a 128-bit add), one fold per
//             16 digits instead of one Shoup product per digit -- no key quotients at all
// against the product's Shoup wide-lazy butterfly / lazy MAC in the same harness as tools/micro/u64_bfly.hip (8 butterflies per
// iteration in registers, two waves per SIMD on every CU).  The host half checks the fold multiplies bit for bit against a 128-bit
// remainder on random operands and on the ends of the lazy ranges.
// Build: hipcc --offload-arch=gfx950 -O3 -o u64_fold u64_fold.hip        Run: ./u64_fold [host-check products, default 1e8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../reference-seal-backend_amd/csrc/modarith.h"
using namespace he355;

#define HD __host__ __device__ __forceinline__

struct TwC { u64 w, w2; }; // w and w * 2^32 mod q

// x * w mod q, any 64-bit x, w < q = 2^60 - c; result < 2^60 + 2^33 c
HD u64 mul_fold2c(u64 x, u64 w, u64 w2, u32 c, u32 c2 /* 2c */)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    const u64 u = (u64)x0 * (u32)w;
    const u64 u2 = (u64)x1 * (u32)w2 + (u32)u;
    const u64 m1 = (u64)x0 * (u32)(w >> 32) + (u >> 32);
    const u64 m2 = (u64)x1 * (u32)(w2 >> 32) + (u2 >> 32);
    const u64 up = m1 + m2; // S = up * 2^32 + lo32(u2) < 2^93
    const u32 h = (u32)(up >> 29), b = (u32)(up >> 28) & 1u;
    const u64 lo60 = ((up & 0x0FFFFFFFull) << 32) | (u32)u2;
    return (u64)h * c2 + ((u64)b * c + lo60);
}
// the same with bit 60 left in the low part: one multiplier op and one bit-field op fewer, result < 2^61 + 2^32 * 2c
HD u64 mul_fold2c_wide(u64 x, u64 w, u64 w2, u32 c2)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    const u64 u = (u64)x0 * (u32)w;
    const u64 u2 = (u64)x1 * (u32)w2 + (u32)u;
    const u64 m1 = (u64)x0 * (u32)(w >> 32) + (u >> 32);
    const u64 m2 = (u64)x1 * (u32)(w2 >> 32) + (u2 >> 32);
    const u64 up = m1 + m2;
    const u32 h = (u32)(up >> 29);
    const u64 lo61 = ((up & 0x1FFFFFFFull) << 32) | (u32)u2;
    return (u64)h * c2 + lo61;
}
// full product, two folds; result < 2^61 + 2^52
HD u64 mul_fold128(u64 x, u64 w, u32 c)
{
    const u128 P = (u128)x * w;
    const u64 lo = (u64)P & 0x0FFFFFFFFFFFFFFFull;
    const u64 hi = (u64)(P >> 60);
    const u128 hc = (u128)hi * c; // < 2^88
    const u64 l2 = (u64)hc & 0x0FFFFFFFFFFFFFFFull;
    const u32 h2 = (u32)(hc >> 60); // < 2^28
    return (u64)h2 * c + lo + l2;
}
// 128-bit accumulator: acc += x * k
HD void mac128(u128 &acc, u64 x, u64 k) { acc += (u128)x * k; }
// acc (any 128-bit value) -> [0, 2^60 + 2^57): fold the top 68 bits
HD u64 fold_acc128(u128 acc, u32 c, u32 c2)
{
    // acc = H * 2^60 + lo60, H < 2^68: H * c < 2^92 -> again
    const u64 lo = (u64)acc & 0x0FFFFFFFFFFFFFFFull;
    const u128 H = acc >> 60;
    const u128 hc = H * c; // < 2^92
    const u64 l2 = (u64)hc & 0x0FFFFFFFFFFFFFFFull;
    const u64 h2 = (u64)(hc >> 60); // < 2^32
    u64 r = h2 * c + lo + l2;        // < 2^56 + 2^61
    const u64 h3 = r >> 60;          // <= 2
    return (r & 0x0FFFFFFFFFFFFFFFull) + h3 * c;
}

// ---- device harness -----------------------------------------------------------------------------------------------------------
template <int MODE> __global__ void __launch_bounds__(256, 2) kb(u64 *x, const TwC *w, u64 q, int n)
{
    ArU64 ar; ar.q = q; ar.two_q = 2 * q;
    const u32 c = (u32)((1ull << 60) - q), c2 = 2 * c;
    u64 X[8], Y[8];
    TwC t[4];
    for (int i = 0; i < 8; ++i) { X[i] = x[threadIdx.x + 256 * i]; Y[i] = x[threadIdx.x + 256 * (i + 8)]; }
    for (int i = 0; i < 4; ++i) t[i] = w[threadIdx.x * 4 + i];
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const TwC &tw = t[(i / 2 + (i & 1)) & 3];
            u64 v;
            if (MODE == 0) v = mul_shoup_lazy_uq(Y[i], tw.w, tw.w2, q);
            else if (MODE == 1) v = mul_fold2c(Y[i], tw.w, tw.w2, c, c2);
            else if (MODE == 2) v = mul_fold2c_wide(Y[i], tw.w, tw.w2, c2);
            else v = mul_fold128(Y[i], tw.w, c);
            const u64 a = X[i];
            X[i] = a + v;
            Y[i] = a + (MODE == 2 || MODE == 3 ? 3 * q : ar.two_q) - v;
        }
#pragma unroll
        for (int i = 0; i < 8; i += 2) { const u64 a = X[i], b = Y[i + 1]; X[i] = Y[i]; Y[i + 1] = X[i + 1]; Y[i] = a; X[i + 1] = b; }
        // keep the values inside the 64-bit lazy window: mask to 2^62 (the same in every mode; two instructions per value)
#pragma unroll
        for (int i = 0; i < 8; ++i) { X[i] &= 0x3FFFFFFFFFFFFFFFull; Y[i] &= 0x3FFFFFFFFFFFFFFFull; }
    }
    for (int i = 0; i < 8; ++i) { x[threadIdx.x + 256 * i] = X[i]; x[threadIdx.x + 256 * (i + 8)] = Y[i]; }
}

// MAC harness: 16 "digits" per iteration into 8 accumulator pairs' worth of elements (x changes per digit by a cheap rotation)
template <int MODE> __global__ void __launch_bounds__(256, 2) km(u64 *x, const TwC *w, u64 q, int n)
{
    ArU64 ar; ar.q = q; ar.two_q = 2 * q;
    const u32 c = (u32)((1ull << 60) - q), c2 = 2 * c;
    u64 X[8];
    TwC k[8];
    for (int i = 0; i < 8; ++i) { X[i] = x[threadIdx.x + 256 * i]; k[i] = w[(threadIdx.x * 8 + i) & 1023]; }
    u64 out[8];
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {
            u64 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = out[i] & 0x0FFFFFFFFFFFFFFFull;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) ar.acc_mac_lazy(acc[i], X[(i + j) & 7] + j, k[i].w, k[i].w2);
                if (j % 6 == 5) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = ar.acc_reduce(acc[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) out[i] = ar.acc_reduce(acc[i]);
        } else {
            u128 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = out[i] & 0x0FFFFFFFFFFFFFFFull;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) mac128(acc[i], (X[(i + j) & 7] + j) & 0x3FFFFFFFFFFFFFFFull, k[i].w);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) out[i] = fold_acc128(acc[i], c, c2);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) X[i] ^= out[i] >> 3;
    }
    for (int i = 0; i < 8; ++i) x[threadIdx.x + 256 * i] = out[i] + X[i];
}

// ---- host check ---------------------------------------------------------------------------------------------------------------
static u64 rng_state = 0x9E3779B97F4A7C15ull;
static u64 rnd() { u64 z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static int host_check(long n)
{
    const u64 primes[3] = {0xFFFFFFFFFFC0001ull, 0xFFFFFFFFF840001ull, 0xFFFFFFFFFFFC001ull};
    long bad = 0;
    double worst2c = 0, worstw = 0, worst128 = 0, worstacc = 0;
    for (int pi = 0; pi < 3; ++pi) {
        const u64 q = primes[pi];
        const u32 c = (u32)((1ull << 60) - q), c2 = 2 * c;
        const u64 edge_x[] = {0, 1, q - 1, q, 2 * q, 4 * q - 1, 8 * q, 16 * q - 1, ~0ull, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull};
        const u64 edge_w[] = {0, 1, q - 1, q / 2, 0xFFFFFFFFull, 0x100000000ull, (1ull << 59) + 12345};
        const long ne = sizeof(edge_x) / 8 * (sizeof(edge_w) / 8);
        for (long i = 0; i < n / 3 + ne; ++i) {
            u64 x, w;
            if (i < ne) { x = edge_x[i / (long)(sizeof(edge_w) / 8)]; w = edge_w[i % (long)(sizeof(edge_w) / 8)]; }
            else { x = rnd(); w = rnd() % q; if ((i & 15) == 0) x |= 0xFFFFFFFF00000000ull; if ((i & 15) == 1) x &= 0xFFFFFFFFull; }
            const u64 w2 = (u64)(((u128)w << 32) % q);
            const u64 want = (u64)(((u128)x * w) % q);
            const u64 r1 = mul_fold2c(x, w, w2, c, c2), r2 = mul_fold2c_wide(x, w, w2, c2), r3 = mul_fold128(x, w, c);
            if (r1 % q != want || r2 % q != want || r3 % q != want) { if (bad < 5) printf("MISMATCH q=%llx x=%llx w=%llx\n", (unsigned long long)q, (unsigned long long)x, (unsigned long long)w); ++bad; }
            if ((double)r1 / q > worst2c) worst2c = (double)r1 / q;
            if ((double)r2 / q > worstw) worstw = (double)r2 / q;
            if ((double)r3 / q > worst128) worst128 = (double)r3 / q;
            if ((i & 15) == 2) { // the deferred accumulator: 17 products of 64-bit x by canonical keys is the most k_k3 adds (16 digits + the start value)
                u128 acc = 0, ref = 0;
                for (int j = 0; j < 17; ++j) {
                    const u64 xx = j == 0 ? ~0ull : rnd(), kk = j < 2 ? q - 1 : rnd() % q;
                    if (j < 16) { mac128(acc, xx, kk); ref = (ref + (u128)(xx % q) * kk) % q; }
                }
                const u64 ra = fold_acc128(acc, c, c2);
                if (ra % q != (u64)ref) { if (bad < 5) printf("ACC MISMATCH q=%llx\n", (unsigned long long)q); ++bad; }
                if ((double)ra / q > worstacc) worstacc = (double)ra / q;
            }
        }
    }
    printf("host check: %ld products per form over 3 primes, %ld mismatches; largest result / q: fold2c %.4f, fold2c_wide %.4f, fold128 %.4f, acc128 fold %.4f\n",
           n / 3 * 3, bad, worst2c, worstw, worst128, worstacc);
    return bad != 0;
}

int main(int argc, char **argv)
{
    const long ncheck = argc > 1 ? atol(argv[1]) : 100000000L;
    if (host_check(ncheck)) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { printf("no HIP device: host check only\n"); return 0; }
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    u64 *x; TwC *w;
    (void)hipMalloc(&x, 256 * 16 * 8);
    (void)hipMalloc(&w, 1024 * 16);
    (void)hipMemset(x, 0x5a, 256 * 16 * 8);
    (void)hipMemset(w, 0x07, 1024 * 16);
    const int n = 100000, blocks = p.multiProcessorCount * 2;
    const u64 q = 0xFFFFFFFFFFC0001ull;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[4] = {"Shoup wide-lazy butterfly (product)", "fold2c butterfly (6 mul, r < 1.13 q)", "fold2c_wide butterfly (5 mul, r < 2.13 q)", "fold128 butterfly (7 mul, no table)"};
    for (int which = 0; which < 4; ++which)
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(kb<0>, dim3(blocks), dim3(256), 0, 0, x, w, q, n);
            if (which == 1) hipLaunchKernelGGL(kb<1>, dim3(blocks), dim3(256), 0, 0, x, w, q, n);
            if (which == 2) hipLaunchKernelGGL(kb<2>, dim3(blocks), dim3(256), 0, 0, x, w, q, n);
            if (which == 3) hipLaunchKernelGGL(kb<3>, dim3(blocks), dim3(256), 0, 0, x, w, q, n);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-45s: %.2f ns per butterfly per wave per SIMD-slot (%.1f ms)\n", names[which], ms * 1e6 / ((double)n * 8 * 2), ms);
        }
    const int nm = 20000;
    const char *mnames[2] = {"Shoup lazy MAC, runs of 6 (product)", "128-bit deferred MAC, one fold per 16"};
    for (int which = 0; which < 2; ++which)
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(km<0>, dim3(blocks), dim3(256), 0, 0, x, w, q, nm);
            if (which == 1) hipLaunchKernelGGL(km<1>, dim3(blocks), dim3(256), 0, 0, x, w, q, nm);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-45s: %.2f ns per MAC per wave per SIMD-slot (%.1f ms)\n", mnames[which], ms * 1e6 / ((double)nm * 8 * 16 * 2), ms);
        }
    return 0;
}
