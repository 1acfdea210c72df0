// u64_bfly.hip — cost of the u64 engine's lazy butterfly in isolation (test tool): 8 butterflies per iteration on registers, twiddles in
// registers, two waves per SIMD on every CU.  Build twice: -DHE355_MAD_ASM=1 (multiply-add chains written out) and =0 (plain C).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../reference-seal-backend_amd/csrc/modarith.h"
using namespace he355;
__global__ void __launch_bounds__(256, 2) kb(u64 *x, const Tw16 *w, u64 q, int n)
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
        for (int i = 0; i < 8; ++i) { X[i] = ar.reduce16_to_4q(X[i]); Y[i] = ar.reduce16_to_4q(Y[i]); }
    }
    for (int i = 0; i < 8; ++i) { x[threadIdx.x + 256 * i] = X[i]; x[threadIdx.x + 256 * (i + 8)] = Y[i]; }
}
__global__ void __launch_bounds__(256, 2) kb_noreduce(u64 *x, const Tw16 *w, u64 q, int n)
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
    }
    for (int i = 0; i < 8; ++i) { x[threadIdx.x + 256 * i] = X[i]; x[threadIdx.x + 256 * (i + 8)] = Y[i]; }
}
int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    u64 *x; Tw16 *w;
    (void)hipMalloc(&x, 256 * 16 * 8);
    (void)hipMalloc(&w, 256 * 4 * 16);
    (void)hipMemset(x, 0x5a, 256 * 16 * 8);
    (void)hipMemset(w, 0x37, 256 * 4 * 16);
    const int n = 100000, blocks = p.multiProcessorCount * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which)
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(kb, dim3(blocks), dim3(256), 0, 0, x, w, 0xFFFFFFFFFFC0001ull, n);
            else hipLaunchKernelGGL(kb_noreduce, dim3(blocks), dim3(256), 0, 0, x, w, 0xFFFFFFFFFFC0001ull, n);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("MAD_ASM=%d %s: %.2f ns per butterfly per wave per SIMD-slot (%.1f ms)\n", HE355_MAD_ASM, which ? "butterflies only" : "with reduce16_to_4q", ms * 1e6 / ((double)n * 8 * 2), ms);
        }
    return 0;
}
