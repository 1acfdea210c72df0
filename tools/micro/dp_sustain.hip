// dp_sustain.hip — sustained issue rate of fp64 FMA chains (test tool): does the chip hold its clock at full fp64 VALU utilisation?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void __launch_bounds__(256) k_chain(double *out, double a, double b, int iters)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    if (s == 1.2345) out[threadIdx.x] = s;
}
template <int CHAINS> static void run(int cus, int wps, double *out, int reps)
{
    const int iters = 200000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain<CHAINS>, dim3(cus * wps), dim3(256), 0, 0, out, 1.0000001, 0.5, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("chains=%d waves/SIMD=%d rep %d: %.3f ns per wave-instruction per SIMD (%.1f ms)\n", CHAINS, wps, r, ms * 1e6 / ((double)iters * 16 * CHAINS * wps), ms);
    }
}
int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    double *out;
    (void)hipMalloc(&out, 1 << 20);
    run<4>(p.multiProcessorCount, 2, out, 12);  // ~100 % utilisation
    run<1>(p.multiProcessorCount, 2, out, 6);   // ~70 %
    (void)hipFree(out);
    return 0;
}
