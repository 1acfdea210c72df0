// dp_latency.hip — issue rate of DEPENDENT fp64 VALU chains on one SIMD as a function of the waves resident on it (test tool).
// Build: hipcc --offload-arch=gfx950 -O3 -o dp_latency dp_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

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
            for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b); // CHAINS independent chains, interleaved
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    if (s == 1.2345) out[threadIdx.x] = s;
}

template <int CHAINS> static void run(int cus, double *out)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps) { // blocks of 256 threads = 1 wave per SIMD each; wps blocks per CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_chain<CHAINS>, dim3(cus * wps), dim3(256), 0, 0, out, 1.0000001, 0.5, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain<CHAINS>, dim3(cus * wps), dim3(256), 0, 0, out, 1.0000001, 0.5, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 16 * CHAINS * wps; // wave-instructions issued on one SIMD
        printf("chains=%d waves/SIMD=%d: %.2f ns per wave-instruction per SIMD (%.1f ms)\n", CHAINS, wps, ms * 1e6 / instr_per_simd, ms);
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    double *out;
    hipMalloc(&out, 1 << 20);
    printf("CUs=%d clock=%d kHz\n", p.multiProcessorCount, p.clockRate);
    run<1>(p.multiProcessorCount, out);
    run<2>(p.multiProcessorCount, out);
    run<4>(p.multiProcessorCount, out);
    hipFree(out);
    return 0;
}
