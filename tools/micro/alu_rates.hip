// Micro-benchmark: which MI355X (gfx950) instructions should carry 64-bit modular arithmetic?
// Measures sustained chip-wide rates of the candidate building blocks and of three complete
// modular-multiply butterflies (integer Shoup, integer Montgomery, fp64-FMA) held in registers.
// Build:  hipcc -O3 --offload-arch=gfx950 -o alu_rates alu_rates.hip
// Output: one line per variant: name, Gops/s (lane-ops per second, chip-wide), ns per op per lane.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned __int128 u128;

constexpr int ILP = 8;      // independent chains per lane
constexpr int ITERS = 4096; // loop trips

// ---------------- primitive chains ----------------
template <int OP>
__global__ void __launch_bounds__(256) k_prim(u64 *out, u64 seed, double dseed)
{
    u64 x[ILP];
    double d[ILP];
    u32 w[ILP];
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
        x[i] = seed * (tid + 1) + i * 0x9E3779B97F4A7C15ull;
        d[i] = dseed * (double)(tid + i + 1);
        w[i] = (u32)(x[i] >> 7) | 1u;
    }
    const u32 c0 = (u32)seed | 1u;
    const double dq = dseed * 3.0 + 1.0, dqi = 1.0 / dq;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) { // v_mad_u64_u32
                x[i] = (u64)(u32)x[i] * (u64)c0 + x[i];
            } else if (OP == 1) { // v_mul_lo_u32
                w[i] = w[i] * c0 + 0; asm volatile("" : "+v"(w[i]));
            } else if (OP == 2) { // v_mul_hi_u32
                w[i] = __umulhi(w[i], c0) | 0x80000001u;
            } else if (OP == 3) { // v_fma_f64
                d[i] = __builtin_fma(d[i], dqi, dq);
            } else if (OP == 4) { // v_mul_f64
                d[i] = d[i] * dqi; asm volatile("" : "+v"(d[i]));
            } else if (OP == 5) { // v_add_f64
                d[i] = d[i] + dq; asm volatile("" : "+v"(d[i]));
            } else if (OP == 6) { // v_rndne_f64
                d[i] = __builtin_rint(d[i]) ; asm volatile("" : "+v"(d[i]));
            } else if (OP == 7) { // 64-bit add (v_add_co_u32 + v_addc_co_u32)
                x[i] = x[i] + seed; asm volatile("" : "+v"(x[i]));
            } else if (OP == 8) { // v_mul_u32_u24
                w[i] = __umul24(w[i], c0); asm volatile("" : "+v"(w[i]));
            } else if (OP == 9) { // full 64x64->128 hi (mulhi64)
                x[i] = __umul64hi(x[i], seed) | 0x8000000000000001ull;
            } else if (OP == 10) { // 64x64 -> lo64
                x[i] = x[i] * seed + 1;
            } else if (OP == 11) { // v_fma_f32 reference
                float f = __builtin_bit_cast(float, w[i]);
                f = __builtin_fmaf(f, 1.0000001f, 0.5f);
                w[i] = __builtin_bit_cast(u32, f);
            } else if (OP == 12) { // v_cvt f64<->u32 pair
                d[i] = (double)(u32)(long long)d[i] + 0.5; // cvt_u32_f64 + cvt_f64_u32 + add
            } else if (OP == 13) { // v_floor_f64
                d[i] = __builtin_floor(d[i]); asm volatile("" : "+v"(d[i]));
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc += x[i] + (u64)w[i] + (u64)__builtin_bit_cast(u64, d[i]);
    if (acc == 0x1234567) out[tid] = acc;
}

// ---------------- complete modular butterflies ----------------
// (a) integer Harvey/Shoup butterfly, lazy [0,4q)
__device__ __forceinline__ void bfly_shoup(u64 &X, u64 &Y, u64 W, u64 Wp, u64 q, u64 two_q)
{
    u64 x = X - (X >= two_q ? two_q : 0);
    u64 Q = __umul64hi(Wp, Y);
    u64 T = W * Y - Q * q;
    X = x + T;
    Y = x - T + two_q;
}
// (a2) the same with the correction of X hoisted out of the butterfly: values grow by 2q per stage (product in [0,2q)), one
//      conditional subtraction of 8q per element every 7 stages keeps them below 16q < 2^64 (q < 2^60)
__device__ __forceinline__ void bfly_shoup_wide(u64 &X, u64 &Y, u64 W, u64 Wp, u64 q, u64 two_q)
{
    u64 Q = __umul64hi(Wp, Y);
    u64 T = W * Y - Q * q;
    u64 x = X;
    X = x + T;
    Y = x - T + two_q;
}
// (a3) approximate quotient: the low x low partial product's carry is dropped (quotient up to 2 too small, product in [0,4q))
__device__ __forceinline__ void bfly_shoup_approx(u64 &X, u64 &Y, u64 W, u64 Wp, u64 q, u64 four_q)
{
    const u32 y0 = (u32)Y, y1 = (u32)(Y >> 32), w0 = (u32)Wp, w1 = (u32)(Wp >> 32);
    u64 Q = (u64)y1 * w1 + __umulhi(y1, w0) + __umulhi(y0, w1);
    u64 T = W * Y - Q * q;
    u64 x = X;
    X = x + T;
    Y = x - T + four_q;
}
// (b) integer Montgomery butterfly (twiddle in Montgomery form), lazy [0,2q) on T
__device__ __forceinline__ void bfly_mont(u64 &X, u64 &Y, u64 Wm, u64 qinv, u64 q, u64 two_q)
{
    u64 x = X - (X >= two_q ? two_q : 0);
    u128 P = (u128)Y * Wm;
    u64 m = (u64)P * qinv;
    u64 t = (u64)(P >> 64) - __umul64hi(m, q) + q; // in (0, 2q)
    X = x + t;
    Y = x - t + two_q;
}
// (c) fp64 butterfly for q < 2^50 : exact integer arithmetic carried in doubles, centred residues
__device__ __forceinline__ void bfly_f64(double &X, double &Y, double w, double winv, double q)
{
    double h = Y * w;
    double l = __builtin_fma(Y, w, -h);
    double c = __builtin_rint(Y * winv);
    double dd = __builtin_fma(-c, q, h);
    double T = dd + l;
    double x = X;
    X = x + T;
    Y = x - T;
}

template <int V>
__global__ void __launch_bounds__(256) k_bfly(u64 *out, u64 seed, u64 q, u64 aux)
{
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    const u64 two_q = q << 1;
    if (V == 3 || V == 4) { // wide lazy ranges: periodic correction instead of one per butterfly
        u64 x[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = (seed * (tid + 1) + i * 0x9E3779B97F4A7C15ull) % q;
        u64 W = (seed ^ 0x5555) % q, Wp = aux;
        const u64 eight_q = q << 3, four_q = q << 2;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < ILP; i += 2) {
                if (V == 3) bfly_shoup_wide(x[i], x[i + 1], W, Wp, q, two_q);
                else bfly_shoup_approx(x[i], x[i + 1], W, Wp, q, four_q);
            }
            u64 t = x[1];
#pragma unroll
            for (int i = 1; i + 2 < ILP; i += 2) x[i] = x[i + 2];
            x[ILP - 1] = t;
            const int period = V == 3 ? 7 : 3;
            if (it % period == period - 1) {
#pragma unroll
                for (int i = 0; i < ILP; ++i) {
                    x[i] = x[i] >= eight_q ? x[i] - eight_q : x[i];
                    if (V == 4) x[i] = x[i] >= four_q ? x[i] - four_q : x[i];
                }
            }
        }
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc += x[i];
        if (acc == 0x1234567) out[tid] = acc;
    } else if (V < 2) {
        u64 x[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = (seed * (tid + 1) + i * 0x9E3779B97F4A7C15ull) % q;
        u64 W = (seed ^ 0x5555) % q, Wp = aux;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < ILP; i += 2) {
                if (V == 0) bfly_shoup(x[i], x[i + 1], W, Wp, q, two_q);
                else bfly_mont(x[i], x[i + 1], W, aux, q, two_q);
            }
            // rotate pairing so chains mix like NTT stages
            u64 t = x[1];
#pragma unroll
            for (int i = 1; i + 2 < ILP; i += 2) x[i] = x[i + 2];
            x[ILP - 1] = t;
        }
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc += x[i];
        if (acc == 0x1234567) out[tid] = acc;
    } else {
        double x[ILP];
        const double dq = (double)q, w = (double)((seed ^ 0x5555) % q), winv = w / dq;
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = (double)((seed * (tid + 1) + i * 0x9E3779B97F4A7C15ull) % q);
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < ILP; i += 2) bfly_f64(x[i], x[i + 1], w, winv, dq);
            double t = x[1];
#pragma unroll
            for (int i = 1; i + 2 < ILP; i += 2) x[i] = x[i + 2];
            x[ILP - 1] = t;
            if ((it & 7) == 7) { // periodic recentring of the never-multiplied lanes
#pragma unroll
                for (int i = 0; i < ILP; i += 2) x[i] = __builtin_fma(-__builtin_rint(x[i] * (1.0 / dq)), dq, x[i]);
            }
        }
        double acc = 0;
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc += x[i];
        if (acc == 1234567.25) out[tid] = (u64)acc;
    }
}

template <typename F>
static double time_ms(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / 5.0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz lds=%zu\n", p.name, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
    const int blocks = p.multiProcessorCount * 8, threads = 256;
    u64 *out; CK(hipMalloc(&out, (size_t)blocks * threads * 8));
    const u64 seed = 0x9E3779B97F4A7C15ull;
    const double lanes = (double)blocks * threads;
    const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_fma_f64", "v_mul_f64", "v_add_f64",
                           "v_rndne_f64", "add_u64(2 instr)", "v_mul_u32_u24", "mulhi64", "mullo64", "v_fma_f32",
                           "cvt_f64<->u32+add", "v_floor_f64"};
#define RUNP(OP) { double ms = time_ms([&] { hipLaunchKernelGGL(k_prim<OP>, dim3(blocks), dim3(threads), 0, 0, out, seed, 1.000001); }); \
    double ops = lanes * ILP * (double)ITERS; printf("%-20s %9.1f Gop/s  (%.3f ms)\n", names[OP], ops / ms / 1e6, ms); }
    RUNP(0) RUNP(1) RUNP(2) RUNP(3) RUNP(4) RUNP(5) RUNP(6) RUNP(7) RUNP(8) RUNP(9) RUNP(10) RUNP(11) RUNP(12) RUNP(13)
    const u64 q60 = 0xffffffffffc0001ull, q45 = 0x1fffffcf0001ull;
    // qinv for Montgomery: q * qinv == 1 mod 2^64
    auto inv64 = [](u64 q) { u64 x = q; for (int i = 0; i < 6; ++i) x *= 2 - q * x; return x; };
    const u64 W60 = (seed ^ 0x5555) % q60;
    const u64 Wp60 = (u64)(((u128)W60 << 64) / q60);
#define RUNB(V, Q, AUX, NAME) { double ms = time_ms([&] { hipLaunchKernelGGL(k_bfly<V>, dim3(blocks), dim3(threads), 0, 0, out, seed, Q, AUX); }); \
    double ops = lanes * (ILP / 2) * (double)ITERS; printf("%-20s %9.1f Gbfly/s (%.3f ms)\n", NAME, ops / ms / 1e6, ms); }
    RUNB(0, q60, Wp60, "bfly_shoup_u64")
    RUNB(1, q60, inv64(q60), "bfly_mont_u64")
    RUNB(2, q45, 0, "bfly_fp64_q45")
    RUNB(3, q60, Wp60, "bfly_shoup_u64_wide7")   // correction every 7 stages instead of per butterfly
    RUNB(4, q60, Wp60, "bfly_shoup_u64_approx3") // approximate quotient, two-step correction every 3 stages
    // occupancy sweep: waves per SIMD = blocks per CU (256-thread blocks -> 1 wave per SIMD per block)
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        const int blk = p.multiProcessorCount * bpc;
        double ms2 = time_ms([&] { hipLaunchKernelGGL(k_bfly<2>, dim3(blk), dim3(threads), 0, 0, out, seed, q45, 0); });
        double ms0 = time_ms([&] { hipLaunchKernelGGL(k_bfly<0>, dim3(blk), dim3(threads), 0, 0, out, seed, q60, Wp60); });
        double ops = (double)blk * threads * (ILP / 2) * (double)ITERS;
        printf("waves/SIMD=%d  fp64 %8.1f Gbfly/s   u64 %8.1f Gbfly/s\n", bpc, ops / ms2 / 1e6, ops / ms0 / 1e6);
    }
    hipFree(out);
    return 0;
}
