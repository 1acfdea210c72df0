// int_rates.hip — issue cost of the instructions the 64-bit integer engine is made of (test tool).  Eight independent chains per
// wave, two waves per SIMD, every CU: ns per wave-instruction per SIMD, to be read against v_fma_f64's 2.2 ns (dp_sustain.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef uint32_t u32;
#define REP16(X) X X X X X X X X X X X X X X X X
enum Op { MAD64, MULLO, MULHI, ADD3, LSHLADD64, ADDCO, MOV, FMA64, MAD24, MUL24, CNDMASK, CMP64, SUBPAIR, MAD64DEP, MULLODEP, FMA64DEP };
template <int OP> __global__ void __launch_bounds__(256) k(u64 *out, u32 a, u32 b, int iters)
{
    u64 x[8];
    u32 y[8];
    double d[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { x[c] = threadIdx.x * 77 + c; y[c] = threadIdx.x + 3 * c; d[c] = threadIdx.x + c; }
    const u32 va = a + threadIdx.x, vb = b ^ threadIdx.x;
    const double da = 1.0000001, db = 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (OP == MAD64) asm volatile(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(va), "v"(vb) : "vcc");
            if (OP == MAD64DEP) asm volatile(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %8, %9, %0\n"
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %8, %9, %0\n"
                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(va), "v"(vb) : "vcc");
#define OP32(NAME, INS) \
            if (OP == NAME) asm volatile( \
                INS " %0, %8, %0\n " INS " %1, %8, %1\n " INS " %2, %8, %2\n " INS " %3, %8, %3\n " INS " %4, %8, %4\n " INS " %5, %8, %5\n " INS " %6, %8, %6\n " INS " %7, %8, %7\n" \
                : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(va), "v"(vb) : "vcc");
            OP32(MULLO, "v_mul_lo_u32")
            OP32(MULHI, "v_mul_hi_u32")
            OP32(MUL24, "v_mul_u32_u24")
            if (OP == MULLODEP) asm volatile(
                "v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %0, %8, %0\n"
                : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(va), "v"(vb) : "vcc");
#define OP32C(NAME, INS) \
            if (OP == NAME) asm volatile( \
                INS " %0, %8, %9, %0\n " INS " %1, %8, %9, %1\n " INS " %2, %8, %9, %2\n " INS " %3, %8, %9, %3\n " INS " %4, %8, %9, %4\n " INS " %5, %8, %9, %5\n " INS " %6, %8, %9, %6\n " INS " %7, %8, %9, %7\n" \
                : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(va), "v"(vb) : "vcc");
            OP32C(ADD3, "v_add3_u32")
            OP32C(MAD24, "v_mad_u32_u24")
            if (OP == MOV) asm volatile(
                "v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %8\n v_mov_b32 %3, %9\n v_mov_b32 %4, %8\n v_mov_b32 %5, %9\n v_mov_b32 %6, %8\n v_mov_b32 %7, %9\n"
                : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(va), "v"(vb) : "vcc");
            if (OP == CNDMASK) asm volatile(
                "v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc\n"
                : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(va), "v"(vb) : "vcc");
#define OP64(NAME, INS, TAIL) \
            if (OP == NAME) asm volatile( \
                INS " %0, %0, " TAIL "\n " INS " %1, %1, " TAIL "\n " INS " %2, %2, " TAIL "\n " INS " %3, %3, " TAIL "\n " INS " %4, %4, " TAIL "\n " INS " %5, %5, " TAIL "\n " INS " %6, %6, " TAIL "\n " INS " %7, %7, " TAIL "\n" \
                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(x[0] ^ va) : "vcc");
            OP64(LSHLADD64, "v_lshl_add_u64", "0, %8")
            if (OP == CMP64) asm volatile(
                "v_cmp_le_u64 vcc, %0, %1\n v_cmp_le_u64 vcc, %1, %2\n v_cmp_le_u64 vcc, %2, %3\n v_cmp_le_u64 vcc, %3, %4\n v_cmp_le_u64 vcc, %4, %5\n v_cmp_le_u64 vcc, %5, %6\n v_cmp_le_u64 vcc, %6, %7\n v_cmp_le_u64 vcc, %7, %0\n"
                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(va) : "vcc");
            if (OP == ADDCO) asm volatile(
                "v_add_co_u32 %0, vcc, %8, %0\n v_add_co_u32 %1, vcc, %8, %1\n v_add_co_u32 %2, vcc, %8, %2\n v_add_co_u32 %3, vcc, %8, %3\n v_add_co_u32 %4, vcc, %8, %4\n v_add_co_u32 %5, vcc, %8, %5\n v_add_co_u32 %6, vcc, %8, %6\n v_add_co_u32 %7, vcc, %8, %7\n"
                : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(va), "v"(vb) : "vcc");
            if (OP == SUBPAIR) asm volatile( // 64-bit subtraction as the compiler writes it (with the wait states it inserts)
                "v_sub_co_u32 %0, vcc, %0, %8\n s_nop 1\n v_subb_co_u32 %1, vcc, %1, %9, vcc\n v_sub_co_u32 %2, vcc, %2, %8\n s_nop 1\n v_subb_co_u32 %3, vcc, %3, %9, vcc\n"
                "v_sub_co_u32 %4, vcc, %4, %8\n s_nop 1\n v_subb_co_u32 %5, vcc, %5, %9, vcc\n v_sub_co_u32 %6, vcc, %6, %8\n s_nop 1\n v_subb_co_u32 %7, vcc, %7, %9, vcc\n"
                : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(va), "v"(vb) : "vcc");
            if (OP == 100) asm volatile(
                "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n"
                : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(va), "v"(vb) : "vcc");
            if (OP == 101) asm volatile(
                "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(va), "v"(vb) : "vcc");
            if (OP == 102) asm volatile( // mads with an SGPR operand and the constant 1, as the Shoup chain uses them
                "v_mad_u64_u32 %0, vcc, %8, 1, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, 1, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                "v_mad_u64_u32 %4, vcc, %8, 1, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, 1, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(va), "s"(b) : "vcc");
            if (OP == 103) asm volatile( // mads writing fresh destinations from a zero addend
                "v_mad_u64_u32 %0, vcc, %8, %9, 0\n v_mad_u64_u32 %1, vcc, %8, %9, 0\n v_mad_u64_u32 %2, vcc, %8, %9, 0\n v_mad_u64_u32 %3, vcc, %8, %9, 0\n"
                "v_mad_u64_u32 %4, vcc, %8, %9, 0\n v_mad_u64_u32 %5, vcc, %8, %9, 0\n v_mad_u64_u32 %6, vcc, %8, %9, 0\n v_mad_u64_u32 %7, vcc, %8, %9, 0\n"
                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(va), "v"(vb) : "vcc");
            if (OP == FMA64) {
#pragma unroll
                for (int c = 0; c < 8; ++c) d[c] = __builtin_fma(d[c], da, db);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (OP == FMA64DEP) {
#pragma unroll
                for (int c = 0; c < 8; ++c) d[0] = __builtin_fma(d[0], da, db);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    u64 s = 0;
    double ds = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) { s += x[c] + y[c]; ds += d[c]; }
    if (s == 0x123456789abcdefull || ds == 1.2345) out[threadIdx.x] = s;
}
template <int OP> static void run(const char *name, int cus, u64 *out)
{
    const int iters = 300000, wps = 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(cus * wps), dim3(256), 0, 0, out, 12345u, 0x9e3779b9u, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    printf("%-28s %.3f ns per wave-instruction per SIMD\n", name, best * 1e6 / ((double)iters * 4 * 8 * wps));
}
int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    u64 *out;
    (void)hipMalloc(&out, 1 << 20);
    const int cus = p.multiProcessorCount;
    run<FMA64>("v_fma_f64", cus, out);
    run<FMA64DEP>("v_fma_f64 (one chain)", cus, out);
    run<MAD64>("v_mad_u64_u32", cus, out);
    run<MAD64DEP>("v_mad_u64_u32 (one chain)", cus, out);
    run<MULLO>("v_mul_lo_u32", cus, out);
    run<MULLODEP>("v_mul_lo_u32 (one chain)", cus, out);
    run<MULHI>("v_mul_hi_u32", cus, out);
    run<MUL24>("v_mul_u32_u24", cus, out);
    run<MAD24>("v_mad_u32_u24", cus, out);
    run<ADD3>("v_add3_u32", cus, out);
    run<LSHLADD64>("v_lshl_add_u64", cus, out);
    run<ADDCO>("v_add_co_u32", cus, out);
    run<SUBPAIR>("v_sub_co + v_subb (per instr)", cus, out);
    run<CMP64>("v_cmp_le_u64", cus, out);
    run<CNDMASK>("v_cndmask_b32", cus, out);
    run<MOV>("v_mov_b32", cus, out);
    run<100>("v_add_u32", cus, out);
    run<101>("v_xor_b32", cus, out);
    run<102>("v_mad_u64_u32 (sgpr / const 1)", cus, out);
    run<103>("v_mad_u64_u32 (addend 0)", cus, out);
    (void)hipFree(out);
    return 0;
}
