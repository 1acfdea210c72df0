// he355_kernels.hip — hand-written HIP kernels for gfx950 (CDNA4): RNS negacyclic NTT, dyadic arithmetic,
// hybrid key switching (relinearize / rotate) and RNS divide-and-round (mod-down, rescale).
//
// Work decomposition (one residue polynomial = N = N1 x 1024 coefficients):
//   * row kernels   : one 64-lane WAVE owns one 1024-element row, 16 elements per lane in VGPRs, the ten
//                     stages run 4+4+2 in registers with two register/lane transposes between them (ntt_core.h),
//                     each an exchange through the wave's own LDS region; a workgroup is four such waves (four
//                     rows of the same residue) and 34 KiB of LDS.
//   * column kernels: one LANE owns one stride-1024 column (N1 <= 32 values in VGPRs); twiddles of a column
//                     pass are wave-uniform, so they come through scalar loads.
// Global accesses are coalesced: layout A reads/writes 512 contiguous bytes per wave instruction, layout C
// moves 16 B per lane (two dwordx4 per 32-byte lane segment; a quad of lanes covers one 128-byte line).
// Reference semantics implemented here: SEAL Evaluator::{add, multiply (CKKS), relinearize_inplace,
// rescale_to_next_inplace, rotate_vector} as called from /root/reference/src/benchmarks/ckks/*.cpp and
// src/engine/seal_context.cpp (see include/he355.h for the call-site map).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <type_traits>

#include "he355_kernels.h"
#include "ntt_core.h"
#include "client/ckks_codec.h"
#include "client/multiword.h"
#include "client/sampler.h"

#if !defined(HE355_KNS) || !defined(HE355_U64_FOLD)
#error "he355_kernels.hip is compiled once per form of the u64 engine: -DHE355_KNS=ks_shoup -DHE355_U64_FOLD=0 and -DHE355_KNS=ks_fold -DHE355_U64_FOLD=1 (Makefile)"
#endif
namespace he355 {
namespace HE355_KNS {
namespace {

#include "kernel_common.inc"

// =======================================================================================================
// Generic transforms over a PolyView
// =======================================================================================================
template <class Ar>
__device__ __forceinline__ void rows_fwd_job(const PrimeDev &P, u64 *row, bool in_raw, u32 rowbase, int lane, u64 *lds, bool valid)
{
    const Ar ar = make_ar(P, (Ar *)nullptr);
    typename Ar::T x[kRowE];
    u64 v[kRowE];
    load_rowA(row, lane, v);
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = in_raw ? ar.from_raw(v[r]) : ar.from_canon(v[r]);
    wave_rows_fwd(ar, tw_table(gtw(P.fwd), rowbase), lane, lds, x);
#pragma unroll
    for (int r = 0; r < kRowE; ++r) v[r] = ar.to_canon(x[r]);
    if (valid) store_rowC(row, lane, v);
}

__global__ void __launch_bounds__(kBlock) k_rows_fwd(PolyView view, const PrimeDev *primes, u64 total_jobs, int logn1, int in_raw)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n1 = 1u << logn1;
    u64 job = (u64)blockIdx.x * kWaves + wave;
    bool valid = job < total_jobs;
    if (!valid) job = total_jobs - 1;
    const u32 a = (u32)(job & (n1 - 1));
    const u64 pj = job >> logn1;
    const int p = (int)(pj % view.polys_per_item);
    const u64 item = pj / view.polys_per_item;
    int prime = view.prime_of[p];
    if (prime == 255) { valid = false; prime = 0; }
    u64 *row = view.base + item * view.item_stride + ((u64)p << (logn1 + kRowLog)) + ((u64)a << kRowLog);
    const PrimeDev &P = primes[prime];
    if (P.f64) rows_fwd_job<ArF64>(P, row, in_raw != 0, n1 + a, lane, lds[wave], valid);
    else rows_fwd_job<ArU64>(P, row, in_raw != 0, n1 + a, lane, lds[wave], valid);
}

template <class Ar>
__device__ __forceinline__ void rows_inv_job(const PrimeDev &P, const u64 *src, u64 *dst, bool last, u32 rowbase, int lane, u64 *lds, bool valid)
{
    const Ar ar = make_ar(P, (Ar *)nullptr);
    typename Ar::T x[kRowE];
    u64 v[kRowE];
    load_rowC(src, lane, v);
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = ar.from_canon(v[r]);
    wave_rows_inv(ar, P, last, rowbase, lane, lds, x);
#pragma unroll
    for (int r = 0; r < kRowE; ++r) v[r] = last ? ar.to_canon(x[r]) : ar.to_raw(x[r]);
    if (valid) store_rowA(dst, lane, v);
}

__global__ void __launch_bounds__(kBlock) k_rows_inv(PolyView view, const PrimeDev *primes, u64 total_jobs, int logn1)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n1 = 1u << logn1;
    u64 job = (u64)blockIdx.x * kWaves + wave;
    bool valid = job < total_jobs;
    if (!valid) job = total_jobs - 1;
    const u32 a = (u32)(job & (n1 - 1));
    const u64 pj = job >> logn1;
    const int p = (int)(pj % view.polys_per_item);
    const u64 item = pj / view.polys_per_item;
    int prime = view.prime_of[p];
    if (prime == 255) { valid = false; prime = 0; }
    u64 *row = view.base + item * view.item_stride + ((u64)p << (logn1 + kRowLog)) + ((u64)a << kRowLog);
    const PrimeDev &P = primes[prime];
    if (P.f64) rows_inv_job<ArF64>(P, row, row, logn1 == 0, n1 + a, lane, lds[wave], valid);
    else rows_inv_job<ArU64>(P, row, row, logn1 == 0, n1 + a, lane, lds[wave], valid);
}

// inverse row pass of one selected residue per poly into a compact tail buffer
__global__ void __launch_bounds__(kBlock) k_rows_inv_select(const u64 *src, u64 src_poly_stride, u64 *tail, const PrimeDev *primes, int prime,
                                                            u64 total_jobs, int logn1)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n1 = 1u << logn1;
    u64 job = (u64)blockIdx.x * kWaves + wave;
    bool valid = job < total_jobs;
    if (!valid) job = total_jobs - 1;
    const u32 a = (u32)(job & (n1 - 1));
    const u64 poly = job >> logn1;
    const u64 *s = src + poly * src_poly_stride + ((u64)a << kRowLog);
    u64 *d = tail + (poly << (logn1 + kRowLog)) + ((u64)a << kRowLog);
    const PrimeDev &P = primes[prime];
    if (P.f64) rows_inv_job<ArF64>(P, s, d, logn1 == 0, n1 + a, lane, lds[wave], valid);
    else rows_inv_job<ArU64>(P, s, d, logn1 == 0, n1 + a, lane, lds[wave], valid);
}

template <int LOGN1>
__global__ void __launch_bounds__(kBlock) k_cols_fwd(PolyView view, const PrimeDev *primes)
{
    constexpr int N1 = 1 << LOGN1;
    const u64 pj = blockIdx.x >> 2;
    const int col = ((blockIdx.x & 3) << 8) | threadIdx.x;
    const int p = (int)(pj % view.polys_per_item);
    const u64 item = pj / view.polys_per_item;
    const int prime = view.prime_of[p];
    if (prime == 255) return;
    u64 *poly = view.base + item * view.item_stride + ((u64)p << (LOGN1 + kRowLog));
    const PrimeDev &P = primes[prime];
    if (P.f64) {
        const ArF64 ar = make_ar(P, (ArF64 *)nullptr);
        double x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = ar.from_canon(poly[(a << kRowLog) + col]);
        col_fwd<ArF64, LOGN1>(ar, x, ctw(P.fwd));
#pragma unroll
        for (int a = 0; a < N1; ++a) poly[(a << kRowLog) + col] = ar.to_raw(x[a]);
    } else {
        const ArU64 ar = make_ar(P, (ArU64 *)nullptr);
        u64 x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = poly[(a << kRowLog) + col];
        col_fwd<ArU64, LOGN1>(ar, x, ctw(P.fwd));
#pragma unroll
        for (int a = 0; a < N1; ++a) poly[(a << kRowLog) + col] = x[a];
    }
}

template <int LOGN1>
__global__ void __launch_bounds__(kBlock) k_cols_inv(PolyView view, const PrimeDev *primes)
{
    constexpr int N1 = 1 << LOGN1;
    const u64 pj = blockIdx.x >> 2;
    const int col = ((blockIdx.x & 3) << 8) | threadIdx.x;
    const int p = (int)(pj % view.polys_per_item);
    const u64 item = pj / view.polys_per_item;
    const int prime = view.prime_of[p];
    if (prime == 255) return;
    u64 *poly = view.base + item * view.item_stride + ((u64)p << (LOGN1 + kRowLog));
    const PrimeDev &P = primes[prime];
    if (P.f64) {
        const ArF64 ar = make_ar(P, (ArF64 *)nullptr);
        double x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = ar.from_raw(poly[(a << kRowLog) + col]);
        col_inv<ArF64, LOGN1>(ar, x, ctw(P.inv), P.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) poly[(a << kRowLog) + col] = ar.to_canon(x[a]);
    } else {
        const ArU64 ar = make_ar(P, (ArU64 *)nullptr);
        u64 x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = poly[(a << kRowLog) + col];
        col_inv<ArU64, LOGN1>(ar, x, ctw(P.inv), P.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) poly[(a << kRowLog) + col] = ar.to_canon(x[a]);
    }
}

// =======================================================================================================
// Element-wise kernels (HBM-bound): 16 bytes per lane per access
// =======================================================================================================
// Evaluator::add / sub: one thread = 2 coefficients of one residue polynomial of one result.
__global__ void __launch_bounds__(kBlock) k_addsub(const u64 *a, const u64 *b, u64 *out, Indexer ix, const PrimeDev *primes, int L, int polys,
                                                   int logN, u64 n_results, int sub)
{
    const u64 pairs_per_poly = (u64)1 << (logN - 1);
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 pp = gid >> (logN - 1);            // (result, poly)
    const u64 e2 = gid & (pairs_per_poly - 1);   // pair index inside the residue polynomial
    const u64 r = pp / polys;
    if (r >= n_results) return;
    const int p = (int)(pp % polys);
    const u64 q = primes[p % L].q;
    const u64 ct = (u64)polys << logN;
    const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(a + idx_a(ix, r) * ct + ((u64)p << logN))[e2];
    const ulonglong2 y = reinterpret_cast<const ulonglong2 *>(b + idx_b(ix, r) * ct + ((u64)p << logN))[e2];
    ulonglong2 z;
    if (sub) { z.x = submod(x.x, y.x, q); z.y = submod(x.y, y.y, q); }
    else { z.x = addmod(x.x, y.x, q); z.y = addmod(x.y, y.y, q); }
    reinterpret_cast<ulonglong2 *>(out + r * ct + ((u64)p << logN))[e2] = z;
}

// Evaluator::multiply_plain (mode 0: every polynomial times the NTT-form plaintext) and Evaluator::add_plain, CKKS
// (mode 1: plaintext added to c0, the other polynomials copied).  One thread = 2 coefficients of one residue polynomial.
__global__ void __launch_bounds__(kBlock) k_plain_op(const u64 *ct, const u64 *pt, u64 *out, Indexer ix, const PrimeDev *primes, int L, int size, int logN,
                                                     u64 n_results, int mode)
{
    const u64 pairs_per_poly = (u64)1 << (logN - 1);
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 pp = gid >> (logN - 1);
    const u64 e2 = gid & (pairs_per_poly - 1);
    const int polys = size * L;
    const u64 r = pp / polys;
    if (r >= n_results) return;
    const int p = (int)(pp % polys), i = p % L;
    const PrimeDev &P = primes[i];
    const u64 ctn = (u64)polys << logN;
    const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(ct + idx_a(ix, r) * ctn + ((u64)p << logN))[e2];
    ulonglong2 z = x;
    if (mode == 0 || p < L) {
        const ulonglong2 y = reinterpret_cast<const ulonglong2 *>(pt + (idx_b(ix, r) * L + i) * ((u64)1 << logN))[e2];
        if (mode == 1) {
            z.x = addmod(x.x, y.x, P.q); z.y = addmod(x.y, y.y, P.q);
        } else if (P.f64) {
            const ArF64 ar = make_ar(P, (ArF64 *)nullptr);
            z.x = ar.dy_out(ar.dy_mul(ar.dy_in(x.x), ar.dy_in(y.x))); z.y = ar.dy_out(ar.dy_mul(ar.dy_in(x.y), ar.dy_in(y.y)));
        } else {
            const ArU64 ar = make_ar(P, (ArU64 *)nullptr);
            z.x = ar.dy_mul(x.x, y.x); z.y = ar.dy_mul(x.y, y.y);
        }
    }
    reinterpret_cast<ulonglong2 *>(out + r * ctn + ((u64)p << logN))[e2] = z;
}

// CKKS mod_switch_to (ciphertexts, NTT-form plaintexts): keep the first L_to residues of every polynomial.
// in [n_polys][L][N] -> out [n_polys][L_to][N]; one thread = 2 coefficients.
__global__ void __launch_bounds__(kBlock) k_drop_residues(const u64 *in, u64 *out, int L, int L_to, int logN, u64 n_polys)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 pp = gid >> (logN - 1), e2 = gid & (((u64)1 << (logN - 1)) - 1);
    const u64 poly = pp / L_to;
    if (poly >= n_polys) return;
    const int i = (int)(pp % L_to);
    reinterpret_cast<ulonglong2 *>(out + ((poly * L_to + i) << logN))[e2] = reinterpret_cast<const ulonglong2 *>(in + ((poly * L + i) << logN))[e2];
}

// Sum of n ciphertexts (the add_inplace accumulation of collapseCKKS, seal_context.cpp:401): out = sum_r in[r].
// One thread = 2 coefficients of one residue polynomial of the result; it walks the n terms.
// blockIdx.y + c_first = result c of n_out: out[c] = (accumulate ? out[c] : 0) + sum_r in[r * n_out + c] (n_out = 1: the sum of a run of ciphertexts;
// n_out > 1: the sums over the inner index of a matrix product's terms, he355_bfv_multiply_relin_accumulate)
__global__ void __launch_bounds__(kBlock) k_sum_cts(const u64 *in, u64 *out, const PrimeDev *primes, int L, int polys, int logN, u64 n_terms, int accumulate,
                                                    u64 n_out, u64 c_first)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 p = gid >> (logN - 1), e2 = gid & (((u64)1 << (logN - 1)) - 1);
    if (p >= (u64)polys) return;
    const u64 c = c_first + blockIdx.y;
    const u64 q = primes[p % L].q;
    const u64 ctn = (u64)polys << logN;
    ulonglong2 *po = reinterpret_cast<ulonglong2 *>(out + c * ctn + (p << logN)) + e2;
    ulonglong2 s = accumulate ? *po : make_ulonglong2(0, 0);
    for (u64 r = 0; r < n_terms; ++r) {
        const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(in + (r * n_out + c) * ctn + (p << logN))[e2];
        s.x = addmod(s.x, x.x, q); s.y = addmod(s.y, x.y, q);
    }
    *po = s;
}

// Sum over the groups of a grouped launch: out[c] (+)= sum_g mult[g] * in[g * n_cts + c] (mult: how many rotation steps end at trie node g;
// he355_rotate_sum).  One thread = 2 coefficients of one residue polynomial of one result ciphertext; it walks the groups.
__global__ void __launch_bounds__(kBlock) k_sum_groups(const u64 *in, u64 *out, const u32 *mult, const PrimeDev *primes, int L, int polys, int logN, u64 n_cts,
                                                       u32 n_groups)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 cp = gid >> (logN - 1), e2 = gid & (((u64)1 << (logN - 1)) - 1);
    const u64 c = cp / polys;
    if (c >= n_cts) return;
    const int p = (int)(cp % polys);
    const u64 q = primes[p % L].q;
    const u64 ctn = (u64)polys << logN;
    ulonglong2 *po = reinterpret_cast<ulonglong2 *>(out + c * ctn + ((u64)p << logN)) + e2;
    ulonglong2 s = *po;
    // four groups' loads in flight per lane (the counts are wave-uniform: scalar loads, uniform branches); every word is read once: non-temporal
    typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
    const v2u *pin = reinterpret_cast<const v2u *>(in + c * ctn + ((u64)p << logN)) + e2;
    const u64 gstride = (n_cts * ctn) >> 1;
    for (u32 g = 0; g < n_groups; g += 4) {
        u32 m[4];
        v2u x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = g + k < n_groups ? mult[g + k] & ~kGroupKeepBit : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (m[k]) x[k] = __builtin_nontemporal_load(pin + (u64)(g + k) * gstride);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            for (u32 r = 0; r < m[k]; ++r) { s.x = addmod(s.x, x[k].x, q); s.y = addmod(s.y, x[k].y, q); }
    }
    *po = s;
}

// Gather / scatter of whole ciphertexts by an index list held in the kernel arguments (rotate_each groups the ciphertexts
// that need the same Galois element): gather: dst[g] = src[idx[g]]; scatter: dst[idx[g]] = src[g].  16 B per lane.
struct MoveList {
    uint32_t idx[kMoveListCap];
};
__global__ void __launch_bounds__(kBlock) k_move_cts(u64 *dst, const u64 *src, MoveList list, u64 pairs_per_ct, int scatter)
{
    const u64 g = blockIdx.y;
    const u64 e2 = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (e2 >= pairs_per_ct) return;
    const u64 i = list.idx[g];
    const u64 s = (scatter ? g : i) * pairs_per_ct + e2, d = (scatter ? i : g) * pairs_per_ct + e2;
    reinterpret_cast<ulonglong2 *>(dst)[d] = reinterpret_cast<const ulonglong2 *>(src)[s];
}

// Evaluator::multiply, CKKS, size 2 x 2 -> 3 (dyadic tensor).  One thread = 2 coefficients of one residue.
template <class Ar>
__device__ __forceinline__ void mul3_pair(const Ar &ar, const ulonglong2 a0, const ulonglong2 a1, const ulonglong2 b0, const ulonglong2 b1,
                                          ulonglong2 &c0, ulonglong2 &c1, ulonglong2 &c2)
{
    typename Ar::T x0 = ar.dy_in(a0.x), x1 = ar.dy_in(a1.x), y0 = ar.dy_in(b0.x), y1 = ar.dy_in(b1.x);
    c0.x = ar.dy_out(ar.dy_mul(x0, y0));
    c1.x = ar.dy_out(ar.dy_add(ar.dy_mul(x0, y1), ar.dy_mul(x1, y0)));
    c2.x = ar.dy_out(ar.dy_mul(x1, y1));
    x0 = ar.dy_in(a0.y); x1 = ar.dy_in(a1.y); y0 = ar.dy_in(b0.y); y1 = ar.dy_in(b1.y);
    c0.y = ar.dy_out(ar.dy_mul(x0, y0));
    c1.y = ar.dy_out(ar.dy_add(ar.dy_mul(x0, y1), ar.dy_mul(x1, y0)));
    c2.y = ar.dy_out(ar.dy_mul(x1, y1));
}

__global__ void __launch_bounds__(kBlock) k_mul3(const u64 *a, const u64 *b, u64 *out, Indexer ix, const PrimeDev *primes, int L, int logN,
                                                 u64 n_results)
{
    const u64 pairs_per_poly = (u64)1 << (logN - 1);
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 ri = gid >> (logN - 1);
    const u64 e2 = gid & (pairs_per_poly - 1);
    const u64 r = ri / L;
    if (r >= n_results) return;
    const int i = (int)(ri % L);
    const u64 poly = (u64)1 << logN, P1 = (u64)L << logN;
    const u64 *pa = a + idx_a(ix, r) * 2 * P1 + i * poly, *pb = b + idx_b(ix, r) * 2 * P1 + i * poly;
    const ulonglong2 a0 = reinterpret_cast<const ulonglong2 *>(pa)[e2], a1 = reinterpret_cast<const ulonglong2 *>(pa + P1)[e2];
    const ulonglong2 b0 = reinterpret_cast<const ulonglong2 *>(pb)[e2], b1 = reinterpret_cast<const ulonglong2 *>(pb + P1)[e2];
    ulonglong2 c0, c1, c2;
    const PrimeDev &P = primes[i];
    if (P.f64) mul3_pair(make_ar(P, (ArF64 *)nullptr), a0, a1, b0, b1, c0, c1, c2);
    else mul3_pair(make_ar(P, (ArU64 *)nullptr), a0, a1, b0, b1, c0, c1, c2);
    u64 *po = out + r * 3 * P1 + i * poly;
    reinterpret_cast<ulonglong2 *>(po)[e2] = c0;
    reinterpret_cast<ulonglong2 *>(po + P1)[e2] = c1;
    reinterpret_cast<ulonglong2 *>(po + 2 * P1)[e2] = c2;
}

// Sum over the inner dimension of size-3 dyadic tensors: out(i,j) = sum_k a(i,k) (x) b(k,j) — the multiply / add_inplace
// loop of the CipherBatchAxis workloads (ckks cipherbatchaxis .cpp:404-420) kept in registers: 4 reads per term, 3 writes
// per result.  One thread = 2 coefficients of one residue of one result.
template <class Ar>
__device__ __forceinline__ void mul3_acc_pair(const Ar &ar, u64 q, const u64 *pa, const u64 *pb, u64 a_step, u64 b_step, u64 P1, u64 e2, int inner,
                                              ulonglong2 &s0, ulonglong2 &s1, ulonglong2 &s2)
{
    s0 = s1 = s2 = make_ulonglong2(0, 0);
    for (int k = 0; k < inner; ++k, pa += a_step, pb += b_step) {
        const ulonglong2 a0 = reinterpret_cast<const ulonglong2 *>(pa)[e2], a1 = reinterpret_cast<const ulonglong2 *>(pa + P1)[e2];
        const ulonglong2 b0 = reinterpret_cast<const ulonglong2 *>(pb)[e2], b1 = reinterpret_cast<const ulonglong2 *>(pb + P1)[e2];
        ulonglong2 c0, c1, c2;
        mul3_pair(ar, a0, a1, b0, b1, c0, c1, c2);
        s0.x = addmod(s0.x, c0.x, q); s0.y = addmod(s0.y, c0.y, q);
        s1.x = addmod(s1.x, c1.x, q); s1.y = addmod(s1.y, c1.y, q);
        s2.x = addmod(s2.x, c2.x, q); s2.y = addmod(s2.y, c2.y, q);
    }
}

__global__ void __launch_bounds__(kBlock) k_mul3_acc(const u64 *a, const u64 *b, u64 *out, const PrimeDev *primes, int L, int logN, u64 rows, u64 cols,
                                                     int inner, u64 a_stride_i, u64 a_stride_k, u64 b_stride_k, u64 b_stride_j)
{
    const u64 pairs_per_poly = (u64)1 << (logN - 1);
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 ri = gid >> (logN - 1);
    const u64 e2 = gid & (pairs_per_poly - 1);
    const u64 r = ri / L;
    if (r >= rows * cols) return;
    const int i = (int)(ri % L);
    const u64 poly = (u64)1 << logN, P1 = (u64)L << logN, ct = 2 * P1;
    const u64 *pa = a + (r / cols) * a_stride_i * ct + i * poly, *pb = b + (r % cols) * b_stride_j * ct + i * poly;
    ulonglong2 s0, s1, s2;
    const PrimeDev &P = primes[i];
    if (P.f64) mul3_acc_pair(make_ar(P, (ArF64 *)nullptr), P.q, pa, pb, a_stride_k * ct, b_stride_k * ct, P1, e2, inner, s0, s1, s2);
    else mul3_acc_pair(make_ar(P, (ArU64 *)nullptr), P.q, pa, pb, a_stride_k * ct, b_stride_k * ct, P1, e2, inner, s0, s1, s2);
    u64 *po = out + r * 3 * P1 + i * poly;
    reinterpret_cast<ulonglong2 *>(po)[e2] = s0;
    reinterpret_cast<ulonglong2 *>(po + P1)[e2] = s1;
    reinterpret_cast<ulonglong2 *>(po + 2 * P1)[e2] = s2;
}

// Stage the 1023 forward twiddles of row `rowbase` of prime P in LDS (one copy per block): every wave, digit and op
// of a (prime, row) tile reads them from there (ds_read, lane-dependent index) instead of L2.  fp64 engine: w only.
// The caller synchronises the block before the first use.
template <class Ar, int BLOCK>
__device__ __forceinline__ typename std::conditional<std::is_same<Ar, ArF64>::value, TwRowF64, TwRow>::type
stage_row_twiddles(const PrimeDev &P, const Ar &ar, u32 rowbase, unsigned char *twl_raw, bool inverse = false)
{
    constexpr bool kF64 = std::is_same<Ar, ArF64>::value;
    typename std::conditional<kF64, TwRowF64, TwRow>::type twr;
    const gtw_t gf = gtw(inverse ? P.inv : P.fwd); // (the two tables share their indexing: entry (rowbase << s) + g of stage s)
    constexpr int kIter = (kRowTw + BLOCK - 1) / BLOCK;
    Tw16 tmp[kIter];
#pragma unroll
    for (int k = 0; k < kIter; ++k) { // all loads in flight together
        const u32 i = threadIdx.x + k * BLOCK;
        tmp[k] = tw_load(gf, tw_row_source(rowbase, i + 1 < (u32)kRowTw ? i : 0));
    }
    if constexpr (kF64) {
        double *twl = reinterpret_cast<double *>(twl_raw);
#pragma unroll
        for (int k = 0; k < kIter; ++k)
            if (threadIdx.x + k * BLOCK < (u32)kRowTw) twl[threadIdx.x + k * BLOCK] = ArF64::tw_w(tmp[k]);
        twr.t = twl;
        twr.qinv = ar.qinv;
    } else {
        Tw16 *twl = reinterpret_cast<Tw16 *>(twl_raw);
#pragma unroll
        for (int k = 0; k < kIter; ++k)
            if (threadIdx.x + k * BLOCK < (u32)kRowTw) twl[threadIdx.x + k * BLOCK] = tmp[k];
        twr.t = twl;
    }
    return twr;
}

// =======================================================================================================
// K1: (multiply | take ct3 | Galois-permute) + inverse row pass of the key-switch target
// =======================================================================================================
constexpr int K1_MUL_C2 = 3; // k_k1 instantiation of K1_MUL with no_c01 (its own register allocation: the full tensor keeps three waves per SIMD)
struct K1Args {
    const u64 *addend; // GALOIS mode, optional: [n][2][L][N] added to the rotated ciphertext (c0' + addend0, addend1); null: (c0', 0)
    const u64 *a, *b;
    Indexer ix;
    const uint32_t *perm;
    u64 *c01; u64 c01_item_stride;
    u64 *c2n, *c2r;
    u64 n_ops, op_offset;
    int L, logn1, mode;
    int no_c1;                 // K1_GALOIS: polynomial 1 of c01 (zeros / the addend's) is left to the fused k_k3 (K3Fuse::c1_mode)
    int no_c0n;                // K1_GALOIS, no addend: the permuted c0 is not written either (the fused k_k3 gathers it from the input: c1_mode 4)
    int n_i;                   // residues handled by this launch (one arithmetic engine per launch)
    unsigned char i_list[64];
    KsGroups groups;           // K1_GALOIS, grouped launch (group_size > 0): op's source ciphertext and permutation table come from its group
};

template <class Ar, int MODE, class ITW = int>
__device__ __forceinline__ void k1_job(const K1Args &A, const PrimeDev &P, u64 op, int i, u32 a_row, int lane, u64 *lds, bool valid, const ITW *staged = nullptr)
{
    typedef typename Ar::T T;
    const Ar ar = make_ar(P, (Ar *)nullptr);
    const u32 n1 = 1u << A.logn1;
    const u64 N = (u64)n1 << kRowLog, P1 = (u64)A.L * N;
    const u64 roff = (u64)i * N + ((u64)a_row << kRowLog);
    u64 *c0p = A.c01 + op * A.c01_item_stride + roff;
    u64 *c1p = c0p + P1;
    u64 *c2np = A.c2n + op * P1 + roff;
    u64 *c2rp = A.c2r + op * P1 + roff;
    T x[kRowE];
    u64 v0[kRowE], v1[kRowE];
    if (MODE == K1_MUL || MODE == K1_MUL_C2) {
        const u64 r = A.op_offset + op;
        const u64 *pa = A.a + idx_a(A.ix, r) * 2 * P1 + roff, *pb = A.b + idx_b(A.ix, r) * 2 * P1 + roff;
        u64 a0[kRowE], a1[kRowE], b0[kRowE], b1[kRowE];
        u64 v2[kRowE];
        if (MODE == K1_MUL_C2) { // only the key-switch target c2 = a1 b1, through the inverse row pass (c0, c1 are computed by the fused k_k3 from the operands): half the reads, a quarter of the writes
            load_rowC(pa + P1, lane, a1);
            load_rowC(pb + P1, lane, b1);
#pragma unroll
            for (int r2 = 0; r2 < kRowE; ++r2) {
                v2[r2] = ar.dy_out(ar.dy_mul(ar.dy_in(a1[r2]), ar.dy_in(b1[r2])));
                x[r2] = ar.from_canon(v2[r2]);
            }
            // (no c2n row either: the fused k_k3 of a ct x ct multiply forms a1 b1 itself, from the rows it reads for c0, c1)
        } else {
        load_rowC(pa, lane, a0); load_rowC(pa + P1, lane, a1);
        load_rowC(pb, lane, b0); load_rowC(pb + P1, lane, b1);
#pragma unroll
        for (int r2 = 0; r2 < kRowE; ++r2) {
            const T x0 = ar.dy_in(a0[r2]), x1 = ar.dy_in(a1[r2]), y0 = ar.dy_in(b0[r2]), y1 = ar.dy_in(b1[r2]);
            v0[r2] = ar.dy_out(ar.dy_mul(x0, y0));
            v1[r2] = ar.dy_out(ar.dy_add(ar.dy_mul(x0, y1), ar.dy_mul(x1, y0)));
            v2[r2] = ar.dy_out(ar.dy_mul(x1, y1));
            x[r2] = ar.from_canon(v2[r2]);
        }
        if (valid) { store_rowC(c0p, lane, v0); store_rowC(c1p, lane, v1); store_rowC(c2np, lane, v2); }
        }
    } else if (MODE == K1_CT3) {
        const u64 *pa = A.a + (A.op_offset + op) * 3 * P1 + roff;
        u64 v2[kRowE];
        load_rowC(pa + 2 * P1, lane, v2);
#pragma unroll
        for (int r2 = 0; r2 < kRowE; ++r2) x[r2] = ar.from_canon(v2[r2]);
        if (!A.no_c1) { // (no_c1: the fused k_k3 reads c0, c1 and the NTT-form c2 from the size-3 ciphertext itself, K3Fuse::c1_mode 3)
            load_rowC(pa, lane, v0); load_rowC(pa + P1, lane, v1);
            if (valid) { store_rowC(c0p, lane, v0); store_rowC(c1p, lane, v1); store_rowC(c2np, lane, v2); }
        }
    } else { // K1_GALOIS: out[idx] = in[perm[idx]] on both polys; c1 := 0; key-switch target = permuted c1
        u64 src_ct = A.op_offset + op;
        const uint32_t *perm = A.perm;
        if (A.groups.group_size) { // (wave-uniform) the op's group names its source block and its Galois element
            const u32 gop = (u32)(A.op_offset + op), grp = gop / A.groups.group_size;
            src_ct = (u64)A.groups.src_block[grp] * A.groups.group_size + gop % A.groups.group_size;
            perm = A.groups.perm[grp];
        }
        const u64 *p0 = A.a + src_ct * 2 * P1 + (u64)i * N;
        const uint32_t *pm = perm + ((u64)a_row << kRowLog);
        u64 v2[kRowE];
        // The permutation of an NTT-form polynomial maps a row ONTO one source row (the low logn1 + 1 bits of an element's exponent are its
        // row's alone, and so are those of g times it: Params::galois_perm_ntt checks it), so the wave reads that row the way it reads any
        // row -- 16 bytes per lane, every line once -- into its exchange buffer and permutes there: sixteen 8-byte reads per lane at
        // addresses of its own were sixty-four cache lines per instruction for the texture path, and what k_k1 spent its time on.
        {
            const u64 *srow = p0 + (u64)(__builtin_amdgcn_readfirstlane(pm[0]) & ~(u32)(kRowN - 1));
            u32 pl[kRowE];
#pragma unroll
            for (int r2 = 0; r2 < kRowE; ++r2) pl[r2] = lds_pad((int)(pm[elemC(lane, r2)] & (u32)(kRowN - 1)));
            u64 t[kRowE];
            load_rowC(srow + P1, lane, t);
            lds_store_C(lds, lane, t);
            HE_WAVE_SYNC();
#pragma unroll
            for (int r2 = 0; r2 < kRowE; ++r2) v2[r2] = lds[pl[r2]];
            if (!A.no_c0n) {
                load_rowC(srow, lane, t);
                HE_WAVE_SYNC();
                lds_store_C(lds, lane, t);
                HE_WAVE_SYNC();
#pragma unroll
                for (int r2 = 0; r2 < kRowE; ++r2) v0[r2] = lds[pl[r2]];
            }
            HE_WAVE_SYNC(); // (the inverse row pass writes the buffer next)
#pragma unroll
            for (int r2 = 0; r2 < kRowE; ++r2) {
                if (A.no_c0n) v0[r2] = 0;
                v1[r2] = 0;
                x[r2] = ar.from_canon(v2[r2]);
            }
        }
        if (A.addend && !A.no_c0n) { // out = addend + rotate(in): the ciphertext the key-switched result is added into starts from the addend
            const u64 *ad = A.addend + (A.op_offset + op) * 2 * P1 + roff;
            u64 a0[kRowE];
            load_rowC(ad, lane, a0);
            if (!A.no_c1) load_rowC(ad + P1, lane, v1);
#pragma unroll
            for (int r2 = 0; r2 < kRowE; ++r2) v0[r2] = addmod(v0[r2], a0[r2], P.q);
        }
        if (valid) {
            if (!A.no_c0n) store_rowC(c0p, lane, v0);
            if (!A.no_c1) store_rowC(c1p, lane, v1);
            store_rowC(c2np, lane, v2);
        }
    }
    const bool last = A.logn1 == 0;
    if constexpr (std::is_same<ITW, int>::value) wave_rows_inv(ar, P, last, n1 + a_row, lane, lds, x);
    else wave_rows_inv_tw(ar, *staged, P.inv_w0_scaled, last, lane, lds, x); // the row's inverse twiddles, staged in LDS by the block
#pragma unroll
    for (int r2 = 0; r2 < kRowE; ++r2) v0[r2] = last ? ar.to_canon(x[r2]) : ar.to_raw(x[r2]);
    if (valid) store_rowA(c2rp, lane, v0);
}

// Throughput shape (round 5): a block's four waves take the SAME (residue, row) of four consecutive ciphertexts, so the block stages that
// row's 1023 inverse twiddles in LDS once (8 KiB as doubles for the fp64 engine, 16 KiB for the u64 engine) and every phase of the
// four inverse row passes reads them from there -- before, each wave fetched its 46 entries per lane from the prime's 512 KiB table
// in L2, twice the bytes of the row it transformed, with the latency in front of every phase.  Blocks of one tile are consecutive
// (the tile's twiddle rows and, for rotations, its permutation rows stay in L2).  Grid: n_i * n1 * ceil(n_ops / 4).
template <int MODE, class Ar>
__global__ void __launch_bounds__(kBlock) k_k1(K1Args A, const PrimeDev *primes)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    constexpr bool kF64 = std::is_same<Ar, ArF64>::value;
    __shared__ __attribute__((aligned(16))) unsigned char twl_raw[kF64 ? kRowTw * 8 : kRowTw * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n1 = 1u << A.logn1;
    const u64 n_opg = (A.n_ops + kWaves - 1) / kWaves;
    const u64 tile = blockIdx.x / n_opg, opg = blockIdx.x % n_opg; // (grid = tiles * n_opg exactly)
    const u32 a_row = (u32)(tile & (n1 - 1));
    const int i = A.i_list[tile >> A.logn1];
    u64 op = opg * kWaves + wave;
    const bool valid = op < A.n_ops;
    if (!valid) op = A.n_ops - 1;
    const PrimeDev &P = primes[i];
    const Ar ar = make_ar(P, (Ar *)nullptr);
    const auto itw = stage_row_twiddles<Ar, kBlock>(P, ar, n1 + a_row, twl_raw, true);
    __syncthreads();
    k1_job<Ar, MODE>(A, P, op, i, a_row, lane, lds[wave], valid, &itw);
}
// Latency shape: the fp64-engine residues (blocks 0 .. n_f - 1) and the u64-engine residues of one key switch in ONE launch (see k_k3_dual)
template <int MODE, class Ar>
__device__ __forceinline__ void k1_block(const K1Args &A, const PrimeDev *primes, const unsigned bid_x, u64 (*lds)[kLdsRow])
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n1 = 1u << A.logn1;
    const u64 total = A.n_ops * A.n_i * n1;
    u64 job = (u64)bid_x * kWaves + wave;
    const bool valid = job < total;
    if (!valid) job = total - 1;
    const u32 a_row = (u32)(job & (n1 - 1));
    const u64 oi = job >> A.logn1;
    const int i = A.i_list[oi % A.n_i];
    const u64 op = oi / A.n_i;
    k1_job<Ar, MODE>(A, primes[i], op, i, a_row, lane, lds[wave], valid);
}
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_k1_dual(K1Args AF, K1Args AU, unsigned n_f, const PrimeDev *primes)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    if (blockIdx.x < n_f) k1_block<MODE, ArF64>(AF, primes, blockIdx.x, lds);
    else k1_block<MODE, ArU64>(AU, primes, blockIdx.x - n_f, lds);
}

// =======================================================================================================
// K2: per (op, digit j, column): finish the inverse transform, lift to every key prime, forward column pass
// =======================================================================================================
struct K2Args {
    const u64 *c2r;
    u64 src_op_stride;
    u64 *d;
    u64 n_ops;
    int L, K, ckks, src_is_coeff;
    u64 f64_mask; // bit t: key prime t belongs to the fp64 engine
    int n_dig;    // digits handled by this launch (one instantiation per digit width: each gets its own register allocation)
    unsigned char dig_list[64];
    int tsplit;   // the targets of a (digit, column block) are dealt to tsplit blocks (blockIdx.y): latency shape, small grids
};

// What surrounds the butterflies is as lean as the butterflies (round 3; the first form of this kernel spent ~1000 VALU instructions per
// fp64 target and lane for 640 instructions of butterflies, HISTORY.md):
//   * the digit column is held in the form the targets consume: canonical doubles when q_j < 2^52 (every target of the fp64
//     engine then reads it as is; the two u64-engine targets convert back), integers only for a 60-bit digit;
//   * column twiddles come from PrimeDev::colw (bare doubles): 62 scalar registers per target instead of 124;
//   * the first stage reads the digit column and writes the target column (no copy of 32 values per target), the last stage
//     produces the 48-bit patterns directly (BIAS);
//   * every store address is (scalar row base) + (one per-lane 32-bit offset computed once per kernel): the slab pointer of the
//     target is made wave-uniform explicitly, the row bases advance on the scalar unit.
// canonical v of a prime >= 2^52 as a lazy value of the fp64 engine: hi * 2^32 + lo == hi * pow32 + lo (mod q), |result| < q/2 + 2^32 + 1
__device__ __forceinline__ double lift_wide(const ArF64 &ar, u64 v, double pow32)
{
    return ar.mulmod_vv((double)(u32)(v >> 32), pow32) + (double)(u32)v;
}
typedef const __attribute__((address_space(4))) double *cdw_t;
__device__ __forceinline__ cdw_t cdw(const double *p) { return (cdw_t)(unsigned long long)p; }
// Buffer resource over one polynomial's row slots (wave-uniform base made explicit with v_readfirstlane): a store through it is
// buffer_store ... v_off, s[rsrc], s_off offen -- address = base + (scalar row offset) + (per-lane 32-bit offset computed once per
// kernel), no VALU instruction per store.
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
constexpr u32 kSlotBytes = kRowN * 8; // one row slot of a slab
__device__ __forceinline__ __amdgpu_buffer_rsrc_t poly_rsrc(const void *p, u32 bytes)
{
    const unsigned long long v = (unsigned long long)p;
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}
// 48-bit pattern rows of one polynomial: row a in slot a, low words at lane offset off4 = 4 * col, high half-words at
// off2h = kPackHiOff + 2 * col
template <int N1> __device__ __forceinline__ void store_pattern_rows(__amdgpu_buffer_rsrc_t dst, u32 off4, u32 off2h, const double x[N1])
{
#pragma unroll
    for (int a = 0; a < N1; ++a) {
        union { u64 u; double d; } c;
        c.d = x[a];
        __builtin_amdgcn_raw_buffer_store_b32((u32)c.u, dst, (int)off4, a * (int)kSlotBytes, 0);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(c.u >> 32), dst, (int)off2h, a * (int)kSlotBytes, 0);
    }
}
template <int N1> __device__ __forceinline__ void store_word_rows(__amdgpu_buffer_rsrc_t dst, u32 off8, const u64 x[N1])
{
#pragma unroll
    for (int a = 0; a < N1; ++a) {
        u32x2_t v;
        v.x = (u32)x[a]; v.y = (u32)(x[a] >> 32);
        __builtin_amdgcn_raw_buffer_store_b64(v, dst, (int)off8, a * (int)kSlotBytes, 0);
    }
}
// N1 >= 8: the pattern rows of a wave's 64 columns go through a 12 KiB LDS tile (8 KiB of low words, 4 KiB of high half-words,
// row-major) and leave as 16-byte stores: 8 + 4 buffer_store_dwordx4 of 1 KiB per target instead of 32 + 32 stores of 256 / 128 bytes.
// Only the issuing wave touches its tile (wavefront-scope hand-off).  lane_lo / lane_hi: the per-lane byte offsets of the 16-byte
// pieces inside a group of 4 (low plane) / 8 (high plane) rows, wave's column offset included.
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
template <int N1>
__device__ __forceinline__ void store_pattern_rows_wide(__amdgpu_buffer_rsrc_t dst, unsigned char *tile, int lane, u32 lane_lo, u32 lane_hi, const double x[N1])
{
    static_assert(N1 % 8 == 0, "written for whole groups of 8 rows");
    u32 *lo = reinterpret_cast<u32 *>(tile);
    unsigned short *hi = reinterpret_cast<unsigned short *>(tile + N1 * 256);
#pragma unroll
    for (int a = 0; a < N1; ++a) {
        union { u64 u; double d; } c;
        c.d = x[a];
        lo[a * 64 + lane] = (u32)c.u;
        hi[a * 64 + lane] = (unsigned short)(c.u >> 32);
    }
    HE_WAVE_SYNC();
    const u32x4_t *lo4 = reinterpret_cast<const u32x4_t *>(tile);
    const u32x4_t *hi4 = reinterpret_cast<const u32x4_t *>(tile + N1 * 256);
    // all reads first, into registers of their own (left to itself the scheduler chains read -> wait -> store through one register quad)
    u32x4_t vl[N1 / 4], vh[N1 / 8];
#pragma unroll
    for (int k = 0; k < N1 / 4; ++k) vl[k] = lo4[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < N1 / 8; ++k) vh[k] = hi4[k * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < N1 / 4; ++k) // 4 rows x 256 bytes per instruction
        __builtin_amdgcn_raw_buffer_store_b128(vl[k], dst, (int)lane_lo, k * 4 * (int)kSlotBytes, 0);
#pragma unroll
    for (int k = 0; k < N1 / 8; ++k) // 8 rows x 128 bytes per instruction
        __builtin_amdgcn_raw_buffer_store_b128(vh[k], dst, (int)lane_hi, k * 8 * (int)kSlotBytes, 0);
    HE_WAVE_SYNC(); // the tile is free again for the next target
}
// worst-case magnitude after the forward column pass of the fp64 engine from inputs |x| < m0 (ArF64::bfly_fwd: m -> m + q (1/2 + m 2^-51))
template <int LOGN1> __device__ __forceinline__ bool fits_48(double m0, double q)
{
    double m = m0;
#pragma unroll
    for (int st = 0; st < LOGN1; ++st) m += q * (0.5 + m * 4.440892098500626e-16); // 2^-51
    return m * 1.0000001 < 140737488355328.0;                                        // 2^47
}

// Per-target constants come through the constant address space (scalar loads): a vector load of a PrimeDev field would put an
// s_waitcnt vmcnt(0) -- i.e. the full HBM write latency of the previous target's stores -- in front of every target.
// butterflies of a column-pass stage issued together (interleaved dependent chains, ArF64::mulmod_vv_g)
#ifndef HE355_K2_G
#define HE355_K2_G 2
#endif
constexpr int kK2G = HE355_K2_G;
// every n_groups-th set bit of mask, starting with the g-th (latency shape: the targets of a column are dealt to n_groups blocks)
__device__ __forceinline__ u64 split_mask(u64 mask, int g, int n_groups)
{
    if (n_groups <= 1) return mask;
    u64 out = 0;
    int r = 0;
    for (u64 m = mask; m; m &= m - 1, ++r)
        if (r % n_groups == g) out |= m & (~m + 1);
    return out;
}
typedef const __attribute__((address_space(4))) PrimeDev *cprime_t;
typedef double d16_t __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) d16_t *cd16_t;
__device__ __forceinline__ d16_t load_colw16(cprime_t cp, int t, int first) { return *(cd16_t)(unsigned long long)&cp[t].colw[first]; }

// Targets of the fp64 engine on the fast path (host-built masks PrimeDev::k2_direct / k2_lift of the digit's prime: the 48-bit row
// format holds the column pass's output).  One iteration per target prime t in `mask` (bit = prime index), software-pipelined
// over the scalar loads: the twiddles of stages 0..3 (entries 0..15) and the modulus constants of the NEXT target are requested
// before the current target's last stage and its stores, the 16 entries of stage 4 at the top of the target, behind stages 0..3 --
// as the compiler places them (one s_load + s_waitcnt lgkmcnt(0) per stage) each target waited five scalar-load latencies.
// DIRECT: the column enters as it is; else it is lifted first (re-centred, or reduced with integers from a 60-bit digit: DF false).
template <int LOGN1, bool DF, bool DIRECT, class Store>
__device__ __forceinline__ void k2n_fast_targets(const K2Args &A, const PrimeDev *primes, int j, u64 op, u64 mask,
                                                 const typename std::conditional<DF, double, u64>::type (&c)[1 << LOGN1], Store store_rows)
{
    constexpr int N1 = 1 << LOGN1;
    constexpr u64 N = (u64)N1 << kRowLog;
    constexpr int LA = LOGN1 < 4 ? LOGN1 : 4; // stages served by entries 0..15
    static_assert(DF || !DIRECT, "an integer digit is always lifted");
    if (!mask) return;
    const cprime_t cp = (cprime_t)(unsigned long long)primes;
    u64 m = mask;
    int t = __builtin_ctzll(m);
    d16_t wa = load_colw16(cp, t, 0);
    double qd = cp[t].qd, qinv = cp[t].qinv;
    // Scalar loads return out of order, so every wait on one is lgkmcnt(0), a wait on all of them.  The waits are therefore placed by
    // hand where nothing young is in flight: here, and before the next target's prefetch below (the stage-4 entries, requested at the
    // top of the target, landed long before); the tile's LDS traffic then covers the prefetch.
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
    for (;;) {
        m &= m - 1;
        const int tn = m ? __builtin_ctzll(m) : t;
        d16_t wb = wa;
        if constexpr (LOGN1 == 5) wb = load_colw16(cp, t, 16);
        __builtin_amdgcn_sched_barrier(0);
        ArF64 ar;
        ar.q = qd; ar.qinv = qinv; ar.ninv = 0; ar.ninv_i = 0;
        double x[N1];
        if constexpr (!DIRECT) {
            if constexpr (DF) {
#pragma unroll
                for (int a = 0; a < N1; ++a) x[a] = ar.renorm(c[a]);
            } else { // a 60-bit digit: hi * (2^32 mod q_t) + lo, one exact fp64 product (lift_wide) instead of a 64-bit Barrett reduction
                const double pow32 = cp[t].pow32;
#pragma unroll
                for (int a = 0; a < N1; ++a) x[a] = lift_wide(ar, c[a], pow32);
            }
        }
#pragma unroll
        for (int s = 0; s < LA; ++s) {
            const int gap = N1 >> (s + 1);
            // the stage's N1/2 butterflies, kK2G at a time (interleaved instruction chains)
            constexpr int G = (N1 / 2) % kK2G == 0 ? kK2G : 1;
#pragma unroll
            for (int b0 = 0; b0 < N1 / 2; b0 += G) {
                double X[G], Y[G], W[G], tw[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int b = b0 + k, a = ((b / gap) * 2 * gap) + (b % gap); // butterfly b of the stage: rows a, a + gap
                    W[k] = wa[(1 << s) + (a / (2 * gap))];
                    if (DIRECT && s == 0) { X[k] = (double)c[a]; Y[k] = (double)c[a + gap]; }
                    else { X[k] = x[a]; Y[k] = x[a + gap]; }
                }
                ar.template mulmod_vv_g<G>(Y, W, tw);
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int b = b0 + k, a = ((b / gap) * 2 * gap) + (b % gap);
                    if (s == LOGN1 - 1) {
                        const double xb = X[k] + kPackBias;
                        x[a] = xb + tw[k]; x[a + gap] = xb - tw[k];
                    } else {
                        x[a] = X[k] + tw[k]; x[a + gap] = X[k] - tw[k];
                    }
                }
            }
        }
        // the next target's first twiddles and constants: in flight behind the last stage and the stores
        __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): this target's stage-4 entries
        const d16_t wa_n = load_colw16(cp, tn, 0);
        const double qd_n = cp[tn].qd, qinv_n = cp[tn].qinv;
        __builtin_amdgcn_sched_barrier(0);
        const int tt = t == A.K - 1 ? A.L : t;
        const __amdgpu_buffer_rsrc_t dst = poly_rsrc(A.d + ((op * (A.L + 1) + tt) * A.L + j) * N, (u32)N1 * kSlotBytes);
        if constexpr (LOGN1 == 5) {
            constexpr int G = kK2G;
#pragma unroll
            for (int a0 = 0; a0 < N1; a0 += 2 * G) {
                double Y[G], W[G], tw[G];
#pragma unroll
                for (int k = 0; k < G; ++k) { Y[k] = x[a0 + 2 * k + 1]; W[k] = wb[(a0 >> 1) + k]; }
                ar.template mulmod_vv_g<G>(Y, W, tw);
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const double xb = x[a0 + 2 * k] + kPackBias;
                    x[a0 + 2 * k] = xb + tw[k]; x[a0 + 2 * k + 1] = xb - tw[k];
                }
            }
        }
        if constexpr (LOGN1 == 0) x[0] = (DIRECT ? (double)c[0] : x[0]) + kPackBias;
        __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the prefetch had the last stage to land; the tile's LDS traffic then waits by count
        store_rows(dst, x);
        if (!m) break;
        t = tn; wa = wa_n; qd = qd_n; qinv = qinv_n;
    }
}

// Targets of the fp64 engine.  DF: the digit's canonical coefficients are held as doubles (q_j < 2^52), else as integers.
template <int LOGN1, bool DF>
__device__ __forceinline__ void k2n_targets_f64(const K2Args &A, const PrimeDev *primes, const PrimeDev &Pj, int j, u64 op,
                                                const typename std::conditional<DF, double, u64>::type (&c)[1 << LOGN1], int col, unsigned char *tile)
{
    constexpr int N1 = 1 << LOGN1;
    constexpr u64 N = (u64)N1 << kRowLog;
    constexpr bool kWide = N1 >= 8;
    const u32 off4 = (u32)col << 2, off2h = ((u32)col << 1) + kPackHiOff;
    const int lane = col & 63;
    const u32 wc = (u32)col & ~63u; // the wave's first column
    const u32 lane_lo = ((u32)lane >> 4) * kSlotBytes + (wc << 2) + (((u32)lane & 15u) << 4);
    const u32 lane_hi = ((u32)lane >> 3) * kSlotBytes + kPackHiOff + (wc << 1) + (((u32)lane & 7u) << 4);
    auto store_rows = [&](__amdgpu_buffer_rsrc_t dst, const double (&x)[N1]) {
        if constexpr (kWide) store_pattern_rows_wide<N1>(dst, tile, lane, lane_lo, lane_hi, x);
        else store_pattern_rows<N1>(dst, off4, off2h, x);
    };
    // this launch's targets by prime index: data primes [0, L) and the special prime, without the digit's own prime (CKKS: that
    // digit is the NTT-form input itself)
    const cprime_t cp = (cprime_t)(unsigned long long)primes;
    u64 level = (A.L >= 64 ? ~(u64)0 : (((u64)1 << A.L) - 1)) | ((u64)1 << (A.K - 1));
    if (A.ckks) level &= ~((u64)1 << j);
    level = split_mask(level, (int)blockIdx.y, A.tsplit);
    const u64 f64_targets = A.f64_mask & level;
    const u64 direct = cp[j].k2_direct & f64_targets, lift = cp[j].k2_lift & f64_targets;
    if constexpr (DF) k2n_fast_targets<LOGN1, DF, true>(A, primes, j, op, direct, c, store_rows);
    k2n_fast_targets<LOGN1, DF, false>(A, primes, j, op, DF ? lift : (lift | direct), c, store_rows);
    // the general path: any lift, results re-centred before they are packed (wider fp64-engine primes)
    for (u64 m = f64_targets & ~(direct | lift); m; m &= m - 1) {
        const int t = __builtin_ctzll(m), tt = t == A.K - 1 ? A.L : t;
        const PrimeDev &Pt = primes[t];
        const __amdgpu_buffer_rsrc_t dst = poly_rsrc(A.d + ((op * (A.L + 1) + tt) * A.L + j) * N, (u32)N1 * kSlotBytes);
        const ArF64 ar = make_ar(Pt, (ArF64 *)nullptr);
        const cdw_t cw = cdw(Pt.colw);
        double x[N1];
        if constexpr (DF) {
            const bool recentre = Pj.q > 2 * Pt.q;
#pragma unroll
            for (int a = 0; a < N1; ++a) x[a] = recentre ? ar.renorm(c[a]) : c[a];
        } else {
            const ModU64 mt = make_modu(Pt);
#pragma unroll
            for (int a = 0; a < N1; ++a) x[a] = u52_to_f64(barrett64(c[a], mt));
        }
        col_fwd_w<LOGN1, false>(ar, [&](int a) { return x[a]; }, x, cw, 0.0);
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = ar.renorm(x[a]) + kPackBias;
        store_rows(dst, x);
    }
}
// Targets of the u64 engine, from the digit's canonical coefficients as integers.
template <int LOGN1>
__device__ __forceinline__ void k2n_targets_u64(const K2Args &A, const PrimeDev *primes, const PrimeDev &Pj, int j, u64 op, const u64 (&c)[1 << LOGN1], int col)
{
    constexpr int N1 = 1 << LOGN1;
    constexpr u64 N = (u64)N1 << kRowLog;
    const u32 off8 = (u32)col << 3;
    u64 level = (A.L >= 64 ? ~(u64)0 : (((u64)1 << A.L) - 1)) | ((u64)1 << (A.K - 1));
    if (A.ckks) level &= ~((u64)1 << j);
    level = split_mask(level, (int)blockIdx.y, A.tsplit);
    for (u64 m = ~A.f64_mask & level; m; m &= m - 1) {
        const int t = __builtin_ctzll(m), tt = t == A.K - 1 ? A.L : t;
        const PrimeDev &Pt = primes[t];
        const __amdgpu_buffer_rsrc_t dst = poly_rsrc(A.d + ((op * (A.L + 1) + tt) * A.L + j) * N, (u32)N1 * kSlotBytes);
        const ArU64 ar = make_ar(Pt, (ArU64 *)nullptr);
        u64 x[N1];
        if (Pj.q > Pt.q) {
            const ModU64 mt = make_modu(Pt);
#pragma unroll
            for (int a = 0; a < N1; ++a) x[a] = barrett64(c[a], mt);
        } else {
#pragma unroll
            for (int a = 0; a < N1; ++a) x[a] = c[a];
        }
        col_fwd<ArU64, LOGN1>(ar, x, ctw(Pt.fwd));
        store_word_rows<N1>(dst, off8, x);
    }
}

// WIDE: the launch's digits are 60-bit (integers throughout); else below 2^52 (doubles).  Two instantiations instead of one kernel with
// both paths: each gets its own register allocation (one kernel at 168 registers spilled 102 through the wide path's live ranges and
// cost 14.9 ms; at 256 it took 12.4; the pair takes 11.0 + 1.0).  Two or three waves per SIMD make no difference to the narrow
// instantiation any more (11.04 / 11.18 ms): K2N_WAVES = 2 leaves it without spills.
#ifndef K2N_WAVES
#define K2N_WAVES 2
#endif
#ifndef K2N_WAVES_WIDE
#define K2N_WAVES_WIDE 2
#endif
template <int LOGN1, bool WIDE>
__global__ void __launch_bounds__(kBlock, WIDE ? K2N_WAVES_WIDE : K2N_WAVES) k_k2n(K2Args A, const PrimeDev *primes)
{
#define K2N_BID_X blockIdx.x
#include "k2n_body.inc"
#undef K2N_BID_X
}
// Latency shape: the digits below 2^52 (blocks 0 .. n_n - 1) and the 60-bit digits of one key switch in ONE launch (see k_k3_dual);
// blockIdx.y = target group for both (same tsplit).
template <int LOGN1, bool WIDE>
__device__ __forceinline__ void k2n_body_fn(const K2Args &A, const PrimeDev *primes, const unsigned bid_x)
{
#define K2N_BID_X bid_x
#include "k2n_body.inc"
#undef K2N_BID_X
}
template <int LOGN1>
__global__ void __launch_bounds__(kBlock) k_k2n_dual(K2Args AN, K2Args AW, unsigned n_n, const PrimeDev *primes)
{
    if (blockIdx.x < n_n) k2n_body_fn<LOGN1, false>(AN, primes, blockIdx.x);
    else k2n_body_fn<LOGN1, true>(AW, primes, blockIdx.x - n_n);
}

// =======================================================================================================
// K3: per (op, key prime tt, row): sum over digits j of NTT_tt(digit j) * key_j[k][tt]
// =======================================================================================================
struct K3Args {
    const u64 *d, *c2n, *key;
    const u64 *cols;   // FUSE: mod-down corrections after the forward column pass [n_ops*2][L][N] (raw of prime tt)
    u64 *c01; u64 c01_item_stride; // FUSE: polys that receive (T - NTT(cols)) * P^-1
    const FloorConst *fc;          // FUSE: floor constants [K][K]
    const u64 *cols2; u64 *out2;   // FUSE + rescale: second correction slab [n_ops*2][L-1][N] and the final output [n_ops][2][L-1][N]
    int c1_mode; const u64 *c1_src; // FUSE, rotations: the floor step's addend of polynomial 1 is zero (1) / a row of c1_src (2) instead of a c01 row (0)
    const u64 *gsrc; const uint32_t *gperm; u64 gsrc_op_offset; // c1_mode 4: the rotation's input, gathered here (K3Fuse::gsrc)
    const u64 *ta, *tb; Indexer tix; u64 t_op_offset; // FUSE, ct x ct multiply: the floor step's addend is a0 b0 / a0 b1 + a1 b0 of these operands (K3Fuse::ta); ta null: read from c01
    KsGroups groups; u64 g_op_offset; u64 keyq_offset; // GROUPED: op (g_op_offset + op) takes its key from groups.key[group]; its quotients follow at keyq_offset
    const u64 *keyq;   // Shoup quotients of the key residues under the u64-engine primes: [L_top][2][n_q][N]
    int n_q;           // u64-engine primes in the key chain
    unsigned char q_slot[64]; // tt_list[k] -> its index among them
    u64 *t, *tp, *tpr; // sums of the data primes; of the special prime (unused: its rows leave through tpr, inverse row pass done)
    u64 n_ops;
    int L, K, logn1, ckks;
    int n_tt;
    u32 og_per_block; // consecutive op-groups one block handles on its tile
    u32 og_stride;    // 1; level-sum launches (KsGroups::sum_out): the block's op-groups are n_ogb apart -- the same eight ciphertexts of every group
    // latency shape (few ops): the digits of a tile are cut into n_split groups, one block each (blockIdx.y); group g writes its canonical
    // partial sums to part[g][op * 2 + k][L + 1][N] and k_k3_combine adds them up (and runs the special prime's inverse row pass)
    int n_split;
    u64 *part;
    unsigned char tt_list[64];
};

// Digit rows reach the wave through an LDS landing buffer filled by LDS-DMA one step ahead.
// FUSE: the mod-down of the key switch is finished here instead of in k_floor_rows: the tile's sums never leave the
// registers — the wave transforms the matching rows of the special-prime correction (forward row pass, same tile twiddles)
// and writes (T - NTT(delta)) * P^-1 + c01 straight into c01.  Needs the special prime's sums first: the caller launches the
// special-prime tiles, the inverse transform and k_floor_colsn before the data-prime tiles.
// TENSOR (FUSE only): the launch belongs to a ct x ct multiply whose c0, c1 this kernel computes from the operand rows (K3Args::ta); an
// instantiation of its own, so that the other users of the fused kernel keep their register allocation.
// WAVES: 8 (throughput shape: one block per CU, two waves per SIMD, each wave on its own op; LDS: 8 x 8.5 KiB exchange + 8 x 8 KiB
// DMA landing + the tile's twiddles) or 1 (latency shape: one wave per block, the digits of a tile dealt to n_split blocks).
// GROUPED: every group of ops has its own key (he355_rotate_sum: all nodes of a trie level in one launch); ops of a group are consecutive,
// so the waves of a block still share their key rows through L1 / L2 for a group's worth of ops.  Instantiations of its own.
template <class Ar, int WAVES, bool FUSE = false, bool TENSOR = false, bool GROUPED = false>
__global__ void __launch_bounds__(WAVES * 64) k_k3(K3Args A, const PrimeDev *primes)
{
#define K3_BID_X blockIdx.x
#define K3_BID_Y blockIdx.y
#define K3_LDS_DECL                                                                                                            \
    __shared__ u64 lds[kWaves][kLdsRow];                                                                                       \
    __shared__ __attribute__((aligned(16))) u64 stage[kWaves][kRowN];                                                          \
    __shared__ __attribute__((aligned(16))) unsigned char twl_raw[kF64 ? kRowTw * 8 : kRowTw * 16];
#include "k3_body.inc"
#undef K3_BID_X
#undef K3_BID_Y
#undef K3_LDS_DECL
}
// Both engines' tiles of one stage in ONE launch: the fp64-engine blocks (0 .. n_f - 1) and the u64-engine blocks side by side, on one
// set of LDS arrays.  Two uses.  Latency shape (k_k3_dual, single-wave blocks; blockIdx.y = digit group, each engine with its own
// count, surplus blocks leave at once): a batch-1 key switch is a chain of launches of 5-12 us each, bound by launch and
// dependent-chain latency, and one launch instead of two per stage takes the shorter engine's time out of the chain.  Small grids of
// the throughput shape (k_k3_dual8: tens of ciphertexts at N = 8192 are 30-130 blocks per engine on 256 CUs): the two launches
// would run one after the other on a mostly idle chip.  (The same body text as k_k3, behind a function boundary here -- which is
// why the large grids, the headline's among them, keep the template kernels.)
// (ARGS: `const K3Args &` for the single-wave instantiations, `const K3Args` for the 8-wave ones -- whichever lets the compiler keep the
// argument struct in the kernarg segment instead of copying it to scratch: 0 vs 416 B and 68-244 vs 880-960 B per lane, tools/kres.py)
template <class Ar, int WAVES, bool FUSE, bool TENSOR, bool GROUPED, class ARGS>
__device__ __forceinline__ void k3_body_fn(ARGS A, const PrimeDev *primes, const unsigned bid_x, const unsigned bid_y, u64 (*lds)[kLdsRow],
                                           u64 (*stage)[kRowN], unsigned char *twl_raw)
{
#define K3_BID_X bid_x
#define K3_BID_Y bid_y
#define K3_LDS_DECL
#include "k3_body.inc"
#undef K3_BID_X
#undef K3_BID_Y
#undef K3_LDS_DECL
}
// the body's early-exit test (k3_body.inc: `if (tile >= total_tiles) return;`) for a block of WAVES waves
template <int WAVES> __device__ __forceinline__ bool k3_block_has_tile(const K3Args &A, unsigned bid_x)
{
    const u64 n_og = (A.n_ops + WAVES - 1) / WAVES;
    const u64 n_ogb = (n_og + A.og_per_block - 1) / A.og_per_block;
    const u64 total_tiles = (u64)A.n_tt << A.logn1;
    const u64 s = bid_x >> 3, xcd = bid_x & 7;
    return (s / n_ogb) * 8 + xcd < total_tiles;
}
__global__ void __launch_bounds__(64) k_k3_dual(K3Args AF, K3Args AU, unsigned n_u, const PrimeDev *primes)
{
    __shared__ u64 lds[1][kLdsRow];
    __shared__ __attribute__((aligned(16))) u64 stage[1][kRowN];
    __shared__ __attribute__((aligned(16))) unsigned char twl_raw[kRowTw * 16];
    if (blockIdx.x < n_u) { // the u64-engine blocks first: the longer ones (see k_k3_dual8)
        if ((int)blockIdx.y < (AU.n_split > 1 ? AU.n_split : 1))
            k3_body_fn<ArU64, 1, false, false, false, const K3Args &>(AU, primes, blockIdx.x, blockIdx.y, lds, stage, twl_raw);
    } else {
        if ((int)blockIdx.y < (AF.n_split > 1 ? AF.n_split : 1))
            k3_body_fn<ArF64, 1, false, false, false, const K3Args &>(AF, primes, blockIdx.x - n_u, blockIdx.y, lds, stage, twl_raw);
    }
}
// (the u64-engine blocks come FIRST: they run 2.5 times as long as an fp64-engine block, and dispatched last they would be the launch's tail;
// UW = 4: the u64-engine blocks work with their first four waves, one per SIMD -- see launch_k3)
template <bool FUSE, bool TENSOR, bool GROUPED, int UW = 8>
__global__ void __launch_bounds__(512) k_k3_dual8(K3Args AF, K3Args AU, unsigned n_u, const PrimeDev *primes)
{
    __shared__ u64 lds[8][kLdsRow];
    __shared__ __attribute__((aligned(16))) u64 stage[8][kRowN];
    __shared__ __attribute__((aligned(16))) unsigned char twl_raw[kRowTw * 16];
    if (blockIdx.x < n_u) {
        if (UW == 4 && threadIdx.x >= 256) {
            // waves 4-7 of a four-wave block do no work, but every wave of the block meets the body's one workgroup barrier (the
            // twiddle staging) -- or none does, when the block has no tile
            if (k3_block_has_tile<4>(AU, blockIdx.x)) __syncthreads();
            return;
        }
        k3_body_fn<ArU64, UW, FUSE, TENSOR, GROUPED, const K3Args>(AU, primes, blockIdx.x, 0, lds, stage, twl_raw);
    } else {
        k3_body_fn<ArF64, 8, FUSE, TENSOR, GROUPED, const K3Args>(AF, primes, blockIdx.x - n_u, 0, lds, stage, twl_raw);
    }
}

// The sums of a digit-split k_k3 launch: part [n_split][n_ops * 2][L + 1][N] canonical -> t (data primes, canonical NTT form) and tpr
// (special prime, after the inverse row pass), exactly what the unsplit launch leaves.  One wave per (op, k, tt, row).
struct K3CombineArgs {
    const u64 *part;
    u64 *t, *tpr;
    u64 n_ops;
    int n_split, n_split_u64, L, K, logn1, ckks; // digit groups of the fp64-engine / u64-engine tiles
};
__global__ void __launch_bounds__(kBlock) k_k3_combine(K3CombineArgs A, const PrimeDev *primes)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n1 = 1u << A.logn1;
    const u64 N = (u64)n1 << kRowLog;
    const u64 total = A.n_ops * 2 * (u64)(A.L + 1) * n1;
    u64 job = (u64)blockIdx.x * kWaves + wave;
    const bool valid = job < total;
    if (!valid) job = total - 1;
    const u32 a_row = (u32)(job & (n1 - 1));
    const u64 pt = job >> A.logn1;
    const int tt = (int)(pt % (A.L + 1));
    const u64 ok = pt / (A.L + 1); // op * 2 + k
    const int t = tt == A.L ? A.K - 1 : tt;
    const PrimeDev &P = primes[t];
    const u64 rowoff = (u64)a_row << kRowLog, q = P.q;
    u64 v[kRowE];
    load_rowC(A.part + (ok * (A.L + 1) + tt) * N + rowoff, lane, v);
    const int groups = P.f64 ? A.n_split : A.n_split_u64;
    for (int g = 1; g < groups; ++g) {
        u64 w[kRowE];
        load_rowC(A.part + (((u64)g * A.n_ops * 2 + ok) * (A.L + 1) + tt) * N + rowoff, lane, w);
#pragma unroll
        for (int r = 0; r < kRowE; ++r) v[r] = addmod(v[r], w[r], q);
    }
    const bool inv_here = tt == A.L || !A.ckks;
    if (!inv_here) {
        if (valid) store_rowC(A.t + (ok * A.L + tt) * N + rowoff, lane, v);
        return;
    }
    const bool last = A.logn1 == 0;
    u64 *dst = tt < A.L ? A.t + (ok * A.L + tt) * N + rowoff : A.tpr + ok * N + rowoff;
    if (P.f64) {
        const ArF64 ar = make_ar(P, (ArF64 *)nullptr);
        double x[kRowE];
#pragma unroll
        for (int r = 0; r < kRowE; ++r) x[r] = ar.from_canon(v[r]);
        wave_rows_inv(ar, P, last, n1 + a_row, lane, lds[wave], x);
#pragma unroll
        for (int r = 0; r < kRowE; ++r) v[r] = last ? ar.to_canon(x[r]) : ar.to_raw(x[r]);
    } else {
        const ArU64 ar = make_ar(P, (ArU64 *)nullptr);
        u64 x[kRowE];
#pragma unroll
        for (int r = 0; r < kRowE; ++r) x[r] = v[r];
        wave_rows_inv(ar, P, last, n1 + a_row, lane, lds[wave], x);
#pragma unroll
        for (int r = 0; r < kRowE; ++r) v[r] = last ? ar.to_canon(x[r]) : ar.to_raw(x[r]);
    }
    if (valid) store_rowA(dst, lane, v);
}

// =======================================================================================================
// Floor step, column half (key-switch mod-down and rescale): finish the inverse transform of the source
// residue, add floor(s/2), reduce into every target prime, subtract floor(s/2), forward column pass.
// =======================================================================================================
struct FloorColsArgs {
    const u64 *src;
    u64 *dst;
    int src_prime, n_tgt, K;
    const FloorConst *fc;
    int tgt_first, dst_ntgt; // targets [tgt_first, tgt_first + n_tgt) of a destination slab laid out for dst_ntgt targets
    // optional (MERGE): an earlier floor step's SOURCE ([n_polys][N], after the inverse row pass, prime src2_prime): its correction is
    // formed here in coefficient form and folded in BEFORE the column pass, delta = delta2 + src2^-1 * delta1 mod q_i, so mod-down and
    // rescale share ONE column pass and ONE row transform per target (both passes are linear) and the earlier correction slab is
    // neither written for these targets nor read back
    const u64 *src2;
    int src2_prime;
    u64 f64_mask; // bit i: key prime i belongs to the fp64 engine
    int tsplit;   // latency shape: the targets of a column are dealt to tsplit blocks (blockIdx.y)
};

// canonical coefficients + floor(s/2) of one column of a source residue after its inverse row pass
template <int LOGN1>
__device__ __forceinline__ void floor_source_column(const PrimeDev &Ps, const u64 *src, int col, u64 c[1 << LOGN1])
{
    constexpr int N1 = 1 << LOGN1;
    if (LOGN1 == 0) {
        c[0] = src[col];
    } else if (Ps.f64) {
        const ArF64 ar = make_ar(Ps, (ArF64 *)nullptr);
        double x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = ar.from_raw(src[(a << kRowLog) + col]);
        col_inv<ArF64, LOGN1>(ar, x, ctw(Ps.inv), Ps.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) c[a] = ar.to_canon(x[a]);
    } else {
        const ArU64 ar = make_ar(Ps, (ArU64 *)nullptr);
        u64 x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = src[(a << kRowLog) + col];
        col_inv<ArU64, LOGN1>(ar, x, ctw(Ps.inv), Ps.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) c[a] = ar.to_canon(x[a]);
    }
    const u64 qs = Ps.q, half = qs >> 1;
#pragma unroll
    for (int a = 0; a < N1; ++a) c[a] = addmod(c[a], half, qs);
}

// The corrections are formed in the target's own engine (round 3; the first form used 64-bit Barrett reductions, HISTORY.md): under an
// fp64-engine target a source residue below 2^52 enters as the double it is (re-centred only where q_s > 2 q_i), a wider one as
// hi * (2^32 mod q_i) + lo, one exact fp64 product (8 instructions instead of ~35 per element for the 60-bit special prime).  Per-target
// constants sit in a small LDS table filled once per block (wave-uniform ds_reads, no scalar registers), the column twiddles come from
// PrimeDev::colw through the same software pipeline as k_k2n's, rows leave through buffer stores whose addresses cost no VALU
// instruction.
struct FcnConsts { // one per target prime, as the fp64 engine wants them
    double half1, half2;   // floor(s/2) mod q_i for the source / the merged earlier source
    double inv2, inv2_i;   // src2^-1 mod q_i and fl(inv2 / q_i)
    double pow32, qd, qinv;
    double recentre;       // bit 0: source 1 needs re-centring (q_s > 2 q_i), bit 1: source 2 does
};

// WIDE1: source 1 is held as integers (q_s >= 2^52), else as doubles
template <int LOGN1, bool MERGE, bool WIDE1>
__device__ __forceinline__ void fcn_targets_f64(const FloorColsArgs &A, const PrimeDev *primes, u64 poly, int col, u64 mask,
                                                const typename std::conditional<WIDE1, u64, double>::type (&c)[1 << LOGN1],
                                                const u64 *park /* [N1][kBlock] when MERGE */, bool wide2, const FcnConsts *ctab)
{
    constexpr int N1 = 1 << LOGN1;
    constexpr u64 N = (u64)N1 << kRowLog;
    constexpr int LA = LOGN1 < 4 ? LOGN1 : 4;
    if (!mask) return;
    const u32 off8 = (u32)col << 3;
    const cprime_t cp = (cprime_t)(unsigned long long)primes;
    u64 m = mask;
    int t = __builtin_ctzll(m);
    d16_t wa = load_colw16(cp, t, 0);
    FcnConsts k = ctab[t];
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0), see k2n_fast_targets
    for (;;) {
        m &= m - 1;
        const int tn = m ? __builtin_ctzll(m) : t;
        d16_t wb = wa;
        if constexpr (LOGN1 == 5) wb = load_colw16(cp, t, 16);
        __builtin_amdgcn_sched_barrier(0);
        ArF64 ar;
        ar.q = k.qd; ar.qinv = k.qinv; ar.ninv = 0; ar.ninv_i = 0;
        const int flags = (int)k.recentre;
        double x[N1];
        // delta2 = [source 1]_{q_i} - floor(s/2)
#pragma unroll
        for (int a = 0; a < N1; ++a) {
            double v;
            if constexpr (WIDE1) v = lift_wide(ar, c[a], k.pow32);
            else v = (flags & 1) ? ar.renorm((double)c[a]) : (double)c[a];
            x[a] = v - k.half1;
        }
        if constexpr (MERGE) { // + src2^-1 * delta1, delta1 = [source 2]_{q_i} - floor(src2/2)
#pragma unroll
            for (int a = 0; a < N1; ++a) {
                const u64 v2 = park[a * kBlock + threadIdx.x];
                double v;
                if (wide2) v = lift_wide(ar, v2, k.pow32);
                else { v = u52_to_f64(v2); if (flags & 2) v = ar.renorm(v); }
                x[a] += ar.mulmod_c(v - k.half2, k.inv2, k.inv2_i);
            }
        }
        // the next target's constants (LDS table): requested here, needed after the stores
        const FcnConsts kn = ctab[tn];
#pragma unroll
        for (int s = 0; s < LA; ++s) {
            const int gap = N1 >> (s + 1);
#pragma unroll
            for (int a = 0; a < N1; ++a) {
                if (a & gap) continue;
                const double tw = ar.mulmod_vv(x[a + gap], wa[(1 << s) + (a / (2 * gap))]);
                const double X = x[a];
                x[a] = X + tw; x[a + gap] = X - tw;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the stage-4 entries (and the constants just read)
        const d16_t wa_n = load_colw16(cp, tn, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LOGN1 == 5) {
#pragma unroll
            for (int a = 0; a < N1; a += 2) {
                const double tw = ar.mulmod_vv(x[a + 1], wb[a >> 1]);
                const double X = x[a];
                x[a] = X + tw; x[a + 1] = X - tw;
            }
        }
        u64 v[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) v[a] = ar.to_raw(x[a]);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        store_word_rows<N1>(poly_rsrc(A.dst + (poly * A.dst_ntgt + t) * N, (u32)N1 * kSlotBytes), off8, v);
        if (!m) break;
        t = tn; wa = wa_n; k = kn;
    }
}

template <int LOGN1, bool MERGE>
__global__ void __launch_bounds__(kBlock, 2) k_floor_colsn(FloorColsArgs A, const PrimeDev *primes)
{
    constexpr int N1 = 1 << LOGN1;
    constexpr u64 N = (u64)N1 << kRowLog;
    const u64 poly = blockIdx.x >> 2;
    const int col = ((blockIdx.x & 3) << 8) | threadIdx.x;
    __shared__ u64 park[MERGE ? N1 : 1][MERGE ? kBlock : 1];
    __shared__ FcnConsts ctab[kMaxPrimes];
    const PrimeDev &Ps = primes[A.src_prime];
    const u64 qs = Ps.q, qs2 = MERGE ? primes[A.src2_prime].q : 0;
    if ((int)threadIdx.x >= A.tgt_first && (int)threadIdx.x < A.tgt_first + A.n_tgt) { // this block's constants, one thread per target
        const int i = threadIdx.x;
        const PrimeDev &Pi = primes[i];
        FcnConsts k;
        k.half1 = (double)A.fc[A.src_prime * A.K + i].half_mod;
        k.half2 = 0; k.inv2 = 0; k.inv2_i = 0;
        int flags = qs > 2 * Pi.q ? 1 : 0;
        if (MERGE) {
            const FloorConst f2 = A.fc[A.src2_prime * A.K + i];
            k.half2 = (double)f2.half_mod; k.inv2 = f2.inv_d; k.inv2_i = f2.inv_i;
            flags |= qs2 > 2 * Pi.q ? 2 : 0;
        }
        k.pow32 = Pi.pow32; k.qd = Pi.qd; k.qinv = Pi.qinv; k.recentre = (double)flags;
        ctab[i] = k;
    }
    u64 c[N1];
    floor_source_column<LOGN1>(Ps, A.src + poly * N, col, c);
    if constexpr (MERGE) {
        u64 c2[N1];
        floor_source_column<LOGN1>(primes[A.src2_prime], A.src2 + poly * N, col, c2);
#pragma unroll
        for (int a = 0; a < N1; ++a) park[a][threadIdx.x] = c2[a]; // each thread reads back only what it wrote itself
    }
    __syncthreads(); // the constants table
    u64 tgt = ((A.tgt_first + A.n_tgt >= 64 ? ~(u64)0 : (((u64)1 << (A.tgt_first + A.n_tgt)) - 1))) & ~(((u64)1 << A.tgt_first) - 1);
    tgt = split_mask(tgt, (int)blockIdx.y, A.tsplit);
    const u64 f64t = tgt & A.f64_mask;
    const bool wide2 = MERGE && (qs2 >> 52) != 0;
    if (qs >> 52) {
        fcn_targets_f64<LOGN1, MERGE, true>(A, primes, poly, col, f64t, c, &park[0][0], wide2, ctab);
    } else {
        double cd[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) cd[a] = u52_to_f64(c[a]);
        fcn_targets_f64<LOGN1, MERGE, false>(A, primes, poly, col, f64t, cd, &park[0][0], wide2, ctab);
#pragma unroll
        for (int a = 0; a < N1; ++a) c[a] = f64_to_u52(cd[a]);
    }
    // u64-engine targets: integers throughout
    const u32 off8 = (u32)col << 3;
    for (u64 m = tgt & ~A.f64_mask; m; m &= m - 1) {
        const int i = __builtin_ctzll(m);
        const PrimeDev &Pi = primes[i];
        const u64 qi = Pi.q;
        const u64 half_i = A.fc[A.src_prime * A.K + i].half_mod;
        const ModU64 mi = make_modu(Pi);
        const ArU64 ar = make_ar(Pi, (ArU64 *)nullptr);
        u64 dl[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) dl[a] = submod(qs > qi ? barrett64(c[a], mi) : c[a], half_i, qi);
        if constexpr (MERGE) { // [0,q) + [0,q): a valid lazy input of the column pass
            const FloorConst f2 = A.fc[A.src2_prime * A.K + i];
#pragma unroll
            for (int a = 0; a < N1; ++a) {
                const u64 v2 = park[a][threadIdx.x];
                const u64 d1 = submod(qs2 > qi ? barrett64(v2, mi) : v2, f2.half_mod, qi);
                dl[a] += mul_pre(d1, f2.inv, f2.inv_shoup, qi);
            }
        }
        col_fwd<ArU64, LOGN1>(ar, dl, ctw(Pi.fwd));
        store_word_rows<N1>(poly_rsrc(A.dst + (poly * A.dst_ntgt + i) * N, (u32)N1 * kSlotBytes), off8, dl);
    }
}

// =======================================================================================================
// Floor step, row half: out = (tsrc - NTT(cols)) * s^-1 (+ addend) mod q_i; optional inverse-row-pass tail
// =======================================================================================================
struct FloorRowsDev {
    FloorRowsArgs a;
    const FloorConst *fc;
    u64 n_ops;
    int K, logn1;
    int n_i;
    u32 jobs_per_block; // multiple of kWaves
    unsigned char i_list[64];
};

// (a device function: the latency shape runs both engines' instantiations in one launch, k_floor_rows_dual, on one set of LDS buffers)
template <class Ar, bool TAIL>
__device__ __forceinline__ void floor_rows_block(const FloorRowsDev &A, const PrimeDev *primes, const unsigned bid_x, u64 (*lds)[kLdsRow],
                                                 unsigned char *twl_raw)
{
    typedef typename Ar::T T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n1 = 1u << A.logn1;
    const u64 N = (u64)n1 << kRowLog;
    // A block owns one (target prime, row) tile and walks jobs_per_block consecutive (op, poly) jobs of it, one job per
    // wave per step: the row's twiddles are staged in LDS once and every later access is a ds_read.
    const u64 n_jobs = A.n_ops * A.a.n_src;
    const u64 n_jb = (n_jobs + A.jobs_per_block - 1) / A.jobs_per_block;
    const u64 tile = bid_x / n_jb;
    const u64 j_begin = (bid_x % n_jb) * A.jobs_per_block;
    const u64 j_end = j_begin + A.jobs_per_block < n_jobs ? j_begin + A.jobs_per_block : n_jobs;
    const int i = A.i_list[tile >> A.logn1];
    const u32 a_row = (u32)(tile & (n1 - 1));
    const PrimeDev &P = primes[i];
    const Ar ar = make_ar(P, (Ar *)nullptr);
    const FloorConst fc = A.fc[A.a.src_prime * A.K + i];
    const u64 rowoff = (u64)a_row << kRowLog;
    const auto twr = stage_row_twiddles<Ar, kBlock>(P, ar, n1 + a_row, twl_raw);
    __syncthreads();
    for (u64 j = j_begin + wave; j < j_end; j += kWaves) { // no block-level synchronisation inside: waves run independently
        const int k = (int)(j % A.a.n_src);
        const u64 op = j / A.a.n_src;
        T x[kRowE];
        u64 v[kRowE];
        load_rowA(A.a.cols + ((op * A.a.n_src + k) * A.a.n_tgt + i) * N + rowoff, lane, v);
#pragma unroll
        for (int r = 0; r < kRowE; ++r) x[r] = ar.from_raw(v[r]);
        wave_rows_fwd(ar, twr, lane, lds[wave], x);
        {
            u64 tv[kRowE], av[kRowE];
            load_rowC(A.a.tsrc + op * A.a.tsrc_op_stride + k * A.a.tsrc_poly_stride + (u64)i * N + rowoff, lane, tv);
            if (A.a.addend) load_rowC(A.a.addend + op * A.a.add_op_stride + k * A.a.add_poly_stride + (u64)i * N + rowoff, lane, av);
#pragma unroll
            for (int r = 0; r < kRowE; ++r) v[r] = ar.floor_fin(tv[r], x[r], fc.inv, fc.inv_shoup, fc.inv_d, fc.inv_i, A.a.addend ? av[r] : 0);
        }
        if (A.a.out) store_rowC(A.a.out + op * A.a.out_op_stride + k * A.a.out_poly_stride + (u64)i * N + rowoff, lane, v);
        if (TAIL) { // this prime is the next floor step's source: start its inverse transform right here
            const bool last = A.logn1 == 0;
#pragma unroll
            for (int r = 0; r < kRowE; ++r) x[r] = ar.from_canon(v[r]);
            wave_rows_inv(ar, P, last, n1 + a_row, lane, lds[wave], x);
#pragma unroll
            for (int r = 0; r < kRowE; ++r) v[r] = last ? ar.to_canon(x[r]) : ar.to_raw(x[r]);
            store_rowA(A.a.tail + (op * A.a.n_src + k) * N + rowoff, lane, v);
        }
    }
}
template <class Ar, bool TAIL>
__global__ void __launch_bounds__(kBlock) k_floor_rows(FloorRowsDev A, const PrimeDev *primes)
{
    constexpr bool kF64 = std::is_same<Ar, ArF64>::value;
    __shared__ u64 lds[kWaves][kLdsRow];
    __shared__ __attribute__((aligned(16))) unsigned char twl_raw[kF64 ? kRowTw * 8 : kRowTw * 16];
    floor_rows_block<Ar, TAIL>(A, primes, blockIdx.x, lds, twl_raw);
}
// Latency shape: the fp64-engine targets (blocks 0 .. n_f - 1) and the u64-engine targets of one floor step in ONE launch (see k_k3_dual)
__global__ void __launch_bounds__(kBlock) k_floor_rows_dual(FloorRowsDev AF, FloorRowsDev AU, unsigned n_f, const PrimeDev *primes)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    __shared__ __attribute__((aligned(16))) unsigned char twl_raw[kRowTw * 16];
    if (blockIdx.x < n_f) floor_rows_block<ArF64, false>(AF, primes, blockIdx.x, lds, twl_raw);
    else floor_rows_block<ArU64, false>(AU, primes, blockIdx.x - n_f, lds, twl_raw);
}

// =======================================================================================================
// BFV: BEHZ base extension / floor kernels (one lane = one coefficient across all residues), coefficient-form Galois,
// and the coefficient-form key-switch tails
// =======================================================================================================
__device__ __forceinline__ ModU64 mod_of(const PrimeDev *primes, int idx) { return make_modu(primes[idx]); }

// ML / MB: compile-time bounds of the bases q and B (loops fully unrolled under them, per-residue values in registers): <4, 6> serves
// the reference's default parameter sets ({60,40,40} and {60,40,40,40}), <kBehzMaxL, kBehzMaxB> everything else.
template <int ML, int MB>
__global__ void __launch_bounds__(kBlock) k_behz_extend(BehzDev Z, const PrimeDev *primes, BehzSrc src_of, u64 *xq, u64 *xbsk, u64 n_polys, int logN)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 n = gid & (((u64)1 << logN) - 1);
    const u64 pid = gid >> logN; // (ciphertext item, which poly)
    if (pid >= n_polys) return;
    const int k = (int)(pid & 1);
    constexpr int kUnrollB = MB <= 6 ? MB + 1 : 1; // the small instantiation unrolls its residue loops, the large one keeps them rolled
    const int L = Z.L, S = Z.nB + 1;
    const u64 N = (u64)1 << logN, P1 = (u64)L * N;
    const u64 *src = behz_src_ct(src_of, pid >> 1, 2 * P1) + (u64)k * P1 + n;
    u64 x[ML], tmp[ML], rmt;
#pragma unroll
    for (int i = 0; i < ML; ++i)
        if (i < L) {
            x[i] = src[(u64)i * N];
            xq[(pid * L + i) * N + n] = x[i];
        }
    behz_ext_prepare<ML>(Z, primes, L, x, tmp, rmt);
#pragma unroll kUnrollB
    for (int j = 0; j < MB + 1; ++j) {
        if (j >= S) break;
        xbsk[(pid * S + j) * N + n] = behz_ext_residue<ML>(Z, mod_of(primes, Z.bsk_prime[j]), L, j, tmp, rmt);
    }
}

// Forward column pass of one residue's column held in registers (canonical in, raw out), by the engine that owns the prime.
template <int LOGN1> __device__ __forceinline__ void col_fwd_store(const PrimeDev &P, const u64 v[1 << LOGN1], u64 *dst)
{
    constexpr int N1 = 1 << LOGN1;
    if (P.f64) {
        const ArF64 ar = make_ar(P, (ArF64 *)nullptr);
        double y[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) y[a] = ar.from_canon(v[a]);
        col_fwd<ArF64, LOGN1>(ar, y, ctw(P.fwd));
#pragma unroll
        for (int a = 0; a < N1; ++a) dst[a << kRowLog] = ar.to_raw(y[a]);
    } else {
        const ArU64 ar = make_ar(P, (ArU64 *)nullptr);
        u64 y[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) y[a] = v[a];
        col_fwd<ArU64, LOGN1>(ar, y, ctw(P.fwd));
#pragma unroll
        for (int a = 0; a < N1; ++a) dst[a << kRowLog] = y[a];
    }
}
// BEHZ steps (1)-(2) AND the forward column pass of all L + S residues in one kernel (N <= 16384, L <= 4, nB <= 6), so that the
// coefficient-form copies xq / xbsk never exist in HBM (k_behz_extend + k_cols_fwd x2 write them and read them back: 2 (L + S)
// polynomial transfers per input polynomial saved).  A block = 64 columns x N1 rows of one input polynomial, one wave per row:
// phase 1, lane (row e, column c) extends its coefficient to Bsk and parks the L + S residues in LDS [residue][e][c]; phase 2, wave w
// takes residues w, w + N1, ...: each lane runs that residue's column pass on the N1 values of its column (wave-uniform prime, so
// the engine branch does not diverge) and stores the raw rows.
template <int LOGN1, int ML, int MB, bool EXACT>
__global__ void __launch_bounds__(64 << LOGN1) k_behz_extend_cols(BehzDev Z, const PrimeDev *primes, BehzSrc src_of, u64 *xq, u64 *xbsk)
{
    constexpr int N1 = 1 << LOGN1;
    extern __shared__ u64 behz_sm[]; // [L + S][N1][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 pid = blockIdx.x >> 4; // (ciphertext item, which poly); 16 column groups per polynomial
    const u64 col = ((blockIdx.x & 15) << 6) | lane;
    const int k = (int)(pid & 1);
    const int L = EXACT ? ML : Z.L, S = (EXACT ? MB : Z.nB) + 1;
    const u64 N = (u64)N1 << kRowLog, P1 = (u64)L * N;
    {
        const u64 *src = behz_src_ct(src_of, pid >> 1, 2 * P1) + (u64)k * P1 + ((u64)wave << kRowLog) + col;
        u64 x[ML], tmp[ML], rmt;
#pragma unroll
        for (int i = 0; i < ML; ++i)
            if (i < L) {
                x[i] = src[(u64)i * N];
                behz_sm[((i * N1 + wave) << 6) | lane] = x[i];
            }
        behz_ext_prepare<ML>(Z, primes, L, x, tmp, rmt);
#pragma unroll
        for (int j = 0; j < MB + 1; ++j) {
            if (j >= S) break;
            behz_sm[(((L + j) * N1 + wave) << 6) | lane] = behz_ext_residue<ML>(Z, mod_of(primes, Z.bsk_prime[j]), L, j, tmp, rmt);
        }
    }
    __syncthreads();
    for (int rr = wave; rr < L + S; rr += N1) {
        const bool isq = rr < L;
        const PrimeDev &P = primes[isq ? rr : Z.bsk_prime[rr - L]];
        u64 *dst = (isq ? xq + (pid * L + rr) * N : xbsk + (pid * S + (rr - L)) * N) + col;
        u64 v[N1];
#pragma unroll
        for (int e = 0; e < N1; ++e) v[e] = behz_sm[((rr * N1 + e) << 6) | lane];
        col_fwd_store<LOGN1>(P, v, dst);
    }
}

// BEHZ steps (3)-(5) on rows: forward row pass of the four polynomials a0, a1, b0, b1 of one (op, residue, row), the dyadic tensor
// c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1, and the inverse row pass of the three products, in ONE kernel: a block is four waves,
// wave w transforms polynomial w's row; the canonical NTT-form rows meet in LDS (each wave's own exchange buffer, free once its
// transform is done; every wave uses the same lane <-> element map, so the hand-over is slot-for-slot); waves 0..2 form one product
// each and run its inverse row pass.  Replaces k_rows_fwd (4 polynomials) + k_tensor4 + k_rows_inv (3 polynomials): 35 instead of 105
// polynomial transfers through HBM per (op, residue).  x [n*4][Lx][N] after the forward column pass (raw; canonical when N = 1024),
// d [n*3][Lx][N] ready for the inverse column pass.
struct BehzRowsArgs {
    const u64 *x[2]; // base q, base Bsk: [n*4][Lx][N]
    u64 *d[2];       //                   [n*3][Lx][N]
    int Lx[2];
    u64 n_ops;
    int logn1, n_r;
    unsigned char r_base[64], r_idx[64], r_prime[64]; // the residues of one arithmetic engine: base, index in it, device prime
};
// (body as a device function: k_behz_rows_tensor_dual runs the blocks of both engines in one launch when the grids are small)
template <class Ar>
__device__ __forceinline__ void behz_rows_tensor_block(const BehzRowsArgs &A, const PrimeDev *primes, const u64 job, u64 (*lds)[kLdsRow])
{
    typedef typename Ar::T T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n1 = 1u << A.logn1;
    const u64 N = (u64)n1 << kRowLog;
    const u32 a_row = (u32)(job & (n1 - 1)); // job = (op, residue slot, row)
    const u64 orr = job >> A.logn1;
    const int slot = (int)(orr % A.n_r);
    const u64 op = orr / A.n_r;
    const int base = A.r_base[slot], r = A.r_idx[slot], Lx = A.Lx[base];
    const PrimeDev &P = primes[A.r_prime[slot]];
    const Ar ar = make_ar(P, (Ar *)nullptr);
    const bool last = A.logn1 == 0;
    const u64 rowoff = ((u64)r << (A.logn1 + kRowLog)) + ((u64)a_row << kRowLog);
    const u64 *xin = A.x[base];
    u64 *dout = A.d[base];
    // phase 1: this wave's polynomial through the forward row pass; canonical values parked in its exchange buffer
    {
        T x[kRowE];
        u64 v[kRowE];
        load_rowA(xin + (op * 4 + wave) * Lx * N + rowoff, lane, v);
#pragma unroll
        for (int e = 0; e < kRowE; ++e) x[e] = last ? ar.from_canon(v[e]) : ar.from_raw(v[e]);
        wave_rows_fwd(ar, tw_table(gtw(P.fwd), n1 + a_row), lane, lds[wave], x);
#pragma unroll
        for (int e = 0; e < kRowE; ++e) lds[wave][(e << 6) | lane] = ar.to_canon(x[e]);
    }
    __syncthreads();
    // phase 2: one product per wave (waves 0..2), from the rows the other waves parked (slot-for-slot: same lane, same register)
    const bool worker = wave < 3; // wave-uniform; wave 3 only keeps the barriers company
    u64 c[kRowE];
    if (worker) {
        u64 p0[kRowE], p1[kRowE];
        const int ia = wave == 2 ? 1 : 0, ib = wave == 0 ? 2 : 3; // c0: a0 b0; c1: a0 b1 (+ a1 b0); c2: a1 b1
#pragma unroll
        for (int e = 0; e < kRowE; ++e) { p0[e] = lds[ia][(e << 6) | lane]; p1[e] = lds[ib][(e << 6) | lane]; }
        if (wave == 1) {
            u64 q0[kRowE], q1[kRowE];
#pragma unroll
            for (int e = 0; e < kRowE; ++e) { q0[e] = lds[1][(e << 6) | lane]; q1[e] = lds[2][(e << 6) | lane]; }
#pragma unroll
            for (int e = 0; e < kRowE; ++e)
                c[e] = ar.dy_out(ar.dy_add(ar.dy_mul(ar.dy_in(p0[e]), ar.dy_in(p1[e])), ar.dy_mul(ar.dy_in(q0[e]), ar.dy_in(q1[e]))));
        } else {
#pragma unroll
            for (int e = 0; e < kRowE; ++e) c[e] = ar.dy_out(ar.dy_mul(ar.dy_in(p0[e]), ar.dy_in(p1[e])));
        }
    }
    __syncthreads(); // every wave has read what it needs before any exchange buffer is written again
    if (!worker) return;
    T x[kRowE];
#pragma unroll
    for (int e = 0; e < kRowE; ++e) x[e] = ar.from_canon(c[e]);
    wave_rows_inv(ar, P, last, n1 + a_row, lane, lds[wave], x);
    u64 v[kRowE];
#pragma unroll
    for (int e = 0; e < kRowE; ++e) v[e] = last ? ar.to_canon(x[e]) : ar.to_raw(x[e]);
    store_rowA(dout + (op * 3 + wave) * Lx * N + rowoff, lane, v);
}
template <class Ar>
__global__ void __launch_bounds__(kBlock) k_behz_rows_tensor(BehzRowsArgs A, const PrimeDev *primes)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    behz_rows_tensor_block<Ar>(A, primes, blockIdx.x, lds);
}
__global__ void __launch_bounds__(kBlock) k_behz_rows_tensor_dual(BehzRowsArgs AF, BehzRowsArgs AU, unsigned n_f, const PrimeDev *primes)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    if (blockIdx.x < n_f) behz_rows_tensor_block<ArF64>(AF, primes, blockIdx.x, lds);
    else behz_rows_tensor_block<ArU64>(AU, primes, blockIdx.x - n_f, lds);
}

// BEHZ steps (4)-(5) when the operands were transformed once each (BehzSrc lists: he355_api.hip, bfv_multiply3): one WAVE per
// (result, residue, row, product k): it reads the canonical NTT-form rows of its operands where the one-off transform left them
// (layout C, as k_rows_fwd stores them), forms c0 = a0 b0, c1 = a0 b1 + a1 b0 or c2 = a1 b1 and runs the inverse row pass.  No
// barrier, no LDS hand-over; an operand row read by many results comes from L2 / MALL.  e [(ordinal * 2 + poly)][Lx][N],
// d [(r * 3 + k)][Lx][N] ready for the inverse column pass.
struct BehzTensorArgs {
    const u64 *e[2]; // base q, base Bsk
    u64 *d[2];
    int Lx[2];
    BehzSrc src;     // result -> operand ordinals
    u64 n_ops, op_offset;
    int logn1, n_r;
    unsigned char r_base[64], r_idx[64], r_prime[64];
};
template <class Ar>
__device__ __forceinline__ void behz_tensor_inv_block(const BehzTensorArgs &A, const PrimeDev *primes, const u64 total_jobs, const unsigned bid_x,
                                                      u64 (*lds)[kLdsRow])
{
    typedef typename Ar::T T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 job = (u64)bid_x * kWaves + wave; // (op, residue slot, row, k)
    if (job >= total_jobs) return;              // (waves are independent: no barrier below)
    const u32 n1 = 1u << A.logn1;
    const u64 N = (u64)n1 << kRowLog;
    const int k = (int)(job % 3);
    const u64 j1 = job / 3;
    const u32 a_row = (u32)(j1 & (n1 - 1));
    const u64 orr = j1 >> A.logn1;
    const int slot = (int)(orr % A.n_r);
    const u64 op = orr / A.n_r;
    const int base = A.r_base[slot], r = A.r_idx[slot], Lx = A.Lx[base];
    const PrimeDev &P = primes[A.r_prime[slot]];
    const Ar ar = make_ar(P, (Ar *)nullptr);
    const bool last = A.logn1 == 0;
    const u64 rowoff = ((u64)r << (A.logn1 + kRowLog)) + ((u64)a_row << kRowLog);
    const u64 oa = ord_a(A.src, A.op_offset + op), ob = ord_b(A.src, A.op_offset + op);
    const u64 *ea = A.e[base] + oa * 2 * Lx * N + rowoff, *eb = A.e[base] + ob * 2 * Lx * N + rowoff;
    const u64 P1 = (u64)Lx * N;
    T x[kRowE];
    {
        u64 p0[kRowE], p1[kRowE];
        load_rowC(ea + (k == 2 ? P1 : 0), lane, p0); // c0: a0 b0; c1: a0 b1 (+ a1 b0); c2: a1 b1
        load_rowC(eb + (k == 0 ? 0 : P1), lane, p1);
        if (k == 1) {
            u64 q0[kRowE], q1[kRowE];
            load_rowC(ea + P1, lane, q0);
            load_rowC(eb, lane, q1);
#pragma unroll
            for (int e = 0; e < kRowE; ++e)
                x[e] = ar.from_canon(ar.dy_out(ar.dy_add(ar.dy_mul(ar.dy_in(p0[e]), ar.dy_in(p1[e])), ar.dy_mul(ar.dy_in(q0[e]), ar.dy_in(q1[e])))));
        } else {
#pragma unroll
            for (int e = 0; e < kRowE; ++e) x[e] = ar.from_canon(ar.dy_out(ar.dy_mul(ar.dy_in(p0[e]), ar.dy_in(p1[e]))));
        }
    }
    wave_rows_inv(ar, P, last, n1 + a_row, lane, lds[wave], x);
    u64 v[kRowE];
#pragma unroll
    for (int e = 0; e < kRowE; ++e) v[e] = last ? ar.to_canon(x[e]) : ar.to_raw(x[e]);
    store_rowA(A.d[base] + (op * 3 + k) * Lx * N + rowoff, lane, v);
}
template <class Ar>
__global__ void __launch_bounds__(kBlock) k_behz_tensor_inv(BehzTensorArgs A, const PrimeDev *primes, u64 total_jobs)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    behz_tensor_inv_block<Ar>(A, primes, total_jobs, blockIdx.x, lds);
}
__global__ void __launch_bounds__(kBlock) k_behz_tensor_inv_dual(BehzTensorArgs AF, BehzTensorArgs AU, u64 jobs_f, u64 jobs_u, unsigned n_f, const PrimeDev *primes)
{
    __shared__ u64 lds[kWaves][kLdsRow];
    if (blockIdx.x < n_f) behz_tensor_inv_block<ArF64>(AF, primes, jobs_f, blockIdx.x, lds);
    else behz_tensor_inv_block<ArU64>(AU, primes, jobs_u, blockIdx.x - n_f, lds);
}

template <int ML, int MB>
__global__ void __launch_bounds__(kBlock) k_behz_floor_sk(BehzDev Z, const PrimeDev *primes, const u64 *dq, const u64 *ds, u64 *out, u64 n_polys, int logN)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 n = gid & (((u64)1 << logN) - 1);
    const u64 pid = gid >> logN; // (op, k)
    if (pid >= n_polys) return;
    constexpr int kUnrollB = MB <= 6 ? MB + 1 : 1, kUnrollL = ML <= 4 ? ML : 1;
    const int L = Z.L, S = Z.nB + 1;
    const u64 N = (u64)1 << logN;
    u64 vq[ML], vs[MB + 1], res[ML];
#pragma unroll kUnrollL
    for (int i = 0; i < ML; ++i)
        if (i < L) vq[i] = dq[(pid * L + i) * N + n];
#pragma unroll kUnrollB
    for (int j = 0; j < MB + 1; ++j)
        if (j < S) vs[j] = ds[(pid * S + j) * N + n];
    if (Z.f64aux) behz_floor_sk_coeff_f64<ML, MB>(Z, primes, L, Z.nB, vq, vs, res);
    else behz_floor_sk_coeff<ML, MB>(Z, primes, L, Z.nB, vq, vs, res);
#pragma unroll kUnrollL
    for (int j = 0; j < ML; ++j)
        if (j < L) out[(pid * L + j) * N + n] = res[j];
}

// Inverse column pass of one residue's column (raw in, canonical out, in registers), by the engine that owns the prime.
template <int LOGN1> __device__ __forceinline__ void col_inv_load(const PrimeDev &P, const u64 *src, u64 v[1 << LOGN1])
{
    constexpr int N1 = 1 << LOGN1;
    if (P.f64) {
        const ArF64 ar = make_ar(P, (ArF64 *)nullptr);
        double y[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) y[a] = ar.from_raw(src[a << kRowLog]);
        col_inv<ArF64, LOGN1>(ar, y, ctw(P.inv), P.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) v[a] = ar.to_canon(y[a]);
    } else {
        const ArU64 ar = make_ar(P, (ArU64 *)nullptr);
        u64 y[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) y[a] = src[a << kRowLog];
        col_inv<ArU64, LOGN1>(ar, y, ctw(P.inv), P.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) v[a] = ar.to_canon(y[a]);
    }
}
// The inverse column pass of all L + S residues AND BEHZ steps (6)-(8) in one kernel (N <= 16384, L <= 4, nB <= 6): the mirror image of
// k_behz_extend_cols -- phase 1, wave w runs the column passes of residues w, w + N1, ... for the block's 64 columns and parks the
// canonical values in LDS [residue][e][c]; phase 2, lane (row e, column c) takes its coefficient's L + S residues through the fast
// floor and the Shenoy-Kumaresan conversion.  The coefficient-form products never exist in HBM.
// EXACT: L == ML and nB == MB (the reference's own parameter sets: {60,40}, {60,40,40}, {60,40,40,40} with as many auxiliary primes):
// every residue loop has a compile-time trip count -- straight-line code, no guards.
template <int LOGN1, int ML, int MB, bool EXACT>
__global__ void __launch_bounds__(64 << LOGN1) k_behz_cols_floor_sk(BehzDev Z, const PrimeDev *primes, const u64 *dq, const u64 *ds, u64 *out)
{
    constexpr int N1 = 1 << LOGN1;
    extern __shared__ u64 behz_sm[]; // [L + S][N1][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 pid = blockIdx.x >> 4; // (op, k)
    const u64 col = ((blockIdx.x & 15) << 6) | lane;
    const int L = EXACT ? ML : Z.L, nB = EXACT ? MB : Z.nB, S = nB + 1;
    const u64 N = (u64)N1 << kRowLog;
    for (int rr = wave; rr < L + S; rr += N1) {
        const bool isq = rr < L;
        const PrimeDev &P = primes[isq ? rr : Z.bsk_prime[rr - L]];
        const u64 *src = (isq ? dq + (pid * L + rr) * N : ds + (pid * S + (rr - L)) * N) + col;
        u64 v[N1];
        col_inv_load<LOGN1>(P, src, v);
#pragma unroll
        for (int e = 0; e < N1; ++e) behz_sm[((rr * N1 + e) << 6) | lane] = v[e];
    }
    __syncthreads();
    u64 vq[ML], vs[MB + 1], res[ML];
#pragma unroll
    for (int i = 0; i < ML; ++i)
        if (i < L) vq[i] = behz_sm[((i * N1 + wave) << 6) | lane];
#pragma unroll
    for (int j = 0; j < MB + 1; ++j)
        if (j < S) vs[j] = behz_sm[(((L + j) * N1 + wave) << 6) | lane];
    if (Z.f64aux) behz_floor_sk_coeff_f64<ML, MB>(Z, primes, L, nB, vq, vs, res);
    else behz_floor_sk_coeff<ML, MB>(Z, primes, L, nB, vq, vs, res);
#pragma unroll
    for (int j = 0; j < ML; ++j)
        if (j < L) out[(pid * L + j) * N + ((u64)wave << kRowLog) + col] = res[j];
}

// coefficient-form automorphism as a gather (2 polys of a size-2 ciphertext)
__global__ void __launch_bounds__(kBlock) k_bfv_galois(const u64 *in, const uint32_t *gather, u64 *c01, u64 c01_item_stride, u64 *tgt,
                                                       const PrimeDev *primes, int L, int logN, u64 n_ops, const u64 *addend)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 o = gid & (((u64)1 << logN) - 1);
    const u64 oi = gid >> logN;
    const u64 op = oi / L;
    if (op >= n_ops) return;
    const int i = (int)(oi % L);
    const u64 N = (u64)1 << logN, P1 = (u64)L * N;
    const u64 q = primes[i].q;
    const uint32_t g = gather[o];
    const u64 src = g & 0x7FFFFFFFu;
    const u64 *p0 = in + op * 2 * P1 + (u64)i * N;
    u64 v0 = p0[src], v1 = p0[P1 + src];
    if (g >> 31) { v0 = v0 ? q - v0 : 0; v1 = v1 ? q - v1 : 0; }
    u64 *o0 = c01 + op * c01_item_stride + (u64)i * N + o;
    u64 w1 = 0;
    if (addend) { // out = addend + rotate(in)
        const u64 *ad = addend + op * 2 * P1 + (u64)i * N + o;
        v0 = addmod(v0, ad[0], q);
        w1 = ad[P1];
    }
    o0[0] = v0;
    o0[P1] = w1;
    tgt[(op * L + i) * N + o] = v1;
}

template <int LOGN1>
__global__ void __launch_bounds__(kBlock) k_bfv_tail_sp(const u64 *tpr, u64 *rp, const PrimeDev *primes, int sp)
{
    constexpr int N1 = 1 << LOGN1;
    constexpr u64 N = (u64)N1 << kRowLog;
    const u64 poly = blockIdx.x >> 2;
    const int col = ((blockIdx.x & 3) << 8) | threadIdx.x;
    const u64 *src = tpr + poly * N;
    const PrimeDev &Ps = primes[sp];
    u64 c[N1];
    if (LOGN1 == 0) {
        c[0] = src[col];
    } else if (Ps.f64) {
        const ArF64 ar = make_ar(Ps, (ArF64 *)nullptr);
        double x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = ar.from_raw(src[(a << kRowLog) + col]);
        col_inv<ArF64, LOGN1>(ar, x, ctw(Ps.inv), Ps.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) c[a] = ar.to_canon(x[a]);
    } else {
        const ArU64 ar = make_ar(Ps, (ArU64 *)nullptr);
        u64 x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = src[(a << kRowLog) + col];
        col_inv<ArU64, LOGN1>(ar, x, ctw(Ps.inv), Ps.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) c[a] = ar.to_canon(x[a]);
    }
    const u64 half = Ps.q >> 1;
#pragma unroll
    for (int a = 0; a < N1; ++a) rp[poly * N + (a << kRowLog) + col] = addmod(c[a], half, Ps.q);
}

// k_bfv_tail_fin is memory-bound and latency-hidden by occupancy alone: at 101 registers five waves fit a SIMD, and the 6144 waves of
// BASELINE configs[4]'s calls (64 ciphertexts x 2 x 3 polynomials) ran as one full round plus a fifth of one.  Held to six waves per SIMD
// (80 registers, 32 spilled) the call is one round: 109 -> 95 us, the matrix product 67.1 -> 65.3 ms.
#ifndef HE355_TAILFIN_WAVES
#define HE355_TAILFIN_WAVES 6
#endif
template <int LOGN1>
__global__ void
#if HE355_TAILFIN_WAVES
__launch_bounds__(kBlock, HE355_TAILFIN_WAVES)
#else
__launch_bounds__(kBlock)
#endif
k_bfv_tail_fin(const u64 *t, const u64 *rp, u64 *c01, u64 c01_item_stride, const u64 *add01, u64 add01_item_stride, const PrimeDev *primes,
                                                         const FloorConst *fcs, int L, int K)
{
    constexpr int N1 = 1 << LOGN1;
    constexpr u64 N = (u64)N1 << kRowLog;
    const u64 pid = blockIdx.x >> 2; // (op, k, i)
    const int col = ((blockIdx.x & 3) << 8) | threadIdx.x;
    const int i = (int)(pid % L);
    const u64 ok = pid / L;
    const int k = (int)(ok & 1);
    const u64 op = ok >> 1;
    const u64 *src = t + pid * N;
    const PrimeDev &Pi = primes[i];
    const PrimeDev &Ps = primes[K - 1];
    u64 c[N1];
    if (LOGN1 == 0) {
        c[0] = src[col];
    } else if (Pi.f64) {
        const ArF64 ar = make_ar(Pi, (ArF64 *)nullptr);
        double x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = ar.from_raw(src[(a << kRowLog) + col]);
        col_inv<ArF64, LOGN1>(ar, x, ctw(Pi.inv), Pi.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) c[a] = ar.to_canon(x[a]);
    } else {
        const ArU64 ar = make_ar(Pi, (ArU64 *)nullptr);
        u64 x[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = src[(a << kRowLog) + col];
        col_inv<ArU64, LOGN1>(ar, x, ctw(Pi.inv), Pi.inv_w0_scaled);
#pragma unroll
        for (int a = 0; a < N1; ++a) c[a] = ar.to_canon(x[a]);
    }
    const FloorConst fc = fcs[(K - 1) * K + i];
    const ModU64 mi = make_modu(Pi);
    const u64 qi = Pi.q, qs = Ps.q;
    u64 *dst = c01 + op * c01_item_stride + ((u64)k * L + i) * N;
    const u64 *add = add01 + op * add01_item_stride + ((u64)k * L + i) * N; // what the key-switched part is added to (may be dst itself)
#pragma unroll
    for (int a = 0; a < N1; ++a) {
        const u64 r = rp[ok * N + (a << kRowLog) + col];
        const u64 delta = submod(qs > qi ? barrett64(r, mi) : r, fc.half_mod, qi);
        const u64 res = mulmod(submod(c[a], delta, qi), fc.inv, mi); // (a prime of either engine: Barrett)
        const u64 idx = (a << kRowLog) + col;
        dst[idx] = addmod(add[idx], res, qi);
    }
}

} // namespace


inline unsigned grid_for(u64 jobs, u64 per_block) { return (unsigned)((jobs + per_block - 1) / per_block); }

// =======================================================================================================
// Launchers
// =======================================================================================================
void launch_ntt_forward(const KernelEnv &env, const PolyView &v, u32 n_items)
{
    const u64 polys = (u64)n_items * v.polys_per_item;
    if (!polys) return;
    if (env.logn1 > 0) {
        const unsigned g = (unsigned)(polys * 4);
        switch (env.logn1) {
        case 1: hipLaunchKernelGGL(k_cols_fwd<1>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        case 2: hipLaunchKernelGGL(k_cols_fwd<2>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        case 3: hipLaunchKernelGGL(k_cols_fwd<3>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        case 4: hipLaunchKernelGGL(k_cols_fwd<4>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        case 5: hipLaunchKernelGGL(k_cols_fwd<5>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        }
    }
    const u64 jobs = polys << env.logn1;
    hipLaunchKernelGGL(k_rows_fwd, dim3(grid_for(jobs, kWaves)), dim3(kBlock), 0, env.stream, v, env.primes, jobs, env.logn1, env.logn1 > 0 ? 1 : 0);
}

void launch_ntt_inverse(const KernelEnv &env, const PolyView &v, u32 n_items)
{
    const u64 polys = (u64)n_items * v.polys_per_item;
    if (!polys) return;
    const u64 jobs = polys << env.logn1;
    hipLaunchKernelGGL(k_rows_inv, dim3(grid_for(jobs, kWaves)), dim3(kBlock), 0, env.stream, v, env.primes, jobs, env.logn1);
    if (env.logn1 > 0) {
        const unsigned g = (unsigned)(polys * 4);
        switch (env.logn1) {
        case 1: hipLaunchKernelGGL(k_cols_inv<1>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        case 2: hipLaunchKernelGGL(k_cols_inv<2>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        case 3: hipLaunchKernelGGL(k_cols_inv<3>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        case 4: hipLaunchKernelGGL(k_cols_inv<4>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        case 5: hipLaunchKernelGGL(k_cols_inv<5>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
        }
    }
}

void launch_addsub(const KernelEnv &env, int L, int size, u64 n_results, const u64 *a, const u64 *b, Indexer ix, u64 *out, bool sub)
{
    if (!n_results) return;
    const int logN = env.logn1 + kRowLog, polys = size * L;
    const u64 threads = (n_results * polys) << (logN - 1);
    hipLaunchKernelGGL(k_addsub, dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, a, b, out, ix, env.primes, L, polys, logN, n_results,
                       sub ? 1 : 0);
}

void launch_plain_op(const KernelEnv &env, int L, int size, u64 n_results, const u64 *ct, const u64 *pt, Indexer ix, u64 *out, int mode)
{
    if (!n_results) return;
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (n_results * size * L) << (logN - 1);
    hipLaunchKernelGGL(k_plain_op, dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, ct, pt, out, ix, env.primes, L, size, logN, n_results, mode);
}
void launch_drop_residues(const KernelEnv &env, int L, int L_to, u64 n_polys, const u64 *in, u64 *out)
{
    if (!n_polys) return;
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (n_polys * L_to) << (logN - 1);
    hipLaunchKernelGGL(k_drop_residues, dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, in, out, L, L_to, logN, n_polys);
}
void launch_sum_cts(const KernelEnv &env, int L, int size, u64 n_terms, const u64 *in, u64 *out, u64 n_out, bool accumulate)
{
    const int logN = env.logn1 + kRowLog;
    const u64 threads = ((u64)size * L) << (logN - 1);
    for (u64 c0 = 0; c0 < n_out; c0 += 65535) { // gridDim.y holds 65535 results
        const unsigned ny = (unsigned)std::min<u64>(65535, n_out - c0);
        hipLaunchKernelGGL(k_sum_cts, dim3(grid_for(threads, kBlock), ny), dim3(kBlock), 0, env.stream, in, out, env.primes, L, size * L, logN, n_terms,
                           accumulate ? 1 : 0, n_out, c0);
    }
}

void launch_sum_groups(const KernelEnv &env, int L, u64 n_cts, u32 n_groups, const u64 *in, const u32 *d_mult, u64 *out)
{
    if (!n_cts || !n_groups) return;
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (n_cts * 2 * L) << (logN - 1);
    hipLaunchKernelGGL(k_sum_groups, dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, in, out, d_mult, env.primes, L, 2 * L, logN, n_cts, n_groups);
}

void launch_move_cts(const KernelEnv &env, u64 *dst, const u64 *src, const uint32_t *idx, u64 n, u64 elems_per_ct, bool scatter)
{
    const u64 pairs = elems_per_ct / 2;
    for (u64 off = 0; off < n; off += kMoveListCap) {
        const u64 m = std::min<u64>(kMoveListCap, n - off);
        MoveList list;
        for (u64 g = 0; g < m; ++g) list.idx[g] = idx[off + g];
        // gather: compact rows [off, off+m) of dst; scatter: compact rows [off, off+m) of src
        hipLaunchKernelGGL(k_move_cts, dim3(grid_for(pairs, kBlock), (unsigned)m), dim3(kBlock), 0, env.stream, scatter ? dst : dst + off * elems_per_ct,
                           scatter ? src + off * elems_per_ct : src, list, pairs, scatter ? 1 : 0);
    }
}

void launch_mul3_acc(const KernelEnv &env, int L, u64 rows, u64 cols, int inner, const u64 *a, u64 a_stride_i, u64 a_stride_k, const u64 *b,
                     u64 b_stride_k, u64 b_stride_j, u64 *out)
{
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (rows * cols * (u64)L) << (logN - 1);
    hipLaunchKernelGGL(k_mul3_acc, dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, a, b, out, env.primes, L, logN, rows, cols, inner,
                       a_stride_i, a_stride_k, b_stride_k, b_stride_j);
}

void launch_mul3(const KernelEnv &env, int L, u64 n_results, const u64 *a, const u64 *b, Indexer ix, u64 *out)
{
    if (!n_results) return;
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (n_results * L) << (logN - 1);
    hipLaunchKernelGGL(k_mul3, dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, a, b, out, ix, env.primes, L, logN, n_results);
}

// Latency shape: the two engines' launches of a stage as one kernel (k_k1_dual, k_k2n_dual, k_k3_dual, k_floor_rows_dual); HE355_DUAL_ENGINE=0: one
// launch per engine as in the throughput shape.
static bool dual_engine_launches()
{
    static const bool off = getenv("HE355_DUAL_ENGINE") && getenv("HE355_DUAL_ENGINE")[0] == '0';
    return !off;
}
// ... also for small grids of the throughput shape: up to this many blocks for both engines together (profiles/r04_dual_engine_latency.txt)
// (round 5: per kernel.  k_k3's shared launch pays up to a few thousand blocks since the fold form of the u64 engine brought its blocks
// close to the fp64 engine's in length; the others -- whose dual kernels are the latency shape's, behind a function boundary -- keep
// the 1024 they were swept at.  profiles/r05_dual_threshold_sweep.txt)
static unsigned dual_max_blocks_k3() { return k3_fuse_policy() == 2 ? 0u : 4096u; }
static unsigned dual_max_blocks() { return k3_fuse_policy() == 2 ? 0u : 1024u; }
void launch_k1(const KernelEnv &env, int L, K1Mode mode, u64 n_ops, u64 op_offset, const u64 *a, const u64 *b, Indexer ix, const uint32_t *perm,
               const KsBuffers &buf, const u64 *addend, bool no_c01, bool no_c1, const KsGroups *groups, bool no_c0n)
{
    if (!n_ops) return;
    if (groups && (mode != K1_GALOIS || addend)) throw std::runtime_error("grouped launch: rotations without addend only");
    if (no_c01 && mode != K1_MUL) throw std::runtime_error("no_c01: the ct x ct multiply only");
    if (no_c1 && mode != K1_GALOIS && mode != K1_CT3) throw std::runtime_error("no_c1: rotations and size-3 inputs only");
    K1Args A;
    A.no_c1 = no_c1 ? 1 : 0;
    if (no_c0n && (mode != K1_GALOIS || !no_c1 || (addend && groups))) throw std::runtime_error("no_c0n: rotations whose polynomial 1 is left to k_k3 only");
    A.no_c0n = no_c0n ? 1 : 0;
    A.groups = groups ? *groups : KsGroups{};
    A.a = a; A.b = b; A.ix = ix; A.perm = perm; A.addend = addend;
    A.c01 = buf.c01; A.c01_item_stride = buf.c01_item_stride; A.c2n = buf.c2n; A.c2r = buf.c2r;
    A.n_ops = n_ops; A.op_offset = op_offset; A.L = L; A.logn1 = env.logn1; A.mode = (int)mode;
    K1Args AP[2];
    unsigned gp[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) { // pass 0: fp64-engine residues, pass 1: u64-engine residues
        A.n_i = 0;
        for (int i = 0; i < L; ++i)
            if ((env.prime_f64[i] != 0) == (pass == 0)) A.i_list[A.n_i++] = (unsigned char)i;
        AP[pass] = A;
        gp[pass] = A.n_i ? grid_for((n_ops * A.n_i) << env.logn1, kWaves) : 0;
    }
    // (k_k1 proper -- the throughput shape -- maps a block onto one (residue, row) tile and four consecutive ops: its own grid below)
    const hipStream_t st = env.stream;
    if (gp[0] && gp[1] && (n_ops <= 8 || gp[0] + gp[1] <= dual_max_blocks()) && mode != K1_MUL_C2 && !(mode == K1_MUL && no_c01) &&
        dual_engine_launches()) { // latency shape / small grids: one launch for both engines
        const dim3 grid(gp[0] + gp[1]);
        if (mode == K1_MUL) hipLaunchKernelGGL((k_k1_dual<K1_MUL>), grid, dim3(kBlock), 0, st, AP[0], AP[1], gp[0], env.primes);
        else if (mode == K1_CT3) hipLaunchKernelGGL((k_k1_dual<K1_CT3>), grid, dim3(kBlock), 0, st, AP[0], AP[1], gp[0], env.primes);
        else hipLaunchKernelGGL((k_k1_dual<K1_GALOIS>), grid, dim3(kBlock), 0, st, AP[0], AP[1], gp[0], env.primes);
        return;
    }
    for (int pass = 0; pass < 2; ++pass) {
        if (!gp[pass]) continue;
        const K1Args &AA = AP[pass];
        const dim3 grid((unsigned)((((u64)AA.n_i) << env.logn1) * ((n_ops + kWaves - 1) / kWaves)));
        if (pass == 0) {
            if (mode == K1_MUL && no_c01) hipLaunchKernelGGL((k_k1<K1_MUL_C2, ArF64>), grid, dim3(kBlock), 0, st, AA, env.primes);
            else if (mode == K1_MUL) hipLaunchKernelGGL((k_k1<K1_MUL, ArF64>), grid, dim3(kBlock), 0, st, AA, env.primes);
            else if (mode == K1_CT3) hipLaunchKernelGGL((k_k1<K1_CT3, ArF64>), grid, dim3(kBlock), 0, st, AA, env.primes);
            else hipLaunchKernelGGL((k_k1<K1_GALOIS, ArF64>), grid, dim3(kBlock), 0, st, AA, env.primes);
        } else {
            if (mode == K1_MUL && no_c01) hipLaunchKernelGGL((k_k1<K1_MUL_C2, ArU64>), grid, dim3(kBlock), 0, st, AA, env.primes);
            else if (mode == K1_MUL) hipLaunchKernelGGL((k_k1<K1_MUL, ArU64>), grid, dim3(kBlock), 0, st, AA, env.primes);
            else if (mode == K1_CT3) hipLaunchKernelGGL((k_k1<K1_CT3, ArU64>), grid, dim3(kBlock), 0, st, AA, env.primes);
            else hipLaunchKernelGGL((k_k1<K1_GALOIS, ArU64>), grid, dim3(kBlock), 0, st, AA, env.primes);
        }
    }
}

void launch_k2(const KernelEnv &env, int L, u64 n_ops, const KsBuffers &buf, const u64 *src, u64 src_op_stride, int tsplit)
{
    if (!n_ops) return;
    K2Args A;
    A.c2r = src ? src : buf.c2r;
    A.src_op_stride = src ? src_op_stride : (u64)L * env.N;
    A.src_is_coeff = src ? 1 : 0;
    A.d = buf.d; A.n_ops = n_ops; A.L = L; A.K = env.K; A.ckks = env.scheme == 2;
    A.f64_mask = 0; A.tsplit = tsplit > 1 ? tsplit : 1;
    for (int t = 0; t < env.K; ++t) A.f64_mask |= (u64)(env.prime_f64[t] != 0) << t;
    // Small grids: the targets of a (digit, column block) dealt to 2 or 4 blocks (blockIdx.y, as the latency shape does) when
    // that fills the rounds of blocks better -- 512 blocks run at a time (two per CU); a block pays the digit's inverse column
    // pass once and a forward column pass + stores per target.  64 BFV ciphertexts at L = 3: 768 blocks = 1.5 rounds of all four
    // targets, or 3 full rounds of two targets each.  Only taken for a modelled gain of 5 % or more (the headline's 61440 blocks
    // stay whole).
    auto target_split = [&](unsigned gw) {
        if (tsplit > 1) return tsplit;
        int ts = 1;
        const int n_tgt = L + 1 - (A.ckks ? 1 : 0);
        auto cost = [&](int c) { return (double)(((u64)gw * c + 511) / 512) * (1.0 + 1.2 * ((n_tgt + c - 1) / c)); };
        const double c1 = cost(1);
        double best = c1;
        for (int c = 2; c <= 4; c <<= 1)
            if (cost(c) < best * 0.95 && cost(c) < c1 * 0.95) { best = cost(c); ts = c; }
        return ts;
    };
    K2Args AK[2];
    unsigned gk[2] = {0, 0};
    for (int wide = 0; wide < 2; ++wide) { // digits below 2^52, then the 60-bit ones: one instantiation each
        A.n_dig = 0;
        for (int j = 0; j < L; ++j)
            if ((env.prime_q[j] >> 52 != 0) == (wide != 0)) A.dig_list[A.n_dig++] = (unsigned char)j;
        AK[wide] = A;
        gk[wide] = (unsigned)(n_ops * A.n_dig * 4);
    }
    if (gk[0] && gk[1] && dual_engine_launches()) { // latency shape or a small grid: both digit kinds in one launch (k_k2n_dual)
        const int ts = target_split(gk[0] + gk[1]);
        if (tsplit > 1 || (gk[0] + gk[1]) * (unsigned)ts <= dual_max_blocks()) {
            AK[0].tsplit = AK[1].tsplit = ts;
            const dim3 gd(gk[0] + gk[1], (unsigned)ts);
#define HE355_K2D(L1) case L1: hipLaunchKernelGGL((k_k2n_dual<L1>), gd, dim3(kBlock), 0, env.stream, AK[0], AK[1], gk[0], env.primes); break;
            switch (env.logn1) { HE355_K2D(0) HE355_K2D(1) HE355_K2D(2) HE355_K2D(3) HE355_K2D(4) HE355_K2D(5) }
#undef HE355_K2D
            return;
        }
    }
    for (int wide = 0; wide < 2; ++wide) {
        if (!gk[wide]) continue;
        AK[wide].tsplit = target_split(gk[wide]);
        const dim3 gd(gk[wide], (unsigned)AK[wide].tsplit);
#define HE355_K2N(L1)                                                                                                         \
    case L1:                                                                                                                  \
        if (wide) hipLaunchKernelGGL((k_k2n<L1, true>), gd, dim3(kBlock), 0, env.stream, AK[wide], env.primes);               \
        else hipLaunchKernelGGL((k_k2n<L1, false>), gd, dim3(kBlock), 0, env.stream, AK[wide], env.primes);                   \
        break;
        switch (env.logn1) { HE355_K2N(0) HE355_K2N(1) HE355_K2N(2) HE355_K2N(3) HE355_K2N(4) HE355_K2N(5) }
#undef HE355_K2N
    }
}


void launch_k3(const KernelEnv &env, int L, u64 n_ops, const KsBuffers &buf, const u64 *key, K3Part part, const K3Fuse *fuse, int n_split, u64 *split_part,
               int n_split_u64, const KsGroups *groups, u64 g_op_offset)
{
    if (groups && (n_split > 1 || (fuse && fuse->ta))) throw std::runtime_error("grouped keys: the 8-wave rotation instantiations only");
    if (n_split_u64 <= 0) n_split_u64 = n_split;
    if (n_split > 1 && (fuse || !split_part)) throw std::runtime_error("digit-split K3: unfused launches with a partial-sum buffer only");
    const unsigned char *prime_f64 = env.prime_f64;
    if (!n_ops) return;
    if (fuse && (part != K3_DATA_ONLY || !k3_can_fuse(env))) throw std::runtime_error("fused mod-down: data-prime tiles only");
    struct { K3Args A; unsigned g; bool f64; int waves; } pend[2];
    int n_pend = 0;
    for (int pass = 0; pass < 2; ++pass) { // pass 0: fp64-engine primes, pass 1: u64-engine primes
        K3Args A;
        A.d = buf.d; A.c2n = buf.c2n; A.key = key; A.t = buf.t; A.tp = buf.tp; A.tpr = buf.tpr;
        A.n_ops = n_ops; A.L = L; A.K = env.K; A.logn1 = env.logn1; A.ckks = env.scheme == 2;
        A.cols = fuse ? fuse->cols : nullptr; A.c01 = fuse ? fuse->c01 : nullptr; A.c01_item_stride = fuse ? fuse->c01_item_stride : 0;
        A.cols2 = fuse ? fuse->cols2 : nullptr; A.out2 = fuse ? fuse->out : nullptr;
        A.c1_mode = fuse ? fuse->c1_mode : 0; A.c1_src = fuse ? fuse->c1_src : nullptr;
        A.gsrc = fuse ? fuse->gsrc : nullptr; A.gperm = fuse ? fuse->gperm : nullptr; A.gsrc_op_offset = fuse ? fuse->gsrc_op_offset : 0;
        if ((A.c1_mode == 4 || A.c1_mode == 5) && (!A.gsrc || (!groups && !A.gperm))) throw std::runtime_error("gathered rotation: the input slab and its permutation are needed");
        if (A.c1_mode == 5 && !A.c1_src) throw std::runtime_error("gathered rotation with addend: the addend rows are needed");
        A.ta = fuse ? fuse->ta : nullptr; A.tb = fuse ? fuse->tb : nullptr; A.tix = fuse ? fuse->tix : Indexer{}; A.t_op_offset = fuse ? fuse->t_op_offset : 0;
        if (fuse && fuse->cols2 && fuse->tt_hi > L - 1) throw std::runtime_error("fused rescale: only primes below the one divided out");
        A.fc = env.floor_consts;
        A.n_split = n_split > 1 ? (pass == 0 ? n_split : n_split_u64) : 1; A.part = split_part;
        A.n_tt = 0;
        A.n_q = 0;
        for (int t = 0; t < env.K; ++t) A.n_q += prime_f64[t] == 0;
        A.keyq_offset = (u64)env.Ltop * 2 * env.K * env.N; // the quotient array follows the key in the same allocation
        A.keyq = key ? key + A.keyq_offset : nullptr;
        A.groups = groups ? *groups : KsGroups{}; A.g_op_offset = g_op_offset;
        for (int tt = 0; tt <= L; ++tt) {
            if ((part == K3_SPECIAL_ONLY && tt != L) || (part == K3_DATA_ONLY && tt == L)) continue;
            if (fuse && (tt < fuse->tt_lo || tt >= fuse->tt_hi)) continue;
            const int t = (tt == L) ? env.K - 1 : tt;
            if ((prime_f64[t] != 0) == (pass == 0)) {
                int slot = 0;
                for (int u = 0; u < t; ++u) slot += prime_f64[u] == 0;
                A.q_slot[A.n_tt] = (unsigned char)slot;
                A.tt_list[A.n_tt++] = (unsigned char)tt;
            }
        }
        if (!A.n_tt) continue;
        int waves = A.n_split > 1 ? 1 : 8; // latency shape: one wave per block, one (tile, op, digit group) each; else the 8-wave shape
        const u64 tiles = (u64)A.n_tt << env.logn1;
        // enough blocks to keep every CU busy for several rounds, few enough that start-up costs are amortised
        // Op-groups per block: more of them amortise the block's start-up (twiddle staging, ~4 us) over more work, fewer of them give the
        // dispatcher more blocks to fill the 256 CUs with.  What counts for small grids is the number of ROUNDS of blocks: 64 tiles x 8
        // op-groups (BFV, 64 ciphertexts, L = 3) are two rounds of 256 one-group blocks or ONE round of two-group blocks, the same work
        // with half the start-ups (configs[4]: 65.3 -> 63.4 ms).  So: the candidate sizes from the shape's maximum down, each priced as
        // rounds x (groups x work per group + start-up), ties to the larger.  (Maximum for the 8-wave shape: 4 once a tile has 128
        // op-groups -- chunks of 1024 ciphertexts: 50.3 vs 50.9 ms per step with 2 -- and 2 below that, where 4 measured slower;
        // profiles/r03_chunk_sweep.txt.)
        u64 n_og = 0;
        u32 ogpb = 1;
        unsigned g = 0;
        auto size_grid = [&](int w) {
            n_og = (n_ops + w - 1) / w;
            const u32 og_max = w == 8 ? (n_og >= 128 ? 4 : 2) : 4;
            ogpb = og_max;
            {
                const double work = (pass == 0 ? 7.0 : 14.0) * (L + 2) * (w == 4 ? 0.5 : 1.0), startup = 4.0; // us per op-group (one row step per digit + epilogue), per block
                double best = 0;
                for (u32 c = og_max; c >= 1; c >>= 1) {
                    const u64 blocks = tiles * ((n_og + c - 1) / c);
                    const double cost = (double)((blocks + 255) / 256) * (c * work + startup);
                    if (c == og_max || cost < best * 0.999) { best = cost; ogpb = c; }
                }
            }
            const u64 n_ogb = (n_og + ogpb - 1) / ogpb;
            g = (unsigned)(((tiles + 7) / 8) * 8 * n_ogb);
        };
        size_grid(waves);
        const bool level_sum = groups && groups->sum_out && fuse; // (the special prime's launch ahead of it writes no ciphertext rows)
        if (level_sum) {
            // one block per (tile, eight ciphertexts), walking this launch's groups in turn (see KsGroups::sum_out)
            const u64 gs8 = groups->group_size / 8;
            if (waves != 8 || groups->group_size % 8 || n_ops % groups->group_size || g_op_offset % groups->group_size)
                throw std::runtime_error("level sum in k_k3: fused 8-wave launches over whole groups of a multiple of eight ciphertexts");
            ogpb = (u32)(n_og / gs8);
            g = (unsigned)(((tiles + 7) / 8) * 8 * gs8);
        }
        // u64-engine tiles of a small grid (at most half the CUs busy with 8-wave blocks): FOUR waves per block -- one per SIMD, each at
        // the full issue rate instead of half of it, and twice the blocks; the serial digit loop of a tile is what such a launch lasts
        const unsigned four_max = k3_fuse_policy() == 2 ? 0u : 128u;
        if (pass == 1 && waves == 8 && g <= four_max && !level_sum) {
            size_grid(4);
            // (a CU holds ONE block of either shape -- the LDS arrays -- so the four-wave blocks must still fit one round together with the
            // fp64-engine blocks they may share the launch with)
            if (g + (n_pend ? pend[0].g : 0) <= 256) waves = 4;
            else size_grid(8);
        }
        A.og_per_block = ogpb;
        A.og_stride = level_sum ? (u32)(groups->group_size / 8) : 1u;
        pend[n_pend].A = A; pend[n_pend].g = g; pend[n_pend].f64 = pass == 0; pend[n_pend].waves = waves;
        ++n_pend;
    }
    const hipStream_t st3 = env.stream;
    const bool tensor = fuse && fuse->ta;
    if (n_pend == 2 && pend[0].waves == 1 && dual_engine_launches()) { // latency shape
        const unsigned ny = (unsigned)std::max(pend[0].A.n_split, pend[1].A.n_split);
        hipLaunchKernelGGL(k_k3_dual, dim3(pend[0].g + pend[1].g, ny), dim3(64), 0, st3, pend[0].A, pend[1].A, pend[1].g, env.primes);
        return;
    }
    // throughput shape, small grids (up to two blocks per CU for both engines together): both engines in one launch
    if (n_pend == 2 && pend[0].waves == 8 && pend[0].g + pend[1].g <= dual_max_blocks_k3() && dual_engine_launches()) {
        const dim3 gd(pend[0].g + pend[1].g);
#define HE355_K3D8(F, T, G)                                                                                                                     \
    do {                                                                                                                                        \
        if (pend[1].waves == 4) hipLaunchKernelGGL((k_k3_dual8<F, T, G, 4>), gd, dim3(512), 0, st3, pend[0].A, pend[1].A, pend[1].g, env.primes); \
        else hipLaunchKernelGGL((k_k3_dual8<F, T, G, 8>), gd, dim3(512), 0, st3, pend[0].A, pend[1].A, pend[1].g, env.primes);                    \
    } while (0)
        if (groups) { if (fuse) HE355_K3D8(true, false, true); else HE355_K3D8(false, false, true); }
        else if (tensor) HE355_K3D8(true, true, false);
        else if (fuse) HE355_K3D8(true, false, false);
        else HE355_K3D8(false, false, false);
#undef HE355_K3D8
        return;
    }
    for (int i = 0; i < n_pend; ++i) {
        const K3Args &A = pend[i].A;
        const unsigned g = pend[i].g;
        const bool f64 = pend[i].f64;
        KernelProbe *pr = f64 && pend[i].waves == 8 ? env.probe : nullptr;
        int slot = -1;
        if (pr && pr->used < KernelProbe::kCap) {
            slot = pr->used++;
            if (slot >= pr->created) {
                (void)hipEventCreate(&pr->start[slot]);
                (void)hipEventCreate(&pr->stop[slot]);
                pr->created = slot + 1;
            }
            pr->ops += n_ops;
            (void)hipEventRecord(pr->start[slot], env.stream);
        }
        if (pend[i].waves == 1) {
            const dim3 gd(g, (unsigned)A.n_split);
            if (f64) hipLaunchKernelGGL((k_k3<ArF64, 1>), gd, dim3(64), 0, st3, A, env.primes);
            else hipLaunchKernelGGL((k_k3<ArU64, 1>), gd, dim3(64), 0, st3, A, env.primes);
        } else if (pend[i].waves == 4) { // (u64 engine, small grid)
            if (groups && fuse) hipLaunchKernelGGL((k_k3<ArU64, 4, true, false, true>), dim3(g), dim3(256), 0, st3, A, env.primes);
            else if (groups) hipLaunchKernelGGL((k_k3<ArU64, 4, false, false, true>), dim3(g), dim3(256), 0, st3, A, env.primes);
            else if (tensor) hipLaunchKernelGGL((k_k3<ArU64, 4, true, true>), dim3(g), dim3(256), 0, st3, A, env.primes);
            else if (fuse) hipLaunchKernelGGL((k_k3<ArU64, 4, true>), dim3(g), dim3(256), 0, st3, A, env.primes);
            else hipLaunchKernelGGL((k_k3<ArU64, 4>), dim3(g), dim3(256), 0, st3, A, env.primes);
        } else if (groups) {
            if (f64 && fuse) hipLaunchKernelGGL((k_k3<ArF64, 8, true, false, true>), dim3(g), dim3(512), 0, st3, A, env.primes);
            else if (f64) hipLaunchKernelGGL((k_k3<ArF64, 8, false, false, true>), dim3(g), dim3(512), 0, st3, A, env.primes);
            else if (fuse) hipLaunchKernelGGL((k_k3<ArU64, 8, true, false, true>), dim3(g), dim3(512), 0, st3, A, env.primes);
            else hipLaunchKernelGGL((k_k3<ArU64, 8, false, false, true>), dim3(g), dim3(512), 0, st3, A, env.primes);
        } else if (f64) {
            if (tensor) hipLaunchKernelGGL((k_k3<ArF64, 8, true, true>), dim3(g), dim3(512), 0, st3, A, env.primes);
            else if (fuse) hipLaunchKernelGGL((k_k3<ArF64, 8, true>), dim3(g), dim3(512), 0, st3, A, env.primes);
            else hipLaunchKernelGGL((k_k3<ArF64, 8>), dim3(g), dim3(512), 0, st3, A, env.primes);
        } else {
            if (tensor) hipLaunchKernelGGL((k_k3<ArU64, 8, true, true>), dim3(g), dim3(512), 0, st3, A, env.primes);
            else if (fuse) hipLaunchKernelGGL((k_k3<ArU64, 8, true>), dim3(g), dim3(512), 0, st3, A, env.primes);
            else hipLaunchKernelGGL((k_k3<ArU64, 8>), dim3(g), dim3(512), 0, st3, A, env.primes);
        }
        if (slot >= 0) (void)hipEventRecord(pr->stop[slot], env.stream);
    }
    // (the inverse row pass of the special-prime sums, and of every prime's sums for BFV, happened in the kernel's epilogue)
}

void launch_k3_combine(const KernelEnv &env, int L, u64 n_ops, const KsBuffers &buf, int n_split, const u64 *split_part, int n_split_u64)
{
    if (n_split_u64 <= 0) n_split_u64 = n_split;
    if (!n_ops) return;
    K3CombineArgs A;
    A.part = split_part; A.t = buf.t; A.tpr = buf.tpr; A.n_ops = n_ops; A.n_split = n_split; A.n_split_u64 = n_split_u64; A.L = L; A.K = env.K; A.logn1 = env.logn1;
    A.ckks = env.scheme == 2;
    const u64 jobs = (n_ops * 2 * (u64)(L + 1)) << env.logn1;
    hipLaunchKernelGGL(k_k3_combine, dim3(grid_for(jobs, kWaves)), dim3(kBlock), 0, env.stream, A, env.primes);
}

void launch_floor_cols(const KernelEnv &env, int src_prime, int n_tgt, u64 n_polys, const u64 *src, u64 *dst, int tgt_first, int dst_ntgt, const u64 *src2,
                       int src2_prime, int tsplit)
{
    if (!n_polys || n_tgt <= 0) return;
    FloorColsArgs A;
    A.src = src; A.dst = dst; A.src_prime = src_prime; A.n_tgt = n_tgt; A.K = env.K; A.fc = env.floor_consts;
    A.tgt_first = tgt_first; A.dst_ntgt = dst_ntgt > 0 ? dst_ntgt : n_tgt;
    A.src2 = src2; A.src2_prime = src2_prime;
    A.tsplit = tsplit > 1 ? tsplit : 1;
    const dim3 g((unsigned)(n_polys * 4), (unsigned)A.tsplit);
    A.f64_mask = 0;
    for (int t = 0; t < env.K; ++t) A.f64_mask |= (u64)(env.prime_f64[t] != 0) << t;
#define HE355_FCN(L1)                                                                                                          \
    case L1:                                                                                                                   \
        if (src2) hipLaunchKernelGGL((k_floor_colsn<L1, true>), g, dim3(kBlock), 0, env.stream, A, env.primes);                \
        else hipLaunchKernelGGL((k_floor_colsn<L1, false>), g, dim3(kBlock), 0, env.stream, A, env.primes);                    \
        break;
    switch (env.logn1) { HE355_FCN(0) HE355_FCN(1) HE355_FCN(2) HE355_FCN(3) HE355_FCN(4) HE355_FCN(5) }
#undef HE355_FCN
}

void launch_floor_rows(const KernelEnv &env, u64 n_ops, const FloorRowsArgs &args)
{
    const unsigned char *prime_f64 = env.prime_f64;
    if (!n_ops) return;
    const u64 n_jobs = n_ops * args.n_src;
    // per block: up to 8 jobs per wave, fewer when that would leave CUs without blocks
    u32 jpb = 8 * kWaves;
    while (jpb > (u32)kWaves && (((u64)args.n_tgt << env.logn1) * ((n_jobs + jpb - 1) / jpb) < 256u * 8 || jpb / 2 >= n_jobs)) jpb >>= 1;
    if (args.tail_prime < 0 && dual_engine_launches()) { // no tail prime; latency shape or small grids: both engines in one launch
        FloorRowsDev AE[2];
        unsigned ge[2] = {0, 0};
        for (int e = 0; e < 2; ++e) {
            FloorRowsDev &A = AE[e];
            A.a = args; A.fc = env.floor_consts; A.n_ops = n_ops; A.K = env.K; A.logn1 = env.logn1;
            A.jobs_per_block = jpb;
            A.n_i = 0;
            for (int i = 0; i < args.n_tgt; ++i)
                if ((prime_f64[i] != 0) == (e == 0)) A.i_list[A.n_i++] = (unsigned char)i;
            ge[e] = (unsigned)((((u64)A.n_i) << env.logn1) * ((n_jobs + jpb - 1) / jpb));
        }
        if (ge[0] && ge[1] && (n_ops <= 8 || ge[0] + ge[1] <= dual_max_blocks())) {
            hipLaunchKernelGGL(k_floor_rows_dual, dim3(ge[0] + ge[1]), dim3(kBlock), 0, env.stream, AE[0], AE[1], ge[0], env.primes);
            return;
        }
    }
    for (int pass = 0; pass < 4; ++pass) { // (engine, tail) combinations; the tail prime gets its own launch
        const bool f64 = pass < 2, tail = pass & 1;
        FloorRowsDev A;
        A.a = args; A.fc = env.floor_consts; A.n_ops = n_ops; A.K = env.K; A.logn1 = env.logn1;
        A.jobs_per_block = jpb;
        A.n_i = 0;
        for (int i = 0; i < args.n_tgt; ++i)
            if ((prime_f64[i] != 0) == f64 && (i == args.tail_prime) == tail) A.i_list[A.n_i++] = (unsigned char)i;
        if (!A.n_i) continue;
        const unsigned g = (unsigned)((((u64)A.n_i) << env.logn1) * ((n_jobs + jpb - 1) / jpb));
        const hipStream_t st = env.stream;
        if (f64 && !tail) hipLaunchKernelGGL((k_floor_rows<ArF64, false>), dim3(g), dim3(kBlock), 0, st, A, env.primes);
        else if (f64) hipLaunchKernelGGL((k_floor_rows<ArF64, true>), dim3(g), dim3(kBlock), 0, st, A, env.primes);
        else if (!tail) hipLaunchKernelGGL((k_floor_rows<ArU64, false>), dim3(g), dim3(kBlock), 0, st, A, env.primes);
        else hipLaunchKernelGGL((k_floor_rows<ArU64, true>), dim3(g), dim3(kBlock), 0, st, A, env.primes);
    }
}

void launch_rows_inv_select(const KernelEnv &env, int prime, u64 n_polys, const u64 *src, u64 src_poly_stride, u64 *tail)
{
    if (!n_polys) return;
    const u64 jobs = n_polys << env.logn1;
    hipLaunchKernelGGL(k_rows_inv_select, dim3(grid_for(jobs, kWaves)), dim3(kBlock), 0, env.stream, src, src_poly_stride, tail, env.primes, prime, jobs,
                       env.logn1);
}

void launch_behz_extend(const KernelEnv &env, const BehzDev &bz, const BehzSrc &src, u64 n_cts, u64 *xq, u64 *xbsk)
{
    if (!n_cts) return;
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (n_cts * 2) << logN;
    if (bz.L <= 4 && bz.nB <= 6)
        hipLaunchKernelGGL((k_behz_extend<4, 6>), dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, bz, env.primes, src, xq, xbsk, n_cts * 2, logN);
    else
        hipLaunchKernelGGL((k_behz_extend<kBehzMaxL, kBehzMaxB>), dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, bz, env.primes, src, xq, xbsk, n_cts * 2, logN);
}
bool behz_cols_fusable(const KernelEnv &env, const BehzDev &bz)
{
    return (behz_fuse_mask() & 1) && bz.L <= 4 && bz.nB <= 6 && env.logn1 >= 1 && env.logn1 <= 4;
}
void launch_behz_extend_cols(const KernelEnv &env, const BehzDev &bz, const BehzSrc &src, u64 n_cts, u64 *xq, u64 *xbsk)
{
    if (!n_cts) return;
    if (!behz_cols_fusable(env, bz)) throw std::logic_error("launch_behz_extend_cols: shape not covered by the fused kernel");
    const dim3 g((unsigned)(n_cts * 2 * 16)), blk(64u << env.logn1);
    const size_t lds = (size_t)(bz.L + bz.nB + 1) * (64u << env.logn1) * 8;
#define HE355_EXT(LOGN1, ML, MB, EXACT) hipLaunchKernelGGL((k_behz_extend_cols<LOGN1, ML, MB, EXACT>), g, blk, lds, env.stream, bz, env.primes, src, xq, xbsk)
    switch (env.logn1) {
    case 1: HE355_EXT(1, 4, 6, false); break;
    case 2: HE355_EXT(2, 4, 6, false); break;
    case 4: HE355_EXT(4, 4, 6, false); break; // N = 16384: 1024 threads, 8 KiB of LDS per residue
    default:
        if (bz.L == 2 && bz.nB == 2) HE355_EXT(3, 2, 2, true);
        else if (bz.L == 3 && bz.nB == 3) HE355_EXT(3, 3, 3, true);
        else if (bz.L == 4 && bz.nB == 4) HE355_EXT(3, 4, 4, true);
        else HE355_EXT(3, 4, 6, false);
        break;
    }
#undef HE355_EXT
}
void launch_behz_cols_floor_sk(const KernelEnv &env, const BehzDev &bz, u64 n_ops, const u64 *dq, const u64 *ds, u64 *out)
{
    if (!n_ops) return;
    if (!behz_cols_fusable(env, bz)) throw std::logic_error("launch_behz_cols_floor_sk: shape not covered by the fused kernel");
    const dim3 g((unsigned)(n_ops * 3 * 16)), blk(64u << env.logn1);
    const size_t lds = (size_t)(bz.L + bz.nB + 1) * (64u << env.logn1) * 8;
#define HE355_FLR(LOGN1, ML, MB, EXACT) hipLaunchKernelGGL((k_behz_cols_floor_sk<LOGN1, ML, MB, EXACT>), g, blk, lds, env.stream, bz, env.primes, dq, ds, out)
    switch (env.logn1) {
    case 1: HE355_FLR(1, 4, 6, false); break;
    case 2: HE355_FLR(2, 4, 6, false); break;
    case 4: HE355_FLR(4, 4, 6, false); break;
    default:
        if (bz.L == 2 && bz.nB == 2) HE355_FLR(3, 2, 2, true);
        else if (bz.L == 3 && bz.nB == 3) HE355_FLR(3, 3, 3, true);
        else if (bz.L == 4 && bz.nB == 4) HE355_FLR(3, 4, 4, true);
        else HE355_FLR(3, 4, 6, false);
        break;
    }
#undef HE355_FLR
}
void launch_behz_rows_tensor(const KernelEnv &env, const BehzDev &bz, u64 n_ops, const u64 *xq, const u64 *xbsk, u64 *dq, u64 *ds)
{
    if (!n_ops) return;
    BehzRowsArgs AE[2];
    unsigned ge[2] = {0, 0};
    const int S = bz.nB + 1;
    for (int pass = 0; pass < 2; ++pass) { // pass 0: fp64-engine residues of both bases, pass 1: u64-engine residues
        BehzRowsArgs &A = AE[pass];
        A.x[0] = xq; A.x[1] = xbsk; A.d[0] = dq; A.d[1] = ds; A.Lx[0] = bz.L; A.Lx[1] = S;
        A.n_ops = n_ops; A.logn1 = env.logn1;
        A.n_r = 0;
        for (int i = 0; i < bz.L + S; ++i) {
            const int base = i < bz.L ? 0 : 1, idx = base ? i - bz.L : i, prime = base ? bz.bsk_prime[idx] : idx;
            if ((env.prime_f64[prime] != 0) != (pass == 0)) continue;
            A.r_base[A.n_r] = (unsigned char)base; A.r_idx[A.n_r] = (unsigned char)idx; A.r_prime[A.n_r] = (unsigned char)prime;
            ++A.n_r;
        }
        ge[pass] = (unsigned)((n_ops * A.n_r) << env.logn1);
    }
    if (ge[0] && ge[1] && ge[0] + ge[1] <= dual_max_blocks() && dual_engine_launches()) {
        hipLaunchKernelGGL(k_behz_rows_tensor_dual, dim3(ge[0] + ge[1]), dim3(kBlock), 0, env.stream, AE[0], AE[1], ge[0], env.primes);
        return;
    }
    if (ge[0]) hipLaunchKernelGGL(k_behz_rows_tensor<ArF64>, dim3(ge[0]), dim3(kBlock), 0, env.stream, AE[0], env.primes);
    if (ge[1]) hipLaunchKernelGGL(k_behz_rows_tensor<ArU64>, dim3(ge[1]), dim3(kBlock), 0, env.stream, AE[1], env.primes);
}
void launch_behz_tensor_inv(const KernelEnv &env, const BehzDev &bz, const BehzSrc &src, u64 n_ops, u64 op_offset, const u64 *eq, const u64 *ebsk, u64 *dq,
                            u64 *ds)
{
    if (!n_ops) return;
    BehzTensorArgs AE[2];
    u64 jobs[2] = {0, 0};
    const int S = bz.nB + 1;
    for (int pass = 0; pass < 2; ++pass) { // pass 0: fp64-engine residues of both bases, pass 1: u64-engine residues
        BehzTensorArgs &A = AE[pass];
        A.e[0] = eq; A.e[1] = ebsk; A.d[0] = dq; A.d[1] = ds; A.Lx[0] = bz.L; A.Lx[1] = S;
        A.src = src; A.n_ops = n_ops; A.op_offset = op_offset; A.logn1 = env.logn1;
        A.n_r = 0;
        for (int i = 0; i < bz.L + S; ++i) {
            const int base = i < bz.L ? 0 : 1, idx = base ? i - bz.L : i, prime = base ? bz.bsk_prime[idx] : idx;
            if ((env.prime_f64[prime] != 0) != (pass == 0)) continue;
            A.r_base[A.n_r] = (unsigned char)base; A.r_idx[A.n_r] = (unsigned char)idx; A.r_prime[A.n_r] = (unsigned char)prime;
            ++A.n_r;
        }
        jobs[pass] = ((n_ops * A.n_r) << env.logn1) * 3;
    }
    const unsigned g0 = grid_for(jobs[0], kWaves), g1 = grid_for(jobs[1], kWaves);
    if (jobs[0] && jobs[1] && g0 + g1 <= dual_max_blocks() && dual_engine_launches()) {
        hipLaunchKernelGGL(k_behz_tensor_inv_dual, dim3(g0 + g1), dim3(kBlock), 0, env.stream, AE[0], AE[1], jobs[0], jobs[1], g0, env.primes);
        return;
    }
    if (jobs[0]) hipLaunchKernelGGL(k_behz_tensor_inv<ArF64>, dim3(g0), dim3(kBlock), 0, env.stream, AE[0], env.primes, jobs[0]);
    if (jobs[1]) hipLaunchKernelGGL(k_behz_tensor_inv<ArU64>, dim3(g1), dim3(kBlock), 0, env.stream, AE[1], env.primes, jobs[1]);
}
void launch_rows_fwd(const KernelEnv &env, const PolyView &v, u32 n_items) // the row half of launch_ntt_forward (raw, or canonical when N = 1024, -> NTT form)
{
    const u64 jobs = ((u64)n_items * v.polys_per_item) << env.logn1;
    if (!jobs) return;
    hipLaunchKernelGGL(k_rows_fwd, dim3(grid_for(jobs, kWaves)), dim3(kBlock), 0, env.stream, v, env.primes, jobs, env.logn1, env.logn1 > 0 ? 1 : 0);
}
void launch_cols_fwd(const KernelEnv &env, const PolyView &v, u32 n_items) // the column half of launch_ntt_forward (canonical -> raw)
{
    const u64 polys = (u64)n_items * v.polys_per_item;
    if (!polys || env.logn1 == 0) return;
    const unsigned g = (unsigned)(polys * 4);
    switch (env.logn1) {
    case 1: hipLaunchKernelGGL(k_cols_fwd<1>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    case 2: hipLaunchKernelGGL(k_cols_fwd<2>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    case 3: hipLaunchKernelGGL(k_cols_fwd<3>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    case 4: hipLaunchKernelGGL(k_cols_fwd<4>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    case 5: hipLaunchKernelGGL(k_cols_fwd<5>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    }
}
void launch_cols_inv(const KernelEnv &env, const PolyView &v, u32 n_items) // the column half of launch_ntt_inverse (raw -> coefficients)
{
    const u64 polys = (u64)n_items * v.polys_per_item;
    if (!polys || env.logn1 == 0) return;
    const unsigned g = (unsigned)(polys * 4);
    switch (env.logn1) {
    case 1: hipLaunchKernelGGL(k_cols_inv<1>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    case 2: hipLaunchKernelGGL(k_cols_inv<2>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    case 3: hipLaunchKernelGGL(k_cols_inv<3>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    case 4: hipLaunchKernelGGL(k_cols_inv<4>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    case 5: hipLaunchKernelGGL(k_cols_inv<5>, dim3(g), dim3(kBlock), 0, env.stream, v, env.primes); break;
    }
}
void launch_behz_floor_sk(const KernelEnv &env, const BehzDev &bz, u64 n_ops, const u64 *dq, const u64 *ds, u64 *out)
{
    if (!n_ops) return;
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (n_ops * 3) << logN;
    if (bz.L <= 4 && bz.nB <= 6)
        hipLaunchKernelGGL((k_behz_floor_sk<4, 6>), dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, bz, env.primes, dq, ds, out, n_ops * 3, logN);
    else
        hipLaunchKernelGGL((k_behz_floor_sk<kBehzMaxL, kBehzMaxB>), dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, bz, env.primes, dq, ds, out, n_ops * 3, logN);
}
void launch_bfv_galois(const KernelEnv &env, int L, u64 n_ops, const u64 *in, const uint32_t *gather, u64 *c01, u64 c01_item_stride, u64 *tgt,
                       const u64 *addend)
{
    if (!n_ops) return;
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (n_ops * L) << logN;
    hipLaunchKernelGGL(k_bfv_galois, dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, in, gather, c01, c01_item_stride, tgt, env.primes, L, logN,
                       n_ops, addend);
}
void launch_bfv_tail_sp(const KernelEnv &env, u64 n_polys, const u64 *tpr, u64 *rp)
{
    if (!n_polys) return;
    const unsigned g = (unsigned)(n_polys * 4);
    const int sp = env.K - 1;
    switch (env.logn1) {
    case 0: hipLaunchKernelGGL(k_bfv_tail_sp<0>, dim3(g), dim3(kBlock), 0, env.stream, tpr, rp, env.primes, sp); break;
    case 1: hipLaunchKernelGGL(k_bfv_tail_sp<1>, dim3(g), dim3(kBlock), 0, env.stream, tpr, rp, env.primes, sp); break;
    case 2: hipLaunchKernelGGL(k_bfv_tail_sp<2>, dim3(g), dim3(kBlock), 0, env.stream, tpr, rp, env.primes, sp); break;
    case 3: hipLaunchKernelGGL(k_bfv_tail_sp<3>, dim3(g), dim3(kBlock), 0, env.stream, tpr, rp, env.primes, sp); break;
    case 4: hipLaunchKernelGGL(k_bfv_tail_sp<4>, dim3(g), dim3(kBlock), 0, env.stream, tpr, rp, env.primes, sp); break;
    case 5: hipLaunchKernelGGL(k_bfv_tail_sp<5>, dim3(g), dim3(kBlock), 0, env.stream, tpr, rp, env.primes, sp); break;
    }
}
void launch_bfv_tail_fin(const KernelEnv &env, int L, u64 n_ops, const u64 *t, const u64 *rp, u64 *c01, u64 c01_item_stride, const u64 *add01,
                         u64 add01_item_stride)
{
    if (!add01) { add01 = c01; add01_item_stride = c01_item_stride; }
    if (!n_ops) return;
    const unsigned g = (unsigned)(n_ops * 2 * L * 4);
    switch (env.logn1) {
    case 0: hipLaunchKernelGGL(k_bfv_tail_fin<0>, dim3(g), dim3(kBlock), 0, env.stream, t, rp, c01, c01_item_stride, add01, add01_item_stride, env.primes, env.floor_consts, L, env.K); break;
    case 1: hipLaunchKernelGGL(k_bfv_tail_fin<1>, dim3(g), dim3(kBlock), 0, env.stream, t, rp, c01, c01_item_stride, add01, add01_item_stride, env.primes, env.floor_consts, L, env.K); break;
    case 2: hipLaunchKernelGGL(k_bfv_tail_fin<2>, dim3(g), dim3(kBlock), 0, env.stream, t, rp, c01, c01_item_stride, add01, add01_item_stride, env.primes, env.floor_consts, L, env.K); break;
    case 3: hipLaunchKernelGGL(k_bfv_tail_fin<3>, dim3(g), dim3(kBlock), 0, env.stream, t, rp, c01, c01_item_stride, add01, add01_item_stride, env.primes, env.floor_consts, L, env.K); break;
    case 4: hipLaunchKernelGGL(k_bfv_tail_fin<4>, dim3(g), dim3(kBlock), 0, env.stream, t, rp, c01, c01_item_stride, add01, add01_item_stride, env.primes, env.floor_consts, L, env.K); break;
    case 5: hipLaunchKernelGGL(k_bfv_tail_fin<5>, dim3(g), dim3(kBlock), 0, env.stream, t, rp, c01, c01_item_stride, add01, add01_item_stride, env.primes, env.floor_consts, L, env.K); break;
    }
}

// (the client-side kernels -- encryption, decryption, encoders, key generation -- and their launchers: he355_kernels_client.hip)
} // namespace HE355_KNS
} // namespace he355
