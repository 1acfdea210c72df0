// he355_kernels_lds.hip -- key switch for rings that fit one CU's LDS (N <= 8192): ONE workgroup owns a whole residue polynomial.
//
// The reference's descriptors default to N = 8192 with two or three data primes, and half of them are Latency (batch 1):
// /root/reference/src/benchmarks/ckks/seal_ckks_dot_product_benchmark.cpp:53-60, .../ckks/seal_ckks_element_wise_benchmark.cpp:138-141.
// There a residue polynomial is 64 KiB, CDNA4 has 160 KiB of LDS per CU, and the N = 2^15 structure of he355_kernels.hip (row pass and
// column pass as separate launches with a global row <-> column transposition between them: 6 dependent launches per key switch) spends its
// time on launch boundaries and single-wave dependency chains.  Here a transform never leaves the CU:
//
//   k_lds_digits   one block per (op, key prime i, digit j != i):  target residue j (the tensor product's c2, a Galois-permuted c1, or a
//                  polynomial in memory) -> inverse row pass (one wave per row, 16 elements per lane) -> rows parked in LDS -> inverse
//                  column pass, lift to q_i, forward column pass (a lane owns a column) -> LDS -> forward row pass -> times
//                  key_j[k][i], k = 0, 1 -> partial products (canonical) to HBM.  Two transforms per block, nothing else leaves the CU.
//   k_lds_floor    (the mod-down; the same kernel is the CKKS rescale) one block per (op, data prime i, polynomial k):  sum of the special prime's partial products -> inverse transform in
//                  LDS -> r = (t + P/2) mod P, delta = (r mod q_i) - (P/2 mod q_i) -> forward transform in LDS -> ((sum of the partial
//                  products under q_i + own digit x key_i[k][i]) - NTT(delta)) P^-1 + addend_k.  Two transforms per block.
//
// 2 launches instead of 6 (the tensor product / the Galois permutation are folded into the loads of both kernels), L^2 + 2L blocks per
// ciphertext.  The inverse transform of digit j is repeated by the L blocks that lift it and the special prime's inverse transform by the L
// blocks that consume it: on a chip that idles at these sizes the redundancy costs nothing, the dependent chain is what counts.
// Same lane programs as every other kernel (ntt_core.h), same exact arithmetic, canonical outputs: bit-identical to the other shapes
// (tests/test_gpu_parity.py::test_lds_shape_*), SEAL's switch_key_inplace step for step.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <stdexcept>
#include <type_traits>

#include "he355_kernels.h"
#include "ntt_core.h"

#if !defined(HE355_KNS) || !defined(HE355_U64_FOLD)
#error "he355_kernels_lds.hip is compiled once per form of the u64 engine (Makefile)"
#endif
namespace he355 {
namespace HE355_KNS {
namespace {

#include "kernel_common.inc"

// -DHE355_LDS_TRACE (a VARIANT build, never the product): lane 0 of one chosen block stamps the 100 MHz clock at its phase boundaries and
// the launcher prints the differences after a synchronisation -- where a batch-1 key switch spends its 60 us (profiles/r06_lds_shape.txt)
#if defined(HE355_LDS_TRACE)
__device__ u64 g_lds_trace[2][16];
#define LDS_STAMP(kern, blk, idx)                                                                          \
    do {                                                                                                   \
        if (blockIdx.x == (blk) && threadIdx.x == 0) g_lds_trace[kern][idx] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define LDS_STAMP(kern, blk, idx) ((void)0)
#endif

struct LdsKsArgs {
    LdsKsOperands src;
    const u64 *key;  // [Ltop][2][K][N] (engine format: fp64-engine residues as doubles)
    const u64 *keyq; // [Ltop][2][n_q][N]: companion words of the key's residues under the u64-engine primes (he355_api.hip: k_key_quotients)
    int n_q;
    u64 *part;       // [n_ops][L + 2][2][L][N], canonical: slot s <= L (key prime s; L = the special prime), polynomial k, digit j: the product of
                     // digit j with key_j[k][s]; slot L + 1, polynomial k, prime i: the row the switched key part is added into
    u64 *out;        // [n_ops] x out_op_stride: [2][L][N]
    u64 out_op_stride;
    int L, K;
    unsigned trace_block; // HE355_LDS_TRACE builds: the block that stamps
};

// x * key as a canonical residue: the engine's key product (modarith.h acc_mac_lazy: the u64 engine multiplies by the key's companion
// word -- exact Shoup quotient or key 2^32 mod q -- and takes ANY 64-bit x; the fp64 engine needs none), x a lazy value of the engine
template <class Ar> __device__ __forceinline__ u64 mul_key_canon(const Ar &ar, typename Ar::T x, u64 keybits, u64 key2)
{
    typename Ar::Acc acc = ar.acc_from_canon(0);
    ar.acc_mac_lazy(acc, x, ar.key_in(keybits), key2);
    return ar.acc_canon(acc);
}
// position of key prime i among the u64-engine primes of the chain (the companion words are stored for those only)
__device__ __forceinline__ int q_slot(const PrimeDev *primes, int i)
{
    int s = 0;
    for (int t = 0; t < i; ++t) s += primes[t].f64 == 0;
    return s;
}
// the wave's row of key_j[k][i] and, for a u64-engine prime, of its companion words
template <class Ar>
__device__ __forceinline__ void key_rows(const LdsKsArgs &A, int qslot, int j, int k, int i, u64 N, u64 rowoff, int lane, u64 kv[kRowE], u64 kq[kRowE])
{
    load_rowC(A.key + (((u64)j * 2 + (u64)k) * A.K + i) * N + rowoff, lane, kv);
    if constexpr (Ar::kKeyQuotient) load_rowC(A.keyq + (((u64)j * 2 + (u64)k) * A.n_q + (u64)qslot) * N + rowoff, lane, kq);
    else {
#pragma unroll
        for (int r = 0; r < kRowE; ++r) kq[r] = 0;
    }
}

// One row (layout C, canonical) of the Galois-permuted polynomial `poly` of the op's source ciphertext: the NTT-domain permutation maps
// a row onto ONE source row (Params::galois_perm_ntt checks it), read whole with 16-byte loads and permuted through the wave's LDS row
// (the source row's index comes with the arguments -- the host knows the table -- so the row's load does not wait for the table's)
template <class T16>
__device__ __forceinline__ void permuted_row(const u64 *src_poly, const uint32_t *perm, u32 src_row, u32 a_row, int lane, u64 *lds, T16 &v)
{
    const uint32_t *pm = perm + ((u64)a_row << kRowLog);
    const u64 *srow = src_poly + ((u64)src_row << kRowLog);
    u64 t[kRowE];
    load_rowC(srow, lane, t);
    HE_WAVE_SYNC();
    lds_store_C(lds, lane, t);
    HE_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < kRowE; ++r) v[r] = lds[lds_pad((int)(pm[elemC(lane, r)] & (u32)(kRowN - 1)))];
    HE_WAVE_SYNC();
}

// Row a_row (layout C, canonical) of the key-switch TARGET's residue under prime i: the polynomial the digits are taken from
template <class Ar>
__device__ __forceinline__ void target_row(const LdsKsOperands &S, const Ar &ar, int L, u64 N, u64 op, int i, u32 a_row, int lane, u64 *lds, u64 v[kRowE])
{
    const u64 P1 = (u64)L * N, roff = (u64)i * N + ((u64)a_row << kRowLog);
    if (S.mode == LDSKS_MUL) { // c2 = a1 b1 (Evaluator::multiply, CKKS: dyadic product of the second polynomials)
        const u64 r = S.op_offset + op;
        u64 a1[kRowE], b1[kRowE];
        load_rowC(S.a + idx_a(S.ix, r) * 2 * P1 + P1 + roff, lane, a1);
        load_rowC(S.b + idx_b(S.ix, r) * 2 * P1 + P1 + roff, lane, b1);
#pragma unroll
        for (int e = 0; e < kRowE; ++e) v[e] = ar.dy_out(ar.dy_mul(ar.dy_in(a1[e]), ar.dy_in(b1[e])));
    } else if (S.mode == LDSKS_GALOIS) { // the permuted c1
        permuted_row(S.a + (S.op_offset + op) * 2 * P1 + P1 + (u64)i * N, S.perm, S.perm_src_row[a_row], a_row, lane, lds, v);
    } else {
        load_rowC(S.tgt + op * S.tgt_op_stride + roff, lane, v);
    }
}
// Row a_row of polynomial k of the ciphertext the switched key part is added into (canonical)
template <class Ar>
__device__ __forceinline__ void addend_row(const LdsKsOperands &S, const Ar &ar, u64 q, int L, u64 N, u64 op, int k, int i, u32 a_row, int lane, u64 *lds, u64 v[kRowE])
{
    const u64 P1 = (u64)L * N, roff = (u64)i * N + ((u64)a_row << kRowLog);
    if (S.mode == LDSKS_MUL) { // c0 = a0 b0, c1 = a0 b1 + a1 b0
        const u64 r = S.op_offset + op;
        const u64 *pa = S.a + idx_a(S.ix, r) * 2 * P1 + roff, *pb = S.b + idx_b(S.ix, r) * 2 * P1 + roff;
        u64 a0[kRowE], b0[kRowE];
        load_rowC(pa, lane, a0);
        if (k == 0) {
            load_rowC(pb, lane, b0);
#pragma unroll
            for (int e = 0; e < kRowE; ++e) v[e] = ar.dy_out(ar.dy_mul(ar.dy_in(a0[e]), ar.dy_in(b0[e])));
        } else {
            u64 a1[kRowE], b1[kRowE];
            load_rowC(pb, lane, b0);
            load_rowC(pa + P1, lane, a1);
            load_rowC(pb + P1, lane, b1);
#pragma unroll
            for (int e = 0; e < kRowE; ++e)
                v[e] = ar.dy_out(ar.dy_add(ar.dy_mul(ar.dy_in(a0[e]), ar.dy_in(b1[e])), ar.dy_mul(ar.dy_in(a1[e]), ar.dy_in(b0[e]))));
        }
        return;
    }
    if (S.mode == LDSKS_GALOIS) { // (permuted c0 [+ addend0], [addend1])
        if (k == 0) permuted_row(S.a + (S.op_offset + op) * 2 * P1 + (u64)i * N, S.perm, S.perm_src_row[a_row], a_row, lane, lds, v);
        else {
#pragma unroll
            for (int e = 0; e < kRowE; ++e) v[e] = 0;
        }
        if (S.add) {
            u64 ad[kRowE];
            load_rowC(S.add + op * S.add_op_stride + (u64)k * P1 + roff, lane, ad);
#pragma unroll
            for (int e = 0; e < kRowE; ++e) v[e] = addmod(v[e], ad[e], q);
        }
        return;
    }
    if (S.add) load_rowC(S.add + op * S.add_op_stride + (u64)k * P1 + roff, lane, v);
    else {
#pragma unroll
        for (int e = 0; e < kRowE; ++e) v[e] = 0;
    }
}

// The inverse row pass of kernel_common.inc (wave_rows_inv) with phase C's twiddles handed in: the kernels here request them at block start,
// next to the row itself, instead of behind it (one exposed memory latency less per transform)
template <class Ar>
__device__ __forceinline__ void wave_rows_inv_pre(const Ar &ar, const PrimeDev &P, bool last, u32 rowbase, int lane, u64 *lds_w, typename Ar::T x[kRowE],
                                                  const Tw16 wc[kTwInvC])
{
    typedef typename Ar::T T;
    T *lds = reinterpret_cast<T *>(lds_w);
    const auto itw = tw_table(gtw(P.inv), rowbase);
    row_inv_C_w(ar, x, wc);
    Tw16 wb[kTwInvB];
    gather_inv_B(itw, lane, wb);
    __builtin_amdgcn_sched_barrier(0);
    lds_store_C(lds, lane, x);
    HE_WAVE_SYNC();
    lds_load_B(lds, lane, x);
    HE_WAVE_SYNC();
    row_inv_B_w(ar, x, wb);
    Tw16 wa[kTwInvA];
    gather_inv_A(tw_table(ctw(P.inv), rowbase), wa);
    __builtin_amdgcn_sched_barrier(0);
    lds_store_B(lds, lane, x);
    HE_WAVE_SYNC();
    lds_load_A(lds, lane, x);
    HE_WAVE_SYNC();
    if (last) row_inv_A_w<Ar, true>(ar, x, wa, P.inv_w0_scaled);
    else row_inv_A_w<Ar, false>(ar, x, wa, P.inv_w0_scaled);
}

// ---- a whole transform pair inside the block: rows in registers -> inverse under prime J -> per-coefficient map -> forward under prime I --
// x: the wave's row of the source (layout C, engine J values).  On return y holds the wave's row of the result in NTT form under prime I
// (layout C, lazy).  `slot` is the block's LDS image of the polynomial: N1 rows of kLdsRow words (the padded row of ntt_core.h, which is also
// each wave's exchange buffer while its row is in registers).  lift(c) maps a canonical coefficient under q_J to a canonical residue under q_I.
// before_c: runs before the last exchange of the forward row pass (kernel_common.inc wave_rows_fwd_n): loads a caller needs right after the
// transform are put in flight there and land behind phase C instead of occupying registers through both transforms.
template <int LOGN1, class ArJ, class ArI, class Lift, class Hook = NoHook>
__device__ __forceinline__ void block_inv_map_fwd(const ArJ &arJ, const PrimeDev &PJ, const ArI &arI, const PrimeDev &PI, int wave, int lane, u64 (*slot)[kLdsRow],
                                                  typename ArJ::T x[kRowE], typename ArI::T y[kRowE], const Tw16 wc_inv[kTwInvC], Lift lift, int kern = 0,
                                                  unsigned tb = 0, Hook before_c = Hook())
{
    (void)kern; (void)tb;
    LDS_STAMP(kern, tb, 1);
    constexpr int N1 = 1 << LOGN1;
    constexpr bool kLast = LOGN1 == 0;
    wave_rows_inv_pre(arJ, PJ, kLast, (u32)(N1 + wave), lane, slot[wave], x, wc_inv); // -> layout A
    {
        u64 raw[kRowE];
#pragma unroll
        for (int r = 0; r < kRowE; ++r) raw[r] = kLast ? arJ.to_canon(x[r]) : arJ.to_raw(x[r]);
        HE_WAVE_SYNC();
        lds_store_A(slot[wave], lane, raw);
    }
    LDS_STAMP(kern, tb, 2);
    __syncthreads();
    LDS_STAMP(kern, tb, 3);
    // column phase: the block's 64 N1 lanes take the 1024 columns, 16 / N1 each
    constexpr int kColsPerLane = kRowE / N1;
#pragma unroll 1
    for (int cc = 0; cc < kColsPerLane; ++cc) {
        const int col = lds_pad((int)threadIdx.x + cc * 64 * N1);
        typename ArJ::T cj[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) cj[a] = kLast ? (typename ArJ::T)0 : arJ.from_raw(slot[a][col]);
        u64 c[N1];
        if constexpr (kLast) {
            c[0] = slot[0][col];
        } else {
            col_inv<ArJ, LOGN1>(arJ, cj, ctw(PJ.inv), PJ.inv_w0_scaled);
#pragma unroll
            for (int a = 0; a < N1; ++a) c[a] = arJ.to_canon(cj[a]);
        }
        typename ArI::T ci[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) ci[a] = arI.from_canon(lift(c[a]));
        if constexpr (!kLast) col_fwd<ArI, LOGN1>(arI, ci, ctw(PI.fwd));
#pragma unroll
        for (int a = 0; a < N1; ++a) slot[a][col] = kLast ? arI.to_canon(ci[a]) : arI.to_raw(ci[a]);
    }
    LDS_STAMP(kern, tb, 4);
    __syncthreads();
    LDS_STAMP(kern, tb, 5);
    {
        u64 raw[kRowE];
        lds_load_A(slot[wave], lane, raw);
#pragma unroll
        for (int r = 0; r < kRowE; ++r) y[r] = kLast ? arI.from_canon(raw[r]) : arI.from_raw(raw[r]);
        HE_WAVE_SYNC();
    }
    // (the wide lazy row pass of the u64 engine -- no conditional subtraction per butterfly: every key-chain prime is below 2^60,
    // he_params.cpp -- out below 12q / 8q + 16c; both consumers reduce once per element.  The fp64 engine's pass is lazy as it is.)
    wave_rows_fwd_n<ArI, decltype(tw_table(gtw(PI.fwd), 0u)), Hook, true>(arI, tw_table(gtw(PI.fwd), (u32)(N1 + wave)), lane, slot[wave],
                                                                          reinterpret_cast<typename ArI::T(*)[kRowE]>(y), before_c); // -> layout C
    LDS_STAMP(kern, tb, 6);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// k_lds_digits: block = (op, key prime slot s in 0..L (L: the special prime), digit j), s != j
// ---------------------------------------------------------------------------------------------------------------------------------------
template <int LOGN1, class ArJ, class ArI>
__device__ __forceinline__ void lds_digits_body(const LdsKsArgs &A, const PrimeDev *primes, u64 op, int s, int i, int j, u64 (*slot)[kLdsRow])
{
    constexpr int N1 = 1 << LOGN1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 N = (u64)N1 << kRowLog;
    const PrimeDev &PJ = primes[j], &PI = primes[i];
    const ArJ arJ = make_ar(PJ, (ArJ *)nullptr);
    const ArI arI = make_ar(PI, (ArI *)nullptr);
    // the key rows this wave multiplies by at the very end: requested now, they land behind the two transforms
    LDS_STAMP(0, A.trace_block, 0);
    // the key rows this wave multiplies by at the very end are requested before the forward row pass's last exchange (before_c): two rows
    // and their companion words are 128 registers, too many to hold through both transforms
    u64 kv0[kRowE], kv1[kRowE], kq0[kRowE], kq1[kRowE];
    const int qs = ArI::kKeyQuotient ? q_slot(primes, i) : 0;
    auto key_loads = [&]() {
        key_rows<ArI>(A, qs, j, 0, i, N, (u64)wave << kRowLog, lane, kv0, kq0);
        key_rows<ArI>(A, qs, j, 1, i, N, (u64)wave << kRowLog, lane, kv1, kq1);
    };
    Tw16 wc_inv[kTwInvC];
    gather_inv_C(tw_table(gtw(PJ.inv), (u32)(N1 + wave)), lane, wc_inv); // (requested with the row itself)
    u64 v[kRowE];
    target_row(A.src, arJ, A.L, N, op, j, (u32)wave, lane, slot[wave], v);
    typename ArJ::T x[kRowE];
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = arJ.from_canon(v[r]);
    typename ArI::T y[kRowE];
    const ModU64 mI = make_modu(PI);
    const bool reduce = PJ.q > PI.q; // (SEAL: the digit is reduced only when q_j > q_i)
    block_inv_map_fwd<LOGN1>(arJ, PJ, arI, PI, wave, lane, slot, x, y, wc_inv, [&](u64 c) { return reduce ? barrett64(c, mI) : c; }, 0, A.trace_block, key_loads);
    u64 *p0 = A.part + ((((op * (u64)(A.L + 2) + (u64)s) * 2 + 0) * A.L + (u64)j) * N) + ((u64)wave << kRowLog);
    u64 *p1 = p0 + (u64)A.L * N;
    u64 o0[kRowE], o1[kRowE];
#pragma unroll
    for (int r = 0; r < kRowE; ++r) {
        o0[r] = mul_key_canon(arI, y[r], kv0[r], kq0[r]);
        o1[r] = mul_key_canon(arI, y[r], kv1[r], kq1[r]);
    }
    store_rowC(p0, lane, o0);
    store_rowC(p1, lane, o1);
    LDS_STAMP(0, A.trace_block, 7);
}

// The diagonal blocks (s == j): no transform.  They form what the mod-down (k_lds_floor) needs of the operands -- the own digit's products
// target_i x key_i[k][i] (into the diagonal slot of `part`, so the sum over the digits there is complete) and the rows the switched key
// part is added into (slot L + 1) -- while the other blocks transform; k_lds_floor then reads nothing but `part`, every load independent.
template <int LOGN1, class ArI>
__device__ __forceinline__ void lds_prep_body(const LdsKsArgs &A, const PrimeDev *primes, u64 op, int i, u64 (*slot)[kLdsRow])
{
    constexpr int N1 = 1 << LOGN1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 N = (u64)N1 << kRowLog, rowoff = (u64)wave << kRowLog;
    const int L = A.L;
    const PrimeDev &PI = primes[i];
    const ArI arI = make_ar(PI, (ArI *)nullptr);
    const int qs = ArI::kKeyQuotient ? q_slot(primes, i) : 0;
    u64 own[kRowE];
    target_row(A.src, arI, L, N, op, i, (u32)wave, lane, slot[wave], own);
    u64 *pbase = A.part + (op * (u64)(L + 2)) * 2 * L * N;
    for (int k = 0; k < 2; ++k) {
        u64 kv[kRowE], kq[kRowE], o[kRowE];
        key_rows<ArI>(A, qs, i, k, i, N, rowoff, lane, kv, kq);
#pragma unroll
        for (int r = 0; r < kRowE; ++r) o[r] = mul_key_canon(arI, arI.from_canon(own[r]), kv[r], kq[r]);
        store_rowC(pbase + ((((u64)i * 2 + (u64)k) * L + (u64)i) * N) + rowoff, lane, o);
        u64 ad[kRowE];
        addend_row(A.src, arI, PI.q, L, N, op, k, i, (u32)wave, lane, slot[wave], ad);
        store_rowC(pbase + ((((u64)(L + 1) * 2 + (u64)k) * L + (u64)i) * N) + rowoff, lane, ad);
    }
}

template <int LOGN1>
__global__ void __launch_bounds__(64 << LOGN1) k_lds_digits(LdsKsArgs A, const PrimeDev *primes)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    u64(*slot)[kLdsRow] = reinterpret_cast<u64(*)[kLdsRow]>(lds_raw);
    const unsigned per_op = (unsigned)((A.L + 1) * A.L);
    const u64 op = blockIdx.x / per_op;
    const unsigned p = blockIdx.x % per_op;
    const int s = (int)(p / (unsigned)A.L), j = (int)(p % (unsigned)A.L);
    if (s == j) { // the own digit needs no transform
        if (primes[s].f64) lds_prep_body<LOGN1, ArF64>(A, primes, op, s, slot);
        else lds_prep_body<LOGN1, ArU64>(A, primes, op, s, slot);
        return;
    }
    const int i = s == A.L ? A.K - 1 : s;
    const bool jf = primes[j].f64 != 0, iff = primes[i].f64 != 0;
    if (jf) {
        if (iff) lds_digits_body<LOGN1, ArF64, ArF64>(A, primes, op, s, i, j, slot);
        else lds_digits_body<LOGN1, ArF64, ArU64>(A, primes, op, s, i, j, slot);
    } else {
        if (iff) lds_digits_body<LOGN1, ArU64, ArF64>(A, primes, op, s, i, j, slot);
        else lds_digits_body<LOGN1, ArU64, ArU64>(A, primes, op, s, i, j, slot);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// k_lds_moddown: block = (op, data prime i, polynomial k)
// ---------------------------------------------------------------------------------------------------------------------------------------
// acc * s^-1 + addend as a canonical residue (acc, addend canonical); a lazy forward-pass value brought below 4q for floor_fin_s
__device__ __forceinline__ u64 scale_add_canon(const ArU64 &ar, u64 acc, const FloorConst &fc, u64 ad) { return addmod(mul_pre(acc, fc.inv, fc.inv_shoup, ar.q), ad, ar.q); }
__device__ __forceinline__ u64 scale_add_canon(const ArF64 &ar, u64 acc, const FloorConst &fc, u64 ad)
{
    return ar.to_canon2(ar.mulmod_c(u52_to_f64(acc), fc.inv_d, fc.inv_i) + u52_to_f64(ad));
}
__device__ __forceinline__ u64 lazy_to_4q(const ArU64 &ar, u64 x) { if constexpr (ArU64::kFold) return ar.lazy_reduce(x); else return ar.reduce16_to_4q(x); }
__device__ __forceinline__ double lazy_to_4q(const ArF64 &, double x) { return x; }

// One RNS floor step (divide-and-round by prime s) with both transforms in LDS -- the mod-down of a key switch (s = the special prime) and
// the CKKS rescale (s = the last data prime; SEAL divide_and_round_q_last_ntt_inplace) are the same step:
//   out[op][k][i] = ((sum of a_terms rows under q_i) - NTT_i(delta)) s^-1 + addend,   delta = ((t + s/2) mod s  mod q_i) - (s/2 mod q_i),
//   t = iNTT_s(sum of t_terms rows under s).   Block = (op, target prime i < n_tgt, polynomial k < n_polys).
struct LdsFloorArgs {
    int src_prime, n_tgt, n_polys;
    const u64 *t; u64 t_op_stride, t_poly_stride, t_term_stride; int t_terms;   // rows under prime s (NTT form, canonical)
    const u64 *a; u64 a_op_stride, a_poly_stride, a_prime_stride, a_term_stride; int a_terms; // rows under the target primes
    const u64 *add; u64 add_op_stride, add_poly_stride, add_prime_stride;        // null: nothing added
    u64 *out; u64 out_op_stride, out_poly_stride, out_prime_stride;
    int K;
};

template <int LOGN1, class ArP, class ArI>
__device__ __forceinline__ void lds_floor_body(const LdsFloorArgs &A, const PrimeDev *primes, const FloorConst *fcs, u64 op, int i, int k, u64 (*slot)[kLdsRow],
                                               u64 (*stash)[kRowN])
{
    constexpr int N1 = 1 << LOGN1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int SP = A.src_prime;
    const PrimeDev &PP = primes[SP], &PI = primes[i];
    const ArP arP = make_ar(PP, (ArP *)nullptr);
    const ArI arI = make_ar(PI, (ArI *)nullptr);
    const FloorConst fc = fcs[(u64)SP * A.K + i];
    const u64 rowoff = (u64)wave << kRowLog;
    LDS_STAMP(1, 0, 0);
    const u64 *sp = A.t + op * A.t_op_stride + (u64)k * A.t_poly_stride + rowoff;
    const u64 *pi = A.a + op * A.a_op_stride + (u64)k * A.a_poly_stride + (u64)i * A.a_prime_stride + rowoff;
    const u64 qi = PI.q;
    // every load of the block is independent of every other: one memory latency, then sums
    Tw16 wc_inv[kTwInvC];
    gather_inv_C(tw_table(gtw(PP.inv), (u32)(N1 + wave)), lane, wc_inv);
    u64 t[kRowE], acc[kRowE], ad[kRowE];
    load_rowC(sp, lane, t);
    load_rowC(pi, lane, acc);
    if (A.add) load_rowC(A.add + op * A.add_op_stride + (u64)k * A.add_poly_stride + (u64)i * A.add_prime_stride + rowoff, lane, ad);
    else {
#pragma unroll
        for (int r = 0; r < kRowE; ++r) ad[r] = 0;
    }
    for (int j = 1; j < A.t_terms; ++j) {
        u64 u[kRowE];
        load_rowC(sp + (u64)j * A.t_term_stride, lane, u);
#pragma unroll
        for (int r = 0; r < kRowE; ++r) t[r] = addmod(t[r], u[r], PP.q);
    }
    for (int j = 1; j < A.a_terms; ++j) {
        u64 w[kRowE];
        load_rowC(pi + (u64)j * A.a_term_stride, lane, w);
#pragma unroll
        for (int r = 0; r < kRowE; ++r) acc[r] = addmod(acc[r], w[r], qi);
    }
    // c = (sum under q_i) s^-1 + addend, parked in the wave's stash row: the epilogue is c - NTT(delta) s^-1
#pragma unroll
    for (int r = 0; r < kRowE; ++r) acc[r] = scale_add_canon(arI, acc[r], fc, ad[r]);
    store_rowC(stash[wave], lane, acc);
    typename ArP::T x[kRowE];
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = arP.from_canon(t[r]);
    typename ArI::T y[kRowE];
    const ModU64 mI = make_modu(PI);
    const u64 half = PP.q >> 1, qp = PP.q, half_i = fc.half_mod;
    block_inv_map_fwd<LOGN1>(arP, PP, arI, PI, wave, lane, slot, x, y, wc_inv, [&](u64 c) {
        u64 r = c + half;             // (t + floor(s/2)) mod s
        if (r >= qp) r -= qp;
        const u64 ri = qp > qi ? barrett64(r, mI) : r;
        return submod(ri, half_i, qi); // - (floor(s/2) mod q_i)
    }, 1, 0);
    u64 c[kRowE], o[kRowE];
    load_rowC(stash[wave], lane, c);
#pragma unroll
    for (int r = 0; r < kRowE; ++r) o[r] = arI.floor_fin_s(c[r], lazy_to_4q(arI, y[r]), fc.inv, fc.inv_shoup, fc.inv_d, fc.inv_i);
    store_rowC(A.out + op * A.out_op_stride + (u64)k * A.out_poly_stride + (u64)i * A.out_prime_stride + rowoff, lane, o);
    LDS_STAMP(1, 0, 7);
}

template <int LOGN1>
__global__ void __launch_bounds__(64 << LOGN1) k_lds_floor(LdsFloorArgs A, const PrimeDev *primes, const FloorConst *fcs)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    u64(*slot)[kLdsRow] = reinterpret_cast<u64(*)[kLdsRow]>(lds_raw);
    u64(*stash)[kRowN] = reinterpret_cast<u64(*)[kRowN]>(lds_raw + ((size_t)sizeof(u64) * kLdsRow << LOGN1)); // one unpadded row per wave behind the polynomial
    const unsigned per_op = (unsigned)(A.n_polys * A.n_tgt);
    const u64 op = blockIdx.x / per_op;
    const unsigned p = blockIdx.x % per_op;
    const int i = (int)(p / (unsigned)A.n_polys), k = (int)(p % (unsigned)A.n_polys);
    const bool pf = primes[A.src_prime].f64 != 0, iff = primes[i].f64 != 0;
    if (pf) {
        if (iff) lds_floor_body<LOGN1, ArF64, ArF64>(A, primes, fcs, op, i, k, slot, stash);
        else lds_floor_body<LOGN1, ArF64, ArU64>(A, primes, fcs, op, i, k, slot, stash);
    } else {
        if (iff) lds_floor_body<LOGN1, ArU64, ArF64>(A, primes, fcs, op, i, k, slot, stash);
        else lds_floor_body<LOGN1, ArU64, ArU64>(A, primes, fcs, op, i, k, slot, stash);
    }
}

template <int LOGN1> void lds_attrs()
{
    constexpr int N1 = 1 << LOGN1;
    constexpr size_t lds_bytes = (size_t)N1 * kLdsRow * sizeof(u64), lds_bytes2 = lds_bytes + (size_t)N1 * kRowN * sizeof(u64);
    // per build of the device code AND per device: a function's attributes belong to the device it was loaded on, and a DeviceGroup
    // (csrc/bridge/multi_device.cpp) drives several devices from one process
    static bool attr_done_dev[64] = {};
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) dev_id = 63; // (slot 63 is never marked done: set every time)
    bool &attr_done = attr_done_dev[dev_id];
    if (attr_done) return;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds_digits<LOGN1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds_floor<LOGN1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes2) != hipSuccess)
        throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the LDS-resident key switch");
    attr_done = dev_id != 63;
}
template <int LOGN1> void launch_floor_lds_n(const KernelEnv &env, const LdsFloorArgs &F, u64 n_ops)
{
    constexpr int N1 = 1 << LOGN1;
    constexpr size_t lds_bytes2 = (size_t)N1 * (kLdsRow + kRowN) * sizeof(u64);
    lds_attrs<LOGN1>();
    const u64 g = n_ops * (u64)F.n_polys * (u64)F.n_tgt;
    if (g > 0x7fffffffull) throw std::invalid_argument("launch_floor_lds: batch too large for one grid");
    if (g) hipLaunchKernelGGL(k_lds_floor<LOGN1>, dim3((unsigned)g), dim3(64 * N1), lds_bytes2, env.stream, F, env.primes, env.floor_consts);
}
void launch_floor_lds_any(const KernelEnv &env, const LdsFloorArgs &F, u64 n_ops)
{
    switch (env.logn1) {
    case 0: launch_floor_lds_n<0>(env, F, n_ops); break;
    case 1: launch_floor_lds_n<1>(env, F, n_ops); break;
    case 2: launch_floor_lds_n<2>(env, F, n_ops); break;
    default: launch_floor_lds_n<3>(env, F, n_ops); break;
    }
}

template <int LOGN1> void launch_ks_lds_n(const KernelEnv &env, const LdsKsArgs &A, u64 n_ops)
{
    constexpr int N1 = 1 << LOGN1;
    constexpr size_t lds_bytes = (size_t)N1 * kLdsRow * sizeof(u64);
    lds_attrs<LOGN1>();
    const u64 g1 = n_ops * (u64)(A.L + 1) * A.L;
    if (g1 > 0x7fffffffull) throw std::invalid_argument("launch_ks_lds: batch too large for one grid");
    hipLaunchKernelGGL(k_lds_digits<LOGN1>, dim3((unsigned)g1), dim3(64 * N1), lds_bytes, env.stream, A, env.primes);
    // the mod-down: a floor step by the special prime over what k_lds_digits left in `part` (slot L: its products under P; slot i: those under
    // q_i, the own digit's on the diagonal; slot L + 1: the rows the result is added into)
    const u64 N = (u64)N1 << kRowLog, L = (u64)A.L, slot_w = 2 * L * N;
    LdsFloorArgs F;
    F.src_prime = A.K - 1; F.n_tgt = A.L; F.n_polys = 2; F.K = A.K;
    F.t = A.part + L * slot_w; F.t_op_stride = (L + 2) * slot_w; F.t_poly_stride = L * N; F.t_term_stride = N; F.t_terms = A.L;
    F.a = A.part; F.a_op_stride = (L + 2) * slot_w; F.a_poly_stride = L * N; F.a_prime_stride = slot_w; F.a_term_stride = N; F.a_terms = A.L;
    F.add = A.part + (L + 1) * slot_w; F.add_op_stride = (L + 2) * slot_w; F.add_poly_stride = L * N; F.add_prime_stride = N;
    F.out = A.out; F.out_op_stride = A.out_op_stride; F.out_poly_stride = L * N; F.out_prime_stride = N;
    launch_floor_lds_n<LOGN1>(env, F, n_ops);
#if defined(HE355_LDS_TRACE)
    static int printed = 0;
    (void)hipStreamSynchronize(env.stream);
    if (++printed == 20) { // (a warm call)
        u64 h[2][16];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lds_trace), sizeof(h));
        const char *names[8] = {"start", "loads+prologue", "inverse row pass", "barrier", "column phase", "barrier", "forward row pass", "products/epilogue+stores"};
        for (int kk = 0; kk < 2; ++kk) {
            fprintf(stderr, "[lds trace] %s block %u:", kk ? "k_lds_floor (mod-down)" : "k_lds_digits", kk ? 0u : A.trace_block);
            for (int i = 1; i < 8; ++i) fprintf(stderr, " %s %.2f us |", names[i], (double)(h[kk][i] - h[kk][i - 1]) * 0.01);
            fprintf(stderr, " total %.2f us\n", (double)(h[kk][7] - h[kk][0]) * 0.01);
        }
        fprintf(stderr, "[lds trace] end of k_lds_digits' traced block -> start of k_lds_floor's block 0: %.2f us\n", (double)((long long)(h[1][0] - h[0][7])) * 0.01);
    }
#endif
}

} // namespace

// (L <= 6: the partial products, (2L + 4) L N words per ciphertext, live in the key-switch arena behind c01 -- (L + 10) L N words)
bool ks_lds_supported(const KernelEnv &env, int L) { return env.logn1 >= 0 && env.logn1 <= 3 && L >= 1 && L <= 6 && env.K >= 2 && L <= env.K - 1; }
u64 ks_lds_part_words(const KernelEnv &env, int L) { return (u64)(L + 2) * 2 * (u64)L * (u64)env.N; }

// CKKS rescale of n_ops size-`size` ciphertexts at level L (src: [size][L][N] per op, src_op_stride words apart) into out [n_ops][size][L-1][N]:
// one launch, block = (op, prime i < L - 1, polynomial)
void launch_rescale_lds(const KernelEnv &env, int L, int size, u64 n_ops, const u64 *src, u64 src_op_stride, u64 *out)
{
    if (!n_ops) return;
    if (!ks_lds_supported(env, L) || L < 2) throw std::invalid_argument("launch_rescale_lds: ring or level outside the LDS-resident shape");
    const u64 N = (u64)env.N, LN = (u64)L * N, L1N = (u64)(L - 1) * N;
    LdsFloorArgs F;
    F.src_prime = L - 1; F.n_tgt = L - 1; F.n_polys = size; F.K = env.K;
    F.t = src + (u64)(L - 1) * N; F.t_op_stride = src_op_stride; F.t_poly_stride = LN; F.t_term_stride = 0; F.t_terms = 1;
    F.a = src; F.a_op_stride = src_op_stride; F.a_poly_stride = LN; F.a_prime_stride = N; F.a_term_stride = 0; F.a_terms = 1;
    F.add = nullptr; F.add_op_stride = F.add_poly_stride = F.add_prime_stride = 0;
    F.out = out; F.out_op_stride = (u64)size * L1N; F.out_poly_stride = L1N; F.out_prime_stride = N;
    launch_floor_lds_any(env, F, n_ops);
}

void launch_ks_lds(const KernelEnv &env, int L, u64 n_ops, const LdsKsOperands &src, const u64 *key, u64 *part, u64 *out, u64 out_op_stride)
{
    int n_q = 0;
    for (int t = 0; t < env.K; ++t) n_q += env.prime_f64[t] == 0;
    if (!n_ops) return;
    if (!ks_lds_supported(env, L)) throw std::invalid_argument("launch_ks_lds: ring or level outside the LDS-resident shape");
    LdsKsArgs A;
    A.keyq = key + (u64)env.Ltop * 2 * (u64)env.K * (u64)env.N; A.n_q = n_q; // (the companion words follow the key: DeviceContext::key_alloc_elems)
    A.src = src; A.key = key; A.part = part; A.out = out; A.out_op_stride = out_op_stride; A.L = L; A.K = env.K;
    A.trace_block = (unsigned)(L * L); // (op 0: special prime, digit 0 -- two u64-engine transforms under the reference's chains)
    switch (env.logn1) {
    case 0: launch_ks_lds_n<0>(env, A, n_ops); break;
    case 1: launch_ks_lds_n<1>(env, A, n_ops); break;
    case 2: launch_ks_lds_n<2>(env, A, n_ops); break;
    default: launch_ks_lds_n<3>(env, A, n_ops); break;
    }
}

} // namespace HE355_KNS
} // namespace he355
