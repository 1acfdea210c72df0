// ntt_core.h — negacyclic NTT building blocks for CDNA4, shared by every kernel.
//
// A size-N transform (N = 2^n, 1024 <= N <= 32768) is split as N = N1 x 1024:
//   * column pass : N1 <= 32 point transforms over stride-1024 columns; one lane owns one column in
//                   registers; its twiddles are the same for every column (wave-uniform -> SGPRs).
//   * row pass    : 1024-point transforms over contiguous rows; ONE 64-lane wave owns one row, 16
//                   elements per lane; the 10 stages run as 4 + 4 + 2 register-resident stages with two
//                   register/lane transposes between them (layouts A -> B -> C below), each an exchange through the
//                   wave's own LDS region (padded against bank conflicts).  (Cross-lane swap steps -- permlane / DPP --
//                   were built and measured slower: HISTORY.md.)
// Forward (Cooley-Tukey, natural -> bit-reversed) = column pass then row pass; inverse (Gentleman-Sande)
// = row pass then column pass with N^-1 folded into the last stage.  The ordering of the NTT form
// (bit-reversed evaluations, twiddle table w[bitrev(i)] = psi^i) is the one SEAL uses, so device slabs are
// interchangeable with seal::Ciphertext::data().
//
// Lane programs are written as "phases" (code between two LDS hand-offs) so that tests/csim can execute
// the identical code lane by lane on the CPU.
#pragma once
#include "modarith.h"

namespace he355 {

constexpr int kRowLog = 10;
constexpr int kRowN = 1 << kRowLog; // 1024 elements per row
constexpr int kRowE = 16;           // elements per lane
constexpr int kLdsRow = kRowN + 64; // exchange buffer: one row plus 4 pad elements per 64-element block (8.5 KiB)

// element e lives at e + 4*(e >> 6): every 64-element block is followed by 4 pad elements (32 bytes), so layout-B
// accesses (4 contiguous elements per 4-lane group, groups 64 elements apart) walk the banks in 32-byte steps.
// The map is ADDITIVE in the lane part and the register part of e under all three layouts (the register part never
// carries into the lane part), so every LDS address is one per-lane base plus a compile-time offset: three address
// registers in all instead of one per element.
HE_HD int lds_pad(int e) { return e + ((e >> 6) << 2); }

// element index held in register r of lane `lane` under the three layouts
HE_HD int elemA(int lane, int r) { return (r << 6) | lane; }                                   // r = bits 9..6
HE_HD int elemB(int lane, int r) { return ((lane >> 2) << 6) | (r << 2) | (lane & 3); }       // r = bits 5..2
HE_HD int elemC(int lane, int r) { return ((lane >> 2) << 6) | ((r >> 2) << 4) | ((lane & 3) << 2) | (r & 3); } // r = bits 5,4,1,0
// Layout B -> C moves element bits 1..0 from lane bits 1..0 into register bits 1..0 (and bits 3..2 the other way): two
// 4x4 transposes inside every quad of lanes.  A lane then owns 4 x 4 consecutive elements; a quad owns 16 consecutive
// elements (128 bytes) for each value of register bits 3..2.

template <class T> HE_HD void lds_store_A(T *lds, int lane, const T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) lds[lds_pad(elemA(lane, r))] = x[r];
}
template <class T> HE_HD void lds_load_A(const T *lds, int lane, T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = lds[lds_pad(elemA(lane, r))];
}
template <class T> HE_HD void lds_store_B(T *lds, int lane, const T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) lds[lds_pad(elemB(lane, r))] = x[r];
}
template <class T> HE_HD void lds_load_B(const T *lds, int lane, T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = lds[lds_pad(elemB(lane, r))];
}
template <class T> HE_HD void lds_store_C(T *lds, int lane, const T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) lds[lds_pad(elemC(lane, r))] = x[r];
}
template <class T> HE_HD void lds_load_C(const T *lds, int lane, T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = lds[lds_pad(elemC(lane, r))];
}

// ---------------------------------------------------------------------------------------------------
// Twiddle accessors for the row pass.  Stage s' (0..9) of row a needs entry (rowbase << s') + g of the prime's
// table, g = e >> (10-s') in [0, 2^s'), rowbase = N1 + a.
//   TwTable : straight from the table (any pointer type, e.g. an address_space(1) pointer)
//   TwRow   : from a row-local copy of those 1023 entries (index 2^s' - 1 + g), e.g. staged in LDS
// ---------------------------------------------------------------------------------------------------
HE_HD Tw16 tw_load(const Tw16 *p, u32 i) { return p[i]; }
#if defined(__HIP__)
// explicit global-memory twiddle pointer: one global_load_dwordx4 per entry, never a flat load
typedef unsigned long long he_u64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) he_u64x2 *gtw_t;
HE_HD Tw16 tw_load(gtw_t p, u32 i)
{
    const he_u64x2 v = p[i];
    Tw16 t;
    t.a = v.x;
    t.b = v.y;
    return t;
}
#endif
#if defined(__HIP__)
// The same table through the constant address space: an entry whose index is the same in every lane (all column-pass
// twiddles are) becomes a scalar load into SGPRs -- no vector registers, no vector-memory latency in front of the butterflies.
// Only for memory no kernel writes while the reader runs (the per-prime twiddle tables).
typedef const __attribute__((address_space(4))) he_u64x2 *ctw_t;
HE_HD Tw16 tw_load(ctw_t p, u32 i)
{
    const he_u64x2 v = p[i];
    Tw16 t;
    t.a = v.x;
    t.b = v.y;
    return t;
}
#endif
// entries read through ctw_t are wave-uniform by contract (scalar loads): the u64 engine's butterflies keep them in scalar registers
template <class TW> struct TwIsUniform { static constexpr bool value = false; };
#if defined(__HIP__)
template <> struct TwIsUniform<ctw_t> { static constexpr bool value = true; };
#endif
template <class P> struct TwTable {
    P base;
    u32 rowbase;
    HE_HD Tw16 get(int s, u32 g) const { return tw_load(base, (rowbase << s) + g); }
};
template <class P> HE_HD TwTable<P> tw_table(P base, u32 rowbase)
{
    TwTable<P> t;
    t.base = base;
    t.rowbase = rowbase;
    return t;
}
constexpr int kRowTw = kRowN; // 1023 entries used
struct TwRow {
    const Tw16 *t;
    HE_HD Tw16 get(int s, u32 g) const { return t[(1u << s) - 1u + g]; }
};
// fp64 engine: the row-local copy keeps only w (8 bytes); its butterflies take the quotient estimate from
// h * (1/q), so nothing else is needed.
struct TwRowF64 {
    const double *t;
    double qinv;
    HE_HD Tw16 get(int s, u32 g) const
    {
        Tw16 r;
        union { u64 u; double d; } c;
        c.d = t[(1u << s) - 1u + g];
        r.a = c.u;
        r.b = 0; // the fp64 butterflies do not use the second word
        return r;
    }
};
// global index of row-local entry i (0..1022): stage s = floor(log2(i+1)), g = i + 1 - 2^s
HE_HD u32 tw_row_source(u32 rowbase, u32 i)
{
    const u32 v = i + 1;
    int s = 0;
    while ((v >> (s + 1)) != 0) ++s;
    return (rowbase << s) + (v - (1u << s));
}

// ---------------------------------------------------------------------------------------------------
// Row pass, forward
// ---------------------------------------------------------------------------------------------------
// A phase (A: stages 0-3, B: 4-7, C: 8-9) runs entirely from registers: its twiddles are gathered up front
// (gather_*) so that a kernel can issue those loads BEFORE the LDS exchange that precedes the phase and have
// them land while the exchange is in flight.  Per lane: phase A 15 (lane-uniform), B 15, C 12 entries.
// U rows of the SAME (prime, row index) can be transformed together (x[u][..]): they share every twiddle, and
// their independent butterfly chains give one wave the instruction-level parallelism to cover latencies.
constexpr int kTwA = 15, kTwB = 15, kTwC = 12;
template <class TW> HE_HD void gather_A(const TW &tw, Tw16 w[kTwA])
{
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < (1 << s); ++g) w[(1 << s) - 1 + g] = tw.get(s, (u32)g);
}
template <class TW> HE_HD void gather_B(const TW &tw, int lane, Tw16 w[kTwB])
{
    const u32 hi4 = (u32)lane >> 2;
#pragma unroll
    for (int s = 4; s < 8; ++s)
#pragma unroll
        for (int g = 0; g < (1 << (s - 4)); ++g) w[(1 << (s - 4)) - 1 + g] = tw.get(s, (hi4 << (s - 4)) | (u32)g);
}
template <class TW> HE_HD void gather_C(const TW &tw, int lane, Tw16 w[kTwC])
{
    // stage 8: one entry per chunk c = r >> 2 (index e >> 2 of its elements); stage 9: two per chunk (e >> 1, h = bit 1 of r)
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = tw.get(8, (u32)elemC(lane, 4 * c) >> 2);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) w[4 + 2 * c + h] = tw.get(9, (u32)elemC(lane, 4 * c + 2 * h) >> 1);
}
// stages 0..3 on layout A (register bit 3-s' is the butterfly bit); twiddles are lane-uniform
// One radix-2 stage over the 16 registers of U rows: butterflies (r, r | 1<<BIT) with twiddle index widx(r), issued
// in groups of kBflyGroup independent butterflies (Ar::bfly_fwd_g).
#ifndef HE355_BFLY_GROUP
#define HE355_BFLY_GROUP 2
#endif
constexpr int kBflyGroup = HE355_BFLY_GROUP;
template <int U, int BIT, bool LAZY, bool SW = false, class Ar, class WIdx> HE_HD void row_fwd_stage_x(const Ar &ar, typename Ar::T (*x)[kRowE], const Tw16 *w, WIdx widx)
{
    constexpr int kTotal = 8 * U, G = kBflyGroup < kTotal ? kBflyGroup : kTotal;
    static_assert(kTotal % G == 0, "group size must divide the butterflies of a stage");
#pragma unroll
    for (int g0 = 0; g0 < kTotal; g0 += G) {
        typename Ar::T X[G], Y[G];
        Tw16 W[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int b = (g0 + k) / U, u = (g0 + k) % U;                        // butterfly b of row u
            const int r = ((b >> BIT) << (BIT + 1)) | (b & ((1 << BIT) - 1));    // b with a 0 inserted at BIT
            X[k] = x[u][r]; Y[k] = x[u][r | (1 << BIT)]; W[k] = w[widx(r)];
        }
        if constexpr (LAZY) ar.template bfly_fwd_lazy_g<G, SW>(X, Y, W);
        else ar.template bfly_fwd_g<G>(X, Y, W);
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int b = (g0 + k) / U, u = (g0 + k) % U;
            const int r = ((b >> BIT) << (BIT + 1)) | (b & ((1 << BIT) - 1));
            x[u][r] = X[k]; x[u][r | (1 << BIT)] = Y[k];
        }
    }
}
template <int U, int BIT, class Ar, class WIdx> HE_HD void row_fwd_stage(const Ar &ar, typename Ar::T (*x)[kRowE], const Tw16 *w, WIdx widx)
{
    row_fwd_stage_x<U, BIT, false>(ar, x, w, widx);
}
template <int U, class Ar> HE_HD void row_fwd_A(const Ar &ar, typename Ar::T (*x)[kRowE], const Tw16 w[kTwA])
{
    row_fwd_stage<U, 3>(ar, x, w, [](int r) { return 0 + (r >> 4); });
    row_fwd_stage<U, 2>(ar, x, w, [](int r) { return 1 + (r >> 3); });
    row_fwd_stage<U, 1>(ar, x, w, [](int r) { return 3 + (r >> 2); });
    row_fwd_stage<U, 0>(ar, x, w, [](int r) { return 7 + (r >> 1); });
}
template <int U, class Ar> HE_HD void row_fwd_B(const Ar &ar, typename Ar::T (*x)[kRowE], const Tw16 w[kTwB])
{
    row_fwd_stage<U, 3>(ar, x, w, [](int r) { return 0 + (r >> 4); });
    row_fwd_stage<U, 2>(ar, x, w, [](int r) { return 1 + (r >> 3); });
    row_fwd_stage<U, 1>(ar, x, w, [](int r) { return 3 + (r >> 2); });
    row_fwd_stage<U, 0>(ar, x, w, [](int r) { return 7 + (r >> 1); });
}
template <int U, class Ar> HE_HD void row_fwd_C(const Ar &ar, typename Ar::T (*x)[kRowE], const Tw16 w[kTwC])
{
    row_fwd_stage<U, 1>(ar, x, w, [](int r) { return r >> 2; });                                  // stage 8: pairs (r, r|2), twiddle of chunk c
    row_fwd_stage<U, 0>(ar, x, w, [](int r) { return 4 + 2 * (r >> 2) + ((r >> 1) & 1); });      // stage 9: pairs (r, r|1), twiddle (c, h = bit 1 of r)
}
// The same three phases over the wide lazy range of the u64 engine (ArU64::bfly_fwd_lazy; q < 2^60): in below 4q, A to 12q, B runs two
// stages (16q), comes back under 4q, runs two more (8q), C to 12q.  Fold build of the engine (stages of 3q): A to 16q, reduce, B to
// 14q, reduce, C to 8q.  The fp64 engine's instantiations are the plain phases.
// SW: w[] is wave-uniform (a caller that keeps phase A's twiddles in scalar registers)
template <int U, bool SW = false, class Ar> HE_HD void row_fwd_A_lazy(const Ar &ar, typename Ar::T (*x)[kRowE], const Tw16 w[kTwA])
{
    row_fwd_stage_x<U, 3, true, SW>(ar, x, w, [](int r) { return 0 + (r >> 4); });
    row_fwd_stage_x<U, 2, true, SW>(ar, x, w, [](int r) { return 1 + (r >> 3); });
    row_fwd_stage_x<U, 1, true, SW>(ar, x, w, [](int r) { return 3 + (r >> 2); });
    row_fwd_stage_x<U, 0, true, SW>(ar, x, w, [](int r) { return 7 + (r >> 1); });
}
template <int U, class Ar> HE_HD void row_fwd_lazy_reduce_all(const Ar &ar, typename Ar::T (*x)[kRowE])
{
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < kRowE; ++r) x[u][r] = ar.lazy_reduce(x[u][r]);
}
template <int U, class Ar> HE_HD void row_fwd_B_lazy(const Ar &ar, typename Ar::T (*x)[kRowE], const Tw16 w[kTwB])
{
    if constexpr (Ar::kFold) row_fwd_lazy_reduce_all<U>(ar, x); // fold build: below 16q -> below 2q + 16c, then four stages of 3q
    row_fwd_stage_x<U, 3, true>(ar, x, w, [](int r) { return 0 + (r >> 4); });
    row_fwd_stage_x<U, 2, true>(ar, x, w, [](int r) { return 1 + (r >> 3); });
    if constexpr (!Ar::kFold) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < kRowE; ++r) x[u][r] = ar.reduce16_to_4q(x[u][r]);
    }
    row_fwd_stage_x<U, 1, true>(ar, x, w, [](int r) { return 3 + (r >> 2); });
    row_fwd_stage_x<U, 0, true>(ar, x, w, [](int r) { return 7 + (r >> 1); });
}
template <int U, class Ar> HE_HD void row_fwd_C_lazy(const Ar &ar, typename Ar::T (*x)[kRowE], const Tw16 w[kTwC])
{
    if constexpr (Ar::kFold) row_fwd_lazy_reduce_all<U>(ar, x); // below 14q + 16c -> below 2q + 16c, two stages: below 8q + 16c
    row_fwd_stage_x<U, 1, true>(ar, x, w, [](int r) { return r >> 2; });
    row_fwd_stage_x<U, 0, true>(ar, x, w, [](int r) { return 4 + 2 * (r >> 2) + ((r >> 1) & 1); });
}
template <class Ar, class TW> HE_HD void row_fwd_A(const Ar &ar, typename Ar::T x[kRowE], const TW &tw)
{
    Tw16 w[kTwA];
    gather_A(tw, w);
    row_fwd_A<1>(ar, reinterpret_cast<typename Ar::T(*)[kRowE]>(x), w);
}
template <class Ar, class TW> HE_HD void row_fwd_B(const Ar &ar, typename Ar::T x[kRowE], const TW &tw, int lane)
{
    Tw16 w[kTwB];
    gather_B(tw, lane, w);
    row_fwd_B<1>(ar, reinterpret_cast<typename Ar::T(*)[kRowE]>(x), w);
}
template <class Ar, class TW> HE_HD void row_fwd_C(const Ar &ar, typename Ar::T x[kRowE], const TW &tw, int lane)
{
    Tw16 w[kTwC];
    gather_C(tw, lane, w);
    row_fwd_C<1>(ar, reinterpret_cast<typename Ar::T(*)[kRowE]>(x), w);
}

// ---------------------------------------------------------------------------------------------------
// Row pass, inverse: stages 9,8 (layout C), 7..4 (layout B), 3..0 (layout A)
// ---------------------------------------------------------------------------------------------------
template <class Ar, class TW> HE_HD void row_inv_C(const Ar &ar, typename Ar::T x[kRowE], const TW &itw, int lane)
{
#pragma unroll
    for (int s = 9; s >= 8; --s) {
        const int bit = 9 - s;
        Tw16 w[8];
        int n = 0;
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            if (s == 8 && (r & 1)) continue;
            w[n++] = itw.get(s, (u32)elemC(lane, r) >> (10 - s));
        }
        n = 0;
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            ar.bfly_inv(x[r], x[r | (1 << bit)], w[s == 8 ? (n++ >> 1) : n++]);
        }
    }
}
// Phase C of the inverse row pass on twiddles gathered up front (gather_inv_C: stage 9's eight entries, then stage 8's four -- the order
// row_inv_C reads them in): a kernel whose row arrives behind other work requests them together with the row (he355_kernels_lds.hip).
constexpr int kTwInvC = 12;
template <class TW> HE_HD void gather_inv_C(const TW &itw, int lane, Tw16 w[kTwInvC])
{
    int n = 0;
#pragma unroll
    for (int r = 0; r < kRowE; ++r) {
        if (r & 1) continue;
        w[n++] = itw.get(9, (u32)elemC(lane, r) >> 1);
    }
#pragma unroll
    for (int r = 0; r < kRowE; ++r) {
        if ((r & 2) || (r & 1)) continue;
        w[n++] = itw.get(8, (u32)elemC(lane, r) >> 2);
    }
}
template <class Ar> HE_HD void row_inv_C_w(const Ar &ar, typename Ar::T x[kRowE], const Tw16 w[kTwInvC])
{
    int n = 0;
#pragma unroll
    for (int r = 0; r < kRowE; ++r) {
        if (r & 1) continue;
        ar.bfly_inv(x[r], x[r | 1], w[n++]);
    }
    n = 0;
#pragma unroll
    for (int r = 0; r < kRowE; ++r) {
        if (r & 2) continue;
        ar.bfly_inv(x[r], x[r | 2], w[8 + (n++ >> 1)]);
    }
}
// Phases B and A of the inverse row pass on twiddles gathered up front (as the forward phases take them): a kernel issues the loads
// BEFORE the LDS exchange that precedes the phase, so they land while the exchange is in flight instead of behind it.  wb: stage s
// (7..4) entry g at (1 << (s - 4)) - 1 + g; wa: stage s (3..0) entry g at (1 << s) - 1 + g (lane-uniform: scalar loads through ctw_t).
constexpr int kTwInvB = 15, kTwInvA = 15;
template <class TW> HE_HD void gather_inv_B(const TW &itw, int lane, Tw16 w[kTwInvB])
{
    const u32 hi4 = (u32)lane >> 2;
#pragma unroll
    for (int s = 4; s < 8; ++s)
#pragma unroll
        for (int g = 0; g < (1 << (s - 4)); ++g) w[(1 << (s - 4)) - 1 + g] = itw.get(s, (hi4 << (s - 4)) | (u32)g);
}
template <class TW> HE_HD void gather_inv_A(const TW &itw, Tw16 w[kTwInvA])
{
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < (1 << s); ++g) w[(1 << s) - 1 + g] = itw.get(s, (u32)g);
}
template <class Ar> HE_HD void row_inv_B_w(const Ar &ar, typename Ar::T x[kRowE], const Tw16 w[kTwInvB])
{
    if (Ar::kNeedsRenormInv) {
#pragma unroll
        for (int r = 0; r < kRowE; ++r) x[r] = ar.renorm(x[r]);
    }
#pragma unroll
    for (int s = 7; s >= 4; --s) {
        const int bit = 7 - s;
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            ar.bfly_inv(x[r], x[r | (1 << bit)], w[(1 << (s - 4)) - 1 + (r >> (8 - s))]);
        }
    }
}
template <class Ar, bool LAST> HE_HD void row_inv_A_w(const Ar &ar, typename Ar::T x[kRowE], const Tw16 w[kTwInvA], const Tw16 &w0_scaled)
{
    if (Ar::kNeedsRenormInv) {
#pragma unroll
        for (int r = 0; r < kRowE; ++r) x[r] = ar.renorm(x[r]);
    }
#pragma unroll
    for (int s = 3; s >= 0; --s) {
        const int bit = 3 - s;
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            if (LAST && s == 0) ar.template bfly_inv_last<true>(x[r], x[r | (1 << bit)], w0_scaled);
            else ar.template bfly_inv<true>(x[r], x[r | (1 << bit)], w[(1 << s) - 1 + (r >> (4 - s))]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Column pass: LOGN1 stages over N1 = 2^LOGN1 registers (row index a = register index)
// ---------------------------------------------------------------------------------------------------
template <class Ar, int LOGN1, class TW> HE_HD void col_fwd(const Ar &ar, typename Ar::T x[1 << LOGN1], TW tw)
{
    constexpr int N1 = 1 << LOGN1;
    if constexpr (Ar::kFold) {
        // Fold build of the u64 engine: the wide lazy butterfly (no conditional subtraction, the five-multiplier product; a stage adds at
        // most 3q).  In below 4q: four stages reach 16q <= 2^64 - 16c, so a fifth stage is preceded by lazy_reduce (three instructions
        // per element); one more lazy_reduce leaves every output below 2q + 16c -- inside the [0, 4q) every consumer of a column pass reads.
#pragma unroll
        for (int s = 0; s < LOGN1; ++s) {
            if (s == 4) {
#pragma unroll
                for (int a = 0; a < N1; ++a) x[a] = ar.lazy_reduce(x[a]);
            }
            const int gap = N1 >> (s + 1);
#pragma unroll
            for (int a = 0; a < N1; ++a) {
                if (a & gap) continue;
                const Tw16 w = tw_load(tw, (u32)((1 << s) + (a / (2 * gap))));
                ar.bfly_fwd_lazy(x[a], x[a + gap], w);
            }
        }
        if (LOGN1 > 0) {
#pragma unroll
            for (int a = 0; a < N1; ++a) x[a] = ar.lazy_reduce(x[a]);
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < LOGN1; ++s) {
        const int gap = N1 >> (s + 1);
#pragma unroll
        for (int a = 0; a < N1; ++a) {
            if (a & gap) continue;
            const Tw16 w = tw_load(tw, (u32)((1 << s) + (a / (2 * gap))));
            ar.template bfly_fwd<TwIsUniform<TW>::value>(x[a], x[a + gap], w);
        }
    }
}
// The fp64 engine's forward column pass as the digit-lift kernels run it: same butterflies in the same order as col_fwd<ArF64>, with
//   * the twiddles as bare doubles (PrimeDev::colw: 8 bytes per entry in scalar registers instead of 16),
//   * the first stage reading its inputs through in(a) (the lifted source column is kept for the next target prime, so the
//     stage writes x out of place instead of transforming a copy),
//   * BIAS: the last stage adds `bias` to both outputs, (X + bias) +- t.  With bias = kPackBias (he355_kernels.hip) and
//     |X +- t| < 2^47 every sum is an integer in [2^52, 2^53), hence exact, and the outputs are the 48-bit row patterns themselves:
//     one addition per butterfly instead of one per element.
template <int LOGN1, bool BIAS, class In, class TW> HE_HD void col_fwd_w(const ArF64 &ar, In in, double x[1 << LOGN1], TW cw, double bias)
{
    constexpr int N1 = 1 << LOGN1;
    if (LOGN1 == 0) {
        x[0] = BIAS ? in(0) + bias : in(0);
        return;
    }
#pragma unroll
    for (int s = 0; s < LOGN1; ++s) {
        const int gap = N1 >> (s + 1);
#pragma unroll
        for (int a = 0; a < N1; ++a) {
            if (a & gap) continue;
            const double w = cw[(1 << s) + (a / (2 * gap))];
            const double X = s == 0 ? in(a) : x[a];
            const double Y = s == 0 ? in(a + gap) : x[a + gap];
            const double t = ar.mulmod_vv(Y, w);
            if (BIAS && s == LOGN1 - 1) {
                const double xb = X + bias;
                x[a] = xb + t;
                x[a + gap] = xb - t;
            } else {
                x[a] = X + t;
                x[a + gap] = X - t;
            }
        }
    }
}

// inverse column pass; the very last stage (s == 0) folds N^-1: w0_scaled = itw[1] * N^-1
template <class Ar, int LOGN1, class TW> HE_HD void col_inv(const Ar &ar, typename Ar::T x[1 << LOGN1], TW itw, const Tw16 &w0_scaled)
{
    constexpr int N1 = 1 << LOGN1;
    if (Ar::kNeedsRenormInv) {
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = ar.renorm(x[a]);
    }
#pragma unroll
    for (int s = LOGN1 - 1; s >= 0; --s) {
        const int gap = N1 >> (s + 1);
#pragma unroll
        for (int a = 0; a < N1; ++a) {
            if (a & gap) continue;
            if (s == 0) {
                ar.template bfly_inv_last<TwIsUniform<TW>::value>(x[a], x[a + gap], w0_scaled);
            } else {
                const Tw16 w = tw_load(itw, (u32)((1 << s) + (a / (2 * gap))));
                ar.template bfly_inv<TwIsUniform<TW>::value>(x[a], x[a + gap], w);
            }
        }
    }
}

} // namespace he355
