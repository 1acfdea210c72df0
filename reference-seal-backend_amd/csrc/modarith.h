// modarith.h — modular arithmetic for RNS residues on CDNA4 (gfx950).
//
// Two arithmetic engines, both producing the same canonical residues:
//   ArU64 : 64-bit integer Harvey/Shoup lazy arithmetic for any prime < 2^61 (v_mad_u64_u32 chains).
//           Measured on MI355X: 1.55 T butterflies/s (profiles/r01_alu_rates_mi355x.txt).
//   ArF64 : exact integer arithmetic carried in fp64 for primes < 2^47 (the 40/45-bit primes the
//           reference's parameter rule produces, seal_context.cpp:79-82).  v_fma_f64 / v_mul_f64 /
//           v_rndne_f64 are full rate on MI355X; a butterfly is 8 fp64 instructions: 4.33 T butterflies/s.
//           Every intermediate is an integer of magnitude < 2^53, every product is split exactly with
//           one FMA, so results are bit-exact — fp64 is the ALU, not a precision choice.
//
// The functions are plain inline code with no HIP dependence so that tests/csim can run the identical
// lane program on the CPU (test-only lane simulator); under hipcc they are __host__ __device__.
#pragma once
#include <cstdint>

#if defined(__HIP__)
#define HE_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define HE_HD inline
#endif

// Scheduling fence for hand-interleaved instruction groups (device code only): the machine scheduler may not move
// anything across it, so independent operations written next to each other stay next to each other.
#if defined(__HIP_DEVICE_COMPILE__)
#define HE_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define HE_SCHED_FENCE() ((void)0)
#endif

namespace he355 {

#if defined(HE355_LANE_SIM)
extern int he355_sim_overflow; // set by the lane simulator's build of the wide-lazy butterflies when a sum leaves 64 bits
#endif

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

HE_HD u64 mulhi64(u64 a, u64 b) { return (u64)(((u128)a * b) >> 64); }

// ---- Barrett constants for a modulus (floor(2^128/q) as two words) ---------------------------------
struct ModU64 {
    u64 q;
    u64 cr0, cr1; // floor(2^128/q) low/high
};

// x < 2^64 -> [0,q)
HE_HD u64 barrett64(u64 x, const ModU64 &m)
{
    u64 t = mulhi64(x, m.cr1);
    u64 r = x - t * m.q;
    return r >= m.q ? r - m.q : r;
}
// x < 2^128 -> [0,q)
HE_HD u64 barrett128(u128 x, const ModU64 &m)
{
    u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    u64 carry = mulhi64(x0, m.cr0);
    u128 t2 = (u128)x0 * m.cr1;
    u128 s = (u128)(u64)t2 + carry;
    u64 tmp1 = (u64)s;
    u64 tmp3 = (u64)(t2 >> 64) + (u64)(s >> 64);
    t2 = (u128)x1 * m.cr0;
    s = (u128)tmp1 + (u64)t2;
    carry = (u64)(t2 >> 64) + (u64)(s >> 64);
    u64 quo = x1 * m.cr1 + tmp3 + carry;
    u64 r = x0 - quo * m.q;
    return r >= m.q ? r - m.q : r;
}
HE_HD u64 mulmod(u64 a, u64 b, const ModU64 &m) { return barrett128((u128)a * b, m); }
HE_HD u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
HE_HD u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
// x * w mod q with Shoup quotient wq = floor(w*2^64/q); result in [0,2q)
HE_HD u64 mul_shoup_lazy(u64 x, u64 w, u64 wq, u64 q) { return w * x - mulhi64(x, wq) * q; }
// the same for a wave-uniform q (a hand-written v_mad_u64_u32 chain was tried here and lost: HISTORY.md, tools/gen_shoup_asm.py)
HE_HD u64 mul_shoup_lazy_uq(u64 x, u64 w, u64 wq, u64 q) { return w * x - mulhi64(x, wq) * q; }
HE_HD u64 mul_shoup(u64 x, u64 w, u64 wq, u64 q)
{
    u64 r = mul_shoup_lazy(x, w, wq, q);
    return r >= q ? r - q : r;
}

// ---- fold reduction for q = 2^60 - c (round 5) -----------------------------------------------------------------------------------
// Every 60-bit prime the reference's parameter rule produces (seal_context.cpp:79-82,107-110: the largest primes 1 (mod 2N) below
// 2^60) is 2^60 - c with c < 2^24, so 2^60 == c (mod q) and a product is reduced by folding its high part back in -- no quotient.
// A constant w travels with w2 = w * 2^32 mod q (the 16-byte entry that holds {w, Shoup quotient} for other primes):
//     x * w == x0 * w + x1 * w2 = S < 2^93      (x = x1 2^32 + x0 ANY 64-bit value: four 32 x 32 -> 64 multiply-adds)
//     tight: r = lo60(S) + (S >> 61) * 2c + bit60(S) * c   < 2^60 + 2^33 c   (< 1.13 q for c < 2^24; 6 multiplier ops)
//     wide : r = lo61(S) + (S >> 61) * 2c                  < 2^61 + 2^33 c   (< 2.13 q; 5 multiplier ops)
// against 10 multiplier ops of x * w - floor(x * wq / 2^64) * q.  Measured (tools/micro/u64_fold.hip, profiles/r05_micro_u64_fold.txt):
// the wide lazy butterfly 26.6 ns against Shoup's 40.7 per wave and SIMD slot.  A context whose u64-engine primes all qualify
// (Params::u64_fold) runs the HE355_U64_FOLD=1 build of the kernels (Makefile: every device file is compiled for both forms).
constexpr u64 kFoldMask60 = 0x0FFFFFFFFFFFFFFFull, kFoldMask61 = 0x1FFFFFFFFFFFFFFFull;
constexpr u32 kFoldMaxC = 1u << 26; // c below this: every bound used here holds with room (see the bounds at each function)
HE_HD bool fold_prime_ok(u64 q) { return (q >> 59) == 1 && (q >> 60) == 0 && ((u64)1 << 60) - q < kFoldMaxC; }
HE_HD u32 fold_c(u64 q) { return (u32)(((u64)1 << 60) - q); }
// S = x0 * w + x1 * w2 as (up, low 32 bits): S = up * 2^32 + lo
HE_HD u64 fold_sum(u64 x, u64 w, u64 w2, u32 &lo)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    const u64 u = (u64)x0 * (u32)w;
    const u64 u2 = (u64)x1 * (u32)w2 + (u32)u;
    const u64 m1 = (u64)x0 * (u32)(w >> 32) + (u >> 32);
    const u64 m2 = (u64)x1 * (u32)(w2 >> 32) + (u2 >> 32);
    lo = (u32)u2;
    return m1 + m2; // < 2^61 + 2^33 (w, w2 < 2^60)
}
// x * w mod q in [0, 2^60 + 2^33 c), any 64-bit x; w, w2 = w 2^32 mod q canonical
HE_HD u64 fold_mul_tight(u64 x, u64 w, u64 w2, u32 c)
{
    u32 lo;
    const u64 up = fold_sum(x, w, w2, lo);
    const u32 h = (u32)(up >> 29), b = (u32)(up >> 28) & 1u; // S >> 61 (S < 2^93: fits 32 bits) and bit 60 of S
    const u64 lo60 = ((up & 0x0FFFFFFFull) << 32) | lo;
    return (u64)h * (2 * c) + ((u64)b * c + lo60);
}
// ... in [0, 2^61 + 2^33 c): bit 60 stays in the low part
HE_HD u64 fold_mul_wide(u64 x, u64 w, u64 w2, u32 c)
{
    u32 lo;
    const u64 up = fold_sum(x, w, w2, lo);
    const u32 h = (u32)(up >> 29);
    const u64 lo61 = ((up & 0x1FFFFFFFull) << 32) | lo;
    return (u64)h * (2 * c) + lo61;
}
// any 64-bit x -> the same residue below 2^61 + 14 c (three instructions: the top three bits folded back in)
HE_HD u64 fold_red2q(u64 x, u32 c) { return (x & kFoldMask61) + (u64)(u32)(x >> 61) * (2 * c); }
// any 128-bit sum -> canonical residue
HE_HD u64 fold_acc128(u128 acc, u64 q, u32 c)
{
    const u64 lo = (u64)acc & kFoldMask60;
    const u128 hc = (acc >> 60) * c;                   // < 2^68 c  (< 2^94)
    const u64 l2 = (u64)hc & kFoldMask60, h2 = (u64)(hc >> 60); // h2 < 2^34
    u64 r = h2 * c + lo + l2;                          // < 2^60 + 2^61
    r = (r & kFoldMask60) + (r >> 60) * c;             // < 2^60 + 3 c < 2 q
    return r >= q ? r - q : r;
}

// One twiddle-table entry is 16 bytes for both engines (one dwordx4 / one s_load_dwordx4).
struct alignas(16) Tw16 {
    u64 a, b;
};

#ifndef HE355_U64_FOLD
#define HE355_U64_FOLD 0
#endif
// Everything whose BODY depends on HE355_U64_FOLD lives in an inline namespace named after the form: the two builds of the device code and
// the host library (default form) are linked into one shared object, and without it `he355::mul_pre_lazy`, `he355::ArU64::bfly_inv` ... would
// be one mangled name with two definitions -- harmless while only device code (per translation unit) uses them, a silent wrong-residue bug
// the day a host function odr-uses one.  Name lookup is unchanged: `he355::ArU64` finds the form of the including build.
#if HE355_U64_FOLD
#define HE355_ARITH_NS u64_fold_form
#else
#define HE355_ARITH_NS u64_shoup_form
#endif
inline namespace HE355_ARITH_NS {
// x * w mod q, lazy in [0, 2q) / canonical, for a constant w with its companion word w2: the Shoup quotient floor(w 2^64 / q), or
// (HE355_U64_FOLD build: every u64-engine prime is 2^60 - c) w * 2^32 mod q.  Which one a table holds is the context's choice
// (Params::u64_fold), and a context only ever runs the build that matches its tables.
#if HE355_U64_FOLD
HE_HD u64 mul_pre_lazy(u64 x, u64 w, u64 w2, u64 q) { return fold_mul_tight(x, w, w2, fold_c(q)); }
#else
HE_HD u64 mul_pre_lazy(u64 x, u64 w, u64 w2, u64 q) { return mul_shoup_lazy(x, w, w2, q); }
#endif
HE_HD u64 mul_pre(u64 x, u64 w, u64 w2, u64 q)
{
    const u64 r = mul_pre_lazy(x, w, w2, q);
    return r >= q ? r - q : r;
}
// the companion word of a canonical constant (host side: tables, floor constants)
HE_HD u64 pre_word(u64 w, u64 q, bool fold) { return fold ? (u64)(((u128)w << 32) % q) : (u64)(((u128)w << 64) / q); }

// ====================================================================================================
// ArU64 — Harvey lazy butterflies, values in [0,4q) forward / [0,2q) inverse
// Two builds of the same engine (HE355_U64_FOLD): products by constants through Shoup quotients (any prime below 2^61), or through the
// fold reduction above (primes 2^60 - c).  Every bound of the Shoup form holds for the fold form as it stands -- its lazy product is
// below 1.13 q where Shoup's is below 2q -- except in the two places that are written for it: the wide lazy row pass (fold_mul_wide,
// below 2.13 q, offsets of 3q, fold_red2q between the phases) and the key multiply-accumulate (128-bit sums, one reduction per 16).
// ====================================================================================================
struct ArU64 {
    typedef u64 T;
    u64 q, two_q;
    u64 ninv, ninv_q; // N^-1 and its companion word (inverse transform scaling)
    u64 cr0, cr1;     // Barrett constant floor(2^128/q)
    static constexpr bool kFold = HE355_U64_FOLD != 0;
    HE_HD u32 c() const { return fold_c(q); }

    HE_HD ModU64 mod() const { ModU64 m; m.q = q; m.cr0 = cr0; m.cr1 = cr1; return m; }
    // ---- dyadic domain: canonical in, canonical out ----
    HE_HD T dy_in(u64 c) const { return c; }
    HE_HD T dy_mul(T x, T y) const { return barrett128((u128)x * y, mod()); }
    HE_HD T dy_add(T x, T y) const { return addmod(x, y, q); }
    HE_HD u64 dy_out(T x) const { return x; }
    HE_HD T key_in(u64 bits) const { return bits; }
    typedef u64 Acc;
#if HE355_U64_FOLD
    // Key products through the wide fold product (below 2^61 + 2^33 c <= 2.5 * 2^60), the key's companion word key * 2^32 mod q where the
    // Shoup build keeps the key's quotient: a sum that starts below 2^61 + 14c takes kAccRun = 5 products (2 + 12.5 < 16, times 2^60)
    // before acc_reduce (three instructions) brings it back.  (128-bit sums reduced once per 16 products cost the same instructions
    // and no companion table, but 64 more registers per wave: 95-235 spilled registers in the 8-wave k_k3 -- built and dropped.)
    static constexpr bool kKeyQuotient = true;
    static constexpr int kAccRun = 5;
    HE_HD void acc_mac_lazy(Acc &acc, T x, T key, u64 key2) const
    {
        const u64 v = fold_mul_wide(x, key, key2, c());
#if defined(HE355_LANE_SIM)
        if (acc > ~(u64)0 - v) he355_sim_overflow = 1;
#endif
        acc += v;
    }
    HE_HD void acc_mac(Acc &acc, T x, T key, u64 key2) const { acc = fold_red2q(acc + fold_mul_wide(x, key, key2, c()), c()); } // acc below 2^61 + 14c stays there
    HE_HD Acc acc_reduce(Acc acc) const { return fold_red2q(acc, c()); }
    HE_HD u64 acc_canon(Acc acc) const { return to_canon(fold_red2q(acc, c())); } // below 2q + 16c < 4q
    HE_HD Acc acc_from_canon(u64 v) const { return v; }
    HE_HD Acc acc_from_lazy(T v) const { return v; }
#else
    // acc += x*key, x lazy (< 4q), key canonical, keyq = its Shoup quotient floor(key*2^64/q) or up to 2 below it
    // (shoup_quotient_est).  The product term lands in [0,3q), the sum stays in [0,4q): 7q < 2^63 for q < 2^60.
    static constexpr bool kKeyQuotient = true;
    HE_HD void acc_mac(Acc &acc, T x, T key, u64 keyq) const
    {
        const u64 s = acc + mul_pre_lazy(x, key, keyq, q);
        acc = s >= 2 * two_q ? s - 2 * two_q : s;
    }
    // The same without the conditional subtraction, for runs of products between two acc_reduce calls.  With the EXACT quotient
    // keyq = floor(key * 2^64 / q) (shoup_quotient) the product of ANY 64-bit x lands in [0, 2q): an accumulator below 4q takes
    // kAccRun = 6 products (4q + 12q = 16q <= 2^64 for q < 2^60) before acc_reduce brings it back under 4q -- two conditional
    // subtractions per six products instead of six.
    static constexpr int kAccRun = 6;
    HE_HD void acc_mac_lazy(Acc &acc, T x, T key, u64 keyq) const
    {
        const u64 v = mul_pre_lazy(x, key, keyq, q);
#if defined(HE355_LANE_SIM)
        if (acc > ~(u64)0 - v) he355_sim_overflow = 1;
#endif
        acc += v;
    }
    HE_HD Acc acc_reduce(Acc acc) const { return reduce16_to_4q(acc); }
    HE_HD u64 acc_canon(Acc acc) const { return to_canon(acc); }
    HE_HD Acc acc_from_canon(u64 v) const { return v; }
    HE_HD Acc acc_from_lazy(T v) const { return v; } // a dy_mul result (canonical here) as the start of a sum
#endif
    // floor(w * 2^64 / q) from the Barrett constant floor(2^128/q) = cr1:cr0, at most 2 too small (never too large)
    HE_HD u64 shoup_quotient_est(u64 w) const { return w * cr1 + mulhi64(w, cr0); }
    // floor(w * 2^64 / q) exactly, w < q: the estimate, then the remainder w * 2^64 - est * q (below 3q, so its low 64 bits are all
    // of it) brought under q
    HE_HD u64 shoup_quotient(u64 w) const
    {
        u64 est = shoup_quotient_est(w);
        u64 r = 0 - est * q;
        if (r >= q) { r -= q; ++est; }
        if (r >= q) { r -= q; ++est; }
        return est;
    }
    // (t - x) * inv (+ addend): t, addend canonical, x lazy < 4q; inv given as Shoup pair
    HE_HD u64 floor_fin(u64 t, T x, u64 inv, u64 inv_shoup, double, double, u64 addend) const
    {
        u64 r = mul_pre(t + 2 * two_q - x, inv, inv_shoup, q);
        return addmod(r, addend, q);
    }

    // two floor steps with one combined correction x: ((t * inv1 + addend) - x) * inv2; t, addend canonical, x lazy < 4q
    template <class FC> HE_HD u64 floor_fin2(u64 t, T x, const FC &f1, const FC &f2, u64 addend) const
    {
        const u64 r1 = mul_pre(t, f1.inv, f1.inv_shoup, q);
        return mul_pre(r1 + addend + 2 * two_q - x, f2.inv, f2.inv_shoup, q);
    }
    // The same two floor steps for sums that were formed with key residues scaled by s^-1 (k_k3<TENSOR>: t stands for t * inv already,
    // and the addend is inside it): t - x * inv, and (t - x) * inv2.  t canonical, x lazy < 4q.
    HE_HD u64 floor_fin_s(u64 t, T x, u64 inv, u64 inv_shoup, double, double) const { return submod(t, mul_pre(x, inv, inv_shoup, q), q); }
    template <class FC> HE_HD u64 floor_fin2_s(u64 t, T x, const FC &f2) const { return mul_pre(t + 2 * two_q - x, f2.inv, f2.inv_shoup, q); }
    HE_HD T from_canon(u64 x) const { return x; }
    HE_HD T from_raw(u64 bits) const { return bits; }
    HE_HD u64 to_raw(T x) const { return x; }
    // [0,4q) -> [0,q)
    HE_HD u64 to_canon(T x) const
    {
        if (x >= two_q) x -= two_q;
        if (x >= q) x -= q;
        return x;
    }
    // forward (Cooley-Tukey): X,Y in [0,4q) -> [0,4q)
    // SW (here and below): the twiddle is wave-uniform and may stay in scalar registers (column passes: tw_load through ctw_t)
    template <bool SW = false> HE_HD void bfly_fwd(T &X, T &Y, const Tw16 &w) const
    {
        u64 u = X >= two_q ? X - two_q : X;
        u64 v = mul_pre_lazy(Y, w.a, w.b, q);
        X = u + v;
        Y = u + two_q - v;
    }
    // inverse (Gentleman-Sande): X,Y in [0,2q) -> [0,2q)
    // G independent butterflies (same meaning as G calls of bfly_fwd)
    template <int G> HE_HD void bfly_fwd_g(T (&X)[G], T (&Y)[G], const Tw16 (&w)[G]) const
    {
#pragma unroll
        for (int k = 0; k < G; ++k) bfly_fwd(X[k], Y[k], w[k]);
    }
    // ---- wide lazy range for the forward row pass of a key prime (q < 2^60, so 16 q <= 2^64) ------------------------------------
    // The Harvey butterfly above spends 4 of its ~28 instructions on bringing X under 2q first.  Without that step a stage takes
    // values below B to values below B + 2q (the Shoup product of ANY 64-bit Y lands in [0, 2q)); a row pass that enters below 4q
    // may run six such stages (16 q), is brought back under 4q once (two conditional subtractions per element), and runs its last
    // four stages to values below 12 q, which the key multiply-accumulate accepts as they are (acc_mac's bound holds for any 64-bit
    // x).  Per 1024-point row and lane: 80 x 4 instructions saved, 16 x 8 spent.
    // Fold build: the product is fold_mul_wide's, below 2^61 + 2^33 c <= 3q, so the offset is 3q and a stage adds at most 3q: a row
    // entering below 4q runs phase A's four stages to below 16q <= 2^64 - 16c, lazy_reduce (three instructions: the top three bits
    // folded back in) brings it below 2^61 + 14c before each of the next two phases, phase B ends below 14q + 16c and phase C below
    // 8q + 16c (ntt_core.h: row_fwd_B_lazy / row_fwd_C_lazy).
    HE_HD void bfly_fwd_lazy(T &X, T &Y, const Tw16 &w) const
    {
#if HE355_U64_FOLD
        const u64 v = fold_mul_wide(Y, w.a, w.b, c());
        const u64 off = two_q + q;
#else
        const u64 v = mul_shoup_lazy_uq(Y, w.a, w.b, q);
        const u64 off = two_q;
#endif
#if defined(HE355_LANE_SIM)
        if (X > ~(u64)0 - v || X > ~(u64)0 - off || X + off < v) he355_sim_overflow = 1;
#endif
        const u64 x = X;
        X = x + v;
        Y = x + off - v;
    }
    // between the phases of the wide lazy row pass (fold build; the Shoup build reduces once, inside phase B: reduce16_to_4q)
    HE_HD T lazy_reduce(T x) const { return fold_red2q(x, c()); }
    // SW: the twiddles are wave-uniform (they may stay in scalar registers)
    template <int G, bool SW = false> HE_HD void bfly_fwd_lazy_g(T (&X)[G], T (&Y)[G], const Tw16 (&w)[G]) const
    {
#pragma unroll
        for (int k = 0; k < G; ++k) bfly_fwd_lazy(X[k], Y[k], w[k]);
    }
    // [0, 16q) -> [0, 4q)
    HE_HD T reduce16_to_4q(T x) const
    {
        const u64 four_q = 2 * two_q, eight_q = 4 * two_q;
        if (x >= eight_q) x -= eight_q;
        if (x >= four_q) x -= four_q;
        return x;
    }
    // [0, 16q) -> [0, q)
    HE_HD u64 to_canon16(T x) const { return to_canon(reduce16_to_4q(x)); }
    template <bool SW = false> HE_HD u64 mul_tw(u64 x, u64 w, u64 wq) const
    {
        return mul_pre_lazy(x, w, wq, q);
    }
#if HE355_U64_FOLD
    // Fold build: values of the inverse transform stay below B = 2^61 + 2^33 c (<= 2.5 * 2^60 <= 3q - 2^58): the sum (below 2B < 2^64)
    // comes back under 2^61 + 14c by fold_red2q, the difference takes the offset 3q >= B and goes through the wide product (below B).
    // The last stage multiplies both outputs (tight products, below 1.5 * 2^60 < 2q), so what leaves a pass is as small as in the Shoup
    // form; what passes BETWEEN the row pass and the column pass (raw rows) is below B, which the next bfly_inv accepts.
    template <bool SW = false> HE_HD void bfly_inv(T &X, T &Y, const Tw16 &w) const
    {
#if defined(HE355_LANE_SIM)
        const u64 B = ((u64)1 << 61) + ((u64)c() << 33);
        if (X >= B || Y >= B) he355_sim_overflow = 1;
#endif
        const u64 s = X + Y;
        const u64 d = X + (two_q + q) - Y;
        X = fold_red2q(s, c());
        Y = fold_mul_wide(d, w.a, w.b, c());
    }
    template <bool SW = false> HE_HD void bfly_inv_last(T &X, T &Y, const Tw16 &w_scaled) const
    {
#if defined(HE355_LANE_SIM)
        const u64 B = ((u64)1 << 61) + ((u64)c() << 33);
        if (X >= B || Y >= B) he355_sim_overflow = 1;
#endif
        const u64 s = X + Y;
        const u64 d = X + (two_q + q) - Y;
        X = mul_tw<SW>(s, ninv, ninv_q);
        Y = mul_tw<SW>(d, w_scaled.a, w_scaled.b);
    }
#else
    template <bool SW = false> HE_HD void bfly_inv(T &X, T &Y, const Tw16 &w) const
    {
        u64 s = X + Y;
        u64 d = X + two_q - Y;
        X = s >= two_q ? s - two_q : s;
        Y = mul_tw<SW>(d, w.a, w.b);
    }
    // last inverse stage with N^-1 folded in: w must already be (w * N^-1) in Shoup form.  (N^-1 and w_scaled are per-prime
    // constants: wave-uniform wherever a wave works on one prime, which the asm path assumes of q anyway.)
    template <bool SW = false> HE_HD void bfly_inv_last(T &X, T &Y, const Tw16 &w_scaled) const
    {
        u64 s = X + Y;
        u64 d = X + two_q - Y;
        s = s >= two_q ? s - two_q : s;
        X = mul_tw<SW>(s, ninv, ninv_q);
        Y = mul_tw<SW>(d, w_scaled.a, w_scaled.b);
    }
#endif
    // scale by N^-1 only (N1 == 1 rings have no column pass)
    HE_HD T scale_ninv(T x) const { return mul_pre_lazy(x, ninv, ninv_q, q); }
    // bring a forward-lazy value into the inverse-lazy range (and vice versa these are no-ops)
    HE_HD T renorm(T x) const { return x >= two_q ? x - two_q : x; }
    static constexpr bool kNeedsRenormInv = false;
};

} // inline namespace HE355_ARITH_NS

// ====================================================================================================
// ArF64 — exact arithmetic in doubles, centred lazy residues, q < 2^47
// ====================================================================================================
HE_HD double u52_to_f64(u64 x) // exact for x < 2^52
{
    union { u64 u; double d; } c;
    c.u = x | 0x4330000000000000ull;
    return c.d - 4503599627370496.0;
}
HE_HD u64 f64_to_u52(double x) // exact for integer 0 <= x < 2^52
{
    union { u64 u; double d; } c;
    c.d = x + 4503599627370496.0;
    return c.u & 0x000FFFFFFFFFFFFFull;
}

struct ArF64 {
    typedef double T;
    double q, qinv;     // q and fl(1/q)
    double ninv, ninv_i; // N^-1 mod q (as double) and fl(ninv/q)

    HE_HD T from_canon(u64 x) const { return u52_to_f64(x); }
    HE_HD T from_raw(u64 bits) const
    {
        union { u64 u; double d; } c;
        c.u = bits;
        return c.d;
    }
    HE_HD u64 to_raw(T x) const
    {
        union { u64 u; double d; } c;
        c.d = x;
        return c.u;
    }
    // any integer |x| < 2^52 -> canonical [0,q), as a double / as an integer
    HE_HD T canon(T x) const
    {
        double c = __builtin_floor(x * qinv);
        double r = __builtin_fma(-c, q, x);
        if (r < 0.0) r += q;
        if (r >= q) r -= q;
        return r;
    }
    HE_HD u64 to_canon(T x) const { return f64_to_u52(canon(x)); }
    // y*w mod q, centred: |result| <= q*(1/2 + |y|*2^-51); tw.a = w, tw.b = fl(w/q) as doubles
    HE_HD T mulmod_c(T y, double w, double winv) const
    {
        double h = y * w;
        double l = __builtin_fma(y, w, -h);
        double c = __builtin_rint(y * winv);
        double d = __builtin_fma(-c, q, h);
        return d + l;
    }
    // x*y mod q for two variable operands (no precomputed quotient), centred; |x*y| < 2^100
    HE_HD T mulmod_vv(T x, T y) const
    {
        double h = x * y;
        double l = __builtin_fma(x, y, -h);
        double c = __builtin_rint(h * qinv);
        double d = __builtin_fma(-c, q, h);
        return d + l;
    }
    // G independent products x[k]*y[k] mod q, same results as G calls of mulmod_vv, issued level by level: a dependent fp64
    // instruction can issue ~11 cycles after its producer on MI355X (tools/micro/dp_latency.hip: one wave running one chain issues
    // every 5.5 ns, two interleaved chains every 2.9 ns, the pipe's rate is one per 2.0 ns), and the compiler's own order is one
    // serial chain per product, so with two or three waves per SIMD the chains must be interleaved by hand to fill the pipe.
    template <int G> HE_HD void mulmod_vv_g(const T (&x)[G], const T (&y)[G], T (&out)[G]) const
    {
        double h[G], l[G], c[G];
#pragma unroll
        for (int k = 0; k < G; ++k) h[k] = x[k] * y[k];
        HE_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G; ++k) c[k] = h[k] * qinv;
#pragma unroll
        for (int k = 0; k < G; ++k) l[k] = __builtin_fma(x[k], y[k], -h[k]);
        HE_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G; ++k) c[k] = __builtin_rint(c[k]);
        HE_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G; ++k) h[k] = __builtin_fma(-c[k], q, h[k]);
        HE_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G; ++k) out[k] = h[k] + l[k];
        HE_SCHED_FENCE();
    }
    // integer |x| < 2q -> canonical [0,q) as double
    HE_HD T canon2(T x) const
    {
        if (x < 0.0) x += q;
        if (x < 0.0) x += q;
        if (x >= q) x -= q;
        return x;
    }
    // canonical u64 of an arbitrary-size integer residue class is to_canon(); this one is for |x| < 2q
    HE_HD u64 to_canon2(T x) const { return f64_to_u52(canon2(x)); }
    // ---- dyadic domain ----
    typedef double Acc;
    HE_HD T dy_in(u64 c) const { return u52_to_f64(c); }
    HE_HD T dy_mul(T x, T y) const { return mulmod_vv(x, y); }
    HE_HD T dy_add(T x, T y) const { return x + y; }
    HE_HD u64 dy_out(T x) const { return to_canon2(x); }
    HE_HD T key_in(u64 bits) const { return from_raw(bits); }
    static constexpr bool kKeyQuotient = false;
    HE_HD void acc_mac(Acc &acc, T x, T key, u64) const { acc += mulmod_vv(x, key); }
    static constexpr int kAccRun = 1; // (the u64 engine's lazy runs: nothing to defer here)
    HE_HD void acc_mac_lazy(Acc &acc, T x, T key, u64) const { acc += mulmod_vv(x, key); }
    HE_HD Acc acc_reduce(Acc acc) const { return acc; }
    HE_HD u64 acc_canon(Acc acc) const { return to_canon(acc); }
    HE_HD Acc acc_from_canon(u64 v) const { return u52_to_f64(v); }
    HE_HD Acc acc_from_lazy(T v) const { return v; } // a dy_mul result (centred lazy) as the start of a sum
    HE_HD u64 floor_fin(u64 t, T x, u64, u64, double inv_d, double inv_i, u64 addend) const
    {
        double m = mulmod_c(u52_to_f64(t) - x, inv_d, inv_i);
        return to_canon2(m + u52_to_f64(addend));
    }
    // two floor steps with one combined correction x: ((t * inv1 + addend) - x) * inv2
    template <class FC> HE_HD u64 floor_fin2(u64 t, T x, const FC &f1, const FC &f2, u64 addend) const
    {
        const double m1 = mulmod_c(u52_to_f64(t), f1.inv_d, f1.inv_i);
        return to_canon2(mulmod_c(m1 + u52_to_f64(addend) - x, f2.inv_d, f2.inv_i));
    }
    // ... and for sums formed with key residues scaled by s^-1 (see ArU64): t - x * inv, (t - x) * inv2; |x| < q, so both stay inside canon2's range
    HE_HD u64 floor_fin_s(u64 t, T x, u64, u64, double inv_d, double inv_i) const { return to_canon2(u52_to_f64(t) - mulmod_c(x, inv_d, inv_i)); }
    template <class FC> HE_HD u64 floor_fin2_s(u64 t, T x, const FC &f2) const { return to_canon2(mulmod_c(u52_to_f64(t) - x, f2.inv_d, f2.inv_i)); }
    HE_HD static double tw_w(const Tw16 &t)
    {
        union { u64 u; double d; } c;
        c.u = t.a;
        return c.d;
    }
    HE_HD static double tw_wi(const Tw16 &t)
    {
        union { u64 u; double d; } c;
        c.u = t.b;
        return c.d;
    }
    // Butterflies use only w: the quotient estimate comes from h * (1/q) (mulmod_vv), so a twiddle is ONE double
    // in registers / LDS.  Same instruction count as the (w, w/q) form, same exactness, |t| <= q(1/2 + |Y| 2^-51).
    template <bool SW = false> HE_HD void bfly_fwd(T &X, T &Y, const Tw16 &w) const
    {
        double t = mulmod_vv(Y, tw_w(w));
        double x = X;
        X = x + t;
        Y = x - t;
    }
    // G independent butterflies, written level by level: every instruction's operands were produced G or more
    // instructions earlier, so one wave keeps the fp64 pipe busy without relying on other waves of the SIMD
    // (dependent fp64 operations issued back to back stall; the compiler's own order is one serial chain per butterfly).
    template <int G> HE_HD void bfly_fwd_g(T (&X)[G], T (&Y)[G], const Tw16 (&w)[G]) const
    {
        double h[G], l[G], c[G];
#pragma unroll
        for (int k = 0; k < G; ++k) h[k] = Y[k] * tw_w(w[k]);
        HE_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G; ++k) c[k] = h[k] * qinv;
#pragma unroll
        for (int k = 0; k < G; ++k) l[k] = __builtin_fma(Y[k], tw_w(w[k]), -h[k]);
        HE_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G; ++k) c[k] = __builtin_rint(c[k]);
        HE_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G; ++k) h[k] = __builtin_fma(-c[k], q, h[k]);
        HE_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G; ++k) l[k] = h[k] + l[k];
        HE_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G; ++k) {
            Y[k] = X[k] - l[k];
            X[k] = X[k] + l[k];
        }
        HE_SCHED_FENCE();
    }
    template <int G, bool SW = false> HE_HD void bfly_fwd_lazy_g(T (&X)[G], T (&Y)[G], const Tw16 (&w)[G]) const { bfly_fwd_g<G>(X, Y, w); } // the fp64 engine is lazy anyway
    HE_HD T reduce16_to_4q(T x) const { return x; }
    static constexpr bool kFold = false;
    HE_HD T lazy_reduce(T x) const { return x; }
    template <bool SW = false> HE_HD void bfly_inv(T &X, T &Y, const Tw16 &w) const
    {
        double s = X + Y;
        double d = X - Y;
        X = s;
        Y = mulmod_vv(d, tw_w(w));
    }
    template <bool SW = false> HE_HD void bfly_inv_last(T &X, T &Y, const Tw16 &w_scaled) const
    {
        double s = X + Y;
        double d = X - Y;
        X = mulmod_c(s, ninv, ninv_i);
        Y = mulmod_c(d, tw_w(w_scaled), tw_wi(w_scaled));
    }
    HE_HD T scale_ninv(T x) const { return mulmod_c(x, ninv, ninv_i); }
    // recentre to |x| <= q/2 (+1): needed on the sum path of the inverse transform, which doubles per stage
    HE_HD T renorm(T x) const { return __builtin_fma(-__builtin_rint(x * qinv), q, x); }
    static constexpr bool kNeedsRenormInv = true;
};

} // namespace he355
