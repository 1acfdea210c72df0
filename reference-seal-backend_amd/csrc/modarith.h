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

// One twiddle-table entry is 16 bytes for both engines (one dwordx4 / one s_load_dwordx4).
struct alignas(16) Tw16 {
    u64 a, b;
};

// ====================================================================================================
// ArU64 — Harvey lazy butterflies, values in [0,4q) forward / [0,2q) inverse
// ====================================================================================================
struct ArU64 {
    typedef u64 T;
    u64 q, two_q;
    u64 ninv, ninv_q; // N^-1 and its Shoup quotient (inverse transform scaling)
    u64 cr0, cr1;     // Barrett constant floor(2^128/q)

    HE_HD ModU64 mod() const { ModU64 m; m.q = q; m.cr0 = cr0; m.cr1 = cr1; return m; }
    // ---- dyadic domain: canonical in, canonical out ----
    typedef u64 Acc;
    HE_HD T dy_in(u64 c) const { return c; }
    HE_HD T dy_mul(T x, T y) const { return barrett128((u128)x * y, mod()); }
    HE_HD T dy_add(T x, T y) const { return addmod(x, y, q); }
    HE_HD u64 dy_out(T x) const { return x; }
    HE_HD T key_in(u64 bits) const { return bits; }
    // acc += x*key, x lazy (< 4q), key canonical, keyq = its Shoup quotient floor(key*2^64/q) or up to 2 below it
    // (shoup_quotient_est).  The product term lands in [0,3q), the sum stays in [0,4q): 7q < 2^63 for q < 2^60.
    static constexpr bool kKeyQuotient = true;
    HE_HD void acc_mac(Acc &acc, T x, T key, u64 keyq) const
    {
        const u64 s = acc + mul_shoup_lazy_uq(x, key, keyq, q);
        acc = s >= 2 * two_q ? s - 2 * two_q : s;
    }
    // The same without the conditional subtraction, for runs of products between two acc_reduce calls.  With the EXACT quotient
    // keyq = floor(key * 2^64 / q) (shoup_quotient) the product of ANY 64-bit x lands in [0, 2q): an accumulator below 4q takes
    // kAccRun = 6 products (4q + 12q = 16q <= 2^64 for q < 2^60) before acc_reduce brings it back under 4q -- two conditional
    // subtractions per six products instead of six.
    static constexpr int kAccRun = 6;
    HE_HD void acc_mac_lazy(Acc &acc, T x, T key, u64 keyq) const
    {
        const u64 v = mul_shoup_lazy_uq(x, key, keyq, q);
#if defined(HE355_LANE_SIM)
        if (acc > ~(u64)0 - v) he355_sim_overflow = 1;
#endif
        acc += v;
    }
    HE_HD Acc acc_reduce(Acc acc) const { return reduce16_to_4q(acc); }
    HE_HD u64 acc_canon(Acc acc) const { return to_canon(acc); }
    HE_HD Acc acc_from_canon(u64 v) const { return v; }
    HE_HD Acc acc_from_lazy(T v) const { return v; } // a dy_mul result (canonical here) as the start of a sum
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
        u64 r = mul_shoup(t + 2 * two_q - x, inv, inv_shoup, q);
        return addmod(r, addend, q);
    }

    // two floor steps with one combined correction x: ((t * inv1 + addend) - x) * inv2; t, addend canonical, x lazy < 4q
    template <class FC> HE_HD u64 floor_fin2(u64 t, T x, const FC &f1, const FC &f2, u64 addend) const
    {
        const u64 r1 = mul_shoup(t, f1.inv, f1.inv_shoup, q);
        return mul_shoup(r1 + addend + 2 * two_q - x, f2.inv, f2.inv_shoup, q);
    }
    // The same two floor steps for sums that were formed with key residues scaled by s^-1 (k_k3<TENSOR>: t stands for t * inv already,
    // and the addend is inside it): t - x * inv, and (t - x) * inv2.  t canonical, x lazy < 4q.
    HE_HD u64 floor_fin_s(u64 t, T x, u64 inv, u64 inv_shoup, double, double) const { return submod(t, mul_shoup(x, inv, inv_shoup, q), q); }
    template <class FC> HE_HD u64 floor_fin2_s(u64 t, T x, const FC &f2) const { return mul_shoup(t + 2 * two_q - x, f2.inv, f2.inv_shoup, q); }
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
        u64 v = mul_shoup_lazy_uq(Y, w.a, w.b, q);
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
    HE_HD void bfly_fwd_lazy(T &X, T &Y, const Tw16 &w) const
    {
        const u64 v = mul_shoup_lazy_uq(Y, w.a, w.b, q);
#if defined(HE355_LANE_SIM)
        if (X > ~(u64)0 - v || X + two_q < v) he355_sim_overflow = 1;
#endif
        const u64 x = X;
        X = x + v;
        Y = x + two_q - v;
    }
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
        return mul_shoup_lazy_uq(x, w, wq, q);
    }
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
    // scale by N^-1 only (N1 == 1 rings have no column pass)
    HE_HD T scale_ninv(T x) const { return mul_shoup_lazy_uq(x, ninv, ninv_q, q); }
    // bring a forward-lazy value into the inverse-lazy range (and vice versa these are no-ops)
    HE_HD T renorm(T x) const { return x >= two_q ? x - two_q : x; }
    static constexpr bool kNeedsRenormInv = false;
};

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
