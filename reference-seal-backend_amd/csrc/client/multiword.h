// multiword.h — fixed-capacity little-endian multiword integers for CRT composition (host client and device decode
// kernels share these inline functions).  Capacity kMwWords covers 16 data primes of 60 bits plus headroom.
#pragma once
#include "../modarith.h"

namespace he355 {
namespace client {

constexpr int kMwWords = 20;

HE_HD void mw_zero(u64 *x, int w)
{
    for (int i = 0; i < w; ++i) x[i] = 0;
}
HE_HD void mw_copy(u64 *x, const u64 *a, int w)
{
    for (int i = 0; i < w; ++i) x[i] = a[i];
}
HE_HD void mw_mul_small(u64 *x, int w, u64 b)
{
    u64 carry = 0;
    for (int i = 0; i < w; ++i) {
        const u128 p = (u128)x[i] * b + carry;
        x[i] = (u64)p;
        carry = (u64)(p >> 64);
    }
}
HE_HD void mw_add_mul(u64 *x, const u64 *a, int w, u64 b) // x += a * b
{
    u64 carry = 0;
    for (int i = 0; i < w; ++i) {
        const u128 p = (u128)a[i] * b + x[i] + carry;
        x[i] = (u64)p;
        carry = (u64)(p >> 64);
    }
}
HE_HD int mw_cmp(const u64 *a, const u64 *b, int w)
{
    for (int i = w; i-- > 0;)
        if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
    return 0;
}
HE_HD void mw_sub(u64 *x, const u64 *o, int w)
{
    u64 borrow = 0;
    for (int i = 0; i < w; ++i) {
        const u128 d = (u128)x[i] - o[i] - borrow;
        x[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}
HE_HD void mw_add(u64 *x, const u64 *o, int w)
{
    u64 carry = 0;
    for (int i = 0; i < w; ++i) {
        const u128 s = (u128)x[i] + o[i] + carry;
        x[i] = (u64)s;
        carry = (u64)(s >> 64);
    }
}
HE_HD double mw_to_double(const u64 *x, int w)
{
#if defined(__clang__)
    _Pragma("clang fp contract(off)") // same roundings on host and device
#endif
    double r = 0;
    for (int i = w; i-- > 0;) r = r * 18446744073709551616.0 + (double)x[i];
    return r;
}

// CRT tables of the first L primes as flat arrays (host builds them, the device gets a copy):
//   Q[words], halfQ[words], punct[L][words] = Q/q_i, inv[L] = (Q/q_i)^-1 mod q_i
struct CrtView {
    int L, words;
    const u64 *Q, *halfQ, *punct, *inv;
};
// x in [0, Q) from residues res[i * stride] (q: the primes' Barrett moduli)
HE_HD void crt_compose(const CrtView &c, const ModU64 *mods, const u64 *res, u64 stride, u64 *x)
{
    mw_zero(x, c.words);
    for (int i = 0; i < c.L; ++i) mw_add_mul(x, c.punct + (u64)i * c.words, c.words, barrett128((u128)res[(u64)i * stride] * c.inv[i], mods[i]));
    while (mw_cmp(x, c.Q, c.words) >= 0) mw_sub(x, c.Q, c.words);
}
// BFV Decryptor scale-and-round, exact form: round(t * x / Q) mod t for x in [0, Q)  (x is destroyed)
HE_HD u64 bfv_scale_round(const CrtView &c, u64 *x, u64 t, double Qd)
{
    u64 prod[kMwWords];
    mw_mul_small(x, c.words, t);
    mw_add(x, c.halfQ, c.words);
    u64 m = (u64)(mw_to_double(x, c.words) / Qd);
    if (m > 0) --m;
    for (;;) { // largest m with m*Q <= x
        mw_copy(prod, c.Q, c.words);
        mw_mul_small(prod, c.words, m + 1);
        if (mw_cmp(prod, x, c.words) <= 0) ++m;
        else break;
    }
    return m % t;
}

} // namespace client
} // namespace he355
