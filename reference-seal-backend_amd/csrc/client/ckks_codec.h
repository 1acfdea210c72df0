// ckks_codec.h — floating-point half of the CKKS encoder / decoder as inline code shared by the host client and the device
// kernels (CKKSEncoder::encode / decode: seal_context.cpp:145-185 call sites; SEAL v3.7.2 ckks.h [UPSTREAM-UNVERIFIED]).
//
// encode: slots -> (place at the evaluation points) -> N-point complex DFT -> twist by zeta^-n, /N, *scale -> round
// decode: coefficients -> twist by zeta^n, /scale -> inverse DFT -> slots
// Every operation is written out (no std::complex, no contraction into FMAs: HE_FP_STRICT) and the twiddle tables are
// built once on the host and copied to the device, so both sides execute the same IEEE operations in the same order and
// produce the same bits.
#pragma once
#include "../modarith.h"

#if defined(__clang__)
#define HE_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define HE_FP_STRICT /* g++: built with -ffp-contract=off (tests/csim/Makefile) */
#endif

namespace he355 {
namespace client {

struct Cplx {
    double re, im;
};
HE_HD Cplx cmul(Cplx a, Cplx w)
{
    HE_FP_STRICT
    Cplx r;
    r.re = a.re * w.re - a.im * w.im;
    r.im = a.re * w.im + a.im * w.re;
    return r;
}
HE_HD Cplx cconj(Cplx a)
{
    Cplx r;
    r.re = a.re; r.im = -a.im;
    return r;
}
// radix-2 decimation-in-time butterfly: (a, b) <- (a + b*w, a - b*w)
HE_HD void fft_bfly(Cplx &a, Cplx &b, Cplx w)
{
    HE_FP_STRICT
    const Cplx v = cmul(b, w);
    const Cplx u = a;
    a.re = u.re + v.re; a.im = u.im + v.im;
    b.re = u.re - v.re; b.im = u.im - v.im;
}
// Stage tables: butterfly k of a stage of span `len` uses W[len/2 + k] = exp(-2 pi i k / len) (forward); the inverse
// transform uses the conjugates.  Twist table: Z[n] = exp(-i pi n / N).
// One stage of the iterative transform on bit-reversed input, butterfly index t in [0, N/2)
HE_HD void fft_stage_bfly(Cplx *z, const Cplx *W, u64 len, u64 t, bool inverse)
{
    const u64 half = len >> 1, k = t & (half - 1), i = ((t - k) << 1) + k;
    Cplx w = W[half + k];
    if (inverse) w = cconj(w);
    fft_bfly(z[i], z[i + half], w);
}
// encode tail: coefficient n from the transformed value; returns the rounded integer as a double (|.| < 9.2e18 checked by caller)
HE_HD double ckks_encode_coeff(Cplx zn, Cplx Zn, double N, double scale)
{
    HE_FP_STRICT
    const double re = zn.re * Zn.re - zn.im * Zn.im;
    const double c = re / N * scale;
    return __builtin_nearbyint(c);
}
// decode head: value v of coefficient n -> transform input
HE_HD Cplx ckks_decode_coeff(double v, Cplx Zn, double scale)
{
    HE_FP_STRICT
    const double x = v / scale;
    Cplx r;
    r.re = Zn.re * x;      // exp(+i pi n/N) = conj(Z[n])
    r.im = -Zn.im * x;
    return r;
}

HE_HD u32 bitrev_u32(u32 x, int bits)
{
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    x = (x >> 16) | (x << 16);
    return x >> (32 - bits);
}

} // namespace client
} // namespace he355
