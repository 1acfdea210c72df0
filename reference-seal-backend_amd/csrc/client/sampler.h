// sampler.h — counter-based RLWE samplers shared by the host client and the device encryption kernels.
//
// Every random polynomial is a pure function of (seed, stream, coefficient index): coefficient n of stream s is drawn
// from the 64-bit word mix(seed, s, n).  The host (csrc/client/he_client.cpp) and the GPU (he355_client.hip) call the
// same inline functions, so a ciphertext encrypted on the device is bit-identical to the host's for the same
// (seed, stream) — that is what the parity tests compare.  Distributions follow SEAL v3.7.2 util/rlwe.cpp
// [UPSTREAM-UNVERIFIED]: sample_poly_ternary (uniform in {-1,0,1}) and sample_poly_cbd (centred binomial, 21 - 21 bits);
// SEAL's own generator (Blake2/SHAKE) is not reproduced — no two SEAL runs share randomness either.
#pragma once
#include "../modarith.h"

namespace he355 {
namespace client {

HE_HD u64 splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
HE_HD u64 sample_word(u64 seed, u64 stream, u64 n) { return splitmix64(splitmix64(seed ^ splitmix64(stream)) + n); }

// uniform in {-1, 0, 1}: the first 2-bit field of the word that is not 3 (0 -> -1, 1 -> 0, 2 -> 1); all 32 fields
// equal to 3 has probability 4^-32 and yields 0
HE_HD int sample_ternary_at(u64 seed, u64 stream, u64 n)
{
    u64 w = sample_word(seed, stream, n);
    for (int k = 0; k < 32; ++k, w >>= 2)
        if ((w & 3) != 3) return (int)(w & 3) - 1;
    return 0;
}
HE_HD int popcount21(u64 v)
{
    v &= 0x1FFFFF;
    v = v - ((v >> 1) & 0x155555);
    v = (v & 0x333333) + ((v >> 2) & 0x333333);
    v = (v + (v >> 4)) & 0x0F0F0F;
    return (int)((v * 0x010101) >> 16) & 0x3F;
}
// centred binomial: popcount of 21 bits minus popcount of the next 21 bits (sigma ~ 3.24)
HE_HD int sample_cbd_at(u64 seed, u64 stream, u64 n)
{
    const u64 w = sample_word(seed, stream, n);
    return popcount21(w) - popcount21(w >> 21);
}
// small signed value -> residue mod q
HE_HD u64 small_to_residue(int e, u64 q) { return e >= 0 ? (u64)e : q - (u64)(-e); }

// uniform in [0, q): rejection sampling below the largest multiple of q (SEAL sample_poly_uniform), at most 8 draws per
// coefficient from the sub-counter 8n..8n+7; all eight rejected (probability < 2^-32 for q < 2^60) falls back to the last
// draw reduced mod q
HE_HD u64 sample_uniform_at(u64 seed, u64 stream, u64 n, u64 q)
{
    const u64 lim = ~(u64)0 - (~(u64)0 % q) - 1; // accept v <= lim
    u64 v = 0;
    for (int k = 0; k < 8; ++k) {
        v = sample_word(seed, stream, 8 * n + (u64)k);
        if (v <= lim) break;
    }
    return v % q;
}
// streams of key generation: key `key_id` (0: public key, 1: relinearization key, 2 + galois_elt: Galois keys), digit j, then
// K streams for the uniform polynomial (one per prime) and one for the error polynomial
HE_HD u64 keygen_stream(u64 key_id, u64 digit, u64 K, u64 which /* prime index, or K for the error */)
{
    return ((u64)1 << 40) + (key_id * 64 + digit) * (K + 1) + which;
}
// streams of one asymmetric encryption, ciphertext index r: u, e0, e1
HE_HD u64 enc_stream(u64 r, int which) { return 3 * r + (u64)which; }

} // namespace client
} // namespace he355
