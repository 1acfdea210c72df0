// he_client.cpp — see he_client.h.  Host-only C++17 (client side of the backend; untimed).
#include "he_client.h"
#include "ckks_codec.h"
#include "multiword.h"
#include "sampler.h"

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace he355 {
namespace client {

namespace {

inline u64 mulm(u64 a, u64 b, const ModU64 &m) { return barrett128((u128)a * b, m); }
inline u64 negm(u64 a, u64 q) { return a ? q - a : 0; }

// little multiword helper for CRT composition (little-endian words)
struct Wide {
    std::vector<u64> w;
    explicit Wide(size_t n = 0) : w(n, 0) {}
    void mul_small(u64 b)
    {
        u64 carry = 0;
        for (auto &x : w) {
            const u128 p = (u128)x * b + carry;
            x = (u64)p;
            carry = (u64)(p >> 64);
        }
    }
    void add_mul(const Wide &a, u64 b)
    {
        u64 carry = 0;
        for (size_t i = 0; i < w.size(); ++i) {
            const u128 p = (u128)a.w[i] * b + w[i] + carry;
            w[i] = (u64)p;
            carry = (u64)(p >> 64);
        }
    }
    int cmp(const Wide &o) const
    {
        for (size_t i = w.size(); i-- > 0;)
            if (w[i] != o.w[i]) return w[i] > o.w[i] ? 1 : -1;
        return 0;
    }
    void sub(const Wide &o)
    {
        u64 borrow = 0;
        for (size_t i = 0; i < w.size(); ++i) {
            const u128 d = (u128)w[i] - o.w[i] - borrow;
            w[i] = (u64)d;
            borrow = (u64)(d >> 64) & 1;
        }
    }
    void add(const Wide &o)
    {
        u64 carry = 0;
        for (size_t i = 0; i < w.size(); ++i) {
            const u128 s = (u128)w[i] + o.w[i] + carry;
            w[i] = (u64)s;
            carry = (u64)(s >> 64);
        }
    }
    double to_double() const
    {
        double r = 0;
        for (size_t i = w.size(); i-- > 0;) r = r * 18446744073709551616.0 + (double)w[i];
        return r;
    }
};

struct Crt {
    size_t L, words;
    Wide Q, halfQ;
    std::vector<Wide> punct;
    std::vector<u64> inv;
    Crt(const Params &P, size_t L_) : L(L_), words(L_ + 2), Q(words), halfQ(words)
    {
        Q.w[0] = 1;
        for (size_t i = 0; i < L; ++i) Q.mul_small(P.primes[i].q);
        for (size_t i = 0; i < words; ++i) halfQ.w[i] = (Q.w[i] >> 1) | (i + 1 < words ? Q.w[i + 1] << 63 : 0);
        for (size_t i = 0; i < L; ++i) {
            Wide p(words);
            p.w[0] = 1;
            u64 pm = 1;
            const u64 qi = P.primes[i].q;
            for (size_t k = 0; k < L; ++k)
                if (k != i) {
                    p.mul_small(P.primes[k].q);
                    pm = (u64)(((u128)pm * (P.primes[k].q % qi)) % qi);
                }
            punct.push_back(p);
            inv.push_back(Params::invmod(pm, qi));
        }
    }
    // x in [0, Q) from residues res[i*stride]
    void compose(const Params &P, const u64 *res, size_t stride, Wide &x) const
    {
        std::fill(x.w.begin(), x.w.end(), 0);
        for (size_t i = 0; i < L; ++i) x.add_mul(punct[i], mulm(res[i * stride], inv[i], P.primes[i].mod));
        while (x.cmp(Q) >= 0) x.sub(Q);
    }
};

// iterative radix-2 transform on bit-reversed input with the shared butterflies (client/ckks_codec.h)
void fft_stages(std::vector<Cplx> &z, const std::vector<Cplx> &W, bool inverse)
{
    const size_t n = z.size();
    for (size_t len = 2; len <= n; len <<= 1)
        for (size_t t = 0; t < n / 2; ++t) fft_stage_bfly(z.data(), W.data(), len, t, inverse);
}

} // namespace

// Harvey butterflies, scalar: same transform as the device kernels and as SEAL's ntt_negacyclic_harvey
void host_ntt_forward(const PrimeTables &pt, size_t N, u64 *x)
{
    const u64 q = pt.q;
    size_t gap = N >> 1;
    for (size_t m = 1; m < N; m <<= 1, gap >>= 1)
        for (size_t i = 0; i < m; ++i) {
            const u64 w = pt.fwd_u64[m + i];
            u64 *a = x + 2 * i * gap, *b = a + gap;
            for (size_t j = 0; j < gap; ++j) {
                const u64 u = a[j], v = mulm(b[j], w, pt.mod);
                a[j] = addmod(u, v, q);
                b[j] = submod(u, v, q);
            }
        }
}
void host_ntt_inverse(const PrimeTables &pt, size_t N, u64 *x)
{
    const u64 q = pt.q;
    size_t gap = 1;
    for (size_t m = N >> 1; m >= 1; m >>= 1, gap <<= 1)
        for (size_t i = 0; i < m; ++i) {
            const u64 w = pt.inv_u64[m + i];
            u64 *a = x + 2 * i * gap, *b = a + gap;
            for (size_t j = 0; j < gap; ++j) {
                const u64 u = a[j], v = b[j];
                a[j] = addmod(u, v, q);
                b[j] = mulm(submod(u, v, q), w, pt.mod);
            }
        }
    for (size_t j = 0; j < N; ++j) x[j] = mulm(x[j], pt.ninv, pt.mod);
}

// slot -> evaluation-point index: powers of the generator 3 (CKKSEncoder / BatchEncoder matrix_reps_index_map)
void build_slot_index(size_t N, std::vector<uint32_t> &slot_index)
{
    const size_t half = N / 2, m = 2 * N;
    slot_index.resize(N);
    u64 pos = 1;
    for (size_t i = 0; i < half; ++i) {
        slot_index[i] = (uint32_t)((pos - 1) >> 1);
        slot_index[half + i] = (uint32_t)((m - pos - 1) >> 1);
        pos = (pos * 3) & (m - 1);
    }
}
// cos and sin through separate, non-inlinable calls: an optimiser that fuses the pair into sincos() gets results that can
// differ from sin()/cos() in the last bit (seen with g++ -O2 against clang), and every build must produce the same tables
__attribute__((noinline)) static double cos_only(double x) { return std::cos(x); }
__attribute__((noinline)) static double sin_only(double x) { return std::sin(x); }
// transform tables of the CKKS encoder (client/ckks_codec.h): W[len/2 + k] = exp(-2 pi i k/len), Z[n] = exp(-i pi n/N)
void build_ckks_tables(size_t N, std::vector<Cplx> &W, std::vector<Cplx> &Z)
{
    W.assign(N, Cplx{1.0, 0.0});
    for (size_t len = 2; len <= N; len <<= 1) {
        const double ang = -2 * M_PI / (double)len;
        for (size_t k = 0; k < len / 2; ++k) W[len / 2 + k] = Cplx{cos_only(ang * (double)k), sin_only(ang * (double)k)};
    }
    Z.resize(N);
    for (size_t n = 0; n < N; ++n) {
        const double ang = -M_PI * (double)n / (double)N;
        Z[n] = Cplx{cos_only(ang), sin_only(ang)};
    }
}

Client::Client(const Params &params, uint64_t seed) : P(params), rng_(seed), enc_seed_(splitmix64(seed ^ 0x656e6372797074ull)), key_seed_(splitmix64(seed ^ 0x6b657967656eull))
{
    const size_t N = P.N, K = P.K;
    build_slot_index(N, slot_index_);
    if (P.scheme == kSchemeBFV) plain_tables_ = Params::make_prime_tables(P.plain_modulus, N, P.logn, false);
    if (P.scheme == kSchemeCKKS) build_ckks_tables(N, fft_w_, zeta_);
    // secret key: ternary, NTT form at the key level
    sample_ternary(sk_, K);
    for (size_t i = 0; i < K; ++i) host_ntt_forward(P.primes[i], N, sk_.data() + i * N);
    pk_.resize(2 * K * N);
    enc_zero_symmetric(pk_.data(), 0, 0);
}

void Client::sample_ternary(std::vector<u64> &out, size_t nmod)
{
    out.assign(nmod * P.N, 0);
    for (size_t n = 0; n < P.N; ++n) {
        u64 v;
        do v = rng_() & 3; while (v == 3);
        for (size_t i = 0; i < nmod; ++i) out[i * P.N + n] = v == 0 ? P.primes[i].q - 1 : v - 1;
    }
}
void Client::sample_cbd(std::vector<u64> &out, size_t nmod)
{
    out.assign(nmod * P.N, 0);
    for (size_t n = 0; n < P.N; ++n) { // centred binomial 21 - 21 bits (SEAL's default noise, sigma ~ 3.2)
        const u64 v = rng_();
        const int e = __builtin_popcountll(v & 0x1FFFFF) - __builtin_popcountll((v >> 21) & 0x1FFFFF);
        for (size_t i = 0; i < nmod; ++i) out[i * P.N + n] = e >= 0 ? (u64)e : P.primes[i].q - (u64)(-e);
    }
}
void Client::sample_uniform(std::vector<u64> &out, size_t nmod)
{
    out.assign(nmod * P.N, 0);
    for (size_t i = 0; i < nmod; ++i) {
        const u64 q = P.primes[i].q, lim = UINT64_MAX - (UINT64_MAX % q) - 1;
        for (size_t n = 0; n < P.N; ++n) {
            u64 v;
            do v = rng_(); while (v > lim);
            out[i * P.N + n] = v % q;
        }
    }
}

// (b, a) with b = -(a*s + e), NTT form, key level.  Randomness: counter-based streams of (key_id, digit) (client/sampler.h),
// so the device key generator (he355_keygen_*) produces the same key from the same seed.
void Client::enc_zero_symmetric(u64 *out, u64 key_id, u64 digit)
{
    const size_t N = P.N, K = P.K;
    std::vector<u64> e(K * N);
    for (size_t n = 0; n < N; ++n) {
        const int v = sample_cbd_at(key_seed_, keygen_stream(key_id, digit, K, K), n);
        for (size_t i = 0; i < K; ++i) e[i * N + n] = small_to_residue(v, P.primes[i].q);
    }
    for (size_t i = 0; i < K; ++i) {
        const PrimeTables &pt = P.primes[i];
        host_ntt_forward(pt, N, e.data() + i * N);
        for (size_t n = 0; n < N; ++n) {
            const u64 a = sample_uniform_at(key_seed_, keygen_stream(key_id, digit, K, i), n, pt.q);
            const u64 v = addmod(mulm(a, sk_[i * N + n], pt.mod), e[i * N + n], pt.q);
            out[i * N + n] = negm(v, pt.q);
            out[(K + i) * N + n] = a;
        }
    }
}

// KeyGenerator::generate_one_kswitch_key: digit j = Enc(0) + (P mod q_j) * new_key on residue j of the first poly
std::vector<u64> Client::make_kswitch_key(const std::vector<u64> &new_key, u64 key_id)
{
    const size_t N = P.N, K = P.K, Ld = P.Ltop;
    if (K < 2) throw std::invalid_argument("encryption parameters do not support key switching");
    std::vector<u64> out(Ld * 2 * K * N);
    const u64 special = P.primes[K - 1].q;
    for (size_t j = 0; j < Ld; ++j) {
        u64 *dig = out.data() + j * 2 * K * N;
        enc_zero_symmetric(dig, key_id, j);
        const PrimeTables &pt = P.primes[j];
        const u64 f = special % pt.q;
        for (size_t n = 0; n < N; ++n) dig[j * N + n] = addmod(dig[j * N + n], mulm(new_key[j * N + n], f, pt.mod), pt.q);
    }
    return out;
}
std::vector<u64> Client::make_relin_key()
{
    std::vector<u64> s2(P.K * P.N);
    for (size_t i = 0; i < P.K; ++i)
        for (size_t n = 0; n < P.N; ++n) s2[i * P.N + n] = mulm(sk_[i * P.N + n], sk_[i * P.N + n], P.primes[i].mod);
    return make_kswitch_key(s2, 1);
}
std::vector<u64> Client::make_galois_key(uint32_t elt)
{
    const std::vector<uint32_t> perm = P.galois_perm_ntt(elt);
    std::vector<u64> rs(P.K * P.N);
    for (size_t i = 0; i < P.K; ++i)
        for (size_t n = 0; n < P.N; ++n) rs[i * P.N + n] = sk_[i * P.N + perm[n]];
    return make_kswitch_key(rs, 2 + (u64)elt);
}

std::vector<u64> Client::ckks_encode(const double *values, size_t count, double scale) const
{
    const size_t N = P.N, half = N / 2;
    if (count > half) throw std::invalid_argument("Not enough slots available to create packed plaintext");
    // z[j] = p(zeta^(2j+1)) = sum_n (c_n zeta^n) e^{2 pi i jn/N}  =>  c_n = zeta^{-n} * DFT(z)[n] / N
    std::vector<Cplx> z(N, Cplx{0.0, 0.0});
    for (size_t i = 0; i < count; ++i) { // values land bit-reversed: the transform below runs on bit-reversed input
        z[bitrev_u32(slot_index_[i], P.logn)].re = values[i];
        z[bitrev_u32(slot_index_[half + i], P.logn)].re = values[i]; // conjugate of a real value
    }
    fft_stages(z, fft_w_, false);
    std::vector<u64> out(P.Ltop * N);
    for (size_t n = 0; n < N; ++n) {
        const double r = ckks_encode_coeff(z[n], zeta_[n], (double)N, scale);
        if (!(std::fabs(r) < 9.2e18)) throw std::invalid_argument("encoded values are too large");
        const long long iv = (long long)r;
        for (size_t i = 0; i < P.Ltop; ++i) {
            const u64 q = P.primes[i].q;
            out[i * N + n] = iv >= 0 ? (u64)iv % q : negm((u64)(-iv) % q, q);
        }
    }
    for (size_t i = 0; i < P.Ltop; ++i) host_ntt_forward(P.primes[i], N, out.data() + i * N);
    return out;
}

void Client::ckks_decode(const u64 *plain_ntt, size_t L, double scale, double *out) const
{
    const size_t N = P.N, half = N / 2;
    std::vector<u64> coeff(plain_ntt, plain_ntt + L * N);
    for (size_t i = 0; i < L; ++i) host_ntt_inverse(P.primes[i], N, coeff.data() + i * N);
    const Crt crt(P, L);
    Wide x(crt.words), y(crt.words);
    std::vector<Cplx> z(N);
    for (size_t n = 0; n < N; ++n) {
        crt.compose(P, coeff.data() + n, N, x);
        double v;
        if (x.cmp(crt.halfQ) > 0) {
            y = crt.Q;
            y.sub(x);
            v = -mw_to_double(y.w.data(), (int)crt.words);
        } else {
            v = mw_to_double(x.w.data(), (int)crt.words);
        }
        z[bitrev_u32((u32)n, P.logn)] = ckks_decode_coeff(v, zeta_[n], scale);
    }
    fft_stages(z, fft_w_, true); // unnormalised inverse transform = evaluation at the slots' points
    for (size_t i = 0; i < half; ++i) out[i] = z[slot_index_[i]].re;
}

std::vector<u64> Client::bfv_encode(const int64_t *values, size_t count) const
{
    const size_t N = P.N;
    if (count > N) throw std::invalid_argument("Not enough slots available to create packed plaintext");
    const u64 t = P.plain_modulus;
    std::vector<u64> ev(N, 0);
    for (size_t i = 0; i < count; ++i) {
        const int64_t v = values[i];
        const u64 r = v >= 0 ? (u64)v % t : negm((u64)(-v) % t, t);
        ev[bitrev(slot_index_[i], P.logn)] = r;
    }
    host_ntt_inverse(plain_tables_, N, ev.data());
    return ev;
}
void Client::bfv_decode(const u64 *plain, int64_t *out) const
{
    const size_t N = P.N;
    const u64 t = P.plain_modulus;
    std::vector<u64> ev(plain, plain + N);
    host_ntt_forward(plain_tables_, N, ev.data());
    for (size_t i = 0; i < N; ++i) {
        const u64 v = ev[bitrev(slot_index_[i], P.logn)];
        out[i] = v > t / 2 ? (int64_t)v - (int64_t)t : (int64_t)v;
    }
}

// RNSTool::divide_and_round_q_last(_ntt)_inplace at the key level: drop the special prime with rounding
void Client::divide_round_last(const std::vector<u64> &in, size_t size, std::vector<u64> &out) const
{
    const size_t N = P.N, K = P.K, L = K - 1;
    const bool ckks = P.scheme == kSchemeCKKS;
    const PrimeTables &pl = P.primes[K - 1];
    const u64 half = pl.q >> 1;
    out.assign(size * L * N, 0);
    std::vector<u64> r(N), tmp(N);
    for (size_t k = 0; k < size; ++k) {
        std::memcpy(r.data(), in.data() + (k * K + K - 1) * N, N * 8);
        if (ckks) host_ntt_inverse(pl, N, r.data());
        for (size_t n = 0; n < N; ++n) r[n] = addmod(r[n], half, pl.q);
        for (size_t i = 0; i < L; ++i) {
            const PrimeTables &pt = P.primes[i];
            const u64 half_i = half % pt.q, inv = Params::invmod(pl.q % pt.q, pt.q);
            for (size_t n = 0; n < N; ++n) tmp[n] = submod(r[n] % pt.q, half_i, pt.q);
            if (ckks) host_ntt_forward(pt, N, tmp.data());
            for (size_t n = 0; n < N; ++n)
                out[(k * L + i) * N + n] = mulm(submod(in[(k * K + i) * N + n], tmp[n], pt.q), inv, pt.mod);
        }
    }
}

std::vector<u64> Client::encrypt_zero()
{
    // Encryptor::encrypt_zero (asymmetric): u ternary, e0,e1 ~ CBD at the key level, then divide-and-round by the
    // special prime
    const size_t N = P.N, K = P.K;
    const bool ckks = P.scheme == kSchemeCKKS;
    std::vector<u64> u(K * N), e(K * N), z(2 * K * N);
    const uint64_t r = enc_index_++;
    for (size_t n = 0; n < N; ++n) {
        const int v = sample_ternary_at(enc_seed_, enc_stream(r, 0), n);
        for (size_t i = 0; i < K; ++i) u[i * N + n] = small_to_residue(v, P.primes[i].q);
    }
    for (size_t i = 0; i < K; ++i) host_ntt_forward(P.primes[i], N, u.data() + i * N);
    for (size_t k = 0; k < 2; ++k) {
        for (size_t n = 0; n < N; ++n) {
            const int v = sample_cbd_at(enc_seed_, enc_stream(r, 1 + (int)k), n);
            for (size_t i = 0; i < K; ++i) e[i * N + n] = small_to_residue(v, P.primes[i].q);
        }
        for (size_t i = 0; i < K; ++i) {
            const PrimeTables &pt = P.primes[i];
            u64 *zi = z.data() + (k * K + i) * N;
            for (size_t n = 0; n < N; ++n) zi[n] = mulm(u[i * N + n], pk_[(k * K + i) * N + n], pt.mod);
            if (ckks) host_ntt_forward(pt, N, e.data() + i * N);
            else host_ntt_inverse(pt, N, zi);
            for (size_t n = 0; n < N; ++n) zi[n] = addmod(zi[n], e[i * N + n], pt.q);
        }
    }
    if (K == 1) return z;
    std::vector<u64> out;
    divide_round_last(z, 2, out);
    return out;
}

std::vector<u64> Client::encrypt(const u64 *plain)
{
    std::vector<u64> ct = encrypt_zero();
    const size_t N = P.N, L = P.Ltop;
    if (P.scheme == kSchemeCKKS) {
        for (size_t i = 0; i < L; ++i)
            for (size_t n = 0; n < N; ++n) ct[i * N + n] = addmod(ct[i * N + n], plain[i * N + n], P.primes[i].q);
    } else {
        // c0 += round(q*m/t)  (multiply_add_plain_with_scaling_variant)
        const u64 t = P.plain_modulus;
        u64 q_mod_t = 1;
        for (size_t i = 0; i < L; ++i) q_mod_t = (u64)(((u128)q_mod_t * (P.primes[i].q % t)) % t);
        const u64 thr = (t + 1) >> 1;
        for (size_t i = 0; i < L; ++i) {
            const PrimeTables &pt = P.primes[i];
            const u64 tinv = Params::invmod(t % pt.q, pt.q);
            const u64 qdivt = mulm(negm(q_mod_t % pt.q, pt.q), tinv, pt.mod); // floor(q/t) mod q_i
            for (size_t n = 0; n < N; ++n) {
                const u64 fix = (u64)(((u128)plain[n] * q_mod_t + thr) / t);
                const u64 v = addmod(mulm(plain[n], qdivt, pt.mod), fix % pt.q, pt.q);
                ct[i * N + n] = addmod(ct[i * N + n], v, pt.q);
            }
        }
    }
    return ct;
}

std::vector<u64> Client::decrypt(const u64 *ct, size_t size, size_t L) const
{
    const size_t N = P.N;
    const bool ckks = P.scheme == kSchemeCKKS;
    if (size < 2) throw std::invalid_argument("ciphertext size must be at least 2");
    std::vector<u64> phase(L * N), tmp(N);
    for (size_t i = 0; i < L; ++i) {
        const PrimeTables &pt = P.primes[i];
        u64 *o = phase.data() + i * N;
        for (size_t k = size; k-- > 0;) { // Horner in s
            std::memcpy(tmp.data(), ct + (k * L + i) * N, N * 8);
            if (!ckks) host_ntt_forward(pt, N, tmp.data());
            for (size_t n = 0; n < N; ++n) {
                const u64 v = (k == size - 1) ? 0 : mulm(o[n], sk_[i * N + n], pt.mod);
                o[n] = addmod(v, tmp[n], pt.q);
            }
        }
        if (!ckks) host_ntt_inverse(pt, N, o);
    }
    if (ckks) return phase;
    // BFV: m = round(t * x / q) mod t on the centred phase (exact form of decrypt_scale_and_round)
    const Crt crt(P, L);
    const u64 t = P.plain_modulus;
    std::vector<u64> plain(N);
    Wide x(crt.words), num(crt.words), prod(crt.words);
    const double Qd = crt.Q.to_double();
    for (size_t n = 0; n < N; ++n) {
        crt.compose(P, phase.data() + n, N, x);
        num = x;
        num.mul_small(t);
        num.add(crt.halfQ);
        u64 m = (u64)(num.to_double() / Qd);
        if (m > 0) --m;
        for (;;) { // largest m with m*Q <= num
            prod = crt.Q;
            prod.mul_small(m + 1);
            if (prod.cmp(num) <= 0) ++m;
            else break;
        }
        plain[n] = m % t;
    }
    return plain;
}

} // namespace client
} // namespace he355
