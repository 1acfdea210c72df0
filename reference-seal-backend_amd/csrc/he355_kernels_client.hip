// he355_kernels_client.hip — the client side on the device: RLWE sampling, asymmetric encryption,
// decryption, BFV scale-and-round, CKKS / BFV encoders, key generation.  Untimed by the harness (encode / encrypt / decrypt / decode
// bracket operate()); each kernel has a bit-identical host twin in csrc/client/he_client.cpp built from the same inline code
// (client/{sampler,multiword,ckks_codec}.h).  Launchers are declared in he355_kernels.h.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "he355_kernels.h"
#include "modarith.h"
#include "ntt_core.h"
#include "client/ckks_codec.h"
#include "client/multiword.h"
#include "client/sampler.h"

#if !defined(HE355_KNS) || !defined(HE355_U64_FOLD)
#error "he355_kernels_client.hip is compiled once per form of the u64 engine (Makefile)"
#endif
namespace he355 {
namespace HE355_KNS {
namespace {

constexpr int kBlock = 256;
__device__ __forceinline__ ModU64 make_modu(const PrimeDev &p)
{
    ModU64 m;
    m.q = p.q; m.cr0 = p.cr0; m.cr1 = p.cr1;
    return m;
}
// =======================================================================================================
// Client side on the device: asymmetric encryption and decryption.
// Reference call sites: encryptor()->encrypt (ckks eltwise .cpp:242, bfv eltwise .cpp:233), SEALContextWrapper::decrypt
// (seal_context.cpp:265-287).  Same arithmetic as csrc/client/he_client.cpp (host) and oracle ho_encrypt / ho_decrypt_phase.
// =======================================================================================================
// u (ternary) and e0, e1 (centred binomial) of ciphertext `first_index + r`, as residues under all K key primes.
// One thread = one coefficient of one ciphertext: three counter-based draws (client/sampler.h), 3*K stores.
__global__ void __launch_bounds__(kBlock) k_enc_sample(u64 *u, u64 *e, const PrimeDev *primes, int K, int logN, u64 n_cts, u64 seed, u64 first_index)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 r = gid >> logN, n = gid & (((u64)1 << logN) - 1);
    if (r >= n_cts) return;
    const u64 idx = first_index + r;
    const int vu = client::sample_ternary_at(seed, client::enc_stream(idx, 0), n);
    const int v0 = client::sample_cbd_at(seed, client::enc_stream(idx, 1), n), v1 = client::sample_cbd_at(seed, client::enc_stream(idx, 2), n);
    for (int i = 0; i < K; ++i) {
        const u64 q = primes[i].q;
        u[((r * K + i) << logN) + n] = client::small_to_residue(vu, q);
        e[(((r * 2 + 0) * K + i) << logN) + n] = client::small_to_residue(v0, q);
        e[(((r * 2 + 1) * K + i) << logN) + n] = client::small_to_residue(v1, q);
    }
}
// z[r][k][i] = u[r][i] (.) pk[k][i] (+ z[r][k][i] when add_in: CKKS, where z holds NTT(e_k)); all NTT form, key level.
__global__ void __launch_bounds__(kBlock) k_enc_mul_pk(const u64 *u, const u64 *pk, u64 *z, const PrimeDev *primes, int K, int logN, u64 n_cts, int add_in)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 pp = gid >> (logN - 1), e2 = gid & (((u64)1 << (logN - 1)) - 1);
    const u64 r = pp / (2 * K);
    if (r >= n_cts) return;
    const int ki = (int)(pp % (2 * K)), i = ki % K;
    const PrimeDev &P = primes[i];
    const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(u + ((r * K + i) << logN))[e2];
    const ulonglong2 y = reinterpret_cast<const ulonglong2 *>(pk + ((u64)ki << logN))[e2];
    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(z + ((r * 2 * K + ki) << logN)) + e2;
    const ModU64 m = make_modu(P);
    ulonglong2 v;
    v.x = barrett128((u128)x.x * y.x, m); v.y = barrett128((u128)x.y * y.y, m);
    if (add_in) {
        const ulonglong2 o = *dst;
        v.x = addmod(v.x, o.x, P.q); v.y = addmod(v.y, o.y, P.q);
    }
    *dst = v;
}
// RNSTool::divide_and_round_q_last_inplace on coefficient-form data at the key level (BFV encryption):
// z [n][2][K][N] -> out [n][2][L][N], L = K-1, dropping the special prime with rounding.
__global__ void __launch_bounds__(kBlock) k_divround_last_coeff(const u64 *z, u64 *out, const PrimeDev *primes, const FloorConst *fc, int K, int logN,
                                                                u64 n_polys)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 poly = gid >> logN, n = gid & (((u64)1 << logN) - 1);
    if (poly >= n_polys) return;
    const int L = K - 1;
    const u64 qs = primes[K - 1].q, half = qs >> 1;
    const u64 r = addmod(z[((poly * K + K - 1) << logN) + n], half, qs);
    for (int i = 0; i < L; ++i) {
        const PrimeDev &Pi = primes[i];
        const FloorConst f = fc[(K - 1) * K + i];
        const u64 ri = qs > Pi.q ? barrett64(r, make_modu(Pi)) : r;
        const u64 delta = submod(ri, f.half_mod, Pi.q);
        out[((poly * L + i) << logN) + n] = mulmod(submod(z[((poly * K + i) << logN) + n], delta, Pi.q), f.inv, make_modu(Pi)) /* any prime: Barrett */;
    }
}
// BFV: c0 += round(q*m/t) (util/scalingvariant.cpp multiply_add_plain_with_scaling_variant); plain [n][N] mod t.
struct ScaleVariantConst {
    u64 t, q_mod_t, thr;
    u64 qdivt[kMaxPrimes]; // floor(q/t) mod q_i
};
__global__ void __launch_bounds__(kBlock) k_bfv_add_scaled_plain(u64 *ct, const u64 *plain, const PrimeDev *primes, ScaleVariantConst sv, int L, int logN,
                                                                 u64 n_cts)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 r = gid >> logN, n = gid & (((u64)1 << logN) - 1);
    if (r >= n_cts) return;
    const u64 m = plain[(r << logN) + n];
    const u64 fix = (u64)(((u128)m * sv.q_mod_t + sv.thr) / sv.t);
    for (int i = 0; i < L; ++i) {
        const PrimeDev &Pi = primes[i];
        const ModU64 mod = make_modu(Pi);
        const u64 v = addmod(barrett128((u128)m * sv.qdivt[i], mod), barrett64(fix, mod), Pi.q);
        u64 *c = ct + ((r * 2 * L + i) << logN) + n;
        *c = addmod(*c, v, Pi.q);
    }
}
// Decryptor dot_product_ct_sk_array: out[r][i] = c0 + c1 s + c2 s^2 ... (Horner in s), NTT form; ct [n][size][L][N], sk [K][N].
__global__ void __launch_bounds__(kBlock) k_dot_sk(const u64 *ct, const u64 *sk, u64 *out, const PrimeDev *primes, int L, int size, int logN, u64 n_cts)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 pp = gid >> (logN - 1), e2 = gid & (((u64)1 << (logN - 1)) - 1);
    const u64 r = pp / L;
    if (r >= n_cts) return;
    const int i = (int)(pp % L);
    const PrimeDev &P = primes[i];
    const ModU64 m = make_modu(P);
    const ulonglong2 s = reinterpret_cast<const ulonglong2 *>(sk + ((u64)i << logN))[e2];
    ulonglong2 acc = reinterpret_cast<const ulonglong2 *>(ct + (((r * size + size - 1) * L + i) << logN))[e2];
    for (int k = size - 2; k >= 0; --k) {
        const ulonglong2 c = reinterpret_cast<const ulonglong2 *>(ct + (((r * size + k) * L + i) << logN))[e2];
        acc.x = addmod(barrett128((u128)acc.x * s.x, m), c.x, P.q);
        acc.y = addmod(barrett128((u128)acc.y * s.y, m), c.y, P.q);
    }
    reinterpret_cast<ulonglong2 *>(out + ((r * L + i) << logN))[e2] = acc;
}
// BFV Decryptor: plain = round(t * [phase]_Q / Q) mod t per coefficient, exact (CRT composition in multiword arithmetic,
// client/multiword.h — the same inline code the host client runs).  phase [n][L][N] coefficient form -> plain [n][N].
struct CrtDev {
    client::CrtView v;
    double Qd;
    u64 t;
};
__global__ void __launch_bounds__(kBlock) k_bfv_scale_round(const u64 *phase, u64 *plain, const PrimeDev *primes, CrtDev c, int logN, u64 n_cts)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 r = gid >> logN, n = gid & (((u64)1 << logN) - 1);
    if (r >= n_cts) return;
    ModU64 mods[16];
    for (int i = 0; i < c.v.L; ++i) mods[i] = make_modu(primes[i]);
    u64 x[client::kMwWords];
    client::crt_compose(c.v, mods, phase + ((r * c.v.L) << logN) + n, (u64)1 << logN, x);
    plain[(r << logN) + n] = client::bfv_scale_round(c.v, x, c.t, c.Qd);
}

// ---- key generation on the device: KeyGenerator::create_relin_keys / create_galois_keys (seal_context.cpp:53,69) -------------
// Digit j of a key-switching key = Enc_sym(0) at the key level with (P mod q_j) * new_key added to residue j of the first
// polynomial (keygenerator.cpp generate_one_kswitch_key).  key [Ld][2][K][N]; e [Ld][K][N] scratch.
// Step 1: the uniform polynomials a (written straight into key[j][1], NTT form as SEAL samples them) and the error polynomials
// (coefficient form, into e); counter-based streams keygen_stream(key_id, j, K, .) of client/sampler.h.
__global__ void __launch_bounds__(kBlock) k_keygen_sample(u64 *key, u64 *e, const PrimeDev *primes, int K, int logN, int Ld, u64 seed, u64 key_id)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 j = gid >> logN, n = gid & (((u64)1 << logN) - 1);
    if (j >= (u64)Ld) return;
    const int v = client::sample_cbd_at(seed, client::keygen_stream(key_id, j, K, K), n);
    for (int i = 0; i < K; ++i) {
        const u64 q = primes[i].q;
        e[((j * K + i) << logN) + n] = client::small_to_residue(v, q);
        key[(((j * 2 + 1) * K + i) << logN) + n] = client::sample_uniform_at(seed, client::keygen_stream(key_id, j, K, i), n, q);
    }
}
// Step 2 (after the forward NTT of e): key[j][0][i] = -(a*s + e) (+ (P mod q_j) * new_key[j] when i == j)
__global__ void __launch_bounds__(kBlock) k_keygen_finish(u64 *key, const u64 *e, const u64 *sk, const u64 *new_key, const PrimeDev *primes, int K, int logN, int Ld)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 ji = gid >> logN, n = gid & (((u64)1 << logN) - 1);
    if (ji >= (u64)Ld * K) return;
    const u64 j = ji / K;
    const int i = (int)(ji % K);
    const PrimeDev &P = primes[i];
    const ModU64 m = make_modu(P);
    const u64 a = key[(((j * 2 + 1) * K + i) << logN) + n];
    u64 b = addmod(barrett128((u128)a * sk[((u64)i << logN) + n], m), e[((j * K + i) << logN) + n], P.q);
    b = b ? P.q - b : 0;
    if ((u64)i == j) b = addmod(b, barrett128((u128)new_key[((u64)i << logN) + n] * barrett64(primes[K - 1].q, m), m), P.q);
    key[(((j * 2 + 0) * K + i) << logN) + n] = b;
}
// new_key for the relinearization key: s^2; for a Galois key: s permuted (NTT-form gather, GaloisTool::apply_galois_ntt)
__global__ void __launch_bounds__(kBlock) k_keygen_target(const u64 *sk, const uint32_t *perm, u64 *out, const PrimeDev *primes, int K, int logN)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 i = gid >> logN, n = gid & (((u64)1 << logN) - 1);
    if (i >= (u64)K) return;
    if (perm) {
        out[gid] = sk[(i << logN) + perm[n]];
    } else {
        const u64 s = sk[gid];
        out[gid] = barrett128((u128)s * s, make_modu(primes[i]));
    }
}

// ---- encoders on the device: CKKSEncoder / BatchEncoder encode and decode (seal_context.cpp:145-185 call sites) ------------
// The floating-point work is the shared inline code of client/ckks_codec.h (same IEEE operations, same tables as the host
// client).  One 1024-thread workgroup owns one vector; the N-point transform runs stage by stage on a per-vector buffer in
// global memory (L2-resident: 16 B x N), workgroup barriers between stages.
constexpr int kEncBlock = 1024;
struct EncTables {
    const uint32_t *slot_index; // [N] slot -> evaluation-point index
    const client::Cplx *W, *Z;  // transform and twist tables
};
__device__ __forceinline__ void fft_stages_block(client::Cplx *z, const client::Cplx *W, u32 N, bool inverse)
{
    for (u32 len = 2; len <= N; len <<= 1) {
        for (u32 t = threadIdx.x; t < N / 2; t += kEncBlock) client::fft_stage_bfly(z, W, len, t, inverse);
        __syncthreads();
    }
}
// values [n][count] doubles -> plain [n][Ltop][N] integer coefficients as residues (coefficient form; the caller transforms them)
__global__ void __launch_bounds__(kEncBlock) k_ckks_encode(const double *values, u64 count, double scale, client::Cplx *zbuf, u64 *plain, EncTables T,
                                                           const PrimeDev *primes, int Ltop, int logN, int *err)
{
    const u32 N = 1u << logN, half = N >> 1;
    const u64 r = blockIdx.x;
    client::Cplx *z = zbuf + r * N;
    for (u32 n = threadIdx.x; n < N; n += kEncBlock) z[n] = client::Cplx{0.0, 0.0};
    __syncthreads();
    for (u32 i = threadIdx.x; i < count; i += kEncBlock) {
        const double v = values[r * count + i];
        z[client::bitrev_u32(T.slot_index[i], logN)].re = v;
        z[client::bitrev_u32(T.slot_index[half + i], logN)].re = v;
    }
    __syncthreads();
    fft_stages_block(z, T.W, N, false);
    for (u32 n = threadIdx.x; n < N; n += kEncBlock) {
        const double rv = client::ckks_encode_coeff(z[n], T.Z[n], (double)N, scale);
        long long iv = 0;
        if (!(fabs(rv) < 9.2e18)) atomicOr(err, 1);
        else iv = (long long)rv;
        for (int i = 0; i < Ltop; ++i) {
            const u64 q = primes[i].q;
            const u64 m = iv >= 0 ? (u64)iv % q : (u64)(-iv) % q;
            plain[((r * Ltop + i) << logN) + n] = (iv >= 0 || m == 0) ? m : q - m;
        }
    }
}
// CKKSEncoder::decode in two kernels.
// (1) k_ckks_decode_compose<W>: ONE THREAD PER COEFFICIENT over the whole batch -- CRT composition of the L = W - 2 residues into a W-word
//     integer, centred, divided by the scale and twisted into the transform's input.  The multiword arithmetic is client/multiword.h's
//     (the host decoder's) with the word count a template parameter: fixed trip counts, so x[] lives in registers -- the run-time-sized
//     version indexed a private array in scratch memory and ran as one 1024-thread block per plaintext (10 ms for ONE result at N = 2^15,
//     L = 16; profiles/r05_bridge_phases.jsonl).  Same operations in the same order: the same doubles, bit for bit.
// (2) k_ckks_decode_fft: one 1024-thread block per plaintext runs the N-point transform and writes the wanted slots.
template <int W>
__global__ void __launch_bounds__(kBlock) k_ckks_decode_compose(const u64 *coeff, double scale, client::Cplx *zbuf, EncTables T, const PrimeDev *primes, CrtDev c,
                                                                int logN, u64 n_vec)
{
    constexpr int L = W - 2;
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 r = gid >> logN;
    const u32 n = (u32)(gid & (((u64)1 << logN) - 1));
    if (r >= n_vec) return;
    u64 x[W];
#pragma unroll
    for (int k = 0; k < W; ++k) x[k] = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) { // client::crt_compose: x += punct_i * ((res_i * inv_i) mod q_i)
        const u64 f = barrett128((u128)coeff[((r * L + (u64)i) << logN) + n] * c.v.inv[i], make_modu(primes[i]));
        const u64 *pu = c.v.punct + (u64)i * W;
        u64 carry = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const u128 p = (u128)pu[k] * f + x[k] + carry;
            x[k] = (u64)p;
            carry = (u64)(p >> 64);
        }
    }
    auto cmp = [&](const u64 *b) { // mw_cmp(x, b)
        int res = 0;
#pragma unroll
        for (int k = 0; k < W; ++k)
            if (x[k] != b[k]) res = x[k] > b[k] ? 1 : -1; // (ascending: the most significant differing word decides last)
        return res;
    };
    while (cmp(c.v.Q) >= 0) { // x -= Q (at most L times)
        u64 borrow = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const u128 d = (u128)x[k] - c.v.Q[k] - borrow;
            x[k] = (u64)d;
            borrow = (u64)(d >> 64) & 1;
        }
    }
    double v;
    if (cmp(c.v.halfQ) > 0) { // centred representative: -(Q - x)
        u64 y[W];
        u64 borrow = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const u128 d = (u128)c.v.Q[k] - x[k] - borrow;
            y[k] = (u64)d;
            borrow = (u64)(d >> 64) & 1;
        }
        v = -client::mw_to_double(y, W);
    } else {
        v = client::mw_to_double(x, W);
    }
    zbuf[(r << logN) + client::bitrev_u32(n, logN)] = client::ckks_decode_coeff(v, T.Z[n], scale);
}
__global__ void __launch_bounds__(kEncBlock) k_ckks_decode_fft(client::Cplx *zbuf, double *out, EncTables T, int logN, SlotRanges sr)
{
    const u32 N = 1u << logN;
    const u64 r = blockIdx.x;
    client::Cplx *z = zbuf + r * N;
    fft_stages_block(z, T.W, N, true);
    u32 base = 0;
    for (u32 g = 0; g < sr.n; ++g) {
        for (u32 i = threadIdx.x; i < sr.count[g]; i += kEncBlock) out[r * sr.total + base + i] = z[T.slot_index[sr.first[g] + i]].re;
        base += sr.count[g];
    }
}
// BatchEncoder::encode: values [n][count] int64 -> evaluations mod t at the bit-reversed slot positions (the caller applies the
// inverse NTT mod t); BatchEncoder::decode: evaluations -> centred int64 slots
__global__ void __launch_bounds__(kBlock) k_bfv_encode_scatter(const long long *values, u64 count, u64 *ev, const uint32_t *slot_index, u64 t, int logN, u64 n_vec)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 r = gid >> logN, i = gid & (((u64)1 << logN) - 1);
    if (r >= n_vec || i >= count) return;
    const long long v = values[r * count + i];
    const u64 m = v >= 0 ? (u64)v % t : (u64)(-v) % t;
    ev[(r << logN) + client::bitrev_u32(slot_index[i], logN)] = (v >= 0 || m == 0) ? m : t - m;
}
// one thread per wanted slot: gid = r * sr.total + position inside the concatenated ranges
__global__ void __launch_bounds__(kBlock) k_bfv_decode_gather(const u64 *ev, long long *out, const uint32_t *slot_index, u64 t, int logN, u64 n_vec, SlotRanges sr)
{
    const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 r = gid / sr.total;
    if (r >= n_vec) return;
    u32 pos = (u32)(gid - r * sr.total), g = 0;
    while (pos >= sr.count[g]) pos -= sr.count[g++];
    const u32 i = sr.first[g] + pos;
    const u64 v = ev[(r << logN) + client::bitrev_u32(slot_index[i], logN)];
    out[gid] = v > t / 2 ? (long long)v - (long long)t : (long long)v;
}

inline unsigned grid_for(u64 jobs, u64 per_block) { return (unsigned)((jobs + per_block - 1) / per_block); }

} // namespace

void launch_enc_sample(const KernelEnv &env, u64 n_cts, u64 seed, u64 first_index, u64 *u, u64 *e)
{
    if (!n_cts) return;
    const int logN = env.logn1 + kRowLog;
    hipLaunchKernelGGL(k_enc_sample, dim3(grid_for(n_cts << logN, kBlock)), dim3(kBlock), 0, env.stream, u, e, env.primes, env.K, logN, n_cts, seed, first_index);
}
void launch_enc_mul_pk(const KernelEnv &env, u64 n_cts, const u64 *u, const u64 *pk, u64 *z, bool add_in)
{
    if (!n_cts) return;
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (n_cts * 2 * env.K) << (logN - 1);
    hipLaunchKernelGGL(k_enc_mul_pk, dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, u, pk, z, env.primes, env.K, logN, n_cts, add_in ? 1 : 0);
}
void launch_divround_last_coeff(const KernelEnv &env, u64 n_polys, const u64 *z, u64 *out)
{
    if (!n_polys) return;
    const int logN = env.logn1 + kRowLog;
    hipLaunchKernelGGL(k_divround_last_coeff, dim3(grid_for(n_polys << logN, kBlock)), dim3(kBlock), 0, env.stream, z, out, env.primes, env.floor_consts, env.K,
                       logN, n_polys);
}
void launch_bfv_add_scaled_plain(const KernelEnv &env, int L, u64 n_cts, u64 *ct, const u64 *plain, u64 t, u64 q_mod_t, const u64 *qdivt)
{
    if (!n_cts) return;
    const int logN = env.logn1 + kRowLog;
    ScaleVariantConst sv;
    sv.t = t; sv.q_mod_t = q_mod_t; sv.thr = (t + 1) >> 1;
    if (L > kMaxPrimes) throw std::invalid_argument("too many data primes");
    for (int i = 0; i < kMaxPrimes; ++i) sv.qdivt[i] = i < L ? qdivt[i] : 0;
    hipLaunchKernelGGL(k_bfv_add_scaled_plain, dim3(grid_for(n_cts << logN, kBlock)), dim3(kBlock), 0, env.stream, ct, plain, env.primes, sv, L, logN, n_cts);
}
void launch_dot_sk(const KernelEnv &env, int L, int size, u64 n_cts, const u64 *ct, const u64 *sk, u64 *out)
{
    if (!n_cts) return;
    const int logN = env.logn1 + kRowLog;
    const u64 threads = (n_cts * L) << (logN - 1);
    hipLaunchKernelGGL(k_dot_sk, dim3(grid_for(threads, kBlock)), dim3(kBlock), 0, env.stream, ct, sk, out, env.primes, L, size, logN, n_cts);
}
void launch_bfv_scale_round(const KernelEnv &env, u64 n_cts, const u64 *phase, u64 *plain, const CrtTablesDev &c)
{
    if (!n_cts) return;
    const int logN = env.logn1 + kRowLog;
    CrtDev d;
    d.v.L = c.L; d.v.words = c.words; d.v.Q = c.Q; d.v.halfQ = c.halfQ; d.v.punct = c.punct; d.v.inv = c.inv;
    d.Qd = c.Qd; d.t = c.t;
    hipLaunchKernelGGL(k_bfv_scale_round, dim3(grid_for(n_cts << logN, kBlock)), dim3(kBlock), 0, env.stream, phase, plain, env.primes, d, logN, n_cts);
}

void launch_ckks_encode(const KernelEnv &env, u64 n_vec, const double *values, u64 count, double scale, void *zbuf, u64 *plain, const EncTablesDev &t, int *err)
{
    if (!n_vec) return;
    EncTables T;
    T.slot_index = t.slot_index; T.W = static_cast<const client::Cplx *>(t.W); T.Z = static_cast<const client::Cplx *>(t.Z);
    hipLaunchKernelGGL(k_ckks_encode, dim3((unsigned)n_vec), dim3(kEncBlock), 0, env.stream, values, count, scale, static_cast<client::Cplx *>(zbuf), plain, T,
                       env.primes, env.Ltop, env.logn1 + kRowLog, err);
}
void launch_ckks_decode(const KernelEnv &env, u64 n_vec, const u64 *coeff, double scale, void *zbuf, double *out, const EncTablesDev &t, const CrtTablesDev &c,
                        const SlotRanges &sr)
{
    if (!n_vec) return;
    EncTables T;
    T.slot_index = t.slot_index; T.W = static_cast<const client::Cplx *>(t.W); T.Z = static_cast<const client::Cplx *>(t.Z);
    CrtDev d;
    d.v.L = c.L; d.v.words = c.words; d.v.Q = c.Q; d.v.halfQ = c.halfQ; d.v.punct = c.punct; d.v.inv = c.inv;
    d.Qd = c.Qd; d.t = c.t;
    const int logN = env.logn1 + kRowLog;
    const unsigned grid = grid_for(n_vec << logN, kBlock);
    client::Cplx *z = static_cast<client::Cplx *>(zbuf);
#define HE355_COMPOSE(W) case W: hipLaunchKernelGGL(k_ckks_decode_compose<W>, dim3(grid), dim3(kBlock), 0, env.stream, coeff, scale, z, T, env.primes, d, logN, n_vec); break;
    switch (c.words) { // words = L + 2 (DeviceContext::crt_tables), L <= 16
        HE355_COMPOSE(3) HE355_COMPOSE(4) HE355_COMPOSE(5) HE355_COMPOSE(6) HE355_COMPOSE(7) HE355_COMPOSE(8) HE355_COMPOSE(9) HE355_COMPOSE(10)
        HE355_COMPOSE(11) HE355_COMPOSE(12) HE355_COMPOSE(13) HE355_COMPOSE(14) HE355_COMPOSE(15) HE355_COMPOSE(16) HE355_COMPOSE(17) HE355_COMPOSE(18)
    default: throw std::invalid_argument("launch_ckks_decode: CRT tables of 1 to 16 data primes");
    }
#undef HE355_COMPOSE
    hipLaunchKernelGGL(k_ckks_decode_fft, dim3((unsigned)n_vec), dim3(kEncBlock), 0, env.stream, z, out, T, logN, sr);
}
void launch_bfv_encode_scatter(const KernelEnv &env, u64 n_vec, const long long *values, u64 count, u64 *ev, const uint32_t *slot_index, u64 t)
{
    if (!n_vec) return;
    const int logN = env.logn1 + kRowLog;
    hipLaunchKernelGGL(k_bfv_encode_scatter, dim3(grid_for(n_vec << logN, kBlock)), dim3(kBlock), 0, env.stream, values, count, ev, slot_index, t, logN, n_vec);
}
void launch_bfv_decode_gather(const KernelEnv &env, u64 n_vec, const u64 *ev, long long *out, const uint32_t *slot_index, u64 t, const SlotRanges &sr)
{
    if (!n_vec || !sr.total) return;
    const int logN = env.logn1 + kRowLog;
    hipLaunchKernelGGL(k_bfv_decode_gather, dim3(grid_for(n_vec * sr.total, kBlock)), dim3(kBlock), 0, env.stream, ev, out, slot_index, t, logN, n_vec, sr);
}

void launch_keygen_kswitch(const KernelEnv &env, u64 *key, u64 *e_scratch, u64 *target_scratch, const u64 *sk, const uint32_t *perm, u64 seed, u64 key_id)
{
    const int logN = env.logn1 + kRowLog, K = env.K, Ld = env.Ltop;
    hipLaunchKernelGGL(k_keygen_target, dim3(grid_for((u64)K << logN, kBlock)), dim3(kBlock), 0, env.stream, sk, perm, target_scratch, env.primes, K, logN);
    hipLaunchKernelGGL(k_keygen_sample, dim3(grid_for((u64)Ld << logN, kBlock)), dim3(kBlock), 0, env.stream, key, e_scratch, env.primes, K, logN, Ld, seed, key_id);
    PolyView v;
    v.base = e_scratch; v.item_stride = (u64)K << logN; v.polys_per_item = K; v.pad_ = 0;
    for (int i = 0; i < K; ++i) v.prime_of[i] = (unsigned char)i;
    launch_ntt_forward(env, v, (u32)Ld);
    hipLaunchKernelGGL(k_keygen_finish, dim3(grid_for(((u64)Ld * K) << logN, kBlock)), dim3(kBlock), 0, env.stream, key, e_scratch, sk, target_scratch, env.primes, K,
                       logN, Ld);
}

} // namespace HE355_KNS
} // namespace he355
