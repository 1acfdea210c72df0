// behz_core.h -- the per-coefficient arithmetic of the BEHZ multiply (base extension with the Montgomery correction; times t, fast
// floor, Shenoy-Kumaresan) and the constants it runs on.  Host-compilable on purpose: the HIP kernels (he355_kernels.hip: k_behz_extend,
// k_behz_extend_cols, k_behz_floor_sk, k_behz_cols_floor_sk) and the test-only lane simulator (tests/csim/sim_behz.cpp, which holds
// these very functions to exact integer arithmetic on the CPU, extremes of the Shenoy-Kumaresan bound included) compile the same text.
#pragma once
#include "device_types.h"

namespace he355 {

// Device constants of the BEHZ multiply, derived from BehzTables (he_params.h) with every chain of constant factors folded into one
// (he355_api.hip, DeviceContext::behz): each step is then ONE 128-bit sum of products and ONE Barrett reduction -- same residues.
// All pointers are HBM.  Bsk order: B_0..B_{nB-1}, m_sk (S = nB + 1); p_j the j-th of them.
struct BehzDev {
    int L, nB;
    // steps (1)-(2), base extension q -> Bsk with the Montgomery correction mod m_tilde = 2^32:
    //   tmp_i = x_i cq_i mod q_i;  r = -(sum_i tmp_i q2mt_i) Q^-1 mod 2^32, centred;  out_j = sum_i tmp_i e_q2bsk_ji + r e_qmod_j mod p_j
    const u64 *cq;          // [L]      2^32 (Q/q_i)^-1 mod q_i
    const u64 *q2mt;        // [L]      (Q/q_i) mod 2^32
    u64 neg_inv_q_mod_mt;   //          -Q^-1 mod 2^32
    const u64 *e_q2bsk;     // [S][L]   (Q/q_i) 2^-32 mod p_j
    const u64 *e_qmod;      // [S]      Q 2^-32 mod p_j
    // steps (6)-(7), times t and fast floor, with (B/b_j)^-1 of step (8) folded in for j < nB:
    //   tmp_i = d_i f_cq_i mod q_i;  fl_j = ds_j f_ds_j + sum_i tmp_i f_neg_ji mod p_j
    const u64 *f_cq;        // [L]      t (Q/q_i)^-1 mod q_i
    const u64 *f_ds;        // [S]      t Q^-1 c_j mod p_j,  c_j = (B/b_j)^-1 mod b_j (j < nB), 1 (m_sk)
    const u64 *f_neg;       // [S][L]   -(Q/q_i) Q^-1 c_j mod p_j
    // step (8), Shenoy-Kumaresan: alpha = sum_j fl_j a_msk_j + fl_sk neg_inv_B mod m_sk, centred;
    //   out_i = sum_j fl_j B2q_ij - alpha B mod q_i
    const u64 *a_msk;       // [nB]     (B/b_j) B^-1 mod m_sk
    u64 neg_inv_B;          //          -B^-1 mod m_sk
    const u64 *B2q;         // [L][nB]  (B/b_j) mod q_i
    const u64 *B_mod_q;     // [L]
    unsigned char bsk_prime[64]; // device prime index of Bsk element j
    // The same constants of steps (6)-(8) as doubles, for the fp64 engine: used when every auxiliary prime is below 2^47 (f64aux;
    // the device's own base).  A residue mod an auxiliary prime is then a handful of exact fp64 products (ArF64::mulmod_vv) instead of
    // 128-bit sums and a Barrett reduction; base-q residues of fp64-engine primes likewise, those of the 60-bit primes stay integers
    // (their value enters the auxiliary sums as hi * 2^30 + lo: f_neg_hi_d = f_neg * 2^30 mod p_j).
    int f64aux, pad_;
    const double *f_cq_d;     // [L]
    const double *f_ds_d;     // [S]
    const double *f_neg_d;    // [S][L]
    const double *f_neg_hi_d; // [S][L]
    const double *a_msk_d;    // [nB]
    double neg_inv_B_d;
    const double *B2q_d;      // [L][nB]
    const double *B_mod_q_d;  // [L]
};
constexpr int kBehzMaxL = 16; // base q
constexpr int kBehzMaxB = 24; // base B (Params::behz_nB: 22 for sixteen 60-bit primes)

HE_HD ModU64 behz_modu(const PrimeDev &p)
{
    ModU64 m;
    m.q = p.q; m.cr0 = p.cr0; m.cr1 = p.cr1;
    return m;
}
HE_HD ModU64 behz_modu_at(const PrimeDev *primes, int idx) { return behz_modu(primes[idx]); }
HE_HD ArF64 behz_arf(const PrimeDev &p)
{
    ArF64 a;
    a.q = p.qd; a.qinv = p.qinv; a.ninv = p.ninv_d; a.ninv_i = p.ninv_i;
    return a;
}

// One coefficient through fastbconv_m_tilde's first half: tmp_i = x_i m_tilde (Q/q_i)^-1 mod q_i, and r = -(sum tmp_i Q/q_i) Q^-1 mod m_tilde
template <int ML> HE_HD void behz_ext_prepare(const BehzDev &Z, const PrimeDev *primes, int L, const u64 x[ML], u64 tmp[ML], u64 &rmt)
{
    u64 mt_acc = 0;
#pragma unroll
    for (int i = 0; i < ML; ++i)
        if (i < L) {
            tmp[i] = mulmod(x[i], Z.cq[i], behz_modu_at(primes, i));
            mt_acc += (tmp[i] & 0xFFFFFFFFull) * Z.q2mt[i];
        }
    rmt = ((mt_acc & 0xFFFFFFFFull) * Z.neg_inv_q_mod_mt) & 0xFFFFFFFFull;
}
// ... and its residue mod Bsk element j after sm_mrq, (sum tmp_i (Q/q_i) + r Q) m_tilde^-1 mod p_j with r centred: one sum of
// products (constants carry m_tilde^-1; at most 17 terms below 2^122), one reduction
template <int ML> HE_HD u64 behz_ext_residue(const BehzDev &Z, const ModU64 &mj, int L, int j, const u64 tmp[ML], u64 rmt)
{
    const u64 MT = (u64)1 << 32;
    const u64 rr = rmt >= (MT >> 1) ? rmt + (mj.q - MT) : rmt; // centred r as a residue mod p_j
    u128 acc = (u128)rr * Z.e_qmod[j];
#pragma unroll
    for (int i = 0; i < ML; ++i)
        if (i < L) acc += (u128)tmp[i] * Z.e_q2bsk[j * L + i];
    return barrett128(acc, mj);
}
// BEHZ steps (6)-(8) for one coefficient: dq[i] (base q), ds[j] (Bsk) canonical residues of a product -> its L output residues.
// Constant factors are folded (BehzDev): every line below is one sum of products below 2^122 (at most 25 of them) and one reduction.
template <int ML, int MB>
HE_HD void behz_floor_sk_coeff(const BehzDev &Z, const PrimeDev *primes, int L, int nB, const u64 dq[ML], const u64 ds[MB + 1], u64 res[ML])
{
    [[maybe_unused]] constexpr int kUnrollB = MB <= 6 ? MB + 1 : 1, kUnrollL = ML <= 4 ? ML : 1; // the large instantiation keeps its residue loops rolled
    const int S = nB + 1;
    u64 tmp[ML], fl[MB];
    // (6) times t, and the base-q part prepared for the fast conversion (canonical: the conversion depends on the representative)
#pragma unroll
    for (int i = 0; i < ML; ++i)
        if (i < L) tmp[i] = mulmod(dq[i], Z.f_cq[i], behz_modu_at(primes, i));
    // (7) fast floor (t x_Bsk - FastBconv(t x_q)) Q^-1 mod p_j, times (B/b_j)^-1 for the B part; (8) first half: alpha_sk
    const ModU64 msk = behz_modu_at(primes, Z.bsk_prime[nB]);
    u128 accs = 0;
#pragma unroll kUnrollB
    for (int j = 0; j < MB + 1; ++j)
        if (j < S) {
            const ModU64 mj = behz_modu_at(primes, Z.bsk_prime[j]);
            u128 acc = (u128)ds[j] * Z.f_ds[j];
#pragma unroll
            for (int i = 0; i < ML; ++i)
                if (i < L) acc += (u128)tmp[i] * Z.f_neg[j * L + i];
            const u64 f = barrett128(acc, mj);
            if (j < nB) {
                if (j < MB) { // (always; keeps the index static)
                    fl[j] = f;
                    accs += (u128)f * Z.a_msk[j];
                }
            } else {
                accs += (u128)f * Z.neg_inv_B;
            }
        }
    const u64 alpha = barrett128(accs, msk);
    const bool neg = alpha > (msk.q >> 1);
    // (8) second half: B -> q with the alpha_sk correction
#pragma unroll kUnrollL
    for (int j = 0; j < ML; ++j) {
        if (j >= L) break;
        const ModU64 mj = behz_modu_at(primes, j);
        const u64 Bq = Z.B_mod_q[j];
        u128 acc = neg ? (u128)(msk.q - alpha) * Bq : (u128)alpha * (Bq ? mj.q - Bq : 0);
#pragma unroll
        for (int i = 0; i < MB; ++i)
            if (i < nB) acc += (u128)fl[i] * Z.B2q[j * nB + i];
        res[j] = barrett128(acc, mj);
    }
}
// The same steps on the fp64 engine, for an auxiliary base of primes below 2^47 (BehzDev::f64aux): a residue mod an auxiliary prime,
// alpha_sk and the outputs under fp64-engine base-q primes are sums of exact fp64 products (ArF64::mulmod_vv: centred, |.| <= p (1/2 +
// eps); at most 27 of them, canon() brings the sum home); the 60-bit base-q primes keep their integer arithmetic and enter the
// auxiliary sums as hi * 2^30 + lo.  The same residues as behz_floor_sk_coeff, at a third of its instructions.
template <int ML, int MB>
HE_HD void behz_floor_sk_coeff_f64(const BehzDev &Z, const PrimeDev *primes, int L, int nB, const u64 dq[ML], const u64 ds[MB + 1],
                                                        u64 res[ML])
{
    [[maybe_unused]] constexpr int kUnrollB = MB <= 6 ? MB + 1 : 1, kUnrollL = ML <= 4 ? ML : 1;
    const int S = nB + 1;
    double ta[ML], tb[ML]; // tmp_i: fp64-engine prime: ta = the canonical value; 60-bit prime: ta = its low 30 bits, tb = the rest
    // (6) times t, and the base-q part prepared for the fast conversion (canonical: the conversion depends on the representative)
#pragma unroll
    for (int i = 0; i < ML; ++i)
        if (i < L) {
            const PrimeDev &Pi = primes[i];
            if (Pi.f64) {
                const ArF64 ar = behz_arf(Pi);
                ta[i] = ar.canon2(ar.mulmod_vv(u52_to_f64(dq[i]), Z.f_cq_d[i]));
                tb[i] = 0.0;
            } else {
                const u64 t = mulmod(dq[i], Z.f_cq[i], behz_modu(Pi));
                ta[i] = u52_to_f64(t & (((u64)1 << 30) - 1));
                tb[i] = u52_to_f64(t >> 30);
            }
        }
    // (7) fast floor times (B/b_j)^-1, (8) first half: alpha_sk
    const ArF64 arsk = behz_arf(primes[Z.bsk_prime[nB]]);
    double fl[MB], acc_sk = 0.0;
#pragma unroll kUnrollB
    for (int j = 0; j < MB + 1; ++j)
        if (j < S) {
            const ArF64 arj = behz_arf(primes[Z.bsk_prime[j]]);
            double sum = arj.mulmod_vv(u52_to_f64(ds[j]), Z.f_ds_d[j]);
#pragma unroll
            for (int i = 0; i < ML; ++i)
                if (i < L) {
                    sum += arj.mulmod_vv(ta[i], Z.f_neg_d[j * L + i]);
                    if (!primes[i].f64) sum += arj.mulmod_vv(tb[i], Z.f_neg_hi_d[j * L + i]);
                }
            const double f = arj.canon(sum);
            if (j < nB) {
                if (j < MB) { // (always; keeps the index static)
                    fl[j] = f;
                    acc_sk += arsk.mulmod_vv(f, Z.a_msk_d[j]);
                }
            } else {
                acc_sk += arsk.mulmod_vv(f, Z.neg_inv_B_d);
            }
        }
    const double alpha = arsk.canon(acc_sk);
    const bool neg = alpha > (arsk.q - 1.0) * 0.5; // alpha > floor(m_sk / 2), m_sk odd
    const double am = neg ? arsk.q - alpha : alpha; // |gamma|
    // (8) second half: B -> q with the alpha_sk correction
#pragma unroll kUnrollL
    for (int i = 0; i < ML; ++i) {
        if (i >= L) break;
        const PrimeDev &Pi = primes[i];
        if (Pi.f64) {
            const ArF64 ar = behz_arf(Pi);
            const double Bq = Z.B_mod_q_d[i];
            double sum = ar.mulmod_vv(am, neg ? Bq : ar.q - Bq);
#pragma unroll
            for (int j = 0; j < MB; ++j)
                if (j < nB) sum += ar.mulmod_vv(fl[j], Z.B2q_d[i * nB + j]);
            res[i] = f64_to_u52(ar.canon(sum));
        } else {
            const ModU64 mi = behz_modu(Pi);
            const u64 Bq = Z.B_mod_q[i];
            u128 acc = (u128)f64_to_u52(am) * (neg ? Bq : (Bq ? mi.q - Bq : 0));
#pragma unroll
            for (int j = 0; j < MB; ++j)
                if (j < nB) acc += (u128)f64_to_u52(fl[j]) * Z.B2q[i * nB + j];
            res[i] = barrett128(acc, mi);
        }
    }
}

} // namespace he355
