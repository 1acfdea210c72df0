// he_params.h — encryption-parameter object of the MI355X backend (host side).
//
// Mirrors what SEALContextWrapper::initCKKS / initBFV build through SEAL
// (/root/reference/src/engine/seal_context.cpp:72-127): the prime chain for {60, bits x (depth-1), 60},
// the BFV batching plain modulus, the 128-bit security gate, the NTT tables and the Galois element rules.
// It owns only host memory; device copies are made by DeviceContext (device_context.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "device_types.h"
#include "modarith.h"
#include "behz_core.h"

namespace he355 {

constexpr int kSchemeBFV = 1;  // seal::scheme_type::bfv
constexpr int kSchemeCKKS = 2; // seal::scheme_type::ckks

struct PrimeTables {
    u64 q = 0;
    int bits = 0;
    ModU64 mod{};
    u64 root = 0;       // minimal primitive 2N-th root of unity
    bool f64 = false;   // true: ArF64 engine (q < 2^47), false: ArU64
    bool fold = false;  // u64 engine: the companion word of every constant is w * 2^32 mod q (fold build, modarith.h), not the Shoup quotient
    u64 ninv = 0;       // N^-1 mod q
    std::vector<Tw16> fwd; // N entries, engine format, index = bit-reversed exponent
    std::vector<Tw16> inv; // inverse of fwd entry-wise, same indexing
    Tw16 inv_w0_scaled{};  // inv[1] * N^-1 (last inverse stage)
    std::vector<u64> fwd_u64; // plain residues of fwd (for host-side client code / checks)
    std::vector<u64> inv_u64; // plain residues of inv
    ArU64 aru() const;
    ArF64 arf() const;
};

// Per-level constants of the BEHZ multiply (RNSTool of SEAL, restated): base q = first L primes, B = B_0..B_{nB-1}, then m_sk
// (S = nB + 1 residues).  The result of the multiply is the same integer for ANY auxiliary base large enough for the
// Shenoy-Kumaresan step (Params::behz_nB), so the device picks primes of the fp64 engine; see the note at Params::aux.
struct BehzTables {
    int L = 0, nB = 0;
    std::vector<u64> inv_punct_q;  // [L]        (Q/q_i)^-1 mod q_i
    std::vector<u64> mtilde_q;     // [L]        2^32 mod q_i
    std::vector<u64> q2bsk;        // [S][L]     (Q/q_i) mod p_j   (p_j: B_0..B_{nB-1}, m_sk)
    std::vector<u64> q2mt;         // [L]        (Q/q_i) mod 2^32
    u64 neg_inv_q_mod_mt = 0;      //            -Q^-1 mod 2^32
    std::vector<u64> q_mod_bsk;    // [S]
    std::vector<u64> inv_mt_bsk;   // [S]        (2^32)^-1 mod p_j
    std::vector<u64> inv_q_bsk;    // [S]        Q^-1 mod p_j
    std::vector<u64> t_mod_q;      // [L]
    std::vector<u64> t_mod_bsk;    // [S]
    std::vector<u64> inv_punct_B;  // [nB]       (B/b_i)^-1 mod b_i
    std::vector<u64> B2q;          // [L][nB]    (B/b_i) mod q_j
    std::vector<u64> B2msk;        // [nB]       (B/b_i) mod m_sk
    u64 inv_B_mod_msk = 0;
    std::vector<u64> B_mod_q;      // [L]
};

// The constants the BEHZ kernels run on (behz_core.h, BehzDev), derived from BehzTables with every chain of constant factors folded into
// one, as two flat arrays (u64 words and their double twins for the fp64 engine) plus the offsets of each table in them.  view() makes
// the BehzDev whose pointers address copies of the two arrays at `w` / `d`: device buffers for the product (he355_api.hip), the host
// vectors themselves for the lane simulator (tests/csim/sim_behz.cpp).
struct BehzHost {
    int L = 0, nB = 0, f64aux = 0;
    u64 neg_inv_q_mod_mt = 0, neg_inv_B = 0;
    std::vector<u64> words;
    std::vector<double> doubles;
    size_t o_cq = 0, o_q2m = 0, o_e2b = 0, o_eqm = 0, o_fcq = 0, o_fds = 0, o_fng = 0, o_am = 0, o_B2q = 0, o_Bq = 0;
    size_t q_fcq = 0, q_fds = 0, q_fng = 0, q_fnh = 0, q_am = 0, q_B2q = 0, q_Bq = 0;
    BehzDev view(const u64 *w, const double *d, size_t K) const; // K: device prime index of m_sk (B_j follow at K + 1 + j)
};

class Params {
public:
    // bit_sizes is the key-level chain; sec128 enforces SEAL's tc128 cap.  Throws std::invalid_argument.
    // allow_fold = false: the Shoup form of the u64 engine's constants whatever the primes (u64_fold below; the lane simulator's Shoup build)
    static Params *create(int scheme, size_t N, const std::vector<int> &bit_sizes, int plain_bits, bool sec128, bool allow_fold = true);
    static Params *create_primes(int scheme, size_t N, const std::vector<u64> &primes, u64 plain_modulus, bool allow_fold = true);

    int scheme = 0;
    size_t N = 0;
    int logn = 0;
    int logn1 = 0;      // N = 2^logn1 * 1024
    size_t K = 0;       // all primes; the special prime is last
    size_t Ltop = 0;    // data residues at the first level
    u64 plain_modulus = 0;
    std::vector<PrimeTables> primes;
    // Every prime the u64 engine owns in this context (chain and auxiliary base) is 2^60 - c with c < 2^26 (fold_prime_ok): the tables
    // hold fold companion words and the device runs the HE355_U64_FOLD build of the kernels.  False as soon as one such prime is of
    // another shape (coefficient bits 47..59, SEAL's 61-bit auxiliary base, HE355_FORCE_U64): then every u64-engine prime takes the
    // Shoup form.  HE355_FORCE_U64=shoup: the Shoup form with the default engine assignment (A/B and the test matrix).
    bool u64_fold = false;
    // BFV only: auxiliary BEHZ base: m_sk, then B_0..B_{nB(Ltop)-1}.  Device prime index of m_sk is K, of B_i is K + 1 + i.
    // SEAL's RNSTool takes 61-bit primes (get_primes(2N, 61, |q| + 2) -> m_sk, gamma, B...; |B| = |q|).  The product of the
    // multiply does not depend on that choice: every step up to the fast floor is a modular identity in each auxiliary prime
    // (the extended operands, their products and V = floor(t Z / Q) - beta are integers fixed by the base q and m_tilde = 2^32
    // alone), and the Shenoy-Kumaresan step returns V mod q_j exactly as soon as nB + |V| / B <= (m_sk - 1) / 2
    // (behz_base_suffices: the bound in integers; SEAL's own sizing rule is the same bound with N rounded up to 2^32).  So the
    // device uses kAuxBits-bit primes, which the fp64 engine owns, and as many of them as the bound asks for (behz_nB) --
    // for the reference's parameter sets the same NUMBER of auxiliary residues as SEAL, each at the fp64 engine's cost.
    // HE355_BEHZ_BASE=seal selects SEAL's 61-bit base (A/B and cross-check; same bits out).
    static constexpr int kAuxBits = 46;
    std::vector<PrimeTables> aux; // [0] = m_sk, [1 + i] = B_i
    bool aux_seal_base = false;
    u64 gamma = 0;                // second prime of SEAL's 61-bit search (its decryption uses it; not used on the device)
    size_t behz_nB(int L) const;  // number of B primes at level L
    bool behz_base_suffices(int L, size_t nB) const; // the exact bound (he_params.cpp)

    u64 modulus(size_t i) const { return primes[i].q; }
    // GaloisTool rules (generator 3)
    uint32_t galois_elt_from_step(int step) const;        // 0 if |step| >= N/2 (SEAL throws)
    std::vector<uint32_t> galois_elts_all() const;
    // permutation tables: out[i] = in[perm[i]] (NTT form); coefficient form: out[idx[i]] = +-in[i]
    std::vector<uint32_t> galois_perm_ntt(uint32_t elt) const;
    // coefficient form: out[o] = (+/-) in[src & 0x7fffffff], negated when bit 31 of the entry is set
    std::vector<uint32_t> galois_gather_coeff(uint32_t elt) const;
    BehzTables behz_tables(int L) const;
    BehzHost behz_host(int L) const; // folded constants of level L (throws std::invalid_argument beyond kBehzMaxL / kBehzMaxB)

    // number theory helpers (also used by the client-side code)
    static bool is_prime(u64 v);
    static std::vector<u64> get_primes(u64 factor, int bit_size, size_t count);
    static u64 powmod(u64 b, u64 e, u64 q);
    static u64 invmod(u64 a, u64 q) { return powmod(a, q - 2, q); }
    static int tc128_max_bits(size_t N);
    // NTT tables of one prime q = 1 (mod 2N) (also used for the BFV plain modulus on the client side)
    static PrimeTables make_prime_tables(u64 q, size_t N, int logn, bool f64, bool fold = false);

private:
    Params() = default;
    void build(int scheme, size_t N, const std::vector<u64> &primes, u64 plain_modulus, bool allow_fold);
};

ModU64 make_mod(u64 q);
uint32_t bitrev(uint32_t x, int bits);

} // namespace he355
