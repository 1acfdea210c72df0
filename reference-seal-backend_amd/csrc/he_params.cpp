// he_params.cpp — see he_params.h.  Host-only C++17; no HIP calls here.
#include "he_params.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

namespace he355 {

ModU64 make_mod(u64 q)
{
    ModU64 m;
    m.q = q;
    const u128 one64 = (u128)1 << 64;
    const u128 hi = one64 / q, rem = one64 % q;
    m.cr1 = (u64)hi;
    m.cr0 = (u64)((rem << 64) / q);
    return m;
}

uint32_t bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i, x >>= 1) r = (r << 1) | (x & 1);
    return r;
}

static inline u64 mm(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }

u64 Params::powmod(u64 b, u64 e, u64 q)
{
    u64 r = 1 % q;
    b %= q;
    for (; e; e >>= 1, b = mm(b, b, q))
        if (e & 1) r = mm(r, b, q);
    return r;
}

bool Params::is_prime(u64 n)
{
    // deterministic Miller-Rabin for 64-bit inputs
    if (n < 2) return false;
    for (u64 p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) d >>= 1, ++s;
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int k = 1; k < s && witness; ++k) {
            x = mm(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

// The search rule behind CoeffModulus::Create / PlainModulus::Batching: candidates = 1 (mod factor),
// descending from just below 2^bit_size, down to 2^(bit_size-1).
std::vector<u64> Params::get_primes(u64 factor, int bit_size, size_t count)
{
    std::vector<u64> out;
    u64 v = (((u64)1 << bit_size) - 1) / factor * factor + 1;
    const u64 floor_v = (u64)1 << (bit_size - 1);
    for (; out.size() < count && v > floor_v; v -= factor)
        if (is_prime(v)) out.push_back(v);
    if (out.size() != count) throw std::logic_error("failed to find enough qualifying primes");
    return out;
}

int Params::tc128_max_bits(size_t N)
{
    switch (N) {
    case 1024: return 27;
    case 2048: return 54;
    case 4096: return 109;
    case 8192: return 218;
    case 16384: return 438;
    case 32768: return 881;
    default: return 0;
    }
}

Params *Params::create(int scheme, size_t N, const std::vector<int> &bit_sizes, int plain_bits, bool sec128, bool allow_fold)
{
    if (bit_sizes.empty() || bit_sizes.size() >= (size_t)kMaxPrimes) throw std::invalid_argument("invalid coefficient modulus count");
    if (sec128) {
        int total = 0;
        for (int b : bit_sizes) total += b;
        const int cap = tc128_max_bits(N);
        if (cap == 0 || total > cap) throw std::invalid_argument("encryption parameters are not set correctly: coefficient modulus too large for 128-bit security");
    }
    // one descending list per distinct bit size; slots are served from the back of the list
    std::map<int, std::vector<u64>> lists;
    for (int b : bit_sizes) {
        if (b < 2 || b > 60) throw std::invalid_argument("coefficient modulus bit sizes must be in [2, 60]");
        if (!lists.count(b)) lists[b] = get_primes(2 * (u64)N, b, (size_t)std::count(bit_sizes.begin(), bit_sizes.end(), b));
    }
    std::vector<u64> chain;
    for (int b : bit_sizes) {
        chain.push_back(lists[b].back());
        lists[b].pop_back();
    }
    u64 t = 0;
    if (scheme == kSchemeBFV) t = get_primes(2 * (u64)N, plain_bits, 1)[0];
    return create_primes(scheme, N, chain, t, allow_fold);
}

Params *Params::create_primes(int scheme, size_t N, const std::vector<u64> &primes, u64 plain_modulus, bool allow_fold)
{
    Params *p = new Params();
    try {
        p->build(scheme, N, primes, plain_modulus, allow_fold);
    } catch (...) {
        delete p;
        throw;
    }
    return p;
}

static u64 minimal_primitive_root(u64 two_n, u64 q)
{
    if ((q - 1) % two_n) throw std::invalid_argument("coefficient modulus is not NTT-friendly");
    const u64 e = (q - 1) / two_n;
    u64 root = 0;
    for (u64 g = 2; g < 4096 && !root; ++g) {
        u64 r = Params::powmod(g, e, q);
        if (Params::powmod(r, two_n / 2, q) == q - 1) root = r;
    }
    if (!root) throw std::invalid_argument("no primitive root found");
    // all primitive 2N-th roots are the odd powers of one of them; keep the smallest
    const u64 step = mm(root, root, q);
    u64 cur = root, best = root;
    for (u64 i = 1; i < two_n / 2; ++i) {
        cur = mm(cur, step, q);
        best = std::min(best, cur);
    }
    return best;
}

static Tw16 make_tw(u64 w, u64 q, bool f64, bool fold)
{
    Tw16 t;
    if (f64) {
        double wd = (double)w, wi = (double)w / (double)q;
        std::memcpy(&t.a, &wd, 8);
        std::memcpy(&t.b, &wi, 8);
    } else {
        t.a = w;
        t.b = pre_word(w, q, fold);
    }
    return t;
}

ArU64 PrimeTables::aru() const
{
    ArU64 a;
    a.q = q;
    a.two_q = 2 * q;
    a.ninv = ninv;
    a.ninv_q = pre_word(ninv, q, fold);
    a.cr0 = mod.cr0;
    a.cr1 = mod.cr1;
    return a;
}
ArF64 PrimeTables::arf() const
{
    ArF64 a;
    a.q = (double)q;
    a.qinv = 1.0 / (double)q;
    a.ninv = (double)ninv;
    a.ninv_i = (double)ninv / (double)q;
    return a;
}

void Params::build(int scheme_, size_t N_, const std::vector<u64> &chain, u64 t, bool allow_fold)
{
    if (scheme_ != kSchemeBFV && scheme_ != kSchemeCKKS) throw std::invalid_argument("unsupported scheme");
    int ln = 0;
    while (((size_t)1 << ln) < N_) ++ln;
    if (((size_t)1 << ln) != N_ || N_ < 1024 || N_ > 32768) throw std::invalid_argument("poly_modulus_degree must be a power of two in [1024, 32768]");
    if (chain.empty() || chain.size() >= (size_t)kMaxPrimes) throw std::invalid_argument("invalid coefficient modulus count");
    scheme = scheme_;
    N = N_;
    logn = ln;
    logn1 = ln - 10;
    K = chain.size();
    Ltop = K > 1 ? K - 1 : 1;
    plain_modulus = t;
    if (scheme == kSchemeBFV && t < 2) throw std::invalid_argument("BFV needs a plain modulus");
    const char *force = std::getenv("HE355_FORCE_U64");
    const bool force_u64 = force && force[0] == '1', force_shoup = force && force[0] == 's';
    const char *base = std::getenv("HE355_BEHZ_BASE");
    aux_seal_base = scheme == kSchemeBFV && base && std::string(base) == "seal";
    primes.resize(K);
    for (size_t i = 0; i < K; ++i) {
        const u64 q = chain[i];
        for (size_t k = 0; k < i; ++k)
            if (chain[k] == q) throw std::invalid_argument("coefficient moduli must be distinct");
        // SEAL's user moduli are at most 60 bits (SEAL_USER_MOD_BIT_COUNT_MAX); the u64 engine's wide lazy row pass needs 16 q <= 2^64
        if (q >> 60 || !is_prime(q)) throw std::invalid_argument("coefficient modulus must be a prime below 2^60");
        primes[i].q = q; // (behz_base_suffices below reads the moduli; the tables follow once the form of their constants is known)
    }
    if (scheme == kSchemeBFV) {
        if (t >= ((u64)1 << 32)) throw std::invalid_argument("plain modulus too large for the BEHZ base");
        // RNSTool::initialize: get_primes(2N, 61, |B| + 2) -> m_sk, gamma, B...  with |B| = |q| at every level
        // (the +1 case needs 32 + bits(t) + bits(q) >= 61|q| + 61, impossible for user primes <= 60 bits)
        const std::vector<u64> ap = get_primes(2 * (u64)N, 61, Ltop + 2);
        gamma = ap[1];
        if (!aux_seal_base) {
            // the device's own base (he_params.h, Params::aux): kAuxBits-bit primes 1 (mod 2N), none of them a coefficient modulus,
            // as many as the Shenoy-Kumaresan bound asks for at the first level (behz_base_suffices).  It needs about 1.3 |q| primes
            // where SEAL's needs |q| + 1: a chain so long that they no longer fit the device prime table (or the kernels' base-B
            // limit), or that exhausts the candidates, takes SEAL's base instead.
            std::vector<u64> cand = get_primes(2 * (u64)N, kAuxBits, 3 * K + 16);
            cand.erase(std::remove_if(cand.begin(), cand.end(), [&](u64 v) { return std::find(chain.begin(), chain.end(), v) != chain.end(); }), cand.end());
            size_t next = 0;
            bool ok = !cand.empty();
            if (ok) aux.push_back(make_prime_tables(cand[next++], N, logn, !force_u64)); // m_sk
            while (ok) {
                if (next == cand.size() || aux.size() > (size_t)kBehzMaxB || K + aux.size() + 2 > (size_t)kMaxPrimes) { ok = false; break; }
                aux.push_back(make_prime_tables(cand[next++], N, logn, !force_u64));
                if (behz_base_suffices((int)Ltop, aux.size() - 1)) break;
            }
            if (!ok) { aux.clear(); aux_seal_base = true; }
        }
        if (aux_seal_base) {
            aux.push_back(make_prime_tables(ap[0], N, logn, false));
            for (size_t i = 0; i < Ltop; ++i) aux.push_back(make_prime_tables(ap[2 + i], N, logn, false));
        }
        // + 1: the plain modulus rides along in the device prime array
        if (K + aux.size() + 1 > (size_t)kMaxPrimes) throw std::invalid_argument("too many primes for the device prime table");
    }
    // the form of the u64 engine's constants (he_params.h, u64_fold), then the chain's tables
    u64_fold = allow_fold && !force_u64 && !force_shoup && !aux_seal_base;
    for (size_t i = 0; i < K && u64_fold; ++i)
        if ((chain[i] >> 47) != 0 && !fold_prime_ok(chain[i])) u64_fold = false;
    for (size_t i = 0; i < K; ++i) primes[i] = make_prime_tables(chain[i], N, logn, !force_u64 && (chain[i] >> 47) == 0, u64_fold);
}

namespace {
// little-endian multi-word unsigned integer: just enough for the one inequality below
struct Wide {
    std::vector<u64> w{1};
    void mul(u64 m)
    {
        u64 carry = 0;
        for (u64 &x : w) {
            const u128 p = (u128)x * m + carry;
            x = (u64)p;
            carry = (u64)(p >> 64);
        }
        if (carry) w.push_back(carry);
    }
    void add(const Wide &o)
    {
        u64 carry = 0;
        if (w.size() < o.w.size()) w.resize(o.w.size(), 0);
        for (size_t i = 0; i < w.size(); ++i) {
            const u128 p = (u128)w[i] + (i < o.w.size() ? o.w[i] : 0) + carry;
            w[i] = (u64)p;
            carry = (u64)(p >> 64);
        }
        if (carry) w.push_back(carry);
    }
    bool less_than(const Wide &o) const
    {
        size_t na = w.size(), nb = o.w.size();
        while (na > 1 && !w[na - 1]) --na;
        while (nb > 1 && !o.w[nb - 1]) --nb;
        if (na != nb) return na < nb;
        for (size_t i = na; i-- > 0;)
            if (w[i] != o.w[i]) return w[i] < o.w[i];
        return false;
    }
};
} // namespace

// Is {B_0..B_{nB-1}}, m_sk large enough for the BEHZ multiply at level L?  With Q = q_0..q_{L-1}, m_tilde = 2^32: the extended
// operands satisfy |Y| <= Q (1/2 + L/2^32), the largest product coefficient (c1 = a0 b1 + a1 b0) is |Z| <= 2 N Y^2, and the fast
// floor yields the integer V = (t Z - W) / Q with 0 <= W < L Q, so |V| <= t N Q (1 + L/2^31)^2 / 2 + L.  The Shenoy-Kumaresan step
// returns V mod q_j exactly iff gamma = (V' - V) / B, which lies in (-|V|/B, nB + |V|/B), is told apart by its centred residue
// mod m_sk: nB + |V|/B <= (m_sk - 1) / 2.  Sufficient, in integers (both sides times 2^62):
//     t N Q (2^31 + L)^2 + 2 L 2^62  <  B (m_sk - 1 - 2 nB) 2^62.
// (SEAL's own rule, 32 + bits(t) + bits(Q) < bits(B m_sk), is this inequality with N (1 + L/2^31)^2 rounded up to 2^32.)
bool Params::behz_base_suffices(int L, size_t nB) const
{
    if (nB < 1 || 1 + nB > aux.size() || aux[0].q <= 1 + 2 * (u64)nB) return false;
    Wide lhs, rhs;
    lhs.mul(plain_modulus);
    lhs.mul((u64)N);
    for (int i = 0; i < L; ++i) lhs.mul(primes[i].q);
    lhs.mul(((u64)1 << 31) + (u64)L);
    lhs.mul(((u64)1 << 31) + (u64)L);
    Wide extra;
    extra.mul(2 * (u64)L);
    extra.mul((u64)1 << 62);
    lhs.add(extra);
    for (size_t i = 0; i < nB; ++i) rhs.mul(aux[1 + i].q);
    rhs.mul(aux[0].q - 1 - 2 * (u64)nB);
    rhs.mul((u64)1 << 62);
    return lhs.less_than(rhs);
}

size_t Params::behz_nB(int L) const
{
    if (aux_seal_base) return (size_t)L;
    for (size_t nB = 1; 1 + nB <= aux.size(); ++nB)
        if (behz_base_suffices(L, nB)) return nB;
    throw std::logic_error("auxiliary BEHZ base too small");
}

std::vector<uint32_t> Params::galois_gather_coeff(uint32_t elt) const
{
    // GaloisTool::apply_galois maps in[i] to out[(i*g mod 2N) mod N], negated when i*g mod 2N >= N; as a gather:
    // i0 = o * g^-1 mod 2N; i0 < N -> +in[i0], else -in[i0 - N]
    const u64 m = 2 * (u64)N;
    u64 ginv = 1;
    for (u64 x = 1; x < m; x += 2)
        if (((x * elt) & (m - 1)) == 1) { ginv = x; break; }
    std::vector<uint32_t> g(N);
    for (size_t o = 0; o < N; ++o) {
        const u64 i0 = ((u64)o * ginv) & (m - 1);
        g[o] = i0 < N ? (uint32_t)i0 : ((uint32_t)(i0 - N) | 0x80000000u);
    }
    return g;
}

BehzTables Params::behz_tables(int L_) const
{
    if (scheme != kSchemeBFV || L_ < 1 || (size_t)L_ > Ltop) throw std::invalid_argument("BEHZ tables need a BFV level");
    const size_t L = (size_t)L_, nB = behz_nB(L_), S = nB + 1;
    BehzTables T;
    T.L = L_;
    T.nB = (int)nB;
    auto qv = [&](size_t i) { return primes[i].q; };
    auto bv = [&](size_t i) { return aux[1 + i].q; };
    auto bsk = [&](size_t j) { return j < nB ? bv(j) : aux[0].q; }; // B_0..B_{nB-1}, m_sk
    const u64 MT = (u64)1 << 32, t = plain_modulus;
    T.inv_punct_q.resize(L); T.mtilde_q.resize(L); T.q2mt.resize(L); T.t_mod_q.resize(L); T.B_mod_q.resize(L);
    T.q2bsk.resize(S * L); T.q_mod_bsk.resize(S); T.inv_mt_bsk.resize(S); T.inv_q_bsk.resize(S); T.t_mod_bsk.resize(S);
    T.inv_punct_B.resize(nB); T.B2q.resize(L * nB); T.B2msk.resize(nB);
    for (size_t i = 0; i < L; ++i) {
        u64 p = 1, pm = 1;
        for (size_t k = 0; k < L; ++k)
            if (k != i) {
                p = mm(p, qv(k) % qv(i), qv(i));
                pm = (pm * (qv(k) & 0xFFFFFFFFull)) & 0xFFFFFFFFull;
            }
        T.inv_punct_q[i] = invmod(p, qv(i));
        T.q2mt[i] = pm;
        T.mtilde_q[i] = MT % qv(i);
        T.t_mod_q[i] = t % qv(i);
    }
    u64 qm = 1;
    for (size_t k = 0; k < L; ++k) qm = (qm * (qv(k) & 0xFFFFFFFFull)) & 0xFFFFFFFFull;
    u64 inv = qm;
    for (int it = 0; it < 6; ++it) inv = (inv * (2 - qm * inv)) & 0xFFFFFFFFull;
    T.neg_inv_q_mod_mt = (MT - inv) & 0xFFFFFFFFull;
    for (size_t j = 0; j < S; ++j) {
        const u64 pj = bsk(j);
        u64 all = 1;
        for (size_t k = 0; k < L; ++k) all = mm(all, qv(k) % pj, pj);
        T.q_mod_bsk[j] = all;
        T.inv_q_bsk[j] = invmod(all, pj);
        T.inv_mt_bsk[j] = invmod(MT % pj, pj);
        T.t_mod_bsk[j] = t % pj;
        for (size_t i = 0; i < L; ++i) {
            u64 p = 1;
            for (size_t k = 0; k < L; ++k)
                if (k != i) p = mm(p, qv(k) % pj, pj);
            T.q2bsk[j * L + i] = p;
        }
    }
    const u64 msk = aux[0].q;
    {
        u64 all = 1;
        for (size_t k = 0; k < nB; ++k) all = mm(all, bv(k) % msk, msk);
        T.inv_B_mod_msk = invmod(all, msk);
    }
    for (size_t i = 0; i < nB; ++i) {
        const u64 bi = bv(i);
        u64 p = 1, pm = 1;
        for (size_t k = 0; k < nB; ++k)
            if (k != i) {
                p = mm(p, bv(k) % bi, bi);
                pm = mm(pm, bv(k) % msk, msk);
            }
        T.inv_punct_B[i] = invmod(p, bi);
        T.B2msk[i] = pm;
    }
    for (size_t j = 0; j < L; ++j) {
        u64 all = 1;
        for (size_t k = 0; k < nB; ++k) all = mm(all, bv(k) % qv(j), qv(j));
        T.B_mod_q[j] = all;
        for (size_t i = 0; i < nB; ++i) {
            u64 p = 1;
            for (size_t k = 0; k < nB; ++k)
                if (k != i) p = mm(p, bv(k) % qv(j), qv(j));
            T.B2q[j * nB + i] = p;
        }
    }
    return T;
}

BehzHost Params::behz_host(int L) const
{
    if (L > kBehzMaxL) throw std::invalid_argument("BFV multiply supports at most 16 data primes in this build");
    const BehzTables T = behz_tables(L);
    const int nB = T.nB, S = nB + 1;
    if (nB > kBehzMaxB) throw std::invalid_argument("BFV multiply: auxiliary base too large for this build");
    auto pj = [&](int j) { return j < nB ? aux[1 + j].q : aux[0].q; };
    std::vector<u64> cq(L), f_cq(L), e_q2bsk((size_t)S * L), e_qmod(S), f_ds(S), f_neg((size_t)S * L), f_neg_hi((size_t)S * L), a_msk(nB);
    for (int i = 0; i < L; ++i) {
        const u64 q = primes[i].q;
        cq[i] = mm(T.mtilde_q[i], T.inv_punct_q[i], q);
        f_cq[i] = mm(T.t_mod_q[i], T.inv_punct_q[i], q);
    }
    for (int j = 0; j < S; ++j) {
        const u64 p = pj(j);
        const u64 c = j < nB ? mm(T.inv_q_bsk[j], T.inv_punct_B[j], p) : T.inv_q_bsk[j];
        e_qmod[j] = mm(T.q_mod_bsk[j], T.inv_mt_bsk[j], p);
        f_ds[j] = mm(T.t_mod_bsk[j], c, p);
        for (int i = 0; i < L; ++i) {
            e_q2bsk[(size_t)j * L + i] = mm(T.q2bsk[(size_t)j * L + i], T.inv_mt_bsk[j], p);
            const u64 v = mm(T.q2bsk[(size_t)j * L + i], c, p);
            f_neg[(size_t)j * L + i] = v ? p - v : 0;
            f_neg_hi[(size_t)j * L + i] = mm(f_neg[(size_t)j * L + i], ((u64)1 << 30) % p, p);
        }
    }
    const u64 msk = aux[0].q;
    for (int j = 0; j < nB; ++j) a_msk[j] = mm(T.B2msk[j], T.inv_B_mod_msk, msk);
    BehzHost H;
    H.L = L; H.nB = nB;
    H.neg_inv_q_mod_mt = T.neg_inv_q_mod_mt;
    H.neg_inv_B = T.inv_B_mod_msk ? msk - T.inv_B_mod_msk : 0;
    auto push = [&](const std::vector<u64> &v) { const size_t off = H.words.size(); H.words.insert(H.words.end(), v.begin(), v.end()); return off; };
    H.o_cq = push(cq); H.o_q2m = push(T.q2mt); H.o_e2b = push(e_q2bsk); H.o_eqm = push(e_qmod); H.o_fcq = push(f_cq); H.o_fds = push(f_ds);
    H.o_fng = push(f_neg); H.o_am = push(a_msk); H.o_B2q = push(T.B2q); H.o_Bq = push(T.B_mod_q);
    // the doubles of steps (6)-(8) (BehzDev::f64aux): used when every auxiliary prime is on the fp64 engine; entries that belong to a
    // 60-bit base-q prime are not exact as doubles and are never read (those residues keep their integer arithmetic)
    H.f64aux = 1;
    for (int j = 0; j < S; ++j) H.f64aux &= (int)(j < nB ? aux[1 + j].f64 : aux[0].f64);
    auto pushd = [&](const std::vector<u64> &v) { const size_t off = H.doubles.size(); for (u64 x : v) H.doubles.push_back((double)x); return off; };
    H.q_fcq = pushd(f_cq); H.q_fds = pushd(f_ds); H.q_fng = pushd(f_neg); H.q_fnh = pushd(f_neg_hi); H.q_am = pushd(a_msk); H.q_B2q = pushd(T.B2q);
    H.q_Bq = pushd(T.B_mod_q);
    return H;
}

BehzDev BehzHost::view(const u64 *w, const double *d, size_t K) const
{
    BehzDev Z{};
    Z.L = L; Z.nB = nB;
    Z.cq = w + o_cq; Z.q2mt = w + o_q2m; Z.neg_inv_q_mod_mt = neg_inv_q_mod_mt; Z.e_q2bsk = w + o_e2b; Z.e_qmod = w + o_eqm;
    Z.f_cq = w + o_fcq; Z.f_ds = w + o_fds; Z.f_neg = w + o_fng;
    Z.a_msk = w + o_am; Z.neg_inv_B = neg_inv_B; Z.B2q = w + o_B2q; Z.B_mod_q = w + o_Bq;
    Z.f64aux = f64aux;
    Z.f_cq_d = d + q_fcq; Z.f_ds_d = d + q_fds; Z.f_neg_d = d + q_fng; Z.f_neg_hi_d = d + q_fnh; Z.a_msk_d = d + q_am;
    Z.neg_inv_B_d = (double)neg_inv_B; Z.B2q_d = d + q_B2q; Z.B_mod_q_d = d + q_Bq;
    for (int j = 0; j < nB; ++j) Z.bsk_prime[j] = (unsigned char)(K + 1 + j); // B_j
    Z.bsk_prime[nB] = (unsigned char)K;                                        // m_sk
    return Z;
}

PrimeTables Params::make_prime_tables(u64 q, size_t N, int logn, bool f64, bool fold)
{
    if (fold && !f64 && !fold_prime_ok(q)) throw std::logic_error("fold tables asked for a prime that is not 2^60 - c");
    PrimeTables pt;
    pt.fold = fold && !f64;
    pt.q = q;
    pt.bits = 64 - __builtin_clzll(q);
    pt.mod = make_mod(q);
    pt.root = minimal_primitive_root(2 * (u64)N, q);
    pt.f64 = f64;
    pt.ninv = invmod((u64)N % q, q);
    pt.fwd.resize(N);
    pt.inv.resize(N);
    pt.fwd_u64.resize(N);
    pt.inv_u64.resize(N);
    const u64 iroot = invmod(pt.root, q);
    u64 pw = 1, ipw = 1;
    for (size_t e = 0; e < N; ++e) {
        const uint32_t k = bitrev((uint32_t)e, logn);
        pt.fwd_u64[k] = pw;
        pt.inv_u64[k] = ipw;
        pt.fwd[k] = make_tw(pw, q, pt.f64, pt.fold);
        pt.inv[k] = make_tw(ipw, q, pt.f64, pt.fold);
        pw = mm(pw, pt.root, q);
        ipw = mm(ipw, iroot, q);
    }
    pt.inv_w0_scaled = make_tw(mm(pt.inv_u64[1], pt.ninv, q), q, pt.f64, pt.fold);
    return pt;
}

uint32_t Params::galois_elt_from_step(int step) const
{
    const uint32_t n = (uint32_t)N, m = 2 * n;
    if (step == 0) return m - 1; // column swap / conjugation
    const uint32_t mag = (uint32_t)(step < 0 ? -(long)step : step);
    if (mag >= n / 2) return 0;
    const uint32_t k = step < 0 ? n / 2 - mag : mag;
    u64 g = 1;
    for (uint32_t i = 0; i < k; ++i) g = (g * 3) & (m - 1);
    return (uint32_t)g;
}

std::vector<uint32_t> Params::galois_elts_all() const
{
    const u64 m = 2 * (u64)N;
    std::vector<uint32_t> out{(uint32_t)(m - 1)};
    u64 pos = 3, neg = 1;
    for (u64 x = 1; x < m; x += 2)
        if (((x * 3) & (m - 1)) == 1) { neg = x; break; }
    for (int i = 0; i < logn - 1; ++i) {
        out.push_back((uint32_t)pos);
        out.push_back((uint32_t)neg);
        pos = (pos * pos) & (m - 1);
        neg = (neg * neg) & (m - 1);
    }
    return out;
}

std::vector<uint32_t> Params::galois_perm_ntt(uint32_t elt) const
{
    std::vector<uint32_t> perm(N);
    for (size_t i = 0; i < N; ++i) {
        const u64 odd = 2 * (u64)bitrev((uint32_t)i, logn) + 1; // exponent of psi evaluated at slot i
        const u64 src = ((odd * elt) & (2 * N - 1)) >> 1;
        perm[i] = bitrev((uint32_t)src, logn);
        // A row of 1024 slots has ONE source row: the row index of slot i is the low logn1 bits of its exponent's upper part, and
        // odd * elt keeps low bits low.  The rotation kernels read the source row as a whole on the strength of this (k_k1, K1_GALOIS).
        if ((perm[i] >> 10) != (perm[i & ~(size_t)1023] >> 10)) throw std::logic_error("galois_perm_ntt: a row with two source rows");
    }
    return perm;
}

} // namespace he355
