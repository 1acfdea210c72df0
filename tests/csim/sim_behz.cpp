// sim_behz.cpp — TEST-ONLY.  Runs the product's per-coefficient BEHZ arithmetic (csrc/behz_core.h: the very functions the HIP kernels
// k_behz_extend(_cols) and k_behz_(cols_)floor_sk compile) on the CPU, on the constants the product uploads (Params::behz_host), so that
// tests/test_behz_sim_cpu.py can hold them to exact integer arithmetic -- at the ends of the Shenoy-Kumaresan bound too -- without a GPU.
// Compiled only into tests/csim/_build/libcsim.so; the product never contains it.
#include <cstring>
#include <map>
#include <stdexcept>
#include <vector>

#include "../../reference-seal-backend_amd/csrc/he_params.h"

using namespace he355;

namespace {
struct BehzSim {
    Params *p = nullptr;
    std::vector<PrimeDev> pd; // key chain, then m_sk, B_0, ... (the device's prime table order)
    std::map<int, BehzHost> hosts;
    const BehzHost &host(int L)
    {
        auto it = hosts.find(L);
        if (it == hosts.end()) it = hosts.emplace(L, p->behz_host(L)).first;
        return it->second;
    }
};
PrimeDev to_dev(const PrimeTables &pt)
{
    PrimeDev d;
    std::memset(&d, 0, sizeof(d));
    const ArU64 au = pt.aru();
    const ArF64 af = pt.arf();
    d.q = pt.q; d.cr0 = pt.mod.cr0; d.cr1 = pt.mod.cr1;
    d.ninv = au.ninv; d.ninv_q = au.ninv_q;
    d.qd = af.q; d.qinv = af.qinv; d.ninv_d = af.ninv; d.ninv_i = af.ninv_i;
    d.f64 = pt.f64 ? 1 : 0;
    return d;
}
} // namespace

extern "C" {

void *sim_behz_create(size_t N, const int *bits, size_t n_bits, int plain_bits)
{
    try {
        BehzSim *s = new BehzSim();
        s->p = Params::create(kSchemeBFV, N, std::vector<int>(bits, bits + n_bits), plain_bits, false);
        for (const PrimeTables &pt : s->p->primes) s->pd.push_back(to_dev(pt));
        for (const PrimeTables &pt : s->p->aux) s->pd.push_back(to_dev(pt));
        return s;
    } catch (const std::exception &) {
        return nullptr;
    }
}
void sim_behz_destroy(void *h)
{
    BehzSim *s = static_cast<BehzSim *>(h);
    if (s) { delete s->p; delete s; }
}
size_t sim_behz_levels(void *h) { return static_cast<BehzSim *>(h)->p->Ltop; }
uint64_t sim_behz_q(void *h, size_t i) { return static_cast<BehzSim *>(h)->p->primes[i].q; }
uint64_t sim_behz_t(void *h) { return static_cast<BehzSim *>(h)->p->plain_modulus; }
// m_sk, B_0, B_1, ...; returns their number at level L
size_t sim_behz_base(void *h, int L, uint64_t *out)
{
    BehzSim *s = static_cast<BehzSim *>(h);
    const size_t nB = s->p->behz_nB(L);
    for (size_t i = 0; i <= nB; ++i) out[i] = s->p->aux[i].q;
    return nB + 1;
}
int sim_behz_f64aux(void *h, int L) { return static_cast<BehzSim *>(h)->host(L).f64aux; }

// steps (1)-(2) of one coefficient: x[L] canonical residues under q -> out[S] residues under B_0 .. B_{nB-1}, m_sk
int sim_behz_extend(void *h, int L, const uint64_t *x, uint64_t *out)
{
    BehzSim *s = static_cast<BehzSim *>(h);
    try {
        const BehzHost &H = s->host(L);
        const BehzDev Z = H.view(H.words.data(), H.doubles.data(), s->p->K);
        const int S = Z.nB + 1;
        if (Z.L <= 4 && Z.nB <= 6) { // the product's small instantiation
            u64 xi[4] = {0, 0, 0, 0}, tmp[4], rmt;
            for (int i = 0; i < L; ++i) xi[i] = x[i];
            behz_ext_prepare<4>(Z, s->pd.data(), L, xi, tmp, rmt);
            for (int j = 0; j < S; ++j) out[j] = behz_ext_residue<4>(Z, behz_modu_at(s->pd.data(), Z.bsk_prime[j]), L, j, tmp, rmt);
        } else {
            u64 xi[kBehzMaxL] = {0}, tmp[kBehzMaxL], rmt;
            for (int i = 0; i < L; ++i) xi[i] = x[i];
            behz_ext_prepare<kBehzMaxL>(Z, s->pd.data(), L, xi, tmp, rmt);
            for (int j = 0; j < S; ++j) out[j] = behz_ext_residue<kBehzMaxL>(Z, behz_modu_at(s->pd.data(), Z.bsk_prime[j]), L, j, tmp, rmt);
        }
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}
// steps (6)-(8) of one coefficient: dq[L], ds[S] canonical residues of a product -> out[L]; variant 0: integer arithmetic, 1: fp64 engine
int sim_behz_floor(void *h, int L, const uint64_t *dq, const uint64_t *ds, uint64_t *out, int variant)
{
    BehzSim *s = static_cast<BehzSim *>(h);
    try {
        const BehzHost &H = s->host(L);
        const BehzDev Z = H.view(H.words.data(), H.doubles.data(), s->p->K);
        if (variant == 1 && !Z.f64aux) return 2;
        const int S = Z.nB + 1;
        if (Z.L <= 4 && Z.nB <= 6) {
            u64 vq[4] = {0, 0, 0, 0}, vs[7] = {0}, res[4];
            for (int i = 0; i < L; ++i) vq[i] = dq[i];
            for (int j = 0; j < S; ++j) vs[j] = ds[j];
            if (variant) behz_floor_sk_coeff_f64<4, 6>(Z, s->pd.data(), L, Z.nB, vq, vs, res);
            else behz_floor_sk_coeff<4, 6>(Z, s->pd.data(), L, Z.nB, vq, vs, res);
            for (int i = 0; i < L; ++i) out[i] = res[i];
        } else {
            u64 vq[kBehzMaxL] = {0}, vs[kBehzMaxB + 1] = {0}, res[kBehzMaxL];
            for (int i = 0; i < L; ++i) vq[i] = dq[i];
            for (int j = 0; j < S; ++j) vs[j] = ds[j];
            if (variant) behz_floor_sk_coeff_f64<kBehzMaxL, kBehzMaxB>(Z, s->pd.data(), L, Z.nB, vq, vs, res);
            else behz_floor_sk_coeff<kBehzMaxL, kBehzMaxB>(Z, s->pd.data(), L, Z.nB, vq, vs, res);
            for (int i = 0; i < L; ++i) out[i] = res[i];
        }
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}

} // extern "C"

// Operand selection of the BFV multiply (device_types.h: Indexer3, BehzSrc) on the CPU: for result r, the operand indices idx_a / idx_b,
// the operand ordinals ord_a / ord_b of the transformed-once lists, and the ciphertext index the extension reads for list item `item`
// (in units of ciphertexts: the functions are called with bases 0 and a ciphertext size of one word).
extern "C" void sim_behz_src_map(const uint64_t *ix8 /* a_base, b_base, gs, b1, a_sg, a_si, b_sg, b_sj */, uint64_t I, uint64_t J, uint64_t na, uint64_t r,
                                 uint64_t item, uint64_t *out /* idx_a, idx_b, ord_a, ord_b, list item's source (a: index, b: 2^62 + index) */)
{
    static const u64 base_a[1] = {0};
    BehzSrc s{};
    s.a = base_a;
    s.b = base_a + ((u64)1 << 59); // (never dereferenced: only the distance to `a` is read back)
    s.ix.a_base = ix8[0]; s.ix.b_base = ix8[1]; s.ix.gs = ix8[2]; s.ix.b1 = ix8[3];
    s.ix.a_sg = ix8[4]; s.ix.a_si = ix8[5]; s.ix.b_sg = ix8[6]; s.ix.b_sj = ix8[7];
    s.I = I; s.J = J; s.na = na; s.lists = 1;
    out[0] = idx_a(s.ix, r);
    out[1] = idx_b(s.ix, r);
    out[2] = ord_a(s, r);
    out[3] = ord_b(s, r);
    const u64 *p = behz_src_ct(s, item, 1);
    const u64 off = (u64)(p - s.a);
    out[4] = off >= ((u64)1 << 59) ? ((u64)1 << 62) + (off - ((u64)1 << 59)) : off;
}
