// sim_ntt.cpp — TEST-ONLY lane simulator.  Runs the product's lane programs (ntt_core.h / modarith.h,
// the very code the HIP kernels execute) on the CPU, one 64-lane wave at a time with a simulated LDS,
// so that the index maps, twiddle addressing and fp64 magnitude bounds can be checked against the oracle
// without a GPU.  It is compiled only into tests/csim/_build/libcsim.so; the product never contains it.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <type_traits>
#include <vector>

#define HE355_LANE_SIM 1 // the wide-lazy u64 butterflies report sums that leave 64 bits (modarith.h)
#include "../../reference-seal-backend_amd/csrc/he_params.h"
#include "../../reference-seal-backend_amd/csrc/ntt_core.h"

using namespace he355;
namespace he355 {
int he355_sim_overflow = 0;
}

static double g_maxmag = 0; // largest |value| seen by the fp64 engine at phase boundaries

template <class T> static void track(const T *, int) {}
template <> void track<double>(const double *x, int n)
{
    for (int i = 0; i < n; ++i) g_maxmag = std::max(g_maxmag, std::fabs(x[i]));
}

// exchange A -> B / B -> C (forward) and C -> B / B -> A (inverse) through the simulated LDS
template <class T> static void xchg(int from, int to, std::vector<T> &lds, T (*regs)[kRowE])
{
    for (int lane = 0; lane < 64; ++lane) {
        if (from == 0) lds_store_A(lds.data(), lane, regs[lane]);
        else if (from == 1) lds_store_B(lds.data(), lane, regs[lane]);
        else lds_store_C(lds.data(), lane, regs[lane]);
    }
    for (int lane = 0; lane < 64; ++lane) {
        if (to == 0) lds_load_A(lds.data(), lane, regs[lane]);
        else if (to == 1) lds_load_B(lds.data(), lane, regs[lane]);
        else lds_load_C(lds.data(), lane, regs[lane]);
    }
}

template <class Ar, int LOGN1> static void cols_fwd(const Ar &ar, const PrimeTables &pt, const u64 *in, u64 *raw)
{
    constexpr int N1 = 1 << LOGN1;
    for (int b = 0; b < kRowN; ++b) {
        typename Ar::T x[N1];
        for (int a = 0; a < N1; ++a) x[a] = ar.from_canon(in[a * kRowN + b]);
        col_fwd<Ar, LOGN1>(ar, x, pt.fwd.data());
        track(x, N1);
        for (int a = 0; a < N1; ++a) raw[a * kRowN + b] = ar.to_raw(x[a]);
    }
}
template <class Ar, int LOGN1> static void cols_inv(const Ar &ar, const PrimeTables &pt, const u64 *raw, u64 *out)
{
    constexpr int N1 = 1 << LOGN1;
    for (int b = 0; b < kRowN; ++b) {
        typename Ar::T x[N1];
        for (int a = 0; a < N1; ++a) x[a] = ar.from_raw(raw[a * kRowN + b]);
        col_inv<Ar, LOGN1>(ar, x, pt.inv.data(), pt.inv_w0_scaled);
        track(x, N1);
        for (int a = 0; a < N1; ++a) out[a * kRowN + b] = ar.to_canon(x[a]);
    }
}

template <class Ar> static void rows_fwd(const Ar &ar, const PrimeTables &pt, int n1, bool in_raw, const u64 *in, u64 *out)
{
    typedef typename Ar::T T;
    std::vector<T> lds(kLdsRow);
    static T regs[64][kRowE];
    for (int a = 0; a < n1; ++a) {
        const u32 rowbase = (u32)(n1 + a);
        const u64 *src = in + (size_t)a * kRowN;
        // the row-local twiddle copy (what K3 stages in LDS) must be equivalent to the table
        std::vector<Tw16> rowtw(kRowTw);
        for (u32 i = 0; i + 1 < (u32)kRowTw; ++i) rowtw[i] = pt.fwd[tw_row_source(rowbase, i)];
        TwRow twr; twr.t = rowtw.data();
        std::vector<double> roww(kRowTw);
        for (u32 i = 0; i + 1 < (u32)kRowTw; ++i) std::memcpy(&roww[i], &rowtw[i].a, 8);
        TwRowF64 twf; twf.t = roww.data(); twf.qinv = 1.0 / (double)pt.q;
        const bool f64row = std::is_same<T, double>::value && (a & 2);
        const auto twt = tw_table(pt.fwd.data(), rowbase);
        const bool use_row = (a & 1) != 0;
        for (int lane = 0; lane < 64; ++lane) {
            T *x = regs[lane];
            for (int r = 0; r < kRowE; ++r) x[r] = in_raw ? ar.from_raw(src[elemA(lane, r)]) : ar.from_canon(src[elemA(lane, r)]);
            if (f64row) row_fwd_A(ar, x, twf); else if (use_row) row_fwd_A(ar, x, twr); else row_fwd_A(ar, x, twt);
            track(x, kRowE);
        }
        // the two transposes run through LDS or through the cross-lane steps: all four combinations must agree
        static T saved[64][kRowE];
        std::memcpy(saved, regs, sizeof(saved));
        std::vector<u64> ref(kRowN);
        // the wide lazy range of the u64 engine (k_k3's digit rows; key primes are below 2^60): same residues, no sum leaves 64 bits
        std::vector<u64> lazy_ref;
        if constexpr (std::is_same<T, u64>::value) {
            if (!(pt.q >> 60)) {
                static u64 lz[64][1][kRowE];
                std::vector<u64> ldl(kLdsRow);
                he355_sim_overflow = 0;
                for (int lane = 0; lane < 64; ++lane) {
                    for (int r = 0; r < kRowE; ++r) lz[lane][0][r] = in_raw ? ar.from_raw(src[elemA(lane, r)]) : ar.from_canon(src[elemA(lane, r)]);
                    Tw16 w[kTwA]; gather_A(twt, w); row_fwd_A_lazy<1>(ar, lz[lane], w);
                }
                for (int lane = 0; lane < 64; ++lane) lds_store_A(ldl.data(), lane, lz[lane][0]);
                for (int lane = 0; lane < 64; ++lane) lds_load_B(ldl.data(), lane, lz[lane][0]);
                for (int lane = 0; lane < 64; ++lane) { Tw16 w[kTwB]; gather_B(twt, lane, w); row_fwd_B_lazy<1>(ar, lz[lane], w); }
                for (int lane = 0; lane < 64; ++lane) lds_store_B(ldl.data(), lane, lz[lane][0]);
                for (int lane = 0; lane < 64; ++lane) lds_load_C(ldl.data(), lane, lz[lane][0]);
                lazy_ref.resize(kRowN);
                for (int lane = 0; lane < 64; ++lane) {
                    Tw16 w[kTwC]; gather_C(twt, lane, w); row_fwd_C_lazy<1>(ar, lz[lane], w);
                    for (int r = 0; r < kRowE; ++r) {
                        // Shoup build: below 12 q; fold build: below 8 q + 16 c (modarith.h, bfly_fwd_lazy)
                        const u64 bound = ArU64::kFold ? 8 * pt.q + 16 * (u64)fold_c(pt.q) : 12 * pt.q;
                        if (lz[lane][0][r] >= bound) throw std::runtime_error("wide-lazy row pass: a result is not below its bound");
                        lazy_ref[elemC(lane, r)] = ArU64::kFold ? ar.to_canon(ar.lazy_reduce(lz[lane][0][r])) : ar.to_canon16(lz[lane][0][r]);
                    }
                }
                if (he355_sim_overflow) throw std::runtime_error("wide-lazy row pass: a sum left 64 bits");
                // k_k3's lazy accumulation runs: kAccRun products of the largest operands on an accumulator just below 4q fit 64 bits, with
                // the exact key quotient (shoup_quotient) and for any 64-bit x
                const u64 xs[4] = {~(u64)0, 12 * pt.q - 1, pt.q - 1, 0x9e3779b97f4a7c15ull};
                const u64 ks[3] = {pt.q - 1, pt.q / 2 + 1, 1};
                for (u64 xv : xs)
                    for (u64 kv : ks) {
                        // the key's companion word: its exact Shoup quotient, or (fold build) the key times 2^32
                        const u64 kq = ArU64::kFold ? pre_word(kv, pt.q, true) : ar.shoup_quotient(kv);
                        if (!ArU64::kFold && ((u128)kq * pt.q > ((u128)kv << 64) || (u128)(kq + 1) * pt.q <= ((u128)kv << 64))) throw std::runtime_error("shoup_quotient is not floor(w * 2^64 / q)");
                        // the largest accumulator a run may start from: just below 4q, or (fold build) what acc_reduce leaves, below 2^61 + 14c
                        const u64 start = ArU64::kFold ? ((u64)1 << 61) + 14 * (u64)fold_c(pt.q) - 1 : 4 * pt.q - 1;
                        u64 acc = start, want = (u64)(((u128)start) % pt.q);
                        for (int k = 0; k < ArU64::kAccRun; ++k) {
                            ar.acc_mac_lazy(acc, xv, kv, kq);
                            want = (u64)((want + (u128)(xv % pt.q) * kv) % pt.q);
                        }
                        if (he355_sim_overflow) throw std::runtime_error("lazy accumulation run: a sum left 64 bits");
                        acc = ar.acc_reduce(acc);
                        if (acc > start || ar.acc_canon(acc) != want) throw std::runtime_error("lazy accumulation run: wrong residue");
                    }
            }
        }
        xchg(0, 1, lds, regs);
        for (int lane = 0; lane < 64; ++lane) { if (f64row) row_fwd_B(ar, regs[lane], twf, lane); else if (use_row) row_fwd_B(ar, regs[lane], twr, lane); else row_fwd_B(ar, regs[lane], twt, lane); track(regs[lane], kRowE); }
        xchg(1, 2, lds, regs);
        for (int lane = 0; lane < 64; ++lane) {
            T *x = regs[lane];
            if (f64row) row_fwd_C(ar, x, twf, lane); else if (use_row) row_fwd_C(ar, x, twr, lane); else row_fwd_C(ar, x, twt, lane);
            track(x, kRowE);
            for (int r = 0; r < kRowE; ++r) {
                const u64 v = ar.to_canon(x[r]);
                if (!lazy_ref.empty() && lazy_ref[elemC(lane, r)] != v) throw std::runtime_error("wide-lazy row pass differs from the Harvey row pass");
                out[(size_t)a * kRowN + elemC(lane, r)] = v;
            }
        }
    }
}
template <class Ar> static void rows_inv(const Ar &ar, const PrimeTables &pt, int n1, const u64 *in, u64 *out)
{
    typedef typename Ar::T T;
    std::vector<T> lds(kLdsRow);
    static T regs[64][kRowE];
    for (int a = 0; a < n1; ++a) {
        const u32 rowbase = (u32)(n1 + a);
        const auto itw = tw_table(pt.inv.data(), rowbase);
        for (int lane = 0; lane < 64; ++lane) {
            T *x = regs[lane];
            for (int r = 0; r < kRowE; ++r) x[r] = ar.from_canon(in[(size_t)a * kRowN + elemC(lane, r)]);
            // phase C both ways: twiddles read where they are used (row_inv_C), and gathered up front (gather_inv_C + row_inv_C_w: what the
            // ring-in-LDS kernels of he355_kernels_lds.hip run) -- the same registers, bit for bit
            T y[kRowE];
            for (int r = 0; r < kRowE; ++r) y[r] = x[r];
            Tw16 wc[kTwInvC];
            gather_inv_C(itw, lane, wc);
            row_inv_C_w(ar, y, wc);
            row_inv_C(ar, x, itw, lane);
            if (std::memcmp(x, y, sizeof(T) * kRowE) != 0) throw std::runtime_error("row_inv_C_w differs from row_inv_C");
            track(x, kRowE);
        }
        xchg(2, 1, lds, regs);
        for (int lane = 0; lane < 64; ++lane) { // (as the kernels run it: the phase's twiddles gathered up front)
            Tw16 wb[kTwInvB];
            gather_inv_B(itw, lane, wb);
            row_inv_B_w(ar, regs[lane], wb);
            track(regs[lane], kRowE);
        }
        xchg(1, 0, lds, regs);
        for (int lane = 0; lane < 64; ++lane) {
            T *x = regs[lane];
            Tw16 wa[kTwInvA];
            gather_inv_A(itw, wa);
            if (n1 == 1) row_inv_A_w<Ar, true>(ar, x, wa, pt.inv_w0_scaled);
            else row_inv_A_w<Ar, false>(ar, x, wa, pt.inv_w0_scaled);
            track(x, kRowE);
            for (int r = 0; r < kRowE; ++r) out[(size_t)a * kRowN + elemA(lane, r)] = (n1 == 1) ? ar.to_canon(x[r]) : ar.to_raw(x[r]);
        }
    }
}

template <class Ar> static void fwd_any(const Ar &ar, const Params &P, const PrimeTables &pt, u64 *poly)
{
    const int n1 = 1 << P.logn1;
    std::vector<u64> raw(P.N), out(P.N);
    switch (P.logn1) {
    case 0: break;
    case 1: cols_fwd<Ar, 1>(ar, pt, poly, raw.data()); break;
    case 2: cols_fwd<Ar, 2>(ar, pt, poly, raw.data()); break;
    case 3: cols_fwd<Ar, 3>(ar, pt, poly, raw.data()); break;
    case 4: cols_fwd<Ar, 4>(ar, pt, poly, raw.data()); break;
    case 5: cols_fwd<Ar, 5>(ar, pt, poly, raw.data()); break;
    }
    rows_fwd(ar, pt, n1, P.logn1 > 0, P.logn1 > 0 ? raw.data() : poly, out.data());
    std::memcpy(poly, out.data(), P.N * 8);
}
template <class Ar> static void inv_any(const Ar &ar, const Params &P, const PrimeTables &pt, u64 *poly)
{
    const int n1 = 1 << P.logn1;
    std::vector<u64> raw(P.N), out(P.N);
    rows_inv(ar, pt, n1, poly, raw.data());
    switch (P.logn1) {
    case 0: std::memcpy(out.data(), raw.data(), P.N * 8); break;
    case 1: cols_inv<Ar, 1>(ar, pt, raw.data(), out.data()); break;
    case 2: cols_inv<Ar, 2>(ar, pt, raw.data(), out.data()); break;
    case 3: cols_inv<Ar, 3>(ar, pt, raw.data(), out.data()); break;
    case 4: cols_inv<Ar, 4>(ar, pt, raw.data(), out.data()); break;
    case 5: cols_inv<Ar, 5>(ar, pt, raw.data(), out.data()); break;
    }
    std::memcpy(poly, out.data(), P.N * 8);
}

// the forward column pass of prime i on a column whose every element is `value` (any lazy input below 4q): no sum leaves 64 bits, every
// output is below 4q, and the residues equal those of the same column entered canonically.  0: ok.
template <int LOGN1> static int col_extreme(const ArU64 &ar, const PrimeTables &pt, u64 value)
{
    constexpr int N1 = 1 << LOGN1;
    u64 x[N1], y[N1];
    for (int a = 0; a < N1; ++a) { x[a] = value; y[a] = value % pt.q; }
    he355_sim_overflow = 0;
    col_fwd<ArU64, LOGN1>(ar, x, pt.fwd.data());
    col_fwd<ArU64, LOGN1>(ar, y, pt.fwd.data());
    if (he355_sim_overflow) return 1;
    for (int a = 0; a < N1; ++a)
        if (x[a] >= 4 * pt.q || ar.to_canon(x[a]) != ar.to_canon(y[a])) return 1;
    return 0;
}
extern "C" {
void *sim_params_create(int scheme, size_t N, const int *bits, size_t n, int plain_bits, int sec128)
{
    try {
        // this library runs ONE form of the u64 engine's arithmetic: the tables must be made for it.  Shoup build: never fold tables;
        // fold build: a parameter set whose u64-engine primes are not all 2^60 - c has no fold form (null: the caller skips it)
        Params *p = Params::create(scheme, N, std::vector<int>(bits, bits + n), plain_bits, sec128 != 0, ArU64::kFold);
        bool has_u64 = false;
        for (const PrimeTables &pt : p->primes) has_u64 |= !pt.f64;
        if (has_u64 && p->u64_fold != ArU64::kFold) {
            delete p;
            return nullptr;
        }
        return p;
    } catch (std::exception &) {
        return nullptr;
    }
}
int sim_u64_fold_build(void) { return ArU64::kFold ? 1 : 0; }
// explicit primes (the ends of the fold form's range of c, long chains of 60-bit primes)
void *sim_params_create_primes(int scheme, size_t N, const uint64_t *primes, size_t n, uint64_t plain_modulus)
{
    try {
        Params *p = Params::create_primes(scheme, N, std::vector<u64>(primes, primes + n), plain_modulus, ArU64::kFold);
        bool has_u64 = false;
        for (const PrimeTables &pt : p->primes) has_u64 |= !pt.f64;
        if (has_u64 && p->u64_fold != ArU64::kFold) {
            delete p;
            return nullptr;
        }
        return p;
    } catch (std::exception &) {
        return nullptr;
    }
}
// The row pass of prime i on rows whose every element is `value`, taken as a RAW lazy input (what the column pass may hand over: anything
// below 4q).  0: the wide-lazy pass stayed inside 64 bits and its bounds and agreed with the Harvey pass; 1: it did not.
int sim_row_pass_extreme(void *p, size_t i, uint64_t value)
{
    const Params &P = *(Params *)p;
    const PrimeTables &pt = P.primes[i];
    if (pt.f64) return 0;
    const int n1 = 1 << P.logn1;
    std::vector<u64> in((size_t)n1 * kRowN, value), out((size_t)n1 * kRowN);
    try {
        rows_fwd(pt.aru(), pt, n1, true, in.data(), out.data());
    } catch (std::exception &e) {
        std::fprintf(stderr, "sim_row_pass_extreme: %s\n", e.what());
        return 1;
    }
    return 0;
}
void sim_params_destroy(void *p) { delete (Params *)p; }
uint64_t sim_modulus(void *p, size_t i) { return ((Params *)p)->primes[i].q; }
uint64_t sim_root(void *p, size_t i) { return ((Params *)p)->primes[i].root; }
int sim_is_f64(void *p, size_t i) { return ((Params *)p)->primes[i].f64; }
size_t sim_K(void *p) { return ((Params *)p)->K; }
double sim_maxmag_reset(void) { double m = g_maxmag; g_maxmag = 0; return m; }
// the bound / overflow flag of the u64 engine's lazy forms (modarith.h, HE355_LANE_SIM), read and cleared
int sim_overflow_reset(void) { const int f = he355_sim_overflow; he355_sim_overflow = 0; return f; }
void sim_ntt_forward(void *p, size_t i, uint64_t *poly)
{
    const Params &P = *(Params *)p;
    const PrimeTables &pt = P.primes[i];
    if (pt.f64) fwd_any(pt.arf(), P, pt, poly);
    else fwd_any(pt.aru(), P, pt, poly);
}
void sim_ntt_inverse(void *p, size_t i, uint64_t *poly)
{
    const Params &P = *(Params *)p;
    const PrimeTables &pt = P.primes[i];
    if (pt.f64) inv_any(pt.arf(), P, pt, poly);
    else inv_any(pt.aru(), P, pt, poly);
}
int sim_col_pass_extreme(void *p, size_t i, uint64_t value)
{
    const Params &P = *(Params *)p;
    const PrimeTables &pt = P.primes[i];
    if (pt.f64) return 0;
    const ArU64 ar = pt.aru();
    switch (P.logn1) {
    case 1: return col_extreme<1>(ar, pt, value);
    case 2: return col_extreme<2>(ar, pt, value);
    case 3: return col_extreme<3>(ar, pt, value);
    case 4: return col_extreme<4>(ar, pt, value);
    case 5: return col_extreme<5>(ar, pt, value);
    default: return 0;
    }
}
uint32_t sim_galois_elt(void *p, int step) { return ((Params *)p)->galois_elt_from_step(step); }
size_t sim_galois_elts_all(void *p, uint32_t *out)
{
    auto v = ((Params *)p)->galois_elts_all();
    std::memcpy(out, v.data(), v.size() * 4);
    return v.size();
}
void sim_galois_perm(void *p, uint32_t elt, uint32_t *out)
{
    auto v = ((Params *)p)->galois_perm_ntt(elt);
    std::memcpy(out, v.data(), v.size() * 4);
}
}
