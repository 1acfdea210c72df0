// he355_api.hip — device context, per-op kernel sequences and the C ABI declared in include/he355.h.
// Host code is C++17; the only way into the GPU is through the launchers of he355_kernels.h.
// There is no CPU implementation of any evaluator op in this library: without a HIP device every device
// entry point fails with HE355_E_DEVICE.
#include <hip/hip_runtime.h>

#include <array>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/he355.h"
#include "device_pool.h"
#include "he355_internal.h"
#include "he355_kernels.h"
#include "he_params.h"
#include "ntt_core.h"
#include "client/he_client.h"
#include "client/multiword.h"

namespace he355 {

#define HIPCHECK(expr)                                                                                             \
    do {                                                                                                           \
        hipError_t e__ = (expr);                                                                                   \
        if (e__ != hipSuccess)                                                                                     \
            throw DeviceError(std::string("HIP error: ") + hipGetErrorString(e__) + " in " #expr " (" __FILE__ ":" + \
                              std::to_string(__LINE__) + ")");                                                     \
    } while (0)

namespace {

__device__ __forceinline__ u64 splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct PrimeMap {
    unsigned char prime_of[64];
    u32 period;
};

// uniform-looking residues in [0, q): counter-based generator, value = floor(rand64 * q / 2^64)
// first_poly: index of dst's first polynomial in the whole (virtual) array the stream belongs to -- a shard of a batch filled with
// its offset holds exactly the values the full batch would hold there, whatever the number of shards
__global__ void k_fill_uniform(u64 *dst, u64 n_polys, int logN, const PrimeDev *primes, PrimeMap pm, u64 seed, u64 first_poly)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 poly = gid >> logN;
    if (poly >= n_polys) return;
    const u64 q = primes[pm.prime_of[(first_poly + poly) % pm.period]].q;
    dst[gid] = mulhi64(splitmix64(seed ^ splitmix64(gid + (first_poly << logN))), q);
}
// key residues of fp64-engine primes are kept as doubles in HBM (exact: q < 2^47)
__global__ void k_key_to_engine(u64 *key, u64 n_polys, int logN, const PrimeDev *primes, int K)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 poly = gid >> logN;
    if (poly >= n_polys) return;
    if (primes[poly % K].f64) {
        union { u64 u; double d; } c;
        c.d = u52_to_f64(key[gid]);
        key[gid] = c.u;
    }
}

// A finished key (k_key_to_engine's format) with the residues under the data primes multiplied by P^-1 (P = the special prime; the
// special prime's own residues are copied): what k_k3<TENSOR> multiplies the lifted digits by, so that its sums are sums * P^-1.
__global__ void k_key_scaled_copy(const u64 *key, u64 *out, u64 n_polys, int logN, const PrimeDev *primes, const FloorConst *fcs, int K)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 poly = gid >> logN;
    if (poly >= n_polys) return;
    const int t = (int)(poly % K);
    u64 v = key[gid];
    if (t != K - 1) {
        const PrimeDev &P = primes[t];
        const FloorConst fc = fcs[(K - 1) * K + t];
        if (P.f64) {
            ArF64 ar;
            ar.q = (double)P.q; ar.qinv = 1.0 / (double)P.q;
            union { u64 u; double d; } c;
            c.u = v;
            c.d = ar.canon2(ar.mulmod_c(c.d, fc.inv_d, fc.inv_i));
            v = c.u;
        } else {
            ModU64 m; // (Barrett: this file is compiled once, whatever companion words the context's tables hold)
            m.q = P.q; m.cr0 = P.cr0; m.cr1 = P.cr1;
            v = mulmod(v, fc.inv, m);
        }
    }
    out[gid] = v;
}

// Companion words of the key residues under the u64-engine primes, appended to the key: [L_top][2][n_q][N] -- their Shoup quotients, or
// (fold: the context's u64-engine primes are 2^60 - c, modarith.h) the residues times 2^32
__global__ void k_key_quotients(const u64 *key, u64 *keyq, u64 n_dk, int logN, const PrimeDev *primes, int K, int n_q, PrimeMap qmap, int fold)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 poly = gid >> logN; // (digit*2 + k) * n_q + slot
    if (poly >= n_dk * n_q) return;
    const int slot = (int)(poly % n_q), t = qmap.prime_of[slot];
    const u64 dk = poly / n_q;
    ArU64 ar;
    ar.q = primes[t].q; ar.two_q = 2 * ar.q; ar.cr0 = primes[t].cr0; ar.cr1 = primes[t].cr1; ar.ninv = ar.ninv_q = 0;
    const u64 kv = key[((dk * K + t) << logN) + (gid & (((u64)1 << logN) - 1))];
    keyq[gid] = fold ? barrett128((u128)kv << 32, ar.mod()) : ar.shoup_quotient(kv); // (exact quotient: k_k3's lazy runs rely on it)
}

} // namespace

// 64 bits from the operating system's entropy source (the context's own encryptions of zero; the bridge's client seeds likewise)
static u64 os_seed()
{
    std::random_device rd;
    return ((u64)rd() << 32) ^ (u64)rd();
}

// In-run shader clock (bench.py's roofline.valu.sustained_mhz): ONE wave on a stream of its own reads the shader-cycle counter
// (s_memtime) and the constant 100 MHz counter (s_memrealtime), sleeps until `ticks_100mhz` of real time have passed -- a bound every
// path reaches, nothing else ends the loop -- and reads both again: clock = d(s_memtime) / d(s_memrealtime) x 100 MHz while the
// evaluator kernels of the probed region run beside it (/opt/skills/guides/MI355X_MICROARCH.md, "DVFS give-back" (6)).
__global__ void k_clock_probe(u64 ticks_100mhz, u64 *out)
{
    const u64 r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    u64 r = r0;
    while (r - r0 < ticks_100mhz) {
        __builtin_amdgcn_s_sleep(127);
        r = __builtin_amdgcn_s_memrealtime();
    }
    const u64 c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r - r0; }
}

class DeviceContext {
public:
    DeviceContext(const Params &p, int device) : P(p), device_(device)
    {
        // every per-prime table of the device side is sized kMaxPrimes (KernelEnv::prime_f64, PrimeMap, kernel argument lists)
        if (p.K + p.aux.size() + 1 > (size_t)kMaxPrimes) throw std::invalid_argument("too many primes for one device context (key chain + auxiliary primes exceed 64)");
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0) throw DeviceError("no HIP device available (the MI355X backend has no CPU fallback)");
        if (device < 0 || device >= count) throw DeviceError("invalid device ordinal");
        HIPCHECK(hipSetDevice(device));
        HIPCHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        HIPCHECK(hipStreamCreateWithFlags(&stream2_, hipStreamNonBlocking));
        HIPCHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
        HIPCHECK(hipEventCreate(&ev0_));
        HIPCHECK(hipEventCreate(&ev1_));
        const size_t N = P.N, K = P.K;
        // BFV: the BEHZ auxiliary primes follow the key chain, then the plain modulus t (BatchEncoder's NTT mod t)
        const bool with_t = P.scheme == kSchemeBFV && P.plain_modulus > 2 && (P.plain_modulus - 1) % (2 * N) == 0;
        if (with_t) plain_tables_ = Params::make_prime_tables(P.plain_modulus, N, P.logn, true); // t < 2^32: the fp64 engine's
        t_index_ = with_t ? (int)(K + P.aux.size()) : -1;
        const size_t n_all = K + P.aux.size() + (with_t ? 1 : 0);
        std::vector<PrimeDev> pd(n_all);
        for (size_t i = 0; i < n_all; ++i) {
            const PrimeTables &pt = i < K ? P.primes[i] : i < K + P.aux.size() ? P.aux[i - K] : plain_tables_;
            Tw16 *dfwd = nullptr, *dinv = nullptr;
            dmalloc(dfwd, N * sizeof(Tw16));
            owned_.push_back(dfwd);
            dmalloc(dinv, N * sizeof(Tw16));
            owned_.push_back(dinv);
            HIPCHECK(hipMemcpy(dfwd, pt.fwd.data(), N * sizeof(Tw16), hipMemcpyHostToDevice));
            HIPCHECK(hipMemcpy(dinv, pt.inv.data(), N * sizeof(Tw16), hipMemcpyHostToDevice));
            PrimeDev &d = pd[i];
            const ArU64 au = pt.aru();
            const ArF64 af = pt.arf();
            d.q = pt.q; d.cr0 = pt.mod.cr0; d.cr1 = pt.mod.cr1;
            d.ninv = au.ninv; d.ninv_q = au.ninv_q;
            d.qd = af.q; d.qinv = af.qinv; d.ninv_d = af.ninv; d.ninv_i = af.ninv_i;
            d.fwd = dfwd; d.inv = dinv; d.inv_w0_scaled = pt.inv_w0_scaled;
            d.f64 = pt.f64 ? 1 : 0; d.pad_ = 0;
            for (size_t k = 0; k < 64; ++k) d.colw[k] = 0.0;
            if (pt.f64)
                for (size_t k = 0; k < 32 && k < N; ++k) { d.colw[k] = ArF64::tw_w(pt.fwd[k]); d.colw[32 + k] = ArF64::tw_w(pt.inv[k]); }
            d.k2_direct = d.k2_lift = 0;
            d.pow32 = (double)((((u64)1) << 32) % pt.q);
            env_.prime_f64[i] = pt.f64 ? 1 : 0;
            env_.prime_q[i] = pt.q;
        }
        // k_k2n's lift classes per (digit prime j, fp64-engine target prime t) -- the same rule the kernel's general path evaluates
        for (size_t j = 0; j < K; ++j)
            for (size_t t = 0; t < K; ++t) {
                if (!P.primes[t].f64) continue;
                const u64 qj = P.primes[j].q, qt = P.primes[t].q;
                const bool df = (qj >> 52) == 0, direct = df && !(qj > 2 * qt);
                double m = df ? (direct ? (double)qj : 0.5 * (double)qt + 1.0) : (double)qt;
                for (int st = 0; st < P.logn1; ++st) m += (double)qt * (0.5 + m * 4.440892098500626e-16);
                if (!(m * 1.0000001 < 140737488355328.0)) continue;
                (direct ? pd[j].k2_direct : pd[j].k2_lift) |= (u64)1 << t;
            }
        dmalloc(d_primes_, n_all * sizeof(PrimeDev));
        HIPCHECK(hipMemcpy(d_primes_, pd.data(), n_all * sizeof(PrimeDev), hipMemcpyHostToDevice));
        std::vector<FloorConst> fc(K * K);
        for (size_t s = 0; s < K; ++s)
            for (size_t i = 0; i < K; ++i) {
                FloorConst &f = fc[s * K + i];
                std::memset(&f, 0, sizeof(f));
                if (s == i) continue;
                const u64 qi = P.primes[i].q, qs = P.primes[s].q;
                const u64 inv = Params::invmod(qs % qi, qi);
                f.inv = inv;
                f.inv_shoup = pre_word(inv, qi, P.primes[i].fold); // the u64 engine's companion word (Shoup quotient, or inv 2^32 mod q_i)
                f.inv_d = (double)inv;
                f.inv_i = (double)inv / (double)qi;
                f.half_mod = (qs >> 1) % qi;
                f.src_mod = qs % qi;
            }
        dmalloc(d_floor_, K * K * sizeof(FloorConst));
        HIPCHECK(hipMemcpy(d_floor_, fc.data(), K * K * sizeof(FloorConst), hipMemcpyHostToDevice));
        env_.primes = d_primes_;
        env_.floor_consts = d_floor_;
        env_.N = (int)N; env_.logn1 = P.logn1; env_.K = (int)K; env_.Ltop = (int)P.Ltop; env_.scheme = P.scheme;
        env_.stream = stream_;
        env_.u64_fold = P.u64_fold;
        const char *ds = std::getenv("HE355_DUAL_STREAM");
        if (ds) dual_stream_ = ds[0] != '0';
        const char *lm = std::getenv("HE355_LATENCY_MAX");
        if (lm) set_latency_max((u64)std::max(0, std::atoi(lm)));
        const char *ch = std::getenv("HE355_CHUNK");
        if (ch && std::atoi(ch) > 0) chunk_ = (size_t)std::atoi(ch);
    }
    ~DeviceContext()
    {
        (void)hipSetDevice(device_);
        (void)hipStreamSynchronize(stream_);
        for (void *p : owned_) pool_.raw_free(p);
        pool_.raw_free(d_primes_);
        pool_.raw_free(d_floor_);
        pool_.raw_free(d_relin_);
        pool_.raw_free(d_relin_scaled_);
        pool_.raw_free(d_clock_);
        for (auto &kv : d_galois_) pool_.raw_free(kv.second);
        for (auto &kv : d_perm_) pool_.raw_free(kv.second);
        (void)hipStreamSynchronize(stream2_);
        pool_.raw_free(scratch_);
        pool_.raw_free(scratch2_);
        pool_.raw_free(rot_tmp_);
        pool_.raw_free(d_groups_);
        pool_.raw_free(lat_part_[0]);
        pool_.raw_free(lat_part_[1]);
        pool_.raw_free(bfv_scratch_);
        pool_.raw_free(client_scratch_);
        for (auto &kv : d_gather_) pool_.raw_free(kv.second);
        pool_.destroy();
        (void)hipEventDestroy(ev_fork_);
        (void)hipEventDestroy(ev_join_);
        (void)hipStreamDestroy(stream2_);
        if (probe_stream_) { (void)hipStreamSynchronize(probe_stream_); (void)hipStreamDestroy(probe_stream_); }
        (void)hipEventDestroy(ev0_);
        (void)hipEventDestroy(ev1_);
        (void)hipStreamDestroy(stream_);
    }

    void use() { HIPCHECK(hipSetDevice(device_)); }
    // ---- device memory (device_pool.h): slabs handed to callers come from the pool, the context's own tables / keys / arenas
    // are raw allocations; both are counted
    template <class T> void dmalloc(T *&p, size_t bytes) { p = static_cast<T *>(pool_.raw_malloc(bytes)); }
    // HE355_POOL=0: every he355_malloc / he355_free is a hipMalloc / drain + hipFree again (the pre-pool behaviour, kept for A/B timing:
    // tools/bench_bridge.py, profiles/r04_bridge_phases.jsonl)
    void *pool_alloc(size_t bytes)
    {
        use();
        return pool_on_ ? pool_.alloc(bytes) : pool_.raw_malloc(bytes ? bytes : 8);
    }
    void pool_free(void *p)
    {
        if (!pool_on_) { // HE355_POOL=0: every block is a raw allocation, freed as before the pool existed
            sync();
            pool_.raw_free(p);
            return;
        }
        switch (pool_.release(p)) {
        case DevicePool::kReleased: return;
        case DevicePool::kCached: throw std::invalid_argument("he355_free: this block was freed already");
        default: throw std::invalid_argument("he355_free: not a block this context allocated (he355_malloc of another context?)");
        }
    }
    DevicePool::Stats alloc_stats() const { return pool_.stats(); }
    // which shape / schedule the key switches of this context took (he355_path_stats): counted per kernel sequence (one chunk of a batch)
    he355_path_stats_t path_stats(bool reset)
    {
        const he355_path_stats_t s = paths_;
        if (reset) paths_ = he355_path_stats_t{};
        return s;
    }
    size_t pool_trim() { sync(); return pool_.trim(); }
    hipStream_t stream() const { return stream_; }
    int device() const { return device_; }
    const KernelEnv &env() const { return env_; }
    void set_chunk(size_t c) { chunk_ = c ? c : 1; }
    void set_latency_max(u64 n) { lat_auto_ = n == ~(u64)0; lat_max_ = lat_auto_ ? 0 : n; } // ~0: back to lat_limit()'s rule
    void set_level_walk(bool on) { level_walk_ = on; }
    void set_lds_max(u64 n) { lds_auto_ = n == ~(u64)0; lds_max_ = lds_auto_ ? 0 : n; }

    size_t key_elems() const { return P.Ltop * 2 * P.K * P.N; }
    size_t n_q_primes() const
    {
        size_t n = 0;
        for (size_t i = 0; i < P.K; ++i) n += env_.prime_f64[i] == 0;
        return n;
    }
    size_t key_alloc_elems() const { return key_elems() + P.Ltop * 2 * n_q_primes() * P.N; } // key + Shoup quotients (u64-engine primes)

    void key_from_host(u64 **slot, const u64 *h_key)
    {
        use();
        if (!*slot) dmalloc(*slot, key_alloc_elems() * 8);
        HIPCHECK(hipMemcpyAsync(*slot, h_key, key_elems() * 8, hipMemcpyHostToDevice, stream_));
        key_finish(*slot);
    }
    void key_synthetic(u64 **slot, u64 seed)
    {
        use();
        if (!*slot) dmalloc(*slot, key_alloc_elems() * 8);
        PrimeMap pm;
        pm.period = (u32)P.K;
        for (size_t i = 0; i < P.K; ++i) pm.prime_of[i] = (unsigned char)i;
        const u64 n_polys = P.Ltop * 2 * P.K, total = n_polys * P.N;
        hipLaunchKernelGGL(k_fill_uniform, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream_, *slot, n_polys, P.logn, d_primes_, pm, seed, (u64)0);
        key_finish(*slot);
    }
    // KeyGenerator on the device: the key for key_id 1 (relinearization) or 2 + galois_elt, from the secret key set with
    // he355_set_secret_key; same bits as the host client's make_relin_key / make_galois_key for the same seed
    void key_generate(u64 **slot, u64 seed, uint32_t galois_elt /* 0: relinearization key */)
    {
        use();
        require_keyswitch();
        if (!d_sk_) throw std::invalid_argument("secret key not set");
        if (!*slot) dmalloc(*slot, key_alloc_elems() * 8);
        const uint32_t *perm_tab = galois_elt ? perm(galois_elt) : nullptr;
        u64 *scr = client_scratch((P.Ltop * P.K + P.K) * P.N);
        launch_keygen_kswitch(env_, *slot, scr, scr + P.Ltop * P.K * P.N, d_sk_, perm_tab, seed, galois_elt ? 2 + (u64)galois_elt : 1);
        key_finish(*slot);
    }
    void key_finish(u64 *d_key)
    {
        const u64 n_polys = P.Ltop * 2 * P.K, total = n_polys * P.N;
        if (d_key == d_relin_) d_relin_scaled_ok_ = false; // the scaled copy follows the key
        key_quotients(d_key);
        hipLaunchKernelGGL(k_key_to_engine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream_, d_key, n_polys, P.logn, d_primes_, (int)P.K);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(stream_));
    }
    void key_quotients(u64 *d_key) // Shoup quotients of the residues under the u64-engine primes (integers in either key format)
    {
        const int n_q = (int)n_q_primes();
        if (n_q) { // before the fp64 conversion below rewrites the other residues; these stay integers
            PrimeMap qm;
            qm.period = (u32)n_q;
            int k = 0;
            for (size_t i = 0; i < P.K; ++i)
                if (!env_.prime_f64[i]) qm.prime_of[k++] = (unsigned char)i;
            const u64 n_dk = P.Ltop * 2, tq = n_dk * n_q * P.N;
            hipLaunchKernelGGL(k_key_quotients, dim3((unsigned)((tq + 255) / 256)), dim3(256), 0, stream_, d_key, d_key + key_elems(), n_dk, P.logn, d_primes_,
                               (int)P.K, n_q, qm, P.u64_fold ? 1 : 0);
        }
    }
    u64 **relin_slot() { return &d_relin_; }
    u64 **galois_slot(uint32_t elt) { return &d_galois_[elt]; }
    const u64 *relin_key() const { return d_relin_; }
    // the relinearization key with its data-prime residues times P^-1 (k_key_scaled_copy), built on first use after every change of the key
    const u64 *relin_scaled()
    {
        if (!d_relin_scaled_ok_) {
            if (!d_relin_scaled_) dmalloc(d_relin_scaled_, key_alloc_elems() * 8);
            const u64 n_polys = P.Ltop * 2 * P.K, total = n_polys * P.N;
            hipLaunchKernelGGL(k_key_scaled_copy, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream_, d_relin_, d_relin_scaled_, n_polys, P.logn, d_primes_,
                               d_floor_, (int)P.K);
            key_quotients(d_relin_scaled_);
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipStreamSynchronize(stream_));
            d_relin_scaled_ok_ = true;
        }
        return d_relin_scaled_;
    }
    const u64 *galois_key(uint32_t elt) const
    {
        auto it = d_galois_.find(elt);
        return it == d_galois_.end() ? nullptr : it->second;
    }
    const uint32_t *perm(uint32_t elt)
    {
        auto it = d_perm_.find(elt);
        if (it != d_perm_.end()) return it->second;
        const std::vector<uint32_t> h = P.galois_perm_ntt(elt);
        uint32_t *d = nullptr;
        dmalloc(d, P.N * 4);
        HIPCHECK(hipMemcpy(d, h.data(), P.N * 4, hipMemcpyHostToDevice));
        d_perm_[elt] = d;
        std::array<unsigned char, 32> rows{};
        for (size_t a = 0; a < ((size_t)1 << P.logn1) && a < rows.size(); ++a) rows[a] = (unsigned char)(h[a << kRowLog] >> kRowLog);
        perm_rows_[elt] = rows; // the one source row of every row of the permuted polynomial (the ring-in-LDS kernels take it with their arguments)
        return d;
    }

    const uint32_t *gather(uint32_t elt)
    {
        auto it = d_gather_.find(elt);
        if (it != d_gather_.end()) return it->second;
        const std::vector<uint32_t> h = P.galois_gather_coeff(elt);
        uint32_t *d = nullptr;
        dmalloc(d, P.N * 4);
        HIPCHECK(hipMemcpy(d, h.data(), P.N * 4, hipMemcpyHostToDevice));
        d_gather_[elt] = d;
        return d;
    }

    void set_dual_stream(bool on) { dual_stream_ = on; }
    void fill_uniform(u64 *dst, u64 n_polys, const uint8_t *prime_of, u32 period, u64 seed, u64 first_poly = 0)
    {
        use();
        if (period == 0 || period > 64) throw std::invalid_argument("prime map period must be in [1, 64]");
        PrimeMap pm;
        pm.period = period;
        for (u32 i = 0; i < period; ++i) {
            if (prime_of[i] >= P.K) throw std::invalid_argument("prime index out of range");
            pm.prime_of[i] = prime_of[i];
        }
        const u64 total = n_polys * P.N;
        hipLaunchKernelGGL(k_fill_uniform, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream_, dst, n_polys, P.logn, d_primes_, pm, seed, first_poly);
        HIPCHECK(hipGetLastError());
    }

    // ---- per-op sequences ------------------------------------------------------------------------------
    // do the u64 ranges [p, p + np) and [q, q + nq) share an element?
    static bool ranges_overlap(const u64 *p, size_t np, const u64 *q, size_t nq) { return np && nq && p < q + nq && q < p + np; }
    void check_level(int L) const
    {
        if (L < 1 || (size_t)L > P.Ltop) throw std::invalid_argument("level out of range");
    }
    void addsub(int L, int size, u64 n, const u64 *a, const u64 *b, Indexer ix, u64 *out, bool sub)
    {
        use();
        check_level(L);
        launch_addsub(env_, L, size, n, a, b, ix, out, sub);
        HIPCHECK(hipGetLastError());
    }
    void multiply(int L, u64 n, const u64 *a, const u64 *b, Indexer ix, u64 *out)
    {
        use();
        check_level(L);
        if (P.scheme != kSchemeCKKS) throw std::invalid_argument("he355_multiply implements the CKKS (NTT-form) product");
        launch_mul3(env_, L, n, a, b, ix, out);
        HIPCHECK(hipGetLastError());
    }

    struct Scratch {
        KsBuffers ks;
        u64 *rlr; // [C][3][N]  tail of the rescale prime after the inverse row pass
        u64 *f;   // [C][3][L][N]
    };
    // layout of the scratch arena (one per stream) for `c` ops at level L
    size_t scratch_words_per_op(int L) const
    {
        const size_t N = P.N, LN = (size_t)L * N;
        return 2 * LN + LN + LN + (size_t)(L + 1) * LN + 2 * LN + 2 * N + 2 * N + 2 * LN + 3 * N + 3 * LN;
    }
    // Ciphertexts per chunk for a batch of n at level L: chunk_ (default 1024, HE355_CHUNK / he355_set_chunk), halved until the scratch
    // arena(s) it needs -- two when the batch is cut and the chunks alternate between the streams -- are RESERVED: each arena is checked by
    // itself (growing one frees only that one), against 0.9 of the memory that is free now plus what this context can give back (the
    // arena being replaced, the pool's cached blocks); a hipMalloc that fails all the same (another context or process took the memory
    // between the query and the call) halves the chunk again instead of failing the operation.  On return scratch(c, L, which) does not
    // allocate.
    size_t chunk_ops(u64 n, int L, bool may_dual)
    {
        size_t c = std::min<u64>(chunk_, n ? n : 1);
        const size_t per_op = scratch_words_per_op(L) * 8;
        for (;;) {
            const int arenas = (may_dual && n > c) ? 2 : 1;
            const size_t need = per_op * c;
            size_t grow = 0, reclaim = pool_.cached_bytes();
            for (int a = 0; a < arenas; ++a) {
                const size_t have = a ? scratch2_bytes_ : scratch_bytes_;
                if (need > have) { grow += need; reclaim += have; }
            }
            bool fits = grow == 0;
            if (!fits) {
                size_t free_b = 0, total_b = 0;
                fits = hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)grow <= 0.9 * (double)(free_b + reclaim);
            }
            if (fits) {
                try {
                    for (int a = 0; a < arenas; ++a) reserve_arena(a, need);
                    return c;
                } catch (const OutOfDeviceMemory &) {
                    if (c == 1) throw;
                }
            } else if (c == 1) {
                throw OutOfDeviceMemory("HIP error: out of device memory: the key-switch scratch of one ciphertext (" + std::to_string(per_op) + " bytes) does not fit");
            }
            c = (c + 1) / 2;
        }
    }
    void reserve_arena(int which, size_t need)
    {
        u64 *&arena = which ? scratch2_ : scratch_;
        size_t &arena_bytes = which ? scratch2_bytes_ : scratch_bytes_;
        if (need <= arena_bytes) return;
        HIPCHECK(hipStreamSynchronize(stream_));
        HIPCHECK(hipStreamSynchronize(stream2_));
        pool_.raw_free(arena);
        arena = nullptr;
        arena_bytes = 0;
        dmalloc(arena, need);
        arena_bytes = need;
    }
    Scratch scratch(size_t c, int L, int which = 0)
    {
        const size_t N = P.N, LN = (size_t)L * N;
        const size_t per_op = scratch_words_per_op(L);
        reserve_arena(which, per_op * c * 8);
        u64 *arena = which ? scratch2_ : scratch_;
        Scratch s;
        u64 *p = arena;
        s.ks.c01 = p; p += c * 2 * LN; s.ks.c01_item_stride = 2 * LN;
        s.ks.c2n = p; p += c * LN;
        s.ks.c2r = p; p += c * LN;
        s.ks.d = p; p += c * (size_t)(L + 1) * LN;
        s.ks.t = p; p += c * 2 * LN;
        s.ks.tp = p; p += c * 2 * N;
        s.ks.tpr = p; p += c * 2 * N;
        s.ks.e = p; p += c * 2 * LN;
        s.rlr = p; p += c * 3 * N;
        s.f = p;
        return s;
    }

    // K2, K3, mod-down; result added into B.c01.  with_tail: also start the rescale (tail of prime L-1)
    // rescale_out != null (size-2 result wanted at level L-1): when the fused path applies, the rescale is finished here too
    // and the function returns true (the caller skips rescale_tail).
    // operands of a ct x ct multiply whose c0, c1 k_k1 did not write (tensor_in_k3): the fused k_k3 computes them where it adds them in
    struct TensorOperands {
        const u64 *a = nullptr, *b = nullptr;
        Indexer ix{};
        u64 op_offset = 0;
        // rotations (a, b null): polynomial 1 of the ciphertext k_k1 prepares is zero (1) or the addend's (2, c1_src = the addend rows of
        // this chunk) and is taken from there by the fused k_k3 instead of being written into c01 by k_k1
        int c1_mode = 0;
        const u64 *c1_src = nullptr;
        // c1_mode 4: the rotation's input and its permutation (K3Fuse::gsrc): k_k3 gathers the permuted c0 and c1 itself
        const u64 *gsrc = nullptr;
        const uint32_t *gperm = nullptr;
        u64 gsrc_op_offset = 0;
    };
    // true when key_switch_tail will take a fused path for this batch, i.e. when k_k3's epilogue is where c0, c1 are consumed
    // The fused mod-down runs the special prime's tiles as a launch of their own, ahead of the data primes' (their epilogue needs its
    // result): n1 rows x nc / 8 op-groups of blocks.  With a handful of ciphertexts at a small ring that launch is a few dozen blocks
    // on 256 CUs -- as long as the data primes' launch and nearly idle -- and the unfused sequence (every prime's tiles in ONE launch,
    // then the two floor kernels) is the shorter chain: fused from 128 such blocks up (HE355_K3_FUSE=all: always).
    bool fuse_pays(const KernelEnv &e, u64 nc, int L) const
    {
        if (k3_fuse_policy() == 2) return true;
        // ... or from 1536 (tile, op-group) units of the data primes' launch up: at L = 16 that launch is 16 times the special prime's, and what
        // the fused epilogue saves (the sums' trip through HBM, k_floor_rows) outweighs the idle launch ahead of it from 17 ciphertexts on at
        // N = 2^15 instead of 25 (batch 20 / 24 of the headline shape: 1.64 -> 1.45 / 1.79 -> 1.52 ms, DotProduct -9 %; profiles/r05_latency_boundary.txt)
        const u64 sp_blocks = ((u64)1 << e.logn1) * ((nc + 7) / 8);
        return sp_blocks >= 128 || sp_blocks * (u64)L >= 1536;
    }
    // he355_rotate_sum: the level's sum formed by k_k3 itself (KsGroups::sum_out).  The launch then has (tiles x n / 8) blocks however many
    // groups the level has, so it needs enough of them to fill the chip, groups of a multiple of eight ciphertexts, the fused path for
    // every chunk, and counts that leave bit 31 free.
    bool level_sum_pays(const KernelEnv &e, int L, u64 n) const
    {
        if (n % 8 || !k3_can_fuse(e) || !fuse_pays(e, n, L) || (u64)chunk_ < n) return false; // (a launch holds whole groups, and takes the fused path)
        return (((u64)L << e.logn1) * (n / 8)) >= 512;                                      // blocks of the data-prime launch
    }
    bool tensor_in_k3(const KernelEnv &env_, int L, u64 nc, const KsBuffers &B) const
    {
        return !latency_shape_env(env_, nc) && k3_can_fuse(env_) && fuse_pays(env_, nc, L) && B.c01_item_stride == 2 * (size_t)L * P.N;
    }
    // groups (grouped rotations, fused path only): per-group keys; g_off: index of the chunk's first op in the grouped batch
    bool key_switch_tail(const KernelEnv &env_, int L, u64 nc, const Scratch &S, const KsBuffers &B, const u64 *key, bool with_tail,
                         hipEvent_t after_k2 = nullptr, u64 *rescale_out = nullptr, const TensorOperands *ten = nullptr, const KsGroups *groups = nullptr,
                         u64 g_off = 0)
    {
        if (groups && (rescale_out || with_tail || !(k3_can_fuse(env_) && B.c01_item_stride == 2 * (size_t)L * P.N)))
            throw std::logic_error("key_switch_tail: grouped keys are for plain rotations into a ciphertext slab");
        const bool lat = !groups && latency_shape_env(env_, nc); // (a grouped launch always takes the throughput shape)
        auto with_operands = [&](K3Fuse f) {
            if (ten) { f.ta = ten->a; f.tb = ten->b; f.tix = ten->ix; f.t_op_offset = ten->op_offset; f.c1_mode = ten->c1_mode; f.c1_src = ten->c1_src; f.gsrc = ten->gsrc; f.gperm = ten->gperm; f.gsrc_op_offset = ten->gsrc_op_offset; }
            return f;
        };
        const size_t N = P.N, LN = (size_t)L * N;
        const int SP = (int)P.K - 1;
        if (lat) {
            ++paths_.ks_latency;
            // Few ciphertexts (HEBench's Latency category is batch 1: ckks eltwise .cpp:138-141): the throughput shape would leave one
            // wave walking all digits of a tile and one lane walking all targets of a column while the chip idles.  Same kernels,
            // unfused, with the serial loops dealt to more blocks: targets of a column over kLatTargets blocks (k_k2n, k_floor_colsn),
            // digits of a tile over kLatSplit (u64 engine: kLatSplitU64) single-wave blocks whose partial sums k_k3_combine adds (k_k3).
            launch_k2(env_, L, nc, B, nullptr, 0, kLatTargets);
            if (after_k2) HIPCHECK(hipEventRecord(after_k2, env_.stream));
            u64 *part = latency_partials((size_t)std::max(kLatSplit, kLatSplitU64) * nc * 2 * (L + 1) * N, env_.stream == stream2_ ? 1 : 0);
            launch_k3(env_, L, nc, B, key, K3_ALL, nullptr, kLatSplit, part, kLatSplitU64);
            launch_k3_combine(env_, L, nc, B, kLatSplit, part, kLatSplitU64);
            launch_floor_cols(env_, SP, L, nc * 2, B.tpr, B.e, 0, 0, nullptr, 0, kLatTargets);
            return key_switch_floor_rows(env_, L, nc, S, B, with_tail);
        }
        launch_k2(env_, L, nc, B);
        if (after_k2) HIPCHECK(hipEventRecord(after_k2, env_.stream));
        if (k3_can_fuse(env_) && B.c01_item_stride == 2 * LN && fuse_pays(env_, nc, L)) {
            // special prime first, its correction through the column pass, then the data primes with the mod-down finished
            // inside K3 (the sums never go to HBM)
            ++paths_.ks_fused;
            if (groups && groups->sum_out) ++paths_.level_sum_launches_in_k3;
            launch_k3(env_, L, nc, B, key, K3_SPECIAL_ONLY, nullptr, 1, nullptr, 0, groups, g_off);
            if (rescale_out && L >= 2) {
                // Mod-down + rescale with ONE column pass and ONE row transform per target: only the prime the rescale divides out needs
                // the mod-down correction by itself (its tiles run first, mod-down only); for every other prime the two corrections are
                // combined in coefficient form, delta2 + P^-1 * delta1, inside one k_floor_colsn launch that reads both sources (the
                // special prime's sums and the divided-out prime's tail) -- 16 column passes per polynomial instead of 31, and the
                // mod-down correction slab is neither written for those primes nor read back.
                launch_floor_cols(env_, SP, 1, nc * 2, B.tpr, B.e, /*tgt_first*/ L - 1, /*dst_ntgt*/ L);
                const K3Fuse last = with_operands(K3Fuse{B.e, B.c01, B.c01_item_stride, L - 1, L, nullptr, nullptr});
                launch_k3(env_, L, nc, B, key, K3_DATA_ONLY, &last);
                launch_rows_inv_select(env_, L - 1, nc * 2, B.c01 + (size_t)(L - 1) * N, (u64)LN, S.rlr);
                launch_floor_cols(env_, L - 1, L - 1, nc * 2, S.rlr, S.f, 0, L - 1, /*src2*/ B.tpr, SP);
                const K3Fuse rest = with_operands(K3Fuse{B.e, B.c01, B.c01_item_stride, 0, L - 1, S.f, rescale_out});
                launch_k3(env_, L, nc, B, key, K3_DATA_ONLY, &rest);
                return true;
            }
            launch_floor_cols(env_, SP, L, nc * 2, B.tpr, B.e);
            const K3Fuse fuse = with_operands(K3Fuse{B.e, B.c01, B.c01_item_stride, 0, L, nullptr, nullptr});
            launch_k3(env_, L, nc, B, key, K3_DATA_ONLY, &fuse, 1, nullptr, 0, groups, g_off);
            if (with_tail) launch_rows_inv_select(env_, L - 1, nc * 2, B.c01 + (size_t)(L - 1) * N, (u64)LN, S.rlr);
            return false;
        }
        if (ten) throw std::logic_error("key_switch_tail: c0, c1 were left to a fused k_k3 that is not running");
        ++paths_.ks_unfused;
        launch_k3(env_, L, nc, B, key, K3_ALL, nullptr, 1, nullptr, 0, groups, g_off);
        launch_floor_cols(env_, SP, L, nc * 2, B.tpr, B.e);
        return key_switch_floor_rows(env_, L, nc, S, B, with_tail);
    }
    // the row half of the unfused mod-down: c01 += (t - NTT(e)) * P^-1, optionally starting the rescale (tail of prime L-1)
    bool key_switch_floor_rows(const KernelEnv &env_, int L, u64 nc, const Scratch &S, const KsBuffers &B, bool with_tail)
    {
        const size_t LN = (size_t)L * P.N;
        const int SP = (int)P.K - 1;
        FloorRowsArgs fr;
        fr.src_prime = SP; fr.n_tgt = L; fr.n_src = 2;
        fr.cols = B.e;
        fr.tsrc = B.t; fr.tsrc_op_stride = 2 * LN; fr.tsrc_poly_stride = LN;
        fr.addend = B.c01; fr.add_op_stride = B.c01_item_stride; fr.add_poly_stride = LN;
        fr.out = B.c01; fr.out_op_stride = B.c01_item_stride; fr.out_poly_stride = LN;
        fr.tail_prime = with_tail ? L - 1 : -1;
        fr.tail = S.rlr;
        launch_floor_rows(env_, nc, fr);
        return false;
    }
    void rescale_tail(const KernelEnv &env_, int L, int size, u64 nc, const Scratch &S, const u64 *src, u64 src_op_stride, u64 *out)
    {
        const size_t N = P.N, LN = (size_t)L * N, L1N = (size_t)(L - 1) * N;
        launch_floor_cols(env_, L - 1, L - 1, nc * size, S.rlr, S.f, 0, 0, nullptr, 0, latency_shape(nc) ? kLatTargets : 1);
        FloorRowsArgs fr;
        fr.src_prime = L - 1; fr.n_tgt = L - 1; fr.n_src = size;
        fr.cols = S.f;
        fr.tsrc = src; fr.tsrc_op_stride = src_op_stride; fr.tsrc_poly_stride = LN;
        fr.addend = nullptr; fr.add_op_stride = 0; fr.add_poly_stride = 0;
        fr.out = out; fr.out_op_stride = (u64)size * L1N; fr.out_poly_stride = L1N;
        fr.tail_prime = -1; fr.tail = nullptr;
        launch_floor_rows(env_, nc, fr);
    }
    // latency shape of the key switch (key_switch_tail): batches of at most lat_max_ ciphertexts, CKKS pipeline
    // digit groups per fp64-engine tile (480 tiles x 4 single-wave blocks fill the chip once at batch 1), per u64-engine tile (64 tiles, rows
    // 2.5x as long), blocks per column for the targets of k_k2n / k_floor_colsn
    static constexpr int kLatTargets = 8;
    // Digit groups per tile of the latency shape (n_split > 1 is what selects the shape).  fp64-engine
    // tiles: 480 x 2 single-wave blocks are ONE round of the chip's 1024 one-wave slots, 480 x 4 were two rounds of half the work each with
    // twice the start-ups and partial sums (batch 1: 0.326 -> 0.312 ms, batch 8: 0.98 -> 0.91 ms; 3 and 8 groups measured slower).
    static constexpr int kLatSplit = 2, kLatSplitU64 = 8;
    // Where the latency shape stops paying is a matter of rows, not of ciphertexts: the throughput kernels fill the chip from about 2^17
    // coefficients per residue on (profiles/r05_latency_boundary.txt: N = 2^15 crosses between 4 and 5 ciphertexts at depth 6 and 16, 2^14
    // between 6 and 8, 2^13 at 12), so the default limit is 2^17 / N ciphertexts, at most 12; he355_set_latency_max replaces it.
    u64 lat_limit() const { return lat_auto_ ? std::min<u64>(12, std::max<u64>(1, ((u64)1 << 17) / P.N)) : lat_max_; }
    bool latency_shape(u64 nc) const { return P.scheme == kSchemeCKKS && nc <= lat_limit() && P.K >= 2; }
    // ... for a given kernel environment (a BFV context runs its rotation chains in the NTT domain on the CKKS pipeline: ntt_env)
    bool latency_shape_env(const KernelEnv &e, u64 nc) const { return e.scheme == kSchemeCKKS && nc <= lat_limit() && P.K >= 2; }
    // Ring-in-LDS shape (he355_kernels_lds.hip): N <= 8192, NTT-domain pipeline, batches up to lds_limit() -- two
    // launches of L^2 + 2L one-polynomial workgroups per ciphertext instead of six launches through HBM.  Where the throughput shape's
    // better use of the chip overtakes it was measured (profiles/r06_lds_shape.txt); he355_set_lds_max / HE355_LDS_MAX replace the rule.
    // The rule: while k_lds_digits' grid, (L + 1) L blocks per ciphertext, runs in at most one and a half rounds of the chip -- 256 CUs, one
    // 8-wave block each at N = 8192 (two, four, eight blocks per CU for the smaller rings): 64 ciphertexts at {60, 40, 60} (32: 61 against 78 us
    // per key switch, 64: 89 against 95, 96: 131 against 113), 32 at {60, 40, 40, 60}; beyond that the HBM throughput shapes use the chip
    // better (profiles/r06_lds_shape.txt; a "one inverse transform, every target" form of the kernels for larger batches was built, bit-exact,
    // and lost everywhere: same file, tools/patches/r06_lds_source_major.patch).  he355_set_lds_max / HE355_LDS_MAX replace the rule.
    u64 lds_limit(const KernelEnv &e, int L) const
    {
        if (!lds_auto_) return lds_max_;
        const u64 blocks = (u64)384 << (3 - std::min(3, e.logn1));
        return std::max<u64>(1, blocks / ((u64)(L + 1) * (u64)L));
    }
    bool lds_shape(const KernelEnv &e, int L, u64 nc) const { return e.scheme == kSchemeCKKS && ks_lds_supported(e, L) && nc <= lds_limit(e, L); }
    // scratch of the shape: the arena behind c01 (k_lds_digits' partial products; (2L + 2) L N words per ciphertext fit there for L <= 8)
    u64 *lds_part(const Scratch &S, int L, u64 nc) const
    {
        const size_t need = (size_t)ks_lds_part_words(env_, L), have = scratch_words_per_op(L) - 2 * (size_t)L * P.N;
        if (need > have) throw std::logic_error("ring-in-LDS key switch: the arena is too small for the partial products");
        (void)nc;
        return S.ks.c2n;
    }
    // the kernel environment of a batch on stream `which`
    // ntt: the NTT-domain (CKKS) pipeline whatever the context's scheme (ntt_env)
    KernelEnv batch_env(u64 /*nc*/, int which = 0, bool ntt = false) const
    {
        KernelEnv env = env_;
        env.stream = which ? stream2_ : stream_;
        if (ntt) env.scheme = kSchemeCKKS;
        return env;
    }
    u64 *latency_partials(size_t elems, int which) // one buffer per stream: chunks of the two streams are in flight together
    {
        if (elems * 8 > lat_part_bytes_[which]) {
            HIPCHECK(hipStreamSynchronize(stream_));
            HIPCHECK(hipStreamSynchronize(stream2_));
            pool_.raw_free(lat_part_[which]);
            lat_part_[which] = nullptr; lat_part_bytes_[which] = 0;
            dmalloc(lat_part_[which], elems * 8);
            lat_part_bytes_[which] = elems * 8;
        }
        return lat_part_[which];
    }
    void require_keyswitch() const
    {
        if (P.K < 2) throw std::invalid_argument("encryption parameters do not support key switching");
    }

    void multiply_relin(int L, u64 n, const u64 *a, const u64 *b, Indexer ix, bool rescale, u64 *out)
    {
        use();
        check_level(L);
        require_keyswitch();
        if (P.scheme != kSchemeCKKS) throw std::invalid_argument("he355_multiply_relin implements the CKKS pipeline");
        if (!d_relin_) throw std::invalid_argument("relinearization key not set");
        if (rescale && L < 2) throw std::invalid_argument("cannot rescale at the last level");
        const size_t N = P.N, LN = (size_t)L * N;
        // Alternate chunks between two streams, each with its own scratch arena: the ALU-bound key-product kernel
        // of one chunk overlaps the HBM-bound multiply / digit-lift / floor kernels of the other.
        const size_t chunk = chunk_ops(n, L, dual_stream_); // this call's chunk size
        const bool dual = dual_stream_ && n > chunk;
        // The second stream starts half a pipeline late (after the first chunk's K1+K2 on the first stream): from then on
        // one stream's ALU-bound kernels (K3, floor column pass) run beside the other's HBM-bound ones (K1, K2, floor row pass)
        // instead of beside their own kind.
        // (starting both streams together measured slower: HISTORY.md)
        if (dual) {
            (void)scratch(std::min<u64>(chunk, n), L, 0);
            (void)scratch(std::min<u64>(chunk, n), L, 1);
        }
        // k_k3 reads the operand rows while it writes results: only when `out` is a slab of its own (it always was meant to be)
        bool out_overlaps_operands = false;
        if (n) {
            auto overlap = [](const u64 *p, size_t np, const u64 *q, size_t nq) { return p < q + nq && q < p + np; };
            const size_t out_words = n * 2 * (size_t)(rescale ? L - 1 : L) * N;
            const u64 a_lo = idx_a(ix, 0), a_hi = idx_a(ix, n - 1), b_lo = ix.pairwise ? ix.b_base : ix.b_base, b_hi = ix.pairwise ? ix.b_base + n - 1 : ix.b_base + std::min<u64>(n, ix.b1) - 1;
            out_overlaps_operands = overlap(out, out_words, a + a_lo * 2 * LN, (size_t)(a_hi - a_lo + 1) * 2 * LN) ||
                                    overlap(out, out_words, b + b_lo * 2 * LN, (size_t)(b_hi - b_lo + 1) * 2 * LN);
        }
        u64 ci = 0;
        for (u64 off = 0; off < n; off += chunk, ++ci) {
            const u64 nc = std::min<u64>(chunk, n - off);
            const int which = dual ? (int)(ci & 1) : 0;
            const KernelEnv env = batch_env(nc, which);
            Scratch S = scratch(std::min<u64>(chunk, n), L, which);
            KsBuffers B = S.ks;
            if (!rescale) { B.c01 = out + off * 2 * LN; B.c01_item_stride = 2 * LN; }
            if (lds_shape(env, L, nc) && !out_overlaps_operands) {
                // a ring that fits LDS: tensor product, key switch and the add in two launches (the operands are read where they lie)
                ++paths_.ks_lds;
                LdsKsOperands src;
                src.mode = LDSKS_MUL; src.a = a; src.b = b; src.ix = ix; src.op_offset = off;
                launch_ks_lds(env, L, nc, src, d_relin_, lds_part(S, L, nc), B.c01, B.c01_item_stride);
                if (rescale) launch_rescale_lds(env, L, 2, nc, B.c01, 2 * LN, out + off * 2 * (size_t)(L - 1) * N);
                continue;
            }
            // c0, c1 of the tensor product: written by k_k1, or (fused key switch) computed by k_k3 where it adds them in -- k_k1 is
            // HBM-bound and then reads half and writes a third of what it did, k_k3 is not and reads the operand rows instead of c01
            const bool in_k3 = tensor_in_k3(env, L, nc, B) && !out_overlaps_operands;
            TensorOperands ten;
            ten.a = a; ten.b = b; ten.ix = ix; ten.op_offset = off;
            launch_k1(env, L, K1_MUL, nc, off, a, b, ix, nullptr, B, nullptr, in_k3);
            const bool fork_here = dual && ci == 0;
            u64 *ro = rescale ? out + off * 2 * (size_t)(L - 1) * N : nullptr;
            const bool done = key_switch_tail(env, L, nc, S, B, in_k3 ? relin_scaled() : d_relin_, rescale, fork_here ? ev_fork_ : nullptr, ro, in_k3 ? &ten : nullptr);
            if (fork_here) HIPCHECK(hipStreamWaitEvent(stream2_, ev_fork_, 0));
            if (rescale && !done) rescale_tail(env, L, 2, nc, S, B.c01, 2 * LN, ro);
        }
        if (dual) {
            HIPCHECK(hipEventRecord(ev_join_, stream2_));
            HIPCHECK(hipStreamWaitEvent(stream_, ev_join_, 0));
        }
        HIPCHECK(hipGetLastError());
    }
    void relinearize(int L, u64 n, const u64 *ct3, u64 *out, bool rescale = false)
    {
        use();
        check_level(L);
        require_keyswitch();
        if (!d_relin_) throw std::invalid_argument("relinearization key not set");
        const size_t N = P.N, LN = (size_t)L * N;
        if (P.scheme == kSchemeBFV) {
            if (rescale) throw std::invalid_argument("rescale is a CKKS operation");
            if (ranges_overlap(out, n * 2 * LN, ct3, n * 3 * LN)) throw std::invalid_argument("relinearize: `out` overlaps the size-3 input");
            const size_t chunk = chunk_ops(n, L, false); // this call's chunk size
            for (u64 off = 0; off < n; off += chunk) {
                const u64 nc = std::min<u64>(chunk, n - off);
                Scratch S = scratch(std::min<u64>(chunk, n), L); // arena sized for the batch actually processed
                KsBuffers B = S.ks;
                B.c01 = out + off * 2 * LN; B.c01_item_stride = 2 * LN;
                const u64 *src = ct3 + off * 3 * LN;
                // out = (c0, c1) of each size-3 ciphertext (read where they lie by the last kernel) + the key-switched c2
                bfv_key_switch(L, nc, S, B, d_relin_, src + 2 * LN, 3 * LN, src, 3 * LN);
            }
            HIPCHECK(hipGetLastError());
            return;
        }
        if (rescale && L < 2) throw std::invalid_argument("cannot rescale at the last level");
        Indexer ix{};
        const size_t chunk = chunk_ops(n, L, false); // this call's chunk size
        for (u64 off = 0; off < n; off += chunk) {
            const u64 nc = std::min<u64>(chunk, n - off);
            Scratch S = scratch(std::min<u64>(chunk, n), L); // arena sized for the batch actually processed
            KsBuffers B = S.ks;
            if (!rescale) { B.c01 = out + off * 2 * LN; B.c01_item_stride = 2 * LN; }
            const KernelEnv env = batch_env(nc);
            // on the fused path k_k3 reads c0, c1 and the NTT-form c2 from the size-3 input where it lies (k_k1 copies nothing: it only
            // sends c2 through the inverse row pass); `out` must then be a slab of its own
            auto overlap = [](const u64 *p, size_t np, const u64 *q, size_t nq) { return p < q + nq && q < p + np; };
            const bool out_apart = !overlap(out, n * 2 * (size_t)(rescale ? L - 1 : L) * N, ct3, n * 3 * LN);
            if (lds_shape(env, L, nc) && out_apart) {
                ++paths_.ks_lds;
                LdsKsOperands src;
                src.mode = LDSKS_PLAIN;
                src.tgt = ct3 + off * 3 * LN + 2 * LN; src.tgt_op_stride = 3 * LN;
                src.add = ct3 + off * 3 * LN; src.add_op_stride = 3 * LN;
                launch_ks_lds(env, L, nc, src, d_relin_, lds_part(S, L, nc), B.c01, B.c01_item_stride);
                if (rescale) launch_rescale_lds(env, L, 2, nc, B.c01, 2 * LN, out + off * 2 * (size_t)(L - 1) * N);
                continue;
            }
            const bool in_k3 = tensor_in_k3(env, L, nc, B) && out_apart;
            TensorOperands ten;
            ten.c1_mode = 3;
            ten.c1_src = ct3 + off * 3 * LN;
            launch_k1(env, L, K1_CT3, nc, off, ct3, nullptr, ix, nullptr, B, nullptr, false, in_k3);
            u64 *ro = rescale ? out + off * 2 * (size_t)(L - 1) * N : nullptr;
            const bool done = key_switch_tail(env, L, nc, S, B, d_relin_, rescale, nullptr, ro, in_k3 ? &ten : nullptr);
            if (rescale && !done) rescale_tail(env, L, 2, nc, S, B.c01, 2 * LN, ro);
        }
        HIPCHECK(hipGetLastError());
    }
    void plain_op(int L, int size, u64 n, const u64 *ct, const u64 *pt, Indexer ix, u64 *out, int mode)
    {
        use();
        check_level(L);
        if (P.scheme != kSchemeCKKS) throw std::invalid_argument("plaintext operands are NTT-form CKKS plaintexts");
        if (size < 1 || size > 3) throw std::invalid_argument("ciphertext size must be 1..3");
        launch_plain_op(env_, L, size, n, ct, pt, ix, out, mode);
        HIPCHECK(hipGetLastError());
    }
    void mod_switch_drop(int L, int L_to, u64 n_polys, const u64 *in, u64 *out)
    {
        use();
        check_level(L);
        if (P.scheme != kSchemeCKKS) throw std::invalid_argument("he355_mod_switch_drop is the CKKS modulus switch");
        if (L_to < 1 || L_to > L) throw std::invalid_argument("target level out of range");
        launch_drop_residues(env_, L, L_to, n_polys, in, out);
        HIPCHECK(hipGetLastError());
    }
    void sum(int L, int size, u64 n, const u64 *in, u64 *out)
    {
        use();
        check_level(L);
        if (size < 1 || size > 3) throw std::invalid_argument("ciphertext size must be 1..3");
        if (n < 1) throw std::invalid_argument("nothing to sum");
        launch_sum_cts(env_, L, size, n, in, out);
        HIPCHECK(hipGetLastError());
    }
    void multiply_accumulate(int L, u64 rows, u64 cols, u64 inner, const u64 *a, u64 a_stride_i, u64 a_stride_k, const u64 *b, u64 b_stride_k,
                             u64 b_stride_j, u64 *out)
    {
        use();
        check_level(L);
        if (P.scheme != kSchemeCKKS) throw std::invalid_argument("he355_multiply_accumulate implements the CKKS (NTT-form) product");
        if (inner < 1 || inner > 0x7fffffff) throw std::invalid_argument("inner dimension out of range");
        if (rows * cols == 0) return;
        launch_mul3_acc(env_, L, rows, cols, (int)inner, a, a_stride_i, a_stride_k, b, b_stride_k, b_stride_j, out);
        HIPCHECK(hipGetLastError());
    }
    void rescale(int L, int size, u64 n, const u64 *in, u64 *out)
    {
        use();
        check_level(L);
        if (P.scheme != kSchemeCKKS) throw std::invalid_argument("he355_rescale is a CKKS operation");
        if (L < 2) throw std::invalid_argument("cannot rescale at the last level");
        if (size < 1 || size > 3) throw std::invalid_argument("ciphertext size must be 1..3");
        const size_t N = P.N, LN = (size_t)L * N;
        const size_t chunk = chunk_ops(n, L, false); // this call's chunk size
        for (u64 off = 0; off < n; off += chunk) {
            const u64 nc = std::min<u64>(chunk, n - off);
            Scratch S = scratch(std::min<u64>(chunk, n), L); // arena sized for the batch actually processed
            const u64 *src = in + off * size * LN;
            if (lds_shape(env_, L, nc) && !ranges_overlap(in, n * size * LN, out, n * size * (size_t)(L - 1) * N)) { // (the blocks of an op read all of its input)
                launch_rescale_lds(env_, L, size, nc, src, (u64)size * LN, out + off * size * (size_t)(L - 1) * N);
                continue;
            }
            launch_rows_inv_select(env_, L - 1, nc * size, src + (size_t)(L - 1) * N, LN, S.rlr);
            rescale_tail(env_, L, size, nc, S, src, (u64)size * LN, out + off * size * (size_t)(L - 1) * N);
        }
        HIPCHECK(hipGetLastError());
    }
    // `addend` (optional, [n][2][L][N]; may alias `out`, never `in`): out = addend + galois(in) -- the add_inplace that follows
    // a rotation in accumulateCKKS/BFV and in the row-major MatMult is folded into the first kernel of the rotation.
    // ntt_form (BFV contexts): the ciphertexts are held in NTT form (a rotation chain that was transformed on the way in runs on the fused
    // NTT-domain pipeline: accumulate, rotate_sum)
    void apply_galois(int L, u64 n, const u64 *in, uint32_t elt, u64 *out, const u64 *addend = nullptr, bool ntt_form = false)
    {
        use();
        check_level(L);
        require_keyswitch();
        if (!(elt & 1) || elt >= 2 * P.N) throw std::invalid_argument("Galois element is not valid");
        const u64 *key = galois_key(elt);
        if (!key) throw std::invalid_argument("Galois key not present");
        // every kernel of the pipeline reads `in` while later ones already write `out`: any overlap (not just in == out) corrupts the input
        if (ranges_overlap(in, n * 2 * (size_t)L * P.N, out, n * 2 * (size_t)L * P.N)) throw std::invalid_argument("apply_galois cannot run in place: `out` overlaps `in`");
        if (P.scheme == kSchemeBFV && !ntt_form) {
            const uint32_t *gt = gather(elt);
            const size_t LN = (size_t)L * P.N;
            const size_t chunk = chunk_ops(n, L, false); // this call's chunk size
            for (u64 off = 0; off < n; off += chunk) {
                const u64 nc = std::min<u64>(chunk, n - off);
                Scratch S = scratch(std::min<u64>(chunk, n), L); // arena sized for the batch actually processed
                KsBuffers B = S.ks;
                B.c01 = out + off * 2 * LN; B.c01_item_stride = 2 * LN;
                launch_bfv_galois(env_, L, nc, in + off * 2 * LN, gt, B.c01, B.c01_item_stride, B.c2n, addend ? addend + off * 2 * LN : nullptr);
                bfv_key_switch(L, nc, S, B, key, B.c2n, LN);
            }
            HIPCHECK(hipGetLastError());
            return;
        }
        const uint32_t *pm = perm(elt);
        const size_t N = P.N, LN = (size_t)L * N;
        Indexer ix{};
        const size_t chunk = chunk_ops(n, L, false); // this call's chunk size
        for (u64 off = 0; off < n; off += chunk) {
            const u64 nc = std::min<u64>(chunk, n - off);
            Scratch S = scratch(std::min<u64>(chunk, n), L); // arena sized for the batch actually processed
            KsBuffers B = S.ks;
            B.c01 = out + off * 2 * LN; B.c01_item_stride = 2 * LN;
            const KernelEnv env = batch_env(nc, 0, true);
            if (lds_shape(env, L, nc)) {
                // a ring that fits LDS: permutation, key switch and the add in two launches (an addend may be `out`: the wave that reads a row
                // of it is the one that writes that row of the result, afterwards)
                ++paths_.ks_lds;
                LdsKsOperands src;
                src.mode = LDSKS_GALOIS; src.a = in; src.op_offset = off; src.perm = pm;
                const std::array<unsigned char, 32> &rows = perm_rows_.at(elt);
                std::copy(rows.begin(), rows.end(), src.perm_src_row);
                src.add = addend ? addend + off * 2 * LN : nullptr; src.add_op_stride = 2 * LN;
                launch_ks_lds(env, L, nc, src, key, lds_part(S, L, nc), B.c01, B.c01_item_stride);
                continue;
            }
            // polynomial 1 of the rotated ciphertext is zero (or the addend's): on the fused path k_k1 does not write it and k_k3 takes it
            // from where it is (the addend may be `out`: the wave that reads a row is the one that writes it, afterwards)
            const bool c1_in_k3 = tensor_in_k3(env, L, nc, B);
            TensorOperands ten;
            // (4 / 5: k_k3's epilogue gathers the permuted c0 from `in` itself -- and adds the addend's polynomial 0 to it -- so k_k1 writes two
            // rows instead of three and reads neither c0 nor the addend)
            ten.c1_mode = addend ? 5 : 4;
            ten.c1_src = addend ? addend + off * 2 * LN : nullptr;
            ten.gsrc = in; ten.gperm = pm; ten.gsrc_op_offset = off;
            launch_k1(env, L, K1_GALOIS, nc, off, in, nullptr, ix, pm, B, addend, false, c1_in_k3, nullptr, c1_in_k3);
            key_switch_tail(env, L, nc, S, B, key, false, nullptr, nullptr, c1_in_k3 ? &ten : nullptr);
        }
        HIPCHECK(hipGetLastError());
    }
    // Evaluator::rotate_internal: use the key of the step if present, otherwise the NAF decomposition
    void rotate(int L, u64 n, const u64 *in, int step, u64 *out, const u64 *addend = nullptr, bool ntt_form = false)
    {
        use();
        const size_t bytes = n * 2 * (size_t)L * P.N * 8;
        auto plain_copy = [&]() { // no rotation left: out = in (+ addend)
            if (addend) {
                Indexer ixp{};
                ixp.pairwise = 1;
                addsub(L, 2, n, in, addend, ixp, out, false);
            } else if (in != out) HIPCHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, stream_));
        };
        if (step == 0) { plain_copy(); return; }
        const uint32_t elt = P.galois_elt_from_step(step);
        if (!elt) throw std::invalid_argument("step count too large");
        if (galois_key(elt)) { apply_galois(L, n, in, elt, out, addend, ntt_form); return; }
        std::vector<int> naf;
        {
            const bool neg = step < 0;
            long v = neg ? -(long)step : step;
            for (int i = 0; v; ++i) {
                const int zi = (v & 1) ? 2 - (int)(v & 3) : 0;
                v = (v - zi) >> 1;
                if (zi) naf.push_back((neg ? -zi : zi) * (1 << i));
            }
        }
        if (naf.size() == 1) throw std::invalid_argument("Galois key not present");
        if (ranges_overlap(in, bytes / 8, out, bytes / 8)) throw std::invalid_argument("rotate cannot run in place: `out` overlaps `in`");
        std::vector<int> steps;
        for (int s : naf)
            if ((size_t)(s < 0 ? -s : s) != P.N / 2) steps.push_back(s); // a term of N/2 is no rotation
        if (steps.empty()) { plain_copy(); return; }
        if (addend == out && steps.size() > 1) throw std::invalid_argument("rotate_add through several Galois steps cannot add in place");
        if (steps.size() > 1 && bytes > rot_tmp_bytes_) {
            HIPCHECK(hipStreamSynchronize(stream_));
            pool_.raw_free(rot_tmp_);
            rot_tmp_ = nullptr; rot_tmp_bytes_ = 0;
            dmalloc(rot_tmp_, bytes);
            rot_tmp_bytes_ = bytes;
        }
        // ping-pong between out and the temporary so that the last rotation lands in out
        const u64 *cur = in;
        const size_t m = steps.size();
        for (size_t t = 0; t < m; ++t) {
            u64 *dst = ((m - 1 - t) % 2 == 0) ? out : rot_tmp_;
            const uint32_t e = P.galois_elt_from_step(steps[t]);
            if (!e || !galois_key(e)) throw std::invalid_argument("Galois key not present");
            apply_galois(L, n, cur, e, dst, t + 1 == m ? addend : nullptr, ntt_form); // the addend joins the last step only
            cur = dst;
        }
    }
    // NAF terms of a rotation step in the order Evaluator::rotate_internal applies them (least significant first; a term of
    // N/2 is no rotation); a step with its own Galois key is one term
    std::vector<int> rotation_terms(int step)
    {
        std::vector<int> terms;
        if (step == 0) return terms;
        const uint32_t elt = P.galois_elt_from_step(step);
        if (!elt) throw std::invalid_argument("step count too large");
        if (galois_key(elt)) { terms.push_back(step); return terms; }
        const bool neg = step < 0;
        long v = neg ? -(long)step : step;
        int n_naf = 0;
        for (int i = 0; v; ++i) {
            const int zi = (v & 1) ? 2 - (int)(v & 3) : 0;
            v = (v - zi) >> 1;
            if (!zi) continue;
            ++n_naf;
            const long t = (neg ? -zi : zi) * (1L << i);
            if ((size_t)(t < 0 ? -t : t) != P.N / 2) terms.push_back((int)t);
        }
        if (n_naf == 1) throw std::invalid_argument("Galois key not present");
        for (int t : terms) {
            const uint32_t e = P.galois_elt_from_step(t);
            if (!e || !galois_key(e)) throw std::invalid_argument("Galois key not present");
        }
        return terms;
    }
    // out = in + sum_j rotate(in, steps[j]): the inner loop of the row-major MatMult (bfv row .cpp:519-531, ckks row .cpp:502-514:
    // result = base; result += rotate_rows(base, j * spacers) for j = 1 .. dim2-1).  Every rotation is what Evaluator::rotate_internal
    // computes -- the step's own Galois key if present, else its NAF terms applied least significant first -- and all of them start from
    // the same ciphertext, so two steps whose term sequences share a prefix share that prefix's intermediate CIPHERTEXT bit for bit
    // (same operations on the same input).  The term sequences form a trie; each node is key-switched once, from its parent's
    // ciphertext, and added to the running sum as often as steps end there (modular additions commute, so the order of the adds is
    // free).  For steps j * 2^k, j = 1 .. 2^m - 1, every prefix of a NAF sequence is the NAF sequence of a smaller j: 127 key
    // switches instead of 313 for the 128-column products of BASELINE configs[4].
    //
    // Round 4: the trie is walked LEVEL BY LEVEL, all nodes of a level in ONE kernel sequence (grouped key switches: every node's n
    // ciphertexts are a group with its own Galois element and key, KsGroups).  A node-by-node walk issues 127 sequences over 64
    // ciphertexts each at configs[4] -- the small-grid regime, 7.0-7.7 us per ciphertext and key switch; a level is 13-50 nodes, i.e.
    // 800-3200 ciphertexts per sequence: 4.8 us (tools/ks_probe.py, profiles/r04_bfv_gather_fold_negative.txt).  The walk runs in the
    // NTT domain on the fused CKKS pipeline for BOTH schemes: a BFV ciphertext is transformed once on the way in and the sum once on the
    // way out; in between every operation (Galois permutation, the key switch's digit lifts, key products and mod-down, modular
    // additions) is the same exact map on residues in either representation, since the NTT is a bijection that commutes with all of
    // them -- (t - delta) P^-1 in coefficient form and (NTT(t) - NTT(delta)) P^-1 in NTT form are the same polynomial.
    // Returns the number of key switches issued.  Not in place.
    struct RotNode { uint32_t elt; int parent; u64 ends; std::vector<int> kids; int level, pos; };
    std::vector<RotNode> rotation_trie(const int *steps, u64 n_steps, size_t &depth)
    {
        std::vector<RotNode> trie(1, RotNode{0, -1, 0, {}, 0, 0}); // node 0: the input itself
        depth = 0;
        for (u64 j = 0; j < n_steps; ++j) {
            const std::vector<int> terms = rotation_terms(steps[j]);
            int at = 0;
            for (int t : terms) {
                const uint32_t te = P.galois_elt_from_step(t); // steps that differ by the row length are the same rotation, same key
                int next = -1;
                for (int k : trie[(size_t)at].kids)
                    if (trie[(size_t)k].elt == te) { next = k; break; }
                if (next < 0) {
                    next = (int)trie.size();
                    trie.push_back(RotNode{te, at, 0, {}, trie[(size_t)at].level + 1, 0});
                    trie[(size_t)at].kids.push_back(next);
                }
                at = next;
            }
            ++trie[(size_t)at].ends;
            depth = std::max(depth, terms.size());
        }
        return trie;
    }
    // device copy of a level's group tables: [perm pointers | key pointers | src_block | mult], grown as needed
    struct GroupTables { KsGroups g; const u32 *d_mult; };
    GroupTables upload_groups(const std::vector<const uint32_t *> &perms, const std::vector<const u64 *> &keys, const std::vector<u32> &src_block,
                              const std::vector<u32> &mult, u32 group_size)
    {
        const size_t G = perms.size(), bytes = (G * (8 + 8 + 4 + 4) + 255) & ~(size_t)255;
        // The tables live in a ring in HBM: every upload takes the next region, so a region is never rewritten while kernels that were
        // launched with it may still be running; when the ring wraps (or has to grow) the stream is drained first.  The copy itself is
        // synchronous (complete on return, whatever the runtime does with pageable memory), and everything that reads the region is
        // launched afterwards.
        if (bytes > groups_bytes_ || groups_next_ + bytes > groups_bytes_) {
            HIPCHECK(hipStreamSynchronize(stream_));
            if (bytes > groups_bytes_ / 4) {
                pool_.raw_free(d_groups_);
                d_groups_ = nullptr; groups_bytes_ = 0;
                dmalloc(d_groups_, std::max<size_t>(bytes * 8, (size_t)64 << 10));
                groups_bytes_ = std::max<size_t>(bytes * 8, (size_t)64 << 10);
            }
            groups_next_ = 0;
        }
        unsigned char *d_tab = d_groups_ + groups_next_;
        groups_next_ += bytes;
        std::vector<unsigned char> h(bytes, 0);
        std::memcpy(h.data(), perms.data(), G * 8);
        std::memcpy(h.data() + G * 8, keys.data(), G * 8);
        std::memcpy(h.data() + G * 16, src_block.data(), G * 4);
        std::memcpy(h.data() + G * 20, mult.data(), G * 4);
        HIPCHECK(hipMemcpy(d_tab, h.data(), bytes, hipMemcpyHostToDevice));
        GroupTables t;
        t.g.perm = reinterpret_cast<const uint32_t *const *>(d_tab);
        t.g.key = reinterpret_cast<const u64 *const *>(d_tab + G * 8);
        t.g.src_block = reinterpret_cast<const u32 *>(d_tab + G * 16);
        t.g.group_size = group_size;
        t.d_mult = reinterpret_cast<const u32 *>(d_tab + G * 20);
        return t;
    }
    // the kernel environment of the NTT-domain pipeline (a BFV context's tables are the same primes; only the data representation differs)
    KernelEnv ntt_env() const
    {
        KernelEnv e = env_;
        e.scheme = kSchemeCKKS;
        return e;
    }
    // NTT-form ciphertexts: out[g * gs + c] = apply_galois(in[src_block[g] * gs + c], element / key of group g), c < gs, g < G
    // groups.sum_out (level_sum_pays): k_k3 adds every group's ciphertext into the sum itself; chunks are then whole groups
    // (returns false where it could not: the scratch arenas hold less than one group per launch -- the caller then sums the groups itself)
    bool apply_galois_grouped(int L, u64 G, u64 gs, const u64 *in, KsGroups groups, u64 *out)
    {
        const KernelEnv env = ntt_env();
        const size_t N = P.N, LN = (size_t)L * N;
        const u64 n = G * gs;
        Indexer ix{};
        size_t chunk = chunk_ops(n, L, false);
        if (groups.sum_out) {
            if (chunk < gs || !fuse_pays(env, gs, L)) groups.sum_out = nullptr;
            else chunk -= chunk % gs;
        }
        // A level sum is a read-modify-write of groups.sum_out WITHOUT atomics.  It is exact because (1) the launch shape gives one wave sole
        // ownership of a (ciphertext, polynomial, tile, row) of the sum across all groups of the launch (launch_k3 checks the whole-group
        // shape), and (2) every chunk of the level is queued on this ONE stream, in order: never alternate these chunks over stream2_ the way
        // multiply_relin does, and never give grouped launches the four-wave or dual shapes.
        if (groups.sum_out && env.stream != stream_) throw std::logic_error("level sum: the chunks of a level must stay on one stream");
        for (u64 off = 0; off < n; off += chunk) {
            const u64 nc = std::min<u64>(chunk, n - off);
            Scratch S = scratch(std::min<u64>(chunk, n), L);
            KsBuffers B = S.ks;
            B.c01 = out + off * 2 * LN; B.c01_item_stride = 2 * LN;
            TensorOperands ten;
            ten.c1_mode = 4; // polynomial 1 of the rotated ciphertext is zero: the fused k_k3 starts it from there, and gathers the permuted c0 from `in` ...
            ten.gsrc = in; ten.gsrc_op_offset = 0; // (grouped: the op's group names its source block; g_op_offset carries the chunk offset)
            const bool fused = fuse_pays(env, nc, L); // ... (small grids take the unfused sequence: k_k1 writes the zero polynomial, k_floor_rows adds into it)
            if (groups.sum_out && !fused) throw std::logic_error("level sum: the fused key switch only"); // (fuse_pays grows with the chunk)
            launch_k1(env, L, K1_GALOIS, nc, off, in, nullptr, ix, nullptr, B, nullptr, false, fused, &groups, fused);
            key_switch_tail(env, L, nc, S, B, nullptr, false, nullptr, nullptr, fused ? &ten : nullptr, &groups, off);
        }
        return groups.sum_out != nullptr;
    }
    u64 rotate_sum(int L, u64 n, const u64 *in, const int *steps, u64 n_steps, u64 *out)
    {
        use();
        check_level(L);
        const size_t per = 2 * (size_t)L * P.N, bytes = n * per * 8;
        if (ranges_overlap(in, n * per, out, n * per)) throw std::invalid_argument("rotate_sum cannot run in place: `out` overlaps `in`");
        size_t depth = 0;
        std::vector<RotNode> trie = rotation_trie(steps, n_steps, depth);
        if (!n) return 0;
        const bool bfs_on = level_walk_;
        const KernelEnv nenv = ntt_env();
        // levels of the trie; a node's position inside its level is its group index
        std::vector<std::vector<int>> levels(depth + 1);
        size_t widest = 0;
        for (size_t id = 0; id < trie.size(); ++id) {
            trie[id].pos = (int)levels[(size_t)trie[id].level].size();
            levels[(size_t)trie[id].level].push_back((int)id);
            widest = std::max(widest, levels[(size_t)trie[id].level].size());
        }
        // node by node where that is the better shape: no fused path; CKKS batches so small that even the widest level stays within the
        // latency shape (digit-split k_k3: he355_set_latency_max) -- a BFV context has no latency shape, its levels always go grouped
        if (!bfs_on || trie.size() == 1 || !k3_can_fuse(nenv) || (P.scheme == kSchemeCKKS && latency_shape(n * widest)) || n > 0xFFFFFFFFull / trie.size())
            return rotate_sum_by_node(L, n, in, trie, depth, out);
        require_keyswitch();
        const bool bfv = P.scheme == kSchemeBFV;
        PolyView pv{};
        pv.polys_per_item = 2 * L; pv.item_stride = per;
        for (int p2 = 0; p2 < 2 * L; ++p2) pv.prime_of[p2] = (unsigned char)(p2 % L);
        // every level's keys before anything is allocated: a missing key must not leave blocks behind
        for (size_t lv = 1; lv <= depth; ++lv)
            for (int id : levels[lv])
                if (!galois_key(trie[(size_t)id].elt)) throw std::invalid_argument("Galois key not present");
        // the one block this walk holds at a time (the previous level's ciphertexts): back to the pool however the walk ends
        struct Held {
            DevicePool &pool;
            u64 *p = nullptr;
            ~Held() { if (p) pool.release(p); }
            void reset(u64 *q) { if (p) pool.release(p); p = q; } // (stream-ordered reuse: whatever takes the block next is queued behind these kernels)
        } held{pool_};
        // level 0: the input in NTT form (BFV: a transformed copy), the running sum starts as (1 + steps of 0) x input
        const u64 *src = in;
        if (bfv) {
            held.reset(static_cast<u64 *>(pool_.alloc(bytes)));
            HIPCHECK(hipMemcpyAsync(held.p, in, bytes, hipMemcpyDeviceToDevice, stream_));
            pv.base = held.p;
            launch_ntt_forward(env_, pv, (u32)n);
            src = held.p;
        }
        HIPCHECK(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToDevice, stream_));
        Indexer ixp{};
        ixp.pairwise = 1;
        for (u64 r = 0; r < trie[0].ends; ++r) addsub(L, 2, n, out, src, ixp, out, false);
        u64 switches = 0;
        for (size_t lv = 1; lv <= depth; ++lv) {
            const std::vector<int> &nodes = levels[lv];
            const size_t G = nodes.size();
            if (!G) break;
            std::vector<const uint32_t *> perms(G);
            std::vector<const u64 *> keys(G);
            std::vector<u32> src_block(G), mult(G);
            // The level's sum inside k_k3 (KsGroups::sum_out) where its grid shape fills the chip -- one block per (tile, eight ciphertexts),
            // each walking the level's groups -- and every chunk takes the fused path: the groups' ciphertexts that nothing starts from are
            // then never written, and k_sum_groups' pass over all of them (an HBM stream of its own, 6 % of configs[4]) is gone.
            const bool in_k3 = level_sum_pays(nenv, L, n) && n_steps < kGroupKeepBit;
            for (size_t g = 0; g < G; ++g) {
                const RotNode &nd = trie[(size_t)nodes[g]];
                perms[g] = perm(nd.elt);
                keys[g] = galois_key(nd.elt);
                src_block[g] = (u32)trie[(size_t)nd.parent].pos;
                mult[g] = (u32)nd.ends;
                if (in_k3 && !nd.kids.empty()) mult[g] |= kGroupKeepBit;
            }
            GroupTables gt = upload_groups(perms, keys, src_block, mult, (u32)n);
            Held cur{pool_};
            cur.reset(static_cast<u64 *>(pool_.alloc(G * bytes)));
            if (in_k3) { gt.g.sum_out = out; gt.g.count = gt.d_mult; }
            if (apply_galois_grouped(L, G, n, src, gt.g, cur.p)) ++paths_.level_sums_in_k3;
            else { ++paths_.level_sums_by_kernel; launch_sum_groups(env_, L, n, (u32)G, cur.p, gt.d_mult, out); }
            held.reset(cur.p);
            cur.p = nullptr;
            src = held.p;
            switches += G;
        }
        held.reset(nullptr);
        if (bfv) {
            pv.base = out;
            launch_ntt_inverse(env_, pv, (u32)n);
        }
        HIPCHECK(hipGetLastError());
        return switches;
    }
    // the node-by-node walk (depth first, one key-switch sequence per node): the latency shape, the unfused sequence
    u64 rotate_sum_by_node(int L, u64 n, const u64 *in, const std::vector<RotNode> &trie, size_t depth, u64 *out)
    {
        const size_t per = 2 * (size_t)L * P.N, bytes = n * per * 8;
        HIPCHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, stream_));
        Indexer ixp{};
        ixp.pairwise = 1;
        for (u64 r = 0; r < trie[0].ends; ++r) addsub(L, 2, n, out, in, ixp, out, false); // steps of 0: the input once more
        if (trie.size() == 1) return 0;
        require_keyswitch();
        // one ciphertext slab per trie level (a node's ciphertext lives until its last child is done; a leaf needs one only when
        // several steps end there)
        const size_t levels = depth;
        if (levels * bytes > rot_tmp_bytes_) {
            HIPCHECK(hipStreamSynchronize(stream_));
            pool_.raw_free(rot_tmp_);
            rot_tmp_ = nullptr; rot_tmp_bytes_ = 0;
            dmalloc(rot_tmp_, levels * bytes);
            rot_tmp_bytes_ = levels * bytes;
        }
        u64 switches = 0;
        // depth-first: (node, level of the node = number of terms applied)
        std::vector<std::pair<int, size_t>> stack;
        for (auto it = trie[0].kids.rbegin(); it != trie[0].kids.rend(); ++it) stack.push_back({*it, 1});
        while (!stack.empty()) {
            const auto [id, lvl] = stack.back();
            stack.pop_back();
            const RotNode &nd = trie[(size_t)id];
            const u64 *src = lvl == 1 ? in : rot_tmp_ + (lvl - 2) * n * per;
            const uint32_t e = nd.elt;
            if (nd.kids.empty() && nd.ends == 1) {
                apply_galois(L, n, src, e, out, out); // a leaf: the add_inplace rides the Galois step (sum += rotate(parent))
            } else {
                u64 *mine = rot_tmp_ + (lvl - 1) * n * per;
                apply_galois(L, n, src, e, mine);
                for (u64 r = 0; r < nd.ends; ++r) addsub(L, 2, n, out, mine, ixp, out, false);
                for (auto it = nd.kids.rbegin(); it != nd.kids.rend(); ++it) stack.push_back({*it, lvl + 1});
            }
            ++switches;
        }
        HIPCHECK(hipGetLastError());
        return switches;
    }
    // out[i] = rotate(in[i], steps[i]): the rotate_vector(dot_i, -i) loop of collapseCKKS (seal_context.cpp:389-392).  Every
    // ciphertext goes through its own NAF terms in its own order; ciphertexts whose t-th term is the same Galois element are
    // gathered and key-switched as one batch (a loop of single-ciphertext rotations is latency-bound: ~1 ms each at N=2^14).
    void rotate_each(int L, u64 n, const u64 *in, const int *steps, u64 *out)
    {
        use();
        check_level(L);
        if (!n) return;
        const size_t per = 2 * (size_t)L * P.N, bytes = n * per * 8;
        if (ranges_overlap(in, n * per, out, n * per)) throw std::invalid_argument("rotate_each cannot run in place: `out` overlaps `in`");
        std::vector<std::vector<int>> terms(n);
        size_t depth = 0;
        for (u64 i = 0; i < n; ++i) {
            terms[i] = rotation_terms(steps[i]);
            depth = std::max(depth, terms[i].size());
        }
        HIPCHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, stream_));
        if (!depth) return;
        require_keyswitch();
        if (2 * bytes > rot_tmp_bytes_) {
            HIPCHECK(hipStreamSynchronize(stream_));
            pool_.raw_free(rot_tmp_);
            rot_tmp_ = nullptr; rot_tmp_bytes_ = 0;
            dmalloc(rot_tmp_, 2 * bytes);
            rot_tmp_bytes_ = 2 * bytes;
        }
        u64 *ga = rot_tmp_, *gb = rot_tmp_ + n * per;
        for (size_t t = 0; t < depth; ++t) {
            std::map<uint32_t, std::vector<uint32_t>> groups; // Galois element -> ciphertexts whose t-th term it is
            u64 m_all = 0;
            for (u64 i = 0; i < n; ++i)
                if (t < terms[i].size()) { groups[P.galois_elt_from_step(terms[i][t])].push_back((uint32_t)i); ++m_all; }
            if (P.scheme == kSchemeCKKS && level_walk_ && k3_can_fuse(env_) && !latency_shape(m_all) && groups.size() > 1) {
                // ONE grouped key-switch sequence for every ciphertext that has a t-th term (groups of one op, each with its own Galois
                // element and key, ordered by element so that neighbouring waves share key rows): the ciphertexts are read where they lie
                // in `out` (KsGroups::src_block), the results land compactly in gb and are scattered back.  A loop over the elements
                // issues one sequence per element over 1-14 ciphertexts each (logreg .cpp's collapse: 35 sequences for 100 samples).
                std::vector<const uint32_t *> perms;
                std::vector<const u64 *> keys;
                std::vector<u32> src_block, mult, order;
                for (const auto &g : groups)
                    for (uint32_t i : g.second) {
                        perms.push_back(perm(g.first));
                        keys.push_back(galois_key(g.first));
                        if (!keys.back()) throw std::invalid_argument("Galois key not present");
                        src_block.push_back(i);
                        mult.push_back(0);
                        order.push_back(i);
                    }
                const GroupTables gt = upload_groups(perms, keys, src_block, mult, 1);
                apply_galois_grouped(L, m_all, 1, out, gt.g, gb);
                launch_move_cts(env_, out, gb, order.data(), m_all, per, true);
                continue;
            }
            for (const auto &g : groups) {
                const u64 m = g.second.size();
                launch_move_cts(env_, ga, out, g.second.data(), m, per, false);
                apply_galois(L, m, ga, g.first, gb);
                launch_move_cts(env_, out, gb, g.second.data(), m, per, true);
            }
        }
        HIPCHECK(hipGetLastError());
    }
    // The stream the context's own encryptions of zero draw from (accumulate with count 0): seeded from the OS at construction,
    // he355_set_zero_stream pins it (tests).
    void set_zero_stream(u64 seed, u64 first_index) { zero_seed_ = seed; zero_index_ = first_index; }
    // SEALContextWrapper::accumulateCKKS / accumulateBFV (seal_context.cpp:321-347, 289-319)
    void accumulate(int L, u64 n, u64 *inout, u64 count, u64 *tmp)
    {
        if (count == 0) {
            // the reference's else-branch (seal_context.cpp:312-316, 341-344): encryptor()->encrypt_zero(retval) -- a FRESH encryption
            // of zero at the first data level replaces every ciphertext.  SEAL returns it at the top level whatever the level of
            // the input was; a slab of lower-level ciphertexts cannot hold that, so only L == Ltop is accepted here.
            check_level(L);
            if ((size_t)L != P.Ltop) throw std::invalid_argument("accumulate with count 0 returns top-level encryptions of zero: the slab must be at the top level");
            encrypt(n, nullptr, zero_seed_, zero_index_, inout);
            zero_index_ += n;
            return;
        }
        if (P.scheme == kSchemeBFV) { // SEALContextWrapper::accumulateBFV (seal_context.cpp:289-319)
            const u64 half = P.N / 2;
            const u64 row_count = count > half ? half : count;
            int rot = 64 - __builtin_clzll(row_count);
            if (((u64)1 << (rot - 1)) == row_count) --rot;
            // The chain of rotate_rows + add_inplace (and the column swap) runs in the NTT domain on the fused pipeline the CKKS path uses
            // (Galois permutation in k_k1, mod-down inside k_k3, the latency shape for small batches): the ciphertexts are transformed
            // once on the way in and once on the way out, every step in between is the same exact map on residues in either
            // representation (see rotate_sum).  Two or more steps pay for the two transforms (20 transforms per key switch at L = 3).
            // Measured through the bridge (profiles/r04_bridge_phases.jsonl): it wins where the fused pipeline has a shape of its own -- the
            // latency shape (1 ciphertext at N = 2^14: 2.02 -> 1.59 ms) -- and in the throughput regime (1024 ciphertexts at N = 2^14:
            // 44.1 -> 38.9 ms); in between (64-80 ciphertexts at N <= 2^14) the BFV kernels' launches fill the chip better: 1.47 -> 1.77 ms.
            const int n_steps = rot + (count > half ? 1 : 0);
            const bool ntt_chain = level_walk_ && n_steps >= 2 && k3_can_fuse(ntt_env()) && (n <= lat_limit() || n * P.N >= ((u64)1 << 23));
            PolyView pv{};
            pv.base = inout; pv.polys_per_item = 2 * L; pv.item_stride = 2 * (u64)L * P.N;
            for (int p2 = 0; p2 < 2 * L; ++p2) pv.prime_of[p2] = (unsigned char)(p2 % L);
            if (ntt_chain) launch_ntt_forward(env_, pv, (u32)n);
            u64 *cur = inout, *nxt = tmp; // ping-pong: nxt = cur + rotate(cur), one pipeline per step and no separate add
            for (int i = 0; i < rot; ++i) {
                rotate(L, n, cur, 1 << i, nxt, cur, ntt_chain); // rotate_rows + add_inplace
                std::swap(cur, nxt);
            }
            if (count > half) {
                apply_galois(L, n, cur, (uint32_t)(2 * P.N - 1), nxt, cur, ntt_chain); // rotate_columns + add_inplace
                std::swap(cur, nxt);
            }
            if (cur != inout) HIPCHECK(hipMemcpyAsync(inout, cur, n * 2 * (size_t)L * P.N * 8, hipMemcpyDeviceToDevice, stream_));
            if (ntt_chain) launch_ntt_inverse(env_, pv, (u32)n);
            return;
        }
        const u64 slots = P.N / 2;
        if (count > slots) count = slots;
        int rotations = 64 - __builtin_clzll(count);
        if (((u64)1 << (rotations - 1)) == count) --rotations;
        Indexer ix{};
        ix.pairwise = 1;
        (void)ix;
        u64 *cur = inout, *nxt = tmp;
        for (int i = 0; i < rotations; ++i) {
            rotate(L, n, cur, 1 << i, nxt, cur); // rotate_vector + add_inplace in one pipeline
            std::swap(cur, nxt);
        }
        if (cur != inout) HIPCHECK(hipMemcpyAsync(inout, cur, n * 2 * (size_t)L * P.N * 8, hipMemcpyDeviceToDevice, stream_));
    }

    // ---- BFV ------------------------------------------------------------------------------------------
    const BehzDev &behz(int L)
    {
        auto it = behz_.find(L);
        if (it != behz_.end()) return it->second;
        const BehzHost H = P.behz_host(L); // the folded constants (he_params.cpp), as two flat arrays
        u64 *d = nullptr;
        dmalloc(d, H.words.size() * 8);
        owned_.push_back(d);
        HIPCHECK(hipMemcpy(d, H.words.data(), H.words.size() * 8, hipMemcpyHostToDevice));
        double *ddev = nullptr;
        dmalloc(ddev, H.doubles.size() * 8);
        owned_.push_back(ddev);
        HIPCHECK(hipMemcpy(ddev, H.doubles.data(), H.doubles.size() * 8, hipMemcpyHostToDevice));
        const BehzDev Z = H.view(d, ddev, P.K);
        return behz_[L] = Z;
    }
    // out(i, j) = sum_k relinearize(multiply(a(i, k), b(k, j))): the multiply / relinearize_inplace / add_inplace loop of the BFV
    // CipherBatchAxis matrix product (bfv cipherbatchaxis .cpp:398-410) with the inner index as part of the batch -- one multiply and
    // one relinearization over rows * cols * k ciphertext pairs, then the sums over k (modular additions: any order, same residues).
    void bfv_multiply_relin_accumulate(int L, u64 rows, u64 cols, u64 inner, const u64 *a, u64 a_stride_i, u64 a_stride_k, const u64 *b,
                                       u64 b_stride_k, u64 b_stride_j, u64 *out)
    {
        use();
        check_level(L);
        if (P.scheme != kSchemeBFV) throw std::invalid_argument("he355_bfv_multiply_relin_accumulate needs a BFV context");
        if (inner < 1 || inner > 0x7fffffff) throw std::invalid_argument("inner dimension out of range");
        const u64 n = rows * cols;
        if (!n) return;
        const size_t LN = (size_t)L * P.N;
        // `out` is written after each pass over the inner index and the operands are read again by the next one
        const size_t a_cts = (size_t)((inner - 1) * a_stride_k + (rows - 1) * a_stride_i + 1), b_cts = (size_t)((inner - 1) * b_stride_k + (cols - 1) * b_stride_j + 1);
        if (ranges_overlap(out, n * 2 * LN, a, a_cts * 2 * LN) || ranges_overlap(out, n * 2 * LN, b, b_cts * 2 * LN))
            throw std::invalid_argument("he355_bfv_multiply_relin_accumulate: `out` overlaps an operand");
        const u64 kc = std::max<u64>(1, std::min<u64>(inner, (u64)4096 / n)); // inner indices per pass: about 4096 products in flight
        u64 *c3 = static_cast<u64 *>(pool_alloc(n * kc * 3 * LN * 8));
        u64 *r2 = nullptr;
        try {
            r2 = static_cast<u64 *>(pool_alloc(n * kc * 2 * LN * 8));
            for (u64 k0 = 0; k0 < inner; k0 += kc) {
                const u64 kn = std::min<u64>(kc, inner - k0);
                Indexer3 ix{};
                ix.a_base = k0 * a_stride_k; ix.b_base = k0 * b_stride_k;
                ix.gs = n; ix.b1 = cols; ix.a_sg = a_stride_k; ix.a_si = a_stride_i; ix.b_sg = b_stride_k; ix.b_sj = b_stride_j;
                bfv_multiply3(L, n * kn, a, b, ix, c3);
                relinearize(L, n * kn, c3, r2);
                launch_sum_cts(env_, L, 2, kn, r2, out, n, k0 != 0);
            }
            HIPCHECK(hipGetLastError());
        } catch (...) {
            pool_free(c3);
            if (r2) pool_free(r2);
            throw;
        }
        pool_free(c3);
        pool_free(r2);
    }
    // Evaluator::bfv_multiply (BEHZ), size 2 x 2 -> 3, coefficient form
    void bfv_multiply(int L, u64 n, const u64 *a, const u64 *b, Indexer ix, u64 *out) { bfv_multiply3(L, n, a, b, to_ix3(ix), out); }
    // the BFV multiply's scratch arena holds at least `bytes` afterwards, or false (not enough device memory: the caller halves its chunk)
    bool reserve_bfv_scratch(size_t bytes)
    {
        if (bytes <= bfv_bytes_) return true;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)bytes > 0.9 * (double)(free_b + bfv_bytes_ + pool_.cached_bytes())) return false;
        try {
            HIPCHECK(hipStreamSynchronize(stream_));
            pool_.raw_free(bfv_scratch_);
            bfv_scratch_ = nullptr; bfv_bytes_ = 0;
            dmalloc(bfv_scratch_, bytes);
            bfv_bytes_ = bytes;
            return true;
        } catch (const OutOfDeviceMemory &) {
            return false;
        }
    }
    void bfv_multiply3(int L, u64 n, const u64 *a, const u64 *b, Indexer3 ix, u64 *out)
    {
        use();
        check_level(L);
        if (P.scheme != kSchemeBFV) throw std::invalid_argument("he355_bfv_multiply needs a BFV context");
        if (!n) return;
        const BehzDev &Z = behz(L);
        const size_t N = P.N, S = (size_t)Z.nB + 1;
        PolyView vq{}, vb{};
        for (int i = 0; i < L; ++i) vq.prime_of[i] = (unsigned char)i;
        for (size_t j = 0; j < S; ++j) vb.prime_of[j] = Z.bsk_prime[j];
        vq.polys_per_item = L; vq.item_stride = (u64)L * N;
        vb.polys_per_item = (int)S; vb.item_stride = (u64)S * N;
        const bool fuse_cols = behz_cols_fusable(env_, Z);
        BehzSrc src{};
        src.a = a; src.b = b; src.ix = ix;
        // Distinct operands: result r = (g, i, j) reads a(g, i) and b(g, j).  Where every operand serves several results (outer products,
        // the terms of a matrix product) each is extended to Bsk and transformed ONCE (steps (1)-(3) per operand instead of per result:
        // SEAL's multiply recomputes them for every pair, the values are the same), and a result costs its dyadic tensor, three inverse
        // transforms and steps (6)-(8).
        const bool hoist_on = (behz_fuse_mask() & 2) != 0;
        const u64 gsz = std::min<u64>(ix.gs, n), G = ix.gs >= n ? 1 : (n + ix.gs - 1) / ix.gs;
        src.I = (gsz + ix.b1 - 1) / ix.b1; src.J = std::min<u64>(ix.b1, gsz); src.na = G * src.I;
        const u64 n_cts = src.na + G * src.J;
        const size_t e_words = (size_t)n_cts * 2 * (L + S) * N, per_res = (3 * L + 3 * S) * N;
        {   // a chunk's results are written before the next chunk's operands are read: `out` needs a slab of its own
            const size_t a_cts = (size_t)(ix.a_base + (G - 1) * ix.a_sg + (src.I - 1) * ix.a_si + 1), b_cts = (size_t)(ix.b_base + (G - 1) * ix.b_sg + (src.J - 1) * ix.b_sj + 1);
            if (ranges_overlap(out, (size_t)n * 3 * L * N, a, a_cts * 2 * L * N) || ranges_overlap(out, (size_t)n * 3 * L * N, b, b_cts * 2 * L * N))
                throw std::invalid_argument("he355_bfv_multiply: `out` overlaps an operand");
        }
        bool lists = hoist_on && (G == 1 || (n % ix.gs == 0 && ix.gs % ix.b1 == 0)) && n_cts <= n; // at least two times fewer extensions than the 2 n of the per-pair path
        size_t c = std::min<size_t>(chunk_, (size_t)n);
        if (lists) {
            while (!reserve_bfv_scratch((e_words + per_res * c) * 8) && c > 1) c = (c + 1) / 2;
            lists = (e_words + per_res * c) * 8 <= bfv_bytes_; // else: the operand set does not fit beside one result -- per-pair path below
        }
        if (lists) {
            src.lists = 1;
            u64 *eq = bfv_scratch_, *eb = eq + (size_t)n_cts * 2 * L * N, *dq = eb + (size_t)n_cts * 2 * S * N, *ds = dq + c * 3 * L * N;
            if (fuse_cols) {
                launch_behz_extend_cols(env_, Z, src, n_cts, eq, eb);
            } else {
                launch_behz_extend(env_, Z, src, n_cts, eq, eb);
                vq.base = eq; launch_cols_fwd(env_, vq, (u32)(n_cts * 2));
                vb.base = eb; launch_cols_fwd(env_, vb, (u32)(n_cts * 2));
            }
            vq.base = eq; launch_rows_fwd(env_, vq, (u32)(n_cts * 2));
            vb.base = eb; launch_rows_fwd(env_, vb, (u32)(n_cts * 2));
            for (u64 off = 0; off < n; off += c) {
                const u64 nc = std::min<u64>(c, n - off);
                launch_behz_tensor_inv(env_, Z, src, nc, off, eq, eb, dq, ds);
                if (fuse_cols) {
                    launch_behz_cols_floor_sk(env_, Z, nc, dq, ds, out + off * 3 * (size_t)L * N);
                } else {
                    vq.base = dq; launch_cols_inv(env_, vq, (u32)(nc * 3));
                    vb.base = ds; launch_cols_inv(env_, vb, (u32)(nc * 3));
                    launch_behz_floor_sk(env_, Z, nc, dq, ds, out + off * 3 * (size_t)L * N);
                }
            }
            HIPCHECK(hipGetLastError());
            return;
        }
        const size_t per_op = (4 * L + 4 * S) * N + per_res;
        while (!reserve_bfv_scratch(per_op * c * 8)) { // as chunk_ops: halved until the arena is reserved
            if (c == 1) throw OutOfDeviceMemory("HIP error: out of device memory: the BFV multiply scratch of one ciphertext does not fit");
            c = (c + 1) / 2;
        }
        u64 *xq = bfv_scratch_, *xb = xq + c * 4 * L * N, *dq = xb + c * 4 * S * N, *ds = dq + c * 3 * L * N;
        for (u64 off = 0; off < n; off += c) {
            const u64 nc = std::min<u64>(c, n - off);
            // extension to Bsk and forward column passes (one kernel where the fused shape applies), then per (op, residue, row) ONE
            // kernel for the forward row pass of the four polynomials, the dyadic tensor and the inverse row pass of the three
            // products (k_behz_rows_tensor), then the inverse column passes and steps (6)-(8) (again one kernel where it applies)
            src.op_offset = off;
            if (fuse_cols) {
                launch_behz_extend_cols(env_, Z, src, nc * 2, xq, xb);
            } else {
                launch_behz_extend(env_, Z, src, nc * 2, xq, xb);
                vq.base = xq; launch_cols_fwd(env_, vq, (u32)(nc * 4));
                vb.base = xb; launch_cols_fwd(env_, vb, (u32)(nc * 4));
            }
            launch_behz_rows_tensor(env_, Z, nc, xq, xb, dq, ds);
            if (fuse_cols) {
                launch_behz_cols_floor_sk(env_, Z, nc, dq, ds, out + off * 3 * (size_t)L * N);
            } else {
                vq.base = dq; launch_cols_inv(env_, vq, (u32)(nc * 3));
                vb.base = ds; launch_cols_inv(env_, vb, (u32)(nc * 3));
                launch_behz_floor_sk(env_, Z, nc, dq, ds, out + off * 3 * (size_t)L * N);
            }
        }
        HIPCHECK(hipGetLastError());
    }
    // key switching for BFV: the target is in coefficient form; result added into c01 (coefficient form)
    void bfv_key_switch(int L, u64 nc, const Scratch &S, const KsBuffers &B, const u64 *key, const u64 *target, u64 target_op_stride,
                        const u64 *add01 = nullptr, u64 add01_item_stride = 0)
    {
        launch_k2(env_, L, nc, B, target, target_op_stride);
        launch_k3(env_, L, nc, B, key);
        launch_bfv_tail_sp(env_, nc * 2, B.tpr, S.rlr);
        launch_bfv_tail_fin(env_, L, nc, B.t, S.rlr, B.c01, B.c01_item_stride, add01, add01_item_stride);
    }
    // ---- client side on the device -----------------------------------------------------
    void set_public_key(const u64 *h_pk) // [2][K][N], NTT form
    {
        use();
        const size_t bytes = 2 * P.K * P.N * 8;
        if (!d_pk_) { dmalloc(d_pk_, bytes); owned_.push_back(d_pk_); }
        HIPCHECK(hipMemcpy(d_pk_, h_pk, bytes, hipMemcpyHostToDevice));
    }
    void set_secret_key(const u64 *h_sk) // [K][N], NTT form
    {
        use();
        const size_t bytes = P.K * P.N * 8;
        if (!d_sk_) { dmalloc(d_sk_, bytes); owned_.push_back(d_sk_); }
        HIPCHECK(hipMemcpy(d_sk_, h_sk, bytes, hipMemcpyHostToDevice));
    }
    u64 *client_scratch(size_t elems)
    {
        if (elems > client_scratch_elems_) {
            HIPCHECK(hipStreamSynchronize(stream_));
            pool_.raw_free(client_scratch_);
            client_scratch_ = nullptr;
            dmalloc(client_scratch_, elems * 8);
            client_scratch_elems_ = elems;
        }
        return client_scratch_;
    }
    static PolyView poly_view(u64 *base, int polys_per_item, size_t N, int period)
    {
        if (polys_per_item > 64) throw std::invalid_argument("too many polynomials per item");
        PolyView v;
        v.base = base; v.item_stride = (u64)polys_per_item * N; v.polys_per_item = polys_per_item; v.pad_ = 0;
        for (int i = 0; i < polys_per_item; ++i) v.prime_of[i] = (unsigned char)(i % period);
        return v;
    }
    // Encryptor::encrypt (asymmetric) of n plaintexts: CKKS plain [n][Ltop][N] NTT form, BFV plain [n][N] mod t;
    // out [n][2][Ltop][N].  Ciphertext r uses the counter-based streams of index first_index + r (client/sampler.h).
    // plain == nullptr: Encryptor::encrypt_zero (the plaintext term is skipped: adding the zero plaintext changes nothing).
    void encrypt(u64 n, const u64 *plain, u64 seed, u64 first_index, u64 *out)
    {
        use();
        if (!d_pk_) throw std::invalid_argument("public key not set");
        const size_t N = P.N, K = P.K, L = P.Ltop;
        const bool ckks = P.scheme == kSchemeCKKS;
        const u64 cmax = 32;
        // per ciphertext: u K, e 2K, z 2K (BFV), tail 2, cols 2L polys
        u64 *base = client_scratch(cmax * (K + 2 * K + 2 * K + 2 + 2 * L) * N);
        u64 *u = base, *e = u + cmax * K * N, *z = e + cmax * 2 * K * N, *tpr = z + cmax * 2 * K * N, *cols = tpr + cmax * 2 * N;
        u64 qdivt[kMaxPrimes] = {0}, q_mod_t = 1; // L <= K <= kMaxPrimes (constructor)
        if (!ckks) {
            const u64 t = P.plain_modulus;
            for (size_t i = 0; i < L; ++i) q_mod_t = (u64)(((u128)q_mod_t * (P.primes[i].q % t)) % t);
            for (size_t i = 0; i < L; ++i) {
                const u64 qi = P.primes[i].q, tinv = Params::invmod(t % qi, qi), neg = (q_mod_t % qi) ? qi - q_mod_t % qi : 0;
                qdivt[i] = (u64)(((u128)neg * tinv) % qi); // floor(q/t) mod q_i = -(q mod t) * t^-1
            }
        }
        for (u64 off = 0; off < n; off += cmax) {
            const u64 c = std::min<u64>(cmax, n - off);
            u64 *o = out + off * 2 * L * N;
            launch_enc_sample(env_, c, seed, first_index + off, u, e);
            launch_ntt_forward(env_, poly_view(u, (int)K, N, (int)K), (u32)c);
            if (ckks) {
                launch_ntt_forward(env_, poly_view(e, (int)(2 * K), N, (int)K), (u32)c);
                launch_enc_mul_pk(env_, c, u, d_pk_, e, true); // z := u*pk + NTT(e), in the e buffer
                if (K > 1) {
                    const int SP = (int)K - 1;
                    launch_rows_inv_select(env_, SP, c * 2, e + (size_t)SP * N, (u64)K * N, tpr);
                    launch_floor_cols(env_, SP, (int)L, c * 2, tpr, cols);
                    FloorRowsArgs fr;
                    fr.src_prime = SP; fr.n_tgt = (int)L; fr.n_src = 2;
                    fr.cols = cols;
                    fr.tsrc = e; fr.tsrc_op_stride = 2 * K * N; fr.tsrc_poly_stride = K * N;
                    fr.addend = nullptr; fr.add_op_stride = 0; fr.add_poly_stride = 0;
                    fr.out = o; fr.out_op_stride = 2 * L * N; fr.out_poly_stride = L * N;
                    fr.tail_prime = -1; fr.tail = nullptr;
                    launch_floor_rows(env_, c, fr);
                } else {
                    HIPCHECK(hipMemcpyAsync(o, e, c * 2 * N * 8, hipMemcpyDeviceToDevice, stream_));
                }
                Indexer pw{};
                pw.b1 = 1; pw.pairwise = 1;
                if (plain) launch_plain_op(env_, (int)L, 2, c, o, plain + off * L * N, pw, o, 1); // c0 += plain
            } else {
                launch_enc_mul_pk(env_, c, u, d_pk_, z, false);
                launch_ntt_inverse(env_, poly_view(z, (int)(2 * K), N, (int)K), (u32)c);
                Indexer pw{};
                pw.b1 = 1; pw.pairwise = 1;
                launch_addsub(env_, (int)K, 2, c, z, e, pw, z, false);
                if (K > 1) launch_divround_last_coeff(env_, c * 2, z, o);
                else HIPCHECK(hipMemcpyAsync(o, z, c * 2 * N * 8, hipMemcpyDeviceToDevice, stream_));
                if (plain) launch_bfv_add_scaled_plain(env_, (int)L, c, o, plain + off * N, P.plain_modulus, q_mod_t, qdivt);
            }
        }
        HIPCHECK(hipGetLastError());
    }
    const CrtTablesDev &crt_tables(int L)
    {
        auto it = crt_.find(L);
        if (it != crt_.end()) return it->second;
        const int words = L + 2;
        std::vector<u64> Q(words, 0), halfQ(words, 0), punct((size_t)L * words, 0), inv(L);
        Q[0] = 1;
        for (int i = 0; i < L; ++i) client::mw_mul_small(Q.data(), words, P.primes[i].q);
        for (int i = 0; i < words; ++i) halfQ[i] = (Q[i] >> 1) | (i + 1 < words ? Q[i + 1] << 63 : 0);
        for (int i = 0; i < L; ++i) {
            u64 *p = punct.data() + (size_t)i * words;
            p[0] = 1;
            u64 pm = 1;
            const u64 qi = P.primes[i].q;
            for (int k = 0; k < L; ++k)
                if (k != i) {
                    client::mw_mul_small(p, words, P.primes[k].q);
                    pm = (u64)(((u128)pm * (P.primes[k].q % qi)) % qi);
                }
            inv[i] = Params::invmod(pm, qi);
        }
        u64 *d = nullptr;
        const size_t tot = (size_t)words * 2 + (size_t)L * words + L;
        dmalloc(d, tot * 8);
        owned_.push_back(d);
        HIPCHECK(hipMemcpy(d, Q.data(), words * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(d + words, halfQ.data(), words * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(d + 2 * words, punct.data(), punct.size() * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(d + 2 * words + punct.size(), inv.data(), L * 8, hipMemcpyHostToDevice));
        CrtTablesDev t;
        t.L = L; t.words = words; t.Q = d; t.halfQ = d + words; t.punct = d + 2 * words; t.inv = d + 2 * words + punct.size();
        t.Qd = client::mw_to_double(Q.data(), words); t.t = P.plain_modulus;
        return crt_[L] = t;
    }
    // Decryptor::decrypt of n size-`size` ciphertexts at level L: CKKS -> [n][L][N] NTT-form plaintext (the phase);
    // BFV -> [n][N] coefficients mod t
    void decrypt(int L, int size, u64 n, const u64 *ct, u64 *out)
    {
        use();
        check_level(L);
        if (!d_sk_) throw std::invalid_argument("secret key not set");
        if (size < 2 || size > 3) throw std::invalid_argument("ciphertext size must be 2 or 3");
        const size_t N = P.N;
        if (P.scheme == kSchemeCKKS) {
            launch_dot_sk(env_, L, size, n, ct, d_sk_, out);
            HIPCHECK(hipGetLastError());
            return;
        }
        if (L > 16) throw std::invalid_argument("BFV decryption supports up to 16 data primes");
        const CrtTablesDev &crt = crt_tables(L);
        const u64 cmax = 64;
        u64 *tmp = client_scratch(cmax * ((size_t)size * L + L) * N), *phase = tmp + cmax * (size_t)size * L * N;
        for (u64 off = 0; off < n; off += cmax) {
            const u64 c = std::min<u64>(cmax, n - off);
            HIPCHECK(hipMemcpyAsync(tmp, ct + off * size * L * N, c * size * L * N * 8, hipMemcpyDeviceToDevice, stream_));
            launch_ntt_forward(env_, poly_view(tmp, size * L, N, L), (u32)c);
            launch_dot_sk(env_, L, size, c, tmp, d_sk_, phase);
            launch_ntt_inverse(env_, poly_view(phase, L, N, L), (u32)c);
            launch_bfv_scale_round(env_, c, phase, out + off * N, crt);
        }
        HIPCHECK(hipGetLastError());
    }
    // ---- encoders (CKKSEncoder / BatchEncoder) -------------------------------------------------------------------
    const EncTablesDev &enc_tables()
    {
        if (enc_.slot_index) return enc_;
        std::vector<uint32_t> si;
        client::build_slot_index(P.N, si);
        uint32_t *dsi = nullptr;
        dmalloc(dsi, P.N * 4);
        owned_.push_back(dsi);
        HIPCHECK(hipMemcpy(dsi, si.data(), P.N * 4, hipMemcpyHostToDevice));
        enc_.slot_index = dsi;
        if (P.scheme == kSchemeCKKS) {
            std::vector<client::Cplx> W, Z;
            client::build_ckks_tables(P.N, W, Z);
            client::Cplx *dw = nullptr;
            dmalloc(dw, 2 * P.N * sizeof(client::Cplx));
            owned_.push_back(dw);
            HIPCHECK(hipMemcpy(dw, W.data(), P.N * sizeof(client::Cplx), hipMemcpyHostToDevice));
            HIPCHECK(hipMemcpy(dw + P.N, Z.data(), P.N * sizeof(client::Cplx), hipMemcpyHostToDevice));
            enc_.W = dw; enc_.Z = dw + P.N;
        }
        if (!d_err_) { dmalloc(d_err_, sizeof(int)); owned_.push_back(d_err_); }
        return enc_;
    }
    // CKKSEncoder::encode: values [n][count] (count <= N/2) at `scale` -> [n][Ltop][N] NTT-form plaintexts
    void ckks_encode(u64 n, const double *values, u64 count, double scale, u64 *plain)
    {
        use();
        if (P.scheme != kSchemeCKKS) throw std::invalid_argument("he355_ckks_encode needs a CKKS context");
        if (count > P.N / 2) throw std::invalid_argument("Not enough slots available to create packed plaintext");
        const EncTablesDev &T = enc_tables();
        const size_t N = P.N, L = P.Ltop;
        const u64 cmax = 256;
        u64 *zbuf = client_scratch(cmax * 2 * N);
        HIPCHECK(hipMemsetAsync(d_err_, 0, sizeof(int), stream_));
        for (u64 off = 0; off < n; off += cmax) {
            const u64 c = std::min<u64>(cmax, n - off);
            launch_ckks_encode(env_, c, values + off * count, count, scale, zbuf, plain + off * L * N, T, d_err_);
            launch_ntt_forward(env_, poly_view(plain + off * L * N, (int)L, N, (int)L), (u32)c);
        }
        int err = 0;
        HIPCHECK(hipMemcpyAsync(&err, d_err_, sizeof(int), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
        if (err) throw std::invalid_argument("encoded values are too large");
    }
    // the slot ranges of a decode call, checked against the encoder's slot count (ranges == null: every slot)
    SlotRanges slot_ranges(const u64 *ranges, u64 n_ranges, u64 slots) const
    {
        SlotRanges sr{};
        if (!ranges) { sr.n = 1; sr.first[0] = 0; sr.count[0] = sr.total = (u32)slots; return sr; }
        if (n_ranges < 1 || n_ranges > (u64)kMaxSlotRanges) throw std::invalid_argument("decode: 1 to 4 slot ranges");
        sr.n = (u32)n_ranges;
        for (u64 g = 0; g < n_ranges; ++g) {
            const u64 first = ranges[2 * g], count = ranges[2 * g + 1];
            if (first > slots || count > slots - first) throw std::invalid_argument("decode: slot range outside the encoder's slots");
            sr.first[g] = (u32)first; sr.count[g] = (u32)count; sr.total += (u32)count;
        }
        return sr;
    }
    // CKKSEncoder::decode: [n][L][N] NTT-form plaintexts -> [n][total] real slot values (ranges == null: all N/2)
    void ckks_decode(int L, u64 n, const u64 *plain, double scale, double *out, const u64 *ranges = nullptr, u64 n_ranges = 0)
    {
        use();
        check_level(L);
        if (P.scheme != kSchemeCKKS) throw std::invalid_argument("he355_ckks_decode needs a CKKS context");
        if (L > 16) throw std::invalid_argument("decoding supports up to 16 data primes");
        const SlotRanges sr = slot_ranges(ranges, n_ranges, P.N / 2);
        const EncTablesDev &T = enc_tables();
        const CrtTablesDev &crt = crt_tables(L);
        const size_t N = P.N;
        const u64 cmax = 128;
        u64 *coeff = client_scratch(cmax * ((size_t)L * N + 2 * N)), *zbuf = coeff + cmax * (size_t)L * N;
        for (u64 off = 0; off < n; off += cmax) {
            const u64 c = std::min<u64>(cmax, n - off);
            HIPCHECK(hipMemcpyAsync(coeff, plain + off * L * N, c * L * N * 8, hipMemcpyDeviceToDevice, stream_));
            launch_ntt_inverse(env_, poly_view(coeff, L, N, L), (u32)c);
            launch_ckks_decode(env_, c, coeff, scale, zbuf, out + off * sr.total, T, crt, sr);
        }
        HIPCHECK(hipGetLastError());
    }
    // BatchEncoder::encode / decode: [n][count] int64 <-> [n][N] coefficients mod t
    void bfv_encode(u64 n, const long long *values, u64 count, u64 *plain)
    {
        use();
        if (t_index_ < 0) throw std::invalid_argument("he355_bfv_encode needs a BFV context with a batching plain modulus");
        if (count > P.N) throw std::invalid_argument("Not enough slots available to create packed plaintext");
        const EncTablesDev &T = enc_tables();
        HIPCHECK(hipMemsetAsync(plain, 0, n * P.N * 8, stream_));
        launch_bfv_encode_scatter(env_, n, values, count, plain, T.slot_index, P.plain_modulus);
        PolyView v = poly_view(plain, 1, P.N, 1);
        v.prime_of[0] = (unsigned char)t_index_;
        launch_ntt_inverse(env_, v, (u32)n);
        HIPCHECK(hipGetLastError());
    }
    void bfv_decode(u64 n, const u64 *plain, long long *out, const u64 *ranges = nullptr, u64 n_ranges = 0)
    {
        use();
        if (t_index_ < 0) throw std::invalid_argument("he355_bfv_decode needs a BFV context with a batching plain modulus");
        const SlotRanges sr = slot_ranges(ranges, n_ranges, P.N);
        const EncTablesDev &T = enc_tables();
        const u64 cmax = 1024;
        u64 *ev = client_scratch(cmax * P.N);
        for (u64 off = 0; off < n; off += cmax) {
            const u64 c = std::min<u64>(cmax, n - off);
            HIPCHECK(hipMemcpyAsync(ev, plain + off * P.N, c * P.N * 8, hipMemcpyDeviceToDevice, stream_));
            PolyView v = poly_view(ev, 1, P.N, 1);
            v.prime_of[0] = (unsigned char)t_index_;
            launch_ntt_forward(env_, v, (u32)c);
            launch_bfv_decode_gather(env_, c, ev, out + off * sr.total, T.slot_index, P.plain_modulus, sr);
        }
        HIPCHECK(hipGetLastError());
    }
    void ntt(u64 *polys, u64 n_polys, const uint8_t *prime_of, u32 period, bool inverse)
    {
        use();
        if (period == 0 || period > 64) throw std::invalid_argument("prime map period must be in [1, 64]");
        if (n_polys % period) throw std::invalid_argument("polynomial count must be a multiple of the prime map period");
        PolyView v;
        v.base = polys; v.item_stride = (u64)period * P.N; v.polys_per_item = (int)period; v.pad_ = 0;
        for (u32 i = 0; i < period; ++i) {
            if (prime_of[i] >= P.K + P.aux.size()) throw std::invalid_argument("prime index out of range");
            v.prime_of[i] = prime_of[i];
        }
        if (inverse) launch_ntt_inverse(env_, v, (u32)(n_polys / period));
        else launch_ntt_forward(env_, v, (u32)(n_polys / period));
        HIPCHECK(hipGetLastError());
    }
    void timer_begin()
    {
        use();
        probe_.used = 0;
        probe_.ops = 0;
        env_.probe = &probe_; // probed until timer_end
        HIPCHECK(hipEventRecord(ev0_, stream_));
    }
    // dominant-kernel probe of the region closed by the last timer_end: total duration, launches, ops covered
    void probe_result(float *total_ms, u64 *launches, u64 *ops)
    {
        use();
        float tot = 0;
        for (int i = 0; i < probe_.used; ++i) {
            float ms = 0;
            HIPCHECK(hipEventSynchronize(probe_.stop[i]));
            HIPCHECK(hipEventElapsedTime(&ms, probe_.start[i], probe_.stop[i]));
            tot += ms;
        }
        if (total_ms) *total_ms = tot;
        if (launches) *launches = (u64)probe_.used;
        if (ops) *ops = probe_.ops;
    }
    float timer_end()
    {
        use();
        HIPCHECK(hipEventRecord(ev1_, stream_));
        HIPCHECK(hipEventSynchronize(ev1_));
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, ev0_, ev1_));
        env_.probe = nullptr;
        return ms;
    }
    void sync() { use(); HIPCHECK(hipStreamSynchronize(stream_)); HIPCHECK(hipStreamSynchronize(stream2_)); }
    // clock probe: started BEFORE the region it measures (the wave takes its slot first), bounded by `duration_us` of real time
    void clock_probe_begin(u64 duration_us)
    {
        use();
        if (duration_us == 0 || duration_us > 10000000) throw std::invalid_argument("clock probe duration must be in (0, 10 s]");
        if (!probe_stream_) HIPCHECK(hipStreamCreateWithFlags(&probe_stream_, hipStreamNonBlocking));
        if (!d_clock_) dmalloc(d_clock_, 2 * sizeof(u64));
        HIPCHECK(hipMemsetAsync(d_clock_, 0, 2 * sizeof(u64), probe_stream_));
        hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, probe_stream_, duration_us * 100, d_clock_);
        HIPCHECK(hipGetLastError());
    }
    void clock_probe_end(double *mhz, double *seconds)
    {
        use();
        if (!probe_stream_ || !d_clock_) throw std::logic_error("clock probe not started");
        u64 h[2] = {0, 0};
        HIPCHECK(hipMemcpyAsync(h, d_clock_, sizeof h, hipMemcpyDeviceToHost, probe_stream_));
        HIPCHECK(hipStreamSynchronize(probe_stream_));
        if (mhz) *mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
        if (seconds) *seconds = (double)h[1] / 1e8;
    }

private:
    KernelProbe probe_;
    const Params &P;
    int device_;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    KernelEnv env_{};
    PrimeDev *d_primes_ = nullptr;
    u64 *lat_part_[2] = {nullptr, nullptr}; // partial sums of the digit-split K3 (latency shape), one per stream
    size_t lat_part_bytes_[2] = {0, 0};
    u64 lat_max_ = 0;              // largest batch that takes the latency shape once set (HE355_LATENCY_MAX; 0: never) ...
    bool lat_auto_ = true;         // ... until then lat_limit()'s rule
    FloorConst *d_floor_ = nullptr;
    std::vector<void *> owned_;
    u64 *d_relin_ = nullptr;
    u64 *d_relin_scaled_ = nullptr; // relin_scaled()
    bool d_relin_scaled_ok_ = false;
    PrimeTables plain_tables_; // BFV: NTT tables mod t
    int t_index_ = -1;         // index of t in the device prime array (-1: none)
    EncTablesDev enc_{nullptr, nullptr, nullptr};
    int *d_err_ = nullptr;
    u64 *d_pk_ = nullptr, *d_sk_ = nullptr;
    u64 zero_seed_ = os_seed(), zero_index_ = 0;
    u64 *client_scratch_ = nullptr;
    size_t client_scratch_elems_ = 0;
    std::map<int, CrtTablesDev> crt_;
    std::map<uint32_t, u64 *> d_galois_;
    std::map<uint32_t, uint32_t *> d_perm_;
    u64 *scratch_ = nullptr, *scratch2_ = nullptr;
    size_t scratch_bytes_ = 0, scratch2_bytes_ = 0;
    hipStream_t stream2_ = nullptr;
    hipStream_t probe_stream_ = nullptr; // clock_probe_begin's own stream
    u64 *d_clock_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
    bool dual_stream_ = true;
    u64 *rot_tmp_ = nullptr;
    size_t rot_tmp_bytes_ = 0;
    std::map<uint32_t, std::array<unsigned char, 32>> perm_rows_;
    he355_path_stats_t paths_{};
    bool lds_auto_ = !getenv("HE355_LDS_MAX");
    u64 lds_max_ = getenv("HE355_LDS_MAX") ? (u64)std::max(0, std::atoi(getenv("HE355_LDS_MAX"))) : 0;
    bool level_walk_ = !(getenv("HE355_LEVEL_WALK") && getenv("HE355_LEVEL_WALK")[0] == '0'); // he355_rotate_sum: trie levels as grouped launches
    unsigned char *d_groups_ = nullptr; // group tables of the grouped key switches (upload_groups)
    size_t groups_bytes_ = 0, groups_next_ = 0;
    u64 *bfv_scratch_ = nullptr;
    size_t bfv_bytes_ = 0;
    std::map<int, BehzDev> behz_;
    std::map<uint32_t, uint32_t *> d_gather_;
    size_t chunk_ = 1024;
    DevicePool pool_;
    const bool pool_on_ = !(getenv("HE355_POOL") && getenv("HE355_POOL")[0] == '0');
};

} // namespace he355

// =========================================================================================================
// C ABI
// =========================================================================================================
using namespace he355;

struct he355_ctx {
    std::unique_ptr<Params> params;
    std::unique_ptr<DeviceContext> dev;
};

static thread_local std::string g_last_error;

template <class F> static int guarded(F &&f)
{
    try {
        f();
        return HE355_OK;
    } catch (const DeviceError &e) {
        g_last_error = e.what();
        return HE355_E_DEVICE;
    } catch (const std::invalid_argument &e) {
        g_last_error = e.what();
        return HE355_E_INVALID_ARGS;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return HE355_E_PARAMS;
    } catch (...) {
        g_last_error = "unknown error";
        return HE355_E_CRITICAL;
    }
}
static DeviceContext &dev(he355_ctx *c)
{
    if (!c) throw std::invalid_argument("null context");
    if (!c->dev) throw DeviceError("device not initialised: call he355_device_init first (no CPU fallback exists)");
    return *c->dev;
}
static Indexer to_ix(const he355_indexer &i)
{
    Indexer x;
    x.a_base = i.a_base; x.b_base = i.b_base; x.b1 = i.b1 ? i.b1 : 1; x.pairwise = i.pairwise; x.pad_ = 0;
    return x;
}

const he355::Params *he355_internal_params(const he355_ctx *ctx) { return ctx ? ctx->params.get() : nullptr; }

extern "C" {

const char *he355_last_error(void) { return g_last_error.c_str(); }

int he355_ctx_create(int scheme, uint64_t N, const int32_t *bit_sizes, uint64_t n, int plain_bits, int sec128, he355_ctx **out)
{
    if (!out || !bit_sizes) { g_last_error = "null argument"; return HE355_E_INVALID_ARGS; }
    *out = nullptr;
    try {
        std::unique_ptr<he355_ctx> c(new he355_ctx());
        c->params.reset(Params::create(scheme, (size_t)N, std::vector<int>(bit_sizes, bit_sizes + n), plain_bits, sec128 != 0));
        *out = c.release();
        return HE355_OK;
    } catch (const std::exception &e) { // SEAL exceptions are reported as code 2 by the reference (seal_context.cpp:94-97)
        g_last_error = e.what();
        return HE355_E_PARAMS;
    }
}
int he355_ctx_create_primes(int scheme, uint64_t N, const uint64_t *primes, uint64_t n, uint64_t plain_modulus, he355_ctx **out)
{
    if (!out || !primes) { g_last_error = "null argument"; return HE355_E_INVALID_ARGS; }
    *out = nullptr;
    try {
        std::unique_ptr<he355_ctx> c(new he355_ctx());
        c->params.reset(Params::create_primes(scheme, (size_t)N, std::vector<u64>(primes, primes + n), plain_modulus));
        *out = c.release();
        return HE355_OK;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return HE355_E_PARAMS;
    }
}
void he355_ctx_destroy(he355_ctx *ctx) { delete ctx; }
uint64_t he355_poly_degree(const he355_ctx *c) { return c->params->N; }
uint64_t he355_key_modulus_count(const he355_ctx *c) { return c->params->K; }
uint64_t he355_data_modulus_count(const he355_ctx *c) { return c->params->Ltop; }
uint64_t he355_modulus(const he355_ctx *c, uint64_t i) { return i < c->params->K ? c->params->primes[i].q : 0; }
uint64_t he355_plain_modulus(const he355_ctx *c) { return c->params->plain_modulus; }
int he355_prime_uses_fp64(const he355_ctx *c, uint64_t i) { return i < c->params->K ? (int)c->params->primes[i].f64 : 0; }
uint64_t he355_bfv_aux_base(const he355_ctx *c, int level, uint64_t *out, uint64_t cap)
{
    const Params &p = *c->params;
    if (p.scheme != kSchemeBFV || level < 1 || (size_t)level > p.Ltop) return 0;
    try {
        const size_t nB = p.behz_nB(level);
        for (size_t i = 0; i <= nB && i < cap; ++i) out[i] = p.aux[i].q;
        return nB + 1;
    } catch (const std::exception &) {
        return 0;
    }
}
uint32_t he355_galois_elt_from_step(const he355_ctx *c, int step) { return c->params->galois_elt_from_step(step); }
uint64_t he355_galois_elts_all(const he355_ctx *c, uint32_t *out, uint64_t cap)
{
    const auto v = c->params->galois_elts_all();
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}

int he355_device_count(int *count)
{
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    if (e != hipSuccess) { g_last_error = std::string("HIP error: ") + hipGetErrorString(e); return HE355_E_DEVICE; }
    return HE355_OK;
}
int he355_device_init(he355_ctx *c, int device)
{
    return guarded([&] {
        if (!c) throw std::invalid_argument("null context");
        c->dev.reset(new DeviceContext(*c->params, device));
    });
}
int he355_mem_info(he355_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes)
{
    return guarded([&] {
        dev(c).use();
        size_t f = 0, t = 0;
        HIPCHECK(hipMemGetInfo(&f, &t));
        if (free_bytes) *free_bytes = f;
        if (total_bytes) *total_bytes = t;
    });
}
int he355_malloc(he355_ctx *c, uint64_t bytes, void **d_ptr)
{
    return guarded([&] {
        if (!d_ptr) throw std::invalid_argument("null pointer");
        *d_ptr = dev(c).pool_alloc(bytes);
    });
}
int he355_free(he355_ctx *c, void *d_ptr)
{
    return guarded([&] {
        dev(c).pool_free(d_ptr);
    });
}
int he355_alloc_stats(he355_ctx *c, he355_alloc_stats_t *out)
{
    return guarded([&] {
        if (!out) throw std::invalid_argument("null pointer");
        if (!c) { // process-wide totals (every context's pool)
            out->raw_mallocs = g_pool_totals.raw_mallocs; out->raw_frees = g_pool_totals.raw_frees;
            out->pool_hits = g_pool_totals.pool_hits; out->pool_misses = g_pool_totals.pool_misses;
            out->cached_bytes = out->live_bytes = 0;
            return;
        }
        const DevicePool::Stats st = dev(c).alloc_stats();
        out->raw_mallocs = st.raw_mallocs; out->raw_frees = st.raw_frees;
        out->pool_hits = st.pool_hits; out->pool_misses = st.pool_misses;
        out->cached_bytes = st.cached_bytes; out->live_bytes = st.live_bytes;
    });
}
int he355_path_stats(he355_ctx *c, he355_path_stats_t *out, int reset)
{
    return guarded([&] {
        if (!out) throw std::invalid_argument("null pointer");
        *out = dev(c).path_stats(reset != 0);
    });
}
int he355_pool_trim(he355_ctx *c, uint64_t *released_bytes)
{
    return guarded([&] {
        const size_t b = dev(c).pool_trim();
        if (released_bytes) *released_bytes = b;
    });
}
int he355_upload(he355_ctx *c, void *d_dst, const void *h_src, uint64_t bytes)
{
    return guarded([&] {
        dev(c).use();
        HIPCHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, dev(c).stream()));
        HIPCHECK(hipStreamSynchronize(dev(c).stream()));
    });
}
int he355_download(he355_ctx *c, void *h_dst, const void *d_src, uint64_t bytes)
{
    return guarded([&] {
        dev(c).use();
        HIPCHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, dev(c).stream()));
        HIPCHECK(hipStreamSynchronize(dev(c).stream()));
    });
}
int he355_copy(he355_ctx *c, void *d_dst, const void *d_src, uint64_t bytes)
{
    return guarded([&] {
        dev(c).use();
        if (bytes) HIPCHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, dev(c).stream()));
    });
}
int he355_copy_peer(he355_ctx *dst_ctx, void *d_dst, he355_ctx *src_ctx, const void *d_src, uint64_t bytes)
{
    return guarded([&] {
        // both contexts' streams are drained first (the source must be complete, the destination idle); the copy itself is
        // synchronous: load() / store() use it outside the timed operate()
        dev(src_ctx).sync();
        dev(dst_ctx).sync();
        dev(dst_ctx).use();
        if (bytes) HIPCHECK(hipMemcpyPeer(d_dst, dev(dst_ctx).device(), d_src, dev(src_ctx).device(), bytes));
    });
}
int he355_sync(he355_ctx *c) { return guarded([&] { dev(c).sync(); }); }
int he355_fill_uniform(he355_ctx *c, uint64_t *d_dst, uint64_t n_polys, const uint8_t *prime_of, uint32_t period, uint64_t seed)
{
    return guarded([&] { dev(c).fill_uniform(d_dst, n_polys, prime_of, period, seed); });
}
int he355_fill_uniform_at(he355_ctx *c, uint64_t *d_dst, uint64_t n_polys, const uint8_t *prime_of, uint32_t period, uint64_t seed, uint64_t first_poly)
{
    return guarded([&] { dev(c).fill_uniform(d_dst, n_polys, prime_of, period, seed, first_poly); });
}
int he355_set_dual_stream(he355_ctx *c, int on) { return guarded([&] { dev(c).set_dual_stream(on != 0); }); }
int he355_set_latency_max(he355_ctx *c, uint64_t n) { return guarded([&] { dev(c).set_latency_max(n); }); }
int he355_set_lds_max(he355_ctx *c, uint64_t n)
{
    return guarded([&] { dev(c).set_lds_max(n); });
}
int he355_set_level_walk(he355_ctx *c, int on) { return guarded([&] { dev(c).set_level_walk(on != 0); }); }
int he355_set_relin_key(he355_ctx *c, const uint64_t *h_key)
{
    return guarded([&] { dev(c).key_from_host(dev(c).relin_slot(), h_key); });
}
int he355_set_galois_key(he355_ctx *c, uint32_t elt, const uint64_t *h_key)
{
    return guarded([&] { dev(c).key_from_host(dev(c).galois_slot(elt), h_key); });
}
int he355_keygen_relin(he355_ctx *c, uint64_t seed)
{
    return guarded([&] { dev(c).key_generate(dev(c).relin_slot(), seed, 0); });
}
int he355_keygen_galois(he355_ctx *c, uint32_t galois_elt, uint64_t seed)
{
    return guarded([&] {
        if (!(galois_elt & 1) || galois_elt >= 2 * he355_poly_degree(c) || galois_elt < 3) throw std::invalid_argument("Galois element is not valid");
        dev(c).key_generate(dev(c).galois_slot(galois_elt), seed, galois_elt);
    });
}
int he355_set_relin_key_synthetic(he355_ctx *c, uint64_t seed)
{
    return guarded([&] { dev(c).key_synthetic(dev(c).relin_slot(), seed); });
}
int he355_set_galois_key_synthetic(he355_ctx *c, uint32_t elt, uint64_t seed)
{
    return guarded([&] { dev(c).key_synthetic(dev(c).galois_slot(elt), seed); });
}
int he355_add(he355_ctx *c, int L, int size, uint64_t n, const uint64_t *a, const uint64_t *b, he355_indexer ix, uint64_t *out)
{
    return guarded([&] { dev(c).addsub(L, size, n, a, b, to_ix(ix), out, false); });
}
int he355_sub(he355_ctx *c, int L, int size, uint64_t n, const uint64_t *a, const uint64_t *b, he355_indexer ix, uint64_t *out)
{
    return guarded([&] { dev(c).addsub(L, size, n, a, b, to_ix(ix), out, true); });
}
int he355_multiply(he355_ctx *c, int L, uint64_t n, const uint64_t *a, const uint64_t *b, he355_indexer ix, uint64_t *out)
{
    return guarded([&] { dev(c).multiply(L, n, a, b, to_ix(ix), out); });
}
int he355_bfv_multiply(he355_ctx *c, int L, uint64_t n, const uint64_t *a, const uint64_t *b, he355_indexer ix, uint64_t *out)
{
    return guarded([&] { dev(c).bfv_multiply(L, n, a, b, to_ix(ix), out); });
}
int he355_bfv_multiply_relin_accumulate(he355_ctx *c, int L, uint64_t rows, uint64_t cols, uint64_t inner, const uint64_t *a, uint64_t a_stride_i,
                                        uint64_t a_stride_k, const uint64_t *b, uint64_t b_stride_k, uint64_t b_stride_j, uint64_t *out)
{
    return guarded([&] { dev(c).bfv_multiply_relin_accumulate(L, rows, cols, inner, a, a_stride_i, a_stride_k, b, b_stride_k, b_stride_j, out); });
}
int he355_multiply_relin(he355_ctx *c, int L, uint64_t n, const uint64_t *a, const uint64_t *b, he355_indexer ix, int rescale, uint64_t *out)
{
    return guarded([&] { dev(c).multiply_relin(L, n, a, b, to_ix(ix), rescale != 0, out); });
}
int he355_multiply_plain(he355_ctx *c, int L, int size, uint64_t n, const uint64_t *ct, const uint64_t *pt, he355_indexer ix, uint64_t *out)
{
    return guarded([&] { dev(c).plain_op(L, size, n, ct, pt, to_ix(ix), out, 0); });
}
int he355_add_plain(he355_ctx *c, int L, int size, uint64_t n, const uint64_t *ct, const uint64_t *pt, he355_indexer ix, uint64_t *out)
{
    return guarded([&] { dev(c).plain_op(L, size, n, ct, pt, to_ix(ix), out, 1); });
}
int he355_mod_switch_drop(he355_ctx *c, int L, int L_to, uint64_t n_polys, const uint64_t *in, uint64_t *out)
{
    return guarded([&] { dev(c).mod_switch_drop(L, L_to, n_polys, in, out); });
}
int he355_sum(he355_ctx *c, int L, int size, uint64_t n, const uint64_t *in, uint64_t *out)
{
    return guarded([&] { dev(c).sum(L, size, n, in, out); });
}
int he355_multiply_accumulate(he355_ctx *c, int L, uint64_t rows, uint64_t cols, uint64_t inner, const uint64_t *a, uint64_t a_stride_i,
                              uint64_t a_stride_k, const uint64_t *b, uint64_t b_stride_k, uint64_t b_stride_j, uint64_t *out)
{
    return guarded([&] { dev(c).multiply_accumulate(L, rows, cols, inner, a, a_stride_i, a_stride_k, b, b_stride_k, b_stride_j, out); });
}
int he355_relinearize_rescale(he355_ctx *c, int L, uint64_t n, const uint64_t *ct3, uint64_t *out)
{
    return guarded([&] { dev(c).relinearize(L, n, ct3, out, true); });
}
int he355_relinearize(he355_ctx *c, int L, uint64_t n, const uint64_t *ct3, uint64_t *out)
{
    return guarded([&] { dev(c).relinearize(L, n, ct3, out); });
}
int he355_rescale(he355_ctx *c, int L, int size, uint64_t n, const uint64_t *in, uint64_t *out)
{
    return guarded([&] { dev(c).rescale(L, size, n, in, out); });
}
int he355_apply_galois(he355_ctx *c, int L, uint64_t n, const uint64_t *in, uint32_t elt, uint64_t *out)
{
    return guarded([&] { dev(c).apply_galois(L, n, in, elt, out); });
}
int he355_rotate(he355_ctx *c, int L, uint64_t n, const uint64_t *in, int step, uint64_t *out)
{
    return guarded([&] { dev(c).rotate(L, n, in, step, out); });
}
int he355_rotate_each(he355_ctx *c, int L, uint64_t n, const uint64_t *in, const int32_t *steps, uint64_t *out)
{
    return guarded([&] {
        if (n && !steps) throw std::invalid_argument("rotate_each needs one step per ciphertext");
        dev(c).rotate_each(L, n, in, steps, out);
    });
}
int he355_rotate_sum(he355_ctx *c, int L, uint64_t n, const uint64_t *in, const int32_t *steps, uint64_t n_steps, uint64_t *out, uint64_t *key_switches)
{
    return guarded([&] {
        if (n_steps && !steps) throw std::invalid_argument("rotate_sum needs the steps");
        const uint64_t k = dev(c).rotate_sum(L, n, in, steps, n_steps, out);
        if (key_switches) *key_switches = k;
    });
}
int he355_rotate_add(he355_ctx *c, int L, uint64_t n, const uint64_t *in, int step, const uint64_t *addend, uint64_t *out)
{
    return guarded([&] {
        if (!addend) throw std::invalid_argument("rotate_add needs an addend");
        dev(c).rotate(L, n, in, step, out, addend);
    });
}
int he355_encrypt_zero(he355_ctx *c, uint64_t n, uint64_t seed, uint64_t first_index, uint64_t *d_out)
{
    return guarded([&] { dev(c).encrypt(n, nullptr, seed, first_index, d_out); });
}
int he355_set_zero_stream(he355_ctx *c, uint64_t seed, uint64_t first_index)
{
    return guarded([&] { dev(c).set_zero_stream(seed, first_index); });
}
int he355_accumulate(he355_ctx *c, int L, uint64_t n, uint64_t *inout, uint64_t count, uint64_t *tmp)
{
    return guarded([&] { dev(c).accumulate(L, n, inout, count, tmp); });
}
int he355_ntt_forward(he355_ctx *c, uint64_t *polys, uint64_t n_polys, const uint8_t *prime_of, uint32_t period)
{
    return guarded([&] { dev(c).ntt(polys, n_polys, prime_of, period, false); });
}
int he355_ntt_inverse(he355_ctx *c, uint64_t *polys, uint64_t n_polys, const uint8_t *prime_of, uint32_t period)
{
    return guarded([&] { dev(c).ntt(polys, n_polys, prime_of, period, true); });
}
int he355_timer_begin(he355_ctx *c) { return guarded([&] { dev(c).timer_begin(); }); }
int he355_timer_end(he355_ctx *c, float *ms)
{
    return guarded([&] {
        const float v = dev(c).timer_end();
        if (ms) *ms = v;
    });
}
int he355_set_public_key(he355_ctx *c, const uint64_t *h_pk)
{
    return guarded([&] { dev(c).set_public_key(h_pk); });
}
int he355_set_secret_key(he355_ctx *c, const uint64_t *h_sk)
{
    return guarded([&] { dev(c).set_secret_key(h_sk); });
}
int he355_encrypt(he355_ctx *c, uint64_t n, const uint64_t *d_plain, uint64_t seed, uint64_t first_index, uint64_t *d_out)
{
    return guarded([&] { dev(c).encrypt(n, d_plain, seed, first_index, d_out); });
}
int he355_decrypt(he355_ctx *c, int L, int size, uint64_t n, const uint64_t *d_ct, uint64_t *d_out)
{
    return guarded([&] { dev(c).decrypt(L, size, n, d_ct, d_out); });
}
int he355_ckks_encode(he355_ctx *c, uint64_t n, const double *d_values, uint64_t count, double scale, uint64_t *d_plain)
{
    return guarded([&] { dev(c).ckks_encode(n, d_values, count, scale, d_plain); });
}
int he355_ckks_decode(he355_ctx *c, int L, uint64_t n, const uint64_t *d_plain, double scale, double *d_out)
{
    return guarded([&] { dev(c).ckks_decode(L, n, d_plain, scale, d_out); });
}
int he355_bfv_encode(he355_ctx *c, uint64_t n, const int64_t *d_values, uint64_t count, uint64_t *d_plain)
{
    return guarded([&] { dev(c).bfv_encode(n, reinterpret_cast<const long long *>(d_values), count, d_plain); });
}
int he355_bfv_decode(he355_ctx *c, uint64_t n, const uint64_t *d_plain, int64_t *d_out)
{
    return guarded([&] { dev(c).bfv_decode(n, d_plain, reinterpret_cast<long long *>(d_out)); });
}
int he355_ckks_decode_slots(he355_ctx *c, int L, uint64_t n, const uint64_t *d_plain, double scale, const uint64_t *ranges, uint64_t n_ranges, double *d_out)
{
    return guarded([&] {
        if (!ranges) throw std::invalid_argument("null pointer");
        dev(c).ckks_decode(L, n, d_plain, scale, d_out, ranges, n_ranges);
    });
}
int he355_bfv_decode_slots(he355_ctx *c, uint64_t n, const uint64_t *d_plain, const uint64_t *ranges, uint64_t n_ranges, int64_t *d_out)
{
    return guarded([&] {
        if (!ranges) throw std::invalid_argument("null pointer");
        dev(c).bfv_decode(n, d_plain, reinterpret_cast<long long *>(d_out), ranges, n_ranges);
    });
}
// Page-locked host memory for the small, frequent transfers of a harness run (decode results, encode inputs): a copy to or from it is one
// DMA with no staging pass through the runtime's own pinned buffer
int he355_host_alloc(he355_ctx *c, uint64_t bytes, void **h_ptr)
{
    return guarded([&] {
        if (!h_ptr) throw std::invalid_argument("null pointer");
        dev(c).use();
        HIPCHECK(hipHostMalloc(h_ptr, bytes ? bytes : 8, hipHostMallocDefault));
    });
}
int he355_host_free(he355_ctx *c, void *h_ptr)
{
    return guarded([&] {
        dev(c).use();
        if (h_ptr) HIPCHECK(hipHostFree(h_ptr));
    });
}
int he355_probe_dominant_kernel(he355_ctx *c, float *total_ms, uint64_t *launches, uint64_t *ops)
{
    return guarded([&] {
        u64 l = 0, o = 0;
        dev(c).probe_result(total_ms, &l, &o);
        if (launches) *launches = l;
        if (ops) *ops = o;
    });
}
int he355_clock_probe_begin(he355_ctx *c, uint64_t duration_us) { return guarded([&] { dev(c).clock_probe_begin(duration_us); }); }
int he355_clock_probe_end(he355_ctx *c, double *mhz, double *seconds) { return guarded([&] { dev(c).clock_probe_end(mhz, seconds); }); }
int he355_set_chunk(he355_ctx *c, uint64_t ops)
{
    return guarded([&] { dev(c).set_chunk((size_t)ops); });
}

} // extern "C"
