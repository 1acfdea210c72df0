/*
 * he_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see he_oracle.h for the rules and the
 * "PARITY UNPINNED" statement).  Plain C11 + unsigned __int128 + OpenMP.
 *
 * Every function names the SEAL v3.7.2 routine it restates (native/src/seal/...; SEAL is the
 * un-vendored dependency behind every seal::Evaluator call of the reference) and the
 * reference call site that reaches it.  Nothing here was copied: the algorithms are restated from
 * their published descriptions (Harvey 2014 butterflies; Barrett; Shoup; Bajard-Eynard-Hasan-Zucca
 * 2016; Cheon-Han-Kim-Kim-Song 2018 RNS-CKKS; hybrid key switching with one special prime).
 */
#include "he_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------------ */
/* Per-thread scratch.  The reference's operate() loop hands SEAL a thread-local memory pool          */
/* (MemoryPoolHandle::ThreadLocal(), src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:343), so */
/* its evaluator takes no system allocation per ciphertext.  The temporaries of the evaluator         */
/* functions below come from the same kind of cache: a block is malloc'ed once per thread and size    */
/* and handed out again by scr_alloc() after scr_free().  (With malloc/free per op, the 27 MiB of    */
/* key-switch temporaries at N = 2^15 are an mmap + page faults + munmap per ciphertext per thread:   */
/* 128 threads serialise on the address-space lock and the timed CPU baseline drops to 14 % parallel  */
/* efficiency.)                                                                                       */
/* ------------------------------------------------------------------------------------------------ */
#define SCR_SLOTS 32
#define SCR_HDR 64
typedef struct { void *blk[SCR_SLOTS]; size_t cap[SCR_SLOTS]; int n; } scr_cache;
static _Thread_local scr_cache g_scr;
static void *scr_alloc(size_t bytes)
{
    scr_cache *c = &g_scr;
    int best = -1;
    for (int i = 0; i < c->n; ++i)
        if (c->cap[i] >= bytes && (best < 0 || c->cap[i] < c->cap[best])) best = i;
    if (best >= 0 && c->cap[best] <= 2 * bytes + 4096) {
        void *b = c->blk[best];
        c->blk[best] = c->blk[--c->n];
        c->cap[best] = c->cap[c->n];
        return (char *)b + SCR_HDR;
    }
    const size_t cap = (bytes + 63) & ~(size_t)63;
    char *b = (char *)aligned_alloc(64, cap + SCR_HDR);
    if (!b) { fprintf(stderr, "he_oracle: out of memory (%zu bytes)\n", bytes); abort(); }
    *(size_t *)b = cap;
    return b + SCR_HDR;
}
static void scr_free(void *p)
{
    if (!p) return;
    char *b = (char *)p - SCR_HDR;
    scr_cache *c = &g_scr;
    if (c->n < SCR_SLOTS) { c->blk[c->n] = b; c->cap[c->n] = *(size_t *)b; ++c->n; }
    else free(b);
}
/* give the calling thread's cached blocks back (ho_scratch_release_all: every thread of the OpenMP team) */
void ho_scratch_release(void)
{
    scr_cache *c = &g_scr;
    for (int i = 0; i < c->n; ++i) free(c->blk[i]);
    c->n = 0;
}
void ho_scratch_release_all(void)
{
#pragma omp parallel
    ho_scratch_release();
    ho_scratch_release();
}

/* ------------------------------------------------------------------------------------------------ */
/* Modulus with Barrett constant  (seal/modulus.h: Modulus::const_ratio = floor(2^128/q))            */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    u64 q;
    u64 cr0, cr1; /* floor(2^128 / q) low, high */
    int bits;
} ho_mod;

static void mod_init(ho_mod *m, u64 q)
{
    m->q = q;
    /* floor(2^128 / q) by two-step long division */
    u128 hi = ((u128)1 << 64) / q;              /* floor(2^64/q)            */
    u128 rem = ((u128)1 << 64) % q;             /* 2^64 mod q               */
    u128 lo = (rem << 64) / q;                  /* floor(rem*2^64 / q)      */
    m->cr1 = (u64)hi;
    m->cr0 = (u64)lo;
    int b = 0;
    for (u64 t = q; t; t >>= 1) ++b;
    m->bits = b;
}

/* util/uintarithsmallmod.h: barrett_reduce_64 */
static inline u64 barrett64(u64 x, const ho_mod *m)
{
    u64 t = (u64)(((u128)x * m->cr1) >> 64);
    u64 r = x - t * m->q;
    return r >= m->q ? r - m->q : r;
}
/* util/uintarithsmallmod.h: barrett_reduce_128 (input < 2^128) */
static inline u64 barrett128(u128 x, const ho_mod *m)
{
    u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    u64 carry = (u64)(((u128)x0 * m->cr0) >> 64);
    u128 t2 = (u128)x0 * m->cr1;
    u128 s = (u128)(u64)t2 + carry;
    u64 tmp1 = (u64)s;
    u64 tmp3 = (u64)(t2 >> 64) + (u64)(s >> 64);
    t2 = (u128)x1 * m->cr0;
    s = (u128)tmp1 + (u64)t2;
    carry = (u64)(t2 >> 64) + (u64)(s >> 64);
    u64 quo = x1 * m->cr1 + tmp3 + carry;
    u64 r = x0 - quo * m->q;
    return r >= m->q ? r - m->q : r;
}
static inline u64 mulmod(u64 a, u64 b, const ho_mod *m) { return barrett128((u128)a * b, m); }
static inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }
static u64 powmod(u64 b, u64 e, const ho_mod *m)
{
    u64 r = 1;
    b = barrett64(b, m);
    while (e) {
        if (e & 1) r = mulmod(r, b, m);
        b = mulmod(b, b, m);
        e >>= 1;
    }
    return r;
}
static u64 invmod_prime(u64 a, const ho_mod *m) { return powmod(a, m->q - 2, m); }
/* Shoup operand: util/uintarithsmallmod.h MultiplyUIntModOperand {operand, quotient=floor(op*2^64/q)} */
static inline u64 shoup_quot(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
static inline u64 mul_shoup_lazy(u64 x, u64 w, u64 wq, u64 q) /* result in [0,2q) */
{
    u64 t = (u64)(((u128)x * wq) >> 64);
    return w * x - t * q;
}
static inline u64 mul_shoup(u64 x, u64 w, u64 wq, u64 q)
{
    u64 r = mul_shoup_lazy(x, w, wq, q);
    return r >= q ? r - q : r;
}

/* ------------------------------------------------------------------------------------------------ */
/* Primes (util/numth.cpp: is_prime, get_primes;  modulus.cpp: CoeffModulus::Create)                 */
/* ------------------------------------------------------------------------------------------------ */
int ho_is_prime(u64 n)
{
    if (n < 2) return 0;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (size_t i = 0; i < 12; ++i) {
        if (n == small[i]) return 1;
        if (n % small[i] == 0) return 0;
    }
    ho_mod m;
    mod_init(&m, n);
    u64 d = n - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; ++r; }
    for (size_t i = 0; i < 12; ++i) { /* these 12 bases are a deterministic test below 2^64 */
        u64 x = powmod(small[i], d, &m);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int k = 1; k < r; ++k) {
            x = mulmod(x, x, &m);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

size_t ho_get_primes(u64 factor, int bit_size, size_t count, u64 *out)
{
    /* numth.cpp get_primes: start at the largest value = 1 mod factor below 2^bit_size, step -factor,
     * stop at 2^(bit_size-1) */
    u64 value = ((((u64)1 << bit_size) - 1) / factor) * factor + 1;
    u64 lower = (u64)1 << (bit_size - 1);
    size_t found = 0;
    while (found < count && value > lower) {
        if (ho_is_prime(value)) out[found++] = value;
        value -= factor;
    }
    return found;
}

/* CoeffModulus::Create: per distinct bit size a descending list; slots are served from the BACK of
 * each list in order of appearance. */
static int coeff_modulus_create(size_t N, const int *bits, size_t n, u64 *out)
{
    u64 *lists[64] = {0};
    size_t left[64] = {0};
    for (size_t i = 0; i < n; ++i) {
        int b = bits[i];
        if (b < 2 || b > 60) return -1;
        if (!lists[b]) {
            size_t cnt = 0;
            for (size_t k = 0; k < n; ++k) cnt += (bits[k] == b);
            lists[b] = (u64 *)malloc(cnt * sizeof(u64));
            if (ho_get_primes(2 * (u64)N, b, cnt, lists[b]) != cnt) {
                for (int k = 0; k < 64; ++k) free(lists[k]);
                return -2;
            }
            left[b] = cnt;
        }
    }
    for (size_t i = 0; i < n; ++i) out[i] = lists[bits[i]][--left[bits[i]]];
    for (int k = 0; k < 64; ++k) free(lists[k]);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* NTT tables (util/ntt.cpp NTTTables::initialize; util/numth.cpp try_minimal_primitive_root)        */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    ho_mod m;
    u64 root;          /* minimal primitive 2N-th root */
    u64 *w, *wq;       /* forward: w[bitrev(i)] = root^i, Shoup quotients */
    u64 *iw, *iwq;     /* inverse: iw[k] = w[k]^{-1} (same indexing as forward) */
    u64 inv_n, inv_n_q;
} ntt_tab;

static uint32_t bitrev32(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

static int find_minimal_root(u64 two_n, const ho_mod *m, u64 *out)
{
    u64 q = m->q;
    if ((q - 1) % two_n) return -1;
    u64 e = (q - 1) / two_n, root = 0;
    for (u64 g = 2; g < 1000; ++g) {
        u64 r = powmod(g, e, m);
        if (powmod(r, two_n / 2, m) == q - 1) { root = r; break; }
    }
    if (!root) return -1;
    /* minimal over all primitive roots = all odd powers */
    u64 gen2 = mulmod(root, root, m), cur = root, best = root;
    for (u64 i = 0; i < two_n; i += 2) {
        if (cur < best) best = cur;
        cur = mulmod(cur, gen2, m);
    }
    *out = best;
    return 0;
}

static int ntt_init(ntt_tab *t, size_t N, int logn, u64 q)
{
    mod_init(&t->m, q);
    if (find_minimal_root(2 * (u64)N, &t->m, &t->root)) return -1;
    t->w = (u64 *)malloc(N * 8); t->wq = (u64 *)malloc(N * 8);
    t->iw = (u64 *)malloc(N * 8); t->iwq = (u64 *)malloc(N * 8);
    u64 inv_root = invmod_prime(t->root, &t->m);
    u64 p = 1, ip = 1;
    for (size_t i = 0; i < N; ++i) {
        uint32_t k = bitrev32((uint32_t)i, logn);
        t->w[k] = p; t->wq[k] = shoup_quot(p, q);
        t->iw[k] = ip; t->iwq[k] = shoup_quot(ip, q);
        p = mulmod(p, t->root, &t->m);
        ip = mulmod(ip, inv_root, &t->m);
    }
    t->inv_n = invmod_prime((u64)N, &t->m);
    t->inv_n_q = shoup_quot(t->inv_n, q);
    return 0;
}
static void ntt_free(ntt_tab *t) { free(t->w); free(t->wq); free(t->iw); free(t->iwq); }

/* util/ntt.cpp ntt_negacyclic_harvey (+ util/dwthandler.h transform_to_rev): Cooley-Tukey, natural
 * in, bit-reversed out, lazy butterflies in [0,4q), final correction to [0,q). */
static void ntt_forward(const ntt_tab *t, size_t N, u64 *x)
{
    const u64 q = t->m.q, two_q = 2 * q;
    size_t gap = N >> 1;
    for (size_t m = 1; m < N; m <<= 1, gap >>= 1) {
        for (size_t i = 0; i < m; ++i) {
            const u64 w = t->w[m + i], wq = t->wq[m + i];
            u64 *a = x + 2 * i * gap, *b = a + gap;
            for (size_t j = 0; j < gap; ++j) {
                u64 u = a[j] >= two_q ? a[j] - two_q : a[j];
                u64 v = mul_shoup_lazy(b[j], w, wq, q);
                a[j] = u + v;
                b[j] = u + two_q - v;
            }
        }
    }
    for (size_t j = 0; j < N; ++j) {
        u64 v = x[j];
        if (v >= two_q) v -= two_q;
        if (v >= q) v -= q;
        x[j] = v;
    }
}
/* util/ntt.cpp inverse_ntt_negacyclic_harvey (+ transform_from_rev): Gentleman-Sande, bit-reversed
 * in, natural out, N^{-1} folded in, output [0,q). */
static void ntt_inverse(const ntt_tab *t, size_t N, u64 *x)
{
    const u64 q = t->m.q, two_q = 2 * q;
    size_t gap = 1;
    for (size_t m = N >> 1; m >= 1; m >>= 1, gap <<= 1) {
        for (size_t i = 0; i < m; ++i) {
            const u64 w = t->iw[m + i], wq = t->iwq[m + i];
            u64 *a = x + 2 * i * gap, *b = a + gap;
            for (size_t j = 0; j < gap; ++j) {
                u64 u = a[j], v = b[j];
                u64 s = u + v;
                a[j] = s >= two_q ? s - two_q : s;
                b[j] = mul_shoup_lazy(u + two_q - v, w, wq, q);
            }
        }
    }
    for (size_t j = 0; j < N; ++j) x[j] = mul_shoup(x[j], t->inv_n, t->inv_n_q, q);
}

/* ------------------------------------------------------------------------------------------------ */
/* Context (context.cpp SEALContext; the reference builds it at seal_context.cpp:79-90,107-119)      */
/* ------------------------------------------------------------------------------------------------ */
struct ho_ctx {
    int scheme;
    size_t N;
    int logn;
    size_t K;      /* all primes, special last */
    size_t Ltop;   /* data residues at first level */
    ntt_tab *t;     /* per prime */
    ho_mod plain;  /* BFV */
    u64 t_value;
    /* BFV (BEHZ) tables are built lazily in he_oracle_bfv part */
    void *bfv;
};

static int max_bits_tc128(size_t N)
{
    switch (N) { /* util/hestdparms.h SEAL_HE_STD_PARMS_128_TC */
    case 1024: return 27;
    case 2048: return 54;
    case 4096: return 109;
    case 8192: return 218;
    case 16384: return 438;
    case 32768: return 881;
    default: return 0;
    }
}

static void seterr(char *err, size_t n, const char *msg)
{
    if (err && n) { strncpy(err, msg, n - 1); err[n - 1] = 0; }
}

static void bfv_free(void *p);

ho_ctx *ho_ctx_create_primes(int scheme, size_t N, const u64 *primes, size_t n, u64 plain_modulus, char *err, size_t errlen)
{
    if (scheme != HO_SCHEME_BFV && scheme != HO_SCHEME_CKKS) { seterr(err, errlen, "unsupported scheme"); return NULL; }
    int logn = 0;
    while (((size_t)1 << logn) < N) ++logn;
    if (((size_t)1 << logn) != N || N < 2) { seterr(err, errlen, "poly_modulus_degree must be a power of two"); return NULL; }
    if (n < 1 || n > 62) { seterr(err, errlen, "invalid coefficient modulus count"); return NULL; }
    ho_ctx *c = (ho_ctx *)calloc(1, sizeof(*c));
    c->scheme = scheme; c->N = N; c->logn = logn; c->K = n;
    c->Ltop = n > 1 ? n - 1 : 1;
    c->t = (ntt_tab *)calloc(n, sizeof(ntt_tab));
    for (size_t i = 0; i < n; ++i) {
        for (size_t k = 0; k < i; ++k)
            if (primes[k] == primes[i]) { seterr(err, errlen, "coefficient moduli must be distinct"); ho_ctx_destroy(c); return NULL; }
        if (!ho_is_prime(primes[i]) || ntt_init(&c->t[i], N, logn, primes[i])) {
            seterr(err, errlen, "coefficient modulus is not an NTT-friendly prime");
            ho_ctx_destroy(c);
            return NULL;
        }
    }
    if (scheme == HO_SCHEME_BFV) {
        if (plain_modulus < 2) { seterr(err, errlen, "BFV needs a plain modulus"); ho_ctx_destroy(c); return NULL; }
        mod_init(&c->plain, plain_modulus);
        c->t_value = plain_modulus;
    }
    return c;
}

ho_ctx *ho_ctx_create(int scheme, size_t N, const int *bit_sizes, size_t n_bits, int plain_bits, int sec128, char *err, size_t errlen)
{
    if (n_bits < 1 || n_bits > 62) { seterr(err, errlen, "invalid coefficient modulus count"); return NULL; }
    if (sec128) {
        int total = 0, cap = max_bits_tc128(N);
        for (size_t i = 0; i < n_bits; ++i) total += bit_sizes[i];
        if (!cap || total > cap) { seterr(err, errlen, "encryption parameters are not valid for 128-bit security"); return NULL; }
    }
    u64 primes[64];
    if (coeff_modulus_create(N, bit_sizes, n_bits, primes)) { seterr(err, errlen, "failed to find enough qualifying primes"); return NULL; }
    u64 t = 0;
    if (scheme == HO_SCHEME_BFV) { /* PlainModulus::Batching = get_primes(2N, bits, 1)[0] */
        if (ho_get_primes(2 * (u64)N, plain_bits, 1, &t) != 1) { seterr(err, errlen, "failed to find plain modulus"); return NULL; }
    }
    return ho_ctx_create_primes(scheme, N, primes, n_bits, t, err, errlen);
}

void ho_ctx_destroy(ho_ctx *c)
{
    if (!c) return;
    if (c->t) {
        for (size_t i = 0; i < c->K; ++i)
            if (c->t[i].w) ntt_free(&c->t[i]);
        free(c->t);
    }
    if (c->bfv) bfv_free(c->bfv);
    free(c);
}
size_t ho_N(const ho_ctx *c) { return c->N; }
size_t ho_key_mod_count(const ho_ctx *c) { return c->K; }
size_t ho_data_mod_count(const ho_ctx *c) { return c->Ltop; }
u64 ho_modulus(const ho_ctx *c, size_t i) { return c->t[i].m.q; }
u64 ho_plain_modulus(const ho_ctx *c) { return c->t_value; }
u64 ho_root(const ho_ctx *c, size_t i) { return c->t[i].root; }
void ho_root_powers(const ho_ctx *c, size_t i, u64 *out) { memcpy(out, c->t[i].w, c->N * 8); }
void ho_ntt(const ho_ctx *c, size_t i, u64 *p) { ntt_forward(&c->t[i], c->N, p); }
void ho_intt(const ho_ctx *c, size_t i, u64 *p) { ntt_inverse(&c->t[i], c->N, p); }

/* ------------------------------------------------------------------------------------------------ */
/* Evaluator                                                                                          */
/* ------------------------------------------------------------------------------------------------ */
/* evaluator.cpp Evaluator::add_inplace -> util/polyarithsmallmod.cpp add_poly_coeffmod.
 * Reference: ckks eltwise .cpp:340, bfv eltwise .cpp:322, seal_context.cpp:303,309,338 */
void ho_add(const ho_ctx *c, size_t L, size_t size, const u64 *a, const u64 *b, u64 *out)
{
    const size_t N = c->N;
    for (size_t k = 0; k < size; ++k)
        for (size_t i = 0; i < L; ++i) {
            const u64 q = c->t[i].m.q;
            const size_t o = (k * L + i) * N;
            for (size_t n = 0; n < N; ++n) out[o + n] = addmod(a[o + n], b[o + n], q);
        }
}
void ho_sub(const ho_ctx *c, size_t L, size_t size, const u64 *a, const u64 *b, u64 *out)
{
    const size_t N = c->N;
    for (size_t k = 0; k < size; ++k)
        for (size_t i = 0; i < L; ++i) {
            const u64 q = c->t[i].m.q;
            const size_t o = (k * L + i) * N;
            for (size_t n = 0; n < N; ++n) out[o + n] = submod(a[o + n], b[o + n], q);
        }
}

/* evaluator.cpp Evaluator::ckks_multiply (size 2 x 2): c0=a0b0, c1=a0b1+a1b0, c2=a1b1 via
 * dyadic_product_coeffmod.  Reference: ckks eltwise .cpp:343, ckks dot .cpp:325 */
void ho_multiply_ntt(const ho_ctx *c, size_t L, const u64 *a, const u64 *b, u64 *out)
{
    const size_t N = c->N, P = L * N;
    for (size_t i = 0; i < L; ++i) {
        const ho_mod *m = &c->t[i].m;
        const u64 *a0 = a + i * N, *a1 = a + P + i * N, *b0 = b + i * N, *b1 = b + P + i * N;
        u64 *c0 = out + i * N, *c1 = out + P + i * N, *c2 = out + 2 * P + i * N;
        for (size_t n = 0; n < N; ++n) {
            u64 x0 = a0[n], x1 = a1[n], y0 = b0[n], y1 = b1[n];
            c0[n] = mulmod(x0, y0, m);
            c1[n] = addmod(mulmod(x0, y1, m), mulmod(x1, y0, m), m->q);
            c2[n] = mulmod(x1, y1, m);
        }
    }
}

/* evaluator.cpp Evaluator::multiply_plain_ntt: dyadic product of every ciphertext polynomial with the plaintext
 * (seal_context.cpp:390 multiply_plain_inplace in collapseCKKS) */
void ho_multiply_plain(const ho_ctx *c, size_t L, size_t size, const u64 *ct, const u64 *plain, u64 *out)
{
    const size_t N = c->N;
    for (size_t k = 0; k < size; ++k)
        for (size_t i = 0; i < L; ++i) {
            const ho_mod *m = &c->t[i].m;
            const size_t o = (k * L + i) * N;
            for (size_t n = 0; n < N; ++n) out[o + n] = mulmod(ct[o + n], plain[i * N + n], m);
        }
}
/* evaluator.cpp Evaluator::add_plain_inplace, CKKS branch: c0 += plain (seal_context.cpp:454) */
void ho_add_plain(const ho_ctx *c, size_t L, size_t size, const u64 *ct, const u64 *plain, u64 *out)
{
    const size_t N = c->N;
    for (size_t k = 0; k < size; ++k)
        for (size_t i = 0; i < L; ++i) {
            const u64 q = c->t[i].m.q;
            const size_t o = (k * L + i) * N;
            for (size_t n = 0; n < N; ++n) out[o + n] = k == 0 ? addmod(ct[o + n], plain[i * N + n], q) : ct[o + n];
        }
}
/* evaluator.cpp Evaluator::mod_switch_drop_to_next, repeated: CKKS data just loses its last residues */
void ho_mod_switch_drop(const ho_ctx *c, size_t L, size_t L_to, size_t size, const u64 *in, u64 *out)
{
    const size_t N = c->N;
    for (size_t k = 0; k < size; ++k)
        for (size_t i = 0; i < L_to; ++i) memcpy(out + (k * L_to + i) * N, in + (k * L + i) * N, N * sizeof(u64));
}

/* evaluator.cpp Evaluator::switch_key_inplace.  Reached from relinearize_inplace (ckks dot .cpp:329,
 * matmultval .cpp:254) and from apply_galois_inplace (rotate_vector/rotate_rows, seal_context.cpp:302,337).
 * target: L residues.  key digits j<L, each [2][K][N].  Result ADDED into ct[0], ct[1]. */
void ho_switch_key(const ho_ctx *c, size_t L, const u64 *target, const u64 *key, u64 *ct)
{
    const size_t N = c->N, K = c->K, SP = K - 1; /* SP: index of the special prime */
    const int ckks = (c->scheme == HO_SCHEME_CKKS);
    u64 *coef = (u64 *)scr_alloc(L * N * 8);   /* target in coefficient form */
    u64 *tmp = (u64 *)scr_alloc(N * 8);
    u64 *prod = (u64 *)scr_alloc(2 * (L + 1) * N * 8); /* [k][i'] i' = 0..L-1 data, L = special */
    u128 *acc = (u128 *)scr_alloc(2 * N * sizeof(u128));
    memcpy(coef, target, L * N * 8);
    if (ckks)
        for (size_t j = 0; j < L; ++j) ntt_inverse(&c->t[j], N, coef + j * N);

    for (size_t ii = 0; ii <= L; ++ii) {
        const size_t ki = (ii == L) ? SP : ii; /* key-level prime index */
        const ho_mod *m = &c->t[ki].m;
        memset(acc, 0, 2 * N * sizeof(u128));
        for (size_t j = 0; j < L; ++j) {
            const u64 *operand;
            if (ckks && ii == j) {
                operand = target + j * N; /* already NTT form under the same prime */
            } else {
                const u64 qj = c->t[j].m.q;
                if (qj <= m->q) memcpy(tmp, coef + j * N, N * 8);
                else for (size_t n = 0; n < N; ++n) tmp[n] = barrett64(coef[j * N + n], m);
                ntt_forward(&c->t[ki], N, tmp);
                operand = tmp;
            }
            const u64 *k0 = key + ((j * 2 + 0) * K + ki) * N;
            const u64 *k1 = key + ((j * 2 + 1) * K + ki) * N;
            for (size_t n = 0; n < N; ++n) { /* lazy 128-bit accumulation (<= 62 summands of < 2^120) */
                acc[n] += (u128)operand[n] * k0[n];
                acc[N + n] += (u128)operand[n] * k1[n];
            }
        }
        for (size_t k = 0; k < 2; ++k)
            for (size_t n = 0; n < N; ++n) prod[(k * (L + 1) + ii) * N + n] = barrett128(acc[k * N + n], m);
    }

    /* mod-down by the special prime P, with rounding */
    const ho_mod *mp = &c->t[SP].m;
    const u64 P = mp->q, half = P >> 1;
    for (size_t k = 0; k < 2; ++k) {
        u64 *last = prod + (k * (L + 1) + L) * N;
        ntt_inverse(&c->t[SP], N, last);
        for (size_t n = 0; n < N; ++n) last[n] = barrett64(last[n] + half, mp);
        for (size_t i = 0; i < L; ++i) {
            const ho_mod *m = &c->t[i].m;
            const u64 qi = m->q;
            const u64 half_i = barrett64(half, m);
            for (size_t n = 0; n < N; ++n) tmp[n] = submod(barrett64(last[n], m), half_i, qi);
            u64 *pi = prod + (k * (L + 1) + i) * N;
            if (ckks) ntt_forward(&c->t[i], N, tmp);
            else ntt_inverse(&c->t[i], N, pi);
            const u64 pinv = invmod_prime(barrett64(P, m), m); /* modswitch_factors */
            u64 *dst = ct + (k * L + i) * N;
            for (size_t n = 0; n < N; ++n) {
                u64 d = mulmod(submod(pi[n], tmp[n], qi), pinv, m);
                dst[n] = addmod(dst[n], d, qi);
            }
        }
    }
    scr_free(coef); scr_free(tmp); scr_free(prod); scr_free(acc);
}

/* evaluator.cpp Evaluator::relinearize_internal (size 3 -> 2): switch_key(c2, relin_keys[0]) */
void ho_relinearize(const ho_ctx *c, size_t L, u64 *ct3, const u64 *relin_key)
{
    ho_switch_key(c, L, ct3 + 2 * L * c->N, relin_key, ct3);
}

/* evaluator.cpp Evaluator::rescale_to_next -> util/rns.cpp RNSTool::divide_and_round_q_last_ntt_inplace.
 * Reference: ckks matmultval .cpp:255, seal_context.cpp:391,448 */
void ho_rescale(const ho_ctx *c, size_t L, size_t size, const u64 *in, u64 *out)
{
    const size_t N = c->N, last = L - 1;
    const ho_mod *ml = &c->t[last].m;
    const u64 half = ml->q >> 1;
    u64 *r = (u64 *)scr_alloc(N * 8), *tmp = (u64 *)scr_alloc(N * 8);
    for (size_t k = 0; k < size; ++k) {
        memcpy(r, in + (k * L + last) * N, N * 8);
        ntt_inverse(&c->t[last], N, r);
        for (size_t n = 0; n < N; ++n) r[n] = addmod(r[n], half, ml->q);
        for (size_t i = 0; i < last; ++i) {
            const ho_mod *m = &c->t[i].m;
            const u64 qi = m->q, half_i = barrett64(half, m);
            for (size_t n = 0; n < N; ++n) tmp[n] = submod(barrett64(r[n], m), half_i, qi);
            ntt_forward(&c->t[i], N, tmp);
            const u64 inv = invmod_prime(barrett64(ml->q, m), m); /* inv_q_last_mod_q */
            const u64 *src = in + (k * L + i) * N;
            u64 *dst = out + (k * last + i) * N;
            for (size_t n = 0; n < N; ++n) dst[n] = mulmod(submod(src[n], tmp[n], qi), inv, m);
        }
    }
    scr_free(r); scr_free(tmp);
}

/* util/rns.cpp RNSTool::divide_and_round_q_last_inplace (coefficient form): BFV mod_switch_to_next and
 * the key-level -> data-level step of Encryptor::encrypt */
void ho_mod_switch_coeff(const ho_ctx *c, size_t L, size_t size, const u64 *in, u64 *out)
{
    const size_t N = c->N, last = L - 1;
    const ho_mod *ml = &c->t[last].m;
    const u64 half = ml->q >> 1;
    u64 *r = (u64 *)scr_alloc(N * 8);
    for (size_t k = 0; k < size; ++k) {
        const u64 *lp = in + (k * L + last) * N;
        for (size_t n = 0; n < N; ++n) r[n] = addmod(lp[n], half, ml->q);
        for (size_t i = 0; i < last; ++i) {
            const ho_mod *m = &c->t[i].m;
            const u64 qi = m->q, half_i = barrett64(half, m);
            const u64 inv = invmod_prime(barrett64(ml->q, m), m);
            const u64 *src = in + (k * L + i) * N;
            u64 *dst = out + (k * last + i) * N;
            for (size_t n = 0; n < N; ++n) {
                u64 t = submod(barrett64(r[n], m), half_i, qi);
                dst[n] = mulmod(submod(src[n], t, qi), inv, m);
            }
        }
    }
    scr_free(r);
}

/* ------------------------------------------------------------------------------------------------ */
/* Galois  (util/galois.cpp GaloisTool)                                                               */
/* ------------------------------------------------------------------------------------------------ */
/* GaloisTool::get_elt_from_step; generator 3.  Reference: rotate_vector(ct, 1<<i) seal_context.cpp:337 */
uint32_t ho_galois_elt_from_step(const ho_ctx *c, int step)
{
    const uint32_t n = (uint32_t)c->N, m = 2 * n;
    if (step == 0) return m - 1;
    uint32_t pos = (uint32_t)(step < 0 ? -step : step);
    if (pos >= (n >> 1)) return 0; /* SEAL throws "step count too large" */
    pos &= m - 1;
    uint32_t s = step < 0 ? (n >> 1) - pos : pos;
    uint64_t g = 1;
    for (uint32_t i = 0; i < s; ++i) g = (g * 3) & (m - 1);
    return (uint32_t)g;
}
/* GaloisTool::get_elts_all: m-1, then 3^(2^i) and 3^-(2^i) for i < log2(N)-1.  Reference:
 * create_galois_keys() at seal_context.cpp:69 */
size_t ho_galois_elts_all(const ho_ctx *c, uint32_t *out)
{
    const uint32_t m = 2 * (uint32_t)c->N;
    size_t cnt = 0;
    out[cnt++] = m - 1;
    uint64_t pos = 3, neg = 0;
    for (uint64_t x = 1; x < m; x += 2)
        if (((x * 3) & (m - 1)) == 1) { neg = x; break; }
    for (int i = 0; i < c->logn - 1; ++i) {
        out[cnt++] = (uint32_t)pos;
        pos = (pos * pos) & (m - 1);
        out[cnt++] = (uint32_t)neg;
        neg = (neg * neg) & (m - 1);
    }
    return cnt;
}
/* GaloisTool::apply_galois (coefficient form) / apply_galois_ntt (bit-reversed evaluation form) */
void ho_apply_galois_poly(const ho_ctx *c, size_t prime_idx, uint32_t elt, int ntt_form, const u64 *in, u64 *out)
{
    const size_t N = c->N;
    const u64 q = c->t[prime_idx].m.q;
    if (ntt_form) {
        for (size_t i = 0; i < N; ++i) {
            uint32_t rev = bitrev32((uint32_t)(i + N), c->logn + 1);
            uint64_t raw = (((uint64_t)elt * rev) >> 1) & (N - 1);
            out[i] = in[bitrev32((uint32_t)raw, c->logn)];
        }
    } else {
        uint64_t raw = 0;
        for (size_t i = 0; i < N; ++i) {
            size_t idx = raw & (N - 1);
            u64 v = in[i];
            if ((raw >> c->logn) & 1) v = negmod(v, q);
            out[idx] = v;
            raw += elt;
        }
    }
}
/* evaluator.cpp Evaluator::apply_galois_inplace: permute c0,c1; c1 := 0; switch_key(permuted c1) */
void ho_apply_galois(const ho_ctx *c, size_t L, uint32_t elt, const u64 *gkey, const u64 *in, u64 *out)
{
    const size_t N = c->N;
    const int ntt_form = (c->scheme == HO_SCHEME_CKKS);
    u64 *t1 = (u64 *)scr_alloc(L * N * 8);
    for (size_t i = 0; i < L; ++i) {
        ho_apply_galois_poly(c, i, elt, ntt_form, in + i * N, out + i * N);
        ho_apply_galois_poly(c, i, elt, ntt_form, in + (L + i) * N, t1 + i * N);
    }
    memset(out + L * N, 0, L * N * 8);
    ho_switch_key(c, L, t1, gkey, out);
    scr_free(t1);
}

/* ------------------------------------------------------------------------------------------------ */
/* Batched pipelines with the reference's OpenMP loop shape                                           */
/* ------------------------------------------------------------------------------------------------ */
int ho_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ho_batch_op(const ho_ctx *c, int op, size_t L, size_t n_results, const u64 *a, const uint32_t *idx_a, const u64 *b,
                 const uint32_t *idx_b, const u64 *relin_key, u64 *out, int threads)
{
    const size_t N = c->N, ct = 2 * L * N;
    size_t ostride = ct;
    if (op == HO_OP_MUL) ostride = 3 * L * N;
    if (op == HO_OP_MUL_RELIN_RESCALE) ostride = 2 * (L - 1) * N;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
    /* same loop shape as `#pragma omp parallel for collapse(2)` at ckks eltwise .cpp:325 once the
     * (i, x) pair is flattened into r */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (size_t r = 0; r < n_results; ++r) {
        const u64 *pa = a + (size_t)idx_a[r] * ct, *pb = b + (size_t)idx_b[r] * ct;
        u64 *po = out + r * ostride;
        if (op == HO_OP_ADD) {
            ho_add(c, L, 2, pa, pb, po);
        } else if (op == HO_OP_MUL) {
            ho_multiply_ntt(c, L, pa, pb, po);
        } else {
            u64 *t3 = (u64 *)scr_alloc(3 * L * N * 8);
            ho_multiply_ntt(c, L, pa, pb, t3);
            ho_relinearize(c, L, t3, relin_key);
            if (op == HO_OP_MUL_RELIN) memcpy(po, t3, ct * 8);
            else ho_rescale(c, L, 2, t3, po);
            scr_free(t3);
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Sampling, keys, encryption (own PRNG: xoshiro256**; distributions as util/rlwe.cpp)               */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { u64 s[4]; } rng_t;
static u64 splitmix(u64 *x) { u64 z = (*x += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static void rng_seed(rng_t *r, u64 seed) { for (int i = 0; i < 4; ++i) r->s[i] = splitmix(&seed); }
static inline u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
static u64 rng_next(rng_t *r)
{
    u64 *s = r->s, res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return res;
}
/* rlwe.cpp sample_poly_uniform: rejection sampling below the largest multiple of q */
static void sample_uniform(rng_t *r, const ho_ctx *c, size_t nmod, u64 *out)
{
    for (size_t i = 0; i < nmod; ++i) {
        const u64 q = c->t[i].m.q, lim = UINT64_MAX - (UINT64_MAX % q) - 1; /* accept v <= lim */
        for (size_t n = 0; n < c->N; ++n) {
            u64 v;
            do v = rng_next(r); while (v > lim);
            out[i * c->N + n] = v % q;
        }
    }
}
/* rlwe.cpp sample_poly_ternary: uniform in {-1,0,1} */
static void sample_ternary(rng_t *r, const ho_ctx *c, size_t nmod, u64 *out)
{
    for (size_t n = 0; n < c->N; ++n) {
        u64 v;
        do v = rng_next(r) & 3; while (v == 3);
        for (size_t i = 0; i < nmod; ++i) {
            const u64 q = c->t[i].m.q;
            out[i * c->N + n] = v == 0 ? q - 1 : v - 1; /* 0 -> -1, 1 -> 0, 2 -> 1 */
        }
    }
}
/* rlwe.cpp sample_poly_cbd: centred binomial, 21 - 21 bits (sigma ~ 3.24) */
static void sample_cbd(rng_t *r, const ho_ctx *c, size_t nmod, u64 *out)
{
    for (size_t n = 0; n < c->N; ++n) {
        u64 v = rng_next(r);
        int e = __builtin_popcountll(v & 0x1FFFFF) - __builtin_popcountll((v >> 21) & 0x1FFFFF);
        for (size_t i = 0; i < nmod; ++i) {
            const u64 q = c->t[i].m.q;
            out[i * c->N + n] = e >= 0 ? (u64)e : q - (u64)(-e);
        }
    }
}

void ho_keygen_secret(const ho_ctx *c, u64 seed, u64 *sk)
{
    rng_t r; rng_seed(&r, seed);
    sample_ternary(&r, c, c->K, sk);
    for (size_t i = 0; i < c->K; ++i) ntt_forward(&c->t[i], c->N, sk + i * c->N);
}

/* rlwe.cpp encrypt_zero_symmetric (NTT form, key level): (b, a) with b = -(a s + e) */
static void enc_zero_sym(const ho_ctx *c, const u64 *sk, rng_t *r, u64 *out)
{
    const size_t N = c->N, K = c->K;
    u64 *b = out, *a = out + K * N;
    u64 *e = (u64 *)malloc(K * N * 8);
    sample_uniform(r, c, K, a);
    sample_cbd(r, c, K, e);
    for (size_t i = 0; i < K; ++i) {
        const ho_mod *m = &c->t[i].m;
        ntt_forward(&c->t[i], N, e + i * N);
        for (size_t n = 0; n < N; ++n) {
            u64 v = addmod(mulmod(a[i * N + n], sk[i * N + n], m), e[i * N + n], m->q);
            b[i * N + n] = negmod(v, m->q);
        }
    }
    free(e);
}
void ho_keygen_public(const ho_ctx *c, const u64 *sk, u64 seed, u64 *pk)
{
    rng_t r; rng_seed(&r, seed);
    enc_zero_sym(c, sk, &r, pk);
}
/* keygenerator.cpp generate_one_kswitch_key: digit j = Enc_sym(0) with (P mod q_j)*new_key added to
 * residue j of the first polynomial.  Reference: create_relin_keys / create_galois_keys at
 * seal_context.cpp:53,69 */
void ho_keygen_kswitch(const ho_ctx *c, const u64 *sk, const u64 *new_key, u64 seed, u64 *out)
{
    const size_t N = c->N, K = c->K, Ld = c->Ltop;
    rng_t r; rng_seed(&r, seed);
    const u64 P = c->t[K - 1].m.q;
    for (size_t j = 0; j < Ld; ++j) {
        u64 *dig = out + j * 2 * K * N;
        enc_zero_sym(c, sk, &r, dig);
        const ho_mod *m = &c->t[j].m;
        const u64 f = barrett64(P, m);
        for (size_t n = 0; n < N; ++n)
            dig[j * N + n] = addmod(dig[j * N + n], mulmod(new_key[j * N + n], f, m), m->q);
    }
}
void ho_keygen_relin(const ho_ctx *c, const u64 *sk, u64 seed, u64 *out)
{
    const size_t N = c->N, K = c->K;
    u64 *s2 = (u64 *)malloc(K * N * 8);
    for (size_t i = 0; i < K; ++i)
        for (size_t n = 0; n < N; ++n) s2[i * N + n] = mulmod(sk[i * N + n], sk[i * N + n], &c->t[i].m);
    ho_keygen_kswitch(c, sk, s2, seed, out);
    free(s2);
}
void ho_keygen_galois(const ho_ctx *c, const u64 *sk, uint32_t elt, u64 seed, u64 *out)
{
    const size_t N = c->N, K = c->K;
    u64 *rs = (u64 *)malloc(K * N * 8);
    for (size_t i = 0; i < K; ++i) ho_apply_galois_poly(c, i, elt, 1, sk + i * N, rs + i * N);
    ho_keygen_kswitch(c, sk, rs, seed, out);
    free(rs);
}

/* encryptor.cpp Encryptor::encrypt_internal (asymmetric): encrypt_zero_asymmetric at the key level
 * (u ternary, e0,e1 CBD), divide-and-round by the special prime, then add the plaintext.
 * Reference: ckks eltwise .cpp:242, bfv eltwise .cpp:233 */
static void small_to_rns(const ho_ctx *c, size_t nmod, const int32_t *small, u64 *out)
{
    for (size_t i = 0; i < nmod; ++i) {
        const u64 q = c->t[i].m.q;
        for (size_t n = 0; n < c->N; ++n) out[i * c->N + n] = small[n] >= 0 ? (u64)small[n] : q - (u64)(-small[n]);
    }
}
static void encrypt_core(const ho_ctx *c, const u64 *pk, const u64 *plain, u64 *u, u64 *e01, u64 *out);
void ho_encrypt(const ho_ctx *c, const u64 *pk, const u64 *plain, u64 seed, u64 *out)
{
    const size_t N = c->N, K = c->K;
    rng_t r; rng_seed(&r, seed);
    u64 *u = (u64 *)malloc(K * N * 8), *e = (u64 *)malloc(2 * K * N * 8);
    sample_ternary(&r, c, K, u);
    sample_cbd(&r, c, K, e);
    sample_cbd(&r, c, K, e + K * N);
    encrypt_core(c, pk, plain, u, e, out);
    free(u); free(e);
}
/* the same encryption with the three sampled polynomials given (coefficients in {-1,0,1} / small integers): lets a test
 * drive it with exactly the randomness another implementation used */
void ho_encrypt_explicit(const ho_ctx *c, const u64 *pk, const u64 *plain, const int32_t *u_small, const int32_t *e0_small,
                         const int32_t *e1_small, u64 *out)
{
    const size_t N = c->N, K = c->K;
    u64 *u = (u64 *)malloc(K * N * 8), *e = (u64 *)malloc(2 * K * N * 8);
    small_to_rns(c, K, u_small, u);
    small_to_rns(c, K, e0_small, e);
    small_to_rns(c, K, e1_small, e + K * N);
    encrypt_core(c, pk, plain, u, e, out);
    free(u); free(e);
}
/* u: [K][N] coefficient form (transformed in place); e01: [2][K][N] coefficient form */
static void encrypt_core(const ho_ctx *c, const u64 *pk, const u64 *plain, u64 *u, u64 *e01, u64 *out)
{
    const size_t N = c->N, K = c->K, L = c->Ltop;
    const int ckks = c->scheme == HO_SCHEME_CKKS;
    u64 *z = (u64 *)malloc(2 * K * N * 8);
    for (size_t i = 0; i < K; ++i) ntt_forward(&c->t[i], N, u + i * N);
    for (size_t k = 0; k < 2; ++k) {
        u64 *e = e01 + k * K * N;
        for (size_t i = 0; i < K; ++i) {
            const ho_mod *m = &c->t[i].m;
            u64 *zi = z + (k * K + i) * N;
            for (size_t n = 0; n < N; ++n) zi[n] = mulmod(u[i * N + n], pk[(k * K + i) * N + n], m);
            if (ckks) {
                ntt_forward(&c->t[i], N, e + i * N);
            } else {
                ntt_inverse(&c->t[i], N, zi);
            }
            for (size_t n = 0; n < N; ++n) zi[n] = addmod(zi[n], e[i * N + n], m->q);
        }
    }
    if (K > 1) {
        if (ckks) ho_rescale(c, K, 2, z, out);       /* divide_and_round_q_last_ntt_inplace */
        else ho_mod_switch_coeff(c, K, 2, z, out);   /* divide_and_round_q_last_inplace     */
    } else {
        memcpy(out, z, 2 * N * 8);
    }
    if (ckks) {
        for (size_t i = 0; i < L; ++i)
            for (size_t n = 0; n < N; ++n) out[i * N + n] = addmod(out[i * N + n], plain[i * N + n], c->t[i].m.q);
    } else {
        /* util/scalingvariant.cpp multiply_add_plain_with_scaling_variant: c0 += round(q*m/t) */
        const u64 t = c->t_value;
        /* q mod t and floor(q/t) mod q_i from the CRT basis: compute q as a bignum */
        /* (small helper: product of L primes mod t, and floor(q/t) mod q_i via q_i-residue of (q - q mod t)/t) */
        u64 q_mod_t = 1;
        for (size_t i = 0; i < L; ++i) q_mod_t = (u64)(((u128)q_mod_t * (c->t[i].m.q % t)) % t);
        const u64 thr = (t + 1) >> 1;
        for (size_t i = 0; i < L; ++i) {
            const ho_mod *m = &c->t[i].m;
            /* floor(q/t) mod q_i = (q - q_mod_t) * t^{-1} mod q_i = (-q_mod_t) * t^{-1} mod q_i */
            u64 tinv = invmod_prime(barrett64(t, m), m);
            u64 qdivt = mulmod(negmod(barrett64(q_mod_t, m), m->q), tinv, m);
            for (size_t n = 0; n < N; ++n) {
                u128 num = (u128)plain[n] * q_mod_t + thr;
                u64 fix = (u64)(num / t);
                u64 v = addmod(mulmod(plain[n], qdivt, m), barrett64(fix, m), m->q);
                out[i * N + n] = addmod(out[i * N + n], v, m->q);
            }
        }
    }
    free(z);
}

/* decryptor.cpp dot_product_ct_sk_array: c0 + c1 s + c2 s^2 (+...).  Size-3 inputs must work: the
 * reference decrypts un-relinearized products (ckks eltwise .cpp:342-344) */
void ho_decrypt_phase(const ho_ctx *c, size_t L, size_t size, const u64 *ct, const u64 *sk, u64 *out)
{
    const size_t N = c->N;
    const int ckks = c->scheme == HO_SCHEME_CKKS;
    u64 *tmp = (u64 *)malloc(N * 8);
    for (size_t i = 0; i < L; ++i) {
        const ho_mod *m = &c->t[i].m;
        u64 *o = out + i * N;
        /* Horner in s: ((c_{size-1} s + c_{size-2}) s + ...) + c0, in NTT form */
        for (size_t n = 0; n < N; ++n) o[n] = 0;
        for (size_t k = size; k-- > 0;) {
            memcpy(tmp, ct + (k * L + i) * N, N * 8);
            if (!ckks) ntt_forward(&c->t[i], N, tmp);
            for (size_t n = 0; n < N; ++n) {
                u64 v = (k == size - 1) ? 0 : mulmod(o[n], sk[i * N + n], m);
                o[n] = addmod(v, tmp[n], m->q);
            }
        }
        if (!ckks) ntt_inverse(&c->t[i], N, o);
    }
    free(tmp);
}

/* ---- tiny multiword helpers for CRT composition (words little-endian) ---------------------------- */
static void mw_mul_small(u64 *a, size_t w, u64 b) /* a *= b in place, a has w words (no overflow by construction) */
{
    u64 carry = 0;
    for (size_t i = 0; i < w; ++i) {
        u128 p = (u128)a[i] * b + carry;
        a[i] = (u64)p;
        carry = (u64)(p >> 64);
    }
}
static void mw_addmul(u64 *acc, const u64 *a, size_t w, u64 b) /* acc += a*b */
{
    u64 carry = 0;
    for (size_t i = 0; i < w; ++i) {
        u128 p = (u128)a[i] * b + acc[i] + carry;
        acc[i] = (u64)p;
        carry = (u64)(p >> 64);
    }
}
static int mw_cmp(const u64 *a, const u64 *b, size_t w)
{
    for (size_t i = w; i-- > 0;) {
        if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
    }
    return 0;
}
static void mw_sub(u64 *a, const u64 *b, size_t w)
{
    u64 borrow = 0;
    for (size_t i = 0; i < w; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        a[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}
static double mw_to_double(const u64 *a, size_t w)
{
    double r = 0;
    for (size_t i = w; i-- > 0;) r = r * 18446744073709551616.0 + (double)a[i];
    return r;
}
typedef struct { size_t w; u64 *Q, *halfQ, *punct; u64 *inv; } crt_t;
static void crt_init(crt_t *t, const ho_ctx *c, size_t L)
{
    size_t w = L + 1;
    t->w = w;
    t->Q = (u64 *)calloc(w, 8); t->halfQ = (u64 *)calloc(w, 8);
    t->punct = (u64 *)calloc(L * w, 8); t->inv = (u64 *)calloc(L, 8);
    t->Q[0] = 1;
    for (size_t i = 0; i < L; ++i) mw_mul_small(t->Q, w, c->t[i].m.q);
    for (size_t i = 0; i < w; ++i) t->halfQ[i] = (t->Q[i] >> 1) | (i + 1 < w ? t->Q[i + 1] << 63 : 0);
    for (size_t i = 0; i < L; ++i) {
        u64 *p = t->punct + i * w;
        p[0] = 1;
        u64 pm = 1;
        const ho_mod *m = &c->t[i].m;
        for (size_t k = 0; k < L; ++k)
            if (k != i) { mw_mul_small(p, w, c->t[k].m.q); pm = mulmod(pm, barrett64(c->t[k].m.q, m), m); }
        t->inv[i] = invmod_prime(pm, m);
    }
}
static void crt_free(crt_t *t) { free(t->Q); free(t->halfQ); free(t->punct); free(t->inv); }
/* x (w words) = CRT(residues) in [0,Q) */
static void crt_compose(const crt_t *t, const ho_ctx *c, size_t L, const u64 *res, size_t stride, u64 *x)
{
    memset(x, 0, t->w * 8);
    for (size_t i = 0; i < L; ++i) {
        u64 y = mulmod(res[i * stride], t->inv[i], &c->t[i].m);
        mw_addmul(x, t->punct + i * t->w, t->w, y);
    }
    while (mw_cmp(x, t->Q, t->w) >= 0) mw_sub(x, t->Q, t->w);
}

/* ckks.h CKKSEncoder::decode_internal: CRT-compose, centre, scale to double (before the FFT) */
void ho_crt_to_double(const ho_ctx *c, size_t L, const u64 *coeff_poly, double inv_scale, double *out)
{
    crt_t t; crt_init(&t, c, L);
    u64 *x = (u64 *)malloc(t.w * 8), *y = (u64 *)malloc(t.w * 8);
    for (size_t n = 0; n < c->N; ++n) {
        crt_compose(&t, c, L, coeff_poly + n, c->N, x);
        if (mw_cmp(x, t.halfQ, t.w) > 0) {
            memcpy(y, t.Q, t.w * 8);
            mw_sub(y, x, t.w);
            out[n] = -mw_to_double(y, t.w) * inv_scale;
        } else {
            out[n] = mw_to_double(x, t.w) * inv_scale;
        }
    }
    free(x); free(y); crt_free(&t);
}
/* BFV decryption, exact form: m = round(t*x/q) mod t with x the centred phase.  (SEAL computes the same
 * value through the gamma/t base, RNSTool::decrypt_scale_and_round.) */
void ho_bfv_decode_phase(const ho_ctx *c, size_t L, const u64 *phase, u64 *plain)
{
    crt_t t; crt_init(&t, c, L);
    const size_t w = t.w + 1;
    const u64 tv = c->t_value;
    u64 *x = (u64 *)calloc(w, 8), *num = (u64 *)calloc(w, 8), *prod = (u64 *)calloc(w, 8), *Qw = (u64 *)calloc(w, 8);
    memcpy(Qw, t.Q, t.w * 8);
    const double Qd = mw_to_double(t.Q, t.w);
    for (size_t n = 0; n < c->N; ++n) {
        memset(x, 0, w * 8);
        crt_compose(&t, c, L, phase + n, c->N, x);
        /* num = t*x + floor(Q/2) ; m = floor(num / Q) mod t */
        memcpy(num, x, w * 8);
        mw_mul_small(num, w, tv);
        {
            u64 carry = 0;
            for (size_t i = 0; i < w; ++i) {
                u128 s = (u128)num[i] + (i < t.w ? t.halfQ[i] : 0) + carry;
                num[i] = (u64)s; carry = (u64)(s >> 64);
            }
        }
        u64 m = (u64)(mw_to_double(num, w) / Qd);
        if (m > 0) --m;
        /* fix the estimate: largest m with m*Q <= num */
        for (;;) {
            memcpy(prod, Qw, w * 8);
            mw_mul_small(prod, w, m + 1);
            if (mw_cmp(prod, num, w) <= 0) ++m; else break;
        }
        plain[n] = m % tv;
    }
    free(x); free(num); free(prod); free(Qw); crt_free(&t);
}

/* ------------------------------------------------------------------------------------------------ */
/* BFV multiply (BEHZ) lives in he_oracle_bfv.c                                                       */
/* ------------------------------------------------------------------------------------------------ */
#include "he_oracle_bfv.inc"
#include "he_oracle_pipelines.inc"
