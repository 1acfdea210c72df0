/*
 * he_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load or call this library.
 * The product (reference-seal-backend_amd/) never links, loads or calls anything in oracle/.
 *
 * What it is: a plain-C restatement of the arithmetic that hebench/reference-seal-backend reaches through
 * seal::Evaluator (Microsoft SEAL v3.7.2, pinned by /root/reference/cmake/third-party/SEAL.version:1 and
 * fetched, un-vendored, by cmake/third-party/SEAL.cmake:5-12).  Reference call sites restated here:
 *   add                       src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:340
 *   multiply (CKKS)           src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:343
 *   multiply+relinearize      src/benchmarks/ckks/seal_ckks_dot_product_benchmark.cpp:325-329
 *   multiply+relin+rescale    src/benchmarks/ckks/seal_ckks_matmultval_benchmark.cpp:253-255
 *   rotate_vector/accumulate  src/engine/seal_context.cpp:321-347
 *   rotate_rows/columns       src/engine/seal_context.cpp:289-319
 *   parameter rule            src/engine/seal_context.cpp:72-127
 *   keys                      src/engine/seal_context.cpp:46-70
 *
 * PARITY UNPINNED: the reference holds no tests, fixtures or golden vectors, and SEAL
 * itself is absent from this image and cannot be built (no network).  The oracle is pinned only by
 * (i) the prime-chain known answers of SURVEY.md Appendix B (tests/golden/primes.json),
 * (ii) algebraic identities (schoolbook negacyclic convolution, NTT round trips) and
 * (iii) Dec(Eval(Enc(x))) == f(x) with its own key generation.  It has never been compared with SEAL.
 *
 * Data layout everywhere: SEAL's  — a polynomial is L residues of N uint64 coefficients, residue-major;
 * a ciphertext of `size` polynomials is [size][L][N]; NTT form = bit-reversed-order evaluations
 * (Harvey forward transform, natural in / bit-reversed out).
 */
#ifndef HE_ORACLE_H
#define HE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HO_SCHEME_BFV 1
#define HO_SCHEME_CKKS 2

typedef struct ho_ctx ho_ctx;

/* ---- context ------------------------------------------------------------------------------------ */
/* bit_sizes: the key-level chain exactly as seal_context.cpp:79-82 builds it, e.g. {60,45,...,45,60}.
 * sec128 != 0 enforces SEAL's tc128 total-bit cap.  plain_bits: BFV batching prime size (0 for CKKS).
 * Returns NULL on error with a message in err. */
ho_ctx *ho_ctx_create(int scheme, size_t N, const int *bit_sizes, size_t n_bits, int plain_bits, int sec128,
                      char *err, size_t errlen);
/* Same, with caller-chosen NTT-friendly primes (small test rings). plain_modulus may be 0. */
ho_ctx *ho_ctx_create_primes(int scheme, size_t N, const uint64_t *primes, size_t n_primes, uint64_t plain_modulus,
                             char *err, size_t errlen);
void ho_ctx_destroy(ho_ctx *c);
/* The evaluator functions keep their temporaries in a per-thread cache (the role of SEAL's MemoryPoolHandle::ThreadLocal() in the
 * reference's operate() loop, ckks eltwise .cpp:343).  These give the cache of the calling thread / of every OpenMP thread back. */
void ho_scratch_release(void);
void ho_scratch_release_all(void);
size_t ho_N(const ho_ctx *c);
size_t ho_key_mod_count(const ho_ctx *c);  /* K: all primes, special prime last            */
size_t ho_data_mod_count(const ho_ctx *c); /* L at the first data level (K-1, or 1 if K==1) */
uint64_t ho_modulus(const ho_ctx *c, size_t i);
uint64_t ho_plain_modulus(const ho_ctx *c);
uint64_t ho_root(const ho_ctx *c, size_t i); /* minimal primitive 2N-th root of unity mod q_i */
/* root_powers table of prime i (N entries, bit-reversed exponent order), for cross-checks */
void ho_root_powers(const ho_ctx *c, size_t i, uint64_t *out);

/* ---- number theory ------------------------------------------------------------------------------ */
int ho_is_prime(uint64_t v);
/* SEAL get_primes(factor, bit_size, count): descending primes = 1 mod factor. Returns number found. */
size_t ho_get_primes(uint64_t factor, int bit_size, size_t count, uint64_t *out);

/* ---- transforms (one residue polynomial, in place, canonical [0,q) output) ---------------------- */
void ho_ntt(const ho_ctx *c, size_t prime_idx, uint64_t *poly);
void ho_intt(const ho_ctx *c, size_t prime_idx, uint64_t *poly);

/* ---- evaluator on raw arrays.  L = residues at the operand's level (the first L primes). -------- */
void ho_add(const ho_ctx *c, size_t L, size_t size, const uint64_t *a, const uint64_t *b, uint64_t *out);
void ho_sub(const ho_ctx *c, size_t L, size_t size, const uint64_t *a, const uint64_t *b, uint64_t *out);
/* CKKS (and any NTT-form) dyadic tensor: a,b size 2 -> out size 3 */
void ho_multiply_ntt(const ho_ctx *c, size_t L, const uint64_t *a, const uint64_t *b, uint64_t *out);
/* Evaluator::multiply_plain (NTT-form plaintext [L][N]; every polynomial of ct is multiplied dyadically; SEAL
 * multiply_plain_ntt; called from seal_context.cpp:390) and Evaluator::add_plain (CKKS: plain added to c0;
 * seal_context.cpp:454).  ct: [size][L][N]. */
void ho_multiply_plain(const ho_ctx *c, size_t L, size_t size, const uint64_t *ct, const uint64_t *plain, uint64_t *out);
void ho_add_plain(const ho_ctx *c, size_t L, size_t size, const uint64_t *ct, const uint64_t *plain, uint64_t *out);
/* CKKS Evaluator::mod_switch_to_next / mod_switch_to (ciphertexts and NTT-form plaintexts): the last residues are
 * dropped, nothing is scaled (SEAL mod_switch_drop_to_next; seal_context.cpp:389,451 and matchLevel).
 * in [size][L][N] -> out [size][L_to][N] */
void ho_mod_switch_drop(const ho_ctx *c, size_t L, size_t L_to, size_t size, const uint64_t *in, uint64_t *out);
/* Key switching (SEAL Evaluator::switch_key_inplace). target: [L][N] (CKKS: NTT form, BFV: coefficient
 * form).  key: [L_top digits][2][K][N] NTT form.  ct (size 2, [2][L][N]) gets the result ADDED in. */
void ho_switch_key(const ho_ctx *c, size_t L, const uint64_t *target, const uint64_t *key, uint64_t *ct);
/* relinearize size 3 -> 2 in place (ct3: [3][L][N], result in first two polys) */
void ho_relinearize(const ho_ctx *c, size_t L, uint64_t *ct3, const uint64_t *relin_key);
/* CKKS rescale_to_next: in [size][L][N] -> out [size][L-1][N] */
void ho_rescale(const ho_ctx *c, size_t L, size_t size, const uint64_t *in, uint64_t *out);
/* BFV/any mod_switch_to_next for coefficient-form data (divide_and_round_q_last_inplace) */
void ho_mod_switch_coeff(const ho_ctx *c, size_t L, size_t size, const uint64_t *in, uint64_t *out);
/* Galois */
uint32_t ho_galois_elt_from_step(const ho_ctx *c, int step);
size_t ho_galois_elts_all(const ho_ctx *c, uint32_t *out); /* default key set, returns count */
void ho_apply_galois_poly(const ho_ctx *c, size_t prime_idx, uint32_t elt, int ntt_form, const uint64_t *in, uint64_t *out);
/* apply_galois_inplace on a size-2 ciphertext with the key of that element (scheme decides the form) */
void ho_apply_galois(const ho_ctx *c, size_t L, uint32_t elt, const uint64_t *gkey, const uint64_t *in, uint64_t *out);
/* BFV ct x ct multiply (BEHZ), coefficient form, a,b size 2 -> out size 3 */
void ho_bfv_multiply(const ho_ctx *c, size_t L, const uint64_t *a, const uint64_t *b, uint64_t *out);

/* ---- the reference's per-pair pipelines, batched with OpenMP exactly like operate() ------------- */
/* op codes */
#define HO_OP_ADD 0            /* ckks/bfv eltwise .cpp:340/:322 */
#define HO_OP_MUL 1            /* ckks eltwise .cpp:343 (size-3 result)                        */
#define HO_OP_MUL_RELIN 2      /* ckks dot .cpp:325-329 without accumulate                     */
#define HO_OP_MUL_RELIN_RESCALE 3 /* ckks matmultval .cpp:253-255                              */
/* result r uses a[idx_a[r]] and b[idx_b[r]]; slabs are arrays of size-2 level-L ciphertexts.
 * out stride: ADD 2*L*N, MUL 3*L*N, MUL_RELIN 2*L*N, MUL_RELIN_RESCALE 2*(L-1)*N. threads<=0: all. */
void ho_batch_op(const ho_ctx *c, int op, size_t L, size_t n_results, const uint64_t *a, const uint32_t *idx_a,
                 const uint64_t *b, const uint32_t *idx_b, const uint64_t *relin_key, uint64_t *out, int threads);
int ho_max_threads(void);
#define HO_OP_DOT 4            /* ckks/bfv dot .cpp: multiply -> relinearize_inplace -> accumulate(count)        */
/* Evaluator::rotate_internal with the NAF decomposition (util::naf) over the Galois keys given as (element, key) arrays */
int ho_rotate(const ho_ctx *c, size_t L, int step, size_t n_gk, const uint32_t *gk_elts, const uint64_t *const *gk_keys, const uint64_t *in, uint64_t *out);
/* SEALContextWrapper::accumulateCKKS / accumulateBFV, count > 0, in place on a size-2 ciphertext */
int ho_accumulate(const ho_ctx *c, size_t L, size_t count, size_t n_gk, const uint32_t *gk_elts, const uint64_t *const *gk_keys, uint64_t *ct);
/* operate() of the element-wise / dot-product workloads: `omp parallel for collapse(2)` over (i < b0, x < b1), result i*b1+x */
int ho_batch_outer(const ho_ctx *c, int op, size_t L, size_t b0, size_t b1, const uint64_t *a, const uint64_t *b, const uint64_t *relin_key,
                   size_t n_gk, const uint32_t *gk_elts, const uint64_t *const *gk_keys, size_t count, uint64_t *out, int threads);
/* BFV MatMultRow: out[i] = sum_j rotate_rows(relin(A[i] * B), j * (N/2)/dim2), two-level parallel region as the reference's */
int ho_bfv_matmul_rows(const ho_ctx *c, size_t L, size_t n_cts, const uint64_t *A, const uint64_t *B, const uint64_t *relin_key, size_t n_gk,
                       const uint32_t *gk_elts, const uint64_t *const *gk_keys, size_t dim2, uint64_t *out, int threads);

/* ---- keys / encryption (own sampling; distribution as SEAL: ternary secret, CBD(21) error) ------- */
/* secret key: [K][N] NTT form */
void ho_keygen_secret(const ho_ctx *c, uint64_t seed, uint64_t *sk);
/* public key: [2][K][N] NTT form */
void ho_keygen_public(const ho_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *pk);
/* key-switch key for new_key ([K][N] NTT): out [L_top][2][K][N] */
void ho_keygen_kswitch(const ho_ctx *c, const uint64_t *sk, const uint64_t *new_key, uint64_t seed, uint64_t *out);
void ho_keygen_relin(const ho_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *out);
void ho_keygen_galois(const ho_ctx *c, const uint64_t *sk, uint32_t elt, uint64_t seed, uint64_t *out);
/* Encrypt a plaintext at the first data level: CKKS plain = [L][N] NTT residues; BFV plain = [N] values mod t.
 * out: [2][L][N] (CKKS NTT form, BFV coefficient form). Follows Encryptor: encrypt zero at key level, then
 * divide-and-round by the special prime. */
void ho_encrypt(const ho_ctx *c, const uint64_t *pk, const uint64_t *plain, uint64_t seed, uint64_t *out);
/* same, with the sampled polynomials given as small signed coefficients (u in {-1,0,1}; e0, e1 centred binomial) */
void ho_encrypt_explicit(const ho_ctx *c, const uint64_t *pk, const uint64_t *plain, const int32_t *u_small, const int32_t *e0_small,
                         const int32_t *e1_small, uint64_t *out);
/* phase = c0 + c1 s + c2 s^2 ... : out [L][N]; CKKS: NTT form, BFV: coefficient form */
void ho_decrypt_phase(const ho_ctx *c, size_t L, size_t size, const uint64_t *ct, const uint64_t *sk, uint64_t *out);
/* BFV: phase (coefficient form, level L) -> plaintext values mod t, exact round(t*x/q) via CRT */
void ho_bfv_decode_phase(const ho_ctx *c, size_t L, const uint64_t *phase, uint64_t *plain);
/* CRT-compose coefficient-form residues to centred values / scale as doubles (CKKS decode helper) */
void ho_crt_to_double(const ho_ctx *c, size_t L, const uint64_t *coeff_poly, double inv_scale, double *out);

#ifdef __cplusplus
}
#endif
#endif
