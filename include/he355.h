/*
 * he355.h — C ABI of the MI355X hot path ("thin extern C FFI" of the backend).
 *
 * These are the entry points a host in any language binds (cgo / JNI / ctypes / the C++ HEBench classes in
 * reference-seal-backend_amd/csrc/bridge) in place of the seal::Evaluator calls the reference makes inside its
 * timed operate() bodies.  Plain pointers and sizes only; no C++ or torch types.  All `d_*` pointers are
 * device (HBM) pointers obtained from he355_malloc; ciphertext slabs are arrays of SEAL-layout ciphertexts
 * [n][size][L][N] of uint64 residues (CKKS: NTT form), i.e. byte-compatible with seal::Ciphertext::data().
 *
 * Reference interface replaced (file:line under /root/reference):
 *   he355_ctx_create            SEALContextWrapper::createCKKSContext/createBFVContext  include/engine/seal_context.h:32-50,
 *                               parameter rule src/engine/seal_context.cpp:79-90,107-119
 *   he355_set_relin_key         KeyGenerator::create_relin_keys                         src/engine/seal_context.cpp:53
 *   he355_set_galois_key        KeyGenerator::create_galois_keys                        src/engine/seal_context.cpp:69
 *   he355_add                   evaluator()->add          src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:340,
 *                                                         src/benchmarks/bfv/seal_bfv_element_wise_benchmark.cpp:322
 *   he355_multiply              evaluator()->multiply     src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:343
 *   he355_multiply_relin        multiply + relinearize_inplace                src/benchmarks/ckks/seal_ckks_dot_product_benchmark.cpp:325-329
 *     (rescale = 1)             ... + rescale_to_next_inplace                 src/benchmarks/ckks/seal_ckks_matmultval_benchmark.cpp:253-255
 *   he355_relinearize           evaluator()->relinearize_inplace              src/engine/seal_context.cpp:390,447
 *   he355_multiply_accumulate   multiply + add_inplace over the inner dimension  src/benchmarks/ckks/seal_ckks_matmult_cipherbatchaxis_benchmark.cpp:404-420
 *   he355_relinearize_rescale   relinearize_inplace + rescale_to_next_inplace    same file :436-437
 *   he355_bfv_multiply_relin_accumulate  multiply + relinearize_inplace + add_inplace over the inner dimension
 *                                                         src/benchmarks/bfv/seal_bfv_matmult_cipherbatchaxis_benchmark.cpp:398-410
 *   he355_rescale               evaluator()->rescale_to_next_inplace          src/engine/seal_context.cpp:391,448
 *   he355_rotate                evaluator()->rotate_vector (CKKS) / rotate_rows (BFV)      src/engine/seal_context.cpp:337,302
 *   he355_apply_galois(2N-1)    evaluator()->rotate_columns_inplace (BFV)                   src/engine/seal_context.cpp:308
 *   he355_accumulate            SEALContextWrapper::accumulateCKKS / accumulateBFV          src/engine/seal_context.cpp:321-347,289-319
 *   the Indexer                 ParameterIndexer{value_index,batch_size} and the result order r = i*b1 + x
 *                                                         src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:322-336
 *
 * Error convention: every function returns 0 on success or a non-zero code (HE355_E_*); the message is kept
 * per thread and read with he355_last_error() — the same shape as the API Bridge's ErrorCode +
 * getLastErrorDescription.  There is NO CPU fallback: without a HIP device every device call fails with
 * HE355_E_DEVICE.
 */
#ifndef HE355_H
#define HE355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HE355_SCHEME_BFV 1
#define HE355_SCHEME_CKKS 2

#define HE355_OK 0
#define HE355_E_INVALID_ARGS 1 /* same meaning as HEBENCH_ECODE_INVALID_ARGS   */
#define HE355_E_PARAMS 2       /* parameter/context error; the reference reports these as HEBSEAL_ECODE_SEAL_ERROR = 2 */
#define HE355_E_DEVICE 3       /* HIP error or no device                        */
#define HE355_E_CRITICAL 0x7FFFFFFF

typedef struct he355_ctx he355_ctx;

/* operand selection for result r of a batch (see the Indexer row above) */
typedef struct {
    uint64_t a_base, b_base; /* value_index of operand 0 / operand 1                  */
    uint64_t b1;             /* batch size of operand 1: a = a_base + r / b1, b = b_base + r % b1 */
    int32_t pairwise;        /* 1: a = a_base + r, b = b_base + r                     */
    int32_t reserved;
} he355_indexer;

const char *he355_last_error(void);

/* ---- context (host only; no HIP call is made until he355_device_init) ---- */
int he355_ctx_create(int scheme, uint64_t poly_modulus_degree, const int32_t *bit_sizes, uint64_t n_bit_sizes, int plain_modulus_bits,
                     int enforce_sec128, he355_ctx **out);
int he355_ctx_create_primes(int scheme, uint64_t poly_modulus_degree, const uint64_t *primes, uint64_t n_primes, uint64_t plain_modulus,
                            he355_ctx **out);
void he355_ctx_destroy(he355_ctx *ctx);
uint64_t he355_poly_degree(const he355_ctx *ctx);
uint64_t he355_key_modulus_count(const he355_ctx *ctx);  /* K, special prime last       */
uint64_t he355_data_modulus_count(const he355_ctx *ctx); /* L at the first data level   */
uint64_t he355_modulus(const he355_ctx *ctx, uint64_t i);
uint64_t he355_plain_modulus(const he355_ctx *ctx);
int he355_prime_uses_fp64(const he355_ctx *ctx, uint64_t i); /* which arithmetic engine owns prime i */
/* BFV: the auxiliary base of the BEHZ multiply at `level` data primes: out[0] = m_sk, out[1..] = B_0.. (at most cap entries written);
 * returns their number, 0 for CKKS or a bad level.  SEAL's RNSTool (seal/util/rns.cpp, RNSTool::initialize) takes 61-bit primes; the
 * product does not depend on the choice and the device takes primes of its fp64 engine (csrc/he_params.h, Params::aux). */
uint64_t he355_bfv_aux_base(const he355_ctx *ctx, int level, uint64_t *out, uint64_t cap);
uint32_t he355_galois_elt_from_step(const he355_ctx *ctx, int step);
uint64_t he355_galois_elts_all(const he355_ctx *ctx, uint32_t *out, uint64_t cap);

/* ---- device ---- */
int he355_device_count(int *count);
int he355_device_init(he355_ctx *ctx, int device_ordinal); /* uploads tables; creates the stream */
/* Device memory comes from the context's pool (csrc/device_pool.h): size-class free lists over hipMalloc'd blocks, the counterpart
 * of the MemoryPoolHandle::ThreadLocal() the reference hands to the evaluator inside operate()
 * (src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:343).  he355_free returns the block to its list: no HIP call and no
 * synchronisation (a re-issued block is only touched by work queued later on the context's streams), so a steady-state operate()
 * performs no hipMalloc / hipFree.  he355_alloc_stats counts every raw hipMalloc / hipFree the context has made since
 * he355_device_init (its tables, keys and scratch arenas included); he355_pool_trim drains the streams and hipFree()s the cached
 * blocks (also done by itself when the device runs out of memory). */
int he355_malloc(he355_ctx *ctx, uint64_t bytes, void **d_ptr);
/* he355_free takes ONLY pointers this context's he355_malloc returned and has not freed yet: a block freed twice, a pointer of another
 * context or one from hipMalloc is refused with HE355_E_INVALID_ARGS and left untouched (since round 5; before, a foreign pointer was
 * drained and hipFree()d).  A host that allocates device memory by other means must release it by the same means. */
int he355_free(he355_ctx *ctx, void *d_ptr);
typedef struct {
    uint64_t raw_mallocs, raw_frees; /* hipMalloc / hipFree calls            */
    uint64_t pool_hits, pool_misses;  /* he355_malloc served from a list / by a new block */
    uint64_t cached_bytes, live_bytes;
} he355_alloc_stats_t;
int he355_alloc_stats(he355_ctx *ctx, he355_alloc_stats_t *out); /* ctx == NULL: totals over every context of the process (byte fields 0) */
int he355_pool_trim(he355_ctx *ctx, uint64_t *released_bytes);
/* Which shape / schedule the context's key switches took since he355_device_init (or the last call with reset != 0).  The choice is the
 * library's (batch, ring size, level: DESIGN.md 5.2-5.3) and never changes a result bit; tests use the counters to prove that the shape they
 * mean to hold to the oracle is the one that ran (tests/test_gpu_bench_shapes.py).  A "sequence" is the kernel sequence of one chunk of a batch. */
typedef struct {
    uint64_t ks_fused;      /* throughput shape, mod-down finished inside k_k3                          */
    uint64_t ks_unfused;    /* throughput shape, separate floor kernels (small grids)                   */
    uint64_t ks_latency;    /* latency shape (digit-split tiles, batch x N <= 2^17)                     */
    uint64_t ks_lds;        /* ring-in-LDS shape (N <= 8192, few ciphertexts): one workgroup per residue polynomial */
    uint64_t level_sums_in_k3;         /* he355_rotate_sum levels whose sum was formed by k_k3 itself   */
    uint64_t level_sums_by_kernel;     /* ... by k_sum_groups                                            */
    uint64_t level_sum_launches_in_k3; /* fused sequences that carried a level sum (several per level when HE355_CHUNK cuts it) */
    uint64_t reserved;
} he355_path_stats_t;
int he355_path_stats(he355_ctx *ctx, he355_path_stats_t *out, int reset);
int he355_upload(he355_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int he355_download(he355_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
int he355_copy(he355_ctx *ctx, void *d_dst, const void *d_src, uint64_t bytes); /* device to device, on the context's stream */
/* device memory of src_ctx's GPU -> device memory of dst_ctx's GPU (hipMemcpyPeer over xGMI; also valid when both contexts sit on the
 * same GPU).  Both contexts are synchronised first, the copy is complete on return.  Used at load() / store() by the bridge's
 * multi-device path: operand replicas out, result parts back; never inside operate(). */
int he355_copy_peer(he355_ctx *dst_ctx, void *d_dst, he355_ctx *src_ctx, const void *d_src, uint64_t bytes);
int he355_sync(he355_ctx *ctx);
/* synthetic data: fill n_polys residue polynomials with uniform residues, polynomial p using prime
 * prime_of[p % period] (throughput-mode inputs) */
int he355_fill_uniform(he355_ctx *ctx, uint64_t *d_dst, uint64_t n_polys, const uint8_t *prime_of, uint32_t period, uint64_t seed);
/* the same stream from polynomial `first_poly` on: a rank's shard of a batch holds what the whole batch would hold there */
int he355_fill_uniform_at(he355_ctx *ctx, uint64_t *d_dst, uint64_t n_polys, const uint8_t *prime_of, uint32_t period, uint64_t seed, uint64_t first_poly);

/* ---- evaluation keys: host arrays [L_top digits][2][K][N], NTT form (SEAL KSwitchKeys layout) ---- */
int he355_set_relin_key(he355_ctx *ctx, const uint64_t *h_key);
int he355_set_galois_key(he355_ctx *ctx, uint32_t galois_elt, const uint64_t *h_key);
/* KeyGenerator on the device (create_relin_keys / create_galois_keys, src/engine/seal_context.cpp:53,69) from the secret key given
 * to he355_set_secret_key; randomness: counter-based streams of `seed` (csrc/client/sampler.h keygen_stream) — the host client's
 * make_relin_key / make_galois_key with the same seed give the same bits */
int he355_keygen_relin(he355_ctx *ctx, uint64_t seed);
int he355_keygen_galois(he355_ctx *ctx, uint32_t galois_elt, uint64_t seed);
int he355_set_relin_key_synthetic(he355_ctx *ctx, uint64_t seed);                       /* uniform residues, generated in HBM */
int he355_set_galois_key_synthetic(he355_ctx *ctx, uint32_t galois_elt, uint64_t seed);

/* ---- batched evaluator ops on device slabs; L = residues at the operands' level ---- */
int he355_add(he355_ctx *ctx, int L, int size, uint64_t n, const uint64_t *d_a, const uint64_t *d_b, he355_indexer ix, uint64_t *d_out);
int he355_sub(he355_ctx *ctx, int L, int size, uint64_t n, const uint64_t *d_a, const uint64_t *d_b, he355_indexer ix, uint64_t *d_out);
/* CKKS multiply: [.][2][L][N] x [.][2][L][N] -> [n][3][L][N] */
int he355_multiply(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_a, const uint64_t *d_b, he355_indexer ix, uint64_t *d_out);
/* BFV multiply (BEHZ, coefficient form): [.][2][L][N] x [.][2][L][N] -> [n][3][L][N]
 * (evaluator()->multiply, src/benchmarks/bfv/seal_bfv_element_wise_benchmark.cpp:325, seal_bfv_dot_product_benchmark.cpp:311).
 * In an outer-product batch every operand serves several results: each is extended to the auxiliary base and transformed once
 * (the values SEAL recomputes per pair), a result then costs its dyadic tensor, three inverse transforms and the floor.
 * d_out may not overlap an operand (HE355_E_INVALID_ARGS). */
int he355_bfv_multiply(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_a, const uint64_t *d_b, he355_indexer ix, uint64_t *d_out);
/* multiply -> relinearize (-> rescale): out [n][2][L][N] or [n][2][L-1][N].  Give d_out a slab of its own: then the tensor product's
 * c0, c1 never travel through HBM (the key-switch kernel forms them from the operand rows); a d_out that overlaps an operand is
 * detected and served by the slower path that materialises them first. */
int he355_multiply_relin(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_a, const uint64_t *d_b, he355_indexer ix, int rescale,
                         uint64_t *d_out);
int he355_relinearize(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_ct3, uint64_t *d_out);            /* [n][3][L][N] -> [n][2][L][N] */
/* CKKS plaintext operands: NTT-form plaintexts [.][L][N] at the ciphertexts' level; ciphertext r uses plaintext
 * b_base + r % b1 (or b_base + r when pairwise), ciphertext operand a_base + r / b1 (the Indexer rule).
 *   he355_multiply_plain: evaluator()->multiply_plain_inplace   src/engine/seal_context.cpp:390   (every polynomial x plain)
 *   he355_add_plain     : evaluator()->add_plain_inplace        src/engine/seal_context.cpp:454   (c0 += plain)        */
int he355_multiply_plain(he355_ctx *ctx, int L, int size, uint64_t n, const uint64_t *d_ct, const uint64_t *d_plain, he355_indexer ix, uint64_t *d_out);
int he355_add_plain(he355_ctx *ctx, int L, int size, uint64_t n, const uint64_t *d_ct, const uint64_t *d_plain, he355_indexer ix, uint64_t *d_out);
/* CKKS evaluator()->mod_switch_to_inplace / mod_switch_to_next_inplace (src/engine/seal_context.cpp:389,451 and matchLevel):
 * the last residues of every polynomial are dropped; in [n_polys][L][N] -> out [n_polys][L_to][N] (n_polys = n * size for
 * ciphertexts, n for plaintexts) */
int he355_mod_switch_drop(he355_ctx *ctx, int L, int L_to, uint64_t n_polys, const uint64_t *d_in, uint64_t *d_out);
/* out = in[0] + ... + in[n-1] (one ciphertext): the add_inplace accumulation of collapseCKKS, src/engine/seal_context.cpp:401 */
int he355_sum(he355_ctx *ctx, int L, int size, uint64_t n, const uint64_t *d_in, uint64_t *d_out);
/* CKKS: out(i,j) = sum_k multiply(a(i,k), b(k,j)), size-3 results [rows*cols][3][L][N]; ciphertext (i,k) of a is at index
 * i*a_stride_i + k*a_stride_k, (k,j) of b at k*b_stride_k + j*b_stride_j — the multiply/add_inplace loop of
 * src/benchmarks/ckks/seal_ckks_matmult_cipherbatchaxis_benchmark.cpp:404-420 */
int he355_multiply_accumulate(he355_ctx *ctx, int L, uint64_t rows, uint64_t cols, uint64_t inner, const uint64_t *d_a, uint64_t a_stride_i,
                              uint64_t a_stride_k, const uint64_t *d_b, uint64_t b_stride_k, uint64_t b_stride_j, uint64_t *d_out);
/* BFV: out(i,j) = sum_k relinearize(multiply(a(i,k), b(k,j))), size-2 results [rows*cols][2][L][N], same addressing -- the
 * multiply / relinearize_inplace / add_inplace loop of src/benchmarks/bfv/seal_bfv_matmult_cipherbatchaxis_benchmark.cpp:398-410 with
 * the inner index inside the batch; d_out may not overlap an operand */
int he355_bfv_multiply_relin_accumulate(he355_ctx *ctx, int L, uint64_t rows, uint64_t cols, uint64_t inner, const uint64_t *d_a, uint64_t a_stride_i,
                                        uint64_t a_stride_k, const uint64_t *d_b, uint64_t b_stride_k, uint64_t b_stride_j, uint64_t *d_out);
/* relinearize_inplace + rescale_to_next_inplace of size-3 ciphertexts (same file :436-437): [n][3][L][N] -> [n][2][L-1][N] */
int he355_relinearize_rescale(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_ct3, uint64_t *d_out);
int he355_rescale(he355_ctx *ctx, int L, int size, uint64_t n, const uint64_t *d_in, uint64_t *d_out);         /* -> [n][size][L-1][N] */
int he355_apply_galois(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_in, uint32_t galois_elt, uint64_t *d_out);
int he355_rotate(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_in, int step, uint64_t *d_out);
/* d_out[i] = rotate(d_in[i], h_steps[i]), i < n: the rotate_vector(dot_i, -i) loop of collapseCKKS (src/engine/seal_context.cpp:389-392)
 * as batched key switches (ciphertexts that need the same Galois element at the same point of their NAF sequence go together).
 * h_steps: host array of n steps.  Not in place. */
int he355_rotate_each(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_in, const int32_t *h_steps, uint64_t *d_out);
/* d_out = d_addend + rotate(d_in, step): the rotate + add_inplace pair of the row-major MatMult inner loop
 * (src/benchmarks/bfv/seal_bfv_matmult_row_benchmark.cpp:525-531) and of accumulateCKKS/BFV (src/engine/seal_context.cpp:337-338,
 * 302-303) as one pipeline.  d_addend may be d_out (add in place) when the step has its own Galois key; d_in may be neither. */
int he355_rotate_add(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_in, int step, const uint64_t *d_addend, uint64_t *d_out);
/* d_out = d_in + sum_j rotate(d_in, h_steps[j]), j < n_steps: the whole inner loop of the row-major MatMult
 * (src/benchmarks/bfv/seal_bfv_matmult_row_benchmark.cpp:519-531, src/benchmarks/ckks/seal_ckks_matmult_row_benchmark.cpp:502-514).  Each rotation
 * is Evaluator::rotate_internal's (own Galois key, else NAF terms least significant first); rotations whose term sequences share a
 * prefix share that prefix's ciphertext, which is computed once -- bit-identical to the unshared loop.  *key_switches (optional):
 * Galois key switches issued per ciphertext.  h_steps: host array.  Not in place. */
int he355_rotate_sum(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_in, const int32_t *h_steps, uint64_t n_steps, uint64_t *d_out,
                     uint64_t *key_switches);
/* accumulateCKKS / accumulateBFV: in place log-tree sum of the first `count` slots; d_tmp: scratch slab of the same size.
 * count == 0 is the reference's else-branch (src/engine/seal_context.cpp:312-316, 341-344): every ciphertext is replaced by a
 * FRESH encryption of zero (Encryptor::encrypt_zero; needs the public key; L must be the top level, as SEAL returns a
 * top-level ciphertext there).  The randomness comes from the context's own stream: seeded from the OS, or pinned with
 * he355_set_zero_stream (ciphertext r then equals he355_encrypt_zero(seed, first_index + r)). */
int he355_accumulate(he355_ctx *ctx, int L, uint64_t n, uint64_t *d_inout, uint64_t count, uint64_t *d_tmp);
/* Encryptor::encrypt_zero at the first data level: d_out [n][2][Ltop][N]; streams as he355_encrypt */
int he355_encrypt_zero(he355_ctx *ctx, uint64_t n, uint64_t seed, uint64_t first_index, uint64_t *d_out);
int he355_set_zero_stream(he355_ctx *ctx, uint64_t seed, uint64_t first_index);
/* Introspection: the API-Bridge ABI (enumerators, sizes, offsets, header used) this library was compiled with, as JSON; returns the
 * size needed including the terminator (csrc/bridge/abi_check.cpp).  A harness binding can check its own numbers against it. */
uint64_t he355_bridge_abi(char *p_buffer, uint64_t size);
/* Introspection: bytes the most recent multi-device load() of the bridge moved to device `device` (> 0) for operand 0 / 1
 * (csrc/bridge/multi_device.cpp): operand 0 travels in per-device blocks, operand 1 whole. */
uint64_t he355_bridge_group_load_bytes(int device, int operand);
/* ---- client side on the device: encryptor()->encrypt (ckks eltwise .cpp:242, bfv eltwise .cpp:233) and
 * SEALContextWrapper::decrypt (src/engine/seal_context.cpp:265-287), batched.  Keys: host arrays in SEAL layout, NTT form:
 * public key [2][K][N], secret key [K][N].  Randomness of he355_encrypt is counter-based: ciphertext r draws u, e0, e1 from
 * (seed, index first_index + r) — csrc/client/sampler.h — so the host client with the same seed/index gives the same bits. */
int he355_set_public_key(he355_ctx *ctx, const uint64_t *h_pk);
int he355_set_secret_key(he355_ctx *ctx, const uint64_t *h_sk);
/* d_plain: CKKS [n][L_top][N] NTT-form plaintexts, BFV [n][N] coefficients mod t; d_out: [n][2][L_top][N] */
int he355_encrypt(he355_ctx *ctx, uint64_t n, const uint64_t *d_plain, uint64_t seed, uint64_t first_index, uint64_t *d_out);
/* d_ct: [n][size][L][N], size 2 or 3; d_out: CKKS [n][L][N] NTT-form plaintext, BFV [n][N] coefficients mod t */
int he355_decrypt(he355_ctx *ctx, int L, int size, uint64_t n, const uint64_t *d_ct, uint64_t *d_out);
/* Encoders on the device: CKKSEncoder()->encode / decode and BatchEncoder (src/engine/seal_context.cpp:145-185 and the
 * benchmarks' encode()/decode(), e.g. src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:163-226).  All pointers are device
 * pointers.  CKKS: values [n][count] doubles (count <= N/2, missing slots are 0) -> [n][L_top][N] NTT-form plaintexts at `scale`;
 * decode: [n][L][N] -> [n][N/2] real parts.  BFV: values [n][count] int64 <-> [n][N] coefficients mod t (centred on decode). */
int he355_ckks_encode(he355_ctx *ctx, uint64_t n, const double *d_values, uint64_t count, double scale, uint64_t *d_plain);
int he355_ckks_decode(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_plain, double scale, double *d_out);
int he355_bfv_encode(he355_ctx *ctx, uint64_t n, const int64_t *d_values, uint64_t count, uint64_t *d_plain);
int he355_bfv_decode(he355_ctx *ctx, uint64_t n, const uint64_t *d_plain, int64_t *d_out);
/* The same decoders writing only the slots a caller reads: `ranges` = n_ranges (1..4) HOST pairs {first_slot, count}; d_out is
 * [n][sum of counts], the ranges one after the other.  A workload's decode() copies the first n (vectors) or dim3 (matrix rows) slots of
 * each result (src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:214-226; both batching rows for the BFV row-major product,
 * bfv/seal_bfv_matmult_row_benchmark.cpp:339-369), so the device writes and the host downloads n x count values instead of n x N/2 (N). */
int he355_ckks_decode_slots(he355_ctx *ctx, int L, uint64_t n, const uint64_t *d_plain, double scale, const uint64_t *ranges, uint64_t n_ranges, double *d_out);
int he355_bfv_decode_slots(he355_ctx *ctx, uint64_t n, const uint64_t *d_plain, const uint64_t *ranges, uint64_t n_ranges, int64_t *d_out);
/* Page-locked host memory (hipHostMalloc) for he355_upload / he355_download of small, frequent transfers (the bridge's decode results and
 * encode inputs live in one such buffer per benchmark object). */
int he355_host_alloc(he355_ctx *ctx, uint64_t bytes, void **h_ptr);
int he355_host_free(he355_ctx *ctx, void *h_ptr);
/* transforms of n_polys residue polynomials, polynomial p under prime prime_of[p % period] (test / client use) */
int he355_ntt_forward(he355_ctx *ctx, uint64_t *d_polys, uint64_t n_polys, const uint8_t *prime_of, uint32_t period);
int he355_ntt_inverse(he355_ctx *ctx, uint64_t *d_polys, uint64_t n_polys, const uint8_t *prime_of, uint32_t period);

/* ---- timing on the stream the kernels run on (HIP events) ---- */
int he355_timer_begin(he355_ctx *ctx);
int he355_timer_end(he355_ctx *ctx, float *elapsed_ms);
/* HIP events around every launch of the dominant kernel (k_k3, fp64-engine primes: the key-product kernel of the key switch)
 * between he355_timer_begin and he355_timer_end: summed duration, number of launches and ops they covered */
int he355_probe_dominant_kernel(he355_ctx *ctx, float *total_ms, uint64_t *launches, uint64_t *ops);
/* Shader clock held while a region runs: he355_clock_probe_begin launches one 64-lane wave on a stream of its own that samples the
 * shader-cycle counter against the constant 100 MHz counter for `duration_us` of real time (a bound it always reaches) while whatever
 * is queued next runs beside it; he355_clock_probe_end waits for it: *mhz = cycles / real time, *seconds = the time it covered.
 * (bench.py: the VALU-issue roofline needs the clock the chip held under THIS load, not a nominal one.) */
int he355_clock_probe_begin(he355_ctx *ctx, uint64_t duration_us);
int he355_clock_probe_end(he355_ctx *ctx, double *mhz, double *seconds);
/* ---- tuning ---- */
int he355_set_dual_stream(he355_ctx *ctx, int on); /* chunks alternate between two HIP streams (default 1; HE355_DUAL_STREAM=0); 0: per-kernel timings without overlap */
/* Key switches over at most n ciphertexts take the latency shape (serial loops of the throughput kernels dealt to more blocks; HEBench's
 * Latency category is batch 1: src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:138-141).  Default: 2^17 / N ciphertexts, at most 12 -- where the throughput kernels start filling the chip (4 at N = 2^15, 8 at 2^14);
 * a call (or HE355_LATENCY_MAX) replaces the rule by n; 0: never; UINT64_MAX: the rule again.
 * Results are bit-identical either way. */
int he355_set_latency_max(he355_ctx *ctx, uint64_t n);
/* he355_rotate_sum walks its NAF-prefix trie level by level, all nodes of a level in one grouped key-switch sequence (default 1;
 * HE355_LEVEL_WALK=0), or node by node (0: one sequence per node; what CKKS batches within the latency shape always take).  Results are
 * bit-identical either way. */
int he355_set_level_walk(he355_ctx *ctx, int on);
/* Rings that fit one CU's LDS (N <= 8192, at most 8 data primes): key switches over at most n ciphertexts per kernel sequence run as TWO
 * launches of one-polynomial workgroups whose transforms never leave LDS (csrc/he355_kernels_lds.hip; the reference's descriptors default
 * to N = 8192: src/benchmarks/ckks/seal_ckks_dot_product_benchmark.cpp:53-60).  Default: the library's rule (one and a half rounds of the chip for the first kernel's grid: 64 ciphertexts at {60, 40, 60}); a call (or
 * HE355_LDS_MAX) replaces it by n; 0: never; UINT64_MAX: the rule again.  Results are bit-identical either way. */
int he355_set_lds_max(he355_ctx *ctx, uint64_t n);
/* ops processed per kernel sequence: default 1024 (HE355_CHUNK), i.e. BASELINE configs[2]'s batch in one piece (scratch ~ 117 MiB/op at
 * N=2^15, L=16).  The size actually used is halved until the scratch arena(s) fit in the device memory that is free at the call. */
int he355_set_chunk(he355_ctx *ctx, uint64_t ops_per_chunk);
int he355_mem_info(he355_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes); /* hipMemGetInfo of the context's device */

#ifdef __cplusplus
}
#endif
#endif
