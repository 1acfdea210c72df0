// he355_kernels.h — host-callable launchers of the HIP kernels (implemented in he355_kernels.hip).
// All launchers enqueue on `stream` and return immediately; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "behz_core.h"

namespace he355 {

// Per (source prime s, target prime i) constants of the RNS divide-and-round ("floor") step used by the
// key-switch mod-down (s = special prime) and by rescale (s = last data prime).  Device array [K][K].
struct FloorConst {
    u64 inv, inv_shoup;   // s^-1 mod q_i (+ Shoup quotient)
    double inv_d, inv_i;  // same for the fp64 engine: value and fl(value/q_i)
    u64 half_mod;         // floor(s/2) mod q_i
    u64 src_mod;          // s mod q_i
};

// Optional HIP-event probe around the launches of the dominant kernel (k_k3, fp64 engine): bench.py reports that
// kernel's average launch duration from inside the timed region (events on the stream the kernel runs on).
struct KernelProbe {
    static constexpr int kCap = 2048;
    hipEvent_t start[kCap], stop[kCap];
    int created = 0, used = 0;
    u64 ops = 0; // ops covered by the probed launches
};

struct KernelEnv {
    KernelProbe *probe = nullptr; // null: no probing
    const PrimeDev *primes; // device array [K]
    const FloorConst *floor_consts; // device array [K*K], entry [s*K + i]
    int N, logn1, K, Ltop, scheme;
    hipStream_t stream;
    unsigned char prime_f64[kMaxPrimes]; // host copy: 1 if the fp64 engine owns prime i
    u64 prime_q[kMaxPrimes];             // host copy of the moduli
};


// ---- generic transforms over a PolyView (in place) ---------------------------------------------------
void launch_ntt_forward(const KernelEnv &env, const PolyView &v, u32 n_items);  // canonical -> canonical NTT form
void launch_ntt_inverse(const KernelEnv &env, const PolyView &v, u32 n_items);  // canonical NTT form -> coefficients

// ---- element-wise ------------------------------------------------------------------------------------
// out[r][p][n] = a[ia(r)][p][n] (+|-) b[ib(r)][p][n] mod q_{p % L}; polys = size * L
void launch_addsub(const KernelEnv &env, int L, int size, u64 n_results, const u64 *a, const u64 *b, Indexer ix, u64 *out, bool sub);
// dyadic tensor (CKKS multiply): a,b size-2 level-L NTT form -> out size 3
void launch_mul3(const KernelEnv &env, int L, u64 n_results, const u64 *a, const u64 *b, Indexer ix, u64 *out);
void launch_plain_op(const KernelEnv &env, int L, int size, u64 n_results, const u64 *ct, const u64 *pt, Indexer ix, u64 *out, int mode); // 0 mul, 1 add
void launch_drop_residues(const KernelEnv &env, int L, int L_to, u64 n_polys, const u64 *in, u64 *out);
// out[c] (+)= sum_r in[r * n_out + c], c < n_out
void launch_sum_cts(const KernelEnv &env, int L, int size, u64 n_terms, const u64 *in, u64 *out, u64 n_out = 1, bool accumulate = false);
void launch_mul3_acc(const KernelEnv &env, int L, u64 rows, u64 cols, int inner, const u64 *a, u64 a_stride_i, u64 a_stride_k, const u64 *b,
                     u64 b_stride_k, u64 b_stride_j, u64 *out);

// ---- key-switch pipeline pieces (see DESIGN.md "Key-switch pipeline") ---------------------------------
struct KsBuffers {
    u64 *c01;   // [C][2][L][N]   canonical: polys that receive the key-switched result (may be the output)
    u64 c01_item_stride;
    u64 *c2n;   // [C][L][N]      canonical NTT-form target (digit i==j operand, CKKS)
    u64 *c2r;   // [C][L][N]      target after the inverse row pass (raw of prime j; canonical if N == 1024)
    u64 *d;     // [C][L+1][L][N] digit j lifted to prime tt after the forward column pass (raw of prime tt)
    u64 *t;     // [C][2][L][N]   accumulated key products, data primes, canonical NTT form
    u64 *tp;    // [C][2][N]      special-prime accumulations, canonical NTT form
    u64 *tpr;   // [C][2][N]      the same after the inverse row pass (raw)
    u64 *e;     // [C][2][L][N]   mod-down corrections after the forward column pass (raw of prime i)
};
// Grouped key switches (he355_rotate_sum: all nodes of a trie level in one kernel sequence): the ops of a launch come in groups of
// group_size consecutive ops; group g rotates the ciphertexts of source block src_block[g] (block = group_size consecutive
// ciphertexts of the launch's input slab) by the Galois element whose NTT-domain permutation table is perm[g], with key key[g].
// All three are device arrays of n_groups entries.  group_size == 0: not a grouped launch.
struct KsGroups {
    const uint32_t *const *perm = nullptr;
    const u64 *const *key = nullptr;
    const u32 *src_block = nullptr;
    u32 group_size = 0;
};
// out[c] += sum_g mult[g] * in[g * n_cts + c] over size-2 ciphertexts at level L (d_mult: device array [n_groups])
void launch_sum_groups(const KernelEnv &env, int L, u64 n_cts, u32 n_groups, const u64 *in, const u32 *d_mult, u64 *out);
constexpr int kMoveListCap = 64;
// whole ciphertexts by index list (host array): gather dst[g] = src[idx[g]], scatter dst[idx[g]] = src[g], g < n
void launch_move_cts(const KernelEnv &env, u64 *dst, const u64 *src, const uint32_t *idx, u64 n, u64 elems_per_ct, bool scatter);
enum K1Mode { K1_MUL = 0, K1_CT3 = 1, K1_GALOIS = 2 };
// K1: produce c01 / c2n / c2r for a chunk of ops.  MUL: a,b via indexer.  CT3: `a` is [n][3][L][N].
// GALOIS: `a` is [n][2][L][N], perm = device permutation table (NTT-form gather); optional addend [n][2][L][N] (indexed like `a`):
// the rotated ciphertext starts from it, i.e. the pipeline computes addend + rotate(a).
void launch_k1(const KernelEnv &env, int L, K1Mode mode, u64 n_ops, u64 op_offset, const u64 *a, const u64 *b, Indexer ix,
               const uint32_t *perm, const KsBuffers &buf, const u64 *addend = nullptr, bool no_c01 = false, bool no_c1 = false,
               const KsGroups *groups = nullptr);
// (no_c01, K1_MUL only: just the key-switch target c2 = a1 b1 is produced; c0, c1 are computed where they are consumed, K3Fuse::ta.
//  no_c1, K1_GALOIS only: polynomial 1 of c01 -- zeros, or the addend's -- is not written; the fused k_k3 takes it from K3Fuse::c1_mode)
// K2: finish iNTT of each digit, lift to every key prime, forward column pass -> d
// src_is_coeff (BFV): `src` already holds coefficient-form digits [op][L][N] (op stride src_op_stride) and every
// (prime, digit) pair is lifted, including the digit's own prime
// tsplit > 1 (latency shape, k_k2n): the targets of a (digit, column block) are dealt to tsplit blocks
void launch_k2(const KernelEnv &env, int L, u64 n_ops, const KsBuffers &buf, const u64 *src = nullptr, u64 src_op_stride = 0, int tsplit = 1);
// K3: forward row pass of every (tt, j) + multiply-accumulate with the key -> t (data primes) / tpr (special)
// BFV (env.scheme == 1): the products of ALL primes continue into the inverse row pass (t then holds raw rows)
// `part`: all tiles, only the special prime's, or only the data primes'.  With `fuse` (data-prime tiles, CKKS: k3_can_fuse)
// the mod-down is finished inside the kernel: c01 += (sums - NTT(cols)) * P^-1, no t slab, no k_floor_rows.
enum K3Part { K3_ALL = 0, K3_SPECIAL_ONLY = 1, K3_DATA_ONLY = 2 };
struct K3Fuse {
    const u64 *cols;     // [n_ops*2][L][N] output of launch_floor_cols(special prime -> L targets)
    u64 *c01;
    u64 c01_item_stride;
    // data-prime range of this launch [tt_lo, tt_hi) (K3_DATA_ONLY)
    int tt_lo, tt_hi;
    // second floor step (CKKS rescale by prime L-1) finished in the same epilogue, for primes < L-1: cols2 is the output of
    // launch_floor_cols(prime L-1 -> L-1 targets, src2 = the special prime's sums) [n_ops*2][L-1][N], i.e. the COMBINED correction
    // delta2 + P^-1 * delta1; out [n_ops][2][L-1][N].  Null: mod-down only (correction from cols).
    const u64 *cols2;
    u64 *out;
    // ct x ct multiply (he355_multiply_relin): the addend of the floor step -- c0 = a0 b0, c1 = a0 b1 + a1 b0 -- is computed here from the
    // operand rows instead of being written by k_k1 and read back (k_k1 is HBM-bound, this kernel is not).  Null: the addend is read
    // from c01.  ta / tb: the operand slabs [.][2][L][N], tix / t_op_offset: result r = t_op_offset + op multiplies a[ia(r)] by b[ib(r)].
    const u64 *ta = nullptr, *tb = nullptr;
    Indexer tix{};
    u64 t_op_offset = 0;
    // rotations: polynomial 1 of the ciphertext the switched key part is added into is zero, or polynomial 1 of the rotation's addend --
    // k_k1 neither writes nor copies it.  c1_mode 0: read from c01 (as polynomial 0 always is); 1: zero; 2: row of c1_src [n_ops][2][L][N];
    // 3 (relinearize of size-3 ciphertexts): polynomials 0 AND 1, and the NTT-form own digit, are rows of c1_src [n_ops][3][L][N], the input
    int c1_mode = 0;
    const u64 *c1_src = nullptr;
};
bool k3_can_fuse(const KernelEnv &env);
int k3_fuse_policy(); // HE355_K3_FUSE: 0 never, 1 where it pays + small-grid rules (default), 2 always, no small-grid rules
int behz_fuse_mask();  // HE355_BEHZ_FUSE: bit 0 column-pass fusion, bit 1 operands transformed once
// n_split > 1 (latency shape, unfused only): the digits of every tile are cut into n_split groups, one single-wave block per (tile, op,
// group), canonical partial sums -> split_part [n_split][n_ops * 2][L + 1][N]; launch_k3_combine then leaves t / tpr as the unsplit launch
void launch_k3(const KernelEnv &env, int L, u64 n_ops, const KsBuffers &buf, const u64 *key, K3Part part = K3_ALL, const K3Fuse *fuse = nullptr,
               int n_split = 1, u64 *split_part = nullptr, int n_split_u64 = 0, // n_split_u64: groups of the u64-engine tiles (0: as n_split)
               const KsGroups *groups = nullptr, u64 g_op_offset = 0);         // groups: per-group keys (`key` unused), op 0 of the launch is op g_op_offset of the grouped batch
void launch_k3_combine(const KernelEnv &env, int L, u64 n_ops, const KsBuffers &buf, int n_split, const u64 *split_part, int n_split_u64 = 0);
// floor step, column half: src [n_ops*n_src][N] raw of prime s -> r = (x + floor(s/2)) mod s ->
// (r mod q_i - floor(s/2) mod q_i) for i in [tgt_first, tgt_first + n_tgt) -> forward column pass -> dst [n_ops*n_src][dst_ntgt][N]
// src2 (optional): the SOURCE of an earlier floor step ([n_polys][N] after its inverse row pass, prime src2_prime): its correction is
// folded in before the column pass, delta2 + src2^-1 * delta1 (mod-down + rescale share one column pass and one row transform)
void launch_floor_cols(const KernelEnv &env, int src_prime, int n_tgt, u64 n_polys, const u64 *src, u64 *dst, int tgt_first = 0, int dst_ntgt = 0,
                       const u64 *src2 = nullptr, int src2_prime = 0, int tsplit = 1);
// floor step, row half: out[(op,k,i)] = (tsrc[(op,k,i)] - NTT(dst_cols[(op,k,i)])) * s^-1 (+ addend) mod q_i.
// Strides are in u64 elements.  If tail_prime >= 0 the rows of that prime additionally go through the
// inverse row pass into tail[(op,k)] (next floor step's source).
struct FloorRowsArgs {
    int src_prime, n_tgt, n_src; // n_src polys per op (2)
    const u64 *cols;             // [n_ops*n_src][n_tgt][N] raw
    const u64 *tsrc; u64 tsrc_op_stride, tsrc_poly_stride;
    const u64 *addend; u64 add_op_stride, add_poly_stride; // may be null
    u64 *out; u64 out_op_stride, out_poly_stride;
    int tail_prime;              // -1: none
    u64 *tail;                   // [n_ops*n_src][N]
};
void launch_floor_rows(const KernelEnv &env, u64 n_ops, const FloorRowsArgs &args);
// ---- BFV -------------------------------------------------------------------------------------------------
// (BehzDev, kBehzMaxL / kBehzMaxB and the per-coefficient BEHZ arithmetic: behz_core.h -- host-compilable, the lane simulator runs it on the CPU)
// BEHZ steps (1)-(2): lift the four input polynomials of each pair to Bsk (fastbconv_m_tilde + sm_mrq) and copy them
// for the base-q transform.  xq [n*4][L][N], xbsk [n*4][S][N], coefficient form.
// results op_offset .. op_offset + n_ops - 1 of the batch (the indexer sees the global result index)
// n_cts ciphertext items selected by `src` (device_types.h, BehzSrc: the two operands of every result, or each distinct operand once)
void launch_behz_extend(const KernelEnv &env, const BehzDev &bz, const BehzSrc &src, u64 n_cts, u64 *xq, u64 *xbsk);
// N <= 16384, L <= 4, nB <= 6 (and HE355_BEHZ_FUSE != 0): the extension with the forward column passes of xq / xbsk in its epilogue,
// and the inverse column passes of dq / ds in the prologue of steps (6)-(8) -- the coefficient-form copies never reach HBM
bool behz_cols_fusable(const KernelEnv &env, const BehzDev &bz);
void launch_behz_extend_cols(const KernelEnv &env, const BehzDev &bz, const BehzSrc &src, u64 n_cts, u64 *xq, u64 *xbsk);
// operands transformed once each (src.lists): dyadic tensor + inverse row pass of results op_offset .. op_offset + n_ops - 1
void launch_behz_tensor_inv(const KernelEnv &env, const BehzDev &bz, const BehzSrc &src, u64 n_ops, u64 op_offset, const u64 *eq, const u64 *ebsk, u64 *dq,
                            u64 *ds);
void launch_rows_fwd(const KernelEnv &env, const PolyView &v, u32 n_items); // row half of the forward transform, in place
void launch_behz_cols_floor_sk(const KernelEnv &env, const BehzDev &bz, u64 n_ops, const u64 *dq, const u64 *ds, u64 *out);
// BEHZ steps (3)-(5) on rows, fused: x [n*4][Lx][N] after launch_cols_fwd -> forward row pass of a0, a1, b0, b1, dyadic tensor, inverse row
// pass -> d [n*3][Lx][N] ready for launch_cols_inv (one block = the four rows of one (op, residue, row); no HBM round trip between them)
void launch_behz_rows_tensor(const KernelEnv &env, const BehzDev &bz, u64 n_ops, const u64 *xq, const u64 *xbsk, u64 *dq, u64 *ds);
void launch_cols_fwd(const KernelEnv &env, const PolyView &v, u32 n_items); // column half of the forward transform, in place
void launch_cols_inv(const KernelEnv &env, const PolyView &v, u32 n_items); // column half of the inverse transform, in place
// BEHZ steps (6)-(8): times t, fast floor, Shenoy-Kumaresan -> out [n][3][L][N]
void launch_behz_floor_sk(const KernelEnv &env, const BehzDev &bz, u64 n_ops, const u64 *dq, const u64 *ds, u64 *out);
// coefficient-form Galois: c01[op][0] = sigma(in0), c01[op][1] = 0, tgt[op] = sigma(in1); gather table has the sign in bit 31
void launch_bfv_galois(const KernelEnv &env, int L, u64 n_ops, const u64 *in, const uint32_t *gather, u64 *c01, u64 c01_item_stride, u64 *tgt,
                       const u64 *addend = nullptr); // addend [n][2][L][N]: out = addend + rotate(in)
// BFV key-switch tails: finish the inverse transform of the special-prime sums and round (-> rp), then finish every data
// prime's inverse transform, apply the floor step in coefficient form and add into c01
void launch_bfv_tail_sp(const KernelEnv &env, u64 n_polys, const u64 *tpr, u64 *rp);
// c01 = add01 + key-switched part (add01 == nullptr: c01 += ...; relinearize hands the size-3 input's (c0, c1) here, nothing is copied)
void launch_bfv_tail_fin(const KernelEnv &env, int L, u64 n_ops, const u64 *t, const u64 *rp, u64 *c01, u64 c01_item_stride, const u64 *add01 = nullptr,
                         u64 add01_item_stride = 0);

// inverse row pass of one residue of each poly: src [(op,k)] residue `prime` -> tail [(op,k)][N]
void launch_rows_inv_select(const KernelEnv &env, int prime, u64 n_polys, const u64 *src, u64 src_poly_stride, u64 *tail);

// ---- client side on the device: encryption / decryption ------------------------------
// u [n][K][N], e [n][2][K][N]: sampled polynomials of ciphertexts first_index.. (coefficient form, canonical residues)
void launch_enc_sample(const KernelEnv &env, u64 n_cts, u64 seed, u64 first_index, u64 *u, u64 *e);
// z[r][k][i] = u[r][i] (.) pk[k][i] (+ z if add_in); NTT form at the key level
void launch_enc_mul_pk(const KernelEnv &env, u64 n_cts, const u64 *u, const u64 *pk, u64 *z, bool add_in);
// coefficient-form divide-and-round by the special prime: z [n_polys][K][N] -> out [n_polys][K-1][N]
void launch_divround_last_coeff(const KernelEnv &env, u64 n_polys, const u64 *z, u64 *out);
void launch_bfv_add_scaled_plain(const KernelEnv &env, int L, u64 n_cts, u64 *ct, const u64 *plain, u64 t, u64 q_mod_t, const u64 *qdivt);
void launch_dot_sk(const KernelEnv &env, int L, int size, u64 n_cts, const u64 *ct, const u64 *sk, u64 *out);
struct CrtTablesDev { // device arrays of the CRT tables of the first L primes (client/multiword.h CrtView)
    int L, words;
    const u64 *Q, *halfQ, *punct, *inv;
    double Qd;
    u64 t;
};
void launch_bfv_scale_round(const KernelEnv &env, u64 n_cts, const u64 *phase, u64 *plain, const CrtTablesDev &c);

// ---- encoders on the device ---------------------------------------------------------------------------------
struct EncTablesDev { // device copies of the host client's encoder tables (client/ckks_codec.h); W, Z: [N] {re, im} doubles
    const uint32_t *slot_index;
    const void *W, *Z;
};
// values [n][count] -> plain [n][Ltop][N] coefficient form (zbuf: [n][N] complex scratch; *err |= 1 if a coefficient overflows)
void launch_ckks_encode(const KernelEnv &env, u64 n_vec, const double *values, u64 count, double scale, void *zbuf, u64 *plain, const EncTablesDev &t, int *err);
// coeff [n][L][N] coefficient form -> out [n][N/2]
void launch_ckks_decode(const KernelEnv &env, u64 n_vec, const u64 *coeff, double scale, void *zbuf, double *out, const EncTablesDev &t, const CrtTablesDev &c);
void launch_bfv_encode_scatter(const KernelEnv &env, u64 n_vec, const long long *values, u64 count, u64 *ev, const uint32_t *slot_index, u64 t);
void launch_bfv_decode_gather(const KernelEnv &env, u64 n_vec, const u64 *ev, long long *out, const uint32_t *slot_index, u64 t);

// ---- key generation on the device --------------------------------------------------------------------------------
// key [Ltop][2][K][N] <- key-switching key for new_key = s^2 (perm == null) or s permuted by `perm` (Galois key);
// e_scratch [Ltop][K][N], target_scratch [K][N]; randomness: keygen_stream(key_id, digit, ...) of `seed` (client/sampler.h)
void launch_keygen_kswitch(const KernelEnv &env, u64 *key, u64 *e_scratch, u64 *target_scratch, const u64 *sk, const uint32_t *perm, u64 seed, u64 key_id);

} // namespace he355
