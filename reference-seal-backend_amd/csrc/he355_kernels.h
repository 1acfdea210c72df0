// he355_kernels.h — host-callable launchers of the HIP kernels (implemented in he355_kernels.hip).
// All launchers enqueue on `stream` and return immediately; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <utility>

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
    bool u64_fold = false; // which build of the device code this context's tables were made for (Params::u64_fold)
    unsigned char prime_f64[kMaxPrimes]; // host copy: 1 if the fp64 engine owns prime i
    u64 prime_q[kMaxPrimes];             // host copy of the moduli
};




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
    // Level sum inside the fused k_k3 (he355_rotate_sum, round 5): sum_out [group_size][2][L][N] += (count[g] & 0x7fffffff) x the group's
    // rotated ciphertext, added by the wave that formed it; a group's own ciphertext is written only where bit 31 of count[g] says some
    // group of the next level starts from it.  Race-free because the launch then gives a block ONE (tile, eight ciphertexts) and lets it
    // walk the groups one after the other (launch_k3 sizes the grid so: needs group_size % 8 == 0 and whole groups per launch).
    u64 *sum_out = nullptr;
    const u32 *count = nullptr;
};
constexpr u32 kGroupKeepBit = 0x80000000u;
constexpr int kMoveListCap = 64;
enum K1Mode { K1_MUL = 0, K1_CT3 = 1, K1_GALOIS = 2 };
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
    // 4 (a rotation without addend, round 5): polynomial 1 is zero, and polynomial 0 -- the Galois-permuted c0 -- is GATHERED by k_k3's
    // epilogue from the rotation's input (gsrc [.][2][L][N], op reads ciphertext gsrc_op_offset + op through the NTT-domain permutation
    // gperm; grouped launches take both from the op's group) behind the correction row's transform: k_k1 does not write it (no_c0n).
    // A row of the permuted polynomial is a permutation of ONE row of the source (the low bits of the evaluation point's exponent
    // depend on the row alone), so the gather touches the 8 KiB it needs and no more.  (The own digit -- the permuted c1 in NTT form --
    // stays a row k_k1 writes: gathered here it sat in front of the first products with two dependent loads, and pulled through the
    // landing buffer it delayed the first digit row; both measured slower than the row they saved, profiles/r05_rotation_gather.txt.)
    // 5: the same for a rotation WITH addend (he355_rotate_add, accumulate): polynomial 0 = the addend's polynomial 0 (row of c1_src) + the gathered
    // permuted c0, polynomial 1 = the addend's (as mode 2); k_k1 reads neither c0 nor the addend and writes neither.
    int c1_mode = 0;
    const u64 *c1_src = nullptr;
    const u64 *gsrc = nullptr;
    const uint32_t *gperm = nullptr;
    u64 gsrc_op_offset = 0;
};
// HE355_K3_FUSE: "0" = the unfused sequence everywhere (k_floor_rows finishes the mod-down); unset / "1" = fused where it pays, with the
// small-grid rules (both engines in one launch, four-wave u64-engine blocks, unfused below a minimum of special-prime blocks); "all" =
// fused for every throughput-shape batch and none of the small-grid rules (the schedule before those rules existed).  The thresholds
// are constants (profiles/r04_dual_engine_latency.txt: swept on one box).
inline int k3_fuse_policy()
{
    static const int v = [] {
        const char *e = std::getenv("HE355_K3_FUSE");
        if (!e) return 1;
        if (e[0] == '0') return 0;
        if (e[0] == 'a' || e[0] == 'A') return 2;
        return 1;
    }();
    return v;
}
inline bool k3_can_fuse(const KernelEnv &env) { return k3_fuse_policy() != 0 && env.scheme == 2 && env.K >= 2; }
// HE355_BEHZ_FUSE=<mask>: bit 0 = extension / floor fused with the column passes, bit 1 = operands shared by several results extended and
// transformed once (he355_api.hip: bfv_multiply3); default 3.
inline int behz_fuse_mask()
{
    static const int v = [] { const char *e = std::getenv("HE355_BEHZ_FUSE"); return e ? std::atoi(e) & 3 : 3; }();
    return v;
}
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


struct CrtTablesDev { // device arrays of the CRT tables of the first L primes (client/multiword.h CrtView)
    int L, words;
    const u64 *Q, *halfQ, *punct, *inv;
    double Qd;
    u64 t;
};

// ---- encoders on the device ---------------------------------------------------------------------------------
struct EncTablesDev { // device copies of the host client's encoder tables (client/ckks_codec.h); W, Z: [N] {re, im} doubles
    const uint32_t *slot_index;
    const void *W, *Z;
};


// ---- key switch for rings that fit one CU's LDS (he355_kernels_lds.hip): where the target polynomial and the ciphertext the switched key
// part is added into come from.  The kernels read them where they lie -- nothing is staged in HBM for them.
enum LdsKsMode {
    LDSKS_PLAIN = 0,  // target: tgt + op * tgt_op_stride, [L][N] NTT form; addend: add + op * add_op_stride, [2][L][N] (null: zero) -- relinearize of size-3 inputs
    LDSKS_MUL = 1,    // ct x ct multiply: target = a1 b1, addend = (a0 b0, a0 b1 + a1 b0); result r = op_offset + op takes a[idx_a(ix, r)], b[idx_b(ix, r)]
    LDSKS_GALOIS = 2, // rotation: target = perm(c1), addend = (perm(c0), 0) [+ add]; a = the input slab [.][2][L][N], op reads ciphertext op_offset + op
};
struct LdsKsOperands {
    int mode = LDSKS_PLAIN;
    const u64 *tgt = nullptr; u64 tgt_op_stride = 0;
    const u64 *add = nullptr; u64 add_op_stride = 0;
    const u64 *a = nullptr, *b = nullptr;
    Indexer ix{};
    u64 op_offset = 0;
    const uint32_t *perm = nullptr;
    unsigned char perm_src_row[32] = {}; // LDSKS_GALOIS: the ONE source row row a of the permuted polynomial comes from (perm[a * 1024] >> 10; Params::galois_perm_ntt)
};

// the slots a decode writes per plaintext: up to kMaxSlotRanges ranges [first, first + count) of the slot vector, written one after the other
// (out row = `total` values).  A workload's decode() reads the first n / dim3 slots of a result -- and BFV MatMultRow those of both batching rows
// (bfv row .cpp:339-369) -- not the N/2 or N the encoder has
constexpr int kMaxSlotRanges = 4;
struct SlotRanges {
    u32 n, total;
    u32 first[kMaxSlotRanges], count[kMaxSlotRanges];
};

// ---- the launchers: two builds of the device code -----------------------------------------------------------------------
namespace ks_shoup {
#include "he355_launchers.inc"
}
namespace ks_fold {
#include "he355_launchers.inc"
}
#if !defined(HE355_KNS)
// Host side (he355_api.hip): every launcher takes the KernelEnv first, and the environment knows which build its tables were made
// for -- the call sites name the launcher, these forwarders pick the namespace.
#define HE355_FWD(name)                                                                                                   \
    template <class... A> inline auto name(const KernelEnv &env, A &&...a) -> decltype(ks_shoup::name(env, std::forward<A>(a)...)) \
    {                                                                                                                     \
        return env.u64_fold ? ks_fold::name(env, std::forward<A>(a)...) : ks_shoup::name(env, std::forward<A>(a)...);     \
    }
HE355_FWD(launch_ntt_forward)
HE355_FWD(launch_ntt_inverse)
HE355_FWD(launch_addsub)
HE355_FWD(launch_mul3)
HE355_FWD(launch_plain_op)
HE355_FWD(launch_drop_residues)
HE355_FWD(launch_sum_cts)
HE355_FWD(launch_mul3_acc)
HE355_FWD(launch_sum_groups)
HE355_FWD(launch_move_cts)
HE355_FWD(launch_k1)
HE355_FWD(launch_k2)
HE355_FWD(launch_k3)
HE355_FWD(launch_k3_combine)
HE355_FWD(launch_floor_cols)
HE355_FWD(launch_floor_rows)
HE355_FWD(launch_behz_extend)
HE355_FWD(behz_cols_fusable)
HE355_FWD(launch_behz_extend_cols)
HE355_FWD(launch_behz_tensor_inv)
HE355_FWD(launch_rows_fwd)
HE355_FWD(launch_behz_cols_floor_sk)
HE355_FWD(launch_behz_rows_tensor)
HE355_FWD(launch_cols_fwd)
HE355_FWD(launch_cols_inv)
HE355_FWD(launch_behz_floor_sk)
HE355_FWD(launch_bfv_galois)
HE355_FWD(launch_bfv_tail_sp)
HE355_FWD(launch_bfv_tail_fin)
HE355_FWD(launch_rows_inv_select)
HE355_FWD(launch_enc_sample)
HE355_FWD(launch_enc_mul_pk)
HE355_FWD(launch_divround_last_coeff)
HE355_FWD(launch_bfv_add_scaled_plain)
HE355_FWD(launch_dot_sk)
HE355_FWD(launch_bfv_scale_round)
HE355_FWD(launch_ckks_encode)
HE355_FWD(launch_ckks_decode)
HE355_FWD(launch_bfv_encode_scatter)
HE355_FWD(launch_bfv_decode_gather)
HE355_FWD(launch_keygen_kswitch)
HE355_FWD(launch_ks_lds)
HE355_FWD(launch_rescale_lds)
HE355_FWD(ks_lds_supported)
HE355_FWD(ks_lds_part_words)
#undef HE355_FWD
#endif

} // namespace he355
