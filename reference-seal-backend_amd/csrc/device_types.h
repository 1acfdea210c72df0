// device_types.h — plain structs shared by host launch code and HIP kernels.
#pragma once
#include "modarith.h"

namespace he355 {

constexpr int kMaxPrimes = 64; // primes one context can hold: key chain + BEHZ auxiliary primes + the plain modulus (per-prime tables are sized by it)

// Per-prime constants + table pointers, resident in HBM (array of K entries), read wave-uniformly.
struct PrimeDev {
    u64 q, cr0, cr1;      // Barrett (u64 engine and element-wise reductions)
    u64 ninv, ninv_q;     // N^-1 and Shoup quotient
    double qd, qinv;      // fp64 engine
    double ninv_d, ninv_i;
    const Tw16 *fwd;      // N entries
    const Tw16 *inv;      // N entries
    Tw16 inv_w0_scaled;
    int f64;              // 1: ArF64 engine owns this prime
    int pad_;
    // k_k2n, this prime as the DIGIT prime j: how the digit's column enters the forward column pass of fp64-engine target prime t
    // (bit t).  k2_direct: as it is (q_j < 2^52, q_j <= 2 q_t) and the 48-bit row format holds the result; k2_lift: after a
    // re-centring (q_j > 2 q_t) or an integer reduction (q_j >= 2^52), result fits too; neither bit: the general path.
    u64 k2_direct, k2_lift;
    double pow32;         // 2^32 mod q: a 64-bit residue of a wider prime enters the fp64 engine as hi * pow32 + lo (k_floor_colsn)
    // fp64 engine: the column passes' twiddles as bare doubles, [0,32) = w of fwd[0..31], [32,64) = w of inv[0..31] (a column pass of
    // N1 <= 32 rows uses entries 1 .. N1-1).  Inside the struct: one scalar load away from the prime index.  u64 engine: zeros
    alignas(64) double colw[64];
};

// How result r picks its operands (HEBench outer product, ckks eltwise .cpp:334-336, or pairwise)
struct Indexer {
    u64 a_base, b_base; // value_index of operand 0 / 1
    u64 b1;             // batch size of operand 1 (row length of the outer product)
    int pairwise;       // 1: a = a_base + r, b = b_base + r
    int pad_;
};
HE_HD u64 idx_a(const Indexer &ix, u64 r) { return ix.pairwise ? ix.a_base + r : ix.a_base + r / ix.b1; }
HE_HD u64 idx_b(const Indexer &ix, u64 r) { return ix.pairwise ? ix.b_base + r : ix.b_base + r % ix.b1; }
// Operand selection of the BFV multiply: result r = (g, i, j) with g = r / gs, i = (r % gs) / b1, j = (r % gs) % b1 takes
// a[a_base + g a_sg + i a_si] x b[b_base + g b_sg + j b_sj] -- pairwise (gs = 1), outer product (gs = "infinity") and the product
// terms of a matrix product over the inner index g (he355_bfv_multiply_relin_accumulate) in one form.  Used by the BEHZ extension
// kernels only (block-uniform there), so the other kernels' Indexer arithmetic stays what it was.
struct Indexer3 {
    u64 a_base, b_base, gs, b1, a_sg, a_si, b_sg, b_sj;
};
HE_HD Indexer3 to_ix3(const Indexer &ix)
{
    Indexer3 x;
    x.a_base = ix.a_base; x.b_base = ix.b_base;
    if (ix.pairwise) { x.gs = 1; x.b1 = 1; x.a_sg = 1; x.b_sg = 1; x.a_si = 0; x.b_sj = 0; }
    else { x.gs = ~(u64)0; x.b1 = ix.b1; x.a_sg = 0; x.b_sg = 0; x.a_si = 1; x.b_sj = 1; }
    return x;
}
HE_HD u64 idx_a(const Indexer3 &ix, u64 r) { return ix.a_base + (r / ix.gs) * ix.a_sg + ((r % ix.gs) / ix.b1) * ix.a_si; }
HE_HD u64 idx_b(const Indexer3 &ix, u64 r) { return ix.b_base + (r / ix.gs) * ix.b_sg + ((r % ix.gs) % ix.b1) * ix.b_sj; }
// Which ciphertexts the BEHZ extension reads.  lists == 0: item 2 r + c is operand c of result op_offset + r (every result extends
// its own two operands).  lists == 1: the DISTINCT operands of a batch, each once -- items 0 .. na-1 are a(g, i) = item / I, item % I,
// items na .. na+nb-1 are b(g, j) = (item - na) / J, (item - na) % J (the multiply then reads result r's operands at ordinals
// ord_a(r), ord_b(r)).
struct BehzSrc {
    const u64 *a, *b;
    Indexer3 ix;
    u64 op_offset;
    u64 I, J, na;
    int lists, pad_;
};
HE_HD const u64 *behz_src_ct(const BehzSrc &s, u64 item, u64 ct_words)
{
    if (!s.lists) {
        const u64 rg = s.op_offset + (item >> 1);
        return (item & 1) ? s.b + idx_b(s.ix, rg) * ct_words : s.a + idx_a(s.ix, rg) * ct_words;
    }
    if (item < s.na) return s.a + (s.ix.a_base + (item / s.I) * s.ix.a_sg + (item % s.I) * s.ix.a_si) * ct_words;
    const u64 e = item - s.na;
    return s.b + (s.ix.b_base + (e / s.J) * s.ix.b_sg + (e % s.J) * s.ix.b_sj) * ct_words;
}
HE_HD u64 ord_a(const BehzSrc &s, u64 r) { return (r / s.ix.gs) * s.I + (r % s.ix.gs) / s.ix.b1; }
HE_HD u64 ord_b(const BehzSrc &s, u64 r) { return s.na + (r / s.ix.gs) * s.J + (r % s.ix.gs) % s.ix.b1; }

// A batch of residue polynomials for the generic transform kernels:
// poly p of item it lives at base + it*item_stride + p*N and belongs to prime prime_of[p] (255 = skip).
struct PolyView {
    u64 *base;
    u64 item_stride;
    int polys_per_item;
    int pad_;
    unsigned char prime_of[64];
};

} // namespace he355
