// he_client.h — client-side (host) half of the backend: key generation, encoders, encryption, decryption.
//
// In the reference these are seal::KeyGenerator / CKKSEncoder / BatchEncoder / Encryptor / Decryptor, created in
// SEALContextWrapper::createKeysAndEncryptors (/root/reference/src/engine/seal_context.cpp:46-70) and called from
// the benchmarks' encode/encrypt/decrypt/decode (e.g. src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:163-275).
// They run outside the timed operate() and stay on the host in this round.
// Nothing here is reachable from the evaluator path: the device never falls back to this code.
#pragma once
#include <complex>
#include <cstdint>
#include <random>
#include <vector>

#include "../he_params.h"
#include "ckks_codec.h"

namespace he355 {
namespace client {

// scalar host transforms over one residue (same ordering as the device: natural <-> bit-reversed)
void host_ntt_forward(const PrimeTables &pt, size_t N, u64 *poly);
void host_ntt_inverse(const PrimeTables &pt, size_t N, u64 *poly);

// encoder tables (also built by the device context, so that both sides use the very same values)
void build_slot_index(size_t N, std::vector<uint32_t> &slot_index);
void build_ckks_tables(size_t N, std::vector<Cplx> &W, std::vector<Cplx> &Z);

class Client {
public:
    Client(const Params &params, uint64_t seed);

    const Params &params() const { return P; }
    size_t slot_count() const { return P.scheme == kSchemeCKKS ? P.N / 2 : P.N; }

    // ---- keys (SEAL layouts, NTT form) ----
    const std::vector<u64> &secret_key() const { return sk_; } // [K][N]
    const std::vector<u64> &public_key() const { return pk_; } // [2][K][N]
    // encoder tables, for the device-side encoders: slot -> evaluation-point index (N entries), transform and twist tables
    const std::vector<uint32_t> &slot_index() const { return slot_index_; }
    const std::vector<Cplx> &fft_table() const { return fft_w_; }
    const std::vector<Cplx> &twist_table() const { return zeta_; }
    std::vector<u64> make_relin_key();                         // [Ltop][2][K][N]
    std::vector<u64> make_galois_key(uint32_t galois_elt);     // [Ltop][2][K][N]

    // ---- encoders ----
    // CKKSEncoder::encode: up to N/2 values -> [Ltop][N] residues in NTT form at `scale`
    std::vector<u64> ckks_encode(const double *values, size_t count, double scale) const;
    // CKKSEncoder::decode: [L][N] NTT-form plaintext -> N/2 real parts
    void ckks_decode(const u64 *plain_ntt, size_t L, double scale, double *out) const;
    // BatchEncoder::encode / decode: N int64 slots <-> [N] coefficients mod t
    std::vector<u64> bfv_encode(const int64_t *values, size_t count) const;
    void bfv_decode(const u64 *plain, int64_t *out) const;

    // ---- encryption at the first data level: ct [2][Ltop][N] (CKKS NTT form, BFV coefficient form) ----
    // Randomness is counter-based (client/sampler.h): the ciphertext with index r draws its three polynomials from
    // (encrypt_seed, streams 3r..3r+2); every call uses the next index.  The device encryption (he355_encrypt) given the
    // same seed and index produces the same bits.
    std::vector<u64> encrypt(const u64 *plain);
    std::vector<u64> encrypt_zero();
    uint64_t encrypt_seed() const { return enc_seed_; }
    uint64_t keygen_seed() const { return key_seed_; } // evaluation keys: streams keygen_stream(key id, digit, ...) of this seed
    uint64_t encrypt_index() const { return enc_index_; }               // index the next encryption will use
    void set_encrypt_index(uint64_t index) { enc_index_ = index; }
    // ---- decryption of a size-`size` ciphertext at level L (size 3 allowed: the reference decrypts
    //      un-relinearized products, ckks eltwise .cpp:342-344). CKKS: [L][N] NTT plaintext; BFV: [N] mod t ----
    std::vector<u64> decrypt(const u64 *ct, size_t size, size_t L) const;

private:
    void sample_ternary(std::vector<u64> &out, size_t nmod);
    void sample_cbd(std::vector<u64> &out, size_t nmod);
    void sample_uniform(std::vector<u64> &out, size_t nmod);
    void enc_zero_symmetric(u64 *out /*[2][K][N]*/, u64 key_id, u64 digit);
    std::vector<u64> make_kswitch_key(const std::vector<u64> &new_key, u64 key_id);
    void divide_round_last(const std::vector<u64> &in, size_t size, std::vector<u64> &out) const; // key level -> data level

    const Params &P;
    std::mt19937_64 rng_;
    uint64_t enc_seed_ = 0, enc_index_ = 0;
    uint64_t key_seed_ = 0;
    std::vector<u64> sk_, pk_;
    std::vector<uint32_t> slot_index_; // encoders' slot -> (bit-reversed) evaluation index map
    std::vector<Cplx> fft_w_, zeta_;   // CKKS encoder transform tables (client/ckks_codec.h); the device gets copies
    PrimeTables plain_tables_;         // BFV: NTT mod t
};

} // namespace client
} // namespace he355
