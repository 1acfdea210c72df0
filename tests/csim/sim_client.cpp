// sim_client.cpp — TEST-ONLY C wrappers around the product's client-side host code (client/he_client.cpp) so that
// pytest can cross-check its keys / ciphertexts / encoders against the oracle on the CPU.
#include <cstring>
#include <memory>

#include "../../reference-seal-backend_amd/csrc/client/he_client.h"
#include "../../reference-seal-backend_amd/csrc/client/multiword.h"
#include "../../reference-seal-backend_amd/csrc/client/sampler.h"

using namespace he355;
using client::Client;

extern "C" {
void *simc_create(void *params, uint64_t seed) { return new Client(*(Params *)params, seed); }
void simc_destroy(void *c) { delete (Client *)c; }
void simc_secret_key(void *c, uint64_t *out) { auto &v = ((Client *)c)->secret_key(); std::memcpy(out, v.data(), v.size() * 8); }
void simc_public_key(void *c, uint64_t *out) { auto &v = ((Client *)c)->public_key(); std::memcpy(out, v.data(), v.size() * 8); }
void simc_relin_key(void *c, uint64_t *out) { auto v = ((Client *)c)->make_relin_key(); std::memcpy(out, v.data(), v.size() * 8); }
void simc_galois_key(void *c, uint32_t elt, uint64_t *out) { auto v = ((Client *)c)->make_galois_key(elt); std::memcpy(out, v.data(), v.size() * 8); }
void simc_ckks_encode(void *c, const double *vals, size_t n, double scale, uint64_t *out) { auto v = ((Client *)c)->ckks_encode(vals, n, scale); std::memcpy(out, v.data(), v.size() * 8); }
void simc_ckks_decode(void *c, const uint64_t *plain, size_t L, double scale, double *out) { ((Client *)c)->ckks_decode(plain, L, scale, out); }
void simc_bfv_encode(void *c, const int64_t *vals, size_t n, uint64_t *out) { auto v = ((Client *)c)->bfv_encode(vals, n); std::memcpy(out, v.data(), v.size() * 8); }
void simc_bfv_decode(void *c, const uint64_t *plain, int64_t *out) { ((Client *)c)->bfv_decode(plain, out); }
void simc_encrypt(void *c, const uint64_t *plain, uint64_t *out) { auto v = ((Client *)c)->encrypt(plain); std::memcpy(out, v.data(), v.size() * 8); }
uint64_t simc_keygen_seed(void *c) { return ((Client *)c)->keygen_seed(); }
uint64_t simc_encrypt_seed(void *c) { return ((Client *)c)->encrypt_seed(); }
uint64_t simc_encrypt_index(void *c) { return ((Client *)c)->encrypt_index(); }
void simc_set_encrypt_index(void *c, uint64_t i) { ((Client *)c)->set_encrypt_index(i); }
// the shared counter-based samplers (client/sampler.h), kind 0: ternary, 1: centred binomial
void simc_sample(uint64_t seed, uint64_t stream, size_t count, int kind, int32_t *out)
{
    for (size_t n = 0; n < count; ++n) out[n] = kind == 0 ? client::sample_ternary_at(seed, stream, n) : client::sample_cbd_at(seed, stream, n);
}
void simc_decrypt(void *c, const uint64_t *ct, size_t size, size_t L, uint64_t *out) { auto v = ((Client *)c)->decrypt(ct, size, L); std::memcpy(out, v.data(), v.size() * 8); }
}
