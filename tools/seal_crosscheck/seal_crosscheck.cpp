// seal_crosscheck.cpp — OPT-IN cross-check against a real Microsoft SEAL (v3.7.x) install; test infrastructure, not product.
//
// The reference links SEAL through -DSEAL_INSTALL_DIR (/root/reference/cmake/utils/import-library.cmake:54-58,
// cmake/third-party/SEAL.cmake:5-14); SEAL is not in this image, so this tool CANNOT be built or run here and is written against
// the public SEAL 3.7 API from knowledge of it [UPSTREAM-UNVERIFIED].  Where a SEAL install exists:
//     make -C tools/seal_crosscheck SEAL_INSTALL_DIR=/path/to/seal/prefix
//     SEAL_INSTALL_DIR=/path/to/seal/prefix python -m pytest tests/test_seal_crosscheck.py
// It builds a SEALContext with the reference's parameter rule ({60, bits x (depth-1), 60}, tc128: seal_context.cpp:79-90,107-119),
// generates real keys, encrypts two random vectors and runs the evaluator calls of the hot path
// (multiply, relinearize_inplace, rescale_to_next_inplace, rotate_vector / rotate_rows), then dumps inputs, keys and outputs as raw
// little-endian u64 arrays in seal::Ciphertext::data() / KSwitchKeys order.  tests/test_seal_crosscheck.py feeds the dumped inputs and
// keys through the oracle (and, on the GPU box, through the HIP path) and compares the outputs bit for bit: with that test green,
// parity against SEAL itself is pinned and DESIGN.md's "parity unpinned" note can be dropped.
//
// usage: seal_crosscheck <ckks|bfv> <N> <depth> <coeff_bits> <extra_bits> <outdir>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "seal/seal.h"

using namespace seal;

static void dump(const std::string &path, const std::uint64_t *p, std::size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(p), (std::streamsize)(n * 8));
}
static void dump_ct(const std::string &dir, const std::string &name, const Ciphertext &ct)
{
    dump(dir + "/" + name + ".bin", ct.data(), ct.size() * ct.coeff_modulus_size() * ct.poly_modulus_degree());
}
// KSwitchKeys entry (one key = vector of PublicKey, one per digit): [digit][2][K][N], NTT form
static void dump_kswitch(const std::string &dir, const std::string &name, const std::vector<PublicKey> &key)
{
    std::ofstream f(dir + "/" + name + ".bin", std::ios::binary);
    for (const PublicKey &pk : key) {
        const Ciphertext &c = pk.data();
        f.write(reinterpret_cast<const char *>(c.data()), (std::streamsize)(c.size() * c.coeff_modulus_size() * c.poly_modulus_degree() * 8));
    }
}

int main(int argc, char **argv)
{
    if (argc != 7) {
        std::fprintf(stderr, "usage: %s <ckks|bfv> <N> <depth> <coeff_bits> <extra_bits> <outdir>\n", argv[0]);
        return 2;
    }
    const bool ckks = std::string(argv[1]) == "ckks";
    const std::size_t N = std::stoull(argv[2]), depth = std::stoull(argv[3]);
    const int bits = std::stoi(argv[4]), extra = std::stoi(argv[5]);
    const std::string dir = argv[6];

    EncryptionParameters parms(ckks ? scheme_type::ckks : scheme_type::bfv);
    parms.set_poly_modulus_degree(N);
    std::vector<int> chain{60};
    for (std::size_t i = 1; i < depth; ++i) chain.push_back(bits);
    chain.push_back(60);
    parms.set_coeff_modulus(CoeffModulus::Create(N, chain));
    if (!ckks) parms.set_plain_modulus(PlainModulus::Batching(N, extra));
    SEALContext context(parms, true, sec_level_type::tc128);

    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    RelinKeys rk;
    keygen.create_relin_keys(rk);
    GaloisKeys gk;
    keygen.create_galois_keys(std::vector<int>{1, -1, 4}, gk);
    Encryptor encryptor(context, pk);
    Evaluator evaluator(context);

    std::mt19937_64 rng(1234);
    Ciphertext a, b;
    if (ckks) {
        CKKSEncoder encoder(context);
        std::uniform_real_distribution<double> u(-1.0, 1.0);
        std::vector<double> x(encoder.slot_count()), y(encoder.slot_count());
        for (auto &v : x) v = u(rng);
        for (auto &v : y) v = u(rng);
        Plaintext px, py;
        encoder.encode(x, std::pow(2.0, extra), px);
        encoder.encode(y, std::pow(2.0, extra), py);
        encryptor.encrypt(px, a);
        encryptor.encrypt(py, b);
    } else {
        BatchEncoder encoder(context);
        std::vector<std::int64_t> x(encoder.slot_count()), y(encoder.slot_count());
        for (auto &v : x) v = (std::int64_t)(rng() % 1000) - 500;
        for (auto &v : y) v = (std::int64_t)(rng() % 1000) - 500;
        Plaintext px, py;
        encoder.encode(x, px);
        encoder.encode(y, py);
        encryptor.encrypt(px, a);
        encryptor.encrypt(py, b);
    }
    dump_ct(dir, "a", a);
    dump_ct(dir, "b", b);
    dump_kswitch(dir, "relin", rk.key(2)); // the key for s^2
    const auto &key_parms = context.key_context_data()->parms();
    auto galois_tool = context.key_context_data()->galois_tool();
    std::string elts;
    for (int step : {1, -1, 4}) {
        const std::uint32_t elt = galois_tool->get_elt_from_step(step);
        dump_kswitch(dir, "galois_" + std::to_string(elt), gk.key(elt));
        elts += (elts.empty() ? "" : ", ") + std::string("\"") + std::to_string(step) + "\": " + std::to_string(elt);
    }

    Ciphertext c;
    evaluator.add(a, b, c);
    dump_ct(dir, "out_add", c);
    evaluator.multiply(a, b, c);
    dump_ct(dir, "out_multiply", c);
    evaluator.relinearize_inplace(c, rk);
    dump_ct(dir, "out_multiply_relin", c);
    if (ckks) {
        evaluator.rescale_to_next_inplace(c);
        dump_ct(dir, "out_multiply_relin_rescale", c);
        evaluator.rotate_vector(a, 1, gk, c);
        dump_ct(dir, "out_rotate_1", c);
        evaluator.rotate_vector(a, 3, gk, c); // no key for 3: NAF -1, +4
        dump_ct(dir, "out_rotate_3", c);
    } else {
        evaluator.rotate_rows(a, 1, gk, c);
        dump_ct(dir, "out_rotate_1", c);
        evaluator.rotate_rows(a, 3, gk, c);
        dump_ct(dir, "out_rotate_3", c);
    }

    std::ofstream meta(dir + "/meta.json");
    meta << "{\"scheme\": \"" << (ckks ? "ckks" : "bfv") << "\", \"N\": " << N << ", \"depth\": " << depth << ", \"coeff_bits\": " << bits
         << ", \"extra_bits\": " << extra << ", \"plain_modulus\": " << (ckks ? 0 : parms.plain_modulus().value()) << ", \"primes\": [";
    const auto &mods = key_parms.coeff_modulus();
    for (std::size_t i = 0; i < mods.size(); ++i) meta << (i ? ", " : "") << mods[i].value();
    meta << "], \"galois_elts\": {" << elts << "}, \"seal_version\": \"" << SEAL_VERSION_MAJOR << "." << SEAL_VERSION_MINOR << "." << SEAL_VERSION_PATCH
         << "\"}\n";
    return 0;
}
