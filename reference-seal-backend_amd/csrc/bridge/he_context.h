// he_context.h — HeContextWrapper: the backend's counterpart of the reference's SEALContextWrapper
// (/root/reference/include/engine/seal_context.h, src/engine/seal_context.cpp): parameters, keys, encoders,
// encrypt/decrypt on the host, and the evaluator on the MI355X through the C ABI of include/he355.h.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/he355.h"
#include "../client/he_client.h"
#include "hebench_cpp.h"

#define HEBENCH_HE_SECURITY_128 0   /* include/engine/seal_types.h:9 */
#define HEB355_ECODE_HE_ERROR 2     /* same value and role as HEBSEAL_ECODE_SEAL_ERROR, seal_types.h:13 */
#define HEB355_ECODE_DEVICE_ERROR 3 /* HIP / no device: this backend has no CPU fallback */

namespace mi355x {

// Objects behind the handles (the reference keeps seal::Plaintext / seal::Ciphertext there).  With the client side on the
// device a plaintext / ciphertext lives in HBM from the moment encode() / encrypt() / operate() produced it: `dev` names
// its slab and position, `data` stays empty until host code asks for it (HeContextWrapper::hostData), and load() / store()
// pass the slab on instead of moving bytes over PCIe and back.
struct DeviceCiphers;
struct DevRef {
    std::shared_ptr<DeviceCiphers> slab;
    uint64_t index = 0;
    explicit operator bool() const { return (bool)slab; }
};
struct Plain {
    mutable std::vector<uint64_t> data; // CKKS: [L][N] NTT form; BFV: [N] mod t (host copy; filled lazily when `dev` is set)
    DevRef dev;
    int L = 0;
    double scale = 1.0;
};
struct Cipher {
    mutable std::vector<uint64_t> data; // [size][L][N] (host copy; filled lazily when `dev` is set)
    DevRef dev;
    int size = 2, L = 0;
    double scale = 1.0;
};
// remote (HBM) slab of n ciphertexts of identical shape
class HeContextWrapper;
class DeviceGroup;
struct DeviceCiphers {
    std::shared_ptr<HeContextWrapper> ctx;
    uint64_t *d = nullptr;
    uint64_t n = 0;
    int size = 2, L = 0;
    double scale = 1.0;
    // multi-device operate() (multi_device.h): a slab that lives on another device of the group is owned through `group` / `device`;
    // a result computed in parts names them here (part p holds results [first, first + slab->n)) and `d` is gathered from them at
    // store() (HeContextWrapper::download), never inside the timed operate()
    std::shared_ptr<DeviceGroup> group;
    int device = 0;
    struct Part { std::shared_ptr<DeviceCiphers> slab; uint64_t first = 0; };
    std::vector<Part> parts;
    DeviceCiphers() = default;
    DeviceCiphers(const DeviceCiphers &) = delete;
    DeviceCiphers &operator=(const DeviceCiphers &) = delete;
    ~DeviceCiphers();
    uint64_t elems_per_ct(uint64_t N) const { return (uint64_t)size * L * N; }
};

class HeContextWrapper : public std::enable_shared_from_this<HeContextWrapper> {
public:
    HEBERROR_DECLARE_CLASS_NAME(HeContextWrapper)
    typedef std::shared_ptr<HeContextWrapper> Ptr;
    // seal_context.cpp:17-39 / 72-127: chain {60, bits x (depth-1), 60}, tc128
    static Ptr createCKKSContext(std::size_t poly_modulus_degree, std::size_t num_coeff_moduli, int coeff_moduli_bits, int scale_bits);
    static Ptr createBFVContext(std::size_t poly_modulus_degree, std::size_t num_coeff_moduli, int coeff_moduli_bits, int plaintext_modulus_bits);
    ~HeContextWrapper();

    he355_ctx *raw() { return m_ctx; }
    // what the context was created from (a device group builds the same context on its other devices)
    int schemeId() const { return m_scheme_id; }
    const std::vector<int32_t> &chainBits() const { return m_chain; }
    int plainBits() const { return m_plain_bits; }
    int deviceOrdinal() const { return m_ordinal; }
    bool relinKeyReady() const { return m_relin; }
    std::vector<uint32_t> galoisKeysReady() const;
    void gatherParts(const std::shared_ptr<DeviceCiphers> &slab); // multi-device result -> one slab on this context's device
    he355::client::Client &client() { return *m_client; }
    const he355::Params &params() const { return *m_params; }
    std::size_t slot_count() const { return m_client->slot_count(); }
    double scale() const { return m_scale; }
    int topLevel() const { return (int)m_params->Ltop; }
    bool isCKKS() const { return m_params->scheme == he355::kSchemeCKKS; }

    // host side
    Plain encodeVector(const std::vector<double> &values);
    Plain encodeVector(const std::vector<std::int64_t> &values);
    // batched encoders / decoders: one device call per operand when a GPU is present (same bits as the host encoders)
    std::vector<Plain> encodeBatch(const std::vector<std::vector<double>> &rows);
    std::vector<Plain> encodeBatch(const std::vector<std::vector<std::int64_t>> &rows);
    std::vector<std::vector<double>> decodeBatchCKKS(const std::vector<Plain> &plains);
    std::vector<std::vector<std::int64_t>> decodeBatchBFV(const std::vector<Plain> &plains);
    // ... writing / downloading only the slots a workload's decode() reads: `ranges` = {first slot, count} (at most 4), the result is flat,
    // [plains.size()][sum of counts] (ckks eltwise .cpp:214-226 copies the first n slots of a result; bfv row .cpp:339-369 the first dim3 of
    // both batching rows).  The values land in a page-locked buffer the context owns and are copied out of it once.
    typedef std::vector<std::pair<std::uint64_t, std::uint64_t>> SlotRanges;
    // the decoded values: in the context's page-locked buffer (small results: valid until the context's next decode) or in a heap block of
    // their own that nobody zero-filled first (large ones) -- read once by the caller, never copied in between
    template <class T> struct Decoded {
        std::unique_ptr<T[]> heap;
        const T *ptr = nullptr;
        std::size_t count = 0;
        const T *data() const { return ptr; }
        const T *begin() const { return ptr; }
        const T *end() const { return ptr + count; }
        std::size_t size() const { return count; }
        const T &operator[](std::size_t i) const { return ptr[i]; }
    };
    Decoded<double> decodeSlotsCKKS(const std::vector<Plain> &plains, const SlotRanges &ranges);
    Decoded<std::int64_t> decodeSlotsBFV(const std::vector<Plain> &plains, const SlotRanges &ranges);
    // Everything the first encode() / encrypt() / decode() of a benchmark would otherwise pay for inside the call: device + streams, the
    // client keys in HBM, the encoders' tables, the client scratch for `batch_hint` objects and the page-locked staging buffer.  Called by
    // the benchmark constructors (createBenchmark is where the reference generates its keys: seal_context.cpp:46-70).
    void prepareClient(std::uint64_t batch_hint = 1, std::uint64_t slots_hint = 1);
    Cipher encrypt(const Plain &plain);
    Plain decrypt(const Cipher &cipher);
    std::vector<Cipher> encryptBatch(const std::vector<Plain> &plains);   // one device call for the whole operand when a GPU is present
    std::vector<Plain> decryptBatch(const std::vector<Cipher> &ciphers);
    bool clientOnDevice(); // true: encrypt/decrypt run on the MI355X (HE355_DEVICE_CLIENT=0 keeps them on the host)

    // device side (lazy device init; uploads the keys a workload declared it needs)
    void ensureDevice();
    void needRelinKey();
    void needRotationKey(int step);
    void needGaloisKey(uint32_t galois_elt);
    void needDefaultGaloisKeys(); // create_galois_keys(): all +-2^k steps and the column swap (seal_context.cpp:69)
    // load() / store(): ciphertexts that already live in one slab are passed on as that slab (no copy); host-resident ones are uploaded
    std::shared_ptr<DeviceCiphers> upload(const std::vector<Cipher> &cts);
    std::shared_ptr<DeviceCiphers> uploadPlains(const std::vector<Plain> &plains); // as size-1 "ciphertexts" (multiply_plain / add_plain operands)
    std::vector<Cipher> download(const std::shared_ptr<DeviceCiphers> &slab);
    // host copy of an object (downloads it once if it lives on the device)
    const uint64_t *hostData(const Plain &p);
    const uint64_t *hostData(const Cipher &c);
    std::shared_ptr<DeviceCiphers> allocResult(uint64_t n, int size, int L, double scale);
    static void check(int code, const char *what); // he355 error -> HEBenchError
    // The ", Number of threads, N" row every description of the reference ends with (e.g. ckks eltwise .cpp:97-112): Latency
    // forces 1 where the reference does, a value <= 0 means every hardware thread.  Kept for report compatibility (the GPU path
    // does not use host threads; the device row follows it).
    static std::string threadsRow(std::uint64_t requested, bool force_one);

private:
    struct Staged { // n objects of `per` words each, contiguous in HBM
        const uint64_t *d = nullptr;
        std::shared_ptr<DeviceCiphers> keep;
    };
    template <class T> Staged stage(const std::vector<T> &items, uint64_t per);
    std::shared_ptr<DeviceCiphers> allocRaw(uint64_t n, uint64_t per); // slab of n objects of `per` words (per a multiple of N)
    HeContextWrapper() = default;
    void init(int scheme, std::size_t N, std::size_t depth, int bits, int plain_bits);
    he355_ctx *m_ctx = nullptr;
    int m_scheme_id = 0, m_plain_bits = 0, m_ordinal = 0;
    std::vector<int32_t> m_chain;
    const he355::Params *m_params = nullptr;
    std::unique_ptr<he355::client::Client> m_client;
    double m_scale = 1.0;
    bool m_device = false, m_relin = false;
    int m_client_dev = -1; // -1: not decided yet
    void *m_pinned = nullptr; // page-locked staging of decode results / encode inputs (he355_host_alloc), sized by prepareClient, never grown inside a phase
    std::uint64_t m_pinned_bytes = 0;
    void *pinned(std::uint64_t bytes, std::uint64_t reserve = 0);
    template <class T> Decoded<T> fetchDecoded(const void *d_src, std::uint64_t count);
    std::map<uint32_t, bool> m_galois;
};

} // namespace mi355x
