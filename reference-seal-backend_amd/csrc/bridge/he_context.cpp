// he_context.cpp — see he_context.h
#include "he_context.h"
#include "multi_device.h"

#include <thread>

#include <algorithm>
#include <cstdlib>

#include <cmath>
#include <cstdlib>
#include <random>

#include "../he355_internal.h"

namespace mi355x {

using hebench::cpp::HEBenchError;

void HeContextWrapper::check(int code, const char *what)
{
    if (code == HE355_OK) return;
    const std::string msg = std::string(what) + ": " + he355_last_error();
    switch (code) {
    case HE355_E_INVALID_ARGS: throw HEBenchError(msg, HEBENCH_ECODE_INVALID_ARGS);
    case HE355_E_PARAMS: throw HEBenchError(msg, HEB355_ECODE_HE_ERROR);
    case HE355_E_DEVICE: throw HEBenchError(msg, HEB355_ECODE_DEVICE_ERROR);
    default: throw HEBenchError(msg, HEBENCH_ECODE_CRITICAL_ERROR);
    }
}

DeviceCiphers::~DeviceCiphers()
{
    if (!d) return;
    if (group) he355_free(group->ctx(device), d); // a slab on another device of a group
    else if (ctx) he355_free(ctx->raw(), d);
}

HeContextWrapper::~HeContextWrapper()
{
    m_client.reset();
    if (m_pinned) (void)he355_host_free(m_ctx, m_pinned);
    if (m_ctx) he355_ctx_destroy(m_ctx);
}
// The page-locked staging buffer: sized once at createBenchmark (prepareClient: the workload's batch x the slots its decode() reads, 1 MiB
// at least, 64 MiB at most) -- page-locking memory costs milliseconds, so it never grows inside a phase; a transfer beyond it goes to
// pageable memory (the runtime's own staging: 1-28 ms for 16 MiB on the MI355X box, profiles/r06_decode_probe.txt).
void *HeContextWrapper::pinned(std::uint64_t bytes, std::uint64_t reserve)
{
    if (!m_pinned) {
        const std::uint64_t want = std::min<std::uint64_t>((std::uint64_t)64 << 20, std::max<std::uint64_t>((std::uint64_t)1 << 20, reserve));
        check(he355_host_alloc(m_ctx, want, &m_pinned), "page-locked buffer");
        m_pinned_bytes = want;
    }
    return bytes <= m_pinned_bytes ? m_pinned : nullptr;
}
template <class T> HeContextWrapper::Decoded<T> HeContextWrapper::fetchDecoded(const void *d_src, std::uint64_t count)
{
    Decoded<T> out;
    out.count = count;
    T *host = static_cast<T *>(pinned(count * sizeof(T)));
    if (!host) {
        out.heap.reset(new T[count]); // (not value-initialised: the download fills it)
        host = out.heap.get();
    }
    check(he355_download(m_ctx, host, d_src, count * sizeof(T)), "download"); // (stream-ordered behind the decode kernels; returns when the bytes are here)
    out.ptr = host;
    return out;
}
void HeContextWrapper::prepareClient(std::uint64_t batch_hint, std::uint64_t slots_hint)
{
    if (!clientOnDevice()) return; // (no device: the host client needs nothing prepared)
    const std::uint64_t slots = std::max<std::uint64_t>(1, std::min<std::uint64_t>(slots_hint, slot_count()));
    (void)pinned(1, std::max<std::uint64_t>(1, batch_hint) * slots * 8);
    // one throw-away encode -> decode of `batch_hint` empty vectors over the slots the workload's decode() reads sizes the client scratch and
    // the pool's size classes, builds the encoder and CRT tables and pages the kernels in: the first real encode() / decode() then cost what
    // the second ones do
    const std::uint64_t n = std::max<std::uint64_t>(1, std::min<std::uint64_t>(batch_hint, 256));
    if (isCKKS()) {
        const std::vector<Plain> pl = encodeBatch(std::vector<std::vector<double>>(n, std::vector<double>(1, 0.0)));
        (void)decodeSlotsCKKS(pl, SlotRanges{{0, slots}});
    } else if (m_params->plain_modulus > 2 && (m_params->plain_modulus - 1) % (2 * m_params->N) == 0) {
        const std::vector<Plain> pl = encodeBatch(std::vector<std::vector<std::int64_t>>(n, std::vector<std::int64_t>(1, 0)));
        (void)decodeSlotsBFV(pl, SlotRanges{{0, slots}});
    }
}

void HeContextWrapper::init(int scheme, std::size_t N, std::size_t depth, int bits, int plain_bits)
{
    // std::vector<int> coeff_modulus = {60}; for i in 1..depth-1 push bits; push 60   (seal_context.cpp:79-82,107-110)
    std::vector<int32_t> chain{60};
    for (std::size_t i = 1; i < depth; ++i) chain.push_back(bits);
    chain.push_back(60);
    // any failure here is what the reference reports as HEBSEAL_ECODE_SEAL_ERROR (seal_context.cpp:94-97,123-126)
    check(he355_ctx_create(scheme, N, chain.data(), chain.size(), plain_bits, 1, &m_ctx), "context creation");
    m_scheme_id = scheme; m_chain = chain; m_plain_bits = plain_bits;
    m_params = he355_internal_params(m_ctx);
    try {
        // Key generation and encryption randomness: seeded from the operating system, as SEAL's default factory is
        // (seal_context.cpp:52-55 create the keys from it); HE355_SEED=<integer> pins the streams (tests, reproducible reports).
        // The sampler itself (client/sampler.h) is a counter-based generator of benchmark grade, not a vetted CSPRNG.
        uint64_t seed;
        if (const char *env = std::getenv("HE355_SEED")) {
            seed = std::strtoull(env, nullptr, 0) ^ (uint64_t)N ^ ((uint64_t)depth << 32);
        } else {
            std::random_device rd;
            seed = ((uint64_t)rd() << 32) ^ (uint64_t)rd();
        }
        m_client.reset(new he355::client::Client(*m_params, seed));
    } catch (const std::exception &ex) {
        throw HEBenchError(ex.what(), HEB355_ECODE_HE_ERROR);
    }
}

HeContextWrapper::Ptr HeContextWrapper::createCKKSContext(std::size_t N, std::size_t depth, int bits, int scale_bits)
{
    Ptr p(new HeContextWrapper());
    p->init(HE355_SCHEME_CKKS, N, depth, bits, 0);
    const int sb = scale_bits < 0 ? 0 : scale_bits;
    p->m_scale = sb == 0 ? 1.0 : std::pow(2.0, sb); // seal_context.cpp:83-84
    return p;
}
HeContextWrapper::Ptr HeContextWrapper::createBFVContext(std::size_t N, std::size_t depth, int bits, int plain_bits)
{
    Ptr p(new HeContextWrapper());
    p->init(HE355_SCHEME_BFV, N, depth, bits, plain_bits);
    return p;
}

Plain HeContextWrapper::encodeVector(const std::vector<double> &values)
{
    if (values.size() > slot_count())
        throw HEBenchError(HEBERROR_MSG_CLASS("Not enough slots available to create packed plaintext"), HEBENCH_ECODE_INVALID_ARGS);
    Plain p;
    p.data = m_client->ckks_encode(values.data(), values.size(), m_scale);
    p.L = topLevel();
    p.scale = m_scale;
    return p;
}
Plain HeContextWrapper::encodeVector(const std::vector<std::int64_t> &values)
{
    if (values.size() > slot_count())
        throw HEBenchError(HEBERROR_MSG_CLASS("Not enough slots available to create packed plaintext"), HEBENCH_ECODE_INVALID_ARGS);
    Plain p;
    p.data = m_client->bfv_encode(values.data(), values.size());
    p.L = topLevel();
    return p;
}
// Batched encoders: on the device when present (bit-identical to the host encoders: tests/test_gpu_client.py)
template <class V> static std::uint64_t max_len(const std::vector<std::vector<V>> &rows)
{
    std::uint64_t m = 0;
    for (const auto &r : rows) m = std::max<std::uint64_t>(m, r.size());
    return m;
}
namespace {
// device temporary of a batch call
struct DevBuf {
    he355_ctx *ctx;
    void *p = nullptr;
    DevBuf(he355_ctx *c, std::uint64_t bytes) : ctx(c) { HeContextWrapper::check(he355_malloc(c, bytes ? bytes : 8, &p), "device allocation"); }
    ~DevBuf() { if (p) (void)he355_free(ctx, p); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <class T> T *as() { return static_cast<T *>(p); }
};
} // namespace

std::shared_ptr<DeviceCiphers> HeContextWrapper::allocRaw(uint64_t n, uint64_t per)
{
    return allocResult(n, 1, (int)(per / m_params->N), 1.0);
}
// Host copy of an object that lives on the device (one download, kept)
const uint64_t *HeContextWrapper::hostData(const Plain &p)
{
    if (p.data.empty() && p.dev) {
        const uint64_t per = p.dev.slab->elems_per_ct(m_params->N);
        p.data.resize(per);
        check(he355_download(m_ctx, p.data.data(), p.dev.slab->d + p.dev.index * per, per * 8), "download");
    }
    return p.data.data();
}
const uint64_t *HeContextWrapper::hostData(const Cipher &c)
{
    if (c.data.empty() && c.dev) {
        const uint64_t per = c.dev.slab->elems_per_ct(m_params->N);
        c.data.resize(per);
        check(he355_download(m_ctx, c.data.data(), c.dev.slab->d + c.dev.index * per, per * 8), "download");
    }
    return c.data.data();
}
// n objects of `per` words as one contiguous device range: the slab they already share (consecutive positions), or a staging
// slab filled by device copies / uploads.
template <class T> HeContextWrapper::Staged HeContextWrapper::stage(const std::vector<T> &items, uint64_t per)
{
    Staged st;
    const uint64_t N = m_params->N;
    bool shared = (bool)items[0].dev;
    for (std::size_t i = 0; shared && i < items.size(); ++i)
        shared = items[i].dev && items[i].dev.slab == items[0].dev.slab && items[i].dev.index == items[0].dev.index + i
                 && items[i].dev.slab->elems_per_ct(N) == per;
    if (shared) {
        st.keep = items[0].dev.slab;
        st.d = st.keep->d + items[0].dev.index * per;
        return st;
    }
    st.keep = allocRaw(items.size(), per);
    for (std::size_t i = 0; i < items.size(); ++i) {
        uint64_t *dst = st.keep->d + i * per;
        if (items[i].dev) {
            if (items[i].dev.slab->elems_per_ct(N) != per) throw HEBenchError(HEBERROR_MSG_CLASS("object shape"), HEBENCH_ECODE_INVALID_ARGS);
            check(he355_copy(m_ctx, dst, items[i].dev.slab->d + items[i].dev.index * per, per * 8), "device copy");
        } else {
            if (items[i].data.size() != per) throw HEBenchError(HEBERROR_MSG_CLASS("object shape"), HEBENCH_ECODE_INVALID_ARGS);
            check(he355_upload(m_ctx, dst, items[i].data.data(), per * 8), "upload");
        }
    }
    st.d = st.keep->d;
    return st;
}

std::vector<Plain> HeContextWrapper::encodeBatch(const std::vector<std::vector<double>> &rows)
{
    std::vector<Plain> out(rows.size());
    const std::uint64_t n = rows.size(), count = max_len(rows);
    if (count > slot_count())
        throw HEBenchError(HEBERROR_MSG_CLASS("Not enough slots available to create packed plaintext"), HEBENCH_ECODE_INVALID_ARGS);
    if (n == 0) return out;
    if (!clientOnDevice() || count == 0) {
        for (std::uint64_t i = 0; i < n; ++i) out[i] = encodeVector(rows[i]);
        return out;
    }
    const std::uint64_t N = m_params->N, pl = (std::uint64_t)topLevel() * N;
    std::vector<double> flat(n * count, 0.0); // missing trailing slots are zeros, as in CKKSEncoder::encode
    for (std::uint64_t i = 0; i < n; ++i) std::copy(rows[i].begin(), rows[i].end(), flat.begin() + i * count);
    DevBuf dv(m_ctx, flat.size() * 8);
    std::shared_ptr<DeviceCiphers> slab = allocRaw(n, pl);
    check(he355_upload(m_ctx, dv.p, flat.data(), flat.size() * 8), "upload");
    check(he355_ckks_encode(m_ctx, n, dv.as<double>(), count, m_scale, slab->d), "encode");
    check(he355_sync(m_ctx), "synchronise");
    for (std::uint64_t i = 0; i < n; ++i) {
        out[i].dev.slab = slab; out[i].dev.index = i; // the plaintexts stay in HBM (encrypt() reads them there)
        out[i].L = topLevel(); out[i].scale = m_scale;
    }
    return out;
}
std::vector<Plain> HeContextWrapper::encodeBatch(const std::vector<std::vector<std::int64_t>> &rows)
{
    std::vector<Plain> out(rows.size());
    const std::uint64_t n = rows.size(), count = max_len(rows);
    if (count > slot_count())
        throw HEBenchError(HEBERROR_MSG_CLASS("Not enough slots available to create packed plaintext"), HEBENCH_ECODE_INVALID_ARGS);
    if (n == 0) return out;
    if (!clientOnDevice() || count == 0) {
        for (std::uint64_t i = 0; i < n; ++i) out[i] = encodeVector(rows[i]);
        return out;
    }
    const std::uint64_t N = m_params->N;
    std::vector<std::int64_t> flat(n * count, 0);
    for (std::uint64_t i = 0; i < n; ++i) std::copy(rows[i].begin(), rows[i].end(), flat.begin() + i * count);
    DevBuf dv(m_ctx, flat.size() * 8);
    std::shared_ptr<DeviceCiphers> slab = allocRaw(n, N);
    check(he355_upload(m_ctx, dv.p, flat.data(), flat.size() * 8), "upload");
    check(he355_bfv_encode(m_ctx, n, dv.as<std::int64_t>(), count, slab->d), "encode");
    check(he355_sync(m_ctx), "synchronise");
    for (std::uint64_t i = 0; i < n; ++i) {
        out[i].dev.slab = slab; out[i].dev.index = i;
        out[i].L = topLevel();
    }
    return out;
}
// decode: slot values of each plaintext (CKKS: N/2 doubles; BFV: N int64)
std::vector<std::vector<double>> HeContextWrapper::decodeBatchCKKS(const std::vector<Plain> &plains)
{
    std::vector<std::vector<double>> out;
    if (plains.empty()) return out;
    bool uniform = true;
    for (const Plain &p : plains) uniform = uniform && p.L == plains[0].L && p.scale == plains[0].scale;
    if (!clientOnDevice() || !uniform || plains[0].L > 16) {
        out.assign(plains.size(), std::vector<double>(slot_count()));
        for (std::size_t i = 0; i < plains.size(); ++i) m_client->ckks_decode(hostData(plains[i]), (size_t)plains[i].L, plains[i].scale, out[i].data());
        return out;
    }
    const int L = plains[0].L;
    const std::uint64_t N = m_params->N, n = plains.size(), pl = (std::uint64_t)L * N, half = N / 2;
    const Staged in = stage(plains, pl);
    DevBuf dv(m_ctx, n * half * 8);
    check(he355_ckks_decode(m_ctx, L, n, in.d, plains[0].scale, dv.as<double>()), "decode");
    check(he355_sync(m_ctx), "synchronise");
    // one transfer for the batch into memory nobody zero-fills first, then one construction per result (no second pass)
    std::unique_ptr<double[]> flat(new double[n * half]);
    check(he355_download(m_ctx, flat.get(), dv.p, n * half * 8), "download");
    out.reserve(n);
    for (std::uint64_t i = 0; i < n; ++i) out.emplace_back(flat.get() + i * half, flat.get() + (i + 1) * half);
    return out;
}
std::vector<std::vector<std::int64_t>> HeContextWrapper::decodeBatchBFV(const std::vector<Plain> &plains)
{
    std::vector<std::vector<std::int64_t>> out;
    if (plains.empty()) return out;
    if (!clientOnDevice()) {
        out.assign(plains.size(), std::vector<std::int64_t>(slot_count()));
        for (std::size_t i = 0; i < plains.size(); ++i) m_client->bfv_decode(hostData(plains[i]), out[i].data());
        return out;
    }
    const std::uint64_t N = m_params->N, n = plains.size();
    const Staged in = stage(plains, N);
    DevBuf dv(m_ctx, n * N * 8);
    check(he355_bfv_decode(m_ctx, n, in.d, dv.as<std::int64_t>()), "decode");
    check(he355_sync(m_ctx), "synchronise");
    std::unique_ptr<std::int64_t[]> flat(new std::int64_t[n * N]); // one transfer for the batch, no zero-fill, one construction per result
    check(he355_download(m_ctx, flat.get(), dv.p, n * N * 8), "download");
    out.reserve(n);
    for (std::uint64_t i = 0; i < n; ++i) out.emplace_back(flat.get() + i * N, flat.get() + (i + 1) * N);
    return out;
}
static std::uint64_t ranges_total(const HeContextWrapper::SlotRanges &ranges, std::uint64_t slots, std::vector<std::uint64_t> &flat)
{
    std::uint64_t total = 0;
    if (ranges.empty() || ranges.size() > 4) throw HEBenchError("decode: 1 to 4 slot ranges", HEBENCH_ECODE_INVALID_ARGS);
    for (const auto &r : ranges) {
        if (r.first > slots || r.second > slots - r.first) throw HEBenchError("decode: slot range outside the encoder's slots", HEBENCH_ECODE_INVALID_ARGS);
        flat.push_back(r.first); flat.push_back(r.second);
        total += r.second;
    }
    return total;
}
HeContextWrapper::Decoded<double> HeContextWrapper::decodeSlotsCKKS(const std::vector<Plain> &plains, const SlotRanges &ranges)
{
    std::vector<std::uint64_t> flat;
    const std::uint64_t total = ranges_total(ranges, slot_count(), flat), n = plains.size();
    Decoded<double> out;
    if (!n || !total) return out;
    bool uniform = true;
    for (const Plain &p : plains) uniform = uniform && p.L == plains[0].L && p.scale == plains[0].scale;
    if (!clientOnDevice() || !uniform || plains[0].L > 16) { // host decoder: all slots, then the wanted ones
        out.heap.reset(new double[n * total]);
        out.ptr = out.heap.get(); out.count = n * total;
        std::vector<double> all(slot_count());
        for (std::size_t i = 0; i < n; ++i) {
            m_client->ckks_decode(hostData(plains[i]), (size_t)plains[i].L, plains[i].scale, all.data());
            double *dst = out.heap.get() + i * total;
            for (const auto &r : ranges) dst = std::copy(all.begin() + r.first, all.begin() + r.first + r.second, dst);
        }
        return out;
    }
    const int L = plains[0].L;
    const std::uint64_t N = m_params->N, pl = (std::uint64_t)L * N, bytes = n * total * 8;
    const Staged in = stage(plains, pl);
    DevBuf dv(m_ctx, bytes);
    check(he355_ckks_decode_slots(m_ctx, L, n, in.d, plains[0].scale, flat.data(), ranges.size(), dv.as<double>()), "decode");
    return fetchDecoded<double>(dv.p, n * total);
}
HeContextWrapper::Decoded<std::int64_t> HeContextWrapper::decodeSlotsBFV(const std::vector<Plain> &plains, const SlotRanges &ranges)
{
    std::vector<std::uint64_t> flat;
    const std::uint64_t total = ranges_total(ranges, slot_count(), flat), n = plains.size();
    Decoded<std::int64_t> out;
    if (!n || !total) return out;
    if (!clientOnDevice()) {
        out.heap.reset(new std::int64_t[n * total]);
        out.ptr = out.heap.get(); out.count = n * total;
        std::vector<std::int64_t> all(slot_count());
        for (std::size_t i = 0; i < n; ++i) {
            m_client->bfv_decode(hostData(plains[i]), all.data());
            std::int64_t *dst = out.heap.get() + i * total;
            for (const auto &r : ranges) dst = std::copy(all.begin() + r.first, all.begin() + r.first + r.second, dst);
        }
        return out;
    }
    const std::uint64_t N = m_params->N, bytes = n * total * 8;
    const Staged in = stage(plains, N);
    DevBuf dv(m_ctx, bytes);
    check(he355_bfv_decode_slots(m_ctx, n, in.d, flat.data(), ranges.size(), dv.as<std::int64_t>()), "decode");
    return fetchDecoded<std::int64_t>(dv.p, n * total);
}
// Client side: on the MI355X when one is present (he355_encrypt / he355_decrypt: same bits as the host code below for the same
// randomness counter — tests/test_gpu_client.py), on the host otherwise, as in the reference, whose Encryptor / Decryptor are
// host code.  This is NOT an evaluator fallback: load() and operate() still need the device.
bool HeContextWrapper::clientOnDevice()
{
    if (m_client_dev < 0) {
        int n = 0;
        const char *env = getenv("HE355_DEVICE_CLIENT");
        m_client_dev = (he355_device_count(&n) == 0 && n > 0 && !(env && env[0] == '0')) ? 1 : 0;
        if (m_client_dev) {
            ensureDevice();
            check(he355_set_public_key(m_ctx, m_client->public_key().data()), "public key upload");
            check(he355_set_secret_key(m_ctx, m_client->secret_key().data()), "secret key upload");
        }
    }
    return m_client_dev == 1;
}
std::vector<Cipher> HeContextWrapper::encryptBatch(const std::vector<Plain> &plains)
{
    std::vector<Cipher> out(plains.size());
    if (plains.empty()) return out;
    const int L = topLevel();
    if (!clientOnDevice()) {
        for (std::size_t i = 0; i < plains.size(); ++i) out[i] = encrypt(plains[i]);
        return out;
    }
    const std::uint64_t N = m_params->N, n = plains.size(), pl = isCKKS() ? (std::uint64_t)L * N : N;
    const Staged in = stage(plains, pl);
    std::shared_ptr<DeviceCiphers> slab = allocResult(n, 2, L, plains[0].scale);
    const std::uint64_t first = m_client->encrypt_index();
    check(he355_encrypt(m_ctx, n, in.d, m_client->encrypt_seed(), first, slab->d), "encrypt");
    m_client->set_encrypt_index(first + n); // the host counter moves on exactly as if it had encrypted them
    check(he355_sync(m_ctx), "synchronise");
    for (std::uint64_t i = 0; i < n; ++i) {
        out[i].dev.slab = slab; out[i].dev.index = i; // the ciphertexts stay in HBM: load() hands the slab to operate()
        out[i].size = 2; out[i].L = L; out[i].scale = plains[i].scale;
    }
    return out;
}
std::vector<Plain> HeContextWrapper::decryptBatch(const std::vector<Cipher> &ciphers)
{
    std::vector<Plain> out(ciphers.size());
    if (ciphers.empty()) return out;
    bool uniform = true;
    for (const Cipher &c : ciphers) uniform = uniform && c.size == ciphers[0].size && c.L == ciphers[0].L;
    if (!clientOnDevice() || !uniform || ciphers[0].size < 2 || ciphers[0].size > 3 || (!isCKKS() && ciphers[0].L > 16)) {
        for (std::size_t i = 0; i < ciphers.size(); ++i) out[i] = decrypt(ciphers[i]);
        return out;
    }
    try {
        const int L = ciphers[0].L, size = ciphers[0].size;
        const std::uint64_t N = m_params->N, n = ciphers.size(), cl = (std::uint64_t)size * L * N, pl = isCKKS() ? (std::uint64_t)L * N : N;
        const Staged in = stage(ciphers, cl);
        std::shared_ptr<DeviceCiphers> slab = allocRaw(n, pl);
        check(he355_decrypt(m_ctx, L, size, n, in.d, slab->d), "decrypt");
        check(he355_sync(m_ctx), "synchronise");
        for (std::uint64_t i = 0; i < n; ++i) {
            out[i].dev.slab = slab; out[i].dev.index = i;
            out[i].L = L; out[i].scale = ciphers[i].scale;
        }
        return out;
    } catch (const HEBenchError &) {
        throw;
    } catch (const std::exception &ex) {
        throw HEBenchError(ex.what(), HEB355_ECODE_HE_ERROR); // seal_context.cpp:166-169
    }
}
Cipher HeContextWrapper::encrypt(const Plain &plain)
{
    if (clientOnDevice()) return encryptBatch(std::vector<Plain>{plain})[0];
    Cipher c;
    c.data = m_client->encrypt(hostData(plain));
    c.size = 2;
    c.L = topLevel();
    c.scale = plain.scale;
    return c;
}
Plain HeContextWrapper::decrypt(const Cipher &cipher)
{
    if (clientOnDevice() && cipher.size >= 2 && cipher.size <= 3 && (isCKKS() || cipher.L <= 16)) return decryptBatch(std::vector<Cipher>{cipher})[0];
    try {
        Plain p;
        p.data = m_client->decrypt(hostData(cipher), (size_t)cipher.size, (size_t)cipher.L);
        p.L = cipher.L;
        p.scale = cipher.scale;
        return p;
    } catch (const std::exception &ex) {
        throw HEBenchError(ex.what(), HEB355_ECODE_HE_ERROR); // seal_context.cpp:166-169
    }
}

void HeContextWrapper::ensureDevice()
{
    if (m_device) return;
    // one harness process per GPU: HE355_DEVICE selects the ordinal, otherwise LOCAL_RANK (torchrun / mpirun conventions) modulo
    // the number of devices, otherwise device 0
    int ordinal = 0, count = 0;
    const char *dev = getenv("HE355_DEVICE"), *lr = getenv("LOCAL_RANK");
    if (dev && *dev) ordinal = std::atoi(dev);
    else if (lr && *lr && he355_device_count(&count) == 0 && count > 0) ordinal = std::atoi(lr) % count;
    check(he355_device_init(m_ctx, ordinal), "device initialisation");
    m_ordinal = ordinal;
    m_device = true;
}
std::vector<uint32_t> HeContextWrapper::galoisKeysReady() const
{
    std::vector<uint32_t> v;
    for (const auto &kv : m_galois) v.push_back(kv.first);
    return v;
}
// store() of a multi-device result: the parts come home over xGMI (hipMemcpyPeer), outside the timed operate()
void HeContextWrapper::gatherParts(const std::shared_ptr<DeviceCiphers> &slab)
{
    if (slab->parts.empty()) return;
    const uint64_t per = slab->elems_per_ct(m_params->N);
    for (const auto &p : slab->parts) {
        if (!p.slab->n) continue;
        he355_ctx *src = p.slab->group ? p.slab->group->ctx(p.slab->device) : m_ctx;
        check(he355_copy_peer(m_ctx, slab->d + p.first * per, src, p.slab->d, p.slab->n * per * 8), "gather of a result part");
    }
    slab->parts.clear();
}
void HeContextWrapper::needRelinKey()
{
    ensureDevice();
    if (m_relin) return;
    if (clientOnDevice()) { // KeyGenerator on the device: same key as make_relin_key() (tests/test_gpu_client.py)
        check(he355_keygen_relin(m_ctx, m_client->keygen_seed()), "relinearization key generation");
    } else {
        const std::vector<uint64_t> k = m_client->make_relin_key();
        check(he355_set_relin_key(m_ctx, k.data()), "relinearization key upload");
    }
    m_relin = true;
}
void HeContextWrapper::needRotationKey(int step)
{
    const uint32_t elt = he355_galois_elt_from_step(m_ctx, step);
    if (!elt) throw HEBenchError(HEBERROR_MSG_CLASS("step count too large"), HEBENCH_ECODE_INVALID_ARGS);
    needGaloisKey(elt);
}
void HeContextWrapper::needGaloisKey(uint32_t elt)
{
    ensureDevice();
    if (m_galois.count(elt)) return;
    if (clientOnDevice()) {
        check(he355_keygen_galois(m_ctx, elt, m_client->keygen_seed()), "Galois key generation");
    } else {
        const std::vector<uint64_t> k = m_client->make_galois_key(elt);
        check(he355_set_galois_key(m_ctx, elt, k.data()), "Galois key upload");
    }
    m_galois[elt] = true;
}
void HeContextWrapper::needDefaultGaloisKeys()
{
    uint32_t elts[64];
    const uint64_t n = he355_galois_elts_all(m_ctx, elts, 64);
    for (uint64_t i = 0; i < n && i < 64; ++i) needGaloisKey(elts[i]);
}

std::string HeContextWrapper::threadsRow(std::uint64_t requested, bool force_one)
{
    std::uint64_t n = force_one ? 1 : requested;
    if (n == 0) n = std::max(1u, std::thread::hardware_concurrency()); // omp_get_max_threads() in the reference
    return ", Number of threads, " + std::to_string(n);
}

std::shared_ptr<DeviceCiphers> HeContextWrapper::allocResult(uint64_t n, int size, int L, double scale)
{
    ensureDevice();
    auto s = std::make_shared<DeviceCiphers>();
    s->ctx = shared_from_this();
    s->n = n; s->size = size; s->L = L; s->scale = scale;
    void *d = nullptr;
    check(he355_malloc(m_ctx, n * s->elems_per_ct(m_params->N) * 8, &d), "device allocation");
    s->d = static_cast<uint64_t *>(d);
    return s;
}
std::shared_ptr<DeviceCiphers> HeContextWrapper::upload(const std::vector<Cipher> &cts)
{
    if (cts.empty()) throw HEBenchError(HEBERROR_MSG_CLASS("empty operand"), HEBENCH_ECODE_INVALID_ARGS);
    ensureDevice();
    const uint64_t per = (uint64_t)cts[0].size * cts[0].L * m_params->N;
    for (const Cipher &c : cts)
        if (c.size != cts[0].size || c.L != cts[0].L || (!c.dev && c.data.size() != per))
            throw HEBenchError(HEBERROR_MSG_CLASS("operand ciphertexts differ in shape"), HEBENCH_ECODE_INVALID_ARGS);
    const Staged st = stage(cts, per);
    const bool whole = st.d == st.keep->d && st.keep->n == cts.size();
    if (whole && st.keep->size == cts[0].size && st.keep->L == cts[0].L) return st.keep; // the operand's own slab: nothing moves
    if (whole && st.keep.use_count() == 1) { // a staging slab stage() has just filled: give it the operand's shape
        st.keep->size = cts[0].size; st.keep->L = cts[0].L; st.keep->scale = cts[0].scale;
        check(he355_sync(m_ctx), "synchronise");
        return st.keep;
    }
    auto s = allocResult(cts.size(), cts[0].size, cts[0].L, cts[0].scale); // a sub-range of a larger slab
    check(he355_copy(m_ctx, s->d, st.d, cts.size() * per * 8), "device copy");
    check(he355_sync(m_ctx), "synchronise");
    return s;
}
std::shared_ptr<DeviceCiphers> HeContextWrapper::uploadPlains(const std::vector<Plain> &plains)
{
    std::vector<Cipher> tmp(plains.size());
    for (std::size_t i = 0; i < plains.size(); ++i) {
        tmp[i].dev = plains[i].dev;
        if (!plains[i].dev) tmp[i].data = plains[i].data;
        tmp[i].size = 1; tmp[i].L = plains[i].L; tmp[i].scale = plains[i].scale;
    }
    return upload(tmp);
}
std::vector<Cipher> HeContextWrapper::download(const std::shared_ptr<DeviceCiphers> &slab)
{
    std::vector<Cipher> out(slab->n);
    gatherParts(slab);
    check(he355_sync(m_ctx), "synchronise");
    for (uint64_t i = 0; i < slab->n; ++i) {
        out[i].size = slab->size; out[i].L = slab->L; out[i].scale = slab->scale;
        if (clientOnDevice()) { // decrypt() reads the result where operate() left it
            out[i].dev.slab = slab; out[i].dev.index = i;
        } else {
            const uint64_t per = slab->elems_per_ct(m_params->N);
            out[i].data.resize(per);
            check(he355_download(m_ctx, out[i].data.data(), slab->d + i * per, per * 8), "download");
        }
    }
    return out;
}

} // namespace mi355x
