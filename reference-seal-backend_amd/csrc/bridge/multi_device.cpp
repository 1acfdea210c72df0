// multi_device.cpp — see multi_device.h
#include "multi_device.h"

#include <algorithm>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>

using namespace mi355x;
using hebench::cpp::HEBenchError;

static int logical_device_count()
{
    int n = 0;
    if (he355_device_count(&n) != 0 || n < 1) return 0;
    if (const char *env = std::getenv("HE355_LOGICAL_DEVICES")) {
        const int k = std::atoi(env);
        if (k > 0) return k; // logical device d runs on physical device d mod n
    }
    return n;
}

int DeviceGroup::resolveCount(std::uint64_t requested)
{
    std::uint64_t n = requested;
    if (n == 0) {
        const char *env = std::getenv("HE355_NUM_DEVICES");
        n = env && std::atoi(env) > 0 ? (std::uint64_t)std::atoi(env) : 1;
    }
    const int avail = logical_device_count();
    if (avail >= 1 && n > (std::uint64_t)avail) n = (std::uint64_t)avail;
    return (int)std::max<std::uint64_t>(1, n);
}

void DeviceGroup::rowsOf(std::uint64_t b0, int n_devices, int d, std::uint64_t &first, std::uint64_t &count)
{
    const std::uint64_t base = b0 / (std::uint64_t)n_devices, extra = b0 % (std::uint64_t)n_devices;
    count = base + ((std::uint64_t)d < extra ? 1 : 0);
    first = (std::uint64_t)d * base + std::min<std::uint64_t>((std::uint64_t)d, extra);
}

std::shared_ptr<DeviceGroup> DeviceGroup::create(const HeContextWrapper::Ptr &primary, int n_devices)
{
    std::shared_ptr<DeviceGroup> g(new DeviceGroup());
    g->m_primary = primary;
    primary->ensureDevice();
    g->m_ctx.push_back(primary->raw());
    g->m_relin.push_back(true);
    g->m_galois.emplace_back();
    g->m_loaded.assign((std::size_t)n_devices, {0, 0});
    int physical = 0;
    HeContextWrapper::check(he355_device_count(&physical), "device count");
    const std::vector<int32_t> &chain = primary->chainBits();
    for (int d = 1; d < n_devices; ++d) {
        he355_ctx *c = nullptr;
        HeContextWrapper::check(he355_ctx_create(primary->schemeId(), primary->params().N, chain.data(), chain.size(), primary->plainBits(), 1, &c),
                                "context creation on another device");
        g->m_ctx.push_back(c);
        g->m_relin.push_back(false);
        g->m_galois.emplace_back();
        // the group's devices follow the primary's ordinal (one harness process may have been given a starting device)
        HeContextWrapper::check(he355_device_init(c, (primary->deviceOrdinal() + d) % physical), "device initialisation");
        HeContextWrapper::check(he355_set_secret_key(c, primary->client().secret_key().data()), "secret key upload");
    }
    return g;
}

DeviceGroup::~DeviceGroup()
{
    for (std::size_t d = 1; d < m_ctx.size(); ++d)
        if (m_ctx[d]) he355_ctx_destroy(m_ctx[d]);
}

void DeviceGroup::syncKeys()
{
    const uint64_t seed = m_primary->client().keygen_seed();
    const std::vector<uint32_t> elts = m_primary->galoisKeysReady();
    for (int d = 1; d < size(); ++d) {
        if (m_primary->relinKeyReady() && !m_relin[(std::size_t)d]) {
            HeContextWrapper::check(he355_keygen_relin(m_ctx[(std::size_t)d], seed), "relinearization key generation");
            m_relin[(std::size_t)d] = true;
        }
        for (uint32_t e : elts) {
            auto &have = m_galois[(std::size_t)d];
            if (std::find(have.begin(), have.end(), e) != have.end()) continue;
            HeContextWrapper::check(he355_keygen_galois(m_ctx[(std::size_t)d], e, seed), "Galois key generation");
            have.push_back(e);
        }
    }
}

std::shared_ptr<DeviceCiphers> DeviceGroup::alloc(int d, std::uint64_t n, int size, int L, double scale)
{
    if (d == 0) return m_primary->allocResult(n, size, L, scale);
    auto s = std::make_shared<DeviceCiphers>();
    s->ctx = m_primary;
    s->group = shared_from_this();
    s->device = d;
    s->n = n; s->size = size; s->L = L; s->scale = scale;
    void *p = nullptr;
    HeContextWrapper::check(he355_malloc(m_ctx[(std::size_t)d], std::max<std::uint64_t>(1, n) * s->elems_per_ct(m_primary->params().N) * 8, &p),
                            "device allocation");
    s->d = static_cast<uint64_t *>(p);
    return s;
}

std::shared_ptr<DeviceCiphers> DeviceGroup::replicate(int d, const std::shared_ptr<DeviceCiphers> &src)
{
    return replicateRows(d, src, 0, src->n, 1);
}

static std::mutex g_last_mtx;
static std::vector<std::array<std::uint64_t, 2>> g_last_loaded;

std::shared_ptr<DeviceCiphers> DeviceGroup::replicateRows(int d, const std::shared_ptr<DeviceCiphers> &src, std::uint64_t first, std::uint64_t count, int operand)
{
    if (first + count > src->n) throw HEBenchError(HEBERROR_MSG("row block out of range"), HEBENCH_ECODE_CRITICAL_ERROR);
    if (d == 0) return src;
    auto s = alloc(d, count, src->size, src->L, src->scale);
    const std::uint64_t per = src->elems_per_ct(m_primary->params().N);
    if (count)
        HeContextWrapper::check(he355_copy_peer(m_ctx[(std::size_t)d], s->d, m_ctx[0], src->d + first * per, count * per * 8), "operand replication");
    m_loaded[(std::size_t)d][operand & 1] += count * per * 8;
    {
        std::scoped_lock<std::mutex> lock(g_last_mtx);
        g_last_loaded = m_loaded;
    }
    return s;
}

std::uint64_t DeviceGroup::lastLoadedBytes(int d, int operand)
{
    std::scoped_lock<std::mutex> lock(g_last_mtx);
    return d >= 0 && (std::size_t)d < g_last_loaded.size() ? g_last_loaded[(std::size_t)d][operand & 1] : 0;
}

void DeviceGroup::parallel(const std::function<void(int)> &fn)
{
    if (size() == 1) { fn(0); return; }
    std::mutex mtx;
    std::exception_ptr p_ex; // as the reference captures exceptions of its OpenMP region (ckks eltwise .cpp:323-324, 351-361)
    std::vector<std::thread> threads;
    for (int d = 0; d < size(); ++d)
        threads.emplace_back([&, d] {
            try {
                fn(d);
            } catch (...) {
                std::scoped_lock<std::mutex> lock(mtx);
                if (!p_ex) p_ex = std::current_exception();
            }
        });
    for (auto &t : threads) t.join();
    if (p_ex) std::rethrow_exception(p_ex);
}

extern "C" std::uint64_t he355_bridge_group_load_bytes(int device, int operand) { return DeviceGroup::lastLoadedBytes(device, operand); }
