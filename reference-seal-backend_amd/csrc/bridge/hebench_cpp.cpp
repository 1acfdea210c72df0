// hebench_cpp.cpp — the extern "C" HEBench API-Bridge entry points (include/hebench_api_bridge.h), forwarding to
// the backend's BaseEngine / BenchmarkDescription / BaseBenchmark objects exactly as hebench_cpp does for the
// reference (CMakeLists.txt:103-108 whole-archive links that wrapper into libhebench_seal_backend.so).
// Error convention: exceptions become ErrorCode; the message is kept for getLastErrorDescription.
#include "hebench_cpp.h"

#include <algorithm>
#include <mutex>
#include <sstream>

namespace hebench {
namespace cpp {

std::string BenchmarkDescription::getBenchmarkDescription(const APIBridge::WorkloadParams *p_w_params) const
{
    std::stringstream ss;
    if (p_w_params && p_w_params->count > 0) {
        ss << ", Workload parameters";
        for (std::uint64_t i = 0; i < p_w_params->count; ++i) {
            const auto &p = p_w_params->params[i];
            ss << std::endl << ", , " << p.name << ", ";
            switch (p.data_type) {
            case APIBridge::WorkloadParamType::Int64: ss << p.i_param; break;
            case APIBridge::WorkloadParamType::UInt64: ss << p.u_param; break;
            default: ss << p.f_param; break;
            }
        }
    }
    return ss.str();
}

std::string BaseEngine::schemeName(APIBridge::Scheme s) const
{
    auto it = m_schemes.find(s);
    return it == m_schemes.end() ? std::string() : it->second;
}
std::string BaseEngine::securityName(APIBridge::Scheme, APIBridge::Security sec) const
{
    auto it = m_security.find(sec);
    return it == m_security.end() ? std::string() : it->second;
}
std::string BaseEngine::errorDescription(APIBridge::ErrorCode code) const
{
    switch (code) {
    case HEBENCH_ECODE_SUCCESS: return "Success";
    case HEBENCH_ECODE_INVALID_ARGS: return "Invalid argument";
    case HEBENCH_ECODE_CRITICAL_ERROR: return "Critical error";
    default: break;
    }
    auto it = m_errors.find(code);
    return it == m_errors.end() ? std::string("Unknown error") : it->second;
}

} // namespace cpp
} // namespace hebench

using namespace hebench::cpp;
namespace AB = hebench::APIBridge;

namespace {

std::mutex g_mtx;
std::string g_last_error_no_engine; // errors raised before an engine exists
BaseEngine *g_last_engine = nullptr;

void record(BaseEngine *e, const std::string &msg, AB::ErrorCode code)
{
    std::lock_guard<std::mutex> lk(g_mtx);
    if (e) e->setLastError(msg, code);
    else g_last_error_no_engine = msg;
}

template <class F> AB::ErrorCode guard(BaseEngine *&e, F &&f)
{
    try {
        f();
        return HEBENCH_ECODE_SUCCESS;
    } catch (const HEBenchError &err) {
        record(e, err.what(), err.getErrorCode());
        return err.getErrorCode();
    } catch (const std::exception &err) {
        record(e, err.what(), HEBENCH_ECODE_CRITICAL_ERROR);
        return HEBENCH_ECODE_CRITICAL_ERROR;
    } catch (...) {
        record(e, "Unexpected error.", HEBENCH_ECODE_CRITICAL_ERROR);
        return HEBENCH_ECODE_CRITICAL_ERROR;
    }
}

BaseEngine *engine_of(AB::Handle h)
{
    if ((h.tag & BaseEngine::tag) == 0 || !h.p) throw HEBenchError("Invalid engine handle.", HEBENCH_ECODE_INVALID_ARGS);
    return static_cast<BaseEngine *>(h.p);
}
BenchmarkDescription *desc_of(AB::Handle h)
{
    if ((h.tag & BenchmarkDescription::tag) == 0 || !h.p) throw HEBenchError("Invalid benchmark description handle.", HEBENCH_ECODE_INVALID_ARGS);
    return static_cast<BenchmarkDescription *>(h.p);
}
BaseBenchmark *bench_of(AB::Handle h)
{
    if ((h.tag & BaseBenchmark::tag) == 0 || !h.p) throw HEBenchError("Invalid benchmark handle.", HEBENCH_ECODE_INVALID_ARGS);
    return static_cast<BaseBenchmark *>(h.p);
}
std::uint64_t copy_string(const std::string &s, char *dst, std::uint64_t size)
{
    const std::uint64_t needed = s.size() + 1;
    if (dst && size > 0) {
        const std::uint64_t n = std::min<std::uint64_t>(size - 1, s.size());
        std::memcpy(dst, s.data(), n);
        dst[n] = 0;
    }
    return needed;
}

} // namespace

namespace hebench {
namespace APIBridge {
extern "C" {

ErrorCode initEngine(Handle *h_engine, const int8_t *p_buffer, uint64_t size)
{
    BaseEngine *none = nullptr;
    return guard(none, [&] {
        if (!h_engine) throw HEBenchError("Invalid null handle.", HEBENCH_ECODE_INVALID_ARGS);
        BaseEngine *e = createEngine(p_buffer, size);
        h_engine->p = e;
        h_engine->size = sizeof(BaseEngine *);
        h_engine->tag = BaseEngine::tag;
        std::lock_guard<std::mutex> lk(g_mtx);
        g_last_engine = e;
    });
}

ErrorCode destroyHandle(Handle h)
{
    BaseEngine *none = nullptr;
    return guard(none, [&] {
        if (!h.p) return;
        if (h.tag & BaseEngine::tag) {
            {
                std::lock_guard<std::mutex> lk(g_mtx);
                if (g_last_engine == h.p) g_last_engine = nullptr;
            }
            destroyEngine(static_cast<BaseEngine *>(h.p));
        } else if (h.tag & BaseBenchmark::tag) {
            BaseBenchmark *b = static_cast<BaseBenchmark *>(h.p);
            if (b->m_p_owner) b->m_p_owner->destroyBenchmark(b);
            else delete b;
        } else if (h.tag & BenchmarkDescription::tag) {
            // owned by the engine
        } else if (h.tag & EngineObject::tag) {
            delete static_cast<EngineObject *>(h.p);
        } else {
            throw HEBenchError("Unknown handle tag.", HEBENCH_ECODE_INVALID_ARGS);
        }
    });
}

ErrorCode subscribeBenchmarksCount(Handle h_engine, uint64_t *p_count)
{
    BaseEngine *e = nullptr;
    return guard(e, [&] {
        e = engine_of(h_engine);
        if (!p_count) throw HEBenchError("Invalid null pointer.", HEBENCH_ECODE_INVALID_ARGS);
        *p_count = e->descriptions().size();
    });
}

ErrorCode subscribeBenchmarks(Handle h_engine, Handle *p_h_bench_descs, uint64_t count)
{
    BaseEngine *e = nullptr;
    return guard(e, [&] {
        e = engine_of(h_engine);
        if (!p_h_bench_descs) throw HEBenchError("Invalid null pointer.", HEBENCH_ECODE_INVALID_ARGS);
        const auto &d = e->descriptions();
        for (uint64_t i = 0; i < count && i < d.size(); ++i) {
            p_h_bench_descs[i].p = d[i].get();
            p_h_bench_descs[i].size = sizeof(void *);
            p_h_bench_descs[i].tag = BenchmarkDescription::tag;
        }
    });
}

ErrorCode getWorkloadParamsDetails(Handle h_engine, Handle h_bench_desc, uint64_t *p_param_count, uint64_t *p_default_count)
{
    BaseEngine *e = nullptr;
    return guard(e, [&] {
        e = engine_of(h_engine);
        BenchmarkDescription *d = desc_of(h_bench_desc);
        if (p_param_count) *p_param_count = d->getWorkloadParameterCount();
        if (p_default_count) *p_default_count = d->getWorkloadDefaultParameters().size();
    });
}

ErrorCode describeBenchmark(Handle h_engine, Handle h_bench_desc, BenchmarkDescriptor *p_bench_desc, WorkloadParams *p_default_params,
                            uint64_t default_count)
{
    BaseEngine *e = nullptr;
    return guard(e, [&] {
        e = engine_of(h_engine);
        BenchmarkDescription *d = desc_of(h_bench_desc);
        if (p_bench_desc) *p_bench_desc = d->getBenchmarkDescriptor();
        if (p_default_params) {
            const auto &defs = d->getWorkloadDefaultParameters();
            for (uint64_t i = 0; i < default_count && i < defs.size(); ++i) {
                const uint64_t n = std::min<uint64_t>(p_default_params[i].count, defs[i].size());
                if (p_default_params[i].params)
                    for (uint64_t k = 0; k < n; ++k) p_default_params[i].params[k] = defs[i][k];
            }
        }
    });
}

ErrorCode createBenchmark(Handle h_engine, Handle h_bench_desc, const WorkloadParams *p_params, Handle *h_benchmark)
{
    BaseEngine *e = nullptr;
    return guard(e, [&] {
        e = engine_of(h_engine);
        BenchmarkDescription *d = desc_of(h_bench_desc);
        if (!h_benchmark) throw HEBenchError("Invalid null handle.", HEBENCH_ECODE_INVALID_ARGS);
        if (d->getWorkloadParameterCount() > 0 && (!p_params || p_params->count < d->getWorkloadParameterCount()))
            throw HEBenchError("Invalid workload parameters: not enough parameters for this workload.", HEBENCH_ECODE_INVALID_ARGS);
        BaseBenchmark *b = d->createBenchmark(*e, p_params);
        b->m_p_owner = d;
        h_benchmark->p = b;
        h_benchmark->size = sizeof(void *);
        h_benchmark->tag = b->classTag();
    });
}

ErrorCode initBenchmark(Handle h_benchmark, const BenchmarkDescriptor *p_concrete_desc)
{
    BaseEngine *e = nullptr;
    return guard(e, [&] {
        BaseBenchmark *b = bench_of(h_benchmark);
        e = &b->getEngine();
        if (!p_concrete_desc) throw HEBenchError("Invalid null descriptor.", HEBENCH_ECODE_INVALID_ARGS);
        b->initialize(*p_concrete_desc);
    });
}

#define BENCH_CALL(body)                  \
    BaseEngine *e = nullptr;              \
    return guard(e, [&] {                 \
        BaseBenchmark *b = bench_of(h_benchmark); \
        e = &b->getEngine();              \
        body;                             \
    })

ErrorCode encode(Handle h_benchmark, const DataPackCollection *p_parameters, Handle *h_plaintext)
{
    BENCH_CALL(if (!p_parameters || !h_plaintext) throw HEBenchError("Invalid null argument.", HEBENCH_ECODE_INVALID_ARGS);
               *h_plaintext = b->encode(p_parameters));
}
ErrorCode decode(Handle h_benchmark, Handle h_plaintext, DataPackCollection *p_native)
{
    BENCH_CALL(if (!p_native) throw HEBenchError("Invalid null argument.", HEBENCH_ECODE_INVALID_ARGS); b->decode(h_plaintext, p_native));
}
ErrorCode encrypt(Handle h_benchmark, Handle h_plaintext, Handle *h_ciphertext)
{
    BENCH_CALL(if (!h_ciphertext) throw HEBenchError("Invalid null argument.", HEBENCH_ECODE_INVALID_ARGS); *h_ciphertext = b->encrypt(h_plaintext));
}
ErrorCode decrypt(Handle h_benchmark, Handle h_ciphertext, Handle *h_plaintext)
{
    BENCH_CALL(if (!h_plaintext) throw HEBenchError("Invalid null argument.", HEBENCH_ECODE_INVALID_ARGS); *h_plaintext = b->decrypt(h_ciphertext));
}
ErrorCode load(Handle h_benchmark, const Handle *h_local_packed_params, uint64_t local_count, Handle *h_remote)
{
    BENCH_CALL(if (!h_remote) throw HEBenchError("Invalid null argument.", HEBENCH_ECODE_INVALID_ARGS);
               *h_remote = b->load(h_local_packed_params, local_count));
}
ErrorCode store(Handle h_benchmark, Handle h_remote, Handle *h_local_packed_params, uint64_t local_count)
{
    BENCH_CALL(b->store(h_remote, h_local_packed_params, local_count));
}
ErrorCode operate(Handle h_benchmark, Handle h_remote_packed_params, const ParameterIndexer *p_param_indexers, uint64_t indexers_count,
                  Handle *h_remote_output)
{
    BENCH_CALL(if (!h_remote_output || !p_param_indexers) throw HEBenchError("Invalid null argument.", HEBENCH_ECODE_INVALID_ARGS);
               *h_remote_output = b->operate(h_remote_packed_params, p_param_indexers, indexers_count));
}

uint64_t getSchemeName(Handle h_engine, Scheme s, char *p_name, uint64_t size)
{
    try {
        return copy_string(engine_of(h_engine)->schemeName(s), p_name, size);
    } catch (...) {
        return 0;
    }
}
uint64_t getSchemeSecurityName(Handle h_engine, Scheme s, Security sec, char *p_name, uint64_t size)
{
    try {
        return copy_string(engine_of(h_engine)->securityName(s, sec), p_name, size);
    } catch (...) {
        return 0;
    }
}
uint64_t getBenchmarkDescriptionEx(Handle h_engine, Handle h_bench_desc, const WorkloadParams *p_w_params, char *p_description, uint64_t size)
{
    try {
        engine_of(h_engine);
        return copy_string(desc_of(h_bench_desc)->getBenchmarkDescription(p_w_params), p_description, size);
    } catch (const std::exception &err) {
        record(nullptr, err.what(), HEBENCH_ECODE_CRITICAL_ERROR);
        return 0;
    }
}
uint64_t getErrorDescription(Handle h_engine, ErrorCode code, char *p_description, uint64_t size)
{
    try {
        return copy_string(engine_of(h_engine)->errorDescription(code), p_description, size);
    } catch (...) {
        return 0;
    }
}
uint64_t getLastErrorDescription(Handle h_engine, char *p_description, uint64_t size)
{
    std::lock_guard<std::mutex> lk(g_mtx);
    if ((h_engine.tag & BaseEngine::tag) && h_engine.p) {
        const std::string &own = static_cast<BaseEngine *>(h_engine.p)->lastError();
        return copy_string(own.empty() ? g_last_error_no_engine : own, p_description, size);
    }
    return copy_string(g_last_error_no_engine, p_description, size);
}

} // extern "C"
} // namespace APIBridge
} // namespace hebench
