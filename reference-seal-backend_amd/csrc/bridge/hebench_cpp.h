// hebench_cpp.h — minimal clean-room counterpart of HEBench's C++ wrapper library `hebench_cpp`
// (hebench/api_bridge/cpp/hebench.hpp, NOT present in this image).  It provides exactly the base classes and
// helpers the reference's sources use, with the same names and meaning, so that the backend classes in
// benchmarks.cpp / engine.cpp read like the reference's:
//   HEBenchError(msg, code), HEBERROR_MSG_CLASS        ckks eltwise .cpp:28-29
//   BaseEngine::{addErrorCode, addSchemeName, addSecurityName, addBenchmarkDescription}  seal_engine.cpp:97-151
//   BaseEngine::{createHandle<T>, retrieveFromHandle<T>, duplicateHandle}               ckks eltwise .cpp:203-212,288
//   BenchmarkDescription::{createBenchmark, destroyBenchmark, getBenchmarkDescription, addDefaultParameters}
//   BaseBenchmark::{encode, decode, encrypt, decrypt, load, store, operate, classTag}  ckks eltwise .h:67-80
//   WorkloadParams::VectorSize (n(), add<T>, get<T>)                                    ckks eltwise .cpp:58-65,133-137
// The extern "C" API-Bridge functions (include/hebench_api_bridge.h) are implemented on top in hebench_cpp.cpp.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#if defined(HEBENCH_REAL_API_HEADERS) // make HEBENCH_API_BRIDGE_DIR=...: the real api-bridge headers (include/hebench_api_bridge.h explains)
#include "hebench/api_bridge/api.h"
#else
#include "../../../include/hebench_api_bridge.h"
#endif

namespace hebench {
namespace cpp {

class HEBenchError : public std::runtime_error {
public:
    HEBenchError(const std::string &msg, APIBridge::ErrorCode code) : std::runtime_error(msg), m_code(code) {}
    APIBridge::ErrorCode getErrorCode() const { return m_code; }

private:
    APIBridge::ErrorCode m_code;
};
#define HEBERROR_DECLARE_CLASS_NAME(name) static constexpr const char *m_private_class_name = #name;
#define HEBERROR_MSG_CLASS(message) (std::string(m_private_class_name) + "::" + __func__ + "(): " + std::string(message))
#define HEBERROR_MSG(message) (std::string(__func__) + "(): " + std::string(message))

// tags carried in Handle.tag
struct ITaggedObject {
    static constexpr std::int64_t MaskReservedBits = 0x7F00000000000000;
    virtual ~ITaggedObject() = default;
    virtual std::int64_t classTag() const = 0;
};
struct EngineObject { // data objects created by createHandle<T>
    static constexpr std::int64_t tag = 0x0800000000000000;
    std::shared_ptr<void> obj;
};

class BaseEngine;
class BaseBenchmark;

namespace WorkloadParams {
class Common {
public:
    Common() = default;
    explicit Common(const APIBridge::WorkloadParams &p) : m_params(p.params, p.params + p.count) {}
    template <class T> void add(const T &value, const std::string &name);
    template <class T> T get(std::size_t i) const;
    const std::vector<APIBridge::WorkloadParam> &getParams() const { return m_params; }

protected:
    std::vector<APIBridge::WorkloadParam> m_params;
};
template <> inline void Common::add<std::uint64_t>(const std::uint64_t &value, const std::string &name)
{
    APIBridge::WorkloadParam p;
    std::memset(&p, 0, sizeof(p));
    p.data_type = APIBridge::WorkloadParamType::UInt64;
    std::strncpy(p.name, name.c_str(), HEBENCH_MAX_BUFFER_SIZE - 1);
    p.u_param = value;
    m_params.push_back(p);
}
template <> inline std::uint64_t Common::get<std::uint64_t>(std::size_t i) const { return m_params.at(i).u_param; }

// workloads whose first parameter is the vector size n (EltwiseAdd, EltwiseMultiply, DotProduct)
class VectorSize : public Common {
public:
    VectorSize() { add<std::uint64_t>(0, "n"); }
    explicit VectorSize(const APIBridge::WorkloadParams &p) : Common(p)
    {
        if (m_params.empty()) throw HEBenchError("Workload requires, at least, 1 parameter: `n`.", HEBENCH_ECODE_INVALID_ARGS);
    }
    std::uint64_t &n() { return m_params[0].u_param; }
    std::uint64_t n() const { return m_params[0].u_param; }
};
} // namespace WorkloadParams

class BenchmarkDescription : public ITaggedObject {
public:
    static constexpr std::int64_t tag = 0x2000000000000000;
    std::int64_t classTag() const override { return tag; }
    const APIBridge::BenchmarkDescriptor &getBenchmarkDescriptor() const { return m_descriptor; }
    const std::vector<std::vector<APIBridge::WorkloadParam>> &getWorkloadDefaultParameters() const { return m_default_params; }
    std::size_t getWorkloadParameterCount() const { return m_default_params.empty() ? 0 : m_default_params.front().size(); }

    virtual BaseBenchmark *createBenchmark(BaseEngine &engine, const APIBridge::WorkloadParams *p_params) = 0;
    virtual void destroyBenchmark(BaseBenchmark *p_bench) = 0;
    // base text: one CSV line per workload parameter, as hebench_cpp's default does
    virtual std::string getBenchmarkDescription(const APIBridge::WorkloadParams *p_w_params) const;

protected:
    void addDefaultParameters(const WorkloadParams::Common &p) { m_default_params.push_back(p.getParams()); }
    APIBridge::BenchmarkDescriptor m_descriptor;

private:
    std::vector<std::vector<APIBridge::WorkloadParam>> m_default_params;
};

class BaseBenchmark : public ITaggedObject {
public:
    static constexpr std::int64_t tag = 0x1000000000000000;
    BaseBenchmark(BaseEngine &engine, const APIBridge::BenchmarkDescriptor &desc, const APIBridge::WorkloadParams &params)
        : m_engine(engine), m_descriptor(desc), m_params(params.params, params.params + params.count)
    {
    }
    std::int64_t classTag() const override { return tag; }
    BaseEngine &getEngine() { return m_engine; }
    const APIBridge::BenchmarkDescriptor &getDescriptor() const { return m_descriptor; }
    virtual void initialize(const APIBridge::BenchmarkDescriptor &concrete) { m_descriptor = concrete; }

    virtual APIBridge::Handle encode(const APIBridge::DataPackCollection *p_parameters) = 0;
    virtual void decode(APIBridge::Handle encoded_data, APIBridge::DataPackCollection *p_native) = 0;
    virtual APIBridge::Handle encrypt(APIBridge::Handle encoded_data) = 0;
    virtual APIBridge::Handle decrypt(APIBridge::Handle encrypted_data) = 0;
    virtual APIBridge::Handle load(const APIBridge::Handle *p_local_data, std::uint64_t count) = 0;
    virtual void store(APIBridge::Handle remote_data, APIBridge::Handle *p_local_data, std::uint64_t count) = 0;
    virtual APIBridge::Handle operate(APIBridge::Handle h_remote_packed, const APIBridge::ParameterIndexer *p_param_indexers,
                                      std::uint64_t indexers_count) = 0;
    // owner description (needed by destroyHandle)
    BenchmarkDescription *m_p_owner = nullptr;

private:
    BaseEngine &m_engine;
    APIBridge::BenchmarkDescriptor m_descriptor;
    std::vector<APIBridge::WorkloadParam> m_params;
};

class BaseEngine : public ITaggedObject {
public:
    static constexpr std::int64_t tag = 0x4000000000000000;
    std::int64_t classTag() const override { return tag; }
    virtual void init() = 0;

    template <class T> APIBridge::Handle createHandle(std::uint64_t size, std::int64_t extra_tags, T &&obj)
    {
        auto *eo = new EngineObject();
        eo->obj = std::make_shared<typename std::decay<T>::type>(std::forward<T>(obj));
        APIBridge::Handle h;
        h.p = eo;
        h.size = size;
        h.tag = EngineObject::tag | (extra_tags & ~ITaggedObject::MaskReservedBits);
        return h;
    }
    template <class T> T &retrieveFromHandle(APIBridge::Handle h, std::int64_t extra_tags = 0) const
    {
        if ((h.tag & EngineObject::tag) == 0 || !h.p)
            throw HEBenchError("retrieveFromHandle(): invalid tag detected. Expected EngineObject::tag.", HEBENCH_ECODE_CRITICAL_ERROR);
        if ((h.tag & extra_tags) != extra_tags)
            throw HEBenchError("retrieveFromHandle(): handle does not carry the expected tags.", HEBENCH_ECODE_CRITICAL_ERROR);
        return *static_cast<T *>(static_cast<EngineObject *>(h.p)->obj.get());
    }
    APIBridge::Handle duplicateHandle(APIBridge::Handle h, std::int64_t check_tags = 0) const
    {
        if ((h.tag & EngineObject::tag) == 0 || !h.p)
            throw HEBenchError("duplicateHandle(): invalid tag detected. Expected EngineObject::tag.", HEBENCH_ECODE_CRITICAL_ERROR);
        if ((h.tag & check_tags) != check_tags)
            throw HEBenchError("duplicateHandle(): handle does not carry the expected tags.", HEBENCH_ECODE_CRITICAL_ERROR);
        auto *eo = new EngineObject(*static_cast<EngineObject *>(h.p)); // shares the object (shared_ptr copy)
        APIBridge::Handle r = h;
        r.p = eo;
        return r;
    }

    // registry (read by the extern "C" layer)
    const std::vector<std::shared_ptr<BenchmarkDescription>> &descriptions() const { return m_descriptions; }
    std::string schemeName(APIBridge::Scheme s) const;
    std::string securityName(APIBridge::Scheme s, APIBridge::Security sec) const;
    std::string errorDescription(APIBridge::ErrorCode code) const;
    void setLastError(const std::string &msg, APIBridge::ErrorCode code) { m_last_error = msg; m_last_code = code; }
    const std::string &lastError() const { return m_last_error; }

protected:
    void addErrorCode(APIBridge::ErrorCode code, const std::string &description) { m_errors[code] = description; }
    void addSchemeName(APIBridge::Scheme s, const std::string &name) { m_schemes[s] = name; }
    void addSecurityName(APIBridge::Security sec, const std::string &name) { m_security[sec] = name; }
    void addBenchmarkDescription(std::shared_ptr<BenchmarkDescription> d) { m_descriptions.push_back(std::move(d)); }

private:
    std::vector<std::shared_ptr<BenchmarkDescription>> m_descriptions;
    std::map<APIBridge::ErrorCode, std::string> m_errors;
    std::map<APIBridge::Scheme, std::string> m_schemes;
    std::map<APIBridge::Security, std::string> m_security;
    std::string m_last_error;
    APIBridge::ErrorCode m_last_code = HEBENCH_ECODE_SUCCESS;
};

// implemented by the backend (engine.cpp), like the reference's seal_engine.cpp:36-63
BaseEngine *createEngine(const std::int8_t *p_buffer, std::uint64_t size);
void destroyEngine(BaseEngine *p);

} // namespace cpp
} // namespace hebench
