// engine.cpp — engine factory and benchmark registration, mirroring /root/reference/src/engine/seal_engine.cpp.
#include <sstream>

#include "benchmarks.h"
#include "matmult_row.h"
#include "logreg.h"
#include "matmult_cba.h"
#include "matmult_val.h"

#define HEBENCH_API_VERSION_NEEDED_MAJOR 0
#define HEBENCH_API_VERSION_NEEDED_MINOR 8
#define HEBENCH_API_VERSION_NEEDED_REVISION 0

namespace mi355x {

class Mi355xEngine : public hebench::cpp::BaseEngine {
public:
    HEBERROR_DECLARE_CLASS_NAME(Mi355xEngine)
    static Mi355xEngine *create()
    {
        Mi355xEngine *p = new Mi355xEngine();
        p->init();
        return p;
    }
    void init() override
    {
        namespace AB = hebench::APIBridge;
        // add any new error codes (seal_engine.cpp:97)
        addErrorCode(HEB355_ECODE_HE_ERROR, "HE parameter / context error");
        addErrorCode(HEB355_ECODE_DEVICE_ERROR, "HIP device error (this backend has no CPU fallback)");
        // add supported schemes (seal_engine.cpp:101-102)
        addSchemeName(HEBENCH_HE_SCHEME_CKKS, "CKKS");
        addSchemeName(HEBENCH_HE_SCHEME_BFV, "BFV");
        // add supported security (seal_engine.cpp:105)
        addSecurityName(HEBENCH_HE_SECURITY_128, "128 bits");
        // benchmark descriptors, same order as seal_engine.cpp:108-151 for the workloads implemented so far
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::BFV, AB::Category::Latency, AB::Workload::EltwiseAdd));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::CKKS, AB::Category::Latency, AB::Workload::EltwiseAdd));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::BFV, AB::Category::Offline, AB::Workload::EltwiseAdd));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::CKKS, AB::Category::Offline, AB::Workload::EltwiseAdd));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::BFV, AB::Category::Latency, AB::Workload::EltwiseMultiply));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::CKKS, AB::Category::Latency, AB::Workload::EltwiseMultiply));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::BFV, AB::Category::Offline, AB::Workload::EltwiseMultiply));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::CKKS, AB::Category::Offline, AB::Workload::EltwiseMultiply));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::BFV, AB::Category::Latency, AB::Workload::DotProduct));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::CKKS, AB::Category::Latency, AB::Workload::DotProduct));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::BFV, AB::Category::Offline, AB::Workload::DotProduct));
        addBenchmarkDescription(std::make_shared<VectorBenchmarkDescription>(Scheme::CKKS, AB::Category::Offline, AB::Workload::DotProduct));
        addBenchmarkDescription(std::make_shared<MatMultCipherBatchAxisBenchmarkDescription>(Scheme::BFV));
        addBenchmarkDescription(std::make_shared<MatMultCipherBatchAxisBenchmarkDescription>(Scheme::CKKS));
        addBenchmarkDescription(std::make_shared<MatMultValBenchmarkDescription>(Scheme::BFV));
        addBenchmarkDescription(std::make_shared<MatMultValBenchmarkDescription>(Scheme::CKKS));
        addBenchmarkDescription(std::make_shared<MatMultRowBenchmarkDescription>(Scheme::BFV));
        addBenchmarkDescription(std::make_shared<MatMultRowBenchmarkDescription>(Scheme::CKKS));
        addBenchmarkDescription(std::make_shared<LogRegHornerBenchmarkDescription>(hebench::APIBridge::Category::Latency));
        addBenchmarkDescription(std::make_shared<LogRegHornerBenchmarkDescription>(hebench::APIBridge::Category::Offline, 0));
    }
};

} // namespace mi355x

namespace hebench {
namespace cpp {

BaseEngine *createEngine(const std::int8_t *p_buffer, std::uint64_t size)
{
    // backend doesn't need extra init data
    (void)p_buffer;
    (void)size;
    if (HEBENCH_API_VERSION_MAJOR != HEBENCH_API_VERSION_NEEDED_MAJOR || HEBENCH_API_VERSION_MINOR != HEBENCH_API_VERSION_NEEDED_MINOR
        || HEBENCH_API_VERSION_REVISION < HEBENCH_API_VERSION_NEEDED_REVISION) {
        std::stringstream ss;
        ss << "Critical: Invalid HEBench API version detected. Required: " << HEBENCH_API_VERSION_NEEDED_MAJOR << "." << HEBENCH_API_VERSION_NEEDED_MINOR
           << "." << HEBENCH_API_VERSION_NEEDED_REVISION << ", but " << HEBENCH_API_VERSION_MAJOR << "." << HEBENCH_API_VERSION_MINOR << "."
           << HEBENCH_API_VERSION_REVISION << " received.";
        throw HEBenchError(HEBERROR_MSG(ss.str()), HEBENCH_ECODE_CRITICAL_ERROR);
    }
    return mi355x::Mi355xEngine::create();
}

void destroyEngine(BaseEngine *p)
{
    delete p;
}

} // namespace cpp
} // namespace hebench
