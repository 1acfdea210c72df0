// benchmarks.h — workload classes of the MI355X backend, mirroring the reference's
//   sbe::ckks::ElementWiseBenchmark   (src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp, .h)
//   sbe::bfv::ElementWiseBenchmark    (src/benchmarks/bfv/seal_bfv_element_wise_benchmark.cpp, .h)   [EltwiseAdd]
//   sbe::ckks::DotProductBenchmark    (src/benchmarks/ckks/seal_ckks_dot_product_benchmark.cpp, .h)
// Same workload-parameter order and defaults, same validation messages and error codes, same
// handle semantics and result ordering; operate() launches HIP kernels on HBM-resident slabs instead of looping
// over seal::Evaluator calls with OpenMP.
#pragma once
#include "he_context.h"
#include "multi_device.h"

namespace mi355x {

enum class Scheme { CKKS, BFV };

class VectorBenchmarkDescription : public hebench::cpp::BenchmarkDescription {
public:
    HEBERROR_DECLARE_CLASS_NAME(VectorBenchmarkDescription)
    static constexpr const char *AlgorithmName = "Vector";
    static constexpr const char *AlgorithmDescription = "One vector per ciphertext";
    static constexpr std::size_t NumOpParams = 2;
    enum : std::uint64_t { // ckks eltwise .h:31-42
        Index_WParamsStart = 0,
        Index_n = Index_WParamsStart,
        Index_ExtraWParamsStart,
        Index_PolyModulusDegree = Index_ExtraWParamsStart,
        Index_NumCoefficientModuli,
        Index_CoefficientModulusBits,
        Index_ScaleExponentBits, // BFV: PlainModulusBits
        Index_NumThreads,
        NumWorkloadParams,
        // optional trailing parameter, NOT part of the declared defaults: GPUs one operate() call is spread over;
        // absent or 0 = HE355_NUM_DEVICES, else 1.  The declared parameter sets stay the reference's six.
        Index_NumDevices = NumWorkloadParams
    };
    VectorBenchmarkDescription(Scheme scheme, hebench::APIBridge::Category category, hebench::APIBridge::Workload op);
    hebench::cpp::BaseBenchmark *createBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::WorkloadParams *p_params) override;
    void destroyBenchmark(hebench::cpp::BaseBenchmark *p_bench) override;
    std::string getBenchmarkDescription(const hebench::APIBridge::WorkloadParams *p_w_params) const override;
    Scheme scheme() const { return m_scheme; }

private:
    Scheme m_scheme;
};

class VectorBenchmark : public hebench::cpp::BaseBenchmark {
public:
    HEBERROR_DECLARE_CLASS_NAME(VectorBenchmark)
    static constexpr std::int64_t tag = 0x1;
    VectorBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::BenchmarkDescriptor &bench_desc,
                    const hebench::APIBridge::WorkloadParams &bench_params, Scheme scheme);

    hebench::APIBridge::Handle encode(const hebench::APIBridge::DataPackCollection *p_parameters) override;
    void decode(hebench::APIBridge::Handle encoded_data, hebench::APIBridge::DataPackCollection *p_native) override;
    hebench::APIBridge::Handle encrypt(hebench::APIBridge::Handle encoded_data) override;
    hebench::APIBridge::Handle decrypt(hebench::APIBridge::Handle encrypted_data) override;
    hebench::APIBridge::Handle load(const hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    void store(hebench::APIBridge::Handle remote_data, hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    hebench::APIBridge::Handle operate(hebench::APIBridge::Handle h_remote_packed, const hebench::APIBridge::ParameterIndexer *p_param_indexers,
                                       std::uint64_t indexers_count) override;
    std::int64_t classTag() const override { return BaseBenchmark::classTag() | VectorBenchmark::tag; }

    // what load() hands to operate(): the operand slabs on the primary device and, with NumDevices > 1, their replicas
    struct RemotePack {
        std::vector<std::shared_ptr<DeviceCiphers>> ops;                    // [operand] on the primary device
        // [device][operand]; [0] = ops.  Device d > 0 holds ITS BLOCK of operand 0 (rows rowsOf(ops[0]->n, devices, d), as
        // reference-seal-backend_amd/sharding.py cuts them) and all of operand 1
        std::vector<std::vector<std::shared_ptr<DeviceCiphers>>> replicas;
    };

private:
    std::shared_ptr<DeviceCiphers> operateOn(he355_ctx *ctx, DeviceGroup *group, int device, const DeviceCiphers &p0, const DeviceCiphers &p1,
                                             he355_indexer ix, std::uint64_t n, std::shared_ptr<DeviceCiphers> into, std::uint64_t into_offset);
    Scheme m_scheme;
    HeContextWrapper::Ptr m_p_ctx_wrapper;
    hebench::cpp::WorkloadParams::VectorSize m_w_params;
    int m_num_devices = 1;
    std::shared_ptr<DeviceGroup> m_group;
};

} // namespace mi355x
