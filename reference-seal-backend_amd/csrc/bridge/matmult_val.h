// matmult_val.h — MatrixMultiply "val" (other = 0), Latency, CKKS and BFV: mirrors sbe::ckks::MatMultValBenchmark and
// sbe::bfv::MatMultValBenchmark (/root/reference/src/benchmarks/{ckks,bfv}/seal_*_matmultval_benchmark.cpp).
// One row of M0 per ciphertext, M1 transposed at encode (one column per ciphertext); result (i,j) =
//   CKKS: accumulateCKKS(rescale(relinearize(M0[i] * M1T[j])), cols_M0)   (ckks .cpp:253-256)
//   BFV : accumulateBFV(relinearize(M0[i] * M1T[j]), cols_M0)              (bfv  .cpp:253-255)
// i.e. the reference's OpenMP collapse(2) loop over (i,j) becomes ONE outer-product batch on the GPU.
#pragma once
#include "benchmarks.h"

namespace mi355x {

class MatMultValBenchmarkDescription : public hebench::cpp::BenchmarkDescription {
public:
    HEBERROR_DECLARE_CLASS_NAME(MatMultValBenchmarkDescription)
    static constexpr std::int64_t MatMultValOtherID = 0;
    static constexpr const char *AlgorithmName = "MatMultVal";
    static constexpr const char *AlgorithmDescription = "One matrix row per ciphertext, Encode transposes second matrix";
    enum : std::uint64_t { Index_rows_M0 = 0, Index_cols_M0, Index_cols_M1, Index_PolyModulusDegree, Index_NumCoefficientModuli,
                           Index_CoefficientModulusBits, Index_ScaleExponentBits /* BFV: PlainModulusBits */, Index_NumThreads, NumWorkloadParams };
    explicit MatMultValBenchmarkDescription(Scheme scheme);
    hebench::cpp::BaseBenchmark *createBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::WorkloadParams *p_params) override;
    void destroyBenchmark(hebench::cpp::BaseBenchmark *p_bench) override;
    std::string getBenchmarkDescription(const hebench::APIBridge::WorkloadParams *p_w_params) const override;

private:
    Scheme m_scheme;
};

class MatMultValBenchmark : public hebench::cpp::BaseBenchmark {
public:
    HEBERROR_DECLARE_CLASS_NAME(MatMultValBenchmark)
    static constexpr std::int64_t tag = 0x20;
    MatMultValBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::BenchmarkDescriptor &bench_desc,
                        const hebench::APIBridge::WorkloadParams &bench_params, Scheme scheme);
    hebench::APIBridge::Handle encode(const hebench::APIBridge::DataPackCollection *p_parameters) override;
    void decode(hebench::APIBridge::Handle encoded_data, hebench::APIBridge::DataPackCollection *p_native) override;
    hebench::APIBridge::Handle encrypt(hebench::APIBridge::Handle encoded_data) override;
    hebench::APIBridge::Handle decrypt(hebench::APIBridge::Handle encrypted_data) override;
    hebench::APIBridge::Handle load(const hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    void store(hebench::APIBridge::Handle remote_data, hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    hebench::APIBridge::Handle operate(hebench::APIBridge::Handle h_remote_packed, const hebench::APIBridge::ParameterIndexer *p_param_indexers,
                                       std::uint64_t indexers_count) override;
    std::int64_t classTag() const override { return BaseBenchmark::classTag() | MatMultValBenchmark::tag; }

private:
    static const hebench::APIBridge::DataPack &findDataPack(const hebench::APIBridge::DataPackCollection &c, std::uint64_t pos);
    std::uint64_t rows_M0() const { return m_w[0]; }
    std::uint64_t cols_M0() const { return m_w[1]; }
    std::uint64_t cols_M1() const { return m_w[2]; }
    Scheme m_scheme;
    std::vector<std::uint64_t> m_w;
    HeContextWrapper::Ptr m_p_ctx_wrapper;
};

} // namespace mi355x
