// matmult_cba.h — MatrixMultiply "cipher batch axis" (other = 1), Latency, CKKS and BFV: mirrors
// sbe::{ckks,bfv}::MatMultCipherBatchAxisBenchmark (/root/reference/src/benchmarks/{ckks,bfv}/seal_*_matmult_cipherbatchaxis_benchmark.cpp).
// Every matrix ELEMENT is one ciphertext; result (i,j) =
//   CKKS: rescale(relinearize(sum_k M0(i,k) * M1(k,j)))      (size-3 sum first, ckks .cpp:404-420, then :436-437)
//   BFV : sum_k relinearize(M0(i,k) * M1(k,j))               (bfv .cpp:395-409)
// The reference's collapse(2) OpenMP loop over (i,j) is one batch of rows_M0*cols_M1 results on the GPU.
#pragma once
#include "benchmarks.h"

namespace mi355x {

class MatMultCipherBatchAxisBenchmarkDescription : public hebench::cpp::BenchmarkDescription {
public:
    HEBERROR_DECLARE_CLASS_NAME(MatMultCipherBatchAxisBenchmarkDescription)
    static constexpr std::int64_t MatMultOtherID = 0x01;
    static constexpr const char *AlgorithmName = "CipherBatchAxis"; // the reference's strings (they name report rows)
    static constexpr const char *AlgorithmDescription = "One matrix element per ciphertext";
    enum : std::uint64_t { Index_rows_M0 = 0, Index_cols_M0, Index_cols_M1, Index_PolyModulusDegree, Index_NumCoefficientModuli,
                           Index_CoefficientModulusBits, Index_ScaleExponentBits /* BFV: PlainModulusBits */, Index_NumThreads, NumWorkloadParams };
    explicit MatMultCipherBatchAxisBenchmarkDescription(Scheme scheme);
    hebench::cpp::BaseBenchmark *createBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::WorkloadParams *p_params) override;
    void destroyBenchmark(hebench::cpp::BaseBenchmark *p_bench) override;
    std::string getBenchmarkDescription(const hebench::APIBridge::WorkloadParams *p_w_params) const override;

private:
    Scheme m_scheme;
};

class MatMultCipherBatchAxisBenchmark : public hebench::cpp::BaseBenchmark {
public:
    HEBERROR_DECLARE_CLASS_NAME(MatMultCipherBatchAxisBenchmark)
    static constexpr std::int64_t tag = 0x40;
    MatMultCipherBatchAxisBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::BenchmarkDescriptor &bench_desc,
                                    const hebench::APIBridge::WorkloadParams &bench_params, Scheme scheme);
    hebench::APIBridge::Handle encode(const hebench::APIBridge::DataPackCollection *p_parameters) override;
    void decode(hebench::APIBridge::Handle encoded_data, hebench::APIBridge::DataPackCollection *p_native) override;
    hebench::APIBridge::Handle encrypt(hebench::APIBridge::Handle encoded_data) override;
    hebench::APIBridge::Handle decrypt(hebench::APIBridge::Handle encrypted_data) override;
    hebench::APIBridge::Handle load(const hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    void store(hebench::APIBridge::Handle remote_data, hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    hebench::APIBridge::Handle operate(hebench::APIBridge::Handle h_remote_packed, const hebench::APIBridge::ParameterIndexer *p_param_indexers,
                                       std::uint64_t indexers_count) override;
    std::int64_t classTag() const override { return BaseBenchmark::classTag() | MatMultCipherBatchAxisBenchmark::tag; }

private:
    std::uint64_t rows_M0() const { return m_w[0]; }
    std::uint64_t cols_M0() const { return m_w[1]; }
    std::uint64_t cols_M1() const { return m_w[2]; }
    Scheme m_scheme;
    std::vector<std::uint64_t> m_w;
    HeContextWrapper::Ptr m_p_ctx_wrapper;
};

} // namespace mi355x
