// matmult_row.h — MatrixMultiply, "row" packing (other = 2), Latency: mirrors sbe::bfv::MatMultRowLatencyBenchmark
// (/root/reference/src/benchmarks/bfv/seal_bfv_matmult_row_benchmark.cpp) and sbe::ckks::MatMultRowLatencyBenchmark
// (/root/reference/src/benchmarks/ckks/seal_ckks_matmult_row_benchmark.cpp).
// BFV : two rows of A per ciphertext (one per batching row), B in one ciphertext; per row-pair:
//       multiply -> relinearize -> sum over j of rotate_rows(base, j * spacers), spacers = (N/2)/dim2   (bfv .cpp:486-539)
// CKKS: one row of A per ciphertext; multiply -> relinearize -> sum over j of rotate_vector(base, j * spacers),
//       spacers = slots/dim2, no rescale                                                              (ckks .cpp:472-523)
#pragma once
#include "benchmarks.h"

namespace mi355x {

class MatMultRowBenchmarkDescription : public hebench::cpp::BenchmarkDescription {
public:
    HEBERROR_DECLARE_CLASS_NAME(MatMultRowBenchmarkDescription)
    static constexpr std::int64_t MatMultRowOtherID = 2;
    static constexpr const char *AlgorithmName = "MatMulRow"; // sic, the reference's spelling
    static constexpr const char *AlgorithmDescription = ""; // empty in the reference (bfv row .h); BFV: two rows of M0 per ciphertext, M1 in one
    static constexpr std::size_t NumOpParams = 2;
    enum : std::uint64_t { // bfv row .h:34-50
        Index_rows_M0 = 0,
        Index_cols_M0,
        Index_cols_M1,
        Index_PolyModulusDegree,
        Index_NumCoefficientModuli,
        Index_CoefficientModulusBits,
        Index_PlainModulusBits,
        Index_NumThreads,
        NumWorkloadParams
    };
    explicit MatMultRowBenchmarkDescription(Scheme scheme = Scheme::BFV);
    hebench::cpp::BaseBenchmark *createBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::WorkloadParams *p_params) override;
    void destroyBenchmark(hebench::cpp::BaseBenchmark *p_bench) override;
    std::string getBenchmarkDescription(const hebench::APIBridge::WorkloadParams *p_w_params) const override;

private:
    Scheme m_scheme;
};

class MatMultRowLatencyBenchmark : public hebench::cpp::BaseBenchmark {
public:
    HEBERROR_DECLARE_CLASS_NAME(MatMultRowLatencyBenchmark)
    static constexpr std::int64_t tag = 0x20 + MatMultRowBenchmarkDescription::MatMultRowOtherID;
    MatMultRowLatencyBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::BenchmarkDescriptor &bench_desc,
                               const hebench::APIBridge::WorkloadParams &bench_params, Scheme scheme = Scheme::BFV);
    hebench::APIBridge::Handle encode(const hebench::APIBridge::DataPackCollection *p_parameters) override;
    void decode(hebench::APIBridge::Handle encoded_data, hebench::APIBridge::DataPackCollection *p_native) override;
    hebench::APIBridge::Handle encrypt(hebench::APIBridge::Handle encoded_data) override;
    hebench::APIBridge::Handle decrypt(hebench::APIBridge::Handle encrypted_data) override;
    hebench::APIBridge::Handle load(const hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    void store(hebench::APIBridge::Handle remote_data, hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    hebench::APIBridge::Handle operate(hebench::APIBridge::Handle h_remote_packed, const hebench::APIBridge::ParameterIndexer *p_param_indexers,
                                       std::uint64_t indexers_count) override;
    std::int64_t classTag() const override { return BaseBenchmark::classTag() | MatMultRowLatencyBenchmark::tag; }

private:
    struct Dims { std::uint64_t rows = 0, cols = 0; };
    struct PlainPack { Dims a, b; std::vector<Plain> A; Plain B; };      // encode() result
    struct CipherPack { Dims a, b; std::vector<Cipher> A; Cipher B; };   // encrypt() result
    struct RemotePack {
        Dims a, b;
        std::shared_ptr<DeviceCiphers> A, B;
        // HE355_NUM_DEVICES > 1: device d > 0 holds ITS block of the row(-pair) ciphertexts of A and all of B (MatMultRow)
        std::vector<std::shared_ptr<DeviceCiphers>> A_dev, B_dev; // [device]; [0] = A, B
    };
    struct ResultRemote { Dims d; std::shared_ptr<DeviceCiphers> C; };
    struct ResultCipher { Dims d; std::vector<Cipher> C; };
    struct ResultPlain { Dims d; std::vector<Plain> C; };
    static const hebench::APIBridge::DataPack &findDataPack(const hebench::APIBridge::DataPackCollection &c, std::uint64_t param_position);
    Scheme m_scheme;
    std::uint64_t rows_M0() const { return m_w[0]; }
    std::uint64_t cols_M0() const { return m_w[1]; }
    std::uint64_t cols_M1() const { return m_w[2]; }
    std::vector<std::uint64_t> m_w;
    HeContextWrapper::Ptr m_p_ctx_wrapper;
    int m_num_devices = 1;
    std::shared_ptr<DeviceGroup> m_group;
    // the body of operate() on one device: nA ciphertexts of A (resident on ctx's device) -> their result ciphertexts
    std::shared_ptr<DeviceCiphers> rowsOn(he355_ctx *ctx, DeviceGroup *group, int device, const DeviceCiphers &A, const DeviceCiphers &B,
                                          std::uint64_t dim2, std::shared_ptr<DeviceCiphers> into, std::uint64_t into_offset);
};

} // namespace mi355x
