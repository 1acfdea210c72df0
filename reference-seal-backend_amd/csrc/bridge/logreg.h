// logreg.h — CKKS logistic-regression inference with a degree-3 sigmoid polynomial (Horner), Latency and Offline:
// mirrors sbe::ckks::LogRegHornerBenchmark (/root/reference/src/benchmarks/ckks/seal_ckks_logreg_horner.cpp) and the
// helpers it calls, SEALContextWrapper::collapseCKKS / evaluatePolynomial / matchLevel (src/engine/seal_context.cpp:349-457).
//   per input X_i : multiply(W, X_i) -> relinearize -> accumulateCKKS(n) -> rescale                       (.cpp:413-416)
//   collapse      : Enc(0) + sum_i rescale(rotate_vector(dot_i, -i) * e_i)                                  (seal_context.cpp:349-415)
//   + bias, then sigmoid(x) ~ 0.5 + 0.15012 x - 0.0015930078125 x^3 by Horner's rule                        (.cpp:446-471)
// On the GPU the per-input loop is one batch; the whole of operate() runs on device ciphertexts.
#pragma once
#include "benchmarks.h"

namespace mi355x {

class LogRegHornerBenchmarkDescription : public hebench::cpp::BenchmarkDescription {
public:
    HEBERROR_DECLARE_CLASS_NAME(LogRegHornerBenchmarkDescription)
    static constexpr std::int64_t LogRegOtherID = 0x01;
    static constexpr const char *AlgorithmName = "HornerPolyEval";                                                                       // ckks logreg .h
    static constexpr const char *AlgorithmDescription = "Horner method for polynomial evaluation, single input vector per ciphertext";
    enum : std::uint64_t { Index_W = 0, Index_b, Index_X, NumOpParams };
    enum : std::uint64_t { Index_n = 0, Index_PolyModulusDegree, Index_NumCoefficientModuli, Index_CoefficientModulusBits, Index_ScaleExponentBits,
                           Index_NumThreads, NumWorkloadParams };
    static constexpr std::size_t DefaultPolyModulusDegree = 16384; // logreg .h:57-61
    LogRegHornerBenchmarkDescription(hebench::APIBridge::Category category, std::size_t batch_size = 0);
    hebench::cpp::BaseBenchmark *createBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::WorkloadParams *p_params) override;
    void destroyBenchmark(hebench::cpp::BaseBenchmark *p_bench) override;
    std::string getBenchmarkDescription(const hebench::APIBridge::WorkloadParams *p_w_params) const override;
};

class LogRegHornerBenchmark : public hebench::cpp::BaseBenchmark {
public:
    HEBERROR_DECLARE_CLASS_NAME(LogRegHornerBenchmark)
    static constexpr std::int64_t tag = 0x80;
    static constexpr std::int64_t EncodedOpParamsTag = 0x10, EncryptedOpParamsTag = 0x20, EncryptedResultTag = 0x40, EncodedResultTag = 0x80; // .h:112-115
    static constexpr double SigmoidPolyCoeff[4] = {0.5, 0.15012, 0.0, -0.0015930078125};                                                        // .h:117
    LogRegHornerBenchmark(hebench::cpp::BaseEngine &engine, const hebench::APIBridge::BenchmarkDescriptor &bench_desc,
                          const hebench::APIBridge::WorkloadParams &bench_params);
    hebench::APIBridge::Handle encode(const hebench::APIBridge::DataPackCollection *p_parameters) override;
    void decode(hebench::APIBridge::Handle encoded_data, hebench::APIBridge::DataPackCollection *p_native) override;
    hebench::APIBridge::Handle encrypt(hebench::APIBridge::Handle encoded_data) override;
    hebench::APIBridge::Handle decrypt(hebench::APIBridge::Handle encrypted_data) override;
    hebench::APIBridge::Handle load(const hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    void store(hebench::APIBridge::Handle remote_data, hebench::APIBridge::Handle *p_local_data, std::uint64_t count) override;
    hebench::APIBridge::Handle operate(hebench::APIBridge::Handle h_remote_packed, const hebench::APIBridge::ParameterIndexer *p_param_indexers,
                                       std::uint64_t indexers_count) override;
    std::int64_t classTag() const override { return BaseBenchmark::classTag() | LogRegHornerBenchmark::tag; }

private:
    struct EncodedOpParams { Plain W, b; std::vector<Plain> X; };
    struct EncryptedOpParams { Cipher W, b; std::vector<Cipher> X; };
    struct RemoteOpParams {
        std::shared_ptr<DeviceCiphers> W, b, X;
        std::shared_ptr<DeviceCiphers> identity; // [batch] plaintexts e_i, top level
        std::shared_ptr<DeviceCiphers> zero;     // Enc(0), top level (collapseCKKS's encrypt_zero)
        std::shared_ptr<DeviceCiphers> coeff;    // [4] sigmoid coefficients, plaintexts, top level
        std::shared_ptr<DeviceCiphers> coeff3;   // Enc(coeff[3]), top level (evaluatePolynomial's first encrypt)
        // The same constants at the levels operate() meets them at (mod_switch_to / matchLevel of a constant drops residues: a function
        // of the parameters alone, prepared at load() like the constants themselves): identity at L-1, Enc(0) / bias / Enc(coeff[3]) at
        // L-2, coeff[k] at L-5+k (the level of Horner step k's product after its rescale)
        std::shared_ptr<DeviceCiphers> identity1, tail2 /* {Enc(0), bias} side by side at L-2 */, coeff3_2, coeff_at[3];
    };
    static const hebench::APIBridge::DataPack &findDataPack(const hebench::APIBridge::DataPackCollection &c, std::uint64_t pos);
    std::shared_ptr<DeviceCiphers> uploadPlains(const std::vector<Plain> &p);
    std::shared_ptr<DeviceCiphers> dropTo(const std::shared_ptr<DeviceCiphers> &x, int L_to); // CKKS mod_switch_to (matchLevel)
    std::uint64_t m_n = 0;
    HeContextWrapper::Ptr m_p_ctx_wrapper;
    std::vector<Plain> m_plain_coeff;
};

} // namespace mi355x
