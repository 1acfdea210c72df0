// matmult_cba.cpp — see matmult_cba.h
#include "matmult_cba.h"

#include <cmath>
#include <sstream>

using namespace mi355x;
using hebench::cpp::HEBenchError;
namespace AB = hebench::APIBridge;

MatMultCipherBatchAxisBenchmarkDescription::MatMultCipherBatchAxisBenchmarkDescription(Scheme scheme) : m_scheme(scheme)
{
    std::memset(&m_descriptor, 0, sizeof(AB::BenchmarkDescriptor));
    m_descriptor.workload = AB::Workload::MatrixMultiply;
    m_descriptor.data_type = scheme == Scheme::CKKS ? AB::DataType::Float64 : AB::DataType::Int64;
    m_descriptor.category = AB::Category::Latency;
    m_descriptor.cat_params.latency.warmup_iterations_count = 1;
    m_descriptor.cat_params.min_test_time_ms = 0;
    m_descriptor.cipher_param_mask = HEBENCH_HE_PARAM_FLAGS_ALL_CIPHER;
    m_descriptor.scheme = scheme == Scheme::CKKS ? HEBENCH_HE_SCHEME_CKKS : HEBENCH_HE_SCHEME_BFV;
    m_descriptor.security = HEBENCH_HE_SECURITY_128;
    m_descriptor.other = MatMultOtherID;
    hebench::cpp::WorkloadParams::Common w; // defaults: ckks cipherbatchaxis .h:30-36, bfv .h:31-37
    w.add<std::uint64_t>(10, "rows_M0");
    w.add<std::uint64_t>(9, "cols_M0");
    w.add<std::uint64_t>(8, "cols_M1");
    w.add<std::uint64_t>(8192, "PolyModulusDegree");
    w.add<std::uint64_t>(3, "MultiplicativeDepth");
    w.add<std::uint64_t>(scheme == Scheme::CKKS ? 45 : 40, "CoefficientModulusBits");
    if (scheme == Scheme::CKKS) w.add<std::uint64_t>(45, "ScaleBits");
    else w.add<std::uint64_t>(20, "PlainModulusBits");
    w.add<std::uint64_t>(0, "NumThreads");
    this->addDefaultParameters(w);
}
hebench::cpp::BaseBenchmark *MatMultCipherBatchAxisBenchmarkDescription::createBenchmark(hebench::cpp::BaseEngine &engine, const AB::WorkloadParams *p_params)
{
    if (!p_params)
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid empty workload parameters. Matrix Multiplication requires parameters."), HEBENCH_ECODE_INVALID_ARGS);
    if ((m_descriptor.cipher_param_mask & 0x03) != 0x03) // all cipher (bfv cipherbatchaxis .cpp:63-75)
        throw HEBenchError(HEBERROR_MSG_CLASS("Cipher/plain combination of operation parameters requested is not supported."), HEBENCH_ECODE_INVALID_ARGS);
    return new MatMultCipherBatchAxisBenchmark(engine, m_descriptor, *p_params, m_scheme);
}
void MatMultCipherBatchAxisBenchmarkDescription::destroyBenchmark(hebench::cpp::BaseBenchmark *p_bench)
{
    if (p_bench) delete p_bench;
}
std::string MatMultCipherBatchAxisBenchmarkDescription::getBenchmarkDescription(const AB::WorkloadParams *p_w_params) const
{
    std::stringstream ss;
    std::string s_tmp = BenchmarkDescription::getBenchmarkDescription(p_w_params);
    if (!p_w_params) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid null workload parameters `p_w_params`"), HEBENCH_ECODE_INVALID_ARGS);
    if (!s_tmp.empty()) ss << s_tmp << std::endl;
    ss << ", Encryption Parameters" << std::endl
       << ", , Poly modulus degree, " << p_w_params->params[Index_PolyModulusDegree].u_param << std::endl
       << ", , Coefficient Modulus, 60";
    for (std::size_t i = 1; i < p_w_params->params[Index_NumCoefficientModuli].u_param; ++i) ss << ", " << p_w_params->params[Index_CoefficientModulusBits].u_param;
    ss << ", 60" << std::endl;
    if (m_scheme == Scheme::CKKS) ss << ", , Scale, 2^" << p_w_params->params[Index_ScaleExponentBits].u_param << std::endl;
    else ss << ", , Plain Modulus, " << p_w_params->params[Index_ScaleExponentBits].u_param << std::endl;
    ss << ", Algorithm, " << AlgorithmName << ", " << AlgorithmDescription << std::endl
       << HeContextWrapper::threadsRow(p_w_params->params[Index_NumThreads].u_param, false) << std::endl // the matrix workloads keep the requested count (bfv row .cpp:89-91)
       << ", Device, AMD Instinct MI355X (HIP; all result elements as one batch)";
    return ss.str();
}

MatMultCipherBatchAxisBenchmark::MatMultCipherBatchAxisBenchmark(hebench::cpp::BaseEngine &engine, const AB::BenchmarkDescriptor &bench_desc,
                                                                 const AB::WorkloadParams &bench_params, Scheme scheme)
    : hebench::cpp::BaseBenchmark(engine, bench_desc, bench_params), m_scheme(scheme)
{
    if (bench_desc.workload != AB::Workload::MatrixMultiply || bench_desc.category != AB::Category::Latency || ((bench_desc.cipher_param_mask & 0x03) != 0x03)
        || bench_desc.security != HEBENCH_HE_SECURITY_128 || bench_desc.other != MatMultCipherBatchAxisBenchmarkDescription::MatMultOtherID)
        throw HEBenchError(HEBERROR_MSG_CLASS("Benchmark descriptor received is not supported."), HEBENCH_ECODE_INVALID_ARGS);
    if (bench_params.count < MatMultCipherBatchAxisBenchmarkDescription::NumWorkloadParams)
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid workload parameters."), HEBENCH_ECODE_INVALID_ARGS);
    for (std::uint64_t i = 0; i < bench_params.count; ++i) m_w.push_back(bench_params.params[i].u_param);
    if (rows_M0() <= 0 || cols_M0() <= 0 || cols_M1() <= 0)
        throw HEBenchError(HEBERROR_MSG_CLASS("Matrix dimensions must be greater than 0."), HEBENCH_ECODE_INVALID_ARGS);
    if (m_w[MatMultCipherBatchAxisBenchmarkDescription::Index_CoefficientModulusBits] < 1)
        throw HEBenchError(HEBERROR_MSG_CLASS("Multiplicative depth must be greater than 0."), HEBENCH_ECODE_INVALID_ARGS);
    const std::uint64_t N = m_w[MatMultCipherBatchAxisBenchmarkDescription::Index_PolyModulusDegree];
    const std::uint64_t depth = m_w[MatMultCipherBatchAxisBenchmarkDescription::Index_NumCoefficientModuli];
    const int bits = (int)m_w[MatMultCipherBatchAxisBenchmarkDescription::Index_CoefficientModulusBits];
    const int extra = (int)m_w[MatMultCipherBatchAxisBenchmarkDescription::Index_ScaleExponentBits];
    m_p_ctx_wrapper = scheme == Scheme::CKKS ? HeContextWrapper::createCKKSContext(N, depth, bits, extra) : HeContextWrapper::createBFVContext(N, depth, bits, extra);
    m_p_ctx_wrapper->prepareClient(256);
}

namespace {
// M0 is kept column-major ([k][i]) and M1 row-major ([k][j]): for a fixed inner index k both operands of the
// rows_M0 x cols_M1 outer product are then contiguous runs of ciphertexts.
struct MatPlain { std::vector<Plain> m[2]; };
struct MatCipher { std::vector<Cipher> m[2]; };
struct MatRemote { std::shared_ptr<DeviceCiphers> m[2]; };
} // namespace

AB::Handle MatMultCipherBatchAxisBenchmark::encode(const AB::DataPackCollection *p_parameters)
{
    if (p_parameters->pack_count != 2)
        throw HEBenchError(HEBERROR_MSG_CLASS("Expected 2 parameter packs, but " + std::to_string(p_parameters->pack_count) + " received."), HEBENCH_ECODE_INVALID_ARGS);
    MatPlain out;
    for (std::uint64_t op = 0; op < 2; ++op) {
        const AB::DataPack *dp = nullptr;
        for (std::uint64_t i = 0; !dp && i < p_parameters->pack_count; ++i)
            if (p_parameters->p_data_packs[i].param_position == op) dp = &p_parameters->p_data_packs[i];
        if (!dp) throw HEBenchError(HEBERROR_MSG_CLASS("Operation parameter " + std::to_string(op) + " not found in 'p_parameters'."), HEBENCH_ECODE_INVALID_ARGS);
        if (dp->buffer_count < 1)
            throw HEBenchError(HEBERROR_MSG_CLASS("Latency test requires, at least, 1 sample per operation parameter. None found for operation parameter "
                                                  + std::to_string(op) + "."), HEBENCH_ECODE_INVALID_ARGS);
        if (!dp->p_buffers || !dp->p_buffers[0].p) throw HEBenchError(HEBERROR_MSG_CLASS("Unexpected empty buffer in data pack."), HEBENCH_ECODE_CRITICAL_ERROR);
        const std::uint64_t r = op ? cols_M0() : rows_M0(), c = op ? cols_M1() : cols_M0();
        if (dp->p_buffers[0].size < r * c * 8) throw HEBenchError(HEBERROR_MSG_CLASS("Insufficient data for matrix."), HEBENCH_ECODE_INVALID_ARGS);
        // storage order: M0 column-major, M1 row-major; encoded in groups by encodeBatch (same bits as one encodeVector per element)
        out.m[op].reserve(r * c);
        const std::uint64_t kGroup = 4096;
        for (std::uint64_t k0 = 0; k0 < r * c; k0 += kGroup) {
            const std::uint64_t k1 = std::min<std::uint64_t>(r * c, k0 + kGroup);
            std::vector<Plain> enc;
            if (m_scheme == Scheme::CKKS) { // CKKSEncoder::encode(double, scale, plain): the value in every slot (ckks .cpp:204)
                std::vector<std::vector<double>> vecs;
                for (std::uint64_t k = k0; k < k1; ++k) {
                    const std::uint64_t row = op ? k / c : k % r, col = op ? k % c : k / r;
                    vecs.emplace_back(m_p_ctx_wrapper->slot_count(), reinterpret_cast<const double *>(dp->p_buffers[0].p)[row * c + col]);
                }
                enc = m_p_ctx_wrapper->encodeBatch(vecs);
            } else { // BatchEncoder::encode(span of 1): slot 0 only (bfv .cpp:201-203)
                std::vector<std::vector<std::int64_t>> vecs;
                for (std::uint64_t k = k0; k < k1; ++k) {
                    const std::uint64_t row = op ? k / c : k % r, col = op ? k % c : k / r;
                    vecs.emplace_back(1, reinterpret_cast<const std::int64_t *>(dp->p_buffers[0].p)[row * c + col]);
                }
                enc = m_p_ctx_wrapper->encodeBatch(vecs);
            }
            out.m[op].insert(out.m[op].end(), std::make_move_iterator(enc.begin()), std::make_move_iterator(enc.end()));
        }
    }
    return this->getEngine().createHandle<decltype(out)>(sizeof(out), 0, std::move(out));
}

void MatMultCipherBatchAxisBenchmark::decode(AB::Handle h_encoded_data, AB::DataPackCollection *p_native)
{
    if (p_native->pack_count == 0) return;
    if (!p_native->p_data_packs) throw HEBenchError(HEBERROR_MSG_CLASS("Unexpected empty 'p_native->p_data_packs'."), HEBENCH_ECODE_CRITICAL_ERROR);
    const std::vector<Plain> &res = this->getEngine().retrieveFromHandle<std::vector<Plain>>(h_encoded_data); // row-major rows_M0 x cols_M1
    const AB::DataPack &rc = p_native->p_data_packs[0];
    if (rc.buffer_count == 0) return;
    if (!rc.p_buffers) throw HEBenchError(HEBERROR_MSG_CLASS("Unexpected empty buffer in data pack."), HEBENCH_ECODE_CRITICAL_ERROR);
    if (rc.p_buffers[0].size == 0) return;
    if (!rc.p_buffers[0].p) throw HEBenchError(HEBERROR_MSG_CLASS("Unexpected empty buffer in data pack."), HEBENCH_ECODE_CRITICAL_ERROR);
    const std::size_t room = rc.p_buffers[0].size / 8;
    const std::size_t total = std::min(res.size(), room), kGroup = 4096; // decode as much as fits (ckks .cpp:246-262), in groups
    for (std::size_t k0 = 0; k0 < total; k0 += kGroup) {
        const std::size_t k1 = std::min(total, k0 + kGroup);
        const std::vector<Plain> group(res.begin() + k0, res.begin() + k1);
        if (m_scheme == Scheme::CKKS) {
            const auto vals = m_p_ctx_wrapper->decodeSlotsCKKS(group, HeContextWrapper::SlotRanges{{0, 1}}); // slot 0 is the entry
            for (std::size_t k = k0; k < k1; ++k) {
                const double v0 = vals[k - k0];
                reinterpret_cast<double *>(rc.p_buffers[0].p)[k] = std::abs(v0) < 0.00005 ? 0.0 : v0;
            }
        } else {
            const auto vals = m_p_ctx_wrapper->decodeSlotsBFV(group, HeContextWrapper::SlotRanges{{0, 1}});
            for (std::size_t k = k0; k < k1; ++k) reinterpret_cast<std::int64_t *>(rc.p_buffers[0].p)[k] = vals[k - k0];
        }
    }
}

AB::Handle MatMultCipherBatchAxisBenchmark::encrypt(AB::Handle h_encoded_data)
{
    const MatPlain &p = this->getEngine().retrieveFromHandle<MatPlain>(h_encoded_data);
    MatCipher c;
    for (int op = 0; op < 2; ++op) c.m[op] = m_p_ctx_wrapper->encryptBatch(p.m[op]);
    return this->getEngine().createHandle<decltype(c)>(sizeof(c), 0, std::move(c));
}

AB::Handle MatMultCipherBatchAxisBenchmark::decrypt(AB::Handle h_encrypted_data)
{
    const std::vector<Cipher> &c = this->getEngine().retrieveFromHandle<std::vector<Cipher>>(h_encrypted_data);
    std::vector<Plain> p = m_p_ctx_wrapper->decryptBatch(c);
    return this->getEngine().createHandle<decltype(p)>(sizeof(p), 0, std::move(p));
}

AB::Handle MatMultCipherBatchAxisBenchmark::load(const AB::Handle *p_h_local_data, std::uint64_t count)
{
    if (count != 1) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid number of handles. Expected 1."), HEBENCH_ECODE_INVALID_ARGS);
    const MatCipher &c = this->getEngine().retrieveFromHandle<MatCipher>(p_h_local_data[0]);
    MatRemote r;
    for (int op = 0; op < 2; ++op) r.m[op] = m_p_ctx_wrapper->upload(c.m[op]);
    m_p_ctx_wrapper->needRelinKey();
    return this->getEngine().createHandle<decltype(r)>(sizeof(r), 0, std::move(r));
}

void MatMultCipherBatchAxisBenchmark::store(AB::Handle h_remote_data, AB::Handle *p_h_local_data, std::uint64_t count)
{
    if (count > 0) {
        std::memset(p_h_local_data, 0, sizeof(AB::Handle) * count);
        const std::shared_ptr<DeviceCiphers> &r = this->getEngine().retrieveFromHandle<std::shared_ptr<DeviceCiphers>>(h_remote_data);
        std::vector<Cipher> local = m_p_ctx_wrapper->download(r);
        p_h_local_data[0] = this->getEngine().createHandle<decltype(local)>(sizeof(local), 0, std::move(local));
    }
}

AB::Handle MatMultCipherBatchAxisBenchmark::operate(AB::Handle h_remote_packed, const AB::ParameterIndexer *p_param_indexers, std::uint64_t indexers_count)
{
    if (indexers_count < std::uint64_t(2)) { // ckks cipherbatchaxis .cpp:352-358
        std::stringstream ss;
        ss << "Invalid number of indexers. Expected " << std::uint64_t(2) << ", but " << indexers_count << " received." << std::endl;
        throw HEBenchError(HEBERROR_MSG_CLASS(ss.str()), HEBENCH_ECODE_INVALID_ARGS);
    }
    for (int i = 0; i < 2; ++i) {
        if (p_param_indexers[i].value_index > 0) throw HEBenchError(HEBERROR_MSG_CLASS("Unexpected index in parameter indexer."), HEBENCH_ECODE_INVALID_ARGS);
        if (p_param_indexers[i].batch_size > 1) throw HEBenchError(HEBERROR_MSG_CLASS("Batch size must be 1 for latency test."), HEBENCH_ECODE_INVALID_ARGS);
    }
    const MatRemote &in = this->getEngine().retrieveFromHandle<MatRemote>(h_remote_packed);
    if (!in.m[0] || !in.m[1]) // .cpp:371-373
        throw HEBenchError(HEBERROR_MSG_CLASS("Insufficient number of arguments for operation parameters."), HEBENCH_ECODE_INVALID_ARGS);
    he355_ctx *ctx = m_p_ctx_wrapper->raw();
    const int L = in.m[0]->L;
    const std::uint64_t r0 = rows_M0(), c0 = cols_M0(), c1 = cols_M1(), n = r0 * c1;
    std::shared_ptr<DeviceCiphers> result;
    if (m_scheme == Scheme::CKKS) {
        const double sc = in.m[0]->scale * in.m[1]->scale / (double)m_p_ctx_wrapper->params().primes[L - 1].q;
        std::shared_ptr<DeviceCiphers> c3 = m_p_ctx_wrapper->allocResult(n, 3, L, in.m[0]->scale * in.m[1]->scale);
        result = m_p_ctx_wrapper->allocResult(n, 2, L - 1, sc);
        // M0(i,k) at k*r0 + i, M1(k,j) at k*c1 + j
        HeContextWrapper::check(he355_multiply_accumulate(ctx, L, r0, c1, c0, in.m[0]->d, 1, r0, in.m[1]->d, c1, 1, c3->d), "multiply+add");
        HeContextWrapper::check(he355_relinearize_rescale(ctx, L, n, c3->d, result->d), "relinearize+rescale");
        HeContextWrapper::check(he355_sync(ctx), "synchronise");
    } else {
        result = m_p_ctx_wrapper->allocResult(n, 2, L, 1.0);
        // multiply / relinearize_inplace / add_inplace over the inner index (bfv cipherbatchaxis .cpp:398-410), the inner index inside the batch
        HeContextWrapper::check(he355_bfv_multiply_relin_accumulate(ctx, L, r0, c1, c0, in.m[0]->d, 1, r0, in.m[1]->d, c1, 1, result->d),
                                "multiply+relinearize+add");
        HeContextWrapper::check(he355_sync(ctx), "synchronise");
    }
    return this->getEngine().createHandle<decltype(result)>(sizeof(result), 0, std::move(result));
}
