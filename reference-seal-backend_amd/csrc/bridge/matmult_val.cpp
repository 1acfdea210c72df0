// matmult_val.cpp — see matmult_val.h
#include "matmult_val.h"

#include <cmath>
#include <sstream>

using namespace mi355x;
using hebench::cpp::HEBenchError;
namespace AB = hebench::APIBridge;

MatMultValBenchmarkDescription::MatMultValBenchmarkDescription(Scheme scheme) : m_scheme(scheme)
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
    m_descriptor.other = MatMultValOtherID;
    hebench::cpp::WorkloadParams::Common w; // defaults: ckks matmultval .h:26-32, bfv .h:29-35
    w.add<std::uint64_t>(10, "rows_M0");
    w.add<std::uint64_t>(9, "cols_M0");
    w.add<std::uint64_t>(8, "cols_M1");
    w.add<std::uint64_t>(8192, "PolyModulusDegree");
    w.add<std::uint64_t>(2, "MultiplicativeDepth");
    w.add<std::uint64_t>(scheme == Scheme::CKKS ? 45 : 40, scheme == Scheme::CKKS ? "CoefficientMudulusBits" : "CoefficientModulusBits"); // sic: ckks matmultval .cpp:50
    if (scheme == Scheme::CKKS) w.add<std::uint64_t>(45, "ScaleBits");
    else w.add<std::uint64_t>(20, "PlainModulusBits");
    w.add<std::uint64_t>(0, "NumThreads");
    this->addDefaultParameters(w);
}
hebench::cpp::BaseBenchmark *MatMultValBenchmarkDescription::createBenchmark(hebench::cpp::BaseEngine &engine, const AB::WorkloadParams *p_params)
{
    if (!p_params) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid empty workload parameters. This workload requires flexible parameters."), HEBENCH_ECODE_CRITICAL_ERROR);
    return new MatMultValBenchmark(engine, m_descriptor, *p_params, m_scheme);
}
void MatMultValBenchmarkDescription::destroyBenchmark(hebench::cpp::BaseBenchmark *p_bench)
{
    if (p_bench) delete p_bench;
}
std::string MatMultValBenchmarkDescription::getBenchmarkDescription(const AB::WorkloadParams *p_w_params) const
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
       << ", Device, AMD Instinct MI355X (HIP; all (row, column) pairs as one batch)";
    return ss.str();
}

MatMultValBenchmark::MatMultValBenchmark(hebench::cpp::BaseEngine &engine, const AB::BenchmarkDescriptor &bench_desc, const AB::WorkloadParams &bench_params,
                                         Scheme scheme)
    : hebench::cpp::BaseBenchmark(engine, bench_desc, bench_params), m_scheme(scheme)
{
    if (bench_desc.workload != AB::Workload::MatrixMultiply || bench_desc.category != AB::Category::Latency || ((bench_desc.cipher_param_mask & 0x03) != 0x03)
        || bench_desc.security != HEBENCH_HE_SECURITY_128 || bench_desc.other != MatMultValBenchmarkDescription::MatMultValOtherID)
        throw HEBenchError(HEBERROR_MSG_CLASS("Benchmark descriptor received is not supported."), HEBENCH_ECODE_INVALID_ARGS);
    if (bench_params.count < MatMultValBenchmarkDescription::NumWorkloadParams)
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid workload parameters."), HEBENCH_ECODE_INVALID_ARGS);
    for (std::uint64_t i = 0; i < bench_params.count; ++i) m_w.push_back(bench_params.params[i].u_param);
    const std::uint64_t N = m_w[MatMultValBenchmarkDescription::Index_PolyModulusDegree];
    if (rows_M0() <= 0 || cols_M0() <= 0 || cols_M1() <= 0)
        throw HEBenchError(HEBERROR_MSG_CLASS("Matrix dimensions must be greater than 0."), HEBENCH_ECODE_INVALID_ARGS);
    const std::uint64_t max_cols = scheme == Scheme::CKKS ? N / 2 : N; // ckks .cpp:147, bfv .cpp:147
    if (cols_M0() > max_cols) {
        std::stringstream ss;
        ss << "Invalid workload parameters. This workload only supports matrices of dimensions (n x " << max_cols << ") x (" << max_cols << " x m).";
        throw HEBenchError(HEBERROR_MSG_CLASS(ss.str()), HEBENCH_ECODE_INVALID_ARGS);
    }
    if (m_w[MatMultValBenchmarkDescription::Index_CoefficientModulusBits] < 1)
        throw HEBenchError(HEBERROR_MSG_CLASS("Multiplicative depth must be greater than 0."), HEBENCH_ECODE_INVALID_ARGS);
    const std::uint64_t depth = m_w[MatMultValBenchmarkDescription::Index_NumCoefficientModuli];
    const int bits = (int)m_w[MatMultValBenchmarkDescription::Index_CoefficientModulusBits], extra = (int)m_w[MatMultValBenchmarkDescription::Index_ScaleExponentBits];
    m_p_ctx_wrapper = scheme == Scheme::CKKS ? HeContextWrapper::createCKKSContext(N, depth, bits, extra) : HeContextWrapper::createBFVContext(N, depth, bits, extra);
    m_p_ctx_wrapper->prepareClient(256);
}

const AB::DataPack &MatMultValBenchmark::findDataPack(const AB::DataPackCollection &c, std::uint64_t pos)
{
    for (std::uint64_t i = 0; i < c.pack_count; ++i)
        if (c.p_data_packs[i].param_position == pos) return c.p_data_packs[i];
    throw HEBenchError(HEBERROR_MSG_CLASS("DataPack for component @ " + std::to_string(pos) + " not found."), HEBENCH_ECODE_INVALID_ARGS);
}

namespace {
struct MatPlain { std::vector<Plain> rows[2]; };               // [0]: rows of M0, [1]: rows of M1^T
struct MatCipher { std::vector<Cipher> rows[2]; };
struct MatRemote { std::shared_ptr<DeviceCiphers> m[2]; };
} // namespace

AB::Handle MatMultValBenchmark::encode(const AB::DataPackCollection *p_parameters)
{
    MatPlain out;
    for (int op = 0; op < 2; ++op) {
        const AB::DataPack &dp = findDataPack(*p_parameters, op);
        if (dp.buffer_count <= 0 || !dp.p_buffers) throw HEBenchError(HEBERROR_MSG_CLASS(op ? "Invalid empty data for M1." : "Invalid empty data for M0."), HEBENCH_ECODE_INVALID_ARGS);
        const AB::NativeDataBuffer &buf = dp.p_buffers[0];
        const std::uint64_t r = op ? cols_M0() : rows_M0(), c = op ? cols_M1() : cols_M0();
        if (!buf.p || buf.size < r * c * 8) throw HEBenchError(HEBERROR_MSG_CLASS("Insufficient data for M0."), HEBENCH_ECODE_INVALID_ARGS);
        // M0: one plaintext per row.  M1: transposed first (ckks .cpp:213-225), so one plaintext per column of M1.
        const std::uint64_t n_vec = op ? c : r, len = op ? r : c;
        // one encodeBatch call per matrix (same bits as one encodeVector per row)
        if (m_scheme == Scheme::CKKS) {
            const double *p = reinterpret_cast<const double *>(buf.p);
            std::vector<std::vector<double>> vecs(n_vec, std::vector<double>(len));
            for (std::uint64_t v = 0; v < n_vec; ++v)
                for (std::uint64_t k = 0; k < len; ++k) vecs[v][k] = op ? p[k * c + v] : p[v * c + k];
            out.rows[op] = m_p_ctx_wrapper->encodeBatch(vecs);
        } else {
            const std::int64_t *p = reinterpret_cast<const std::int64_t *>(buf.p);
            std::vector<std::vector<std::int64_t>> vecs(n_vec, std::vector<std::int64_t>(len));
            for (std::uint64_t v = 0; v < n_vec; ++v)
                for (std::uint64_t k = 0; k < len; ++k) vecs[v][k] = op ? p[k * c + v] : p[v * c + k];
            out.rows[op] = m_p_ctx_wrapper->encodeBatch(vecs);
        }
    }
    return this->getEngine().createHandle<decltype(out)>(sizeof(out), 0, std::move(out));
}

void MatMultValBenchmark::decode(AB::Handle h_encoded_data, AB::DataPackCollection *p_native)
{
    const std::vector<Plain> &res = this->getEngine().retrieveFromHandle<std::vector<Plain>>(h_encoded_data); // row-major rows_M0 x cols_M1
    if (res.size() < rows_M0() * cols_M1()) { // the reference keeps rows of columns (ckks matmultval .cpp:322-338); here the result is flat, row-major
        std::stringstream ss;
        ss << "Invalid number of rows in encoded result. Expected " << rows_M0() << ", but received " << res.size() / std::max<std::uint64_t>(1, cols_M1()) << ".";
        throw HEBenchError(HEBERROR_MSG_CLASS(ss.str()), HEBENCH_ECODE_INVALID_ARGS);
    }
    const AB::DataPack &rc = findDataPack(*p_native, 0);
    if (rc.buffer_count == 0 || !rc.p_buffers || !rc.p_buffers[0].p) return;
    const std::size_t room = rc.p_buffers[0].size / 8;
    const std::size_t total = std::min<std::size_t>(rows_M0() * cols_M1(), room), kGroup = 4096; // decoded in groups: bounded host memory
    for (std::size_t k0 = 0; k0 < total; k0 += kGroup) {
        const std::size_t k1 = std::min(total, k0 + kGroup);
        const std::vector<Plain> group(res.begin() + k0, res.begin() + k1);
        if (m_scheme == Scheme::CKKS) {
            const auto vals = m_p_ctx_wrapper->decodeSlotsCKKS(group, HeContextWrapper::SlotRanges{{0, 1}}); // slot 0 is the entry
            for (std::size_t k = k0; k < k1; ++k) {
                const double v0 = vals[k - k0];
                reinterpret_cast<double *>(rc.p_buffers[0].p)[k] = std::abs(v0) < 0.00005 ? 0.0 : v0; // ckks .cpp:356-359
            }
        } else {
            const auto vals = m_p_ctx_wrapper->decodeSlotsBFV(group, HeContextWrapper::SlotRanges{{0, 1}});
            for (std::size_t k = k0; k < k1; ++k) reinterpret_cast<std::int64_t *>(rc.p_buffers[0].p)[k] = vals[k - k0];
        }
    }
}

AB::Handle MatMultValBenchmark::encrypt(AB::Handle h_encoded_data)
{
    const MatPlain &p = this->getEngine().retrieveFromHandle<MatPlain>(h_encoded_data);
    MatCipher c;
    for (int op = 0; op < 2; ++op) c.rows[op] = m_p_ctx_wrapper->encryptBatch(p.rows[op]);
    return this->getEngine().createHandle<decltype(c)>(sizeof(c), 0, std::move(c));
}

AB::Handle MatMultValBenchmark::decrypt(AB::Handle h_encrypted_data)
{
    const std::vector<Cipher> &c = this->getEngine().retrieveFromHandle<std::vector<Cipher>>(h_encrypted_data);
    std::vector<Plain> p = m_p_ctx_wrapper->decryptBatch(c);
    return this->getEngine().createHandle<decltype(p)>(sizeof(p), 0, std::move(p));
}

AB::Handle MatMultValBenchmark::load(const AB::Handle *p_h_local_data, std::uint64_t count)
{
    if (count != 1) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid number of handles. Expected 1."), HEBENCH_ECODE_INVALID_ARGS);
    const MatCipher &c = this->getEngine().retrieveFromHandle<MatCipher>(p_h_local_data[0]);
    MatRemote r;
    for (int op = 0; op < 2; ++op) r.m[op] = m_p_ctx_wrapper->upload(c.rows[op]);
    m_p_ctx_wrapper->needRelinKey();
    const std::uint64_t row = m_scheme == Scheme::CKKS ? m_p_ctx_wrapper->slot_count() : m_p_ctx_wrapper->slot_count() / 2;
    const std::uint64_t cnt = std::min<std::uint64_t>(cols_M0(), row);
    int rotations = 64 - __builtin_clzll(cnt);
    if (((std::uint64_t)1 << (rotations - 1)) == cnt) --rotations;
    for (int i = 0; i < rotations; ++i) m_p_ctx_wrapper->needRotationKey(1 << i);
    if (m_scheme == Scheme::BFV && cols_M0() > row) m_p_ctx_wrapper->needRotationKey(0);
    return this->getEngine().createHandle<decltype(r)>(sizeof(r), 0, std::move(r));
}

void MatMultValBenchmark::store(AB::Handle h_remote_data, AB::Handle *p_h_local_data, std::uint64_t count)
{
    if (count > 0 && !p_h_local_data) // bfv matmultval .cpp:418-420
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid null array of handles: \"p_local_data\""), HEBENCH_ECODE_INVALID_ARGS);
    if (count > 0) {
        std::memset(p_h_local_data, 0, sizeof(AB::Handle) * count);
        const std::shared_ptr<DeviceCiphers> &r = this->getEngine().retrieveFromHandle<std::shared_ptr<DeviceCiphers>>(h_remote_data);
        std::vector<Cipher> local = m_p_ctx_wrapper->download(r);
        p_h_local_data[0] = this->getEngine().createHandle<decltype(local)>(sizeof(local), 0, std::move(local));
    }
}

AB::Handle MatMultValBenchmark::operate(AB::Handle h_remote_packed, const AB::ParameterIndexer *p_param_indexers, std::uint64_t indexers_count)
{
    if (indexers_count < std::uint64_t(2)) { // ckks matmultval .cpp:436-442
        std::stringstream ss;
        ss << "Invalid number of indexers. Expected " << std::uint64_t(2) << ", but " << indexers_count << " received." << std::endl;
        throw HEBenchError(HEBERROR_MSG_CLASS(ss.str()), HEBENCH_ECODE_INVALID_ARGS);
    }
    for (std::size_t i = 0; i < std::uint64_t(2); ++i)
        if (p_param_indexers[i].value_index > 0 || p_param_indexers[i].batch_size != 1) { // :450-459
            std::stringstream ss;
            ss << "Invalid parameter indexer for operation parameter " << i << ". Expected index in range [0, 1), but [" << p_param_indexers[i].value_index << ", "
               << p_param_indexers[i].batch_size << ") received.";
            throw HEBenchError(HEBERROR_MSG_CLASS(ss.str()), HEBENCH_ECODE_INVALID_ARGS);
        }
    const MatRemote &in = this->getEngine().retrieveFromHandle<MatRemote>(h_remote_packed);
    he355_ctx *ctx = m_p_ctx_wrapper->raw();
    const int L = in.m[0]->L;
    const std::uint64_t n = rows_M0() * cols_M1();
    he355_indexer ix{0, 0, cols_M1(), 0, 0}; // result i*cols_M1 + j <- (M0[i], M1T[j]): doMatMultVal's collapse(2) loop
    std::shared_ptr<DeviceCiphers> result;
    if (m_scheme == Scheme::CKKS) {
        const double sc = in.m[0]->scale * in.m[1]->scale / (double)m_p_ctx_wrapper->params().primes[L - 1].q;
        result = m_p_ctx_wrapper->allocResult(n, 2, L - 1, sc);
        HeContextWrapper::check(he355_multiply_relin(ctx, L, n, in.m[0]->d, in.m[1]->d, ix, 1, result->d), "multiply+relinearize+rescale"); // :253-255
    } else {
        std::shared_ptr<DeviceCiphers> c3 = m_p_ctx_wrapper->allocResult(n, 3, L, 1.0);
        result = m_p_ctx_wrapper->allocResult(n, 2, L, 1.0);
        HeContextWrapper::check(he355_bfv_multiply(ctx, L, n, in.m[0]->d, in.m[1]->d, ix, c3->d), "multiply");
        HeContextWrapper::check(he355_relinearize(ctx, L, n, c3->d, result->d), "relinearize");
    }
    std::shared_ptr<DeviceCiphers> tmp = m_p_ctx_wrapper->allocResult(n, 2, result->L, result->scale);
    HeContextWrapper::check(he355_accumulate(ctx, result->L, n, result->d, cols_M0(), tmp->d), "accumulate"); // :256 / bfv :255
    HeContextWrapper::check(he355_sync(ctx), "synchronise");
    return this->getEngine().createHandle<decltype(result)>(sizeof(result), 0, std::move(result));
}
