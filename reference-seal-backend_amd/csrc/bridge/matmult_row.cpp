// matmult_row.cpp — see matmult_row.h
#include "matmult_row.h"

#include <algorithm>
#include <cmath>
#include <sstream>

using namespace mi355x;
using hebench::cpp::HEBenchError;
namespace AB = hebench::APIBridge;

MatMultRowBenchmarkDescription::MatMultRowBenchmarkDescription(Scheme scheme) : m_scheme(scheme)
{
    std::memset(&m_descriptor, 0, sizeof(AB::BenchmarkDescriptor)); // bfv row .cpp:28-38
    m_descriptor.workload = AB::Workload::MatrixMultiply;
    m_descriptor.data_type = scheme == Scheme::CKKS ? AB::DataType::Float64 : AB::DataType::Int64;
    m_descriptor.category = AB::Category::Latency;
    m_descriptor.cat_params.latency.warmup_iterations_count = 1;
    m_descriptor.cat_params.min_test_time_ms = 0;
    m_descriptor.cipher_param_mask = HEBENCH_HE_PARAM_FLAGS_ALL_CIPHER;
    m_descriptor.scheme = scheme == Scheme::CKKS ? HEBENCH_HE_SCHEME_CKKS : HEBENCH_HE_SCHEME_BFV;
    m_descriptor.security = HEBENCH_HE_SECURITY_128;
    m_descriptor.other = MatMultRowOtherID;
    hebench::cpp::WorkloadParams::Common w; // defaults: bfv row .cpp:41-48, .h:29-32; ckks row .h:29-32
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
hebench::cpp::BaseBenchmark *MatMultRowBenchmarkDescription::createBenchmark(hebench::cpp::BaseEngine &engine, const AB::WorkloadParams *p_params)
{
    if (!p_params) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid empty workload parameters. This workload requires flexible parameters."), HEBENCH_ECODE_CRITICAL_ERROR);
    return new MatMultRowLatencyBenchmark(engine, m_descriptor, *p_params, m_scheme);
}
void MatMultRowBenchmarkDescription::destroyBenchmark(hebench::cpp::BaseBenchmark *p_bench)
{
    if (p_bench) delete p_bench;
}
std::string MatMultRowBenchmarkDescription::getBenchmarkDescription(const AB::WorkloadParams *p_w_params) const
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
    if (m_scheme == Scheme::CKKS) ss << ", , Scale, 2^" << p_w_params->params[Index_PlainModulusBits].u_param << std::endl;
    else ss << ", , Plain Modulus, " << p_w_params->params[Index_PlainModulusBits].u_param << std::endl;
    ss << ", Algorithm, " << AlgorithmName << ", " << AlgorithmDescription << std::endl
       << HeContextWrapper::threadsRow(p_w_params->params[Index_NumThreads].u_param, false) << std::endl // the matrix workloads keep the requested count (bfv row .cpp:89-91)
       << ", Device, AMD Instinct MI355X (HIP; row-pair ciphertexts batched on the grid)";
    return ss.str();
}

MatMultRowLatencyBenchmark::MatMultRowLatencyBenchmark(hebench::cpp::BaseEngine &engine, const AB::BenchmarkDescriptor &bench_desc,
                                                       const AB::WorkloadParams &bench_params, Scheme scheme)
    : hebench::cpp::BaseBenchmark(engine, bench_desc, bench_params), m_scheme(scheme)
{
    if (bench_desc.workload != AB::Workload::MatrixMultiply || bench_desc.category != AB::Category::Latency
        || bench_desc.data_type != (scheme == Scheme::CKKS ? AB::DataType::Float64 : AB::DataType::Int64)
        || bench_desc.scheme != (scheme == Scheme::CKKS ? HEBENCH_HE_SCHEME_CKKS : HEBENCH_HE_SCHEME_BFV)
        || ((bench_desc.cipher_param_mask & 0x03) != 0x03) || bench_desc.security != HEBENCH_HE_SECURITY_128
        || bench_desc.other != MatMultRowBenchmarkDescription::MatMultRowOtherID)
        throw HEBenchError(HEBERROR_MSG_CLASS("Benchmark descriptor received is not supported."), HEBENCH_ECODE_INVALID_ARGS);
    if (bench_params.count < MatMultRowBenchmarkDescription::NumWorkloadParams)
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid workload parameters."), HEBENCH_ECODE_INVALID_ARGS);
    for (std::uint64_t i = 0; i < bench_params.count; ++i) m_w.push_back(bench_params.params[i].u_param);
    const std::uint64_t N = m_w[MatMultRowBenchmarkDescription::Index_PolyModulusDegree];
    if (rows_M0() <= 0 || cols_M0() <= 0 || cols_M1() <= 0)
        throw HEBenchError(HEBERROR_MSG_CLASS("Matrix dimensions must be greater than 0."), HEBENCH_ECODE_INVALID_ARGS);
    if (m_w[MatMultRowBenchmarkDescription::Index_CoefficientModulusBits] < 1)
        throw HEBenchError(HEBERROR_MSG_CLASS("Multiplicative depth must be greater than 0."), HEBENCH_ECODE_INVALID_ARGS);
    if (cols_M0() > (N / 2) || cols_M0() * cols_M1() > (N / 2)) { // bfv row .cpp:141-147
        std::stringstream ss;
        ss << "Invalid workload parameters. This workload only supports matrices of dimensions (a x b) x (b x c) where 'b' and b * c is at max " << (N / 2)
           << " (e.g. PolyModulusDegree / 2).";
        throw HEBenchError(HEBERROR_MSG_CLASS(ss.str()), HEBENCH_ECODE_INVALID_ARGS);
    }
    const std::uint64_t depth = m_w[MatMultRowBenchmarkDescription::Index_NumCoefficientModuli];
    const int bits = (int)m_w[MatMultRowBenchmarkDescription::Index_CoefficientModulusBits], extra = (int)m_w[MatMultRowBenchmarkDescription::Index_PlainModulusBits];
    m_p_ctx_wrapper = scheme == Scheme::CKKS ? HeContextWrapper::createCKKSContext(N, depth, bits, extra) : HeContextWrapper::createBFVContext(N, depth, bits, extra);
    m_p_ctx_wrapper->prepareClient(256, cols_M1()); // (decode() reads dim3 slots per batching row)
    m_num_devices = DeviceGroup::resolveCount(0); // HE355_NUM_DEVICES: the declared workload parameters stay the reference's
}

const AB::DataPack &MatMultRowLatencyBenchmark::findDataPack(const AB::DataPackCollection &c, std::uint64_t param_position)
{
    for (std::uint64_t i = 0; i < c.pack_count; ++i)
        if (c.p_data_packs[i].param_position == param_position) return c.p_data_packs[i];
    throw HEBenchError(HEBERROR_MSG_CLASS("DataPack for component @ " + std::to_string(param_position) + " not found."), HEBENCH_ECODE_INVALID_ARGS);
}

AB::Handle MatMultRowLatencyBenchmark::encode(const AB::DataPackCollection *p_parameters)
{
    if (p_parameters->pack_count != MatMultRowBenchmarkDescription::NumOpParams)
        throw HEBenchError(HEBERROR_MSG_CLASS("Expected 2 parameter packs, but " + std::to_string(p_parameters->pack_count) + " received."),
                           HEBENCH_ECODE_INVALID_ARGS);
    const void *mats_raw[2];
    for (std::uint64_t op = 0; op < 2; ++op) {
        const std::uint64_t r = op ? cols_M0() : rows_M0(), cl = op ? cols_M1() : cols_M0();
        const AB::DataPack &dp = findDataPack(*p_parameters, op);
        if (dp.buffer_count < 1)
            // (the reference's text names parameter 0 for either operand: bfv row .cpp:192, ckks row .cpp:193)
            throw HEBenchError(HEBERROR_MSG_CLASS("Latency test requires, at least, 1 sample per operation parameter. None found for operation parameter 0."),
                               HEBENCH_ECODE_INVALID_ARGS);
        if (!dp.p_buffers || !dp.p_buffers[0].p) throw HEBenchError(HEBERROR_MSG_CLASS("Unexpected empty buffer in data pack."), HEBENCH_ECODE_CRITICAL_ERROR);
        if (dp.p_buffers[0].size / sizeof(std::int64_t) < r * cl)
            throw HEBenchError(HEBERROR_MSG_CLASS("Insufficient data for parameter 0 sample."), HEBENCH_ECODE_CRITICAL_ERROR); // :203 / :204
        mats_raw[op] = dp.p_buffers[0].p;
    }
    const std::size_t dim1 = rows_M0(), dim2 = cols_M0(), dim3 = cols_M1();
    if (m_scheme == Scheme::CKKS) { // encodeM0 / encodeM1, ckks row .cpp:215-283: one row per ciphertext, spacers = slots / dim2
        const double *m0 = reinterpret_cast<const double *>(mats_raw[0]), *m1 = reinterpret_cast<const double *>(mats_raw[1]);
        const std::size_t slots = m_p_ctx_wrapper->slot_count(), spacers = slots / dim2;
        PlainPack pack;
        pack.a.rows = dim1; pack.a.cols = dim2; pack.b.rows = dim2; pack.b.cols = dim3;
        std::vector<double> va(slots, 0.0), vb(slots, 0.0);
        std::vector<std::vector<double>> rows; // the rows of A, then B: one encodeBatch call (same bits as one encodeVector each)
        for (std::size_t i = 0; i < dim1; ++i) {
            for (std::size_t j = 0; j < dim2; ++j)
                for (std::size_t k = 0; k < dim3; ++k) va[spacers * j + k] = m0[i * dim2 + j];
            rows.push_back(va);
        }
        for (std::size_t j = 0; j < dim2; ++j)
            for (std::size_t k = 0; k < dim3; ++k) vb[spacers * j + k] = m1[j * dim3 + k];
        rows.push_back(vb);
        pack.A = m_p_ctx_wrapper->encodeBatch(rows);
        pack.B = std::move(pack.A.back());
        pack.A.pop_back();
        return this->getEngine().createHandle<decltype(pack)>(sizeof(pack), 0, std::move(pack));
    }
    const std::int64_t *mats[2] = {reinterpret_cast<const std::int64_t *>(mats_raw[0]), reinterpret_cast<const std::int64_t *>(mats_raw[1])};
    const std::size_t slots = m_p_ctx_wrapper->slot_count(), row_size = slots / 2, spacers = row_size / dim2;
    PlainPack pack;
    pack.a.rows = dim1; pack.a.cols = dim2; pack.b.rows = dim2; pack.b.cols = dim3;
    // encodeM0 (.cpp:221-263): A[i][j] replicated dim3 times at slot spacers*j + k; row i in batching row 0, row i+1 in row 1.
    // The cleartext vector is reused across row pairs, exactly as the reference does.
    std::vector<std::int64_t> va(slots, 0);
    std::vector<std::vector<std::int64_t>> rows; // the row pairs of A, then B: one encodeBatch call
    for (std::size_t i = 0; i < dim1; i += 2) {
        for (std::size_t j = 0; j < dim2; ++j)
            for (std::size_t k = 0; k < dim3; ++k) {
                va[spacers * j + k] = mats[0][i * dim2 + j];
                if (i + 1 < dim1) va[row_size + spacers * j + k] = mats[0][(i + 1) * dim2 + j];
            }
        rows.push_back(va);
    }
    // encodeM1 (.cpp:265-297): B[j][k] at spacers*j + k in both batching rows
    std::vector<std::int64_t> vb(slots, 0);
    for (std::size_t j = 0; j < dim2; ++j)
        for (std::size_t k = 0; k < dim3; ++k) {
            vb[spacers * j + k] = mats[1][j * dim3 + k];
            vb[row_size + spacers * j + k] = mats[1][j * dim3 + k];
        }
    rows.push_back(vb);
    pack.A = m_p_ctx_wrapper->encodeBatch(rows);
    pack.B = std::move(pack.A.back());
    pack.A.pop_back();
    return this->getEngine().createHandle<decltype(pack)>(sizeof(pack), 0, std::move(pack));
}

void MatMultRowLatencyBenchmark::decode(AB::Handle h_encoded_data, AB::DataPackCollection *p_native)
{
    if (p_native->pack_count == 0) return;
    if (!p_native->p_data_packs) throw HEBenchError(HEBERROR_MSG_CLASS("Unexpected empty 'p_native->p_data_packs'."), HEBENCH_ECODE_CRITICAL_ERROR);
    const AB::DataPack &rc = findDataPack(*p_native, 0);
    if (rc.buffer_count == 0 || !rc.p_buffers[0].p) return;
    const ResultPlain &enc = this->getEngine().retrieveFromHandle<ResultPlain>(h_encoded_data);
    if (enc.C.empty()) throw HEBenchError(HEBERROR_MSG_CLASS("Unexpected empty handle 'h_encoded_data'."), HEBENCH_ECODE_CRITICAL_ERROR); // bfv row .cpp:314-316
    if (m_scheme == Scheme::CKKS) { // decodeResult, ckks row .cpp:330-356: row i = first dim3 slots of ciphertext i, |x| < 0.00005 -> 0
        double *raw = reinterpret_cast<double *>(rc.p_buffers[0].p);
        std::size_t room = rc.p_buffers[0].size / sizeof(double), pos = 0;
        const auto vals = m_p_ctx_wrapper->decodeSlotsCKKS(enc.C, HeContextWrapper::SlotRanges{{0, enc.d.cols}});
        for (std::size_t i = 0; i < enc.d.rows && i < enc.C.size() && pos < room; ++i) {
            const double *v = vals.data() + i * enc.d.cols;
            for (std::size_t j = 0; j < enc.d.cols && pos < room; ++j) raw[pos++] = std::abs(v[j]) < 0.00005 ? 0.0 : v[j];
        }
        return;
    }
    // decodeResult (.cpp:339-369): result row i = first dim3 slots of batching row (i mod 2) of ciphertext i/2
    const std::size_t dim1 = enc.d.rows, dim3 = enc.d.cols, slots = m_p_ctx_wrapper->slot_count(), row_size = slots / 2;
    std::int64_t *raw = reinterpret_cast<std::int64_t *>(rc.p_buffers[0].p);
    std::size_t room = rc.p_buffers[0].size / sizeof(std::int64_t), pos = 0;
    // the first dim3 slots of both batching rows of every ciphertext: [ciphertext][2][dim3]
    const auto vals = m_p_ctx_wrapper->decodeSlotsBFV(enc.C, HeContextWrapper::SlotRanges{{0, dim3}, {row_size, dim3}});
    (void)slots;
    for (std::size_t i = 0; i < dim1 && pos < room; ++i) {
        if (i / 2 >= enc.C.size()) throw std::out_of_range("MatMultRow decode: result row beyond the decrypted ciphertexts");
        const std::int64_t *v = vals.data() + (i / 2) * 2 * dim3 + ((i & 1) ? dim3 : 0);
        for (std::size_t j = 0; j < dim3 && pos < room; ++j) raw[pos++] = v[j];
    }
}

AB::Handle MatMultRowLatencyBenchmark::encrypt(AB::Handle h_encoded_data)
{
    const PlainPack &p = this->getEngine().retrieveFromHandle<PlainPack>(h_encoded_data);
    CipherPack c;
    c.a = p.a; c.b = p.b;
    std::vector<Plain> all(p.A); // the rows of A, then B: the order the reference encrypts them in
    all.push_back(p.B);
    c.A = m_p_ctx_wrapper->encryptBatch(all);
    c.B = std::move(c.A.back());
    c.A.pop_back();
    return this->getEngine().createHandle<decltype(c)>(sizeof(c), 0, std::move(c));
}

AB::Handle MatMultRowLatencyBenchmark::decrypt(AB::Handle h_encrypted_data)
{
    const ResultCipher &c = this->getEngine().retrieveFromHandle<ResultCipher>(h_encrypted_data);
    ResultPlain p;
    p.d = c.d;
    p.C = m_p_ctx_wrapper->decryptBatch(c.C);
    return this->getEngine().createHandle<decltype(p)>(sizeof(p), 0, std::move(p));
}

AB::Handle MatMultRowLatencyBenchmark::load(const AB::Handle *p_h_local_data, std::uint64_t count)
{
    if (count != 1) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid number of handles. Expected 1."), HEBENCH_ECODE_INVALID_ARGS);
    if (!p_h_local_data) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid null array of handles: \"p_h_local_data\""), HEBENCH_ECODE_INVALID_ARGS);
    const CipherPack &c = this->getEngine().retrieveFromHandle<CipherPack>(p_h_local_data[0]);
    RemotePack r;
    r.a = c.a; r.b = c.b;
    r.A = m_p_ctx_wrapper->upload(c.A);
    r.B = m_p_ctx_wrapper->upload(std::vector<Cipher>{c.B});
    m_p_ctx_wrapper->needRelinKey();
    m_p_ctx_wrapper->needDefaultGaloisKeys(); // the reference creates the full default set (seal_context.cpp:69); rotations use its NAF terms
    if (m_num_devices > 1) { // keys on every device (generated there from the shared seed), A in blocks, B whole: outside the timed call
        if (!m_group) m_group = DeviceGroup::create(m_p_ctx_wrapper, m_num_devices);
        m_group->syncKeys();
        r.A_dev.assign((std::size_t)m_group->size(), nullptr);
        r.B_dev.assign((std::size_t)m_group->size(), nullptr);
        r.A_dev[0] = r.A; r.B_dev[0] = r.B;
        for (int d = 1; d < m_group->size(); ++d) {
            std::uint64_t first = 0, rows = 0;
            DeviceGroup::rowsOf(r.A->n, m_group->size(), d, first, rows);
            r.A_dev[(std::size_t)d] = m_group->replicateRows(d, r.A, first, rows, 0);
            r.B_dev[(std::size_t)d] = m_group->replicateRows(d, r.B, 0, r.B->n, 1);
        }
    }
    return this->getEngine().createHandle<decltype(r)>(sizeof(r), 0, std::move(r));
}

void MatMultRowLatencyBenchmark::store(AB::Handle h_remote_data, AB::Handle *p_h_local_data, std::uint64_t count)
{
    if (count > 0 && !p_h_local_data) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid null array of handles: \"p_h_local_data\""), HEBENCH_ECODE_INVALID_ARGS);
    if (count > 0) {
        std::memset(p_h_local_data, 0, sizeof(AB::Handle) * count);
        const ResultRemote &r = this->getEngine().retrieveFromHandle<ResultRemote>(h_remote_data);
        ResultCipher c;
        c.d = r.d;
        c.C = m_p_ctx_wrapper->download(r.C);
        p_h_local_data[0] = this->getEngine().createHandle<decltype(c)>(sizeof(c), 0, std::move(c));
    }
}

AB::Handle MatMultRowLatencyBenchmark::operate(AB::Handle h_remote_packed, const AB::ParameterIndexer *p_param_indexers, std::uint64_t indexers_count)
{
    if (indexers_count < MatMultRowBenchmarkDescription::NumOpParams) {
        std::stringstream ss;
        ss << "Invalid number of indexers. Expected " << MatMultRowBenchmarkDescription::NumOpParams << ", but " << indexers_count << " received." << std::endl;
        throw HEBenchError(HEBERROR_MSG_CLASS(ss.str()), HEBENCH_ECODE_INVALID_ARGS);
    }
    for (std::size_t i = 0; i < MatMultRowBenchmarkDescription::NumOpParams; ++i) { // this method does not support indexing portions of the batch
        if (p_param_indexers[i].value_index > 0) throw HEBenchError(HEBERROR_MSG_CLASS("Unexpected index in parameter indexer."), HEBENCH_ECODE_INVALID_ARGS);
        if (p_param_indexers[i].batch_size != 1) throw HEBenchError(HEBERROR_MSG_CLASS("Batch size must be 1 for latency test."), HEBENCH_ECODE_INVALID_ARGS);
    }
    const RemotePack &in = this->getEngine().retrieveFromHandle<RemotePack>(h_remote_packed);
    const std::uint64_t nA = in.A->n, dim2 = in.a.cols;
    const bool ckks = m_scheme == Scheme::CKKS;
    ResultRemote res;
    res.d.rows = in.a.rows; res.d.cols = in.b.cols;
    if (in.A_dev.size() > 1 && nA > 1) {
        // the row(-pair) ciphertexts are independent (bfv row .cpp:512-533; the reference nests OpenMP over them, :498-536): contiguous
        // blocks of them per device, one host thread each; device 0 writes into the result slab, the other parts stay where they
        // were computed until store() gathers them
        const double sc = ckks ? in.A->scale * in.B->scale : 1.0;
        res.C = m_p_ctx_wrapper->allocResult(nA, 2, in.A->L, sc);
        std::vector<std::shared_ptr<DeviceCiphers>> parts(in.A_dev.size());
        std::vector<std::uint64_t> firsts(in.A_dev.size(), 0);
        m_group->parallel([&](int d) {
            std::uint64_t first = 0, rows = 0;
            DeviceGroup::rowsOf(nA, m_group->size(), d, first, rows);
            firsts[(std::size_t)d] = first;
            if (!rows) return;
            if (d == 0) { // device 0 holds all of A: its block is a view
                DeviceCiphers view;
                view.d = in.A->d + first * in.A->elems_per_ct(m_p_ctx_wrapper->params().N);
                view.n = rows; view.size = in.A->size; view.L = in.A->L; view.scale = in.A->scale;
                parts[0] = rowsOn(m_group->ctx(0), m_group.get(), 0, view, *in.B, dim2, res.C, first);
                view.d = nullptr; // not owned
            } else {
                parts[(std::size_t)d] = rowsOn(m_group->ctx(d), m_group.get(), d, *in.A_dev[(std::size_t)d], *in.B_dev[(std::size_t)d], dim2, nullptr, 0);
            }
        });
        for (int d = 1; d < m_group->size(); ++d)
            if (parts[(std::size_t)d]) res.C->parts.push_back(DeviceCiphers::Part{parts[(std::size_t)d], firsts[(std::size_t)d]});
    } else {
        res.C = rowsOn(m_p_ctx_wrapper->raw(), nullptr, 0, *in.A, *in.B, dim2, nullptr, 0);
    }
    return this->getEngine().createHandle<decltype(res)>(sizeof(res), 0, std::move(res));
}

// matmultrow (bfv row .cpp:486-539, ckks row .cpp:472-523) for the ciphertexts of A resident on one device, as one batch
std::shared_ptr<DeviceCiphers> MatMultRowLatencyBenchmark::rowsOn(he355_ctx *ctx, DeviceGroup *group, int device, const DeviceCiphers &A, const DeviceCiphers &B,
                                                                  std::uint64_t dim2, std::shared_ptr<DeviceCiphers> into, std::uint64_t into_offset)
{
    const int L = A.L;
    const std::uint64_t nA = A.n, N = m_p_ctx_wrapper->params().N;
    const bool ckks = m_scheme == Scheme::CKKS;
    const int spacers = (int)((ckks ? m_p_ctx_wrapper->slot_count() : m_p_ctx_wrapper->slot_count() / 2) / dim2);
    const double sc = ckks ? A.scale * B.scale : 1.0;
    auto alloc = [&](int size, double scale) { return group ? group->alloc(device, nA, size, L, scale) : m_p_ctx_wrapper->allocResult(nA, size, L, scale); };
    he355_indexer ix{0, 0, 1, 0, 0}; // result i <- (A[i], B)
    std::shared_ptr<DeviceCiphers> base = alloc(2, sc);
    std::shared_ptr<DeviceCiphers> result = into ? into : alloc(2, sc);
    uint64_t *out = into ? into->d + into_offset * 2 * (std::uint64_t)L * N : result->d;
    if (ckks) {
        HeContextWrapper::check(he355_multiply_relin(ctx, L, nA, A.d, B.d, ix, 0, base->d), "multiply+relinearize"); // ckks row .cpp:499-500
    } else {
        std::shared_ptr<DeviceCiphers> c3 = alloc(3, 1.0);
        HeContextWrapper::check(he355_bfv_multiply(ctx, L, nA, A.d, B.d, ix, c3->d), "multiply");                     // :515
        HeContextWrapper::check(he355_relinearize(ctx, L, nA, c3->d, base->d), "relinearize");                       // :516
    }
    // result[i] = base (:519), then result[i] += rotate_rows(base, j * spacers), j = 1 .. dim2-1 (:520-531): all rotations start from
    // `base`, so the ones whose NAF term sequences share a prefix share that prefix's ciphertext (he355_rotate_sum) -- bit-identical
    // to the loop, 127 key switches instead of 313 per ciphertext at dim2 = 128
    std::vector<int32_t> steps;
    for (std::uint64_t j = 1; j < dim2; ++j) steps.push_back((int32_t)((int)j * spacers));
    HeContextWrapper::check(he355_rotate_sum(ctx, L, nA, base->d, steps.data(), steps.size(), out, nullptr), "rotate_rows+add_inplace");
    HeContextWrapper::check(he355_sync(ctx), "synchronise");
    return result;
}
