// benchmarks.cpp — see benchmarks.h
#include "benchmarks.h"

#include <cassert>
#include <cmath>
#include <sstream>

using namespace mi355x;
using hebench::cpp::HEBenchError;
namespace AB = hebench::APIBridge;

//------------------------
// class VectorBenchmarkDescription
//------------------------
VectorBenchmarkDescription::VectorBenchmarkDescription(Scheme scheme, AB::Category category, AB::Workload op) : m_scheme(scheme)
{
    const bool ok = op == AB::Workload::EltwiseAdd || op == AB::Workload::EltwiseMultiply || op == AB::Workload::DotProduct;
    if (!ok) throw HEBenchError(HEBERROR_MSG_CLASS("Workload operation not supported."), HEBENCH_ECODE_CRITICAL_ERROR);
    // initialize the descriptor for this benchmark (ckks eltwise .cpp:32-56)
    std::memset(&m_descriptor, 0, sizeof(AB::BenchmarkDescriptor));
    m_descriptor.workload = op;
    m_descriptor.data_type = scheme == Scheme::CKKS ? AB::DataType::Float64 : AB::DataType::Int64;
    m_descriptor.category = category;
    switch (category) {
    case AB::Category::Latency:
        m_descriptor.cat_params.min_test_time_ms = 0; // any
        m_descriptor.cat_params.latency.warmup_iterations_count = 1;
        break;
    case AB::Category::Offline:
        m_descriptor.cat_params.offline.data_count[0] = 0; // flexible
        m_descriptor.cat_params.offline.data_count[1] = 0;
        break;
    default:
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid category received."), HEBENCH_ECODE_CRITICAL_ERROR);
    }
    m_descriptor.cipher_param_mask = HEBENCH_HE_PARAM_FLAGS_ALL_CIPHER;
    m_descriptor.scheme = scheme == Scheme::CKKS ? HEBENCH_HE_SCHEME_CKKS : HEBENCH_HE_SCHEME_BFV;
    m_descriptor.security = HEBENCH_HE_SECURITY_128;
    m_descriptor.other = 0; // no extra parameters

    // defaults: SURVEY.md App. C (ckks eltwise .h:23-29, bfv eltwise .h:23-26, ckks dot .cpp:53-60)
    const bool dot = op == AB::Workload::DotProduct;
    hebench::cpp::WorkloadParams::VectorSize w;
    w.n() = dot ? 100 : 1000;
    w.add<std::uint64_t>(8192, "PolyModulusDegree");
    w.add<std::uint64_t>(2, "MultiplicativeDepth");
    w.add<std::uint64_t>(scheme == Scheme::CKKS ? (dot ? 40 : 45) : (dot ? 45 : 40), "CoefficientModulusBits"); // bfv dot .h:23-26: 45
    if (scheme == Scheme::CKKS) w.add<std::uint64_t>(dot ? 40 : 45, "ScaleBits");
    else w.add<std::uint64_t>(20, "PlainModulusBits");
    w.add<std::uint64_t>(0, "NumThreads"); // kept so existing HEBench YAML configs load; the GPU path ignores it
    // The parameter sets stay exactly the reference's six (an existing HEBench YAML must keep loading: the wrapper rejects a set
    // shorter than the declared one).  The number of GPUs one operate() is spread over comes from HE355_NUM_DEVICES, or from an
    // optional seventh parameter "NumDevices" when a caller supplies one (multi_device.h).
    this->addDefaultParameters(w);
}

hebench::cpp::BaseBenchmark *VectorBenchmarkDescription::createBenchmark(hebench::cpp::BaseEngine &engine, const AB::WorkloadParams *p_params)
{
    return new VectorBenchmark(engine, m_descriptor, *p_params, m_scheme);
}
void VectorBenchmarkDescription::destroyBenchmark(hebench::cpp::BaseBenchmark *p_bench)
{
    if (p_bench) delete p_bench;
}

std::string VectorBenchmarkDescription::getBenchmarkDescription(const AB::WorkloadParams *p_w_params) const
{
    std::stringstream ss;
    std::string s_tmp = BenchmarkDescription::getBenchmarkDescription(p_w_params);
    if (!p_w_params)
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid null workload parameters `p_w_params`"), HEBENCH_ECODE_INVALID_ARGS);
    const std::uint64_t poly_modulus_degree = p_w_params->params[Index_PolyModulusDegree].u_param;
    const std::uint64_t multiplicative_depth = p_w_params->params[Index_NumCoefficientModuli].u_param;
    const std::uint64_t coeff_modulus_bits = p_w_params->params[Index_CoefficientModulusBits].u_param;
    const std::uint64_t extra_bits = p_w_params->params[Index_ScaleExponentBits].u_param;
    if (!s_tmp.empty()) ss << s_tmp << std::endl;
    ss << ", Encryption Parameters" << std::endl
       << ", , Poly modulus degree, " << poly_modulus_degree << std::endl
       << ", , Coefficient Modulus, 60";
    for (std::size_t i = 1; i < multiplicative_depth; ++i) ss << ", " << coeff_modulus_bits;
    ss << ", 60" << std::endl;
    if (m_scheme == Scheme::CKKS) ss << ", , Scale, 2^" << extra_bits << std::endl;
    else ss << ", , Plain Text Modulus Bits, " << extra_bits << std::endl; // bfv eltwise .cpp:111, bfv dot .cpp:105
    ss << ", Algorithm, " << AlgorithmName << ", " << AlgorithmDescription << std::endl
       << HeContextWrapper::threadsRow(p_w_params->params[Index_NumThreads].u_param, m_descriptor.category == AB::Category::Latency) << std::endl
       << ", Device, AMD Instinct MI355X (HIP; batch mapped to the grid, no host threads)";
    const int n_dev = DeviceGroup::resolveCount(p_w_params->count > Index_NumDevices ? p_w_params->params[Index_NumDevices].u_param : 0);
    if (n_dev > 1 || p_w_params->count > Index_NumDevices) ss << std::endl << ", Number of devices, " << n_dev;
    return ss.str();
}

//------------------------
// class VectorBenchmark
//------------------------
VectorBenchmark::VectorBenchmark(hebench::cpp::BaseEngine &engine, const AB::BenchmarkDescriptor &bench_desc, const AB::WorkloadParams &bench_params,
                                 Scheme scheme)
    : hebench::cpp::BaseBenchmark(engine, bench_desc, bench_params), m_scheme(scheme), m_w_params(bench_params)
{
    if (bench_params.count < VectorBenchmarkDescription::NumWorkloadParams)
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid workload parameters."), HEBENCH_ECODE_INVALID_ARGS);
    if (m_w_params.n() <= 0)
        throw HEBenchError(HEBERROR_MSG_CLASS("Vector size must be greater than 0."), HEBENCH_ECODE_INVALID_ARGS);
    const std::uint64_t poly_modulus_degree = m_w_params.get<std::uint64_t>(VectorBenchmarkDescription::Index_PolyModulusDegree);
    const std::uint64_t multiplicative_depth = m_w_params.get<std::uint64_t>(VectorBenchmarkDescription::Index_NumCoefficientModuli);
    const std::uint64_t coeff_modulus_bits = m_w_params.get<std::uint64_t>(VectorBenchmarkDescription::Index_CoefficientModulusBits);
    const std::uint64_t extra_bits = m_w_params.get<std::uint64_t>(VectorBenchmarkDescription::Index_ScaleExponentBits);
    if (coeff_modulus_bits < 1)
        throw HEBenchError(HEBERROR_MSG_CLASS("Multiplicative depth must be greater than 0."), HEBENCH_ECODE_INVALID_ARGS);
    if (m_scheme == Scheme::CKKS)
        m_p_ctx_wrapper = HeContextWrapper::createCKKSContext(poly_modulus_degree, multiplicative_depth, (int)coeff_modulus_bits, (int)extra_bits);
    else
        m_p_ctx_wrapper = HeContextWrapper::createBFVContext(poly_modulus_degree, multiplicative_depth, (int)coeff_modulus_bits, (int)extra_bits);
    // (decode() reads n slots of every result, one for a dot product: the warm-up and the staging buffer are sized for that)
    m_p_ctx_wrapper->prepareClient(256, bench_desc.workload == AB::Workload::DotProduct ? 1 : m_w_params.n());
    const std::size_t slot_count = m_p_ctx_wrapper->slot_count();
    if (m_w_params.n() > slot_count)
        throw HEBenchError(HEBERROR_MSG_CLASS("Vector size cannot be greater than " + std::to_string(slot_count) + "."), HEBENCH_ECODE_INVALID_ARGS);
    if (bench_params.count > VectorBenchmarkDescription::Index_NumDevices)
        m_num_devices = DeviceGroup::resolveCount(m_w_params.get<std::uint64_t>(VectorBenchmarkDescription::Index_NumDevices));
    else
        m_num_devices = DeviceGroup::resolveCount(0);
}

AB::Handle VectorBenchmark::encode(const AB::DataPackCollection *p_parameters)
{
    if (p_parameters->pack_count != VectorBenchmarkDescription::NumOpParams)
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid number of parameters detected in parameter pack. Expected 2."), HEBENCH_ECODE_INVALID_ARGS);
    std::vector<std::vector<Plain>> params(p_parameters->pack_count);
    for (std::size_t x = 0; x < params.size(); ++x) {
        const AB::DataPack &parameter = p_parameters->p_data_packs[x];
        std::vector<std::vector<double>> rows_d;
        std::vector<std::vector<std::int64_t>> rows_i;
        for (std::size_t y = 0; y < parameter.buffer_count; ++y) {
            const AB::NativeDataBuffer &sample = parameter.p_buffers[y];
            if (!sample.p || sample.size < m_w_params.n() * 8)
                throw HEBenchError(HEBERROR_MSG_CLASS("Invalid sample buffer."), HEBENCH_ECODE_INVALID_ARGS);
            if (m_scheme == Scheme::CKKS) {
                const double *p_row = reinterpret_cast<const double *>(sample.p);
                rows_d.emplace_back(p_row, p_row + m_w_params.n());
            } else {
                const std::int64_t *p_row = reinterpret_cast<const std::int64_t *>(sample.p);
                rows_i.emplace_back(p_row, p_row + m_w_params.n());
            }
        }
        params[x] = m_scheme == Scheme::CKKS ? m_p_ctx_wrapper->encodeBatch(rows_d) : m_p_ctx_wrapper->encodeBatch(rows_i); // one plaintext per sample
    }
    return this->getEngine().createHandle<decltype(params)>(sizeof(params), 0, std::move(params));
}

void VectorBenchmark::decode(AB::Handle encoded_data, AB::DataPackCollection *p_native)
{
    const std::vector<Plain> &params = this->getEngine().retrieveFromHandle<std::vector<Plain>>(encoded_data);
    if (p_native->pack_count < 1) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid output data pack."), HEBENCH_ECODE_INVALID_ARGS);
    const bool dot = this->getDescriptor().workload == AB::Workload::DotProduct;
    const std::size_t out_n = dot ? 1 : m_w_params.n();
    const std::size_t n_res = std::min<std::size_t>(params.size(), p_native->p_data_packs[0].buffer_count);
    const std::vector<Plain> wanted(params.begin(), params.begin() + n_res);
    // only the out_n slots the loop below copies are decoded to the host (the reference decodes all and copies the first n: .cpp:214-226)
    const HeContextWrapper::SlotRanges head{{0, out_n}};
    if (m_scheme == Scheme::CKKS) {
        const auto vals = m_p_ctx_wrapper->decodeSlotsCKKS(wanted, head);
        for (std::size_t result_i = 0; result_i < n_res; ++result_i) {
            double *output_location = reinterpret_cast<double *>(p_native->p_data_packs[0].p_buffers[result_i].p);
            const double *v = vals.data() + result_i * out_n;
            for (std::size_t x = 0; x < out_n; ++x) // same clamp as ckks eltwise .cpp:222-225
                output_location[x] = std::abs(v[x]) < 0.00005 ? 0 : v[x];
        }
    } else {
        const auto vals = m_p_ctx_wrapper->decodeSlotsBFV(wanted, head);
        for (std::size_t result_i = 0; result_i < n_res; ++result_i) {
            std::int64_t *output_location = reinterpret_cast<std::int64_t *>(p_native->p_data_packs[0].p_buffers[result_i].p);
            std::copy(vals.begin() + result_i * out_n, vals.begin() + (result_i + 1) * out_n, output_location);
        }
    }
}

AB::Handle VectorBenchmark::encrypt(AB::Handle encoded_data)
{
    const std::vector<std::vector<Plain>> &encoded = this->getEngine().retrieveFromHandle<std::vector<std::vector<Plain>>>(encoded_data);
    std::vector<std::vector<Cipher>> encrypted(encoded.size());
    for (std::size_t param_i = 0; param_i < encoded.size(); ++param_i) encrypted[param_i] = m_p_ctx_wrapper->encryptBatch(encoded[param_i]);
    return this->getEngine().createHandle<decltype(encrypted)>(sizeof(encrypted), 0, std::move(encrypted));
}

AB::Handle VectorBenchmark::decrypt(AB::Handle encrypted_data)
{
    if ((encrypted_data.tag & hebench::cpp::EngineObject::tag) == 0)
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid tag detected. Expected EngineObject::tag."), HEBENCH_ECODE_INVALID_ARGS);
    const std::vector<Cipher> &encrypted = this->getEngine().retrieveFromHandle<std::vector<Cipher>>(encrypted_data);
    std::vector<Plain> plaintext_data = m_p_ctx_wrapper->decryptBatch(encrypted); // size-3 results included
    return this->getEngine().createHandle<decltype(plaintext_data)>(sizeof(plaintext_data), 0, std::move(plaintext_data));
}

// load(): the host -> HBM boundary.  The reference only duplicates the handle (ckks eltwise .cpp:277-289) because
// its "remote" is the host; here each operand becomes one device slab of its samples, and the evaluation keys the
// workload needs are generated and uploaded.
AB::Handle VectorBenchmark::load(const AB::Handle *p_local_data, std::uint64_t count)
{
    if (count != 1)
        // we do all ops in ciphertext, so, we should get only one pack of data
        throw HEBenchError(HEBERROR_MSG_CLASS("Invalid number of handles. Expected 1."), HEBENCH_ECODE_INVALID_ARGS);
    assert(p_local_data);
    const std::vector<std::vector<Cipher>> &local = this->getEngine().retrieveFromHandle<std::vector<std::vector<Cipher>>>(p_local_data[0]);
    RemotePack remote;
    for (const auto &operand : local) remote.ops.push_back(m_p_ctx_wrapper->upload(operand));
    if (this->getDescriptor().workload == AB::Workload::DotProduct) {
        m_p_ctx_wrapper->needRelinKey();
        // accumulateCKKS(n) / accumulateBFV(n): rotations by 2^i, i < bit_count(count) (seal_context.cpp:331-339, 295-304);
        // BFV counts within one batching row and adds the column swap when n exceeds a row (:305-310)
        const std::uint64_t row = m_scheme == Scheme::CKKS ? m_p_ctx_wrapper->slot_count() : m_p_ctx_wrapper->slot_count() / 2;
        std::uint64_t cnt = std::min<std::uint64_t>(m_w_params.n(), row);
        int rotations = 64 - __builtin_clzll(cnt);
        if (((std::uint64_t)1 << (rotations - 1)) == cnt) --rotations;
        for (int i = 0; i < rotations; ++i) m_p_ctx_wrapper->needRotationKey(1 << i);
        if (m_scheme == Scheme::BFV && m_w_params.n() > row) m_p_ctx_wrapper->needRotationKey(0);
    }
    if (m_num_devices > 1) {
        // the batch loop of operate() is spread over the group's devices: every device gets the evaluation keys (generated there from
        // the shared seed), ITS block of operand-0 rows and all of operand 1 (hipMemcpyPeer over xGMI), here, outside the timed call
        // (shard operand 0, broadcast operand 1)
        if (!m_group) m_group = DeviceGroup::create(m_p_ctx_wrapper, m_num_devices);
        m_group->syncKeys();
        remote.replicas.resize((std::size_t)m_group->size());
        remote.replicas[0] = remote.ops;
        for (int d = 1; d < m_group->size() && remote.ops.size() >= 2; ++d) {
            std::uint64_t first = 0, rows = 0;
            DeviceGroup::rowsOf(remote.ops[0]->n, m_group->size(), d, first, rows);
            remote.replicas[(std::size_t)d].push_back(m_group->replicateRows(d, remote.ops[0], first, rows, 0));
            remote.replicas[(std::size_t)d].push_back(m_group->replicateRows(d, remote.ops[1], 0, remote.ops[1]->n, 1));
        }
    }
    return this->getEngine().createHandle<decltype(remote)>(sizeof(remote), 0, std::move(remote));
}

// store(): HBM -> host
void VectorBenchmark::store(AB::Handle remote_data, AB::Handle *p_local_data, std::uint64_t count)
{
    assert(count == 0 || p_local_data);
    if (count > 0) {
        // pad with zeros any excess local handles as per specifications
        std::memset(p_local_data, 0, sizeof(AB::Handle) * count);
        const std::shared_ptr<DeviceCiphers> &remote = this->getEngine().retrieveFromHandle<std::shared_ptr<DeviceCiphers>>(remote_data);
        std::vector<Cipher> local = m_p_ctx_wrapper->download(remote);
        p_local_data[0] = this->getEngine().createHandle<decltype(local)>(sizeof(local), 0, std::move(local));
    }
}

// operate(): the timed hot path.  Result r = i * batch1 + x  <-  op(operand0[value_index0 + i], operand1[value_index1 + x])
// (ckks eltwise .cpp:322-336).  One kernel sequence over the whole batch instead of an OpenMP loop.
AB::Handle VectorBenchmark::operate(AB::Handle h_remote_packed, const AB::ParameterIndexer *p_param_indexers, std::uint64_t indexers_count)
{
    if (indexers_count < VectorBenchmarkDescription::NumOpParams) {
        std::stringstream ss;
        ss << "Invalid number of indexers. Expected " << VectorBenchmarkDescription::NumOpParams << ", but " << indexers_count << " received." << std::endl;
        throw HEBenchError(HEBERROR_MSG_CLASS(ss.str()), HEBENCH_ECODE_INVALID_ARGS);
    }
    const RemotePack &pack = this->getEngine().retrieveFromHandle<RemotePack>(h_remote_packed);
    const std::vector<std::shared_ptr<DeviceCiphers>> &params = pack.ops;
    if (params.size() < 2) throw HEBenchError(HEBERROR_MSG_CLASS("Invalid remote operand pack."), HEBENCH_ECODE_INVALID_ARGS);
    const DeviceCiphers &p0 = *params[0], &p1 = *params[1];
    const std::uint64_t b0 = p_param_indexers[0].batch_size, b1 = p_param_indexers[1].batch_size;
    if (p_param_indexers[0].value_index + b0 > p0.n || p_param_indexers[1].value_index + b1 > p1.n)
        throw HEBenchError(HEBERROR_MSG_CLASS("Parameter indexer out of range."), HEBENCH_ECODE_INVALID_ARGS);
    const std::uint64_t n = b0 * b1;
    he355_indexer ix;
    ix.a_base = p_param_indexers[0].value_index;
    ix.b_base = p_param_indexers[1].value_index;
    ix.b1 = b1 ? b1 : 1;
    ix.pairwise = 0;
    ix.reserved = 0;
    std::shared_ptr<DeviceCiphers> result;
    // the devices hold operand 0 in the blocks of the whole-batch call (value_index 0, every sample: what the harness's offline and
    // latency categories issue); any other indexing runs on the primary device, which holds everything
    if (pack.replicas.size() > 1 && b0 > 1 && p_param_indexers[0].value_index == 0 && b0 == p0.n) {
        // Contiguous blocks of operand-0 rows, one per device, one host thread each (the reference spreads the same loop over
        // NumThreads OpenMP threads, ckks eltwise .cpp:325).  Device 0 writes straight into the result slab; the other devices'
        // blocks stay where they were computed until store() gathers them: no copy and no collective inside the timed call.
        const int out_size = this->getDescriptor().workload == AB::Workload::EltwiseMultiply ? 3 : 2;
        const double out_scale = this->getDescriptor().workload == AB::Workload::EltwiseAdd ? p0.scale : p0.scale * p1.scale;
        result = m_p_ctx_wrapper->allocResult(n, out_size, p0.L, out_scale);
        std::vector<std::shared_ptr<DeviceCiphers>> parts((std::size_t)m_group->size());
        std::vector<std::uint64_t> firsts((std::size_t)m_group->size(), 0);
        m_group->parallel([&](int d) {
            std::uint64_t first = 0, rows = 0;
            DeviceGroup::rowsOf(b0, m_group->size(), d, first, rows);
            firsts[(std::size_t)d] = first * ix.b1;
            if (!rows) return;
            he355_indexer ixd = ix;
            ixd.a_base = d == 0 ? first : 0; // device d > 0 holds exactly its block
            const auto &ops = pack.replicas[(std::size_t)d];
            parts[(std::size_t)d] = operateOn(m_group->ctx(d), m_group.get(), d, *ops[0], *ops[1], ixd, rows * ix.b1, d == 0 ? result : nullptr, first * ix.b1);
        });
        for (int d = 1; d < m_group->size(); ++d)
            if (parts[(std::size_t)d]) result->parts.push_back(DeviceCiphers::Part{parts[(std::size_t)d], firsts[(std::size_t)d]});
    } else {
        result = operateOn(m_p_ctx_wrapper->raw(), nullptr, 0, p0, p1, ix, n, nullptr, 0);
    }
    return this->getEngine().createHandle<decltype(result)>(sizeof(result), 0, std::move(result));
}

// the per-device body of operate(): n results from operand slabs p0, p1 resident on `ctx`'s device.  `into` (device 0 of a group):
// write at result offset `into_offset` of that slab instead of allocating one.
std::shared_ptr<DeviceCiphers> VectorBenchmark::operateOn(he355_ctx *ctx, DeviceGroup *group, int device, const DeviceCiphers &p0, const DeviceCiphers &p1,
                                                          he355_indexer ix, std::uint64_t n, std::shared_ptr<DeviceCiphers> into, std::uint64_t into_offset)
{
    const int L = p0.L;
    auto alloc = [&](std::uint64_t cnt, int size, double scale) {
        return group ? group->alloc(device, cnt, size, L, scale) : m_p_ctx_wrapper->allocResult(cnt, size, L, scale);
    };
    std::shared_ptr<DeviceCiphers> result = into;
    uint64_t *out = nullptr;
    auto target = [&](int size, double scale) {
        if (!result) result = alloc(n, size, scale);
        out = into ? into->d + into_offset * (std::uint64_t)size * L * m_p_ctx_wrapper->params().N : result->d;
    };
    switch (this->getDescriptor().workload) {
    case AB::Workload::EltwiseAdd:
        target(2, p0.scale);
        HeContextWrapper::check(he355_add(ctx, L, 2, n, p0.d, p1.d, ix, out), "add");
        break;
    case AB::Workload::EltwiseMultiply: // multiply only: the size-3 result is decrypted as is (ckks eltwise .cpp:342-344, bfv :324-326)
        target(3, p0.scale * p1.scale);
        if (m_scheme == Scheme::CKKS) HeContextWrapper::check(he355_multiply(ctx, L, n, p0.d, p1.d, ix, out), "multiply");
        else HeContextWrapper::check(he355_bfv_multiply(ctx, L, n, p0.d, p1.d, ix, out), "multiply");
        break;
    case AB::Workload::DotProduct: { // multiply -> relinearize_inplace -> accumulate(n)  (ckks dot .cpp:325-330, bfv dot .cpp:311-315)
        target(2, p0.scale * p1.scale);
        if (m_scheme == Scheme::CKKS) {
            HeContextWrapper::check(he355_multiply_relin(ctx, L, n, p0.d, p1.d, ix, 0, out), "multiply+relinearize");
        } else {
            std::shared_ptr<DeviceCiphers> c3 = alloc(n, 3, 1.0);
            HeContextWrapper::check(he355_bfv_multiply(ctx, L, n, p0.d, p1.d, ix, c3->d), "multiply");
            HeContextWrapper::check(he355_relinearize(ctx, L, n, c3->d, out), "relinearize");
            // c3 goes back to the pool here: whatever reuses it is queued behind the relinearization on the same stream
        }
        std::shared_ptr<DeviceCiphers> tmp = alloc(n, 2, result->scale);
        HeContextWrapper::check(he355_accumulate(ctx, L, n, out, m_w_params.n(), tmp->d), "accumulate");
        break;
    }
    default:
        // the reference's two element-wise classes word this differently (bfv eltwise .cpp:329, ckks eltwise .cpp:346)
        throw HEBenchError(HEBERROR_MSG_CLASS(m_scheme == Scheme::BFV ? "Operation not implimented." : "Operation not supported."), HEBENCH_ECODE_INVALID_ARGS);
    }
    HeContextWrapper::check(he355_sync(ctx), "synchronise"); // operate() returns with the result complete, as the reference's does
    return result;
}
