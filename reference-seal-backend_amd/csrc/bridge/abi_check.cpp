// abi_check.cpp — the API-Bridge ABI this library was compiled against, pinned at compile time and exported at run time.
//
// The bridge sources use symbolic names only, so they compile against include/hebench_api_bridge.h (clean-room,
// [UPSTREAM-UNVERIFIED]) or, with `make HEBENCH_API_BRIDGE_DIR=...`, against the real hebench/api_bridge/api.h.  The
// static_asserts below state every size, offset and enumerator the clean-room header, the in-repo harness
// (tests/hebench_harness.py) and INTEGRATION.md assume: built against the real header, any difference is a compile error
// here (first contact with upstream is loud, not a silently shifted workload id).  -DHEBENCH_ABI_CHECK_RELAXED turns the
// asserts off for a deliberate build against a header that differs (the exported table below then tells a harness what the
// library really uses).
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>

#include "hebench_cpp.h"

namespace ab = hebench::APIBridge;

#if !defined(HEBENCH_ABI_CHECK_RELAXED)
#define ABI_ASSERT(cond) static_assert(cond, "API-Bridge ABI differs from what this backend and its harness assume: " #cond)
#else
#define ABI_ASSERT(cond) static_assert(true, "")
#endif

// limits
ABI_ASSERT(HEBENCH_MAX_BUFFER_SIZE == 256);
ABI_ASSERT(HEBENCH_MAX_OP_PARAMS == 32);
ABI_ASSERT(HEBENCH_MAX_CATEGORY_PARAMS == 64);
// error codes
ABI_ASSERT(HEBENCH_ECODE_SUCCESS == 0);
ABI_ASSERT(HEBENCH_ECODE_INVALID_ARGS == 0x7FFFFFFE);
ABI_ASSERT(HEBENCH_ECODE_CRITICAL_ERROR == 0x7FFFFFFF);
ABI_ASSERT(sizeof(ab::ErrorCode) == 4);
// handles and data packs
ABI_ASSERT(sizeof(ab::Handle) == 24);
ABI_ASSERT(offsetof(ab::Handle, p) == 0);
ABI_ASSERT(offsetof(ab::Handle, size) == 8);
ABI_ASSERT(offsetof(ab::Handle, tag) == 16);
ABI_ASSERT(sizeof(ab::NativeDataBuffer) == 24);
ABI_ASSERT(sizeof(ab::DataPack) == 24);
ABI_ASSERT(offsetof(ab::DataPack, p_buffers) == 0);
ABI_ASSERT(offsetof(ab::DataPack, buffer_count) == 8);
ABI_ASSERT(offsetof(ab::DataPack, param_position) == 16);
ABI_ASSERT(sizeof(ab::DataPackCollection) == 16);
ABI_ASSERT(offsetof(ab::DataPackCollection, p_data_packs) == 0);
ABI_ASSERT(offsetof(ab::DataPackCollection, pack_count) == 8);
ABI_ASSERT(sizeof(ab::ParameterIndexer) == 16);
ABI_ASSERT(offsetof(ab::ParameterIndexer, value_index) == 0);
ABI_ASSERT(offsetof(ab::ParameterIndexer, batch_size) == 8);
// enumerators
ABI_ASSERT(sizeof(ab::Workload) == 4 && sizeof(ab::DataType) == 4 && sizeof(ab::Category) == 4);
ABI_ASSERT(ab::MatrixMultiply == 1);
ABI_ASSERT(ab::EltwiseMultiply == 2);
ABI_ASSERT(ab::EltwiseAdd == 3);
ABI_ASSERT(ab::DotProduct == 4);
ABI_ASSERT(ab::LogisticRegression == 5);
ABI_ASSERT(ab::LogisticRegression_PolyD3 == 6);
ABI_ASSERT(ab::Int32 == 1 && ab::Int64 == 2 && ab::Float32 == 3 && ab::Float64 == 4);
ABI_ASSERT(ab::Latency == 0 && ab::Offline == 1);
ABI_ASSERT(ab::WorkloadParamType::Int64 == 0 && ab::WorkloadParamType::UInt64 == 1 && ab::WorkloadParamType::Float64 == 2);
ABI_ASSERT(HEBENCH_HE_SCHEME_CKKS == 100 && HEBENCH_HE_SCHEME_BFV == 101);
ABI_ASSERT(HEBENCH_HE_PARAM_FLAGS_ALL_CIPHER == 0xFFFFFFFF);
// descriptor
ABI_ASSERT(sizeof(ab::CategoryParams) == 8 + 8 * 64);
ABI_ASSERT(offsetof(ab::CategoryParams, min_test_time_ms) == 0);
ABI_ASSERT(offsetof(ab::CategoryParams, latency) == 8);
ABI_ASSERT(offsetof(ab::CategoryParams, offline) == 8);
ABI_ASSERT(sizeof(ab::BenchmarkDescriptor) == 560);
ABI_ASSERT(offsetof(ab::BenchmarkDescriptor, workload) == 0);
ABI_ASSERT(offsetof(ab::BenchmarkDescriptor, data_type) == 4);
ABI_ASSERT(offsetof(ab::BenchmarkDescriptor, category) == 8);
ABI_ASSERT(offsetof(ab::BenchmarkDescriptor, cat_params) == 16);
ABI_ASSERT(offsetof(ab::BenchmarkDescriptor, cipher_param_mask) == 536);
ABI_ASSERT(offsetof(ab::BenchmarkDescriptor, scheme) == 540);
ABI_ASSERT(offsetof(ab::BenchmarkDescriptor, security) == 544);
ABI_ASSERT(offsetof(ab::BenchmarkDescriptor, other) == 552);
// workload parameters
ABI_ASSERT(sizeof(ab::WorkloadParam) == 272);
ABI_ASSERT(offsetof(ab::WorkloadParam, data_type) == 0);
ABI_ASSERT(offsetof(ab::WorkloadParam, name) == 4);
ABI_ASSERT(offsetof(ab::WorkloadParam, u_param) == 264);
ABI_ASSERT(sizeof(ab::WorkloadParams) == 16);
ABI_ASSERT(offsetof(ab::WorkloadParams, params) == 0);
ABI_ASSERT(offsetof(ab::WorkloadParams, count) == 8);

// The same numbers as the library was really built with, for harnesses (JSON; returns the size needed incl. the terminator).
extern "C" std::uint64_t he355_bridge_abi(char *p_buffer, std::uint64_t size)
{
    char tmp[2048];
    std::snprintf(tmp, sizeof(tmp),
                  "{\"header\": \"%s\", \"api_version\": [%d, %d, %d], \"max_buffer_size\": %d, \"max_op_params\": %d, \"max_category_params\": %d, "
                  "\"ecode_invalid_args\": %lld, \"ecode_critical_error\": %lld, "
                  "\"workload\": {\"MatrixMultiply\": %d, \"EltwiseMultiply\": %d, \"EltwiseAdd\": %d, \"DotProduct\": %d, \"LogisticRegression\": %d, "
                  "\"LogisticRegression_PolyD3\": %d}, "
                  "\"data_type\": {\"Int32\": %d, \"Int64\": %d, \"Float32\": %d, \"Float64\": %d}, \"category\": {\"Latency\": %d, \"Offline\": %d}, "
                  "\"workload_param_type\": {\"Int64\": %d, \"UInt64\": %d, \"Float64\": %d}, \"scheme\": {\"CKKS\": %d, \"BFV\": %d}, "
                  "\"sizeof\": {\"Handle\": %zu, \"DataPack\": %zu, \"DataPackCollection\": %zu, \"ParameterIndexer\": %zu, \"CategoryParams\": %zu, "
                  "\"BenchmarkDescriptor\": %zu, \"WorkloadParam\": %zu, \"WorkloadParams\": %zu}, "
                  "\"offsetof\": {\"BenchmarkDescriptor.cat_params\": %zu, \"BenchmarkDescriptor.cipher_param_mask\": %zu, \"BenchmarkDescriptor.scheme\": %zu, "
                  "\"BenchmarkDescriptor.security\": %zu, \"BenchmarkDescriptor.other\": %zu, \"WorkloadParam.name\": %zu, \"WorkloadParam.u_param\": %zu}}",
#if defined(HEBENCH_REAL_API_HEADERS)
                  "hebench/api_bridge/api.h",
#else
                  "include/hebench_api_bridge.h (clean-room, UPSTREAM-UNVERIFIED)",
#endif
                  (int)HEBENCH_API_VERSION_MAJOR, (int)HEBENCH_API_VERSION_MINOR, (int)HEBENCH_API_VERSION_REVISION, (int)HEBENCH_MAX_BUFFER_SIZE,
                  (int)HEBENCH_MAX_OP_PARAMS, (int)HEBENCH_MAX_CATEGORY_PARAMS, (long long)HEBENCH_ECODE_INVALID_ARGS, (long long)HEBENCH_ECODE_CRITICAL_ERROR,
                  (int)ab::MatrixMultiply, (int)ab::EltwiseMultiply, (int)ab::EltwiseAdd, (int)ab::DotProduct, (int)ab::LogisticRegression,
                  (int)ab::LogisticRegression_PolyD3, (int)ab::Int32, (int)ab::Int64, (int)ab::Float32, (int)ab::Float64, (int)ab::Latency, (int)ab::Offline,
                  (int)ab::WorkloadParamType::Int64, (int)ab::WorkloadParamType::UInt64, (int)ab::WorkloadParamType::Float64, (int)HEBENCH_HE_SCHEME_CKKS,
                  (int)HEBENCH_HE_SCHEME_BFV, sizeof(ab::Handle), sizeof(ab::DataPack), sizeof(ab::DataPackCollection), sizeof(ab::ParameterIndexer),
                  sizeof(ab::CategoryParams), sizeof(ab::BenchmarkDescriptor), sizeof(ab::WorkloadParam), sizeof(ab::WorkloadParams),
                  offsetof(ab::BenchmarkDescriptor, cat_params), offsetof(ab::BenchmarkDescriptor, cipher_param_mask), offsetof(ab::BenchmarkDescriptor, scheme),
                  offsetof(ab::BenchmarkDescriptor, security), offsetof(ab::BenchmarkDescriptor, other), offsetof(ab::WorkloadParam, name),
                  offsetof(ab::WorkloadParam, u_param));
    const std::uint64_t need = std::strlen(tmp) + 1;
    if (p_buffer && size) {
        std::strncpy(p_buffer, tmp, size);
        p_buffer[size - 1] = 0;
    }
    return need;
}
