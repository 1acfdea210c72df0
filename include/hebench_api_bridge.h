/*
 * hebench_api_bridge.h — clean-room declaration of the HEBench API Bridge C ABI that this backend exports, so
 * that libhebench_mi355x_backend.so can be loaded by HEBench's test_harness in place of
 * libhebench_seal_backend.so (README.md:104 of the reference: --backend_lib_path).
 *
 * The reference does not write these functions itself: it links the C++ wrapper `hebench_cpp` whole-archive
 * (/root/reference/CMakeLists.txt:103-108, cmake/utils/import-library.cmake:98), which exports them and forwards
 * to the virtuals the reference overrides (BaseEngine / BenchmarkDescription / BaseBenchmark).  Neither the
 * api-bridge header (hebench/api_bridge/api.h, types.h, v0.8.0-beta per cmake/third-party/API_BRIDGE.version) nor
 * hebench_cpp is present in this image, so this header is written from the way the reference USES the types
 * (every field below is referenced in /root/reference/src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp or
 * src/engine/seal_engine.cpp, cited per item) plus knowledge of the public api-bridge sources.
 * [UPSTREAM-UNVERIFIED]: enum numeric values, HEBENCH_MAX_* sizes and struct padding are recollections of the public
 * api-bridge v0.8 header (Workload and DataType numbered from 1, EltwiseMultiply before EltwiseAdd), not a copy of it.
 * Drop-in compatibility with a real test_harness therefore stays unverified until the library has been built against
 * the real header, which the build supports directly:
 *     make -C reference-seal-backend_amd/csrc HEBENCH_API_BRIDGE_DIR=<api-bridge checkout or install prefix>
 * then compiles every bridge source against <dir>/include/hebench/api_bridge/api.h instead of this file (the sources use
 * symbolic names only), and csrc/bridge/abi_check.cpp turns any difference in a size, an offset or an enumerator between
 * the two headers into a compile error.  he355_bridge_abi() (same file) exports the numbers the library was built with;
 * tests/test_api_bridge_cpu.py holds the in-repo harness to them.
 */
#ifndef HEBENCH_API_BRIDGE_CLEANROOM_H
#define HEBENCH_API_BRIDGE_CLEANROOM_H
#include <stdint.h>

#ifdef __cplusplus
namespace hebench {
namespace APIBridge {
extern "C" {
#endif

#define HEBENCH_MAX_BUFFER_SIZE 256
#define HEBENCH_MAX_OP_PARAMS 32
#define HEBENCH_MAX_CATEGORY_PARAMS (HEBENCH_MAX_OP_PARAMS * 2)

typedef int32_t ErrorCode;
#define HEBENCH_ECODE_SUCCESS 0
#define HEBENCH_ECODE_INVALID_ARGS 0x7FFFFFFE  /* used: ckks eltwise .cpp:91,131,167 */
#define HEBENCH_ECODE_CRITICAL_ERROR 0x7FFFFFFF /* used: ckks eltwise .cpp:29,50; seal_engine.cpp:53 */

#define HEBENCH_API_VERSION_MAJOR 0 /* compared at seal_engine.cpp:41-43 */
#define HEBENCH_API_VERSION_MINOR 8
#define HEBENCH_API_VERSION_REVISION 0
#define HEBENCH_API_VERSION_BUILD "beta"

/* Handle{p,size,tag}: .tag tested at ckks eltwise .cpp:254 */
typedef struct _FlexibleData {
    void *p;
    uint64_t size;
    int64_t tag;
} _FlexibleData;
typedef _FlexibleData Handle;
typedef _FlexibleData NativeDataBuffer; /* .p ckks eltwise .cpp:194, .size bfv matmultval .cpp:180 */

typedef struct DataPack { /* ckks eltwise .cpp:181,190-192; param_position bfv cipherbatchaxis .cpp:181 */
    NativeDataBuffer *p_buffers;
    uint64_t buffer_count;
    uint64_t param_position;
} DataPack;
typedef struct DataPackCollection { /* ckks eltwise .cpp:165,177 */
    DataPack *p_data_packs;
    uint64_t pack_count;
} DataPackCollection;

typedef struct ParameterIndexer { /* ckks eltwise .cpp:322,334-335 */
    uint64_t value_index;
    uint64_t batch_size;
} ParameterIndexer;

typedef enum Workload { /* seal_engine.cpp:108-151 */
    MatrixMultiply = 1,
    EltwiseMultiply,
    EltwiseAdd,
    DotProduct,
    LogisticRegression,
    LogisticRegression_PolyD3,
    LogisticRegression_PolyD5,
    LogisticRegression_PolyD7,
    Generic
} Workload;
typedef enum DataType { Int32 = 1, Int64, Float32, Float64 } DataType; /* ckks eltwise .cpp:34, bfv eltwise .cpp:34 */
typedef enum Category { Latency = 0, Offline } Category;               /* ckks eltwise .cpp:38,43 */

typedef struct CategoryParams { /* ckks eltwise .cpp:39-45 */
    uint64_t min_test_time_ms;
    union {
        uint64_t reserved[HEBENCH_MAX_CATEGORY_PARAMS];
        struct {
            uint64_t warmup_iterations_count;
        } latency;
        struct {
            uint64_t data_count[HEBENCH_MAX_OP_PARAMS];
        } offline;
    };
} CategoryParams;

typedef int32_t Scheme;
typedef int32_t Security;
#define HEBENCH_HE_SCHEME_PLAIN 0
#define HEBENCH_HE_SCHEME_CKKS 100 /* seal_engine.cpp:101 */
#define HEBENCH_HE_SCHEME_BFV 101  /* seal_engine.cpp:102 */
#define HEBENCH_HE_SCHEME_BGV 102
#define HEBENCH_HE_PARAM_FLAGS_ALL_PLAIN 0x0
#define HEBENCH_HE_PARAM_FLAGS_ALL_CIPHER 0xFFFFFFFF /* ckks eltwise .cpp:52 */

typedef struct BenchmarkDescriptor { /* ckks eltwise .cpp:32-56 */
    Workload workload;
    DataType data_type;
    Category category;
    CategoryParams cat_params;
    uint32_t cipher_param_mask;
    Scheme scheme;
    Security security;
    int64_t other;
} BenchmarkDescriptor;

#ifdef __cplusplus
namespace WorkloadParamType { /* as upstream: the enumerators live in a namespace of the enum's name */
enum WorkloadParamType { Int64 = 0, UInt64, Float64 };
}
typedef WorkloadParamType::WorkloadParamType WorkloadParamTypeT;
#else
typedef enum WorkloadParamType { WP_Int64 = 0, WP_UInt64, WP_Float64 } WorkloadParamTypeT;
#endif
typedef struct WorkloadParam { /* params[i].u_param: ckks eltwise .cpp:93-97 */
    WorkloadParamTypeT data_type;
    char name[HEBENCH_MAX_BUFFER_SIZE];
    union {
        int64_t i_param;
        uint64_t u_param;
        double f_param;
    };
} WorkloadParam;
typedef struct WorkloadParams { /* .count ckks eltwise .cpp:127 */
    WorkloadParam *params;
    uint64_t count;
} WorkloadParams;

/* ---- entry points (what hebench_cpp exports for the reference) ---- */
ErrorCode initEngine(Handle *h_engine, const int8_t *p_buffer, uint64_t size);
ErrorCode destroyHandle(Handle h);
ErrorCode subscribeBenchmarksCount(Handle h_engine, uint64_t *p_count);
ErrorCode subscribeBenchmarks(Handle h_engine, Handle *p_h_bench_descs, uint64_t count);
ErrorCode getWorkloadParamsDetails(Handle h_engine, Handle h_bench_desc, uint64_t *p_param_count, uint64_t *p_default_count);
ErrorCode describeBenchmark(Handle h_engine, Handle h_bench_desc, BenchmarkDescriptor *p_bench_desc, WorkloadParams *p_default_params,
                            uint64_t default_count);
ErrorCode createBenchmark(Handle h_engine, Handle h_bench_desc, const WorkloadParams *p_params, Handle *h_benchmark);
ErrorCode initBenchmark(Handle h_benchmark, const BenchmarkDescriptor *p_concrete_desc);
ErrorCode encode(Handle h_benchmark, const DataPackCollection *p_parameters, Handle *h_plaintext);
ErrorCode decode(Handle h_benchmark, Handle h_plaintext, DataPackCollection *p_native);
ErrorCode encrypt(Handle h_benchmark, Handle h_plaintext, Handle *h_ciphertext);
ErrorCode decrypt(Handle h_benchmark, Handle h_ciphertext, Handle *h_plaintext);
ErrorCode load(Handle h_benchmark, const Handle *h_local_packed_params, uint64_t local_count, Handle *h_remote);
ErrorCode store(Handle h_benchmark, Handle h_remote, Handle *h_local_packed_params, uint64_t local_count);
ErrorCode operate(Handle h_benchmark, Handle h_remote_packed_params, const ParameterIndexer *p_param_indexers, uint64_t indexers_count,
                  Handle *h_remote_output);
uint64_t getSchemeName(Handle h_engine, Scheme s, char *p_name, uint64_t size);
uint64_t getSchemeSecurityName(Handle h_engine, Scheme s, Security sec, char *p_name, uint64_t size);
uint64_t getBenchmarkDescriptionEx(Handle h_engine, Handle h_bench_desc, const WorkloadParams *p_w_params, char *p_description, uint64_t size);
uint64_t getErrorDescription(Handle h_engine, ErrorCode code, char *p_description, uint64_t size);
uint64_t getLastErrorDescription(Handle h_engine, char *p_description, uint64_t size);

#ifdef __cplusplus
} /* extern "C" */
} /* namespace APIBridge */
} /* namespace hebench */
#endif
#endif
