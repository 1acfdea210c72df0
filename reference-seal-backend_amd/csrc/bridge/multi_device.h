// multi_device.h — one operate() over several MI355X inside ONE process (the API-Bridge harness is a single process:
// /root/reference/README.md:104; the reference spreads the same batch loop over host threads with `NumThreads`,
// include/benchmarks/ckks/seal_ckks_element_wise_benchmark.h:31-42, src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:325).
//
// A DeviceGroup owns one more he355 context per extra device, built from the primary context's parameters, with the evaluation
// keys generated ON that device from the client's secret key and key seed (the generators are counter-based, so every device
// holds bit-identical keys: tests/test_gpu_client.py::test_device_keygen_equals_host_keygen).  load() sends every device
// ITS block of operand-0 rows and all of operand 1 (he355_copy_peer = hipMemcpyPeer over xGMI), operate() cuts the flattened result range [0, b0 * b1) into
// contiguous blocks of operand-0 rows, one per device, each driven by its own host thread, and returns a result whose parts stay
// where they were computed; store() gathers them.  No collective and no copy inside the timed operate().
//
// HE355_LOGICAL_DEVICES=<k> lets a box with fewer GPUs run a k-device group (logical device d -> physical d mod count): the whole
// path — contexts, keys, replicas, threads, parts, gather — is then testable on one GPU (tests/test_api_bridge_gpu.py).
#pragma once
#include <array>
#include <functional>
#include <memory>
#include <vector>

#include "he_context.h"

namespace mi355x {

class DeviceGroup : public std::enable_shared_from_this<DeviceGroup> {
public:
    // n_devices >= 1 (already resolved: workload parameter NumDevices, HE355_NUM_DEVICES, clamped to what is visible)
    static std::shared_ptr<DeviceGroup> create(const HeContextWrapper::Ptr &primary, int n_devices);
    ~DeviceGroup();
    static int resolveCount(std::uint64_t requested); // 0: HE355_NUM_DEVICES or 1; clamped to the (logical) device count
    int size() const { return (int)m_ctx.size(); }
    he355_ctx *ctx(int d) const { return m_ctx[(std::size_t)d]; } // d == 0: the primary's context
    void syncKeys();                                             // every key the primary holds now exists on every device
    std::shared_ptr<DeviceCiphers> replicate(int d, const std::shared_ptr<DeviceCiphers> &src); // d > 0: a copy on device d
    // d > 0: ciphertexts [first, first + count) of src, copied to device d (the rows of operand 0 that device works on: SURVEY.md 8e,
    // "shard operand 0, broadcast operand 1").  operand: 0 / 1, for the load statistics below.
    std::shared_ptr<DeviceCiphers> replicateRows(int d, const std::shared_ptr<DeviceCiphers> &src, std::uint64_t first, std::uint64_t count, int operand);
    // bytes load() moved to device d for operand 0 / 1 since the group was created (tests assert that operand 0 travels in blocks)
    std::uint64_t loadedBytes(int d, int operand) const { return m_loaded[(std::size_t)d][operand & 1]; }
    static std::uint64_t lastLoadedBytes(int d, int operand); // of the group that loaded most recently (he355_bridge_group_load_bytes)
    std::shared_ptr<DeviceCiphers> alloc(int d, std::uint64_t n, int size, int L, double scale);
    // rows [first, first + count) of a b0-row operand for device d (contiguous, balanced to one row: as reference-seal-backend_amd/sharding.py)
    static void rowsOf(std::uint64_t b0, int n_devices, int d, std::uint64_t &first, std::uint64_t &count);
    void parallel(const std::function<void(int)> &fn); // fn(d) on one host thread per device; the first exception is rethrown

private:
    DeviceGroup() = default;
    HeContextWrapper::Ptr m_primary;
    std::vector<he355_ctx *> m_ctx;
    std::vector<bool> m_relin;
    std::vector<std::vector<uint32_t>> m_galois;
    std::vector<std::array<std::uint64_t, 2>> m_loaded;
};

} // namespace mi355x
