// capi_internal.hpp -- handle / workspace structures shared by the C-ABI translation units (capi.hip, gfa.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gbwt_hip.h"
#include "device_index.hpp"
#include "host_index.hpp"
#include "kernels.hpp"

namespace gbwt_hip {

std::string &last_error_slot();   // thread-local message of the last failing call

inline gbwt_hip_status fail(gbwt_hip_status st, const std::string &msg) {
    last_error_slot() = msg;
    return st;
}

struct HipError { hipError_t err; const char *what; };

#define HIP_CHECK(expr)                                            \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) throw gbwt_hip::HipError{e_, #expr}; \
    } while (0)

// Grow-only device buffer.
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    ~DeviceBuffer() { if (ptr) (void)hipFree(ptr); }
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    void reserve(size_t need) {
        if (need <= bytes) return;
        if (ptr) { HIP_CHECK(hipFree(ptr)); ptr = nullptr; bytes = 0; }
        size_t want = std::max<size_t>(need, 256);
        HIP_CHECK(hipMalloc(&ptr, want));
        bytes = want;
    }
    template <class T> T *as() const { return static_cast<T *>(ptr); }
};

inline gbwt_hip_status status_of(const HipError &e) {
    std::string msg = std::string(e.what) + ": " + hipGetErrorString(e.err);
    if (e.err == hipErrorNoDevice || e.err == hipErrorInvalidDevice) return fail(GBWT_HIP_NO_DEVICE, msg);
    return fail(GBWT_HIP_DEVICE_ERROR, msg);
}

// Lippincott function: the status of the exception in flight.  No exception may cross the C ABI (the caller is Rust / C /
// ctypes: it would be std::terminate), so every extern "C" entry point that can throw wraps its body in
// GBWT_HIP_GUARD_BEGIN / GBWT_HIP_GUARD_END, whatever it already catches inside.
inline gbwt_hip_status status_of_current_exception() noexcept {
    try {
        throw;
    } catch (const InvalidData &e) {
        return fail(GBWT_HIP_INVALID_DATA, e.what());
    } catch (const IoError &e) {
        return fail(GBWT_HIP_IO_ERROR, e.what());
    } catch (const HipError &e) {
        return status_of(e);
    } catch (const std::bad_alloc &) {
        return fail(GBWT_HIP_DEVICE_ERROR, "out of host memory");
    } catch (const std::length_error &e) {   // a size taken from a file that no container can hold
        return fail(GBWT_HIP_INVALID_DATA, std::string("length error: ") + e.what());
    } catch (const std::exception &e) {
        return fail(GBWT_HIP_DEVICE_ERROR, std::string("unexpected exception: ") + e.what());
    } catch (...) {
        return fail(GBWT_HIP_DEVICE_ERROR, "unexpected exception");
    }
}
#define GBWT_HIP_GUARD_BEGIN try {
#define GBWT_HIP_GUARD_END } catch (...) { return gbwt_hip::status_of_current_exception(); }

}  // namespace gbwt_hip

struct gbwt_hip_index {
    gbwt_hip::HostIndex host;
    int device = 0;
    gbwt_hip::DeviceBuffer data, starts, endmarker, desc, desc_raw, block_base, blocks, desc2, cblocks, gblocks, tables, wtables, seq_len, samples, sample_base;
    gbwt_hip::DeviceBuffer label_len;   // GBZ only: label length per potential node (0 for nodes that do not exist)
    // GBZ with a node-to-segment translation (src/graph.rs:186-218), flattened for the line formatter:
    gbwt_hip::DeviceBuffer seg_of;        // u32 per node id < mapping_len: segment holding the node (~0 before the first segment)
    gbwt_hip::DeviceBuffer seg_start;     // u32 per segment + 1: first node id of the segment, last = mapping_len
    gbwt_hip::DeviceBuffer seg_name_off;  // u64 per segment + 1: offsets into seg_names
    gbwt_hip::DeviceBuffer seg_names;     // segment names, concatenated
    gbwt_hip::DeviceBuffer seg_seq_len;   // u64 per segment: length of the segment's sequence
    gbwt_hip::DeviceBuffer node_real;     // u8 per node id < mapping_len: GBZ::has_node
    gbwt_hip::DeviceIndex dev{};
    uint32_t uniform_len = 0;         // every sequence has this many nodes (0: lengths differ, or unknown): an extraction then knows its offsets without asking the device
    bool orientation_pairs = false;   // verified at open: sequence 2k + 1 is sequence 2k reversed (rows can be filled from both ends)
    gbwt_hip_stats stats{};
};

struct gbwt_hip_workspace {
    const gbwt_hip_index *index = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t qev[2] = {nullptr, nullptr};   // around the kernel(s) of the last navigation / search call
    bool timed = false, query_timed = false;
    uint64_t last_n = 0, last_total = 0;   // shape of the last device-resident extraction
    uint32_t walk_mode = gbwt_hip::WALK_TWO_STEP, paths_per_wave = 0, small_record = 16;   // paths_per_wave 0 = automatic
    gbwt_hip::DeviceBuffer seq_ids, lengths, offsets, head, pool, next, counters, nodes, scan_temp;
    gbwt_hip::DeviceBuffer order_keys, order_rows, order_counts, order_level, order_temp;   // walker order of a segmented extraction
    gbwt_hip::DeviceBuffer in_a, in_b, out_a, out_valid, follow_off;  // search staging
    gbwt_hip::DeviceBuffer gfa_a, gfa_b, gfa_c, gfa_text, gfa_valid, gfa_chunk_first, gfa_chunks;  // GFA line formatting
    // What the device-resident results answer.  The C idiom "size query, then the same call with a buffer" (gbwt_hip_extract,
    // _follow, _path_lines) must not compute twice: a call that repeats the request of the results still in the workspace
    // copies them out.  Keys are host copies of the ids / states of the request.
    bool extract_cached = false, follow_cached = false, lines_cached = false;
    std::vector<uint64_t> extract_key, lines_key;
    std::vector<uint8_t> follow_key;
    int follow_backward = 0, lines_mode = 0;
    uint64_t follow_total = 0, lines_total = 0;
    ~gbwt_hip_workspace() {
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
        for (auto &e : qev) if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
