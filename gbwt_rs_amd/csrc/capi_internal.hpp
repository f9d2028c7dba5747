// capi_internal.hpp -- handle / workspace structures shared by the C-ABI translation units (capi.hip, gfa.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
// Where a pass's 13 GB of rows land physically is worth up to 10 % of its time: on a fresh box one hipMalloc of that size takes one
// contiguous stretch of VRAM, and a pass over it takes 4.95 ms where the same buffer put together from chunks that lie SPREAD over the
// VRAM takes 4.45 (profiles/r02_walk_bounds.txt #25; the fast and slow "states" of a box in round 1 were this).  So the rows buffer of
// a workspace (may_spread), from 4 GiB, is built with the virtual-memory API: `spread` times as many physical chunks as needed are created, every spread-th is
// kept and mapped into one virtual range, the others are given back.  Chunks of 2 GiB: with 256 MB chunks the rows of ragged batches
// (walks out of lock step) were written 5 % slower than into one hipMalloc -- smaller mappings, less TLB reach -- with 2 GiB 1 %.  The
// price is paid when a workspace is sized: the driver clears what it hands out, about 15 ms per GB created (1.7 s for a 13 GB buffer at
// spread 8; spread 4 keeps half of the gain, 2 nothing).  hipMalloc whenever any of that fails.
// GBWT_HIP_VMM="0": always hipMalloc;  "<chunk MiB>:<spread, 0 = as much as 3/4 of the free memory allows, at most 8>:<min MiB>".
struct VmmPolicy { size_t chunk = size_t(2048) << 20, min = size_t(4) << 30; unsigned spread = 0; };
inline const VmmPolicy &vmm_policy() {
    static const VmmPolicy policy = [] {
        VmmPolicy p;
        if (const char *v = std::getenv("GBWT_HIP_VMM")) {
            unsigned long chunk = 0, spread = 0, min = 4096;
            const int got = std::sscanf(v, "%lu:%lu:%lu", &chunk, &spread, &min);
            if (got >= 1) { p.chunk = chunk << 20; p.spread = static_cast<unsigned>(spread); p.min = min << 20; }
        }
        return p;
    }();
    return policy;
}

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> chunks;   // empty: ptr comes from hipMalloc
    size_t mapped = 0;                                       // bytes of virtual range reserved at ptr
    bool may_spread = false;                                 // the extracted rows of a workspace only: spreading the index arrays gains nothing (#25)
    ~DeviceBuffer() { release(); }
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    void release() noexcept {
        if (mapped) {
            (void)hipMemUnmap(ptr, mapped);
            (void)hipMemAddressFree(ptr, mapped);
        } else if (ptr) (void)hipFree(ptr);
        for (auto h : chunks) (void)hipMemRelease(h);
        chunks.clear();
        ptr = nullptr; bytes = 0; mapped = 0;
    }
    void reserve(size_t need) {
        if (need <= bytes) return;
        release();
        const size_t want = std::max<size_t>(need, 256);
        const VmmPolicy &policy = vmm_policy();
        if (may_spread && policy.chunk != 0 && want >= policy.min && spread_chunks(want, policy)) { bytes = want; return; }
        HIP_CHECK(hipMalloc(&ptr, want));
        bytes = want;
    }
    template <class T> T *as() const { return static_cast<T *>(ptr); }

private:
    // false (and nothing held) when the virtual-memory API does not cooperate
    bool spread_chunks(size_t want, const VmmPolicy &policy) noexcept {
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess) return false;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t granule = 0, free_bytes = 0, total_bytes = 0;
        if (hipMemGetAllocationGranularity(&granule, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || granule == 0) return false;
        if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess || want > free_bytes) return false;   // sizes out of a corrupt file: let hipMalloc say no
        const size_t chunk = (policy.chunk + granule - 1) / granule * granule, n = want / chunk + (want % chunk != 0 ? 1 : 0);
        size_t spread = policy.spread;
        if (spread == 0) spread = std::min<size_t>(8, std::max<size_t>(1, free_bytes / 4 * 3 / (n * chunk)));
        // more chunks than needed, every spread-th kept: the kept ones lie `spread` chunks apart in whatever order the driver hands them out
        std::vector<hipMemGenericAllocationHandle_t> all;
        all.reserve(n * spread);
        for (size_t i = 0; i < n * spread; i++) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) break;
            all.push_back(h);
        }
        (void)hipGetLastError();
        const size_t step = all.size() / n;     // >= 1 unless not even n chunks could be had
        for (size_t i = 0; i < all.size(); i++) {
            if (step != 0 && i % step == 0 && chunks.size() < n) chunks.push_back(all[i]);
            else (void)hipMemRelease(all[i]);
        }
        bool ok = chunks.size() == n && hipMemAddressReserve(&ptr, n * chunk, 0, nullptr, 0) == hipSuccess;
        if (ok) {
            mapped = n * chunk;
            for (size_t i = 0; i < n && ok; i++) ok = hipMemMap(static_cast<char *>(ptr) + i * chunk, chunk, 0, chunks[i], 0) == hipSuccess;
            hipMemAccessDesc access{};
            access.location = prop.location;
            access.flags = hipMemAccessFlagsProtReadWrite;
            ok = ok && hipMemSetAccess(ptr, mapped, &access, 1) == hipSuccess;
        }
        if (!ok) { (void)hipGetLastError(); release(); }
        return ok;
    }
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
    bool packed_blocks = true;        // gblocks was built (false: the index is too large for 32-bit half-block indices, or GBWT_HIP_GATHER_LIMIT=0)
    uint32_t max_samples = 0;         // the largest number of samples of a sequence
    uint32_t uniform_samples = 0;     // every sequence has this many samples (0: they differ): the walkers of an extraction are then w = segment * n + row
    gbwt_hip_open_times times{};      // where the time of the open went (gbwt_hip_get_open_times)
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
