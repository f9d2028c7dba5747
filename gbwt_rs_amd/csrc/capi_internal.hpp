// capi_internal.hpp -- handle / workspace structures shared by the C-ABI translation units (capi_open.hip, capi_extract.hip, capi_query.hip, gfa.hip, comm.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
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
// GBWT_HIP_VMM="0": always hipMalloc;  "<chunk MiB>:<spread, 0 = as much as half of the free memory allows, at most 8>:<min MiB>:<after>"
// (after = requests served as one hipMalloc before the buffer is rebuilt from spread chunks; 0 = spread at once).
struct VmmPolicy { size_t chunk = size_t(2048) << 20, min = size_t(4) << 30; unsigned spread = 0, after = 2; };
// GBWT_HIP_VMM, read when a workspace is created (the policy travels with its rows buffer)
inline VmmPolicy vmm_policy_from_env() {
    VmmPolicy p;
    if (const char *v = std::getenv("GBWT_HIP_VMM")) {
        unsigned long chunk = 0, spread = 0, min = 4096, after = 2;
        const int got = std::sscanf(v, "%lu:%lu:%lu:%lu", &chunk, &spread, &min, &after);
        if (got >= 1) { p.chunk = chunk << 20; p.spread = static_cast<unsigned>(spread); p.min = min << 20; }
        if (got >= 4) p.after = static_cast<unsigned>(after);
    }
    return p;
}

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> chunks;   // empty: ptr comes from hipMalloc
    size_t reserved = 0;                                     // bytes of virtual range reserved at ptr (0: hipMalloc)
    size_t chunk_bytes = 0, mapped_chunks = 0;               // the first mapped_chunks chunks are mapped at ptr + i * chunk_bytes
    bool may_spread = false;                                 // the extracted rows of a workspace only: spreading the index arrays gains nothing (#25)
    VmmPolicy policy;                                        // how (set by the workspace from GBWT_HIP_VMM)
    unsigned uses = 0;                                       // requests the buffer has served at its present size
    ~DeviceBuffer() { release(); }
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    // hipFree waits for the work that may still use the memory; unmapping does not.  The rows of a workspace are handed out as raw
    // device pointers (gbwt_hip_paths, dist.device_view) that a consumer may still be reading on ANOTHER stream -- a torch kernel,
    // an RCCL send -- so a mapped buffer waits for the whole device before it goes.
    void release() noexcept {
        if (reserved) {
            (void)hipDeviceSynchronize();
            // one hipMemUnmap per hipMemMap: a single call over the whole range is not documented to undo several mappings, and
            // a chunk that stays mapped keeps its 2 GiB of VRAM for good
            for (size_t i = 0; i < mapped_chunks; i++) (void)hipMemUnmap(static_cast<char *>(ptr) + i * chunk_bytes, chunk_bytes);
            (void)hipMemAddressFree(ptr, reserved);
        } else if (ptr) (void)hipFree(ptr);
        for (auto h : chunks) (void)hipMemRelease(h);
        (void)hipGetLastError();
        chunks.clear();
        ptr = nullptr; bytes = 0; reserved = 0; chunk_bytes = 0; mapped_chunks = 0;
    }
    // Spreading costs seconds (the driver clears every chunk it hands out, and gives the unused ones back lazily: the NEXT large
    // allocation of the process waits for that), so a buffer starts as one hipMalloc -- what a one-shot extraction sees -- and is
    // rebuilt from spread chunks once it has served `policy.after` requests: a workspace that is used again and again (a server, the
    // timed passes of bench.py) pays once and gains on every pass after that.  The contents do not survive; callers reserve before
    // they write.
    void reserve(size_t need) {
        if (need <= bytes) {
            if (may_spread && reserved == 0 && policy.chunk != 0 && bytes >= policy.min && ++uses == policy.after) {
                const size_t want = bytes;
                release();
                if (!spread_chunks(want)) HIP_CHECK(hipMalloc(&ptr, want));
                bytes = want;
            }
            return;
        }
        release();
        uses = 0;
        const size_t want = std::max<size_t>(need, 256);
        if (may_spread && policy.chunk != 0 && want >= policy.min && policy.after == 0 && spread_chunks(want)) { bytes = want; return; }
        HIP_CHECK(hipMalloc(&ptr, want));
        bytes = want;
    }
    template <class T> T *as() const { return static_cast<T *>(ptr); }

private:
    // false (and nothing held) when the virtual-memory API does not cooperate
    bool spread_chunks(size_t want) noexcept {
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
        // spread 0 = automatic: at most 8, and never more than HALF of the free memory in flight while the chunks are chosen -- other
        // contexts on the device (a second rank sharing the GPU, a torch allocator) must not see a spurious out-of-memory
        size_t spread = policy.spread;
        if (spread == 0) spread = std::min<size_t>(8, std::max<size_t>(1, free_bytes / 2 / (n * chunk)));
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
            reserved = n * chunk; chunk_bytes = chunk;
            for (size_t i = 0; i < n && ok; i++) {
                ok = hipMemMap(static_cast<char *>(ptr) + i * chunk, chunk, 0, chunks[i], 0) == hipSuccess;
                if (ok) mapped_chunks = i + 1;                 // a failure half-way unwinds exactly what was mapped
            }
            hipMemAccessDesc access{};
            access.location = prop.location;
            access.flags = hipMemAccessFlagsProtReadWrite;
            ok = ok && hipMemSetAccess(ptr, reserved, &access, 1) == hipSuccess;
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
    } catch (const Unsupported &e) {
        return fail(GBWT_HIP_UNSUPPORTED, e.what());
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
// The pinned staging buffers and streams of a workspace's device-to-host copies (capi_extract.hip: copy_to_host), made on first use.
struct HostCopier {
    static constexpr size_t CHUNK = size_t(16) << 20;
    struct Lane { void *pinned[2] = {nullptr, nullptr}; hipEvent_t landed[2] = {nullptr, nullptr}; hipStream_t stream = nullptr; };
    std::vector<Lane> lanes;
    std::mutex busy;
    HostCopier() = default;
    HostCopier(const HostCopier &) = delete;
    HostCopier &operator=(const HostCopier &) = delete;
    ~HostCopier();
    bool ensure(int device, unsigned threads);
};

#define GBWT_HIP_GUARD_BEGIN try {
#define GBWT_HIP_GUARD_END } catch (...) { return gbwt_hip::status_of_current_exception(); }

// The full-width two-step blocks of a handle, built on first need, and a copy of its DeviceIndex with them put in (capi_open.hip)
const uint4 *ensure_cblocks(const gbwt_hip_index *index);
DeviceIndex with_cblocks(const gbwt_hip_index *ix);

}  // namespace gbwt_hip

struct gbwt_hip_index {
    gbwt_hip::HostIndex host;
    int device = 0;
    bool lean_extract = false;           // opened without SEARCH and no record needs the generic decoder: desc_raw was given back after the open (capi_open.hip: open_common)
    uint64_t slow_records = ~uint64_t(0); // non-empty records whose walk descriptor says "generic decoder" (k_link_desc2's count; ~0 = not counted)
    uint32_t caps = GBWT_HIP_OPEN_ALL;   // what the handle was opened for (gbwt_hip_open_*_flags): which arrays exist, which entry points answer
    uint64_t table_positions = 0;     // BWT positions in records with LF tables (outdegree > 2)
    gbwt_hip::DeviceBuffer data, starts, endmarker, desc, desc_raw, block_base, blocks, desc2, cblocks, gblocks, tables, wtables, wtables_deep, seq_len, samples, sample_base;
    gbwt_hip::DeviceBuffer label_len;   // GBZ only: label length per potential node (0 for nodes that do not exist)
    // GBZ with a node-to-segment translation (src/graph.rs:186-218), flattened for the line formatter:
    gbwt_hip::DeviceBuffer seg_of;        // u32 per node id < mapping_len: segment holding the node (~0 before the first segment)
    gbwt_hip::DeviceBuffer seg_start;     // u32 per segment + 1: first node id of the segment, last = mapping_len
    gbwt_hip::DeviceBuffer seg_name_off;  // u64 per segment + 1: offsets into seg_names
    gbwt_hip::DeviceBuffer seg_names;     // segment names, concatenated
    gbwt_hip::DeviceBuffer seg_seq_len;   // u64 per segment: length of the segment's sequence
    gbwt_hip::DeviceBuffer node_real;     // u8 per node id < mapping_len: GBZ::has_node
    // GFA line headers, one table per line mode (0 = P-line named by the contig, 1 = W-line, 2 = P-line with the PanSN name): for every path of
    // the metadata the bytes of its line up to the node tokens -- for a W-line up to and including "<fragment>\t"; the end coordinate
    // (fragment + summed label lengths, path_to_w_line, src/bin/gbunzip.rs:532-540) is only known once the path has been walked and is
    // appended by the device.  Built once at open (gfa.hip: upload_label_lengths), so that a request formats its lines without the host.
    gbwt_hip::DeviceBuffer line_prefix[3], line_prefix_off[3], line_fragment;
    std::vector<char> host_line_prefix[3];
    std::vector<uint64_t> host_line_prefix_off[3];
    // LINE CACHE (gfa.hip): what the GFA line of a path is made of -- the text bytes of its node tokens, chunk by chunk, and its summed label
    // lengths (the W-line's end coordinate) -- is a property of the index, not of the request.  Round 5 let the first request that formats a
    // path leave them here; since round 6 ONE walk at open fills them for every path (fill_line_cache_at_open; kernels.hpp: LineCacheFill), so
    // that no request sizes a line: the node ids cross HBM twice (written by the walk, read by the formatter), never three times.
    // 8 bytes per 4 096 path positions + 16 per path.  Absent (-1) without sequence samples, for graphs with a node-to-segment translation
    // and with GBWT_HIP_LINE_CACHE=0: requests then size their lines themselves (k_chunk_stats).
    gbwt_hip::DeviceBuffer lc_chunk_first, lc_text, lc_path;    // u64[paths + 1]: first chunk of every path; u64 per chunk: W-token bytes of the path in front of it; u64[2] per path: {W-token bytes, summed label lengths}
    int lc_state = 0;                                             // 1 = filled at open, -1 = not for this index (written before the handle is handed out)
    std::vector<uint32_t> host_seq_len;   // host copy of seq_len (empty when the lengths are not known): sizes byte-bounded batches of a whole-file write
    gbwt_hip::DeviceIndex dev{};
    // The full-width two-step blocks (cblocks, as large as gblocks: 1.7 GB on the headline index) are only read by the loops for records
    // whose counts do not fit the packed half-blocks, by the pool-output kernel and by the serial walks at open: built at open when one of
    // those is certain to run, else on the first request that needs them (ensure_cblocks; once, whichever thread comes first).
    mutable std::once_flag cblocks_once;
    std::atomic<const uint4 *> lazy_cblocks{nullptr};
    bool packed_blocks = true;        // gblocks was built (false: the index is too large for 32-bit half-block indices, or GBWT_HIP_GATHER_LIMIT=0)
    uint32_t max_samples = 0;         // the largest number of samples of a sequence
    uint32_t sample_coarse = 1;       // the samples are this many times finer than a batch of the whole index wants: extractions stride over them (capi_extract.hip)
    std::vector<uint32_t> sample_counts;   // samples of every sequence (host copy: an extraction looks whether its rows all have the same number)
    uint32_t uniform_samples = 0;     // every sequence has this many samples (0: they differ): the walkers of an extraction are then w = segment * n + row
    bool starts_uploaded = false;         // ... and the record starts
    bool record_bytes_uploaded = false;   // gbwt_hip_open_file has copied the record bytes to `data` while the loader was still decoding
    gbwt_hip_open_times times{};      // where the time of the open went (gbwt_hip_get_open_times)
    uint32_t uniform_len = 0;         // every sequence has this many nodes (0: lengths differ, or unknown): an extraction then knows its offsets without asking the device
    bool orientation_pairs = false;   // verified at open: sequence 2k + 1 is sequence 2k reversed (rows can be filled from both ends)
    gbwt_hip_stats stats{};
};

// The GBWT_HIP_* tuning / measurement switches of an extraction, read ONCE when the workspace is created: gbwt_hip_extract_device
// itself never looks at the environment (getenv is not safe against a concurrent setenv, and several host threads extract from one
// index at the same time, each with its own workspace -- include/gbwt_hip.h).  -1 = not set: the library's default for the batch.
struct ExtractKnobs {
    int direct = 1, segments = 1, both_ends = 1;             // GBWT_HIP_DIRECT / _SEGMENTS / _BOTH_ENDS (0 switches the feature off)
    int all4 = 1;                                             // GBWT_HIP_ALL4: 0 = the uniform loop counts every node it stages (rounds 1-3)
    int sample_stride = -1;                                   // GBWT_HIP_SAMPLE_STRIDE: a walker per this many samples of a row; -1 = by the size of the batch (gbwt_hip_extract_device)
    int defer_total = 1;                                      // GBWT_HIP_DEFER_TOTAL: 0 = every request waits for the total of its row lengths before it launches the walk (rounds 1-3)
    int fused_offsets = 1;                                    // GBWT_HIP_FUSED_OFFSETS: 0 = the offsets of a batch of equally long rows come from k_row_offsets in front of the walk, as every batch's did until round 6 (WalkArgs::uniform_len)
    int ring_slots = -1, helper_naps = -1, xcd_map = -1, uniform_loop = -1, packed_blocks = -1, row_piece = -1, catch_up = -1, headroom = 0;
    int gather_reach = -1;                                    // GBWT_HIP_GATHER_REACH: look-ahead of mixed waves (WalkArgs::gather_reach); -1 = by the index (rows per record), 0 = none
    bool wide_addresses = false;                              // GBWT_HIP_WIDE_ADDRESSES set (any value)
    uint32_t debug = 0;                                       // GBWT_HIP_DEBUG_DRY_ROWS (measurement switches, WalkArgs::debug)
    unsigned copy_threads = 8;                                // GBWT_HIP_COPY_THREADS
    static ExtractKnobs from_env() {
        ExtractKnobs k;
        const auto num = [](const char *name, int unset) { const char *v = std::getenv(name); return v ? std::atoi(v) : unset; };
        k.direct = num("GBWT_HIP_DIRECT", 1); k.segments = num("GBWT_HIP_SEGMENTS", 1); k.both_ends = num("GBWT_HIP_BOTH_ENDS", 1);
        k.ring_slots = num("GBWT_HIP_RING_SLOTS", -1); if (k.ring_slots != 32 && k.ring_slots != 64 && k.ring_slots != 128) k.ring_slots = -1;
        k.helper_naps = std::max(-1, num("GBWT_HIP_HELPER_NAPS", -1));
        k.xcd_map = num("GBWT_HIP_XCD_MAP", -1); k.uniform_loop = num("GBWT_HIP_UNIFORM_LOOP", -1); k.packed_blocks = num("GBWT_HIP_PACKED_BLOCKS", -1);
        k.catch_up = num("GBWT_HIP_CATCH_UP", -1);
        k.gather_reach = std::min(64, std::max(-1, num("GBWT_HIP_GATHER_REACH", -1)));
        k.sample_stride = num("GBWT_HIP_SAMPLE_STRIDE", -1);
        k.defer_total = num("GBWT_HIP_DEFER_TOTAL", 1);
        k.fused_offsets = num("GBWT_HIP_FUSED_OFFSETS", 1);
        k.all4 = num("GBWT_HIP_ALL4", 1);
        k.headroom = std::min(32, std::max(0, num("GBWT_HIP_HEADROOM", 0)));
        k.row_piece = num("GBWT_HIP_ROW_PIECE", -1); if (k.row_piece != 0 && k.row_piece != 16 && k.row_piece != 32) k.row_piece = -1;
        k.wide_addresses = std::getenv("GBWT_HIP_WIDE_ADDRESSES") != nullptr;
        k.debug = static_cast<uint32_t>(num("GBWT_HIP_DEBUG_DRY_ROWS", 0));
        k.copy_threads = static_cast<unsigned>(std::min(64, std::max(1, num("GBWT_HIP_COPY_THREADS", 8))));
        return k;
    }
};

struct gbwt_hip_workspace {
    const gbwt_hip_index *index = nullptr;
    ExtractKnobs knobs;
    hipStream_t stream = nullptr;
    uint64_t *pinned_words = nullptr;         // four words of pinned host memory: the total and extremes of a request whose rows are sized after its launch (capi_extract.hip)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t qev[2] = {nullptr, nullptr};   // around the kernel(s) of the last navigation / search call
    hipEvent_t gev[2] = {nullptr, nullptr};   // around the formatting of the last GFA lines request (behind its walk)
    bool timed = false, query_timed = false, lines_timed = false;
    uint64_t last_n = 0, last_total = 0;   // shape of the last device-resident extraction
    uint32_t walk_mode = gbwt_hip::WALK_TWO_STEP, paths_per_wave = 0, small_record = 16;   // paths_per_wave 0 = automatic
    gbwt_hip::DeviceBuffer seq_ids, lengths, offsets, head, pool, next, counters, nodes, scan_temp;
    gbwt_hip::DeviceBuffer order_keys, order_rows, order_counts, order_level, order_temp;   // walker order of a segmented extraction
    gbwt_hip::DeviceBuffer in_a, in_b, out_a, out_valid, follow_off;  // search staging
    gbwt_hip::DeviceBuffer gfa_a, gfa_b, gfa_c, gfa_text, gfa_text2, gfa_valid, gfa_chunk_first, gfa_chunks, gfa_plan;  // GFA line formatting (gfa_text2: the second text buffer of a pipelined whole-file write)
    // What the device-resident results answer.  The C idiom "size query, then the same call with a buffer" (gbwt_hip_extract,
    // _follow, _path_lines) must not compute twice: a call that repeats the request of the results still in the workspace
    // copies them out.  Keys are host copies of the ids / states of the request.
    bool extract_cached = false, follow_cached = false, lines_cached = false;
    std::vector<uint64_t> extract_key, lines_key;
    std::vector<uint8_t> follow_key;
    int follow_backward = 0, lines_mode = 0, lines_slot = 0;
    gbwt_hip::HostCopier copier;      // pinned staging of the large device-to-host copies
    uint64_t follow_total = 0, lines_total = 0;
    ~gbwt_hip_workspace() {
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
        for (auto &e : qev) if (e) (void)hipEventDestroy(e);
        for (auto &e : gev) if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
        if (pinned_words) (void)hipHostFree(pinned_words);
    }
};

namespace gbwt_hip {
// Device -> pageable host memory over the workspace's copy threads (GBWT_HIP_COPY_THREADS, default 8), each with two pinned staging
// buffers and a stream of its own (capi_extract.hip)
void copy_to_host(gbwt_hip_workspace *ws, void *dst, const void *src, size_t bytes, size_t piece = HostCopier::CHUNK);
}
