// capi_open.hip -- the C ABI of libgbwt_hip.so (include/gbwt_hip.h), part 1 of 3: opening an index -- the loader's image to the device, the
// device passes that build descriptors, rank blocks, tables, sequence lengths and samples (upload), what a handle keeps (open_common) -- and
// the entry points that open, close and describe a handle.  (capi_extract.hip: workspaces + extraction; capi_query.hip: navigation + search;
// gfa.hip: GFA lines; comm.hip: the gather.)  There is deliberately no CPU implementation of any compute entry point in this library:
// without a HIP device they fail with GBWT_HIP_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "capi_internal.hpp"

using namespace gbwt_hip;


namespace gbwt_hip {
std::string &last_error_slot() {
    thread_local std::string slot;
    return slot;
}
}  // namespace gbwt_hip

namespace gbwt_hip { void upload_label_lengths(gbwt_hip_index &ix); void mask_label_lengths(gbwt_hip_index &ix); void fill_line_cache_at_open(gbwt_hip_index &ix); }

namespace gbwt_hip {
// The full-width two-step blocks, built on first need (gbwt_hip_index::cblocks_once).  After open, `dev.cblocks` of a handle is never
// written again -- other threads copy `dev` into their launches all the time -- so the pointer of a lazily built array lives in an atomic
// of its own, and a launch that needs it takes a copy of `dev` with the pointer put in (with_cblocks).
const uint4 *ensure_cblocks(const gbwt_hip_index *index) {
    gbwt_hip_index *ix = const_cast<gbwt_hip_index *>(index);     // the lazily built part of an otherwise immutable handle
    if (ix->dev.cblocks != nullptr) return ix->dev.cblocks;        // built at open
    if (ix->lean_extract) throw Unsupported("this handle was opened for extraction only and has given its raw descriptors back: the full-width blocks cannot be built (open with GBWT_HIP_OPEN_ALL)");
    std::call_once(ix->cblocks_once, [ix]() {
        HIP_CHECK(hipSetDevice(ix->device));
        ix->cblocks.reserve(std::max<uint64_t>(ix->dev.n_blocks, 1) * 2 * sizeof(uint4));
        HIP_CHECK(hipMemsetAsync(ix->cblocks.ptr, 0, 2 * sizeof(uint4), nullptr));
        if (ix->dev.n_blocks > 1) launch_fill_two_step_blocks(ix->dev, ix->cblocks.as<uint4>(), nullptr, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipGetLastError());
        ix->lazy_cblocks.store(ix->cblocks.as<uint4>(), std::memory_order_release);
    });
    return ix->lazy_cblocks.load(std::memory_order_acquire);
}

DeviceIndex with_cblocks(const gbwt_hip_index *ix) {
    DeviceIndex d = ix->dev;
    d.cblocks = ensure_cblocks(ix);
    return d;
}

}  // namespace gbwt_hip

namespace {

void fill_stats(gbwt_hip_index &ix) {
    const HostIndex &h = ix.host;
    gbwt_hip_stats &s = ix.stats;
    s.size = h.size; s.sequences = h.sequences; s.alphabet_size = h.alphabet_size; s.alphabet_offset = h.alphabet_offset;
    s.records = h.records(); s.data_bytes = h.record_bytes_len(); s.paths = h.path_names.size();
    s.bidirectional = h.bidirectional; s.has_metadata = h.has_metadata; s.is_gbz = h.is_gbz; s.has_translation = h.has_translation;
}

// GBWT_HIP_TRACE_OPEN=1: where the wall time of an open goes, phase by phase, on stderr (each mark waits for the device first)
struct OpenTrace {
    bool on = std::getenv("GBWT_HIP_TRACE_OPEN") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[open] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = std::chrono::steady_clock::now();
    }
};

// max / min / common number of samples per sequence from the host copy of sample_base
void note_sample_counts(gbwt_hip_index &ix, const std::vector<uint64_t> &base) {
    uint64_t lo = ~uint64_t(0), hi = 0;
    for (size_t k = 0; k + 1 < base.size(); k++) { const uint64_t c = base[k + 1] - base[k]; lo = std::min(lo, c); hi = std::max(hi, c); }
    ix.max_samples = static_cast<uint32_t>(std::min<uint64_t>(hi, 0xFFFFFFFFull));
    ix.uniform_samples = (base.size() > 1 && lo == hi) ? ix.max_samples : 0u;
    ix.sample_counts.resize(base.size() - 1);
    for (size_t k = 0; k + 1 < base.size(); k++) ix.sample_counts[k] = static_cast<uint32_t>(std::min<uint64_t>(base[k + 1] - base[k], 0xFFFFFFFFull));
}

// Lengths and samples of all sequences by checkpoint sampling (open_walks.hip; kernels.hpp: CheckpointWalk).  d_flags[0] collects
// the overflow bits of the chase like the serial walks do.  false: nothing usable was built (the caller walks every sequence).
bool checkpoint_samples(gbwt_hip_index &ix, uint32_t interval, uint32_t *d_flags, OpenTrace &trace) {
    const HostIndex &h = ix.host;
    DeviceIndex &d = ix.dev;
    const uint64_t S = h.sequences, nr = d.n_records;
    if (nr == 0 || S == 0 || d.desc2 == nullptr) return false;
    // a checkpoint per `interval` LF steps that end on a record that can be one (an LF step emits one node, two where it is fused with a
    // unary successor: every interval .. 2 x interval nodes), and at most `cap` = interval (+ 3) nodes per hop: most hops end at the cap,
    // the checkpoints are where rows that have drifted apart meet again (profiles/r03_sampling_ab.txt)
    // LF steps (that end on a record which can be a checkpoint) per checkpoint.  Where rows move in lock step a hop that ends at the cap
    // ends on the same record for all of them; where they do not (indexes with chained steps: insertions, chopped nodes) it ends a few
    // sites apart and the wave that takes up those segments stays mixed to their end -- there most hops should end at a checkpoint, which
    // every row passes: gap = interval / 8 (insertion chain 486 -> 650 G LF-steps/s, insertions at every 8th site 580 -> 680 G, chopped
    // nodes with insertions 510 -> 650 G; the lock-step chain is indifferent down to interval / 8 and loses a tenth at interval / 16, but
    // pays four times the hops of the chase at open: profiles/r03_chained_steps.txt)
    // (... of an interval of 2 048 nodes.  The shorter intervals of smaller indexes keep at least 128 LF steps between checkpoints, or half
    // the interval: with interval / 8 of 256 a config-4-shaped index -- 90 haplotypes over 480 components -- was cut into segments of 30 to
    // 60 nodes, ten million walkers of a dozen iterations each, and walked at 2.53 ms where a gap of 128 takes 2.07; round 4,
    // profiles/r04_walk_experiments.txt)
    double gap = d.chained != 0 ? std::max(0.125 * interval, std::min(128.0, 0.5 * interval)) : 0.5 * interval;
    if (const char *v = std::getenv("GBWT_HIP_CHECKPOINT_GAP")) gap = std::max(2.0, std::atof(v));
    const double q = std::min(0.5, 1.0 / gap);
    CheckpointWalk w{};
    w.threshold = static_cast<uint32_t>(q * 4294967296.0);
    w.cap = interval;     // A/B on the headline (profiles/r03_sampling_ab.txt): cap = interval 4.31 ms per pass, 1.5 x 4.38, 2 x 4.46; the serial samples 4.34
    w.packed = ix.packed_blocks ? 1u : 0u;
    if (const char *v = std::getenv("GBWT_HIP_CHECKPOINT_CAP")) w.cap = static_cast<uint32_t>(std::max(1, std::atoi(v)));
    // two allocations for all temporaries (seven of each, and seven frees, were a millisecond of the open): what is sized by the records
    // and sequences, then what is sized by the checkpoint positions
    struct Carver {
        DeviceBuffer arena; size_t at = 0;
        static size_t padded(size_t bytes) { return (bytes + 255) / 256 * 256; }
        void *take(size_t bytes) { void *p = static_cast<char *>(arena.ptr) + at; at += padded(bytes); return p; }
    } small, large;
    const size_t tb = scan_temp_bytes(nr), sb = scan_temp_bytes(S), scan_bytes = std::max<size_t>(std::max(tb, sb), 16);
    small.arena.reserve(Carver::padded(nr * sizeof(uint64_t)) + Carver::padded((nr + 1) * sizeof(uint64_t)) + Carver::padded(scan_bytes) + Carver::padded(2 * sizeof(uint64_t)) +
                        Carver::padded(S * sizeof(uint64_t)));
    uint64_t *counts = static_cast<uint64_t *>(small.take(nr * sizeof(uint64_t))), *cp_first = static_cast<uint64_t *>(small.take((nr + 1) * sizeof(uint64_t)));
    void *scan_tmp = small.take(scan_bytes);
    uint64_t *misc = static_cast<uint64_t *>(small.take(2 * sizeof(uint64_t))), *per_sequence = static_cast<uint64_t *>(small.take(S * sizeof(uint64_t)));
    trace.mark("  (before the checkpoint passes)");
    launch_checkpoint_counts(d, w.threshold, counts, nullptr);
    launch_scan(counts, cp_first, nr, scan_tmp, tb, nullptr);
    uint64_t positions = 0;
    HIP_CHECK(hipMemcpy(&positions, cp_first + nr, sizeof(uint64_t), hipMemcpyDeviceToHost));
    trace.mark("    checkpoint counts + scan");
    // every orphan stands for a finished walk of at least `cap` nodes, and no two walkers share a BWT position
    // (every node of a sequence is one BWT position: at most max_walk / cap hops end at the cap.  A quarter more for the hops of up to cap + 3
    // nodes; 2 x until round 4 -- 50 MB of summaries more on the headline index, and every allocation of an open is paid in milliseconds)
    const uint64_t orphan_capacity = d.max_walk / w.cap + d.max_walk / w.cap / 4 + S + 1024;
    const uint64_t n_summaries = S + positions + orphan_capacity;
    if (n_summaries >= 0xFFFFFFF0ull) return false;
    const size_t span_slots = S + n_summaries / 16 + 2;                     // one splitter per sixteen summaries, behind the sequence starts (open_walks.hip: span_slot)
    large.arena.reserve(Carver::padded(n_summaries * sizeof(uint4)) + Carver::padded(span_slots * sizeof(uint4)));
    uint4 *summaries = static_cast<uint4 *>(large.take(n_summaries * sizeof(uint4))), *spans = static_cast<uint4 *>(large.take(span_slots * sizeof(uint4)));
    trace.mark("    summaries allocated");
    HIP_CHECK(hipMemset(misc, 0, 2 * sizeof(uint64_t)));
    w.cp_first = cp_first; w.summaries = summaries;
    w.orphan_count = misc; w.orphan_capacity = orphan_capacity; w.positions = positions;
    w.flags = reinterpret_cast<uint32_t *>(misc + 1);
    trace.mark("  checkpoint counts + scan + allocations");
    launch_checkpoint_walk(d, w, nullptr);
    trace.mark("  k_checkpoint_walk");
    uint64_t state[2] = {0, 0};
    HIP_CHECK(hipMemcpy(state, misc, sizeof(state), hipMemcpyDeviceToHost));
    HIP_CHECK(hipGetLastError());
    if ((state[1] & 4u) || state[0] > orphan_capacity) return false;   // more orphans than a consistent index can have (walks in circles)
    ix.times.checkpoint_rounds = 1; ix.times.checkpoint_walkers = S + positions; ix.times.checkpoint_orphans = state[0];
    // the chase: count, scan, write
    ix.seq_len.reserve(S * sizeof(uint32_t));
    ix.sample_base.reserve((S + 1) * sizeof(uint64_t));
    const uint64_t used = S + positions + state[0];                         // (the orphan slots nobody took hold nothing)
    launch_chase_counts(d, summaries, used, spans, ix.seq_len.as<uint32_t>(), per_sequence, d_flags, nullptr);
    launch_scan(per_sequence, ix.sample_base.as<uint64_t>(), S, scan_tmp, sb, nullptr);
    std::vector<uint64_t> base(S + 1);
    HIP_CHECK(hipMemcpy(base.data(), ix.sample_base.ptr, base.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint32_t overflow = 0;
    HIP_CHECK(hipMemcpy(&overflow, d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost));
    trace.mark("  chase: counts + scan");
    if (overflow & 8u) {                                                    // lists that run into each other: no list ranking, the serial walk takes over
        HIP_CHECK(hipMemset(d_flags, 0, sizeof(uint32_t)));
        return false;
    }
    if (overflow) return true;                                              // the caller reports it; no samples
    ix.samples.reserve(std::max<uint64_t>(base[S], 1) * sizeof(uint4));
    launch_chase_samples(d, summaries, used, spans, ix.sample_base.as<uint64_t>(), ix.samples.as<uint4>(), nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipGetLastError());
    trace.mark("  chase: samples");
    ix.times.samples = base[S];
    note_sample_counts(ix, base);
    return true;
}

// The record starts to the device: narrowed to u32 where the record stream is shorter than 4 GiB.
void upload_starts(gbwt_hip_index &ix) {
    const HostIndex &h = ix.host;
    const bool narrow = ix.stats.data_bytes < (uint64_t(1) << 32);
    ix.starts.reserve(std::max<size_t>(h.starts.size(), 1) * (narrow ? sizeof(uint32_t) : sizeof(uint64_t)));
    if (narrow) {
        // (218 M starts of an HPRC-sized index: narrowed on a few threads, 0.2 s on one)
        const size_t n = h.starts.size();
        std::unique_ptr<uint32_t[]> s32(new uint32_t[std::max<size_t>(n, 1)]);
        s32[0] = 0;
        const unsigned pieces = n >= (size_t(1) << 23) ? std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1u;
        auto piece = [&](unsigned p) { for (size_t k = n * p / pieces, end = n * (p + 1) / pieces; k < end; k++) s32[k] = static_cast<uint32_t>(h.starts[k]); };
        std::vector<std::thread> pool;
        for (unsigned p = 1; p < pieces; p++) pool.emplace_back(piece, p);
        piece(0);
        for (auto &t : pool) t.join();
        HIP_CHECK(hipMemcpy(ix.starts.ptr, s32.get(), std::max<size_t>(n, 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    } else if (!h.starts.empty()) {
        HIP_CHECK(hipMemcpy(ix.starts.ptr, h.starts.data(), h.starts.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    ix.starts_uploaded = true;
}

// The record starts decoded on the device from the Elias-Fano words in the mapped file (HostIndex::starts_view; kernels.hpp: launch_ef_*).
// Round 5 decoded them on the host, narrowed them to u32 and sent them: 0.35 s of config 4's open for its 218 M starts (0.19 s of decode on up
// to 32 threads in front of it), 2.5 ms of the headline's -- all of it in front of the first device pass.  The words are a sixth of the table
// (137 of 872 MB for config 4) and the decode is three launches; the host's own decode (for the GFA tables and the graph lines) runs in the
// loader's background thread meanwhile and reports what is wrong with the words as it always did (HostIndex::finish).
void decode_starts_on_device(gbwt_hip_index &ix) {
    const HostIndex::StartsView v = ix.host.starts_view;
    const uint64_t data_len = ix.stats.data_bytes, ones = v.ones;
    const bool narrow = data_len < (uint64_t(1) << 32);
    DeviceBuffer words, ranks, scan_tmp, flags;
    words.reserve(std::max<uint64_t>(v.high_words + v.low_words, 1) * sizeof(uint64_t));
    ranks.reserve(2 * (v.high_words + 1) * sizeof(uint64_t));
    const size_t tb = scan_temp_bytes(std::max<uint64_t>(v.high_words, 1));
    scan_tmp.reserve(std::max<size_t>(tb, 16));
    flags.reserve(sizeof(uint32_t));
    HIP_CHECK(hipMemsetAsync(flags.ptr, 0, sizeof(uint32_t), nullptr));
    uint64_t *d_high = words.as<uint64_t>(), *d_low = d_high + v.high_words, *d_counts = ranks.as<uint64_t>(), *d_rank = d_counts + (v.high_words + 1);
    if (v.high_words) HIP_CHECK(hipMemcpy(d_high, v.high, v.high_words * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (v.low_words) HIP_CHECK(hipMemcpy(d_low, v.low, v.low_words * sizeof(uint64_t), hipMemcpyHostToDevice));
    launch_ef_counts(d_high, v.high_words, d_counts, nullptr);
    if (v.high_words) launch_scan(d_counts, d_rank, v.high_words, scan_tmp.ptr, tb, nullptr);
    else HIP_CHECK(hipMemsetAsync(d_rank, 0, sizeof(uint64_t), nullptr));
    // the declared number of ones must be the number of set bits BEFORE it sizes the table (a corrupt count could ask for 32 times the file
    // size of device memory and surface as a device error instead of InvalidData: the host decode checks in the same order)
    uint64_t total = 0;
    HIP_CHECK(hipMemcpy(&total, d_rank + v.high_words, sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (total != ones) throw InvalidData("SparseVector: high bitvector does not have the declared number of ones");
    ix.starts.reserve((ones + 1) * (narrow ? sizeof(uint32_t) : sizeof(uint64_t)));
    launch_ef_values(d_high, v.high_words, d_rank, d_low, v.low_words, static_cast<uint32_t>(v.low_width), ones, data_len,
                     narrow ? ix.starts.as<uint32_t>() : nullptr, narrow ? nullptr : ix.starts.as<uint64_t>(), nullptr);
    launch_starts_check(narrow ? ix.starts.as<uint32_t>() : nullptr, narrow ? nullptr : ix.starts.as<uint64_t>(), ones, flags.as<uint32_t>(), nullptr);
    uint32_t bad = 0;
    HIP_CHECK(hipMemcpy(&bad, flags.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipGetLastError());
    if (bad != 0) throw InvalidData("BWT: record starts are not sorted offsets into the data");
    ix.starts_uploaded = true;
}

// Uploads the host image and runs the load-time device passes.  `endmarker`: record 0 already decompressed by the caller (who is then free
// to let another thread finish the loader's background work meanwhile: nothing below touches the record bytes of the host image), or null.
void upload(gbwt_hip_index &ix, const std::vector<std::pair<uint32_t, uint32_t>> *early_endmarker) {
    OpenTrace trace;
    HostIndex &h = ix.host;
    if (h.alphabet_size > (uint64_t(1) << 32)) throw Unsupported("alphabet_size > 2^32 is not supported (u32 node ids on device)");
    HIP_CHECK(hipSetDevice(ix.device));
    const uint64_t n_records = ix.stats.records;           // (fill_stats; the host image's own starts and record bytes may still be on their way: HostIndex::pending)
    const uint64_t data_bytes = ix.stats.data_bytes;
    if (!ix.record_bytes_uploaded) {
        ix.data.reserve(data_bytes + DATA_PAD);
        HIP_CHECK(hipMemset(ix.data.ptr, 0, data_bytes + DATA_PAD));
    }
    DeviceIndex &d = ix.dev;
    d = DeviceIndex{};
    d.data = ix.data.as<uint8_t>();
    d.data_len = data_bytes;
    d.n_records = n_records;
    d.n_sequences = h.sequences;
    d.alphabet_offset = static_cast<uint32_t>(h.alphabet_offset);
    d.first_node = static_cast<uint32_t>(h.alphabet_offset + 1);
    // the record starts: dense, u32 where the stream is shorter than 4 GiB (gbwt_hip_open_file has sent them already, under the tail of
    // the copy of the record bytes)
    const bool narrow_starts = data_bytes < (uint64_t(1) << 32);
    if (!ix.starts_uploaded) upload_starts(ix);
    if (!ix.record_bytes_uploaded && data_bytes) HIP_CHECK(hipMemcpy(ix.data.ptr, h.record_bytes(), data_bytes, hipMemcpyHostToDevice));
    if (narrow_starts) d.starts32 = ix.starts.as<uint32_t>(); else d.starts64 = ix.starts.as<uint64_t>();
    trace.mark("record bytes + starts to the device");

    // Load-time device passes: per-record descriptors + rank blocks + record statistics, then the endmarker
    // (src/gbwt.rs:413-414).
    DeviceBuffer tmp;
    tmp.reserve(8 * sizeof(uint64_t));
    HIP_CHECK(hipMemset(tmp.ptr, 0, 8 * sizeof(uint64_t)));
    uint64_t *d_stats = tmp.as<uint64_t>();
    {
        const uint64_t nr = std::max<uint64_t>(n_records, 1);
        if (ix.caps & GBWT_HIP_OPEN_EXTRACT) ix.desc.reserve(nr * 4 * sizeof(uint4));
        ix.desc_raw.reserve(nr * 4 * sizeof(uint4));
        ix.block_base.reserve(nr * sizeof(uint32_t));
        DeviceBuffer counts, scan_tmp;
        counts.reserve(nr * sizeof(uint32_t));
        trace.mark("starts + allocations");
        launch_build_desc(d, ix.desc_raw.as<uint4>(), counts.as<uint32_t>(), d_stats, nullptr);
        trace.mark("k_build_desc");
        d.desc_raw = ix.desc_raw.as<uint4>();
        d.desc = ix.desc.as<uint4>();
        uint64_t n_blocks = 1;  // block 0: all zero, read by the records that have no blocks of their own
        uint64_t generic_records = n_records;   // class 0 records (k_build_desc's count, read below)
        if (n_records > 0) {
            if (n_records >= (uint64_t(1) << 30)) throw Unsupported("more than 2^30 records are not supported");
            if ((h.size >> RANK_BLOCK_SHIFT) + n_records >= 0xFFFFFFF0ull) throw Unsupported("index too large for 32-bit rank block indices");
            size_t tb = block_scan_temp_bytes(n_records);
            scan_tmp.reserve(std::max<size_t>(tb, 16));
            launch_block_scan(counts.as<uint32_t>(), ix.block_base.as<uint32_t>(), n_records, scan_tmp.ptr, tb, nullptr);
            uint32_t last_base = 0, last_count = 0;
            HIP_CHECK(hipMemcpy(&last_base, ix.block_base.as<uint32_t>() + (n_records - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(&last_count, counts.as<uint32_t>() + (n_records - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
            n_blocks += static_cast<uint64_t>(last_base) + last_count;
            launch_finish_block_base(counts.as<uint32_t>(), ix.block_base.as<uint32_t>(), n_records, ix.desc_raw.as<uint4>(), nullptr);
        }
        ix.blocks.reserve(n_blocks * sizeof(uint4));
        HIP_CHECK(hipMemsetAsync(ix.blocks.ptr, 0, sizeof(uint4), nullptr));
        d.block_base = ix.block_base.as<uint32_t>();
        d.blocks = ix.blocks.as<uint4>();
        d.n_blocks = n_blocks;
        trace.mark("block scan + allocation");
        if (n_blocks > 1) launch_fill_blocks(d, counts.as<uint32_t>(), ix.block_base.as<uint32_t>(), ix.blocks.as<uint4>(), nullptr);
        trace.mark("k_fill_blocks");
        const bool for_extract = (ix.caps & GBWT_HIP_OPEN_EXTRACT) != 0;
        uint64_t generic_count = n_records;
        if (!for_extract) {
            // a handle opened for SEARCH only: no walk descriptors, no two-step blocks (the statistics the passes below would have left are read here)
            uint64_t early_stats[6] = {0, 0, 0, 0, 0, 0};
            HIP_CHECK(hipMemcpy(early_stats, d_stats, sizeof(early_stats), hipMemcpyDeviceToHost));
            generic_count = early_stats[4];
            d.desc = nullptr; d.desc2 = nullptr; d.gblocks = nullptr; d.cblocks = nullptr;
            ix.packed_blocks = false;
        }
        if (for_extract) launch_link_desc(d, ix.desc.as<uint4>(), nullptr);
        if (for_extract) {
            uint32_t hops = 15;   // LF steps between a walk and its look-ahead target (7 until the packed blocks and the spread rows: fresh processes,
                                  // 7 / 11 / 15 / 23 hops = 4.40-4.44 / 4.30-4.33 / 4.27-4.28 / 4.29-4.30 ms; profiles/r02_walk_bounds.txt #26)
            if (const char *v = std::getenv("GBWT_HIP_LOOKAHEAD_HOPS")) hops = static_cast<uint32_t>(std::max(0, std::atoi(v)));
            launch_link_lookahead(d, ix.desc.as<uint4>(), counts.as<uint32_t>(), hops, nullptr);
            // two-step walk: composed descriptors + two-step blocks
            ix.desc2.reserve(nr * 8 * sizeof(uint4));
            d.desc2 = ix.desc2.as<uint4>();
            d.cblocks = nullptr;
            uint32_t gather_limit = 1u << 21;     // the counts of the gather loop's packed blocks (tests lower it)
            if (const char *v = std::getenv("GBWT_HIP_GATHER_LIMIT")) gather_limit = static_cast<uint32_t>(std::min<long>(1l << 21, std::max<long>(0, std::atol(v))));
            if (n_blocks >= (uint64_t(1) << 31)) gather_limit = 0;   // half-block indices 2 bb + offset / 32 are 32-bit in the loops: beyond that, full-width blocks only
            ix.packed_blocks = gather_limit != 0;
            trace.mark("link + lookahead + allocations");
            // steps through runs of unary records with consecutive ids (k_link_desc2: CHAINS); GBWT_HIP_CHAINS=0: none, k: at most k more nodes per step
            uint32_t chain_max = CHAIN_MAX;
            if (const char *v = std::getenv("GBWT_HIP_CHAINS")) chain_max = static_cast<uint32_t>(std::min<long>(CHAIN_MAX, std::max<long>(0, std::atol(v))));
            HIP_CHECK(hipMemsetAsync(d_stats + 6, 0, sizeof(uint64_t), nullptr));      // (word 2 behind the chain statistics: slow records)
            launch_link_desc2(d, ix.desc2.as<uint4>(), gather_limit, chain_max | (h.bidirectional ? 0x100u : 0u), reinterpret_cast<uint32_t *>(d_stats + 5), nullptr);
            ix.gblocks.reserve((gather_limit ? n_blocks : 1) * 2 * sizeof(uint4));   // no record takes the packed path: only the zero block
            HIP_CHECK(hipMemsetAsync(ix.gblocks.ptr, 0, 2 * sizeof(uint4), nullptr));
            d.gblocks = ix.gblocks.as<uint4>();
            // both layouts in one pass when the full-width one is certain to be read: a record whose counts do not fit the packed blocks
            // (k_link_desc2 clears GATHER_OK from 2^21 positions, for the record and for what lies behind its edges) or no packed blocks at all
            uint64_t early_stats[6] = {0, 0, 0, 0, 0, 0};
            HIP_CHECK(hipMemcpy(early_stats, d_stats, sizeof(early_stats), hipMemcpyDeviceToHost));
            d.chained = static_cast<uint32_t>(early_stats[5] & 0xFFFFFFFFu);      // (k_link_desc2's atomicMax on the low word; the high word counts the records)
            // a handful of chained records (44 of two million on the headline index, where monomorphic sites meet) are not worth the ring
            // headroom every walker then keeps free: below one record in a thousand the descriptors are linked again without chains
            // (0.1 ms), unless GBWT_HIP_CHAINS asked for them
            if (d.chained != 0 && std::getenv("GBWT_HIP_CHAINS") == nullptr && (early_stats[5] >> 32) * 1000 < n_records) {
                HIP_CHECK(hipMemset(d_stats + 5, 0, 2 * sizeof(uint64_t)));
                launch_link_desc2(d, ix.desc2.as<uint4>(), gather_limit, h.bidirectional ? 0x100u : 0u, reinterpret_cast<uint32_t *>(d_stats + 5), nullptr);
                d.chained = 0;
            }
            if (trace.on) std::fprintf(stderr, "[open] records with a chained step: %llu of %llu (up to %u nodes per iteration)\n",
                                       static_cast<unsigned long long>(early_stats[5] >> 32), static_cast<unsigned long long>(n_records), d.chained);
            const uint64_t longest = early_stats[0];
            generic_records = early_stats[4];
            generic_count = generic_records;
            const bool full_width_now = gather_limit == 0 || longest >= gather_limit;
            if (full_width_now) {
                ix.cblocks.reserve(n_blocks * 2 * sizeof(uint4));
                HIP_CHECK(hipMemsetAsync(ix.cblocks.ptr, 0, 2 * sizeof(uint4), nullptr));
            }
            if (n_blocks > 1) launch_fill_two_step_blocks(d, full_width_now ? ix.cblocks.as<uint4>() : nullptr, gather_limit ? ix.gblocks.as<uint4>() : nullptr, nullptr);
            if (full_width_now) d.cblocks = ix.cblocks.as<uint4>();
            trace.mark("two-step descriptors + blocks");
            launch_link_lookahead2(d, ix.desc2.as<uint4>(), counts.as<uint32_t>(), std::max<uint32_t>(1, (hops + 1) / 2), nullptr);
        }
        // LF tables for the class 0 records, while they fit the budget (none in an index of outdegree <= 2 whose streams are all lean: nothing to count)
        d.tables = nullptr;
        d.wtables = nullptr;
        d.wtables_deep = nullptr;
        if (!for_extract) generic_records = generic_count;
        if (n_records > 0 && generic_records > 0) {
            DeviceBuffer positions, sigmas, table_base, edge_base, edges;
            positions.reserve(n_records * sizeof(uint64_t)); sigmas.reserve(n_records * sizeof(uint64_t));
            table_base.reserve((n_records + 1) * sizeof(uint64_t)); edge_base.reserve((n_records + 1) * sizeof(uint64_t));
            launch_table_counts(d, positions.as<uint64_t>(), sigmas.as<uint64_t>(), nullptr);
            const size_t tb = scan_temp_bytes(n_records);
            scan_tmp.reserve(std::max<size_t>(tb, 16));
            launch_scan(positions.as<uint64_t>(), table_base.as<uint64_t>(), n_records, scan_tmp.ptr, tb, nullptr);
            launch_scan(sigmas.as<uint64_t>(), edge_base.as<uint64_t>(), n_records, scan_tmp.ptr, tb, nullptr);
            uint64_t total_positions = 0, total_edges = 0;
            HIP_CHECK(hipMemcpy(&total_positions, table_base.as<uint64_t>() + n_records, sizeof(uint64_t), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(&total_edges, edge_base.as<uint64_t>() + n_records, sizeof(uint64_t), hipMemcpyDeviceToHost));
            // 16 + 16 + 64 bytes per position of such a record (LF table, walk table, deep walk table): a quarter of the device's memory
            // while half of what is free covers it (72 GiB of an MI355X's 288), or what GBWT_HIP_TABLE_BYTES says
            uint64_t budget = uint64_t(16) << 30;
            {
                size_t free_bytes = 0, total_bytes = 0;
                if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) budget = std::min<uint64_t>(total_bytes / 4, free_bytes / 2);
                else (void)hipGetLastError();
            }
            if (const char *v = std::getenv("GBWT_HIP_TABLE_BYTES")) budget = std::strtoull(v, nullptr, 10);
            if (total_positions > 0 && total_positions < 0xFFFFFFFFull && total_positions * sizeof(uint4) <= budget) {
                ix.tables.reserve(total_positions * sizeof(uint4));
                edges.reserve(std::max<uint64_t>(total_edges, 1) * sizeof(uint2));
                d.tables = ix.tables.as<uint4>();
                launch_fill_tables(d, ix.desc_raw.as<uint4>(), table_base.as<uint64_t>(), edge_base.as<uint64_t>(), ix.tables.as<uint4>(),
                                   edges.as<uint2>(), nullptr);
                HIP_CHECK(hipDeviceSynchronize());
                HIP_CHECK(hipGetLastError());
                // walk tables next to them while both fit (GBWT_HIP_WALK_TABLES=0: walks take one plain table step at a time)
                const char *wt = std::getenv("GBWT_HIP_WALK_TABLES");
                if (for_extract && 2 * total_positions * sizeof(uint4) <= budget && n_records <= 0x7FFFFFFFull && !(wt && std::atoi(wt) == 0)) {
                    ix.wtables.reserve(total_positions * sizeof(uint4));
                    launch_fill_wtables(d, ix.wtables.as<uint4>(), nullptr);
                    HIP_CHECK(hipDeviceSynchronize());
                    HIP_CHECK(hipGetLastError());
                    d.wtables = ix.wtables.as<uint4>();
                    // ... and the deep walk tables (seven steps per 64-byte entry) while all three fit (GBWT_HIP_DEEP_TABLES=0: one step per load)
                    const char *deep = std::getenv("GBWT_HIP_DEEP_TABLES");
                    ix.table_positions = total_positions;
                    if (6 * total_positions * sizeof(uint4) <= budget && !(deep && std::atoi(deep) == 0)) {
                        ix.wtables_deep.reserve(total_positions * 4 * sizeof(uint4));
                        const char *compact = std::getenv("GBWT_HIP_COMPACT_TABLES");      // 0: seven-step entries only (round 3)
                        launch_fill_wtables_deep(d, ix.wtables_deep.as<uint4>(), !(compact && std::atoi(compact) == 0), nullptr);
                        HIP_CHECK(hipDeviceSynchronize());
                        HIP_CHECK(hipGetLastError());
                        d.wtables_deep = ix.wtables_deep.as<uint4>();
                    }
                }
            }
        }
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipGetLastError());
    }
    trace.mark("tables");
    uint64_t hs[8];
    HIP_CHECK(hipMemcpy(hs, d_stats, sizeof(hs), hipMemcpyDeviceToHost));
    HIP_CHECK(hipGetLastError());
    ix.stats.max_record_len = hs[0];
    ix.stats.max_outdegree = hs[1];
    if (ix.caps & GBWT_HIP_OPEN_EXTRACT) ix.slow_records = hs[6] & 0xFFFFFFFFull;
    if (hs[2] != 0) throw InvalidData("BWT: record without a readable outdegree");
    if (hs[0] >= (uint64_t(1) << 32)) throw Unsupported("a record with 2^32 or more positions is not supported (u32 offsets on device)");
    d.max_walk = hs[3];
    // the endmarker: record 0 decompressed by the loader (host_index.cpp; the reference's GBWT::load does it on the CPU as well,
    // src/gbwt.rs:413-414) -- a single lane of the GPU took 6 ms for the 10 000 runs of the headline index, the host 0.1
    const std::vector<std::pair<uint32_t, uint32_t>> endmarker =
        early_endmarker ? *early_endmarker : decompress_endmarker(h, std::max<uint64_t>(hs[0], 1));
    const uint64_t end_len = endmarker.size();
    ix.endmarker.reserve(std::max<uint64_t>(end_len, 1) * sizeof(uint2));
    static_assert(sizeof(std::pair<uint32_t, uint32_t>) == sizeof(uint2), "pairs are uploaded as uint2");
    if (end_len > 0) HIP_CHECK(hipMemcpy(ix.endmarker.ptr, endmarker.data(), end_len * sizeof(uint2), hipMemcpyHostToDevice));
    d.endmarker = ix.endmarker.as<uint2>();
    d.n_endmarker = end_len;
    trace.mark("endmarker");
    // Lengths of all sequences and the sequence samples (GBWT_HIP_SEQ_LEN=0 skips both; extractions then go through the
    // pool of chained blocks and walk every sequence from one end).
    d.seq_len = nullptr;
    const char *want_len = std::getenv("GBWT_HIP_SEQ_LEN");
    if ((ix.caps & GBWT_HIP_OPEN_EXTRACT) != 0 && h.sequences > 0 && !(want_len && std::atoi(want_len) == 0)) {
        const auto t_samples = std::chrono::steady_clock::now();
        ix.seq_len.reserve(h.sequences * sizeof(uint32_t));
        // 16 bytes per sample.  Large indexes: about every 2 048 nodes (1 024 .. 4 096 measure within 3 % of each other on the
        // headline); smaller ones get shorter intervals, down to 64 nodes, so that an extraction still has enough
        // walkers to fill the GPU -- about four million samples per index at most.  GBWT_HIP_SAMPLE_INTERVAL=0: none.
        const uint64_t all_nodes = h.size >= h.sequences ? h.size - h.sequences : 0;
        uint32_t interval = 64;
        while (interval < 2048 && (all_nodes >> 22) > interval) interval *= 2;
        // ... and at most about four thousand per sequence: the chain of one sequence is followed by ONE lane when the samples are put
        // together (k_chase), a microsecond per hop -- ninety haplotypes of two million nodes at 128 nodes per hop were 23 000 hops,
        // 2 x 23 ms of a 35 ms sampling pass (profiles/r03_open_c4.txt)
        while (interval < 2048 && ((all_nodes / h.sequences) >> 12) > interval) interval *= 2;
        // walkers on deep walk tables take seven steps (up to fourteen nodes) per load: segments of 64 nodes would be five loads, a
        // fifth of them past the end of the segment (config C5: 265 G LF-steps/s at 64 nodes, 359 G from 128 up)
        // ... where most of the index is table records, and as long as about 64 000 walkers are left: the 7-allele chain of DESIGN section 4 walks
        // at 331 G LF-steps/s with 128-node segments and at 407 G with 512; C5 at 339 and 357)
        if (d.wtables_deep != nullptr && 3 * ix.table_positions >= h.size)   // (a star site is a table record and a unary one: half of the positions)
            while (interval < 512 && (interval < 128 || (all_nodes >> 17) > interval)) interval *= 2;
        // FINE SAMPLES, STRIDED WALKERS (round 4).  The numbers above are what a batch the size of the whole index wants: about 2 300 nodes per
        // walker on the headline, 34 000 workgroups of half a millisecond each.  A batch of an eighth of the paths -- one rank of eight --
        // then has 4 200 workgroups for 2 048 places, and a kernel that is two workgroup lifetimes long whatever it does (1.09 ms for an
        // eighth of 4.2 ms of work: profiles/r04_shard_probe.txt).  So the largest indexes keep a sample every 512 nodes, and an
        // extraction starts a walker at every `stride`-th of them, by the size of the batch (gbwt_hip_extract_device): strides of 4 or 5
        // for the headline batch (the segments it always had), 2 for an eighth of it (0.89 ms).  16 bytes per sample: 371 MB instead of 69.
        ix.sample_coarse = 1;
        if (const char *v = std::getenv("GBWT_HIP_SAMPLE_INTERVAL")) interval = static_cast<uint32_t>(std::max(0, std::atoi(v)));
        else if (interval >= 1024 && d.chained == 0) {
            const char *c = std::getenv("GBWT_HIP_SAMPLE_COARSE");
            const uint32_t coarse = c ? static_cast<uint32_t>(std::max(1, std::atoi(c))) : interval / 512;
            ix.sample_coarse = coarse; interval = std::max(64u, interval / coarse);
        }
        const bool sampled = interval >= 8;
        // Without samples an extraction fills every row from both ends, which needs the proof that sequence 2k + 1 is
        // sequence 2k reversed (fingerprints); with samples nothing does, and the counting walk is twice as fast
        // without them (0.42 -> 0.19 s on the headline index).  GBWT_HIP_ORIENTATION_CHECK=1 forces the proof.
        const char *force = std::getenv("GBWT_HIP_ORIENTATION_CHECK");
        const bool want_pairs = h.bidirectional && h.sequences % 2 == 0 && (!sampled || (force && std::atoi(force) != 0));
        DeviceBuffer prints;
        if (want_pairs) prints.reserve(h.sequences * 2 * sizeof(uint64_t));
        HIP_CHECK(hipMemset(d_stats, 0, 2 * sizeof(uint64_t)));
        uint32_t *d_flags = reinterpret_cast<uint32_t *>(d_stats);   // [0] = a length overflowed, [2] = a pair does not match
        // Samples by checkpoint sampling (open_walks.hip): no sequence is walked from end to end.  GBWT_HIP_SERIAL_SAMPLES=1 (and
        // indexes it cannot number with 32 bits, or whose records send walks in circles) take the walk of every sequence instead.
        const char *serial = std::getenv("GBWT_HIP_SERIAL_SAMPLES");
        bool by_checkpoints = sampled && !want_pairs && !(serial && std::atoi(serial) != 0) && checkpoint_samples(ix, interval, d_flags, trace);
        if (by_checkpoints) trace.mark("  temporaries freed");
        // serial: with samples and without fingerprints, lengths and samples come out of ONE walk (pooled samples, placed afterwards)
        DeviceBuffer pool, tags;
        uint64_t pooled = 0;
        const uint64_t pool_capacity = all_nodes / std::max<uint32_t>(interval, 1) + 2 * h.sequences + 1024;
        const char *two = std::getenv("GBWT_HIP_TWO_PASS_OPEN");
        bool one_walk = !by_checkpoints && sampled && !want_pairs && h.sequences <= 0xFFFFFFFFull && !(two && std::atoi(two) != 0);
        if (!by_checkpoints) d.cblocks = ensure_cblocks(&ix);   // the walks of every sequence below step on the full-width blocks (quiet_walk); still single-threaded here
        if (one_walk) {
            pool.reserve(pool_capacity * sizeof(uint4)); tags.reserve(pool_capacity * sizeof(uint2));
            HIP_CHECK(hipMemset(d_stats + 2, 0, sizeof(uint64_t)));
            HIP_CHECK(hipMemsetAsync(d_flags + 1, 0, sizeof(uint32_t), nullptr));   // the step budget the walks of ONE pass share (walk_loops.hpp: quiet_walk)
            launch_lengths_and_samples(d, interval, ix.seq_len.as<uint32_t>(), pool.as<uint4>(), tags.as<uint2>(), d_stats + 2, pool_capacity, d_flags, nullptr);
            HIP_CHECK(hipMemcpy(&pooled, d_stats + 2, sizeof(uint64_t), hipMemcpyDeviceToHost));
            if (pooled > pool_capacity) one_walk = false;            // cannot happen (one sample per interval + one per sequence): walk twice
        }
        if (!one_walk && !by_checkpoints) {
            HIP_CHECK(hipMemsetAsync(d_flags + 1, 0, sizeof(uint32_t), nullptr));
            launch_sequence_lengths(d, ix.seq_len.as<uint32_t>(), want_pairs ? prints.as<uint64_t>() : nullptr, d_flags, nullptr);
        }
        if (want_pairs)
            launch_check_orientation_pairs(ix.seq_len.as<uint32_t>(), prints.as<uint64_t>(), h.sequences / 2, d_flags + 2, nullptr);
        uint32_t flags[4] = {0, 0, 0, 0};
        HIP_CHECK(hipMemcpy(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost));
        HIP_CHECK(hipGetLastError());
        if (flags[0] & 2u) throw InvalidData("BWT: a sequence takes more LF steps than the records hold positions: the index is not consistent");
        if (!flags[0]) {
            d.seq_len = ix.seq_len.as<uint32_t>();
            ix.orientation_pairs = want_pairs && !flags[2];
            {   // all sequences of one length?  (haplotypes over one reference frame: the headline's shape)
                std::vector<uint32_t> &lens = ix.host_seq_len;
                lens.resize(h.sequences);
                HIP_CHECK(hipMemcpy(lens.data(), ix.seq_len.ptr, h.sequences * sizeof(uint32_t), hipMemcpyDeviceToHost));
                const auto mm = std::minmax_element(lens.begin(), lens.end());
                ix.uniform_len = (*mm.first == *mm.second) ? *mm.first : 0u;
            }
            // Sequence samples: where every sequence is about every `interval` nodes, so that extractions
            // can fill a row with many walkers at once.
            if (sampled && !by_checkpoints) {
                DeviceBuffer counts, scan_tmp;
                counts.reserve(h.sequences * sizeof(uint64_t));
                ix.sample_base.reserve((h.sequences + 1) * sizeof(uint64_t));
                launch_sample_counts(ix.seq_len.as<uint32_t>(), h.sequences, interval, counts.as<uint64_t>(), nullptr);
                const size_t tb = scan_temp_bytes(h.sequences);
                scan_tmp.reserve(std::max<size_t>(tb, 16));
                launch_scan(counts.as<uint64_t>(), ix.sample_base.as<uint64_t>(), h.sequences, scan_tmp.ptr, tb, nullptr);
                uint64_t total_samples = 0;
                HIP_CHECK(hipMemcpy(&total_samples, ix.sample_base.as<uint64_t>() + h.sequences, sizeof(uint64_t), hipMemcpyDeviceToHost));
                ix.samples.reserve(std::max<uint64_t>(total_samples, 1) * sizeof(uint4));
                if (one_walk) launch_place_samples(ix.seq_len.as<uint32_t>(), pool.as<uint4>(), tags.as<uint2>(), pooled, ix.sample_base.as<uint64_t>(), h.sequences,
                                                   ix.samples.as<uint4>(), nullptr);
                else launch_record_samples(d, ix.sample_base.as<uint64_t>(), interval, ix.samples.as<uint4>(), nullptr);
                HIP_CHECK(hipDeviceSynchronize());
                HIP_CHECK(hipGetLastError());
                ix.times.samples = total_samples;
                std::vector<uint64_t> base(h.sequences + 1);
                HIP_CHECK(hipMemcpy(base.data(), ix.sample_base.ptr, base.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
                note_sample_counts(ix, base);
            }
            if (sampled) {
                d.samples = ix.samples.as<uint4>();
                d.sample_base = ix.sample_base.as<uint64_t>();
                d.sample_interval = interval;
            }
        }
        trace.mark("lengths + samples");
        ix.times.checkpoint_sampling = by_checkpoints ? 1u : 0u;
        ix.times.sample_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_samples).count();
    }
}

// Takes ownership of `ix`; anything thrown on the way (corrupt sizes, HIP failures, allocation failures) destroys it and is
// turned into a status by the guard of the calling entry point.
gbwt_hip_status open_common(std::unique_ptr<gbwt_hip_index> ix, gbwt_hip_index **out, std::chrono::steady_clock::time_point t_open) {
    const auto t_parsed = std::chrono::steady_clock::now();
    fill_stats(*ix);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(GBWT_HIP_NO_DEVICE, "no HIP device available (libgbwt_hip has no CPU fallback)");
    OpenTrace trace;
    if (ix->host.pending && ix->record_bytes_uploaded) {
        // The loader still has work in the background (the record bytes into the host image, the node labels) and the GFA tables need
        // both: a thread waits for it and builds + uploads them while this one runs the device passes, which read nothing of either
        // (the record bytes are on the device already; record 0 is decompressed here, from the mapping, before the thread starts).
        const std::vector<std::pair<uint32_t, uint32_t>> endmarker = decompress_endmarker(ix->host, ix->host.sequences + 1);
        std::exception_ptr tail_failure;
        gbwt_hip_index *raw = ix.get();
        std::thread tail([raw, &tail_failure]() {
            try {
                HIP_CHECK(hipSetDevice(raw->device));
                const bool trace_tail = std::getenv("GBWT_HIP_TRACE_OPEN") != nullptr;
                const auto t0 = std::chrono::steady_clock::now();
                raw->host.finish();
                const auto t1 = std::chrono::steady_clock::now();
                if (raw->caps & GBWT_HIP_OPEN_GFA) upload_label_lengths(*raw);
                if (trace_tail) std::fprintf(stderr, "[open] (next to the device passes: the loader's background decodes %8.3f ms, GFA tables %8.3f ms)\n",
                                             std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
            } catch (...) { tail_failure = std::current_exception(); }
        });
        try { upload(*ix, &endmarker); } catch (...) { tail.join(); throw; }
        tail.join();
        if (tail_failure) std::rethrow_exception(tail_failure);
        trace.mark("(device passes + GFA tables)");
    } else {
        upload(*ix, nullptr);
        ix->host.finish();                // the loader's background work, if any: needed from here on
        if (ix->caps & GBWT_HIP_OPEN_GFA) upload_label_lengths(*ix);
    }
    // the sizes of every path's GFA line, once (gfa.hip): needs the device passes (samples, descriptors) AND the GFA tables (label lengths)
    if (ix->caps & GBWT_HIP_OPEN_GFA) {
        mask_label_lengths(*ix);          // (needs both as well: the label lengths from that thread, the record bytes and starts from this one)
        fill_line_cache_at_open(*ix);
        trace.mark("line sizes of every path");
    }
    // A handle that was NOT opened for search and whose walks never leave the descriptors and rank blocks (no record takes the generic
    // decoder, lengths and samples are there, the full-width blocks are built or not needed) gives its raw descriptors back: 64 bytes per
    // record -- 14 of config 4's 65 GB.  What still needs them -- the pool-output walk modes, blocks built on first need -- is refused
    // on such a handle with GBWT_HIP_UNSUPPORTED (gbwt_hip_extract_part_device, ensure_cblocks).
    {
        gbwt_hip_index &x = *ix;
        if (!(x.caps & GBWT_HIP_OPEN_SEARCH) && x.slow_records == 0 && x.dev.seq_len != nullptr && x.dev.samples != nullptr && x.max_samples > 0 &&
            x.dev.tables == nullptr && (x.dev.cblocks != nullptr || x.packed_blocks)) {
            HIP_CHECK(hipDeviceSynchronize());
            x.desc_raw.release();
            x.dev.desc_raw = nullptr;
            x.lean_extract = true;
            // ... and with them the one-step walk descriptors and the plain rank blocks (64 B per record + 16 B per 64 positions): all that read
            // them after the open were the catch-up steps of k_walk_direct, which on such a handle step on the two-step descriptors and the
            // packed half-blocks instead (WalkArgs::catch_up == 2), and the walk modes that are refused here anyway
            if (x.packed_blocks && x.dev.gblocks != nullptr) {
                x.desc.release(); x.blocks.release();
                x.dev.desc = nullptr; x.dev.blocks = nullptr;
            }
        }
    }
    const auto t_done = std::chrono::steady_clock::now();
    const auto ms = [](std::chrono::steady_clock::duration d) { return std::chrono::duration<double, std::milli>(d).count(); };
    ix->times.parse_ms = ms(t_parsed - t_open);
    ix->times.total_ms = ms(t_done - t_open);
    ix->times.upload_ms = ms(t_done - t_parsed) - ix->times.sample_ms;
    *out = ix.release();
    return GBWT_HIP_OK;
}

}  // namespace

extern "C" {

const char *gbwt_hip_last_error(void) { return last_error_slot().c_str(); }

int gbwt_hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

gbwt_hip_status gbwt_hip_parse_file(const char *path, gbwt_hip_stats *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!path || !out) return fail(GBWT_HIP_BAD_ARGUMENT, "null argument");
    gbwt_hip_index tmp;
    tmp.host = load_index_file(path);
    fill_stats(tmp);
    *out = tmp.stats;
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

static uint32_t normalised_caps(uint32_t flags) { return (flags & GBWT_HIP_OPEN_GFA) ? (flags | GBWT_HIP_OPEN_EXTRACT) : flags; }   // the lines are formatted from extracted rows

gbwt_hip_status gbwt_hip_open_file(const char *path, int device, gbwt_hip_index **out) { return gbwt_hip_open_file_flags(path, device, GBWT_HIP_OPEN_ALL, out); }

gbwt_hip_status gbwt_hip_open_file_flags(const char *path, int device, uint32_t flags, gbwt_hip_index **out) {
    GBWT_HIP_GUARD_BEGIN
    if (!path || !out) return fail(GBWT_HIP_BAD_ARGUMENT, "null argument");
    if (flags == 0 || (flags & ~uint32_t(GBWT_HIP_OPEN_ALL)) != 0) return fail(GBWT_HIP_BAD_ARGUMENT, "flags: a non-empty set of GBWT_HIP_OPEN_EXTRACT | _SEARCH | _GFA");
    *out = nullptr;
    const auto t_open = std::chrono::steady_clock::now();
    std::unique_ptr<gbwt_hip_index> ix(new gbwt_hip_index);
    ix->device = device;
    ix->caps = normalised_caps(flags);
    // The record bytes start for the device as soon as the loader knows where they are, next to its decoding of the record starts
    // (9 ms of staged copy next to 6.7 ms of Elias-Fano decode on the headline index, one after the other until round 3).
    struct EarlyCopy {
        std::thread worker;
        hipError_t result = hipSuccess;
        ~EarlyCopy() { if (worker.joinable()) worker.join(); }
    } early;
    gbwt_hip_index *raw = ix.get();
    static const bool lazy_host_records = [] { const char *e = std::getenv("GBWT_HIP_LAZY_HOST_RECORDS"); return !(e && e[0] == '0'); }();
    load_index_file_into(path, ix->host, true, [raw, &early](HostIndex &h) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0 || hipSetDevice(raw->device) != hipSuccess) { (void)hipGetLastError(); return; }   // open_common says so
        const uint64_t bytes = h.record_bytes_len();
        if (bytes < (uint64_t(1) << 20)) return;              // not worth a thread
        raw->data.reserve(bytes + DATA_PAD);
        HIP_CHECK(hipMemset(raw->data.ptr, 0, bytes + DATA_PAD));
        raw->record_bytes_uploaded = true;
        h.starts_on_device = true;                            // ... and the record starts are decoded there, from the words of the file (decode_starts_on_device)
        const uint8_t *src = h.record_bytes();
        void *dst = raw->data.ptr;
        const int dev = raw->device;
        // (one hipMemcpy from the mapping: 9 ms = 6.7 GB/s for the headline's 60 MB.  Measured and not kept, rounds 3 and 4: four threads
        // with slices, registered pages, and -- round 4 -- four threads staging 4 MiB chunks through pinned buffers and streams of their
        // own, made once per process in the background: 16-25 ms, and a first extraction that waited 370 ms behind the pinned
        // allocations; NOTEBOOK.md, round 4)
        // ... for the 60 MB of the headline index.  The 1.5 GB of an HPRC-sized one are cut into slices that several threads copy side by
        // side (from 256 MB: 4): profiles/r05_c4_open_trace.txt
        unsigned slices = bytes >= (uint64_t(256) << 20) ? 4u : 1u;
        early.worker = std::thread([&early, src, dst, bytes, dev, slices]() {
            early.result = hipSetDevice(dev);
            if (early.result != hipSuccess) return;
            const bool trace = std::getenv("GBWT_HIP_TRACE_OPEN") != nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            struct Note { bool on; std::chrono::steady_clock::time_point t0; uint64_t bytes; unsigned slices;
                          ~Note() { if (on) std::fprintf(stderr, "[open] (record bytes to the device, %u slice(s), next to the decodes) %8.3f ms for %.1f MB\n", slices,
                                                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), bytes / 1e6); } } note{trace, t0, bytes, slices};
            if (slices <= 1) { early.result = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice); return; }
            std::vector<hipError_t> results(slices, hipSuccess);
            std::vector<std::thread> pool;
            for (unsigned k = 0; k < slices; k++)
                pool.emplace_back([&results, k, slices, src, dst, bytes, dev]() {
                    const uint64_t lo = (bytes / 4096 * k / slices) * 4096, hi = k + 1 == slices ? bytes : (bytes / 4096 * (k + 1) / slices) * 4096;
                    results[k] = hipSetDevice(dev);
                    if (results[k] == hipSuccess && hi > lo) results[k] = hipMemcpy(static_cast<char *>(dst) + lo, src + lo, hi - lo, hipMemcpyHostToDevice);
                });
            for (auto &t : pool) t.join();
            for (hipError_t r : results) if (r != hipSuccess) early.result = r;
        });
    }, lazy_host_records);
    if (early.worker.joinable()) {
        // the starts are decoded by now and the copy of the record bytes has a few milliseconds to go: the starts go out under them
        fill_stats(*ix);
        OpenTrace trace;
        try { if (ix->host.starts_on_device) decode_starts_on_device(*ix); else upload_starts(*ix); } catch (...) { early.worker.join(); throw; }
        trace.mark("(record starts on the device)");
        early.worker.join();
        trace.mark("(waited for the record bytes)");
    }
    if (early.result != hipSuccess) throw HipError{early.result, "hipMemcpy (record bytes, early)"};
    return open_common(std::move(ix), out, t_open);
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_open_records(const uint8_t *data, uint64_t data_len, const uint64_t *starts, uint64_t n_records,
                                      uint64_t alphabet_offset, uint64_t alphabet_size, uint64_t n_sequences, uint64_t size,
                                      int bidirectional, int device, gbwt_hip_index **out) {
    return gbwt_hip_open_records_flags(data, data_len, starts, n_records, alphabet_offset, alphabet_size, n_sequences, size, bidirectional, device, GBWT_HIP_OPEN_ALL, out);
}

gbwt_hip_status gbwt_hip_open_records_flags(const uint8_t *data, uint64_t data_len, const uint64_t *starts, uint64_t n_records,
                                            uint64_t alphabet_offset, uint64_t alphabet_size, uint64_t n_sequences, uint64_t size,
                                            int bidirectional, int device, uint32_t flags, gbwt_hip_index **out) {
    GBWT_HIP_GUARD_BEGIN
    if (!out || (data_len && !data) || (n_records && !starts)) return fail(GBWT_HIP_BAD_ARGUMENT, "null argument");
    if (flags == 0 || (flags & ~uint32_t(GBWT_HIP_OPEN_ALL)) != 0) return fail(GBWT_HIP_BAD_ARGUMENT, "flags: a non-empty set of GBWT_HIP_OPEN_EXTRACT | _SEARCH | _GFA");
    *out = nullptr;
    const auto t_open = std::chrono::steady_clock::now();
    std::unique_ptr<gbwt_hip_index> ix(new gbwt_hip_index);
    ix->device = device;
    ix->caps = normalised_caps(flags);
    ix->host = index_from_records(data, data_len, starts, n_records, alphabet_offset, alphabet_size, n_sequences, size,
                                  bidirectional != 0);
    return open_common(std::move(ix), out, t_open);
    GBWT_HIP_GUARD_END
}

void gbwt_hip_close(gbwt_hip_index *index) { delete index; }

gbwt_hip_status gbwt_hip_get_stats(const gbwt_hip_index *index, gbwt_hip_stats *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!index || !out) return fail(GBWT_HIP_BAD_ARGUMENT, "null argument");
    *out = index->stats;
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    GBWT_HIP_GUARD_BEGIN
    if (!free_bytes || !total_bytes) return fail(GBWT_HIP_BAD_ARGUMENT, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(GBWT_HIP_NO_DEVICE, "no HIP device available");
    try {
        HIP_CHECK(hipSetDevice(device));
        size_t f = 0, t = 0;
        HIP_CHECK(hipMemGetInfo(&f, &t));
        *free_bytes = f; *total_bytes = t;
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_memory_usage(const gbwt_hip_index *index, const gbwt_hip_workspace *ws, gbwt_hip_memory *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!index || !out || (ws && ws->index != index)) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace");
    *out = gbwt_hip_memory{};
    const auto sum = [](std::initializer_list<const DeviceBuffer *> buffers) { uint64_t b = 0; for (const DeviceBuffer *q : buffers) b += q->bytes; return b; };
    const gbwt_hip_index &ix = *index;
    out->index_device_bytes = sum({&ix.data, &ix.starts, &ix.endmarker, &ix.desc, &ix.desc_raw, &ix.block_base, &ix.blocks, &ix.desc2, &ix.gblocks, &ix.tables,
                                   &ix.wtables, &ix.wtables_deep, &ix.seq_len, &ix.samples, &ix.sample_base, &ix.label_len, &ix.seg_of, &ix.seg_start,
                                   &ix.seg_name_off, &ix.seg_names, &ix.seg_seq_len, &ix.node_real, &ix.line_prefix[0], &ix.line_prefix[1], &ix.line_prefix[2],
                                   &ix.line_prefix_off[0], &ix.line_prefix_off[1], &ix.line_prefix_off[2], &ix.line_fragment, &ix.lc_chunk_first, &ix.lc_text, &ix.lc_path});
    // (the full-width two-step blocks: at open, or by the first request that needs them -- the atomic says when they are there)
    if (ix.dev.cblocks != nullptr || ix.lazy_cblocks.load(std::memory_order_acquire) != nullptr) out->index_device_bytes += ix.cblocks.bytes;
    const HostIndex &h = ix.host;
    out->index_host_bytes = (h.records_made() ? h.data.size() + h.starts.size() * sizeof(uint64_t) : 0) + h.da_samples.size() * sizeof(uint64_t) + h.path_names.size() * sizeof(PathName) +
                            h.sample_names.bytes.size() + h.contig_names.bytes.size() + h.sequences_labels.bytes.size() + h.sequences_labels.offsets.size() * sizeof(uint64_t) +
                            h.segment_names.bytes.size() + h.segment_names.offsets.size() * sizeof(uint64_t) + h.segment_starts.size() * sizeof(uint64_t) +
                            ix.sample_counts.size() * sizeof(uint32_t);
    if (ws) {
        out->workspace_device_bytes = sum({&ws->seq_ids, &ws->lengths, &ws->offsets, &ws->head, &ws->pool, &ws->next, &ws->counters, &ws->nodes, &ws->scan_temp,
                                           &ws->order_keys, &ws->order_rows, &ws->order_counts, &ws->order_level, &ws->order_temp, &ws->in_a, &ws->in_b, &ws->out_a,
                                           &ws->out_valid, &ws->follow_off, &ws->gfa_a, &ws->gfa_b, &ws->gfa_c, &ws->gfa_text, &ws->gfa_text2, &ws->gfa_valid, &ws->gfa_chunk_first,
                                           &ws->gfa_chunks, &ws->gfa_plan});
        out->rows_bytes = ws->nodes.bytes;
        out->rows_chunks = ws->nodes.chunks.size();
        out->text_bytes = ws->gfa_text.bytes + ws->gfa_text2.bytes;
    }
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_get_open_times(const gbwt_hip_index *index, gbwt_hip_open_times *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!index || !out) return fail(GBWT_HIP_BAD_ARGUMENT, "null argument");
    *out = index->times;
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

}  // extern "C"
