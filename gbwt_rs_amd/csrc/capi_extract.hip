// capi_extract.hip -- the C ABI of libgbwt_hip.so (include/gbwt_hip.h), part 2 of 3: workspaces and batched path extraction
// (gbwt_hip_extract*: SequenceIter / GBZ::path, src/gbwt.rs:253-261, 557-568, src/gbz.rs:461-466), the copies of results to the host, and
// the per-row checksums.  No CPU implementation of any compute entry point.
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

extern "C" {

gbwt_hip_status gbwt_hip_workspace_create(const gbwt_hip_index *index, gbwt_hip_workspace **out) {
    GBWT_HIP_GUARD_BEGIN
    if (!index || !out) return fail(GBWT_HIP_BAD_ARGUMENT, "null argument");
    *out = nullptr;
    std::unique_ptr<gbwt_hip_workspace> ws(new gbwt_hip_workspace);
    ws->nodes.may_spread = true;
    ws->nodes.policy = vmm_policy_from_env();
    ws->knobs = ExtractKnobs::from_env();
    ws->index = index;
    HIP_CHECK(hipSetDevice(index->device));
    HIP_CHECK(hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking));
    for (auto &e : ws->ev) HIP_CHECK(hipEventCreate(&e));
    ws->counters.reserve(4 * sizeof(uint32_t));
    // optional overrides for experiments / tests (same meaning as gbwt_hip_workspace_tune)
    if (const char *v = std::getenv("GBWT_HIP_WALK_MODE")) { int m = std::atoi(v); if (m >= 0 && m <= 3) ws->walk_mode = static_cast<uint32_t>(m); }
    if (const char *v = std::getenv("GBWT_HIP_PATHS_PER_WAVE")) { int p = std::atoi(v); if (p >= 0 && p <= 64) ws->paths_per_wave = static_cast<uint32_t>(p); }
    if (const char *v = std::getenv("GBWT_HIP_SMALL_RECORD")) { long r = std::atol(v); if (r >= 0) ws->small_record = static_cast<uint32_t>(r); }
    *out = ws.release();
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

void gbwt_hip_workspace_destroy(gbwt_hip_workspace *ws) { delete ws; }

gbwt_hip_status gbwt_hip_workspace_tune(gbwt_hip_workspace *ws, uint32_t walk_mode, uint32_t paths_per_wave, uint32_t small_record) {
    GBWT_HIP_GUARD_BEGIN
    if (!ws || walk_mode > WALK_ONE_STEP || paths_per_wave > 64) return fail(GBWT_HIP_BAD_ARGUMENT, "bad tuning values");
    ws->walk_mode = walk_mode; ws->paths_per_wave = paths_per_wave; ws->small_record = small_record;
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

void *gbwt_hip_workspace_stream(gbwt_hip_workspace *ws) { return ws ? static_cast<void *>(ws->stream) : nullptr; }

gbwt_hip_status gbwt_hip_extract_device(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *seq_ids, uint64_t n,
                                        gbwt_hip_paths *out) {
    return gbwt_hip_extract_part_device(ix, ws, seq_ids, n, 0, 1, out);
}

gbwt_hip_status gbwt_hip_extract_part_device(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *seq_ids, uint64_t n,
                                             uint32_t part, uint32_t parts, gbwt_hip_paths *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!ix || !ws || ws->index != ix || !out) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace");
    if (n && !seq_ids) return fail(GBWT_HIP_BAD_ARGUMENT, "null seq_ids");
    if (parts == 0 || part >= parts) return fail(GBWT_HIP_BAD_ARGUMENT, "part must be < parts");
    if (!(ix->caps & GBWT_HIP_OPEN_EXTRACT)) return fail(GBWT_HIP_BAD_ARGUMENT, "the handle was not opened for extraction (GBWT_HIP_OPEN_EXTRACT)");
    // GBWT::sequence: id >= sequences -> no iterator (src/gbwt.rs:254-256).  "Not found" is a value here as well: such an id
    // gets an empty row (the kernels test the id), the rest of the batch is extracted; callers tell None from an empty
    // sequence by id < sequences.
    ws->extract_cached = false;
    ws->timed = false; ws->last_n = 0; ws->last_total = 0;   // what copy_result / path_sums / copy_path trust: set again only when this call succeeds
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        hipStream_t s = ws->stream;
        ws->seq_ids.reserve(std::max<uint64_t>(n, 1) * sizeof(uint64_t));
        ws->lengths.reserve(std::max<uint64_t>(n, 1) * sizeof(uint64_t));
        ws->offsets.reserve((n + 1) * sizeof(uint64_t));
        ws->head.reserve(std::max<uint64_t>(n, 1) * sizeof(uint32_t));
        size_t temp_bytes = n ? scan_temp_bytes(n) : 0;
        ws->scan_temp.reserve(std::max<size_t>(temp_bytes, 16));
        // Pool bound for distinct ids: all sequences together hold size - sequences nodes
        // (src/gbwt.rs:108-122), and every path wastes less than one block.
        const uint64_t all_nodes = ix->host.size >= ix->host.sequences ? ix->host.size - ix->host.sequences : 0;
        uint64_t pool_blocks = all_nodes / POOL_BLOCK_NODES + n + 1;
        HIP_CHECK(hipEventRecord(ws->ev[3], s));   // everything the extraction puts on the stream lies between ev[3] and ev[2]
        if (n) HIP_CHECK(hipMemcpyAsync(ws->seq_ids.ptr, seq_ids, n * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        const ExtractKnobs &knobs = ws->knobs;
        if (n && ix->dev.seq_len && ws->walk_mode == WALK_TWO_STEP && knobs.direct != 0) {
            // Lengths known: offsets first, then every lane writes into its row; in a bidirectional index two walkers per
            // sequence, one from each end.
            bool ids_valid = true;
            for (uint64_t k = 0; k < n && ids_valid; k++) ids_valid = seq_ids[k] < ix->host.sequences;
            // ONE PART OF EVERY ROW (gbwt_hip_extract_part_device; round 4): rows are cut where their walkers start anyway, at sequence samples.
            // An index without samples (GBWT_HIP_SAMPLE_INTERVAL=0, GBWT_HIP_SEGMENTS=0) cannot cut: the last part is the whole row there.
            const bool can_cut = ix->dev.samples != nullptr && ix->max_samples > 0 && knobs.segments != 0 && n <= 0x7FFFFFFFull;
            if (parts > 1 && !can_cut && part + 1 < parts) {
                HIP_CHECK(hipMemsetAsync(ws->offsets.ptr, 0, (n + 1) * sizeof(uint64_t), s));
                ws->nodes.reserve(sizeof(uint32_t));
                HIP_CHECK(hipEventRecord(ws->ev[0], s)); HIP_CHECK(hipEventRecord(ws->ev[1], s)); HIP_CHECK(hipEventRecord(ws->ev[2], s));
                HIP_CHECK(hipStreamSynchronize(s));
                ws->timed = true; ws->last_n = n; ws->last_total = 0;
                out->d_offsets = ws->offsets.as<uint64_t>(); out->d_nodes = ws->nodes.as<uint32_t>(); out->total = 0; out->n = n;
                return GBWT_HIP_OK;
            }
            const bool parted = parts > 1 && can_cut;
            // a walker per `stride` samples of a row (fine samples, gbwt_hip_open): about 2.9 million walkers for the biggest batches, no segments
            // shorter than two samples unless the batch is so small that it needs every walker it can get
            uint32_t stride = static_cast<uint32_t>(std::max(1, knobs.sample_stride));
            if (knobs.sample_stride < 0 && can_cut && ix->sample_coarse > 1) {
                uint64_t fine = 0;
                for (uint64_t k = 0; k < n; k++) if (seq_ids[k] < ix->host.sequences) fine += ix->sample_counts[seq_ids[k]];
                if (parted) fine /= parts;
                const uint64_t want = (fine + 1450000) / 2900000;
                stride = static_cast<uint32_t>(std::min<uint64_t>(2 * ix->sample_coarse, std::max<uint64_t>(want, fine >= 524288 ? 2 : 1)));
            }
            DeviceIndex strided = ix->dev; strided.sample_stride = stride;
            if (parted) { strided.sample_part = part; strided.sample_parts = parts; }
            // every row of one, known length (the headline's shape, config 5): no table of offsets is needed to walk, and the walk kernel
            // writes the one the caller gets (WalkArgs::uniform_len) -- nothing is launched in front of it
            const bool all_valid = ids_valid && ix->uniform_len != 0 && !parted;
            const bool fused_offsets = all_valid && knobs.fused_offsets != 0;
            // lengths, offsets and extremes: one launch for the batch sizes there are, else a memset and three
            if (!fused_offsets && !launch_row_offsets(strided, ws->seq_ids.as<uint64_t>(), n, ws->lengths.as<uint64_t>(), ws->offsets.as<uint64_t>(), ws->counters.as<uint32_t>(), s)) {
                HIP_CHECK(hipMemsetAsync(ws->counters.ptr, 0, 4 * sizeof(uint32_t), s));
                if (parted) launch_part_lengths(strided, ws->seq_ids.as<uint64_t>(), n, ws->lengths.as<uint64_t>(), ws->counters.as<uint32_t>(), s);
                else launch_gather_lengths(ix->dev.seq_len, ix->dev.n_sequences, ws->seq_ids.as<uint64_t>(), n, ws->lengths.as<uint64_t>(), ws->counters.as<uint32_t>(), s);
                launch_scan(ws->lengths.as<uint64_t>(), ws->offsets.as<uint64_t>(), n, ws->scan_temp.ptr, temp_bytes, s);
            }
            uint64_t total = 0;
            uint32_t extremes[2] = {0, 0};   // the longest row, ~(the shortest)
            // every row of the batch with the same number of samples (the forward sequences of haplotypes over one reference frame: the
            // headline's shape): walker w = segment * n + row, no order to compute -- and nothing the host has to ask the device for
            uint32_t common = ix->uniform_samples;
            if (can_cut && ids_valid && common == 0 && n != 0) {
                common = ix->sample_counts[seq_ids[0]];
                for (uint64_t k = 1; k < n && common != 0; k++) if (ix->sample_counts[seq_ids[k]] != common) common = 0;
            }
            // ROWS SIZED AFTER THE LAUNCH (round 4): a request that knows its walkers without the device (above) and finds rows in its
            // workspace does not wait for the total of its row lengths either -- 20 us of a round trip in front of a walk of 0.6 ms: the walk
            // is launched into the rows that are there with their size as `capacity`, every workgroup looks at offsets[n] first, and
            // if the rows were too small (the first request of its size) the host makes them and launches again.  One host wait per request.
            const uint64_t capacity = ws->nodes.bytes / sizeof(uint32_t);
            const bool defer = knobs.defer_total != 0 && !all_valid && can_cut && ids_valid && common != 0 && capacity != 0;
            if (all_valid) {
                // every row has the same, known length: total and extremes without a round trip to the device (the offsets are
                // still scanned there, behind which the walk is simply enqueued)
                total = n * static_cast<uint64_t>(ix->uniform_len);
                extremes[0] = ix->uniform_len; extremes[1] = ~ix->uniform_len;
            } else if (defer) {
                if (!ws->pinned_words) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ws->pinned_words), 4 * sizeof(uint64_t)));
                HIP_CHECK(hipMemcpyAsync(ws->pinned_words, ws->offsets.as<uint64_t>() + n, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
                total = capacity;                                  // (stands in until the wait at the end)
                extremes[0] = 1; extremes[1] = ~1u;                // (a row may have nodes: the walk is a segmented one)
            } else {
                HIP_CHECK(hipMemcpyAsync(&total, ws->offsets.as<uint64_t>() + n, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipMemcpyAsync(extremes, ws->counters.ptr, sizeof(extremes), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
            }
            const uint32_t max_len = extremes[0];
            ws->nodes.reserve(std::max<uint64_t>(total, 1) * sizeof(uint32_t));
            WalkArgs a{};
            a.seq_ids = ws->seq_ids.as<uint64_t>(); a.n = n;
            a.mode = ws->walk_mode;
            // many walkers per row when the index has sequence samples (GBWT_HIP_SEGMENTS=0: one walker per end instead)
            const bool segmented = ix->dev.samples != nullptr && max_len > 0 && n <= 0x7FFFFFFFull && ix->max_samples > 0 && knobs.segments != 0;   // (the walker order sorts 32-bit row numbers)
            // A lean handle (open_common) has given back the raw descriptors that a walker without a sample to start from reads when it arrives
            // on its first record (walk_loops.hpp: arrive): whole-row walkers -- GBWT_HIP_SEGMENTS=0, more than 2^31 - 1 rows -- are refused
            // there like the pool-output modes below, before anything is launched (rows of no nodes have no walker and are served).
            if (ix->lean_extract && !segmented && max_len > 0)
                return fail(GBWT_HIP_UNSUPPORTED, "this handle was opened for extraction only and has given its raw descriptors back: walks that do not start from sequence samples (GBWT_HIP_SEGMENTS=0, more than 2^31 - 1 rows) need GBWT_HIP_OPEN_ALL");
            // the samples lie where the sequences pass checkpoint records, so the number of segments of a row comes from its samples, not from its length
            a.segments = segmented ? (ix->max_samples + stride - 1) / stride : 0u;
            if (parted) a.segments = a.segments / parts + 1;          // (no row has more segments in one part)
            uint64_t walkers = ix->orientation_pairs ? 2 * n : n;
            const bool same_segments = segmented && ids_valid && common != 0;
            if (same_segments) {
                a.segments = (common + stride - 1) / stride;
                if (parted) a.segments = static_cast<uint32_t>(uint64_t(a.segments) * (part + 1) / parts - uint64_t(a.segments) * part / parts);   // (device_index.hpp: row_segments)
                walkers = static_cast<uint64_t>(a.segments) * n;   // every row has every segment: no order to compute
                a.sorted_rows = nullptr; a.level = nullptr; a.walkers = walkers;
            } else if (segmented) {
                // rows with different numbers of segments: walkers segment by segment over the rows that have the segment (rows sorted by their segment count)
                const size_t ob = walker_order_temp_bytes(n), sb = scan_temp_bytes(a.segments);
                ws->order_keys.reserve(2 * n * sizeof(uint32_t)); ws->order_rows.reserve(2 * n * sizeof(uint32_t));
                ws->order_counts.reserve(a.segments * sizeof(uint64_t)); ws->order_level.reserve((a.segments + 1ull) * sizeof(uint64_t));
                ws->order_temp.reserve(std::max<size_t>(std::max(ob, sb), 16));
                const uint32_t *sorted_rows = nullptr;
                launch_walker_order(strided, ws->seq_ids.as<uint64_t>(), n, a.segments, ws->order_keys.as<uint32_t>(), ws->order_rows.as<uint32_t>(),
                                    ws->order_counts.as<uint64_t>(), ws->order_level.as<uint64_t>(), ws->order_temp.ptr, ob, &sorted_rows, s);
                launch_scan(ws->order_counts.as<uint64_t>(), ws->order_level.as<uint64_t>(), a.segments, ws->order_temp.ptr, sb, s);
                HIP_CHECK(hipMemcpyAsync(&walkers, ws->order_level.as<uint64_t>() + a.segments, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                a.sorted_rows = sorted_rows; a.level = ws->order_level.as<uint64_t>(); a.walkers = walkers;
            }
            // many walkers: the walk is a throughput problem, full waves; few walkers (one or two per row): latency, and
            // about one workgroup per four SIMDs keeps every workgroup resident (32 KB of LDS each)
            a.paths_per_wave = ws->paths_per_wave ? ws->paths_per_wave
                               : segmented ? 64u : static_cast<uint32_t>(std::min<uint64_t>(64, std::max<uint64_t>(16, (walkers + 1023) / 1024)));
            a.helper_lanes = 64u;
            a.wide_addresses = knobs.wide_addresses ? 1u : 0u;
            a.ring_slots = knobs.ring_slots > 0 ? static_cast<uint32_t>(knobs.ring_slots) : (segmented ? 64u : 128u);   // many walkers: smaller rings, more workgroups per CU (7.4 vs 8.1 ms on the headline)
            // many walkers per CU: a helper that polls less leaves more issue slots and LDS cycles to them (6 until the packed half-blocks
            // made the walkers faster: 3 / 4 / 6 / 8 / 10 naps = 4.30 / 4.31 / 4.35 / 4.42 / 4.51 ms)
            a.helper_naps = knobs.helper_naps >= 0 ? static_cast<uint32_t>(knobs.helper_naps) : (segmented ? 4u : 1u);
            a.out_nodes = ws->nodes.as<uint32_t>(); a.out_offsets = ws->offsets.as<uint64_t>();
            a.xcd_map = knobs.xcd_map >= 0 ? (knobs.xcd_map ? 1u : 0u) : (segmented ? 1u : 0u);
            a.uniform_loop = knobs.uniform_loop >= 0 ? (knobs.uniform_loop ? 1u : 0u) : 1u;
            // (2: single steps on the two-step descriptors + packed half-blocks: what a lean handle has -- and GBWT_HIP_CATCH_UP=2 on any)
            a.catch_up = knobs.catch_up >= 0 ? static_cast<uint32_t>(std::min(knobs.catch_up, 2)) : 1u;
            if (ix->lean_extract && a.catch_up == 1u) a.catch_up = 2u;
            if (a.catch_up == 2u && !(ix->packed_blocks && knobs.packed_blocks != 0)) a.catch_up = ix->lean_extract ? 0u : 1u;
            // look-ahead in mixed waves where few rows pass a record (fewer than 512 BWT positions per record on average: config 4 has 52, the
            // headline 3 300 -- there seventy waves share every line and the extra load of the gather loop costs what it saves, round 2)
            a.gather_reach = knobs.gather_reach >= 0 ? static_cast<uint32_t>(knobs.gather_reach)
                                                     : (ix->stats.records != 0 && ix->host.size / ix->stats.records < 512 ? 1u : 0u);
            a.all4 = knobs.all4 != 0 ? 1u : 0u;
            a.headroom = static_cast<uint32_t>(knobs.headroom);
            // without packed half-blocks every wave starts on the full-width loops (the packed ones load before they look at GATHER_OK)
            a.packed_blocks = (ix->packed_blocks && knobs.packed_blocks != 0) ? 1u : 0u;
            DeviceIndex dev = a.packed_blocks ? ix->dev : with_cblocks(ix);
            dev.sample_stride = stride; dev.sample_part = strided.sample_part; dev.sample_parts = strided.sample_parts;
            a.row_piece = knobs.row_piece >= 0 ? static_cast<uint32_t>(knobs.row_piece) : 32u;
            if (a.ring_slots < 2 * a.row_piece) a.ring_slots = 2 * a.row_piece;   // a walker stops staging 8 slots before its ring is full: a ring of one piece would never hold one
            // GBWT_HIP_HEADROOM: the walkers wait while more than ring - headroom nodes are pending, the cooperative helper moves whole pieces only:
            // a headroom above ring - piece would leave a row with fewer than a piece pending and its walker waiting for ever
            if (a.headroom > a.ring_slots - std::max(a.row_piece, 1u)) a.headroom = a.ring_slots - std::max(a.row_piece, 1u);
            a.debug = knobs.debug;                               // timing experiments only, see WalkArgs::debug
            a.both_ends = (ix->orientation_pairs && knobs.both_ends != 0) ? 1u : 0u;
            a.capacity = defer ? capacity : 0;
            if (fused_offsets) { a.uniform_len = ix->uniform_len; a.fill_offsets = ws->offsets.as<uint64_t>(); }
            HIP_CHECK(hipEventRecord(ws->ev[0], s));
            if (!parted || (total != 0 && walkers != 0)) launch_walk(dev, a, s);   // (a part of nothing but empty stretches: no walkers)
            HIP_CHECK(hipEventRecord(ws->ev[1], s));
            HIP_CHECK(hipEventRecord(ws->ev[2], s));
            HIP_CHECK(hipStreamSynchronize(s));
            HIP_CHECK(hipGetLastError());
            if (defer) {
                total = ws->pinned_words[0];
                if (total > capacity) {                              // the rows were too small and nobody walked: make them, walk
                    ws->nodes.reserve(total * sizeof(uint32_t));
                    a.out_nodes = ws->nodes.as<uint32_t>(); a.capacity = 0;
                    HIP_CHECK(hipEventRecord(ws->ev[0], s));
                    if (walkers != 0) launch_walk(dev, a, s);
                    HIP_CHECK(hipEventRecord(ws->ev[1], s));
                    HIP_CHECK(hipEventRecord(ws->ev[2], s));
                    HIP_CHECK(hipStreamSynchronize(s));
                    HIP_CHECK(hipGetLastError());
                }
            }
            ws->timed = true; ws->last_n = n; ws->last_total = total;
            if (!parted) { ws->extract_key.assign(seq_ids, seq_ids + n); ws->extract_cached = true; }   // (gbwt_hip_extract's fill call asks for whole rows)
            out->d_offsets = ws->offsets.as<uint64_t>(); out->d_nodes = ws->nodes.as<uint32_t>(); out->total = total; out->n = n;
            return GBWT_HIP_OK;
        }
        if (n == 0) {                                        // nothing asked for: an empty CSR, whatever the handle holds
            HIP_CHECK(hipMemsetAsync(ws->offsets.ptr, 0, sizeof(uint64_t), s));
            ws->nodes.reserve(sizeof(uint32_t));
            HIP_CHECK(hipEventRecord(ws->ev[0], s)); HIP_CHECK(hipEventRecord(ws->ev[1], s)); HIP_CHECK(hipEventRecord(ws->ev[2], s));
            HIP_CHECK(hipStreamSynchronize(s));
            ws->timed = true; ws->last_n = 0; ws->last_total = 0;
            ws->extract_key.clear(); ws->extract_cached = true;
            out->d_offsets = ws->offsets.as<uint64_t>(); out->d_nodes = ws->nodes.as<uint32_t>(); out->total = 0; out->n = 0;
            return GBWT_HIP_OK;
        }
        if (ix->lean_extract) return fail(GBWT_HIP_UNSUPPORTED, "this handle was opened for extraction only and has given its raw descriptors back: the pool-output walk modes need GBWT_HIP_OPEN_ALL");
        // The pool-output kernels (no sequence lengths, GBWT_HIP_DIRECT=0, a tuned walk mode) walk whole rows and cannot cut them: as the
        // header says for rows that cannot be cut, the LAST part is the whole row and every earlier part is n empty rows -- never the whole
        // row from every part (a gather of parts would then hold every row `parts` times).
        if (parts > 1 && part + 1 < parts) {
            HIP_CHECK(hipMemsetAsync(ws->offsets.ptr, 0, (n + 1) * sizeof(uint64_t), s));
            ws->nodes.reserve(sizeof(uint32_t));
            HIP_CHECK(hipEventRecord(ws->ev[0], s)); HIP_CHECK(hipEventRecord(ws->ev[1], s)); HIP_CHECK(hipEventRecord(ws->ev[2], s));
            HIP_CHECK(hipStreamSynchronize(s));
            ws->timed = true; ws->last_n = n; ws->last_total = 0;
            out->d_offsets = ws->offsets.as<uint64_t>(); out->d_nodes = ws->nodes.as<uint32_t>(); out->total = 0; out->n = n;
            return GBWT_HIP_OK;
        }
        uint32_t flags = 0;
        WalkArgs a{};
        const DeviceIndex dev = ws->walk_mode == WALK_TWO_STEP ? with_cblocks(ix) : ix->dev;   // the pool-output kernel walks on the full-width two-step blocks
        for (int attempt = 0; attempt < 8; attempt++) {
            if (pool_blocks >= POOL_NONE) return fail(GBWT_HIP_UNSUPPORTED, "path pool would exceed 2^32 blocks");
            ws->pool.reserve(pool_blocks * POOL_BLOCK_NODES * sizeof(uint32_t));
            ws->next.reserve(pool_blocks * sizeof(uint32_t));
            a.seq_ids = ws->seq_ids.as<uint64_t>(); a.n = n;
            a.pool = ws->pool.as<uint32_t>(); a.next = ws->next.as<uint32_t>(); a.pool_blocks = static_cast<uint32_t>(pool_blocks);
            a.counter = ws->counters.as<uint32_t>(); a.flags = ws->counters.as<uint32_t>() + 1;
            a.head = ws->head.as<uint32_t>(); a.lengths = ws->lengths.as<uint64_t>();
            a.mode = ws->walk_mode; a.small_record = ws->small_record;
            // automatic: the walk is latency-bound and every lane of a wave runs the same instructions whether it owns a
            // sequence or not, so owners per wave cost nothing; at least 32 measured best (fewer, fuller waves keep the walks
            // of an XCD closer together, which is what the look-ahead relies on), more only when there are > 32k sequences
            a.paths_per_wave = ws->paths_per_wave ? ws->paths_per_wave
                                                   : static_cast<uint32_t>(std::min<uint64_t>(64, std::max<uint64_t>(32, (n + 1023) / 1024)));
            a.pack16 = ix->stats.max_record_len < 65536 ? 1u : 0u;
            a.helper_lanes = 64u;
            a.wide_addresses = knobs.wide_addresses ? 1u : 0u;
            HIP_CHECK(hipMemsetAsync(ws->counters.ptr, 0, 4 * sizeof(uint32_t), s));
            HIP_CHECK(hipEventRecord(ws->ev[0], s));
            launch_walk(dev, a, s);
            HIP_CHECK(hipEventRecord(ws->ev[1], s));
            HIP_CHECK(hipMemcpyAsync(&flags, a.flags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            launch_scan(a.lengths, ws->offsets.as<uint64_t>(), n, ws->scan_temp.ptr, temp_bytes, s);
            HIP_CHECK(hipStreamSynchronize(s));
            HIP_CHECK(hipGetLastError());
            if (!(flags & FLAG_POOL_OVERFLOW)) break;
            pool_blocks *= 2;  // duplicate ids can exceed the distinct-id bound: grow and walk again
        }
        if (flags & FLAG_POOL_OVERFLOW) return fail(GBWT_HIP_DEVICE_ERROR, "path pool overflow");
        uint64_t total = 0;
        HIP_CHECK(hipMemcpyAsync(&total, ws->offsets.as<uint64_t>() + n, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        ws->nodes.reserve(std::max<uint64_t>(total, 1) * sizeof(uint32_t));
        launch_compact(a, ws->offsets.as<uint64_t>(), ws->nodes.as<uint32_t>(), s);
        HIP_CHECK(hipEventRecord(ws->ev[2], s));
        HIP_CHECK(hipStreamSynchronize(s));
        HIP_CHECK(hipGetLastError());
        ws->timed = true;
        ws->last_n = n; ws->last_total = total;
        ws->extract_key.assign(seq_ids, seq_ids + n); ws->extract_cached = true;
        out->d_offsets = ws->offsets.as<uint64_t>();
        out->d_nodes = ws->nodes.as<uint32_t>();
        out->total = total;
        out->n = n;
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
    GBWT_HIP_GUARD_END
}

}  // extern "C"

namespace gbwt_hip {

// Device -> pageable host memory for the large results (13 GB of node ids on the headline index, gigabytes of GFA lines).  A plain
// hipMemcpy stages through one pinned buffer on one thread and also pays the first-touch page faults of a fresh destination on that
// thread (0.5 - 1.1 s for 13.3 GB; 2.4 GB/s for a gigabyte of W-lines).  Here a few threads take alternate 16 MiB chunks, each with two
// pinned buffers and a stream of its own: the copy of a thread's next chunk over PCIe runs under its memcpy of the present one out of
// the pinned buffer, and the threads' page faults run side by side.  Buffers and streams belong to the workspace (HostCopier) and are
// made once: allocating pinned memory per call cost as much as copying a gigabyte.
HostCopier::~HostCopier() {
    for (Lane &l : lanes) {
        for (void *p : l.pinned) if (p) (void)hipHostFree(p);
        for (hipEvent_t e : l.landed) if (e) (void)hipEventDestroy(e);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
}

bool HostCopier::ensure(int device, unsigned threads) {
    if (hipSetDevice(device) != hipSuccess) return false;
    while (lanes.size() < threads) {
        Lane l;
        bool ok = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess;
        for (int b = 0; b < 2 && ok; b++) ok = hipHostMalloc(&l.pinned[b], CHUNK, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&l.landed[b], hipEventDisableTiming) == hipSuccess;
        lanes.push_back(l);            // (its destructor frees whatever a failed lane got)
        if (!ok) { (void)hipGetLastError(); return false; }
    }
    return true;
}

void copy_to_host(gbwt_hip_workspace *ws, void *dst, const void *src, size_t bytes, size_t piece) {
    const size_t CHUNK = std::min(std::max<size_t>(piece, 4096), HostCopier::CHUNK);
    const int device = ws->index->device;
    const unsigned threads = ws->knobs.copy_threads;
    if (bytes < 4 * CHUNK || threads < 2) { HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return; }
    std::lock_guard<std::mutex> guard(ws->copier.busy);
    if (!ws->copier.ensure(device, threads)) { HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return; }
    const size_t chunks = (bytes + CHUNK - 1) / CHUNK;
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&](unsigned t) {
        HostCopier::Lane &l = ws->copier.lanes[t];
        if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
        auto issue = [&](size_t c, int b) {
            const size_t at = c * CHUNK, len = std::min(CHUNK, bytes - at);
            return hipMemcpyAsync(l.pinned[b], static_cast<const char *>(src) + at, len, hipMemcpyDeviceToHost, l.stream) == hipSuccess &&
                   hipEventRecord(l.landed[b], l.stream) == hipSuccess;
        };
        size_t cur = next++;
        int b = 0;
        if (cur >= chunks) return;
        if (!issue(cur, b)) { failed = 1; return; }
        while (!failed) {
            const size_t nxt = next++;
            if (nxt < chunks && !issue(nxt, b ^ 1)) { failed = 1; break; }
            if (hipEventSynchronize(l.landed[b]) != hipSuccess) { failed = 1; break; }
            std::memcpy(static_cast<char *>(dst) + cur * CHUNK, l.pinned[b], std::min(CHUNK, bytes - cur * CHUNK));
            if (nxt >= chunks) break;
            cur = nxt; b ^= 1;
        }
        (void)hipStreamSynchronize(l.stream);
    };
    const unsigned used = static_cast<unsigned>(std::min<size_t>(threads, chunks));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < used; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &t : pool) t.join();
    if (failed) { (void)hipGetLastError(); HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); }   // whatever went wrong: the plain way, which reports it
}


}  // namespace gbwt_hip

extern "C" {

gbwt_hip_status gbwt_hip_extract(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *seq_ids, uint64_t n,
                                 uint64_t *out_offsets, uint32_t *out_nodes, uint64_t capacity, uint64_t *total) {
    GBWT_HIP_GUARD_BEGIN
    if (!out_offsets || !total) return fail(GBWT_HIP_BAD_ARGUMENT, "null output");
    gbwt_hip_paths p{};
    // the fill call after a size query (or after gbwt_hip_extract_device) with the same ids: the rows are still in the workspace
    if (ix && ws && ws->index == ix && ws->extract_cached && ws->extract_key.size() == n && (n == 0 || (seq_ids && std::memcmp(ws->extract_key.data(), seq_ids, n * sizeof(uint64_t)) == 0))) {
        p.d_offsets = ws->offsets.as<uint64_t>(); p.d_nodes = ws->nodes.as<uint32_t>(); p.total = ws->last_total; p.n = n;
    } else {
        gbwt_hip_status st = gbwt_hip_extract_device(ix, ws, seq_ids, n, &p);
        if (st != GBWT_HIP_OK) return st;
    }
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        HIP_CHECK(hipMemcpy(out_offsets, p.d_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
        *total = p.total;
        if (!out_nodes) return GBWT_HIP_OK;
        if (capacity < p.total) return fail(GBWT_HIP_CAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(p.total));
        if (p.total) copy_to_host(ws, out_nodes, p.d_nodes, p.total * sizeof(uint32_t));
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_extract_paths(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *path_ids, uint64_t n,
                                       int reverse, uint64_t *out_offsets, uint32_t *out_nodes, uint64_t capacity,
                                       uint64_t *total) {
    GBWT_HIP_GUARD_BEGIN
    if (n && !path_ids) return fail(GBWT_HIP_BAD_ARGUMENT, "null path_ids");
    std::vector<uint64_t> ids(n);
    for (uint64_t k = 0; k < n; k++) ids[k] = 2 * path_ids[k] + (reverse ? 1 : 0);  // support::encode_path
    return gbwt_hip_extract(ix, ws, ids.data(), n, out_offsets, out_nodes, capacity, total);
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_copy_result(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, uint64_t *out_offsets, uint32_t *out_nodes, uint64_t capacity) {
    GBWT_HIP_GUARD_BEGIN
    if (!ix || !ws || ws->index != ix || !ws->timed) return fail(GBWT_HIP_BAD_ARGUMENT, "no device-resident extraction on this workspace");
    if (!out_offsets && !out_nodes) return fail(GBWT_HIP_BAD_ARGUMENT, "null outputs");
    HIP_CHECK(hipSetDevice(ix->device));
    if (out_offsets) HIP_CHECK(hipMemcpy(out_offsets, ws->offsets.ptr, (ws->last_n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (out_nodes) {
        if (capacity < ws->last_total) return fail(GBWT_HIP_CAPACITY, "output capacity " + std::to_string(capacity) + " < " + std::to_string(ws->last_total));
        if (ws->last_total) copy_to_host(ws, out_nodes, ws->nodes.ptr, ws->last_total * sizeof(uint32_t));
    }
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

static gbwt_hip_status path_sums(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, uint64_t *out_sums, uint64_t n, bool hashed) {
    if (!ix || !ws || ws->index != ix || !ws->timed) return fail(GBWT_HIP_BAD_ARGUMENT, "no device-resident extraction on this workspace");
    if (n != ws->last_n || (n && !out_sums)) return fail(GBWT_HIP_BAD_ARGUMENT, "n does not match the last extraction");
    if (n == 0) return GBWT_HIP_OK;
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        ws->out_a.reserve(n * sizeof(uint64_t));
        launch_path_sums(ws->offsets.as<uint64_t>(), ws->nodes.as<uint32_t>(), n, ws->out_a.as<uint64_t>(), hashed, ws->stream);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out_sums, ws->out_a.ptr, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ws->stream));
        HIP_CHECK(hipStreamSynchronize(ws->stream));
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
}

gbwt_hip_status gbwt_hip_path_sums(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, uint64_t *out_sums, uint64_t n) {
    GBWT_HIP_GUARD_BEGIN
    return path_sums(ix, ws, out_sums, n, false);
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_path_hashes(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, uint64_t *out_hashes, uint64_t n) {
    GBWT_HIP_GUARD_BEGIN
    return path_sums(ix, ws, out_hashes, n, true);
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_copy_path(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, uint64_t k, uint32_t *out_nodes,
                                   uint64_t capacity, uint64_t *len) {
    GBWT_HIP_GUARD_BEGIN
    if (!ix || !ws || ws->index != ix || !ws->timed || !len) return fail(GBWT_HIP_BAD_ARGUMENT, "no device-resident extraction on this workspace");
    if (k >= ws->last_n) return fail(GBWT_HIP_BAD_ARGUMENT, "row out of range");
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        uint64_t range[2];
        HIP_CHECK(hipMemcpy(range, ws->offsets.as<uint64_t>() + k, sizeof(range), hipMemcpyDeviceToHost));
        *len = range[1] - range[0];
        uint64_t count = std::min(*len, capacity);
        if (count && out_nodes) HIP_CHECK(hipMemcpy(out_nodes, ws->nodes.as<uint32_t>() + range[0], count * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_last_kernel_ms(const gbwt_hip_workspace *ws, float *walk_ms, float *total_ms) {
    GBWT_HIP_GUARD_BEGIN
    if (!ws || !ws->timed) return fail(GBWT_HIP_BAD_ARGUMENT, "no timed extraction on this workspace");
    float a = 0, b = 0;
    if (hipEventElapsedTime(&a, ws->ev[0], ws->ev[1]) != hipSuccess || hipEventElapsedTime(&b, ws->ev[3], ws->ev[2]) != hipSuccess)
        return fail(GBWT_HIP_DEVICE_ERROR, "hipEventElapsedTime failed");
    if (walk_ms) *walk_ms = a;
    if (total_ms) *total_ms = b;
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

}  // extern "C"
