// kernels.hpp -- launch wrappers of the gfx950 kernels (load_kernels.hip, walk_kernels.hip, query_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/gbwt_hip.h"
#include "device_index.hpp"

namespace gbwt_hip {

constexpr uint32_t POOL_BLOCK_NODES = 256;        // node ids per pool block (1 KiB)
constexpr uint32_t POOL_NONE = 0xFFFFFFFFu;

constexpr uint32_t FLAG_POOL_OVERFLOW = 1u;

// ---- load-time passes -------------------------------------------------------------------------
// per-record descriptors + rank blocks (device_index.hpp): d_desc has 4 * n_records entries; build_desc also
// writes the number of rank blocks of every record and the statistics of the index (d_stats, zeroed by the caller: [0] = max
// Record::len, [1] = max outdegree, [2] = number of malformed records, [3] = all BWT positions); block_scan + finish_block_base turn
// the counts into first-block indices (BLOCK_NONE where there are none; block 0 is the shared zero block), fill_blocks decodes
// the outdegree-2 records into their blocks, link_desc stores the successors' block bases into the descriptors.
void launch_build_desc(const DeviceIndex &ix, uint4 *d_desc, uint32_t *d_block_counts, uint64_t *d_stats, hipStream_t stream);
size_t block_scan_temp_bytes(uint64_t n);
void launch_block_scan(const uint32_t *d_counts, uint32_t *d_block_base, uint64_t n, void *d_temp, size_t temp_bytes, hipStream_t stream);
void launch_finish_block_base(const uint32_t *d_counts, uint32_t *d_block_base, uint64_t n, uint4 *d_desc_raw /* class 2 records also get their block base in C.z */, hipStream_t stream);
void launch_link_desc(const DeviceIndex &ix, uint4 *d_desc, hipStream_t stream);
void launch_link_lookahead(const DeviceIndex &ix, uint4 *d_desc, const uint32_t *d_block_counts, uint32_t hops, hipStream_t stream);
// LF tables of the class 0 records (device_index.hpp): per-record positions / outdegrees, then the fill
void launch_table_counts(const DeviceIndex &ix, uint64_t *d_positions, uint64_t *d_sigmas, hipStream_t stream);
void launch_fill_tables(const DeviceIndex &ix, uint4 *d_desc_raw, const uint64_t *d_table_base, const uint64_t *d_edge_base, uint4 *d_tables,
                        uint2 *d_edges, hipStream_t stream);
// walk tables: the LF tables with the step through a unary successor and the landing record's base folded in (after fill_tables)
void launch_fill_wtables(const DeviceIndex &ix, uint4 *d_wtables, hipStream_t stream);
void launch_fill_wtables_deep(const DeviceIndex &ix, uint4 *d_deep, bool compact, hipStream_t stream);   // ix.wtables set; 4 uint4 per table position; compact: twelve-step entries where they fit (device_index.hpp: WT_COMPACT)
// two-step walk (device_index.hpp): composed descriptors, two-step rank blocks, look-ahead targets
void launch_link_desc2(const DeviceIndex &ix, uint4 *d_desc2, uint32_t gather_limit, uint32_t chain_max, uint32_t *d_chained, hipStream_t stream);   // chain_max: bits 0-7 = most nodes a chained step adds, bit 8 = the index is bidirectional; *d_chained = the most nodes one iteration can stage where a step was chained (zeroed by the caller)   // gather_limit: Record::len below which the packed blocks can count
void launch_fill_two_step_blocks(const DeviceIndex &ix, uint4 *d_cblocks /* or null */, uint4 *d_gblocks /* or null */, hipStream_t stream);   // one of the two layouts, or both in one pass
void launch_link_lookahead2(const DeviceIndex &ix, uint4 *d_desc2, const uint32_t *d_block_counts, uint32_t hops, hipStream_t stream);
void launch_fill_blocks(const DeviceIndex &ix, const uint32_t *d_block_counts, const uint32_t *d_block_base, uint4 *d_blocks, hipStream_t stream);

// ---- extraction -------------------------------------------------------------------------------
struct WalkArgs {
    const uint64_t *seq_ids;   // [n]
    uint64_t n;
    uint32_t *pool;            // [pool_blocks * POOL_BLOCK_NODES]
    uint32_t *next;            // [pool_blocks] chain links
    uint32_t pool_blocks;
    uint32_t *counter;         // [1] next free block
    uint32_t *head;            // [n] first block of each path
    uint64_t *lengths;         // [n]
    uint32_t *flags;           // [1]
    // tuning (gbwt_hip_workspace_tune)
    uint32_t mode;             // WALK_TWO_STEP (default), WALK_LANE_SERIAL, WALK_COOP or WALK_ONE_STEP
    uint32_t paths_per_wave;   // lanes of a wave that own a path (1..64)
    uint32_t small_record;     // WALK_COOP: records of at most this many bytes are decoded lane-serially
    uint32_t pack16;           // WALK_COOP: every record is shorter than 2^16 (stats.max_record_len): one packed scan
    uint32_t helper_lanes;     // lanes of the look-ahead helper wave that touch (64; 0 = no look-ahead; experiments)
    uint32_t wide_addresses;   // force the 64-bit-address variant of the two-step loop (used above 4 GiB; tests)
    // direct output (sequence lengths known): rows of the CSR are written in place, both ends of a sequence at once
    uint32_t *out_nodes;       // [offsets[n]] CSR nodes, or null = pool output
    const uint64_t *out_offsets;   // [n + 1]
    uint32_t both_ends;        // 1: walkers n .. 2n-1 walk sequence id ^ 1 for the second half of row k = w - n
    uint32_t ring_slots;       // k_walk_direct: LDS ring slots per lane (32, 64 or 128)
    uint32_t helper_naps;      // helper wave of k_walk_direct: s_sleep 4 (256 cycles) this many times between polls
    uint32_t row_piece;        // segmented extraction: nodes per cooperative row write (16 = 64 bytes, 32 = 128 bytes; 0 = every lane writes its own row)
    uint32_t xcd_map;          // k_walk_direct: XCD x (workgroups x, x + 8, ...) takes the x-th eighth of the walkers
    uint32_t uniform_loop;     // k_walk_direct: try the wave-uniform loop (scalar descriptor fetch) first
    uint32_t headroom;         // k_walk_direct: ring slots a loop keeps free (0 = by the index: 8, or the longest chained iteration; measurements)
    uint32_t all4;             // k_walk_direct: the uniform loop stages the four nodes of an iteration in a row on E_ALL4 records (0: counts every node; measurements)
    uint32_t catch_up;         // k_walk_direct: a mixed wave first lets the lanes that are behind take single steps (walk_direct.hip: CATCH-UP)
    uint32_t gather_reach;     // k_walk_direct: > 0 = the gather loop posts the look-ahead target of every record it leaves and the helper touches the descriptors
                               // and blocks of this many consecutive records from it (round 6: indexes with few rows per record); 0 = no look-ahead in mixed waves
    uint32_t packed_blocks;    // k_walk_direct: the uniform loop starts on the packed half-blocks (0: on the full-width blocks at once; measurements)
    uint32_t debug;            // measurement switches of k_walk_direct (GBWT_HIP_DEBUG_DRY_ROWS; never set by the library itself; the output is
                               // wrong with 1, 2 and 128): 1 = no row stores, 2 = all rows written into one 64 MB window, 128 = into 1 MB, 4 = plain
                               // instead of non-temporal row stores, 32 = row groups in row order, 16384 = XCDs 0-3 only (with XCD_MAP=0)
    // segmented extraction with rows of different lengths: walkers in (segment, row) order with the rows that have no such
    // segment left out.  rows sorted by their number of segments (descending, stable); level[j] = walkers before segment j.
    const uint32_t *sorted_rows;   // [n]
    const uint64_t *level;         // [segments + 1]; level[segments] = number of walkers; null = every row has every segment (w = j * n + k)
    uint64_t walkers;              // = level[segments] (host copy)
    uint64_t capacity;         // > 0: the rows hold this many nodes and out_offsets[n] may say that more are needed: then nobody walks (gbwt_hip_extract_part_device launches again)
    uint32_t segments;         // > 0: walker w fills segment w / n of row w % n, starting at that sequence sample (DeviceIndex::samples)
    // ROWS OF ONE LENGTH (round 6): every row of the batch has uniform_len nodes (all ids valid, whole rows): row k starts at k * uniform_len --
    // the walkers compute it, and the kernel itself writes the n + 1 offsets the caller gets to fill_offsets (= out_offsets): the launch that
    // computed them in front of the walk (k_row_offsets: 11 us + a launch gap, a sixth of config 5's 80 us walk) is gone for such batches
    uint32_t uniform_len;      // 0: read out_offsets
    uint64_t *fill_offsets;
};
constexpr uint32_t WALK_TWO_STEP = 0, WALK_LANE_SERIAL = 1, WALK_COOP = 2, WALK_ONE_STEP = 3;
void launch_walk(const DeviceIndex &ix, const WalkArgs &args, hipStream_t stream);
// lengths of all sequences (one lane per sequence, counting walk) and their gather for a batch of ids
void launch_sequence_lengths(const DeviceIndex &ix, uint32_t *d_seq_len, uint64_t *d_prints, uint32_t *d_overflow, hipStream_t stream);
// bidirectional indexes: is sequence 2k + 1 the reverse of sequence 2k for every k (fingerprints from the pass above)?
void launch_check_orientation_pairs(const uint32_t *d_seq_len, const uint64_t *d_prints, uint64_t n_pairs, uint32_t *d_mismatch, hipStream_t stream);
// lengths[k] = seq_len[ids[k]]; d_max_len[0] = the largest of them, d_max_len[1] = ~(the smallest) (both zeroed by the caller)
void launch_gather_lengths(const uint32_t *d_seq_len, uint64_t n_sequences, const uint64_t *d_ids, uint64_t n, uint64_t *d_lengths, uint32_t *d_max_len, hipStream_t stream);
// lengths and samples in one walk: samples into a pool of `capacity` entries (*d_counter = samples met, may exceed the
// capacity: then nothing usable was recorded), then blank + place into the final table
void launch_lengths_and_samples(const DeviceIndex &ix, uint32_t interval, uint32_t *d_seq_len, uint4 *d_pool, uint2 *d_tags, uint64_t *d_counter,
                                uint64_t capacity, uint32_t *d_overflow, hipStream_t stream);
void launch_place_samples(const uint32_t *d_seq_len, const uint4 *d_pool, const uint2 *d_tags, uint64_t pooled, const uint64_t *d_sample_base,
                          uint64_t n_sequences, uint4 *d_samples, hipStream_t stream);
// sequence samples: counts per sequence from the lengths, then the recording walk
void launch_sample_counts(const uint32_t *d_seq_len, uint64_t n_sequences, uint32_t interval, uint64_t *d_counts, hipStream_t stream);
void launch_record_samples(const DeviceIndex &ix, const uint64_t *d_sample_base, uint32_t interval, uint4 *d_samples, hipStream_t stream);
// checkpoint sampling (open_walks.hip): lengths and samples of all sequences without walking any sequence from end to end.
// Records whose hashed index is below `threshold` are checkpoints; one walker per position of every checkpoint record and one per
// sequence start walks to the next checkpoint (at most `cap` nodes per hop: there it takes a summary slot for the position it has reached -- an
// orphan -- and walks on) and stores
// a summary {landing record, offset, nodes walked, summary index of the landing position}; launch_chase_counts follows the summaries of
// every sequence (by splitters: d_spans, n_sequences + n_summaries / 16 + 1 entries) and counts (d_seq_len, d_counts), launch_chase_samples writes the samples at d_sample_base.
struct CheckpointWalk {
    const uint64_t *cp_first;      // [n_records + 1] exclusive scan of launch_checkpoint_counts
    uint4 *summaries;              // [n_sequences + positions + orphan_capacity]
    uint64_t *orphan_count;        // [1] zeroed by the caller
    uint64_t orphan_capacity, positions;
    uint32_t threshold, cap;
    uint32_t packed;               // the index has packed half-blocks
    uint32_t *flags;               // bit 2: orphan pool full
};
void launch_checkpoint_counts(const DeviceIndex &ix, uint32_t threshold, uint64_t *d_counts, hipStream_t stream);
void launch_checkpoint_walk(const DeviceIndex &ix, const CheckpointWalk &w, hipStream_t stream);
void launch_chase_counts(const DeviceIndex &ix, const uint4 *d_summaries, uint64_t n_summaries, uint4 *d_spans, uint32_t *d_seq_len, uint64_t *d_counts,
                         uint32_t *d_overflow, hipStream_t stream);
void launch_chase_samples(const DeviceIndex &ix, const uint4 *d_summaries, uint64_t n_summaries, const uint4 *d_spans, const uint64_t *d_sample_base,
                          uint4 *d_samples, hipStream_t stream);
// THE LINE CACHE, FILLED AT OPEN (round 6; gfa.hip: fill_line_cache_at_open).  What the GFA line of a path is made of -- the bytes of its
// W-line node tokens in front of every GFA_LINE_CHUNK positions, their total, and the summed label lengths (the W-line's end coordinate,
// src/bin/gbunzip.rs:532-540) -- is a property of the index.  Rounds 1-5 found it per request (a sizing pass over the extracted rows: 12 of
// the 52 ms of config 4's first request) and, round 5, left it in the handle for later requests of the same path.  Now one walker per
// SEGMENT of every forward sequence (from one sequence sample to the next: the walkers of an extraction, in plain C++) walks its segment
// once at open, counting token bytes and label lengths instead of emitting nodes; a scan over the segments of each sequence and one
// thread per chunk put the numbers where the requests read them.  No request sizes a line any more: the first one is like every other.
constexpr uint32_t GFA_LINE_CHUNK = 4096;      // path positions per formatting chunk (gfa.hip)
struct LineCacheFill {
    const uint32_t *label_len;     // [n_labels] label length per potential node (gfa.hip: upload_label_lengths)
    uint64_t n_labels;
    uint64_t paths;                // path p = sequence 2p (support::encode_path)
    uint32_t max_samples;          // the most samples a forward sequence has
    const uint64_t *chunk_first;   // [paths + 1] first chunk of every path in chunk_text
    uint64_t chunks;               // = chunk_first[paths] (host copy)
    uint64_t *chunk_text;          // out, per chunk: W-token bytes of the path in front of the chunk
    uint64_t *path_totals;         // out, [2 * paths]: {W-token bytes, summed label lengths}
    uint32_t *chunk_seg;           // scratch, per chunk: the segment its first position lies in
    uint64_t *seg_text;            // scratch, [2 * samples of all sequences]: {token bytes, label lengths} of every segment, then the token bytes in front of it
    uint32_t *flags;               // [1] zeroed by the caller; bit 0: a segment did not deliver its nodes (the cache is not used)
};
void launch_fill_line_cache(const DeviceIndex &ix, const LineCacheFill &f, hipStream_t stream);
// walker order of a segmented extraction: per-row segment counts -> rows sorted by count (descending, stable) and
// level[j] = number of walkers in segments < j.  d_keys / d_rows: 2 x n scratch each (double buffers of the sort).
size_t walker_order_temp_bytes(uint64_t n);
// lengths (whole rows, or the part of them the request fills: ix.sample_parts), their exclusive scan (n + 1 offsets) and d_max_len = {longest,
// ~shortest} in ONE launch of one workgroup; false (nothing launched) above ROW_OFFSETS_MAX rows, where the three launches are faster
constexpr uint64_t ROW_OFFSETS_MAX = uint64_t(1) << 13;   // (one workgroup takes about 3 us per 1 024 rows: config 4's 32 000 rows were 70 us slower this way than by three launches)
bool launch_row_offsets(const DeviceIndex &ix, const uint64_t *d_ids, uint64_t n, uint64_t *d_lengths, uint64_t *d_offsets, uint32_t *d_max_len, hipStream_t stream);
// nodes of every row inside the part of it that the request fills (ix.sample_part of ix.sample_parts, at ix.sample_stride); d_max_len as launch_gather_lengths
void launch_part_lengths(const DeviceIndex &ix, const uint64_t *d_ids, uint64_t n, uint64_t *d_lengths, uint32_t *d_max_len, hipStream_t stream);
void launch_walker_order(const DeviceIndex &ix, const uint64_t *d_ids, uint64_t n, uint32_t segments, uint32_t *d_keys, uint32_t *d_rows,
                         uint64_t *d_level_counts, uint64_t *d_level, void *d_temp, size_t temp_bytes, const uint32_t **d_sorted_rows, hipStream_t stream);
void launch_compact(const WalkArgs &args, const uint64_t *d_offsets, uint32_t *d_nodes, hipStream_t stream);

// per-path sum of node ids over CSR rows (checking hook)
void launch_path_sums(const uint64_t *d_offsets, const uint32_t *d_nodes, uint64_t n, uint64_t *d_sums, bool hashed, hipStream_t stream);

// ---- navigation / search ----------------------------------------------------------------------
void launch_start(const DeviceIndex &ix, const uint64_t *ids, uint64_t n, gbwt_hip_pos *out, uint8_t *valid, hipStream_t s);
void launch_forward(const DeviceIndex &ix, const gbwt_hip_pos *in, uint64_t n, gbwt_hip_pos *out, uint8_t *valid, hipStream_t s);
void launch_backward(const DeviceIndex &ix, const gbwt_hip_pos *in, uint64_t n, gbwt_hip_pos *out, uint8_t *valid, hipStream_t s);
void launch_find(const DeviceIndex &ix, const uint64_t *nodes, uint64_t n, gbwt_hip_state *out, uint8_t *valid, hipStream_t s);
void launch_extend(const DeviceIndex &ix, const gbwt_hip_state *states, const uint64_t *nodes, uint64_t n,
                   gbwt_hip_state *out, uint8_t *valid, hipStream_t s);
void launch_bd_find(const DeviceIndex &ix, const uint64_t *nodes, uint64_t n, gbwt_hip_bd_state *out, uint8_t *valid, hipStream_t s);
void launch_bd_extend(const DeviceIndex &ix, const gbwt_hip_bd_state *states, const uint64_t *nodes, uint64_t n,
                      bool backward, gbwt_hip_bd_state *out, uint8_t *valid, hipStream_t s);
// GBZ::follow_forward / follow_backward: count pass (per-state number of extensions, valid = the iterator exists), then
// fill pass at the offsets of an exclusive scan of the counts (offsets has n + 1 entries)
void launch_follow_count(const DeviceIndex &ix, const gbwt_hip_bd_state *states, uint64_t n, bool backward, uint64_t *counts, uint8_t *valid, hipStream_t s);
void launch_follow_fill(const DeviceIndex &ix, const gbwt_hip_bd_state *states, uint64_t n, bool backward, const uint64_t *offsets,
                        gbwt_hip_bd_state *out, hipStream_t s);
void launch_search(const DeviceIndex &ix, const uint64_t *queries, uint64_t n, uint64_t len, gbwt_hip_state *out,
                   uint8_t *valid, hipStream_t s);

void launch_bd_search(const DeviceIndex &ix, const uint64_t *queries, uint64_t n, uint64_t len, uint64_t first,
                      gbwt_hip_bd_state *out, uint8_t *valid, hipStream_t s);

// The record starts decoded ON THE DEVICE from the Elias-Fano words of the file (round 6; SparseVector, SURVEY Appendix A: value_k =
// ((pos_k - k) << w) | low[k], pos_k = k-th set bit of `high`): d_counts[i] = popcount(high[i]) (launch_scan turns them into the rank of
// every word's first one), then every word writes the values of its ones -- clamped to data_len, at most `ones` of them -- narrowed to
// u32 where d_starts32 is given; entry `ones` = data_len (the sentinel).  launch_starts_check: flags[0] |= 1 where two neighbours are out
// of order (BWT::load's callers rely on sorted starts, src/bwt.rs:116-121).
void launch_ef_counts(const uint64_t *d_high, uint64_t high_words, uint64_t *d_counts, hipStream_t s);
void launch_ef_values(const uint64_t *d_high, uint64_t high_words, const uint64_t *d_rank, const uint64_t *d_low, uint64_t low_words, uint32_t width, uint64_t ones,
                      uint64_t data_len, uint32_t *d_starts32, uint64_t *d_starts64, hipStream_t s);
void launch_starts_check(const uint32_t *d_starts32, const uint64_t *d_starts64, uint64_t ones, uint32_t *d_flags, hipStream_t s);
// label_len[s] = 0 for every potential node s < n whose forward record (2 s + 1) is empty or has no edge (GBZ::has_node); reads the record bytes
// and starts on the device -- the host's image of them may not exist (HostIndex::ensure_records)
void launch_mask_label_lengths(const DeviceIndex &ix, uint32_t *d_label_len, uint64_t n, hipStream_t s);

// inclusive scan of lengths[n] into offsets[1..n], offsets[0] = 0 (hipcub); temp storage managed by caller
size_t scan_temp_bytes(uint64_t n);
void launch_scan(const uint64_t *d_lengths, uint64_t *d_offsets, uint64_t n, void *d_temp, size_t temp_bytes, hipStream_t s);

}  // namespace gbwt_hip
