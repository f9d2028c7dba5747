// gfa.hip -- gbunzip's GFA text (src/bin/gbunzip.rs:193-550) on top of the device extraction.
//
// P- and W-lines are where the LF-steps go, so they are produced on the device: the forward sequences of a
// batch of paths are extracted (k_walk_*), a statistics kernel sizes every line, the host -- which owns the
// metadata strings -- builds the per-line headers, and a formatting kernel writes the node tokens
// (">123" / "<123" for walks, "123+" / "123-" joined by commas for paths).  H-, S- and L-lines are serial
// host work in the reference as well (write_segments / write_links) and stay on the host.
//
// Graphs with a node-to-segment translation (Graph::has_translation, src/graph.rs:158-160) print segment names:
// SegmentPathIter (src/gbz.rs:1098-1169) turns the node sequence into one token per segment.  A path that is a
// concatenation of whole segments -- every path of a GBZ built from a GFA -- is recognised and formatted on the
// device (a position is a token iff it is the first node of its segment in the orientation of travel, and every
// other position must continue its predecessor).  A path that is not makes the reference's iterator stop at an
// input-dependent place; such paths are flagged by the device and formatted by the host with the reference's state
// machine, from the node ids the device extracted.
#include <fcntl.h>
#include <hipcub/hipcub.hpp>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "capi_internal.hpp"
#include "gfa_tokens.hpp"

using namespace gbwt_hip;

namespace {

constexpr int WAVE = 64;
constexpr int FORMAT_THREADS = 256;
const char *const GENERIC_SAMPLE = "_gbwt_ref";  // src/lib.rs

__device__ __forceinline__ uint32_t decimal_digits(uint32_t v) {
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) +
           (v >= 100000000u) + (v >= 1000000000u);
}

__device__ __forceinline__ uint32_t decimal_digits64(uint64_t v) {
    uint32_t d = 1;
    while (v >= 10u) { v /= 10u; d++; }
    return d;
}

// The headers of one line mode as the kernels see them (gbwt_hip_index::line_prefix): everything in front of the node tokens, per path of
// the metadata.  W-lines (fragment != null) end in "<fragment>\t<fragment + summed label lengths>\t" (path_to_w_line,
// src/bin/gbunzip.rs:532-540): the table holds the line up to and including "<fragment>\t", the end coordinate and its tab are
// appended where the line is formatted.
struct LineHeaders {
    const uint8_t *prefix;
    const uint64_t *prefix_off;    // [paths + 1]
    const uint32_t *fragment;      // [paths], or null: no end coordinate (P-lines)
};

// Lines are formatted in chunks of LINE_CHUNK path positions, so that the work is as parallel for ninety haplotypes of two
// million nodes as it is for fifty thousand short walks (one workgroup per LINE had 0.6 G nodes/s on the former, 50 on the
// latter).  Chunk c belongs to the path with chunk_first[path] <= c < chunk_first[path + 1] (every path has at least one
// chunk: an empty path still has a header and a trailer).
constexpr uint32_t LINE_CHUNK = GFA_LINE_CHUNK;   // (kernels.hpp: the line cache is filled per chunk of this size at open)

__global__ void __launch_bounds__(256) k_chunk_counts(const uint64_t *offsets, uint64_t n, uint64_t *counts) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= n) return;
    const uint64_t len = offsets[p + 1] - offsets[p];
    counts[p] = len == 0 ? 1 : (len + LINE_CHUNK - 1) / LINE_CHUNK;
}

struct ChunkRange { uint64_t path, begin, lo, hi; bool first, last; };

__device__ __forceinline__ uint64_t chunk_path_of(const uint64_t *chunk_first, uint64_t n, uint64_t c) {
    uint64_t lo = 0, hi = n;                                        // chunk_first[lo] <= c < chunk_first[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) / 2;
        if (chunk_first[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

// The path of every chunk, once per request: every workgroup (wave) of the kernels below started with this search -- fifteen dependent
// loads for 32 000 paths -- and now starts with one load (round 4: 3 % of config 4's formatting time; with the node ids of a batch asked
// for one batch ahead, 10 %: profiles/r04_gfa_pmc.txt).
__global__ void __launch_bounds__(256) k_chunk_paths(const uint64_t *chunk_first, uint64_t n, uint64_t chunks_cap, uint32_t *chunk_path) {
    const uint64_t c = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (c < chunks_cap && c < chunk_first[n]) chunk_path[c] = static_cast<uint32_t>(chunk_path_of(chunk_first, n, c));
}

__device__ __forceinline__ ChunkRange chunk_range(const uint64_t *chunk_first, const uint32_t *chunk_path, const uint64_t *offsets, uint64_t c) {
    const uint64_t lo = chunk_path[c];
    ChunkRange r;
    r.path = lo;
    r.begin = offsets[lo];
    const uint64_t end = offsets[lo + 1];
    r.lo = r.begin + (c - chunk_first[lo]) * LINE_CHUNK;
    r.hi = r.lo + LINE_CHUNK < end ? r.lo + LINE_CHUNK : end;
    r.first = c == chunk_first[lo];
    r.last = c + 1 == chunk_first[lo + 1];
    return r;
}

// One wave per chunk: text bytes of its node tokens and the summed label lengths (W-line end coordinate,
// src/bin/gbunzip.rs:532-536: sequence_len(node).unwrap_or(0)).
// The number of chunks of a request is known on the device only (chunk_first[n]); the host launches for its upper bound
// (total / LINE_CHUNK + n) and the chunks past the end count nothing, so that the scans over the bound are those over the chunks.
__global__ void __launch_bounds__(256) k_chunk_stats(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first, const uint32_t *chunk_path, uint64_t chunks_cap,
                                                      const uint32_t *label_len, uint64_t n_labels, uint32_t first_node, int p_lines, uint64_t *chunk_text,
                                                      uint64_t *chunk_seq) {
    const uint64_t c = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (c >= chunks_cap) return;
    if (c >= chunk_first[n]) { if (lane == 0) { chunk_text[c] = 0; chunk_seq[c] = 0; } return; }
    const ChunkRange r = chunk_range(chunk_first, chunk_path, offsets, c);
    uint64_t text = 0, labels = 0;
    for (uint64_t k0 = r.lo + 4 * lane; k0 < r.hi; k0 += 4 * WAVE) {   // four consecutive positions per lane: the loads and the label gathers of a round overlap
        uint32_t node[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) node[i] = k0 + i < r.hi ? nodes[k0 + i] : 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            if (k0 + i >= r.hi) continue;
            text += decimal_digits(node[i] >> 1) + 1 + ((p_lines && k0 + i > r.begin) ? 1 : 0);
            const uint64_t seq = (static_cast<uint64_t>(node[i] & ~1u) - first_node) / 2;   // GBZ::graph_node_to_sequence, src/gbz.rs:246-255
            if ((node[i] & ~1u) >= first_node && seq < n_labels) labels += label_len[seq];
        }
    }
    for (int d = WAVE / 2; d > 0; d >>= 1) { text += __shfl_down(text, d, WAVE); labels += __shfl_down(labels, d, WAVE); }
    if (lane == 0) { chunk_text[c] = text; chunk_seq[c] = labels; }
}

// Per path, from the scans over the chunks: the length of its line -- header (prefix from the table + for W-lines the end coordinate and
// its tab), node tokens, trailer -- the end coordinate itself, and for translation graphs whether every position fitted a whole segment
// (bad_before = scan of the chunks' bad flags, or null).  A flagged line gets length 0 here: the host replays it and puts its length in.
__global__ void __launch_bounds__(256) k_line_sizes(const uint64_t *seq_ids, const uint64_t *chunk_first, uint64_t n, const uint64_t *text_before,
                                                     const uint64_t *seq_before, const uint64_t *bad_before, LineHeaders hdr, int p_lines, uint64_t *line_len,
                                                     uint64_t *line_end, uint8_t *valid) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= n) return;
    const uint64_t a = chunk_first[p], b = chunk_first[p + 1];
    const uint64_t text = text_before[b] - text_before[a], labels = seq_before[b] - seq_before[a];
    const uint64_t path = seq_ids[p] >> 1;                              // the forward sequence of path p is sequence 2p (support::encode_path)
    uint64_t header = hdr.prefix_off[path + 1] - hdr.prefix_off[path], end = 0;
    if (hdr.fragment) { end = hdr.fragment[path] + labels; header += decimal_digits64(end) + 1; }
    const bool ok = bad_before == nullptr || bad_before[b] == bad_before[a];
    line_end[p] = end;
    line_len[p] = ok ? header + text + (p_lines ? 3u : 1u) : 0u;        // "\t*\n" / "\n"
    if (valid) valid[p] = ok ? 1 : 0;
}

// The line cache of the index as the kernels see it (gbwt_hip_index::lc_*): per path its first chunk, per chunk the W-line token bytes of
// the path in front of the chunk, per path {its W-line token bytes, its summed label lengths}.  A W-line token is '>' + digits, a P-line
// token digits + '+' and a ',' in front of all but the first: the P-line text in front of chunk k is the W-line text + cache_p_extra(k).
struct LineCache { const uint64_t *chunk_first; uint64_t *text; uint64_t *path; };

__host__ __device__ __forceinline__ uint64_t cache_p_extra(uint64_t k_chunk) { return k_chunk == 0 ? 0 : k_chunk * LINE_CHUNK - 1; }

// Line lengths and end coordinates from the cache: no node id is read.
__global__ void __launch_bounds__(256) k_line_sizes_cached(const uint64_t *offsets, const uint64_t *seq_ids, uint64_t n, LineCache cache, LineHeaders hdr, int p_lines,
                                                            uint64_t *line_len, uint64_t *line_end) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= n) return;
    const uint64_t path = seq_ids[p] >> 1, len = offsets[p + 1] - offsets[p];
    const uint64_t text = cache.path[2 * path] + (p_lines && len != 0 ? len - 1 : 0), labels = cache.path[2 * path + 1];
    uint64_t header = hdr.prefix_off[path + 1] - hdr.prefix_off[path], end = 0;
    if (hdr.fragment) { end = hdr.fragment[path] + labels; header += decimal_digits64(end) + 1; }
    line_end[p] = end;
    line_len[p] = header + text + (p_lines ? 3u : 1u);
}

// The header of a line, written by the workgroup of its first chunk; every chunk needs its length.
__device__ __forceinline__ uint64_t line_header(const LineHeaders &hdr, uint64_t path, const uint64_t *line_end, uint64_t row, bool write, uint8_t *line, uint32_t t,
                                                uint32_t threads) {
    const uint64_t h0 = hdr.prefix_off[path], plen = hdr.prefix_off[path + 1] - h0;
    if (write) for (uint64_t k = t; k < plen; k += threads) line[k] = hdr.prefix[h0 + k];
    if (hdr.fragment == nullptr) return plen;
    uint64_t end = line_end[row];
    const uint32_t digits = decimal_digits64(end);
    if (write && t == 0) {
        for (uint32_t d = 0; d < digits; d++) { line[plen + digits - 1 - d] = static_cast<uint8_t>('0' + end % 10u); end /= 10u; }
        line[plen + digits] = '\t';
    }
    return plen + digits + 1;
}

// What the workgroup of a chunk starts from, put together once per request by one thread per chunk: where its node ids and its text begin, how
// many positions it has, whether it opens / closes its line.  The workgroups used to find all that themselves -- four levels of dependent
// loads and the decimal length of the W-line's end coordinate in every one of their 256 threads -- before they asked for their first node id.
struct __attribute__((aligned(16))) ChunkPlan {
    uint64_t ids_at;        // index of the chunk's first position in the rows
    uint64_t text_at;       // byte of the text where its first token begins
    uint32_t count, row;    // positions; the request's row (= line) it belongs to
    uint32_t flags;         // 1: first chunk of its line (writes the header), 2: last (writes the trailer)
    uint32_t header_len;
};
static_assert(sizeof(ChunkPlan) == 32, "two 16-byte loads");

__global__ void __launch_bounds__(256) k_plan_chunks(const uint64_t *offsets, uint64_t n, const uint64_t *chunk_first, const uint32_t *chunk_path, uint64_t chunks_cap,
                                                      const uint64_t *text_before, int p_lines, const uint64_t *line_start, const uint64_t *seq_ids, LineHeaders hdr,
                                                      const uint64_t *line_end, LineCache cache, ChunkPlan *plans) {
    const uint64_t c = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (c >= chunks_cap || c >= chunk_first[n]) return;
    const ChunkRange r = chunk_range(chunk_first, chunk_path, offsets, c);
    const uint64_t path = seq_ids[r.path] >> 1;
    const uint64_t header_len = line_header(hdr, path, line_end, r.path, false, nullptr, 0, 1);
    // where the chunk's tokens start inside the line: from the scan of this request's sizing pass, or from the line cache of the index
    const uint64_t k_chunk = c - chunk_first[r.path];
    const uint64_t before = cache.text ? cache.text[cache.chunk_first[path] + k_chunk] + (p_lines ? cache_p_extra(k_chunk) : 0) : text_before[c] - text_before[chunk_first[r.path]];
    ChunkPlan plan;
    plan.ids_at = r.lo;
    plan.text_at = line_start[r.path] + header_len + before;
    plan.count = static_cast<uint32_t>(r.hi - r.lo);
    plan.row = static_cast<uint32_t>(r.path);
    plan.flags = (r.first ? 1u : 0u) | (r.last ? 2u : 0u);
    plan.header_len = static_cast<uint32_t>(header_len);
    plans[c] = plan;
}

// One workgroup per chunk: the header (first chunk of a line), the node tokens of the chunk, the trailer (last chunk).
// The tokens of 1 024 positions are put together in LDS (a block scan of their widths places them) and leave as aligned 16-byte
// stores; only the first and the last bytes of such a batch, where the text does not fill a 16-byte unit, go out one by one.
// (With every lane storing its own six bytes one at a time the formatter wrote 330 GB/s of text.)
constexpr uint32_t TOKEN_MAX = 12;   // ',' + ten digits + '+' (P-lines); '>' + ten digits (W-lines)
// PER_THREAD: consecutive positions per thread and batch (one scan and two barriers per 256 * PER_THREAD positions)
// A token is put together in registers (gfa_tokens.hpp) and OR-ed into the staging buffer as the three or four aligned dwords it covers -- the
// buffer is zero wherever no token has been placed, and whoever copies a unit out leaves it zero again.  (Rounds 1-4 made a token with a
// division and a one-byte store per character; that form was bound by exactly those -- vector ALU 83 % busy, LDS 70 %,
// profiles/r05_format_stream.txt -- and lived on behind GBWT_HIP_FORMAT_TOKENS=0 until round 6.)
template <uint32_t PER_THREAD>
__global__ void __launch_bounds__(FORMAT_THREADS) k_format_chunks(const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first, const ChunkPlan *plans, int p_lines,
                                                                   const uint64_t *seq_ids, LineHeaders hdr, const uint64_t *line_end, uint8_t *out) {
    using BlockScan = hipcub::BlockScan<uint32_t, FORMAT_THREADS, hipcub::BLOCK_SCAN_WARP_SCANS>;
    __shared__ typename BlockScan::TempStorage scan_storage;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr uint32_t BATCH = FORMAT_THREADS * PER_THREAD;
    constexpr uint32_t STAGE_BYTES = FORMAT_THREADS * PER_THREAD * TOKEN_MAX + 48;   // a batch, the bytes in front of it in its first unit, the dwords a last token spreads over
    __shared__ __attribute__((aligned(16))) uint8_t stage[STAGE_BYTES];
    if (blockIdx.x >= chunk_first[n]) return;                           // (launched for the host's upper bound of the chunk count)
    const uint32_t t = threadIdx.x;
    const ChunkPlan plan = plans[blockIdx.x];
    uint8_t *const text = out + plan.text_at;
    if (plan.flags & 1u) line_header(hdr, seq_ids[plan.row] >> 1, line_end, plan.row, true, text - plan.header_len, t, FORMAT_THREADS);
    uint64_t cursor = 0;                                                // bytes of the chunk's text written so far
    // positions counted from the start of the chunk (32-bit arithmetic in the loop): `count` of them, the path's first one at `first_at` or nowhere
    const uint32_t count = plan.count, first_at = (plan.flags & 1u) ? 0u : ~0u;
    const uint32_t *const ids = nodes + plan.ids_at;
    uint32_t ahead[PER_THREAD];                                        // the node ids of the next batch are asked for before this one is put together
#pragma unroll
    for (uint32_t i = 0; i < PER_THREAD; i++) { const uint32_t k = PER_THREAD * t + i; ahead[i] = k < count ? ids[k] : 0u; }
    for (uint32_t lo = 16 * t; lo < STAGE_BYTES; lo += 16 * FORMAT_THREADS) *reinterpret_cast<u32x4 *>(stage + lo) = u32x4{0u, 0u, 0u, 0u};   // (while the first node ids are on their way)
    __syncthreads();
    for (uint32_t base = 0; base < count; base += BATCH) {
        const uint32_t k0 = base + PER_THREAD * t;
        uint32_t node[PER_THREAD], digits[PER_THREAD], len = 0;
        Token token[PER_THREAD];
#pragma unroll
        for (uint32_t i = 0; i < PER_THREAD; i++) {
            node[i] = ahead[i]; digits[i] = 0;
            const uint32_t next = k0 + BATCH + i;
            ahead[i] = next < count ? ids[next] : 0u;
            if (k0 + i >= count) continue;
            // ',' between the tokens of a P-line and '+' / '-' behind each, '>' / '<' in front of a W-line's
            token[i] = (node[i] >> 1) < 100000000u ? make_token_short(node[i], p_lines != 0, k0 + i == first_at) : make_token(node[i], p_lines != 0, k0 + i == first_at);
            digits[i] = token[i].len;
            len += token[i].len;
        }
        uint32_t pos, total;
        BlockScan(scan_storage).ExclusiveSum(len, pos, total);
        uint8_t *const to = text + cursor;
        const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(to) & 15u);   // the batch's text lies at stage[mis ...]: LDS and memory are aligned alike
        uint8_t *w = stage + mis + pos;
#pragma unroll
        for (uint32_t i = 0; i < PER_THREAD; i++) {
            if (digits[i] == 0) continue;
            const uint32_t at = static_cast<uint32_t>(w - stage);
            uint32_t spread[4], *const dwords = reinterpret_cast<uint32_t *>(stage + (at & ~3u));
            spread_token(token[i], at & 3u, spread);
            __hip_atomic_fetch_or(dwords, spread[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_or(dwords + 1, spread[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_or(dwords + 2, spread[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (spread[3] != 0) __hip_atomic_fetch_or(dwords + 3, spread[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            w += token[i].len;
        }
        __syncthreads();
        const uint32_t end = mis + total;
        uint8_t *const aligned = to - mis;
        for (uint32_t lo = 16 * t; lo < end; lo += 16 * FORMAT_THREADS) {
            if (lo >= mis && lo + 16 <= end) {
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(stage + lo), reinterpret_cast<u32x4 *>(aligned + lo));
            } else {
                const uint32_t from = lo < mis ? mis : lo, upto = lo + 16 < end ? lo + 16 : end;
                for (uint32_t q = from; q < upto; q++) aligned[q] = stage[q];
            }
            *reinterpret_cast<u32x4 *>(stage + lo) = u32x4{0u, 0u, 0u, 0u};
        }
        cursor += total;
        __syncthreads();
    }
    // trailer: "\t*\n" for P-lines (src/bin/gbunzip.rs:476), "\n" for W-lines (:548)
    if ((plan.flags & 2u) && t == 0) {
        if (p_lines) { text[cursor] = '\t'; text[cursor + 1] = '*'; text[cursor + 2] = '\n'; }
        else text[cursor] = '\n';
    }
}

// The translation tables as the kernels see them.
struct SegmentTables {
    const uint32_t *seg_of;
    const uint32_t *seg_start;
    const uint64_t *name_off;
    const uint8_t *names;
    const uint64_t *seq_len;
    const uint8_t *node_real;
    uint64_t mapping_len;
};

// What position k of a path is: 0 = continues the segment of position k - 1, 1 = first node of a segment (a token),
// 2 = neither (the path is not a concatenation of whole segments here).
__device__ __forceinline__ uint32_t classify_position(const SegmentTables &t, const uint32_t *nodes, uint64_t begin, uint64_t k, uint32_t &segment) {
    const uint32_t node = nodes[k], id = node >> 1, rev = node & 1u;
    segment = 0;
    if (id >= t.mapping_len) return 2;
    const uint32_t s = t.seg_of[id];
    if (s == 0xFFFFFFFFu) return 2;
    segment = s;
    const uint32_t lo = t.seg_start[s], hi = t.seg_start[s + 1];
    const bool first = rev ? (id == hi - 1) : (id == lo);
    if (k == begin) return first && t.node_real[id] ? 1u : 2u;
    const uint32_t prev = nodes[k - 1], pid = prev >> 1, prev_rev = prev & 1u;
    if (pid >= t.mapping_len) return 2;
    const uint32_t ps = t.seg_of[pid];
    if (ps == 0xFFFFFFFFu) return 2;
    const bool prev_last = prev_rev ? (pid == t.seg_start[ps]) : (pid == t.seg_start[ps + 1] - 1);
    if (first && prev_last) return t.node_real[id] ? 1u : 2u;
    if (!first && !prev_last && rev == prev_rev && s == ps && id == (rev ? pid - 1 : pid + 1)) return 0;
    return 2;
}

// One wave per chunk, translation graphs: text bytes of the segment tokens, summed segment lengths, positions that fit no segment.
__global__ void __launch_bounds__(256) k_chunk_stats_segments(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first, const uint32_t *chunk_path,
                                                               uint64_t chunks_cap, SegmentTables t, int p_lines, uint64_t *chunk_text, uint64_t *chunk_seq,
                                                               uint64_t *chunk_bad) {
    const uint64_t c = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (c >= chunks_cap) return;
    if (c >= chunk_first[n]) { if (lane == 0) { chunk_text[c] = 0; chunk_seq[c] = 0; chunk_bad[c] = 0; } return; }
    const ChunkRange r = chunk_range(chunk_first, chunk_path, offsets, c);
    uint64_t text = 0, labels = 0;
    uint32_t bad = 0;
    for (uint64_t k = r.lo + lane; k < r.hi; k += WAVE) {
        uint32_t s;
        const uint32_t kind = classify_position(t, nodes, r.begin, k, s);
        if (kind == 2) bad = 1;
        if (kind == 1) {
            text += (t.name_off[s + 1] - t.name_off[s]) + 1 + ((p_lines && k > r.begin) ? 1 : 0);
            labels += t.seq_len[s];
        }
    }
    for (int d = WAVE / 2; d > 0; d >>= 1) { text += __shfl_down(text, d, WAVE); labels += __shfl_down(labels, d, WAVE); bad |= __shfl_down(bad, d, WAVE); }
    if (lane == 0) { chunk_text[c] = text; chunk_seq[c] = labels; chunk_bad[c] = bad; }
}

// One workgroup per chunk, translation graphs.  Lines of flagged paths are left to the host.
__global__ void __launch_bounds__(FORMAT_THREADS) k_format_chunks_segments(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first, const uint32_t *chunk_path,
                                                                            const uint64_t *text_before, SegmentTables t, int p_lines, const uint8_t *valid,
                                                                            const uint64_t *line_start, const uint64_t *seq_ids, LineHeaders hdr,
                                                                            const uint64_t *line_end, uint8_t *out) {
    using BlockScan = hipcub::BlockScan<uint32_t, FORMAT_THREADS>;
    __shared__ typename BlockScan::TempStorage scan_storage;
    if (blockIdx.x >= chunk_first[n]) return;
    const ChunkRange r = chunk_range(chunk_first, chunk_path, offsets, blockIdx.x);
    if (!valid[r.path]) return;
    const uint32_t tid = threadIdx.x;
    uint8_t *line = out + line_start[r.path];
    const uint64_t header_len = line_header(hdr, seq_ids[r.path] >> 1, line_end, r.path, r.first, line, tid, FORMAT_THREADS);
    uint64_t cursor = header_len + (text_before[blockIdx.x] - text_before[chunk_first[r.path]]);
    for (uint64_t base = r.lo; base < r.hi; base += FORMAT_THREADS) {
        const uint64_t k = base + tid;
        uint32_t len = 0, s = 0, name_len = 0;
        bool token = false, rev = false;
        if (k < r.hi) {
            token = classify_position(t, nodes, r.begin, k, s) == 1;
            if (token) {
                rev = (nodes[k] & 1u) != 0;
                name_len = static_cast<uint32_t>(t.name_off[s + 1] - t.name_off[s]);
                len = name_len + 1 + ((p_lines && k > r.begin) ? 1 : 0);
            }
        }
        uint32_t pos, total;
        BlockScan(scan_storage).ExclusiveSum(len, pos, total);
        if (token) {
            uint8_t *w = line + cursor + pos;
            if (p_lines) { if (k > r.begin) *w++ = ','; }
            else *w++ = rev ? '<' : '>';
            const uint8_t *name = t.names + t.name_off[s];
            for (uint32_t j = 0; j < name_len; j++) w[j] = name[j];
            if (p_lines) w[name_len] = rev ? '-' : '+';
        }
        cursor += total;
        __syncthreads();
    }
    if (r.last && tid == 0) {
        if (p_lines) { line[cursor] = '\t'; line[cursor + 1] = '*'; line[cursor + 2] = '\n'; }
        else line[cursor] = '\n';
    }
}

// ---- GBZ::segment_path as data (src/gbz.rs:477-489; SegmentPathIter 1098-1169) ------------------------------------------------------------
// One wave per chunk: the segment tokens of the chunk (positions that are the first node of a segment in the direction of travel) and
// whether some position fits no segment.
__global__ void __launch_bounds__(256) k_segment_chunk_tokens(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first, const uint32_t *chunk_path,
                                                               uint64_t chunks_cap, SegmentTables t, uint64_t *chunk_tokens, uint64_t *chunk_bad) {
    const uint64_t c = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (c >= chunks_cap) return;
    if (c >= chunk_first[n]) { if (lane == 0) { chunk_tokens[c] = 0; chunk_bad[c] = 0; } return; }
    const ChunkRange r = chunk_range(chunk_first, chunk_path, offsets, c);
    uint64_t tokens = 0;
    uint32_t bad = 0;
    for (uint64_t k = r.lo + lane; k < r.hi; k += WAVE) {
        uint32_t s;
        const uint32_t kind = classify_position(t, nodes, r.begin, k, s);
        if (kind == 2) bad = 1;
        if (kind == 1) tokens++;
    }
    for (int d = WAVE / 2; d > 0; d >>= 1) { tokens += __shfl_down(tokens, d, WAVE); bad |= __shfl_down(bad, d, WAVE); }
    if (lane == 0) { chunk_tokens[c] = tokens; chunk_bad[c] = bad; }
}

// per row: its tokens, and whether every position fitted a whole segment (else the host replays the reference's iterator on the row)
__global__ void __launch_bounds__(256) k_segment_row_tokens(const uint64_t *chunk_first, uint64_t n, const uint64_t *tokens_before, const uint64_t *bad_before, uint64_t *row_tokens, uint8_t *valid) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= n) return;
    const uint64_t a = chunk_first[p], b = chunk_first[p + 1];
    row_tokens[p] = tokens_before[b] - tokens_before[a];
    valid[p] = bad_before[b] == bad_before[a] ? 1 : 0;
}

// One workgroup per chunk: token k of a valid row = (segment << 1) | orientation, at row_start[row] + tokens of the row in front of the chunk + its rank in the chunk
__global__ void __launch_bounds__(FORMAT_THREADS) k_segment_fill_tokens(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first, const uint32_t *chunk_path,
                                                                         const uint64_t *tokens_before, SegmentTables t, const uint8_t *valid, const uint64_t *row_start, uint64_t *out) {
    using BlockScan = hipcub::BlockScan<uint32_t, FORMAT_THREADS>;
    __shared__ typename BlockScan::TempStorage scan_storage;
    if (blockIdx.x >= chunk_first[n]) return;
    const ChunkRange r = chunk_range(chunk_first, chunk_path, offsets, blockIdx.x);
    if (!valid[r.path]) return;
    uint64_t cursor = row_start[r.path] + (tokens_before[blockIdx.x] - tokens_before[chunk_first[r.path]]);
    for (uint64_t base = r.lo; base < r.hi; base += FORMAT_THREADS) {
        const uint64_t k = base + threadIdx.x;
        uint32_t s = 0, is_token = 0;
        if (k < r.hi) is_token = classify_position(t, nodes, r.begin, k, s) == 1 ? 1u : 0u;
        uint32_t pos, total;
        BlockScan(scan_storage).ExclusiveSum(is_token, pos, total);
        if (is_token) out[cursor + pos] = (static_cast<uint64_t>(s) << 1) | (nodes[k] & 1u);
        cursor += total;
        __syncthreads();
    }
}

void require_gfa_capable(const gbwt_hip_index *ix) {
    if (!(ix->caps & GBWT_HIP_OPEN_GFA)) throw InvalidData("the handle was not opened for GFA lines (GBWT_HIP_OPEN_GFA)");
    if (!ix->host.is_gbz) throw InvalidData("GFA lines need a GBZ (graph + metadata), this handle holds a bare GBWT");
    if (!ix->host.has_metadata) throw InvalidData("GFA lines need path metadata");
}

// Edge list of a record, decoded on the host (Record::decompress_edges, src/bwt.rs:378-395) for the L-lines.
bool host_edges(const HostIndex &h, uint64_t rec, std::vector<std::pair<uint64_t, uint64_t>> &edges) {
    edges.clear();
    if (rec >= h.records()) return false;
    h.ensure_records();
    const uint8_t *p = h.data.data() + h.starts[rec], *end = h.data.data() + h.starts[rec + 1];
    auto varint = [&](uint64_t &v) -> bool {
        v = 0;
        unsigned shift = 0;
        while (p < end) {
            uint8_t b = *p++;
            if (shift < 64) v += static_cast<uint64_t>(b & 0x7F) << shift;
            shift += 7;
            if (!(b & 0x80)) return true;
        }
        return false;
    };
    uint64_t sigma = 0, node = 0;
    if (!varint(sigma) || sigma == 0) return false;
    for (uint64_t e = 0; e < sigma; e++) {
        uint64_t delta, off;
        if (!varint(delta) || !varint(off)) return false;
        node += delta;
        edges.emplace_back(node, off);
    }
    return true;
}

// GBZ::has_node (src/gbz.rs:286-289): the forward GBWT node is in the alphabet and its record is not empty
bool host_has_node(const HostIndex &h, uint64_t node_id) {
    const uint64_t node = 2 * node_id, first = h.alphabet_offset + 1;
    if (node < first || node >= h.alphabet_size) return false;
    const uint64_t rec = node - h.alphabet_offset;
    h.ensure_records();
    return rec < h.records() && h.starts[rec + 1] > h.starts[rec] && h.data[h.starts[rec]] != 0;
}

// Graph::segment / node_to_segment (src/graph.rs:172-198): node range [start, end) of a segment, the segment of a node
struct HostSegment { uint64_t id, start, end; };

HostSegment host_segment(const HostIndex &h, uint64_t id) {
    return HostSegment{id, h.segment_starts[id], id + 1 < h.segment_starts.size() ? h.segment_starts[id + 1] : h.mapping_len};
}

HostSegment host_node_to_segment(const HostIndex &h, uint64_t node_id) {
    // SparseVector::predecessor: the last one at or before node_id
    const auto it = std::upper_bound(h.segment_starts.begin(), h.segment_starts.end(), node_id);
    return host_segment(h, static_cast<uint64_t>(it - h.segment_starts.begin()) - 1);
}

// Segment::sequence = labels of nodes start .. end - 1 concatenated (sequences.range(start - 1 .. end - 1))
uint64_t host_segment_seq_len(const HostIndex &h, const HostSegment &s) {
    return h.sequences_labels.offsets[s.end - 1] - h.sequences_labels.offsets[s.start - 1];
}

// SegmentPathIter (src/gbz.rs:1098-1169) over the extracted node ids of one path: the (segment, orientation) tokens up
// to the point where the reference's iterator stops, and the summed segment lengths.
void host_segment_path(const HostIndex &h, const uint32_t *nodes, uint64_t len, std::vector<std::pair<uint64_t, bool>> &tokens, uint64_t &seq_len) {
    tokens.clear();
    seq_len = 0;
    bool have_next = false, next_rev = false;
    uint64_t next_node = 0, seg_start = 0, seg_end = 0;
    for (uint64_t k = 0; k < len; k++) {
        const uint64_t node_id = nodes[k] >> 1;
        const bool rev = (nodes[k] & 1u) != 0;
        if (have_next) {
            if (node_id != next_node || rev != next_rev) return;                  // fail
        } else {
            if (!host_has_node(h, node_id) || node_id >= h.mapping_len || node_id < h.segment_starts[0]) return;   // node_to_segment -> None
            const HostSegment s = host_node_to_segment(h, node_id);
            tokens.emplace_back(s.id, rev);
            seq_len += host_segment_seq_len(h, s);
            seg_start = s.start; seg_end = s.end;
            next_node = rev ? seg_end - 1 : seg_start; next_rev = rev; have_next = true;   // visit()
        }
        if (!next_rev) { if (next_node + 1 < seg_end) next_node++; else have_next = false; }   // advance()
        else { if (next_node > seg_start) next_node--; else have_next = false; }
    }
}

SegmentTables segment_tables(const gbwt_hip_index *ix) {
    return SegmentTables{ix->seg_of.as<uint32_t>(), ix->seg_start.as<uint32_t>(), ix->seg_name_off.as<uint64_t>(), ix->seg_names.as<uint8_t>(),
                         ix->seg_seq_len.as<uint64_t>(), ix->node_real.as<uint8_t>(), ix->host.mapping_len};
}

}  // namespace

namespace gbwt_hip {

// The header of every path's line in the three line modes, up to the node tokens (path_to_p_line / path_to_pan_sn / path_to_w_line,
// src/bin/gbunzip.rs:480-524; Metadata::pan_sn_path, src/gbwt.rs:709-713; sample_name / contig_name fall back to the number,
// src/gbwt.rs:744-761, 792-809).  A W-line's end coordinate follows its fragment field and is appended on the device.
static void upload_line_headers(gbwt_hip_index &ix) {
    const HostIndex &h = ix.host;
    if (!h.has_metadata || h.path_names.empty()) return;
    const bool sample_names = (h.metadata_flags & 2) != 0, contig_names = (h.metadata_flags & 4) != 0;
    const uint64_t n = h.path_names.size();
    std::vector<uint32_t> fragment(n);
    for (uint64_t p = 0; p < n; p++) fragment[p] = h.path_names[p].fragment;
    auto build = [&](int mode) {
        std::vector<char> &out = ix.host_line_prefix[mode];
        std::vector<uint64_t> &off = ix.host_line_prefix_off[mode];
        out.clear(); out.reserve(32 * n);
        off.assign(n + 1, 0);
        auto text = [&](const char *t) { while (*t) out.push_back(*t++); };
        auto number = [&](uint64_t v) {
            char digits[24];
            int len = 0;
            do { digits[len++] = static_cast<char>('0' + v % 10); v /= 10; } while (v != 0);
            while (len > 0) out.push_back(digits[--len]);
        };
        auto name = [&](const Strings &names, bool has_names, uint64_t id) {
            if (has_names && id < names.size()) out.insert(out.end(), names.bytes.begin() + names.offsets[id], names.bytes.begin() + names.offsets[id + 1]);
            else number(id);
        };
        for (uint64_t p = 0; p < n; p++) {
            const PathName &pn = h.path_names[p];
            if (mode == 0) { text("P\t"); name(h.contig_names, contig_names, pn.contig); out.push_back('\t'); }
            else if (mode == 2) {
                text("P\t"); name(h.sample_names, sample_names, pn.sample); out.push_back('#'); number(pn.phase); out.push_back('#');
                name(h.contig_names, contig_names, pn.contig); out.push_back('\t');
            } else {
                text("W\t"); name(h.sample_names, sample_names, pn.sample); out.push_back('\t'); number(pn.phase); out.push_back('\t');
                name(h.contig_names, contig_names, pn.contig); out.push_back('\t'); number(pn.fragment); out.push_back('\t');
            }
            off[p + 1] = out.size();
        }
    };
    std::thread others[2] = {std::thread(build, 0), std::thread(build, 2)};    // (tens of thousands of walks in a config-4-shaped GBZ)
    build(1);
    for (auto &t : others) t.join();
    for (int mode = 0; mode < 3; mode++) {
        ix.line_prefix[mode].reserve(std::max<size_t>(ix.host_line_prefix[mode].size(), 16));
        ix.line_prefix_off[mode].reserve((n + 1) * sizeof(uint64_t));
        if (!ix.host_line_prefix[mode].empty())
            HIP_CHECK(hipMemcpy(ix.line_prefix[mode].ptr, ix.host_line_prefix[mode].data(), ix.host_line_prefix[mode].size(), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(ix.line_prefix_off[mode].ptr, ix.host_line_prefix_off[mode].data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    ix.line_fragment.reserve(n * sizeof(uint32_t));
    HIP_CHECK(hipMemcpy(ix.line_fragment.ptr, fragment.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
}

// Uploads the label length of every potential node; mask_label_lengths below then zeroes those of the nodes that do not exist (GBZ::has_node,
// src/gbz.rs:286-289) ON THE DEVICE, from the record bytes and starts there -- the host's image of them is not made by an open
// (HostIndex::ensure_records), and the device passes that put them there may still be running on another thread when this one is.
void upload_label_lengths(gbwt_hip_index &ix) {
    const HostIndex &h = ix.host;
    if (!h.is_gbz) return;
    std::vector<uint32_t, DefaultInitAllocator<uint32_t>> len(h.sequences_labels.size() + 1);
    len.back() = 0;
    {   // (sixteen million nodes in a config-4-shaped GBZ: a few threads)
        const uint64_t n = h.sequences_labels.size();
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned pieces = n >= (uint64_t(1) << 23) ? std::max(std::min(8u, hw), std::min(static_cast<unsigned>(std::min<uint64_t>(n >> 21, 32)), hw)) : 1u;   // (109 M nodes: 32 threads)
        auto piece = [&](unsigned p) {
            for (uint64_t s = n * p / pieces, end = n * (p + 1) / pieces; s < end; s++) len[s] = static_cast<uint32_t>(h.sequences_labels.len(s));
        };
        std::vector<std::thread> pool;
        for (unsigned p = 1; p < pieces; p++) pool.emplace_back(piece, p);
        piece(0);
        for (auto &t : pool) t.join();
    }
    ix.label_len.reserve(len.size() * sizeof(uint32_t));
    HIP_CHECK(hipMemcpy(ix.label_len.ptr, len.data(), len.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    upload_line_headers(ix);
    if (!h.has_translation || h.segment_starts.empty()) return;
    // node-to-segment translation, flattened (Graph::node_to_segment is a predecessor query on a sparse vector in the
    // reference, src/graph.rs:186-198; here every node id gets its segment)
    if (h.mapping_len >= 0xFFFFFFFFull) throw InvalidData("Graph: node ids of the translation do not fit 32 bits");
    const uint64_t n_seg = h.segment_starts.size();
    std::vector<uint32_t> seg_of(h.mapping_len, 0xFFFFFFFFu), seg_start(n_seg + 1);
    std::vector<uint64_t> seq_len(n_seg);
    std::vector<uint8_t> node_real(h.mapping_len, 0);
    for (uint64_t s = 0; s < n_seg; s++) {
        const HostSegment seg = host_segment(h, s);
        if (seg.start == 0 || seg.end > h.mapping_len || seg.end < seg.start) throw InvalidData("Graph: malformed node-to-segment mapping");
        seg_start[s] = static_cast<uint32_t>(seg.start);
        for (uint64_t v = seg.start; v < seg.end; v++) seg_of[v] = static_cast<uint32_t>(s);
        seq_len[s] = host_segment_seq_len(h, seg);
    }
    seg_start[n_seg] = static_cast<uint32_t>(h.mapping_len);
    for (uint64_t v = 0; v < h.mapping_len; v++) node_real[v] = host_has_node(h, v) ? 1 : 0;
    auto put = [](DeviceBuffer &b, const void *src, size_t bytes) {
        b.reserve(std::max<size_t>(bytes, 16));
        if (bytes) HIP_CHECK(hipMemcpy(b.ptr, src, bytes, hipMemcpyHostToDevice));
    };
    put(ix.seg_of, seg_of.data(), seg_of.size() * sizeof(uint32_t));
    put(ix.seg_start, seg_start.data(), seg_start.size() * sizeof(uint32_t));
    put(ix.seg_name_off, h.segment_names.offsets.data(), h.segment_names.offsets.size() * sizeof(uint64_t));
    put(ix.seg_names, h.segment_names.bytes.data(), h.segment_names.bytes.size());
    put(ix.seg_seq_len, seq_len.data(), seq_len.size() * sizeof(uint64_t));
    put(ix.node_real, node_real.data(), node_real.size());
}

}  // namespace gbwt_hip

namespace gbwt_hip {
void mask_label_lengths(gbwt_hip_index &ix) {
    if (!ix.host.is_gbz || ix.label_len.ptr == nullptr) return;
    launch_mask_label_lengths(ix.dev, ix.label_len.as<uint32_t>(), ix.host.sequences_labels.size(), nullptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
}
}  // namespace gbwt_hip

// The line cache of a handle, filled by ONE walk at open (kernels.hpp: LineCacheFill; open_walks.hip: k_segment_text): usable when the index
// has sequence samples (the walkers start from them, as those of an extraction do) and the graph has no node-to-segment translation
// (tokens are then segment names, sized by other rules).  Called by open_common once the device passes and the GFA tables are done, while
// the raw descriptors are still there (slow records take the generic decoder).
namespace gbwt_hip {
void fill_line_cache_at_open(gbwt_hip_index &ix) {
    const HostIndex &h = ix.host;
    ix.lc_state = -1;
    const uint64_t paths = h.path_names.size();
    const bool translated = h.has_translation && !h.segment_starts.empty();
    const char *off = std::getenv("GBWT_HIP_LINE_CACHE");
    if (!(ix.caps & GBWT_HIP_OPEN_GFA) || !h.is_gbz || !h.has_metadata || translated || paths == 0 || (off && std::atoi(off) == 0)) return;
    if (ix.host_seq_len.size() < 2 * paths || ix.sample_counts.size() < 2 * paths || 2 * paths > h.sequences) return;
    const DeviceIndex &d = ix.dev;
    if (d.samples == nullptr || d.sample_base == nullptr || d.seq_len == nullptr || d.desc2 == nullptr || (d.gblocks == nullptr && d.cblocks == nullptr)) return;
    if (ix.label_len.ptr == nullptr) return;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> first(paths + 1, 0);
    uint32_t max_samples = 0;
    for (uint64_t p = 0; p < paths; p++) {
        const uint64_t len = ix.host_seq_len[2 * p];                   // the forward sequence of path p (support::encode_path)
        first[p + 1] = first[p] + (len == 0 ? 1 : (len + LINE_CHUNK - 1) / LINE_CHUNK);
        max_samples = std::max(max_samples, ix.sample_counts[2 * p]);
    }
    if (static_cast<uint64_t>(max_samples) * paths > (uint64_t(1) << 36)) return;      // (walkers = most samples of a path x paths: beyond any launch; such a handle sizes per request)
    HIP_CHECK(hipSetDevice(ix.device));
    ix.lc_chunk_first.reserve((paths + 1) * sizeof(uint64_t));
    ix.lc_text.reserve(std::max<uint64_t>(first[paths], 1) * sizeof(uint64_t));
    ix.lc_path.reserve(2 * paths * sizeof(uint64_t));
    HIP_CHECK(hipMemcpy(ix.lc_chunk_first.ptr, first.data(), (paths + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    DeviceBuffer chunk_seg, seg_text, flags;
    chunk_seg.reserve(std::max<uint64_t>(first[paths], 1) * sizeof(uint32_t));
    seg_text.reserve(std::max<uint64_t>(ix.times.samples, 1) * 2 * sizeof(uint64_t));
    flags.reserve(sizeof(uint32_t));
    HIP_CHECK(hipMemsetAsync(flags.ptr, 0, sizeof(uint32_t), nullptr));
    HIP_CHECK(hipMemsetAsync(ix.lc_text.ptr, 0, std::max<uint64_t>(first[paths], 1) * sizeof(uint64_t), nullptr));
    HIP_CHECK(hipMemsetAsync(chunk_seg.ptr, 0, std::max<uint64_t>(first[paths], 1) * sizeof(uint32_t), nullptr));
    LineCacheFill f{};
    f.label_len = ix.label_len.as<uint32_t>(); f.n_labels = h.sequences_labels.size(); f.paths = paths; f.max_samples = max_samples;
    f.chunk_first = ix.lc_chunk_first.as<uint64_t>(); f.chunks = first[paths]; f.chunk_text = ix.lc_text.as<uint64_t>(); f.path_totals = ix.lc_path.as<uint64_t>();
    f.chunk_seg = chunk_seg.as<uint32_t>(); f.seg_text = seg_text.as<uint64_t>(); f.flags = flags.as<uint32_t>();
    launch_fill_line_cache(d, f, nullptr);
    uint32_t bad = 0;
    HIP_CHECK(hipMemcpy(&bad, flags.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipGetLastError());
    if (bad != 0) {                                                    // samples that promise more nodes than the walk delivers: requests size their lines themselves
        ix.lc_chunk_first.release(); ix.lc_text.release(); ix.lc_path.release();
        return;
    }
    ix.lc_state = 1;
    ix.times.line_sizes_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace gbwt_hip

// The lines of a batch of paths, formatted ONCE into device memory (the text buffer of `slot`: ws->gfa_text or gfa_text2; line k at
// [line_start[k], line_start[k + 1]), offsets also on the device in ws->gfa_b).  The request is remembered in the workspace: the fill call that
// follows a size query, and the copy-out of gbwt_hip_path_lines after gbwt_hip_path_lines_device, find the text there.
//
// Everything between the walk and the text stays on the device (round 4): chunk counts, per-chunk sizes, their scans, the line lengths --
// headers included, from the per-path header tables built at open and the end coordinate the walk yields -- and the scan that places the
// lines.  The host waits ONCE, for the total that sizes the text buffer (and, with a node-to-segment translation, for the flags of the
// paths it has to replay).  Until round 3 it built every header itself, between two more synchronisations: 0.9 ms per request whatever its
// size -- config 4's 32 000 walks in rounds of 512 spent 58 ms there.
static gbwt_hip_status path_lines_compute(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *path_ids, uint64_t n, int mode, int slot = 0) {
    if (!ix || !ws || ws->index != ix) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace");
    if (n && !path_ids) return fail(GBWT_HIP_BAD_ARGUMENT, "null path_ids");
    if (mode < 0 || mode > 2) return fail(GBWT_HIP_BAD_ARGUMENT, "mode must be 0 (P-lines), 1 (W-lines) or 2 (P-lines with PanSN names)");
    const int p_lines = mode != 1 ? 1 : 0;
    if (ws->lines_cached && ws->lines_mode == mode && ws->lines_slot == slot && ws->lines_key.size() == n &&
        (n == 0 || std::memcmp(ws->lines_key.data(), path_ids, n * sizeof(uint64_t)) == 0))
        return GBWT_HIP_OK;
    ws->lines_cached = false;
    try {
        require_gfa_capable(ix);
        const HostIndex &h = ix->host;
        const bool translated = h.has_translation && !h.segment_starts.empty();
        for (uint64_t k = 0; k < n; k++)
            if (path_ids[k] >= h.path_names.size()) return fail(GBWT_HIP_BAD_ARGUMENT, "path id out of range");
        ws->lines_total = 0;
        if (n == 0) {
            ws->lines_key.clear(); ws->lines_mode = mode; ws->lines_slot = slot; ws->lines_cached = true;
            return GBWT_HIP_OK;
        }
        // 1. forward sequences of the paths (GBZ::path(id, Forward), src/bin/gbunzip.rs:462, 532)
        std::vector<uint64_t> seq_ids(n);
        for (uint64_t k = 0; k < n; k++) seq_ids[k] = 2 * path_ids[k];
        gbwt_hip_paths paths{};
        gbwt_hip_status st = gbwt_hip_extract_device(ix, ws, seq_ids.data(), n, &paths);
        if (st != GBWT_HIP_OK) return st;
        HIP_CHECK(hipSetDevice(ix->device));
        hipStream_t s = ws->stream;
        for (auto &e : ws->gev) if (!e) HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventRecord(ws->gev[0], s));
        const uint64_t *d_seq_ids = ws->seq_ids.as<uint64_t>();         // (the extraction has left the ids on the device)
        // 2. size of every line: chunks of LINE_CHUNK positions, a wave per chunk, scans over the chunks.  Every row has at least one chunk
        // and at most len / LINE_CHUNK + 1: the launches and scans below run over that bound, the kernels read the count from the device.
        const uint64_t chunks_cap = paths.total / LINE_CHUNK + n;
        if (chunks_cap > 0x7FFFFFFFull) return fail(GBWT_HIP_UNSUPPORTED, "too many line chunks in one batch: format fewer paths per call");
        const size_t tb = scan_temp_bytes(std::max(n, chunks_cap));
        ws->gfa_a.reserve(2 * n * sizeof(uint64_t));
        ws->gfa_b.reserve((n + 1) * sizeof(uint64_t));
        ws->gfa_chunk_first.reserve(2 * (n + 1) * sizeof(uint64_t));
        ws->gfa_chunks.reserve((3 * chunks_cap + 3 * (chunks_cap + 1)) * sizeof(uint64_t) + (chunks_cap + 1) * sizeof(uint32_t));
        ws->scan_temp.reserve(std::max<size_t>(tb, 16));
        ws->gfa_valid.reserve(std::max<uint64_t>(n, 16));
        uint64_t *d_line_len = ws->gfa_a.as<uint64_t>(), *d_line_end = d_line_len + n, *d_line_start = ws->gfa_b.as<uint64_t>();
        uint64_t *d_chunk_first = ws->gfa_chunk_first.as<uint64_t>(), *d_chunk_counts = d_chunk_first + (n + 1);
        uint64_t *d_chunk_text = ws->gfa_chunks.as<uint64_t>(), *d_chunk_seq = d_chunk_text + chunks_cap, *d_chunk_bad = d_chunk_seq + chunks_cap;
        uint64_t *d_text_before = d_chunk_bad + chunks_cap, *d_seq_before = d_text_before + (chunks_cap + 1), *d_bad_before = d_seq_before + (chunks_cap + 1);
        uint32_t *d_chunk_path = reinterpret_cast<uint32_t *>(d_bad_before + (chunks_cap + 1));
        uint8_t *d_valid = translated ? ws->gfa_valid.as<uint8_t>() : nullptr;
        const LineHeaders hdr{ix->line_prefix[mode].as<uint8_t>(), ix->line_prefix_off[mode].as<uint64_t>(), mode == 1 ? ix->line_fragment.as<uint32_t>() : nullptr};
        // the line cache of the index (filled at open): nothing below reads a node id before the formatter does
        const bool all_cached = !translated && ix->lc_state == 1;
        const LineCache cache{ix->lc_chunk_first.as<uint64_t>(), ix->lc_text.as<uint64_t>(), ix->lc_path.as<uint64_t>()};
        hipLaunchKernelGGL(k_chunk_counts, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, paths.d_offsets, n, d_chunk_counts);
        launch_scan(d_chunk_counts, d_chunk_first, n, ws->scan_temp.ptr, tb, s);
        hipLaunchKernelGGL(k_chunk_paths, dim3(static_cast<unsigned>((chunks_cap + 255) / 256)), dim3(256), 0, s, d_chunk_first, n, chunks_cap, d_chunk_path);
        const unsigned stat_blocks = static_cast<unsigned>((chunks_cap + 3) / 4);
        if (all_cached) {
            hipLaunchKernelGGL(k_line_sizes_cached, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, paths.d_offsets, d_seq_ids, n, cache, hdr, p_lines, d_line_len, d_line_end);
        } else {
            if (translated) {
                hipLaunchKernelGGL(k_chunk_stats_segments, dim3(stat_blocks), dim3(256), 0, s, paths.d_offsets, paths.d_nodes, n, d_chunk_first, d_chunk_path, chunks_cap,
                                   segment_tables(ix), p_lines, d_chunk_text, d_chunk_seq, d_chunk_bad);
                launch_scan(d_chunk_bad, d_bad_before, chunks_cap, ws->scan_temp.ptr, tb, s);
            } else {
                hipLaunchKernelGGL(k_chunk_stats, dim3(stat_blocks), dim3(256), 0, s, paths.d_offsets, paths.d_nodes, n, d_chunk_first, d_chunk_path, chunks_cap,
                                   ix->label_len.as<uint32_t>(), static_cast<uint64_t>(h.sequences_labels.size()),
                                   static_cast<uint32_t>(h.alphabet_offset + 1), p_lines, d_chunk_text, d_chunk_seq);
            }
            launch_scan(d_chunk_text, d_text_before, chunks_cap, ws->scan_temp.ptr, tb, s);
            launch_scan(d_chunk_seq, d_seq_before, chunks_cap, ws->scan_temp.ptr, tb, s);
            hipLaunchKernelGGL(k_line_sizes, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, d_seq_ids, d_chunk_first, n, d_text_before, d_seq_before,
                               translated ? d_bad_before : nullptr, hdr, p_lines, d_line_len, d_line_end, d_valid);
        }
        launch_scan(d_line_len, d_line_start, n, ws->scan_temp.ptr, tb, s);
        uint64_t total = 0;
        std::vector<uint8_t> valid;
        HIP_CHECK(hipMemcpyAsync(&total, d_line_start + n, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        if (translated) { valid.resize(n); HIP_CHECK(hipMemcpyAsync(valid.data(), d_valid, n, hipMemcpyDeviceToHost, s)); }
        HIP_CHECK(hipStreamSynchronize(s));                             // the one wait of a request
        HIP_CHECK(hipGetLastError());
        // 2b. paths that are not concatenations of whole segments: the reference's iterator stops somewhere inside them; the host replays
        // it on the extracted node ids, puts the lengths of those lines in and places the lines again
        std::vector<std::string> host_lines;
        std::vector<uint64_t> line_start;
        if (translated && std::find(valid.begin(), valid.end(), uint8_t(0)) != valid.end()) {
            host_lines.resize(n);
            std::vector<uint64_t> offs(n + 1), lens(n);
            HIP_CHECK(hipMemcpy(offs.data(), paths.d_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(lens.data(), d_line_len, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
            std::vector<uint32_t> nodes;
            std::vector<std::pair<uint64_t, bool>> tokens;
            const std::vector<char> &prefix = ix->host_line_prefix[mode];
            const std::vector<uint64_t> &prefix_off = ix->host_line_prefix_off[mode];
            for (uint64_t k = 0; k < n; k++) {
                if (valid[k]) continue;
                nodes.resize(offs[k + 1] - offs[k]);
                if (!nodes.empty()) HIP_CHECK(hipMemcpy(nodes.data(), paths.d_nodes + offs[k], nodes.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                uint64_t seq_len = 0;
                host_segment_path(h, nodes.data(), nodes.size(), tokens, seq_len);
                std::string &line = host_lines[k];
                line.assign(prefix.data() + prefix_off[path_ids[k]], prefix_off[path_ids[k] + 1] - prefix_off[path_ids[k]]);
                if (mode == 1) line += std::to_string(static_cast<uint64_t>(h.path_names[path_ids[k]].fragment) + seq_len) + "\t";
                for (size_t j = 0; j < tokens.size(); j++) {
                    const std::string name = h.segment_names.str(tokens[j].first);
                    if (p_lines) line += (j ? "," : "") + name + (tokens[j].second ? "-" : "+");
                    else line += (tokens[j].second ? "<" : ">") + name;
                }
                line += p_lines ? "\t*\n" : "\n";
                lens[k] = line.size();
            }
            line_start.assign(n + 1, 0);
            for (uint64_t k = 0; k < n; k++) line_start[k + 1] = line_start[k] + lens[k];
            total = line_start[n];
            HIP_CHECK(hipMemcpyAsync(d_line_start, line_start.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        }
        // 3. format on the device; the lines the host had to replay are copied into their places
        DeviceBuffer &text = slot == 0 ? ws->gfa_text : ws->gfa_text2;
        text.reserve(std::max<uint64_t>(total, 16));
        if (translated)
            hipLaunchKernelGGL(k_format_chunks_segments, dim3(static_cast<unsigned>(chunks_cap)), dim3(FORMAT_THREADS), 0, s, paths.d_offsets, paths.d_nodes, n,
                               d_chunk_first, d_chunk_path, d_text_before, segment_tables(ix), p_lines, d_valid, d_line_start, d_seq_ids, hdr, d_line_end, text.as<uint8_t>());
        else {
            const auto kernel = k_format_chunks<4>;   // (eight positions per thread: five waves per SIMD, 23 % slower)
            ws->gfa_plan.reserve(chunks_cap * sizeof(ChunkPlan));
            hipLaunchKernelGGL(k_plan_chunks, dim3(static_cast<unsigned>((chunks_cap + 255) / 256)), dim3(256), 0, s, paths.d_offsets, n, d_chunk_first, d_chunk_path, chunks_cap,
                               d_text_before, p_lines, d_line_start, d_seq_ids, hdr, d_line_end, all_cached ? cache : LineCache{nullptr, nullptr, nullptr}, ws->gfa_plan.as<ChunkPlan>());
            hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(chunks_cap)), dim3(FORMAT_THREADS), 0, s, paths.d_nodes, n, d_chunk_first, ws->gfa_plan.as<ChunkPlan>(), p_lines,
                               d_seq_ids, hdr, d_line_end, text.as<uint8_t>());
        }
        HIP_CHECK(hipGetLastError());
        for (uint64_t k = 0; k < host_lines.size(); k++)
            if (!valid[k] && !host_lines[k].empty())
                HIP_CHECK(hipMemcpyAsync(text.as<char>() + line_start[k], host_lines[k].data(), host_lines[k].size(), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipEventRecord(ws->gev[1], s));
        HIP_CHECK(hipStreamSynchronize(s));   // the text is there when the call returns (and host_lines / line_start go out of scope)
        ws->lines_timed = true;
        ws->lines_total = total;
        ws->lines_key.assign(path_ids, path_ids + n);
        ws->lines_mode = mode;
        ws->lines_slot = slot;
        ws->lines_cached = true;
        return GBWT_HIP_OK;
    } catch (const InvalidData &e) {
        return fail(GBWT_HIP_BAD_ARGUMENT, e.what());
    } catch (const HipError &e) {
        return status_of(e);
    }
}

// gbwt_hip_path_lines: formats (or finds) the lines and copies them to the host buffer -- over several threads with pinned staging
// buffers (copy_to_host), as the CSR of gbwt_hip_extract is: one pageable hipMemcpy moved a gigabyte of W-lines at 2.4 GB/s.
static gbwt_hip_status path_lines_impl(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *path_ids, uint64_t n, int mode,
                                       char *out, uint64_t capacity, uint64_t *total) {
    if (!total) return fail(GBWT_HIP_BAD_ARGUMENT, "null total");
    *total = 0;
    const gbwt_hip_status st = path_lines_compute(ix, ws, path_ids, n, mode);
    if (st != GBWT_HIP_OK) return st;
    *total = ws->lines_total;
    if (!out || *total == 0) return GBWT_HIP_OK;
    if (capacity < *total) return fail(GBWT_HIP_CAPACITY, "output capacity too small for the lines");
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        copy_to_host(ws, out, ws->gfa_text.ptr, *total);
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
}

namespace {

// A file written at positions: every producer knows (or is told, in order) where its bytes go, so several threads write at once --
// one thread moved config 4's 4.5 GB to /dev/shm at 2.4 GB/s, most of it spent in the page cache's per-page work.
struct PositionalFile {
    int fd = -1;
    std::atomic<int> failed{0};
    ~PositionalFile() { if (fd >= 0) ::close(fd); }
    bool write_at(const char *data, size_t bytes, uint64_t at) {
        while (bytes != 0) {
            const ssize_t w = ::pwrite(fd, data, bytes, static_cast<off_t>(at));
            if (w <= 0) { failed = 1; return false; }
            data += w; bytes -= static_cast<size_t>(w); at += static_cast<uint64_t>(w);
        }
        return true;
    }
};

// H-, S- and L-lines of the whole graph (write_gfa_header / write_segments / write_links, src/bin/gbunzip.rs:193-317).  Serial host work in
// the reference; here the node ids are cut into ranges that a few threads turn into text side by side (the S-lines of all ranges, then
// the L-lines of all ranges: the order of the file).  A range's place in the file is behind the ranges before it, so a thread that has
// its text waits for its turn to take the next stretch of the file (a counter, no I/O under it) and then writes it on its own:
// config 4's 462 MB of S- and L-lines: 1 090 ms on one thread.  At most `threads` ranges of text exist at a time.  Returns the bytes written
// from `cursor` on.  Graphs with a node-to-segment translation (segment names, links between segments) take one thread, as before.
struct GraphLineRanges {
    const HostIndex &h;
    PositionalFile &file;
    uint64_t cursor = 0;                   // next free byte of the file
    GraphLineRanges(const HostIndex &index, PositionalFile &f) : h(index), file(f) {}
    std::mutex turn_lock;
    std::condition_variable turn_cv;
    uint64_t turn = 0;                     // the range whose text goes next

    template <class Make>
    void phase(uint64_t items, uint64_t per_range, unsigned threads, Make make) {
        const uint64_t ranges = (items + per_range - 1) / per_range;
        std::atomic<uint64_t> next{0};
        turn = 0;
        auto work = [&]() {
            std::string text;
            for (uint64_t r = next++; r < ranges; r = next++) {
                text.clear();
                make(r * per_range, std::min(items, (r + 1) * per_range), text);
                uint64_t at;
                {
                    std::unique_lock<std::mutex> lock(turn_lock);
                    turn_cv.wait(lock, [&] { return turn == r; });
                    at = cursor; cursor += text.size(); turn = r + 1;
                }
                turn_cv.notify_all();
                (void)file.write_at(text.data(), text.size(), at);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
};

void append_number(std::string &text, uint64_t v) {
    char digits[24];
    int len = 0;
    do { digits[len++] = static_cast<char>('0' + v % 10); v /= 10; } while (v != 0);
    while (len > 0) text.push_back(digits[--len]);
}

uint64_t host_graph_lines(const HostIndex &h, bool translated, PositionalFile &file) {
    GraphLineRanges out(h, file);
    {   // header (write_gfa_header, src/bin/gbunzip.rs:193-203)
        std::string head;
        if (const std::string *rs = h.tag("reference_samples")) head = "H\tVN:Z:1.1\tRS:Z:" + *rs + "\n";
        else head = "H\tVN:Z:1.1\n";
        (void)file.write_at(head.data(), head.size(), 0);
        out.cursor = head.size();
    }
    // segments + links over the real nodes (write_segments / write_links, src/bin/gbunzip.rs:230-317)
    const uint64_t first = h.alphabet_offset + 1, potential = h.sequences_labels.size();
    h.ensure_records();                                            // the graph lines read the host's image of the records (made now if this is its first use)
    auto real = [&](uint64_t seq) {
        const uint64_t rec = 2 * seq + first - h.alphabet_offset;
        return rec < h.records() && h.starts[rec + 1] > h.starts[rec] && h.data[h.starts[rec]] != 0;
    };
    const unsigned threads = potential >= (uint64_t(1) << 18) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u;
    if (translated) {
        // GBZ::segment_iter keeps the segments whose first node exists (src/gbz.rs:927-929); links go from the last
        // node of the segment in the orientation of travel (segment_successors, src/gbz.rs:402-415) and are named by
        // the segment of the successor; LinkIter ends at a successor without a segment (src/gbz.rs:996-999)
        const uint64_t n_seg = h.segment_starts.size();
        out.phase(n_seg, uint64_t(1) << 15, threads, [&](uint64_t lo, uint64_t hi, std::string &text) {
            for (uint64_t id = lo; id < hi; id++) {
                const HostSegment seg = host_segment(h, id);
                if (!host_has_node(h, seg.start)) continue;
                text += "S\t"; text += h.segment_names.str(id); text += "\t";
                text.append(reinterpret_cast<const char *>(h.sequences_labels.bytes.data()) + h.sequences_labels.offsets[seg.start - 1], host_segment_seq_len(h, seg));
                text += "\n";
            }
        });
        out.phase(n_seg, uint64_t(1) << 15, threads, [&](uint64_t lo, uint64_t hi, std::string &text) {
            std::vector<std::pair<uint64_t, uint64_t>> edges;
            for (uint64_t id = lo; id < hi; id++) {
                const HostSegment seg = host_segment(h, id);
                if (!host_has_node(h, seg.start)) continue;
                const std::string name = h.segment_names.str(id);
                for (int rev = 0; rev < 2; rev++) {
                    const uint64_t node_id = rev ? seg.start : seg.end - 1;
                    if (!host_has_node(h, node_id) || !host_edges(h, 2 * node_id + rev - h.alphabet_offset, edges)) continue;
                    for (auto &e : edges) {
                        if (e.first == 0) continue;                   // EdgeIter skips the ENDMARKER edge (src/gbz.rs:833)
                        const uint64_t succ = e.first / 2;
                        const bool succ_rev = (e.first & 1) != 0;
                        if (!host_has_node(h, succ) || succ >= h.mapping_len || succ < h.segment_starts[0]) break;
                        const HostSegment to = host_node_to_segment(h, succ);
                        const bool canonical = rev ? (to.id > id || (to.id == id && !succ_rev)) : (to.id >= id);
                        if (!canonical) continue;
                        text += "L\t" + name + (rev ? "\t-\t" : "\t+\t") + h.segment_names.str(to.id) + (succ_rev ? "\t-\t*\n" : "\t+\t*\n");
                    }
                }
            }
        });
        return out.cursor;
    }
    out.phase(potential, uint64_t(1) << 16, threads, [&](uint64_t lo, uint64_t hi, std::string &text) {
        for (uint64_t seq = lo; seq < hi; seq++) {
            if (!real(seq)) continue;
            text += "S\t"; append_number(text, (2 * seq + first) / 2); text.push_back('\t');
            text.append(reinterpret_cast<const char *>(h.sequences_labels.bytes.data()) + h.sequences_labels.offsets[seq], h.sequences_labels.len(seq));
            text.push_back('\n');
        }
    });
    out.phase(potential, uint64_t(1) << 16, threads, [&](uint64_t lo, uint64_t hi, std::string &text) {
        std::vector<std::pair<uint64_t, uint64_t>> edges;
        for (uint64_t seq = lo; seq < hi; seq++) {
            if (!real(seq)) continue;
            const uint64_t node_id = (2 * seq + first) / 2;
            for (int rev = 0; rev < 2; rev++) {
                if (!host_edges(h, 2 * node_id + rev - h.alphabet_offset, edges)) continue;
                for (auto &e : edges) {
                    if (e.first == 0) continue;                       // EdgeIter skips the ENDMARKER edge (src/gbz.rs:833)
                    const uint64_t succ = e.first / 2;
                    const bool succ_rev = (e.first & 1) != 0;
                    const bool canonical = rev ? (succ > node_id || (succ == node_id && !succ_rev)) : (succ >= node_id);
                    if (!canonical) continue;
                    text += "L\t"; append_number(text, node_id); text += rev ? "\t-\t" : "\t+\t"; append_number(text, succ); text += succ_rev ? "\t-\t*\n" : "\t+\t*\n";
                }
            }
        }
    });
    return out.cursor;
}

// The writer's side of a whole-file write: one thread that first puts out the graph lines (which precede the paths in the file; see
// host_graph_lines), then takes finished batches of path lines -- device text -- and moves them to the file in pieces of 32 MiB: the
// piece travels into one of four pinned buffers (the copy of the next piece runs under whatever happens to this one) and a small pool of
// threads writes the buffers at their positions.  The main thread formats the next batch into the other device text buffer meanwhile.
struct GfaWriter {
    static constexpr size_t PIECE = size_t(32) << 20;
    int WRITERS = 3, BUFFERS = 5;            // GBWT_HIP_GFA_WRITERS (1 .. 16); two more pinned buffers than writing threads
    struct Job { const char *text; uint64_t bytes; int slot; };
    struct Piece { int buffer; uint64_t bytes, at; };
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> jobs;
    std::deque<Piece> pieces;              // pinned buffers that hold a piece on its way to the file
    std::vector<char> buffer_busy;
    bool closing = false, no_more_pieces = false, slot_busy[2] = {false, false};
    gbwt_hip_status status = GBWT_HIP_OK;
    std::string message;
    std::thread worker;
    PositionalFile *file = nullptr;
    const HostIndex *host = nullptr;
    bool translated = false;
    int device = 0;

    void fail_with(gbwt_hip_status st, const std::string &msg) {
        std::lock_guard<std::mutex> lock(m);
        if (status == GBWT_HIP_OK) { status = st; message = msg; }
        cv.notify_all();
    }
    void run() {
        if (const char *v = std::getenv("GBWT_HIP_GFA_WRITERS")) WRITERS = std::min(16, std::max(1, std::atoi(v)));
        BUFFERS = WRITERS + 2;
        std::vector<void *> pinned(BUFFERS, nullptr);
        buffer_busy.assign(BUFFERS, 0);
        hipStream_t stream = nullptr;
        const bool trace = std::getenv("GBWT_HIP_TRACE_GFA") != nullptr;       // phases of a whole-file write on stderr
        const auto t0 = std::chrono::steady_clock::now();
        const auto since = [&t0]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        uint64_t path_bytes = 0;
        std::vector<std::thread> pool;
        try {
            uint64_t cursor = host_graph_lines(*host, translated, *file);
            if (file->failed) throw std::runtime_error("short write");
            if (trace) std::fprintf(stderr, "[gfa] H/S/L lines: %llu bytes generated and written in %.1f ms\n", static_cast<unsigned long long>(cursor), since());
            HIP_CHECK(hipSetDevice(device));
            HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            for (int i = 0; i < BUFFERS; i++) HIP_CHECK(hipHostMalloc(&pinned[i], PIECE, hipHostMallocDefault));
            for (int t = 0; t < WRITERS; t++)
                pool.emplace_back([this, &pinned]() {
                    for (;;) {
                        Piece p;
                        {
                            std::unique_lock<std::mutex> lock(m);
                            cv.wait(lock, [&] { return !pieces.empty() || no_more_pieces; });
                            if (pieces.empty()) return;
                            p = pieces.front(); pieces.pop_front();
                        }
                        if (!file->write_at(static_cast<const char *>(pinned[p.buffer]), p.bytes, p.at)) fail_with(GBWT_HIP_IO_ERROR, "short write");
                        { std::lock_guard<std::mutex> lock(m); buffer_busy[p.buffer] = 0; }
                        cv.notify_all();
                    }
                });
            for (;;) {
                Job job;
                {
                    std::unique_lock<std::mutex> lock(m);
                    cv.wait(lock, [&] { return !jobs.empty() || closing || status != GBWT_HIP_OK; });
                    if (status != GBWT_HIP_OK || jobs.empty()) break;
                    job = jobs.front(); jobs.pop_front();
                }
                path_bytes += job.bytes;
                for (uint64_t done = 0; done < job.bytes; done += PIECE) {
                    int b = -1;
                    {
                        std::unique_lock<std::mutex> lock(m);
                        cv.wait(lock, [&] { for (int i = 0; i < BUFFERS; i++) if (!buffer_busy[i]) return true; return status != GBWT_HIP_OK; });
                        if (status != GBWT_HIP_OK) break;
                        for (int i = 0; i < BUFFERS; i++) if (!buffer_busy[i]) { b = i; break; }
                        buffer_busy[b] = 1;
                    }
                    const uint64_t len = std::min<uint64_t>(PIECE, job.bytes - done);
                    HIP_CHECK(hipMemcpyAsync(pinned[b], job.text + done, len, hipMemcpyDeviceToHost, stream));
                    HIP_CHECK(hipStreamSynchronize(stream));
                    { std::lock_guard<std::mutex> lock(m); pieces.push_back(Piece{b, len, cursor + done}); }
                    cv.notify_all();
                }
                cursor += job.bytes;
                {   // the device text of this batch has left: the formatter may have the slot back
                    std::lock_guard<std::mutex> lock(m);
                    slot_busy[job.slot] = false;
                }
                cv.notify_all();
            }
        } catch (const HipError &e) {
            fail_with(GBWT_HIP_DEVICE_ERROR, std::string(e.what) + ": " + hipGetErrorString(e.err));
        } catch (const std::exception &e) {
            fail_with(GBWT_HIP_IO_ERROR, std::string("GFA writer: ") + e.what());
        }
        { std::lock_guard<std::mutex> lock(m); no_more_pieces = true; }
        cv.notify_all();
        for (auto &t : pool) t.join();
        if (trace) std::fprintf(stderr, "[gfa] path lines: %llu bytes; writer threads done at %.1f ms\n", static_cast<unsigned long long>(path_bytes), since());
        for (int i = 0; i < BUFFERS; i++) if (pinned[i]) (void)hipHostFree(pinned[i]);
        if (stream) (void)hipStreamDestroy(stream);
        std::lock_guard<std::mutex> lock(m);
        closing = true; slot_busy[0] = slot_busy[1] = false;
        cv.notify_all();
    }
    // the formatter's side: wait until the device text buffer of `slot` has been read out; false when the writer has failed
    bool acquire(int slot) {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return !slot_busy[slot] || status != GBWT_HIP_OK; });
        if (status != GBWT_HIP_OK) return false;
        slot_busy[slot] = true;
        return true;
    }
    void submit(const char *text, uint64_t bytes, int slot) {
        { std::lock_guard<std::mutex> lock(m); jobs.push_back(Job{text, bytes, slot}); }
        cv.notify_all();
    }
    gbwt_hip_status finish() {
        { std::lock_guard<std::mutex> lock(m); closing = true; }
        cv.notify_all();
        if (worker.joinable()) worker.join();
        return status;
    }
    ~GfaWriter() { (void)finish(); }
};

}  // namespace

extern "C" {

gbwt_hip_status gbwt_hip_path_lines(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *path_ids, uint64_t n, int mode,
                                    char *out, uint64_t capacity, uint64_t *total) {
    GBWT_HIP_GUARD_BEGIN
    return path_lines_impl(ix, ws, path_ids, n, mode, out, capacity, total);
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_path_lines_device(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *path_ids, uint64_t n, int mode,
                                           gbwt_hip_lines *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!out) return fail(GBWT_HIP_BAD_ARGUMENT, "null output");
    *out = gbwt_hip_lines{nullptr, nullptr, 0, 0};
    const gbwt_hip_status st = path_lines_compute(ix, ws, path_ids, n, mode);
    if (st != GBWT_HIP_OK) return st;
    out->d_text = n ? ws->gfa_text.as<char>() : nullptr;
    out->d_line_offsets = n ? ws->gfa_b.as<uint64_t>() : nullptr;
    out->total = ws->lines_total;
    out->n = n;
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_segment_paths(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *seq_ids, uint64_t n, uint64_t *out_offsets, uint64_t *out_tokens,
                                       uint64_t capacity, uint64_t *total) {
    GBWT_HIP_GUARD_BEGIN
    if (!ix || !ws || ws->index != ix || !total) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace / total");
    if (n && (!seq_ids || !out_offsets)) return fail(GBWT_HIP_BAD_ARGUMENT, "null seq_ids / out_offsets");
    *total = 0;
    const HostIndex &h = ix->host;
    if (!(ix->caps & GBWT_HIP_OPEN_GFA)) return fail(GBWT_HIP_BAD_ARGUMENT, "the handle was not opened for GFA lines (GBWT_HIP_OPEN_GFA): the node-to-segment tables belong to that group");
    if (!h.is_gbz || !h.has_translation || h.segment_starts.empty())
        return fail(GBWT_HIP_BAD_ARGUMENT, "no node-to-segment translation (GBZ::segment_path returns None, src/gbz.rs:477-480)");
    if (out_offsets) out_offsets[0] = 0;
    if (n == 0) return GBWT_HIP_OK;
    try {
        gbwt_hip_paths paths{};
        const gbwt_hip_status st = gbwt_hip_extract_device(ix, ws, seq_ids, n, &paths);
        if (st != GBWT_HIP_OK) return st;
        ws->lines_cached = false;                                        // (the line buffers of the workspace are used below)
        HIP_CHECK(hipSetDevice(ix->device));
        hipStream_t s = ws->stream;
        const uint64_t chunks_cap = paths.total / LINE_CHUNK + n;
        if (chunks_cap > 0x7FFFFFFFull) return fail(GBWT_HIP_UNSUPPORTED, "too many chunks in one batch: ask for fewer paths per call");
        const size_t tb = scan_temp_bytes(std::max(n, chunks_cap));
        ws->gfa_a.reserve(2 * (n + 1) * sizeof(uint64_t));
        ws->gfa_chunk_first.reserve(2 * (n + 1) * sizeof(uint64_t));
        ws->gfa_chunks.reserve((2 * chunks_cap + 2 * (chunks_cap + 1)) * sizeof(uint64_t) + (chunks_cap + 1) * sizeof(uint32_t));
        ws->scan_temp.reserve(std::max<size_t>(tb, 16));
        ws->gfa_valid.reserve(std::max<uint64_t>(n, 16));
        uint64_t *d_row_tokens = ws->gfa_a.as<uint64_t>(), *d_row_start = d_row_tokens + (n + 1);
        uint64_t *d_chunk_first = ws->gfa_chunk_first.as<uint64_t>(), *d_chunk_counts = d_chunk_first + (n + 1);
        uint64_t *d_chunk_tokens = ws->gfa_chunks.as<uint64_t>(), *d_chunk_bad = d_chunk_tokens + chunks_cap, *d_tokens_before = d_chunk_bad + chunks_cap,
                 *d_bad_before = d_tokens_before + (chunks_cap + 1);
        uint32_t *d_chunk_path = reinterpret_cast<uint32_t *>(d_bad_before + (chunks_cap + 1));
        uint8_t *d_valid = ws->gfa_valid.as<uint8_t>();
        const SegmentTables tables = segment_tables(ix);
        hipLaunchKernelGGL(k_chunk_counts, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, paths.d_offsets, n, d_chunk_counts);
        launch_scan(d_chunk_counts, d_chunk_first, n, ws->scan_temp.ptr, tb, s);
        hipLaunchKernelGGL(k_chunk_paths, dim3(static_cast<unsigned>((chunks_cap + 255) / 256)), dim3(256), 0, s, d_chunk_first, n, chunks_cap, d_chunk_path);
        hipLaunchKernelGGL(k_segment_chunk_tokens, dim3(static_cast<unsigned>((chunks_cap + 3) / 4)), dim3(256), 0, s, paths.d_offsets, paths.d_nodes, n, d_chunk_first, d_chunk_path, chunks_cap,
                           tables, d_chunk_tokens, d_chunk_bad);
        launch_scan(d_chunk_tokens, d_tokens_before, chunks_cap, ws->scan_temp.ptr, tb, s);
        launch_scan(d_chunk_bad, d_bad_before, chunks_cap, ws->scan_temp.ptr, tb, s);
        hipLaunchKernelGGL(k_segment_row_tokens, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, d_chunk_first, n, d_tokens_before, d_bad_before, d_row_tokens, d_valid);
        std::vector<uint64_t> counts(n), offs(n + 1);
        std::vector<uint8_t> valid(n);
        HIP_CHECK(hipMemcpyAsync(counts.data(), d_row_tokens, n * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(valid.data(), d_valid, n, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(offs.data(), paths.d_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        HIP_CHECK(hipGetLastError());
        // rows that are not concatenations of whole segments: the reference's iterator stops somewhere inside them; replayed on the host
        std::vector<std::vector<std::pair<uint64_t, bool>>> replayed(n);
        std::vector<uint32_t> row;
        for (uint64_t k = 0; k < n; k++) {
            if (valid[k]) continue;
            row.resize(offs[k + 1] - offs[k]);
            if (!row.empty()) HIP_CHECK(hipMemcpy(row.data(), paths.d_nodes + offs[k], row.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            uint64_t seq_len = 0;
            host_segment_path(h, row.data(), row.size(), replayed[k], seq_len);
            counts[k] = replayed[k].size();
        }
        out_offsets[0] = 0;
        for (uint64_t k = 0; k < n; k++) out_offsets[k + 1] = out_offsets[k] + counts[k];
        *total = out_offsets[n];
        if (!out_tokens || *total == 0) return GBWT_HIP_OK;               // the size query
        if (capacity < *total) return fail(GBWT_HIP_CAPACITY, "output capacity too small for the segment tokens");
        ws->gfa_text.reserve(std::max<uint64_t>(*total, 1) * sizeof(uint64_t));
        HIP_CHECK(hipMemcpyAsync(d_row_start, out_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_segment_fill_tokens, dim3(static_cast<unsigned>(chunks_cap)), dim3(FORMAT_THREADS), 0, s, paths.d_offsets, paths.d_nodes, n, d_chunk_first, d_chunk_path,
                           d_tokens_before, tables, d_valid, d_row_start, ws->gfa_text.as<uint64_t>());
        HIP_CHECK(hipMemcpyAsync(out_tokens, ws->gfa_text.ptr, *total * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        HIP_CHECK(hipGetLastError());
        for (uint64_t k = 0; k < n; k++)
            for (size_t j = 0; j < replayed[k].size(); j++) out_tokens[out_offsets[k] + j] = (replayed[k][j].first << 1) | (replayed[k][j].second ? 1u : 0u);
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_last_lines_ms(const gbwt_hip_workspace *ws, float *walk_ms, float *format_ms) {
    GBWT_HIP_GUARD_BEGIN
    if (!ws || !ws->lines_timed || !ws->timed) return fail(GBWT_HIP_BAD_ARGUMENT, "no timed GFA lines request on this workspace");
    float a = 0, b = 0;
    if (hipEventElapsedTime(&a, ws->ev[0], ws->ev[1]) != hipSuccess || hipEventElapsedTime(&b, ws->gev[0], ws->gev[1]) != hipSuccess)
        return fail(GBWT_HIP_DEVICE_ERROR, "hipEventElapsedTime failed");
    if (walk_ms) *walk_ms = a;
    if (format_ms) *format_ms = b;
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_write_gfa(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const char *path) {
    return gbwt_hip_write_gfa_mode(ix, ws, path, GBWT_HIP_PATHS_DEFAULT);
}

// The reference builds the lines of the paths in parallel and hands them to a writer under a mutex, behind an 8 MiB BufWriter
// (src/bin/gbunzip.rs:96, 205-226, 421-434).  Here: batches bounded by BYTES of text (GBWT_HIP_GFA_BATCH_MIB, default 1 GiB: config 4's
// 32 000 walks are four batches, the headline's 5 000 paths of 4.5 MB each a few hundred), formatted on the device into two text
// buffers in turn, while a writer thread moves the previous batch to the file through two 64 MiB pinned buffers (GfaWriter) -- and
// writes the H-, S- and L-lines of the graph, host work, while the first batch is walked.
gbwt_hip_status gbwt_hip_write_gfa_mode(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const char *path, int path_mode) {
    GBWT_HIP_GUARD_BEGIN
    if (!ix || !ws || ws->index != ix || !path) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace");
    if (path_mode < GBWT_HIP_PATHS_DEFAULT || path_mode > GBWT_HIP_PATHS_REF_ONLY) return fail(GBWT_HIP_BAD_ARGUMENT, "unknown path mode");
    try {
        require_gfa_capable(ix);
        const HostIndex &h = ix->host;
        const bool translated = h.has_translation && !h.segment_starts.empty();
        PositionalFile file;
        file.fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (file.fd < 0) return fail(GBWT_HIP_IO_ERROR, std::string("cannot create ") + path);
        GfaWriter writer;
        writer.file = &file; writer.host = &h; writer.translated = translated; writer.device = ix->device;
        writer.worker = std::thread([&writer]() { writer.run(); });
        // write_gfa_impl's match on the path mode (src/bin/gbunzip.rs:212-222), ascending path id (-t 1 order):
        //   default: paths of the generic sample as P-lines, then the others as W-lines (write_paths / write_walks, 343-417)
        //   pan-sn : every path as a P-line with its PanSN name (write_pan_sn, 371-393)
        //   ref-only: the P-lines of the default mode only
        uint64_t ref_sample = 0;
        const bool have_ref = (h.metadata_flags & 2) && h.sample_names.find(GENERIC_SAMPLE, ref_sample);
        if (!have_ref) ref_sample = h.sample_count;
        struct Pass { int line_mode; int which; };           // which: 0 = paths of the generic sample, 1 = the others, 2 = all
        std::vector<Pass> passes;
        if (path_mode == GBWT_HIP_PATHS_PAN_SN) passes.push_back(Pass{2, 2});
        else {
            if (have_ref) passes.push_back(Pass{0, 0});
            if (path_mode == GBWT_HIP_PATHS_DEFAULT) passes.push_back(Pass{1, 1});
        }
        // a path of `len` nodes prints at most len tokens of (digits of the largest node id, or the longest segment name) + 2 bytes
        uint64_t token = 2 + std::to_string(h.alphabet_size / 2).size();
        if (translated) { token = 2; for (size_t i = 0; i < h.segment_names.size(); i++) token = std::max<uint64_t>(token, 2 + h.segment_names.len(i)); }
        uint64_t budget = uint64_t(1) << 30;
        if (const char *v = std::getenv("GBWT_HIP_GFA_BATCH_MIB")) budget = std::max<uint64_t>(1, std::strtoull(v, nullptr, 10)) << 20;
        const uint64_t fallback_batch = 4096;                // without sequence lengths: by count, as until round 3
        int slot = 0;
        gbwt_hip_status st = GBWT_HIP_OK;
        for (const Pass &pass : passes) {
            std::vector<uint64_t> ids;
            for (uint64_t p = 0; p < h.path_names.size(); p++)
                if (pass.which == 2 || (h.path_names[p].sample == ref_sample) == (pass.which == 0)) ids.push_back(p);
            for (uint64_t b0 = 0; b0 < ids.size() && st == GBWT_HIP_OK;) {
                uint64_t nb = 0, bytes = 0;
                while (b0 + nb < ids.size()) {
                    const uint64_t seq = 2 * ids[b0 + nb];
                    const uint64_t est = seq < ix->host_seq_len.size() ? 128 + token * ix->host_seq_len[seq] : 0;
                    if (nb != 0 && (ix->host_seq_len.empty() ? nb >= fallback_batch : bytes + est > budget)) break;
                    bytes += est; nb++;
                }
                if (!writer.acquire(slot)) { st = GBWT_HIP_IO_ERROR; break; }
                st = path_lines_compute(ix, ws, ids.data() + b0, nb, pass.line_mode, slot);
                if (st != GBWT_HIP_OK) break;
                writer.submit((slot == 0 ? ws->gfa_text : ws->gfa_text2).as<char>(), ws->lines_total, slot);
                slot ^= 1;
                b0 += nb;
            }
            if (st != GBWT_HIP_OK) break;
        }
        const gbwt_hip_status wst = writer.finish();
        ws->lines_cached = false;                            // (both text buffers have been reused)
        if (st == GBWT_HIP_OK && wst != GBWT_HIP_OK) return fail(wst, writer.message);
        if (st != GBWT_HIP_OK) return wst != GBWT_HIP_OK ? fail(wst, writer.message) : st;
        if (file.failed) return fail(GBWT_HIP_IO_ERROR, "short write");
        return GBWT_HIP_OK;
    } catch (const InvalidData &e) {
        return fail(GBWT_HIP_BAD_ARGUMENT, e.what());
    } catch (const HipError &e) {
        return status_of(e);
    }
    GBWT_HIP_GUARD_END
}

}  // extern "C"
