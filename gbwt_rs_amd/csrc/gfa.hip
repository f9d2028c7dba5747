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
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "capi_internal.hpp"

using namespace gbwt_hip;

namespace {

constexpr int WAVE = 64;
constexpr int FORMAT_THREADS = 256;
const char *const GENERIC_SAMPLE = "_gbwt_ref";  // src/lib.rs

__device__ __forceinline__ uint32_t decimal_digits(uint32_t v) {
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) +
           (v >= 100000000u) + (v >= 1000000000u);
}

// Lines are formatted in chunks of LINE_CHUNK path positions, so that the work is as parallel for ninety haplotypes of two
// million nodes as it is for fifty thousand short walks (one workgroup per LINE had 0.6 G nodes/s on the former, 50 on the
// latter).  Chunk c belongs to the path with chunk_first[path] <= c < chunk_first[path + 1] (every path has at least one
// chunk: an empty path still has a header and a trailer).
constexpr uint32_t LINE_CHUNK = 4096;

__global__ void __launch_bounds__(256) k_chunk_counts(const uint64_t *offsets, uint64_t n, uint64_t *counts) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= n) return;
    const uint64_t len = offsets[p + 1] - offsets[p];
    counts[p] = len == 0 ? 1 : (len + LINE_CHUNK - 1) / LINE_CHUNK;
}

struct ChunkRange { uint64_t path, begin, lo, hi; bool first, last; };

__device__ __forceinline__ ChunkRange chunk_range(const uint64_t *chunk_first, uint64_t n, const uint64_t *offsets, uint64_t c) {
    uint64_t lo = 0, hi = n;                                        // chunk_first[lo] <= c < chunk_first[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) / 2;
        if (chunk_first[mid] <= c) lo = mid; else hi = mid;
    }
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
__global__ void __launch_bounds__(256) k_chunk_stats(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first, uint64_t chunks,
                                                      const uint32_t *label_len, uint64_t n_labels, uint32_t first_node, int p_lines, uint64_t *chunk_text,
                                                      uint64_t *chunk_seq) {
    const uint64_t c = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (c >= chunks) return;
    const ChunkRange r = chunk_range(chunk_first, n, offsets, c);
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

// Per path, from the scans over the chunks: text bytes of the tokens, summed label lengths, and -- translation graphs --
// whether every position fitted a whole segment (bad_before = scan of the chunks' bad flags, or null).
__global__ void __launch_bounds__(256) k_path_totals(const uint64_t *chunk_first, uint64_t n, const uint64_t *text_before, const uint64_t *seq_before,
                                                      const uint64_t *bad_before, uint64_t *text_len, uint64_t *seq_len, uint8_t *valid) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= n) return;
    const uint64_t a = chunk_first[p], b = chunk_first[p + 1];
    text_len[p] = text_before[b] - text_before[a];
    seq_len[p] = seq_before[b] - seq_before[a];
    if (valid) valid[p] = bad_before[b] == bad_before[a] ? 1 : 0;
}

// One workgroup per chunk: the header (first chunk of a line), the node tokens of the chunk, the trailer (last chunk).
// The tokens of 1 024 positions are put together in LDS (a block scan of their widths places them) and leave as aligned 16-byte
// stores; only the first and the last bytes of such a batch, where the text does not fill a 16-byte unit, go out one by one.
// (With every lane storing its own six bytes one at a time the formatter wrote 330 GB/s of text.)
constexpr uint32_t TOKEN_MAX = 12;   // ',' + ten digits + '+' (P-lines); '>' + ten digits (W-lines)
constexpr uint32_t PER_THREAD = 4;   // consecutive positions per thread and batch (one scan and two barriers per 1 024 positions)
__global__ void __launch_bounds__(FORMAT_THREADS) k_format_chunks(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first,
                                                                   const uint64_t *text_before, int p_lines, const uint64_t *line_start, const uint8_t *headers,
                                                                   const uint64_t *header_off, uint8_t *out) {
    using BlockScan = hipcub::BlockScan<uint32_t, FORMAT_THREADS>;
    __shared__ typename BlockScan::TempStorage scan_storage;
    __shared__ __attribute__((aligned(16))) uint8_t stage[FORMAT_THREADS * PER_THREAD * TOKEN_MAX + 32];
    const ChunkRange r = chunk_range(chunk_first, n, offsets, blockIdx.x);
    const uint32_t t = threadIdx.x;
    uint8_t *line = out + line_start[r.path];
    const uint64_t h0 = header_off[r.path], h1 = header_off[r.path + 1];
    if (r.first) for (uint64_t k = t; k < h1 - h0; k += FORMAT_THREADS) line[k] = headers[h0 + k];
    uint64_t cursor = (h1 - h0) + (text_before[blockIdx.x] - text_before[chunk_first[r.path]]);
    for (uint64_t base = r.lo; base < r.hi; base += FORMAT_THREADS * PER_THREAD) {
        const uint64_t k0 = base + PER_THREAD * t;
        uint32_t node[PER_THREAD], digits[PER_THREAD], len = 0;
#pragma unroll
        for (uint32_t i = 0; i < PER_THREAD; i++) {
            node[i] = 0; digits[i] = 0;
            if (k0 + i < r.hi) {
                node[i] = nodes[k0 + i];
                digits[i] = decimal_digits(node[i] >> 1);
                // ',' between the tokens of a P-line and '+' / '-' behind each, '>' / '<' in front of a W-line's
                len += digits[i] + (p_lines ? (k0 + i > r.begin ? 2u : 1u) : 1u);
            }
        }
        uint32_t pos, total;
        BlockScan(scan_storage).ExclusiveSum(len, pos, total);
        uint8_t *const to = line + cursor;
        const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(to) & 15u);   // the batch's text lies at stage[mis ...]: LDS and memory are aligned alike
        uint8_t *w = stage + mis + pos;
#pragma unroll
        for (uint32_t i = 0; i < PER_THREAD; i++) {
            if (digits[i] == 0) continue;
            if (!p_lines) *w++ = (node[i] & 1u) ? '<' : '>';
            else if (k0 + i > r.begin) *w++ = ',';
            uint32_t v = node[i] >> 1;
            for (uint32_t d = 0; d < digits[i]; d++) { w[digits[i] - 1 - d] = static_cast<uint8_t>('0' + v % 10u); v /= 10u; }
            w += digits[i];
            if (p_lines) *w++ = (node[i] & 1u) ? '-' : '+';
        }
        __syncthreads();
        const uint32_t end = mis + total;
        uint8_t *const aligned = to - mis;
        for (uint32_t lo = 16 * t; lo < end; lo += 16 * FORMAT_THREADS) {
            if (lo >= mis && lo + 16 <= end) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(stage + lo), reinterpret_cast<u32x4 *>(aligned + lo));
            } else {
                const uint32_t from = lo < mis ? mis : lo, upto = lo + 16 < end ? lo + 16 : end;
                for (uint32_t q = from; q < upto; q++) aligned[q] = stage[q];
            }
        }
        cursor += total;
        __syncthreads();
    }
    // trailer: "\t*\n" for P-lines (src/bin/gbunzip.rs:476), "\n" for W-lines (:548)
    if (r.last && t == 0) {
        if (p_lines) { line[cursor] = '\t'; line[cursor + 1] = '*'; line[cursor + 2] = '\n'; }
        else line[cursor] = '\n';
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
__global__ void __launch_bounds__(256) k_chunk_stats_segments(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first,
                                                               uint64_t chunks, SegmentTables t, int p_lines, uint64_t *chunk_text, uint64_t *chunk_seq,
                                                               uint64_t *chunk_bad) {
    const uint64_t c = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (c >= chunks) return;
    const ChunkRange r = chunk_range(chunk_first, n, offsets, c);
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
__global__ void __launch_bounds__(FORMAT_THREADS) k_format_chunks_segments(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, const uint64_t *chunk_first,
                                                                            const uint64_t *text_before, SegmentTables t, int p_lines, const uint8_t *valid,
                                                                            const uint64_t *line_start, const uint8_t *headers, const uint64_t *header_off,
                                                                            uint8_t *out) {
    using BlockScan = hipcub::BlockScan<uint32_t, FORMAT_THREADS>;
    __shared__ typename BlockScan::TempStorage scan_storage;
    const ChunkRange r = chunk_range(chunk_first, n, offsets, blockIdx.x);
    if (!valid[r.path]) return;
    const uint32_t tid = threadIdx.x;
    uint8_t *line = out + line_start[r.path];
    const uint64_t h0 = header_off[r.path], h1 = header_off[r.path + 1];
    if (r.first) for (uint64_t k = tid; k < h1 - h0; k += FORMAT_THREADS) line[k] = headers[h0 + k];
    uint64_t cursor = (h1 - h0) + (text_before[blockIdx.x] - text_before[chunk_first[r.path]]);
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

std::string name_or_id(const Strings &names, bool has_names, uint64_t id) {
    // Metadata::sample_name / contig_name fall back to the number (src/gbwt.rs:744-761, 792-809)
    if (has_names && id < names.size()) return names.str(id);
    return std::to_string(id);
}

void require_gfa_capable(const gbwt_hip_index *ix) {
    if (!ix->host.is_gbz) throw InvalidData("GFA lines need a GBZ (graph + metadata), this handle holds a bare GBWT");
    if (!ix->host.has_metadata) throw InvalidData("GFA lines need path metadata");
}

// Edge list of a record, decoded on the host (Record::decompress_edges, src/bwt.rs:378-395) for the L-lines.
bool host_edges(const HostIndex &h, uint64_t rec, std::vector<std::pair<uint64_t, uint64_t>> &edges) {
    edges.clear();
    if (rec >= h.records()) return false;
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

// Uploads the label length of every potential node (0 where GBZ::has_node is false, src/gbz.rs:286-289).
void upload_label_lengths(gbwt_hip_index &ix) {
    const HostIndex &h = ix.host;
    if (!h.is_gbz) return;
    const uint64_t first = h.alphabet_offset + 1;
    std::vector<uint32_t> len(h.sequences_labels.size() + 1, 0);
    {   // (sixteen million nodes in a config-4-shaped GBZ: a few threads)
        const uint64_t n = h.sequences_labels.size();
        const unsigned pieces = n >= (uint64_t(1) << 23) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u;
        auto piece = [&](unsigned p) {
            for (uint64_t s = n * p / pieces, end = n * (p + 1) / pieces; s < end; s++) {
                const uint64_t node = 2 * s + first;              // forward GBWT node of sequence s
                const uint64_t rec = node - h.alphabet_offset;
                const bool real = rec < h.records() && h.starts[rec + 1] > h.starts[rec] && h.data[h.starts[rec]] != 0;   // BWT::id_iter
                len[s] = real ? static_cast<uint32_t>(h.sequences_labels.len(s)) : 0u;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned p = 1; p < pieces; p++) pool.emplace_back(piece, p);
        piece(0);
        for (auto &t : pool) t.join();
    }
    ix.label_len.reserve(len.size() * sizeof(uint32_t));
    HIP_CHECK(hipMemcpy(ix.label_len.ptr, len.data(), len.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
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

// The lines of a batch of paths, formatted ONCE into device memory (ws->gfa_text, line k at [line_start[k], line_start[k + 1]),
// offsets also on the device in ws->gfa_b).  The request is remembered in the workspace: the fill call that follows a size
// query, and the copy-out of gbwt_hip_path_lines after gbwt_hip_path_lines_device, find the text there.
static gbwt_hip_status path_lines_compute(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *path_ids, uint64_t n, int mode) {
    if (!ix || !ws || ws->index != ix) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace");
    if (n && !path_ids) return fail(GBWT_HIP_BAD_ARGUMENT, "null path_ids");
    if (mode < 0 || mode > 2) return fail(GBWT_HIP_BAD_ARGUMENT, "mode must be 0 (P-lines), 1 (W-lines) or 2 (P-lines with PanSN names)");
    const int p_lines = mode != 1 ? 1 : 0;
    if (ws->lines_cached && ws->lines_mode == mode && ws->lines_key.size() == n && (n == 0 || std::memcmp(ws->lines_key.data(), path_ids, n * sizeof(uint64_t)) == 0))
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
            ws->lines_key.clear(); ws->lines_mode = mode; ws->lines_cached = true;
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
        // 2. size of every line: chunks of LINE_CHUNK positions, a wave per chunk, scans over the chunks
        const size_t tb_n = scan_temp_bytes(n);
        ws->gfa_a.reserve(2 * n * sizeof(uint64_t));
        ws->gfa_chunk_first.reserve(2 * (n + 1) * sizeof(uint64_t));
        ws->scan_temp.reserve(std::max<size_t>(tb_n, 16));
        uint64_t *d_text_len = ws->gfa_a.as<uint64_t>(), *d_seq_len = d_text_len + n;
        uint64_t *d_chunk_first = ws->gfa_chunk_first.as<uint64_t>(), *d_chunk_counts = d_chunk_first + (n + 1);
        hipLaunchKernelGGL(k_chunk_counts, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, paths.d_offsets, n, d_chunk_counts);
        launch_scan(d_chunk_counts, d_chunk_first, n, ws->scan_temp.ptr, tb_n, s);
        uint64_t chunks = 0;
        HIP_CHECK(hipMemcpyAsync(&chunks, d_chunk_first + n, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (chunks > 0x7FFFFFFFull) return fail(GBWT_HIP_UNSUPPORTED, "too many line chunks in one batch: format fewer paths per call");
        const size_t tb_c = scan_temp_bytes(chunks);
        ws->gfa_chunks.reserve((3 * chunks + 3 * (chunks + 1)) * sizeof(uint64_t));
        ws->scan_temp.reserve(std::max<size_t>(tb_c, 16));
        uint64_t *d_chunk_text = ws->gfa_chunks.as<uint64_t>(), *d_chunk_seq = d_chunk_text + chunks, *d_chunk_bad = d_chunk_seq + chunks;
        uint64_t *d_text_before = d_chunk_bad + chunks, *d_seq_before = d_text_before + (chunks + 1), *d_bad_before = d_seq_before + (chunks + 1);
        std::vector<uint8_t> valid(n, 1);
        const unsigned stat_blocks = static_cast<unsigned>((chunks + 3) / 4);
        if (translated) {
            ws->gfa_valid.reserve(std::max<uint64_t>(n, 16));
            hipLaunchKernelGGL(k_chunk_stats_segments, dim3(stat_blocks), dim3(256), 0, s, paths.d_offsets, paths.d_nodes, n, d_chunk_first, chunks,
                               segment_tables(ix), p_lines, d_chunk_text, d_chunk_seq, d_chunk_bad);
            launch_scan(d_chunk_bad, d_bad_before, chunks, ws->scan_temp.ptr, tb_c, s);
        } else {
            hipLaunchKernelGGL(k_chunk_stats, dim3(stat_blocks), dim3(256), 0, s, paths.d_offsets, paths.d_nodes, n, d_chunk_first, chunks,
                               ix->label_len.as<uint32_t>(), static_cast<uint64_t>(h.sequences_labels.size()),
                               static_cast<uint32_t>(h.alphabet_offset + 1), p_lines, d_chunk_text, d_chunk_seq);
        }
        launch_scan(d_chunk_text, d_text_before, chunks, ws->scan_temp.ptr, tb_c, s);
        launch_scan(d_chunk_seq, d_seq_before, chunks, ws->scan_temp.ptr, tb_c, s);
        hipLaunchKernelGGL(k_path_totals, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, d_chunk_first, n, d_text_before, d_seq_before,
                           translated ? d_bad_before : nullptr, d_text_len, d_seq_len, translated ? ws->gfa_valid.as<uint8_t>() : nullptr);
        if (translated) HIP_CHECK(hipMemcpyAsync(valid.data(), ws->gfa_valid.ptr, n, hipMemcpyDeviceToHost, s));
        std::vector<uint64_t> lens(2 * n);
        HIP_CHECK(hipMemcpyAsync(lens.data(), d_text_len, 2 * n * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        HIP_CHECK(hipGetLastError());
        const bool sample_names = (h.metadata_flags & 2) != 0, contig_names = (h.metadata_flags & 4) != 0;
        auto line_header = [&](uint64_t k, uint64_t seq_len) {
            const PathName &pn = h.path_names[path_ids[k]];
            if (mode == 0) return "P\t" + name_or_id(h.contig_names, contig_names, pn.contig) + "\t";
            if (mode == 2) return "P\t" + name_or_id(h.sample_names, sample_names, pn.sample) + "#" + std::to_string(pn.phase) + "#" +
                                  name_or_id(h.contig_names, contig_names, pn.contig) + "\t";   // Metadata::pan_sn_path, src/gbwt.rs:709-713
            return "W\t" + name_or_id(h.sample_names, sample_names, pn.sample) + "\t" + std::to_string(pn.phase) + "\t" +
                   name_or_id(h.contig_names, contig_names, pn.contig) + "\t" + std::to_string(pn.fragment) + "\t" +
                   std::to_string(static_cast<uint64_t>(pn.fragment) + seq_len) + "\t";
        };
        // 2b. paths that are not concatenations of whole segments: the reference's iterator stops somewhere inside them;
        // the host replays it on the extracted node ids
        std::vector<std::string> host_lines(translated ? n : 0);
        if (translated && std::find(valid.begin(), valid.end(), uint8_t(0)) != valid.end()) {
            std::vector<uint64_t> offs(n + 1);
            HIP_CHECK(hipMemcpy(offs.data(), paths.d_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
            std::vector<uint32_t> nodes;
            std::vector<std::pair<uint64_t, bool>> tokens;
            for (uint64_t k = 0; k < n; k++) {
                if (valid[k]) continue;
                nodes.resize(offs[k + 1] - offs[k]);
                if (!nodes.empty()) HIP_CHECK(hipMemcpy(nodes.data(), paths.d_nodes + offs[k], nodes.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                uint64_t seq_len = 0;
                host_segment_path(h, nodes.data(), nodes.size(), tokens, seq_len);
                std::string &line = host_lines[k];
                line = line_header(k, seq_len);
                for (size_t j = 0; j < tokens.size(); j++) {
                    const std::string name = h.segment_names.str(tokens[j].first);
                    if (p_lines) line += (j ? "," : "") + name + (tokens[j].second ? "-" : "+");
                    else line += (tokens[j].second ? "<" : ">") + name;
                }
                line += p_lines ? "\t*\n" : "\n";
            }
        }
        // 3. headers (path_to_p_line / path_to_pan_sn / path_to_w_line, src/bin/gbunzip.rs:480-524)
        // (appended in place: five thousand headers put together from temporaries were a sixth of a call that formats a gigabyte)
        std::string headers;
        headers.reserve(64 * n);
        auto append_number = [&](uint64_t v) {
            char digits[24];
            int len = 0;
            do { digits[len++] = static_cast<char>('0' + v % 10); v /= 10; } while (v != 0);
            while (len > 0) headers.push_back(digits[--len]);
        };
        auto append_name = [&](const Strings &names, bool has_names, uint64_t id) {
            if (has_names && id < names.size()) headers.append(reinterpret_cast<const char *>(names.bytes.data()) + names.offsets[id], names.offsets[id + 1] - names.offsets[id]);
            else append_number(id);                          // Metadata::sample_name / contig_name fall back to the number
        };
        auto append_header = [&](uint64_t k, uint64_t seq_len) {
            const PathName &pn = h.path_names[path_ids[k]];
            if (mode == 0) { headers += "P\t"; append_name(h.contig_names, contig_names, pn.contig); headers.push_back('\t'); return; }
            if (mode == 2) {                                 // sample#phase#contig (Metadata::pan_sn_path, src/gbwt.rs:709-713)
                headers += "P\t"; append_name(h.sample_names, sample_names, pn.sample); headers.push_back('#');
                append_number(pn.phase); headers.push_back('#');
                append_name(h.contig_names, contig_names, pn.contig); headers.push_back('\t');
                return;
            }
            headers += "W\t"; append_name(h.sample_names, sample_names, pn.sample); headers.push_back('\t');
            append_number(pn.phase); headers.push_back('\t');
            append_name(h.contig_names, contig_names, pn.contig); headers.push_back('\t');
            append_number(pn.fragment); headers.push_back('\t');
            append_number(static_cast<uint64_t>(pn.fragment) + seq_len); headers.push_back('\t');
        };
        std::vector<uint64_t> header_off(n + 1, 0), line_start(n + 1, 0);
        for (uint64_t k = 0; k < n; k++) {
            if (valid[k]) append_header(k, lens[n + k]);
            header_off[k + 1] = headers.size();
            line_start[k + 1] = line_start[k] + (valid[k] ? (header_off[k + 1] - header_off[k]) + lens[k] + (p_lines ? 3 : 1) : host_lines[k].size());
        }
        const uint64_t total = line_start[n];
        // 4. format on the device; the lines the host had to replay are copied into their places
        ws->gfa_b.reserve((2 * (n + 1)) * sizeof(uint64_t));
        ws->gfa_c.reserve(std::max<size_t>(headers.size(), 16));
        ws->gfa_text.reserve(std::max<uint64_t>(total, 16));
        uint64_t *d_line_start = ws->gfa_b.as<uint64_t>(), *d_header_off = d_line_start + (n + 1);
        HIP_CHECK(hipMemcpyAsync(d_line_start, line_start.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_header_off, header_off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(ws->gfa_c.ptr, headers.data(), headers.size(), hipMemcpyHostToDevice, s));
        if (translated)
            hipLaunchKernelGGL(k_format_chunks_segments, dim3(static_cast<unsigned>(chunks)), dim3(FORMAT_THREADS), 0, s, paths.d_offsets, paths.d_nodes, n,
                               d_chunk_first, d_text_before, segment_tables(ix), p_lines, ws->gfa_valid.as<uint8_t>(), d_line_start,
                               ws->gfa_c.as<uint8_t>(), d_header_off, ws->gfa_text.as<uint8_t>());
        else
            hipLaunchKernelGGL(k_format_chunks, dim3(static_cast<unsigned>(chunks)), dim3(FORMAT_THREADS), 0, s, paths.d_offsets, paths.d_nodes, n,
                               d_chunk_first, d_text_before, p_lines, d_line_start, ws->gfa_c.as<uint8_t>(), d_header_off,
                               ws->gfa_text.as<uint8_t>());
        HIP_CHECK(hipGetLastError());
        for (uint64_t k = 0; k < n; k++)
            if (!valid[k] && !host_lines[k].empty())
                HIP_CHECK(hipMemcpyAsync(ws->gfa_text.as<char>() + line_start[k], host_lines[k].data(), host_lines[k].size(), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));   // host_lines / headers / line_start go out of scope
        ws->lines_total = total;
        ws->lines_key.assign(path_ids, path_ids + n);
        ws->lines_mode = mode;
        ws->lines_cached = true;
        return GBWT_HIP_OK;
    } catch (const InvalidData &e) {
        return fail(GBWT_HIP_BAD_ARGUMENT, e.what());
    } catch (const HipError &e) {
        return status_of(e);
    }
}

// gbwt_hip_path_lines; with `grow` the text goes into that vector (sized here) instead of `out`.
static gbwt_hip_status path_lines_impl(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *path_ids, uint64_t n, int mode,
                                       char *out, uint64_t capacity, uint64_t *total, std::vector<char> *grow) {
    if (!total) return fail(GBWT_HIP_BAD_ARGUMENT, "null total");
    *total = 0;
    const gbwt_hip_status st = path_lines_compute(ix, ws, path_ids, n, mode);
    if (st != GBWT_HIP_OK) return st;
    *total = ws->lines_total;
    if (grow) { grow->resize(*total); out = grow->data(); capacity = *total; }
    if (!out || *total == 0) return GBWT_HIP_OK;
    if (capacity < *total) return fail(GBWT_HIP_CAPACITY, "output capacity too small for the lines");
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        HIP_CHECK(hipMemcpyAsync(out, ws->gfa_text.ptr, *total, hipMemcpyDeviceToHost, ws->stream));
        HIP_CHECK(hipStreamSynchronize(ws->stream));
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
}

extern "C" {

gbwt_hip_status gbwt_hip_path_lines(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *path_ids, uint64_t n, int mode,
                                    char *out, uint64_t capacity, uint64_t *total) {
    GBWT_HIP_GUARD_BEGIN
    return path_lines_impl(ix, ws, path_ids, n, mode, out, capacity, total, nullptr);
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

gbwt_hip_status gbwt_hip_write_gfa(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const char *path) {
    return gbwt_hip_write_gfa_mode(ix, ws, path, GBWT_HIP_PATHS_DEFAULT);
}

gbwt_hip_status gbwt_hip_write_gfa_mode(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const char *path, int path_mode) {
    GBWT_HIP_GUARD_BEGIN
    if (!ix || !ws || ws->index != ix || !path) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace");
    if (path_mode < GBWT_HIP_PATHS_DEFAULT || path_mode > GBWT_HIP_PATHS_REF_ONLY) return fail(GBWT_HIP_BAD_ARGUMENT, "unknown path mode");
    try {
        require_gfa_capable(ix);
        const HostIndex &h = ix->host;
        const bool translated = h.has_translation && !h.segment_starts.empty();
        std::unique_ptr<FILE, int (*)(FILE *)> f(std::fopen(path, "wb"), std::fclose);
        if (!f) return fail(GBWT_HIP_IO_ERROR, std::string("cannot create ") + path);
        std::string text;
        // header (write_gfa_header, src/bin/gbunzip.rs:193-203)
        if (const std::string *rs = h.tag("reference_samples")) text = "H\tVN:Z:1.1\tRS:Z:" + *rs + "\n";
        else text = "H\tVN:Z:1.1\n";
        // segments + links over the real nodes (write_segments / write_links, src/bin/gbunzip.rs:230-317)
        const uint64_t first = h.alphabet_offset + 1, potential = h.sequences_labels.size();
        auto real = [&](uint64_t seq) {
            const uint64_t rec = 2 * seq + first - h.alphabet_offset;
            return rec < h.records() && h.starts[rec + 1] > h.starts[rec] && h.data[h.starts[rec]] != 0;
        };
        std::vector<std::pair<uint64_t, uint64_t>> edges;
        if (translated) {
            // GBZ::segment_iter keeps the segments whose first node exists (src/gbz.rs:927-929); links go from the last
            // node of the segment in the orientation of travel (segment_successors, src/gbz.rs:402-415) and are named by
            // the segment of the successor; LinkIter ends at a successor without a segment (src/gbz.rs:996-999)
            const uint64_t n_seg = h.segment_starts.size();
            for (uint64_t id = 0; id < n_seg; id++) {
                const HostSegment seg = host_segment(h, id);
                if (!host_has_node(h, seg.start)) continue;
                text += "S\t" + h.segment_names.str(id) + "\t";
                text.append(reinterpret_cast<const char *>(h.sequences_labels.bytes.data()) + h.sequences_labels.offsets[seg.start - 1], host_segment_seq_len(h, seg));
                text += "\n";
                if (text.size() > (8u << 20)) { std::fwrite(text.data(), 1, text.size(), f.get()); text.clear(); }
            }
            for (uint64_t id = 0; id < n_seg; id++) {
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
                if (text.size() > (8u << 20)) { std::fwrite(text.data(), 1, text.size(), f.get()); text.clear(); }
            }
        }
        for (uint64_t seq = 0; seq < potential && !translated; seq++) {
            if (!real(seq)) continue;
            const uint64_t node_id = (2 * seq + first) / 2;
            text += "S\t" + std::to_string(node_id) + "\t" + h.sequences_labels.str(seq) + "\n";
            if (text.size() > (8u << 20)) { std::fwrite(text.data(), 1, text.size(), f.get()); text.clear(); }
        }
        for (uint64_t seq = 0; seq < potential && !translated; seq++) {
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
                    text += "L\t" + std::to_string(node_id) + (rev ? "\t-\t" : "\t+\t") + std::to_string(succ) + (succ_rev ? "\t-\t*\n" : "\t+\t*\n");
                }
            }
            if (text.size() > (8u << 20)) { std::fwrite(text.data(), 1, text.size(), f.get()); text.clear(); }
        }
        std::fwrite(text.data(), 1, text.size(), f.get());
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
        for (const Pass &pass : passes) {
            std::vector<uint64_t> ids;
            for (uint64_t p = 0; p < h.path_names.size(); p++)
                if (pass.which == 2 || (h.path_names[p].sample == ref_sample) == (pass.which == 0)) ids.push_back(p);
            const uint64_t batch = 4096;
            std::vector<char> buf;
            for (uint64_t b0 = 0; b0 < ids.size(); b0 += batch) {
                const uint64_t nb = std::min<uint64_t>(batch, ids.size() - b0);
                uint64_t total = 0;
                gbwt_hip_status st = path_lines_impl(ix, ws, ids.data() + b0, nb, pass.line_mode, nullptr, 0, &total, &buf);
                if (st != GBWT_HIP_OK) return st;
                if (std::fwrite(buf.data(), 1, total, f.get()) != total) return fail(GBWT_HIP_IO_ERROR, "short write");
            }
        }
        return GBWT_HIP_OK;
    } catch (const InvalidData &e) {
        return fail(GBWT_HIP_BAD_ARGUMENT, e.what());
    } catch (const HipError &e) {
        return status_of(e);
    }
    GBWT_HIP_GUARD_END
}

}  // extern "C"
