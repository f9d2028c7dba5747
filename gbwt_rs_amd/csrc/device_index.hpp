// device_index.hpp -- the index as the kernels see it (plain pointers into HBM).
//
// HBM layout (DESIGN.md "Data layout"):
//   data      : the record byte stream of bwt::BWT (src/bwt.rs:97-100), verbatim, followed by
//               DATA_PAD zero bytes so that unaligned 8-byte window loads never leave the buffer
//   starts    : dense record starts, n_records + 1 entries (sentinel = data_len); u32 when the stream
//               is < 4 GiB, else u64.  Replaces the Elias-Fano select of BWT::record_bytes
//               (src/bwt.rs:116-121); used by the search kernels
//   endmarker : record 0 fully decompressed at open (src/gbwt.rs:413-414), one (node, offset) per sequence
//
// Built on the device at open and read by the walk kernels:
//   desc_raw  : one 64-byte descriptor per record, four uint4 (desc_raw[4 * rec + 0..3]):
//                 A = {successor 0, offset 0, successor 1, offset 1}         decoded edge list (class 1 / 2 only)
//                 B = {start (low 32 bits), length in bytes, meta, Record::len}
//                     meta = body offset (bits 0-15) | class (bits 16-17) | start (bits 32-39 of it, in bits 24-31)
//                     class 1 / 2 = outdegree 1 / 2 (class 1 is always unary, class 2 has rank blocks) with the edges in A and "body offset" = where the run stream
//                     starts inside the record; class 0 = any other non-empty record (generic lane-serial decode)
//                 C = {value-0 positions of the record, Record::len, first LF table entry (class 0) / first rank block (class 2), 1 if the record has an LF table}
//                 D = the first 16 bytes of the run stream, so short records need no second load
//               empty / None record : B.y = 0
//               unary record        : B.y = DESC_UNARY.  "Unary" = outdegree 1 (every node on a linear stretch of
//                 the graph): Record::lf(i) = (A.x, A.y + i) for i < B.w, however the body splits its runs.
//   desc      : the walk descriptor derived from desc_raw (k_link_desc, k_link_lookahead), fetched by the default
//               walk with three aligned dwordx4 loads that travel together (desc[4 * rec + 0..2]; slot 3 is unused):
//                 [0] = edge 0, [1] = edge 1 = {node to emit (0 = ENDMARKER: nothing), offset base, landing record
//                       index (0 when the walk ends behind the edge), block base of the landing record}
//                 [2] = {look-ahead base 0 | DESC_SLOW, flags 0 | look-ahead count 0, look-ahead base 1,
//                        flags 1 | look-ahead count 1};  flags: EDGE_CONT = the walk continues behind the edge,
//                       EDGE_EMIT2 = the edge is FUSED with a unary successor: taking it emits that successor AND the
//                       node of the landing record, one iteration -- one round trip to memory -- for two nodes.
//               Every offset an edge can produce was checked against the length of the landing record when the
//               descriptors were linked, so the walk tests nothing per step; records that could not be vouched for
//               (class 0, edges failing the check) carry DESC_SLOW and take the generic decoder.  Record 0 (the
//               endmarker, never landed on) holds the PARKING descriptor: nothing to emit, lands on record 0.
//               Look-ahead: per edge, the rank blocks {first, count} of the record a walk reaches a few iterations
//               after taking the edge; the walk touches one of them per iteration to warm the L2 of its XCD.
//   block_base: per record, index of its first rank block or BLOCK_NONE (needed where a walk starts; afterwards
//               the block base of the next record rides along in the edge taken)
//   tables    : the class 0 records (outdegree > 2, streams outside the descriptor's limits) decompressed once at open
//               (Record::decompress, src/bwt.rs:466-478) into one 16-byte entry per position: {successor node,
//               offset in the successor, landing record index or 0 where the walk ends, block base of the landing
//               record} -- Record::lf plus the arrival tests of GBWT::forward as one lookup.  Built only while they fit
//               the budget (GBWT_HIP_TABLE_BYTES, default 16 GiB); otherwise such records are decoded serially.
//   seq_len   : length (nodes) of every sequence, counted by one walk of all sequences at open.  With the lengths known
//               an extraction knows its CSR offsets before it starts: lanes write straight into their rows (no pool of
//               chained blocks, no compaction pass), and in a bidirectional index every sequence is walked from BOTH
//               ends at once -- sequence id from its start for the first half, sequence id ^ 1 (the same path
//               reversed, src/support.rs:310-314) from ITS start for the second half, written back to front with the
//               nodes flipped.  The LF steps are the same ones; the dependent chain per sequence is half as long.
//   samples   : the position of every sequence about every `sample_interval` nodes (recorded by a second walk at open):
//               {record, offset, block base, nodes emitted so far}.  An extraction starts one walker per sample, so a
//               row is filled by many walkers at once and the dependent chain is one sample interval long instead of
//               the whole sequence -- the walk stops being bound by the latency of one chain and becomes a
//               throughput problem.  Every LF step is still taken, by exactly one walker.
//   blocks    : the outdegree-2 records decoded once at open (k_fill_blocks) into RANK BLOCKS of 64 offsets, 16 bytes
//               each: {values of offsets 64k .. 64k+63 as one bit each (two words), value-1 offsets before 64k, 0}.
//               Record::lf (src/bwt.rs:480-496) at offset i becomes: value = bit i, rank = ones-before (value 1) or
//               i - ones-before (value 0) -- one popcount instead of the reference's scan over the runs from the
//               start of the record.  A lane fetches the descriptor and the block of its offset in ONE round trip.
//               Cost: 2.5 bits per BWT position of an outdegree-2 record, whatever its run structure (sized for
//               HBM, not for disk).  Block 0 is all zero and shared: it is what unary records read (value 0, rank = i).
//               Like simple-sds's rank/select supports, descriptors and blocks are rebuilt at load, never stored.
#pragma once

#include <cstdint>

namespace gbwt_hip {

constexpr uint32_t DESC_UNARY = 0xFFFFFFFFu;
constexpr uint32_t EDGE_EMIT2 = 1u << 31;            // walk descriptor, per edge: fused with a unary successor
constexpr uint32_t EDGE_CONT = 1u << 30;             // walk descriptor, per edge: the walk continues behind the edge
constexpr uint32_t DESC_SLOW = 1u << 31;             // walk descriptor, D.x: generic decode (class 0, unchecked edges)
constexpr uint32_t LOOKAHEAD_COUNT_MASK = (1u << 29) - 1;
constexpr uint32_t REC_MASK = (1u << 30) - 1;       // record indices are < 2^30; bits 30-31 of a record word carry flags
constexpr uint32_t LEAF_EMIT2 = 1u << 31;            // two-step descriptor: this step is fused with a unary successor
constexpr uint32_t DESC2_SLOW = 1u << 30;            // two-step descriptor, word E_a.z: generic decode
constexpr uint32_t GATHER_OK = 1u;                   // two-step descriptor, word E_a.w: the record's packed blocks (gblocks) can count it
constexpr uint32_t E_CHAIN = 2u;                     // two-step descriptor, word E_a.w: the first step runs through a chain of unary records (below)
constexpr uint32_t E_ALL4 = 8u;                      // two-step descriptor, word E_0.w: whichever edge and leaf a lane takes here, the iteration emits four nodes (no ENDMARKER, both steps fused, nothing chained)
constexpr uint32_t E_ANYCHAIN = 4u;                  // two-step descriptor, word E_0.w: some step of the record (E_0, E_1 or a leaf) is chained
constexpr uint32_t LEAF_CHAIN = 1u << 30;            // two-step descriptor, leaf word z: the second step does
constexpr uint32_t CHAIN_MAX = 6;                    // at most this many nodes between the first node of a step and its landing node
constexpr uint32_t WT_TABLE = 1u << 30;              // walk table entry: the landing record is a table record, word 3 = its table base
constexpr uint32_t WT_DEEP_STEPS = 7;                // deep walk table entry: this many table steps in 64 bytes
constexpr uint32_t WT_COMPACT_STEPS = 12;            // ... or this many, where all of them emit two nodes whose ids lie within 16-bit deltas of each other (k_fill_wtables_deep)
constexpr uint32_t WT_COMPACT = 1u << 30;            // deep walk table entry, word 1: the compact form
constexpr uint32_t WT_COMPACT_TABLE = 1u << 29;      // compact form, word 1: the last landing record is a table record (word 15 = its table base)
constexpr uint32_t BLOCK_NONE = 0xFFFFFFFFu;
constexpr uint32_t RANK_BLOCK_SHIFT = 6;   // 64 offsets per rank block
constexpr uint32_t DATA_PAD = 128;  // lane 63 of a cooperative chunk reads up to 71 bytes past the chunk start

struct DeviceIndex {
    const uint8_t *data;
    const uint32_t *starts32;  // exactly one of starts32 / starts64 is non-null
    const uint64_t *starts64;
    const uint2 *endmarker;    // .x = node, .y = offset
    const uint4 *desc;         // 4 * n_records entries (walk descriptors)
    const uint4 *desc_raw;     // 4 * n_records entries (raw descriptors)
    const uint32_t *block_base; // n_records entries
    const uint4 *blocks;       // n_blocks entries
    const uint4 *desc2;        // 8 * n_records entries (two-step walk descriptors)
    const uint4 *tables;       // LF tables of the class 0 records (desc_raw C.z = first entry, C.w = 1), or null
    const uint4 *wtables;      // walk tables, same indexing: {node to emit, offset, landing record | LEAF_EMIT2 | WT_TABLE, its block / table base}, or null
    const uint4 *wtables_deep; // deep walk tables: four uint4 per position = WT_DEEP_STEPS walk-table steps (load_kernels.hip: k_fill_wtables_deep), or null
    const uint32_t *seq_len;   // number of nodes of every sequence (counted once at open), or null
    const uint4 *samples;      // sequence samples {record, offset, block base, nodes emitted so far}, or null
    const uint64_t *sample_base;   // n_sequences + 1: first sample of every sequence
    uint32_t sample_interval;  // a sample about every this many nodes
    uint32_t sample_stride;    // an extraction starts a walker at every sample_stride-th sample of a sequence (0 = 1 = at every one; set per request: gbwt_hip_extract_device)
    uint32_t sample_part, sample_parts;   // sample_parts > 1: the extraction fills part sample_part of sample_parts of every row only (gbwt_hip_extract_part_device; row_segments below)
    const uint4 *cblocks;      // 2 * n_blocks entries (two-step rank blocks, same indexing as blocks)
    const uint4 *gblocks;      // 2 * n_blocks entries: the same, one 16-byte entry per 32 offsets with packed counts (gather loop)
    uint64_t data_len;
    uint64_t n_records;
    uint64_t n_sequences;      // header.sequences
    uint64_t n_endmarker;      // decompressed endmarker length
    uint64_t n_blocks;
    uint64_t max_walk;         // BWT positions in all records together: no sequence of a consistent index visits more (bounds the walks at open)
    uint32_t alphabet_offset;
    uint32_t first_node;       // alphabet_offset + 1
    uint32_t chained;          // 0: no two-step descriptor has a chained step (E_CHAIN / LEAF_CHAIN); else the most nodes an iteration of the walk can stage (5 .. 16; 4 without chains)
};

// The segments of the row of sequence `id` in one extraction: the samples base .. of the sequence, `count` segments at the request's stride
// (segment j = from sample j * stride to sample (j + 1) * stride or the end), of which this extraction fills lo .. hi - 1: all of them, or
// the sample_part-th of sample_parts equal shares (a rank of a multi-GPU extraction: every rank walks every path over ITS stretch of it).
struct RowSegments { uint64_t base, stride, count, lo, hi; };
__device__ inline RowSegments row_segments(const DeviceIndex &ix, uint64_t id) {
    RowSegments r;
    r.stride = ix.sample_stride ? ix.sample_stride : 1u;
    r.base = ix.sample_base[id];
    r.count = (ix.sample_base[id + 1] - r.base + r.stride - 1) / r.stride;
    r.lo = 0; r.hi = r.count;
    if (ix.sample_parts > 1) { r.lo = r.count * ix.sample_part / ix.sample_parts; r.hi = r.count * (ix.sample_part + 1) / ix.sample_parts; }
    return r;
}
// nodes of the row in front of segment j (j = count: the whole row; sample 0 is the state AFTER the start node, segment 0 starts with that node)
__device__ inline uint64_t segment_position(const DeviceIndex &ix, const RowSegments &r, uint64_t id, uint64_t j) {
    return j == 0 ? 0u : (j < r.count ? ix.samples[r.base + j * r.stride].w : ix.seq_len[id]);
}

// A chained step emits its first node x, the nodes between x and the node L of its landing record on the progression x + 2, x + 4, ...
// (x - 2, ... for reverse nodes), then L: a run of unary records with consecutive ids in one orientation, as a GFA segment chopped into
// nodes is (k_link_desc2).  A lane runs `n = x + d; n != L; n += d`.
__host__ __device__ inline uint32_t chain_stride(uint32_t x, uint32_t L) { return L > x ? 2u : 0xFFFFFFFEu; }
__host__ __device__ inline uint32_t chain_mids(uint32_t x, uint32_t L) { return (L > x ? L - x : x - L) / 2 - 1; }

__host__ __device__ inline uint32_t desc_body_offset(uint32_t meta) { return meta & 0xFFFFu; }
__host__ __device__ inline uint32_t desc_class(uint32_t meta) { return (meta >> 16) & 3u; }
__host__ __device__ inline uint64_t desc_start(uint32_t start_lo, uint32_t meta) { return (static_cast<uint64_t>(meta >> 24) << 32) | start_lo; }

}  // namespace gbwt_hip
