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
//                     meta = body offset (bits 0-15) | class (bits 16-17) | fused flags (bits 18-19, walk descriptor
//                     only) | start (bits 32-39 of it, in bits 24-31)
//                     class 1 / 2 = outdegree 1 / 2 with the edges in A and "body offset" = where the run stream
//                     starts inside the record; class 0 = any other non-empty record (generic lane-serial decode)
//                 C = {value-0 positions of the record, 0, 0, 0}
//                 D = the first 16 bytes of the run stream, so short records need no second load
//               empty / None record : B.y = 0
//               unary record        : B.y = DESC_UNARY.  "Unary" = outdegree 1 and a body that is exactly one run
//                 (every node on a linear stretch of the graph): Record::lf(i) = (A.x, A.y + i) for i < B.w.
//   desc      : the walk descriptor derived from desc_raw (k_link_desc), fetched by the sampled walk with four
//               aligned dwordx4 loads that travel together: B and D as above, and one uint4 per edge,
//                 A = edge 0, C = edge 1 = {successor, offset base, landing node, sample base of the landing record}.
//               An edge whose successor is a unary record is FUSED with it (flag in meta): taking the edge emits the
//               successor and lands on the successor's successor at offset base + rank, so a walk spends one
//               iteration -- one round trip to memory -- on a branching node plus the unary node behind it.
//   sbase     : per record, index of its first rank sample or SAMPLE_NONE (needed where a walk starts; afterwards
//               the sample base of the next record rides along in C of the current one)
//   samples   : rank samples ("superblocks") of the long class 1 / 2 records, 32 bytes each (two uint4): sample k of
//               a record describes the run that contains offset k << sample_shift:
//                 S0 = {byte position of that run relative to the record start, offsets before the run,
//                       value-0 offsets before the run, 0},  S1 = 16 bytes of the run stream from that run on.
//               A lane fetches the descriptor and the sample of its offset in ONE round trip and usually finishes
//               the scan from S1 in registers, so a step costs the same whatever the length of the record (the
//               reference scans from the start of the record, src/bwt.rs:483-494).  Like simple-sds's rank/select
//               supports, descriptors and samples are rebuilt at load, never stored.
#pragma once

#include <cstdint>

namespace gbwt_hip {

constexpr uint32_t DESC_UNARY = 0xFFFFFFFFu;
constexpr uint32_t DESC_FUSED_SHIFT = 18;
constexpr uint32_t DESC_FUSED0 = 1u << DESC_FUSED_SHIFT;
constexpr uint32_t SAMPLE_NONE = 0xFFFFFFFFu;
constexpr uint32_t DATA_PAD = 128;  // lane 63 of a cooperative chunk reads up to 71 bytes past the chunk start

struct DeviceIndex {
    const uint8_t *data;
    const uint32_t *starts32;  // exactly one of starts32 / starts64 is non-null
    const uint64_t *starts64;
    const uint2 *endmarker;    // .x = node, .y = offset
    const uint4 *desc;         // 4 * n_records entries (walk descriptors)
    const uint4 *desc_raw;     // 4 * n_records entries (raw descriptors)
    const uint32_t *sbase;     // n_records entries
    const uint4 *samples;      // 2 * n_samples entries
    uint64_t data_len;
    uint64_t n_records;
    uint64_t n_sequences;      // header.sequences
    uint64_t n_endmarker;      // decompressed endmarker length
    uint64_t n_samples;
    uint32_t alphabet_offset;
    uint32_t first_node;       // alphabet_offset + 1
    uint32_t sample_shift;     // log2 of the sampling interval (in record offsets)
};

__host__ __device__ inline uint32_t desc_body_offset(uint32_t meta) { return meta & 0xFFFFu; }
__host__ __device__ inline uint32_t desc_class(uint32_t meta) { return (meta >> 16) & 3u; }
__host__ __device__ inline uint64_t desc_start(uint32_t start_lo, uint32_t meta) { return (static_cast<uint64_t>(meta >> 24) << 32) | start_lo; }

}  // namespace gbwt_hip
