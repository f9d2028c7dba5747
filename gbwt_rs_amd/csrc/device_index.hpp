// device_index.hpp -- the index as the kernels see it (plain pointers into HBM).
//
// HBM layout (DESIGN.md "Data layout"):
//   data      : the record byte stream of bwt::BWT (src/bwt.rs:97-100), verbatim, followed by
//               DATA_PAD zero bytes so that unaligned 8/16-byte window loads never leave the buffer
//   starts    : dense record starts, n_records + 1 entries (sentinel = data_len); u32 when the stream
//               is < 4 GiB, else u64.  Replaces the Elias-Fano select of BWT::record_bytes
//               (src/bwt.rs:116-121): one 8-byte load gives [start, limit)
//   endmarker : record 0 fully decompressed at open (src/gbwt.rs:413-414), one (node, offset) per sequence
//   desc      : one 16-byte descriptor per record, built on the device at open, read by the walk kernels
//               with a single aligned dwordx4 load:
//                 ordinary record : x = start (low 32 bits), y = length in bytes, z = start (high 32 bits), w = outdegree (clamped)
//                 empty / None    : y = 0
//                 unary record    : y = DESC_UNARY, x = Record::len, z = successor node, w = successor offset
//               "unary" = outdegree 1 and a body that is exactly one run (every node on a linear stretch of
//               the graph): Record::lf(i) is then (z, w + i) for i < x, so a step costs one load and one add.
#pragma once

#include <cstdint>

namespace gbwt_hip {

constexpr uint32_t DESC_UNARY = 0xFFFFFFFFu;
constexpr uint32_t DATA_PAD = 128;  // lane 63 of a cooperative chunk reads up to 71 bytes past the chunk start

struct DeviceIndex {
    const uint8_t *data;
    const uint32_t *starts32;  // exactly one of starts32 / starts64 is non-null
    const uint64_t *starts64;
    const uint2 *endmarker;    // .x = node, .y = offset
    const uint4 *desc;         // n_records descriptors (see above)
    uint64_t data_len;
    uint64_t n_records;
    uint64_t n_sequences;      // header.sequences
    uint64_t n_endmarker;      // decompressed endmarker length
    uint32_t alphabet_offset;
    uint32_t first_node;       // alphabet_offset + 1
};

}  // namespace gbwt_hip
