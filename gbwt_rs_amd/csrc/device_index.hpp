// device_index.hpp -- the index as the kernels see it (plain pointers into HBM).
//
// HBM layout (DESIGN.md "Data layout"):
//   data      : the record byte stream of bwt::BWT (src/bwt.rs:97-100), verbatim, followed by
//               DATA_PAD zero bytes so that unaligned 8/16-byte window loads never leave the buffer
//   starts    : dense record starts, n_records + 1 entries (sentinel = data_len); u32 when the stream
//               is < 4 GiB, else u64.  Replaces the Elias-Fano select of BWT::record_bytes
//               (src/bwt.rs:116-121): one 8-byte load gives [start, limit)
//   endmarker : record 0 fully decompressed at open (src/gbwt.rs:413-414), one (node, offset) per sequence
//   desc      : one 32-byte descriptor per record (two uint4: A = desc[2 * rec], B = desc[2 * rec + 1]), built on the
//               device at open and read by the walk kernels with two aligned dwordx4 loads:
//                 ordinary record : A = {start low 32, length in bytes, successor 0, offset 0}
//                                   B = {successor 1, offset 1, body offset | class << 16, start high 32}
//                                   class 1 / 2 = outdegree 1 / 2 with the edge list decoded into A.z, A.w, B.x, B.y and
//                                   "body offset" = where the run stream starts inside the record; class 0 = anything else
//                 empty / None    : A.y = 0
//                 unary record    : A = {Record::len, DESC_UNARY, successor node, successor offset}
//               "unary" = outdegree 1 and a body that is exactly one run (every node on a linear stretch of
//               the graph): Record::lf(i) is then (A.z, A.w + i) for i < A.x, so a step costs one load and one add.
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
    const uint4 *desc;         // 2 * n_records entries (see above)
    uint64_t data_len;
    uint64_t n_records;
    uint64_t n_sequences;      // header.sequences
    uint64_t n_endmarker;      // decompressed endmarker length
    uint32_t alphabet_offset;
    uint32_t first_node;       // alphabet_offset + 1
};

}  // namespace gbwt_hip
