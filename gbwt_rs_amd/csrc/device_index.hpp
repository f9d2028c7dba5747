// device_index.hpp -- the index as the kernels see it (plain pointers into HBM).
//
// HBM layout (DESIGN.md "Data layout"):
//   data      : the record byte stream of bwt::BWT (src/bwt.rs:97-100), verbatim, followed by
//               DATA_PAD zero bytes so that unaligned 8/16-byte window loads never leave the buffer
//   starts    : dense record starts, n_records + 1 entries (sentinel = data_len); u32 when the stream
//               is < 4 GiB, else u64.  Replaces the Elias-Fano select of BWT::record_bytes
//               (src/bwt.rs:116-121): one 8-byte load gives [start, limit)
//   endmarker : record 0 fully decompressed at open (src/gbwt.rs:413-414), one (node, offset) per sequence
#pragma once

#include <cstdint>

namespace gbwt_hip {

constexpr uint32_t DATA_PAD = 128;  // lane 63 of a cooperative chunk reads up to 71 bytes past the chunk start

struct DeviceIndex {
    const uint8_t *data;
    const uint32_t *starts32;  // exactly one of starts32 / starts64 is non-null
    const uint64_t *starts64;
    const uint2 *endmarker;    // .x = node, .y = offset
    uint64_t data_len;
    uint64_t n_records;
    uint64_t n_sequences;      // header.sequences
    uint64_t n_endmarker;      // decompressed endmarker length
    uint32_t alphabet_offset;
    uint32_t first_node;       // alphabet_offset + 1
};

}  // namespace gbwt_hip
