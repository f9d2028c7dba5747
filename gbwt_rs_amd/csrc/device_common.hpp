// device_common.hpp -- small helpers shared by the kernel translation units.
#pragma once

#include <hip/hip_runtime.h>

#include "device_index.hpp"

namespace gbwt_hip {

constexpr int WAVE = 64;

inline unsigned grid_for(uint64_t n, unsigned block) { return static_cast<unsigned>((n + block - 1) / block); }

// GBWT::forward's guards (src/gbwt.rs:222-229): the node is in the alphabet
__device__ __forceinline__ bool landing_record(const DeviceIndex &ix, uint32_t node, uint64_t &rec) {
    if (node < ix.first_node) return false;
    rec = node - ix.alphabet_offset;
    return rec < ix.n_records;
}

// One raw descriptor (device_index.hpp) in registers.
struct RawDesc { uint4 A, B, C, D; };

__device__ __forceinline__ bool load_raw_desc(const DeviceIndex &ix, uint64_t node, RawDesc &d, uint64_t &rec) {
    if (node < ix.first_node) return false;
    rec = node - ix.alphabet_offset;
    if (rec >= ix.n_records) return false;
    d.A = ix.desc_raw[4 * rec]; d.B = ix.desc_raw[4 * rec + 1]; d.C = ix.desc_raw[4 * rec + 2]; d.D = ix.desc_raw[4 * rec + 3];
    return d.B.y != 0;   // empty record / sigma == 0 -> None
}

}  // namespace gbwt_hip
