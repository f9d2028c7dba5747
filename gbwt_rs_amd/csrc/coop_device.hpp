// coop_device.hpp -- wave-cooperative Record::lf for gfx950: all 64 lanes of a wavefront decode ONE
// record together while several lanes (the "members") own a path standing in that record.
//
// Why: in a pangenome the sequences of a batch tend to visit the same high-coverage records at the
// same time (lock-step through a chain of bubbles), and such records are the long ones: thousands of
// visits, tens to thousands of runs.  A lane-serial scan (lf_device.hpp) makes every lane repeat that
// scan byte by byte; here each lane takes one byte of the run stream instead:
//
//   * every lane j loads an unaligned 8-byte window at data[pos + j]  (one coalesced request per wave)
//   * run starts are found with two __ballot masks and ONE 64-bit add: byte j continues a varint iff
//     the carry into bit j of (C + O) is set, where C = "byte has its MSB set" and O = "byte is a run
//     head whose length field is saturated" (RLE code of src/support.rs:1238-1248 for sigma < 255;
//     O implies C for sigma <= 64, so generate = O, propagate = C)
//   * run lengths and per-value ranks come from two DPP inclusive scans across the wave
//   * each member binary-searches the scanned run ends for its offset with 6 ds_bpermute rounds
//
// This file handles outdegree 1 and 2 (the bulk of a variation graph); other records are left to the
// lane-serial code by returning COOP_UNSUPPORTED.  Semantics restated: Record::lf src/bwt.rs:480-496,
// decompress_edges src/bwt.rs:378-395, RLEIter::next src/support.rs:1413-1430.
#pragma once

#include <hip/hip_runtime.h>

#include "device_index.hpp"
#include "lf_device.hpp"

namespace gbwt_hip {

constexpr int COOP_DONE = 0;         // results for all members are final
constexpr int COOP_UNSUPPORTED = 1;  // outdegree > 2, over-long varint, ...: caller must use the lane-serial path

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t mask_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Wave64 inclusive prefix sum with DPP (row shifts inside each row of 16, then row broadcasts).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xF, 0xF, false));  // row_shr:1
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xF, 0xF, false));  // row_shr:2
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xF, 0xF, false));  // row_shr:4
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xF, 0xF, false));  // row_shr:8
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
    return x;
}

__device__ __forceinline__ uint32_t read_lane(uint32_t x, uint32_t lane) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x), static_cast<int>(lane)));
}

__device__ __forceinline__ uint64_t read_lane64(uint64_t x, uint32_t lane) {
    return (static_cast<uint64_t>(read_lane(static_cast<uint32_t>(x >> 32), lane)) << 32) | read_lane(static_cast<uint32_t>(x), lane);
}

// Varint that starts at the low byte of `w` (ByteCode, 7 data bits per byte).  nbytes = 0 when no byte
// of the window terminates it (longer than 8 bytes).
__device__ __forceinline__ uint64_t window_varint64(uint64_t w, uint32_t &nbytes) {
    uint64_t stops = ~w & 0x8080808080808080ull;
    if (stops == 0) { nbytes = 0; return 0; }
    uint32_t t = static_cast<uint32_t>(__builtin_ctzll(stops)) >> 3;  // index of the terminating byte
    nbytes = t + 1;
    uint64_t x = t == 7 ? w : (w & ((uint64_t(1) << (8 * (t + 1))) - 1));
    return (x & 0x7F) | ((x >> 1) & (uint64_t(0x7F) << 7)) | ((x >> 2) & (uint64_t(0x7F) << 14)) | ((x >> 3) & (uint64_t(0x7F) << 21)) |
           ((x >> 4) & (uint64_t(0x7F) << 28)) | ((x >> 5) & (uint64_t(0x7F) << 35)) | ((x >> 6) & (uint64_t(0x7F) << 42)) |
           ((x >> 7) & (uint64_t(0x7F) << 49));
}

// Same for a 4-byte window (values < 2^28): the run-length extension of long runs.
__device__ __forceinline__ uint32_t window_varint32(uint32_t w, uint32_t &nbytes) {
    uint32_t stops = ~w & 0x80808080u;
    if (stops == 0) { nbytes = 0; return 0; }
    uint32_t t = static_cast<uint32_t>(__builtin_ctz(stops)) >> 3;
    nbytes = t + 1;
    uint32_t x = t == 3 ? w : (w & ((1u << (8 * (t + 1))) - 1));
    return (x & 0x7F) | ((x >> 1) & (0x7Fu << 7)) | ((x >> 2) & (0x7Fu << 14)) | ((x >> 3) & (0x7Fu << 21));
}

// Record::lf for the members of one group.  Must be called by all 64 lanes with identical
// (rec_start, rec_limit); `member` marks the lanes whose path stands in this record at offset `i`.
// On COOP_DONE: ok/out_node/out_offset are set for members (ok = false <=> lf() is None).
__device__ __forceinline__ int coop_record_lf(const DeviceIndex &ix, uint64_t rec_start, uint64_t rec_limit, bool member, uint32_t i,
                                              bool &ok, uint32_t &out_node, uint32_t &out_offset) {
    const uint32_t lane = lane_id();
    const uint64_t lane_bit = uint64_t(1) << lane;
    uint64_t pos = rec_start;
    uint64_t w = load_u64_unaligned(ix.data + pos + lane);
    uint64_t remaining = rec_limit - pos;
    uint64_t valid_mask = remaining >= 64 ? ~uint64_t(0) : ((uint64_t(1) << remaining) - 1);

    // ---- header: sigma, then sigma x (delta node, offset) -------------------------------------------
    uint32_t b = static_cast<uint32_t>(w) & 0xFFu;
    const uint64_t terminators = __ballot(b < 0x80u) & valid_mask;
    const uint64_t varint_starts = ((terminators << 1) | 1) & valid_mask;
    const bool is_vstart = (varint_starts & lane_bit) != 0;
    const uint32_t vidx = mask_rank(varint_starts);  // index of the varint starting at this lane
    uint32_t hbytes;
    const uint64_t hv = window_varint64(w, hbytes);
    if (remaining == 0) { if (member) ok = false; return COOP_DONE; }                 // Record::new -> None
    if (read_lane(hbytes, 0) == 0) return COOP_UNSUPPORTED;                            // sigma varint longer than 8 bytes
    const uint64_t sigma = read_lane64(hv, 0);
    if (sigma == 0) { if (member) ok = false; return COOP_DONE; }                     // Record::new -> None
    if (sigma > 2) return COOP_UNSUPPORTED;
    const uint32_t n_header = 1 + 2 * static_cast<uint32_t>(sigma);
    if (__ballot(is_vstart && vidx < n_header && hbytes == 0) != 0) return COOP_UNSUPPORTED;  // > 8-byte varint in the edge list
    // header varint k sits at the varint start with index k; all of them are inside this chunk (<= 50 bytes)
    uint32_t node0 = 0, off0 = 0, node1 = 0, off1 = 0;
    {
        const uint64_t m1 = __ballot(is_vstart && vidx == 1), m2 = __ballot(is_vstart && vidx == 2);
        const uint64_t m3 = __ballot(is_vstart && vidx == 3), m4 = __ballot(is_vstart && vidx == 4);
        const bool complete = m1 != 0 && m2 != 0 && (sigma == 1 || (m3 != 0 && m4 != 0));
        if (!complete) { if (member) ok = false; return COOP_DONE; }  // edge list runs past the record: malformed
        node0 = static_cast<uint32_t>(read_lane64(hv, static_cast<uint32_t>(__builtin_ctzll(m1))));
        off0 = static_cast<uint32_t>(read_lane64(hv, static_cast<uint32_t>(__builtin_ctzll(m2))));
        if (sigma == 2) {
            node1 = node0 + static_cast<uint32_t>(read_lane64(hv, static_cast<uint32_t>(__builtin_ctzll(m3))));
            off1 = static_cast<uint32_t>(read_lane64(hv, static_cast<uint32_t>(__builtin_ctzll(m4))));
        }
    }
    // first body byte = the varint start with index n_header (absent when the body is empty)
    uint64_t body_lanes = __ballot(is_vstart && vidx >= n_header);
    if (body_lanes == 0) { if (member) ok = false; return COOP_DONE; }  // no runs: every offset is past the end
    uint32_t lo = static_cast<uint32_t>(__builtin_ctzll(body_lanes));  // body starts at this lane of the current chunk

    // ---- body: run stream ----------------------------------------------------------------------------
    const bool one = sigma == 1;
    const uint32_t threshold = one ? 256u : 128u;   // RLE::sanitize: 256 / sigma
    const uint32_t saturated = one ? 255u : 254u;   // head bytes >= this carry len == threshold and a varint follows
    uint32_t base_cum = 0, base_c0 = 0;
    bool pending = member;
    if (member) ok = false;
    for (;;) {
        const uint64_t body_mask = valid_mask & ~((uint64_t(1) << lo) - 1);
        const bool in_body = (body_mask & lane_bit) != 0;
        const uint64_t O = __ballot(in_body && b >= saturated);
        const uint64_t C = __ballot(in_body && b >= 0x80u);
        const uint64_t cont = (C + O) ^ C ^ O;  // carry-in per bit: this byte belongs to a length varint
        bool is_start = in_body && (cont & lane_bit) == 0;
        // decode the run that starts here
        uint32_t value = one ? 0u : (b & 1u);
        uint32_t len = (one ? b : (b >> 1)) + 1;
        uint32_t rbytes = 1;
        bool too_long = false;
        if (b >= saturated) {
            uint32_t nb;
            uint32_t extra = window_varint32(static_cast<uint32_t>(w >> 8), nb);
            too_long = nb == 0;
            len = threshold + extra;
            rbytes = 1 + nb;
        }
        if (__ballot(is_start && too_long) != 0) return COOP_UNSUPPORTED;  // run longer than 2^28 + threshold
        // a run cut off by the end of the record ends the stream (RLEIter::next -> None)
        const uint64_t cut = __ballot(is_start && (pos + lane + rbytes > rec_limit));
        uint64_t start_mask = __ballot(is_start);
        if (cut != 0) start_mask &= (uint64_t(1) << __builtin_ctzll(cut)) - 1;
        is_start = (start_mask & lane_bit) != 0;
        const uint32_t run_len = is_start ? len : 0u;
        const uint32_t cum = base_cum + wave_inclusive_sum(run_len);                          // offset just past this run
        const uint32_t c0 = base_c0 + wave_inclusive_sum((is_start && value == 0) ? len : 0u); // value-0 positions so far
        const uint32_t total = read_lane(cum, 63), total0 = read_lane(c0, 63);
        // members whose offset falls into this chunk: first lane with cum > i
        const bool hit = pending && i < total;
        uint32_t j = 0;
#pragma unroll
        for (uint32_t s = 32; s >= 1; s >>= 1) {
            uint32_t probe = __shfl(cum, static_cast<int>(j + s - 1));
            if (probe <= i) j += s;
        }
        j &= 63u;
        const uint32_t lv = (len << 1) | value;
        const uint32_t cum_j = __shfl(cum, static_cast<int>(j)), c0_j = __shfl(c0, static_cast<int>(j)), lv_j = __shfl(lv, static_cast<int>(j));
        if (hit) {
            const uint32_t len_j = lv_j >> 1, val_j = lv_j & 1u;
            const uint32_t before = cum_j - len_j;                       // offset where the run starts
            const uint32_t rank = val_j ? (before - c0_j) : (c0_j - len_j);
            const uint32_t node = val_j ? node1 : node0;
            ok = node != 0;                                              // ENDMARKER successor: the sequence ends
            out_node = node;
            out_offset = (val_j ? off1 : off0) + rank + (i - before);
            pending = false;
        }
        if (__ballot(pending) == 0) return COOP_DONE;
        if (start_mask == 0 || cut != 0) return COOP_DONE;  // stream exhausted: remaining members are past the end (None)
        // next chunk starts right after the last complete run of this one
        const uint32_t last = 63u - static_cast<uint32_t>(__builtin_clzll(start_mask));
        pos += last + read_lane(rbytes, last);
        if (pos >= rec_limit) return COOP_DONE;
        base_cum = total; base_c0 = total0;
        remaining = rec_limit - pos;
        valid_mask = remaining >= 64 ? ~uint64_t(0) : ((uint64_t(1) << remaining) - 1);
        w = load_u64_unaligned(ix.data + pos + lane);
        b = static_cast<uint32_t>(w) & 0xFFu;
        lo = 0;
    }
}

}  // namespace gbwt_hip
