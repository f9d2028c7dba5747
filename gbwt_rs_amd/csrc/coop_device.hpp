// coop_device.hpp -- wave-cooperative Record::lf for gfx950: all 64 lanes of a wavefront decode the
// run stream of ONE record together while several lanes (the "members") own a path standing in it.
//
// Why: in a pangenome the sequences of a batch tend to visit the same high-coverage records at the
// same time (lock-step through a chain of bubbles), and such records are the long ones: thousands of
// visits, tens to thousands of runs.  A lane-serial scan (lf_device.hpp) makes every lane repeat that
// scan byte by byte -- and a lone wavefront issues roughly one instruction every 4-5 cycles, so the
// step latency IS the instruction count.  Here each lane takes one byte of the run stream instead:
//
//   * every lane j loads an unaligned 8-byte window at body[done + j]  (one coalesced request per wave)
//   * run heads are found with two __ballot masks and ONE 64-bit scalar add: byte j continues a length
//     varint iff the carry into bit j of (C + O) is set, where C = "MSB set" and O = "head byte whose
//     length field is saturated" (RLE code of src/support.rs:1238-1248; for sigma <= 2 O implies C, so
//     generate = O and propagate = C)
//   * run ends and the per-value rank come from one DPP inclusive scan across the wave (both sums
//     packed into one register when every record is shorter than 2^16, else two scans)
//   * each member finds its run: small groups one member at a time with readlane / ballot broadcasts,
//     large groups with a 6-round binary search over ds_bpermute
//
// The edge list is not parsed here: the per-record descriptor built at open (device_index.hpp) carries
// the decoded edges of every record with outdegree <= 2 and the offset of its run stream.
// Semantics restated: Record::lf src/bwt.rs:480-496, RLEIter::next src/support.rs:1413-1430.
#pragma once

#include <hip/hip_runtime.h>

#include "device_index.hpp"
#include "lf_device.hpp"

namespace gbwt_hip {

constexpr int COOP_DONE = 0;         // results for all members are final
constexpr int COOP_UNSUPPORTED = 1;  // over-long run-length varint: caller must use the lane-serial path
constexpr int SERIAL_MEMBERS = 8;    // groups up to this size are resolved member by member instead of by ds_bpermute search

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

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

// Varint starting at the low byte of a 4-byte window (values < 2^28): the length extension of a long run.
// nbytes = 0 when no byte of the window terminates it.
__device__ __forceinline__ uint32_t window_varint32(uint32_t w, uint32_t &nbytes) {
    uint32_t stops = ~w & 0x80808080u;
    if (stops == 0) { nbytes = 0; return 0; }
    uint32_t t = static_cast<uint32_t>(__builtin_ctz(stops)) >> 3;
    nbytes = t + 1;
    uint32_t x = t == 3 ? w : (w & ((1u << (8 * (t + 1))) - 1));
    return (x & 0x7F) | ((x >> 1) & (0x7Fu << 7)) | ((x >> 2) & (0x7Fu << 14)) | ((x >> 3) & (0x7Fu << 21));
}

// Record::lf over the run stream body[0, body_len) of a record with outdegree 1 (two = false) or 2.
// Called by all 64 lanes with identical body / body_len / two; `member` marks the lanes whose path stands
// in this record at offset i, and (n0, o0, n1, o1) are that record's decoded edges (from the descriptor).
// PACK16: every record of the index is shorter than 2^16, so run ends and value-0 counts share one scan.
struct CoopProf { uint64_t load = 0, scan = 0, search = 0, t = 0; };  // cycle counters for experiments

template <bool PACK16, bool PROF = false>
__device__ __forceinline__ int coop_runs_lf(const uint8_t *body, uint32_t body_len, bool two, bool member, uint32_t i,
                                            uint32_t n0, uint32_t o0, uint32_t n1, uint32_t o1,
                                            bool &ok, uint32_t &out_node, uint32_t &out_offset, CoopProf *prof = nullptr) {
#define COOP_MARK(field) do { if (PROF) { uint64_t now_ = __builtin_amdgcn_s_memtime(); prof->field += now_ - prof->t; prof->t = now_; } } while (0)
    if (PROF) prof->t = __builtin_amdgcn_s_memtime();
    const uint32_t lane = lane_id();
    const uint32_t threshold = two ? 128u : 256u;  // RLE::sanitize: 256 / sigma
    const uint32_t saturated = two ? 254u : 255u;  // head bytes >= this have len == threshold and a varint follows
    uint32_t done = 0, base_cum = 0, base_c0 = 0;
    bool pending = member;
    if (member) ok = false;
    for (;;) {
        const uint32_t rem = body_len - done;
        const uint64_t w = load_u64_unaligned(body + done + lane);
        if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        COOP_MARK(load);
        const uint32_t b = static_cast<uint32_t>(w) & 0xFFu;
        const uint64_t in_body = rem >= 64 ? ~uint64_t(0) : ((uint64_t(1) << rem) - 1);
        const uint64_t C = __ballot(b >= 0x80u) & in_body;
        const uint64_t O = __ballot(b >= saturated) & in_body;
        uint64_t heads = in_body & ~((C + O) ^ C ^ O);  // carry-in set <=> the byte belongs to a length varint
        uint32_t value = two ? (b & 1u) : 0u;
        uint32_t len = (two ? (b >> 1) : b) + 1;
        uint32_t rbytes = 1;
        if (O != 0) {  // some head of this chunk carries a length extension
            uint32_t nb;
            const uint32_t extra = window_varint32(static_cast<uint32_t>(w >> 8), nb);
            const bool sat = b >= saturated;
            if ((__ballot(sat && nb == 0) & heads) != 0) return COOP_UNSUPPORTED;  // run longer than 2^28
            if (sat) { len = threshold + extra; rbytes = 1 + nb; }
        }
        bool exhausted = false;
        if (rem < 72) {  // a run cut off by the end of the record ends the stream (RLEIter::next -> None)
            const uint64_t cut = __ballot(lane + rbytes > rem) & heads;
            if (cut != 0) { heads &= (uint64_t(1) << __builtin_ctzll(cut)) - 1; exhausted = true; }
        }
        const bool is_head = __builtin_amdgcn_inverse_ballot_w64(heads);
        uint32_t cum, c0, total, total0;
        if (PACK16) {
            const uint32_t x = is_head ? (value ? len : len * 0x10001u) : 0u;      // low half: run ends, high half: value-0 positions
            const uint32_t scan = ((base_c0 << 16) | base_cum) + wave_inclusive_sum(x);
            cum = scan & 0xFFFFu; c0 = scan >> 16;
            const uint32_t t = read_lane(scan, 63);
            total = t & 0xFFFFu; total0 = t >> 16;
        } else {
            cum = base_cum + wave_inclusive_sum(is_head ? len : 0u);
            c0 = base_c0 + wave_inclusive_sum((is_head && value == 0) ? len : 0u);
            total = read_lane(cum, 63); total0 = read_lane(c0, 63);
        }
        COOP_MARK(scan);
        // members whose offset falls into this chunk: first lane with cum > i (cum is non-decreasing over the lanes)
        const bool hit = pending && i < total;
        const uint32_t packed = PACK16 ? ((c0 << 16) | cum) : cum;
        const uint32_t lv = (len << 1) | value;
        uint32_t packed_j = 0, lv_j = 0, c0_wide = 0;
        uint64_t hits = __ballot(hit);
        if (__builtin_popcountll(hits) <= SERIAL_MEMBERS) {
            // few members: resolve them one by one with scalar broadcasts (readlane / ballot / writelane, no LDS round trips)
            while (hits != 0) {
                const uint32_t p = static_cast<uint32_t>(__builtin_ctzll(hits));
                hits &= hits - 1;
                const uint32_t target = read_lane(i, p);
                const uint32_t j = static_cast<uint32_t>(__builtin_ctzll(__ballot(cum > target)));
                const bool mine = lane == p;
                packed_j = mine ? read_lane(packed, j) : packed_j;
                lv_j = mine ? read_lane(lv, j) : lv_j;
                if (!PACK16) c0_wide = mine ? read_lane(c0, j) : c0_wide;
            }
        } else {
            // many members: every lane binary-searches the scanned run ends through the LDS crossbar
            uint32_t j = 0;
#pragma unroll
            for (uint32_t s = 32; s >= 1; s >>= 1) {
                const uint32_t probe = __shfl(cum, static_cast<int>(j + s - 1));
                if (probe <= i) j += s;
            }
            j &= 63u;
            packed_j = __shfl(packed, static_cast<int>(j)); lv_j = __shfl(lv, static_cast<int>(j));
            if (!PACK16) c0_wide = __shfl(c0, static_cast<int>(j));
        }
        if (hit) {
            const uint32_t cum_j = PACK16 ? (packed_j & 0xFFFFu) : packed_j, c0_j = PACK16 ? (packed_j >> 16) : c0_wide;
            const uint32_t len_j = lv_j >> 1, val_j = lv_j & 1u;
            const uint32_t before = cum_j - len_j;                        // offset where the run starts
            const uint32_t rank = val_j ? (before - c0_j) : (c0_j - len_j);
            const uint32_t node = val_j ? n1 : n0;
            ok = node != 0;                                               // ENDMARKER successor: the sequence ends
            out_node = node;
            out_offset = (val_j ? o1 : o0) + rank + (i - before);
            pending = false;
        }
        COOP_MARK(search);
        if (__ballot(pending) == 0) return COOP_DONE;
        if (heads == 0 || exhausted) return COOP_DONE;  // stream exhausted: the remaining members are past the end (None)
        // the next chunk starts right after the last complete run of this one
        const uint32_t last = 63u - static_cast<uint32_t>(__builtin_clzll(heads));
        done += last + read_lane(rbytes, last);
        if (done >= body_len) return COOP_DONE;
        base_cum = total; base_c0 = total0;
    }
#undef COOP_MARK
}

}  // namespace gbwt_hip
