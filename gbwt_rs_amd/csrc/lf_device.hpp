// lf_device.hpp -- per-lane ("lane-serial") decode of one GBWT record on gfx950.
//
// One lane owns one query and walks its record's bytes through a 64-bit register window that is
// refilled with unaligned 8-byte loads (the record stream is byte-packed, so there is no useful
// alignment to exploit).  No edge table is materialised: where the reference builds a Vec<Pos>
// of sigma entries per step (src/bwt.rs:385-392, 481), these routines re-scan the (short) header,
// so the register footprint is independent of the out-degree.
//
// Reference semantics restated here (file:line into the reference):
//   ByteCodeIter::next   src/support.rs:1151-1164     varint()
//   RLEIter::next        src/support.rs:1413-1430     RunDecoder::next()
//   RLE::sanitize        src/support.rs:1292-1296     RunDecoder ctor
//   Record::new / decompress_edges  src/bwt.rs:341-351, 378-395
//   Record::lf           src/bwt.rs:480-496           record_lf()
//   Record::len          src/bwt.rs:449-455           record_len()
//   Record::edge_to      src/bwt.rs:543-555           (linear scan of the sorted edge list)
//   Record::follow / bd_follow   src/bwt.rs:595-616, 630-656   record_follow()
//   GBWT::forward        src/gbwt.rs:222-229          gbwt_forward()
//   Record::predecessor_at / offset_to   src/bwt.rs:502-540, 558-584   record_predecessor_at(), record_offset_to()
#pragma once

#include <hip/hip_runtime.h>

#include "device_index.hpp"

namespace gbwt_hip {

__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// Byte cursor over data[pos, limit).
struct ByteCursor {
    const uint8_t *data;
    uint64_t pos, limit;
    uint64_t window;
    uint32_t avail;

    __device__ __forceinline__ ByteCursor(const uint8_t *d, uint64_t start, uint64_t lim)
        : data(d), pos(start), limit(lim), window(0), avail(0) {}

    __device__ __forceinline__ void seek(uint64_t p) { pos = p; avail = 0; }
    __device__ __forceinline__ bool at_end() const { return pos >= limit; }

    // next byte; caller has checked !at_end()
    __device__ __forceinline__ uint32_t byte() {
        if (avail == 0) { window = load_u64_unaligned(data + pos); avail = 8; }
        uint32_t b = static_cast<uint32_t>(window) & 0xFFu;
        window >>= 8;
        avail--;
        pos++;
        return b;
    }

    // ByteCodeIter::next: false when the slice ends inside (or before) the integer.
    __device__ __forceinline__ bool varint(uint64_t &out) {
        uint64_t result = 0;
        uint32_t shift = 0;
        while (pos < limit) {
            uint32_t b = byte();
            if (shift < 64) result += static_cast<uint64_t>(b & 0x7Fu) << shift;
            shift += 7;
            if ((b & 0x80u) == 0) { out = result; return true; }
        }
        return false;
    }

    __device__ __forceinline__ bool skip_varint() {
        while (pos < limit) {
            if ((byte() & 0x80u) == 0) return true;
        }
        return false;
    }
};

// RLEIter for a fixed sigma (>= 1; sigma == 0 never reaches here because such records are "None").
struct RunDecoder {
    uint32_t sigma;      // clamped: anything >= 255 uses the two-varint code
    uint32_t threshold;  // 256 / sigma for sigma < 255, else 0
    uint32_t magic;      // floor(65536 / sigma) + 1: exact b / sigma for b <= 256, sigma < 255

    __device__ __forceinline__ explicit RunDecoder(uint64_t s) {
        sigma = s >= 255 ? 255u : static_cast<uint32_t>(s);
        magic = 65536u / sigma + 1u;
        threshold = sigma < 255 ? (256u * magic) >> 16 : 0u;
    }

    __device__ __forceinline__ bool next(ByteCursor &c, uint64_t &value, uint64_t &len) const {
        if (sigma >= 255) {
            uint64_t v, l;
            if (!c.varint(v)) return false;
            if (!c.varint(l)) return false;
            value = v; len = l + 1;
            return true;
        }
        if (c.at_end()) return false;
        uint32_t b = c.byte();
        uint32_t q = (b * magic) >> 16;  // b / sigma
        value = b - q * sigma;           // b % sigma
        uint64_t l = q + 1;
        if (l == threshold) {
            uint64_t extra;
            if (!c.varint(extra)) return false;
            l += extra;
        }
        len = l;
        return true;
    }
};

// [start, limit) of record `rec` (BWT::record_bytes); rec < n_records.
__device__ __forceinline__ void record_bounds(const DeviceIndex &ix, uint64_t rec, uint64_t &start, uint64_t &limit) {
    if (ix.starts32) { start = ix.starts32[rec]; limit = ix.starts32[rec + 1]; }
    else { start = ix.starts64[rec]; limit = ix.starts64[rec + 1]; }
}

// Opens the record of GBWT node `node` the way GBWT::forward / find / extend do: None when
// node < first_node, when the record id is out of range, when the slice is empty or sigma == 0.
// On success the cursor stands right after the sigma varint.
__device__ __forceinline__ bool open_record(const DeviceIndex &ix, uint64_t node, ByteCursor &c, uint64_t &sigma) {
    if (node < ix.first_node) return false;
    uint64_t rec = node - ix.alphabet_offset;
    if (rec >= ix.n_records) return false;
    uint64_t start, limit;
    record_bounds(ix, rec, start, limit);
    if (start >= limit) return false;
    c = ByteCursor(ix.data, start, limit);
    if (!c.varint(sigma)) return false;
    return sigma != 0;
}

// Record::lf on an opened record.  i = offset in the record.
__device__ __forceinline__ bool record_lf(ByteCursor &c, uint64_t sigma, uint64_t i, uint64_t &out_node, uint64_t &out_offset) {
    RunDecoder rd(sigma);
    if (sigma <= 2) {
        // Common case: the edge list lives in registers and one pass over the runs is enough.
        uint64_t n0 = 0, o0 = 0, n1 = 0, o1 = 0;
        if (!c.varint(n0) || !c.varint(o0)) return false;
        if (sigma == 2) {
            if (!c.varint(n1) || !c.varint(o1)) return false;
            n1 += n0;
        }
        uint64_t cum = 0, value, len;
        while (rd.next(c, value, len)) {
            if (cum + len > i) {
                uint64_t node = value ? n1 : n0;
                if (node == 0) return false;  // successor is the ENDMARKER: the sequence ends
                out_node = node;
                out_offset = (value ? o1 : o0) + (i - cum);
                return true;
            }
            if (value) o1 += len; else o0 += len;
            cum += len;
        }
        return false;
    }
    // General out-degree: (A) skip the edge list, (B) find the run holding offset i, (C) re-scan the
    // earlier runs for the rank of its value, (D) re-scan the edge list up to that value.
    const uint64_t header = c.pos;
    for (uint64_t k = 0; k < 2 * sigma; k++)
        if (!c.skip_varint()) return false;
    const uint64_t body = c.pos;
    uint64_t cum = 0, value, len, hit_value = 0, runs_before = 0;
    bool hit = false;
    while (rd.next(c, value, len)) {
        if (cum + len > i) { hit = true; hit_value = value; break; }
        cum += len;
        runs_before++;
    }
    if (!hit) return false;
    c.seek(body);
    uint64_t rank = 0;
    for (uint64_t k = 0; k < runs_before; k++) {
        rd.next(c, value, len);
        if (value == hit_value) rank += len;
    }
    c.seek(header);
    uint64_t node = 0, delta, off = 0;
    for (uint64_t e = 0; e <= hit_value; e++) {
        if (!c.varint(delta) || !c.varint(off)) return false;
        node += delta;
    }
    if (node == 0) return false;
    out_node = node;
    out_offset = off + rank + (i - cum);
    return true;
}

// GBWT::forward
__device__ __forceinline__ bool gbwt_forward(const DeviceIndex &ix, uint64_t node, uint64_t offset, uint64_t &out_node, uint64_t &out_offset) {
    ByteCursor c(ix.data, 0, 0);
    uint64_t sigma;
    if (!open_record(ix, node, c, sigma)) return false;
    return record_lf(c, sigma, offset, out_node, out_offset);
}

// Record::len on an opened record (cursor after sigma).
__device__ __forceinline__ uint64_t record_len(ByteCursor &c, uint64_t sigma) {
    for (uint64_t k = 0; k < 2 * sigma; k++)
        if (!c.skip_varint()) return 0;
    RunDecoder rd(sigma);
    uint64_t total = 0, value, len;
    while (rd.next(c, value, len)) total += len;
    return total;
}

__device__ __forceinline__ uint64_t overlap(uint64_t as, uint64_t ae, uint64_t bs, uint64_t be) {
    uint64_t s = as > bs ? as : bs, e = ae < be ? ae : be;
    return e > s ? e - s : 0;
}

// Record::follow (BD = false) / Record::bd_follow (BD = true) on an opened record.
// `count` (BD only) = number of positions in [start, end) whose successor s has flip(s) < flip(dest).
// Because the edge list is sorted by node, {s : flip(s) < flip(dest)} = {s < (dest & ~1)} plus,
// when dest is a forward node, its reverse partner dest + 1; so two integers picked up during the
// edge scan (`lo`, `partner`) replace the reference's per-run successor lookup (src/bwt.rs:646).
template <bool BD>
__device__ __forceinline__ bool record_follow(ByteCursor &c, uint64_t sigma, uint64_t start, uint64_t end, uint64_t dest,
                                              uint64_t &rstart, uint64_t &rend, uint64_t &count) {
    if (start >= end || dest == 0) return false;
    const uint64_t base = dest & ~uint64_t(1);
    uint64_t node = 0, rank = 0, edge_offset = 0, lo = 0, partner = ~uint64_t(0);
    bool found = false;
    for (uint64_t e = 0; e < sigma; e++) {
        uint64_t delta, off;
        if (!c.varint(delta) || !c.varint(off)) return false;
        node += delta;
        if (node == dest) { found = true; rank = e; edge_offset = off; }
        if (BD) {
            if (node < base) lo = e + 1;
            if (node == base + 1 && dest == base) partner = e;
        }
    }
    if (!found) return false;
    RunDecoder rd(sigma);
    uint64_t rs = edge_offset, re = edge_offset, cnt = 0, offset = 0, value, len;
    while (rd.next(c, value, len)) {
        if (value == rank) {
            rs += overlap(offset, offset + len, 0, start);
            re += overlap(offset, offset + len, 0, end);
        }
        if (BD && (value < lo || value == partner)) cnt += overlap(offset, offset + len, start, end);
        offset += len;
        if (offset >= end) break;
    }
    if (rs >= re) return false;
    rstart = rs; rend = re; count = cnt;
    return true;
}

// Record::predecessor_at on an opened record (cursor after sigma): the predecessor of the sequence at offset i of the
// OTHER orientation of this node.  The reference counts the positions per edge, flips the successors, swaps
// neighbours that are the two orientations of one node, and takes the first edge whose cumulative count exceeds i.
// Here the per-edge counts live in a 32-entry window (one pass over the runs per 32 edges; one pass in total for
// every record of practical out-degree) and the swap is a one-edge delay line: an edge is emitted after its
// successor in the list when both have the same node id.  (Three consecutive edges with one node id cannot occur:
// a node has two orientations.)
struct PredecessorScan {
    uint64_t i, cum = 0, result = 0;
    bool done = false, found = false;
    __device__ __forceinline__ explicit PredecessorScan(uint64_t offset) : i(offset) {}
    __device__ __forceinline__ void emit(uint64_t node, uint64_t count) {
        if (done) return;
        cum += count;
        if (cum > i) { done = true; found = node != 0; result = node; }   // ENDMARKER: None
    }
};

__device__ __forceinline__ bool record_predecessor_at(ByteCursor &c, uint64_t sigma, uint64_t i, uint64_t &out) {
    constexpr uint32_t WINDOW = 32;
    const uint64_t header = c.pos;
    for (uint64_t k = 0; k < 2 * sigma; k++)
        if (!c.skip_varint()) return false;
    const uint64_t body = c.pos;
    PredecessorScan scan(i);
    uint64_t pending_node = 0, pending_count = 0, node = 0, edge_pos = header;
    bool pending = false;
    for (uint64_t lo = 0; lo < sigma && !scan.done; lo += WINDOW) {
        const uint64_t hi = lo + WINDOW < sigma ? lo + WINDOW : sigma;
        uint32_t counts[WINDOW];
        for (uint32_t k = 0; k < WINDOW; k++) counts[k] = 0;
        c.seek(body);
        RunDecoder rd(sigma);
        uint64_t value, len;
        while (rd.next(c, value, len))
            if (value >= lo && value < hi) counts[value - lo] += static_cast<uint32_t>(len);
        c.seek(edge_pos);
        for (uint64_t e = lo; e < hi; e++) {
            uint64_t delta, off;
            if (!c.varint(delta) || !c.varint(off)) return false;
            node += delta;
            const uint64_t flipped = node == 0 ? 0 : node ^ 1;
            if (pending && (pending_node >> 1) == (flipped >> 1)) {
                scan.emit(flipped, counts[e - lo]);
                scan.emit(pending_node, pending_count);
                pending = false;
            } else {
                if (pending) scan.emit(pending_node, pending_count);
                pending = true; pending_node = flipped; pending_count = counts[e - lo];
            }
        }
        edge_pos = c.pos;
    }
    if (pending) scan.emit(pending_node, pending_count);
    out = scan.result;
    return scan.found;
}

// Record::offset_to on an opened record: the offset for which Record::lf would return (node, offset).
__device__ __forceinline__ bool record_offset_to(ByteCursor &c, uint64_t sigma, uint64_t node, uint64_t offset, uint64_t &out) {
    if (node == 0) return false;
    uint64_t cur = 0, outrank = 0, succ_rank = 0;
    bool found = false;
    for (uint64_t e = 0; e < sigma; e++) {   // Record::edge_to (the edge list is sorted and duplicate-free)
        uint64_t delta, off;
        if (!c.varint(delta) || !c.varint(off)) return false;
        cur += delta;
        if (cur == node) { found = true; outrank = e; succ_rank = off; }
    }
    if (!found || succ_rank > offset) return false;
    RunDecoder rd(sigma);
    uint64_t pos = 0, value, len;
    while (rd.next(c, value, len)) {
        pos += len;
        if (value != outrank) continue;
        succ_rank += len;
        if (succ_rank > offset) { out = pos - (succ_rank - offset); return true; }
    }
    return false;
}

}  // namespace gbwt_hip
