// kernels.hip -- gfx950 kernels of the GBWT LF-step path (hand-written HIP, no MFMA: this is
// integer pointer-chasing over a byte stream).  Launch wrappers are declared in kernels.hpp.
#include "kernels.hpp"

#include <hipcub/hipcub.hpp>

#include "coop_device.hpp"
#include "lf_device.hpp"

namespace gbwt_hip {

namespace {

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------------------------
// Load-time passes

// One lane per record: Record::len and outdegree maxima (sizes u32 offsets on device, feeds stats).
__global__ void __launch_bounds__(256) k_record_stats(DeviceIndex ix, uint64_t *stats) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec >= ix.n_records) return;
    uint64_t start, limit;
    record_bounds(ix, rec, start, limit);
    if (start >= limit) return;
    ByteCursor c(ix.data, start, limit);
    uint64_t sigma;
    if (!c.varint(sigma)) { atomicAdd(reinterpret_cast<unsigned long long *>(stats + 2), 1ull); return; }
    if (sigma == 0) return;
    uint64_t len = record_len(c, sigma);
    atomicMax(reinterpret_cast<unsigned long long *>(stats + 0), static_cast<unsigned long long>(len));
    atomicMax(reinterpret_cast<unsigned long long *>(stats + 1), static_cast<unsigned long long>(sigma));
}

// 16 bytes of the stream at data[pos..], zero-filled past `limit`.
__device__ __forceinline__ uint4 stream_bytes16(const uint8_t *data, uint64_t pos, uint64_t limit) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16 && pos + k < limit; k++) w[k >> 2] |= static_cast<uint32_t>(data[pos + k]) << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// One lane per record: the RAW descriptor (device_index.hpp) and the number of rank blocks the record gets.
__global__ void __launch_bounds__(256) k_build_desc(DeviceIndex ix, uint4 *desc, uint32_t *block_counts) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec >= ix.n_records) return;
    uint64_t start, limit;
    record_bounds(ix, rec, start, limit);
    uint4 A = make_uint4(0, 0, 0, 0), B = make_uint4(0, 0, 0, 0), C = make_uint4(0, 0, 0, 0), D = make_uint4(0, 0, 0, 0);
    uint32_t n_blocks = 0;
    if (limit > start) {
        ByteCursor c(ix.data, start, limit);
        uint64_t sigma = 0;
        if (c.varint(sigma) && sigma != 0) {
            B.x = static_cast<uint32_t>(start); B.y = static_cast<uint32_t>(limit - start);
            B.z = static_cast<uint32_t>((start >> 32) & 0xFF) << 24;
            bool classed = false;
            if (sigma <= 2 && (start >> 40) == 0) {
                uint64_t n0 = 0, o0 = 0, d1 = 0, o1 = 0;
                bool good = c.varint(n0) && c.varint(o0);
                if (good && sigma == 2) good = c.varint(d1) && c.varint(o1);
                const uint64_t body = c.pos - start;
                if (good && body <= 0xFFFF && n0 + d1 <= 0xFFFFFFFFull && o0 <= 0xFFFFFFFFull && o1 <= 0xFFFFFFFFull) {
                    // Record::len and the shape of the run stream.  The walk's scanner trusts class 1 / 2 streams: they
                    // must parse to the last byte and every run must fit in 1 + 4 bytes (length < 2^28 + threshold).
                    RunDecoder rd(sigma);
                    uint64_t total = 0, total0 = 0, runs = 0, value, len;
                    bool lean = true;
                    for (;;) {
                        const uint64_t before = c.pos;
                        if (!rd.next(c, value, len)) break;
                        if (c.pos - before > 5) lean = false;
                        total += len; runs++;
                        if (value == 0) total0 += len;
                    }
                    if (!c.at_end() || runs == 0) lean = false;
                    if (lean && total < 0xFFFFFFFFull) {
                        classed = true;
                        A.x = static_cast<uint32_t>(n0); A.y = static_cast<uint32_t>(o0);
                        A.z = static_cast<uint32_t>(n0 + d1); A.w = static_cast<uint32_t>(o1);
                        B.z |= static_cast<uint32_t>(body) | (static_cast<uint32_t>(sigma) << 16);
                        B.w = static_cast<uint32_t>(total);
                        C.x = static_cast<uint32_t>(total0); C.y = static_cast<uint32_t>(total);
                        D = stream_bytes16(ix.data, start + body, limit);
                        // outdegree 1: Record::lf(i) = (n0, o0 + i) however the body splits its runs
                        if (sigma == 1) B.y = DESC_UNARY;
                        else n_blocks = static_cast<uint32_t>((total >> RANK_BLOCK_SHIFT) + 1);  // position `total` is addressable too
                    }
                }
            }
            if (!classed) {  // class 0: B.w = 0 keeps the walk's fast path out; Record::len goes to C.y for find()
                A = make_uint4(0, 0, 0, 0); C = make_uint4(0, 0, 0, 0); D = make_uint4(0, 0, 0, 0); B.w = 0;
                ByteCursor c2(ix.data, start, limit);
                uint64_t s2 = 0;
                c2.varint(s2);
                const uint64_t total = record_len(c2, s2);
                C.y = total < 0xFFFFFFFFull ? static_cast<uint32_t>(total) : 0xFFFFFFFFu;
            }
        }
    }
    desc[4 * rec] = A;
    desc[4 * rec + 1] = B;
    desc[4 * rec + 2] = C;
    desc[4 * rec + 3] = D;
    block_counts[rec] = n_blocks;
}

// One lane per record: raw descriptor -> walk descriptor (device_index.hpp).  Everything the walk would otherwise
// test per step is decided here, once:
//  * an edge whose successor is a unary record is FUSED with it: the walk emits that successor and lands directly on
//    the successor's successor, one iteration (one round trip to memory) for two nodes;
//  * every edge knows whether the walk continues behind it (EDGE_CONT: the landing record exists and is not empty --
//    GBWT::forward's guards and BWT::record, src/gbwt.rs:222-229, src/bwt.rs:124-130), the record index of the
//    landing record and its block base;
//  * every offset an edge can produce is checked against the length of the landing record (offset base + number of
//    positions of this record that take the edge <= Record::len of the landing record; always true in a valid GBWT),
//    so the walk needs no "i >= Record::len -> None" test (src/bwt.rs:481).  A record with an edge that fails the
//    check is marked DESC_SLOW and goes through the generic decoder, which tests everything the reference tests.
__device__ __forceinline__ bool landing_record(const DeviceIndex &ix, uint32_t node, uint64_t &rec) {
    if (node < ix.first_node) return false;
    rec = node - ix.alphabet_offset;
    return rec < ix.n_records;
}

__global__ void __launch_bounds__(256) k_link_desc(DeviceIndex ix, uint4 *out) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec >= ix.n_records) return;
    const uint4 *raw = ix.desc_raw;
    const uint4 A = raw[4 * rec], B = raw[4 * rec + 1], C = raw[4 * rec + 2];
    uint4 E[2] = {make_uint4(0, 0, 0, BLOCK_NONE), make_uint4(0, 0, 0, BLOCK_NONE)};
    uint32_t flags[2] = {0, 0};
    const uint32_t cls = B.y != 0 ? desc_class(B.z) : 0u;
    bool slow = cls == 0;   // class 0 and empty records (the latter are never landed on)
    if (cls != 0) {
        const uint64_t count[2] = {cls == 2 ? C.x : B.w, cls == 2 ? B.w - C.x : 0u};
        const uint32_t succ[2] = {A.x, A.z}, off[2] = {A.y, A.w};
        for (uint32_t e = 0; e < cls; e++) {
            const uint32_t node = succ[e];
            uint64_t base = off[e], r = 0, land = 0;
            uint32_t z = 0, bb = BLOCK_NONE;
            bool cont = false, emit2 = false;
            if (node != 0 && landing_record(ix, node, r)) {
                const uint4 SA = raw[4 * r], SB = raw[4 * r + 1];
                if (SB.y == DESC_UNARY && base + count[e] <= SB.w) {
                    // plain edge to the unary record is safe; fuse when what lies behind it is safe too
                    cont = true; z = static_cast<uint32_t>(r);
                    const uint64_t base2 = base + SA.y;
                    if (SA.x == 0) { cont = false; z = 0; }            // the unary node is the last one of these sequences
                    else if (landing_record(ix, SA.x, land) && base2 + count[e] <= 0xFFFFFFFFull) {
                        const uint4 LB = raw[4 * land + 1];
                        if (LB.y != 0 && (desc_class(LB.z) == 0 || base2 + count[e] <= LB.w)) {
                            emit2 = true; z = static_cast<uint32_t>(land); base = base2; bb = ix.block_base[land];
                        }
                    }
                } else if (SB.y != 0) {
                    if (desc_class(SB.z) == 0 || base + count[e] <= SB.w) { cont = true; z = static_cast<uint32_t>(r); bb = ix.block_base[r]; }
                    else slow = true;
                }
            }
            E[e] = make_uint4(node, static_cast<uint32_t>(base), z, cont ? bb : BLOCK_NONE);
            flags[e] = (cont ? EDGE_CONT : 0u) | (emit2 ? EDGE_EMIT2 : 0u);
        }
    }
    if (rec == 0) {
        // record 0 (the endmarker) is never landed on (GBWT::forward, src/gbwt.rs:224): its walk descriptor is where
        // lanes without a walk are parked -- nothing to emit, does not continue, lands on record 0, not DESC_SLOW
        E[0] = E[1] = make_uint4(0, 0, 0, BLOCK_NONE); flags[0] = flags[1] = 0; slow = false;
    }
    out[4 * rec] = E[0];
    out[4 * rec + 1] = E[1];
    out[4 * rec + 2] = make_uint4(slow ? DESC_SLOW : 0u, flags[0], 0u, flags[1]);
    out[4 * rec + 3] = make_uint4(0u, 0u, 0u, 0u);
}

// One lane per record, after k_link_desc: the look-ahead targets.  For edge e: the record a walk that takes e
// reaches `hops` iterations later if it keeps taking edge 0 afterwards (a guess in general graphs; exact where the
// alleles of a site rejoin), as {first rank block, number of rank blocks} and, in slot 3 of the descriptor, its record
// index.  The helper wave of the walk touches that record's descriptor and one line of its block array per iteration,
// so both are already in the L2 of the XCD when the walk gets there.
__global__ void __launch_bounds__(256) k_link_lookahead(DeviceIndex ix, uint4 *desc, const uint32_t *block_counts, uint32_t hops) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec >= ix.n_records) return;
    uint4 D = desc[4 * rec + 2];
    if (D.x & DESC_SLOW) return;
    uint32_t base[2] = {0, 0}, count[2] = {0, 0}, target[2] = {0, 0};
    for (uint32_t e = 0; e < 2; e++) {
        uint64_t r = rec;
        uint32_t edge = e;
        bool good = true;
        for (uint32_t h = 0; h <= hops; h++) {
            const uint4 RD = desc[4 * r + 2];
            const uint32_t f = edge ? RD.w : RD.y;
            if ((RD.x & DESC_SLOW) || !(f & EDGE_CONT)) { good = false; break; }
            r = desc[4 * r + edge].z;
            edge = 0;
        }
        if (good) target[e] = static_cast<uint32_t>(r);
        if (good && ix.block_base[r] != BLOCK_NONE && ix.block_base[r] < DESC_SLOW) { base[e] = ix.block_base[r]; count[e] = block_counts[r] & LOOKAHEAD_COUNT_MASK; }
    }
    D.x |= base[0]; D.y |= count[0]; D.z = base[1]; D.w |= count[1];
    desc[4 * rec + 2] = D;
    desc[4 * rec + 3] = make_uint4(target[0], target[1], 0u, 0u);
}

// ---- LF tables for class 0 records -------------------------------------------------------------------------
// One lane per record: number of positions and outdegree of the records that get a table (0 for all others).
__global__ void __launch_bounds__(256) k_table_counts(DeviceIndex ix, uint64_t *positions, uint64_t *sigmas) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec >= ix.n_records) return;
    const uint4 B = ix.desc_raw[4 * rec + 1], C = ix.desc_raw[4 * rec + 2];
    uint64_t pos = 0, sigma = 0;
    if (rec != 0 && B.y != 0 && B.y != DESC_UNARY && desc_class(B.z) == 0 && C.y != 0xFFFFFFFFu && C.y != 0) {
        const uint64_t start = desc_start(B.x, B.z);
        ByteCursor c(ix.data, start, start + B.y);
        if (c.varint(sigma) && sigma != 0) pos = C.y; else sigma = 0;
    }
    positions[rec] = pos; sigmas[rec] = sigma;
}

// One lane per class 0 record: Record::decompress (src/bwt.rs:466-478) with the arrival tests of GBWT::forward folded in.
// `edges` is scratch: {successor, running offset} per edge of the record.
__global__ void __launch_bounds__(64) k_fill_tables(DeviceIndex ix, uint4 *desc_raw, const uint64_t *table_base, const uint64_t *edge_base, uint4 *tables,
                                                    uint2 *edges) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec >= ix.n_records) return;
    const uint64_t count = table_base[rec + 1] - table_base[rec];
    if (count == 0) return;
    const uint4 B = desc_raw[4 * rec + 1];
    const uint64_t start = desc_start(B.x, B.z);
    ByteCursor c(ix.data, start, start + B.y);
    uint64_t sigma = 0;
    c.varint(sigma);
    uint2 *e = edges + edge_base[rec];
    uint64_t node = 0;
    for (uint64_t k = 0; k < sigma; k++) {
        uint64_t delta = 0, off = 0;
        c.varint(delta); c.varint(off);
        node += delta;
        e[k] = make_uint2(static_cast<uint32_t>(node), static_cast<uint32_t>(off));
    }
    uint4 *out = tables + table_base[rec];
    RunDecoder rd(sigma);
    uint64_t pos = 0, value, len;
    while (pos < count && rd.next(c, value, len)) {
        if (value >= sigma) break;
        const uint32_t succ = e[value].x;
        uint32_t off = e[value].y;
        uint64_t land = 0;
        uint32_t lrec = 0, bb = BLOCK_NONE, llen = 0;
        bool exists = false, checked = false;
        if (succ != 0 && landing_record(ix, succ, land)) {
            const uint4 LB = desc_raw[4 * land + 1];
            exists = LB.y != 0;
            checked = desc_class(LB.z) != 0;   // records with descriptors: the offset must be inside (src/bwt.rs:481)
            llen = LB.w;
            if (exists) { lrec = static_cast<uint32_t>(land); bb = ix.block_base[land]; }
        }
        for (uint64_t k = 0; k < len && pos < count; k++, pos++, off++) {
            const bool cont = exists && (!checked || off < llen);
            out[pos] = make_uint4(succ, off, cont ? lrec : 0u, cont ? bb : BLOCK_NONE);
        }
        e[value].y = off;
    }
    uint4 C = desc_raw[4 * rec + 2];
    C.z = static_cast<uint32_t>(table_base[rec]); C.w = 1u;
    desc_raw[4 * rec + 2] = C;
}

// ---- two-step walk: descriptors and blocks -----------------------------------------------------------------
// The single-step descriptor says, per edge of record v: what to emit and where the walk lands (record w, offset base).
// The two-step descriptor composes that with the edges of w, so that one iteration of the walk -- one round trip to
// memory -- takes TWO LF steps (up to four nodes with fused unary successors):
//   desc2[8 * v + 0] = F0 = {node to emit for edge 0, offset base in w_0, node to emit for edge 1, offset base in w_1}
//   desc2[8 * v + 1] = F1 = {w_0 | LEAF_EMIT2 | DESC2_SLOW, w_1 | LEAF_EMIT2, 0, 0}      (first step; DESC2_SLOW: whole record)
//   desc2[8 * v + 2 + 2 * a + b] = leaf (a, b) = {node to emit, offset base, landing record | LEAF_EMIT2, block base}
//   desc2[8 * v + 6] = look-ahead {record, first block, number of blocks, 0};  [7] unused
// The second step exists (is "real") when w_a is unary (descriptor only: its value is always 0) or when both v and w_a
// have rank blocks: v's two-step block then carries, for each of its 64 offsets, the value the sequence has in w_a
// (bits2) and the number of value-1 positions of w_a before the landing offset of the block's first a-path (R_a), so
// the rank inside w_a is again one popcount.  Where the second step is not real (w_a generic, sequence ending, v
// unary and w_a branching) the leaf (a, 0) is the identity: "emit nothing, stay in w_a at the offset reached".
__global__ void __launch_bounds__(256) k_link_desc2(DeviceIndex ix, uint4 *out) {
    uint64_t v = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (v >= ix.n_records) return;
    const uint4 *d1 = ix.desc;
    const uint4 D = d1[4 * v + 2];
    const uint4 VB = ix.desc_raw[4 * v + 1];
    const uint32_t cls_v = VB.y != 0 ? desc_class(VB.z) : 0u;
    uint32_t n1[2] = {0, 0}, base[2] = {0, 0}, wword[2] = {0, 0};
    uint4 leaf[4] = {make_uint4(0, 0, 0, BLOCK_NONE), make_uint4(0, 0, 0, BLOCK_NONE), make_uint4(0, 0, 0, BLOCK_NONE), make_uint4(0, 0, 0, BLOCK_NONE)};
    const bool slow = (D.x & DESC_SLOW) != 0;
    if (!slow) {
        for (uint32_t a = 0; a < 2; a++) {
            const uint4 E = d1[4 * v + a];
            const uint32_t f = a ? D.w : D.y;
            n1[a] = E.x; base[a] = E.y;
            if (!(f & EDGE_CONT)) continue;                       // the walk ends behind this edge: both leaves park it
            const uint32_t w = E.z;
            wword[a] = w | ((f & EDGE_EMIT2) ? LEAF_EMIT2 : 0u);
            const uint4 WD = d1[4 * static_cast<uint64_t>(w) + 2];
            const uint4 WB = ix.desc_raw[4 * static_cast<uint64_t>(w) + 1];
            const uint32_t cls_w = WB.y != 0 ? desc_class(WB.z) : 0u;
            const bool real = !(WD.x & DESC_SLOW) && (cls_w == 1 || (cls_w == 2 && cls_v == 2));
            if (!real) { leaf[2 * a] = make_uint4(0u, 0u, w, E.w); continue; }   // identity
            for (uint32_t b = 0; b < cls_w; b++) {
                const uint4 WE = d1[4 * static_cast<uint64_t>(w) + b];
                const uint32_t wf = b ? WD.w : WD.y;
                const bool cont = (wf & EDGE_CONT) != 0;
                leaf[2 * a + b] = make_uint4(WE.x, WE.y, cont ? (WE.z | ((wf & EDGE_EMIT2) ? LEAF_EMIT2 : 0u)) : 0u, cont ? WE.w : BLOCK_NONE);
            }
        }
    }
    uint4 *o = out + 8 * v;
    o[0] = make_uint4(n1[0], base[0], n1[1], base[1]);
    o[1] = make_uint4(wword[0] | (slow ? DESC2_SLOW : 0u), wword[1], 0u, 0u);
    o[2] = leaf[0]; o[3] = leaf[1]; o[4] = leaf[2]; o[5] = leaf[3];
    o[6] = make_uint4(0u, 0u, 0u, 0u);
    o[7] = make_uint4(0u, 0u, 0u, 0u);
}

// One lane per record with rank blocks: the two-step blocks (32 bytes per 64 offsets):
//   cblocks[2 * k]     = {bits1 (values of v), bits2 (value in w_a of the sequence at each offset; 0 where the second
//                         step is not a real step through a record with blocks)}
//   cblocks[2 * k + 1] = {value-1 positions of v before the block, R_0, R_1, 0}
// The a-paths of a block land on consecutive offsets of w_a (LF keeps their order), so their values there are a
// contiguous bit range of w_a's blocks, spread back onto the positions of the a-paths.
__global__ void __launch_bounds__(256) k_fill_cblocks(DeviceIndex ix, const uint32_t *block_counts, uint4 *cblocks) {
    uint64_t v = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (v >= ix.n_records) return;
    const uint32_t count = block_counts[v];
    if (count == 0) return;
    const uint4 *d1 = ix.desc;
    const uint4 D = d1[4 * v + 2];
    const uint32_t len = ix.desc_raw[4 * v + 1].w;
    const uint32_t bb = ix.block_base[v];
    // per edge: landing record with blocks of its own, or none
    const uint4 *wblocks[2] = {nullptr, nullptr};
    uint32_t wbase[2] = {0, 0};
    if (!(D.x & DESC_SLOW)) {
        for (uint32_t a = 0; a < 2; a++) {
            const uint32_t f = a ? D.w : D.y;
            if (!(f & EDGE_CONT)) continue;
            const uint4 E = d1[4 * v + a];
            const uint64_t w = E.z;
            const uint4 WB = ix.desc_raw[4 * w + 1];
            if ((d1[4 * w + 2].x & DESC_SLOW) || WB.y == 0 || desc_class(WB.z) != 2) continue;
            wblocks[a] = ix.blocks + ix.block_base[w];
            wbase[a] = E.y;
        }
    }
    for (uint32_t k = 0; k < count; k++) {
        const uint4 P = ix.blocks[bb + k];
        const uint64_t bits1 = (static_cast<uint64_t>(P.y) << 32) | P.x;
        const uint32_t remaining = len - (k << RANK_BLOCK_SHIFT) > len ? 0u : len - (k << RANK_BLOCK_SHIFT);   // k * 64 <= len
        const uint64_t valid = remaining >= 64 ? ~uint64_t(0) : ((uint64_t(1) << remaining) - 1);
        uint64_t bits2 = 0;
        uint32_t R[2] = {0, 0};
        for (uint32_t a = 0; a < 2; a++) {
            if (!wblocks[a]) continue;
            uint64_t m = (a ? bits1 : ~bits1) & valid;
            const uint32_t cnt = __popcll(m);
            if (cnt == 0) continue;
            const uint32_t before = a ? P.z : (k << RANK_BLOCK_SHIFT) - P.z;          // a-paths of v before this block
            const uint32_t j = wbase[a] + before;                                      // where the first a-path lands in w_a
            const uint32_t q = j >> RANK_BLOCK_SHIFT, sh = j & 63u;
            const uint4 W0 = wblocks[a][q];
            const uint64_t w0 = (static_cast<uint64_t>(W0.y) << 32) | W0.x;
            R[a] = W0.z + __popcll(w0 & ((uint64_t(1) << sh) - 1));
            uint64_t val = w0 >> sh;
            if (sh != 0 && cnt > 64 - sh) {
                const uint4 W1 = wblocks[a][q + 1];
                val |= ((static_cast<uint64_t>(W1.y) << 32) | W1.x) << (64 - sh);
            }
            while (m) {                                                                 // spread the low cnt bits of val over the set bits of m
                const uint64_t low = m & (~m + 1);
                if (val & 1) bits2 |= low;
                val >>= 1;
                m ^= low;
            }
        }
        cblocks[2 * static_cast<uint64_t>(bb + k)] = make_uint4(P.x, P.y, static_cast<uint32_t>(bits2), static_cast<uint32_t>(bits2 >> 32));
        cblocks[2 * static_cast<uint64_t>(bb + k) + 1] = make_uint4(P.z, R[0], R[1], 0u);
    }
}

// One lane per record: where a walk that is at this record will be `hops` iterations later if it keeps taking leaf
// (0, 0) (a guess in general graphs; exact where the alleles of a site rejoin): {record, first block, number of blocks}.
__global__ void __launch_bounds__(256) k_link_lookahead2(DeviceIndex ix, uint4 *desc2, const uint32_t *block_counts, uint32_t hops) {
    uint64_t v = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (v >= ix.n_records) return;
    uint64_t r = v;
    bool good = true;
    for (uint32_t h = 0; h < hops && good; h++) {
        if (desc2[8 * r + 1].x & DESC2_SLOW) { good = false; break; }
        const uint32_t x = desc2[8 * r + 2].z & REC_MASK;
        if (x == 0) good = false; else r = x;
    }
    uint4 look = make_uint4(0u, 0u, 0u, 0u);
    if (good && r != v) {
        look.x = static_cast<uint32_t>(r);
        if (ix.block_base[r] != BLOCK_NONE) { look.y = ix.block_base[r]; look.z = block_counts[r]; }
    }
    desc2[8 * v + 6] = look;
}

// One lane per outdegree-2 record: decode the runs ONCE and lay the record out as rank blocks (device_index.hpp):
// block k = {64 values (one bit each), value-1 positions before the block}.  Record::lf (src/bwt.rs:480-496) at
// offset i is then value = bit i, rank = ones-before or i - ones-before, without scanning any run.
__global__ void __launch_bounds__(256) k_fill_blocks(DeviceIndex ix, const uint32_t *block_counts, const uint32_t *block_base, uint4 *blocks) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec >= ix.n_records) return;
    const uint32_t count = block_counts[rec];
    if (count == 0) return;
    const uint4 B = ix.desc_raw[4 * rec + 1];
    const uint64_t start = desc_start(B.x, B.z), limit = start + B.y;
    ByteCursor c(ix.data, start + desc_body_offset(B.z), limit);
    RunDecoder rd(desc_class(B.z));
    uint4 *out = blocks + block_base[rec];
    uint64_t bits = 0, value, len;
    uint32_t k = 0, fill = 0, ones = 0;
    while (k < count && rd.next(c, value, len)) {
        while (len > 0 && k < count) {
            const uint32_t take = static_cast<uint32_t>(len < 64 - fill ? len : 64 - fill);
            if (value) bits |= (take == 64 ? ~uint64_t(0) : ((uint64_t(1) << take) - 1)) << fill;
            fill += take; len -= take;
            if (fill == 64) {
                out[k++] = make_uint4(static_cast<uint32_t>(bits), static_cast<uint32_t>(bits >> 32), ones, 0u);
                ones += __popcll(bits);
                bits = 0; fill = 0;
            }
        }
    }
    if (k < count) out[k] = make_uint4(static_cast<uint32_t>(bits), static_cast<uint32_t>(bits >> 32), ones, 0u);
}

// Outdegree + length of the endmarker record (record 0), to size the decompression scratch.
__global__ void k_endmarker_sigma(DeviceIndex ix, uint64_t *result) {
    result[0] = 0; result[1] = 0;
    if (ix.n_records == 0) return;
    uint64_t start, limit;
    record_bounds(ix, 0, start, limit);
    if (start >= limit) return;
    ByteCursor c(ix.data, start, limit);
    uint64_t sigma;
    if (!c.varint(sigma) || sigma == 0) return;
    result[1] = sigma;
    result[0] = record_len(c, sigma);
}

// Record::decompress (src/bwt.rs:465-475) of the endmarker record, done once at open.  Single
// lane: the record has one run per sequence in the worst case and this is load-time work.
__global__ void k_endmarker_decompress(DeviceIndex ix, uint2 *out, uint64_t n_out, uint64_t *scratch, uint64_t *result) {
    uint64_t start, limit;
    record_bounds(ix, 0, start, limit);
    ByteCursor c(ix.data, start, limit);
    uint64_t sigma = 0;
    c.varint(sigma);
    uint64_t *nodes = scratch, *offsets = scratch + sigma;
    uint64_t node = 0;
    for (uint64_t e = 0; e < sigma; e++) {
        uint64_t delta = 0, off = 0;
        c.varint(delta); c.varint(off);
        node += delta;
        nodes[e] = node; offsets[e] = off;
    }
    RunDecoder rd(sigma);
    uint64_t produced = 0, value, len;
    while (rd.next(c, value, len)) {
        if (value >= sigma) break;  // malformed
        for (uint64_t k = 0; k < len && produced < n_out; k++) {
            out[produced++] = make_uint2(static_cast<uint32_t>(nodes[value]), static_cast<uint32_t>(offsets[value]));
            offsets[value]++;
        }
    }
    result[0] = produced;
}

// ---------------------------------------------------------------------------------------------
// Path extraction: one lane per sequence (GBWT::sequence + SequenceIter::next, src/gbwt.rs:253-261,
// 557-568).  Lengths are unknown until a sequence ends, so every lane appends the nodes it visits
// to a chain of 1 KiB blocks drawn from a shared pool; a second, bandwidth-bound kernel lays the
// chains out as CSR once the lengths (and their prefix sums) exist.

// Output sink of one lane: the nodes a sequence visits go to a chain of 1 KiB pool blocks.  Stores are staged in
// LDS (SINK_STAGE entries per lane, entry-major so a wave's writes hit 64 distinct banks) and flushed as 64-byte
// pieces: on gfx9 loads and stores share the in-order vmcnt counter, so a store issued every step would put its
// ~700-cycle acknowledgement on the critical path of the next dependent load.
constexpr uint32_t SINK_STAGE = 16;
static_assert(POOL_BLOCK_NODES % SINK_STAGE == 0, "a block must hold a whole number of flushes");

struct PathSink {
    uint32_t *stage;             // this lane's column of the wave's LDS staging buffer (stride WAVE)
    uint32_t *wp = nullptr;      // next slot in the current block
    uint32_t left = 0;           // free slots in the current block
    uint32_t staged = 0;         // entries waiting in LDS
    uint32_t cur = POOL_NONE, head = POOL_NONE, blocks = 0;
    bool overflow = false;
    __device__ __forceinline__ explicit PathSink(uint32_t *lds, uint32_t lane) : stage(lds + lane) {}
    __device__ __forceinline__ void flush(const WalkArgs &a) {
        if (staged == 0 || overflow) { staged = 0; return; }
        if (left == 0) {
            uint32_t nb = atomicAdd(a.counter, 1u);
            if (nb >= a.pool_blocks) { atomicOr(a.flags, FLAG_POOL_OVERFLOW); overflow = true; staged = 0; return; }
            a.next[nb] = POOL_NONE;
            if (cur == POOL_NONE) head = nb; else a.next[cur] = nb;
            cur = nb; blocks++;
            wp = a.pool + static_cast<uint64_t>(nb) * POOL_BLOCK_NODES;
            left = POOL_BLOCK_NODES;
        }
        if (staged == SINK_STAGE) {
            uint4 *dst = reinterpret_cast<uint4 *>(wp);
#pragma unroll
            for (uint32_t q = 0; q < SINK_STAGE / 4; q++)
                dst[q] = make_uint4(stage[(4 * q) * WAVE], stage[(4 * q + 1) * WAVE], stage[(4 * q + 2) * WAVE], stage[(4 * q + 3) * WAVE]);
        } else {
            for (uint32_t e = 0; e < staged; e++) wp[e] = stage[e * WAVE];
        }
        wp += staged; left -= staged; staged = 0;
    }
    // after a pool overflow the sink drops what it gets: the host grows the pool and walks again
    __device__ __forceinline__ void push(const WalkArgs &a, uint32_t node) {
        stage[staged * WAVE] = node;
        staged++;
        if (__builtin_expect(staged == SINK_STAGE, 0)) flush(a);
    }
    __device__ __forceinline__ uint64_t finish(const WalkArgs &a) {
        flush(a);
        return static_cast<uint64_t>(blocks) * POOL_BLOCK_NODES - left;
    }
};

__global__ void __launch_bounds__(WAVE) k_walk(DeviceIndex ix, WalkArgs a) {
    __shared__ uint32_t sink_lds[SINK_STAGE * WAVE];
    PathSink sink(sink_lds, threadIdx.x);
    uint64_t k = blockIdx.x * static_cast<uint64_t>(WAVE) + threadIdx.x;
    if (k >= a.n) return;
    const uint64_t id = a.seq_ids[k];
    uint64_t node = 0, offset = 0;
    bool valid = false;
    if (id < ix.n_endmarker) {  // GBWT::start, src/gbwt.rs:213-219
        uint2 e = ix.endmarker[id];
        node = e.x; offset = e.y;
        valid = node != 0;
    }
    while (valid) {
        sink.push(a, static_cast<uint32_t>(node));
        if (sink.overflow) break;
        uint64_t nn, no;
        valid = gbwt_forward(ix, node, offset, nn, no);
        node = nn; offset = no;
    }
    a.lengths[k] = sink.finish(a);
    a.head[k] = sink.head;
}


// Generic lane-serial Record::lf on the record bytes [start, start + bytes) (class 0 records, fallbacks).  Out of
// line and by-value only, so that the hot loops stay small and nothing is forced into scratch.  Returns
// (node, offset); node == 0 <=> None.
__device__ __attribute__((noinline)) uint2 serial_record_lf(const uint8_t *data, uint64_t start, uint32_t bytes, uint32_t offset) {
    ByteCursor c(data, start, start + bytes);
    uint64_t sigma, nn, no;
    if (c.varint(sigma) && sigma != 0 && record_lf(c, sigma, offset, nn, no)) return make_uint2(static_cast<uint32_t>(nn), static_cast<uint32_t>(no));
    return make_uint2(0u, 0u);
}

// Output staging of the default walk: a ring of RING slots per lane in LDS.  Pushes are unconditional LDS writes
// (the slot only advances when the node counts), and a lane moves 16 slots = 64 bytes to its pool block with four
// dwordx4 stores whenever that many are waiting.  The pool is the same chain of POOL_BLOCK_NODES-sized blocks as
// PathSink's.  After a pool overflow the sink drops what it gets: the host grows the pool and walks again.
constexpr uint32_t RING = 64;                // single-step walk: at most 2 nodes per iteration
constexpr uint32_t RING2 = 128;              // two-step walk: at most 4 nodes per iteration
constexpr uint32_t RING_FLUSH = 16;
constexpr uint32_t RING_URGENT = RING - 4;   // the hot loop hands over to the flush code once a lane has more than this waiting
constexpr uint32_t RING2_URGENT = RING2 - 8;
static_assert(POOL_BLOCK_NODES % RING_FLUSH == 0, "a block must hold a whole number of flushes");

template <uint32_t SLOTS>
struct RingSinkT {
    uint32_t *stage;             // this lane's column of the ring: slot s at stage[s * WAVE]
    uint32_t wr = 0, flushed = 0;   // nodes pushed / nodes written to the pool
    uint32_t *wp = nullptr;
    uint32_t left = 0;
    uint32_t cur = POOL_NONE, head = POOL_NONE, blocks = 0;
    bool overflow = false;
    __device__ __forceinline__ RingSinkT(uint32_t *lds, uint32_t lane) : stage(lds + lane) {}
    __device__ __forceinline__ void push(uint32_t node, bool counts) {
        stage[(wr & (SLOTS - 1)) * WAVE] = node;
        wr += counts ? 1u : 0u;
    }
    __device__ __forceinline__ bool needs_flush() const { return wr - flushed >= RING_FLUSH; }
    __device__ __forceinline__ bool new_block(const WalkArgs &a) {
        uint32_t nb = atomicAdd(a.counter, 1u);
        if (nb >= a.pool_blocks) { atomicOr(a.flags, FLAG_POOL_OVERFLOW); overflow = true; return false; }
        a.next[nb] = POOL_NONE;
        if (cur == POOL_NONE) head = nb; else a.next[cur] = nb;
        cur = nb; blocks++;
        wp = a.pool + static_cast<uint64_t>(nb) * POOL_BLOCK_NODES;
        left = POOL_BLOCK_NODES;
        return true;
    }
    __device__ __forceinline__ void flush16(const WalkArgs &a) {
        if (!overflow && (left != 0 || new_block(a))) {
            const uint32_t *src = stage + (flushed & (SLOTS - 1)) * WAVE;   // slot 0, 16, 32 or 48
            uint4 *dst = reinterpret_cast<uint4 *>(wp);
#pragma unroll
            for (uint32_t q = 0; q < RING_FLUSH / 4; q++)
                dst[q] = make_uint4(src[(4 * q) * WAVE], src[(4 * q + 1) * WAVE], src[(4 * q + 2) * WAVE], src[(4 * q + 3) * WAVE]);
            wp += RING_FLUSH; left -= RING_FLUSH;
        }
        flushed += RING_FLUSH;
    }
    __device__ __forceinline__ uint64_t finish(const WalkArgs &a) {
        while (needs_flush()) flush16(a);
        const uint32_t tail = wr - flushed;
        if (tail != 0 && !overflow && (left != 0 || new_block(a))) {
            for (uint32_t e = 0; e < tail; e++) wp[e] = stage[((flushed + e) & (SLOTS - 1)) * WAVE];
            left -= tail;
        }
        return static_cast<uint64_t>(blocks) * POOL_BLOCK_NODES - left;
    }
};
using RingSink = RingSinkT<RING>;

// Arrival at (node, offset) from outside the linked descriptors (the start of a sequence, a generic step): the tests
// of GBWT::forward / BWT::record / Record::lf (src/gbwt.rs:222-229, src/bwt.rs:124-130, 481) that k_link_desc
// settles in advance for the linked edges.
__device__ __forceinline__ bool arrive(const DeviceIndex &ix, uint32_t node, uint32_t offset, uint32_t &rec, uint32_t &bb) {
    uint64_t r;
    if (!landing_record(ix, node, r)) return false;
    const uint4 LB = ix.desc_raw[4 * r + 1];
    if (LB.y == 0 || (desc_class(LB.z) != 0 && offset >= LB.w)) return false;
    rec = static_cast<uint32_t>(r); bb = ix.block_base[r];
    return true;
}

// One generic step of a walk at a DESC_SLOW record: a lookup in the record's LF table when it has one, else Record::lf
// on the record bytes followed by the arrival tests.
template <class Sink>
__device__ __forceinline__ void generic_step(const DeviceIndex &ix, Sink &sink, uint32_t &rec, uint32_t &offset, uint32_t &bb) {
    const uint4 B = ix.desc_raw[4 * static_cast<uint64_t>(rec) + 1], C = ix.desc_raw[4 * static_cast<uint64_t>(rec) + 2];
    if (C.w == 1u) {
        uint4 e = make_uint4(0u, 0u, 0u, BLOCK_NONE);
        if (offset < C.y) e = ix.tables[static_cast<uint64_t>(C.z) + offset];   // i >= Record::len -> None (src/bwt.rs:481)
        sink.push(e.x, e.x != 0);
        offset = e.y; rec = e.z; bb = e.w;
        return;
    }
    const uint2 r = serial_record_lf(ix.data, desc_start(B.x, B.z), B.y, offset);
    sink.push(r.x, r.x != 0);
    offset = r.y;
    if (r.x == 0 || !arrive(ix, r.x, r.y, rec, bb)) { rec = 0; bb = BLOCK_NONE; }
}

// The hot loop of the default walk, written in gfx950 assembly: hipcc's register shuffling around the cold paths and
// its SGPR mask algebra more than doubled the instruction count of the loop, and the position of every load and wait
// matters here.
//
// All 64 lanes run every instruction; lanes without a walk are PARKED on record 0, whose walk descriptor says
// "nothing to emit, lands on record 0" and which reads the zero block, so a parked lane stays parked.
// One iteration (Record::lf src/bwt.rs:480-496 + GBWT::forward src/gbwt.rs:222-229 for one or, fused, two nodes):
//     wait for A, C, D (walk descriptor) and K (rank block)                      s_waitcnt vmcnt(0)
//     any lane on a DESC_SLOW record -> leave BEFORE changing any state          (generic decode outside)
//     value = bit `offset` of K, ones = K.z + popcount(K bits below `offset`)
//     rank = value ? ones : offset - ones;  E = value ? C : A;  flags/look-ahead = value ? D.zw : D.xy
//     rec = E.z; offset = E.y + rank; bb = E.w                                    (the new position)
//     issue the four loads of the new position; post the look-ahead target in the helper wave's mailbox
//     push E.x (counts if != 0), push rec + alphabet_offset (counts if EDGE_EMIT2) into the LDS ring
//     leave if no lane is walking any more, or a lane has more than RING_URGENT nodes waiting in its ring
// On exit nothing is in flight (vmcnt(0), lgkmcnt(0)).  Returns 1 when it left because of a DESC_SLOW record.
// Hazards: a VALU write of VCC / an SGPR needs two wait states before a VALU reads it (gfx940+); the string keeps two
// independent instructions (or an s_nop) in every such pair.  Registers v40-v89 and s41, s44-s45 are named literally
// and listed as clobbers.
__device__ __forceinline__ uint32_t walk_hot_loop(const uint4 *desc, const uint4 *blocks, uint32_t alphabet_offset, uint32_t ring_base,
                                                  uint32_t mail_slot, uint32_t flushed, uint32_t &rec, uint32_t &offset, uint32_t &bb, uint32_t &wr, uint32_t &hash) {
#ifdef GBWT_HIP_CXX_LOOP
    // the same loop in plain C++ (no pipelining, no look-ahead): what the assembly below must compute
    for (;;) {
        const uint4 *d = desc + 4 * static_cast<uint64_t>(rec);
        const uint4 A = d[0], C = d[1], D = d[2];
        const uint4 K = blocks[bb == BLOCK_NONE ? 0u : bb + (offset >> RANK_BLOCK_SHIFT)];
        if (__ballot(static_cast<int32_t>(D.x) < 0) != 0) return 1;
        const uint64_t bits = (static_cast<uint64_t>(K.y) << 32) | K.x;
        const uint32_t bit = offset & 63u;
        const uint32_t value = static_cast<uint32_t>(bits >> bit) & 1u;
        const uint32_t ones = K.z + __popcll(bits & ((uint64_t(1) << bit) - 1));
        const uint32_t rank = value ? ones : offset - ones;
        const uint4 E = value ? C : A;
        const uint32_t flags = value ? D.w : D.y;
        rec = E.z; offset = E.y + rank; bb = E.w;
        __attribute__((address_space(3))) uint32_t *ring = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)ring_base;
        ring[(wr & (RING - 1)) * WAVE] = E.x;
        wr += E.x != 0 ? 1u : 0u;
        ring[(wr & (RING - 1)) * WAVE] = rec + alphabet_offset;
        wr += static_cast<int32_t>(flags) < 0 ? 1u : 0u;
        if (__ballot(rec != 0) == 0 || __ballot(wr - flushed > RING_URGENT) != 0) return 0;
    }
#else
    uint32_t reason;
#define GBWT_WALK_ISSUE                                                                                   \
    "v_cmp_ne_u32_e32 vcc, -1, v43\n\t"                   /* bb != BLOCK_NONE */                          \
    "v_lshrrev_b32_e32 v70, 6, v42\n\t"                                                                   \
    "v_add_u32_e32 v70, v70, v43\n\t"                                                                     \
    "v_lshlrev_b32_e32 v82, 2, v40\n\t"                 /* v_lshl_add_u64 shifts by at most 4 */        \
    "v_cndmask_b32_e32 v70, 0, v70, vcc\n\t"              /* block bb + offset / 64, or the zero block */ \
    "v_lshl_add_u64 v[68:69], v[70:71], 4, %[blocks]\n\t"                                                 \
    "v_lshl_add_u64 v[66:67], v[82:83], 4, %[desc]\n\t"   /* descriptor of rec (64 bytes each) */        \
    "s_mov_b64 exec, s[44:45]\n\t"                      /* only the lanes that were walking before this step load: a  */ \
    "global_load_dwordx4 v[60:63], v[68:69], off\n\t"   /* lane that has just parked fetches the parking descriptor   */ \
    "global_load_dwordx4 v[48:51], v[66:67], off\n\t"   /* once and keeps it; the texture path spends cycles on every */ \
    "global_load_dwordx4 v[52:55], v[66:67], off offset:16\n\t" /* enabled lane                                       */ \
    "global_load_dwordx4 v[56:59], v[66:67], off offset:32\n\t"                                           \
    "global_load_dwordx2 v[90:91], v[66:67], off offset:48\n\t"                                           \
    "s_mov_b64 exec, -1\n\t"                                                                              \
    "v_add_u32_e32 v86, 0x9e3779b1, v86\n\t"              /* new sequence number = new pseudo-random number */ \
    "ds_write_b128 %[mail], v[84:87]\n\t"                 /* look-ahead target for the helper wave */
    asm volatile(
        "v_mov_b32_e32 v40, %[rec]\n\t"
        "v_mov_b32_e32 v83, 0\n\t"
        "v_mov_b32_e32 v42, %[offset]\n\t"
        "v_mov_b32_e32 v43, %[bb]\n\t"
        "v_mov_b32_e32 v44, %[wr]\n\t"
        "v_mov_b32_e32 v86, %[hash]\n\t"
        "v_mov_b32_e32 v87, 0\n\t"
        "v_mov_b32_e32 v71, 0\n\t"
        "v_mov_b32_e32 v84, 0\n\t"
        "v_mov_b32_e32 v85, 0\n\t"
        "s_mov_b32 %[reason], 0\n\t"
        "s_mov_b64 s[44:45], -1\n\t"
        GBWT_WALK_ISSUE
        ".Lgbwt_walk_loop_%=:\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_cmp_gt_i32_e32 vcc, 0, v56\n\t"                  /* DESC_SLOW = sign of D.x */
        "v_lshrrev_b64 v[76:77], v42, v[60:61]\n\t"         /* bit `offset & 63` -> bit 0 */
        "v_lshlrev_b64 v[78:79], v42, -1\n\t"               /* bits at and above it */
        "s_cbranch_vccnz .Lgbwt_walk_slow_%=\n\t"
        "v_and_b32_e32 v76, 1, v76\n\t"                     /* value */
        "v_bfi_b32 v78, v78, 0, v60\n\t"                    /* K bits below */
        "v_bfi_b32 v79, v79, 0, v61\n\t"
        "v_cmp_eq_u32_e32 vcc, 1, v76\n\t"
        "v_bcnt_u32_b32 v78, v78, v62\n\t"                  /* + value-1 positions before the block */
        "v_bcnt_u32_b32 v78, v79, v78\n\t"                  /* ones */
        "v_sub_u32_e32 v79, v42, v78\n\t"                   /* offset - ones */
        "v_cndmask_b32_e32 v79, v79, v78, vcc\n\t"          /* rank */
        "v_cndmask_b32_e32 v88, v48, v52, vcc\n\t"          /* E.x: node to emit */
        "v_cndmask_b32_e32 v80, v49, v53, vcc\n\t"          /* E.y: offset base */
        "v_cndmask_b32_e32 v40, v50, v54, vcc\n\t"          /* E.z: landing record */
        "v_cndmask_b32_e32 v43, v51, v55, vcc\n\t"          /* E.w: its block base */
        "v_cndmask_b32_e32 v84, v56, v58, vcc\n\t"          /* look-ahead base */
        "v_cndmask_b32_e32 v85, v57, v59, vcc\n\t"          /* flags | look-ahead count */
        "v_cndmask_b32_e32 v87, v90, v91, vcc\n\t"          /* look-ahead record */
        "v_add_u32_e32 v42, v80, v79\n\t"                   /* offset in the landing record */
        GBWT_WALK_ISSUE
        "v_and_b32_e32 v76, 63, v44\n\t"                    /* ring slot of the next node */
        "v_cmp_ne_u32_e32 vcc, 0, v88\n\t"
        "v_lshl_add_u32 v76, v76, 8, %[ring]\n\t"
        "v_add_u32_e32 v89, s41, v40\n\t"                   /* node of the landing record */
        "ds_write_b32 v76, v88\n\t"
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t"       /* counts if it is not the ENDMARKER */
        "v_cmp_gt_i32_e32 vcc, 0, v85\n\t"                  /* EDGE_EMIT2 = sign of the flags */
        "v_and_b32_e32 v76, 63, v44\n\t"
        "v_lshl_add_u32 v76, v76, 8, %[ring]\n\t"
        "ds_write_b32 v76, v89\n\t"
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t"
        "v_cmp_ne_u32_e64 s[44:45], 0, v40\n\t"             /* lanes still walking */
        "v_sub_u32_e32 v76, v44, %[flushed]\n\t"
        "v_cmp_lt_u32_e32 vcc, %[urgent], v76\n\t"
        "s_cmp_eq_u64 s[44:45], 0\n\t"
        "s_cbranch_scc1 .Lgbwt_walk_out_%=\n\t"
        "s_cbranch_vccz .Lgbwt_walk_loop_%=\n\t"
        "s_branch .Lgbwt_walk_out_%=\n\t"
        ".Lgbwt_walk_slow_%=:\n\t"
        "s_mov_b32 %[reason], 1\n\t"
        ".Lgbwt_walk_out_%=:\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "v_mov_b32_e32 %[rec], v40\n\t"
        "v_mov_b32_e32 %[offset], v42\n\t"
        "v_mov_b32_e32 %[bb], v43\n\t"
        "v_mov_b32_e32 %[wr], v44\n\t"
        "v_mov_b32_e32 %[hash], v86\n\t"
        : [rec] "+v"(rec), [offset] "+v"(offset), [bb] "+v"(bb), [wr] "+v"(wr), [hash] "+v"(hash), [reason] "=&s"(reason)
        : [desc] "s"(desc), [blocks] "s"(blocks), [ring] "v"(ring_base), [mail] "v"(mail_slot), [flushed] "v"(flushed), [urgent] "i"(RING_URGENT),
          "{s41}"(alphabet_offset)
        : "memory", "vcc", "scc", "s44", "s45",
          "v40", "v42", "v43", "v44", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
          "v60", "v61", "v62", "v63", "v66", "v67", "v68", "v69", "v70", "v71", "v76", "v77", "v78",
          "v79", "v80", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91");
#undef GBWT_WALK_ISSUE
    return reason;
#endif
}

// The look-ahead helper.  All walks of an XCD reach a record at about the same time, so the first one pays an L2
// miss and the others wait on the same fill.  Touching the rank blocks a few records ahead fixes that, but not from
// the walking wave: gfx9 returns a wave's loads in order, so a touch that misses holds back the demand loads issued
// behind it, and the leader would still be paced by the miss.  The touches therefore come from a second wave of the
// workgroup with a vmcnt of its own.  Every iteration a walking lane posts {first block, flags | block count, sequence
// number, record} of the record it will reach a few iterations later (k_link_lookahead) in its LDS mailbox slot; the
// helper polls the slots, and for every slot that changed loads the descriptor and one block of that record (lanes and
// iterations follow one golden-ratio sequence, so together they cover the block array evenly) -- into registers nobody
// reads, never waiting for them.  Leaves when the walking wave raises the done flag.
__device__ __forceinline__ void lookahead_helper(const uint4 *desc, const uint4 *blocks, uint32_t mail_slot, uint32_t done_addr) {
    asm volatile(
        "v_mov_b32_e32 v40, 0\n\t"                          /* last sequence number seen */
        "v_mov_b32_e32 v47, 0\n\t"
        "s_mov_b32 s42, 0x1fffffff\n\t"
        ".Lgbwt_helper_loop_%=:\n\t"
        "ds_read_b128 v[48:51], %[mail]\n\t"
        "ds_read_b32 v52, %[done]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_ne_u32_e32 vcc, v50, v40\n\t"                /* slots with a new target */
        "v_and_b32_e32 v53, s42, v49\n\t"                   /* number of blocks of the target */
        "v_mov_b32_e32 v40, v50\n\t"
        "v_cmp_ne_u32_e64 s[46:47], 0, v53\n\t"
        "v_mul_hi_u32 v46, v50, v53\n\t"                    /* pseudo-random block of it */
        "s_and_b64 vcc, vcc, s[46:47]\n\t"
        "v_add_u32_e32 v46, v46, v48\n\t"
        "v_lshlrev_b32_e32 v58, 2, v51\n\t"                /* descriptor of the target record: 64 bytes each */
        "v_mov_b32_e32 v59, 0\n\t"
        "s_and_saveexec_b64 s[44:45], vcc\n\t"
        "v_lshl_add_u64 v[54:55], v[46:47], 4, %[blocks]\n\t"
        "v_lshl_add_u64 v[58:59], v[58:59], 4, %[desc]\n\t"
        "global_load_dword v56, v[54:55], off offset:12\n\t"
        "global_load_dword v57, v[58:59], off\n\t"
        "s_mov_b64 exec, s[44:45]\n\t"
        "v_readfirstlane_b32 s46, v52\n\t"
        "s_cmp_lg_u32 s46, 0\n\t"
        "s_cbranch_scc1 .Lgbwt_helper_out_%=\n\t"
        "s_sleep 8\n\t"
        "s_branch .Lgbwt_helper_loop_%=\n\t"
        ".Lgbwt_helper_out_%=:\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        :
        : [mail] "v"(mail_slot), [done] "v"(done_addr), [blocks] "s"(blocks), [desc] "s"(desc)
        : "memory", "vcc", "scc", "s42", "s44", "s45", "s46", "s47", "v40", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56",
          "v57", "v58", "v59");
}

// Default walk: one lane per sequence, no cross-lane work.  An iteration of the hot loop is ONE round trip to memory
// (descriptor + rank block travel together; record index and block base of the next record arrived with the edge
// taken), a popcount, and one or two emitted nodes.  Nodes are emitted on arrival: SequenceIter::next
// (src/gbwt.rs:560-567) yields pos.node and then steps; here the start node is pushed before the loop and every
// iteration pushes the node(s) it steps to.  Wave 0 of the workgroup walks; this function is the cold frame around
// walk_hot_loop: the start of the sequences, the generic step for DESC_SLOW records, and moving full ring chunks to
// the pool.  Wave 1 is the look-ahead helper.
__global__ void __launch_bounds__(2 * WAVE) k_walk_blocks(DeviceIndex ix, WalkArgs a) {
    __shared__ uint32_t ring_lds[RING * WAVE];
    __shared__ uint4 mailbox[WAVE];
    __shared__ uint32_t mail_done;
    const uint32_t lane = threadIdx.x % WAVE;
    const bool helper = __builtin_amdgcn_readfirstlane(threadIdx.x) >= WAVE;
    if (!helper) {
        mailbox[lane] = make_uint4(0, 0, 0, 0);
        if (lane == 0) mail_done = 0;
    }
    __syncthreads();
    const uint32_t mail_slot = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&mailbox[lane]));   // LDS byte addresses
    if (helper) {
        if (lane >= a.helper_lanes) return;
        lookahead_helper(ix.desc, ix.blocks, mail_slot, static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&mail_done)));
        return;
    }
    RingSink sink(ring_lds, lane);
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(a.paths_per_wave) + lane;
    const bool owner = lane < a.paths_per_wave && k < a.n;
    uint32_t rec = 0, offset = 0, bb = BLOCK_NONE;   // position of the walk (record index; 0 = parked) + block base of the record
    if (owner) {
        const uint64_t id = a.seq_ids[k];
        if (id < ix.n_endmarker) {  // GBWT::start, src/gbwt.rs:213-219
            const uint2 e = ix.endmarker[id];
            if (e.x != 0) {
                sink.push(e.x, true);
                offset = e.y;
                if (!arrive(ix, e.x, e.y, rec, bb)) { rec = 0; bb = BLOCK_NONE; }
            }
        }
    }
    const uint32_t ring_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(sink.stage));   // LDS byte address (low half of the flat one)
    uint32_t hash = (lane + WAVE * blockIdx.x) * 0x9E3779B1u;   // lanes and iterations walk one golden-ratio sequence: consecutive values spread evenly over the target's blocks
    while (__ballot(rec != 0) != 0) {
        const uint32_t slow_exit = walk_hot_loop(ix.desc, ix.blocks, ix.alphabet_offset, ring_base, mail_slot, sink.flushed, rec, offset, bb, sink.wr, hash);
        if (slow_exit) {
            // generic step for the lanes on a DESC_SLOW record (outdegree > 2, streams outside the descriptor's limits,
            // edges k_link_desc could not vouch for): Record::lf on the record bytes, then the arrival tests
            const bool slow = rec != 0 && static_cast<int32_t>(ix.desc[4 * static_cast<uint64_t>(rec) + 2].x) < 0;
            if (slow) generic_step(ix, sink, rec, offset, bb);
        }
        while (sink.needs_flush()) sink.flush16(a);
    }
    if (lane == 0) *const_cast<volatile uint32_t *>(&mail_done) = 1;
    if (owner) {
        a.lengths[k] = sink.finish(a);
        a.head[k] = sink.head;
    }
}

// ---- two-step walk --------------------------------------------------------------------------------------
// Same frame as k_walk_blocks; an iteration of the hot loop composes two LF steps (k_link_desc2, k_fill_cblocks):
//     a = bit `offset` of bits1;  rank_a = equal values of v before it;        j = base_a + rank_a  (offset in w_a)
//     b = bit `offset` of bits2;  rank_b = equal values of w_a before j = R_a + (a-paths of this block before `offset`
//                                          whose value in w_a is 1), or j minus that
//     leaf (a, b): rec = its landing record, offset = its base + rank_b
//     emit: node of edge a, node of w_a if that step was fused, node of the leaf, node of rec if that step was fused
__device__ __forceinline__ uint32_t walk2_hot_loop(const uint4 *desc2, const uint4 *cblocks, uint32_t alphabet_offset, uint32_t ring_base,
                                                   uint32_t mail_slot, uint32_t flushed, bool narrow, uint32_t &rec, uint32_t &offset, uint32_t &bb,
                                                   uint32_t &wr, uint32_t &seq) {
#ifdef GBWT_HIP_CXX_LOOP
    // plain C++ statement of the loop (no pipelining)
    __attribute__((address_space(3))) uint32_t *ring = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)ring_base;
    __attribute__((address_space(3))) uint32_t *mail = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)mail_slot;
    for (;;) {
        const uint4 *d = desc2 + 8 * static_cast<uint64_t>(rec);
        const uint4 F0 = d[0], F1 = d[1], L00 = d[2], L01 = d[3], L10 = d[4], L11 = d[5], look = d[6];
        const uint64_t idx = bb == BLOCK_NONE ? 0u : bb + (offset >> RANK_BLOCK_SHIFT);
        const uint4 K0 = cblocks[2 * idx], K1 = cblocks[2 * idx + 1];
        if (__ballot((F1.x & DESC2_SLOW) != 0) != 0) return 1;
        const uint64_t bits1 = (static_cast<uint64_t>(K0.y) << 32) | K0.x, bits2 = (static_cast<uint64_t>(K0.w) << 32) | K0.z;
        const uint32_t bit = offset & 63u;
        const uint64_t below = (uint64_t(1) << bit) - 1;
        const uint32_t a = static_cast<uint32_t>(bits1 >> bit) & 1u;
        const uint64_t m = a ? bits1 : ~bits1;
        const uint32_t p = __popcll(m & below);
        const uint32_t rank_a = a ? K1.x + p : (offset - bit) - K1.x + p;
        const uint32_t j = (a ? F0.w : F0.y) + rank_a;
        const uint32_t b = static_cast<uint32_t>(bits2 >> bit) & 1u;
        const uint32_t ones_w = (a ? K1.z : K1.y) + __popcll(m & bits2 & below);
        const uint32_t rank_b = b ? ones_w : j - ones_w;
        const uint4 leaf = a ? (b ? L11 : L10) : (b ? L01 : L00);
        const uint32_t n1 = a ? F0.z : F0.x, wword = a ? F1.y : F1.x;
        rec = leaf.z & REC_MASK; offset = leaf.y + rank_b; bb = leaf.w;
        ring[(wr & (RING2 - 1)) * WAVE] = n1;
        wr += n1 != 0 ? 1u : 0u;
        ring[(wr & (RING2 - 1)) * WAVE] = (wword & REC_MASK) + alphabet_offset;
        wr += (wword & LEAF_EMIT2) ? 1u : 0u;
        ring[(wr & (RING2 - 1)) * WAVE] = leaf.x;
        wr += leaf.x != 0 ? 1u : 0u;
        ring[(wr & (RING2 - 1)) * WAVE] = rec + alphabet_offset;
        wr += (leaf.z & LEAF_EMIT2) ? 1u : 0u;
        seq += 0x9E3779B1u;
        mail[0] = look.x; mail[1] = look.y; mail[2] = look.z; mail[3] = seq;
        if (__ballot(rec != 0) == 0 || __ballot(wr - flushed > RING2_URGENT) != 0) return 0;
    }
#else
    // The same loop in gfx950 assembly (see walk_hot_loop for the conventions: all lanes run everything, parked lanes sit
    // on record 0, loads of the next position go out as early as possible, exits leave nothing in flight; a VALU write
    // of VCC / an SGPR is kept two instructions away from the VALU that reads it).  Registers v40-v125, s41, s44-s47.
    uint32_t reason;
#define GBWT_WALK2_ISSUE_WIDE                                                                                  \
    "v_cmp_ne_u32_e32 vcc, -1, v43\n\t"                   /* bb != BLOCK_NONE */                          \
    "v_lshrrev_b32_e32 v70, 6, v42\n\t"                                                                   \
    "v_add_u32_e32 v70, v70, v43\n\t"                                                                     \
    "v_lshlrev_b64 v[88:89], 7, v[40:41]\n\t"             /* two-step descriptors are 128 bytes */        \
    "v_cndmask_b32_e32 v70, 0, v70, vcc\n\t"              /* block bb + offset / 64, or the zero block */ \
    "v_lshl_add_u64 v[88:89], v[88:89], 0, %[desc2]\n\t"                                                  \
    "v_lshlrev_b64 v[90:91], 5, v[70:71]\n\t"             /* two-step blocks are 32 bytes */              \
    "v_lshl_add_u64 v[90:91], v[90:91], 0, %[cblocks]\n\t"                                                \
    "s_mov_b64 exec, s[44:45]\n\t"                        /* only lanes that were walking before this step */ \
    "global_load_dwordx4 v[80:83], v[90:91], off\n\t"             /* K0: bits1, bits2 */                  \
    "global_load_dwordx3 v[84:86], v[90:91], off offset:16\n\t"   /* K1: ones1, R0, R1 */                 \
    "global_load_dwordx4 v[48:51], v[88:89], off\n\t"             /* F0 */                                \
    "global_load_dwordx2 v[52:53], v[88:89], off offset:16\n\t"   /* F1 */                                \
    "global_load_dwordx4 v[56:59], v[88:89], off offset:32\n\t"   /* leaf (0, 0) */                       \
    "global_load_dwordx4 v[60:63], v[88:89], off offset:48\n\t"   /* leaf (0, 1) */                       \
    "global_load_dwordx4 v[64:67], v[88:89], off offset:64\n\t"   /* leaf (1, 0) */                       \
    "global_load_dwordx4 v[72:75], v[88:89], off offset:80\n\t"   /* leaf (1, 1) */                       \
    "global_load_dwordx3 v[76:78], v[88:89], off offset:96\n\t"   /* look-ahead target */                 \
    "s_mov_b64 exec, -1\n\t"
#define GBWT_WALK2_ISSUE_NARROW                                                                           \
    "v_cmp_ne_u32_e32 vcc, -1, v43\n\t"                   /* bb != BLOCK_NONE */                          \
    "v_lshrrev_b32_e32 v70, 6, v42\n\t"                                                                   \
    "v_add_u32_e32 v70, v70, v43\n\t"                                                                     \
    "v_lshlrev_b32_e32 v88, 7, v40\n\t"                   /* two-step descriptors are 128 bytes */        \
    "v_cndmask_b32_e32 v70, 0, v70, vcc\n\t"              /* block bb + offset / 64, or the zero block */ \
    "v_lshlrev_b32_e32 v90, 5, v70\n\t"                   /* two-step blocks are 32 bytes */              \
    "s_mov_b64 exec, s[44:45]\n\t"                        /* only lanes that were walking before this step */ \
    "global_load_dwordx4 v[80:83], v90, %[cblocks]\n\t"             /* K0: bits1, bits2 */                \
    "global_load_dwordx3 v[84:86], v90, %[cblocks] offset:16\n\t"   /* K1: ones1, R0, R1 */               \
    "global_load_dwordx4 v[48:51], v88, %[desc2]\n\t"               /* F0 */                              \
    "global_load_dwordx2 v[52:53], v88, %[desc2] offset:16\n\t"     /* F1 */                              \
    "global_load_dwordx4 v[56:59], v88, %[desc2] offset:32\n\t"     /* leaf (0, 0) */                     \
    "global_load_dwordx4 v[60:63], v88, %[desc2] offset:48\n\t"     /* leaf (0, 1) */                     \
    "global_load_dwordx4 v[64:67], v88, %[desc2] offset:64\n\t"     /* leaf (1, 0) */                     \
    "global_load_dwordx4 v[72:75], v88, %[desc2] offset:80\n\t"     /* leaf (1, 1) */                     \
    "global_load_dwordx3 v[76:78], v88, %[desc2] offset:96\n\t"     /* look-ahead target */               \
    "s_mov_b64 exec, -1\n\t"
#define GBWT_WALK2_LOOP(ISSUE)                                                                             \
    asm volatile( \
        "v_mov_b32_e32 v40, %[rec]\n\t" \
        "v_mov_b32_e32 v41, 0\n\t" \
        "v_mov_b32_e32 v42, %[offset]\n\t" \
        "v_mov_b32_e32 v43, %[bb]\n\t" \
        "v_mov_b32_e32 v44, %[wr]\n\t" \
        "v_mov_b32_e32 v79, %[seq]\n\t" \
        "v_mov_b32_e32 v71, 0\n\t" \
        "s_mov_b32 %[reason], 0\n\t" \
        "s_mov_b64 s[44:45], -1\n\t" \
        ISSUE \
        ".Lgbwt_walk2_loop_%=:\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "v_lshlrev_b32_e32 v92, 1, v52\n\t"                 /* DESC2_SLOW (bit 30 of F1.x) -> sign */ \
        "v_lshrrev_b64 v[94:95], v42, v[80:81]\n\t"         /* bits1 >> bit */ \
        "v_cmp_gt_i32_e32 vcc, 0, v92\n\t" \
        "v_lshlrev_b64 v[96:97], v42, -1\n\t"               /* bits at and above `bit` */ \
        "v_and_b32_e32 v94, 1, v94\n\t"                     /* a */ \
        "s_cbranch_vccnz .Lgbwt_walk2_slow_%=\n\t" \
        "v_add_u32_e32 v98, -1, v94\n\t"                    /* a ? 0 : ~0 */ \
        "v_cmp_eq_u32_e32 vcc, 1, v94\n\t"                  /* vcc = a */ \
        "v_and_b32_e32 v99, 0xffffffc0, v42\n\t"            /* offset - bit */ \
        "v_xor_b32_e32 v100, v80, v98\n\t"                  /* m = a ? bits1 : ~bits1 */ \
        "v_xor_b32_e32 v101, v81, v98\n\t" \
        "v_bfi_b32 v100, v96, 0, v100\n\t"                  /* m below `bit` */ \
        "v_bfi_b32 v101, v97, 0, v101\n\t" \
        "v_sub_u32_e32 v99, v99, v84\n\t"                   /* (offset - bit) - ones1 */ \
        "v_bcnt_u32_b32 v102, v100, 0\n\t" \
        "v_cndmask_b32_e32 v99, v99, v84, vcc\n\t"          /* a ? ones1 : that */ \
        "v_bcnt_u32_b32 v102, v101, v102\n\t"               /* p */ \
        "v_cndmask_b32_e32 v103, v49, v51, vcc\n\t"         /* offset base of edge a */ \
        "v_add_u32_e32 v99, v99, v102\n\t"                  /* rank_a */ \
        "v_cndmask_b32_e32 v104, v48, v50, vcc\n\t"         /* node of edge a */ \
        "v_add_u32_e32 v103, v103, v99\n\t"                 /* j: offset in w_a */ \
        "v_cndmask_b32_e32 v105, v52, v53, vcc\n\t"         /* w_a | flags */ \
        "v_cndmask_b32_e32 v106, v85, v86, vcc\n\t"         /* R_a */ \
        "v_lshrrev_b64 v[108:109], v42, v[82:83]\n\t"       /* bits2 >> bit */ \
        "v_and_b32_e32 v100, v100, v82\n\t"                 /* a-paths below `bit` with value 1 in w_a */ \
        "v_and_b32_e32 v101, v101, v83\n\t" \
        "v_and_b32_e32 v108, 1, v108\n\t"                   /* b */ \
        "v_bcnt_u32_b32 v106, v100, v106\n\t" \
        "v_cmp_eq_u32_e64 s[46:47], 1, v108\n\t"            /* s[46:47] = b */ \
        "v_bcnt_u32_b32 v106, v101, v106\n\t"               /* ones of w_a before j */ \
        "v_sub_u32_e32 v107, v103, v106\n\t"                /* j - ones */ \
        "v_and_b32_e32 v110, 0x3fffffff, v105\n\t"          /* w_a */ \
        "v_cndmask_b32_e64 v107, v107, v106, s[46:47]\n\t"  /* rank_b */ \
        "v_cndmask_b32_e64 v112, v56, v60, s[46:47]\n\t"    /* leaf (0, b) */ \
        "v_cndmask_b32_e64 v113, v57, v61, s[46:47]\n\t" \
        "v_cndmask_b32_e64 v114, v58, v62, s[46:47]\n\t" \
        "v_cndmask_b32_e64 v115, v59, v63, s[46:47]\n\t" \
        "v_cndmask_b32_e64 v116, v64, v72, s[46:47]\n\t"    /* leaf (1, b) */ \
        "v_cndmask_b32_e64 v117, v65, v73, s[46:47]\n\t" \
        "v_cndmask_b32_e64 v118, v66, v74, s[46:47]\n\t" \
        "v_cndmask_b32_e64 v119, v67, v75, s[46:47]\n\t" \
        "v_cndmask_b32_e32 v112, v112, v116, vcc\n\t"       /* leaf (a, b): node to emit */ \
        "v_cndmask_b32_e32 v113, v113, v117, vcc\n\t"       /* offset base */ \
        "v_cndmask_b32_e32 v114, v114, v118, vcc\n\t"       /* landing record | flags */ \
        "v_cndmask_b32_e32 v43, v115, v119, vcc\n\t"        /* its block base */ \
        "v_add_u32_e32 v42, v113, v107\n\t"                 /* the new offset */ \
        "v_and_b32_e32 v40, 0x3fffffff, v114\n\t"           /* the new record */ \
        "v_add_u32_e32 v79, 0x9e3779b1, v79\n\t"            /* new sequence number for the look-ahead target of the record just left */ \
        "v_and_b32_e32 v92, 0x7f, v44\n\t"                  /* ring slot of the next node */ \
        "v_cmp_ne_u32_e32 vcc, 0, v104\n\t" \
        "v_lshl_add_u32 v92, v92, 8, %[ring]\n\t" \
        "ds_write_b32 v92, v104\n\t"                        /* node of edge a */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t"       /* counts if it is not the ENDMARKER */ \
        "v_cmp_gt_i32_e32 vcc, 0, v105\n\t"                 /* first step fused? */ \
        "v_and_b32_e32 v92, 0x7f, v44\n\t" \
        "v_add_u32_e32 v110, s41, v110\n\t"                 /* node of w_a */ \
        "v_lshl_add_u32 v92, v92, 8, %[ring]\n\t" \
        "ds_write_b32 v92, v110\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_ne_u32_e32 vcc, 0, v112\n\t" \
        "v_and_b32_e32 v92, 0x7f, v44\n\t" \
        "v_add_u32_e32 v111, s41, v40\n\t"                  /* node of the landing record */ \
        "v_lshl_add_u32 v92, v92, 8, %[ring]\n\t" \
        "ds_write_b32 v92, v112\n\t"                        /* node of the leaf */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_gt_i32_e32 vcc, 0, v114\n\t"                 /* second step fused? */ \
        "v_and_b32_e32 v92, 0x7f, v44\n\t" \
        "ds_write_b128 %[mail], v[76:79]\n\t" \
        "v_lshl_add_u32 v92, v92, 8, %[ring]\n\t" \
        "ds_write_b32 v92, v111\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        ISSUE \
        "v_cmp_ne_u32_e64 s[44:45], 0, v40\n\t"             /* lanes still walking */ \
        "v_cmp_lt_u32_e32 vcc, %[limit], v44\n\t"             /* more than RING2_URGENT nodes waiting in a ring */ \
        "s_cmp_eq_u64 s[44:45], 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_walk2_out_%=\n\t" \
        "s_cbranch_vccz .Lgbwt_walk2_loop_%=\n\t" \
        "s_branch .Lgbwt_walk2_out_%=\n\t" \
        ".Lgbwt_walk2_slow_%=:\n\t" \
        "s_mov_b32 %[reason], 1\n\t" \
        ".Lgbwt_walk2_out_%=:\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
        "v_mov_b32_e32 %[rec], v40\n\t" \
        "v_mov_b32_e32 %[offset], v42\n\t" \
        "v_mov_b32_e32 %[bb], v43\n\t" \
        "v_mov_b32_e32 %[wr], v44\n\t" \
        "v_mov_b32_e32 %[seq], v79\n\t" \
        : [rec] "+v"(rec), [offset] "+v"(offset), [bb] "+v"(bb), [wr] "+v"(wr), [seq] "+v"(seq), [reason] "=&s"(reason) \
        : [desc2] "s"(desc2), [cblocks] "s"(cblocks), [ring] "v"(ring_base), [mail] "v"(mail_slot), [limit] "v"(limit), \
          "{s41}"(alphabet_offset) \
        : "memory", "vcc", "scc", "s44", "s45", "s46", "s47", \
          "v40", "v41", "v42", "v43", "v44", "v48", "v49", "v50", "v51", "v52", "v53", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", \
          "v64", "v65", "v66", "v67", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", \
          "v88", "v89", "v90", "v91", "v92", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", \
          "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
    const uint32_t limit = flushed + RING2_URGENT;
    if (narrow) { GBWT_WALK2_LOOP(GBWT_WALK2_ISSUE_NARROW) } else { GBWT_WALK2_LOOP(GBWT_WALK2_ISSUE_WIDE) }
#undef GBWT_WALK2_LOOP
#undef GBWT_WALK2_ISSUE_NARROW
#undef GBWT_WALK2_ISSUE_WIDE
    return reason;
#endif
}

// Look-ahead helper of the two-step walk: mailbox slot = {record, first block, number of blocks, sequence number} of
// the record the walk reaches a few iterations later; touches its descriptor (128 bytes = two sectors) and one of its
// two-step blocks.  Fire and forget, as lookahead_helper.
__device__ __forceinline__ void lookahead_helper2(const uint4 *desc2, const uint4 *cblocks, uint32_t mail_slot, uint32_t done_addr, uint32_t spread) {
    asm volatile(
        "v_mov_b32_e32 v40, 0\n\t"                          /* last sequence number seen */
        "v_mov_b32_e32 v47, 0\n\t"
        "v_mov_b32_e32 v59, 0\n\t"
        ".Lgbwt_helper2_loop_%=:\n\t"
        "ds_read_b128 v[48:51], %[mail]\n\t"
        "ds_read_b32 v52, %[done]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_ne_u32_e32 vcc, v51, v40\n\t"                /* slots with a new target ... */
        "v_mov_b32_e32 v58, v48\n\t"
        "v_mov_b32_e32 v40, v51\n\t"
        "v_cmp_ne_u32_e64 s[46:47], 0, v48\n\t"             /* ... that is a record */
        "v_mov_b32_e32 v46, %[spread]\n\t"                  /* helper lane l takes the block (l + 1/2) / 64 of the way through */
        "v_mul_hi_u32 v46, v46, v50\n\t"
        "s_and_b64 vcc, vcc, s[46:47]\n\t"
        "v_add_u32_e32 v46, v46, v49\n\t"
        "s_and_saveexec_b64 s[44:45], vcc\n\t"
        "v_lshlrev_b64 v[54:55], 5, v[46:47]\n\t"           /* two-step blocks are 32 bytes */
        "v_lshlrev_b64 v[60:61], 7, v[58:59]\n\t"           /* two-step descriptors are 128 bytes */
        "v_lshl_add_u64 v[54:55], v[54:55], 0, %[cblocks]\n\t"
        "v_lshl_add_u64 v[60:61], v[60:61], 0, %[desc2]\n\t"
        "global_load_dword v56, v[54:55], off\n\t"
        "global_load_dword v57, v[60:61], off\n\t"
        "global_load_dword v53, v[60:61], off offset:64\n\t"
        "s_mov_b64 exec, s[44:45]\n\t"
        "v_readfirstlane_b32 s46, v52\n\t"
        "s_cmp_lg_u32 s46, 0\n\t"
        "s_cbranch_scc1 .Lgbwt_helper2_out_%=\n\t"
        "s_sleep 8\n\t"
        "s_branch .Lgbwt_helper2_loop_%=\n\t"
        ".Lgbwt_helper2_out_%=:\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        :
        : [mail] "v"(mail_slot), [done] "v"(done_addr), [spread] "v"(spread), [cblocks] "s"(cblocks), [desc2] "s"(desc2)
        : "memory", "vcc", "scc", "s44", "s45", "s46", "s47", "v40", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56",
          "v57", "v58", "v59", "v60", "v61");
}

__global__ void __launch_bounds__(2 * WAVE) k_walk_two(DeviceIndex ix, WalkArgs a) {
    __shared__ uint32_t ring_lds[RING2 * WAVE];
    __shared__ uint4 mailbox[WAVE];
    __shared__ uint32_t mail_done;
    const uint32_t lane = threadIdx.x % WAVE;
    const bool helper = __builtin_amdgcn_readfirstlane(threadIdx.x) >= WAVE;
    if (!helper) {
        mailbox[lane] = make_uint4(0, 0, 0, 0);
        if (lane == 0) mail_done = 0;
    }
    __syncthreads();
    if (helper) {
        // the 64 helper lanes share the slots of the owners: lane l serves slot l mod owners, and the lanes of one slot
        // spread over the target's blocks (lane / 64 of the way round)
        const uint32_t owners = a.paths_per_wave ? a.paths_per_wave : WAVE;
        if (lane >= a.helper_lanes) return;
        lookahead_helper2(ix.desc2, ix.cblocks, static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&mailbox[lane % owners])),
                          static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&mail_done)), (lane << 26) | (1u << 25));
        return;
    }
    const uint32_t mail_slot = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&mailbox[lane]));   // LDS byte address
    RingSinkT<RING2> sink(ring_lds, lane);
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(a.paths_per_wave) + lane;
    const bool owner = lane < a.paths_per_wave && k < a.n;
    uint32_t rec = 0, offset = 0, bb = BLOCK_NONE;   // position of the walk (record index; 0 = parked) + block base of the record
    if (owner) {
        const uint64_t id = a.seq_ids[k];
        if (id < ix.n_endmarker) {  // GBWT::start, src/gbwt.rs:213-219
            const uint2 e = ix.endmarker[id];
            if (e.x != 0) {
                sink.push(e.x, true);
                offset = e.y;
                if (!arrive(ix, e.x, e.y, rec, bb)) { rec = 0; bb = BLOCK_NONE; }
            }
        }
    }
    const uint32_t ring_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(sink.stage));
    uint32_t seq = (lane + WAVE * blockIdx.x) * 0x9E3779B1u;
    // SGPR base + 32-bit byte offsets while both arrays are below 4 GiB, 64-bit addresses otherwise
    const bool narrow = !a.wide_addresses && ix.n_records * 128 <= 0xFFFFFFFFull && ix.n_blocks * 32 <= 0xFFFFFFFFull;
    while (__ballot(rec != 0) != 0) {
        const uint32_t slow_exit = walk2_hot_loop(ix.desc2, ix.cblocks, ix.alphabet_offset, ring_base, mail_slot, sink.flushed, narrow, rec, offset, bb, sink.wr, seq);
        if (slow_exit) {
            const bool slow = rec != 0 && (ix.desc2[8 * static_cast<uint64_t>(rec) + 1].x & DESC2_SLOW) != 0;
            if (slow) generic_step(ix, sink, rec, offset, bb);
        }
        while (sink.needs_flush()) sink.flush16(a);
    }
    if (lane == 0) *const_cast<volatile uint32_t *>(&mail_done) = 1;
    if (owner) {
        a.lengths[k] = sink.finish(a);
        a.head[k] = sink.head;
    }
}

// Wave-cooperative walk (WALK_COOP): lanes 0..P-1 of each wave own one sequence each; long class 1 / 2 records are
// decoded one distinct record at a time by the whole wave (coop_device.hpp), so sequences that sit in the same record
// share one decode.  Kept as an alternative to the default walk: it needs no rank blocks.
template <bool PACK16>
__global__ void __launch_bounds__(WAVE) k_walk_coop(DeviceIndex ix, WalkArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(a.paths_per_wave) + lane;
    const bool owner = lane < a.paths_per_wave && k < a.n;
    uint32_t node = 0, offset = 0;
    bool active = false;
    if (owner) {
        const uint64_t id = a.seq_ids[k];
        if (id < ix.n_endmarker) {
            uint2 e = ix.endmarker[id];
            node = e.x; offset = e.y;
            active = node != 0;
        }
    }
    __shared__ uint32_t sink_lds[SINK_STAGE * WAVE];
    PathSink sink(sink_lds, lane);
    while (__ballot(active) != 0) {
        if (active) sink.push(a, node);
        uint64_t start = 0;
        uint32_t bytes = 0, meta = 0, n0 = 0, o0 = 0, n1 = 0, o1 = 0;
        bool has_record = false, ok = false;
        uint32_t next_node = 0, next_offset = 0;
        if (active && node >= ix.first_node && node - ix.alphabet_offset < ix.n_records) {
            const uint64_t rec = node - ix.alphabet_offset;
            const uint4 A = ix.desc_raw[4 * rec], B = ix.desc_raw[4 * rec + 1];
            if (B.y == DESC_UNARY) {
                ok = offset < B.w && A.x != 0;
                next_node = A.x; next_offset = A.y + offset;
            } else if (B.y != 0) {
                start = desc_start(B.x, B.z);
                bytes = B.y; meta = B.z;
                n0 = A.x; o0 = A.y; n1 = A.z; o1 = A.w;
                has_record = true;
            }
        }
        const bool big = has_record && bytes > a.small_record && desc_class(meta) != 0;
        bool serial = has_record && !big;
        uint64_t todo = __ballot(big);
        while (todo != 0) {
            const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(todo));
            const uint64_t gs = read_lane64(start, leader);
            const uint32_t gbytes = read_lane(bytes, leader), gmeta = read_lane(meta, leader);
            const bool member = big && start == gs;
            const uint32_t body_off = desc_body_offset(gmeta);
            const int status = coop_runs_lf<PACK16>(ix.data + gs + body_off, gbytes - body_off, desc_class(gmeta) == 2, member, offset,
                                                    n0, o0, n1, o1, ok, next_node, next_offset);
            if (status != COOP_DONE && member) serial = true;
            todo &= ~__ballot(member);
        }
        if (serial) {
            const uint2 r = serial_record_lf(ix.data, start, bytes, offset);
            next_node = r.x; next_offset = r.y; ok = r.x != 0;
        }
        if (active) {
            active = ok;
            node = next_node; offset = next_offset;
        }
    }
    if (owner) {
        a.lengths[k] = sink.finish(a);
        a.head[k] = sink.head;
    }
}

// One wave per path: follow the block chain and copy it to its CSR row.
__global__ void __launch_bounds__(256) k_compact(WalkArgs a, const uint64_t *offsets, uint32_t *nodes) {
    const uint64_t path = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (path >= a.n) return;
    uint64_t remaining = offsets[path + 1] - offsets[path];
    uint32_t *dst = nodes + offsets[path];
    // The chain is a pointer chase; the link of the NEXT block is fetched while the current block is copied, and a
    // full block moves as one 16-byte load + store per lane (rows start at arbitrary offsets, so the stores are only
    // 4-byte aligned: they are split when the row start is not 16-byte aligned).
    uint32_t b = a.head[path];
    uint32_t nb = (remaining > 0 && b != POOL_NONE) ? a.next[b] : POOL_NONE;
    const bool aligned = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
    while (remaining > 0 && b != POOL_NONE) {
        const uint32_t cnt = remaining < POOL_BLOCK_NODES ? static_cast<uint32_t>(remaining) : POOL_BLOCK_NODES;
        const uint32_t *src = a.pool + static_cast<uint64_t>(b) * POOL_BLOCK_NODES;
        const uint32_t nnb = (remaining > cnt && nb != POOL_NONE) ? a.next[nb] : POOL_NONE;
        if (cnt == POOL_BLOCK_NODES) {
            const uint4 v = reinterpret_cast<const uint4 *>(src)[lane];
            if (aligned) reinterpret_cast<uint4 *>(dst)[lane] = v;
            else { dst[4 * lane] = v.x; dst[4 * lane + 1] = v.y; dst[4 * lane + 2] = v.z; dst[4 * lane + 3] = v.w; }
        } else {
            for (uint32_t idx = lane; idx < cnt; idx += WAVE) dst[idx] = src[idx];
        }
        dst += cnt;
        remaining -= cnt;
        b = nb; nb = nnb;
    }
}

// One wave per CSR row: sum of the node ids (checking hook for device-resident results).
__global__ void __launch_bounds__(256) k_path_sums(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, uint64_t *sums) {
    const uint64_t path = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (path >= n) return;
    uint64_t acc = 0;
    for (uint64_t k = offsets[path] + lane; k < offsets[path + 1]; k += WAVE) acc += nodes[k];
    for (int d = WAVE / 2; d > 0; d >>= 1) acc += __shfl_down(acc, d, WAVE);
    if (lane == 0) sums[path] = acc;
}

// ---------------------------------------------------------------------------------------------
// Navigation and search: one lane per query.

__global__ void __launch_bounds__(256) k_start(DeviceIndex ix, const uint64_t *ids, uint64_t n, gbwt_hip_pos *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    gbwt_hip_pos p{0, 0};
    uint8_t ok = 0;
    uint64_t id = ids[k];
    if (id < ix.n_endmarker) {
        uint2 e = ix.endmarker[id];
        if (e.x != 0) { p.node = e.x; p.offset = e.y; ok = 1; }
    }
    out[k] = p; valid[k] = ok;
}

// ---- search on descriptors + rank blocks --------------------------------------------------------------
// For class 1 / 2 records everything Record::follow / bd_follow compute (src/bwt.rs:595-656) is a difference of
// "how many of the first p positions take edge r", which the rank blocks answer in O(1): two block lookups
// replace the reference's scan of all runs up to range.end.  Other records use the generic scan of lf_device.hpp.

struct RawDesc { uint4 A, B, C, D; };

__device__ __forceinline__ bool load_raw_desc(const DeviceIndex &ix, uint64_t node, RawDesc &d, uint64_t &rec) {
    if (node < ix.first_node) return false;
    rec = node - ix.alphabet_offset;
    if (rec >= ix.n_records) return false;
    d.A = ix.desc_raw[4 * rec]; d.B = ix.desc_raw[4 * rec + 1]; d.C = ix.desc_raw[4 * rec + 2]; d.D = ix.desc_raw[4 * rec + 3];
    return d.B.y != 0;   // empty record / sigma == 0 -> None
}

// GBWT::forward (src/gbwt.rs:222-229) for independent positions: one block lookup on class 1 / 2 records, one LF-table
// lookup on class 0 records that have a table, the generic scan otherwise.
__global__ void __launch_bounds__(256) k_forward(DeviceIndex ix, const gbwt_hip_pos *in, uint64_t n, gbwt_hip_pos *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const gbwt_hip_pos p = in[k];
    gbwt_hip_pos r{0, 0};
    uint8_t ok = 0;
    RawDesc d;
    uint64_t rec;
    if (load_raw_desc(ix, p.node, d, rec)) {
        const uint32_t cls = desc_class(d.B.z);
        if (cls != 0) {
            if (p.offset < d.B.w) {
                const uint32_t i = static_cast<uint32_t>(p.offset);
                uint32_t value = 0, rank = i;
                if (cls == 2) {
                    const uint4 K = ix.blocks[ix.block_base[rec] + (i >> RANK_BLOCK_SHIFT)];
                    const uint64_t bits = (static_cast<uint64_t>(K.y) << 32) | K.x;
                    value = static_cast<uint32_t>(bits >> (i & 63u)) & 1u;
                    const uint32_t ones = K.z + __popcll(bits & ((uint64_t(1) << (i & 63u)) - 1));
                    rank = value ? ones : i - ones;
                }
                r.node = value ? d.A.z : d.A.x;
                r.offset = static_cast<uint64_t>(value ? d.A.w : d.A.y) + rank;
                ok = r.node != 0;
            }
        } else if (d.C.w == 1u) {
            if (p.offset < d.C.y) {
                const uint4 e = ix.tables[static_cast<uint64_t>(d.C.z) + p.offset];
                r.node = e.x; r.offset = e.y;
                ok = e.x != 0;
            }
        } else ok = gbwt_forward(ix, p.node, p.offset, r.node, r.offset) ? 1 : 0;
    }
    if (!ok) { r.node = 0; r.offset = 0; }
    out[k] = r; valid[k] = ok;
}

// value-0 positions among the first p positions (p <= Record::len) of a class 1 / 2 record
__device__ __forceinline__ uint32_t count0_before(const DeviceIndex &ix, const RawDesc &d, uint64_t rec, uint32_t p) {
    if (desc_class(d.B.z) == 1) return p;
    if (p >= d.B.w) return d.C.x;
    const uint4 K = ix.blocks[ix.block_base[rec] + (p >> RANK_BLOCK_SHIFT)];
    const uint64_t bits = (static_cast<uint64_t>(K.y) << 32) | K.x;
    return p - (K.z + __popcll(bits & ((uint64_t(1) << (p & 63u)) - 1)));
}

// Record::follow / bd_follow on a class 1 / 2 record.
template <bool BD>
__device__ __forceinline__ bool block_follow(const DeviceIndex &ix, const RawDesc &d, uint64_t rec, uint64_t start, uint64_t end, uint64_t dest,
                                               uint64_t &rstart, uint64_t &rend, uint64_t &count) {
    if (start >= end || dest == 0) return false;
    const bool two = desc_class(d.B.z) == 2;
    uint32_t rank;                                   // Record::edge_to
    if (d.A.x == dest) rank = 0;
    else if (two && d.A.z == dest) rank = 1;
    else return false;
    const uint32_t len = d.B.w;
    const uint32_t ps = start < len ? static_cast<uint32_t>(start) : len, pe = end < len ? static_cast<uint32_t>(end) : len;
    const uint32_t z0 = count0_before(ix, d, rec, ps), z1 = count0_before(ix, d, rec, pe);
    const uint32_t before_s = rank ? ps - z0 : z0, before_e = rank ? pe - z1 : z1;
    const uint64_t base = rank ? d.A.w : d.A.y;
    rstart = base + before_s; rend = base + before_e;
    if (rstart >= rend) return false;
    if (BD) {  // positions of [start, end) whose successor s has flip(s) < flip(dest)  (src/bwt.rs:646-648)
        const uint64_t reverse = dest ^ 1;
        uint64_t c = 0;
        if ((static_cast<uint64_t>(d.A.x) ^ 1) < reverse) c += z1 - z0;
        if (two && (static_cast<uint64_t>(d.A.z) ^ 1) < reverse) c += (pe - z1) - (ps - z0);
        count = c;
    }
    return true;
}

// GBWT::find, src/gbwt.rs:269-281: Record::len was computed when the descriptors were built
__device__ __forceinline__ bool dev_find(const DeviceIndex &ix, uint64_t node, gbwt_hip_state &st) {
    RawDesc d;
    uint64_t rec;
    if (!load_raw_desc(ix, node, d, rec)) return false;
    st.node = node; st.start = 0; st.end = d.C.y;
    return true;
}

template <bool BD>
__device__ __forceinline__ bool dev_follow(const DeviceIndex &ix, uint64_t from, uint64_t start, uint64_t end, uint64_t dest,
                                           uint64_t &rs, uint64_t &re, uint64_t &count) {
    RawDesc d;
    uint64_t rec;
    if (!load_raw_desc(ix, from, d, rec)) return false;
    if (desc_class(d.B.z) != 0) return block_follow<BD>(ix, d, rec, start, end, dest, rs, re, count);
    const uint64_t rstart = desc_start(d.B.x, d.B.z);
    ByteCursor c(ix.data, rstart, rstart + d.B.y);
    uint64_t sigma;
    if (!c.varint(sigma) || sigma == 0) return false;
    return record_follow<BD>(c, sigma, start, end, dest, rs, re, count);
}

// GBWT::extend, src/gbwt.rs:292-304
__device__ __forceinline__ bool dev_extend(const DeviceIndex &ix, const gbwt_hip_state &st, uint64_t node, gbwt_hip_state &out) {
    if (node < ix.first_node) return false;
    uint64_t rs, re, count;
    if (!dev_follow<false>(ix, st.node, st.start, st.end, node, rs, re, count)) return false;
    out.node = node; out.start = rs; out.end = re;
    return true;
}

// GBWT::extend_forward + bd_internal, src/gbwt.rs:339-347, 371-384
__device__ __forceinline__ bool dev_extend_forward(const DeviceIndex &ix, const gbwt_hip_bd_state &st, uint64_t node, gbwt_hip_bd_state &out) {
    if (node < ix.first_node) return false;
    uint64_t rs, re, count = 0;
    if (!dev_follow<true>(ix, st.forward.node, st.forward.start, st.forward.end, node, rs, re, count)) return false;
    out.forward.node = node; out.forward.start = rs; out.forward.end = re;
    uint64_t pos = st.reverse.start + count;
    out.reverse.node = st.reverse.node; out.reverse.start = pos; out.reverse.end = pos + (re - rs);
    return true;
}

// GBWT::backward, src/gbwt.rs:236-250: predecessor_at on the record of the flipped node, then offset_to in the
// predecessor's record.  Class 1 / 2 records answer both from the descriptor and the rank blocks: the per-edge
// counts are in the descriptor, and "the offset of the k-th position with value v" is a binary search over the
// blocks' running counts plus a select inside one 64-bit word.
__device__ __forceinline__ bool dev_predecessor_at(const DeviceIndex &ix, uint64_t node, uint64_t i, uint64_t &pred) {
    RawDesc d;
    uint64_t rec;
    if (!load_raw_desc(ix, node, d, rec)) return false;
    const uint32_t cls = desc_class(d.B.z);
    if (cls == 0) {
        const uint64_t start = desc_start(d.B.x, d.B.z);
        ByteCursor c(ix.data, start, start + d.B.y);
        uint64_t sigma;
        if (!c.varint(sigma) || sigma == 0) return false;
        return record_predecessor_at(c, sigma, i, pred);
    }
    const uint64_t count0 = cls == 2 ? d.C.x : d.B.w, count1 = cls == 2 ? d.B.w - d.C.x : 0;
    uint64_t n0 = d.A.x == 0 ? 0 : (d.A.x ^ 1u), n1 = d.A.z == 0 ? 0 : (d.A.z ^ 1u);
    uint64_t c0 = count0, c1 = count1;
    if (cls == 2 && (n0 >> 1) == (n1 >> 1)) { uint64_t t = n0; n0 = n1; n1 = t; t = c0; c0 = c1; c1 = t; }
    if (c0 > i) { pred = n0; return n0 != 0; }
    if (cls == 2 && c0 + c1 > i) { pred = n1; return n1 != 0; }
    return false;
}

__device__ __forceinline__ bool dev_offset_to(const DeviceIndex &ix, uint64_t pred, uint64_t node, uint64_t offset, uint64_t &out) {
    RawDesc d;
    uint64_t rec;
    if (!load_raw_desc(ix, pred, d, rec)) return false;
    const uint32_t cls = desc_class(d.B.z);
    if (cls == 0) {
        const uint64_t start = desc_start(d.B.x, d.B.z);
        ByteCursor c(ix.data, start, start + d.B.y);
        uint64_t sigma;
        if (!c.varint(sigma) || sigma == 0) return false;
        return record_offset_to(c, sigma, node, offset, out);
    }
    if (node == 0) return false;
    uint32_t value;
    uint64_t succ_rank;
    if (d.A.x == node) { value = 0; succ_rank = d.A.y; }
    else if (cls == 2 && d.A.z == node) { value = 1; succ_rank = d.A.w; }
    else return false;
    if (succ_rank > offset) return false;
    const uint64_t k = offset - succ_rank;                      // the k-th position (from 0) with this value
    const uint64_t total = value ? d.B.w - d.C.x : (cls == 2 ? d.C.x : d.B.w);
    if (k >= total) return false;
    if (cls == 1) { out = k; return true; }
    // largest block whose running count of `value` is <= k
    const uint4 *blocks = ix.blocks + ix.block_base[rec];
    uint32_t lo = 0, hi = d.B.w >> RANK_BLOCK_SHIFT;            // last block index
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        const uint32_t ones = blocks[mid].z;
        const uint64_t before = value ? ones : (static_cast<uint64_t>(mid) << RANK_BLOCK_SHIFT) - ones;
        if (before <= k) lo = mid; else hi = mid - 1;
    }
    const uint4 K = blocks[lo];
    uint64_t word = (static_cast<uint64_t>(K.y) << 32) | K.x;
    if (!value) word = ~word;
    uint64_t r = k - (value ? K.z : (static_cast<uint64_t>(lo) << RANK_BLOCK_SHIFT) - K.z);
    while (r-- > 0) word &= word - 1;                           // drop the r lowest set bits
    out = (static_cast<uint64_t>(lo) << RANK_BLOCK_SHIFT) + __builtin_ctzll(word);
    return true;
}

__global__ void __launch_bounds__(256) k_backward(DeviceIndex ix, const gbwt_hip_pos *in, uint64_t n, gbwt_hip_pos *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const gbwt_hip_pos p = in[k];
    gbwt_hip_pos r{0, 0};
    uint8_t ok = 0;
    uint64_t pred = 0, off = 0;
    // "This also catches the endmarker" (src/gbwt.rs:239): pos.node <= first_node -> None
    if (p.node > ix.first_node && dev_predecessor_at(ix, p.node ^ 1, p.offset, pred) && dev_offset_to(ix, pred, p.node, p.offset, off)) {
        r.node = pred; r.offset = off; ok = 1;
    }
    out[k] = r; valid[k] = ok;
}

__global__ void __launch_bounds__(256) k_find(DeviceIndex ix, const uint64_t *nodes, uint64_t n, gbwt_hip_state *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    gbwt_hip_state st{0, 0, 0};
    uint8_t ok = dev_find(ix, nodes[k], st) ? 1 : 0;
    out[k] = st; valid[k] = ok;
}

__global__ void __launch_bounds__(256) k_extend(DeviceIndex ix, const gbwt_hip_state *states, const uint64_t *nodes, uint64_t n,
                                                 gbwt_hip_state *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    gbwt_hip_state st = states[k], r{0, 0, 0};
    uint8_t ok = dev_extend(ix, st, nodes[k], r) ? 1 : 0;
    out[k] = r; valid[k] = ok;
}

// GBWT::bd_find, src/gbwt.rs:311-324
__global__ void __launch_bounds__(256) k_bd_find(DeviceIndex ix, const uint64_t *nodes, uint64_t n, gbwt_hip_bd_state *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    gbwt_hip_bd_state r{{0, 0, 0}, {0, 0, 0}};
    gbwt_hip_state st{0, 0, 0};
    uint8_t ok = dev_find(ix, nodes[k], st) ? 1 : 0;
    if (ok) { r.forward = st; r.reverse.node = st.node ^ 1; r.reverse.start = st.start; r.reverse.end = st.end; }
    out[k] = r; valid[k] = ok;
}

// extend_forward, or extend_backward = flip(extend_forward(flip(state), node ^ 1)) (src/gbwt.rs:362-367, 506-511)
__global__ void __launch_bounds__(256) k_bd_extend(DeviceIndex ix, const gbwt_hip_bd_state *states, const uint64_t *nodes, uint64_t n,
                                                    bool backward, gbwt_hip_bd_state *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    gbwt_hip_bd_state st = states[k], r{{0, 0, 0}, {0, 0, 0}}, zero{{0, 0, 0}, {0, 0, 0}};
    uint64_t node = nodes[k];
    if (backward) { gbwt_hip_state t = st.forward; st.forward = st.reverse; st.reverse = t; node ^= 1; }
    uint8_t ok = dev_extend_forward(ix, st, node, r) ? 1 : 0;
    if (ok && backward) { gbwt_hip_state t = r.forward; r.forward = r.reverse; r.reverse = t; }
    out[k] = ok ? r : zero; valid[k] = ok;
}

// GBZ::follow_forward / follow_backward + StateIter (src/gbz.rs:519-544, 1211-1251): every non-empty extension of a
// bidirectional state by one node, in the order of the edge list of the state's last (first) node.  One lane per
// state; `out` == nullptr only counts.  Returns the number of extensions, or -1 where the reference returns no
// iterator (GBZ::successors: the node does not exist).  `backward`: the state is flipped, followed forward, and the
// results are flipped back.
__device__ __forceinline__ int64_t dev_follow_all(const DeviceIndex &ix, gbwt_hip_bd_state st, bool backward, gbwt_hip_bd_state *out) {
    if (backward) { const gbwt_hip_state t = st.forward; st.forward = st.reverse; st.reverse = t; }
    const uint64_t node = st.forward.node;
    RawDesc d;
    uint64_t rec;
    if (!load_raw_desc(ix, node & ~uint64_t(1), d, rec)) return -1;   // GBZ::has_node: the forward record exists
    if (!load_raw_desc(ix, node, d, rec)) return -1;
    const uint32_t cls = desc_class(d.B.z);
    const uint64_t start = desc_start(d.B.x, d.B.z);
    ByteCursor c(ix.data, start, start + (cls == 0 ? d.B.y : 0u));
    uint64_t sigma = cls;
    if (cls == 0 && (!c.varint(sigma) || sigma == 0)) return -1;
    int64_t count = 0;
    uint64_t succ = 0;
    for (uint64_t e = 0; e < sigma; e++) {
        if (cls == 0) {
            uint64_t delta, off;
            if (!c.varint(delta) || !c.varint(off)) break;
            succ += delta;
        } else succ = e == 0 ? d.A.x : d.A.z;
        if (succ == 0) continue;                       // EdgeIter starts behind an ENDMARKER edge (src/gbz.rs:833)
        gbwt_hip_bd_state r;
        if (!dev_extend_forward(ix, st, succ, r)) continue;   // bd_internal -> None: the extension is empty
        if (out) {
            if (backward) { const gbwt_hip_state t = r.forward; r.forward = r.reverse; r.reverse = t; }
            out[count] = r;
        }
        count++;
    }
    return count;
}

__global__ void __launch_bounds__(256) k_follow_count(DeviceIndex ix, const gbwt_hip_bd_state *states, uint64_t n, bool backward, uint64_t *counts, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const int64_t c = dev_follow_all(ix, states[k], backward, nullptr);
    counts[k] = c < 0 ? 0 : static_cast<uint64_t>(c);
    valid[k] = c < 0 ? 0 : 1;
}

__global__ void __launch_bounds__(256) k_follow_fill(DeviceIndex ix, const gbwt_hip_bd_state *states, uint64_t n, bool backward, const uint64_t *offsets,
                                                      gbwt_hip_bd_state *out) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    if (offsets[k + 1] > offsets[k]) dev_follow_all(ix, states[k], backward, out + offsets[k]);
}

// find(q[0]) then extend over q[1..len) in one launch (src/bin/benchmark.rs:155-169)
__global__ void __launch_bounds__(256) k_search(DeviceIndex ix, const uint64_t *queries, uint64_t n, uint64_t len,
                                                 gbwt_hip_state *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const uint64_t *q = queries + k * len;
    gbwt_hip_state st{0, 0, 0}, zero{0, 0, 0};
    bool ok = len > 0 && dev_find(ix, q[0], st);
    for (uint64_t j = 1; ok && j < len; j++) {
        gbwt_hip_state nx;
        ok = dev_extend(ix, st, q[j], nx);
        st = nx;
    }
    out[k] = ok ? st : zero; valid[k] = ok ? 1 : 0;
}

// bd_find(q[first]) then alternating extend_forward / extend_backward until the whole row is consumed
__global__ void __launch_bounds__(256) k_bd_search(DeviceIndex ix, const uint64_t *queries, uint64_t n, uint64_t len, uint64_t first,
                                                    gbwt_hip_bd_state *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const uint64_t *q = queries + k * len;
    gbwt_hip_bd_state st{{0, 0, 0}, {0, 0, 0}}, zero{{0, 0, 0}, {0, 0, 0}};
    gbwt_hip_state f{0, 0, 0};
    bool ok = first < len && dev_find(ix, q[first], f);
    if (ok) { st.forward = f; st.reverse.node = f.node ^ 1; st.reverse.start = f.start; st.reverse.end = f.end; }
    uint64_t fw = first + 1, bw = first;
    while (ok && (fw < len || bw > 0)) {
        gbwt_hip_bd_state nx;
        if (fw < len) { ok = dev_extend_forward(ix, st, q[fw], nx); st = nx; fw++; }
        if (ok && bw > 0) {  // extend_backward = flip(extend_forward(flip(state), node ^ 1)), src/gbwt.rs:362-367
            gbwt_hip_bd_state fl;
            fl.forward = st.reverse; fl.reverse = st.forward;
            ok = dev_extend_forward(ix, fl, q[bw - 1] ^ 1, nx);
            st.forward = nx.reverse; st.reverse = nx.forward;
            bw--;
        }
    }
    out[k] = ok ? st : zero; valid[k] = ok ? 1 : 0;
}

inline unsigned grid_for(uint64_t n, unsigned block) { return static_cast<unsigned>((n + block - 1) / block); }

}  // namespace

void launch_record_stats(const DeviceIndex &ix, uint64_t *d_stats, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_record_stats, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_stats);
}

void launch_build_desc(const DeviceIndex &ix, uint4 *d_desc, uint32_t *d_block_counts, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_build_desc, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_desc, d_block_counts);
}

void launch_link_desc(const DeviceIndex &ix, uint4 *d_desc, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_link_desc, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_desc);
}

void launch_link_lookahead(const DeviceIndex &ix, uint4 *d_desc, const uint32_t *d_block_counts, uint32_t hops, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_link_lookahead, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_desc, d_block_counts, hops);
}

void launch_table_counts(const DeviceIndex &ix, uint64_t *d_positions, uint64_t *d_sigmas, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_table_counts, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_positions, d_sigmas);
}

void launch_fill_tables(const DeviceIndex &ix, uint4 *d_desc_raw, const uint64_t *d_table_base, const uint64_t *d_edge_base, uint4 *d_tables,
                        uint2 *d_edges, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_fill_tables, dim3(grid_for(ix.n_records, 64)), dim3(64), 0, stream, ix, d_desc_raw, d_table_base, d_edge_base, d_tables, d_edges);
}

void launch_link_desc2(const DeviceIndex &ix, uint4 *d_desc2, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_link_desc2, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_desc2);
}

void launch_fill_cblocks(const DeviceIndex &ix, const uint32_t *d_block_counts, uint4 *d_cblocks, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_fill_cblocks, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_block_counts, d_cblocks);
}

void launch_link_lookahead2(const DeviceIndex &ix, uint4 *d_desc2, const uint32_t *d_block_counts, uint32_t hops, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_link_lookahead2, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_desc2, d_block_counts, hops);
}

void launch_fill_blocks(const DeviceIndex &ix, const uint32_t *d_block_counts, const uint32_t *d_block_base, uint4 *d_blocks, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_fill_blocks, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_block_counts, d_block_base, d_blocks);
}

// exclusive scan of the per-record block counts, then block_base = 1 + scan (block 0 is the shared all-zero block)
// or BLOCK_NONE where the count is 0
__global__ void __launch_bounds__(256) k_finish_block_base(const uint32_t *counts, uint32_t *block_base, uint64_t n) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec < n) block_base[rec] = counts[rec] == 0 ? BLOCK_NONE : block_base[rec] + 1;
}

size_t block_scan_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, static_cast<const uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr), static_cast<int>(n));
    return bytes;
}

void launch_block_scan(const uint32_t *d_counts, uint32_t *d_block_base, uint64_t n, void *d_temp, size_t temp_bytes, hipStream_t stream) {
    if (n == 0) return;
    (void)hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_counts, d_block_base, static_cast<int>(n), stream);
}

void launch_finish_block_base(const uint32_t *d_counts, uint32_t *d_block_base, uint64_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_finish_block_base, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_counts, d_block_base, n);
}

void launch_endmarker_sigma(const DeviceIndex &ix, uint64_t *d_result, hipStream_t stream) {
    hipLaunchKernelGGL(k_endmarker_sigma, dim3(1), dim3(1), 0, stream, ix, d_result);
}

void launch_endmarker_decompress(const DeviceIndex &ix, uint2 *d_out, uint64_t n_out, uint64_t *d_scratch,
                                 uint64_t *d_result, hipStream_t stream) {
    hipLaunchKernelGGL(k_endmarker_decompress, dim3(1), dim3(1), 0, stream, ix, d_out, n_out, d_scratch, d_result);
}

void launch_walk(const DeviceIndex &ix, const WalkArgs &args, hipStream_t stream) {
    if (args.n == 0) return;
    if (args.mode == WALK_LANE_SERIAL) {
        hipLaunchKernelGGL(k_walk, dim3(grid_for(args.n, WAVE)), dim3(WAVE), 0, stream, ix, args);
        return;
    }
    const unsigned p = args.paths_per_wave ? args.paths_per_wave : WAVE;
    const dim3 grid(grid_for(args.n, p)), block(WAVE);
    if (args.mode == WALK_COOP) {
        if (args.pack16) hipLaunchKernelGGL((k_walk_coop<true>), grid, block, 0, stream, ix, args);
        else hipLaunchKernelGGL((k_walk_coop<false>), grid, block, 0, stream, ix, args);
        return;
    }
    // walking wave + look-ahead helper wave
    if (args.mode == WALK_ONE_STEP) { hipLaunchKernelGGL(k_walk_blocks, grid, dim3(2 * WAVE), 0, stream, ix, args); return; }
    hipLaunchKernelGGL(k_walk_two, grid, dim3(2 * WAVE), 0, stream, ix, args);
}

void launch_compact(const WalkArgs &args, const uint64_t *d_offsets, uint32_t *d_nodes, hipStream_t stream) {
    if (args.n == 0) return;
    hipLaunchKernelGGL(k_compact, dim3(grid_for(args.n, 256 / WAVE)), dim3(256), 0, stream, args, d_offsets, d_nodes);
}

void launch_path_sums(const uint64_t *d_offsets, const uint32_t *d_nodes, uint64_t n, uint64_t *d_sums, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_path_sums, dim3(grid_for(n, 256 / WAVE)), dim3(256), 0, stream, d_offsets, d_nodes, n, d_sums);
}

void launch_start(const DeviceIndex &ix, const uint64_t *ids, uint64_t n, gbwt_hip_pos *out, uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_start, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, ids, n, out, valid);
}
void launch_forward(const DeviceIndex &ix, const gbwt_hip_pos *in, uint64_t n, gbwt_hip_pos *out, uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_forward, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, in, n, out, valid);
}
void launch_backward(const DeviceIndex &ix, const gbwt_hip_pos *in, uint64_t n, gbwt_hip_pos *out, uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_backward, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, in, n, out, valid);
}
void launch_find(const DeviceIndex &ix, const uint64_t *nodes, uint64_t n, gbwt_hip_state *out, uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_find, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, nodes, n, out, valid);
}
void launch_extend(const DeviceIndex &ix, const gbwt_hip_state *states, const uint64_t *nodes, uint64_t n,
                   gbwt_hip_state *out, uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_extend, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, states, nodes, n, out, valid);
}
void launch_bd_find(const DeviceIndex &ix, const uint64_t *nodes, uint64_t n, gbwt_hip_bd_state *out, uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_bd_find, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, nodes, n, out, valid);
}
void launch_bd_extend(const DeviceIndex &ix, const gbwt_hip_bd_state *states, const uint64_t *nodes, uint64_t n,
                      bool backward, gbwt_hip_bd_state *out, uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_bd_extend, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, states, nodes, n, backward, out, valid);
}
void launch_follow_count(const DeviceIndex &ix, const gbwt_hip_bd_state *states, uint64_t n, bool backward, uint64_t *counts, uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_follow_count, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, states, n, backward, counts, valid);
}
void launch_follow_fill(const DeviceIndex &ix, const gbwt_hip_bd_state *states, uint64_t n, bool backward, const uint64_t *offsets,
                        gbwt_hip_bd_state *out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_follow_fill, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, states, n, backward, offsets, out);
}
void launch_search(const DeviceIndex &ix, const uint64_t *queries, uint64_t n, uint64_t len, gbwt_hip_state *out,
                   uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_search, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, queries, n, len, out, valid);
}

void launch_bd_search(const DeviceIndex &ix, const uint64_t *queries, uint64_t n, uint64_t len, uint64_t first,
                      gbwt_hip_bd_state *out, uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_bd_search, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, queries, n, len, first, out, valid);
}

size_t scan_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, bytes, static_cast<const uint64_t *>(nullptr), static_cast<uint64_t *>(nullptr),
                                     static_cast<int>(n));
    return bytes;
}

void launch_scan(const uint64_t *d_lengths, uint64_t *d_offsets, uint64_t n, void *d_temp, size_t temp_bytes, hipStream_t s) {
    (void)hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), s);
    if (n == 0) return;
    (void)hipcub::DeviceScan::InclusiveSum(d_temp, temp_bytes, d_lengths, d_offsets + 1, static_cast<int>(n), s);
}

}  // namespace gbwt_hip
