// load_kernels.hip -- load-time passes: descriptors, rank blocks, two-step descriptors and blocks, LF tables, endmarker
// (hand-written HIP for gfx950, no MFMA: integer pointer-chasing over a byte stream).  Launch wrappers are declared in kernels.hpp.
#include "kernels.hpp"

#include <hipcub/hipcub.hpp>

#include "device_common.hpp"
#include "lf_device.hpp"

namespace gbwt_hip {

namespace {

// ---------------------------------------------------------------------------------------------
// Load-time passes

// 16 bytes of the stream at data[pos..], zero-filled past `limit`.
__device__ __forceinline__ uint4 stream_bytes16(const uint8_t *data, uint64_t pos, uint64_t limit) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16 && pos + k < limit; k++) w[k >> 2] |= static_cast<uint32_t>(data[pos + k]) << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// One lane per record: the RAW descriptor (device_index.hpp), the number of rank blocks the record gets, and the statistics of the
// index: stats[0] = max Record::len, [1] = max outdegree, [2] = records without a readable outdegree, [3] = all BWT positions
// (= GBWT::len of a consistent index; bounds the walks at open), [4] = class 0 records (the candidates for LF tables).  Reduced over the wave first: two million lanes on four addresses
// took 25 ms of atomics on the headline index.
__global__ void __launch_bounds__(256) k_build_desc(DeviceIndex ix, uint4 *desc, uint32_t *block_counts, uint64_t *stats) {
    const uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    uint64_t stat_len = 0, stat_sigma = 0, stat_bad = 0, stat_generic = 0;
    if (rec < ix.n_records) {
    uint64_t start, limit;
    record_bounds(ix, rec, start, limit);
    uint4 A = make_uint4(0, 0, 0, 0), B = make_uint4(0, 0, 0, 0), C = make_uint4(0, 0, 0, 0), D = make_uint4(0, 0, 0, 0);
    uint32_t n_blocks = 0;
    if (limit > start) {
        ByteCursor c(ix.data, start, limit);
        uint64_t sigma = 0;
        const bool readable = c.varint(sigma);
        if (!readable) stat_bad = 1;
        if (readable && sigma != 0) {
            stat_sigma = sigma;
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
                        stat_len = total;
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
                stat_generic = rec != 0 ? 1 : 0;            // (record 0, the endmarker, gets no table)
                A = make_uint4(0, 0, 0, 0); C = make_uint4(0, 0, 0, 0); D = make_uint4(0, 0, 0, 0); B.w = 0;
                // Record::len by scanning the runs -- except for record 0, the endmarker, which has a run per sequence in a bidirectional
                // index (10 000 on the headline: one lane scanning them was 2 of this kernel's 2.6 ms) and one position per sequence by
                // definition (src/gbwt.rs:413-414; the loader decompresses it and stops at that many)
                uint64_t total = ix.n_sequences;
                if (rec != 0) {
                    ByteCursor c2(ix.data, start, limit);
                    uint64_t s2 = 0;
                    c2.varint(s2);
                    total = record_len(c2, s2);
                }
                stat_len = total;
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
    uint64_t max_len = stat_len, max_sigma = stat_sigma, sum = stat_len, bad = stat_bad + (stat_generic << 32);   // two counts in one word: records < 2^30
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        max_len = max(max_len, static_cast<uint64_t>(__shfl_down(static_cast<unsigned long long>(max_len), d)));
        max_sigma = max(max_sigma, static_cast<uint64_t>(__shfl_down(static_cast<unsigned long long>(max_sigma), d)));
        sum += __shfl_down(static_cast<unsigned long long>(sum), d);
        bad += __shfl_down(static_cast<unsigned long long>(bad), d);
    }
    // ... and over the workgroup, and a maximum is only sent where it would change what is there (it only grows: a stale read sends one
    // atomic too many, never one too few): 218 M records were 10 M atomics on three addresses
    __shared__ uint64_t part[4][4];
    const uint32_t wave = threadIdx.x / WAVE;
    if (threadIdx.x % WAVE == 0) { part[wave][0] = max_len; part[wave][1] = max_sigma; part[wave][2] = sum; part[wave][3] = bad; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (uint32_t w = 1; w < 4; w++) { max_len = max(max_len, part[w][0]); max_sigma = max(max_sigma, part[w][1]); sum += part[w][2]; bad += part[w][3]; }
    const volatile uint64_t *seen = stats;
    if (max_len > seen[0]) atomicMax(reinterpret_cast<unsigned long long *>(stats + 0), static_cast<unsigned long long>(max_len));
    if (max_sigma > seen[1]) atomicMax(reinterpret_cast<unsigned long long *>(stats + 1), static_cast<unsigned long long>(max_sigma));
    if (bad & 0xFFFFFFFFull) atomicAdd(reinterpret_cast<unsigned long long *>(stats + 2), static_cast<unsigned long long>(bad & 0xFFFFFFFFull));
    if (bad >> 32) atomicAdd(reinterpret_cast<unsigned long long *>(stats + 4), static_cast<unsigned long long>(bad >> 32));
    if (sum) atomicAdd(reinterpret_cast<unsigned long long *>(stats + 3), static_cast<unsigned long long>(sum));
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

// Walk tables (DeviceIndex::wtables): the LF tables with what a WALK does next folded in, one block per table record.
// Entry i of record v starts from the plain entry (successor n_u, offset y, landing record u):
//  * u unary, not the end of the sequence, and the record w behind it exists with o0(u) + y < Record::len(w): the step
//    through u is taken here -- emit n_u and the node of w, land on w at o0(u) + y (LEAF_EMIT2), exactly the two
//    SequenceIter steps of the reference (src/gbwt.rs:557-568, src/bwt.rs:480-496 with outdegree 1: every position maps
//    to (n0, o0 + i));
//  * otherwise the plain step: emit n_u, land on u (0 = the walk ends behind n_u).
// Either way word 3 is what the walk needs to go on without another lookup: the table base of the landing record when
// that is a table record itself (WT_TABLE; the offset has been checked against its length, src/bwt.rs:481), else its
// block base.  A chain of multi-allelic sites is then one 16-byte load per site.
__global__ void __launch_bounds__(256) k_fill_wtables(DeviceIndex ix, uint4 *wtables) {
    const uint64_t v = blockIdx.x;
    if (v >= ix.n_records) return;
    const uint4 *raw = ix.desc_raw;
    const uint4 C = raw[4 * v + 2];
    if (C.w != 1u) return;
    for (uint32_t i = threadIdx.x; i < C.y; i += blockDim.x) {
        const uint4 e = ix.tables[static_cast<uint64_t>(C.z) + i];
        uint4 out = make_uint4(e.x, e.y, 0u, BLOCK_NONE);
        uint64_t land = e.z;
        uint32_t offset = e.y, flags = 0;
        if (land != 0) {
            const uint4 UA = raw[4 * land], UB = raw[4 * land + 1];
            uint64_t w = 0;
            if (UB.y == DESC_UNARY && UA.x != 0 && landing_record(ix, UA.x, w) && static_cast<uint64_t>(UA.y) + e.y <= 0xFFFFFFFFull) {
                const uint4 WB = raw[4 * w + 1], WC = raw[4 * w + 2];
                const uint64_t wlen = WB.y == 0 ? 0 : (desc_class(WB.z) == 0 ? (WC.w == 1u ? WC.y : 0) : WB.w);   // class 0 without a table: not taken here
                if (static_cast<uint64_t>(UA.y) + e.y < wlen) { land = w; offset = UA.y + e.y; flags = LEAF_EMIT2; }
            }
            const uint4 LB = raw[4 * land + 1], LC = raw[4 * land + 2];
            if (LB.y != 0 && desc_class(LB.z) == 0 && LC.w == 1u) {
                if (offset < LC.y) out = make_uint4(e.x, offset, static_cast<uint32_t>(land) | flags | WT_TABLE, LC.z);
                else out = make_uint4(e.x, offset, 0u, BLOCK_NONE);          // i >= Record::len -> None: the walk ends behind n_u (never with LEAF_EMIT2: checked above)
            } else {
                out = make_uint4(e.x, offset, static_cast<uint32_t>(land) | flags, flags ? ix.block_base[land] : e.w);
            }
        }
        wtables[static_cast<uint64_t>(C.z) + i] = out;
    }
}

// Deep walk tables (DeviceIndex::wtables_deep): WT_DEEP_STEPS table steps of the walk folded into ONE 64-byte entry per position.  A table
// step is one dependent load from a table far larger than the caches, so a walker spends its life waiting for HBM (config C5: 2 500 cycles
// per step with 512 walkers per CU, rows not even stored); an entry that already holds the next seven steps -- what each of them emits and
// where the last one lands -- makes that one wait per seven steps, and the 64 bytes are exactly the sector the single entry was fetched with.
// Entry of position i of table record v, as four uint4: words 2 k, 2 k + 1 = step k {node to emit (0: nothing), landing record | LEAF_EMIT2},
// k = 0 .. 6; word 14 = offset in the last landing record, word 15 = its table base (WT_TABLE set in word 13) or its block base.  The chain
// stops early where the walk ends (record 0) or leaves the table records; the steps not taken repeat the last landing record and emit nothing.
// The order of the LF steps (src/gbwt.rs:557-568) is untouched: the entry is the memo of seven of them.
//
// COMPACT ENTRIES (round 4).  Where the next TWELVE steps all stay on table records, each emits two nodes (a successor and the node of the
// record behind its unary record: LEAF_EMIT2), and every one of those 24 node ids lies within a signed 16-bit delta of the one before it
// -- a chain of multi-allelic sites in a graph whose ids are sorted topologically -- the same 64 bytes hold all twelve:
//   word 0 = the first node, word 1 = WT_COMPACT | WT_COMPACT_TABLE?, words 2 .. 13 = 23 deltas of 16 bits (node k + 1 - node k; two per
//   word, low half first), word 14 = offset in the last landing record (= last node - alphabet_offset), word 15 = its table / block base.
// 64 bytes read per 96 bytes emitted instead of per 56: config 5's traffic 2.4 -> ... x the emitted bytes, one dependent load per 24 nodes.
// Bit 30 of word 1 tells the forms apart (a wide entry has the landing record of step 0 there: record indices are below 2^30, and
// WT_TABLE is masked off every step but the last).
__global__ void __launch_bounds__(256) k_fill_wtables_deep(DeviceIndex ix, uint4 *deep, uint32_t compact) {
    const uint64_t v = blockIdx.x;
    if (v >= ix.n_records) return;
    const uint4 C = ix.desc_raw[4 * v + 2];
    if (C.w != 1u) return;
    for (uint32_t i = threadIdx.x; i < C.y; i += blockDim.x) {
        uint32_t w[16];
        uint4 e = ix.wtables[static_cast<uint64_t>(C.z) + i];
        if (compact) {
            uint32_t node[2 * WT_COMPACT_STEPS];
            uint4 f = e;
            bool ok = true;
            for (uint32_t k = 0; k < WT_COMPACT_STEPS && ok; k++) {
                if (k > 0) f = ix.wtables[static_cast<uint64_t>(f.w) + f.y];
                const bool table_next = (f.z & WT_TABLE) != 0;
                ok = f.x != 0 && (f.z & LEAF_EMIT2) != 0 && (f.z & REC_MASK) != 0 && (table_next || k + 1 == WT_COMPACT_STEPS);
                node[2 * k] = f.x; node[2 * k + 1] = (f.z & REC_MASK) + ix.alphabet_offset;
            }
            for (uint32_t k = 1; k < 2 * WT_COMPACT_STEPS && ok; k++) {
                const int64_t d = static_cast<int64_t>(node[k]) - static_cast<int64_t>(node[k - 1]);
                ok = d >= -32768 && d <= 32767;
            }
            if (ok) {
                w[0] = node[0];
                w[1] = WT_COMPACT | ((f.z & WT_TABLE) ? WT_COMPACT_TABLE : 0u);
                for (uint32_t q = 0; q < 12; q++) {
                    const uint32_t a = 2 * q + 1, b = 2 * q + 2;                       // deltas into node a and node b
                    const uint32_t lo = (node[a] - node[a - 1]) & 0xFFFFu;
                    const uint32_t hi = b < 2 * WT_COMPACT_STEPS ? ((node[b] - node[b - 1]) & 0xFFFFu) : 0u;
                    w[2 + q] = lo | (hi << 16);
                }
                w[14] = f.y; w[15] = f.w;
                uint4 *out = deep + 4 * (static_cast<uint64_t>(C.z) + i);
                out[0] = make_uint4(w[0], w[1], w[2], w[3]);
                out[1] = make_uint4(w[4], w[5], w[6], w[7]);
                out[2] = make_uint4(w[8], w[9], w[10], w[11]);
                out[3] = make_uint4(w[12], w[13], w[14], w[15]);
                continue;
            }
        }
        bool going = true;
#pragma unroll
        for (uint32_t k = 0; k < WT_DEEP_STEPS; k++) {
            if (going) {
                if (k > 0) e = ix.wtables[static_cast<uint64_t>(e.w) + e.y];
                w[2 * k] = e.x; w[2 * k + 1] = e.z & ~WT_TABLE;
                going = (e.z & REC_MASK) != 0 && (e.z & WT_TABLE) != 0;
            } else {
                w[2 * k] = 0u; w[2 * k + 1] = e.z & REC_MASK;
            }
        }
        if ((e.z & REC_MASK) != 0 && (e.z & WT_TABLE) != 0) w[2 * WT_DEEP_STEPS - 1] |= WT_TABLE;
        w[14] = e.y; w[15] = e.w;
        uint4 *out = deep + 4 * (static_cast<uint64_t>(C.z) + i);
        out[0] = make_uint4(w[0], w[1], w[2], w[3]);
        out[1] = make_uint4(w[4], w[5], w[6], w[7]);
        out[2] = make_uint4(w[8], w[9], w[10], w[11]);
        out[3] = make_uint4(w[12], w[13], w[14], w[15]);
    }
}

// ---- two-step walk: descriptors and blocks -----------------------------------------------------------------
// The single-step descriptor says, per edge of record v: what to emit and where the walk lands (record w, offset base).
// The two-step descriptor composes that with the edges of w, so that one iteration of the walk -- one round trip to
// memory -- takes TWO LF steps (up to four nodes with fused unary successors):
//   desc2[8 * v + a] = E_a = {node to emit for edge a, offset base in w_a, w_a | LEAF_EMIT2 | DESC2_SLOW, GATHER_OK}   (first step, a = 0, 1;
//                            DESC2_SLOW: whole record, set in both; one 12-byte load per edge, so that a walker that knows a needs one)
//   desc2[8 * v + 2 + 2 * a + b] = leaf (a, b) = {node to emit, offset base, landing record | LEAF_EMIT2, block base}
//   desc2[8 * v + 6] = look-ahead {record, first block, number of blocks, 0};  [7] unused
// The second step exists (is "real") when w_a is unary (descriptor only: its value is always 0) or when both v and w_a
// have rank blocks: v's two-step block then carries, for each of its 64 offsets, the value the sequence has in w_a
// (bits2) and the number of value-1 positions of w_a before the landing offset of the block's first a-path (R_a), so
// the rank inside w_a is again one popcount.  Where the second step is not real (w_a generic, sequence ending, v
// unary and w_a branching) the leaf (a, 0) is the identity: "emit nothing, stay in w_a at the offset reached".
// CHAINS.  A fused step emits the successor x and the node of the record w behind it (a unary record stepped through).  Where w is
// unary as well, and the node behind it is the next id in the same direction (x, x + 2 = node of w, x + 4, ...: a GFA segment chopped
// into nodes, an insertion that got consecutive ids), the step runs on: every further unary record costs an addition of its o0 to the
// offset base -- Record::lf of an outdegree-1 record is (n0, o0 + i), src/bwt.rs:480-496 -- and one more node to emit, which the walk
// derives from x and the landing node (chain_stride / chain_mids, device_index.hpp).  The same tests as for the first fusion apply at
// every record (the landing record exists, is not empty, and holds every offset the step can produce).  Rows that took alleles of
// different lengths then land on the SAME record after one step and the wave stays in the uniform loop; a chopped segment is one step.
// `count` = positions that can take the step; returns the number of records added (at most `room`).
__device__ __forceinline__ uint32_t extend_chain(const DeviceIndex &ix, uint32_t x, uint32_t &w, uint64_t &base, uint64_t count, uint32_t room, bool bidirectional) {
    const uint4 *raw = ix.desc_raw;
    const uint32_t first = w + ix.alphabet_offset;
    if (first != x + 2u && first != x - 2u) return 0;
    const uint32_t d = chain_stride(x, first);
    uint32_t added = 0;
    while (added < room) {
        const uint4 WA = raw[4 * static_cast<uint64_t>(w)], WB = raw[4 * static_cast<uint64_t>(w) + 1];
        if (WB.y != DESC_UNARY || WA.x == 0 || WA.x != w + ix.alphabet_offset + d) break;
        // ... and never THROUGH a record where paths merge: rows that arrive over different alleles must all stop there, or the chain of
        // the one allele whose ids happen to run on into the merge node would carry its rows past the others.  In a bidirectional index
        // the predecessors of a node are the successors of its reverse (src/gbwt.rs:229-241): one predecessor = the reverse record is unary.
        if (bidirectional) {
            uint64_t rev = 0;
            if (!landing_record(ix, (w + ix.alphabet_offset) ^ 1u, rev) || raw[4 * rev + 1].y != DESC_UNARY) break;
        }
        uint64_t next = 0;
        if (!landing_record(ix, WA.x, next)) break;
        const uint4 NB = raw[4 * next + 1];
        const uint64_t base2 = base + WA.y;
        if (NB.y == 0 || base2 + count > 0xFFFFFFFFull || !(desc_class(NB.z) == 0 || base2 + count <= NB.w)) break;
        w = static_cast<uint32_t>(next); base = base2; added++;
    }
    return added;
}

// positions of record r that take edge e (0 for an edge it does not have)
__device__ __forceinline__ uint64_t edge_positions(const DeviceIndex &ix, uint64_t r, uint32_t e) {
    const uint4 B = ix.desc_raw[4 * r + 1], C = ix.desc_raw[4 * r + 2];
    const uint32_t cls = B.y != 0 ? desc_class(B.z) : 0u;
    if (cls == 2) return e ? B.w - C.x : C.x;
    return cls == 1 && e == 0 ? B.w : 0u;
}

__global__ void __launch_bounds__(256) k_link_desc2(DeviceIndex ix, uint4 *out, uint32_t gather_limit, uint32_t chain_max, uint32_t *chained) {
    uint64_t v = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (v >= ix.n_records) return;
    const uint4 *d1 = ix.desc;
    const uint4 D = d1[4 * v + 2];
    const uint4 VB = ix.desc_raw[4 * v + 1];
    const uint32_t cls_v = VB.y != 0 ? desc_class(VB.z) : 0u;
    uint32_t n1[2] = {0, 0}, base[2] = {0, 0}, wword[2] = {0, 0}, chain[2] = {0, 0};
    bool any_chain = false;
    uint4 leaf[4] = {make_uint4(0, 0, 0, BLOCK_NONE), make_uint4(0, 0, 0, BLOCK_NONE), make_uint4(0, 0, 0, BLOCK_NONE), make_uint4(0, 0, 0, BLOCK_NONE)};
    const bool slow = (D.x & DESC_SLOW) != 0;
    if (!slow) {
        for (uint32_t a = 0; a < 2; a++) {
            const uint4 E = d1[4 * v + a];
            const uint32_t f = a ? D.w : D.y;
            n1[a] = E.x; base[a] = E.y;
            if (!(f & EDGE_CONT)) continue;                       // the walk ends behind this edge: both leaves park it
            uint32_t w = E.z, wbb = E.w;
            if ((f & EDGE_EMIT2) && chain_max != 0) {
                uint64_t chained_base = base[a];
                if (extend_chain(ix, E.x, w, chained_base, edge_positions(ix, v, a), chain_max & 0xFFu, (chain_max >> 8) != 0) != 0) {
                    base[a] = static_cast<uint32_t>(chained_base); wbb = ix.block_base[w]; chain[a] = E_CHAIN; any_chain = true;
                }
            }
            wword[a] = w | ((f & EDGE_EMIT2) ? LEAF_EMIT2 : 0u);
            const uint4 WD = d1[4 * static_cast<uint64_t>(w) + 2];
            const uint4 WB = ix.desc_raw[4 * static_cast<uint64_t>(w) + 1];
            const uint32_t cls_w = WB.y != 0 ? desc_class(WB.z) : 0u;
            const bool real = !(WD.x & DESC_SLOW) && (cls_w == 1 || (cls_w == 2 && cls_v == 2));
            if (!real) { leaf[2 * a] = make_uint4(0u, 0u, w, wbb); continue; }   // identity
            for (uint32_t b = 0; b < cls_w; b++) {
                const uint4 WE = d1[4 * static_cast<uint64_t>(w) + b];
                const uint32_t wf = b ? WD.w : WD.y;
                const bool cont = (wf & EDGE_CONT) != 0;
                uint32_t land = WE.z, lbb = WE.w, lflag = 0;
                uint64_t lbase = WE.y;
                if (cont && (wf & EDGE_EMIT2) && chain_max != 0 && extend_chain(ix, WE.x, land, lbase, edge_positions(ix, w, b), chain_max & 0xFFu, (chain_max >> 8) != 0) != 0) {
                    lbb = ix.block_base[land]; lflag = LEAF_CHAIN; any_chain = true;
                }
                leaf[2 * a + b] = make_uint4(WE.x, static_cast<uint32_t>(lbase), cont ? (land | lflag | ((wf & EDGE_EMIT2) ? LEAF_EMIT2 : 0u)) : 0u, cont ? lbb : BLOCK_NONE);
            }
        }
    }
    // the gather loop's packed blocks hold counts of 21 / 22 bits: positions of v and of the records behind its edges
    bool packed = VB.w < gather_limit;
    for (uint32_t a = 0; a < 2; a++)
        if (wword[a] & REC_MASK) packed = packed && ix.desc_raw[4 * static_cast<uint64_t>(wword[a] & REC_MASK) + 1].w < gather_limit;
    // E_ALL4: every edge this record has emits a node and is fused (two nodes), and so does every leaf behind it: the uniform loop then
    // stages four nodes per lane and iteration without counting them (walk_loops.hpp)
    bool all4 = !slow && !any_chain && cls_v != 0;
    for (uint32_t a = 0; a < cls_v && all4; a++) {
        all4 = n1[a] != 0 && (wword[a] & LEAF_EMIT2) != 0;
        const uint4 WB = ix.desc_raw[4 * static_cast<uint64_t>(wword[a] & REC_MASK) + 1];
        const uint32_t cls_w = WB.y != 0 ? desc_class(WB.z) : 0u;
        all4 = all4 && cls_w != 0;
        for (uint32_t b = 0; b < cls_w && all4; b++) all4 = leaf[2 * a + b].x != 0 && (leaf[2 * a + b].z & LEAF_EMIT2) != 0;
    }
    if (slow && VB.y != 0 && chained != nullptr) atomicAdd(chained + 2, 1u);        // non-empty records that take the generic decoder (class 0, edges that failed a check)
    uint4 *o = out + 8 * v;
    o[0] = make_uint4(n1[0], base[0], wword[0] | (slow ? DESC2_SLOW : 0u), (packed ? GATHER_OK : 0u) | chain[0] | (any_chain ? E_ANYCHAIN : 0u) | (all4 ? E_ALL4 : 0u));
    o[1] = make_uint4(n1[1], base[1], wword[1] | (slow ? DESC2_SLOW : 0u), (packed ? GATHER_OK : 0u) | chain[1]);
    if (any_chain && chained != nullptr) {
        // the most nodes one iteration of the walk can stage on this record: what the rings must have free when a loop is entered
        uint32_t most = 0;
        for (uint32_t a = 0; a < 2; a++) {
            const uint32_t first = 2u + (chain[a] ? chain_mids(n1[a], (wword[a] & REC_MASK) + ix.alphabet_offset) : 0u);
            for (uint32_t b = 0; b < 2; b++) {
                const uint4 l = leaf[2 * a + b];
                most = max(most, first + 2u + ((l.z & LEAF_CHAIN) ? chain_mids(l.x, (l.z & REC_MASK) + ix.alphabet_offset) : 0u));
            }
        }
        if (*chained < most) atomicMax(chained, most);
    }
    {   // records with a chained step, counted per wave (ten million of config 4's records have one: as many atomics on one address before)
        const uint64_t lanes = __ballot(any_chain && chained != nullptr);
        if (lanes != 0 && (threadIdx.x % WAVE) == static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(lanes)) - 1)) atomicAdd(chained + 1, static_cast<uint32_t>(__popcll(lanes)));
    }
    o[2] = leaf[0]; o[3] = leaf[1]; o[4] = leaf[2]; o[5] = leaf[3];
    o[6] = make_uint4(0u, 0u, 0u, 0u);
    o[7] = make_uint4(0u, 0u, 0u, 0u);
}

// The two-step blocks of every record with rank blocks, in both layouts.  Full width (32 bytes per 64 offsets):
//   cblocks[2 * k]     = {bits1 (values of v), bits2 (value in w_a of the sequence at each offset; 0 where the second
//                         step is not a real step through a record with blocks)}
//   cblocks[2 * k + 1] = {value-1 positions of v before the block, R_0, R_1, 0}
// The a-paths of a block land on consecutive offsets of w_a (LF keeps their order), so their values there are a
// contiguous bit range of w_a's blocks, spread back onto the positions of the a-paths.
// Packed (gblocks, may be null), for the loops where every load instruction costs a pass of the address path over sixty-four lines:
// ONE 16-byte load per step instead of two.  Half a block each -- 32 offsets -- with the three counts packed:
//   gblocks[2 * k + h] = {bits1 (32 values of v), bits2, ones1 | R_0 << 21, R_0 >> 11 | R_1 << 10}      (ones1, R_0 < 2^21, R_1 < 2^22)
// for the offsets 64 k + 32 h ...; ones1 and R_a count up to the half's first offset / first a-path.  Records that do not fit the
// counts have GATHER_OK cleared in their descriptor (k_link_desc2).
// A lane per BLOCK (the plain block says which record it belongs to): the blocks of a record are independent of each other once its plain
// rank blocks exist.  A lane per record (79 blocks one after the other for a record of 5 000 positions, every one behind two dependent
// loads) made these two arrays 12 of the 33 ms of kernel time of an open of the headline index; a wave per record still left two thirds
// of the waves (unary records) and a fifth of the lanes (79 = 64 + 15 blocks) without work.
struct SecondStep { const uint4 *wblocks[2]; uint32_t wbase[2]; };

// the values in w_a of the a-paths selected by m (an ascending subset of 64 / 32 positions), and the count R_a before the first of them
template <class Mask>
__device__ __forceinline__ Mask second_step_bits(const uint4 *wblocks, uint32_t j, Mask m, uint32_t &R) {
    const uint32_t q = j >> RANK_BLOCK_SHIFT, sh = j & 63u;
    const uint32_t cnt = sizeof(Mask) == 8 ? __popcll(m) : __popc(static_cast<uint32_t>(m));
    const uint4 W0 = wblocks[q];
    const uint64_t w0 = (static_cast<uint64_t>(W0.y) << 32) | W0.x;
    R = W0.z + __popcll(w0 & ((uint64_t(1) << sh) - 1));
    uint64_t val = w0 >> sh;
    if (sh != 0 && cnt > 64 - sh) {
        const uint4 W1 = wblocks[q + 1];
        val |= ((static_cast<uint64_t>(W1.y) << 32) | W1.x) << (64 - sh);
    }
    Mask bits2 = 0;
    while (m) {                                                                     // spread the low cnt bits of val over the set bits of m
        const Mask low = m & (~m + 1);
        if (val & 1) bits2 |= low;
        val >>= 1;
        m ^= low;
    }
    return bits2;
}

__global__ void __launch_bounds__(256) k_fill_two_step_blocks(DeviceIndex ix, uint4 *cblocks, uint4 *gblocks) {
    const uint64_t b = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x + 1;      // block 0 is the shared zero block
    if (b >= ix.n_blocks) return;
    const uint4 P = ix.blocks[b];
    const uint64_t v = P.w;                                                                   // the owner (k_fill_blocks)
    const uint4 *d1 = ix.desc;
    const uint4 D = d1[4 * v + 2];
    const uint32_t len = ix.desc_raw[4 * v + 1].w;
    const uint32_t bb = ix.block_base[v];
    const uint32_t k = static_cast<uint32_t>(b - bb);
    // per edge: landing record with blocks of its own, or none
    SecondStep s2{{nullptr, nullptr}, {0, 0}};
    if (!(D.x & DESC_SLOW)) {
        for (uint32_t a = 0; a < 2; a++) {
            // where the first step lands and with which offset base: the two-step descriptor's word (k_link_desc2 may have run the step
            // through a chain of unary records, further than the one-step descriptor goes)
            const uint4 E = ix.desc2[8 * v + a];
            const uint64_t w = E.z & REC_MASK;
            if (w == 0) continue;                                             // the walk ends behind this edge
            const uint4 WB = ix.desc_raw[4 * w + 1];
            if ((d1[4 * w + 2].x & DESC_SLOW) || WB.y == 0 || desc_class(WB.z) != 2) continue;
            s2.wblocks[a] = ix.blocks + ix.block_base[w];
            s2.wbase[a] = E.y;
        }
    }
    {
        const uint64_t bits1 = (static_cast<uint64_t>(P.y) << 32) | P.x;
        const uint32_t remaining = len - (k << RANK_BLOCK_SHIFT) > len ? 0u : len - (k << RANK_BLOCK_SHIFT);   // k * 64 <= len
        const uint64_t valid = remaining >= 64 ? ~uint64_t(0) : ((uint64_t(1) << remaining) - 1);
        uint64_t bits2 = 0, m_of[2] = {0, 0};
        uint32_t R[2] = {0, 0};
        for (uint32_t a = 0; a < 2; a++) {
            const uint64_t m = (a ? bits1 : ~bits1) & valid;
            if (!s2.wblocks[a] || m == 0) continue;
            m_of[a] = m;
            const uint32_t before = a ? P.z : (k << RANK_BLOCK_SHIFT) - P.z;          // a-paths of v before this block
            bits2 |= second_step_bits<uint64_t>(s2.wblocks[a], s2.wbase[a] + before, m, R[a]);
        }
        if (cblocks != nullptr) {
            cblocks[2 * static_cast<uint64_t>(bb + k)] = make_uint4(P.x, P.y, static_cast<uint32_t>(bits2), static_cast<uint32_t>(bits2 >> 32));
            cblocks[2 * static_cast<uint64_t>(bb + k) + 1] = make_uint4(P.z, R[0], R[1], 0u);
        }
        if (gblocks == nullptr) return;
        // the two halves of the same block: the bits are the same bits; the counts of the upper half start behind the lower half's
        // positions (ones1) and behind the a-paths of the lower half that have value 1 in w_a (R_a).  A half without a-paths (or an
        // edge without blocks behind it) carries R_a = 0, which nothing reads.
        const uint32_t lo2 = static_cast<uint32_t>(bits2), hi2 = static_cast<uint32_t>(bits2 >> 32);
        uint32_t Rlo[2], Rhi[2];
        for (uint32_t a = 0; a < 2; a++) {
            const uint32_t mlo = static_cast<uint32_t>(m_of[a]), mhi = static_cast<uint32_t>(m_of[a] >> 32);
            Rlo[a] = mlo ? R[a] : 0u;
            Rhi[a] = mhi ? R[a] + __popc(lo2 & mlo) : 0u;
        }
        const uint32_t ones_hi = P.z + __popc(P.x);
        gblocks[2 * static_cast<uint64_t>(bb + k)] = make_uint4(P.x, lo2, (P.z & 0x1FFFFFu) | (Rlo[0] << 21), ((Rlo[0] >> 11) & 0x3FFu) | (Rlo[1] << 10));
        gblocks[2 * static_cast<uint64_t>(bb + k) + 1] = make_uint4(P.y, hi2, (ones_hi & 0x1FFFFFu) | (Rhi[0] << 21), ((Rhi[0] >> 11) & 0x3FFu) | (Rhi[1] << 10));
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
        if (desc2[8 * r].z & DESC2_SLOW) { good = false; break; }
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
// block k = {64 values (one bit each), value-1 positions before the block, the record it belongs to}.  Record::lf (src/bwt.rs:480-496) at
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
                out[k++] = make_uint4(static_cast<uint32_t>(bits), static_cast<uint32_t>(bits >> 32), ones, static_cast<uint32_t>(rec));
                ones += __popcll(bits);
                bits = 0; fill = 0;
            }
        }
    }
    if (k < count) out[k] = make_uint4(static_cast<uint32_t>(bits), static_cast<uint32_t>(bits >> 32), ones, static_cast<uint32_t>(rec));
}

}  // namespace

void launch_build_desc(const DeviceIndex &ix, uint4 *d_desc, uint32_t *d_block_counts, uint64_t *d_stats, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_build_desc, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_desc, d_block_counts, d_stats);
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

void launch_fill_wtables(const DeviceIndex &ix, uint4 *d_wtables, hipStream_t stream) {
    if (ix.n_records == 0 || ix.n_records > 0x7FFFFFFFull) return;
    hipLaunchKernelGGL(k_fill_wtables, dim3(static_cast<unsigned>(ix.n_records)), dim3(256), 0, stream, ix, d_wtables);
}

void launch_fill_wtables_deep(const DeviceIndex &ix, uint4 *d_deep, bool compact, hipStream_t stream) {
    if (ix.n_records == 0 || ix.n_records > 0x7FFFFFFFull || ix.wtables == nullptr) return;
    hipLaunchKernelGGL(k_fill_wtables_deep, dim3(static_cast<unsigned>(ix.n_records)), dim3(256), 0, stream, ix, d_deep, compact ? 1u : 0u);
}

void launch_link_desc2(const DeviceIndex &ix, uint4 *d_desc2, uint32_t gather_limit, uint32_t chain_max, uint32_t *d_chained, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_link_desc2, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_desc2, gather_limit, chain_max, d_chained);
}

void launch_fill_two_step_blocks(const DeviceIndex &ix, uint4 *d_cblocks, uint4 *d_gblocks, hipStream_t stream) {
    if (ix.n_records == 0 || (d_cblocks == nullptr && d_gblocks == nullptr)) return;
    if (ix.n_blocks <= 1) return;
    hipLaunchKernelGGL(k_fill_two_step_blocks, dim3(grid_for(ix.n_blocks - 1, 256)), dim3(256), 0, stream, ix, d_cblocks, d_gblocks);
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
// ... and, for the records that have blocks (class 2), a copy in the raw descriptor's word C.z, which only class 0 records use otherwise
// (first LF table entry): a search step then finds the block base in the descriptor line it has fetched anyway, instead of behind one
// more dependent load from one more cache line (config 3: k_search 0.58 -> 0.49 ms for a million queries of ten nodes, k_bd_search 0.59 -> 0.51; NOTEBOOK.md round 4)
__global__ void __launch_bounds__(256) k_finish_block_base(const uint32_t *counts, uint32_t *block_base, uint64_t n, uint4 *desc_raw) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec >= n) return;
    const uint32_t bb = counts[rec] == 0 ? BLOCK_NONE : block_base[rec] + 1;
    block_base[rec] = bb;
    if (bb != BLOCK_NONE && desc_raw != nullptr) reinterpret_cast<uint32_t *>(desc_raw + 4 * rec + 2)[2] = bb;
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

void launch_finish_block_base(const uint32_t *d_counts, uint32_t *d_block_base, uint64_t n, uint4 *d_desc_raw, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_finish_block_base, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_counts, d_block_base, n, d_desc_raw);
}

// ---- the record starts from the Elias-Fano words of the file (kernels.hpp) ---------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ef_counts(const uint64_t *high, uint64_t n, uint64_t *counts) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i < n) counts[i] = static_cast<uint64_t>(__popcll(high[i]));
}

__global__ void __launch_bounds__(256) k_ef_values(const uint64_t *high, uint64_t n, const uint64_t *rank, const uint64_t *low, uint64_t low_words, uint32_t w, uint64_t ones,
                                                    uint64_t data_len, uint32_t *out32, uint64_t *out64) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i == 0) { if (out32) out32[ones] = static_cast<uint32_t>(data_len); else out64[ones] = data_len; }   // the sentinel
    if (i >= n) return;
    uint64_t word = high[i], k = rank[i];
    while (word != 0 && k < ones) {                                 // (a word count that disagrees with `ones` is the caller's to report: nothing is written past the table)
        const uint64_t pos = i * 64 + static_cast<uint64_t>(__ffsll(static_cast<unsigned long long>(word)) - 1);
        word &= word - 1;
        const uint64_t upper = pos - k, bit = k * w, lw = bit >> 6, off = bit & 63;
        uint64_t lo = lw < low_words ? low[lw] >> off : 0;
        if (off + w > 64 && lw + 1 < low_words) lo |= low[lw + 1] << (64 - off);
        if (w < 64) lo &= (uint64_t(1) << w) - 1;
        uint64_t value = (w >= 64 ? 0 : upper << w) | lo;
        if ((w < 64 && upper != 0 && (upper >> (64 - w)) != 0) || value > data_len) value = data_len;       // a value outside the data: the host's decode of the same words reports it (InvalidData)
        if (out32) out32[k] = static_cast<uint32_t>(value); else out64[k] = value;
        k++;
    }
}

__global__ void __launch_bounds__(256) k_starts_check(const uint32_t *s32, const uint64_t *s64, uint64_t ones, uint32_t *flags) {
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= ones) return;
    const uint64_t a = s32 ? s32[k] : s64[k], b = s32 ? s32[k + 1] : s64[k + 1];
    if (a > b) atomicOr(flags, 1u);
}

// GBZ::has_node (src/gbz.rs:286-289) for every potential node s (forward GBWT node 2 s + first, record 2 s + 1): where the record is empty or
// holds no edge the node does not exist and its label length becomes 0 (BWT::id_iter skips such records, src/bwt.rs:341-351)
__global__ void __launch_bounds__(256) k_mask_label_lengths(DeviceIndex ix, uint32_t *label_len, uint64_t n) {
    const uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (s >= n) return;
    const uint64_t rec = 2 * s + 1;
    bool real = false;
    if (rec < ix.n_records) {
        const uint64_t a = ix.starts32 ? ix.starts32[rec] : ix.starts64[rec], b = ix.starts32 ? ix.starts32[rec + 1] : ix.starts64[rec + 1];
        real = b > a && b <= ix.data_len && ix.data[a] != 0;
    }
    if (!real) label_len[s] = 0;
}

void launch_mask_label_lengths(const DeviceIndex &ix, uint32_t *d_label_len, uint64_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_mask_label_lengths, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, d_label_len, n);
}

void launch_ef_counts(const uint64_t *d_high, uint64_t high_words, uint64_t *d_counts, hipStream_t s) {
    if (high_words) hipLaunchKernelGGL(k_ef_counts, dim3(grid_for(high_words, 256)), dim3(256), 0, s, d_high, high_words, d_counts);
}

void launch_ef_values(const uint64_t *d_high, uint64_t high_words, const uint64_t *d_rank, const uint64_t *d_low, uint64_t low_words, uint32_t width, uint64_t ones,
                      uint64_t data_len, uint32_t *d_starts32, uint64_t *d_starts64, hipStream_t s) {
    hipLaunchKernelGGL(k_ef_values, dim3(grid_for(std::max<uint64_t>(high_words, 1), 256)), dim3(256), 0, s, d_high, high_words, d_rank, d_low, low_words, width, ones, data_len,
                       d_starts32, d_starts64);
}

void launch_starts_check(const uint32_t *d_starts32, const uint64_t *d_starts64, uint64_t ones, uint32_t *d_flags, hipStream_t s) {
    if (ones) hipLaunchKernelGGL(k_starts_check, dim3(grid_for(ones, 256)), dim3(256), 0, s, d_starts32, d_starts64, ones, d_flags);
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
