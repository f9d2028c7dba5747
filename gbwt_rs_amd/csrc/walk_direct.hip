// walk_direct.hip -- extraction with known sequence lengths: rows of the CSR written in place (k_walk_direct: segmented
// extraction from sequence samples, or one walker per end of every row), the cooperative row writes of the helper wave,
// and the walker order of a segmented extraction.  Hot loops in walk_loops.hpp; launch wrappers declared in kernels.hpp.
#include "kernels.hpp"

#include <hipcub/hipcub.hpp>

#include "walk_loops.hpp"

namespace gbwt_hip {

namespace {

// ---- extraction with known lengths ---------------------------------------------------------------------------
// With the lengths of the sequences known (device_index.hpp: seq_len) the CSR offsets exist before the walk starts, so
// the nodes go straight into their rows -- and in a bidirectional index every row is filled from BOTH ends at once:
// walker k walks sequence id from its start and writes front to back, walker n + k walks sequence id ^ 1 (the same
// path reversed and flipped) and writes back to front, flipping the nodes.  Each stops at the middle.
//
// The walking wave never stores to global memory here.  Row starts are megabytes apart, so 10 000 write streams miss
// the TLB all the time, and on gfx9 a store in flight delays every load behind it (one in-order vmcnt): with the walker
// storing, filling rows from both ends gained 1.2x instead of 2x on the headline index.  The nodes therefore stay in
// the LDS ring until the HELPER wave -- which already does the look-ahead touches and has a vmcnt of its own -- moves
// them to the row, 64 bytes at a time.  The walker publishes how many nodes it has staged (mailbox word 3), the
// helper publishes how many it has written (`drained`), and the walker only stalls when its ring is full.

// Where the nodes of one walker go.
struct RowTarget {
    uint32_t *row = nullptr;     // first node of the CSR row
    uint64_t len = 0;            // nodes in the row
    bool backward = false;       // this walker comes from the other end: node k goes to row[len - 1 - k], flipped
    uint32_t share = 0;          // nodes this walker has to deliver
};

// first node of row k in out_nodes (WalkArgs::uniform_len: rows of one length need no table)
__device__ __forceinline__ uint64_t row_offset(const WalkArgs &a, uint64_t k) { return a.uniform_len != 0 ? k * a.uniform_len : a.out_offsets[k]; }

__device__ __forceinline__ RowTarget row_target(const WalkArgs &a, uint64_t w) {
    RowTarget t;
    const uint64_t k = w < a.n ? w : w - a.n;
    t.backward = w >= a.n;
    t.len = row_offset(a, k + 1) - row_offset(a, k);
    t.row = a.out_nodes + row_offset(a, k);
    const uint64_t share = !a.both_ends ? t.len : (t.backward ? t.len / 2 : t.len - t.len / 2);
    t.share = static_cast<uint32_t>(share);
    return t;
}

// Segmented extraction (DeviceIndex::samples): a walker fills one segment of one row -- the nodes from sample j of the
// sequence up to sample j + 1 (or the end of the row).  Walkers are numbered segment by segment, within a segment over
// the rows that have it (rows sorted by their number of segments, stable: with rows of one length simply w = j * n + k),
// so the walkers of a wave hold the same segment of neighbouring rows and travel together like whole-sequence walkers
// do, and a batch with one long row and many short ones has as many walkers as it has segments, not rows x longest.
struct WalkerStart { uint32_t rec = 0, offset = 0, bb = BLOCK_NONE, first_node = 0; };

// PROBE (here and in k_walk_direct): the instantiation that still looks at the measurement switches (WalkArgs::debug); the product
// instantiation has none of them compiled in.  (Two frames that round 4 measured and did not keep -- segment boundaries aligned to the lines
// of the row's memory, walkers ordered by their start record -- lived here as knobs until round 6: NOTEBOOK.md, round 4.)
template <bool PROBE>
__device__ __forceinline__ WalkerStart segment_start(const DeviceIndex &ix, const WalkArgs &a, uint64_t w, RowTarget &t) {
    WalkerStart s;
    // walker w -> segment j = the level it falls into, row = the (w - level[j])-th of the rows that have a segment j
    uint32_t lo = 0, hi = a.level == nullptr ? 1u : a.segments;   // level[lo] <= w < level[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (a.level[mid] <= w) lo = mid; else hi = mid;
    }
    uint64_t j = lo, k = 0;
    if (a.level == nullptr) { j = w / a.n; k = w % a.n; }           // every row has every segment: w = j * n + k
    else k = a.sorted_rows[w - a.level[lo]];
    const uint64_t id = a.seq_ids[k];
    if (id >= ix.n_sequences) return s;                      // GBWT::sequence: no such sequence -> an empty row
    // (segment j of a row = from sample j * stride of its sequence to sample (j + 1) * stride: big batches skip samples, small ones use them all;
    // an extraction of one part of every row numbers its walkers over the segments lo .. hi - 1 of each row and writes the part as a row of its own)
    const RowSegments rs = row_segments(ix, id);
    const uint64_t base = rs.base, stride = rs.stride, count = rs.count;
    j += rs.lo;
    const bool parted = ix.sample_parts > 1;
    const uint64_t len = parted ? ix.seq_len[id] : row_offset(a, k + 1) - row_offset(a, k);
    if (j >= rs.hi) return s;                                 // this row has fewer segments: nothing to do
    const uint4 here = ix.samples[base + j * stride];
    const uint64_t from = j == 0 ? 0 : here.w;                // segment 0 starts with the start node (sample 0 is the state after it)
    const uint64_t to = j + 1 < count ? ix.samples[base + (j + 1) * stride].w : len;
    t.row = a.out_nodes + row_offset(a, k) + (from - (parted ? segment_position(ix, rs, id, rs.lo) : 0u));
    t.len = to > from ? to - from : 0;
    if (PROBE && (a.debug & 2u)) t.row = a.out_nodes + (w % 4096u) * 4096u;   // measurement switch: all rows land in one 64 MB window (wrong output)
    if (PROBE && (a.debug & 128u)) t.row = a.out_nodes + (w % 64u) * 4096u;   //                     ... in 1 MB (stays in every L2)
    t.backward = false;
    t.share = static_cast<uint32_t>(t.len);
    s.rec = here.x; s.offset = here.y; s.bb = here.z;
    if (j == 0 && id < ix.n_endmarker) s.first_node = ix.endmarker[id].x;
    return s;
}

// LDS through pointers that say so.  A `volatile uint32_t *` into __shared__ memory is a generic pointer: hipcc turns
// every access into flat_load / flat_store sc0 sc1, which travel through the vector-memory path (address coalescer,
// vmcnt AND lgkmcnt) like a global access -- the helper's polling and its sixteen ring reads per 64 bytes written were
// competing with the walk's own loads for the same unit.  The low 32 bits of a flat LDS address are the LDS offset.
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lds_u32_t *lds_ptr(const void *p) { return (lds_u32_t *)static_cast<uintptr_t>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p))); }
__device__ __forceinline__ uint32_t lds_peek(const lds_u32_t *p) { return *const_cast<const volatile lds_u32_t *>(p); }
__device__ __forceinline__ void lds_poke(lds_u32_t *p, uint32_t v) { *const_cast<volatile lds_u32_t *>(p) = v; }
__device__ __forceinline__ u32x4_t lds_peek4(const lds_u32_t *p) {   // one ds_read_b128 (16-byte aligned)
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// k_walk_direct's ring: slot s of lane l at dword s * RING_PITCH + l.  The pitch is 65, not 64, so that the slots of one
// lane fall into different LDS banks: the cooperative row writes read several slots of the same lane in one
// instruction (with a pitch of 64 they were all in one bank: 60 % of the LDS cycles of the kernel were bank conflicts).
constexpr uint32_t RING_PITCH = WAVE + 1;
static_assert(RING_PITCH == 65, "walk_loops.hpp: the four-in-a-row staging of the uniform loop (E_ALL4) writes at byte offsets k * 4 * 65");

// Staging only: the walking wave's side of the ring.
struct StageSink {
    lds_u32_t *stage;
    uint32_t wr = 0, mask;
    __device__ __forceinline__ StageSink(uint32_t *lds, uint32_t lane, uint32_t ring_mask) : stage(lds_ptr(lds + lane)), mask(ring_mask) {}
    __device__ __forceinline__ void push(uint32_t node, bool counts) {
        stage[(wr & mask) * RING_PITCH] = node;
        wr += counts ? 1u : 0u;
    }
};

// The helper's side: moves staged nodes [drained, upto) of one lane's ring column to the row.
struct RowWriter {
    const lds_u32_t *stage;      // not volatile: the caller puts a compiler barrier between polls, the reads of one piece can then go out together
    RowTarget t;
    uint32_t drained = 0, mask = RING2 - 1;
    bool dry = false;            // measurement switch (GBWT_HIP_DEBUG_DRY_ROWS): read the ring, store nothing
    __device__ __forceinline__ uint32_t slot(uint32_t k) const { return stage[(k & mask) * RING_PITCH]; }
    __device__ __forceinline__ void put(uint32_t k) {
        if (k >= t.len || dry) return;   // k >= len cannot happen in a consistent index; never write outside the row
        if (t.backward) t.row[t.len - 1 - k] = slot(k) ^ 1u; else t.row[k] = slot(k);
    }
    __device__ __forceinline__ void chunk() {   // 16 nodes = 64 bytes
        const uint32_t c = drained;
        if (static_cast<uint64_t>(c) + RING_FLUSH <= t.len) {
            uint32_t v[RING_FLUSH];
#pragma unroll
            for (uint32_t i = 0; i < RING_FLUSH; i++) v[i] = slot(c + i);
            if (dry) { uint32_t x = 0; for (uint32_t i = 0; i < RING_FLUSH; i++) x ^= v[i]; asm volatile("" :: "v"(x)); drained += RING_FLUSH; return; }
            uint32_t *dst = t.backward ? t.row + (t.len - c - RING_FLUSH) : t.row + c;
            const bool aligned = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
            if (!t.backward) {
                if (aligned) {
#pragma unroll
                    for (uint32_t q = 0; q < RING_FLUSH / 4; q++) reinterpret_cast<uint4 *>(dst)[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                } else {
#pragma unroll
                    for (uint32_t i = 0; i < RING_FLUSH; i++) dst[i] = v[i];
                }
            } else {   // node c + i goes to dst[15 - i]
                if (aligned) {
#pragma unroll
                    for (uint32_t q = 0; q < RING_FLUSH / 4; q++)
                        reinterpret_cast<uint4 *>(dst)[q] = make_uint4(v[15 - 4 * q] ^ 1u, v[14 - 4 * q] ^ 1u, v[13 - 4 * q] ^ 1u, v[12 - 4 * q] ^ 1u);
                } else {
#pragma unroll
                    for (uint32_t i = 0; i < RING_FLUSH; i++) dst[RING_FLUSH - 1 - i] = v[i] ^ 1u;
                }
            }
        } else {
            for (uint32_t i = 0; i < RING_FLUSH; i++) put(c + i);
        }
        drained += RING_FLUSH;
    }
    // Everything that is staged and can go out in whole 64-byte pieces.  Segments start at arbitrary node counts, so a
    // front-to-back writer first brings itself to a 64-byte boundary of the row with single stores; from there on every
    // piece is one aligned cache-line half (unaligned pieces would go out as sixteen 4-byte stores each and reach HBM as
    // partial lines: 22.7 GB written for 13.3 GB of node ids before this).
    __device__ __forceinline__ void drain(uint32_t staged) {
        if (!t.backward) {
            const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(t.row + drained) >> 2) & (RING_FLUSH - 1));
            if (mis != 0) {
                const uint32_t need = RING_FLUSH - mis;
                if (staged - drained < need) return;
                for (uint32_t i = 0; i < need; i++) put(drained + i);
                drained += need;
            }
        }
        while (staged - drained >= RING_FLUSH) chunk();
    }
};

// Cooperative row writes (segmented extraction: front-to-back rows only).  LPR lanes share one row: each moves four
// nodes of a piece of 4 * LPR nodes, so one store instruction writes WAVE / LPR whole pieces of 16 * LPR contiguous,
// aligned bytes -- the memory system sees one request per piece instead of one 16-byte request per lane (with every
// lane writing its own row, a wave's store touched 64 different cache lines with 16 bytes each, four times in a row).
// A row first brings itself to a piece boundary with single stores (segments start anywhere), the tail goes out the
// same way once the walk is over.  Row state lives in LDS: `staged` in the mailboxes (word 3), and one uint4 per row
// {address low, address high, length, drained} in row_state -- the helper writes all of it, the walker reads `drained`.
// A visit costs two LDS round trips (state + count, then the four nodes) and about two dozen VALU instructions: the
// helper's instructions compete with the walkers' for the same SIMDs (profiles/r01_final_pmc_headline.txt: VALU busy
// 60 % of the kernel, 134 VALU instructions per walker iteration of which the walker's own are 83).
struct CoopRows {
    uint32_t ring;               // LDS byte address of the ring: slot * RING_PITCH + lane (dwords)
    uint32_t mail;               //                  of mailbox[0]; staged count of row r = word 4 r + 3
    uint32_t state;              //                  of row_state[0]
    uint32_t mask;
    bool dry;
    bool plain_stores;           // row pieces as ordinary stores instead of non-temporal ones (measurement switch)
    bool skip_reads;             // measurement switch: nothing is read from the ring or stored
    bool pipelined;              // coop_drain fetches the next group's state under this group's nodes (measurement switch 256 of WalkArgs::debug: off)
};
__device__ __forceinline__ uint32_t lds_word(uint32_t byte_address) { return *(const volatile lds_u32_t *)static_cast<uintptr_t>(byte_address); }

// The lanes that serve row r (lane p of LPR): one piece, or what the rules above allow instead.
template <uint32_t LPR>
__device__ __forceinline__ void coop_visit(const CoopRows &c, uint32_t r, uint32_t p, uint32_t done) {
    constexpr uint32_t PIECE = 4 * LPR;
    const uint32_t staged = lds_word(c.mail + 16 * r + 12);
    const u32x4_t st = lds_peek4((const lds_u32_t *)static_cast<uintptr_t>(c.state + 16 * r));   // waits for both
    const uint32_t drained = st.w, len = st.z;
    const uint32_t pend = staged - drained;
    const uint32_t mis = ((st.x >> 2) + drained) & (PIECE - 1);      // nodes past the last piece boundary of the row's memory
    uint32_t n = PIECE - mis;                                        // nodes up to the next boundary
    if (pend < n) { if (!done || pend == 0) return; n = pend; }      // short pieces only once the walk is over
    volatile lds_u32_t *const publish = (volatile lds_u32_t *)static_cast<uintptr_t>(c.state + 16 * r + 12);   // only after the nodes have left the ring
    if (c.skip_reads) { if (p == 0) *publish = drained + n; return; }   // measurement switch: the ring is emptied unread
    // a pointer rebuilt from integers is a generic one: say that it is global memory, or the stores become flat_store
    // (which also count in lgkmcnt, so that every LDS wait of the helper would wait for its row writes as well)
    typedef __attribute__((address_space(1))) uint32_t global_u32_t;
    typedef __attribute__((address_space(1))) u32x4_t global_u32x4_t;
    global_u32_t *dst = (global_u32_t *)((static_cast<uint64_t>(st.y) << 32) | st.x) + drained;
    const uint32_t col = c.ring + 4 * r;                             // slot s of the row at col + s * 4 * RING_PITCH (24-bit multiply: LDS is small)
    if (n == PIECE && drained + PIECE <= len) {
        const uint32_t k = drained + 4 * p;
        const uint32_t v0 = lds_word(col + __umul24((k + 0) & c.mask, 4 * RING_PITCH)), v1 = lds_word(col + __umul24((k + 1) & c.mask, 4 * RING_PITCH)),
                       v2 = lds_word(col + __umul24((k + 2) & c.mask, 4 * RING_PITCH)), v3 = lds_word(col + __umul24((k + 3) & c.mask, 4 * RING_PITCH));
        if (!c.dry) {
            u32x4_t v; v.x = v0; v.y = v1; v.z = v2; v.w = v3;
            global_u32x4_t *at = (global_u32x4_t *)dst + p;
            // Rows are written once and never read by this kernel: as plain stores they fill the L2s with dirty lines whose
            // write-back gets in the way of the walk's own traffic (6.9 ms per headline pass; 4.1 ms when all rows are
            // aimed at one megabyte that never leaves the L2s).  Non-temporal stores stream out: 5.1 ms.  Measured with
            // every sc0 / sc1 / nt combination: nt and nt sc0 are equal, nt sc0 sc1 is halfway, the others change nothing.
            if (c.plain_stores) *at = v;
            else asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(at), "v"(v) : "memory");
        }
        else asm volatile("" :: "v"(v0 ^ v1 ^ v2 ^ v3));
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t k = p + LPR * i;
            if (k < n && drained + k < len && !c.dry) dst[k] = lds_word(col + __umul24((drained + k) & c.mask, 4 * RING_PITCH));   // never outside the row
        }
    }
    if (p == 0) *publish = drained + n;
}

// One round over the rows that have something to write.  Lane l looks at row `mine` = order[l] to find them; rows[g] =
// the row this lane serves in group g (both from the helper's sort of the rows by address phase); returns the rows that
// still hold staged nodes afterwards (as seen before the round).
//
// Rows are grouped by the phase of their addresses: walkers that travel together stage nodes at the same rate, so rows
// whose memory has the same offset within a piece complete their pieces in the same iteration and one store
// instruction then carries WAVE / LPR full pieces.  Grouped by row number, the rows of a group had eight different
// phases, became ready one or two at a time, and the kernel issued 2.2 store instructions per kilobyte.
template <uint32_t LPR>
__device__ __forceinline__ uint64_t coop_drain(const CoopRows &c, uint32_t lane, uint32_t mine, const uint32_t (&rows)[8], uint32_t done) {
    constexpr uint32_t PIECE = 4 * LPR, ROWS = WAVE / LPR;
    const uint32_t staged = lds_word(c.mail + 16 * mine + 12);
    const u32x4_t st = lds_peek4((const lds_u32_t *)static_cast<uintptr_t>(c.state + 16 * mine));
    const uint32_t pend = staged - st.w;
    const uint32_t mis = ((st.x >> 2) + st.w) & (PIECE - 1);
    const uint64_t todo = __ballot(pend >= PIECE - mis || (done && pend != 0));
    const uint64_t left = __ballot(pend != 0);
    if (todo != 0 && c.pipelined && !done && !c.dry && !c.skip_reads && !c.plain_stores) {
        // The groups one after the other as below, but the count and state of the NEXT group are fetched while this group's nodes are on
        // their way out of the ring: a visit is two dependent LDS round trips, and sixteen of them in a row are what a round of the helper
        // takes -- the helper's lag is what the walkers' rings fill up behind (section 3 item 23).  Plain loads, so that the compiler can
        // count them down (lgkmcnt) instead of waiting for all; rows that are not at a piece boundary take the visit.
        typedef __attribute__((address_space(3))) u32x4_t lds_u32x4_t;
        typedef __attribute__((address_space(1))) uint32_t global_u32_t;
        typedef __attribute__((address_space(1))) u32x4_t global_u32x4_t;
        const uint32_t p = lane % LPR;
        uint32_t staged_next = 0;
        u32x4_t st_next = {0, 0, 0, 0};
        const bool on0 = (todo & ((uint64_t(1) << ROWS) - 1)) != 0;
        if (on0) {
            staged_next = *(const lds_u32_t *)static_cast<uintptr_t>(c.mail + 16 * rows[0] + 12);
            st_next = *(const lds_u32x4_t *)static_cast<uintptr_t>(c.state + 16 * rows[0]);
        }
#pragma unroll
        for (uint32_t g = 0; g < LPR; g++) {
            const bool on = ((todo >> (g * ROWS)) & ((uint64_t(1) << ROWS) - 1)) != 0;                       // wave-uniform
            const uint32_t staged = staged_next;
            const u32x4_t st = st_next;
            if (g + 1 < LPR && ((todo >> ((g + 1) * ROWS)) & ((uint64_t(1) << ROWS) - 1)) != 0) {
                staged_next = *(const lds_u32_t *)static_cast<uintptr_t>(c.mail + 16 * rows[g + 1] + 12);
                st_next = *(const lds_u32x4_t *)static_cast<uintptr_t>(c.state + 16 * rows[g + 1]);
            }
            if (!on) continue;
            const uint32_t drained = st.w, waiting = staged - drained;
            const uint32_t off = ((st.x >> 2) + drained) & (PIECE - 1);
            const bool ready = waiting >= PIECE - off;
            const bool fast = ready && off == 0 && drained + PIECE <= st.z;
            if (fast) {
                const uint32_t col = c.ring + 4 * rows[g], k = drained + 4 * p;
                u32x4_t out;
                out.x = *(const lds_u32_t *)static_cast<uintptr_t>(col + __umul24((k + 0) & c.mask, 4 * RING_PITCH));
                out.y = *(const lds_u32_t *)static_cast<uintptr_t>(col + __umul24((k + 1) & c.mask, 4 * RING_PITCH));
                out.z = *(const lds_u32_t *)static_cast<uintptr_t>(col + __umul24((k + 2) & c.mask, 4 * RING_PITCH));
                out.w = *(const lds_u32_t *)static_cast<uintptr_t>(col + __umul24((k + 3) & c.mask, 4 * RING_PITCH));
                global_u32_t *dst = (global_u32_t *)((static_cast<uint64_t>(st.y) << 32) | st.x) + drained;
                global_u32x4_t *at = (global_u32x4_t *)dst + p;
                asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(at), "v"(out) : "memory");
                if (p == 0) *(volatile lds_u32_t *)static_cast<uintptr_t>(c.state + 16 * rows[g] + 12) = drained + PIECE;   // only after the nodes have left the ring
            } else if (ready) {
                coop_visit<LPR>(c, rows[g], p, done);
            }
        }
        return left;
    }
    if (todo != 0) {
#pragma unroll
        for (uint32_t g = 0; g < LPR; g++)                           // wave-uniform tests; rows[g] stays in a register
            if (((todo >> (g * ROWS)) & ((uint64_t(1) << ROWS) - 1)) != 0) coop_visit<LPR>(c, rows[g], lane % LPR, done);
    }
    return left;
}

// One look-ahead touch from compiler-scheduled code: an LDS-direct load has no register destination, so nothing can be
// corrupted by the data arriving late, and nobody ever waits for it.  `lds_dummy` = wave-uniform LDS byte address of a
// 256-byte scratch area.
__device__ __forceinline__ void touch_line(const void *p, uint32_t lds_dummy) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(p), "s"(lds_dummy) : "memory");
}

// Register budget: the two assembly loops live in v40-v87 (walk_loops.hpp) and the compiler takes what it likes of the rest: 128
// VGPRs = four waves per SIMD = eight workgroups per CU, which is also what the LDS allows (64 ring slots x 65 lanes x 4 bytes + 2.5
// KB per workgroup).  -DGBWT_HIP_WALK_WAVES=5 fits the kernel into 96 VGPRs (3 spilled outside the loops); it changes nothing while
// the rings are this size, and smaller rings cannot hold a 128-byte row piece (profiles/r02_walk_bounds.txt #17).
template <bool PROBE>
#ifdef GBWT_HIP_WALK_WAVES
__attribute__((amdgpu_waves_per_eu(GBWT_HIP_WALK_WAVES, GBWT_HIP_WALK_WAVES)))
#endif
__global__ void __launch_bounds__(2 * WAVE) k_walk_direct(DeviceIndex ix, WalkArgs a) {
    const uint32_t debug = PROBE ? a.debug : 0u;             // measurement switches: none in the product instantiation
    // rows sized AFTER the launch (WalkArgs::capacity): the request did not wait for the total of its row lengths; where the rows it was
    // given are too small for it -- or there is nothing to walk -- every workgroup goes home and the host launches again
    if (a.capacity != 0) {
        const uint64_t total = a.out_offsets[a.n];
        if (total == 0 || total > a.capacity) return;
    }
    if (a.fill_offsets != nullptr)            // rows of one length: the offsets the caller reads, written here (WalkArgs::uniform_len)
        for (uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; k <= a.n; k += static_cast<uint64_t>(gridDim.x) * blockDim.x)
            a.fill_offsets[k] = k * a.uniform_len;
    extern __shared__ uint32_t ring_lds[];   // a.ring_slots * RING_PITCH entries (dynamic: the ring size sets how many workgroups fit a CU)
    __shared__ uint4 mailbox[WAVE];          // per walking lane: {look-ahead record, first block, blocks, nodes staged so far}
    __shared__ uint4 row_state[WAVE];        // per walking lane: {row address low, high, length (cooperative row writes), nodes the helper has moved to the row}
    __shared__ uint32_t touch_dummy[WAVE];
    __shared__ uint32_t row_order[WAVE];     // cooperative row writes: the rows sorted by address phase (helper's own table)
    __shared__ uint32_t mail_done;
    const uint32_t lane = threadIdx.x % WAVE;
    const bool helper = __builtin_amdgcn_readfirstlane(threadIdx.x) >= WAVE;
    if (!helper) {
        mailbox[lane] = make_uint4(0, 0, 0, 0);
        row_state[lane] = make_uint4(0, 0, 0, 0);
        if (lane == 0) mail_done = 0;
    }
    __syncthreads();
    const uint64_t walkers = a.segments ? a.walkers : (a.both_ends ? 2 * a.n : a.n);
    // Workgroup i runs on XCD i % 8 (round-robin dispatch), every XCD has an L2 of its own, and the walkers that pass
    // through the same records at the same time are neighbours in w (the same segment of neighbouring rows).  With
    // xcd_map the grid is a multiple of 8 and XCD x takes the x-th eighth of the walkers, in order, so that a record is
    // fetched into ONE L2 instead of all eight.
    uint64_t group = blockIdx.x;
    if (a.xcd_map) group = (blockIdx.x % 8u) * static_cast<uint64_t>(gridDim.x / 8u) + blockIdx.x / 8u;
    if (PROBE && (debug & 16384u)) {   // measurement switch (tools/occupancy_probe.py): the same work on XCDs 0-3 only -- the grid is twice as large, XCDs 4-7 leave at once
        if (blockIdx.x % 8u >= 4u) return;
        group = (blockIdx.x / 8u) * 4ull + blockIdx.x % 8u;
    }
    const uint64_t w = group * a.paths_per_wave + lane;
    const bool owner = lane < a.paths_per_wave && w < walkers;
    const uint32_t ring_mask = a.ring_slots - 1;
    RowTarget target;
    WalkerStart begin;
    if (owner) {
        if (a.segments) begin = segment_start<PROBE>(ix, a, w, target);
        else target = row_target(a, w);
    }
    lds_u32_t *const my_mail = lds_ptr(&mailbox[lane]);          // word 3 = nodes staged so far
    lds_u32_t *const my_drained = lds_ptr(&row_state[lane].w);
    lds_u32_t *const done_flag = lds_ptr(&mail_done);

    if (helper) {
        // ---- helper wave: look-ahead touches for every slot, row writes for its own lane's column
        const uint32_t owners = a.paths_per_wave ? a.paths_per_wave : WAVE;
        const uint32_t serve = lane % owners;                        // the 64 lanes share the owners' look-ahead slots ...
        const uint32_t spread = (lane << 26) | (1u << 25);           // ... and spread over the target's blocks
        const uint32_t dummy = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(touch_dummy));
        RowWriter writer{lds_ptr(ring_lds + lane), target, 0, ring_mask, (debug & 1u) != 0};
        const lds_u32_t *const served_mail = lds_ptr(&mailbox[serve]);
        const uint32_t piece = a.segments ? a.row_piece : 0u;       // rows filled back to front stay with the lane-per-row writer
        const auto lds_address = [](const void *q) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(q)); };
        const CoopRows rows{lds_address(ring_lds), lds_address(mailbox), lds_address(row_state), ring_mask, (debug & 1u) != 0, (debug & 4u) != 0,
                            (debug & 64u) != 0, (debug & 256u) == 0};
        uint32_t mine = lane;                                        // the row this lane watches
        uint32_t served[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // the row this lane serves in group g
        if (piece) {
            const uint64_t at = reinterpret_cast<uintptr_t>(target.row);
            lds_poke(lds_ptr(&row_state[lane].x), static_cast<uint32_t>(at));
            lds_poke(lds_ptr(&row_state[lane].y), static_cast<uint32_t>(at >> 32));
            lds_poke(lds_ptr(&row_state[lane].z), static_cast<uint32_t>(std::min<uint64_t>(target.len, 0x7FFFFFF0u)));
            // order = the rows sorted by (address phase within a piece, row): rank by counting, once per workgroup
            // (by ballots over the 16 / 32 possible phases: the loop over the 64 rows' LDS entries this replaces was a quarter of what a
            // workgroup spends outside its walk -- short segments, i.e. small batches and the per-rank shards of a multi-GPU run, pay that
            // per 300 nodes; tools/shard_probe.py)
            const uint32_t phase = (static_cast<uint32_t>(at) >> 2) & (piece - 1);
            const uint64_t lanes_below = (uint64_t(1) << lane) - 1;
            uint32_t rank = 0;
            for (uint32_t v = 0; v < piece; v++) {
                const uint64_t same = __ballot(phase == v);
                rank += v < phase ? static_cast<uint32_t>(__popcll(same)) : (v == phase ? static_cast<uint32_t>(__popcll(same & lanes_below)) : 0u);
            }
            if (PROBE && (debug & 32u)) rank = lane;                          // measurement switch: groups of consecutive rows
            lds_poke(lds_ptr(row_order) + rank, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mine = lds_peek(lds_ptr(row_order) + lane);
            const uint32_t lanes_per_row = piece / 4, rows_per_group = WAVE / lanes_per_row;
#pragma unroll
            for (uint32_t g = 0; g < 8; g++) served[g] = g < lanes_per_row ? lds_peek(lds_ptr(row_order) + g * rows_per_group + lane / lanes_per_row) : 0u;
        }
        uint32_t seen = 0;
        for (;;) {
            asm volatile("" ::: "memory");                           // the ring and the mailboxes have changed since the last poll
            const uint32_t done = lds_peek(done_flag);               // read before the counts: the final count is then complete
            const u32x4_t mail = lds_peek4(served_mail);
            const uint32_t look_rec = mail.x, look_base = mail.y, look_count = mail.z, stamp = mail.w;
            // MIXED WAVES ON RECORDS THAT FEW ROWS PASS (WalkArgs::gather_reach, round 6).  The helper looks at a lane's mailbox once per
            // four or five iterations of its walker, so a touch of the target alone covers a quarter of the records the lane will visit --
            // enough where seventy other waves pass the same records, nothing where one or two do (config 4: ~50 positions per record).
            // There ONE lane per distinct target (neighbouring lanes hold rows of the same graph component: a touch per lane was 64 address
            // passes for two lines) touches the descriptor lines of the target and of the records BEHIND it (ids along a walk are close
            // to consecutive in a graph whose ids are sorted topologically, as vg's are) and the block lines behind its first block:
            // what the walkers need until the helper looks again.
            const uint32_t before = static_cast<uint32_t>(__shfl_up(static_cast<int>(look_rec), 1));
            const bool mixed_targets = a.gather_reach != 0 && __ballot(look_rec != static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(look_rec)))) != 0;
            if (mixed_targets) {
                if (look_rec != 0 && stamp != seen && (lane == 0 || look_rec != before)) {
                    const uint4 *const blocks = (look_count & 0x80000000u) ? ix.cblocks : ix.gblocks;
                    for (uint32_t r = 0; r < a.gather_reach; r++) {
                        const uint64_t next = min(static_cast<uint64_t>(look_rec) + r, ix.n_records - 1);
                        touch_line(ix.desc2 + 8 * next, dummy);
                        touch_line(ix.desc2 + 8 * next + 4, dummy);
                        if ((r & 1u) == 0 && (look_count & 0x7FFFFFFFu) != 0)
                            touch_line(blocks + min(2 * static_cast<uint64_t>(look_base) + 2 * r, 2 * ix.n_blocks - 1), dummy);   // (64-byte lines: two blocks of 64 offsets each)
                    }
                }
            } else if (lane < a.helper_lanes && look_rec != 0 && stamp != seen) {
                const uint4 *d = ix.desc2 + 8 * static_cast<uint64_t>(look_rec);
                touch_line(d, dummy);
                touch_line(d + 4, dummy);
                // the two block arrays have the same geometry (32 bytes per 64 offsets); bit 31 of the count: the walker is on the full-width one
                const uint4 *const blocks = (look_count & 0x80000000u) ? ix.cblocks : ix.gblocks;
                touch_line(blocks + 2 * (static_cast<uint64_t>(look_base) + __umulhi(spread, look_count & 0x7FFFFFFFu)), dummy);
            }
            seen = stamp;
            if (piece) {
                const uint64_t left = piece == 32 ? coop_drain<8>(rows, lane, mine, served, done) : coop_drain<4>(rows, lane, mine, served, done);
                if (done && left == 0) break;
                if (done) continue;
            } else if (owner) {
                const uint32_t staged = lds_peek(my_mail + 3);
                asm volatile("" ::: "memory");                       // ring reads stay behind the count
                writer.drain(staged);
                if (done) { for (uint32_t k = writer.drained; k < staged; k++) writer.put(k); }
                lds_poke(my_drained, writer.drained);
            }
            if (done && !piece) break;
            for (uint32_t nap = 0; nap < a.helper_naps; nap++) __builtin_amdgcn_s_sleep(4);
        }
        return;
    }

    // ---- walking wave
    const uint32_t mail_slot = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&mailbox[lane]));
    StageSink sink(ring_lds, lane, ring_mask);
    uint32_t rec = 0, offset = 0, bb = BLOCK_NONE;
    const uint32_t quota = target.share;
    if (owner && a.segments) {
        if (quota > 0) {
            if (begin.first_node != 0) sink.push(begin.first_node, true);   // segment 0 also delivers the start node
            if (sink.wr < quota) { rec = begin.rec; offset = begin.offset; bb = begin.bb; }
        }
    } else if (owner) {
        const uint64_t k = w < a.n ? w : w - a.n;
        const uint64_t id = a.seq_ids[k] ^ (target.backward ? 1u : 0u);
        if (quota > 0 && id < ix.n_endmarker && id < ix.n_sequences) {  // GBWT::start, src/gbwt.rs:213-219
            const uint2 e = ix.endmarker[id];
            if (e.x != 0) {
                sink.push(e.x, true);
                offset = e.y;
                if (quota <= 1 || !arrive(ix, e.x, e.y, rec, bb)) { rec = 0; bb = BLOCK_NONE; }
            }
        }
    }
    lds_poke(my_mail + 3, sink.wr);
    const uint32_t ring_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ring_lds + lane));
    const bool narrow = !a.wide_addresses && ix.n_records * 128 <= 0xFFFFFFFFull && ix.n_blocks * 32 <= 0xFFFFFFFFull;
#ifdef GBWT_HIP_PROBE_LOOP_SHARE   // measurement only: nodes staged by each of the two loops, for a few workgroups (profiles/r02_walk_bounds.txt #16)
    uint32_t probe_uniform = 0, probe_vector = 0, probe_entries = 0, probe_mixed_entries = 0, probe_fell_out = 0;
#endif
    // Free ring slots a loop needs when it is entered, and leaves below.  A loop looks at the ring AFTER an iteration, so whatever follows it
    // (the other loop, a table or generic step) may start with one iteration's nodes fewer.  Without chained steps an iteration stages four
    // nodes and the headroom of eight covers two of them.  With chained steps an iteration stages up to ix.chained nodes (5 .. 16): the
    // headroom covers ONE iteration and what follows a loop looks at the ring first (`recheck`) -- usable slots are worth more than the
    // look: 48 of 64 instead of 56 cost the lock-step chain a tenth of its speed, and the insertion chain walks at 490 G LF-steps/s with a
    // headroom of 8 and the look, at 460 G with a headroom of 14 without it.
    const uint32_t most = ix.chained != 0 ? ix.chained : 4u;               // nodes an iteration can stage
    bool recheck = ix.chained != 0;
    uint32_t headroom = max(8u, ix.chained);
    if (a.headroom != 0) { headroom = max(a.headroom, most); recheck = recheck || headroom < 2 * most; }   // GBWT_HIP_HEADROOM (measurements)
    bool full_blocks = a.packed_blocks == 0;    // wave-uniform
    uint32_t catch_credit = 2, catch_pause = 0, catch_backoff = 8;   // wave-uniform: see CATCH-UP below
    while (__ballot(rec != 0) != 0) {
        const uint32_t drained = lds_peek(my_drained);
        if (__ballot(sink.wr - drained > ring_mask + 1 - headroom) != 0) { __builtin_amdgcn_s_sleep(2); continue; }   // ring full: let the helper catch up
        // all lanes on one record: scalar descriptor fetch; otherwise every lane fetches its own
        // (the uniform loop finds out by itself, but only after it has issued a round of loads for nothing: in graphs whose rows do
        // not move in lock-step the waves are mixed at almost every entry)
        bool together = __ballot(rec != static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(rec))) == 0;
        bool caught = false;
        // CATCH-UP.  A wave that has left lock step is often only a step apart: the rows that took an insertion are one record behind
        // the others.  Any lane may take an LF step at any time -- the order of the steps changes nothing in what is emitted -- so the
        // lanes that are BEHIND take single steps (one-step descriptors + plain rank blocks, plain C++) until the wave stands on one
        // record again and the uniform loop can have it back.  "Behind" is a guess from the orientation of the node the wave is on
        // (forward nodes are walked in ascending id order in a graph whose ids are sorted topologically, as vg's are); where the guess
        // is wrong, or the rows really have gone different ways, the attempt fails after four steps, the gather loop takes over as
        // before, and the wave stops trying for a while.
        if (!together && a.uniform_loop && a.catch_up && catch_pause == 0) {
            // ... and only a wave whose rows are NEAR each other: lanes on records thousands apart are rows of different graph components (a
            // batch of ragged walks over hundreds of components: config 4's shape), which no number of single steps brings together --
            // such a wave keeps to the gather loop for a long while (round 4: 2.08 -> 1.9 ms on that batch with catch-up switched off)
            uint32_t lo_rec = rec != 0 ? rec : 0xFFFFFFFFu, hi_rec = rec;
            for (int d = 32; d > 0; d >>= 1) {
                lo_rec = min(lo_rec, static_cast<uint32_t>(__shfl_xor(static_cast<int>(lo_rec), d)));
                hi_rec = max(hi_rec, static_cast<uint32_t>(__shfl_xor(static_cast<int>(hi_rec), d)));
            }
            if (hi_rec - lo_rec > 64u) catch_pause = 256;
        }
        // (catch_up == 2: the single steps read the two-step descriptors and the packed half-blocks the loops read anyway -- a handle that has
        // given its one-step descriptors and plain rank blocks back, capi_open.hip: open_common -- and only while the wave is on the packed blocks)
        if (!together && a.uniform_loop && a.catch_up && !(a.catch_up == 2u && full_blocks) && catch_pause == 0 && __ballot(sink.wr - drained > ring_mask + 1 - 8) == 0) {   // (its single steps stage up to eight nodes)
            const uint64_t walking = __ballot(rec != 0);
            for (uint32_t tries = 0; tries < 4 && !together; tries++) {
                const uint32_t any = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(rec), __builtin_ctzll(walking)));
                const bool ascending = ((any + ix.alphabet_offset) & 1u) == 0;
                // the record furthest ahead among the walking lanes
                uint32_t key = rec != 0 ? (ascending ? rec : ~rec) : 0u;
                uint32_t front = key;
                for (int d = 32; d > 0; d >>= 1) front = max(front, static_cast<uint32_t>(__shfl_xor(static_cast<int>(front), d)));
                const bool behind = rec != 0 && key != front;
                bool slow_here = false;
                if (behind && a.catch_up == 2u) {
                    // the first step of the two-step descriptor, taken alone: E_a = {node to emit, offset base, landing record | LEAF_EMIT2 |
                    // DESC2_SLOW, flags} (load_kernels.hip: k_link_desc2), value and rank from the packed half-block of the offset (32 offsets:
                    // {values, ., ones before | ..}); the block base of the landing record is one more, dependent, load.  Records whose steps
                    // run through chains, or whose counts do not fit the packed blocks, leave the attempt to the loops.
                    const uint4 *d2 = ix.desc2 + 8 * static_cast<uint64_t>(rec);
                    const uint4 E0 = d2[0], E1 = d2[1];
                    const uint4 K = ix.gblocks[bb == BLOCK_NONE ? 0u : 2 * static_cast<uint64_t>(bb) + (offset >> 5)];
                    slow_here = (E0.z & DESC2_SLOW) != 0 || (E0.w & GATHER_OK) == 0 || (E0.w & (E_CHAIN | E_ANYCHAIN)) != 0 || (E1.w & E_CHAIN) != 0;
                    if (!slow_here) {
                        const uint32_t bit = offset & 31u;
                        const uint32_t value = (K.x >> bit) & 1u;
                        const uint32_t ones = (K.z & 0x1FFFFFu) + static_cast<uint32_t>(__popc(K.x & ((1u << bit) - 1u)));
                        const uint4 E = value ? E1 : E0;
                        const uint32_t land = E.z & REC_MASK;
                        rec = land; offset = E.y + (value ? ones : offset - ones);
                        bb = land != 0 ? ix.block_base[land] : BLOCK_NONE;
                        sink.push(E.x, E.x != 0);
                        sink.push(rec + ix.alphabet_offset, (E.z & LEAF_EMIT2) != 0);
                        if (sink.wr >= quota) { rec = 0; bb = BLOCK_NONE; }
                    }
                } else if (behind) {
                    // everything the step can need in ONE round trip: both edges, the flags and the rank block
                    const uint4 *d1 = ix.desc + 4 * static_cast<uint64_t>(rec);
                    const uint4 E0 = d1[0], E1 = d1[1], D = d1[2];
                    const uint4 K = ix.blocks[bb == BLOCK_NONE ? 0u : bb + (offset >> RANK_BLOCK_SHIFT)];
                    slow_here = (D.x & DESC_SLOW) != 0;
                    if (!slow_here) {
                        const uint64_t bits = (static_cast<uint64_t>(K.y) << 32) | K.x;
                        const uint32_t bit = offset & 63u;
                        const uint32_t value = static_cast<uint32_t>(bits >> bit) & 1u;
                        const uint32_t ones = K.z + __popcll(bits & ((uint64_t(1) << bit) - 1));
                        const uint4 E = value ? E1 : E0;
                        const uint32_t flags = value ? D.w : D.y;
                        rec = E.z; offset = E.y + (value ? ones : offset - ones); bb = E.w;
                        sink.push(E.x, E.x != 0);
                        sink.push(rec + ix.alphabet_offset, (flags & EDGE_EMIT2) != 0);
                        if (sink.wr >= quota) { rec = 0; bb = BLOCK_NONE; }
                    }
                }
                if (__ballot(slow_here) != 0) break;
                together = __ballot(rec != static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(rec))) == 0;
                if (__ballot(rec != 0) != walking) break;      // a lane has finished its segment: the wave will not be uniform again
            }
            lds_poke(my_mail + 3, sink.wr);
            caught = together;
            if (!together && --catch_credit == 0) { catch_pause = catch_backoff; catch_backoff = min(2 * catch_backoff, 256u); catch_credit = 1; }
            // the single steps have used the slack the loops count on (they are entered with at least eight free slots and push four
            // per iteration before they look again): back to the top, which waits for the helper if the ring is that full
            if (__ballot(sink.wr - lds_peek(my_drained) > ring_mask + 1 - headroom) != 0) continue;
        } else if (catch_pause != 0) catch_pause--;
        const uint32_t wr_entry = sink.wr;
#ifdef GBWT_HIP_PROBE_LOOP_SHARE
        const uint32_t wr0 = sink.wr;
        probe_entries++;
#endif
        uint32_t slow_exit = 2;
        if (a.uniform_loop && together) {
            // on the packed half-blocks until the wave meets a record they cannot count (reason 3), on the full-width blocks from then on
            if (!full_blocks) slow_exit = walk2_uniform_loop(ix.desc2, ix.gblocks, ix.alphabet_offset, ring_base, mail_slot, static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&row_state[lane].w)), narrow, quota, ring_mask, RING_PITCH, rec, offset, bb, sink.wr, headroom, a.all4 != 0);
            if (slow_exit == 3) full_blocks = true;
            if (full_blocks) slow_exit = walk2_uniform_loop_full(ix.desc2, ix.cblocks, ix.alphabet_offset, ring_base, mail_slot, static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&row_state[lane].w)), narrow, quota, ring_mask, RING_PITCH, rec, offset, bb, sink.wr, headroom, a.all4 != 0);
        }
#ifdef GBWT_HIP_PROBE_LOOP_SHARE
        const uint32_t wr1 = sink.wr;
        probe_uniform += wr1 - wr0;
        if (!together) probe_mixed_entries++;
        if (slow_exit == 2 && together) probe_fell_out++;
#endif
        if (caught) {
            // was it worth it?  Where the rows part again within a few iterations (indels at every other site) the single steps cost
            // more than the uniform loop returns: such waves stop trying for a while (profiles/r03_catch_up.txt)
            const uint32_t gained = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(sink.wr - wr_entry)));
            if (gained >= 24) { catch_credit = min(catch_credit + 1u, 4u); catch_backoff = 8; }
            else if (--catch_credit == 0) { catch_pause = catch_backoff; catch_backoff = min(2 * catch_backoff, 256u); catch_credit = 1; }
        }
        const bool mixed = slow_exit == 2;
        // (chained steps: the loop may have left with less than the headroom free -- back to the top, which waits for the helper)
        if (recheck && slow_exit != 0 && __ballot(sink.wr - lds_peek(my_drained) > ring_mask + 1 - headroom) != 0) continue;
        // How long the gather loop stays once it has used up the ring slots it was entered with: it can wait for the helper where it stands
        // (cheaper than the way out and back in), but only out here can a wave that is one step apart be brought together again -- so while
        // catch-up is being tried the loop comes back as it always did, while catch-up is backing off it stays for as long as the pause
        // would have lasted, and without catch-up it stays.
        const uint32_t patience = (a.catch_up && a.uniform_loop) ? 8u * catch_pause : 0x7FFFFFFFu;
        if (mixed && patience != 0 && catch_pause != 0) catch_pause = 1;      // (the stay replaces the pause: one more outer round, then a new attempt)
        if (mixed) slow_exit = full_blocks ? walk2_gather_loop_full(ix.desc2, ix.cblocks, ix.alphabet_offset, ring_base, mail_slot, drained, narrow, quota, ring_mask, RING_PITCH, rec, offset, bb, sink.wr, headroom, static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&row_state[lane].w)), patience)
                                           : walk2_gather_loop(ix.desc2, ix.gblocks, ix.alphabet_offset, ring_base, mail_slot, drained, narrow, quota, ring_mask, RING_PITCH, rec, offset, bb, sink.wr, headroom, static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&row_state[lane].w)), patience, a.gather_reach);
#ifdef GBWT_HIP_PROBE_LOOP_SHARE
        probe_vector += sink.wr - wr1;
#endif
        if (recheck && mixed && slow_exit != 0 && __ballot(sink.wr - lds_peek(my_drained) > ring_mask + 1 - headroom) != 0) continue;   // (the gather loop has run)
        if (slow_exit) {
            const uint4 here = ix.desc2[8 * static_cast<uint64_t>(rec)];
            bool generic = rec != 0 && (here.z & DESC2_SLOW) != 0;   // lanes on a slow record
            // a lane on a record too long for the packed counts: the wave continues on the full-width blocks (both loops), for good
            if (mixed && !full_blocks && __ballot(rec != 0 && !generic && !(here.w & GATHER_OK)) != 0) full_blocks = true;
            if (ix.wtables != nullptr) {
                // Lanes on a table record walk on the walk tables: one 16-byte entry per step says what to emit (the
                // successor and, where a unary record follows, the node behind it), where the walk lands and -- when that
                // is a table record again -- where its table is, so a chain of multi-allelic sites never goes back to the
                // hot loops.  The others wait; the loop ends when fewer than half of the walking lanes are still in it.
                bool in_table = false;
                uint32_t tb = 0;
                if (generic) {
                    const uint4 C = ix.desc_raw[4 * static_cast<uint64_t>(rec) + 2];
                    if (C.w == 1u && offset < C.y) { in_table = true; tb = C.z; generic = false; }   // offset >= Record::len: generic_step ends the walk (src/bwt.rs:481)
                }
                // With deep walk tables (k_fill_wtables_deep) a load brings the next seven steps: up to fourteen nodes, so the ring must have
                // sixteen free slots (the loops above leave with as few as four).  Nodes past the end of the segment are staged like
                // all others and never written (the row writers stop at the length of the row piece they serve).
                const bool deep = ix.wtables_deep != nullptr && ring_mask + 1 >= 64;
                const uint32_t table_slack = deep ? 16u : headroom;
                while (__ballot(in_table) != 0) {
                    if (deep && __ballot(in_table && sink.wr - lds_peek(my_drained) > ring_mask + 1 - table_slack) != 0) { __builtin_amdgcn_s_sleep(2); continue; }
                    // (compact entries, device_index.hpp: twelve steps = 24 nodes as 16-bit deltas.  They are staged in two halves of twelve
                    // with a look at the ring in between, so that the ring needs no more free slots than a seven-step entry asks for; the second
                    // half of the entry is fetched -- from the line the first half came with -- when it is needed: all sixteen words live across
                    // the first half and the wait cost the kernel its fourth wave per SIMD, 135 VGPRs, and the headline 9 %.)
                    bool compact = false;
                    const uint4 *entry = nullptr;
                    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
                    if (in_table && deep) {
                        entry = ix.wtables_deep + 4 * (static_cast<uint64_t>(tb) + offset);
                        c0 = entry[0]; c1 = entry[1];
                        compact = (c0.y & WT_COMPACT) != 0;
                    }
                    if (__ballot(compact) != 0) {
                        uint32_t node = c0.x;
                        const auto step16 = [&](uint32_t half) { node += static_cast<uint32_t>(static_cast<int32_t>(static_cast<int16_t>(half & 0xFFFFu))); sink.push(node, compact); };
                        const auto pair = [&](uint32_t word) { step16(word); step16(word >> 16); };
                        sink.push(node, compact);
                        pair(c0.z); pair(c0.w); pair(c1.x); pair(c1.y); pair(c1.z);
                        step16(c1.w);                                                                       // node 11
                        const uint32_t carry = c1.w >> 16, flags = c0.y;
                        lds_poke(my_mail + 3, sink.wr);
                        while (__ballot(compact && sink.wr - lds_peek(my_drained) > ring_mask + 1 - table_slack) != 0) __builtin_amdgcn_s_sleep(2);
                        uint4 c2 = make_uint4(0, 0, 0, 0), c3 = c2;
                        if (compact) { c2 = entry[2]; c3 = entry[3]; }
                        step16(carry);                                                                      // node 12
                        pair(c2.x); pair(c2.y); pair(c2.z); pair(c2.w); pair(c3.x);
                        step16(c3.y);                                                                       // node 23
                        if (compact) {
                            rec = node - ix.alphabet_offset; offset = c3.z;
                            if (flags & WT_COMPACT_TABLE) tb = c3.w; else { bb = c3.w; in_table = false; }
                            if (sink.wr >= quota) { rec = 0; bb = BLOCK_NONE; in_table = false; }
                        }
                    }
                    if (in_table && deep && !compact) {
                        const uint4 q0 = c0, q1 = c1, q2 = entry[2], q3 = entry[3];
                        const uint32_t ao = ix.alphabet_offset;
                        sink.push(q0.x, q0.x != 0); sink.push((q0.y & REC_MASK) + ao, (q0.y & LEAF_EMIT2) != 0);
                        sink.push(q0.z, q0.z != 0); sink.push((q0.w & REC_MASK) + ao, (q0.w & LEAF_EMIT2) != 0);
                        sink.push(q1.x, q1.x != 0); sink.push((q1.y & REC_MASK) + ao, (q1.y & LEAF_EMIT2) != 0);
                        sink.push(q1.z, q1.z != 0); sink.push((q1.w & REC_MASK) + ao, (q1.w & LEAF_EMIT2) != 0);
                        sink.push(q2.x, q2.x != 0); sink.push((q2.y & REC_MASK) + ao, (q2.y & LEAF_EMIT2) != 0);
                        sink.push(q2.z, q2.z != 0); sink.push((q2.w & REC_MASK) + ao, (q2.w & LEAF_EMIT2) != 0);
                        sink.push(q3.x, q3.x != 0); sink.push((q3.y & REC_MASK) + ao, (q3.y & LEAF_EMIT2) != 0);
                        rec = q3.y & REC_MASK; offset = q3.z;
                        if (q3.y & WT_TABLE) tb = q3.w; else { bb = q3.w; in_table = false; }
                        if (rec == 0 || sink.wr >= quota) { rec = 0; bb = BLOCK_NONE; in_table = false; }
                    } else if (in_table && !deep) {
                        const uint4 e = ix.wtables[static_cast<uint64_t>(tb) + offset];
                        sink.push(e.x, e.x != 0);
                        sink.push((e.z & REC_MASK) + ix.alphabet_offset, (e.z & LEAF_EMIT2) != 0);
                        rec = e.z & REC_MASK; offset = e.y;
                        if (e.z & WT_TABLE) tb = e.w; else { bb = e.w; in_table = false; }
                        if (rec == 0 || sink.wr >= quota) { rec = 0; bb = BLOCK_NONE; in_table = false; }
                    }
                    lds_poke(my_mail + 3, sink.wr);
                    const uint64_t still = __ballot(in_table);
                    if (2 * __popcll(still) < __popcll(__ballot(rec != 0))) break;
                    if (__ballot(sink.wr - lds_peek(my_drained) > ring_mask + 1 - table_slack) != 0) break;   // ring full: the outer loop waits for the helper
                }
                if (in_table) bb = BLOCK_NONE;     // left on a table record (it has no blocks): back here after the next look at the ring
            }
            if (generic) {   // no table (or no walk tables at all): one step of the generic decoder with all the reference's tests
                generic_step(ix, sink, rec, offset, bb);
                if (sink.wr >= quota) { rec = 0; bb = BLOCK_NONE; }
            }
            lds_poke(my_mail + 3, sink.wr);
        }
    }
    lds_poke(my_mail + 3, sink.wr);
#ifdef GBWT_HIP_PROBE_LOOP_SHARE
    if ((lane == 0 || lane == 37) && blockIdx.x % 20011u == 7u)
        printf("workgroup %u lane %u: %u nodes in the uniform loop, %u in the vector loop; %u entries, %u of them mixed, %u fell out at once\n", blockIdx.x, lane,
               probe_uniform, probe_vector, probe_entries, probe_mixed_entries, probe_fell_out);
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_poke(done_flag, 1);
}

}  // namespace

// keys[k] = number of segments of row k = samples of its sequence (0 for an empty sequence), rows[k] = k
__global__ void __launch_bounds__(256) k_segment_counts(DeviceIndex ix, const uint64_t *ids, uint64_t n, uint32_t *keys, uint32_t *rows) {
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const uint64_t id = ids[k];
    uint32_t segments = 0;
    if (id < ix.n_sequences) { const RowSegments rs = row_segments(ix, id); segments = static_cast<uint32_t>(rs.hi - rs.lo); }
    keys[k] = segments;
    rows[k] = static_cast<uint32_t>(k);
}

// lengths[k] = nodes of row k that lie in the part of it this extraction fills (DeviceIndex::sample_part); max_len as k_gather_lengths has it
__global__ void __launch_bounds__(256) k_part_lengths(DeviceIndex ix, const uint64_t *ids, uint64_t n, uint64_t *lengths, uint32_t *max_len) {
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const uint64_t id = ids[k];
    uint32_t len = 0;
    if (id < ix.n_sequences) {
        const RowSegments rs = row_segments(ix, id);
        len = static_cast<uint32_t>(segment_position(ix, rs, id, rs.hi) - segment_position(ix, rs, id, rs.lo));
    }
    lengths[k] = len;
    atomicMax(max_len, len);
    atomicMax(max_len + 1, ~len);
}

void launch_part_lengths(const DeviceIndex &ix, const uint64_t *d_ids, uint64_t n, uint64_t *d_lengths, uint32_t *d_max_len, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_part_lengths, dim3(grid_for(n, 256)), dim3(256), 0, stream, ix, d_ids, n, d_lengths, d_max_len);
}

// ROW OFFSETS IN ONE LAUNCH (round 4).  Lengths, their exclusive scan and the extremes of a batch were a memset and three launches (the
// lengths, hipcub's two scan kernels): 36 us of host time per request at 9 us a launch -- a twentieth of a pass that takes 0.6 ms (one
// rank of eight).  One workgroup does all of it for batches of up to 8 192 rows (a thread per stretch of rows, a block scan of the
// stretch sums: about 3 us per 1 024 rows); launch_row_offsets says no above ROW_OFFSETS_MAX rows and the caller takes the three launches.
constexpr uint32_t ROW_OFFSETS_THREADS = 1024;
__global__ void __launch_bounds__(ROW_OFFSETS_THREADS) k_row_offsets(DeviceIndex ix, const uint64_t *ids, uint64_t n, uint64_t *lengths, uint64_t *offsets, uint32_t *max_len) {
    __shared__ uint64_t sums[ROW_OFFSETS_THREADS];
    __shared__ uint32_t longest[ROW_OFFSETS_THREADS], shortest[ROW_OFFSETS_THREADS];
    const uint32_t t = threadIdx.x;
    const uint64_t per = (n + ROW_OFFSETS_THREADS - 1) / ROW_OFFSETS_THREADS, first = min(n, t * per), last = min(n, first + per);
    uint64_t sum = 0;
    uint32_t hi = 0, lo = 0xFFFFFFFFu;
    for (uint64_t k = first; k < last; k++) {
        const uint64_t id = ids[k];
        uint32_t len = 0;
        if (id < ix.n_sequences) {
            if (ix.sample_parts > 1) {
                const RowSegments rs = row_segments(ix, id);
                len = static_cast<uint32_t>(segment_position(ix, rs, id, rs.hi) - segment_position(ix, rs, id, rs.lo));
            } else len = ix.seq_len[id];
        }
        lengths[k] = len;
        sum += len; hi = max(hi, len); lo = min(lo, len);
    }
    sums[t] = sum; longest[t] = hi; shortest[t] = lo;
    __syncthreads();
    for (uint32_t d = 1; d < ROW_OFFSETS_THREADS; d *= 2) {      // inclusive scan of the stretch sums; the extremes ride along as plain reductions
        const uint64_t add = t >= d ? sums[t - d] : 0;
        const uint32_t h = t >= d ? longest[t - d] : 0, l = t >= d ? shortest[t - d] : 0xFFFFFFFFu;
        __syncthreads();
        sums[t] += add; longest[t] = max(longest[t], h); shortest[t] = min(shortest[t], l);
        __syncthreads();
    }
    uint64_t at = sums[t] - sum;
    for (uint64_t k = first; k < last; k++) { offsets[k] = at; at += lengths[k]; }
    if (t == ROW_OFFSETS_THREADS - 1) { offsets[n] = sums[t]; max_len[0] = longest[t]; max_len[1] = ~shortest[t]; }
}

bool launch_row_offsets(const DeviceIndex &ix, const uint64_t *d_ids, uint64_t n, uint64_t *d_lengths, uint64_t *d_offsets, uint32_t *d_max_len, hipStream_t stream) {
    if (n == 0 || n > ROW_OFFSETS_MAX) return false;
    hipLaunchKernelGGL(k_row_offsets, dim3(1), dim3(ROW_OFFSETS_THREADS), 0, stream, ix, d_ids, n, d_lengths, d_offsets, d_max_len);
    return true;
}

// counts[j] = rows with more than j segments = the first position of the descending keys that is <= j
__global__ void __launch_bounds__(256) k_level_counts(const uint32_t *sorted_keys, uint64_t n, uint32_t segments, uint64_t *counts) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= segments) return;
    uint64_t lo = 0, hi = n;                                        // keys[< lo] > j, keys[>= hi] <= j
    while (lo < hi) {
        const uint64_t mid = (lo + hi) / 2;
        if (sorted_keys[mid] > j) lo = mid + 1; else hi = mid;
    }
    counts[j] = lo;
}

size_t walker_order_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    hipcub::DoubleBuffer<uint32_t> keys(nullptr, nullptr), rows(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, keys, rows, static_cast<int>(n));
    return bytes;
}

void launch_walker_order(const DeviceIndex &ix, const uint64_t *d_ids, uint64_t n, uint32_t segments, uint32_t *d_keys, uint32_t *d_rows,
                         uint64_t *d_level_counts, uint64_t *d_level, void *d_temp, size_t temp_bytes, const uint32_t **d_sorted_rows, hipStream_t stream) {
    hipLaunchKernelGGL(k_segment_counts, dim3(grid_for(n, 256)), dim3(256), 0, stream, ix, d_ids, n, d_keys, d_rows);
    hipcub::DoubleBuffer<uint32_t> keys(d_keys, d_keys + n), rows(d_rows, d_rows + n);
    (void)hipcub::DeviceRadixSort::SortPairsDescending(d_temp, temp_bytes, keys, rows, static_cast<int>(n), 0, 32, stream);   // radix sort: stable
    hipLaunchKernelGGL(k_level_counts, dim3(grid_for(segments, 256)), dim3(256), 0, stream, keys.Current(), n, segments, d_level_counts);
    *d_sorted_rows = rows.Current();
    // d_level[0] = 0, d_level[j + 1] = counts[0] + ... + counts[j]: the caller runs launch_scan on d_level_counts
    (void)d_level;
}

void launch_walk_direct(const DeviceIndex &ix, const WalkArgs &args, hipStream_t stream) {
    const unsigned p = args.paths_per_wave ? args.paths_per_wave : WAVE;
    const uint64_t walkers = args.segments ? args.walkers : (args.both_ends ? 2 * args.n : args.n);
    unsigned groups = grid_for(walkers, p);
    if (args.xcd_map) groups = (groups + 7u) / 8u * 8u;   // whole eighths; the workgroups past the end own nothing
    if (args.debug & 16384u) groups = grid_for(walkers, p) * 2u + 8u;
    // the product kernel, or -- for a workspace whose knobs ask for a measurement switch (GBWT_HIP_DEBUG_DRY_ROWS) -- the probe one
    const bool probe = args.debug != 0;
    if (probe) hipLaunchKernelGGL(k_walk_direct<true>, dim3(groups), dim3(2 * WAVE), args.ring_slots * RING_PITCH * sizeof(uint32_t), stream, ix, args);
    else hipLaunchKernelGGL(k_walk_direct<false>, dim3(groups), dim3(2 * WAVE), args.ring_slots * RING_PITCH * sizeof(uint32_t), stream, ix, args);
}

}  // namespace gbwt_hip
