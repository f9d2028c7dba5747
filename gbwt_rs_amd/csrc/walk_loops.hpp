// walk_loops.hpp -- device code shared by the walk kernels (walk_kernels.hip, walk_direct.hip) and the one-time walks at
// open (open_walks.hip): the output sinks of the pool walks, the arrival tests and the generic step, the hot loops in
// gfx950 assembly (one LF step per iteration, two LF steps, two LF steps with a wave-uniform scalar descriptor fetch),
// their look-ahead helpers, and the two-step walk in plain C++ (quiet_walk).  Everything lives in an anonymous
// namespace: every translation unit gets its own copy.
#pragma once

#include <hip/hip_runtime.h>

#include "coop_device.hpp"
#include "device_common.hpp"
#include "kernels.hpp"
#include "lf_device.hpp"

namespace gbwt_hip {

namespace {

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


// ---- two-step walk --------------------------------------------------------------------------------------
// Same frame as k_walk_blocks; an iteration of the hot loop composes two LF steps (k_link_desc2, k_fill_cblocks):
//     a = bit `offset` of bits1;  rank_a = equal values of v before it;        j = base_a + rank_a  (offset in w_a)
//     b = bit `offset` of bits2;  rank_b = equal values of w_a before j = R_a + (a-paths of this block before `offset`
//                                          whose value in w_a is 1), or j minus that
//     leaf (a, b): rec = its landing record, offset = its base + rank_b
//     emit: node of edge a, node of w_a if that step was fused, node of the leaf, node of rec if that step was fused
__device__ __forceinline__ uint32_t walk2_hot_loop(const uint4 *desc2, const uint4 *cblocks, uint32_t alphabet_offset, uint32_t ring_base,
                                                   uint32_t mail_slot, uint32_t flushed, bool narrow, uint32_t quota, uint32_t ring_mask, uint32_t ring_stride,
                                                   uint32_t &rec, uint32_t &offset, uint32_t &bb, uint32_t &wr) {
#ifdef GBWT_HIP_CXX_LOOP
    // plain C++ statement of the loop (no pipelining)
    __attribute__((address_space(3))) uint32_t *ring = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)ring_base;
    __attribute__((address_space(3))) uint32_t *mail = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)mail_slot;
    for (;;) {
        const uint4 *d = desc2 + 8 * static_cast<uint64_t>(rec);
        const uint4 E0 = d[0], E1 = d[1], L00 = d[2], L01 = d[3], L10 = d[4], L11 = d[5], look = d[6];
        const uint64_t idx = bb == BLOCK_NONE ? 0u : bb + (offset >> RANK_BLOCK_SHIFT);
        const uint4 K0 = cblocks[2 * idx], K1 = cblocks[2 * idx + 1];
        if (__ballot((E0.z & DESC2_SLOW) != 0) != 0) return 1;
        const uint64_t bits1 = (static_cast<uint64_t>(K0.y) << 32) | K0.x, bits2 = (static_cast<uint64_t>(K0.w) << 32) | K0.z;
        const uint32_t bit = offset & 63u;
        const uint64_t below = (uint64_t(1) << bit) - 1;
        const uint32_t a = static_cast<uint32_t>(bits1 >> bit) & 1u;
        const uint64_t m = a ? bits1 : ~bits1;
        const uint32_t p = __popcll(m & below);
        const uint32_t rank_a = a ? K1.x + p : (offset - bit) - K1.x + p;
        const uint32_t j = (a ? E1.y : E0.y) + rank_a;
        const uint32_t b = static_cast<uint32_t>(bits2 >> bit) & 1u;
        const uint32_t ones_w = (a ? K1.z : K1.y) + __popcll(m & bits2 & below);
        const uint32_t rank_b = b ? ones_w : j - ones_w;
        const uint4 leaf = a ? (b ? L11 : L10) : (b ? L01 : L00);
        const uint32_t n1 = a ? E1.x : E0.x, wword = a ? E1.z : E0.z;
        rec = leaf.z & REC_MASK; offset = leaf.y + rank_b; bb = leaf.w;
        ring[(wr & ring_mask) * ring_stride] = n1;
        wr += n1 != 0 ? 1u : 0u;
        ring[(wr & ring_mask) * ring_stride] = (wword & REC_MASK) + alphabet_offset;
        wr += (wword & LEAF_EMIT2) ? 1u : 0u;
        ring[(wr & ring_mask) * ring_stride] = leaf.x;
        wr += leaf.x != 0 ? 1u : 0u;
        ring[(wr & ring_mask) * ring_stride] = rec + alphabet_offset;
        wr += (leaf.z & LEAF_EMIT2) ? 1u : 0u;
        mail[0] = look.x; mail[1] = look.y; mail[2] = look.z; mail[3] = wr;
        if (wr >= quota) { rec = 0; bb = BLOCK_NONE; }
        if (__ballot(rec != 0) == 0 || __ballot(wr - flushed > ring_mask + 1 - 8) != 0) return 0;
    }
#else
    // The same loop in gfx950 assembly (see walk_hot_loop for the conventions: all lanes run everything, parked lanes sit
    // on record 0, loads of the next position go out as early as possible, exits leave nothing in flight; a VALU write
    // of VCC / an SGPR is kept two instructions away from the VALU that reads it).  Registers v40-v125, s41, s44-s47.
    uint32_t reason;
#ifdef GBWT_HIP_PROBE_MORE_LOADS   // measurement only: two more loads per lane and step from the descriptor's line (profiles/r02_walk_bounds.txt #16,
                                   // taken when k_walk_direct still ran this loop for mixed waves; today the loop serves the pool-output kernel)
#define GBWT_WALK2_PROBE_WIDE "global_load_dwordx2 v[46:47], v[88:89], off offset:112\n\t" "global_load_dwordx2 v[68:69], v[88:89], off offset:24\n\t"
#define GBWT_WALK2_PROBE_NARROW "global_load_dwordx2 v[46:47], v88, %[desc2] offset:112\n\t" "global_load_dwordx2 v[68:69], v88, %[desc2] offset:24\n\t"
#define GBWT_WALK2_PROBE_CLOBBERS "v46", "v47", "v68", "v69",
#else
#define GBWT_WALK2_PROBE_WIDE
#define GBWT_WALK2_PROBE_NARROW
#define GBWT_WALK2_PROBE_CLOBBERS
#endif
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
    "global_load_dwordx3 v[48:50], v[88:89], off\n\t"             /* E_0 */                               \
    "global_load_dwordx3 v[52:54], v[88:89], off offset:16\n\t"   /* E_1 */                                \
    "global_load_dwordx4 v[56:59], v[88:89], off offset:32\n\t"   /* leaf (0, 0) */                       \
    "global_load_dwordx4 v[60:63], v[88:89], off offset:48\n\t"   /* leaf (0, 1) */                       \
    "global_load_dwordx4 v[64:67], v[88:89], off offset:64\n\t"   /* leaf (1, 0) */                       \
    "global_load_dwordx4 v[72:75], v[88:89], off offset:80\n\t"   /* leaf (1, 1) */                       \
    "global_load_dwordx3 v[76:78], v[88:89], off offset:96\n\t"   /* look-ahead target */                 \
    GBWT_WALK2_PROBE_WIDE \
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
    "global_load_dwordx3 v[48:50], v88, %[desc2]\n\t"               /* E_0 */                             \
    "global_load_dwordx3 v[52:54], v88, %[desc2] offset:16\n\t"     /* E_1 */                              \
    "global_load_dwordx4 v[56:59], v88, %[desc2] offset:32\n\t"     /* leaf (0, 0) */                     \
    "global_load_dwordx4 v[60:63], v88, %[desc2] offset:48\n\t"     /* leaf (0, 1) */                     \
    "global_load_dwordx4 v[64:67], v88, %[desc2] offset:64\n\t"     /* leaf (1, 0) */                     \
    "global_load_dwordx4 v[72:75], v88, %[desc2] offset:80\n\t"     /* leaf (1, 1) */                     \
    "global_load_dwordx3 v[76:78], v88, %[desc2] offset:96\n\t"     /* look-ahead target */               \
    GBWT_WALK2_PROBE_NARROW \
    "s_mov_b64 exec, -1\n\t"
#define GBWT_WALK2_LOOP(ISSUE)                                                                             \
    asm volatile( \
        "v_mov_b32_e32 v40, %[rec]\n\t" \
        "v_mov_b32_e32 v41, 0\n\t" \
        "v_mov_b32_e32 v42, %[offset]\n\t" \
        "v_mov_b32_e32 v43, %[bb]\n\t" \
        "v_mov_b32_e32 v44, %[wr]\n\t" \
        "v_mov_b32_e32 v71, 0\n\t" \
        "s_mov_b32 %[reason], 0\n\t" \
        "s_mov_b64 s[44:45], -1\n\t" \
        ISSUE \
        ".Lgbwt_walk2_loop_%=:\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "v_lshlrev_b32_e32 v92, 1, v50\n\t"                 /* DESC2_SLOW (bit 30 of E_0.z) -> sign */ \
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
        "v_cndmask_b32_e32 v103, v49, v53, vcc\n\t"         /* offset base of edge a */ \
        "v_add_u32_e32 v99, v99, v102\n\t"                  /* rank_a */ \
        "v_cndmask_b32_e32 v104, v48, v52, vcc\n\t"         /* node of edge a */ \
        "v_add_u32_e32 v103, v103, v99\n\t"                 /* j: offset in w_a */ \
        "v_cndmask_b32_e32 v105, v50, v54, vcc\n\t"         /* w_a | flags */ \
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
        "v_and_b32_e32 v92, %[ringmask], v44\n\t"                  /* ring slot of the next node */ \
        "v_cmp_ne_u32_e32 vcc, 0, v104\n\t" \
        "v_mad_u32_u24 v92, v92, %[stride], %[ring]\n\t" \
        "ds_write_b32 v92, v104\n\t"                        /* node of edge a */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t"       /* counts if it is not the ENDMARKER */ \
        "v_cmp_gt_i32_e32 vcc, 0, v105\n\t"                 /* first step fused? */ \
        "v_and_b32_e32 v92, %[ringmask], v44\n\t" \
        "v_add_u32_e32 v110, s41, v110\n\t"                 /* node of w_a */ \
        "v_mad_u32_u24 v92, v92, %[stride], %[ring]\n\t" \
        "ds_write_b32 v92, v110\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_ne_u32_e32 vcc, 0, v112\n\t" \
        "v_and_b32_e32 v92, %[ringmask], v44\n\t" \
        "v_add_u32_e32 v111, s41, v40\n\t"                  /* node of the landing record */ \
        "v_mad_u32_u24 v92, v92, %[stride], %[ring]\n\t" \
        "ds_write_b32 v92, v112\n\t"                        /* node of the leaf */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_gt_i32_e32 vcc, 0, v114\n\t"                 /* second step fused? */ \
        "v_and_b32_e32 v92, %[ringmask], v44\n\t" \
        "v_mad_u32_u24 v92, v92, %[stride], %[ring]\n\t" \
        "ds_write_b32 v92, v111\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_lt_u32_e32 vcc, v44, %[quota]\n\t"            /* a walker that has emitted its share parks (both-ends walks) */ \
        "v_mov_b32_e32 v79, v44\n\t"                        /* mailbox: look-ahead target of the record just left + nodes staged so far */ \
        "ds_write_b128 %[mail], v[76:79]\n\t" \
        "v_cndmask_b32_e32 v40, 0, v40, vcc\n\t" \
        "v_cndmask_b32_e32 v43, -1, v43, vcc\n\t" \
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
        : [rec] "+v"(rec), [offset] "+v"(offset), [bb] "+v"(bb), [wr] "+v"(wr), [reason] "=&s"(reason) \
        : [desc2] "s"(desc2), [cblocks] "s"(cblocks), [ring] "v"(ring_base), [mail] "v"(mail_slot), [limit] "v"(limit), [quota] "v"(quota), [ringmask] "s"(ring_mask), [stride] "s"(4 * ring_stride), \
          "{s41}"(alphabet_offset) \
        : "memory", "vcc", "scc", "s44", "s45", "s46", "s47", GBWT_WALK2_PROBE_CLOBBERS \
          "v40", "v41", "v42", "v43", "v44", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", \
          "v64", "v65", "v66", "v67", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", \
          "v88", "v89", "v90", "v91", "v92", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", \
          "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
    const uint32_t limit = flushed + (ring_mask + 1 - 8);   // leave with more than slots - 8 nodes waiting
    if (narrow) { GBWT_WALK2_LOOP(GBWT_WALK2_ISSUE_NARROW) } else { GBWT_WALK2_LOOP(GBWT_WALK2_ISSUE_WIDE) }
#undef GBWT_WALK2_LOOP
#undef GBWT_WALK2_ISSUE_NARROW
#undef GBWT_WALK2_ISSUE_WIDE
    return reason;
#endif
}

// Chained steps in the assembly loops (device_index.hpp: E_CHAIN / LEAF_CHAIN; k_link_desc2): between the first node X of a step (already
// staged) and the node of its landing record the lanes whose step is chained stage X + d, X + 2 d, ... (d = +-2).  FLAG_TEST sets vcc for
// those lanes; LREC = the landing record; L, D, C, A = four scratch VGPRs; s[80:81] keeps exec; SKIP = a scalar test that branches to
// .Lgbwt_chain_skip_<TAG> (the uniform loop: no step of the record is chained -- E_ANYCHAIN of E_0.w, kept in s82), or nothing.
#ifndef GBWT_HIP_CXX_LOOP
#define GBWT_CHAIN_BLOCK(TAG, SKIP, FLAG_TEST, X, LREC, L, D, C, A) \
        SKIP \
        FLAG_TEST \
        "s_nop 1\n\t" \
        "s_and_saveexec_b64 s[80:81], vcc\n\t" \
        "s_cbranch_execz .Lgbwt_chain_done_" TAG "_%=\n\t" \
        "v_add_u32_e32 " L ", s41, " LREC "\n\t"             /* node of the landing record */ \
        "v_mov_b32_e32 " D ", -2\n\t" \
        "v_cmp_gt_u32_e32 vcc, " L ", " X "\n\t"             /* ascending ids? */ \
        "v_cndmask_b32_e64 " D ", " D ", 2, vcc\n\t"          /* the stride */ \
        "v_add_u32_e32 " C ", " X ", " D "\n\t"               /* the first node between */ \
        ".Lgbwt_chain_next_" TAG "_%=:\n\t" \
        "v_cmp_ne_u32_e32 vcc, " C ", " L "\n\t" \
        "s_nop 1\n\t" \
        "s_and_b64 exec, exec, vcc\n\t" \
        "s_cbranch_execz .Lgbwt_chain_done_" TAG "_%=\n\t" \
        "v_and_b32_e32 " A ", %[ringmask], v44\n\t" \
        "v_mad_u32_u24 " A ", " A ", %[stride], %[ring]\n\t" \
        "ds_write_b32 " A ", " C "\n\t" \
        "v_add_u32_e32 v44, 1, v44\n\t" \
        "v_add_u32_e32 " C ", " C ", " D "\n\t" \
        "s_branch .Lgbwt_chain_next_" TAG "_%=\n\t" \
        ".Lgbwt_chain_done_" TAG "_%=:\n\t" \
        "s_mov_b64 exec, s[80:81]\n\t" \
        ".Lgbwt_chain_skip_" TAG "_%=:\n\t"
#endif

// ---- two-step walk, gather variant -------------------------------------------------------------------------------
// For waves whose lanes sit on DIFFERENT records (graphs with indels: rows of a batch leave lock step after the first
// site).  There every vector-memory instruction costs the address path a pass over sixty-four distinct lines -- about
// one lane per cycle and CU whatever the width of the load -- and that, not latency, bounds the walk
// (profiles/r02_walk_bounds.txt #16: time follows the number of load instructions per step, 12 with the helper's
// touches in walk2_hot_loop).  So this loop fetches only what the step needs, in three instructions: the lane's packed
// half-block (gblocks: 32 offsets, 16 bytes), then -- once a and b are known -- E_a and leaf (a, b); by default it posts no
// look-ahead target.  The price is a second round trip per iteration; the other waves of the CU cover it.
// `lookahead` (round 6): where few rows pass a record (config 4: ~50 positions per record, a record's descriptor and blocks are fetched
// by one or two waves and never again) every one of those three loads is a first touch that goes to HBM; there a fourth load brings the
// record's look-ahead target and the helper wave touches it -- and a stretch of records behind it -- like it does for the uniform loop.
// Same contract as walk2_hot_loop; reason 1 also when a lane sits on a record without GATHER_OK (the caller then moves the wave
// to walk2_gather_loop_full).
__device__ __forceinline__ uint32_t walk2_gather_loop(const uint4 *desc2, const uint4 *gblocks, uint32_t alphabet_offset, uint32_t ring_base,
                                                      uint32_t mail_slot, uint32_t flushed, bool narrow, uint32_t quota, uint32_t ring_mask, uint32_t ring_stride,
                                                      uint32_t &rec, uint32_t &offset, uint32_t &bb, uint32_t &wr, uint32_t headroom = 8, uint32_t drained_addr = 0, uint32_t patience = 0,
                                                      uint32_t lookahead = 0) {
#ifdef GBWT_HIP_CXX_LOOP
    __attribute__((address_space(3))) uint32_t *ring = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)ring_base;
    __attribute__((address_space(3))) uint32_t *mail = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)mail_slot;
    mail[0] = 0;                                             // no look-ahead target yet (and none at all without `lookahead`: see below)
    for (;;) {
        bool slow = false;
        if (rec != 0) {
            const uint64_t idx = bb == BLOCK_NONE ? 0u : 2 * static_cast<uint64_t>(bb) + (offset >> 5);
            const uint4 K = gblocks[idx];
            const uint32_t bit = offset & 31u, below = (1u << bit) - 1;
            const uint32_t a = (K.x >> bit) & 1u, b = (K.y >> bit) & 1u;
            const uint4 *d = desc2 + 8 * static_cast<uint64_t>(rec);
            const uint4 E = d[a], leaf = d[2 + 2 * a + b];
            slow = (E.z & DESC2_SLOW) != 0 || !(E.w & GATHER_OK);
            if (!slow) {
                const uint32_t ones1 = K.z & 0x1FFFFFu, R0 = ((K.z >> 21) | (K.w << 11)) & 0x1FFFFFu, R1 = K.w >> 10;
                const uint32_t m = (a ? K.x : ~K.x) & below;
                const uint32_t p = __popc(m);
                const uint32_t rank_a = a ? ones1 + p : (offset - bit) - ones1 + p;
                const uint32_t j = E.y + rank_a;
                const uint32_t ones_w = (a ? R1 : R0) + __popc(m & K.y);
                rec = leaf.z & REC_MASK; offset = leaf.y + (b ? ones_w : j - ones_w); bb = leaf.w;
                ring[(wr & ring_mask) * ring_stride] = E.x;
                wr += E.x != 0 ? 1u : 0u;
                if (E.w & E_CHAIN) { const uint32_t L = (E.z & REC_MASK) + alphabet_offset, d = chain_stride(E.x, L); for (uint32_t n = E.x + d; n != L; n += d) ring[(wr++ & ring_mask) * ring_stride] = n; }
                ring[(wr & ring_mask) * ring_stride] = (E.z & REC_MASK) + alphabet_offset;
                wr += (E.z & LEAF_EMIT2) ? 1u : 0u;
                ring[(wr & ring_mask) * ring_stride] = leaf.x;
                wr += leaf.x != 0 ? 1u : 0u;
                if (leaf.z & LEAF_CHAIN) { const uint32_t L = rec + alphabet_offset, d = chain_stride(leaf.x, L); for (uint32_t n = leaf.x + d; n != L; n += d) ring[(wr++ & ring_mask) * ring_stride] = n; }
                ring[(wr & ring_mask) * ring_stride] = rec + alphabet_offset;
                wr += (leaf.z & LEAF_EMIT2) ? 1u : 0u;
                if (lookahead) { const uint4 T = d[6]; mail[0] = T.x; mail[1] = T.y; mail[2] = T.z; }   // the look-ahead target of the record just left
                mail[3] = wr;
                if (wr >= quota) { rec = 0; bb = BLOCK_NONE; }
            }
        }
        if (__ballot(slow) != 0) return 1;
        if (__ballot(rec != 0) == 0 || __ballot(wr - flushed > ring_mask + 1 - headroom) != 0) return 0;
    }
#else
    // gfx950 assembly: exactly three vector loads per iteration in two rounds (four with `lookahead`: the look-ahead target of the record,
    // in the line E_a and the leaf come from, posted in the mailbox like the uniform loop's).  Conventions of walk2_hot_loop; registers v40-v87;
    // s[44:45] = the lanes that load (those walking at the start of the PREVIOUS iteration, so that a lane that parks fetches
    // record 0 once and keeps emitting nothing), s[42:43] = the lanes walking now.
    uint32_t reason;
    const uint32_t limit = flushed + (ring_mask + 1 - headroom);   // leave with more than slots - 8 nodes waiting
#define GBWT_GATHER_K_NARROW                                                                              \
    "v_lshlrev_b32_e32 v70, 4, v58\n\t"                   /* packed half-blocks are 16 bytes */            \
    "v_lshlrev_b32_e32 v68, 7, v40\n\t"                   /* two-step descriptors are 128 bytes */        \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                        \
    "global_load_dwordx4 v[60:63], v70, %[gblocks]\n\t"   /* bits1, bits2, ones1 | R_0 << 21, R_0 >> 11 | R_1 << 10 */ \
    "s_mov_b64 exec, -1\n\t"
#define GBWT_GATHER_K_WIDE                                                                                \
    "v_lshlrev_b64 v[70:71], 4, v[58:59]\n\t"                                                             \
    "v_lshlrev_b64 v[68:69], 7, v[40:41]\n\t"                                                             \
    "v_lshl_add_u64 v[70:71], v[70:71], 0, %[gblocks]\n\t"                                                \
    "v_lshl_add_u64 v[68:69], v[68:69], 0, %[desc2]\n\t"                                                  \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                        \
    "global_load_dwordx4 v[60:63], v[70:71], off\n\t"                                                     \
    "s_mov_b64 exec, -1\n\t"
// E_a at 16 a, leaf (a, b) at 32 + 16 (2 a + b) of the descriptor; v72 = a, v84 = b
#define GBWT_GATHER_D_NARROW                                                                              \
    "v_lshl_add_u32 v46, v72, 4, v68\n\t"                                                                 \
    "v_lshl_add_u32 v47, v72, 1, v84\n\t"                                                                 \
    "v_lshl_add_u32 v47, v47, 4, v68\n\t"                                                                 \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                        \
    "global_load_dwordx4 v[48:51], v46, %[desc2]\n\t"               /* E_a: node, offset base, w_a | flags, GATHER_OK */ \
    "global_load_dwordx4 v[52:55], v47, %[desc2] offset:32\n\t"     /* leaf (a, b) */                     \
    "s_cmp_eq_u32 %[look], 0\n\t"                                                                         \
    "s_cbranch_scc1 .Lgbwt_gather_nolook_%=\n\t"                                                          \
    "global_load_dwordx3 v[64:66], v68, %[desc2] offset:96\n\t"     /* look-ahead target {record, first block, blocks} */ \
    ".Lgbwt_gather_nolook_%=:\n\t"                                                                         \
    "s_mov_b64 exec, -1\n\t"
#define GBWT_GATHER_D_WIDE                                                                                \
    "v_lshl_add_u32 v56, v72, 1, v84\n\t"                                                                 \
    "v_mov_b32_e32 v73, 0\n\t"                                                                            \
    "v_mov_b32_e32 v57, 0\n\t"                                                                            \
    "v_lshl_add_u64 v[46:47], v[72:73], 4, v[68:69]\n\t"                                                  \
    "v_lshl_add_u64 v[56:57], v[56:57], 4, v[68:69]\n\t"                                                  \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                        \
    "global_load_dwordx4 v[48:51], v[46:47], off\n\t"                                                     \
    "global_load_dwordx4 v[52:55], v[56:57], off offset:32\n\t"                                           \
    "s_cmp_eq_u32 %[look], 0\n\t"                                                                         \
    "s_cbranch_scc1 .Lgbwt_gather_nolook_%=\n\t"                                                          \
    "global_load_dwordx3 v[64:66], v[68:69], off offset:96\n\t"                                           \
    ".Lgbwt_gather_nolook_%=:\n\t"                                                                         \
    "s_mov_b64 exec, -1\n\t"
#define GBWT_GATHER_LOOP(KLOAD, DLOAD) \
    asm volatile( \
        "v_mov_b32_e32 v40, %[rec]\n\t" \
        "v_mov_b32_e32 v41, 0\n\t" \
        "v_mov_b32_e32 v42, %[offset]\n\t" \
        "v_mov_b32_e32 v43, %[bb]\n\t" \
        "v_mov_b32_e32 v44, %[wr]\n\t" \
        "v_mov_b32_e32 v59, 0\n\t" \
        "s_mov_b32 %[reason], 0\n\t" \
        "v_readfirstlane_b32 s82, %[patience]\n\t" \
        "s_mov_b64 s[44:45], -1\n\t"                        /* everybody loads in the first round (parked lanes: record 0) */ \
        "v_cmp_ne_u32_e64 s[42:43], 0, v40\n\t" \
        "ds_write_b32 %[mail], v41\n\t"                     /* no look-ahead target */ \
        ".Lgbwt_gather_loop_%=:\n\t" \
        "v_cmp_ne_u32_e32 vcc, -1, v43\n\t"                 /* bb != BLOCK_NONE */ \
        "v_lshrrev_b32_e32 v58, 5, v42\n\t" \
        "v_lshl_add_u32 v58, v43, 1, v58\n\t" \
        "v_cndmask_b32_e32 v58, 0, v58, vcc\n\t"            /* half-block 2 bb + offset / 32, or the zero block */ \
        KLOAD \
        "v_bfm_b32 v74, v42, 0\n\t"                         /* bits below `bit` = offset mod 32 */ \
        "v_and_b32_e32 v77, 0xffffffe0, v42\n\t"            /* offset - bit */ \
        "s_waitcnt vmcnt(0)\n\t" \
        "v_bfe_u32 v72, v60, v42, 1\n\t"                    /* a */ \
        "v_bfe_u32 v84, v61, v42, 1\n\t"                    /* b */ \
        DLOAD \
        "v_add_u32_e32 v76, -1, v72\n\t"                    /* a ? 0 : ~0 */ \
        "v_cmp_eq_u32_e32 vcc, 1, v72\n\t"                  /* vcc = a */ \
        "v_cmp_eq_u32_e64 s[46:47], 1, v84\n\t"             /* s[46:47] = b */ \
        "v_and_b32_e32 v79, 0x1fffff, v62\n\t"              /* ones1 */ \
        "v_bitop3_b32 v78, v60, v76, v74 bitop3:0x28\n\t"   /* m: the equal values of v below `bit` = (bits1 ^ (a ? 0 : ~0)) & below, one three-input operation (gfx950) */ \
        "v_alignbit_b32 v82, v63, v62, 21\n\t" \
        "v_sub_u32_e32 v77, v77, v79\n\t"                   /* (offset - bit) - ones1 */ \
        "v_cndmask_b32_e32 v77, v77, v79, vcc\n\t"          /* a ? ones1 : that */ \
        "v_and_b32_e32 v82, 0x1fffff, v82\n\t"              /* R_0 */ \
        "v_lshrrev_b32_e32 v81, 10, v63\n\t"                /* R_1 */ \
        "v_bcnt_u32_b32 v77, v78, v77\n\t"                  /* rank_a = that + popcount(m) (the count accumulates) */ \
        "v_and_b32_e32 v78, v78, v61\n\t"                   /* a-paths below `bit` with value 1 in w_a */ \
        "v_cndmask_b32_e32 v82, v82, v81, vcc\n\t"          /* R_a */ \
        "v_bcnt_u32_b32 v82, v78, v82\n\t"                  /* ones of w_a before j */ \
        "s_waitcnt vmcnt(0)\n\t" \
        "v_lshlrev_b32_e32 v67, 1, v50\n\t"                 /* DESC2_SLOW (bit 30 of E_a.z) -> sign */ \
        "v_add_u32_e32 v81, v49, v77\n\t"                   /* j: offset in w_a */ \
        "v_bfi_b32 v67, 1, v51, v67\n\t"                    /* ... with GATHER_OK in bit 0 */ \
        "v_sub_u32_e32 v83, v81, v82\n\t"                   /* j - ones */ \
        "v_and_b32_e32 v86, 0x3fffffff, v50\n\t"            /* w_a */ \
        "v_xor_b32_e32 v67, 1, v67\n\t"                     /* sign: slow; bit 0: no packed blocks */ \
        "v_and_b32_e32 v67, 0x80000001, v67\n\t" \
        "v_cmp_ne_u32_e32 vcc, 0, v67\n\t" \
        "v_cndmask_b32_e64 v83, v83, v82, s[46:47]\n\t"     /* rank_b */ \
        "s_nop 0\n\t" \
        "s_and_b64 vcc, vcc, s[42:43]\n\t"                  /* only lanes that walk count (parked ones hold record 0) */ \
        "s_cbranch_vccnz .Lgbwt_gather_slow_%=\n\t" \
        "s_cmp_eq_u32 %[look], 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_gather_nopost_%=\n\t" \
        "ds_write_b96 %[mail], v[64:66]\n\t"               /* mailbox: the look-ahead target of the record just left (v65, v66 are the chain blocks' scratch below) */ \
        ".Lgbwt_gather_nopost_%=:\n\t" \
        "v_mov_b32_e32 v43, v55\n\t"                        /* block base of the landing record */ \
        "v_add_u32_e32 v42, v53, v83\n\t"                   /* the new offset */ \
        "v_and_b32_e32 v40, 0x3fffffff, v54\n\t"            /* the new record */ \
        "v_and_b32_e32 v67, %[ringmask], v44\n\t"           /* ring slot of the next node */ \
        "v_cmp_ne_u32_e32 vcc, 0, v48\n\t" \
        "v_mad_u32_u24 v67, v67, %[stride], %[ring]\n\t" \
        "ds_write_b32 v67, v48\n\t"                         /* node of edge a */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t"       /* counts if it is not the ENDMARKER */ \
        GBWT_CHAIN_BLOCK("e", "", "v_and_b32_e32 v75, 2, v51\n\t" "v_cmp_ne_u32_e32 vcc, 0, v75\n\t", "v48", "v86", "v45", "v65", "v66", "v75") \
        "v_cmp_gt_i32_e32 vcc, 0, v50\n\t"                  /* first step fused? */ \
        "v_and_b32_e32 v67, %[ringmask], v44\n\t" \
        "v_add_u32_e32 v86, s41, v86\n\t"                   /* node of w_a */ \
        "v_mad_u32_u24 v67, v67, %[stride], %[ring]\n\t" \
        "ds_write_b32 v67, v86\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_ne_u32_e32 vcc, 0, v52\n\t" \
        "v_and_b32_e32 v67, %[ringmask], v44\n\t" \
        "v_add_u32_e32 v87, s41, v40\n\t"                   /* node of the landing record */ \
        "v_mad_u32_u24 v67, v67, %[stride], %[ring]\n\t" \
        "ds_write_b32 v67, v52\n\t"                         /* node of the leaf */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        GBWT_CHAIN_BLOCK("l", "", "v_and_b32_e32 v75, 0x40000000, v54\n\t" "v_cmp_ne_u32_e32 vcc, 0, v75\n\t", "v52", "v40", "v45", "v65", "v66", "v75") \
        "v_cmp_gt_i32_e32 vcc, 0, v54\n\t"                  /* second step fused? */ \
        "v_and_b32_e32 v67, %[ringmask], v44\n\t" \
        "v_mad_u32_u24 v67, v67, %[stride], %[ring]\n\t" \
        "ds_write_b32 v67, v87\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_lt_u32_e32 vcc, v44, %[quota]\n\t"            /* a walker that has emitted its share parks */ \
        "s_mov_b64 s[44:45], s[42:43]\n\t"                  /* the next round of loads: everybody who walked in this one */ \
        "ds_write_b32 %[mail], v44 offset:12\n\t"           /* mailbox: nodes staged so far */ \
        "v_cndmask_b32_e32 v40, 0, v40, vcc\n\t" \
        "v_cndmask_b32_e32 v43, -1, v43, vcc\n\t" \
        "v_cmp_lt_u32_e32 vcc, %[limit], v44\n\t"           /* more than slots - 8 nodes waiting in a ring */ \
        "v_cmp_ne_u32_e64 s[42:43], 0, v40\n\t"             /* lanes still walking */ \
        "s_nop 1\n\t" \
        "s_cmp_eq_u64 s[42:43], 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_gather_out_%=\n\t" \
        "s_cbranch_vccz .Lgbwt_gather_loop_%=\n\t" \
        ".Lgbwt_gather_full_%=:\n\t"                          /* over the limit the loop was entered with: how far has the helper come?  A ring that */ \
        "ds_read_b32 v45, %[drained]\n\t"                    /* is still full is waited for here (leaving costs the way out and back in) */ \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_add_u32_e32 v45, %[room], v45\n\t" \
        "v_cmp_lt_u32_e32 vcc, v45, v44\n\t" \
        "s_nop 1\n\t" \
        "s_cbranch_vccnz .Lgbwt_gather_sleep_%=\n\t" \
        "s_sub_u32 s82, s82, 1\n\t"                         /* room again: one round of the caller's patience used; without any left the loop */ \
        "s_cbranch_scc1 .Lgbwt_gather_out_%=\n\t"              /* leaves (the outer loop may have a catch-up to try) */ \
        "s_branch .Lgbwt_gather_loop_%=\n\t" \
        ".Lgbwt_gather_sleep_%=:\n\t" \
        "s_sleep 2\n\t" \
        "s_branch .Lgbwt_gather_full_%=\n\t" \
        ".Lgbwt_gather_slow_%=:\n\t" \
        "s_mov_b32 %[reason], 1\n\t" \
        ".Lgbwt_gather_out_%=:\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
        "v_mov_b32_e32 %[rec], v40\n\t" \
        "v_mov_b32_e32 %[offset], v42\n\t" \
        "v_mov_b32_e32 %[bb], v43\n\t" \
        "v_mov_b32_e32 %[wr], v44\n\t" \
        : [rec] "+v"(rec), [offset] "+v"(offset), [bb] "+v"(bb), [wr] "+v"(wr), [reason] "=&s"(reason) \
        : [desc2] "s"(desc2), [gblocks] "s"(gblocks), [ring] "v"(ring_base), [mail] "v"(mail_slot), [limit] "v"(limit), [quota] "v"(quota), [ringmask] "s"(ring_mask), [stride] "s"(4 * ring_stride), \
          [drained] "v"(drained_addr), [room] "s"(ring_mask + 1 - headroom), [patience] "v"(patience), [look] "s"(lookahead), \
          "{s41}"(alphabet_offset) \
        : "memory", "vcc", "scc", "s42", "s43", "s44", "s45", "s46", "s47", "s80", "s81", "s82", \
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", \
          "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v86", "v87");
    if (narrow) { GBWT_GATHER_LOOP(GBWT_GATHER_K_NARROW, GBWT_GATHER_D_NARROW) } else { GBWT_GATHER_LOOP(GBWT_GATHER_K_WIDE, GBWT_GATHER_D_WIDE) }
#undef GBWT_GATHER_LOOP
#undef GBWT_GATHER_K_NARROW
#undef GBWT_GATHER_K_WIDE
#undef GBWT_GATHER_D_NARROW
#undef GBWT_GATHER_D_WIDE
    return reason;
#endif
}

// The same loop on the full-width blocks (cblocks: 16 + 12 bytes per lane, four loads per iteration), for waves that have met a
// record whose counts do not fit the packed half-blocks (2^21 positions or more).  Registers v40-v87.
__device__ __forceinline__ uint32_t walk2_gather_loop_full(const uint4 *desc2, const uint4 *cblocks, uint32_t alphabet_offset, uint32_t ring_base,
                                                      uint32_t mail_slot, uint32_t flushed, bool narrow, uint32_t quota, uint32_t ring_mask, uint32_t ring_stride,
                                                      uint32_t &rec, uint32_t &offset, uint32_t &bb, uint32_t &wr, uint32_t headroom = 8, uint32_t drained_addr = 0, uint32_t patience = 0) {
#ifdef GBWT_HIP_CXX_LOOP
    __attribute__((address_space(3))) uint32_t *ring = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)ring_base;
    __attribute__((address_space(3))) uint32_t *mail = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)mail_slot;
    mail[0] = 0;                                             // no look-ahead target: the helper's touches would cost what they save
    for (;;) {
        bool slow = false;
        if (rec != 0) {
            const uint64_t idx = bb == BLOCK_NONE ? 0u : bb + (offset >> RANK_BLOCK_SHIFT);
            const uint4 K0 = cblocks[2 * idx];
            const uint32_t *k1 = reinterpret_cast<const uint32_t *>(cblocks + 2 * idx + 1);
            const uint32_t ones1 = k1[0], R0 = k1[1], R1 = k1[2];
            const uint64_t bits1 = (static_cast<uint64_t>(K0.y) << 32) | K0.x, bits2 = (static_cast<uint64_t>(K0.w) << 32) | K0.z;
            const uint32_t bit = offset & 63u;
            const uint64_t below = (uint64_t(1) << bit) - 1;
            const uint32_t a = static_cast<uint32_t>(bits1 >> bit) & 1u, b = static_cast<uint32_t>(bits2 >> bit) & 1u;
            const uint4 *d = desc2 + 8 * static_cast<uint64_t>(rec);
            const uint32_t *e = reinterpret_cast<const uint32_t *>(d + a);
            const uint32_t n1 = e[0], base_a = e[1], wword = e[2], eflags = e[3];
            const uint4 leaf = d[2 + 2 * a + b];
            slow = (wword & DESC2_SLOW) != 0;
            if (!slow) {
                const uint64_t m = a ? bits1 : ~bits1;
                const uint32_t p = __popcll(m & below);
                const uint32_t rank_a = a ? ones1 + p : (offset - bit) - ones1 + p;
                const uint32_t j = base_a + rank_a;
                const uint32_t ones_w = (a ? R1 : R0) + __popcll(m & bits2 & below);
                rec = leaf.z & REC_MASK; offset = leaf.y + (b ? ones_w : j - ones_w); bb = leaf.w;
                ring[(wr & ring_mask) * ring_stride] = n1;
                wr += n1 != 0 ? 1u : 0u;
                if (eflags & E_CHAIN) { const uint32_t L = (wword & REC_MASK) + alphabet_offset, d = chain_stride(n1, L); for (uint32_t n = n1 + d; n != L; n += d) ring[(wr++ & ring_mask) * ring_stride] = n; }
                ring[(wr & ring_mask) * ring_stride] = (wword & REC_MASK) + alphabet_offset;
                wr += (wword & LEAF_EMIT2) ? 1u : 0u;
                ring[(wr & ring_mask) * ring_stride] = leaf.x;
                wr += leaf.x != 0 ? 1u : 0u;
                if (leaf.z & LEAF_CHAIN) { const uint32_t L = rec + alphabet_offset, d = chain_stride(leaf.x, L); for (uint32_t n = leaf.x + d; n != L; n += d) ring[(wr++ & ring_mask) * ring_stride] = n; }
                ring[(wr & ring_mask) * ring_stride] = rec + alphabet_offset;
                wr += (leaf.z & LEAF_EMIT2) ? 1u : 0u;
                mail[3] = wr;
                if (wr >= quota) { rec = 0; bb = BLOCK_NONE; }
            }
        }
        if (__ballot(slow) != 0) return 1;
        if (__ballot(rec != 0) == 0 || __ballot(wr - flushed > ring_mask + 1 - headroom) != 0) return 0;
    }
#else
    // gfx950 assembly: four vector loads per iteration in two rounds (hipcc turns the C++ above into seven in three).  Conventions of walk2_hot_loop; s[44:45] = the lanes that load (those walking at the start of the PREVIOUS
    // iteration, so that a lane that parks fetches record 0 once and keeps emitting nothing), s[42:43] = the lanes walking now.
    uint32_t reason;
    const uint32_t limit = flushed + (ring_mask + 1 - headroom);   // leave with more than slots - 8 nodes waiting
#define GBWT_GATHERF_K_NARROW                                                                              \
    "v_lshlrev_b32_e32 v70, 5, v58\n\t"                   /* two-step blocks are 32 bytes */              \
    "v_lshlrev_b32_e32 v68, 7, v40\n\t"                   /* two-step descriptors are 128 bytes */        \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                        \
    "global_load_dwordx4 v[60:63], v70, %[cblocks]\n\t"             /* K0: bits1, bits2 */                \
    "global_load_dwordx3 v[64:66], v70, %[cblocks] offset:16\n\t"   /* K1: ones1, R0, R1 */               \
    "s_mov_b64 exec, -1\n\t"
#define GBWT_GATHERF_K_WIDE                                                                                \
    "v_lshlrev_b64 v[70:71], 5, v[58:59]\n\t"                                                             \
    "v_lshlrev_b64 v[68:69], 7, v[40:41]\n\t"                                                             \
    "v_lshl_add_u64 v[70:71], v[70:71], 0, %[cblocks]\n\t"                                                \
    "v_lshl_add_u64 v[68:69], v[68:69], 0, %[desc2]\n\t"                                                  \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                        \
    "global_load_dwordx4 v[60:63], v[70:71], off\n\t"                                                     \
    "global_load_dwordx3 v[64:66], v[70:71], off offset:16\n\t"                                           \
    "s_mov_b64 exec, -1\n\t"
// E_a at 16 a, leaf (a, b) at 32 + 16 (2 a + b) of the descriptor; v72 = a, v84 = b
#define GBWT_GATHERF_D_NARROW                                                                              \
    "v_lshl_add_u32 v46, v72, 4, v68\n\t"                                                                 \
    "v_lshl_add_u32 v47, v72, 1, v84\n\t"                                                                \
    "v_lshl_add_u32 v47, v47, 4, v68\n\t"                                                                 \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                        \
    "global_load_dwordx4 v[48:51], v46, %[desc2]\n\t"               /* E_a: node, offset base, w_a | flags, E_CHAIN */ \
    "global_load_dwordx4 v[52:55], v47, %[desc2] offset:32\n\t"     /* leaf (a, b) */                     \
    "s_mov_b64 exec, -1\n\t"
#define GBWT_GATHERF_D_WIDE                                                                                \
    "v_lshl_add_u32 v56, v72, 1, v84\n\t"                                                                \
    "v_mov_b32_e32 v73, 0\n\t"                                                                            \
    "v_mov_b32_e32 v57, 0\n\t"                                                                            \
    "v_lshl_add_u64 v[46:47], v[72:73], 4, v[68:69]\n\t"                                                  \
    "v_lshl_add_u64 v[56:57], v[56:57], 4, v[68:69]\n\t"                                                  \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                        \
    "global_load_dwordx4 v[48:51], v[46:47], off\n\t"                                                     \
    "global_load_dwordx4 v[52:55], v[56:57], off offset:32\n\t"                                           \
    "s_mov_b64 exec, -1\n\t"
#define GBWT_GATHERF_LOOP(KLOAD, DLOAD) \
    asm volatile( \
        "v_mov_b32_e32 v40, %[rec]\n\t" \
        "v_mov_b32_e32 v41, 0\n\t" \
        "v_mov_b32_e32 v42, %[offset]\n\t" \
        "v_mov_b32_e32 v43, %[bb]\n\t" \
        "v_mov_b32_e32 v44, %[wr]\n\t" \
        "v_mov_b32_e32 v59, 0\n\t" \
        "s_mov_b32 %[reason], 0\n\t" \
        "v_readfirstlane_b32 s82, %[patience]\n\t" \
        "s_mov_b64 s[44:45], -1\n\t"                        /* everybody loads in the first round (parked lanes: record 0) */ \
        "v_cmp_ne_u32_e64 s[42:43], 0, v40\n\t" \
        "ds_write_b32 %[mail], v41\n\t"                     /* no look-ahead target */ \
        ".Lgbwt_gatherf_loop_%=:\n\t" \
        "v_cmp_ne_u32_e32 vcc, -1, v43\n\t"                 /* bb != BLOCK_NONE */ \
        "v_lshrrev_b32_e32 v58, 6, v42\n\t" \
        "v_add_u32_e32 v58, v58, v43\n\t" \
        "v_cndmask_b32_e32 v58, 0, v58, vcc\n\t"            /* block bb + offset / 64, or the zero block */ \
        KLOAD \
        "v_lshlrev_b64 v[74:75], v42, -1\n\t"               /* bits at and above `bit` */ \
        "v_and_b32_e32 v77, 0xffffffc0, v42\n\t"            /* offset - bit */ \
        "s_waitcnt vmcnt(0)\n\t" \
        "v_lshrrev_b64 v[72:73], v42, v[60:61]\n\t"         /* bits1 >> bit */ \
        "v_lshrrev_b64 v[84:85], v42, v[62:63]\n\t"       /* bits2 >> bit */ \
        "v_and_b32_e32 v72, 1, v72\n\t"                     /* a */ \
        "v_and_b32_e32 v84, 1, v84\n\t"                   /* b */ \
        DLOAD \
        "v_add_u32_e32 v76, -1, v72\n\t"                    /* a ? 0 : ~0 */ \
        "v_cmp_eq_u32_e32 vcc, 1, v72\n\t"                  /* vcc = a */ \
        "v_cmp_eq_u32_e64 s[46:47], 1, v84\n\t"            /* s[46:47] = b */ \
        "v_xor_b32_e32 v78, v60, v76\n\t"                  /* m = a ? bits1 : ~bits1 */ \
        "v_xor_b32_e32 v79, v61, v76\n\t" \
        "v_bfi_b32 v78, v74, 0, v78\n\t"                  /* m below `bit` */ \
        "v_bfi_b32 v79, v75, 0, v79\n\t" \
        "v_sub_u32_e32 v77, v77, v64\n\t"                   /* (offset - bit) - ones1 */ \
        "v_bcnt_u32_b32 v80, v78, 0\n\t" \
        "v_cndmask_b32_e32 v77, v77, v64, vcc\n\t"          /* a ? ones1 : that */ \
        "v_bcnt_u32_b32 v80, v79, v80\n\t"               /* p */ \
        "v_cndmask_b32_e32 v82, v65, v66, vcc\n\t"         /* R_a */ \
        "v_add_u32_e32 v77, v77, v80\n\t"                  /* rank_a */ \
        "v_and_b32_e32 v78, v78, v62\n\t"                 /* a-paths below `bit` with value 1 in w_a */ \
        "v_and_b32_e32 v79, v79, v63\n\t" \
        "v_bcnt_u32_b32 v82, v78, v82\n\t" \
        "v_bcnt_u32_b32 v82, v79, v82\n\t"               /* ones of w_a before j */ \
        "s_waitcnt vmcnt(0)\n\t" \
        "v_lshlrev_b32_e32 v67, 1, v50\n\t"                 /* DESC2_SLOW (bit 30 of E_a.z) -> sign */ \
        "v_add_u32_e32 v81, v49, v77\n\t"                  /* j: offset in w_a */ \
        "v_cmp_gt_i32_e32 vcc, 0, v67\n\t" \
        "v_sub_u32_e32 v83, v81, v82\n\t"                /* j - ones */ \
        "v_and_b32_e32 v86, 0x3fffffff, v50\n\t"           /* w_a */ \
        "s_cbranch_vccnz .Lgbwt_gatherf_slow_%=\n\t" \
        "v_cndmask_b32_e64 v83, v83, v82, s[46:47]\n\t"  /* rank_b */ \
        "v_mov_b32_e32 v43, v55\n\t"                        /* block base of the landing record */ \
        "v_add_u32_e32 v42, v53, v83\n\t"                  /* the new offset */ \
        "v_and_b32_e32 v40, 0x3fffffff, v54\n\t"            /* the new record */ \
        "v_and_b32_e32 v67, %[ringmask], v44\n\t"           /* ring slot of the next node */ \
        "v_cmp_ne_u32_e32 vcc, 0, v48\n\t" \
        "v_mad_u32_u24 v67, v67, %[stride], %[ring]\n\t" \
        "ds_write_b32 v67, v48\n\t"                         /* node of edge a */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t"       /* counts if it is not the ENDMARKER */ \
        GBWT_CHAIN_BLOCK("e", "", "v_and_b32_e32 v75, 2, v51\n\t" "v_cmp_ne_u32_e32 vcc, 0, v75\n\t", "v48", "v86", "v45", "v65", "v66", "v75") \
        "v_cmp_gt_i32_e32 vcc, 0, v50\n\t"                  /* first step fused? */ \
        "v_and_b32_e32 v67, %[ringmask], v44\n\t" \
        "v_add_u32_e32 v86, s41, v86\n\t"                 /* node of w_a */ \
        "v_mad_u32_u24 v67, v67, %[stride], %[ring]\n\t" \
        "ds_write_b32 v67, v86\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_ne_u32_e32 vcc, 0, v52\n\t" \
        "v_and_b32_e32 v67, %[ringmask], v44\n\t" \
        "v_add_u32_e32 v87, s41, v40\n\t"                  /* node of the landing record */ \
        "v_mad_u32_u24 v67, v67, %[stride], %[ring]\n\t" \
        "ds_write_b32 v67, v52\n\t"                         /* node of the leaf */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        GBWT_CHAIN_BLOCK("l", "", "v_and_b32_e32 v75, 0x40000000, v54\n\t" "v_cmp_ne_u32_e32 vcc, 0, v75\n\t", "v52", "v40", "v45", "v65", "v66", "v75") \
        "v_cmp_gt_i32_e32 vcc, 0, v54\n\t"                  /* second step fused? */ \
        "v_and_b32_e32 v67, %[ringmask], v44\n\t" \
        "v_mad_u32_u24 v67, v67, %[stride], %[ring]\n\t" \
        "ds_write_b32 v67, v87\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_lt_u32_e32 vcc, v44, %[quota]\n\t"            /* a walker that has emitted its share parks */ \
        "s_mov_b64 s[44:45], s[42:43]\n\t"                  /* the next round of loads: everybody who walked in this one */ \
        "ds_write_b32 %[mail], v44 offset:12\n\t"           /* mailbox: nodes staged so far */ \
        "v_cndmask_b32_e32 v40, 0, v40, vcc\n\t" \
        "v_cndmask_b32_e32 v43, -1, v43, vcc\n\t" \
        "v_cmp_lt_u32_e32 vcc, %[limit], v44\n\t"           /* more than slots - 8 nodes waiting in a ring */ \
        "v_cmp_ne_u32_e64 s[42:43], 0, v40\n\t"             /* lanes still walking */ \
        "s_nop 1\n\t" \
        "s_cmp_eq_u64 s[42:43], 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_gatherf_out_%=\n\t" \
        "s_cbranch_vccz .Lgbwt_gatherf_loop_%=\n\t" \
        ".Lgbwt_gatherf_full_%=:\n\t"                          /* over the limit the loop was entered with: how far has the helper come?  A ring that */ \
        "ds_read_b32 v45, %[drained]\n\t"                    /* is still full is waited for here (leaving costs the way out and back in) */ \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_add_u32_e32 v45, %[room], v45\n\t" \
        "v_cmp_lt_u32_e32 vcc, v45, v44\n\t" \
        "s_nop 1\n\t" \
        "s_cbranch_vccnz .Lgbwt_gatherf_sleep_%=\n\t" \
        "s_sub_u32 s82, s82, 1\n\t"                         /* room again: one round of the caller's patience used; without any left the loop */ \
        "s_cbranch_scc1 .Lgbwt_gatherf_out_%=\n\t"              /* leaves (the outer loop may have a catch-up to try) */ \
        "s_branch .Lgbwt_gatherf_loop_%=\n\t" \
        ".Lgbwt_gatherf_sleep_%=:\n\t" \
        "s_sleep 2\n\t" \
        "s_branch .Lgbwt_gatherf_full_%=\n\t" \
        ".Lgbwt_gatherf_slow_%=:\n\t" \
        "s_mov_b32 %[reason], 1\n\t" \
        ".Lgbwt_gatherf_out_%=:\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
        "v_mov_b32_e32 %[rec], v40\n\t" \
        "v_mov_b32_e32 %[offset], v42\n\t" \
        "v_mov_b32_e32 %[bb], v43\n\t" \
        "v_mov_b32_e32 %[wr], v44\n\t" \
        : [rec] "+v"(rec), [offset] "+v"(offset), [bb] "+v"(bb), [wr] "+v"(wr), [reason] "=&s"(reason) \
        : [desc2] "s"(desc2), [cblocks] "s"(cblocks), [ring] "v"(ring_base), [mail] "v"(mail_slot), [limit] "v"(limit), [quota] "v"(quota), [ringmask] "s"(ring_mask), [stride] "s"(4 * ring_stride), \
          [drained] "v"(drained_addr), [room] "s"(ring_mask + 1 - headroom), [patience] "v"(patience), \
          "{s41}"(alphabet_offset) \
        : "memory", "vcc", "scc", "s42", "s43", "s44", "s45", "s46", "s47", "s80", "s81", "s82", \
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", \
          "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v68", "v69", "v70", "v71", "v67", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", \
          "v82", "v83", "v84", "v85", "v86", "v87");
    if (narrow) { GBWT_GATHERF_LOOP(GBWT_GATHERF_K_NARROW, GBWT_GATHERF_D_NARROW) } else { GBWT_GATHERF_LOOP(GBWT_GATHERF_K_WIDE, GBWT_GATHERF_D_WIDE) }
#undef GBWT_GATHERF_LOOP
#undef GBWT_GATHERF_K_NARROW
#undef GBWT_GATHERF_K_WIDE
#undef GBWT_GATHERF_D_NARROW
#undef GBWT_GATHERF_D_WIDE
    return reason;
#endif
}


// ---- two-step walk, wave-uniform variant ----------------------------------------------------------------------
// With sixty-four lanes walking, the vector-memory path is what the walk saturates (TA / TD busy 92-95 % of the kernel,
// profiles/r01_coop_pmc_headline.txt): every lane fetches its own copy of the 104 descriptor bytes, 16 bytes per lane
// per load instruction.  The walkers of a wave hold the same segment of neighbouring rows, and haplotypes travel
// together, so very often ALL lanes sit on the same record: then the descriptor is one scalar fetch (s_load, 28 SGPRs)
// and only the two-step rank block -- the one thing that differs between lanes -- goes through the vector path
// (28 bytes per lane instead of 132).  gfx9 VALU instructions read at most one SGPR, so instead of v_cndmask the
// per-lane choices are made by running the same `v_mov / v_add  vgpr, sgpr` under the exec mask of each choice.
// Conventions of walk2_hot_loop, registers v40-v81.  Leaves with reason 2 -- nothing in flight, state intact -- as
// soon as the lanes are not all on one record (or some are parked); the caller then continues with the loop for mixed waves.
// SGPRs: s[48:63] E_0 E_1 L00 L01, s[64:71] L10 L11, s[72:75] look-ahead target, s[76:85] masks, s[88:89] descriptor
// address, s78 the record.
#ifndef GBWT_HIP_CXX_LOOP
#define GBWT_WALK2U_ISSUE(KLOAD, REFRESH)                                                                             \
    "v_readfirstlane_b32 s78, v40\n\t"                    /* the record of lane 0 */                       \
    "v_cmp_ne_u32_e32 vcc, -1, v43\n\t"                   /* bb != BLOCK_NONE */                          \
    GBWT_WALK2U_INDEX                                                                                     \
    "s_lshl_b32 s76, s78, 7\n\t"                          /* two-step descriptors are 128 bytes */        \
    "s_lshr_b32 s77, s78, 25\n\t"                                                                         \
    "v_cndmask_b32_e32 v46, 0, v46, vcc\n\t"              /* block bb + offset / 64, or the zero block */ \
    "s_add_u32 s76, s76, %[dlo]\n\t"                                                                      \
    "s_addc_u32 s77, s77, %[dhi]\n\t"                                                                     \
    GBWT_WALK2U_BYTES                                                                                     \
    "v_cmp_ne_u32_e32 vcc, s78, v40\n\t"                  /* lanes on another record */                   \
    "s_load_dwordx16 s[48:63], s[76:77], 0x0\n\t"         /* E_0, E_1, leaf (0, 0), leaf (0, 1) */         \
    "s_load_dwordx8 s[64:71], s[76:77], 0x40\n\t"         /* leaf (1, 0), leaf (1, 1) */                  \
    "s_load_dwordx4 s[72:75], s[76:77], 0x60\n\t"         /* look-ahead target */                         \
    KLOAD                                                                                                 \
    REFRESH
// K0 = {bits1, bits2}, K1 = {ones1, R0, R1} of the lane's two-step block: SGPR base + 32-bit byte offset while the block
// array is below 4 GiB, a 64-bit address per lane above
// A leaf {node to emit, offset base, landing record | flags, its block base} lies in four consecutive SGPRs: two 64-bit moves per leaf
// (v_mov_b64 from an SGPR pair runs at the rate of v_mov_b32, tools/microbench_mov64.hip) instead of three moves and an add -- the offset
// base lands in v81 and the landing word in v42 (the old offset is dead by now), GBWT_WALK2U_LEAVES_DONE puts them where the rest of the loop
// expects them with the add done once for all four leaves: 10 VALU instructions for the selection instead of 16 (round 6).
#define GBWT_WALK2U_LEAF(MASK, XY, ZW)                                                                     \
    MASK "\n\t"                                                                                           \
    "v_mov_b64 v[80:81], " XY "\n\t"                     /* node to emit, offset base */                 \
    "v_mov_b64 v[42:43], " ZW "\n\t"                     /* landing record | flags, its block base */
#define GBWT_WALK2U_LEAVES_DONE                                                                           \
    "v_mov_b32_e32 v87, v42\n\t"                         /* landing record | flags */                    \
    "v_add_u32_e32 v42, v81, v75\n\t"                    /* the new offset = offset base + rank_b */
// The up to four nodes of an iteration into the ring (v72 / v78 first step, v80 / v40 second step); CHAIN_E / CHAIN_L = the nodes between them
// where a step is chained (GBWT_CHAIN_BLOCK), or nothing.
#define GBWT_WALK2U_STAGE(CHAIN_E, CHAIN_L) \
        "v_and_b32_e32 v47, %[ringmask], v44\n\t"           /* ring slot of the next node */ \
        "v_cmp_ne_u32_e32 vcc, 0, v72\n\t" \
        "v_mad_u32_u24 v47, v47, %[stride], %[ring]\n\t" \
        "ds_write_b32 v47, v72\n\t"                        /* node of edge a */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t"       /* counts if it is not the ENDMARKER */ \
        CHAIN_E \
        "v_cmp_gt_i32_e32 vcc, 0, v73\n\t"                 /* first step fused? */ \
        "v_and_b32_e32 v47, %[ringmask], v44\n\t" \
        "v_add_u32_e32 v78, s41, v78\n\t"                 /* node of w_a */ \
        "v_mad_u32_u24 v47, v47, %[stride], %[ring]\n\t" \
        "ds_write_b32 v47, v78\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        "v_cmp_ne_u32_e32 vcc, 0, v80\n\t" \
        "v_and_b32_e32 v47, %[ringmask], v44\n\t" \
        "v_add_u32_e32 v79, s41, v40\n\t"                  /* node of the landing record */ \
        "v_mad_u32_u24 v47, v47, %[stride], %[ring]\n\t" \
        "ds_write_b32 v47, v80\n\t"                        /* node of the leaf */ \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t" \
        CHAIN_L \
        "v_cmp_gt_i32_e32 vcc, 0, v87\n\t"                 /* second step fused? */ \
        "v_and_b32_e32 v47, %[ringmask], v44\n\t" \
        "v_mad_u32_u24 v47, v47, %[stride], %[ring]\n\t" \
        "ds_write_b32 v47, v79\n\t" \
        "v_addc_co_u32_e32 v44, vcc, 0, v44, vcc\n\t"
#define GBWT_WALK2U_LOOP(KLOAD, REFRESH) \
    asm volatile( \
        "v_mov_b32_e32 v40, %[rec]\n\t" \
        "v_mov_b32_e32 v42, %[offset]\n\t" \
        "v_mov_b32_e32 v43, %[bb]\n\t" \
        "v_mov_b32_e32 v44, %[wr]\n\t" \
        "s_mov_b32 %[reason], 0\n\t" \
        GBWT_WALK2U_ISSUE(KLOAD, REFRESH) \
        "s_nop 1\n\t" \
        "s_cbranch_vccnz .Lgbwt_walk2u_mixed_%=\n\t" \
        "s_cmp_eq_u32 s78, 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_walk2u_out_%=\n\t" \
        ".Lgbwt_walk2u_loop_%=:\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
        "s_bitcmp1_b32 s50, 30\n\t"                         /* DESC2_SLOW */ \
        "s_cbranch_scc1 .Lgbwt_walk2u_slow_%=\n\t" \
        "s_and_b32 s82, s51, %[flagmask]\n\t"               /* E_ANYCHAIN: some step of this record is chained; E_ALL4: every iteration on it stages four nodes (kept, with the flag words of the two edges: */ \
        "s_mov_b32 s83, s51\n\t"                            /*  the SGPRs are reloaded before the nodes are staged) */ \
        "s_mov_b32 s84, s55\n\t" \
        GBWT_WALK2U_FLAG_CHECK \
        GBWT_WALK2U_RANK_A \
        "v_mov_b32_e32 v72, s48\n\t"                       /* edge 0: node */ \
        "v_mov_b32_e32 v73, s50\n\t"                       /*         w_0 | flags */ \
        "v_add_u32_e32 v71, s49, v67\n\t"                  /*         j = offset base + rank_a: offset in w_a */ \
        "s_mov_b64 exec, s[44:45]\n\t"                      /* the lanes that take edge 1 */ \
        "v_mov_b32_e32 v72, s52\n\t" \
        "v_mov_b32_e32 v73, s54\n\t" \
        "v_add_u32_e32 v71, s53, v67\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        GBWT_WALK2U_RANK_B \
        "v_sub_u32_e32 v75, v71, v74\n\t"                /* j - ones */ \
        "v_and_b32_e32 v78, 0x3fffffff, v73\n\t"          /* w_a */ \
        "v_cndmask_b32_e64 v75, v75, v74, s[46:47]\n\t"  /* rank_b */ \
        GBWT_WALK2U_LEAF("s_nor_b64 exec, s[44:45], s[46:47]", "s[56:57]", "s[58:59]")     /* lanes of leaf (0, 0) */ \
        GBWT_WALK2U_LEAF("s_andn2_b64 exec, s[46:47], s[44:45]", "s[60:61]", "s[62:63]")   /*          leaf (0, 1) */ \
        GBWT_WALK2U_LEAF("s_andn2_b64 exec, s[44:45], s[46:47]", "s[64:65]", "s[66:67]")   /*          leaf (1, 0) */ \
        GBWT_WALK2U_LEAF("s_and_b64 exec, s[44:45], s[46:47]", "s[68:69]", "s[70:71]")     /*          leaf (1, 1) */ \
        "s_mov_b64 exec, -1\n\t" \
        GBWT_WALK2U_LEAVES_DONE \
        "v_and_b32_e32 v40, 0x3fffffff, v87\n\t"           /* the new record */ \
        "v_mov_b64 v[48:49], s[72:73]\n\t"                  /* mailbox: look-ahead target of the record just left (before its SGPRs are reloaded) ... */ \
        "v_mov_b32_e32 v50, s74\n\t" \
        GBWT_WALK2U_MAIL_FLAG \
        GBWT_WALK2U_ISSUE(KLOAD, "")                        /* the loads of the next position go out now; staging the nodes runs underneath them */ \
        "s_nop 1\n\t" \
        "s_mov_b64 s[44:45], vcc\n\t"                       /* lanes that are not on the record of lane 0 */ \
        "s_cmp_lg_u32 s82, 0\n\t"                          /* a record with chained steps, or one whose iterations all stage four nodes: out of line (below) */ \
        "s_cbranch_scc1 .Lgbwt_walk2u_special_%=\n\t" \
        ".Lgbwt_walk2u_plain_%=:\n\t" \
        GBWT_WALK2U_STAGE("", "") \
        ".Lgbwt_walk2u_staged_%=:\n\t" \
        "v_cmp_lt_u32_e32 vcc, v44, %[quota]\n\t"           /* a walker that has emitted its share parks */ \
        "v_mov_b32_e32 v51, v44\n\t"                        /* ... + nodes staged so far */ \
        "ds_write_b128 %[mail], v[48:51]\n\t" \
        "v_cndmask_b32_e32 v40, 0, v40, vcc\n\t" \
        "v_cndmask_b32_e32 v43, -1, v43, vcc\n\t" \
        "s_andn2_b64 s[46:47], exec, vcc\n\t"               /* lanes that have just parked: their loads were for nothing, and the wave is no longer on one record */ \
        "v_sub_u32_e32 v47, v44, v45\n\t"                   /* nodes waiting in the ring (drained as of the last iteration) */ \
        "s_or_b64 s[44:45], s[44:45], s[46:47]\n\t" \
        REFRESH \
        "v_cmp_lt_u32_e64 s[46:47], %[slack], v47\n\t"      /* more than slots - 8 of them */ \
        "s_cmp_lg_u64 s[44:45], 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_walk2u_mixed_%=\n\t" \
        "s_cmp_eq_u32 s78, 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_walk2u_out_%=\n\t"           /* everyone has parked */ \
        "s_cmp_eq_u64 s[46:47], 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_walk2u_loop_%=\n\t" \
        ".Lgbwt_walk2u_full_%=:\n\t"                        /* a ring is full: wait HERE for the helper -- the loads of the next position are out and stay */ \
        "s_sleep 2\n\t"                                     /* valid; leaving meant a wasted round of loads and another one on the way back in */ \
        "ds_read_b32 v45, %[drained]\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_sub_u32_e32 v47, v44, v45\n\t" \
        "v_cmp_lt_u32_e64 s[46:47], %[slack], v47\n\t" \
        "s_nop 1\n\t" \
        "s_cmp_eq_u64 s[46:47], 0\n\t" \
        "s_cbranch_scc0 .Lgbwt_walk2u_full_%=\n\t" \
        "s_branch .Lgbwt_walk2u_loop_%=\n\t" \
        ".Lgbwt_walk2u_special_%=:\n\t" \
        "s_bitcmp1_b32 s82, 2\n\t"                          /* E_ANYCHAIN */ \
        "s_cbranch_scc1 .Lgbwt_walk2u_chained_%=\n\t" \
        /* E_ALL4 (k_link_desc2): whichever edge and leaf a lane takes on this record, it emits four nodes -- neither an ENDMARKER nor an unfused \
           step -- so nothing has to be counted: four consecutive ring slots (unless they wrap), one address, wr += 4.  6 VALU instead of 18. */ \
        "v_and_b32_e32 v47, %[ringmask], v44\n\t" \
        "v_cmp_lt_u32_e64 s[46:47], %[wrapmax], v47\n\t"    /* slot > slots - 4: the four would wrap */ \
        "s_nop 1\n\t" \
        "s_cmp_lg_u64 s[46:47], 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_walk2u_plain_%=\n\t" \
        "v_mad_u32_u24 v47, v47, %[stride], %[ring]\n\t" \
        "v_add_u32_e32 v78, s41, v78\n\t"                 /* node of w_a */ \
        "v_add_u32_e32 v79, s41, v40\n\t"                  /* node of the landing record */ \
        "ds_write_b32 v47, v72\n\t" \
        "ds_write_b32 v47, v78 offset:260\n\t"              /* (slot + k) * 4 * RING_PITCH: the pitch is 65 dwords (walk_direct.hip; flagmask drops E_ALL4 for any other) */ \
        "ds_write_b32 v47, v80 offset:520\n\t" \
        "ds_write_b32 v47, v79 offset:780\n\t" \
        "v_add_u32_e32 v44, 4, v44\n\t" \
        "s_branch .Lgbwt_walk2u_staged_%=\n\t" \
        ".Lgbwt_walk2u_chained_%=:\n\t"                     /* the same with the nodes between; v62 = a */ \
        "v_mov_b32_e32 v82, s83\n\t" \
        "v_mov_b32_e32 v86, s84\n\t" \
        "v_cmp_eq_u32_e32 vcc, 1, v62\n\t" \
        "v_cndmask_b32_e32 v82, v82, v86, vcc\n\t"          /* E_a.w of the lane's edge */ \
        GBWT_WALK2U_STAGE(GBWT_CHAIN_BLOCK("e", "", "v_and_b32_e32 v86, 2, v82\n\t" "v_cmp_ne_u32_e32 vcc, 0, v86\n\t", "v72", "v78", "v83", "v84", "v85", "v86"), \
                          GBWT_CHAIN_BLOCK("l", "", "v_and_b32_e32 v86, 0x40000000, v87\n\t" "v_cmp_ne_u32_e32 vcc, 0, v86\n\t", "v80", "v40", "v83", "v84", "v85", "v86")) \
        "s_branch .Lgbwt_walk2u_staged_%=\n\t" \
        ".Lgbwt_walk2u_slow_%=:\n\t" \
        "s_mov_b32 %[reason], 1\n\t" \
        "s_branch .Lgbwt_walk2u_out_%=\n\t" \
        ".Lgbwt_walk2u_unpacked_%=:\n\t" \
        "s_mov_b32 %[reason], 3\n\t" \
        "s_branch .Lgbwt_walk2u_out_%=\n\t" \
        ".Lgbwt_walk2u_mixed_%=:\n\t" \
        "s_mov_b32 %[reason], 2\n\t" \
        ".Lgbwt_walk2u_out_%=:\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
        "v_mov_b32_e32 %[rec], v40\n\t" \
        "v_mov_b32_e32 %[offset], v42\n\t" \
        "v_mov_b32_e32 %[bb], v43\n\t" \
        "v_mov_b32_e32 %[wr], v44\n\t" \
        : [rec] "+v"(rec), [offset] "+v"(offset), [bb] "+v"(bb), [wr] "+v"(wr), [reason] "=&s"(reason) \
        : [dlo] "s"(dlo), [dhi] "s"(dhi), [cblocks] "s"(cblocks), [ring] "v"(ring_base), [mail] "v"(mail_slot), [drained] "v"(drained_addr), [slack] "s"(slack), [quota] "v"(quota), \
          [ringmask] "s"(ring_mask), [stride] "s"(4 * ring_stride), [flagmask] "s"(flagmask), [wrapmax] "s"(ring_mask - 3), "{s41}"(alphabet_offset) \
        : "memory", "vcc", "scc", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", \
          "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", \
          "v40", "v42", "v43", "v44", "v45", "v46", "v61", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v60", "v47", "v62", "v63", \
          "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", \
          "s80", "s81", "s82", "s83", "s84");
#define GBWT_WALK2U_BODY \
    uint32_t reason; \
    const uint32_t slack = ring_mask + 1 - headroom;   /* leave with more than slots - headroom nodes waiting in a ring (8; 16 where steps are chained) */ \
    const uint32_t flagmask = (ring_stride == 65u && all4) ? (E_ANYCHAIN | E_ALL4) : E_ANYCHAIN;   /* the four-in-a-row staging hardcodes the ring pitch */ \
    const uint32_t dlo = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(desc2)), dhi = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(desc2) >> 32); \
    /* the loop re-reads how far the helper has emptied the ring every iteration (working with the count it was entered with, it had to \
       leave after (slots - 8 - waiting) / 4 iterations: 180 exits per 1 024 iterations, each with a wasted round of loads) */ \
    if (narrow) { GBWT_WALK2U_LOOP(GBWT_WALK2U_KLOAD_NARROW, "ds_read_b32 v45, %[drained]\n\t") } \
    else { GBWT_WALK2U_LOOP(GBWT_WALK2U_KLOAD_WIDE, "ds_read_b32 v45, %[drained]\n\t") } \
    return reason;
// The loop on the packed half-blocks (gblocks: one 16-byte load per lane and iteration instead of 16 + 12, 32-bit rank arithmetic; + 9 % on
// the headline, profiles/r02_walk_bounds.txt #21).  Leaves with reason 3 on a record whose counts do not fit them (no GATHER_OK).
#define GBWT_WALK2U_INDEX                                                                                 \
    "v_lshrrev_b32_e32 v46, 5, v42\n\t"                                                                   \
    "v_lshl_add_u32 v46, v43, 1, v46\n\t"
#define GBWT_WALK2U_BYTES "v_lshlrev_b32_e32 v60, 4, v46\n\t"
#define GBWT_WALK2U_RANK_A \
        "v_bfe_u32 v62, v52, v42, 1\n\t"                    /* a */ \
        "v_bfm_b32 v64, v42, 0\n\t"                         /* bits below `bit` */ \
        "v_and_b32_e32 v67, 0xffffffe0, v42\n\t"            /* offset - bit */ \
        "v_and_b32_e32 v56, 0x1fffff, v54\n\t"              /* ones1 */ \
        "v_add_u32_e32 v66, -1, v62\n\t"                    /* a ? 0 : ~0 */ \
        "v_cmp_eq_u32_e32 vcc, 1, v62\n\t"                  /* vcc = a */ \
        "v_xor_b32_e32 v68, v52, v66\n\t"                   /* a ? bits1 : ~bits1 */ \
        "v_alignbit_b32 v57, v55, v54, 21\n\t" \
        "v_and_b32_e32 v68, v68, v64\n\t"                   /* m */ \
        "v_sub_u32_e32 v67, v67, v56\n\t"                   /* (offset - bit) - ones1 */ \
        "v_bcnt_u32_b32 v70, v68, 0\n\t"                    /* p */ \
        "v_cndmask_b32_e32 v67, v67, v56, vcc\n\t"          /* a ? ones1 : that */ \
        "v_and_b32_e32 v57, 0x1fffff, v57\n\t"              /* R_0 */ \
        "v_lshrrev_b32_e32 v58, 10, v55\n\t"                /* R_1 */ \
        "s_mov_b64 s[44:45], vcc\n\t"                       /* a */ \
        "v_add_u32_e32 v67, v67, v70\n\t"                   /* rank_a */ \
        "v_cndmask_b32_e32 v74, v57, v58, vcc\n\t"          /* R_a */ \
        "v_bfe_u32 v76, v53, v42, 1\n\t"                    /* b */
#define GBWT_WALK2U_RANK_B \
        "v_and_b32_e32 v68, v68, v53\n\t"                   /* a-paths below `bit` with value 1 in w_a */ \
        "v_cmp_eq_u32_e64 s[46:47], 1, v76\n\t"             /* s[46:47] = b */ \
        "v_bcnt_u32_b32 v74, v68, v74\n\t"                  /* ones of w_a before j */
#define GBWT_WALK2U_KLOAD_NARROW                                                                          \
    "global_load_dwordx4 v[52:55], v60, %[cblocks]\n\t"
#define GBWT_WALK2U_KLOAD_WIDE                                                                            \
    "v_lshrrev_b32_e32 v61, 28, v46\n\t"                                                                  \
    "v_lshl_add_u64 v[60:61], v[60:61], 0, %[cblocks]\n\t"                                                \
    "global_load_dwordx4 v[52:55], v[60:61], off\n\t"
#define GBWT_WALK2U_FLAG_CHECK \
        "s_bitcmp0_b32 s51, 0\n\t"                          /* GATHER_OK (E_0.w) */ \
        "s_cbranch_scc1 .Lgbwt_walk2u_unpacked_%=\n\t"
#define GBWT_WALK2U_MAIL_FLAG
#endif
__device__ __forceinline__ uint32_t walk2_uniform_loop(const uint4 *desc2, const uint4 *cblocks /* = gblocks */, uint32_t alphabet_offset, uint32_t ring_base,
                                                       uint32_t mail_slot, uint32_t drained_addr, bool narrow, uint32_t quota, uint32_t ring_mask,
                                                       uint32_t ring_stride, uint32_t &rec, uint32_t &offset, uint32_t &bb, uint32_t &wr, uint32_t headroom = 8, bool all4 = true) {
#ifdef GBWT_HIP_CXX_LOOP
    return 2;
#else
    GBWT_WALK2U_BODY
#endif
}
#ifndef GBWT_HIP_CXX_LOOP
#undef GBWT_WALK2U_INDEX
#undef GBWT_WALK2U_BYTES
#undef GBWT_WALK2U_RANK_A
#undef GBWT_WALK2U_RANK_B
#undef GBWT_WALK2U_KLOAD_WIDE
#undef GBWT_WALK2U_KLOAD_NARROW
#undef GBWT_WALK2U_FLAG_CHECK
#undef GBWT_WALK2U_MAIL_FLAG
// The same loop on the full-width blocks (cblocks), for waves that have met such a record; bit 31 of the look-ahead count tells the
// helper which array to touch.
#define GBWT_WALK2U_INDEX                                                                                 \
    "v_lshrrev_b32_e32 v46, 6, v42\n\t"                                                                   \
    "v_add_u32_e32 v46, v46, v43\n\t"
#define GBWT_WALK2U_BYTES "v_lshlrev_b32_e32 v60, 5, v46\n\t"                   /* two-step blocks are 32 bytes */
#define GBWT_WALK2U_RANK_A \
        "v_lshrrev_b64 v[62:63], v42, v[52:53]\n\t"         /* bits1 >> bit */ \
        "v_lshlrev_b64 v[64:65], v42, -1\n\t"               /* bits at and above `bit` */ \
        "v_and_b32_e32 v62, 1, v62\n\t"                     /* a */ \
        "v_and_b32_e32 v67, 0xffffffc0, v42\n\t"            /* offset - bit */ \
        "v_add_u32_e32 v66, -1, v62\n\t"                    /* a ? 0 : ~0 */ \
        "v_cmp_eq_u32_e32 vcc, 1, v62\n\t"                  /* vcc = a */ \
        "v_xor_b32_e32 v68, v52, v66\n\t"                  /* m = a ? bits1 : ~bits1 */ \
        "v_xor_b32_e32 v69, v53, v66\n\t" \
        "v_bfi_b32 v68, v64, 0, v68\n\t"                  /* m below `bit` */ \
        "v_bfi_b32 v69, v65, 0, v69\n\t" \
        "v_sub_u32_e32 v67, v67, v56\n\t"                   /* (offset - bit) - ones1 */ \
        "v_bcnt_u32_b32 v70, v68, 0\n\t" \
        "v_cndmask_b32_e32 v67, v67, v56, vcc\n\t"          /* a ? ones1 : that */ \
        "v_bcnt_u32_b32 v70, v69, v70\n\t"               /* p */ \
        "v_cndmask_b32_e32 v74, v57, v58, vcc\n\t"         /* R_a */ \
        "s_mov_b64 s[44:45], vcc\n\t"                       /* a */ \
        "v_add_u32_e32 v67, v67, v70\n\t"                  /* rank_a */ \
        "v_lshrrev_b64 v[76:77], v42, v[54:55]\n\t"       /* bits2 >> bit */
#define GBWT_WALK2U_RANK_B \
        "v_and_b32_e32 v68, v68, v54\n\t"                 /* a-paths below `bit` with value 1 in w_a */ \
        "v_and_b32_e32 v69, v69, v55\n\t" \
        "v_and_b32_e32 v76, 1, v76\n\t"                   /* b */ \
        "v_bcnt_u32_b32 v74, v68, v74\n\t" \
        "v_cmp_eq_u32_e64 s[46:47], 1, v76\n\t"            /* s[46:47] = b */ \
        "v_bcnt_u32_b32 v74, v69, v74\n\t"               /* ones of w_a before j */
#define GBWT_WALK2U_KLOAD_NARROW                                                                          \
    "global_load_dwordx4 v[52:55], v60, %[cblocks]\n\t"                                                   \
    "global_load_dwordx3 v[56:58], v60, %[cblocks] offset:16\n\t"
#define GBWT_WALK2U_KLOAD_WIDE                                                                            \
    "v_lshrrev_b32_e32 v61, 27, v46\n\t"                                                                  \
    "v_lshl_add_u64 v[60:61], v[60:61], 0, %[cblocks]\n\t"                                                \
    "global_load_dwordx4 v[52:55], v[60:61], off\n\t"                                                     \
    "global_load_dwordx3 v[56:58], v[60:61], off offset:16\n\t"
#define GBWT_WALK2U_FLAG_CHECK
#define GBWT_WALK2U_MAIL_FLAG "v_or_b32_e32 v50, 0x80000000, v50\n\t"
#endif
__device__ __forceinline__ uint32_t walk2_uniform_loop_full(const uint4 *desc2, const uint4 *cblocks, uint32_t alphabet_offset, uint32_t ring_base,
                                                       uint32_t mail_slot, uint32_t drained_addr, bool narrow, uint32_t quota, uint32_t ring_mask,
                                                       uint32_t ring_stride, uint32_t &rec, uint32_t &offset, uint32_t &bb, uint32_t &wr, uint32_t headroom = 8, bool all4 = true) {
#ifdef GBWT_HIP_CXX_LOOP
    return 2;
#else
    GBWT_WALK2U_BODY
#endif
}
#ifndef GBWT_HIP_CXX_LOOP
#undef GBWT_WALK2U_BODY
#undef GBWT_WALK2U_LOOP
#undef GBWT_WALK2U_INDEX
#undef GBWT_WALK2U_BYTES
#undef GBWT_WALK2U_RANK_A
#undef GBWT_WALK2U_RANK_B
#undef GBWT_WALK2U_LEAF
#undef GBWT_WALK2U_STAGE
#undef GBWT_WALK2U_KLOAD_WIDE
#undef GBWT_WALK2U_KLOAD_NARROW
#undef GBWT_WALK2U_ISSUE
#undef GBWT_WALK2U_FLAG_CHECK
#undef GBWT_WALK2U_LEAVES_DONE
#undef GBWT_WALK2U_MAIL_FLAG
#endif

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


// The nodes between the first node x of a chained step and its landing node L (device_index.hpp: E_CHAIN / LEAF_CHAIN)
template <class Sink>
__device__ __forceinline__ void push_chain(Sink &sink, uint32_t x, uint32_t L) {
    const uint32_t d = chain_stride(x, L);
    for (uint32_t n = x + d; n != L; n += d) sink.push(n, true);
}

// One iteration of the two-step walk in plain C++ (a generic step on a DESC2_SLOW record): for the one-time passes at open and
// for the lanes of a mixed wave that sit on a record the gather loop's packed blocks cannot count.
template <class Sink>
__device__ __forceinline__ void two_step(const DeviceIndex &ix, Sink &sink, uint32_t &rec, uint32_t &offset, uint32_t &bb) {
    const uint4 *d = ix.desc2 + 8 * static_cast<uint64_t>(rec);
    const uint4 E0 = d[0];
    if (E0.z & DESC2_SLOW) { generic_step(ix, sink, rec, offset, bb); return; }
    const uint64_t idx = bb == BLOCK_NONE ? 0u : bb + (offset >> RANK_BLOCK_SHIFT);
    const uint4 K0 = ix.cblocks[2 * idx], K1 = ix.cblocks[2 * idx + 1];
    const uint64_t bits1 = (static_cast<uint64_t>(K0.y) << 32) | K0.x, bits2 = (static_cast<uint64_t>(K0.w) << 32) | K0.z;
    const uint32_t bit = offset & 63u;
    const uint64_t below = (uint64_t(1) << bit) - 1;
    const uint32_t a = static_cast<uint32_t>(bits1 >> bit) & 1u;
    const uint64_t m = a ? bits1 : ~bits1;
    const uint32_t rank_a = a ? K1.x + __popcll(m & below) : (offset - bit) - K1.x + __popcll(m & below);
    const uint4 E = a ? d[1] : E0;
    const uint32_t j = E.y + rank_a;
    const uint32_t b = static_cast<uint32_t>(bits2 >> bit) & 1u;
    const uint32_t ones_w = (a ? K1.z : K1.y) + __popcll(m & bits2 & below);
    const uint4 leaf = d[2 + 2 * a + b];
    const uint32_t n1 = E.x, wword = E.z;
    rec = leaf.z & REC_MASK; offset = leaf.y + (b ? ones_w : j - ones_w); bb = leaf.w;
    sink.push(n1, n1 != 0);
    if (E.w & E_CHAIN) push_chain(sink, n1, (wword & REC_MASK) + ix.alphabet_offset);
    sink.push((wword & REC_MASK) + ix.alphabet_offset, (wword & LEAF_EMIT2) != 0);
    sink.push(leaf.x, leaf.x != 0);
    if (leaf.z & LEAF_CHAIN) push_chain(sink, leaf.x, rec + ix.alphabet_offset);
    sink.push(rec + ix.alphabet_offset, (leaf.z & LEAF_EMIT2) != 0);
}

// The two-step walk in plain C++ without an output, for the one-time passes at open: `sink.push` sees every node in
// order, `sink.checkpoint` sees the position of the walk after every iteration (a state from which a walker can go on).
template <class Sink>
__device__ __forceinline__ void quiet_walk(const DeviceIndex &ix, uint64_t id, Sink &sink, uint32_t *overflow) {
    uint32_t rec = 0, offset = 0, bb = BLOCK_NONE;
    if (id < ix.n_endmarker) {
        const uint2 e = ix.endmarker[id];
        if (e.x != 0) {
            sink.push(e.x, true);
            offset = e.y;
            if (!arrive(ix, e.x, e.y, rec, bb)) { rec = 0; bb = BLOCK_NONE; }
            sink.checkpoint(rec, offset, bb);
        }
    }
    uint64_t guard = 0;
    while (rec != 0) {
        // A sequence longer than 2^32 nodes is unsupported (bit 0); one that takes more steps than there are BWT positions
        // never ends -- records of a corrupt file can send a walk in circles (bit 1).  Every iteration emits a node or ends.
        if (sink.wr > 0xFFFFFFF0u) { if (overflow) atomicOr(overflow, 1u); break; }
        if (++guard > ix.max_walk) { if (overflow) atomicOr(overflow, 2u); break; }
        // ... and ALL walks of a pass together take no more steps than there are BWT positions either: overflow[1] counts them in units of
        // 4 096 iterations, so that the walks a corrupt record sends in circles share ONE budget and end within the time of a normal pass
        // (round 5: two mutations of a 3 MB file each cost an open three minutes -- every circling lane ran to the bound on its own)
        if (overflow && (guard & 4095u) == 0) {
            const uint64_t spent = static_cast<uint64_t>(atomicAdd(overflow + 1, 1u)) + 1;
            if (spent * 4096 > ix.max_walk || (*reinterpret_cast<volatile uint32_t *>(overflow) & 2u) != 0) { atomicOr(overflow, 2u); break; }
        }
        two_step(ix, sink, rec, offset, bb);
        sink.checkpoint(rec, offset, bb);
    }
}

}  // namespace

}  // namespace gbwt_hip
