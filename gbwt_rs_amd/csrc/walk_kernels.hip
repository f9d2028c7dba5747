// walk_kernels.hip -- path extraction: the walk kernels (two-step / one-step hot loops in gfx950 assembly, cooperative, lane-serial) and the CSR compaction
// (hand-written HIP for gfx950, no MFMA: integer pointer-chasing over a byte stream).  Launch wrappers are declared in kernels.hpp.
#include "kernels.hpp"

#include <hipcub/hipcub.hpp>

#include "coop_device.hpp"
#include "device_common.hpp"
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
                                                   uint32_t mail_slot, uint32_t flushed, bool narrow, uint32_t quota, uint32_t ring_mask, uint32_t ring_stride,
                                                   uint32_t &rec, uint32_t &offset, uint32_t &bb, uint32_t &wr) {
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
        : "memory", "vcc", "scc", "s44", "s45", "s46", "s47", \
          "v40", "v41", "v42", "v43", "v44", "v48", "v49", "v50", "v51", "v52", "v53", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", \
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

// ---- two-step walk, wave-uniform variant ----------------------------------------------------------------------
// With sixty-four lanes walking, the vector-memory path is what the walk saturates (TA / TD busy 92-95 % of the kernel,
// profiles/r01_coop_pmc_headline.txt): every lane fetches its own copy of the 104 descriptor bytes, 16 bytes per lane
// per load instruction.  The walkers of a wave hold the same segment of neighbouring rows, and haplotypes travel
// together, so very often ALL lanes sit on the same record: then the descriptor is one scalar fetch (s_load, 28 SGPRs)
// and only the two-step rank block -- the one thing that differs between lanes -- goes through the vector path
// (28 bytes per lane instead of 132).  gfx9 VALU instructions read at most one SGPR, so instead of v_cndmask the
// per-lane choices are made by running the same `v_mov / v_add  vgpr, sgpr` under the exec mask of each choice.
// Same registers and conventions as walk2_hot_loop.  Leaves with reason 2 -- nothing in flight, state intact -- as
// soon as the lanes are not all on one record (or some are parked); the caller then continues with walk2_hot_loop.
// SGPRs: s[48:63] F0 F1 L00 L01, s[64:71] L10 L11, s[72:75] look-ahead target, s[76:85] masks, s[88:89] descriptor
// address, s78 the record.
__device__ __forceinline__ uint32_t walk2_uniform_loop(const uint4 *desc2, const uint4 *cblocks, uint32_t alphabet_offset, uint32_t ring_base,
                                                       uint32_t mail_slot, uint32_t drained_addr, bool narrow, uint32_t quota, uint32_t ring_mask,
                                                       uint32_t ring_stride, uint32_t &rec, uint32_t &offset, uint32_t &bb, uint32_t &wr) {
#ifdef GBWT_HIP_CXX_LOOP
    return 2;
#else
    uint32_t reason;
    const uint32_t slack = ring_mask + 1 - 8;   // leave with more than slots - 8 nodes waiting in a ring
    const uint32_t dlo = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(desc2)), dhi = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(desc2) >> 32);
#define GBWT_WALK2U_ISSUE(KLOAD, REFRESH)                                                                             \
    "v_readfirstlane_b32 s78, v40\n\t"                    /* the record of lane 0 */                       \
    "v_cmp_ne_u32_e32 vcc, -1, v43\n\t"                   /* bb != BLOCK_NONE */                          \
    "v_lshrrev_b32_e32 v70, 6, v42\n\t"                                                                   \
    "v_add_u32_e32 v70, v70, v43\n\t"                                                                     \
    "s_lshl_b32 s76, s78, 7\n\t"                          /* two-step descriptors are 128 bytes */        \
    "s_lshr_b32 s77, s78, 25\n\t"                                                                         \
    "v_cndmask_b32_e32 v70, 0, v70, vcc\n\t"              /* block bb + offset / 64, or the zero block */ \
    "s_add_u32 s76, s76, %[dlo]\n\t"                                                                      \
    "s_addc_u32 s77, s77, %[dhi]\n\t"                                                                     \
    "v_lshlrev_b32_e32 v90, 5, v70\n\t"                   /* two-step blocks are 32 bytes */              \
    "v_cmp_ne_u32_e32 vcc, s78, v40\n\t"                  /* lanes on another record */                   \
    "s_load_dwordx16 s[48:63], s[76:77], 0x0\n\t"         /* F0, F1, leaf (0, 0), leaf (0, 1) */          \
    "s_load_dwordx8 s[64:71], s[76:77], 0x40\n\t"         /* leaf (1, 0), leaf (1, 1) */                  \
    "s_load_dwordx4 s[72:75], s[76:77], 0x60\n\t"         /* look-ahead target */                         \
    KLOAD                                                                                                 \
    REFRESH
// K0 = {bits1, bits2}, K1 = {ones1, R0, R1} of the lane's two-step block: SGPR base + 32-bit byte offset while the block
// array is below 4 GiB, a 64-bit address per lane above
#define GBWT_WALK2U_KLOAD_NARROW                                                                          \
    "global_load_dwordx4 v[80:83], v90, %[cblocks]\n\t"                                                   \
    "global_load_dwordx3 v[84:86], v90, %[cblocks] offset:16\n\t"
#define GBWT_WALK2U_KLOAD_WIDE                                                                            \
    "v_lshrrev_b32_e32 v91, 27, v70\n\t"                                                                  \
    "v_lshl_add_u64 v[90:91], v[90:91], 0, %[cblocks]\n\t"                                                \
    "global_load_dwordx4 v[80:83], v[90:91], off\n\t"                                                     \
    "global_load_dwordx3 v[84:86], v[90:91], off offset:16\n\t"
#define GBWT_WALK2U_LEAF(MASK, X, Y, Z, W)                                                                 \
    MASK "\n\t"                                                                                           \
    "v_mov_b32_e32 v112, " X "\n\t"                       /* node to emit */                              \
    "v_add_u32_e32 v42, " Y ", v107\n\t"                  /* the new offset = offset base + rank_b */     \
    "v_mov_b32_e32 v114, " Z "\n\t"                       /* landing record | flags */                    \
    "v_mov_b32_e32 v43, " W "\n\t"                        /* its block base */
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
        "s_bitcmp1_b32 s52, 30\n\t"                         /* DESC2_SLOW */ \
        "s_cbranch_scc1 .Lgbwt_walk2u_slow_%=\n\t" \
        "v_lshrrev_b64 v[94:95], v42, v[80:81]\n\t"         /* bits1 >> bit */ \
        "v_lshlrev_b64 v[96:97], v42, -1\n\t"               /* bits at and above `bit` */ \
        "v_and_b32_e32 v94, 1, v94\n\t"                     /* a */ \
        "v_and_b32_e32 v99, 0xffffffc0, v42\n\t"            /* offset - bit */ \
        "v_add_u32_e32 v98, -1, v94\n\t"                    /* a ? 0 : ~0 */ \
        "v_cmp_eq_u32_e32 vcc, 1, v94\n\t"                  /* vcc = a */ \
        "v_xor_b32_e32 v100, v80, v98\n\t"                  /* m = a ? bits1 : ~bits1 */ \
        "v_xor_b32_e32 v101, v81, v98\n\t" \
        "v_bfi_b32 v100, v96, 0, v100\n\t"                  /* m below `bit` */ \
        "v_bfi_b32 v101, v97, 0, v101\n\t" \
        "v_sub_u32_e32 v99, v99, v84\n\t"                   /* (offset - bit) - ones1 */ \
        "v_bcnt_u32_b32 v102, v100, 0\n\t" \
        "v_cndmask_b32_e32 v99, v99, v84, vcc\n\t"          /* a ? ones1 : that */ \
        "v_bcnt_u32_b32 v102, v101, v102\n\t"               /* p */ \
        "v_cndmask_b32_e32 v106, v85, v86, vcc\n\t"         /* R_a */ \
        "s_mov_b64 s[44:45], vcc\n\t"                       /* a */ \
        "v_add_u32_e32 v99, v99, v102\n\t"                  /* rank_a */ \
        "v_lshrrev_b64 v[108:109], v42, v[82:83]\n\t"       /* bits2 >> bit */ \
        "v_mov_b32_e32 v104, s48\n\t"                       /* edge 0: node */ \
        "v_mov_b32_e32 v105, s52\n\t"                       /*         w_0 | flags */ \
        "v_add_u32_e32 v103, s49, v99\n\t"                  /*         j = offset base + rank_a: offset in w_a */ \
        "s_mov_b64 exec, s[44:45]\n\t"                      /* the lanes that take edge 1 */ \
        "v_mov_b32_e32 v104, s50\n\t" \
        "v_mov_b32_e32 v105, s53\n\t" \
        "v_add_u32_e32 v103, s51, v99\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        "v_and_b32_e32 v100, v100, v82\n\t"                 /* a-paths below `bit` with value 1 in w_a */ \
        "v_and_b32_e32 v101, v101, v83\n\t" \
        "v_and_b32_e32 v108, 1, v108\n\t"                   /* b */ \
        "v_bcnt_u32_b32 v106, v100, v106\n\t" \
        "v_cmp_eq_u32_e64 s[46:47], 1, v108\n\t"            /* s[46:47] = b */ \
        "v_bcnt_u32_b32 v106, v101, v106\n\t"               /* ones of w_a before j */ \
        "v_sub_u32_e32 v107, v103, v106\n\t"                /* j - ones */ \
        "v_and_b32_e32 v110, 0x3fffffff, v105\n\t"          /* w_a */ \
        "v_cndmask_b32_e64 v107, v107, v106, s[46:47]\n\t"  /* rank_b */ \
        GBWT_WALK2U_LEAF("s_nor_b64 exec, s[44:45], s[46:47]", "s56", "s57", "s58", "s59")     /* lanes of leaf (0, 0) */ \
        GBWT_WALK2U_LEAF("s_andn2_b64 exec, s[46:47], s[44:45]", "s60", "s61", "s62", "s63")   /*          leaf (0, 1) */ \
        GBWT_WALK2U_LEAF("s_andn2_b64 exec, s[44:45], s[46:47]", "s64", "s65", "s66", "s67")   /*          leaf (1, 0) */ \
        GBWT_WALK2U_LEAF("s_and_b64 exec, s[44:45], s[46:47]", "s68", "s69", "s70", "s71")     /*          leaf (1, 1) */ \
        "s_mov_b64 exec, -1\n\t" \
        "v_and_b32_e32 v40, 0x3fffffff, v114\n\t"           /* the new record */ \
        "v_and_b32_e32 v92, %[ringmask], v44\n\t"           /* ring slot of the next node */ \
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
        "v_mov_b32_e32 v76, s72\n\t"                        /* mailbox: look-ahead target of the record just left ... */ \
        "v_mov_b32_e32 v77, s73\n\t" \
        "v_cmp_lt_u32_e32 vcc, v44, %[quota]\n\t"           /* a walker that has emitted its share parks */ \
        "v_mov_b32_e32 v78, s74\n\t" \
        "v_mov_b32_e32 v79, v44\n\t"                        /* ... + nodes staged so far */ \
        "ds_write_b128 %[mail], v[76:79]\n\t" \
        "v_cndmask_b32_e32 v40, 0, v40, vcc\n\t" \
        "v_cndmask_b32_e32 v43, -1, v43, vcc\n\t" \
        "v_sub_u32_e32 v92, v44, v45\n\t"                   /* nodes waiting in the ring (drained as of the last iteration) */ \
        GBWT_WALK2U_ISSUE(KLOAD, REFRESH) \
        "v_cmp_lt_u32_e64 s[46:47], %[slack], v92\n\t"      /* more than slots - 8 of them */ \
        "s_nop 0\n\t" \
        "s_cbranch_vccnz .Lgbwt_walk2u_mixed_%=\n\t" \
        "s_cmp_eq_u32 s78, 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_walk2u_out_%=\n\t"           /* everyone has parked */ \
        "s_cmp_eq_u64 s[46:47], 0\n\t" \
        "s_cbranch_scc1 .Lgbwt_walk2u_loop_%=\n\t" \
        "s_branch .Lgbwt_walk2u_out_%=\n\t" \
        ".Lgbwt_walk2u_slow_%=:\n\t" \
        "s_mov_b32 %[reason], 1\n\t" \
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
          [ringmask] "s"(ring_mask), [stride] "s"(4 * ring_stride), "{s41}"(alphabet_offset) \
        : "memory", "vcc", "scc", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", \
          "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", \
          "v40", "v42", "v43", "v44", "v45", "v70", "v91", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v90", "v92", "v94", "v95", \
          "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v114");
    // the loop re-reads how far the helper has emptied the ring every iteration (working with the count it was entered
    // with, it had to leave after (slots - 8 - waiting) / 4 iterations: 180 exits per 1 024 iterations, each with a wasted
    // round of loads)
    if (narrow) { GBWT_WALK2U_LOOP(GBWT_WALK2U_KLOAD_NARROW, "ds_read_b32 v45, %[drained]\n\t") }
    else { GBWT_WALK2U_LOOP(GBWT_WALK2U_KLOAD_WIDE, "ds_read_b32 v45, %[drained]\n\t") }
#undef GBWT_WALK2U_LOOP
#undef GBWT_WALK2U_LEAF
#undef GBWT_WALK2U_KLOAD_WIDE
#undef GBWT_WALK2U_KLOAD_NARROW
#undef GBWT_WALK2U_ISSUE
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
    // SGPR base + 32-bit byte offsets while both arrays are below 4 GiB, 64-bit addresses otherwise
    const bool narrow = !a.wide_addresses && ix.n_records * 128 <= 0xFFFFFFFFull && ix.n_blocks * 32 <= 0xFFFFFFFFull;
    while (__ballot(rec != 0) != 0) {
        const uint32_t slow_exit = walk2_hot_loop(ix.desc2, ix.cblocks, ix.alphabet_offset, ring_base, mail_slot, sink.flushed, narrow, 0xFFFFFFFFu, RING2 - 1, WAVE, rec, offset, bb, sink.wr);
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

__device__ __forceinline__ RowTarget row_target(const WalkArgs &a, uint64_t w) {
    RowTarget t;
    const uint64_t k = w < a.n ? w : w - a.n;
    t.backward = w >= a.n;
    t.len = a.out_offsets[k + 1] - a.out_offsets[k];
    t.row = a.out_nodes + a.out_offsets[k];
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
    const uint64_t base = ix.sample_base[id], count = ix.sample_base[id + 1] - base;
    const uint64_t len = a.out_offsets[k + 1] - a.out_offsets[k];
    if (j >= count) return s;                                 // this row has fewer segments: nothing to do
    const uint4 here = ix.samples[base + j];
    const uint64_t from = j == 0 ? 0 : here.w;                // segment 0 starts with the start node (sample 0 is the state after it)
    const uint64_t to = j + 1 < count ? ix.samples[base + j + 1].w : len;
    t.row = a.out_nodes + a.out_offsets[k] + from;
    if (a.debug & 2u) t.row = a.out_nodes + (w % 4096u) * 4096u;   // measurement switch: all rows land in one 64 MB window (wrong output)
    if (a.debug & 128u) t.row = a.out_nodes + (w % 64u) * 4096u;   //                     ... in 1 MB (stays in every L2)
    t.len = to > from ? to - from : 0;
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

__global__ void __launch_bounds__(2 * WAVE) k_walk_direct(DeviceIndex ix, WalkArgs a) {
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
    const uint64_t w = group * a.paths_per_wave + lane;
    const bool owner = lane < a.paths_per_wave && w < walkers;
    const uint32_t ring_mask = a.ring_slots - 1;
    RowTarget target;
    WalkerStart begin;
    if (owner) {
        if (a.segments) begin = segment_start(ix, a, w, target);
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
        RowWriter writer{lds_ptr(ring_lds + lane), target, 0, ring_mask, (a.debug & 1u) != 0};
        const lds_u32_t *const served_mail = lds_ptr(&mailbox[serve]);
        const uint32_t piece = a.segments ? a.row_piece : 0u;       // rows filled back to front stay with the lane-per-row writer
        const auto lds_address = [](const void *q) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(q)); };
        const CoopRows rows{lds_address(ring_lds), lds_address(mailbox), lds_address(row_state), ring_mask, (a.debug & 1u) != 0, (a.debug & 4u) != 0,
                            (a.debug & 64u) != 0};
        uint32_t mine = lane;                                        // the row this lane watches
        uint32_t served[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // the row this lane serves in group g
        if (piece) {
            const uint64_t at = reinterpret_cast<uintptr_t>(target.row);
            lds_poke(lds_ptr(&row_state[lane].x), static_cast<uint32_t>(at));
            lds_poke(lds_ptr(&row_state[lane].y), static_cast<uint32_t>(at >> 32));
            lds_poke(lds_ptr(&row_state[lane].z), static_cast<uint32_t>(std::min<uint64_t>(target.len, 0xFFFFFFF0u)));
            // order = the rows sorted by (address phase within a piece, row): rank by counting, once per workgroup
            const uint32_t phase = (static_cast<uint32_t>(at) >> 2) & (piece - 1);
            uint32_t rank = 0;
            for (uint32_t other = 0; other < WAVE; other++) {
                const uint32_t theirs = (lds_peek(lds_ptr(&row_state[other].x)) >> 2) & (piece - 1);
                rank += (theirs < phase || (theirs == phase && other < lane)) ? 1u : 0u;
            }
            if (a.debug & 32u) rank = lane;                          // measurement switch: groups of consecutive rows
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
            if (lane < a.helper_lanes && look_rec != 0 && stamp != seen) {
                const uint4 *d = ix.desc2 + 8 * static_cast<uint64_t>(look_rec);
                touch_line(d, dummy);
                touch_line(d + 4, dummy);
                touch_line(ix.cblocks + 2 * (static_cast<uint64_t>(look_base) + __umulhi(spread, look_count)), dummy);
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
        if (quota > 0 && id < ix.n_endmarker) {  // GBWT::start, src/gbwt.rs:213-219
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
    while (__ballot(rec != 0) != 0) {
        const uint32_t drained = lds_peek(my_drained);
        if (__ballot(sink.wr - drained > ring_mask + 1 - 8) != 0) { __builtin_amdgcn_s_sleep(2); continue; }   // ring full: let the helper catch up
        // all lanes on one record: scalar descriptor fetch; otherwise every lane fetches its own
        uint32_t slow_exit = a.uniform_loop ? walk2_uniform_loop(ix.desc2, ix.cblocks, ix.alphabet_offset, ring_base, mail_slot, static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&row_state[lane].w)), narrow, quota, ring_mask, RING_PITCH, rec, offset, bb, sink.wr) : 2u;
        if (slow_exit == 2) slow_exit = walk2_hot_loop(ix.desc2, ix.cblocks, ix.alphabet_offset, ring_base, mail_slot, drained, narrow, quota, ring_mask, RING_PITCH, rec, offset, bb, sink.wr);
        if (slow_exit) {
            bool generic = rec != 0 && (ix.desc2[8 * static_cast<uint64_t>(rec) + 1].x & DESC2_SLOW) != 0;   // lanes on a slow record
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
                while (__ballot(in_table) != 0) {
                    if (in_table) {
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
                    if (__ballot(sink.wr - lds_peek(my_drained) > ring_mask + 1 - 8) != 0) break;   // ring full: the outer loop waits for the helper
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_poke(done_flag, 1);
}

// Arithmetic modulo the Mersenne prime 2^61 - 1 for the order-sensitive fingerprints below.
constexpr uint64_t FP_P = (uint64_t(1) << 61) - 1;
constexpr uint64_t FP_X = 0x1D4F5C6B7A891234ull % FP_P;           // base of the polynomial
__host__ __device__ inline uint64_t fp_mul(uint64_t a, uint64_t b) {
#ifdef __HIP_DEVICE_COMPILE__
    const uint64_t hi = __umul64hi(a, b), lo = a * b;
#else
    const unsigned __int128 t = static_cast<unsigned __int128>(a) * b;
    const uint64_t hi = static_cast<uint64_t>(t >> 64), lo = static_cast<uint64_t>(t);
#endif
    uint64_t r = (lo & FP_P) + ((lo >> 61) | (hi << 3));          // a, b < 2^61: hi < 2^58
    r = (r & FP_P) + (r >> 61);
    return r >= FP_P ? r - FP_P : r;
}
__host__ __device__ inline uint64_t fp_add(uint64_t a, uint64_t b) { const uint64_t r = a + b; return r >= FP_P ? r - FP_P : r; }
__host__ __device__ inline uint64_t fp_pow(uint64_t base, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = fp_mul(r, base); base = fp_mul(base, base); e >>= 1; }
    return r;
}
__host__ __device__ inline uint64_t fp_hash(uint64_t v) {           // splitmix64 finaliser, reduced
    v += 0x9E3779B97F4A7C15ull; v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull; v = (v ^ (v >> 27)) * 0x94D049BB133111EBull; v ^= v >> 31;
    return v % FP_P;
}

// Sink that counts the nodes of a sequence (its length) and keeps two fingerprints of it:
//   fwd = sum h(v_i) x^i          (the sequence as it is)
//   rev = sum h(v_i ^ 1) x^-i     (times x^(len-1): the fingerprint `fwd` of the sequence reversed and flipped)
// k_check_orientation_pairs uses them to prove that sequence 2k+1 is sequence 2k reversed before an extraction is
// allowed to fill a row from both ends.
struct CountSink {
    uint32_t wr = 0;
    uint64_t fwd = 0, rev = 0, xp = 1, xm = 1, xinv;
    __device__ __forceinline__ explicit CountSink(uint64_t x_inverse) : xinv(x_inverse) {}
    __device__ __forceinline__ void push(uint32_t node, bool counts) {
        if (!counts) return;
        fwd = fp_add(fwd, fp_mul(fp_hash(node), xp));
        rev = fp_add(rev, fp_mul(fp_hash(node ^ 1u), xm));
        xp = fp_mul(xp, FP_X); xm = fp_mul(xm, xinv);
        wr++;
    }
    __device__ __forceinline__ void checkpoint(uint32_t, uint32_t, uint32_t) {}
};

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
        if (++guard > 0xFFFFFFF0ull || sink.wr > 0xFFFFFFF0u) { if (overflow) atomicOr(overflow, 1u); break; }
        const uint4 *d = ix.desc2 + 8 * static_cast<uint64_t>(rec);
        const uint4 F1 = d[1];
        if (F1.x & DESC2_SLOW) { generic_step(ix, sink, rec, offset, bb); sink.checkpoint(rec, offset, bb); continue; }
        const uint4 F0 = d[0];
        const uint64_t idx = bb == BLOCK_NONE ? 0u : bb + (offset >> RANK_BLOCK_SHIFT);
        const uint4 K0 = ix.cblocks[2 * idx], K1 = ix.cblocks[2 * idx + 1];
        const uint64_t bits1 = (static_cast<uint64_t>(K0.y) << 32) | K0.x, bits2 = (static_cast<uint64_t>(K0.w) << 32) | K0.z;
        const uint32_t bit = offset & 63u;
        const uint64_t below = (uint64_t(1) << bit) - 1;
        const uint32_t a = static_cast<uint32_t>(bits1 >> bit) & 1u;
        const uint64_t m = a ? bits1 : ~bits1;
        const uint32_t rank_a = a ? K1.x + __popcll(m & below) : (offset - bit) - K1.x + __popcll(m & below);
        const uint32_t j = (a ? F0.w : F0.y) + rank_a;
        const uint32_t b = static_cast<uint32_t>(bits2 >> bit) & 1u;
        const uint32_t ones_w = (a ? K1.z : K1.y) + __popcll(m & bits2 & below);
        const uint4 leaf = d[2 + 2 * a + b];
        const uint32_t n1 = a ? F0.z : F0.x, wword = a ? F1.y : F1.x;
        rec = leaf.z & REC_MASK; offset = leaf.y + (b ? ones_w : j - ones_w); bb = leaf.w;
        sink.push(n1, n1 != 0);
        sink.push((wword & REC_MASK) + ix.alphabet_offset, (wword & LEAF_EMIT2) != 0);
        sink.push(leaf.x, leaf.x != 0);
        sink.push(rec + ix.alphabet_offset, (leaf.z & LEAF_EMIT2) != 0);
        sink.checkpoint(rec, offset, bb);
    }
}

// One lane per sequence: the number of nodes SequenceIter would yield (src/gbwt.rs:557-568) and, with `prints`, the two
// fingerprints (two modular multiplications per node: more than half of the pass, so they are only computed when an
// extraction could fill rows from both ends, i.e. when the index gets no sequence samples).
struct LengthSink {
    uint32_t wr = 0;
    __device__ __forceinline__ void push(uint32_t, bool counts) { wr += counts ? 1u : 0u; }
    __device__ __forceinline__ void checkpoint(uint32_t, uint32_t, uint32_t) {}
};

__global__ void __launch_bounds__(256) k_sequence_lengths(DeviceIndex ix, uint32_t *seq_len, uint64_t *prints, uint64_t x_inverse, uint32_t *overflow) {
    const uint64_t id = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (id >= ix.n_sequences) return;
    if (prints == nullptr) {
        LengthSink sink;
        quiet_walk(ix, id, sink, overflow);
        seq_len[id] = sink.wr;
        return;
    }
    CountSink sink(x_inverse);
    quiet_walk(ix, id, sink, overflow);
    seq_len[id] = sink.wr;
    prints[2 * id] = sink.fwd; prints[2 * id + 1] = sink.rev;
}

// Sequence samples: sample 0 = the position after the start node, sample j = the first position at which at least
// j * interval nodes have been emitted (an iteration emits at most four, so no boundary is skipped).
__global__ void __launch_bounds__(256) k_sample_counts(const uint32_t *seq_len, uint64_t n_sequences, uint32_t interval, uint64_t *counts) {
    const uint64_t id = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (id < n_sequences) counts[id] = seq_len[id] == 0 ? 0 : (seq_len[id] - 1) / interval + 1;
}

struct SampleSink {
    uint32_t wr = 0, next = 0, interval;
    uint4 *out;
    uint64_t written = 0, capacity;
    __device__ __forceinline__ void push(uint32_t, bool counts) { wr += counts ? 1u : 0u; }
    __device__ __forceinline__ void checkpoint(uint32_t rec, uint32_t offset, uint32_t bb) {
        if (wr >= next && written < capacity) { out[written++] = make_uint4(rec, offset, bb, wr); next += interval; }
    }
};

__global__ void __launch_bounds__(256) k_record_samples(DeviceIndex ix, const uint64_t *sample_base, uint32_t interval, uint4 *samples, uint32_t *overflow) {
    const uint64_t id = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (id >= ix.n_sequences) return;
    SampleSink sink;
    sink.interval = interval;
    sink.out = samples + sample_base[id];
    sink.capacity = sample_base[id + 1] - sample_base[id];
    if (sink.capacity == 0) return;
    quiet_walk(ix, id, sink, overflow);
    // a sample that was never reached cannot exist (every boundary lies below the length); keep the table well-formed anyway
    for (; sink.written < sink.capacity; sink.written++) sink.out[sink.written] = make_uint4(0u, 0u, BLOCK_NONE, sink.wr);
}

// One lane per path of a bidirectional index: sequence 2k + 1 must be sequence 2k reversed with every node flipped
// (support::reverse_path, src/support.rs:310-314) -- same length, and the fingerprint of each as it is equals the
// fingerprint of the other one reversed and flipped.  Any failure clears the flag: rows are then filled from one end.
__global__ void __launch_bounds__(256) k_check_orientation_pairs(const uint32_t *seq_len, const uint64_t *prints, uint64_t n_pairs, uint32_t *mismatch) {
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n_pairs) return;
    const uint32_t lf = seq_len[2 * k], lr = seq_len[2 * k + 1];
    bool good = lf == lr;
    if (good && lf > 0) {
        const uint64_t shift = fp_pow(FP_X, lf - 1);
        good = prints[4 * k] == fp_mul(prints[4 * k + 3], shift) && prints[4 * k + 2] == fp_mul(prints[4 * k + 1], shift);
    }
    if (!good) atomicOr(mismatch, 1u);
}

__global__ void __launch_bounds__(256) k_gather_lengths(const uint32_t *seq_len, const uint64_t *ids, uint64_t n, uint64_t *lengths, uint32_t *max_len) {
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const uint32_t len = seq_len[ids[k]];
    lengths[k] = len;
    atomicMax(max_len, len);
    atomicMax(max_len + 1, ~len);      // max_len[1] = ~(the shortest)
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

}  // namespace

// keys[k] = number of segments of row k = samples of its sequence (0 for an empty sequence), rows[k] = k
__global__ void __launch_bounds__(256) k_segment_counts(const uint64_t *sample_base, const uint64_t *ids, uint64_t n, uint32_t *keys, uint32_t *rows) {
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const uint64_t id = ids[k];
    keys[k] = static_cast<uint32_t>(sample_base[id + 1] - sample_base[id]);
    rows[k] = static_cast<uint32_t>(k);
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
    hipLaunchKernelGGL(k_segment_counts, dim3(grid_for(n, 256)), dim3(256), 0, stream, ix.sample_base, d_ids, n, d_keys, d_rows);
    hipcub::DoubleBuffer<uint32_t> keys(d_keys, d_keys + n), rows(d_rows, d_rows + n);
    (void)hipcub::DeviceRadixSort::SortPairsDescending(d_temp, temp_bytes, keys, rows, static_cast<int>(n), 0, 32, stream);   // radix sort: stable
    hipLaunchKernelGGL(k_level_counts, dim3(grid_for(segments, 256)), dim3(256), 0, stream, keys.Current(), n, segments, d_level_counts);
    *d_sorted_rows = rows.Current();
    // d_level[0] = 0, d_level[j + 1] = counts[0] + ... + counts[j]: the caller runs launch_scan on d_level_counts
    (void)d_level;
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
    if (args.out_nodes != nullptr) {   // lengths known: rows written in place, both ends at once
        const uint64_t walkers = args.segments ? args.walkers : (args.both_ends ? 2 * args.n : args.n);
        unsigned groups = grid_for(walkers, p);
        if (args.xcd_map) groups = (groups + 7u) / 8u * 8u;   // whole eighths; the workgroups past the end own nothing
        hipLaunchKernelGGL(k_walk_direct, dim3(groups), dim3(2 * WAVE), args.ring_slots * RING_PITCH * sizeof(uint32_t), stream, ix, args);
        return;
    }
    if (args.mode == WALK_ONE_STEP) { hipLaunchKernelGGL(k_walk_blocks, grid, dim3(2 * WAVE), 0, stream, ix, args); return; }
    hipLaunchKernelGGL(k_walk_two, grid, dim3(2 * WAVE), 0, stream, ix, args);
}

void launch_sequence_lengths(const DeviceIndex &ix, uint32_t *d_seq_len, uint64_t *d_prints, uint32_t *d_overflow, hipStream_t stream) {
    if (ix.n_sequences == 0) return;
    const uint64_t x_inverse = fp_pow(FP_X, FP_P - 2);   // Fermat
    hipLaunchKernelGGL(k_sequence_lengths, dim3(grid_for(ix.n_sequences, 256)), dim3(256), 0, stream, ix, d_seq_len, d_prints, x_inverse, d_overflow);
}

void launch_check_orientation_pairs(const uint32_t *d_seq_len, const uint64_t *d_prints, uint64_t n_pairs, uint32_t *d_mismatch, hipStream_t stream) {
    if (n_pairs) hipLaunchKernelGGL(k_check_orientation_pairs, dim3(grid_for(n_pairs, 256)), dim3(256), 0, stream, d_seq_len, d_prints, n_pairs, d_mismatch);
}

void launch_gather_lengths(const uint32_t *d_seq_len, const uint64_t *d_ids, uint64_t n, uint64_t *d_lengths, uint32_t *d_max_len, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_gather_lengths, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_seq_len, d_ids, n, d_lengths, d_max_len);
}

void launch_sample_counts(const uint32_t *d_seq_len, uint64_t n_sequences, uint32_t interval, uint64_t *d_counts, hipStream_t stream) {
    if (n_sequences) hipLaunchKernelGGL(k_sample_counts, dim3(grid_for(n_sequences, 256)), dim3(256), 0, stream, d_seq_len, n_sequences, interval, d_counts);
}

void launch_record_samples(const DeviceIndex &ix, const uint64_t *d_sample_base, uint32_t interval, uint4 *d_samples, hipStream_t stream) {
    if (ix.n_sequences == 0) return;
    hipLaunchKernelGGL(k_record_samples, dim3(grid_for(ix.n_sequences, 256)), dim3(256), 0, stream, ix, d_sample_base, interval, d_samples,
                       static_cast<uint32_t *>(nullptr));
}

void launch_compact(const WalkArgs &args, const uint64_t *d_offsets, uint32_t *d_nodes, hipStream_t stream) {
    if (args.n == 0) return;
    hipLaunchKernelGGL(k_compact, dim3(grid_for(args.n, 256 / WAVE)), dim3(256), 0, stream, args, d_offsets, d_nodes);
}

void launch_path_sums(const uint64_t *d_offsets, const uint32_t *d_nodes, uint64_t n, uint64_t *d_sums, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_path_sums, dim3(grid_for(n, 256 / WAVE)), dim3(256), 0, stream, d_offsets, d_nodes, n, d_sums);
}

}  // namespace gbwt_hip
