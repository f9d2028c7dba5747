// query_kernels.hip -- navigation and search: start / forward / backward, find / extend / bidirectional search, follow
// (hand-written HIP for gfx950, no MFMA: integer pointer-chasing over a byte stream).  Launch wrappers are declared in kernels.hpp.
#include "kernels.hpp"

#include <hipcub/hipcub.hpp>

#include "device_common.hpp"
#include "lf_device.hpp"

namespace gbwt_hip {

namespace {

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
                    const uint4 K = ix.blocks[d.C.z + (i >> RANK_BLOCK_SHIFT)];   // (class 2: C.z = the record's first rank block, k_finish_block_base)
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
    const uint4 K = ix.blocks[d.C.z + (p >> RANK_BLOCK_SHIFT)];
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
    // two rank-block lookups -- or fewer: nothing lies before position 0 (the first extension of every find), and a range that ends in the
    // block it starts in (most ranges after a few extensions) reads that block once: one 16-byte load instead of two per lane
    uint32_t z0, z1;
    if (two && ps != 0 && pe < len && (ps >> RANK_BLOCK_SHIFT) == (pe >> RANK_BLOCK_SHIFT)) {
        const uint4 K = ix.blocks[d.C.z + (ps >> RANK_BLOCK_SHIFT)];
        const uint64_t bits = (static_cast<uint64_t>(K.y) << 32) | K.x;
        z0 = ps - (K.z + __popcll(bits & ((uint64_t(1) << (ps & 63u)) - 1)));
        z1 = pe - (K.z + __popcll(bits & ((uint64_t(1) << (pe & 63u)) - 1)));
    } else {
        z0 = ps == 0 ? 0u : count0_before(ix, d, rec, ps);
        z1 = count0_before(ix, d, rec, pe);
    }
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
    const uint4 *blocks = ix.blocks + d.C.z;
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

}  // namespace

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

}  // namespace gbwt_hip
