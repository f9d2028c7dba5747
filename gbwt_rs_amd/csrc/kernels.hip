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

// One lane per record: the 32-byte descriptor the walk kernels read instead of starts[] (device_index.hpp).
__global__ void __launch_bounds__(256) k_build_desc(DeviceIndex ix, uint4 *desc) {
    uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec >= ix.n_records) return;
    uint64_t start, limit;
    record_bounds(ix, rec, start, limit);
    uint4 A = make_uint4(0, 0, 0, 0), B = make_uint4(0, 0, 0, 0);
    if (limit > start) {
        ByteCursor c(ix.data, start, limit);
        uint64_t sigma = 0;
        if (c.varint(sigma) && sigma != 0) {
            A.x = static_cast<uint32_t>(start); A.y = static_cast<uint32_t>(limit - start);
            B.w = static_cast<uint32_t>(start >> 32);
            if (sigma <= 2) {
                uint64_t n0 = 0, o0 = 0, d1 = 0, o1 = 0;
                bool good = c.varint(n0) && c.varint(o0);
                if (good && sigma == 2) good = c.varint(d1) && c.varint(o1);
                const uint64_t body = c.pos - start;
                if (good && body <= 0xFFFF && n0 + d1 <= 0xFFFFFFFFull && o0 <= 0xFFFFFFFFull && o1 <= 0xFFFFFFFFull) {
                    A.z = static_cast<uint32_t>(n0); A.w = static_cast<uint32_t>(o0);
                    B.x = static_cast<uint32_t>(n0 + d1); B.y = static_cast<uint32_t>(o1);
                    B.z = static_cast<uint32_t>(body) | (static_cast<uint32_t>(sigma) << 16);
                    if (sigma == 1) {
                        uint64_t value, len;
                        RunDecoder rd(1);
                        if (rd.next(c, value, len) && c.at_end() && len < 0xFFFFFFFFull) {
                            A.x = static_cast<uint32_t>(len); A.y = DESC_UNARY;
                        }
                    }
                }
            }
        }
    }
    desc[2 * rec] = A;
    desc[2 * rec + 1] = B;
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

// Appends `node` to the lane's block chain (shared by both walk kernels).  Returns false on pool overflow.
struct PathSink {
    uint32_t *wp = nullptr;      // next slot in the current block
    uint32_t left = 0;           // free slots in the current block
    uint32_t cur = POOL_NONE, head = POOL_NONE, blocks = 0;
    __device__ __forceinline__ bool push(const WalkArgs &a, uint32_t node) {
        if (left == 0) {
            uint32_t nb = atomicAdd(a.counter, 1u);
            if (nb >= a.pool_blocks) { atomicOr(a.flags, FLAG_POOL_OVERFLOW); return false; }
            a.next[nb] = POOL_NONE;
            if (cur == POOL_NONE) head = nb; else a.next[cur] = nb;
            cur = nb; blocks++;
            wp = a.pool + static_cast<uint64_t>(nb) * POOL_BLOCK_NODES;
            left = POOL_BLOCK_NODES;
        }
        if (!a.debug_nostore) *wp = node;
        wp++;
        left--;
        return true;
    }
    __device__ __forceinline__ uint64_t length() const { return static_cast<uint64_t>(blocks) * POOL_BLOCK_NODES - left; }
};

__global__ void __launch_bounds__(WAVE) k_walk(DeviceIndex ix, WalkArgs a) {
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
    PathSink sink;
    while (valid) {
        if (!sink.push(a, static_cast<uint32_t>(node))) break;
        uint64_t nn, no;
        valid = gbwt_forward(ix, node, offset, nn, no);
        node = nn; offset = no;
    }
    a.head[k] = sink.head;
    a.lengths[k] = sink.length();
}

// Wave-cooperative walk: lanes 0..P-1 of each wave own one sequence each.  Per step every owner
// looks up its record; short records are decoded by their own lane (lf_device.hpp), long ones are
// handled one distinct record at a time by the whole wave (coop_device.hpp), so sequences that sit in
// the same record -- the common case for the high-coverage records of a pangenome -- share one decode.
template <bool PROF, bool PACK16>
__global__ void __launch_bounds__(WAVE) k_walk_coop(DeviceIndex ix, WalkArgs a) {
    // PROF: per-phase cycle counters (s_memtime) of wave 0, to see where a step's latency goes
    uint64_t t_push = 0, t_bounds = 0, t_small = 0, t_coop = 0, n_steps = 0, n_groups = 0, t0 = 0, t1 = 0;
    CoopProf cprof;
#define PROF_MARK(acc) do { if (PROF) { t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; } } while (0)
    const uint32_t lane = threadIdx.x;
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(a.paths_per_wave) + lane;
    const bool owner = lane < a.paths_per_wave && k < a.n;
    uint32_t node = 0, offset = 0;
    bool active = false;
    if (owner) {
        const uint64_t id = a.seq_ids[k];
        if (id < ix.n_endmarker) {  // GBWT::start, src/gbwt.rs:213-219
            uint2 e = ix.endmarker[id];
            node = e.x; offset = e.y;
            active = node != 0;
        }
    }
    PathSink sink;
    // Touch-ahead: records of neighbouring nodes sit next to each other in the descriptor table and in the
    // byte stream, and walks move to nearby node ids, so every cooperative step also requests the cache lines
    // AHEAD bytes further on in both.  The values are only folded into `touched` (never used), but the lines
    // are then resident when the following steps need them.
    constexpr uint32_t AHEAD = 256;
    const uint64_t desc_bytes = ix.n_records * 2 * sizeof(uint4);
    uint32_t touched = 0, pf[2] = {0, 0};
    if (PROF) t0 = __builtin_amdgcn_s_memtime();
    while (__ballot(active) != 0) {
        touched ^= pf[0] ^ pf[1];
        // SequenceIter::next (src/gbwt.rs:560-567): emit pos.node, then next = forward(pos)
        if (active && !sink.push(a, node)) active = false;
        PROF_MARK(t_push);
        // GBWT::forward guards + BWT::record_bytes (src/gbwt.rs:222-229, src/bwt.rs:116-130) via the descriptor
        uint64_t start = 0;
        uint32_t bytes = 0, meta = 0, n0 = 0, o0 = 0, n1 = 0, o1 = 0, rec32 = 0;
        bool has_record = false, ok = false;
        uint32_t next_node = 0, next_offset = 0;
        if (active && node >= ix.first_node) {
            const uint64_t rec = node - ix.alphabet_offset;
            if (rec < ix.n_records) {
                const uint4 A = ix.desc[2 * rec], B = ix.desc[2 * rec + 1];
                rec32 = static_cast<uint32_t>(rec);
                if (A.y == DESC_UNARY) {             // one run, one successor: lf(i) = (z, w + i) for i < len
                    ok = offset < A.x && A.z != 0;
                    next_node = A.z; next_offset = A.w + offset;
                } else if (A.y != 0) {
                    start = (static_cast<uint64_t>(B.w) << 32) | A.x;
                    bytes = A.y; meta = B.z;
                    n0 = A.z; o0 = A.w; n1 = B.x; o1 = B.y;
                    has_record = true;
                }
            }
        }
        if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        PROF_MARK(t_bounds);
        // long records with outdegree <= 2 go through the cooperative scan, everything else is decoded by its own lane
        const bool big = has_record && bytes > a.small_record && (meta >> 16) != 0;
        bool serial = has_record && !big;
        uint64_t todo = __ballot(big);
        while (todo != 0) {
            if (PROF) n_groups++;
            const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(todo));
            const uint64_t gs = read_lane64(start, leader);
            const uint32_t gbytes = read_lane(bytes, leader), gmeta = read_lane(meta, leader);
            const bool member = big && start == gs;
            const uint32_t body_off = gmeta & 0xFFFFu;
            const uint32_t *touch_a = nullptr, *touch_b = nullptr;
            if (a.touch_ahead) {
                const uint64_t da = static_cast<uint64_t>(read_lane(rec32, leader)) * 2 * sizeof(uint4) + AHEAD;
                const uint64_t db = (gs + AHEAD) & ~uint64_t(3);
                touch_a = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(ix.desc) + (da < desc_bytes ? da : 0));
                touch_b = reinterpret_cast<const uint32_t *>(ix.data + (db < ix.data_len ? db : 0));
            }
            const int status = coop_runs_lf<PACK16, PROF>(ix.data + gs + body_off, gbytes - body_off, (gmeta >> 16) == 2, member, offset,
                                                          n0, o0, n1, o1, ok, next_node, next_offset, &cprof, touch_a, touch_b, pf);
            if (status != COOP_DONE && member) serial = true;
            todo &= ~__ballot(member);
        }
        PROF_MARK(t_coop);
        if (serial) {
            ByteCursor c(ix.data, start, start + bytes);
            uint64_t sigma, nn, no;
            ok = false;
            if (c.varint(sigma) && sigma != 0 && record_lf(c, sigma, offset, nn, no)) {
                ok = true; next_node = static_cast<uint32_t>(nn); next_offset = static_cast<uint32_t>(no);
            }
        }
        PROF_MARK(t_small);
        if (PROF) n_steps++;
        if (active) {
            active = ok;
            node = next_node; offset = next_offset;
        }
    }
    if (owner) {
        a.head[k] = sink.head;
        a.lengths[k] = sink.length();
    }
    if (touched == 0x9E3779B9u && a.flags) atomicOr(a.flags, 0u);  // keeps the touch-ahead loads alive; changes nothing
    if (PROF && a.prof && blockIdx.x == 0 && lane == 0) {
        a.prof[0] = n_steps; a.prof[1] = n_groups; a.prof[2] = t_push; a.prof[3] = t_bounds; a.prof[4] = t_small; a.prof[5] = t_coop;
        a.prof[6] = cprof.load; a.prof[7] = cprof.scan; a.prof[8] = cprof.search;
    }
#undef PROF_MARK
}

// One wave per path: follow the block chain and copy it to its CSR row.
__global__ void __launch_bounds__(256) k_compact(WalkArgs a, const uint64_t *offsets, uint32_t *nodes) {
    const uint64_t path = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (path >= a.n) return;
    uint64_t remaining = offsets[path + 1] - offsets[path];
    uint32_t *dst = nodes + offsets[path];
    uint32_t b = a.head[path];
    while (remaining > 0 && b != POOL_NONE) {
        const uint32_t cnt = remaining < POOL_BLOCK_NODES ? static_cast<uint32_t>(remaining) : POOL_BLOCK_NODES;
        const uint32_t *src = a.pool + static_cast<uint64_t>(b) * POOL_BLOCK_NODES;
        const uint32_t nb = a.next[b];
#pragma unroll
        for (uint32_t j = 0; j < POOL_BLOCK_NODES / WAVE; j++) {
            uint32_t idx = j * WAVE + lane;
            if (idx < cnt) dst[idx] = src[idx];
        }
        dst += cnt;
        remaining -= cnt;
        b = nb;
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

__global__ void __launch_bounds__(256) k_forward(DeviceIndex ix, const gbwt_hip_pos *in, uint64_t n, gbwt_hip_pos *out, uint8_t *valid) {
    uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    gbwt_hip_pos p = in[k], r{0, 0};
    uint8_t ok = gbwt_forward(ix, p.node, p.offset, r.node, r.offset) ? 1 : 0;
    if (!ok) { r.node = 0; r.offset = 0; }
    out[k] = r; valid[k] = ok;
}

// GBWT::find, src/gbwt.rs:269-281
__device__ __forceinline__ bool dev_find(const DeviceIndex &ix, uint64_t node, gbwt_hip_state &st) {
    ByteCursor c(ix.data, 0, 0);
    uint64_t sigma;
    if (!open_record(ix, node, c, sigma)) return false;
    st.node = node; st.start = 0; st.end = record_len(c, sigma);
    return true;
}

// GBWT::extend, src/gbwt.rs:292-304
__device__ __forceinline__ bool dev_extend(const DeviceIndex &ix, const gbwt_hip_state &st, uint64_t node, gbwt_hip_state &out) {
    if (node < ix.first_node) return false;
    ByteCursor c(ix.data, 0, 0);
    uint64_t sigma, rs, re, count;
    if (!open_record(ix, st.node, c, sigma)) return false;
    if (!record_follow<false>(c, sigma, st.start, st.end, node, rs, re, count)) return false;
    out.node = node; out.start = rs; out.end = re;
    return true;
}

// GBWT::extend_forward + bd_internal, src/gbwt.rs:339-347, 371-384
__device__ __forceinline__ bool dev_extend_forward(const DeviceIndex &ix, const gbwt_hip_bd_state &st, uint64_t node, gbwt_hip_bd_state &out) {
    if (node < ix.first_node) return false;
    ByteCursor c(ix.data, 0, 0);
    uint64_t sigma, rs, re, count;
    if (!open_record(ix, st.forward.node, c, sigma)) return false;
    if (!record_follow<true>(c, sigma, st.forward.start, st.forward.end, node, rs, re, count)) return false;
    out.forward.node = node; out.forward.start = rs; out.forward.end = re;
    uint64_t pos = st.reverse.start + count;
    out.reverse.node = st.reverse.node; out.reverse.start = pos; out.reverse.end = pos + (re - rs);
    return true;
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

inline unsigned grid_for(uint64_t n, unsigned block) { return static_cast<unsigned>((n + block - 1) / block); }

}  // namespace

void launch_record_stats(const DeviceIndex &ix, uint64_t *d_stats, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_record_stats, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_stats);
}

void launch_build_desc(const DeviceIndex &ix, uint4 *d_desc, hipStream_t stream) {
    if (ix.n_records == 0) return;
    hipLaunchKernelGGL(k_build_desc, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, d_desc);
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
    if (args.prof) {
        if (args.pack16) hipLaunchKernelGGL((k_walk_coop<true, true>), grid, block, 0, stream, ix, args);
        else hipLaunchKernelGGL((k_walk_coop<true, false>), grid, block, 0, stream, ix, args);
    } else {
        if (args.pack16) hipLaunchKernelGGL((k_walk_coop<false, true>), grid, block, 0, stream, ix, args);
        else hipLaunchKernelGGL((k_walk_coop<false, false>), grid, block, 0, stream, ix, args);
    }
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
void launch_search(const DeviceIndex &ix, const uint64_t *queries, uint64_t n, uint64_t len, gbwt_hip_state *out,
                   uint8_t *valid, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_search, dim3(grid_for(n, 256)), dim3(256), 0, s, ix, queries, n, len, out, valid);
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
