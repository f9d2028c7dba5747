// walk_kernels.hip -- path extraction through the pool of chained blocks: the lane-serial, one-step, two-step and
// cooperative walk kernels and the CSR compaction (hand-written HIP for gfx950, no MFMA: integer pointer-chasing over a
// byte stream).  The hot loops are in walk_loops.hpp, the extraction with known lengths in walk_direct.hip.  Launch
// wrappers are declared in kernels.hpp.
#include "kernels.hpp"

#include <hipcub/hipcub.hpp>

#include "walk_loops.hpp"

namespace gbwt_hip {

void launch_walk_direct(const DeviceIndex &ix, const WalkArgs &args, hipStream_t stream);   // walk_direct.hip

namespace {

__global__ void __launch_bounds__(WAVE) k_walk(DeviceIndex ix, WalkArgs a) {
    __shared__ uint32_t sink_lds[SINK_STAGE * WAVE];
    PathSink sink(sink_lds, threadIdx.x);
    uint64_t k = blockIdx.x * static_cast<uint64_t>(WAVE) + threadIdx.x;
    if (k >= a.n) return;
    const uint64_t id = a.seq_ids[k];
    uint64_t node = 0, offset = 0;
    bool valid = false;
    if (id < ix.n_endmarker && id < ix.n_sequences) {  // GBWT::start, src/gbwt.rs:213-219
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
        if (id < ix.n_endmarker && id < ix.n_sequences) {  // GBWT::start, src/gbwt.rs:213-219
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
        if (id < ix.n_endmarker && id < ix.n_sequences) {  // GBWT::start, src/gbwt.rs:213-219
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
        if (ix.chained) {
            // chained steps (up to sixteen nodes per iteration) are known to k_walk_direct's loops and to the plain C++ step; this kernel --
            // the pool output for indexes without sequence lengths -- takes the latter where an index has them
            if (rec != 0) two_step(ix, sink, rec, offset, bb);
            while (sink.needs_flush()) sink.flush16(a);
            continue;
        }
        const uint32_t slow_exit = walk2_hot_loop(ix.desc2, ix.cblocks, ix.alphabet_offset, ring_base, mail_slot, sink.flushed, narrow, 0xFFFFFFFFu, RING2 - 1, WAVE, rec, offset, bb, sink.wr);
        if (slow_exit) {
            const bool slow = rec != 0 && (ix.desc2[8 * static_cast<uint64_t>(rec)].z & DESC2_SLOW) != 0;
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

__global__ void __launch_bounds__(256) k_gather_lengths(const uint32_t *seq_len, uint64_t n_sequences, const uint64_t *ids, uint64_t n, uint64_t *lengths, uint32_t *max_len) {
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const uint32_t len = ids[k] < n_sequences ? seq_len[ids[k]] : 0u;   // GBWT::sequence: id >= sequences -> no iterator (src/gbwt.rs:254-256): an empty row
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
        if (id < ix.n_endmarker && id < ix.n_sequences) {
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
// splitmix64 of the position: the weight of the node at position i of its row in gbwt_hip_path_hashes (include/gbwt_hip.h)
__device__ __forceinline__ uint64_t position_weight(uint64_t i) {
    uint64_t z = i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// hashed: sum of (node + 1) * splitmix64(position) -- depends on the ORDER of the nodes, which a plain sum does not
template <bool HASHED>
__global__ void __launch_bounds__(256) k_path_sums(const uint64_t *offsets, const uint32_t *nodes, uint64_t n, uint64_t *sums) {
    const uint64_t path = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (path >= n) return;
    uint64_t acc = 0;
    const uint64_t first = offsets[path];
    for (uint64_t k = first + lane; k < offsets[path + 1]; k += WAVE) acc += HASHED ? (static_cast<uint64_t>(nodes[k]) + 1) * position_weight(k - first) : nodes[k];
    for (int d = WAVE / 2; d > 0; d >>= 1) acc += __shfl_down(acc, d, WAVE);
    if (lane == 0) sums[path] = acc;
}

}  // namespace

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
    if (args.out_nodes != nullptr) { launch_walk_direct(ix, args, stream); return; }   // lengths known: rows written in place (walk_direct.hip)
    if (args.mode == WALK_ONE_STEP) { hipLaunchKernelGGL(k_walk_blocks, grid, dim3(2 * WAVE), 0, stream, ix, args); return; }
    hipLaunchKernelGGL(k_walk_two, grid, dim3(2 * WAVE), 0, stream, ix, args);
}

void launch_gather_lengths(const uint32_t *d_seq_len, uint64_t n_sequences, const uint64_t *d_ids, uint64_t n, uint64_t *d_lengths, uint32_t *d_max_len, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_gather_lengths, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_seq_len, n_sequences, d_ids, n, d_lengths, d_max_len);
}

void launch_compact(const WalkArgs &args, const uint64_t *d_offsets, uint32_t *d_nodes, hipStream_t stream) {
    if (args.n == 0) return;
    hipLaunchKernelGGL(k_compact, dim3(grid_for(args.n, 256 / WAVE)), dim3(256), 0, stream, args, d_offsets, d_nodes);
}

void launch_path_sums(const uint64_t *d_offsets, const uint32_t *d_nodes, uint64_t n, uint64_t *d_sums, bool hashed, hipStream_t stream) {
    if (n == 0) return;
    if (hashed) hipLaunchKernelGGL(k_path_sums<true>, dim3(grid_for(n, 256 / WAVE)), dim3(256), 0, stream, d_offsets, d_nodes, n, d_sums);
    else hipLaunchKernelGGL(k_path_sums<false>, dim3(grid_for(n, 256 / WAVE)), dim3(256), 0, stream, d_offsets, d_nodes, n, d_sums);
}

}  // namespace gbwt_hip
