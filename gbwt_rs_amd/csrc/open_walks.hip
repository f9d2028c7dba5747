// open_walks.hip -- the one-time walks at open (one lane per sequence, the two-step walk in plain C++): sequence lengths,
// the fingerprints that prove sequence 2k + 1 is sequence 2k reversed, and the sequence samples of the segmented
// extraction.  Launch wrappers declared in kernels.hpp.
#include "kernels.hpp"

#include "walk_loops.hpp"

namespace gbwt_hip {

namespace {

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

// Lengths AND samples in one walk: a sequence does not know how many samples it will have until it ends, so the samples
// go to a pool in the order they are met (one atomic per sample), tagged with (sequence, sample number); once the lengths
// are there, k_place_samples puts them where k_record_samples would have.  Halves the open time of a large index.
struct PooledSampleSink {
    uint32_t wr = 0, next = 0, interval, number = 0, id;
    uint4 *pool;
    uint2 *tags;
    unsigned long long *counter;
    uint64_t capacity;
    __device__ __forceinline__ void push(uint32_t, bool counts) { wr += counts ? 1u : 0u; }
    __device__ __forceinline__ void checkpoint(uint32_t rec, uint32_t offset, uint32_t bb) {
        if (wr < next) return;
        const uint64_t at = atomicAdd(counter, 1ull);                // past the capacity: counted, not stored (the caller falls back)
        if (at < capacity) { pool[at] = make_uint4(rec, offset, bb, wr); tags[at] = make_uint2(id, number); }
        number++; next += interval;
    }
};

__global__ void __launch_bounds__(256) k_lengths_and_samples(DeviceIndex ix, uint32_t interval, uint32_t *seq_len, uint4 *pool, uint2 *tags,
                                                             unsigned long long *counter, uint64_t capacity, uint32_t *overflow) {
    const uint64_t id = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (id >= ix.n_sequences) return;
    PooledSampleSink sink;
    sink.interval = interval; sink.id = static_cast<uint32_t>(id);
    sink.pool = pool; sink.tags = tags; sink.counter = counter; sink.capacity = capacity;
    quiet_walk(ix, id, sink, overflow);
    seq_len[id] = sink.wr;
}

// every sample slot starts as "never reached" (cannot happen: every boundary lies below the length) ...
__global__ void __launch_bounds__(256) k_blank_samples(const uint32_t *seq_len, const uint64_t *sample_base, uint64_t n_sequences, uint4 *samples) {
    const uint64_t id = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (id >= n_sequences) return;
    for (uint64_t k = sample_base[id]; k < sample_base[id + 1]; k++) samples[k] = make_uint4(0u, 0u, BLOCK_NONE, seq_len[id]);
}

// ... and the pooled ones move to their places (a walk can meet one boundary more than the sequence has samples: the
// one at its very end)
__global__ void __launch_bounds__(256) k_place_samples(const uint4 *pool, const uint2 *tags, uint64_t pooled, const uint64_t *sample_base, uint4 *samples) {
    const uint64_t at = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (at >= pooled) return;
    const uint2 t = tags[at];
    const uint64_t base = sample_base[t.x], count = sample_base[t.x + 1] - base;
    if (t.y < count) samples[base + t.y] = pool[at];
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


// ---- checkpoint sampling: the sequence samples WITHOUT a walk of every sequence from end to end --------------------------
// The counting walk above follows each sequence with one lane: 666 668 dependent steps per lane on the headline index, 10 000 lanes
// -- 185 ms for what an extraction does in 4 (the one-shot flow of gbunzip, src/bin/gbunzip.rs:24-59, pays it once per file).
// Checkpoint sampling cuts the chains by RECORD instead of by sequence:
//   * a record is a CHECKPOINT when a multiplicative hash of its index falls below a threshold (no table, three instructions);
//   * one walker per BWT position of every checkpoint record, plus one per sequence start, walks forward -- the two-step walk of
//     the extraction, counting instead of emitting -- until an iteration ends on a checkpoint record, the sequence ends, or `cap`
//     nodes have been emitted.  Every BWT position is walked once by exactly one walker (LF is injective), at full occupancy;
//   * a walker that reaches the cap ends its hop there, at a position nobody was started for -- an ORPHAN, which gets a summary slot
//     of its own -- and walks on as the walker of that position; so every gap between two hops is at most cap + 3 nodes;
//   * what a walker leaves is a SUMMARY {record, offset, nodes walked, summary index of the position it landed on}: the
//     positions of one sequence form a linked list, about a thousand nodes per hop;
//   * one lane per sequence then chases its list (k_chase: one 16-byte load per hop instead of a thousand LF steps) and writes
//     the sequence samples and the length.
// The samples lie where sequences pass the same record, not every so many nodes of each sequence: rows that travel through the
// same records have their samples at the same records whatever happened to them upstream (an insertion shifts node counts, not
// checkpoints), so the walkers of a segment start together again.
constexpr uint32_t CP_HASH = 2654435761u;
constexpr uint32_t CP_END = 0xFFFFFFFFu;
// Checkpoints are records that are not unary: a unary record can lie INSIDE a fused edge (k_link_desc: an edge into a unary record
// emits it and lands behind it), where no walk ever stops; every other record a walk visits, it visits at the end of an LF step.
__device__ __forceinline__ bool hashed_checkpoint(uint32_t rec, uint32_t threshold) { return rec != 0 && rec * CP_HASH < threshold; }
__device__ __forceinline__ uint32_t checkpoint_len(const DeviceIndex &ix, uint64_t rec) {   // Record::len of a record that can be a checkpoint, else 0
    const uint4 B = ix.desc_raw[4 * rec + 1], C = ix.desc_raw[4 * rec + 2];
    return (B.y != 0 && B.y != DESC_UNARY) ? C.y : 0u;
}

__global__ void __launch_bounds__(256) k_checkpoint_counts(DeviceIndex ix, uint32_t threshold, uint64_t *counts) {
    const uint64_t rec = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (rec < ix.n_records) counts[rec] = hashed_checkpoint(static_cast<uint32_t>(rec), threshold) ? checkpoint_len(ix, rec) : 0u;
}

struct NodeCounter {
    uint32_t wr = 0;
    __device__ __forceinline__ void push(uint32_t, bool counts) { wr += counts ? 1u : 0u; }
};

// One iteration of the two-step walk that counts its nodes and emits nothing: on the packed half-blocks (three loads: the lane's
// half-block, then E_a and leaf (a, b) -- the gather loop of the extraction in plain C++, walk_loops.hpp), or one generic step where
// a record asks for it.  It stops BETWEEN its two LF steps when the first one lands on a hashed record (`threshold`; the caller
// then looks whether that record is a checkpoint): true, with (rec, offset) = that position and bb unknown.
__device__ __forceinline__ bool counting_step(const DeviceIndex &ix, bool packed, uint32_t threshold, NodeCounter &sink, uint32_t &rec, uint32_t &offset, uint32_t &bb) {
    const uint4 *d = ix.desc2 + 8 * static_cast<uint64_t>(rec);
    uint32_t a, b, rank_a, ones_w_base, mw;     // first-step value, second-step value, rank of the offset among the a-positions, counts of the second step
    uint4 E;
    if (packed) {
        const uint64_t idx = bb == BLOCK_NONE ? 0u : 2 * static_cast<uint64_t>(bb) + (offset >> 5);
        const uint4 K = ix.gblocks[idx];
        const uint32_t bit = offset & 31u, below = (1u << bit) - 1;
        a = (K.x >> bit) & 1u; b = (K.y >> bit) & 1u;
        E = d[a];
        if ((E.z & DESC2_SLOW) || !(E.w & GATHER_OK)) { packed = false; }
        else {
            const uint32_t ones1 = K.z & 0x1FFFFFu, R0 = ((K.z >> 21) | (K.w << 11)) & 0x1FFFFFu, R1 = K.w >> 10;
            const uint32_t m = (a ? K.x : ~K.x) & below;
            const uint32_t p = __popc(m);
            rank_a = a ? ones1 + p : (offset - bit) - ones1 + p;
            ones_w_base = a ? R1 : R0; mw = __popc(m & K.y);
        }
    }
    if (!packed) {
        const uint4 E0 = d[0];
        if (E0.z & DESC2_SLOW) { generic_step(ix, sink, rec, offset, bb); return false; }
        const uint64_t idx = bb == BLOCK_NONE ? 0u : bb + (offset >> RANK_BLOCK_SHIFT);
        const uint4 K0 = ix.cblocks[2 * idx], K1 = ix.cblocks[2 * idx + 1];
        const uint64_t bits1 = (static_cast<uint64_t>(K0.y) << 32) | K0.x, bits2 = (static_cast<uint64_t>(K0.w) << 32) | K0.z;
        const uint32_t bit = offset & 63u;
        const uint64_t below = (uint64_t(1) << bit) - 1;
        a = static_cast<uint32_t>(bits1 >> bit) & 1u; b = static_cast<uint32_t>(bits2 >> bit) & 1u;
        const uint64_t m = (a ? bits1 : ~bits1) & below;
        rank_a = a ? K1.x + __popcll(m) : (offset - bit) - K1.x + __popcll(m);
        E = a ? d[1] : E0;
        ones_w_base = a ? K1.z : K1.y; mw = __popcll(m & bits2);
    }
    const uint32_t j = E.y + rank_a, w = E.z & REC_MASK;
    sink.wr += (E.x != 0 ? 1u : 0u) + ((E.z & LEAF_EMIT2) ? 1u : 0u) + ((E.w & E_CHAIN) ? chain_mids(E.x, w + ix.alphabet_offset) : 0u);
    if (hashed_checkpoint(w, threshold)) { rec = w; offset = j; bb = BLOCK_NONE; return true; }
    const uint4 leaf = d[2 + 2 * a + b];
    const uint32_t ones_w = ones_w_base + mw;
    rec = leaf.z & REC_MASK; offset = leaf.y + (b ? ones_w : j - ones_w); bb = leaf.w;
    sink.wr += (leaf.x != 0 ? 1u : 0u) + ((leaf.z & LEAF_EMIT2) ? 1u : 0u) + ((leaf.z & LEAF_CHAIN) ? chain_mids(leaf.x, rec + ix.alphabet_offset) : 0u);
    return false;
}

struct CheckpointArgs {
    const uint64_t *cp_first;      // [n_records + 1]: first checkpoint position of every record (exclusive scan of k_checkpoint_counts)
    uint4 *summaries;              // [sequences + positions + orphan capacity]
    unsigned long long *orphan_count;
    uint64_t orphan_capacity;
    uint64_t positions;            // checkpoint positions = cp_first[n_records]
    uint32_t threshold, cap;
    uint32_t packed;               // the index has packed half-blocks (gblocks)
    uint32_t *flags;               // bit 2: the orphan pool is full (no samples from this pass)
};

__global__ void __launch_bounds__(256) k_checkpoint_walk(DeviceIndex ix, CheckpointArgs c) {
    const uint64_t S = ix.n_sequences;
    uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;       // the summary this walker is writing
    uint32_t rec = 0, offset = 0, bb = BLOCK_NONE;
    bool active = g < S + c.positions;
    if (active) {
        if (g < S) {                                           // a sequence start: the position after the start node (GBWT::start, src/gbwt.rs:213-219)
            if (g < ix.n_endmarker) {
                const uint2 e = ix.endmarker[g];
                offset = e.y;
                if (e.x == 0 || !arrive(ix, e.x, e.y, rec, bb)) { rec = 0; bb = BLOCK_NONE; }
            }
        } else {                                               // position w of the checkpoint records: the last record with cp_first <= w
            const uint64_t w = g - S;
            uint64_t lo = 0, hi = ix.n_records;                // cp_first[lo] <= w < cp_first[hi]
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) / 2;
                if (c.cp_first[mid] <= w) lo = mid; else hi = mid;
            }
            rec = static_cast<uint32_t>(lo); offset = static_cast<uint32_t>(w - c.cp_first[lo]); bb = ix.block_base[lo];
        }
        if (rec == 0) { c.summaries[g] = make_uint4(0u, 0u, 0u, CP_END); active = false; }
    }
    NodeCounter sink;
    const uint32_t lane = threadIdx.x % WAVE;
    while (__ballot(active) != 0) {
        bool orphan = false;
        if (active) {
            const bool between = counting_step(ix, c.packed != 0, c.threshold, sink, rec, offset, bb);
            if (rec == 0) { c.summaries[g] = make_uint4(0u, 0u, sink.wr, CP_END); active = false; }
            else if (hashed_checkpoint(rec, c.threshold) && offset < checkpoint_len(ix, rec)) {
                c.summaries[g] = make_uint4(rec, offset, sink.wr, static_cast<uint32_t>(S + c.cp_first[rec] + offset));
                active = false;
            } else {
                if (between) bb = ix.block_base[rec];          // a hashed record that is no checkpoint after all (unary): the walk goes on from it
                orphan = sink.wr >= c.cap;
            }
        }
        // The cap: the hop ends here, at a position no walker was started for -- an ORPHAN.  It gets a summary slot of its own and the
        // same lane walks on as its walker (the state is in its registers).  One atomic per wave: in a lock-step batch all 64 lanes
        // reach the cap together.
        const uint64_t mask = __ballot(orphan);
        if (mask != 0) {
            const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(mask))) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(c.orphan_count, static_cast<unsigned long long>(__popcll(mask)));
            base = __shfl(base, static_cast<int>(leader));
            if (orphan) {
                const uint64_t slot = base + __popcll(mask & ((uint64_t(1) << lane) - 1));
                if (slot < c.orphan_capacity) {
                    c.summaries[g] = make_uint4(rec, offset, sink.wr, static_cast<uint32_t>(S + c.positions + slot));
                    g = S + c.positions + slot;
                    sink.wr = 0;
                } else {
                    c.summaries[g] = make_uint4(0u, 0u, sink.wr, CP_END);
                    atomicOr(c.flags, 4u);
                    active = false;
                }
            }
        }
    }
}

// THE CHASE (round 4: by splitters).  The summaries of one sequence form a linked list, and the samples of the sequence are its
// elements in list order with the running sum of the nodes: list ranking.  One lane per sequence following its list took a
// microsecond per hop (a dependent 16-byte load from hundreds of megabytes), twice -- once to count, once to write: 4.4 ms of the
// headline's open with a sample every 512 nodes (2 320 hops per sequence).  Now one summary of sixteen (by a hash of its index) and
// every sequence start is a SPLITTER:
//   k_chase_spans     : every splitter walks to the next splitter: {next splitter, samples on the way, nodes on the way} -- a million
//                       and a half short walks at once;
//   k_chase_splitters : one lane per sequence follows the splitters only (a sixteenth of the hops), leaves each of them its
//                       {sequence, first sample number, nodes before it}, and the sequence its length and sample count;
//   k_chase_samples   : (after the scan of the counts) every splitter walks its span again and writes the samples: sample 0 of a
//                       sequence = the position after the start node (as the walker of segment 0 expects it), then one per hop.
constexpr uint32_t SPAN_END = 0xFFFFFFFFu, SPAN_UNVISITED = 0xFFFFFFFFu, SPAN_LIMIT = 4096;   // (4096 hops without a splitter: (15/16)^4096 -- a cycle of a corrupt index)
// exactly one of every sixteen consecutive summary indices, which one by a hash of the others' common bits: a splitter has the compact
// index p / 16 (behind the sequence starts), and which summaries are splitters has nothing to do with how records number their positions
__device__ __forceinline__ bool chase_splitter(uint64_t p, uint64_t n_sequences) {
    return p < n_sequences || (static_cast<uint32_t>(p) & 15u) == ((static_cast<uint32_t>(p >> 4) * CP_HASH) >> 28);
}
__device__ __forceinline__ uint64_t span_slot(uint64_t p, uint64_t n_sequences) { return p < n_sequences ? p : n_sequences + (p >> 4); }
// ... and the other way round: the splitter of slot t, or n_summaries where the slot has none (lanes are started per SLOT: one per summary
// had a sixteenth of every wave at work, 360 000 waves of 32 us each: 1.0 and 1.4 ms for the two walks over the spans)
__device__ __forceinline__ uint64_t slot_splitter(uint64_t t, uint64_t n_sequences, uint64_t n_summaries) {
    if (t < n_sequences) return t;
    const uint64_t g = t - n_sequences, p = 16 * g + ((static_cast<uint32_t>(g) * CP_HASH) >> 28);
    return (p < n_sequences || p >= n_summaries) ? n_summaries : p;
}

__global__ void __launch_bounds__(256) k_chase_spans(const uint4 *summaries, uint64_t n_summaries, uint64_t n_sequences, uint4 *spans, uint32_t *overflow) {
    const uint64_t p = slot_splitter(blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x, n_sequences, n_summaries);
    if (p >= n_summaries) return;
    uint64_t q = p, nodes = 0;
    uint32_t hops = 0, next = SPAN_END;
    for (;;) {
        const uint4 s = summaries[q];
        nodes += s.z;
        if (s.x == 0) break;                                        // the sequence ends inside this hop
        q = s.w;
        if (q >= n_summaries) { atomicOr(overflow, 2u); break; }
        if (++hops > SPAN_LIMIT) { atomicOr(overflow, 8u); break; }  // (a list that meets no splitter: the caller walks every sequence instead)
        if (chase_splitter(q, n_sequences)) { next = static_cast<uint32_t>(q); break; }
    }
    if (nodes > 0xFFFFFFF0ull) { atomicOr(overflow, 1u); nodes = 0; }
    spans[span_slot(p, n_sequences)] = make_uint4(next, hops, static_cast<uint32_t>(nodes), SPAN_UNVISITED);
}

__global__ void __launch_bounds__(64) k_chase_splitters(DeviceIndex ix, uint4 *spans, uint64_t n_summaries, uint32_t *seq_len, uint64_t *counts, uint32_t *overflow) {
    const uint64_t id = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (id >= ix.n_sequences) return;
    uint64_t n = 0, wr = 0;
    if (id < ix.n_endmarker && ix.endmarker[id].x != 0) {
        const uint2 e = ix.endmarker[id];
        uint32_t rec = 0, bb = BLOCK_NONE;
        n = 1; wr = 1;
        if (arrive(ix, e.x, e.y, rec, bb)) {
            uint64_t p = id, hops = 0;
            for (;;) {
                if (p >= n_summaries || ++hops > ix.max_walk) { atomicOr(overflow, 2u); break; }
                const uint64_t slot = span_slot(p, ix.n_sequences);
                const uint4 sp = spans[slot];
                if (sp.w != SPAN_UNVISITED) { atomicOr(overflow, 8u); break; }       // two sequences through one position (LF is injective in a GBWT; a mutated one is what the reference makes of it): the caller walks every sequence instead
                spans[slot] = make_uint4(sp.x, static_cast<uint32_t>(n), static_cast<uint32_t>(wr), static_cast<uint32_t>(id));
                n += sp.y; wr += sp.z;
                if (wr > 0xFFFFFFF0ull || n > 0xFFFFFFF0ull) { atomicOr(overflow, 1u); break; }
                if (sp.x == SPAN_END) break;
                p = sp.x;
            }
        }
    }
    seq_len[id] = static_cast<uint32_t>(wr); counts[id] = n;
}

__global__ void __launch_bounds__(256) k_chase_samples(DeviceIndex ix, const uint4 *summaries, uint64_t n_summaries, const uint4 *spans, const uint64_t *sample_base, uint4 *samples) {
    const uint64_t p = slot_splitter(blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x, ix.n_sequences, n_summaries);
    if (p >= n_summaries) return;
    const uint4 sp = spans[span_slot(p, ix.n_sequences)];
    if (p < ix.n_sequences && p < ix.n_endmarker && ix.endmarker[p].x != 0) {       // sample 0 of sequence p
        const uint2 e = ix.endmarker[p];
        uint32_t rec = 0, bb = BLOCK_NONE;
        if (!arrive(ix, e.x, e.y, rec, bb)) { rec = 0; bb = BLOCK_NONE; }
        if (sample_base[p + 1] > sample_base[p]) samples[sample_base[p]] = make_uint4(rec, e.y, bb, 1u);
    }
    if (sp.w == SPAN_UNVISITED) return;                              // on no sequence's list
    uint4 *out = samples + sample_base[sp.w];
    const uint64_t room = sample_base[sp.w + 1] - sample_base[sp.w];
    uint64_t q = p, n = sp.y, wr = sp.z;
    for (uint32_t hops = 0; hops <= SPAN_LIMIT; hops++) {
        const uint4 s = summaries[q];
        wr += s.z;
        if (s.x == 0) break;
        if (n < room) out[n] = make_uint4(s.x, s.y, ix.block_base[s.x], static_cast<uint32_t>(wr));
        n++;
        q = s.w;
        if (q >= n_summaries || chase_splitter(q, ix.n_sequences)) break;
    }
}

// ---- the line cache filled at open (kernels.hpp: LineCacheFill) -------------------------------------------------------------------------

// What a W-line makes of one node: '>' or '<' and the decimal digits of the node id (path_to_w_line, src/bin/gbunzip.rs:541-546); a P-line
// token is as long (digits and '+' / '-') plus the comma in front of all but the first (gfa.hip: cache_p_extra).
__device__ __forceinline__ uint32_t w_token_bytes(uint32_t node) {
    const uint32_t v = node >> 1;
    return 2u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}

// Counts what the nodes of one segment add to their line.  Only the first `quota` nodes belong to the segment (a sample may lie between the
// two LF steps of an iteration: the walk stages the rest like the extraction's walkers do, and nobody counts them).
struct TextSink {
    uint32_t wr = 0, quota = 0;
    uint64_t text = 0, labels = 0;
    uint64_t position = 0;           // of the next node in its path
    const uint32_t *label_len;
    uint64_t n_labels;
    uint64_t *chunk_text;            // of the path: chunk k of it at [k]
    uint32_t *chunk_seg;
    uint32_t segment = 0, first_node = 0;
    __device__ __forceinline__ void push(uint32_t node, bool counts) {
        if (!counts) return;
        if (wr < quota) {
            // a chunk of the formatter starts here: what this segment has in front of it (the segments before it are added by k_place_chunk_text)
            if ((position & (GFA_LINE_CHUNK - 1)) == 0 && position != 0) {
                const uint64_t c = position / GFA_LINE_CHUNK;
                chunk_text[c] = text; chunk_seg[c] = segment;
            }
            text += w_token_bytes(node);
            const uint64_t seq = (static_cast<uint64_t>(node & ~1u) - first_node) / 2;        // GBZ::graph_node_to_sequence, src/gbz.rs:246-255
            if ((node & ~1u) >= first_node && seq < n_labels) labels += label_len[seq];       // sequence_len(node).unwrap_or(0), src/bin/gbunzip.rs:534
            position++;
        }
        wr++;
    }
};

// One iteration of the two-step walk that hands every node to `sink` in the order the extraction emits them: on the packed half-blocks where
// the record allows it (counting_step above, with the nodes), else on the full-width blocks / the generic decoder (two_step).
template <class Sink>
__device__ __forceinline__ void emitting_step(const DeviceIndex &ix, bool packed, Sink &sink, uint32_t &rec, uint32_t &offset, uint32_t &bb) {
    const uint4 *d = ix.desc2 + 8 * static_cast<uint64_t>(rec);
    if (packed) {
        const uint64_t idx = bb == BLOCK_NONE ? 0u : 2 * static_cast<uint64_t>(bb) + (offset >> 5);
        const uint4 K = ix.gblocks[idx];
        const uint32_t bit = offset & 31u, below = (1u << bit) - 1;
        const uint32_t a = (K.x >> bit) & 1u, b = (K.y >> bit) & 1u;
        const uint4 E = d[a];
        if (!(E.z & DESC2_SLOW) && (E.w & GATHER_OK)) {
            const uint32_t ones1 = K.z & 0x1FFFFFu, R0 = ((K.z >> 21) | (K.w << 11)) & 0x1FFFFFu, R1 = K.w >> 10;
            const uint32_t m = (a ? K.x : ~K.x) & below;
            const uint32_t p = __popc(m);
            const uint32_t rank_a = a ? ones1 + p : (offset - bit) - ones1 + p;
            const uint32_t ones_w = (a ? R1 : R0) + __popc(m & K.y);
            const uint32_t j = E.y + rank_a, w = E.z & REC_MASK;
            const uint4 leaf = d[2 + 2 * a + b];
            rec = leaf.z & REC_MASK; offset = leaf.y + (b ? ones_w : j - ones_w); bb = leaf.w;
            sink.push(E.x, E.x != 0);
            if (E.w & E_CHAIN) push_chain(sink, E.x, w + ix.alphabet_offset);
            sink.push(w + ix.alphabet_offset, (E.z & LEAF_EMIT2) != 0);
            sink.push(leaf.x, leaf.x != 0);
            if (leaf.z & LEAF_CHAIN) push_chain(sink, leaf.x, rec + ix.alphabet_offset);
            sink.push(rec + ix.alphabet_offset, (leaf.z & LEAF_EMIT2) != 0);
            return;
        }
    }
    two_step(ix, sink, rec, offset, bb);
}

// Walker w = segment j of path p, j-major (w = j * paths + p): the lanes of a wave hold the same segment of neighbouring paths, as the
// walkers of an extraction do.  Paths with fewer segments leave their lanes idle.
__global__ void __launch_bounds__(256) k_segment_text(DeviceIndex ix, LineCacheFill f, uint32_t packed) {
    const uint64_t w = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (w >= static_cast<uint64_t>(f.max_samples) * f.paths) return;
    const uint64_t j = w / f.paths, p = w % f.paths, id = 2 * p;
    const uint64_t base = ix.sample_base[id], count = ix.sample_base[id + 1] - base;
    if (j >= count) return;
    const uint4 here = ix.samples[base + j];
    const uint64_t len = ix.seq_len[id];
    const uint64_t from = j == 0 ? 0 : here.w;                     // segment 0 starts with the start node (sample 0 is the state after it)
    const uint64_t to = j + 1 < count ? ix.samples[base + j + 1].w : len;
    TextSink sink;
    sink.label_len = f.label_len; sink.n_labels = f.n_labels; sink.chunk_text = f.chunk_text + f.chunk_first[p]; sink.chunk_seg = f.chunk_seg + f.chunk_first[p];
    sink.quota = to > from ? static_cast<uint32_t>(to - from) : 0u; sink.position = from;
    sink.segment = static_cast<uint32_t>(j); sink.first_node = ix.first_node;
    uint32_t rec = here.x, offset = here.y, bb = here.z;
    if (j == 0 && id < ix.n_endmarker && ix.endmarker[id].x != 0) sink.push(ix.endmarker[id].x, true);
    while (rec != 0 && sink.wr < sink.quota) emitting_step(ix, packed != 0, sink, rec, offset, bb);   // (every iteration emits a node or ends the walk)
    if (sink.wr < sink.quota) atomicOr(f.flags, 1u);               // the samples promise more nodes than the walk delivers: a corrupt index
    f.seg_text[2 * (base + j)] = sink.text; f.seg_text[2 * (base + j) + 1] = sink.labels;
}

// One wave per path: the token bytes in front of every segment (exclusive scan over the segments of the forward sequence, in place) and the
// totals of the path.
__global__ void __launch_bounds__(256) k_path_text(DeviceIndex ix, LineCacheFill f) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x / WAVE) + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    if (p >= f.paths) return;
    const uint64_t base = ix.sample_base[2 * p], count = ix.sample_base[2 * p + 1] - base;
    uint64_t text_before = 0, labels = 0;
    for (uint64_t j0 = 0; j0 < count; j0 += WAVE) {
        const uint64_t j = j0 + lane;
        const uint64_t mine = j < count ? f.seg_text[2 * (base + j)] : 0u;
        labels += j < count ? f.seg_text[2 * (base + j) + 1] : 0u;
        uint64_t incl = mine;
        for (int d = 1; d < WAVE; d <<= 1) { const uint64_t up = __shfl_up(incl, d, WAVE); if (static_cast<int>(lane) >= d) incl += up; }
        if (j < count) f.seg_text[2 * (base + j)] = text_before + incl - mine;
        text_before += __shfl(incl, WAVE - 1, WAVE);
    }
    for (int d = WAVE / 2; d > 0; d >>= 1) labels += __shfl_down(labels, d, WAVE);
    if (lane == 0) { f.path_totals[2 * p] = text_before; f.path_totals[2 * p + 1] = labels; }
}

// One thread per chunk: what its segment has in front of it + what the segments before that one have.
__global__ void __launch_bounds__(256) k_place_chunk_text(DeviceIndex ix, LineCacheFill f) {
    const uint64_t c = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (c >= f.chunks) return;
    uint64_t lo = 0, hi = f.paths;                                  // chunk_first[lo] <= c < chunk_first[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) / 2;
        if (f.chunk_first[mid] <= c) lo = mid; else hi = mid;
    }
    if (c == f.chunk_first[lo]) { f.chunk_text[c] = 0; return; }   // the first chunk of a path: nothing in front of it
    f.chunk_text[c] += f.seg_text[2 * (ix.sample_base[2 * lo] + f.chunk_seg[c])];
}

}  // namespace

void launch_fill_line_cache(const DeviceIndex &ix, const LineCacheFill &f, hipStream_t stream) {
    if (f.paths == 0) return;
    const uint64_t walkers = static_cast<uint64_t>(f.max_samples) * f.paths;
    const uint32_t packed = ix.gblocks != nullptr ? 1u : 0u;
    if (walkers) hipLaunchKernelGGL(k_segment_text, dim3(grid_for(walkers, 256)), dim3(256), 0, stream, ix, f, packed);
    hipLaunchKernelGGL(k_path_text, dim3(grid_for(f.paths, 256 / WAVE)), dim3(256), 0, stream, ix, f);
    if (f.chunks) hipLaunchKernelGGL(k_place_chunk_text, dim3(grid_for(f.chunks, 256)), dim3(256), 0, stream, ix, f);
}

void launch_sequence_lengths(const DeviceIndex &ix, uint32_t *d_seq_len, uint64_t *d_prints, uint32_t *d_overflow, hipStream_t stream) {
    if (ix.n_sequences == 0) return;
    const uint64_t x_inverse = fp_pow(FP_X, FP_P - 2);   // Fermat
    hipLaunchKernelGGL(k_sequence_lengths, dim3(grid_for(ix.n_sequences, 256)), dim3(256), 0, stream, ix, d_seq_len, d_prints, x_inverse, d_overflow);
}

void launch_check_orientation_pairs(const uint32_t *d_seq_len, const uint64_t *d_prints, uint64_t n_pairs, uint32_t *d_mismatch, hipStream_t stream) {
    if (n_pairs) hipLaunchKernelGGL(k_check_orientation_pairs, dim3(grid_for(n_pairs, 256)), dim3(256), 0, stream, d_seq_len, d_prints, n_pairs, d_mismatch);
}

void launch_lengths_and_samples(const DeviceIndex &ix, uint32_t interval, uint32_t *d_seq_len, uint4 *d_pool, uint2 *d_tags, uint64_t *d_counter,
                                uint64_t capacity, uint32_t *d_overflow, hipStream_t stream) {
    if (ix.n_sequences == 0) return;
    hipLaunchKernelGGL(k_lengths_and_samples, dim3(grid_for(ix.n_sequences, 256)), dim3(256), 0, stream, ix, interval, d_seq_len, d_pool, d_tags,
                       reinterpret_cast<unsigned long long *>(d_counter), capacity, d_overflow);
}

void launch_place_samples(const uint32_t *d_seq_len, const uint4 *d_pool, const uint2 *d_tags, uint64_t pooled, const uint64_t *d_sample_base,
                          uint64_t n_sequences, uint4 *d_samples, hipStream_t stream) {
    if (n_sequences == 0) return;
    hipLaunchKernelGGL(k_blank_samples, dim3(grid_for(n_sequences, 256)), dim3(256), 0, stream, d_seq_len, d_sample_base, n_sequences, d_samples);
    if (pooled) hipLaunchKernelGGL(k_place_samples, dim3(grid_for(pooled, 256)), dim3(256), 0, stream, d_pool, d_tags, pooled, d_sample_base, d_samples);
}

void launch_sample_counts(const uint32_t *d_seq_len, uint64_t n_sequences, uint32_t interval, uint64_t *d_counts, hipStream_t stream) {
    if (n_sequences) hipLaunchKernelGGL(k_sample_counts, dim3(grid_for(n_sequences, 256)), dim3(256), 0, stream, d_seq_len, n_sequences, interval, d_counts);
}

void launch_record_samples(const DeviceIndex &ix, const uint64_t *d_sample_base, uint32_t interval, uint4 *d_samples, hipStream_t stream) {
    if (ix.n_sequences == 0) return;
    hipLaunchKernelGGL(k_record_samples, dim3(grid_for(ix.n_sequences, 256)), dim3(256), 0, stream, ix, d_sample_base, interval, d_samples,
                       static_cast<uint32_t *>(nullptr));
}


void launch_checkpoint_counts(const DeviceIndex &ix, uint32_t threshold, uint64_t *d_counts, hipStream_t stream) {
    if (ix.n_records) hipLaunchKernelGGL(k_checkpoint_counts, dim3(grid_for(ix.n_records, 256)), dim3(256), 0, stream, ix, threshold, d_counts);
}

void launch_checkpoint_walk(const DeviceIndex &ix, const CheckpointWalk &w, hipStream_t stream) {
    const uint64_t count = ix.n_sequences + w.positions;
    if (count == 0) return;
    CheckpointArgs c{};
    c.cp_first = w.cp_first; c.summaries = w.summaries; c.orphan_count = reinterpret_cast<unsigned long long *>(w.orphan_count);
    c.orphan_capacity = w.orphan_capacity; c.positions = w.positions; c.threshold = w.threshold; c.cap = w.cap; c.packed = w.packed;
    c.flags = w.flags;
    hipLaunchKernelGGL(k_checkpoint_walk, dim3(grid_for(count, 256)), dim3(256), 0, stream, ix, c);
}

void launch_chase_counts(const DeviceIndex &ix, const uint4 *d_summaries, uint64_t n_summaries, uint4 *d_spans, uint32_t *d_seq_len, uint64_t *d_counts,
                         uint32_t *d_overflow, hipStream_t stream) {
    if (ix.n_sequences == 0) return;
    hipLaunchKernelGGL(k_chase_spans, dim3(grid_for(ix.n_sequences + n_summaries / 16 + 1, 256)), dim3(256), 0, stream, d_summaries, n_summaries, ix.n_sequences, d_spans, d_overflow);
    hipLaunchKernelGGL(k_chase_splitters, dim3(grid_for(ix.n_sequences, 64)), dim3(64), 0, stream, ix, d_spans, n_summaries, d_seq_len, d_counts, d_overflow);
}

void launch_chase_samples(const DeviceIndex &ix, const uint4 *d_summaries, uint64_t n_summaries, const uint4 *d_spans, const uint64_t *d_sample_base,
                          uint4 *d_samples, hipStream_t stream) {
    if (ix.n_sequences == 0) return;
    hipLaunchKernelGGL(k_chase_samples, dim3(grid_for(ix.n_sequences + n_summaries / 16 + 1, 256)), dim3(256), 0, stream, ix, d_summaries, n_summaries, d_spans, d_sample_base, d_samples);
}

}  // namespace gbwt_hip
