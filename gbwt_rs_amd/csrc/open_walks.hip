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

}  // namespace

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

}  // namespace gbwt_hip
