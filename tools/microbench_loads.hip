// Latency of one "round trip" made of K independent 16-byte loads per lane, as a function of K, the number of enabled
// lanes and whether the lanes read the same or different lines.  Every round's addresses depend on the previous
// round's data (pointer chase), so rounds cannot overlap.  Working set 1 MiB: everything is an L2 hit after warm-up.
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_loads tools/microbench_loads.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// table: n entries of 128 bytes; entry e word 0 = next entry (random permutation)
template <int K, bool SAME_LINE>
__global__ void k_round(const uint4 *table, uint32_t n, int rounds, uint32_t lanes, uint64_t *out) {
    const uint32_t lane = threadIdx.x;
    uint32_t e = (lane * 2654435761u) % n;
    if (lane >= lanes) return;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint4 v[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            // SAME_LINE: the K loads of a lane read one 128-byte entry; otherwise K different entries
            const uint64_t idx = SAME_LINE ? (static_cast<uint64_t>(e) * 8 + (k & 7)) : (static_cast<uint64_t>((e + k * 7919u) % n) * 8);
            v[k] = table[idx];
        }
        uint32_t nx = v[0].x;
#pragma unroll
        for (int k = 1; k < K; k++) acc += v[k].y;
        e = nx;
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = acc + e; }
}

// S loads from an entry all lanes share (a record's descriptor) + D loads from a lane's own entry (its rank block)
template <int S, int D>
__global__ void k_mixed(const uint4 *table, uint32_t n, int rounds, uint32_t lanes, uint64_t *out) {
    const uint32_t lane = threadIdx.x;
    uint32_t e = (lane * 2654435761u) % n, shared = 17;
    if (lane >= lanes) return;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint4 v[S + D];
#pragma unroll
        for (int k = 0; k < S; k++) v[k] = table[static_cast<uint64_t>(shared) * 8 + (k & 7)];
#pragma unroll
        for (int k = 0; k < D; k++) v[S + k] = table[static_cast<uint64_t>(e) * 8 + (k & 7)];
#pragma unroll
        for (int k = 1; k < S + D; k++) acc += v[k].y;
        shared = v[0].x;            // both chains advance with loaded data
        e = v[S].x;
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = acc + e + shared; }
}

template <int S, int D>
static double run_mixed(const uint4 *d_table, uint32_t n, uint32_t lanes, uint64_t *d_out) {
    const int rounds = 4000;
    hipLaunchKernelGGL((k_mixed<S, D>), dim3(1), dim3(64), 0, 0, d_table, n, 200, lanes, d_out);
    hipLaunchKernelGGL((k_mixed<S, D>), dim3(1), dim3(64), 0, 0, d_table, n, rounds, lanes, d_out);
    uint64_t h[2];
    (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    return double(h[0]) / rounds;
}

template <int K, bool SAME>
static double run(const uint4 *d_table, uint32_t n, uint32_t lanes, uint64_t *d_out) {
    const int rounds = 4000;
    hipLaunchKernelGGL((k_round<K, SAME>), dim3(1), dim3(64), 0, 0, d_table, n, 200, lanes, d_out);   // warm the L2
    hipLaunchKernelGGL((k_round<K, SAME>), dim3(1), dim3(64), 0, 0, d_table, n, rounds, lanes, d_out);
    uint64_t h[2];
    (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    return double(h[0]) / rounds;
}

int main() {
    const uint32_t n = 8192;   // 1 MiB
    std::vector<uint4> table(size_t(n) * 8);
    std::vector<uint32_t> perm(n);
    for (uint32_t i = 0; i < n; i++) perm[i] = i;
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = n - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; uint32_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
    for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 8; k++) table[size_t(perm[i]) * 8 + k] = make_uint4(perm[(i + 1) % n], k, 0, 0);
    uint4 *d_table; uint64_t *d_out;
    CHECK(hipMalloc(&d_table, table.size() * sizeof(uint4)));
    CHECK(hipMalloc(&d_out, 16));
    CHECK(hipMemcpy(d_table, table.data(), table.size() * sizeof(uint4), hipMemcpyHostToDevice));
    printf("one wave, s_memtime ticks per round (a round = K independent dwordx4 loads per lane, next round depends on it)\n");
    printf("%-28s %8s %8s %8s %8s\n", "", "1 lane", "16", "32", "64");
#define ROW(K, SAME, name) printf("%-28s %8.0f %8.0f %8.0f %8.0f\n", name, run<K, SAME>(d_table, n, 1, d_out), run<K, SAME>(d_table, n, 16, d_out), run<K, SAME>(d_table, n, 32, d_out), run<K, SAME>(d_table, n, 64, d_out))
    ROW(1, true, "K=1");
    ROW(2, true, "K=2 same line");
    ROW(4, true, "K=4 same line");
    ROW(8, true, "K=8 same line");
    ROW(2, false, "K=2 different lines");
    ROW(4, false, "K=4 different lines");
    ROW(8, false, "K=8 different lines");
#define MROW(S, D, name) printf("%-28s %8.0f %8.0f %8.0f %8.0f\n", name, run_mixed<S, D>(d_table, n, 1, d_out), run_mixed<S, D>(d_table, n, 16, d_out), run_mixed<S, D>(d_table, n, 32, d_out), run_mixed<S, D>(d_table, n, 64, d_out))
    MROW(1, 1, "1 shared + 1 own");
    MROW(2, 1, "2 shared + 1 own");
    MROW(4, 1, "4 shared + 1 own");
    MROW(7, 2, "7 shared + 2 own");
    MROW(3, 2, "3 shared + 2 own");
    return 0;
}
