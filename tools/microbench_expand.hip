// The stream of the GFA formatter without the formatting: every workgroup reads a tile of 4 KB (the node ids of 1 024 positions) and
// writes a tile of 9 KB (their text, ~9 bytes per position on config 4) with 16-byte stores -- what the memory system gives a kernel
// with k_format_chunks' mix of reads and writes (1 : 2.25), its grid (one workgroup per 4 096 positions = four tiles) and nothing else.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_expand tools/microbench_expand.hip
//   run:   tools/microbench_expand [GiB of input, default 4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr unsigned THREADS = 256, IN_UNITS = 256, OUT_UNITS = 576, TILES = 4;   // 16-byte units per tile; tiles per workgroup
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, bool THROUGH_LDS>
__global__ void __launch_bounds__(THREADS) k_expand(const u32x4 *in, u32x4 *out, size_t groups) {
    __shared__ u32x4 stage[OUT_UNITS];
    const size_t g = blockIdx.x;
    if (g >= groups) return;
    const unsigned t = threadIdx.x;
    u32x4 ahead = in[(g * TILES) * IN_UNITS + t];
    for (unsigned tile = 0; tile < TILES; tile++) {
        const u32x4 v = ahead;
        if (tile + 1 < TILES) ahead = in[(g * TILES + tile + 1) * IN_UNITS + t];
        u32x4 *to = out + (g * TILES + tile) * OUT_UNITS;
        if (THROUGH_LDS) {
            stage[t] = v;
            stage[t + 256] = v + 1u;
            if (t < OUT_UNITS - 512) stage[t + 512] = v + 2u;
            __syncthreads();
            for (unsigned u = t; u < OUT_UNITS; u += THREADS) {
                if (NT) __builtin_nontemporal_store(stage[OUT_UNITS - 1 - u], to + u); else to[u] = stage[OUT_UNITS - 1 - u];
            }
            __syncthreads();
        } else {
            for (unsigned u = t, k = 0; u < OUT_UNITS; u += THREADS, k++) {
                if (NT) __builtin_nontemporal_store(v + k, to + u); else to[u] = v + k;
            }
        }
    }
}

template <bool NT, bool THROUGH_LDS>
static void run(const char *name, const u32x4 *in, u32x4 *out, size_t groups) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps + 1; r++) {
        hipEventRecord(a, nullptr);
        hipLaunchKernelGGL((k_expand<NT, THROUGH_LDS>), dim3(static_cast<unsigned>(groups)), dim3(THREADS), 0, nullptr, in, out, groups);
        hipEventRecord(b, nullptr);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (r == 0) continue;
        sum += ms;
        if (ms < best) best = ms;
    }
    const double bytes = static_cast<double>(groups) * TILES * (IN_UNITS + OUT_UNITS) * 16.0;
    printf("%-44s %8.3f ms (best %8.3f)   %6.2f TB/s read + written (%.2f GB read, %.2f GB written)\n", name, sum / reps, best, bytes / (sum / reps) / 1e9,
           groups * TILES * IN_UNITS * 16.0 / 1e9, groups * TILES * OUT_UNITS * 16.0 / 1e9);
}

int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4;
    const size_t groups = (gib << 30) / (TILES * IN_UNITS * 16);
    u32x4 *in = nullptr, *out = nullptr;
    if (hipMalloc(&in, groups * TILES * IN_UNITS * 16) != hipSuccess || hipMalloc(&out, groups * TILES * OUT_UNITS * 16) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(in, 1, groups * TILES * IN_UNITS * 16);
    hipMemset(out, 0, groups * TILES * OUT_UNITS * 16);
    hipDeviceSynchronize();
    printf("%zu workgroups of %u threads, %u tiles each: 4 KB read, 9 KB written per tile\n", groups, THREADS, TILES);
    run<false, false>("plain stores from registers", in, out, groups);
    run<true, false>("nontemporal stores from registers", in, out, groups);
    run<false, true>("plain stores through LDS, two barriers", in, out, groups);
    run<true, true>("nontemporal stores through LDS, two barriers", in, out, groups);
    hipFree(in); hipFree(out);
    return 0;
}
