// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the search kernels (a blanket x 2 -- the guide's
// factor for wide coalesced streaming reads -- turned 1.79 GB of counted fetches into 3.47 GB = 7.1 TB/s, which the part cannot deliver).
// Every lane reads ONE random, aligned entry of WIDTH bytes (16: a rank block; 64: a record descriptor, four 16-byte loads of one lane)
// out of a table far larger than the 256 MiB Infinity Cache, so that the bytes fetched from HBM are known: lanes x WIDTH (x the line
// granularity: a 16-byte read still moves a whole 64- or 128-byte line -- which is what the counter should then show).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_fetch tools/microbench_fetch.hip
//   run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- tools/microbench_fetch
// Prints per kernel the bytes its lanes asked for; tools/fetch_calibration.py puts the counter values next to them.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

// scattered: lane g reads entry mix(g) % entries, LOADS x 16 bytes of it
template <int LOADS>
__global__ void __launch_bounds__(256) k_scattered(const uint4 *table, uint64_t entries, uint64_t entry_uint4, uint32_t *sink) {
    const uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const uint4 *e = table + (mix(g) % entries) * entry_uint4;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < LOADS; k++) { const uint4 v = e[k]; acc += v.x ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// streaming: lane g reads uint4 g (the guide's calibrated case: the counter shows half)
__global__ void __launch_bounds__(256) k_streaming(const uint4 *table, uint64_t n, uint32_t *sink) {
    const uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (g >= n) return;
    const uint4 v = table[g];
    if ((v.x ^ v.w) == 0x12345678u) sink[0] = v.x;
}

int main() {
    const uint64_t bytes = uint64_t(16) << 30;                 // 16 GiB: 64 x the Infinity Cache
    uint4 *table = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(table, 1, bytes));
    const uint64_t lanes = uint64_t(1) << 25;                  // 33.5 M lanes per launch
    const unsigned blocks = static_cast<unsigned>(lanes / 256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_scattered<1>, dim3(blocks), dim3(256), 0, nullptr, table, bytes / 16, uint64_t(1), sink);      // 16 B of a 16-byte entry
        hipLaunchKernelGGL(k_scattered<1>, dim3(blocks), dim3(256), 0, nullptr, table, bytes / 64, uint64_t(4), sink);      // 16 B of a 64-byte entry
        hipLaunchKernelGGL(k_scattered<4>, dim3(blocks), dim3(256), 0, nullptr, table, bytes / 64, uint64_t(4), sink);      // a whole 64-byte entry
        hipLaunchKernelGGL(k_scattered<8>, dim3(blocks), dim3(256), 0, nullptr, table, bytes / 128, uint64_t(8), sink);     // a whole 128-byte entry
        hipLaunchKernelGGL(k_streaming, dim3(blocks), dim3(256), 0, nullptr, table, lanes, sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("lanes per launch %llu\n", static_cast<unsigned long long>(lanes));
    printf("k_scattered<1> entry 16 B : asked %llu bytes, touches %llu lines of 64 B\n", (unsigned long long)(lanes * 16), (unsigned long long)lanes);
    printf("k_scattered<1> entry 64 B : asked %llu bytes, touches %llu lines of 64 B\n", (unsigned long long)(lanes * 16), (unsigned long long)lanes);
    printf("k_scattered<4> entry 64 B : asked %llu bytes\n", (unsigned long long)(lanes * 64));
    printf("k_scattered<8> entry 128 B: asked %llu bytes\n", (unsigned long long)(lanes * 128));
    printf("k_streaming               : asked %llu bytes\n", (unsigned long long)(lanes * 16));
    return 0;
}
