// Store-side floor of the segmented extraction: how fast can 13.3 GB be written (a) fully coalesced, (b) in the
// pattern of the cooperative row writes: a wave writes eight 128-byte pieces per instruction, each piece the next one
// of a 16 KB segment of its own; (c) as (b) with a sleep between pieces (the rate at which a walker produces them).
// hipcc --offload-arch=gfx950 -O3 -o microbench_stores microbench_stores.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_coalesced(u32x4 *out, uint64_t n16) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) out[i] = u32x4{1u, 2u, 3u, (uint32_t)i};
}

// segments of `seg_bytes`; wave w owns segments 8w .. 8w+7, lane l writes 16 bytes of piece `it` of segment 8w + l / 8
__global__ void __launch_bounds__(64) k_pieces(uint8_t *out, uint64_t segments, uint32_t seg_bytes, uint32_t nap) {
    const uint32_t lane = threadIdx.x;
    const uint64_t seg = blockIdx.x * 8ull + lane / 8;
    if (seg >= segments) return;
    uint8_t *p = out + seg * seg_bytes + (lane % 8) * 16;
    for (uint32_t it = 0; it < seg_bytes / 128; it++) {
        *reinterpret_cast<u32x4 *>(p + it * 128ull) = u32x4{1u, 2u, 3u, it};
        for (uint32_t k = 0; k < nap; k++) __builtin_amdgcn_s_sleep(16);
    }
}

int main(int argc, char **argv) {
    const uint64_t bytes = 13333360000ull / 16384 * 16384;
    uint8_t *buf;
    CHECK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto report = [&](const char *what) {
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-60s %8.3f ms  %7.2f TB/s\n", what, ms, bytes / ms / 1e9);
    };
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_coalesced, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<u32x4 *>(buf), bytes / 16);
        report("coalesced 16 B per lane, grid-stride");
    }
    const uint32_t seg_bytes = 16384;
    const uint64_t segments = bytes / seg_bytes;
    for (uint32_t nap : {0u, 1u, 2u, 4u, 8u}) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_pieces, dim3((segments + 7) / 8), dim3(64), 0, 0, buf, segments, seg_bytes, nap);
            char what[128];
            snprintf(what, sizeof what, "8 x 128 B pieces per wave store, 16 KB segments, %u naps of 1024 cycles", nap);
            report(what);
        }
    }
    CHECK(hipFree(buf));
    return 0;
}
