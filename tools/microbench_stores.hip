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

// (d) the REAL address pattern of the segmented extraction: 5 000 rows of 666 668 nodes, 326 segments of 2 048 nodes; workgroup g
// (one wave here) holds segment g / 79 of rows 64 (g % 79) .. + 63 and writes, round after round, the next 128-byte piece of
// every one of its 64 rows: eight store instructions of eight pieces each (rows 2.67 MB apart), `nap` sleeps between rounds.
// Pieces are whole 128-byte lines, as the row writer of the kernel aligns them (unaligned pieces under nt: 1.2 TB/s).
template <bool NT>
__global__ void __launch_bounds__(64) k_rows(uint8_t *out, uint32_t rows, uint32_t row_nodes, uint32_t seg_nodes, uint32_t groups, uint32_t nap) {
    const uint32_t lane = threadIdx.x;
    const uint32_t j = blockIdx.x / groups, first = (blockIdx.x % groups) * 64u;
    for (uint32_t round = 0; round < seg_nodes / 32; round++) {
        for (uint32_t g = 0; g < 8; g++) {
            const uint32_t row = first + 8 * g + lane / 8;
            if (row >= rows) continue;
            const uint64_t node = static_cast<uint64_t>(row) * row_nodes + static_cast<uint64_t>(j) * seg_nodes + round * 32ull + (lane % 8) * 4;
            if (static_cast<uint64_t>(j) * seg_nodes + round * 32ull + 32 > row_nodes) continue;
            u32x4 *at = reinterpret_cast<u32x4 *>(out + (((node - (lane % 8) * 4) * 4) & ~127ull) + (lane % 8) * 16);   // whole lines, as the row writer aligns them
            const u32x4 v = u32x4{1u, 2u, 3u, round};
            if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(at), "v"(v) : "memory");
            else *at = v;
        }
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
    {
        const uint32_t rows = 5000, row_nodes = 666668, seg_nodes = 2048, groups = (rows + 63) / 64, segs = (row_nodes + seg_nodes - 1) / seg_nodes;
        for (int nt = 1; nt >= 0; nt--)
            for (uint32_t nap : {0u, 1u, 2u, 4u}) {
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0));
                    if (nt) hipLaunchKernelGGL(k_rows<true>, dim3(groups * segs), dim3(64), 0, 0, buf, rows, row_nodes, seg_nodes, groups, nap);
                    else hipLaunchKernelGGL(k_rows<false>, dim3(groups * segs), dim3(64), 0, 0, buf, rows, row_nodes, seg_nodes, groups, nap);
                    char what[160];
                    snprintf(what, sizeof what, "real pattern: 64 rows x 128 B per round, rows 2.67 MB apart, %s, %u naps", nt ? "nt" : "plain", nap);
                    report(what);
                }
            }
    }
    CHECK(hipFree(buf));
    return 0;
}
