// microbench.hip -- single-wave latency of the primitives the walk kernels are built from (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip ; run: ./microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int N = 256;

__device__ __forceinline__ uint32_t dpp_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return x;
}

__global__ void k_alu(uint64_t *out, uint32_t seed) {
    uint32_t lane = threadIdx.x, x = seed + lane;
    uint64_t t0, t1;
    // (0) empty timer pair
    t0 = __builtin_amdgcn_s_memtime(); t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = t1 - t0;
    // (1) dependent VALU chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; i++) { x = x * 3u + 1u; x ^= x >> 3; }  // 2-3 VALU per iteration
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[1] = t1 - t0;
    // (2) ballot -> scalar add -> inverse ballot -> select   (VALU -> SALU -> VALU round trip)
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        uint64_t m = __ballot((x & 1u) != 0);
        m = m + (m << 1);
        x += __builtin_amdgcn_inverse_ballot_w64(m) ? 3u : 5u;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[2] = t1 - t0;
    // (3) readlane with a data-dependent (scalar) lane, result fed back into a VALU op
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        uint32_t l = __builtin_amdgcn_readfirstlane(x) & 63u;
        x += (uint32_t)__builtin_amdgcn_readlane((int)x, (int)l);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[3] = t1 - t0;
    // (4) ds_bpermute dependent chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; i++) x += (uint32_t)__shfl((int)x, (int)((x >> 2) & 63u));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[4] = t1 - t0;
    // (5) DPP inclusive scan dependent chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; i++) x = dpp_scan(x & 0xFFu);
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[5] = t1 - t0;
    // (6) ballot + ctz + readlane (find first lane with property, fetch its value)
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        uint64_t m = __ballot(x > (uint32_t)i) | (1ull << 63);
        uint32_t j = (uint32_t)__builtin_ctzll(m);
        x += (uint32_t)__builtin_amdgcn_readlane((int)x, (int)j);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[6] = t1 - t0;
    // (7) 4 independent bpermutes per round (ILP), dependent rounds
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        uint32_t a = (uint32_t)__shfl((int)x, (int)((x) & 63u)), b = (uint32_t)__shfl((int)x, (int)((x >> 6) & 63u));
        uint32_t c = (uint32_t)__shfl((int)x, (int)((x >> 12) & 63u)), d = (uint32_t)__shfl((int)x, (int)((x >> 18) & 63u));
        x += a ^ b ^ c ^ d;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[7] = t1 - t0;
    // (8) LDS write + read dependent chain
    __shared__ uint32_t lds[64];
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; i++) { lds[lane] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); x += lds[(x >> 1) & 63u]; }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[8] = t1 - t0;
    out[16 + lane] = x;
}

// pointer chase: each lane follows its own chain; all lanes read the same line (broadcast) when `same`.
__global__ void k_chase(const uint32_t *next, uint32_t start, int hops, uint64_t *out, int slot) {
    uint32_t p = start;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < hops; i++) p = next[p];
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[slot] = t1 - t0; out[slot + 1] = p; }
}

int main() {
    uint64_t *d_out, h_out[96];
    CHECK(hipMalloc(&d_out, sizeof(h_out)));
    CHECK(hipMemset(d_out, 0, sizeof(h_out)));
    hipLaunchKernelGGL(k_alu, dim3(1), dim3(64), 0, 0, d_out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
    const char *names[] = {"empty s_memtime pair", "VALU dependent (3 ops/iter)", "ballot->s_add->inverse_ballot->select", "readfirstlane->readlane->add",
                           "ds_bpermute dependent", "DPP scan (6 dpp adds)", "ballot+ctz+readlane", "4x bpermute (ILP) per round", "LDS write+fence+read"};
    printf("single wave, cycles per iteration (s_memtime ticks; %d iterations)\n", N);
    for (int i = 0; i < 9; i++) printf("  %-42s %8.1f\n", names[i], i == 0 ? (double)h_out[0] : (double)(h_out[i] - h_out[0]) / N);
    // pointer chase at several working-set sizes (stride 256 B to defeat line reuse)
    size_t sizes[] = {4u << 10, 16u << 10, 256u << 10, 2u << 20, 16u << 20, 128u << 20, 1024u << 20};
    for (size_t sz : sizes) {
        size_t n = sz / 4, stride = 64;  // 64 words = 256 B
        std::vector<uint32_t> h(n, 0);
        size_t slots = n / stride;
        // random cyclic permutation over the slots
        std::vector<uint32_t> perm(slots);
        for (size_t i = 0; i < slots; i++) perm[i] = (uint32_t)i;
        uint64_t s = 88172645463325252ull;
        for (size_t i = slots - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
        for (size_t i = 0; i < slots; i++) h[perm[i] * stride] = perm[(i + 1) % slots] * (uint32_t)stride;
        uint32_t *d;
        CHECK(hipMalloc(&d, sz));
        CHECK(hipMemcpy(d, h.data(), sz, hipMemcpyHostToDevice));
        int hops = 2000;
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d, 0u, (int)slots < hops ? (int)slots * 4 : hops, d_out, 64);  // warm
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d, 0u, hops, d_out, 66);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
        printf("  pointer chase, working set %8zu KiB: %8.1f cycles/load\n", sz >> 10, (double)h_out[66] / hops);
        CHECK(hipFree(d));
    }
    return 0;
}
