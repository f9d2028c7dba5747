// Is v_mov_b64 (SGPR pair -> VGPR pair) a full-rate VALU instruction on gfx950?  N x 32 of them against N x 64 v_mov_b32 writing the same registers.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/microbench_mov64 tools/microbench_mov64.hip && /tmp/microbench_mov64
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X X X X X X X X
template <int WIDE>
__global__ void __launch_bounds__(256) k(uint32_t *out, const uint32_t *in, int n) {
    uint32_t s0 = in[0], s1 = in[1], a = 0, b = 0;
    for (int i = 0; i < n; i++) {
        if (WIDE) asm volatile(R8("v_mov_b64 v[80:81], s[56:57]\n\tv_mov_b64 v[82:83], s[56:57]\n\tv_mov_b64 v[84:85], s[56:57]\n\tv_mov_b64 v[86:87], s[56:57]\n\t")
                               "v_mov_b32 %0, v80\n\tv_mov_b32 %1, v87" : "=v"(a), "=v"(b) : "{s56}"(s0 + i), "{s57}"(s1) : "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
        else asm volatile(R8("v_mov_b32 v80, s56\n\tv_mov_b32 v81, s57\n\tv_mov_b32 v82, s56\n\tv_mov_b32 v83, s57\n\tv_mov_b32 v84, s56\n\tv_mov_b32 v85, s57\n\tv_mov_b32 v86, s56\n\tv_mov_b32 v87, s57\n\t")
                          "v_mov_b32 %0, v80\n\tv_mov_b32 %1, v87" : "=v"(a), "=v"(b) : "{s56}"(s0 + i), "{s57}"(s1) : "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}
int main() {
    uint32_t *out, *in; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 64); hipMemset(in, 1, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) for (int wide = 0; wide < 2; wide++) {
        hipEventRecord(e0);
        if (wide) hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, out, in, 2000); else hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, out, in, 2000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms for the same 64 registers written x 2000 x 16384 waves (%d instructions per iteration)\n", wide ? "v_mov_b64" : "v_mov_b32", ms, wide ? 32 : 64);
    }
    return 0;
}
