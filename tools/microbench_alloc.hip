// What a first request pays for its buffers: hipMalloc of N GiB, the first kernel that writes all of it, the second one, hipFree.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_alloc tools/microbench_alloc.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_fill(uint4 *p, size_t n) { size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; const size_t step = size_t(gridDim.x) * blockDim.x; for (; i < n; i += step) p[i] = make_uint4(1, 2, 3, 4); }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    hipFree(nullptr);
    for (int a = 1; a < argc; a++) {
        const size_t gib = std::strtoull(argv[a], nullptr, 10), bytes = gib << 30;
        void *p = nullptr;
        double t0 = now();
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("%zu GiB: hipMalloc failed\n", gib); continue; }
        double t1 = now();
        hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, nullptr, static_cast<uint4 *>(p), bytes / 16);
        hipDeviceSynchronize();
        double t2 = now();
        hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, nullptr, static_cast<uint4 *>(p), bytes / 16);
        hipDeviceSynchronize();
        double t3 = now();
        hipFree(p);
        double t4 = now();
        printf("%3zu GiB: hipMalloc %8.1f ms   first fill %8.1f ms   second fill %7.1f ms   hipFree %8.1f ms\n", gib, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    }
    return 0;
}
