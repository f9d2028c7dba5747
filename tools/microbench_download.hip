// microbench_download.hip -- device-to-host copies of a query call's results (round 6): 48 MB into fresh / touched / pinned host memory, in one
// piece and through pinned staging buffers emptied by T threads (what copy_to_host does).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    CK(hipSetDevice(0)); CK(hipFree(nullptr));
    const size_t bytes = 48u << 20;
    void *src; CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 5, bytes));
    { void *w = malloc(bytes); memset(w, 1, bytes); CK(hipMemcpy(w, src, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(src, w, bytes, hipMemcpyHostToDevice)); free(w); }   // runtime warm
    for (int rep = 0; rep < 3; rep++) {
        char *fresh = static_cast<char *>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        double t0 = now(); CK(hipMemcpy(fresh, src, bytes, hipMemcpyDeviceToHost)); double a = now() - t0;
        t0 = now(); CK(hipMemcpy(fresh, src, bytes, hipMemcpyDeviceToHost)); double b = now() - t0;
        munmap(fresh, bytes);
        char *zeroed = static_cast<char *>(calloc(bytes, 1));
        t0 = now(); CK(hipMemcpy(zeroed, src, bytes, hipMemcpyDeviceToHost)); double c = now() - t0;
        free(zeroed);
        char *touched = static_cast<char *>(malloc(bytes)); 
        t0 = now(); memset(touched, 0, bytes); double m = now() - t0;
        t0 = now(); CK(hipMemcpy(touched, src, bytes, hipMemcpyDeviceToHost)); double d = now() - t0;
        // pre-fault by 8 threads, then one copy
        char *par = static_cast<char *>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        t0 = now();
        { std::vector<std::thread> pool; for (int t = 0; t < 8; t++) pool.emplace_back([=]() { for (size_t o = bytes / 8 * t; o < bytes / 8 * (t + 1); o += 4096) par[o] = 0; }); for (auto &t : pool) t.join(); }
        double pf = now() - t0;
        t0 = now(); CK(hipMemcpy(par, src, bytes, hipMemcpyDeviceToHost)); double e = now() - t0;
        munmap(par, bytes);
        printf("D2H 48 MB: fresh mmap %.2f ms (again %.2f), calloc %.2f, malloc+memset(%.2f) then copy %.2f, 8-thread prefault(%.2f) then copy %.2f\n", a, b, c, m, d, pf, e);
        free(touched);
    }
    // staged: T threads, pinned 2 x 2 MB each, memcpy out of pinned into fresh memory
    for (unsigned threads : {4u, 8u}) {
        const size_t piece = 2u << 20;
        std::vector<void *> pinned(2 * threads); std::vector<hipStream_t> streams(threads);
        for (auto &p : pinned) CK(hipHostMalloc(&p, piece));
        for (auto &s : streams) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int rep = 0; rep < 3; rep++) {
            char *fresh = static_cast<char *>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
            double t0 = now();
            std::atomic<size_t> next{0};
            const size_t chunks = bytes / piece;
            auto work = [&](unsigned t) {
                CK(hipSetDevice(0));
                size_t mine[2] = {~size_t(0), ~size_t(0)}; unsigned b = 0;
                for (size_t c = next++; ; c = next++) {
                    if (c < chunks) { CK(hipMemcpyAsync(pinned[2 * t + b], static_cast<char *>(src) + c * piece, piece, hipMemcpyDeviceToHost, streams[t])); mine[b] = c; }
                    b ^= 1;
                    if (mine[b] != ~size_t(0)) { /* the older one */ CK(hipStreamSynchronize(streams[t])); memcpy(fresh + mine[b] * piece, pinned[2 * t + b], piece); mine[b] = ~size_t(0); }
                    if (c >= chunks) { b ^= 1; if (mine[b] != ~size_t(0)) { CK(hipStreamSynchronize(streams[t])); memcpy(fresh + mine[b] * piece, pinned[2 * t + b], piece); } break; }
                }
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < threads; t++) pool.emplace_back(work, t);
            work(0);
            for (auto &t : pool) t.join();
            double a = now() - t0;
            printf("D2H 48 MB staged through %u threads x 2 x 2 MB pinned into fresh memory: %.2f ms (%.1f GB/s)\n", threads, a, bytes / a / 1e6);
            munmap(fresh, bytes);
        }
        for (auto &p : pinned) CK(hipHostFree(p));
        for (auto &s : streams) CK(hipStreamDestroy(s));
    }
    return 0;
}
