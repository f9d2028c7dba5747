// microbench_upload.hip -- what an open can do about its host-to-device copy of the record bytes (round 6; tools/r06_upload_probe.sh):
// pageable hipMemcpy of 60 MB (from malloc'd memory, from a populated file mapping), the cost of hipHostMalloc by size, and a staged
// copy through small pinned buffers filled by T threads.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double staged(void *dst, const char *src, size_t bytes, unsigned threads, size_t piece, std::vector<void *> &pinned, std::vector<hipStream_t> &streams, std::vector<hipEvent_t> &events) {
    const double t0 = now();
    const size_t chunks = (bytes + piece - 1) / piece;
    std::atomic<size_t> next{0};
    auto work = [&](unsigned t) {
        CK(hipSetDevice(0));
        unsigned b = 0;
        bool used[2] = {false, false};
        for (size_t c = next++; c < chunks; c = next++) {
            const size_t lo = c * piece, n = std::min(piece, bytes - lo);
            if (used[b]) CK(hipEventSynchronize(events[2 * t + b]));
            memcpy(pinned[2 * t + b], src + lo, n);
            CK(hipMemcpyAsync(static_cast<char *>(dst) + lo, pinned[2 * t + b], n, hipMemcpyHostToDevice, streams[t]));
            CK(hipEventRecord(events[2 * t + b], streams[t]));
            used[b] = true; b ^= 1;
        }
        CK(hipStreamSynchronize(streams[t]));
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &t : pool) t.join();
    return now() - t0;
}

int main() {
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    const size_t bytes = 60u << 20;
    void *dst; CK(hipMalloc(&dst, bytes));
    char *heap = static_cast<char *>(malloc(bytes)); memset(heap, 7, bytes);
    // a file in /dev/shm, mapped and populated like the loader's
    const char *path = "/dev/shm/gbwt_upload_probe.bin";
    { int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, 0600); size_t w = 0; while (w < bytes) w += write(fd, heap + w, bytes - w); close(fd); }
    int fd = open(path, O_RDONLY);
    char *map = static_cast<char *>(mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0));
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); CK(hipMemcpy(dst, heap, bytes, hipMemcpyHostToDevice)); double a = now() - t0;
        t0 = now(); CK(hipMemcpy(dst, map, bytes, hipMemcpyHostToDevice)); double b = now() - t0;
        printf("pageable hipMemcpy 60 MB: heap %.2f ms (%.1f GB/s), mapping %.2f ms (%.1f GB/s)\n", a, bytes / a / 1e6, b, bytes / b / 1e6);
    }
    for (unsigned slices : {2u, 4u}) {
        double t0 = now();
        std::vector<std::thread> pool;
        for (unsigned k = 0; k < slices; k++) pool.emplace_back([&, k]() { CK(hipSetDevice(0)); const size_t lo = bytes / slices * k, hi = k + 1 == slices ? bytes : bytes / slices * (k + 1); CK(hipMemcpy(static_cast<char *>(dst) + lo, map + lo, hi - lo, hipMemcpyHostToDevice)); });
        for (auto &t : pool) t.join();
        double a = now() - t0;
        printf("pageable hipMemcpy 60 MB in %u slices on %u threads: %.2f ms (%.1f GB/s)\n", slices, slices, a, bytes / a / 1e6);
    }
    for (size_t mb : {1u, 2u, 4u, 16u, 64u}) {
        void *p; double t0 = now(); CK(hipHostMalloc(&p, mb << 20)); double a = now() - t0;
        t0 = now(); CK(hipHostFree(p)); double b = now() - t0;
        void *q; t0 = now(); CK(hipHostMalloc(&q, mb << 20)); double c = now() - t0; CK(hipHostFree(q));
        printf("hipHostMalloc %3zu MB: %.2f ms (again %.2f ms), free %.2f ms\n", mb, a, c, b);
    }
    for (size_t mb : {16u, 60u}) {
        void *p = malloc(mb << 20); memset(p, 1, mb << 20);
        double t0 = now(); CK(hipHostRegister(p, mb << 20, hipHostRegisterDefault)); double a = now() - t0;
        t0 = now(); CK(hipMemcpy(dst, p, std::min<size_t>(bytes, mb << 20), hipMemcpyHostToDevice)); double c = now() - t0;
        t0 = now(); CK(hipHostUnregister(p)); double b = now() - t0;
        printf("hipHostRegister %3zu MB: %.2f ms, copy %.2f ms, unregister %.2f ms\n", mb, a, c, b);
        free(p);
    }
    for (unsigned threads : {1u, 2u, 4u}) for (size_t piece : {size_t(512) << 10, size_t(1) << 20, size_t(4) << 20}) {
        std::vector<void *> pinned(2 * threads); std::vector<hipStream_t> streams(threads); std::vector<hipEvent_t> events(2 * threads);
        double t0 = now();
        for (auto &p : pinned) CK(hipHostMalloc(&p, piece));
        for (auto &s : streams) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto &e : events) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        double setup = now() - t0;
        double first = staged(dst, map, bytes, threads, piece, pinned, streams, events), second = staged(dst, map, bytes, threads, piece, pinned, streams, events);
        printf("staged 60 MB, %u threads x 2 x %zu KB pinned: setup %.2f ms, copy %.2f ms (%.1f GB/s), again %.2f ms (%.1f GB/s)\n", threads, piece >> 10, setup, first, bytes / first / 1e6, second, bytes / second / 1e6);
        for (auto &p : pinned) CK(hipHostFree(p));
        for (auto &s : streams) CK(hipStreamDestroy(s));
        for (auto &e : events) CK(hipEventDestroy(e));
    }
    munmap(map, bytes); close(fd); unlink(path);
    return 0;
}
