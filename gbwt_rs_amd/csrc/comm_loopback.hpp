// comm_loopback.hpp -- TEST INFRASTRUCTURE, not part of libgbwt_hip.so: a stand-in for RCCL whose ranks are threads of one process on
// one GPU.  comm.hip includes this file only when it is compiled with -DGBWT_HIP_TEST_TRANSPORT, i.e. into libgbwt_hip_testtransport.so
// (Makefile: the product objects + this), which tests/test_gpu_dist.py loads through GBWT_HIP_LIB.  The product library holds none of it.
// (included inside comm.hip's unnamed namespace, behind its own #include <condition_variable> / <deque> / <map>)
#pragma once

// ---- loopback transport (GBWT_HIP_COMM_LOOPBACK=1): test infrastructure --------------------------------------------------------------
// The ranks are THREADS of one process that share one GPU, and the nine RCCL entry points above are served by the functions below:
// an all-gather and point-to-point groups with RCCL's matching rules (sends and receives of a pair of ranks pair up in order), moved
// by device-to-device copies.  No box of rounds 1-4 had a second GPU, and RCCL refuses two ranks on one device: this is how everything
// in this file EXCEPT RCCL itself -- the counts, the buffers, the three placements on the root for world sizes above one -- runs under
// `pytest -m gpu` (tests/test_gpu_dist.py).  Operations complete before the call returns (RCCL's complete on the stream).
namespace loopback {

struct Post { const void *src; size_t bytes; bool taken; };
struct World {
    int world = 0, joined = 0, alive = 0;                      // alive: handles not yet destroyed
    std::mutex m;
    std::condition_variable cv;
    std::vector<const void *> gather_src;
    int arrived = 0, copied = 0;
    uint64_t round = 0;
    std::map<std::pair<int, int>, std::deque<Post *>> mail;       // (from, to): sends waiting for their receive, in order
};
struct Handle { std::shared_ptr<World> w; int rank; };
struct Op { bool send; const void *src; void *dst; size_t bytes; int peer; Handle *h; hipStream_t stream; };

std::mutex registry_mutex;
std::map<std::string, std::shared_ptr<World>> registry;
thread_local int group_depth = 0;
thread_local std::vector<Op> group_ops;

// every wait of the transport ends: a rank that never comes (it failed, or was given another id) is an error after two minutes, not a hang
template <class Pred>
bool wait_for(World &w, std::unique_lock<std::mutex> &lock, Pred pred) { return w.cv.wait_for(lock, std::chrono::seconds(120), pred); }

size_t type_bytes(ncclDataType_t t) { return t == ncclUint64 || t == ncclInt64 || t == ncclFloat64 ? 8 : (t == ncclUint32 || t == ncclInt32 || t == ncclFloat32 ? 4 : 1); }

ncclResult_t GetUniqueId(ncclUniqueId *id) {
    static std::atomic<uint64_t> next{1};
    std::memset(id->internal, 0, sizeof(id->internal));
    std::snprintf(id->internal, sizeof(id->internal), "gbwt_hip loopback %llu %p", static_cast<unsigned long long>(next++), static_cast<void *>(&next));
    return ncclSuccess;
}

ncclResult_t CommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lock(registry_mutex);
        std::shared_ptr<World> &slot = registry[std::string(id.internal, sizeof(id.internal))];
        if (!slot) { slot = std::make_shared<World>(); slot->world = world; slot->gather_src.assign(world, nullptr); }
        w = slot;
    }
    if (w->world != world || rank < 0 || rank >= world) return ncclInvalidArgument;
    std::unique_lock<std::mutex> lock(w->m);
    w->joined++; w->alive++;
    if (std::getenv("GBWT_HIP_COMM_TRACE")) std::fprintf(stderr, "[loopback] rank %d of %d joined (%d so far) world %p\n", rank, world, w->joined, static_cast<void *>(w.get()));
    w->cv.notify_all();
    if (!wait_for(*w, lock, [&]() { return w->joined >= world; })) return ncclInternalError;
    *out = reinterpret_cast<ncclComm_t>(new Handle{w, rank});
    return ncclSuccess;
}

// the world leaves the registry with its last handle
ncclResult_t CommDestroy(ncclComm_t comm) {
    Handle *h = reinterpret_cast<Handle *>(comm);
    {
        std::lock_guard<std::mutex> lock(registry_mutex);
        bool last = false;
        { std::lock_guard<std::mutex> inner(h->w->m); last = --h->w->alive == 0; }
        if (last) for (auto it = registry.begin(); it != registry.end(); ++it) if (it->second == h->w) { registry.erase(it); break; }
    }
    delete h;
    return ncclSuccess;
}

ncclResult_t AllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
    Handle *h = reinterpret_cast<Handle *>(comm);
    World &w = *h->w;
    const size_t bytes = count * type_bytes(type);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;        // what this rank contributes is written
    {
        std::unique_lock<std::mutex> lock(w.m);
        w.gather_src[h->rank] = send;
        w.arrived++;
        w.cv.notify_all();
        if (!wait_for(w, lock, [&]() { return w.arrived >= w.world; })) return ncclInternalError;
    }
    for (int r = 0; r < w.world; r++)
        if (hipMemcpyAsync(static_cast<char *>(recv) + r * bytes, w.gather_src[r], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lock(w.m);
    const uint64_t round = w.round;
    if (++w.copied == w.world) { w.arrived = 0; w.copied = 0; w.round++; w.cv.notify_all(); }
    else if (!wait_for(w, lock, [&]() { return w.round != round; })) return ncclInternalError;   // nobody's buffer changes while somebody still reads it
    return ncclSuccess;
}

ncclResult_t run(std::vector<Op> &ops) {
    std::vector<Post *> posted;
    for (const Op &op : ops) if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (const Op &op : ops) {
        if (!op.send) continue;
        World &w = *op.h->w;
        Post *p = new Post{op.src, op.bytes, false};
        posted.push_back(p);
        std::lock_guard<std::mutex> lock(w.m);
        w.mail[{op.h->rank, op.peer}].push_back(p);
        w.cv.notify_all();
    }
    ncclResult_t result = ncclSuccess;
    for (const Op &op : ops) {
        if (op.send) continue;
        World &w = *op.h->w;
        Post *p = nullptr;
        {
            std::unique_lock<std::mutex> lock(w.m);
            std::deque<Post *> &box = w.mail[{op.peer, op.h->rank}];
            if (!wait_for(w, lock, [&]() { return !box.empty(); })) { result = ncclInternalError; continue; }
            p = box.front();
            box.pop_front();
        }
        if (p->bytes != op.bytes) result = ncclInvalidArgument;                              // RCCL: a send and its receive have one size
        else if (hipMemcpyAsync(op.dst, p->src, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess || hipStreamSynchronize(op.stream) != hipSuccess) result = ncclUnhandledCudaError;
        std::lock_guard<std::mutex> lock(w.m);
        p->taken = true;
        w.cv.notify_all();
    }
    for (size_t i = 0, k = 0; i < ops.size(); i++) {
        if (!ops[i].send) continue;
        World &w = *ops[i].h->w;
        Post *p = posted[k++];
        std::unique_lock<std::mutex> lock(w.m);
        if (wait_for(w, lock, [&]() { return p->taken; })) delete p;
        else result = ncclInternalError;                                                     // (the post stays in its box: leaked, not dangling)
    }
    ops.clear();
    return result;
}

ncclResult_t GroupStart() { group_depth++; return ncclSuccess; }
ncclResult_t GroupEnd() { return --group_depth == 0 ? run(group_ops) : ncclSuccess; }
ncclResult_t Send(const void *src, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    group_ops.push_back(Op{true, src, nullptr, count * type_bytes(type), peer, reinterpret_cast<Handle *>(comm), stream});
    return group_depth == 0 ? run(group_ops) : ncclSuccess;
}
ncclResult_t Recv(void *dst, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    group_ops.push_back(Op{false, nullptr, dst, count * type_bytes(type), peer, reinterpret_cast<Handle *>(comm), stream});
    return group_depth == 0 ? run(group_ops) : ncclSuccess;
}
const char *GetErrorString(ncclResult_t e) {
    return e == ncclInvalidArgument ? "loopback: invalid argument (sizes of a send and its receive differ?)"
         : e == ncclInternalError ? "loopback: a rank did not arrive within two minutes (did every rank get the same unique id?)" : "loopback: HIP error";
}

}  // namespace loopback
