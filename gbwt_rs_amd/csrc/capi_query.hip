// capi_query.hip -- the C ABI of libgbwt_hip.so (include/gbwt_hip.h), part 3 of 3: the one-lane-per-query entry points -- start / forward /
// backward / find / extend / bd_* / follow / search (src/gbwt.rs:213-384, src/gbz.rs:519-544, src/bin/benchmark.rs:124-169).  No CPU
// implementation of any compute entry point.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "capi_internal.hpp"

using namespace gbwt_hip;

namespace {

// Staging helper for the one-lane-per-query entry points.  A query is a ROW: a_row bytes of `in_a` (+ b_row bytes of `in_b`) in,
// out_row bytes of `out` + one byte of `valid` out; launch(d_a, d_b, d_out, d_valid, rows, stream) runs the kernel over `rows` rows.
// Copy in, one launch, copy out, on the workspace stream (a million 10-node queries: 2.9-3.3 ms per call around a 0.49 ms kernel -- 105 MB over
// PCIe; src/bin/benchmark.rs:161-164 times the whole call, not the kernel).  The pageable copies are the fastest way this driver offers between
// caller-owned host memory and the device (profiles/r06_download_probe.txt: 1 ms per 48 MB into touched memory, 2.4 ms into fresh pages; staging
// through pinned buffers on several threads -- round 5's GBWT_HIP_QUERY_PIPELINE, removed -- 4-6 ms).  The device-resident forms below skip the
// copies altogether.
template <class Launch>
gbwt_hip_status run_query(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const void *in_a, size_t a_row, const void *in_b, size_t b_row,
                          void *out, size_t out_row, uint8_t *valid, uint64_t n, Launch launch) {
    if (!ix || !ws || ws->index != ix) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace");
    if (!(ix->caps & GBWT_HIP_OPEN_SEARCH)) return fail(GBWT_HIP_BAD_ARGUMENT, "the handle was not opened for navigation / search (GBWT_HIP_OPEN_SEARCH)");
    if (n == 0) return GBWT_HIP_OK;
    if ((!in_a && a_row != 0) || !out || !valid) return fail(GBWT_HIP_BAD_ARGUMENT, "null buffer");   // (rows of no bytes -- queries of no nodes -- have no buffer)
    ws->follow_cached = false;   // the staging buffers are shared with gbwt_hip_follow
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        ws->in_a.reserve(std::max<size_t>(n * a_row, 16));
        ws->out_a.reserve(n * out_row);
        ws->out_valid.reserve(n);
        if (in_b) ws->in_b.reserve(n * b_row);
        for (auto &e : ws->qev) if (!e) HIP_CHECK(hipEventCreate(&e));
        hipStream_t s = ws->stream;
        if (a_row != 0) HIP_CHECK(hipMemcpyAsync(ws->in_a.ptr, in_a, n * a_row, hipMemcpyHostToDevice, s));
        if (in_b) HIP_CHECK(hipMemcpyAsync(ws->in_b.ptr, in_b, n * b_row, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipEventRecord(ws->qev[0], s));
        launch(ws->in_a.as<char>(), in_b ? ws->in_b.as<char>() : nullptr, ws->out_a.as<char>(), ws->out_valid.as<uint8_t>(), n, s);
        HIP_CHECK(hipEventRecord(ws->qev[1], s));
        ws->query_timed = true;
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out, ws->out_a.ptr, n * out_row, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(valid, ws->out_valid.ptr, n, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
}

// Device-resident form: the rows of `d_in` are in HBM already (the caller's buffer, read on the workspace stream), the results stay
// in the workspace.
template <class Launch>
gbwt_hip_status run_query_device(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const void *d_in, size_t out_row, uint64_t n, Launch launch) {
    if (!ix || !ws || ws->index != ix) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace");
    if (!(ix->caps & GBWT_HIP_OPEN_SEARCH)) return fail(GBWT_HIP_BAD_ARGUMENT, "the handle was not opened for navigation / search (GBWT_HIP_OPEN_SEARCH)");
    if (n && !d_in) return fail(GBWT_HIP_BAD_ARGUMENT, "null buffer");
    ws->follow_cached = false;
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        ws->out_a.reserve(std::max<uint64_t>(n, 1) * out_row);
        ws->out_valid.reserve(std::max<uint64_t>(n, 1));
        for (auto &e : ws->qev) if (!e) HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventRecord(ws->qev[0], ws->stream));
        if (n) launch(ws->out_a.as<char>(), ws->out_valid.as<uint8_t>(), ws->stream);
        HIP_CHECK(hipEventRecord(ws->qev[1], ws->stream));
        ws->query_timed = true;
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(ws->stream));
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
}

}  // namespace

extern "C" {

gbwt_hip_status gbwt_hip_last_query_ms(const gbwt_hip_workspace *ws, float *kernel_ms) {
    GBWT_HIP_GUARD_BEGIN
    if (!ws || !ws->query_timed || !kernel_ms) return fail(GBWT_HIP_BAD_ARGUMENT, "no timed query on this workspace");
    if (hipEventElapsedTime(kernel_ms, ws->qev[0], ws->qev[1]) != hipSuccess) return fail(GBWT_HIP_DEVICE_ERROR, "hipEventElapsedTime failed");
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_start(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *seq_ids, uint64_t n,
                               gbwt_hip_pos *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    return run_query(ix, ws, seq_ids, sizeof(uint64_t), nullptr, 0, out, sizeof(gbwt_hip_pos), valid, n, [&](const void *a, const void *, void *o, uint8_t *v, uint64_t rows, hipStream_t s) {
        launch_start(ix->dev, static_cast<const uint64_t *>(a), rows, static_cast<gbwt_hip_pos *>(o), v, s);
    });
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_forward(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const gbwt_hip_pos *in, uint64_t n,
                                 gbwt_hip_pos *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    return run_query(ix, ws, in, sizeof(gbwt_hip_pos), nullptr, 0, out, sizeof(gbwt_hip_pos), valid, n, [&](const void *a, const void *, void *o, uint8_t *v, uint64_t rows, hipStream_t s) {
        launch_forward(ix->dev, static_cast<const gbwt_hip_pos *>(a), rows, static_cast<gbwt_hip_pos *>(o), v, s);
    });
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_backward(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const gbwt_hip_pos *in, uint64_t n,
                                  gbwt_hip_pos *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    if (ix && !ix->host.bidirectional) return fail(GBWT_HIP_BAD_ARGUMENT, "Following sequences backward requires a bidirectional GBWT");
    return run_query(ix, ws, in, sizeof(gbwt_hip_pos), nullptr, 0, out, sizeof(gbwt_hip_pos), valid, n, [&](const void *a, const void *, void *o, uint8_t *v, uint64_t rows, hipStream_t s) {
        launch_backward(ix->dev, static_cast<const gbwt_hip_pos *>(a), rows, static_cast<gbwt_hip_pos *>(o), v, s);
    });
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_find(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *nodes, uint64_t n,
                              gbwt_hip_state *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    return run_query(ix, ws, nodes, sizeof(uint64_t), nullptr, 0, out, sizeof(gbwt_hip_state), valid, n, [&](const void *a, const void *, void *o, uint8_t *v, uint64_t rows, hipStream_t s) {
        launch_find(ix->dev, static_cast<const uint64_t *>(a), rows, static_cast<gbwt_hip_state *>(o), v, s);
    });
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_extend(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const gbwt_hip_state *states,
                                const uint64_t *nodes, uint64_t n, gbwt_hip_state *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    if (n && !nodes) return fail(GBWT_HIP_BAD_ARGUMENT, "null nodes");
    return run_query(ix, ws, states, sizeof(gbwt_hip_state), nodes, sizeof(uint64_t), out, sizeof(gbwt_hip_state), valid, n, [&](const void *a, const void *b, void *o, uint8_t *v, uint64_t rows, hipStream_t s) {
        launch_extend(ix->dev, static_cast<const gbwt_hip_state *>(a), static_cast<const uint64_t *>(b), rows, static_cast<gbwt_hip_state *>(o), v, s);
    });
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_bd_find(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *nodes, uint64_t n,
                                 gbwt_hip_bd_state *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    // the reference asserts here (src/gbwt.rs:312)
    if (ix && !ix->host.bidirectional) return fail(GBWT_HIP_BAD_ARGUMENT, "Bidirectional search requires a bidirectional GBWT");
    return run_query(ix, ws, nodes, sizeof(uint64_t), nullptr, 0, out, sizeof(gbwt_hip_bd_state), valid, n, [&](const void *a, const void *, void *o, uint8_t *v, uint64_t rows, hipStream_t s) {
        launch_bd_find(ix->dev, static_cast<const uint64_t *>(a), rows, static_cast<gbwt_hip_bd_state *>(o), v, s);
    });
    GBWT_HIP_GUARD_END
}

static gbwt_hip_status bd_extend(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const gbwt_hip_bd_state *states,
                                 const uint64_t *nodes, uint64_t n, bool backward, gbwt_hip_bd_state *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    if (ix && !ix->host.bidirectional) return fail(GBWT_HIP_BAD_ARGUMENT, "Bidirectional search requires a bidirectional GBWT");
    if (n && !nodes) return fail(GBWT_HIP_BAD_ARGUMENT, "null nodes");
    return run_query(ix, ws, states, sizeof(gbwt_hip_bd_state), nodes, sizeof(uint64_t), out, sizeof(gbwt_hip_bd_state), valid, n, [&](const void *a, const void *b, void *o, uint8_t *v, uint64_t rows, hipStream_t s) {
        launch_bd_extend(ix->dev, static_cast<const gbwt_hip_bd_state *>(a), static_cast<const uint64_t *>(b), rows, backward, static_cast<gbwt_hip_bd_state *>(o), v, s);
    });
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_extend_forward(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const gbwt_hip_bd_state *states,
                                        const uint64_t *nodes, uint64_t n, gbwt_hip_bd_state *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    return bd_extend(ix, ws, states, nodes, n, false, out, valid);
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_extend_backward(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const gbwt_hip_bd_state *states,
                                         const uint64_t *nodes, uint64_t n, gbwt_hip_bd_state *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    return bd_extend(ix, ws, states, nodes, n, true, out, valid);
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_follow(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const gbwt_hip_bd_state *states, uint64_t n, int backward,
                                uint64_t *out_offsets, gbwt_hip_bd_state *out_states, uint64_t capacity, uint64_t *total, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    if (!ix || !ws || ws->index != ix) return fail(GBWT_HIP_BAD_ARGUMENT, "null or mismatched index / workspace");
    if (!ix->host.bidirectional) return fail(GBWT_HIP_BAD_ARGUMENT, "Bidirectional search requires a bidirectional GBWT");
    if (!(ix->caps & GBWT_HIP_OPEN_SEARCH)) return fail(GBWT_HIP_BAD_ARGUMENT, "the handle was not opened for navigation / search (GBWT_HIP_OPEN_SEARCH)");
    if (!total || !out_offsets || (n && (!states || !valid))) return fail(GBWT_HIP_BAD_ARGUMENT, "null buffer");
    *total = 0;
    out_offsets[0] = 0;
    if (n == 0) return GBWT_HIP_OK;
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        hipStream_t s = ws->stream;
        // the fill call after a size query with the same states: counts and offsets are still in the workspace
        const size_t key_bytes = n * sizeof(gbwt_hip_bd_state);
        const bool hit = ws->follow_cached && ws->follow_backward == (backward != 0) && ws->follow_key.size() == key_bytes &&
                         std::memcmp(ws->follow_key.data(), states, key_bytes) == 0;
        if (!hit) {
            ws->follow_cached = false;
            ws->in_a.reserve(key_bytes);
            ws->in_b.reserve(n * sizeof(uint64_t));
            ws->out_valid.reserve(n);
            ws->follow_off.reserve((n + 1) * sizeof(uint64_t));
            const size_t temp_bytes = scan_temp_bytes(n);
            ws->scan_temp.reserve(std::max<size_t>(temp_bytes, 16));
            HIP_CHECK(hipMemcpyAsync(ws->in_a.ptr, states, key_bytes, hipMemcpyHostToDevice, s));
            launch_follow_count(ix->dev, ws->in_a.as<gbwt_hip_bd_state>(), n, backward != 0, ws->in_b.as<uint64_t>(), ws->out_valid.as<uint8_t>(), s);
            launch_scan(ws->in_b.as<uint64_t>(), ws->follow_off.as<uint64_t>(), n, ws->scan_temp.ptr, temp_bytes, s);
            HIP_CHECK(hipGetLastError());
            ws->follow_key.assign(reinterpret_cast<const uint8_t *>(states), reinterpret_cast<const uint8_t *>(states) + key_bytes);
            ws->follow_backward = backward != 0;
        }
        HIP_CHECK(hipMemcpyAsync(out_offsets, ws->follow_off.ptr, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(valid, ws->out_valid.ptr, n, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        *total = out_offsets[n];
        ws->follow_cached = true;
        if (!out_states) return GBWT_HIP_OK;
        if (capacity < *total) return fail(GBWT_HIP_CAPACITY, "output capacity too small for the extensions");
        if (*total == 0) return GBWT_HIP_OK;
        ws->out_a.reserve(*total * sizeof(gbwt_hip_bd_state));
        launch_follow_fill(ix->dev, ws->in_a.as<gbwt_hip_bd_state>(), n, backward != 0, ws->follow_off.as<uint64_t>(), ws->out_a.as<gbwt_hip_bd_state>(), s);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out_states, ws->out_a.ptr, *total * sizeof(gbwt_hip_bd_state), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return status_of(e);
    }
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_search(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *queries, uint64_t n,
                                uint64_t len, gbwt_hip_state *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    if (len > (uint64_t(1) << 32)) return fail(GBWT_HIP_BAD_ARGUMENT, "query length out of range");
    return run_query(ix, ws, queries, len * sizeof(uint64_t), nullptr, 0, out, sizeof(gbwt_hip_state), valid, n, [&](const void *a, const void *, void *o, uint8_t *v, uint64_t rows, hipStream_t s) {
        launch_search(ix->dev, static_cast<const uint64_t *>(a), rows, len, static_cast<gbwt_hip_state *>(o), v, s);
    });
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_search_device(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *d_queries, uint64_t n, uint64_t len,
                                       gbwt_hip_states *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!out) return fail(GBWT_HIP_BAD_ARGUMENT, "null output");
    *out = gbwt_hip_states{nullptr, nullptr, 0};
    if (len > (uint64_t(1) << 32)) return fail(GBWT_HIP_BAD_ARGUMENT, "query length out of range");
    const gbwt_hip_status st = run_query_device(ix, ws, d_queries, sizeof(gbwt_hip_state), n, [&](void *o, uint8_t *v, hipStream_t s) {
        launch_search(ix->dev, d_queries, n, len, static_cast<gbwt_hip_state *>(o), v, s);
    });
    if (st == GBWT_HIP_OK) *out = gbwt_hip_states{ws->out_a.as<gbwt_hip_state>(), ws->out_valid.as<uint8_t>(), n};
    return st;
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_bd_search(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *queries, uint64_t n,
                                   uint64_t len, uint64_t first, gbwt_hip_bd_state *out, uint8_t *valid) {
    GBWT_HIP_GUARD_BEGIN
    if (ix && !ix->host.bidirectional) return fail(GBWT_HIP_BAD_ARGUMENT, "Bidirectional search requires a bidirectional GBWT");
    if (len > (uint64_t(1) << 32)) return fail(GBWT_HIP_BAD_ARGUMENT, "query length out of range");
    return run_query(ix, ws, queries, len * sizeof(uint64_t), nullptr, 0, out, sizeof(gbwt_hip_bd_state), valid, n, [&](const void *a, const void *, void *o, uint8_t *v, uint64_t rows, hipStream_t s) {
        launch_bd_search(ix->dev, static_cast<const uint64_t *>(a), rows, len, first, static_cast<gbwt_hip_bd_state *>(o), v, s);
    });
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_bd_search_device(const gbwt_hip_index *ix, gbwt_hip_workspace *ws, const uint64_t *d_queries, uint64_t n, uint64_t len,
                                          uint64_t first, gbwt_hip_bd_states *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!out) return fail(GBWT_HIP_BAD_ARGUMENT, "null output");
    *out = gbwt_hip_bd_states{nullptr, nullptr, 0};
    if (ix && !ix->host.bidirectional) return fail(GBWT_HIP_BAD_ARGUMENT, "Bidirectional search requires a bidirectional GBWT");
    if (len > (uint64_t(1) << 32)) return fail(GBWT_HIP_BAD_ARGUMENT, "query length out of range");
    const gbwt_hip_status st = run_query_device(ix, ws, d_queries, sizeof(gbwt_hip_bd_state), n, [&](void *o, uint8_t *v, hipStream_t s) {
        launch_bd_search(ix->dev, d_queries, n, len, first, static_cast<gbwt_hip_bd_state *>(o), v, s);
    });
    if (st == GBWT_HIP_OK) *out = gbwt_hip_bd_states{ws->out_a.as<gbwt_hip_bd_state>(), ws->out_valid.as<uint8_t>(), n};
    return st;
    GBWT_HIP_GUARD_END
}

}  // extern "C"
