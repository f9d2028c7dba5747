// comm.hip -- the one exchange of a sharded extraction, behind the C ABI (include/gbwt_hip.h: gbwt_hip_comm_*).
//
// The reference's parallel axis is the path (rayon over path ids, src/bin/gbunzip.rs:27, 421-434); its only shared state is the writer
// behind a mutex, which puts the finished lines into one file.  Here the paths are dealt to one process per GPU, every rank walks and
// formats its own shard against its own replica of the index (no collective in the walk), and this file is the writer's mutex: the rows
// of all ranks -- node ids or finished GFA lines -- gathered on one rank in path order, over RCCL:
//
//   1. ncclAllGather of {rows, payload bytes} per rank (16 B each), one host wait: the receive buffers are sized from it;
//   2. ONE group of point-to-point operations (ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd): every peer streams its row
//      lengths and its payload to the root at once, each over its own xGMI link (a ring all-gather would be bound by one link);
//   3. on the root, kernels put the rows into path order: contiguous shards arrive in place; interleaved shards (row k of rank r is
//      global row k * world + r -- SURVEY 8e: path lengths correlate with neighbouring ids) are scattered row by row.
//
// RCCL is loaded at run time (dlopen of librccl.so.1: the library already in the process when the caller is a torch program, ROCm's
// own otherwise), so libgbwt_hip.so loads and extracts on a box without it; the comm entry points then return GBWT_HIP_UNSUPPORTED.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#ifdef GBWT_HIP_TEST_TRANSPORT
#include <condition_variable>
#include <deque>
#include <map>
#endif
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "capi_internal.hpp"

using namespace gbwt_hip;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

#ifdef GBWT_HIP_TEST_TRANSPORT
#include "comm_loopback.hpp"   // the test build only (libgbwt_hip_testtransport.so): ranks = threads of one process on one GPU
#endif

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
#ifdef GBWT_HIP_TEST_TRANSPORT
        if (const char *v = std::getenv("GBWT_HIP_COMM_LOOPBACK"); v && std::atoi(v) != 0) {
            r.GetUniqueId = loopback::GetUniqueId; r.CommInitRank = loopback::CommInitRank; r.CommDestroy = loopback::CommDestroy;
            r.GroupStart = loopback::GroupStart; r.GroupEnd = loopback::GroupEnd; r.Send = loopback::Send; r.Recv = loopback::Recv;
            r.AllGather = loopback::AllGather; r.GetErrorString = loopback::GetErrorString;
            return;
        }
#endif
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // the copy the process already has (torch's)
        for (const char *n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) for (const char *n : {"/opt/rocm/lib/librccl.so.1"}) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) { r.why = std::string("RCCL is not available: ") + dlerror(); return; }
        auto sym = [&](const char *name) { void *p = dlsym(r.handle, name); if (!p && r.why.empty()) r.why = std::string("RCCL lacks ") + name; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

struct RcclError { ncclResult_t err; const char *what; };
#define RCCL_CHECK(expr)                                         \
    do {                                                         \
        ncclResult_t e_ = (expr);                                \
        if (e_ != ncclSuccess) throw RcclError{e_, #expr};       \
    } while (0)

// ---- placement kernels (root) ---------------------------------------------------------------------------------------------------
// lengths of all rows in path order: row k of rank r is global row k * world + r (interleaved) -- part_first[r] = first entry of rank r in
// the concatenated per-rank lengths
__global__ void __launch_bounds__(256) k_interleave_lengths(const uint64_t *part_len, const uint64_t *part_first, uint32_t world, uint64_t total_rows, uint64_t *all_len) {
    const uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (g >= total_rows) return;
    const uint32_t r = static_cast<uint32_t>(g % world);
    const uint64_t k = g / world;
    all_len[g] = part_len[part_first[r] + k];
}

constexpr uint64_t SCATTER_GRID = 32768;   // workgroups (x slices) of k_scatter_rows: enough to fill the chip; more rows than that take turns

// Row g of the result <- row k of the part of rank r, as bytes.  One workgroup per (row, slice).  The destination is written in whole
// dwords (the bytes in front of its first and behind its last dword one by one); the source may start at any byte (GFA lines), so a
// dword is put together from the two aligned dwords it straddles (v_alignbyte_b32).  Every part is followed by 16 bytes of slack, so
// the dword behind the last byte of a part may be read.
__global__ void __launch_bounds__(256) k_scatter_rows(const uint8_t *parts, const uint64_t *part_byte_first, const uint64_t *part_row_start /* per part: exclusive scan of its lengths (rows + 1 entries), concatenated */,
                                                       const uint64_t *part_first /* first entry of every part in part_row_start */, uint32_t world, uint64_t total_rows,
                                                       const uint64_t *out_offsets, uint32_t unit, uint8_t *out, uint32_t slices) {
  // (a grid-stride loop over the rows: gridDim.x * blockDim.x must stay below 2^32, i.e. 2^24 workgroups of 256, far below the 2^31 rows a gather takes)
  for (uint64_t g = blockIdx.x; g < total_rows; g += gridDim.x) {
    const uint32_t r = static_cast<uint32_t>(g % world), t = threadIdx.x;
    const uint64_t k = g / world;
    const uint64_t bytes = (out_offsets[g + 1] - out_offsets[g]) * unit;
    if (bytes == 0) continue;
    const uint8_t *src = parts + part_byte_first[r] + part_row_start[part_first[r] + k] * unit;
    uint8_t *dst = out + out_offsets[g] * unit;
    const uintptr_t d0 = reinterpret_cast<uintptr_t>(dst), d1 = d0 + bytes;
    uintptr_t a0 = (d0 + 3) & ~uintptr_t(3), a1 = d1 & ~uintptr_t(3);      // the whole dwords of the destination: [a0, a1)
    uint64_t head = a0 - d0, tail = d1 - a1;
    if (a0 >= a1) { head = bytes; tail = 0; a0 = a1 = d1; }                // at most six bytes: no whole dword
    if (blockIdx.y == 0 && t < head) dst[t] = src[t];
    if (blockIdx.y + 1 == slices && t < tail) dst[bytes - tail + t] = src[bytes - tail + t];
    const uint64_t words = (a1 - a0) / 4, w_lo = words * blockIdx.y / slices, w_hi = words * (blockIdx.y + 1) / slices;
    const uint8_t *s = src + head;                                          // the source byte that goes into the first whole dword
    const uint32_t shift = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s) & 3u);
    const uint32_t *sa = reinterpret_cast<const uint32_t *>(s - shift);
    uint32_t *da = reinterpret_cast<uint32_t *>(a0);
    if (shift == 0) {
        for (uint64_t i = w_lo + t; i < w_hi; i += 256) __builtin_nontemporal_store(sa[i], da + i);
    } else {
        for (uint64_t i = w_lo + t; i < w_hi; i += 256) __builtin_nontemporal_store(__builtin_amdgcn_alignbyte(sa[i + 1], sa[i], shift), da + i);
    }
  }
}

// parts of rows (GBWT_HIP_GATHER_PARTS): the interleaved placement has made world "rows" of every row, one per rank, back to back:
// the row offsets are every world-th of theirs
__global__ void __launch_bounds__(256) k_every_nth_offset(const uint64_t *offsets, uint32_t world, uint64_t rows, uint64_t *row_offsets) {
    const uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (g <= rows) row_offsets[g] = offsets[g * world];
}

}  // namespace

struct gbwt_hip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    DeviceBuffer counts, part_len, part_start, parts, all_len, offsets, row_offsets, out, scan_temp, meta, staged;
    gbwt_hip_comm_stats last{};
    ~gbwt_hip_comm() {
        if (comm && rccl().CommDestroy) (void)rccl().CommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

gbwt_hip_status status_of(const RcclError &e) {
    const char *text = rccl().GetErrorString ? rccl().GetErrorString(e.err) : "?";
    return fail(GBWT_HIP_DEVICE_ERROR, std::string(e.what) + ": " + text);
}

// The gather itself.  `lengths`: this rank's row lengths (u64, device, n rows), `payload`: its rows back to back (device, units of
// `unit` bytes).  On the root: *out_offsets (total_rows + 1, in units) and *out_payload, device memory of the communicator.
gbwt_hip_status gather(gbwt_hip_comm *c, const uint64_t *d_lengths, uint64_t n, const void *d_payload, uint64_t units, uint32_t unit, int root, int layout,
                       bool payload_is_mapped, const uint64_t **out_offsets, const void **out_payload, uint64_t *out_rows, uint64_t *out_units) {
    const Rccl &R = rccl();
    if (!R.why.empty()) return fail(GBWT_HIP_UNSUPPORTED, R.why);
    if (root < 0 || root >= c->world) return fail(GBWT_HIP_BAD_ARGUMENT, "root out of range");
    if (layout < 0 || layout > GBWT_HIP_GATHER_PARTS) return fail(GBWT_HIP_BAD_ARGUMENT, "unknown gather layout");
    // Parts of rows (every rank holds ITS stretch of every row: gbwt_hip_extract_part_device with part = rank) are placed like interleaved
    // rows -- "row" k * world + r = the part of row k that rank r holds, and these back to back ARE row k --, then the offsets are thinned.
    const bool parts_of_rows = layout == GBWT_HIP_GATHER_PARTS;
    const int interleaved = layout != GBWT_HIP_GATHER_BLOCKS;
    try {
        HIP_CHECK(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        const int world = c->world, rank = c->rank;
        const auto t0 = std::chrono::steady_clock::now();
        // 1. {rows, payload units} of every rank
        c->counts.reserve((2 + 2 * static_cast<size_t>(world)) * sizeof(uint64_t));
        uint64_t mine[2] = {n, units};
        uint64_t *d_mine = c->counts.as<uint64_t>(), *d_all = d_mine + 2;
        HIP_CHECK(hipMemcpyAsync(d_mine, mine, sizeof(mine), hipMemcpyHostToDevice, s));
        RCCL_CHECK(R.AllGather(d_mine, d_all, 2, ncclUint64, c->comm, s));
        std::vector<uint64_t> all(2 * static_cast<size_t>(world));
        HIP_CHECK(hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));                      // the one host wait: the receive buffers are sized from it
        // A payload in memory mapped from physical chunks (the rows of a workspace, GBWT_HIP_VMM) is staged through an ordinary allocation
        // before it is sent, unless GBWT_HIP_COMM_DIRECT=1: RCCL's point-to-point path reads the send buffer with a local kernel, which
        // works on any mapping, but no multi-GPU box has been available to this build to prove it on
        const void *send_from = d_payload;
        c->last.staged_send = 0;
        const char *direct = std::getenv("GBWT_HIP_COMM_DIRECT");
        if (rank != root && payload_is_mapped && units != 0 && !(direct && std::atoi(direct) != 0)) {
            c->staged.reserve(units * unit);
            HIP_CHECK(hipMemcpyAsync(c->staged.ptr, d_payload, units * unit, hipMemcpyDeviceToDevice, s));
            send_from = c->staged.ptr;
            c->last.staged_send = 1;
        }
#ifdef GBWT_HIP_TEST_TRANSPORT
        const bool self_send = std::getenv("GBWT_HIP_COMM_SELF_SEND") != nullptr;     // tests on one GPU (the test build only): the root's own part travels through RCCL too
#else
        constexpr bool self_send = false;
#endif
        uint64_t total_rows = 0, total_units = 0;
        std::vector<uint64_t> row_first(world + 1, 0), byte_first(world + 1, 0);
        for (int r = 0; r < world; r++) {
            row_first[r + 1] = row_first[r] + all[2 * r];
            byte_first[r + 1] = byte_first[r] + ((all[2 * r + 1] * unit + 15) & ~uint64_t(15)) + 16;      // parts start at 16 bytes, 16 bytes of slack behind each
        }
        total_rows = row_first[world];
        for (int r = 0; r < world; r++) total_units += all[2 * r + 1];
        // interleaved shards are those of path p -> rank p mod world: the first (rows mod world) ranks hold one row more.  Every rank sees
        // the same counts and takes the same way out, so nobody is left waiting in the exchange.
        // the scans over the rows take an int count (hipcub) whatever the layout: checked here, where every rank sees the same counts and
        // takes the same way out -- before the point-to-point group, not as a device error behind it
        if (total_rows > 0x7FFFFFFFull) return fail(GBWT_HIP_UNSUPPORTED, "more than 2^31 - 1 rows in one gather");
        if (interleaved) {
            for (int r = 0; r < world; r++)
                if (all[2 * r] != total_rows / world + (static_cast<uint64_t>(r) < total_rows % world ? 1 : 0) || (parts_of_rows && all[2 * r] != all[0]))
                    return fail(GBWT_HIP_BAD_ARGUMENT, parts_of_rows ? "gather of row parts: the ranks do not hold the same number of rows"
                                                                     : "interleaved gather: the ranks' row counts are not those of path p -> rank p mod world");
        }
        if (rank != root) {
            // 2. send lengths and payload
            RCCL_CHECK(R.GroupStart());
            if (n) RCCL_CHECK(R.Send(d_lengths, n, ncclUint64, root, c->comm, s));
            if (units) RCCL_CHECK(R.Send(send_from, units * unit, ncclUint8, root, c->comm, s));
            RCCL_CHECK(R.GroupEnd());
            HIP_CHECK(hipStreamSynchronize(s));
            c->last.bytes = 8 * n + units * unit;
            c->last.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (out_offsets) *out_offsets = nullptr;
            if (out_payload) *out_payload = nullptr;
            if (out_rows) *out_rows = 0;
            if (out_units) *out_units = 0;
            return GBWT_HIP_OK;
        }
        // root: receive every part (contiguous shards: the payload straight into its final place)
        const bool in_place = !interleaved;
        c->part_len.reserve(std::max<uint64_t>(total_rows, 1) * sizeof(uint64_t));
        c->offsets.reserve((total_rows + 1) * sizeof(uint64_t));
        c->out.reserve(std::max<uint64_t>(total_units * unit, 16) + 16);
        if (!in_place) c->parts.reserve(byte_first[world] + 16);
        uint64_t *d_part_len = c->part_len.as<uint64_t>();
        uint8_t *d_parts = in_place ? nullptr : c->parts.as<uint8_t>();
        std::vector<uint64_t> unit_first(world + 1, 0);
        for (int r = 0; r < world; r++) unit_first[r + 1] = unit_first[r] + all[2 * r + 1];
        auto part_at = [&](int r) -> uint8_t * { return in_place ? c->out.as<uint8_t>() + unit_first[r] * unit : d_parts + byte_first[r]; };
        RCCL_CHECK(R.GroupStart());
        for (int r = 0; r < world; r++) {
            if (r == rank && !self_send) continue;
            if (all[2 * r]) RCCL_CHECK(R.Recv(d_part_len + row_first[r], all[2 * r], ncclUint64, r, c->comm, s));
            if (all[2 * r + 1]) RCCL_CHECK(R.Recv(part_at(r), all[2 * r + 1] * unit, ncclUint8, r, c->comm, s));
        }
        if (self_send) {
            if (n) RCCL_CHECK(R.Send(d_lengths, n, ncclUint64, root, c->comm, s));
            if (units) RCCL_CHECK(R.Send(d_payload, units * unit, ncclUint8, root, c->comm, s));
        }
        RCCL_CHECK(R.GroupEnd());
        if (!self_send) {   // the root's own part: a copy on the device
            if (n) HIP_CHECK(hipMemcpyAsync(d_part_len + row_first[rank], d_lengths, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
            if (units) HIP_CHECK(hipMemcpyAsync(part_at(rank), d_payload, units * unit, hipMemcpyDeviceToDevice, s));
        }
        // 3. path order
        const size_t tb = scan_temp_bytes(std::max<uint64_t>(total_rows, 1));
        c->scan_temp.reserve(std::max<size_t>(tb, 16));
        std::vector<uint64_t> start_first(world + 1, 0);              // (interleaved) part r's scan of its lengths has rows + 1 entries
        if (in_place) {
            launch_scan(d_part_len, c->offsets.as<uint64_t>(), total_rows, c->scan_temp.ptr, tb, s);
        } else if (total_rows == 0) {
            HIP_CHECK(hipMemsetAsync(c->offsets.ptr, 0, sizeof(uint64_t), s));
        } else {
            c->all_len.reserve(total_rows * sizeof(uint64_t));
            c->part_start.reserve((total_rows + world + 1) * sizeof(uint64_t));
            c->meta.reserve(3 * (static_cast<size_t>(world) + 1) * sizeof(uint64_t));
            uint64_t *d_row_first = c->meta.as<uint64_t>(), *d_byte_first = d_row_first + (world + 1), *d_start_first = d_byte_first + (world + 1);
            uint64_t *d_part_start = c->part_start.as<uint64_t>();
            for (int r = 0; r < world; r++) start_first[r + 1] = start_first[r] + all[2 * r] + 1;
            HIP_CHECK(hipMemcpyAsync(d_row_first, row_first.data(), (world + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(d_byte_first, byte_first.data(), (world + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(d_start_first, start_first.data(), (world + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
            // where every row starts inside its part: one scan per part (world is small)
            for (int r = 0; r < world; r++) launch_scan(d_part_len + row_first[r], d_part_start + start_first[r], all[2 * r], c->scan_temp.ptr, tb, s);
            hipLaunchKernelGGL(k_interleave_lengths, dim3(static_cast<unsigned>((total_rows + 255) / 256)), dim3(256), 0, s, d_part_len, d_row_first,
                               static_cast<uint32_t>(world), total_rows, c->all_len.as<uint64_t>());
            launch_scan(c->all_len.as<uint64_t>(), c->offsets.as<uint64_t>(), total_rows, c->scan_temp.ptr, tb, s);
            const uint32_t slices = static_cast<uint32_t>(std::min<uint64_t>(64, std::max<uint64_t>(1, 4096 / total_rows)));   // few long rows: several workgroups per row
            hipLaunchKernelGGL(k_scatter_rows, dim3(static_cast<unsigned>(std::min<uint64_t>(total_rows, SCATTER_GRID)), slices), dim3(256), 0, s, d_parts, d_byte_first, d_part_start, d_start_first,
                               static_cast<uint32_t>(world), total_rows, c->offsets.as<uint64_t>(), unit, c->out.as<uint8_t>(), slices);
        }
        const uint64_t *d_offsets = c->offsets.as<uint64_t>();
        uint64_t rows_out = total_rows;
        if (parts_of_rows) {
            rows_out = total_rows / world;
            c->row_offsets.reserve((rows_out + 1) * sizeof(uint64_t));
            hipLaunchKernelGGL(k_every_nth_offset, dim3(static_cast<unsigned>((rows_out + 256) / 256)), dim3(256), 0, s, c->offsets.as<uint64_t>(), static_cast<uint32_t>(world), rows_out,
                               c->row_offsets.as<uint64_t>());
            d_offsets = c->row_offsets.as<uint64_t>();
        }
        HIP_CHECK(hipStreamSynchronize(s));
        HIP_CHECK(hipGetLastError());
        c->last.bytes = 8 * total_rows + total_units * unit;
        c->last.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (out_offsets) *out_offsets = d_offsets;
        if (out_payload) *out_payload = c->out.ptr;
        if (out_rows) *out_rows = rows_out;
        if (out_units) *out_units = total_units;
        return GBWT_HIP_OK;
    } catch (const HipError &e) {
        return gbwt_hip::status_of(e);
    } catch (const RcclError &e) {
        return status_of(e);
    }
}

// lengths[k] = offsets[k + 1] - offsets[k]
__global__ void __launch_bounds__(256) k_row_lengths(const uint64_t *offsets, uint64_t n, uint64_t *lengths) {
    const uint64_t k = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (k < n) lengths[k] = offsets[k + 1] - offsets[k];
}

}  // namespace

extern "C" {

gbwt_hip_status gbwt_hip_comm_unique_id(gbwt_hip_unique_id *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!out) return fail(GBWT_HIP_BAD_ARGUMENT, "null output");
    const Rccl &R = rccl();
    if (!R.why.empty()) return fail(GBWT_HIP_UNSUPPORTED, R.why);
    static_assert(sizeof(gbwt_hip_unique_id) == sizeof(ncclUniqueId), "gbwt_hip_unique_id is an ncclUniqueId");
    ncclUniqueId id;
    const ncclResult_t e = R.GetUniqueId(&id);
    if (e != ncclSuccess) return fail(GBWT_HIP_DEVICE_ERROR, std::string("ncclGetUniqueId: ") + R.GetErrorString(e));
    std::memcpy(out->bytes, id.internal, sizeof(id.internal));
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_comm_create(const gbwt_hip_unique_id *id, int rank, int world, int device, gbwt_hip_comm **out) {
    GBWT_HIP_GUARD_BEGIN
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(GBWT_HIP_BAD_ARGUMENT, "bad communicator arguments");
    *out = nullptr;
    const Rccl &R = rccl();
    if (!R.why.empty()) return fail(GBWT_HIP_UNSUPPORTED, R.why);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(GBWT_HIP_NO_DEVICE, "no HIP device available");
    std::unique_ptr<gbwt_hip_comm> c(new gbwt_hip_comm);
    c->rank = rank; c->world = world; c->device = device;
    try {
        const bool trace = std::getenv("GBWT_HIP_COMM_TRACE") != nullptr;
        if (trace) std::fprintf(stderr, "[comm] rank %d of %d: create\n", rank, world);
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        if (trace) std::fprintf(stderr, "[comm] rank %d of %d: stream made\n", rank, world);
        ncclUniqueId nid;
        std::memcpy(nid.internal, id->bytes, sizeof(nid.internal));
        RCCL_CHECK(R.CommInitRank(&c->comm, world, nid, rank));
    } catch (const HipError &e) {
        return gbwt_hip::status_of(e);
    } catch (const RcclError &e) {
        return status_of(e);
    }
    *out = c.release();
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

void gbwt_hip_comm_destroy(gbwt_hip_comm *comm) { delete comm; }

gbwt_hip_status gbwt_hip_comm_last(const gbwt_hip_comm *comm, gbwt_hip_comm_stats *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!comm || !out) return fail(GBWT_HIP_BAD_ARGUMENT, "null argument");
    *out = comm->last;
    return GBWT_HIP_OK;
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_gather_rows(gbwt_hip_comm *comm, const gbwt_hip_index *ix, gbwt_hip_workspace *ws, int root, int interleaved, gbwt_hip_paths *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!comm || !ix || !ws || ws->index != ix || !ws->timed) return fail(GBWT_HIP_BAD_ARGUMENT, "no device-resident extraction on this workspace");
    if (out) *out = gbwt_hip_paths{nullptr, nullptr, 0, 0};
    if (comm->device != ix->device) return fail(GBWT_HIP_BAD_ARGUMENT, "communicator and index are on different devices");
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        const uint64_t n = ws->last_n;
        ws->lengths.reserve(std::max<uint64_t>(n, 1) * sizeof(uint64_t));
        if (n) hipLaunchKernelGGL(k_row_lengths, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, comm->stream, ws->offsets.as<uint64_t>(), n, ws->lengths.as<uint64_t>());
    } catch (const HipError &e) {
        return gbwt_hip::status_of(e);
    }
    const uint64_t *offsets = nullptr;
    const void *payload = nullptr;
    uint64_t rows = 0, units = 0;
    const gbwt_hip_status st = gather(comm, ws->lengths.as<uint64_t>(), ws->last_n, ws->nodes.ptr, ws->last_total, sizeof(uint32_t), root, interleaved,
                                      ws->nodes.reserved != 0, &offsets, &payload, &rows, &units);
    if (st == GBWT_HIP_OK && out) *out = gbwt_hip_paths{offsets, static_cast<const uint32_t *>(payload), units, rows};
    return st;
    GBWT_HIP_GUARD_END
}

gbwt_hip_status gbwt_hip_gather_lines(gbwt_hip_comm *comm, const gbwt_hip_index *ix, gbwt_hip_workspace *ws, int root, int interleaved, gbwt_hip_lines *out) {
    GBWT_HIP_GUARD_BEGIN
    if (!comm || !ix || !ws || ws->index != ix || !ws->lines_cached || ws->lines_slot != 0)
        return fail(GBWT_HIP_BAD_ARGUMENT, "no device-resident GFA lines on this workspace (gbwt_hip_path_lines_device first)");
    if (interleaved == GBWT_HIP_GATHER_PARTS) return fail(GBWT_HIP_BAD_ARGUMENT, "GFA lines are not cut into parts: shard them by path");
    if (out) *out = gbwt_hip_lines{nullptr, nullptr, 0, 0};
    if (comm->device != ix->device) return fail(GBWT_HIP_BAD_ARGUMENT, "communicator and index are on different devices");
    const uint64_t n = ws->lines_key.size();
    try {
        HIP_CHECK(hipSetDevice(ix->device));
        ws->lengths.reserve(std::max<uint64_t>(n, 1) * sizeof(uint64_t));
        if (n) hipLaunchKernelGGL(k_row_lengths, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, comm->stream, ws->gfa_b.as<uint64_t>(), n, ws->lengths.as<uint64_t>());
    } catch (const HipError &e) {
        return gbwt_hip::status_of(e);
    }
    const uint64_t *offsets = nullptr;
    const void *payload = nullptr;
    uint64_t rows = 0, units = 0;
    const gbwt_hip_status st = gather(comm, ws->lengths.as<uint64_t>(), n, n ? ws->gfa_text.ptr : nullptr, ws->lines_total, 1, root, interleaved, false, &offsets, &payload, &rows, &units);
    if (st == GBWT_HIP_OK && out) *out = gbwt_hip_lines{static_cast<const char *>(payload), offsets, units, rows};
    return st;
    GBWT_HIP_GUARD_END
}

}  // extern "C"
