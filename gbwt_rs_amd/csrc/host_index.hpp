// host_index.hpp -- host-side image of a GBWT / GBZ file (product code).
//
// Reads the simple-sds serialization the reference loads with serialize::load_from
// (src/gbwt.rs:402-438, src/gbz.rs:674-717, src/graph.rs:296-338, src/headers.rs) and flattens it
// into the arrays the device needs: the record byte stream, a dense record-start array decoded
// from the Elias-Fano index (the device never runs select), and -- for the GFA rows -- path names,
// node label lengths and the node->segment translation.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace gbwt_hip {

// Raised for anything the reference reports as io::ErrorKind::InvalidData.
struct InvalidData : std::runtime_error { using std::runtime_error::runtime_error; };
struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };
// A well-formed index beyond the 32-bit widths of the device side (include/gbwt_hip.h: "Widths"): GBWT_HIP_UNSUPPORTED.
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

// gbwt::PathName, src/gbwt.rs:912-926
struct PathName { uint32_t sample, contig, phase, fragment; };

// An allocator whose resize() leaves new elements uninitialised: the 4.4 G label characters of an HPRC-sized GBZ are unpacked by many
// threads into a vector that a value-initialising resize() would first zero -- and page in -- on ONE thread (a second of a two-second open,
// profiles/r05_c4_open_trace.txt).  Every byte such a vector is resized to is written before it is read.
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAllocator<U>; };
    DefaultInitAllocator() = default;
    template <class U> DefaultInitAllocator(const DefaultInitAllocator<U> &) noexcept {}
    template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... Args> void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};
using Bytes = std::vector<uint8_t, DefaultInitAllocator<uint8_t>>;
using Words = std::vector<uint64_t, DefaultInitAllocator<uint64_t>>;   // (the same for the 218 M record starts / 109 M label offsets of such a file)

// support::StringArray flattened: offsets[n+1] into bytes
struct Strings {
    Words offsets{0};
    Bytes bytes;
    size_t size() const { return offsets.size() - 1; }
    std::string str(size_t i) const { return std::string(bytes.begin() + offsets[i], bytes.begin() + offsets[i + 1]); }
    size_t len(size_t i) const { return offsets[i + 1] - offsets[i]; }
    bool find(const std::string &s, uint64_t &id) const;
};

struct HostIndex {
    // Header<GBWTPayload>, src/headers.rs:190-234
    uint64_t sequences = 0, size = 0, alphabet_offset = 0, alphabet_size = 0;
    bool bidirectional = false;
    // BWT, src/bwt.rs:97-100: data + record starts (n_records + 1 entries, last = data.size())
    Bytes data;
    Words starts;
    // The Elias-Fano index of the record starts as it lies in the mapped file (SparseVector, Appendix A: value_k = ((pos_k - k) << w) | low[k]).
    // An open that decodes the starts ON THE DEVICE (starts_on_device, set by its on_located callback) uploads these words and lets the host
    // decode into `starts` -- which the GFA tables and the graph lines read, nothing the device passes need -- run in the background: until
    // finish() has returned `starts` must not be touched; records() and record_start() answer from the located index.
    struct StartsView { uint64_t ones = 0, universe = 0, high_words = 0, low_width = 1, low_words = 0; const uint64_t *high = nullptr, *low = nullptr; };
    StartsView starts_view;
    bool starts_on_device = false;
    uint64_t records() const { return starts_on_device ? starts_view.ones : (starts.empty() ? 0 : starts.size() - 1); }
    uint64_t record_start(uint64_t k) const;      // starts[k]; while the host decode is in the background: select(k) on the located index (k of a few: a scan from the front)

    // tags of the GBWT (key -> value, lower-cased keys), src/support.rs:915-1020
    std::vector<std::pair<std::string, std::string>> tags;
    // tags of the GBZ container (src/gbz.rs:124-130) and the opaque document-array samples
    // (src/gbwt.rs:100,417); kept only so that a loaded file can be written back unchanged
    std::vector<std::pair<std::string, std::string>> gbz_tags;
    std::vector<uint64_t> da_samples;
    const std::string *tag(const std::string &key) const;

    // Metadata, src/gbwt.rs:623-896
    bool has_metadata = false;
    uint64_t metadata_flags = 0, sample_count = 0, haplotype_count = 0, contig_count = 0;
    std::vector<PathName> path_names;
    Strings sample_names, contig_names;
    // generic paths were stored with phase GENERIC_HAPLOTYPE (vg convention) and converted to 0 at load
    // (src/gbwt.rs:879-886); older files store 0 directly.  Remembered so write-back is unchanged.
    bool generic_phase_on_disk = true;

    // Graph (GBZ only), src/graph.rs:84-89
    bool is_gbz = false, has_translation = false;
    uint64_t graph_nodes = 0;              // Header<GraphPayload>.nodes
    Strings sequences_labels;   // node labels, one per potential node
    Strings segment_names;
    Words segment_starts;  // node id of the first node of each segment (mapping ones)
    uint64_t mapping_len = 0;              // universe of the node-to-segment mapping

    // load_index_file_into(..., background = true): the copy of the record bytes into `data` and the decoding of the node labels --
    // nothing the device passes of an open need -- go on in a thread of their own while the caller uploads and builds on the GPU.
    // Until finish() has returned, `data`, `sequences_labels` must not be touched; the record bytes are read through
    // record_bytes() (the mapped file).  finish() joins the thread and throws what it threw; without a background it does nothing.
    struct Pending;
    std::shared_ptr<Pending> pending;
    const uint8_t *file_data = nullptr;
    uint64_t file_data_len = 0;
    const uint8_t *record_bytes() const { return file_data ? file_data : data.data(); }
    uint64_t record_bytes_len() const { return file_data ? file_data_len : data.size(); }
    void finish();

    // THE HOST'S OWN IMAGE OF THE RECORDS, MADE ON FIRST USE (round 6).  `data` and `starts` are read by the graph lines of the whole-file
    // writer (S / L lines: GBZ::has_node, Record::decompress_edges), by the tables of a node-to-segment translation and by save_index_file --
    // by nothing an open, an extraction, a search or a lines request does.  For an HPRC-sized file they are 3.5 GB of freshly faulted pages
    // (0.6-0.9 s on sixteen threads next to the device passes, which take 0.4: the longest thing in that open, profiles/r06_c4_open.txt).  A
    // load with `lazy_records` whose caller decodes the starts on the device therefore leaves them unmade and keeps the file mapped for as
    // long as the index lives; ensure_records() -- idempotent, safe from several threads -- makes them when somebody asks.  Everything
    // else answers from the mapping: record_bytes(), records(), record_start().
    struct LazyRecords;
    std::shared_ptr<LazyRecords> lazy_records;
    void ensure_records() const;
    bool records_made() const;
};

// Parses a .gbwt or .gbz (detected by the header tag).  Throws InvalidData / IoError.
HostIndex load_index_file(const std::string &path);
// The same into `out`, which must stay where it is until out.finish() has returned when `background` is set (see HostIndex::pending).
// `on_located` (optional) runs once the file has been walked -- header fields set, record_bytes() valid -- and before anything is decoded:
// an open starts the host-to-device copy of the record bytes there, next to the Elias-Fano decode of the starts.
// `lazy_records`: see HostIndex::ensure_records (only with `background`, a file of 4 MB or more, and an on_located that has set starts_on_device).
void load_index_file_into(const std::string &path, HostIndex &out, bool background, const std::function<void(HostIndex &)> &on_located = nullptr, bool lazy_records = false);

// Writes the index back in the simple-sds format (GBWT v5; GBZ v1 container with an uncompressed
// graph, version 3), following the Serialize impls src/gbwt.rs:389-400, src/gbz.rs:662-672,
// src/graph.rs:284-294.  Used by the synthetic generator and by the writer round-trip tests.
void save_index_file(const HostIndex &index, const std::string &path, bool as_gbz);

// Record::decompress of record 0, the endmarker (GBWT::load does the same at load time, src/gbwt.rs:413-414: `endmarker =
// record.decompress()`): one (node, offset) per sequence = GBWT::start(id) without the guards.  Part of reading the file, like the
// Elias-Fano index: the run stream is parsed byte by byte (RLEIter::next, src/support.rs:1413-1430; Record::decompress
// src/bwt.rs:466-478).  A malformed record ends the list where the reference's iterator would end it; at most `limit` entries.
std::vector<std::pair<uint32_t, uint32_t>> decompress_endmarker(const HostIndex &index, uint64_t limit);

// Builds the host image from raw parts (gbwt_hip_open_records).
HostIndex index_from_records(const uint8_t *data, uint64_t data_len, const uint64_t *starts, uint64_t n_records,
                             uint64_t alphabet_offset, uint64_t alphabet_size, uint64_t n_sequences, uint64_t size,
                             bool bidirectional);

}  // namespace gbwt_hip
