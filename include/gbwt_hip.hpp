// gbwt_hip.hpp -- header-only C++17 mirror of the reference's interface for the LF-step path, over the C ABI of
// gbwt_hip.h.  Same names, argument meaning and "not found" behaviour as jltsiren/gbwt-rs (crate gbz 0.5.1):
//   gbwt_hip::GBWT  <->  gbz::GBWT   src/gbwt.rs:108-384  (len, sequences, ..., start, forward, backward, sequence,
//                                                          find, extend, bd_find, extend_forward, extend_backward)
//   gbwt_hip::GBZ   <->  gbz::GBZ    src/gbz.rs:446-544   (paths, path, search_state, follow_forward, follow_backward)
// Option<T> is std::optional<T>; io::ErrorKind::InvalidData and the reference's asserts are gbwt_hip::Error.
// Every method has a batched form (std::vector in, std::vector out) because one device launch per element would
// waste the GPU; the single-element forms exist so that code -- and tests -- written against the reference read the
// same here.  A handle may be shared by threads; each thread needs its own object of this class for the workspace
// (the reference shares &GBZ across rayon workers, src/bin/gbunzip.rs:421-434): use clone_workspace().
#pragma once

#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gbwt_hip.h"

namespace gbwt_hip {

struct Error : std::runtime_error {
    gbwt_hip_status status;
    Error(gbwt_hip_status s, const std::string &what) : std::runtime_error(what), status(s) {}
};

inline void check(gbwt_hip_status s) {
    if (s != GBWT_HIP_OK) throw Error(s, gbwt_hip_last_error());
}

using Pos = gbwt_hip_pos;                        // bwt::Pos
using SearchState = gbwt_hip_state;              // gbwt::SearchState
using BidirectionalState = gbwt_hip_bd_state;    // gbwt::BidirectionalState
enum class Orientation { Forward = 0, Reverse = 1 };   // support::Orientation

constexpr uint64_t ENDMARKER = 0;
inline uint64_t encode_node(uint64_t id, Orientation o) { return 2 * id + static_cast<uint64_t>(o); }        // src/support.rs:160-162
inline uint64_t node_id(uint64_t node) { return node / 2; }                                                  // src/support.rs:166-168
inline Orientation node_orientation(uint64_t node) { return static_cast<Orientation>(node & 1); }            // src/support.rs:172-174
inline uint64_t flip_node(uint64_t node) { return node ^ 1; }                                                // src/support.rs:186-188
inline uint64_t encode_path(uint64_t path_id, Orientation o) { return 2 * path_id + static_cast<uint64_t>(o); }   // src/support.rs:229-231

// CSR result of a batched extraction: row k = nodes[offsets[k] .. offsets[k + 1])
struct Rows {
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> nodes;
    std::vector<uint32_t> row(size_t k) const { return std::vector<uint32_t>(nodes.begin() + offsets[k], nodes.begin() + offsets[k + 1]); }
};

class GBWT {
public:
    // serialize::load_from::<GBWT | GBZ>(path)
    // `flags`: what the handle is for (GBWT_HIP_OPEN_EXTRACT | _SEARCH | _GFA): only those structures are built in HBM
    explicit GBWT(const std::string &path, int device = 0, uint32_t flags = GBWT_HIP_OPEN_ALL) {
        gbwt_hip_index *h = nullptr;
        check(gbwt_hip_open_file_flags(path.c_str(), device, flags, &h));
        index_.reset(h, gbwt_hip_close);
        init();
    }
    // an index that is already in memory: raw record stream + record starts + header fields (gbwt_hip_open_records)
    GBWT(const uint8_t *data, uint64_t data_len, const std::vector<uint64_t> &starts, uint64_t alphabet_offset, uint64_t alphabet_size,
         uint64_t sequences, uint64_t size, bool bidirectional, int device = 0, uint32_t flags = GBWT_HIP_OPEN_ALL) {
        gbwt_hip_index *h = nullptr;
        check(gbwt_hip_open_records_flags(data, data_len, starts.data(), starts.size(), alphabet_offset, alphabet_size, sequences, size,
                                          bidirectional ? 1 : 0, device, flags, &h));
        index_.reset(h, gbwt_hip_close);
        init();
    }
    GBWT clone_workspace() const { GBWT other(*this, 0); return other; }

    // ---- statistics, src/gbwt.rs:108-182
    uint64_t len() const { return stats_.size; }
    bool is_empty() const { return stats_.size == 0; }
    uint64_t sequences() const { return stats_.sequences; }
    uint64_t alphabet_size() const { return stats_.alphabet_size; }
    uint64_t alphabet_offset() const { return stats_.alphabet_offset; }
    uint64_t effective_size() const { return stats_.alphabet_size - stats_.alphabet_offset; }
    uint64_t first_node() const { return stats_.alphabet_offset + 1; }
    bool has_node(uint64_t node) const { return node > stats_.alphabet_offset && node < stats_.alphabet_size; }
    bool is_bidirectional() const { return stats_.bidirectional != 0; }
    bool has_metadata() const { return stats_.has_metadata != 0; }
    const gbwt_hip_stats &stats() const { return stats_; }
    // bytes the index holds on the device and on the host, and what this object's workspace holds (gbwt_hip_memory_usage)
    gbwt_hip_memory memory_usage() const {
        gbwt_hip_memory m{};
        check(gbwt_hip_memory_usage(index_.get(), ws_.get(), &m));
        return m;
    }

    // ---- navigation, src/gbwt.rs:213-261
    std::vector<std::optional<Pos>> start(const std::vector<uint64_t> &ids) const {
        std::vector<Pos> out(ids.size());
        std::vector<uint8_t> valid(ids.size());
        check(gbwt_hip_start(index_.get(), ws_.get(), ids.data(), ids.size(), out.data(), valid.data()));
        return options(out, valid);
    }
    std::optional<Pos> start(uint64_t id) const { return start(std::vector<uint64_t>{id})[0]; }
    std::vector<std::optional<Pos>> forward(const std::vector<Pos> &pos) const { return step(pos, gbwt_hip_forward); }
    std::optional<Pos> forward(Pos pos) const { return forward(std::vector<Pos>{pos})[0]; }
    // panics in the reference when the index is not bidirectional: throws Error here
    std::vector<std::optional<Pos>> backward(const std::vector<Pos> &pos) const { return step(pos, gbwt_hip_backward); }
    std::optional<Pos> backward(Pos pos) const { return backward(std::vector<Pos>{pos})[0]; }
    // GBWT::sequence(id).collect() for every id; an id >= sequences() (no iterator in the reference) gets an empty row.
    // Size query + fill call: the sequences are walked once, the second call finds the rows in the workspace.
    Rows sequences(const std::vector<uint64_t> &ids) const {
        Rows r;
        r.offsets.assign(ids.size() + 1, 0);
        uint64_t total = 0;
        check(gbwt_hip_extract(index_.get(), ws_.get(), ids.data(), ids.size(), r.offsets.data(), nullptr, 0, &total));
        r.nodes.resize(total);
        if (total) check(gbwt_hip_extract(index_.get(), ws_.get(), ids.data(), ids.size(), r.offsets.data(), r.nodes.data(), total, &total));
        return r;
    }
    // Stretch `part` of `parts` of every row (gbwt_hip_extract_part_device): what one of `parts` GPUs walks of a batch they share; the
    // stretches of a row, in order, are sequences(ids).row(k).  Not in the reference: its parallel axis is the path (src/bin/gbunzip.rs:421-434).
    Rows sequences_part(const std::vector<uint64_t> &ids, uint32_t part, uint32_t parts) const {
        Rows r;
        gbwt_hip_paths p{};
        check(gbwt_hip_extract_part_device(index_.get(), ws_.get(), ids.data(), ids.size(), part, parts, &p));
        r.offsets.assign(ids.size() + 1, 0);
        r.nodes.resize(p.total);
        check(gbwt_hip_copy_result(index_.get(), ws_.get(), r.offsets.data(), p.total ? r.nodes.data() : nullptr, p.total));
        return r;
    }
    std::optional<std::vector<uint32_t>> sequence(uint64_t id) const {
        if (id >= sequences()) return std::nullopt;
        return sequences(std::vector<uint64_t>{id}).row(0);
    }

    // ---- search, src/gbwt.rs:269-384
    std::vector<std::optional<SearchState>> find(const std::vector<uint64_t> &nodes) const {
        std::vector<SearchState> out(nodes.size());
        std::vector<uint8_t> valid(nodes.size());
        check(gbwt_hip_find(index_.get(), ws_.get(), nodes.data(), nodes.size(), out.data(), valid.data()));
        return options(out, valid);
    }
    std::optional<SearchState> find(uint64_t node) const { return find(std::vector<uint64_t>{node})[0]; }
    std::vector<std::optional<SearchState>> extend(const std::vector<SearchState> &states, const std::vector<uint64_t> &nodes) const {
        std::vector<SearchState> out(states.size());
        std::vector<uint8_t> valid(states.size());
        check(gbwt_hip_extend(index_.get(), ws_.get(), states.data(), nodes.data(), states.size(), out.data(), valid.data()));
        return options(out, valid);
    }
    std::optional<SearchState> extend(const SearchState &state, uint64_t node) const {
        return extend(std::vector<SearchState>{state}, std::vector<uint64_t>{node})[0];
    }
    std::vector<std::optional<BidirectionalState>> bd_find(const std::vector<uint64_t> &nodes) const {
        std::vector<BidirectionalState> out(nodes.size());
        std::vector<uint8_t> valid(nodes.size());
        check(gbwt_hip_bd_find(index_.get(), ws_.get(), nodes.data(), nodes.size(), out.data(), valid.data()));
        return options(out, valid);
    }
    std::optional<BidirectionalState> bd_find(uint64_t node) const { return bd_find(std::vector<uint64_t>{node})[0]; }
    std::vector<std::optional<BidirectionalState>> extend_forward(const std::vector<BidirectionalState> &states, const std::vector<uint64_t> &nodes) const {
        return bd_extend(states, nodes, gbwt_hip_extend_forward);
    }
    std::optional<BidirectionalState> extend_forward(const BidirectionalState &state, uint64_t node) const {
        return extend_forward(std::vector<BidirectionalState>{state}, std::vector<uint64_t>{node})[0];
    }
    std::vector<std::optional<BidirectionalState>> extend_backward(const std::vector<BidirectionalState> &states, const std::vector<uint64_t> &nodes) const {
        return bd_extend(states, nodes, gbwt_hip_extend_backward);
    }
    std::optional<BidirectionalState> extend_backward(const BidirectionalState &state, uint64_t node) const {
        return extend_backward(std::vector<BidirectionalState>{state}, std::vector<uint64_t>{node})[0];
    }

    const gbwt_hip_index *handle() const { return index_.get(); }
    gbwt_hip_workspace *workspace() const { return ws_.get(); }

protected:
    GBWT(const GBWT &other, int) : index_(other.index_), stats_(other.stats_) { make_workspace(); }
    void init() {
        check(gbwt_hip_get_stats(index_.get(), &stats_));
        make_workspace();
    }
    void make_workspace() {
        gbwt_hip_workspace *w = nullptr;
        check(gbwt_hip_workspace_create(index_.get(), &w));
        ws_.reset(w, gbwt_hip_workspace_destroy);
    }
    template <class T>
    static std::vector<std::optional<T>> options(const std::vector<T> &values, const std::vector<uint8_t> &valid) {
        std::vector<std::optional<T>> out(values.size());
        for (size_t k = 0; k < values.size(); k++)
            if (valid[k]) out[k] = values[k];
        return out;
    }
    template <class F>
    std::vector<std::optional<Pos>> step(const std::vector<Pos> &pos, F fn) const {
        std::vector<Pos> out(pos.size());
        std::vector<uint8_t> valid(pos.size());
        check(fn(index_.get(), ws_.get(), pos.data(), pos.size(), out.data(), valid.data()));
        return options(out, valid);
    }
    template <class F>
    std::vector<std::optional<BidirectionalState>> bd_extend(const std::vector<BidirectionalState> &states, const std::vector<uint64_t> &nodes, F fn) const {
        std::vector<BidirectionalState> out(states.size());
        std::vector<uint8_t> valid(states.size());
        check(fn(index_.get(), ws_.get(), states.data(), nodes.data(), states.size(), out.data(), valid.data()));
        return options(out, valid);
    }

    std::shared_ptr<gbwt_hip_index> index_;
    std::shared_ptr<gbwt_hip_workspace> ws_;
    gbwt_hip_stats stats_{};
};

class GBZ : public GBWT {
public:
    explicit GBZ(const std::string &path, int device = 0, uint32_t flags = GBWT_HIP_OPEN_ALL) : GBWT(path, device, flags) {}

    uint64_t paths() const { return sequences() / 2; }   // GBZ::paths, src/gbz.rs:446-452
    // GBZ::path(path_id, orientation).collect(): (node id, orientation) pairs, or nullopt (src/gbz.rs:461-466)
    std::optional<std::vector<std::pair<uint64_t, Orientation>>> path(uint64_t path_id, Orientation orientation) const {
        if (path_id >= paths()) return std::nullopt;
        const Rows r = sequences(std::vector<uint64_t>{encode_path(path_id, orientation)});
        std::vector<std::pair<uint64_t, Orientation>> out;
        for (uint32_t v : r.nodes) out.emplace_back(node_id(v), node_orientation(v));
        return out;
    }
    // GBZ::segment_path(path_id, orientation).collect(): (segment id, orientation) pairs -- the segment as its index in the node-to-segment
    // translation --, or nullopt: no such path, or no translation (src/gbz.rs:477-489; SegmentPathIter 1098-1169)
    std::optional<std::vector<std::pair<uint64_t, Orientation>>> segment_path(uint64_t path_id, Orientation orientation) const {
        if (path_id >= paths() || !stats().has_translation) return std::nullopt;
        const uint64_t id = encode_path(path_id, orientation);
        uint64_t offsets[2] = {0, 0}, total = 0;
        check(gbwt_hip_segment_paths(index_.get(), ws_.get(), &id, 1, offsets, nullptr, 0, &total));
        std::vector<uint64_t> tokens(total);
        if (total) check(gbwt_hip_segment_paths(index_.get(), ws_.get(), &id, 1, offsets, tokens.data(), total, &total));
        std::vector<std::pair<uint64_t, Orientation>> out;
        for (uint64_t t : tokens) out.emplace_back(t >> 1, (t & 1) ? Orientation::Reverse : Orientation::Forward);
        return out;
    }
    // GBZ::search_state, src/gbz.rs:508-510
    std::optional<BidirectionalState> search_state(uint64_t id, Orientation orientation) const { return bd_find(encode_node(id, orientation)); }
    // GBZ::follow_forward / follow_backward collected (StateIter, src/gbz.rs:519-544, 1211-1251); nullopt = no iterator
    std::optional<std::vector<BidirectionalState>> follow_forward(const BidirectionalState &state) const { return follow(state, false); }
    std::optional<std::vector<BidirectionalState>> follow_backward(const BidirectionalState &state) const { return follow(state, true); }
    // the lines gbunzip writes for these paths (mode 0 = P-lines, 1 = W-lines) and the whole GFA file
    std::string path_lines(const std::vector<uint64_t> &path_ids, int mode) const {
        uint64_t total = 0;
        check(gbwt_hip_path_lines(index_.get(), ws_.get(), path_ids.data(), path_ids.size(), mode, nullptr, 0, &total));
        std::string text(total, '\0');
        if (total) check(gbwt_hip_path_lines(index_.get(), ws_.get(), path_ids.data(), path_ids.size(), mode, text.data(), total, &total));
        return text;
    }
    // gbunzip --paths MODE (PathMode, src/bin/gbunzip.rs:63-76): GBWT_HIP_PATHS_DEFAULT, _PAN_SN or _REF_ONLY
    void write_gfa(const std::string &path, int path_mode = GBWT_HIP_PATHS_DEFAULT) const {
        check(gbwt_hip_write_gfa_mode(index_.get(), ws_.get(), path.c_str(), path_mode));
    }
    // Metadata::pan_sn_path(path_id) (src/gbwt.rs:709-713) as gbunzip prints it: the name field of the PanSN P-line
    std::string pan_sn_path(uint64_t path_id) const {
        const std::string line = path_lines({path_id}, 2);                 // "P\t<name>\t..."
        const size_t a = line.find('\t'), b = line.find('\t', a + 1);
        return line.substr(a + 1, b - a - 1);
    }

private:
    std::optional<std::vector<BidirectionalState>> follow(const BidirectionalState &state, bool backward) const {
        uint64_t offsets[2] = {0, 0}, total = 0;
        uint8_t valid = 0;
        check(gbwt_hip_follow(index_.get(), ws_.get(), &state, 1, backward ? 1 : 0, offsets, nullptr, 0, &total, &valid));
        if (!valid) return std::nullopt;
        std::vector<BidirectionalState> out(total);
        if (total) check(gbwt_hip_follow(index_.get(), ws_.get(), &state, 1, backward ? 1 : 0, offsets, out.data(), total, &total, &valid));
        return out;
    }
};

// The ordered gather of a sharded extraction (one process per GPU; the reference's writer mutex, src/bin/gbunzip.rs:421-434).  Rank 0 makes
// the id (Comm::unique_id), every rank constructs a Comm from it (collective), and after gbwt_hip_extract_device / _path_lines_device on
// each rank's shard gather_rows / gather_lines leave all rows in path order on `root` (device memory of the communicator).
class Comm {
public:
    static gbwt_hip_unique_id unique_id() {
        gbwt_hip_unique_id id{};
        const gbwt_hip_status st = gbwt_hip_comm_unique_id(&id);
        if (st != GBWT_HIP_OK) throw Error(st, gbwt_hip_last_error());
        return id;
    }
    Comm(const gbwt_hip_unique_id &id, int rank, int world, int device) {
        gbwt_hip_comm *c = nullptr;
        const gbwt_hip_status st = gbwt_hip_comm_create(&id, rank, world, device, &c);
        if (st != GBWT_HIP_OK) throw Error(st, gbwt_hip_last_error());
        comm_.reset(c, gbwt_hip_comm_destroy);
    }
    gbwt_hip_comm *get() const { return comm_.get(); }

private:
    std::shared_ptr<gbwt_hip_comm> comm_;
};

}  // namespace gbwt_hip
