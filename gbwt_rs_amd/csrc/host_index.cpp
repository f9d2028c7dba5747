// host_index.cpp -- simple-sds reader for .gbwt / .gbz (product code; see host_index.hpp).
// Layout: SURVEY.md Appendix A.  Every structure is a run of little-endian u64 "elements".
#include "host_index.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace gbwt_hip {

namespace {

constexpr uint32_t GBWT_TAG = 0x6B376B37u, METADATA_TAG = 0x6B375E7Au, GRAPH_TAG = 0x6B3764AFu, GBZ_TAG = 0x205A4247u;
constexpr uint32_t GENERIC_HAPLOTYPE = 0xFFFFFFFFu;  // src/lib.rs
const char *const GENERIC_SAMPLE = "_gbwt_ref";      // src/lib.rs

class Elements {
public:
    Elements(const uint64_t *w, uint64_t n) : w_(w), n_(n) {}
    uint64_t word() {
        if (pos_ >= n_) throw InvalidData("unexpected end of file");
        return w_[pos_++];
    }
    const uint64_t *words(uint64_t count) {
        if (count > n_ - pos_) throw InvalidData("unexpected end of file");
        const uint64_t *p = w_ + pos_;
        pos_ += count;
        return p;
    }
    uint64_t pos() const { return pos_; }
    uint64_t remaining() const { return n_ - pos_; }   // elements left: the bound of every length field that follows
    bool at_end() const { return pos_ == n_; }
    uint64_t peek() const { return pos_ < n_ ? w_[pos_] : 0; }

private:
    const uint64_t *w_;
    uint64_t n_, pos_ = 0;
};

// A packed integer array (IntVector) viewed in place.
struct Packed {
    uint64_t len = 0, width = 1;
    const uint64_t *words = nullptr;
    uint64_t n_words = 0;
    uint64_t get(uint64_t k) const {
        uint64_t bit = k * width, wi = bit >> 6, off = bit & 63;
        uint64_t x = words[wi] >> off;
        if (off + width > 64) x |= words[wi + 1] << (64 - off);
        return width == 64 ? x : (x & ((uint64_t(1) << width) - 1));
    }
};

struct RawBits { uint64_t len = 0; const uint64_t *words = nullptr; uint64_t n_words = 0; };

// Every length below comes from the file: it is compared with what is left of the file BEFORE it is rounded,
// multiplied or used to allocate, so that a corrupt word is io::ErrorKind::InvalidData as in the reference and never an
// overflow, a std::length_error or an out-of-bounds read.
RawBits read_raw(Elements &in) {
    RawBits r;
    r.len = in.word();
    r.n_words = in.word();
    if (r.n_words != r.len / 64 + (r.len % 64 != 0 ? 1 : 0)) throw InvalidData("RawVector: word count does not match length");
    r.words = in.words(r.n_words);
    return r;
}

Packed read_packed(Elements &in) {
    Packed p;
    p.len = in.word();
    p.width = in.word();
    RawBits raw = read_raw(in);
    uint64_t bits = 0;
    if (p.width == 0 || p.width > 64 || __builtin_mul_overflow(p.len, p.width, &bits) || raw.len != bits)
        throw InvalidData("IntVector: invalid width / length");
    p.words = raw.words;
    p.n_words = raw.n_words;
    return p;
}

void skip_option(Elements &in) {
    uint64_t size = in.word();
    in.words(size);   // checked against the rest of the file
}

// The large sections of a file -- the record starts, the record bytes, the node labels -- are LOCATED while the file is walked (a few
// header words each) and DECODED afterwards, side by side on a few threads (Deferred): the Elias-Fano decode of two million starts,
// the copy of the record bytes and the unpacking of a million labels were three quarters of the 36 ms the headline GBZ took to parse.
struct Deferred {
    std::vector<std::function<void()>> tasks, background;      // background: may still be running when the loader returns (HostIndex::pending)
    std::function<void()> starts_task;                         // the decode of the record starts: foreground, or background when the caller decodes them on the device
    std::function<void()> records_copy;                        // the host's copy of the record bytes: background -- or, with the starts, on first use (HostIndex::ensure_records)
    bool small = false;                                        // a file of a few megabytes: threads would cost more than they save
    void run() {
        if (tasks.empty()) return;
        const unsigned workers = small ? 1u : std::min<unsigned>(static_cast<unsigned>(tasks.size()), std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
        std::atomic<size_t> next{0};
        std::exception_ptr failure;
        std::atomic<bool> failed{false};
        auto work = [&]() {
            for (size_t k = next++; k < tasks.size(); k = next++) {
                try { tasks[k](); }
                catch (...) { if (!failed.exchange(true)) failure = std::current_exception(); }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < workers; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        tasks.clear();
        if (failure) std::rethrow_exception(failure);
    }
};

// SparseVector, located: value_k = ((pos_k - k) << w) | low[k], pos_k = k-th set bit of high.
struct SparseView { uint64_t universe = 0, ones = 0; RawBits high; Packed low; };

SparseView locate_sparse(Elements &in) {
    SparseView v;
    v.universe = in.word();
    v.ones = in.word();
    v.high = read_raw(in);
    skip_option(in); skip_option(in); skip_option(in);  // rank / select / select_zero supports
    v.low = read_packed(in);
    if (v.low.len != v.ones) throw InvalidData("SparseVector: low length does not match the number of ones");
    if (v.ones > v.high.len) throw InvalidData("SparseVector: more ones than bits in the high bitvector");   // also bounds the allocation
    return v;
}

// fn(piece, pieces) on a few threads (the caller's included); exceptions of the pieces are rethrown here
template <class Fn>
void run_pieces(unsigned pieces, Fn fn) {
    std::exception_ptr failure;
    std::atomic<bool> failed{false};
    auto guarded = [&](unsigned p) {
        try { fn(p); }
        catch (...) { if (!failed.exchange(true)) failure = std::current_exception(); }
    };
    std::vector<std::thread> pool;
    for (unsigned p = 1; p < pieces; p++) pool.emplace_back(guarded, p);
    guarded(0);
    for (auto &t : pool) t.join();
    if (failure) std::rethrow_exception(failure);
}

// A thread per 256 K items up to four, and from eight million items a thread per two million up to 32.
// (Rounds 3-5 kept ONE thread below eight million items: on those boxes the copy of the record bytes that runs next to the decodes took 9 ms
// for the headline's 60 MB and hid a 5.7 ms decode of its two million starts.  Round 6 measured both again inside an open
// (profiles/r06_open_probe.txt): the copy takes 1.4-1.5 ms, the one-thread decode 3.8 ms -- the decode IS the parse; four threads: 1.9 ms,
// parse 5.1 -> 3.1 ms; eight add nothing.)
// (... the 4.4 G label characters and 218 M record starts of config 4 at its stated size took 0.9 + 0.4 s of a 2 s open on eight threads --
// profiles/r05_c4_open_trace.txt)
inline unsigned pieces_for(uint64_t items) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    if (items < (uint64_t(1) << 23)) return std::max(1u, std::min(std::min(4u, hw), static_cast<unsigned>(items >> 18)));
    const unsigned by_size = static_cast<unsigned>(std::min<uint64_t>(items >> 21, 32));
    return std::max(std::min(8u, hw), std::min(by_size, hw));
}

Words decode_sparse(const SparseView &v) {
    // A few threads, each with a stretch of the high bitvector (the rank of its first one comes from the popcounts before it): the
    // 32 million record starts of a config-4-shaped GBZ were 110 ms on one thread.  The declared number of ones must be the number of
    // set bits BEFORE it sizes an allocation (a corrupt count would ask for up to 64 times the file size and surface as "out of host
    // memory" instead of InvalidData).
    const uint64_t n_words = v.high.n_words;
    const unsigned pieces = pieces_for(v.ones);
    std::vector<uint64_t> first_word(pieces + 1), first_rank(pieces + 1, 0), count(pieces, 0);
    for (unsigned p = 0; p <= pieces; p++) first_word[p] = n_words * p / pieces;
    run_pieces(pieces, [&](unsigned p) {
        uint64_t c = 0;
        for (uint64_t wi = first_word[p]; wi < first_word[p + 1]; wi++) c += static_cast<uint64_t>(__builtin_popcountll(v.high.words[wi]));
        count[p] = c;
    });
    for (unsigned p = 0; p < pieces; p++) first_rank[p + 1] = first_rank[p] + count[p];
    if (first_rank[pieces] != v.ones) throw InvalidData("SparseVector: high bitvector does not have the declared number of ones");
    Words values;
    values.reserve(v.ones + 1);                               // the callers append a sentinel
    values.resize(v.ones);
    const uint64_t w = v.low.width;
    run_pieces(pieces, [&](unsigned p) {
        uint64_t k = first_rank[p];
        for (uint64_t wi = first_word[p]; wi < first_word[p + 1]; wi++) {
            uint64_t word = v.high.words[wi];
            while (word) {
                const uint64_t pos = wi * 64 + static_cast<uint64_t>(__builtin_ctzll(word));
                word &= word - 1;
                const uint64_t upper = pos - k;
                if (w < 64 && upper != 0 && (upper >> (64 - w)) != 0) throw InvalidData("SparseVector: value does not fit 64 bits");
                values[k] = ((w >= 64) ? 0 : (upper << w)) | v.low.get(k);
                k++;
            }
        }
    });
    return values;
}

Words read_sparse(Elements &in, uint64_t &universe) {
    const SparseView v = locate_sparse(in);
    universe = v.universe;
    return decode_sparse(v);
}

void read_bytes(Elements &in, std::vector<uint8_t> &out) {
    uint64_t len = in.word();
    if (len / 8 > in.remaining()) throw InvalidData("Vector<u8>: length exceeds the file");   // before rounding: (len + 7) / 8 wraps
    const uint8_t *p = reinterpret_cast<const uint8_t *>(in.words(len / 8 + (len % 8 != 0 ? 1 : 0)));
    out.assign(p, p + len);
}

void finish_strings(Strings &s, Words &&offsets) {
    if (!offsets.empty() && offsets[0] != 0) throw InvalidData("StringArray: First string does not start at offset 0");
    s.offsets = std::move(offsets);
    s.offsets.push_back(s.bytes.size());
    const size_t n = s.offsets.size();
    const unsigned pieces = pieces_for(n);
    run_pieces(pieces, [&](unsigned p) {
        for (size_t i = std::max<size_t>(1, n * p / pieces), end = n * (p + 1) / pieces; i < end; i++)
            if (s.offsets[i] < s.offsets[i - 1]) throw InvalidData("StringArray: offsets are not sorted");
    });
}

// StringArray::load (packed form), src/support.rs:601-647; with `later` the decoding is a deferred task
void read_strings(Elements &in, Strings &s, Deferred *later = nullptr, bool background = false) {
    const SparseView view = locate_sparse(in);
    std::vector<uint8_t> alphabet;
    read_bytes(in, alphabet);
    const Packed packed = read_packed(in);
    auto decode = [view, alphabet, packed, &s]() {
        // the offsets and the characters side by side (one after the other until round 6: 109 M offsets, then a gigabyte of characters --
        // the longest thing left on the host side of an HPRC-sized open once the record image became lazy)
        Words offsets;
        s.bytes.resize(packed.len);
        const unsigned pieces = pieces_for(packed.len);
        auto characters = [&]() {
            run_pieces(pieces, [&](unsigned p) {
                for (uint64_t i = packed.len * p / pieces, end = packed.len * (p + 1) / pieces; i < end; i++) {
                    uint64_t x = packed.get(i);
                    if (x >= alphabet.size()) throw InvalidData("StringArray: packed character outside the alphabet");
                    s.bytes[i] = alphabet[x];
                }
            });
        };
        if (view.ones >= (uint64_t(1) << 18)) run_pieces(2, [&](unsigned p) { if (p == 0) offsets = decode_sparse(view); else characters(); });
        else { offsets = decode_sparse(view); characters(); }
        finish_strings(s, std::move(offsets));
    };
    if (later) (background ? later->background : later->tasks).push_back(decode); else decode();
}

// StringArray::decompress (zstd form, graph version >= 4), src/support.rs:543-571
void read_strings_zstd(Elements &in, Strings &s) {
    uint64_t universe;
    Words offsets = read_sparse(in, universe);
    uint64_t total = in.word();
    std::vector<uint8_t> compressed;
    read_bytes(in, compressed);
    // Streaming decompression into a buffer that grows with the data actually produced: `total` comes from the file and
    // must not size an allocation before the stream has proved it (the reference reads to the end of the stream and then
    // compares the lengths, src/support.rs:562-566).  Frames written by the reference's streaming encoder do not carry
    // their content size, so there is nothing to check `total` against in advance.
    struct InBuf { const void *src; size_t size, pos; };
    struct OutBuf { void *dst; size_t size, pos; };
    using create_fn = void *(*)();
    using free_fn = size_t (*)(void *);
    using stream_fn = size_t (*)(void *, OutBuf *, InBuf *);
    using iserror_fn = unsigned (*)(size_t);
    static void *lib = nullptr;
    if (!lib) lib = dlopen("libzstd.so.1", RTLD_NOW);
    if (!lib) lib = dlopen("libzstd.so", RTLD_NOW);
    if (!lib) throw IoError("zstd-compressed node labels need libzstd.so.1, which could not be loaded");
    auto create = reinterpret_cast<create_fn>(dlsym(lib, "ZSTD_createDStream"));
    auto release = reinterpret_cast<free_fn>(dlsym(lib, "ZSTD_freeDStream"));
    auto step = reinterpret_cast<stream_fn>(dlsym(lib, "ZSTD_decompressStream"));
    auto iserr = reinterpret_cast<iserror_fn>(dlsym(lib, "ZSTD_isError"));
    if (!create || !release || !step || !iserr) throw IoError("libzstd.so.1 lacks the streaming decompression API");
    std::unique_ptr<void, free_fn> stream(create(), release);
    if (!stream) throw IoError("ZSTD_createDStream failed");
    const char *mismatch = "StringArray: Decompressed string length does not match the expected length";
    InBuf src{compressed.data(), compressed.size(), 0};
    s.bytes.clear();
    size_t produced = 0;
    bool more = src.size != 0;   // one or more frames, like the reference's decoder reading to the end
    while (more) {
        if (produced == s.bytes.size()) {
            if (produced > total) throw InvalidData(mismatch);
            // at most one byte more than `total`: a longer stream shows itself by filling that byte
            const uint64_t cap = total == ~uint64_t(0) ? total : total + 1;
            const uint64_t room = std::min<uint64_t>(cap - produced, std::max<uint64_t>(produced, uint64_t(1) << 16));
            s.bytes.resize(produced + std::max<uint64_t>(room, 1));
        }
        OutBuf dst{s.bytes.data(), s.bytes.size(), produced};
        const size_t before_in = src.pos;
        const size_t hint = step(stream.get(), &dst, &src);   // 0 = a frame is complete and flushed
        if (iserr(hint)) throw InvalidData("StringArray: zstd stream is corrupt");
        const bool full = dst.pos == s.bytes.size();
        if (dst.pos == produced && src.pos == before_in) throw InvalidData("StringArray: zstd stream makes no progress");
        produced = dst.pos;
        if (src.pos == src.size && !(full && hint != 0)) {     // input used up and nothing left to flush
            if (hint != 0) throw InvalidData("StringArray: zstd stream is truncated");
            more = false;
        }
    }
    if (produced != total) throw InvalidData(mismatch);
    s.bytes.resize(total);
    finish_strings(s, std::move(offsets));
}

// Dictionary::load, src/support.rs:821-838 (the sorted-id permutation is not needed here)
void read_dictionary(Elements &in, Strings &s) {
    read_strings(in, s);
    read_packed(in);
}

// Tags::load, src/support.rs:988-1007
void read_tags(Elements &in, std::vector<std::pair<std::string, std::string>> &tags) {
    Strings lin;
    read_strings(in, lin);
    if (lin.size() % 2 != 0) throw InvalidData("Tags: Key without a value");
    for (size_t i = 0; i < lin.size() / 2; i++) {
        std::string key = lin.str(2 * i), value = lin.str(2 * i + 1);
        std::transform(key.begin(), key.end(), key.begin(), [](unsigned char c) { return std::tolower(c); });
        for (auto &kv : tags)
            if (kv.first == key) throw InvalidData("Tags: Duplicate keys");
        tags.emplace_back(key, value);
    }
}

// Header<T>::validate, src/headers.rs:101-115
void check_header(const char *name, uint64_t word0, uint64_t flags, uint32_t tag, uint32_t min_version,
                  uint32_t max_version, uint64_t mask) {
    uint32_t t = static_cast<uint32_t>(word0), v = static_cast<uint32_t>(word0 >> 32);
    char msg[128];
    if (t != tag) { snprintf(msg, sizeof(msg), "%s: Invalid tag %X", name, t); throw InvalidData(msg); }
    if (v < min_version || v > max_version) {
        snprintf(msg, sizeof(msg), "%s: Invalid version %u (expected %u to %u)", name, v, min_version, max_version);
        throw InvalidData(msg);
    }
    if ((flags & mask) != flags) {
        snprintf(msg, sizeof(msg), "%s: Invalid flags %llX for version %u", name, static_cast<unsigned long long>(flags), v);
        throw InvalidData(msg);
    }
}

// Metadata::load, src/gbwt.rs:846-890
void read_metadata(Elements &in, HostIndex &h) {
    uint64_t word0 = in.word();
    h.sample_count = in.word(); h.haplotype_count = in.word(); h.contig_count = in.word();
    h.metadata_flags = in.word();
    check_header("MetadataHeader", word0, h.metadata_flags, METADATA_TAG, 2, 2, 0x7);
    uint64_t n_paths = in.word();
    if (n_paths > in.remaining() / 2) throw InvalidData("Metadata: path name count exceeds the file");
    const uint64_t *pw = in.words(2 * n_paths);
    h.path_names.resize(n_paths);
    if (n_paths) std::memcpy(h.path_names.data(), pw, n_paths * sizeof(PathName));
    if (((h.metadata_flags & 1) != 0) == h.path_names.empty())
        throw InvalidData("Metadata: Path name flag does not match the presence of path names");
    read_dictionary(in, h.sample_names);
    if (h.metadata_flags & 2) {
        if (h.sample_count != h.sample_names.size()) throw InvalidData("Metadata: Sample count does not match the number of sample names");
    } else if (h.sample_names.size() != 0) throw InvalidData("Metadata: Sample names are present without the sample name flag");
    read_dictionary(in, h.contig_names);
    if (h.metadata_flags & 4) {
        if (h.contig_count != h.contig_names.size()) throw InvalidData("Metadata: Contig count does not match the number of contig names");
    } else if (h.contig_names.size() != 0) throw InvalidData("Metadata: Contig names are present without the contig name flag");
    uint64_t generic;
    if (h.sample_names.find(GENERIC_SAMPLE, generic)) {
        h.generic_phase_on_disk = false;
        for (auto &p : h.path_names)
            if (p.sample == generic && p.phase == GENERIC_HAPLOTYPE) { p.phase = 0; h.generic_phase_on_disk = true; }
    }
    h.has_metadata = true;
}

// GBWT::load, src/gbwt.rs:402-438 (+ BWT::load, src/bwt.rs:176-185)
void read_gbwt(Elements &in, HostIndex &h, Deferred &later) {
    uint64_t word0 = in.word();
    h.sequences = in.word(); h.size = in.word(); h.alphabet_offset = in.word(); h.alphabet_size = in.word();
    uint64_t flags = in.word();
    check_header("GBWTHeader", word0, flags, GBWT_TAG, 5, 5, 0x7);
    if (!(flags & 4)) throw InvalidData("GBWTHeader: SDSL format is not supported");
    h.bidirectional = (flags & 1) != 0;
    // The reference also overwrites the `source` tag in memory (src/gbwt.rs:409); nothing on the hot
    // path reads it, and keeping the file's value lets save_index_file() write a loaded file back unchanged.
    read_tags(in, h.tags);

    const SparseView index = locate_sparse(in);
    const uint64_t data_len = in.word();
    if (data_len / 8 > in.remaining()) throw InvalidData("Vector<u8>: length exceeds the file");   // before rounding: (len + 7) / 8 wraps
    const uint8_t *data = reinterpret_cast<const uint8_t *>(in.words(data_len / 8 + (data_len % 8 != 0 ? 1 : 0)));
    if (index.universe != data_len) throw InvalidData("BWT: Index / data length mismatch");
    h.file_data = data; h.file_data_len = data_len;
    // (the host's copy of the record bytes: 1.8 GB for config 4 at its stated size -- 0.9 s as ONE assign() that pages its target in on one thread,
    // the longest thing in that open; in pieces on a few threads into a vector whose resize() does not touch it: profiles/r06_c4_open.txt)
    later.records_copy = [&h, data, data_len]() {
        h.data.resize(data_len);
        const unsigned pieces = data_len >= (uint64_t(64) << 20) ? std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1u;
        run_pieces(pieces, [&](unsigned p) {
            const uint64_t lo = data_len / pieces * p, hi = p + 1 == pieces ? data_len : data_len / pieces * (p + 1);
            if (hi > lo) std::memcpy(h.data.data() + lo, data + lo, hi - lo);
        });
    };
    h.starts_view = HostIndex::StartsView{index.ones, index.universe, index.high.n_words, index.low.width, index.low.n_words, index.high.words, index.low.words};
    later.starts_task = [&h, index, data_len]() {
        h.starts = decode_sparse(index);
        for (size_t i = 0; i < h.starts.size(); i++)
            if (h.starts[i] > data_len || (i > 0 && h.starts[i] < h.starts[i - 1]))
                throw InvalidData("BWT: record starts are not sorted offsets into the data");
        h.starts.push_back(data_len);
    };

    uint64_t da_len = in.word();  // document array samples: opaque pass-through in the reference (417)
    const uint64_t *da = in.words(da_len);
    h.da_samples.assign(da, da + da_len);
    uint64_t meta_size = in.word();  // Option<Metadata>
    if (meta_size > 0) {
        uint64_t before = in.pos();
        read_metadata(in, h);
        if (in.pos() - before != meta_size) throw InvalidData("GBWT: Metadata size does not match the option header");
    }
    if (((flags & 2) != 0) != h.has_metadata) throw InvalidData("GBWT: Invalid metadata flag in the header");
    if (h.has_metadata && (h.metadata_flags & 1)) {
        uint64_t expected = h.bidirectional ? h.sequences / 2 : h.sequences;
        if (!h.path_names.empty() && h.path_names.size() != expected) throw InvalidData("GBWT: Invalid path count in the metadata");
    }
}

// Graph::load, src/graph.rs:296-338
void read_graph(Elements &in, HostIndex &h, Deferred &later) {
    uint64_t word0 = in.word();
    uint64_t nodes = in.word();
    h.graph_nodes = nodes;
    uint64_t flags = in.word();
    check_header("GraphHeader", word0, flags, GRAPH_TAG, 3, 4, 0x3);
    if (!(flags & 2)) throw InvalidData("GraphHeader: SDSL format is not supported");
    h.has_translation = (flags & 1) != 0;
    if ((word0 >> 32) >= 4) read_strings_zstd(in, h.sequences_labels); else read_strings(in, h.sequences_labels, &later, true);
    read_strings(in, h.segment_names);
    if (h.has_translation == (h.segment_names.size() == 0))
        throw InvalidData("Graph: Translation flag does not match the presence of segment names");
    h.segment_starts = read_sparse(in, h.mapping_len);
    if (h.has_translation) {
        if (h.mapping_len <= nodes) throw InvalidData("Graph: Node-to-segment mapping does not match the number of nodes");
        if (h.segment_starts.size() != h.segment_names.size()) throw InvalidData("Graph: Node-to-segment mapping does not match the number of segments");
    }
}

// the checks of Graph::load / GBZ::load that need the decoded node labels
void check_graph(const HostIndex &h) {
    if (h.has_translation && h.mapping_len != h.sequences_labels.size() + 1) throw InvalidData("Graph: Node-to-segment mapping does not match the number of sequences");
    const uint64_t potential_nodes = (h.alphabet_size - (h.alphabet_offset + 1)) / 2;
    if (h.sequences_labels.size() != potential_nodes)
        throw InvalidData("GBZ: Mismatch between GBWT alphabet size and Graph sequence count");
}

}  // namespace

bool Strings::find(const std::string &s, uint64_t &id) const {
    for (size_t i = 0; i < size(); i++) {
        if (len(i) == s.size() && std::memcmp(bytes.data() + offsets[i], s.data(), s.size()) == 0) { id = i; return true; }
    }
    return false;
}

const std::string *HostIndex::tag(const std::string &key) const {
    for (auto &kv : tags)
        if (kv.first == key) return &kv.second;
    return nullptr;
}

namespace {
// The file, mapped (no copy, no zero-filled staging buffer) or -- where mmap does not work -- read into memory.
struct FileImage {
    const uint64_t *words = nullptr;
    uint64_t n_words = 0;
    void *mapping = nullptr;
    size_t mapped = 0;
    std::unique_ptr<uint64_t[]> owned;
    FileImage() = default;
    FileImage(const FileImage &) = delete;
    FileImage &operator=(const FileImage &) = delete;
    ~FileImage() { if (mapping) munmap(mapping, mapped); }
};

void open_image(const std::string &path, FileImage &image) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw IoError("cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); throw IoError("cannot stat " + path); }
    const size_t sz = static_cast<size_t>(st.st_size);
    if (sz % 8 != 0) { ::close(fd); throw InvalidData("file size is not a multiple of 8 bytes"); }
    image.n_words = sz / 8;
    if (sz == 0) { ::close(fd); return; }
    void *m = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    if (m != MAP_FAILED) {
        image.mapping = m; image.mapped = sz; image.words = static_cast<const uint64_t *>(m);
        ::close(fd);
        return;
    }
    image.owned.reset(new uint64_t[sz / 8]);
    size_t got = 0;
    while (got < sz) {
        const ssize_t r = ::read(fd, reinterpret_cast<char *>(image.owned.get()) + got, sz - got);
        if (r <= 0) { ::close(fd); throw IoError("short read on " + path); }
        got += static_cast<size_t>(r);
    }
    ::close(fd);
    image.words = image.owned.get();
}
}  // namespace

struct HostIndex::Pending {
    FileImage image;
    std::thread worker;
    std::exception_ptr failure;
    ~Pending() { if (worker.joinable()) worker.join(); }
};

struct HostIndex::LazyRecords {
    std::shared_ptr<Pending> image;                            // the mapping `make` reads -- and record_bytes() / starts_view point into: lives as long as the index
    std::function<void()> make;
    std::once_flag once;
    std::atomic<bool> made{false};
};

void HostIndex::finish() {
    if (!pending) return;
    std::shared_ptr<Pending> p = std::move(pending);
    pending.reset();
    if (p->worker.joinable()) p->worker.join();
    if (!lazy_records) {                                           // record_bytes() now answers from `data`, `starts` is there (or the decode has
        file_data = nullptr; file_data_len = 0;                    // thrown, below): the mapping goes
        starts_on_device = false; starts_view = StartsView{};
    }
    if (p->failure) std::rethrow_exception(p->failure);
    if (is_gbz) check_graph(*this);
}

bool HostIndex::records_made() const { return !lazy_records || lazy_records->made.load(std::memory_order_acquire); }

void HostIndex::ensure_records() const {
    if (!lazy_records) return;
    LazyRecords &lazy = *lazy_records;
    std::call_once(lazy.once, [&lazy]() { lazy.make(); lazy.made.store(true, std::memory_order_release); });   // (a throw leaves the flag unset: the next caller tries again)
}

namespace {
// GBWT_HIP_TRACE_OPEN=1: the loader's phases on stderr, like the device side of an open (capi_open.hip: OpenTrace)
struct LoadTrace {
    bool on = std::getenv("GBWT_HIP_TRACE_OPEN") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[load] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};
}  // namespace

void load_index_file_into(const std::string &path, HostIndex &h, bool background, const std::function<void(HostIndex &)> &on_located, bool lazy_records) {
    LoadTrace trace;
    h = HostIndex();
    std::shared_ptr<HostIndex::Pending> pending = std::make_shared<HostIndex::Pending>();
    open_image(path, pending->image);
    trace.mark("file mapped");
    Elements in(pending->image.words, pending->image.n_words);
    Deferred later;
    later.small = pending->image.n_words < (uint64_t(1) << 19);          // < 4 MB
    uint32_t tag = static_cast<uint32_t>(in.peek());
    if (tag == GBZ_TAG) {
        // GBZ::load, src/gbz.rs:674-717
        uint64_t word0 = in.word(), flags = in.word();
        check_header("GBZHeader", word0, flags, GBZ_TAG, 1, 2, 0);
        read_tags(in, h.gbz_tags);
        read_gbwt(in, h, later);
        if (!h.bidirectional) throw InvalidData("GBZ: The GBWT index is not bidirectional");
        read_graph(in, h, later);
        h.is_gbz = true;
    } else {
        read_gbwt(in, h, later);
    }
    if (!in.at_end()) throw InvalidData("trailing data after the index");
    if (!background || later.small) {
        for (auto &t : later.background) later.tasks.push_back(std::move(t));
        later.background.clear();
    }
    const bool in_background = background && !later.small;
    trace.mark("sections located");
    if (on_located) {
        // record_bytes() answers from the mapping from here on.  `h` keeps its share of the mapping across the decodes below as well:
        // what on_located started (gbwt_hip_open_file: a thread copying the record bytes to the device) may still be reading the file
        // when a decode throws, and the caller can only stop it after this function has unwound -- the mapping then goes with `h`,
        // which the caller destroys after that thread.
        h.pending = pending;
        on_located(h);
    }
    // the record starts: decoded here, or -- when the caller has taken the located index to the device -- behind the caller's back
    // ... and with them the host's copy of the record bytes: behind the caller's back as well, or -- lazy_records -- when somebody asks
    if (h.starts_on_device && in_background && lazy_records && later.starts_task && later.records_copy) {
        h.lazy_records = std::make_shared<HostIndex::LazyRecords>();
        h.lazy_records->image = pending;
        HostIndex *target = &h;                                    // (the index stays where it is: it is a member of the handle)
        h.lazy_records->make = [target, copy = std::move(later.records_copy), decode = std::move(later.starts_task)]() {
            try { run_pieces(2, [&](unsigned p) { if (p == 0) copy(); else decode(); }); }
            catch (...) { target->data = Bytes(); target->starts = Words(); throw; }
        };
        later.records_copy = nullptr; later.starts_task = nullptr;
    }
    if (later.starts_task) {
        if (h.starts_on_device && in_background) later.background.push_back(std::move(later.starts_task));
        else { h.starts_on_device = false; later.tasks.push_back(std::move(later.starts_task)); }
    }
    if (later.records_copy) (in_background ? later.background : later.tasks).push_back(std::move(later.records_copy));
    later.run();
    trace.mark("foreground decodes");
    if (!background) {
        h.pending.reset();
        h.file_data = nullptr; h.file_data_len = 0;
        h.starts_view = HostIndex::StartsView{};
        if (h.is_gbz) check_graph(h);
        return;
    }
    // the mapping stays until finish(), whether or not anything is left to do in the background (a caller may still be copying from it)
    HostIndex::Pending *raw = pending.get();
    std::vector<std::function<void()>> jobs = std::move(later.background);
    if (!jobs.empty())
        raw->worker = std::thread([raw, jobs]() {
            const bool trace_jobs = std::getenv("GBWT_HIP_TRACE_OPEN") != nullptr;
            try { run_pieces(static_cast<unsigned>(jobs.size()), [&](unsigned p) {        // the record bytes, the record starts and the labels side by side
                const auto t0 = std::chrono::steady_clock::now();
                jobs[p]();
                if (trace_jobs) std::fprintf(stderr, "[load] (background job %u of %zu)     %8.3f ms\n", p, jobs.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }); }
            catch (...) { raw->failure = std::current_exception(); }
        });
    h.pending = std::move(pending);
}

HostIndex load_index_file(const std::string &path) {
    HostIndex h;
    load_index_file_into(path, h, false);
    return h;
}

uint64_t HostIndex::record_start(uint64_t k) const {
    if (!starts_on_device) return k < starts.size() ? starts[k] : record_bytes_len();
    const StartsView &v = starts_view;
    if (k >= v.ones) return record_bytes_len();
    uint64_t seen = 0;
    for (uint64_t wi = 0; wi < v.high_words; wi++) {
        uint64_t word = v.high[wi];
        const uint64_t c = static_cast<uint64_t>(__builtin_popcountll(word));
        if (seen + c <= k) { seen += c; continue; }
        for (; seen < k; seen++) word &= word - 1;
        const uint64_t pos = wi * 64 + static_cast<uint64_t>(__builtin_ctzll(word)), upper = pos - k, w = v.low_width;
        const uint64_t bit = k * w, lw = bit >> 6, off = bit & 63;
        uint64_t low = lw < v.low_words ? v.low[lw] >> off : 0;
        if (off + w > 64 && lw + 1 < v.low_words) low |= v.low[lw + 1] << (64 - off);
        if (w < 64) low &= (uint64_t(1) << w) - 1;
        return std::min(record_bytes_len(), (w >= 64 ? 0 : upper << w) | low);
    }
    return record_bytes_len();
}

std::vector<std::pair<uint32_t, uint32_t>> decompress_endmarker(const HostIndex &h, uint64_t limit) {
    std::vector<std::pair<uint32_t, uint32_t>> out;
    const uint64_t first = h.record_start(0), second = h.record_start(1);
    if (h.records() == 0 || second <= first) return out;
    const uint8_t *p = h.record_bytes() + first, *end = h.record_bytes() + second;
    auto varint = [&](uint64_t &v) -> bool {                  // ByteCodeIter::next, src/support.rs:1151-1164
        v = 0;
        unsigned shift = 0;
        while (p < end) {
            const uint8_t b = *p++;
            if (shift < 64) v += static_cast<uint64_t>(b & 0x7F) << shift;
            shift += 7;
            if (!(b & 0x80)) return true;
        }
        return false;
    };
    uint64_t sigma = 0, node = 0;
    if (!varint(sigma) || sigma == 0 || sigma > static_cast<uint64_t>(end - p)) return out;   // every edge takes at least two bytes
    std::vector<uint64_t> nodes(sigma), offsets(sigma);
    for (uint64_t e = 0; e < sigma; e++) {                       // Record::decompress_edges, src/bwt.rs:378-395
        uint64_t delta = 0, off = 0;
        if (!varint(delta) || !varint(off)) return out;
        node += delta;
        nodes[e] = node; offsets[e] = off;
    }
    // RLE::sanitize, src/support.rs:1292-1296: sigma >= 255 -> two varints per run; else value + sigma * (len - 1) in one byte,
    // and a varint with the rest of the length behind a byte that holds threshold - 1
    const uint64_t threshold = sigma >= 255 ? 0 : 256 / sigma;
    out.reserve(std::min<uint64_t>(limit, h.sequences));
    while (p < end && out.size() < limit) {
        uint64_t value = 0, len = 0;
        if (sigma >= 255) {
            if (!varint(value) || !varint(len)) break;
            len++;
        } else {
            const uint64_t b = *p++;
            value = b % sigma; len = b / sigma + 1;
            if (len == threshold) {
                uint64_t extra = 0;
                if (!varint(extra)) break;
                len += extra;
            }
        }
        if (value >= sigma) break;                               // malformed
        for (uint64_t k = 0; k < len && out.size() < limit; k++)
            out.emplace_back(static_cast<uint32_t>(nodes[value]), static_cast<uint32_t>(offsets[value]++));
    }
    return out;
}

HostIndex index_from_records(const uint8_t *data, uint64_t data_len, const uint64_t *starts, uint64_t n_records,
                             uint64_t alphabet_offset, uint64_t alphabet_size, uint64_t n_sequences, uint64_t size,
                             bool bidirectional) {
    HostIndex h;
    h.sequences = n_sequences; h.size = size; h.alphabet_offset = alphabet_offset; h.alphabet_size = alphabet_size;
    h.bidirectional = bidirectional;
    h.data.assign(data, data + data_len);
    h.starts.assign(starts, starts + n_records);
    for (uint64_t i = 0; i < n_records; i++)
        if (h.starts[i] > data_len || (i > 0 && h.starts[i] < h.starts[i - 1]))
            throw InvalidData("BWT: record starts are not sorted offsets into the data");
    h.starts.push_back(data_len);
    return h;
}

}  // namespace gbwt_hip
