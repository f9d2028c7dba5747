// host_index.cpp -- simple-sds reader for .gbwt / .gbz (product code; see host_index.hpp).
// Layout: SURVEY.md Appendix A.  Every structure is a run of little-endian u64 "elements".
#include "host_index.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <memory>

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

// SparseVector -> sorted values.  value_k = ((pos_k - k) << w) | low[k], pos_k = k-th set bit of high.
std::vector<uint64_t> read_sparse(Elements &in, uint64_t &universe) {
    universe = in.word();
    uint64_t ones = in.word();
    RawBits high = read_raw(in);
    skip_option(in); skip_option(in); skip_option(in);  // rank / select / select_zero supports
    Packed low = read_packed(in);
    if (low.len != ones) throw InvalidData("SparseVector: low length does not match the number of ones");
    if (ones > high.len) throw InvalidData("SparseVector: more ones than bits in the high bitvector");   // also bounds the allocation
    std::vector<uint64_t> values;
    values.reserve(ones + 1);
    const uint64_t w = low.width;
    uint64_t k = 0;
    for (uint64_t wi = 0; wi < high.n_words; wi++) {
        uint64_t word = high.words[wi];
        while (word) {
            uint64_t pos = wi * 64 + static_cast<uint64_t>(__builtin_ctzll(word));
            word &= word - 1;
            if (k >= ones) throw InvalidData("SparseVector: too many ones in the high bitvector");
            const uint64_t upper = pos - k;
            if (w < 64 && upper != 0 && (upper >> (64 - w)) != 0) throw InvalidData("SparseVector: value does not fit 64 bits");
            uint64_t hi = (w >= 64) ? 0 : (upper << w);
            values.push_back(hi | low.get(k));
            k++;
        }
    }
    if (k != ones) throw InvalidData("SparseVector: high bitvector does not have the declared number of ones");
    return values;
}

void read_bytes(Elements &in, std::vector<uint8_t> &out) {
    uint64_t len = in.word();
    if (len / 8 > in.remaining()) throw InvalidData("Vector<u8>: length exceeds the file");   // before rounding: (len + 7) / 8 wraps
    const uint8_t *p = reinterpret_cast<const uint8_t *>(in.words(len / 8 + (len % 8 != 0 ? 1 : 0)));
    out.assign(p, p + len);
}

void finish_strings(Strings &s, std::vector<uint64_t> &&offsets) {
    if (!offsets.empty() && offsets[0] != 0) throw InvalidData("StringArray: First string does not start at offset 0");
    s.offsets = std::move(offsets);
    s.offsets.push_back(s.bytes.size());
    for (size_t i = 1; i < s.offsets.size(); i++)
        if (s.offsets[i] < s.offsets[i - 1]) throw InvalidData("StringArray: offsets are not sorted");
}

// StringArray::load (packed form), src/support.rs:601-647
void read_strings(Elements &in, Strings &s) {
    uint64_t universe;
    std::vector<uint64_t> offsets = read_sparse(in, universe);
    std::vector<uint8_t> alphabet;
    read_bytes(in, alphabet);
    Packed packed = read_packed(in);
    s.bytes.resize(packed.len);
    for (uint64_t i = 0; i < packed.len; i++) {
        uint64_t x = packed.get(i);
        if (x >= alphabet.size()) throw InvalidData("StringArray: packed character outside the alphabet");
        s.bytes[i] = alphabet[x];
    }
    finish_strings(s, std::move(offsets));
}

// StringArray::decompress (zstd form, graph version >= 4), src/support.rs:543-571
void read_strings_zstd(Elements &in, Strings &s) {
    uint64_t universe;
    std::vector<uint64_t> offsets = read_sparse(in, universe);
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
void read_gbwt(Elements &in, HostIndex &h) {
    uint64_t word0 = in.word();
    h.sequences = in.word(); h.size = in.word(); h.alphabet_offset = in.word(); h.alphabet_size = in.word();
    uint64_t flags = in.word();
    check_header("GBWTHeader", word0, flags, GBWT_TAG, 5, 5, 0x7);
    if (!(flags & 4)) throw InvalidData("GBWTHeader: SDSL format is not supported");
    h.bidirectional = (flags & 1) != 0;
    // The reference also overwrites the `source` tag in memory (src/gbwt.rs:409); nothing on the hot
    // path reads it, and keeping the file's value lets save_index_file() write a loaded file back unchanged.
    read_tags(in, h.tags);

    uint64_t universe;
    h.starts = read_sparse(in, universe);
    read_bytes(in, h.data);
    if (universe != h.data.size()) throw InvalidData("BWT: Index / data length mismatch");
    for (size_t i = 0; i < h.starts.size(); i++)
        if (h.starts[i] > h.data.size() || (i > 0 && h.starts[i] < h.starts[i - 1]))
            throw InvalidData("BWT: record starts are not sorted offsets into the data");
    h.starts.push_back(h.data.size());

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
void read_graph(Elements &in, HostIndex &h) {
    uint64_t word0 = in.word();
    uint64_t nodes = in.word();
    h.graph_nodes = nodes;
    uint64_t flags = in.word();
    check_header("GraphHeader", word0, flags, GRAPH_TAG, 3, 4, 0x3);
    if (!(flags & 2)) throw InvalidData("GraphHeader: SDSL format is not supported");
    h.has_translation = (flags & 1) != 0;
    if ((word0 >> 32) >= 4) read_strings_zstd(in, h.sequences_labels); else read_strings(in, h.sequences_labels);
    read_strings(in, h.segment_names);
    if (h.has_translation == (h.segment_names.size() == 0))
        throw InvalidData("Graph: Translation flag does not match the presence of segment names");
    h.segment_starts = read_sparse(in, h.mapping_len);
    if (h.has_translation) {
        if (h.mapping_len <= nodes) throw InvalidData("Graph: Node-to-segment mapping does not match the number of nodes");
        if (h.mapping_len != h.sequences_labels.size() + 1) throw InvalidData("Graph: Node-to-segment mapping does not match the number of sequences");
        if (h.segment_starts.size() != h.segment_names.size()) throw InvalidData("Graph: Node-to-segment mapping does not match the number of segments");
    }
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

HostIndex load_index_file(const std::string &path) {
    std::unique_ptr<FILE, int (*)(FILE *)> f(std::fopen(path.c_str(), "rb"), std::fclose);
    if (!f) throw IoError("cannot open " + path);
    std::fseek(f.get(), 0, SEEK_END);
    long sz = std::ftell(f.get());
    std::fseek(f.get(), 0, SEEK_SET);
    if (sz < 0) throw IoError("cannot stat " + path);
    if (sz % 8 != 0) throw InvalidData("file size is not a multiple of 8 bytes");
    std::vector<uint64_t> buf(static_cast<size_t>(sz) / 8 + 1);
    if (sz > 0 && std::fread(buf.data(), 1, static_cast<size_t>(sz), f.get()) != static_cast<size_t>(sz)) throw IoError("short read on " + path);
    Elements in(buf.data(), static_cast<uint64_t>(sz) / 8);

    HostIndex h;
    uint32_t tag = static_cast<uint32_t>(in.peek());
    if (tag == GBZ_TAG) {
        // GBZ::load, src/gbz.rs:674-717
        uint64_t word0 = in.word(), flags = in.word();
        check_header("GBZHeader", word0, flags, GBZ_TAG, 1, 2, 0);
        read_tags(in, h.gbz_tags);
        read_gbwt(in, h);
        if (!h.bidirectional) throw InvalidData("GBZ: The GBWT index is not bidirectional");
        read_graph(in, h);
        uint64_t potential_nodes = (h.alphabet_size - (h.alphabet_offset + 1)) / 2;
        if (h.sequences_labels.size() != potential_nodes)
            throw InvalidData("GBZ: Mismatch between GBWT alphabet size and Graph sequence count");
        h.is_gbz = true;
    } else {
        read_gbwt(in, h);
    }
    if (!in.at_end()) throw InvalidData("trailing data after the index");
    return h;
}

std::vector<std::pair<uint32_t, uint32_t>> decompress_endmarker(const HostIndex &h, uint64_t limit) {
    std::vector<std::pair<uint32_t, uint32_t>> out;
    if (h.records() == 0 || h.starts[1] <= h.starts[0]) return out;
    const uint8_t *p = h.data.data() + h.starts[0], *end = h.data.data() + h.starts[1];
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
