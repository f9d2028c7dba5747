// sds_writer.cpp -- writes a HostIndex in the simple-sds serialization (product code).
// Mirrors the Serialize impls of the reference: GBWT src/gbwt.rs:389-400, BWT src/bwt.rs:165-174,
// Metadata src/gbwt.rs:822-844, Tags src/support.rs:981-986, StringArray src/support.rs:580-599,
// Dictionary src/support.rs:816-819, Graph src/graph.rs:284-294 (uncompressed, version 3),
// GBZ src/gbz.rs:662-672 (container version 1).  Elias-Fano parameters follow simple-sds 0.4
// (SURVEY.md Appendix A); the round-trip tests rewrite the reference fixtures byte for byte.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

#include "host_index.hpp"

namespace gbwt_hip {

namespace {

constexpr uint32_t GBWT_TAG = 0x6B376B37u, METADATA_TAG = 0x6B375E7Au, GRAPH_TAG = 0x6B3764AFu, GBZ_TAG = 0x205A4247u;
constexpr uint32_t GENERIC_HAPLOTYPE = 0xFFFFFFFFu;
const char *const GENERIC_SAMPLE = "_gbwt_ref";

struct Out {
    std::vector<uint64_t> w;
    void word(uint64_t x) { w.push_back(x); }
    void bytes(const uint8_t *p, uint64_t len) {  // Vec<u8>
        word(len);
        size_t at = w.size();
        w.resize(at + (len + 7) / 8, 0);
        if (len) std::memcpy(w.data() + at, p, len);
    }
};

uint64_t bit_len(uint64_t x) { return x == 0 ? 1 : 64 - static_cast<uint64_t>(__builtin_clzll(x)); }

// Packed integer array under construction (IntVector).
struct PackedOut {
    uint64_t len = 0, width;
    std::vector<uint64_t> words;
    explicit PackedOut(uint64_t wd) : width(wd) {}
    void push(uint64_t v) {
        uint64_t bit = len * width, wi = bit >> 6, off = bit & 63;
        if (words.size() < wi + 2) words.resize(wi + 2, 0);
        words[wi] |= v << off;
        if (off + width > 64) words[wi + 1] |= v >> (64 - off);
        len++;
    }
    void write(Out &out) const {
        uint64_t bits = len * width, n_words = (bits + 63) / 64;
        out.word(len); out.word(width);
        out.word(bits); out.word(n_words);
        for (uint64_t i = 0; i < n_words; i++) out.word(words[i]);
    }
};

// SparseVector (Elias-Fano).  Empty vectors are SparseVector::default(): low width 64.
void write_sparse(Out &out, uint64_t universe, const uint64_t *values, uint64_t ones) {
    uint64_t w = 1;
    if (ones == 0 && universe == 0) w = 64;
    else if (ones > 0 && ones <= universe) {
        double ideal = std::log2(static_cast<double>(universe) * std::log(2.0) / static_cast<double>(ones));
        double r = std::round(ideal);
        w = r < 1.0 ? 1 : static_cast<uint64_t>(r);
    }
    uint64_t buckets = 0;
    if (w < 64) { buckets = universe >> w; if (universe & ((uint64_t(1) << w) - 1)) buckets++; }
    else if (universe != 0) buckets = 1;
    uint64_t high_bits = ones + buckets, n_words = (high_bits + 63) / 64;
    std::vector<uint64_t> high(n_words + 1, 0);
    PackedOut low(w);
    for (uint64_t k = 0; k < ones; k++) {
        uint64_t v = values[k];
        uint64_t pos = (w < 64 ? (v >> w) : 0) + k;
        high[pos >> 6] |= uint64_t(1) << (pos & 63);
        low.push(w < 64 ? (v & ((uint64_t(1) << w) - 1)) : v);
    }
    out.word(universe);
    out.word(ones);                                   // BitVector: ones, RawVector, three absent supports
    out.word(high_bits); out.word(n_words);
    for (uint64_t i = 0; i < n_words; i++) out.word(high[i]);
    out.word(0); out.word(0); out.word(0);
    low.write(out);
}

// StringArray::serialize_body, src/support.rs:584-599
void write_strings(Out &out, const Strings &s) {
    uint64_t n = s.size();
    uint64_t universe = n ? s.offsets[n - 1] + 1 : 0;
    write_sparse(out, universe, s.offsets.data(), n);
    bool present[256] = {false};
    for (uint8_t b : s.bytes) present[b] = true;
    std::vector<uint8_t> alphabet;
    uint64_t pack[256] = {0};
    for (int c = 0; c < 256; c++)
        if (present[c]) { pack[c] = alphabet.size(); alphabet.push_back(static_cast<uint8_t>(c)); }
    out.bytes(alphabet.data(), alphabet.size());
    PackedOut packed(bit_len(std::max<uint64_t>(alphabet.size(), 1) - 1));
    for (uint8_t b : s.bytes) packed.push(pack[b]);
    packed.write(out);
}

// Dictionary::serialize_body + TryFrom<StringArray>, src/support.rs:816-871
void write_dictionary(Out &out, const Strings &s) {
    write_strings(out, s);
    std::vector<uint64_t> sorted(s.size());
    for (uint64_t i = 0; i < sorted.size(); i++) sorted[i] = i;
    std::sort(sorted.begin(), sorted.end(), [&](uint64_t a, uint64_t b) {
        return std::lexicographical_compare(s.bytes.begin() + s.offsets[a], s.bytes.begin() + s.offsets[a + 1],
                                            s.bytes.begin() + s.offsets[b], s.bytes.begin() + s.offsets[b + 1]);
    });
    PackedOut ids(sorted.empty() ? 1 : bit_len(sorted.size() - 1));
    for (uint64_t v : sorted) ids.push(v);
    ids.write(out);
}

// Tags::serialize_body, src/support.rs:981-986 (keys in BTreeMap order)
void write_tags(Out &out, std::vector<std::pair<std::string, std::string>> tags) {
    std::sort(tags.begin(), tags.end());
    Strings lin;
    for (auto &kv : tags) {
        for (const std::string *str : {&kv.first, &kv.second}) {
            lin.bytes.insert(lin.bytes.end(), str->begin(), str->end());
            lin.offsets.push_back(lin.bytes.size());
        }
    }
    write_strings(out, lin);
}

void write_metadata(Out &out, const HostIndex &h) {
    out.word(static_cast<uint64_t>(METADATA_TAG) | (uint64_t(2) << 32));
    out.word(h.sample_count); out.word(h.haplotype_count); out.word(h.contig_count);
    out.word(h.metadata_flags);
    // generic paths are stored with phase GENERIC_HAPLOTYPE (src/gbwt.rs:829-838)
    uint64_t generic = 0;
    bool have_generic = h.generic_phase_on_disk && h.sample_names.find(GENERIC_SAMPLE, generic);
    out.word(h.path_names.size());
    for (PathName p : h.path_names) {
        if (have_generic && p.sample == generic && p.phase == 0) p.phase = GENERIC_HAPLOTYPE;
        out.word(static_cast<uint64_t>(p.sample) | (static_cast<uint64_t>(p.contig) << 32));
        out.word(static_cast<uint64_t>(p.phase) | (static_cast<uint64_t>(p.fragment) << 32));
    }
    write_dictionary(out, h.sample_names);
    write_dictionary(out, h.contig_names);
}

void write_gbwt(Out &out, const HostIndex &h) {
    out.word(static_cast<uint64_t>(GBWT_TAG) | (uint64_t(5) << 32));
    out.word(h.sequences); out.word(h.size); out.word(h.alphabet_offset); out.word(h.alphabet_size);
    out.word((h.bidirectional ? 1u : 0u) | (h.has_metadata ? 2u : 0u) | 4u);
    write_tags(out, h.tags);
    h.ensure_records();
    write_sparse(out, h.data.size(), h.starts.data(), h.records());
    out.bytes(h.data.data(), h.data.size());
    out.word(h.da_samples.size());
    for (uint64_t x : h.da_samples) out.word(x);
    if (h.has_metadata) {
        Out meta;
        write_metadata(meta, h);
        out.word(meta.w.size());
        out.w.insert(out.w.end(), meta.w.begin(), meta.w.end());
    } else out.word(0);
}

void write_graph(Out &out, const HostIndex &h) {
    out.word(static_cast<uint64_t>(GRAPH_TAG) | (uint64_t(3) << 32));
    out.word(h.graph_nodes);
    out.word((h.has_translation ? 1u : 0u) | 2u);
    write_strings(out, h.sequences_labels);
    write_strings(out, h.segment_names);
    write_sparse(out, h.mapping_len, h.segment_starts.data(), h.segment_starts.size());
}

}  // namespace

void save_index_file(const HostIndex &h, const std::string &path, bool as_gbz) {
    Out out;
    if (as_gbz) {
        out.word(static_cast<uint64_t>(GBZ_TAG) | (uint64_t(1) << 32));
        out.word(0);
        write_tags(out, h.gbz_tags);
        write_gbwt(out, h);
        write_graph(out, h);
    } else write_gbwt(out, h);
    std::unique_ptr<FILE, int (*)(FILE *)> f(std::fopen(path.c_str(), "wb"), std::fclose);
    if (!f) throw IoError("cannot create " + path);
    if (std::fwrite(out.w.data(), 8, out.w.size(), f.get()) != out.w.size()) throw IoError("short write on " + path);
}

}  // namespace gbwt_hip
