// synth.cpp -- synthetic GBWT / GBZ generator (host-only; see gbwt_synth.h).
//
// Construction rule: the record of node v lists, for every visit of v, the
// successor on that sequence (0 = ENDMARKER at the end).  Visits inside a record are ordered by
// (predecessor node, position of the visit in the predecessor's record), visits that start a
// sequence come first in sequence-id order.  Edge (v -> w) stores the number of visits in record w
// whose predecessor is smaller than v.  Runs are maximal equal-successor stretches, written with
// the reference's codecs (ByteCode src/support.rs:1063-1070, RLE src/support.rs:1238-1248,
// BWTBuilder::append src/bwt.rs:241-253).
#include "gbwt_synth.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../csrc/host_index.hpp"

using gbwt_hip::HostIndex;
using gbwt_hip::PathName;

namespace {

// ---- codecs (encoder side) ------------------------------------------------------------------
template <class Out> inline void put_varint(Out &out, uint64_t v) {
    while (v > 0x7F) { out.push_back(static_cast<uint8_t>((v & 0x7F) | 0x80)); v >>= 7; }
    out.push_back(static_cast<uint8_t>(v));
}

template <class Out> inline void put_run(Out &out, uint64_t sigma, uint64_t value, uint64_t len) {
    if (sigma >= 255) { put_varint(out, value); put_varint(out, len - 1); return; }
    uint64_t threshold = 256 / sigma;
    if (len < threshold) out.push_back(static_cast<uint8_t>(value + sigma * (len - 1)));
    else { out.push_back(static_cast<uint8_t>(value + sigma * (threshold - 1))); put_varint(out, len - threshold); }
}

// Appends one record: edges (node ascending) + body given as successor ranks with run merging.
template <class Out>
struct RecordWriterT {
    Out &out;
    uint64_t sigma = 0, run_value = 0, run_len = 0;
    explicit RecordWriterT(Out &o) : out(o) {}
    void begin(const std::vector<std::pair<uint64_t, uint64_t>> &edges) {
        sigma = edges.size();
        put_varint(out, sigma);
        uint64_t prev = 0;
        for (auto &e : edges) { put_varint(out, e.first - prev); put_varint(out, e.second); prev = e.first; }
        run_len = 0;
    }
    inline void push(uint64_t rank, uint64_t count = 1) {
        if (run_len && rank == run_value) { run_len += count; return; }
        if (run_len) put_run(out, sigma, run_value, run_len);
        run_value = rank; run_len = count;
    }
    void end() { if (run_len) put_run(out, sigma, run_value, run_len); run_len = 0; }
};
template <class Out> RecordWriterT<Out> make_record_writer(Out &o) { return RecordWriterT<Out>(o); }

// ---- deterministic RNG (xoshiro256** seeded by splitmix64) -------------------------------------
struct Rng {
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        for (auto &x : s) { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; x = z ^ (z >> 31); }
    }
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    inline uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    inline double uniform() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
    inline uint64_t below(uint64_t n) { return static_cast<uint64_t>((static_cast<unsigned __int128>(next()) * n) >> 64); }
};

}  // namespace

struct gbwt_synth {
    HostIndex index;
    // chain truth
    uint64_t sites = 0, haplotypes = 0, alleles = 0;
    uint64_t extra = 0;             // alleles >= 1 are insertions: `extra` more nodes behind the allele node ...
    uint64_t indel_every = 1;       // ... at the sites s with s % indel_every == 0 (the tail ids of the other sites stay unused)
    uint64_t chop = 1;              // every logical node (anchor, allele, inserted node) is a chain of `chop` nodes with consecutive ids
    uint32_t label_mode = 0;        // 0: 1 bp labels; 1: labels of realistic lengths, 1 .. 1024 bp (gbwt_synth.h: gbwt_synth_chain_labeled)
    std::vector<uint64_t> bits;     // alleles == 2: site-major bit rows
    uint64_t row_words = 0;
    std::vector<uint16_t> choices;  // alleles > 2: site-major
    // explicit truth (from_paths)
    std::vector<uint64_t> path_offsets;
    std::vector<uint32_t> path_nodes;
    // merged truth (gbwt_synth_merge): the chains this index was put together from (without their record streams), the first path of
    // each in the merged numbering and the shift of its node ids
    std::vector<gbwt_synth> parts;
    std::vector<uint64_t> part_first_path, part_shift;

    inline uint32_t allele(uint64_t s, uint64_t h) const {
        if (alleles == 2) return static_cast<uint32_t>((bits[s * row_words + (h >> 6)] >> (h & 63)) & 1);
        return choices[s * haplotypes + h];
    }
    inline void set_allele(uint64_t s, uint64_t h, uint32_t a) {
        if (alleles == 2) { if (a) bits[s * row_words + (h >> 6)] |= uint64_t(1) << (h & 63); }
        else choices[s * haplotypes + h] = static_cast<uint16_t>(a);
    }
    // node ids of site s, ascending: the `chop` pieces of the anchor (the last one branches), then allele by allele its pieces -- the first
    // one and its tails: chop - 1 for allele 0, chop * (1 + extra) - 1 for the others (an insertion site uses all of them) -- so that the
    // pieces of every logical node have consecutive ids, as the nodes of a chopped GFA segment have
    inline uint64_t tails_max(uint32_t a) const { return a == 0 ? chop - 1 : chop * (1 + extra) - 1; }
    inline uint64_t tails_before(uint32_t a) const { return a == 0 ? 0 : (chop - 1) + (a - 1) * (chop * (1 + extra) - 1); }
    inline uint64_t stride() const { return chop + alleles + tails_before(static_cast<uint32_t>(alleles)); }
    inline uint64_t anchor_first(uint64_t s) const { return s * stride() + 1; }
    inline uint64_t anchor_last(uint64_t s) const { return s * stride() + chop; }
    inline uint64_t allele_slot(uint32_t a) const { return chop + a + tails_before(a); }            // slot of the first piece within the site (slot k <-> id s * stride + k + 1)
    inline uint64_t allele_id(uint64_t s, uint32_t a) const { return s * stride() + allele_slot(a) + 1; }
    inline uint64_t tail_id(uint64_t s, uint32_t a, uint64_t e) const { return allele_id(s, a) + 1 + e; }
    inline uint64_t extra_at(uint64_t s) const { return s % indel_every == 0 ? extra : 0; }
    inline uint64_t tails_at(uint64_t s, uint32_t a) const { return a == 0 ? chop - 1 : chop * (1 + extra_at(s)) - 1; }   // pieces behind the first one
    inline uint64_t last_id(uint64_t s, uint32_t a) const { const uint64_t t = tails_at(s, a); return t == 0 ? allele_id(s, a) : tail_id(s, a, t - 1); }
};

namespace {

void draw_alleles(gbwt_synth &g, uint32_t model, uint32_t founders, double rho, double zipf, uint64_t seed) {
    const uint64_t S = g.sites, n = g.haplotypes, A = g.alleles;
    Rng rng(seed);
    std::vector<double> cdf;
    if (A > 2) {
        cdf.resize(A);
        double total = 0;
        for (uint64_t a = 0; a < A; a++) { total += 1.0 / std::pow(static_cast<double>(a + 1), zipf); cdf[a] = total; }
        for (auto &x : cdf) x /= total;
    }
    auto draw = [&](double p_site) -> uint32_t {
        if (A == 2) return rng.uniform() < p_site ? 1u : 0u;
        double u = rng.uniform();
        uint32_t a = static_cast<uint32_t>(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
        return a >= A ? static_cast<uint32_t>(A - 1) : a;
    };
    if (model == GBWT_SYNTH_IID) {
        for (uint64_t s = 0; s < S; s++) {
            double p = 0.05 + 0.9 * rng.uniform();
            for (uint64_t h = 0; h < n; h++) g.set_allele(s, h, draw(p));
        }
        return;
    }
    if (founders == 0) founders = 1;
    std::vector<uint32_t> founder(n), founder_allele(founders);
    std::vector<uint64_t> next_switch(n);
    const double log1m = rho > 0 && rho < 1 ? std::log(1.0 - rho) : 0.0;
    auto gap = [&]() -> uint64_t {  // sites until the next founder switch (geometric)
        if (rho <= 0) return UINT64_MAX / 2;
        if (rho >= 1) return 1;
        double u = rng.uniform();
        if (u <= 0) u = 1e-300;
        return 1 + static_cast<uint64_t>(std::log(u) / log1m);
    };
    for (uint64_t h = 0; h < n; h++) { founder[h] = static_cast<uint32_t>(rng.below(founders)); next_switch[h] = gap(); }
    for (uint64_t s = 0; s < S; s++) {
        double p = 0.05 + 0.9 * rng.uniform();
        for (uint32_t f = 0; f < founders; f++) founder_allele[f] = draw(p);
        for (uint64_t h = 0; h < n; h++) {
            if (next_switch[h] == s) { founder[h] = static_cast<uint32_t>(rng.below(founders)); next_switch[h] = s + gap(); }
            g.set_allele(s, h, founder_allele[founder[h]]);
        }
    }
}

// Per-site allele counts and rank among the alleles that occur.
struct SiteStats {
    std::vector<uint32_t> cnt, rank;
    uint32_t present = 0;
    void compute(const gbwt_synth &g, uint64_t s) {
        const uint64_t A = g.alleles, n = g.haplotypes;
        cnt.assign(A, 0); rank.assign(A, 0);
        if (A == 2) {
            uint64_t ones = 0;
            const uint64_t *row = g.bits.data() + s * g.row_words;
            for (uint64_t w = 0; w < g.row_words; w++) ones += static_cast<uint64_t>(__builtin_popcountll(row[w]));
            cnt[1] = static_cast<uint32_t>(ones); cnt[0] = static_cast<uint32_t>(n - ones);
        } else {
            const uint16_t *row = g.choices.data() + s * n;
            for (uint64_t h = 0; h < n; h++) cnt[row[h]]++;
        }
        present = 0;
        for (uint64_t a = 0; a < A; a++) if (cnt[a]) rank[a] = present++;
    }
};

// Stable counting sort of `ord` by the allele at site s.
void partition(const gbwt_synth &g, uint64_t s, const SiteStats &st, const std::vector<uint32_t> &ord, std::vector<uint32_t> &out,
               std::vector<uint32_t> &cursor) {
    const uint64_t A = g.alleles;
    cursor.assign(A, 0);
    uint32_t acc = 0;
    for (uint64_t a = 0; a < A; a++) { cursor[a] = acc; acc += st.cnt[a]; }
    out.resize(ord.size());
    for (uint32_t h : ord) out[cursor[g.allele(s, h)]++] = h;
}

// Pool of records produced by one sweep: slot (s, k) with k = node id - anchor id (0 anchor, 1 + a = allele a, then the tails).
struct Pool {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> start;  // (S * stride + 1) entries: where the record of a slot begins ...
    std::vector<uint32_t> len;    // ... and how long it is (slots are written in any order)
    void begin(uint64_t slot) { start[slot] = bytes.size(); }
    void end(uint64_t slot) { len[slot] = static_cast<uint32_t>(bytes.size() - start[slot]); }
    void empty(uint64_t slot) { begin(slot); bytes.push_back(0); end(slot); }
};

void forward_sweep(const gbwt_synth &g, Pool &pool) {
    const uint64_t S = g.sites, n = g.haplotypes, A = g.alleles, K = g.chop, W = g.stride();
    pool.start.assign(S * W + 1, 0);
    pool.len.assign(S * W + 1, 0);
    pool.bytes.reserve(S * 24);
    std::vector<uint32_t> ord(n), nxt, cursor;
    for (uint64_t h = 0; h < n; h++) ord[h] = static_cast<uint32_t>(h);
    SiteStats st;
    std::vector<std::pair<uint64_t, uint64_t>> edges;
    std::vector<uint64_t> before(A);
    auto rw = make_record_writer(pool.bytes);
    auto unary = [&](uint64_t slot, uint64_t to, uint64_t offset, uint64_t visits) {   // every visit comes from the one predecessor and goes on to `to`
        pool.begin(slot);
        edges.clear();
        edges.emplace_back(to, offset);
        rw.begin(edges);
        rw.push(0, visits);
        rw.end();
        pool.end(slot);
    };
    for (uint64_t s = 0; s < S; s++) {
        st.compute(g, s);
        const uint64_t base = s * W;
        // anchor, forward orientation: chop - 1 unary pieces, then the piece whose successors are the allele nodes of this site
        for (uint64_t i = 0; i + 1 < K; i++) unary(base + i, 2 * (g.anchor_first(s) + i + 1), 0, n);
        pool.begin(base + K - 1);
        edges.clear();
        for (uint64_t a = 0; a < A; a++) if (st.cnt[a]) edges.emplace_back(2 * g.allele_id(s, static_cast<uint32_t>(a)), 0);
        rw.begin(edges);
        for (uint32_t h : ord) rw.push(st.rank[g.allele(s, h)]);
        rw.end();
        pool.end(base + K - 1);
        partition(g, s, st, ord, nxt, cursor);
        ord.swap(nxt);
        uint64_t acc = 0;
        for (uint64_t a = 0; a < A; a++) { before[a] = acc; acc += st.cnt[a]; }
        // allele a: first piece, then its tails; the last piece has one edge to the next anchor (or the ENDMARKER at the last site) --
        // the next anchor's visits are ordered by predecessor = last piece of the allele, ascending with the allele
        for (uint32_t a = 0; a < A; a++) {
            const uint64_t tails = st.cnt[a] ? g.tails_at(s, a) : 0, first_slot = base + g.allele_slot(a), tail_slot = first_slot + 1;
            for (uint64_t e = tails; e < g.tails_max(a); e++) pool.empty(tail_slot + e);          // tails this site does not use
            if (!st.cnt[a]) { pool.empty(first_slot); continue; }
            const uint64_t out = s + 1 < S ? 2 * g.anchor_first(s + 1) : 0, out_offset = s + 1 < S ? before[a] : 0;
            for (uint64_t piece = 0; piece <= tails; piece++) {          // piece 0 = the first piece
                const uint64_t slot = piece == 0 ? first_slot : tail_slot + piece - 1;
                if (piece == tails) unary(slot, out, out_offset, st.cnt[a]);
                else unary(slot, 2 * g.tail_id(s, a, piece), 0, st.cnt[a]);
            }
        }
    }
}

// Reverse orientation, generated from the last site down; slot index for site s = (S - 1 - s) * stride + k.
void reverse_sweep(const gbwt_synth &g, Pool &pool) {
    const uint64_t S = g.sites, n = g.haplotypes, A = g.alleles, K = g.chop, W = g.stride();
    pool.start.assign(S * W + 1, 0);
    pool.len.assign(S * W + 1, 0);
    pool.bytes.reserve(S * 24);
    std::vector<uint32_t> ord(n), nxt, cursor;
    for (uint64_t h = 0; h < n; h++) ord[h] = static_cast<uint32_t>(h);
    SiteStats st, prev;
    std::vector<std::pair<uint64_t, uint64_t>> edges;
    auto rw = make_record_writer(pool.bytes);
    auto unary = [&](uint64_t slot, uint64_t to, uint64_t offset, uint64_t visits) {
        pool.begin(slot);
        edges.clear();
        edges.emplace_back(to, offset);
        rw.begin(edges);
        rw.push(0, visits);
        rw.end();
        pool.end(slot);
    };
    for (uint64_t s = S; s-- > 0;) {
        st.compute(g, s);
        const uint64_t base = (S - 1 - s) * W;
        partition(g, s, st, ord, nxt, cursor);
        ord.swap(nxt);  // order of the visits in the records of the reverse anchor chain
        // anchor, reverse orientation: entered at its last piece, left at its first, whose successors are the reverse last pieces of
        // the alleles of site s - 1
        pool.begin(base);
        edges.clear();
        if (s > 0) {
            prev.compute(g, s - 1);
            for (uint64_t a = 0; a < A; a++) if (prev.cnt[a]) edges.emplace_back(2 * g.last_id(s - 1, static_cast<uint32_t>(a)) + 1, 0);
            rw.begin(edges);
            for (uint32_t h : ord) rw.push(prev.rank[g.allele(s - 1, h)]);
            rw.end();
        } else {
            edges.emplace_back(0, 0);
            rw.begin(edges);
            rw.push(0, n);
            rw.end();
        }
        pool.end(base);
        for (uint64_t i = 1; i < K; i++) unary(base + i, 2 * (g.anchor_first(s) + i - 1) + 1, 0, n);
        // alleles, reverse: the first piece has one edge to the reverse last piece of this site's anchor, every tail one back towards it
        uint64_t before = 0;
        for (uint32_t a = 0; a < A; a++) {
            const uint64_t tails = st.cnt[a] ? g.tails_at(s, a) : 0, first_slot = base + g.allele_slot(a), tail_slot = first_slot + 1;
            for (uint64_t e = tails; e < g.tails_max(a); e++) pool.empty(tail_slot + e);
            if (!st.cnt[a]) { pool.empty(first_slot); continue; }
            unary(first_slot, 2 * g.anchor_last(s) + 1, before, st.cnt[a]);
            for (uint64_t e = 0; e < tails; e++) unary(tail_slot + e, 2 * (e == 0 ? g.allele_id(s, a) : g.tail_id(s, a, e - 1)) + 1, 0, st.cnt[a]);
            before += st.cnt[a];
        }
    }
}

void add_string(gbwt_hip::Strings &s, const std::string &x) {
    s.bytes.insert(s.bytes.end(), x.begin(), x.end());
    s.offsets.push_back(s.bytes.size());
}

void build_chain(gbwt_synth &g, uint64_t seed) {
    const uint64_t S = g.sites, n = g.haplotypes, A = g.alleles, W = g.stride();
    HostIndex &ix = g.index;
    Pool fwd, rev;
    std::thread t([&] { reverse_sweep(g, rev); });
    forward_sweep(g, fwd);
    t.join();

    // endmarker record: sequence 2h starts at the first anchor, sequence 2h + 1 at the reverse of the last allele's last node
    SiteStats last;
    last.compute(g, S - 1);
    std::vector<std::pair<uint64_t, uint64_t>> edges;
    edges.emplace_back(2 * g.anchor_first(0), 0);
    for (uint64_t a = 0; a < A; a++) if (last.cnt[a]) edges.emplace_back(2 * g.last_id(S - 1, static_cast<uint32_t>(a)) + 1, 0);
    ix.data.clear();
    ix.data.reserve(fwd.bytes.size() + rev.bytes.size() + 4 * n + 64);
    ix.starts.clear();
    ix.starts.reserve(2 * S * W + 2);
    ix.starts.push_back(0);
    {
        auto rw = make_record_writer(ix.data);
        rw.begin(edges);
        for (uint64_t h = 0; h < n; h++) { rw.push(0); rw.push(1 + last.rank[g.allele(S - 1, h)]); }
        rw.end();
    }
    // interleave: node id ascending, forward record then reverse record
    for (uint64_t s = 0; s < S; s++) {
        for (uint64_t k = 0; k < W; k++) {
            uint64_t fs = s * W + k, rs = (S - 1 - s) * W + k;
            ix.starts.push_back(ix.data.size());
            ix.data.insert(ix.data.end(), fwd.bytes.begin() + fwd.start[fs], fwd.bytes.begin() + fwd.start[fs] + fwd.len[fs]);
            ix.starts.push_back(ix.data.size());
            ix.data.insert(ix.data.end(), rev.bytes.begin() + rev.start[rs], rev.bytes.begin() + rev.start[rs] + rev.len[rs]);
        }
    }
    // alphabet_size = largest visited GBWT node + 1: drop the records of unused trailing allele nodes
    uint32_t top = 0;
    for (uint64_t a = 0; a < A; a++) if (last.cnt[a]) top = static_cast<uint32_t>(a);
    const uint64_t max_node = 2 * g.last_id(S - 1, top) + 1;
    if (max_node < ix.starts.size()) {   // records 0 .. max_node - 1 (record r <-> node r + 1)
        ix.data.resize(ix.starts[max_node]);
        ix.starts.resize(max_node);
    }
    ix.starts.push_back(ix.data.size());
    ix.sequences = 2 * n;
    ix.size = 0;   // set below, once the insertion visits are counted
    ix.alphabet_offset = 1;
    ix.alphabet_size = max_node + 1;
    ix.bidirectional = true;
    ix.tags.emplace_back("source", "gbwt_rs_amd/synth");
    ix.gbz_tags.emplace_back("source", "gbwt_rs_amd/synth");

    // metadata: path 0 generic, the rest haplotypes of samples s0, s1, ...
    ix.has_metadata = true;
    ix.metadata_flags = 7;
    const uint64_t n_samples = (n > 1 ? (n - 1 + 1) / 2 : 0);
    for (uint64_t k = 0; k < n_samples; k++) add_string(ix.sample_names, "s" + std::to_string(k));
    add_string(ix.sample_names, "_gbwt_ref");
    add_string(ix.contig_names, "chr1");
    ix.sample_count = n_samples + 1; ix.haplotype_count = n; ix.contig_count = 1;
    ix.path_names.resize(n);
    ix.path_names[0] = PathName{static_cast<uint32_t>(n_samples), 0, 0, 0};
    for (uint64_t h = 1; h < n; h++) ix.path_names[h] = PathName{static_cast<uint32_t>((h - 1) / 2), 0, static_cast<uint32_t>((h - 1) % 2 + 1), 0};

    // graph: 1 bp labels for nodes that occur, empty labels for allele nodes nobody uses
    ix.is_gbz = true; ix.has_translation = false;
    Rng rng(seed ^ 0xACDCACDCull);
    SiteStats st;
    uint64_t real = 0, visits = 0;
    ix.sequences_labels.bytes.reserve(g.label_mode == 0 ? S * W : S * W * 40);
    ix.sequences_labels.offsets.reserve(S * W + 1);
    const uint64_t K = g.chop;
    for (uint64_t s = 0; s < S; s++) {
        st.compute(g, s);
        visits += n * K;                                             // the anchor's pieces
        for (uint32_t a = 0; a < A; a++) visits += st.cnt[a] * (1 + g.tails_at(s, a));
        for (uint64_t k = 0; k < W; k++) {
            if (s * W + k + 1 > g.last_id(S - 1, top)) break;  // ids past the largest visited node
            bool exists = k < K;                                     // anchor pieces
            if (!exists) {                                           // which allele, which piece (0 = the first one)
                const uint64_t q = k - K, wide = K * (1 + g.extra);
                const uint32_t a = q < K ? 0u : static_cast<uint32_t>(1 + (q - K) / wide);
                const uint64_t piece = q < K ? q : (q - K) % wide;
                exists = a < A && st.cnt[a] != 0 && piece <= g.tails_at(s, a);
            }
            if (exists && g.label_mode == 0) { ix.sequences_labels.bytes.push_back("ACGT"[rng.next() >> 62]); real++; }
            else if (exists) {
                // lengths as a chopped minigraph-cactus graph has them: the stretches between variants are tens of bases, one in twelve a
                // full 1 024 bp piece of a long segment; most alleles are single bases, the rest short indel alleles; inserted nodes
                // a few dozen bases
                const double u = rng.uniform();
                uint64_t len;
                if (k < K) len = u < 0.08 ? 1024 : 1 + static_cast<uint64_t>(-40.0 * std::log(1.0 - rng.uniform()));
                else if ((k - K) < K || ((k - K - K) % (K * (1 + g.extra))) < K) len = u < 0.85 ? 1 : 1 + static_cast<uint64_t>(-6.0 * std::log(1.0 - rng.uniform()));
                else len = 1 + static_cast<uint64_t>(-20.0 * std::log(1.0 - rng.uniform()));
                len = std::min<uint64_t>(len, 1024);
                uint64_t word = 0;
                for (uint64_t i = 0; i < len; i++) {
                    if ((i & 31) == 0) word = rng.next();
                    ix.sequences_labels.bytes.push_back("ACGT"[word & 3]);
                    word >>= 2;
                }
                real++;
            }
            ix.sequences_labels.offsets.push_back(ix.sequences_labels.bytes.size());
        }
    }
    ix.graph_nodes = real;
    ix.size = 2 * visits + 2 * n;
}

// ---- general path sets: brute-force reverse-prefix sort ------------------------------------------
void build_from_paths(gbwt_synth &g, const uint64_t *offsets, const uint64_t *nodes, uint64_t n_paths, bool bidirectional) {
    std::vector<std::vector<uint64_t>> seqs;
    for (uint64_t p = 0; p < n_paths; p++) {
        std::vector<uint64_t> f(nodes + offsets[p], nodes + offsets[p + 1]);
        seqs.push_back(f);
        if (bidirectional) {
            std::vector<uint64_t> r(f.rbegin(), f.rend());
            for (auto &x : r) x ^= 1;
            seqs.push_back(r);
        }
    }
    g.path_offsets.assign(1, 0);
    for (uint64_t p = 0; p < n_paths; p++) {
        for (uint64_t k = offsets[p]; k < offsets[p + 1]; k++) g.path_nodes.push_back(static_cast<uint32_t>(nodes[k]));
        g.path_offsets.push_back(g.path_nodes.size());
    }
    HostIndex &ix = g.index;
    uint64_t min_node = UINT64_MAX, max_node = 0, total = 0;
    for (auto &s : seqs) for (uint64_t v : s) { min_node = std::min(min_node, v); max_node = std::max(max_node, v); total++; }
    ix.sequences = seqs.size();
    ix.size = total + seqs.size();
    ix.bidirectional = bidirectional;
    if (total == 0) {  // only empty sequences: endmarker record alone
        ix.alphabet_offset = 0; ix.alphabet_size = 1;
    } else {
        ix.alphabet_offset = min_node - 1; ix.alphabet_size = max_node + 1;
    }
    struct Visit { uint32_t seq, pos; };
    const uint64_t n_records = ix.alphabet_size - ix.alphabet_offset;
    std::vector<std::vector<Visit>> visits(n_records);
    for (uint32_t i = 0; i < seqs.size(); i++)
        for (uint32_t j = 0; j < seqs[i].size(); j++) visits[seqs[i][j] - ix.alphabet_offset].push_back(Visit{i, j});
    auto less = [&](const Visit &a, const Visit &b) {  // compare reverse prefixes, ENDMARKER < every node, ties by sequence id
        uint32_t ja = a.pos, jb = b.pos;
        while (ja > 0 && jb > 0) {
            uint64_t x = seqs[a.seq][ja - 1], y = seqs[b.seq][jb - 1];
            if (x != y) return x < y;
            ja--; jb--;
        }
        if (ja == 0 && jb == 0) return a.seq < b.seq;
        return ja == 0;
    };
    for (auto &v : visits) std::stable_sort(v.begin(), v.end(), less);
    auto pred_of = [&](const Visit &v) -> uint64_t { return v.pos == 0 ? 0 : seqs[v.seq][v.pos - 1]; };
    auto count_smaller_preds = [&](uint64_t w, uint64_t v) -> uint64_t {  // visits of w whose predecessor < v
        uint64_t c = 0;
        for (auto &x : visits[w - ix.alphabet_offset]) if (pred_of(x) < v) c++;
        return c;
    };
    ix.data.clear(); ix.starts.clear();
    auto rw = make_record_writer(ix.data);
    for (uint64_t r = 0; r < n_records; r++) {
        ix.starts.push_back(ix.data.size());
        std::vector<uint64_t> succ;
        uint64_t v = r == 0 ? 0 : r + ix.alphabet_offset;
        if (r == 0) {
            for (auto &s : seqs) succ.push_back(s.empty() ? 0 : s[0]);
        } else {
            for (auto &x : visits[r]) succ.push_back(x.pos + 1 < seqs[x.seq].size() ? seqs[x.seq][x.pos + 1] : 0);
        }
        if (succ.empty()) { ix.data.push_back(0); continue; }
        std::vector<uint64_t> distinct(succ);
        std::sort(distinct.begin(), distinct.end());
        distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        std::vector<std::pair<uint64_t, uint64_t>> edges;
        for (uint64_t w : distinct) edges.emplace_back(w, (w == 0 || r == 0) ? 0 : count_smaller_preds(w, v));
        rw.begin(edges);
        for (uint64_t w : succ) rw.push(static_cast<uint64_t>(std::lower_bound(distinct.begin(), distinct.end(), w) - distinct.begin()));
        rw.end();
    }
    ix.starts.push_back(ix.data.size());
    ix.tags.emplace_back("source", "gbwt_rs_amd/synth");
}

}  // namespace

extern "C" {

gbwt_synth *gbwt_synth_chain(uint64_t sites, uint64_t haplotypes, uint32_t alleles, uint32_t model, uint32_t founders,
                             double switch_rate, double zipf, uint64_t seed) {
    return gbwt_synth_chain_indel(sites, haplotypes, alleles, model, founders, switch_rate, zipf, seed, 0, 1);
}

gbwt_synth *gbwt_synth_chain_indel(uint64_t sites, uint64_t haplotypes, uint32_t alleles, uint32_t model, uint32_t founders,
                                   double switch_rate, double zipf, uint64_t seed, uint32_t extra, uint32_t indel_every) {
    return gbwt_synth_chain_chopped(sites, haplotypes, alleles, model, founders, switch_rate, zipf, seed, extra, indel_every, 1);
}

gbwt_synth *gbwt_synth_chain_chopped(uint64_t sites, uint64_t haplotypes, uint32_t alleles, uint32_t model, uint32_t founders,
                                     double switch_rate, double zipf, uint64_t seed, uint32_t extra, uint32_t indel_every, uint32_t chop) {
    return gbwt_synth_chain_labeled(sites, haplotypes, alleles, model, founders, switch_rate, zipf, seed, extra, indel_every, chop, 0);
}

gbwt_synth *gbwt_synth_chain_labeled(uint64_t sites, uint64_t haplotypes, uint32_t alleles, uint32_t model, uint32_t founders,
                                     double switch_rate, double zipf, uint64_t seed, uint32_t extra, uint32_t indel_every, uint32_t chop, uint32_t label_mode) {
    if (sites == 0 || haplotypes == 0 || alleles < 2 || alleles > 60000 || extra > 64 || indel_every == 0 || chop == 0 || chop > 64 || label_mode > 1) return nullptr;
    gbwt_synth *g = new gbwt_synth;
    g->sites = sites; g->haplotypes = haplotypes; g->alleles = alleles; g->extra = extra; g->indel_every = indel_every; g->chop = chop; g->label_mode = label_mode;
    if (alleles == 2) { g->row_words = (haplotypes + 63) / 64; g->bits.assign(sites * g->row_words, 0); }
    else g->choices.assign(sites * haplotypes, 0);
    draw_alleles(*g, model, founders, switch_rate, zipf, seed);
    build_chain(*g, seed);
    return g;
}

gbwt_synth *gbwt_synth_from_paths(const uint64_t *offsets, const uint64_t *nodes, uint64_t n_paths, int bidirectional) {
    gbwt_synth *g = new gbwt_synth;
    build_from_paths(*g, offsets, nodes, n_paths, bidirectional != 0);
    return g;
}

int gbwt_synth_attach_gbz(gbwt_synth *g, const uint64_t *segment_starts, uint64_t n_segments, uint64_t seed) {
    gbwt_hip::HostIndex &ix = g->index;
    if (!ix.bidirectional || ix.alphabet_offset != 1) return 1;   // node id v <-> label v - 1, as the translation assumes
    const uint64_t n = ix.sequences / 2;
    ix.has_metadata = true;
    ix.metadata_flags = 7;
    ix.sample_names = gbwt_hip::Strings(); ix.contig_names = gbwt_hip::Strings();
    const uint64_t n_samples = (n > 1 ? n / 2 : 0);
    for (uint64_t k = 0; k < n_samples; k++) add_string(ix.sample_names, "s" + std::to_string(k));
    add_string(ix.sample_names, "_gbwt_ref");
    add_string(ix.contig_names, "chr1");
    ix.sample_count = n_samples + 1; ix.haplotype_count = n; ix.contig_count = 1;
    ix.path_names.resize(n);
    if (n > 0) ix.path_names[0] = gbwt_hip::PathName{static_cast<uint32_t>(n_samples), 0, 0, 0};
    for (uint64_t h = 1; h < n; h++) ix.path_names[h] = gbwt_hip::PathName{static_cast<uint32_t>((h - 1) / 2), 0, static_cast<uint32_t>((h - 1) % 2 + 1), 0};
    // labels of 1-3 bases for the nodes that have a record, empty labels for the rest
    ix.is_gbz = true;
    const uint64_t first = ix.alphabet_offset + 1, potential = ix.alphabet_size > first ? (ix.alphabet_size - first + 1) / 2 : 0;
    Rng rng(seed ^ 0x5E65E65Eull);
    ix.sequences_labels = gbwt_hip::Strings();
    uint64_t real = 0;
    for (uint64_t q = 0; q < potential; q++) {
        const uint64_t rec = 2 * q + first - ix.alphabet_offset;
        const bool exists = rec < ix.records() && ix.starts[rec + 1] > ix.starts[rec] && ix.data[ix.starts[rec]] != 0;
        if (exists) {
            real++;
            const uint64_t len = 1 + (rng.next() >> 62) % 3;
            for (uint64_t k = 0; k < len; k++) ix.sequences_labels.bytes.push_back("ACGT"[rng.next() >> 62]);
        }
        ix.sequences_labels.offsets.push_back(ix.sequences_labels.bytes.size());
    }
    ix.graph_nodes = real;
    ix.gbz_tags.clear();
    ix.gbz_tags.emplace_back("source", "gbwt_rs_amd/synth");
    ix.has_translation = n_segments > 0;
    ix.segment_names = gbwt_hip::Strings();
    ix.segment_starts.clear();
    ix.mapping_len = 0;
    if (n_segments > 0) {
        for (uint64_t k = 0; k < n_segments; k++) {
            if (segment_starts[k] == 0 || segment_starts[k] > potential || (k > 0 && segment_starts[k] <= segment_starts[k - 1])) return 2;
            ix.segment_starts.push_back(segment_starts[k]);
            add_string(ix.segment_names, "seg" + std::to_string(segment_starts[k]));
        }
        ix.mapping_len = potential + 1;
    }
    return 0;
}

gbwt_synth *gbwt_synth_merge(const gbwt_synth *const *parts, uint64_t n_parts, const uint32_t *path_names, const char *const *sample_names, uint64_t n_samples,
                             const char *const *contig_names, uint64_t n_contigs, uint64_t haplotype_count) {
    if (n_parts == 0) return nullptr;
    uint64_t total_paths = 0;
    for (uint64_t k = 0; k < n_parts; k++) {
        const gbwt_synth *p = parts[k];
        if (!p || p->sites == 0 || !p->index.bidirectional || p->index.alphabet_offset != 1 || p->index.alphabet_size % 2 != 0 || !p->index.is_gbz) return nullptr;
        total_paths += p->haplotypes;
    }
    gbwt_synth *g = new gbwt_synth;
    HostIndex &ix = g->index;
    ix.alphabet_offset = 1; ix.bidirectional = true;
    ix.sequences = 2 * total_paths;
    ix.tags.emplace_back("source", "gbwt_rs_amd/synth");
    ix.gbz_tags.emplace_back("source", "gbwt_rs_amd/synth");
    // node ids of part k are shifted by the node ids of the parts before it; GBWT nodes by twice that
    uint64_t id_shift = 0;
    std::vector<std::pair<uint64_t, uint64_t>> edges, end_edges;
    std::vector<uint8_t> end_runs;          // (edge rank, run length) pairs of the merged endmarker record, re-encoded below
    std::vector<std::pair<uint64_t, uint64_t>> end_run_list;
    auto get_varint = [](const uint8_t *&p, const uint8_t *end, uint64_t &v) {
        v = 0;
        unsigned shift = 0;
        while (p < end) { const uint8_t b = *p++; v += static_cast<uint64_t>(b & 0x7F) << shift; shift += 7; if (!(b & 0x80)) return true; }
        return false;
    };
    std::vector<uint8_t> body;              // records 1 .. of the merged index
    std::vector<uint64_t> body_starts;
    for (uint64_t k = 0; k < n_parts; k++) {
        const gbwt_synth &p = *parts[k];
        const HostIndex &px = p.index;
        const uint64_t node_shift = 2 * id_shift;
        g->part_first_path.push_back(ix.path_names.size());
        g->part_shift.push_back(id_shift);
        // its endmarker record: edges shifted, runs decoded (they are re-encoded against the merged edge list)
        {
            const uint8_t *q = px.data.data() + px.starts[0], *end = px.data.data() + px.starts[1];
            uint64_t sigma = 0, node = 0;
            get_varint(q, end, sigma);
            const uint64_t rank_base = end_edges.size();
            for (uint64_t e = 0; e < sigma; e++) {
                uint64_t delta = 0, off = 0;
                get_varint(q, end, delta); get_varint(q, end, off);
                node += delta;
                end_edges.emplace_back(node == 0 ? 0 : node + node_shift, 0);
            }
            const uint64_t threshold = sigma >= 255 ? 0 : 256 / sigma;
            while (q < end) {
                uint64_t value = 0, len = 0;
                if (sigma >= 255) { get_varint(q, end, value); get_varint(q, end, len); len++; }
                else { const uint64_t b = *q++; value = b % sigma; len = b / sigma + 1; if (len == threshold) { uint64_t x = 0; get_varint(q, end, x); len += x; } }
                end_run_list.emplace_back(rank_base + value, len);
            }
        }
        // its other records: the nodes of the edge list shifted, the run stream as it is
        for (uint64_t r = 1; r < px.records(); r++) {
            body_starts.push_back(body.size());
            const uint8_t *q = px.data.data() + px.starts[r], *end = px.data.data() + px.starts[r + 1];
            uint64_t sigma = 0, node = 0;
            if (!get_varint(q, end, sigma) || sigma == 0) { body.push_back(0); continue; }
            edges.clear();
            for (uint64_t e = 0; e < sigma; e++) {
                uint64_t delta = 0, off = 0;
                get_varint(q, end, delta); get_varint(q, end, off);
                node += delta;
                edges.emplace_back(node == 0 ? 0 : node + node_shift, off);
            }
            put_varint(body, sigma);
            uint64_t prev = 0;
            for (auto &e : edges) { put_varint(body, e.first - prev); put_varint(body, e.second); prev = e.first; }
            body.insert(body.end(), q, end);
        }
        // labels, truth
        for (uint64_t q = 0; q < px.sequences_labels.size(); q++) {
            ix.sequences_labels.bytes.insert(ix.sequences_labels.bytes.end(), px.sequences_labels.bytes.begin() + px.sequences_labels.offsets[q],
                                             px.sequences_labels.bytes.begin() + px.sequences_labels.offsets[q + 1]);
            ix.sequences_labels.offsets.push_back(ix.sequences_labels.bytes.size());
        }
        const uint64_t ids = (px.alphabet_size - 2) / 2;           // node ids 1 .. ids
        for (uint64_t q = px.sequences_labels.size(); q < ids; q++) ix.sequences_labels.offsets.push_back(ix.sequences_labels.bytes.size());
        ix.graph_nodes += px.graph_nodes;
        ix.size += px.size;
        for (uint64_t h = 0; h < p.haplotypes; h++) {
            const uint32_t *f = path_names + 4 * ix.path_names.size();
            ix.path_names.push_back(PathName{f[0], f[1], f[2], f[3]});
        }
        gbwt_synth truth;                                          // the allele matrix only
        truth.sites = p.sites; truth.haplotypes = p.haplotypes; truth.alleles = p.alleles; truth.extra = p.extra; truth.indel_every = p.indel_every;
        truth.chop = p.chop; truth.bits = p.bits; truth.row_words = p.row_words; truth.choices = p.choices;
        g->parts.push_back(std::move(truth));
        id_shift += ids;
    }
    // the merged endmarker record: a consistent GBWT lists an edge once, and the parts' nodes are disjoint and ascending
    ix.data.clear();
    ix.starts.assign(1, 0);
    {
        auto rw = make_record_writer(ix.data);
        rw.begin(end_edges);
        for (auto &r : end_run_list) rw.push(r.first, r.second);
        rw.end();
    }
    const uint64_t base = ix.data.size();
    for (uint64_t b : body_starts) ix.starts.push_back(base + b);
    ix.data.insert(ix.data.end(), body.begin(), body.end());
    ix.starts.push_back(ix.data.size());
    ix.alphabet_size = 2 * id_shift + 2;
    ix.is_gbz = true; ix.has_translation = false;
    ix.has_metadata = true; ix.metadata_flags = 7;
    for (uint64_t k = 0; k < n_samples; k++) add_string(ix.sample_names, sample_names[k]);
    for (uint64_t k = 0; k < n_contigs; k++) add_string(ix.contig_names, contig_names[k]);
    ix.sample_count = n_samples; ix.contig_count = n_contigs; ix.haplotype_count = haplotype_count;
    g->haplotypes = total_paths;
    return g;
}

gbwt_synth *gbwt_synth_from_file(const char *path, char *err, uint64_t errlen) {
    gbwt_synth *g = new gbwt_synth;
    try {
        g->index = gbwt_hip::load_index_file(path);
    } catch (const std::exception &e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        delete g;
        return nullptr;
    }
    return g;
}

void gbwt_synth_free(gbwt_synth *s) { delete s; }

const uint8_t *gbwt_synth_data(const gbwt_synth *s, uint64_t *len) { *len = s->index.data.size(); return s->index.data.data(); }

const uint64_t *gbwt_synth_starts(const gbwt_synth *s, uint64_t *n_records) { *n_records = s->index.records(); return s->index.starts.data(); }

void gbwt_synth_header(const gbwt_synth *s, uint64_t out[8]) {
    out[0] = s->index.sequences; out[1] = s->index.size; out[2] = s->index.alphabet_offset; out[3] = s->index.alphabet_size;
    out[4] = s->index.bidirectional; out[5] = s->index.bidirectional ? s->index.sequences / 2 : s->index.sequences;
    out[6] = s->sites; out[7] = s->alleles;
}

int gbwt_synth_save(const gbwt_synth *s, const char *path, int as_gbz) {
    try {
        gbwt_hip::save_index_file(s->index, path, as_gbz != 0);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "gbwt_synth_save: %s\n", e.what());
        return 1;
    }
}

void gbwt_synth_set_tag(gbwt_synth *s, const char *key, const char *value) {
    if (!s || !key || !value) return;
    for (auto &kv : s->index.tags) if (kv.first == key) { kv.second = value; return; }
    s->index.tags.emplace_back(key, value);
}

static bool merged_part(const gbwt_synth *s, uint64_t path_id, uint64_t &part, uint64_t &local) {
    if (s->parts.empty() || path_id >= s->haplotypes) return false;
    part = std::upper_bound(s->part_first_path.begin(), s->part_first_path.end(), path_id) - s->part_first_path.begin() - 1;
    local = path_id - s->part_first_path[part];
    return true;
}

uint64_t gbwt_synth_path(const gbwt_synth *s, uint64_t path_id, uint32_t *out, uint64_t cap) {
    uint64_t part = 0, local = 0;
    if (merged_part(s, path_id, part, local)) {
        const uint64_t len = gbwt_synth_path(&s->parts[part], local, out, cap);
        for (uint64_t k = 0; k < len && k < cap; k++) out[k] += static_cast<uint32_t>(2 * s->part_shift[part]);
        return len;
    }
    if (s->sites) {
        if (path_id >= s->haplotypes) return 0;
        uint64_t len = 0;
        auto put = [&](uint64_t node) { if (len < cap) out[len] = static_cast<uint32_t>(2 * node); len++; };
        for (uint64_t site = 0; site < s->sites; site++) {
            const uint32_t a = s->allele(site, path_id);
            for (uint64_t i = 0; i < s->chop; i++) put(s->anchor_first(site) + i);
            put(s->allele_id(site, a));
            for (uint64_t e = 0; e < s->tails_at(site, a); e++) put(s->tail_id(site, a, e));
        }
        return len;
    }
    if (path_id + 1 >= s->path_offsets.size()) return 0;
    uint64_t a = s->path_offsets[path_id], b = s->path_offsets[path_id + 1];
    for (uint64_t k = a; k < b && k - a < cap; k++) out[k - a] = s->path_nodes[k];
    return b - a;
}

// nodes, decimal digits of their ids and summed label lengths of a path: what its GFA line is made of
static void add_node_stats(const gbwt_synth *s, uint64_t id, uint64_t out[3]) {
    out[0]++;
    uint64_t digits = 1;
    for (uint64_t v = id; v >= 10; v /= 10) digits++;
    out[1] += digits;
    const gbwt_hip::Strings &l = s->index.sequences_labels;
    if (id >= 1 && id <= l.size()) out[2] += l.offsets[id] - l.offsets[id - 1];
}

void gbwt_synth_path_text_stats(const gbwt_synth *s, uint64_t path_id, uint64_t out[3]) {
    out[0] = out[1] = out[2] = 0;
    uint64_t part = 0, local = 0;
    const gbwt_synth *chain = s;
    uint64_t shift = 0;
    if (merged_part(s, path_id, part, local)) { chain = &s->parts[part]; shift = s->part_shift[part]; path_id = local; }
    if (chain->sites) {
        if (path_id >= chain->haplotypes) return;
        for (uint64_t site = 0; site < chain->sites; site++) {
            const uint32_t a = chain->allele(site, path_id);
            for (uint64_t i = 0; i < chain->chop; i++) add_node_stats(s, shift + chain->anchor_first(site) + i, out);
            add_node_stats(s, shift + chain->allele_id(site, a), out);
            for (uint64_t e = 0; e < chain->tails_at(site, a); e++) add_node_stats(s, shift + chain->tail_id(site, a, e), out);
        }
        return;
    }
    if (path_id + 1 >= s->path_offsets.size()) return;
    for (uint64_t k = s->path_offsets[path_id]; k < s->path_offsets[path_id + 1]; k++) add_node_stats(s, s->path_nodes[k] / 2, out);
}

uint64_t gbwt_synth_path_checksum(const gbwt_synth *s, uint64_t path_id) {
    uint64_t sum = 0, part = 0, local = 0;
    if (merged_part(s, path_id, part, local))
        return gbwt_synth_path_checksum(&s->parts[part], local) + 2 * s->part_shift[part] * gbwt_synth_path(&s->parts[part], local, nullptr, 0);
    if (s->sites) {
        if (path_id >= s->haplotypes) return 0;
        for (uint64_t site = 0; site < s->sites; site++) {
            const uint32_t a = s->allele(site, path_id);
            for (uint64_t i = 0; i < s->chop; i++) sum += 2 * (s->anchor_first(site) + i);
            sum += 2 * s->allele_id(site, a);
            for (uint64_t e = 0; e < s->tails_at(site, a); e++) sum += 2 * s->tail_id(site, a, e);
        }
        return sum;
    }
    if (path_id + 1 >= s->path_offsets.size()) return 0;
    for (uint64_t k = s->path_offsets[path_id]; k < s->path_offsets[path_id + 1]; k++) sum += s->path_nodes[k];
    return sum;
}

}  // extern "C"
