/*
 * gbz_oracle.c -- CPU ORACLE (test infrastructure, not product): simple-sds file reader for
 * .gbwt / .gbz, metadata + graph containers, and the gbunzip GFA text.
 * Citations are file:line into /root/reference.  On-disk layout: SURVEY.md Appendix A.
 */
#define _GNU_SOURCE
#include "gbwt_oracle.h"
#include "oracle_internal.h"

#include <ctype.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* reader over the file as little-endian u64 elements                                           */

typedef struct { const uint64_t *w; uint64_t n, pos; int fail; char *err; size_t errlen; } reader;

static void rd_fail(reader *r, const char *fmt, ...) {
    if (r->fail) return;
    r->fail = 1;
    if (r->err && r->errlen) {
        va_list ap; va_start(ap, fmt);
        vsnprintf(r->err, r->errlen, fmt, ap);
        va_end(ap);
    }
}

static uint64_t rd_word(reader *r) {
    if (r->fail || r->pos >= r->n) { rd_fail(r, "unexpected end of file at element %llu", (unsigned long long)r->pos); return 0; }
    return r->w[r->pos++];
}

static const uint64_t *rd_words(reader *r, uint64_t count) {
    if (r->fail || count > r->n - r->pos) { rd_fail(r, "unexpected end of file (need %llu elements)", (unsigned long long)count); return NULL; }
    const uint64_t *p = r->w + r->pos;
    r->pos += count;
    return p;
}

/* Vec<u8>: len in bytes, data padded to 8 */
static const uint8_t *rd_vec_u8(reader *r, uint64_t *len) {
    *len = rd_word(r);
    return (const uint8_t *)rd_words(r, (*len + 7) / 8);
}

/* Option<T>: size in elements (0 = None); skip */
static void rd_skip_option(reader *r) {
    uint64_t sz = rd_word(r);
    rd_words(r, sz);
}

/* RawVector: len_bits, Vec<u64> */
static const uint64_t *rd_rawvector(reader *r, uint64_t *len_bits, uint64_t *n_words) {
    *len_bits = rd_word(r);
    *n_words = rd_word(r);
    if (!r->fail && *n_words != (*len_bits + 63) / 64) rd_fail(r, "RawVector: word count does not match length");
    return rd_words(r, *n_words);
}

typedef struct { uint64_t len, width; const uint64_t *words; uint64_t n_words; } intvec;

static intvec rd_intvector(reader *r) {
    intvec v; memset(&v, 0, sizeof(v));
    v.len = rd_word(r); v.width = rd_word(r);
    uint64_t bits;
    v.words = rd_rawvector(r, &bits, &v.n_words);
    if (!r->fail && (v.width == 0 || v.width > 64 || bits != v.len * v.width)) rd_fail(r, "IntVector: invalid width / length");
    return v;
}

static inline uint64_t intvec_get(const intvec *v, uint64_t k) {
    uint64_t w = v->width, bit = k * w, word = bit >> 6, off = bit & 63;
    uint64_t x = v->words[word] >> off;
    if (off + w > 64) x |= v->words[word + 1] << (64 - off);
    return (w == 64) ? x : (x & ((1ULL << w) - 1));
}

/* SparseVector: len, BitVector high {ones, RawVector, 3 x Option support}, IntVector low */
static void rd_sparse(reader *r, go_sparse *sv) {
    memset(sv, 0, sizeof(*sv));
    sv->universe = rd_word(r);
    sv->ones = rd_word(r);
    uint64_t bits, n_words;
    const uint64_t *hw = rd_rawvector(r, &bits, &n_words);
    rd_skip_option(r); rd_skip_option(r); rd_skip_option(r);
    intvec low = rd_intvector(r);
    if (r->fail) return;
    if (low.len != sv->ones) { rd_fail(r, "SparseVector: low length does not match the number of ones"); return; }
    sv->high_bits = bits;
    sv->high = (uint64_t *)calloc(n_words + 2, 8);
    memcpy(sv->high, hw, n_words * 8);
    sv->low_width = low.width; sv->low_len = low.len;
    sv->low = (uint64_t *)calloc(low.n_words + 2, 8);
    memcpy(sv->low, low.words, low.n_words * 8);
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < n_words; i++) cnt += (uint64_t)__builtin_popcountll(sv->high[i]);
    if (cnt != sv->ones) { rd_fail(r, "SparseVector: high bitvector does not have the declared number of ones"); go_sparse_free(sv); return; }
    go_sparse_build_support(sv);
}

/* all values of a sparse vector (one_iter) */
static uint64_t *sparse_values(const go_sparse *sv) {
    uint64_t *vals = (uint64_t *)malloc((sv->ones + 1) * sizeof(uint64_t));
    uint64_t pos = 0;
    for (uint64_t i = 0; i < sv->ones; i++) vals[i] = (i == 0) ? go_sparse_select(sv, 0, &pos) : go_sparse_next(sv, i - 1, &pos);
    return vals;
}

void go_strings_free(go_strings *s) { free(s->offsets); free(s->bytes); memset(s, 0, sizeof(*s)); }

static void strings_finish(reader *r, go_strings *out, go_sparse *sv, uint8_t *bytes, uint64_t total) {
    out->n = sv->ones; out->total = total; out->bytes = bytes;
    out->offsets = sparse_values(sv);
    out->offsets[out->n] = total;
    if (out->n > 0 && out->offsets[0] != 0) rd_fail(r, "StringArray: First string does not start at offset 0");
    go_sparse_free(sv);
}

/* StringArray::load, src/support.rs:601-647 (packed form) */
static void rd_strings(reader *r, go_strings *out) {
    memset(out, 0, sizeof(*out));
    go_sparse sv;
    rd_sparse(r, &sv);
    uint64_t alen;
    const uint8_t *alphabet = rd_vec_u8(r, &alen);
    intvec packed = rd_intvector(r);
    if (r->fail) { go_sparse_free(&sv); return; }
    uint8_t *bytes = (uint8_t *)malloc(packed.len + 1);
    for (uint64_t i = 0; i < packed.len; i++) {
        uint64_t x = intvec_get(&packed, i);
        if (x >= alen) { rd_fail(r, "StringArray: packed character outside the alphabet"); free(bytes); go_sparse_free(&sv); return; }
        bytes[i] = alphabet[x];
    }
    strings_finish(r, out, &sv, bytes, packed.len);
}

typedef size_t (*zstd_decompress_fn)(void *, size_t, const void *, size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);

/* StringArray::decompress, src/support.rs:543-571 (zstd form, GBZ graph version >= 4) */
static void rd_strings_zstd(reader *r, go_strings *out) {
    memset(out, 0, sizeof(*out));
    go_sparse sv;
    rd_sparse(r, &sv);
    uint64_t total = rd_word(r);
    uint64_t clen;
    const uint8_t *compressed = rd_vec_u8(r, &clen);
    if (r->fail) { go_sparse_free(&sv); return; }
    static void *lib = NULL;
    if (!lib) lib = dlopen("libzstd.so.1", RTLD_NOW);
    if (!lib) lib = dlopen("libzstd.so", RTLD_NOW);
    if (!lib) { rd_fail(r, "zstd: libzstd.so.1 not available"); go_sparse_free(&sv); return; }
    zstd_decompress_fn dec = (zstd_decompress_fn)dlsym(lib, "ZSTD_decompress");
    zstd_iserror_fn iserr = (zstd_iserror_fn)dlsym(lib, "ZSTD_isError");
    uint8_t *bytes = (uint8_t *)malloc(total + 1);
    size_t got = dec(bytes, total, compressed, clen);
    if (iserr(got) || got != total) {
        rd_fail(r, "StringArray: Decompressed string length does not match the expected length");
        free(bytes); go_sparse_free(&sv); return;
    }
    strings_finish(r, out, &sv, bytes, total);
}

/* Dictionary::load, src/support.rs:821-838 */
static void rd_dictionary(reader *r, go_strings *strings, uint64_t **sorted) {
    rd_strings(r, strings);
    intvec ids = rd_intvector(r);
    *sorted = NULL;
    if (r->fail) return;
    *sorted = (uint64_t *)malloc((ids.len + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < ids.len; i++) (*sorted)[i] = intvec_get(&ids, i);
}

void go_tags_free(go_tags *t) {
    for (uint64_t i = 0; i < t->n; i++) { free(t->keys[i]); free(t->values[i]); }
    free(t->keys); free(t->values);
    memset(t, 0, sizeof(*t));
}

const char *go_tags_get(const go_tags *t, const char *key) {
    for (uint64_t i = 0; i < t->n; i++) if (strcmp(t->keys[i], key) == 0) return t->values[i];
    return NULL;
}

static char *dup_range(const uint8_t *p, uint64_t len) {
    char *s = (char *)malloc(len + 1);
    memcpy(s, p, len); s[len] = 0;
    return s;
}

/* Tags::load, src/support.rs:988-1007 */
static void rd_tags(reader *r, go_tags *tags) {
    memset(tags, 0, sizeof(*tags));
    go_strings lin;
    rd_strings(r, &lin);
    if (r->fail) return;
    if (lin.n % 2 != 0) { rd_fail(r, "Tags: Key without a value"); go_strings_free(&lin); return; }
    tags->keys = (char **)calloc(lin.n / 2 + 2, sizeof(char *));
    tags->values = (char **)calloc(lin.n / 2 + 2, sizeof(char *));
    for (uint64_t i = 0; i < lin.n / 2; i++) {
        char *key = dup_range(lin.bytes + lin.offsets[2 * i], lin.offsets[2 * i + 1] - lin.offsets[2 * i]);
        char *value = dup_range(lin.bytes + lin.offsets[2 * i + 1], lin.offsets[2 * i + 2] - lin.offsets[2 * i + 1]);
        for (char *c = key; *c; c++) *c = (char)tolower((unsigned char)*c);
        if (go_tags_get(tags, key)) { rd_fail(r, "Tags: Duplicate keys"); free(key); free(value); break; }
        tags->keys[tags->n] = key; tags->values[tags->n] = value; tags->n++;
    }
    go_strings_free(&lin);
}

/* tags.insert(SOURCE_KEY, SOURCE_VALUE), src/gbwt.rs:409 */
static void tags_insert(go_tags *t, const char *key, const char *value) {
    for (uint64_t i = 0; i < t->n; i++) {
        if (strcmp(t->keys[i], key) == 0) { free(t->values[i]); t->values[i] = strdup(value); return; }
    }
    t->keys = (char **)realloc(t->keys, (t->n + 1) * sizeof(char *));
    t->values = (char **)realloc(t->values, (t->n + 1) * sizeof(char *));
    t->keys[t->n] = strdup(key); t->values[t->n] = strdup(value); t->n++;
}

/* Header<T>::validate, src/headers.rs:101-115 */
static void check_header(reader *r, const char *name, uint64_t word0, uint64_t flags, uint32_t tag,
                         uint32_t min_version, uint32_t max_version, uint64_t mask) {
    uint32_t t = (uint32_t)(word0 & 0xFFFFFFFFu), v = (uint32_t)(word0 >> 32);
    if (t != tag) { rd_fail(r, "%s: Invalid tag %X", name, t); return; }
    if (v < min_version || v > max_version) { rd_fail(r, "%s: Invalid version %u (expected %u to %u)", name, v, min_version, max_version); return; }
    if ((flags & mask) != flags) { rd_fail(r, "%s: Invalid flags %llX for version %u", name, (unsigned long long)flags, v); return; }
}

void go_metadata_free(go_metadata *m) {
    if (!m) return;
    free(m->path_names);
    go_strings_free(&m->sample_names); free(m->sample_sorted);
    go_strings_free(&m->contig_names); free(m->contig_sorted);
    free(m);
}

static int strings_find(const go_strings *s, const char *name, uint64_t *id) {
    size_t len = strlen(name);
    for (uint64_t i = 0; i < s->n; i++) {
        if (s->offsets[i + 1] - s->offsets[i] == len && memcmp(s->bytes + s->offsets[i], name, len) == 0) { *id = i; return 1; }
    }
    return 0;
}

#define GENERIC_SAMPLE "_gbwt_ref"      /* src/lib.rs */
#define GENERIC_HAPLOTYPE 0xFFFFFFFFu   /* src/lib.rs */

/* Metadata::load, src/gbwt.rs:846-890 */
static go_metadata *rd_metadata(reader *r) {
    uint64_t word0 = rd_word(r);
    uint64_t samples = rd_word(r), haplotypes = rd_word(r), contigs = rd_word(r);
    uint64_t flags = rd_word(r);
    check_header(r, "MetadataHeader", word0, flags, 0x6B375E7Au, 2, 2, 0x7);
    if (r->fail) return NULL;
    go_metadata *m = (go_metadata *)calloc(1, sizeof(go_metadata));
    m->flags = flags; m->sample_count = samples; m->haplotype_count = haplotypes; m->contig_count = contigs;
    m->n_paths = rd_word(r);
    const uint64_t *pw = rd_words(r, 2 * m->n_paths);
    if (!r->fail) {
        m->path_names = (go_path_name *)malloc((m->n_paths + 1) * sizeof(go_path_name));
        memcpy(m->path_names, pw, m->n_paths * sizeof(go_path_name));
        if (((flags & 1) != 0) == (m->n_paths == 0)) rd_fail(r, "Metadata: Path name flag does not match the presence of path names");
    }
    rd_dictionary(r, &m->sample_names, &m->sample_sorted);
    if (!r->fail) {
        if (flags & 2) { if (samples != m->sample_names.n) rd_fail(r, "Metadata: Sample count does not match the number of sample names"); }
        else if (m->sample_names.n != 0) rd_fail(r, "Metadata: Sample names are present without the sample name flag");
    }
    rd_dictionary(r, &m->contig_names, &m->contig_sorted);
    if (!r->fail) {
        if (flags & 4) { if (contigs != m->contig_names.n) rd_fail(r, "Metadata: Contig count does not match the number of contig names"); }
        else if (m->contig_names.n != 0) rd_fail(r, "Metadata: Contig names are present without the contig name flag");
    }
    if (r->fail) { go_metadata_free(m); return NULL; }
    uint64_t gid;
    if (strings_find(&m->sample_names, GENERIC_SAMPLE, &gid)) {
        for (uint64_t i = 0; i < m->n_paths; i++)
            if (m->path_names[i].sample == gid && m->path_names[i].phase == GENERIC_HAPLOTYPE) m->path_names[i].phase = 0;
    }
    return m;
}

/* BWT::load, src/bwt.rs:176-185 */
static go_bwt *rd_bwt(reader *r) {
    go_bwt *bwt = (go_bwt *)calloc(1, sizeof(go_bwt));
    rd_sparse(r, &bwt->index);
    uint64_t len;
    const uint8_t *data = rd_vec_u8(r, &len);
    if (!r->fail && bwt->index.universe != len) rd_fail(r, "BWT: Index / data length mismatch");
    if (r->fail) { go_bwt_free(bwt); return NULL; }
    bwt->data = (uint8_t *)malloc(len + 8);
    memcpy(bwt->data, data, len);
    bwt->data_len = len;
    return bwt;
}

/* GBWT::load, src/gbwt.rs:402-438 */
static go_gbwt *rd_gbwt(reader *r) {
    uint64_t word0 = rd_word(r);
    uint64_t sequences = rd_word(r), size = rd_word(r), offset = rd_word(r), alphabet_size = rd_word(r);
    uint64_t flags = rd_word(r);
    check_header(r, "GBWTHeader", word0, flags, 0x6B376B37u, 5, 5, 0x7);
    if (!r->fail && !(flags & 4)) rd_fail(r, "GBWTHeader: SDSL format is not supported");
    if (r->fail) return NULL;
    go_tags tags;
    rd_tags(r, &tags);
    if (r->fail) return NULL;
    tags_insert(&tags, "source", "jltsiren/gbwt-rs");
    go_bwt *bwt = rd_bwt(r);
    if (!bwt) { go_tags_free(&tags); return NULL; }
    go_gbwt *g = go_gbwt_from_bwt(bwt, sequences, size, offset, alphabet_size, (flags & 1) != 0);
    g->flags = flags; g->tags = tags;
    uint64_t da_len = rd_word(r);          /* Vec<u64> document array samples, opaque (417) */
    rd_words(r, da_len);
    uint64_t meta_size = rd_word(r);       /* Option<Metadata> (420) */
    if (!r->fail && meta_size > 0) {
        uint64_t before = r->pos;
        g->metadata = rd_metadata(r);
        if (!r->fail && r->pos - before != meta_size) rd_fail(r, "GBWT: Metadata size does not match the option header");
    }
    if (!r->fail && (((flags & 2) != 0) != (g->metadata != NULL))) rd_fail(r, "GBWT: Invalid metadata flag in the header");
    if (!r->fail && g->metadata && (g->metadata->flags & 1)) {
        uint64_t expected = (flags & 1) ? sequences / 2 : sequences;
        if (g->metadata->n_paths > 0 && g->metadata->n_paths != expected) rd_fail(r, "GBWT: Invalid path count in the metadata");
    }
    if (r->fail) { go_gbwt_free(g); return NULL; }
    return g;
}

static uint64_t *read_file(const char *path, uint64_t *n_words, char *err, size_t errlen) {
    FILE *f = fopen(path, "rb");
    if (!f) { if (err) snprintf(err, errlen, "cannot open %s", path); return NULL; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0 || sz % 8 != 0) { if (err) snprintf(err, errlen, "file size is not a multiple of 8"); fclose(f); return NULL; }
    uint64_t *buf = (uint64_t *)malloc((size_t)sz + 8);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { if (err) snprintf(err, errlen, "short read"); free(buf); fclose(f); return NULL; }
    fclose(f);
    *n_words = (uint64_t)sz / 8;
    return buf;
}

go_gbwt *go_gbwt_load(const char *path, char *err, size_t errlen) {
    uint64_t n;
    uint64_t *buf = read_file(path, &n, err, errlen);
    if (!buf) return NULL;
    reader r = { buf, n, 0, 0, err, errlen };
    go_gbwt *g = rd_gbwt(&r);
    if (g && r.pos != r.n) { if (err) snprintf(err, errlen, "trailing data after the GBWT"); go_gbwt_free(g); g = NULL; }
    free(buf);
    return g;
}

/* Graph::load, src/graph.rs:296-338 */
static void rd_graph(reader *r, go_graph *graph) {
    memset(graph, 0, sizeof(*graph));
    uint64_t word0 = rd_word(r);
    graph->nodes = rd_word(r);
    graph->flags = rd_word(r);
    check_header(r, "GraphHeader", word0, graph->flags, 0x6B3764AFu, 3, 4, 0x3);
    if (!r->fail && !(graph->flags & 2)) rd_fail(r, "GraphHeader: SDSL format is not supported");
    if (r->fail) return;
    graph->version = word0 >> 32;
    if (graph->version >= 4) rd_strings_zstd(r, &graph->sequences); else rd_strings(r, &graph->sequences);
    rd_strings(r, &graph->segments);
    if (!r->fail && (((graph->flags & 1) != 0) == (graph->segments.n == 0)))
        rd_fail(r, "Graph: Translation flag does not match the presence of segment names");
    rd_sparse(r, &graph->mapping);
    if (r->fail) return;
    if (graph->flags & 1) {
        if (graph->mapping.universe <= graph->nodes) rd_fail(r, "Graph: Node-to-segment mapping does not match the number of nodes");
        else if (graph->mapping.universe != graph->sequences.n + 1) rd_fail(r, "Graph: Node-to-segment mapping does not match the number of sequences");
        else if (graph->mapping.ones != graph->segments.n) rd_fail(r, "Graph: Node-to-segment mapping does not match the number of segments");
    }
}

void go_gbz_free(go_gbz *z) {
    if (!z) return;
    go_tags_free(&z->tags);
    go_gbwt_free(z->index);
    go_strings_free(&z->graph.sequences);
    go_strings_free(&z->graph.segments);
    go_sparse_free(&z->graph.mapping);
    free(z->real_nodes);
    free(z);
}

/* GBZ::load, src/gbz.rs:674-717 */
go_gbz *go_gbz_load(const char *path, char *err, size_t errlen) {
    uint64_t n;
    uint64_t *buf = read_file(path, &n, err, errlen);
    if (!buf) return NULL;
    reader r = { buf, n, 0, 0, err, errlen };
    uint64_t word0 = rd_word(&r), flags = rd_word(&r);
    check_header(&r, "GBZHeader", word0, flags, 0x205A4247u, 1, 2, 0);
    go_gbz *z = (go_gbz *)calloc(1, sizeof(go_gbz));
    if (!r.fail) rd_tags(&r, &z->tags);
    if (!r.fail) tags_insert(&z->tags, "source", "jltsiren/gbwt-rs");
    if (!r.fail) z->index = rd_gbwt(&r);
    if (!r.fail && !go_gbwt_is_bidirectional(z->index)) rd_fail(&r, "GBZ: The GBWT index is not bidirectional");
    if (!r.fail) {
        z->potential_nodes = (z->index->alphabet_size - (z->index->offset + 1)) / 2;
        rd_graph(&r, &z->graph);
    }
    if (!r.fail && z->graph.sequences.n != z->potential_nodes) rd_fail(&r, "GBZ: Mismatch between GBWT alphabet size and Graph sequence count");
    if (!r.fail && r.pos != r.n) rd_fail(&r, "trailing data after the GBZ");
    if (r.fail) { go_gbz_free(z); free(buf); return NULL; }
    /* real_nodes cache, src/gbz.rs:694-704 (BWT::id_iter skips records whose first byte is 0) */
    z->real_nodes = (uint8_t *)calloc(z->potential_nodes + 1, 1);
    const go_bwt *bwt = z->index->bwt;
    for (uint64_t rid = 1; rid < go_bwt_len(bwt); rid++) {
        const uint8_t *bytes; size_t len;
        go_bwt_record_bytes(bwt, rid, &bytes, &len);
        if (len == 0 || bytes[0] == 0) continue;
        uint64_t gbwt_node = rid + z->index->offset;
        if ((gbwt_node & 1) == 0) z->real_nodes[(gbwt_node - (z->index->offset + 1)) / 2] = 1;
    }
    free(buf);
    return z;
}

const go_gbwt *go_gbz_gbwt(const go_gbz *z) { return z->index; }
uint64_t go_gbz_paths(const go_gbz *z) { return z->index->sequences / 2; }
uint64_t go_gbwt_metadata_paths(const go_gbwt *g) { return g->metadata ? g->metadata->n_paths : 0; }

/* ------------------------------------------------------------------------------------------ */
/* GFA text (src/bin/gbunzip.rs:193-550)                                                        */

typedef struct { char *p; size_t len, cap; } sbuf;

static void sb_write(sbuf *b, const void *data, size_t n) {
    if (b->len + n + 1 > b->cap) {
        while (b->len + n + 1 > b->cap) b->cap = b->cap ? 2 * b->cap : 4096;
        b->p = (char *)realloc(b->p, b->cap);
    }
    memcpy(b->p + b->len, data, n);
    b->len += n;
}
static void sb_str(sbuf *b, const char *s) { sb_write(b, s, strlen(s)); }
static void sb_u64(sbuf *b, uint64_t v) { char tmp[32]; int n = snprintf(tmp, sizeof(tmp), "%llu", (unsigned long long)v); sb_write(b, tmp, (size_t)n); }

static inline uint64_t z_first_node(const go_gbz *z) { return z->index->offset + 1; }

/* GBZ::has_node, src/gbz.rs:286-289 */
static int z_has_node(const go_gbz *z, uint64_t node_id) {
    uint64_t gbwt_node = 2 * node_id;
    if (!(gbwt_node > z->index->offset && gbwt_node < z->index->alphabet_size)) return 0;
    return z->real_nodes[(gbwt_node - z_first_node(z)) / 2] != 0;
}

typedef struct { uint64_t id; const uint8_t *name; uint64_t name_len; uint64_t nodes_start, nodes_end; const uint8_t *seq; uint64_t seq_len; } segment;

static uint64_t mapping_value(const go_gbz *z, uint64_t k) { uint64_t pos; return go_sparse_select(&z->graph.mapping, k, &pos); }

/* Graph::segment, src/graph.rs:179-184 / segment_nodes 213-218 */
static segment graph_segment(const go_gbz *z, uint64_t id) {
    const go_graph *g = &z->graph;
    segment s;
    s.id = id;
    s.name = g->segments.bytes + g->segments.offsets[id];
    s.name_len = g->segments.offsets[id + 1] - g->segments.offsets[id];
    s.nodes_start = mapping_value(z, id);
    s.nodes_end = (id + 1 < g->mapping.ones) ? mapping_value(z, id + 1) : g->mapping.universe;
    s.seq = g->sequences.bytes + g->sequences.offsets[s.nodes_start - 1];
    s.seq_len = g->sequences.offsets[s.nodes_end - 1] - g->sequences.offsets[s.nodes_start - 1];
    return s;
}

/* Graph::node_to_segment, src/graph.rs:186-198 (SparseVector::predecessor = last one at or before node_id) */
static segment graph_node_to_segment(const go_gbz *z, uint64_t node_id) {
    uint64_t lo = 0, hi = z->graph.mapping.ones;   /* find the last k with mapping[k] <= node_id */
    while (hi - lo > 1) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (mapping_value(z, mid) <= node_id) lo = mid; else hi = mid;
    }
    return graph_segment(z, lo);
}

/* EdgeIter over a record, src/gbz.rs:819-855: skips a leading ENDMARKER edge */
static uint64_t edge_iter_first(const go_record *rec) { return (rec->outdegree > 0 && rec->edges[0].node == GO_ENDMARKER) ? 1 : 0; }

/* write_link, src/bin/gbunzip.rs:319-332 */
static void write_link(sbuf *b, const uint8_t *from, size_t from_len, int from_rev, const uint8_t *to, size_t to_len, int to_rev) {
    sb_str(b, "L\t"); sb_write(b, from, from_len);
    sb_str(b, from_rev ? "\t-\t" : "\t+\t");
    sb_write(b, to, to_len);
    sb_str(b, to_rev ? "\t-\t*\n" : "\t+\t*\n");
}

/* GBZ::successors, src/gbz.rs:327-335 -> record or none */
static int z_successor_record(const go_gbz *z, uint64_t node_id, int reverse, go_record *rec) {
    if (!z_has_node(z, node_id)) return 0;
    uint64_t gbwt_node = 2 * node_id + (reverse ? 1 : 0);
    return go_bwt_record(z->index->bwt, gbwt_node - z->index->offset, rec);
}

/* write_segments + write_links, src/bin/gbunzip.rs:230-317 */
static void write_segments_links(const go_gbz *z, sbuf *b) {
    int translation = (z->graph.flags & 1) != 0;
    if (translation) {
        for (uint64_t id = 0; id < z->graph.segments.n; id++) {     /* GBZ::segment_iter filters on has_node(nodes.start) */
            segment s = graph_segment(z, id);
            if (!z_has_node(z, s.nodes_start)) continue;
            sb_str(b, "S\t"); sb_write(b, s.name, s.name_len); sb_str(b, "\t"); sb_write(b, s.seq, s.seq_len); sb_str(b, "\n");
        }
        for (uint64_t id = 0; id < z->graph.segments.n; id++) {
            segment s = graph_segment(z, id);
            if (!z_has_node(z, s.nodes_start)) continue;
            for (int rev = 0; rev < 2; rev++) {
                /* segment_successors, src/gbz.rs:391-405 */
                uint64_t node_id = rev ? s.nodes_start : s.nodes_end - 1;
                go_record rec;
                if (!z_successor_record(z, node_id, rev, &rec)) continue;   /* reference unwraps: would panic */
                for (uint64_t k = edge_iter_first(&rec); k < rec.outdegree; k++) {
                    uint64_t succ_node = rec.edges[k].node / 2; int succ_rev = (int)(rec.edges[k].node & 1);
                    if (!((z->graph.flags & 1) && z_has_node(z, succ_node))) break;   /* LinkIter::next (src/gbz.rs:996-999): node_to_segment -> None ends the iteration */
                    segment t = graph_node_to_segment(z, succ_node);
                    int canonical = rev ? (t.id > s.id || (t.id == s.id && !succ_rev)) : (t.id >= s.id);
                    if (canonical) write_link(b, s.name, s.name_len, rev, t.name, t.name_len, succ_rev);
                }
                go_record_free(&rec);
            }
        }
    } else {
        char name[32], name2[32];
        for (uint64_t sid = 0; sid < z->potential_nodes; sid++) {    /* node_iter over real_nodes */
            if (!z->real_nodes[sid]) continue;
            uint64_t node_id = (2 * sid + z_first_node(z)) / 2;
            int n = snprintf(name, sizeof(name), "%llu", (unsigned long long)node_id);
            uint64_t seq_id = (2 * node_id - z_first_node(z)) / 2;
            sb_str(b, "S\t"); sb_write(b, name, (size_t)n); sb_str(b, "\t");
            sb_write(b, z->graph.sequences.bytes + z->graph.sequences.offsets[seq_id],
                     z->graph.sequences.offsets[seq_id + 1] - z->graph.sequences.offsets[seq_id]);
            sb_str(b, "\n");
        }
        for (uint64_t sid = 0; sid < z->potential_nodes; sid++) {
            if (!z->real_nodes[sid]) continue;
            uint64_t node_id = (2 * sid + z_first_node(z)) / 2;
            int n = snprintf(name, sizeof(name), "%llu", (unsigned long long)node_id);
            for (int rev = 0; rev < 2; rev++) {
                go_record rec;
                if (!z_successor_record(z, node_id, rev, &rec)) continue;
                for (uint64_t k = edge_iter_first(&rec); k < rec.outdegree; k++) {
                    uint64_t succ = rec.edges[k].node / 2; int succ_rev = (int)(rec.edges[k].node & 1);
                    int canonical = rev ? (succ > node_id || (succ == node_id && !succ_rev)) : (succ >= node_id);
                    if (canonical) {
                        int n2 = snprintf(name2, sizeof(name2), "%llu", (unsigned long long)succ);
                        write_link(b, (const uint8_t *)name, (size_t)n, rev, (const uint8_t *)name2, (size_t)n2, succ_rev);
                    }
                }
                go_record_free(&rec);
            }
        }
    }
}

/* One item of a path line: (name bytes, orientation, sequence length) */
typedef struct { const uint8_t *name; uint64_t name_len; uint64_t node_id; int rev; } path_item;

/* Collects the forward path as items.  With translation this is SegmentPathIter (src/gbz.rs:1098-1169),
 * otherwise PathIter (1053-1059).  *seq_len accumulates label lengths for the W-line end coordinate. */
static path_item *collect_path(const go_gbz *z, uint64_t path_id, uint64_t *n_items, uint64_t *seq_len) {
    int64_t len = go_gbwt_sequence(z->index, 2 * path_id, NULL, 0);
    *n_items = 0; *seq_len = 0;
    if (len < 0) return NULL;
    uint64_t *nodes = (uint64_t *)malloc(((size_t)len + 1) * sizeof(uint64_t));
    go_gbwt_sequence(z->index, 2 * path_id, nodes, (uint64_t)len);
    path_item *items = (path_item *)malloc(((size_t)len + 1) * sizeof(path_item));
    uint64_t cnt = 0;
    if (z->graph.flags & 1) {
        int have_next = 0; uint64_t next_node = 0; int next_rev = 0;
        uint64_t seg_start = 0, seg_end = 0;
        for (int64_t k = 0; k < len; k++) {
            uint64_t node_id = nodes[k] / 2; int rev = (int)(nodes[k] & 1);
            if (have_next) {
                if (node_id != next_node || rev != next_rev) break;            /* fail */
            } else {
                if (!z_has_node(z, node_id)) break;                            /* node_to_segment -> None: fail */
                segment s = graph_node_to_segment(z, node_id);
                items[cnt].name = s.name; items[cnt].name_len = s.name_len; items[cnt].node_id = node_id; items[cnt].rev = rev;
                cnt++;
                *seq_len += s.seq_len;
                seg_start = s.nodes_start; seg_end = s.nodes_end;
                /* visit(): next = first node of the segment in this orientation, then advance() */
                next_node = rev ? seg_end - 1 : seg_start; next_rev = rev; have_next = 1;
            }
            /* advance(), src/gbz.rs:1124-1141 */
            if (!next_rev) { if (next_node + 1 < seg_end) next_node += 1; else have_next = 0; }
            else { if (next_node > seg_start) next_node -= 1; else have_next = 0; }
        }
    } else {
        for (int64_t k = 0; k < len; k++) {
            uint64_t node_id = nodes[k] / 2;
            items[cnt].name = NULL; items[cnt].name_len = 0; items[cnt].node_id = node_id; items[cnt].rev = (int)(nodes[k] & 1);
            cnt++;
            if (z_has_node(z, node_id)) {                                        /* sequence_len(node).unwrap_or(0) */
                uint64_t seq_id = (2 * node_id - z_first_node(z)) / 2;
                *seq_len += z->graph.sequences.offsets[seq_id + 1] - z->graph.sequences.offsets[seq_id];
            }
        }
    }
    free(nodes);
    *n_items = cnt;
    return items;
}

/* GBZ::segment_path(path, orientation) collected (src/gbz.rs:477-489; SegmentPathIter::next 1146-1169, visit / advance 1110-1141): the
 * (segment, orientation) pairs the iterator yields for SEQUENCE seq_id = 2 * path + orientation, as (segment id << 1) | orientation; the
 * iterator stops for good where a node is not the one the current segment expects, or maps to no segment.  Returns the number of pairs
 * (at most `cap` are written), -1 without a node-to-segment translation or for a sequence that does not exist (None). */
int64_t go_gbz_segment_path(const go_gbz *z, uint64_t seq_id, uint64_t *out, uint64_t cap) {
    if (!(z->graph.flags & 1)) return -1;                                       /* has_translation */
    int64_t len = go_gbwt_sequence(z->index, seq_id, NULL, 0);
    if (len < 0) return -1;
    uint64_t *nodes = (uint64_t *)malloc(((size_t)len + 1) * sizeof(uint64_t));
    go_gbwt_sequence(z->index, seq_id, nodes, (uint64_t)len);
    int have_next = 0, next_rev = 0;
    uint64_t next_node = 0, seg_start = 0, seg_end = 0, cnt = 0;
    for (int64_t k = 0; k < len; k++) {
        uint64_t node_id = nodes[k] / 2; int rev = (int)(nodes[k] & 1);          /* support::decode_node */
        if (have_next) {
            if (node_id != next_node || rev != next_rev) break;                 /* fail = true */
        } else {
            if (!z_has_node(z, node_id)) break;                                 /* node_to_segment -> None: fail */
            segment s = graph_node_to_segment(z, node_id);
            if (cnt < cap && out) out[cnt] = (s.id << 1) | (uint64_t)rev;
            cnt++;
            seg_start = s.nodes_start; seg_end = s.nodes_end;
            next_node = rev ? seg_end - 1 : seg_start; next_rev = rev; have_next = 1;    /* visit() */
        }
        if (!next_rev) { if (next_node + 1 < seg_end) next_node += 1; else have_next = 0; }     /* advance() */
        else { if (next_node > seg_start) next_node -= 1; else have_next = 0; }
    }
    free(nodes);
    return (int64_t)cnt;
}

static void sb_item_name(sbuf *b, const path_item *it) {
    if (it->name) sb_write(b, it->name, it->name_len); else sb_u64(b, it->node_id);
}

static void sb_name(sbuf *b, const go_strings *names, uint64_t id, int has_names) {
    if (has_names && id < names->n) sb_write(b, names->bytes + names->offsets[id], names->offsets[id + 1] - names->offsets[id]);
    else sb_u64(b, id);
}

/* write_p_line, src/bin/gbunzip.rs:438-478: "P", name, the path as name+/- tokens, overlaps "*" */
static void write_p_line(const go_gbz *z, uint64_t path_id, const char *name, size_t name_len, sbuf *b) {
    sb_str(b, "P\t");
    sb_write(b, name, name_len);
    sb_str(b, "\t");
    uint64_t n, seq_len;
    path_item *items = collect_path(z, path_id, &n, &seq_len);
    for (uint64_t k = 0; k < n; k++) {
        if (k > 0) sb_str(b, ",");
        sb_item_name(b, &items[k]);
        sb_str(b, items[k].rev ? "-" : "+");
    }
    free(items);
    sb_str(b, "\t*\n");
}

/* path_to_p_line, src/bin/gbunzip.rs:480-485: named by the contig */
static void p_line(const go_gbz *z, uint64_t path_id, sbuf *b) {
    const go_metadata *m = z->index->metadata;
    go_path_name pn = m->path_names[path_id];
    sbuf name = { NULL, 0, 0 };
    sb_name(&name, &m->contig_names, pn.contig, (m->flags & 4) != 0);
    write_p_line(z, path_id, name.p, name.len, b);
    free(name.p);
}

/* Metadata::pan_sn_path, src/gbwt.rs:709-713: sample_name # phase # contig_name (sample_name / contig_name fall back to the
   number, src/gbwt.rs:752-758, 800-806) */
char *go_metadata_pan_sn_path(const go_gbwt *g, uint64_t path_id, size_t *len) {
    const go_metadata *m = g->metadata;
    *len = 0;
    if (!m || path_id >= m->n_paths) return NULL;
    go_path_name pn = m->path_names[path_id];
    sbuf name = { NULL, 0, 0 };
    sb_name(&name, &m->sample_names, pn.sample, (m->flags & 2) != 0);
    sb_str(&name, "#"); sb_u64(&name, pn.phase); sb_str(&name, "#");
    sb_name(&name, &m->contig_names, pn.contig, (m->flags & 4) != 0);
    *len = name.len;
    return name.p;
}

/* path_to_pan_sn, src/bin/gbunzip.rs:487-491 */
static void pan_sn_line(const go_gbz *z, uint64_t path_id, sbuf *b) {
    size_t len = 0;
    char *name = go_metadata_pan_sn_path(z->index, path_id, &len);
    write_p_line(z, path_id, name, len, b);
    free(name);
}

/* path_to_w_line, src/bin/gbunzip.rs:495-550 */
static void w_line(const go_gbz *z, uint64_t path_id, sbuf *b) {
    const go_metadata *m = z->index->metadata;
    go_path_name pn = m->path_names[path_id];
    sb_str(b, "W\t");
    sb_name(b, &m->sample_names, pn.sample, (m->flags & 2) != 0);
    sb_str(b, "\t"); sb_u64(b, pn.phase); sb_str(b, "\t");
    sb_name(b, &m->contig_names, pn.contig, (m->flags & 4) != 0);
    sb_str(b, "\t"); sb_u64(b, pn.fragment); sb_str(b, "\t");
    uint64_t n, seq_len;
    path_item *items = collect_path(z, path_id, &n, &seq_len);
    sb_u64(b, (uint64_t)pn.fragment + seq_len);
    sb_str(b, "\t");
    for (uint64_t k = 0; k < n; k++) {
        sb_str(b, items[k].rev ? "<" : ">");
        sb_item_name(b, &items[k]);
    }
    free(items);
    sb_str(b, "\n");
}

char *go_gbz_path_lines(const go_gbz *z, const uint64_t *path_ids, uint64_t n, int mode, size_t *len) {
    sbuf b = { NULL, 0, 0 };
    sb_write(&b, "", 0);
    if (!z->index->metadata) { *len = 0; return b.p; }
    for (uint64_t k = 0; k < n; k++) {
        if (path_ids[k] >= z->index->metadata->n_paths) continue;
        if (mode == 0) p_line(z, path_ids[k], &b); else if (mode == 2) pan_sn_line(z, path_ids[k], &b); else w_line(z, path_ids[k], &b);
    }
    *len = b.len;
    if (!b.p) b.p = (char *)calloc(1, 1);
    return b.p;
}

/* write_gfa_impl, single-thread order (src/bin/gbunzip.rs:193-226, 343-417); path_mode 0 = default (P-lines of the generic sample,
   W-lines of the others), 1 = pan-sn (every path a P-line with its PanSN name), 2 = ref-only (the P-lines only): PathMode,
   src/bin/gbunzip.rs:63-76, 212-222 */
char *go_gbz_write_gfa_mode(const go_gbz *z, int path_mode, size_t *len) {
    sbuf b = { NULL, 0, 0 };
    const char *rs = go_tags_get(&z->index->tags, "reference_samples");
    if (rs) { sb_str(&b, "H\tVN:Z:1.1\tRS:Z:"); sb_str(&b, rs); sb_str(&b, "\n"); }
    else sb_str(&b, "H\tVN:Z:1.1\n");
    write_segments_links(z, &b);
    const go_metadata *m = z->index->metadata;
    if (m && path_mode == 1) {
        for (uint64_t p = 0; p < m->n_paths; p++) pan_sn_line(z, p, &b);
    } else if (m) {
        uint64_t ref_sample = 0;
        int have_ref = (m->flags & 2) && strings_find(&m->sample_names, GENERIC_SAMPLE, &ref_sample);
        if (have_ref) {
            for (uint64_t p = 0; p < m->n_paths; p++) if (m->path_names[p].sample == ref_sample) p_line(z, p, &b);
        } else ref_sample = m->sample_count;
        if (path_mode == 0)
            for (uint64_t p = 0; p < m->n_paths; p++) if (m->path_names[p].sample != ref_sample) w_line(z, p, &b);
    }
    *len = b.len;
    return b.p;
}

char *go_gbz_write_gfa(const go_gbz *z, size_t *len) { return go_gbz_write_gfa_mode(z, 0, len); }
