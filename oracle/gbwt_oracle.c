/*
 * gbwt_oracle.c -- CPU ORACLE (test infrastructure, not product): codecs, Elias-Fano index,
 * BWT/Record, GBWT navigation and search.  See gbwt_oracle.h for the pinning statement.
 * Every function cites the reference lines it restates (file:line into /root/reference).
 * The per-step cost structure of the reference is kept on purpose (EF select per record lookup,
 * heap-allocated edge table per Record::new, second allocation per Record::lf), because this code
 * is also the timed CPU baseline ("port").
 */
#define _GNU_SOURCE
#include "gbwt_oracle.h"
#include "oracle_internal.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* byte buffer                                                                                 */

static void bytes_push(go_bytes *b, uint8_t byte) {
    if (b->len == b->cap) {
        b->cap = b->cap ? 2 * b->cap : 64;
        b->bytes = (uint8_t *)realloc(b->bytes, b->cap);
    }
    b->bytes[b->len++] = byte;
}

void go_bytes_free(go_bytes *b) { free(b->bytes); b->bytes = NULL; b->len = b->cap = 0; }

/* ------------------------------------------------------------------------------------------ */
/* ByteCode (src/support.rs:1048-1167)                                                         */

#define BC_MASK 0x7Fu  /* 1053 */
#define BC_FLAG 0x80u  /* 1054 */
#define BC_SHIFT 7u    /* 1055 */

/* ByteCode::write, src/support.rs:1063-1070 */
void go_bytecode_write(go_bytes *b, uint64_t value) {
    while (value > BC_MASK) {
        bytes_push(b, (uint8_t)((value & BC_MASK) | BC_FLAG));
        value >>= BC_SHIFT;
    }
    bytes_push(b, (uint8_t)value);
}

void go_bytecode_write_byte(go_bytes *b, uint8_t byte) { bytes_push(b, byte); }

/* ByteCodeIter::next, src/support.rs:1151-1164.  `result +=` with a shift that may exceed 63 is
 * a wrapping shift-by->=64 in release Rust only for malformed input; we mask the shift the same way
 * hardware does for well-formed (<= 10 byte) varints. */
int go_bytecode_next(const uint8_t *bytes, size_t len, size_t *offset, uint64_t *value) {
    unsigned shift = 0;
    uint64_t result = 0;
    while (*offset < len) {
        uint8_t v = bytes[*offset];
        *offset += 1;
        if (shift < 64) result += ((uint64_t)(v & BC_MASK)) << shift;
        shift += BC_SHIFT;
        if ((v & BC_FLAG) == 0) { *value = result; return 1; }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* RLE (src/support.rs:1193-1433)                                                              */

#define RLE_THRESHOLD 255u /* 1200 */
#define RLE_UNIVERSE 256u  /* 1201 */

/* RLE::sanitize, src/support.rs:1292-1296 */
static void rle_sanitize(uint64_t sigma, uint64_t *s, uint64_t *t) {
    *s = (sigma == 0) ? UINT64_MAX : sigma;
    *t = (*s < RLE_THRESHOLD) ? (RLE_UNIVERSE / *s) : 0;
}

void go_rle_init(go_rle *r, uint64_t sigma) {
    memset(r, 0, sizeof(*r));
    rle_sanitize(sigma, &r->sigma, &r->threshold);
}

void go_rle_set_sigma(go_rle *r, uint64_t sigma) { rle_sanitize(sigma, &r->sigma, &r->threshold); }

/* write_basic, src/support.rs:1286-1289 */
static void rle_write_basic(go_rle *r, uint64_t value, uint64_t len) {
    uint64_t code = value + r->sigma * (len - 1);
    bytes_push(&r->bytes, (uint8_t)code);
}

/* RLE::write + write_unchecked, src/support.rs:1225-1248 */
void go_rle_write(go_rle *r, go_run run) {
    if (run.len == 0) return;
    if (r->sigma >= RLE_THRESHOLD) {
        go_bytecode_write(&r->bytes, run.value);
        go_bytecode_write(&r->bytes, run.len - 1);
    } else if (run.len < r->threshold) {
        rle_write_basic(r, run.value, run.len);
    } else {
        rle_write_basic(r, run.value, r->threshold);
        go_bytecode_write(&r->bytes, run.len - r->threshold);
    }
}

void go_rle_write_int(go_rle *r, uint64_t value) { go_bytecode_write(&r->bytes, value); }

void go_rle_iter_init(go_rle_iter *it, const uint8_t *bytes, size_t len, uint64_t sigma) {
    it->bytes = bytes; it->len = len; it->offset = 0;
    rle_sanitize(sigma, &it->sigma, &it->threshold);
}

/* RLEIter::next, src/support.rs:1413-1430 */
int go_rle_iter_next(go_rle_iter *it, go_run *run) {
    run->value = 0; run->len = 0;
    if (it->sigma >= RLE_THRESHOLD) {
        uint64_t v;
        if (!go_bytecode_next(it->bytes, it->len, &it->offset, &v)) return 0;
        run->value = v;
        if (!go_bytecode_next(it->bytes, it->len, &it->offset, &v)) return 0;
        run->len = v + 1;
    } else {
        if (it->offset >= it->len) return 0;           /* ByteCodeIter::byte 1132-1139 */
        uint64_t byte = it->bytes[it->offset++];
        run->value = byte % it->sigma;
        run->len = byte / it->sigma + 1;
        if (run->len == it->threshold) {
            uint64_t v;
            if (!go_bytecode_next(it->bytes, it->len, &it->offset, &v)) return 0;
            run->len += v;
        }
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Elias-Fano SparseVector (simple-sds 0.4, un-vendored; published algorithm restated).         */
/* value_k = ((select1(high, k) - k) << w) | low[k]      (verified on fixtures) */

#define SEL_RATE 64u

static inline uint64_t low_get(const go_sparse *sv, uint64_t k) {
    uint64_t w = sv->low_width;
    if (w == 0) return 0;
    uint64_t bit = k * w, word = bit >> 6, off = bit & 63;
    uint64_t v = sv->low[word] >> off;
    if (off + w > 64) v |= sv->low[word + 1] << (64 - off);
    return (w == 64) ? v : (v & ((1ULL << w) - 1));
}

void go_sparse_free(go_sparse *sv) {
    free(sv->high); free(sv->low); free(sv->samples);
    memset(sv, 0, sizeof(*sv));
}

void go_sparse_build_support(go_sparse *sv) {
    uint64_t n_samples = sv->ones / SEL_RATE + 1;
    sv->samples = (uint64_t *)calloc(n_samples, sizeof(uint64_t));
    uint64_t seen = 0, words = (sv->high_bits + 63) / 64;
    for (uint64_t wi = 0; wi < words; wi++) {
        uint64_t word = sv->high[wi];
        while (word) {
            unsigned b = (unsigned)__builtin_ctzll(word);
            if (seen % SEL_RATE == 0) sv->samples[seen / SEL_RATE] = wi * 64 + b;
            seen++;
            word &= word - 1;
        }
    }
}

/* SparseVector::get_params + SparseBuilder (simple-sds): low width = max(1, round(log2(u*ln2/n))) */
int go_sparse_build(go_sparse *sv, uint64_t universe, const uint64_t *values, uint64_t n) {
    memset(sv, 0, sizeof(*sv));
    uint64_t w = go_sparse_low_width(universe, n);
    uint64_t buckets = (w >= 64) ? (universe ? 1 : 0) : ((universe >> w) + ((universe & ((1ULL << w) - 1)) ? 1 : 0));
    sv->universe = universe; sv->ones = n; sv->low_width = w; sv->low_len = n;
    sv->high_bits = n + buckets;
    sv->high = (uint64_t *)calloc(sv->high_bits / 64 + 2, 8);
    sv->low = (uint64_t *)calloc((n * w) / 64 + 2, 8);
    for (uint64_t k = 0; k < n; k++) {
        uint64_t v = values[k];
        if (v >= universe || (k > 0 && v < values[k - 1])) { go_sparse_free(sv); return -1; }
        uint64_t hi = (w >= 64) ? 0 : (v >> w), pos = hi + k;
        sv->high[pos >> 6] |= 1ULL << (pos & 63);
        uint64_t lo = (w >= 64) ? v : (v & ((1ULL << w) - 1));
        uint64_t bit = k * w, word = bit >> 6, off = bit & 63;
        sv->low[word] |= lo << off;
        if (off + w > 64) sv->low[word + 1] |= lo >> (64 - off);
    }
    go_sparse_build_support(sv);
    return 0;
}

uint64_t go_sparse_low_width(uint64_t universe, uint64_t ones) {
    uint64_t w = 1;
    if (ones > 0 && ones <= universe) {
        double ideal = go_log2((double)universe * 0.6931471805599453 / (double)ones);
        double r = go_round(ideal);
        w = (r < 1.0) ? 1 : (uint64_t)r;
    }
    return w;
}

/* position of the i-th set bit of high (select support: sample every 64th one, then popcount scan) */
static inline uint64_t high_select(const go_sparse *sv, uint64_t i) {
    uint64_t pos = sv->samples[i / SEL_RATE];
    uint64_t remaining = i % SEL_RATE;
    uint64_t wi = pos >> 6;
    uint64_t word = sv->high[wi] & (~0ULL << (pos & 63));
    for (;;) {
        uint64_t c = (uint64_t)__builtin_popcountll(word);
        if (remaining < c) break;
        remaining -= c;
        word = sv->high[++wi];
    }
    while (remaining--) word &= word - 1;
    return wi * 64 + (unsigned)__builtin_ctzll(word);
}

uint64_t go_sparse_select(const go_sparse *sv, uint64_t i, uint64_t *pos) {
    uint64_t p = high_select(sv, i);
    *pos = p;
    uint64_t w = sv->low_width;
    return ((w >= 64) ? 0 : ((p - i) << w)) | low_get(sv, i);
}

uint64_t go_sparse_next(const go_sparse *sv, uint64_t i, uint64_t *pos) {
    uint64_t p = *pos + 1;
    uint64_t wi = p >> 6;
    uint64_t word = sv->high[wi] & (~0ULL << (p & 63));
    while (word == 0) word = sv->high[++wi];
    p = wi * 64 + (unsigned)__builtin_ctzll(word);
    *pos = p;
    uint64_t w = sv->low_width;
    return ((w >= 64) ? 0 : ((p - (i + 1)) << w)) | low_get(sv, i + 1);
}

/* ------------------------------------------------------------------------------------------ */
/* BWTBuilder / BWT (src/bwt.rs:97-253)                                                        */

void go_builder_init(go_bwt_builder *b) {
    memset(b, 0, sizeof(*b));
    go_rle_init(&b->encoder, 0);
}

/* BWTBuilder::append, src/bwt.rs:241-253 */
void go_builder_append(go_bwt_builder *b, const go_pos *edges, size_t n_edges, const go_run *runs, size_t n_runs) {
    if (b->n == b->cap) {
        b->cap = b->cap ? 2 * b->cap : 16;
        b->offsets = (uint64_t *)realloc(b->offsets, b->cap * sizeof(uint64_t));
    }
    b->offsets[b->n++] = b->encoder.bytes.len;
    go_rle_write_int(&b->encoder, n_edges);
    uint64_t prev = 0;
    for (size_t i = 0; i < n_edges; i++) {
        go_rle_write_int(&b->encoder, edges[i].node - prev);
        go_rle_write_int(&b->encoder, edges[i].offset);
        prev = edges[i].node;
    }
    go_rle_set_sigma(&b->encoder, n_edges);
    for (size_t i = 0; i < n_runs; i++) go_rle_write(&b->encoder, runs[i]);
}

go_bwt *go_bwt_from_parts(const uint8_t *data, uint64_t data_len, const uint64_t *starts, uint64_t n_records) {
    go_bwt *bwt = (go_bwt *)calloc(1, sizeof(go_bwt));
    if (go_sparse_build(&bwt->index, data_len, starts, n_records) != 0) { free(bwt); return NULL; }
    bwt->data = (uint8_t *)malloc(data_len + 8);
    if (data_len) memcpy(bwt->data, data, data_len);
    bwt->data_len = data_len;
    return bwt;
}

/* From<BWTBuilder> for BWT, src/bwt.rs:192-203 */
go_bwt *go_bwt_from_builder(go_bwt_builder *b) {
    go_bwt *bwt = go_bwt_from_parts(b->encoder.bytes.bytes, b->encoder.bytes.len, b->offsets, b->n);
    free(b->offsets);
    go_bytes_free(&b->encoder.bytes);
    memset(b, 0, sizeof(*b));
    return bwt;
}

void go_bwt_free(go_bwt *bwt) {
    if (!bwt) return;
    go_sparse_free(&bwt->index);
    free(bwt->data);
    free(bwt);
}

uint64_t go_bwt_len(const go_bwt *bwt) { return bwt->index.ones; }          /* src/bwt.rs:105-107 */
uint64_t go_bwt_data_len(const go_bwt *bwt) { return bwt->data_len; }
const uint8_t *go_bwt_data(const go_bwt *bwt) { return bwt->data; }

/* BWT::record_bytes, src/bwt.rs:116-121 */
void go_bwt_record_bytes(const go_bwt *bwt, uint64_t i, const uint8_t **bytes, size_t *len) {
    uint64_t pos;
    uint64_t start = go_sparse_select(&bwt->index, i, &pos);
    uint64_t limit = (i + 1 < go_bwt_len(bwt)) ? go_sparse_next(&bwt->index, i, &pos) : bwt->data_len;
    *bytes = bwt->data + start;
    *len = (size_t)(limit - start);
}

/* Record::decompress_edges, src/bwt.rs:378-395.  Returns 0 for None (sigma == 0). */
static int decompress_edges(const uint8_t *bytes, size_t len, go_pos **edges_out, uint64_t *sigma_out, size_t *offset_out) {
    size_t off = 0;
    uint64_t sigma = 0;
    if (!go_bytecode_next(bytes, len, &off, &sigma)) return 0; /* reference would panic (unwrap) */
    if (sigma == 0) return 0;
    go_pos *edges = (go_pos *)malloc((size_t)sigma * sizeof(go_pos));  /* Vec<Pos> heap allocation */
    uint64_t prev = 0;
    for (uint64_t k = 0; k < sigma; k++) {
        uint64_t d = 0, o = 0;
        go_bytecode_next(bytes, len, &off, &d);
        uint64_t node = d + prev;
        prev = node;
        go_bytecode_next(bytes, len, &off, &o);
        edges[k].node = node; edges[k].offset = o;
    }
    *edges_out = edges; *sigma_out = sigma; *offset_out = off;
    return 1;
}

/* Record::skip_edges, src/bwt.rs:399-412 */
static int64_t skip_edges(const uint8_t *bytes, size_t len) {
    size_t off = 0;
    uint64_t sigma = 0, tmp;
    if (!go_bytecode_next(bytes, len, &off, &sigma)) return -1;
    if (sigma == 0) return -1;
    for (uint64_t k = 0; k < sigma; k++) {
        go_bytecode_next(bytes, len, &off, &tmp);
        go_bytecode_next(bytes, len, &off, &tmp);
    }
    return (int64_t)off;
}

/* Record::new, src/bwt.rs:341-351 */
static int record_new(uint64_t id, const uint8_t *bytes, size_t len, go_record *rec) {
    if (len == 0) return 0;
    go_pos *edges; uint64_t sigma; size_t off;
    if (!decompress_edges(bytes, len, &edges, &sigma, &off)) return 0;
    rec->id = id; rec->edges = edges; rec->outdegree = sigma;
    rec->bwt = bytes + off; rec->bwt_len = len - off;
    return 1;
}

/* BWT::record, src/bwt.rs:124-130 */
int go_bwt_record(const go_bwt *bwt, uint64_t i, go_record *rec) {
    if (i >= go_bwt_len(bwt)) return 0;
    const uint8_t *bytes; size_t len;
    go_bwt_record_bytes(bwt, i, &bytes, &len);
    return record_new(i, bytes, len, rec);
}

void go_record_free(go_record *rec) { free(rec->edges); rec->edges = NULL; }

/* BWT::compressed_record, src/bwt.rs:134-143 */
int64_t go_bwt_compressed_record(const go_bwt *bwt, uint64_t i, const uint8_t **bytes, size_t *len) {
    if (i >= go_bwt_len(bwt)) return -1;
    go_bwt_record_bytes(bwt, i, bytes, len);
    return skip_edges(*bytes, *len);
}

/* Record::len, src/bwt.rs:449-455 */
uint64_t go_record_len(const go_record *rec) {
    uint64_t result = 0;
    go_rle_iter it; go_run run;
    go_rle_iter_init(&it, rec->bwt, rec->bwt_len, rec->outdegree);
    while (go_rle_iter_next(&it, &run)) result += run.len;
    return result;
}

/* Record::decompress, src/bwt.rs:465-475 */
go_pos *go_record_decompress(const go_record *rec, uint64_t *n) {
    go_pos *edges = (go_pos *)malloc((size_t)rec->outdegree * sizeof(go_pos));
    memcpy(edges, rec->edges, (size_t)rec->outdegree * sizeof(go_pos));
    size_t cap = 16, cnt = 0;
    go_pos *result = (go_pos *)malloc(cap * sizeof(go_pos));
    go_rle_iter it; go_run run;
    go_rle_iter_init(&it, rec->bwt, rec->bwt_len, rec->outdegree);
    while (go_rle_iter_next(&it, &run)) {
        for (uint64_t k = 0; k < run.len; k++) {
            if (cnt == cap) { cap *= 2; result = (go_pos *)realloc(result, cap * sizeof(go_pos)); }
            result[cnt++] = edges[run.value];
            edges[run.value].offset += 1;
        }
    }
    free(edges);
    *n = cnt;
    return result;
}

/* Record::lf, src/bwt.rs:480-496 */
int go_record_lf(const go_record *rec, uint64_t i, go_pos *out) {
    go_pos *edges = (go_pos *)malloc((size_t)rec->outdegree * sizeof(go_pos));   /* self.edges.clone() */
    memcpy(edges, rec->edges, (size_t)rec->outdegree * sizeof(go_pos));
    uint64_t offset = 0;
    go_rle_iter it; go_run run;
    go_rle_iter_init(&it, rec->bwt, rec->bwt_len, rec->outdegree);
    while (go_rle_iter_next(&it, &run)) {
        if (offset + run.len > i) {
            int ok = 0;
            if (rec->edges[run.value].node != GO_ENDMARKER) {
                edges[run.value].offset += i - offset;
                *out = edges[run.value];
                ok = 1;
            }
            free(edges);
            return ok;
        }
        edges[run.value].offset += run.len;
        offset += run.len;
    }
    free(edges);
    return 0;
}

/* Record::predecessor_at, src/bwt.rs:502-540 */
int go_record_predecessor_at(const go_record *rec, uint64_t i, uint64_t *out) {
    uint64_t n = rec->outdegree;
    go_pos *edges = (go_pos *)malloc((size_t)n * sizeof(go_pos));
    for (uint64_t r = 0; r < n; r++) { edges[r].node = rec->edges[r].node; edges[r].offset = 0; }
    go_rle_iter it; go_run run;
    go_rle_iter_init(&it, rec->bwt, rec->bwt_len, n);
    while (go_rle_iter_next(&it, &run)) edges[run.value].offset += run.len;
    for (uint64_t r = 0; r < n; r++) if (edges[r].node != GO_ENDMARKER) edges[r].node ^= 1;
    for (uint64_t r = 1; r < n; r++) {
        if (edges[r - 1].node / 2 == edges[r].node / 2) { go_pos t = edges[r - 1]; edges[r - 1] = edges[r]; edges[r] = t; }
    }
    uint64_t offset = 0;
    int found = 0;
    for (uint64_t r = 0; r < n; r++) {
        offset += edges[r].offset;
        if (offset > i) {
            if (edges[r].node != GO_ENDMARKER) { *out = edges[r].node; found = 1; }
            break;
        }
    }
    free(edges);
    return found;
}

/* Record::edge_to, src/bwt.rs:543-555 */
int go_record_edge_to(const go_record *rec, uint64_t node, uint64_t *rank) {
    uint64_t low = 0, high = rec->outdegree;
    while (low < high) {
        uint64_t mid = low + (high - low) / 2;
        uint64_t m = rec->edges[mid].node;
        if (node < m) high = mid;
        else if (node == m) { *rank = mid; return 1; }
        else low = mid + 1;
    }
    return 0;
}

/* Record::offset_to, src/bwt.rs:558-584 */
int go_record_offset_to(const go_record *rec, go_pos pos, uint64_t *out) {
    if (pos.node == GO_ENDMARKER) return 0;
    uint64_t outrank;
    if (!go_record_edge_to(rec, pos.node, &outrank)) return 0;
    uint64_t succ_rank = rec->edges[outrank].offset;
    if (succ_rank > pos.offset) return 0;
    uint64_t offset = 0;
    go_rle_iter it; go_run run;
    go_rle_iter_init(&it, rec->bwt, rec->bwt_len, rec->outdegree);
    while (go_rle_iter_next(&it, &run)) {
        offset += run.len;
        if (run.value != outrank) continue;
        succ_rank += run.len;
        if (succ_rank > pos.offset) { *out = offset - (succ_rank - pos.offset); return 1; }
    }
    return 0;
}

/* support::intersect(a, b).len(), src/support.rs:332-334 (Range::len of max(starts)..min(ends)) */
static inline uint64_t intersect_len(uint64_t as, uint64_t ae, uint64_t bs, uint64_t be) {
    uint64_t s = as > bs ? as : bs, e = ae < be ? ae : be;
    return e > s ? e - s : 0;
}

/* Record::follow, src/bwt.rs:595-616 */
int go_record_follow(const go_record *rec, uint64_t start, uint64_t end, uint64_t node, uint64_t *rstart, uint64_t *rend) {
    if (start >= end || node == GO_ENDMARKER) return 0;
    uint64_t rank;
    if (!go_record_edge_to(rec, node, &rank)) return 0;
    uint64_t rs = rec->edges[rank].offset, re = rs, offset = 0;
    go_rle_iter it; go_run run;
    go_rle_iter_init(&it, rec->bwt, rec->bwt_len, rec->outdegree);
    while (go_rle_iter_next(&it, &run)) {
        if (run.value == rank) {
            rs += intersect_len(offset, offset + run.len, 0, start);
            re += intersect_len(offset, offset + run.len, 0, end);
        }
        offset += run.len;
        if (offset >= end) break;
    }
    if (rs >= re) return 0;
    *rstart = rs; *rend = re;
    return 1;
}

/* Record::bd_follow, src/bwt.rs:630-656 */
int go_record_bd_follow(const go_record *rec, uint64_t start, uint64_t end, uint64_t node,
                        uint64_t *rstart, uint64_t *rend, uint64_t *count_out) {
    if (start >= end || node == GO_ENDMARKER) return 0;
    uint64_t rank;
    if (!go_record_edge_to(rec, node, &rank)) return 0;
    uint64_t reverse = node ^ 1;
    uint64_t rs = rec->edges[rank].offset, re = rs, count = 0, offset = 0;
    go_rle_iter it; go_run run;
    go_rle_iter_init(&it, rec->bwt, rec->bwt_len, rec->outdegree);
    while (go_rle_iter_next(&it, &run)) {
        if (run.value == rank) {
            rs += intersect_len(offset, offset + run.len, 0, start);
            re += intersect_len(offset, offset + run.len, 0, end);
        }
        if ((rec->edges[run.value].node ^ 1) < reverse) count += intersect_len(offset, offset + run.len, start, end);
        offset += run.len;
        if (offset >= end) break;
    }
    if (rs >= re) return 0;
    *rstart = rs; *rend = re; *count_out = count;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* GBWT (src/gbwt.rs:95-385, 551-568)                                                          */

go_gbwt *go_gbwt_from_bwt(go_bwt *bwt, uint64_t sequences, uint64_t size, uint64_t offset,
                          uint64_t alphabet_size, int bidirectional) {
    go_gbwt *g = (go_gbwt *)calloc(1, sizeof(go_gbwt));
    g->bwt = bwt;
    g->sequences = sequences; g->size = size; g->offset = offset; g->alphabet_size = alphabet_size;
    g->flags = 4u | (bidirectional ? 1u : 0u);
    /* src/gbwt.rs:413-414: endmarker = record(ENDMARKER).decompress() unless the BWT is empty */
    if (go_bwt_len(bwt) > 0) {
        go_record rec;
        if (go_bwt_record(bwt, GO_ENDMARKER, &rec)) {
            g->endmarker = go_record_decompress(&rec, &g->endmarker_len);
            go_record_free(&rec);
        }
    }
    return g;
}

void go_gbwt_free(go_gbwt *g) {
    if (!g) return;
    go_bwt_free(g->bwt);
    free(g->endmarker);
    go_metadata_free(g->metadata);
    go_tags_free(&g->tags);
    free(g);
}

const go_bwt *go_gbwt_bwt(const go_gbwt *g) { return g->bwt; }
uint64_t go_gbwt_len(const go_gbwt *g) { return g->size; }
uint64_t go_gbwt_sequences(const go_gbwt *g) { return g->sequences; }
uint64_t go_gbwt_alphabet_size(const go_gbwt *g) { return g->alphabet_size; }
uint64_t go_gbwt_alphabet_offset(const go_gbwt *g) { return g->offset; }
int go_gbwt_is_bidirectional(const go_gbwt *g) { return (g->flags & 1u) != 0; }
int go_gbwt_has_metadata(const go_gbwt *g) { return g->metadata != NULL; }

static inline uint64_t first_node(const go_gbwt *g) { return g->offset + 1; }          /* 144-146 */
static inline uint64_t node_to_record(const go_gbwt *g, uint64_t node) { return node - g->offset; } /* 152-154 */

/* GBWT::start, src/gbwt.rs:213-219 */
int go_gbwt_start(const go_gbwt *g, uint64_t id, go_pos *out) {
    if (id < g->endmarker_len && g->endmarker[id].node != GO_ENDMARKER) { *out = g->endmarker[id]; return 1; }
    return 0;
}

/* GBWT::forward, src/gbwt.rs:222-229 */
int go_gbwt_forward(const go_gbwt *g, go_pos pos, go_pos *out) {
    if (pos.node < first_node(g)) return 0;
    go_record rec;
    if (!go_bwt_record(g->bwt, node_to_record(g, pos.node), &rec)) return 0;
    int ok = go_record_lf(&rec, pos.offset, out);
    go_record_free(&rec);
    return ok;
}

/* GBWT::backward, src/gbwt.rs:236-250 */
int go_gbwt_backward(const go_gbwt *g, go_pos pos, go_pos *out) {
    if (!go_gbwt_is_bidirectional(g)) return -1;          /* reference: assert! */
    if (pos.node <= first_node(g)) return 0;
    go_record rec, pred_rec;
    if (!go_bwt_record(g->bwt, node_to_record(g, pos.node ^ 1), &rec)) return 0;
    uint64_t predecessor;
    int ok = go_record_predecessor_at(&rec, pos.offset, &predecessor);
    go_record_free(&rec);
    if (!ok) return 0;
    if (!go_bwt_record(g->bwt, node_to_record(g, predecessor), &pred_rec)) return 0;
    uint64_t offset;
    ok = go_record_offset_to(&pred_rec, pos, &offset);
    go_record_free(&pred_rec);
    if (!ok) return 0;
    out->node = predecessor; out->offset = offset;
    return 1;
}

/* GBWT::sequence + SequenceIter::next, src/gbwt.rs:253-261, 557-568 */
int64_t go_gbwt_sequence(const go_gbwt *g, uint64_t id, uint64_t *out, uint64_t cap) {
    if (id >= g->sequences) return -1;
    go_pos pos;
    int have = go_gbwt_start(g, id, &pos);
    uint64_t n = 0;
    while (have) {
        go_pos next;
        int have_next = go_gbwt_forward(g, pos, &next);
        if (out && n < cap) out[n] = pos.node;
        n++;
        pos = next; have = have_next;
    }
    return (int64_t)n;
}

/* GBWT::find, src/gbwt.rs:269-281 */
int go_gbwt_find(const go_gbwt *g, uint64_t node, go_state *out) {
    if (node < first_node(g)) return 0;
    go_record rec;
    if (!go_bwt_record(g->bwt, node_to_record(g, node), &rec)) return 0;
    out->node = node; out->start = 0; out->end = go_record_len(&rec);
    go_record_free(&rec);
    return 1;
}

/* GBWT::extend, src/gbwt.rs:292-304 */
int go_gbwt_extend(const go_gbwt *g, const go_state *state, uint64_t node, go_state *out) {
    if (node < first_node(g)) return 0;
    go_record rec;
    /* node_to_record of a state node below the offset underflows in the reference (debug panic /
       release wrap -> record id out of range -> None); treat as None. */
    if (state->node < g->offset) return 0;
    if (!go_bwt_record(g->bwt, node_to_record(g, state->node), &rec)) return 0;
    uint64_t rs, re;
    int ok = go_record_follow(&rec, state->start, state->end, node, &rs, &re);
    go_record_free(&rec);
    if (!ok) return 0;
    out->node = node; out->start = rs; out->end = re;
    return 1;
}

/* GBWT::bd_find, src/gbwt.rs:311-324 */
int go_gbwt_bd_find(const go_gbwt *g, uint64_t node, go_bdstate *out) {
    if (!go_gbwt_is_bidirectional(g)) return -1;          /* reference: assert! */
    go_state st;
    if (!go_gbwt_find(g, node, &st)) return 0;
    out->forward = st;
    out->reverse.node = st.node ^ 1; out->reverse.start = st.start; out->reverse.end = st.end;
    return 1;
}

/* GBWT::bd_internal, src/gbwt.rs:371-384 */
static int bd_internal(const go_record *rec, const go_bdstate *state, uint64_t node, go_bdstate *out) {
    uint64_t rs, re, count;
    if (!go_record_bd_follow(rec, state->forward.start, state->forward.end, node, &rs, &re, &count)) return 0;
    go_bdstate r;
    r.forward.node = node; r.forward.start = rs; r.forward.end = re;
    uint64_t pos = state->reverse.start + count;
    r.reverse.node = state->reverse.node; r.reverse.start = pos; r.reverse.end = pos + (re - rs);
    *out = r;
    return 1;
}

/* GBWT::extend_forward, src/gbwt.rs:339-347 */
int go_gbwt_extend_forward(const go_gbwt *g, const go_bdstate *state, uint64_t node, go_bdstate *out) {
    if (!go_gbwt_is_bidirectional(g)) return -1;          /* reference: assert! */
    if (node < first_node(g)) return 0;
    if (state->forward.node < g->offset) return 0;
    go_record rec;
    if (!go_bwt_record(g->bwt, node_to_record(g, state->forward.node), &rec)) return 0;
    int ok = bd_internal(&rec, state, node, out);
    go_record_free(&rec);
    return ok;
}

/* BidirectionalState::flip, src/gbwt.rs:506-511; GBWT::extend_backward 362-367 */
int go_gbwt_extend_backward(const go_gbwt *g, const go_bdstate *state, uint64_t node, go_bdstate *out) {
    go_bdstate flipped, result;
    flipped.forward = state->reverse; flipped.reverse = state->forward;
    int ok = go_gbwt_extend_forward(g, &flipped, node ^ 1, &result);
    if (ok != 1) return ok;
    out->forward = result.reverse; out->reverse = result.forward;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Batched extraction = what gbunzip's write_lines does per path (src/bin/gbunzip.rs:421-434):   */
/* a pool of workers pulls path ids and runs SequenceIter over a shared read-only index.        */

typedef struct {
    const go_gbwt *g; const uint64_t *seq_ids; uint64_t n;
    uint64_t *lengths; const uint64_t *offsets; uint32_t *nodes;
    uint64_t *sums, *hashes;      /* per sequence: sum of its node ids, and the order-dependent hash of include/gbwt_hip.h */
    uint64_t next; uint64_t steps;
    pthread_mutex_t lock;
} extract_job;

/* the weight of position i in a path hash (include/gbwt_hip.h: gbwt_hip_path_hashes): splitmix64(i) */
static inline uint64_t position_weight(uint64_t i) {
    uint64_t z = i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void *extract_worker(void *arg) {
    extract_job *job = (extract_job *)arg;
    uint64_t local_steps = 0;
    for (;;) {
        uint64_t k = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
        if (k >= job->n) break;
        uint64_t id = job->seq_ids[k], n = 0, sum = 0, hash = 0;
        if (id < job->g->sequences) {
            go_pos pos;
            int have = go_gbwt_start(job->g, id, &pos);
            uint32_t *dst = job->nodes ? job->nodes + job->offsets[k] : NULL;
            while (have) {
                go_pos next;
                int have_next = go_gbwt_forward(job->g, pos, &next);
                if (dst) dst[n] = (uint32_t)pos.node;
                if (job->sums) { sum += pos.node; hash += (pos.node + 1) * position_weight(n); }
                n++;
                pos = next; have = have_next;
            }
        }
        if (job->lengths) job->lengths[k] = n;
        if (job->sums) { job->sums[k] = sum; job->hashes[k] = hash; }
        local_steps += n;
    }
    __atomic_fetch_add(&job->steps, local_steps, __ATOMIC_RELAXED);
    return NULL;
}

static uint64_t run_extract(extract_job *job, int threads) {
    if (threads < 1) threads = 1;
    if (threads == 1) { extract_worker(job); return job->steps; }
    pthread_t *tids = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    for (int t = 0; t < threads; t++) pthread_create(&tids[t], NULL, extract_worker, job);
    for (int t = 0; t < threads; t++) pthread_join(tids[t], NULL);
    free(tids);
    return job->steps;
}

uint64_t go_gbwt_extract_mt(const go_gbwt *g, const uint64_t *seq_ids, uint64_t n, int threads,
                            uint64_t *lengths, const uint64_t *offsets, uint32_t *nodes) {
    extract_job job;
    memset(&job, 0, sizeof(job));
    job.g = g; job.seq_ids = seq_ids; job.n = n; job.lengths = lengths; job.offsets = offsets; job.nodes = nodes;
    return run_extract(&job, threads);
}

uint64_t go_gbwt_extract_sums_mt(const go_gbwt *g, const uint64_t *seq_ids, uint64_t n, int threads,
                                 uint64_t *lengths, uint64_t *sums, uint64_t *hashes) {
    extract_job job;
    memset(&job, 0, sizeof(job));
    job.g = g; job.seq_ids = seq_ids; job.n = n; job.lengths = lengths; job.sums = sums; job.hashes = hashes;
    return run_extract(&job, threads);
}

/* Batched search (src/bin/benchmark.rs:155-169 shape) ------------------------------------------------- */
typedef struct {
    const go_gbwt *g; const uint64_t *queries; uint64_t n, len, first; int bd;
    go_state *out; go_bdstate *out_bd; uint8_t *valid; uint64_t next, found;
} search_job;

static void *search_worker(void *arg) {
    search_job *job = (search_job *)arg;
    uint64_t found = 0;
    for (;;) {
        uint64_t base = __atomic_fetch_add(&job->next, 256, __ATOMIC_RELAXED);
        if (base >= job->n) break;
        uint64_t limit = base + 256 < job->n ? base + 256 : job->n;
        for (uint64_t k = base; k < limit; k++) {
            const uint64_t *q = job->queries + k * job->len;
            int ok;
            if (!job->bd) {
                go_state st = {0, 0, 0}, nx;
                ok = job->len > 0 && go_gbwt_find(job->g, q[0], &st);
                for (uint64_t j = 1; ok && j < job->len; j++) { ok = go_gbwt_extend(job->g, &st, q[j], &nx); st = nx; }
                if (!ok) { st.node = 0; st.start = 0; st.end = 0; }
                job->out[k] = st;
            } else {
                go_bdstate st, nx;
                memset(&st, 0, sizeof(st));
                ok = job->first < job->len && go_gbwt_bd_find(job->g, q[job->first], &st) == 1;
                uint64_t fw = job->first + 1, bw = job->first;   /* next forward column, columns below bw are still to do */
                while (ok && (fw < job->len || bw > 0)) {
                    if (fw < job->len) { ok = go_gbwt_extend_forward(job->g, &st, q[fw], &nx) == 1; st = nx; fw++; }
                    if (ok && bw > 0) { ok = go_gbwt_extend_backward(job->g, &st, q[bw - 1], &nx) == 1; st = nx; bw--; }
                }
                if (!ok) memset(&st, 0, sizeof(st));
                job->out_bd[k] = st;
            }
            job->valid[k] = (uint8_t)(ok ? 1 : 0);
            found += ok ? 1 : 0;
        }
    }
    __atomic_fetch_add(&job->found, found, __ATOMIC_RELAXED);
    return NULL;
}

static uint64_t run_search(search_job *job, int threads) {
    if (threads < 1) threads = 1;
    if (threads == 1) { search_worker(job); return job->found; }
    pthread_t *tids = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    for (int t = 0; t < threads; t++) pthread_create(&tids[t], NULL, search_worker, job);
    for (int t = 0; t < threads; t++) pthread_join(tids[t], NULL);
    free(tids);
    return job->found;
}

uint64_t go_gbwt_search_mt(const go_gbwt *g, const uint64_t *queries, uint64_t n, uint64_t len, int threads, go_state *out, uint8_t *valid) {
    search_job job;
    memset(&job, 0, sizeof(job));
    job.g = g; job.queries = queries; job.n = n; job.len = len; job.out = out; job.valid = valid;
    return run_search(&job, threads);
}

uint64_t go_gbwt_bd_search_mt(const go_gbwt *g, const uint64_t *queries, uint64_t n, uint64_t len, uint64_t first, int threads,
                              go_bdstate *out, uint8_t *valid) {
    search_job job;
    memset(&job, 0, sizeof(job));
    job.g = g; job.queries = queries; job.n = n; job.len = len; job.first = first; job.bd = 1; job.out_bd = out; job.valid = valid;
    return run_search(&job, threads);
}

/* Algorithmic bytes per LF-step (SURVEY 8d / BASELINE.md 3): B(v,i) = H(v) + P(v,i) + 4, where H is
 * the header length (sigma varint + edge list, src/bwt.rs:378-395), P the run-stream bytes through
 * the end of the run containing i (src/bwt.rs:483-494), 4 the emitted u32 node id. */
uint64_t go_gbwt_extract_bytes(const go_gbwt *g, const uint64_t *seq_ids, uint64_t n, uint64_t *steps_out) {
    uint64_t total = 0, steps = 0;
    for (uint64_t k = 0; k < n; k++) {
        uint64_t id = seq_ids[k];
        if (id >= g->sequences) continue;
        go_pos pos;
        int have = go_gbwt_start(g, id, &pos);
        while (have) {
            steps++;
            total += 4;
            have = 0;
            if (pos.node < first_node(g)) break;
            const uint8_t *bytes; size_t len;
            uint64_t rid = node_to_record(g, pos.node);
            if (rid >= go_bwt_len(g->bwt)) break;
            go_bwt_record_bytes(g->bwt, rid, &bytes, &len);
            go_record rec;
            if (!record_new(rid, bytes, len, &rec)) { total += len ? 1 : 0; break; }
            total += (uint64_t)(rec.bwt - bytes);
            go_rle_iter it; go_run run;
            go_rle_iter_init(&it, rec.bwt, rec.bwt_len, rec.outdegree);
            uint64_t offset = 0;
            go_pos next = {0, 0};
            uint64_t *acc = (uint64_t *)calloc((size_t)rec.outdegree, sizeof(uint64_t));
            while (go_rle_iter_next(&it, &run)) {
                if (offset + run.len > pos.offset) {
                    if (rec.edges[run.value].node != GO_ENDMARKER) {
                        next.node = rec.edges[run.value].node;
                        next.offset = rec.edges[run.value].offset + acc[run.value] + (pos.offset - offset);
                        have = 1;
                    }
                    break;
                }
                acc[run.value] += run.len;
                offset += run.len;
            }
            total += it.offset;
            free(acc);
            go_record_free(&rec);
            pos = next;
        }
    }
    if (steps_out) *steps_out = steps;
    return total;
}

void go_free(void *p) { free(p); }
