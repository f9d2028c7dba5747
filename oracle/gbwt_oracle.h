/*
 * gbwt_oracle.h -- CPU ORACLE for the GBWT LF-step hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so.  The product library
 * (gbwt_rs_amd/csrc/libgbwt_hip.so) never links, loads or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference algorithm (jltsiren/gbwt-rs, crate gbz 0.5.1;
 * citations are file:line into /root/reference).  The reference is Rust and cannot be built in
 * this image (no cargo/rustc), and part of the path (Elias-Fano SparseVector) lives in the
 * un-vendored crate simple-sds 0.4 (Cargo.toml:14), whose published algorithm is restated here.
 *
 * PARITY PINNING: this oracle is pinned against every golden vector the reference's own tests
 * hold for the path (tests/test_oracle_*.py): paper-example records (src/bwt/tests.rs:10-87),
 * fixture known-answer paths (src/gbwt/tests.rs:116-162, src/gbz/tests.rs:371-381), ByteCode/RLE
 * KATs (src/support.rs:1042-1045,1188-1190, src/support/tests.rs:439-469), doc-test search states
 * (src/gbwt.rs:70-83) and the brute-force search oracle (src/gbwt/tests.rs:252-266).
 */
#ifndef GBWT_ORACLE_H
#define GBWT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO_ENDMARKER 0u /* src/lib.rs ENDMARKER */

/* src/bwt.rs:63-69 */
typedef struct { uint64_t node, offset; } go_pos;
/* src/support.rs Run */
typedef struct { uint64_t value, len; } go_run;
/* src/gbwt.rs:455-460 (range is start..end) */
typedef struct { uint64_t node, start, end; } go_state;
/* src/gbwt.rs:485-490 */
typedef struct { go_state forward, reverse; } go_bdstate;

/* ---- codecs (src/support.rs:1048-1167, 1193-1433) ---- */
typedef struct { uint8_t *bytes; size_t len, cap; } go_bytes;
void go_bytes_free(go_bytes *b);
void go_bytecode_write(go_bytes *b, uint64_t value);                 /* ByteCode::write 1063-1070 */
void go_bytecode_write_byte(go_bytes *b, uint8_t byte);              /* 1073-1075 */
/* ByteCodeIter::next 1151-1164: returns 1 and stores value, or 0 when the slice ends. */
int go_bytecode_next(const uint8_t *bytes, size_t len, size_t *offset, uint64_t *value);

typedef struct { go_bytes bytes; uint64_t sigma, threshold; } go_rle;
void go_rle_init(go_rle *r, uint64_t sigma);                          /* RLE::with_sigma 1209-1216 */
void go_rle_set_sigma(go_rle *r, uint64_t sigma);                     /* 1279-1283 */
void go_rle_write(go_rle *r, go_run run);                             /* 1225-1248 */
void go_rle_write_int(go_rle *r, uint64_t value);                     /* 1256-1258 */
typedef struct { const uint8_t *bytes; size_t len, offset; uint64_t sigma, threshold; } go_rle_iter;
void go_rle_iter_init(go_rle_iter *it, const uint8_t *bytes, size_t len, uint64_t sigma); /* 1369-1376 */
int go_rle_iter_next(go_rle_iter *it, go_run *run);                   /* 1413-1430 */

/* ---- Elias-Fano SparseVector (simple-sds 0.4; call sites src/bwt.rs:106,117-119) ---- */
typedef struct {
    uint64_t universe, ones;
    uint64_t *high; uint64_t high_bits;
    uint64_t *low;  uint64_t low_width; uint64_t low_len;
    uint64_t *samples; /* position in high of every 64th one (select support, rebuilt in memory) */
} go_sparse;
void go_sparse_free(go_sparse *sv);
/* builds from a sorted list of values (SparseBuilder / try_from_iter) with the writer rule of Appendix A */
int go_sparse_build(go_sparse *sv, uint64_t universe, const uint64_t *values, uint64_t n);
/* select_iter(i).next(): value of the i-th one; *pos receives the high-bit position for the following next() */
uint64_t go_sparse_select(const go_sparse *sv, uint64_t i, uint64_t *pos);
/* OneIter::next after a select at (i,pos): value of one i+1 (caller guarantees i+1 < ones) */
uint64_t go_sparse_next(const go_sparse *sv, uint64_t i, uint64_t *pos);

/* ---- BWT + Record (src/bwt.rs) ---- */
typedef struct { go_sparse index; uint8_t *data; uint64_t data_len; } go_bwt;
typedef struct { uint64_t id; go_pos *edges; uint64_t outdegree; const uint8_t *bwt; size_t bwt_len; } go_record;

/* BWTBuilder (src/bwt.rs:212-253) */
typedef struct { uint64_t *offsets; size_t n, cap; go_rle encoder; } go_bwt_builder;
void go_builder_init(go_bwt_builder *b);
void go_builder_append(go_bwt_builder *b, const go_pos *edges, size_t n_edges, const go_run *runs, size_t n_runs);
go_bwt *go_bwt_from_builder(go_bwt_builder *b);                       /* From<BWTBuilder> 192-203; frees builder */
go_bwt *go_bwt_from_parts(const uint8_t *data, uint64_t data_len, const uint64_t *starts, uint64_t n_records);
void go_bwt_free(go_bwt *bwt);
uint64_t go_bwt_len(const go_bwt *bwt);                               /* 105-107 */
uint64_t go_bwt_data_len(const go_bwt *bwt);
const uint8_t *go_bwt_data(const go_bwt *bwt);
/* record_bytes 116-121 */
void go_bwt_record_bytes(const go_bwt *bwt, uint64_t i, const uint8_t **bytes, size_t *len);
/* BWT::record 124-130 -> 1 + filled record (caller must go_record_free), or 0 for None */
int go_bwt_record(const go_bwt *bwt, uint64_t i, go_record *rec);
void go_record_free(go_record *rec);
/* compressed_record 134-143: offset of the split between edges and bwt, or -1 */
int64_t go_bwt_compressed_record(const go_bwt *bwt, uint64_t i, const uint8_t **bytes, size_t *len);
uint64_t go_record_len(const go_record *rec);                         /* 449-455 */
/* decompress 465-475: returns malloc'd array of *n positions */
go_pos *go_record_decompress(const go_record *rec, uint64_t *n);
int go_record_lf(const go_record *rec, uint64_t i, go_pos *out);      /* 480-496 */
int go_record_predecessor_at(const go_record *rec, uint64_t i, uint64_t *out); /* 502-540 */
int go_record_edge_to(const go_record *rec, uint64_t node, uint64_t *rank);    /* 543-555 */
int go_record_offset_to(const go_record *rec, go_pos pos, uint64_t *out);      /* 558-584 */
int go_record_follow(const go_record *rec, uint64_t start, uint64_t end, uint64_t node,
                     uint64_t *rstart, uint64_t *rend);               /* 595-616 */
int go_record_bd_follow(const go_record *rec, uint64_t start, uint64_t end, uint64_t node,
                        uint64_t *rstart, uint64_t *rend, uint64_t *count); /* 630-656 */

/* ---- GBWT navigation / search (src/gbwt.rs) ---- */
typedef struct go_gbwt go_gbwt;
go_gbwt *go_gbwt_from_bwt(go_bwt *bwt, uint64_t sequences, uint64_t size, uint64_t offset,
                          uint64_t alphabet_size, int bidirectional); /* takes ownership of bwt */
void go_gbwt_free(go_gbwt *g);
const go_bwt *go_gbwt_bwt(const go_gbwt *g);
uint64_t go_gbwt_len(const go_gbwt *g);            /* 108-110 */
uint64_t go_gbwt_sequences(const go_gbwt *g);      /* 120-122 */
uint64_t go_gbwt_alphabet_size(const go_gbwt *g);  /* 126-128 */
uint64_t go_gbwt_alphabet_offset(const go_gbwt *g);/* 132-134 */
int go_gbwt_is_bidirectional(const go_gbwt *g);    /* 172-174 */
int go_gbwt_start(const go_gbwt *g, uint64_t id, go_pos *out);                 /* 213-219 */
int go_gbwt_forward(const go_gbwt *g, go_pos pos, go_pos *out);                /* 222-229 */
int go_gbwt_backward(const go_gbwt *g, go_pos pos, go_pos *out);               /* 236-250 */
/* sequence(id).collect() 253-261,557-568: -1 if id >= sequences, else number of nodes (written up to cap) */
int64_t go_gbwt_sequence(const go_gbwt *g, uint64_t id, uint64_t *out, uint64_t cap);
int go_gbwt_find(const go_gbwt *g, uint64_t node, go_state *out);              /* 269-281 */
int go_gbwt_extend(const go_gbwt *g, const go_state *state, uint64_t node, go_state *out);       /* 292-304 */
int go_gbwt_bd_find(const go_gbwt *g, uint64_t node, go_bdstate *out);         /* 311-324; -1 if not bidirectional */
int go_gbwt_extend_forward(const go_gbwt *g, const go_bdstate *state, uint64_t node, go_bdstate *out);  /* 339-347 */
int go_gbwt_extend_backward(const go_gbwt *g, const go_bdstate *state, uint64_t node, go_bdstate *out); /* 362-367 */

/* Batched extraction like gbunzip's write_lines (src/bin/gbunzip.rs:421-434): sequence ids are pulled
 * from a shared counter by `threads` workers sharing the read-only index.  Output is CSR:
 * offsets[n+1] must be precomputed by the caller when nodes != NULL (use lengths from a counting call).
 * Returns total LF steps (nodes emitted).  With nodes == NULL only lengths[] is filled. */
uint64_t go_gbwt_extract_mt(const go_gbwt *g, const uint64_t *seq_ids, uint64_t n, int threads,
                            uint64_t *lengths, const uint64_t *offsets, uint32_t *nodes);
/* The same walk without the rows: per sequence its length, the sum of its node ids and the order-dependent checksum that
 * gbwt_hip_path_hashes (include/gbwt_hip.h) computes on the device, sum of (node + 1) * splitmix64(position).  This is what bench.py's
 * cpu_baseline leg times, so that every bench run is also a parity check of the sampled paths at full size. */
uint64_t go_gbwt_extract_sums_mt(const go_gbwt *g, const uint64_t *seq_ids, uint64_t n, int threads,
                                 uint64_t *lengths, uint64_t *sums, uint64_t *hashes);
/* Batched search with a worker pool, the loop of src/bin/benchmark.rs:155-169: for query q (row q of the n x len
 * matrix) find(q[0]) then extend by q[1..]; out[q] / valid[q] describe the final state.  With bidirectional != 0 the
 * query starts with bd_find(q[first]) and alternates extend_forward (q[first+1], ...) / extend_backward (q[first-1], ...)
 * until both ends are reached; out_bd is filled instead.  Returns the number of successful queries. */
uint64_t go_gbwt_search_mt(const go_gbwt *g, const uint64_t *queries, uint64_t n, uint64_t len, int threads,
                           go_state *out, uint8_t *valid);
uint64_t go_gbwt_bd_search_mt(const go_gbwt *g, const uint64_t *queries, uint64_t n, uint64_t len, uint64_t first, int threads,
                              go_bdstate *out, uint8_t *valid);
/* Instrumented (untimed) pass: algorithmic bytes W = sum over steps of H(v)+P(v,i)+4. */
uint64_t go_gbwt_extract_bytes(const go_gbwt *g, const uint64_t *seq_ids, uint64_t n, uint64_t *steps);

/* ---- files (simple-sds Serialize; src/headers.rs, src/gbwt.rs:389-438, src/gbz.rs:662-717) ---- */
typedef struct go_gbz go_gbz;
go_gbwt *go_gbwt_load(const char *path, char *err, size_t errlen);
go_gbz *go_gbz_load(const char *path, char *err, size_t errlen);
void go_gbz_free(go_gbz *z);
const go_gbwt *go_gbz_gbwt(const go_gbz *z);
/* gbunzip GFA text (src/bin/gbunzip.rs:193-550, default path mode, single-thread order).
 * Returns malloc'd buffer, *len bytes. */
char *go_gbz_write_gfa(const go_gbz *z, size_t *len);
char *go_gbz_write_gfa_mode(const go_gbz *z, int path_mode, size_t *len);   /* 0 default, 1 pan-sn, 2 ref-only */
char *go_metadata_pan_sn_path(const go_gbwt *g, uint64_t path_id, size_t *len);   /* Metadata::pan_sn_path; NULL = None */
/* only the lines for the given path ids: mode 0 = P (path_to_p_line), 1 = W (path_to_w_line), 2 = P with PanSN names (path_to_pan_sn) */
char *go_gbz_path_lines(const go_gbz *z, const uint64_t *path_ids, uint64_t n, int mode, size_t *len);
/* GBZ::segment_path collected for sequence seq_id = 2 * path + orientation (src/gbz.rs:477-489, 1098-1169): (segment id << 1) | orientation per pair */
int64_t go_gbz_segment_path(const go_gbz *z, uint64_t seq_id, uint64_t *out, uint64_t cap);
void go_free(void *p);
/* metadata peek for tests */
uint64_t go_gbz_paths(const go_gbz *z);
int go_gbwt_has_metadata(const go_gbwt *g);
uint64_t go_gbwt_metadata_paths(const go_gbwt *g);

#ifdef __cplusplus
}
#endif
#endif
