/*
 * gbwt_hip.h -- C ABI of libgbwt_hip.so: the MI355X (gfx950) drop-in for the GBWT LF-step hot path
 * of jltsiren/gbwt-rs (crate `gbz` 0.5.1).
 *
 * The reference has no FFI of its own (no `extern`, no build.rs); every entry point below names the
 * Rust interface it replaces (file:line into the reference repository) and mirrors its semantics in
 * batched form.  Plain pointers and sizes only; no torch / HIP types in any signature (streams and
 * device buffers are passed as `void *`).  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions:
 *   - every function returns a gbwt_hip_status; 0 = success.  "Not found" is a value
 *     (Option::None <-> valid[i] = 0), never an error.
 *   - a handle is immutable after open and safe for concurrent read-only calls from several host
 *     threads as long as each thread uses its own gbwt_hip_workspace (the reference shares &GBZ across
 *     rayon workers, src/bin/gbunzip.rs:421-434).
 *   - malformed *files* -> GBWT_HIP_INVALID_DATA (io::ErrorKind::InvalidData in the reference);
 *     where the reference asserts/panics (non-bidirectional index in bd_* calls) -> GBWT_HIP_BAD_ARGUMENT.
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails with
 *     GBWT_HIP_NO_DEVICE / GBWT_HIP_DEVICE_ERROR.
 *
 * Widths.  The reference is usize (u64) everywhere (bwt::Pos, src/bwt.rs:63-69); this ABI carries u64 in every struct, but the
 * device side computes in u32: node ids, record indices, offsets inside a record and rank-block indices.  An index outside that
 * range is refused AT OPEN with GBWT_HIP_UNSUPPORTED (never truncated, never a wrong answer later):
 *     alphabet_size > 2^32                          (node ids; SURVEY 8b allows u32 node ids iff alphabet_size <= 2^32)
 *     2^30 records or more                          (bits 30-31 of a record word carry flags)
 *     a record with 2^32 or more positions          (Record::len: offsets inside a record are u32; counted by the device pass at open)
 *     (size >> 6) + records >= 2^32 - 16            (rank-block indices)
 * Sequence lengths are u32 as well: an index in which one sequence has 2^32 - 16 or more nodes opens without sequence lengths and
 * samples (extractions then take the pool-output kernel).  Record byte streams and CSR outputs are 64-bit throughout (an index
 * whose rank blocks exceed 4 GiB is walked with 64-bit block addresses).  Query inputs are u64 and compared as such: a node or
 * offset that does not fit u32 cannot exist in an index that was opened, and yields "not found" (valid = 0).
 */
#ifndef GBWT_HIP_H
#define GBWT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GBWT_HIP_OK = 0,
    GBWT_HIP_INVALID_DATA = 1, /* io::ErrorKind::InvalidData: bad tag/version/flags, length mismatch, ... */
    GBWT_HIP_IO_ERROR = 2,     /* file cannot be opened / read */
    GBWT_HIP_BAD_ARGUMENT = 3, /* id out of range, index not bidirectional, null pointer, ... */
    GBWT_HIP_NO_DEVICE = 4,    /* no HIP device available */
    GBWT_HIP_DEVICE_ERROR = 5, /* a HIP runtime call failed */
    GBWT_HIP_CAPACITY = 6,     /* caller-provided output capacity too small (total is still reported) */
    GBWT_HIP_UNSUPPORTED = 7   /* e.g. alphabet_size > 2^32 (u32 node ids on device) */
} gbwt_hip_status;

/* bwt::Pos, src/bwt.rs:63-69 */
typedef struct { uint64_t node, offset; } gbwt_hip_pos;
/* gbwt::SearchState, src/gbwt.rs:455-460 (range = start..end) */
typedef struct { uint64_t node, start, end; } gbwt_hip_state;
/* gbwt::BidirectionalState, src/gbwt.rs:485-490 */
typedef struct { gbwt_hip_state forward, reverse; } gbwt_hip_bd_state;

/* GBWT statistics: len/sequences/alphabet_size/alphabet_offset/is_bidirectional/has_metadata,
 * src/gbwt.rs:108-182; records = BWT::len (src/bwt.rs:105-107). */
typedef struct {
    uint64_t size;            /* GBWT::len */
    uint64_t sequences;       /* GBWT::sequences */
    uint64_t alphabet_size;   /* GBWT::alphabet_size */
    uint64_t alphabet_offset; /* GBWT::alphabet_offset */
    uint64_t records;         /* BWT::len */
    uint64_t data_bytes;      /* length of the record byte stream */
    uint64_t paths;           /* metadata path names (0 if none) */
    uint32_t bidirectional;   /* GBWT::is_bidirectional */
    uint32_t has_metadata;    /* GBWT::has_metadata */
    uint32_t is_gbz;          /* file was a GBZ container (graph + translation available) */
    uint32_t has_translation; /* Graph::has_translation, src/graph.rs:158-160 */
    uint64_t max_record_len;  /* largest Record::len over all records (device pass at open) */
    uint64_t max_outdegree;   /* largest Record::outdegree */
} gbwt_hip_stats;

typedef struct gbwt_hip_index gbwt_hip_index;         /* opaque: host + device copies of one index */
typedef struct gbwt_hip_workspace gbwt_hip_workspace; /* opaque: per-caller device scratch + stream */

/* Message for the last failing call on this thread. */
const char *gbwt_hip_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only box; never fails). */
int gbwt_hip_device_count(void);

/* ---- load --------------------------------------------------------------------------------------
 * Replaces simple_sds::serialize::load_from::<GBWT | GBZ> (src/gbwt.rs:402-438, src/gbz.rs:674-717).
 * Detects GBWT vs GBZ by the header tag, rejects what the reference rejects (src/headers.rs:101-115,
 * 229-231; src/bwt.rs:179-181; src/gbz.rs:684-692) and uploads the record stream, the dense record
 * start array (decoded from the Elias-Fano index) and the decompressed endmarker (src/gbwt.rs:413-414)
 * to `device`.
 * THE FILE STAYS MAPPED while the handle is open (files of 4 MB or more): the host's own copy of the record bytes and starts -- read by the
 * S / L lines of gbwt_hip_write_gfa* and by a node-to-segment translation, by nothing else -- is made from the mapping when it is first
 * needed, not by the open (3.5 GB and 0.4 s of a 1.0 s open for an HPRC-sized GBZ).  Replace such a file by rename, never truncate or
 * rewrite it in place under an open handle.  GBWT_HIP_LAZY_HOST_RECORDS=0 in the environment: the copy is made by the open, as until round 5. */
gbwt_hip_status gbwt_hip_open_file(const char *path, int device, gbwt_hip_index **out);

/* Host-only parse + validation of a .gbwt/.gbz (no device needed): what load_from would accept. */
gbwt_hip_status gbwt_hip_parse_file(const char *path, gbwt_hip_stats *out);

/* Replaces constructing a GBWT from an in-memory BWT: the thin Rust shim passes the raw record stream
 * it already owns -- base pointer from BWT::compressed_record(0) (src/bwt.rs:134-143), record starts
 * recovered per record -- plus the header fields (src/gbwt.rs:108-174).  `starts` has n_records
 * entries; record i spans [starts[i], starts[i+1]) and the last one ends at data_len
 * (BWT::record_bytes, src/bwt.rs:116-121).  Inputs are borrowed only for the call. */
gbwt_hip_status gbwt_hip_open_records(const uint8_t *data, uint64_t data_len, const uint64_t *starts,
                                      uint64_t n_records, uint64_t alphabet_offset, uint64_t alphabet_size,
                                      uint64_t n_sequences, uint64_t size, int bidirectional, int device,
                                      gbwt_hip_index **out);
/* The same with a statement of what the handle is FOR (round 5): an index is replicated per GPU and most of what a handle holds in HBM
 * is built for one group of entry points only, so a caller that names its group pays for that group:
 *   GBWT_HIP_OPEN_EXTRACT  gbwt_hip_extract*, path sums / hashes / copies: walk descriptors (64 + 128 B per record), rank blocks and packed
 *                          two-step half-blocks (16 + 32 B per 64 positions of an outdegree-2 record), walk tables, sequence lengths + samples
 *   GBWT_HIP_OPEN_SEARCH   start / forward / backward / find / extend / bd_* / follow / search: raw descriptors (64 B per record), rank
 *                          blocks (16 B per 64 positions), LF tables of the records with outdegree > 2 (16 B per position)
 *   GBWT_HIP_OPEN_GFA      gbwt_hip_path_lines*, gbwt_hip_write_gfa* (implies EXTRACT): label lengths, translation and line header tables
 * A handle opened WITHOUT SEARCH whose walks never leave the descriptors and rank blocks (no record of outdegree > 2, no edge that failed
 * a check at open) also gives back, once it is open, what only the open itself, the search kernels and the non-default walk modes read: the
 * raw descriptors, the one-step walk descriptors and the plain rank blocks (128 B per record + 16 B per 64 positions: config 4 at its
 * stated size 65 -> 36 GB, the headline index 3.3 -> 2.2 GB); the catch-up steps of the walk then read the two-step descriptors and the
 * packed half-blocks, and the pool-output walk modes of gbwt_hip_workspace_tune return GBWT_HIP_UNSUPPORTED.
 * Record bytes, record starts and the endmarker are always there.  An entry point outside the handle's groups returns
 * GBWT_HIP_BAD_ARGUMENT.  gbwt_hip_open_file / gbwt_hip_open_records = GBWT_HIP_OPEN_ALL.  Config 3's index (1.1 M sites x 5 008
 * haplotypes): 11.4 GB opened for everything, 3.6 GB for SEARCH; gbwt_hip_memory_usage reports what a handle holds. */
enum { GBWT_HIP_OPEN_EXTRACT = 1, GBWT_HIP_OPEN_SEARCH = 2, GBWT_HIP_OPEN_GFA = 4, GBWT_HIP_OPEN_ALL = 7 };
gbwt_hip_status gbwt_hip_open_file_flags(const char *path, int device, uint32_t flags, gbwt_hip_index **out);
gbwt_hip_status gbwt_hip_open_records_flags(const uint8_t *data, uint64_t data_len, const uint64_t *starts,
                                            uint64_t n_records, uint64_t alphabet_offset, uint64_t alphabet_size,
                                            uint64_t n_sequences, uint64_t size, int bidirectional, int device, uint32_t flags,
                                            gbwt_hip_index **out);
void gbwt_hip_close(gbwt_hip_index *index);
gbwt_hip_status gbwt_hip_get_stats(const gbwt_hip_index *index, gbwt_hip_stats *out);

/* What a handle and a workspace hold (bytes).  index_device_bytes: every array of the handle in HBM -- the record bytes, starts,
 * descriptors, rank blocks, tables, samples, GFA tables; the full-width two-step blocks are counted once they have been built (on first
 * need).  An index is replicated per GPU, so this is also the cost of one more rank.  index_host_bytes: the host image (record
 * bytes, starts, names, node labels).  workspace_device_bytes (0 for ws == NULL): all scratch of the workspace, of which rows_bytes are
 * the extracted rows (the CSR node ids) and text_bytes the formatted GFA lines. */
typedef struct {
    uint64_t index_device_bytes, index_host_bytes;
    uint64_t workspace_device_bytes, rows_bytes, text_bytes;
    uint64_t rows_chunks;   /* physical chunks the rows are mapped from (virtual-memory API, GBWT_HIP_VMM); 0 = one hipMalloc */
} gbwt_hip_memory;
gbwt_hip_status gbwt_hip_memory_usage(const gbwt_hip_index *index, const gbwt_hip_workspace *ws, gbwt_hip_memory *out);

/* Where the time of gbwt_hip_open_* went (host clock, milliseconds; the one-shot flow of gbunzip, src/bin/gbunzip.rs:24-59, pays all
 * of it once per file): parse_ms = reading and validating the file (0 for gbwt_hip_open_records), upload_ms = host-to-device copies
 * and the per-record passes (descriptors, rank blocks, tables, endmarker), sample_ms = sequence lengths + sequence samples,
 * total_ms = the whole call.  samples = sequence samples built; checkpoint_sampling = 1 when they came from checkpoint sampling
 * (no sequence walked from end to end), with its number of launches, walkers, and hops that ended at the length cap (orphans).
 * line_sizes_ms (inside upload_ms; handles opened for GFA lines) = the walk that sizes the GFA line of every path once, at open -- token
 * bytes per 4 096 positions and summed label lengths, i.e. the W-line's end coordinate (src/bin/gbunzip.rs:532-540) -- so that no
 * gbwt_hip_path_lines* request sizes a line: 0 where the handle has no such table (no GFA group, a node-to-segment translation, no
 * sequence samples). */
typedef struct {
    double parse_ms, upload_ms, sample_ms, total_ms;
    uint64_t samples, checkpoint_walkers, checkpoint_orphans;
    uint32_t checkpoint_sampling, checkpoint_rounds;
    double line_sizes_ms;
} gbwt_hip_open_times;
gbwt_hip_status gbwt_hip_get_open_times(const gbwt_hip_index *index, gbwt_hip_open_times *out);

/* Workspaces own a HIP stream and reusable device scratch (path pool, CSR outputs). */
gbwt_hip_status gbwt_hip_workspace_create(const gbwt_hip_index *index, gbwt_hip_workspace **out);
void gbwt_hip_workspace_destroy(gbwt_hip_workspace *ws);
/* Tuning of the extraction kernel for this workspace (results never depend on it):
 *   walk_mode      0 = one lane per sequence, two LF steps per iteration on the two-step rank blocks built at open (default),
 *                  1 = lane-serial scan from the start of every record (the reference's access pattern),
 *                  2 = wave-cooperative decode of long records (all 64 lanes scan one record's runs),
 *                  3 = one lane per sequence, one LF step per iteration on the plain rank blocks
 *   paths_per_wave lanes of each wavefront that own a sequence, 1..64; 0 = automatic (default): at least 32 owners per
 *                  wave -- the walk is latency-bound, not throughput-bound, and every lane runs the same instructions
 *   small_record   mode 2 only: records of at most this many bytes are decoded by their own lane (default 16)
 * Environment overrides read at workspace creation: GBWT_HIP_WALK_MODE, GBWT_HIP_PATHS_PER_WAVE, GBWT_HIP_SMALL_RECORD. */
gbwt_hip_status gbwt_hip_workspace_tune(gbwt_hip_workspace *ws, uint32_t walk_mode, uint32_t paths_per_wave, uint32_t small_record);
/* The hipStream_t the workspace launches on (for event timing by the caller). */
void *gbwt_hip_workspace_stream(gbwt_hip_workspace *ws);

/* ---- path extraction ---------------------------------------------------------------------------
 * gbwt_hip_extract: GBWT::sequence(id).collect::<Vec<_>>() for every id (src/gbwt.rs:253-261,
 * SequenceIter::next 557-568).  CSR output: out_offsets[n+1], nodes of sequence k at
 * out_nodes[out_offsets[k] .. out_offsets[k+1]).  id >= sequences (the reference returns no iterator) and
 * an empty sequence both give a zero-length row: "not found" never fails the batch, and the caller tells
 * the two apart by id < sequences.  `*total` always receives the number of nodes (= LF steps); with
 * out_nodes == NULL it is a size query; if capacity < total the call returns GBWT_HIP_CAPACITY.  Host
 * buffers.  The sequences are walked ONCE per request: the rows stay in the workspace, and the call that
 * repeats the ids of the previous gbwt_hip_extract / gbwt_hip_extract_device on it (the fill call after a
 * size query) only copies them out.  gbwt_hip_follow and gbwt_hip_path_lines do the same. */
gbwt_hip_status gbwt_hip_extract(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *seq_ids,
                                 uint64_t n, uint64_t *out_offsets, uint32_t *out_nodes, uint64_t capacity,
                                 uint64_t *total);

/* Device-resident form: results stay in HBM inside the workspace (valid until the next call on it).
 * d_offsets: uint64_t[n+1], d_nodes: uint32_t[total] (device pointers).  This is what bench.py times. */
typedef struct { const uint64_t *d_offsets; const uint32_t *d_nodes; uint64_t total; uint64_t n; } gbwt_hip_paths;
gbwt_hip_status gbwt_hip_extract_device(const gbwt_hip_index *index, gbwt_hip_workspace *ws,
                                        const uint64_t *seq_ids, uint64_t n, gbwt_hip_paths *out);

/* One part of every row (round 4; SURVEY 8e's partition, cut the other way): the rows of gbwt_hip_extract_device are cut where their
 * walkers start anyway -- at the sequence samples the index made at open -- into `parts` stretches of (nearly) equal numbers of samples, and
 * this call walks and returns stretch `part` of every row only: row k of the result is nodes [from_k, to_k) of GBWT::sequence(seq_ids[k]),
 * and the stretches part = 0 .. parts - 1 of a row, back to back, are the row (rows with fewer samples than parts leave some stretches
 * empty; an index opened without samples gives the whole row as the LAST part).  This is how N GPUs share one batch: every rank walks
 * EVERY path over ITS N-th of the way -- it reads an N-th of the index and keeps whole waves on every record -- instead of every N-th
 * path over the whole way (measured on one GPU, an eighth of the headline batch: 0.53-0.60 ms per pass against 0.88 ms;
 * profiles/r04_shard_probe.txt); gbwt_hip_gather_rows(..., GBWT_HIP_GATHER_PARTS, ...) puts the rows together.  parts = 1 is
 * gbwt_hip_extract_device. */
gbwt_hip_status gbwt_hip_extract_part_device(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *seq_ids, uint64_t n,
                                             uint32_t part, uint32_t parts, gbwt_hip_paths *out);

/* Copies the last device-resident extraction of `ws` to host buffers: out_offsets[n + 1] and/or out_nodes[total]
 * (either may be NULL; capacity < total -> GBWT_HIP_CAPACITY).  With gbwt_hip_extract_device this is "extract once,
 * size the buffer from gbwt_hip_paths.total, copy". */
gbwt_hip_status gbwt_hip_copy_result(const gbwt_hip_index *index, gbwt_hip_workspace *ws, uint64_t *out_offsets,
                                     uint32_t *out_nodes, uint64_t capacity);

/* GBZ::path(path_id, orientation) (src/gbz.rs:461-466, PathIter 1053-1059): sequence id =
 * 2*path_id + orientation (support::encode_path, src/support.rs:229-231); output nodes stay GBWT-encoded
 * (node_id = v / 2, orientation = v & 1: support::decode_node, src/support.rs:180-182). */
gbwt_hip_status gbwt_hip_extract_paths(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *path_ids,
                                       uint64_t n, int reverse, uint64_t *out_offsets, uint32_t *out_nodes,
                                       uint64_t capacity, uint64_t *total);

/* ---- navigation --------------------------------------------------------------------------------
 * GBWT::start (src/gbwt.rs:213-219) and GBWT::forward (222-229) for n independent inputs. */
gbwt_hip_status gbwt_hip_start(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *seq_ids, uint64_t n,
                               gbwt_hip_pos *out, uint8_t *valid);
gbwt_hip_status gbwt_hip_forward(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const gbwt_hip_pos *in, uint64_t n,
                                 gbwt_hip_pos *out, uint8_t *valid);
/* GBWT::backward (src/gbwt.rs:236-250): Record::predecessor_at (src/bwt.rs:502-540) on the record of the flipped node,
 * then Record::offset_to (558-584) in the predecessor's record.  The reference asserts a bidirectional index; here a
 * unidirectional one returns GBWT_HIP_BAD_ARGUMENT. */
gbwt_hip_status gbwt_hip_backward(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const gbwt_hip_pos *in, uint64_t n,
                                  gbwt_hip_pos *out, uint8_t *valid);

/* ---- search ------------------------------------------------------------------------------------
 * GBWT::find (src/gbwt.rs:269-281), GBWT::extend (292-304). */
gbwt_hip_status gbwt_hip_find(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *nodes, uint64_t n,
                              gbwt_hip_state *out, uint8_t *valid);
gbwt_hip_status gbwt_hip_extend(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const gbwt_hip_state *states,
                                const uint64_t *nodes, uint64_t n, gbwt_hip_state *out, uint8_t *valid);
/* GBWT::bd_find (311-324), extend_forward (339-347), extend_backward (362-367).  The reference asserts a
 * bidirectional index; here a unidirectional one returns GBWT_HIP_BAD_ARGUMENT. */
gbwt_hip_status gbwt_hip_bd_find(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *nodes, uint64_t n,
                                 gbwt_hip_bd_state *out, uint8_t *valid);
gbwt_hip_status gbwt_hip_extend_forward(const gbwt_hip_index *index, gbwt_hip_workspace *ws,
                                        const gbwt_hip_bd_state *states, const uint64_t *nodes, uint64_t n,
                                        gbwt_hip_bd_state *out, uint8_t *valid);
gbwt_hip_status gbwt_hip_extend_backward(const gbwt_hip_index *index, gbwt_hip_workspace *ws,
                                         const gbwt_hip_bd_state *states, const uint64_t *nodes, uint64_t n,
                                         gbwt_hip_bd_state *out, uint8_t *valid);
/* GBZ::follow_forward / follow_backward + StateIter (src/gbz.rs:519-544, 1211-1251): every non-empty extension of each
 * state by one node, listed in the order of the edge list (EdgeIter, src/gbz.rs:819-855).  CSR output: the extensions
 * of state i are out_states[out_offsets[i] .. out_offsets[i+1]); valid[i] = 0 where the reference returns no iterator
 * (GBZ::successors: the node does not exist).  `*total` always receives the number of extensions; out_states == NULL
 * is a size query (out_offsets and valid are still filled); capacity < total -> GBWT_HIP_CAPACITY. */
gbwt_hip_status gbwt_hip_follow(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const gbwt_hip_bd_state *states, uint64_t n,
                                int backward, uint64_t *out_offsets, gbwt_hip_bd_state *out_states, uint64_t capacity,
                                uint64_t *total, uint8_t *valid);
/* Whole query in one launch, the shape of src/bin/benchmark.rs:155-169: for query q (row q of the
 * n x len matrix `queries`), find(q[0]) then extend by q[1..]; out/valid describe the final state
 * (valid = 0 as soon as any step returns None). */
gbwt_hip_status gbwt_hip_search(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *queries,
                                uint64_t n, uint64_t len, gbwt_hip_state *out, uint8_t *valid);

/* Bidirectional form of the same: bd_find(q[first]), then alternately extend_forward over q[first+1..] and
 * extend_backward over q[first-1..0] until both ends of the row are consumed (the usage pattern of
 * GBWT::bd_find / extend_forward / extend_backward, src/gbwt.rs:311-367). */
gbwt_hip_status gbwt_hip_bd_search(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *queries,
                                   uint64_t n, uint64_t len, uint64_t first, gbwt_hip_bd_state *out, uint8_t *valid);

/* Device-resident forms (round 5): the queries are in HBM already (`d_queries`: n x len u64, a device pointer of the caller, read on the
 * workspace stream), the final states stay in HBM inside the workspace (valid until the next query call on it): d_states[n], d_valid[n].
 * What a pipeline that produces its queries on the GPU calls, and what separates the kernel from PCIe: a million 10-node queries are
 * 80 MB in and 25 MB out around a 0.5 ms kernel: a host-pointer call is bound by PCIe (3 ms), whichever way the copies are staged
 * (GBWT_HIP_QUERY_PIPELINE, profiles/r05_query_call_sweep.txt). */
typedef struct { const gbwt_hip_state *d_states; const uint8_t *d_valid; uint64_t n; } gbwt_hip_states;
typedef struct { const gbwt_hip_bd_state *d_states; const uint8_t *d_valid; uint64_t n; } gbwt_hip_bd_states;
gbwt_hip_status gbwt_hip_search_device(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *d_queries, uint64_t n, uint64_t len,
                                       gbwt_hip_states *out);
gbwt_hip_status gbwt_hip_bd_search_device(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *d_queries, uint64_t n, uint64_t len,
                                          uint64_t first, gbwt_hip_bd_states *out);

/* ---- GFA text (GBZ handles with metadata) ------------------------------------------------------------
 * gbwt_hip_path_lines: the lines gbunzip writes for the given paths, in the order given, byte for byte:
 * mode 0 = P-lines named by the contig (write_p_line / path_to_p_line, src/bin/gbunzip.rs:438-485), mode 1 = W-lines
 * (path_to_w_line, src/bin/gbunzip.rs:495-550), mode 2 = P-lines with PanSN names sample#phase#contig (path_to_pan_sn,
 * src/bin/gbunzip.rs:487-491; Metadata::pan_sn_path, src/gbwt.rs:709-713).  The forward sequences are walked and the node tokens
 * formatted on the device; the host only contributes the name fields.  `*total` receives the number of bytes;
 * out == NULL is a size query; capacity < total -> GBWT_HIP_CAPACITY.  Graphs with a node-to-segment
 * translation print segment names (GBZ::segment_path / SegmentPathIter, src/gbz.rs:477-486, 1098-1169). */
gbwt_hip_status gbwt_hip_path_lines(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *path_ids, uint64_t n,
                                    int mode, char *out, uint64_t capacity, uint64_t *total);
/* Device-resident form: the text stays in HBM inside the workspace (valid until the next GFA call on it); line k is
 * d_text[d_line_offsets[k] .. d_line_offsets[k + 1]).  This is what a multi-GPU extraction hands to the RCCL gather
 * (gbwt_rs_amd/dist.py) and what gbwt_hip_path_lines copies out. */
typedef struct { const char *d_text; const uint64_t *d_line_offsets; uint64_t total; uint64_t n; } gbwt_hip_lines;
gbwt_hip_status gbwt_hip_path_lines_device(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *path_ids,
                                           uint64_t n, int mode, gbwt_hip_lines *out);
/* GBZ::segment_path(path, orientation) for a batch of SEQUENCE ids (2 * path + orientation; src/gbz.rs:477-489, SegmentPathIter 1098-1169):
 * a CSR of tokens, token = (segment id << 1) | orientation (0 forward, 1 reverse) -- the (Segment, Orientation) pairs the iterator yields, the
 * segment as its index in the translation (GBZ::segment_iter order; names and sequences belong to the host's Graph).  A path that is not a
 * concatenation of whole segments yields the tokens up to the place where the reference's iterator stops.  GBWT_HIP_BAD_ARGUMENT for a graph
 * without a node-to-segment translation (the reference returns None) and for a handle that was not opened for GFA lines.  out_offsets has n + 1
 * entries; out_tokens NULL = size query (*total = tokens of all rows; the rows are walked again by the fill call). */
gbwt_hip_status gbwt_hip_segment_paths(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const uint64_t *seq_ids, uint64_t n, uint64_t *out_offsets,
                                       uint64_t *out_tokens, uint64_t capacity, uint64_t *total);
/* gbwt_hip_write_gfa: the whole file `gbunzip -t 1` writes (write_gfa_impl, src/bin/gbunzip.rs:205-226, default
 * path mode): H, S and L lines from the host copy of the graph, then P-lines and W-lines in ascending path id. */
gbwt_hip_status gbwt_hip_write_gfa(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const char *path);
/* The same with gbunzip's `--paths MODE` (PathMode, src/bin/gbunzip.rs:63-76; dispatch 212-222): default = P-lines of the paths of
 * the generic sample `_gbwt_ref`, then W-lines of all others; pan-sn = every path as a P-line with its PanSN name (write_pan_sn,
 * 371-393); ref-only = the P-lines of the default mode and nothing else. */
enum { GBWT_HIP_PATHS_DEFAULT = 0, GBWT_HIP_PATHS_PAN_SN = 1, GBWT_HIP_PATHS_REF_ONLY = 2 };
gbwt_hip_status gbwt_hip_write_gfa_mode(const gbwt_hip_index *index, gbwt_hip_workspace *ws, const char *path, int path_mode);

/* ---- multi-GPU: the one exchange of a sharded extraction -------------------------------------------------------------
 * The reference's parallel axis is the path: rayon workers pull path ids and hand their finished lines to ONE writer behind a mutex
 * (src/bin/gbunzip.rs:27, 421-434).  Sharded over GPUs -- one process per GPU, the index replicated, path p on rank p mod world
 * (interleaved) or in contiguous blocks, no collective inside the walk -- that writer becomes an ordered gather on one rank, over RCCL:
 * an all-gather of the per-rank counts, ONE group of point-to-point sends (every peer streams to the root over its own xGMI link at
 * once), and kernels on the root that put the rows into path order.
 *
 * gbwt_hip_comm_unique_id: called by one rank; the 128 bytes travel to the others by whatever channel the host program has (a file,
 * a socket, MPI, a torch store).  gbwt_hip_comm_create: collective over the `world` ranks (ncclCommInitRank), each with its device.
 * RCCL is loaded at run time; without it these calls return GBWT_HIP_UNSUPPORTED and everything else works. */
typedef struct { char bytes[128]; } gbwt_hip_unique_id;
typedef struct gbwt_hip_comm gbwt_hip_comm;
gbwt_hip_status gbwt_hip_comm_unique_id(gbwt_hip_unique_id *out);
gbwt_hip_status gbwt_hip_comm_create(const gbwt_hip_unique_id *id, int rank, int world, int device, gbwt_hip_comm **out);
void gbwt_hip_comm_destroy(gbwt_hip_comm *comm);
/* gbwt_hip_gather_rows: the rows of the last gbwt_hip_extract_device on `ws` of EVERY rank (collective), gathered on `root` in path
 * order.  `interleaved` is the layout of the shards: GBWT_HIP_GATHER_BLOCKS = the rows of rank 0, then of rank 1, ...;
 * GBWT_HIP_GATHER_INTERLEAVED = row k of rank r is global row k * world + r (the ranks' row counts must be those of p -> rank p mod
 * world); GBWT_HIP_GATHER_PARTS = every rank holds its part of every row (gbwt_hip_extract_part_device with part = rank, parts = world,
 * the same ids everywhere): row k of the result is the parts of row k in rank order.  On the root *out describes the result -- device
 * memory of the communicator, valid until its next gather: d_offsets[rows + 1], d_nodes[total] --, elsewhere it is zeroed.
 * gbwt_hip_gather_lines: the same for the GFA lines of the last gbwt_hip_path_lines_device on `ws` (d_text, d_line_offsets): the final
 * GFA concatenation of north_star. */
enum { GBWT_HIP_GATHER_BLOCKS = 0, GBWT_HIP_GATHER_INTERLEAVED = 1, GBWT_HIP_GATHER_PARTS = 2 };
gbwt_hip_status gbwt_hip_gather_rows(gbwt_hip_comm *comm, const gbwt_hip_index *index, gbwt_hip_workspace *ws, int root, int interleaved,
                                     gbwt_hip_paths *out);
gbwt_hip_status gbwt_hip_gather_lines(gbwt_hip_comm *comm, const gbwt_hip_index *index, gbwt_hip_workspace *ws, int root, int interleaved,
                                      gbwt_hip_lines *out);
/* The last gather on `comm`, as this rank saw it: wall time from the first collective to the last kernel, bytes sent (peers) or
 * gathered (root), and whether the payload was staged through an ordinary allocation before the send (rows mapped from spread chunks;
 * GBWT_HIP_COMM_DIRECT=1 sends from the mapping). */
typedef struct { double ms; uint64_t bytes; uint32_t staged_send, reserved; } gbwt_hip_comm_stats;
gbwt_hip_status gbwt_hip_comm_last(const gbwt_hip_comm *comm, gbwt_hip_comm_stats *out);

/* ---- checking hooks for device-resident results -------------------------------------------------
 * Per-path sums of the node ids of the last gbwt_hip_extract_device call on `ws` (a wave-per-path
 * reduction on the device), copied to out_sums[n]: a cheap full-size checksum of the extraction. */
gbwt_hip_status gbwt_hip_path_sums(const gbwt_hip_index *index, gbwt_hip_workspace *ws, uint64_t *out_sums, uint64_t n);
/* The same with a checksum that depends on the ORDER of the nodes: out_hashes[k] = sum over the positions i of row k of
 * (node_i + 1) * splitmix64(i)  (mod 2^64; splitmix64(i): z = i + 0x9E3779B97F4A7C15, z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9,
 * z = (z ^ z >> 27) * 0x94D049BB133111EB, z ^ z >> 31).  What bench.py compares with the CPU oracle's walk of the same paths
 * (SequenceIter, src/gbwt.rs:557-568) at full size, where copying 13 GB of rows out to compare them would dominate the run. */
gbwt_hip_status gbwt_hip_path_hashes(const gbwt_hip_index *index, gbwt_hip_workspace *ws, uint64_t *out_hashes, uint64_t n);
/* Copies row k of the last device-resident extraction to host: out_nodes[min(len, capacity)], *len = row length. */
gbwt_hip_status gbwt_hip_copy_path(const gbwt_hip_index *index, gbwt_hip_workspace *ws, uint64_t k, uint32_t *out_nodes,
                                   uint64_t capacity, uint64_t *len);

/* ---- measurement hooks -------------------------------------------------------------------------
 * From HIP events on the workspace stream, for the last gbwt_hip_extract_device call: *walk_ms = duration of its
 * dominant kernel (the walk; bench.py's roofline object), *total_ms = everything the call put on the stream, from the
 * upload of the ids to the last kernel (lengths, offsets, walker order, walk; host waits in between included). */
gbwt_hip_status gbwt_hip_last_kernel_ms(const gbwt_hip_workspace *ws, float *walk_ms, float *total_ms);
/* The same for the last GFA lines request (gbwt_hip_path_lines / _device) on `ws`: *walk_ms = the walk kernel of its extraction,
 * *format_ms = everything behind the walk on the stream -- sizes, scans, the one host wait for the total, the formatting kernel. */
gbwt_hip_status gbwt_hip_last_lines_ms(const gbwt_hip_workspace *ws, float *walk_ms, float *format_ms);
/* Free and total memory of a device in bytes (hipMemGetInfo): what the tests use to see that a workspace gives its rows back. */
gbwt_hip_status gbwt_hip_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);
/* Kernel time (ms, HIP events on the workspace stream, host staging excluded) of the last navigation / search call
 * (start, forward, backward, find, extend, bd_*, search, bd_search and the *_device forms) on `ws`.  (With GBWT_HIP_QUERY_PIPELINE=1 a large
 * host-pointer call moves in chunks through the copy lanes and this is the span of the whole pipeline on the device, copies included.) */
gbwt_hip_status gbwt_hip_last_query_ms(const gbwt_hip_workspace *ws, float *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif
