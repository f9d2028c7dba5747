/* oracle_internal.h -- private structures of the CPU ORACLE (test infrastructure, not product). */
#ifndef GBWT_ORACLE_INTERNAL_H
#define GBWT_ORACLE_INTERNAL_H

#include "gbwt_oracle.h"
#include <math.h>

#define go_log2(x) log2(x)
#define go_round(x) round(x)

uint64_t go_sparse_low_width(uint64_t universe, uint64_t ones);
void go_sparse_build_support(go_sparse *sv);

/* StringArray in memory: offsets[n+1] + bytes (src/support.rs:373-377) */
typedef struct { uint64_t n; uint64_t *offsets; uint8_t *bytes; uint64_t total; } go_strings;
void go_strings_free(go_strings *s);

/* Tags (src/support.rs:915-1020): lower-cased keys, sorted by key */
typedef struct { uint64_t n; char **keys; char **values; } go_tags;
void go_tags_free(go_tags *t);
const char *go_tags_get(const go_tags *t, const char *key);

/* PathName, src/gbwt.rs:912-926 */
typedef struct { uint32_t sample, contig, phase, fragment; } go_path_name;

/* Metadata, src/gbwt.rs:623-630 */
typedef struct {
    uint64_t flags, sample_count, haplotype_count, contig_count;
    go_path_name *path_names; uint64_t n_paths;
    go_strings sample_names; uint64_t *sample_sorted;
    go_strings contig_names; uint64_t *contig_sorted;
} go_metadata;
void go_metadata_free(go_metadata *m);

/* GBWT, src/gbwt.rs:95-102 */
struct go_gbwt {
    uint64_t sequences, size, offset, alphabet_size, flags;   /* Header<GBWTPayload> */
    go_tags tags;
    go_bwt *bwt;
    go_pos *endmarker; uint64_t endmarker_len;
    go_metadata *metadata;
};

/* Graph, src/graph.rs:84-89 */
typedef struct {
    uint64_t nodes, flags, version;
    go_strings sequences;
    go_strings segments;
    go_sparse mapping;
} go_graph;

/* GBZ, src/gbz.rs:124-130 */
struct go_gbz {
    go_tags tags;
    go_gbwt *index;
    go_graph graph;
    uint8_t *real_nodes; uint64_t potential_nodes;
};

#endif
