/*
 * gbwt_synth.h -- synthetic GBWT / GBZ generator (host-only C API of libgbwt_synth.so).
 *
 * The reference cannot build a GBWT ("no BWT construction algorithms have been implemented",
 * src/bwt.rs:209; README.md:27), so every synthetic config of BASELINE.json needs a constructor.
 * Two are provided:
 *   - gbwt_synth_chain: "bubble chain" / "star chain" pangenome: S sites, each an
 *     anchor node followed by one of A allele nodes; bidirectional GBWT built by a PBWT-style
 *     sweep, O(haplotypes) per site.  The generator keeps the allele
 *     matrix, i.e. the ground-truth paths, so extraction can be checked at full size.
 *   - gbwt_synth_from_paths: any path set (cycles, both orientations) by brute-force sorting of
 *     reverse prefixes; for small test inputs.
 * Files are written in the simple-sds format the reference loads (GBWT v5, GBZ v1 container).
 */
#ifndef GBWT_SYNTH_H
#define GBWT_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gbwt_synth gbwt_synth;

enum { GBWT_SYNTH_MOSAIC = 0, GBWT_SYNTH_IID = 1 };

/* alleles == 2: allele 1 ~ Bernoulli(p_site), p_site ~ U(0.05, 0.95); alleles > 2: Zipf(zipf) over the alleles.
 * MOSAIC: `founders` founder haplotypes carry the per-site draws and every haplotype copies a founder,
 * switching founder with probability switch_rate per site.  IID: every haplotype draws independently.
 * Node ids: site s has anchor s*(alleles+1)+1 and allele nodes s*(alleles+1)+2+a; 1 bp labels.
 * Path 0 is a generic path (sample _gbwt_ref, contig chr1); path h >= 1 is sample s{(h-1)/2}, phase (h-1)%2+1. */
gbwt_synth *gbwt_synth_chain(uint64_t sites, uint64_t haplotypes, uint32_t alleles, uint32_t model, uint32_t founders,
                             double switch_rate, double zipf, uint64_t seed);

/* The same chain with insertion alleles: every allele a >= 1 is the allele node followed by `extra` more nodes
 * (allele 0 stays one node), so paths differ in length and the walks of a batch leave lock step after the first
 * site: the regime of graphs with indels.  Node ids of site s (stride = alleles + 1 + (alleles - 1) * extra):
 * anchor s*stride+1, allele nodes s*stride+2+a, tail e of allele a >= 1 at s*stride+2+alleles+(a-1)*extra+e.
 * Only the sites s with s % indel_every == 0 carry insertions (the others are plain bubbles; their tail ids stay unused).
 * extra == 0 is gbwt_synth_chain. */
gbwt_synth *gbwt_synth_chain_indel(uint64_t sites, uint64_t haplotypes, uint32_t alleles, uint32_t model, uint32_t founders,
                                   double switch_rate, double zipf, uint64_t seed, uint32_t extra, uint32_t indel_every);

/* ... and with every logical node (anchor, allele, inserted node) chopped into a chain of `chop` nodes with consecutive ids, as a GBZ
 * built from a GFA chops long segments: most records are then unary (chop == 1 is gbwt_synth_chain_indel).  Node ids of site s
 * (stride = chop + alleles + (chop - 1) + (alleles - 1) * (chop * (1 + extra) - 1)): anchor pieces s*stride+1 .., first pieces of the alleles
 * s*stride+chop+1+a, then the remaining pieces of allele 0, of allele 1, ... */
gbwt_synth *gbwt_synth_chain_chopped(uint64_t sites, uint64_t haplotypes, uint32_t alleles, uint32_t model, uint32_t founders,
                                     double switch_rate, double zipf, uint64_t seed, uint32_t extra, uint32_t indel_every, uint32_t chop);

/* ... and with node labels of realistic lengths (label_mode 1; 0 = the 1 bp labels of the functions above): anchors 1 + Exp(40) bp, one in
 * twelve a full 1 024 bp piece; alleles 85 % single bases, else 1 + Exp(6); inserted nodes 1 + Exp(20); all capped at 1 024, seeded -- so
 * that the end coordinate of a W-line (fragment + summed label lengths, src/bin/gbunzip.rs:532-540) and the S-lines are those of a real
 * graph, not of a one-base-per-node stand-in. */
gbwt_synth *gbwt_synth_chain_labeled(uint64_t sites, uint64_t haplotypes, uint32_t alleles, uint32_t model, uint32_t founders,
                                     double switch_rate, double zipf, uint64_t seed, uint32_t extra, uint32_t indel_every, uint32_t chop, uint32_t label_mode);

/* Paths as CSR over GBWT-encoded nodes (2 * id + orientation, id >= 1).  bidirectional != 0 adds the
 * reverse sequence of every path (src/support.rs:310-314).  No metadata, no graph. */
gbwt_synth *gbwt_synth_from_paths(const uint64_t *offsets, const uint64_t *nodes, uint64_t n_paths, int bidirectional);

/* Turns a bidirectional index with alphabet offset 1 (e.g. one made by gbwt_synth_from_paths over node ids starting
 * at 1) into a GBZ: path metadata as in gbwt_synth_chain, seeded 1-3 base labels for the nodes that have a record,
 * and -- with n_segments > 0 -- a node-to-segment translation: segment k covers node ids segment_starts[k] up to the
 * next start and is named "seg<start>" (src/graph.rs:84-89, 186-218).  Returns 0 on success. */
int gbwt_synth_attach_gbz(gbwt_synth *s, const uint64_t *segment_starts, uint64_t n_segments, uint64_t seed);

/* One index out of several chains (gbwt_synth_chain*): the graph components of a whole-genome GBZ.  The node ids of part k follow those
 * of part k - 1 (its ids are shifted, its records re-encoded with the shifted edge lists), its paths follow the paths of part k - 1, and
 * the endmarker records are merged.  `path_names` = 4 numbers per path of the merged index (sample, contig, phase, fragment: PathName,
 * src/gbwt.rs:912-926) -- any assignment: several samples and phases, contigs, non-zero fragment offsets, several generic paths (sample =
 * the id of "_gbwt_ref").  The parts are only read.  Config C4's shape: contigs x fragments = parts, haplotypes = paths per part. */
gbwt_synth *gbwt_synth_merge(const gbwt_synth *const *parts, uint64_t n_parts, const uint32_t *path_names, const char *const *sample_names,
                             uint64_t n_samples, const char *const *contig_names, uint64_t n_contigs, uint64_t haplotype_count);

/* Loads a .gbwt / .gbz with the product loader (for writer round-trip tests). */
gbwt_synth *gbwt_synth_from_file(const char *path, char *err, uint64_t errlen);

void gbwt_synth_free(gbwt_synth *s);

const uint8_t *gbwt_synth_data(const gbwt_synth *s, uint64_t *len);
const uint64_t *gbwt_synth_starts(const gbwt_synth *s, uint64_t *n_records);
/* out = {sequences, size, alphabet_offset, alphabet_size, bidirectional, paths, sites, alleles} */
void gbwt_synth_header(const gbwt_synth *s, uint64_t out[8]);
int gbwt_synth_save(const gbwt_synth *s, const char *path, int as_gbz);
/* Sets (or replaces) a tag of the GBWT index that gbwt_synth_save writes (Tags, src/support.rs:870-986) -- e.g. "reference_samples",
 * which gbunzip's header line prints as RS:Z: (src/bin/gbunzip.rs:193-203). */
void gbwt_synth_set_tag(gbwt_synth *s, const char *key, const char *value);

/* Ground truth of chain generators: GBWT-encoded nodes of path `path_id` (forward orientation).
 * Returns the path length; writes at most `cap` nodes. */
uint64_t gbwt_synth_path(const gbwt_synth *s, uint64_t path_id, uint32_t *out, uint64_t cap);
/* out = {nodes of the path, decimal digits of their node ids, summed lengths of their labels}: the ground truth of the path's GFA line
 * (one token per node; a W-line ends at fragment + out[2]). */
void gbwt_synth_path_text_stats(const gbwt_synth *s, uint64_t path_id, uint64_t out[3]);
/* Sum over the path of its GBWT-encoded node values (cheap full-size checksum), computed from the allele matrix. */
uint64_t gbwt_synth_path_checksum(const gbwt_synth *s, uint64_t path_id);

#ifdef __cplusplus
}
#endif
#endif
