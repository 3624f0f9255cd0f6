/*
 * bisbm_io.h -- host-side text I/O of the engine, format-identical with the reference's
 * graph_utilities.cc / output_functions.hh (paths relative to /root/reference/src).  Part of
 * libbisbm_hip.so; plain C ABI.  Buffers returned through ** are malloc'ed: release them with
 * bisbm_io_free.
 */
#ifndef BISBM_IO_H
#define BISBM_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* load_edge_list (graph_utilities.cc:20-34): one "a b" pair per line, any blanks between.  Same
 * quirks as the stringstream loop: a blank line repeats the previous pair, a non-numeric line
 * gives (0, previous b), a single number keeps the previous b.  Returns the number of lines, or
 * -1 when the file cannot be opened. */
long bisbm_io_read_edge_list(const char *path, uint64_t **a, uint64_t **b);

/* load_memberships (graph_utilities.cc:5-18): one label per line.  Returns the count or -1. */
long bisbm_io_read_memberships(const char *path, uint32_t **labels);

/* edge_to_adj (graph_utilities.cc:36-49) as CSR: undirected, duplicates kept, each row in edge-file
 * order.  rowptr has n+1 entries, col has 2*n_edges.  Returns 0, or -1 when an id is >= n. */
int bisbm_io_edges_to_csr(const uint64_t *a, const uint64_t *b, size_t n_edges, uint64_t n,
                          uint64_t *rowptr, uint32_t *col);

/* load_edge_list + edge_to_adj in one call, with an optional binary cache beside the text file
 * (1e7..5e7-line edge lists through a line parser are slow; the TEXT FORMAT IS UNTOUCHED and stays the
 * source of truth).  The cache is `<path>.bisbm_csr`: a 48-byte header {magic "BISBMCSR", version, size and mtime
 * (ns) of the text file, n, n_edges}, then rowptr (n+1 x u64) and col (2 n_edges x u32) exactly as
 * bisbm_io_edges_to_csr builds them.  It is used only when size, mtime, n and the file's own length all match, and
 * rewritten (temporary file + rename) otherwise; a directory that cannot be written to just means no cache.
 * use_cache = 0: always parse the text, never touch a cache file.
 * Returns 0 and malloc'ed arrays (bisbm_io_free), -1 when the text file cannot be opened, -2 when an id is >= n.
 * *cache_hit (may be NULL) tells whether the arrays came from the cache. */
int bisbm_io_load_csr(const char *path, uint64_t n, int use_cache, uint64_t **rowptr, uint32_t **col,
                      uint64_t *n_edges, int *cache_hit);

/* Optional ingest-time renumbering for graphs whose node ids carry no structure.  The production
 * kernel visits ids in an id-local order and gathers the labels of their neighbours; that pays when nearby ids have
 * nearby neighbours.  This pass embeds both node types in a few dimensions (power iteration on the degree-normalised
 * bipartite adjacency, orthonormalised every round: the leading block structure), cuts the embedding with a median
 * kd-tree shared by both types, and numbers the nodes cell by cell.  new_id[v] is the new id of node v: a bijection
 * that keeps type-a nodes in [0, na) and type-b nodes in [na, n).  Deterministic.  Purely an optimisation of memory
 * locality: the engine run on the renumbered graph is a different (equally valid) chain, so this is never applied
 * behind the caller's back.  Returns 0, or -1 on bad arguments. */
int bisbm_io_locality_order(uint64_t n, uint64_t na, const uint64_t *rowptr, const uint32_t *col, uint32_t *new_id);

/* CSR of the renumbered graph: row new_id[v] = row v with every neighbour id mapped, order inside rows kept. */
int bisbm_io_permute_csr(uint64_t n, const uint64_t *rowptr, const uint32_t *col, const uint32_t *new_id,
                         uint64_t *rowptr_out, uint32_t *col_out);

/* output_vec (output_functions.hh:20-29): every element followed by one blank, then '\n'.
 * Returns the length written (without the NUL); call with out == NULL to size the buffer. */
size_t bisbm_io_format_labels(const uint32_t *labels, size_t n, char *out, size_t cap);

void bisbm_io_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
