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

/* output_vec (output_functions.hh:20-29): every element followed by one blank, then '\n'.
 * Returns the length written (without the NUL); call with out == NULL to size the buffer. */
size_t bisbm_io_format_labels(const uint32_t *labels, size_t n, char *out, size_t cap);

void bisbm_io_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
