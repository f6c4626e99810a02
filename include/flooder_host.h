/*
 * flooder_host.h - C ABI of the HOST-side pieces of the path (libflooder_host.so, plain C++ / g++, no GPU; and
 * libflooder_py.so, C against the CPython API).  They replace what the reference takes from third-party C++ libraries
 * on the host either side of the sweep: gudhi's Delaunay triangulation of the landmarks (flooder/core.py:130-138),
 * gudhi's persistence computation (flooder/cli.py:473-476, tests/test_flooder.py:55-71) and the Python loop that
 * fills the result dict (flooder/core.py:258-263, 285-288).  Bound with ctypes: flooder_amd/simplex_tree.py,
 * flooder_amd/persistence.py.
 */
#ifndef FLOODER_HOST_H
#define FLOODER_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Delaunay triangulation of n >= 5 points in THREE dimensions (csrc/delaunay3d.cpp; two dimensions: below; four to
 * eight: flooder_delaunay_nd).  Replaces
 * gudhi.DelaunayComplex(landmarks) of core.py:130-132 (the top cells; core.py:136-138 buckets their faces) and the
 * Qhull call this build made through scipy.  Incremental Bowyer-Watson with ghost tetrahedra, orient3d / insphere
 * decided exactly (double filter, then 512-bit integers on a common dyadic grid of the coordinates).
 *   pts: n x 3 float64, row-major (host).  tets: cap x 4 int32 (host), vertex ids in no particular order.
 *   returns the number of tetrahedra written; -needed (a small negative number) when cap is too small; a code below
 *   -(1 << 40) when the routine declines the input - duplicate points, all points coplanar, non-finite values,
 *   coordinates whose exponents spread over more than 57 bits (float64 clouds with full mantissas) - and the caller
 *   triangulates with Qhull instead (flooder_amd.simplex_tree.delaunay_cells does).
 */
int64_t flooder_delaunay3d(const double* pts, int64_t n, int32_t* tets, int64_t cap);

/*
 * The same in TWO dimensions (csrc/delaunay2d.cpp; n >= 3 points, the reference's 2-D clouds: annulus, figure eight):
 * ghost triangles, orient2d / incircle behind a static bound, Shewchuk's bound, then exact integers.
 *   pts: n x 2 float64, row-major (host).  tris: cap x 3 int32 (host), vertex ids counter-clockwise.
 *   returns as flooder_delaunay3d: triangles written; -needed; or a code below -(1 << 40) when it declines (duplicate
 *   points, all points collinear, non-finite values, an exponent spread of more than 57 bits).
 */
int64_t flooder_delaunay2d(const double* pts, int64_t n, int32_t* tris, int64_t cap);

/*
 * Delaunay triangulation in 2 .. 8 dimensions on all host cores (csrc/delaunay_nd.cpp) - what flood_complex uses above
 * three dimensions, where the reference's gudhi.DelaunayComplex (core.py:130-132) runs CGAL's d-dimensional
 * triangulation and this build used to run Qhull (one thread: 8 s for the 2000 6-D landmarks of BASELINE cfg 4).
 * Gift wrapping over the facets, level by level: the simplex beyond a facet of a Delaunay simplex is found by one
 * pass over the points (two linear forms per point: power to the circumsphere, barycentric coordinate), the open
 * facets of a level are spread over `n_threads` threads (<= 0: one per CPU of the affinity mask, at most 128), new
 * simplices are deduplicated and their facets matched in lock-free tables.  Floating-point forms with rigorous error
 * bounds decide almost every comparison; what they cannot decide is decided exactly over multi-word integers.
 *   pts: n x dim float64, row-major (host).  *out_cells: malloc'ed (count, dim + 1) int32, ascending vertex ids per
 *   row, rows in no particular order - release with flooder_host_free.
 *   returns the number of simplices, or a code below -(1 << 40) when the routine declines the input (duplicate
 *   points; an exact tie: dim + 2 cospherical points or dim + 1 points on a hyperplane met by a pivot; non-finite
 *   values; an exponent spread of more than 120 bits; dim outside 2 .. 8): the caller triangulates with Qhull then.
 */
int64_t flooder_delaunay_nd(const double* pts, int64_t n, int dim, int n_threads, int32_t** out_cells);
void flooder_host_free(void* p);
/* diagnostics of the last flooder_delaunay_nd call: 0 exact predicate evaluations, 1 contenders that reached the exact
 * stage, 2 threads used.  flooder_delaunay_nd_isa: test hook, forces the scan's instruction set (0 generic, 1 AVX2,
 * 2 AVX-512, -1 detect); returns the previous value. */
long flooder_delaunay_nd_stat(int what);
int flooder_delaunay_nd_isa(int isa);

/*
 * The k-vertex faces of the top cells of a complex as ONE sorted table of distinct rows, on all host cores
 * (csrc/cell_faces.cpp).  Replaces the Python loop of core.py:135-138 that buckets `stree.get_simplices()` by
 * dimension (and this build's numpy enumeration: every cell's C(width, k) faces packed into keys and np.unique'd -
 * 51 million keys for the triangles of the 6-D complex of BASELINE cfg 4).
 *   cells: n_cells x width int32, ascending vertex ids < n_points per row (host).  *out_rows: malloc'ed (count, k) int32,
 *   ascending ids per row, rows in lexicographic order - release with flooder_host_free.
 *   returns the number of distinct faces, or a code below -(1 << 40): bad arguments, or n_points^k >= 2^126 (the packed
 *   keys fit neither 64 nor 128 bits: the caller enumerates with numpy then).
 */
int64_t flooder_cell_faces(const int32_t* cells, int64_t n_cells, int width, int k, int64_t n_points, int n_threads,
                           int32_t** out_rows);

/*
 * One step of gudhi's make_filtration_non_decreasing (reference core.py:280) on all host cores: the values of the
 * dimension-d table (rows: n x k int64, k = d + 1, ascending ids) raised to the maxima of their facets, which are
 * located in the sorted dimension-(d-1) table (lower_rows: n_lo x (k-1), lexicographic order; lower_vals).  NaN facet
 * values do not take part; a NaN own value becomes the facets' maximum.  Returns the number of rows changed, or a code
 * below -(1 << 40) (n_points^(k-1) >= 2^126: the caller runs the numpy pass).
 */
int64_t flooder_raise_dimension(const int64_t* rows, int64_t n, int k, const int64_t* lower_rows, int64_t n_lo,
                                const double* lower_vals, double* vals, int64_t n_points, int n_threads);

/*
 * out[i] = the row of `table` (m x k int64, ascending ids, lexicographic order) equal to query row i (n x k), or -1:
 * binary search on packed keys on all host cores - the hand-off's lookup of simplices by their vertex tuple
 * (core.py:258-263, 278-280).  0, or a code below -(1 << 40) when n_points^k >= 2^126 (numpy path then).
 */
int64_t flooder_locate_rows(const int64_t* query, int64_t n, int k, const int64_t* table, int64_t m, int64_t n_points,
                            int64_t* out, int n_threads);

/* count int32 values widened into a caller-owned int64 array, on all cores (numpy's index dtype). */
void flooder_widen_i32(const int32_t* src, int64_t count, int64_t* dst, int n_threads);

/*
 * Test hook of flooder_delaunay3d: the new tetrahedra of an insertion are linked to each other through the edges of
 * the cavity's boundary - a table over the boundary's locally numbered vertices while there are at most
 * `max_vertices` of them (default 160), a hash table beyond.  0 forces the hash table.  Returns the previous value;
 * a value outside 0 .. 4096 only reads it.  Either way the triangulation is the same.
 */
int flooder_delaunay3d_local_edges(int max_vertices);

/*
 * Z/2 persistent homology of a filtered complex (csrc/persistence.cpp): column reduction with clearing.  Replaces
 * gudhi.SimplexTree.compute_persistence / persistence_intervals_in_dimension (cli.py:473-476).
 *   n simplices in filtration order (a face before its cofaces); dims[j] = dimension of simplex j; the boundary of j
 *   is bidx[bptr[j] .. bptr[j+1]) (indices < j).  pair[j] = the simplex j is paired with, or -1 (essential class).
 */
int flooder_persistence_z2(int64_t n, const int32_t* dims, const int64_t* bptr, const int64_t* bidx, int64_t* pair);

/*
 * The simplices of a complex in filtration order with their boundaries - the input of flooder_persistence_z2 - from the
 * per-dimension simplex tables (what gudhi's Simplex_tree hands to its persistence module, cli.py:473-476).
 *   top: highest dimension; counts[d]: simplices of dimension d; rows: the tables back to back, dimension d as
 *   (counts[d], d+1) ascending vertex ids, rows in lexicographic order; vals: the filtration values in the same order.
 *   Order: (value with NaN last, dimension, table position).  Outputs, in filtration order: dims_out, filt_out, bptr
 *   (n+1), bidx (facet j = the simplex without its j-th vertex, as positions in filtration order), order_out (global
 *   id = offset of the dimension + row).  0; -1 bad arguments; -2 a facet is missing from its table.
 */
int flooder_filtration_order(int top, const int64_t* counts, const int64_t* rows, const double* vals, int32_t* dims_out,
                             double* filt_out, int64_t* bptr, int64_t* bidx, int64_t* order_out);

/*
 * libflooder_py.so (csrc/pyhandoff.c; loaded with ctypes.PyDLL, the GIL held): dict[tuple[int, ...], float] entries
 * for n simplices of k vertices each in one pass - what core.py:258-263 / 285-288 build with zip() over .tolist().
 *   dict, cache: PyObject* (a dict; a list whose entry v is the int object v, grown as needed).  0 / -1 (exception set).
 */
int flooder_dict_update(void* dict, const int64_t* rows, int64_t n, int k, const double* vals, void* cache);

#ifdef __cplusplus
}
#endif
#endif
