/*
 * The host-side C ABI from plain C (include/flooder_host.h): Delaunay triangulation of d-dimensional landmarks on all
 * cores, the sorted table of their k-vertex faces, a row lookup - what a maintainer of the reference would call from
 * cgo / JNI / a CPython extension in place of gudhi.DelaunayComplex + the Python bucketing loop (core.py:130-138).
 *   gcc -O2 -Iinclude examples/host_abi_example.c -o /tmp/host_abi_example flooder_amd/libflooder_host.so -Wl,-rpath,$PWD/flooder_amd
 *   /tmp/host_abi_example 6 300      ->  "dim 6, 300 points: <cells> cells, <n> triangles, ..."
 */
#include <stdio.h>
#include <stdlib.h>

#include "flooder_host.h"

int main(int argc, char** argv) {
  const int dim = argc > 1 ? atoi(argv[1]) : 4;
  const long n = argc > 2 ? atol(argv[2]) : 200;
  double* pts = (double*)malloc(sizeof(double) * (size_t)n * (size_t)dim);
  unsigned long long s = 88172645463325252ull;                 /* xorshift: a reproducible cloud */
  for (long i = 0; i < n * dim; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    pts[i] = (double)(float)((double)(s >> 11) / 9007199254740992.0 - 0.5);   /* float32 values, as landmarks are */
  }
  int32_t* cells = NULL;
  const int64_t n_cells = flooder_delaunay_nd(pts, n, dim, 0, &cells);
  if (n_cells < 0) {
    printf("declined: code %lld (the caller would use Qhull / gudhi)\n", (long long)n_cells);
    return 2;
  }
  int32_t* tri = NULL;
  const int64_t n_tri = flooder_cell_faces(cells, n_cells, dim + 1, 3, n, 0, &tri);
  /* every triangle of the first cell must be found in the table */
  int64_t q[3], at = -1;
  int found = 0;
  int64_t* tri64 = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_tri * 3);
  flooder_widen_i32(tri, n_tri * 3, tri64, 0);
  for (int a = 0; a <= dim; ++a)
    for (int b = a + 1; b <= dim; ++b)
      for (int c = b + 1; c <= dim; ++c) {
        q[0] = cells[a]; q[1] = cells[b]; q[2] = cells[c];
        if (flooder_locate_rows(q, 1, 3, tri64, n_tri, n, &at, 1) == 0 && at >= 0) ++found;
      }
  printf("dim %d, %ld points: %lld cells, %lld triangles, exact predicate calls %ld, threads %ld, faces of cell 0 found %d\n", dim, n,
         (long long)n_cells, (long long)n_tri, flooder_delaunay_nd_stat(0), flooder_delaunay_nd_stat(2), found);
  flooder_host_free(cells);
  flooder_host_free(tri);
  free(tri64);
  free(pts);
  return 0;
}
