#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../include/flooder_host.h"
int main(int argc, char** argv) {
  int dim = argc > 1 ? atoi(argv[1]) : 5; long n = argc > 2 ? atol(argv[2]) : 200; int threads = argc > 3 ? atoi(argv[3]) : 4;
  std::mt19937_64 rng(7); std::normal_distribution<double> nd;
  std::vector<double> pts(n * dim);
  for (auto& v : pts) v = (double)(float)nd(rng);
  int32_t* cells = nullptr;
  long long nc = flooder_delaunay_nd(pts.data(), n, dim, threads, &cells);
  if (nc < 0) { printf("declined %lld\n", nc); return 0; }
  for (int k = 1; k <= dim + 1; ++k) {
    int32_t* rows = nullptr;
    long long nr = flooder_cell_faces(cells, nc, dim + 1, k, n, threads, &rows);
    if (rows) flooder_host_free(rows);
    rows = nullptr;
    long long nw = flooder_cell_faces(cells, nc, dim + 1, k, 1 << 20, threads, &rows);   // (128-bit keys from k = 4)
    printf("k=%d: %lld faces (wide base: %lld)\n", k, nr, nw);
    if (rows) flooder_host_free(rows);
  }
  std::vector<long long> t64(nc * (dim + 1));
  flooder_widen_i32(cells, nc * (dim + 1), (int64_t*)t64.data(), threads);
  std::vector<long long> out(nc);
  flooder_locate_rows((int64_t*)t64.data(), nc, dim + 1, (int64_t*)t64.data(), nc, n, (int64_t*)out.data(), threads);
  long long bad = 0; for (long long i = 0; i < nc; ++i) bad += out[i] != i;
  printf("%lld cells, locate mismatches %lld, exact %ld\n", nc, bad, flooder_delaunay_nd_stat(0));
  flooder_host_free(cells);
}
