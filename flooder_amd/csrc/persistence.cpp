// persistence.cpp - Z/2 persistent homology of a filtered simplicial complex (host C++, no GPU).
//
// Replaces what the reference obtains from gudhi's C++ Simplex_tree (third-party; reference call sites
// flooder/cli.py:473-476, tests/test_flooder.py:55-71: stree.compute_persistence(),
// persistence_intervals_in_dimension(i)).  Standard column reduction of the boundary matrix in filtration
// order with the "twist" clearing optimisation (columns of simplices that are already known to be
// negative are never reduced).  Embedded complexes in R^2 / R^3 carry no torsion, so Z/2 intervals equal the
// Z/11 intervals gudhi computes by default.
//
// C ABI:  int flooder_persistence_z2(n, dims[n], bptr[n+1], bidx[bptr[n]], pair[n])
//   simplices are given in filtration order (a face always before its cofaces); the boundary of simplex j
//   is bidx[bptr[j] .. bptr[j+1]) (indices < j, any order).  On return pair[j] = index of the simplex j is
//   paired with (birth <-> death), or -1 if j is unpaired (an essential class is born at j).
#include <algorithm>
#include <cstdint>
#include <vector>

extern "C" int flooder_persistence_z2(int64_t n, const int32_t* dims, const int64_t* bptr,
                                      const int64_t* bidx, int64_t* pair) {
  if (n < 0 || (n > 0 && (!dims || !bptr || !pair))) return -1;
  int max_dim = 0;
  for (int64_t j = 0; j < n; ++j) {
    pair[j] = -1;
    if (dims[j] > max_dim) max_dim = dims[j];
  }
  std::vector<std::vector<int64_t>> col(static_cast<size_t>(n));
  std::vector<int64_t> low_to_col(static_cast<size_t>(n), -1);  // pivot row -> reduced column
  std::vector<char> cleared(static_cast<size_t>(n), 0);
  std::vector<int64_t> tmp;
  // by_dim[d] = columns of dimension d in filtration order
  std::vector<std::vector<int64_t>> by_dim(static_cast<size_t>(max_dim) + 1);
  for (int64_t j = 0; j < n; ++j) by_dim[static_cast<size_t>(dims[j])].push_back(j);
  for (int d = max_dim; d >= 1; --d) {
    for (int64_t j : by_dim[static_cast<size_t>(d)]) {
      if (cleared[static_cast<size_t>(j)]) continue;  // j is a death-creating pivot of a (d+1)-column: negative... cleared
      std::vector<int64_t>& c = col[static_cast<size_t>(j)];
      c.assign(bidx + bptr[j], bidx + bptr[j + 1]);
      std::sort(c.begin(), c.end());
      while (!c.empty()) {
        const int64_t low = c.back();
        const int64_t k = low_to_col[static_cast<size_t>(low)];
        if (k < 0) break;
        const std::vector<int64_t>& o = col[static_cast<size_t>(k)];
        tmp.clear();
        std::set_symmetric_difference(c.begin(), c.end(), o.begin(), o.end(), std::back_inserter(tmp));
        c.swap(tmp);
      }
      if (!c.empty()) {
        const int64_t low = c.back();
        low_to_col[static_cast<size_t>(low)] = j;
        pair[low] = j;   // class born at `low` dies at j
        pair[j] = low;
        cleared[static_cast<size_t>(low)] = 1;  // twist: the column of `low` (dimension d-1) reduces to zero
      } else {
        std::vector<int64_t>().swap(c);
      }
    }
  }
  return 0;
}
