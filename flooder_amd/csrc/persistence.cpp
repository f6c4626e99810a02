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
#include <cmath>
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
  // reduced columns live back to back in one arena (offset, length per column): no allocation per column
  std::vector<int64_t> arena;
  arena.reserve(static_cast<size_t>(n) * 4);
  std::vector<int64_t> col_off(static_cast<size_t>(n), -1);
  std::vector<int32_t> col_len(static_cast<size_t>(n), 0);
  std::vector<int64_t> low_to_col(static_cast<size_t>(n), -1);  // pivot row -> reduced column
  std::vector<char> cleared(static_cast<size_t>(n), 0);
  std::vector<int64_t> work, tmp;
  // columns by dimension, in filtration order (a counting sort)
  std::vector<int64_t> start(static_cast<size_t>(max_dim) + 2, 0), by_dim(static_cast<size_t>(n));
  for (int64_t j = 0; j < n; ++j) ++start[static_cast<size_t>(dims[j]) + 1];
  for (int d = 0; d <= max_dim; ++d) start[static_cast<size_t>(d) + 1] += start[static_cast<size_t>(d)];
  {
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    for (int64_t j = 0; j < n; ++j) by_dim[static_cast<size_t>(fill[static_cast<size_t>(dims[j])]++)] = j;
  }
  for (int d = max_dim; d >= 1; --d) {
    for (int64_t q = start[static_cast<size_t>(d)]; q < start[static_cast<size_t>(d) + 1]; ++q) {
      const int64_t j = by_dim[static_cast<size_t>(q)];
      if (cleared[static_cast<size_t>(j)]) continue;  // (twist: j is the pivot of a (d+1)-column, its own column reduces to zero)
      work.assign(bidx + bptr[j], bidx + bptr[j + 1]);
      std::sort(work.begin(), work.end());
      while (!work.empty()) {
        const int64_t low = work.back();
        const int64_t k = low_to_col[static_cast<size_t>(low)];
        if (k < 0) break;
        const int64_t* o = arena.data() + col_off[static_cast<size_t>(k)];
        const int64_t no = col_len[static_cast<size_t>(k)];
        tmp.clear();
        std::set_symmetric_difference(work.begin(), work.end(), o, o + no, std::back_inserter(tmp));
        work.swap(tmp);
      }
      if (!work.empty()) {
        const int64_t low = work.back();
        low_to_col[static_cast<size_t>(low)] = j;
        col_off[static_cast<size_t>(j)] = static_cast<int64_t>(arena.size());
        col_len[static_cast<size_t>(j)] = static_cast<int32_t>(work.size());
        arena.insert(arena.end(), work.begin(), work.end());
        pair[low] = j;   // class born at `low` dies at j
        pair[j] = low;
        cleared[static_cast<size_t>(low)] = 1;
      }
    }
  }
  return 0;
}

// flooder_filtration_order: the simplices of a complex in filtration order with their boundaries - the input of
// flooder_persistence_z2 - from the per-dimension simplex tables (what gudhi's Simplex_tree keeps internally and hands
// to its persistence module, reference call site flooder/cli.py:473-476).  Was 14 of the 17 ms of a
// compute_persistence call on 26 k simplices as numpy code (packed keys + searchsorted per facet column, lexsort).
//   top: highest dimension; counts[d]: simplices of dimension d; rows: the tables back to back, dimension d as
//   (counts[d], d + 1) ascending vertex ids, rows in lexicographic order; vals: their filtration values in the same order.
//   Order = (filtration value with NaN last, dimension, table position).  Outputs in filtration order: dims_out,
//   filt_out, bptr (n + 1), bidx (sum over d >= 1 of counts[d] * (d + 1); facet j = the simplex without its j-th
//   vertex, as positions in filtration order), order_out (global id = offset of the dimension + row).
//   Returns 0, -1 on bad arguments, -2 when a facet is missing from its table (the complex is not closed), -3 when the
//   rows do not pack into 62-bit keys (the caller's general path takes over).
extern "C" int flooder_filtration_order(int top, const int64_t* counts, const int64_t* rows, const double* vals,
                                        int32_t* dims_out, double* filt_out, int64_t* bptr, int64_t* bidx,
                                        int64_t* order_out) {
  if (top < 0 || top > 62 || !counts || !dims_out || !filt_out || !bptr || !order_out) return -1;
  std::vector<int64_t> offs(static_cast<size_t>(top) + 2, 0), roff(static_cast<size_t>(top) + 2, 0);
  for (int d = 0; d <= top; ++d) {
    if (counts[d] < 0) return -1;
    offs[static_cast<size_t>(d) + 1] = offs[static_cast<size_t>(d)] + counts[d];
    roff[static_cast<size_t>(d) + 1] = roff[static_cast<size_t>(d)] + counts[d] * (d + 1);
  }
  const int64_t n = offs[static_cast<size_t>(top) + 1];
  if (n > 0 && (!rows || !vals)) return -1;
  // filtration order: one sort of (value with NaN last, global id) pairs - global ids grow with the dimension, then
  // with the table position
  std::vector<int32_t> dim_of(static_cast<size_t>(n));
  for (int d = 0; d <= top; ++d)
    for (int64_t i = offs[static_cast<size_t>(d)]; i < offs[static_cast<size_t>(d) + 1]; ++i) dim_of[static_cast<size_t>(i)] = d;
  struct Ent { double v; int64_t g; };
  std::vector<Ent> ent(static_cast<size_t>(n));
  for (int64_t g = 0; g < n; ++g) {
    const double v = vals[g];
    ent[static_cast<size_t>(g)] = Ent{v != v ? HUGE_VAL : v, g};
  }
  std::sort(ent.begin(), ent.end(), [](const Ent& a, const Ent& b) { return a.v != b.v ? a.v < b.v : a.g < b.g; });
  std::vector<int64_t> pos(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i) {
    order_out[i] = ent[static_cast<size_t>(i)].g;
    pos[static_cast<size_t>(order_out[i])] = i;
  }
  // every table's rows packed into one 64-bit key each (vertex ids in `base` digits) and hashed: a facet is one probe
  int64_t vmax = 0;
  for (int64_t i = 0; i < roff[static_cast<size_t>(top) + 1]; ++i) vmax = rows[i] > vmax ? rows[i] : vmax;
  const uint64_t base = static_cast<uint64_t>(vmax) + 1;
  {
    long double room = 1.0L;
    for (int k = 0; k <= top; ++k) room *= static_cast<long double>(base);
    if (room >= 4.0e18L) return -3;   // (keys would not fit 62 bits: the caller's general path)
  }
  auto pack = [&](const int64_t* r, int k, int skip) {
    uint64_t key = 0;
    for (int c = 0; c < k; ++c)
      if (c != skip) key = key * base + static_cast<uint64_t>(r[c]);
    return key;
  };
  struct Slot { uint64_t key; int64_t row; };
  std::vector<std::vector<Slot>> tabs(static_cast<size_t>(top) + 1);
  std::vector<uint64_t> masks(static_cast<size_t>(top) + 1, 0);
  for (int d = 0; d < top; ++d) {   // (the top table is nobody's facet table)
    uint64_t cap = 16;
    while (cap < 2 * static_cast<uint64_t>(counts[d]) + 2) cap <<= 1;
    tabs[static_cast<size_t>(d)].assign(cap, Slot{~0ull, -1});
    masks[static_cast<size_t>(d)] = cap - 1;
    const int64_t* t = rows + roff[static_cast<size_t>(d)];
    for (int64_t i = 0; i < counts[d]; ++i) {
      const uint64_t key = pack(t + i * (d + 1), d + 1, -1);
      uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 17 & masks[static_cast<size_t>(d)];
      while (tabs[static_cast<size_t>(d)][h].row >= 0) h = (h + 1) & masks[static_cast<size_t>(d)];
      tabs[static_cast<size_t>(d)][h] = Slot{key, i};
    }
  }
  bptr[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t g = order_out[i];
    const int d = dim_of[static_cast<size_t>(g)];
    dims_out[i] = d;
    filt_out[i] = vals[g];
    bptr[i + 1] = bptr[i] + (d > 0 ? d + 1 : 0);
    if (d == 0) continue;
    if (!bidx) return -1;
    const int64_t* r = rows + roff[static_cast<size_t>(d)] + (g - offs[static_cast<size_t>(d)]) * (d + 1);
    const std::vector<Slot>& tab = tabs[static_cast<size_t>(d) - 1];
    const uint64_t mask = masks[static_cast<size_t>(d) - 1];
    for (int j = 0; j <= d; ++j) {
      const uint64_t key = pack(r, d + 1, j);
      uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 17 & mask;
      while (tab[h].row >= 0 && tab[h].key != key) h = (h + 1) & mask;
      if (tab[h].row < 0) return -2;
      bidx[bptr[i] + j] = pos[static_cast<size_t>(offs[static_cast<size_t>(d) - 1] + tab[h].row)];
    }
  }
  return 0;
}
