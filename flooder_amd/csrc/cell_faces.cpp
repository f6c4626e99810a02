// cell_faces.cpp - the k-vertex faces of the top cells of a complex as a sorted table of distinct rows, on all host
// cores.
//
// What the reference collects in Python from `stree.get_simplices()` (flooder/core.py:135-138: every simplex of the
// Delaunay complex bucketed by dimension) and what this build's SimplexTree enumerated with numpy: every cell's
// C(width, k) faces packed into 64-bit keys and `np.unique`d - one thread sorting 51 million keys for the triangles
// of BASELINE cfg 4 (1.47 M 6-simplices x 35), 5.6 s around a 34 ms sweep.  Here the keys go through a lock-free
// hash set (the distinct faces are few: 1.24 M), the distinct keys are bucketed, sorted and unpacked in parallel.
//
// C ABI (include/flooder_host.h): flooder_cell_faces.
#include "exact_int.hpp"
#include "host_parallel.hpp"

#include <cstdlib>
#include <exception>
#include <memory>

namespace {

// combinations of k positions out of w, lexicographic (the order of itertools.combinations)
std::vector<std::vector<int>> combos(int w, int k) {
  std::vector<std::vector<int>> out;
  std::vector<int> c((size_t)k);
  for (int i = 0; i < k; ++i) c[(size_t)i] = i;
  for (;;) {
    out.push_back(c);
    int i = k - 1;
    while (i >= 0 && c[(size_t)i] == w - k + i) --i;
    if (i < 0) break;
    ++c[(size_t)i];
    for (int j = i + 1; j < k; ++j) c[(size_t)j] = c[(size_t)j - 1] + 1;
  }
  return out;
}

using Key128 = unsigned __int128;

// bits of n_points^k: below 62 the packed key fits a uint64 (and a slot of the lock-free set), below 126 a Key128
inline int key_bits(int64_t n_points, int k) {
  long double bits = 0;
  for (int i = 0; i < k; ++i) bits += std::log2((long double)n_points);
  return bits >= 126.0L ? 126 : (int)std::ceil(bits);
}

template <class K>
inline K pack_row(const int64_t* r, int k, uint64_t base) {
  K key = 0;
  for (int j = 0; j < k; ++j) key = key * (K)base + (K)(uint64_t)r[j];
  return key;
}

// keys into ascending order: buckets over the key range (counting pass, scatter), every bucket sorted by one thread
template <class K>
void sort_keys_parallel(Pool& pool, std::vector<K>& keys, int k, uint64_t base) {
  const int64_t nd = (int64_t)keys.size();
  if (nd < 2) return;
  long double top = 1;
  for (int i = 0; i < k; ++i) top *= (long double)base;
  const int64_t nb = std::max<int64_t>(1, std::min<int64_t>(4096, nd / 1024));
  const long double scale = (long double)nb / top;
  auto bucket_of = [&](K key) {
    const int64_t b = (int64_t)((long double)key * scale);
    return b < 0 ? 0 : (b >= nb ? nb - 1 : b);
  };
  const int nt = pool.nt;
  const int64_t per = (nd + nt - 1) / nt;
  std::vector<int64_t> hist((size_t)nt * (size_t)nb, 0), start((size_t)nb + 1, 0);
  pool.run([&](int tid) {
    const int64_t a = tid * per, b = std::min(nd, a + per);
    int64_t* h = &hist[(size_t)tid * (size_t)nb];
    for (int64_t i = a; i < b; ++i) ++h[bucket_of(keys[(size_t)i])];
  });
  for (int64_t b = 0; b < nb; ++b) {
    int64_t c = 0;
    for (int t = 0; t < nt; ++t) {
      const int64_t v = hist[(size_t)t * (size_t)nb + (size_t)b];
      hist[(size_t)t * (size_t)nb + (size_t)b] = start[(size_t)b] + c;
      c += v;
    }
    start[(size_t)b + 1] = start[(size_t)b] + c;
  }
  std::vector<K> tmp((size_t)nd);
  pool.run([&](int tid) {
    const int64_t a = tid * per, b = std::min(nd, a + per);
    int64_t* h = &hist[(size_t)tid * (size_t)nb];
    for (int64_t i = a; i < b; ++i) tmp[(size_t)h[bucket_of(keys[(size_t)i])]++] = keys[(size_t)i];
  });
  pool.parallel_for(nb, 1, [&](int64_t b0, int64_t b1, int) {
    for (int64_t b = b0; b < b1; ++b) std::sort(tmp.begin() + start[(size_t)b], tmp.begin() + start[(size_t)b + 1]);
  });
  keys.swap(tmp);
}

template <class K>
int32_t* unpack_keys(Pool& pool, const std::vector<K>& keys, int k, uint64_t base) {
  const int64_t nd = (int64_t)keys.size();
  int32_t* out = (int32_t*)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nd, 1) * (size_t)k);
  if (!out) return nullptr;
  pool.parallel_for(nd, 1 << 14, [&](int64_t a, int64_t b, int) {
    for (int64_t i = a; i < b; ++i) {
      K key = keys[(size_t)i];
      for (int j = k - 1; j >= 0; --j) {
        out[i * k + j] = (int32_t)(uint64_t)(key % (K)base);
        key /= (K)base;
      }
    }
  });
  return out;
}

// Keys too wide for the 64-bit set (n_points^k >= 2^62: the 6- and 7-vertex faces of a 6-D complex over 2000
// landmarks): few faces per cell there - all keys are written out, sorted and made distinct.
int64_t cell_faces_wide(Pool& pool, const int32_t* cells, int64_t n_cells, int width, int k, uint64_t base,
                        const std::vector<std::vector<int>>& cb, int32_t** out_rows) {
  const int64_t ncb = (int64_t)cb.size();
  if (n_cells * ncb > ((int64_t)1 << 28)) return E_RANGE;
  std::vector<Key128> keys((size_t)(n_cells * ncb));
  pool.parallel_for(n_cells, 1024, [&](int64_t a, int64_t b, int) {
    for (int64_t c = a; c < b; ++c) {
      const int32_t* v = cells + c * width;
      for (int64_t f = 0; f < ncb; ++f) {
        Key128 key = 0;
        for (int i = 0; i < k; ++i) key = key * (Key128)base + (Key128)(uint32_t)v[cb[(size_t)f][(size_t)i]];
        keys[(size_t)(c * ncb + f)] = key;
      }
    }
  });
  sort_keys_parallel(pool, keys, k, base);
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  int32_t* out = unpack_keys(pool, keys, k, base);
  if (!out) return E_FEW;
  *out_rows = out;
  return (int64_t)keys.size();
}

}  // namespace

extern "C" int64_t flooder_cell_faces(const int32_t* cells, int64_t n_cells, int width, int k, int64_t n_points,
                                      int n_threads, int32_t** out_rows) try {
  if (!cells || !out_rows || n_cells < 0 || width < 1 || width > 16 || k < 1 || k > width || n_points < 1) return E_FEW;
  *out_rows = nullptr;
  // keys: the face's vertex ids as digits to the base n_points, first vertex most significant
  const int bits = key_bits(n_points, k);
  if (bits >= 126) return E_RANGE;
  const uint64_t base = (uint64_t)n_points;
  const std::vector<std::vector<int>> cb = combos(width, k);
  const int64_t ncb = (int64_t)cb.size();
  Pool pool(host_threads(n_threads));
  if (bits >= 62) return cell_faces_wide(pool, cells, n_cells, width, k, base, cb, out_rows);

  // ---- distinct keys through a lock-free set (0 = empty: keys are stored + 1); grown and refilled when it gets full
  size_t cap = 1024;
  while (cap < (size_t)std::max<int64_t>(4 * n_cells, 1024)) cap <<= 1;
  std::unique_ptr<std::atomic<uint64_t>[]> tab;
  std::atomic<int64_t> distinct{0};
  for (;;) {
    tab.reset(new std::atomic<uint64_t>[cap]);
    pool.parallel_for((int64_t)cap, 1 << 16, [&](int64_t a, int64_t b, int) {
      for (int64_t i = a; i < b; ++i) tab[(size_t)i].store(0, std::memory_order_relaxed);
    });
    distinct.store(0);
    std::atomic<int> full{0};
    const int64_t limit = (int64_t)(cap / 2);
    pool.parallel_for(n_cells, 256, [&](int64_t a, int64_t b, int) {
      int64_t mine = 0;
      for (int64_t c = a; c < b && !full.load(std::memory_order_relaxed); ++c) {
        const int32_t* v = cells + c * width;
        for (int64_t f = 0; f < ncb; ++f) {
          uint64_t key = 0;
          for (int i = 0; i < k; ++i) key = key * base + (uint64_t)(uint32_t)v[cb[(size_t)f][(size_t)i]];
          size_t slot = (size_t)mix64(key) & (cap - 1);
          for (;;) {
            uint64_t cur = tab[slot].load(std::memory_order_relaxed);
            if (cur == key + 1) break;
            if (cur == 0) {
              if (tab[slot].compare_exchange_strong(cur, key + 1, std::memory_order_relaxed)) { ++mine; break; }
              if (cur == key + 1) break;
            }
            slot = (slot + 1) & (cap - 1);
          }
        }
      }
      if (distinct.fetch_add(mine, std::memory_order_relaxed) + mine > limit) full.store(1);
    });
    if (!full.load()) break;
    cap <<= 1;
  }
  const int64_t nd = distinct.load();

  // ---- collect, sort, unpack
  std::vector<uint64_t> keys((size_t)nd);
  {
    std::atomic<int64_t> pos{0};
    pool.parallel_for((int64_t)cap, 1 << 14, [&](int64_t a, int64_t b, int) {
      uint64_t buf[512];
      int m = 0;
      auto flush = [&] {
        const int64_t at = pos.fetch_add(m, std::memory_order_relaxed);
        std::memcpy(&keys[(size_t)at], buf, sizeof(uint64_t) * (size_t)m);
        m = 0;
      };
      for (int64_t i = a; i < b; ++i) {
        const uint64_t v = tab[(size_t)i].load(std::memory_order_relaxed);
        if (v) {
          buf[m++] = v - 1;
          if (m == 512) flush();
        }
      }
      if (m) flush();
    });
  }
  tab.reset();
  sort_keys_parallel(pool, keys, k, base);
  int32_t* out = unpack_keys(pool, keys, k, base);
  if (!out) return E_FEW;
  *out_rows = out;
  return nd;
} catch (const std::exception&) {   // (out of memory: nothing may cross the C ABI)
  return E_FEW;
}

// One step of the monotone pass (gudhi Simplex_tree::make_filtration_non_decreasing, reference core.py:280) on all
// cores: every row of the dimension-d table is raised to the largest value among its facets, which are located in the
// sorted table of dimension d - 1 by binary search on packed keys (64-bit, or 128-bit where n_points^(k-1) >= 2^62).
// A NaN facet value does not take part, a NaN own value is replaced by the facets' maximum, a row none of whose facets
// is found (or all are NaN) keeps its value.  Returns the number of rows whose value changed, or a code below
// -(1 << 40) (keys that do not fit 126 bits).
namespace {
template <class K>
int64_t raise_dimension(Pool& pool, const int64_t* rows, int64_t n, int k, const int64_t* lower_rows, int64_t n_lo,
                        const double* lower_vals, double* vals, uint64_t base) {
  std::vector<K> keys((size_t)n_lo);
  pool.parallel_for(n_lo, 1 << 14, [&](int64_t a, int64_t b, int) {
    for (int64_t i = a; i < b; ++i) keys[(size_t)i] = pack_row<K>(lower_rows + i * (k - 1), k - 1, base);
  });
  std::atomic<int64_t> changed{0};
  pool.parallel_for(n, 1 << 12, [&](int64_t a, int64_t b, int) {
    int64_t mine = 0;
    for (int64_t i = a; i < b; ++i) {
      const int64_t* r = rows + i * k;
      double face_max = -HUGE_VAL;
      for (int omit = 0; omit < k; ++omit) {
        K key = 0;
        for (int j = 0; j < k; ++j)
          if (j != omit) key = key * (K)base + (K)(uint64_t)r[j];
        const K* it = std::lower_bound(keys.data(), keys.data() + n_lo, key);
        if (it == keys.data() + n_lo || *it != key) continue;
        const double v = lower_vals[it - keys.data()];
        if (v == v && v > face_max) face_max = v;
      }
      const double own = vals[i];
      double raised = own != own ? face_max : (face_max > own ? face_max : own);
      if (raised == -HUGE_VAL) raised = own;
      if (!(raised == own || (raised != raised && own != own))) {
        vals[i] = raised;
        ++mine;
      }
    }
    changed.fetch_add(mine, std::memory_order_relaxed);
  });
  return changed.load();
}

template <class K>
void locate_rows(Pool& pool, const int64_t* query, int64_t n, int k, const int64_t* table, int64_t m, int64_t n_points,
                 int64_t* out, uint64_t base) {
  std::vector<K> keys((size_t)m);
  pool.parallel_for(m, 1 << 14, [&](int64_t a, int64_t b, int) {
    for (int64_t i = a; i < b; ++i) keys[(size_t)i] = pack_row<K>(table + i * k, k, base);
  });
  pool.parallel_for(n, 1 << 13, [&](int64_t a, int64_t b, int) {
    for (int64_t i = a; i < b; ++i) {
      const int64_t* r = query + i * k;
      bool in_range = true;
      for (int j = 0; j < k; ++j) in_range &= r[j] >= 0 && r[j] < n_points;
      int64_t at = -1;
      if (in_range) {
        const K key = pack_row<K>(r, k, base);
        const K* it = std::lower_bound(keys.data(), keys.data() + m, key);
        if (it != keys.data() + m && *it == key) at = it - keys.data();
      }
      out[i] = at;
    }
  });
}
}  // namespace

extern "C" int64_t flooder_raise_dimension(const int64_t* rows, int64_t n, int k, const int64_t* lower_rows, int64_t n_lo,
                                           const double* lower_vals, double* vals, int64_t n_points, int n_threads) try {
  if (!rows || !lower_rows || !lower_vals || !vals || n < 0 || n_lo < 0 || k < 2 || k > 16 || n_points < 1) return E_FEW;
  const int bits = key_bits(n_points, k - 1);
  if (bits >= 126) return E_RANGE;
  Pool pool(n * k < (1 << 16) ? 1 : host_threads(n_threads));
  return bits < 62 ? raise_dimension<uint64_t>(pool, rows, n, k, lower_rows, n_lo, lower_vals, vals, (uint64_t)n_points)
                   : raise_dimension<Key128>(pool, rows, n, k, lower_rows, n_lo, lower_vals, vals, (uint64_t)n_points);
} catch (const std::exception&) {
  return E_FEW;
}

// Row of `table` (m x k int64, ascending ids per row, rows in lexicographic order) that equals each query row
// (n x k, ascending ids), -1 if there is none: binary search on packed keys, on all cores.  What numpy's searchsorted
// does on one thread for the three million (triangle, edge) pairs of cfg 4's hand-off (core.py:258-263: the dict
// update by simplex tuple).  Returns 0, or a code below -(1 << 40) (keys that do not fit 126 bits).
extern "C" int64_t flooder_locate_rows(const int64_t* query, int64_t n, int k, const int64_t* table, int64_t m,
                                       int64_t n_points, int64_t* out, int n_threads) try {
  if (!query || !table || !out || n < 0 || m < 0 || k < 1 || k > 16 || n_points < 1) return E_FEW;
  const int bits = key_bits(n_points, k);
  if (bits >= 126) return E_RANGE;
  Pool pool((n + m) * k < (1 << 17) ? 1 : host_threads(n_threads));
  if (bits < 62) locate_rows<uint64_t>(pool, query, n, k, table, m, n_points, out, (uint64_t)n_points);
  else locate_rows<Key128>(pool, query, n, k, table, m, n_points, out, (uint64_t)n_points);
  return 0;
} catch (const std::exception&) {
  return E_FEW;
}

// int32 rows widened into a caller-owned int64 array on all cores (the tables above are handed over as int32; numpy
// wants int64 index arrays, and its single-threaded astype of 10 million entries costs as much as building them)
extern "C" void flooder_widen_i32(const int32_t* src, int64_t count, int64_t* dst, int n_threads) {
  if (!src || !dst || count <= 0) return;
  Pool pool(count < (1 << 18) ? 1 : host_threads(n_threads));
  pool.parallel_for(count, 1 << 16, [&](int64_t a, int64_t b, int) {
    for (int64_t i = a; i < b; ++i) dst[i] = src[i];
  });
}
