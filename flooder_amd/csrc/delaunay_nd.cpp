// delaunay_nd.cpp - Delaunay triangulation of the landmarks in 2 .. 8 dimensions on ALL host cores (no GPU).
//
// Replaces what the reference obtains from gudhi.DelaunayComplex (CGAL; call site flooder/core.py:130-138) above three
// dimensions, where this build used to ask Qhull: ONE thread, 8 s for the 2000 landmarks of BASELINE cfg 4 (6-D,
// 1.49 M 6-simplices) around a 34 ms device sweep.
//
// Algorithm: gift wrapping over the facets, level by level.  A Delaunay simplex F and one of its facets determine
// the simplex on the other side: among the points q strictly beyond the facet it is the one whose sphere through the
// facet is the first the pencil of spheres through the facet reaches when it leaves F's circumsphere, i.e. the q that
// minimises  power_F(q) / (-lambda_k(q))   (power_F: power of q with respect to the circumsphere of F, >= 0 for every
// point of a Delaunay simplex; lambda_k: barycentric coordinate of q in F belonging to the vertex opposite the facet,
// negative beyond it).  Both are LINEAR in (q, |q|^2): one pass over the points per simplex - d multiply-adds for the
// power and d per open facet - in vector code over a structure-of-arrays copy of the points.  Every pivot is
// independent of every other one: the open facets of a level are spread over the threads, the simplices they produce
// are deduplicated and their facets matched in lock-free hash tables, what stays unmatched is the next level.  (An
// incremental insertion does a tenth of the arithmetic and all of it in sequence; this does ten times the arithmetic
// in perfectly parallel dot products - 7 core-seconds at cfg 4: 1.2 s on the 8 CPUs of the build container, 0.35 s on
// 16 - 24 threads of the GPU box's host; Qhull: 21 s / 8 s on one.)
//
// Exactness: the linear forms come from a floating-point inverse of the simplex's edge matrix with RIGOROUS error
// bounds (a-posteriori: the residual I - U X bounds |U^-1 - X|); a point whose interval may beat the best upper bound
// seen so far is a contender, and where more than one contender is left - or a point cannot be put on one side of the
// facet - the comparison is decided EXACTLY: the coordinates are dyadic rationals, scaled once to integers (at most 121
// bits), the determinants behind power and lambda are evaluated over multi-word integers.  Points in general position
// have a unique Delaunay triangulation: the result equals Qhull's and CGAL's as a set of simplices
// (tests/test_delaunay_nd.py; pinned to gudhi's own output on the reference's committed clouds).  An exact tie (d + 2 cospherical points, d + 1 points on a hyperplane) is declined - the
// caller then uses Qhull, as with the 2-D / 3-D routines.
//
// C ABI (include/flooder_host.h):  flooder_delaunay_nd, flooder_host_free.
#include "exact_int.hpp"
#include "host_parallel.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <cstdlib>
#include <exception>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>

#include <sched.h>

namespace {

constexpr int64_t E_DEGENERATE = E_BASE - 7, E_INCONSISTENT = E_BASE - 8, E_DIM = E_BASE - 9, E_TOO_MANY = E_BASE - 10;
constexpr double EPS = 1.1102230246251565e-16;   // 2^-53
constexpr double SLACK = 1.000001;                // room for the roundings of the bounds themselves

// ------------------------------------------------------------------------------------------------ multi-word integers
// sign + magnitude, up to 40 words (2560 bits: a 9 x 9 determinant of 122-bit differences and 247-bit lifts times an
// 8 x 8 one stays below 2250)
struct Wide {
  static constexpr int CAP = 40;
  int len = 0;        // used words (0: the value zero)
  bool neg = false;
  uint64_t w[CAP];
};
inline void wide_trim(Wide& a) {
  while (a.len > 0 && a.w[a.len - 1] == 0) --a.len;
  if (a.len == 0) a.neg = false;
}
inline Wide wide_from(__int128 v) {
  Wide r;
  r.neg = v < 0;
  unsigned __int128 m = v < 0 ? (unsigned __int128)(-(v + 1)) + 1 : (unsigned __int128)v;
  r.w[0] = (uint64_t)m;
  r.w[1] = (uint64_t)(m >> 64);
  r.len = 2;
  wide_trim(r);
  return r;
}
inline int wide_cmp_mag(const Wide& a, const Wide& b) {
  if (a.len != b.len) return a.len < b.len ? -1 : 1;
  for (int i = a.len - 1; i >= 0; --i)
    if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
  return 0;
}
inline void wide_add_mag(const Wide& a, const Wide& b, Wide& r) {   // |a| + |b|
  const Wide& x = a.len >= b.len ? a : b;
  const Wide& y = a.len >= b.len ? b : a;
  unsigned __int128 c = 0;
  for (int i = 0; i < x.len; ++i) {
    c += (unsigned __int128)x.w[i] + (i < y.len ? y.w[i] : 0);
    r.w[i] = (uint64_t)c;
    c >>= 64;
  }
  r.len = x.len;
  if (c && r.len < Wide::CAP) r.w[r.len++] = (uint64_t)c;
}
inline void wide_sub_mag(const Wide& a, const Wide& b, Wide& r) {   // |a| - |b|, |a| >= |b|
  uint64_t borrow = 0;
  for (int i = 0; i < a.len; ++i) {
    const uint64_t bi = i < b.len ? b.w[i] : 0;
    const uint64_t d = a.w[i] - bi - borrow;
    borrow = (a.w[i] < bi) || (a.w[i] == bi && borrow) ? 1 : 0;
    r.w[i] = d;
  }
  r.len = a.len;
}
inline Wide wide_add(const Wide& a, const Wide& b) {
  Wide r;
  if (a.neg == b.neg) {
    wide_add_mag(a, b, r);
    r.neg = a.neg;
  } else {
    const int c = wide_cmp_mag(a, b);
    if (c == 0) return Wide{};
    if (c > 0) { wide_sub_mag(a, b, r); r.neg = a.neg; } else { wide_sub_mag(b, a, r); r.neg = b.neg; }
  }
  wide_trim(r);
  return r;
}
inline Wide wide_neg(Wide a) {
  if (a.len) a.neg = !a.neg;
  return a;
}
inline Wide wide_sub(const Wide& a, const Wide& b) { return wide_add(a, wide_neg(b)); }
inline Wide wide_mul(const Wide& a, const Wide& b) {
  Wide r;
  if (a.len == 0 || b.len == 0) return r;
  const int n = std::min(a.len + b.len, (int)Wide::CAP);
  for (int i = 0; i < n; ++i) r.w[i] = 0;
  for (int i = 0; i < a.len; ++i) {
    unsigned __int128 c = 0;
    for (int j = 0; j < b.len && i + j < n; ++j) {
      c += (unsigned __int128)a.w[i] * b.w[j] + r.w[i + j];
      r.w[i + j] = (uint64_t)c;
      c >>= 64;
    }
    if (i + b.len < n) r.w[i + b.len] = (uint64_t)c;
  }
  r.len = n;
  r.neg = a.neg != b.neg;
  wide_trim(r);
  return r;
}
inline int wide_sign(const Wide& a) { return a.len == 0 ? 0 : (a.neg ? -1 : 1); }

// determinant of an m x m matrix of Wide (row-major, m <= 9) by Laplace expansion over column subsets: row r is
// expanded against the minors of the rows before it - m 2^(m-1) products, ring operations only
Wide wide_det(const Wide* a, int m, std::vector<Wide>& dp) {
  dp.assign((size_t)1 << m, Wide{});
  dp[0] = wide_from(1);
  for (unsigned mask = 1; mask < (1u << m); ++mask) {
    const int r = __builtin_popcount(mask) - 1;     // this minor uses rows 0 .. r and the columns of mask
    Wide acc;
    int pos = 0;                                    // position of column j among the columns of mask
    for (int j = 0; j < m; ++j) {
      if (!(mask >> j & 1)) continue;
      // expanding along the LAST row (r): sign (-1)^(r + pos)
      const Wide term = wide_mul(a[r * m + j], dp[mask & ~(1u << j)]);
      acc = ((r + pos) & 1) ? wide_sub(acc, term) : wide_add(acc, term);
      ++pos;
    }
    dp[mask] = acc;
  }
  return dp[(1u << m) - 1];
}

// ------------------------------------------------------------------------------------------------ the triangulation
constexpr int BLK = 128;   // points per block of the scan
constexpr int GRP = 16;    // points per group: the scan reports which groups of a block hold a possible contender

// Floating-point linear forms of one simplex over the homogeneous point (x_1 .. x_D, |x|^2, 1), x relative to the
// centre of the cloud's box, with rigorous error bounds:
//   lambda_k(x) = ck[k] + sum_j x_j a[k][j]          barycentric coordinate of vertex position k
//   power(x)    = |x|^2 + sum_j x_j b[j] + cp        power with respect to the circumsphere
template <int D>
struct Geometry {
  double a[D + 1][D], ck[D + 1];
  double b[D], cp;
  double e_lam[D + 1];       // |computed lambda_k - lambda_k| <= e_lam[k] for every point of the cloud
  double e_pow;              // the same for the power
  bool usable = false;       // false: the residual does not bound the inverse (a sliver): everything goes exact
};

template <int D>
struct Engine {
  static constexpr int V = D + 1;
  int64_t n = 0, npad = 0;
  const double* p = nullptr;             // (n, D) row-major, as given
  std::vector<double> pc;                // (n, D) row-major, relative to the centre of the box
  std::vector<double> xs;                // structure of arrays of pc: coordinate j of point i at xs[j * npad + i]
  std::vector<double> sq;                // |pc_i|^2 (npad)
  std::vector<__int128> qi;              // integer coordinates on the common dyadic grid (exact stage; up to 121 bits)
  double ext[D];                         // upper bound of |q_j - v_j| over the box
  double amax[D];                        // upper bound of |pc_ij|
  std::vector<int32_t> verts;            // simplices: V ascending vertex ids each
  std::atomic<int> error{0};
  std::atomic<long> exact_calls{0}, contenders_total{0};

  const double* P(int i) const { return pc.data() + (size_t)i * D; }

  // ---- exact predicates (rare).  base b: a vertex position other than k.
  struct ExactCtx {
    std::vector<Wide> dp, mat;
  };
  void rows_from(const int32_t* vs, int k, int q, int b, Wide* mat_g, Wide* mat_h) const {
    // mat_g (D x D): rows v_i - v_b for the positions i != b in order, the row of position k replaced by q - v_b
    // mat_h ((D+1) x (D+1)): rows [v_i - v_b, |v_i - v_b|^2] for i != b in order, then [q - v_b, |q - v_b|^2]
    const __int128* B = &qi[(size_t)vs[b] * D];
    int r = 0;
    for (int i = 0; i < V; ++i) {
      if (i == b) continue;
      const __int128* A = &qi[(size_t)vs[i] * D];
      const __int128* Q = &qi[(size_t)q * D];
      Wide lift;
      for (int j = 0; j < D; ++j) {
        const __int128 d = A[j] - B[j];
        const Wide dj = wide_from(d);
        if (mat_h) {
          mat_h[r * (D + 1) + j] = dj;
          lift = wide_add(lift, wide_mul(dj, dj));
        }
        if (mat_g) mat_g[r * D + j] = (i == k) ? wide_from(Q[j] - B[j]) : dj;
      }
      if (mat_h) mat_h[r * (D + 1) + D] = lift;
      ++r;
    }
    if (mat_h) {
      const __int128* Q = &qi[(size_t)q * D];
      Wide lift;
      for (int j = 0; j < D; ++j) {
        const Wide dj = wide_from(Q[j] - B[j]);
        mat_h[D * (D + 1) + j] = dj;
        lift = wide_add(lift, wide_mul(dj, dj));
      }
      mat_h[D * (D + 1) + D] = lift;
    }
  }
  Wide det_edges(const int32_t* vs, int b, ExactCtx& cx) const {   // det of the rows v_i - v_b (i != b, in order)
    cx.mat.assign((size_t)D * D, Wide{});
    rows_from(vs, -1, vs[b], b, cx.mat.data(), nullptr);
    return wide_det(cx.mat.data(), D, cx.dp);
  }
  Wide det_g(const int32_t* vs, int k, int q, int b, ExactCtx& cx) const {
    cx.mat.assign((size_t)D * D, Wide{});
    rows_from(vs, k, q, b, cx.mat.data(), nullptr);
    return wide_det(cx.mat.data(), D, cx.dp);
  }
  Wide det_h(const int32_t* vs, int q, int b, ExactCtx& cx) const {
    cx.mat.assign((size_t)(D + 1) * (D + 1), Wide{});
    rows_from(vs, -1, q, b, nullptr, cx.mat.data());
    return wide_det(cx.mat.data(), D + 1, cx.dp);
  }
  // sign of lambda_k(q): -1 strictly beyond the facet opposite position k
  int exact_side(const int32_t* vs, int k, int q, ExactCtx& cx) {
    exact_calls.fetch_add(1, std::memory_order_relaxed);
    const int b = k == 0 ? 1 : 0;
    return wide_sign(det_g(vs, k, q, b, cx)) * wide_sign(det_edges(vs, b, cx));
  }
  // sign of power_F(q) (0: on the circumsphere)
  int exact_power_sign(const int32_t* vs, int q, ExactCtx& cx) {
    exact_calls.fetch_add(1, std::memory_order_relaxed);
    return wide_sign(det_h(vs, q, 0, cx)) * wide_sign(det_edges(vs, 0, cx));
  }
  // both beyond facet k: -1 if q1's ratio power / (-lambda_k) is smaller, +1 if q2's, 0 on a tie
  int exact_compare(const int32_t* vs, int k, int q1, int q2, ExactCtx& cx) {
    exact_calls.fetch_add(1, std::memory_order_relaxed);
    const int b = k == 0 ? 1 : 0;
    const Wide g1 = det_g(vs, k, q1, b, cx), g2 = det_g(vs, k, q2, b, cx);
    const Wide h1 = det_h(vs, q1, b, cx), h2 = det_h(vs, q2, b, cx);
    // ratio = -H / g with g1, g2 of one sign:  ratio1 < ratio2  <=>  H2 g1 - H1 g2 < 0
    return wide_sign(wide_sub(wide_mul(h2, g1), wide_mul(h1, g2)));
  }

  // ---- floating-point geometry of a simplex
  void geometry(const int32_t* vs, Geometry<D>& G) const {
    double U[D][D], A[D][2 * D], X[D][D];
    const double* v0 = P(vs[0]);
    for (int i = 0; i < D; ++i) {
      const double* a = P(vs[i + 1]);
      for (int j = 0; j < D; ++j) {
        U[i][j] = a[j] - v0[j];
        A[i][j] = U[i][j];
        A[i][D + j] = i == j ? 1.0 : 0.0;
      }
    }
    G.usable = false;
    // X ~ U^-1 by Gauss-Jordan with partial pivoting
    for (int c = 0; c < D; ++c) {
      int piv = c;
      for (int r = c + 1; r < D; ++r)
        if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
      if (A[piv][c] == 0.0) return;
      if (piv != c)
        for (int j = 0; j < 2 * D; ++j) std::swap(A[c][j], A[piv][j]);
      const double inv = 1.0 / A[c][c];
      for (int j = 0; j < 2 * D; ++j) A[c][j] *= inv;
      for (int r = 0; r < D; ++r) {
        if (r == c) continue;
        const double f = A[r][c];
        if (f == 0.0) continue;
        for (int j = 0; j < 2 * D; ++j) A[r][j] -= f * A[c][j];
      }
    }
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) X[i][j] = A[i][D + j];
    // residual R = I - U X and its bound (the computed entries are off by at most (D + 3) eps (|U||X| + 1), which also
    // covers differences u that are themselves rounded): |U^-1 - X| <= |X| rho / (1 - rho) in the infinity norm
    double rho = 0.0, norm_x = 0.0;
    for (int i = 0; i < D; ++i) {
      double row = 0.0;
      for (int k = 0; k < D; ++k) {
        double s = i == k ? 1.0 : 0.0, t = 0.0;
        for (int j = 0; j < D; ++j) {
          s -= U[i][j] * X[j][k];
          t += std::fabs(U[i][j] * X[j][k]);
        }
        row += std::fabs(s) + (D + 3) * EPS * (t + 1.0);
      }
      rho = std::max(rho, row);
    }
    for (int i = 0; i < D; ++i) {
      double row = 0.0;
      for (int j = 0; j < D; ++j) row += std::fabs(X[i][j]);
      norm_x = std::max(norm_x, row);
    }
    if (!(rho < 0.25) || !std::isfinite(norm_x)) return;
    const double eta = norm_x * rho / (1.0 - rho) * SLACK;     // |U^-1 - X| <= eta entrywise
    double ext_sum = 0.0;
    for (int j = 0; j < D; ++j) ext_sum += ext[j];
    // lambda_k = (x - v0) . X[:, k-1], k >= 1, evaluated as ck + x . a_k with ck = -v0 . X[:, k-1]: the error of the
    // inverse acts on x - v0 (at most ext), the roundings on terms bounded through amax
    double e_sum = 0.0;
    for (int k = 0; k < D; ++k) {
      double s = 0.0, c = 0.0;
      for (int j = 0; j < D; ++j) {
        G.a[k + 1][j] = X[j][k];
        c -= v0[j] * X[j][k];
        s += amax[j] * std::fabs(X[j][k]);
      }
      G.ck[k + 1] = c;
      G.e_lam[k + 1] = (ext_sum * eta + (2 * D + 6) * EPS * s) * SLACK;
      e_sum += G.e_lam[k + 1];
    }
    // lambda_0 = 1 - sum_k lambda_k
    {
      double s = 0.0, c = 1.0;
      for (int j = 0; j < D; ++j) {
        double acc = 0.0, abs_acc = 0.0;
        for (int k = 0; k < D; ++k) {
          acc -= X[j][k];
          abs_acc += std::fabs(X[j][k]);
        }
        G.a[0][j] = acc;
        s += amax[j] * abs_acc;
      }
      for (int k = 0; k < D; ++k) c -= G.ck[k + 1];
      G.ck[0] = c;
      G.e_lam[0] = (e_sum + (4 * D + 12) * EPS * (1.0 + s)) * SLACK;
    }
    // circumsphere: c2 = X l (twice the centre relative to v0), l_i = |u_i|^2;
    // power(x) = |x - v0|^2 - (x - v0) . c2 = |x|^2 - x . (2 v0 + c2) + v0 . (v0 + c2)
    double l[D], el[D];
    for (int i = 0; i < D; ++i) {
      double s = 0.0;
      for (int j = 0; j < D; ++j) s += U[i][j] * U[i][j];
      l[i] = s;
      el[i] = (D + 3) * EPS * s;
    }
    double ep = 0.0, cp = 0.0;
    for (int j = 0; j < D; ++j) {
      double s = 0.0, es = 0.0;
      for (int i = 0; i < D; ++i) {
        s += X[j][i] * l[i];
        es += eta * l[i] + std::fabs(X[j][i]) * (el[i] + (D + 2) * EPS * l[i]);
      }
      G.b[j] = -(2.0 * v0[j] + s);
      cp += v0[j] * (v0[j] + s);
      ep += ext[j] * es + (4 * D + 16) * EPS * amax[j] * (2.0 * amax[j] + std::fabs(s));
    }
    G.cp = cp;
    G.e_pow = ep * SLACK;
    G.usable = std::isfinite(G.e_pow) && std::isfinite(G.e_lam[0]);
  }
  double lam_of(const Geometry<D>& G, int k, int q) const {
    double l = G.ck[k];
    for (int j = 0; j < D; ++j) l += xs[(size_t)j * npad + (size_t)q] * G.a[k][j];
    return l;
  }
  double pow_of(const Geometry<D>& G, int q) const {
    double s = sq[(size_t)q] + G.cp;
    for (int j = 0; j < D; ++j) s += xs[(size_t)j * npad + (size_t)q] * G.b[j];
    return s;
  }

  // ---- one pivot result per (simplex, open facet)
  struct Contender { int q; double lo, hi; };
  struct alignas(128) Scan {          // per thread (aligned: the threads' states must not share cache lines)
    std::vector<Contender> cont[D + 1];
    double ustar[D + 1];
    long flagged = 0, pivots = 0;
    ExactCtx cx;
    char pad[128];
  };

  // One block of points against the forms of one simplex: numc = max(0, power - e_pow) (+inf for the simplex's own
  // vertices and the padding: never a candidate), and per open facet the groups of GRP points that hold a point which
  // may lie beyond the facet (lambda < e_lam) with a lower bound of its ratio not above the best upper bound so far.
  static inline __attribute__((always_inline)) void scan_block(const double* __restrict xs, const double* __restrict sq,
                                                               int64_t npad, int64_t base, int cnt,
                                                               const Geometry<D>& G, const int* ks, int nk,
                                                               const double* ustar, unsigned* hits,
                                                               double* __restrict numc, double (*__restrict lbuf)[BLK],
                                                               const int32_t* own) {
    const int full = (cnt + GRP - 1) / GRP * GRP;     // (the arrays are padded to whole blocks)
    const double* x[D];
    for (int j = 0; j < D; ++j) x[j] = xs + (size_t)j * npad + base;
    const double* s2 = sq + base;
    for (int i = 0; i < full; ++i) {
      double s = s2[i] + G.cp;
      for (int j = 0; j < D; ++j) s += x[j][i] * G.b[j];
      const double t = s - G.e_pow;
      numc[i] = t > 0.0 ? t : 0.0;
    }
    for (int i = cnt; i < full; ++i) numc[i] = std::numeric_limits<double>::infinity();
    for (int t = 0; t < D + 1; ++t) {
      const int64_t o = (int64_t)own[t] - base;
      if (o >= 0 && o < cnt) numc[o] = std::numeric_limits<double>::infinity();
    }
    for (int a = 0; a < nk; ++a) {
      const int k = ks[a];
      double col[D];
      for (int j = 0; j < D; ++j) col[j] = G.a[k][j];
      const double c0 = G.ck[k], el = G.e_lam[k], us = ustar[k];
      unsigned mask = 0;
      for (int g = 0; g < full / GRP; ++g) {
        unsigned any = 0;
        for (int i = g * GRP; i < (g + 1) * GRP; ++i) {
          double l = c0;
          for (int j = 0; j < D; ++j) l += x[j][i] * col[j];
          lbuf[a][i] = l;
          any |= (unsigned)((l < el) & (numc[i] <= us * (el - l)));
        }
        mask |= (any ? 1u : 0u) << g;
      }
      hits[a] = mask;
    }
  }
  __attribute__((target("avx512f,avx512dq,avx512vl,fma"))) static void scan_block_avx512(
      const double* xs, const double* sq, int64_t npad, int64_t base, int cnt, const Geometry<D>& G, const int* ks, int nk,
      const double* ustar, unsigned* hits, double* numc, double (*lbuf)[BLK], const int32_t* own) {
    scan_block(xs, sq, npad, base, cnt, G, ks, nk, ustar, hits, numc, lbuf, own);
  }
  __attribute__((target("avx2,fma"))) static void scan_block_avx2(const double* xs, const double* sq, int64_t npad,
                                                                  int64_t base, int cnt, const Geometry<D>& G,
                                                                  const int* ks, int nk, const double* ustar,
                                                                  unsigned* hits, double* numc, double (*lbuf)[BLK],
                                                                  const int32_t* own) {
    scan_block(xs, sq, npad, base, cnt, G, ks, nk, ustar, hits, numc, lbuf, own);
  }
  static void scan_block_generic(const double* xs, const double* sq, int64_t npad, int64_t base, int cnt,
                                 const Geometry<D>& G, const int* ks, int nk, const double* ustar, unsigned* hits,
                                 double* numc, double (*lbuf)[BLK], const int32_t* own) {
    scan_block(xs, sq, npad, base, cnt, G, ks, nk, ustar, hits, numc, lbuf, own);
  }
  using ScanFn = void (*)(const double*, const double*, int64_t, int64_t, int, const Geometry<D>&, const int*, int,
                          const double*, unsigned*, double*, double (*)[BLK], const int32_t*);
  ScanFn scan_fn = scan_block_generic;

  // a point the vector stage could not rule out (or one of the seed group): scalar look with the current bound
  inline void consider(const int32_t* vs, const Geometry<D>& G, int k, int q, double numc_q, double l, Scan& sc) {
    const double el = G.e_lam[k];
    if (!(l < el)) return;
    if (!(numc_q <= sc.ustar[k] * (el - l))) return;
    ++sc.flagged;
    const double lo = numc_q / (el - l);
    double hi = std::numeric_limits<double>::infinity();
    if (-l - el > 0.0) {
      hi = (numc_q + 2.0 * G.e_pow) / (-l - el) * SLACK;   // (numc + 2 e_pow >= power + e_pow)
    } else if (exact_side(vs, k, q, sc.cx) >= 0) {
      return;      // on the facet's hyperplane or on the simplex's side of it: not a candidate
    }
    std::vector<Contender>& cont = sc.cont[k];
    cont.push_back(Contender{q, lo, hi});
    if (hi < sc.ustar[k]) sc.ustar[k] = hi;
    if (cont.size() > 64) {   // drop what the bound has overtaken
      size_t m = 0;
      for (size_t c = 0; c < cont.size(); ++c)
        if (cont[c].lo <= sc.ustar[k]) cont[m++] = cont[c];
      cont.resize(m);
    }
  }

  // apex[k] for the open facets of one simplex: the point id, -1 for a facet of the hull (nothing beyond), -2 on error
  void pivots(const int32_t* vs, unsigned open_mask, int* apex, Scan& sc) {
    int ks[V], nk = 0;
    for (int k = 0; k < V; ++k)
      if (open_mask >> k & 1) {
        ks[nk++] = k;
        sc.cont[k].clear();
        sc.ustar[k] = std::numeric_limits<double>::infinity();
      }
    if (nk == 0) return;
    Geometry<D> G;
    geometry(vs, G);
    sc.pivots += nk;
    if (!G.usable) {   // a sliver the double inverse cannot bound: every point is a contender, decided exactly
      for (int a = 0; a < nk; ++a) {
        const int k = ks[a];
        for (int q = 0; q < (int)n; ++q) {
          bool own = false;
          for (int i = 0; i < V; ++i) own |= vs[i] == q;
          if (!own && exact_side(vs, k, q, sc.cx) < 0) sc.cont[k].push_back(Contender{q, 0.0, 0.0});
        }
      }
    } else {
      alignas(64) double numc[BLK];
      alignas(64) double lbuf[V][BLK];
      unsigned hits[V];
      // a first bound from the first group of points: without one, every candidate of the first block - half of its
      // points - would go through the scalar stage
      for (int q = 0; q < (int)std::min<int64_t>(GRP, n); ++q) {
        bool own = false;
        for (int t = 0; t < V; ++t) own |= vs[t] == q;
        if (own) continue;
        const double t = pow_of(G, q) - G.e_pow;
        for (int a = 0; a < nk; ++a) consider(vs, G, ks[a], q, t > 0.0 ? t : 0.0, lam_of(G, ks[a], q), sc);
      }
      for (int64_t base = 0; base < n; base += BLK) {
        const int cnt = (int)std::min<int64_t>(BLK, n - base);
        scan_fn(xs.data(), sq.data(), npad, base, cnt, G, ks, nk, sc.ustar, hits, numc, lbuf, vs);
        if (base == 0)
          for (int a = 0; a < nk; ++a) hits[a] &= ~1u;        // (the first group has been looked at)
        for (int a = 0; a < nk; ++a) {
          unsigned m = hits[a];
          const int k = ks[a];
          while (m) {
            const int g = __builtin_ctz(m);
            m &= m - 1;
            const int i1 = std::min(cnt, (g + 1) * GRP);
            for (int i = g * GRP; i < i1; ++i)
              if (numc[i] != std::numeric_limits<double>::infinity()) consider(vs, G, k, (int)(base + i), numc[i], lbuf[a][i], sc);
          }
        }
      }
    }
    for (int a = 0; a < nk; ++a) {
      const int k = ks[a];
      std::vector<Contender>& cont = sc.cont[k];
      size_t m = 0;
      for (size_t c = 0; c < cont.size(); ++c)
        if (cont[c].lo <= sc.ustar[k]) cont[m++] = cont[c];
      cont.resize(m);
      if (m == 0) { apex[k] = -1; continue; }
      if (m > 1) contenders_total.fetch_add((long)m, std::memory_order_relaxed);
      int best = cont[0].q;
      bool tie = false;
      for (size_t c = 1; c < m; ++c) {
        const int s = exact_compare(vs, k, best, cont[c].q, sc.cx);
        if (s > 0) { best = cont[c].q; tie = false; }
        else if (s == 0) tie = true;
      }
      if (tie) { error.store(1); apex[k] = -2; continue; }   // cospherical points: not for this routine
      apex[k] = best;
    }
  }
  // ---- first simplex: grown from a point by the smallest circumsphere (floating point), then verified exactly
  bool first_simplex(int start, int32_t* out, Scan& sc) {
    int s[V];
    s[0] = start;
    for (int j = 0; j < D; ++j) {      // s[0 .. j] chosen; pick s[j + 1]
      double best = std::numeric_limits<double>::infinity();
      int arg = -1;
      for (int q = 0; q < (int)n; ++q) {
        bool own = false;
        for (int t = 0; t <= j; ++t) own |= s[t] == q;
        if (own) continue;
        // Gram system of the edges e_1 .. e_j, e_(j+1) = q - s0:  G a = diag(G) / 2,  R^2 = a . diag(G) / 2
        const int m = j + 1;
        double E[D][D], Gm[D][D + 1];
        for (int a = 0; a < m; ++a) {
          const double* A = P(a < j ? s[a + 1] : q);
          for (int c = 0; c < D; ++c) E[a][c] = A[c] - P(s[0])[c];
        }
        for (int a = 0; a < m; ++a) {
          for (int b2 = 0; b2 < m; ++b2) {
            double t = 0.0;
            for (int c = 0; c < D; ++c) t += E[a][c] * E[b2][c];
            Gm[a][b2] = t;
          }
          Gm[a][m] = 0.5 * Gm[a][a];
        }
        double diag[D];
        for (int a = 0; a < m; ++a) diag[a] = Gm[a][a];
        bool ok = true;
        for (int c = 0; c < m && ok; ++c) {
          int piv = c;
          for (int r = c + 1; r < m; ++r)
            if (std::fabs(Gm[r][c]) > std::fabs(Gm[piv][c])) piv = r;
          if (std::fabs(Gm[piv][c]) < 1e-300) { ok = false; break; }
          if (piv != c)
            for (int t = 0; t <= m; ++t) std::swap(Gm[c][t], Gm[piv][t]);
          for (int r = c + 1; r < m; ++r) {
            const double f = Gm[r][c] / Gm[c][c];
            for (int t = c; t <= m; ++t) Gm[r][t] -= f * Gm[c][t];
          }
        }
        if (!ok) continue;
        double al[D];
        for (int r = m - 1; r >= 0; --r) {
          double t = Gm[r][m];
          for (int c = r + 1; c < m; ++c) t -= Gm[r][c] * al[c];
          al[r] = t / Gm[r][r];
        }
        double r2 = 0.0;
        for (int a = 0; a < m; ++a) r2 += 0.5 * al[a] * diag[a];
        if (std::isfinite(r2) && r2 < best) { best = r2; arg = q; }
      }
      if (arg < 0) return false;
      s[j + 1] = arg;
    }
    std::sort(s, s + V);
    for (int i = 0; i < V; ++i) out[i] = s[i];
    // exact verification: a non-degenerate simplex whose circumsphere has no point strictly inside or on it
    if (wide_sign(det_edges(out, 0, sc.cx)) == 0) return false;
    Geometry<D> G;
    geometry(out, G);
    for (int q = 0; q < (int)n; ++q) {
      bool own = false;
      for (int i = 0; i < V; ++i) own |= out[i] == q;
      if (own) continue;
      if (G.usable && pow_of(G, q) - G.e_pow > 0.0) continue;
      if (exact_power_sign(out, q, sc.cx) <= 0) return false;
    }
    return true;
  }

  // ---- level-synchronous gift wrapping
  int64_t run(Pool& pool, int32_t** out_cells) {
    std::vector<Scan> scans((size_t)pool.nt);
    int32_t first[V];
    {
      // start near the centroid; other starting points if the greedy construction meets a degenerate configuration
      double cen[D] = {0};
      for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < D; ++j) cen[j] += P((int)i)[j];
      std::vector<std::pair<double, int>> byc((size_t)n);
      for (int64_t i = 0; i < n; ++i) {
        double t = 0.0;
        for (int j = 0; j < D; ++j) {
          const double d = P((int)i)[j] - cen[j] / (double)n;
          t += d * d;
        }
        byc[(size_t)i] = {t, (int)i};
      }
      const size_t tries = std::min<size_t>(8, (size_t)n);
      std::partial_sort(byc.begin(), byc.begin() + (long)tries, byc.end());
      bool ok = false;
      for (size_t t = 0; t < tries && !ok; ++t) ok = first_simplex(byc[t].second, first, scans[0]);
      if (!ok) return E_DEGENERATE;
    }
    verts.assign(first, first + V);
    std::vector<uint16_t> open_mask(1, (uint16_t)((1u << V) - 1));
    int64_t lo = 0, hi = 1;
    const int64_t max_simplices = (int64_t)1 << 28;

    // Two lock-free open-addressing tables (one word per slot, claimed by CAS):
    //   A, per sub-round: the candidate simplices by their vertices - the same simplex is reached through several
    //      facets; the smallest candidate index represents it;
    //   T, per level: facets by their vertices - the still-open facets of the level's simplices ("old") and every facet
    //      of the simplices the level has produced so far ("new").  The second owner of a facet closes it for both.
    // (a slot holds the upper half of the key's hash beside the entry: a probe that passes over another key's slot
    // compares one word and never touches that key's vertices - a cache miss each in tables of millions of slots)
    struct Table {
      std::unique_ptr<std::atomic<uint64_t>[]> idx;   // hash tag << 32 | entry (0: empty)
      size_t cap = 0, mask = 0;
    } A, T;
    auto need_table = [&](Table& t, size_t items) {
      size_t cap = 1024;
      while (cap < 2 * items + 16) cap <<= 1;
      if (cap > t.cap) {
        t.idx.reset(new std::atomic<uint64_t>[cap]);
        t.cap = cap;
      }
      t.mask = cap - 1;
      pool.parallel_for((int64_t)cap, 1 << 16, [&](int64_t a, int64_t b, int) {
        for (int64_t i = a; i < b; ++i) t.idx[(size_t)i].store(0, std::memory_order_relaxed);
      });
    };
    std::vector<int> apex;
    std::vector<int64_t> task, cand_base;
    std::vector<int32_t> cverts;
    std::vector<int32_t> cslot;
    std::vector<uint8_t> cpos;

    int levels = 0;
    long rounds = 0;
    double t_piv = 0, t_rest = 0, t_ph[6] = {0, 0, 0, 0, 0, 0};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const char* env_r = std::getenv("FLOODER_DELAUNAY_ROUNDS");
    while (lo < hi) {
      const int64_t nf = hi - lo;
      ++levels;
      double t0 = now();
      // A level is worked off in R sub-rounds: the simplices a sub-round produces close facets of the level's other
      // simplices BEFORE those are pivoted (one sweep of the level reaches a new simplex from two sides on average:
      // twice the pivots).  R grows with the width of the level, up to 4 (8 on at most 8 threads; measured on 32 threads of the GPU box's host at
      // cfg 4: R = 1 438 ms, 4 354, 8 387, 16 443 - every sub-round is six more barriers); a narrow level keeps all
      // threads busy only in one go.
      int64_t n_open = 0;
      for (int64_t s = 0; s < nf; ++s) n_open += __builtin_popcount(open_mask[(size_t)(lo + s)]);
      if (n_open == 0) break;
      if (nf >= ((int64_t)1 << 24) || n_open >= ((int64_t)1 << 24)) return E_TOO_MANY;   // (table entries are 31-bit)
      int R = env_r ? std::atoi(env_r) : (int)std::min<int64_t>(pool.nt <= 8 ? 8 : 4, nf / ((int64_t)pool.nt * 48));
      R = std::max(1, std::min(R, 64));
      // facets closed during this level: of the level's own simplices, and of the new ones (at most one per open facet)
      std::unique_ptr<std::atomic<uint32_t>[]> fclosed(new std::atomic<uint32_t>[(size_t)nf]);
      std::unique_ptr<std::atomic<uint32_t>[]> nclosed(new std::atomic<uint32_t>[(size_t)n_open]);
      pool.parallel_for(std::max(nf, n_open), 1 << 15, [&](int64_t a, int64_t b, int) {
        for (int64_t i = a; i < b; ++i) {
          if (i < nf) fclosed[(size_t)i].store(0, std::memory_order_relaxed);
          if (i < n_open) nclosed[(size_t)i].store(0, std::memory_order_relaxed);
        }
      });
      need_table(T, (size_t)n_open * (size_t)(V + 1));
      std::atomic<int> bad{0};
      // insert facet k of a simplex (old: index into the level, new: index among the level's new simplices) into T; the
      // second owner closes the facet for both, a third means the pivots were inconsistent
      auto facet_insert = [&](const int32_t* v, int k, int32_t me) {
        uint64_t h = 0xD6E8FEB86659FD93ull;
        for (int i = 0; i < V; ++i)
          if (i != k) h = mix64(h ^ (uint64_t)(uint32_t)v[i]);
        size_t slot = (size_t)h & T.mask;
        const uint64_t tag = h & 0xFFFFFFFF00000000ull, mine = tag | (uint32_t)me;
        for (;;) {
          uint64_t word = T.idx[slot].load(std::memory_order_acquire);
          if (word == 0) {
            if (T.idx[slot].compare_exchange_strong(word, mine, std::memory_order_acq_rel)) return;
          }
          if ((word & 0xFFFFFFFF00000000ull) != tag) {
            slot = (slot + 1) & T.mask;
            continue;
          }
          const int32_t cur = (int32_t)((uint32_t)word & 0x7FFFFFFFu);
          const bool cur_new = (cur - 1) & 1;
          const int64_t s2 = (cur - 1) >> 5;
          const int k2 = ((cur - 1) >> 1) & 15;
          const int32_t* u = &verts[(size_t)((cur_new ? hi : lo) + s2) * V];
          bool same = true;
          for (int i = 0, i2 = 0; i < V && same; ++i) {
            if (i == k) continue;
            if (i2 == k2) ++i2;
            same = u[i2] == v[i];
            ++i2;
          }
          if (same) {
            // (bit 31 of the entry: the facet has its second owner - a third means the pivots were inconsistent)
            if (T.idx[slot].fetch_or(0x80000000ull, std::memory_order_relaxed) & 0x80000000ull) bad.store(1);
            const bool me_new = (me - 1) & 1;
            const int64_t s1 = (me - 1) >> 5;
            (me_new ? nclosed : fclosed)[(size_t)s1].fetch_or(1u << k, std::memory_order_relaxed);
            (cur_new ? nclosed : fclosed)[(size_t)s2].fetch_or(1u << k2, std::memory_order_relaxed);
            return;
          }
          slot = (slot + 1) & T.mask;
        }
      };
      auto entry = [](int64_t s, int k, bool is_new) { return (int32_t)(((s << 4 | k) << 1 | (is_new ? 1 : 0)) + 1); };
      pool.parallel_for(nf, 1024, [&](int64_t a, int64_t b, int) {
        for (int64_t s = a; s < b; ++s) {
          const unsigned m = open_mask[(size_t)(lo + s)];
          for (int k = 0; k < V; ++k)
            if (m >> k & 1) facet_insert(&verts[(size_t)(lo + s) * V], k, entry(s, k, false));
        }
      });
      if (bad.load()) return E_INCONSISTENT;
      t_rest += now() - t0;
      t_ph[0] += now() - t0;
      int64_t nn_level = 0;
      for (int r = 0; r < R; ++r) {
        t0 = now();
        ++rounds;
        // (a) pivots across the facets of this sub-round's simplices that are still open
        task.clear();
        for (int64_t s = r; s < nf; s += R)
          if (open_mask[(size_t)(lo + s)] & ~fclosed[(size_t)s].load(std::memory_order_relaxed)) task.push_back(s);
        const int64_t nt_ = (int64_t)task.size();
        if (nt_ == 0) { t_rest += now() - t0; continue; }
        apex.assign((size_t)nt_ * V, -3);
        pool.parallel_for(nt_, std::max<int64_t>(1, std::min<int64_t>(16, nt_ / (8 * (int64_t)pool.nt))), [&](int64_t a, int64_t b, int tid) {
          for (int64_t i = a; i < b; ++i) {
            if (error.load(std::memory_order_relaxed)) return;
            const int64_t s = task[(size_t)i];
            const unsigned m = open_mask[(size_t)(lo + s)] & ~fclosed[(size_t)s].load(std::memory_order_relaxed);
            pivots(&verts[(size_t)(lo + s) * V], m, &apex[(size_t)i * V], scans[(size_t)tid]);
          }
        });
        if (error.load()) return E_DEGENERATE;
        const double t1 = now();
        t_piv += t1 - t0;
        // (b) candidates: the simplex beyond every pivoted facet
        cand_base.assign((size_t)nt_ + 1, 0);
        for (int64_t i = 0; i < nt_; ++i) {
          int c = 0;
          for (int k = 0; k < V; ++k) c += apex[(size_t)i * V + k] >= 0;
          cand_base[(size_t)i + 1] = cand_base[(size_t)i] + c;
        }
        const int64_t nc = cand_base[(size_t)nt_];
        if (nc == 0) { t_rest += now() - t1; continue; }
        if (hi + nn_level + nc > max_simplices) return E_TOO_MANY;
        cverts.resize((size_t)nc * V);
        cslot.resize((size_t)nc);
        cpos.resize((size_t)nc);
        pool.parallel_for(nt_, 256, [&](int64_t a, int64_t b, int) {
          for (int64_t i = a; i < b; ++i) {
            int64_t c = cand_base[(size_t)i];
            const int32_t* vs = &verts[(size_t)(lo + task[(size_t)i]) * V];
            for (int k = 0; k < V; ++k) {
              const int q = apex[(size_t)i * V + k];
              if (q < 0) continue;
              int32_t* o = &cverts[(size_t)c * V];
              int m = 0, pos = -1;
              for (int j = 0; j < V; ++j) {
                if (j == k) continue;
                if (pos < 0 && q < vs[j]) { pos = m; o[m++] = q; }
                o[m++] = vs[j];
              }
              if (pos < 0) { pos = m; o[m++] = q; }
              cpos[(size_t)c] = (uint8_t)pos;     // the facet opposite the apex is the parent's: closed
              ++c;
            }
          }
        });
        double tp = now();
        t_ph[1] += tp - t1;
        // (c) the same simplex reached through several facets: the smallest candidate index represents it
        need_table(A, (size_t)nc);
        pool.parallel_for(nc, 1024, [&](int64_t a, int64_t b, int) {
          for (int64_t c = a; c < b; ++c) {
            const int32_t* v = &cverts[(size_t)c * V];
            uint64_t h = 0x9E3779B97F4A7C15ull;
            for (int i = 0; i < V; ++i) h = mix64(h ^ (uint64_t)(uint32_t)v[i]);
            size_t slot = (size_t)h & A.mask;
            const uint64_t tag = h & 0xFFFFFFFF00000000ull, mine = tag | (uint32_t)((int32_t)c + 1);
            for (;;) {
              uint64_t word = A.idx[slot].load(std::memory_order_acquire);
              if (word == 0) {
                if (A.idx[slot].compare_exchange_strong(word, mine, std::memory_order_acq_rel)) break;
              }
              bool same = (word & 0xFFFFFFFF00000000ull) == tag;
              if (same) {
                const int32_t* u = &cverts[(size_t)((int32_t)(uint32_t)word - 1) * V];
                for (int i = 0; i < V; ++i) same &= u[i] == v[i];
              }
              if (same) {   // (the smallest candidate index stays in the slot)
                while ((int32_t)(uint32_t)word - 1 > c && !A.idx[slot].compare_exchange_weak(word, mine, std::memory_order_acq_rel)) {
                }
                break;
              }
              slot = (slot + 1) & A.mask;
            }
            cslot[(size_t)c] = (int32_t)slot;
          }
        });
        t_ph[2] += now() - tp;
        tp = now();
        // (d) the new simplices, in candidate order, behind the ones the level has produced so far
        std::vector<int64_t> new_of((size_t)nc + 1, 0);
        for (int64_t c = 0; c < nc; ++c)
          new_of[(size_t)c + 1] = new_of[(size_t)c] + ((int32_t)(uint32_t)A.idx[(size_t)cslot[(size_t)c]].load(std::memory_order_relaxed) == (int32_t)c + 1);
        const int64_t nn = new_of[(size_t)nc];
        if (nn_level + nn > n_open) return E_INCONSISTENT;     // (more simplices than open facets: impossible)
        verts.resize((size_t)(hi + nn_level + nn) * V);
        pool.parallel_for(nc, 4096, [&](int64_t a, int64_t b, int) {
          for (int64_t c = a; c < b; ++c) {
            if ((int32_t)(uint32_t)A.idx[(size_t)cslot[(size_t)c]].load(std::memory_order_relaxed) != (int32_t)c + 1) continue;
            std::memcpy(&verts[(size_t)(hi + nn_level + new_of[(size_t)c]) * V], &cverts[(size_t)c * V], sizeof(int32_t) * V);
            nclosed[(size_t)(nn_level + new_of[(size_t)c])].store(1u << cpos[(size_t)c], std::memory_order_relaxed);
          }
        });
        t_ph[3] += now() - tp;
        tp = now();
        // (e) the facets of the new simplices into the level's table (not the one a simplex was reached through: its
        // parent's, pivoted and done): they meet still-open facets of the level - no pivot needed there any more -,
        // facets of other new simplices, or nobody yet
        pool.parallel_for(nn, 512, [&](int64_t a, int64_t b, int) {
          for (int64_t s = a; s < b; ++s) {
            const int64_t id = nn_level + s;
            const uint32_t origin = nclosed[(size_t)id].load(std::memory_order_relaxed) & ((1u << V) - 1);
            for (int k = 0; k < V; ++k)
              if (!((origin >> k) & 1)) facet_insert(&verts[(size_t)(hi + id) * V], k, entry(id, k, true));
          }
        });
        if (bad.load()) return E_INCONSISTENT;
        nn_level += nn;
        t_ph[4] += now() - tp;
        t_rest += now() - t1;
      }
      open_mask.resize((size_t)(hi + nn_level));
      for (int64_t s = 0; s < nn_level; ++s)
        open_mask[(size_t)(hi + s)] = (uint16_t)(((1u << V) - 1) & ~nclosed[(size_t)s].load(std::memory_order_relaxed));
      lo = hi;
      hi += nn_level;
    }
    if (std::getenv("FLOODER_DELAUNAY_VERBOSE"))
      std::fprintf(stderr, "  tables: level set-up %.3f, candidates %.3f, dedupe %.3f, new simplices %.3f, facets %.3f s\n", t_ph[0], t_ph[1], t_ph[2], t_ph[3], t_ph[4]);
    if (std::getenv("FLOODER_DELAUNAY_VERBOSE"))
      std::fprintf(stderr, "delaunay_nd<%d>: %ld simplices, %d levels (%ld sub-rounds), pivots %.3f s, tables %.3f s, %d threads; %ld pivots, %ld flagged points\n", D,
                   (long)hi, levels, rounds, t_piv, t_rest, pool.nt, [&] { long f = 0; for (auto& s : scans) f += s.pivots; return f; }(), [&] { long f = 0; for (auto& s : scans) f += s.flagged; return f; }());
    const int64_t total = hi;
    int32_t* out = (int32_t*)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(total, 1) * V);
    if (!out) return E_TOO_MANY;
    std::memcpy(out, verts.data(), sizeof(int32_t) * (size_t)total * V);
    sort_rows_parallel(pool, out, total, V, n);          // lexicographic order: the order the face tables are built in
    *out_cells = out;
    return total;
  }
};

long g_last_exact = 0, g_last_contenders = 0;
int g_last_threads = 0, g_force_isa = -1;

template <int D>
int64_t delaunay_nd(const double* pts, int64_t n, int n_threads, int32_t** out_cells) {
  Engine<D> e;
  e.n = n;
  e.p = pts;
  int emin, emax;
  if (!dyadic_range(pts, (int64_t)D * n, emin, emax)) return E_RANGE;
  if (emin > emax) return E_FLAT;
  if (emax - emin > 120) return E_RANGE;         // the grid would not fit 121-bit integers
  e.qi.resize((size_t)n * D);
  for (int64_t i = 0; i < (int64_t)D * n; ++i) e.qi[(size_t)i] = (__int128)std::ldexp(pts[i], -emin);
  // duplicates: two equal rows can never be separated by a sphere
  {
    std::vector<int> idx((size_t)n);
    for (int64_t i = 0; i < n; ++i) idx[(size_t)i] = (int)i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) {
      return std::lexicographical_compare(pts + (size_t)a * D, pts + (size_t)a * D + D, pts + (size_t)b * D, pts + (size_t)b * D + D);
    });
    for (int64_t i = 1; i < n; ++i)
      if (std::equal(pts + (size_t)idx[(size_t)i] * D, pts + (size_t)idx[(size_t)i] * D + D, pts + (size_t)idx[(size_t)i - 1] * D)) return E_DUP;
  }
  // floating-point copies relative to the centre of the box (the forms are evaluated on absolute coordinates: their
  // roundings scale with the largest coordinate, which the centring keeps at half the extent; a rounded difference
  // is off by at most eps relative, which the constants of the bounds include)
  e.npad = (n + BLK - 1) / BLK * BLK;
  e.pc.resize((size_t)n * D);
  e.xs.assign((size_t)e.npad * D, 0.0);
  e.sq.assign((size_t)e.npad, 0.0);
  for (int j = 0; j < D; ++j) {
    double lo = pts[j], hi = pts[j];
    for (int64_t i = 0; i < n; ++i) {
      lo = std::min(lo, pts[(size_t)i * D + j]);
      hi = std::max(hi, pts[(size_t)i * D + j]);
    }
    if (!(hi > lo)) return E_FLAT;
    const double mid = 0.5 * (lo + hi);
    double am = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      const double v = pts[(size_t)i * D + j] - mid;
      e.pc[(size_t)i * D + j] = v;
      e.xs[(size_t)j * e.npad + (size_t)i] = v;
      am = std::max(am, std::fabs(v));
    }
    e.ext[j] = (hi - lo) * SLACK;
    e.amax[j] = am * SLACK;
  }
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int j = 0; j < D; ++j) s += e.pc[(size_t)i * D + j] * e.pc[(size_t)i * D + j];
    e.sq[(size_t)i] = s;
  }
  const int isa = g_force_isa >= 0 ? g_force_isa
                  : (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl")) ? 2
                  : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
  e.scan_fn = isa == 2 ? Engine<D>::scan_block_avx512 : isa == 1 ? Engine<D>::scan_block_avx2 : Engine<D>::scan_block_generic;
  n_threads = host_threads(n_threads);
  Pool pool(n_threads);
  g_last_threads = n_threads;
  const int64_t rc = e.run(pool, out_cells);
  g_last_exact = e.exact_calls.load();
  g_last_contenders = e.contenders_total.load();
  return rc;
}

}  // namespace

extern "C" int64_t flooder_delaunay_nd(const double* pts, int64_t n, int dim, int n_threads, int32_t** out_cells) {
  if (!pts || !out_cells) return E_FEW;
  *out_cells = nullptr;
  if (dim < 2 || dim > 8) return E_DIM;
  if (n < dim + 2 || n > 0x3fffffff) return E_FEW;
  try {
  switch (dim) {
    case 2: return delaunay_nd<2>(pts, n, n_threads, out_cells);
    case 3: return delaunay_nd<3>(pts, n, n_threads, out_cells);
    case 4: return delaunay_nd<4>(pts, n, n_threads, out_cells);
    case 5: return delaunay_nd<5>(pts, n, n_threads, out_cells);
    case 6: return delaunay_nd<6>(pts, n, n_threads, out_cells);
    case 7: return delaunay_nd<7>(pts, n, n_threads, out_cells);
    default: return delaunay_nd<8>(pts, n, n_threads, out_cells);
  }
  } catch (const std::exception&) {   // (out of memory: a complex too large for this host - nothing may cross the C ABI)
    return E_TOO_MANY;
  }
}

extern "C" void flooder_host_free(void* p) { std::free(p); }

// diagnostics of the last call: 0 exact predicate evaluations, 1 contenders that reached the exact stage, 2 threads
extern "C" long flooder_delaunay_nd_stat(int what) {
  return what == 0 ? g_last_exact : what == 1 ? g_last_contenders : what == 2 ? (long)g_last_threads : 0;
}

// test hook: force the scan's instruction set (0 generic, 1 AVX2, 2 AVX-512; -1 detect).  Returns the previous value.
extern "C" int flooder_delaunay_nd_isa(int isa) {
  const int old = g_force_isa;
  if (isa >= -1 && isa <= 2) g_force_isa = isa;
  return old;
}
