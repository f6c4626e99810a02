// delaunay2d.cpp - Delaunay triangulation of the landmarks in the plane (host C++, no GPU).
//
// The two-dimensional sibling of delaunay3d.cpp; replaces the same reference call (gudhi.DelaunayComplex(landmarks),
// flooder/core.py:130-138, for 2-D clouds such as the reference's annulus and figure-eight generators) and the Qhull
// call this build made through scipy.
//
// Incremental Bowyer-Watson insertion with ghost triangles (an infinite vertex closes the hull), points taken along a
// Morton curve and located by a visibility walk.  orient2d and incircle are evaluated in double precision behind a
// static bound from the bounding box, then Shewchuk's bound from the operands, and where neither decides EXACTLY: the
// coordinates are scaled once to integers on a common dyadic grid (< 58 bits, checked: else the caller uses Qhull),
// differences fit 64 bits, products and 2 x 2 minors 128, and the incircle sum runs in 512-bit integers.  Points in
// general position have one Delaunay triangulation - the same triangles as Qhull's and CGAL's (tests/test_delaunay.py);
// cocircular points give one of the valid ones.
//
// The new triangles of an insertion fan around the new vertex along the boundary of the cavity, a closed chain of
// directed edges: the triangle on edge x -> y meets the one whose edge STARTS at y and the one whose edge ENDS at x,
// so two stamped per-vertex slots link the fan - no search, no table of edges.
//
// C ABI:  int64_t flooder_delaunay2d(pts float64 (n, 2) row-major, n, tris int32 (cap, 3), cap)
//   returns the number of triangles written (vertex ids counter-clockwise), -needed when cap is too small, or a code
//   < -(1 << 40) when the input is not one this routine takes (duplicates, all points collinear, coordinates that do
//   not scale to 58-bit integers, an inconsistent cavity): the caller uses Qhull then.
#include "exact_int.hpp"

namespace {

struct Mesh2 {
  int64_t n = 0;
  const double* p = nullptr;  // (n, 2) doubles
  std::vector<int64_t> q;     // the same points as integers on a common dyadic grid
  struct Tri {
    int v[3];   // counter-clockwise (with the infinite vertex in any position of a ghost)
    int nb[3];  // nb[i]: the triangle across the edge opposite v[i], i.e. across (v[i+1], v[i+2])
  };
  std::vector<Tri> t;
  std::vector<int> free_slots, mark;
  int INF = 0;
  long exact_calls = 0;
  double orient_static = 0.0, incircle_static = 0.0;  // error bounds from the box (exact_int.hpp / delaunay3d.cpp)

  // ---- sign of det [a - c; b - c]: positive when a, b, c turn counter-clockwise
  int orient_exact(int a, int b, int c) {
    ++exact_calls;
    const int64_t* A = &q[2 * (size_t)a]; const int64_t* B = &q[2 * (size_t)b]; const int64_t* C = &q[2 * (size_t)c];
    const __int128 l = (__int128)(A[0] - C[0]) * (B[1] - C[1]), r = (__int128)(A[1] - C[1]) * (B[0] - C[0]);
    return l > r ? 1 : (l < r ? -1 : 0);
  }
  int orient(int a, int b, int c) {
    const double* A = p + 2 * (size_t)a; const double* B = p + 2 * (size_t)b; const double* C = p + 2 * (size_t)c;
    const double l = (A[0] - C[0]) * (B[1] - C[1]), r = (A[1] - C[1]) * (B[0] - C[0]);
    const double det = l - r;
    if (det > orient_static) return 1;
    if (-det > orient_static) return -1;
    const double err = 3.3306690738754716e-16 * (std::fabs(l) + std::fabs(r));  // (3 + 16 eps) eps, eps = 2^-53
    if (det > err) return 1;
    if (-det > err) return -1;
    return orient_exact(a, b, c);
  }
  // ---- sign of the lifted 3 x 3 determinant: positive when d lies inside the circle through a, b, c (counter-clockwise)
  int incircle_exact(int a, int b, int c, int d) {
    ++exact_calls;
    const int64_t* D = &q[2 * (size_t)d];
    __int128 x[3], y[3], w[3];
    const int id[3] = {a, b, c};
    for (int i = 0; i < 3; ++i) {
      const int64_t* P = &q[2 * (size_t)id[i]];
      x[i] = (__int128)(P[0] - D[0]);
      y[i] = (__int128)(P[1] - D[1]);
      w[i] = x[i] * x[i] + y[i] * y[i];
    }
    Big r = big_mul(big_from(w[0]), big_from(x[1] * y[2] - x[2] * y[1]));
    r = big_add(r, big_mul(big_from(w[1]), big_from(x[2] * y[0] - x[0] * y[2])));
    r = big_add(r, big_mul(big_from(w[2]), big_from(x[0] * y[1] - x[1] * y[0])));
    return big_sign(r);
  }
  int incircle(int a, int b, int c, int d) {
    const double* A = p + 2 * (size_t)a; const double* B = p + 2 * (size_t)b; const double* C = p + 2 * (size_t)c;
    const double* D = p + 2 * (size_t)d;
    const double adx = A[0] - D[0], ady = A[1] - D[1], bdx = B[0] - D[0], bdy = B[1] - D[1], cdx = C[0] - D[0],
                 cdy = C[1] - D[1];
    const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy, cdxady = cdx * ady, adxcdy = adx * cdy, adxbdy = adx * bdy,
                 bdxady = bdx * ady;
    const double alift = adx * adx + ady * ady, blift = bdx * bdx + bdy * bdy, clift = cdx * cdx + cdy * cdy;
    const double det = alift * (bdxcdy - cdxbdy) + blift * (cdxady - adxcdy) + clift * (adxbdy - bdxady);
    if (det > incircle_static) return 1;
    if (-det > incircle_static) return -1;
    const double perm = (std::fabs(bdxcdy) + std::fabs(cdxbdy)) * alift + (std::fabs(cdxady) + std::fabs(adxcdy)) * blift +
                        (std::fabs(adxbdy) + std::fabs(bdxady)) * clift;
    const double err = 1.1102230246251577e-15 * perm;  // (10 + 96 eps) eps
    if (det > err) return 1;
    if (-det > err) return -1;
    return incircle_exact(a, b, c, d);
  }

  // is point e inside the open circumcircle of triangle ti (a ghost: strictly beyond its hull edge, or on the edge's
  // line and in conflict with the finite triangle behind it - i.e. strictly between the edge's end points)?
  bool conflict(int ti, int e) {
    const Tri& T = t[(size_t)ti];
    for (int i = 0; i < 3; ++i) {
      if (T.v[i] == INF) {
        const int o = orient(T.v[(i + 1) % 3], T.v[(i + 2) % 3], e);
        if (o != 0) return o > 0;
        const Tri& U = t[(size_t)T.nb[i]];
        return incircle(U.v[0], U.v[1], U.v[2], e) > 0;
      }
    }
    return incircle(T.v[0], T.v[1], T.v[2], e) > 0;
  }
  bool is_ghost(int ti) const {
    const Tri& T = t[(size_t)ti];
    return T.v[0] == INF || T.v[1] == INF || T.v[2] == INF;
  }
  int new_tri() {
    if (!free_slots.empty()) {
      const int i = free_slots.back();
      free_slots.pop_back();
      return i;
    }
    t.push_back(Tri{});
    mark.push_back(-1);
    return (int)t.size() - 1;
  }
};

uint64_t morton2(uint32_t x, uint32_t y) {
  auto spread = [](uint64_t v) {
    v &= 0xffffffffull;
    v = (v | v << 16) & 0x0000ffff0000ffffull;
    v = (v | v << 8) & 0x00ff00ff00ff00ffull;
    v = (v | v << 4) & 0x0f0f0f0f0f0f0f0full;
    v = (v | v << 2) & 0x3333333333333333ull;
    v = (v | v << 1) & 0x5555555555555555ull;
    return v;
  };
  return spread(x) | spread(y) << 1;
}

}  // namespace

extern "C" int64_t flooder_delaunay2d(const double* pts, int64_t n, int32_t* tris, int64_t cap) {
  if (!pts || n < 3 || n > 0x3fffffff) return E_FEW;
  Mesh2 m;
  m.n = n;
  m.p = pts;
  m.INF = (int)n;
  int emin, emax;
  if (!dyadic_range(pts, 2 * n, emin, emax)) return E_RANGE;
  if (emin > emax) return E_FLAT;            // (all coordinates zero)
  if (emax - emin > 57) return E_RANGE;      // would not fit 58-bit integers: not for this routine
  m.q.resize(2 * (size_t)n);
  for (int64_t i = 0; i < 2 * n; ++i) m.q[(size_t)i] = (int64_t)std::ldexp(pts[i], -emin);

  double lo[2] = {pts[0], pts[1]}, hi[2] = {pts[0], pts[1]};
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < 2; ++k) {
      lo[k] = std::min(lo[k], pts[2 * i + k]);
      hi[k] = std::max(hi[k], pts[2 * i + k]);
    }
  {
    // every coordinate difference is at most D = the largest extent of the box: |l| + |r| <= 2 D^2 in orient, the
    // permanent of incircle at most 3 * (2 D^2) * (2 D^2) (the factors 1 + 1e-6: roundings of the bound itself)
    double D = std::max(hi[0] - lo[0], hi[1] - lo[1]) * 1.000001;
    m.orient_static = 3.3306690738754716e-16 * 2.0 * D * D * 1.000001;
    m.incircle_static = 1.1102230246251577e-15 * 12.0 * D * D * D * D * 1.000001;
  }
  // ---- insertion order: Morton curve over the bounding box
  std::vector<std::pair<uint64_t, int>> order((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    uint32_t c[2];
    for (int k = 0; k < 2; ++k) {
      const double e = hi[k] - lo[k];
      const double u = e > 0 ? (pts[2 * i + k] - lo[k]) / e : 0.0;
      c[k] = (uint32_t)std::min(4294967295.0, u * 4294967296.0);
    }
    order[(size_t)i] = {morton2(c[0], c[1]), (int)i};
  }
  std::sort(order.begin(), order.end());

  // ---- first triangle: three points of the order that are not collinear
  int s[3] = {order[0].second, -1, -1};
  {
    auto same = [&](int a, int b) { return pts[2 * a] == pts[2 * b] && pts[2 * a + 1] == pts[2 * b + 1]; };
    size_t i1 = 1;
    while (i1 < (size_t)n && same(s[0], order[i1].second)) ++i1;
    if (i1 == (size_t)n) return E_DUP;
    s[1] = order[i1].second;
    for (size_t i2 = 1; i2 < (size_t)n; ++i2) {
      const int c2 = order[i2].second;
      if (c2 != s[1] && m.orient(s[0], s[1], c2) != 0) {
        s[2] = c2;
        break;
      }
    }
    if (s[2] < 0) return E_FLAT;
    if (m.orient(s[0], s[1], s[2]) < 0) std::swap(s[1], s[2]);
  }
  m.t.reserve((size_t)n * 2 + 16);
  m.mark.reserve((size_t)n * 2 + 16);
  {
    const int t0 = m.new_tri();
    for (int i = 0; i < 3; ++i) m.t[(size_t)t0].v[i] = s[i];
    int g[3];
    for (int i = 0; i < 3; ++i) g[i] = m.new_tri();
    for (int i = 0; i < 3; ++i) {  // ghost on the edge opposite s[i]: the edge reversed, then the infinite vertex
      Mesh2::Tri& G = m.t[(size_t)g[i]];
      G.v[0] = s[(i + 2) % 3];
      G.v[1] = s[(i + 1) % 3];
      G.v[2] = m.INF;
      G.nb[2] = t0;
      G.nb[0] = g[(i + 2) % 3];   // across (s[i+1], INF): the ghost of the edge that ends in s[i+1]
      G.nb[1] = g[(i + 1) % 3];   // across (INF, s[i+2])
      m.t[(size_t)t0].nb[i] = g[i];
    }
  }
  int last = 0;  // a finite triangle to start the walk from
  std::vector<int> cavity, stack, fresh, bedges;
  std::vector<uint32_t> vstamp((size_t)n + 1, 0u);
  std::vector<int> starts((size_t)n + 1, -1), ends((size_t)n + 1, -1);  // new triangle whose boundary edge starts / ends here
  uint32_t stamp = 0;
  std::vector<char> used((size_t)n, 0);
  for (int i = 0; i < 3; ++i) used[(size_t)s[i]] = 1;

  for (size_t oi = 0; oi < (size_t)n; ++oi) {
    const int pi = order[oi].second;
    if (used[(size_t)pi]) continue;
    // ---- locate: visibility walk over the finite triangles
    int cur = last;
    for (long steps = 0;; ++steps) {
      if (steps > 4 * (long)m.t.size() + 64) return E_LOCATE;
      if (m.is_ghost(cur)) break;
      const Mesh2::Tri& T = m.t[(size_t)cur];
      int go = -1;
      for (int k = 0; k < 3; ++k) {
        const int i = (int)((steps + k) % 3);  // (a different first edge every step)
        if (m.orient(T.v[(i + 1) % 3], T.v[(i + 2) % 3], pi) < 0) {
          go = i;
          break;
        }
      }
      if (go < 0) break;
      cur = T.nb[go];
    }
    if (!m.conflict(cur, pi)) {
      // (a point on the line of a hull edge, or a copy of a vertex: look around before giving up)
      int hit = -1;
      stack.assign(1, cur);
      std::vector<int> seen(1, cur);
      for (size_t h = 0; h < stack.size() && h < 256 && hit < 0; ++h) {
        for (int i = 0; i < 3 && hit < 0; ++i) {
          const int u = m.t[(size_t)stack[h]].nb[i];
          if (std::find(seen.begin(), seen.end(), u) != seen.end()) continue;
          seen.push_back(u);
          if (m.conflict(u, pi)) hit = u;
          else stack.push_back(u);
        }
      }
      if (hit < 0) return E_DUP;
      cur = hit;
    }
    // ---- cavity: the connected set of triangles in conflict with the point
    cavity.clear();
    stack.assign(1, cur);
    const int in_cav = 2 * pi, not_cav = 2 * pi + 1;   // (marks: in the cavity / tested and not in conflict)
    m.mark[(size_t)cur] = in_cav;
    while (!stack.empty()) {
      const int c = stack.back();
      stack.pop_back();
      cavity.push_back(c);
      for (int i = 0; i < 3; ++i) {
        const int u = m.t[(size_t)c].nb[i];
        if (m.mark[(size_t)u] == in_cav || m.mark[(size_t)u] == not_cav) continue;
        if (m.conflict(u, pi)) {
          m.mark[(size_t)u] = in_cav;
          stack.push_back(u);
        } else {
          m.mark[(size_t)u] = not_cav;
        }
      }
    }
    // ---- a new triangle on every boundary edge of the cavity
    if (++stamp == 0) { std::fill(vstamp.begin(), vstamp.end(), 0u); stamp = 1; }
    bedges.clear();
    for (const int c : cavity)
      for (int i = 0; i < 3; ++i)
        if (m.mark[(size_t)m.t[(size_t)c].nb[i]] != in_cav) bedges.push_back(3 * c + i);
    fresh.clear();
    for (const int ce : bedges) {
      const int c = ce / 3, i = ce % 3;
      const int u = m.t[(size_t)c].nb[i];
      const int x = m.t[(size_t)c].v[(i + 1) % 3], y = m.t[(size_t)c].v[(i + 2) % 3];
      const int nt = m.new_tri();   // (may move m.t: no references held across it)
      m.mark[(size_t)nt] = -1;
      Mesh2::Tri& N = m.t[(size_t)nt];
      N.v[0] = x; N.v[1] = y; N.v[2] = pi;
      N.nb[0] = N.nb[1] = -1;
      N.nb[2] = u;
      Mesh2::Tri& U = m.t[(size_t)u];
      int back = -1;
      for (int j = 0; j < 3; ++j)
        if (U.nb[j] == c && U.v[(j + 1) % 3] == y && U.v[(j + 2) % 3] == x) back = j;
      if (back < 0) return E_CAVITY;
      U.nb[back] = nt;
      // the boundary is one closed chain: every vertex starts exactly one edge and ends exactly one
      if (vstamp[(size_t)x] != stamp) { vstamp[(size_t)x] = stamp; starts[(size_t)x] = -1; ends[(size_t)x] = -1; }
      if (vstamp[(size_t)y] != stamp) { vstamp[(size_t)y] = stamp; starts[(size_t)y] = -1; ends[(size_t)y] = -1; }
      if (starts[(size_t)x] >= 0 || ends[(size_t)y] >= 0) return E_CAVITY;
      starts[(size_t)x] = nt;
      ends[(size_t)y] = nt;
      fresh.push_back(nt);
    }
    for (const int f : fresh) {
      Mesh2::Tri& N = m.t[(size_t)f];
      const int a = starts[(size_t)N.v[1]], b = ends[(size_t)N.v[0]];
      if (a < 0 || b < 0) return E_CAVITY;
      N.nb[0] = a;   // across (y, p): the triangle whose edge starts at y
      N.nb[1] = b;   // across (p, x): the triangle whose edge ends at x
    }
    for (const int c : cavity) {
      m.t[(size_t)c].v[0] = -2;   // dead
      m.mark[(size_t)c] = -1;
      m.free_slots.push_back(c);
    }
    for (const int f : fresh)
      if (!m.is_ghost(f)) { last = f; break; }
    used[(size_t)pi] = 1;
  }
  int64_t count = 0;
  for (size_t i = 0; i < m.t.size(); ++i)
    if (m.t[i].v[0] != -2 && !m.is_ghost((int)i)) ++count;
  if (count > cap || !tris) return -count;
  int64_t o = 0;
  for (size_t i = 0; i < m.t.size(); ++i)
    if (m.t[i].v[0] != -2 && !m.is_ghost((int)i)) {
      for (int k = 0; k < 3; ++k) tris[3 * o + k] = m.t[i].v[k];
      ++o;
    }
  return count;
}
