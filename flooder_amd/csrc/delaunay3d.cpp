// delaunay3d.cpp - Delaunay triangulation of the landmarks in three dimensions (host C++, no GPU).
//
// Replaces what the reference obtains from gudhi.DelaunayComplex (CGAL; reference call site flooder/core.py:130-138:
// `gudhi.DelaunayComplex(landmarks).create_simplex_tree()`, whose top cells are the tetrahedra of the Delaunay
// triangulation) and what this build used to take from Qhull through scipy - 6 ms for 1000 landmarks, half of a
// flood_complex call whose device part is 1.2 ms.
//
// Incremental Bowyer-Watson insertion with ghost tetrahedra (an infinite vertex closes the hull, so a point outside
// the hull is inserted like any other), points taken along a Morton curve and located by a visibility walk.  The two
// predicates - orient3d and insphere - are evaluated in double precision behind Shewchuk's static error bounds and,
// where the bound does not decide, EXACTLY: the coordinates are dyadic rationals, scaled once to integers (float32
// landmarks in a bounded box need < 58 bits: checked, else the caller falls back to Qhull), differences fit 64 bits,
// 2 x 2 minors 128, and the final sums run in 512-bit integers.  For points in general position the Delaunay
// triangulation is unique: the result equals Qhull's and CGAL's as a set of tetrahedra (tests/test_delaunay.py);
// cospherical points give one of the valid triangulations.
//
// C ABI:  int64_t flooder_delaunay3d(pts float64 (n, 3) row-major, n, tets int32 (cap, 4), cap)
//   returns the number of tetrahedra written (vertex ids in no particular order), -needed when cap is too small,
//   or a FLOODER_DELAUNAY_* code < -(1 << 40) when the input is not one this routine takes (duplicates, all points
//   coplanar, coordinates that do not scale to 58-bit integers, an inconsistent cavity): the caller uses Qhull then.
#include "exact_int.hpp"

namespace {

struct Mesh {
  int64_t n = 0;
  const double* p = nullptr;       // (n, 3) doubles
  std::vector<int64_t> q;          // the same points as integers on a common dyadic grid
  struct Tet {
    int v[4];
    int nb[4];  // nb[i]: the tetrahedron across the face opposite v[i]
  };
  std::vector<Tet> t;
  std::vector<int> free_slots, mark;
  int INF = 0;
  long exact_calls = 0;
  // static filters (first stage of both predicates): with every coordinate difference bounded by D = the largest
  // extent of the bounding box, the permanent of Shewchuk's bound is at most 6 D^3 (orient) / 72 D^5 (insphere); a
  // determinant beyond that error needs neither the permanent nor the exact stage
  double orient_static = 0.0, insphere_static = 0.0;

  // face opposite vertex i, oriented so that (face, v[i]) is an even permutation of (v0, v1, v2, v3)
  static constexpr int FACE[4][3] = {{1, 3, 2}, {0, 2, 3}, {0, 3, 1}, {0, 1, 2}};

  // ---- predicates: sign of det [a - d; b - d; c - d]  /  of the lifted 4 x 4 determinant (Shewchuk's conventions)
  int orient_exact(int a, int b, int c, int d) {
    ++exact_calls;
    const int64_t* A = &q[3 * (size_t)a]; const int64_t* B = &q[3 * (size_t)b];
    const int64_t* C = &q[3 * (size_t)c]; const int64_t* D = &q[3 * (size_t)d];
    const __int128 ax = A[0] - D[0], ay = A[1] - D[1], az = A[2] - D[2];
    const __int128 bx = B[0] - D[0], by = B[1] - D[1], bz = B[2] - D[2];
    const __int128 cx = C[0] - D[0], cy = C[1] - D[1], cz = C[2] - D[2];
    Big r = big_mul(big_from(ax), big_from(by * cz - bz * cy));
    r = big_sub(r, big_mul(big_from(ay), big_from(bx * cz - bz * cx)));
    r = big_add(r, big_mul(big_from(az), big_from(bx * cy - by * cx)));
    return big_sign(r);
  }
  int orient(int a, int b, int c, int d) {
    const double* A = p + 3 * (size_t)a; const double* B = p + 3 * (size_t)b;
    const double* C = p + 3 * (size_t)c; const double* D = p + 3 * (size_t)d;
    const double adx = A[0] - D[0], ady = A[1] - D[1], adz = A[2] - D[2];
    const double bdx = B[0] - D[0], bdy = B[1] - D[1], bdz = B[2] - D[2];
    const double cdx = C[0] - D[0], cdy = C[1] - D[1], cdz = C[2] - D[2];
    const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy, cdxady = cdx * ady, adxcdy = adx * cdy, adxbdy = adx * bdy,
                 bdxady = bdx * ady;
    const double det = adz * (bdxcdy - cdxbdy) + bdz * (cdxady - adxcdy) + cdz * (adxbdy - bdxady);
    if (det > orient_static) return 1;
    if (-det > orient_static) return -1;
    const double perm = (std::fabs(bdxcdy) + std::fabs(cdxbdy)) * std::fabs(adz) +
                        (std::fabs(cdxady) + std::fabs(adxcdy)) * std::fabs(bdz) +
                        (std::fabs(adxbdy) + std::fabs(bdxady)) * std::fabs(cdz);
    const double err = 7.771561172376103e-16 * perm;  // (7 + 56 eps) eps, eps = 2^-53
    if (det > err) return 1;
    if (-det > err) return -1;
    return orient_exact(a, b, c, d);
  }
  int insphere_exact(int a, int b, int c, int d, int e) {
    ++exact_calls;
    const int64_t* E = &q[3 * (size_t)e];
    __int128 x[4][3];
    Big w[4];
    const int id[4] = {a, b, c, d};
    for (int i = 0; i < 4; ++i) {
      const int64_t* P = &q[3 * (size_t)id[i]];
      for (int k = 0; k < 3; ++k) x[i][k] = (__int128)(P[k] - E[k]);
      w[i] = big_add(big_add(big_from(x[i][0] * x[i][0]), big_from(x[i][1] * x[i][1])), big_from(x[i][2] * x[i][2]));
    }
    auto det3 = [&](int i, int j, int k) {  // det of rows i, j, k of the 4 x 3 matrix x
      Big r = big_mul(big_from(x[i][0]), big_from(x[j][1] * x[k][2] - x[j][2] * x[k][1]));
      r = big_sub(r, big_mul(big_from(x[i][1]), big_from(x[j][0] * x[k][2] - x[j][2] * x[k][0])));
      r = big_add(r, big_mul(big_from(x[i][2]), big_from(x[j][0] * x[k][1] - x[j][1] * x[k][0])));
      return r;
    };
    // | a b c d |^T with columns (x, y, z, w): expansion along the w column
    Big r = big_mul(w[3], det3(0, 1, 2));
    r = big_sub(r, big_mul(w[2], det3(0, 1, 3)));
    r = big_add(r, big_mul(w[1], det3(0, 2, 3)));
    r = big_sub(r, big_mul(w[0], det3(1, 2, 3)));
    return big_sign(r);
  }
  int insphere(int a, int b, int c, int d, int e) {
    const double* A = p + 3 * (size_t)a; const double* B = p + 3 * (size_t)b; const double* C = p + 3 * (size_t)c;
    const double* D = p + 3 * (size_t)d; const double* E = p + 3 * (size_t)e;
    const double aex = A[0] - E[0], aey = A[1] - E[1], aez = A[2] - E[2];
    const double bex = B[0] - E[0], bey = B[1] - E[1], bez = B[2] - E[2];
    const double cex = C[0] - E[0], cey = C[1] - E[1], cez = C[2] - E[2];
    const double dex = D[0] - E[0], dey = D[1] - E[1], dez = D[2] - E[2];
    const double aexbey = aex * bey, bexaey = bex * aey, ab = aexbey - bexaey;
    const double bexcey = bex * cey, cexbey = cex * bey, bc = bexcey - cexbey;
    const double cexdey = cex * dey, dexcey = dex * cey, cd = cexdey - dexcey;
    const double dexaey = dex * aey, aexdey = aex * dey, da = dexaey - aexdey;
    const double aexcey = aex * cey, cexaey = cex * aey, ac = aexcey - cexaey;
    const double bexdey = bex * dey, dexbey = dex * bey, bd = bexdey - dexbey;
    const double abc = aez * bc - bez * ac + cez * ab;
    const double bcd = bez * cd - cez * bd + dez * bc;
    const double cda = cez * da + dez * ac + aez * cd;
    const double dab = dez * ab + aez * bd + bez * da;
    const double alift = aex * aex + aey * aey + aez * aez, blift = bex * bex + bey * bey + bez * bez;
    const double clift = cex * cex + cey * cey + cez * cez, dlift = dex * dex + dey * dey + dez * dez;
    const double det = (dlift * abc - clift * dab) + (blift * cda - alift * bcd);
    if (det > insphere_static) return 1;
    if (-det > insphere_static) return -1;
    const double az = std::fabs(aez), bz = std::fabs(bez), cz = std::fabs(cez), dz = std::fabs(dez);
    const double perm =
        ((std::fabs(cexdey) + std::fabs(dexcey)) * bz + (std::fabs(dexbey) + std::fabs(bexdey)) * cz + (std::fabs(bexcey) + std::fabs(cexbey)) * dz) * alift +
        ((std::fabs(dexaey) + std::fabs(aexdey)) * cz + (std::fabs(aexcey) + std::fabs(cexaey)) * dz + (std::fabs(cexdey) + std::fabs(dexcey)) * az) * blift +
        ((std::fabs(aexbey) + std::fabs(bexaey)) * dz + (std::fabs(bexdey) + std::fabs(dexbey)) * az + (std::fabs(dexaey) + std::fabs(aexdey)) * bz) * clift +
        ((std::fabs(bexcey) + std::fabs(cexbey)) * az + (std::fabs(cexaey) + std::fabs(aexcey)) * bz + (std::fabs(aexbey) + std::fabs(bexaey)) * cz) * dlift;
    const double err = 1.7763568394002532e-15 * perm;  // (16 + 224 eps) eps
    if (det > err) return 1;
    if (-det > err) return -1;
    return insphere_exact(a, b, c, d, e);
  }

  // is point e inside the open circumball of tetrahedron ti (a ghost: strictly beyond its hull face, or in its plane
  // and in conflict with the finite tetrahedron behind it)?
  bool conflict(int ti, int e) {
    const Tet& T = t[(size_t)ti];
    for (int i = 0; i < 4; ++i) {
      if (T.v[i] == INF) {
        const int a = T.v[FACE[i][0]], b = T.v[FACE[i][1]], c = T.v[FACE[i][2]];
        const int o = orient(a, b, c, e);
        if (o != 0) return o > 0;
        const Tet& U = t[(size_t)T.nb[i]];
        return insphere(U.v[0], U.v[1], U.v[2], U.v[3], e) > 0;
      }
    }
    return insphere(T.v[0], T.v[1], T.v[2], T.v[3], e) > 0;
  }
  bool is_ghost(int ti) const {
    const Tet& T = t[(size_t)ti];
    return T.v[0] == INF || T.v[1] == INF || T.v[2] == INF || T.v[3] == INF;
  }
  int new_tet() {
    if (!free_slots.empty()) {
      const int i = free_slots.back();
      free_slots.pop_back();
      return i;
    }
    t.push_back(Tet{});
    mark.push_back(-1);
    return (int)t.size() - 1;
  }
};
constexpr int Mesh::FACE[4][3];

// Links the faces of freshly made tetrahedra that meet in the new vertex: exactly two of them share every edge of the
// cavity's boundary.  A small open-addressing table keyed by the edge: the second face to arrive finds the first.
// Entries carry the stamp of the insertion they belong to, so the table is never cleared.
struct EdgeLinks {
  struct Ent { uint64_t key; int tet, face; uint32_t stamp; };
  std::vector<Ent> tab;
  size_t mask = 0, open = 0, used = 0;
  uint32_t stamp = 0;
  bool bad = false;
  void begin(size_t expected) {
    size_t cap = tab.size() ? tab.size() : 256;
    while (cap < 4 * expected) cap <<= 1;
    if (tab.size() != cap) { tab.assign(cap, Ent{0, -1, -1, 0}); stamp = 0; }
    if (++stamp == 0) { std::fill(tab.begin(), tab.end(), Ent{0, -1, -1, 0}); stamp = 1; }
    mask = cap - 1;
    open = used = 0;
    bad = false;
  }
  void add(std::vector<Mesh::Tet>& t, int a, int b, int tet, int face) {
    const uint64_t lo = (uint64_t)(a < b ? a : b), hi = (uint64_t)(a < b ? b : a);
    const uint64_t key = (hi << 32) | lo;
    size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & mask;
    for (size_t probes = 0; probes <= mask; ++probes, h = (h + 1) & mask) {
      Ent& e = tab[h];
      if (e.stamp != stamp) {
        e = Ent{key, tet, face, stamp};
        ++open;
        if (2 * ++used > mask + 1) bad = true;   // (table too full: the caller sized it from the cavity)
        return;
      }
      if (e.key == key) {
        if (e.tet < 0) { bad = true; return; }       // a third face on this edge: not a manifold boundary
        t[(size_t)e.tet].nb[e.face] = tet;
        t[(size_t)tet].nb[face] = e.tet;
        e.tet = -1;                                  // matched
        --open;
        return;
      }
    }
    bad = true;
  }
  bool done() const { return !bad && open == 0; }
};

uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  auto spread = [](uint64_t v) {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
  };
  return spread(x) | spread(y) << 1 | spread(z) << 2;
}

int g_local_max = 160;   // boundary vertices up to which the edges of a cavity's boundary are linked through the local table

}  // namespace

extern "C" int flooder_delaunay3d_local_edges(int max_vertices) {
  const int old = g_local_max;
  if (max_vertices >= 0 && max_vertices <= 4096) g_local_max = max_vertices;
  return old;
}

extern "C" int64_t flooder_delaunay3d(const double* pts, int64_t n, int32_t* tets, int64_t cap) {
  if (!pts || n < 5 || n > 0x3fffffff) return E_FEW;
  Mesh m;
  m.n = n;
  m.p = pts;
  m.INF = (int)n;
  // ---- integer coordinates on a common dyadic grid
  int emin, emax;
  if (!dyadic_range(pts, 3 * n, emin, emax)) return E_RANGE;
  if (emin > emax) return E_FLAT;            // (all coordinates zero)
  if (emax - emin > 57) return E_RANGE;      // would not fit 58-bit integers: not for this routine
  m.q.resize(3 * (size_t)n);
  for (int64_t i = 0; i < 3 * n; ++i) m.q[(size_t)i] = (int64_t)std::ldexp(pts[i], -emin);

  // ---- insertion order: Morton curve over the bounding box
  double lo[3] = {pts[0], pts[1], pts[2]}, hi[3] = {pts[0], pts[1], pts[2]};
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::min(lo[k], pts[3 * i + k]);
      hi[k] = std::max(hi[k], pts[3 * i + k]);
    }
  {
    // (1 + 2^-20: room for the roundings of the bound itself; differences of points inside the box are at most D,
    // rounded differences at most D (1 + eps), which the factor covers as well)
    double D = 0.0;
    for (int k = 0; k < 3; ++k) D = std::max(D, hi[k] - lo[k]);
    D *= 1.000001;
    m.orient_static = 7.771561172376103e-16 * 6.0 * D * D * D * 1.000001;
    m.insphere_static = 1.7763568394002532e-15 * 72.0 * D * D * D * D * D * 1.000001;
  }
  std::vector<std::pair<uint64_t, int>> order((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    uint32_t c[3];
    for (int k = 0; k < 3; ++k) {
      const double e = hi[k] - lo[k];
      const double u = e > 0 ? (pts[3 * i + k] - lo[k]) / e : 0.0;
      c[k] = (uint32_t)std::min(2097151.0, u * 2097152.0);
    }
    order[(size_t)i] = {morton3(c[0], c[1], c[2]), (int)i};
  }
  std::sort(order.begin(), order.end());

  // ---- first tetrahedron: four points of the order that are not coplanar
  int s[4] = {order[0].second, -1, -1, -1};
  {
    size_t i1 = 1;
    auto same = [&](int a, int b) { return pts[3 * a] == pts[3 * b] && pts[3 * a + 1] == pts[3 * b + 1] && pts[3 * a + 2] == pts[3 * b + 2]; };
    while (i1 < (size_t)n && same(s[0], order[i1].second)) ++i1;
    if (i1 == (size_t)n) return E_DUP;
    s[1] = order[i1].second;
    bool found = false;
    long tries = 0;
    for (size_t i2 = 1; i2 < (size_t)n && !found && tries < 2000000; ++i2) {
      const int c2 = order[i2].second;
      if (c2 == s[1]) continue;
      for (size_t i3 = i2 + 1; i3 < (size_t)n && tries < 2000000; ++i3) {
        const int c3 = order[i3].second;
        if (c3 == s[1]) continue;
        ++tries;
        if (m.orient(s[0], s[1], c2, c3) != 0) {
          s[2] = c2;
          s[3] = c3;
          found = true;
          break;
        }
        if (i3 > i2 + 64) break;  // (this c2 looks collinear with the first two: try the next)
      }
    }
    if (!found) return E_FLAT;
    if (m.orient(s[0], s[1], s[2], s[3]) < 0) std::swap(s[2], s[3]);
  }
  m.t.reserve((size_t)n * 8);
  m.mark.reserve((size_t)n * 8);
  EdgeLinks links;
  {
    const int t0 = m.new_tet();
    for (int i = 0; i < 4; ++i) m.t[(size_t)t0].v[i] = s[i];
    int g[4];
    for (int i = 0; i < 4; ++i) {
      g[i] = m.new_tet();
      Mesh::Tet& G = m.t[(size_t)g[i]];
      G.v[0] = s[Mesh::FACE[i][0]];
      G.v[1] = s[Mesh::FACE[i][2]];
      G.v[2] = s[Mesh::FACE[i][1]];
      G.v[3] = m.INF;
      G.nb[3] = t0;
      m.t[(size_t)t0].nb[i] = g[i];
    }
    links.begin(16);
    for (int i = 0; i < 4; ++i) {
      const Mesh::Tet G = m.t[(size_t)g[i]];
      links.add(m.t, G.v[1], G.v[2], g[i], 0);
      links.add(m.t, G.v[0], G.v[2], g[i], 1);
      links.add(m.t, G.v[0], G.v[1], g[i], 2);
    }
    if (!links.done()) return E_CAVITY;
  }
  int last = 0;  // a finite tetrahedron to start the walk from
  std::vector<int> cavity, stack, fresh, bfaces;
  struct LocalEdge { uint32_t stamp; int tf; };   // tf: 4 * tetrahedron + face waiting on this edge, -1 once matched
  const int LOCAL_MAX = g_local_max;
  std::vector<LocalEdge> etab;
  std::vector<uint32_t> vstamp((size_t)n + 1, 0u);
  std::vector<int> vloc((size_t)n + 1, 0);
  uint32_t vstamp_now = 0, estamp = 0;
  std::vector<char> used((size_t)n, 0);
  for (int i = 0; i < 4; ++i) used[(size_t)s[i]] = 1;

  for (size_t oi = 0; oi < (size_t)n; ++oi) {
    const int pi = order[oi].second;
    if (used[(size_t)pi]) continue;
    // ---- locate: visibility walk over the finite tetrahedra
    int cur = last;
    for (long steps = 0;; ++steps) {
      if (steps > 4 * (long)m.t.size() + 64) return E_LOCATE;
      if (m.is_ghost(cur)) break;
      const Mesh::Tet& T = m.t[(size_t)cur];
      int go = -1;
      for (int k = 0; k < 4; ++k) {
        const int i = (int)((steps + k) & 3);  // (a different first face every step)
        if (m.orient(T.v[Mesh::FACE[i][0]], T.v[Mesh::FACE[i][1]], T.v[Mesh::FACE[i][2]], pi) < 0) {
          go = i;
          break;
        }
      }
      if (go < 0) break;
      cur = T.nb[go];
    }
    if (!m.conflict(cur, pi)) {
      // (a point in the plane of a hull face, or a copy of a vertex: look around before giving up)
      int hit = -1;
      stack.assign(1, cur);
      std::vector<int> seen(1, cur);
      for (size_t h = 0; h < stack.size() && h < 256 && hit < 0; ++h) {
        for (int i = 0; i < 4 && hit < 0; ++i) {
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
    // ---- cavity: the connected set of tetrahedra in conflict with the point
    cavity.clear();
    stack.assign(1, cur);
    const int in_cav = 2 * pi, not_cav = 2 * pi + 1;   // (marks: in the cavity / tested and not in conflict)
    m.mark[(size_t)cur] = in_cav;
    while (!stack.empty()) {
      const int c = stack.back();
      stack.pop_back();
      cavity.push_back(c);
      for (int i = 0; i < 4; ++i) {
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
    // ---- a new tetrahedron on every boundary face of the cavity.  The faces of the new tetrahedra that meet in the
    // new vertex are linked through the boundary's edges: exactly two of them share every edge.  The boundary has a
    // few dozen vertices: they are numbered locally (stamped per insertion) and an edge is a cell of a K x K table -
    // no hashing, no probing (the hashed EdgeLinks, 66 cycles per face, was over half of an insertion); a boundary
    // of more than LOCAL_MAX vertices falls back to it.
    bfaces.clear();
    int K = 0;
    if (++vstamp_now == 0) { std::fill(vstamp.begin(), vstamp.end(), 0u); vstamp_now = 1; }
    for (const int c : cavity) {
      for (int i = 0; i < 4; ++i) {
        const int u = m.t[(size_t)c].nb[i];
        if (m.mark[(size_t)u] == in_cav) continue;   // (inside the cavity)
        bfaces.push_back(4 * c + i);
        for (int k = 0; k < 3; ++k) {
          const int v = m.t[(size_t)c].v[Mesh::FACE[i][k]];
          if (vstamp[(size_t)v] != vstamp_now) { vstamp[(size_t)v] = vstamp_now; vloc[(size_t)v] = K++; }
        }
      }
    }
    const bool local = K <= LOCAL_MAX;
    if (local) {
      if (etab.size() < (size_t)K * (size_t)K) etab.resize((size_t)K * (size_t)K, LocalEdge{0u, -1});
      if (++estamp == 0) { std::fill(etab.begin(), etab.end(), LocalEdge{0u, -1}); estamp = 1; }
    } else {
      links.begin(3 * cavity.size() + 16);   // (about 2 c + 2 boundary faces, 1.5 distinct edges each)
    }
    int open_edges = 0;
    bool bad_edge = false;
    auto link_local = [&](int a, int b, int tet, int face) {
      int la = vloc[(size_t)a], lb = vloc[(size_t)b];
      if (la > lb) std::swap(la, lb);
      LocalEdge& e = etab[(size_t)la * (size_t)K + (size_t)lb];
      if (e.stamp != estamp) {
        e.stamp = estamp;
        e.tf = 4 * tet + face;
        ++open_edges;
      } else if (e.tf < 0) {
        bad_edge = true;                          // a third face on this edge: not a manifold boundary
      } else {
        const int ot = e.tf >> 2, of = e.tf & 3;
        m.t[(size_t)ot].nb[of] = tet;
        m.t[(size_t)tet].nb[face] = ot;
        e.tf = -1;                                // matched
        --open_edges;
      }
    };
    fresh.clear();
    for (const int cf : bfaces) {
      const int c = cf >> 2, i = cf & 3;
      const int u = m.t[(size_t)c].nb[i];
      const int a = m.t[(size_t)c].v[Mesh::FACE[i][0]], b = m.t[(size_t)c].v[Mesh::FACE[i][1]],
                d = m.t[(size_t)c].v[Mesh::FACE[i][2]];
      const int nt = m.new_tet();   // (may move m.t: no references held across it)
      m.mark[(size_t)nt] = -1;
      Mesh::Tet& N = m.t[(size_t)nt];
      N.v[0] = a; N.v[1] = b; N.v[2] = d; N.v[3] = pi;
      N.nb[3] = u;
      Mesh::Tet& U = m.t[(size_t)u];
      int back = -1;
      for (int j = 0; j < 4; ++j)
        if (U.nb[j] == c) {
          // (two tetrahedra can share two faces only in degenerate cavities: the face must also match)
          const int x = U.v[Mesh::FACE[j][0]], y = U.v[Mesh::FACE[j][1]], z = U.v[Mesh::FACE[j][2]];
          if ((x == a || x == b || x == d) && (y == a || y == b || y == d) && (z == a || z == b || z == d)) back = j;
        }
      if (back < 0) return E_CAVITY;
      U.nb[back] = nt;
      if (local) {
        link_local(b, d, nt, 0);
        link_local(a, d, nt, 1);
        link_local(a, b, nt, 2);
      } else {
        links.add(m.t, b, d, nt, 0);
        links.add(m.t, a, d, nt, 1);
        links.add(m.t, a, b, nt, 2);
      }
      fresh.push_back(nt);
    }
    if (local ? (bad_edge || open_edges != 0) : !links.done()) return E_CAVITY;
    for (const int c : cavity) {
      m.t[(size_t)c].v[0] = -2;   // dead
      m.mark[(size_t)c] = -1;
      m.free_slots.push_back(c);
    }
    // (freed slots may be handed out again only from the next insertion on: fresh tetrahedra never reuse a cavity
    // slot of the same insertion because new_tet ran before the slots were freed)
    for (const int f : fresh)
      if (!m.is_ghost(f)) { last = f; break; }
    used[(size_t)pi] = 1;
  }
  int64_t count = 0;
  for (size_t i = 0; i < m.t.size(); ++i)
    if (m.t[i].v[0] != -2 && !m.is_ghost((int)i)) ++count;
  if (count > cap || !tets) return -count;
  int64_t o = 0;
  for (size_t i = 0; i < m.t.size(); ++i)
    if (m.t[i].v[0] != -2 && !m.is_ghost((int)i)) {
      for (int k = 0; k < 4; ++k) tets[4 * o + k] = m.t[i].v[k];
      ++o;
    }
  return count;
}
