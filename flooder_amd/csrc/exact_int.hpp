// exact_int.hpp - what the host Delaunay routines share (delaunay2d.cpp, delaunay3d.cpp): the codes with which they
// decline an input, and 512-bit integers for the exact stage of their predicates.  Internal; the C ABI is
// include/flooder_host.h.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr int64_t E_BASE = -((int64_t)1 << 40);
constexpr int64_t E_FEW = E_BASE - 1, E_RANGE = E_BASE - 2, E_FLAT = E_BASE - 3, E_DUP = E_BASE - 4, E_CAVITY = E_BASE - 5,
                  E_LOCATE = E_BASE - 6;

// ---- 512-bit two's complement integers: just enough for the exact predicates
struct Big {
  static constexpr int L = 8;
  uint64_t w[L];
};
inline Big big_from(__int128 v) {
  Big r;
  r.w[0] = (uint64_t)v;
  r.w[1] = (uint64_t)(v >> 64);
  const uint64_t ext = v < 0 ? ~0ull : 0ull;
  for (int i = 2; i < Big::L; ++i) r.w[i] = ext;
  return r;
}
inline bool big_neg(const Big& a) { return (a.w[Big::L - 1] >> 63) != 0; }
inline Big big_add(const Big& a, const Big& b) {
  Big r;
  unsigned __int128 c = 0;
  for (int i = 0; i < Big::L; ++i) {
    c += (unsigned __int128)a.w[i] + b.w[i];
    r.w[i] = (uint64_t)c;
    c >>= 64;
  }
  return r;
}
inline Big big_negate(const Big& a) {
  Big r;
  unsigned __int128 c = 1;
  for (int i = 0; i < Big::L; ++i) {
    c += (unsigned __int128)(~a.w[i]);
    r.w[i] = (uint64_t)c;
    c >>= 64;
  }
  return r;
}
inline Big big_sub(const Big& a, const Big& b) { return big_add(a, big_negate(b)); }
inline Big big_mul(const Big& a, const Big& b) {  // (magnitudes stay far below 2^511 here: no overflow check)
  const bool na = big_neg(a), nb = big_neg(b);
  const Big x = na ? big_negate(a) : a, y = nb ? big_negate(b) : b;
  Big r;
  std::memset(r.w, 0, sizeof(r.w));
  for (int i = 0; i < Big::L; ++i) {
    if (!x.w[i]) continue;
    unsigned __int128 c = 0;
    for (int j = 0; i + j < Big::L; ++j) {
      c += (unsigned __int128)x.w[i] * y.w[j] + r.w[i + j];
      r.w[i + j] = (uint64_t)c;
      c >>= 64;
    }
  }
  return na != nb ? big_negate(r) : r;
}
inline int big_sign(const Big& a) {
  if (big_neg(a)) return -1;
  for (int i = 0; i < Big::L; ++i)
    if (a.w[i]) return 1;
  return 0;
}

// Exponent range of n finite doubles: the exponent of the lowest set mantissa bit (emin) and of the value (emax) over
// the non-zero ones; false for a non-finite value.  emax - emin <= 57 means the values scale to 58-bit integers.
inline bool dyadic_range(const double* x, int64_t n, int& emin, int& emax) {
  emin = 1 << 30;
  emax = -(1 << 30);
  for (int64_t i = 0; i < n; ++i) {
    if (!std::isfinite(x[i])) return false;
    if (x[i] == 0.0) continue;
    int e;
    const double f = std::frexp(std::fabs(x[i]), &e);          // |x| = f 2^e, f in [0.5, 1)
    const uint64_t M = (uint64_t)std::ldexp(f, 53);              // 53-bit integer mantissa
    const int low = e - 53 + __builtin_ctzll(M);                 // exponent of the lowest set bit
    emin = low < emin ? low : emin;
    emax = e > emax ? e : emax;
  }
  return true;
}

}  // namespace
