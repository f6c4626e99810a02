// flood_planes.hpp - the face-plane table row of one simplex (flood_cell.hip's comment on the slab test says what
// the numbers are for): shared by simplex_planes_kernel (flood_cell.hip) and the launch that prepares a sweep
// (simplex weights + plane rows + the zero fill of the control words in ONE launch: flood_finish.hip).
#pragma once
#include "flood_common.hpp"

namespace flooder {

constexpr int PLANE_ROW = 24;

template <int DIM>
__device__ __forceinline__ void simplex_planes_row(const float* __restrict__ verts, int k1, int64_t s, float* __restrict__ tab) {
  const float* vs = verts + s * (int64_t)k1 * DIM;
  float pn[DIM + 1][DIM], po[DIM + 1], pslack[DIM + 1], org[DIM];
  float sext2 = 0.f;  // squared extent of the simplex around org
#pragma unroll
  for (int k = 0; k < DIM; ++k) org[k] = vs[k];
  for (int j = 1; j < k1; ++j) {
    float e2 = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) e2 = __builtin_fmaf(vs[j * DIM + k] - org[k], vs[j * DIM + k] - org[k], e2);
    sext2 = __builtin_fmaxf(sext2, e2);
  }
  const float sext = __builtin_sqrtf(sext2);
#pragma unroll
  for (int f = 0; f <= DIM; ++f) {
    po[f] = 3.0e38f;  // disabled plane: the test always passes
    pslack[f] = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) pn[f][k] = 0.f;
  }
  if (k1 == DIM + 1) {
#pragma unroll
    for (int f = 0; f <= DIM; ++f) {
      int id[DIM];
      int qq = 0;
#pragma unroll
      for (int j = 0; j <= DIM; ++j)
        if (j != f) id[qq++] = j;
      float nrm[DIM];
      float l12;  // |e1|^2 |e2|^2 (3D) or |e|^2 (2D): len2 / l12 = sin^2 of the angle between the edges
      if constexpr (DIM == 3) {
        float e1[3], e2[3];
        float l1 = 0.f, l2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          e1[k] = vs[id[1] * 3 + k] - vs[id[0] * 3 + k];
          e2[k] = vs[id[2] * 3 + k] - vs[id[0] * 3 + k];
          l1 = __builtin_fmaf(e1[k], e1[k], l1);
          l2 = __builtin_fmaf(e2[k], e2[k], l2);
        }
        nrm[0] = e1[1] * e2[2] - e1[2] * e2[1];
        nrm[1] = e1[2] * e2[0] - e1[0] * e2[2];
        nrm[2] = e1[0] * e2[1] - e1[1] * e2[0];
        l12 = l1 * l2;
      } else {
        const float ex = vs[id[1] * 2 + 0] - vs[id[0] * 2 + 0];
        const float ey = vs[id[1] * 2 + 1] - vs[id[0] * 2 + 1];
        nrm[0] = ey;
        nrm[1] = -ex;
        l12 = ex * ex + ey * ey;
      }
      float len2 = 0.f, side = 0.f;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        len2 = __builtin_fmaf(nrm[k], nrm[k], len2);
        side = __builtin_fmaf(nrm[k], vs[f * DIM + k] - vs[id[0] * DIM + k], side);
      }
      const bool ok = len2 > 1e-30f && len2 >= 1e-8f * l12 && side * side >= 1e-8f * len2 * sext2;
      const float sc = ok ? (side > 0.f ? -1.f : 1.f) / __builtin_sqrtf(len2) : 0.f;
      float off = 0.f;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        pn[f][k] = nrm[k] * sc;
        off = __builtin_fmaf(pn[f][k], vs[id[0] * DIM + k] - org[k], off);
      }
      po[f] = ok ? off : 3.0e38f;
      pslack[f] = ok ? 1e-6f * __builtin_sqrtf(l12 / len2) : 0.f;
    }
  }
  float* row = tab + s * PLANE_ROW;
#pragma unroll
  for (int i = 0; i < PLANE_ROW; ++i) row[i] = 0.f;
#pragma unroll
  for (int k = 0; k < DIM; ++k) row[k] = org[k];
  row[3] = sext;
#pragma unroll
  for (int f = 0; f <= DIM; ++f) {
#pragma unroll
    for (int k = 0; k < DIM; ++k) row[4 + 5 * f + k] = pn[f][k];
    row[4 + 5 * f + 3] = po[f];
    row[4 + 5 * f + 4] = pslack[f];
  }
}

}  // namespace flooder
