// flood_f64.hip - float64 coverage sweep (gfx950): the reference's kernels take DTYPE = fp64 for float64 inputs
// (flooder/triton_kernels.py:226-229, test tests/test_flooder.py:214-246: float32 and float64 results within 3e-6).
//
// Same culled exact nearest-neighbour sweep as flood_bvh.hip - one wave per tile of 64 samples, wave-uniform
// nearest-first traversal of the box tree, lane = child box - with the samples, the points and every distance in
// double precision.  The tree itself is the float32 one of the sorted cloud (PointIndex): its boxes were built from
// the float32-rounded coordinates, so each box is widened by one float32 ulp per side before use and all bounds
// are evaluated in double; a bound is a true lower bound of the double-precision distance, and the result is the
// exact double-precision minimum over all points (direct differences, fma chain).

#include "flood_common.hpp"
#include "flood_bvh.hpp"

using namespace flooder;

namespace {

constexpr double SAFE64 = 1.0 - 1e-12;

__device__ __forceinline__ double wave_min_f64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o));
  return x;
}
__device__ __forceinline__ double wave_max_f64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
  return x;
}
// one float32 ulp outwards (the box was computed from coordinates rounded to float32)
__device__ __forceinline__ float step_down(float v) {  // next float32 below v (infinities of empty boxes stay)
  if (!(v > -3.4e38f && v < 3.4e38f)) return v;
  uint32_t b = __float_as_uint(v);
  if (v > 0.f) b -= 1u;
  else if (v < 0.f) b += 1u;
  else b = 0x80000001u;  // below (+/-)0: the smallest negative denormal
  return __uint_as_float(b);
}
__device__ __forceinline__ double widen_lo(float v) { return (double)step_down(v); }
__device__ __forceinline__ double widen_hi(float v) { return -(double)step_down(-v); }

template <int DIM>
__global__ __launch_bounds__(256) void sweep_bvh_f64_kernel(
    const double* __restrict__ pts, const float* __restrict__ nodes, Levels lv, const double* __restrict__ verts,
    const double* __restrict__ weights, int k1, int R, int64_t n_simplices, int32_t* __restrict__ queue,
    unsigned long long* __restrict__ out_d2) {
  constexpr int DP = padded_dim(DIM);
  __shared__ double s_lb[4][MAXL][FAN];
  __shared__ int64_t s_grp[4][MAXL];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int tiles = (R + 63) >> 6;
  const int64_t n_items = n_simplices * tiles;
  const int top = lv.n_levels - 1;
  const double INF = __builtin_inf();
  int q_shard = (int)((blockIdx.x * 4 + (threadIdx.x >> 6)) % QSHARDS), q_tried = 0;
  for (;;) {
    const int64_t g = queue_pop(queue, q_shard, q_tried, n_items, lane);  // sharded heads (flood_common.hpp)
    if (g < 0) break;
    const int64_t s = g / tiles;
    const int tile = (int)(g - s * tiles);
    int r = tile * 64 + lane;
    const bool exists = r < R;
    if (!exists) r = R - 1;
    // sample p = sum_j w[r, j] * v[s, j, :] in double (core.py:188)
    double p[DIM];
    const double* vs = verts + s * (int64_t)k1 * DIM;
#pragma unroll
    for (int k = 0; k < DIM; ++k) p[k] = 0.0;
    for (int j = 0; j < k1; ++j) {
      const double w = weights[(int64_t)r * k1 + j];
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[k] = __builtin_fma(w, vs[j * DIM + k], p[k]);
    }
    double best = INF;
    double tlo[DIM], thi[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      tlo[k] = wave_min_f64(p[k]);
      thi[k] = wave_max_f64(p[k]);
    }
    double M = INF;  // largest running minimum of the tile

    double c_lo[DIM], c_hi[DIM];
    auto child_bounds = [&](int lvl, int64_t grp) -> double {
      const int64_t idx = grp * FAN + lane;
      double lb = INF;
#pragma unroll
      for (int k = 0; k < DIM; ++k) { c_lo[k] = INF; c_hi[k] = -INF; }
      if (idx < lv.count[lvl]) {
        float lo[DP], hi[DP];
        const float* nb = nodes + (lv.off[lvl] + idx) * 2 * DP;
        load_row<DP>(nb, lo);
        load_row<DP>(nb + DP, hi);
        lb = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          c_lo[k] = widen_lo(lo[k]);
          c_hi[k] = widen_hi(hi[k]);
          const double gap = fmax(fmax(c_lo[k] - thi[k], tlo[k] - c_hi[k]), 0.0);
          lb = __builtin_fma(gap, gap, lb);
        }
      }
      return lb;
    };

    int lvl = top;
    double lb0 = child_bounds(top, 0);
    int64_t grp0 = 0;
    if (top > 0) {
      s_lb[wv][top][lane] = lb0;
      if (lane == 0) s_grp[wv][top] = 0;
    }
    for (;;) {
      if (lvl > 0) {
        const double lbv = s_lb[wv][lvl][lane];
        const double mn = wave_min_f64(lbv);
        if (!(mn * SAFE64 < M)) {
          if (++lvl > top) break;
          continue;
        }
        const int j = __builtin_ctzll(__ballot(lbv == mn));
        if (lane == j) s_lb[wv][lvl][lane] = INF;  // visited
        const int64_t c = wave_uniform64(s_grp[wv][lvl]) * FAN + j;  // (readfirstlane: keeps the leaf address scalar, rows in SGPRs)
        --lvl;
        const double lb = child_bounds(lvl, c);
        if (lvl > 0) {
          s_lb[wv][lvl][lane] = lb;
          if (lane == 0) s_grp[wv][lvl] = c;
        } else {
          lb0 = lb;
          grp0 = c;
        }
        continue;
      }
      // leaf level: nearest unvisited leaf of the current group
      const double mn = wave_min_f64(lb0);
      if (!(mn * SAFE64 < M)) {
        if (++lvl > top) break;
        continue;
      }
      const int j = __builtin_ctzll(__ballot(lb0 == mn));
      if (lane == j) lb0 = INF;  // visited
      const int64_t c = grp0 * FAN + j;
      double lbp = 0.0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const double blo = __shfl(c_lo[k], j), bhi = __shfl(c_hi[k], j);
        const double gap = fmax(fmax(blo - p[k], p[k] - bhi), 0.0);
        lbp = __builtin_fma(gap, gap, lbp);
      }
      if (__ballot(lbp * SAFE64 < best) == 0ull) continue;
      const double* cp = pts + c * (int64_t)LEAF * DP;  // (wave-uniform address: the rows come through the scalar cache)
#pragma unroll 4
      for (int h = 0; h < LEAF; ++h) {
        double d2 = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const double t = p[k] - cp[h * DP + k];
          d2 = k == 0 ? t * t : __builtin_fma(t, t, d2);
        }
        best = fmin(best, d2);
      }
      M = wave_max_f64(best);
    }
    if (exists) out_d2[s * (int64_t)R + r] = (unsigned long long)__double_as_longlong(best);
  }
}

__global__ __launch_bounds__(256) void face_max_f64_kernel(const unsigned long long* __restrict__ d2, int R,
                                                           const int32_t* __restrict__ face_ptr,
                                                           const int32_t* __restrict__ face_rows, int n_faces,
                                                           double* __restrict__ out_face, double* __restrict__ out_dist) {
  const int64_t s = blockIdx.x;
  const unsigned long long* row = d2 + s * (int64_t)R;
  __shared__ unsigned long long red[4];
  if (out_dist)
    for (int r = threadIdx.x; r < R; r += blockDim.x) out_dist[s * (int64_t)R + r] = sqrt(__longlong_as_double((long long)row[r]));
  for (int f = 0; f < n_faces; ++f) {
    const int b = face_ptr[f], e = face_ptr[f + 1];
    unsigned long long m = 0ull;  // (non-negative doubles order like their bit patterns)
    for (int q = b + threadIdx.x; q < e; q += blockDim.x) {
      const unsigned long long v = row[face_rows[q]];
      m = v > m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_xor(m, o);
      m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long a = red[0];
      for (int w = 1; w < 4; ++w) a = red[w] > a ? red[w] : a;
      out_face[s * (int64_t)n_faces + f] = sqrt(__longlong_as_double((long long)a));
    }
    __syncthreads();
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void gather_rows_f64_kernel(const double* __restrict__ pts, int64_t n, int ld,
                                                              const uint32_t* __restrict__ order,
                                                              double* __restrict__ out, int64_t n_pad) {
  constexpr int DP = padded_dim(DIM);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_pad; j += stride) {
    const bool real = j < n;
    const int64_t src = real ? (int64_t)order[j] : 0;
#pragma unroll
    for (int k = 0; k < DP; ++k)
      out[j * DP + k] = real ? (k < DIM ? pts[src * ld + k] : 0.0) : __builtin_inf();
  }
}

template <int DIM>
struct Sweep64Op {
  static int run(const double* pts, const float* nodes, const Levels& lv, const double* verts, const double* weights,
                 int k1, int R, int64_t ns, int32_t* queue, unsigned long long* out, hipStream_t st) {
    hipLaunchKernelGGL((sweep_bvh_f64_kernel<DIM>), dim3(g_bvh_grid), dim3(256), 0, st, pts, nodes, lv, verts, weights,
                       k1, R, ns, queue, out);
    return check_launch("sweep_bvh_f64");
  }
};

template <int DIM>
struct Gather64Op {
  static int run(const double* pts, int64_t n, int ld, const uint32_t* order, double* out, int64_t n_pad,
                 hipStream_t st) {
    int64_t blocks = (n_pad + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((gather_rows_f64_kernel<DIM>), dim3((int)blocks), dim3(256), 0, st, pts, n, ld, order, out, n_pad);
    return check_launch("gather_rows_f64");
  }
};

}  // namespace

extern "C" {

int flooder_gather_rows_f64(const double* pts, int64_t n_pts, int dim, int ld, const int32_t* order, double* out,
                            int64_t n_pad, void* stream) {
  if (!pts || !order || !out || n_pts < 1 || n_pad < n_pts || ld < dim)
    return fail(FLOODER_E_ARG, "flooder_gather_rows_f64: bad argument");
  return dispatch_dim<Gather64Op>(dim, pts, n_pts, ld, reinterpret_cast<const uint32_t*>(order), out, n_pad,
                                  (hipStream_t)stream);
}

int flooder_sweep_bvh_f64(const double* pts_sorted, int64_t n_pts, int dim, const float* nodes, const double* verts,
                          const double* weights, int k1, int R, int64_t n_simplices, int32_t* queue,
                          uint64_t* out_d2, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !queue || !out_d2 || n_pts < 1 || k1 < 1 ||
      k1 > FLOODER_MAX_VERTS || R < 0)
    return fail(FLOODER_E_ARG, "flooder_sweep_bvh_f64: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<Sweep64Op>(dim, pts_sorted, nodes, lv, verts, weights, k1, R, n_simplices, queue,
                                 reinterpret_cast<unsigned long long*>(out_d2), (hipStream_t)stream);
}

int flooder_face_max_f64(const uint64_t* d2, int64_t n_simplices, int R, const int32_t* face_ptr,
                         const int32_t* face_rows, int n_faces, double* out_face, double* out_dist, void* stream) {
  if (n_simplices == 0) return FLOODER_OK;
  if (!d2 || !face_ptr || !face_rows || !out_face || n_faces < 1 || R < 1 || n_simplices > 0x7fffffff)
    return fail(FLOODER_E_ARG, "flooder_face_max_f64: bad argument");
  hipLaunchKernelGGL(face_max_f64_kernel, dim3((unsigned)n_simplices), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const unsigned long long*>(d2), R, face_ptr, face_rows, n_faces, out_face, out_dist);
  return check_launch("face_max_f64");
}

}  // extern "C"
