// flood_fps.hip - bucketed exact farthest-point sampling over the curve-sorted cloud (gfx950; dim <= 3).
//
// Replaces fpsample.bucket_fps_kdline_sampling as called by generate_landmarks (flooder/core.py:329-343): the
// reference's library keeps the cloud in kd-tree buckets and, for a new landmark, only updates the buckets it can
// still lower.  Same idea on the Hilbert-sorted copy of the cloud the sweeps use anyway:
//
//   rows     (x, y, z, running min d^2) per point in curve order, 16 B;
//   bucket   RPL*64 consecutive rows: bounding box + key = (largest running minimum << 32 | ~original index);
//   step     one launch per landmark.  Every lane owns buckets (interleaved over the waves, so a landmark's
//            neighbourhood spreads over the whole chip); a bucket whose box is at least as far from the new landmark
//            as its largest running minimum cannot change (d^2 >= box bound >= every minimum in it) and is skipped
//            with one box test; the others are updated by the whole wave (RPL rows per lane), re-reduced, and the
//            wave's best key goes to one of 64 arg-max slots.
// The arithmetic per point is that of the brute-force kernels (fps_fast_kernel), direct differences with the same
// fma chain, and the box bound uses the same chain on |gaps| <= |differences|, so the selection is bit-identical to
// the brute-force order (ties: lowest original index, as numpy's argmax).  The first iterations (most buckets
// touched) run brute force over the sorted rows; the bucket keys are built when the bucketed steps take over.

#include "flood_common.hpp"

using namespace flooder;

namespace {

constexpr int SLOTS = 64;  // arg-max slots per iteration (same-address atomics serialise at ~13 ns each)

__device__ __forceinline__ uint32_t winner_of(const unsigned long long* __restrict__ best, int it) {
  const unsigned long long k = best[(int64_t)it * SLOTS + (threadIdx.x & 63)];
  const float m = __uint_as_float((uint32_t)(k >> 32));
  const uint32_t lowinv = (uint32_t)(k & 0xffffffffu);
  const float wm = wave_max_f32(m);
  return wave_min_u32((m == wm && k != 0ull) ? 0xffffffffu - lowinv : 0xffffffffu);
}

// wave-wide maximum of 64-bit keys held as (hi, lo)
__device__ __forceinline__ unsigned long long wave_max_key(uint32_t hi, uint32_t lo) {
  const uint32_t mh = wave_max_u32(hi);
  const uint32_t ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | (unsigned long long)ml;
}

template <int DP>
__global__ __launch_bounds__(256) void fps_rows_from_sorted_kernel(const float* __restrict__ pts_sorted, int64_t n_pad,
                                                                   int dim, float4* __restrict__ rows) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_pad; j += stride) {
    float x[DP];
    load_row<DP>(pts_sorted + j * DP, x);
    float4 r;
    r.x = x[0];
    r.y = dim > 1 ? x[DP > 1 ? 1 : 0] : 0.f;
    r.z = dim > 2 ? x[DP > 2 ? 2 : 0] : 0.f;
    r.w = __builtin_inff();
    rows[j] = r;
  }
}

__device__ __forceinline__ float4 centre_of(const float* __restrict__ pts, int ld, int dim, uint32_t q) {
  float4 c;
  c.x = pts[(int64_t)q * ld];
  c.y = dim > 1 ? pts[(int64_t)q * ld + 1] : 0.f;
  c.z = dim > 2 ? pts[(int64_t)q * ld + 2] : 0.f;
  c.w = 0.f;
  return c;
}

// brute-force step over the sorted rows (first iterations): 4 rows per thread, keys carry ORIGINAL indices
__global__ __launch_bounds__(256) void fps_sorted_step_kernel(float4* __restrict__ rows, int64_t n,
                                                              const int32_t* __restrict__ order,
                                                              const float* __restrict__ pts, int ld, int dim, int it,
                                                              unsigned long long* __restrict__ best,
                                                              int64_t* __restrict__ out_idx) {
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  float4 r[4];
  uint32_t o[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t j = base + u * 256;
    r[u] = rows[j < n ? j : n - 1];
    o[u] = (uint32_t)order[j < n ? j : n - 1];
  }
  const uint32_t q = winner_of(best, it - 1);
  const float4 c = centre_of(pts, ld, dim, q);
  if (blockIdx.x == 0 && threadIdx.x == 0) out_idx[it - 1] = (int64_t)q;
  uint32_t bh = 0u, bl = 0u;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t j = base + u * 256;
    if (j < n) {
      float t = r[u].x - c.x;
      float d2 = t * t;
      t = r[u].y - c.y;
      d2 = __builtin_fmaf(t, t, d2);
      t = r[u].z - c.z;
      d2 = __builtin_fmaf(t, t, d2);
      const float m = d2 < r[u].w ? d2 : r[u].w;
      if (m < r[u].w) reinterpret_cast<float*>(rows + j)[3] = m;
      const uint32_t h = __float_as_uint(m), l = 0xffffffffu - o[u];
      if (h > bh || (h == bh && l > bl)) { bh = h; bl = l; }
    }
  }
  const unsigned long long wk = wave_max_key(bh, bl);
  __shared__ unsigned long long s_k[4];
  if ((threadIdx.x & 63) == 0) s_k[threadIdx.x >> 6] = wk;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long k = s_k[0];
    for (int w = 1; w < 4; ++w) k = s_k[w] > k ? s_k[w] : k;
    atomicMax(&best[(int64_t)it * SLOTS + (blockIdx.x % SLOTS)], k);
  }
}

// bucket b = RPL*64 rows: box + key from the current rows (one wave per bucket)
template <int RPL>
__global__ __launch_bounds__(256) void fps_bucket_init_kernel(const float4* __restrict__ rows, int64_t n,
                                                              const int32_t* __restrict__ order, int64_t n_buckets,
                                                              float* __restrict__ box,
                                                              unsigned long long* __restrict__ key) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n_buckets) return;
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  uint32_t bh = 0u, bl = 0u;
#pragma unroll
  for (int u = 0; u < RPL; ++u) {
    const int64_t j = b * (RPL * 64) + u * 64 + lane;
    if (j < n) {
      const float4 r = rows[j];
      lo[0] = __builtin_fminf(lo[0], r.x); hi[0] = __builtin_fmaxf(hi[0], r.x);
      lo[1] = __builtin_fminf(lo[1], r.y); hi[1] = __builtin_fmaxf(hi[1], r.y);
      lo[2] = __builtin_fminf(lo[2], r.z); hi[2] = __builtin_fmaxf(hi[2], r.z);
      const uint32_t h = __float_as_uint(r.w), l = 0xffffffffu - (uint32_t)order[j];
      if (h > bh || (h == bh && l > bl)) { bh = h; bl = l; }
    }
  }
  const unsigned long long k = wave_max_key(bh, bl);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = wave_min_f32(lo[d]);
    hi[d] = wave_max_f32(hi[d]);
  }
  if (lane == 0) {
    float* bb = box + b * 8;
    bb[0] = lo[0]; bb[1] = lo[1]; bb[2] = lo[2]; bb[3] = 0.f;
    bb[4] = hi[0]; bb[5] = hi[1]; bb[6] = hi[2]; bb[7] = 0.f;
    key[b] = k;
  }
}

// one bucketed step: lane l of wave w owns bucket l * n_waves + w
template <int RPL>
__global__ __launch_bounds__(256) void fps_bucket_step_kernel(float4* __restrict__ rows, int64_t n,
                                                              const int32_t* __restrict__ order,
                                                              const float* __restrict__ pts, int ld, int dim,
                                                              int64_t n_buckets, const float* __restrict__ box,
                                                              unsigned long long* __restrict__ key, int it,
                                                              unsigned long long* __restrict__ best,
                                                              int64_t* __restrict__ out_idx) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t b = (int64_t)lane * n_waves + w;
  const bool has = b < n_buckets;
  // this lane's bucket first: the loads are in flight while the previous winner is reduced and fetched
  float4 blo = make_float4(0.f, 0.f, 0.f, 0.f), bhi = blo;
  unsigned long long k = 0ull;
  if (has) {
    blo = *reinterpret_cast<const float4*>(box + b * 8);
    bhi = *reinterpret_cast<const float4*>(box + b * 8 + 4);
    k = key[b];
  }
  const uint32_t q = winner_of(best, it - 1);
  const float4 c = centre_of(pts, ld, dim, q);
  if (blockIdx.x == 0 && threadIdx.x == 0) out_idx[it - 1] = (int64_t)q;
  // lower bound of d^2 from c to any point of the bucket: the per-point chain on |gap| <= |difference|
  float g = __builtin_fmaxf(__builtin_fmaxf(blo.x - c.x, c.x - bhi.x), 0.f);
  float lb = g * g;
  g = __builtin_fmaxf(__builtin_fmaxf(blo.y - c.y, c.y - bhi.y), 0.f);
  lb = __builtin_fmaf(g, g, lb);
  g = __builtin_fmaxf(__builtin_fmaxf(blo.z - c.z, c.z - bhi.z), 0.f);
  lb = __builtin_fmaf(g, g, lb);
  const bool touched = has && lb < __uint_as_float((uint32_t)(k >> 32));
  unsigned long long tm = __ballot(touched);
  while (tm) {  // (wave-uniform) the whole wave updates one touched bucket at a time
    const int src = __builtin_ctzll(tm);
    tm &= tm - 1ull;
    const int64_t tb = (int64_t)src * n_waves + w;
    uint32_t bh = 0u, bl = 0u;
    float4 r[RPL];
    uint32_t o[RPL];
#pragma unroll
    for (int u = 0; u < RPL; ++u) {
      const int64_t j = tb * (RPL * 64) + u * 64 + lane;
      r[u] = rows[j < n ? j : n - 1];
      o[u] = (uint32_t)order[j < n ? j : n - 1];
    }
#pragma unroll
    for (int u = 0; u < RPL; ++u) {
      const int64_t j = tb * (RPL * 64) + u * 64 + lane;
      if (j < n) {
        float t = r[u].x - c.x;
        float d2 = t * t;
        t = r[u].y - c.y;
        d2 = __builtin_fmaf(t, t, d2);
        t = r[u].z - c.z;
        d2 = __builtin_fmaf(t, t, d2);
        const float m = d2 < r[u].w ? d2 : r[u].w;
        if (m < r[u].w) reinterpret_cast<float*>(rows + j)[3] = m;
        const uint32_t h = __float_as_uint(m), l = 0xffffffffu - o[u];
        if (h > bh || (h == bh && l > bl)) { bh = h; bl = l; }
      }
    }
    const unsigned long long nk = wave_max_key(bh, bl);
    if (lane == src) {
      k = nk;
      key[tb] = nk;
    }
  }
  const unsigned long long wk = wave_max_key((uint32_t)(k >> 32), (uint32_t)(k & 0xffffffffu));
  if (lane == 0 && wk != 0ull) atomicMax(&best[(int64_t)it * SLOTS + (w % SLOTS)], wk);
}

__global__ void fps_start_kernel(unsigned long long* best, int64_t start) {
  // iteration 0 "winner" = the start point: a positive distance so that the slot counts as filled
  best[0] = ((unsigned long long)__float_as_uint(1.0f) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)start);
}

__global__ void fps_last_kernel2(const unsigned long long* best, int n_lms, int64_t* out_idx) {
  const uint32_t q = winner_of(best, n_lms - 1);
  if (threadIdx.x == 0) out_idx[n_lms - 1] = (int64_t)q;
}

template <int RPL>
void launch_bucketed(float4* rows, int64_t n, const int32_t* order, const float* pts, int ld, int dim, int n_lms,
                     int k0, float* box, unsigned long long* key, unsigned long long* best, int64_t* out_idx,
                     hipStream_t st) {
  const int64_t n_buckets = (n + RPL * 64 - 1) / (RPL * 64);
  const int64_t n_waves = (n_buckets + 63) / 64;
  const unsigned grid = (unsigned)((n_waves + 3) / 4);
  const int64_t brute_blocks = (n + 1023) / 1024;
  int it = 1;
  for (; it < n_lms && it < k0; ++it)
    hipLaunchKernelGGL(fps_sorted_step_kernel, dim3((unsigned)brute_blocks), dim3(256), 0, st, rows, n, order, pts, ld,
                       dim, it, best, out_idx);
  if (it < n_lms) {
    hipLaunchKernelGGL((fps_bucket_init_kernel<RPL>), dim3((unsigned)((n_buckets + 3) / 4)), dim3(256), 0, st, rows, n,
                       order, n_buckets, box, key);
    for (; it < n_lms; ++it)
      hipLaunchKernelGGL((fps_bucket_step_kernel<RPL>), dim3(grid), dim3(256), 0, st, rows, n, order, pts, ld, dim,
                         n_buckets, box, key, it, best, out_idx);
  }
}

}  // namespace

namespace flooder { int g_fps_switch = 0; int g_fps_rpl = 0; int g_fps_rounds = 0; int g_fps_lane_best = 0; }

extern "C" {

int64_t flooder_fps_bucket_count(int64_t n_pts) { return (n_pts + 63) / 64; }

int flooder_fps_indexed_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* pts_sorted,
                            const int32_t* order, int n_lms, int64_t start, int64_t* out_idx, float* rows,
                            float* bucket_box, uint64_t* bucket_key, uint64_t* work_best, void* stream) {
  if (!pts || !pts_sorted || !order || !out_idx || !rows || !bucket_box || !bucket_key || !work_best || n_pts < 1 ||
      n_lms < 1 || n_lms > n_pts || start < 0 || start >= n_pts || ld < dim || dim < 1 || dim > 3 ||
      n_pts > 0xfffffffeLL)
    return fail(FLOODER_E_ARG, "flooder_fps_indexed_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* best = reinterpret_cast<unsigned long long*>(work_best);
  unsigned long long* key = reinterpret_cast<unsigned long long*>(bucket_key);
  float4* r4 = reinterpret_cast<float4*>(rows);
  int64_t ib = (n_pts + 255) / 256;
  if (ib > 4096) ib = 4096;
  if (dim <= 2)
    hipLaunchKernelGGL((fps_rows_from_sorted_kernel<2>), dim3((int)ib), dim3(256), 0, st, pts_sorted, n_pts, dim, r4);
  else
    hipLaunchKernelGGL((fps_rows_from_sorted_kernel<4>), dim3((int)ib), dim3(256), 0, st, pts_sorted, n_pts, dim, r4);
  hipLaunchKernelGGL(fps_start_kernel, dim3(1), dim3(1), 0, st, best, start);
  // small clouds: 64-row buckets (more waves to spread a landmark's neighbourhood over) and a late switch (a brute
  // step over an L2-resident cloud costs about as much as the launch); large clouds: 256-row buckets, early switch
  const int rpl = g_fps_rpl ? g_fps_rpl : (n_pts >= (4 << 20) ? 4 : 1);
  const int k0 = g_fps_switch ? g_fps_switch : (n_pts >= (4 << 20) ? 32 : 160);
  if (rpl == 4)
    launch_bucketed<4>(r4, n_pts, order, pts, ld, dim, n_lms, k0, bucket_box, key, best, out_idx, st);
  else
    launch_bucketed<1>(r4, n_pts, order, pts, ld, dim, n_lms, k0, bucket_box, key, best, out_idx, st);
  hipLaunchKernelGGL(fps_last_kernel2, dim3(1), dim3(64), 0, st, best, n_lms, out_idx);
  return check_launch("fps_indexed");
}

}  // extern "C"
