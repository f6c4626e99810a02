// flood_fps2.hip - bucketed exact farthest-point sampling, several landmarks per launch (gfx950; dim <= 8).
//
// generate_landmarks (flooder/core.py:291-343) delegates to fpsample.bucket_fps_kdline_sampling: exact FPS, one
// landmark after the other, over kd-tree buckets.  flood_fps.hip does the same over buckets of the curve-sorted cloud
// with ONE launch per landmark - and a launch costs 4 - 5 us (start, four dependent memory round trips, drain)
// whatever little work it holds: 4.4 ms for the 1000 landmarks of a million points, 2.5x the whole coverage sweep.
//
// FPS is sequential by definition: landmark k+1 is the arg-max of the running minima AFTER landmark k has lowered
// them.  But late in the selection a new landmark only lowers its own neighbourhood, and the runners-up are far away:
//
//   Let y1 > y2 > ... be the points in descending (running minimum, lower index first) order BEFORE an update.
//   If d2(y2, y1) >= m(y2), then inserting y1 leaves m(y2) untouched while no other minimum grows, so y2 IS the next
//   landmark; by induction y_i is the (i)th next landmark as long as d2(y_i, y_l) >= m(y_i) > 0 for every l < i.
//
// One launch therefore selects a whole prefix y1 .. y_nb of the ranking (up to KMAX), all waves deciding the same
// prefix redundantly from the same data, and applies all nb updates in one pass over the touched buckets.  The result
// is the sequential selection, index for index (test_bucketed_fps_equals_brute_force); 1000 landmarks of a million
// Gaussian points take ~240 launches instead of 1000.
//
// What the ranking needs - without sorting anything: every bucket keeps its best point key AND its second best
// (k1, k2: (minimum bits << 32 | ~original index), unique per point); every wave owns 64 buckets and pushes
// w1 = its best point and w2 = the best of everything else it owns (an upper bound of all its other points) into slot
// (wave mod 256) of the NEXT launch's slot table, which keeps the top two of what it receives (two integer atomics:
// a = max, b = max of the losers).  Then every slot's `a` is a true point, every other point of the cloud is <= B =
// max over slots of b, and the points above B, in descending order, ARE the head of the ranking.
// Slot tables and the landmark counter are indexed by launch number (zeroed by the caller), so no launch ever
// clears or re-reads what a concurrent block still uses.  The host cannot know the number of launches: it enqueues
// them in rounds and reads the counter (4 bytes) between rounds - the one entry point of this library that
// synchronises its stream; surplus launches of a round see "done" and return at once.
//
// Rows: coordinates come from the curve-sorted padded rows of the PointIndex (the cloud is not copied again), the
// running minima live in their own array.  Arithmetic per point as the brute-force kernels (direct differences, one
// fma chain), box bounds by the same chain on |gap| <= |difference|: no margin anywhere.

#include "flood_common.hpp"

using namespace flooder;

namespace {

constexpr int BSLOTS = 64;    // arg-max slots per iteration of the brute-force phase
constexpr int SLOTS2 = 256;   // (a, b) slot pairs per launch of the batched phase
constexpr int KMAX = 8;       // landmarks per launch at most

typedef unsigned long long u64;

__device__ __forceinline__ u64 wave_max_key(u64 k) {
  const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
  const uint32_t mh = wave_max_u32(hi);
  const uint32_t ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((u64)mh << 32) | (u64)ml;
}

__device__ __forceinline__ u64 make_key(float m, uint32_t orig) {
  return ((u64)__float_as_uint(m) << 32) | (u64)(0xffffffffu - orig);
}

// winner of brute-force iteration `it` (every wave reduces the 64 slots redundantly, one slot per lane)
__device__ __forceinline__ uint32_t brute_winner(const u64* __restrict__ best, int it) {
  const u64 k = wave_max_key(best[(int64_t)it * BSLOTS + (threadIdx.x & 63)]);
  return 0xffffffffu - (uint32_t)k;
}

template <int DIM>
__device__ __forceinline__ float dist2(const float (&x)[DIM], const float (&c)[DIM]) {
  float d2 = 0.f;
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    const float t = x[k] - c[k];
    d2 = __builtin_fmaf(t, t, d2);
  }
  return d2;
}

template <int DIM, int DP>
__device__ __forceinline__ void row_coords(const float* __restrict__ pts_sorted, int64_t j, float (&x)[DIM]) {
  float r[DP];
  load_row<DP>(pts_sorted + j * DP, r);
#pragma unroll
  for (int k = 0; k < DIM; ++k) x[k] = r[k];
}

// ---- brute-force step over the sorted rows (first iterations: most buckets would be touched anyway)
template <int DIM>
__global__ __launch_bounds__(256) void fps2_sorted_step_kernel(const float* __restrict__ pts_sorted,
                                                               float* __restrict__ minsq, int64_t n,
                                                               const int32_t* __restrict__ order,
                                                               const float* __restrict__ pts, int ld, int it,
                                                               u64* __restrict__ best, int64_t* __restrict__ out_idx) {
  constexpr int DP = padded_dim(DIM);
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  float x[4][DIM], m0[4];
  uint32_t o[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {  // this thread's rows first: in flight while the previous winner is fetched
    const int64_t j = base + u * 256;
    const int64_t jj = j < n ? j : n - 1;
    row_coords<DIM, DP>(pts_sorted, jj, x[u]);
    m0[u] = it == 1 ? __builtin_inff() : minsq[jj];
    o[u] = (uint32_t)order[jj];
  }
  const uint32_t q = brute_winner(best, it - 1);
  float c[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) c[k] = pts[(int64_t)q * ld + k];
  if (blockIdx.x == 0 && threadIdx.x == 0) out_idx[it - 1] = (int64_t)q;
  u64 bk = 0ull;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t j = base + u * 256;
    if (j < n) {
      const float d2 = dist2<DIM>(x[u], c);
      const float m = d2 < m0[u] ? d2 : m0[u];
      if (m < m0[u] || it == 1) minsq[j] = m;
      const u64 k = make_key(m, o[u]);
      bk = k > bk ? k : bk;
    }
  }
  const u64 wk = wave_max_key(bk);
  __shared__ u64 s_k[4];
  if ((threadIdx.x & 63) == 0) s_k[threadIdx.x >> 6] = wk;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 k = s_k[0];
    for (int w = 1; w < 4; ++w) k = s_k[w] > k ? s_k[w] : k;
    atomicMax(&best[(int64_t)it * BSLOTS + (blockIdx.x % BSLOTS)], k);
  }
}

// ---- bucket b = RPL * 64 consecutive rows: box, best and second-best point key (one wave per bucket)
template <int DIM, int RPL>
__global__ __launch_bounds__(256) void fps2_bucket_init_kernel(const float* __restrict__ pts_sorted,
                                                               const float* __restrict__ minsq, int64_t n,
                                                               const int32_t* __restrict__ order, int64_t n_buckets,
                                                               float* __restrict__ box, u64* __restrict__ keys) {
  constexpr int DP = padded_dim(DIM);
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n_buckets) return;
  float lo[DIM], hi[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
  u64 b1 = 0ull, b2 = 0ull;
#pragma unroll
  for (int u = 0; u < RPL; ++u) {
    const int64_t j = b * (RPL * 64) + u * 64 + lane;
    if (j < n) {
      float x[DIM];
      row_coords<DIM, DP>(pts_sorted, j, x);
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        lo[k] = __builtin_fminf(lo[k], x[k]);
        hi[k] = __builtin_fmaxf(hi[k], x[k]);
      }
      const u64 k = make_key(minsq[j], (uint32_t)order[j]);
      if (k > b1) { b2 = b1; b1 = k; } else if (k > b2) { b2 = k; }
    }
  }
  const u64 k1 = wave_max_key(b1);
  const u64 k2 = wave_max_key(b1 == k1 ? b2 : b1);
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    lo[k] = wave_min_f32(lo[k]);
    hi[k] = wave_max_f32(hi[k]);
  }
  if (lane == 0) {
    float* bb = box + b * 2 * DP;
#pragma unroll
    for (int k = 0; k < DP; ++k) { bb[k] = k < DIM ? lo[k < DIM ? k : 0] : 0.f; bb[DP + k] = k < DIM ? hi[k < DIM ? k : 0] : 0.f; }
    keys[2 * b] = k1;
    keys[2 * b + 1] = k2;
  }
}

// ---- one batched step (launch number L >= 1; L == 0 with init_only: only the slot pushes)
// lane l of wave w owns bucket l * n_waves + w (a landmark's neighbourhood spreads over the whole chip)
template <int DIM, int RPL>
__global__ __launch_bounds__(256) void fps2_batch_step_kernel(
    const float* __restrict__ pts_sorted, float* __restrict__ minsq, int64_t n, const int32_t* __restrict__ order,
    const float* __restrict__ pts, int ld, int64_t n_buckets, const float* __restrict__ box,
    u64* __restrict__ keys, int L, int n_lms, int32_t* __restrict__ ctr, u64* __restrict__ slots,
    int64_t* __restrict__ out_idx, int init_only) {
  constexpr int DP = padded_dim(DIM);
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t b = (int64_t)lane * n_waves + w;
  const bool has = b < n_buckets;
  const int it = ctr[L];  // landmarks applied so far (written by launch L - 1; the same for every block)
  if (it >= n_lms) {      // done: surplus launch of a round
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr[L + 1] = it;
    return;
  }
  float blo[DIM], bhi[DIM];
  u64 k1 = 0ull, k2 = 0ull;
#pragma unroll
  for (int k = 0; k < DIM; ++k) { blo[k] = 0.f; bhi[k] = 0.f; }
  if (has) {
    float r0[DP], r1[DP];
    load_row<DP>(box + b * 2 * DP, r0);
    load_row<DP>(box + b * 2 * DP + DP, r1);
#pragma unroll
    for (int k = 0; k < DIM; ++k) { blo[k] = r0[k]; bhi[k] = r1[k]; }
    k1 = keys[2 * b];
    k2 = keys[2 * b + 1];
  }
  if (!init_only) {
    // ---- head of the ranking: slot winners above B = the best of everything that is not a slot winner
    const u64* sl = slots + (int64_t)L * (SLOTS2 * 2);
    u64 a[SLOTS2 / 64], bb = 0ull;
#pragma unroll
    for (int t = 0; t < SLOTS2 / 64; ++t) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(sl + 2 * (lane + 64 * t));
      a[t] = v.x;
      bb = v.y > bb ? v.y : bb;
    }
    const u64 B = wave_max_key(bb);
    u64 ck[KMAX];
    int nc = 0;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      ck[i] = 0ull;
      if (nc == i) {  // (wave-uniform)
        u64 m = 0ull;
#pragma unroll
        for (int t = 0; t < SLOTS2 / 64; ++t) m = (a[t] > B && a[t] > m) ? a[t] : m;
        const u64 wk = wave_max_key(m);
        if (wk != 0ull) {
          ck[i] = wk;
          ++nc;
#pragma unroll
          for (int t = 0; t < SLOTS2 / 64; ++t) a[t] = a[t] == wk ? 0ull : a[t];
        }
      }
    }
    // coordinates of the candidates: lane i fetches candidate i, then broadcasts
    u64 myk = 0ull;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) myk = lane == i ? ck[i] : myk;
    float myc[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) myc[k] = 0.f;
    if (lane < nc) {
      const uint32_t q = 0xffffffffu - (uint32_t)myk;
#pragma unroll
      for (int k = 0; k < DIM; ++k) myc[k] = pts[(int64_t)q * ld + k];
    }
    float cc[KMAX][DIM];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) cc[i][k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myc[k]), i));
    }
    // ---- the prefix that is provably the sequential selection
    int nb = nc > 0 ? 1 : 0;
#pragma unroll
    for (int i = 1; i < KMAX; ++i) {
      if (nb == i && i < nc) {  // (wave-uniform: every candidate before i was accepted)
        const float mi = __uint_as_float((uint32_t)(ck[i] >> 32));
        bool ok = mi > 0.f;
#pragma unroll
        for (int l = 0; l < KMAX; ++l) {
          if (l < i) ok = ok && !(dist2<DIM>(cc[i], cc[l]) < mi);
        }
        if (ok) nb = i + 1;
      }
    }
    if (nb > n_lms - it) nb = n_lms - it;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < KMAX; ++i)
        if (i < nb) out_idx[it + i] = (int64_t)(0xffffffffu - (uint32_t)ck[i]);
      ctr[L + 1] = it + nb;
    }
    // ---- which of this wave's buckets can still change?  (lower bound of d2 from a landmark to the bucket's box)
    bool touched = false;
    const float mk1 = __uint_as_float((uint32_t)(k1 >> 32));
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      if (i < nb) {
        float lb = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float g = __builtin_fmaxf(__builtin_fmaxf(blo[k] - cc[i][k], cc[i][k] - bhi[k]), 0.f);
          lb = __builtin_fmaf(g, g, lb);
        }
        touched = touched || lb < mk1;
      }
    }
    u64 tm = __ballot(touched && has);
    while (tm) {  // (wave-uniform) the whole wave updates one touched bucket at a time
      const int src = __builtin_ctzll(tm);
      tm &= tm - 1ull;
      const int64_t tb = (int64_t)src * n_waves + w;
      u64 b1 = 0ull, b2 = 0ull;
      float x[RPL][DIM], m0[RPL];
      uint32_t o[RPL];
#pragma unroll
      for (int u = 0; u < RPL; ++u) {
        const int64_t j = tb * (RPL * 64) + u * 64 + lane;
        const int64_t jj = j < n ? j : n - 1;
        row_coords<DIM, DP>(pts_sorted, jj, x[u]);
        m0[u] = minsq[jj];
        o[u] = (uint32_t)order[jj];
      }
#pragma unroll
      for (int u = 0; u < RPL; ++u) {
        const int64_t j = tb * (RPL * 64) + u * 64 + lane;
        if (j < n) {
          float m = m0[u];
#pragma unroll
          for (int i = 0; i < KMAX; ++i) {
            if (i < nb) {
              const float d2 = dist2<DIM>(x[u], cc[i]);
              m = d2 < m ? d2 : m;
            }
          }
          if (m < m0[u]) minsq[j] = m;
          const u64 k = make_key(m, o[u]);
          if (k > b1) { b2 = b1; b1 = k; } else if (k > b2) { b2 = k; }
        }
      }
      const u64 nk1 = wave_max_key(b1);
      const u64 nk2 = wave_max_key(b1 == nk1 ? b2 : b1);
      if (lane == src) {
        k1 = nk1;
        k2 = nk2;
        keys[2 * tb] = nk1;
        keys[2 * tb + 1] = nk2;
      }
    }
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctr[L + 1] = it;
  }
  // ---- this wave's best point and the best of everything else it owns -> top-2 slot of the next launch
  const u64 w1 = wave_max_key(k1);
  const u64 w2 = wave_max_key(k1 == w1 ? k2 : k1);
  if (lane == 0 && w1 != 0ull) {
    u64* nx = slots + (int64_t)(L + 1) * (SLOTS2 * 2) + 2 * (w % SLOTS2);
    const u64 old = atomicMax(&nx[0], w1);
    const u64 loser = old < w1 ? old : w1;
    const u64 v = loser > w2 ? loser : w2;
    if (v != 0ull) atomicMax(&nx[1], v);
  }
}

__global__ void fps2_start_kernel(u64* best, int64_t start) {
  best[0] = make_key(1.0f, (uint32_t)start);  // iteration 0 "winner" = the start point
}

__global__ void fps2_last_kernel(const u64* best, int n_lms, int64_t* out_idx) {
  const uint32_t q = brute_winner(best, n_lms - 1);
  if (threadIdx.x == 0) out_idx[n_lms - 1] = (int64_t)q;
}

__global__ void fps2_set_ctr_kernel(int32_t* ctr, int value) { ctr[0] = value; }

template <int DIM, int RPL>
int run_batched(const float* pts, int64_t n, int ld, const float* pts_sorted, const int32_t* order, int n_lms,
                int64_t start, int k0, int64_t* out_idx, float* minsq, float* box, u64* keys, u64* best, u64* slots,
                int32_t* ctr, int32_t* launches_out, hipStream_t st) {
  const int64_t n_buckets = (n + RPL * 64 - 1) / (RPL * 64);
  const int64_t n_waves = (n_buckets + 63) / 64;
  const unsigned grid = (unsigned)((n_waves + 3) / 4);
  const int64_t brute_blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(fps2_start_kernel, dim3(1), dim3(1), 0, st, best, start);
  const int k_brute = k0 < n_lms ? k0 : n_lms;
  for (int it = 1; it < k_brute; ++it)
    hipLaunchKernelGGL((fps2_sorted_step_kernel<DIM>), dim3((unsigned)brute_blocks), dim3(256), 0, st, pts_sorted, minsq,
                       n, order, pts, ld, it, best, out_idx);
  int launches = k_brute - 1;
  if (k_brute == n_lms) {
    hipLaunchKernelGGL(fps2_last_kernel, dim3(1), dim3(64), 0, st, best, n_lms, out_idx);
    if (launches_out) *launches_out = launches;
    return check_launch("fps_batched (brute only)");
  }
  // landmarks 0 .. k_brute - 2 are applied; the batched phase re-selects landmark k_brute - 1 (the same arg-max)
  hipLaunchKernelGGL((fps2_bucket_init_kernel<DIM, RPL>), dim3((unsigned)((n_buckets + 3) / 4)), dim3(256), 0, st,
                     pts_sorted, minsq, n, order, n_buckets, box, keys);
  hipLaunchKernelGGL(fps2_set_ctr_kernel, dim3(1), dim3(1), 0, st, ctr, k_brute - 1);
  hipLaunchKernelGGL((fps2_batch_step_kernel<DIM, RPL>), dim3(grid), dim3(256), 0, st, pts_sorted, minsq, n, order, pts,
                     ld, n_buckets, box, keys, 0, n_lms, ctr, slots, out_idx, 1);
  int L = 1, done = k_brute - 1;
  int round = 64;
  while (done < n_lms) {
    const int remaining = n_lms - done;
    if (round > remaining) round = remaining;  // (a launch selects at least one landmark)
    for (int i = 0; i < round; ++i, ++L)
      hipLaunchKernelGGL((fps2_batch_step_kernel<DIM, RPL>), dim3(grid), dim3(256), 0, st, pts_sorted, minsq, n, order,
                         pts, ld, n_buckets, box, keys, L, n_lms, ctr, slots, out_idx, 0);
    int32_t now = 0;
    if (hipMemcpyAsync(&now, ctr + L, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return fail(FLOODER_E_LAUNCH, "flooder_fps_batched_f32: reading the landmark counter failed");
    if (now <= done) return fail(FLOODER_E_LAUNCH, "flooder_fps_batched_f32: no progress (internal error)");
    // next round: what is left at the rate seen so far, plus a margin (surplus launches return at once)
    const double per_launch = (double)(now - (k_brute - 1)) / (double)(L - 1);
    done = now;
    launches += round;
    round = (int)((double)(n_lms - done) / per_launch * 1.2) + 4;
  }
  if (launches_out) *launches_out = launches;
  return check_launch("fps_batched");
}

template <int DIM>
struct FpsBatchedOp {
  static int run(const float* pts, int64_t n, int ld, const float* pts_sorted, const int32_t* order, int n_lms,
                 int64_t start, int64_t* out_idx, float* minsq, float* box, u64* keys, u64* best, u64* slots,
                 int32_t* ctr, int32_t* launches_out, hipStream_t st) {
    // small clouds: 64-row buckets (more waves to spread a landmark's neighbourhood over) and a late switch (a
    // brute step over an L2-resident cloud costs about as much as a launch); large clouds: 256-row buckets
    const int rpl = g_fps_rpl ? g_fps_rpl : (n >= (4 << 20) ? 4 : 1);
    // (measured: 1 M / 1 k 2.92 ms at 96, 3.15 at 32, 3.07 at 256; 16 M / 4 k 19.3 ms at 8, 20.2 at 32, 29.9 at 256)
    int k0 = g_fps_switch ? g_fps_switch : (n >= (4 << 20) ? 8 : 96);
    if (k0 < 2) k0 = 2;  // (the start point is applied by a brute-force step)
    if (rpl == 4)
      return run_batched<DIM, 4>(pts, n, ld, pts_sorted, order, n_lms, start, k0, out_idx, minsq, box, keys, best, slots,
                                 ctr, launches_out, st);
    return run_batched<DIM, 1>(pts, n, ld, pts_sorted, order, n_lms, start, k0, out_idx, minsq, box, keys, best, slots,
                               ctr, launches_out, st);
  }
};

}  // namespace

extern "C" {

int64_t flooder_fps_batched_slot_words(int n_lms) { return (int64_t)(n_lms + 4) * (SLOTS2 * 2); }

int flooder_fps_batched_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* pts_sorted,
                            const int32_t* order, int n_lms, int64_t start, int64_t* out_idx, float* minsq,
                            float* bucket_box, uint64_t* bucket_keys, uint64_t* work_best, uint64_t* work_slots,
                            int32_t* work_ctr, int32_t* launches_out, void* stream) {
  if (!pts || !pts_sorted || !order || !out_idx || !minsq || !bucket_box || !bucket_keys || !work_best ||
      !work_slots || !work_ctr || n_pts < 1 || n_lms < 1 || n_lms > n_pts || start < 0 || start >= n_pts || ld < dim ||
      dim < 1 || dim > FLOODER_MAX_DIM || n_pts > 0xfffffffeLL)
    return fail(FLOODER_E_ARG, "flooder_fps_batched_f32: bad argument");
  return dispatch_dim<FpsBatchedOp>(dim, pts, n_pts, ld, pts_sorted, order, n_lms, start, out_idx, minsq, bucket_box,
                                    reinterpret_cast<u64*>(bucket_keys), reinterpret_cast<u64*>(work_best),
                                    reinterpret_cast<u64*>(work_slots), work_ctr, launches_out, (hipStream_t)stream);
}

}  // extern "C"
