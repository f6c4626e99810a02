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
// Gaussian points take ~175 batched launches (after 95 brute-force steps) instead of 1000, 4000 of 16 M ~710.
//
// What the ranking needs - without sorting anything: every bucket keeps its best point key, its second best (k1, k2:
// (minimum bits << 32 | ~original index), unique per point) and the coordinates of its best point; every workgroup
// (4 waves x 64 buckets, consecutive buckets in different workgroups) ends a launch by writing ONE record for the
// next launch - plain stores, no atomics: its best point b1 with coordinates, hb of THAT point's bucket (below), bo =
// the best of all its other points (an upper bound) and the number of landmarks so far.  Every wave of the next launch
// reads all records (<= 256) with its first loads, together with its own buckets - one memory round trip; the block
// winners above B = max bo are the head of the ranking (compacted through LDS, one per lane, each lane counts the
// candidates above its own: no chain of reductions), EXCEPT for the points hidden in a
// winner's own bucket (in a dense cloud the runner-up of the arg-max is its neighbour).  Those are settled at
// acceptance: every bucket also keeps hb = max over its points of min(m(x), d2(x, best point)) - what its best minimum
// becomes the moment its best point is a landmark (computed with the update's own d2 chain while the rows are in
// registers; later landmarks only lower it) - and candidate y_i is accepted only if its minimum exceeds hb of every
// accepted y_c.  Records are indexed by launch number, so no launch re-reads what a concurrent block still writes.
// The host cannot know the number of launches, and a surplus launch costs a launch (the first version enqueued rounds
// sized by the rate seen so far: 3331 launches for the 709 that 16 M / 4 k needs, because the rate grows 4x during a
// selection).  Every launch now stores (launches done, landmarks so far) into a pinned host word; the host keeps 24
// launches in flight beyond the last one it has seen complete and stops when the word says "all selected" - it never
// drains the stream (option "fps_rounds": doubling rounds with a counter read-back, the fallback without pinned
// memory).
//
// Rows: coordinates come from the curve-sorted padded rows of the PointIndex (the cloud is not copied again), the
// running minima live in their own array.  Arithmetic per point as the brute-force kernels (direct differences, one
// fma chain), box bounds by the same chain on |gap| <= |difference|: no margin anywhere.

#include "flood_common.hpp"

#include <mutex>

using namespace flooder;

namespace {

constexpr int BSLOTS = 64;    // arg-max slots per iteration of the brute-force phase
#ifndef FLOODER_FPS_KMAX
#define FLOODER_FPS_KMAX 8
#endif
constexpr int KMAX = FLOODER_FPS_KMAX;  // landmarks per launch at most (buckets of 64 rows: clouds below 4 M points)
#ifndef FLOODER_FPS_KMAX_LARGE
#define FLOODER_FPS_KMAX_LARGE 32
#endif
// ... and with buckets of 256 rows (RPL = 4, larger clouds).  There two batches in three used to end at the cap of 8
// while a launch takes 12 us whatever it selects (tools/fps_batches.py), so the cap is 32 and the loops over the
// accepted landmarks run over the lanes that hold them instead of unrolled register arrays (a cap that costs no
// registers): 16 M / 4 k 12.9 -> 11.3 ms.  The small clouds keep the unrolled form - the same loops are 5 - 10 %
// slower there (1 M / 1 k 2.31 -> 2.38 ms, 2 M in 6-D 9.6 -> 10.5) and only a third of their batches reach the cap.
constexpr int KMAX_LARGE = FLOODER_FPS_KMAX_LARGE;
template <int RPL> struct BatchCap { static constexpr bool DYN = RPL > 1; static constexpr int CAP = DYN ? KMAX_LARGE : KMAX; };
static_assert(KMAX >= 1 && KMAX <= 64 && KMAX_LARGE >= 1 && KMAX_LARGE <= 64, "one accepted candidate per lane");

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

// ---- bucket b = RPL * 64 consecutive rows: box, best and second-best point key, coordinates of the best point
template <int DIM, int RPL>
__global__ __launch_bounds__(256) void fps2_bucket_init_kernel(const float* __restrict__ pts_sorted,
                                                               const float* __restrict__ minsq, int64_t n,
                                                               const int32_t* __restrict__ order, int64_t n_buckets,
                                                               float* __restrict__ box, u64* __restrict__ keys,
                                                               float* __restrict__ bcoord) {
  constexpr int DP = padded_dim(DIM);
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n_buckets) return;
  float lo[DIM], hi[DIM], bx[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); bx[k] = 0.f; }
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
      if (k > b1) {
        b2 = b1;
        b1 = k;
#pragma unroll
        for (int kk = 0; kk < DIM; ++kk) bx[kk] = x[kk];
      } else if (k > b2) {
        b2 = k;
      }
    }
  }
  const u64 k1 = wave_max_key(b1);
  const u64 k2 = wave_max_key(b1 == k1 ? b2 : b1);
  const int bl = __builtin_ctzll(__ballot(b1 == k1));
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    lo[k] = wave_min_f32(lo[k]);
    hi[k] = wave_max_f32(hi[k]);
    bx[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx[k]), bl));
  }
  // what the bucket's best minimum WOULD be once its best point is a landmark (the update's own d2 chain): the bound
  // of the points hidden behind that point
  float rad = 0.f;
#pragma unroll
  for (int u = 0; u < RPL; ++u) {
    const int64_t j = b * (RPL * 64) + u * 64 + lane;
    if (j < n) {
      float x[DIM];
      row_coords<DIM, DP>(pts_sorted, j, x);
      rad = __builtin_fmaxf(rad, __builtin_fminf(minsq[j], dist2<DIM>(x, bx)));
    }
  }
  rad = wave_max_f32(rad);
  if (lane == 0) {
    float* bb = box + b * 2 * DP;
    float* bc = bcoord + b * DP;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      bb[k] = k < DIM ? lo[k < DIM ? k : 0] : 0.f;
      bb[DP + k] = k < DIM ? hi[k < DIM ? k : 0] : 0.f;
      bc[k] = k < DIM ? bx[k < DIM ? k : 0] : 0.f;
    }
    keys[3 * b] = k1;
    keys[3 * b + 1] = k2;
    keys[3 * b + 2] = (u64)__float_as_uint(rad);
  }
}

// Block record of a launch (plain stores by the block that owns it, read by every wave of the next launch):
//   b1 best point of the block's buckets | hb = best minimum of the other points of THAT point's bucket once b1 is a
//   landmark: max over the bucket of min(m(x), d2(x, b1)) | bo best of every other point of the block (an upper
//   bound) | landmarks applied after the launch | m2 = the minimum of the second-best point of b1's bucket (what bounds
//   the points behind b1 while b1 is NOT a landmark) | coordinates of the b1 point.  32-bit words:
// (sections of 16 bytes or whole padded rows: a record is read with 16-byte loads)
template <int DP>
struct Rec {
  static constexpr int DPR = DP < 4 ? 4 : DP;  // row section (floats)
  static constexpr int B1 = 0, HB = 2, BO = 4, IT = 6, M2 = 7, T0 = 8, C = 12, WORDS = 12 + DPR;
};
constexpr int NREC = 4;  // records per lane: at most 256 blocks

__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((u64)hi << 32) | (u64)lo;
}

// ---- one batched step (launch number L >= 1; L == 0 with init_only: only the block records)
// Progress word in pinned host memory (one 8-byte store per launch, fire and forget): tag << 56 | launches done (24
// bits, wraps) << 32 | landmarks selected.  The host reads it while it enqueues - it never drains the stream.
__device__ __forceinline__ void report_progress(u64* progress, u64 tag, int launches_done, int landmarks) {
  if (progress)
    __hip_atomic_store(progress, (tag << 56) | ((u64)((uint32_t)launches_done & 0xffffffu) << 32) | (u64)(uint32_t)landmarks,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// lane l of wave w owns bucket l * n_waves + w (a landmark's neighbourhood spreads over the whole chip)
template <int DIM, int RPL>
__global__ __launch_bounds__(256) void fps2_batch_step_kernel(
    const float* __restrict__ pts_sorted, float* __restrict__ minsq, int64_t n, const int32_t* __restrict__ order,
    int64_t n_buckets, const float* __restrict__ box, u64* __restrict__ keys, float* __restrict__ bcoord, int L,
    int n_lms, int32_t* __restrict__ ctr, uint32_t* __restrict__ rec, int64_t* __restrict__ out_idx, int flags,
    u64* __restrict__ progress, u64 progress_tag) {
  const bool init_only = (flags & 1) != 0;  // (bit 1: test hook, see below)
  constexpr int DP = padded_dim(DIM);
  typedef Rec<DP> RC;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  // consecutive buckets (neighbours in space) belong to different workgroups: the runner-up next to the arg-max then
  // is another block's winner - a candidate - and not the `bo` that closes every batch
  const int64_t w = (int64_t)wv * gridDim.x + blockIdx.x;
  const int64_t b = (int64_t)lane * n_waves + w;
  const bool has = b < n_buckets;
  const uint32_t t_start = (uint32_t)wall_clock64();  // (diagnostic: record words 8, 9 of block 0; 100 MHz)
  // ---- every load of the launch's head goes out together: this wave's buckets and ALL block records of the launch
  // before (<= 256, NREC per lane), which also carry the number of landmarks applied so far
  float blo[DIM], bhi[DIM], bc[DIM], brad = 0.f;
  u64 k1 = 0ull, k2 = 0ull;
  int s_why = 0;
#pragma unroll
  for (int k = 0; k < DIM; ++k) { blo[k] = 0.f; bhi[k] = 0.f; bc[k] = 0.f; }
  if (has) {
    float r0[DP], r1[DP], r2[DP];
    load_row<DP>(box + b * 2 * DP, r0);
    load_row<DP>(box + b * 2 * DP + DP, r1);
    load_row<DP>(bcoord + b * DP, r2);
#pragma unroll
    for (int k = 0; k < DIM; ++k) { blo[k] = r0[k]; bhi[k] = r1[k]; bc[k] = r2[k]; }
    k1 = keys[3 * b];
    k2 = keys[3 * b + 1];
    brad = __uint_as_float((uint32_t)keys[3 * b + 2]);
  }
  u64 a[NREC];
  float ac[NREC][DIM], ahb[NREC], am2[NREC];
  u64 bo = 0ull;
  int it_rec = 0;
#pragma unroll
  for (int t = 0; t < NREC; ++t) {
    const int idx = lane + 64 * t;
    a[t] = 0ull;
    ahb[t] = 0.f;
    am2[t] = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) ac[t][k] = 0.f;
    if (!init_only && idx < (int)gridDim.x) {
      const uint32_t* r = rec + ((int64_t)L * gridDim.x + idx) * RC::WORDS;
      const uint4 h0 = *reinterpret_cast<const uint4*>(r);            // b1, hb
      const uint4 h1 = *reinterpret_cast<const uint4*>(r + RC::BO);   // bo, landmarks so far
      a[t] = ((u64)h0.y << 32) | (u64)h0.x;
      ahb[t] = __uint_as_float(h0.z);
      am2[t] = __uint_as_float(h1.w);
      const u64 o = ((u64)h1.y << 32) | (u64)h1.x;
      bo = o > bo ? o : bo;
      if (t == 0) it_rec = (int)h1.z;
      float rc_[RC::DPR];
      load_row<RC::DPR>(reinterpret_cast<const float*>(r + RC::C), rc_);
#pragma unroll
      for (int k = 0; k < DIM; ++k) ac[t][k] = rc_[k];
    }
  }
  // landmarks applied so far (the same for every block: block 0's record of the launch before)
  const int it = init_only ? ctr[0] : __builtin_amdgcn_readfirstlane(it_rec);
  if (it >= n_lms) {  // done: a surplus launch.  The next one still reads "landmarks so far" from block 0's record
    if (threadIdx.x == 0) {
      rec[((int64_t)(L + 1) * gridDim.x + blockIdx.x) * RC::WORDS + RC::IT] = (uint32_t)it;
      if (blockIdx.x == 0) {
        ctr[L + 1] = it;
        report_progress(progress, progress_tag, L + 1, it);
      }
    }
    return;
  }
  int nb = 0;
  if (!init_only) {
    // ---- head of the ranking: block winners above B = the best of every point that is neither a block winner nor
    // hidden behind one (the points hidden behind a winner are accounted for at acceptance).  All candidates are
    // ranked at once: compacted through LDS (one per lane), each lane counts the candidates above its own.
    __shared__ u64 s_ckey[4][64];
    __shared__ float s_cdat[4][64][DIM + 2];
    u64 B = wave_max_key(bo);
    int total = 0;
#pragma unroll
    for (int t = 0; t < NREC; ++t) total += __builtin_popcountll(__ballot(a[t] > B));
    if (total > 64 || (flags & 2)) {  // more candidates than lanes (or option "fps_lane_best", a test hook): every lane keeps its best, the others close the ranking like any
                       // other point (the arg-max is some lane's best)
      u64 best = 0ull, oth = 0ull;
#pragma unroll
      for (int t = 0; t < NREC; ++t) best = a[t] > best ? a[t] : best;
#pragma unroll
      for (int t = 0; t < NREC; ++t) {
        if (a[t] != best) {
          oth = a[t] > oth ? a[t] : oth;
          a[t] = 0ull;
        }
      }
      const u64 d = wave_max_key(oth);
      B = d > B ? d : B;
    }
    int n_c = 0;
#pragma unroll
    for (int t = 0; t < NREC; ++t) {
      const bool is = a[t] > B;
      const u64 mask = __ballot(is);
      if (mask) {  // (wave-uniform)
        const int slot = n_c + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (is) {
          s_ckey[wv][slot] = a[t];
          s_cdat[wv][slot][DIM] = ahb[t];
          s_cdat[wv][slot][DIM + 1] = am2[t];
#pragma unroll
          for (int k = 0; k < DIM; ++k) s_cdat[wv][slot][k] = ac[t][k];
        }
        n_c += __builtin_popcountll(mask);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    u64 key = 0ull;
    float ci[DIM], hbi = 0.f, m2i = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) ci[k] = 0.f;
    if (lane < n_c) {
      key = s_ckey[wv][lane];
      hbi = s_cdat[wv][lane][DIM];
      m2i = s_cdat[wv][lane][DIM + 1];
#pragma unroll
      for (int k = 0; k < DIM; ++k) ci[k] = s_cdat[wv][lane][k];
    }
    const bool valid = key != 0ull;
    int rank = 0;
    for (int j = 0; j < n_c; ++j) rank += readlane_u64(key, j) > key ? 1 : 0;
    const float mi = __uint_as_float((uint32_t)(key >> 32));
    // ---- acceptance, candidate by candidate in ranking order; every lane tracks its own candidate against the
    // accepted ones (how far has its minimum fallen? is it below the points hidden behind one of them?), so a step
    // costs a few cross-lane reads, not a reduction.  A candidate whose minimum an accepted landmark has LOWERED is
    // not the next landmark at its old rank - but it does not close the batch either (that closed 56 - 73 % of the
    // batches): it is skipped, and what it and the points of its bucket can still amount to - its lowered key, the
    // minimum m2 of its bucket's second-best point - bounds every later acceptance (`low`).  The next candidate in
    // the ranking with an untouched minimum above that bound, above B and above the hidden points is the arg-max of
    // the sequential selection: unprocessed candidates and the points behind them are below it by the ranking,
    // skipped ones by `low`, everything else by B / hmax.
    u64 acc_key = 0ull;    // lane i: the i-th accepted candidate
    float acc_c[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) acc_c[k] = 0.f;
    float mlow = mi;       // this lane's candidate: its minimum after the landmarks accepted so far
    float hmax = 0.f;      // largest bound of the points hidden in an accepted candidate's own bucket
    u64 low = 0ull;        // bound of the skipped candidates and the points behind them
    int why = 4;           // (diagnostic, record word 3 of block 0) what closed the batch: 1 nothing above B, 2 a
                           // skipped candidate (or the points behind it) may come first, 3 hidden points, 4 KMAX / the end
    constexpr bool DYN = BatchCap<RPL>::DYN;
    constexpr int CAP = BatchCap<RPL>::CAP;
    const int nb_max = CAP < n_lms - it ? CAP : n_lms - it;
    for (int j = 0; j < 64; ++j) {   // (wave-uniform)
      if (nb >= nb_max) break;
      const u64 sm = __ballot(valid && rank == j);
      if (sm == 0ull) {
        why = 1;
        break;
      }
      const int src = __builtin_ctzll(sm);
      const u64 ksrc = readlane_u64(key, src);
      if (j > 0) {
        const float msrc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mi), src));
        const float lsrc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mlow), src));
        if (lsrc < msrc) {   // lowered: skipped
          const float m2s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m2i), src));
          const u64 lk = ((u64)__float_as_uint(lsrc) << 32) | (ksrc & 0xffffffffull);
          const u64 bk = ((u64)__float_as_uint(m2s) << 32) | 0xffffffffull;
          low = lk > low ? lk : low;
          low = bk > low ? bk : low;
          continue;
        }
        if (!(msrc > 0.f && msrc > hmax)) {
          why = 3;
          break;
        }
        if (!(ksrc > low)) {
          why = 2;
          break;
        }
      }
      float cj[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) cj[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ci[k]), src));
      const float hbj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hbi), src));
      if (lane == nb) {
        acc_key = ksrc;
#pragma unroll
        for (int k = 0; k < DIM; ++k) acc_c[k] = cj[k];
      }
      ++nb;
      if (valid && rank > j) mlow = __builtin_fminf(mlow, dist2<DIM>(ci, cj));
      hmax = __builtin_fmaxf(hmax, hbj);
    }
    // (lane i holds the i-th accepted candidate.  DYN: the loops below run over the nb accepted ones with their
    // coordinates read out of that lane; else they are unrolled over register copies)
    float cc[DYN ? 1 : KMAX][DIM];   // coordinates of the accepted candidates
    if constexpr (!DYN) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) cc[j][k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc_c[k]), j));
      }
    }
    s_why = why;
    if (nb > n_lms - it) nb = n_lms - it;
    if (blockIdx.x == 0 && wv == 0) {
      if (lane < nb) out_idx[it + lane] = (int64_t)(0xffffffffu - (uint32_t)acc_key);
      if (lane == 0) {
        ctr[L + 1] = it + nb;
        report_progress(progress, progress_tag, L + 1, it + nb);
      }
    }
    // ---- which of this wave's buckets can still change?  (lower bound of d2 from a landmark to the bucket's box)
    bool touched = false;
    const float mk1 = __uint_as_float((uint32_t)(k1 >> 32));
    if constexpr (DYN) {
      for (int i = 0; i < nb; ++i) {   // (wave-uniform)
        float lb = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc_c[k]), i));
          const float g = __builtin_fmaxf(__builtin_fmaxf(blo[k] - c, c - bhi[k]), 0.f);
          lb = __builtin_fmaf(g, g, lb);
        }
        touched = touched || lb < mk1;
      }
    } else {
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
    }
    u64 tm = __ballot(touched && has);
    constexpr int TB = 4;  // touched buckets in flight: their rows are fetched together, one memory round trip
    while (tm) {           // (wave-uniform) the whole wave updates its touched buckets, TB at a time
      int srcs[TB];
#pragma unroll
      for (int g = 0; g < TB; ++g) {
        srcs[g] = -1;
        if (tm) {
          srcs[g] = __builtin_ctzll(tm);
          tm &= tm - 1ull;
        }
      }
      float x[TB][RPL][DIM], m0[TB][RPL];
      uint32_t o[TB][RPL];
#pragma unroll
      for (int g = 0; g < TB; ++g) {
        if (srcs[g] >= 0) {
          const int64_t tb = (int64_t)srcs[g] * n_waves + w;
#pragma unroll
          for (int u = 0; u < RPL; ++u) {
            const int64_t j = tb * (RPL * 64) + u * 64 + lane;
            const int64_t jj = j < n ? j : n - 1;
            row_coords<DIM, DP>(pts_sorted, jj, x[g][u]);
            m0[g][u] = minsq[jj];
            o[g][u] = (uint32_t)order[jj];
          }
        }
      }
#pragma unroll
      for (int g = 0; g < TB; ++g) {
        if (srcs[g] >= 0) {
          const int src = srcs[g];
          const int64_t tb = (int64_t)src * n_waves + w;
          u64 b1 = 0ull, b2 = 0ull;
          float bx[DIM];
#pragma unroll
          for (int k = 0; k < DIM; ++k) bx[k] = 0.f;
          float mnew[RPL];
#pragma unroll
          for (int u = 0; u < RPL; ++u) mnew[u] = m0[g][u];
          if constexpr (DYN) {
            for (int i = 0; i < nb; ++i) {   // (wave-uniform; rows past the end hold a copy of the last row: harmless)
              float c[DIM];
#pragma unroll
              for (int k = 0; k < DIM; ++k) c[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc_c[k]), i));
#pragma unroll
              for (int u = 0; u < RPL; ++u) {
                const float d2 = dist2<DIM>(x[g][u], c);
                mnew[u] = d2 < mnew[u] ? d2 : mnew[u];
              }
            }
          } else {
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
#pragma unroll
              for (int i = 0; i < KMAX; ++i) {
                if (i < nb) {
                  const float d2 = dist2<DIM>(x[g][u], cc[i]);
                  mnew[u] = d2 < mnew[u] ? d2 : mnew[u];
                }
              }
            }
          }
#pragma unroll
          for (int u = 0; u < RPL; ++u) {
            const int64_t j = tb * (RPL * 64) + u * 64 + lane;
            if (j < n) {
              const float m = mnew[u];
              if (m < m0[g][u]) minsq[j] = m;
              m0[g][u] = m;
              const u64 k = make_key(m, o[g][u]);
              if (k > b1) {
                b2 = b1;
                b1 = k;
#pragma unroll
                for (int kk = 0; kk < DIM; ++kk) bx[kk] = x[g][u][kk];
              } else if (k > b2) {
                b2 = k;
              }
            }
          }
          const u64 nk1 = wave_max_key(b1);
          const u64 nk2 = wave_max_key(b1 == nk1 ? b2 : b1);
          const int bl = __builtin_ctzll(__ballot(b1 == nk1));
          float nbx[DIM];
#pragma unroll
          for (int k = 0; k < DIM; ++k) nbx[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx[k]), bl));
          float rad = 0.f;   // the bucket's best minimum once its new best point is a landmark (rows still in registers)
#pragma unroll
          for (int u = 0; u < RPL; ++u) {
            const int64_t j = tb * (RPL * 64) + u * 64 + lane;
            if (j < n) rad = __builtin_fmaxf(rad, __builtin_fminf(m0[g][u], dist2<DIM>(x[g][u], nbx)));
          }
          rad = wave_max_f32(rad);
          if (lane == src) {
            k1 = nk1;
            k2 = nk2;
            brad = rad;
            keys[3 * tb] = nk1;
            keys[3 * tb + 1] = nk2;
            keys[3 * tb + 2] = (u64)__float_as_uint(rad);
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              bc[k] = nbx[k];
              bcoord[tb * DP + k] = nbx[k];
            }
          }
        }
      }
    }
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctr[L + 1] = it;
    report_progress(progress, progress_tag, L + 1, it);
  }
  // ---- this block's record for the next launch
  __shared__ u64 s_w1[4], s_oth[4];
  __shared__ float s_hb[4], s_m2[4];
  __shared__ float s_c[4][DIM];
  {
    const u64 w1 = wave_max_key(k1);
    const int wl = w1 != 0ull ? __builtin_ctzll(__ballot(k1 == w1 && has)) : 0;
    const u64 oth = wave_max_key(lane == wl ? 0ull : k1);  // (the other buckets' second-best points are below their best)
    const float m2 = __uint_as_float((uint32_t)(readlane_u64(k2, wl) >> 32));
    const float rd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(brad), wl));
    if (lane == 0) { s_w1[wv] = w1; s_oth[wv] = oth; s_hb[wv] = rd < m2 ? rd : m2; s_m2[wv] = m2; }  // (rd <= m2 already)
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bc[k]), wl));
      if (lane == 0) s_c[wv][k] = c;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int wb = 0;
    for (int q = 1; q < 4; ++q) wb = s_w1[q] > s_w1[wb] ? q : wb;
    u64 bo = 0ull;
    for (int q = 0; q < 4; ++q) {
      if (q != wb && s_w1[q] > bo) bo = s_w1[q];
      if (s_oth[q] > bo) bo = s_oth[q];
    }
    uint32_t* r = rec + ((int64_t)(L + 1) * gridDim.x + blockIdx.x) * RC::WORDS;
    *reinterpret_cast<u64*>(r + RC::B1) = s_w1[wb];
    r[RC::HB] = __float_as_uint(s_hb[wb]);
    r[RC::HB + 1] = (uint32_t)s_why;
    r[RC::IT] = (uint32_t)(it + nb);
    r[RC::M2] = __float_as_uint(s_m2[wb]);
    r[RC::T0] = t_start;
    r[RC::T0 + 1] = (uint32_t)wall_clock64();
    *reinterpret_cast<u64*>(r + RC::BO) = bo;
    for (int k = 0; k < DIM; ++k) r[RC::C + k] = __float_as_uint(s_c[wb][k]);
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

// pinned progress words: eight per device, picked by the stream (allocated at first use, kept for the life of the
// process).  A call returns with surplus launches of its own still in flight; their late stores carry the old tag and
// land in the word of THEIR stream, so a call on another stream is not disturbed (two streams that share a word are
// still safe: a lost store is noticed when the stream goes idle and the device counter is read back instead)
struct ProgressSlot {
  std::mutex busy;
  u64* host = nullptr;
  u64* dev = nullptr;
  unsigned calls = 0;
  bool tried = false;
};
inline ProgressSlot* progress_slot(int device, hipStream_t stream) {
  static ProgressSlot slots[64 * 8];
  static std::mutex init;
  if (device < 0 || device >= 64 || g_fps_rounds) return nullptr;
  ProgressSlot& s = slots[device * 8 + (int)((reinterpret_cast<uintptr_t>(stream) >> 6) & 7u)];
  std::lock_guard<std::mutex> guard(init);
  if (!s.tried) {
    s.tried = true;
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
        s.host = reinterpret_cast<u64*>(h);
        s.dev = reinterpret_cast<u64*>(d);
      } else {
        hipHostFree(h);
      }
    }
    (void)hipGetLastError();
  }
  return s.host ? &s : nullptr;
}

template <int RPL>
int64_t batched_blocks(int64_t n) {
  const int64_t n_buckets = (n + RPL * 64 - 1) / (RPL * 64);
  const int64_t n_waves = (n_buckets + 63) / 64;
  return (n_waves + 3) / 4;
}

inline int batched_rpl(int64_t n) { return g_fps_rpl ? g_fps_rpl : (n >= (4 << 20) ? 4 : 1); }

template <int DIM, int RPL>
int run_batched(const float* pts, int64_t n, int ld, const float* pts_sorted, const int32_t* order, int n_lms,
                int64_t start, int k0, int64_t* out_idx, float* minsq, float* box, u64* keys, float* bcoord, u64* best,
                uint32_t* rec, int32_t* ctr, int32_t* launches_out, hipStream_t st) {
  const int64_t n_buckets = (n + RPL * 64 - 1) / (RPL * 64);
  const unsigned grid = (unsigned)batched_blocks<RPL>(n);
  if (grid > 64 * NREC) return fail(FLOODER_E_ARG, "flooder_fps_batched_f32: cloud too large (use flooder_fps_indexed_f32)");
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
                     pts_sorted, minsq, n, order, n_buckets, box, keys, bcoord);
  hipLaunchKernelGGL(fps2_set_ctr_kernel, dim3(1), dim3(1), 0, st, ctr, k_brute - 1);
  // The host cannot know how many launches the selection takes, and a surplus launch costs a launch.  Every launch
  // stores (launches done, landmarks selected) into a pinned host word; the host keeps a bounded number of launches
  // in flight beyond the last one it has seen complete and stops enqueuing when the word says "all selected".  The
  // stream is never drained; without pinned memory: rounds of launches with a counter read-back between them.
  int device = 0;
  hipGetDevice(&device);
  ProgressSlot* slot = progress_slot(device, st);
  if (slot) {
    std::lock_guard<std::mutex> guard(slot->busy);
    const u64 tag = (u64)(++slot->calls % 255u) + 1ull;   // 1 .. 255: never the tag of the freshly zeroed word
    volatile u64* word = slot->host;
    *word = 0ull;   // (a stale store of an earlier call carries that call's tag)
    hipLaunchKernelGGL((fps2_batch_step_kernel<DIM, RPL>), dim3(grid), dim3(256), 0, st, pts_sorted, minsq, n, order,
                       n_buckets, box, keys, bcoord, 0, n_lms, ctr, rec, out_idx, 1, slot->dev, tag);
    const int flags = g_fps_lane_best ? 2 : 0;
    constexpr int IN_FLIGHT = 24;
    int enq = 1;                       // launches enqueued (launch numbers 0 .. enq - 1)
    int seen_l = 0, seen_it = k_brute - 1;
    long spins = 0;
    for (;;) {
      const u64 v = *word;
      if ((v >> 56) == tag) {
        const int lw = (int)((v >> 32) & 0xffffffu);
        seen_l = enq - (int)(((uint32_t)enq - (uint32_t)lw) & 0xffffffu);
        seen_it = (int)(uint32_t)v;
      }
      if (seen_it >= n_lms) break;
      // never past what the remaining landmarks can need (a launch selects at least one): bounds the record array
      int room = IN_FLIGHT - (enq - seen_l);
      const int need = seen_l + (n_lms - seen_it) - enq;
      if (room > need) room = need;
      if (room > 0) {
        for (int i = 0; i < room; ++i, ++enq)
          hipLaunchKernelGGL((fps2_batch_step_kernel<DIM, RPL>), dim3(grid), dim3(256), 0, st, pts_sorted, minsq, n,
                             order, n_buckets, box, keys, bcoord, enq, n_lms, ctr, rec, out_idx, flags, slot->dev, tag);
        if (hipGetLastError() != hipSuccess)   // (a launch that failed would never report: the wait below must not start)
          return fail(FLOODER_E_LAUNCH, "flooder_fps_batched_f32: a launch of the selection failed");
        spins = 0;
      } else if (++spins > (1L << 22)) {   // nothing reported for a long time: is the stream still alive?
        const hipError_t q = hipStreamQuery(st);
        if (q != hipSuccess && q != hipErrorNotReady)
          return fail(FLOODER_E_LAUNCH, "flooder_fps_batched_f32: the stream failed during the selection");
        if (q == hipSuccess) {
          // every enqueued launch has run.  If the progress word is behind (a store of this call overwritten by a late
          // store of an earlier call on another stream, or no store at all) the device counter is the truth: read it
          // back, as the path without pinned memory does
          const u64 v2 = *word;
          if (!((v2 >> 56) == tag && (int)(uint32_t)v2 > seen_it)) {
            int32_t now = 0;
            if (hipMemcpyAsync(&now, ctr + enq, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
              return fail(FLOODER_E_LAUNCH, "flooder_fps_batched_f32: reading the landmark counter failed");
            if (now <= seen_it && now < n_lms)
              return fail(FLOODER_E_LAUNCH, "flooder_fps_batched_f32: no progress (internal error)");
            seen_it = now;
            seen_l = enq;
          }
        }
        spins = 0;
      }
    }
    launches += enq - 1;
    if (launches_out) *launches_out = launches;
    return check_launch("fps_batched");
  }
  hipLaunchKernelGGL((fps2_batch_step_kernel<DIM, RPL>), dim3(grid), dim3(256), 0, st, pts_sorted, minsq, n, order,
                     n_buckets, box, keys, bcoord, 0, n_lms, ctr, rec, out_idx, 1, (u64*)nullptr, 0ull);
  int L = 1, done = k_brute - 1;
  int round = 32, prev_round = 16;
  while (done < n_lms) {
    const int remaining = n_lms - done;
    if (round > 2 * prev_round) round = 2 * prev_round;  // (the rate grows as the landmarks get denser)
    if (round > remaining) round = remaining;            // (a launch selects at least one landmark)
    for (int i = 0; i < round; ++i, ++L)
      hipLaunchKernelGGL((fps2_batch_step_kernel<DIM, RPL>), dim3(grid), dim3(256), 0, st, pts_sorted, minsq, n, order,
                         n_buckets, box, keys, bcoord, L, n_lms, ctr, rec, out_idx, g_fps_lane_best ? 2 : 0, (u64*)nullptr, 0ull);
    int32_t now = 0;
    if (hipMemcpyAsync(&now, ctr + L, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return fail(FLOODER_E_LAUNCH, "flooder_fps_batched_f32: reading the landmark counter failed");
    if (now <= done) return fail(FLOODER_E_LAUNCH, "flooder_fps_batched_f32: no progress (internal error)");
    const double rate = (double)(now - done) / (double)round;  // landmarks per launch of this round
    done = now;
    launches += round;
    prev_round = round;
    const double est = (double)(n_lms - done) / rate;
    round = est > 24.0 ? (int)(est * 0.8) : (int)(est * 1.25) + 1;
    const int least = (n_lms - done + BatchCap<RPL>::CAP - 1) / BatchCap<RPL>::CAP;
    if (round < least) round = least;
  }
  if (launches_out) *launches_out = launches;
  return check_launch("fps_batched");
}

template <int DIM>
struct FpsBatchedOp {
  static int run(const float* pts, int64_t n, int ld, const float* pts_sorted, const int32_t* order, int n_lms,
                 int64_t start, int64_t* out_idx, float* minsq, float* box, u64* keys, float* bcoord, u64* best,
                 uint32_t* rec, int32_t* ctr, int32_t* launches_out, hipStream_t st) {
    // small clouds: 64-row buckets (more waves to spread a landmark's neighbourhood over) and a late switch (a
    // brute step over an L2-resident cloud costs about as much as a launch); large clouds: 256-row buckets
    const int rpl = batched_rpl(n);
    // (measured: 1 M / 1 k 2.44 ms at 64, 2.46 at 32, 2.52 at 96 and 128, 2.63 at 4; 16 M / 4 k 13.1 ms at 4, 13.2 at 8,
    // 14.3 at 32, 19.7 at 128)
    int k0 = g_fps_switch ? g_fps_switch : (n >= (4 << 20) ? 8 : 64);
    if (k0 < 2) k0 = 2;  // (the start point is applied by a brute-force step)
    if (rpl == 4)
      return run_batched<DIM, 4>(pts, n, ld, pts_sorted, order, n_lms, start, k0, out_idx, minsq, box, keys, bcoord, best,
                                 rec, ctr, launches_out, st);
    return run_batched<DIM, 1>(pts, n, ld, pts_sorted, order, n_lms, start, k0, out_idx, minsq, box, keys, bcoord, best,
                               rec, ctr, launches_out, st);
  }
};

}  // namespace

extern "C" {

int64_t flooder_fps_batched_max_points(void) { return (int64_t)64 * NREC * 4 * 64 * 64 * 4; }  // 256 blocks of 256-row buckets

int64_t flooder_fps_batched_rec_words(int64_t n_pts, int dim, int n_lms) {
  if (n_pts < 1 || dim < 1 || dim > FLOODER_MAX_DIM || n_lms < 1) return 0;
  const int64_t blocks = batched_rpl(n_pts) == 4 ? batched_blocks<4>(n_pts) : batched_blocks<1>(n_pts);
  const int dpr = padded_dim(dim) < 4 ? 4 : padded_dim(dim);
  return (int64_t)(n_lms + 4) * blocks * (12 + dpr);
}

int flooder_fps_batched_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* pts_sorted,
                            const int32_t* order, int n_lms, int64_t start, int64_t* out_idx, float* minsq,
                            float* bucket_box, uint64_t* bucket_keys, float* bucket_coord, uint64_t* work_best,
                            uint32_t* work_rec, int32_t* work_ctr, int32_t* launches_out, void* stream) {
  if (!pts || !pts_sorted || !order || !out_idx || !minsq || !bucket_box || !bucket_keys || !bucket_coord || !work_best ||
      !work_rec || !work_ctr || n_pts < 1 || n_lms < 1 || n_lms > n_pts || start < 0 || start >= n_pts || ld < dim ||
      dim < 1 || dim > FLOODER_MAX_DIM || n_pts > 0xfffffffeLL)
    return fail(FLOODER_E_ARG, "flooder_fps_batched_f32: bad argument");
  return dispatch_dim<FpsBatchedOp>(dim, pts, n_pts, ld, pts_sorted, order, n_lms, start, out_idx, minsq, bucket_box,
                                    reinterpret_cast<u64*>(bucket_keys), bucket_coord, reinterpret_cast<u64*>(work_best),
                                    work_rec, work_ctr, launches_out, (hipStream_t)stream);
}

}  // extern "C"
