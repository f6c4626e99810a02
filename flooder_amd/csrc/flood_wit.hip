// flood_wit.hip - witness sweep: a whole sparse simplex per wave, coarse samples first (gfx950; dim 2 and 3).
//
// Only the per-face MAXIMA of the nearest-neighbour distance are wanted (core.py:251-276), and in the sparse parts of
// a cloud - most simplices of a Gaussian sit in its tails - neighbouring lattice samples share their nearest point.
// A work item is ONE simplex with all its R samples; the points around the whole simplex are gathered, filtered and
// staged once (the cell sweep of flood_cell.hip does that per chunk of 256 samples, twenty times per tetrahedron):
//
//   1. region    box of the vertices and the extent along every face normal (the simplex as a 2(DIM+1)-plane
//                polytope P).  Every point x of the cloud gets an excess e(x) <= dist(x, P): the largest violation of a
//                box side or face plane.
//   2. stage     the leaves overlapping P grown by c_max are gathered through the box tree; a 64-bin histogram of the
//                excesses picks the largest c_sel <= c_max whose points (e(x) < c_sel) fit the LDS stage.  A sample p
//                with inner slack delta(p) (its distance to the nearest side of P) then has EVERY point within
//                c_sel + delta(p) of it on the stage: a minimum below (0.999 (c_sel + delta))^2 is exact ("certified").
//   3. coarse    a coarse sub-lattice of the samples (every M-th lattice point of every face, <= 256 rows chosen by
//                the host) is evaluated against the stage; every coarse sample keeps its WITNESS - the staged point
//                that attains its minimum - and the certified ones raise the running maxima of their faces.
//   4. fine      every other sample takes the distance to the witnesses of its (up to four) nearest coarse samples as
//                an upper bound: ub(p) >= d(p), computed with the arithmetic of an evaluated pair.  If ub(p) does not
//                exceed the running maximum of any face p lies on, p cannot raise a face value and is dropped; at
//                BASELINE cfg 2 that is 93 % of all samples, 99 % in the sparse simplices.  The others are queued in
//                LDS and evaluated against the stage 64 at a time; certified ones are delivered (the maxima rise and
//                later samples drop more easily), the rest - samples whose nearest point may lie beyond the staged
//                region - go to the exact finish (flood_finish.hip) through the flag list, with their bound as seed.
//
// A dropped sample never held a face maximum and a delivered value is exact, so the face values equal the exhaustive
// result bit for bit.  Simplices that are too dense for one stage are left to the cell sweep (their weight stays
// non-negative; handled ones are marked with weight -1).

#include "flood_common.hpp"
#include "flood_bvh.hpp"

using namespace flooder;

namespace flooder {
int g_wit_weight = 800;     // simplices with at most this many cloud points in their box (flooder_simplex_weight_f32) are tried
int g_wit_cmax_pct = 250;   // c_max in percent of the local point spacing
int g_wit_grid = 256 * 4;   // persistent workgroups (one simplex at a time each)
int g_wit_min_bins = 48;    // (measured: sparse simplices of a Gaussian keep 50 - 64 bins, the slivers along a surface 10 - 25)
//    // the stage must hold the points of at least this many of the 64 excess bins
int g_wit_max_in_pct = 8;   // points inside the simplex itself, in percent of its samples, it may hold at most (denser: every sample has a nearest point of its own, nothing to share)
int g_wit_max_leaves = 400; // leaves (16 points each) the region around a simplex may overlap; more: too dense around it, no attempt
int g_wit_max_eval = 768;   // queued samples an item evaluates against its stage at most (the rest: to the finish)
int g_wit_max_open = 48;    // coarse samples the stage may leave open; more: far field, the simplex is left to the cell sweep
int g_wit_max_live_pct = 12; // ... and the share of all samples that may survive the bound
int g_wit_adaptive = 0;     // 1: once half of the simplices tried had to be abandoned, only every 16th is still tried
int g_wit_flags = 0;        // test switches: 1 = no exact pass for the open samples, 2 = rounds not shared between waves
int g_wit_cmax_ext_pct = 60;  // ... and at most this share of the simplex's extent
int g_wit_surface_pct = 60;   // no attempt at all on a cloud that lies on a surface (the statistic of flood_common.hpp's cloud_kind_block,
//    // same threshold as the cell sweep's "cell_surface_pct"): every box that meets the sheet is too dense for one stage
//    // (cfg 3: 5581 simplices tried, none handled, 52 us); 0 = always try
}  // namespace flooder

namespace {

#ifndef FLOODER_WIT_BLOCKS
#define FLOODER_WIT_BLOCKS 4
#endif
constexpr int WBLOCKS = FLOODER_WIT_BLOCKS;  // workgroups per CU the register budget is set for
constexpr int WTHREADS = 256;  // one workgroup of four waves per simplex
constexpr int WWAVES = WTHREADS / 64;
constexpr int WCAP = 960;      // points staged per item
constexpr int WLEAF = 1024;    // leaves gathered per item (16 K candidate points)
constexpr int WFRONT = 192;    // inner nodes per level of the gather
constexpr int WCOARSE = FLOODER_WIT_MAX_COARSE;  // coarse samples per item (one per thread)
constexpr int WFOCUS = 16;      // open samples settled per focus round of the exact pass
constexpr int WROUNDS = 8;      // focus rounds per item at most
constexpr int WUR = 256;        // open samples resolved by the exact pass per item (beyond: to the finish)
constexpr int WQ = 768;        // queued live samples (beyond: handed to the finish with their bound)
constexpr int WROWS = FLOODER_WIT_MAX_ROWS;      // samples per simplex at most
constexpr int UNR = 4;         // candidate rows in flight per lane
constexpr int UNRF = 1;        // groups of 256 rows in flight in the pass over all samples
#ifndef FLOODER_WIT_FDEPTH
#define FLOODER_WIT_FDEPTH 1
#endif
constexpr int FDEPTH = FLOODER_WIT_FDEPTH;  // steps of the pass over all samples whose table rows are in flight
constexpr int NBIN = 64;
constexpr int PLANE_ROW = 24;  // (layout of simplex_planes_kernel, flood_cell.hip)
static_assert(WCOARSE == WTHREADS, "one coarse sample per thread");
static_assert(WUR == WTHREADS && WUR >= WCOARSE, "one open sample per thread in the exact pass");

template <int DP>
__device__ __forceinline__ void load_row_at(const float* __restrict__ base, uint32_t byte_off, float (&out)[DP]) {
  load_row<DP>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)byte_off), out);
}

__device__ __forceinline__ int lane_rank(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

struct WitPlan {
  const int32_t* coarse_rows;  // WCOARSE entries: row of coarse sample c, -1 beyond n_coarse
  const uint32_t* parents;     // per row: four coarse slots, 8 bits each (a coarse row's first parent is itself)
  int n_coarse;
};

struct WitOut {
  uint32_t* d2;          // (S, R) scratch: written for the tiles handed to the finish only (bit 31 = settled)
  int32_t* flag_list;
  int32_t* flag_count;
  float* weight;         // in: rough point count per simplex box; out: -1 for the simplices handled here
};

// The simplex as a polytope (box of the vertices, extents along the face normals).  Wave-uniform, computed once per
// item and parked in LDS: every phase loads what it needs into registers of its own scope - held across the whole
// item these sixty values (the compiler keeps uniform floats in vector registers) cost the kernel its occupancy.
template <int DIM>
struct Region {
  float pn[DIM + 1][DIM], po[DIM + 1], ps[DIM + 1], org[DIM], sext, blo[DIM], bhi[DIM], slo[DIM + 1], shi[DIM + 1], epsb;
  static constexpr int WORDS = (DIM + 1) * DIM + 4 * (DIM + 1) + 3 * DIM + 2;
  __device__ __forceinline__ void store(float* c) const {
    int i = 0;
#pragma unroll
    for (int f = 0; f <= DIM; ++f) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) c[i++] = pn[f][k];
      c[i++] = po[f]; c[i++] = ps[f]; c[i++] = slo[f]; c[i++] = shi[f];
    }
#pragma unroll
    for (int k = 0; k < DIM; ++k) { c[i++] = org[k]; c[i++] = blo[k]; c[i++] = bhi[k]; }
    c[i++] = sext;
    c[i++] = epsb;
  }
  __device__ __forceinline__ void load(const float* c) {
    int i = 0;
#pragma unroll
    for (int f = 0; f <= DIM; ++f) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) pn[f][k] = c[i++];
      po[f] = c[i++]; ps[f] = c[i++]; slo[f] = c[i++]; shi[f] = c[i++];
    }
#pragma unroll
    for (int k = 0; k < DIM; ++k) { org[k] = c[i++]; blo[k] = c[i++]; bhi[k] = c[i++]; }
    sext = c[i++];
    epsb = c[i++];
  }
  // squared radius within which EVERY point around sample p is on a stage that holds all points of excess < c_sel:
  // c_sel plus the sample's own distance to the nearest side of the polytope
  __device__ __forceinline__ float cert_limit(const float (&p)[DIM], float c_sel) const {
    float dl = __builtin_inff();
#pragma unroll
    for (int k = 0; k < DIM; ++k) dl = __builtin_fminf(dl, __builtin_fminf(p[k] - blo[k], bhi[k] - p[k]));
#pragma unroll
    for (int f = 0; f <= DIM; ++f) {
      if (po[f] < 1.0e37f) {  // (uniform: the plane is in use)
        float dd = -po[f];
#pragma unroll
        for (int k = 0; k < DIM; ++k) dd = __builtin_fmaf(pn[f][k], p[k] - org[k], dd);
        dl = __builtin_fminf(dl, __builtin_fminf(shi[f] - dd, dd - slo[f]));
      }
    }
    const float rr = 0.999f * (c_sel + __builtin_fmaxf(dl, 0.f));
    return rr * rr;
  }
};

// excess of a point: the smallest c for which it counts as "within c of the simplex" (<= its distance to the polytope)
template <int DIM>
struct Excess {
  float pn[DIM + 1][DIM], po[DIM + 1], org[DIM], blo_e[DIM], bhi_e[DIM], slo_t[DIM + 1], shi_t[DIM + 1], inv_den[DIM + 1];
  __device__ __forceinline__ Excess(const Region<DIM>& g, float c_max) {
#pragma unroll
    for (int k = 0; k < DIM; ++k) { blo_e[k] = g.blo[k] - g.epsb; bhi_e[k] = g.bhi[k] + g.epsb; org[k] = g.org[k]; }
#pragma unroll
    for (int f = 0; f <= DIM; ++f) {
      const float tol = g.ps[f] * (g.sext + c_max);
      slo_t[f] = g.slo[f] - tol;
      shi_t[f] = g.shi[f] + tol;
      inv_den[f] = 1.f / (1.001f + g.ps[f]);
      po[f] = g.po[f];
#pragma unroll
      for (int k = 0; k < DIM; ++k) pn[f][k] = g.pn[f][k];
    }
  }
  template <int DP>
  __device__ __forceinline__ float operator()(const float (&x)[DP]) const {
    float e = 0.f;
    float xr[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      e = __builtin_fmaxf(e, __builtin_fmaxf(x[k] - bhi_e[k], blo_e[k] - x[k]));
      xr[k] = x[k] - org[k];
    }
#pragma unroll
    for (int f = 0; f <= DIM; ++f) {
      float dd = -po[f];
#pragma unroll
      for (int k = 0; k < DIM; ++k) dd = __builtin_fmaf(pn[f][k], xr[k], dd);
      e = __builtin_fmaxf(e, __builtin_fmaxf(dd - shi_t[f], slo_t[f] - dd) * inv_den[f]);
    }
    return e;
  }
};

template <int DIM>
__global__ __launch_bounds__(WTHREADS, WBLOCKS) void wit_sweep_kernel(
    const float* __restrict__ pts, const float* __restrict__ nodes, Levels lv, const float* __restrict__ verts,
    const float* __restrict__ plane_tab, const float* __restrict__ weights, int k1, int R, int64_t n_simplices,
    float w_limit, float cmax_mult, float cmax_ext, int min_bins, int flags, int max_open, int max_live, int max_in, int adaptive, int leaf_cap, int max_eval, const int32_t* __restrict__ item_list, const int32_t* __restrict__ item_count, WitPlan plan, int32_t* __restrict__ queue, WitOut out, FaceAcc acc,
    unsigned long long* __restrict__ stats) {
  constexpr int DP = padded_dim(DIM);
  __shared__ float4 s_pts[WCAP + 4];
  __shared__ int s_leaf[WLEAF];
  __shared__ uint32_t s_qbuf[WQ + WQ / 2];  // queue of live samples: bounds, then rows (16 bit); focus rounds: point batches
  __shared__ int s_front[2 * WFRONT];       // gather: frontier of the current and the next tree level
  __shared__ float s_wit[3 * WCOARSE];      // witness of every coarse sample
  __shared__ int s_hist[NBIN];
  __shared__ uint32_t s_mf[32];
  __shared__ uint32_t s_unres[WROWS / 32];
  __shared__ uint32_t s_tkey[WROWS / 64];
  __shared__ int s_ftile[WROWS / 64];
  __shared__ int s_gn[MAXL + 2];  // nodes found per tree level ([0]: leaves), [MAXL]: overflow flag
  __shared__ int s_ctr[4];        // [0] points staged, [1] samples queued, [2] tiles flagged, [3] open samples
  __shared__ uint16_t s_ur_row[WUR];   // samples left open by the stage: row, bound, coordinates
  __shared__ uint32_t s_ur_best[WUR];
  __shared__ int16_t s_ur_slot[WUR];   // coarse slot of the entry (-1: not a coarse sample)
  __shared__ float s_ur_p[WUR * 3];
  __shared__ uint16_t s_focus[WFOCUS];
  __shared__ uint32_t s_fx[8];       // exact pass: [0] largest open bound, [1] focus samples
  __shared__ float s_rg[Region<DIM>::WORDS];
  __shared__ long long s_item;
  __shared__ int s_off;            // the sweep has switched itself off (see below)
  uint32_t* s_qub = s_qbuf;
  uint16_t* s_qrow = reinterpret_cast<uint16_t*>(s_qbuf + WQ);
  // (the focus rounds run while the queue is empty - before the pass over all samples, after the rounds - and stream
  // their candidate points through its storage: the stage itself stays intact for the rounds in between)
  float4* s_xb = reinterpret_cast<float4*>(s_qbuf);
  constexpr int XB = 256;  // points per batch
  static_assert(XB * sizeof(float4) <= sizeof(uint32_t) * (WQ + WQ / 2) && XB == WTHREADS, "a batch fits the queue's storage");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int top = lv.n_levels - 1;
  const int tiles64 = (R + 63) >> 6;
  // work counters (diagnostic runs only: stats != NULL) live in LDS - fourteen 64-bit counters in registers cost the
  // kernel 28 VGPRs it does not have
  // (an empty item list - a cloud on a surface, wit_list_kernel - : 1024 workgroups popping an empty queue through its
  // shards and barriers were 22 us of cfg 3's step)
  if (__builtin_amdgcn_readfirstlane(item_count[0]) == 0) return;
  __shared__ unsigned long long s_stat[24];
  enum { ST_HANDLED = 0, ST_HEAVY = 1, ST_OVER = 2, ST_DENSE = 3, ST_STAGED = 4, ST_CCERT = 5, ST_LIVE = 6, ST_ROUNDS = 7,
         ST_UNRES = 8, ST_FLAGGED = 9, ST_PAIRS = 10, ST_BINS = 11, ST_EXACT = 22, ST_EXACT_OVER = 23 };
  if (threadIdx.x < 24) s_stat[threadIdx.x] = 0ull;
  auto count = [&](int what, unsigned long long n) {  // (call from one lane per event)
    if (stats) atomicAdd(&s_stat[what], n);
  };

#ifdef FLOODER_WIT_TIMERS
  unsigned long long t_ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#define WPHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); t_ph[i] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define WPHASE(i) do {} while (0)
#endif
  int q_shard = (int)(blockIdx.x % QSHARDS), q_tried = 0;
  for (;;) {
    WPHASE(8);
    __syncthreads();  // (the previous item's LDS is free)
    if (wv == 0) {
      const int64_t it = queue_pop(queue, q_shard, q_tried, (int64_t)item_count[0], lane);
      if (lane == 0) {
        s_item = (long long)it;
        // (one reading for the whole workgroup: its waves must agree)
        const int tried = __hip_atomic_load(&queue[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int dropped = __hip_atomic_load(&queue[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_off = (adaptive && tried >= 8 && 2 * dropped > tried) ? 1 : 0;
      }
    }
    if (tid < MAXL + 2) s_gn[tid] = 0;
    if (tid < 4) s_ctr[tid] = 0;
    if (tid < NBIN) s_hist[tid] = 0;
    for (int i = tid; i < WROWS / 32; i += WTHREADS) s_unres[i] = 0u;
    for (int i = tid; i < WROWS / 64; i += WTHREADS) s_tkey[i] = 0u;
    __syncthreads();
    if (s_item < 0) break;
    const int64_t s = (int64_t)item_list[s_item];   // (wit_list_kernel: the simplices light enough, heaviest class first)
    const float w_s = out.weight[s];
    // the sweep watches its own success: a cloud whose light simplices are mostly abandoned again (a surface: few
    // points per box but dense where they are; voids of a dense cloud) is not worth the attempts - only every 16th
    // is still made, which keeps the counters alive.  (Racy reads, any outcome: the values do not depend on who
    // sweeps a simplex.)
    if (s_off != 0 && (s & 15) != 0) { if (tid == 0) count(ST_HEAVY, 1); continue; }
#ifdef FLOODER_WIT_TIMERS
    const unsigned long long t_item0 = __builtin_amdgcn_s_memrealtime();
#define WIT_REC(slot, val) do { if (stats && tid == 0) stats[64 + 12 * s + (slot)] = (unsigned long long)(val); } while (0)
#else
#define WIT_REC(slot, val) do {} while (0)
#endif
    WIT_REC(0, __float_as_uint(w_s));
#define WIT_ABANDON(what) { WIT_REC(10, 100 + what); WIT_REC(11, __builtin_amdgcn_s_memrealtime() - t_item0); if (tid == 0) { count(what, 1); if ((s & 7) == 0) { atomicAdd(&queue[8], 1); atomicAdd(&queue[9], 1); } } continue; }  // (every 8th simplex is counted: same-address atomics are slow)
    const float* vs = verts + s * (int64_t)k1 * DIM;
    WPHASE(0);

    // ---- 1. region: face planes of the simplex (table row written by simplex_planes_kernel), box of the vertices,
    // extents along the face normals -> LDS
    float c_max;
    {
      Region<DIM> rg;
      const float* pt = plane_tab + s * PLANE_ROW;
      typename RowVec<4>::type t[PLANE_ROW / 4];
#pragma unroll
      for (int i = 0; i < PLANE_ROW / 4; ++i) t[i] = load_uniform_row<4>(pt + 4 * i);
      auto at = [&](int i) { return t[i >> 2][i & 3]; };
#pragma unroll
      for (int k = 0; k < DIM; ++k) rg.org[k] = at(k);
      rg.sext = at(3);
#pragma unroll
      for (int f = 0; f <= DIM; ++f) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) rg.pn[f][k] = at(4 + 5 * f + k);
        rg.po[f] = at(4 + 5 * f + 3);
        rg.ps[f] = at(4 + 5 * f + 4) + 1e-6f;
      }
#pragma unroll
      for (int k = 0; k < DIM; ++k) { rg.blo[k] = __builtin_inff(); rg.bhi[k] = -__builtin_inff(); }
#pragma unroll
      for (int f = 0; f <= DIM; ++f) { rg.slo[f] = __builtin_inff(); rg.shi[f] = -__builtin_inff(); }
      float amax = 0.f;
      for (int j = 0; j < k1; ++j) {
        float v[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          v[k] = vs[j * DIM + k];
          rg.blo[k] = __builtin_fminf(rg.blo[k], v[k]);
          rg.bhi[k] = __builtin_fmaxf(rg.bhi[k], v[k]);
          amax = __builtin_fmaxf(amax, __builtin_fabsf(v[k]));
        }
#pragma unroll
        for (int f = 0; f <= DIM; ++f) {
          float dd = -rg.po[f];
#pragma unroll
          for (int k = 0; k < DIM; ++k) dd = __builtin_fmaf(rg.pn[f][k], v[k] - rg.org[k], dd);
          rg.slo[f] = __builtin_fminf(rg.slo[f], dd);
          rg.shi[f] = __builtin_fmaxf(rg.shi[f], dd);
        }
      }
      // (a sample is a rounded combination of the vertices: it may leave their box by a few ulps)
      rg.epsb = 8.f * 1.1920929e-7f * amax;
      float ext = 0.f, vol = 1.f;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        ext = __builtin_fmaxf(ext, rg.bhi[k] - rg.blo[k]);
        vol *= (rg.bhi[k] - rg.blo[k]);
      }
      const float n0 = __builtin_fmaxf(w_s, 1.f);
      const float h = DIM == 3 ? cbrtf(vol / n0) : __builtin_sqrtf(vol / n0);
      c_max = __builtin_fminf(cmax_mult * h, cmax_ext * ext);
      if (tid == 0) rg.store(s_rg);
    }
    if (!(c_max > 0.f) || !(c_max < 3.0e38f)) WIT_ABANDON(ST_OVER)
    __syncthreads();

    // ---- gather: leaves of the box tree overlapping [qlo, qhi] -> s_leaf, s_gn[0] of them; the frontier groups of a
    // level are dealt to the four waves, every level ends with a barrier
    float qlo[DIM], qhi[DIM];
    // children of up to GB frontier nodes per call: their boxes are loaded together (a call is a memory round trip)
    constexpr int GB = 4;
    auto test_children = [&](int lvl, const int (&grp)[GB], int ng, int* out_list, int* out_n, int cap) {
      const int lvl_count = (int)lv.count[lvl], lvl_off = (int)lv.off[lvl];
      bool hit[GB];
      float lo[GB][DP], hi[GB][DP];
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int idx = grp[u] * FAN + lane;
        hit[u] = u < ng && idx < lvl_count;
        const uint32_t nb_ = (uint32_t)(lvl_off + (hit[u] ? idx : 0)) * (uint32_t)(2 * DP * sizeof(float));
        load_row_at<DP>(nodes, nb_, lo[u]);
        load_row_at<DP>(nodes, nb_ + (uint32_t)(DP * sizeof(float)), hi[u]);
      }
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        if (u < ng) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) hit[u] = hit[u] && (lo[u][k] <= qhi[k]) && (hi[u][k] >= qlo[k]);
          const unsigned long long m = __ballot(hit[u]);
          const int cnt = __popcll(m);
          if (cnt != 0) {
            int base = 0;
            if (lane == 0) base = atomicAdd(out_n, cnt);
            base = wave_uniform(base);
            if (base + cnt > cap) {
              if (lane == 0) s_gn[MAXL] = 1;
            } else if (hit[u]) {
              out_list[base + lane_rank(m)] = grp[u] * FAN + lane;
            }
          }
        }
      }
    };
    // one level of the walk: the frontier's groups dealt to the waves, GB at a time; stops at an overflow
    auto walk_level = [&](int lvl, const int* fa, int* fb, int leaf_cap_) {
      const int na = s_gn[lvl];
      for (int f = wv * GB; f < na; f += WWAVES * GB) {
        if (*(volatile int*)&s_gn[MAXL] != 0) break;
        int grp[GB];
        const int ng = na - f < GB ? na - f : GB;
#pragma unroll
        for (int u = 0; u < GB; ++u) grp[u] = wave_uniform(fa[f + u < na ? f + u : f]);
        if (lvl == 1) test_children(0, grp, ng, s_leaf, &s_gn[0], leaf_cap_);
        else test_children(lvl - 1, grp, ng, fb, &s_gn[lvl - 1], WFRONT);
      }
    };
    auto walk_top = [&](int* fa, int leaf_cap_) {
      if (wv == 0) {
        const int g0_[GB] = {};
        if (top == 0) test_children(0, g0_, 1, s_leaf, &s_gn[0], leaf_cap_);
        else test_children(top, g0_, 1, fa, &s_gn[top], WFRONT);
      }
    };
    bool gathered = false;
    for (int att = 0; att < 1; ++att) {  // (one try: a region with more leaves than the cap is no case for this sweep)
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        qlo[k] = s_rg[Region<DIM>::WORDS - 2 - 3 * DIM + 3 * k + 1] - s_rg[Region<DIM>::WORDS - 1] - c_max;
        qhi[k] = s_rg[Region<DIM>::WORDS - 2 - 3 * DIM + 3 * k + 2] + s_rg[Region<DIM>::WORDS - 1] + c_max;
      }
      int* fa = s_front;
      int* fb = s_front + WFRONT;
      walk_top(fa, leaf_cap);
      __syncthreads();
      for (int lvl = top; lvl >= 1; --lvl) {
        if (s_gn[MAXL] != 0) break;  // (overflow: the lists are incomplete - block-uniform, read behind a barrier)
        walk_level(lvl, fa, fb, leaf_cap);
        __syncthreads();
        int* t = fa; fa = fb; fb = t;
      }
      if (s_gn[MAXL] == 0) { gathered = true; break; }
      __syncthreads();  // (everybody has seen the overflow flag)
      if (tid < MAXL + 2) s_gn[tid] = 0;
      c_max *= 0.5f;
      __syncthreads();
    }
    if (!gathered) WIT_ABANDON(ST_OVER)
    const int n_leaves = s_gn[0];
    WIT_REC(1, n_leaves);
    WPHASE(1);

    const float bin_scale = (float)NBIN / c_max;
    const int n_cand = n_leaves * LEAF;
    int n_keep_bins, n_stage;
    float c_sel;
    {  // (scope of the region's registers)
    Region<DIM> rg_;
    rg_.load(s_rg);
    const Excess<DIM> excess(rg_, c_max);
    // ---- 2a. histogram of the excesses (LDS atomics)
    for (int ib = 0; ib < n_cand; ib += WTHREADS * UNR) {
      float x[UNR][DP];
      bool in[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = ib + u * WTHREADS + tid;
        in[u] = idx < n_cand;
        const uint32_t row = in[u] ? (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF) : 0u;
        load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const float eb = excess(x[u]) * bin_scale;   // (+inf padding rows: eb = inf, not counted)
        if (in[u] && eb < (float)NBIN) atomicAdd(&s_hist[(int)eb], 1);
      }
    }
    __syncthreads();
    WPHASE(2);
    {
      int cum = s_hist[lane];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(cum, o);
        if (lane >= o) cum += t;
      }
      const unsigned long long fit = __ballot(cum <= WCAP);  // (cum is non-decreasing: a prefix of the lanes)
      n_keep_bins = __popcll(fit);
      n_stage = n_keep_bins > 0 ? __shfl(cum, n_keep_bins - 1) : 0;
    }
    WIT_REC(2, s_hist[0]); WIT_REC(3, n_stage); WIT_REC(4, n_keep_bins);
    // (bin 0: the points inside the simplex or within c_max / 64 of it)
    if (s_hist[0] > max_in) WIT_ABANDON(ST_DENSE)
    if (n_keep_bins < min_bins || n_stage == 0) WIT_ABANDON(ST_DENSE)   // too dense for one stage (or nothing near)
    c_sel = (float)n_keep_bins / bin_scale;
    // ---- 2b. stage the points of the kept bins (any order: a minimum does not care)
    __syncthreads();  // (the frontier inside the stage is dead)
    for (int ib = 0; ib < n_cand; ib += WTHREADS * UNR) {
      float x[UNR][DP];
      bool in[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = ib + u * WTHREADS + tid;
        in[u] = idx < n_cand;
        const uint32_t row = in[u] ? (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF) : 0u;
        load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const float eb = excess(x[u]) * bin_scale;
        const bool keep = in[u] && eb < (float)n_keep_bins;   // (the same test as the histogram's: bin < n_keep_bins)
        const unsigned long long m = __ballot(keep);
        if (m != 0ull) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&s_ctr[0], __popcll(m));
          base = wave_uniform(base);
          if (keep) {
            float4 v;
            v.x = x[u][0];
            v.y = x[u][1];
            v.z = DIM > 2 ? x[u][DIM > 2 ? 2 : 0] : 0.f;
            v.w = 0.f;
            s_pts[base + lane_rank(m)] = v;
          }
        }
      }
    }
    }
    __syncthreads();
    const int K = s_ctr[0];   // (= n_stage)
    if (tid < 4) s_pts[K + tid] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
    // running maxima of this simplex's faces as other workgroups have left them
    if (tid < acc.n_faces)
      s_mf[tid] = __hip_atomic_load(acc.face_bits + acc.slot_of(s, tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) { count(ST_STAGED, (unsigned long long)K); count(ST_BINS, (unsigned long long)n_keep_bins); }
    WPHASE(3);

    // ---- helpers: a sample from its weight row; its certification limit; delivery of certified values
    auto make_sample = [&](int r, float (&p)[DIM]) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[k] = 0.f;
      if (k1 == 4) {
        const float4 w4 = *reinterpret_cast<const float4*>(weights + (int64_t)r * 4);
        const float wj[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) p[k] = __builtin_fmaf(wj[j], vs[j * DIM + k], p[k]);
        }
      } else {
        for (int j = 0; j < k1; ++j) {
          const float w = weights[(int64_t)r * k1 + j];
#pragma unroll
          for (int k = 0; k < DIM; ++k) p[k] = __builtin_fmaf(w, vs[j * DIM + k], p[k]);
        }
      }
    };
    // certified samples raise the running maxima of their faces (per wave: one global atomic per face at most,
    // and only when it can raise the value this workgroup has seen)
    auto deliver = [&](bool on, uint32_t mb, float val) {
      uint32_t um = wave_or_u32(on ? mb : 0u);
      while (um) {  // (wave-uniform)
        const int f = __builtin_ctz(um);
        um &= um - 1u;
        const uint32_t v = wave_max_u32((on && ((mb >> f) & 1u)) ? __float_as_uint(val) : 0u);
        if (lane == 0 && v > s_mf[f]) {
          atomicMax(&acc.face_bits[acc.slot_of(s, f)], v);
          atomicMax(&s_mf[f], v);
        }
      }
    };
    // threshold of a sample: the smallest running maximum among the faces it lies on
    auto threshold = [&](uint32_t mb) -> uint32_t {
      uint32_t thr = 0xffffffffu;
      uint32_t um = wave_or_u32(mb);
      while (um) {  // (wave-uniform: the faces present among this wave's 64 rows)
        const int f = __builtin_ctz(um);
        um &= um - 1u;
        const uint32_t v = s_mf[f];
        if ((mb >> f) & 1u) thr = v < thr ? v : thr;
      }
      return thr;
    };
    // a sample goes straight to the finish with its bound (queue or list full)
    auto to_finish = [&](int r, float val) {
      out.d2[s * (int64_t)R + r] = __float_as_uint(val);
      atomicOr(&s_unres[r >> 5], 1u << (r & 31));
      atomicMax(&s_tkey[r >> 6], __float_as_uint(val));
    };
    // ---- open samples (their nearest point may lie beyond the staged region) are settled in FOCUS ROUNDS: the open samples
    // with the largest bounds (within focus_frac of the largest, at most WFOCUS) get every point within that bound of
    // their box streamed through the stage, lanes over points - exact by construction; their values raise the face
    // maxima, and most of the other open samples - the neighbours of the deepest one in a void - drop against those
    // without a search of their own.  What is still open after WROUNDS rounds goes to the finish.
    // Returns false when `give_up` is set and samples are left open (too many points within their bounds, or out of
    // rounds): the caller abandons the item - nothing has been handed to the finish then.
    auto focus_rounds = [&](int n_ur, bool give_up) -> bool {
      bool all_settled = true;
      if (n_ur > 0) {
        const bool mine = tid < n_ur;
        const int r = mine ? (int)s_ur_row[tid] : 0;
        const uint32_t mb = mine ? acc.memb[r] : 0u;
        bool done = !mine;
        for (int round = 0; round <= WROUNDS; ++round) {
          // who is still alive?  (every lane takes part in the threshold's wave-wide reductions)
          const uint32_t thr = threshold(done ? 0u : mb);
          const uint32_t bb = mine ? s_ur_best[tid] : 0u;
          const bool alive = !done && bb > thr;
          done = done || !alive;
          if (tid < 8) s_fx[tid] = tid < 4 ? 0u : 0u;
          __syncthreads();
          {
            const uint32_t wm = wave_max_u32(alive ? bb : 0u);
            if (lane == 0 && wm != 0u) atomicMax(&s_fx[0], wm);
          }
          __syncthreads();
          const uint32_t bmax = s_fx[0];
          if (bmax == 0u) break;                       // (block-uniform) nothing left
          if (round == WROUNDS) {                      // out of rounds: the rest goes to the finish
            if (alive && !give_up) to_finish(r, __uint_as_float(bb));
            all_settled = false;
            break;
          }
          // focus set: the largest bounds, compacted (at most WFOCUS)
          bool focus = alive && __uint_as_float(bb) >= 0.97f * __uint_as_float(bmax);
          {
            const unsigned long long mf_ = __ballot(focus);
            if (mf_ != 0ull) {
              int base = 0;
              if (lane == 0) base = atomicAdd((int*)&s_fx[1], __popcll(mf_));
              base = wave_uniform(base);
              const int pos = base + lane_rank(mf_);
              focus = focus && pos < WFOCUS;
              if (focus) s_focus[pos] = (uint16_t)tid;
            }
          }
          __syncthreads();
          const int n_f = (int)s_fx[1] < WFOCUS ? (int)s_fx[1] : WFOCUS;
          // box of the focus samples and their largest bound (every wave computes the same from the list)
          float rmax2 = 0.f;
          {
            const bool on = lane < n_f;
            const int e = on ? (int)s_focus[lane] : 0;
            rmax2 = wave_max_f32(on ? __uint_as_float(s_ur_best[e]) : 0.f);
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const float v = s_ur_p[e * 3 + k];
              qlo[k] = wave_min_f32(on ? v : __builtin_inff());
              qhi[k] = wave_max_f32(on ? v : -__builtin_inff());
            }
          }
          const float rho = __builtin_sqrtf(rmax2) * 1.00001f + 1e-30f;
#pragma unroll
          for (int k = 0; k < DIM; ++k) { qlo[k] -= rho; qhi[k] += rho; }
          if (tid < MAXL + 2) s_gn[tid] = 0;
          __syncthreads();
          {
            int* fa = s_front;
            int* fb = s_front + WFRONT;
            walk_top(fa, WLEAF);
            __syncthreads();
            for (int lvl = top; lvl >= 1; --lvl) {
              if (s_gn[MAXL] != 0) break;
              walk_level(lvl, fa, fb, WLEAF);
              __syncthreads();
              int* t = fa; fa = fb; fb = t;
            }
          }
          if (s_gn[MAXL] != 0) {  // too many points within the bound: the tree search of the finish is the better tool
            if (alive && !give_up) to_finish(r, __uint_as_float(bb));
            if (tid == 0) count(ST_EXACT_OVER, 1);
            all_settled = false;
            break;
          }
          const int n_cand2 = s_gn[0] * LEAF;
          for (int ib = 0; ib < n_cand2; ib += XB) {
            __syncthreads();  // (the previous batch has been read)
            if (tid == 0) s_ctr[0] = 0;
            __syncthreads();
            {
              float x[DP];
              const int idx = ib + tid;
              bool keep = idx < n_cand2;
              const uint32_t row = keep ? (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF) : 0u;
              load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x);
#pragma unroll
              for (int k = 0; k < DIM; ++k) keep = keep && (x[k] >= qlo[k]) && (x[k] <= qhi[k]);
              const unsigned long long m = __ballot(keep);
              if (m != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_ctr[0], __popcll(m));
                base = wave_uniform(base);
                if (keep) {
                  float4 v;
                  v.x = x[0];
                  v.y = x[1];
                  v.z = DIM > 2 ? x[DIM > 2 ? 2 : 0] : 0.f;
                  v.w = 0.f;
                  s_xb[base + lane_rank(m)] = v;
                }
              }
            }
            __syncthreads();
            const int nb = s_ctr[0];
            for (int j = wv; j < n_f; j += WWAVES) {  // (focus sample j belongs to this wave alone)
              const int e = (int)s_focus[j];
              float pj[DIM];
#pragma unroll
              for (int k = 0; k < DIM; ++k) pj[k] = s_ur_p[e * 3 + k];
              float m2 = __builtin_inff();
              int bi = 0;
              for (int i = lane; i < nb; i += 64) {
                const float4 xx = s_xb[i];
                float t0 = pj[0] - xx.x;
                float d2 = t0 * t0;
                t0 = pj[1] - xx.y;
                d2 = __builtin_fmaf(t0, t0, d2);
                if constexpr (DIM == 3) {
                  t0 = pj[2] - xx.z;
                  d2 = __builtin_fmaf(t0, t0, d2);
                }
                if (d2 < m2) { m2 = d2; bi = i; }
              }
              const float wm = wave_min_f32(m2);
              if (__float_as_uint(wm) < s_ur_best[e]) {  // (wave-uniform) a nearer point: new minimum, new witness
                const int lw = __builtin_ctzll(__ballot(m2 == wm));
                if (lane == lw) {
                  s_ur_best[e] = __float_as_uint(wm);
                  const int slot = (int)s_ur_slot[e];
                  if (slot >= 0) {
                    const float4 xx = s_xb[bi];
                    s_wit[3 * slot + 0] = xx.x;
                    s_wit[3 * slot + 1] = xx.y;
                    s_wit[3 * slot + 2] = xx.z;
                  }
                }
              }
            }
            if (tid == 0) count(ST_PAIRS, (unsigned long long)nb * (unsigned long long)n_f);
          }
          __syncthreads();
          // the focus samples are exact now: deliver them (all lanes of every wave take part in the reductions)
          deliver(focus, mb, __uint_as_float(focus ? s_ur_best[tid] : 0u));
          done = done || focus;
          if (tid == 0) count(ST_EXACT, 1);
          __syncthreads();
        }
      }
      return all_settled;
    };

    // ---- 3. coarse samples (one per thread) against the stage, with witnesses
    {
      const int c = tid;
      const int crow = c < plan.n_coarse ? plan.coarse_rows[c] : -1;
      float p[DIM];
      make_sample(crow < 0 ? 0 : crow, p);
      const uint32_t mb = crow >= 0 ? acc.memb[crow] : 0u;
      float best = __builtin_inff();
      int wj = 0;
      for (int j = 0; j < K; j += 4) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = s_pts[j + u];
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float t0 = p[0] - x[u].x;
          float d2 = t0 * t0;
          t0 = p[1] - x[u].y;
          d2 = __builtin_fmaf(t0, t0, d2);
          if constexpr (DIM == 3) {
            t0 = p[2] - x[u].z;
            d2 = __builtin_fmaf(t0, t0, d2);
          }
          d[u] = d2;
        }
        const float m = __builtin_fminf(__builtin_fminf(d[0], d[1]), __builtin_fminf(d[2], d[3]));
        if (m < best) { best = m; wj = j; }
      }
      if (lane == 0) count(ST_PAIRS, (unsigned long long)K * 64ull);
      {
        // the witness: the first point of the winning group of four that attains the minimum
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = s_pts[wj + u];
        float4 w = x[3];
#pragma unroll
        for (int u = 2; u >= 0; --u) {
          float t0 = p[0] - x[u].x;
          float d2 = t0 * t0;
          t0 = p[1] - x[u].y;
          d2 = __builtin_fmaf(t0, t0, d2);
          if constexpr (DIM == 3) {
            t0 = p[2] - x[u].z;
            d2 = __builtin_fmaf(t0, t0, d2);
          }
          if (d2 == best) w = x[u];
        }
        // (the leaf list is dead since the barrier behind the staging pass)
        s_wit[3 * c + 0] = w.x;
        s_wit[3 * c + 1] = w.y;
        s_wit[3 * c + 2] = w.z;
      }
      Region<DIM> rg_;
      rg_.load(s_rg);
      const bool cert = crow >= 0 && best <= rg_.cert_limit(p, c_sel);
      if (stats) { const unsigned long long mc_ = __ballot(cert); if (lane == 0) count(ST_CCERT, (unsigned long long)__popcll(mc_)); }
      deliver(cert, mb, best);
      // coarse samples the stage leaves open: on the list of the focus rounds below
      const bool open_c = crow >= 0 && !cert;
      const unsigned long long mo = __ballot(open_c);
      if (mo != 0ull) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_ctr[3], __popcll(mo));
        base = wave_uniform(base);
        if (open_c) {
          const int pos = base + lane_rank(mo);   // (< WCOARSE = WUR)
          s_ur_row[pos] = (uint16_t)crow;
          s_ur_slot[pos] = (int16_t)c;
          s_ur_best[pos] = __float_as_uint(best);
#pragma unroll
          for (int k = 0; k < DIM; ++k) s_ur_p[pos * 3 + k] = p[k];
        }
      }
    }
    __syncthreads();
    WPHASE(4);
    // ---- 3b. the face maxima must be (close to) final BEFORE the other samples are bounded against them: a face
    // whose deepest coarse sample is still open would let every sample of its void through.  Focus rounds settle
    // the open coarse samples that matter; a simplex where that takes too many points or rounds - far field: the
    // inside of a tube, a void - is left to the cell sweep and the tree search of the finish.
    WIT_REC(5, s_ctr[3]);
    if (!(flags & 4)) {
      if (s_ctr[3] > max_open) WIT_ABANDON(ST_DENSE)
      const bool ok = focus_rounds(s_ctr[3], true);
      if (!ok) WIT_ABANDON(ST_DENSE)
      __syncthreads();
    }
    if (tid == 0) s_ctr[3] = 0;
    __syncthreads();

    // ---- 4. all samples: bound from the witnesses of the nearest coarse samples; the live ones are queued.
    // (the table rows of the next step are in flight while this one is worked on)
    struct FineRows {
      float4 w4[UNRF];
      uint32_t par[UNRF], mb[UNRF];
      int rr[UNRF];
      bool valid[UNRF];
    };
    auto load_rows = [&](int g0, FineRows& fr) {
#pragma unroll
      for (int u = 0; u < UNRF; ++u) {
        const int r = g0 + u * WTHREADS + tid;
        fr.valid[u] = r < R;
        fr.rr[u] = fr.valid[u] ? r : R - 1;
        if (k1 == 4) fr.w4[u] = *reinterpret_cast<const float4*>(weights + (int64_t)fr.rr[u] * 4);
        fr.par[u] = plan.parents[fr.rr[u]];
        fr.mb[u] = fr.valid[u] ? acc.memb[fr.rr[u]] : 0u;
      }
    };
    // (FDEPTH steps ahead: a step is an L2 round trip - ~1 us - for a few hundred instructions of work)
    FineRows ring[FDEPTH];
#pragma unroll
    for (int a = 0; a < FDEPTH; ++a)
      if (a * WTHREADS * UNRF < R) load_rows(a * WTHREADS * UNRF, ring[a]);
    for (int g0 = 0; g0 < R; g0 += WTHREADS * UNRF) {
      const FineRows cur = ring[0];
#pragma unroll
      for (int a = 0; a + 1 < FDEPTH; ++a) ring[a] = ring[a + 1];
      if (g0 + FDEPTH * WTHREADS * UNRF < R) load_rows(g0 + FDEPTH * WTHREADS * UNRF, ring[FDEPTH - 1]);
#pragma unroll
      for (int u = 0; u < UNRF; ++u) {
        if (g0 + u * WTHREADS >= R) break;  // (block-uniform)
        float p[DIM];
        if (k1 == 4) {
          const float wj[4] = {cur.w4[u].x, cur.w4[u].y, cur.w4[u].z, cur.w4[u].w};
#pragma unroll
          for (int k = 0; k < DIM; ++k) p[k] = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < DIM; ++k) p[k] = __builtin_fmaf(wj[j], vs[j * DIM + k], p[k]);
          }
        } else {
          make_sample(cur.rr[u], p);
        }
        float ub = __builtin_inff();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (int)((cur.par[u] >> (8 * j)) & 0xffu);
          float t0 = p[0] - s_wit[3 * c + 0];
          float d2 = t0 * t0;
          t0 = p[1] - s_wit[3 * c + 1];
          d2 = __builtin_fmaf(t0, t0, d2);
          if constexpr (DIM == 3) {
            t0 = p[2] - s_wit[3 * c + 2];
            d2 = __builtin_fmaf(t0, t0, d2);
          }
          ub = __builtin_fminf(ub, d2);
        }
        const uint32_t thr = threshold(cur.mb[u]);
        const bool live = cur.valid[u] && __float_as_uint(ub) > thr;
        const unsigned long long m = __ballot(live);
        if (m != 0ull) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&s_ctr[1], __popcll(m));
          base = wave_uniform(base);
          if (live) {
            const int pos = base + lane_rank(m);
            if (pos < WQ) {
              s_qub[pos] = __float_as_uint(ub);
              s_qrow[pos] = (uint16_t)cur.rr[u];
            } else {
              to_finish(cur.rr[u], ub);
            }
          }
          if (lane == 0) count(ST_LIVE, (unsigned long long)__popcll(m));
        }
      }
    }
    __syncthreads();
    WPHASE(5);
    WIT_REC(6, s_ctr[1]);
    // too many samples survive the bound (a ridge of the distance function - the axis of a tube, a medial surface:
    // whole patches of samples within a lattice step of the maximum): no gain here, the simplex goes to the cell sweep
    if (s_ctr[1] > max_live && s_ctr[2] == 0 && !(flags & 8)) WIT_ABANDON(ST_OVER)

    // ---- 5. the queued samples against the stage: rounds of 64 samples; while there are fewer rounds than waves a
    // round is shared by several waves, each taking a part of the stage (minima combined in the queue entry)
    {
      const int n_q_all = s_ctr[1] < WQ ? s_ctr[1] : WQ;
      if (n_q_all < s_ctr[1] && tid == 0) count(ST_UNRES, (unsigned long long)(s_ctr[1] - n_q_all));
      // (an item's time is bounded: at most max_eval queued samples are evaluated here - a round per wave -, the rest
      // goes to the finish with its bound; a simplex with hundreds of live samples used to be the tail of the launch)
      const int n_q = n_q_all < max_eval ? n_q_all : max_eval;
      for (int qi = n_q + tid; qi < n_q_all; qi += WTHREADS) to_finish((int)s_qrow[qi], __uint_as_float(s_qub[qi]));
      if (n_q < n_q_all && tid == 0) count(ST_UNRES, (unsigned long long)(n_q_all - n_q));
      const int n_rounds_q = (n_q + 63) >> 6;
      const int parts = (n_rounds_q >= WWAVES || (flags & 2)) ? 1 : (n_rounds_q >= 2 ? 2 : 4);
      const int k_part = (((K + parts - 1) / parts) + 3) & ~3;
      for (int t = wv; t < n_rounds_q * parts; t += WWAVES) {
        const int rnd = t / parts, part = t - rnd * parts;
        const int qi = rnd * 64 + lane;
        const bool mine = qi < n_q;
        const int r = mine ? (int)s_qrow[qi] : 0;
        float best = mine ? __uint_as_float(s_qub[qi]) : 0.f;
        float p[DIM];
        make_sample(r, p);
        const uint32_t mb = mine ? acc.memb[r] : 0u;
        // still live?  (the maxima have risen since the sample was queued)
        const uint32_t thr = threshold(mb);  // (every lane takes part: wave-wide reductions inside)
        const bool act = mine && __float_as_uint(best) > thr;
        if (__ballot(act) == 0ull) continue;
        if (part == 0 && lane == 0) count(ST_ROUNDS, 1);
        const int j1 = (part + 1) * k_part < K ? (part + 1) * k_part : K;
        const float seed = best;
        for (int j = part * k_part; j < j1; j += 4) {
          float4 x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = s_pts[j + u];   // (entries behind K: +inf pads or real points - harmless)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float t0 = p[0] - x[u].x;
            float d2 = t0 * t0;
            t0 = p[1] - x[u].y;
            d2 = __builtin_fmaf(t0, t0, d2);
            if constexpr (DIM == 3) {
              t0 = p[2] - x[u].z;
              d2 = __builtin_fmaf(t0, t0, d2);
            }
            best = __builtin_fminf(best, d2);
          }
        }
        if (lane == 0) count(ST_PAIRS, (unsigned long long)(j1 - part * k_part) * 64ull);
        if (act && best < seed) atomicMin(&s_qub[qi], __float_as_uint(best));
      }
      __syncthreads();
      Region<DIM> rg_;
      rg_.load(s_rg);
      for (int rnd = wv; rnd < n_rounds_q; rnd += WWAVES) {
        const int qi = rnd * 64 + lane;
        const bool mine = qi < n_q;
        const int r = mine ? (int)s_qrow[qi] : 0;
        const float best = mine ? __uint_as_float(s_qub[qi]) : 0.f;
        float p[DIM];
        make_sample(r, p);
        const uint32_t mb = mine ? acc.memb[r] : 0u;
        const uint32_t thr = threshold(mb);  // (every lane takes part: wave-wide reductions inside)
        const bool act = mine && __float_as_uint(best) > thr;
        const bool cert = act && best <= rg_.cert_limit(p, c_sel);
        const bool unres = act && !cert;
        const unsigned long long mu = __ballot(unres);
        if (mu != 0ull) {  // open samples: on the list of the exact pass below (or, list full, to the finish)
          int base = 0;
          if (lane == 0) base = atomicAdd(&s_ctr[3], __popcll(mu));
          base = wave_uniform(base);
          if (unres) {
            const int pos = base + lane_rank(mu);
            if (pos < WUR && !(flags & 1)) {
              s_ur_row[pos] = (uint16_t)r;
              s_ur_slot[pos] = (int16_t)-1;
              s_ur_best[pos] = __float_as_uint(best);
#pragma unroll
              for (int k = 0; k < DIM; ++k) s_ur_p[pos * 3 + k] = p[k];
            } else {
              to_finish(r, best);
            }
          }
          if (lane == 0) count(ST_UNRES, (unsigned long long)__popcll(mu));
        }
        deliver(cert, mb, best);
      }
    }
    __syncthreads();
    WPHASE(6);

    // ---- 5b. the samples the stage left open: focus rounds (above); what they cannot settle goes to the finish
    if (!(flags & 1)) focus_rounds(s_ctr[3] < WUR ? s_ctr[3] : WUR, false);
    __syncthreads();
    WPHASE(9);

    // ---- 6. tiles with samples still unresolved go to the exact finish: the other rows of such a tile are marked settled
    for (int t = tid; t < tiles64; t += WTHREADS)
      if (s_tkey[t] != 0u) s_ftile[atomicAdd(&s_ctr[2], 1)] = t;
    __syncthreads();
    {
      const int n_ft = s_ctr[2];
      if (n_ft > 0) {
        if (wv == 0) {
          int base = 0;
          if (lane == 0) base = atomicAdd(out.flag_count, n_ft);
          base = wave_uniform(base);
          for (int i = lane; i < n_ft; i += 64) {
            const int t = s_ftile[i];
            const uint32_t key = s_tkey[t];
            const int item = (int)(s * tiles64 + t);
            out.flag_list[base + i] = item;
            if (acc.flag_key) {
              acc.flag_key[base + i] = key;
              atomicAdd(&acc.flag_hist[key >> 19], 1);
            }
            if (acc.top) {
              const unsigned long long old =
                  atomicMax(&acc.top[s], ((unsigned long long)key << 32) | (unsigned long long)(uint32_t)item);
              if (old == 0ull) acc.top_list[atomicAdd(acc.top_count, 1)] = (int)s;
            }
          }
          if (lane == 0) count(ST_FLAGGED, (unsigned long long)n_ft);
        }
        for (int i = wv; i < n_ft; i += WWAVES) {
          const int r = s_ftile[i] * 64 + lane;
          if (r < R && !((s_unres[r >> 5] >> (r & 31)) & 1u)) out.d2[s * (int64_t)R + r] = SETTLED_BIT;
        }
      }
    }
    WIT_REC(10, 1); WIT_REC(11, __builtin_amdgcn_s_memrealtime() - t_item0); WIT_REC(7, s_ctr[2]);
    if (tid == 0) { out.weight[s] = -1.f; if ((s & 7) == 0) atomicAdd(&queue[8], 1); }
    if (tid == 0) count(ST_HANDLED, 1);
    WPHASE(7);
  }
  __syncthreads();
  if (stats) {
#ifdef FLOODER_WIT_TIMERS
    if (tid == 0)
      for (int i = 0; i < 10; ++i) atomicAdd(&stats[12 + i], t_ph[i]);
#endif
    if (tid < 24 && (tid < 12 || tid >= 22) && s_stat[tid] != 0ull) atomicAdd(&stats[tid], s_stat[tid]);
  }
}

// The simplices the witness sweep tries (0 <= weight <= limit), heaviest class first (the long items - time goes with
// the points around a simplex - start first, the light ones fill the tail): one block.  A persistent
// workgroup that pops a simplex only to find it too heavy pays an atomic round trip and two barriers for nothing -
// 77 us per launch where every simplex is heavy.
constexpr int WCLASSES = 4;
// A stable counting sort in one block: every thread owns a RUN of consecutive simplices (their weights staged in LDS by
// coalesced loads first - a run read straight from memory is a chain of cache misses), counts its classes, one
// block-wide exclusive scan per class (wave scans + partial sums through LDS), and writes its run: two barriers.
// (The first version took the simplices in strides of 1024 and ran ballots and two barriers per stride: 16 us for
// cfg 2's 6052 simplices, a launch that every step waits for.)
constexpr int WLIST_LDS = 8192;   // weights staged; more simplices: the runs are read from memory
__global__ __launch_bounds__(1024) void wit_list_kernel(const float* __restrict__ weight, int n, float limit,
                                                        int32_t* __restrict__ list, int32_t* __restrict__ count,
                                                        const int32_t* __restrict__ kind, int surface_pct) {
  __shared__ int s_cnt[WCLASSES][16];
  __shared__ float s_w[WLIST_LDS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (kind != nullptr) {   // a cloud on a surface: an empty list, the sweep's workgroups leave at once
    const int inner = kind[2], all = kind[3];
    if (all > 0 && (long long)inner * 100 < (long long)all * surface_pct) {
      if (threadIdx.x == 0) count[0] = 0;
      return;
    }
  }
  auto cls = [&](float w) -> int {  // 0 = heaviest; -1: not for this sweep
    if (!(w >= 0.f) || !(w <= limit)) return -1;
    return w > 0.5f * limit ? 0 : (w > 0.25f * limit ? 1 : (w > 0.125f * limit ? 2 : 3));
  };
  const bool staged = n <= WLIST_LDS;
  if (staged) {
#pragma unroll
    for (int u = 0; u < WLIST_LDS / 1024; ++u) {   // (eight loads in flight)
      const int i = u * 1024 + (int)threadIdx.x;
      if (i < n) s_w[i] = weight[i];
    }
    __syncthreads();
  }
  const int per = ((n + 1023) / 1024) | 1;   // (odd: the runs start in different LDS banks)
  const int i0 = (int)threadIdx.x * per < n ? (int)threadIdx.x * per : n;
  const int i1 = i0 + per < n ? i0 + per : n;
  int cnt[WCLASSES];
#pragma unroll
  for (int k = 0; k < WCLASSES; ++k) cnt[k] = 0;
  for (int i = i0; i < i1; ++i) {
    const int c = cls(staged ? s_w[i] : weight[i]);
#pragma unroll
    for (int k = 0; k < WCLASSES; ++k) cnt[k] += c == k ? 1 : 0;
  }
  int excl[WCLASSES];   // simplices of the class in the threads before this one
#pragma unroll
  for (int k = 0; k < WCLASSES; ++k) {
    int v = cnt[k];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o);
      v += lane >= o ? t : 0;
    }
    excl[k] = v - cnt[k];
    if (lane == 63) s_cnt[k][wv] = v;
  }
  __syncthreads();
  int base = 0;   // classes in order, heaviest first
#pragma unroll
  for (int k = 0; k < WCLASSES; ++k) {
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int c = s_cnt[k][w];
      before += w < wv ? c : 0;
      total += c;
    }
    excl[k] += base + before;
    base += total;
  }
  if (threadIdx.x == 0) count[0] = base;
  for (int i = i0; i < i1; ++i) {   // (order kept inside a class)
    const int c = cls(staged ? s_w[i] : weight[i]);
#pragma unroll
    for (int k = 0; k < WCLASSES; ++k)
      if (c == k) list[excl[k]++] = i;
  }
}

template <int DIM>
struct WitOp {
  static int run(const float* pts, const float* nodes, const Levels& lv, const float* verts, float* plane_tab,
                 const float* weights, int k1, int R, int64_t ns, WitPlan plan, int32_t* queue, int32_t* item_list,
                 int32_t* item_count, WitOut out, FaceAcc acc, unsigned long long* stats, const int32_t* kind,
                 hipStream_t st) {
    if constexpr (DIM == 2 || DIM == 3) {
      if (!planes_are_done(verts, plane_tab, ns, st)) {   // (flooder_simplex_prepare_f32 may have written the rows already)
        const int rc = launch_simplex_planes(DIM, verts, k1, ns, plane_tab, st);
        if (rc != FLOODER_OK) return rc;
      }
      planes_done_for(verts, plane_tab, ns, st);   // (the cell sweep's entry, next on this stream, need not repeat it)
      hipLaunchKernelGGL(wit_list_kernel, dim3(1), dim3(1024), 0, st, out.weight, (int)ns, (float)g_wit_weight, item_list,
                         item_count, g_wit_surface_pct > 0 ? kind : nullptr, g_wit_surface_pct);
      const int grid = (int)(ns < g_wit_grid ? ns : g_wit_grid);
      hipLaunchKernelGGL((wit_sweep_kernel<DIM>), dim3(grid), dim3(WTHREADS), 0, st, pts, nodes, lv, verts, plane_tab, weights, k1,
                         R, ns, (float)g_wit_weight, 0.01f * (float)g_wit_cmax_pct, 0.01f * (float)g_wit_cmax_ext_pct, g_wit_min_bins, g_wit_flags, g_wit_max_open, (int)((int64_t)R * g_wit_max_live_pct / 100), (int)((int64_t)R * g_wit_max_in_pct / 100), g_wit_adaptive, g_wit_max_leaves < WLEAF ? g_wit_max_leaves : WLEAF, g_wit_max_eval, item_list, item_count, plan, queue, out, acc,
                         stats);
      return check_launch("wit_sweep");
    } else {
      return fail(FLOODER_E_ARG, "flooder_sweep_witness_f32: only dim 2 and 3");
    }
  }
};

}  // namespace

namespace flooder {

// flooder_sweep_witness_f32 plus the density grid of the index (NULL: none): with it the sweep stands back on a cloud
// that lies on a surface ("wit_surface_pct").  The parameter-block form passes the grid it holds anyway.
int sweep_witness(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes, const float* verts,
                  const float* weights, int k1, int R, int64_t n_simplices, const int32_t* coarse_rows, int n_coarse,
                  const uint32_t* parents, int32_t* queue, uint32_t* d2_scratch, const uint32_t* memb, int n_faces,
                  uint32_t* face_bits, const int32_t* face_slot, int32_t* flag_list, int32_t* flag_count,
                  uint32_t* flag_key, int32_t* flag_hist, uint64_t* top, int32_t* top_list, int32_t* top_count,
                  float* simplex_weight, int32_t* item_list, float* plane_scratch, uint64_t* stats,
                  const int32_t* density_grid, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !coarse_rows || !parents || !queue || !d2_scratch || !memb ||
      !face_bits || !flag_list || !flag_count || !simplex_weight || !item_list || !plane_scratch || n_pts < 1 || k1 < 1 ||
      k1 > FLOODER_MAX_VERTS || R < 1 || R > WROWS || n_coarse < 1 || n_coarse > WCOARSE || n_faces < 1 || n_faces > 32 ||
      (top && (!top_list || !top_count)) || (flag_key && !flag_hist) || n_simplices > 0x7fffffffLL)
    return fail(FLOODER_E_ARG, "flooder_sweep_witness_f32: bad argument");
  if (dim != 2 && dim != 3) return fail(FLOODER_E_ARG, "flooder_sweep_witness_f32: only dim 2 and 3");
  if (n_simplices * (int64_t)((R + 63) / 64) > 0x7fffffffLL)
    return fail(FLOODER_E_ARG, "flooder_sweep_witness_f32: too many (simplex, tile) pairs");
  const Levels lv = make_levels(n_pts);
  if ((n_pts + FLOODER_BVH_LEAF) * (int64_t)(padded_dim(dim) * sizeof(float)) >= (1LL << 32) ||
      total_nodes(lv) * (int64_t)(2 * padded_dim(dim) * sizeof(float)) >= (1LL << 32))
    return fail(FLOODER_E_ARG, "flooder_sweep_witness_f32: cloud too large");
  FaceAcc acc{memb, face_bits, n_faces, reinterpret_cast<unsigned long long*>(top), top_list, top_count, face_slot,
              flag_key, flag_hist};
  return dispatch_dim<WitOp>(dim, pts_sorted, nodes, lv, verts, plane_scratch, weights, k1, R, n_simplices,
                             WitPlan{coarse_rows, parents, n_coarse}, queue, item_list, queue + FLOODER_QUEUE_WORDS - 1,
                             WitOut{d2_scratch, flag_list, flag_count, simplex_weight}, acc,
                             reinterpret_cast<unsigned long long*>(stats),
                             density_grid ? density_grid + (flooder_density_grid_words(dim) - KIND_WORDS) : nullptr,
                             (hipStream_t)stream);
}

}  // namespace flooder

extern "C" {

int flooder_wit_max_rows(void) { return WROWS; }
int flooder_wit_max_coarse(void) { return WCOARSE; }

int flooder_sweep_witness_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes, const float* verts,
                              const float* weights, int k1, int R, int64_t n_simplices, const int32_t* coarse_rows,
                              int n_coarse, const uint32_t* parents, int32_t* queue, uint32_t* d2_scratch,
                              const uint32_t* memb, int n_faces, uint32_t* face_bits, const int32_t* face_slot,
                              int32_t* flag_list, int32_t* flag_count, uint32_t* flag_key, int32_t* flag_hist,
                              uint64_t* top, int32_t* top_list, int32_t* top_count, float* simplex_weight,
                              int32_t* item_list, float* plane_scratch, uint64_t* stats, void* stream) {
  return sweep_witness(pts_sorted, n_pts, dim, nodes, verts, weights, k1, R, n_simplices, coarse_rows, n_coarse, parents,
                       queue, d2_scratch, memb, n_faces, face_bits, face_slot, flag_list, flag_count, flag_key, flag_hist,
                       top, top_list, top_count, simplex_weight, item_list, plane_scratch, stats, nullptr, stream);
}

}  // extern "C"
