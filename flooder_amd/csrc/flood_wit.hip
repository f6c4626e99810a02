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
int g_wit_weight = 1500;    // simplices with at most this many cloud points in their box (flooder_simplex_weight_f32) are tried
int g_wit_cmax_pct = 250;   // c_max in percent of the local point spacing
int g_wit_grid = 256 * 11;  // persistent one-wave workgroups
int g_wit_min_bins = 6;     // the stage must hold the points of at least this many of the 64 excess bins
}  // namespace flooder

namespace {

constexpr int WCAP = 480;      // points staged per item
constexpr int WLEAF = 1024;    // leaves gathered per item (16 K candidate points)
constexpr int WFRONT = 192;    // inner nodes per level of the gather
constexpr int WCOARSE = FLOODER_WIT_MAX_COARSE;  // coarse samples per item (4 per lane)
constexpr int WPEND = 128;     // queued live samples
constexpr int WROWS = FLOODER_WIT_MAX_ROWS;      // samples per simplex at most
constexpr int UNR = 4;         // candidate rows in flight per lane
constexpr int NBIN = 64;
constexpr int PLANE_ROW = 24;  // (layout of simplex_planes_kernel, flood_cell.hip)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

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

template <int DIM>
__global__ __launch_bounds__(64, 12) void wit_sweep_kernel(
    const float* __restrict__ pts, const float* __restrict__ nodes, Levels lv, const float* __restrict__ verts,
    const float* __restrict__ plane_tab, const float* __restrict__ weights, int k1, int R, int64_t n_simplices,
    float w_limit, float cmax_mult, int min_bins, WitPlan plan, int32_t* __restrict__ queue, WitOut out, FaceAcc acc,
    unsigned long long* __restrict__ stats) {
  constexpr int DP = padded_dim(DIM);
  __shared__ float4 s_pts[WCAP + 4];
  __shared__ int s_leaf[WLEAF];
  __shared__ int s_hist[NBIN];
  __shared__ uint32_t s_mf[32];
  __shared__ uint32_t s_unres[WROWS / 32];
  __shared__ uint32_t s_tkey[WROWS / 64];
  static_assert(2 * WFRONT * sizeof(int) <= WCAP * sizeof(float4), "the gather's frontier lives inside the empty stage");
  static_assert(WCOARSE * 3 * sizeof(float) + WPEND * 6 <= WLEAF * sizeof(int), "witness table + queue alias the leaf list");
  int* s_front = reinterpret_cast<int*>(s_pts);
  // (after staging the leaf list is dead: witnesses and the queue of live samples take its place)
  float* s_wit = reinterpret_cast<float*>(s_leaf);
  uint32_t* s_pub = reinterpret_cast<uint32_t*>(s_leaf) + WCOARSE * 3;
  uint16_t* s_prow = reinterpret_cast<uint16_t*>(s_pub + WPEND);
  const int lane = threadIdx.x;
  const int top = lv.n_levels - 1;
  const int tiles64 = (R + 63) >> 6;
  unsigned long long n_handled = 0, n_dense = 0, n_over = 0, n_staged = 0, n_ccert = 0, n_live = 0, n_rounds = 0,
                     n_unres = 0, n_flagged = 0, n_pairs = 0, n_heavy = 0, n_bins = 0;

  int q_shard = (int)(blockIdx.x % QSHARDS), q_tried = 0;
  for (;;) {
    const int64_t s = queue_pop(queue, q_shard, q_tried, n_simplices, lane);
    if (s < 0) break;
    const float w_s = out.weight[s];
    if (!(w_s <= w_limit) || w_s < 0.f) { ++n_heavy; continue; }
    const float* vs = verts + s * (int64_t)k1 * DIM;

    // ---- face planes of the simplex (table row written by simplex_planes_kernel)
    float pn[DIM + 1][DIM], po[DIM + 1], ps[DIM + 1], org[DIM];
    float sext;
    {
      const float* pt = plane_tab + s * PLANE_ROW;
      typename RowVec<4>::type t[PLANE_ROW / 4];
#pragma unroll
      for (int i = 0; i < PLANE_ROW / 4; ++i) t[i] = load_uniform_row<4>(pt + 4 * i);
      auto at = [&](int i) { return t[i >> 2][i & 3]; };
#pragma unroll
      for (int k = 0; k < DIM; ++k) org[k] = at(k);
      sext = at(3);
#pragma unroll
      for (int f = 0; f <= DIM; ++f) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) pn[f][k] = at(4 + 5 * f + k);
        po[f] = at(4 + 5 * f + 3);
        ps[f] = at(4 + 5 * f + 4) + 1e-6f;
      }
    }
    // ---- 1. region: box of the vertices, extents along the face normals
    float blo[DIM], bhi[DIM], slo[DIM + 1], shi[DIM + 1];
#pragma unroll
    for (int k = 0; k < DIM; ++k) { blo[k] = __builtin_inff(); bhi[k] = -__builtin_inff(); }
#pragma unroll
    for (int f = 0; f <= DIM; ++f) { slo[f] = __builtin_inff(); shi[f] = -__builtin_inff(); }
    float amax = 0.f;
    for (int j = 0; j < k1; ++j) {
      float v[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        v[k] = vs[j * DIM + k];
        blo[k] = __builtin_fminf(blo[k], v[k]);
        bhi[k] = __builtin_fmaxf(bhi[k], v[k]);
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v[k]));
      }
#pragma unroll
      for (int f = 0; f <= DIM; ++f) {
        float dd = -po[f];
#pragma unroll
        for (int k = 0; k < DIM; ++k) dd = __builtin_fmaf(pn[f][k], v[k] - org[k], dd);
        slo[f] = __builtin_fminf(slo[f], dd);
        shi[f] = __builtin_fmaxf(shi[f], dd);
      }
    }
    // (a sample is a rounded combination of the vertices: it may leave their box by a few ulps)
    const float epsb = 8.f * 1.1920929e-7f * amax;
    float ext = 0.f, vol = 1.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      ext = __builtin_fmaxf(ext, bhi[k] - blo[k]);
      vol *= (bhi[k] - blo[k]);
    }
    const float n0 = __builtin_fmaxf(w_s, 1.f);
    const float h = DIM == 3 ? cbrtf(vol / n0) : __builtin_sqrtf(vol / n0);
    float c_max = __builtin_fminf(cmax_mult * h, 0.6f * ext);
    if (!(c_max > 0.f) || !(c_max < 3.0e38f)) { ++n_over; continue; }

    // ---- gather: leaves of the box tree overlapping [qlo, qhi]; returns their number or -1 (overflow)
    float qlo[DIM], qhi[DIM];
    auto gather = [&]() -> int {
      constexpr int GB = 4;
      int* fa = s_front;
      int* fb = s_front + WFRONT;
      int na = 0, nb = 0, n_leaf = 0;
      bool over = false;
      auto test_children = [&](int lvl, const int (&grp)[GB], int ng, int* out_list, int& out_n, int cap) {
        const int lvl_count = (int)lv.count[lvl], lvl_off = (int)lv.off[lvl];
        bool hit[GB];
        float lo[GB][DP], hi[GB][DP];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int idx = grp[u] * FAN + lane;
          hit[u] = (u < ng) && (idx < lvl_count);
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
            if (out_n + cnt > cap) {
              over = true;
            } else {
              if (hit[u]) out_list[out_n + lane_rank(m)] = grp[u] * FAN + lane;
              out_n += cnt;
            }
          }
        }
        wave_lds_sync();
      };
      {
        const int g0_[GB] = {};
        if (top == 0) test_children(0, g0_, 1, s_leaf, n_leaf, WLEAF);
        else test_children(top, g0_, 1, fa, na, WFRONT);
      }
      for (int lvl = top; lvl >= 1 && !over; --lvl) {
        nb = 0;
        for (int f = 0; f < na && !over; f += GB) {
          int grp[GB];
          const int ng = na - f < GB ? na - f : GB;
#pragma unroll
          for (int u = 0; u < GB; ++u) grp[u] = wave_uniform(fa[f + u < na ? f + u : f]);
          if (lvl == 1) test_children(0, grp, ng, s_leaf, n_leaf, WLEAF);
          else test_children(lvl - 1, grp, ng, fb, nb, WFRONT);
        }
        int* t = fa; fa = fb; fb = t;
        na = nb;
      }
      return over ? -1 : n_leaf;
    };
    int n_leaves = -1;
    for (int att = 0; att < 3; ++att) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) { qlo[k] = blo[k] - epsb - c_max; qhi[k] = bhi[k] + epsb + c_max; }
      n_leaves = gather();
      if (n_leaves >= 0) break;
      c_max *= 0.5f;
    }
    if (n_leaves < 0) { ++n_over; continue; }

    // ---- excess of a point: the smallest c for which it counts as "within c of the simplex"
    float blo_e[DIM], bhi_e[DIM], slo_t[DIM + 1], shi_t[DIM + 1], inv_den[DIM + 1];
#pragma unroll
    for (int k = 0; k < DIM; ++k) { blo_e[k] = blo[k] - epsb; bhi_e[k] = bhi[k] + epsb; }
#pragma unroll
    for (int f = 0; f <= DIM; ++f) {
      const float tol = ps[f] * (sext + c_max);
      slo_t[f] = slo[f] - tol;
      shi_t[f] = shi[f] + tol;
      inv_den[f] = 1.f / (1.001f + ps[f]);
    }
    auto excess = [&](const float (&x)[DP]) -> float {
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
    };
    const float bin_scale = (float)NBIN / c_max;
    const int n_cand = n_leaves * LEAF;
    // ---- 2a. histogram of the excesses
    s_hist[lane] = 0;
    wave_lds_sync();
    for (int ib = 0; ib < n_cand; ib += 64 * UNR) {
      float x[UNR][DP];
      bool in[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = ib + u * 64 + lane;
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
    wave_lds_sync();
    int n_keep_bins, n_stage;
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
    if (n_keep_bins < min_bins || n_stage == 0) { ++n_dense; continue; }   // too dense for one stage (or nothing near)
    const float c_sel = (float)n_keep_bins / bin_scale;
    // ---- 2b. stage the points of the kept bins
    int n_st = 0;
    for (int ib = 0; ib < n_cand; ib += 64 * UNR) {
      float x[UNR][DP];
      bool in[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = ib + u * 64 + lane;
        in[u] = idx < n_cand;
        const uint32_t row = in[u] ? (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF) : 0u;
        load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const float eb = excess(x[u]) * bin_scale;
        const bool keep = in[u] && eb < (float)n_keep_bins;   // (the same test as the histogram's: bin < n_keep_bins)
        const unsigned long long m = __ballot(keep);
        if (keep) {
          float4 v;
          v.x = x[u][0];
          v.y = x[u][1];
          v.z = DIM > 2 ? x[u][DIM > 2 ? 2 : 0] : 0.f;
          v.w = 0.f;
          s_pts[n_st + lane_rank(m)] = v;
        }
        n_st += __popcll(m);
      }
    }
    if (lane < 4) s_pts[n_st + lane] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
    for (int i = lane; i < WROWS / 32; i += 64) s_unres[i] = 0u;
    for (int i = lane; i < WROWS / 64; i += 64) s_tkey[i] = 0u;
    wave_lds_sync();
    n_staged += (unsigned long long)n_st;
    n_bins += (unsigned long long)n_keep_bins;
    const int K = n_st;

    // ---- helpers: a sample from its weight row; its inner slack; its certification limit
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
    auto cert_limit = [&](const float (&p)[DIM]) -> float {
      float dl = __builtin_inff();
#pragma unroll
      for (int k = 0; k < DIM; ++k) dl = __builtin_fminf(dl, __builtin_fminf(p[k] - blo[k], bhi[k] - p[k]));
#pragma unroll
      for (int f = 0; f <= DIM; ++f) {
        if (po[f] < 1.0e37f) {  // (wave-uniform: the plane is in use)
          float dd = -po[f];
#pragma unroll
          for (int k = 0; k < DIM; ++k) dd = __builtin_fmaf(pn[f][k], p[k] - org[k], dd);
          dl = __builtin_fminf(dl, __builtin_fminf(shi[f] - dd, dd - slo[f]));
        }
      }
      const float rr = 0.999f * (c_sel + __builtin_fmaxf(dl, 0.f));
      return rr * rr;
    };
    // certified samples raise the running maxima of their faces (one integer atomic per face and call at most)
    auto deliver = [&](bool on, uint32_t mb, float val) {
      uint32_t um = wave_or_u32(on ? mb : 0u);
      while (um) {  // (wave-uniform)
        const int f = __builtin_ctz(um);
        um &= um - 1u;
        const uint32_t v = wave_max_u32((on && ((mb >> f) & 1u)) ? __float_as_uint(val) : 0u);
        if (v > s_mf[f]) {
          if (lane == 0) {
            atomicMax(&acc.face_bits[acc.slot_of(s, f)], v);
            s_mf[f] = v;
          }
        }
      }
      wave_lds_sync();
    };
    // running maxima of this simplex's faces as other waves have left them
    if (lane < acc.n_faces)
      s_mf[lane] = __hip_atomic_load(acc.face_bits + acc.slot_of(s, lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wave_lds_sync();

    // ---- 3. coarse samples (four per lane) against the stage, with witnesses
    {
      constexpr int CPL = WCOARSE / 64;
      float p[CPL][DIM], best[CPL];
      int wj[CPL], crow[CPL];
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = i * 64 + lane;
        crow[i] = c < plan.n_coarse ? plan.coarse_rows[c] : -1;
        make_sample(crow[i] < 0 ? 0 : crow[i], p[i]);
        best[i] = __builtin_inff();
        wj[i] = 0;
      }
      for (int j = 0; j < K; j += 4) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = s_pts[j + u];
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          float d[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float t0 = p[i][0] - x[u].x;
            float d2 = t0 * t0;
            t0 = p[i][1] - x[u].y;
            d2 = __builtin_fmaf(t0, t0, d2);
            if constexpr (DIM == 3) {
              t0 = p[i][2] - x[u].z;
              d2 = __builtin_fmaf(t0, t0, d2);
            }
            d[u] = d2;
          }
          const float m = __builtin_fminf(__builtin_fminf(d[0], d[1]), __builtin_fminf(d[2], d[3]));
          if (m < best[i]) { best[i] = m; wj[i] = j; }
        }
      }
      if (stats) n_pairs += (unsigned long long)K * CPL;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        // the witness: the first point of the winning group of four that attains the minimum
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = s_pts[wj[i] + u];
        float4 w = x[3];
#pragma unroll
        for (int u = 2; u >= 0; --u) {
          float t0 = p[i][0] - x[u].x;
          float d2 = t0 * t0;
          t0 = p[i][1] - x[u].y;
          d2 = __builtin_fmaf(t0, t0, d2);
          if constexpr (DIM == 3) {
            t0 = p[i][2] - x[u].z;
            d2 = __builtin_fmaf(t0, t0, d2);
          }
          if (d2 == best[i]) w = x[u];
        }
        const int c = i * 64 + lane;
        s_wit[3 * c + 0] = w.x;
        s_wit[3 * c + 1] = w.y;
        s_wit[3 * c + 2] = w.z;
      }
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const bool cert = crow[i] >= 0 && best[i] <= cert_limit(p[i]);
        if (stats) n_ccert += (unsigned long long)__popcll(__ballot(cert));
        deliver(cert, cert ? acc.memb[crow[i]] : 0u, best[i]);
      }
    }

    // ---- 4. all samples: bound from the witnesses of the nearest coarse samples; live ones are queued and
    // evaluated 64 at a time
    int n_pend = 0;
    auto run_round = [&]() {
      const int n_take = n_pend < 64 ? n_pend : 64;
      const bool mine = lane < n_take;
      const int r = mine ? (int)s_prow[lane] : 0;
      float best = mine ? __uint_as_float(s_pub[lane]) : 0.f;
      // what is left of the queue moves to its front
      const int n_rest = n_pend - n_take;
      uint32_t mv_ub = 0u;
      uint16_t mv_row = 0;
      if (lane < n_rest) { mv_ub = s_pub[64 + lane]; mv_row = s_prow[64 + lane]; }
      wave_lds_sync();
      if (lane < n_rest) { s_pub[lane] = mv_ub; s_prow[lane] = mv_row; }
      n_pend = n_rest;
      float p[DIM];
      make_sample(r, p);
      const uint32_t mb = mine ? acc.memb[r] : 0u;
      // still live?  (the maxima have risen since the sample was queued)
      uint32_t thr = 0xffffffffu;
      {
        uint32_t um = wave_or_u32(mb);
        while (um) {
          const int f = __builtin_ctz(um);
          um &= um - 1u;
          const uint32_t v = s_mf[f];
          if ((mb >> f) & 1u) thr = v < thr ? v : thr;
        }
      }
      const bool act = mine && __float_as_uint(best) > thr;
      if (__ballot(act) == 0ull) { wave_lds_sync(); return; }
      ++n_rounds;
      for (int j = 0; j < K; j += 4) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = s_pts[j + u];
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
      if (stats) n_pairs += (unsigned long long)K;
      const bool cert = act && best <= cert_limit(p);
      const bool unres = act && !cert;
      if (unres) {
        out.d2[s * (int64_t)R + r] = __float_as_uint(best);
        atomicOr(&s_unres[r >> 5], 1u << (r & 31));
        atomicMax(&s_tkey[r >> 6], __float_as_uint(best));
      }
      if (stats) n_unres += (unsigned long long)__popcll(__ballot(unres));
      deliver(cert, mb, best);
    };
    for (int g0 = 0; g0 < R; g0 += 64) {
      const int r = g0 + lane;
      const bool valid = r < R;
      const int rr = valid ? r : R - 1;
      float p[DIM];
      make_sample(rr, p);
      const uint32_t par = plan.parents[rr];
      const uint32_t mb = valid ? acc.memb[rr] : 0u;
      float ub = __builtin_inff();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (int)((par >> (8 * j)) & 0xffu);
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
      uint32_t thr = 0xffffffffu;
      {
        uint32_t um = wave_or_u32(mb);
        while (um) {  // (wave-uniform: the faces present in this group of 64 rows)
          const int f = __builtin_ctz(um);
          um &= um - 1u;
          const uint32_t v = s_mf[f];
          if ((mb >> f) & 1u) thr = v < thr ? v : thr;
        }
      }
      const bool live = valid && __float_as_uint(ub) > thr;
      const unsigned long long m = __ballot(live);
      if (m != 0ull) {
        if (live) {
          const int pos = n_pend + lane_rank(m);
          s_pub[pos] = __float_as_uint(ub);
          s_prow[pos] = (uint16_t)r;
        }
        n_pend += __popcll(m);
        if (stats) n_live += (unsigned long long)__popcll(m);
        wave_lds_sync();
        if (n_pend >= 64) run_round();
      }
    }
    while (n_pend > 0) run_round();

    // ---- tiles with unresolved samples go to the exact finish: the other rows of such a tile are marked settled
    for (int t0 = 0; t0 < tiles64; t0 += 64) {
      const int t = t0 + lane;
      const uint32_t key = t < tiles64 ? s_tkey[t] : 0u;
      const unsigned long long fm = __ballot(key != 0u);
      if (fm == 0ull) continue;
      const int nf = __popcll(fm);
      int base = 0;
      if (lane == 0) base = atomicAdd(out.flag_count, nf);
      base = wave_uniform(base);
      if (key != 0u) {
        const int pos = base + lane_rank(fm);
        const int item = (int)(s * tiles64 + t);
        out.flag_list[pos] = item;
        if (acc.flag_key) {
          acc.flag_key[pos] = key;
          atomicAdd(&acc.flag_hist[key >> 19], 1);
        }
        if (acc.top) {
          const unsigned long long old = atomicMax(&acc.top[s], ((unsigned long long)key << 32) | (unsigned long long)(uint32_t)item);
          if (old == 0ull) acc.top_list[atomicAdd(acc.top_count, 1)] = (int)s;
        }
      }
      n_flagged += (unsigned long long)nf;
      unsigned long long rest = fm;
      while (rest) {  // (wave-uniform)
        const int tl = __builtin_ctzll(rest);
        rest &= rest - 1ull;
        const int r = (t0 + tl) * 64 + lane;
        if (r < R && !((s_unres[r >> 5] >> (r & 31)) & 1u)) out.d2[s * (int64_t)R + r] = SETTLED_BIT;
      }
    }
    if (lane == 0) out.weight[s] = -1.f;
    ++n_handled;
    wave_lds_sync();
  }
  if (stats) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      n_pairs += __shfl_xor(n_pairs, o);
    }
    if (lane == 0) {
      atomicAdd(&stats[0], n_handled);
      atomicAdd(&stats[1], n_heavy);
      atomicAdd(&stats[2], n_over);
      atomicAdd(&stats[3], n_dense);
      atomicAdd(&stats[4], n_staged);
      atomicAdd(&stats[5], n_ccert);
      atomicAdd(&stats[6], n_live);
      atomicAdd(&stats[7], n_rounds);
      atomicAdd(&stats[8], n_unres);
      atomicAdd(&stats[9], n_flagged);
      atomicAdd(&stats[10], n_pairs);
      atomicAdd(&stats[11], n_bins);
    }
  }
}

template <int DIM>
struct WitOp {
  static int run(const float* pts, const float* nodes, const Levels& lv, const float* verts, float* plane_tab,
                 const float* weights, int k1, int R, int64_t ns, WitPlan plan, int32_t* queue, WitOut out, FaceAcc acc,
                 unsigned long long* stats, hipStream_t st) {
    if constexpr (DIM == 2 || DIM == 3) {
      const int rc = launch_simplex_planes(DIM, verts, k1, ns, plane_tab, st);
      if (rc != FLOODER_OK) return rc;
      const int grid = (int)(ns < g_wit_grid ? ns : g_wit_grid);
      hipLaunchKernelGGL((wit_sweep_kernel<DIM>), dim3(grid), dim3(64), 0, st, pts, nodes, lv, verts, plane_tab, weights, k1,
                         R, ns, (float)g_wit_weight, 0.01f * (float)g_wit_cmax_pct, g_wit_min_bins, plan, queue, out, acc,
                         stats);
      return check_launch("wit_sweep");
    } else {
      return fail(FLOODER_E_ARG, "flooder_sweep_witness_f32: only dim 2 and 3");
    }
  }
};

}  // namespace

extern "C" {

int flooder_wit_max_rows(void) { return WROWS; }
int flooder_wit_max_coarse(void) { return WCOARSE; }

int flooder_sweep_witness_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes, const float* verts,
                              const float* weights, int k1, int R, int64_t n_simplices, const int32_t* coarse_rows,
                              int n_coarse, const uint32_t* parents, int32_t* queue, uint32_t* d2_scratch,
                              const uint32_t* memb, int n_faces, uint32_t* face_bits, const int32_t* face_slot,
                              int32_t* flag_list, int32_t* flag_count, uint32_t* flag_key, int32_t* flag_hist,
                              uint64_t* top, int32_t* top_list, int32_t* top_count, float* simplex_weight,
                              float* plane_scratch, uint64_t* stats, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !coarse_rows || !parents || !queue || !d2_scratch || !memb ||
      !face_bits || !flag_list || !flag_count || !simplex_weight || !plane_scratch || n_pts < 1 || k1 < 1 ||
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
                             WitPlan{coarse_rows, parents, n_coarse}, queue,
                             WitOut{d2_scratch, flag_list, flag_count, simplex_weight}, acc,
                             reinterpret_cast<unsigned long long*>(stats), (hipStream_t)stream);
}

}  // extern "C"
