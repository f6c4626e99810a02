// flood_cell.hip - coverage sweep through a per-simplex cell grid in LDS (gfx950; dim 2 and 3).
//
// One workgroup owns one simplex at a time:
//   1. gather   the box tree over the Morton-sorted cloud is walked breadth-first (lane = child box)
//               for the leaves that overlap the simplex's bounding box grown by c (the cell size);
//   2. stage    their points are filtered (box grown by c, and within c of every face plane of the
//               simplex), counting-sorted by cell into LDS (<= 3072 points, 16 B each);
//   3. query    every thread takes samples of the simplex (rebuilt in registers from vertices x
//               weights), visits the 3^dim cells around the sample - 3^(dim-1) contiguous runs of the
//               staged list - and keeps the minimum direct-difference squared distance;
//   4. verify   a sample is final when its minimum is <= (0.999 c)^2: every point that close lies in
//               the visited cells.  Tiles (64 consecutive samples) with an unverified sample, and
//               simplices whose region does not fit (too many leaves / points), are appended to a
//               work list that the tree sweep (flood_bvh.hip, seeded with the minima found here)
//               finishes exactly.
// c = max(rho, extent / 13) (3D) where rho is the caller's estimate of the largest nearest-neighbour
// distance over the simplex (from a probe sweep); a bad estimate costs time, never correctness.

#include "flood_common.hpp"
#include "flood_bvh.hpp"

using namespace flooder;

namespace {

constexpr int CELL_THREADS = 256;
constexpr int MAX_LEAVES = 1024;   // leaves gathered per simplex (16 K points before filtering)
constexpr int MAX_FRONT = 256;     // inner nodes per level of the gather
constexpr int CAP = 3072;          // points staged in LDS per simplex

template <int DIM>
struct CellCfg {
  static constexpr int G = DIM == 2 ? 64 : 16;                 // cells per axis
  static constexpr int NC = DIM == 2 ? 64 * 64 : 16 * 16 * 16;  // cells in the grid
};

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

template <int DIM>
__global__ __launch_bounds__(CELL_THREADS) void cell_sweep_kernel(
    const float* __restrict__ pts, const float* __restrict__ nodes, Levels lv,
    const float* __restrict__ verts, const float* __restrict__ weights, int k1, int R,
    int64_t n_simplices, const float* __restrict__ rho, int32_t* __restrict__ queue,
    uint32_t* __restrict__ out_d2, int32_t* __restrict__ flag_list, int32_t* __restrict__ flag_count,
    unsigned long long* __restrict__ stats) {
  constexpr int DP = padded_dim(DIM);
  constexpr int G = CellCfg<DIM>::G;
  constexpr int NC = CellCfg<DIM>::NC;
  __shared__ float4 s_pts[CAP];
  __shared__ int s_cell[NC + 4];      // [i+1]: count -> start -> end of cell i (see below)
  __shared__ int s_leaf[MAX_LEAVES];
  __shared__ int s_front[2][MAX_FRONT];
  __shared__ int s_part[CELL_THREADS];
  __shared__ int s_n[8];              // 0,1: frontier sizes  2: leaves  3: overflow  4: simplex id
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int top = lv.n_levels - 1;
  const int tiles64 = (R + 63) >> 6;
  unsigned long long n_pairs = 0, n_cand = 0, n_fallback = 0;

  for (;;) {
    __syncthreads();
    if (tid == 0) {
      s_n[4] = atomicAdd(queue, 1);
      s_n[0] = 0; s_n[1] = 0; s_n[2] = 0; s_n[3] = 0;
    }
    __syncthreads();
    const int64_t s = s_n[4];
    if (s >= n_simplices) break;

    // ---- simplex geometry (wave-uniform): box, cell size, grid, face planes
    const float* vs = verts + s * (int64_t)k1 * DIM;
    float blo[DIM], bhi[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) { blo[k] = vs[k]; bhi[k] = vs[k]; }
    for (int j = 1; j < k1; ++j)
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        blo[k] = __builtin_fminf(blo[k], vs[j * DIM + k]);
        bhi[k] = __builtin_fmaxf(bhi[k], vs[j * DIM + k]);
      }
    float ext = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) ext = __builtin_fmaxf(ext, bhi[k] - blo[k]);
    float c = __builtin_fmaxf(rho[s], ext / (float)(G - 3));
    if (!(c > 0.f) || !(c < 3.0e38f)) c = 1.f;  // degenerate input: everything goes to the fallback
    const float inv_c = 1.f / c;
    int nc[DIM];
    float g0[DIM], qlo[DIM], qhi[DIM];
    int ncells = 1;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      g0[k] = blo[k] - c;
      qlo[k] = blo[k] - c;
      qhi[k] = bhi[k] + c;
      int n = (int)((bhi[k] + c - g0[k]) * inv_c) + 1;
      n = n < 3 ? 3 : (n > G ? G : n);
      nc[k] = n;
      ncells *= n;
    }
    // outward unit normals of the faces of a full-dimensional simplex (face f is opposite vertex f)
    float pn[DIM + 1][DIM], po[DIM + 1];
#pragma unroll
    for (int f = 0; f <= DIM; ++f) {
      po[f] = 3.0e38f;  // disabled plane: the test below always passes
#pragma unroll
      for (int k = 0; k < DIM; ++k) pn[f][k] = 0.f;
    }
    if (k1 == DIM + 1) {
#pragma unroll
      for (int f = 0; f <= DIM; ++f) {
        // vertices of face f: all but f; a = first of them, opp = vertex f
        int id[DIM];
        int q = 0;
#pragma unroll
        for (int j = 0; j <= DIM; ++j)
          if (j != f) id[q++] = j;
        float nrm[DIM];
        if constexpr (DIM == 3) {
          float e1[3], e2[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            e1[k] = vs[id[1] * 3 + k] - vs[id[0] * 3 + k];
            e2[k] = vs[id[2] * 3 + k] - vs[id[0] * 3 + k];
          }
          nrm[0] = e1[1] * e2[2] - e1[2] * e2[1];
          nrm[1] = e1[2] * e2[0] - e1[0] * e2[2];
          nrm[2] = e1[0] * e2[1] - e1[1] * e2[0];
        } else {
          const float ex = vs[id[1] * 2 + 0] - vs[id[0] * 2 + 0];
          const float ey = vs[id[1] * 2 + 1] - vs[id[0] * 2 + 1];
          nrm[0] = ey;
          nrm[1] = -ex;
        }
        float len2 = 0.f, side = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          len2 = __builtin_fmaf(nrm[k], nrm[k], len2);
          side = __builtin_fmaf(nrm[k], vs[f * DIM + k] - vs[id[0] * DIM + k], side);
        }
        const bool ok = len2 > 1e-30f;
        const float sc = ok ? (side > 0.f ? -1.f : 1.f) / __builtin_sqrtf(len2) : 0.f;
        float off = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          pn[f][k] = nrm[k] * sc;
          off = __builtin_fmaf(pn[f][k], vs[id[0] * DIM + k], off);
        }
        po[f] = ok ? off : 3.0e38f;  // degenerate face: plane test always passes
      }
    }
    const float plane_tol = c * 1.001f + 1e-6f * ext;

    auto cell_of = [&](const float (&x)[DP]) {
      int id = 0;
#pragma unroll
      for (int k = DIM - 1; k >= 0; --k) {
        int ck = (int)((x[k] - g0[k]) * inv_c);
        ck = ck < 0 ? 0 : (ck >= nc[k] ? nc[k] - 1 : ck);
        id = id * nc[k] + ck;
      }
      return id;
    };
    auto keep_point = [&](const float (&x)[DP]) {
      bool in = true;
#pragma unroll
      for (int k = 0; k < DIM; ++k) in = in && (x[k] >= qlo[k]) && (x[k] <= qhi[k]);
#pragma unroll
      for (int f = 0; f <= DIM; ++f) {
        float dd = -po[f];
#pragma unroll
        for (int k = 0; k < DIM; ++k) dd = __builtin_fmaf(pn[f][k], x[k], dd);
        in = in && (dd <= plane_tol);
      }
      return in;
    };

    // ---- 1. gather the leaves whose box overlaps [qlo, qhi] (breadth-first, one frontier per level)
    for (int i = tid; i < NC + 4; i += CELL_THREADS) s_cell[i] = 0;
    auto test_children = [&](int lvl, int64_t grp, int* out_list, int* out_count, int cap) {
      const int64_t idx = grp * FAN + lane;
      bool hit = false;
      if (idx < lv.count[lvl]) {
        float lo[DP], hi[DP];
        const float* nb = nodes + (lv.off[lvl] + idx) * 2 * DP;
        load_row<DP>(nb, lo);
        load_row<DP>(nb + DP, hi);
        hit = true;
#pragma unroll
        for (int k = 0; k < DIM; ++k) hit = hit && (lo[k] <= qhi[k]) && (hi[k] >= qlo[k]);
      }
      const unsigned long long m = __ballot(hit);
      if (m) {
        const int cnt = __popcll(m);
        int base = 0;
        if (lane == 0) base = atomicAdd(out_count, cnt);
        base = wave_uniform(base);
        if (base + cnt > cap) {
          if (lane == 0) s_n[3] = 1;  // overflow
        } else if (hit) {
          const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
          out_list[base + rank] = (int)idx;
        }
      }
    };
    __syncthreads();
    int cur = 0;
    if (wv == 0) {
      if (top == 0) test_children(0, 0, s_leaf, &s_n[2], MAX_LEAVES);
      else test_children(top, 0, s_front[0], &s_n[0], MAX_FRONT);
    }
    __syncthreads();
    for (int lvl = top; lvl >= 1; --lvl) {
      const int nf = s_n[cur];
      int* out_list = (lvl == 1) ? s_leaf : s_front[cur ^ 1];
      int* out_count = (lvl == 1) ? &s_n[2] : &s_n[cur ^ 1];
      const int cap = (lvl == 1) ? MAX_LEAVES : MAX_FRONT;
      for (int f = wv; f < nf; f += 4) test_children(lvl - 1, s_front[cur][f], out_list, out_count, cap);
      __syncthreads();
      if (tid == 0) s_n[cur] = 0;
      cur ^= 1;
      __syncthreads();
    }
    bool overflow = s_n[3] != 0;
    const int n_leaves = overflow ? 0 : s_n[2];

    // ---- 2a. count candidates per cell
    for (int idx = tid; idx < n_leaves * LEAF; idx += CELL_THREADS) {
      const int64_t row = (int64_t)s_leaf[idx / LEAF] * LEAF + (idx % LEAF);
      float x[DP];
      load_row<DP>(pts + row * DP, x);
      if (keep_point(x)) atomicAdd(&s_cell[cell_of(x) + 1], 1);
    }
    __syncthreads();
    // ---- 2b. exclusive prefix over the cells: s_cell[i+1] = start of cell i
    const int chunk = (ncells + CELL_THREADS - 1) / CELL_THREADS;
    {
      int sum = 0;
      for (int i = 0; i < chunk; ++i) {
        const int cidx = tid * chunk + i;
        if (cidx < ncells) sum += s_cell[cidx + 1];
      }
      s_part[tid] = sum;
    }
    __syncthreads();
    if (wv == 0) {
      int v[4], tot = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = s_part[lane * 4 + i]; tot += v[i]; }
      const int incl = wave_incl_scan(tot, lane);
      int run = incl - tot;
#pragma unroll
      for (int i = 0; i < 4; ++i) { s_part[lane * 4 + i] = run; run += v[i]; }
      if (lane == 63) s_n[5] = incl;  // total candidates
    }
    __syncthreads();
    const int total = s_n[5];
    if (total > CAP) overflow = true;
    {
      int run = s_part[tid];
      for (int i = 0; i < chunk; ++i) {
        const int cidx = tid * chunk + i;
        if (cidx < ncells) {
          const int cnt = s_cell[cidx + 1];
          s_cell[cidx + 1] = run;
          run += cnt;
        }
      }
    }
    __syncthreads();

    if (overflow) {
      // region too large for the LDS stage: the whole simplex goes to the exact tree sweep
      for (int r = tid; r < R; r += CELL_THREADS) out_d2[s * (int64_t)R + r] = INF_BITS;
      for (int t = tid; t < tiles64; t += CELL_THREADS) {
        const int pos = atomicAdd(flag_count, 1);
        flag_list[pos] = (int)(s * tiles64 + t);
      }
      n_fallback += (tid < tiles64) ? 1 : 0;
      continue;
    }

    // ---- 2c. scatter into the cell-sorted LDS list; afterwards s_cell[i] = begin, s_cell[i+1] = end
    for (int idx = tid; idx < n_leaves * LEAF; idx += CELL_THREADS) {
      const int64_t row = (int64_t)s_leaf[idx / LEAF] * LEAF + (idx % LEAF);
      float x[DP];
      load_row<DP>(pts + row * DP, x);
      if (keep_point(x)) {
        const int pos = atomicAdd(&s_cell[cell_of(x) + 1], 1);
        float4 v;
        v.x = x[0];
        v.y = x[1];
        v.z = DIM > 2 ? x[DIM > 2 ? 2 : 0] : 0.f;
        v.w = 0.f;
        s_pts[pos] = v;
      }
    }
    __syncthreads();
    if (tid == 0) n_cand += (unsigned long long)total;

    // ---- 3. query: one sample per thread and iteration
    const float c_ok = (0.999f * c) * (0.999f * c);
    for (int r0 = 0; r0 < R; r0 += CELL_THREADS) {
      const int r = r0 + tid;
      const bool live = r < R;
      const int rr = live ? r : R - 1;
      float p[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[k] = 0.f;
      for (int j = 0; j < k1; ++j) {
        const float w = weights[(int64_t)rr * k1 + j];
#pragma unroll
        for (int k = 0; k < DIM; ++k) p[k] = __builtin_fmaf(w, vs[j * DIM + k], p[k]);
      }
      int ck[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        int t = (int)((p[k] - g0[k]) * inv_c);
        ck[k] = t < 1 ? 1 : (t > nc[k] - 2 ? nc[k] - 2 : t);
      }
      float best = __builtin_inff();
      constexpr int NROW = DIM == 3 ? 9 : 3;
#pragma unroll
      for (int rw = 0; rw < NROW; ++rw) {
        int base;
        if constexpr (DIM == 3) {
          const int dz = rw / 3 - 1, dy = rw % 3 - 1;
          base = ((ck[2] + dz) * nc[1] + (ck[1] + dy)) * nc[0] + ck[0] - 1;
        } else {
          const int dy = rw - 1;
          base = (ck[1] + dy) * nc[0] + ck[0] - 1;
        }
        const int b = s_cell[base];
        const int e = s_cell[base + 3];
        n_pairs += (unsigned long long)(e - b);
        for (int i = b; i < e; ++i) {
          const float4 q = s_pts[i];
          float t0 = p[0] - q.x;
          float d2 = t0 * t0;
          t0 = p[1] - q.y;
          d2 = __builtin_fmaf(t0, t0, d2);
          if constexpr (DIM == 3) {
            t0 = p[2] - q.z;
            d2 = __builtin_fmaf(t0, t0, d2);
          }
          best = __builtin_fminf(best, d2);
        }
      }
      if (live) out_d2[s * (int64_t)R + r] = __float_as_uint(best);
      const bool unresolved = live && !(best <= c_ok);
      if (__ballot(unresolved) != 0ull) {
        if (lane == 0) {
          const int pos = atomicAdd(flag_count, 1);
          flag_list[pos] = (int)(s * tiles64 + ((r0 + wv * 64) >> 6));
        }
        n_fallback += (lane == 0) ? 1 : 0;
      }
    }
  }
  if (stats) {
    // per-wave totals -> global (pairs are per-thread counts)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_pairs += __shfl_xor(n_pairs, o);
    if (lane == 0) atomicAdd(&stats[0], n_pairs);
    if (tid == 0) atomicAdd(&stats[1], n_cand);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_fallback += __shfl_xor(n_fallback, o);
    if (lane == 0) atomicAdd(&stats[2], n_fallback);
  }
}

template <int DIM>
struct CellOp {
  static int run(const float* pts, const float* nodes, const Levels& lv, const float* verts,
                 const float* weights, int k1, int R, int64_t ns, const float* rho, int32_t* queue,
                 uint32_t* out, int32_t* flag_list, int32_t* flag_count, unsigned long long* stats,
                 hipStream_t st) {
    if constexpr (DIM == 2 || DIM == 3) {
      int64_t grid = ns < 256 * 2 ? ns : 256 * 2;  // persistent: 2 workgroups per CU (LDS-limited)
      if (grid < 1) grid = 1;
      hipLaunchKernelGGL((cell_sweep_kernel<DIM>), dim3((int)grid), dim3(CELL_THREADS), 0, st, pts, nodes, lv,
                         verts, weights, k1, R, ns, rho, queue, out, flag_list, flag_count, stats);
      return check_launch("cell_sweep");
    } else {
      return fail(FLOODER_E_ARG, "flooder_sweep_cell_f32: only dim 2 and 3");
    }
  }
};

}  // namespace

extern "C" {

int flooder_sweep_cell_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                           const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                           const float* rho, int32_t* queue, uint32_t* out_d2, int32_t* flag_list,
                           int32_t* flag_count, uint64_t* stats, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !rho || !queue || !out_d2 || !flag_list ||
      !flag_count || n_pts < 1 || k1 < 1 || k1 > FLOODER_MAX_VERTS || R < 0)
    return fail(FLOODER_E_ARG, "flooder_sweep_cell_f32: bad argument");
  if (dim != 2 && dim != 3) return fail(FLOODER_E_ARG, "flooder_sweep_cell_f32: only dim 2 and 3");
  if (n_simplices * (int64_t)((R + 63) / 64) > 0x7fffffffLL)
    return fail(FLOODER_E_ARG, "flooder_sweep_cell_f32: too many (simplex, tile) pairs");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<CellOp>(dim, pts_sorted, nodes, lv, verts, weights, k1, R, n_simplices, rho, queue,
                              out_d2, flag_list, flag_count, reinterpret_cast<unsigned long long*>(stats),
                              (hipStream_t)stream);
}

}  // extern "C"
