// flood_bvh.hip - coverage sweep with wave-cooperative hierarchical culling (gfx950).
//
// Same result as the brute-force sweep (min over ALL points of the direct-difference squared
// distance, i.e. the exact nearest neighbour the reference's CPU path returns, core.py:197-199), but
// a tile of samples only evaluates points whose bounding box can still beat its running minima.
//
//   morton          64-bit Morton code of every point (dim x floor(63/dim) bits); the host sorts by it.
//   bvh_build       implicit tree over the Morton-sorted cloud: leaf = 16 consecutive points,
//                   fan-out 64 (one child per lane), node = axis-aligned box (lo[DP], hi[DP]).
//   sweep_bvh       one wave owns (simplex, tile of 64*KS samples).  Samples live in registers
//                   (rebuilt from vertices x barycentric weights).  Traversal is wave-uniform and
//                   nearest-first: at a node the 64 lanes each test one child box against the tile's
//                   bounding box (lower bound lb), the closest unvisited child is taken next, and a
//                   level is abandoned as soon as its closest child has lb >= M, M = the largest
//                   running minimum in the tile.  At a leaf every lane first checks its own samples
//                   against the leaf box; if no lane can improve the leaf is skipped, otherwise its 16
//                   points stream through SGPRs (scalar loads) against all samples of the tile -
//                   the same inner loop as the brute-force sweep.
//
// Bounds are computed in fp32 and compared with a 1e-5 relative safety margin, so culling never
// removes a pair that could change a minimum: results are bit-identical to the exhaustive sweep.

#include "flood_common.hpp"
#include "flood_bvh.hpp"

using namespace flooder;

namespace {

constexpr float SAFE = 0.99999f;

}  // namespace
namespace flooder { int g_bvh_ks = 0; int g_bvh_subs = 16; int g_bvh_grid = 256 * 4; int g_cell_grid = 256 * 4; int g_cell_exh_dense = 64 * 512; int g_bvh_leaf_batch = 1; int g_bvh_refine_pct = 100; int g_cell_exh_sparse = 4 * 480; int g_cell_brute_max = 160; int g_finish_focus_pct = 99; int g_finish_refresh = 16; int g_curve_bits = 0; int g_cell_super_weight = 2000; int g_cell_super_n0 = 480; int g_cell_super_sparse = 600; int g_cell_super_min_chunks = 49152; int g_cell_tries = 2; int g_cell_exh_tries = 3; int g_finish_items_cap = 65536; int g_curve = 1; int g_finish_budget = 14; int g_finish_budget_min = 64; int g_finish_wide_points = 4 << 20; int g_cell_surface_pct = 60; int g_cell_split_launches = 1; int g_finish_order = 1; int g_cell_retry_pct = 50; int g_cell_retry_keep = 200; int g_finish_top = 0; int g_cell_tiles = 0; int g_cell_density_grid = 16; int g_cell_chunks_per_block = 12; int g_cell_min_grid = 384; int g_cell_weight_classes = 1; int g_cell_listed_first = 1; int g_cell_tail_waves = 200; int g_cell_one_pass = 125; int g_cell_chunk_major = 1; int g_cell_chunk_major_max = 262144; int g_cell_drop = 1; int g_cell_queue_block = 5; }  // 0 = by R; else samples per lane (1, 2, 4, 8)
namespace {

// ------------------------------------------------------------------------------------ morton
// bits per axis of the curve codes: option "curve_bits"; default 8 in 3D and 12 in 2D - 24-bit keys, written and
// sorted as uint32 words in three 8-bit passes of the radix sort (a finer curve than the 256^3 grid does not make
// the 16-point leaves measurably tighter, not even for 16 M points: sweeps unchanged, a pass saved) -, 12 in
// higher dimensions; at most floor(63 / dim) and 21
inline int curve_bits_per_axis(int dim) {
  int cap = 63 / dim;
  if (cap > 21) cap = 21;
  int b = g_curve_bits > 0 ? g_curve_bits : (dim == 3 ? 8 : 12);
  if (dim == 1) b = cap;          // (one axis: the code is the coordinate; keep its full resolution)
  return b < cap ? b : cap;
}

struct Box {
  float lo[FLOODER_MAX_DIM];
  float scale[FLOODER_MAX_DIM];
};

// ------------------------------------------------------------------------------------ bounding box
// Stage 1: per-block partial (lo[8], hi[8]); stage 2: one block folds the partials into box[0:dim] = lo,
// box[8:8+dim] = hi.  No atomics, no host round trip.
template <int DIM>
__global__ __launch_bounds__(256) void bbox_partial_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                           float* __restrict__ partial) {
  float lo[DIM], hi[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const float v = pts[j * ld + k];
      lo[k] = __builtin_fminf(lo[k], v);
      hi[k] = __builtin_fmaxf(hi[k], v);
    }
  }
  __shared__ float s_lo[4][DIM], s_hi[4][DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    const float a = wave_min_f32(lo[k]);
    const float b = wave_max_f32(hi[k]);
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][k] = a; s_hi[threadIdx.x >> 6][k] = b; }
  }
  __syncthreads();
  if (threadIdx.x < DIM) {
    const int k = threadIdx.x;
    float a = s_lo[0][k], b = s_hi[0][k];
    for (int w = 1; w < 4; ++w) { a = __builtin_fminf(a, s_lo[w][k]); b = __builtin_fmaxf(b, s_hi[w][k]); }
    partial[blockIdx.x * 16 + k] = a;
    partial[blockIdx.x * 16 + 8 + k] = b;
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void bbox_final_kernel(const float* __restrict__ partial, int n_partial,
                                                         float* __restrict__ box) {
  // all partials in flight at once (256 threads x 16 floats), then wave + block reduction
  float lo[DIM], hi[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
  for (int i = threadIdx.x; i < n_partial; i += 256) {
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      lo[k] = __builtin_fminf(lo[k], partial[i * 16 + k]);
      hi[k] = __builtin_fmaxf(hi[k], partial[i * 16 + 8 + k]);
    }
  }
  __shared__ float s_lo[4][DIM], s_hi[4][DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    const float a = wave_min_f32(lo[k]);
    const float b = wave_max_f32(hi[k]);
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][k] = a; s_hi[threadIdx.x >> 6][k] = b; }
  }
  __syncthreads();
  if (threadIdx.x < DIM) {
    const int k = threadIdx.x;
    float a = s_lo[0][k], b = s_hi[0][k];
    for (int w = 1; w < 4; ++w) { a = __builtin_fminf(a, s_lo[w][k]); b = __builtin_fmaxf(b, s_hi[w][k]); }
    box[k] = a;
    box[8 + k] = b;
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void morton_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                     const float* __restrict__ dbox, int64_t* __restrict__ codes,
                                                     int curve, int BITS, int narrow, int32_t* __restrict__ zero_buf = nullptr,
                                                     int64_t zero_words = 0) {
  // (rides along: a buffer the NEXT kernels of the index build add into - the density grid - is zeroed here instead of
  // by a fill launch of its own, 5 us on the critical path of a 157 us build)
  if (zero_buf != nullptr)
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < zero_words; j += (int64_t)gridDim.x * blockDim.x)
      zero_buf[j] = 0;
  // curve 0: Morton (Z-order) codes; 1: Hilbert codes (Skilling's axes-to-transpose transform, then the same
  // bit interleave) - consecutive codes are neighbours in space, so 16 consecutive points make tighter leaves.
  // BITS per axis (flooder_curve_key_bits(dim) / dim): the radix sort of the codes costs one pass per 8 key bits
  Box box;
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    const float ext = dbox[8 + k] - dbox[k];
    box.lo[k] = dbox[k];
    box.scale[k] = ext > 0.f ? (float)((1u << BITS) - 1u) / ext : 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    uint32_t q[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      float t = (pts[j * ld + k] - box.lo[k]) * box.scale[k];
      t = t < 0.f ? 0.f : t;
      const float top = (float)((1u << BITS) - 1u);
      t = t > top ? top : t;
      q[k] = (uint32_t)t;
    }
    uint64_t code = 0;
    if (curve == 1 && DIM > 1) {
      const uint32_t MTOP = 1u << (BITS - 1);
      for (uint32_t Q = MTOP; Q > 1u; Q >>= 1) {
        const uint32_t P = Q - 1u;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          if (q[k] & Q) {
            q[0] ^= P;
          } else {
            const uint32_t t = (q[0] ^ q[k]) & P;
            q[0] ^= t;
            q[k] ^= t;
          }
        }
      }
#pragma unroll
      for (int k = 1; k < DIM; ++k) q[k] ^= q[k - 1];
      uint32_t t = 0u;
      for (uint32_t Q = MTOP; Q > 1u; Q >>= 1)
        if (q[DIM - 1] & Q) t ^= Q - 1u;
#pragma unroll
      for (int k = 0; k < DIM; ++k) q[k] ^= t;
      for (int b = BITS - 1; b >= 0; --b)
#pragma unroll
        for (int k = 0; k < DIM; ++k) code = (code << 1) | (uint64_t)((q[k] >> b) & 1u);
    } else {
#pragma unroll
      for (int b = 0; b < BITS; ++b)
#pragma unroll
        for (int k = 0; k < DIM; ++k) code |= (uint64_t)((q[k] >> b) & 1u) << (b * DIM + k);
    }
    if (narrow) reinterpret_cast<uint32_t*>(codes)[j] = (uint32_t)code;  // (keys of <= 32 bits: n uint32 words)
    else codes[j] = (int64_t)code;
  }
}

// ------------------------------------------------------------------------------------ build
// Node layout: lo[DP] then hi[DP] (2*DP floats).  Empty node: lo = +inf, hi = -inf.
template <int DIM>
__global__ __launch_bounds__(256) void bvh_leaf_kernel(const float* __restrict__ pts, int64_t n_pts,
                                                       int64_t n_leaves_pad, float* __restrict__ nodes) {
  constexpr int DP = padded_dim(DIM);
  const int64_t leaf = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (leaf >= n_leaves_pad) return;
  float lo[DP], hi[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
  for (int u = 0; u < LEAF; ++u) {
    const int64_t j = leaf * LEAF + u;
    if (j < n_pts) {
      float x[DP];
      load_row<DP>(pts + j * DP, x);
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        lo[k] = __builtin_fminf(lo[k], x[k]);
        hi[k] = __builtin_fmaxf(hi[k], x[k]);
      }
    }
  }
  float* dst = nodes + leaf * 2 * DP;
#pragma unroll
  for (int k = 0; k < DP; ++k) { dst[k] = lo[k]; dst[DP + k] = hi[k]; }
}

// Gather of the cloud into curve order AND the leaf boxes in one pass: lane j writes row j (padded, +inf beyond the
// cloud), and the 16 lanes of a leaf reduce its box while the row is still in registers (the separate leaf kernel
// re-read the 256 MB of a 16 M-point cloud).  Rows beyond n_pad exist only as empty leaves (levels are padded to x64).
template <int DIM>
__global__ __launch_bounds__(256) void gather_leaf_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                          const uint32_t* __restrict__ order,
                                                          float* __restrict__ out, int64_t n_pad, int64_t n_rows_all,
                                                          float* __restrict__ leaves, int32_t* __restrict__ dens,
                                                          const float* __restrict__ cbox, int dens_g) {
  constexpr int DP = padded_dim(DIM);
  static_assert(LEAF == 16, "one DPP row of 16 lanes per leaf");
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_rows_all; j += stride) {
    float x[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) x[k] = j < n ? 0.f : __builtin_inff();  // pad rows: +inf (never a nearest neighbour)
    if (j < n) {
      const int64_t src = (int64_t)order[j];
#pragma unroll
      for (int k = 0; k < DIM; ++k) x[k] = pts[src * ld + k];
    }
    if (j < n_pad) {
      float* dst = out + j * DP;
      if constexpr (DP == 2) {
        *reinterpret_cast<float2*>(dst) = make_float2(x[0], x[1]);
      } else if constexpr (DP == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(x[4], x[5], x[6], x[7]);
      }
    }
    float lo[DP], hi[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      lo[k] = (j < n && k < DIM) ? x[k] : __builtin_inff();
      hi[k] = (j < n && k < DIM) ? x[k] : -__builtin_inff();
    }
#pragma unroll
    for (int o = 1; o < LEAF; o <<= 1) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        lo[k] = __builtin_fminf(lo[k], __shfl_xor(lo[k], o));
        hi[k] = __builtin_fmaxf(hi[k], __shfl_xor(hi[k], o));
      }
    }
    if ((threadIdx.x & (LEAF - 1)) == 0) {
      float* dst = leaves + (j / LEAF) * 2 * DP;
#pragma unroll
      for (int k = 0; k < DP; ++k) { dst[k] = lo[k]; dst[DP + k] = hi[k]; }
      // density grid of the cloud for the cell sweep (dens_g^DIM cells over the cloud's box): every leaf adds its
      // point count to the cell under the centre of its box - a sixteenth of the atomics of a pass over the points
      if (dens != nullptr && j < n) {
        int cell = 0;
#pragma unroll
        for (int k = DIM - 1; k >= 0; --k) {
          const float e = cbox[8 + k] - cbox[k];
          const float sc = e > 0.f ? (float)dens_g / e : 0.f;
          int ck = (int)((0.5f * (lo[k] + hi[k]) - cbox[k]) * sc);
          ck = ck < 0 ? 0 : (ck >= dens_g ? dens_g - 1 : ck);
          cell = cell * dens_g + ck;
        }
        const int64_t left = n - j;
        atomicAdd(&dens[cell], (int)(left < LEAF ? left : LEAF));

      }
    }
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void bvh_inner_kernel(const float* __restrict__ child, int64_t n_child,
                                                        int64_t n_nodes_pad, float* __restrict__ nodes,
                                                        int32_t* __restrict__ dens = nullptr, int kind_phase = 0,
                                                        int kind_block0 = 0) {
  // one WAVE per node: lane = child box, 2 x DIM wave reductions (the 64 child boxes are one coalesced read)
  constexpr int DP = padded_dim(DIM);
  if constexpr (DIM == 2 || DIM == 3) {
    // sixteen spare workgroups behind the node builders of the level-1 launch: the cloud-kind statistic of the density
    // grid (flood_common.hpp) - no launch of its own
    if (kind_phase != 0 && (int)blockIdx.x >= kind_block0) {
      __shared__ int s_own[256];
      __shared__ int s_red[8];
      cloud_kind_block<DIM>(dens, dens + (DIM == 2 ? 256 * 256 : 64 * 64 * 64), (int)blockIdx.x - kind_block0, threadIdx.x, s_own,
                            s_red);
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= n_nodes_pad) return;
  float lo[DP], hi[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
  const int64_t c = node * FAN + lane;
  if (c < n_child) {
    float a[DP], b[DP];
    load_row<DP>(child + c * 2 * DP, a);
    load_row<DP>(child + c * 2 * DP + DP, b);
#pragma unroll
    for (int k = 0; k < DIM; ++k) { lo[k] = a[k]; hi[k] = b[k]; }
  }
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    lo[k] = wave_min_f32(lo[k]);
    hi[k] = wave_max_f32(hi[k]);
  }
  if (lane == 0) {
    float* dst = nodes + node * 2 * DP;
#pragma unroll
    for (int k = 0; k < DP; ++k) { dst[k] = lo[k]; dst[DP + k] = hi[k]; }
  }
}

// ------------------------------------------------------------------------------------ sweep
template <int DIM, int KSV, int LB>
__global__ __launch_bounds__(256) void sweep_bvh_kernel(
    const float* __restrict__ pts, const float* __restrict__ nodes, Levels lv,
    const float* __restrict__ verts, const float* __restrict__ weights, int k1, int R,
    int64_t n_simplices, int32_t* __restrict__ queue, uint32_t* __restrict__ out_d2,
    unsigned long long* __restrict__ stats, const int32_t* __restrict__ item_list,
    const int32_t* __restrict__ n_list, int seed, int subs_max, int budget, int refine_pct,
    int32_t* __restrict__ list2, int32_t* __restrict__ count2) {
  // budget > 0 (work-list mode): a wave abandons a tile after `budget` box tests, stores the minima it has
  // (valid upper bounds) and appends the tile to list2; a second pass finishes those tiles split over
  // many more waves.  Bounds the tail caused by tiles near the medial axis of the cloud.
  // subs > 1 (work-list mode, KSV = 1): a tile of 64 samples is split over `subs` waves, each wave
  // carrying 64/subs distinct samples (replicated across its lanes) - tight boxes for the hard tiles.
  constexpr int DP = padded_dim(DIM);
  __shared__ float s_lb[4][MAXL][FAN];
  __shared__ int64_t s_grp[4][MAXL];
  // LB > 1 (work-list mode): the LB nearest candidate leaves of a group are fetched together - lane l loads
  // point l % 16 of leaf slot l / 16 - and staged here, one memory round trip instead of LB dependent ones
  __shared__ float s_stage[LB > 1 ? 4 : 1][LB > 1 ? LB * LEAF : 1][DP];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int n_slots = R;  // sample slots per simplex
  const int tiles = (n_slots + 64 * KSV - 1) / (64 * KSV);
  // few flagged tiles: split each over up to subs_max waves (short tail); many: keep lanes distinct
  int subs = 1;
  if (item_list) {
    subs = subs_max;
    while (subs > 1 && (int64_t)n_list[0] * subs > 32768) subs >>= 1;
    // a list short enough for one sample per wave (every wave at most one item): shortest traversals
    if (subs_max > 1 && (int64_t)n_list[0] * 64 <= (int64_t)gridDim.x * 4) subs = 64;
  }
  const int64_t n_items = item_list ? (int64_t)n_list[0] * subs : n_simplices * tiles;
  const int per_sub = 64 / subs;
  // budgeted pass over a SHORT list: hand every tile straight to the split pass (the GPU would idle)
  if (budget > 0 && n_list[0] <= 2048) budget = 1;
  const int top = lv.n_levels - 1;
  unsigned long long n_leaf_eval = 0, n_leaf_test = 0, n_node_test = 0, max_item_tests = 0;
#ifdef FLOODER_PHASE_TIMERS
  unsigned long long tb_setup = 0, tb_node = 0, tb_skip = 0, tb_eval = 0, tb_prev = __builtin_amdgcn_s_memtime();
#define BVH_PHASE(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - tb_prev; tb_prev = t_; } while (0)
#else
#define BVH_PHASE(acc) do {} while (0)
#endif

  // A list with no more items than waves is dealt out statically, one item per wave: 4096 waves popping an
  // almost empty queue serialise on its one address (~12 ns per atomic - 0.1 ms before any work is done).
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wv;
  const bool static_deal = item_list && n_items <= (int64_t)gridDim.x * 4;
  bool dealt = false;
  int q_shard = (int)(wave_id % QSHARDS), q_tried = 0;
  for (;;) {
    int64_t g;
    if (static_deal) {
      if (dealt || wave_id >= n_items) break;
      dealt = true;
      g = wave_id;
    } else {
      g = queue_pop(queue, q_shard, q_tried, n_items, lane);  // sharded heads (flood_common.hpp)
      if (g < 0) break;
    }
    int sub = 0;
    if (item_list) {  // explicit (simplex, tile) work list
      sub = (int)(g % subs);
      g = (int64_t)item_list[g / subs];
    }
    const int slane = sub * per_sub + (lane & (per_sub - 1));  // sample slot of this lane inside the tile
    const unsigned long long tests_before = n_leaf_test + n_node_test;
    const int64_t s = g / tiles;
    const int tile = (int)(g - s * tiles);
    const int n_live = R;
    if (tile * 64 * KSV >= n_live) continue;

    // ---- this lane's KSV samples: p = sum_j w[r,j] * v[s,j,:]   (core.py:188)
    float p[KSV][DIM];
    float best[KSV];
    int row[KSV];
    const float* vs = verts + s * (int64_t)k1 * DIM;
#pragma unroll
    for (int i = 0; i < KSV; ++i) {
      int slot = tile * 64 * KSV + i * 64 + slane;
      if (slot >= n_live) slot = n_live - 1;  // duplicate of the last live sample, never stored
      const int r = slot;
      row[i] = r;
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[i][k] = 0.f;
      for (int j = 0; j < k1; ++j) {
        const float w = weights[(int64_t)r * k1 + j];
#pragma unroll
        for (int k = 0; k < DIM; ++k) p[i][k] = __builtin_fmaf(w, vs[j * DIM + k], p[i][k]);
      }
      // seed: start from the minima already in out_d2 (upper bounds found by an earlier pass)
      best[i] = seed ? __uint_as_float(out_d2[s * (int64_t)R + r]) : __builtin_inff();
    }
    // ---- bounding box of the tile (wave-uniform)
    float tlo[DIM], thi[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      float mn = p[0][k], mx = p[0][k];
#pragma unroll
      for (int i = 1; i < KSV; ++i) {
        mn = __builtin_fminf(mn, p[i][k]);
        mx = __builtin_fmaxf(mx, p[i][k]);
      }
      tlo[k] = wave_min_f32(mn);
      thi[k] = wave_max_f32(mx);
    }
    float M = __builtin_inff();  // largest running minimum of the tile (wave-uniform)
    if (seed) {
      float bm = best[0];
#pragma unroll
      for (int i = 1; i < KSV; ++i) bm = __builtin_fmaxf(bm, best[i]);
      M = wave_max_f32(bm);
    }

    // child `lane` of group `grp` at level `lvl`: its box (registers) and the lower bound to the tile box
    float c_lo[DIM], c_hi[DIM];
    auto child_bounds = [&](int lvl, int64_t grp) -> float {
      const int64_t idx = grp * FAN + lane;
      float lb = __builtin_inff();
#pragma unroll
      for (int k = 0; k < DIM; ++k) { c_lo[k] = __builtin_inff(); c_hi[k] = -__builtin_inff(); }
      if (idx < lv.count[lvl]) {
        float lo[DP], hi[DP];
        const float* nb = nodes + (lv.off[lvl] + idx) * 2 * DP;
        load_row<DP>(nb, lo);
        load_row<DP>(nb + DP, hi);
        lb = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          c_lo[k] = lo[k];
          c_hi[k] = hi[k];
          const float gap = __builtin_fmaxf(__builtin_fmaxf(lo[k] - thi[k], tlo[k] - hi[k]), 0.f);
          lb = __builtin_fmaf(gap, gap, lb);
        }
      }
      return lb;
    };

    // Inner levels keep their per-lane bounds in LDS (touched once per node expansion); the leaf level
    // runs entirely in registers: bound in lb0, the 64 leaf boxes in c_lo/c_hi, read back by readlane.
    int lvl = top;
    float lb0 = child_bounds(top, 0);
    int64_t grp0 = 0;
    ++n_node_test;
    if (top > 0) {
      s_lb[wv][top][lane] = lb0;
      if (lane == 0) s_grp[wv][top] = 0;
    }
    bool abandoned = false;
    BVH_PHASE(tb_setup);
    for (;;) {
      if (budget > 0 && (int)(n_leaf_test + n_node_test - tests_before) >= budget) {
        abandoned = true;
        break;
      }
      if (lvl > 0) {
        const float lbv = s_lb[wv][lvl][lane];
        const float mn = wave_min_f32(lbv);
        if (!(mn * SAFE < M)) {  // nothing left at this level can improve any sample of the tile
          if (++lvl > top) break;
          continue;
        }
        const int j = __builtin_ctzll(__ballot(lbv == mn));
        if (lane == j) s_lb[wv][lvl][lane] = __builtin_inff();  // visited
        const int64_t c = wave_uniform64(s_grp[wv][lvl]) * FAN + j;  // (readfirstlane: keeps the leaf address scalar, rows in SGPRs)
        --lvl;
        const float lb = child_bounds(lvl, c);
        ++n_node_test;
        if (lvl > 0) {
          s_lb[wv][lvl][lane] = lb;
          if (lane == 0) s_grp[wv][lvl] = c;
        } else {
          lb0 = lb;
          grp0 = c;
          // ---- transposed refine: all 64 leaf boxes of the group (one per lane) against every sample of
          // the tile (broadcast one lane's samples at a time).  A leaf that no sample can still improve on
          // is dropped here, in 1/64 of a per-leaf test each, before the nearest-first loop pops it.
          const bool cand = lb * SAFE < M;
          // worth it when the per-leaf tests it replaces cost more than one pass over the tile's samples
          constexpr int PER_LEAF = KSV * 4 * DIM + 40, PER_GROUP = 64 * KSV * (4 * DIM + 3);
          if (KSV <= 2 && (int64_t)__popcll(__ballot(cand)) * PER_LEAF * 100 * 64 > (int64_t)PER_GROUP * refine_pct * per_sub) {
            bool need = false;
            // (a split tile holds only per_sub distinct samples, in lanes 0 .. per_sub - 1)
#pragma unroll 2
            for (int src = 0; src < per_sub; ++src) {
#pragma unroll
              for (int i = 0; i < KSV; ++i) {
                float lbp = 0.f;
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                  const float pk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[i][k]), src));
                  const float gap = __builtin_fmaxf(__builtin_fmaxf(c_lo[k] - pk, pk - c_hi[k]), 0.f);
                  lbp = __builtin_fmaf(gap, gap, lbp);
                }
                const float bi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best[i]), src));
                need = need || (lbp * SAFE < bi);
              }
            }
            if (!(cand && need)) lb0 = __builtin_inff();
          }
        }
        BVH_PHASE(tb_node);
        continue;
      }
      if constexpr (LB > 1) {
        // ---- leaf level, batched: the (up to) LB nearest unvisited leaves of the current group
        static_assert(LB * LEAF == 64, "one staged point per lane");
        int js[LB];
        bool any = false;
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          js[u] = -1;
          const float mn = wave_min_f32(lb0);
          if (mn * SAFE < M) {  // (wave-uniform; once it fails it fails for the rest of the batch)
            const int j = __builtin_ctzll(__ballot(lb0 == mn));
            if (lane == j) lb0 = __builtin_inff();  // visited
            js[u] = j;
            any = true;
            ++n_leaf_test;
          }
        }
        if (!any) {
          if (++lvl > top) break;
          continue;
        }
        // which of them can still improve a sample of some lane?  (their boxes come from lanes js[u])
        bool use[LB];
        bool any_use = false;
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          use[u] = false;
          if (js[u] >= 0) {
            bool need = false;
            float blo[DIM], bhi[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              blo[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_lo[k]), js[u]));
              bhi[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_hi[k]), js[u]));
            }
#pragma unroll
            for (int i = 0; i < KSV; ++i) {
              float lbp = 0.f;
#pragma unroll
              for (int k = 0; k < DIM; ++k) {
                const float gap = __builtin_fmaxf(__builtin_fmaxf(blo[k] - p[i][k], p[i][k] - bhi[k]), 0.f);
                lbp = __builtin_fmaf(gap, gap, lbp);
              }
              need |= (lbp * SAFE < best[i]);
            }
            use[u] = __ballot(need) != 0ull;
            any_use = any_use || use[u];
          }
        }
        if (!any_use) continue;
        {
          const int slot = lane / LEAF;
          int jl = js[0];
          bool on = use[0];
#pragma unroll
          for (int u = 1; u < LB; ++u) {
            jl = slot == u ? js[u] : jl;
            on = slot == u ? use[u] : on;
          }
          float x[DP];
#pragma unroll
          for (int k = 0; k < DP; ++k) x[k] = __builtin_inff();
          if (on) load_row<DP>(pts + ((grp0 * FAN + jl) * (int64_t)LEAF + (lane % LEAF)) * DP, x);
#pragma unroll
          for (int k = 0; k < DP; ++k) s_stage[wv][lane][k] = x[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          if (!use[u]) continue;  // (wave-uniform)
          ++n_leaf_eval;
#pragma unroll
          for (int h = 0; h < LEAF; h += 2) {
            float ca[DP], cb[DP];
#pragma unroll
            for (int k = 0; k < DP; ++k) {
              ca[k] = s_stage[wv][u * LEAF + h][k];
              cb[k] = s_stage[wv][u * LEAF + h + 1][k];
            }
#pragma unroll
            for (int i = 0; i < KSV; ++i) {
              float da, db;
#pragma unroll
              for (int k = 0; k < DIM; ++k) {
                const float ta = p[i][k] - ca[k];
                const float tb = p[i][k] - cb[k];
                if (k == 0) {
                  da = ta * ta;
                  db = tb * tb;
                } else {
                  da = __builtin_fmaf(ta, ta, da);
                  db = __builtin_fmaf(tb, tb, db);
                }
              }
              best[i] = __builtin_fminf(best[i], __builtin_fminf(da, db));
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        float bmx = best[0];
#pragma unroll
        for (int i = 1; i < KSV; ++i) bmx = __builtin_fmaxf(bmx, best[i]);
        M = wave_max_f32(bmx);
        continue;
      }
      // ---- leaf level: nearest unvisited leaf of the current group
      const float mn = wave_min_f32(lb0);
      if (!(mn * SAFE < M)) {
        if (++lvl > top) break;
        continue;
      }
      const int j = __builtin_ctzll(__ballot(lb0 == mn));
      if (lane == j) lb0 = __builtin_inff();  // visited
      const int64_t c = grp0 * FAN + j;
      ++n_leaf_test;
      // can any sample of any lane still improve against leaf c?  (its box comes from lane j)
      float blo[DIM], bhi[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        blo[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_lo[k]), j));
        bhi[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_hi[k]), j));
      }
      bool need = false;
#pragma unroll
      for (int i = 0; i < KSV; ++i) {
        float lbp = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float gap = __builtin_fmaxf(__builtin_fmaxf(blo[k] - p[i][k], p[i][k] - bhi[k]), 0.f);
          lbp = __builtin_fmaf(gap, gap, lbp);
        }
        need |= (lbp * SAFE < best[i]);
      }
      if (__ballot(need) == 0ull) {
        BVH_PHASE(tb_skip);
        continue;
      }
      BVH_PHASE(tb_skip);
      ++n_leaf_eval;
      const float* cp = pts + c * (int64_t)LEAF * DP;
#pragma unroll
      for (int h = 0; h < LEAF; h += 8) {
        typename RowVec<DP>::type cc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cc[u] = load_uniform_row<DP>(cp + (h + u) * DP);
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
#pragma unroll
          for (int i = 0; i < KSV; ++i) {
            float da, db;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const float ta = p[i][k] - cc[u][k];
              const float tb = p[i][k] - cc[u + 1][k];
              if (k == 0) {
                da = ta * ta;
                db = tb * tb;
              } else {
                da = __builtin_fmaf(ta, ta, da);
                db = __builtin_fmaf(tb, tb, db);
              }
            }
            best[i] = __builtin_fminf(best[i], __builtin_fminf(da, db));
          }
        }
      }
      float bm = best[0];
#pragma unroll
      for (int i = 1; i < KSV; ++i) bm = __builtin_fmaxf(bm, best[i]);
      M = wave_max_f32(bm);
      BVH_PHASE(tb_eval);
    }

#pragma unroll
    for (int i = 0; i < KSV; ++i) {
      if (tile * 64 * KSV + i * 64 + slane < n_live && lane < per_sub)
        out_d2[s * (int64_t)R + row[i]] = __float_as_uint(best[i]);
    }
    if (abandoned && sub == 0 && lane == 0) {  // (budgeted passes run with subs = 1)
      const int pos = atomicAdd(count2, 1);
      list2[pos] = (int)g;
    }
    const unsigned long long item_tests = n_leaf_test + n_node_test - tests_before;
    max_item_tests = item_tests > max_item_tests ? item_tests : max_item_tests;
  }
  // (a wave that had no item adds nothing: thousands of same-address atomics cost ~12 ns each)
  if (stats && lane == 0 && (n_node_test | n_leaf_test) != 0ull) {
    atomicMax(&stats[3], max_item_tests);
    atomicAdd(&stats[0], n_leaf_eval);
    atomicAdd(&stats[1], n_leaf_test);
    atomicAdd(&stats[2], n_node_test);
#ifdef FLOODER_PHASE_TIMERS
    atomicAdd(&stats[40], tb_setup);  // diagnostic build only (tools/bvh_phase.py passes a large buffer)
    atomicAdd(&stats[41], tb_node);
    atomicAdd(&stats[42], tb_skip);
    atomicAdd(&stats[43], tb_eval);
#endif
  }
}

// ------------------------------------------------------------------------------------ host ops
template <int DIM>
struct MortonOp {
  static int run(const float* pts, int64_t n, int ld, const float* box, int64_t* codes, int32_t* zero_buf,
                 int64_t zero_words, hipStream_t st) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((morton_kernel<DIM>), dim3((int)blocks), dim3(256), 0, st, pts, n, ld, box, codes, g_curve,
                       curve_bits_per_axis(DIM), curve_bits_per_axis(DIM) * DIM <= 32 ? 1 : 0, zero_buf, zero_words);
    return check_launch("morton");
  }
};

template <int DIM>
struct BboxOp {
  static int run(const float* pts, int64_t n, int ld, float* box, float* partial, hipStream_t st) {
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > FLOODER_BBOX_BLOCKS) blocks = FLOODER_BBOX_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((bbox_partial_kernel<DIM>), dim3((int)blocks), dim3(256), 0, st, pts, n, ld, partial);
    hipLaunchKernelGGL((bbox_final_kernel<DIM>), dim3(1), dim3(256), 0, st, partial, (int)blocks, box);
    return check_launch("bbox");
  }
};

// the two stages of BboxOp on their own: a cloud that arrives in chunks (host -> device copies on another stream)
// has every chunk reduced as soon as it lands, into its own rows of `partial`
template <int DIM>
struct BboxChunkOp {
  static int run(const float* pts, int64_t n, int ld, float* partial, int n_blocks, hipStream_t st) {
    hipLaunchKernelGGL((bbox_partial_kernel<DIM>), dim3(n_blocks), dim3(256), 0, st, pts, n, ld, partial);
    return check_launch("bbox_chunk");
  }
};

template <int DIM>
struct BboxReduceOp {
  static int run(const float* partial, int n_partial, float* box, hipStream_t st) {
    hipLaunchKernelGGL((bbox_final_kernel<DIM>), dim3(1), dim3(256), 0, st, partial, n_partial, box);
    return check_launch("bbox_reduce");
  }
};

template <int DIM>
struct BuildOp {
  static int run(const float* pts, int64_t n_pts, const Levels& lv, float* nodes, hipStream_t st) {
    constexpr int DP = padded_dim(DIM);
    int64_t pad0 = (lv.count[0] + FAN - 1) / FAN * FAN;
    hipLaunchKernelGGL((bvh_leaf_kernel<DIM>), dim3((unsigned)((pad0 + 255) / 256)), dim3(256), 0, st,
                       pts, n_pts, pad0, nodes + lv.off[0] * 2 * DP);
    for (int l = 1; l < lv.n_levels; ++l) {
      int64_t pad = (lv.count[l] + FAN - 1) / FAN * FAN;
      hipLaunchKernelGGL((bvh_inner_kernel<DIM>), dim3((unsigned)((pad + 3) / 4)), dim3(256), 0, st,
                         nodes + lv.off[l - 1] * 2 * DP, lv.count[l - 1], pad, nodes + lv.off[l] * 2 * DP);
    }
    return check_launch("bvh_build");
  }
};

template <int DIM>
struct IndexRowsOp {
  static int run(const float* pts, int64_t n, int ld, const uint32_t* order, float* rows, int64_t n_pad,
                 const Levels& lv, float* nodes, int32_t* dens, const float* cbox, hipStream_t st) {
    constexpr int DP = padded_dim(DIM);
    const int64_t pad0 = (lv.count[0] + FAN - 1) / FAN * FAN;  // leaves incl. the empty ones that fill the level
    const int64_t n_rows_all = pad0 * LEAF;
    int64_t blocks = (n_rows_all + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const int dens_g = DIM == 2 ? 256 : (DIM == 3 ? 64 : 0);  // (flooder_density_grid_words: the cell sweep's dimensions)
    if (dens_g == 0) dens = nullptr;
    hipLaunchKernelGGL((gather_leaf_kernel<DIM>), dim3((int)blocks), dim3(256), 0, st, pts, n, ld, order, rows, n_pad,
                       n_rows_all, nodes + lv.off[0] * 2 * DP, dens, cbox, dens_g);
    for (int l = 1; l < lv.n_levels; ++l) {
      int64_t pad = (lv.count[l] + FAN - 1) / FAN * FAN;
      const unsigned nb = (unsigned)((pad + 3) / 4);
      // (the cloud-kind statistic rides in sixteen spare workgroups of the first of these launches: flood_common.hpp)
      const int phase = (dens != nullptr && l == 1) ? 1 : 0;
      hipLaunchKernelGGL((bvh_inner_kernel<DIM>), dim3(nb + (phase == 1 ? 16u : 0u)), dim3(256), 0, st,
                         nodes + lv.off[l - 1] * 2 * DP, lv.count[l - 1], pad, nodes + lv.off[l] * 2 * DP, dens, phase, (int)nb);
    }
    return check_launch("index_rows");
  }
};

template <int DIM>
struct SweepBvhOp {
  static int run(const float* pts, const float* nodes, const Levels& lv, const float* verts,
                 const float* weights, int k1, int R, int64_t ns, int32_t* queue, uint32_t* out,
                 unsigned long long* stats, const int32_t* item_list, const int32_t* n_list, int seed,
                 int force_ks, int subs_max, int budget, int32_t* list2, int32_t* count2, hipStream_t st) {
    const int grid = g_bvh_grid;  // persistent blocks; 4 independent waves each
    int ks = force_ks ? force_ks : g_bvh_ks;
    if (ks == 0) ks = R <= 64 ? 1 : 2;  // (measured at cfg 2: 1: 10.7 ms, 2: 10.1, 4: 10.3, 8: 12.1)
    // the transposed refine pays at its cost model's threshold when finishing flagged tiles and at 3x the
    // threshold in the full sweep (cfg 3 finish 12.5 ms vs 17.0 without it; cfg 2 full tree sweep 6.4 vs 8.6 ms)
    const int refine_pct = item_list ? g_bvh_refine_pct : 3 * g_bvh_refine_pct;
    const bool batch = item_list != nullptr && ks == 1 && g_bvh_leaf_batch > 1;
#define FLOODER_LAUNCH_BVH(KS_, LB_)                                                                              \
  hipLaunchKernelGGL((sweep_bvh_kernel<DIM, KS_, LB_>), dim3(grid), dim3(256), 0, st, pts, nodes, lv, verts, weights, \
                     k1, R, ns, queue, out, stats, item_list, n_list, seed, subs_max, budget, refine_pct, list2, count2)
    if (batch) FLOODER_LAUNCH_BVH(1, 4);
    else if (ks == 1) FLOODER_LAUNCH_BVH(1, 1);
    else if (ks == 2) FLOODER_LAUNCH_BVH(2, 1);
    else if (ks == 4) FLOODER_LAUNCH_BVH(4, 1);
    else FLOODER_LAUNCH_BVH(8, 1);
#undef FLOODER_LAUNCH_BVH
    return check_launch("sweep_bvh");
  }
};

// self-test of the DPP reductions
__global__ void selftest_kernel(const float* in, float* out) {
  const float x = in[threadIdx.x];
  out[threadIdx.x] = wave_min_f32(x);
  out[64 + threadIdx.x] = wave_max_f32(x);
  // inclusive scan by __shfl_up, as used by the cell sweep's counting sort
  int v = (int)(x * 100.f);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  out[128 + threadIdx.x] = (float)v;
}

}  // namespace

extern "C" {

int64_t flooder_bvh_node_count(int64_t n_pts) {
  const Levels lv = make_levels(n_pts);
  return total_nodes(lv);
}

int flooder_bbox_f32(const float* pts, int64_t n_pts, int dim, int ld, float* box, float* partial,
                     void* stream) {
  if (!pts || !box || !partial || n_pts < 1 || ld < dim || dim < 1 || dim > FLOODER_MAX_DIM)
    return fail(FLOODER_E_ARG, "flooder_bbox_f32: bad argument");
  return dispatch_dim<BboxOp>(dim, pts, n_pts, ld, box, partial, (hipStream_t)stream);
}

int flooder_bbox_chunk_f32(const float* pts, int64_t n_rows, int dim, int ld, float* partial, int n_blocks,
                           void* stream) {
  if (!pts || !partial || n_rows < 1 || ld < dim || dim < 1 || dim > FLOODER_MAX_DIM || n_blocks < 1 || n_blocks > 65535)
    return fail(FLOODER_E_ARG, "flooder_bbox_chunk_f32: bad argument");
  return dispatch_dim<BboxChunkOp>(dim, pts, n_rows, ld, partial, n_blocks, (hipStream_t)stream);
}

int flooder_bbox_reduce_f32(const float* partial, int n_partial, int dim, float* box, void* stream) {
  if (!partial || !box || n_partial < 1 || dim < 1 || dim > FLOODER_MAX_DIM)
    return fail(FLOODER_E_ARG, "flooder_bbox_reduce_f32: bad argument");
  return dispatch_dim<BboxReduceOp>(dim, partial, n_partial, box, (hipStream_t)stream);
}

int flooder_morton_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* box, int64_t* codes,
                       void* stream) {
  if (n_pts == 0) return FLOODER_OK;
  if (!pts || !box || !codes || n_pts < 0 || ld < dim || dim < 1 || dim > FLOODER_MAX_DIM)
    return fail(FLOODER_E_ARG, "flooder_morton_f32: bad argument");
  return dispatch_dim<MortonOp>(dim, pts, n_pts, ld, box, codes, (int32_t*)nullptr, (int64_t)0, (hipStream_t)stream);
}

int flooder_morton_zero_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* box, int64_t* codes,
                            int32_t* zero_buf, int64_t zero_words, void* stream) {
  if (n_pts == 0) return FLOODER_OK;
  if (!pts || !box || !codes || n_pts < 0 || ld < dim || dim < 1 || dim > FLOODER_MAX_DIM || zero_words < 0 ||
      (zero_words > 0 && !zero_buf))
    return fail(FLOODER_E_ARG, "flooder_morton_zero_f32: bad argument");
  return dispatch_dim<MortonOp>(dim, pts, n_pts, ld, box, codes, zero_buf, zero_words, (hipStream_t)stream);
}

int flooder_curve_key_bits(int dim) {
  if (dim < 1 || dim > FLOODER_MAX_DIM) return 0;
  return curve_bits_per_axis(dim) * dim;
}

int flooder_index_rows_f32(const float* pts, int64_t n_pts, int dim, int ld, const int32_t* order, float* rows,
                           int64_t n_pad, float* nodes, int32_t* density_grid, const float* cloud_box, void* stream) {
  if (!pts || !order || !rows || !nodes || n_pts < 1 || n_pad < n_pts || ld < dim || (density_grid && !cloud_box))
    return fail(FLOODER_E_ARG, "flooder_index_rows_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<IndexRowsOp>(dim, pts, n_pts, ld, reinterpret_cast<const uint32_t*>(order), rows, n_pad, lv,
                                   nodes, density_grid, cloud_box, (hipStream_t)stream);
}

int flooder_bvh_build_f32(const float* pts_sorted, int64_t n_pts, int dim, float* nodes, void* stream) {
  if (!pts_sorted || !nodes || n_pts < 1) return fail(FLOODER_E_ARG, "flooder_bvh_build_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<BuildOp>(dim, pts_sorted, n_pts, lv, nodes, (hipStream_t)stream);
}

int flooder_sweep_bvh_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                          const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                          int32_t* queue, uint32_t* out_d2, uint64_t* stats, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !queue || !out_d2 || n_pts < 1 || k1 < 1 ||
      k1 > FLOODER_MAX_VERTS || R < 0)
    return fail(FLOODER_E_ARG, "flooder_sweep_bvh_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<SweepBvhOp>(dim, pts_sorted, nodes, lv, verts, weights, k1, R, n_simplices, queue,
                                  out_d2, reinterpret_cast<unsigned long long*>(stats), nullptr, nullptr, 0, 0, 1,
                                  0, nullptr, nullptr, (hipStream_t)stream);
}

int flooder_sweep_bvh_items_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                const float* verts, const float* weights, int k1, int R,
                                int64_t n_simplices, const int32_t* item_list, const int32_t* n_items,
                                int32_t* queue, uint32_t* out_d2, int budget, int32_t* list2,
                                int32_t* count2, uint64_t* stats, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !queue || !out_d2 || !item_list || !n_items ||
      n_pts < 1 || k1 < 1 || k1 > FLOODER_MAX_VERTS || R < 0 || budget < 0 || (budget > 0 && (!list2 || !count2)))
    return fail(FLOODER_E_ARG, "flooder_sweep_bvh_items_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  // a budgeted pass keeps 64 distinct samples per wave; the unbudgeted pass may split tiles
  const int subs = budget > 0 ? 1 : g_bvh_subs;
  return dispatch_dim<SweepBvhOp>(dim, pts_sorted, nodes, lv, verts, weights, k1, R, n_simplices, queue,
                                  out_d2, reinterpret_cast<unsigned long long*>(stats), item_list, n_items, 1,
                                  1, subs, budget, list2, count2, (hipStream_t)stream);
}

int flooder_selftest(const float* in64, float* out128, void* stream) {
  if (!in64 || !out128) return fail(FLOODER_E_ARG, "flooder_selftest: bad argument");
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in64, out128);
  return check_launch("selftest");
}

}  // extern "C"
