// flood_sorted.hip - tree sweep over spatially sorted samples (gfx950; any dimension, default above 3D).
//
// The tree sweep of flood_bvh.hip gives a wave the samples of ONE simplex (a tile of 64 * KS lattice rows) and
// walks the box tree with the tile's bounding box.  In 2D / 3D, with thousands of samples per simplex, a tile is a
// small patch.  In higher dimensions only coarse lattices are tractable (cfg 4 of BASELINE.json: triangles in 6D
// with 36 samples each), a tile is the WHOLE simplex, its box spans a landmark spacing in every axis and most of
// the tree overlaps it: 6400 leaf-box tests and 270 node expansions per tile for 260 evaluated leaves.
//
// Here the samples of all simplices are ordered along a Z-order curve first (one key per sample, one radix sort)
// and a wave takes 64 CONSECUTIVE samples of that order - neighbours in space whatever simplex they belong to, all
// 64 lanes busy.  The per-sample minima land in the same (S, R) buffer as before (scattered 4-byte stores), so the
// face maxima, the cross-shard reduction and the results are unchanged: the minimum over ALL points, bit for bit.
//
//   sample_keys     key[i] = Morton code (floor(32 / dim) bits per axis) of sample i = (s, r) inside the cloud's box
//   (sort)          flooder_index_sort: rocprim radix sort of (key, i), as for the cloud itself
//   sweep_sorted    wave = 64 consecutive sorted samples; wave-uniform nearest-first traversal as in flood_bvh.hip

#include "flood_common.hpp"
#include "flood_bvh.hpp"

using namespace flooder;

namespace flooder { int g_sorted_ks = 1; int g_sorted_refresh = 4; int g_sorted_blocks = 0; int g_sorted_batch_pct = 400; }  // samples per lane of the sorted sweep (option "sorted_ks": 1 or 2)

namespace {

constexpr float SAFE = 0.99999f;

// SUB-TILES.  The tile's bounding box against a leaf box is a weak test in 6D (the gaps of six axes add up, and the box of
// 64 samples is as wide as a leaf): three of four leaves that pass it fail the per-sample test that follows, and that
// test - the leaf's box broadcast from its lane, every sample against it, a ballot - is a serial chain that took 45 % of
// the kernel's cycles.  The tile is therefore cut into NSUB runs of 64 / NSUB consecutive lanes (consecutive samples of
// the sorted order: each run is a tighter box), every leaf lane keeps its NSUB lower bounds, and a leaf stays a
// candidate only while one of them is below the largest running minimum OF THAT RUN (four DPP steps after every
// evaluated leaf).  Only bounds: what a tile evaluates shrinks, the minima do not change.
#ifndef FLOODER_SORTED_NSUB
#define FLOODER_SORTED_NSUB 4
#endif
constexpr int NSUB = FLOODER_SORTED_NSUB;   // 1 (off), 4, 8 or 16
// the transposed refine (below) paid while every leaf test began with a wave-wide minimum; with batched tests it is a
// loss at every threshold (cfg 4: 47.7 ms without, 49.7 / 51.5 / 59.6 ms at 200 / 100 / 50 %): compiled out
// the box of the leaf under test: 1 = read from LDS (written once per group), 0 = twelve v_readlane from its lane
#ifndef FLOODER_SORTED_LDSBOX
#define FLOODER_SORTED_LDSBOX 1
#endif
static_assert(!FLOODER_SORTED_LDSBOX || FLOODER_SORTED_NSUB > 1, "the leaf boxes are published by sub_bounds()");
// sub-tile bounds also for the parents of the leaf groups (1) or for leaves only (0): 41.3 -> 40.7 ms at cfg 4 on their
// own, nothing beside the exact test of the parent's box (34.4 with, 34.3 without): off
#ifndef FLOODER_SORTED_NODE_SUB
#define FLOODER_SORTED_NODE_SUB 0
#endif
constexpr bool SORTED_NODE_SUB = FLOODER_SORTED_NODE_SUB != 0;
// exact per-sample test of a leaf group's parent box before the group is opened (1) or not (0)
#ifndef FLOODER_SORTED_NODE_EXACT
#define FLOODER_SORTED_NODE_EXACT 1
#endif
constexpr bool SORTED_NODE_EXACT = FLOODER_SORTED_NODE_EXACT != 0;
// rows of a leaf per batch of scalar loads for 8-float rows: 8 (1) or 4 (0).  8 spills 37 more scalar registers at 7 waves
// per SIMD and is still 1.0 ms faster at cfg 4 (32.85 against 33.85 ms, twice on the same box)
#ifndef FLOODER_SORTED_UB8
#define FLOODER_SORTED_UB8 1
#endif
#ifndef FLOODER_SORTED_REFINE
#define FLOODER_SORTED_REFINE 0
#endif
constexpr bool SORTED_REFINE = FLOODER_SORTED_REFINE != 0;
constexpr int SUBSZ = 64 / NSUB;
static_assert(NSUB == 1 || NSUB == 4 || NSUB == 8 || NSUB == 16, "sub-tiles of 64, 16, 8 or 4 lanes");

// all-reduce inside runs of SUBSZ lanes (aligned): every lane ends up with its run's value
#define FLOODER_SUB_STEP(OP, x, PATTERN) \
  asm volatile("s_nop 4\n\t" OP " %0, %0, %0 " PATTERN " row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(x))
template <bool MAX>
__device__ __forceinline__ float sub_reduce(float x) {
#if FLOODER_DPP_ASM
  // (one instruction per step: flood_common.hpp, FLOODER_DPP_ASM)
  if constexpr (SUBSZ == 64) {
    return MAX ? wave_max_f32(x) : wave_min_f32(x);
  } else if constexpr (MAX) {
    FLOODER_SUB_STEP("v_max_f32_dpp", x, "quad_perm:[1,0,3,2]");
    FLOODER_SUB_STEP("v_max_f32_dpp", x, "quad_perm:[2,3,0,1]");
    if constexpr (SUBSZ >= 8) FLOODER_SUB_STEP("v_max_f32_dpp", x, "row_half_mirror");
    if constexpr (SUBSZ >= 16) FLOODER_SUB_STEP("v_max_f32_dpp", x, "row_mirror");
    return x;
  } else {
    FLOODER_SUB_STEP("v_min_f32_dpp", x, "quad_perm:[1,0,3,2]");
    FLOODER_SUB_STEP("v_min_f32_dpp", x, "quad_perm:[2,3,0,1]");
    if constexpr (SUBSZ >= 8) FLOODER_SUB_STEP("v_min_f32_dpp", x, "row_half_mirror");
    if constexpr (SUBSZ >= 16) FLOODER_SUB_STEP("v_min_f32_dpp", x, "row_mirror");
    return x;
  }
#endif
  auto op = [](float a, float b) { return MAX ? __builtin_fmaxf(a, b) : __builtin_fminf(a, b); };
  x = op(x, dpp_move<0xB1, 0xF>(x));                            // quad_perm [1,0,3,2]
  x = op(x, dpp_move<0x4E, 0xF>(x));                            // quad_perm [2,3,0,1]
  if constexpr (SUBSZ >= 8) x = op(x, dpp_move<0x141, 0xF>(x));  // row_half_mirror
  if constexpr (SUBSZ >= 16) x = op(x, dpp_move<0x140, 0xF>(x)); // row_mirror
  if constexpr (SUBSZ == 64) {
    x = op(x, dpp_move<0x142, 0xA>(x));
    x = op(x, dpp_move<0x143, 0xC>(x));
    x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
  }
  return x;
}

template <int DIM>
__global__ __launch_bounds__(256) void sample_keys_kernel(const float* __restrict__ verts,
                                                          const float* __restrict__ weights, int k1, int R,
                                                          int64_t n_samples, const float* __restrict__ dbox,
                                                          uint32_t* __restrict__ keys, int hilbert,
                                                          const uint8_t* __restrict__ late) {
  constexpr int BITS = 32 / DIM > 10 ? 10 : 32 / DIM;
  float lo[DIM], scale[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    const float ext = dbox[8 + k] - dbox[k];
    lo[k] = dbox[k];
    scale[k] = ext > 0.f ? (float)((1u << BITS) - 1u) / ext : 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_samples; i += stride) {
    const int64_t s = i / R;
    const int r = (int)(i - s * R);
    float p[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) p[k] = 0.f;
    for (int j = 0; j < k1; ++j) {
      const float w = weights[(int64_t)r * k1 + j];
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[k] = __builtin_fmaf(w, verts[(s * k1 + j) * DIM + k], p[k]);
    }
    uint32_t q[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      float t = (p[k] - lo[k]) * scale[k];
      t = t < 0.f ? 0.f : t;  // (samples of landmarks off the cloud may leave the box: clamped, order only)
      const float top = (float)((1u << BITS) - 1u);
      t = t > top ? top : t;
      q[k] = (uint32_t)t;
    }
    uint32_t code = 0u;
    if (hilbert && DIM > 1) {
      // Hilbert index (Skilling's axes-to-transpose transform, as morton_kernel does for the cloud): consecutive
      // keys are neighbours in space - a Z-order jumps across the box at every power-of-two boundary, and a tile that
      // straddles a jump has a box as large as the jump
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
        for (int k = 0; k < DIM; ++k) code = (code << 1) | ((q[k] >> b) & 1u);
    } else {
#pragma unroll
      for (int k = 0; k < DIM; ++k)
#pragma unroll
        for (int b = 0; b < BITS; ++b) code |= ((q[k] >> b) & 1u) << (b * DIM + k);
    }
    // fused sweep: the PILOT rows of every simplex (one sample near the centre of each face) sort ahead of all other
    // rows (late[r] != 0: top key bit set), so that the running face maxima are close to final when the bulk arrives
    if (late != nullptr) {
      if (BITS * DIM >= 32) code >>= 1;
      if (late[r]) code |= 0x80000000u;
    }
    keys[i] = code;
  }
}

// FUSED: only the per-face maxima are wanted (core.py:251-276 folded in).  A sample whose running minimum - the
// distance to a real point, an upper bound of its nearest-neighbour distance - does not exceed the running maximum
// of any face it lies on can never raise a face value: it leaves the tile's pruning radius M and the per-leaf tests
// ("dead"), and the traversal ends when the LIVE samples are exact.  Those deliver their values (integer atomic
// max on face_bits); a dead sample delivers nothing.  face_bits only ever receives exact values, so the face values
// equal the exhaustive result bit for bit.  Proving a minimum exact costs most of a traversal (every leaf whose box
// is nearer than the minimum found must be looked at); seeing that a sample is out of the running takes the first
// one or two leaves.
struct SortedFaces {
  const uint32_t* memb;     // per row: bit f set = the row lies on face f
  uint32_t* face_bits;      // running maxima (d2 bits), zeroed by the caller
  const int32_t* slot;      // NULL, or slot[s * n_faces + f]
  int n_faces;
  int refresh;              // the face maxima are re-read every so many evaluated leaves
  __device__ __forceinline__ int64_t slot_of(int64_t s, int f) const {
    return slot ? (int64_t)slot[s * (int64_t)n_faces + f] : s * (int64_t)n_faces + f;
  }
};

// 7 waves per SIMD for the default instantiation: left alone the compiler takes all 106 scalar registers, which holds
// the kernel at 6 (MI355X_MICROARCH.md: 800 per SIMD in blocks of 16); at 94 it spills 12 more of them and is 2 % faster
// (cfg 4: 46.9 -> 46.0 ms; 8 waves = 78 registers, 64 spilled: 47.8 ms)
#ifndef FLOODER_SORTED_MIN_WAVES
#define FLOODER_SORTED_MIN_WAVES 7
#endif
// A rank of a multi-GPU run takes a CONTIGUOUS world-th of the tiles of the sorted order: its tiles are tiles of the
// unsharded sweep, and they lie in one region of space - an eighth of cfg 4's tiles touches an eighth of the cloud
// (6.9 ms of sweep; every 8th chunk of 256 tiles, which evens out the 12 % a tile at the sparse end of the curve costs
// more, works on the whole cloud from a cold cache: 8.7 - 9.7 ms).  world <= 1: all tiles.
struct TileShard {
  int rank, world;
};

template <int DIM, int KS, bool FUSED>
__global__ __launch_bounds__(256, (KS == 1 && !FUSED && DIM <= 6) ? FLOODER_SORTED_MIN_WAVES : 1) void sweep_sorted_kernel(
    const float* __restrict__ pts, const float* __restrict__ nodes, Levels lv,
    const float* __restrict__ verts, const float* __restrict__ weights, int k1, int R,
    int64_t n_samples, const uint32_t* __restrict__ order, int32_t* __restrict__ queue,
    uint32_t* __restrict__ out_d2, unsigned long long* __restrict__ stats, int refine_pct, float batch_scale,
    SortedFaces sf, TileShard ts) {
  // KS samples per lane: a tile is 64 * KS consecutive samples of the sorted order (lane l holds l, l + 64, ...)
  constexpr int DP = padded_dim(DIM);
  constexpr int TILE = 64 * KS;
  __shared__ float s_lb[4][MAXL][FAN];
  __shared__ int64_t s_grp[4][MAXL];
  __shared__ float s_sub[4][NSUB][2 * FLOODER_MAX_DIM];   // boxes of the tile's sub-tiles (lo[8], hi[8])
#if FLOODER_SORTED_LDSBOX
  __shared__ float s_box[4][FAN][2 * DIM];                // boxes of the current group's leaves (a test reads one)
#endif
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int64_t n_tiles_all = (n_samples + TILE - 1) / TILE;
  int64_t n_tiles = n_tiles_all;
  int64_t tile_first = 0;
  if (ts.world > 1) {
    tile_first = n_tiles_all * ts.rank / ts.world;
    n_tiles = n_tiles_all * (ts.rank + 1) / ts.world - tile_first;
  }
  const int top = lv.n_levels - 1;
  unsigned long long n_leaf_eval = 0, n_leaf_test = 0, n_node_test = 0, max_item_tests = 0, n_node_spared = 0;

#ifdef FLOODER_SORTED_TIMERS
  // diagnostic build: cycles per phase (s_memtime), summed over the waves into stats[4..9], refine passes in stats[10]
  unsigned long long t_ph[6] = {0, 0, 0, 0, 0, 0}, ts_prev = __builtin_amdgcn_s_memtime(), n_refine = 0;
  unsigned long long n_groups = 0, n_groups_empty = 0, n_groups_idle = 0;
  int evals_in_group = 0;
#define SPHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); t_ph[i] += t_ - ts_prev; ts_prev = t_; } while (0)
#else
#define SPHASE(i) do { } while (0)
#endif
  int q_shard = (int)((blockIdx.x * 4 + wv) % QSHARDS), q_tried = 0;
  for (;;) {
    int64_t g = queue_pop(queue, q_shard, q_tried, n_tiles, lane);  // sharded heads (flood_common.hpp)
    if (g < 0) break;
    g += tile_first;
    SPHASE(0);
    const unsigned long long tests_before = n_leaf_test + n_node_test;
    // ---- this lane's samples: p = sum_j w[r,j] * v[s,j,:]   (core.py:188; same fma order as every other sweep)
    float p[KS][DIM], best[KS];
    uint32_t id[KS];
    bool live[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int64_t pos = g * TILE + i * 64 + lane;
      live[i] = pos < n_samples;
      id[i] = order[live[i] ? pos : n_samples - 1];  // (dead lanes repeat the last sample, never stored)
      const uint32_t s = id[i] / (uint32_t)R;
      const uint32_t r = id[i] - s * (uint32_t)R;
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[i][k] = 0.f;
      const float* vs = verts + (int64_t)s * k1 * DIM;
      const float* ws = weights + (int64_t)r * k1;
      for (int j = 0; j < k1; ++j) {
        const float w = ws[j];
#pragma unroll
        for (int k = 0; k < DIM; ++k) p[i][k] = __builtin_fmaf(w, vs[j * DIM + k], p[i][k]);
      }
      best[i] = __builtin_inff();
    }
    // FUSED: faces of this lane's samples, the smallest running maximum among them, who is still in the running
    uint32_t mb[KS], thr[KS];
    bool alive[KS];
    auto read_thr = [&]() {
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        uint32_t t = 0xffffffffu, m = mb[i];
        const int64_t s_i = (int64_t)(id[i] / (uint32_t)R);
        while (m) {  // (per lane: at most dimension + 1 faces for a vertex, one for an interior sample)
          const int f = __builtin_ctz(m);
          m &= m - 1u;
          const uint32_t v = __hip_atomic_load(sf.face_bits + sf.slot_of(s_i, f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          t = v < t ? v : t;
        }
        thr[i] = t;
      }
    };
    if constexpr (FUSED) {
#pragma unroll
      for (int i = 0; i < KS; ++i) mb[i] = live[i] ? sf.memb[id[i] - (id[i] / (uint32_t)R) * (uint32_t)R] : 0u;
      read_thr();
#pragma unroll
      for (int i = 0; i < KS; ++i) alive[i] = live[i] && mb[i] != 0u;
    } else {
#pragma unroll
      for (int i = 0; i < KS; ++i) { alive[i] = true; mb[i] = 0u; thr[i] = 0u; }
    }
    int evals_since = 0;
    float tlo[DIM], thi[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      float mn = p[0][k], mx = p[0][k];
#pragma unroll
      for (int i = 1; i < KS; ++i) {
        mn = __builtin_fminf(mn, p[i][k]);
        mx = __builtin_fmaxf(mx, p[i][k]);
      }
      tlo[k] = wave_min_f32(mn);
      thi[k] = wave_max_f32(mx);
      if constexpr (NSUB > 1) {
        const float smn = sub_reduce<false>(mn), smx = sub_reduce<true>(mx);
        if ((lane & (SUBSZ - 1)) == 0) {
          s_sub[wv][lane / SUBSZ][k] = smn;
          s_sub[wv][lane / SUBSZ][FLOODER_MAX_DIM + k] = smx;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    float M = __builtin_inff();  // largest running minimum of the tile (wave-uniform)
    float Mq[NSUB], lbq[NSUB];   // ... of every sub-tile (wave-uniform); this lane's leaf against the sub-tiles' boxes
#pragma unroll
    for (int q = 0; q < NSUB; ++q) { Mq[q] = __builtin_inff(); lbq[q] = __builtin_inff(); }
    // is this lane's leaf of the current group still worth a look?
    auto leaf_cand = [&](float lb_tile) -> bool {
      if constexpr (NSUB == 1) {
        return lb_tile * SAFE < M;
      } else {
        bool c = false;
#pragma unroll
        for (int q = 0; q < NSUB; ++q) c = c || (lbq[q] * SAFE < Mq[q]);
        return c;
      }
    };

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

    // this lane's leaf box (c_lo, c_hi) against every sub-tile's box; the smallest of them is the leaf's place in the
    // nearest-first order (a lower bound for every sample, tighter than the tile's)
    auto sub_bounds_of = [&](float (&dst)[NSUB]) -> float {
      float nearest = __builtin_inff();
#pragma unroll
      for (int q = 0; q < NSUB; ++q) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float gap = __builtin_fmaxf(__builtin_fmaxf(c_lo[k] - s_sub[wv][q][FLOODER_MAX_DIM + k],
                                                            s_sub[wv][q][k] - c_hi[k]), 0.f);
          a = __builtin_fmaf(gap, gap, a);
        }
        dst[q] = a;
        nearest = __builtin_fminf(nearest, a);
      }
      return nearest;
    };
    auto sub_bounds = [&]() -> float {
#if FLOODER_SORTED_LDSBOX
#pragma unroll
      for (int k = 0; k < DIM; ++k) {   // the group's leaf boxes, for the per-leaf tests
        s_box[wv][lane][k] = c_lo[k];
        s_box[wv][lane][DIM + k] = c_hi[k];
      }
#endif
      return sub_bounds_of(lbq);
    };
    // the same bounds for the nodes ONE level above the leaves (each the parent of a leaf group; one group of them is
    // open at a time, so a register set does): 35 of the 56 leaf groups a cfg 4 tile used to open were left without
    // a single evaluation - the tile's box reaches them, no run of 16 samples does
    // (kept as halves rounded toward zero - still lower bounds - two to a register: four more live floats spill)
    typedef __fp16 half2_t __attribute__((ext_vector_type(2)));
    half2_t lbq1[(NSUB + 1) / 2];
#pragma unroll
    for (int q = 0; q < (NSUB + 1) / 2; ++q) lbq1[q] = half2_t{(__fp16)0.f, (__fp16)0.f};
    auto node_bounds = [&]() -> float {
      float tmp[NSUB];
      const float nearest = sub_bounds_of(tmp);
#pragma unroll
      for (int q = 0; q + 1 < NSUB; q += 2) lbq1[q / 2] = __builtin_amdgcn_cvt_pkrtz(tmp[q], tmp[q + 1]);
      return nearest;
    };
    auto node_cand = [&]() -> bool {
      bool c = false;
#pragma unroll
      for (int q = 0; q + 1 < NSUB; q += 2) {
        c = c || ((float)lbq1[q / 2][0] * SAFE < Mq[q]) || ((float)lbq1[q / 2][1] * SAFE < Mq[q + 1]);
      }
      return c;
    };
    SPHASE(1);
    int lvl = top;
    unsigned long long batch = 0ull;   // leaves of the current group taken out of lb0 and waiting for their test
    float lb0 = child_bounds(top, 0);
    if constexpr (NSUB > 1) {
      if (top == 0) lb0 = sub_bounds();
    }
    int64_t grp0 = 0;
    ++n_node_test;
    if (top > 0) {
      if constexpr (NSUB > 1 && SORTED_NODE_SUB) {
        if (top == 1) lb0 = node_bounds();
      }
      s_lb[wv][top][lane] = lb0;
      if (lane == 0) s_grp[wv][top] = 0;
    }
    for (;;) {
      if (lvl > 0) {
        float lbv = s_lb[wv][lvl][lane];
        if constexpr (NSUB > 1 && SORTED_NODE_SUB) {
          if (lvl == 1 && !node_cand()) lbv = __builtin_inff();   // (for good: the sub-tiles' maxima only fall)
        }
        const float mn = wave_min_f32(lbv);
        if (!(mn * SAFE < M)) {  // nothing left at this level can improve any sample of the tile
          if (++lvl > top) break;
          continue;
        }
        const int j = __builtin_ctzll(__ballot(lbv == mn));
        if (lane == j) s_lb[wv][lvl][lane] = __builtin_inff();  // visited
        // (the group index comes back from LDS in a VGPR: without the readfirstlane the compiler cannot know the
        // leaf address below is wave-uniform and loads the 16 rows of a leaf into VGPRs instead of SGPRs)
        const int64_t c = wave_uniform64(s_grp[wv][lvl]) * FAN + j;
        --lvl;
        if constexpr (SORTED_NODE_EXACT) {
          // the parent of a leaf group, before the group is opened: its box (two scalar loads) against every sample.
          // 28 of the 49 groups a cfg 4 tile opened were left without an evaluation; this test spares 21 of them the
          // 64 box loads, the sub-tile bounds and the leaf tests that all fail
          if (lvl == 0) {
            const float* nb = nodes + (lv.off[1] + c) * 2 * DP;
            const typename RowVec<DP>::type nlo = load_uniform_row<DP>(nb), nhi = load_uniform_row<DP>(nb + DP);
            bool nd = false;
#pragma unroll
            for (int i = 0; i < KS; ++i) {
              float lbp = 0.f;
#pragma unroll
              for (int k = 0; k < DIM; ++k) {
                const float gap = __builtin_fmaxf(__builtin_fmaxf(nlo[k] - p[i][k], p[i][k] - nhi[k]), 0.f);
                lbp = __builtin_fmaf(gap, gap, lbp);
              }
              nd = nd || (alive[i] && lbp * SAFE < best[i]);
            }
            if (__ballot(nd) == 0ull) {
              lvl = 1;
              ++n_node_spared;
              continue;
            }
          }
        }
        const float lb = child_bounds(lvl, c);
        ++n_node_test;
        if (lvl > 0) {
          float lbn = lb;
          if constexpr (NSUB > 1 && SORTED_NODE_SUB) {
            if (lvl == 1) lbn = node_bounds();
          }
          s_lb[wv][lvl][lane] = lbn;
          if (lane == 0) s_grp[wv][lvl] = c;
        } else {
          lb0 = lb;
          grp0 = c;
          if constexpr (NSUB > 1) lb0 = sub_bounds();
#ifdef FLOODER_SORTED_TIMERS
          ++n_groups;
          if (__ballot(leaf_cand(lb0)) == 0ull) ++n_groups_empty;   // (no leaf of the group is a candidate on arrival)
          if (evals_in_group == 0 && n_groups > 1) ++n_groups_idle;  // (the PREVIOUS group was left without an evaluation)
          evals_in_group = 0;
#endif
          // ---- transposed refine: the 64 leaf boxes of the group (one per lane) against every sample of the tile
          // (one lane's samples broadcast at a time); a leaf no sample can improve on is dropped here, in 1/64 of a
          // per-leaf test, before the nearest-first loop pops it.  Worth it when the per-leaf tests it replaces
          // cost more than one pass over the tile's samples.
          SPHASE(2);
          constexpr int PER_LEAF = KS * 4 * DIM + 40, PER_GROUP = 64 * KS * (4 * DIM + 3);
          const bool cand = SORTED_REFINE ? leaf_cand(lb) : false;
          if (SORTED_REFINE && (int64_t)__popcll(__ballot(cand)) * PER_LEAF * 100 > (int64_t)PER_GROUP * refine_pct) {
            bool need = false;
#pragma unroll 2
            for (int src = 0; src < 64; ++src) {
#pragma unroll
              for (int i = 0; i < KS; ++i) {
                float lbp = 0.f;
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                  const float pk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[i][k]), src));
                  const float gap = __builtin_fmaxf(__builtin_fmaxf(c_lo[k] - pk, pk - c_hi[k]), 0.f);
                  lbp = __builtin_fmaf(gap, gap, lbp);
                }
                const float bi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(alive[i] ? best[i] : -1.f), src));
                need = need || (lbp * SAFE < bi);
              }
            }
            if (!(cand && need)) lb0 = __builtin_inff();
#ifdef FLOODER_SORTED_TIMERS
            ++n_refine;
#endif
            SPHASE(3);
          }
        }
        SPHASE(2);
        continue;
      }
      // ---- leaf level: the nearest unvisited leaves of the current group that are still candidates.  Finding THE
      // nearest costs a wave-wide minimum (six dependent DPP steps) per leaf; one minimum opens a BATCH instead - every
      // candidate whose bound is within batch_pct % of the nearest one - and the batch is worked off in lane order with
      // scalar bit operations only.  After an evaluation the batch loses the leaves that have stopped being candidates.
      if (batch == 0ull) {
        if constexpr (NSUB > 1) {
          if (!leaf_cand(lb0)) lb0 = __builtin_inff();   // (for good: the sub-tiles' maxima only fall)
        }
        const float mn = wave_min_f32(lb0);
        if (!(mn * SAFE < M)) {
          if (++lvl > top) break;
          continue;
        }
        const bool in_batch = lb0 <= mn * batch_scale;
        batch = __ballot(in_batch);
        if (in_batch) lb0 = __builtin_inff();  // visited
      }
      const int j = __builtin_ctzll(batch);
      batch &= batch - 1ull;
      const int64_t c = grp0 * FAN + j;
      ++n_leaf_test;
      // can any sample of any lane still improve against leaf c?  (its box comes from lane j)
      float blo[DIM], bhi[DIM];
#if FLOODER_SORTED_LDSBOX
#pragma unroll
      for (int k = 0; k < DIM; ++k) {   // (wave-uniform address: broadcast reads)
        blo[k] = s_box[wv][j][k];
        bhi[k] = s_box[wv][j][DIM + k];
      }
#else
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        blo[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_lo[k]), j));
        bhi[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_hi[k]), j));
      }
#endif
      bool need = false;
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        float lbp = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float gap = __builtin_fmaxf(__builtin_fmaxf(blo[k] - p[i][k], p[i][k] - bhi[k]), 0.f);
          lbp = __builtin_fmaf(gap, gap, lbp);
        }
        need = need || (alive[i] && lbp * SAFE < best[i]);
      }
      SPHASE(4);
      if (__ballot(need) == 0ull) continue;
      ++n_leaf_eval;
#ifdef FLOODER_SORTED_TIMERS
      ++evals_in_group;
#endif
      const float* cp = pts + c * (int64_t)LEAF * DP;
      // rows stream through SGPRs (scalar loads), UB at a time (FLOODER_SORTED_UB8 above)
      constexpr int UB = FLOODER_SORTED_UB8 ? 8 : (DP == 8 ? 4 : 8);
#pragma unroll
      for (int h = 0; h < LEAF; h += UB) {
        typename RowVec<DP>::type cc[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) cc[u] = load_uniform_row<DP>(cp + (h + u) * DP);
#pragma unroll
        for (int u = 0; u < UB; u += 2) {
#pragma unroll
          for (int i = 0; i < KS; ++i) {
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
      if constexpr (FUSED) {
        if (++evals_since >= sf.refresh) {  // (wave-uniform counter) the maxima have risen meanwhile
          evals_since = 0;
          read_thr();
        }
#pragma unroll
        for (int i = 0; i < KS; ++i) alive[i] = alive[i] && __float_as_uint(best[i]) > thr[i];
      }
      float bm = -1.f;
#pragma unroll
      for (int i = 0; i < KS; ++i) bm = __builtin_fmaxf(bm, alive[i] ? best[i] : -1.f);
      if constexpr (NSUB > 1) {
        const float sm = sub_reduce<true>(bm);
        M = -1.f;
#pragma unroll
        for (int q = 0; q < NSUB; ++q) {
          Mq[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), q * SUBSZ));
          M = __builtin_fmaxf(M, Mq[q]);
        }
        if (batch != 0ull) batch &= __ballot(leaf_cand(0.f));
      } else {
        M = wave_max_f32(bm);   // (nobody left alive: M = -1, every remaining bound test fails, the walk unwinds)
      }
      SPHASE(5);
    }
    if constexpr (FUSED) {
      // the samples still alive are exact: deliver (only where the value can raise the maximum last seen or read now)
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        if (alive[i]) {
          uint32_t m = mb[i];
          const int64_t s_i = (int64_t)(id[i] / (uint32_t)R);
          const uint32_t bb = __float_as_uint(best[i]);
          while (m) {
            const int f = __builtin_ctz(m);
            m &= m - 1u;
            uint32_t* w = sf.face_bits + sf.slot_of(s_i, f);
            if (bb > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, bb);
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < KS; ++i)
        if (live[i]) out_d2[id[i]] = __float_as_uint(best[i]);
    }
    const unsigned long long item_tests = n_leaf_test + n_node_test - tests_before;
    max_item_tests = item_tests > max_item_tests ? item_tests : max_item_tests;
  }
  if (stats && lane == 0 && (n_node_test | n_leaf_test) != 0ull) {
    atomicMax(&stats[3], max_item_tests);
    atomicAdd(&stats[0], n_leaf_eval);
    atomicAdd(&stats[1], n_leaf_test);
    atomicAdd(&stats[2], n_node_test);
#ifdef FLOODER_SORTED_TIMERS
    for (int i = 0; i < 6; ++i) atomicAdd(&stats[4 + i], t_ph[i]);
    atomicAdd(&stats[10], n_refine);
    atomicAdd(&stats[11], n_groups);
    atomicAdd(&stats[12], n_groups_empty);
    atomicAdd(&stats[13], n_groups_idle);
    atomicAdd(&stats[14], n_node_spared);
#endif
  }
}
#undef SPHASE

template <int DIM>
struct SampleKeysOp {
  static int run(const float* verts, const float* weights, int k1, int R, int64_t n_samples, const float* box,
                 uint32_t* keys, const uint8_t* late, hipStream_t st) {
    int64_t blocks = (n_samples + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL((sample_keys_kernel<DIM>), dim3((int)blocks), dim3(256), 0, st, verts, weights, k1, R,
                       n_samples, box, keys, g_curve, late);
    return check_launch("sample_keys");
  }
};

template <int DIM>
struct SweepSortedOp {
  template <int KS, bool FUSED>
  static int launch(const float* pts, const float* nodes, const Levels& lv, const float* verts, const float* weights,
                    int k1, int R, int64_t n_samples, const uint32_t* order, int32_t* queue, uint32_t* out,
                    unsigned long long* stats, SortedFaces sf, TileShard ts, hipStream_t st) {
    // persistent blocks of 4 independent waves, as many as the registers let a CU hold (asked once per instantiation)
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sweep_sorted_kernel<DIM, KS, FUSED>, 256, 0) != hipSuccess || nb < 1) nb = 4;
      blocks_per_cu = nb > 8 ? 8 : nb;
    }
    int64_t n_tiles = (n_samples + 64 * KS - 1) / (64 * KS);
    if (ts.world > 1) {
      n_tiles = n_tiles * (ts.rank + 1) / ts.world - n_tiles * ts.rank / ts.world;
      if (n_tiles == 0) return FLOODER_OK;
    }
    // (option "sorted_blocks": fewer resident blocks per CU than fit - a diagnostic for how much of the kernel is latency)
    int64_t grid = (int64_t)(g_sorted_blocks > 0 && g_sorted_blocks < blocks_per_cu ? g_sorted_blocks : blocks_per_cu) * 256;
    if (grid * 4 > n_tiles) grid = (n_tiles + 3) / 4;
    hipLaunchKernelGGL((sweep_sorted_kernel<DIM, KS, FUSED>), dim3((unsigned)grid), dim3(256), 0, st, pts, nodes, lv, verts,
                       weights, k1, R, n_samples, order, queue, out, stats, g_bvh_refine_pct,
                       (float)(g_sorted_batch_pct < 100 ? 100 : g_sorted_batch_pct) * 0.01f, sf, ts);
    return check_launch("sweep_sorted");
  }
  static int run(const float* pts, const float* nodes, const Levels& lv, const float* verts, const float* weights,
                 int k1, int R, int64_t n_samples, const uint32_t* order, int32_t* queue, uint32_t* out,
                 unsigned long long* stats, SortedFaces sf, TileShard ts, hipStream_t st) {
    if (sf.face_bits != nullptr)
      return launch<1, true>(pts, nodes, lv, verts, weights, k1, R, n_samples, order, queue, out, stats, sf, ts, st);
    if (g_sorted_ks == 2) return launch<2, false>(pts, nodes, lv, verts, weights, k1, R, n_samples, order, queue, out, stats, sf, ts, st);
    return launch<1, false>(pts, nodes, lv, verts, weights, k1, R, n_samples, order, queue, out, stats, sf, ts, st);
  }
};

}  // namespace

extern "C" {

int flooder_sorted_tile_samples(void) { return 64 * (g_sorted_ks == 2 ? 2 : 1); }

int flooder_sample_key_bits(int dim) {
  if (dim < 1 || dim > FLOODER_MAX_DIM) return 0;
  const int b = 32 / dim > 10 ? 10 : 32 / dim;
  return b * dim;
}

int flooder_sample_keys_f32(const float* verts, const float* weights, int k1, int R, int64_t n_simplices, int dim,
                            const float* box, uint32_t* keys, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!verts || !weights || !box || !keys || k1 < 1 || k1 > FLOODER_MAX_VERTS || R < 0 || n_simplices < 0 ||
      n_simplices * (int64_t)R > 0xfffffffeLL)
    return fail(FLOODER_E_ARG, "flooder_sample_keys_f32: bad argument (or more than 2^32 - 2 samples)");
  return dispatch_dim<SampleKeysOp>(dim, verts, weights, k1, R, n_simplices * (int64_t)R, box, keys,
                                    (const uint8_t*)nullptr, (hipStream_t)stream);
}

int flooder_sample_keys_late_f32(const float* verts, const float* weights, int k1, int R, int64_t n_simplices, int dim,
                                 const float* box, const uint8_t* late_rows, uint32_t* keys, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!verts || !weights || !box || !keys || !late_rows || k1 < 1 || k1 > FLOODER_MAX_VERTS || R < 0 || n_simplices < 0 ||
      n_simplices * (int64_t)R > 0xfffffffeLL)
    return fail(FLOODER_E_ARG, "flooder_sample_keys_late_f32: bad argument (or more than 2^32 - 2 samples)");
  return dispatch_dim<SampleKeysOp>(dim, verts, weights, k1, R, n_simplices * (int64_t)R, box, keys, late_rows,
                                    (hipStream_t)stream);
}

int flooder_sweep_bvh_sorted_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                 const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                                 const int32_t* sample_order, int32_t* queue, uint32_t* out_d2, uint64_t* stats,
                                 void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !sample_order || !queue || !out_d2 || n_pts < 1 || k1 < 1 ||
      k1 > FLOODER_MAX_VERTS || R < 0 || n_simplices * (int64_t)R > 0xfffffffeLL)
    return fail(FLOODER_E_ARG, "flooder_sweep_bvh_sorted_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<SweepSortedOp>(dim, pts_sorted, nodes, lv, verts, weights, k1, R, n_simplices * (int64_t)R,
                                     reinterpret_cast<const uint32_t*>(sample_order), queue, out_d2,
                                     reinterpret_cast<unsigned long long*>(stats), SortedFaces{nullptr, nullptr, nullptr, 0, 0},
                                     TileShard{0, 1}, (hipStream_t)stream);
}

int flooder_sweep_bvh_sorted_shard_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                       const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                                       const int32_t* sample_order, int shard_rank, int shard_world, int32_t* queue,
                                       uint32_t* out_d2, uint64_t* stats, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !sample_order || !queue || !out_d2 || n_pts < 1 || k1 < 1 ||
      k1 > FLOODER_MAX_VERTS || R < 0 || n_simplices * (int64_t)R > 0xfffffffeLL || shard_world < 1 || shard_rank < 0 ||
      shard_rank >= shard_world)
    return fail(FLOODER_E_ARG, "flooder_sweep_bvh_sorted_shard_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<SweepSortedOp>(dim, pts_sorted, nodes, lv, verts, weights, k1, R, n_simplices * (int64_t)R,
                                     reinterpret_cast<const uint32_t*>(sample_order), queue, out_d2,
                                     reinterpret_cast<unsigned long long*>(stats), SortedFaces{nullptr, nullptr, nullptr, 0, 0},
                                     TileShard{shard_rank, shard_world}, (hipStream_t)stream);
}

int flooder_sweep_bvh_sorted_faces_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                       const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                                       const int32_t* sample_order, int32_t* queue, const uint32_t* memb, int n_faces,
                                       uint32_t* face_bits, const int32_t* face_slot, uint64_t* stats, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !sample_order || !queue || !memb || !face_bits || n_pts < 1 || k1 < 1 ||
      k1 > FLOODER_MAX_VERTS || R < 0 || n_faces < 1 || n_faces > 32 || n_simplices * (int64_t)R > 0xfffffffeLL)
    return fail(FLOODER_E_ARG, "flooder_sweep_bvh_sorted_faces_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<SweepSortedOp>(dim, pts_sorted, nodes, lv, verts, weights, k1, R, n_simplices * (int64_t)R,
                                     reinterpret_cast<const uint32_t*>(sample_order), queue, (uint32_t*)nullptr,
                                     reinterpret_cast<unsigned long long*>(stats),
                                     SortedFaces{memb, face_bits, face_slot, n_faces, g_sorted_refresh}, TileShard{0, 1}, (hipStream_t)stream);
}

}  // extern "C"
