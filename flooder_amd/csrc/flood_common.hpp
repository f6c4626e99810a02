// flood_common.hpp - helpers shared by the kernels of libflooder_hip.so (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/flooder_hip.h"

namespace flooder {

constexpr uint32_t INF_BITS = 0x7f800000u;

extern int g_bvh_ks;
extern int g_bvh_subs;
extern int g_bvh_grid;
extern int g_cell_grid;
extern int g_cell_exh_dense;
extern int g_bvh_leaf_batch;
extern int g_bvh_refine_pct;
extern int g_cell_exh_sparse;
extern int g_cell_brute_max;
extern int g_finish_focus_pct;
extern int g_finish_refresh;
extern int g_cell_tries;
extern int g_cell_super_weight;
extern int g_cell_super_n0;
extern int g_cell_super_sparse;
extern int g_cell_super_min_chunks;
extern int g_cell_density_grid;
extern int g_cell_chunks_per_block;  // cell sweep: chunks per persistent workgroup that size a short queue's launch
extern int g_cell_weight_classes;   // cell sweep: the simplex lists in descending weight class (0: in the given order)
extern int g_cell_listed_first;     // cell sweep, chunk launch: deferred chunks ahead of the heavy simplices
extern int g_cell_tail_waves;       // cell_tiles = 2: the tail = the last (this percentage of the launch's waves) items
extern int g_cell_chunk_major;    // cell sweep, chunk launch: heavy simplices chunk by chunk (all first chunks, then all second ones ...)
extern int g_cell_chunk_major_max;  // ... also for heavy lists of at least this many chunks (long queues: cfg 5)
extern int g_cell_drop;           // cell query drops interior samples that cannot raise the simplex's maximum
extern int g_cell_one_pass;         // cell sweep: dense chunks are classified and evaluated in ONE pass over their candidates
extern int g_cell_queue_block;     // cell sweep: log2 of the item blocks of the XCD-local work queue (-1: interleaved items, any XCD)
extern int g_cell_min_grid;          // ... and the smallest launch  // > 0: the cell sweep reads the local density from the index's density grid where every probed cell holds at least this many points (no first tree walk there); 0: never
extern int g_sorted_ks;
extern int g_sorted_blocks;
extern int g_sorted_batch_pct;  // sorted sweep: a batch of leaf tests = the bounds within this % of the nearest one (100: one leaf)
extern int g_sorted_refresh;  // fused sorted sweep: evaluated leaves between two readings of the face maxima
extern int g_cell_tiles;   // 1: dense chunks hand their tiles to a third launch (one sample per lane) instead of the exhaustive loop (measured slower: off)
extern int g_curve_bits;
extern int g_cell_exh_tries;
extern int g_cell_retry_keep;  // ... and the attempt may have kept at most this many points per chunk
extern int g_cell_retry_pct;  // share (percent) of a chunk's open samples that must have a point within twice the cell size for a second try (0: always)
extern int g_finish_items_cap;
extern int g_finish_budget_min;  // leaves a tile of a SHORT list may evaluate before it counts as hard
extern int g_cell_surface_pct;    // cell sweep: one cell size per chunk on clouds with less than this percentage of their points in interior cells of the density grid (0: never)
extern int g_cell_split_launches;  // cell sweep, long queue: light / heavy lists by 1 launch (class_order_kernel) or the 2 of rounds 3 - 5
extern int g_sort_shape;  // flooder_index_sort_zeroed: block shape of the radix passes (0: by cloud size; 1 small, 2 the library's, 3 large)
extern int g_finish_wide_points;  // clouds of at least this many points run the finish's per-wave passes with 8 waves per workgroup (0: never)
extern int g_finish_budget;  // scale of the leaf budget beyond which a tile of the finish counts as hard (0: off)
extern int g_finish_top;     // 1: the finish settles one sample per simplex (its largest bound) before everything else
extern int g_finish_order;   // 1: the finish works the flagged tiles off by descending probe bound
extern int g_fps_switch;
extern int g_fps_rpl;
extern int g_fps_lane_best;  // 1: batched FPS ranks one candidate per lane at most (the > 64 candidates path; test hook)
extern int g_fps_rounds;  // 1: batched FPS enqueues rounds of launches and reads the counter back between them
extern int g_curve;
extern int g_wit_weight;    // witness sweep: heaviest simplex (points in its box) it tries
extern int g_wit_cmax_pct;  // ... its gather radius in percent of the local point spacing
extern int g_wit_grid;      // ... its persistent one-wave workgroups
extern int g_wit_cmax_ext_pct;
extern int g_wit_flags;
extern int g_wit_adaptive;
extern int g_wit_max_open;
extern int g_wit_max_eval;
extern int g_wit_max_leaves;
extern int g_wit_max_in_pct;
extern int g_wit_max_live_pct;
extern int g_wit_min_bins;  // ... excess bins (of 64) the stage must hold at least
extern int g_wit_surface_pct;  // ... and no attempt on a cloud with less than this percentage of its points in interior cells (0: always)
// face planes of every simplex (flood_cell.hip: simplex_planes_kernel), 24 floats per simplex
int launch_simplex_planes(int dim, const float* verts, int k1, int64_t n_simplices, float* tab, hipStream_t st);
// the witness sweep's entry has just filled `tab` for these simplices on this stream: the cell sweep's entry, called
// next with the same buffers, skips its own launch (consumed by the first match; any other call clears it)
void planes_done_for(const float* verts, const float* tab, int64_t n_simplices, hipStream_t st);
bool planes_are_done(const float* verts, const float* tab, int64_t n_simplices, hipStream_t st);
char* err_buf();
int fail(int code, const char* msg);
int check_launch(const char* what);

__host__ __device__ constexpr int padded_dim(int dim) { return dim <= 2 ? 2 : (dim <= 4 ? 4 : 8); }

// One padded row (DP floats) with a single vector load (per-lane address).
template <int DP>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&out)[DP]) {
  if constexpr (DP == 2) {
    float2 t = *reinterpret_cast<const float2*>(p);
    out[0] = t.x; out[1] = t.y;
  } else if constexpr (DP == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
  } else {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
    out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
  }
}

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int64_t wave_uniform64(int64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(v & 0xffffffffu));
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Wave-uniform rows come in through the scalar cache: the pointer is cast to the constant address
// space so each padded row is one s_load_dwordx2/x4/x8 into SGPRs, and the VALU reads the
// coordinates as scalar operands (no LDS staging, no barrier, no VGPR per candidate).
typedef float v2f __attribute__((ext_vector_type(2)));

template <int DP>
struct RowVec {
  typedef float type __attribute__((ext_vector_type(DP)));
};

template <int DP>
__device__ __forceinline__ typename RowVec<DP>::type load_uniform_row(const float* p) {
  typedef typename RowVec<DP>::type vec_t;
  typedef const __attribute__((address_space(4))) vec_t* cptr_t;
  return *((cptr_t)(uintptr_t)p);
}

// ---- sharded work queue of the persistent kernels.  One returning atomic on ONE head word serves ~88 pops per
// microsecond chip-wide (MI355X_MICROARCH.md, "dequeue") - half a million chunks of a cfg 5 sweep would spend 5.7 ms
// just queueing for it.  FLOODER_QUEUE_SHARDS heads, 128 B apart; item i lives in shard i % SHARDS (every shard keeps
// the global order); a wave starts at its home shard and moves on, for good, when a shard is exhausted.
constexpr int QSHARDS = FLOODER_QUEUE_SHARDS;
constexpr int QSTRIDE = FLOODER_QUEUE_WORDS / FLOODER_QUEUE_SHARDS;
__device__ __forceinline__ int64_t queue_pop(int32_t* __restrict__ heads, int& shard, int& tried, int64_t n_items, int lane) {
  // (shard and tried are wave-uniform by construction; say so, or they cost vector registers in kernels that sit at
  // their register limit: the cell sweep went from 0 to 21 spilled VGPRs = 91 MB of scratch writes per launch)
  shard = __builtin_amdgcn_readfirstlane(shard);
  tried = __builtin_amdgcn_readfirstlane(tried);
  while (tried < QSHARDS) {
    int j = 0;
    if (lane == 0) j = atomicAdd(&heads[shard * QSTRIDE], 1);
    j = __builtin_amdgcn_readfirstlane(j);
    const int64_t item = (int64_t)j * QSHARDS + shard;
    if (item < n_items) return item;
    shard = shard + 1 == QSHARDS ? 0 : shard + 1;
    ++tried;
  }
  return -1;
}

// ---- the same queue, XCD-local.  The eight XCDs have an L2 each (4 MB, not coherent with the others); items that
// follow each other in a queue are neighbours in space - the ~20 chunks of a simplex share most of their leaves, the
// next simplex of the axis-ordered queue many - and the interleaved queue above deals them to all eight L2s (cfg 5:
// 17 GB of L2 misses per launch for a 256 MB cloud).  Here item blocks of 2^blk CONSECUTIVE items go to the shards in
// turn, shards 2x and 2x+1 are the home of the waves of XCD x (HW_REG_XCC_ID), and a wave leaves its XCD's shards
// only when both are dry: j-th pop of shard h -> item ((j >> blk) * QSHARDS + h) << blk | (j & (2^blk - 1)), which
// grows with j, so a shard is exhausted at the first item past the end.  Shards are visited in the order
// home ^ 0, 1, 2 ...: the partner shard of the own XCD first, then another XCD's pair, a different one for every XCD.
__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7; }  // hwreg(HW_REG_XCC_ID, 0, 4)
static_assert(FLOODER_QUEUE_SHARDS == 16, "two shards per XCD");
__device__ __forceinline__ int queue_home_local(int wave_in_block) { return xcc_id() * 2 + (wave_in_block & 1); }
__device__ __forceinline__ int64_t queue_pop_local(int32_t* __restrict__ heads, int home, int& tried, int64_t n_items, int lane, int blk) {
  home = __builtin_amdgcn_readfirstlane(home);
  tried = __builtin_amdgcn_readfirstlane(tried);
  while (tried < QSHARDS) {
    const int shard = home ^ tried;
    int j = 0;
    if (lane == 0) j = atomicAdd(&heads[shard * QSTRIDE], 1);
    j = __builtin_amdgcn_readfirstlane(j);
    const int64_t item = ((((int64_t)(j >> blk) * QSHARDS + shard) << blk) | (int64_t)(j & ((1 << blk) - 1)));
    if (item < n_items) return item;
    ++tried;
  }
  return -1;
}

// ---- 64-lane reductions on the DPP network (no LDS): result is wave-uniform.
// FLOODER_DPP_ASM (default): every step is ONE instruction - the operation itself with the DPP lane pattern on its
// first operand (v_min_f32_dpp x, x, x quad_perm:...; a lane whose source lane is masked off keeps x).  Written through
// __builtin_amdgcn_update_dpp the compiler emits a copy, the permuting move, a canonicalising max (floats) or a
// compare + select (unsigned) and the operation: 3 - 4 vector instructions per step, 18 - 24 per reduction, in
// kernels whose searches are chains of such reductions.  (s_nop 1: the two wait states a DPP read needs after the
// vector write of its source, which the compiler cannot place inside an asm block; the block OPENS with s_nop 4: five
// wait states cover a VALU write of EXEC directly in front of it as well - hipcc's wave64 control flow writes EXEC with
// scalar instructions today, tests/test_host.py greps the ISA for v_cmpx, but nothing here depends on that any more.)
#ifndef FLOODER_DPP_ASM
#define FLOODER_DPP_ASM 1
#endif
#define FLOODER_DPP_REDUCE(OP, x)                                                   \
  asm volatile("s_nop 4\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
               "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
               "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
               "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
               "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"        \
               "s_nop 1"                                                            \
               : "+v"(x))
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}

__device__ __forceinline__ float wave_min_f32(float x) {
#if FLOODER_DPP_ASM
  FLOODER_DPP_REDUCE("v_min_f32_dpp", x);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
#endif
  x = __builtin_fminf(x, dpp_move<0xB1, 0xF>(x));   // quad_perm [1,0,3,2]
  x = __builtin_fminf(x, dpp_move<0x4E, 0xF>(x));   // quad_perm [2,3,0,1]
  x = __builtin_fminf(x, dpp_move<0x141, 0xF>(x));  // row_half_mirror
  x = __builtin_fminf(x, dpp_move<0x140, 0xF>(x));  // row_mirror: every row of 16 holds its min
  x = __builtin_fminf(x, dpp_move<0x142, 0xA>(x));  // row_bcast15 into rows 1 and 3
  x = __builtin_fminf(x, dpp_move<0x143, 0xC>(x));  // row_bcast31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

__device__ __forceinline__ float wave_max_f32(float x) {
#if FLOODER_DPP_ASM
  FLOODER_DPP_REDUCE("v_max_f32_dpp", x);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
#endif
  x = __builtin_fmaxf(x, dpp_move<0xB1, 0xF>(x));
  x = __builtin_fmaxf(x, dpp_move<0x4E, 0xF>(x));
  x = __builtin_fmaxf(x, dpp_move<0x141, 0xF>(x));
  x = __builtin_fmaxf(x, dpp_move<0x140, 0xF>(x));
  x = __builtin_fmaxf(x, dpp_move<0x142, 0xA>(x));
  x = __builtin_fmaxf(x, dpp_move<0x143, 0xC>(x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_move_u32(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
#if FLOODER_DPP_ASM
  FLOODER_DPP_REDUCE("v_min_u32_dpp", x);
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
#endif
  uint32_t t;
  t = dpp_move_u32<0xB1, 0xF>(x); x = t < x ? t : x;
  t = dpp_move_u32<0x4E, 0xF>(x); x = t < x ? t : x;
  t = dpp_move_u32<0x141, 0xF>(x); x = t < x ? t : x;
  t = dpp_move_u32<0x140, 0xF>(x); x = t < x ? t : x;
  t = dpp_move_u32<0x142, 0xA>(x); x = t < x ? t : x;
  t = dpp_move_u32<0x143, 0xC>(x); x = t < x ? t : x;
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
#if FLOODER_DPP_ASM
  FLOODER_DPP_REDUCE("v_max_u32_dpp", x);
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
#endif
  uint32_t t;
  t = dpp_move_u32<0xB1, 0xF>(x); x = t > x ? t : x;
  t = dpp_move_u32<0x4E, 0xF>(x); x = t > x ? t : x;
  t = dpp_move_u32<0x141, 0xF>(x); x = t > x ? t : x;
  t = dpp_move_u32<0x140, 0xF>(x); x = t > x ? t : x;
  t = dpp_move_u32<0x142, 0xA>(x); x = t > x ? t : x;
  t = dpp_move_u32<0x143, 0xC>(x); x = t > x ? t : x;
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t x) {
#if FLOODER_DPP_ASM
  FLOODER_DPP_REDUCE("v_or_b32_dpp", x);
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
#endif
  x |= dpp_move_u32<0xB1, 0xF>(x);
  x |= dpp_move_u32<0x4E, 0xF>(x);
  x |= dpp_move_u32<0x141, 0xF>(x);
  x |= dpp_move_u32<0x140, 0xF>(x);
  x |= dpp_move_u32<0x142, 0xA>(x);
  x |= dpp_move_u32<0x143, 0xC>(x);
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// Fused per-face maxima (flooder_sweep_cell_faces_f32 / flooder_finish_faces_f32): memb[r] = bit mask of the
// faces sample row r lies on, face_bits[s * n_faces + f] = running maximum of the d2 bits over the samples of
// face f whose nearest neighbour is settled.  In this mode the d2 buffer is scratch: only tiles handed to the
// finish are written, and bit 31 of a word marks a sample the cell sweep has already settled.
constexpr uint32_t SETTLED_BIT = 0x80000000u;
struct FaceAcc {
  const uint32_t* memb;
  uint32_t* face_bits;
  int n_faces;
  // per simplex: (largest upper bound of an open sample << 32 | tile id) of its flagged tiles, and the list of
  // simplices that have one; filled by the cell sweep's probe, read by the finish (may be null: no probe)
  unsigned long long* top;
  int32_t* top_list;
  int32_t* top_count;
  // slot of (simplex s, face f) in face_bits: slot[s * n_faces + f], or s * n_faces + f when null.  A triangle, an
  // edge or a vertex is a face of several simplices and its samples have bit-identical coordinates from each of them
  // (the zero weights contribute exact zeros): with one slot per DISTINCT face the first simplex that settles a face
  // lets every other one drop that face's samples in the finish.
  const int32_t* slot;
  // cell sweep -> finish: the probe's bound of every flagged tile (parallel to the flag list) and a histogram of the
  // bounds' top 12 bits: the finish starts the tiles with the largest bounds - the longest searches - first
  uint32_t* flag_key = nullptr;
  int32_t* flag_hist = nullptr;
  __device__ __forceinline__ int64_t slot_of(int64_t s, int f) const {
    return slot ? (int64_t)slot[s * (int64_t)n_faces + f] : s * (int64_t)n_faces + f;
  }
};

template <template <int> class F, typename... Args>
int dispatch_dim(int dim, Args&&... args) {
  switch (dim) {
    case 1: return F<1>::run(args...);
    case 2: return F<2>::run(args...);
    case 3: return F<3>::run(args...);
    case 4: return F<4>::run(args...);
    case 5: return F<5>::run(args...);
    case 6: return F<6>::run(args...);
    case 7: return F<7>::run(args...);
    case 8: return F<8>::run(args...);
    default: return fail(FLOODER_E_ARG, "dim must be in 1..8");
  }
}


// ---- what kind of cloud is this?  Four words behind the density grid (flooder_density_grid_words counts them): [2] =
// points in INTERIOR cells of a COARSE grid (16^3 / 64^2: four fine cells per axis pooled) - occupied cells whose axis
// neighbours inside the grid are all occupied -, [3] = all points.  A cloud that fills a volume (Gaussian, swiss
// cheese; 20 k to 16 M points) has 91 - 99 % of its points in interior cells, one that lies on a surface (the noisy
// torus) 7 - 21 %: there a second, doubled cell size per chunk of the cell sweep keeps thousands of points for samples
// the finish settles sooner (cfg 3 3.88 -> 3.75 ms with one try; cfg 2 1.18 -> 1.20) - the sweep reads the two words
// and decides (option "cell_surface_pct").  Results do not depend on it.  (The fine grid itself will not do: at a
// million points it holds a quarter of a leaf per cell, and a uniform cloud looks as hollow as a surface.)
// 16 workgroups of 256 threads, no communication between them: workgroup b owns the coarse slab z = b (3-D; in the
// plane: rows 4b .. 4b+3), a thread pools its own cell AND its two neighbours along z (y) - 48 (12) vector loads in
// flight together, one memory round trip -, the neighbours along the other axes come from the workgroup's LDS; two
// atomics per workgroup.  flooder_index_rows_f32 lets them ride in spare workgroups of the launch that builds the
// first inner tree level (a launch of their own after the index: +17 us; on a side stream: +40 us of cross-stream
// dependencies; coarse counts added atomically by the leaf pass: +46 us on the Gaussian, whose dense core hits a few
// hundred addresses: all measured); flooder_cloud_kind is the stand-alone form.
constexpr int KIND_WORDS = 4;
template <int DIM>
__device__ __forceinline__ void cloud_kind_block(const int32_t* __restrict__ fine, int32_t* __restrict__ kind, int block, int tid,
                                                 int* s_own, int* s_red) {
  constexpr int G = DIM == 2 ? 64 : 16;      // coarse cells per axis
  constexpr int GF = 4 * G;                  // fine cells per axis
  // coarse cell of this thread: (cx, cy [, cz]); the axis handled through the thread's own extra sums is the LAST one
  // in 3-D (z = block) and y in the plane (y = 4 * block + tid / 64)
  const int cx = DIM == 3 ? (tid & 15) : (tid & 63);
  const int cy = DIM == 3 ? (tid >> 4) : (4 * block + (tid >> 6));
  const int cz = DIM == 3 ? block : 0;
  const int own_axis = DIM == 3 ? cz : cy;
  int4 r[3][DIM == 3 ? 16 : 4];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int a = own_axis + d - 1;          // the pooled cell along the own axis: below, own, above
    const bool ok = a >= 0 && a < G;
    if constexpr (DIM == 3) {
#pragma unroll
      for (int z = 0; z < 4; ++z)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          r[d][4 * z + y] = ok ? *reinterpret_cast<const int4*>(fine + ((4 * a + z) * GF + (4 * cy + y)) * GF + 4 * cx)
                               : int4{0, 0, 0, 0};
    } else {
#pragma unroll
      for (int y = 0; y < 4; ++y)
        r[d][y] = ok ? *reinterpret_cast<const int4*>(fine + (4 * a + y) * GF + 4 * cx) : int4{0, 0, 0, 0};
    }
  }
  int sum[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int t = 0;
#pragma unroll
    for (int q = 0; q < (DIM == 3 ? 16 : 4); ++q) t += r[d][q].x + r[d][q].y + r[d][q].z + r[d][q].w;
    sum[d] = t;
  }
  s_own[tid] = sum[1];
  __syncthreads();
  const int c = sum[1];
  bool in = c > 0;
  in = in && (own_axis == 0 || sum[0] > 0) && (own_axis == G - 1 || sum[2] > 0);
  in = in && (cx == 0 || s_own[tid - 1] > 0) && (cx == G - 1 || s_own[tid + 1] > 0);
  if constexpr (DIM == 3) in = in && (cy == 0 || s_own[tid - 16] > 0) && (cy == 15 || s_own[tid + 16] > 0);
  int inner = in ? c : 0, all = c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    inner += __shfl_xor(inner, o);
    all += __shfl_xor(all, o);
  }
  if ((tid & 63) == 0) { s_red[tid >> 6] = inner; s_red[4 + (tid >> 6)] = all; }
  __syncthreads();
  if (tid == 0) {
    const int a = s_red[0] + s_red[1] + s_red[2] + s_red[3], b = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    if (b != 0) {
      atomicAdd(&kind[2], a);
      atomicAdd(&kind[3], b);
    }
  }
}
template <int DIM>
__global__ __launch_bounds__(256) void cloud_kind_kernel(const int32_t* __restrict__ fine, int32_t* __restrict__ kind) {
  __shared__ int s_own[256];
  __shared__ int s_red[8];
  cloud_kind_block<DIM>(fine, kind, blockIdx.x, threadIdx.x, s_own, s_red);
}
inline void launch_cloud_kind(int dim, int32_t* grid, hipStream_t st) {   // grid: the fine grid; the words sit behind it
  if (dim == 2) hipLaunchKernelGGL((cloud_kind_kernel<2>), dim3(16), dim3(256), 0, st, grid, grid + 256 * 256);
  else if (dim == 3) hipLaunchKernelGGL((cloud_kind_kernel<3>), dim3(16), dim3(256), 0, st, grid, grid + 64 * 64 * 64);
}

// flood_wit.hip: flooder_sweep_witness_f32 with the index's density grid (what flooder_fused_witness calls)
int sweep_witness(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes, const float* verts,
                  const float* weights, int k1, int R, int64_t n_simplices, const int32_t* coarse_rows, int n_coarse,
                  const uint32_t* parents, int32_t* queue, uint32_t* d2_scratch, const uint32_t* memb, int n_faces,
                  uint32_t* face_bits, const int32_t* face_slot, int32_t* flag_list, int32_t* flag_count,
                  uint32_t* flag_key, int32_t* flag_hist, uint64_t* top, int32_t* top_list, int32_t* top_count,
                  float* simplex_weight, int32_t* item_list, float* plane_scratch, uint64_t* stats,
                  const int32_t* density_grid, void* stream);

}  // namespace flooder
