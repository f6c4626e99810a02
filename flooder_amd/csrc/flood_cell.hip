// flood_cell.hip - coverage sweep through wave-local cell grids in LDS (gfx950; dim 2 and 3).
//
// One WAVE owns a chunk of 256 consecutive samples of one simplex (4 per lane; the host orders the
// barycentric weights by recursive bisection - core.sample_order - so a chunk is a compact patch of the
// simplex).  No workgroup barrier is used anywhere: the four waves of a block are independent and each has
// its own LDS partition.
//
//   0. samples   rebuilt in registers from vertices x weights; chunk bounding box by DPP reductions.
//   1. density   the box tree over the curve-sorted (Hilbert) cloud is walked breadth-first (lane = child
//                box) for the leaves overlapping the chunk box; N0 = points inside the box gives the local
//                spacing h = (V / N0)^(1/3) and the first cell size c = alpha * h.
//   2. stage     leaves overlapping the box grown by c are gathered again; their points are filtered
//                (box grown by c, within c of every face plane of the simplex) and counting-sorted by
//                cell into the wave's LDS partition (<= 480 points, <= 10^3 cells).
//   3. query     each lane visits, for each of its samples, the 3^dim cells around it - 3^(dim-1)
//                contiguous runs of the staged list - and keeps the minimum direct-difference d^2.
//   4. verify    a minimum <= (0.999 c)^2 is provably the nearest neighbour (every point that close is
//                in the visited cells).  While unverified samples remain, c doubles and 2-4 repeat
//                for them (up to 3 times); what is still open - or does not fit the LDS stage - is
//                appended, per tile of 64 samples, to a work list that the exact tree sweep
//                (flood_bvh.hip, seeded with the minima found here) finishes.
// A bad c costs time, never correctness; the result equals the exhaustive minimum bit for bit.

#include <type_traits>

#include "flood_common.hpp"
#include "flood_bvh.hpp"
#include "flood_planes.hpp"

using namespace flooder;

namespace {

constexpr int SPL_CHUNK = 4;      // samples per lane of a chunk item (256 samples per wave)
#ifndef FLOODER_CELL_CAPW
#define FLOODER_CELL_CAPW 480
#endif
#ifndef FLOODER_CELL_MAXLEAF
#define FLOODER_CELL_MAXLEAF 584
#endif
#ifndef FLOODER_CELL_WAVES
#define FLOODER_CELL_WAVES 4
#endif
#ifndef FLOODER_CELL_COMPACT
#define FLOODER_CELL_COMPACT 0   // 1: the cell query compacts the samples that need more than their own row across lanes (measured: no gain, see there)
#endif
constexpr int CAPW = FLOODER_CELL_CAPW;  // points staged per wave (480 + 896 leaves: 13 KB per wave, see the kernel)
constexpr int MAXLEAF = FLOODER_CELL_MAXLEAF;  // leaves gathered per wave item (896: 14 K points before filtering)
constexpr int MAXFRONT = 192;     // inner nodes per level of the gather
constexpr int MAX_TRIES = 3;      // cell sizes tried per chunk at most (option cell_tries)
constexpr int EXH_MAX = 4 * CAPW;         // kept points evaluated exhaustively at most (sparse chunk box)
constexpr int UNR = 4;            // candidate rows in flight per lane in the staging loops
constexpr int KEEP0 = CAPW - CAPW / 3;      // first stage slot under the kept-candidate list (its CAPW ints = a third of the stage)
constexpr int BRUTE_CAP = KEEP0 - 4;  // compacted points the classification pass may leave in the stage

// Density grid of the cloud (flooder_density_grid_f32: G^DIM point counts over the cloud's box).  Where its cells are
// well filled a chunk reads the local density from the cells under its box - nine loads - instead of walking the
// tree for the leaves under the box and counting their points (15 % of the sweep's wave time).
struct DensGrid {
  const int32_t* grid = nullptr;  // G^DIM point counts
  const float* box = nullptr;     // the cloud's box (16 floats: [0:dim] min, [8:8+dim] max), device memory
  int min_count = 16;             // the grid is used where every probed cell holds at least this many points
  int one_pass = 125;             // (rides along: single-pass exhaustive evaluation of dense chunks, see the kernel;
                                  // percent of the give-up cap the extrapolated kept set may reach, 0 = off)
  int surface_pct = 0;            // (rides along: one cell size per chunk when fewer than this percentage of the cloud's
                                  // points sit in interior cells of the grid - the words behind it, cloud_kind_kernel)
};
template <int DIM>
struct DensCfg {
  static constexpr int G = DIM == 2 ? 256 : 64;   // cells per axis
  static constexpr int NF = DIM == 2 ? G * G : G * G * G;
};

template <int DIM>
struct CellCfg {
  static constexpr int G = DIM == 2 ? 32 : 10;                  // cells per axis
  static constexpr int NC = DIM == 2 ? 32 * 32 : 10 * 10 * 10;  // cells in the grid
};

// Inclusive prefix sum over the 64 lanes on the DPP network: seven adds (three neighbours of the own row first, then
// shifts by 4 and 8 inside the row, then the totals of the rows before), no LDS - the __shfl_up form is six
// ds_bpermute round trips with a select each, and the cell table's prefix runs it sixteen times per chunk.
// (All 64 lanes must be active.  s_nop 1: the wait states of a DPP read after the write of its source; the block opens
// with s_nop 4 - the five a DPP read needs after a VALU write of EXEC, which the compiler's hazard recognizer cannot
// place inside an asm block.)
#ifndef FLOODER_DPP_SCAN
#define FLOODER_DPP_SCAN FLOODER_DPP_ASM
#endif
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#if FLOODER_DPP_SCAN
  int r;
  asm volatile("s_nop 4\n\t"
               "v_add_u32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_u32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_u32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "s_nop 1"
               : "=&v"(r)
               : "v"(v));
  return r;
#else
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
#endif
}

// LDS hand-off between lanes of ONE wave: make earlier LDS writes/atomics of every lane visible to
// later LDS reads of every lane (compiler and hardware ordering), without a workgroup barrier.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Row at a 32-bit BYTE offset from a wave-uniform base (the sweep's arrays stay below 4 GiB: checked at the entry
// point): one VGPR of address arithmetic per load instead of a 64-bit multiply-add.
template <int DP>
__device__ __forceinline__ void load_row_at(const float* __restrict__ base, uint32_t byte_off, float (&out)[DP]) {
  load_row<DP>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)byte_off), out);
}

// The point stage in LDS: blocks of FOUR points, 48 bytes each - x[4], y[4], z[4] - 12 bytes per point instead of a
// padded 16-byte row.  Four consecutive points are three 16-byte reads (they were four 12-byte ones), and the stage of
// a wave shrinks by a quarter: with it the kernel's LDS fits FOUR workgroups per CU (40 KB each) where 13 KB per wave
// allowed three - the sweep is bound by issue + dependent latency and a fourth wave per SIMD is worth 14 - 20 % of it.
// Readers take whole blocks: a list that starts inside a block is read from the block's start (the extra entries are
// real points of an earlier cell or list: a minimum over more real points is still a valid upper bound, and exact once
// verified), and up to three entries past a list's end are real points or the +inf pads behind it.
struct Stage {
  float* b;
  __device__ __forceinline__ void put(int slot, float x, float y, float z) const {
    float* q = b + (slot >> 2) * 12 + (slot & 3);
    q[0] = x;
    q[4] = y;
    q[8] = z;
  }
  __device__ __forceinline__ void pad(int n, int lane) const {  // four +inf entries behind a list of n
    if (lane < 4) put(n + lane, __builtin_inff(), __builtin_inff(), __builtin_inff());
  }
  // points j .. j+3, j a multiple of four (.w unused)
  __device__ __forceinline__ void get4(int j, float4 (&x)[4]) const {
    const float4* q = reinterpret_cast<const float4*>(b + (j >> 2) * 12);
    const float4 X = q[0], Y = q[1], Z = q[2];
    x[0] = make_float4(X.x, Y.x, Z.x, 0.f);
    x[1] = make_float4(X.y, Y.y, Z.y, 0.f);
    x[2] = make_float4(X.z, Y.z, Z.z, 0.f);
    x[3] = make_float4(X.w, Y.w, Z.w, 0.f);
  }
};

__device__ __forceinline__ int lane_rank(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

#ifdef FLOODER_PHASE_TIMERS
#define PHASE_T0() unsigned long long _t_prev = __builtin_amdgcn_s_memtime()
#define PHASE(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); t_phase[i] += _t - _t_prev; _t_prev = _t; } while (0)
#else
#define PHASE_T0() do {} while (0)
#define PHASE(i) do {} while (0)
#endif

// Chunks that a SUPER launch hands on to the per-chunk launch: entry = (simplex * chunks + chunk) << 1 | seeded;
// seeded: the (S, R) buffer holds the minima found so far and c[] the next cell size to try.
struct DeferList {
  int32_t* list;
  float* c;
  int32_t* count;
  // simplices split by their rough point count (flooder_simplex_weight_f32; both lists null: no split): the light
  // ones are worked off in runs of four chunks by the SUPER launch, the heavy ones - no run of theirs would fit the
  // stage - chunk by chunk by the second launch, ahead of the deferred chunks.  split[0], split[1] = list lengths.
  const int32_t* light;
  const int32_t* heavy;
  const int32_t* split;
  int n0_limit;         // a run with more points than this inside its box is not staged either
  // dense chunks: a per-chunk item whose kept points overflow the LDS stage hands its open tiles of 64 samples on to
  // a third launch (SPLV = 1: one sample per lane, a re-centred region a quarter the size, which mostly fits the stage)
  // instead of evaluating every kept point against every sample.  entry = (simplex * tiles + tile) << 1 | seeded.
  int32_t* tile_list = nullptr;
  float* tile_c = nullptr;
  int32_t* tile_count = nullptr;
  int64_t tail_items = INT64_MAX;  // only the last so many items of the chunk launch hand dense chunks on to the tiles
  int listed_first = 1;  // chunk launch: the deferred chunks ahead of the heavy simplices (0: behind them)
  int chunk_major = 0;   // chunk launch: pilots first - chunk 0 of every heavy simplex ahead of all other chunks (2: always)
  int pilot_min_items = 262144;  // ... also where the heavy list has at least this many chunks (option "cell_chunk_major_max")
  int drop = 0;          // cell query: interior samples whose running minimum cannot raise the simplex's maximum stop early
};

// The arguments of cell_sweep_kernel: ONE struct in the kernarg segment.  As separate kernel parameters they were
// ~116 SGPRs (Levels 26, DeferList 25, FaceAcc 19, ...) that the compiler loaded at the top of the kernel and kept
// alive across the persistent loop - with 102 SGPRs available, 215 of them were parked in VGPR lanes (v_writelane /
// v_readlane: 4 VGPRs of a kernel that sits at its 168-VGPR limit).  Now a field is read where it is used, through a
// pointer to the kernarg segment that an empty asm statement hides from the optimiser (ARG below): a scalar load
// from the scalar cache per use site and nothing to keep alive.
struct CellParams {
  const float* pts;
  const float* nodes;
  Levels lv;
  const float* verts;
  const float* plane_tab;
  const float* weights;
  int k1, R;
  int64_t n_simplices;
  float alpha;
  int exh_dense, exh_sparse, brute_max, max_tries, exh_tries, retry_pct, retry_keep;
  int32_t* queue;
  uint32_t* out_d2;
  int32_t* flag_list;
  int32_t* flag_count;
  unsigned long long* stats;
  FaceAcc acc;
  DeferList dl;
  DensGrid dg;
  int qblk;  // work queue: items are dealt to the shards in blocks of 2^qblk (flood_common.hpp: queue_pop_local); < 0: the interleaved queue
};
#define FLOODER_AS4 __attribute__((address_space(4)))
__device__ __forceinline__ const FLOODER_AS4 CellParams* cell_args() {
  const FLOODER_AS4 CellParams* p = (const FLOODER_AS4 CellParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));  // (a fresh pointer every time: loads through it are neither merged nor hoisted)
  return p;
}
template <typename T>
__device__ __forceinline__ T arg_copy(const FLOODER_AS4 T* p) {
  T t;
  __builtin_memcpy(&t, p, sizeof(T));
  return t;
}
#define ARG(field) arg_copy(&cell_args()->field)

// Outward unit normals of the faces of a full-dimensional simplex (face f is opposite vertex f), as planes
// pn . (x - org) <= po relative to the LOCAL origin org = vertex 0 (no cancellation for clouds far from the
// coordinate origin).  A point within c of a sample lies within c of every such half-space; the test in the sweep
// allows c plus the fp32 error of the plane itself: the direction of a cross product of two edges is off by
// about 4 eps / sin(angle between them), which the per-plane slack pslack (8 eps / sin) times the simplex
// extent covers.  Needle faces (sin < 1e-4), flat simplices (height below 1e-4 of the extent: the sign of
// `side` is not trustworthy) and degenerate faces switch their plane off.
// One table row per simplex (PLANE_ROW floats: org[3], sext, then pn[3], po, pslack per face), written by
// simplex_planes_kernel before the sweep: the same for all ~20 chunks of a simplex, and ~150 wave-uniform
// instructions with two square roots and a division that every chunk used to repeat on the vector ALU.
template <int DIM>
__global__ __launch_bounds__(256) void simplex_planes_kernel(const float* __restrict__ verts, int k1, int64_t n_simplices,
                                                             float* __restrict__ tab) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_simplices) return;
  simplex_planes_row<DIM>(verts, k1, s, tab);   // (flood_planes.hpp; PLANE_ROW floats per simplex)
}

// SUPER = true: a work item is a run of GS consecutive chunks of one simplex (1024 samples: with the bisection order
// of the samples still a compact patch).  The points around the whole run are gathered, filtered and staged ONCE
// (one density estimate, one cell size) and the chunks are then queried one after the other against that stage - the
// per-chunk cost of tree walks and classification, half of a chunk's instructions, is shared by four chunks.  Runs
// whose neighbourhood does not fit the stage, and chunks that keep open samples (they would need a larger cell
// size, i.e. a new stage), are appended to a deferred list that a SUPER = false launch works off chunk by chunk.
template <int DIM, bool SUPER, int SPLV>
__global__ __launch_bounds__(256, FLOODER_CELL_WAVES) void cell_sweep_kernel(CellParams args_in_the_kernarg_segment) {  // (3 waves per SIMD: what the 13 KB of LDS per wave allow)
  // Launch-invariant arguments that the whole item loop needs (the rest: ARG(field) where it is used)
  const int R = ARG(R), k1 = ARG(k1);
  const bool has_stats = ARG(stats) != nullptr, fused = ARG(acc.face_bits) != nullptr;
  constexpr int DP = padded_dim(DIM);
  constexpr int SPL = SPLV;          // samples per lane
  constexpr int CHUNK = 64 * SPL;    // samples per wave item (SPLV = 1: the tile launch)
  constexpr bool TILES = SPLV == 1;
  static_assert(!(SUPER && TILES), "runs of four are made of chunks");
  constexpr int GS = SUPER ? 4 : 1;  // chunks per work item
  constexpr int G = CellCfg<DIM>::G;
  constexpr int NC = CellCfg<DIM>::NC;
  // LDS per wave: 7.5 KB point stage + 2 KB cell table (16-bit entries, two per word) + 3.5 KB leaf list = 13 KB,
  // 52 KB per block, so THREE blocks (12 waves) fit a CU's 160 KB; the issue rate of a CU grows with the waves
  // it holds and this kernel is short of them.  Two lists live inside the point stage while it holds no points:
  // the gather's frontier (first 1.5 KB, gather phase only) and the kept-candidate list of the classification
  // pass (last quarter; read completely into registers before the first point is scattered).
  __shared__ __attribute__((aligned(16))) float s_pts_all[4][(CAPW + 4) * 3];  // (+4: the readers run up to three entries past a list's end)
  __shared__ uint32_t s_cell_all[4][(NC + 8) / 2];
  __shared__ int s_leaf_all[4][MAXLEAF];
  static_assert(2 * MAXFRONT * sizeof(int) <= KEEP0 * 12, "the frontier alias ends below the kept list");
  static_assert(CAPW % 12 == 0 && KEEP0 % 4 == 0 && CAPW <= 512, "kept-list alias on a block boundary and the 10-bit cell field");
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const Stage s_pts{s_pts_all[wv]};
  uint32_t* s_cell32 = s_cell_all[wv];
  const uint16_t* s_cell = reinterpret_cast<const uint16_t*>(s_cell32);  // [i+1]: count -> start -> end of cell i
  uint16_t* s_cell_w = reinterpret_cast<uint16_t*>(s_cell32);
  int* s_leaf = s_leaf_all[wv];
  int* s_keep = reinterpret_cast<int*>(s_pts.b + KEEP0 * 3);  // (candidate slot << 10) | cell
  int* s_front = reinterpret_cast<int*>(s_pts.b);
  // entry i of the 16-bit cell table: word i >> 1, half i & 1 (counts stay below 2^16: no carry into the neighbour)
  auto cell_add = [&](int i) -> int {
    const int sh = (i & 1) * 16;
    return (int)((atomicAdd(&s_cell32[i >> 1], 1u << sh) >> sh) & 0xffffu);
  };
  const int top = ARG(lv.n_levels) - 1;
  const int n_slots = R;  // sample slots per simplex
  const int chunks = (n_slots + CHUNK - 1) / CHUNK;
  const int tiles64 = (n_slots + 63) >> 6;
  const int supers = (chunks + GS - 1) / GS;
  // (split[2] == 0: lists not filled - not produced any more, kept for callers that zero the counts themselves; a
  // cloud too dense for runs of four, or a short queue, has every simplex on the heavy list)
  // (item counts of a launch fit 32 bits: the entry point checks S x tiles against 2^31)
  bool use_lists, chunk_major, lfirst;
  int n_heavy_items, n_listed, n_items;
  {
    const DeferList dl = ARG(dl);
    const int64_t n_simplices = ARG(n_simplices);
    use_lists = TILES || (!SUPER && dl.list && (!dl.split || dl.split[2] != 0));
    n_heavy_items = (!TILES && use_lists && dl.heavy) ? dl.split[1] * chunks : 0;
    // (decided once per launch, wave-uniform: see the item decoding below)
    chunk_major = wave_uniform((!SUPER && !TILES && dl.chunk_major && use_lists && dl.heavy && chunks > 1 &&
                                (dl.chunk_major == 2 || 3 * (int64_t)(n_heavy_items / chunks) < n_simplices ||
                                 n_heavy_items >= dl.pilot_min_items)) ? 1 : 0) != 0;
    // chunk launch: the chunks the runs deferred come FIRST - they are the long items of this launch (a neighbourhood
    // that overflowed the shared stage), and at the end of the queue they were its tail
    lfirst = TILES || dl.listed_first != 0;
    n_listed = TILES ? dl.tile_count[0] : (use_lists ? dl.count[0] : 0);
    n_items = TILES ? dl.tile_count[0]
              : SUPER ? (int)((dl.light ? (int64_t)dl.split[0] : n_simplices) * supers)
                      : (use_lists ? n_heavy_items + n_listed : (int)(n_simplices * chunks));
  }
  if (n_items == 0) return;  // (nothing for this launch: no need for 3072 waves to pop an empty queue)
  unsigned long long n_pairs = 0;
  unsigned n_staged = 0, n_flagged = 0, n_retries = 0;  // (per-wave diagnostic counts: 32 bits hold them)
#ifdef FLOODER_PHASE_TIMERS
  unsigned long long t_phase[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  unsigned g_gather0 = 0, g_gather = 0, g_cap = 0, g_tries = 0, g_brute = 0;
#ifdef FLOODER_WAVE_END
  const unsigned long long t_wave0 = __builtin_amdgcn_s_memrealtime();
#endif

  // sharded work queue (flood_common.hpp): XCD-local blocks of items, or (qblk < 0) interleaved items
  const int qblk = ARG(qblk);
  int q_shard = qblk >= 0 ? queue_home_local(wv) : (int)((blockIdx.x * 4 + wv) % QSHARDS), q_tried = 0;
  for (;;) {
    PHASE_T0();
#ifdef FLOODER_PHASE_TIMERS
    const unsigned long long t_chunk0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, chip-wide
    unsigned long long d_steps = 0;  // child-box tests of the tree gathers of this chunk
    unsigned long long t_phase0[12];
    for (int i = 0; i < 12; ++i) t_phase0[i] = t_phase[i];
#endif
    const int g = (int)(qblk >= 0 ? queue_pop_local(ARG(queue), q_shard, q_tried, n_items, lane, qblk)
                                  : queue_pop(ARG(queue), q_shard, q_tried, n_items, lane));
    if (g < 0) break;
    // hot arguments of this item (the cold ones are read where they are used: ARG)
    const float* __restrict__ const pts = ARG(pts);
    const float* __restrict__ const nodes = ARG(nodes);
    const float* __restrict__ const weights = ARG(weights);
    int s;
    int q;            // current chunk of the simplex
    int n_sub = 1;    // chunks of this work item
    bool seeded = false;
    float c_seed = 0.f;
    if constexpr (SUPER) {
      s = g / supers;
      q = (int)(g - s * supers) * GS;
      n_sub = chunks - q < GS ? chunks - q : GS;
      const int32_t* light = ARG(dl.light);
      if (light) s = light[s];
    } else if (use_lists && (lfirst ? g >= n_listed : g < n_heavy_items)) {
      const int gh = lfirst ? g - n_listed : g;
      const int32_t* heavy = ARG(dl.heavy);
      // (only where the heavy list is the small dense rest of a cloud whose sparse simplices the witness sweep has
      // taken - cfg 2: 1725 of 6052; with every simplex on the list the order cost a rank's share of cfg 3 half of its
      // sweep time again, for reasons not understood: 2.11 / 1.10 ms against 1.43 / 0.75 at W = 2 / 4)
      if (chunk_major) {
        // pilots first: chunk 0 of every heavy simplex, then the other chunks simplex by simplex in list order - by the
        // time the later chunks of a simplex are swept its first one has raised the running maximum of the full
        // simplex, against which interior samples are dropped in the cell query (below), and the long items (the
        // chunks of the densest simplices, first in the list) still start first.  (ALL chunks in chunk-major order
        // put the last chunk of the densest simplex at the very end of the queue: a rank's share of cfg 3 took 2.11
        // instead of 1.43 ms.)
        const int nh = n_heavy_items / chunks;
        if (gh < nh) {
          q = 0;
          s = heavy[gh];
        } else {
          const int g2 = gh - nh;
          const int si = g2 / (chunks - 1);
          q = 1 + (g2 - si * (chunks - 1));
          s = heavy[si];
        }
      } else {
        s = gh / chunks;
        q = gh - s * chunks;
        s = heavy[s];
      }
    } else if (use_lists) {
      const int gl = lfirst ? g : g - n_heavy_items;
      const int e = (TILES ? ARG(dl.tile_list) : ARG(dl.list))[gl];
      seeded = (e & 1) != 0;
      c_seed = (TILES ? ARG(dl.tile_c) : ARG(dl.c))[gl];
      s = (e >> 1) / chunks;
      q = (e >> 1) - s * chunks;
    } else {
      s = g / chunks;
      q = g - s * chunks;
    }
    s = wave_uniform(s);
    const int q_first = q;
    const float* vs = ARG(verts) + (int64_t)s * k1 * DIM;
    const int n_live = R;  // live slots of this simplex
    if (q * CHUNK >= n_live) continue;
    auto defer = [&](int qq, int seeded_flag, float c_next) {
      if (lane == 0) {
        const int pos = atomicAdd(ARG(dl.count), 1);
        ARG(dl.list)[pos] = (int)(((s * chunks + qq) << 1) | seeded_flag);
        ARG(dl.c)[pos] = c_next;
      }
    };
    PHASE(0);

    // ---- 0. samples of the current chunk q (rebuilt from vertices x weights)
    float p[SPL][DIM];
    float best[SPL];
    bool open[SPL];  // still unverified
    int row[SPL];    // row of the weight table / column of the output
    float blo[DIM], bhi[DIM];
    auto make_samples = [&]() {
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        int slot = q * CHUNK + i * 64 + lane;
        open[i] = slot < n_live;
        if (slot >= n_live) slot = n_live - 1;  // duplicate of the last live sample, never stored
        const int r = slot;
        row[i] = r;
#pragma unroll
        for (int k = 0; k < DIM; ++k) p[i][k] = 0.f;
        if (k1 == 4) {  // tetrahedra: the weight row is one 16 B load (same fma order as the general loop)
          const float4 w4 = *reinterpret_cast<const float4*>(weights + (int64_t)r * 4);
          const float wj[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < DIM; ++k) p[i][k] = __builtin_fmaf(wj[j], vs[j * DIM + k], p[i][k]);
          }
        } else {
          for (int j = 0; j < k1; ++j) {
            const float w = weights[(int64_t)r * k1 + j];
#pragma unroll
            for (int k = 0; k < DIM; ++k) p[i][k] = __builtin_fmaf(w, vs[j * DIM + k], p[i][k]);
          }
        }
        best[i] = __builtin_inff();
      }
    };
    // face planes of the simplex: one table row, six scalar 16-byte loads (simplex_planes_kernel above)
    float pn[DIM + 1][DIM], po[DIM + 1], pslack[DIM + 1], org[DIM];
    float sext;
    {
      const float* pt = ARG(plane_tab) + (int64_t)s * PLANE_ROW;
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
        pslack[f] = at(4 + 5 * f + 4);
      }
    }

    // ---- region of the work item: bounding box of its samples and their extent along every face normal (a
    // 2(DIM+1)-plane polytope around them, tighter than the box for skewed simplices and than the simplex's
    // half-spaces for patches in its interior): a point within c of a sample has pn.(x - org) within c of that
    // sample's value
    float slo[DIM + 1], shi[DIM + 1];
    {
      float mn[DIM], mx[DIM], dmn[DIM + 1], dmx[DIM + 1];
#pragma unroll
      for (int k = 0; k < DIM; ++k) { mn[k] = __builtin_inff(); mx[k] = -__builtin_inff(); }
#pragma unroll
      for (int f = 0; f <= DIM; ++f) { dmn[f] = __builtin_inff(); dmx[f] = -__builtin_inff(); }
      for (int sub = 0; sub < n_sub; ++sub) {  // (SUPER: the samples of the item's last chunk stay in registers)
        q = q_first + sub;
        make_samples();
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            mn[k] = __builtin_fminf(mn[k], p[i][k]);
            mx[k] = __builtin_fmaxf(mx[k], p[i][k]);
          }
#pragma unroll
          for (int f = 0; f <= DIM; ++f) {
            float dd = -po[f];
#pragma unroll
            for (int k = 0; k < DIM; ++k) dd = __builtin_fmaf(pn[f][k], p[i][k] - org[k], dd);
            dmn[f] = __builtin_fminf(dmn[f], dd);
            dmx[f] = __builtin_fmaxf(dmx[f], dd);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        blo[k] = wave_min_f32(mn[k]);
        bhi[k] = wave_max_f32(mx[k]);
      }
#pragma unroll
      for (int f = 0; f <= DIM; ++f) {
        slo[f] = wave_min_f32(dmn[f]);
        shi[f] = wave_max_f32(dmx[f]);
      }
    }
    float ext = 0.f, vol = 1.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      ext = __builtin_fmaxf(ext, bhi[k] - blo[k]);
      vol *= (bhi[k] - blo[k]);
    }
    if (seeded) {
      // minima found so far (the SUPER launch's query at half this cell size); a sample is settled when its
      // minimum passed that launch's test (bit 31 marks it in the fused layout: masked off here)
      const float c_prev = 0.5f * c_seed;
      const float c_ok_prev = (0.999f * c_prev) * (0.999f * c_prev);
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        const uint32_t w = ARG(out_d2)[s * (int64_t)R + row[i]];
        best[i] = __uint_as_float(w & ~SETTLED_BIT);
        open[i] = open[i] && !(best[i] <= c_ok_prev);
      }
    }

    PHASE(1);
    // ---- gather: leaves of the box tree overlapping [qlo, qhi]; returns their number or -1 (overflow)
    float qlo[DIM], qhi[DIM];
    auto gather = [&]() -> int {
      constexpr int GB = 4;  // frontier nodes tested per step
      int* fa = s_front;
      int* fb = s_front + MAXFRONT;
      int na = 0, nb = 0, n_leaf = 0;
      bool over = false;
      // children of up to GB nodes `grp[u]` (valid for u < ng) at level lvl -> append hits to out_list
      auto test_children = [&](int lvl, const int (&grp)[GB], int ng, int* out_list, int& out_n, int cap) {
        const int lvl_count = (int)ARG(lv.count[lvl]), lvl_off = (int)ARG(lv.off[lvl]);  // (node indices fit 32 bits here)
#ifdef FLOODER_PHASE_TIMERS
        ++d_steps;
#endif
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
        if (top == 0) test_children(0, g0_, 1, s_leaf, n_leaf, MAXLEAF);
        else test_children(top, g0_, 1, fa, na, MAXFRONT);
      }
      for (int lvl = top; lvl >= 1 && !over; --lvl) {
        nb = 0;
        for (int f = 0; f < na && !over; f += GB) {
          int grp[GB];
          const int ng = na - f < GB ? na - f : GB;
#pragma unroll
          for (int u = 0; u < GB; ++u) grp[u] = wave_uniform(fa[f + u < na ? f + u : f]);
          if (lvl == 1) test_children(0, grp, ng, s_leaf, n_leaf, MAXLEAF);
          else test_children(lvl - 1, grp, ng, fb, nb, MAXFRONT);
        }
        int* t = fa; fa = fb; fb = t;
        na = nb;
      }
      return over ? -1 : n_leaf;
    };

    // ---- 1. density of the cloud inside the region's box -> first cell size
#pragma unroll
    for (int k = 0; k < DIM; ++k) { qlo[k] = blo[k]; qhi[k] = bhi[k]; }
    // local density from the grid where the grid is telling (cells under the centre and the corners of the box, all
    // loads in flight at once, their mean as the box's density): dense and uniform regions.  Where the cells hold only
    // a few points each - the tails of a Gaussian cloud, where the density changes by an order of magnitude across
    // a coarse cell - the estimate is off by factors either way (cfg 2: 30 % of the tiles flagged instead of 0.04 %),
    // and the chunk counts the points under its box through the tree as before.
    bool use_grid = false;
    float c = seeded ? c_seed : ext;
    int n0 = 0;  // points of the cloud inside the box
    const DensGrid dg = ARG(dg);
    if (dg.grid != nullptr && !seeded) {
      typedef DensCfg<DIM> DC;
      const typename RowVec<4>::type b0 = load_uniform_row<4>(dg.box), b1 = load_uniform_row<4>(dg.box + 8);
      float glo[DIM], gsc[DIM], cell_vol = 1.f;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const float e = b1[k] - b0[k];
        glo[k] = b0[k];
        gsc[k] = e > 0.f ? (float)DC::G / e : 0.f;
        cell_vol *= e > 0.f ? e / (float)DC::G : 1.f;
      }
      constexpr int NP = (1 << DIM) + 1;
      int cnt[NP];
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) {
        int fine = 0;
#pragma unroll
        for (int k = DIM - 1; k >= 0; --k) {
          const float xk = pi == NP - 1 ? 0.5f * (blo[k] + bhi[k]) : (((pi >> k) & 1) ? bhi[k] : blo[k]);
          int ck = (int)((xk - glo[k]) * gsc[k]);
          ck = ck < 0 ? 0 : (ck >= DC::G ? DC::G - 1 : ck);
          fine = fine * DC::G + ck;
        }
        cnt[pi] = dg.grid[fine];
      }
      int sum = 0, mn = cnt[0];
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) {
        sum += cnt[pi];
        mn = cnt[pi] < mn ? cnt[pi] : mn;
      }
      if (mn >= dg.min_count) {  // every probed cell is well filled
        const float dens = (float)sum / ((float)NP * cell_vol);
        const float h = DIM == 3 ? cbrtf(1.f / dens) : __builtin_sqrtf(1.f / dens);
        c = ARG(alpha) * h;
        const float est = dens * vol;
        n0 = est < 1.f ? 1 : (est > 1.0e9f ? 1000000000 : (int)est);
        use_grid = true;
      }
    }
    int n_leaves = (seeded || use_grid) ? 0 : gather();
    PHASE(2);
    bool give_up = n_leaves < 0;
    if (give_up) ++g_gather0;
    if (!give_up && !seeded && !use_grid) {
      // (uniform trip count: every lane takes part in every ballot, so n0 stays wave-uniform)
      const int n_cand0 = n_leaves * LEAF;
      for (int ib = 0; ib < n_cand0; ib += 64 * UNR) {
        float x[UNR][DP];
        bool in[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {  // issue all loads of the step first
          const int idx = ib + u * 64 + lane;
          in[u] = idx < n_cand0;
          const uint32_t row = in[u] ? (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF) : 0u;
          load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) in[u] = in[u] && (x[u][k] >= blo[k]) && (x[u][k] <= bhi[k]);
          n0 += __popcll(__ballot(in[u]));
        }
      }
      if (n0 > 0 && vol > 0.f) {
        const float h = DIM == 3 ? cbrtf(vol / (float)n0) : __builtin_sqrtf(vol / (float)n0);
        c = ARG(alpha) * h;
      }
    }
    if (!(c > 0.f) || !(c < 3.0e38f)) c = 1.f;
    if (seeded) n0 = 1 << 27;  // (unknown: let the dense limit decide about the exhaustive evaluation)
    PHASE(3);

    // ---- 2-4. stage, query, verify; double c while samples stay open
#ifdef FLOODER_PHASE_TIMERS
    unsigned long long d_info = 0, d_tb = 0, d_flush = 0, d_wait = 0;  // diagnostics of the last attempt
#endif
    // ---- results of the current chunk; tiles of 64 samples that are still open go to the exact tree sweep.
    // seed_mode (SUPER launch, chunk with open samples): the minima are parked in the (S, R) buffer for the deferred
    // per-chunk pass instead (every tile is written, nothing is flagged).
    auto finalize = [&](bool seed_mode) {
    uint32_t* const out_d2 = ARG(out_d2);
    if (fused) {
      const FaceAcc acc = ARG(acc);
      // fused face maxima: every settled sample raises the running maximum of each face it lies on (one integer
      // atomic per face present in the chunk: interior chunks touch one face); the (S, R) buffer is only
      // written for the tiles the finish has to look at (bit 31 = already settled)
      uint32_t mb[SPL], um = 0u;
      bool fl_on[SPL];
      uint32_t fl_key[SPL];
#pragma unroll
      for (int i = 0; i < SPL; ++i) { fl_on[i] = false; fl_key[i] = 0u; }
      // lane f: the slot of face f of this simplex and the running maximum it holds now - ONE chain of two loads for
      // all faces, in flight beside the membership words, where every face of the chunk used to wait for its own slot
      // look-up in turn (the delivery was a fifth of the sweep's wave time at cfg 2 / cfg 3: serial round trips) -
      // and a value that cannot raise the maximum is not sent (most deliveries of a shared face are such: every one of
      // them queued on the word's atomic unit)
      const int64_t my_slot = lane < acc.n_faces ? acc.slot_of(s, lane) : 0;
      const uint32_t my_fb = lane < acc.n_faces
                                 ? __hip_atomic_load(acc.face_bits + my_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : 0xffffffffu;
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        const bool settled = (q * CHUNK + i * 64 + lane < n_live) && !open[i];
        mb[i] = settled ? acc.memb[row[i]] : 0u;
        um |= mb[i];
      }
      um = wave_or_u32(um);
      while (um) {  // (wave-uniform)
        const int f = __builtin_ctz(um);
        um &= um - 1u;
        uint32_t v = 0u;
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          const uint32_t b = ((mb[i] >> f) & 1u) ? __float_as_uint(best[i]) : 0u;
          v = b > v ? b : v;
        }
        v = wave_max_u32(v);
        // (lane f sends its face's value: it holds the slot)
        if (lane == f && v > my_fb) atomicMax(&acc.face_bits[my_slot], v);
      }
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        if (seed_mode) {
          if (q * CHUNK + i * 64 + lane < n_live)
            out_d2[s * (int64_t)R + row[i]] = __float_as_uint(best[i]) | (open[i] ? 0u : SETTLED_BIT);
        } else if (__ballot(open[i]) != 0ull) {
          uint32_t tile_key = 0u;
          if (acc.top) {
            // probe: one greedy descent of the box tree for the tile's open samples (nearest child box at every
            // level, one leaf evaluated) gives each of them a finite upper bound; the tile with the largest one
            // becomes its simplex's "top" tile, which the finish settles first
            float tlo[DIM], thi[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              tlo[k] = wave_min_f32(open[i] ? p[i][k] : __builtin_inff());
              thi[k] = wave_max_f32(open[i] ? p[i][k] : -__builtin_inff());
            }
            int grp = 0;
            for (int lvl = top; lvl >= 0; --lvl) {
              const int idx = grp * FAN + lane;
              const bool ok = idx < (int)ARG(lv.count[lvl]);
              float lo[DP], hi[DP];
              const uint32_t nb_ = (uint32_t)((int)ARG(lv.off[lvl]) + (ok ? idx : 0)) * (uint32_t)(2 * DP * sizeof(float));
              load_row_at<DP>(nodes, nb_, lo);
              load_row_at<DP>(nodes, nb_ + (uint32_t)(DP * sizeof(float)), hi);
              float lb = 0.f;
#pragma unroll
              for (int k = 0; k < DIM; ++k) {
                const float gap = __builtin_fmaxf(__builtin_fmaxf(lo[k] - thi[k], tlo[k] - hi[k]), 0.f);
                lb = __builtin_fmaf(gap, gap, lb);
              }
              lb = ok ? lb : __builtin_inff();
              const float mn = wave_min_f32(lb);
              grp = grp * FAN + __builtin_ctzll(__ballot(lb == mn));
            }
            const float* cp = pts + (int64_t)grp * LEAF * DP;
            float bb = best[i];
#pragma unroll
            for (int h = 0; h < LEAF; h += 8) {
              typename RowVec<DP>::type cc[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) cc[u] = load_uniform_row<DP>(cp + (h + u) * DP);
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                float t0 = p[i][0] - cc[u][0];
                float d2 = t0 * t0;
                t0 = p[i][1] - cc[u][1];
                d2 = __builtin_fmaf(t0, t0, d2);
                if constexpr (DIM == 3) {
                  t0 = p[i][2] - cc[u][2];
                  d2 = __builtin_fmaf(t0, t0, d2);
                }
                bb = __builtin_fminf(bb, d2);
              }
            }
            best[i] = bb;
            const uint32_t key = wave_max_u32(open[i] ? __float_as_uint(bb) : 0u);
            tile_key = key;
            if (lane == 0 && key != 0u) {
              const unsigned long long old =
                  atomicMax(&acc.top[s], ((unsigned long long)key << 32) | (unsigned long long)(uint32_t)(s * tiles64 + q * SPL + i));
              if (old == 0ull) acc.top_list[atomicAdd(acc.top_count, 1)] = (int)s;
            }
          }
          if (q * CHUNK + i * 64 + lane < n_live)
            out_d2[s * (int64_t)R + row[i]] = __float_as_uint(best[i]) | (open[i] ? 0u : SETTLED_BIT);
          fl_on[i] = true;      // (appended below: one reservation for the chunk's tiles, not one atomic each)
          fl_key[i] = tile_key;
          ++n_flagged;
        }
      }
      // the flag list's one counter word takes every append of the launch: a returning atomic per flagged tile
      // (138 k at cfg 3) queues on it at ~11 ns each - reserve the chunk's entries together, and count equal
      // histogram bins once
      int nf = 0;
#pragma unroll
      for (int i = 0; i < SPL; ++i) nf += fl_on[i] ? 1 : 0;
      if (nf > 0) {  // (wave-uniform)
        int base = 0;
        if (lane == 0) base = atomicAdd(ARG(flag_count), nf);
        base = wave_uniform(base);
        int32_t* const flag_list = ARG(flag_list);
        int k = 0;
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          if (fl_on[i]) {
            if (lane == 0) {
              flag_list[base + k] = (int)(s * tiles64 + q * SPL + i);
              if (acc.flag_key) {
                acc.flag_key[base + k] = fl_key[i];
                bool first = true;
                int same = 0;
#pragma unroll
                for (int j = 0; j < SPL; ++j) {
                  if (fl_on[j] && (fl_key[j] >> 19) == (fl_key[i] >> 19)) {
                    if (j < i) first = false;
                    ++same;
                  }
                }
                if (first) atomicAdd(&acc.flag_hist[fl_key[i] >> 19], same);
              }
            }
            ++k;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        if (q * CHUNK + i * 64 + lane < n_live) out_d2[s * (int64_t)R + row[i]] = __float_as_uint(best[i]);
        if (!seed_mode && __ballot(open[i]) != 0ull) {
          if (lane == 0) {
            const int pos = atomicAdd(ARG(flag_count), 1);
            ARG(flag_list)[pos] = (int)(s * tiles64 + q * SPL + i);
          }
          ++n_flagged;
        }
      }
    }
    };
    // SUPER: a chunk is done after its one query; with open samples left it is parked and deferred (next cell size 2c)
    // Is a second try at twice the cell size worth it?  (Per-chunk launch only: a run of four defers its open chunks
    // as before.)  An open sample whose minimum over the staged points lies within 2c is certain to settle then; one
    // that has seen nothing nearer (or nothing at all: far field) most likely stays open and costs a second gather,
    // classification and query before the finish gets it anyway; and where the first try already kept hundreds of
    // points the second keeps thousands, while the finish's search for a sample in such a dense place is short.
    // Retry when the try kept at most retry_keep points and at least retry_pct percent of the open samples are of the
    // first kind (0: whatever they are).
    int kept_last = 0;  // points the last attempt kept per chunk (the next one keeps several times as many)
    auto retry_pays = [&](float c_now) -> bool {
      if (kept_last > ARG(retry_keep)) return false;
      const int retry_pct = ARG(retry_pct);
      if (retry_pct <= 0) return true;
      const float lim = (1.998f * c_now) * (1.998f * c_now);
      int n_open = 0, n_near = 0;
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        n_open += __popcll(__ballot(open[i]));
        n_near += __popcll(__ballot(open[i] && best[i] <= lim));
      }
      return n_near * 100 >= n_open * retry_pct;
    };
    auto finalize_sub = [&](bool any_open_lane, float c_now) {
      const bool any = __ballot(any_open_lane) != 0ull;
      finalize(any);
      if (any) defer(q, 1, 2.f * c_now);
    };
    bool sc_done = false;
    bool to_tiles = false;  // per-chunk launch: the chunk was handed to the tile launch
    // SUPER: the stage is built once for the whole run of chunks; anything that does not work out is deferred
    auto defer_all_fresh = [&]() {
      for (int sub = 0; sub < n_sub; ++sub) defer(q_first + sub, 0, 0.f);
    };
    if constexpr (SUPER) {
      // will the neighbourhood fit the stage?  points in the box x growth of the box by c on every side x the share
      // the slab filter keeps (about half): a run that is predicted not to fit is not gathered at all
      float grow = 0.5f;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const float e = bhi[k] - blo[k];
        grow *= e > 0.f ? __builtin_fminf((e + 2.f * c) / e, 8.f) : 8.f;
      }
      // (an empty box gives no density: the cell size would fall back to the extent of the whole run, far more than
      // its chunks would try on their own - leave those to the per-chunk pass)
      if (give_up || n0 == 0 || (float)n0 * grow > (float)ARG(dl.n0_limit)) { defer_all_fresh(); continue; }
    }
    int max_tries = SUPER ? 1 : ARG(max_tries);
    if constexpr (!SUPER) {
      const int spct = ARG(dg.surface_pct);
      if (spct > 0 && dg.grid != nullptr) {   // a cloud on a surface: the doubled cell size does not pay (flood_common.hpp)
        const int32_t* kind = dg.grid + DensCfg<DIM>::NF;
        const int inner = __builtin_amdgcn_readfirstlane(kind[2]), all = __builtin_amdgcn_readfirstlane(kind[3]);
        if (all > 0 && (long long)inner * 100 < (long long)all * spct) max_tries = 1;
      }
    }
    for (int attempt = seeded ? 1 : 0; attempt < max_tries && !give_up; ++attempt) {
      c = __builtin_fmaxf(c, ext / (float)(G - 3));
      const float inv_c = 1.f / c;
      int nc[DIM];
      float g0[DIM];
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
      float slab_lo[DIM + 1], slab_hi[DIM + 1];
#pragma unroll
      for (int f = 0; f <= DIM; ++f) {
        const float tol = c * 1.001f + (pslack[f] + 1e-6f) * (sext + c);
        slab_lo[f] = slo[f] - tol;
        slab_hi[f] = shi[f] + tol;
      }
      auto cell_of = [&](const float (&x)[DP]) {
        int id = 0;
#pragma unroll
        for (int k = DIM - 1; k >= 0; --k) {
          int ck = (int)((x[k] - g0[k]) * inv_c);
          ck = ck < 0 ? 0 : (ck >= nc[k] ? nc[k] - 1 : ck);
          id = __mul24(id, nc[k]) + ck;
        }
        return id;
      };
      auto keep_point = [&](const float (&x)[DP]) {
        bool in = true;
#pragma unroll
        for (int k = 0; k < DIM; ++k) in = in && (x[k] >= qlo[k]) && (x[k] <= qhi[k]);
        float xr[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) xr[k] = x[k] - org[k];
#pragma unroll
        for (int f = 0; f <= DIM; ++f) {
          float dd = -po[f];
#pragma unroll
          for (int k = 0; k < DIM; ++k) dd = __builtin_fmaf(pn[f][k], xr[k], dd);
          in = in && (dd <= slab_hi[f]) && (dd >= slab_lo[f]);
        }
        return in;
      };

      n_leaves = gather();
      PHASE(4);
      if (n_leaves < 0) { give_up = true; ++g_gather; break; }
      for (int i = lane; i < (ncells + 3) / 2; i += 64) s_cell32[i] = 0u;
      wave_lds_sync();
      // 2a. classify every candidate ONCE (keep? which cell?), count per cell, remember the kept ones
      int n_keep = 0;  // wave-uniform: the loop has a uniform trip count
      const int n_cand = n_leaves * LEAF;
      // Single-pass exhaustive evaluation (per-chunk launch): the moment the kept set outgrows the stage the pass
      // stops recording and starts EVALUATING - the kept points of the current and all following candidate rows are
      // compacted into the first three quarters of the stage (the last quarter still holds the kept list recorded so
      // far) and every lane runs its samples against each staged batch; the recorded ones are fetched again at the end
      // (<= 480 rows).  The candidates of a dense chunk - 7 to 12 thousand rows for 2 to 3 thousand kept points -
      // are streamed and classified once instead of twice.
      bool one_pass = false;
      int n_st = 0, n_rec = 0, ib_resume = 0;
      bool decided = false;
      constexpr int CAPE = KEEP0 - 4;  // stage entries below the kept-list alias (and four of padding)
      static_assert(CAPE >= 64 * UNR + 32, "a batch of candidate rows fits the part of the stage below the kept list");
      auto flush_stage = [&]() {
        s_pts.pad(n_st, lane);
        wave_lds_sync();
        for (int j = 0; j < n_st; j += 4) {
          float4 x[4];
s_pts.get4(j, x);
#pragma unroll
          for (int i = 0; i < SPL; ++i) {
            float bb = best[i];
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
              bb = __builtin_fminf(bb, d2);
            }
            best[i] = bb;
          }
        }
        if (has_stats) n_pairs += (unsigned long long)n_st * SPL;
        n_st = 0;
        wave_lds_sync();
      };
      auto stage_row = [&](const float (&x)[DP], bool k) {
        const unsigned long long m = __ballot(k);
        if (k) {
          s_pts.put(n_st + lane_rank(m), x[0], x[1], DIM > 2 ? x[DIM > 2 ? 2 : 0] : 0.f);
        }
        n_st += __popcll(m);
        return __popcll(m);
      };
      for (int ib = 0; ib < n_cand; ib += 64 * UNR) {
        float x[UNR][DP];
        bool keep[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {  // issue all loads of the step first
          const int idx = ib + u * 64 + lane;
          keep[u] = idx < n_cand;
          const uint32_t row = keep[u] ? (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF) : 0u;
          load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x[u]);
        }
        const int n_before = n_keep;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int idx = ib + u * 64 + lane;
          keep[u] = keep[u] && keep_point(x[u]);
          const int cid = cell_of(x[u]);
          const unsigned long long m = __ballot(keep[u]);
          if (keep[u]) {
            cell_add(cid + 1);
            const int slot = n_keep + lane_rank(m);
            if (slot < CAPW) s_keep[slot] = (idx << 10) | cid;
            if (slot < BRUTE_CAP) {  // (below the kept-list alias) small sets are evaluated straight from here
              s_pts.put(slot, x[u][0], x[u][1], DIM > 2 ? x[u][DIM > 2 ? 2 : 0] : 0.f);
            }
          }
          n_keep += __popcll(m);
        }
        if constexpr (!TILES) {
          if (n_keep > CAPW && !decided) {  // the kept set has just outgrown the stage
            decided = true;
            if constexpr (SUPER) {
              break;  // (a run that does not fit the stage is deferred: no need to count on)
            } else if (ARG(dg.one_pass) && !ARG(dl.tile_list) && attempt < ARG(exh_tries)) {
              // will the kept set be given up anyway (the cap below)?  Extrapolate from the share of the candidates
              // seen so far; a chunk that is likely to be dropped only counts on, as before
              const int seen = ib + 64 * UNR < n_cand ? ib + 64 * UNR : n_cand;
              const float est = (float)n_keep * (float)n_cand / (float)seen;
              const float cap = (float)n0 * 8.f >= est ? (float)ARG(exh_dense) : (float)ARG(exh_sparse);
              if (est * 100.f <= (float)ARG(dg.one_pass) * cap) {
                // switch: the first CAPW kept points are on record (last quarter of the stage), some of them from
                // this batch; the batch is streamed again below - a point evaluated twice does no harm to a minimum
                one_pass = true;
                n_rec = CAPW;
                ib_resume = ib;
                n_keep = n_before;
                wave_lds_sync();
                break;
              }
            }
          }
        }
      }
      if constexpr (!SUPER && !TILES) {
        if (one_pass) {  // the rest of the candidates: classified, staged and evaluated as they come
          for (int ib = ib_resume; ib < n_cand; ib += 64 * UNR) {
            float x[UNR][DP];
            bool keep[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
              const int idx = ib + u * 64 + lane;
              keep[u] = idx < n_cand;
              const uint32_t row = keep[u] ? (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF) : 0u;
              load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x[u]);
            }
            if (n_st + 64 * UNR > CAPE) flush_stage();
#pragma unroll
            for (int u = 0; u < UNR; ++u) n_keep += stage_row(x[u], keep[u] && keep_point(x[u]));
            if (n_keep > (n0 * 8 >= n_keep ? ARG(exh_dense) : ARG(exh_sparse))) break;  // (hopeless after all: decided below)
          }
        }
      }
      PHASE(5);
#ifdef FLOODER_PHASE_TIMERS
      d_info = (unsigned long long)n_cand | ((unsigned long long)n_keep << 20) | ((unsigned long long)(attempt + 1) << 40) |
               ((unsigned long long)(n_keep > CAPW) << 44);
      d_tb = 0;
#endif
      kept_last = n_keep;
      const float c_ok = (0.999f * c) * (0.999f * c);
      // Exhaustive evaluation pays when the kept points really are the samples' neighbours (a chunk box full
      // of points); when they form a distant shell around an empty chunk the tree sweep's culling is cheaper.
      if constexpr (SUPER) {
        if (n_keep > CAPW) break;  // the neighbourhood of the whole run does not fit the stage: chunk by chunk
      }
      if constexpr (!SUPER && !TILES) {
        if (n_keep > CAPW && ARG(dl.tile_list) && (int64_t)g >= (int64_t)n_items - ARG(dl.tail_items)) {
          // dense chunk: its open tiles of 64 samples go to the tile launch (a quarter of the region each: most fit the
          // stage there) instead of an exhaustive evaluation of every kept point against all 256 samples.  A chunk
          // that already holds minima (seeded, or a second attempt) parks them first, as a run of four does.
          const bool fresh = attempt == 0 && !seeded;
          if (!fresh) finalize(true);
#pragma unroll
          for (int i = 0; i < SPL; ++i) {
            if (q * CHUNK + i * 64 < n_live && __ballot(open[i]) != 0ull) {  // (wave-uniform)
              if (lane == 0) {
                const int pos = atomicAdd(ARG(dl.tile_count), 1);
                ARG(dl.tile_list)[pos] = (int)(((s * tiles64 + q * SPL + i) << 1) | (fresh ? 0 : 1));
                ARG(dl.tile_c)[pos] = fresh ? 0.f : c;
              }
            }
          }
          to_tiles = true;
          ++g_brute;
          break;
        }
      }
      if (n_keep > (n0 * 8 >= n_keep ? ARG(exh_dense) : ARG(exh_sparse))) { give_up = true; ++g_cap; break; }
      if (n_keep <= ARG(brute_max)) {
        // ---- few kept points: every sample against every one of them, straight from the compacted list the
        // classification pass left in the stage (broadcast LDS reads, no cell table, no second pass over the
        // candidates).  Cheaper than the cell query's dependent LDS chains while the list is short.
        s_pts.pad(n_keep, lane);
        wave_lds_sync();
        auto brute_eval = [&]() {
          for (int j = 0; j < n_keep; j += 4) {
            float4 x[4];
  s_pts.get4(j, x);
  #pragma unroll
            for (int i = 0; i < SPL; ++i) {
              float bb = best[i];
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
                bb = __builtin_fminf(bb, d2);
              }
              best[i] = bb;
            }
          }
          if (has_stats) n_pairs += (unsigned long long)n_keep * SPL;
        };
        if constexpr (SUPER) {
          for (int sub = n_sub - 1; sub >= 0; --sub) {  // (the last chunk's samples are still in registers)
            q = q_first + sub;
            if (sub != n_sub - 1) make_samples();
            PHASE(1);
            brute_eval();
            PHASE(9);
            bool any_open = false;
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
              open[i] = open[i] && !(best[i] <= c_ok);
              any_open = any_open || open[i];
            }
            finalize_sub(any_open, c);
          }
          n_staged += (unsigned long long)n_keep;
          sc_done = true;
          wave_lds_sync();
          break;
        }
        brute_eval();
        n_staged += (unsigned long long)n_keep;
        bool any_open = false;
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          open[i] = open[i] && !(best[i] <= c_ok);
          any_open = any_open || open[i];
        }
        if (attempt > 0) ++n_retries;
        wave_lds_sync();
        PHASE(8);
        if (__ballot(any_open) == 0ull) break;
        if (attempt == max_tries - 1) ++g_tries;
        if (!retry_pays(c)) break;
        c *= 2.f;
        continue;
      }
      if (n_keep > CAPW && attempt >= ARG(exh_tries)) { give_up = true; ++g_cap; break; }  // leave it to the finish
      if (n_keep > CAPW) {
        // ---- too many points for the LDS cell stage: evaluate them exhaustively instead.  The candidates
        // are streamed once more, the kept ones are compacted into LDS (<= CAPW at a time) and every lane
        // runs its open samples against the staged batch with broadcast LDS reads.
        wave_lds_sync();
        ++g_brute;
        if (one_pass) {
          // ... the rows that were recorded before the switch (their slots are still in the last quarter of the stage)
          for (int h0 = 0; h0 < n_rec; h0 += 64 * UNR) {
            int ent[UNR];
            float x[UNR][DP];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
              const int k = h0 + u * 64 + lane;
              ent[u] = k < n_rec ? s_keep[k] : -1;
              const int idx = ent[u] < 0 ? 0 : ent[u] >> 10;
              const uint32_t row = (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF);
              load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x[u]);
            }
            if (n_st + 64 * UNR > CAPE) flush_stage();
#pragma unroll
            for (int u = 0; u < UNR; ++u) stage_row(x[u], ent[u] >= 0);
          }
          if (n_st > 0) flush_stage();
        } else {
        int n_st = 0;
        auto flush = [&]() {
#ifdef FLOODER_PHASE_TIMERS
          const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
#endif
          s_pts.pad(n_st, lane);
          wave_lds_sync();
          for (int j = 0; j < n_st; j += 4) {
            float4 x[4];
s_pts.get4(j, x);
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
              float bb = best[i];
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
                bb = __builtin_fminf(bb, d2);
              }
              best[i] = bb;
            }
          }
          if (has_stats) n_pairs += (unsigned long long)n_st * SPL;
          n_st = 0;
          wave_lds_sync();
#ifdef FLOODER_PHASE_TIMERS
          d_flush += __builtin_amdgcn_s_memtime() - tf0;
#endif
        };
        for (int ib = 0; ib < n_cand; ib += 64 * UNR) {
          float x[UNR][DP];
          bool keep[UNR];
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int idx = ib + u * 64 + lane;
            keep[u] = idx < n_cand;
            const uint32_t row = keep[u] ? (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF) : 0u;
            load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x[u]);
          }
#ifdef FLOODER_PHASE_TIMERS
          {
            const unsigned long long ta = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            d_wait += __builtin_amdgcn_s_memtime() - ta;
          }
#endif
          if (n_st + 64 * UNR > CAPW) flush();  // (one call site: room for a whole step is checked up front)
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            keep[u] = keep[u] && keep_point(x[u]);
            const unsigned long long m = __ballot(keep[u]);
            if (keep[u]) {
              s_pts.put(n_st + lane_rank(m), x[u][0], x[u][1], DIM > 2 ? x[u][DIM > 2 ? 2 : 0] : 0.f);
            }
            n_st += __popcll(m);
          }
        }
        if (n_st > 0) flush();
        }
        bool any_open = false;
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          open[i] = open[i] && !(best[i] <= c_ok);
          any_open = any_open || open[i];
        }
        if (attempt > 0) ++n_retries;
        PHASE(9);
        if (__ballot(any_open) == 0ull) break;
        if (attempt == max_tries - 1) ++g_tries;
        if (!retry_pays(c)) break;
        c *= 2.f;
        continue;
      }
      wave_lds_sync();
      // 2b. exclusive prefix, 64 cells per step: s_cell[i+1] = start of cell i
      int total = 0;
      for (int cb = 0; cb < ncells; cb += 64) {
        const int ci = cb + lane;
        const int cnt = ci < ncells ? s_cell[ci + 1] : 0;
        const int incl = wave_incl_scan(cnt, lane);
        if (ci < ncells) s_cell_w[ci + 1] = (uint16_t)(total + incl - cnt);
        total += __builtin_amdgcn_readlane(incl, 63);
      }
      wave_lds_sync();
      if (total > CAPW) { give_up = true; break; }
      PHASE(6);
      // 2c. scatter the kept candidates; afterwards s_cell[i] = begin and s_cell[i+1] = end of cell i
      {
        constexpr int KPL = (CAPW + 63) / 64;  // kept entries per lane (n_keep <= CAPW here)
        int ent[KPL];
#pragma unroll
        for (int u = 0; u < KPL; ++u) {
          const int k = u * 64 + lane;
          ent[u] = k < n_keep ? s_keep[k] : -1;
        }
        wave_lds_sync();  // the kept list (inside the point stage) is in registers now: the stage may be written
#pragma unroll
        for (int h = 0; h < KPL; h += UNR) {
          if (h * 64 >= n_keep) break;  // (wave-uniform)
          float x[UNR][DP];
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int idx = ent[h + u] < 0 ? 0 : ent[h + u] >> 10;
            const uint32_t row = (uint32_t)s_leaf[idx / LEAF] * LEAF + (uint32_t)(idx % LEAF);
            load_row_at<DP>(pts, row * (uint32_t)(DP * sizeof(float)), x[u]);
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            if (ent[h + u] >= 0) {
              const int pos = cell_add((ent[h + u] & 1023) + 1);
              s_pts.put(pos, x[u][0], x[u][1], DIM > 2 ? x[u][DIM > 2 ? 2 : 0] : 0.f);
            }
          }
        }
      }
      s_pts.pad(total, lane);
      wave_lds_sync();
      n_staged += (unsigned long long)total;
      if (attempt > 0) ++n_retries;

      PHASE(7);
      // 3. query the open samples of the current chunk
      // Only face MAXIMA are wanted: an INTERIOR sample (it lies on the full simplex alone: memb == 1) whose running
      // minimum - the distance to a real point - has fallen to the running maximum of the simplex can never raise it.
      // It is dropped on the spot: its minimum becomes 0, which skips its remaining rows of cells (no slab is nearer
      // than 0), counts as verified, and delivers nothing.  In a dense region the first row of cells (own cell and its
      // two x-neighbours, ~7 points) usually suffices where the exact query looks at all 27 cells (~66 points).
      uint32_t thr_top = 0u;
      if (fused && ARG(dl.drop)) {
        const FaceAcc acc = ARG(acc);
        thr_top = __hip_atomic_load(acc.face_bits + acc.slot_of(s, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const uint32_t* const memb = thr_top != 0u ? ARG(acc.memb) : nullptr;
      // One sample's rows of cells ROW_FROM .. ROW_TO - 1 (of the 3^(dim-1) rows of 3 cells along x around its own cell,
      // own row first): a row is skipped when even its slab is no closer than the running minimum.  Returns the new
      // minimum; `more`: a row at or beyond ROW_TO could still lower it.
      constexpr int NROW = DIM == 3 ? 9 : 3;
      auto sample_rows = [&](const float (&ps)[DIM], float b, uint32_t thr_i, auto row_from, auto row_to, bool& more) -> float {
        constexpr int ROW_FROM = decltype(row_from)::value, ROW_TO = decltype(row_to)::value;
        int ck[DIM];
        float gap2[DIM][3];  // squared distance from the sample to the cell slab at offset -1 / 0 / +1
  #pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float tf = (ps[k] - g0[k]) * inv_c;
          const int t = (int)tf;
          ck[k] = t < 1 ? 1 : (t > nc[k] - 2 ? nc[k] - 2 : t);
          const float f = tf - (float)ck[k];  // position inside the (clamped) cell, in cells
          const float lo_gap = __builtin_fmaxf(f, 0.f) * c * 0.999f;
          const float hi_gap = __builtin_fmaxf(1.f - f, 0.f) * c * 0.999f;
          gap2[k][0] = lo_gap * lo_gap;
          gap2[k][1] = 0.f;
          gap2[k][2] = hi_gap * hi_gap;
        }
        constexpr int ORD3[9][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
        constexpr int ORD2[3] = {0, -1, 1};
        // (row strides are wave-uniform; 24-bit multiplies run at full rate, 32-bit ones at a quarter)
        const int stride_y = nc[0], stride_z = DIM == 3 ? nc[0] * nc[1] : 0;
        int base0 = __mul24(ck[1], stride_y) + ck[0] - 1;
        if constexpr (DIM == 3) base0 += __mul24(ck[DIM - 1], stride_z);
#ifdef FLOODER_QUERY_DIAG
        int rows_seen = 0;
#endif
  #pragma unroll
        for (int rw = ROW_FROM; rw < ROW_TO; ++rw) {
          int base;
          float lb;
          if constexpr (DIM == 3) {
            const int dy = ORD3[rw][0], dz = ORD3[rw][1];
            base = base0 + dy * stride_y + dz * stride_z;
            lb = gap2[1][dy + 1] + gap2[DIM - 1][dz + 1];
          } else {
            const int dy = ORD2[rw];
            base = base0 + dy * stride_y;
            lb = gap2[1][dy + 1];
          }
          if (!(lb < b)) continue;
#ifdef FLOODER_QUERY_DIAG
          ++rows_seen;
#endif
          // the row's outer cells are dropped too when their slab is no closer than the running minimum
          const int first = (lb + gap2[0][0] < b) ? 0 : 1;
          const int last = (lb + gap2[0][2] < b) ? 3 : 2;
          const int bg = s_cell[base + first];
          const int en = s_cell[base + last];
          if (has_stats) n_pairs += (unsigned long long)(en - bg);
          // whole blocks of four staged points; entries before `bg` and past `en` are real points of other cells or
          // the +inf pads behind the list - a minimum over more real points is still a valid upper bound, and exact
          // once verified
          for (int j = bg & ~3; j < en; j += 4) {
            float4 x[4];
            s_pts.get4(j, x);
  #pragma unroll
            for (int u = 0; u < 4; ++u) {
              float t0 = ps[0] - x[u].x;
              float d2 = t0 * t0;
              t0 = ps[1] - x[u].y;
              d2 = __builtin_fmaf(t0, t0, d2);
              if constexpr (DIM == 3) {
                t0 = ps[2] - x[u].z;
                d2 = __builtin_fmaf(t0, t0, d2);
              }
              b = __builtin_fminf(b, d2);
            }
          }
          if (__float_as_uint(b) <= thr_i) b = 0.f;  // (dropped: cannot raise the simplex's maximum)
        }
        more = false;
  #pragma unroll
        for (int rw = ROW_TO; rw < NROW; ++rw) {
          float lb;
          if constexpr (DIM == 3) lb = gap2[1][ORD3[rw][0] + 1] + gap2[DIM - 1][ORD3[rw][1] + 1];
          else lb = gap2[1][ORD2[rw] + 1];
          more = more || (lb < b);
        }
#ifdef FLOODER_QUERY_DIAG
        {  // diagnostic build: rows of cells visited per call, [10]: samples dropped against the maximum
          unsigned long long* const dst = ARG(stats);
          if (dst) { atomicAdd(&dst[100 + rows_seen], 1ull); if (b == 0.f) atomicAdd(&dst[110], 1ull); }
        }
#endif
        return b;
      };
      typedef std::integral_constant<int, 0> Row0;
      typedef std::integral_constant<int, 1> Row1;
      typedef std::integral_constant<int, NROW> RowN;
#if FLOODER_CELL_COMPACT
      // A queried sample visits 2.1 of its 9 rows on average - 57 - 64 % only their own row (tools/query_rows.py) - but
      // which ones differs from lane to lane, and a wave that takes its lanes' samples one after the other runs every
      // one of its four phases to the slowest lane: 6 - 7 rows executed at a quarter of the lanes.  So: (A) every open
      // sample visits its OWN row; (B) the samples that another row could still improve - a third - are compacted
      // ACROSS lanes through LDS (the leaf list's storage, idle during the query: coordinates, minimum, threshold) and
      // worked off 64 at a time, whatever lane they came from; their minima go back the same way.
      // MEASURED (round 5, bit-identical on all 171 GPU tests): the rows executed per chunk fall by a third as
      // predicted, the time does not - cfg 5 sweep 6.41 vs 6.37 ms, cfg 3 2.22 vs 2.16, cfg 2 0.907 vs 0.902: a row
      // executed for three lanes ends after the one or two blocks of points those lanes have, a row executed for 64
      // lanes runs to the longest of 64 lists, and the second phase costs 29 spilled VGPRs at the 128-register cap.
      // Off by default.
      auto query_cells = [&]() -> bool {
        uint32_t flags = 0u;  // bit i: sample i goes on to phase B; bit 8 + i: it is an interior sample (threshold applies)
  #pragma unroll
        for (int i = 0; i < SPL; ++i) {
          if (open[i]) {
            const bool interior = thr_top != 0u && memb[row[i]] == 1u;
            bool more;
            best[i] = sample_rows(p[i], best[i], interior ? thr_top : 0u, Row0{}, Row1{}, more);
            flags |= (more ? 1u : 0u) << i | (interior ? 256u : 0u) << i;
            if (!more) open[i] = !(best[i] <= c_ok);
          }
        }
        float4* const s_x = reinterpret_cast<float4*>(s_leaf);            // 64 entries: x, y, z, minimum
        uint32_t* const s_t = reinterpret_cast<uint32_t*>(s_leaf) + 256;  // 64 thresholds
        static_assert(MAXLEAF * sizeof(int) >= 64 * 16 + 64 * 4, "the compacted samples fit the leaf list's storage");
        unsigned long long m[SPL];
        int n_surv = 0;
  #pragma unroll
        for (int i = 0; i < SPL; ++i) {
          m[i] = __ballot(((flags >> i) & 1u) != 0u);
          n_surv += __popcll(m[i]);
        }
        for (int b0 = 0; b0 < n_surv; b0 += 64) {  // (wave-uniform)
          int off = 0;
  #pragma unroll
          for (int i = 0; i < SPL; ++i) {
            const int pos = off + lane_rank(m[i]) - b0;
            if (((flags >> i) & 1u) && pos >= 0 && pos < 64) {
              s_x[pos] = make_float4(p[i][0], p[i][1], DIM > 2 ? p[i][DIM > 2 ? 2 : 0] : 0.f, best[i]);
              s_t[pos] = ((flags >> (8 + i)) & 1u) ? thr_top : 0u;
            }
            off += __popcll(m[i]);
          }
          wave_lds_sync();
          if (b0 + lane < n_surv) {
            const float4 e = s_x[lane];
            float ps[DIM];
            ps[0] = e.x;
            ps[1] = e.y;
            if constexpr (DIM == 3) ps[2] = e.z;
            bool more;
            const float bq = sample_rows(ps, e.w, s_t[lane], Row1{}, RowN{}, more);
            s_x[lane].w = bq;
          }
          wave_lds_sync();
          off = 0;
  #pragma unroll
          for (int i = 0; i < SPL; ++i) {
            const int pos = off + lane_rank(m[i]) - b0;
            if (((flags >> i) & 1u) && pos >= 0 && pos < 64) {
              best[i] = s_x[pos].w;
              open[i] = !(best[i] <= c_ok);
            }
            off += __popcll(m[i]);
          }
          wave_lds_sync();
        }
        bool any_open = false;
  #pragma unroll
        for (int i = 0; i < SPL; ++i) any_open = any_open || open[i];
        return any_open;
      };
#else
      auto query_cells = [&]() -> bool {
        bool any_open = false;
  #pragma unroll
        for (int i = 0; i < SPL; ++i) {
          if (open[i]) {
            const uint32_t thr_i = (thr_top != 0u && memb[row[i]] == 1u) ? thr_top : 0u;
            bool more;
            best[i] = sample_rows(p[i], best[i], thr_i, Row0{}, RowN{}, more);
            open[i] = !(best[i] <= c_ok);
          }
          any_open = any_open || open[i];
        }
        return any_open;
      };
#endif
      if constexpr (SUPER) {
        for (int sub = n_sub - 1; sub >= 0; --sub) {  // (the last chunk's samples are still in registers)
          q = q_first + sub;
          if (sub != n_sub - 1) make_samples();
          PHASE(1);
          const bool ao = query_cells();
          PHASE(8);
          finalize_sub(ao, c);
          PHASE(10);
        }
        sc_done = true;
        wave_lds_sync();
        break;
      }
      const bool any_open = query_cells();
      PHASE(8);
      if (__ballot(any_open) == 0ull) break;
      if (attempt == max_tries - 1) ++g_tries;
      if (!retry_pays(c)) break;
      c *= 2.f;
    }

    if constexpr (SUPER) {
      if (!sc_done) defer_all_fresh();
      PHASE(10);
      continue;
    }
    if (to_tiles) continue;
    finalize(false);
    PHASE(10);
#ifdef FLOODER_PHASE_TIMERS
    unsigned long long* const stats = ARG(stats);
    if (stats && lane == 0) {  // diagnostic build only: sixteen values per chunk
      stats[64 + 16 * g] = t_chunk0;
      stats[64 + 16 * g + 1] = __builtin_amdgcn_s_memrealtime();
      stats[64 + 16 * g + 2] = d_info;
      stats[64 + 16 * g + 3] = d_wait | (d_steps << 40);   // cycles waiting for the streamed rows in the exhaustive mode
      stats[64 + 16 * g + 4] = d_flush;  // cycles inside flush() of the exhaustive mode
      for (int i = 0; i < 11; ++i) stats[64 + 16 * g + 5 + i] = t_phase[i] - t_phase0[i];  // core-clock cycles per phase
    }
#endif
  }
  unsigned long long* const stats = ARG(stats);
#ifdef FLOODER_WAVE_END
  // diagnostic build: when did every wave start and finish?  (plain per-wave stores instead of the shared counters,
  // whose same-address atomics would themselves stretch the end of the kernel)
  if (stats) {
    if (lane == 0) {
      const int slot = (SUPER ? 0 : 4096) + (int)(blockIdx.x * 4 + wv);  // (runs launch | chunk launch: <= 4096 waves each)
      if (slot < 8192) {
        stats[64 + 2 * slot] = t_wave0;
        stats[65 + 2 * slot] = __builtin_amdgcn_s_memrealtime();
      }
    }
    return;
  }
#endif
  if (stats) {
#ifdef FLOODER_PHASE_TIMERS
    if (lane == 0)
      for (int i = 0; i < 12; ++i) atomicAdd(&stats[16 + i], t_phase[i]);
#endif
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_pairs += __shfl_xor(n_pairs, o);
    if (lane == 0) {
      atomicAdd(&stats[0], n_pairs);
      atomicAdd(&stats[1], n_staged);
      atomicAdd(&stats[2], n_flagged);
      atomicAdd(&stats[3], n_retries);
      atomicAdd(&stats[4], g_gather0);
      atomicAdd(&stats[5], g_gather);
      atomicAdd(&stats[6], g_cap);
      atomicAdd(&stats[7], g_tries);
      atomicAdd(&stats[8], g_brute);
    }
  }
}

template <int DIM>
struct CellOp {
  static int run(const float* pts, const float* nodes, const Levels& lv, const float* verts, float* plane_tab,
                 const float* weights, int k1, int R, int64_t ns, float alpha, int32_t* queue,
                 uint32_t* out, int32_t* flag_list, int32_t* flag_count, unsigned long long* stats,
                 FaceAcc acc, DeferList dl, int32_t* queue2, int32_t* queue3, DensGrid dg, hipStream_t st) {
    if constexpr (DIM == 2 || DIM == 3) {
      if (!g_cell_density_grid) dg = DensGrid{};
      dg.min_count = g_cell_density_grid;
      dg.one_pass = g_cell_one_pass;
      dg.surface_pct = g_cell_surface_pct;
      // persistent blocks of 4 independent waves: 3 per CU fit, but a short queue is swept faster by fewer
      // waves (its longest chunks then share their SIMD with fewer others): about 48 chunks per block,
      // measured on 1/4 and 1/8 shares of cfg 2
      if (!planes_are_done(verts, plane_tab, ns, st))
        hipLaunchKernelGGL((simplex_planes_kernel<DIM>), dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, verts, k1, ns,
                           plane_tab);
      constexpr int CHUNK = 64 * SPL_CHUNK;
      const int64_t n_chunks = ns * ((R + CHUNK - 1) / CHUNK);
      int64_t want = n_chunks / (g_cell_chunks_per_block > 0 ? g_cell_chunks_per_block : 48);
      want = want < g_cell_min_grid ? g_cell_min_grid : want;
      const int grid = (int)(want < g_cell_grid ? want : g_cell_grid);
      const int brute_max = g_cell_brute_max < BRUTE_CAP ? g_cell_brute_max : BRUTE_CAP;
      // (a short queue - a rank's share of a multi-GPU run - is balanced better chunk by chunk than in runs of four)
      const bool runs_launch = dl.list && (!dl.split || n_chunks >= (int64_t)g_cell_super_min_chunks);
      if (dl.list && !dl.split && n_chunks < (int64_t)g_cell_super_min_chunks) {  // (no weights: plain order, no lists)
        DeferList d2{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        d2.tile_list = dl.tile_list; d2.tile_c = dl.tile_c; d2.tile_count = dl.tile_count;
        dl = d2;
      }
      if (!g_cell_tiles) { dl.tile_list = nullptr; dl.tile_c = nullptr; dl.tile_count = nullptr; }
      // (2: only where an exhaustive evaluation by ONE wave - 150 us and more - would be the tail of the launch)
      if (g_cell_tiles == 2) dl.tail_items = (int64_t)g_cell_tail_waves * grid * 4 / 100;
      dl.listed_first = g_cell_listed_first;
      // (chunk-major order costs the L2 locality of a simplex's neighbouring chunks: measured a gain on queues up to
      // ~100 k chunks - cfg 2: 1.290 -> 1.246 ms per step, cfg 3: 4.72 -> 4.69 -, a small loss at cfg 5's 504 k)
      // (pilots first: where the heavy list is the dense rest of a cloud whose sparse simplices the witness sweep took
      // - cfg 2: 1.290 -> 1.246 ms - and on very long queues, where a simplex's chunks are far apart in time anyway -
      // cfg 5, 504 k chunks: sweep 6.51 -> 6.42 ms; in between - cfg 3, 110 k chunks, every simplex heavy - it costs
      // 0.4 %; option "cell_chunk_major" 2: always)
      dl.chunk_major = g_cell_chunk_major == 2 ? 2 : ((g_cell_chunk_major && n_chunks > ns) ? 1 : 0);
      dl.pilot_min_items = g_cell_chunk_major_max;
      dl.drop = g_cell_drop;
      CellParams cp{pts, nodes, lv, verts, plane_tab, weights, k1, R, ns, alpha, g_cell_exh_dense, g_cell_exh_sparse,
                    brute_max, g_cell_tries, g_cell_exh_tries, g_cell_retry_pct, g_cell_retry_keep, queue, out, flag_list,
                    flag_count, stats, acc, dl, dg, 0};
      // blocks of consecutive items per XCD-local queue shard (flood_common.hpp): about a simplex and a half of chunks,
      // two simplices of runs of four
      const int qb = g_cell_queue_block;
#define FLOODER_CELL_LAUNCH(SUPER_, SPL_, QUEUE_)                                                                    \
  do {                                                                                                               \
    cp.queue = QUEUE_;                                                                                               \
    cp.qblk = qb < 0 ? -1 : (SUPER_ ? (qb > 2 ? qb - 2 : 0) : qb);                                                   \
    hipLaunchKernelGGL((cell_sweep_kernel<DIM, SUPER_, SPL_>), dim3(grid), dim3(256), 0, st, cp);                    \
  } while (0)
      if (dl.list) {
        // runs of four chunks against one shared stage, then whatever they deferred chunk by chunk
        // (split: every simplex is on the heavy list of a short queue - the split kernel was told - and the chunk
        // launch takes them heaviest first)
        if (runs_launch) FLOODER_CELL_LAUNCH(true, SPL_CHUNK, queue);
        FLOODER_CELL_LAUNCH(false, SPL_CHUNK, queue2);
      } else {
        FLOODER_CELL_LAUNCH(false, SPL_CHUNK, queue);
      }
      // ... and the tiles of the chunks whose neighbourhood overflowed the stage, one sample per lane
      if (dl.tile_list) FLOODER_CELL_LAUNCH(false, 1, queue3);
#undef FLOODER_CELL_LAUNCH
      return check_launch("cell_sweep");
    } else {
      return fail(FLOODER_E_ARG, "flooder_sweep_cell_f32: only dim 2 and 3");
    }
  }
};

}  // namespace

namespace flooder {
namespace {
struct PlanesDone { const float* verts; const float* tab; int64_t ns; hipStream_t st; };
thread_local PlanesDone g_planes_done = {nullptr, nullptr, 0, nullptr};
}  // namespace
void planes_done_for(const float* verts, const float* tab, int64_t n_simplices, hipStream_t st) {
  g_planes_done = PlanesDone{verts, tab, n_simplices, st};
}
bool planes_are_done(const float* verts, const float* tab, int64_t n_simplices, hipStream_t st) {
  const bool hit = g_planes_done.verts == verts && g_planes_done.tab == tab && g_planes_done.ns == n_simplices &&
                   g_planes_done.st == st && verts != nullptr;
  g_planes_done = PlanesDone{nullptr, nullptr, 0, nullptr};
  return hit;
}
int launch_simplex_planes(int dim, const float* verts, int k1, int64_t n_simplices, float* tab, hipStream_t st) {
  if (dim == 2)
    hipLaunchKernelGGL((simplex_planes_kernel<2>), dim3((unsigned)((n_simplices + 255) / 256)), dim3(256), 0, st, verts, k1,
                       n_simplices, tab);
  else if (dim == 3)
    hipLaunchKernelGGL((simplex_planes_kernel<3>), dim3((unsigned)((n_simplices + 255) / 256)), dim3(256), 0, st, verts, k1,
                       n_simplices, tab);
  else
    return fail(FLOODER_E_ARG, "simplex planes: only dim 2 and 3");
  return check_launch("simplex_planes");
}
}  // namespace flooder

namespace {

// Order-preserving split of 0 .. n-1 by weight[i] <= limit: one block, ballot scans (n is a few thousand).
// Runs of four chunks pay in sparse volumetric clouds (most runs fit the stage); where fewer than half of the simplices
// are sparse (weight <= sparse_limit) the whole sweep stays chunk by chunk: every simplex on the heavy list.
__global__ __launch_bounds__(1024) void split_simplices_kernel(const float* __restrict__ weight, int n, float limit,
                                                               float sparse_limit, int32_t* __restrict__ light,
                                                               int32_t* __restrict__ heavy, int32_t* __restrict__ counts) {
  // (a negative weight marks a simplex the witness sweep has already handled: on neither list)
  __shared__ int s_cl[16], s_ch[16];
  __shared__ int s_sparse, s_act;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) { s_sparse = 0; s_act = 0; }
  __syncthreads();
  int mine = 0, act = 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float w = weight[i];
    mine += (w >= 0.f && w <= sparse_limit) ? 1 : 0;
    act += w >= 0.f ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mine += __shfl_xor(mine, o);
    act += __shfl_xor(act, o);
  }
  if (lane == 0) { atomicAdd(&s_sparse, mine); atomicAdd(&s_act, act); }
  __syncthreads();
  const bool all_heavy = 2 * s_sparse < s_act;  // every simplex heavy, in the given order
  int n_light = 0, n_heavy = 0;  // (block-uniform running totals)
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const float w = i < n ? weight[i] : -1.f;
    const bool is_light = w >= 0.f && !all_heavy && w <= limit;
    const bool is_heavy = w >= 0.f && !is_light;
    const unsigned long long ml = __ballot(is_light), mh = __ballot(is_heavy);
    if (lane == 0) { s_cl[wv] = __popcll(ml); s_ch[wv] = __popcll(mh); }
    __syncthreads();
    int bl = 0, tl = 0, bh = 0, th = 0;
    for (int w_ = 0; w_ < 16; ++w_) {
      bl += w_ < wv ? s_cl[w_] : 0;
      tl += s_cl[w_];
      bh += w_ < wv ? s_ch[w_] : 0;
      th += s_ch[w_];
    }
    if (is_light) light[n_light + bl + __builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0))] = i;
    if (is_heavy) heavy[n_heavy + bh + __builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0))] = i;
    n_light += tl;
    n_heavy += th;
    __syncthreads();
  }
  if (threadIdx.x == 0) { counts[0] = n_light; counts[1] = n_heavy; counts[2] = 1; }
}

// A SHORT queue (a rank's share of a multi-GPU run: no runs of four, every simplex on the heavy list) in DESCENDING
// weight class (powers of two around the limit), the given order kept inside a class: one block.  The persistent kernels pop their items in list order, and the long items - the
// chunks of the densest simplices, whose neighbourhood overflows the stage and is evaluated exhaustively by ONE wave
// for 150 - 200 us - used to sit wherever the axis order put them: the last of them to start was the tail of the
// launch (a quarter of it on cfg 2, half of it on an eighth of cfg 2).
// Runs of four chunks pay in sparse volumetric clouds (most runs fit the stage); where fewer than half of the simplices
// are sparse (weight <= sparse_limit), or the caller says so (runs_allowed = 0: a short queue), every simplex is heavy.
constexpr int SPLIT_THREADS_LONG = 512;  // the one-launch split of a long queue: all simplices, a dozen per thread
constexpr int SPLIT_THREADS = 256;  // (1024 threads leave 128 VGPRs each: the class arrays spilled, and scratch costs a launch 25 us)
constexpr int SPLIT_CLASSES = 9;  // weight > limit x 32, 16, 8, 4, 2, 1, 1/2, 1/4, rest
// (a stable counting sort: every thread owns a run of consecutive simplices, counts its classes, one block-wide
// exclusive scan per class - wave scans + partial sums through LDS - and every thread writes its run: two barriers)
// reorder != 0: the heavy list of split_simplices_kernel (counts[1] entries) is put into class order IN PLACE (its
// entries are staged in LDS first; longer than the stage: left as it is) and the counts are not touched.
template <int NT>
__global__ __launch_bounds__(NT) void class_order_kernel(const float* __restrict__ weight, int n, float limit,
                                                               float sparse_limit, int runs_allowed, int reorder,
                                                               int32_t* __restrict__ light, int32_t* __restrict__ heavy,
                                                               int32_t* __restrict__ counts) {
  __shared__ int s_cnt[NT / 64][SPLIT_CLASSES + 1];
  constexpr int ID_LDS = 7680;
  __shared__ int s_id[ID_LDS];
  if (reorder) {
    n = counts[1];
    // (no light list: a cloud too dense for runs, every simplex in the given order - measured: reordering those
    // costs cfg 3 20 us and gains nothing)
    if (n > ID_LDS || n < 2 || counts[0] == 0) return;
    for (int i = threadIdx.x; i < n; i += NT) s_id[i] = heavy[i];
    __syncthreads();
  }
  // the weights come in coalesced and are read back run by run from LDS (a run of consecutive simplices per thread
  // straight from memory is a chain of dependent cache misses: 34 us for 6000 simplices instead of 8)
  constexpr int W_LDS = 7680;
  __shared__ float s_w[W_LDS];
  const bool staged = n <= W_LDS;
  if (staged) {
    for (int base = 0; base < n; base += NT * 8) {  // (eight loads in flight: the loop is latency, not bytes)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * NT + (int)threadIdx.x;
        v[u] = i < n ? weight[reorder ? s_id[i] : i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * NT + (int)threadIdx.x;
        if (i < n) s_w[i] = v[u];
      }
    }
    __syncthreads();
  }
  auto wt = [&](int i) -> float { return staged ? s_w[i] : weight[i]; };  // (reorder: always staged)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool by_class = (runs_allowed & 2) != 0;
  // runs_allowed & 4: a LONG queue in one launch (what split_simplices_kernel + a reorder launch used to do in two):
  // the light simplices are ONE class, kept in the given order for the runs of four; without runs every simplex goes
  // on the heavy list in the given order (a cloud too dense for runs: class order costs cfg 3 20 us and gains nothing)
  const bool one_light = (runs_allowed & 4) != 0;
  constexpr int FIRST_LIGHT = 6;  // classes 6, 7, 8: weight <= limit
  auto cls = [&](float w) -> int {  // 0 = heaviest; -1: handled by the witness sweep already (on no list)
    if (w < 0.f) return -1;
    if (!by_class) return w > limit ? 5 : 6;
    int c = 0;
    float t = limit * 32.f;
#pragma unroll
    for (int k = 0; k < SPLIT_CLASSES - 1; ++k) {
      c += w > t ? 0 : 1;
      t *= 0.5f;
    }
    return one_light && c > FIRST_LIGHT ? FIRST_LIGHT : c;
  };
  const int per = ((n + NT - 1) / NT) | 1;  // (odd: the runs start in different LDS banks)
  const int i0 = threadIdx.x * per < n ? threadIdx.x * per : n;
  const int i1 = i0 + per < n ? i0 + per : n;
  int cnt[SPLIT_CLASSES + 1];  // [SPLIT_CLASSES]: sparse simplices
#pragma unroll
  for (int k = 0; k <= SPLIT_CLASSES; ++k) cnt[k] = 0;
  for (int ib = i0; ib < i1; ib += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ib + u < i1 ? wt(ib + u) : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (ib + u < i1) {
        const int c = cls(v[u]);
#pragma unroll
        for (int k = 0; k < SPLIT_CLASSES; ++k) cnt[k] += c == k ? 1 : 0;
        cnt[SPLIT_CLASSES] += (v[u] >= 0.f && v[u] <= sparse_limit) ? 1 : 0;
      }
    }
  }
  int excl[SPLIT_CLASSES];  // simplices of the class in the threads before this one (this wave)
#pragma unroll
  for (int k = 0; k <= SPLIT_CLASSES; ++k) {
    int v = cnt[k];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o);
      v += lane >= o ? u : 0;
    }
    if (k < SPLIT_CLASSES) excl[k] = v - cnt[k];
    if (lane == 63) s_cnt[wv][k] = v;
  }
  __syncthreads();
  int before[SPLIT_CLASSES], total[SPLIT_CLASSES];
  int n_sparse = 0;
#pragma unroll
  for (int k = 0; k < SPLIT_CLASSES; ++k) { before[k] = 0; total[k] = 0; }
  for (int w = 0; w < NT / 64; ++w) {
#pragma unroll
    for (int k = 0; k < SPLIT_CLASSES; ++k) {
      const int v = s_cnt[w][k];
      before[k] += w < wv ? v : 0;
      total[k] += v;
    }
    n_sparse += s_cnt[w][SPLIT_CLASSES];
  }
  int n_act = 0;  // simplices still to be swept
#pragma unroll
  for (int k = 0; k < SPLIT_CLASSES; ++k) n_act += total[k];
  const bool runs = (runs_allowed & 1) != 0 && 2 * n_sparse >= n_act;
  const bool flat = !by_class && !runs && n_act == n;  // one list in the given order
  const bool given = one_light && !runs && !flat;      // ... of the simplices still to be swept
  int pos[SPLIT_CLASSES];  // where this thread's first simplex of the class goes
  int n_light = 0, n_heavy = 0;
  int pos_given = 0;
#pragma unroll
  for (int k = 0; k < SPLIT_CLASSES; ++k) {
    const bool is_light = runs && k >= FIRST_LIGHT;
    pos[k] = (is_light ? n_light : n_heavy) + before[k] + excl[k];
    pos_given += before[k] + excl[k];
    if (is_light) n_light += total[k];
    else n_heavy += total[k];
  }
  if (flat) {
    for (int i = i0; i < i1; ++i) heavy[i] = reorder ? s_id[i] : i;
  } else if (given) {
    // (the threads own consecutive runs and the waves consecutive threads: everything before this thread's run)
    for (int i = i0; i < i1; ++i)
      if (wt(i) >= 0.f) heavy[pos_given++] = i;
  } else {
    for (int ib = i0; ib < i1; ib += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ib + u < i1 ? wt(ib + u) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (ib + u < i1) {
          const int c = cls(v[u]);
#pragma unroll
          for (int k = 0; k < SPLIT_CLASSES; ++k) {
            if (c == k) {
              int32_t* dst = (runs && k >= FIRST_LIGHT) ? light : heavy;
              dst[pos[k]] = reorder ? s_id[ib + u] : ib + u;
              pos[k] += 1;
            }
          }
        }
      }
    }
  }
  if (threadIdx.x == 0 && !reorder) { counts[0] = n_light; counts[1] = n_heavy; counts[2] = 1; }
}

// every leaf of the box tree (16 consecutive points of the curve order) adds its point count to the fine cell under
// the centre of its box: a sixteenth of the atomics of a pass over the points, accurate to a leaf
template <int DIM>
__global__ __launch_bounds__(256) void density_leaves_kernel(const float* __restrict__ nodes, int64_t n_pts,
                                                             int64_t n_leaves, const float* __restrict__ cbox,
                                                             int32_t* __restrict__ grid) {
  constexpr int DP = padded_dim(DIM);
  typedef DensCfg<DIM> DC;
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_leaves) return;
  float lo[DP], hi[DP];
  load_row<DP>(nodes + l * 2 * DP, lo);
  load_row<DP>(nodes + l * 2 * DP + DP, hi);
  int fine = 0;
#pragma unroll
  for (int k = DIM - 1; k >= 0; --k) {
    const float e = cbox[8 + k] - cbox[k];
    const float sc = e > 0.f ? (float)DC::G / e : 0.f;
    int ck = (int)((0.5f * (lo[k] + hi[k]) - cbox[k]) * sc);
    ck = ck < 0 ? 0 : (ck >= DC::G ? DC::G - 1 : ck);
    fine = fine * DC::G + ck;
  }
  const int64_t left = n_pts - l * LEAF;
  atomicAdd(&grid[fine], (int)(left < LEAF ? left : LEAF));
}

int sweep_cell_entry(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes, const float* verts,
                     float* plane_tab, const float* weights, int k1, int R, int64_t n_simplices, float alpha, int32_t* queue,
                     uint32_t* out_d2, int32_t* flag_list, int32_t* flag_count, uint64_t* stats, FaceAcc acc,
                     DeferList dl, int32_t* queue2, int32_t* queue3, DensGrid dg, void* stream, const char* who) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !plane_tab || !weights || !queue || !out_d2 || !flag_list || !flag_count ||
      n_pts < 1 || k1 < 1 || k1 > FLOODER_MAX_VERTS || R < 0 || !(alpha > 0.f))
    return fail(FLOODER_E_ARG, who);
  if (dim != 2 && dim != 3) return fail(FLOODER_E_ARG, "cell sweep: only dim 2 and 3");
  if (n_simplices * (int64_t)((R + 63) / 64) > 0x7fffffffLL)
    return fail(FLOODER_E_ARG, "cell sweep: too many (simplex, tile) pairs");
  const Levels lv = make_levels(n_pts);
  // the kernel addresses rows and node boxes with 32-bit byte offsets
  if ((n_pts + FLOODER_BVH_LEAF) * (int64_t)(padded_dim(dim) * sizeof(float)) >= (1LL << 32) ||
      total_nodes(lv) * (int64_t)(2 * padded_dim(dim) * sizeof(float)) >= (1LL << 32))
    return fail(FLOODER_E_ARG, "cell sweep: cloud too large for the cell sweep (use the tree sweep)");
  return dispatch_dim<CellOp>(dim, pts_sorted, nodes, lv, verts, plane_tab, weights, k1, R, n_simplices, alpha, queue,
                              out_d2, flag_list, flag_count, reinterpret_cast<unsigned long long*>(stats), acc,
                              dl, queue2, queue3, dg, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int flooder_sweep_cell_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                           const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                           float alpha, int32_t* queue, uint32_t* out_d2, int32_t* flag_list, int32_t* flag_count,
                           float* plane_scratch, const int32_t* density_grid, const float* cloud_box, uint64_t* stats,
                           void* stream) {
  DensGrid dg;
  if (density_grid && cloud_box) { dg.grid = density_grid; dg.box = cloud_box; }
  return sweep_cell_entry(pts_sorted, n_pts, dim, nodes, verts, plane_scratch, weights, k1, R, n_simplices, alpha, queue, out_d2,
                          flag_list, flag_count, stats, FaceAcc{nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr},
                          DeferList{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0}, nullptr, nullptr, dg, stream,
                          "flooder_sweep_cell_f32: bad argument");
}

int flooder_sweep_cell_faces_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                 const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                                 float alpha, int32_t* queue, uint32_t* d2_scratch, const uint32_t* memb,
                                 int n_faces, uint32_t* face_bits, const int32_t* face_slot, int32_t* flag_list,
                                 int32_t* flag_count, uint32_t* flag_key, int32_t* flag_hist, uint64_t* top,
                                 int32_t* top_list, int32_t* top_count, int32_t* defer_list,
                                 float* defer_c, int32_t* defer_ctl, const float* simplex_weight,
                                 int32_t* light_list, int32_t* heavy_list, float* plane_scratch,
                                 const int32_t* density_grid, const float* cloud_box, uint64_t* stats, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  DensGrid dg;
  if (density_grid && cloud_box) { dg.grid = density_grid; dg.box = cloud_box; }
  if (!memb || !face_bits || n_faces < 1 || n_faces > 32 || (top && (!top_list || !top_count)) ||
      (defer_list && (!defer_c || !defer_ctl)) || (simplex_weight && (!defer_list || !light_list || !heavy_list)) ||
      (flag_key && (!flag_hist || !top)) || n_simplices > 0x7fffffffLL)
    return fail(FLOODER_E_ARG, "flooder_sweep_cell_faces_f32: bad argument");
  if (!simplex_weight) light_list = heavy_list = nullptr;
  if (simplex_weight) {  // split the simplices (order kept) into the light and the heavy list
    const bool long_queue = n_simplices * (int64_t)((R + 255) / 256) >= (int64_t)g_cell_super_min_chunks;
    // (more simplices than the one-launch form stages in LDS - its runs would be read from memory, a chain of cache
    // misses: 50 us for cfg 5's 25 217 - take the pair; option 2: the pair always, for A/B runs and the tests)
    if (long_queue && (g_cell_split_launches == 2 || n_simplices > 7680)) {
      hipLaunchKernelGGL(split_simplices_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, simplex_weight,
                         (int)n_simplices, (float)g_cell_super_weight, (float)g_cell_super_sparse, light_list, heavy_list,
                         defer_ctl + 2);
      // ... and the heavy list heaviest first: its densest simplices hold the chunks that one wave evaluates
      // exhaustively for 150 us and more, and the last of them to start was the tail of the chunk launch
      if (g_cell_weight_classes)
        hipLaunchKernelGGL(class_order_kernel<SPLIT_THREADS>, dim3(1), dim3(SPLIT_THREADS), 0, (hipStream_t)stream, simplex_weight,
                           (int)n_simplices, (float)g_cell_super_weight, (float)g_cell_super_sparse, 2, 1, light_list,
                           heavy_list, defer_ctl + 2);
    } else if (long_queue) {
      // light list in the given order + heavy list heaviest first (its densest simplices hold the chunks that one wave
      // evaluates exhaustively for 150 us and more: started last they were the tail of the chunk launch) in ONE launch
      hipLaunchKernelGGL(class_order_kernel<SPLIT_THREADS_LONG>, dim3(1), dim3(SPLIT_THREADS_LONG), 0, (hipStream_t)stream, simplex_weight,
                         (int)n_simplices, (float)g_cell_super_weight, (float)g_cell_super_sparse,
                         (g_cell_weight_classes ? 2 : 0) | 1 | 4, 0, light_list, heavy_list, defer_ctl + 2);
    } else {
      hipLaunchKernelGGL(class_order_kernel<SPLIT_THREADS>, dim3(1), dim3(SPLIT_THREADS), 0, (hipStream_t)stream, simplex_weight,
                         (int)n_simplices, (float)g_cell_super_weight, (float)g_cell_super_sparse,
                         g_cell_weight_classes ? 2 : 0, 0, light_list, heavy_list, defer_ctl + 2);
    }
  }
  DeferList dl{defer_list, defer_c, defer_list ? defer_ctl : nullptr, light_list, heavy_list,
               light_list ? defer_ctl + 2 : nullptr, g_cell_super_n0};
  if (defer_list) {  // the tile list lives behind the chunk list (the caller sizes both buffers for 5 x S x chunks)
    const int64_t n_chunk_slots = n_simplices * (int64_t)((R + 255) / 256);
    dl.tile_list = defer_list + n_chunk_slots;
    dl.tile_c = defer_c + n_chunk_slots;
    dl.tile_count = defer_ctl + 6;
  }
  return sweep_cell_entry(pts_sorted, n_pts, dim, nodes, verts, plane_scratch, weights, k1, R, n_simplices, alpha, queue,
                          d2_scratch, flag_list, flag_count, stats,
                          FaceAcc{memb, face_bits, n_faces, reinterpret_cast<unsigned long long*>(top), top_list,
                                  top_count, face_slot, flag_key, flag_hist},
                          dl, queue + FLOODER_QUEUE_WORDS, queue + 2 * FLOODER_QUEUE_WORDS, dg,
                          stream, "flooder_sweep_cell_faces_f32: bad argument");
}


int64_t flooder_density_grid_words(int dim) {   // (the grid, then four words about it: cloud_kind_kernel)
  if (dim == 2) return DensCfg<2>::NF + KIND_WORDS;
  if (dim == 3) return DensCfg<3>::NF + KIND_WORDS;
  return 0;
}

int flooder_density_grid_f32(const float* nodes, int64_t n_pts, int dim, const float* cloud_box, int32_t* grid,
                             void* stream) {
  if (!nodes || !cloud_box || !grid || n_pts < 1 || (dim != 2 && dim != 3))
    return fail(FLOODER_E_ARG, "flooder_density_grid_f32: bad argument (dim 2 and 3 only)");
  const int64_t n_leaves = (n_pts + LEAF - 1) / LEAF;
  hipStream_t st = (hipStream_t)stream;
  if (dim == 2) {
    hipLaunchKernelGGL((density_leaves_kernel<2>), dim3((unsigned)((n_leaves + 255) / 256)), dim3(256), 0, st, nodes, n_pts,
                       n_leaves, cloud_box, grid);
  } else {
    hipLaunchKernelGGL((density_leaves_kernel<3>), dim3((unsigned)((n_leaves + 255) / 256)), dim3(256), 0, st, nodes, n_pts,
                       n_leaves, cloud_box, grid);
  }
  return check_launch("density_grid");
}


int flooder_cloud_kind(int32_t* density_grid, int dim, void* stream) {
  if (!density_grid || (dim != 2 && dim != 3)) return fail(FLOODER_E_ARG, "flooder_cloud_kind: bad argument (dim 2 and 3 only)");
  launch_cloud_kind(dim, density_grid, (hipStream_t)stream);
  return check_launch("cloud_kind");
}

}  // extern "C"
