// flood_finish.hip - exact finish of the cell sweep's open tiles when only per-face MAXIMA are wanted (gfx950).
//
// The filtration value of a face is a maximum over its samples, so a sample whose upper bound cannot exceed the
// running maximum of every face it lies on needs no exact nearest neighbour.  face_bits[s, f] holds the running
// maximum over the samples of face f that are settled (exact); it only ever receives exact values, so it never
// exceeds the true maximum, and a sample i with  best_i <= min over its faces f of face_bits[s, f]  can be dropped:
// true_i <= best_i <= face_bits[s, f] <= max_f.  The sample that attains max_f is dropped only when another sample
// has already delivered the same value, otherwise it stays live to the end of its traversal, is then exact, and
// delivers it.  Face values therefore equal the exhaustive result bit for bit.
//
// Three passes over the list of flagged tiles (64 consecutive samples of a simplex each):
//   probe  one greedy descent of the box tree per tile (nearest child at every level, one leaf evaluated) gives
//          every sample a finite upper bound; the tile with the largest one becomes its simplex's "top" tile;
//   top    the top tile of every simplex is finished exactly first (one per simplex, all at once): it usually holds
//          the simplex's deepest sample, so face_bits of the full simplex is (almost) final afterwards;
//   rest   all other tiles: most of their samples are dropped against face_bits straight away; what stays live
//          (samples on lower-dimensional faces, near-ties) is traversed with bounds that only consider the live
//          samples (tile box, pruning radius M, refine loop over live lanes only).
// Traversal as in flood_bvh.hip: wave-uniform, nearest first, lane = child box, leaf points through SGPRs.

#include "flood_common.hpp"
#include "flood_bvh.hpp"
#include "flood_planes.hpp"

using namespace flooder;

namespace {

constexpr float SAFE = 0.99999f;
constexpr int SHORT_LIST = 1024;
#ifndef FLOODER_TOP_NODES
#define FLOODER_TOP_NODES 1024
#endif
#ifndef FLOODER_FINISH_PRIO_AFTER
#define FLOODER_FINISH_PRIO_AFTER 32  // leaves after which a tile's wave raises its issue priority (0: never)
#endif
#ifndef FLOODER_FINISH_PRIO
#define FLOODER_FINISH_PRIO 3
#endif
#ifndef FLOODER_FINISH_WAVES
#define FLOODER_FINISH_WAVES 4
#endif
constexpr int TOP_NODES = FLOODER_TOP_NODES;  // nodes of the two top tree levels kept in LDS (32 KB in 3D)

// Hard tiles.  A sample near the medial axis of the cloud is almost equidistant to thousands of leaves, and a tile of
// such samples keeps ONE wave busy for milliseconds while the rest of the chip idles at the end of the pass (and on
// a shard of the simplices nothing else is left to hide it).  A tile that has evaluated more leaves than the budget
// (over all its rounds so far; the budget scales with the length of the list, see the kernel) abandons its round:
// the bounds reached so far go back to the scratch matrix and the tile is appended to a hard list.  The next launch
// (mode 3, TEAM = true) gives every hard tile to a whole workgroup of 16 waves: all of them hold the tile's samples
// and take the same decisions from the same data (the face maxima are read by wave 0 and passed on through LDS), but
// wave j descends only into the level-1 nodes j, j + 16, ... (an interleaved 1/16 of the sorted cloud).  At the end
// of a round the minima are combined in LDS (integer atomic min), wave 0 delivers, and the next round starts - all
// rounds of the tile back to back, three workgroup barriers each.
// Before that (late round 5): a search is a chain of DEPENDENT steps - select, test, fetch, evaluate, reduce - that
// uses a quarter of its wave's issue slots, and four such waves share a SIMD: a long search is long mostly because it
// waits for its turn.  A wave whose tile has evaluated FLOODER_FINISH_PRIO_AFTER leaves raises its issue priority
// (s_setprio): the few long searches run nearly unimpeded, the many short ones give up slots they were not the tail
// of.  cfg 3: pass 1.87 -> 1.67 ms (1.60 with the budget at 40 instead of 14 - but a rank's share of the simplices,
// whose budget shrinks with its list, then hands its long searches over late AND still needs the team launch:
// 1.54 instead of 1.21 ms on a half, so 14 it stays).
struct HardLists {
  const unsigned long long* ent_in;   // item | sub << 32 | subs << 40 | one_round_only << 48
  const unsigned long long* mask_in;  // lanes of the focus samples of the abandoned round
  const int32_t* cnt_in;
  unsigned long long* ent_out;
  unsigned long long* mask_out;
  int32_t* cnt_out;
  int cap;     // entries per list
  int budget;  // scale of the leaf budget of a tile (0: no hard tiles)
  int budget_min = 64;  // ... and what a tile of a SHORT list may evaluate at least before it counts as hard (leaves)
};

constexpr int TEAM_WAVES = 16;

// WAVES: waves per workgroup of the per-wave passes.  4 (default): four workgroups per CU, 4 waves per SIMD.  8: eight
// waves share ONE staged tree top - three workgroups per CU, 6 waves per SIMD at an 80-register cap (a dozen spills) -
// for clouds whose tree is deep: a search there is a longer chain of dependent steps and more waves hide more of it
// (cfg 5, 16 M points: finish 1.68 -> 1.47 ms, step 8.71 -> 8.49; cfg 3, 1 M points: 1.580 -> 1.564; cfg 2: 0.077 ->
// 0.082 - chosen by cloud size, option "finish_wide_points").
template <int DIM, bool TEAM, int WAVES = 4>
__global__ __launch_bounds__(TEAM ? 64 * TEAM_WAVES : 64 * WAVES, TEAM ? 4 : (WAVES == 8 ? 6 : FLOODER_FINISH_WAVES)) void finish_faces_kernel(
    const float* __restrict__ pts, const float* __restrict__ nodes, Levels lv,
    const float* __restrict__ verts, const float* __restrict__ weights, int k1, int R,
    int64_t n_simplices, const int32_t* __restrict__ flag_list, const int32_t* __restrict__ flag_sorted,
    const int32_t* __restrict__ flag_count,
    int mode /* 0 probe, 1 top tiles, 2 rest, 3 hard entries */, int subs_max, int items_cap, int refine_pct, float focus_frac, int refresh_every, int32_t* __restrict__ queue,
    uint32_t* __restrict__ d2, FaceAcc acc, unsigned long long* __restrict__ top,
    int32_t* __restrict__ top_list, int32_t* __restrict__ top_count, HardLists hl,
    unsigned long long* __restrict__ stats) {
  constexpr int DP = padded_dim(DIM);
  constexpr int NW = TEAM ? TEAM_WAVES : WAVES;  // waves of the workgroup
  __shared__ float s_lb[NW][MAXL][FAN];
  __shared__ int64_t s_grp[NW][MAXL];
  // TEAM: the tile being worked on, wave 0's reading of the face maxima, the combined minima of a round
  __shared__ long long s_titem;
  __shared__ uint32_t s_tfb[64];
  __shared__ uint32_t s_tbest[64];
  // the two top levels of the box tree (993 nodes for a million points) live in LDS, shared by the block's waves:
  // every search starts there, and a focus round restarts there - only the leaf groups are fetched from L2
  __shared__ float s_top[TOP_NODES * 2 * DP];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int tiles = (R + 63) >> 6;
  const int n_list = flag_count[0];
  // few items: split each tile over several waves (short tail, tight boxes); many: 64 distinct samples per wave
  int subs = 1;
  int64_t n_base = mode == 1 ? (int64_t)top_count[0] : (int64_t)n_list;  // (top_count: filled by the probe)
  if constexpr (TEAM) {
    n_base = hl.cnt_in[0];
    if (n_base > hl.cap) n_base = hl.cap;  // (entries that did not fit were finished by their producer)
  }
  if (mode == 2) {
    subs = subs_max;
    while (subs > 1 && n_base * subs > items_cap) subs >>= 1;
    if (subs_max > 1 && n_base * 64 <= (int64_t)gridDim.x * NW) subs = 64;
  }
  if (n_base == 0) return;
  // a short list goes straight to the last pass (every tile is searched at once anyway; two launches saved)
  if (mode < 2 && n_list <= SHORT_LIST) return;
  const int64_t n_items = TEAM ? n_base : n_base * subs;
  // a workgroup that will find nothing leaves before it stages the tree top (32 KB from L2): with a wave per item
  // (the static deal below) those are the workgroups behind the last item - 800 of 1024 on cfg 2's short list -, in a
  // team launch the workgroups beyond the number of entries (each takes at least one)
  if (TEAM ? (int64_t)blockIdx.x >= n_items
           : (n_items <= (int64_t)gridDim.x * NW && (int64_t)blockIdx.x * NW >= n_items)) return;
  const int topl = lv.n_levels - 1;
  // (the split of a hard entry is by level-1 node: a tree without that level has no hard entries)
  const bool budgeted = hl.budget > 0 && hl.ent_out != nullptr && topl >= 1;
  // leaves a tile may evaluate before it counts as hard: hl.budget per flagged tile and wave - a multiple of the
  // share of the whole list one wave would work off if the tiles were all alike (the last pass evaluates about 25
  // leaves per tile), so that only tiles that would outlast a balanced pass are split; a short list (a shard of the
  // simplices) lowers it with the share
  // (top pass: one point query per simplex and about one query per wave - the pass lasts as long as its longest
  // query, so the budget is a fixed few dozen leaves)
  const int64_t budget_raw = mode == 1 ? (int64_t)hl.budget * 4 : (int64_t)hl.budget * n_list / ((int64_t)gridDim.x * NW);
  // (a short list - cfg 2's 200 tiles, a rank's share - used to get min(budget, 64) = 14 leaves: every dense tile went
  // to the team launch, 50 us of launch for work the first pass does in 20; cfg 2 finish 0.115 -> 0.082 ms)
  const int budget_min = hl.budget_min;
  const int budget_eff = (int)(budget_raw < budget_min ? budget_min : (budget_raw > (1 << 20) ? (1 << 20) : budget_raw));
  // stage levels topl (first) and topl - 1 (behind it) when they fit
  const int top_cnt = (int)lv.count[topl];
  const int sub_cnt = topl >= 2 ? (int)lv.count[topl - 1] : 0;    // (level 0 = leaves: fetched per group)
  const int staged_sub = (sub_cnt > 0 && top_cnt + sub_cnt <= TOP_NODES) ? sub_cnt : 0;
  const int stage_min_lvl = topl >= 1 ? (staged_sub ? topl - 1 : topl) : MAXL;  // levels >= this are read from LDS
  if (topl >= 1) {
    const float* src_top = nodes + lv.off[topl] * 2 * DP;
    for (int i = threadIdx.x; i < top_cnt * 2 * DP; i += 64 * NW) s_top[i] = src_top[i];
    if (staged_sub) {
      const float* src_sub = nodes + lv.off[topl - 1] * 2 * DP;
      for (int i = threadIdx.x; i < staged_sub * 2 * DP; i += 64 * NW) s_top[top_cnt * 2 * DP + i] = src_sub[i];
    }
    __syncthreads();
  }
  unsigned long long n_leaf_eval = 0, n_leaf_test = 0, n_node_test = 0, n_dropped = 0, n_live0 = 0, n_rounds = 0;
#ifdef FLOODER_PHASE_TIMERS
  unsigned long long tf[6] = {0, 0, 0, 0, 0, 0}, tf_prev = __builtin_amdgcn_s_memtime();
#define FIN_PHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tf[i] += t_ - tf_prev; tf_prev = t_; } while (0)
#else
#define FIN_PHASE(i) do {} while (0)
#endif

  const int64_t wave_id = (int64_t)blockIdx.x * NW + wv;
#ifdef FLOODER_WAVE_END_FIN
  const unsigned long long t_wave0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_longest = 0, t_item = 0;  // longest single item of this wave (10 ns ticks)
  unsigned long long c_longest = 0, c_item[4] = {0, 0, 0, 0};  // its rounds, leaf evaluations, expansions, leaf tests
  long long g_prev = -1;
  unsigned long long m_prev = 0;
#endif
  const bool static_deal = !TEAM && n_items <= (int64_t)gridDim.x * NW;
  bool dealt = false;
  int q_shard = (int)(wave_id % QSHARDS), q_tried = 0;
  for (;;) {
#ifdef FLOODER_WAVE_END_FIN
    if (t_item) {
      const unsigned long long d = __builtin_amdgcn_s_memrealtime() - t_item;
      if (mode == 2 && stats && lane == 0 && g_prev >= 0 && g_prev < 200000) {
        stats[40000 + 2 * g_prev] = d | ((n_rounds - c_item[0]) << 32) | ((n_leaf_eval - c_item[1]) << 44);
        stats[40000 + 2 * g_prev + 1] = m_prev;
      }
      if (d > t_longest) {
        t_longest = d;
        c_longest = ((n_rounds - c_item[0]) << 48) | ((n_leaf_eval - c_item[1]) << 32) | ((n_node_test - c_item[2]) << 16) |
                    ((n_leaf_test - c_item[3]) & 0xffffull);
      }
    }
#endif
    int64_t g;
    if constexpr (TEAM) {  // one pop for the workgroup
      __syncthreads();
      if (threadIdx.x == 0) s_titem = (long long)atomicAdd(queue, 1);
      __syncthreads();
      g = (int64_t)s_titem;
      if (g >= n_items) break;
    } else if (static_deal) {
      if (dealt || wave_id >= n_items) break;
      dealt = true;
      g = wave_id;
    } else {
      g = queue_pop(queue, q_shard, q_tried, n_items, lane);  // sharded heads (flood_common.hpp)
      if (g < 0) break;
    }
    FIN_PHASE(5);  // (queue pop, bookkeeping)
#ifdef FLOODER_WAVE_END_FIN
    t_item = __builtin_amdgcn_s_memrealtime();
    g_prev = g;
    m_prev = 0;
    c_item[0] = n_rounds; c_item[1] = n_leaf_eval; c_item[2] = n_node_test; c_item[3] = n_leaf_test;
#endif
    int sub, subs_i = subs;
    const int part = TEAM ? wv : -1;  // TEAM: this wave's share of the level-1 nodes
    bool stop_after = false;
    unsigned long long fmask = 0ull;
    int64_t s;
    int item;  // global tile id: simplex * tiles + tile
    if constexpr (TEAM) {
      const unsigned long long ent = hl.ent_in[g];
      item = (int)(uint32_t)(ent & 0xffffffffull);
      sub = (int)((ent >> 32) & 0xffull);
      subs_i = (int)((ent >> 40) & 0xffull);
      stop_after = ((ent >> 48) & 1ull) != 0ull;
      fmask = hl.mask_in[g];
      s = item / tiles;
    } else {
      sub = (int)(g % subs);
      g /= subs;
    }
    const int per_sub = 64 / subs_i;
    if constexpr (TEAM) {
    } else if (mode == 1) {
      s = top_list[g];
      item = (int)(uint32_t)(top[s] & 0xffffffffull);
    } else {
      item = (n_list > SHORT_LIST && flag_sorted ? flag_sorted : flag_list)[g];
      s = item / tiles;
    }
    const int tile = item - (int)(s * tiles);
    const int slane = sub * per_sub + (lane & (per_sub - 1));  // sample slot of this lane inside the tile
    const bool mine = lane < per_sub;                           // (other lanes hold replicas)
    int r = tile * 64 + slane;
    const bool exists = r < R;
    if (!exists) r = R - 1;

    // ---- this lane's sample, its seed and the faces it lies on
    float p[DIM];
    const float* vs = verts + s * (int64_t)k1 * DIM;
#pragma unroll
    for (int k = 0; k < DIM; ++k) p[k] = 0.f;
    for (int j = 0; j < k1; ++j) {
      const float w = weights[(int64_t)r * k1 + j];
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[k] = __builtin_fmaf(w, vs[j * DIM + k], p[k]);
    }
    const uint32_t seed = d2[s * (int64_t)R + r];
    const bool settled = !exists || (seed & SETTLED_BIT) != 0u;  // settled by the cell sweep: delivered already
    float best = __uint_as_float(seed & ~SETTLED_BIT);
    const uint32_t mb = settled ? 0u : acc.memb[r];
    // this lane's face slot (lane f < n_faces holds the slot of face f of the simplex)
    const int64_t my_slot = lane < acc.n_faces ? acc.slot_of(s, lane) : 0;

    // ---- live set: unsettled samples whose upper bound still exceeds the running maximum of one of their faces
    bool live, done = settled || mb == 0u;  // done: settled exactly and delivered (or nothing to deliver)
    float M, tlo[DIM], thi[DIM];
    uint32_t fb_seen = 0u;  // lane f: the last value read of face f's running maximum (a lower bound of the current one)
    auto refresh = [&]() {
      // only the faces some not yet settled sample of the tile lies on are (re)loaded: usually one to three
      uint32_t um = wave_or_u32(done ? 0u : mb);
      uint32_t fb = 0xffffffffu;
      if (!TEAM || wv == 0) {
        if (lane < acc.n_faces && ((um >> lane) & 1u))
          fb = __hip_atomic_load(acc.face_bits + my_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if constexpr (TEAM) {  // one reading for the whole workgroup: its waves must agree on who is live
        __syncthreads();
        if (wv == 0) s_tfb[lane] = fb;
        __syncthreads();
        fb = s_tfb[lane];
      }
      if (fb != 0xffffffffu) fb_seen = fb;
      uint32_t thr = 0xffffffffu;
      while (um) {  // (wave-uniform)
        const int f = __builtin_ctz(um);
        um &= um - 1u;
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)fb, f);
        if ((mb >> f) & 1u) thr = v < thr ? v : thr;
      }
      live = !done && __float_as_uint(best) > thr;
      M = wave_max_f32(live ? best : -1.f);
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        tlo[k] = wave_min_f32(live ? p[k] : __builtin_inff());
        thi[k] = wave_max_f32(live ? p[k] : -__builtin_inff());
      }
    };
    refresh();
    if (mode == 0) {  // the probe descends for every unsettled sample (their bounds order the tiles)
      M = wave_max_f32(settled ? -1.f : best);
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        tlo[k] = wave_min_f32(settled ? __builtin_inff() : p[k]);
        thi[k] = wave_max_f32(settled ? -__builtin_inff() : p[k]);
      }
    }
#ifdef FLOODER_WAVE_END_FIN
    m_prev = (unsigned long long)__float_as_uint(M >= 0.f ? M : 0.f) | ((unsigned long long)__popcll(__ballot(live)) << 32);
#endif
    int pick = -1;
    if (mode == 1 && M >= 0.f) {
      // top pass: only the sample with the largest bound of the simplex's top tile is settled here - one point query
      // per simplex gives every face maximum a value close to its final one; the other samples of the tile are
      // handled by the last pass like those of any tile (most of them drop against that value)
      pick = __builtin_ctzll(__ballot(live && best == M));
      done = done || lane != pick;
      refresh();
    }
    if (sub == 0) n_live0 += __popcll(__ballot(live && mine));
    // nothing to do for this tile (mode 3: then no wave on the entry has, now or later - liveness only ever goes
    // away - and that the entry is never counted off loses nothing)
    if (!(M >= 0.f)) {
      if (sub == 0) ++n_dropped;
      continue;
    }

    // child `lane` of group `grp` at level `lvl`: its box (registers) and the lower bound to the live box
    float c_lo[DIM], c_hi[DIM];
    auto child_bounds = [&](int lvl, int64_t grp) -> float {
      const int64_t idx = grp * FAN + lane;
      float lb = __builtin_inff();
#pragma unroll
      for (int k = 0; k < DIM; ++k) { c_lo[k] = __builtin_inff(); c_hi[k] = -__builtin_inff(); }
      if (idx < lv.count[lvl]) {
        float lo[DP], hi[DP];
        if (lvl >= stage_min_lvl) {
          const float* nb = s_top + ((lvl == topl ? 0 : top_cnt) + (int)idx) * 2 * DP;
#pragma unroll
          for (int k = 0; k < DP; ++k) { lo[k] = nb[k]; hi[k] = nb[DP + k]; }
        } else {
          const float* nb = nodes + (lv.off[lvl] + idx) * 2 * DP;
          load_row<DP>(nb, lo);
          load_row<DP>(nb + DP, hi);
        }
        lb = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          c_lo[k] = lo[k];
          c_hi[k] = hi[k];
          const float gap = __builtin_fmaxf(__builtin_fmaxf(lo[k] - thi[k], tlo[k] - hi[k]), 0.f);
          lb = __builtin_fmaf(gap, gap, lb);
        }
        // one wave of several on a hard entry: only its share of the level-1 nodes
        if (TEAM && lvl == 1 && (int)(idx % TEAM_WAVES) != part) lb = __builtin_inff();
      }
      return lb;
    };
    // all 16 points of leaf c against this lane's sample (points through the scalar cache)
    auto eval_leaf = [&](int64_t c) {
      const float* cp = pts + c * (int64_t)LEAF * DP;
#pragma unroll
      for (int h = 0; h < LEAF; h += 8) {
        typename RowVec<DP>::type cc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cc[u] = load_uniform_row<DP>(cp + (h + u) * DP);
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          float da, db;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            const float ta = p[k] - cc[u][k];
            const float tb = p[k] - cc[u + 1][k];
            if (k == 0) {
              da = ta * ta;
              db = tb * tb;
            } else {
              da = __builtin_fmaf(ta, ta, da);
              db = __builtin_fmaf(tb, tb, db);
            }
          }
          best = __builtin_fminf(best, __builtin_fminf(da, db));
        }
      }
      ++n_leaf_eval;
    };

    if (mode == 0) {
      // ---- probe: greedy descent, nearest child at every level, then the nearest leaf of that group
      int64_t grp = 0;
      for (int lvl = topl; lvl >= 0; --lvl) {
        const float lb = child_bounds(lvl, grp);
        ++n_node_test;
        const float mn = wave_min_f32(lb);
        const int j = __builtin_ctzll(__ballot(lb == mn));
        grp = grp * FAN + j;
      }
      eval_leaf(grp);
      if (exists && !settled) d2[s * (int64_t)R + r] = __float_as_uint(best);
      // upper bound of the tile among the samples that still matter
      const uint32_t fb = lane < acc.n_faces ? acc.face_bits[my_slot] : 0xffffffffu;
      uint32_t thr = 0xffffffffu, um = wave_or_u32(mb);
      while (um) {
        const int f = __builtin_ctz(um);
        um &= um - 1u;
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)fb, f);
        if ((mb >> f) & 1u) thr = v < thr ? v : thr;
      }
      const bool lv_ = !settled && mb != 0u && __float_as_uint(best) > thr;
      const uint32_t key = wave_max_u32(lv_ ? __float_as_uint(best) : 0u);
      if (lane == 0 && key != 0u) {
        const unsigned long long old = atomicMax(&top[s], ((unsigned long long)key << 32) | (unsigned long long)(uint32_t)item);
        if (old == 0ull) top_list[atomicAdd(top_count, 1)] = (int)s;  // first tile of this simplex that matters
      }
      continue;
    }

    // ---- exact traversal in FOCUS ROUNDS.  A round settles only the live samples whose bound is within
    // FOCUS of the largest one (squared distances): its pruning radius, its box and its per-leaf tests look at
    // them alone - a few nearby points instead of a 64-sample tile - while every leaf it evaluates still tightens
    // the bounds of all lanes.  The settled values are delivered, face_bits rises, and most of the remaining
    // samples drop out without a search of their own; whatever is still live forms the next round.
    bool first_round = true;
    bool aborted = false;
    bool on_budget = budgeted;
    int evals = 0;  // leaves evaluated for this tile (all rounds)
#if FLOODER_FINISH_PRIO_AFTER > 0
    int evals_all = 0;
    __builtin_amdgcn_s_setprio(0);
#endif
    for (;;) {
      // (TEAM: every wave of the workgroup computes the same focus set from the same minima and face maxima.  An
      // entry of the top pass asks for one sample only: its first round takes the lanes recorded with the entry.)
      bool focus = (TEAM && stop_after && first_round) ? (live && ((fmask >> lane) & 1ull) != 0ull)
                                                       : (live && best >= focus_frac * M);
      first_round = false;
      float Mf;
      auto rebound = [&]() {
        Mf = wave_max_f32(focus ? best : -1.f);
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          tlo[k] = wave_min_f32(focus ? p[k] : __builtin_inff());
          thi[k] = wave_max_f32(focus ? p[k] : -__builtin_inff());
        }
      };
      rebound();
      ++n_rounds;
      FIN_PHASE(0);  // item setup / round setup (loads, face maxima, live set, boxes)
      int lvl = topl;
      float lb0 = child_bounds(topl, 0);
      int64_t grp0 = 0;
      ++n_node_test;
      if (topl > 0) {
        s_lb[wv][topl][lane] = lb0;
        if (lane == 0) s_grp[wv][topl] = 0;
      }
      int since = 0;
      for (;;) {
        if (TEAM && !(Mf >= 0.f)) break;  // (nothing of the focus set is live any more)
        if (lvl > 0) {
          const float lbv = s_lb[wv][lvl][lane];
          const float mn = wave_min_f32(lbv);
          if (!(mn * SAFE < Mf)) {  // nothing left at this level can improve a focus sample
            if (++lvl > topl) break;
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
            // transposed refine: the 64 leaf boxes of the group (one per lane) against every FOCUS sample
            const bool cand = lb * SAFE < Mf;
            unsigned long long lm = __ballot(focus && mine);
            constexpr int PER_LEAF = 4 * DIM + 40, PER_SAMPLE = 4 * DIM + 3;
            if ((int64_t)__popcll(__ballot(cand)) * PER_LEAF * 100 > (int64_t)__popcll(lm) * PER_SAMPLE * refine_pct) {
              bool need = false;
              while (lm) {  // (wave-uniform)
                const int src = __builtin_ctzll(lm);
                lm &= lm - 1ull;
                float lbp = 0.f;
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                  const float pk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[k]), src));
                  const float gap = __builtin_fmaxf(__builtin_fmaxf(c_lo[k] - pk, pk - c_hi[k]), 0.f);
                  lbp = __builtin_fmaf(gap, gap, lbp);
                }
                const float bi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), src));
                need = need || (lbp * SAFE < bi);
              }
              if (!(cand && need)) lb0 = __builtin_inff();
            }
          }
          FIN_PHASE(1);  // node expansion (+ refine at the leaf level)
          continue;
        }
        // ---- leaf level: nearest unvisited leaf of the current group
        const float mn = wave_min_f32(lb0);
        if (!(mn * SAFE < Mf)) {
          if (++lvl > topl) break;
          continue;
        }
        const int j = __builtin_ctzll(__ballot(lb0 == mn));
        if (lane == j) lb0 = __builtin_inff();  // visited
        const int64_t c = grp0 * FAN + j;
        ++n_leaf_test;
        float blo[DIM], bhi[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          blo[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_lo[k]), j));
          bhi[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_hi[k]), j));
        }
        float lbp = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float gap = __builtin_fmaxf(__builtin_fmaxf(blo[k] - p[k], p[k] - bhi[k]), 0.f);
          lbp = __builtin_fmaf(gap, gap, lbp);
        }
        if (__ballot(focus && (lbp * SAFE < best)) == 0ull) { FIN_PHASE(2); continue; }
        FIN_PHASE(2);  // leaf selection + test
        const float best_before = best;
        eval_leaf(c);
        FIN_PHASE(3);  // leaf evaluation
#if FLOODER_FINISH_PRIO_AFTER > 0
        // a search that has outlasted the median one by far is the tail of the pass in the making: its wave takes
        // precedence over the three it shares the SIMD with (issue is arbitrated by priority, then age)
        if (!TEAM && ++evals_all == FLOODER_FINISH_PRIO_AFTER) __builtin_amdgcn_s_setprio(FLOODER_FINISH_PRIO);
#endif
        if (!TEAM && on_budget && ++evals > budget_eff) {
          // a hard round: hand the tile to the next launch (unless its list is full: then carry on alone)
          const unsigned long long fm = __ballot(focus && mine);
          int h = 0;
          if (lane == 0) h = atomicAdd(hl.cnt_out, 1);
          h = wave_uniform(h);
          if (h < hl.cap) {
            if (mine && !settled && !done) d2[s * (int64_t)R + r] = __float_as_uint(best);  // bounds reached so far
            if (lane == 0) {
              hl.ent_out[h] = (unsigned long long)(uint32_t)item | ((unsigned long long)sub << 32) |
                              ((unsigned long long)subs_i << 40) | (mode == 1 ? (1ull << 48) : 0ull);
              hl.mask_out[h] = fm;
            }
            aborted = true;
            break;
          }
          on_budget = false;
        }
        if (!TEAM && ++since >= refresh_every) {  // (TEAM: no workgroup barrier inside a search)
          since = 0;
          refresh();  // other waves may have raised the face maxima meanwhile: focus samples may drop out
          focus = focus && live;
          rebound();
        } else if (__ballot(focus && best != best_before) != 0ull) {
          // (the pruning radius is the largest bound of the focus samples: it moves only when one of THEIR bounds did -
          // in the far field a leaf in ten; the reduction is a chain of six dependent cross-lane steps)
          Mf = wave_max_f32(focus ? best : -1.f);
        }
        FIN_PHASE(4);  // bounds / live set upkeep
        if (!(Mf >= 0.f)) break;
      }
      if (aborted) break;
      if constexpr (TEAM) {
        // ---- minima of the workgroup's waves (each searched its share of the tree) combined in LDS
        __syncthreads();
        if (wv == 0) s_tbest[lane] = __float_as_uint(best);
        __syncthreads();
        if (wv != 0) atomicMin(&s_tbest[lane], __float_as_uint(best));
        __syncthreads();
        best = __uint_as_float(s_tbest[lane]);
        // a focus sample that some wave dropped on the way is below its faces' maxima: whatever is delivered for
        // it changes nothing; every other focus sample has been searched by all sixteen and is exact
        focus = mine && !done && !settled && focus;
        if (wv == 0 && focus) d2[s * (int64_t)R + r] = __float_as_uint(best) | SETTLED_BIT;
      }
      // ---- deliver the round: focus samples that stayed live to its end are exact (a focus sample that dropped out
      // on the way is below the running maximum of each of its faces: its atomic changes nothing)
      {
        const uint32_t mbm = (mine && focus) ? mb : 0u;
        uint32_t um = wave_or_u32(mbm);
        while (um) {
          const int f = __builtin_ctz(um);
          um &= um - 1u;
          const uint32_t v = wave_max_u32(((mbm >> f) & 1u) ? __float_as_uint(best) : 0u);
          // (no atomic for a value that cannot raise the maximum: most deliveries of a shared face are such)
          const uint32_t seen = (uint32_t)__builtin_amdgcn_readlane((int)fb_seen, f);
          if (lane == 0 && v > seen && (!TEAM || wv == 0)) atomicMax(&acc.face_bits[acc.slot_of(s, f)], v);
        }
      }
      done = done || focus;
      if (stop_after) break;  // (an entry of the top pass: one sample)
      refresh();
      FIN_PHASE(4);
      if (!(M >= 0.f)) break;
    }
    if (mode == 1 && !aborted && lane == pick && exists) d2[s * (int64_t)R + r] = __float_as_uint(best) | SETTLED_BIT;
  }
#ifdef FLOODER_WAVE_END_FIN
  // diagnostic build (tools/wave_ends.py finish): per-wave start, end and longest item of the last pass
  if (stats && mode == 2 && lane == 0 && wave_id < 4096) {
    stats[64 + 16384 + 3 * wave_id] = t_wave0;
    stats[64 + 16384 + 3 * wave_id + 1] = __builtin_amdgcn_s_memrealtime();
    stats[64 + 16384 + 3 * wave_id + 2] = t_longest;
    stats[64 + 16384 + 3 * 4096 + wave_id] = c_longest;
  }
#endif
  if (stats && lane == 0 && (n_node_test | n_leaf_test | n_dropped) != 0ull) {
    atomicAdd(&stats[0], n_leaf_eval);
    atomicAdd(&stats[1], n_leaf_test);
    atomicAdd(&stats[2], n_node_test);
    if (mode == 2) {
      atomicAdd(&stats[4], n_dropped);
      atomicAdd(&stats[5], n_live0);
    }
    atomicAdd(&stats[6], n_rounds);
#ifdef FLOODER_PHASE_TIMERS
    if (mode == 2)
      for (int i = 0; i < 6; ++i) atomicAdd(&stats[40 + i], tf[i]);  // diagnostic build only (tools/bvh_phase.py)
#endif
  }
}

// Counting sort of the flagged tiles by the top 12 bits of their probe bound, largest first (sweep: keys and the
// histogram; here: every block scans the histogram for itself, then scatters its share of the list).  The bound
// predicts the length of a tile's search well (rank correlation 0.8 on the torus of cfg 3); with the long searches
// started first the last pass ends when the work does, not when the last long tile that happened to be queued
// late does (cfg 3: -27 % of the pass, cfg 5: -29 %, simulated from measured tile times).
constexpr int KEY_BUCKETS = 4096;
__global__ __launch_bounds__(256) void order_flags_kernel(const int32_t* __restrict__ flag_list,
                                                          const uint32_t* __restrict__ flag_key,
                                                          const int32_t* __restrict__ flag_count,
                                                          const int32_t* __restrict__ hist,
                                                          int32_t* __restrict__ cursor,
                                                          int32_t* __restrict__ flag_sorted) {
  __shared__ int s_base[KEY_BUCKETS];  // start of the bucket in the sorted list, then of this block's share of it
  __shared__ int s_cnt[KEY_BUCKETS];   // this block's entries per bucket, then its running fill
  __shared__ int s_part[256];
  const int n = flag_count[0];
  if (n <= SHORT_LIST) return;
  // descending exclusive scan: thread t owns buckets [4095 - 16 t - 15, 4095 - 16 t]
  constexpr int PER = KEY_BUCKETS / 256;
  int own[PER], sum = 0;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    own[u] = hist[KEY_BUCKETS - 1 - (threadIdx.x * PER + u)];
    sum += own[u];
    s_cnt[threadIdx.x * PER + u] = 0;
  }
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = s_part[threadIdx.x] - sum;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    s_base[KEY_BUCKETS - 1 - (threadIdx.x * PER + u)] = run;
    run += own[u];
  }
  __syncthreads();
  // the block's contiguous share of the list: count per bucket in LDS, ONE global atomic per bucket the block has
  // entries in (the bounds crowd into a few dozen buckets: an atomic per entry would queue up on those words),
  // then scatter
  const int per_block = (n + (int)gridDim.x - 1) / (int)gridDim.x;
  const int i0 = blockIdx.x * per_block;
  const int i1 = i0 + per_block < n ? i0 + per_block : n;
  for (int i = i0 + threadIdx.x; i < i1; i += 256) atomicAdd(&s_cnt[flag_key[i] >> 19], 1);
  __syncthreads();
  for (int b = threadIdx.x; b < KEY_BUCKETS; b += 256) {
    const int c = s_cnt[b];
    if (c > 0) s_base[b] += atomicAdd(&cursor[b], c);
    s_cnt[b] = 0;
  }
  __syncthreads();
  for (int i = i0 + threadIdx.x; i < i1; i += 256) {
    const int b = (int)(flag_key[i] >> 19);
    flag_sorted[s_base[b] + atomicAdd(&s_cnt[b], 1)] = flag_list[i];
  }
}

__global__ void face_values_kernel(const uint32_t* __restrict__ bits, int64_t n, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = __builtin_sqrtf(__uint_as_float(bits[i]));
}

template <int DIM>
struct FinishOp {
  static int run(const float* pts, const float* nodes, const Levels& lv, const float* verts, const float* weights,
                 int k1, int R, int64_t ns, const int32_t* flag_list, const int32_t* flag_count,
                 const uint32_t* flag_key, int32_t* flag_hist, int32_t* flag_sorted, int32_t* ctl,
                 uint32_t* d2, FaceAcc acc, unsigned long long* top, int32_t* top_list, int probed,
                 unsigned long long* hard, int hard_cap, unsigned long long* stats, hipStream_t st) {
    // (the finish follows the cell sweep, which exists in 2D and 3D: the other dimensions are not instantiated -
    // they were 91 KB of LDS and spilled registers for kernels nobody launches)
    if constexpr (DIM != 2 && DIM != 3) {
      return fail(FLOODER_E_ARG, "flooder_finish_faces_f32: only dim 2 and 3");
    } else {
    const int grid = g_bvh_grid;
    const bool wide = g_finish_wide_points > 0 && lv.count[0] * (int64_t)FLOODER_BVH_LEAF >= (int64_t)g_finish_wide_points;
    // ctl[0..2]: work-queue heads of the three passes, ctl[3]: simplices with a top tile (filled by the probe -
    // the cell sweep's when `probed`, else pass 0 here); the hard-entry launches' words: below
    const bool hard_on = hard != nullptr && hard_cap > 0 && g_finish_budget > 0;
    const bool ordered = flag_key != nullptr && flag_hist != nullptr && flag_sorted != nullptr && g_finish_order != 0;
    if (ordered)
      hipLaunchKernelGGL(order_flags_kernel, dim3(256), dim3(256), 0, st, flag_list, flag_key, flag_count, flag_hist,
                         flag_hist + KEY_BUCKETS, flag_sorted);
    const int64_t words = (int64_t)hard_cap * 2;  // entries, masks (u64)
    auto list = [&](int which, bool in, int32_t* cnt, HardLists& hl) {
      unsigned long long* base = hard + which * words;
      if (in) {
        hl.ent_in = base;
        hl.mask_in = base + hard_cap;
        hl.cnt_in = cnt;
      } else {
        hl.ent_out = base;
        hl.mask_out = base + hard_cap;
        hl.cnt_out = cnt;
      }
    };
    auto launch = [&](int mode, int32_t* queue, const HardLists& hl) {
      if (mode == 3)  // one workgroup of 16 waves per hard tile, one workgroup per CU
        hipLaunchKernelGGL((finish_faces_kernel<DIM, true>), dim3(g_bvh_grid / 4), dim3(64 * TEAM_WAVES), 0, st, pts,
                           nodes, lv, verts, weights, k1, R, ns, flag_list, ordered ? flag_sorted : nullptr, flag_count,
                           mode, g_bvh_subs, g_finish_items_cap, g_bvh_refine_pct, (float)g_finish_focus_pct * 0.01f,
                           g_finish_refresh, queue, d2, acc, top, top_list, ctl + 3, hl, stats);
      else if (wide)   // (deep tree: eight waves per workgroup, three workgroups per CU)
        hipLaunchKernelGGL((finish_faces_kernel<DIM, false, 8>), dim3(grid * 3 / 4), dim3(512), 0, st, pts, nodes, lv, verts,
                           weights, k1, R, ns, flag_list, ordered ? flag_sorted : nullptr, flag_count, mode, g_bvh_subs,
                           g_finish_items_cap, g_bvh_refine_pct, (float)g_finish_focus_pct * 0.01f, g_finish_refresh,
                           queue, d2, acc, top, top_list, ctl + 3, hl, stats);
      else
        hipLaunchKernelGGL((finish_faces_kernel<DIM, false>), dim3(grid), dim3(256), 0, st, pts, nodes, lv, verts,
                           weights, k1, R, ns, flag_list, ordered ? flag_sorted : nullptr, flag_count, mode, g_bvh_subs,
                           g_finish_items_cap, g_bvh_refine_pct, (float)g_finish_focus_pct * 0.01f, g_finish_refresh,
                           queue, d2, acc, top, top_list, ctl + 3, hl, stats);
    };
    const HardLists none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, hard_cap, 0};
    // (the per-wave passes pop from sharded heads behind the 24 control words; the team passes keep one word)
    int32_t* q0 = ctl + 24;
    int32_t* q1 = ctl + 24 + FLOODER_QUEUE_WORDS;
    int32_t* q2 = ctl + 24 + 2 * FLOODER_QUEUE_WORDS;
    if (!probed) launch(0, q0, none);
    if (!hard_on) {
      if (g_finish_top) launch(1, q1, none);
      launch(2, q2, none);
      return check_launch("finish_faces");
    }
    // ctl: [0..2] queue heads of the probe / top / rest passes, [3] simplices with a top tile, [4], [5] queue head
    // and list length of the top pass's hard entries, [6], [7] those of the rest pass's
    HardLists a = none, b = none, c = none, d = none;
    a.budget = c.budget = g_finish_budget;
    a.budget_min = c.budget_min = g_finish_budget_min;
    if (g_finish_top) {
      list(0, false, ctl + 5, a);  // top pass: hard entries -> list 0
      launch(1, q1, a);
      list(0, true, ctl + 5, b);   // ... one workgroup each (one sample per entry: one round)
      launch(3, ctl + 4, b);
    }
    list(1, false, ctl + 7, c);  // the other samples: hard tiles -> list 1
    launch(2, q2, c);
    list(1, true, ctl + 7, d);   // ... one workgroup each, all their rounds
    launch(3, ctl + 6, d);
    return check_launch("finish_faces");
    }
  }
};

}  // namespace

extern "C" {

int flooder_finish_faces_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                             const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                             const int32_t* flag_list, const int32_t* flag_count, const uint32_t* flag_key,
                             int32_t* flag_hist, int32_t* flag_sorted, int32_t* ctl,
                             uint64_t* top, int32_t* top_list, int probed, uint32_t* d2_scratch,
                             const uint32_t* memb, int n_faces, uint32_t* face_bits, const int32_t* face_slot,
                             uint64_t* hard_scratch, int hard_cap, uint64_t* stats, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!pts_sorted || !nodes || !verts || !weights || !flag_list || !flag_count || !ctl || !top || !top_list || !d2_scratch ||
      !memb || !face_bits || n_pts < 1 || k1 < 1 || k1 > FLOODER_MAX_VERTS || R < 1 || n_faces < 1 || n_faces > 32 ||
      hard_cap < 0)
    return fail(FLOODER_E_ARG, "flooder_finish_faces_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<FinishOp>(dim, pts_sorted, nodes, lv, verts, weights, k1, R, n_simplices, flag_list,
                                flag_count, flag_key, flag_hist, flag_sorted, ctl, d2_scratch, FaceAcc{memb, face_bits, n_faces, nullptr, nullptr, nullptr, face_slot},
                                reinterpret_cast<unsigned long long*>(top), top_list, probed,
                                reinterpret_cast<unsigned long long*>(hard_scratch), hard_cap,
                                reinterpret_cast<unsigned long long*>(stats), (hipStream_t)stream);
}

int flooder_face_values_f32(const uint32_t* face_bits, int64_t n, float* out_face, void* stream) {
  if (n == 0) return FLOODER_OK;
  if (!face_bits || !out_face || n < 0) return fail(FLOODER_E_ARG, "flooder_face_values_f32: bad argument");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(face_values_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, face_bits, n, out_face);
  return check_launch("face_values");
}

}  // extern "C"

// ------------------------------------------------------------------------------------ simplex weights
// Rough number of cloud points inside each simplex's bounding box: the box tree is walked from the top to level 1
// (1024 points per node); every node overlapping the box contributes its point count times the overlapped
// fraction of its own box.  Heavy simplices (dense regions) take 100x longer than the median one in the sweeps; the
// host queues the simplices by descending weight so that the long ones start first and the short ones fill the tail.
namespace {

constexpr int WFRONT = 256;

// The same launch PREPARES a fused sweep (flooder_simplex_prepare_f32): every thread clears its share of `zero_buf` (the
// control words, queue heads and face words the three launches of the sweep start from - torch.zeros was a launch of
// its own), and the blocks behind the first `weight_blocks` write the face-plane rows (flood_planes.hpp - a launch of
// its own in front of the witness / cell sweep).  Three launches in one: ~8 us of cfg 2's step.
template <int DIM>
__global__ __launch_bounds__(256) void simplex_weight_kernel(const float* __restrict__ nodes, Levels lv,
                                                             const float* __restrict__ verts, int k1,
                                                             int64_t n_simplices, float* __restrict__ weight,
                                                             int weight_blocks, float* __restrict__ plane_tab,
                                                             int32_t* __restrict__ zero_buf, int64_t zero_words) {
  constexpr int DP = padded_dim(DIM);
  __shared__ int s_front[4][2][WFRONT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int topl = lv.n_levels - 1;
  const int stop = topl >= 1 ? 1 : 0;
  if (zero_buf != nullptr)
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < zero_words; j += (int64_t)gridDim.x * 256) zero_buf[j] = 0;
  if ((int)blockIdx.x >= weight_blocks) {   // (plane rows: one simplex per thread)
    if constexpr (DIM == 2 || DIM == 3) {
      const int64_t s = (int64_t)((int)blockIdx.x - weight_blocks) * 256 + threadIdx.x;
      if (s < n_simplices) simplex_planes_row<DIM>(verts, k1, s, plane_tab);
    }
    return;
  }
  for (int64_t s = (int64_t)blockIdx.x * 4 + wv; s < n_simplices; s += (int64_t)weight_blocks * 4) {
    float blo[DIM], bhi[DIM];
    const float* vs = verts + s * (int64_t)k1 * DIM;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      float mn = vs[k], mx = vs[k];
      for (int j = 1; j < k1; ++j) {
        mn = __builtin_fminf(mn, vs[j * DIM + k]);
        mx = __builtin_fmaxf(mx, vs[j * DIM + k]);
      }
      blo[k] = mn;
      bhi[k] = mx;
    }
    float acc = 0.f;  // per lane
    int* fa = s_front[wv][0];
    int* fb = s_front[wv][1];
    int na = 1;
    if (lane == 0) fa[0] = 0;  // pseudo-parent of the top level: group 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int lvl = topl; lvl >= stop; --lvl) {
      float per_node = (float)LEAF;
      for (int l = 0; l < lvl; ++l) per_node *= (float)FAN;
      int nb = 0;
      for (int f = 0; f < na; ++f) {
        const int grp = wave_uniform(fa[f]);
        const int64_t idx = (int64_t)grp * FAN + lane;
        bool hit = idx < lv.count[lvl];
        float frac = 0.f;
        if (hit) {
          float lo[DP], hi[DP];
          const float* nbp = nodes + (lv.off[lvl] + idx) * 2 * DP;
          load_row<DP>(nbp, lo);
          load_row<DP>(nbp + DP, hi);
          frac = 1.f;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            const float ov = __builtin_fminf(hi[k], bhi[k]) - __builtin_fmaxf(lo[k], blo[k]);
            const float ext = hi[k] - lo[k];
            hit = hit && ov >= 0.f;
            frac *= ext > 0.f ? __builtin_fminf(__builtin_fmaxf(ov, 0.f) / ext, 1.f) : 1.f;
          }
        }
        const unsigned long long m = __ballot(hit);
        const int cnt = __popcll(m);
        if (lvl > stop && nb + cnt <= WFRONT) {
          if (hit) fb[nb + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = (int)idx;
          nb += cnt;
        } else if (hit) {
          acc += per_node * frac;  // last level, or no room to descend: estimate at this level
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      int* t = fa; fa = fb; fb = t;
      na = nb;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) weight[s] = acc;
  }
}

template <int DIM>
struct WeightOp {
  static int run(const float* nodes, const Levels& lv, const float* verts, int k1, int64_t ns, float* weight,
                 float* plane_tab, int32_t* zero_buf, int64_t zero_words, hipStream_t st) {
    int64_t blocks = (ns + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    int64_t plane_blocks = 0;
    if constexpr (DIM == 2 || DIM == 3) plane_blocks = plane_tab ? (ns + 255) / 256 : 0;
    else if (plane_tab) return fail(FLOODER_E_ARG, "flooder_simplex_prepare_f32: plane rows only in dim 2 and 3");
    hipLaunchKernelGGL((simplex_weight_kernel<DIM>), dim3((int)(blocks + plane_blocks)), dim3(256), 0, st, nodes, lv, verts,
                       k1, ns, weight, (int)blocks, plane_tab, zero_buf, zero_words);
    if (plane_blocks) planes_done_for(verts, plane_tab, ns, st);   // (the sweep's entries need not launch theirs)
    return check_launch("simplex_weight");
  }
};

}  // namespace

extern "C" int flooder_simplex_weight_f32(const float* nodes, int64_t n_pts, int dim, const float* verts, int k1,
                                          int64_t n_simplices, float* weight, void* stream) {
  if (n_simplices == 0) return FLOODER_OK;
  if (!nodes || !verts || !weight || n_pts < 1 || k1 < 1 || k1 > FLOODER_MAX_VERTS)
    return fail(FLOODER_E_ARG, "flooder_simplex_weight_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<WeightOp>(dim, nodes, lv, verts, k1, n_simplices, weight, (float*)nullptr, (int32_t*)nullptr,
                                (int64_t)0, (hipStream_t)stream);
}

extern "C" void flooder_simplex_planes_forget(void) { (void)planes_are_done(nullptr, nullptr, 0, nullptr); }

extern "C" int flooder_simplex_prepare_f32(const float* nodes, int64_t n_pts, int dim, const float* verts, int k1,
                                           int64_t n_simplices, float* weight, float* plane_scratch, int32_t* zero_buf,
                                           int64_t zero_words, void* stream) {
  if (n_simplices == 0 && zero_words == 0) return FLOODER_OK;
  if (!nodes || !verts || !weight || n_pts < 1 || k1 < 1 || k1 > FLOODER_MAX_VERTS || n_simplices < 1 || zero_words < 0 ||
      (zero_words > 0 && !zero_buf))
    return fail(FLOODER_E_ARG, "flooder_simplex_prepare_f32: bad argument");
  const Levels lv = make_levels(n_pts);
  return dispatch_dim<WeightOp>(dim, nodes, lv, verts, k1, n_simplices, weight, plane_scratch, zero_buf, zero_words,
                                (hipStream_t)stream);
}
