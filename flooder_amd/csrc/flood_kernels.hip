// flood_kernels.hip - CDNA4 (gfx950) kernels + C ABI of the Flood-complex coverage sweep.
//
// Path (reference plus-rkwitt/flooder): flooder/core.py:200-226 + 251-276 and the two Triton kernels
// in flooder/triton_kernels.py.  The reference materialises a (B, m+512) bool mask, runs
// torch.nonzero over it, casts the indices and then re-gathers every 512-candidate tile from 310
// programs with a float atomic_min per (tile, sample).  Here:
//
//   ball_scan<COUNT|FILL>  one streaming pass over each simplex's slab of the axis-sorted cloud:
//                          count the points inside the bounding ball, then (second pass) compact
//                          their coordinates into a per-simplex candidate list (padded rows,
//                          16 B per candidate in 3D).  HBM-bound, coalesced 16 B/lane loads.
//   sweep                  one WAVE owns (simplex, 512-sample tile, <=2048-candidate chunk): every
//                          lane keeps 8 samples in registers (computed from the simplex vertices
//                          and the barycentric weights, never materialised in HBM), candidates are
//                          wave-uniform and stream through the SCALAR cache into SGPRs (no LDS, no
//                          barrier), the running minimum stays in registers; one integer atomic min
//                          per (sample, chunk) at the end.  VALU-bound (fp32, no MFMA: K = dim = 3).
//   face_max               per-face maxima + sqrt (core.py:251-276).
//   fps_step               farthest-point sampling iteration (generate_landmarks, core.py:337-343).
//
// Work is pulled from device-side queues (atomic head) so heavy-tailed candidate counts balance.

#include "flood_common.hpp"

namespace flooder {

thread_local char g_err[256] = "";
char* err_buf() { return g_err; }

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return FLOODER_E_LAUNCH;
  }
  return FLOODER_OK;
}

}  // namespace flooder

using namespace flooder;

namespace {

constexpr int SCAN_PARTS = 8;        // each simplex's slab is scanned by 8 independent work units
constexpr int SCAN_THREADS = 256;
constexpr int SWEEP_THREADS = 256;   // 4 independent waves per block
constexpr int KS = 8;                // samples per lane  -> 512 samples per wave tile
constexpr int CHUNK = FLOODER_SWEEP_CHUNK;
static_assert(64 * KS == FLOODER_TILE_SAMPLES, "tile size");

int g_sweep_variant = 0;  // 0 = packed fp32 (v_pk_*), 1 = plain fp32; flooder_set_option("sweep_variant")

// ---------------------------------------------------------------------------------- ball scan
// Work unit g = simplex * SCAN_PARTS + part.  Blocks stride over the units (uniform cost per point).
template <int DIM, bool FILL>
__global__ __launch_bounds__(SCAN_THREADS) void ball_scan_kernel(
    const float* __restrict__ pts, int ld, const float* __restrict__ centers,
    const float* __restrict__ radii, const int64_t* __restrict__ slab_lo,
    const int64_t* __restrict__ slab_hi, int64_t n_simplices, int32_t* __restrict__ counts_out,
    const int32_t* __restrict__ counts_in, const int64_t* __restrict__ cand_off,
    int32_t* __restrict__ cursor, float* __restrict__ cand) {
  constexpr int DP = padded_dim(DIM);
  const int lane = threadIdx.x & 63;
  const int64_t n_units = n_simplices * SCAN_PARTS;
  for (int64_t g = blockIdx.x; g < n_units; g += gridDim.x) {
    const int64_t s = g / SCAN_PARTS;
    const int part = (int)(g - s * SCAN_PARTS);
    const int64_t lo = slab_lo[s], hi = slab_hi[s];
    const int64_t len = hi - lo;
    if (len <= 0) {
      if (FILL && part == 0 && threadIdx.x == 0) { /* nothing to pad: count is 0 */ }
      continue;
    }
    const int64_t per = (len + SCAN_PARTS - 1) / SCAN_PARTS;
    const int64_t b = lo + per * part;
    const int64_t e = (b + per < hi) ? b + per : hi;
    float c[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) c[k] = centers[s * DIM + k];
    const float r = radii[s];
    const float r2 = r * r;
    const int64_t off = FILL ? cand_off[s] : 0;
    int wave_count = 0;
    for (int64_t j0 = b; j0 < e; j0 += SCAN_THREADS) {
      const int64_t j = j0 + threadIdx.x;
      bool inside = false;
      float x[DP];
      if (j < e) {
        if (ld == DP) {
          load_row<DP>(pts + j * DP, x);
        } else {
#pragma unroll
          for (int k = 0; k < DIM; ++k) x[k] = pts[j * ld + k];
        }
        float d2 = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float diff = x[k] - c[k];
          d2 += diff * diff;  // same accumulation order as triton_kernels.py:138-146
        }
        inside = d2 <= r2;
      }
      const unsigned long long m = __ballot(inside);
      if (m == 0ull) continue;
      const int n_in = __popcll(m);
      if constexpr (!FILL) {
        wave_count += n_in;
      } else {
        int base = 0;
        if (lane == 0) base = atomicAdd(&cursor[s], n_in);
        base = wave_uniform(base);
        if (inside) {
          const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
          float* dst = cand + (off + base + rank) * DP;
#pragma unroll
          for (int k = 0; k < DP; ++k) dst[k] = (k < DIM) ? x[k] : 0.f;
        }
      }
    }
    if constexpr (!FILL) {
      if (lane == 0 && wave_count) atomicAdd(&counts_out[s], wave_count);
    } else {
      // pad rows [count, next offset) with +inf so that full groups of 8 can be read blindly
      if (part == 0) {
        const int cnt = counts_in[s];
        const int64_t pad_end = cand_off[s + 1] - off;
        for (int64_t q = cnt + threadIdx.x; q < pad_end; q += SCAN_THREADS) {
          float* dst = cand + (off + q) * DP;
#pragma unroll
          for (int k = 0; k < DP; ++k) dst[k] = __builtin_inff();
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------- sweep
__device__ __forceinline__ int64_t upper_bound_minus1(const int64_t* __restrict__ prefix, int64_t n,
                                                      int64_t g) {
  // largest s in [0, n) with prefix[s] <= g   (prefix[0] = 0, prefix[n] = total > g)
  int64_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (prefix[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// PACKED: two samples share one 64-bit register pair and every sub/mul/fma is a v_pk_*_f32
// (2 lanes-worth of fp32 per issue slot); otherwise plain v_sub/v_fma.  Same arithmetic, same bits.
template <int DIM, bool PACKED>
__global__ __launch_bounds__(SWEEP_THREADS) void sweep_kernel(
    const float* __restrict__ cand, const int64_t* __restrict__ cand_off,
    const int32_t* __restrict__ counts, const float* __restrict__ verts,
    const float* __restrict__ weights, int k1, int R, int64_t n_simplices,
    const int64_t* __restrict__ item_prefix, int32_t* __restrict__ queue,
    uint32_t* __restrict__ out_d2) {
  constexpr int DP = padded_dim(DIM);
  const int lane = threadIdx.x & 63;
  const int tiles = (R + 64 * KS - 1) / (64 * KS);
  const int64_t n_items = item_prefix[n_simplices];

  for (;;) {
    int g32 = 0;
    if (lane == 0) g32 = atomicAdd(queue, 1);
    const int64_t g = (int64_t)wave_uniform(g32);
    if (g >= n_items) break;
    const int64_t s = wave_uniform64(upper_bound_minus1(item_prefix, n_simplices, g));
    const int local = (int)(g - item_prefix[s]);
    const int tile = local % tiles;
    const int chunk = local / tiles;
    const int cnt = counts[s];
    const int c_begin = chunk * CHUNK;
    int c_end = c_begin + CHUNK;
    if (c_end > cnt) c_end = cnt;
    // lists are padded to a multiple of 8 rows with +inf rows, which never win
    const int n_groups = (c_end - c_begin + 7) >> 3;

    // ---- this lane's KS samples: p = sum_j w[r,j] * v[s,j,:]   (core.py:188)
    float p[KS][DIM];
    const float* vs = verts + s * (int64_t)k1 * DIM;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      int r = tile * 64 * KS + i * 64 + lane;
      if (r >= R) r = R - 1;  // duplicate of the last sample, never stored
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[i][k] = 0.f;
      for (int j = 0; j < k1; ++j) {
        const float w = weights[(int64_t)r * k1 + j];
#pragma unroll
        for (int k = 0; k < DIM; ++k) p[i][k] = __builtin_fmaf(w, vs[j * DIM + k], p[i][k]);
      }
    }

    float best[KS];
    const float* cp = cand + (cand_off[s] + c_begin) * DP;

    if constexpr (PACKED) {
      v2f P[KS / 2][DIM];
      v2f B[KS / 2];
#pragma unroll
      for (int i = 0; i < KS / 2; ++i) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) P[i][k] = v2f{p[2 * i][k], p[2 * i + 1][k]};
        B[i] = v2f{__builtin_inff(), __builtin_inff()};
      }
      for (int gi = 0; gi < n_groups; ++gi) {
        typename RowVec<DP>::type c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = load_uniform_row<DP>(cp + u * DP);
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
#pragma unroll
          for (int i = 0; i < KS / 2; ++i) {
            v2f da, db;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const v2f ta = P[i][k] - v2f{c[u][k], c[u][k]};
              const v2f tb = P[i][k] - v2f{c[u + 1][k], c[u + 1][k]};
              if (k == 0) {
                da = ta * ta;
                db = tb * tb;
              } else {
                da = __builtin_elementwise_fma(ta, ta, da);
                db = __builtin_elementwise_fma(tb, tb, db);
              }
            }
            B[i].x = __builtin_fminf(B[i].x, __builtin_fminf(da.x, db.x));
            B[i].y = __builtin_fminf(B[i].y, __builtin_fminf(da.y, db.y));
          }
        }
        cp += 8 * DP;
      }
#pragma unroll
      for (int i = 0; i < KS / 2; ++i) {
        best[2 * i] = B[i].x;
        best[2 * i + 1] = B[i].y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < KS; ++i) best[i] = __builtin_inff();
      for (int gi = 0; gi < n_groups; ++gi) {
        typename RowVec<DP>::type c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = load_uniform_row<DP>(cp + u * DP);
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
#pragma unroll
          for (int i = 0; i < KS; ++i) {
            float da, db;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const float ta = p[i][k] - c[u][k];
              const float tb = p[i][k] - c[u + 1][k];
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
        cp += 8 * DP;
      }
    }

#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int r = tile * 64 * KS + i * 64 + lane;
      if (r < R) atomicMin(&out_d2[s * (int64_t)R + r], __float_as_uint(best[i]));
    }
  }
}

// ---------------------------------------------------------------------------------- face max
__global__ __launch_bounds__(256) void face_max_kernel(const uint32_t* __restrict__ d2, int R,
                                                       const int32_t* __restrict__ face_ptr,
                                                       const int32_t* __restrict__ face_rows,
                                                       int n_faces, float* __restrict__ out_face,
                                                       float* __restrict__ out_dist) {
  const int64_t s = blockIdx.x;
  const uint32_t* row = d2 + s * (int64_t)R;
  __shared__ uint32_t red[4];
  if (out_dist) {
    for (int r = threadIdx.x; r < R; r += blockDim.x)
      out_dist[s * (int64_t)R + r] = __builtin_sqrtf(__uint_as_float(row[r]));
  }
  // Faces with many rows (the simplex itself: all R rows) are reduced by the whole block - a face that lists
  // every row is scanned directly, coalesced, without its index list; the many small faces (triangles, edges,
  // vertices) go one per wave, round robin, with no block barrier.
  constexpr int BIG = 1024;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  for (int f = 0; f < n_faces; ++f) {
    const int b = face_ptr[f], e = face_ptr[f + 1];
    if (e - b < BIG) continue;  // (block-uniform)
    uint32_t m = 0u;
    if (e - b == R && (R & 3) == 0) {  // rows are 16 B aligned: four values per load
      const uint4* row4 = reinterpret_cast<const uint4*>(row);
      for (int r = threadIdx.x; r < (R >> 2); r += blockDim.x) {
        const uint4 v = row4[r];
        const uint32_t a = v.x > v.y ? v.x : v.y, c = v.z > v.w ? v.z : v.w;
        const uint32_t t = a > c ? a : c;
        m = t > m ? t : m;
      }
    } else if (e - b == R) {
      for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const uint32_t v = row[r];
        m = v > m ? v : m;
      }
    } else {
      for (int q = b + threadIdx.x; q < e; q += blockDim.x) {
        const uint32_t v = row[face_rows[q]];
        m = v > m ? v : m;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(m, o);
      m = t > m ? t : m;
    }
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t a = red[0];
      for (int w = 1; w < n_waves; ++w) a = red[w] > a ? red[w] : a;
      out_face[s * (int64_t)n_faces + f] = __builtin_sqrtf(__uint_as_float(a));
    }
    __syncthreads();
  }
  int turn = 0;
  for (int f = 0; f < n_faces; ++f) {
    const int b = face_ptr[f], e = face_ptr[f + 1];
    if (e - b >= BIG) continue;
    if ((turn++ % n_waves) != wave) continue;
    uint32_t m = 0u;
    for (int q = b + lane; q < e; q += 64) {
      const uint32_t v = row[face_rows[q]];
      m = v > m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(m, o);
      m = t > m ? t : m;
    }
    if (lane == 0) out_face[s * (int64_t)n_faces + f] = __builtin_sqrtf(__uint_as_float(m));
  }
}

// Coarse lattices (R <= 64 rows, <= 32 faces: the triangles and edges of a high-dimensional run - a million of them at
// cfg 4): a block per simplex is a million 36-row blocks whose dispatch alone took 1.8 ms, and a wave reduction per face
// is 250 instructions per simplex (0.46 ms).  Here a wave takes U = 64 / F simplices per step (F = the number of faces
// rounded up to a power of two): their rows go through LDS (one coalesced load each), and lane (u, f) runs through the
// row list of face f of simplex u.
__global__ __launch_bounds__(256) void face_max_small_kernel(const uint32_t* __restrict__ d2, int64_t n_simplices, int R,
                                                             const int32_t* __restrict__ face_ptr,
                                                             const int32_t* __restrict__ face_rows, int n_faces, int lgF,
                                                             float* __restrict__ out_face, float* __restrict__ out_dist) {
  // [wave][simplex of the step][row]: 4 x U x 64 words of dynamic LDS - a static array for U = 32 was 32 KB per block,
  // four blocks per CU, when seven faces (U = 8) need 8 KB: the loop is a chain of load latencies and wants the waves
  extern __shared__ uint32_t s_v_dyn[];
  __shared__ uint8_t s_rows[32 * 64];       // the faces' row lists (every face lists at most R <= 64 rows)
  __shared__ int s_ptr[33];
  for (int f = threadIdx.x; f <= n_faces; f += blockDim.x) s_ptr[f] = face_ptr[f];
  __syncthreads();
  const int n_rows = s_ptr[n_faces] < 32 * 64 ? s_ptr[n_faces] : 32 * 64;
  for (int q = threadIdx.x; q < n_rows; q += blockDim.x) s_rows[q] = (uint8_t)face_rows[q];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int U = 64 >> lgF, f = lane & ((1 << lgF) - 1), u_mine = lane >> lgF;
  uint32_t* s_v = s_v_dyn + (size_t)wv * U * 64;   // s_v[u * 64 + row]
  const int qb = f < n_faces ? s_ptr[f] : 0, qe = f < n_faces ? s_ptr[f + 1] : 0;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * U; s0 < n_simplices; s0 += waves * U) {
    for (int u = 0; u < U; ++u) {   // (wave-uniform trip count)
      const int64_t s = s0 + u;
      const uint32_t v = (lane < R && s < n_simplices) ? d2[s * (int64_t)R + lane] : 0u;
      s_v[u * 64 + lane] = v;
      if (out_dist && lane < R && s < n_simplices) out_dist[s * (int64_t)R + lane] = __builtin_sqrtf(__uint_as_float(v));
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t m = 0u;   // (d2 bits of non-negative floats order like the floats)
    for (int q = qb; q < qe; ++q) {
      const uint32_t t = s_v[u_mine * 64 + s_rows[q]];
      m = t > m ? t : m;
    }
    const int64_t s = s0 + u_mine;
    if (f < n_faces && s < n_simplices) out_face[s * (int64_t)n_faces + f] = __builtin_sqrtf(__uint_as_float(m));
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void fill_u32_kernel(uint32_t* __restrict__ buf, int64_t n, uint32_t value) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) buf[i] = value;
}

// ---------------------------------------------------------------------------------- FPS
// Iteration `it` (>= 1): the previously selected point is packed in best[it-1]; update the running
// squared distance of every point, reduce (max distance, then lowest index) into best[it].
// key = (d2 bits << 32) | (0xffffffff - index): unsigned max picks the largest distance and, on a
// tie, the lowest index (numpy argmax order).
template <int DIM>
__global__ __launch_bounds__(256) void fps_step_kernel(const float* __restrict__ pts, int64_t n,
                                                       int ld, int it, float* __restrict__ mind,
                                                       unsigned long long* __restrict__ best,
                                                       int64_t* __restrict__ out_idx) {
  const unsigned long long prev = best[it - 1];
  const int64_t q = (int64_t)(0xffffffffu - (uint32_t)(prev & 0xffffffffu));
  float c[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) c[k] = pts[q * ld + k];
  if (blockIdx.x == 0 && threadIdx.x == 0) out_idx[it - 1] = q;
  unsigned long long key = 0ull;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const float t = pts[j * ld + k] - c[k];
      d2 = __builtin_fmaf(t, t, d2);
    }
    const float old = (it == 1) ? __builtin_inff() : mind[j];
    const float m = d2 < old ? d2 : old;
    mind[j] = m;
    const unsigned long long kj =
        ((unsigned long long)__float_as_uint(m) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)j);
    key = kj > key ? kj : key;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(key, o);
    key = t > key ? t : key;
  }
  __shared__ unsigned long long red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = red[0];
    for (int w = 1; w < 4; ++w) a = red[w] > a ? red[w] : a;
    atomicMax(&best[it], a);
  }
}

// Fast path for dim <= 3: rows (x, y, z, running min d^2) are one 16-byte load, the running minimum one
// 4-byte store; 4 rows per thread, all loads issued before any use; the block owns a fixed 1024-row
// slice, so across the launches of one FPS run every XCD keeps re-reading the same ~2.5 MB from its L2.
template <int DIM>
__global__ __launch_bounds__(256) void fps_rows_init_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                            float4* __restrict__ rows) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    float4 r;
    r.x = pts[j * ld];
    r.y = DIM > 1 ? pts[j * ld + (DIM > 1 ? 1 : 0)] : 0.f;
    r.z = DIM > 2 ? pts[j * ld + (DIM > 2 ? 2 : 0)] : 0.f;
    r.w = __builtin_inff();
    rows[j] = r;
  }
}

constexpr int FPS_SLOTS = 64;  // arg-max slots per iteration (same-address atomics serialise at ~13 ns each)

// winner of iteration `it`: reduce its FPS_SLOTS slots (every wave does this redundantly, one slot per lane)
__device__ __forceinline__ uint32_t fps_winner(const unsigned long long* __restrict__ best, int it) {
  const unsigned long long k = best[(int64_t)it * FPS_SLOTS + (threadIdx.x & 63)];
  const float m = __uint_as_float((uint32_t)(k >> 32));       // d^2 >= 0 (or 0 for an empty slot)
  const uint32_t lowinv = (uint32_t)(k & 0xffffffffu);        // 0xffffffff - index (0 for an empty slot)
  const float wm = wave_max_f32(m);
  const uint32_t idx = wave_min_u32((m == wm && k != 0ull) ? 0xffffffffu - lowinv : 0xffffffffu);
  return idx;
}

__global__ __launch_bounds__(256) void fps_fast_kernel(float4* __restrict__ rows, int64_t n, int it,
                                                       unsigned long long* __restrict__ best,
                                                       int64_t* __restrict__ out_idx) {
  // this thread's rows first: the loads are in flight while the previous winner is reduced and fetched
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  float4 r[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t j = base + u * 256;
    r[u] = rows[j < n ? j : n - 1];
  }
  const uint32_t q = fps_winner(best, it - 1);
  const float4 c = rows[q];
  if (blockIdx.x == 0 && threadIdx.x == 0) out_idx[it - 1] = (int64_t)q;
  float bm = -1.f;
  uint32_t bi = 0xffffffffu;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t j = base + u * 256;
    if (j < n) {
      float t = r[u].x - c.x;
      float d2 = t * t;
      t = r[u].y - c.y;
      d2 = __builtin_fmaf(t, t, d2);
      t = r[u].z - c.z;
      d2 = __builtin_fmaf(t, t, d2);
      const float m = d2 < r[u].w ? d2 : r[u].w;
      if (m < r[u].w) reinterpret_cast<float*>(rows + j)[3] = m;
      if (m > bm) { bm = m; bi = (uint32_t)j; }  // ascending j: the first maximum keeps the lowest index
    }
  }
  // wave: largest distance, then lowest index among the lanes that hold it
  const float wm = wave_max_f32(bm);
  const uint32_t wi = wave_min_u32(bm == wm ? bi : 0xffffffffu);
  __shared__ float s_m[4];
  __shared__ uint32_t s_i[4];
  if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = wm; s_i[threadIdx.x >> 6] = wi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = s_m[0];
    uint32_t i = s_i[0];
    for (int w = 1; w < 4; ++w)
      if (s_m[w] > m || (s_m[w] == m && s_i[w] < i)) { m = s_m[w]; i = s_i[w]; }
    if (m >= 0.f) {
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(m) << 32) | (unsigned long long)(0xffffffffu - i);
      atomicMax(&best[(int64_t)it * FPS_SLOTS + (blockIdx.x % FPS_SLOTS)], key);
    }
  }
}

__global__ void fps_fast_init_kernel(unsigned long long* best, int64_t start) {
  // iteration 0 "winner" = the start point: a positive distance so that the slot counts as filled
  best[0] = ((unsigned long long)__float_as_uint(1.0f) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)start);
}

__global__ void fps_fast_last_kernel(const unsigned long long* best, int n_lms, int64_t* out_idx) {
  const uint32_t q = fps_winner(best, n_lms - 1);
  if (threadIdx.x == 0) out_idx[n_lms - 1] = (int64_t)q;
}

__global__ void fps_init_kernel(unsigned long long* best, int64_t start) {
  best[0] = (unsigned long long)(0xffffffffu - (uint32_t)start);
}

__global__ void fps_last_kernel(const unsigned long long* best, int n_lms, int64_t* out_idx) {
  const unsigned long long prev = best[n_lms - 1];
  out_idx[n_lms - 1] = (int64_t)(0xffffffffu - (uint32_t)(prev & 0xffffffffu));
}

// ---------------------------------------------------------------------------------- dispatch
int scan_grid(int64_t n_simplices) {
  int64_t units = n_simplices * SCAN_PARTS;
  int64_t g = units < 256 * 8 ? units : 256 * 8;
  return (int)(g < 1 ? 1 : g);
}

template <int DIM>
struct CountOp {
  static int run(const float* pts, int ld, const float* centers, const float* radii,
                 const int64_t* lo, const int64_t* hi, int64_t ns, int32_t* counts,
                 hipStream_t st) {
    hipLaunchKernelGGL((ball_scan_kernel<DIM, false>), dim3(scan_grid(ns)), dim3(SCAN_THREADS), 0,
                       st, pts, ld, centers, radii, lo, hi, ns, counts, nullptr, nullptr, nullptr,
                       nullptr);
    return check_launch("ball_count");
  }
};

template <int DIM>
struct FillOp {
  static int run(const float* pts, int ld, const float* centers, const float* radii,
                 const int64_t* lo, const int64_t* hi, int64_t ns, const int32_t* counts,
                 const int64_t* cand_off, int32_t* cursor, float* cand, hipStream_t st) {
    hipLaunchKernelGGL((ball_scan_kernel<DIM, true>), dim3(scan_grid(ns)), dim3(SCAN_THREADS), 0, st,
                       pts, ld, centers, radii, lo, hi, ns, nullptr, counts, cand_off, cursor, cand);
    return check_launch("ball_fill");
  }
};

template <int DIM>
struct SweepOp {
  static int run(const float* cand, const int64_t* cand_off, const int32_t* counts,
                 const float* verts, const float* weights, int k1, int R, int64_t ns,
                 const int64_t* item_prefix, int32_t* queue, uint32_t* out, hipStream_t st) {
    // persistent waves: 256 CUs x 8 waves/SIMD x 4 SIMDs / 4 waves per block
    if (g_sweep_variant == 1)
      hipLaunchKernelGGL((sweep_kernel<DIM, false>), dim3(256 * 8), dim3(SWEEP_THREADS), 0, st, cand,
                         cand_off, counts, verts, weights, k1, R, ns, item_prefix, queue, out);
    else
      hipLaunchKernelGGL((sweep_kernel<DIM, true>), dim3(256 * 8), dim3(SWEEP_THREADS), 0, st, cand,
                         cand_off, counts, verts, weights, k1, R, ns, item_prefix, queue, out);
    return check_launch("sweep");
  }
};

template <int DIM>
struct FpsOp {
  static int run(const float* pts, int64_t n, int ld, int n_lms, int64_t start, float* mind,
                 unsigned long long* best, int64_t* out_idx, hipStream_t st) {
    if constexpr (DIM <= 3) {
      // rows = (x, y, z, running min): mind is used as the 4*n float row buffer
      float4* rows = reinterpret_cast<float4*>(mind);
      int64_t ib = (n + 255) / 256;
      if (ib > 4096) ib = 4096;
      hipLaunchKernelGGL((fps_rows_init_kernel<DIM>), dim3((int)ib), dim3(256), 0, st, pts, n, ld, rows);
      const int64_t blocks = (n + 1023) / 1024;
      hipLaunchKernelGGL(fps_fast_init_kernel, dim3(1), dim3(1), 0, st, best, start);
      for (int it = 1; it < n_lms; ++it)
        hipLaunchKernelGGL(fps_fast_kernel, dim3((unsigned)blocks), dim3(256), 0, st, rows, n, it, best, out_idx);
      hipLaunchKernelGGL(fps_fast_last_kernel, dim3(1), dim3(64), 0, st, best, n_lms, out_idx);
      return check_launch("fps_fast");
    } else {
      int64_t blocks = (n + 256 * 4 - 1) / (256 * 4);
      if (blocks > 2048) blocks = 2048;
      if (blocks < 1) blocks = 1;
      hipLaunchKernelGGL(fps_init_kernel, dim3(1), dim3(1), 0, st, best, start);
      for (int it = 1; it < n_lms; ++it) {
        hipLaunchKernelGGL((fps_step_kernel<DIM>), dim3((int)blocks), dim3(256), 0, st, pts, n, ld, it,
                           mind, best, out_idx);
      }
      hipLaunchKernelGGL(fps_last_kernel, dim3(1), dim3(1), 0, st, best, n_lms, out_idx);
      return check_launch("fps_step");
    }
  }
};

}  // namespace

extern "C" {

int flooder_abi_version(void) { return FLOODER_ABI_VERSION; }

const char* flooder_last_error(void) { return err_buf(); }

int flooder_padded_dim(int dim) { return padded_dim(dim); }

int flooder_set_option(const char* name, int value) {
  if (name && strcmp(name, "sweep_variant") == 0 && (value == 0 || value == 1)) {
    g_sweep_variant = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "bvh_ks") == 0 && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8)) {
    g_bvh_ks = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "bvh_refine_pct") == 0 && value >= 1) {
    g_bvh_refine_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "bvh_leaf_batch") == 0 && (value == 1 || value == 4)) {
    g_bvh_leaf_batch = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "curve") == 0 && (value == 0 || value == 1)) {
    g_curve = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_exh_sparse") == 0 && value >= 480) {
    g_cell_exh_sparse = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "finish_budget_min") == 0 && value >= 1) {
    g_finish_budget_min = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "sort_shape") == 0 && value >= 0 && value <= 3) {
    g_sort_shape = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_split_launches") == 0 && (value == 1 || value == 2)) {
    g_cell_split_launches = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_surface_pct") == 0 && value >= 0 && value <= 100) {
    g_cell_surface_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "finish_wide_points") == 0 && value >= 0) {
    g_finish_wide_points = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "finish_refresh") == 0 && value >= 1) {
    g_finish_refresh = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "finish_focus_pct") == 0 && value >= 0 && value <= 100) {
    g_finish_focus_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "fps_switch") == 0 && value >= 0) {
    g_fps_switch = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "fps_lane_best") == 0 && (value == 0 || value == 1)) {
    g_fps_lane_best = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "fps_rounds") == 0 && (value == 0 || value == 1)) {
    g_fps_rounds = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "fps_rpl") == 0 && (value == 0 || value == 1 || value == 4)) {
    g_fps_rpl = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_exh_tries") == 0 && value >= 0 && value <= 8) {
    g_cell_exh_tries = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "finish_items_cap") == 0 && value >= 1024) {
    g_finish_items_cap = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "finish_budget") == 0 && value >= 0) {
    g_finish_budget = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "finish_top") == 0 && (value == 0 || value == 1)) {
    g_finish_top = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_retry_keep") == 0 && value >= 0) {
    g_cell_retry_keep = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_retry_pct") == 0 && value >= 0) {
    g_cell_retry_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "finish_order") == 0 && (value == 0 || value == 1)) {
    g_finish_order = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "curve_bits") == 0 && value >= 0 && value <= 21) {
    g_curve_bits = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_super_weight") == 0 && value >= 0) {
    g_cell_super_weight = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_chunks_per_block") == 0 && value >= 1) {
    g_cell_chunks_per_block = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_weight_classes") == 0 && (value == 0 || value == 1)) {
    g_cell_weight_classes = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_listed_first") == 0 && (value == 0 || value == 1)) {
    g_cell_listed_first = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_tail_waves") == 0 && value >= 0) {
    g_cell_tail_waves = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_weight") == 0 && value >= 0) {
    g_wit_weight = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_cmax_pct") == 0 && value >= 10 && value <= 10000) {
    g_wit_cmax_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_cmax_ext_pct") == 0 && value >= 1 && value <= 10000) {
    g_wit_cmax_ext_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_max_in_pct") == 0 && value >= 0) {
    g_wit_max_in_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_max_leaves") == 0 && value >= 1) {
    g_wit_max_leaves = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_chunk_major") == 0 && (value == 0 || value == 1 || value == 2)) {
    g_cell_chunk_major = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_drop") == 0 && (value == 0 || value == 1)) {
    g_cell_drop = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "sorted_batch_pct") == 0 && value >= 100) {
    g_sorted_batch_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "sorted_blocks") == 0 && value >= 0) {
    g_sorted_blocks = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "sorted_refresh") == 0 && value >= 1) {
    g_sorted_refresh = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_max_eval") == 0 && value >= 0) {
    g_wit_max_eval = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_max_open") == 0 && value >= 0) {
    g_wit_max_open = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_max_live_pct") == 0 && value >= 0 && value <= 100) {
    g_wit_max_live_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_adaptive") == 0 && (value == 0 || value == 1)) {
    g_wit_adaptive = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_flags") == 0 && value >= 0) {
    g_wit_flags = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_grid") == 0 && value >= 1) {
    g_wit_grid = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_surface_pct") == 0 && value >= 0 && value <= 100) {
    g_wit_surface_pct = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "wit_min_bins") == 0 && value >= 1 && value <= 64) {
    g_wit_min_bins = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_one_pass") == 0 && value >= 0 && value <= 100000) {
    g_cell_one_pass = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_queue_block") == 0 && value >= -1 && value <= 12) {
    g_cell_queue_block = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_min_grid") == 0 && value >= 1) {
    g_cell_min_grid = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_super_min_chunks") == 0 && value >= 0) {
    g_cell_super_min_chunks = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_super_sparse") == 0 && value >= 0) {
    g_cell_super_sparse = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_super_n0") == 0 && value >= 0) {
    g_cell_super_n0 = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_density_grid") == 0 && value >= 0) {
    g_cell_density_grid = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "sorted_ks") == 0 && (value == 1 || value == 2)) {
    g_sorted_ks = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_tiles") == 0 && value >= 0 && value <= 2) {
    g_cell_tiles = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_tries") == 0 && value >= 1 && value <= 8) {
    g_cell_tries = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_brute_max") == 0 && value >= 0) {
    g_cell_brute_max = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_exh_dense") == 0 && value >= 512) {
    g_cell_exh_dense = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "cell_grid") == 0 && value >= 1 && value <= 65536) {
    g_cell_grid = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "bvh_grid") == 0 && value >= 1 && value <= 65536) {
    g_bvh_grid = value;
    return FLOODER_OK;
  }
  if (name && strcmp(name, "bvh_subs") == 0 && value >= 1 && value <= 64 && (value & (value - 1)) == 0) {
    g_bvh_subs = value;
    return FLOODER_OK;
  }
  return fail(FLOODER_E_ARG, "flooder_set_option: unknown option or value");
}

int flooder_device_arch(int device, char* buf, int buflen) {
  if (!buf || buflen <= 0) return fail(FLOODER_E_ARG, "flooder_device_arch: null buffer");
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail(FLOODER_E_DEVICE, hipGetErrorString(e));
  snprintf(buf, (size_t)buflen, "%s", prop.gcnArchName);
  return FLOODER_OK;
}

int flooder_ball_count_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* centers,
                           const float* radii, const int64_t* slab_lo, const int64_t* slab_hi,
                           int64_t n_simplices, int32_t* counts, void* stream) {
  if (n_simplices == 0) return FLOODER_OK;
  if (!pts || !centers || !radii || !slab_lo || !slab_hi || !counts || n_pts < 0 || ld < dim)
    return fail(FLOODER_E_ARG, "flooder_ball_count_f32: bad argument");
  return dispatch_dim<CountOp>(dim, pts, ld, centers, radii, slab_lo, slab_hi, n_simplices, counts,
                               (hipStream_t)stream);
}

int flooder_ball_fill_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* centers,
                          const float* radii, const int64_t* slab_lo, const int64_t* slab_hi,
                          int64_t n_simplices, const int32_t* counts, const int64_t* cand_off,
                          int32_t* cursor, float* cand, void* stream) {
  if (n_simplices == 0) return FLOODER_OK;
  if (!pts || !centers || !radii || !slab_lo || !slab_hi || !counts || !cand_off || !cursor ||
      !cand || n_pts < 0 || ld < dim)
    return fail(FLOODER_E_ARG, "flooder_ball_fill_f32: bad argument");
  return dispatch_dim<FillOp>(dim, pts, ld, centers, radii, slab_lo, slab_hi, n_simplices, counts,
                              cand_off, cursor, cand, (hipStream_t)stream);
}

int flooder_sweep_f32(const float* cand, const int64_t* cand_off, const int32_t* counts, int dim,
                      const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                      const int64_t* item_prefix, int32_t* queue, uint32_t* out_d2, void* stream) {
  if (n_simplices == 0 || R == 0) return FLOODER_OK;
  if (!cand_off || !counts || !verts || !weights || !item_prefix || !queue || !out_d2 || k1 < 1 ||
      k1 > FLOODER_MAX_VERTS || R < 0)
    return fail(FLOODER_E_ARG, "flooder_sweep_f32: bad argument");
  return dispatch_dim<SweepOp>(dim, cand, cand_off, counts, verts, weights, k1, R, n_simplices,
                               item_prefix, queue, out_d2, (hipStream_t)stream);
}

int flooder_face_max_f32(const uint32_t* d2, int64_t n_simplices, int R, const int32_t* face_ptr,
                         const int32_t* face_rows, int n_faces, float* out_face, float* out_dist,
                         void* stream) {
  if (n_simplices == 0) return FLOODER_OK;
  if (!d2 || !face_ptr || !face_rows || !out_face || n_faces < 1 || R < 1)
    return fail(FLOODER_E_ARG, "flooder_face_max_f32: bad argument");
  if (n_simplices > 0x7fffffff) return fail(FLOODER_E_ARG, "flooder_face_max_f32: too many simplices");
  if (R <= 64 && n_faces <= 32) {
    int lgF = 1;   // (at most 32 simplices per step: the LDS rows of a wave)
    while ((1 << lgF) < n_faces) ++lgF;
    const int per_block = 4 * (64 >> lgF);
    int64_t blocks = (n_simplices + per_block - 1) / per_block;
    const size_t lds = (size_t)4 * (64 >> lgF) * 64 * sizeof(uint32_t);   // 2 - 32 KB
    const int64_t per_cu = lds <= 8192 ? 8 : (lds <= 16384 ? 6 : 4);      // resident blocks (32 waves, 160 KB a CU)
    if (blocks > per_cu * 256) blocks = per_cu * 256;
    hipLaunchKernelGGL(face_max_small_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, d2, n_simplices, R,
                       face_ptr, face_rows, n_faces, lgF, out_face, out_dist);
    return check_launch("face_max");
  }
  hipLaunchKernelGGL(face_max_kernel, dim3((unsigned)n_simplices), dim3(256), 0, (hipStream_t)stream,
                     d2, R, face_ptr, face_rows, n_faces, out_face, out_dist);
  return check_launch("face_max");
}

int flooder_fill_u32(uint32_t* buf, int64_t n, uint32_t value, void* stream) {
  if (n == 0) return FLOODER_OK;
  if (!buf || n < 0) return fail(FLOODER_E_ARG, "flooder_fill_u32: bad argument");
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, buf, n,
                     value);
  return check_launch("fill_u32");
}

int flooder_fps_f32(const float* pts, int64_t n_pts, int dim, int ld, int n_lms, int64_t start,
                    int64_t* out_idx, float* work_min, uint64_t* work_best, void* stream) {
  if (!pts || !out_idx || !work_min || !work_best || n_pts < 1 || n_lms < 1 || n_lms > n_pts ||
      start < 0 || start >= n_pts || ld < dim || n_pts > 0xfffffffeLL)
    return fail(FLOODER_E_ARG, "flooder_fps_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* best = reinterpret_cast<unsigned long long*>(work_best);
  return dispatch_dim<FpsOp>(dim, pts, n_pts, ld, n_lms, start, work_min, best, out_idx, st);
}

}  // extern "C"
