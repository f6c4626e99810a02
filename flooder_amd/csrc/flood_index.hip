// flood_index.hip - sort of the curve codes and gather of the cloud into curve order (gfx950).
//
// The cloud is sorted once per flood_complex call (the counterpart of the reference's argsort of the widest
// coordinate, flooder/core.py:140-144).  The sort itself is the library primitive rocprim::radix_sort_pairs -
// restricted to the bits the codes really use (flooder_curve_key_bits: 36 in 3D = five 8-bit passes instead of
// the eight a 64-bit argsort pays) and producing 32-bit row indices directly; the gather writes the padded,
// +inf-terminated row layout the box tree and the sweeps read.

#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "flood_common.hpp"

using namespace flooder;

namespace {

template <int DIM>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                          const uint32_t* __restrict__ order,
                                                          float* __restrict__ out, int64_t n_pad) {
  constexpr int DP = padded_dim(DIM);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_pad; j += stride) {
    float x[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) x[k] = j < n ? 0.f : __builtin_inff();  // pad rows: +inf (never a nearest neighbour)
    if (j < n) {
      const int64_t src = (int64_t)order[j];
#pragma unroll
      for (int k = 0; k < DIM; ++k) x[k] = pts[src * ld + k];
    }
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
}

// Sub-cloud of a block of simplices in a block-sharded run: the rows of the cloud inside the UNION of the simplices'
// bounding balls - with landmarks that are cloud points every witness of a simplex lies in its ball (core.py:156-172
// of the reference).  The union is taken on a coarse grid over the cloud's box (SELECT_G cells per axis in 2D / 3D;
// one flag per cell; any other dimension: the balls' common bounding box): ball_cells_kernel flags every cell a ball
// reaches, select_rows_kernel compacts the rows whose cell is flagged (any order).
template <int DIM>
struct SelectCfg {
  static constexpr int G = DIM == 2 ? 128 : 32;
  static constexpr int NC = DIM == 2 ? G * G : (DIM == 3 ? G * G * G : 1);
};

template <int DIM>
__global__ __launch_bounds__(256) void ball_cells_kernel(const float* __restrict__ centers, const float* __restrict__ radii,
                                                         int64_t n_balls, const float* __restrict__ cbox,
                                                         uint8_t* __restrict__ flags) {
  typedef SelectCfg<DIM> C;
  const int64_t b = blockIdx.x;
  if (b >= n_balls) return;
  float c[DIM], lo[DIM], cell[DIM];
  int i0[DIM], n[DIM];
  const float r = radii[b];
  int total = 1;
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    c[k] = centers[b * DIM + k];
    lo[k] = cbox[k];
    const float e = cbox[8 + k] - cbox[k];
    cell[k] = e > 0.f ? e / (float)C::G : 1.f;
    int a = (int)__builtin_floorf((c[k] - r - lo[k]) / cell[k]) - 1;   // (one cell of slack either side: rounding)
    int z = (int)__builtin_floorf((c[k] + r - lo[k]) / cell[k]) + 1;
    a = a < 0 ? 0 : (a > C::G - 1 ? C::G - 1 : a);
    z = z < 0 ? 0 : (z > C::G - 1 ? C::G - 1 : z);
    i0[k] = a;
    n[k] = z - a + 1;
    total *= n[k];
  }
  const float r2 = r * r * 1.0001f + 1e-30f;
  for (int t = threadIdx.x; t < total; t += blockDim.x) {
    int rem = t, id = 0, mul = 1;
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const int ck = i0[k] + rem % n[k];
      rem /= n[k];
      id += ck * mul;
      mul *= C::G;
      // distance from the centre to the cell's slab along k (cells at the rim of the grid reach to infinity: points
      // on the cloud's box round into them)
      const float clo = ck == 0 ? -__builtin_inff() : lo[k] + (float)ck * cell[k] - 1e-6f * __builtin_fabsf(lo[k]) - 1e-30f;
      const float chi = ck == C::G - 1 ? __builtin_inff() : lo[k] + (float)(ck + 1) * cell[k] + 1e-6f * __builtin_fabsf(lo[k]) + 1e-30f;
      const float gap = __builtin_fmaxf(__builtin_fmaxf(clo - c[k], c[k] - chi), 0.f);
      d2 = __builtin_fmaf(gap, gap, d2);
    }
    if (d2 <= r2 && flags[id] == 0) flags[id] = 1;
  }
}

// rows whose cell is flagged (flags != NULL) or that lie inside the box, compacted; every thread takes ROWS rows per
// step and a workgroup reserves its rows of a step with ONE atomic (a same-address atomic per wave is 12 ns each:
// 3 ms for 16 M rows)
template <int DIM, bool CELLS>
__global__ __launch_bounds__(256) void select_rows_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                          const float* __restrict__ box, const float* __restrict__ cbox,
                                                          const uint8_t* __restrict__ flags, float* __restrict__ out,
                                                          int32_t* __restrict__ count) {
  typedef SelectCfg<DIM> C;
  constexpr int ROWS = 8;
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float lo[DIM], hi[DIM], glo[DIM], gsc[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    lo[k] = box[k];
    hi[k] = box[DIM + k];
    if constexpr (CELLS) {
      glo[k] = cbox[k];
      const float e = cbox[8 + k] - cbox[k];
      gsc[k] = e > 0.f ? (float)C::G / e : 0.f;
    }
  }
  const int64_t per_step = (int64_t)blockDim.x * ROWS;
  const int64_t stride = (int64_t)gridDim.x * per_step;
  for (int64_t j0 = (int64_t)blockIdx.x * per_step; j0 < n; j0 += stride) {   // (block-uniform trip count)
    float x[ROWS][DIM];
    bool in[ROWS];
    int mine = 0;
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      const int64_t j = j0 + (int64_t)u * blockDim.x + threadIdx.x;
      in[u] = j < n;
#pragma unroll
      for (int k = 0; k < DIM; ++k) x[u][k] = in[u] ? pts[j * ld + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) in[u] = in[u] && x[u][k] >= lo[k] && x[u][k] <= hi[k];
      if constexpr (CELLS) {
        int id = 0, mul = 1;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          int ck = (int)((x[u][k] - glo[k]) * gsc[k]);
          ck = ck < 0 ? 0 : (ck > C::G - 1 ? C::G - 1 : ck);
          id += ck * mul;
          mul *= C::G;
        }
        in[u] = in[u] && flags[in[u] ? id : 0] != 0;
      }
      mine += in[u] ? 1 : 0;
    }
    // exclusive offsets: lanes inside the wave, waves inside the block, the block in the output
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
      s_base = tot > 0 ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    int64_t dst = (int64_t)s_base + (incl - mine);
    for (int w = 0; w < wv; ++w) dst += s_wave[w];
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      if (in[u]) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) out[dst * DIM + k] = x[u][k];
        ++dst;
      }
    }
    __syncthreads();
  }
}

template <int DIM>
struct BoxSelectOp {
  static int run(const float* pts, int64_t n, int ld, const float* box, const float* cbox, const float* centers,
                 const float* radii, int64_t n_balls, uint8_t* flags, float* out, int32_t* count, hipStream_t st) {
    int64_t blocks = (n + 2047) / 2048;
    if (blocks > 2048) blocks = 2048;
    if constexpr (DIM == 2 || DIM == 3) {
      if (flags && centers && radii && cbox && n_balls > 0) {
        hipLaunchKernelGGL((ball_cells_kernel<DIM>), dim3((unsigned)n_balls), dim3(256), 0, st, centers, radii, n_balls, cbox,
                           flags);
        hipLaunchKernelGGL((select_rows_kernel<DIM, true>), dim3((int)blocks), dim3(256), 0, st, pts, n, ld, box, cbox, flags,
                           out, count);
        return check_launch("select_rows");
      }
    }
    hipLaunchKernelGGL((select_rows_kernel<DIM, false>), dim3((int)blocks), dim3(256), 0, st, pts, n, ld, box, nullptr,
                       nullptr, out, count);
    return check_launch("select_rows");
  }
};

template <int DIM>
struct GatherOp {
  static int run(const float* pts, int64_t n, int ld, const uint32_t* order, float* out, int64_t n_pad,
                 hipStream_t st) {
    int64_t blocks = (n_pad + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((gather_rows_kernel<DIM>), dim3((int)blocks), dim3(256), 0, st, pts, n, ld, order, out, n_pad);
    return check_launch("gather_rows");
  }
};

}  // namespace

// rocprim sorts up to 2^20 keys by merge sort: a block sort and ten merge passes of two launches each - 21 launches,
// 150 us for the million points of cfg 2.  The onesweep radix sort (a histogram launch and one launch per 8 bits) is
// what it uses above that size and is the faster one from a tenth of it on.
// (the four-parameter radix_sort_config - single / merge / onesweep configuration + merge-sort limit - is the form of
// rocPRIM 3.x and later; an older toolchain gets the library's default configuration: same results, slower below 2^20 keys)
#if defined(ROCPRIM_VERSION) && ROCPRIM_VERSION >= 300000
using SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config,
                                           (size_t)128 * 1024>;
#else
using SortCfg = rocprim::default_config;
#endif

namespace {

// ------------------------------------------------------------------------------------ k-d order (above 3D)
// A space-filling curve cuts space on a FIXED grid: in 6D a run of 16 consecutive points of the Hilbert order straddles
// cell boundaries, and the 1024-point nodes above overlap each other - a nearest-neighbour ball of a cfg 4 sample meets
// 19 of them where a k-d partition of the same points offers 6.  kd_order sorts the cloud into the order of a balanced
// k-d tree whose cells are EXACTLY the aligned groups of 16 * 2^j rows the implicit box tree is made of: position
// space is padded (virtually) to P = 16 * 2^k rows; at level t every aligned segment of P >> t positions is sorted
// along the widest axis of its bounding box and thereby split, by position, into its two halves.  One level = a
// segmented bounding box (kd_box_kernel), a key (segment, coordinate quantised to 8 bits inside the segment's box)
// per point (kd_key_kernel) and one radix sort of the whole cloud over the key's bits.  The quantised coordinate makes
// the split approximate by 2^-8 of the segment's extent (one radix pass less per level than 16 bits, same sweep); any
// order is a valid index (the boxes are taken from the rows, whatever they are), so results do not depend on it.
__device__ __forceinline__ uint32_t ordered_bits(float x) {
  const uint32_t b = __float_as_uint(x);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu));
}

// boxes[seg * 16 + k] = ordered bits of the minimum, [seg * 16 + 8 + k] of the maximum; lg = log2(segment size).
// A wave takes 64 * R consecutive positions (inside one segment, or, for 32-position segments, two of them).
// TEAM (segments of 16 chunks and more): sixteen waves of one workgroup, whose chunks lie in ONE segment, combine their
// boxes in LDS and the workgroup adds one box - at the top levels every wave of the launch adds into the same one or two
// boxes, and 12 same-address atomics from each of a thousand waves were 139 / 80 / 65 us for the first three levels of
// cfg 4 where a level without contention takes 13.
template <int DIM, bool TEAM>
__global__ __launch_bounds__(TEAM ? 1024 : 256) void kd_box_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                                   const uint32_t* __restrict__ order, int lg, int R,
                                                                   uint32_t* __restrict__ boxes, int direct) {
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t per_wave = (int64_t)64 * R;
  const int64_t n_chunks = (n + per_wave - 1) / per_wave;
  if constexpr (TEAM) {   // (one chunk per wave, the grid covers the chunks; lg >= 6 + log2(16 R))
    __shared__ float s_box[16][2 * DIM];
    const int wv = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 16 + wv;
    float lo[DIM], hi[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
    if (w < n_chunks) {
      const int64_t base = w * per_wave;
      for (int r = 0; r < R; r += (R >= 32 ? 4 : 1)) {   // (as below: every fourth row group)
        const int64_t pos = base + (int64_t)r * 64 + lane;
        if (pos < n) {
          const float* x = pts + (int64_t)order[pos] * ld;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            const float v = x[k];
            lo[k] = __builtin_fminf(lo[k], v);
            hi[k] = __builtin_fmaxf(hi[k], v);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < DIM; ++k) { lo[k] = wave_min_f32(lo[k]); hi[k] = wave_max_f32(hi[k]); }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) { s_box[wv][k] = lo[k]; s_box[wv][DIM + k] = hi[k]; }
    }
    __syncthreads();
    if (wv == 0 && lane < 2 * DIM && (int64_t)blockIdx.x * 16 < n_chunks) {
      float v = s_box[0][lane];
#pragma unroll
      for (int j = 1; j < 16; ++j) v = lane < DIM ? __builtin_fminf(v, s_box[j][lane]) : __builtin_fmaxf(v, s_box[j][lane]);
      uint32_t* bx = boxes + (((int64_t)blockIdx.x * 16 * per_wave) >> lg) * 16;
      if (lane < DIM) atomicMin(bx + lane, ordered_bits(v));
      else atomicMax(bx + 8 + (lane - DIM), ordered_bits(v));
    }
    return;
  }
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_chunks; w += waves) {
    float lo[DIM], hi[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
    const int64_t base = w * per_wave;
    // (R = 32, segments of 2048 positions and more: every fourth row group - the box of 512+ sampled rows picks the same
    // axis, and a coordinate outside it is clamped into the first or last of the key's 256 buckets, far from the median)
    for (int r = 0; r < R; r += (R >= 32 ? 4 : 1)) {
      const int64_t pos = base + (int64_t)r * 64 + lane;
      if (pos < n) {
        const float* x = pts + (int64_t)order[pos] * ld;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float v = x[k];
          lo[k] = __builtin_fminf(lo[k], v);
          hi[k] = __builtin_fmaxf(hi[k], v);
        }
      }
    }
    if (lg >= 6) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) { lo[k] = wave_min_f32(lo[k]); hi[k] = wave_max_f32(hi[k]); }
      const int64_t seg = base >> lg;
      if (lane == 0) {
        uint32_t* b = boxes + seg * 16;
        if (direct) {  // the wave held the whole segment
#pragma unroll
          for (int k = 0; k < DIM; ++k) { b[k] = ordered_bits(lo[k]); b[8 + k] = ordered_bits(hi[k]); }
        } else {
#pragma unroll
          for (int k = 0; k < DIM; ++k) { atomicMin(b + k, ordered_bits(lo[k])); atomicMax(b + 8 + k, ordered_bits(hi[k])); }
        }
      }
    } else {  // 32 positions per segment: the two halves of the wave
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          lo[k] = __builtin_fminf(lo[k], __shfl_xor(lo[k], o));
          hi[k] = __builtin_fmaxf(hi[k], __shfl_xor(hi[k], o));
        }
      }
      if ((lane & 31) == 0 && base + lane < n) {
        uint32_t* b = boxes + ((base + lane) >> 5) * 16;
#pragma unroll
        for (int k = 0; k < DIM; ++k) { b[k] = ordered_bits(lo[k]); b[8 + k] = ordered_bits(hi[k]); }
      }
    }
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void kd_key_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                     const uint32_t* __restrict__ order, int lg, int cbits,
                                                     const uint32_t* __restrict__ boxes, uint32_t* __restrict__ keys,
                                                     uint32_t* __restrict__ zero_buf, int64_t zero_words) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float top = (float)((1u << cbits) - 1u);
  // (the state of the radix sort that follows - its histograms, look-back arrays, block tickets: os_sort_pairs below)
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < zero_words; j += stride) zero_buf[j] = 0u;
  for (int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += stride) {
    const int64_t seg = pos >> lg;
    const uint32_t* b = boxes + seg * 16;
    int axis = 0;
    float ext = -1.f, alo = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const float l = from_ordered_bits(b[k]), h = from_ordered_bits(b[8 + k]);
      const float e = h - l;
      if (e > ext) { ext = e; axis = k; alo = l; }
    }
    const float v = pts[(int64_t)order[pos] * ld + axis];
    float t = ext > 0.f ? (v - alo) / ext * top : 0.f;
    t = t < 0.f ? 0.f : (t > top ? top : t);   // (also a NaN extent - an infinite coordinate - lands on 0)
    keys[pos] = ((uint32_t)seg << cbits) | (uint32_t)t;
  }
}

// The last KD_LOCAL_LG - 4 levels of the partition inside a workgroup: a segment of 1024 positions (its rows' coordinates
// in LDS, one array per axis) is split down to its 16-row leaves by bitonic sorts of a (coordinate, slot) pair per row -
// per level a bounding box per sub-segment (LDS atomics on ordered bits), the widest axis, one sort of every sub-segment
// along its axis (exact coordinates: no quantisation here).  Replaces six rounds of (box, key, radix sort of the whole
// cloud) - 3.3 -> 1.9 ms at 2 M points.  Positions beyond n_pts are +inf rows: they sort to the end and stay there.
constexpr int KD_LOCAL_LG = 10;
constexpr int KD_LOCAL = 1 << KD_LOCAL_LG;

template <int DIM>
__global__ __launch_bounds__(256) void kd_local_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                       const uint32_t* __restrict__ order_in,
                                                       uint32_t* __restrict__ order_out) {
  __shared__ float s_x[DIM][KD_LOCAL];
  __shared__ uint32_t s_id[KD_LOCAL];
  __shared__ uint32_t s_key[KD_LOCAL];
  __shared__ uint16_t s_perm[KD_LOCAL];
  __shared__ uint32_t s_box[KD_LOCAL / 32][2 * DIM];
  __shared__ uint8_t s_axis[KD_LOCAL / 32];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * KD_LOCAL;
  const int cnt = (int)((n - base) < KD_LOCAL ? (n - base) : KD_LOCAL);
  for (int i = tid; i < KD_LOCAL; i += 256) {
    const bool real = i < cnt;
    const uint32_t id = real ? order_in[base + i] : 0u;
    s_id[i] = id;
    s_perm[i] = (uint16_t)i;
#pragma unroll
    for (int k = 0; k < DIM; ++k) s_x[k][i] = real ? pts[(int64_t)id * ld + k] : __builtin_inff();
  }
  __syncthreads();
  for (int lg = KD_LOCAL_LG; lg > 4; --lg) {   // sub-segments of 2^lg positions are split in two
    const int m = 1 << lg, n_sub = KD_LOCAL >> lg;
    for (int i = tid; i < n_sub * 2 * DIM; i += 256) s_box[i / (2 * DIM)][i % (2 * DIM)] = (i % (2 * DIM)) < DIM ? 0xffffffffu : 0u;
    __syncthreads();
    {  // bounding boxes: a thread's four consecutive positions lie in one sub-segment
      float lo[DIM], hi[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
      bool any = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int slot = s_perm[tid * 4 + u];
        if (slot < cnt) {
          any = true;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            const float v = s_x[k][slot];
            lo[k] = __builtin_fminf(lo[k], v);
            hi[k] = __builtin_fmaxf(hi[k], v);
          }
        }
      }
      if (any) {
        uint32_t* b = s_box[(tid * 4) >> lg];
#pragma unroll
        for (int k = 0; k < DIM; ++k) { atomicMin(b + k, ordered_bits(lo[k])); atomicMax(b + DIM + k, ordered_bits(hi[k])); }
      }
    }
    __syncthreads();
    if (tid < n_sub) {
      int axis = 0;
      float ext = -1.f;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const float e = from_ordered_bits(s_box[tid][DIM + k]) - from_ordered_bits(s_box[tid][k]);
        if (e > ext) { ext = e; axis = k; }
      }
      s_axis[tid] = (uint8_t)axis;
    }
    __syncthreads();
    for (int i = tid; i < KD_LOCAL; i += 256) {
      const int axis = s_axis[i >> lg];
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < DIM; ++k) v = axis == k ? s_x[k][s_perm[i]] : v;
      s_key[i] = ordered_bits(v);
    }
    __syncthreads();
    // bitonic sort of every aligned run of m positions, ascending
    for (int k = 2; k <= m; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
        for (int u = 0; u < KD_LOCAL / 512; ++u) {
          const int pidx = tid + 256 * u;
          const int i = ((pidx & ~(j - 1)) << 1) | (pidx & (j - 1));
          const int q = i | j;
          const bool asc = k == m ? true : ((i & k) == 0);
          const uint32_t a = s_key[i], b = s_key[q];
          if ((a > b) == asc && a != b) {
            s_key[i] = b;
            s_key[q] = a;
            const uint16_t t = s_perm[i];
            s_perm[i] = s_perm[q];
            s_perm[q] = t;
          }
        }
        __syncthreads();
      }
    }
  }
  for (int i = tid; i < cnt; i += 256) order_out[base + i] = s_id[s_perm[i]];
}

__global__ __launch_bounds__(256) void kd_iota_kernel(uint32_t* __restrict__ order, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) order[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void kd_box_init_kernel(uint32_t* __restrict__ boxes, int64_t n_words) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride)
    boxes[i] = (i & 8) ? 0u : 0xffffffffu;
}

// coordinate bits of a level's sort key: the split lands within 2^-KD_CBITS of the segment's extent of the median
constexpr int KD_CBITS = 8;
inline int kd_levels(int64_t n, int& log_p) {   // P = 16 << k >= n; levels t = 0 .. k-1
  int k = 0;
  while (((int64_t)FLOODER_BVH_LEAF << k) < n) ++k;
  log_p = 4 + k;
  return k;
}
inline int64_t kd_align(int64_t b) { return (b + 255) / 256 * 256; }

// (defined with flooder_index_sort_zeroed below: the library's radix passes on caller-zeroed state, no fill launches)
int64_t os_state_words(int64_t n, int key_bits);
int os_sort_pairs(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t* keys_tmp,
                  uint32_t* vals_tmp, uint32_t n, int key_bits, uint32_t* state, hipStream_t st);

template <int DIM>
struct KdOrderOp {
  static int run(const float* pts, int64_t n, int ld, uint32_t* order_out, uint8_t* tmp, int64_t sort_bytes,
                 hipStream_t st) {
    int log_p = 0;
    const int levels = kd_levels(n, log_p);
    // tmp: keys | keys_sorted | second order buffer | boxes | rocprim scratch
    uint32_t* keys = reinterpret_cast<uint32_t*>(tmp);
    uint32_t* keys2 = reinterpret_cast<uint32_t*>(tmp + kd_align(n * 4));
    uint32_t* other = reinterpret_cast<uint32_t*>(tmp + 2 * kd_align(n * 4));
    const int64_t box_segs = levels > 0 ? ((int64_t)1 << (levels - 1)) : 1;
    uint32_t* boxes = reinterpret_cast<uint32_t*>(tmp + 3 * kd_align(n * 4));
    void* scratch = tmp + 3 * kd_align(n * 4) + kd_align(box_segs * 64);
    // the global rounds split the segments down to KD_LOCAL positions, kd_local_kernel does the rest; the order
    // ping-pongs between the two buffers: start so that the last hop writes order_out
    const int global_levels = log_p > KD_LOCAL_LG ? log_p - KD_LOCAL_LG : 0;
    const int hops = global_levels + 1;
    uint32_t* cur = (hops & 1) ? other : order_out;
    uint32_t* nxt = (hops & 1) ? order_out : other;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(kd_iota_kernel, dim3((int)blocks), dim3(256), 0, st, cur, n);
    for (int t = 0; t < global_levels; ++t) {
      const int lg = log_p - t;                       // segment size 2^lg > KD_LOCAL
      const int64_t n_seg = ((n - 1) >> lg) + 1;
      const int R = lg - 6 >= 5 ? 32 : (1 << (lg - 6));
      const int direct = ((int64_t)64 * R) == ((int64_t)1 << lg) ? 1 : 0;
      if (!direct) {
        int64_t ib = (n_seg * 16 + 255) / 256;
        hipLaunchKernelGGL(kd_box_init_kernel, dim3((int)(ib > 1024 ? 1024 : ib)), dim3(256), 0, st, boxes, n_seg * 16);
      }
      const int64_t chunks = (n + (int64_t)64 * R - 1) / ((int64_t)64 * R);
      int64_t bb = (chunks + 3) / 4;
      if (bb > 8192) bb = 8192;
      if (!direct && ((int64_t)1 << lg) >= (int64_t)16 * 64 * R)   // a segment holds sixteen chunks and more: teams
        hipLaunchKernelGGL((kd_box_kernel<DIM, true>), dim3((unsigned)((chunks + 15) / 16)), dim3(1024), 0, st, pts, n, ld, cur,
                           lg, R, boxes, direct);
      else
        hipLaunchKernelGGL((kd_box_kernel<DIM, false>), dim3((int)bb), dim3(256), 0, st, pts, n, ld, cur, lg, R, boxes, direct);
      const int cbits = 32 - t < KD_CBITS ? 32 - t : KD_CBITS;
      // scratch: keys and rows of the odd passes | the sort's state, cleared by the key kernel on its way (a level's sort
      // has run when the next level's key kernel starts); clouds of 2^30 rows and more: the library call
      const int64_t state_words = os_state_words(n, 32);
      uint32_t* state = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(scratch) + 2 * kd_align(n * 4));
      const bool own_sort = state_words > 0 && sort_bytes >= 2 * kd_align(n * 4) + 4 * state_words;
      hipLaunchKernelGGL((kd_key_kernel<DIM>), dim3((int)blocks), dim3(256), 0, st, pts, n, ld, cur, lg, cbits, boxes, keys,
                         own_sort ? state : nullptr, own_sort ? state_words : (int64_t)0);
      if (own_sort) {
        uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(scratch);
        uint32_t* vals_tmp = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(scratch) + kd_align(n * 4));
        const int rc = os_sort_pairs(keys, keys2, cur, nxt, keys_tmp, vals_tmp, (uint32_t)n, t + cbits, state, st);
        if (rc != FLOODER_OK) return rc;
      } else {
        size_t bytes = (size_t)sort_bytes;
        hipError_t e = rocprim::radix_sort_pairs<SortCfg>(scratch, bytes, keys, keys2, cur, nxt, (size_t)n, 0u,
                                                          (unsigned)(t + cbits), st);
        if (e != hipSuccess) return fail(FLOODER_E_LAUNCH, hipGetErrorString(e));
      }
      uint32_t* s = cur; cur = nxt; nxt = s;
    }
    hipLaunchKernelGGL((kd_local_kernel<DIM>), dim3((unsigned)((n + KD_LOCAL - 1) / KD_LOCAL)), dim3(256), 0, st, pts, n, ld,
                       cur, nxt);
    return check_launch("kd_order");
  }
};

}  // namespace

extern "C" {

int64_t flooder_index_sort_bytes(int64_t n_pts) {
  if (n_pts < 1) return 0;
  // (the caller does not say yet which key width flooder_index_sort will be given: the larger of the two needs)
  size_t bytes = 0, bytes32 = 0;
  const uint64_t* k = nullptr;
  uint64_t* ko = nullptr;
  const uint32_t* k32 = nullptr;
  uint32_t* ko32 = nullptr;
  uint32_t* vo = nullptr;
  hipError_t e = rocprim::radix_sort_pairs<SortCfg>(nullptr, bytes, k, ko, rocprim::counting_iterator<uint32_t>(0u), vo,
                                           (size_t)n_pts, 0u, 64u, (hipStream_t)0);
  if (e != hipSuccess) return -1;
  e = rocprim::radix_sort_pairs<SortCfg>(nullptr, bytes32, k32, ko32, rocprim::counting_iterator<uint32_t>(0u), vo,
                                (size_t)n_pts, 0u, 32u, (hipStream_t)0);
  if (e != hipSuccess) return -1;
  const size_t own = (size_t)n_pts * 8;   // (flooder_index_sort_zeroed: keys and row numbers of the odd passes)
  bytes = bytes > bytes32 ? bytes : bytes32;
  return (int64_t)(bytes > own ? bytes : own) + 256;
}

int flooder_index_sort(const int64_t* codes, int64_t n_pts, int key_bits, int64_t* codes_sorted, int32_t* order,
                       void* tmp, int64_t tmp_bytes, void* stream) {
  if (n_pts == 0) return FLOODER_OK;
  if (!codes || !codes_sorted || !order || !tmp || n_pts < 0 || n_pts > 0xfffffffeLL || key_bits < 1 || key_bits > 64)
    return fail(FLOODER_E_ARG, "flooder_index_sort: bad argument");
  size_t bytes = (size_t)tmp_bytes;
  hipError_t e;
  if (key_bits <= 32)  // narrow keys: flooder_morton_f32 wrote n uint32 words
    e = rocprim::radix_sort_pairs<SortCfg>(tmp, bytes, reinterpret_cast<const uint32_t*>(codes),
                                  reinterpret_cast<uint32_t*>(codes_sorted), rocprim::counting_iterator<uint32_t>(0u),
                                  reinterpret_cast<uint32_t*>(order), (size_t)n_pts, 0u, (unsigned)key_bits,
                                  (hipStream_t)stream);
  else
    e = rocprim::radix_sort_pairs<SortCfg>(tmp, bytes, reinterpret_cast<const uint64_t*>(codes),
                                  reinterpret_cast<uint64_t*>(codes_sorted), rocprim::counting_iterator<uint32_t>(0u),
                                  reinterpret_cast<uint32_t*>(order), (size_t)n_pts, 0u, (unsigned)key_bits,
                                  (hipStream_t)stream);
  if (e != hipSuccess) return fail(FLOODER_E_LAUNCH, hipGetErrorString(e));
  return check_launch("index_sort");
}

}  // extern "C"

// ---- the same sort without its seven fill launches ------------------------------------------------------------------
// rocprim::radix_sort_pairs resets its state on the way: the digit histograms once, and per 8-bit pass the look-back
// states of the decoupled scan and the ticket counter that hands out block numbers - seven memset launches of ~4 us for
// the three passes of a 24-bit curve code, a sixth of cfg 2's index build.  flooder_index_sort_zeroed launches the
// library's own device functions (rocprim::detail::onesweep_histograms / onesweep_scan_histograms / onesweep_iteration,
// the tuned gfx950 configuration: same kernels, same stable order, same bits) on state the CALLER has zeroed - the
// curve-code kernel does it on its way (flooder_morton_zero_f32's zero_buf) - with one look-back array and one
// ticket per pass.  Narrow keys (<= 32 bits), values = row numbers, fewer than 2^30 rows; anything else: flooder_index_sort.
namespace {
namespace rpd = rocprim::detail;
constexpr rpd::radix_sort_onesweep_config_params OSP =
    rpd::wrapped_radix_sort_onesweep_config<rocprim::default_config, uint32_t, uint32_t>::architecture_config<
        rpd::target_arch::gfx950>::params;
constexpr unsigned OS_BITS = OSP.radix_bits_per_place, OS_RADIX = 1u << OS_BITS;
constexpr unsigned OS_HB = OSP.histogram.block_size, OS_HI = OSP.histogram.items_per_thread;
using OsTicket = rpd::block_id_wrapper<unsigned int, true>;   // (gfx950: blocks take their number from an atomic ticket)
static_assert(sizeof(rpd::onesweep_lookback_state) == sizeof(uint32_t), "one word per look-back state");

// Block shape of a pass by cloud size (measured, tools/time_index.py: whole index build, us)
//                 0.3 M    1 M    4 M    16 M
//   1024 x 16      142     166    279     960     the library's shape for gfx950: 62 blocks for a million keys - a
//   1024 x  8      111     150    284     912     quarter of the CUs, and a look-back chain of 62 long items
//    512 x  8      108     138    312     976
//   1024 x  4      100     142    294     970
// (a second run, 2 M / 8 M keys: 198 / 529 with 512 x 8, 201 / 498 with the library's shape, 182 / 483 with 1024 x 8)
// -> 512 x 8 below 1.5 M keys, 1024 x 8 from there on; the library's shape is level at 4 M and behind everywhere else
// (option "sort_shape": 0 by size, 1 / 2 / 3 = 512 x 8 / the library's / 1024 x 8 forced).
template <unsigned SB, unsigned SI>
struct OsShape {
  static constexpr unsigned sb = SB, si = SI, items = SB * SI;
};
using OsSmall = OsShape<512, 8>;
using OsMid = OsShape<OSP.sort.block_size, OSP.sort.items_per_thread>;
using OsLarge = OsShape<1024, 8>;

__global__ __launch_bounds__(OS_HB) void os_histogram_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ counts,
                                                             uint32_t n, uint32_t full_blocks, unsigned end_bit) {
  rpd::onesweep_histograms<OS_HB, OS_HI, OS_BITS, false>(keys, counts, n, full_blocks, rocprim::identity_decomposer{}, 0u,
                                                         end_bit);
}

__global__ __launch_bounds__(OS_HB) void os_scan_kernel(uint32_t* __restrict__ counts) {
  rpd::onesweep_scan_histograms<OS_HB, OS_BITS>(counts);
}

template <class Shape, class ValuesIn>
__global__ __launch_bounds__(Shape::sb) void os_pass_kernel(const uint32_t* __restrict__ keys_in, uint32_t* __restrict__ keys_out,
                                                            ValuesIn values_in, uint32_t* __restrict__ values_out, uint32_t n,
                                                            uint32_t* offsets, uint32_t* offsets_out,
                                                            rpd::onesweep_lookback_state* states, unsigned bit,
                                                            unsigned bits_now, uint32_t full_blocks, OsTicket ticket) {
  rpd::onesweep_iteration<Shape::sb, Shape::si, OS_BITS, false, OSP.radix_rank_algorithm>(
      keys_in, keys_out, values_in, values_out, n, offsets, offsets_out, states, rocprim::identity_decomposer{}, bit,
      bits_now, full_blocks, ticket);
}

inline unsigned os_places(int key_bits) { return ((unsigned)key_bits + OS_BITS - 1) / OS_BITS; }
inline int os_shape_of(int64_t n) {   // 1 small, 2 mid, 3 large
  if (flooder::g_sort_shape >= 1 && flooder::g_sort_shape <= 3) return flooder::g_sort_shape;
  return n < 1500000 ? 1 : 3;
}
inline uint32_t os_items(int shape) { return shape == 1 ? OsSmall::items : (shape == 2 ? OsMid::items : OsLarge::items); }
inline uint32_t os_blocks(int64_t n, int shape) { return (uint32_t)((n + os_items(shape) - 1) / os_items(shape)); }

template <class Shape>
void os_passes(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t* keys_tmp,
               uint32_t* vals_tmp, uint32_t n, int key_bits, uint32_t* counts, rpd::onesweep_lookback_state* states,
               uint32_t* tickets, uint32_t* offsets_out, hipStream_t st) {
  const unsigned places = os_places(key_bits);
  const uint32_t blocks = (n + Shape::items - 1) / Shape::items;
  const uint32_t full_blocks = n % Shape::items == 0 ? blocks : blocks - 1;
  bool to_output = (places - 1) % 2 == 0;   // (the last pass lands in the output arrays)
  for (unsigned place = 0, bit = 0; place < places; ++place, bit += OS_BITS) {
    const unsigned bits_now = (unsigned)key_bits - bit < OS_BITS ? (unsigned)key_bits - bit : OS_BITS;
    uint32_t* ko = to_output ? keys_out : keys_tmp;
    uint32_t* vo = to_output ? vals_out : vals_tmp;
    OsTicket ticket = OsTicket::create(tickets + place);
    if (place == 0 && vals_in == nullptr)   // (values = row numbers)
      hipLaunchKernelGGL((os_pass_kernel<Shape, rocprim::counting_iterator<uint32_t>>), dim3(blocks), dim3(Shape::sb), 0, st,
                         keys_in, ko, rocprim::counting_iterator<uint32_t>(0u), vo, n, counts, offsets_out, states, bit,
                         bits_now, full_blocks, ticket);
    else if (place == 0)
      hipLaunchKernelGGL((os_pass_kernel<Shape, const uint32_t*>), dim3(blocks), dim3(Shape::sb), 0, st, keys_in, ko, vals_in,
                         vo, n, counts, offsets_out, states, bit, bits_now, full_blocks, ticket);
    else
      hipLaunchKernelGGL((os_pass_kernel<Shape, const uint32_t*>), dim3(blocks), dim3(Shape::sb), 0, st,
                         (const uint32_t*)(to_output ? keys_tmp : keys_out), ko,
                         (const uint32_t*)(to_output ? vals_tmp : vals_out), vo, n, counts + place * OS_RADIX, offsets_out,
                         states + (size_t)place * OS_RADIX * blocks, bit, bits_now, full_blocks, ticket);
    to_output = !to_output;
  }
}
}  // namespace

namespace flooder { int g_sort_shape = 0; }

namespace {
int64_t os_state_words(int64_t n, int key_bits) {
  if (n < 1 || n >= (1LL << 30) || key_bits < 1 || key_bits > 32) return 0;
  const int64_t places = os_places(key_bits);
  // histograms of every digit place | one look-back array per pass | one ticket per pass | the last block's offsets
  // (sized for the smallest block shape: the option may change between the sizing call and the sort)
  return places * OS_RADIX + places * (int64_t)OS_RADIX * os_blocks(n, 1) + places + OS_RADIX;
}

// keys_in / vals_in (NULL: row numbers) are only read; the result lands in keys_out / vals_out; keys_tmp / vals_tmp: n
// words each; state: os_state_words(n, key_bits) words, ZERO when the first launch below starts.
int os_sort_pairs(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t* keys_tmp,
                  uint32_t* vals_tmp, uint32_t n, int key_bits, uint32_t* state, hipStream_t st) {
  const unsigned places = os_places(key_bits);
  const int shape = os_shape_of(n);
  const uint32_t blocks = os_blocks(n, shape);
  uint32_t* counts = state;
  auto* states = reinterpret_cast<rpd::onesweep_lookback_state*>(counts + places * OS_RADIX);
  uint32_t* tickets = counts + places * OS_RADIX + (size_t)places * OS_RADIX * blocks;
  uint32_t* offsets_out = tickets + places;
  {
    const uint32_t hb = (uint32_t)(((int64_t)n + OS_HB * OS_HI - 1) / (OS_HB * OS_HI));
    const uint32_t hfull = n % (OS_HB * OS_HI) == 0 ? hb : hb - 1;
    hipLaunchKernelGGL(os_histogram_kernel, dim3(hb), dim3(OS_HB), 0, st, keys_in, counts, n, hfull, (unsigned)key_bits);
    hipLaunchKernelGGL(os_scan_kernel, dim3(places), dim3(OS_HB), 0, st, counts);
  }
  if (shape == 1)
    os_passes<OsSmall>(keys_in, keys_out, vals_in, vals_out, keys_tmp, vals_tmp, n, key_bits, counts, states, tickets, offsets_out, st);
  else if (shape == 2)
    os_passes<OsMid>(keys_in, keys_out, vals_in, vals_out, keys_tmp, vals_tmp, n, key_bits, counts, states, tickets, offsets_out, st);
  else
    os_passes<OsLarge>(keys_in, keys_out, vals_in, vals_out, keys_tmp, vals_tmp, n, key_bits, counts, states, tickets, offsets_out, st);
  return check_launch("index sort (own launches)");
}
}  // namespace

extern "C" {

int64_t flooder_index_sort_state_words(int64_t n_pts, int key_bits) {   // (0: use flooder_index_sort)
  return os_state_words(n_pts, key_bits);
}

int flooder_index_sort_zeroed(const int64_t* codes, int64_t n_pts, int key_bits, int64_t* codes_sorted, int32_t* order,
                              void* tmp, int64_t tmp_bytes, int32_t* state, void* stream) {
  if (n_pts == 0) return FLOODER_OK;
  if (!codes || !codes_sorted || !order || !tmp || !state || n_pts < 0 || n_pts >= (1LL << 30) || key_bits < 1 ||
      key_bits > 32 || tmp_bytes < 8 * n_pts || (reinterpret_cast<uintptr_t>(tmp) & 3u))
    return fail(FLOODER_E_ARG, "flooder_index_sort_zeroed: bad argument (narrow keys, fewer than 2^30 rows, 8 n bytes of tmp)");
  uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(tmp);
  return os_sort_pairs(reinterpret_cast<const uint32_t*>(codes), reinterpret_cast<uint32_t*>(codes_sorted), nullptr,
                       reinterpret_cast<uint32_t*>(order), keys_tmp, keys_tmp + n_pts, (uint32_t)n_pts, key_bits,
                       reinterpret_cast<uint32_t*>(state), (hipStream_t)stream);
}

int64_t flooder_kd_order_bytes(int64_t n_pts) {
  if (n_pts < 1) return 0;
  if (n_pts > 0x7fffffffLL) return -1;  // (the segment number and a coordinate share 32 key bits)
  size_t bytes = 0;
  const uint32_t* k = nullptr;
  uint32_t* ko = nullptr;
  hipError_t e = rocprim::radix_sort_pairs<SortCfg>(nullptr, bytes, k, ko, k, ko, (size_t)n_pts, 0u, 32u, (hipStream_t)0);
  if (e != hipSuccess) return -1;
  int log_p = 0;
  const int levels = kd_levels(n_pts, log_p);
  const int64_t box_segs = levels > 0 ? ((int64_t)1 << (levels - 1)) : 1;
  // (the sort scratch: what the library call wants, or keys + rows of the odd passes + the state of the own launches)
  const int64_t own = 2 * kd_align(n_pts * 4) + 4 * os_state_words(n_pts, 32);
  return 3 * kd_align(n_pts * 4) + kd_align(box_segs * 64) + ((int64_t)bytes > own ? (int64_t)bytes : own) + 256;
}

int flooder_kd_order_f32(const float* pts, int64_t n_pts, int dim, int ld, int32_t* order, void* tmp, int64_t tmp_bytes,
                         void* stream) {
  if (n_pts == 0) return FLOODER_OK;
  const int64_t need = flooder_kd_order_bytes(n_pts);
  if (!pts || !order || !tmp || n_pts < 0 || ld < dim || need < 0 || tmp_bytes < need)
    return fail(FLOODER_E_ARG, "flooder_kd_order_f32: bad argument");
  int log_p = 0;
  const int levels = kd_levels(n_pts, log_p);
  const int64_t box_segs = levels > 0 ? ((int64_t)1 << (levels - 1)) : 1;
  const int64_t sort_bytes = tmp_bytes - 3 * kd_align(n_pts * 4) - kd_align(box_segs * 64);
  return dispatch_dim<KdOrderOp>(dim, pts, n_pts, ld, reinterpret_cast<uint32_t*>(order), reinterpret_cast<uint8_t*>(tmp),
                                 sort_bytes, (hipStream_t)stream);
}

int64_t flooder_select_grid_bytes(int dim) { return dim == 2 ? SelectCfg<2>::NC : (dim == 3 ? SelectCfg<3>::NC : 0); }

int flooder_box_select_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* box, const float* cloud_box,
                           const float* centers, const float* radii, int64_t n_balls, uint8_t* cell_flags, float* out,
                           int32_t* count, void* stream) {
  if (n_pts == 0) return FLOODER_OK;
  if (!pts || !box || !out || !count || n_pts < 0 || n_pts > 0x7fffffffLL || ld < dim || n_balls < 0 || n_balls > 0x7fffffffLL)
    return fail(FLOODER_E_ARG, "flooder_box_select_f32: bad argument");
  return dispatch_dim<BoxSelectOp>(dim, pts, n_pts, ld, box, cloud_box, centers, radii, n_balls, cell_flags, out, count,
                                   (hipStream_t)stream);
}

int flooder_gather_rows_f32(const float* pts, int64_t n_pts, int dim, int ld, const int32_t* order, float* out,
                            int64_t n_pad, void* stream) {
  if (!pts || !order || !out || n_pts < 1 || n_pad < n_pts || ld < dim)
    return fail(FLOODER_E_ARG, "flooder_gather_rows_f32: bad argument");
  return dispatch_dim<GatherOp>(dim, pts, n_pts, ld, reinterpret_cast<const uint32_t*>(order), out, n_pad,
                                (hipStream_t)stream);
}

}  // extern "C"
