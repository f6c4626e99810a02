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
  return (int64_t)(bytes > bytes32 ? bytes : bytes32) + 256;
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
