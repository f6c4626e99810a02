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

int flooder_gather_rows_f32(const float* pts, int64_t n_pts, int dim, int ld, const int32_t* order, float* out,
                            int64_t n_pad, void* stream) {
  if (!pts || !order || !out || n_pts < 1 || n_pad < n_pts || ld < dim)
    return fail(FLOODER_E_ARG, "flooder_gather_rows_f32: bad argument");
  return dispatch_dim<GatherOp>(dim, pts, n_pts, ld, reinterpret_cast<const uint32_t*>(order), out, n_pad,
                                (hipStream_t)stream);
}

}  // extern "C"
