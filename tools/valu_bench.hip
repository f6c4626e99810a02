// valu_bench.hip - measures fp32 VALU issue rates on gfx950 that bound the coverage sweep:
// v_fma_f32, v_pk_fma_f32, v_pk_add_f32 with an SGPR operand, and the sweep's distance body.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/valu_bench tools/valu_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NACC = 16;

__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a, float b) {
  float acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pkfma(float* out, int iters, float a, float b) {
  v2f acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v2f{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
  const v2f a2 = {a, a * 1.0001f}, b2 = {b, b * 1.0001f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_elementwise_fma(acc[i], a2, b2);
  }
  v2f s = {0, 0}; for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// distance body: 8 samples per lane against wave-uniform candidates (SGPR operands)
template <bool PACKED>
__global__ __launch_bounds__(256) void k_dist(const float* __restrict__ cand, float* out, int n_groups) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(4))) v4f* cptr;
  float p[8][3];
  for (int i = 0; i < 8; ++i) for (int k = 0; k < 3; ++k) p[i][k] = (threadIdx.x * 8 + i) * 1e-3f + k;
  cptr cp = (cptr)(uintptr_t)cand;
  float best[8];
  if constexpr (PACKED) {
    v2f P[4][3], B[4];
    for (int i = 0; i < 4; ++i) { for (int k = 0; k < 3; ++k) P[i][k] = v2f{p[2*i][k], p[2*i+1][k]}; B[i] = v2f{1e30f, 1e30f}; }
    for (int g = 0; g < n_groups; ++g) {
      v4f c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = cp[(g & 63) * 8 + u];
#pragma unroll
      for (int u = 0; u < 8; u += 2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v2f da, db;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            v2f ta = P[i][k] - v2f{c[u][k], c[u][k]};
            v2f tb = P[i][k] - v2f{c[u+1][k], c[u+1][k]};
            if (k == 0) { da = ta * ta; db = tb * tb; }
            else { da = __builtin_elementwise_fma(ta, ta, da); db = __builtin_elementwise_fma(tb, tb, db); }
          }
          B[i].x = __builtin_fminf(B[i].x, __builtin_fminf(da.x, db.x));
          B[i].y = __builtin_fminf(B[i].y, __builtin_fminf(da.y, db.y));
        }
    }
    for (int i = 0; i < 4; ++i) { best[2*i] = B[i].x; best[2*i+1] = B[i].y; }
  } else {
    for (int i = 0; i < 8; ++i) best[i] = 1e30f;
    for (int g = 0; g < n_groups; ++g) {
      v4f c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = cp[(g & 63) * 8 + u];
#pragma unroll
      for (int u = 0; u < 8; u += 2)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float da, db;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float ta = p[i][k] - c[u][k], tb = p[i][k] - c[u+1][k];
            if (k == 0) { da = ta * ta; db = tb * tb; }
            else { da = __builtin_fmaf(ta, ta, da); db = __builtin_fmaf(tb, tb, db); }
          }
          best[i] = __builtin_fminf(best[i], __builtin_fminf(da, db));
        }
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += best[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F launch, int reps) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) launch();
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * 4));
  float* cand; CHECK(hipMalloc(&cand, 64 * 8 * 16));
  float h[64 * 8 * 4]; for (int i = 0; i < 64 * 8 * 4; ++i) h[i] = (i % 97) * 0.01f;
  CHECK(hipMemcpy(cand, h, sizeof(h), hipMemcpyHostToDevice));
  const int iters = 4096;
  for (int wpb : {1, 2, 4, 8}) {   // blocks per CU (x4 waves each) -> waves per SIMD
    const int grid = prop.multiProcessorCount * wpb;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); }, 5);
    double ops = (double)grid * 256 * iters * 4 * NACC;
    printf("waves/SIMD %d  v_fma_f32     %8.3f ms  %7.2f TFLOP/s  (%.3f Tinstr-lanes/s)\n", wpb, ms, 2 * ops / ms * 1e-9, ops / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); }, 5);
    printf("waves/SIMD %d  v_pk_fma_f32  %8.3f ms  %7.2f TFLOP/s\n", wpb, ms, 4 * ops / ms * 1e-9);
    const int ng = 2048;
    double pairs = (double)grid * 256 * 8 * ng * 8;
    ms = time_ms([&] { hipLaunchKernelGGL(k_dist<false>, dim3(grid), dim3(256), 0, 0, cand, out, ng); }, 5);
    printf("waves/SIMD %d  dist plain    %8.3f ms  %7.3f Tpair/s  (%.2f TFLOP/s at 10 flop/pair)\n", wpb, ms, pairs / ms * 1e-9, 10 * pairs / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_dist<true>, dim3(grid), dim3(256), 0, 0, cand, out, ng); }, 5);
    printf("waves/SIMD %d  dist packed   %8.3f ms  %7.3f Tpair/s  (%.2f TFLOP/s at 10 flop/pair)\n", wpb, ms, pairs / ms * 1e-9, 10 * pairs / ms * 1e-9);
  }
  return 0;
}
