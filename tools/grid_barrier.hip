// What does a grid-wide barrier cost on this GPU, against the gap between two dependent kernel launches?  (The batched
// landmark selection is ~240 dependent launches of ~9 us with ~3 us between them: would ONE persistent launch with a
// device-side barrier per batch be faster?)   hipcc -O3 --offload-arch=gfx950 tools/grid_barrier.hip -o /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void barrier_kernel(unsigned* ctr, int rounds, int work) {
  __shared__ float sink;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    for (int i = 0; i < work; ++i) acc = acc * 1.0001f + (float)threadIdx.x;   // a little work between barriers
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned target = (unsigned)(r + 1) * gridDim.x;
      atomicAdd(ctr, 1u);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __threadfence();
    }
    __syncthreads();
  }
  if (acc == 12345.f) sink = acc;
}

__global__ void tiny_kernel(unsigned* ctr) {
  if (threadIdx.x == 0 && blockIdx.x == 0) ctr[1] += 1;
}

int main() {
  unsigned* ctr;
  hipMalloc(&ctr, 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int grid : {64, 256, 512, 1024}) {
    for (int threads : {64, 256}) {
      hipMemset(ctr, 0, 64);
      const int rounds = 2000;
      hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(threads), 0, 0, ctr, 10, 0);   // warm-up
      hipDeviceSynchronize();
      hipMemset(ctr, 0, 64);
      hipEventRecord(a);
      hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(threads), 0, 0, ctr, rounds, 0);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      printf("grid %4d x %3d threads: %.2f us per grid barrier\n", grid, threads, ms * 1e3 / rounds);
    }
  }
  // dependent launches: N tiny kernels back to back on one stream
  for (int rep = 0; rep < 2; ++rep) {
    const int n = 2000;
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1024), dim3(256), 0, 0, ctr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%d dependent launches of an empty 1024 x 256 kernel: %.2f us each\n", n, ms * 1e3 / n);
  }
  return 0;
}
