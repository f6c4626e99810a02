// Diagnostic: rocprim::radix_sort_pairs on curve-code-like keys - key width / used bits vs time.
// build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/sort_bench.hip -o /tmp/sort_bench && /tmp/sort_bench
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>

template <class K>
void run(size_t n, unsigned bits, const char* name) {
  std::vector<K> h(n);
  std::mt19937_64 rng(1);
  for (auto& x : h) x = (K)(rng() & ((bits >= 64 ? ~0ull : ((1ull << bits) - 1))));
  K *k_in, *k_out; uint32_t* v_out;
  hipMalloc(&k_in, n * sizeof(K)); hipMalloc(&k_out, n * sizeof(K)); hipMalloc(&v_out, n * 4);
  hipMemcpy(k_in, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
  size_t bytes = 0;
  rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, rocprim::counting_iterator<uint32_t>(0u), v_out, n, 0u, bits);
  void* tmp; hipMalloc(&tmp, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs(tmp, bytes, k_in, k_out, rocprim::counting_iterator<uint32_t>(0u), v_out, n, 0u, bits);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) rocprim::radix_sort_pairs(tmp, bytes, k_in, k_out, rocprim::counting_iterator<uint32_t>(0u), v_out, n, 0u, bits);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-10s n=%zu bits=%u : %.1f us per sort (tmp %zu B)\n", name, n, bits, ms / 20 * 1e3, bytes);
  hipFree(k_in); hipFree(k_out); hipFree(v_out); hipFree(tmp);
}

int main() {
  for (size_t n : {1000000ul, 16000000ul}) {
    run<uint64_t>(n, 63, "u64");
    run<uint64_t>(n, 36, "u64");
    run<uint64_t>(n, 30, "u64");
    run<uint32_t>(n, 32, "u32");
    run<uint32_t>(n, 30, "u32");
    run<uint32_t>(n, 24, "u32");
  }
  return 0;
}
