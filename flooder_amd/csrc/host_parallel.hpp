// host_parallel.hpp - what the host-parallel routines share (delaunay_nd.cpp, cell_faces.cpp): a pool of threads that
// all run one function (work is handed out through atomic counters), a 64-bit mixer for the lock-free tables, and a
// parallel lexicographic sort of integer rows.  Internal; the C ABI is include/flooder_host.h.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

namespace {

// All threads run the same function; the work inside is handed out through atomic counters.
struct Pool {
  int nt = 1;
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv, cv_done;
  const std::function<void(int)>* job = nullptr;
  alignas(128) std::atomic<long> gen{0};        // bumped for every parallel region: idle threads spin on it for a while
  alignas(128) std::atomic<int> pending{0};
  alignas(128) std::atomic<int> sleepers{0};
  bool stop = false;
  explicit Pool(int n) : nt(n < 1 ? 1 : n) {
    for (int t = 1; t < nt; ++t) th.emplace_back([this, t] { loop(t); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> l(m);
      stop = true;
      gen.fetch_add(1, std::memory_order_release);
    }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
  // The regions of one call follow each other within microseconds (level after level): a thread that finds no work
  // spins on the generation counter for about 100 us before it goes to sleep on the condition variable - waking a
  // hundred sleepers costs more than most regions last.
  void loop(int tid) {
    long seen = 0;
    for (;;) {
      int spins = 0;
      while (gen.load(std::memory_order_acquire) == seen) {
        if (++spins < 4000) {
          __builtin_ia32_pause();
          continue;
        }
        std::unique_lock<std::mutex> l(m);
        sleepers.fetch_add(1, std::memory_order_relaxed);
        cv.wait(l, [&] { return gen.load(std::memory_order_acquire) != seen; });
        sleepers.fetch_sub(1, std::memory_order_relaxed);
      }
      seen = gen.load(std::memory_order_acquire);
      if (stop) return;
      (*job)(tid);
      if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> l(m);
        cv_done.notify_one();
      }
    }
  }
  void run(const std::function<void(int)>& f) {
    if (nt == 1) { f(0); return; }
    {
      std::lock_guard<std::mutex> l(m);
      job = &f;
      pending.store(nt - 1, std::memory_order_relaxed);
      gen.fetch_add(1, std::memory_order_release);
    }
    if (sleepers.load(std::memory_order_relaxed) > 0) cv.notify_all();
    f(0);
    int spins = 0;
    while (pending.load(std::memory_order_acquire) != 0) {
      if (++spins < 20000) {
        __builtin_ia32_pause();
        continue;
      }
      std::unique_lock<std::mutex> l(m);
      cv_done.wait(l, [&] { return pending.load(std::memory_order_acquire) == 0; });
    }
  }
  // f(i0, i1, tid) over [0, n) in chunks; small ranges run on the calling thread
  template <class F>
  void parallel_for(int64_t n, int64_t chunk, F f) {
    if (n <= 0) return;
    if (nt == 1 || n <= chunk) { f((int64_t)0, n, 0); return; }
    alignas(128) std::atomic<int64_t> next{0};
    const std::function<void(int)> body = [&](int tid) {
      for (;;) {
        const int64_t i0 = next.fetch_add(chunk, std::memory_order_relaxed);
        if (i0 >= n) break;
        f(i0, std::min(n, i0 + chunk), tid);
      }
    };
    run(body);
  }
};

inline uint64_t mix64(uint64_t h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
  return h;
}


inline int host_threads(int n_threads) {
  if (n_threads <= 0) {
    cpu_set_t set;
    n_threads = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
    n_threads = std::min(n_threads, 128);
  }
  return std::max(1, n_threads);
}

// Rows of w int32 (ascending ids < base within a row) into lexicographic order, in place: buckets over the range of
// the first two columns (counting pass, scatter), every bucket sorted by one thread.
inline void sort_rows_parallel(Pool& pool, int32_t* rows, int64_t n, int w, int64_t base) {
  if (n < 2) return;
  const int64_t nb = std::max<int64_t>(1, std::min<int64_t>(4096, n / 256));
  const double scale = (double)nb / ((double)base * (double)(w > 1 ? base : 1));
  auto bucket_of = [&](const int32_t* r) {
    const double key = w > 1 ? (double)r[0] * (double)base + (double)r[1] : (double)r[0];
    const int64_t b = (int64_t)(key * scale);
    return b < 0 ? 0 : (b >= nb ? nb - 1 : b);
  };
  const int nt = pool.nt;
  std::vector<int64_t> hist((size_t)nt * (size_t)nb, 0);
  const int64_t per = (n + nt - 1) / nt;
  pool.run([&](int tid) {
    const int64_t a = tid * per, b = std::min(n, a + per);
    int64_t* h = &hist[(size_t)tid * (size_t)nb];
    for (int64_t i = a; i < b; ++i) ++h[bucket_of(rows + i * w)];
  });
  std::vector<int64_t> start((size_t)nb + 1, 0);
  for (int64_t b = 0; b < nb; ++b) {
    int64_t c = 0;
    for (int t = 0; t < nt; ++t) {
      const int64_t v = hist[(size_t)t * (size_t)nb + (size_t)b];
      hist[(size_t)t * (size_t)nb + (size_t)b] = start[(size_t)b] + c;    // where thread t writes its rows of bucket b
      c += v;
    }
    start[(size_t)b + 1] = start[(size_t)b] + c;
  }
  std::vector<int32_t> tmp((size_t)n * (size_t)w);
  pool.run([&](int tid) {
    const int64_t a = tid * per, b = std::min(n, a + per);
    int64_t* h = &hist[(size_t)tid * (size_t)nb];
    for (int64_t i = a; i < b; ++i) {
      const int64_t dst = h[bucket_of(rows + i * w)]++;
      std::memcpy(&tmp[(size_t)dst * (size_t)w], rows + i * w, sizeof(int32_t) * (size_t)w);
    }
  });
  pool.parallel_for(nb, 1, [&](int64_t b0, int64_t b1, int) {
    std::vector<int64_t> idx;
    std::vector<int32_t> buf;
    for (int64_t b = b0; b < b1; ++b) {
      const int64_t lo = start[(size_t)b], cnt = start[(size_t)b + 1] - lo;
      if (cnt == 0) continue;
      idx.resize((size_t)cnt);
      for (int64_t i = 0; i < cnt; ++i) idx[(size_t)i] = lo + i;
      std::sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) {
        return std::lexicographical_compare(&tmp[(size_t)x * (size_t)w], &tmp[(size_t)x * (size_t)w] + w,
                                            &tmp[(size_t)y * (size_t)w], &tmp[(size_t)y * (size_t)w] + w);
      });
      for (int64_t i = 0; i < cnt; ++i)
        std::memcpy(rows + (lo + i) * w, &tmp[(size_t)idx[(size_t)i] * (size_t)w], sizeof(int32_t) * (size_t)w);
    }
  });
}

}  // namespace
