// Host threads for the setup passes (AD structure, tape compiler, KKT plan, symbolic LDLT): a small pool
// made on first use, `parallel_chunks(n, grain, f)` = f(begin, end, chunk) over [0, n) in chunks of at
// least `grain` items, the caller's thread working along; results must not depend on who ran which chunk
// (every pass that uses it writes disjoint ranges or per-chunk buffers concatenated in chunk order, so the
// plans come out the same to the byte with one thread and with sixty-four:
// tests/test_plans_cpu.py::test_plans_do_not_depend_on_the_thread_count).  SLPX_SETUP_THREADS=1: everything
// on the caller's thread.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace slpx {

class SetupPool {
 public:
  static SetupPool& get() {
    static SetupPool pool;
    return pool;
  }
  unsigned threads() const { return m_threads; }
  // For the lifetime of one of these the calling thread runs its parallel regions itself, in order (a side job
  // that should leave the pool to the job on the critical path: the values-only tape beside the full one).
  struct InlineScope {
    bool was;
    InlineScope() : was(t_inside) { t_inside = true; }
    ~InlineScope() { t_inside = was; }
  };

  // runs job(k) for k in [0, count) on the pool's threads and the caller's; returns when all are done.
  // Nested calls (a job that calls run) execute inline on the calling thread.
  void run(unsigned count, const std::function<void(unsigned)>& job) {
    if (count == 0) return;
    if (count == 1 || m_threads <= 1 || t_inside) {
      for (unsigned k = 0; k < count; ++k) job(k);
      return;
    }
    std::unique_lock<std::mutex> outer(m_run_mutex);  // one parallel region at a time
    Run r;
    r.job = &job;
    r.count = count;
    r.pending = count;
    {
      std::lock_guard<std::mutex> lk(m_mutex);
      m_run = &r;
      ++m_generation;
    }
    m_wake.notify_all();
    work(r);
    {
      std::unique_lock<std::mutex> lk(m_mutex);
      m_done.wait(lk, [&] { return r.pending == 0 && r.users == 0; });
      m_run = nullptr;
    }
    // (a job that threw — bad_alloc of a vector sized to the graph — was counted as done by whoever ran it: every
    // worker has let go of `r` by now; the first exception goes to the caller)
    if (r.failure) std::rethrow_exception(r.failure);
  }

 private:
  // one parallel region: lives on run()'s stack until every job is done and every worker has let go of it
  struct Run {
    const std::function<void(unsigned)>* job = nullptr;
    unsigned count = 0, pending = 0, users = 0;  // pending, users: under m_mutex
    std::atomic<unsigned> next{0};
    std::exception_ptr failure;  // the first exception a job threw (under m_mutex)
  };
  SetupPool() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    n = std::min(n, 16u);
    if (const char* env = std::getenv("SLPX_SETUP_THREADS")) n = static_cast<unsigned>(std::max(1, std::atoi(env)));
    m_threads = n;
    for (unsigned i = 1; i < n; ++i) m_workers.emplace_back([this] { loop(); });
  }
  ~SetupPool() {
    {
      std::lock_guard<std::mutex> lk(m_mutex);
      m_stop = true;
      ++m_generation;
    }
    m_wake.notify_all();
    for (auto& t : m_workers) t.join();
  }
  void work(Run& r) {
    InlineScope inside;  // (restored however the loop is left)
    for (;;) {
      const unsigned k = r.next.fetch_add(1, std::memory_order_relaxed);
      if (k >= r.count) break;
      std::exception_ptr thrown;
      try {
        (*r.job)(k);
      } catch (...) {
        thrown = std::current_exception();
      }
      std::lock_guard<std::mutex> lk(m_mutex);
      if (thrown && !r.failure) r.failure = thrown;
      if (--r.pending == 0) m_done.notify_all();
    }
  }
  void loop() {
    unsigned seen = 0;
    for (;;) {
      Run* r;
      {
        std::unique_lock<std::mutex> lk(m_mutex);
        m_wake.wait(lk, [&] { return m_generation != seen; });
        seen = m_generation;
        if (m_stop) return;
        r = m_run;
        if (r == nullptr) continue;
        ++r->users;
      }
      work(*r);
      std::lock_guard<std::mutex> lk(m_mutex);
      if (--r->users == 0) m_done.notify_all();
    }
  }
  static inline thread_local bool t_inside = false;
  unsigned m_threads = 1;
  std::vector<std::thread> m_workers;
  std::mutex m_mutex, m_run_mutex;
  std::condition_variable m_wake, m_done;
  Run* m_run = nullptr;
  unsigned m_generation = 0;
  bool m_stop = false;
};

// f(begin, end, chunk_index) over [0, n): at most one chunk per pool thread x 4, none shorter than `grain`
template <class F>
inline unsigned parallel_chunks(size_t n, size_t grain, F&& f) {
  if (n == 0) return 0;
  const size_t by_grain = std::max<size_t>(1, n / std::max<size_t>(1, grain));
  const unsigned chunks = static_cast<unsigned>(std::min<size_t>(by_grain, 4u * SetupPool::get().threads()));
  if (chunks <= 1) {
    f(size_t{0}, n, 0u);
    return 1;
  }
  // (an exception of a chunk — on whichever thread — is the call's: the first one thrown is rethrown here)
  std::exception_ptr failed;
  std::mutex failed_mutex;
  SetupPool::get().run(chunks, [&](unsigned c) {
    try {
      f(n * c / chunks, n * (c + 1) / chunks, c);
    } catch (...) {
      std::lock_guard<std::mutex> lk(failed_mutex);
      if (!failed) failed = std::current_exception();
    }
  });
  if (failed) std::rethrow_exception(failed);
  return chunks;
}
// the number of chunks parallel_chunks(n, grain, ..) will make (to size per-chunk buffers beforehand)
inline unsigned parallel_chunk_count(size_t n, size_t grain) {
  if (n == 0) return 0;
  const size_t by_grain = std::max<size_t>(1, n / std::max<size_t>(1, grain));
  return static_cast<unsigned>(std::min<size_t>(by_grain, 4u * SetupPool::get().threads()));
}

}  // namespace slpx
