// view_mutex.h -- the lock around an index's committed view (index.h sdb_index::view_mu).  Free of HIP so that it
// compiles -- and is stressed under ThreadSanitizer -- on a box without a GPU (tests/host/test_concurrency.cpp).
#pragma once
#include <atomic>
#include <shared_mutex>
#include <thread>

namespace sdb {

// The lock around an index's committed view.  Searches take it shared for the few microseconds between reading the
// view and recording their event; commit / compact / reserve / attach_pq take it exclusively.  glibc's rwlock prefers
// readers, and two always-busy batcher workers can keep a shared lock held back to back for as long as they like --
// a writer would starve.  So a waiting writer raises a flag and new readers stand aside until it has had its turn.
class ViewMutex {
  std::shared_mutex m_;
  std::atomic<int> writers_{0};

 public:
  void lock() {
    writers_.fetch_add(1, std::memory_order_acq_rel);
    m_.lock();
    writers_.fetch_sub(1, std::memory_order_acq_rel);
  }
  void unlock() { m_.unlock(); }
  void lock_shared() {
    while (writers_.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    m_.lock_shared();
  }
  void unlock_shared() { m_.unlock_shared(); }
};

}  // namespace sdb
