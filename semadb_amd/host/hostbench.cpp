// hostbench.cpp -- serving-shaped measurement of the host side: many submitting threads, each with several
// Search requests outstanding (the shape of a Go server: thousands of request goroutines on a few OS threads),
// through the micro-batcher of semadb_host.hpp into sdb_index_search_batch with SDB_MEM_HOST buffers.  This is
// the SURVEY 8d "wall time including H2D of queries and D2H of results" rate of the drop-in as a Go host would
// drive it.  Built into libsemadb_hostbench.so; bench.py calls it on the index it has just built.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "semadb_host.hpp"

extern "C" {

// queries [n_queries][dim] host memory; every query index is answered at least once when `seconds` allows;
// first_ids [n_queries][limit] (optional) receives the ids of the FIRST answer to each query, for a parity check.
int sdb_hostbench_batcher(sdb_index *h, uint32_t dim, const float *queries, uint64_t n_queries, uint32_t limit,
                          uint32_t search_size, uint32_t threads, uint32_t depth, uint32_t max_batch,
                          uint32_t window_us, uint32_t workers, double seconds, uint64_t *first_ids,
                          uint32_t *first_counts, double *qps, uint64_t *device_batches, uint64_t *served,
                          double *lat_p50_us, double *lat_p99_us) {
  using namespace semadb;
  if (!h || !queries || !n_queries || !threads || !depth || !qps) return 1;
  // a Go server runs GOMAXPROCS = cores OS threads however many request goroutines it has: the same number of
  // outstanding requests on at most one submitting thread per core
  {
    unsigned cores = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // a container's quota, not the host's core count
      long long quota = 0, period = 0;
      if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
        cores = std::min<unsigned>(cores, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
      std::fclose(f);
    }
    const uint64_t outstanding = (uint64_t)threads * depth;
    if (threads > cores) threads = cores, depth = (uint32_t)((outstanding + threads - 1) / threads);
  }
  SearchBatcher batcher(h, dim, max_batch, std::chrono::microseconds(window_us), workers);
  std::atomic<uint64_t> next{0}, completed{0}, errors{0};
  std::atomic<bool> stop{false};
  std::mutex lat_mu;
  std::vector<uint32_t> lat_all;  // per-request time inside the batcher (submit -> answered), microseconds
  auto worker = [&](unsigned) {
    std::vector<uint32_t> lat;
    lat.reserve(1 << 16);
    SearchBatcher::Client client;
    std::vector<SearchBatcher::Request> reqs(depth);
    std::vector<uint64_t> qidx(depth);
    std::vector<uint64_t> ids((size_t)depth * limit);
    std::vector<float> dists((size_t)depth * limit);
    uint64_t submitted = 0;
    auto issue = [&](uint32_t s) {
      const uint64_t t = next.fetch_add(1);
      qidx[s] = t;
      SearchBatcher::Request &r = reqs[s];
      r.vector = queries + (t % n_queries) * dim;
      r.limit = limit, r.search_size = search_size, r.filter = nullptr;
      r.ids = ids.data() + (size_t)s * limit, r.dists = dists.data() + (size_t)s * limit;
      r.count = 0, r.client = &client;
      r.done.store(false, std::memory_order_relaxed);
      batcher.submit(&r);
      submitted++;
    };
    for (uint32_t s = 0; s < depth; s++) issue(s);
    uint64_t harvested = 0;
    while (harvested < submitted) {
      SearchBatcher::waitFor(&client, harvested + 1);
      for (uint32_t s = 0; s < depth; s++) {
        SearchBatcher::Request &r = reqs[s];
        if (r.client && r.done.load(std::memory_order_acquire)) {
          if (r.err) errors++;
          if (lat.size() < (1u << 20)) lat.push_back((uint32_t)std::min<int64_t>((r.t_done_ns - r.t_submit_ns) / 1000, 0xFFFFFFFFll));
          if (qidx[s] < n_queries && first_ids) {  // first pass over the query set: keep the answer
            for (uint32_t i = 0; i < limit; i++) first_ids[qidx[s] * limit + i] = i < r.count ? r.ids[i] : 0;
            if (first_counts) first_counts[qidx[s]] = r.count;
          }
          harvested++;
          completed++;
          r.client = nullptr;
          if (!stop.load(std::memory_order_relaxed)) issue(s);
        }
      }
    }
    // `done` flags are raised before the batcher takes client.mu to count and notify: a request can be harvested
    // while its wake-up is still on its way.  The client (a stack object) may only go away once every wake-up has
    // been delivered, i.e. once the count under the lock has reached what was submitted.
    SearchBatcher::waitFor(&client, submitted);
    std::lock_guard<std::mutex> g(lat_mu);
    lat_all.insert(lat_all.end(), lat.begin(), lat.end());
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < threads; t++) pool.emplace_back(worker, t);
  // run for `seconds`, but at least until every query has been handed out once
  for (;;) {
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (el >= seconds && next.load() >= n_queries) break;
    if (el >= 20 * seconds + 30) break;  // never hang a bench run
  }
  const uint64_t done_at_stop = completed.load();
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  stop = true;
  for (auto &t : pool) t.join();
  *qps = (double)done_at_stop / el;
  if (!lat_all.empty()) {
    std::sort(lat_all.begin(), lat_all.end());
    if (lat_p50_us) *lat_p50_us = lat_all[lat_all.size() / 2];
    if (lat_p99_us) *lat_p99_us = lat_all[(size_t)((double)lat_all.size() * 0.99)];
  }
  if (device_batches) *device_batches = batcher.deviceBatches();
  if (served) *served = batcher.queriesServed();
  return errors.load() ? 2 : 0;
}

// One blocking sdb_index_search_batch(SDB_MEM_HOST) call per batch, from page-locked slabs, timed from C: what a Go
// host pays per call through cgo (no interpreter between the calls).  queries [n_batches][nq][dim] host memory (copied
// into a slab of sdb_host_alloc once, before the clock starts); first_ids / first_counts (optional) receive batch 0's
// answer for a parity check.  pinned == 0: pageable slabs (the driver stages them).
int sdb_hostbench_blocking(sdb_index *h, uint32_t dim, const float *queries, uint32_t n_batches, uint32_t nq, uint32_t limit,
                           uint32_t search_size, uint32_t reps, int pinned, uint64_t *first_ids, uint32_t *first_counts,
                           double *qps, double *ms_per_batch) {
  if (!h || !queries || !n_batches || !nq || !reps || !qps) return 1;
  const size_t qb = (size_t)n_batches * nq * dim * 4, ib = (size_t)nq * limit * 8, db = (size_t)nq * limit * 4, cb = (size_t)nq * 4;
  void *q = nullptr, *ids = nullptr, *d = nullptr, *c = nullptr;
  auto get = [&](size_t bytes, void **p) {
    if (pinned) return sdb_host_alloc(bytes, p) == SDB_OK && *p;
    *p = std::malloc(bytes);
    return *p != nullptr;
  };
  auto drop = [&](void *p) {
    if (!p) return;
    if (pinned) sdb_host_free(p);
    else std::free(p);
  };
  int rc = 0;
  if (!get(qb, &q) || !get(ib, &ids) || !get(db, &d) || !get(cb, &c)) rc = 3;
  if (!rc) {
    std::memcpy(q, queries, qb);
    auto call = [&](uint32_t b) {
      return sdb_index_search_batch(h, nq, (const float *)q + (size_t)b * nq * dim, limit, search_size, nullptr, nullptr,
                                    (uint64_t *)ids, (float *)d, (uint32_t *)c, nullptr, SDB_MEM_HOST, nullptr);
    };
    if (call(0) != SDB_OK) rc = 2;
    if (!rc && first_ids) std::memcpy(first_ids, ids, ib);
    if (!rc && first_counts) std::memcpy(first_counts, c, cb);
    for (uint32_t b = 0; b < std::min(3u, n_batches) && !rc; b++)
      if (call(b) != SDB_OK) rc = 2;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t r = 0; r < reps && !rc; r++)
      for (uint32_t b = 0; b < n_batches && !rc; b++)
        if (call(b) != SDB_OK) rc = 2;
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *qps = (double)reps * n_batches * nq / el;
    if (ms_per_batch) *ms_per_batch = el * 1e3 / ((double)reps * n_batches);
  }
  drop(q), drop(ids), drop(d), drop(c);
  return rc;
}

}  // extern "C"
