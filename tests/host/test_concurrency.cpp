// test_concurrency.cpp -- stress of the product's CONCURRENT host code on the CPU box, against the mock C ABI of
// mock_sdb.cpp, meant to be built with -fsanitize=thread and with -fsanitize=address,undefined
// (tests/test_host_concurrency.py does both and demands zero reports).  The reference's contract for this path is
// concurrency: one goroutine per request calls IndexVamana.Search under the shard's RLock
// (shard/index/search.go:53-87, shard/cache/manager.go:159-181) and ClusterNode.SearchPoints fans every request out
// to the shards at once (cluster/actions.go:316-351); its own suite runs under the race detector (.vscode/tasks.json:7).
//
//   (a) SearchBatcher   semadb_host.hpp: 64 submitters, mixed (limit, searchSize) so slabs are recycled under other
//                       tags, filters (lists and bitmaps), cancellations, back-pressure (every slab in use), windows
//                       that expire while the last slot is taken, destruction with requests outstanding, pageable slabs
//   (b) the turnstile   csrc/turnstile.h (the code cluster.hip runs): 8 ranks x 8 threads, tickets arriving in any
//                       order, a ticket that is never presented, skips (also of a ticket that is waiting), ring slots
//   (c) the exchange    GpuFanout + the collective skeleton over turnstile.h: concurrent requests, a failing shard, a
//                       rank that never arrives (deadline, withdrawal, skip), destruction
//   (d) IndexVamana     searches from many threads while writers run InsertUpdateDelete
// Every answer is checked against the request that asked for it (mock_expect).
#include <algorithm>
#include <atomic>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../semadb_amd/csrc/turnstile.h"
#include "../../semadb_amd/csrc/view_mutex.h"
#include "../../semadb_amd/host/semadb_host.hpp"
#include "mock_sdb.h"

using namespace semadb;

static std::atomic<int> g_failures{0};
#define CHECK(cond, ...)                                         \
  do {                                                           \
    if (!(cond)) {                                               \
      if (g_failures.fetch_add(1) < 20) {                        \
        std::fprintf(stderr, "FAIL %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
        std::fprintf(stderr, __VA_ARGS__);                       \
        std::fprintf(stderr, "\n");                              \
      }                                                          \
    }                                                            \
  } while (0)

static const uint32_t kDim = 8;
static double g_scale = 1.0;  // SDB_STRESS_SCALE: how long each stress runs (1 = the CPU suite's few seconds)
static int scaled(int n) { return std::max(1, (int)(n * g_scale)); }

static sdb_index *new_index() {
  sdb_index_params p{};
  p.dim = kDim, p.metric = SDB_METRIC_EUCLIDEAN, p.search_size = 75, p.degree_bound = 64, p.alpha = 1.2f;
  sdb_index *h = nullptr;
  if (sdb_index_create(&p, &h) != SDB_OK) std::abort();
  std::vector<float> start(kDim, 0.5f);
  sdb_index_set_start(h, start.data(), SDB_MEM_HOST);
  return h;
}

// one request of the stress: its query encodes who asked (serial), so the answer can be checked
struct Ask {
  SearchBatcher::Request r;
  std::vector<float> q;
  std::vector<uint64_t> ids;
  std::vector<float> dists;
  SearchBatcher::Filter filter;
  bool filtered = false;
  void make(uint32_t serial, uint32_t who, uint32_t limit, uint32_t L, SearchBatcher::Client *c) {
    q.assign(kDim, 0.f);
    q[0] = (float)(serial & 0xFFFFFF), q[1] = (float)who, q[2] = (float)(serial >> 24);
    ids.assign(limit, ~0ull), dists.assign(limit, -1.f);
    r.vector = q.data(), r.limit = limit, r.search_size = L;
    r.filter = filtered ? &filter : nullptr;
    r.ids = ids.data(), r.dists = dists.data(), r.count = 0xFFFFFFFFu, r.err = Error();
    r.client = c;
    r.done.store(false, std::memory_order_relaxed);
  }
  void verify(const char *what) {
    CHECK(r.done.load(std::memory_order_acquire), "%s: request not done", what);
    if (r.search_size == MOCK_FAILING_SEARCH_SIZE) {
      CHECK((bool)r.err, "%s: a failing batch must report its error", what);
      return;
    }
    if (r.err) {
      CHECK(false, "%s: unexpected error: %s", what, r.err.msg.c_str());
      return;
    }
    std::vector<uint64_t> f(filter.begin(), filter.end()), e_ids(r.limit);
    std::vector<float> e_d(r.limit);
    uint32_t e_c = 0;
    const uint64_t none = 0;  // an EMPTY filter is still a filter: no result
    mock_expect(q.data(), kDim, r.limit, filtered ? (f.empty() ? &none : f.data()) : nullptr, f.size(), e_ids.data(), e_d.data(), &e_c);
    CHECK(r.count == e_c, "%s: count %u, expected %u (limit %u, L %u) filtered %d cancelled %d q0 %g", what, r.count, e_c, r.limit, r.search_size, (int)filtered, (int)r.cancelled.load(), (double)q[0]);
    for (uint32_t j = 0; j < std::min(r.count, e_c); j++) {
      CHECK(ids[j] == e_ids[j], "%s: id[%u] = %" PRIu64 ", expected %" PRIu64 " -- the answer of another request?", what, j,
            ids[j], e_ids[j]);
      CHECK(dists[j] == e_d[j], "%s: dist[%u]", what, j);
    }
  }
};

struct Mix {  // what the submitters draw their parameters from
  bool filters = true, failing = true, odd_params = true, big_filters = true;
};

// `threads` submitters, each `rounds` times: up to `depth` requests outstanding, then wait for them and check
static void hammer(SearchBatcher &b, int threads, int rounds, int depth, const Mix &mix, const char *what,
                   std::atomic<int> *phase = nullptr) {
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++)
    pool.emplace_back([&, t] {
      std::minstd_rand rng((unsigned)(t * 7919 + 13));
      SearchBatcher::Client client;
      uint64_t submitted = 0;
      std::vector<Ask> asks((size_t)depth);
      for (int it = 0; it < rounds; it++) {
        const int n = 1 + (int)(rng() % (unsigned)depth);
        for (int s = 0; s < n; s++) {
          Ask &a = asks[(size_t)s];
          uint32_t limit = 10, L = 75;
          if (phase && (phase->load(std::memory_order_relaxed) & 1)) limit = 5, L = 50;  // the prevailing parameters move
          const unsigned dice = rng() % 100;
          a.filtered = false;
          a.filter.clear();
          if (mix.odd_params && dice < 8) limit = 1 + rng() % 75, L = std::max<uint32_t>(limit, 25 + rng() % 51);
          else if (mix.odd_params && dice < 10) limit = 200, L = 300;  // beyond the result slabs: the queue serves it
          else if (mix.failing && dice < 13) L = MOCK_FAILING_SEARCH_SIZE;
          else if (mix.filters && dice < 25) {
            a.filtered = true;
            const unsigned m = rng() % 30;
            for (unsigned k = 0; k < m; k++) a.filter.insert(2 + rng() % 5000);
          } else if (mix.big_filters && dice < 26) {  // dense: goes up as a bitmap
            a.filtered = true;
            const uint64_t f0 = 2 + rng() % 1000;
            for (uint64_t k = 0; k < 4200; k++) a.filter.insert(f0 + k + (k % 7 == 0));
          }
          if (L == MOCK_FAILING_SEARCH_SIZE && limit > L) limit = 10;
          a.make((uint32_t)(it * depth + s) | (uint32_t)t << 24, (uint32_t)t, limit, L, &client);
          if (rng() % 16 == 0) a.r.cancelled.store(true);  // context cancellation: answered all the same
          b.submit(&a.r);
          submitted++;
        }
        SearchBatcher::waitFor(&client, submitted);
        for (int s = 0; s < n; s++) asks[(size_t)s].verify(what);
      }
    });
  for (auto &t : pool) t.join();
}

// ---------------------------------------------------------------------------------------------------------------
static void test_batcher_mixed() {
  mock_set_latency_us(50, 300);
  sdb_index *h = new_index();
  {
    SearchBatcher b(h, kDim, 16, std::chrono::microseconds(50), 2);
    std::atomic<int> phase{0};
    std::atomic<bool> stop{false};
    std::thread flipper([&] {
      while (!stop.load()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(15));
        phase.fetch_add(1);
      }
    });
    hammer(b, 64, scaled(40), 4, Mix{}, "mixed", &phase);
    stop = true;
    flipper.join();
    CHECK(b.queriesServed() > 0 && b.deviceBatches() > 0, "nothing served");
    std::printf("  mixed: %" PRIu64 " queries in %" PRIu64 " device batches\n", b.queriesServed(), b.deviceBatches());
  }
  sdb_index_destroy(h);
}

static void test_batcher_backpressure() {
  mock_set_latency_us(1500, 2500);  // slow device, tiny batches, one worker: every slab is in use most of the time
  sdb_index *h = new_index();
  {
    SearchBatcher b(h, kDim, 4, std::chrono::microseconds(100), 1);
    Mix m;
    m.filters = m.failing = m.odd_params = m.big_filters = false;
    hammer(b, 32, scaled(12), 3, m, "back-pressure");
    CHECK(b.backpressureWaits() > 0, "the stress never saw every slab in use (waits %" PRIu64 ")", b.backpressureWaits());
    std::printf("  back-pressure: %" PRIu64 " submits waited for a slab\n", b.backpressureWaits());
  }
  sdb_index_destroy(h);
}

static void test_batcher_window_race() {
  mock_set_latency_us(5, 40);  // the window ends while submitters are taking the last slots
  sdb_index *h = new_index();
  for (int w : {0, 1, 20}) {
    SearchBatcher b(h, kDim, 8, std::chrono::microseconds(w), 3);
    Mix m;
    m.big_filters = false;
    hammer(b, 16, scaled(150), 2, m, "window");
  }
  sdb_index_destroy(h);
}

static void test_batcher_pageable() {
  mock_set_latency_us(20, 100);
  mock_fail_host_alloc(1);  // no pinned memory to be had: malloc'ed slabs, same answers
  sdb_index *h = new_index();
  {
    SearchBatcher b(h, kDim, 16, std::chrono::microseconds(50), 2);
    hammer(b, 8, scaled(40), 4, Mix{}, "pageable");
  }
  mock_fail_host_alloc(0);
  sdb_index_destroy(h);
}

// the serving shape of hostbench.cpp: a client keeps `depth` requests outstanding, harvests by polling the done flags and
// re-issues a slot the moment it is answered -- while the worker that answered it is still on its way to the client's
// counter
static void test_batcher_poll_and_reissue() {
  mock_set_latency_us(20, 120);
  sdb_index *h = new_index();
  {
    SearchBatcher b(h, kDim, 32, std::chrono::microseconds(100), 3);
    std::vector<std::thread> pool;
    for (int t = 0; t < 8; t++)
      pool.emplace_back([&, t] {
        const int depth = 24, total = scaled(600);
        SearchBatcher::Client client;
        std::vector<Ask> asks((size_t)depth);
        uint64_t submitted = 0, harvested = 0;
        auto issue = [&](int s) {
          asks[(size_t)s].make((uint32_t)submitted | (uint32_t)t << 24, (uint32_t)t, 10, 75, &client);
          b.submit(&asks[(size_t)s].r);
          submitted++;
        };
        for (int s = 0; s < depth; s++) issue(s);
        while (harvested < submitted) {
          SearchBatcher::waitFor(&client, harvested + 1);
          for (int s = 0; s < depth; s++) {
            Ask &a = asks[(size_t)s];
            if (a.r.client && a.r.done.load(std::memory_order_acquire)) {
              a.verify("poll");
              harvested++;
              a.r.client = nullptr;
              if ((int)submitted < total) issue(s);
            }
          }
        }
        SearchBatcher::waitFor(&client, submitted);  // every wake-up delivered before the client leaves the stack
      });
    for (auto &t : pool) t.join();
  }
  sdb_index_destroy(h);
}

// the batcher goes away while requests are outstanding: every one of them is answered (or refused) first
static void test_batcher_destructor() {
  mock_set_latency_us(300, 900);
  sdb_index *h = new_index();
  for (int round = 0; round < scaled(6); round++) {
    const int threads = 12, per = 9;
    std::vector<SearchBatcher::Client> clients((size_t)threads);
    std::vector<std::vector<Ask>> asks;
    for (int t = 0; t < threads; t++) asks.emplace_back((size_t)per);
    {
      SearchBatcher b(h, kDim, 8, std::chrono::microseconds(200), 2);
      std::vector<std::thread> pool;
      for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
          for (int s = 0; s < per; s++) {
            Ask &a = asks[(size_t)t][(size_t)s];
            a.filtered = (s % 4 == 3);
            if (a.filtered) a.filter = {3, 5, 8, 13};
            a.make((uint32_t)(round * 1000 + s) | (uint32_t)t << 24, (uint32_t)t, s % 3 ? 10 : 7, 75, &clients[(size_t)t]);
            b.submit(&a.r);
          }
        });
      for (auto &t : pool) t.join();  // every submit() has returned; nothing has been waited for
    }                                 // ~SearchBatcher under load
    for (int t = 0; t < threads; t++) {
      SearchBatcher::waitFor(&clients[(size_t)t], (uint64_t)per);
      for (auto &a : asks[(size_t)t]) {
        CHECK(a.r.done.load(), "destructor left a request unanswered");
        if (a.r.err) CHECK(a.r.err.msg == "batcher stopped", "unexpected error: %s", a.r.err.msg.c_str());
        else a.verify("destructor");
      }
    }
  }
  sdb_index_destroy(h);
}

// ---------------------------------------------------------------------------------------------------------------
// (b) the turnstile itself
struct RawSlot : sdb::SlotState {
  std::atomic<int> holder{-1};
};
struct RawRank : sdb::OrderState {
  RawSlot ring[4];
  std::vector<uint64_t> order;  // tickets in the order they entered (under the lock)
};

static void test_turnstile_order() {
  const int kRanks = 8, kThreads = 8, kTickets = scaled(400);
  std::vector<std::unique_ptr<RawRank>> ranks;
  for (int r = 0; r < kRanks; r++) {
    ranks.emplace_back(new RawRank());
    ranks.back()->rank = r;
  }
  std::vector<std::thread> pool;
  std::atomic<int> max_busy{0};
  for (int r = 0; r < kRanks; r++) {
    // deal the tickets to this rank's threads at random; a thread presents its own in ascending order (a fan-out
    // worker pops its queue front to back), the threads race each other
    std::vector<std::vector<uint64_t>> mine((size_t)kThreads);
    std::minstd_rand rng((unsigned)(r + 1) * 101);
    for (int t = 1; t <= kTickets; t++) mine[rng() % kThreads].push_back((uint64_t)t);
    for (int th = 0; th < kThreads; th++)
      pool.emplace_back([&, r, th, tickets = mine[(size_t)th]] {
        RawRank &c = *ranks[(size_t)r];
        std::minstd_rand rr((unsigned)(r * 64 + th));
        for (uint64_t ticket : tickets) {
          if (rr() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rr() % 200));
          std::unique_lock<std::mutex> lk(*c.mu);
          sdb::Turn turn{&c, ticket};
          const int rc = turn.enter(lk);
          CHECK(rc == SDB_OK, "ticket %" PRIu64 " refused: %s", ticket, sdb_last_error());
          if (rc) continue;
          c.order.push_back(ticket);
          RawSlot *s = sdb::take_slot(&c, c.ring, lk);
          int expected = -1;
          CHECK(s->holder.compare_exchange_strong(expected, th), "ring slot handed to two calls at once");
          s->busy = true;
          int busy = 0;
          for (auto &x : c.ring) busy += x.busy;
          int m = max_busy.load();
          while (busy > m && !max_busy.compare_exchange_weak(m, busy)) {
          }
          turn.pass();  // the next ticket may enter while this call "waits for the device"
          lk.unlock();
          std::this_thread::sleep_for(std::chrono::microseconds(rr() % 120));
          lk.lock();
          s->holder.store(-1);
          s->busy = false;
          c.cv->notify_all();
        }
      });
  }
  for (auto &t : pool) t.join();
  for (auto &c : ranks) {
    CHECK((int)c->order.size() == kTickets, "rank %d: %zu tickets entered", c->rank, c->order.size());
    for (size_t i = 0; i < c->order.size(); i++)
      if (c->order[i] != i + 1) {
        CHECK(false, "rank %d: entry %zu was ticket %" PRIu64, c->rank, i, c->order[i]);
        break;
      }
    CHECK(c->next_ticket == (uint64_t)kTickets + 1, "rank %d: next ticket %" PRIu64, c->rank, c->next_ticket);
  }
  CHECK(max_busy.load() > 1 && max_busy.load() <= 4, "calls in flight per rank: %d", max_busy.load());
}

static void test_turnstile_missing_ticket() {
  RawRank c;
  c.deadline_ms = 150;
  auto present = [&](uint64_t ticket) {
    std::unique_lock<std::mutex> lk(*c.mu);
    sdb::Turn turn{&c, ticket};
    const int rc = turn.enter(lk);
    if (rc == SDB_OK) c.order.push_back(ticket);
    return rc;
  };
  CHECK(present(1) == SDB_OK && present(2) == SDB_OK, "first tickets");
  CHECK(present(1) == SDB_ERR_INVALID, "a ticket enters once");
  // ticket 3 is never presented: its successors give up after the deadline having done nothing ...
  std::vector<std::thread> pool;
  std::atomic<int> timed_out{0};
  for (uint64_t t : {4, 5, 6})
    pool.emplace_back([&, t] {
      if (present(t) == SDB_ERR_STATE) timed_out++;
    });
  for (auto &t : pool) t.join();
  pool.clear();
  CHECK(timed_out.load() == 3, "successors of a missing ticket must time out (%d did)", timed_out.load());
  CHECK(c.next_ticket == 3 && c.order.size() == 2, "a call that timed out did nothing");
  // ... and enter, in order, once the fan-out declares it lost -- presented again from racing threads
  for (uint64_t t : {6, 5, 4})
    pool.emplace_back([&, t] { CHECK(present(t) == SDB_OK, "ticket %" PRIu64 " after the skip: %s", t, sdb_last_error()); });
  std::this_thread::sleep_for(std::chrono::milliseconds(5));
  {
    std::lock_guard<std::mutex> g(*c.mu);
    CHECK(sdb::skip_unentered(&c, 3) == SDB_OK, "skip");
  }
  for (auto &t : pool) t.join();
  pool.clear();
  CHECK((c.order == std::vector<uint64_t>{1, 2, 4, 5, 6}), "order after the skip");
  // skips ahead of the turn are passed over when their turn comes
  {
    std::lock_guard<std::mutex> g(*c.mu);
    CHECK(sdb::skip_unentered(&c, 8) == SDB_OK && sdb::skip_unentered(&c, 9) == SDB_OK, "skip ahead");
    CHECK(sdb::skip_unentered(&c, 2) == SDB_ERR_INVALID, "a ticket that has entered cannot be skipped");
  }
  CHECK(present(8) == SDB_ERR_INVALID, "a skipped ticket is refused");
  CHECK(present(7) == SDB_OK, "ticket 7");
  CHECK(c.next_ticket == 10, "the turn passed over 8 and 9 (next %" PRIu64 ")", c.next_ticket);
  // a ticket that is skipped WHILE it waits must not wait out the deadline for a turn that will never be its own
  c.deadline_ms = 0;  // for ever: before the fix this wedged
  std::atomic<int> rc11{-1};
  std::thread waiter([&] { rc11 = present(11); });
  std::this_thread::sleep_for(std::chrono::milliseconds(5));
  {
    std::lock_guard<std::mutex> g(*c.mu);
    CHECK(sdb::skip_unentered(&c, 11) == SDB_OK, "skip of a waiting ticket");
  }
  CHECK(present(10) == SDB_OK, "ticket 10");
  waiter.join();
  CHECK(rc11.load() == SDB_ERR_INVALID, "the skipped waiter is refused, not wedged (rc %d)", rc11.load());
  CHECK(c.next_ticket == 12, "next %" PRIu64, c.next_ticket);
}

// the lock around an index's committed view (index.h view_mu): searches hold it shared back to back from several
// batcher workers; a writer (commit, compact, table growth) must still get its turn -- and see a consistent view
static void test_view_mutex_writer_gets_its_turn() {
  sdb::ViewMutex mu;
  struct View {
    uint64_t a = 0, b = 0;  // a writer keeps a == b
  } view;
  std::atomic<bool> stop{false};
  std::atomic<uint64_t> reads{0}, torn{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < 8; t++)  // always-busy readers: as soon as one lets go another holds it
    pool.emplace_back([&] {
      while (!stop.load(std::memory_order_relaxed)) {
        std::shared_lock<sdb::ViewMutex> g(mu);
        const View v = view;
        if (v.a != v.b) torn++;
        reads++;
      }
    });
  const int commits = scaled(300);
  int64_t worst_ns = 0;
  std::thread writer([&] {
    for (int i = 0; i < commits; i++) {
      const auto t0 = std::chrono::steady_clock::now();
      {
        std::unique_lock<sdb::ViewMutex> g(mu);
        worst_ns = std::max<int64_t>(worst_ns, std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
        view.a++;
        std::this_thread::yield();
        view.b++;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  });
  writer.join();
  stop = true;
  for (auto &t : pool) t.join();
  CHECK(torn.load() == 0, "a reader saw half a commit");
  CHECK(view.a == (uint64_t)commits && view.b == view.a, "commits lost");
  CHECK(reads.load() > (uint64_t)commits, "the readers never ran");
  CHECK(worst_ns < 2000000000ll, "a writer waited %.1f ms behind readers", worst_ns / 1e6);
  std::printf("  view lock: %d commits among %" PRIu64 " reads, the longest wait of a writer %.2f ms\n", commits, reads.load(), worst_ns / 1e6);
}

// ---------------------------------------------------------------------------------------------------------------
// (c) the exchange through the C ABI: what GpuFanout drives
static void expected_merge(const float *q, int shards, uint32_t limit, std::vector<uint64_t> *ids, std::vector<float> *d,
                           std::vector<uint32_t> *sh) {
  uint32_t per = 0;
  sdb_shard_limit(limit, (uint32_t)shards, 75, &per);
  struct E {
    float d;
    uint32_t s;
    uint64_t id;
  };
  std::vector<E> all;
  std::vector<uint64_t> b_ids(per);
  std::vector<float> b_d(per);
  uint32_t cnt = 0;
  mock_expect(q, kDim, per, nullptr, 0, b_ids.data(), b_d.data(), &cnt);
  for (int r = 0; r < shards; r++)
    for (uint32_t j = 0; j < cnt; j++) all.push_back({b_d[j] + (float)r / 16.0f, (uint32_t)r, b_ids[j] + ((uint64_t)r << 56)});
  std::sort(all.begin(), all.end(), [](const E &x, const E &y) { return x.d != y.d ? x.d < y.d : x.s != y.s ? x.s < y.s : x.id < y.id; });
  all.resize(std::min<size_t>(all.size(), limit));
  ids->clear(), d->clear(), sh->clear();
  for (auto &e : all) ids->push_back(e.id), d->push_back(e.d), sh->push_back(e.s);
}

static void test_fanout() {
  mock_set_latency_us(50, 400);
  const int kShards = 4;
  std::vector<sdb_index *> idx;
  for (int r = 0; r < kShards; r++) idx.push_back(new_index());
  {
    auto made = cluster::GpuFanout::New(idx, std::vector<int>(kShards, 0), 2);
    CHECK(!made.second, "fan-out: %s", made.second.msg.c_str());
    auto &f = *made.first;
    std::vector<std::thread> pool;
    for (int t = 0; t < 16; t++)
      pool.emplace_back([&, t] {
        std::minstd_rand rng((unsigned)t + 99);
        for (int it = 0; it < scaled(25); it++) {
          const size_t nq = 1 + rng() % 6;
          const int limit = 1 + (int)(rng() % 20);
          const bool failing = rng() % 10 == 0;
          std::vector<float> q(nq * kDim, 0.f);
          for (size_t i = 0; i < nq; i++) q[i * kDim] = (float)(t * 100000 + it * 10 + (int)i), q[i * kDim + 1] = (float)t;
          auto res = f.SearchPoints(q.data(), nq, limit, failing ? (int)MOCK_FAILING_SEARCH_SIZE : 75);
          if (failing) {
            CHECK((bool)res.err, "a request whose shards fail must fail");
            continue;
          }
          CHECK(!res.err, "fan-out request failed: %s", res.err.msg.c_str());
          if (res.err) continue;
          for (size_t i = 0; i < nq; i++) {
            std::vector<uint64_t> e_ids;
            std::vector<float> e_d;
            std::vector<uint32_t> e_s;
            expected_merge(q.data() + i * kDim, kShards, (uint32_t)limit, &e_ids, &e_d, &e_s);
            CHECK(res.counts[i] == e_ids.size(), "merged count %u, expected %zu", res.counts[i], e_ids.size());
            for (size_t j = 0; j < e_ids.size() && j < res.counts[i]; j++) {
              CHECK(res.ids[i * (size_t)limit + j] == e_ids[j], "merged id");
              CHECK(res.shards[i * (size_t)limit + j] == e_s[j], "merged shard");
              CHECK(res.dists[i * (size_t)limit + j] == e_d[j], "merged distance");
            }
          }
        }
      });
    for (auto &t : pool) t.join();
  }
  for (auto *h : idx) sdb_index_destroy(h);
}

// the raw collective: 8 ranks x 8 threads present the same tickets in any order; every exchange must pair blocks of
// ONE request (the mock counts a violation otherwise) and every answer is the request's
static void test_exchange_order() {
  mock_set_latency_us(20, 200);
  const int kRanks = 8, kThreads = 8, kTickets = scaled(160);
  std::vector<sdb_index *> idx;
  for (int r = 0; r < kRanks; r++) idx.push_back(new_index());
  std::vector<sdb_cluster *> ranks((size_t)kRanks, nullptr);
  std::vector<int> devs((size_t)kRanks, 0);
  CHECK(sdb_cluster_create_local(kRanks, devs.data(), ranks.data()) == SDB_OK, "create_local");
  std::vector<std::thread> pool;
  for (int r = 0; r < kRanks; r++) {
    std::vector<std::vector<uint64_t>> mine((size_t)kThreads);
    std::minstd_rand rng((unsigned)(r + 5) * 31);
    for (int t = 1; t <= kTickets; t++) mine[rng() % kThreads].push_back((uint64_t)t);
    for (int th = 0; th < kThreads; th++)
      pool.emplace_back([&, r, tickets = mine[(size_t)th]] {
        for (uint64_t ticket : tickets) {
          const size_t nq = 1 + ticket % 3;
          const uint32_t limit = 10;
          std::vector<float> q(nq * kDim, 0.f);
          for (size_t i = 0; i < nq; i++) q[i * kDim] = (float)(ticket * 8 + i);
          std::vector<uint64_t> ids(nq * limit);
          std::vector<float> d(nq * limit);
          std::vector<uint32_t> sh(nq * limit), cnt(nq);
          const int rc = sdb_cluster_search_batch(ranks[(size_t)r], idx[(size_t)r], ticket, nq, q.data(), limit, 75, ids.data(),
                                                  d.data(), sh.data(), cnt.data(), SDB_MEM_HOST, nullptr);
          CHECK(rc == SDB_OK, "rank %d ticket %" PRIu64 ": %s", r, ticket, sdb_last_error());
          if (rc) continue;
          for (size_t i = 0; i < nq; i++) {
            std::vector<uint64_t> e_ids;
            std::vector<float> e_d;
            std::vector<uint32_t> e_s;
            expected_merge(q.data() + i * kDim, kRanks, limit, &e_ids, &e_d, &e_s);
            CHECK(cnt[i] == e_ids.size(), "count");
            for (size_t j = 0; j < e_ids.size() && j < cnt[i]; j++) CHECK(ids[i * limit + j] == e_ids[j], "rank %d: merged id", r);
          }
        }
      });
  }
  for (auto &t : pool) t.join();
  for (auto *c : ranks) sdb_cluster_destroy(c);
  for (auto *h : idx) sdb_index_destroy(h);
}

// a rank that never arrives for a request: its peers withdraw after the deadline and stay in step; the lost ticket is
// skipped on the rank that missed it; the next request is served
static void test_exchange_missing_rank() {
  mock_set_latency_us(10, 50);
  const int kRanks = 4;
  std::vector<sdb_index *> idx;
  for (int r = 0; r < kRanks; r++) idx.push_back(new_index());
  std::vector<sdb_cluster *> ranks((size_t)kRanks, nullptr);
  std::vector<int> devs((size_t)kRanks, 0);
  CHECK(sdb_cluster_create_local(kRanks, devs.data(), ranks.data()) == SDB_OK, "create_local");
  for (auto *c : ranks) sdb_cluster_set_deadline(c, 400);  // (generous: a loaded CI box under TSan must not time a healthy request out)
  auto call = [&](int r, uint64_t ticket, int *rc_out) {
    std::vector<float> q(kDim, 0.f);
    q[0] = (float)ticket;
    std::vector<uint64_t> ids(10);
    std::vector<float> d(10);
    std::vector<uint32_t> sh(10), cnt(1);
    *rc_out = sdb_cluster_search_batch(ranks[(size_t)r], idx[(size_t)r], ticket, 1, q.data(), 10, 75, ids.data(), d.data(),
                                       sh.data(), cnt.data(), SDB_MEM_HOST, nullptr);
    if (*rc_out == SDB_OK) {
      std::vector<uint64_t> e_ids;
      std::vector<float> e_d;
      std::vector<uint32_t> e_s;
      expected_merge(q.data(), kRanks, 10, &e_ids, &e_d, &e_s);
      CHECK(cnt[0] == 10 && ids[0] == e_ids[0] && ids[9] == e_ids[9], "answer of ticket %" PRIu64, ticket);
    }
  };
  auto everybody = [&](uint64_t ticket, int absent, std::vector<int> *rcs) {
    rcs->assign((size_t)kRanks, -1);
    std::vector<std::thread> pool;
    for (int r = 0; r < kRanks; r++)
      if (r != absent) pool.emplace_back([&, r] { call(r, ticket, &(*rcs)[(size_t)r]); });
    for (auto &t : pool) t.join();
  };
  std::vector<int> rcs;
  everybody(1, -1, &rcs);
  for (int rc : rcs) CHECK(rc == SDB_OK, "ticket 1");
  everybody(2, 3, &rcs);  // rank 3 never presents ticket 2
  for (int r = 0; r < 3; r++) CHECK(rcs[(size_t)r] == SDB_ERR_STATE, "rank %d must withdraw ticket 2 (rc %d)", r, rcs[(size_t)r]);
  CHECK(sdb_cluster_skip_ticket(ranks[3], 2, 0, 0, 0) == SDB_OK, "skip on the rank that missed the request");
  everybody(3, -1, &rcs);  // ... and the next request is served by all four
  for (int r = 0; r < kRanks; r++) CHECK(rcs[(size_t)r] == SDB_OK, "ticket 3 on rank %d: rc %d %s", r, rcs[(size_t)r], sdb_last_error());
  // a rank that stands in for a request it lost while its peers ARE inside: they fail that request and serve the next
  {
    std::vector<std::thread> pool;
    std::vector<int> rc4((size_t)kRanks, -1);
    for (int r = 0; r < 3; r++) pool.emplace_back([&, r] { call(r, 4, &rc4[(size_t)r]); });
    pool.emplace_back([&] { rc4[3] = sdb_cluster_skip_ticket(ranks[3], 4, 1, 0, 10); });
    for (auto &t : pool) t.join();
    for (int r = 0; r < 3; r++) CHECK(rc4[(size_t)r] == SDB_ERR_STATE, "rank %d: a request one rank skipped must fail (rc %d)", r, rc4[(size_t)r]);
    CHECK(rc4[3] == SDB_OK, "the stand-in: rc %d %s", rc4[3], sdb_last_error());
  }
  everybody(5, -1, &rcs);
  for (int r = 0; r < kRanks; r++) CHECK(rcs[(size_t)r] == SDB_OK, "ticket 5 on rank %d", r);
  // destruction of one rank while its peers wait for it: they are released with an error, not left waiting
  for (auto *c : ranks) sdb_cluster_set_deadline(c, 0);
  {
    std::vector<std::thread> pool;
    std::vector<int> rc6((size_t)kRanks, -1);
    for (int r = 0; r < 3; r++) pool.emplace_back([&, r] { call(r, 6, &rc6[(size_t)r]); });
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    sdb_cluster_destroy(ranks[3]);
    ranks[3] = nullptr;
    for (auto &t : pool) t.join();
    for (int r = 0; r < 3; r++) CHECK(rc6[(size_t)r] != SDB_OK, "rank %d: its peer is gone", r);
  }
  for (auto *c : ranks) sdb_cluster_destroy(c);
  for (auto *h : idx) sdb_index_destroy(h);
}

// ---------------------------------------------------------------------------------------------------------------
// (d) IndexVamana: searches from many threads while writers insert / update / delete
static void test_vamana_readers_and_writers() {
  mock_set_latency_us(30, 200);
  diskstore::MemBucket bucket;
  models::IndexVectorVamanaParameters params;
  params.VectorSize = kDim, params.DistanceMetric = models::DistanceEuclidean;
  auto made = vamana::NewIndexVamana("stress", params, &bucket);
  CHECK(!made.second, "NewIndexVamana: %s", made.second.msg.c_str());
  auto &ix = *made.first;
  std::atomic<bool> stop{false};
  std::vector<std::thread> pool;
  for (int w = 0; w < 2; w++)
    pool.emplace_back([&, w] {
      for (int it = 0; it < scaled(30); it++) {
        std::vector<vamana::IndexVectorChange> ch;
        for (int i = 0; i < 5; i++) {
          vamana::IndexVectorChange c;
          c.Id = (uint64_t)(10 + w * 100000 + it * 5 + i);
          c.Vector.assign(kDim, (float)i);
          ch.push_back(c);
        }
        if (it > 0) {  // delete one of the last round's, update another
          vamana::IndexVectorChange del;
          del.Id = (uint64_t)(10 + w * 100000 + (it - 1) * 5);
          ch.push_back(del);
          vamana::IndexVectorChange upd;
          upd.Id = (uint64_t)(10 + w * 100000 + (it - 1) * 5 + 1);
          upd.Vector.assign(kDim, 9.f);
          ch.push_back(upd);
        }
        Error e = ix.InsertUpdateDelete(ch);
        CHECK(!e, "InsertUpdateDelete: %s", e.msg.c_str());
      }
    });
  for (int t = 0; t < 12; t++)
    pool.emplace_back([&, t] {
      int n = 0;
      while (!stop.load() || n < 10) {
        models::SearchVectorVamanaOptions q;
        q.Vector.assign(kDim, 0.f);
        q.Vector[0] = (float)(t * 10000 + n), q.Vector[1] = (float)t;
        q.Limit = 1 + n % 20, q.SearchSize = 75;
        auto out = ix.Search(q);
        CHECK(!out.err, "Search: %s", out.err.msg.c_str());
        std::vector<uint64_t> e_ids((size_t)q.Limit);
        std::vector<float> e_d((size_t)q.Limit);
        uint32_t e_c = 0;
        mock_expect(q.Vector.data(), kDim, (uint32_t)q.Limit, nullptr, 0, e_ids.data(), e_d.data(), &e_c);
        CHECK(out.results.size() == e_c, "result count");
        for (size_t j = 0; j < out.results.size() && j < e_c; j++) CHECK(out.results[j].NodeId == e_ids[j], "result id");
        n++;
      }
    });
  pool[0].join(), pool[1].join();
  stop = true;
  for (size_t i = 2; i < pool.size(); i++) pool[i].join();
  CHECK(ix.Exists(10 + 1 * 100000 + 2), "a point that was inserted and never deleted");
  CHECK(!ix.Exists(10), "a deleted point");
}

// not part of the suite: proves that the sanitizer of this build is awake (tests/test_host_concurrency.py runs it by
// name and demands a report): an unsynchronised counter and a read past a heap block
static int g_racy = 0;
static void test_seeded_bugs() {
  std::thread a([] { for (int i = 0; i < 100000; i++) g_racy++; }), b([] { for (int i = 0; i < 100000; i++) g_racy++; });
  a.join(), b.join();
  volatile int *p = new int[4];
  volatile int idx = 4;
  g_racy += p[idx];
  delete[] p;
}

int main(int argc, char **argv) {
  if (const char *s = std::getenv("SDB_STRESS_SCALE")) g_scale = std::atof(s) > 0 ? std::atof(s) : 1.0;
  struct T {
    const char *name;
    void (*fn)();
  } tests[] = {
      {"batcher_mixed", test_batcher_mixed},
      {"batcher_backpressure", test_batcher_backpressure},
      {"batcher_window_race", test_batcher_window_race},
      {"batcher_pageable", test_batcher_pageable},
      {"batcher_poll_and_reissue", test_batcher_poll_and_reissue},
      {"batcher_destructor", test_batcher_destructor},
      {"turnstile_order", test_turnstile_order},
      {"turnstile_missing_ticket", test_turnstile_missing_ticket},
      {"view_mutex_writer_gets_its_turn", test_view_mutex_writer_gets_its_turn},
      {"fanout", test_fanout},
      {"exchange_order", test_exchange_order},
      {"exchange_missing_rank", test_exchange_missing_rank},
      {"vamana_readers_and_writers", test_vamana_readers_and_writers},
  };
  int ran = 0;
  if (argc > 1 && std::string(argv[1]) == "seeded_bugs") {
    test_seeded_bugs();
    std::printf("seeded bugs ran (%d)\n", g_racy);
    return 0;
  }
  for (auto &t : tests) {
    if (argc > 1 && std::string(argv[1]) != t.name) continue;
    const int before = g_failures.load();
    const auto t0 = std::chrono::steady_clock::now();
    t.fn();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%s %s (%.1f s)\n", g_failures.load() == before ? "ok  " : "FAIL", t.name, s);
    std::fflush(stdout);
    ran++;
  }
  if (mock_violations()) {
    std::fprintf(stderr, "the mock saw %" PRIu64 " breaches of the calling contract; first: %s\n", mock_violations(), mock_first_violation());
    g_failures++;
  }
  std::printf("%d tests, %d failures, %" PRIu64 " device calls\n", ran, g_failures.load(), mock_search_calls());
  return g_failures.load() ? 1 : 0;
}
