// test_faults.cpp -- the C ABI's "never aborts the process" under host-memory exhaustion (CONTRIBUTING.md:150: no
// panics; the reference returns an `error` and its cache manager scraps a shard's cache after a failed transaction,
// shard/cache/manager.go:231-240).
//
// The global operator new is replaced by one that throws std::bad_alloc at the k-th allocation made FROM INSIDE
// libsemadb_amd.so (the caller's return address decides; the HIP runtime's and libstdc++'s own allocations are left
// alone -- they are not this library's to make exception-safe) on the armed thread.  k is swept from 0 until the call
// goes through untouched, over load / insert_batch / delete_batch / filtered search_batch / compact /
// cluster_search_batch.  After every injected failure:
//   - the call returned a status != SDB_OK and sdb_last_error() names it; the process is alive;
//   - the index answers a reference batch bit-identically (ids, distance bits, counts) to what it answered before,
//     or -- write calls that had begun to change the graph -- is unusable (every call SDB_ERR_STATE) until reloaded;
//   - a cluster handle either serves the next request or says it is out of step (recreate).
// When the sweep ends the call has succeeded and its result is what an undisturbed run produces.
#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/semadb_amd.h"

// ---------------------------------------------------------------------------------------------------------------
// fault injector
// ---------------------------------------------------------------------------------------------------------------
static uintptr_t g_lo = 0, g_hi = 0;  // address range of libsemadb_amd.so
static thread_local long t_countdown = -1;  // < 0: not armed
static thread_local long t_seen = 0;        // library allocations on this thread since arm()
static thread_local bool t_fired = false;

static int find_lib(struct dl_phdr_info *info, size_t, void *) {
  if (!info->dlpi_name || !strstr(info->dlpi_name, "libsemadb_amd.so")) return 0;
  uintptr_t lo = ~(uintptr_t)0, hi = 0;
  for (int i = 0; i < info->dlpi_phnum; i++)
    if (info->dlpi_phdr[i].p_type == PT_LOAD) {
      const uintptr_t b = info->dlpi_addr + info->dlpi_phdr[i].p_vaddr, e = b + info->dlpi_phdr[i].p_memsz;
      if (b < lo) lo = b;
      if (e > hi) hi = e;
    }
  g_lo = lo, g_hi = hi;
  return 1;
}

static inline bool inject(void *ra) {
  if (t_countdown < 0) return false;
  const uintptr_t a = (uintptr_t)ra;
  if (a < g_lo || a >= g_hi) return false;
  t_seen++;
  if (t_countdown == 0) {
    t_countdown = -1;  // one failure per arming
    t_fired = true;
    return true;
  }
  t_countdown--;
  return false;
}

__attribute__((noinline)) void *operator new(size_t n) {
  if (inject(__builtin_return_address(0))) throw std::bad_alloc();
  void *p = malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
__attribute__((noinline)) void *operator new[](size_t n) {
  if (inject(__builtin_return_address(0))) throw std::bad_alloc();
  void *p = malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
__attribute__((noinline)) void *operator new(size_t n, const std::nothrow_t &) noexcept {
  if (inject(__builtin_return_address(0))) return nullptr;
  return malloc(n ? n : 1);
}
__attribute__((noinline)) void *operator new[](size_t n, const std::nothrow_t &) noexcept {
  if (inject(__builtin_return_address(0))) return nullptr;
  return malloc(n ? n : 1);
}
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
void operator delete(void *p, const std::nothrow_t &) noexcept { free(p); }
void operator delete[](void *p, const std::nothrow_t &) noexcept { free(p); }

static void arm(long k) { t_seen = 0, t_fired = false, t_countdown = k; }
static bool disarm() {  // did the armed failure happen?
  t_countdown = -1;
  return t_fired;
}

// ---------------------------------------------------------------------------------------------------------------
static int g_fail = 0;
#define CHECK(cond)                                               \
  do {                                                            \
    if (!(cond)) {                                                \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      g_fail++;                                                   \
    }                                                             \
  } while (0)
#define OK(expr)                                                                                  \
  do {                                                                                            \
    int _rc = (expr);                                                                             \
    if (_rc != SDB_OK) {                                                                          \
      std::printf("FAIL %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #expr, _rc, sdb_last_error()); \
      g_fail++;                                                                                   \
    }                                                                                             \
  } while (0)

constexpr uint32_t D = 48, NQ = 24, LIMIT = 10, L = 50;

struct Graph {  // what a bucket holds: ids, vectors, CSR edge lists
  std::vector<uint64_t> ids, offsets, edges;
  std::vector<float> vecs;
};
struct Answer {
  std::vector<uint64_t> ids;
  std::vector<float> d;
  std::vector<uint32_t> c;
  bool operator==(const Answer &o) const {
    return ids == o.ids && c == o.c && !memcmp(d.data(), o.d.data(), d.size() * 4);
  }
};

static sdb_index *new_index() {
  sdb_index_params p{};
  p.dim = D, p.metric = SDB_METRIC_EUCLIDEAN, p.search_size = L, p.degree_bound = 32, p.alpha = 1.2f, p.device = 0;
  p.capacity = 4096;
  sdb_index *ix = nullptr;
  OK(sdb_index_create(&p, &ix));
  return ix;
}

static Graph export_graph(sdb_index *ix) {
  Graph g;
  uint64_t n = 0, ne = 0;
  OK(sdb_index_stats(ix, &n, &ne, nullptr));
  g.ids.resize(n), g.vecs.resize(n * D), g.offsets.resize(n + 1), g.edges.resize(ne ? ne : 1);
  OK(sdb_index_export(ix, g.ids.data(), g.vecs.data(), g.offsets.data(), g.edges.data()));
  return g;
}

static int load_graph(sdb_index *ix, const Graph &g) {
  return sdb_index_load(ix, g.ids.size(), g.ids.data(), g.vecs.data(), g.offsets.data(), g.edges.data(), SDB_MEM_HOST);
}

static int search(sdb_index *ix, const std::vector<float> &q, Answer *a, const std::vector<uint64_t> *foff = nullptr,
                  const std::vector<uint64_t> *fids = nullptr) {
  a->ids.assign(NQ * LIMIT, 0), a->d.assign(NQ * LIMIT, 0.f), a->c.assign(NQ, 0);
  return sdb_index_search_batch(ix, NQ, q.data(), LIMIT, L, foff ? foff->data() : nullptr, fids ? fids->data() : nullptr,
                                a->ids.data(), a->d.data(), a->c.data(), nullptr, SDB_MEM_HOST, nullptr);
}

static bool is_broken(sdb_index *ix, const std::vector<float> &q) {
  Answer a;
  return search(ix, q, &a) == SDB_ERR_STATE && sdb_index_begin_write(ix) == SDB_ERR_STATE;
}

// sweeps the countdown over `call` (made on a state that `fresh` sets up each time); `after_fail` judges the state a
// failed call left behind.  Returns the number of injected failures.
template <class Fresh, class Call, class AfterFail, class AfterOk>
static long sweep(const char *name, Fresh fresh, Call call, AfterFail after_fail, AfterOk after_ok, long max_k = 4000) {
  long injected = 0;
  for (long k = 0; k < max_k; k++) {
    fresh();
    arm(k);
    const int rc = call();
    const bool fired = disarm();
    if (!fired) {  // the call made fewer than k + 1 allocations: it ran undisturbed
      if (rc != SDB_OK) std::printf("FAIL %s: undisturbed call returned %d (%s)\n", name, rc, sdb_last_error()), g_fail++;
      after_ok();
      std::printf("%-28s %ld injected failures, every one a status; final call ok\n", name, injected);
      return injected;
    }
    injected++;
    if (rc == SDB_OK) {
      // an allocation failure the library absorbed (a fallback path): the result must still be right
      after_ok();
      continue;
    }
    if (!sdb_last_error()[0]) std::printf("FAIL %s k=%ld: status %d without a message\n", name, k, rc), g_fail++;
    after_fail(k, rc);
  }
  std::printf("FAIL %s: still failing after %ld countdowns\n", name, max_k);
  g_fail++;
  return injected;
}

int main() {
  dl_iterate_phdr(find_lib, nullptr);
  if (!g_lo) {
    std::printf("FAIL: libsemadb_amd.so not found among the loaded objects\n");
    return 1;
  }
  int ndev = 0;
  if (sdb_device_count(&ndev) != SDB_OK) {
    std::printf("no GPU: %s\n", sdb_last_error());
    return 2;
  }
  std::mt19937 rng(20251004);
  std::normal_distribution<float> nd;
  const uint32_t N = 3000, NEXTRA = 400;
  std::vector<float> base(N * D), extra(NEXTRA * D), start(D), queries(NQ * D);
  for (auto &x : base) x = nd(rng);
  for (auto &x : extra) x = nd(rng);
  for (auto &x : start) x = nd(rng);
  for (auto &x : queries) x = nd(rng);
  // ids with holes (every third id skipped): the id -> slot tables are hash maps, the host filter path does real work
  std::vector<uint64_t> base_ids(N), extra_ids(NEXTRA);
  for (uint32_t i = 0; i < N; i++) base_ids[i] = 2 + (uint64_t)i + i / 2;
  for (uint32_t i = 0; i < NEXTRA; i++) extra_ids[i] = 100000 + 3 * (uint64_t)i;

  // ---- the undisturbed run: base graph, its answers, the graph after the extra inserts / a delete / compact
  sdb_index *ref = new_index();
  OK(sdb_index_set_start(ref, start.data(), SDB_MEM_HOST));
  OK(sdb_index_insert_batch(ref, N, base_ids.data(), base.data(), SDB_MEM_HOST, 0, nullptr));
  const Graph g0 = export_graph(ref);
  Answer a0;
  OK(search(ref, queries, &a0));
  std::vector<uint64_t> foff(NQ + 1), fids;
  for (uint32_t q = 0; q < NQ; q++) {
    foff[q] = fids.size();
    for (uint32_t i = q % 5; i < N; i += 5 + q % 3) fids.push_back(base_ids[i]);
    fids.push_back(999999999ull + q);  // unknown ids are skipped (itemcache.go:109-128)
  }
  foff[NQ] = fids.size();
  OK(sdb_index_set_tuning(ref, SDB_TUNE_HOST_FILTERS, 1));  // the host's hash map + threads: the path that allocates
  Answer af0;
  OK(search(ref, queries, &af0, &foff, &fids));
  OK(sdb_index_set_tuning(ref, SDB_TUNE_HOST_FILTERS, 0));
  Answer af0_dev;
  OK(search(ref, queries, &af0_dev, &foff, &fids));
  CHECK(af0 == af0_dev);
  OK(sdb_index_insert_batch(ref, NEXTRA, extra_ids.data(), extra.data(), SDB_MEM_HOST, 0, nullptr));
  const Graph g1 = export_graph(ref);
  Answer a1;
  OK(search(ref, queries, &a1));
  std::vector<uint64_t> del_ids;
  for (uint32_t i = 0; i < N; i += 7) del_ids.push_back(base_ids[i]);
  OK(sdb_index_delete_batch(ref, del_ids.size(), del_ids.data(), nullptr));
  const Graph g2 = export_graph(ref);
  Answer a2;
  OK(search(ref, queries, &a2));
  OK(sdb_index_destroy(ref));

  auto same_graph = [](const Graph &a, const Graph &b) {
    return a.ids == b.ids && a.offsets == b.offsets && a.edges == b.edges && a.vecs == b.vecs;
  };

  // ---- 1. sdb_index_load: a failed load leaves the index empty and loadable
  {
    sdb_index *ix = new_index();
    sweep(
        "sdb_index_load", [] {}, [&] { return load_graph(ix, g0); },
        [&](long k, int) {
          uint64_t n = 1;
          OK(sdb_index_stats(ix, &n, nullptr, nullptr));
          if (n != 0) std::printf("FAIL load k=%ld: %llu rows left behind\n", k, (unsigned long long)n), g_fail++;
          Answer a;
          CHECK(search(ix, queries, &a) == SDB_ERR_STATE);  // no start point: an error, not a crash
        },
        [&] {
          Answer a;
          OK(search(ix, queries, &a));
          CHECK(a == a0);
        });
    OK(sdb_index_destroy(ix));
  }

  // ---- 2. filtered sdb_index_search_batch (host-resolved filters: vectors, threads) and the plain one
  {
    sdb_index *ix = new_index();
    OK(load_graph(ix, g0));
    for (int host_filters = 0; host_filters < 2; host_filters++) {
      OK(sdb_index_set_tuning(ix, SDB_TUNE_HOST_FILTERS, host_filters));
      Answer a;
      sweep(
          host_filters ? "search_batch (host filters)" : "search_batch (device filters)", [] {},
          [&] { return search(ix, queries, &a, &foff, &fids); },
          [&](long, int) {
            Answer b;
            OK(search(ix, queries, &b, &foff, &fids));
            CHECK(b == af0);
          },
          [&] { CHECK(a == af0); });
    }
    Answer a;
    sweep(
        "search_batch (unfiltered)", [] {}, [&] { return search(ix, queries, &a); },
        [&](long, int) {
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == a0);
        },
        [&] { CHECK(a == a0); });
    // bitmaps over a table with holes
    {
      std::vector<uint64_t> first(NQ), woff(NQ + 1), words;
      for (uint32_t q = 0; q < NQ; q++) {
        first[q] = 0, woff[q] = words.size();
        std::vector<uint64_t> w((base_ids[N - 1] + 64) / 64, 0);
        for (uint64_t i = foff[q]; i + 1 < foff[q + 1]; i++) w[fids[i] / 64] |= 1ull << (fids[i] % 64);
        words.insert(words.end(), w.begin(), w.end());
      }
      woff[NQ] = words.size();
      auto bsearch = [&](Answer *x) {
        x->ids.assign(NQ * LIMIT, 0), x->d.assign(NQ * LIMIT, 0.f), x->c.assign(NQ, 0);
        return sdb_index_search_batch_bitmap(ix, NQ, queries.data(), LIMIT, L, first.data(), woff.data(), words.data(),
                                             x->ids.data(), x->d.data(), x->c.data(), nullptr, SDB_MEM_HOST, nullptr);
      };
      sweep(
          "search_batch_bitmap (host)", [] {}, [&] { return bsearch(&a); },
          [&](long, int) {
            Answer b;
            OK(bsearch(&b));
            CHECK(b == af0);
          },
          [&] { CHECK(a == af0); });
    }
    OK(sdb_index_destroy(ix));
  }

  // ---- 3. sdb_index_insert_batch: as it was, or unusable; never half a transaction
  {
    sdb_index *ix = nullptr;
    long unusable = 0, intact = 0;
    sweep(
        "sdb_index_insert_batch",
        [&] {
          if (ix) OK(sdb_index_destroy(ix));
          ix = new_index();
          OK(load_graph(ix, g0));
        },
        [&] { return sdb_index_insert_batch(ix, NEXTRA, extra_ids.data(), extra.data(), SDB_MEM_HOST, 0, nullptr); },
        [&](long k, int) {
          if (is_broken(ix, queries)) {
            unusable++;
            return;
          }
          intact++;
          Answer b;
          OK(search(ix, queries, &b));
          if (!(b == a0)) std::printf("FAIL insert k=%ld: a failed insert changed the answers\n", k), g_fail++;
          uint64_t diff = 1;
          OK(sdb_index_version_diff(ix, &diff));
          CHECK(diff == 0);
          // and the same insert goes through afterwards
          OK(sdb_index_insert_batch(ix, NEXTRA, extra_ids.data(), extra.data(), SDB_MEM_HOST, 0, nullptr));
          CHECK(same_graph(export_graph(ix), g1));
        },
        [&] {
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == a1);
          CHECK(same_graph(export_graph(ix), g1));
        });
    std::printf("    insert: %ld failures left the index as it was, %ld left it unusable (reload)\n", intact, unusable);
    // ---- 4. sdb_index_delete_batch on the grown graph
    unusable = intact = 0;
    sweep(
        "sdb_index_delete_batch",
        [&] {
          if (ix) OK(sdb_index_destroy(ix));
          ix = new_index();
          OK(load_graph(ix, g1));
        },
        [&] { return sdb_index_delete_batch(ix, del_ids.size(), del_ids.data(), nullptr); },
        [&](long k, int) {
          if (is_broken(ix, queries)) {
            unusable++;
            return;
          }
          intact++;
          Answer b;
          OK(search(ix, queries, &b));
          if (!(b == a1)) std::printf("FAIL delete k=%ld: a failed delete changed the answers\n", k), g_fail++;
          OK(sdb_index_delete_batch(ix, del_ids.size(), del_ids.data(), nullptr));
          CHECK(same_graph(export_graph(ix), g2));
        },
        [&] {
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == a2);
          CHECK(same_graph(export_graph(ix), g2));
        });
    std::printf("    delete: %ld failures left the index as it was, %ld left it unusable (reload)\n", intact, unusable);
    // ---- 5. sdb_index_compact: a failed compact changes nothing
    sweep(
        "sdb_index_compact", [] {}, [&] { return sdb_index_compact(ix); },
        [&](long k, int) {
          uint64_t rows = 0, dead = 0;
          OK(sdb_index_row_usage(ix, &rows, &dead));
          if (dead != del_ids.size()) std::printf("FAIL compact k=%ld: tombstones %llu\n", k, (unsigned long long)dead), g_fail++;
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == a2);
        },
        [&] {
          uint64_t rows = 0, dead = 1;
          OK(sdb_index_row_usage(ix, &rows, &dead));
          CHECK(dead == 0);
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == a2);
          CHECK(same_graph(export_graph(ix), g2));
        });
    // export / get_vectors / edge_scan are read-only: a status, nothing else
    {
      Graph g;
      uint64_t n = 0, ne = 0;
      OK(sdb_index_stats(ix, &n, &ne, nullptr));
      g.ids.resize(n), g.vecs.resize(n * D), g.offsets.resize(n + 1), g.edges.resize(ne ? ne : 1);
      sweep(
          "sdb_index_export", [] {},
          [&] { return sdb_index_export(ix, g.ids.data(), g.vecs.data(), g.offsets.data(), g.edges.data()); }, [&](long, int) {},
          [&] { CHECK(same_graph(g, g2)); });
      std::vector<uint64_t> tp(n), ts(n);
      uint64_t np = 0, ns = 0;
      sweep(
          "sdb_index_edge_scan", [] {},
          [&] { return sdb_index_edge_scan(ix, 5, extra_ids.data(), tp.data(), n, &np, ts.data(), n, &ns, nullptr); },
          [&](long, int) {}, [&] { CHECK(np > 0); });
    }
    OK(sdb_index_destroy(ix));
  }

  // ---- 6. sdb_cluster_search_batch: two shards of one GPU, the fault on rank 0's thread
  {
    const uint32_t H = N / 2;
    sdb_index *sh[2];
    for (int r = 0; r < 2; r++) {
      sh[r] = new_index();
      OK(sdb_index_set_start(sh[r], start.data(), SDB_MEM_HOST));
      OK(sdb_index_insert_batch(sh[r], H, base_ids.data() + r * H, base.data() + (size_t)r * H * D, SDB_MEM_HOST, 0, nullptr));
    }
    sdb_cluster *cl[2] = {nullptr, nullptr};
    const int devs[2] = {0, 0};
    auto make = [&] {
      OK(sdb_cluster_create_local(2, devs, cl));
      for (auto *c : cl) OK(sdb_cluster_set_deadline(c, 400));
    };
    auto drop = [&] {
      for (auto *&c : cl) {
        OK(sdb_cluster_destroy(c));
        c = nullptr;
      }
    };
    auto request = [&](long k0, Answer out[2], int rc[2]) {  // k0 < 0: undisturbed.  returns: did the fault fire
      bool fired = false;
      std::thread peer([&] {
        out[1].ids.assign(NQ * LIMIT, 0), out[1].d.assign(NQ * LIMIT, 0.f), out[1].c.assign(NQ, 0);
        rc[1] = sdb_cluster_search_batch(cl[1], sh[1], 0, NQ, queries.data(), LIMIT, L, out[1].ids.data(), out[1].d.data(), nullptr,
                                         out[1].c.data(), SDB_MEM_HOST, nullptr);
      });
      out[0].ids.assign(NQ * LIMIT, 0), out[0].d.assign(NQ * LIMIT, 0.f), out[0].c.assign(NQ, 0);
      if (k0 >= 0) arm(k0);
      rc[0] = sdb_cluster_search_batch(cl[0], sh[0], 0, NQ, queries.data(), LIMIT, L, out[0].ids.data(), out[0].d.data(), nullptr,
                                       out[0].c.data(), SDB_MEM_HOST, nullptr);
      if (k0 >= 0) fired = disarm();
      peer.join();
      return fired;
    };
    make();
    Answer want[2];
    int rc[2];
    request(-1, want, rc);
    CHECK(rc[0] == SDB_OK && rc[1] == SDB_OK && want[0] == want[1] && want[0].c[0] == LIMIT);
    long injected = 0, recreated = 0;
    for (long k = 0; k < 4000; k++) {
      Answer got[2];
      const bool fired = request(k, got, rc);
      if (!fired) {
        CHECK(rc[0] == SDB_OK && rc[1] == SDB_OK && got[0] == want[0] && got[1] == want[0]);
        break;
      }
      injected++;
      if (rc[0] == SDB_OK) {
        CHECK(got[0] == want[0]);
      } else {
        bool empty = true;
        for (auto c : got[0].c) empty &= c == 0;
        CHECK(empty);  // a failed request has no answer
      }
      // the next request: served, or the handle says it is out of step and a new group serves it
      Answer nxt[2];
      request(-1, nxt, rc);
      if (rc[0] != SDB_OK || rc[1] != SDB_OK) {
        CHECK(rc[0] == SDB_ERR_STATE || rc[1] == SDB_ERR_STATE);
        drop(), make();
        recreated++;
        request(-1, nxt, rc);
      }
      CHECK(rc[0] == SDB_OK && rc[1] == SDB_OK && nxt[0] == want[0] && nxt[1] == want[0]);
    }
    std::printf("%-28s %ld injected failures, every one a status; group recreated %ld times\n", "sdb_cluster_search_batch", injected,
                recreated);
    // create_local itself
    drop();
    sweep(
        "sdb_cluster_create_local", [] {}, [&] { return sdb_cluster_create_local(2, devs, cl); },
        [&](long, int) { CHECK(cl[0] == nullptr && cl[1] == nullptr); }, [&] { CHECK(cl[0] && cl[1]); });
    drop();
    for (auto *s : sh) OK(sdb_index_destroy(s));
  }

  // ---- 6b. the flat index's calls, the stored-vector reads, K1 over stored ids, the quantizer's calls
  {
    sdb_index *fx = new_index();  // no start node: a flat index (flat.go)
    std::vector<uint64_t> fids(600);
    for (uint32_t i = 0; i < 600; i++) fids[i] = 10 + 2 * (uint64_t)i;
    sweep(
        "sdb_index_set_vectors", [] {}, [&] { return sdb_index_set_vectors(fx, 600, fids.data(), base.data(), SDB_MEM_HOST); },
        [&](long k, int) {
          if (sdb_index_begin_write(fx) == SDB_ERR_STATE) {  // half-changed id tables: unusable, start over
            OK(sdb_index_destroy(fx));
            fx = new_index();
          } else {
            OK(sdb_index_abort_write(fx));
            uint64_t rows = 1, dead = 0;
            OK(sdb_index_row_usage(fx, &rows, &dead));
            if (rows != 0) std::printf("FAIL set_vectors k=%ld: %llu rows after a failed call\n", k, (unsigned long long)rows), g_fail++;
          }
        },
        [&] {
          uint64_t rows = 0, dead = 0;
          OK(sdb_index_row_usage(fx, &rows, &dead));
          CHECK(rows == 600 && dead == 0);
        });
    Answer fa, fb;
    auto fsearch = [&](Answer *x, bool filtered) {
      x->ids.assign(NQ * LIMIT, 0), x->d.assign(NQ * LIMIT, 0.f), x->c.assign(NQ, 0);
      std::vector<uint64_t> o(NQ + 1), ids;  // 50 stored ids per query, ascending (flat.go:100 scans only the filter's ids)
      for (uint32_t q = 0; q <= NQ; q++) o[q] = (uint64_t)q * 50;
      for (uint32_t q = 0; q < NQ; q++)
        for (uint32_t i = 0; i < 50; i++) ids.push_back(fids[q + 11 * i]);
      return sdb_index_flat_search(fx, NQ, queries.data(), LIMIT, filtered ? o.data() : nullptr, filtered ? ids.data() : nullptr,
                                   x->ids.data(), x->d.data(), x->c.data(), SDB_MEM_HOST, nullptr);
    };
    OK(fsearch(&fa, false));
    sweep(
        "sdb_index_flat_search", [] {}, [&] { return fsearch(&fb, false); }, [&](long, int) {}, [&] { CHECK(fb == fa); });
    Answer ff;
    OK(fsearch(&ff, true));
    sweep(
        "sdb_index_flat_search (filter)", [] {}, [&] { return fsearch(&fb, true); }, [&](long, int) {}, [&] { CHECK(fb == ff); });
    std::vector<float> got(40 * D), want(40 * D);
    std::vector<uint8_t> found(40);
    OK(sdb_index_get_vectors(fx, 40, fids.data(), want.data(), found.data()));
    sweep(
        "sdb_index_get_vectors", [] {}, [&] { return sdb_index_get_vectors(fx, 40, fids.data(), got.data(), found.data()); },
        [&](long, int) {}, [&] { CHECK(got == want && !memcmp(want.data(), base.data(), 40 * D * 4)); });
    std::vector<float> dd(8 * 40), dw(8 * 40);
    std::vector<uint64_t> cand(8 * 40);
    for (uint32_t i = 0; i < 8 * 40; i++) cand[i] = fids[(i * 13) % 600];
    OK(sdb_index_distance_batch(fx, 8, queries.data(), 40, cand.data(), dw.data(), SDB_MEM_HOST, nullptr));
    sweep(
        "sdb_index_distance_batch", [] {},
        [&] { return sdb_index_distance_batch(fx, 8, queries.data(), 40, cand.data(), dd.data(), SDB_MEM_HOST, nullptr); },
        [&](long, int) {}, [&] { CHECK(!memcmp(dd.data(), dw.data(), dd.size() * 4)); });
    long rm_unusable = 0, rm_intact = 0;
    sweep(
        "sdb_index_remove_vectors", [] {}, [&] { return sdb_index_remove_vectors(fx, 100, fids.data()); },
        [&](long k, int) {
          if (sdb_index_begin_write(fx) == SDB_ERR_STATE) {  // rows half-marked: unusable; a host reloads
            rm_unusable++;
            OK(sdb_index_destroy(fx));
            fx = new_index();
            OK(sdb_index_set_vectors(fx, 600, fids.data(), base.data(), SDB_MEM_HOST));
            return;
          }
          rm_intact++;
          OK(sdb_index_abort_write(fx));
          uint64_t rows = 0, dead = 1;
          OK(sdb_index_row_usage(fx, &rows, &dead));
          if (rows != 600 || dead != 0) std::printf("FAIL remove_vectors k=%ld: rows %llu dead %llu\n", k, (unsigned long long)rows, (unsigned long long)dead), g_fail++;
          Answer b;
          OK(fsearch(&b, false));
          CHECK(b == fa);
        },
        [&] {
          uint64_t rows = 0, dead = 0;
          OK(sdb_index_row_usage(fx, &rows, &dead));
          CHECK(rows == 600 && dead == 100);
        },
        400);
    std::printf("    remove_vectors: %ld failures left the index as it was, %ld left it unusable (reload)\n", rm_intact, rm_unusable);
    OK(sdb_index_destroy(fx));
    // the quantizer: fit, codes, table distances, and an index switched over to it
    sdb_pq *pq = nullptr;
    OK(sdb_pq_create(D, SDB_METRIC_EUCLIDEAN, 8, 16, 0, &pq));
    std::vector<uint32_t> first(8);
    for (uint32_t i = 0; i < 8; i++) first[i] = 37 * i;
    std::vector<float> train(base.begin(), base.begin() + 1000 * D);
    std::vector<uint8_t> codes_w(1000 * 8), codes(1000 * 8);
    {
      std::vector<float> t2(train);
      OK(sdb_pq_fit(pq, t2.data(), 1000, first.data(), 1, codes_w.data(), SDB_MEM_HOST, nullptr));
    }
    sweep(
        "sdb_pq_fit", [] {},
        [&] {
          std::vector<float> t2(train);
          return sdb_pq_fit(pq, t2.data(), 1000, first.data(), 1, codes.data(), SDB_MEM_HOST, nullptr);
        },
        [&](long, int) {}, [&] { CHECK(codes == codes_w); });
    std::vector<uint8_t> enc_w(500 * 8), enc(500 * 8);
    OK(sdb_pq_encode(pq, base.data(), 500, enc_w.data(), SDB_MEM_HOST, nullptr));
    sweep(
        "sdb_pq_encode", [] {}, [&] { return sdb_pq_encode(pq, base.data(), 500, enc.data(), SDB_MEM_HOST, nullptr); },
        [&](long, int) {}, [&] { CHECK(enc == enc_w); });
    std::vector<float> ld_w(NQ * 500), ld(NQ * 500);
    OK(sdb_pq_lut_distance(pq, queries.data(), NQ, enc_w.data(), 500, ld_w.data(), SDB_MEM_HOST, nullptr));
    sweep(
        "sdb_pq_lut_distance", [] {},
        [&] { return sdb_pq_lut_distance(pq, queries.data(), NQ, enc_w.data(), 500, ld.data(), SDB_MEM_HOST, nullptr); },
        [&](long, int) {}, [&] { CHECK(!memcmp(ld.data(), ld_w.data(), ld.size() * 4)); });
    sdb_index *qx = new_index();
    OK(load_graph(qx, g0));
    sweep(
        "sdb_index_attach_pq", [] {}, [&] { return sdb_index_attach_pq(qx, pq, nullptr); },
        [&](long, int) {
          Answer b;
          OK(search(qx, queries, &b));
          CHECK(b == a0);  // still the full-precision store
        },
        [&] {
          Answer b;
          OK(search(qx, queries, &b));
          CHECK(b.c == a0.c);
        });
    Answer qa, qb;
    OK(search(qx, queries, &qa));
    sweep(
        "search_batch (quantized)", [] {}, [&] { return search(qx, queries, &qb); }, [&](long, int) {}, [&] { CHECK(qb == qa); });
    std::vector<uint64_t> cid(g0.ids.begin() + 1, g0.ids.begin() + 101);
    std::vector<uint8_t> cc(100 * 8);
    for (size_t i = 0; i < cc.size(); i++) cc[i] = (uint8_t)(i % 16);
    sweep(
        "sdb_index_set_codes", [] {}, [&] { return sdb_index_set_codes(qx, 100, cid.data(), cc.data()); }, [&](long, int) {},
        [&] {
          std::vector<uint8_t> back(100 * 8);
          OK(sdb_index_get_codes(qx, 100, cid.data(), back.data()));
          CHECK(back == cc);
        });
    uint64_t extra1[3] = {g0.ids[5], g0.ids[9], g0.ids[11]};
    sweep(
        "sdb_index_union_prune", [] {}, [&] { return sdb_index_union_prune(qx, g0.ids[3], 3, extra1, 0, nullptr); },
        [&](long, int) {}, [&] {}, 200);
    OK(sdb_index_destroy(qx));
    OK(sdb_pq_destroy(pq));
  }

  // ---- 6c. the calls that take no index: batched distances, the shard merge, k-means
  {
    std::vector<float> out_w(NQ * 300), out(NQ * 300);
    OK(sdb_distance_batch(SDB_METRIC_COSINE, D, queries.data(), NQ, base.data(), 300, out_w.data(), SDB_MEM_HOST, 0, nullptr));
    sweep(
        "sdb_distance_batch", [] {},
        [&] { return sdb_distance_batch(SDB_METRIC_COSINE, D, queries.data(), NQ, base.data(), 300, out.data(), SDB_MEM_HOST, 0, nullptr); },
        [&](long, int) {}, [&] { CHECK(!memcmp(out.data(), out_w.data(), out.size() * 4)); });
    const uint32_t S = 3, per = 6, lim = 8;
    std::vector<uint64_t> mi(S * NQ * per), mo_w(NQ * lim), mo(NQ * lim);
    std::vector<float> md(S * NQ * per), mdo_w(NQ * lim), mdo(NQ * lim);
    std::vector<uint32_t> mc(S * NQ, per), msh(NQ * lim), mcnt_w(NQ), mcnt(NQ);
    for (size_t i = 0; i < mi.size(); i++) mi[i] = 1000 + i, md[i] = (float)((i * 7919) % 1000) + (float)(i % per) * 1000.f;
    OK(sdb_topk_merge(S, NQ, per, mi.data(), md.data(), mc.data(), lim, mo_w.data(), mdo_w.data(), msh.data(), mcnt_w.data(), SDB_MEM_HOST, 0, nullptr));
    sweep(
        "sdb_topk_merge", [] {},
        [&] { return sdb_topk_merge(S, NQ, per, mi.data(), md.data(), mc.data(), lim, mo.data(), mdo.data(), msh.data(), mcnt.data(), SDB_MEM_HOST, 0, nullptr); },
        [&](long, int) {}, [&] { CHECK(mo == mo_w && mcnt == mcnt_w && !memcmp(mdo.data(), mdo_w.data(), mdo.size() * 4)); });
    std::vector<float> X(base.begin(), base.begin() + 800 * D), cent_w(16 * 8), cent(16 * 8);
    std::vector<uint8_t> lab_w(800), lab(800);
    uint32_t it_w = 0, it = 0;
    {
      std::vector<float> X2(X);
      OK(sdb_kmeans_fit(X2.data(), 800, D, 8, 8, 16, 100, 5, 0, cent_w.data(), lab_w.data(), &it_w, SDB_MEM_HOST, 0, nullptr));
    }
    sweep(
        "sdb_kmeans_fit", [] {},
        [&] {
          std::vector<float> X2(X);
          return sdb_kmeans_fit(X2.data(), 800, D, 8, 8, 16, 100, 5, 0, cent.data(), lab.data(), &it, SDB_MEM_HOST, 0, nullptr);
        },
        [&](long, int) {}, [&] { CHECK(lab == lab_w && it == it_w && !memcmp(cent.data(), cent_w.data(), cent.size() * 4)); });
  }

  // ---- 7. create / quantizer objects
  {
    sdb_index_params p{};
    p.dim = D, p.metric = SDB_METRIC_COSINE, p.search_size = L, p.degree_bound = 32, p.alpha = 1.2f;
    sdb_index *ix = nullptr;
    sweep(
        "sdb_index_create", [] {}, [&] { return sdb_index_create(&p, &ix); }, [&](long, int) { CHECK(ix == nullptr); },
        [&] { CHECK(ix != nullptr); });
    OK(sdb_index_destroy(ix));
    sdb_pq *pq = nullptr;
    sweep(
        "sdb_pq_create", [] {}, [&] { return sdb_pq_create(D, SDB_METRIC_EUCLIDEAN, 8, 16, 0, &pq); },
        [&](long, int) { CHECK(pq == nullptr); }, [&] { CHECK(pq != nullptr); });
    OK(sdb_pq_destroy(pq));
  }

  if (g_fail) {
    std::printf("%d FAILURES\n", g_fail);
    return 1;
  }
  std::printf("ALL FAULT-INJECTION TESTS PASSED\n");
  return 0;
}
