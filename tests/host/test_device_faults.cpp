// test_device_faults.cpp -- the C ABI under DEVICE-memory exhaustion, the twin of test_faults.cpp (which throws
// std::bad_alloc at every host allocation).  The reference's contract is an `error`, never a panic
// (CONTRIBUTING.md:150), and a shard whose transaction failed is scrapped and rebuilt from its bucket
// (shard/cache/manager.go:231-240).
//
// hipMalloc, hipMallocAsync and hipHostMalloc are interposed BY THIS EXECUTABLE (it is first in the dynamic lookup
// order; built with -rdynamic): the k-th such call made FROM INSIDE libsemadb_amd.so on the armed thread returns
// hipErrorOutOfMemory -- the runtime's, rocPRIM's and RCCL's own allocations are left alone.  k is swept from 0 until
// the call goes through untouched, over
//     sdb_index_load, sdb_index_insert_batch (grows the table, the optional pair-distance cache), sdb_index_attach_pq
//     (the 2 x 64 x M x rows code-row copies), sdb_index_compact, sdb_index_search_batch (workspaces: bitsets, scratch,
//     filter buffers, quantizer table), sdb_cluster_search_batch (ring slots, staging, gathered buffer).
// After every injected failure:
//   - the call returned SDB_ERR_DEVICE (or SDB_OK when the allocation was an optional cache) with a message;
//   - the index answers a reference batch bit-identically to what it answered before, or -- a write that had begun to
//     change the graph -- is unusable (every call SDB_ERR_STATE) until reloaded; a cluster handle serves the next
//     request or says it is out of step;
// and when a section ends and its handles are destroyed, hipMemGetInfo is back at the section's baseline: nothing a
// failed call had allocated before the failing allocation stays behind.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <link.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/semadb_amd.h"

// ---------------------------------------------------------------------------------------------------------------
static uintptr_t g_lo = 0, g_hi = 0;  // address range of libsemadb_amd.so
static thread_local long t_countdown = -1;
static thread_local bool t_fired = false;
static thread_local long t_calls = 0;

static int find_lib(struct dl_phdr_info *info, size_t, void *) {
  if (!info->dlpi_name || !strstr(info->dlpi_name, "libsemadb_amd.so")) return 0;
  uintptr_t lo = ~(uintptr_t)0, hi = 0;
  for (int i = 0; i < info->dlpi_phnum; i++)
    if (info->dlpi_phdr[i].p_type == PT_LOAD) {
      const uintptr_t b = info->dlpi_addr + info->dlpi_phdr[i].p_vaddr, e = b + info->dlpi_phdr[i].p_memsz;
      lo = std::min(lo, b), hi = std::max(hi, e);
    }
  g_lo = lo, g_hi = hi;
  return 1;
}
static inline bool inject(void *ra) {
  const uintptr_t a = (uintptr_t)ra;
  if (a < g_lo || a >= g_hi) return false;
  t_calls++;
  if (t_countdown < 0) return false;
  if (t_countdown == 0) {
    t_countdown = -1, t_fired = true;
    return true;
  }
  t_countdown--;
  return false;
}
template <class F>
static F real(const char *name) {
  static F f = nullptr;
  if (!f) f = reinterpret_cast<F>(dlsym(RTLD_NEXT, name));
  if (!f) {
    std::fprintf(stderr, "cannot resolve %s\n", name);
    std::abort();
  }
  return f;
}
extern "C" {
__attribute__((noinline, visibility("default"))) hipError_t hipMalloc(void **p, size_t n) {
  if (inject(__builtin_return_address(0))) {
    if (p) *p = nullptr;
    return hipErrorOutOfMemory;
  }
  return real<hipError_t (*)(void **, size_t)>("hipMalloc")(p, n);
}
__attribute__((noinline, visibility("default"))) hipError_t hipMallocAsync(void **p, size_t n, hipStream_t s) {
  if (inject(__builtin_return_address(0))) {
    if (p) *p = nullptr;
    return hipErrorOutOfMemory;
  }
  return real<hipError_t (*)(void **, size_t, hipStream_t)>("hipMallocAsync")(p, n, s);
}
__attribute__((noinline, visibility("default"))) hipError_t hipHostMalloc(void **p, size_t n, unsigned int flags) {
  if (inject(__builtin_return_address(0))) {
    if (p) *p = nullptr;
    return hipErrorOutOfMemory;
  }
  return real<hipError_t (*)(void **, size_t, unsigned int)>("hipHostMalloc")(p, n, flags);
}
}
static void arm(long k) { t_fired = false, t_calls = 0, t_countdown = k; }
static bool disarm() {
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

constexpr uint32_t D = 64, NQ = 24, LIMIT = 10, L = 50, PQ_M = 8, PQ_K = 16;

struct Graph {
  std::vector<uint64_t> ids, offsets, edges;
  std::vector<float> vecs;
};
struct Answer {
  std::vector<uint64_t> ids;
  std::vector<float> d;
  std::vector<uint32_t> c;
  bool operator==(const Answer &o) const { return ids == o.ids && c == o.c && !memcmp(d.data(), o.d.data(), d.size() * 4); }
};
static sdb_index *new_index(uint64_t capacity = 2048) {
  sdb_index_params p{};
  p.dim = D, p.metric = SDB_METRIC_EUCLIDEAN, p.search_size = L, p.degree_bound = 32, p.alpha = 1.2f, p.device = 0;
  p.capacity = capacity;
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
static size_t free_bytes() {
  (void)hipDeviceSynchronize();
  hipMemPool_t pool = nullptr;
  if (hipDeviceGetDefaultMemPool(&pool, 0) == hipSuccess && pool) (void)hipMemPoolTrimTo(pool, 0);  // what hipFreeAsync parked
  size_t f = 0, t = 0;
  if (hipMemGetInfo(&f, &t) != hipSuccess) return 0;
  return f;
}
// a section: everything it creates is destroyed inside; afterwards the device holds what it held before
struct LeakCheck {
  const char *name;
  size_t before;
  explicit LeakCheck(const char *n) : name(n), before(free_bytes()) {}
  ~LeakCheck() {
    const size_t after = free_bytes();
    const long long lost = (long long)before - (long long)after;
    if (lost > (4ll << 20)) std::printf("FAIL %s: %lld bytes of device memory did not come back\n", name, lost), g_fail++;
    else std::printf("    %-26s device memory back at its baseline (%+lld bytes)\n", name, -lost);
  }
};

template <class Fresh, class Call, class AfterFail, class AfterOk>
static long sweep(const char *name, Fresh fresh, Call call, AfterFail after_fail, AfterOk after_ok, long max_k = 400) {
  long injected = 0, absorbed = 0;
  for (long k = 0; k < max_k; k++) {
    fresh();
    arm(k);
    const int rc = call();
    const bool fired = disarm();
    if (!fired) {
      if (rc != SDB_OK) std::printf("FAIL %s: undisturbed call returned %d (%s)\n", name, rc, sdb_last_error()), g_fail++;
      after_ok();
      std::printf("%-30s %ld device allocations failed in turn: %ld -> a status, %ld absorbed (optional buffers); final call ok\n",
                  name, injected, injected - absorbed, absorbed);
      return injected;
    }
    injected++;
    if (rc == SDB_OK) {  // an optional buffer (a cache): the call went on without it and must still be right
      absorbed++;
      after_ok();
      continue;
    }
    if (rc != SDB_ERR_DEVICE) std::printf("FAIL %s k=%ld: status %d, expected SDB_ERR_DEVICE (%s)\n", name, k, rc, sdb_last_error()), g_fail++;
    if (!sdb_last_error()[0]) std::printf("FAIL %s k=%ld: status %d without a message\n", name, k, rc), g_fail++;
    after_fail(k, rc);
  }
  std::printf("FAIL %s: still failing after %ld countdowns\n", name, max_k);
  g_fail++;
  return injected;
}

int main() {
  int ndev = 0;
  if (sdb_device_count(&ndev) != SDB_OK) {
    std::printf("no GPU: %s\n", sdb_last_error());
    return 2;
  }
  dl_iterate_phdr(find_lib, nullptr);
  if (!g_lo) {
    std::printf("FAIL: libsemadb_amd.so not found among the loaded objects\n");
    return 1;
  }
  std::mt19937 rng(20251005);
  std::normal_distribution<float> nd;
  const uint32_t N = 2500, NEXTRA = 1800;  // the extra inserts push the table past its first capacity: reserve() runs
  std::vector<float> base(N * D), extra(NEXTRA * D), start(D), queries(NQ * D);
  for (auto &x : base) x = nd(rng);
  for (auto &x : extra) x = nd(rng);
  for (auto &x : start) x = nd(rng);
  for (auto &x : queries) x = nd(rng);
  std::vector<uint64_t> base_ids(N), extra_ids(NEXTRA);
  for (uint32_t i = 0; i < N; i++) base_ids[i] = 2 + (uint64_t)i + i / 2;
  for (uint32_t i = 0; i < NEXTRA; i++) extra_ids[i] = 100000 + 3 * (uint64_t)i;

  // ---- the undisturbed run
  sdb_index *ref = new_index(8192);
  OK(sdb_index_set_start(ref, start.data(), SDB_MEM_HOST));
  OK(sdb_index_insert_batch(ref, N, base_ids.data(), base.data(), SDB_MEM_HOST, 0, nullptr));
  const Graph g0 = export_graph(ref);
  Answer a0;
  OK(search(ref, queries, &a0));
  std::vector<uint64_t> foff(NQ + 1), fids;
  for (uint32_t q = 0; q < NQ; q++) {
    foff[q] = fids.size();
    for (uint32_t i = q % 5; i < N; i += 5 + q % 3) fids.push_back(base_ids[i]);
  }
  foff[NQ] = fids.size();
  Answer af0;
  OK(search(ref, queries, &af0, &foff, &fids));
  OK(sdb_index_insert_batch(ref, NEXTRA, extra_ids.data(), extra.data(), SDB_MEM_HOST, 0, nullptr));
  const Graph g1 = export_graph(ref);
  Answer a1;
  OK(search(ref, queries, &a1));
  std::vector<uint64_t> del_ids;
  for (uint32_t i = 0; i < N; i += 3) del_ids.push_back(base_ids[i]);
  OK(sdb_index_delete_batch(ref, del_ids.size(), del_ids.data(), nullptr));
  const Graph g2 = export_graph(ref);
  Answer a2;
  OK(search(ref, queries, &a2));
  // a fitted quantizer (M = 8: the neighbours' code rows sit behind the adjacency rows) and the answers over it
  sdb_pq *pq = nullptr;
  OK(sdb_pq_create(D, SDB_METRIC_EUCLIDEAN, PQ_M, PQ_K, 0, &pq));
  {
    std::vector<float> train(base.begin(), base.begin() + 1000 * D);
    std::vector<uint32_t> first(PQ_M);
    for (uint32_t i = 0; i < PQ_M; i++) first[i] = i * 37 % 1000;
    OK(sdb_pq_fit(pq, train.data(), 1000, first.data(), 0, nullptr, SDB_MEM_HOST, nullptr));
  }
  sdb_index *refq = new_index(8192);
  OK(load_graph(refq, g0));
  OK(sdb_index_attach_pq(refq, pq, nullptr));
  Answer aq0;
  OK(search(refq, queries, &aq0));
  OK(sdb_index_destroy(refq));
  OK(sdb_index_destroy(ref));
  auto same_graph = [](const Graph &a, const Graph &b) {
    return a.ids == b.ids && a.offsets == b.offsets && a.edges == b.edges && a.vecs == b.vecs;
  };

  // ---- 1. load: a failed load leaves the index empty and loadable
  {
    LeakCheck lc("sdb_index_load");
    sdb_index *ix = new_index(16);  // small: the load has to grow every table
    sweep(
        "sdb_index_load", [] {}, [&] { return load_graph(ix, g0); },
        [&](long k, int) {
          uint64_t n = 1;
          OK(sdb_index_stats(ix, &n, nullptr, nullptr));
          if (n != 0) std::printf("FAIL load k=%ld: %llu rows left behind\n", k, (unsigned long long)n), g_fail++;
        },
        [&] {
          Answer a;
          OK(search(ix, queries, &a));
          CHECK(a == a0);
        });
    OK(sdb_index_destroy(ix));
  }

  // ---- 2. search_batch on a handle that has no workspace yet: plain, filtered, quantized
  {
    LeakCheck lc("sdb_index_search_batch");
    sdb_index *ix = nullptr;
    auto fresh = [&] {  // a new handle every time: its workspaces (bitsets, scratch, filter buffers) are allocated by the call
      if (ix) OK(sdb_index_destroy(ix));
      ix = new_index(8192);
      OK(load_graph(ix, g0));
    };
    Answer a;
    sweep(
        "search_batch (first call)", fresh, [&] { return search(ix, queries, &a); },
        [&](long, int) {
          Answer b;
          OK(search(ix, queries, &b));  // the same handle serves the next request
          CHECK(b == a0);
        },
        [&] { CHECK(a == a0); });
    sweep(
        "search_batch (filtered)", fresh, [&] { return search(ix, queries, &a, &foff, &fids); },
        [&](long, int) {
          Answer b;
          OK(search(ix, queries, &b, &foff, &fids));
          CHECK(b == af0);
        },
        [&] { CHECK(a == af0); });
    sweep(
        "search_batch (bitset walk)", fresh,
        [&] {  // searchSize > 96: the HBM bitset from the start (ensure_bitsets + memset)
          a.ids.assign(NQ * LIMIT, 0), a.d.assign(NQ * LIMIT, 0.f), a.c.assign(NQ, 0);
          return sdb_index_search_batch(ix, NQ, queries.data(), LIMIT, 120, nullptr, nullptr, a.ids.data(), a.d.data(), a.c.data(),
                                        nullptr, SDB_MEM_HOST, nullptr);
        },
        [&](long, int) {
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == a0);
        },
        [&] {});
    if (ix) OK(sdb_index_destroy(ix));
  }

  // ---- 3. attach_pq: the code table and the 2 x 64 x M x rows code-row copies; then the quantized search's table
  {
    LeakCheck lc("sdb_index_attach_pq");
    sdb_index *ix = nullptr;
    auto fresh = [&] {
      if (ix) OK(sdb_index_destroy(ix));
      ix = new_index(8192);
      OK(load_graph(ix, g0));
    };
    sweep(
        "sdb_index_attach_pq", fresh, [&] { return sdb_index_attach_pq(ix, pq, nullptr); },
        [&](long k, int) {
          Answer b;  // the full-precision store answers as before, or the handle says it is unusable
          const int rc = search(ix, queries, &b);
          if (rc == SDB_OK) {
            if (!(b == a0)) std::printf("FAIL attach k=%ld: a failed attach changed the answers\n", k), g_fail++;
          } else if (rc != SDB_ERR_STATE) {
            std::printf("FAIL attach k=%ld: search after a failed attach -> %d (%s)\n", k, rc, sdb_last_error()), g_fail++;
          }
        },
        [&] {  // attached -- with or without the optional code-row copies: the same walk
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == aq0);
        });
    Answer a;
    sweep(
        "search_batch (quantized)",
        [&] {
          fresh();
          OK(sdb_index_attach_pq(ix, pq, nullptr));
        },
        [&] { return search(ix, queries, &a); },
        [&](long, int) {
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == aq0);
        },
        [&] { CHECK(a == aq0); });
    if (ix) OK(sdb_index_destroy(ix));
  }

  // ---- 4. insert_batch that has to grow the table: as it was, or unusable; never half a transaction
  {
    LeakCheck lc("sdb_index_insert_batch");
    sdb_index *ix = nullptr;
    long unusable = 0, intact = 0;
    sweep(
        "sdb_index_insert_batch",
        [&] {
          if (ix) OK(sdb_index_destroy(ix));
          ix = new_index(N + 8);  // no room for the extra rows: reserve() reallocates every table
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
    // ---- 5. delete_batch and compact on the grown graph
    unusable = intact = 0;
    sweep(
        "sdb_index_delete_batch",
        [&] {
          if (ix) OK(sdb_index_destroy(ix));
          ix = new_index(8192);
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
        },
        [&] {
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == a2);
        });
    std::printf("    delete: %ld failures left the index as it was, %ld left it unusable (reload)\n", intact, unusable);
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
    if (ix) OK(sdb_index_destroy(ix));
  }

  // ---- 6. compact of a quantized index: the renumbered code rows are optional, the rest is not
  {
    LeakCheck lc("compact (quantized)");
    sdb_index *ix = nullptr;
    Answer want;
    bool have = false;
    sweep(
        "sdb_index_compact (quantized)",
        [&] {
          if (ix) OK(sdb_index_destroy(ix));
          ix = new_index(8192);
          OK(load_graph(ix, g2));  // (no tombstones in a loaded graph: delete some rows to make them)
          OK(sdb_index_attach_pq(ix, pq, nullptr));
          OK(sdb_index_delete_batch(ix, 200, extra_ids.data(), nullptr));
          if (!have) {
            OK(search(ix, queries, &want));
            have = true;
          }
        },
        [&] { return sdb_index_compact(ix); },
        [&](long, int) {
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == want);
        },
        [&] {
          Answer b;
          OK(search(ix, queries, &b));
          CHECK(b == want);
        });
    if (ix) OK(sdb_index_destroy(ix));
  }

  // ---- 7. sdb_cluster_search_batch: two shards of one GPU, the fault on rank 0's thread
  {
    // (one undisturbed create / request / destroy cycle first: what the runtime keeps for streams and events it has
    // seen once is not this library's to give back, and must not be counted against the faulty calls)
    {
      sdb_index *w[2];
      sdb_cluster *wc[2] = {nullptr, nullptr};
      const int wd[2] = {0, 0};
      for (int r = 0; r < 2; r++) {
        w[r] = new_index(8192);
        OK(sdb_index_set_start(w[r], start.data(), SDB_MEM_HOST));
        OK(sdb_index_insert_batch(w[r], 600, base_ids.data() + r * 600, base.data() + (size_t)r * 600 * D, SDB_MEM_HOST, 0, nullptr));
      }
      OK(sdb_cluster_create_local(2, wd, wc));
      Answer o[2];
      int wrc[2] = {0, 0};
      std::thread peer([&] {
        o[1].ids.assign(NQ * LIMIT, 0), o[1].d.assign(NQ * LIMIT, 0.f), o[1].c.assign(NQ, 0);
        wrc[1] = sdb_cluster_search_batch(wc[1], w[1], 0, NQ, queries.data(), LIMIT, L, o[1].ids.data(), o[1].d.data(), nullptr,
                                          o[1].c.data(), SDB_MEM_HOST, nullptr);
      });
      o[0].ids.assign(NQ * LIMIT, 0), o[0].d.assign(NQ * LIMIT, 0.f), o[0].c.assign(NQ, 0);
      wrc[0] = sdb_cluster_search_batch(wc[0], w[0], 0, NQ, queries.data(), LIMIT, L, o[0].ids.data(), o[0].d.data(), nullptr,
                                        o[0].c.data(), SDB_MEM_HOST, nullptr);
      peer.join();
      CHECK(wrc[0] == SDB_OK && wrc[1] == SDB_OK);
      for (auto *c : wc) OK(sdb_cluster_destroy(c));
      for (auto *x : w) OK(sdb_index_destroy(x));
    }
    LeakCheck lc("sdb_cluster_search_batch");
    const uint32_t H = N / 2;
    sdb_index *sh[2];
    for (int r = 0; r < 2; r++) {
      sh[r] = new_index(8192);
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
        if (c) OK(sdb_cluster_destroy(c));
        c = nullptr;
      }
    };
    auto request = [&](long k0, Answer out[2], int rc[2]) {
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
    Answer good[2];
    int rc[2];
    request(-1, good, rc);
    CHECK(rc[0] == SDB_OK && rc[1] == SDB_OK && good[0] == good[1]);
    drop();
    long injected = 0, recreated = 0;
    for (long k = 0; k < 200; k++) {
      make();  // fresh handles: the request allocates its ring slot, staging and the gathered buffer
      Answer out[2];
      const bool fired = request(k, out, rc);
      if (!fired) {
        CHECK(rc[0] == SDB_OK && rc[1] == SDB_OK && out[0] == good[0]);
        drop();
        break;
      }
      injected++;
      if (rc[0] == SDB_OK) {
        CHECK(out[0] == good[0]);
      } else {
        CHECK(rc[0] == SDB_ERR_DEVICE || rc[0] == SDB_ERR_STATE);
        CHECK(sdb_last_error()[0] != 0);
        // the rank that failed stayed outside the exchange: its peer gave up after the deadline -- an error, not a hang
        CHECK(rc[1] != SDB_OK || out[1] == good[0]);
      }
      // the next request: served by the same handles, or they say they are out of step and fresh ones serve it
      Answer nxt[2];
      request(-1, nxt, rc);
      if (!(rc[0] == SDB_OK && rc[1] == SDB_OK && nxt[0] == good[0])) {
        CHECK(rc[0] == SDB_ERR_STATE || rc[1] == SDB_ERR_STATE || rc[0] == SDB_ERR_DEVICE);
        drop();
        make();
        recreated++;
        request(-1, nxt, rc);
        CHECK(rc[0] == SDB_OK && rc[1] == SDB_OK && nxt[0] == good[0]);
      }
      drop();
    }
    std::printf("%-30s %ld device allocations failed in turn; the handles had to be recreated %ld times\n",
                "sdb_cluster_search_batch", injected, recreated);
    for (auto *s : sh) OK(sdb_index_destroy(s));
  }
  // ---- 8. the two-precision hop's float16 copy of the rows (SDB_TUNE_SKETCH): an optional cache -- switching it on, and
  // a commit that has to grow it, without room for it must leave searches on float32 rows with the same answers
  {
    LeakCheck lc("SDB_TUNE_SKETCH");
    auto cos_index = [&](uint64_t capacity) {
      sdb_index_params p{};
      p.dim = D, p.metric = SDB_METRIC_COSINE, p.search_size = L, p.degree_bound = 32, p.alpha = 1.2f, p.device = 0;
      p.capacity = capacity;
      sdb_index *x = nullptr;
      OK(sdb_index_create(&p, &x));
      OK(sdb_index_set_tuning(x, SDB_TUNE_WIDE_WALK, 1));  // the batch walk has the stage
      OK(sdb_index_set_start(x, start.data(), SDB_MEM_HOST));
      OK(sdb_index_insert_batch(x, N, base_ids.data(), base.data(), SDB_MEM_HOST, 0, nullptr));
      return x;
    };
    sdb_index *cref = cos_index(8192);
    Answer c0, c1;
    OK(search(cref, queries, &c0));
    OK(sdb_index_insert_batch(cref, NEXTRA, extra_ids.data(), extra.data(), SDB_MEM_HOST, 0, nullptr));
    OK(search(cref, queries, &c1));
    OK(sdb_index_destroy(cref));
    sdb_index *ix = nullptr;
    Answer a;
    auto fresh = [&] {
      if (ix) OK(sdb_index_destroy(ix));
      ix = cos_index(4096);  // (the extra inserts below push the table past this capacity)
    };
    sweep(
        "set_tuning(SDB_TUNE_SKETCH)", fresh, [&] { return sdb_index_set_tuning(ix, SDB_TUNE_SKETCH, 2); },
        [&](long, int) { CHECK(false); },  // nothing here may fail the call: the copy is optional
        [&] {
          OK(search(ix, queries, &a));
          CHECK(a == c0);
          uint64_t st[3] = {0, 0, 0};
          OK(sdb_index_sketch_stats(ix, st));
          CHECK(st[1] == 0);
        });
    auto fresh_on = [&] {
      fresh();
      OK(sdb_index_set_tuning(ix, SDB_TUNE_SKETCH, 2));
    };
    sweep(
        "insert_batch with the copy", fresh_on,
        [&] { return sdb_index_insert_batch(ix, NEXTRA, extra_ids.data(), extra.data(), SDB_MEM_HOST, 0, nullptr); },
        [&](long, int) {  // as it was, or unusable and said so (section 4's contract)
          if (!is_broken(ix, queries)) {
            Answer b;
            OK(search(ix, queries, &b));
            CHECK(b == c0);
          }
        },
        [&] {
          OK(search(ix, queries, &a));
          CHECK(a == c1);
          uint64_t st[3] = {0, 0, 0};
          OK(sdb_index_sketch_stats(ix, st));
          CHECK(st[1] == 0);
        });
    if (ix) OK(sdb_index_destroy(ix));
  }
  OK(sdb_pq_destroy(pq));
  std::printf("%s (%d failures)\n", g_fail ? "FAILED" : "all device-memory faults ended in a status", g_fail);
  return g_fail ? 1 : 0;
}
