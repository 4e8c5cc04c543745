// mock_sdb.cpp -- a CPU stand-in for libsemadb_amd.so, TEST INFRASTRUCTURE ONLY: it implements the entry points of
// include/semadb_amd.h that the host side calls (semadb_amd/host/semadb_host.hpp: SearchBatcher, IndexVamana,
// IndexFlat, GpuFanout) without a GPU, so that the product's concurrent host code can run under ThreadSanitizer and
// AddressSanitizer on the CPU box (tests/host/test_concurrency.cpp, tests/test_host_concurrency.py).  Nothing in the
// product links it.
//
// A search answers after a short sleep with a pure function of the query (mock_expect below), so every answer a
// caller gets back can be checked against the request it made -- a batcher that hands a caller another caller's slot
// shows up as a wrong id, not only as a data race.  The mock also watches its own calling contract: two writers at
// once, output buffers shared by two calls in flight, more queries than the host announced.
//
// The shard exchange (sdb_cluster_*) is NOT mocked away: ticket order, ring slots and the group rendezvous are the
// product's own code (semadb_amd/csrc/turnstile.h); only the device work behind them (walk, copies, merge kernel)
// is replaced by host loops here.
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../../semadb_amd/csrc/turnstile.h"
#include "mock_sdb.h"

namespace {
thread_local char t_err[1024];
std::atomic<uint64_t> g_violations{0}, g_search_calls{0}, g_host_allocs{0};
std::atomic<uint32_t> g_lat_lo_us{100}, g_lat_hi_us{400};
std::atomic<int> g_fail_host_alloc{0};
std::mutex g_viol_mu;
std::string g_first_violation;

void violation(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_violations++;
  std::lock_guard<std::mutex> g(g_viol_mu);
  if (g_first_violation.empty()) g_first_violation = buf;
}

void nap() {
  thread_local std::minstd_rand rng((unsigned)std::hash<std::thread::id>()(std::this_thread::get_id()));
  const uint32_t lo = g_lat_lo_us.load(), hi = g_lat_hi_us.load();
  if (!hi) return;
  const uint32_t us = lo + (hi > lo ? rng() % (hi - lo + 1) : 0);
  std::this_thread::sleep_for(std::chrono::microseconds(us));
}

// output buffers of the calls in flight: two calls that write the same memory are a host bug
struct InFlight {
  std::mutex mu;
  std::multimap<const char *, const char *> ranges;  // begin -> end
  bool add(const void *p, size_t bytes) {
    const char *b = (const char *)p, *e = b + bytes;
    std::lock_guard<std::mutex> g(mu);
    for (auto &r : ranges)
      if (b < r.second && r.first < e) return false;
    ranges.emplace(b, e);
    return true;
  }
  void drop(const void *p) {
    std::lock_guard<std::mutex> g(mu);
    auto it = ranges.find((const char *)p);
    if (it != ranges.end()) ranges.erase(it);
  }
} g_inflight;
}  // namespace

namespace sdb {
int fail(int code, const char *fmt, ...) noexcept {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_err, sizeof(t_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace sdb
using sdb::fail;

struct sdb_index {
  sdb_index_params P{};
  std::mutex mu;  // the table
  std::vector<uint64_t> ids;
  std::vector<float> vecs;
  std::vector<uint8_t> codes;
  uint32_t M = 0;
  bool has_start = false, in_tx = false;
  uint64_t dead = 0;
  std::atomic<int> writers{0};
  std::vector<uint64_t> tx_ids;  // snapshot for abort
  std::vector<float> tx_vecs;
};
struct sdb_pq {
  uint32_t dim, metric, M, K;
  std::vector<float> fc;
};

// ---------------------------------------------------------------------------------------------------------------
// the answer of the mock to one query: ids base + j, distances q[1] + j; with a filter the first `limit` filter ids
void mock_expect(const float *q, uint32_t dim, uint32_t limit, const uint64_t *filter, uint64_t n_filter, uint64_t *ids,
                 float *dists, uint32_t *count) {
  uint32_t b0;
  memcpy(&b0, &q[0], 4);
  const uint64_t base = ((uint64_t)b0 << 8) ^ (dim > 2 ? (uint64_t)(int64_t)q[2] : 0);
  const float d0 = dim > 1 ? q[1] : 0.f;
  uint32_t n = limit;
  if (filter) n = (uint32_t)std::min<uint64_t>(limit, n_filter);
  for (uint32_t j = 0; j < n; j++) {
    ids[j] = filter ? filter[j] : base + j + 2;
    dists[j] = d0 + (float)j;
  }
  *count = n;
}
void mock_set_latency_us(uint32_t lo, uint32_t hi) { g_lat_lo_us = lo, g_lat_hi_us = hi; }
void mock_fail_host_alloc(int on) { g_fail_host_alloc = on; }
uint64_t mock_violations(void) { return g_violations.load(); }
const char *mock_first_violation(void) {
  std::lock_guard<std::mutex> g(g_viol_mu);
  static thread_local std::string copy;
  copy = g_first_violation;
  return copy.c_str();
}
uint64_t mock_search_calls(void) { return g_search_calls.load(); }

extern "C" {

const char *sdb_last_error(void) { return t_err; }
int sdb_abi_version(void) { return SDB_ABI_VERSION; }
int sdb_device_count(int *count) {
  if (!count) return fail(SDB_ERR_INVALID, "count is NULL");
  *count = 8;
  return SDB_OK;
}
int sdb_host_alloc(size_t bytes, void **out) {
  if (!out) return fail(SDB_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (g_fail_host_alloc.load()) return fail(SDB_ERR_DEVICE, "mock: no pinned memory");
  *out = std::malloc(bytes ? bytes : 1);
  g_host_allocs++;
  return *out ? SDB_OK : fail(SDB_ERR_DEVICE, "out of host memory");
}
int sdb_host_free(void *p) {
  std::free(p);
  return SDB_OK;
}

int sdb_distance_batch(int metric, uint32_t dim, const float *queries, uint64_t nq, const float *candidates, uint64_t nc,
                       float *out, int, int, void *) {
  for (uint64_t q = 0; q < nq; q++)
    for (uint64_t c = 0; c < nc; c++) {
      float s = 0;
      for (uint32_t i = 0; i < dim; i++) {
        const float x = queries[q * dim + i], y = candidates[c * dim + i];
        s += metric == SDB_METRIC_EUCLIDEAN ? (x - y) * (x - y) : x * y;
      }
      out[q * nc + c] = metric == SDB_METRIC_EUCLIDEAN ? s : metric == SDB_METRIC_COSINE ? 1 - s : -s;
    }
  return SDB_OK;
}

int sdb_index_create(const sdb_index_params *p, sdb_index **out) {
  if (!p || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  if (p->dim < 1 || p->dim > 4096) return fail(SDB_ERR_INVALID, "vector size must be between 1 and 4096, got %u", p->dim);
  auto *ix = new sdb_index();
  ix->P = *p;
  *out = ix;
  return SDB_OK;
}
int sdb_index_destroy(sdb_index *ix) {
  delete ix;
  return SDB_OK;
}
int sdb_index_set_start(sdb_index *ix, const float *vec, int) {
  if (!ix || !vec) return fail(SDB_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ix->mu);
  if (!ix->ids.empty()) return fail(SDB_ERR_STATE, "start node must be the first node of an empty index");
  ix->ids.push_back(SDB_STARTID);
  ix->vecs.assign(vec, vec + ix->P.dim);
  ix->has_start = true;
  return SDB_OK;
}
int sdb_index_load(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors, const uint64_t *offsets,
                   const uint64_t *, int) {
  if (!ix || !vectors || !offsets) return fail(SDB_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ix->mu);
  ix->ids.clear();
  for (uint64_t i = 0; i < n; i++) ix->ids.push_back(ids ? ids[i] : i + 1);
  ix->vecs.assign(vectors, vectors + n * ix->P.dim);
  ix->has_start = true;
  return SDB_OK;
}

// one writer at a time is the host's duty (the shard's write lock): the mock counts who breaks it
struct WriterGuard {
  sdb_index *ix;
  explicit WriterGuard(sdb_index *i) : ix(i) {
    if (ix->writers.fetch_add(1) != 0) violation("two writers inside one index at once");
  }
  ~WriterGuard() { ix->writers.fetch_sub(1); }
};

int sdb_index_begin_write(sdb_index *ix) {
  if (!ix) return fail(SDB_ERR_INVALID, "NULL handle");
  WriterGuard w(ix);
  std::lock_guard<std::mutex> g(ix->mu);
  if (ix->in_tx) return fail(SDB_ERR_STATE, "a write transaction is already open");
  ix->in_tx = true;
  ix->tx_ids = ix->ids, ix->tx_vecs = ix->vecs;
  return SDB_OK;
}
int sdb_index_commit(sdb_index *ix, void *) {
  if (!ix) return fail(SDB_ERR_INVALID, "NULL handle");
  WriterGuard w(ix);
  nap();
  std::lock_guard<std::mutex> g(ix->mu);
  ix->in_tx = false;
  return SDB_OK;
}
int sdb_index_abort_write(sdb_index *ix) {
  if (!ix) return fail(SDB_ERR_INVALID, "NULL handle");
  WriterGuard w(ix);
  std::lock_guard<std::mutex> g(ix->mu);
  if (ix->in_tx) ix->ids = ix->tx_ids, ix->vecs = ix->tx_vecs;
  ix->in_tx = false;
  return SDB_OK;
}
int sdb_index_insert_batch(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors, int, uint32_t, void *) {
  if (!ix || !vectors) return fail(SDB_ERR_INVALID, "NULL argument");
  WriterGuard w(ix);
  nap();
  std::lock_guard<std::mutex> g(ix->mu);
  if (!ix->has_start) return fail(SDB_ERR_STATE, "no start node");
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t id = ids ? ids[i] : ix->ids.size() + 1;
    if (id < 2) return fail(SDB_ERR_INVALID, "invalid point id: %llu", (unsigned long long)id);
    ix->ids.push_back(id);
    ix->vecs.insert(ix->vecs.end(), vectors + i * ix->P.dim, vectors + (i + 1) * ix->P.dim);
  }
  return SDB_OK;
}
static void remove_ids(sdb_index *ix, uint64_t n, const uint64_t *ids) {
  const size_t d = ix->P.dim;
  for (uint64_t k = 0; k < n; k++)
    for (size_t i = 0; i < ix->ids.size(); i++)
      if (ix->ids[i] == ids[k] && ids[k] != SDB_STARTID) {
        ix->ids.erase(ix->ids.begin() + (long)i);
        ix->vecs.erase(ix->vecs.begin() + (long)(i * d), ix->vecs.begin() + (long)((i + 1) * d));
        ix->dead++;
        break;
      }
}
int sdb_index_delete_batch(sdb_index *ix, uint64_t n, const uint64_t *ids, void *) {
  if (!ix || (n && !ids)) return fail(SDB_ERR_INVALID, "NULL argument");
  WriterGuard w(ix);
  nap();
  std::lock_guard<std::mutex> g(ix->mu);
  remove_ids(ix, n, ids);
  return SDB_OK;
}
int sdb_index_set_vectors(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors, int) {
  if (!ix || !vectors) return fail(SDB_ERR_INVALID, "NULL argument");
  WriterGuard w(ix);
  std::lock_guard<std::mutex> g(ix->mu);
  if (ids) remove_ids(ix, n, ids);
  for (uint64_t i = 0; i < n; i++) {
    ix->ids.push_back(ids ? ids[i] : ix->ids.size() + 2);
    ix->vecs.insert(ix->vecs.end(), vectors + i * ix->P.dim, vectors + (i + 1) * ix->P.dim);
  }
  return SDB_OK;
}
int sdb_index_remove_vectors(sdb_index *ix, uint64_t n, const uint64_t *ids) {
  if (!ix) return fail(SDB_ERR_INVALID, "NULL argument");
  WriterGuard w(ix);
  std::lock_guard<std::mutex> g(ix->mu);
  remove_ids(ix, n, ids);
  return SDB_OK;
}
int sdb_index_size_in_memory(const sdb_index *ix, int64_t *bytes) {
  if (!ix || !bytes) return fail(SDB_ERR_INVALID, "NULL argument");
  auto *m = const_cast<sdb_index *>(ix);
  std::lock_guard<std::mutex> g(m->mu);
  *bytes = (int64_t)(ix->vecs.size() * 4 + ix->ids.size() * 8);
  return SDB_OK;
}
int sdb_index_stats(const sdb_index *ix, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *max_node_id) {
  if (!ix) return fail(SDB_ERR_INVALID, "NULL argument");
  auto *m = const_cast<sdb_index *>(ix);
  std::lock_guard<std::mutex> g(m->mu);
  uint64_t mx = 0;
  for (uint64_t id : ix->ids) mx = std::max(mx, id);
  if (n_nodes) *n_nodes = ix->ids.size();
  if (n_edges) *n_edges = 0;
  if (max_node_id) *max_node_id = mx;
  return SDB_OK;
}
int sdb_index_row_usage(const sdb_index *ix, uint64_t *rows, uint64_t *dead) {
  if (!ix) return fail(SDB_ERR_INVALID, "NULL argument");
  auto *m = const_cast<sdb_index *>(ix);
  std::lock_guard<std::mutex> g(m->mu);
  if (rows) *rows = ix->ids.size() + ix->dead;
  if (dead) *dead = ix->dead;
  return SDB_OK;
}
int sdb_index_compact(sdb_index *ix) {
  if (!ix) return fail(SDB_ERR_INVALID, "NULL argument");
  WriterGuard w(ix);
  nap();
  std::lock_guard<std::mutex> g(ix->mu);
  ix->dead = 0;
  return SDB_OK;
}
int sdb_index_export(const sdb_index *ix, uint64_t *ids, float *vectors, uint64_t *offsets, uint64_t *) {
  if (!ix || !ids || !offsets) return fail(SDB_ERR_INVALID, "NULL argument");
  auto *m = const_cast<sdb_index *>(ix);
  std::lock_guard<std::mutex> g(m->mu);
  memcpy(ids, ix->ids.data(), ix->ids.size() * 8);
  if (vectors) memcpy(vectors, ix->vecs.data(), ix->vecs.size() * 4);
  for (size_t i = 0; i <= ix->ids.size(); i++) offsets[i] = 0;
  return SDB_OK;
}

static int mock_search(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit, uint32_t search_size,
                       const uint64_t *f_off, const uint64_t *f_ids, uint64_t *out_ids, float *out_dists,
                       uint32_t *out_counts, bool sleep) {
  if (!ix || (nq && (!queries || !out_ids || !out_dists || !out_counts))) return fail(SDB_ERR_INVALID, "NULL argument");
  if (limit < 1) return fail(SDB_ERR_INVALID, "invalid limit %u", limit);
  if (search_size < limit) return fail(SDB_ERR_INVALID, "searchSize (%u) must be greater than k (%u)", search_size, limit);
  g_search_calls++;
  const bool tracked = nq != 0;
  if (tracked && !g_inflight.add(out_ids, nq * limit * 8)) violation("two searches in flight write the same id buffer");
  if (tracked && !g_inflight.add(out_counts, nq * 4)) violation("two searches in flight write the same count buffer");
  if (sleep) nap();
  int rc = SDB_OK;
  if (search_size == MOCK_FAILING_SEARCH_SIZE) {
    rc = fail(SDB_ERR_DEVICE, "mock: the device failed this batch (searchSize %u)", search_size);
    for (uint64_t q = 0; q < nq; q++) out_counts[q] = 0;
  } else {
    const uint32_t dim = ix->P.dim;
    for (uint64_t q = 0; q < nq; q++)
      mock_expect(queries + q * dim, dim, limit, f_off ? f_ids + f_off[q] : nullptr, f_off ? f_off[q + 1] - f_off[q] : 0,
                  out_ids + q * limit, out_dists + q * limit, out_counts + q);
  }
  if (tracked) g_inflight.drop(out_ids), g_inflight.drop(out_counts);
  return rc;
}

int sdb_index_search_batch(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit, uint32_t search_size,
                           const uint64_t *filter_offsets, const uint64_t *filter_ids, uint64_t *out_ids, float *out_dists,
                           uint32_t *out_counts, const sdb_search_trace *, int, void *) {
  return mock_search(ix, nq, queries, limit, search_size, filter_offsets, filter_ids, out_ids, out_dists, out_counts, true);
}
int sdb_index_search_batch_bitmap(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit, uint32_t search_size,
                                  const uint64_t *first, const uint64_t *w_off, const uint64_t *words, uint64_t *out_ids,
                                  float *out_dists, uint32_t *out_counts, const sdb_search_trace *, int, void *) {
  if (!first || !w_off || !words) return fail(SDB_ERR_INVALID, "NULL argument");
  std::vector<uint64_t> off{0}, ids;  // the bitmaps, expanded to ascending id lists
  for (uint64_t q = 0; q < nq; q++) {
    for (uint64_t w = w_off[q]; w < w_off[q + 1]; w++)
      for (int b = 0; b < 64; b++)
        if (words[w] >> b & 1) ids.push_back(first[q] + (w - w_off[q]) * 64 + (uint64_t)b);
    off.push_back(ids.size());
  }
  if (ids.empty()) ids.push_back(0);
  return mock_search(ix, nq, queries, limit, search_size, off.data(), ids.data(), out_ids, out_dists, out_counts, true);
}
int sdb_index_flat_search(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit, const uint64_t *f_off,
                          const uint64_t *f_ids, uint64_t *out_ids, float *out_dists, uint32_t *out_counts, int, void *) {
  return mock_search(ix, nq, queries, limit, limit, f_off, f_ids, out_ids, out_dists, out_counts, true);
}

int sdb_shard_limit(uint32_t limit, uint32_t n_shards, uint32_t max_search_limit, uint32_t *out) {  // actions.go:291-299
  if (!out || !n_shards) return fail(SDB_ERR_INVALID, "bad argument");
  const uint32_t target = (uint32_t)((float)limit * (1.0f / (float)n_shards) * 1.42f + 10);
  *out = std::min(std::min(limit, max_search_limit), target);
  return SDB_OK;
}

// ---- product quantizer: enough for IndexVamana::fit / flush / loadFromBucket to run
int sdb_pq_create(uint32_t dim, uint32_t metric, uint32_t M, uint32_t K, int, sdb_pq **out) {
  if (!out || !M || dim % M) return fail(SDB_ERR_INVALID, "bad quantizer shape");
  *out = new sdb_pq{dim, metric, M, K, std::vector<float>((size_t)K * dim, 0.f)};
  return SDB_OK;
}
int sdb_pq_destroy(sdb_pq *pq) {
  delete pq;
  return SDB_OK;
}
int sdb_pq_fit(sdb_pq *pq, float *X, uint32_t n, const uint32_t *, int, uint8_t *codes_out, int, void *) {
  if (!pq || !X) return fail(SDB_ERR_INVALID, "NULL argument");
  nap();
  if (codes_out)
    for (size_t i = 0; i < (size_t)n * pq->M; i++) codes_out[i] = (uint8_t)(i % pq->K);
  return SDB_OK;
}
int sdb_pq_set_codebook(sdb_pq *pq, const float *fc, int) {
  if (!pq || !fc) return fail(SDB_ERR_INVALID, "NULL argument");
  pq->fc.assign(fc, fc + (size_t)pq->K * pq->dim);
  return SDB_OK;
}
int sdb_pq_get_codebook(const sdb_pq *pq, float *fc, float *cd) {
  if (!pq) return fail(SDB_ERR_INVALID, "NULL argument");
  if (fc) memcpy(fc, pq->fc.data(), pq->fc.size() * 4);
  if (cd) memset(cd, 0, (size_t)pq->M * pq->K * pq->K * 4);
  return SDB_OK;
}
int sdb_index_attach_pq(sdb_index *ix, const sdb_pq *pq, void *) {
  if (!ix || !pq) return fail(SDB_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ix->mu);
  ix->M = pq->M;
  return SDB_OK;
}
int sdb_index_set_codes(sdb_index *ix, uint64_t, const uint64_t *, const uint8_t *) {
  return ix ? SDB_OK : fail(SDB_ERR_INVALID, "NULL argument");
}
int sdb_index_get_codes(const sdb_index *ix, uint64_t n, const uint64_t *, uint8_t *codes) {
  if (!ix || !codes) return fail(SDB_ERR_INVALID, "NULL argument");
  memset(codes, 0, n * ix->M);
  return SDB_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The shard exchange on the CPU: cluster.hip's collective() with the device work replaced by host loops.  Order,
// ring slots and rendezvous are turnstile.h, as in the library.
namespace {
struct MockTag {
  uint64_t seq = 0, ticket = 0, nq = 0, qhash = 0;
  uint32_t per_shard = 0, limit = 0, status = 0;
};
struct MockSlot : sdb::SlotState {
  std::vector<uint64_t> b_ids, m_ids;  // this rank's block / the merged answer
  std::vector<float> b_d, m_d;
  std::vector<uint32_t> b_c, m_c, m_s;
  MockTag tag;
  int verdict = 0;  // 1 ok, 2 mismatch, 3 shard failed
  std::string why;
};
struct MockArrival;
}  // namespace

struct sdb_cluster : sdb::OrderState {
  int world = 1, device = 0;
  sdb::GroupState<MockArrival> *group = nullptr;
  MockSlot ring[SDB_CLUSTER_RING];
};

namespace {
struct MockArrival {
  sdb_cluster *c = nullptr;
  sdb::OrderState *owner = nullptr;
  MockSlot *slot = nullptr;
};
using MockGroup = sdb::GroupState<MockArrival>;

uint64_t hash_queries(const float *q, uint64_t n) {
  uint64_t h = 0;
  for (uint64_t i = 0; i < n; i++) {
    uint32_t b;
    memcpy(&b, &q[i], 4);
    uint64_t x = (i << 32) ^ b ^ 0x9e3779b97f4a7c15ull;
    x ^= x >> 30, x *= 0xbf58476d1ce4e5b9ull, x ^= x >> 27, x *= 0x94d049bb133111ebull, x ^= x >> 31;
    h += x;
  }
  return h;
}

// what exchange_shared + k_topk_merge do: compare the tags, merge by (distance, shard, id); group lock held
void mock_exchange(std::vector<MockArrival> &arr) {
  std::sort(arr.begin(), arr.end(), [](const MockArrival &a, const MockArrival &b) { return a.c->rank < b.c->rank; });
  const MockTag &t0 = arr[0].slot->tag;
  int verdict = 1;
  std::string why;
  for (auto &a : arr) {
    const MockTag &t = a.slot->tag;
    if (t.seq != t0.seq || t.ticket != t0.ticket || t.nq != t0.nq || t.per_shard != t0.per_shard || t.limit != t0.limit ||
        t.qhash != t0.qhash) {
      verdict = 2;
      why = "the blocks gathered belong to different requests (rank " + std::to_string(a.c->rank) + " vs rank 0: ticket " +
            std::to_string(t.ticket) + " / " + std::to_string(t0.ticket) + ", seq " + std::to_string(t.seq) + " / " +
            std::to_string(t0.seq) + ")";
      violation("%s", why.c_str());
      break;
    }
  }
  if (verdict == 1)
    for (auto &a : arr)
      if (a.slot->tag.status) {
        verdict = 3;
        why = "the search on shard " + std::to_string(a.c->rank) + " failed with status " + std::to_string(a.slot->tag.status);
        break;
      }
  const uint64_t nq = t0.nq;
  const uint32_t per = t0.per_shard, limit = t0.limit;
  for (auto &a : arr) {
    MockSlot &s = *a.slot;
    s.verdict = verdict, s.why = why;
    s.m_ids.assign(nq * limit, 0), s.m_d.assign(nq * limit, 0.f), s.m_s.assign(nq * limit, 0), s.m_c.assign(nq, 0);
    if (verdict != 1) continue;
    struct E {
      float d;
      uint32_t shard;
      uint64_t id;
    };
    std::vector<E> all;
    for (uint64_t q = 0; q < nq; q++) {
      all.clear();
      for (auto &b : arr)
        for (uint32_t j = 0; j < b.slot->b_c[q]; j++)
          all.push_back({b.slot->b_d[q * per + j], (uint32_t)b.c->rank, b.slot->b_ids[q * per + j]});
      std::sort(all.begin(), all.end(), [](const E &x, const E &y) {
        return x.d != y.d ? x.d < y.d : x.shard != y.shard ? x.shard < y.shard : x.id < y.id;
      });
      const uint32_t n = (uint32_t)std::min<size_t>(limit, all.size());
      for (uint32_t j = 0; j < n; j++)
        s.m_ids[q * limit + j] = all[j].id, s.m_d[q * limit + j] = all[j].d, s.m_s[q * limit + j] = all[j].shard;
      s.m_c[q] = n;
    }
  }
  for (auto &a : arr) a.slot->pending = false;
}

int mock_collective(sdb_cluster *c, sdb_index *ix, uint64_t ticket, uint64_t nq, const float *queries, uint32_t limit,
                    uint32_t search_size, uint64_t *out_ids, float *out_dists, uint32_t *out_shards, uint32_t *out_counts,
                    bool skip) {
  std::unique_lock<std::mutex> lk(*c->mu);
  sdb::Turn turn{c, ticket};
  if (int rc = turn.enter(lk)) return rc;
  if (nq == 0) return SDB_OK;
  if (c->group->gone >= 0) return fail(SDB_ERR_STATE, "rank %d of this shard group has been destroyed", c->group->gone);
  if (c->desync) return fail(SDB_ERR_STATE, "this rank left an earlier exchange half-way");
  if (limit < 1) return fail(SDB_ERR_INVALID, "invalid limit %u", limit);
  if (ix && search_size < limit) return fail(SDB_ERR_INVALID, "searchSize (%u) must be greater than k (%u)", search_size, limit);
  uint32_t per_shard = 0;
  sdb_shard_limit(limit, (uint32_t)c->world, 75, &per_shard);
  MockSlot *slot = sdb::take_slot(c, c->ring, lk);
  c->group->reserve(c->seq);
  const uint64_t seq = c->seq++;
  slot->b_ids.assign(nq * per_shard, 0), slot->b_d.assign(nq * per_shard, 0.f), slot->b_c.assign(nq, 0);
  int local_rc = SDB_OK;
  char local_msg[1024] = {0};
  if (skip) {
    local_rc = fail(SDB_ERR_STATE, "ticket %llu was skipped on rank %d", (unsigned long long)ticket, c->rank);
  } else {
    // the shard's own walk (no sleep: the library only ENQUEUES here, the lock is held); shard r's distances are
    // offset by r / 16 so that the merge has something to interleave
    local_rc = mock_search(ix, nq, queries, per_shard, search_size, nullptr, nullptr, slot->b_ids.data(), slot->b_d.data(),
                           slot->b_c.data(), false);
    for (auto &d : slot->b_d) d += (float)c->rank / 16.0f;
    for (auto &id : slot->b_ids) id += (uint64_t)c->rank << 56;
  }
  if (local_rc) memcpy(local_msg, t_err, sizeof(local_msg));
  slot->tag = MockTag{seq, ticket, nq, queries && !local_rc ? hash_queries(queries, nq * ix->P.dim) : 0, per_shard, limit,
                      (uint32_t)local_rc};
  slot->used = true, slot->pending = true;
  MockArrival a{c, c, slot};
  std::vector<MockArrival> all;
  if (c->group->arrive(seq, a, &all)) {
    // a failed shard's hash is 0: compare hashes only among the ranks that searched
    uint64_t h = 0;
    for (auto &x : all)
      if (!x.slot->tag.status) h = x.slot->tag.qhash;
    for (auto &x : all)
      if (x.slot->tag.status) x.slot->tag.qhash = h;
    mock_exchange(all);
    c->cv->notify_all();
  }
  slot->busy = true;
  turn.pass();
  auto enqueued = [&] { return !slot->pending; };
  if (c->deadline_ms == 0) {
    c->cv->wait(lk, enqueued);
  } else if (!c->cv->wait_for(lk, std::chrono::milliseconds(c->deadline_ms), enqueued)) {
    c->group->withdraw(seq, c);
    slot->pending = false, slot->busy = false;
    c->cv->notify_all();
    if (out_counts) memset(out_counts, 0, nq * 4);
    return fail(SDB_ERR_STATE, "shard exchange %llu (ticket %llu): the other ranks did not join within %u ms; the request was "
                "withdrawn on rank %d%s", (unsigned long long)seq, (unsigned long long)ticket, c->deadline_ms, c->rank,
                c->desync ? " and the handle is out of step with its peers, recreate it" : "");
  }
  lk.unlock();
  nap();  // the device runs walk, copies and merge
  int rc = SDB_OK;
  if (slot->verdict != 1) {
    if (local_rc != SDB_OK) rc = fail(local_rc, "%s", local_msg);
    else rc = fail(SDB_ERR_STATE, "%s", slot->why.c_str());
  }
  if (skip) rc = slot->verdict >= 2 ? SDB_OK : fail(SDB_ERR_DEVICE, "the skipped exchange left no failure verdict");
  if (rc == SDB_OK && out_ids && !skip) {
    memcpy(out_ids, slot->m_ids.data(), nq * limit * 8);
    memcpy(out_dists, slot->m_d.data(), nq * limit * 4);
    if (out_shards) memcpy(out_shards, slot->m_s.data(), nq * limit * 4);
    memcpy(out_counts, slot->m_c.data(), nq * 4);
  } else if (out_counts) {
    memset(out_counts, 0, nq * 4);
  }
  lk.lock();
  slot->busy = false;
  c->cv->notify_all();
  return rc;
}
}  // namespace

extern "C" {

int sdb_cluster_create_local(int n, const int *devices, sdb_cluster **out) {
  if (!out || n < 1 || n > 64) return fail(SDB_ERR_INVALID, "bad argument");
  auto *g = new MockGroup();
  g->world = n, g->alive = n;
  for (int i = 0; i < n; i++) {
    auto *c = new sdb_cluster();
    c->rank = i, c->world = n, c->device = devices ? devices[i] : i, c->group = g;
    c->mu = &g->mu, c->cv = &g->cv;
    out[i] = c;
  }
  return SDB_OK;
}
int sdb_cluster_destroy(sdb_cluster *c) {
  if (!c) return SDB_OK;
  MockGroup *g = c->group;
  bool last;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->gone = c->rank;
    for (auto &kv : g->rv)
      for (auto &a : kv.second) a.slot->pending = false, a.slot->verdict = 3, a.slot->why = "a rank of the group was destroyed";
    g->rv.clear();
    g->cv.notify_all();
    last = --g->alive == 0;
  }
  delete c;
  if (last) delete g;
  return SDB_OK;
}
int sdb_cluster_info(const sdb_cluster *c, int *rank, int *world, int *device) {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (device) *device = c->device;
  return SDB_OK;
}
int sdb_cluster_next_ticket(const sdb_cluster *c, uint64_t *ticket) {
  if (!c || !ticket) return fail(SDB_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(*c->mu);
  *ticket = c->next_ticket;
  return SDB_OK;
}
int sdb_cluster_set_deadline(sdb_cluster *c, uint32_t ms) {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  std::lock_guard<std::mutex> g(*c->mu);
  c->deadline_ms = ms;
  c->cv->notify_all();
  return SDB_OK;
}
int sdb_cluster_skip_ticket(sdb_cluster *c, uint64_t ticket, uint64_t nq, uint32_t, uint32_t limit) {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  if (!ticket) return fail(SDB_ERR_INVALID, "ticket 0 is not a ticket");
  if (nq == 0) {
    std::lock_guard<std::mutex> g(*c->mu);
    return sdb::skip_unentered(c, ticket);
  }
  return mock_collective(c, nullptr, ticket, nq, nullptr, limit, 0, nullptr, nullptr, nullptr, nullptr, true);
}
int sdb_cluster_search_batch(sdb_cluster *c, sdb_index *ix, uint64_t ticket, uint64_t nq, const float *queries, uint32_t limit,
                             uint32_t search_size, uint64_t *out_ids, float *out_dists, uint32_t *out_shards,
                             uint32_t *out_counts, int, void *) {
  if (!c || !ix) return fail(SDB_ERR_INVALID, "NULL handle");
  return mock_collective(c, ix, ticket, nq, queries, limit, search_size, out_ids, out_dists, out_shards, out_counts, false);
}

}  // extern "C"
