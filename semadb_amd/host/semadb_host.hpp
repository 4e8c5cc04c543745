// semadb_host.hpp -- C++ host-side mirror of the reference's Go interfaces for the hot path, on top of
// the C ABI (include/semadb_amd.h).  The reference is Go; this image has no Go toolchain, so this header
// is the compiled stand-in for the cgo shim of INTEGRATION.md: same package / type / method names,
// same argument meaning, same error behaviour, so that host code and tests read like the reference's.
//
//   semadb::conversion   <-> conversion/            (NodeKey, float32 / edge-list byte codecs)
//   semadb::diskstore    <-> diskstore/             (Bucket interface, MemBucket = NewMemBucket)
//   semadb::distance     <-> distance/              (GetFloatDistanceFn)
//   semadb::models       <-> models/                (IndexVectorVamanaParameters, SearchVectorVamanaOptions, SearchResult)
//   semadb::vamana       <-> shard/index/vamana/    (STARTID, IndexVectorChange, NewIndexVamana, IndexVamana)
//   semadb::flat         <-> shard/index/flat/      (IndexFlat)
//   semadb::cluster      <-> cluster/actions.go     (the in-node fan-out of SearchPoints: GpuFanout, PerShardLimit)
//
// IndexVamana keeps its state in HBM and treats the bucket as the source of truth exactly like the
// reference's ItemCache: NewIndexVamana fills HBM from the 'n<id>v' / 'n<id>e' keys (plain.go:125-141,
// node.go:96-111), InsertUpdateDelete flushes the touched rows back (vamana.go:265-276).  Search may be
// called from many threads at once; a micro-batcher coalesces the calls into sdb_index_search_batch
// (the reference has no batch entry point: one goroutine per request, shard/cache/manager.go:163).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <optional>
#include <set>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/semadb_amd.h"

namespace semadb {

// Go `error`: empty message == nil
struct Error {
  std::string msg;
  Error() = default;
  explicit Error(std::string m) : msg(std::move(m)) {}
  explicit operator bool() const { return !msg.empty(); }
  static Error wrap(const std::string &what, int rc) {
    return Error(what + ": " + std::string(sdb_last_error()) + " (status " + std::to_string(rc) + ")");
  }
};

// ---------------------------------------------------------------------------------------------------
namespace conversion {  // conversion/keys.go, conversion/conversion.go
inline std::string NodeKey(uint64_t id, char suffix) {  // keys.go:6-12
  std::string k(10, '\0');
  k[0] = 'n';
  for (int i = 0; i < 8; i++) k[1 + i] = (char)((id >> (8 * i)) & 0xFF);
  k[9] = suffix;
  return k;
}
inline bool NodeIdFromKey(const std::string &key, char suffix, uint64_t *id) {  // keys.go:15-20
  if (key.size() != 10 || key[0] != 'n' || key[9] != suffix) return false;
  uint64_t v = 0;
  for (int i = 0; i < 8; i++) v |= (uint64_t)(uint8_t)key[1 + i] << (8 * i);
  *id = v;
  return true;
}
inline std::string Uint64ToBytes(uint64_t v) {  // conversion.go:59-63
  std::string b(8, '\0');
  for (int i = 0; i < 8; i++) b[i] = (char)((v >> (8 * i)) & 0xFF);
  return b;
}
inline uint64_t BytesToUint64(const std::string &b) {
  uint64_t v = 0;
  for (int i = 0; i < 8 && i < (int)b.size(); i++) v |= (uint64_t)(uint8_t)b[i] << (8 * i);
  return v;
}
inline std::string Float32ToBytes(const float *f, size_t n) {  // raw little-endian, conversion.go:97-99
  return std::string(reinterpret_cast<const char *>(f), n * 4);
}
inline std::vector<float> BytesToFloat32(const std::string &b) {  // conversion.go:101-108
  std::vector<float> f(b.size() / 4);
  if (!f.empty()) std::memcpy(f.data(), b.data(), f.size() * 4);
  return f;
}
inline std::string EdgeListToBytes(const std::vector<uint64_t> &e) {  // conversion.go:110-116
  std::string b;
  b.reserve(e.size() * 8);
  for (uint64_t v : e) b += Uint64ToBytes(v);
  return b;
}
inline std::vector<uint64_t> BytesToEdgeList(const std::string &b) {  // conversion.go:118-124
  std::vector<uint64_t> e(b.size() / 8);
  for (size_t i = 0; i < e.size(); i++) e[i] = BytesToUint64(b.substr(i * 8, 8));
  return e;
}
}  // namespace conversion

// ---------------------------------------------------------------------------------------------------
namespace diskstore {  // diskstore/diskstore.go:45-71
class Bucket {
 public:
  virtual ~Bucket() = default;
  virtual bool Get(const std::string &key, std::string *value) const = 0;
  virtual Error Put(const std::string &key, const std::string &value) = 0;
  virtual Error Delete(const std::string &key) = 0;
  virtual Error ForEach(const std::function<Error(const std::string &, const std::string &)> &fn) const = 0;
};
// diskstore.NewMemBucket (diskstore/memstore.go:16): the reference's universal fake backend
class MemBucket : public Bucket {
  std::map<std::string, std::string> kv_;
  mutable std::mutex mu_;

 public:
  bool Get(const std::string &key, std::string *value) const override {
    std::lock_guard<std::mutex> g(mu_);
    auto it = kv_.find(key);
    if (it == kv_.end()) return false;
    if (value) *value = it->second;
    return true;
  }
  Error Put(const std::string &key, const std::string &value) override {
    std::lock_guard<std::mutex> g(mu_);
    kv_[key] = value;
    return Error();
  }
  Error Delete(const std::string &key) override {
    std::lock_guard<std::mutex> g(mu_);
    kv_.erase(key);
    return Error();
  }
  Error ForEach(const std::function<Error(const std::string &, const std::string &)> &fn) const override {
    std::map<std::string, std::string> snap;
    {
      std::lock_guard<std::mutex> g(mu_);
      snap = kv_;
    }
    for (auto &p : snap)
      if (Error e = fn(p.first, p.second)) return e;
    return Error();
  }
  size_t size() const {
    std::lock_guard<std::mutex> g(mu_);
    return kv_.size();
  }
};
}  // namespace diskstore

// ---------------------------------------------------------------------------------------------------
namespace models {  // models/index.go:275-282, models/search.go:238-275
constexpr const char *DistanceEuclidean = "euclidean";
constexpr const char *DistanceCosine = "cosine";
constexpr const char *DistanceDot = "dot";
// models/quantizer.go:5-76 (binary quantizer: out of scope, hamming/jaccard are not on the path)
constexpr const char *QuantizerNone = "none";
constexpr const char *QuantizerProduct = "product";
struct ProductQuantizerParameters {
  int NumCentroids = 256;       // 2..256 (uint8 centroid ids)
  int NumSubVectors = 8;        // >= 2, divides the vector size
  int TriggerThreshold = 10000;  // 1000..10000: Fit runs once the store holds this many points
  Error Validate() const {       // :65-76
    if (NumCentroids < 2 || NumCentroids > 256)
      return Error("numCentroids must be between 2 and 256, got " + std::to_string(NumCentroids));
    if (NumSubVectors < 2) return Error("numSubVectors must be at least 2, got " + std::to_string(NumSubVectors));
    if (TriggerThreshold < 1000 || TriggerThreshold > 10000)
      return Error("triggerThreshold must be between 1000 and 10000, got " + std::to_string(TriggerThreshold));
    return Error();
  }
};
struct Quantizer {
  std::string Type = QuantizerNone;
  std::optional<ProductQuantizerParameters> Product;
  Error Validate() const {  // :11-28
    if (Type == QuantizerNone) return Error();
    if (Type == QuantizerProduct) {
      if (!Product) return Error("product quantizer parameters not provided");
      return Product->Validate();
    }
    return Error("unknown quantizer type " + Type);
  }
};
struct IndexVectorVamanaParameters {
  uint32_t VectorSize = 0;
  std::string DistanceMetric;
  int SearchSize = 75;
  int DegreeBound = 64;
  float Alpha = 1.2f;
  std::optional<models::Quantizer> Quantizer;  // models/index.go:281
};
struct IndexVectorFlatParameters {  // models/index.go:238-246
  uint32_t VectorSize = 0;
  std::string DistanceMetric;
  std::optional<models::Quantizer> Quantizer;
};
struct SearchVectorFlatOptions {  // models/search.go:308-314
  std::vector<float> Vector;
  int Limit = 10;
  std::optional<float> Weight;
};
struct SearchVectorVamanaOptions {
  std::vector<float> Vector;
  int SearchSize = 75;
  int Limit = 10;
  std::optional<float> Weight;
};
struct SearchResult {
  uint64_t NodeId = 0;
  float Distance = 0;
  float HybridScore = 0;
};
}  // namespace models

inline int metric_code(const std::string &name) {
  if (name == models::DistanceEuclidean) return SDB_METRIC_EUCLIDEAN;
  if (name == models::DistanceCosine) return SDB_METRIC_COSINE;
  if (name == models::DistanceDot) return SDB_METRIC_DOT;
  return -1;
}

// ---------------------------------------------------------------------------------------------------
namespace distance {  // distance/distance.go:11,70-83
using FloatDistFunc = std::function<float(const std::vector<float> &, const std::vector<float> &)>;
inline Error GetFloatDistanceFn(const std::string &name, FloatDistFunc *out, int device = 0) {
  const int mc = metric_code(name);
  if (mc < 0) return Error("unknown float32 distance function: " + name);  // distance.go:81
  *out = [mc, device](const std::vector<float> &x, const std::vector<float> &y) {
    float d = 0;
    // like asm.Dot the length comes from x only (dot.s:10)
    sdb_distance_batch(mc, (uint32_t)x.size(), x.data(), 1, y.data(), 1, &d, SDB_MEM_HOST, device, nullptr);
    return d;
  };
  return Error();
}
}  // namespace distance

// ---------------------------------------------------------------------------------------------------
// Micro-batcher: the reference has no batch entry point -- every REST request is its own goroutine calling
// IndexVamana.Search under the shard's RLock (shard/cache/manager.go:163, shard/index/search.go:53-87).  One
// query alone leaves the GPU idle (a walk is ~80 dependent hops), so concurrent Search calls are coalesced
// into sdb_index_search_batch calls; `workers` threads run the device calls, so up to `workers` batches are in
// flight (a second batch fills the SIMDs the first one's finished walks have left).  Four by
// default since round 6: a worker spends part of its cycle on the host (waiting for the last copies into its slab,
// scattering 1 024 answers, waking clients), and with two workers the device then runs ONE batch for that long;
// with the slabs page-locked a device call is a single kernel launch that reads the slab and writes the result
// slabs in place, so more calls in flight cost nothing on the copy engines (measured on one MI355X, 1M x 384:
// 2 / 3 / 4 / 6 workers = 1.06 / 1.15 / 1.18 / 1.18 M queries/s; with staged copies 1.05 / 0.97 / 1.00 / 1.05).
//
// What a request costs on the host, in the order it happens (round 3; the first version took one mutex and one
// condition variable per request, re-gathered the query vectors into a fresh pageable buffer per batch, and moved
// between 0.65 and 1.19 M queries/s from box to box):
//   submit    ONE atomic add reserves slot i of the batch that is filling -- no lock; the query vector is then
//             copied by the submitting thread, all of them at once, straight into that batch's PINNED slab
//             (sdb_host_alloc), from which the device call's H2D copy is one DMA.
//   seal      the submit that takes the last slot hands the batch to the workers (the only lock of the fast path, once
//             per batch); a batch that does not fill within `window` is sealed by a worker's timed wait.  Unfiltered
//             requests with the prevailing (limit, searchSize) take this path; filtered ones (they carry id sets) and
//             other parameters go through a queue that a worker groups.
//   run       the worker waits until every reserved slot has been written, calls the device, scatters the answers
//             from its pinned result slabs and adds to each submitting client's counter of finished requests.
//   wake      a client polls its counter for a moment before it sleeps; only a client that is actually asleep costs
//             the worker a futex call -- a Go host's equivalent is a channel send per request.
class SearchBatcher {
 public:
  using Filter = std::set<uint64_t>;  // roaring64.Bitmap: ascending iteration, Contains
  struct Client {
    std::atomic<uint64_t> completed{0};  // requests of this client finished so far (written inside mu)
    bool sleeping = false;               // inside mu
    std::mutex mu;
    std::condition_variable cv;
  };
  struct Request {
    const float *vector = nullptr;  // `dim` floats, caller-owned until done
    uint32_t limit = 10, search_size = 75;
    const Filter *filter = nullptr;
    uint64_t *ids = nullptr;  // [limit] caller-owned outputs
    float *dists = nullptr;
    uint32_t count = 0;
    Error err;
    std::atomic<bool> done{false};
    std::atomic<bool> cancelled{false};  // context cancellation: still answered, result ignored by the caller
    Client *client = nullptr;
    int64_t t_submit_ns = 0, t_done_ns = 0;  // steady clock: the request's time inside the batcher
  };

  SearchBatcher(sdb_index *h, uint32_t dim, size_t max_batch = 1024,
                std::chrono::microseconds window = std::chrono::microseconds(200), unsigned workers = 4)
      : h_(h), dim_(dim), max_batch_(max_batch ? max_batch : 1), window_(window) {
    workers = workers ? workers : 1;
    for (unsigned i = 0; i < workers + 2; i++) free_.push_back(newBatch());
    cur_.store(takeFree());
    for (unsigned i = 0; i < workers; i++) threads_.emplace_back([this] { loop(); });
  }
  ~SearchBatcher() {
    {
      std::lock_guard<std::mutex> g(qmu_);
      stop_ = true;
    }
    qcv_.notify_all();
    for (auto &t : threads_)
      if (t.joinable()) t.join();
    for (Batch *b : all_) {
      if (b->pinned) sdb_host_free(b->queries);
      else std::free(b->queries);
      delete b;
    }
  }
  SearchBatcher(const SearchBatcher &) = delete;
  SearchBatcher &operator=(const SearchBatcher &) = delete;

  // call before the first submit (the slabs were sized at construction; a smaller batch just uses less of them)
  void configure(size_t max_batch, std::chrono::microseconds window) {
    std::lock_guard<std::mutex> g(qmu_);
    if (max_batch && max_batch <= cap_) max_batch_ = max_batch;
    window_ = window;
  }
  uint64_t deviceBatches() const { return n_batches_.load(); }
  uint64_t queriesServed() const { return n_queries_.load(); }
  uint64_t backpressureWaits() const { return n_backpressure_.load(); }  // submits that found every slab in use

  // enqueue; the request (and what it points to) must stay alive until r->done
  void submit(Request *r) {
    r->t_submit_ns = nowNs();
    const uint64_t key = ((uint64_t)r->limit << 32) | r->search_size;
    uint32_t tag = 0;
    const bool fast = !r->filter && tagOf(key, &tag);
    while (fast) {
      Batch *b = cur_.load(std::memory_order_acquire);
      if (!b) {  // every slab is in use (back-pressure) or a rotation is under way
        n_backpressure_.fetch_add(1, std::memory_order_relaxed);
        std::unique_lock<std::mutex> lk(qmu_);
        fcv_.wait(lk, [&] { return cur_.load(std::memory_order_acquire) != nullptr || stop_; });
        if (stop_) {
          lk.unlock();
          r->err = Error("batcher stopped");
          tell(finish(r), 1);
          return;
        }
        continue;
      }
      // The batch's parameters and its slot counter live in ONE atomic (tag << 32 | slots reserved) and a slot is
      // reserved by compare-and-swap on both: a slab that was sealed, run, recycled for other parameters and installed
      // again between the look at its parameters and the reservation fails the swap instead of handing this request
      // a slot of a batch with another (limit, searchSize), whose answers would not fit the caller's buffers.
      uint64_t st = b->st.load(std::memory_order_acquire);
      if ((uint32_t)(st >> 32) != tag) break;  // other parameters than the filling batch's: the queue
      const uint32_t i = (uint32_t)st;
      if (i >= max_batch_) {  // full or sealed: whoever filled / sealed it is installing the next one
        while (cur_.load(std::memory_order_acquire) == b && (uint32_t)b->st.load(std::memory_order_acquire) >= max_batch_)
          std::this_thread::yield();
        continue;
      }
      if (!b->st.compare_exchange_weak(st, st + 1, std::memory_order_acq_rel)) continue;
      if (i == 0) b->t_first_ns.store(r->t_submit_ns, std::memory_order_relaxed);
      b->reqs[i] = r;
      std::memcpy(b->queries + (size_t)i * dim_, r->vector, (size_t)dim_ * 4);  // into the pinned slab
      b->written.fetch_add(1, std::memory_order_release);
      if (i + 1 == max_batch_) {  // the submit that takes the last slot hands the batch on
        std::lock_guard<std::mutex> g(qmu_);
        rotateLocked(b, (uint32_t)max_batch_);
      } else if (i == 0) {
        // a worker starts this batch's window.  Through the lock: a worker that has just read "no first request
        // yet" under it is inside its wait by the time this notify is sent (a notify between its look and its wait
        // would be lost, and the batch would sit for the worker's 2 ms poll instead of its window)
        { std::lock_guard<std::mutex> g(qmu_); }
        qcv_.notify_one();
      }
      return;
    }
    // filtered requests (they carry id sets) and requests whose (limit, searchSize) differ from the filling batch's:
    // a worker groups them.  When the unfiltered traffic has moved to other parameters the fast path follows it.
    std::lock_guard<std::mutex> g(qmu_);
    if (stop_) {  // nobody will run the queue any more
      r->err = Error("batcher stopped");
      tell(finish(r), 1);
      return;
    }
    queued_.push_back(r);
    if (fast && (fast_key_ == 0 || ++other_streak_ > 4 * max_batch_)) {
      fast_key_ = key, other_streak_ = 0;
      Batch *b = cur_.load(std::memory_order_acquire);
      if (b && b->key != key) {
        const uint32_t got = seal(b);
        if (got < max_batch_) rotateLocked(b, got);  // else: the submit that filled it is rotating
      }
    }
    qcv_.notify_one();
  }
  // block the client until at least `target` of its requests have completed.  A Client must outlive its wake-ups:
  // before it is destroyed, wait for the count of everything it submitted (Request::done alone is raised earlier)
  static void waitFor(Client *c, uint64_t target) {
    for (int spin = 0; spin < 200; spin++) {  // ~a batch period of polling before paying for a futex sleep
      if (c->completed.load(std::memory_order_acquire) >= target) {
        // the worker adds to the count inside c->mu: passing through it once means the worker has let go of `c`
        std::lock_guard<std::mutex> g(c->mu);
        return;
      }
      std::this_thread::yield();
    }
    std::unique_lock<std::mutex> lk(c->mu);
    c->sleeping = true;
    c->cv.wait(lk, [&] { return c->completed.load(std::memory_order_acquire) >= target; });
    c->sleeping = false;
  }
  // the synchronous form IndexVamana.Search uses
  void run(Request *r) {
    Client c;
    r->client = &c;
    submit(r);
    waitFor(&c, 1);
    r->client = nullptr;  // `c` ends here; the request may live on in the caller
  }

 private:
  struct Batch {
    float *queries = nullptr;  // pinned [cap][dim]
    std::vector<Request *> reqs;
    std::atomic<uint64_t> st{0};         // tagOf(key) << 32 | slots reserved; slots >= max_batch: full or sealed
    bool pinned = false;                 // queries came from sdb_host_alloc (else malloc)
    std::atomic<uint32_t> written{0};    // slots whose vector has been copied in
    uint32_t count = 0;                  // slots that belong to the batch once it is sealed
    uint64_t key = 0;                    // (limit << 32 | searchSize) of every request in it; set when installed
    std::atomic<int64_t> t_first_ns{0};
  };
  // `b` is full (count = max_batch) or was sealed (count = what had been reserved): queue it for the workers and
  // install a fresh slab as the filling batch -- or none, if all are in use: submitters then wait.  qmu_ held.
  void rotateLocked(Batch *b, uint32_t count) {
    b->count = count;
    if (count == max_batch_) other_streak_ = 0;  // the prevailing parameters are alive
    if (count) sealed_.push_back(b);
    else free_.push_back(b);
    cur_.store(takeFree(), std::memory_order_release);
    qcv_.notify_one();
    fcv_.notify_all();
  }
  // the 32-bit form of (limit << 32 | searchSize) that shares the slot atomic; parameters beyond 16 bits and limits
  // beyond the workers' result slabs (the API's maxima are 75 and 75, models/search.go:287-297) take the queue,
  // whose calls size their own buffers -- so that a request's answer never depends on which path it took
  static constexpr size_t kMaxLimit = 128;
  static bool tagOf(uint64_t key, uint32_t *tag) {
    const uint64_t limit = key >> 32, L = key & 0xFFFFFFFFull;
    if (limit > kMaxLimit || L > 0xFFFF) return false;
    *tag = (uint32_t)(limit << 16 | L);
    return true;
  }
  // close a batch to further reservations: what its counter held before is what belongs to it (>= max_batch: it was
  // full or sealed already and whoever did that is rotating it).  At most one seal per life adds to the counter.
  uint32_t seal(Batch *b) {
    uint64_t st = b->st.load(std::memory_order_acquire);
    for (;;) {
      if ((uint32_t)st >= max_batch_) return (uint32_t)st;
      if (b->st.compare_exchange_weak(st, st + max_batch_, std::memory_order_acq_rel)) return (uint32_t)st;
    }
  }
  static int64_t nowNs() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  Batch *newBatch() {
    Batch *b = new Batch();
    cap_ = max_batch_;
    void *p = nullptr;
    b->pinned = sdb_host_alloc(cap_ * (size_t)dim_ * 4, &p) == SDB_OK && p;
    if (!b->pinned) p = std::malloc(cap_ * (size_t)dim_ * 4 + 4);  // pageable: the device call stages it, slower, same answers
    if (!p) throw std::bad_alloc();
    b->queries = static_cast<float *>(p);
    b->reqs.resize(cap_);
    all_.push_back(b);
    return b;
  }
  Batch *takeFree() {  // qmu_ or construction
    if (free_.empty()) return nullptr;
    Batch *b = free_.back();
    free_.pop_back();
    b->written.store(0, std::memory_order_relaxed);
    b->t_first_ns.store(0, std::memory_order_relaxed);
    b->count = 0;
    b->key = fast_key_;
    uint32_t tag = 0xFFFFFFFFu;  // no request's tag: parameters that do not fit one fill no batch
    (void)tagOf(fast_key_, &tag);
    b->st.store((uint64_t)tag << 32, std::memory_order_release);
    return b;
  }
  // a request is answered: `r` may be gone as soon as its client has been told (tell()), so the client is returned
  Client *finish(Request *r) {
    Client *c = r->client;
    r->t_done_ns = nowNs();
    r->done.store(true, std::memory_order_release);
    return c;
  }
  // `k` more requests of client `c` are done: one short critical section per client and batch; the futex call only
  // for a client that is asleep
  static void tell(Client *c, uint64_t k) {
    if (!c) return;
    std::lock_guard<std::mutex> g(c->mu);
    c->completed.fetch_add(k, std::memory_order_release);
    if (c->sleeping) c->cv.notify_all();
  }
  struct Told {  // (client, finished requests) pairs of one device batch
    std::vector<std::pair<Client *, uint64_t>> v;
    void add(Client *c) {
      for (auto &w : v)
        if (w.first == c) {
          w.second++;
          return;
        }
      v.emplace_back(c, 1);
    }
    ~Told() {
      for (auto &w : v) tell(w.first, w.second);
    }
  };

  void loop() {
    // this worker's pinned result slabs, for the largest limit the API allows (models/search.go:287-297)
    // pinned when that can be had (the limit on locked memory), else ordinary pages: the device call then stages the
    // copies back, slower, same answers -- like the query slabs (newBatch)
    struct Slab {
      void *p = nullptr;
      bool pinned = false;
      void get(size_t bytes) {
        pinned = sdb_host_alloc(bytes, &p) == SDB_OK && p;
        if (!pinned) p = std::calloc(1, bytes + 8);
      }
      void drop() {
        if (pinned) sdb_host_free(p);
        else std::free(p);
        p = nullptr;
      }
    } s_ids, s_d, s_c;
    s_ids.get(cap_ * kMaxLimit * 8), s_d.get(cap_ * kMaxLimit * 4), s_c.get(cap_ * 4);
    void *p_ids = s_ids.p, *p_d = s_d.p, *p_c = s_c.p;
    // (not even pageable memory: runBatch answers every request of this worker with an error)
    for (;;) {
      Batch *b = nullptr;
      std::vector<Request *> others;
      {
        std::unique_lock<std::mutex> lk(qmu_);
        for (;;) {
          if (!sealed_.empty()) {
            b = sealed_.front();
            sealed_.pop_front();
            break;
          }
          if (!queued_.empty()) {
            others.swap(queued_);
            break;
          }
          // a partial batch whose first request has waited `window`: seal it.  Adding max_batch to its slot counter
          // closes it (later reservations see a full batch); what the counter held before is what belongs to it.
          Batch *c = cur_.load(std::memory_order_acquire);
          const int64_t first = c ? c->t_first_ns.load(std::memory_order_relaxed) : 0;
          const int64_t age = first ? nowNs() - first : 0;
          // ... or at once when no device batch is running: waiting then buys nothing, the window would only be added to
          // the request's latency (a lone Search call: 0.37 ms instead of 0.37 + window).  Under load a batch is running,
          // arrivals pile up behind it for a window or until it ends, and the batches stay large.
          if (c && first && (age >= window_.count() * 1000 || stop_ || busy_.load(std::memory_order_acquire) == 0)) {
            const uint32_t got = seal(c);
            if (got < max_batch_) rotateLocked(c, got);  // ours to seal (otherwise the submit that filled it is rotating)
            else lk.unlock(), std::this_thread::yield(), lk.lock();
            continue;
          }
          if (stop_) {
            lk.unlock();
            s_ids.drop(), s_d.drop(), s_c.drop();
            return;
          }
          if (c && first) qcv_.wait_for(lk, std::chrono::nanoseconds(window_.count() * 1000 - age));
          else qcv_.wait_for(lk, std::chrono::milliseconds(2));
        }
      }
      if (b) {
        busy_.fetch_add(1, std::memory_order_acq_rel);
        runBatch(b, (uint64_t *)p_ids, (float *)p_d, (uint32_t *)p_c);
        busy_.fetch_sub(1, std::memory_order_acq_rel);
        qcv_.notify_one();  // what piled up behind this batch can go now
        std::lock_guard<std::mutex> g(qmu_);
        free_.push_back(b);
        if (!cur_.load(std::memory_order_acquire)) cur_.store(takeFree(), std::memory_order_release);
        fcv_.notify_all();
      } else if (!others.empty()) {
        busy_.fetch_add(1, std::memory_order_acq_rel);
        runQueued(others);
        busy_.fetch_sub(1, std::memory_order_acq_rel);
        qcv_.notify_one();
      }
    }
  }

  void runBatch(Batch *b, uint64_t *ids, float *dists, uint32_t *counts) {
    const uint32_t nq = b->count, limit = (uint32_t)(b->key >> 32), L = (uint32_t)b->key;
    while (b->written.load(std::memory_order_acquire) < nq) std::this_thread::yield();  // the last copies in flight
    int rc;
    const char *why = nullptr;
    if (limit > kMaxLimit) rc = SDB_ERR_INVALID, why = "limit beyond the batcher's result slabs (128)";
    else if (!ids || !dists || !counts) rc = SDB_ERR_DEVICE, why = "out of host memory for the batcher's result slabs";
    else rc = sdb_index_search_batch(h_, nq, b->queries, limit, L, nullptr, nullptr, ids, dists, counts, nullptr, SDB_MEM_HOST, nullptr);
    n_batches_++;
    n_queries_ += nq;
    const Error err = rc ? Error(std::string(why ? why : sdb_last_error())) : Error();
    Told told;
    for (uint32_t i = 0; i < nq; i++) {
      Request *r = b->reqs[i];
      if (!rc) {
        r->count = counts[i];
        std::memcpy(r->ids, ids + (size_t)i * limit, (size_t)counts[i] * 8);
        std::memcpy(r->dists, dists + (size_t)i * limit, (size_t)counts[i] * 4);
      } else {
        r->err = err;
      }
      told.add(finish(r));
    }
  }

  // queued requests: one device call per (limit, searchSize, filtered) group, filters flattened in ascending order
  // like roaring
  void runQueued(std::vector<Request *> &all) {
    while (!all.empty()) {
      std::vector<Request *> reqs, rest;
      for (Request *r : all)
        (reqs.size() < max_batch_ && r->limit == all[0]->limit && r->search_size == all[0]->search_size &&
                 (r->filter != nullptr) == (all[0]->filter != nullptr)
             ? reqs
             : rest)
            .push_back(r);
      all.swap(rest);
      const bool filtered = reqs[0]->filter != nullptr;
      const size_t nq = reqs.size(), d = dim_;
      const uint32_t limit = reqs[0]->limit, L = reqs[0]->search_size;
      std::vector<float> queries(nq * d);
      std::vector<uint64_t> ids(nq * limit), f_off{0}, f_ids;
      std::vector<float> dists(nq * limit);
      std::vector<uint32_t> counts(nq);
      for (size_t i = 0; i < nq; i++) {
        std::memcpy(queries.data() + i * d, reqs[i]->vector, d * 4);
        if (filtered) f_ids.insert(f_ids.end(), reqs[i]->filter->begin(), reqs[i]->filter->end());
        f_off.push_back(f_ids.size());
      }
      if (f_ids.empty()) f_ids.push_back(0);
      // dense filters go up as bitmaps over [min, max] when that is fewer bytes than the id lists (the Go twin's
      // bitmapsAreSmaller / packFilterBitmaps): the upload is what a large filter costs
      uint64_t words = 0, card = f_ids.size();
      if (filtered)
        for (size_t i = 0; i < nq; i++)
          if (!reqs[i]->filter->empty()) words += (*reqs[i]->filter->rbegin() - (*reqs[i]->filter->begin() & ~63ull)) / 64 + 1;
      int rc;
      if (filtered && card > 4096 && words < card) {
        std::vector<uint64_t> first(nq, 0), w_off{0}, w;
        for (size_t i = 0; i < nq; i++) {
          const Filter &f = *reqs[i]->filter;
          if (!f.empty()) {
            const uint64_t f0 = *f.begin() & ~63ull;
            const size_t base = w.size();
            w.resize(base + (*f.rbegin() - f0) / 64 + 1, 0);
            for (uint64_t id : f) w[base + (id - f0) / 64] |= 1ull << ((id - f0) % 64);
            first[i] = f0;
          }
          w_off.push_back(w.size());
        }
        if (w.empty()) w.push_back(0);
        rc = sdb_index_search_batch_bitmap(h_, nq, queries.data(), limit, L, first.data(), w_off.data(), w.data(), ids.data(),
                                           dists.data(), counts.data(), nullptr, SDB_MEM_HOST, nullptr);
      } else {
        rc = sdb_index_search_batch(h_, nq, queries.data(), limit, L, filtered ? f_off.data() : nullptr,
                                    filtered ? f_ids.data() : nullptr, ids.data(), dists.data(), counts.data(),
                                    nullptr, SDB_MEM_HOST, nullptr);
      }
      n_batches_++;
      n_queries_ += nq;
      const Error err = rc ? Error(std::string(sdb_last_error())) : Error();
      Told told;
      for (size_t i = 0; i < nq; i++) {
        Request *r = reqs[i];
        if (!rc) {
          r->count = counts[i];
          std::memcpy(r->ids, ids.data() + i * limit, (size_t)counts[i] * 8);
          std::memcpy(r->dists, dists.data() + i * limit, (size_t)counts[i] * 4);
        } else {
          r->err = err;
        }
        told.add(finish(r));
      }
    }
  }

  sdb_index *h_;
  uint32_t dim_;
  std::vector<std::thread> threads_;
  // the queue lock: sealed batches, queued requests, the free list, the workers' sleep -- taken once per BATCH on the
  // fast path (by the submit that fills it), never per request
  std::mutex qmu_;
  std::condition_variable qcv_, fcv_;
  std::atomic<Batch *> cur_{nullptr};  // the batch that is filling
  uint64_t fast_key_ = 0;              // (limit << 32 | searchSize) the next installed batch is for (qmu_)
  uint64_t other_streak_ = 0;          // unfiltered requests with other parameters since the last re-key (qmu_)
  std::deque<Batch *> sealed_;
  std::vector<Request *> queued_;
  std::vector<Batch *> free_, all_;
  bool stop_ = false;
  size_t max_batch_, cap_ = 0;
  std::chrono::microseconds window_;
  std::atomic<uint64_t> n_batches_{0}, n_queries_{0}, n_backpressure_{0};
  std::atomic<int> busy_{0};  // workers inside a device call
};

// ---------------------------------------------------------------------------------------------------
namespace vamana {  // shard/index/vamana/vamana.go
constexpr uint64_t STARTID = 1;                        // :28
constexpr const char *MAXNODEIDKEY = "_vamanaMaxNodeId";  // :31
// shard/vectorstore/product.go:17-18
constexpr const char *productQuantizerCentroidDistsKey = "_productQuantizerCentroidDists";
constexpr const char *productQuantizerFlatCentroidsKey = "_productQuantizerFlatCentroids";
struct IndexVectorChange {                             // :122-125 ; empty Vector == nil == delete
  uint64_t Id = 0;
  std::vector<float> Vector;
};

class IndexVamana {
 public:
  using Filter = std::set<uint64_t>;  // roaring64.Bitmap: ascending iteration, Contains
  struct SearchReturn {
    std::set<uint64_t> set;
    std::vector<models::SearchResult> results;
    Error err;
  };

  // vamana.NewIndexVamana (vamana.go:54-81)
  static std::pair<std::unique_ptr<IndexVamana>, Error> NewIndexVamana(const std::string &name,
                                                                      const models::IndexVectorVamanaParameters &params,
                                                                      diskstore::Bucket *bucket, int device = 0,
                                                                      const std::vector<float> *start_vector = nullptr) {
    const int mc = metric_code(params.DistanceMetric);
    if (mc < 0) return {nullptr, Error("could not create vector store: unknown float32 distance function: " + params.DistanceMetric)};
    std::unique_ptr<IndexVamana> v(new IndexVamana());
    v->name_ = name, v->parameters_ = params, v->bucket_ = bucket;
    sdb_index_params p{};
    p.dim = params.VectorSize, p.metric = (uint32_t)mc, p.search_size = (uint32_t)params.SearchSize;
    p.degree_bound = (uint32_t)params.DegreeBound, p.alpha = params.Alpha, p.device = device, p.strict = 1;
    if (int rc = sdb_index_create(&p, &v->h_)) return {nullptr, Error::wrap("could not create device index", rc)};
    // vectorstore.New (vectorstore.go:47-96) -> newProductQuantizer (product.go:42-88)
    if (params.Quantizer && params.Quantizer->Type != models::QuantizerNone) {
      if (params.Quantizer->Type != models::QuantizerProduct)
        return {nullptr, Error("could not create vector store: unknown vector store type " + params.Quantizer->Type)};
      if (!params.Quantizer->Product) return {nullptr, Error("could not create vector store: product quantizer parameters are nil")};
      const auto &pp = *params.Quantizer->Product;
      if (pp.NumSubVectors <= 0 || params.VectorSize % (uint32_t)pp.NumSubVectors != 0)
        return {nullptr, Error("could not create vector store: vector length " + std::to_string(params.VectorSize) +
                               " must be divisible by num subvectors " + std::to_string(pp.NumSubVectors))};
      if (pp.NumCentroids > 256)
        return {nullptr, Error("could not create vector store: number of centroids " + std::to_string(pp.NumCentroids) +
                               " cannot exceed 256")};
      if (int rc = sdb_pq_create(params.VectorSize, (uint32_t)mc, (uint32_t)pp.NumSubVectors, (uint32_t)pp.NumCentroids,
                                 device, &v->pq_))
        return {nullptr, Error::wrap("could not create vector store", rc)};
    }
    if (Error e = v->loadFromBucket(start_vector)) return {nullptr, e};
    v->batcher_.reset(new SearchBatcher(v->h_, params.VectorSize));
    return {std::move(v), Error()};
  }

  ~IndexVamana() {
    batcher_.reset();  // joins its workers; no device call is in flight afterwards
    if (h_) sdb_index_destroy(h_);
    if (pq_) sdb_pq_destroy(pq_);
  }
  bool quantized() const { return pq_fitted_; }
  // seed of the k-means first-centroid draws in Fit (kmeans.go:61-63 uses the global RNG); tests pin it
  void setFitSeed(uint64_t s) { fit_seed_ = s; }
  // SDB_TUNE_SKETCH (include/semadb_amd.h): batch searches of plain tables read a float16 copy of the rows first
  // and a neighbour's float32 row only when AddWithLimit may keep it -- same answers, + 50 % of the rows' memory.
  // (The Go twin is the package variable TwoPrecisionSearch, integration/go/vamana/vamana_mi355x.go.)
  Error setTwoPrecisionSearch(bool on) {
    const int rc = sdb_index_set_tuning(h_, SDB_TUNE_SKETCH, on ? 1 : 0);
    return rc == SDB_OK ? Error() : Error(std::string("could not switch the two-precision search: ") + sdb_last_error());
  }

  int64_t SizeInMemory() const {  // vamana.go:83-85
    int64_t b = 0;
    sdb_index_size_in_memory(h_, &b);
    return b;
  }
  void UpdateBucket(diskstore::Bucket *b) { bucket_ = b; }  // vamana.go:87-91
  uint64_t maxNodeId() const {
    uint64_t n = 0, e = 0, m = 0;
    sdb_index_stats(h_, &n, &e, &m);
    return m;
  }

  // IndexVamana.Search (vamana.go:278-310).  Thread-safe; concurrent calls share device batches.
  SearchReturn Search(const models::SearchVectorVamanaOptions &q, const Filter *filter = nullptr) {
    SearchReturn out;
    if (q.Vector.size() != parameters_.VectorSize) {  // models/search.go:198-200 (rejected upstream)
      out.err = Error("query vector length mismatch");
      return out;
    }
    if (q.SearchSize < q.Limit) {  // search.go:23-25
      out.err = Error("could not perform graph search: searchSize (" + std::to_string(q.SearchSize) +
                      ") must be greater than k (" + std::to_string(q.Limit) + ")");
      return out;
    }
    if (q.Limit < 1) {
      out.err = Error("could not perform graph search: invalid limit " + std::to_string(q.Limit));
      return out;
    }
    std::vector<uint64_t> ids((size_t)q.Limit);
    std::vector<float> dists((size_t)q.Limit);
    SearchBatcher::Request r;
    r.vector = q.Vector.data(), r.limit = (uint32_t)q.Limit, r.search_size = (uint32_t)q.SearchSize;
    r.filter = filter, r.ids = ids.data(), r.dists = dists.data();
    batcher_->run(&r);
    if (r.err) {
      out.err = Error("could not perform graph search: " + r.err.msg);
      return out;
    }
    const float weight = q.Weight ? *q.Weight : 1.0f;  // vamana.go:289-292
    for (uint32_t i = 0; i < r.count; i++) {
      models::SearchResult sr;
      sr.NodeId = ids[i], sr.Distance = dists[i];
      sr.HybridScore = (-1 * dists[i] * weight);  // :303
      out.results.push_back(sr);
      out.set.insert(ids[i]);
    }
    return out;
  }

  // vecStore.Exists (plain.go:21-24), from the bucket-side view this mirror keeps of the live ids
  bool Exists(uint64_t id) const { return live_.count(id) != 0; }

  // IndexVamana.InsertUpdateDelete (vamana.go:127-263): same classification and order as :149-251 -- new
  // ids are inserted first; the inbound edges of deleted AND updated ids are removed in one scan and the
  // deleted nodes dropped; the updated points are re-inserted one by one; then the bucket is brought up
  // to date (flush, :265-276).
  Error InsertUpdateDelete(const std::vector<IndexVectorChange> &points, uint32_t round_size = 0) {
    std::lock_guard<std::mutex> wl(write_mu_);
    std::vector<uint64_t> ins_ids, upd_ids, del_ids;
    std::vector<float> ins_vecs, upd_vecs;
    std::set<uint64_t> fresh;
    for (const auto &p : points) {
      if (p.Id == STARTID) return Error("could not distribute or insert points: cannot modify point with start id: 1");
      if (p.Id == 0) return Error("could not distribute or insert points: invalid point id: 0");
      const bool exists = Exists(p.Id) || fresh.count(p.Id);
      if (p.Vector.empty()) {  // nil vector: delete, or nothing to do (vamana.go:161-163,175-179)
        if (exists) del_ids.push_back(p.Id);
        continue;
      }
      if (p.Vector.size() != parameters_.VectorSize) return Error("vector length mismatch");  // models/index.go:182-184
      if (exists) {  // update :170-174
        upd_ids.push_back(p.Id);
        upd_vecs.insert(upd_vecs.end(), p.Vector.begin(), p.Vector.end());
      } else {  // insert; copied: never retains caller memory
        ins_ids.push_back(p.Id);
        ins_vecs.insert(ins_vecs.end(), p.Vector.begin(), p.Vector.end());
        fresh.insert(p.Id);
      }
    }
    // one write transaction, like the shard's: searches see all of it or none of it (sdb_index_begin_write)
    const bool any = !ins_ids.empty() || !del_ids.empty() || !upd_ids.empty();
    // (every point was validated above: nothing below fails on the caller's input)
    if (any)
      if (int rc = sdb_index_begin_write(h_)) return Error::wrap("could not start the write", rc);
    // an error inside the transaction: roll it back (sdb_index_abort_write) -- the index is what it was before the
    // call and takes the next write; the id set and the bucket are only touched after the commit.  (The reference's
    // cache manager scraps the shard's cache after an error inside a transaction and rebuilds it, manager.go:231-240.)
    auto failed = [&](const std::string &what, int rc) {
      Error e = Error::wrap(what, rc);
      if (sdb_index_abort_write(h_) != SDB_OK) e.msg += "; the index must be reloaded from the bucket";
      return e;
    };
    if (!ins_ids.empty())
      if (int rc = sdb_index_insert_batch(h_, ins_ids.size(), ins_ids.data(), ins_vecs.data(), SDB_MEM_HOST, round_size, nullptr))
        return failed("could not distribute or insert points", rc);
    std::vector<uint64_t> gone(del_ids);
    gone.insert(gone.end(), upd_ids.begin(), upd_ids.end());
    if (!gone.empty())
      if (int rc = sdb_index_delete_batch(h_, gone.size(), gone.data(), nullptr))
        return failed("could not remove inbound edges", rc);
    const size_t d = parameters_.VectorSize;
    for (size_t i = 0; i < upd_ids.size(); i++)  // :247-251
      if (int rc = sdb_index_insert_batch(h_, 1, &upd_ids[i], upd_vecs.data() + i * d, SDB_MEM_HOST, 1, nullptr))
        return failed("could not re-insert updated point", rc);
    if (any)
      if (int rc = sdb_index_commit(h_, nullptr)) return failed("could not commit the write", rc);
    for (uint64_t id : ins_ids) live_.insert(id);
    for (uint64_t id : del_ids) {  // DeleteFrom (plain.go:143-148, node.go:129-134)
      live_.erase(id);
      if (bucket_) {
        bucket_->Delete(conversion::NodeKey(id, 'v'));
        bucket_->Delete(conversion::NodeKey(id, 'q'));  // product.go:375-383
        bucket_->Delete(conversion::NodeKey(id, 'e'));
      }
    }
    if (Error e = fit()) return Error("could not fit vector store: " + e.msg);  // vamana.go:257-260
    if (Error e = flush()) return e;
    // the reference frees deleted nodes at flush (node.go:129-134); here their rows stay behind as tombstones
    // until they are worth squeezing out
    uint64_t rows = 0, dead = 0;
    if (sdb_index_row_usage(h_, &rows, &dead) == SDB_OK && dead * 4 > rows)
      if (int rc = sdb_index_compact(h_)) return Error::wrap("could not compact the index", rc);
    return Error();
  }

  // productQuantizer.Fit (product.go:175-236): once, when the store holds TriggerThreshold points (the start
  // node counts, it lives in the same store).  k-means runs per sub-quantizer over ALL stored vectors in
  // storage order (the reference iterates a Go map), its labels become the points' centroid ids (:216-218),
  // and from here on every distance of the index is a table distance.
  Error fit() {
    if (!pq_ || pq_fitted_) return Error();
    const auto &pp = *parameters_.Quantizer->Product;
    uint64_t n = 0, ne = 0, mx = 0;
    sdb_index_stats(h_, &n, &ne, &mx);
    if (n < (uint64_t)pp.TriggerThreshold) return Error();
    const size_t d = parameters_.VectorSize;
    std::vector<uint64_t> ids(n), off(n + 1), edges(ne ? ne : 1);
    std::vector<float> vecs(n * d);
    if (int rc = sdb_index_export(h_, ids.data(), vecs.data(), off.data(), edges.data()))
      return Error::wrap("could not collect vectors for kmeans", rc);
    std::vector<uint32_t> first((size_t)pp.NumSubVectors);
    uint64_t s = fit_seed_ ? fit_seed_ : ((uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() | 1);
    for (auto &f : first) {
      s ^= s << 13, s ^= s >> 7, s ^= s << 17;
      f = (uint32_t)(s % n);
    }
    std::vector<uint8_t> codes(n * (size_t)pp.NumSubVectors);
    // alias = 1: centroid slices alias the data rows and are overwritten by the means, as in the reference
    // (kmeans.go:63,82,144); `vecs` is this function's private copy, the slab keeps the vectors
    if (int rc = sdb_pq_fit(pq_, vecs.data(), (uint32_t)n, first.data(), 1, codes.data(), SDB_MEM_HOST, nullptr))
      return Error::wrap("kmeans", rc);
    if (int rc = sdb_index_attach_pq(h_, pq_, nullptr)) return Error::wrap("could not attach quantizer", rc);
    if (int rc = sdb_index_set_codes(h_, n, ids.data(), codes.data())) return Error::wrap("could not store centroid ids", rc);
    pq_fitted_ = true;
    return Error();
  }

  // vamana.go:265-276: vectors -> 'n<id>v', edges -> 'n<id>e', max node id
  Error flush() {
    if (!bucket_) return Error();
    uint64_t n = 0, ne = 0, mx = 0;
    sdb_index_stats(h_, &n, &ne, &mx);
    std::vector<uint64_t> ids(n), off(n + 1), edges(ne ? ne : 1);
    std::vector<float> vecs(n * parameters_.VectorSize);
    if (int rc = sdb_index_export(h_, ids.data(), vecs.data(), off.data(), edges.data()))
      return Error::wrap("could not flush", rc);
    const size_t d = parameters_.VectorSize;
    for (uint64_t i = 0; i < n; i++) {
      bucket_->Put(conversion::NodeKey(ids[i], 'v'), conversion::Float32ToBytes(vecs.data() + i * d, d));
      std::vector<uint64_t> e(edges.begin() + off[i], edges.begin() + off[i + 1]);
      bucket_->Put(conversion::NodeKey(ids[i], 'e'), conversion::EdgeListToBytes(e));
    }
    bucket_->Put(MAXNODEIDKEY, conversion::Uint64ToBytes(mx));
    if (pq_fitted_) {  // productQuantizer.Flush (product.go:307-320) + productQuantizedPoint.WriteTo (:361-373)
      const auto &pp = *parameters_.Quantizer->Product;
      const size_t M = (size_t)pp.NumSubVectors, K = (size_t)pp.NumCentroids, sub = d / M;
      std::vector<uint8_t> codes(n * M);
      if (int rc = sdb_index_get_codes(h_, n, ids.data(), codes.data())) return Error::wrap("could not flush", rc);
      for (uint64_t i = 0; i < n; i++)
        bucket_->Put(conversion::NodeKey(ids[i], 'q'), std::string((const char *)codes.data() + i * M, M));
      std::vector<float> fc(M * K * sub), cd(M * K * K);
      if (int rc = sdb_pq_get_codebook(pq_, fc.data(), cd.data())) return Error::wrap("could not flush", rc);
      bucket_->Put(productQuantizerCentroidDistsKey, conversion::Float32ToBytes(cd.data(), cd.size()));
      bucket_->Put(productQuantizerFlatCentroidsKey, conversion::Float32ToBytes(fc.data(), fc.size()));
    }
    return Error();
  }

  // tuning knobs of the micro-batcher (not part of the reference surface)
  void setBatching(size_t max_batch, std::chrono::microseconds window) { batcher_->configure(max_batch, window); }
  uint64_t deviceBatches() const { return batcher_->deviceBatches(); }
  sdb_index *handle() const { return h_; }

 private:
  IndexVamana() = default;
  // what plainPoint.ReadFrom / graphNode.ReadFrom read lazily (plain.go:125-141, node.go:96-111)
  Error loadFromBucket(const std::vector<float> *start_vector) {
    std::vector<uint64_t> ids, offsets{0}, edges;
    std::vector<float> vectors;
    const size_t d = parameters_.VectorSize;
    if (bucket_) {
      Error e = bucket_->ForEach([&](const std::string &k, const std::string &val) {
        uint64_t id;
        if (!conversion::NodeIdFromKey(k, 'v', &id)) return Error();
        if (val.size() != d * 4) return Error("vector of node " + std::to_string(id) + " has the wrong length");
        ids.push_back(id);
        auto f = conversion::BytesToFloat32(val);
        vectors.insert(vectors.end(), f.begin(), f.end());
        std::string eb;
        if (bucket_->Get(conversion::NodeKey(id, 'e'), &eb)) {
          auto el = conversion::BytesToEdgeList(eb);
          edges.insert(edges.end(), el.begin(), el.end());
        }
        offsets.push_back(edges.size());
        return Error();
      });
      if (e) return e;
    }
    if (ids.empty()) {  // setupStartNode (vamana.go:93-120)
      std::vector<float> sv;
      if (start_vector) sv = *start_vector;
      else sv = randomUnitVector(d);
      if (int rc = sdb_index_set_start(h_, sv.data(), SDB_MEM_HOST)) return Error::wrap("could not setup start node", rc);
      if (bucket_) {
        bucket_->Put(conversion::NodeKey(STARTID, 'v'), conversion::Float32ToBytes(sv.data(), d));
        bucket_->Put(conversion::NodeKey(STARTID, 'e'), "");
      }
      return Error();
    }
    if (edges.empty()) edges.push_back(0);
    for (uint64_t id : ids)
      if (id != STARTID) live_.insert(id);
    if (int rc = sdb_index_load(h_, ids.size(), ids.data(), vectors.data(), offsets.data(), edges.data(), SDB_MEM_HOST))
      return Error::wrap("could not load index into HBM", rc);
    // newProductQuantizer reads the tables back (product.go:80-86); points carry their ids under 'q'
    // (ReadFrom :334-357).  The centroid-pair table is recomputed from the centroids: same arithmetic.
    std::string fcb;
    if (pq_ && bucket_ && bucket_->Get(productQuantizerFlatCentroidsKey, &fcb)) {
      const auto &pp = *parameters_.Quantizer->Product;
      const size_t M = (size_t)pp.NumSubVectors, K = (size_t)pp.NumCentroids;
      auto fc = conversion::BytesToFloat32(fcb);
      if (fc.size() != M * K * (d / M)) return Error("stored centroids have the wrong size");
      if (int rc = sdb_pq_set_codebook(pq_, fc.data(), SDB_MEM_HOST)) return Error::wrap("could not load centroids", rc);
      if (int rc = sdb_index_attach_pq(h_, pq_, nullptr)) return Error::wrap("could not attach quantizer", rc);
      std::vector<uint64_t> qids;
      std::vector<uint8_t> qcodes;
      for (uint64_t id : ids) {
        std::string qb;
        if (bucket_->Get(conversion::NodeKey(id, 'q'), &qb) && qb.size() == M) {
          qids.push_back(id);
          qcodes.insert(qcodes.end(), qb.begin(), qb.end());
        }
      }
      if (!qids.empty())
        if (int rc = sdb_index_set_codes(h_, qids.size(), qids.data(), qcodes.data()))
          return Error::wrap("could not load centroid ids", rc);
      pq_fitted_ = true;
    }
    return Error();
  }

  static std::vector<float> randomUnitVector(size_t d) {  // vamana.go:99-110
    std::vector<float> v(d);
    uint64_t s = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() | 1;
    float sum = 0;
    for (size_t i = 0; i < d; i++) {
      s ^= s << 13, s ^= s >> 7, s ^= s << 17;
      v[i] = (float)((s >> 40) & 0xFFFFFF) / (float)0x1000000 * 2 - 1;
      sum += v[i] * v[i];
    }
    const float norm = 1 / (float)std::sqrt((double)sum);
    for (auto &x : v) x *= norm;
    return v;
  }

  std::string name_;
  models::IndexVectorVamanaParameters parameters_;
  diskstore::Bucket *bucket_ = nullptr;
  sdb_index *h_ = nullptr;
  sdb_pq *pq_ = nullptr;  // product quantizer of the vector store (vectorstore.New), if configured
  bool pq_fitted_ = false;
  uint64_t fit_seed_ = 0;
  std::set<uint64_t> live_;  // ids currently in the store (start node excluded)
  std::mutex write_mu_;
  std::unique_ptr<SearchBatcher> batcher_;  // coalesces concurrent Search calls
};

inline std::pair<std::unique_ptr<IndexVamana>, Error> NewIndexVamana(const std::string &name,
                                                                    const models::IndexVectorVamanaParameters &params,
                                                                    diskstore::Bucket *bucket, int device = 0,
                                                                    const std::vector<float> *start_vector = nullptr) {
  return IndexVamana::NewIndexVamana(name, params, bucket, device, start_vector);
}
}  // namespace vamana

// ---------------------------------------------------------------------------------------------------
namespace flat {  // shard/index/flat/flat.go
// flat.IndexFlat: a vector store and an exact scan, nothing else.  The plain store only (a quantized flat index is
// a configuration the BASELINE does not name).  InsertUpdateDelete is vecStore.Set / vecStore.Delete in the order
// given, then Flush (flat.go:41-74); Search is the scan of flat.go:76-132 on the device.
class IndexFlat {
 public:
  using Filter = std::set<uint64_t>;
  struct SearchReturn {
    std::set<uint64_t> set;
    std::vector<models::SearchResult> results;
    Error err;
  };
  static std::pair<std::unique_ptr<IndexFlat>, Error> NewIndexFlat(const models::IndexVectorFlatParameters &params,
                                                                  diskstore::Bucket *bucket, int device = 0) {
    const int mc = metric_code(params.DistanceMetric);
    if (mc < 0) return {nullptr, Error("failed to create vector store: unknown float32 distance function: " + params.DistanceMetric)};
    if (params.Quantizer && params.Quantizer->Type != models::QuantizerNone)
      return {nullptr, Error("failed to create vector store: a quantized flat index is not on the MI355X path")};
    std::unique_ptr<IndexFlat> f(new IndexFlat());
    f->parameters_ = params, f->bucket_ = bucket;
    sdb_index_params p{};
    p.dim = params.VectorSize, p.metric = (uint32_t)mc, p.search_size = 75, p.degree_bound = 64, p.alpha = 1.2f, p.device = device;
    if (int rc = sdb_index_create(&p, &f->h_)) return {nullptr, Error::wrap("failed to create vector store", rc)};
    if (bucket) {  // plainPoint.ReadFrom for every stored point (plain.go:125-141)
      std::vector<uint64_t> ids;
      std::vector<float> vecs;
      Error e = bucket->ForEach([&](const std::string &k, const std::string &val) {
        uint64_t id;
        if (!conversion::NodeIdFromKey(k, 'v', &id)) return Error();
        if (val.size() != (size_t)params.VectorSize * 4) return Error("vector of point " + std::to_string(id) + " has the wrong length");
        ids.push_back(id);
        auto v = conversion::BytesToFloat32(val);
        vecs.insert(vecs.end(), v.begin(), v.end());
        return Error();
      });
      if (e) return {nullptr, e};
      if (!ids.empty())
        if (int rc = sdb_index_set_vectors(f->h_, ids.size(), ids.data(), vecs.data(), SDB_MEM_HOST))
          return {nullptr, Error::wrap("failed to load vector store", rc)};
    }
    return {std::move(f), Error()};
  }
  ~IndexFlat() {
    if (h_) sdb_index_destroy(h_);
  }
  int64_t SizeInMemory() const {  // flat.go:34-36
    int64_t b = 0;
    sdb_index_size_in_memory(h_, &b);
    return b;
  }
  void UpdateBucket(diskstore::Bucket *b) { bucket_ = b; }  // flat.go:38-40

  // flat.go:41-74
  Error InsertUpdateDelete(const std::vector<vamana::IndexVectorChange> &points) {
    std::lock_guard<std::mutex> wl(write_mu_);
    if (points.empty()) return Error();
    const size_t d = parameters_.VectorSize;
    for (const auto &p : points)  // everything the caller can get wrong, before the transaction opens
      if (!p.Vector.empty() && p.Vector.size() != d) return Error("failed to insert/update/delete: vector length mismatch");
    if (int rc = sdb_index_begin_write(h_)) return Error::wrap("failed to insert/update/delete", rc);
    auto failed = [&](int rc) {
      Error e = Error::wrap("failed to insert/update/delete", rc);
      if (sdb_index_abort_write(h_) != SDB_OK) e.msg += "; the index must be reloaded from the bucket";
      return e;
    };
    // runs of consecutive sets / deletes, so that the two kinds keep the order the caller gave them
    size_t i = 0;
    while (i < points.size()) {
      const bool del = points[i].Vector.empty();
      std::vector<uint64_t> ids;
      std::vector<float> vecs;
      std::set<uint64_t> seen;
      while (i < points.size() && points[i].Vector.empty() == del && (del || !seen.count(points[i].Id))) {
        ids.push_back(points[i].Id), seen.insert(points[i].Id);
        vecs.insert(vecs.end(), points[i].Vector.begin(), points[i].Vector.end());
        i++;
      }
      const int rc = del ? sdb_index_remove_vectors(h_, ids.size(), ids.data())
                         : sdb_index_set_vectors(h_, ids.size(), ids.data(), vecs.data(), SDB_MEM_HOST);
      if (rc) return failed(rc);
    }
    if (int rc = sdb_index_commit(h_, nullptr)) return failed(rc);
    if (bucket_)  // vecStore.Flush, once the device has committed: plainPoint.WriteTo / DeleteFrom (plain.go:112-123,143-148)
      for (const auto &p : points) {
        if (p.Vector.empty()) bucket_->Delete(conversion::NodeKey(p.Id, 'v'));
        else bucket_->Put(conversion::NodeKey(p.Id, 'v'), conversion::Float32ToBytes(p.Vector.data(), d));
      }
    uint64_t rows = 0, dead = 0;
    if (sdb_index_row_usage(h_, &rows, &dead) == SDB_OK && dead * 4 > rows)
      if (int rc = sdb_index_compact(h_)) return Error::wrap("could not compact the store", rc);
    return Error();
  }

  // flat.go:76-132
  SearchReturn Search(const models::SearchVectorFlatOptions &q, const Filter *filter = nullptr) {
    SearchReturn out;
    if (q.Vector.size() != parameters_.VectorSize) {
      out.err = Error("query vector length mismatch");
      return out;
    }
    const uint32_t limit = (uint32_t)q.Limit;
    std::vector<uint64_t> ids(limit), f_off, f_ids;
    std::vector<float> dists(limit);
    uint32_t count = 0;
    if (filter) {
      f_off = {0, (uint64_t)filter->size()};
      f_ids.assign(filter->begin(), filter->end());
      if (f_ids.empty()) f_ids.push_back(0);
    }
    if (int rc = sdb_index_flat_search(h_, 1, q.Vector.data(), limit, filter ? f_off.data() : nullptr,
                                       filter ? f_ids.data() : nullptr, ids.data(), dists.data(), &count, SDB_MEM_HOST,
                                       nullptr)) {
      out.err = Error::wrap("failed to iterate over points", rc);
      return out;
    }
    const float weight = q.Weight ? *q.Weight : 1.0f;
    for (uint32_t i = 0; i < count; i++) {
      models::SearchResult sr;
      sr.NodeId = ids[i], sr.Distance = dists[i];
      sr.HybridScore = (-1 * weight * dists[i]);  // flat.go:114
      out.results.push_back(sr);
      out.set.insert(ids[i]);
    }
    return out;
  }

 private:
  IndexFlat() = default;
  models::IndexVectorFlatParameters parameters_;
  diskstore::Bucket *bucket_ = nullptr;
  sdb_index *h_ = nullptr;
  std::mutex write_mu_;
};
}  // namespace flat
// ---------------------------------------------------------------------------------------------------
namespace cluster {  // cluster/actions.go:275-379, the part that stays inside one server
// per-shard limit, actions.go:291-299
inline int PerShardLimit(int limit, int nShards, int maxSearchLimit = 75) {
  uint32_t out = 0;
  sdb_shard_limit((uint32_t)limit, (uint32_t)nShards, (uint32_t)maxSearchLimit, &out);
  return (int)out;
}

// ClusterNode.SearchPoints for the shards of this server (actions.go:316-376): instead of one RPCSearchPoints per
// shard and a host-side sort, every shard's rank makes ONE collective call that searches its shard, exchanges the
// per-shard top-k blocks (RCCL all-gather between GPUs; device copies when the shards share a GPU) and merges them
// on its GPU.  The compiled twin of integration/go/cluster/fanout_mi355x.go.
//
// Requests arrive concurrently (one goroutine per REST request in the reference, httpapi/v2/handlers.go:435-489).
// Each draws ONE ticket under a lock and presents it to every rank; the library lets a rank's calls into the exchange
// in ticket order whichever worker arrives first and refuses to merge blocks whose tags differ, so two racing
// requests can neither be paired up wrongly nor deadlock the collective.  Each rank has `workers` threads: that
// many requests are in flight per rank, the exchange of one under the graph walk of the next.
class GpuFanout {
 public:
  struct Result {
    std::vector<uint64_t> ids;    // [nq * limit] shard-local node ids
    std::vector<uint32_t> shards; // [nq * limit] which shard each id belongs to
    std::vector<float> dists;
    std::vector<uint32_t> counts; // [nq]
    Error err;
  };

  // indexes[r] lives on devices[r]; devices all distinct (one shard per GPU) or all the same (shards share a GPU)
  static std::pair<std::unique_ptr<GpuFanout>, Error> New(const std::vector<sdb_index *> &indexes,
                                                          const std::vector<int> &devices, unsigned workers = 2) {
    std::unique_ptr<GpuFanout> f(new GpuFanout());
    const int n = (int)indexes.size();
    if (n < 1 || devices.size() != indexes.size()) return {nullptr, Error("one device per shard index")};
    f->indexes_ = indexes;
    f->ranks_.assign((size_t)n, nullptr);
    if (int rc = sdb_cluster_create_local(n, devices.data(), f->ranks_.data()))
      return {nullptr, Error::wrap("could not create the shard exchange", rc)};
    uint64_t next = 1;
    sdb_cluster_next_ticket(f->ranks_[0], &next);
    f->ticket_ = next - 1;
    f->queues_.resize((size_t)n);
    for (int r = 0; r < n; r++) f->queues_[(size_t)r].reset(new Queue());
    for (int r = 0; r < n; r++)
      for (unsigned w = 0; w < (workers ? workers : 1); w++) f->threads_.emplace_back([p = f.get(), r] { p->loop(r); });
    return {std::move(f), Error()};
  }
  ~GpuFanout() {
    for (auto &q : queues_) {
      std::lock_guard<std::mutex> g(q->mu);
      q->stop = true;
      q->cv.notify_all();
    }
    for (auto &t : threads_)
      if (t.joinable()) t.join();
    for (auto *r : ranks_) sdb_cluster_destroy(r);
  }
  GpuFanout(const GpuFanout &) = delete;
  GpuFanout &operator=(const GpuFanout &) = delete;
  int shards() const { return (int)ranks_.size(); }
  // replace a shard's index (a shard that was rebuilt from its bucket); no request may be in flight
  void setIndex(int r, sdb_index *ix) { indexes_[(size_t)r] = ix; }

  // nq queries (row-major, dim floats each) over all local shards; `limit` is the request's original limit, the
  // per-shard limit of actions.go:291-299 is applied inside.  Thread-safe.
  Result SearchPoints(const float *queries, size_t nq, int limit, int searchSize) {
    Result out;
    const int n = (int)ranks_.size();
    Request req;
    req.queries = queries, req.nq = nq, req.limit = limit, req.search_size = searchSize;
    req.parts.resize((size_t)n);
    {  // every rank's GPU ends up with the same merged answer: only rank 0 copies it to the host
      Part &p = req.parts[0];
      p.ids.resize(nq * (size_t)limit), p.shards.resize(nq * (size_t)limit);
      p.dists.resize(nq * (size_t)limit), p.counts.resize(nq);
    }
    req.left = n;
    {
      // ticket and hand-over in one critical section: every rank's queue receives the requests in ticket order
      // (not needed for correctness -- the library orders by ticket -- but a worker then never waits on a ticket
      // whose request sits BEHIND its own in the same queue, which with one worker per rank would be a deadlock)
      std::lock_guard<std::mutex> g(mu_);
      req.ticket = ++ticket_;
      for (int r = 0; r < n; r++) {
        std::lock_guard<std::mutex> qg(queues_[(size_t)r]->mu);
        queues_[(size_t)r]->items.push_back(&req);
        queues_[(size_t)r]->cv.notify_one();
      }
    }
    {
      std::unique_lock<std::mutex> lk(req.mu);
      req.cv.wait(lk, [&] { return req.left == 0; });
    }
    // a shard that failed has said so inside the exchange: every rank reports an error for this request, and the
    // next request is served (actions.go:339-353 "could not search points")
    for (int r = 0; r < n; r++)
      if (req.parts[(size_t)r].rc != SDB_OK) {
        out.err = Error("shard " + std::to_string(r) + " could not search points: " + req.parts[(size_t)r].msg);
        return out;
      }
    Part &p0 = req.parts[0];  // every rank holds the same merged answer
    out.ids.swap(p0.ids), out.shards.swap(p0.shards), out.dists.swap(p0.dists), out.counts.swap(p0.counts);
    return out;
  }

 private:
  GpuFanout() = default;
  struct Part {
    std::vector<uint64_t> ids;
    std::vector<uint32_t> shards, counts;
    std::vector<float> dists;
    int rc = SDB_OK;
    std::string msg;
  };
  struct Request {
    const float *queries = nullptr;
    size_t nq = 0;
    int limit = 0, search_size = 0;
    uint64_t ticket = 0;
    std::vector<Part> parts;
    std::mutex mu;
    std::condition_variable cv;
    int left = 0;
  };
  struct Queue {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Request *> items;
    bool stop = false;
  };
  void loop(int r) {
    Queue &q = *queues_[(size_t)r];
    for (;;) {
      Request *req = nullptr;
      {
        std::unique_lock<std::mutex> lk(q.mu);
        q.cv.wait(lk, [&] { return q.stop || !q.items.empty(); });
        if (q.items.empty()) return;
        req = q.items.front();
        q.items.pop_front();
      }
      Part &p = req->parts[(size_t)r];
      const bool want = !p.ids.empty();  // rank 0 (NULL outputs: take part, return the verdict, copy nothing back)
      p.rc = sdb_cluster_search_batch(ranks_[(size_t)r], indexes_[(size_t)r], req->ticket, req->nq, req->queries,
                                      (uint32_t)req->limit, (uint32_t)req->search_size, want ? p.ids.data() : nullptr,
                                      want ? p.dists.data() : nullptr, want ? p.shards.data() : nullptr,
                                      want ? p.counts.data() : nullptr, SDB_MEM_HOST, nullptr);
      if (p.rc != SDB_OK) p.msg = sdb_last_error();
      std::lock_guard<std::mutex> g(req->mu);  // notified under the lock: `req` lives on its caller's stack
      if (--req->left == 0) req->cv.notify_all();
    }
  }
  std::vector<sdb_cluster *> ranks_;
  std::vector<sdb_index *> indexes_;
  std::vector<std::unique_ptr<Queue>> queues_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  uint64_t ticket_ = 0;
};
}  // namespace cluster
}  // namespace semadb
