// index.h -- host-side state of one HBM-resident Vamana index (internal; the public surface is
// include/semadb_amd.h).
#pragma once
#include <map>
#include <shared_mutex>
#include <thread>
#include <unordered_map>

#include "common.h"
#include "view_mutex.h"

namespace sdb {

constexpr uint32_t kAdjStride = 64;  // adjacency row stride in u32 (DegreeBound <= 64, models/index.go:279)

// Everything one in-flight batch needs besides the index itself.  One per stream.
struct Workspace {
  int device = 0;
  hipStream_t own_stream = nullptr;  // used by host-memory calls
  uint32_t *bitsets = nullptr;       // [nq][words]: the per-search VisitedBitSet (distset.go:89-116)
  size_t bitset_bytes = 0;
  void *scratch = nullptr;  // staging for host-memory calls
  size_t scratch_bytes = 0;
  float *lut = nullptr;  // [nq][M*K] product-quantizer distance tables of the batch
  size_t lut_bytes = 0;
  void *filter = nullptr;  // seeds / filter slot lists of a filtered batch
  size_t filter_bytes = 0;
  void *filter_aux = nullptr;  // a batch's filter bitmaps and their per-query counts (sdb_index_search_batch_bitmap)
  size_t filter_aux_bytes = 0;
  void *filter_host = nullptr;  // the same lists as the host threads write them: pinned, so that the upload is one DMA
  size_t filter_host_bytes = 0;
  hipEvent_t launched = nullptr;   // recorded behind the last search kernels that were given a graph version
  bool launched_valid = false;
  // a device-memory search returns right behind its kernels: `launched` then also marks the end of the call's device
  // work, and release_ws lets it stand for `done` instead of recording a second event behind it (one marker packet less
  // per batch on the stream)
  bool launched_is_tail = false, done_is_launched = false;
  bool busy = false;               // held by a call that has not returned yet
  bool pending = false;            // device work of an asynchronous call may still be running
  hipStream_t bound_stream = nullptr;
  hipEvent_t done = nullptr;
  int ensure_filter(size_t bytes);
  int ensure_filter_host(size_t bytes);
  int ensure_filter_aux(size_t bytes);
  int ensure_lut(size_t bytes);
  int ensure_bitsets(size_t bytes);
  int ensure_scratch(size_t bytes);
  void release();
};

struct PQState;  // pq.hip

// (sdb::ViewMutex, the writer-preferring lock around an index's committed view: view_mutex.h -- HIP-free, stress-tested
// under ThreadSanitizer by tests/host/test_concurrency.cpp)
}  // namespace sdb

struct sdb_index {
  sdb_index_params P{};
  sdb::RowLayout lay;
  uint32_t n = 0;    // rows in use (slot ids are 0..n-1)
  uint32_t n_dead = 0;  // of which deleted: tombstones with id 0, empty row, unreachable
  uint32_t cap = 0;  // rows allocated
  int64_t start_slot = -1;
  uint64_t max_node_id = 0;  // vamana.go:47
  float *d_slab = nullptr;   // [cap][lay.ld] float32, permuted rows (common.h RowLayout)
  uint32_t *d_adj = nullptr; // [cap][kAdjStride] neighbour slots, kNoSlot padded, edge order kept
  uint32_t *d_deg = nullptr; // [cap]
  uint32_t *d_clean = nullptr; // [cap] leading edges of a row produced by its last robustPrune (build.hip)
  // write-path cache: d_adjdist[r][e] = distFn(r, adj[r][e]) for e < d_dcount[r] (a prefix of the row).  A full
  // node that gets one more back-edge is re-pruned over neighbours + new point (insert.go:47-58); the
  // distances to its old neighbours are the ones computed when those edges were made -- same vectors, same
  // arithmetic -- so they are read back (256 B) instead of recomputed from 64 rows.  Full-precision store only.
  float *d_adjdist = nullptr;    // [cap][kAdjStride]
  uint32_t *d_dcount = nullptr;  // [cap]
  uint64_t *d_ids = nullptr; // [cap] slot -> node id
  // The start node's edges beyond the 64 of its adjacency row.  Stragglers of a delete are appended to the start
  // node with no bound (AddNeighbourIfNotExists, node.go:73-80 / prune.go:131-151); the list stays that long until
  // the start node is next pruned (a back-edge from an insert, insert.go:47-58, or a delete among its edges).
  // Host copy in edge order; the device copy is padded with kNoSlot to a multiple of 64 so the search reads it in
  // row-sized chunks.  Every other node is bounded by DegreeBound <= 64.
  std::vector<uint32_t> h_start_ext;
  uint32_t *d_start_ext = nullptr;
  uint32_t start_ext_cap = 0;
  std::vector<uint64_t> h_ids;
  bool dense_ids = true;  // ids[i] == ids[0] + i  (then no hash map is needed)
  std::unordered_map<uint64_t, uint32_t> id2slot;
  // product quantizer attachment (product.go): codes per slot + tables
  const sdb_pq *pq = nullptr;
  uint8_t *d_codes = nullptr;
  // Quantizers of up to kAdjCodesMaxM sub-vectors: every node's neighbours' code rows once more, BEHIND ITS ADJACENCY
  // ROW -- [cap][kAdjStride][M] bytes, entry e = the code row of edge e -- the way the reference keeps a node's
  // neighbours as cached point objects (node.go:37-54 LoadNeighbours: one fetch gives ids and codes).  A hop of the
  // quantized walk then reads 256 B of ids and 64 M contiguous bytes of codes with ONE round trip, where the gather
  // by slot was a second, dependent one and paid a 64-byte sector per 8-byte code (M = 8: 6.1 x the algorithmic
  // bytes).  One block per adjacency copy (graph versions below): searches read the committed one; the writer's is
  // brought up to date from the dirty-row flags when a transaction commits (k_adjcodes_rows) -- no write kernel
  // maintains it, and the build's own searches gather by slot as before.
  uint8_t *d_adjcodes = nullptr;  // the writer's copy (beside d_adj)
  uint8_t *r_adjcodes = nullptr;  // the committed copy (beside r_adj)
  static constexpr uint32_t kAdjCodesMaxM = 32;
  bool has_adjcodes() const { return d_adjcodes != nullptr; }
  int alloc_adjcodes();                       // after a quantizer has been attached (cap rows)
  int rebuild_adjcodes(hipStream_t stream);   // every row of both copies from (adjacency, codes); maintenance calls
  // ---- graph versions (SURVEY 8b Threading; shard/cache/manager.go:159-181) --------------------------------
  // A search walks the last COMMITTED graph while a write transaction changes the graph: everything a walk reads
  // and a write changes in place exists twice -- adjacency rows, the slot -> id table, the start node's overflow
  // list.  d_adj / d_ids / d_start_ext above are the writer's copies (all write paths use them unchanged);
  // r_adj / r_ids / r_start_ext are what `view` hands to searches.  Commit swaps the two sets and then brings the
  // writer's (now stale) set up to date from the dirty-row flags the write kernels left, once the searches that
  // were launched on it have drained (stream-side waits on their events).  Outside a transaction both sets are
  // identical.  The slab is append-only (rows past view.n are invisible), so it exists once.
  uint32_t *r_adj = nullptr;
  uint64_t *r_ids = nullptr;
  uint32_t *r_start_ext = nullptr;
  uint32_t r_start_ext_cap = 0;
  uint8_t *d_dirty = nullptr;  // [cap] != 0: the open transaction wrote this row's adjacency or id
  struct View {
    uint32_t n = 0;  // committed rows
    const uint32_t *adj = nullptr;
    const uint64_t *ids = nullptr;
    const uint32_t *start_ext = nullptr;
    uint32_t start_ext_n = 0;
    const uint8_t *adj_codes = nullptr;  // r_adjcodes, or NULL
  } view;
  // readers: shared from taking `view` until their kernels are enqueued and their event recorded; writers:
  // exclusive while they change the host-side id tables or publish a view
  mutable sdb::ViewMutex view_mu;
  uint64_t view_gen = 1;  // counts the views published (view_mu held exclusively)
  // The committed view's id -> slot table on the device, for the filters of a table whose ids are no longer
  // consecutive (deletes, arbitrary ids): an open-addressing table over view.ids[0, view.n), tombstones left out,
  // built by the first filtered search of a view and kept until the next view is published.  A search reads it
  // under the shared view lock, so a rebuild (another view) never meets a reader of the old one.
  struct IdMap {
    uint64_t *keys = nullptr;  // 0 = empty cell (0 is no node id)
    uint32_t *vals = nullptr;
    uint32_t cells = 0;        // power of two, >= 2 * view.n
    uint64_t gen = 0;          // the view it was built from; 0 = none
    std::mutex mu;
  };
  mutable IdMap idmap;
  int ensure_idmap(const View &vw, hipStream_t stream) const;  // view_mu held (shared)
  bool in_tx = false, tx_explicit = false;
  bool tx_dirty = false;  // the open transaction has changed the writer's copy or the host tables (sdb_index_abort_write)
  uint32_t tx_n0 = 0;  // rows at the start of the open transaction
  uint32_t tx_dead0 = 0;        // tombstones, largest node id and id-table form at the start of it (rollback)
  uint64_t tx_max_id0 = 0;
  int rollback();               // sdb_index_abort_write: the writer's copy and the host tables back to the committed state
  std::unordered_map<uint64_t, uint32_t> tx_deleted;  // ids the open transaction has removed -> their slots
  int begin_write();
  int commit(hipStream_t stream);   // publish the writer's state; `stream` carries the write
  int publish_full();               // exclusive maintenance calls (load, attach_pq ...): drain, copy everything
  int64_t slot_of_committed(uint64_t id, uint32_t view_n) const;  // what a search may resolve; view_mu held
  // sdb_index_set_tuning
  uint32_t tune_hub_min = 512, tune_hash_limit = 0;
  bool tune_no_hash = false;
  bool tune_wide_hash = false;  // quantized searches with 32-bit visited-set cells (4 walks per CU) instead of 16-bit ones (6)
  uint32_t tune_hash16_probes = 0;  // test knob: probe budget of the 16-bit visited set (0 = 15 buckets)
  bool tune_host_filters = false;  // filter ids are resolved to slots by the host's hash map even when the table's ids are consecutive
  uint32_t tune_wide_walk = 0;  // the workgroup-per-query walk of small calls: 0 = up to 256 queries, 1 = never, 2 = always
  bool tune_no_zero_copy = false;  // A/B and parity tests: host-memory searches stage even page-locked buffers
  // Two-precision hop (SDB_TUNE_SKETCH; search_kernel.h SearchArgs::sketch): a float16 copy of the slab's rows, an
  // optional cache like d_adjcodes.  It describes the rows of the view published as number `sketch_gen`; a search
  // uses it only while that is the current view, and commit / publish_full rebuild it behind the searches of the old one.
  uint32_t tune_sketch = 0;  // 0 off, 1 on, 2 on + audit (every discarded neighbour is evaluated exactly as well and checked)
  uint16_t *d_sketch = nullptr;
  float *d_sketch_norm = nullptr;  // [sketch_cap] ||y16||^2 per row (the euclidean form of the first stage)
  uint32_t sketch_cap = 0;   // rows d_sketch has room for
  std::atomic<uint64_t> sketch_gen{0};  // view_gen the copy was built for (0: none); written last by build_sketch, read by searches under the shared view lock
  float sk_emax = 0.0f, sk_ymax = 0.0f;
  unsigned long long *d_sk_counters = nullptr;  // [0] neighbours discarded on their float16 distance, [1] contradicted (audit)
  bool sketch_supported() const;               // cosine / dot rows of whole 32-float blocks, one of the walk's register layouts
  int build_sketch(hipStream_t stream, uint32_t from = 0);  // (re)build for the rows as they are (from > 0: only the rows from there on); failure to allocate leaves it off
  void drop_sketch();
  bool tune_no_defer = false;  // A/B and parity tests: every back-edge re-prune runs in k_backedges (BuildArgs::def_*)
  uint32_t tune_pq_narrow = 0;  // 1: quantized searches never take a multi-wave walk (k_greedy_search_pqw, k_greedy_search_pq2): A/B and parity tests
  bool tune_no_mfma = false;  // exact scan of dot/cosine rows on the packed-FMA kernel instead of the matrix cores
  uint32_t tune_no_tile = 0;  // 0: LDS-tiled prune of new nodes, 1: one-wave kernel only, 2: tiled with 4 waves instead of 8, 3: tiled without the separate selection kernel (measurement)
  // a write that failed after it had started to change the graph leaves it unusable: every later call fails
  // until the host rebuilds the index from the bucket -- the reference scraps its cache on any error inside a
  // write transaction the same way (shard/cache/manager.go:231-240)
  bool broken = false;
  // counters of the last insert_batch (sdb_index_build_stats); device-side, added to by the kernels.  kStatCopies
  // copies of kStatStride slots, one 128-byte line each: a wave adds to the copy its block index picks, so tens of
  // millions of waves do not queue up on one address (a single set cost the build 1 s of its 3); summed on read
  static constexpr uint32_t kStatCopies = 64, kStatStride = 16;
  uint64_t *d_bstats = nullptr;
  // measurement hook: events around the last K2 launch
  bool profiling = false;
  static constexpr uint32_t kProfRing = 256;
  std::vector<hipEvent_t> ev0, ev1;  // ring of event pairs
  uint64_t prof_count = 0;           // launches recorded since the last read
  mutable std::mutex mu;
  mutable std::vector<sdb::Workspace *> pool;

  int64_t slot_of(uint64_t id) const;
  int reserve(uint32_t rows);
  int sync_start_ext();  // h_start_ext -> device
  sdb::Workspace *acquire_ws(hipStream_t stream, bool async) const;
  void release_ws(sdb::Workspace *ws, hipStream_t stream, bool async) const;
};
