/*
 * semadb_amd.h -- C ABI of the MI355X-native Vamana search path for SemaDB.
 *
 * This is the drop-in boundary: the entry points a SemaDB maintainer binds from Go through cgo
 * (INTEGRATION.md shows the shim) to put the hot path -- distance/, shard/vectorstore,
 * shard/index/vamana search + insert, utils/kmeans + the product quantizer, and the cluster
 * top-k merge -- on an MI355X.  Plain pointers and sizes only; no C++/torch types.
 *
 * Each function cites the reference interface it replaces (file:line relative to the reference
 * repository Semafind/semadb).
 *
 * Conventions
 *   - every function returns an sdb_status (0 = ok) and never aborts the process
 *     (CONTRIBUTING.md:150 "no panics"); sdb_last_error() returns a thread-local message.
 *   - `mem` says where caller buffers live: SDB_MEM_HOST (Go slices; the callee copies and never
 *     retains the pointer, as cgo requires) or SDB_MEM_DEVICE (HBM pointers; the call is
 *     asynchronous on `stream` and the caller synchronises).  Host calls synchronise before
 *     returning.
 *   - node ids are the reference's uint64 node ids (shard/idcounter.go); id 1 is the start node
 *     (vamana.go:28) and ids 0/1 are rejected on write (vamana.go:150-157).
 *   - vectors are row-major float32, `dim` floats per row (conversion.BytesToFloat32 layout).
 *   - search calls may be issued concurrently from many threads on one index (the reference
 *     serves searches from concurrent goroutines under an RLock, shard/cache/manager.go:163), also
 *     while ONE thread writes (insert / delete): they see the last committed graph (see
 *     sdb_index_begin_write).  Writers are exclusive among themselves, like the shard's write lock;
 *     load / attach_pq / set_codes / set_start are maintenance calls with no search in flight.
 */
#ifndef SEMADB_AMD_H
#define SEMADB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDB_ABI_VERSION 2

typedef enum {
  SDB_OK = 0,
  SDB_ERR_INVALID = 1,   /* bad argument / parameter out of the reference's range        */
  SDB_ERR_DEVICE = 2,    /* HIP runtime error (no GPU, launch failure, out of HBM)        */
  SDB_ERR_STATE = 3,     /* call not valid in this state (e.g. search before start node)  */
  SDB_ERR_NOT_FOUND = 4, /* unknown node id                                               */
  SDB_ERR_EXISTS = 5     /* id already present (update path is host-side, not here)       */
} sdb_status;

/* models.Distance* names accepted by distance.GetFloatDistanceFn (distance/distance.go:70-83) */
#define SDB_METRIC_EUCLIDEAN 0
#define SDB_METRIC_COSINE 1
#define SDB_METRIC_DOT 2

#define SDB_MEM_HOST 0
#define SDB_MEM_DEVICE 1

#define SDB_STARTID 1ull /* vamana.go:28 */

typedef struct sdb_index sdb_index;
typedef struct sdb_pq sdb_pq;

const char *sdb_last_error(void);
int sdb_abi_version(void);
/* number of visible MI355X devices; SDB_ERR_DEVICE if the HIP runtime finds none */
int sdb_device_count(int *count);
/* Page-locked host memory for SDB_MEM_HOST buffers a host re-uses call after call (a batcher's query and result
 * slabs).  A search whose queries and outputs live in such memory (or in memory the caller page-locked itself:
 * hipHostMalloc / hipHostRegister) is ONE kernel launch: the walk reads each query from the caller's buffer once, when
 * its wave starts, and writes the [nq][limit] results straight into the caller's arrays -- no staging copies, no copy
 * packets around the kernel.  Other copies to and from it are single DMAs and truly asynchronous, where pageable memory
 * (a Go slice) is staged through the driver.  Plain pageable buffers keep working everywhere; this is an optimisation,
 * not a requirement. */
int sdb_host_alloc(size_t bytes, void **out);
int sdb_host_free(void *p);

/* ---------------------------------------------------------------------------------------------
 * distance/  (K1)
 * ------------------------------------------------------------------------------------------- */
/* Replaces distance.FloatDistFunc as obtained from GetFloatDistanceFn (distance/distance.go:11,
 * 70-83) with the amd64 overrides installed (distance/distance_amd64.go:19-27): asm.Dot
 * (distance/asm/dot.s:7-55) and asm.SquaredEuclideanDistance (distance/asm/euclidean.s:7-65),
 * wrapped by dotProductDistance / cosineDistance (distance.go:19-25).  Results are bit-identical
 * to that path: the kernel keeps the assembly's 32 partial sums, FMA order and reduce tree.
 * Batched: out[q*nc + c] = dist(queries[q], candidates[c]); rows are `dim` floats. */
int sdb_distance_batch(int metric, uint32_t dim, const float *queries, uint64_t nq,
                       const float *candidates, uint64_t nc, float *out, int mem, int device,
                       void *stream);

/* ---------------------------------------------------------------------------------------------
 * shard/index/vamana + shard/vectorstore (plain)  (K2, K3, K4)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  uint32_t dim;          /* models.IndexVectorVamanaParameters.VectorSize  1..4096 (models/index.go:276) */
  uint32_t metric;       /* DistanceMetric: SDB_METRIC_*                                         */
  uint32_t search_size;  /* SearchSize used while inserting, 25..75 (models/index.go:278)        */
  uint32_t degree_bound; /* DegreeBound 32..64 (models/index.go:279)                             */
  float alpha;           /* Alpha 1.1..1.5 (models/index.go:280)                                 */
  int32_t device;        /* HIP device ordinal that pins the slab and the graph                  */
  uint64_t capacity;     /* rows to reserve in HBM up front (grows on demand); 0 = default       */
  uint32_t strict;       /* != 0: enforce the reference's parameter ranges (models/index.go:284-313) */
} sdb_index_params;

/* vamana.NewIndexVamana (vamana.go:54-81) + vectorstore.New for the plain store (vectorstore.go:47-96).
 * The HBM state is a cache of the bucket contents, like the reference's ItemCache. */
int sdb_index_create(const sdb_index_params *params, sdb_index **out);
int sdb_index_destroy(sdb_index *ix);

/* setupStartNode (vamana.go:93-120).  The reference draws the start vector from an unseeded RNG;
 * the caller supplies it (a Go shim passes the vector it read from / wrote to the bucket). */
int sdb_index_set_start(sdb_index *ix, const float *vec, int mem);

/* Bulk load of an existing graph: what the reference reads lazily from the bucket through
 * plainPoint.ReadFrom (shard/vectorstore/plain.go:125-141) and graphNode.ReadFrom
 * (shard/index/vamana/node.go:96-111).  CSR: offsets[n+1], edges hold uint64 node ids
 * (conversion.BytesToEdgeList).  Edges to ids that are not loaded are dropped, the way
 * ItemCache.GetMany skips them (shard/cache/itemcache.go:109-128); a repeated id inside one row
 * keeps its first occurrence (later ones can never pass CheckAndVisit, distset.go:174).
 * ids == NULL means row i has id i+1 (row 0 is the start node).  The start node (id 1) must be
 * present.  vectors follow `mem`; ids/offsets/edges are always host memory. */
int sdb_index_load(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors,
                   const uint64_t *offsets, const uint64_t *edges, int mem);

/* IndexVamana.InsertUpdateDelete, insert branch (vamana.go:127-201) -> insertSinglePoint
 * (insert.go:16-68): greedySearch(vec, 1, SearchSize) -> robustPrune (search.go:106-138) ->
 * back-edges with re-prune.  The reference runs NumCPU-1 inserts concurrently, so its graph
 * depends on goroutine interleaving; here inserts run in deterministic rounds whose points search
 * one frozen snapshot.  A round holds at most `round_size` points (0 = 16384) and never more than
 * 2 % of the nodes already in the graph, so early rounds are sequential; round_size = 1 reproduces
 * a sequential insertSinglePoint loop exactly.
 * ids == NULL assigns max_id+1.. in order.  ids 0 and 1 are rejected (vamana.go:150-157). */
int sdb_index_insert_batch(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors,
                           int mem, uint32_t round_size, void *stream);

/* IndexVamana.InsertUpdateDelete, delete branch (vamana.go:175-233): removeInboundEdges (prune.go:88-154) =
 * EdgeScan (node.go:142-199) + pruneDeleteNeighbour (prune.go:12-84) for every node with an edge into the
 * delete set + stragglers (valid nodes nobody points at) re-attached to the start node (prune.go:131-151),
 * then the nodes are dropped from the store (vamana.go:228-233).  Unknown ids are skipped
 * (vamana.go:161-163); ids 0 and 1 are errors.  The reference re-attaches stragglers in Go-map order
 * (unspecified): here in storage order.  Its start node grows without bound (AddNeighbourIfNotExists,
 * node.go:73-80) and so does this one: edges past the 64 of a device row are kept on an overflow list that
 * searches, inserts, deletes, export and load read after the row, until the start node is next pruned.
 * An update (vamana.go:170-174,247-251) is delete_batch followed by insert_batch with the same id. */
int sdb_index_delete_batch(sdb_index *ix, uint64_t n, const uint64_t *ids, void *stream);

/* Write transactions.  IndexVamana.InsertUpdateDelete (vamana.go:127-263) is ONE transaction of the shard made of
 * several calls here (inserts, one delete scan, re-inserts of updated points); the reference runs it under the
 * shard's write lock while searches are served from the cache or, if that is locked, from a cold index built on
 * the last committed bucket (shard/cache/manager.go:159-181).  Here searches (sdb_index_search_batch,
 * sdb_index_flat_search, sdb_cluster_search_batch) always walk the last COMMITTED graph: whatever a walk reads and
 * a write changes in place exists twice in HBM, a write changes the writer's copy only, and commit publishes it
 * (the searches launched before drain on the old copy; nobody blocks).  Between sdb_index_begin_write and
 * sdb_index_commit all insert / delete calls belong to one transaction; without begin_write every such call is a
 * transaction of its own.  One writer at a time (the shard's write lock).  sdb_index_abort_write rolls an open
 * transaction back.  A write that a DEVICE failure interrupts half-way leaves the handle unusable (SDB_ERR_STATE), the
 * host reloads from the bucket -- the reference scraps its cache after any error inside a transaction the same way
 * (manager.go:231-240). */
int sdb_index_begin_write(sdb_index *ix);
int sdb_index_commit(sdb_index *ix, void *stream);
/* Leave a transaction without committing it -- the error path of a host's InsertUpdateDelete (a bad point found
 * after begin_write, a failed call, a cancelled request).  The index goes back to what it was at begin_write: the
 * rows the transaction wrote take back the committed copy that the searches have been walking all along, the rows it
 * appended are dropped, the id tables follow (SDB_OK; milliseconds per million rows).  The reference has no rollback
 * -- after an error inside a transaction its cache manager scraps the shard's cache and rebuilds it from the bucket
 * (shard/cache/manager.go:231-240).  SDB_ERR_STATE only for a handle that a device failure had already left unusable.
 * Without an open transaction: SDB_OK.  Insert / delete calls that fail before their first change close the
 * transaction they opened for themselves. */
int sdb_index_abort_write(sdb_index *ix);
/* test support: rows on which the committed and the writer's copy of the graph differ (0 outside a transaction) */
int sdb_index_version_diff(const sdb_index *ix, uint64_t *rows);

/* IndexVamana.EdgeScan (node.go:142-199), read-only: the ids of the valid nodes that have an edge to a member of
 * the delete set (to_prune) and of the valid nodes that no valid node points at (to_save; the start node never
 * is).  The reference returns both in Go-map order; here storage order.  Counts are always returned; ids are
 * written when the arrays are given and large enough (the node count always is).  sdb_index_delete_batch runs
 * the same scan internally -- this entry point exists because EdgeScan is part of the package's exported surface. */
int sdb_index_edge_scan(const sdb_index *ix, uint64_t n, const uint64_t *delete_ids, uint64_t *to_prune,
                        uint64_t cap_prune, uint64_t *n_prune, uint64_t *to_save, uint64_t cap_save,
                        uint64_t *n_save, void *stream);

/* per-query search trace; every pointer may be NULL.  Arrays follow the same `mem` as the
 * outputs of the call. */
typedef struct {
  uint32_t *n_dist;     /* [nq] distFn evaluations (distset.go:179)                           */
  uint32_t *n_hop;      /* [nq] expanded nodes = len(visitedSet) (search.go:73)               */
  uint32_t *n_edges;    /* [nq] edge ids read = sum deg(expanded)                             */
  uint64_t *visit_ids;  /* [nq][visit_cap] node ids in expansion order                        */
  uint32_t visit_cap;
} sdb_search_trace;

/* IndexVamana.Search (vamana.go:278-310) over greedySearch (search.go:9-102), for nq queries
 * at once (the reference has no batch entry point: a host-side micro-batcher coalesces
 * concurrent Search calls).  limit = query.Limit, search_size =
 * query.SearchSize; search_size < limit is an error (search.go:23-25).
 * Optional filter (the roaring bitmap argument): filter_offsets[nq+1] into filter_ids, each
 * query's ids ascending; NULL = no filter.  Filter arrays are host memory (pinned -- sdb_host_alloc -- they go up
 * in one DMA); they are resolved to slots on the device (a subtraction for a table with consecutive ids, a probe of
 * the committed view's id -> slot table otherwise).
 * Outputs: out_ids[nq*limit], out_dists[nq*limit] (ascending distance, start node removed),
 * out_counts[nq].  HybridScore = -1 * dist * weight is left to the caller (vamana.go:303). */
int sdb_index_search_batch(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit,
                           uint32_t search_size, const uint64_t *filter_offsets,
                           const uint64_t *filter_ids, uint64_t *out_ids, float *out_dists,
                           uint32_t *out_counts, const sdb_search_trace *trace, int mem,
                           void *stream);

/* The same search with the filters handed over as BITMAPS: query q's filter is the set
 * { filter_first_id[q] + i : bit i of its words is set }, its words being filter_words[filter_word_offsets[q] ..
 * filter_word_offsets[q + 1]) -- bit i of the set is bit i % 64 of word i / 64.  The reference's filter IS a bitmap
 * (roaring64, search.go:33-51,93): a dense roaring container is 1 024 such words, and a filter of 100 000 ids out of a
 * million is an eighth of the bytes of its id list in this form -- what a large filter costs is its upload.  Unknown ids
 * (bits outside the table) are skipped like GetMany does (itemcache.go:109-128); the seeds are the first searchSize set
 * bits (:41-48).  Filter arrays are host memory; everything else as sdb_index_search_batch, same answers bit for bit.
 * The bitmaps are expanded to slots on the device (for a table whose ids are not consecutive: to 64-bit ids first,
 * which the view's id -> slot table resolves). */
int sdb_index_search_batch_bitmap(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit, uint32_t search_size,
                                  const uint64_t *filter_first_id, const uint64_t *filter_word_offsets,
                                  const uint64_t *filter_words, uint64_t *out_ids, float *out_dists, uint32_t *out_counts,
                                  const sdb_search_trace *trace, int mem, void *stream);

/* plainStore.DistanceFromFloat (shard/vectorstore/plain.go:76-85) batched: distances from each
 * query to an explicit list of stored node ids (nc per query, cand_ids[nq*nc], host memory).
 * Unknown ids give math.MaxFloat32, as the reference does for a foreign point type
 * (plain.go:78-82).  out[nq*nc] follows `mem`. */
int sdb_index_distance_batch(sdb_index *ix, uint64_t nq, const float *queries, uint64_t nc,
                             const uint64_t *cand_ids, float *out, int mem, void *stream);

/* Test / measurement knobs of ONE index (never read from the environment: a serving process must not change
 * algorithmic thresholds through getenv).  None of them changes any result -- the parity suites run with
 * several settings of each to prove that.
 *   SDB_TUNE_HUB_MIN     back-edge requests in one build round that send a target to the chip-wide prune
 *                        (bigprune.inc); default 512, minimum 2
 *   SDB_TUNE_HASH_LIMIT  ids a query's LDS visited hash set may hold before it spills to the HBM bitset;
 *                        default 6000 (also the maximum)
 *   SDB_TUNE_NO_HASH     != 0: the visited set is the HBM bitset from the start
 *   SDB_TUNE_NO_TILE     1: a new node's robustPrune reads candidate rows from global memory instead of
 *                        staging them in LDS; 2: staged, four waves per node instead of eight; 3: staged, the
 *                        selection loop inside the staging kernel instead of a kernel of its own
 *   SDB_TUNE_NO_MFMA     != 0: the exact scan of dot / cosine tables runs on the packed-FMA kernel (the one
 *                        euclidean uses) instead of the matrix cores
 *   SDB_TUNE_WIDE_HASH   != 0: searches over a quantized store keep their visited ids in 32-bit LDS cells (four
 *                        walks per CU) instead of the 16-bit cells used for stores of up to 2^24 rows (six)
 *   SDB_TUNE_PQ_NARROW   1: every search over a quantized store stays on the one-wave kernel -- a per-query table of
 *                        more than 64 KB (M = 128 .. 384 at K = 256) in global memory (round 2) instead of the four-wave
 *                        walk that keeps it in LDS and registers, a table of up to 8 KB (M = 8 at K = 256) without the
 *                        second wave that runs AddWithLimit (round 5); 2: M = 192 takes the four-wave variant with one
 *                        query per CU instead of two; 3: the four-wave walk of M = 128 / 192 keeps the candidate array in
 *                        the walker instead of in the helper wave next to it
 *   SDB_TUNE_WIDE_WALK   the walk of calls with few queries (one REST request is one query, vamana.go:278-310): a
 *                        workgroup of sixteen waves per query -- one walks, all split every hop's rows and compute the
 *                        likely next hop's distances ahead -- instead of one wave per query (eight waves for 257 .. 512
 *                        queries).  0 (default): calls of up to 512 queries on a full-precision store, plain or
 *                        filtered, vectors of 32 .. 1055 floats; 1: never; 2: always (any number of queries).
 *                        Same ids, distance bits, visit order and counters either way.
 *   SDB_TUNE_HOST_FILTERS  != 0: the filter ids of a search are turned into slots by the host (hash map, threads) even
 *                        where the device would do it (A/B and parity tests)
 *   SDB_TUNE_NO_DEFER    != 0: the build's back-edge re-prunes all run in the one-wave kernel and recompute what they
 *                        need from rows; by default those that cannot be settled from a few rows of pair distances (many
 *                        new candidates) are handed to the LDS-tiled prune, which reads the candidates' rows once, and
 *                        a bulk insert keeps the pair distances of edges it appends without a prune (4 KB per row for
 *                        the length of the call) for the re-prune that meets them later.  Same graph either way.
 *   SDB_TUNE_HASH16_PROBES  buckets a key of the 16-bit-cell set may try before the walk spills to the HBM bitset
 *                        (0 = all 15; 1..15).  With 15 that spill is a one-in-ten-million event; a test sets 1 or 2
 *                        to walk through it
 *   SDB_TUNE_NO_ZERO_COPY  != 0: a host-memory search stages queries and results through device buffers even when the
 *                        caller's buffers are page-locked (sdb_host_alloc) and the kernel could read / write them in
 *                        place (A/B and parity tests)
 *   SDB_TUNE_SKETCH      1: two-precision hop for batch searches of plain tables, any metric (rows of whole 32-float blocks,
 *                        up to 768 floats; no quantizer, no filter): the index keeps a float16 copy of its rows (+ 50 %
 *                        of their memory; a commit converts the rows it appended) and a hop reads a new neighbour's float32 row only
 *                        when its float16 distance does not PROVE that AddWithLimit discards it (distset.go:184: the
 *                        distance of a discarded neighbour is never used again).  Ids, distances, visit order and
 *                        counters are the same bits either way.  2: the same, and every discarded neighbour is
 *                        evaluated exactly as well; sdb_index_sketch_stats counts decisions the exact distance
 *                        contradicts (must stay 0).  0: off, the copy is freed.  Off by default. */
#define SDB_TUNE_HUB_MIN 1
#define SDB_TUNE_HASH_LIMIT 2
#define SDB_TUNE_NO_HASH 3
#define SDB_TUNE_NO_TILE 4
#define SDB_TUNE_NO_MFMA 5
#define SDB_TUNE_WIDE_HASH 6
#define SDB_TUNE_HASH16_PROBES 7
#define SDB_TUNE_PQ_NARROW 8
#define SDB_TUNE_WIDE_WALK 9
#define SDB_TUNE_HOST_FILTERS 10
#define SDB_TUNE_NO_DEFER 11
#define SDB_TUNE_NO_ZERO_COPY 12
#define SDB_TUNE_SKETCH 13
int sdb_index_set_tuning(sdb_index *ix, int key, uint64_t value);
/* SDB_TUNE_SKETCH's counters since the knob was last set: out[0] = neighbours discarded on their float16 distance,
 * out[1] = of those, the ones whose exact distance would have been kept (audit mode only; a non-zero value is a bug),
 * out[2] = 1 when searches currently use the float16 copy (it exists and describes the committed rows), else 0. */
int sdb_index_sketch_stats(sdb_index *ix, uint64_t out[3]);

/* Counters of the most recent sdb_index_insert_batch call (the C3 roofline, SURVEY 8d: bytes = sum over inserts
 * of search bytes + prune pair-distance rows * d * 4).  out[0..n) with n = min(cap, SDB_BUILD_STATS):
 *   [0] distFn evaluations of the insert searches     [1] edge ids read by them
 *   [2] pair distances evaluated by the new nodes' robustPrune (search.go:132)
 *   [3] pair / node distances evaluated from rows by the back-edge re-prunes (insert.go:47-58)
 *   [4] distances those re-prunes took from a cache instead (per-edge cache, the searches' tables)
 *   [5] back-edge requests (insert.go:36)              [6] re-prunes (:47-58)      [7] plain appends (:62)
 *   [8] candidate rows staged from HBM by the new nodes' robustPrune (each read once)
 *   [9] rounds                                         [10] hub prunes (bigprune.inc) */
#define SDB_BUILD_STATS 11
int sdb_index_build_stats(const sdb_index *ix, uint64_t *out, uint32_t cap);

/* vecStore.GetMany (shard/vectorstore/plain.go:26-45 over ItemCache.GetMany, shard/cache/itemcache.go:109-128):
 * the stored float32 vectors of n ids, in request order, out[i * dim ..]; found[i] = 0 (and a zero row) for an id
 * that is not stored -- the reference silently skips those, the flag lets the caller do the same.  Get = n of 1;
 * ForEach = sdb_index_export.  Committed state, like a search.  out / found are host memory. */
int sdb_index_get_vectors(const sdb_index *ix, uint64_t n, const uint64_t *ids, float *out, uint8_t *found);

/* vecStore.Exists (shard/vectorstore/plain.go:21-24) for n ids at once: out[i] = 1 if the id is stored
 * (the start node counts), else 0.  Host-side table lookup, no device work. */
int sdb_index_exists_batch(const sdb_index *ix, uint64_t n, const uint64_t *ids, uint8_t *out);

/* Measurement hook (the reference logs the greedy-search duration at debug level, vamana.go:284):
 * when enabled, HIP events are recorded on the launch stream immediately around the K2 kernel of
 * every search_batch (not thread-safe: a single measuring caller); sdb_index_last_search_ms waits for
 * the most recent one and returns its duration in milliseconds. */
int sdb_index_set_profiling(sdb_index *ix, int enabled);
int sdb_index_last_search_ms(sdb_index *ix, float *ms);
/* durations of the most recent profiled K2 launches (oldest first, at most 256 kept); resets the log */
int sdb_index_profile_read(sdb_index *ix, float *ms, uint32_t cap, uint32_t *n);

/* IndexFlat.Search (shard/index/flat/flat.go:76-132): exact scan of the stored vectors with the same
 * distance closure (vecStore.DistanceFromFloat), keeping the `limit` closest with the reference's
 * `dist >= tail -> skip` rule (:104).  The reference walks the store in Go-map order (unspecified); here
 * the walk is in storage order, so among equal distances the first stored stays.  The graph's start node
 * is not a point and never appears.  Optional filter as in sdb_index_search_batch (only ids in the filter
 * are scanned, flat.go:100).  limit <= 128.  Also the exact-kNN ground truth for recall. */
int sdb_index_flat_search(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit,
                          const uint64_t *filter_offsets, const uint64_t *filter_ids, uint64_t *out_ids,
                          float *out_dists, uint32_t *out_counts, int mem, void *stream);

/* vecStore.Set without graph maintenance -- what IndexFlat.InsertUpdateDelete does for a point with a vector
 * (flat.go:46-49): insert, or replace the stored vector of an id that exists (the old row becomes a tombstone, the
 * new one is appended).  ids == NULL assigns max_id+1.. .  On an index WITH a graph an existing id is rejected: a
 * graph update is delete_batch + insert_batch.  An id may appear once per call. */
int sdb_index_set_vectors(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors, int mem);
/* vecStore.Delete as IndexFlat.InsertUpdateDelete calls it for a point without a vector (flat.go:50-52); ids that are
 * not stored are skipped.  Only for an index without a graph. */
int sdb_index_remove_vectors(sdb_index *ix, uint64_t n, const uint64_t *ids);

/* cache.Cachable.SizeInMemory (vamana.go:83-85): bytes of HBM held */
int sdb_index_size_in_memory(const sdb_index *ix, int64_t *bytes);
/* number of nodes (start node included) / edges */
int sdb_index_stats(const sdb_index *ix, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *max_node_id);
/* Storage rows: in use (live nodes + tombstones of deleted / updated points) and of those dead.  A deleted point's
 * row stays behind as a tombstone (id 0, no edges, unreachable) and inserts always append, so under an update-heavy
 * load `rows` grows until the host rebuilds the index from the bucket (what a SemaDB cache eviction + reload does
 * anyway: shard/cache/manager.go evicts whole shards); per-search cost depends on live nodes only, except for the
 * bitset fallback's clear, which is sized by `rows`.  sdb_index_compact squeezes the tombstones out. */
int sdb_index_row_usage(const sdb_index *ix, uint64_t *rows, uint64_t *dead);
/* Drop the tombstones: the live rows move down in storage order (so everything that depends on storage order --
 * the exact scan's tie rule, straggler re-attachment -- is unchanged), adjacency is renumbered, ids, graph and every
 * later answer stay what they were.  The reference frees deleted nodes when it flushes (node.go:129-134); call
 * this from the same place when dead / rows is worth it.  A maintenance call: no write transaction open; searches
 * wait for it (tens of milliseconds per million rows).  Needs room for a second copy of the index while it runs. */
int sdb_index_compact(sdb_index *ix);
/* Copy the graph back out in bucket order (what flush() writes, vamana.go:265-276): ids[n],
 * vectors[n*dim] (NULL to skip), offsets[n+1], edges[n_edges] as node ids.  Host memory. */
int sdb_index_export(const sdb_index *ix, uint64_t *ids, float *vectors, uint64_t *offsets,
                     uint64_t *edges);

/* ---------------------------------------------------------------------------------------------
 * cluster fan-out merge  (C1's host half)
 * ------------------------------------------------------------------------------------------- */
/* ClusterNode.SearchPoints merge (cluster/actions.go:357-376): for each of nq queries take the
 * n_shards per-shard result lists (shard-major: ids[s][q][per_shard], dists likewise,
 * counts[s][q]) -- e.g. the buffer an RCCL all-gather produced -- sort by HybridScore
 * descending (= distance ascending for weight > 0) and truncate to `limit`.  The reference's
 * sort is unstable on arrival-ordered input; ties here break by (shard, id) ascending.
 * out_shards (optional) receives the originating shard.  Buffers follow `mem`. */
int sdb_topk_merge(uint32_t n_shards, uint64_t nq, uint32_t per_shard, const uint64_t *ids,
                   const float *dists, const uint32_t *counts, uint32_t limit, uint64_t *out_ids,
                   float *out_dists, uint32_t *out_shards, uint32_t *out_counts, int mem, int device,
                   void *stream);
/* per-shard limit rule, cluster/actions.go:291-299 */
int sdb_shard_limit(uint32_t limit, uint32_t n_shards, uint32_t max_search_limit, uint32_t *out);

/* ---------------------------------------------------------------------------------------------
 * cluster fan-out exchange  (C1's device half: RCCL over xGMI, one shard per GPU)
 * ------------------------------------------------------------------------------------------- */
/* ClusterNode.SearchPoints (cluster/actions.go:275-379) scatters a request to every shard over
 * msgpack net/rpc (:316-351), gathers the per-shard results and sorts / truncates them (:357-376).
 * With the shards of one node living one per MI355X the gather step is ONE RCCL all-gather of the
 * fixed-size per-shard result blocks, issued by this library on a stream of its own, followed by
 * the device merge above.  A cluster handle = one rank (one shard, one GPU) of that exchange.
 *
 * Bootstrap is the NCCL one: some rank calls sdb_cluster_unique_id and hands the 128 bytes to the
 * others over whatever channel the host already has (SemaDB: its cluster RPC; the tests:
 * torch.distributed / a pipe); every rank then calls sdb_cluster_create, which blocks until all
 * `world` ranks have arrived.  One process per GPU or one thread per GPU in one process (the shape
 * of a Go server that owns the whole node) -- sdb_cluster_create_local sets the latter up in one call.
 * Transports: distinct devices exchange through RCCL (ncclAllGather over xGMI).  When create_local is given the
 * SAME device for every rank -- several shards of a collection living on one GPU, which is also how a one-GPU box
 * runs the whole N-shard protocol -- the ranks exchange through device memory copies behind a host rendezvous
 * (the last rank to arrive enqueues the copies and merges for all); everything else (tickets, tags, error
 * propagation, merge) is the same code.  A mix of repeated and distinct devices is rejected. */
#define SDB_CLUSTER_ID_BYTES 128
typedef struct sdb_cluster sdb_cluster;
int sdb_cluster_unique_id(uint8_t *id /* [SDB_CLUSTER_ID_BYTES] */);
int sdb_cluster_create(int rank, int world, const uint8_t *id, int device, sdb_cluster **out);
/* all `n` ranks of a single-process node at once: rank r lives on devices[r] (NULL: device r) */
int sdb_cluster_create_local(int n, const int *devices, sdb_cluster **out /* [n] */);
int sdb_cluster_destroy(sdb_cluster *c);
int sdb_cluster_info(const sdb_cluster *c, int *rank, int *world, int *device);

/* Layout of one shard's result block for nq queries x per_shard results, the all-gather message:
 *   [0, off_dists)            uint64 ids   [nq][per_shard]
 *   [off_dists, off_counts)   float  dists [nq][per_shard]
 *   [off_counts, off_tag)     uint32 counts[nq]            (padded to 16 bytes)
 *   [off_tag, bytes)          sdb_block_tag, SDB_BLOCK_TAG_BYTES: written by the library, compared after the gather
 * sdb_index_search_batch can write straight into it (out_ids = block, out_dists = block + off_dists,
 * out_counts = block + off_counts). */
#define SDB_BLOCK_TAG_BYTES 64
typedef struct {
  uint32_t magic;      /* SDB_BLOCK_MAGIC                                                                   */
  uint32_t status;     /* sdb_status of this rank's shard search: != 0 fails the request on every rank      */
  uint64_t seq;        /* position of the call in this rank's sequence of collectives (ticket order)         */
  uint64_t ticket;     /* the caller's request ticket (0: none)                                              */
  uint64_t nq;         /* the shape every rank must agree on                                                  */
  uint32_t per_shard;
  uint32_t limit;
  uint64_t query_hash; /* order-independent 64-bit hash of the nq x dim query floats (search_batch; else 0)  */
  uint32_t rank;
  uint32_t reserved[3];
} sdb_block_tag;
#define SDB_BLOCK_MAGIC 0x53444254u /* "SDBT" */
int sdb_cluster_block_layout(uint64_t nq, uint32_t per_shard, size_t *off_dists, size_t *off_counts, size_t *off_tag,
                             size_t *bytes);

/* Collective calls, order and failure.  The reference fans one request out to its shards from one goroutine each,
 * and requests run concurrently (cluster/actions.go:316-351): shard A may see request 1 before request 2 and shard B
 * the other way round, which is harmless over RPC (every reply names its request) and fatal for a collective -- equal
 * sized all-gathers would pair request 1's block of one rank with request 2's of another and merge them into
 * plausible, wrong answers.  Three guards:
 *   (1) `ticket`: a request number the fan-out draws once per request (1, 2, 3, ... without gaps, shared by all ranks
 *       of the node) and passes to every rank's call.  A rank's calls then ENTER the exchange in ticket order whatever
 *       order its threads arrive in: a call whose ticket is not next waits for its predecessors.  ticket 0 = no
 *       ordering by the library (a single caller thread per rank that issues in the same order everywhere).
 *   (2) every block carries a tag (above): sequence number, ticket, shape and -- for search_batch -- a hash of the
 *       queries the rank searched.  After the gather every rank compares all tags; on any difference NO answer is
 *       produced (counts 0) and the call fails with SDB_ERR_STATE on every rank, naming the ranks that disagree.
 *   (3) a rank whose own shard search fails (index unusable after a failed write, no start node, out of memory)
 *       still enters the exchange with status != 0 in its tag and empty counts, so that its peers are not left
 *       waiting inside the all-gather; every rank then returns that error for this request and the next request is
 *       served normally -- the reference returns an error for the request and keeps serving (actions.go:339-353).
 * Device-memory calls are asynchronous: their verdict is delivered by sdb_cluster_synchronize (first failure since the
 * last call, SDB_ERR_STATE), and a failed request's out_counts are all zero. */

/* The exchange step of SearchPoints (actions.go:316-376) for a block this rank's shard produced:
 * all-gather of `block` (device memory, layout above, written by work already enqueued on `stream`; the library
 * writes its tag) over all ranks, then the merge (sdb_topk_merge rule) to the original `limit` on this rank's GPU.
 * Collective: every rank calls it with the same ticket / nq / per_shard / limit.
 * The exchange runs on the cluster's own stream, ordered after `stream`'s work at the time of the
 * call, so the caller's next search overlaps it.  `block` must stay untouched and the outputs are
 * not valid until sdb_cluster_wait(c, stream) (a stream-side wait, no host block) or
 * sdb_cluster_synchronize(c).  out_* follow `mem`; SDB_MEM_HOST outputs are copied back and the
 * call synchronises.  out_shards (optional) = rank of the shard each result came from. */
int sdb_cluster_allgather_merge(sdb_cluster *c, uint64_t ticket, uint64_t nq, uint32_t per_shard, void *block,
                                uint32_t limit, uint64_t *out_ids, float *out_dists, uint32_t *out_shards,
                                uint32_t *out_counts, int mem, void *stream);

/* ClusterNode.SearchPoints for this rank's shard, whole: per-shard limit (actions.go:291-299;
 * the query's own Limit / SearchSize are not rewritten per shard, :301-314) -> IndexVamana.Search
 * of all nq queries on `ix` into a block of the cluster's ring -> all-gather -> merge to `limit`.
 * Collective like sdb_cluster_allgather_merge.  `queries` and out_* follow `mem`; with
 * SDB_MEM_DEVICE the search is enqueued on `stream`, the exchange on the cluster's stream, and up
 * to SDB_CLUSTER_RING batches may be in flight before the caller waits (sdb_cluster_wait / _synchronize).
 * SDB_MEM_HOST calls block until their own answer is there, and up to SDB_CLUSTER_RING of them may be in flight
 * on one rank from different threads (each has its own staging; the exchange of one runs under the walk of the
 * next).  Every rank ends up with the same merged answer: a single-process fan-out that needs it once passes NULL
 * for all four out_* (SDB_MEM_HOST only) on the other ranks, which then take part in the exchange -- and return its
 * verdict -- without copying their copy of the answer back. */
#define SDB_CLUSTER_RING 8
int sdb_cluster_search_batch(sdb_cluster *c, sdb_index *ix, uint64_t ticket, uint64_t nq, const float *queries,
                             uint32_t limit, uint32_t search_size, uint64_t *out_ids,
                             float *out_dists, uint32_t *out_shards, uint32_t *out_counts, int mem,
                             void *stream);
/* make `stream` wait (on the device) for every exchange enqueued so far / block the host for them.
 * synchronize also reports: SDB_ERR_STATE (with the ranks and fields that disagreed, or the failing shard's status)
 * if any device-memory exchange since the previous synchronize failed its tag check. */
int sdb_cluster_wait(sdb_cluster *c, void *stream);
int sdb_cluster_synchronize(sdb_cluster *c);
/* the next ticket this rank will let into the exchange (tickets below it have entered) */
int sdb_cluster_next_ticket(const sdb_cluster *c, uint64_t *ticket);

/* A way out of the turnstile.  The reference fails ONE request and serves the next (cluster/actions.go:339-353);
 * a ticket that is drawn and never presented to some rank (the fan-out's goroutine panicked, the client went away
 * between two ranks' calls) must not wedge every later request of that rank.
 *
 * sdb_cluster_set_deadline: the longest a call waits (a) at the turnstile for its predecessors' tickets, (b) for the
 * other ranks to join its exchange.  Default 30 000 ms; 0 = for ever.  (a) expiring fails the call with
 * SDB_ERR_STATE having done NOTHING on this rank: the same call (same ticket) may be presented again, or its ticket
 * skipped.  (b) expiring fails it with SDB_ERR_STATE; on the shared-device transport the request is withdrawn and the
 * handle stays in step when it was this rank's latest, otherwise -- and always on RCCL, where the all-gather already
 * sits on the stream -- the handle is out of step for good (every later call SDB_ERR_STATE) and must be recreated;
 * sdb_cluster_destroy then aborts the communicator instead of draining it.
 *
 * sdb_cluster_skip_ticket: the fan-out declares that `ticket` will never be presented to this rank.
 *   nq == 0: no rank has entered the request's exchange or will (the request died before any call): the turnstile
 *            passes over the ticket, now or when its turn comes.  Nothing is exchanged; call it on every rank.
 *   nq  > 0: other ranks may already be inside the request's exchange: this rank enters it with an empty answer and
 *            SDB_ERR_STATE in its tag (guard 3), so those ranks fail THAT request and serve the next.  nq / limit (and
 *            per_shard for an sdb_cluster_allgather_merge request; 0 = the search_batch rule) must be the request's.
 *            Blocks until the exchange has run on this rank (at most the deadline). */
int sdb_cluster_set_deadline(sdb_cluster *c, uint32_t milliseconds);
int sdb_cluster_skip_ticket(sdb_cluster *c, uint64_t ticket, uint64_t nq, uint32_t per_shard, uint32_t limit);
/* one line for logs and measurement records: the transport that carries this rank's gathers -- for RCCL the library
 * version, the path of the librccl the process has loaded, the communicator's size and this rank in it */
int sdb_cluster_transport(const sdb_cluster *c, char *buf, size_t cap);

/* The tag check + merge on a gathered buffer [world][bytes] that some other transport delivered (the tests' gloo
 * path; a host that moves the blocks over its own RPC): sdb_cluster_stamp_block writes this rank's tag into its
 * block before it is sent (query_hash over `queries` when given; device memory, asynchronous on `stream`),
 * sdb_cluster_merge_gathered compares the tags of all blocks and merges, blocking; on a tag mismatch or a failed
 * shard it returns SDB_ERR_STATE and leaves all counts zero.  Buffers are device memory. */
int sdb_cluster_stamp_block(void *block, uint64_t nq, uint32_t per_shard, uint32_t limit, uint32_t rank, uint64_t seq,
                            uint64_t ticket, uint32_t status, const float *queries, uint32_t dim, int device,
                            void *stream);
int sdb_cluster_merge_gathered(uint32_t world, uint64_t nq, uint32_t per_shard, const void *gathered, uint32_t limit,
                               uint64_t *out_ids, float *out_dists, uint32_t *out_shards, uint32_t *out_counts,
                               int device, void *stream);

/* ---------------------------------------------------------------------------------------------
 * utils/kmeans.go + shard/vectorstore/product.go  (K5..K8)
 * ------------------------------------------------------------------------------------------- */
/* utils.KMeans.Fit (utils/kmeans.go:34-150) on X[n][stride] sub-vectors [offset, offset+len).
 * first_idx replaces rand.IntN (kmeans.go:61).  alias != 0 keeps the reference's behaviour that
 * centroids are views into X and the mean update overwrites those rows (kmeans.go:63,82,144);
 * X is then modified in place.  centroids_out[K*len], labels_out[n].  Buffers follow `mem`. */
int sdb_kmeans_fit(float *X, uint32_t n, uint32_t stride, uint32_t offset, uint32_t len, uint32_t K,
                   uint32_t max_iter, uint32_t first_idx, int alias, float *centroids_out,
                   uint8_t *labels_out, uint32_t *iters_out, int mem, int device, void *stream);

/* newProductQuantizer (product.go:42-88): cosine is replaced by euclidean (:52-61). */
int sdb_pq_create(uint32_t dim, uint32_t metric, uint32_t num_subvectors, uint32_t num_centroids,
                  int device, sdb_pq **out);
int sdb_pq_destroy(sdb_pq *pq);
/* productQuantizer.Fit (product.go:175-236): k-means per sub-vector over X[n][dim] in the given
 * row order, then the centroid-pair table (:225-230).  first_idx[M].  codes_out[n*M] (optional). */
int sdb_pq_fit(sdb_pq *pq, float *X, uint32_t n, const uint32_t *first_idx, int alias,
               uint8_t *codes_out, int mem, void *stream);
/* install a codebook read from the bucket (product.go:79-86): flat_centroids[M][K][subLen] */
int sdb_pq_set_codebook(sdb_pq *pq, const float *flat_centroids, int mem);
int sdb_pq_get_codebook(const sdb_pq *pq, float *flat_centroids, float *centroid_dists);
/* productQuantizer.encode (product.go:136-159) for n vectors: codes[n*M] */
int sdb_pq_encode(const sdb_pq *pq, const float *vectors, uint64_t n, uint8_t *codes, int mem,
                  void *stream);
/* asymmetric distance (product.go:238-277): per query a LUT of M*K sub-distances (:255-263), then
 * out[q*nc + c] = sum_i lut[i][codes[c][i]] in index order (:271-275).  codes[nc*M]. */
int sdb_pq_lut_distance(const sdb_pq *pq, const float *queries, uint64_t nq, const uint8_t *codes,
                        uint64_t nc, float *out, int mem, void *stream);
/* symmetric distance via the centroid-pair table (product.go:279-305): out[i] for pairs
 * (codes_x[i], codes_y[i]) */
int sdb_pq_sym_distance(const sdb_pq *pq, const uint8_t *codes_x, const uint8_t *codes_y, uint64_t n,
                        float *out, int mem, void *stream);
/* switch an index to the fitted quantizer: its searches then use the LUT distance and its
 * prunes the symmetric table, exactly as a fitted productQuantizer store does.  The index
 * encodes every stored vector (product.go:161-169 Set -> encode). */
int sdb_index_attach_pq(sdb_index *ix, const sdb_pq *pq, void *stream);
/* insert.go:47-58 for a node and SEVERAL candidates at once: candidateSet.Add(neighbours...),
 * Add(extra...), Sort, robustPrune(node).  The device build applies this rule to a target that has
 * several back-edge requests in one round.  chip_wide = 0: one wavefront; 1: the sequence of
 * chip-wide kernels the build uses for hub nodes.  Both give the oracle's row
 * (tests/test_gpu_build.py).  Not available for a start node that holds an overflow list. */
int sdb_index_union_prune(sdb_index *ix, uint64_t id, uint64_t m, const uint64_t *extra_ids,
                          int chip_wide, void *stream);
/* Centroid ids that do not come from encode(): the k-means labels productQuantizer.Fit leaves on
 * its training points (product.go:216-218) and the codes a bucket holds under NodeKey(id,'q')
 * (productQuantizedPoint.ReadFrom / WriteTo, product.go:349-383).  ids [n] u64, codes [n][M] u8,
 * host memory, on an index with an attached quantizer.  Not inside a write transaction (SDB_ERR_STATE): code rows exist
 * once, not per graph version, and could not be rolled back with it. */
int sdb_index_set_codes(sdb_index *ix, uint64_t n, const uint64_t *ids, const uint8_t *codes);
int sdb_index_get_codes(const sdb_index *ix, uint64_t n, const uint64_t *ids, uint8_t *codes);

#ifdef __cplusplus
}
#endif
#endif /* SEMADB_AMD_H */
