/*
 * sdb_oracle.h -- CPU ORACLE for the SemaDB Vamana hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm (Semafind/semadb, Go + Plan 9
 * AVX2 assembly).  It exists so that the HIP path can be checked against it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under
 * semadb_amd/ (the product) links, imports or calls it.
 *
 * Parity status: RESTATEMENT-PINNED.  The reference cannot be compiled here (no Go toolchain),
 * so the oracle is pinned against (a) every known-answer test the reference holds for this
 * path (tests/test_oracle_kat.py lists them with file:line) and (b) a three-way agreement of
 * the scalar lane model, an AVX2 intrinsics transcription of distance/asm/{dot,euclidean}.s
 * and the HIP kernels.  The reference has no golden vectors for random high-d inputs (all of
 * its large tests use an unseeded RNG).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * repository root).
 */
#ifndef SDB_ORACLE_H
#define SDB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* distance metric names: models consts used by distance/distance.go:70-83 */
#define ORC_METRIC_EUCLIDEAN 0
#define ORC_METRIC_COSINE 1
#define ORC_METRIC_DOT 2

/* which arithmetic the raw dot / squared-L2 uses */
#define ORC_IMPL_ASM 0  /* scalar lane model of distance/asm/dot.s, euclidean.s (AVX2+FMA path) */
#define ORC_IMPL_AVX2 1 /* AVX2/FMA intrinsics transcription of the same two files (fast)   */
#define ORC_IMPL_PURE 2 /* distance/puredist.go:3-18 (sequential, unfused)                   */

/* vamana.go:28 */
#define ORC_STARTID 1ull

/* ---- raw distances --------------------------------------------------------------- */
float orc_dot(const float *x, const float *y, size_t n, int impl);
float orc_sqeuclid(const float *x, const float *y, size_t n, int impl);
/* distance.go:19-25,70-83: euclidean -> sq-L2, dot -> -dot, cosine -> 1 - dot */
float orc_distance(const float *x, const float *y, size_t n, int metric, int impl);
/* nq x nc matrix of distances, row-major out[q*nc + c] */
void orc_distance_matrix(const float *q, size_t nq, const float *c, size_t nc, size_t d, int metric,
                         int impl, float *out);

/* ---- DistSet known-answer harness (distset.go:133-238) --------------------------- */
/* Runs a script on a DistSet whose distFn is the table lookup dists[id] (distset_test.go:19-24).
 * ops[i] in {0 = Add, 1 = AddWithLimit, 2 = Sort}; for ops 0/1 the ids are
 * args[arg_off[i] .. arg_off[i+1]).  use_bitset selects VisitedBitSet vs VisitedMap
 * (distset.go:140-155).  Writes the final item ids to out_ids and returns their count. */
int orc_distset_script(int capacity, int use_bitset, const float *dists, int n_dists, const int *ops,
                       int n_ops, const uint64_t *args, const int *arg_off, uint64_t *out_ids,
                       int out_cap);

/* ---- Vamana index ----------------------------------------------------------------- */
typedef struct orc_index orc_index;

/* vamana.go:54-81; params models/index.go:275-313.  Ranges are NOT enforced here (the
 * tests use out-of-range values the way vamana_test.go does); the product boundary checks
 * them.  `impl` selects the distance arithmetic. */
orc_index *orc_index_new(int dim, int metric, int impl, int degree_bound, int search_size, float alpha);
void orc_index_free(orc_index *ix);

/* setupStartNode vamana.go:93-120: the reference draws an unseeded random unit vector; the
 * oracle takes it from the caller so runs are reproducible.  Must be called before insert. */
int orc_index_set_start(orc_index *ix, const float *vec);

/* insertSinglePoint insert.go:16-68, applied sequentially (the reference runs NumCPU-1 of
 * them concurrently, vamana.go:190-196, so its graph is non-deterministic; the oracle fixes
 * the order to the call order).  ids 0 and 1 are rejected (vamana.go:150-157). returns 0 ok. */
int orc_index_insert(orc_index *ix, uint64_t id, const float *vec);

/* IndexVamana.InsertUpdateDelete, delete branch (vamana.go:175-233): removeInboundEdges (prune.go:88-154) over
 * EdgeScan (node.go:142-199) and pruneDeleteNeighbour (prune.go:12-84), stragglers re-attached to the start
 * node, then the nodes are dropped.  Unknown ids are skipped (vamana.go:161-163); ids 0/1 are errors.  An
 * update (vamana.go:170-174,247-251) is this followed by orc_index_insert with the same id. */
int orc_index_delete(orc_index *ix, const uint64_t *ids, uint64_t n);
/* insert.go:47-58 over a node's neighbours + several new candidates at once (Add, Sort, robustPrune) */
int orc_index_union_prune(orc_index *ix, uint64_t id, const uint64_t *extra, uint64_t m);
/* one round of the device's batched build schedule (see the .c file) */
int orc_index_insert_round(orc_index *ix, const uint64_t *ids, const float *vecs, int rs, int group_cap, int big_min);

/* Bulk load of an existing graph (what ItemCache would read from the bucket, node.go:96-111,
 * plain.go:125-141).  edges hold node ids; ids missing from `ids` are silently dropped the way
 * ItemCache.GetMany skips them (itemcache.go:109-128).  The start node (id 1) must be among
 * the ids.  offsets has n+1 entries. */
int orc_index_load(orc_index *ix, uint64_t n, const uint64_t *ids, const float *vectors,
                   const uint64_t *offsets, const uint64_t *edges);

uint64_t orc_index_size(const orc_index *ix); /* number of nodes incl. start node */
uint64_t orc_index_num_edges(const orc_index *ix);
/* export in load order: ids[n], vectors[n*dim] (may be NULL), offsets[n+1], edges[num_edges] (ids) */
int orc_index_export(const orc_index *ix, uint64_t *ids, float *vectors, uint64_t *offsets,
                     uint64_t *edges);

typedef struct {
  uint64_t n_dist;     /* distFn evaluations (distset.go:179 + the start node)            */
  uint64_t n_hop;      /* expanded nodes = len(visitedSet) (search.go:73)                 */
  uint64_t n_edges;    /* sum of deg(v) over expanded nodes                               */
  uint64_t n_visit_written; /* entries written to visit_ids                               */
} orc_trace;

/* IndexVamana.Search vamana.go:278-310 on top of greedySearch search.go:9-102.
 * filter_ids: NULL for no filter, else an ascending id list (roaring iteration order).
 * out_ids/out_dists need room for `limit`; *out_count receives the number written.
 * visit_ids (optional, visit_cap entries): expansion order (search.go:73, before Sort).
 * returns 0, or -1 for searchSize < k (search.go:23-25). */
int orc_index_search(const orc_index *ix, const float *query, int limit, int search_size,
                     const uint64_t *filter_ids, int n_filter, uint64_t *out_ids, float *out_dists,
                     int *out_count, uint64_t *visit_ids, int visit_cap, orc_trace *trace);

/* nq searches, OpenMP-parallel over queries (one query per thread: mirrors goroutine per
 * request).  Outputs are [nq][limit]; counts[nq]; traces[nq] optional. n_threads<=0: all. */
int orc_index_search_batch(const orc_index *ix, const float *queries, int nq, int limit,
                           int search_size, uint64_t *out_ids, float *out_dists, int *out_counts,
                           orc_trace *traces, int n_threads);

/* greedySearch's second return value: the visited set after Sort (search.go:100), i.e. what
 * insertSinglePoint feeds to robustPrune.  Returns count written. */
int orc_index_visited_sorted(const orc_index *ix, const float *query, int search_size,
                             uint64_t *out_ids, float *out_dists, int cap);

/* ---- k-means (utils/kmeans.go:34-150) ---------------------------------------------- */
/* X: n rows, row stride `stride` floats; clusters X[i][offset:offset+len].  first_idx stands in
 * for rand.IntN (kmeans.go:61).  alias != 0 reproduces the reference's aliasing: Centroids[i]
 * are sub-slices of X rows and the mean update overwrites those rows (kmeans.go:63,82,144);
 * alias == 0 works on copies.  centroids_out: K*len; labels: n; *iters_out: Lloyd iterations
 * that ran an assignment pass. */
int orc_kmeans_fit(float *X, int n, int stride, int offset, int len, int K, int max_iter,
                   int first_idx, int alias, int impl, float *centroids_out, uint8_t *labels,
                   int *iters_out);

/* ---- product quantizer (shard/vectorstore/product.go) ------------------------------- */
typedef struct orc_pq orc_pq;
/* newProductQuantizer product.go:42-88: cosine is swapped for euclidean (:52-61). */
orc_pq *orc_pq_new(int dim, int metric, int impl, int num_subvectors, int num_centroids);
void orc_pq_free(orc_pq *pq);
/* Fit product.go:175-236 over n row-major vectors (rows are visited in the given order; the
 * reference's order is Go-map random, itemcache.go:221).  first_idx[M] seeds each sub-quantizer's
 * k-means.  X is modified when alias != 0.  codes_out: n*M labels. */
int orc_pq_fit(orc_pq *pq, float *X, int n, const int *first_idx, int alias, uint8_t *codes_out);
int orc_pq_set_codebook(orc_pq *pq, const float *flat_centroids); /* also fills centroidDists :225-230 */
const float *orc_pq_flat_centroids(const orc_pq *pq);             /* [M][K][subLen]  */
const float *orc_pq_centroid_dists(const orc_pq *pq);             /* [M][K][K]       */
void orc_pq_encode(const orc_pq *pq, const float *vec, uint8_t *codes);      /* product.go:136-159 */
void orc_pq_lut(const orc_pq *pq, const float *query, float *lut);           /* product.go:255-263 */
float orc_pq_dist_lut(const orc_pq *pq, const float *lut, const uint8_t *codes);   /* :271-275 */
float orc_pq_dist_sym(const orc_pq *pq, const uint8_t *cx, const uint8_t *cy);     /* :300-302 */

/* Attach a fitted quantizer + codes to an index: greedySearch then uses the LUT distance
 * (product.go:238-277) and robustPrune the symmetric table (product.go:279-305).
 * codes: one row of M bytes per node in load/insert order (start node included). */
int orc_index_attach_pq(orc_index *ix, const orc_pq *pq, const uint8_t *codes);

/* ---- cluster fan-out merge (cluster/actions.go:291-376) ----------------------------- */
/* per-shard limit, actions.go:291-299 */
int orc_shard_limit(int limit, int n_shards, int max_search_limit);
/* Merge n_shards result lists (dists ascending per shard) for one query: sort by
 * HybridScore = -dist * weight descending (actions.go:357-364), truncate to limit (:372-374).
 * The reference sort is unstable on arrival-ordered input; the oracle uses the total order
 * (score desc, shard asc, id asc).  Returns count written. */
int orc_cluster_merge(int n_shards, const int *counts, const uint64_t *ids, const float *dists,
                      int per_shard_cap, float weight, int limit, uint64_t *out_ids,
                      float *out_dists, int *out_shards);

int orc_has_avx2(void);
int orc_max_threads(void);
void orc_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
