// pq.h -- product quantizer state shared between pq.hip and the search path (internal).
#pragma once
#include "dist_core.h"
#include "index.h"

// productQuantizer (shard/vectorstore/product.go:28-40) with its tables pinned in HBM
struct sdb_pq {
  uint32_t dim = 0, M = 0, K = 0, sub_len = 0;
  int metric = 0;  // after the cosine -> euclidean swap (product.go:52-61)
  int device = 0;
  float *d_centroids = nullptr;  // flatCentroids [M][K][sub_len]  (product.go:37)
  float *d_centroids_t = nullptr;  // the same, element-major [M][sub_len][K]: the table kernel's coalesced row loads
  float *d_cdists = nullptr;     // centroidDists [M][K][K]        (product.go:36)
  bool fitted = false;
};

namespace sdb {
// lut[q][i][j] = distFn(q_sub_i, centroid_ij) for nq device-resident queries (product.go:255-263)
int pq_build_lut(const sdb_pq *pq, const float *d_queries, uint64_t nq, float *d_lut, hipStream_t stream);
// codes[v][i] = argmin_j distFn(v_sub_i, centroid_ij) (product.go:136-159); device buffers
int pq_encode_device(const sdb_pq *pq, const float *d_vecs, uint64_t n, uint8_t *d_codes, hipStream_t stream);
// KMeans.Fit (utils/kmeans.go:34-150) for M problems at once: problem m = columns [offset0 + m * len, + len) of dX's rows
int kmeans_device(float *dX, uint32_t n, uint32_t stride, uint32_t offset0, uint32_t len, uint32_t M, uint32_t K,
                  uint32_t max_iter, const uint32_t *h_first_idx, int alias, float *d_centroids_out, uint8_t *d_labels_out,
                  uint32_t labels_stride, uint32_t *h_iters_out, hipStream_t stream);
// original-layout copy of slab rows [first, first+n) into dst (device)
int unpermute_rows_public(const sdb_index *ix, uint32_t first, uint32_t n, float *dst, hipStream_t stream);
}  // namespace sdb
