// pq.hip -- K5..K8: k-means (utils/kmeans.go:34-150) and the product quantizer
// (shard/vectorstore/product.go) as batched assignment / LUT-distance kernels.
//
// Every sub-vector distance is the reference's distFn on a sub-slice, i.e. asm.Dot /
// asm.SquaredEuclideanDistance with n = subLen (32 partial sums over whole 32-float blocks, a
// sequential scalar chain for the rest -- for subLen < 32 the chain alone).  dist_serial() replays
// that arithmetic inside ONE thread, so argmin ties, labels, codes and LUT entries are bit-identical
// to the reference; the parallelism is across (vector, sub-quantizer, centroid), which is where it is.
#include <cfloat>

#include "pq.h"

namespace sdb {

// asm.Dot / asm.SquaredEuclideanDistance (dot.s:7-55 / euclidean.s:7-65) in one thread.
template <bool L2>
__device__ __forceinline__ float dist_serial(const float *__restrict__ x, const float *__restrict__ y, uint32_t n) {
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; i++) acc[i] = 0.0f;
  const uint32_t nblk = n / 32;
  for (uint32_t b = 0; b < nblk; b++) {
#pragma unroll
    for (int L = 0; L < 32; L++) acc[L] = chain1<L2>(acc[L], x[32 * b + L], y[32 * b + L]);
  }
  float t = 0.0f;
  for (uint32_t i = nblk * 32; i < n; i++) t = chain1<L2>(t, x[i], y[i]);
  float r[4];
#pragma unroll
  for (int l = 0; l < 4; l++) {
    const float s0 = ((acc[l] + acc[8 + l]) + acc[16 + l]) + acc[24 + l];
    const float s1 = ((acc[l + 4] + acc[12 + l]) + acc[20 + l]) + acc[28 + l];
    r[l] = s0 + s1;
  }
  r[0] = r[0] + t;  // VADDPS X0, X4, X0 with X4 = {t, 0, 0, 0}
  r[1] = r[1] + 0.0f;
  r[2] = r[2] + 0.0f;
  r[3] = r[3] + 0.0f;
  return (r[0] + r[1]) + (r[2] + r[3]);
}

__device__ __forceinline__ float dist_serial_metric(const float *x, const float *y, uint32_t n, int metric) {
  if (metric == SDB_METRIC_EUCLIDEAN) return dist_serial<true>(x, y, n);
  return metric_finish(dist_serial<false>(x, y, n), metric);
}

// The same arithmetic with one operand's NB whole 32-float blocks already in the thread's registers
// (r) and the other operand (u) at a wave-uniform address, so its elements arrive through the scalar
// cache and cost no vector memory instruction.  The 32 partial sums are produced in the order the
// reduce tree consumes them, so only a handful are live at a time.  rt = the register operand's tail
// elements in memory (sub_len % 32 of them), chained sequentially as the asm does.
constexpr int kRegBlocksMax = 8;  // sub-vectors of up to 8 * 32 + 31 floats take the register path
typedef float pq_f2v __attribute__((ext_vector_type(2)));
template <bool L2, int NB>
__device__ __forceinline__ float dist_regs(const float *r, const float *__restrict__ u, const float *__restrict__ rt,
                                           uint32_t tail) {
  // partial sums 2p and 2p + 1 advance together: one v_pk_fma_f32 (and, for euclidean, one packed subtract, rounded per
  // element like VSUBPS) per pair of elements -- each half is the reference's own operation on its own chain
  pq_f2v acc2[16];
#pragma unroll
  for (int p = 0; p < 16; p++) acc2[p] = pq_f2v{0.0f, 0.0f};
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int p = 0; p < 16; p++) {
      const pq_f2v x = {r[32 * b + 2 * p], r[32 * b + 2 * p + 1]}, y = {u[32 * b + 2 * p], u[32 * b + 2 * p + 1]};
      if constexpr (L2) {
        const pq_f2v d = x - y;
        acc2[p] = __builtin_elementwise_fma(d, d, acc2[p]);
      } else {
        acc2[p] = __builtin_elementwise_fma(x, y, acc2[p]);
      }
    }
  float rr[4];
#pragma unroll
  for (int l = 0; l < 4; l++) {
    float s[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      float part[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int L = l + 4 * h + 8 * k;
        part[k] = acc2[L >> 1][L & 1];
      }
      s[h] = ((part[0] + part[1]) + part[2]) + part[3];
    }
    rr[l] = s[0] + s[1];
  }
  // the tail chain (dot.s:35-43): rt holds the bound vector's tail elements in REGISTERS (31 slots, statically
  // indexed: a sub-vector shorter than 32 floats -- M = 32 or 192 at d = 768 -- is nothing but this chain, and
  // fetching its elements from memory once per centroid made the encode 40x slower than at M = 8)
  float t = 0.0f;
#pragma unroll
  for (int i = 0; i < 31; i++)
    if ((uint32_t)i < tail) t = chain1<L2>(t, rt[i], u[NB * 32 + i]);
  rr[0] = rr[0] + t;
  rr[1] = rr[1] + 0.0f;
  rr[2] = rr[2] + 0.0f;
  rr[3] = rr[3] + 0.0f;
  return (rr[0] + rr[1]) + (rr[2] + rr[3]);
}

// ---------------------------------------------------------------------------------------------
// k-means.  Centroid j is a VIEW: row cent_row[j] of cent_base (stride cent_stride) at cent_off --
// either into X itself (the reference's aliasing, kmeans.go:63,82,144) or into a private copy.
// ---------------------------------------------------------------------------------------------
struct KmArgs {
  float *X;
  uint32_t n, stride, offset, len, K;
  float *cent_base;
  uint32_t cent_stride, cent_off;
  uint32_t *cent_row;  // [K]
  float *min_dist;     // [n]
  uint8_t *labels;     // [n]
  uint32_t *change;    // [1]
  float *sums;         // [K][len]
  uint32_t *counts;    // [K]
  uint32_t first_idx;
};

__device__ __forceinline__ float *km_centroid(const KmArgs &a, uint32_t j) {
  return a.cent_base + (size_t)a.cent_row[j] * a.cent_stride + a.cent_off;
}

// one furthest-point step (kmeans.go:65-78): distance to centroid i-1, running minimum per point
__global__ void k_km_init_dist(const KmArgs a, uint32_t i) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.n || j == a.first_idx) return;  // alreadyCentroid only ever holds randId (:60-62,68)
  const float d = dist_serial<true>(a.X + (size_t)j * a.stride + a.offset, km_centroid(a, i - 1), a.len);
  if (d < a.min_dist[j]) a.min_dist[j] = d;
}

// furthestId (kmeans.go:66-80): strict '>' scanning ascending, so the lowest index among equal maxima
// wins and index 0 is returned when nothing is > 0.
__global__ __launch_bounds__(1024) void k_km_argmax(const KmArgs a, uint32_t i) {
  __shared__ float s_v[1024];
  __shared__ uint32_t s_i[1024];
  float best = 0.0f;
  uint32_t best_i = 0;
  for (uint32_t j = threadIdx.x; j < a.n; j += blockDim.x) {
    if (j == a.first_idx) continue;
    const float v = a.min_dist[j];
    if (v > best) best = v, best_i = j;
  }
  s_v[threadIdx.x] = best, s_i[threadIdx.x] = best_i;
  __syncthreads();
  for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float ov = s_v[threadIdx.x + s], mv = s_v[threadIdx.x];
      const uint32_t oi = s_i[threadIdx.x + s], mi = s_i[threadIdx.x];
      if (ov > mv || (ov == mv && ov > 0.0f && oi < mi)) s_v[threadIdx.x] = ov, s_i[threadIdx.x] = oi;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) a.cent_row[i] = s_v[0] > 0.0f ? s_i[0] : 0u;
}

// assignment (kmeans.go:100-115): argmin over centroids, strict '<' starting from centroid 0
__global__ void k_km_assign(const KmArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float *sv = a.X + (size_t)i * a.stride + a.offset;
  float best = dist_serial<true>(sv, km_centroid(a, 0), a.len);
  uint32_t best_id = 0;
  for (uint32_t j = 1; j < a.K; j++) {
    const float d = dist_serial<true>(sv, km_centroid(a, j), a.len);
    if (d < best) best = d, best_id = j;
  }
  if (a.labels[i] != (uint8_t)best_id) {
    a.labels[i] = (uint8_t)best_id;
    atomicAdd(a.change, 1u);
  }
}

// update sums (kmeans.go:125-137): per label, members added IN DATA ORDER (fp32 adds are not
// associative).  Thread (label, component) walks the points in order.
__global__ void k_km_sums(const KmArgs a) {
  const uint32_t lb = blockIdx.x;
  for (uint32_t j = threadIdx.x; j < a.len; j += blockDim.x) {
    float s = 0.0f;
    uint32_t c = 0;
    for (uint32_t i = 0; i < a.n; i++)
      if (a.labels[i] == lb) {
        s += a.X[(size_t)i * a.stride + a.offset + j];
        c++;
      }
    a.sums[(size_t)lb * a.len + j] = s;
    if (j == 0) a.counts[lb] = c;
  }
}

// means (kmeans.go:139-146) written through the centroid views in centroid order: with aliasing two
// centroids can share a row and the later one wins, exactly as in the reference.
__global__ void k_km_means(const KmArgs a) {
  for (uint32_t i = 0; i < a.K; i++) {
    const uint32_t c = a.counts[i];
    if (c != 0) {
      float *dst = km_centroid(a, i);
      for (uint32_t j = threadIdx.x; j < a.len; j += blockDim.x) dst[j] = a.sums[(size_t)i * a.len + j] / (float)c;
    }
    __syncthreads();
  }
}

__global__ void k_km_gather_centroids(const KmArgs a, float *out /* [K][len] */) {
  const uint32_t j = blockIdx.x;
  const float *c = km_centroid(a, j);
  for (uint32_t t = threadIdx.x; t < a.len; t += blockDim.x) out[(size_t)j * a.len + t] = c[t];
}

__global__ void k_fill_f32(float *p, float v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void k_iota_u32(uint32_t *p, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// KMeans.Fit on device buffers.  d_centroids_out [K][len], d_labels [n].
int kmeans_device(float *dX, uint32_t n, uint32_t stride, uint32_t offset, uint32_t len, uint32_t K,
                  uint32_t max_iter, uint32_t first_idx, int alias, float *d_centroids_out, uint8_t *d_labels,
                  uint32_t *iters_out, hipStream_t stream) {
  if (n == 0 || K == 0 || K > 256) return fail(SDB_ERR_INVALID, "kmeans: need n > 0 and 1 <= K <= 256");
  if (first_idx >= n) return fail(SDB_ERR_INVALID, "kmeans: first_idx out of range");
  if (len == 0 || offset + len > stride) return fail(SDB_ERR_INVALID, "kmeans: sub-vector out of range");
  char *buf = nullptr;
  const size_t b_min = ((size_t)n * 4 + 255) & ~(size_t)255, b_rows = 1024, b_sums = ((size_t)K * len * 4 + 255) & ~(size_t)255;
  const size_t b_priv = alias ? 0 : b_sums;
  SDB_HIP(hipMalloc(&buf, b_min + 3 * b_rows + b_sums + 256 + b_priv));
  struct Free {
    char *p;
    hipStream_t s;
    ~Free() {
      (void)hipStreamSynchronize(s);
      (void)hipFree(p);
    }
  } fr{buf, stream};
  KmArgs a{};
  a.X = dX, a.n = n, a.stride = stride, a.offset = offset, a.len = len, a.K = K, a.first_idx = first_idx;
  a.min_dist = (float *)buf;
  uint32_t *rows_x = (uint32_t *)(buf + b_min);
  uint32_t *rows_id = (uint32_t *)(buf + b_min + b_rows);
  a.counts = (uint32_t *)(buf + b_min + 2 * b_rows);
  a.sums = (float *)(buf + b_min + 3 * b_rows);
  a.change = (uint32_t *)(buf + b_min + 3 * b_rows + b_sums);
  float *priv = alias ? nullptr : (float *)(buf + b_min + 3 * b_rows + b_sums + 256);
  a.labels = d_labels;
  // ---- furthest-point initialisation (kmeans.go:56-83): centroids are views into X
  a.cent_base = dX, a.cent_stride = stride, a.cent_off = offset, a.cent_row = rows_x;
  hipLaunchKernelGGL(k_fill_f32, dim3((n + 255) / 256), dim3(256), 0, stream, a.min_dist, FLT_MAX, (size_t)n);
  SDB_HIP(hipMemcpyAsync(rows_x, &a.first_idx, 4, hipMemcpyHostToDevice, stream));
  for (uint32_t i = 1; i < K; i++) {
    hipLaunchKernelGGL(k_km_init_dist, dim3((n + 127) / 128), dim3(128), 0, stream, a, i);
    hipLaunchKernelGGL(k_km_argmax, dim3(1), dim3(1024), 0, stream, a, i);
  }
  SDB_HIP(hipGetLastError());
  if (!alias) {  // fenced mode: work on copies, X stays untouched
    hipLaunchKernelGGL(k_km_gather_centroids, dim3(K), dim3(64), 0, stream, a, priv);
    hipLaunchKernelGGL(k_iota_u32, dim3(1), dim3(256), 0, stream, rows_id, 256u);
    a.cent_base = priv, a.cent_stride = len, a.cent_off = 0, a.cent_row = rows_id;
  }
  SDB_HIP(hipMemsetAsync(d_labels, 0, n, stream));  // Labels start at 0 (kmeans.go:87)
  uint32_t iters = 0;
  for (uint32_t it = 0; it < max_iter; it++) {  // kmeans.go:96
    iters++;
    SDB_HIP(hipMemsetAsync(a.change, 0, 4, stream));
    hipLaunchKernelGGL(k_km_assign, dim3((n + 63) / 64), dim3(64), 0, stream, a);
    uint32_t change = 0;
    SDB_HIP(hipMemcpyAsync(&change, a.change, 4, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipStreamSynchronize(stream));
    if (change == 0) break;  // :116-118
    hipLaunchKernelGGL(k_km_sums, dim3(K), dim3(128), 0, stream, a);
    hipLaunchKernelGGL(k_km_means, dim3(1), dim3(256), 0, stream, a);
  }
  hipLaunchKernelGGL(k_km_gather_centroids, dim3(K), dim3(64), 0, stream, a, d_centroids_out);
  SDB_HIP(hipGetLastError());
  if (iters_out) *iters_out = iters;
  return SDB_OK;
}

// ---------------------------------------------------------------------------------------------
// product quantizer kernels
// ---------------------------------------------------------------------------------------------
// centroidDists[i][j][k] = distFn(centroid_ij, centroid_ik)   product.go:225-230
__global__ void k_pq_cdists(const float *__restrict__ cent, float *__restrict__ out, uint32_t M, uint32_t K,
                            uint32_t sub_len, int metric) {
  const uint32_t i = blockIdx.y, j = blockIdx.x;
  for (uint32_t k = threadIdx.x; k < K; k += blockDim.x)
    out[((size_t)i * K + j) * K + k] = dist_serial_metric(cent + ((size_t)i * K + j) * sub_len,
                                                           cent + ((size_t)i * K + k) * sub_len, sub_len, metric);
}

// encode (product.go:136-159): per sub-vector argmin over the K centroids, init MaxFloat32, strict '<'
__global__ void k_pq_encode(const float *__restrict__ vecs, uint64_t n, uint32_t dim, const float *__restrict__ cent,
                            uint32_t M, uint32_t K, uint32_t sub_len, int metric, uint8_t *__restrict__ codes) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = blockIdx.y;
  if (v >= n) return;
  const float *sub = vecs + v * dim + (size_t)i * sub_len;
  float best = FLT_MAX;
  uint32_t best_id = 0;
  for (uint32_t j = 0; j < K; j++) {
    const float d = dist_serial_metric(sub, cent + ((size_t)i * K + j) * sub_len, sub_len, metric);
    if (d < best) best = d, best_id = j;
  }
  codes[v * M + i] = (uint8_t)best_id;
}

// asymmetric table (product.go:255-263): lut[q][i][j] = distFn(q_sub_i, centroid_ij)
__global__ void k_pq_lut(const float *__restrict__ queries, uint32_t dim, const float *__restrict__ cent, uint32_t M,
                         uint32_t K, uint32_t sub_len, int metric, float *__restrict__ lut) {
  const uint32_t q = blockIdx.y;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * K) return;
  const uint32_t i = t / K, j = t % K;
  lut[(size_t)q * M * K + t] = dist_serial_metric(queries + (size_t)q * dim + (size_t)i * sub_len,
                                                  cent + ((size_t)i * K + j) * sub_len, sub_len, metric);
}

// Register-tiled forms of the two kernels above for sub_len < 32 * (kRegBlocksMax + 1).
// LUT: thread = centroid j (its row in registers), the block walks kLutQT queries whose sub-vectors
// are wave-uniform -- the centroid table is read once per 16 queries instead of once per query.
#ifndef SDB_LUT_QT
#define SDB_LUT_QT 16
#endif
constexpr uint32_t kLutQT = SDB_LUT_QT;
template <bool L2, int NB>
__global__ __launch_bounds__(256) void k_pq_lut_t(const float *__restrict__ queries, uint32_t nq, uint32_t dim,
                                                  const float *__restrict__ cent, uint32_t M, uint32_t K,
                                                  uint32_t sub_len, int metric, float *__restrict__ lut) {
  const uint32_t i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  const uint32_t q0 = blockIdx.z * kLutQT, q1 = min(q0 + kLutQT, nq);
  const float *row = cent + ((size_t)i * K + min(j, K - 1)) * sub_len;
  float r[NB > 0 ? NB * 32 : 1];
#pragma unroll
  for (int e = 0; e < NB * 32; e++) r[e] = row[e];
  const uint32_t tail = sub_len - NB * 32;
  float rt[31];
#pragma unroll
  for (int e = 0; e < 31; e++) rt[e] = (uint32_t)e < tail ? row[NB * 32 + e] : 0.0f;
  for (uint32_t q = q0; q < q1; q++) {
    const float *x = queries + (size_t)q * dim + (size_t)i * sub_len;
    float d = dist_regs<L2, NB>(r, x, rt, tail);
    if constexpr (!L2) d = metric_finish(d, metric);
    if (j < K) lut[((size_t)q * M + i) * K + j] = d;
  }
}

// encode: thread = vector (its sub-vector in registers), the K centroids are wave-uniform
template <bool L2, int NB>
__global__ __launch_bounds__(256) void k_pq_encode_t(const float *__restrict__ vecs, uint64_t n, uint32_t dim,
                                                     const float *__restrict__ cent, uint32_t M, uint32_t K,
                                                     uint32_t sub_len, int metric, uint8_t *__restrict__ codes) {
  const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t i = blockIdx.y;
  const float *sub = vecs + (v < n ? v : n - 1) * dim + (size_t)i * sub_len;
  float r[NB > 0 ? NB * 32 : 1];
#pragma unroll
  for (int e = 0; e < NB * 32; e++) r[e] = sub[e];
  const uint32_t tail = sub_len - NB * 32;
  float rt[31];
#pragma unroll
  for (int e = 0; e < 31; e++) rt[e] = (uint32_t)e < tail ? sub[NB * 32 + e] : 0.0f;
  float best = FLT_MAX;
  uint32_t best_id = 0;
  for (uint32_t j = 0; j < K; j++) {
    float d = dist_regs<L2, NB>(r, cent + ((size_t)i * K + j) * sub_len, rt, tail);
    if constexpr (!L2) d = metric_finish(d, metric);
    if (d < best) best = d, best_id = j;
  }
  if (v < n) codes[v * M + i] = (uint8_t)best_id;
}

template <int NB>
static void launch_lut_t(const sdb_pq *pq, const float *d_queries, uint64_t nq, float *d_lut, hipStream_t stream) {
  const dim3 grid((pq->K + 255) / 256, pq->M, (unsigned)((nq + kLutQT - 1) / kLutQT));
  if (pq->metric == SDB_METRIC_EUCLIDEAN)
    hipLaunchKernelGGL((k_pq_lut_t<true, NB>), grid, dim3(256), 0, stream, d_queries, (uint32_t)nq, pq->dim,
                       pq->d_centroids, pq->M, pq->K, pq->sub_len, pq->metric, d_lut);
  else
    hipLaunchKernelGGL((k_pq_lut_t<false, NB>), grid, dim3(256), 0, stream, d_queries, (uint32_t)nq, pq->dim,
                       pq->d_centroids, pq->M, pq->K, pq->sub_len, pq->metric, d_lut);
}

template <int NB>
static void launch_encode_t(const sdb_pq *pq, const float *d_vecs, uint64_t n, uint8_t *d_codes, hipStream_t stream) {
  const dim3 grid((unsigned)((n + 255) / 256), pq->M);
  if (pq->metric == SDB_METRIC_EUCLIDEAN)
    hipLaunchKernelGGL((k_pq_encode_t<true, NB>), grid, dim3(256), 0, stream, d_vecs, n, pq->dim, pq->d_centroids,
                       pq->M, pq->K, pq->sub_len, pq->metric, d_codes);
  else
    hipLaunchKernelGGL((k_pq_encode_t<false, NB>), grid, dim3(256), 0, stream, d_vecs, n, pq->dim, pq->d_centroids,
                       pq->M, pq->K, pq->sub_len, pq->metric, d_codes);
}

// The same table on the matrix cores, for dot / cosine sub-vectors of whole 32-float blocks and K a multiple of 16: per
// sub-quantizer the table is queries x centroids^T with the reference's summation order, and one v_mfma_f32_16x16x1 is one
// fused multiply-add of that order for 16 queries x 16 centroids x 4 partial-sum quarters (the mapping of flat.hip's
// k_flat_scan_mfma: accumulator set k = 2a + tt, block = quarter, reduce tree in the lane).  A wave takes 16 queries x 16
// centroids; lane 16 blk + i reads query i and centroid i at elements 32b + 8a + 4tt + blk straight from global memory (the
// centroid table is L2-resident); a lane's four results are 16 consecutive centroids across the lanes: 64-byte stores.
typedef float pq_f16v __attribute__((ext_vector_type(16)));
template <int NB>
__global__ __launch_bounds__(256) void k_pq_lut_mfma(const float *__restrict__ queries, uint32_t nq, uint32_t dim,
                                                     const float *__restrict__ cent, uint32_t M, uint32_t K,
                                                     int metric, float *__restrict__ lut) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t m = blockIdx.y, c0 = blockIdx.x * 16, q0 = (blockIdx.z * 4 + wave) * 16;
  if (q0 >= nq) return;
  constexpr uint32_t sub_len = NB * 32;
  const uint32_t blk = lane >> 4;
  const float *qrow = queries + (size_t)min(q0 + (lane & 15), nq - 1) * dim + (size_t)m * sub_len + blk;
  const float *crow = cent + ((size_t)m * K + c0 + (lane & 15)) * sub_len + blk;
  pq_f16v acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[k][r] = 0.0f;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    float x[8], y[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = qrow[32 * b + 4 * k], y[k] = crow[32 * b + 4 * k];  // 8a + 4tt = 4k
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_mfma_f32_16x16x1f32(x[k], y[k], acc[k], 0, 0, 0);
  }
  const pq_f16v s0 = ((acc[0] + acc[2]) + acc[4]) + acc[6];
  const pq_f16v s1 = ((acc[1] + acc[3]) + acc[5]) + acc[7];
  const pq_f16v r4 = (s0 + s1) + 0.0f;
#pragma unroll
  for (int i4 = 0; i4 < 4; i4++) {
    const uint32_t q = q0 + 4 * blk + i4;
    const float d = metric_finish((r4[i4] + r4[4 + i4]) + (r4[8 + i4] + r4[12 + i4]), metric);
    if (q < nq) lut[((size_t)q * M + m) * K + c0 + (lane & 15)] = d;
  }
}
template <int NB>
static void launch_lut_mfma(const sdb_pq *pq, const float *d_queries, uint64_t nq, float *d_lut, hipStream_t stream) {
  if constexpr (NB >= 1) {
    const dim3 grid(pq->K / 16, pq->M, (unsigned)((nq + 63) / 64));
    hipLaunchKernelGGL((k_pq_lut_mfma<NB>), grid, dim3(256), 0, stream, d_queries, (uint32_t)nq, pq->dim, pq->d_centroids, pq->M,
                       pq->K, pq->metric, d_lut);
  }
}

#define SDB_NB_SWITCH(fn, ...)                 \
  switch (pq->sub_len / 32) {                  \
    case 0: fn<0>(__VA_ARGS__); break;         \
    case 1: fn<1>(__VA_ARGS__); break;         \
    case 2: fn<2>(__VA_ARGS__); break;         \
    case 3: fn<3>(__VA_ARGS__); break;         \
    case 4: fn<4>(__VA_ARGS__); break;         \
    case 5: fn<5>(__VA_ARGS__); break;         \
    case 6: fn<6>(__VA_ARGS__); break;         \
    case 7: fn<7>(__VA_ARGS__); break;         \
    default: fn<8>(__VA_ARGS__); break;        \
  }

// out[q][c] = sum_i lut[q][i][code_c_i], sequential fp32 adds in index order (product.go:271-275)
__global__ void k_pq_lut_dist(const float *__restrict__ lut, const uint8_t *__restrict__ codes, uint64_t nc,
                              uint32_t M, uint32_t K, float *__restrict__ out) {
  const uint32_t q = blockIdx.y;
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const float *l = lut + (size_t)q * M * K;
  const uint8_t *cd = codes + c * M;
  float dist = 0.0f;
  for (uint32_t i = 0; i < M; i++) dist += l[i * K + cd[i]];
  out[(size_t)q * nc + c] = dist;
}

// symmetric distance via the centroid-pair table (product.go:300-302)
__global__ void k_pq_sym(const float *__restrict__ cdists, const uint8_t *__restrict__ cx,
                         const uint8_t *__restrict__ cy, uint64_t n, uint32_t M, uint32_t K, float *__restrict__ out) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float dist = 0.0f;
  for (uint32_t i = 0; i < M; i++) dist += cdists[((size_t)i * K + cx[p * M + i]) * K + cy[p * M + i]];
  out[p] = dist;
}

__global__ void k_scatter_labels(const uint8_t *__restrict__ labels, uint8_t *__restrict__ codes, uint32_t n,
                                 uint32_t M, uint32_t i) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) codes[(size_t)j * M + i] = labels[j];
}

int pq_build_lut(const sdb_pq *pq, const float *d_queries, uint64_t nq, float *d_lut, hipStream_t stream) {
  if (nq == 0) return SDB_OK;
  if (pq->metric != SDB_METRIC_EUCLIDEAN && pq->sub_len % 32 == 0 && pq->sub_len >= 32 && pq->sub_len <= 32 * kRegBlocksMax &&
      pq->K % 16 == 0) {
    SDB_NB_SWITCH(launch_lut_mfma, pq, d_queries, nq, d_lut, stream)
  } else if (pq->sub_len < 32 * (kRegBlocksMax + 1)) {
    SDB_NB_SWITCH(launch_lut_t, pq, d_queries, nq, d_lut, stream)
  } else {
    const uint32_t MK = pq->M * pq->K;
    hipLaunchKernelGGL(k_pq_lut, dim3((MK + 127) / 128, (unsigned)nq), dim3(128), 0, stream, d_queries, pq->dim,
                       pq->d_centroids, pq->M, pq->K, pq->sub_len, pq->metric, d_lut);
  }
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}

int pq_encode_device(const sdb_pq *pq, const float *d_vecs, uint64_t n, uint8_t *d_codes, hipStream_t stream) {
  if (n == 0) return SDB_OK;
  if (pq->sub_len < 32 * (kRegBlocksMax + 1)) {
    SDB_NB_SWITCH(launch_encode_t, pq, d_vecs, n, d_codes, stream)
  } else {
    hipLaunchKernelGGL(k_pq_encode, dim3((unsigned)((n + 127) / 128), pq->M), dim3(128), 0, stream, d_vecs, n, pq->dim,
                       pq->d_centroids, pq->M, pq->K, pq->sub_len, pq->metric, d_codes);
  }
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}

static int pq_fill_cdists(sdb_pq *pq, hipStream_t stream) {
  hipLaunchKernelGGL(k_pq_cdists, dim3(pq->K, pq->M), dim3(64), 0, stream, pq->d_centroids, pq->d_cdists, pq->M, pq->K,
                     pq->sub_len, pq->metric);
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}

// host/device staging helper: returns a device pointer for `p` (copying when it is host memory)
struct Staged {
  void *dev = nullptr;
  void *owned = nullptr;
  void *host_dst = nullptr;
  size_t bytes = 0;
  ~Staged() {
    if (owned) (void)hipFree(owned);
  }
};

static int stage_in(Staged &s, const void *p, size_t bytes, int mem, hipStream_t stream, bool copy = true) {
  s.bytes = bytes;
  if (mem == SDB_MEM_DEVICE) {
    s.dev = const_cast<void *>(p);
    return SDB_OK;
  }
  SDB_HIP(hipMalloc(&s.owned, bytes ? bytes : 16));
  s.dev = s.owned;
  s.host_dst = const_cast<void *>(p);
  if (copy && bytes) SDB_HIP(hipMemcpyAsync(s.dev, p, bytes, hipMemcpyHostToDevice, stream));
  return SDB_OK;
}

static int stage_out(Staged &s, hipStream_t stream) {
  if (s.owned && s.host_dst && s.bytes) SDB_HIP(hipMemcpyAsync(s.host_dst, s.dev, s.bytes, hipMemcpyDeviceToHost, stream));
  return SDB_OK;
}

}  // namespace sdb

using namespace sdb;

extern "C" {

int sdb_kmeans_fit(float *X, uint32_t n, uint32_t stride, uint32_t offset, uint32_t len, uint32_t K,
                   uint32_t max_iter, uint32_t first_idx, int alias, float *centroids_out, uint8_t *labels_out,
                   uint32_t *iters_out, int mem, int device, void *stream_) {
  if (!X || !centroids_out || !labels_out) return fail(SDB_ERR_INVALID, "NULL argument");
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  if (device < 0 || device >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", device);
  DeviceGuard dg(device);
  hipStream_t stream = as_stream(stream_);
  Staged sx, sc, sl;
  SDB_TRY(stage_in(sx, X, (size_t)n * stride * 4, mem, stream));
  SDB_TRY(stage_in(sc, centroids_out, (size_t)K * len * 4, mem, stream, false));
  SDB_TRY(stage_in(sl, labels_out, n, mem, stream, false));
  int rc = kmeans_device((float *)sx.dev, n, stride, offset, len, K, max_iter, first_idx, alias, (float *)sc.dev,
                         (uint8_t *)sl.dev, iters_out, stream);
  if (rc != SDB_OK) {
    (void)hipStreamSynchronize(stream);
    return rc;
  }
  SDB_TRY(stage_out(sc, stream));
  SDB_TRY(stage_out(sl, stream));
  if (alias) SDB_TRY(stage_out(sx, stream));  // the reference overwrites the caller's rows (kmeans.go:144)
  SDB_HIP(hipStreamSynchronize(stream));
  return SDB_OK;
}

int sdb_pq_create(uint32_t dim, uint32_t metric, uint32_t num_subvectors, uint32_t num_centroids, int device,
                  sdb_pq **out) {
  if (!out) return fail(SDB_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (dim < 1 || dim > 4096) return fail(SDB_ERR_INVALID, "vector size must be between 1 and 4096, got %u", dim);
  if (num_subvectors == 0 || dim % num_subvectors != 0)  // product.go:44-46
    return fail(SDB_ERR_INVALID, "vector length %u must be divisible by num subvectors %u", dim, num_subvectors);
  if (metric > SDB_METRIC_DOT)  // product.go:48-50
    return fail(SDB_ERR_INVALID, "distance function %u not supported for product quantisation", metric);
  if (num_centroids > 256)  // product.go:63-65
    return fail(SDB_ERR_INVALID, "number of centroids %u cannot exceed 256", num_centroids);
  if (num_centroids < 1) return fail(SDB_ERR_INVALID, "number of centroids must be positive");
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  if (device < 0 || device >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", device);
  DeviceGuard dg(device);
  auto *pq = new sdb_pq();
  pq->dim = dim, pq->M = num_subvectors, pq->K = num_centroids, pq->sub_len = dim / num_subvectors;
  pq->metric = metric == SDB_METRIC_COSINE ? SDB_METRIC_EUCLIDEAN : (int)metric;  // product.go:52-61
  pq->device = device;
  hipError_t e = hipMalloc(&pq->d_centroids, (size_t)pq->M * pq->K * pq->sub_len * 4);
  if (e == hipSuccess) e = hipMalloc(&pq->d_cdists, (size_t)pq->M * pq->K * pq->K * 4);
  if (e != hipSuccess) {
    if (pq->d_centroids) (void)hipFree(pq->d_centroids);
    delete pq;
    return fail(SDB_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  *out = pq;
  return SDB_OK;
}

int sdb_pq_destroy(sdb_pq *pq) {
  if (!pq) return SDB_OK;
  DeviceGuard dg(pq->device);
  (void)hipDeviceSynchronize();
  if (pq->d_centroids) (void)hipFree(pq->d_centroids);
  if (pq->d_cdists) (void)hipFree(pq->d_cdists);
  delete pq;
  return SDB_OK;
}

int sdb_pq_fit(sdb_pq *pq, float *X, uint32_t n, const uint32_t *first_idx, int alias, uint8_t *codes_out, int mem,
               void *stream_) {
  if (!pq || !X || !first_idx) return fail(SDB_ERR_INVALID, "NULL argument");
  if (n == 0) return fail(SDB_ERR_INVALID, "no vectors to fit");
  DeviceGuard dg(pq->device);
  hipStream_t stream = as_stream(stream_);
  Staged sx, sc;
  SDB_TRY(stage_in(sx, X, (size_t)n * pq->dim * 4, mem, stream));
  SDB_TRY(stage_in(sc, codes_out, codes_out ? (size_t)n * pq->M : 0, codes_out ? mem : SDB_MEM_HOST, stream, false));
  uint8_t *labels = nullptr;
  SDB_HIP(hipMalloc(&labels, n));
  int rc = SDB_OK;
  for (uint32_t i = 0; i < pq->M && rc == SDB_OK; i++) {  // one goroutine per sub-quantizer, product.go:202-232
    rc = kmeans_device((float *)sx.dev, n, pq->dim, i * pq->sub_len, pq->sub_len, pq->K, 100, first_idx[i], alias,
                       pq->d_centroids + (size_t)i * pq->K * pq->sub_len, labels, nullptr, stream);
    if (rc == SDB_OK && codes_out)  // :216-218
      hipLaunchKernelGGL(k_scatter_labels, dim3((n + 255) / 256), dim3(256), 0, stream, labels, (uint8_t *)sc.dev, n,
                         pq->M, i);
  }
  if (rc == SDB_OK) rc = pq_fill_cdists(pq, stream);
  if (rc == SDB_OK && codes_out) rc = stage_out(sc, stream);
  if (rc == SDB_OK && alias) rc = stage_out(sx, stream);
  (void)hipStreamSynchronize(stream);
  (void)hipFree(labels);
  if (rc == SDB_OK) pq->fitted = true;
  return rc;
}

int sdb_pq_set_codebook(sdb_pq *pq, const float *flat_centroids, int mem) {
  if (!pq || !flat_centroids) return fail(SDB_ERR_INVALID, "NULL argument");
  DeviceGuard dg(pq->device);
  SDB_HIP(hipMemcpy(pq->d_centroids, flat_centroids, (size_t)pq->M * pq->K * pq->sub_len * 4,
                    mem == SDB_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
  SDB_TRY(pq_fill_cdists(pq, nullptr));
  SDB_HIP(hipDeviceSynchronize());
  pq->fitted = true;
  return SDB_OK;
}

int sdb_pq_get_codebook(const sdb_pq *pq, float *flat_centroids, float *centroid_dists) {
  if (!pq) return fail(SDB_ERR_INVALID, "NULL argument");
  if (!pq->fitted) return fail(SDB_ERR_STATE, "quantizer is not fitted");
  DeviceGuard dg(pq->device);
  SDB_HIP(hipDeviceSynchronize());
  if (flat_centroids)
    SDB_HIP(hipMemcpy(flat_centroids, pq->d_centroids, (size_t)pq->M * pq->K * pq->sub_len * 4, hipMemcpyDeviceToHost));
  if (centroid_dists)
    SDB_HIP(hipMemcpy(centroid_dists, pq->d_cdists, (size_t)pq->M * pq->K * pq->K * 4, hipMemcpyDeviceToHost));
  return SDB_OK;
}

int sdb_pq_encode(const sdb_pq *pq, const float *vectors, uint64_t n, uint8_t *codes, int mem, void *stream_) {
  if (!pq || !vectors || !codes) return fail(SDB_ERR_INVALID, "NULL argument");
  if (!pq->fitted) return fail(SDB_ERR_STATE, "quantizer is not fitted");  // encode returns nil, product.go:137-139
  if (n == 0) return SDB_OK;
  DeviceGuard dg(pq->device);
  hipStream_t stream = as_stream(stream_);
  Staged sv, sc;
  SDB_TRY(stage_in(sv, vectors, n * pq->dim * 4, mem, stream));
  SDB_TRY(stage_in(sc, codes, n * pq->M, mem, stream, false));
  SDB_TRY(pq_encode_device(pq, (const float *)sv.dev, n, (uint8_t *)sc.dev, stream));
  SDB_TRY(stage_out(sc, stream));
  if (mem == SDB_MEM_HOST) SDB_HIP(hipStreamSynchronize(stream));
  return SDB_OK;
}

int sdb_pq_lut_distance(const sdb_pq *pq, const float *queries, uint64_t nq, const uint8_t *codes, uint64_t nc,
                        float *out, int mem, void *stream_) {
  if (!pq || !queries || !codes || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  if (!pq->fitted) return fail(SDB_ERR_STATE, "quantizer is not fitted");
  if (nq == 0 || nc == 0) return SDB_OK;
  if (nq > 65535) return fail(SDB_ERR_INVALID, "at most 65535 queries per call");
  DeviceGuard dg(pq->device);
  hipStream_t stream = as_stream(stream_);
  Staged sq, sc, so;
  SDB_TRY(stage_in(sq, queries, nq * pq->dim * 4, mem, stream));
  SDB_TRY(stage_in(sc, codes, nc * pq->M, mem, stream));
  SDB_TRY(stage_in(so, out, nq * nc * 4, mem, stream, false));
  float *lut = nullptr;
  SDB_HIP(hipMalloc(&lut, nq * pq->M * pq->K * 4));
  int rc = pq_build_lut(pq, (const float *)sq.dev, nq, lut, stream);
  if (rc == SDB_OK) {
    hipLaunchKernelGGL(k_pq_lut_dist, dim3((unsigned)((nc + 255) / 256), (unsigned)nq), dim3(256), 0, stream, lut,
                       (const uint8_t *)sc.dev, nc, pq->M, pq->K, (float *)so.dev);
    if (hipGetLastError() != hipSuccess) rc = fail(SDB_ERR_DEVICE, "lut_distance launch failed");
  }
  if (rc == SDB_OK) rc = stage_out(so, stream);
  (void)hipStreamSynchronize(stream);
  (void)hipFree(lut);
  return rc;
}

int sdb_pq_sym_distance(const sdb_pq *pq, const uint8_t *codes_x, const uint8_t *codes_y, uint64_t n, float *out,
                        int mem, void *stream_) {
  if (!pq || !codes_x || !codes_y || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  if (!pq->fitted) return fail(SDB_ERR_STATE, "quantizer is not fitted");
  if (n == 0) return SDB_OK;
  DeviceGuard dg(pq->device);
  hipStream_t stream = as_stream(stream_);
  Staged sx, sy, so;
  SDB_TRY(stage_in(sx, codes_x, n * pq->M, mem, stream));
  SDB_TRY(stage_in(sy, codes_y, n * pq->M, mem, stream));
  SDB_TRY(stage_in(so, out, n * 4, mem, stream, false));
  hipLaunchKernelGGL(k_pq_sym, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pq->d_cdists,
                     (const uint8_t *)sx.dev, (const uint8_t *)sy.dev, n, pq->M, pq->K, (float *)so.dev);
  SDB_HIP(hipGetLastError());
  SDB_TRY(stage_out(so, stream));
  if (mem == SDB_MEM_HOST) SDB_HIP(hipStreamSynchronize(stream));
  return SDB_OK;
}

}  // extern "C"
