#include <type_traits>
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
typedef int pq_f16s __attribute__((ext_vector_type(16)));  // (an integer tuple: a float one is not accepted as a scalar asm operand)
// 32 consecutive floats at a wave-uniform address + a constant byte offset, into two 16-register scalar tuples; the
// caller waits (s_waitcnt lgkmcnt(0) tied to the tuples) before it reads them
__device__ __forceinline__ void sload_block(uint64_t a, int byte_off, pq_f16s &lo, pq_f16s &hi) {
  asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"
               : "=&s"(lo), "=&s"(hi)
               : "s"(a), "i"(byte_off), "i"(byte_off + 64)
               : "memory");
}
template <bool L2, int NB, int TC = -1, typename UP = const float *__restrict__>  // TC >= 0: the tail's length, known at compile time
__device__ __forceinline__ float dist_regs(const float *r, UP u, const float *__restrict__ rt, uint32_t tail_rt) {
  const uint32_t tail = TC >= 0 ? (uint32_t)TC : tail_rt;
  // partial sums 2p and 2p + 1 advance together: one v_pk_fma_f32 (and, for euclidean, one packed subtract, rounded per
  // element like VSUBPS) per pair of elements -- each half is the reference's own operation on its own chain
  pq_f2v acc2[16];
#pragma unroll
  for (int p = 0; p < 16; p++) acc2[p] = pq_f2v{0.0f, 0.0f};
  if constexpr (NB >= 4 || (NB == 3 && TC < 0)) {  // (three blocks + a run-time tail: the same, its chain adds 31 scalars)
    // Long sub-vectors (128 floats and more): `u` is wave-uniform and its elements arrive through scalar loads.  Left to
    // the compiler ALL 32 NB of them are asked for before the first multiply -- 128 .. 256 scalars into ~100 SGPRs: up to
    // 4 093 were spilled into VGPR lanes (two lane moves per scalar, against one packed FMA per two scalars), and at
    // NB = 8 the lanes' registers pushed 32 VGPRs into scratch.  (Invariant loads carry no ordering a scheduling barrier
    // could hold on to.)  So the loads are written out: one block = two s_load_dwordx16, the next block's issued when
    // this one's have arrived and in flight while it is multiplied -- 64 scalars live, nothing spilled.
    // (the address is uniform by construction -- a centroid's or a query's sub-vector -- but not always provably so: a
    // centroid row index read from memory sits in a VGPR)
    uint64_t ua = (uint64_t)(uintptr_t)(const void *)u;
    const uint32_t ua_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ua);  // (the builtin returns a SIGNED int)
    const uint32_t ua_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ua >> 32));
    ua = ((uint64_t)ua_hi << 32) | ua_lo;
    pq_f16s c0, c1;
    sload_block(ua, 0, c0, c1);
#pragma unroll
    for (int b = 0; b < NB; b++) {
      // (the accumulators ride through the wait: the previous block's multiplies are thereby in front of it -- without
      // that they sink behind ALL the loads, and every block's scalars are parked in VGPR lanes until then)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+s"(c0), "+s"(c1), "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]), "+v"(acc2[4]), "+v"(acc2[5]),
                     "+v"(acc2[6]), "+v"(acc2[7]), "+v"(acc2[8]), "+v"(acc2[9]), "+v"(acc2[10]), "+v"(acc2[11]), "+v"(acc2[12]),
                     "+v"(acc2[13]), "+v"(acc2[14]), "+v"(acc2[15]));
      pq_f16s n0 = c0, n1 = c1;
      if (b + 1 < NB) sload_block(ua, 128 * (b + 1), n0, n1);
#pragma unroll
      for (int p = 0; p < 16; p++) {
        const pq_f2v x = {r[32 * b + 2 * p], r[32 * b + 2 * p + 1]};
        const pq_f2v yy = p < 8 ? pq_f2v{__int_as_float(c0[2 * p]), __int_as_float(c0[2 * p + 1])}
                                : pq_f2v{__int_as_float(c1[2 * p - 16]), __int_as_float(c1[2 * p - 15])};
        if constexpr (L2) {
          const pq_f2v d = x - yy;
          acc2[p] = __builtin_elementwise_fma(d, d, acc2[p]);
        } else {
          acc2[p] = __builtin_elementwise_fma(x, yy, acc2[p]);
        }
      }
      c0 = n0, c1 = n1;
    }
  } else {
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
// k-means, M problems at once.  The reference fits the M sub-quantizers of a product quantizer concurrently, one
// goroutine each over its own columns of the same rows (product.go:202-232); here problem m is the sub-vector
// [offset0 + m * len, offset0 + (m + 1) * len) of every row and every launch below carries all M problems in one
// grid dimension.  A problem that has converged (no label changed, kmeans.go:116-118) idles through the remaining
// launches on a device-side flag: no host round trip per iteration, none per problem (round 3 ran the problems one
// after another with two launches per furthest-point step and a stream synchronise per Lloyd iteration: 10.45 s at
// M = 192, K = 256, 10 000 rows).  KMeans.Fit itself (sdb_kmeans_fit) is the M = 1 case.
//
// Centroid (m, j) is a VIEW: row cent_row[m][j] of cent_base at the problem's columns -- either into X itself (the
// reference's aliasing, kmeans.go:63,82,144) or into a private copy.
// ---------------------------------------------------------------------------------------------
struct KmArgs {
  float *X;
  uint32_t n, stride, offset0, len, K, M;
  // centroid (m, j) = cent_base + m * cent_m_stride + cent_row[m * K + j] * cent_stride + cent_off0 + m * cent_off_step
  float *cent_base;
  size_t cent_m_stride;
  uint32_t cent_stride, cent_off0, cent_off_step;
  uint32_t *cent_row;         // [M][K]
  const float *Xt;            // [M * len][n] the problems' columns, transposed (furthest-point phase: coalesced)
  float *min_dist;            // [M][n]
  uint8_t *labels;            // [M][lab_stride]
  uint32_t lab_stride;        // n rounded up to 16
  uint32_t *change;           // [M] labels changed by the current assignment
  uint32_t *done;             // [M] converged
  uint32_t *iters;            // [M] assignment passes run (kmeans.go:96)
  uint32_t *n_done;           // [1]
  float *sums;                // [M][K][len]
  uint32_t *counts;           // [M][K]
  uint32_t *lbase;            // [M][K] first member of label l in order[]
  uint32_t *order;            // [M][n] members by label, in data order within a label
  const uint32_t *first_idx;  // [M]
};

__device__ __forceinline__ float *km_centroid(const KmArgs &a, uint32_t m, uint32_t j) {
  return a.cent_base + (size_t)m * a.cent_m_stride + (size_t)a.cent_row[(size_t)m * a.K + j] * a.cent_stride + a.cent_off0 +
         m * a.cent_off_step;
}

// Xt[c][j] = X[j][offset0 + c]: a 32 x 32 tile through LDS
__global__ __launch_bounds__(256) void k_km_transpose(const float *__restrict__ X, uint32_t n, uint32_t stride, uint32_t offset0,
                                                      uint32_t cols, float *__restrict__ Xt) {
  __shared__ float tile[32][33];
  const uint32_t j0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (uint32_t r = ty; r < 32; r += 8)
    if (j0 + r < n && c0 + tx < cols) tile[r][tx] = X[(size_t)(j0 + r) * stride + offset0 + c0 + tx];
  __syncthreads();
  for (uint32_t r = ty; r < 32; r += 8)
    if (c0 + r < cols && j0 + tx < n) Xt[(size_t)(c0 + r) * n + j0 + tx] = tile[tx][r];
}

// asm.SquaredEuclideanDistance (euclidean.s:7-65) of point j's sub-vector, read down a column of Xt (element e at
// xt[e * n]), against a centroid whose elements arrive through the scalar cache -- dist_serial's arithmetic
__device__ __forceinline__ float dist_column(const float *__restrict__ xt, size_t n, uniform_float *c, uint32_t len) {
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; i++) acc[i] = 0.0f;
  const uint32_t nblk = len / 32;
  for (uint32_t b = 0; b < nblk; b++) {
#pragma unroll
    for (int L = 0; L < 32; L++) acc[L] = chain1<true>(acc[L], xt[(size_t)(32 * b + L) * n], c[32 * b + L]);
  }
  float t = 0.0f;
  for (uint32_t i = nblk * 32; i < len; i++) t = chain1<true>(t, xt[(size_t)i * n], c[i]);
  float r[4];
#pragma unroll
  for (int l = 0; l < 4; l++) {
    const float s0 = ((acc[l] + acc[8 + l]) + acc[16 + l]) + acc[24 + l];
    const float s1 = ((acc[l + 4] + acc[12 + l]) + acc[20 + l]) + acc[28 + l];
    r[l] = s0 + s1;
  }
  r[0] = r[0] + t;
  r[1] = r[1] + 0.0f;
  r[2] = r[2] + 0.0f;
  r[3] = r[3] + 0.0f;
  return (r[0] + r[1]) + (r[2] + r[3]);
}

// furthestId's order (kmeans.go:66-80): strict '>' scanning ascending -- the lowest index among equal maxima wins and
// index 0 is returned when nothing is > 0
__device__ __forceinline__ void km_far_combine(float &v, uint32_t &i, float ov, uint32_t oi) {
  if (ov > v || (ov == v && ov > 0.0f && oi < i)) v = ov, i = oi;
}

// The whole furthest-point initialisation (kmeans.go:56-83) of problem m by ONE workgroup: K - 1 steps of (distance of
// every point to the centroid chosen last, running minimum per point, arg max), the chosen row handed from step to step
// through LDS.  Nothing is written but min_dist and the row list, so the points are read from the transposed copy.
__global__ __launch_bounds__(1024) void k_km_init(const KmArgs a) {
  __shared__ float s_v[16];
  __shared__ uint32_t s_i[16];
  __shared__ uint32_t s_row;
  const uint32_t m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t first = a.first_idx[m];
  float *md = a.min_dist + (size_t)m * a.n;
  const float *xt = a.Xt + (size_t)m * a.len * a.n;
  uint32_t *rows = a.cent_row + (size_t)m * a.K;
  for (uint32_t j = tid; j < a.n; j += 1024) md[j] = FLT_MAX;
  if (tid == 0) rows[0] = first;
  uint32_t prev = first;
  for (uint32_t i = 1; i < a.K; i++) {
    uniform_float *c = as_uniform(a.X + (size_t)prev * a.stride + a.offset0 + (size_t)m * a.len);
    float best = 0.0f;
    uint32_t best_i = 0;
    for (uint32_t j = tid; j < a.n; j += 1024) {
      if (j == first) continue;  // alreadyCentroid only ever holds randId (:60-62,68)
      const float d = dist_column(xt + j, a.n, c, a.len);
      float v = md[j];
      if (d < v) v = d, md[j] = d;
      if (v > best) best = v, best_i = j;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) km_far_combine(best, best_i, __shfl_down(best, o, 64), __shfl_down(best_i, o, 64));
    if (lane == 0) s_v[wave] = best, s_i[wave] = best_i;
    __syncthreads();
    if (wave == 0) {
      float v = lane < 16 ? s_v[lane] : 0.0f;
      uint32_t vi = lane < 16 ? s_i[lane] : 0u;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) km_far_combine(v, vi, __shfl_down(v, o, 64), __shfl_down(vi, o, 64));
      if (lane == 0) {
        const uint32_t r = v > 0.0f ? vi : 0u;
        s_row = r, rows[i] = r;
      }
    }
    __syncthreads();
    prev = s_row;
    __syncthreads();  // s_v / s_i / s_row are rewritten by the next step
  }
}

// assignment (kmeans.go:100-115): argmin over centroids, strict '<' starting from centroid 0.  Thread = point, its
// sub-vector in registers, the centroids wave-uniform through the scalar cache (k_pq_encode_t's scheme).
template <int NB, int LEN = 0>  // LEN: the sub-vector length when it is a usual one (k_pq_encode_pair), 0 = any
__global__ __launch_bounds__(256) void k_km_assign_t(const KmArgs a_in) {
  KmArgs a = a_in;
  if (LEN > 0) a.len = LEN;
  const uint32_t m = blockIdx.y;
  if (a.done[m]) return;
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  const float *sub = a.X + (size_t)(v < a.n ? v : a.n - 1) * a.stride + a.offset0 + (size_t)m * a.len;
  float r[NB > 0 ? NB * 32 : 1];
#pragma unroll
  for (int e = 0; e < NB * 32; e++) r[e] = sub[e];
  const uint32_t tail = a.len - NB * 32;
  float rt[31];
#pragma unroll
  for (int e = 0; e < 31; e++) rt[e] = (uint32_t)e < tail ? sub[NB * 32 + e] : 0.0f;
  constexpr int TCV = LEN > 0 ? LEN - NB * 32 : -1;
  float best = dist_regs<true, NB, TCV>(r, as_uniform(km_centroid(a, m, 0)), rt, tail);
  uint32_t best_id = 0;
  for (uint32_t j = 1; j < a.K; j++) {
    const float d = dist_regs<true, NB, TCV>(r, as_uniform(km_centroid(a, m, j)), rt, tail);
    if (d < best) best = d, best_id = j;
  }
  uint8_t *lab = a.labels + (size_t)m * a.lab_stride;
  const bool ch = v < a.n && lab[v] != (uint8_t)best_id;
  if (ch) lab[v] = (uint8_t)best_id;
  const uint64_t b = __ballot(ch);
  if (b && (threadIdx.x & 63) == 0) atomicAdd(a.change + m, (uint32_t)__popcll(b));
}

// the same for sub-vectors too long for the register file
__global__ __launch_bounds__(64) void k_km_assign(const KmArgs a) {
  const uint32_t m = blockIdx.y;
  if (a.done[m]) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float *sv = a.X + (size_t)i * a.stride + a.offset0 + (size_t)m * a.len;
  float best = dist_serial<true>(sv, km_centroid(a, m, 0), a.len);
  uint32_t best_id = 0;
  for (uint32_t j = 1; j < a.K; j++) {
    const float d = dist_serial<true>(sv, km_centroid(a, m, j), a.len);
    if (d < best) best = d, best_id = j;
  }
  uint8_t *lab = a.labels + (size_t)m * a.lab_stride;
  if (lab[i] != (uint8_t)best_id) {
    lab[i] = (uint8_t)best_id;
    atomicAdd(a.change + m, 1u);
  }
}

// The members of every label in DATA ORDER (the sums below are fp32 additions in that order, kmeans.go:125-137, and
// fp32 addition is not associative): thread (label l, segment s) counts, then lists, the points of segment s that
// carry label l -- the segments are four contiguous quarters of the points, so (l, s = 0..3) in sequence is data order.
// The labels are wave-uniform reads (a wave = 64 labels of one segment): scalar loads, 4 labels per word.
__global__ __launch_bounds__(1024) void k_km_members(const KmArgs a) {
  const uint32_t m = blockIdx.x;
  if (a.done[m] || a.change[m] == 0) return;  // converged before, or with this very assignment (:116-118)
  __shared__ uint32_t s_cnt[4][256];
  __shared__ uint32_t s_scan[256];
  const uint32_t tid = threadIdx.x, l = tid & 255, s = tid >> 8;
  typedef const __attribute__((address_space(4))) uint32_t uniform_u32;
  uniform_u32 *lab = (uniform_u32 *)(a.labels + (size_t)m * a.lab_stride);
  const uint32_t words = (a.n + 3) / 4, seg = (((words + 3) / 4) + 3) & ~3u;
  const uint32_t w0 = min(words, s * seg), w1 = min(words, w0 + seg);
  uint32_t cnt = 0;
  for (uint32_t w = w0; w < w1; w++) {
    const uint32_t x = lab[w];
#pragma unroll
    for (int b = 0; b < 4; b++) cnt += (4 * w + b < a.n && ((x >> (8 * b)) & 255u) == l) ? 1u : 0u;
  }
  s_cnt[s][l] = cnt;
  __syncthreads();
  uint32_t tot = 0;
  if (tid < 256) {
    tot = s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
    s_scan[tid] = tot;
  }
  __syncthreads();
  for (uint32_t o = 1; o < 256; o <<= 1) {  // inclusive scan over the labels
    uint32_t add = 0;
    if (tid < 256 && tid >= o) add = s_scan[tid - o];
    __syncthreads();
    if (tid < 256) s_scan[tid] += add;
    __syncthreads();
  }
  if (tid < 256 && tid < a.K) {
    a.counts[(size_t)m * a.K + tid] = tot;
    a.lbase[(size_t)m * a.K + tid] = s_scan[tid] - tot;
  }
  uint32_t pos = s_scan[l] - (s_cnt[0][l] + s_cnt[1][l] + s_cnt[2][l] + s_cnt[3][l]);
  for (uint32_t k = 0; k < s; k++) pos += s_cnt[k][l];
  uint32_t *ord = a.order + (size_t)m * a.n;
  for (uint32_t w = w0; w < w1; w++) {
    const uint32_t x = lab[w];
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (4 * w + b < a.n && ((x >> (8 * b)) & 255u) == l) ord[pos++] = 4 * w + b;
  }
}

// update sums (kmeans.go:125-137): thread (label, component) adds its members' components in data order
__global__ __launch_bounds__(256) void k_km_sums(const KmArgs a) {
  const uint32_t m = blockIdx.y;
  if (a.done[m] || a.change[m] == 0) return;
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.K * a.len) return;
  const uint32_t l = t / a.len, j = t % a.len;
  const uint32_t c = a.counts[(size_t)m * a.K + l];
  if (c == 0) return;  // no member: the stale sum is never read (:140-142)
  const uint32_t *ord = a.order + (size_t)m * a.n + a.lbase[(size_t)m * a.K + l];
  const float *col = a.X + a.offset0 + (size_t)m * a.len + j;
  float sum = 0.0f;
  uint32_t k = 0;
  for (; k + 4 <= c; k += 4) {  // the loads do not depend on the additions: four in flight
    const float x0 = col[(size_t)ord[k] * a.stride], x1 = col[(size_t)ord[k + 1] * a.stride];
    const float x2 = col[(size_t)ord[k + 2] * a.stride], x3 = col[(size_t)ord[k + 3] * a.stride];
    sum += x0, sum += x1, sum += x2, sum += x3;
  }
  for (; k < c; k++) sum += col[(size_t)ord[k] * a.stride];
  a.sums[((size_t)m * a.K + l) * a.len + j] = sum;
}

// means (kmeans.go:139-146) written through the centroid views.  The reference writes them in centroid order, so when
// two centroids share a row (aliasing over duplicate points) the later one with members wins: a centroid shadowed by
// such a successor does not write.  Then the iteration's bookkeeping of problem m.
__global__ __launch_bounds__(256) void k_km_means(const KmArgs a) {
  const uint32_t m = blockIdx.x, tid = threadIdx.x;
  if (a.done[m]) return;
  __shared__ uint8_t s_shadow[256];
  const uint32_t change = a.change[m];
  if (change != 0) {
    const uint32_t *rows = a.cent_row + (size_t)m * a.K, *cnt = a.counts + (size_t)m * a.K;
    if (tid < a.K) {
      bool sh = false;
      const uint32_t r = rows[tid];
      for (uint32_t k = tid + 1; k < a.K; k++) sh |= rows[k] == r && cnt[k] != 0;
      s_shadow[tid] = sh ? 1 : 0;
    }
    __syncthreads();
    for (uint32_t t = tid; t < a.K * a.len; t += 256) {
      const uint32_t l = t / a.len, j = t % a.len, c = cnt[l];
      if (c != 0 && !s_shadow[l]) km_centroid(a, m, l)[j] = a.sums[((size_t)m * a.K + l) * a.len + j] / (float)c;
    }
  }
  __syncthreads();
  if (tid == 0) {
    a.iters[m] += 1;
    if (change == 0) {
      a.done[m] = 1;
      atomicAdd(a.n_done, 1u);
    }
    a.change[m] = 0;
  }
}

__global__ void k_km_gather_centroids(const KmArgs a, float *out /* [M][K][len] */) {
  const uint32_t j = blockIdx.x, m = blockIdx.y;
  const float *c = km_centroid(a, m, j);
  for (uint32_t t = threadIdx.x; t < a.len; t += blockDim.x) out[((size_t)m * a.K + j) * a.len + t] = c[t];
}

__global__ void k_km_iota_rows(uint32_t *p, uint32_t K, uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) p[i] = i % K;
}

// labels [M][lab_stride] -> out[j * out_stride + m]  (the codes of product.go:216-218; M = 1: the labels themselves)
__global__ void k_km_scatter_labels(const uint8_t *__restrict__ labels, uint32_t lab_stride, uint8_t *__restrict__ out, uint32_t n,
                                    uint32_t out_stride) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (j < n) out[(size_t)j * out_stride + m] = labels[(size_t)m * lab_stride + j];
}

template <int NB>
static void launch_km_assign_t(const KmArgs &a, hipStream_t stream) {
  const dim3 grid((a.n + 255) / 256, a.M);
  if constexpr (NB > 0) {
    if (a.len == (uint32_t)NB * 32) {
      hipLaunchKernelGGL((k_km_assign_t<NB, NB * 32>), grid, dim3(256), 0, stream, a);
      return;
    }
  } else {
    switch (a.len) {
      case 4: hipLaunchKernelGGL((k_km_assign_t<0, 4>), grid, dim3(256), 0, stream, a); return;
      case 8: hipLaunchKernelGGL((k_km_assign_t<0, 8>), grid, dim3(256), 0, stream, a); return;
      case 16: hipLaunchKernelGGL((k_km_assign_t<0, 16>), grid, dim3(256), 0, stream, a); return;
      case 24: hipLaunchKernelGGL((k_km_assign_t<0, 24>), grid, dim3(256), 0, stream, a); return;
      default: break;
    }
  }
  hipLaunchKernelGGL((k_km_assign_t<NB>), grid, dim3(256), 0, stream, a);
}

// KMeans.Fit for M problems on device buffers: problem m = columns [offset0 + m * len, + len) of dX's rows, first
// centroid = row h_first_idx[m].  d_centroids_out [M][K][len]; d_labels_out[j * labels_stride + m] (may be NULL);
// h_iters_out [M] (may be NULL).  Returns after the stream has drained.
int kmeans_device(float *dX, uint32_t n, uint32_t stride, uint32_t offset0, uint32_t len, uint32_t M, uint32_t K,
                  uint32_t max_iter, const uint32_t *h_first_idx, int alias, float *d_centroids_out, uint8_t *d_labels_out,
                  uint32_t labels_stride, uint32_t *h_iters_out, hipStream_t stream) {
  if (n == 0 || K == 0 || K > 256) return fail(SDB_ERR_INVALID, "kmeans: need n > 0 and 1 <= K <= 256");
  if (M == 0 || len == 0 || (uint64_t)offset0 + (uint64_t)M * len > stride) return fail(SDB_ERR_INVALID, "kmeans: sub-vector out of range");
  for (uint32_t m = 0; m < M; m++)
    if (h_first_idx[m] >= n) return fail(SDB_ERR_INVALID, "kmeans: first_idx out of range");
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const uint32_t lab_stride = (n + 15) & ~15u;
  const size_t b_xt = up((size_t)M * len * n * 4), b_min = up((size_t)M * n * 4), b_lab = up((size_t)M * lab_stride);
  const size_t b_rows = up((size_t)M * K * 4), b_sums = up((size_t)M * K * len * 4), b_ord = up((size_t)M * n * 4);
  const size_t b_m = up((size_t)M * 4), b_priv = alias ? 0 : b_sums;
  char *buf = nullptr;
  uint32_t *h_done = nullptr;
  SDB_HIP(hipMalloc(&buf, b_xt + b_min + b_lab + 4 * b_rows + b_sums + b_ord + 4 * b_m + 256 + b_priv));
  struct Free {
    char *p;
    uint32_t **h;
    hipStream_t s;
    ~Free() {
      (void)hipStreamSynchronize(s);
      (void)hipFree(p);
      if (*h) (void)hipHostFree(*h);
    }
  } fr{buf, &h_done, stream};
  SDB_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_done), 64, hipHostMallocDefault));
  KmArgs a{};
  a.X = dX, a.n = n, a.stride = stride, a.offset0 = offset0, a.len = len, a.K = K, a.M = M, a.lab_stride = lab_stride;
  char *p = buf;
  auto take = [&](size_t b) {
    char *r = p;
    p += b;
    return r;
  };
  float *xt = (float *)take(b_xt);
  a.Xt = xt;
  a.min_dist = (float *)take(b_min);
  a.labels = (uint8_t *)take(b_lab);
  uint32_t *rows_x = (uint32_t *)take(b_rows), *rows_id = (uint32_t *)take(b_rows);
  a.counts = (uint32_t *)take(b_rows), a.lbase = (uint32_t *)take(b_rows);
  a.sums = (float *)take(b_sums);
  a.order = (uint32_t *)take(b_ord);
  uint32_t *d_first = (uint32_t *)take(b_m);
  a.first_idx = d_first;
  a.change = (uint32_t *)take(b_m), a.done = (uint32_t *)take(b_m), a.iters = (uint32_t *)take(b_m);
  a.n_done = (uint32_t *)take(256);
  float *priv = alias ? nullptr : (float *)take(b_priv);
  SDB_HIP(hipMemcpyAsync(d_first, h_first_idx, (size_t)M * 4, hipMemcpyHostToDevice, stream));
  SDB_HIP(hipMemsetAsync(a.change, 0, 3 * b_m + 256, stream));  // change, done, iters, n_done
  SDB_HIP(hipMemsetAsync(a.labels, 0, b_lab, stream));           // Labels start at 0 (kmeans.go:87)
  // ---- furthest-point initialisation (kmeans.go:56-83): centroids are views into X
  a.cent_base = dX, a.cent_m_stride = 0, a.cent_stride = stride, a.cent_off0 = offset0, a.cent_off_step = len, a.cent_row = rows_x;
  hipLaunchKernelGGL(k_km_transpose, dim3((n + 31) / 32, (M * len + 31) / 32), dim3(256), 0, stream, dX, n, stride, offset0,
                     M * len, xt);
  hipLaunchKernelGGL(k_km_init, dim3(M), dim3(1024), 0, stream, a);
  SDB_HIP(hipGetLastError());
  if (!alias) {  // fenced mode: work on copies, X stays untouched
    hipLaunchKernelGGL(k_km_gather_centroids, dim3(K, M), dim3(64), 0, stream, a, priv);
    hipLaunchKernelGGL(k_km_iota_rows, dim3((M * K + 255) / 256), dim3(256), 0, stream, rows_id, K, M * K);
    a.cent_base = priv, a.cent_m_stride = (size_t)K * len, a.cent_stride = len, a.cent_off0 = 0, a.cent_off_step = 0;
    a.cent_row = rows_id;
  }
  // ---- Lloyd iterations (kmeans.go:96-147); the host looks at the number of converged problems every 8th one
  for (uint32_t it = 0; it < max_iter; it++) {
    if (len < 32 * (kRegBlocksMax + 1)) {
      switch (len / 32) {
        case 0: launch_km_assign_t<0>(a, stream); break;
        case 1: launch_km_assign_t<1>(a, stream); break;
        case 2: launch_km_assign_t<2>(a, stream); break;
        case 3: launch_km_assign_t<3>(a, stream); break;
        case 4: launch_km_assign_t<4>(a, stream); break;
        case 5: launch_km_assign_t<5>(a, stream); break;
        case 6: launch_km_assign_t<6>(a, stream); break;
        case 7: launch_km_assign_t<7>(a, stream); break;
        default: launch_km_assign_t<8>(a, stream); break;
      }
    } else {
      hipLaunchKernelGGL(k_km_assign, dim3((n + 63) / 64, M), dim3(64), 0, stream, a);
    }
    hipLaunchKernelGGL(k_km_members, dim3(M), dim3(1024), 0, stream, a);
    hipLaunchKernelGGL(k_km_sums, dim3((K * len + 255) / 256, M), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_km_means, dim3(M), dim3(256), 0, stream, a);
    if ((it & 7) == 7 && it + 1 < max_iter) {
      SDB_HIP(hipMemcpyAsync(h_done, a.n_done, 4, hipMemcpyDeviceToHost, stream));
      SDB_HIP(hipStreamSynchronize(stream));
      if (*h_done == M) break;
    }
  }
  SDB_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_km_gather_centroids, dim3(K, M), dim3(64), 0, stream, a, d_centroids_out);
  if (d_labels_out)
    hipLaunchKernelGGL(k_km_scatter_labels, dim3((n + 255) / 256, M), dim3(256), 0, stream, a.labels, lab_stride, d_labels_out, n,
                       labels_stride);
  SDB_HIP(hipGetLastError());
  if (h_iters_out) SDB_HIP(hipMemcpyAsync(h_iters_out, a.iters, (size_t)M * 4, hipMemcpyDeviceToHost, stream));
  SDB_HIP(hipStreamSynchronize(stream));
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
template <bool L2, int NB, int SL = 0>  // SL: the sub-vector length when it is a usual short one (k_pq_encode_pair), 0 = any
__global__ __launch_bounds__(256) void k_pq_lut_t(const float *__restrict__ queries, uint32_t nq, uint32_t dim,
                                                  const float *__restrict__ cent_t /* [M][sub_len][K] */, uint32_t M, uint32_t K,
                                                  uint32_t sub_len_rt, int metric, float *__restrict__ lut) {
  const uint32_t sub_len = SL > 0 ? (uint32_t)SL : sub_len_rt;
  const uint32_t i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  const uint32_t q0 = blockIdx.z * kLutQT, q1 = min(q0 + kLutQT, nq);
  // the thread's centroid row, from the element-major copy of the table (sdb_pq::d_centroids_t): element e of the 64
  // centroids of a wave is one 256-byte line.  (From [K][sub_len] every load instruction of a 96-float row touched 64
  // lines, 384 bytes apart -- the row loads were a fifth of the kernel at M = 8.)
  const float *col = cent_t + (size_t)i * sub_len * K + min(j, K - 1);
  float r[NB > 0 ? NB * 32 : 1];
#pragma unroll
  for (int e = 0; e < NB * 32; e++) r[e] = col[(size_t)e * K];
  const uint32_t tail = sub_len - NB * 32;
  float rt[31];
#pragma unroll
  for (int e = 0; e < 31; e++) rt[e] = (uint32_t)e < tail ? col[(size_t)(NB * 32 + e) * K] : 0.0f;
  // a fixed trip count: the sub-vectors of all kLutQT queries are wave-uniform scalar loads, and unrolled the compiler
  // issues them together instead of one query's, a wait, the next query's
  (void)q1;
  // (sub-vectors of 128 floats and more: one query at a time -- their scalars alone fill the SGPR file)
#pragma unroll NB >= 4 ? 1 : (int)kLutQT
  for (uint32_t k = 0; k < kLutQT; k++) {
    const uint32_t q = q0 + k < nq ? q0 + k : nq - 1;  // past the end: the last query again (the same value is stored twice)
    const float *x = queries + (size_t)q * dim + (size_t)i * sub_len;
    float d = dist_regs<L2, NB, (SL > 0 ? SL - NB * 32 : -1)>(r, x, rt, tail);
    if constexpr (!L2) d = metric_finish(d, metric);
    if (j < K) lut[((size_t)q * M + i) * K + j] = d;
  }
}

// encode: thread = vector (its sub-vector in registers), the K centroids are wave-uniform
template <bool L2, int NB, bool WHOLE = false>  // WHOLE: sub_len == 32 NB, no tail chain
__global__ __launch_bounds__(256) void k_pq_encode_t(const float *__restrict__ vecs, uint64_t n, uint32_t dim,
                                                     const float *__restrict__ cent, uint32_t M, uint32_t K,
                                                     uint32_t sub_len_rt, int metric, uint8_t *__restrict__ codes) {
  const uint32_t sub_len = WHOLE ? (uint32_t)NB * 32 : sub_len_rt;
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
    float d = dist_regs<L2, NB, (WHOLE ? 0 : -1)>(r, cent + ((size_t)i * K + j) * sub_len, rt, tail);
    if constexpr (!L2) d = metric_finish(d, metric);
    if (d < best) best = d, best_id = j;
  }
  if (v < n) codes[v * M + i] = (uint8_t)best_id;
}

// Sub-vectors shorter than 32 floats (M = 32, 192 at d = 768) are nothing but the sequential tail chain: 2 x sub_len
// dependent operations per centroid and thread, which no packing inside ONE pair can shorten.  Two VECTORS per thread
// can: the chains of vector 2t and 2t + 1 against the same centroid advance together in the halves of one packed
// subtract and one packed FMA (the centroid element broadcast to both halves) -- each half is the reference's own
// operation on its own chain.  Half the instructions per (vector, centroid).
// SL: the sub-vector length when it is one of the usual ones (4, 8, 16, 24: d = 768 at M = 192, 96, 48, 32), 0 = any.  With
// the length a run-time value every element of the 31-slot chain is a compare and a branch around two operations -- at
// 4 floats 27 of 31 steps are nothing but that, per centroid.
template <bool L2, int SL = 0>
__global__ __launch_bounds__(256) void k_pq_encode_pair(const float *__restrict__ vecs, uint64_t n, uint32_t dim,
                                                        const float *__restrict__ cent, uint32_t M, uint32_t K,
                                                        uint32_t sub_len_rt, int metric, uint8_t *__restrict__ codes) {
  const uint32_t sub_len = SL > 0 ? (uint32_t)SL : sub_len_rt;
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t va = 2 * t, vb = 2 * t + 1;
  const uint32_t i = blockIdx.y;
  const float *sa = vecs + (va < n ? va : n - 1) * dim + (size_t)i * sub_len;
  const float *sb = vecs + (vb < n ? vb : n - 1) * dim + (size_t)i * sub_len;
  pq_f2v x[31];
#pragma unroll
  for (int e = 0; e < 31; e++) x[e] = (uint32_t)e < sub_len ? pq_f2v{sa[e], sb[e]} : pq_f2v{0.0f, 0.0f};
  float best_a = FLT_MAX, best_b = FLT_MAX;
  uint32_t id_a = 0, id_b = 0;
  for (uint32_t j = 0; j < K; j++) {
    const float *__restrict__ u = cent + ((size_t)i * K + j) * sub_len;
    pq_f2v tt = {0.0f, 0.0f};
#pragma unroll
    for (int e = 0; e < 31; e++)
      if ((uint32_t)e < sub_len) {
        const pq_f2v y = {u[e], u[e]};
        if constexpr (L2) {
          const pq_f2v d = x[e] - y;
          tt = __builtin_elementwise_fma(d, d, tt);
        } else {
          tt = __builtin_elementwise_fma(x[e], y, tt);
        }
      }
    // dist_regs with no whole block: ((0 + t) + 0) + (0 + 0)
    pq_f2v r0 = pq_f2v{0.0f, 0.0f} + tt;
    r0 = (r0 + pq_f2v{0.0f, 0.0f}) + pq_f2v{0.0f, 0.0f};
    float da = r0[0], db = r0[1];
    if constexpr (!L2) da = metric_finish(da, metric), db = metric_finish(db, metric);
    if (da < best_a) best_a = da, id_a = j;
    if (db < best_b) best_b = db, id_b = j;
  }
  if (va < n) codes[va * M + i] = (uint8_t)id_a;
  if (vb < n) codes[vb * M + i] = (uint8_t)id_b;
}

template <int NB>
static void launch_lut_t(const sdb_pq *pq, const float *d_queries, uint64_t nq, float *d_lut, hipStream_t stream) {
  const dim3 grid((pq->K + 255) / 256, pq->M, (unsigned)((nq + kLutQT - 1) / kLutQT));
  auto go = [&](auto sl) {
    constexpr int SL = decltype(sl)::value;
    if (pq->metric == SDB_METRIC_EUCLIDEAN)
      hipLaunchKernelGGL((k_pq_lut_t<true, NB, SL>), grid, dim3(256), 0, stream, d_queries, (uint32_t)nq, pq->dim,
                         pq->d_centroids_t, pq->M, pq->K, pq->sub_len, pq->metric, d_lut);
    else
      hipLaunchKernelGGL((k_pq_lut_t<false, NB, SL>), grid, dim3(256), 0, stream, d_queries, (uint32_t)nq, pq->dim,
                         pq->d_centroids_t, pq->M, pq->K, pq->sub_len, pq->metric, d_lut);
  };
  if constexpr (NB > 0) {
    if (pq->sub_len == (uint32_t)NB * 32) {  // whole blocks, no tail chain at all (M = 8 at d = 768: 96 floats)
      go(std::integral_constant<int, NB * 32>{});
      return;
    }
  }
  if constexpr (NB == 0) {
    switch (pq->sub_len) {
      case 4: go(std::integral_constant<int, 4>{}); return;
      case 8: go(std::integral_constant<int, 8>{}); return;
      case 16: go(std::integral_constant<int, 16>{}); return;
      case 24: go(std::integral_constant<int, 24>{}); return;
      default: break;
    }
  }
  go(std::integral_constant<int, 0>{});
}

template <int NB>
static void launch_encode_t(const sdb_pq *pq, const float *d_vecs, uint64_t n, uint8_t *d_codes, hipStream_t stream) {
  if constexpr (NB == 0) {
    const dim3 pgrid((unsigned)(((n + 1) / 2 + 255) / 256), pq->M);
    auto go = [&](auto sl) {
      constexpr int SL = decltype(sl)::value;
      if (pq->metric == SDB_METRIC_EUCLIDEAN)
        hipLaunchKernelGGL((k_pq_encode_pair<true, SL>), pgrid, dim3(256), 0, stream, d_vecs, n, pq->dim, pq->d_centroids, pq->M,
                           pq->K, pq->sub_len, pq->metric, d_codes);
      else
        hipLaunchKernelGGL((k_pq_encode_pair<false, SL>), pgrid, dim3(256), 0, stream, d_vecs, n, pq->dim, pq->d_centroids, pq->M,
                           pq->K, pq->sub_len, pq->metric, d_codes);
    };
    switch (pq->sub_len) {
      case 4: go(std::integral_constant<int, 4>{}); break;
      case 8: go(std::integral_constant<int, 8>{}); break;
      case 16: go(std::integral_constant<int, 16>{}); break;
      case 24: go(std::integral_constant<int, 24>{}); break;
      default: go(std::integral_constant<int, 0>{}); break;
    }
    return;
  }
  const dim3 grid((unsigned)((n + 255) / 256), pq->M);
  const bool whole = NB > 0 && pq->sub_len == (uint32_t)NB * 32;
  const bool l2 = pq->metric == SDB_METRIC_EUCLIDEAN;
#define SDB_ENC(L2_, WH_)                                                                                                \
  hipLaunchKernelGGL((k_pq_encode_t<L2_, NB, WH_>), grid, dim3(256), 0, stream, d_vecs, n, pq->dim, pq->d_centroids, pq->M, \
                     pq->K, pq->sub_len, pq->metric, d_codes)
  if (l2 && whole) SDB_ENC(true, true);
  else if (l2) SDB_ENC(true, false);
  else if (whole) SDB_ENC(false, true);
  else SDB_ENC(false, false);
#undef SDB_ENC
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

// what is derived from a new codebook: the centroid-pair table (product.go:225-230) and the element-major copy the
// table kernel reads its rows from
__global__ void k_pq_transpose_cent(const float *__restrict__ cent, float *__restrict__ cent_t, uint32_t K, uint32_t sub_len) {
  const uint32_t i = blockIdx.y;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // = e * K + j
  if (t >= K * sub_len) return;
  const uint32_t e = t / K, j = t % K;
  cent_t[(size_t)i * sub_len * K + t] = cent[((size_t)i * K + j) * sub_len + e];
}
static int pq_fill_cdists(sdb_pq *pq, hipStream_t stream) {
  hipLaunchKernelGGL(k_pq_cdists, dim3(pq->K, pq->M), dim3(64), 0, stream, pq->d_centroids, pq->d_cdists, pq->M, pq->K,
                     pq->sub_len, pq->metric);
  hipLaunchKernelGGL(k_pq_transpose_cent, dim3((pq->K * pq->sub_len + 255) / 256, pq->M), dim3(256), 0, stream, pq->d_centroids,
                     pq->d_centroids_t, pq->K, pq->sub_len);
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
                   uint32_t *iters_out, int mem, int device, void *stream_) try {
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
  int rc = kmeans_device((float *)sx.dev, n, stride, offset, len, 1, K, max_iter, &first_idx, alias, (float *)sc.dev,
                         (uint8_t *)sl.dev, 1, iters_out, stream);
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
SDB_API_CATCH("sdb_kmeans_fit")

int sdb_pq_create(uint32_t dim, uint32_t metric, uint32_t num_subvectors, uint32_t num_centroids, int device,
                  sdb_pq **out) try {
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
  if (e == hipSuccess) e = hipMalloc(&pq->d_centroids_t, (size_t)pq->M * pq->K * pq->sub_len * 4);
  if (e == hipSuccess) e = hipMalloc(&pq->d_cdists, (size_t)pq->M * pq->K * pq->K * 4);
  if (e != hipSuccess) {
    if (pq->d_centroids) (void)hipFree(pq->d_centroids);
    if (pq->d_centroids_t) (void)hipFree(pq->d_centroids_t);
    delete pq;
    return fail(SDB_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  *out = pq;
  return SDB_OK;
}
SDB_API_CATCH("sdb_pq_create")

int sdb_pq_destroy(sdb_pq *pq) try {
  if (!pq) return SDB_OK;
  DeviceGuard dg(pq->device);
  (void)hipDeviceSynchronize();
  if (pq->d_centroids) (void)hipFree(pq->d_centroids);
  if (pq->d_centroids_t) (void)hipFree(pq->d_centroids_t);
  if (pq->d_cdists) (void)hipFree(pq->d_cdists);
  delete pq;
  return SDB_OK;
}
SDB_API_CATCH("sdb_pq_destroy")

int sdb_pq_fit(sdb_pq *pq, float *X, uint32_t n, const uint32_t *first_idx, int alias, uint8_t *codes_out, int mem,
               void *stream_) try {
  if (!pq || !X || !first_idx) return fail(SDB_ERR_INVALID, "NULL argument");
  if (n == 0) return fail(SDB_ERR_INVALID, "no vectors to fit");
  DeviceGuard dg(pq->device);
  hipStream_t stream = as_stream(stream_);
  Staged sx, sc;
  SDB_TRY(stage_in(sx, X, (size_t)n * pq->dim * 4, mem, stream));
  SDB_TRY(stage_in(sc, codes_out, codes_out ? (size_t)n * pq->M : 0, codes_out ? mem : SDB_MEM_HOST, stream, false));
  // one KMeans.Fit per sub-quantizer, all of them at once (one goroutine each in the reference, product.go:202-232);
  // the labels are the codes (:216-218)
  int rc = kmeans_device((float *)sx.dev, n, pq->dim, 0, pq->sub_len, pq->M, pq->K, 100, first_idx, alias, pq->d_centroids,
                         codes_out ? (uint8_t *)sc.dev : nullptr, pq->M, nullptr, stream);
  if (rc == SDB_OK) rc = pq_fill_cdists(pq, stream);
  if (rc == SDB_OK && codes_out) rc = stage_out(sc, stream);
  if (rc == SDB_OK && alias) rc = stage_out(sx, stream);
  (void)hipStreamSynchronize(stream);
  if (rc == SDB_OK) pq->fitted = true;
  return rc;
}
SDB_API_CATCH("sdb_pq_fit")

int sdb_pq_set_codebook(sdb_pq *pq, const float *flat_centroids, int mem) try {
  if (!pq || !flat_centroids) return fail(SDB_ERR_INVALID, "NULL argument");
  DeviceGuard dg(pq->device);
  SDB_HIP(hipMemcpy(pq->d_centroids, flat_centroids, (size_t)pq->M * pq->K * pq->sub_len * 4,
                    mem == SDB_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
  SDB_TRY(pq_fill_cdists(pq, nullptr));
  SDB_HIP(hipDeviceSynchronize());
  pq->fitted = true;
  return SDB_OK;
}
SDB_API_CATCH("sdb_pq_set_codebook")

int sdb_pq_get_codebook(const sdb_pq *pq, float *flat_centroids, float *centroid_dists) try {
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
SDB_API_CATCH("sdb_pq_get_codebook")

int sdb_pq_encode(const sdb_pq *pq, const float *vectors, uint64_t n, uint8_t *codes, int mem, void *stream_) try {
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
SDB_API_CATCH("sdb_pq_encode")

int sdb_pq_lut_distance(const sdb_pq *pq, const float *queries, uint64_t nq, const uint8_t *codes, uint64_t nc,
                        float *out, int mem, void *stream_) try {
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
SDB_API_CATCH("sdb_pq_lut_distance")

int sdb_pq_sym_distance(const sdb_pq *pq, const uint8_t *codes_x, const uint8_t *codes_y, uint64_t n, float *out,
                        int mem, void *stream_) try {
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
SDB_API_CATCH("sdb_pq_sym_distance")

}  // extern "C"
