// distance_tile.hip -- K1 with row reuse: out[q][c] = dist(queries[q], candidates[c]) for caller-supplied row-major
// blocks (distance.FloatDistFunc batched, distance/distance.go:70-83 over distance/asm/dot.s, euclidean.s).
//
// The first K1 kernel (distance.hip k_distance_batch) is one workgroup per (query, 64 candidates): every query
// re-reads the candidate matrix through L2 / Infinity Cache, 4 bytes per lane and load.  Here a workgroup owns 64
// candidate rows for the whole launch and walks ALL queries over them, so the candidate matrix is read from HBM once
// and the queries (small) are what is re-read:
//   dot / cosine   k_k1_tile_mfma   a wave keeps 16 candidate rows in registers (two 16-byte loads per lane and
//                  32-float block, straight from the caller's rows) and multiplies them with 16 queries at a time on
//                  the matrix cores.  v_mfma_f32_16x16x1_4b_f32 is one step of the reference's chain for 4 x 16 x 16
//                  accumulators: D = fma(A, B, D), one product, one rounding (tools/probes/mfma_exact.hip).  Partial
//                  sum L = 8 blk + k of pair (row i, query j) lives in accumulator set k, block blk -- L / 8 is the YMM
//                  register of dot.s:16-30, L % 8 its lane -- so a lane's eight operands per block are 32 contiguous
//                  bytes of the row, and the reduce tree of dot.s:45-53 is plain adds inside the lane.  Query operands
//                  arrive through LDS in operand order (k_k1_swizzle_queries, once per call), LDS-DMA, double buffered.
//   euclidean      k_k1_tile_pk     (x - y) is rounded per pair before the multiply (euclidean.s:27), which is not an
//                  outer product: up to 64 rows staged in LDS, lane r owns row r, query elements through the scalar cache,
//                  v_pk_fma_f32 on pairs of partial sums -- the exact scan's scheme (flat.hip k_flat_scan) on the
//                  caller's layout with the full distance block as output.
// Both write the [nq][nc] block with 64-byte (matrix) or 256-byte (packed) contiguous pieces per wave and query.
#include "dist_core.h"
#include "common.h"

#include <algorithm>

namespace sdb {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

constexpr uint32_t kK1Rows = 64;            // candidate rows per workgroup
constexpr uint32_t kK1TailPitch = 36;       // floats per row / query of tail elements in LDS (16-byte aligned, bank 4j)
constexpr uint32_t kK1TailImgFloats = 768;  // 16 x 36 = 576, rounded up to whole 1 KB pieces

// A query group's image: [b][h][l][c] = query 16 G + l % 16, element 32 b + 8 (l / 16) + 4 h + c for the nblk whole
// blocks; then, for rows with a tail, [j][kK1TailPitch]: the tail elements of query 16 G + j in order, zero padded.
__global__ void k_k1_swizzle_queries(const float *__restrict__ q, float *__restrict__ out, uint32_t nq, uint32_t dim,
                                     uint32_t nblk, uint32_t tail, uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t grp = nblk * 512 + (tail ? kK1TailImgFloats : 0);
  const uint32_t G = i / grp, o = i % grp;
  if (o >= nblk * 512) {
    const uint32_t t = o - nblk * 512, j = t / kK1TailPitch, m = t % kK1TailPitch, qi = 16 * G + j;
    out[i] = (j < 16 && m < tail && qi < nq) ? q[(size_t)qi * dim + 32 * nblk + m] : 0.0f;
    return;
  }
  const uint32_t c = o & 3, l = (o >> 2) & 63, h = (o >> 8) & 1, b = o >> 9;
  const uint32_t e = 32 * b + 8 * (l >> 4) + 4 * h + c, qi = 16 * G + (l & 15);
  out[i] = qi < nq ? q[(size_t)qi * dim + e] : 0.0f;
}

// NBLK: the row's whole blocks, exactly -- with a run-time count in the multiply loop the compiler copies the 128
// accumulators around its branches (300 moves per group of queries)
template <int NBLK, bool TAIL>
__global__ __launch_bounds__(256, (NBLK <= 12 && !TAIL) ? 2 : 1) void k_k1_tile_mfma(
    const float *__restrict__ cands, const float *__restrict__ qsw, float *__restrict__ out, uint64_t nc, uint32_t nq,
    uint32_t dim, uint32_t tail, int metric, int vec_store) {
  constexpr int NB = NBLK;
  constexpr uint32_t nblk = NBLK;
  constexpr uint32_t grp_f4 = nblk * 128 + (TAIL ? kK1TailImgFloats / 4 : 0);  // float4 per query group image
  constexpr int pieces = (int)(grp_f4 / 64);                                   // 1 KB pieces of it
  extern __shared__ __attribute__((aligned(16))) float bs[];  // [2][group image], then [64][kK1TailPitch] row tails
  float *rowtail = bs + 2 * (size_t)grp_f4 * 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint64_t row0 = (uint64_t)blockIdx.x * kK1Rows;
  // ---- the wave's 16 rows: lane 16 blk + i holds elements 32 b + 8 blk + (0..7) of row i for every block b
  f4v A[NB][2];
  {
    const uint64_t r = row0 + 16 * wave + (lane & 15);
    const float *src = cands + (size_t)(r < nc ? r : nc - 1) * dim + 8 * (lane >> 4);
#pragma unroll
    for (int b = 0; b < NB; b++) {
      A[b][0] = *reinterpret_cast<const f4v *>(src + 32 * b);
      A[b][1] = *reinterpret_cast<const f4v *>(src + 32 * b + 4);
    }
  }
  const uint32_t ngroups = (nq + 15) / 16;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto dma = [&](uint32_t G) __attribute__((always_inline)) {
    const char *src = reinterpret_cast<const char *>(qsw) + (size_t)min(G, ngroups - 1) * ((size_t)grp_f4 * 16);
    char *dst = reinterpret_cast<char *>(bs) + (size_t)(G & 1) * ((size_t)grp_f4 * 16);
    const uint32_t lane_off = lane * 16;
    for (int piece = wave; piece < pieces; piece += 4)
      __builtin_amdgcn_global_load_lds((glb_void *)(src + (size_t)piece * 1024 + lane_off), (lds_void *)(dst + (size_t)piece * 1024), 16, 0, 0);
  };
  auto multiply = [&](f16v (&acc)[8], f16v &T, uint32_t G) __attribute__((always_inline)) {
    const f4v *bq = reinterpret_cast<const f4v *>(bs) + (size_t)(G & 1) * grp_f4 + lane;
    if constexpr (TAIL) {  // the tail chains (dot.s:35-43): block 0 of the instruction, lanes 0..15 = row i / query j
#pragma unroll
      for (int r = 0; r < 16; r++) T[r] = 0.0f;
      const float *ta = rowtail + (16 * wave + (lane & 15)) * kK1TailPitch;
      const float *tb = bs + (size_t)(G & 1) * grp_f4 * 4 + (size_t)nblk * 512 + (lane & 15) * kK1TailPitch;
      const bool low = lane < 16;
      for (uint32_t m = 0; m < tail; m += 4) {
        const f4v x4 = *reinterpret_cast<const f4v *>(ta + m), y4 = *reinterpret_cast<const f4v *>(tb + m);
#pragma unroll
        for (int c = 0; c < 4; c++)
          if (m + c < tail)  // uniform: exactly `tail` steps, like the reference's loop
            T = __builtin_amdgcn_mfma_f32_16x16x1f32(low ? x4[c] : 0.0f, low ? y4[c] : 0.0f, T, 0, 0, 0);
      }
    }
    f4v b0 = bq[0], b1 = bq[64];
#pragma unroll
    for (int b = 0; b < NB; b++) {
      f4v n0 = b0, n1 = b1;
      if (b + 1 < NB) n0 = bq[(b + 1) * 128], n1 = bq[(b + 1) * 128 + 64];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        f16v c = acc[k];
        if (b == 0)
#pragma unroll
          for (int r = 0; r < 16; r++) c[r] = 0.0f;
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x1f32(A[b][k >> 2][k & 3], k < 4 ? b0[k & 3] : b1[k & 3], c, 0, 0, 0);
      }
      b0 = n0, b1 = n1;
    }
  };
  // dot.s:45-53 in the lane: s[k] = ((P[0][k] + P[1][k]) + P[2][k]) + P[3][k]; r[l] = s[l] + s[l + 4]; r[0] += t;
  // result = (r[0] + r[1]) + (r[2] + r[3]).  P[blk][k] = register 4 blk + i4 of accumulator set k.
  auto reduce = [&](const f16v (&acc)[8], const f16v &T, float (&dist)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i4 = 0; i4 < 4; i4++) {
      float s[8];
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = ((acc[k][i4] + acc[k][4 + i4]) + acc[k][8 + i4]) + acc[k][12 + i4];
      float r0 = s[0] + s[4], r1 = s[1] + s[5], r2 = s[2] + s[6], r3 = s[3] + s[7];
      if constexpr (TAIL) r0 = r0 + T[i4];
      else r0 = r0 + 0.0f;
      r1 = r1 + 0.0f, r2 = r2 + 0.0f, r3 = r3 + 0.0f;  // VADDPS with the {t, 0, 0, 0} vector (dot.s:51)
      dist[i4] = (r0 + r1) + (r2 + r3);
    }
  };
  const bool cosine = metric == SDB_METRIC_COSINE;
  const uint64_t c_lane = row0 + 16 * wave + 4 * (lane >> 4);  // the lane emits rows c_lane .. c_lane + 3
  auto emit = [&](const float (&dist)[4], uint32_t G) __attribute__((always_inline)) {
    const uint32_t q = 16 * G + (lane & 15);
    if (q >= nq || c_lane >= nc) return;
    float d[4];
#pragma unroll
    for (int i4 = 0; i4 < 4; i4++) d[i4] = cosine ? 1.0f - dist[i4] : -dist[i4];  // distance.go:19-25
    float *o = out + (size_t)q * nc + c_lane;
    if (vec_store && c_lane + 3 < nc) {
      *reinterpret_cast<f4v *>(o) = f4v{d[0], d[1], d[2], d[3]};
    } else {
#pragma unroll
      for (int i4 = 0; i4 < 4; i4++)
        if (c_lane + i4 < nc) o[i4] = d[i4];
    }
  };
  dma(0);
  if constexpr (TAIL) {
    for (uint32_t i = tid; i < kK1Rows * 32; i += 256) {
      const uint32_t r = i >> 5, m = i & 31;
      const uint64_t row = row0 + r;
      rowtail[r * kK1TailPitch + m] = m < tail ? cands[(size_t)(row < nc ? row : nc - 1) * dim + 32 * nblk + m] : 0.0f;
    }
    for (uint32_t i = tid; i < kK1Rows * 4; i += 256) rowtail[(i >> 2) * kK1TailPitch + 32 + (i & 3)] = 0.0f;
  }
  __syncthreads();  // waits for this wave's DMAs (vmcnt) and for everybody else's
  f16v acc[8], T;
  float dist[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  // a group's distances are stored at the start of the next group's multiply (see k_k1_stream_mfma)
  for (uint32_t G = 0; G < ngroups; G++) {
    dma(G + 1);
    if (G) emit(dist, G - 1);
    multiply(acc, T, G);
    reduce(acc, T, dist);
    __syncthreads();
  }
  emit(dist, ngroups - 1);
}

// ---- few queries (up to a few hundred): the roles turned around ------------------------------------------------------
// k_k1_tile_mfma keeps 64 candidate rows in registers and streams the query groups: with 64 queries that is four groups
// per workgroup, and a workgroup's life is mostly the load of its rows (77 G pairs/s at 64 x 1M x 384, 0.38 of the
// matrix pipe).  Here a wave keeps 16 QUERIES in registers for the whole launch and the candidates stream through LDS,
// 16 rows per group, double buffered: LDS-DMA takes the operand layout straight from the caller's row-major rows -- the
// destination of a wave's piece is lane-linear, the source is per lane, and lane l's 16 bytes of piece (b, h) are
// elements 32 b + 8 (l / 16) + 4 h .. + 3 of candidate 16 G + l % 16, exactly the [b][h][l][c] image k_k1_swizzle_queries
// writes for the other kernel.  Same instruction, same chains, same reduce tree: the same bits.  A workgroup (four
// waves = 64 queries; blockIdx.x counts query blocks, blockIdx.y spans) walks `span` candidates.
template <int NBLK>
__global__ __launch_bounds__(256, NBLK <= 12 ? 2 : 1) void k_k1_stream_mfma(const float *__restrict__ queries,
                                                                          const float *__restrict__ cands,
                                                                          float *__restrict__ out, uint64_t nc, uint32_t nq,
                                                                          uint32_t dim, uint32_t span, int metric) {
  constexpr int NB = NBLK;  // the row's blocks, exactly: a run-time block count in the multiply loop makes the compiler
                            // copy the 128 accumulators around its branches (300 moves per group)
  constexpr uint32_t nblk = NBLK;
  constexpr uint32_t grp_f4 = nblk * 128;  // float4 per candidate group image
  extern __shared__ __attribute__((aligned(16))) float bs[];  // [2][group image]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t q0 = 64 * blockIdx.x + 16 * wave;  // the query blocks of a span follow each other: its rows come from HBM once
  // ---- the wave's 16 queries: lane 16 blk + i holds elements 32 b + 8 blk + (0..7) of query i for every block b
  f4v A[NB][2];
  {
    const uint32_t r = q0 + (lane & 15);
    const float *src = queries + (size_t)(r < nq ? r : nq - 1) * dim + 8 * (lane >> 4);
#pragma unroll
    for (int b = 0; b < NB; b++) {
      A[b][0] = *reinterpret_cast<const f4v *>(src + 32 * b);
      A[b][1] = *reinterpret_cast<const f4v *>(src + 32 * b + 4);
    }
  }
  const uint64_t c0 = (uint64_t)blockIdx.y * span;
  const uint64_t c_end = c0 + span < nc ? c0 + span : nc;
  const uint32_t ngroups = (uint32_t)((c_end - c0 + 15) / 16);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  constexpr int pieces = 2 * NBLK;
  // wave w takes pieces w, w + 4, ..: piece (b, h) = 2 b + h starts 128 b + 16 h bytes into the lane's 8 floats per block,
  // so the wave's pieces are 256 bytes apart from a start of its own -- immediates, one address register pair
  const uint32_t wave_off = 32u * (uint32_t)(wave >> 1) + 4u * (uint32_t)(wave & 1);  // floats
  auto dma = [&](uint32_t G) __attribute__((always_inline)) {
    const uint64_t row = c0 + 16 * (uint64_t)min(G, ngroups - 1) + (uint32_t)(lane & 15);
    const float *src = cands + (size_t)(row < nc ? row : nc - 1) * dim + 8 * (lane >> 4) + wave_off;
    char *dst = reinterpret_cast<char *>(bs) + (size_t)(G & 1) * ((size_t)grp_f4 * 16) + (size_t)wave * 1024;
#pragma unroll
    for (int i = 0; 4 * i < pieces; i++)
      if (4 * i + 3 < pieces || wave + 4 * i < pieces)
        __builtin_amdgcn_global_load_lds((glb_void *)(src + 64 * i), (lds_void *)(dst + (size_t)i * 4096), 16, 0, 0);
  };
  auto multiply = [&](f16v (&acc)[8], uint32_t G) __attribute__((always_inline)) {
    const f4v *bq = reinterpret_cast<const f4v *>(bs) + (size_t)(G & 1) * grp_f4 + lane;
    f4v b0 = bq[0], b1 = bq[64];
#pragma unroll
    for (int b = 0; b < NB; b++) {
      f4v n0 = b0, n1 = b1;
      if (b + 1 < NB) n0 = bq[(b + 1) * 128], n1 = bq[(b + 1) * 128 + 64];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        f16v c = acc[k];
        if (b == 0)
#pragma unroll
          for (int r = 0; r < 16; r++) c[r] = 0.0f;
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x1f32(A[b][k >> 2][k & 3], k < 4 ? b0[k & 3] : b1[k & 3], c, 0, 0, 0);
      }
      b0 = n0, b1 = n1;
    }
  };
  // dot.s:45-53 in the lane (see k_k1_tile_mfma): P[blk][k] = register 4 blk + i4 of accumulator set k
  auto reduce = [&](const f16v (&acc)[8], float (&dist)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i4 = 0; i4 < 4; i4++) {
      float s[8];
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = ((acc[k][i4] + acc[k][4 + i4]) + acc[k][8 + i4]) + acc[k][12 + i4];
      float r0 = s[0] + s[4], r1 = s[1] + s[5], r2 = s[2] + s[6], r3 = s[3] + s[7];
      r0 = r0 + 0.0f, r1 = r1 + 0.0f, r2 = r2 + 0.0f, r3 = r3 + 0.0f;  // VADDPS with the {t, 0, 0, 0} vector (dot.s:51)
      dist[i4] = (r0 + r1) + (r2 + r3);
    }
  };
  const bool cosine = metric == SDB_METRIC_COSINE;
  const uint32_t q_lane = q0 + 4 * (lane >> 4);  // the lane emits queries q_lane .. q_lane + 3 (the A side of the tile)
  auto emit = [&](const float (&dist)[4], uint32_t G) __attribute__((always_inline)) {
    const uint64_t c = c0 + 16 * (uint64_t)G + (uint32_t)(lane & 15);  // 16 lanes: 64 contiguous bytes of a query's row
    if (c >= c_end) return;
#pragma unroll
    for (int i4 = 0; i4 < 4; i4++)
      if (q_lane + i4 < nq) out[(size_t)(q_lane + i4) * nc + c] = cosine ? 1.0f - dist[i4] : -dist[i4];  // distance.go:19-25
  };
  dma(0);
  __syncthreads();  // waits for this wave's DMAs (vmcnt) and for everybody else's
  f16v acc[8];
  float dist[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  // a group's distances are stored at the start of the NEXT group's multiply: the barrier's vmcnt(0) -- it has to wait for
  // the LDS-DMA -- would otherwise also wait for stores issued a moment before it, every group.  (A ring of three
  // buffers with two groups' DMAs in flight behind counted vmcnt waits and bare barriers measured the same: 0.69 ms at
  // 64 x 1M x 384 either way; what is left is the two waves of a SIMD not overlapping, matrix pipe 64 % busy at 1.75 GHz.)
  for (uint32_t G = 0; G < ngroups; G++) {
    dma(G + 1);
    if (G) emit(dist, G - 1);
    multiply(acc, G);
    reduce(acc, dist);
    __syncthreads();
  }
  emit(dist, ngroups - 1);
}

// ---- euclidean: packed FMAs over an LDS tile of 64 rows (original element order, padded by 16 B per row) --------
template <bool L2>
__device__ __forceinline__ f2v k1_chain_pk(f2v acc, f2v x, f2v y) {
  if constexpr (L2) {
    const f2v d = x - y;  // separately rounded, like VSUBPS (euclidean.s:27)
    return __builtin_elementwise_fma(d, d, acc);
  } else {
    return __builtin_elementwise_fma(x, y, acc);
  }
}
// waves per workgroup x queries per pass, measured on 1 024 x 1M x 384 / 64 x 1M x 384 (ms).  With a fixed share of the
// query groups per wave: 16 x 2 31.0 / 1.81; 12 x 3 37.7 / 2.27; 12 x 4 (168 VGPRs, 12 B of scratch) 33.7 / 2.49; 8 x 4
// (169 VGPRs, two waves per SIMD, each row block read once per four queries) 23.3 / 1.81.  With the groups handed out from
// a counter in LDS (SDB_K1_DYN, as in the exact scan): 16 x 1 **17.9 / 1.48** (shipped), 8 x 4 21.4 / 1.73, 16 x 2
// 28.2 / 1.73 -- with two queries per pass the compiler reuses one set of scalar registers for the queries' loads and
// waits for each (seven s_waitcnt per block instead of one)
#ifndef SDB_K1_DYN
#define SDB_K1_DYN 1
#endif
#ifndef SDB_K1_WAVES
#define SDB_K1_WAVES 16
#define SDB_K1_QPP 1
#endif
constexpr int kK1L2Waves = SDB_K1_WAVES;
constexpr int kK1Qpp = SDB_K1_QPP;
template <bool L2>
__global__ __launch_bounds__(kK1L2Waves * 64) void k_k1_tile_pk(const float *__restrict__ cands,
                                                                const float *__restrict__ queries,
                                                                float *__restrict__ out, uint64_t nc, uint32_t nq,
                                                                uint32_t dim, uint32_t nblk, uint32_t tail,
                                                                uint32_t tile_rows, int metric) {
  // [tile_rows][kstride]: as many rows as fit LDS, at most one per lane (64 up to d = 608, 52 at 768, 39 at 1024)
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const uint32_t kstride = dim + 4;  // +16 B per row: lane r starts at bank 4r, the lanes' ds_read_b128 never collide
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint64_t row0 = (uint64_t)blockIdx.x * tile_rows;
  const uint32_t nrows = (uint32_t)min<uint64_t>(tile_rows, nc - row0);
  const uint32_t row_f4 = dim / 4;
#if SDB_K1_DYN
  if (tid == 0) *reinterpret_cast<uint32_t *>(tile + (size_t)tile_rows * kstride) = 0;
#endif
  for (uint32_t i = tid; i < tile_rows * row_f4; i += kK1L2Waves * 64) {  // consecutive threads: consecutive 16 B of a row
    const uint32_t r = i / row_f4, c = i % row_f4;
    const uint32_t rr = r < nrows ? r : nrows - 1;
    *reinterpret_cast<float4 *>(tile + (size_t)r * kstride + 4 * c) =
        reinterpret_cast<const float4 *>(cands + (size_t)(row0 + rr) * dim)[c];
  }
  __syncthreads();
  const float *myrow_f = tile + (size_t)((uint32_t)lane < tile_rows ? lane : 0) * kstride;  // idle lanes: row 0, dropped
  const float4 *myrow = reinterpret_cast<const float4 *>(myrow_f);
  const uint32_t ngroups = (nq + kK1Qpp - 1) / kK1Qpp;
#if SDB_K1_DYN
  float *next_group = tile + (size_t)tile_rows * kstride;  // the waves' work counter, behind the tile (flat.hip k_flat_scan)
  (void)wave;
  uint32_t dim_u = dim;  // cut off from the phi that the staging loop's divergent exit makes of zext(dim): scalar loads stay scalar
  asm volatile("" : "+s"(dim_u));
  for (;;) {
    const uint32_t grp = next_query_group(next_group, lane);
    if (grp >= ngroups) break;
#else
  const uint32_t dim_u = dim;
  for (uint32_t grp = (uint32_t)wave; grp < ngroups; grp += kK1L2Waves) {
#endif
    uniform_float *xq[kK1Qpp];
#pragma unroll
    for (int k = 0; k < kK1Qpp; k++) {
      const uint32_t q = grp * kK1Qpp + k;
      xq[k] = as_uniform(queries) + (size_t)(q < nq ? q : nq - 1) * dim_u;  // past the end: the last query again, dropped
    }
    f2v acc[kK1Qpp][16];
#pragma unroll
    for (int k = 0; k < kK1Qpp; k++)
#pragma unroll
      for (int j = 0; j < 16; j++) acc[k][j] = f2v{0.0f, 0.0f};
#pragma unroll 1
    for (uint32_t b = 0; b < nblk; b++) {
      float4 y[8];
#pragma unroll
      for (int i = 0; i < 8; i++) y[i] = myrow[b * 8 + i];
#pragma unroll
      for (int i = 0; i < 8; i++) {  // float4 i of the block: partial sums 4i .. 4i + 3
#pragma unroll
        for (int k = 0; k < kK1Qpp; k++) {
          const uniform_f4v u = reinterpret_cast<uniform_float4 *>(xq[k] + b * 32)[i];  // wave-uniform: scalar loads
          acc[k][2 * i] = k1_chain_pk<L2>(acc[k][2 * i], f2v{u.x, u.y}, f2v{y[i].x, y[i].y});
          acc[k][2 * i + 1] = k1_chain_pk<L2>(acc[k][2 * i + 1], f2v{u.z, u.w}, f2v{y[i].z, y[i].w});
        }
      }
    }
    // the tail chain (dot.s:35-43 / euclidean.s:44-53): the n % 32 last elements, one after the other
    float t[kK1Qpp];
#pragma unroll
    for (int k = 0; k < kK1Qpp; k++) t[k] = 0.0f;
    for (uint32_t m = 0; m < tail; m++) {
      const float yv = myrow_f[nblk * 32 + m];
#pragma unroll
      for (int k = 0; k < kK1Qpp; k++) t[k] = chain1<L2>(t[k], xq[k][nblk * 32 + m], yv);
    }
#pragma unroll
    for (int k = 0; k < kK1Qpp; k++) {
      auto A = [&](int L) { return acc[k][L >> 1][L & 1]; };
      float r4[4];
#pragma unroll
      for (int l = 0; l < 4; l++) {
        const float s0 = ((A(l) + A(8 + l)) + A(16 + l)) + A(24 + l);
        const float s1 = ((A(l + 4) + A(12 + l)) + A(20 + l)) + A(28 + l);
        r4[l] = (s0 + s1) + (l == 0 ? t[k] : 0.0f);  // + {t, 0, 0, 0} (dot.s:51)
      }
      const float dist = metric_finish((r4[0] + r4[1]) + (r4[2] + r4[3]), metric);
      const uint32_t q = grp * kK1Qpp + k;
      if (q < nq && (uint32_t)lane < nrows) out[(size_t)q * nc + row0 + lane] = dist;  // 256 contiguous bytes per wave
    }
  }
}

static int k1_mfma_launch(bool tail_k, dim3 grid, size_t lds, hipStream_t stream, const float *dc, const float *qsw,
                          float *dout, uint64_t nc, uint32_t nq, uint32_t dim, uint32_t nblk, uint32_t tail, int metric,
                          int vec_store) {
#define SDB_K1_CASE(N)                                                                                              \
  case N: {                                                                                                         \
    static std::atomic<uint64_t> at0{0}, at1{0};                                                                    \
    if (tail_k) {                                                                                                   \
      if (first_use_on_this_device(at1))                                                                            \
        SDB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_k1_tile_mfma<N, true>),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                       \
      hipLaunchKernelGGL((k_k1_tile_mfma<N, true>), grid, dim3(256), lds, stream, dc, qsw, dout, nc, nq, dim, tail, \
                         metric, vec_store);                                                                        \
    } else {                                                                                                        \
      if (first_use_on_this_device(at0))                                                                            \
        SDB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_k1_tile_mfma<N, false>),                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                       \
      hipLaunchKernelGGL((k_k1_tile_mfma<N, false>), grid, dim3(256), lds, stream, dc, qsw, dout, nc, nq, dim,      \
                         tail, metric, vec_store);                                                                  \
    }                                                                                                               \
    break;                                                                                                          \
  }
  switch (nblk) {
    SDB_K1_CASE(1) SDB_K1_CASE(2) SDB_K1_CASE(3) SDB_K1_CASE(4) SDB_K1_CASE(5) SDB_K1_CASE(6) SDB_K1_CASE(7) SDB_K1_CASE(8)
    SDB_K1_CASE(9) SDB_K1_CASE(10) SDB_K1_CASE(11) SDB_K1_CASE(12) SDB_K1_CASE(13) SDB_K1_CASE(14) SDB_K1_CASE(15)
    SDB_K1_CASE(16) SDB_K1_CASE(17) SDB_K1_CASE(18) SDB_K1_CASE(19) SDB_K1_CASE(20) SDB_K1_CASE(21) SDB_K1_CASE(22)
    SDB_K1_CASE(23) SDB_K1_CASE(24) SDB_K1_CASE(25) SDB_K1_CASE(26) SDB_K1_CASE(27) SDB_K1_CASE(28) SDB_K1_CASE(29)
    SDB_K1_CASE(30) SDB_K1_CASE(31) SDB_K1_CASE(32)
    default: return fail(SDB_ERR_INVALID, "row too long for the matrix-core tile");
  }
#undef SDB_K1_CASE
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}

// 1: ran on the tile kernels; 0: shape not covered (the caller takes k_distance_batch); < 0: error (-status)
int launch_k1_tiles(int metric, uint32_t dim, const float *dq, uint64_t nq, const float *dc, uint64_t nc, float *dout,
                    hipStream_t stream) {
  const uint32_t nblk = dim / 32, tail = dim % 32;
  if (nq < 2 || nblk < 1 || (dim & 3)) return 0;  // one query reuses nothing; rows must be whole float4s
  if ((reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dc)) & 15) return 0;
  const dim3 grid((unsigned)((nc + kK1Rows - 1) / kK1Rows));
  if (metric == SDB_METRIC_EUCLIDEAN) {
    // as many rows per workgroup as fit LDS, at most one per lane
    const size_t row_bytes = (size_t)(dim + 4) * sizeof(float);
    const uint32_t tile_rows = (uint32_t)std::min<size_t>(kK1Rows, (160 * 1024 - 16) / row_bytes);
    if (tile_rows < 8) return 0;
    const size_t lds = tile_rows * row_bytes + 16;  // the tile and the group counter
    static std::atomic<uint64_t> attr{0};
    if (first_use_on_this_device(attr))
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_k1_tile_pk<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024) != hipSuccess)
        return -fail(SDB_ERR_DEVICE, "hipFuncSetAttribute failed");
    hipLaunchKernelGGL((k_k1_tile_pk<true>), dim3((unsigned)((nc + tile_rows - 1) / tile_rows)), dim3(kK1L2Waves * 64), lds,
                       stream, dc, dq, dout, nc, (uint32_t)nq, dim, nblk, tail, tile_rows, metric);
    if (hipGetLastError() != hipSuccess) return -fail(SDB_ERR_DEVICE, "k_k1_tile_pk launch failed");
    return 1;
  }
  if (nblk > 32) return 0;
#ifndef SDB_K1_STREAM_MAX_NQ
#define SDB_K1_STREAM_MAX_NQ 256
#endif
  if (tail == 0 && nblk <= 32 && nq <= SDB_K1_STREAM_MAX_NQ) {  // few queries: they stay in registers, the candidates stream
    uint32_t span = 512;
    while ((nc + span - 1) / span > 65535) span *= 2;  // grid.y
    const dim3 sgrid((unsigned)((nq + 63) / 64), (unsigned)((nc + span - 1) / span));
    const size_t slds = (size_t)2 * nblk * 2048;
#define SDB_K1_STREAM(N)                                                                                               \
  case N: {                                                                                                            \
    static std::atomic<uint64_t> at{0};                                                                                \
    if (first_use_on_this_device(at) &&                                                                                \
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_k1_stream_mfma<N>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            160 * 1024) != hipSuccess)                                                                 \
      return -fail(SDB_ERR_DEVICE, "hipFuncSetAttribute failed");                                                      \
    hipLaunchKernelGGL((k_k1_stream_mfma<N>), sgrid, dim3(256), slds, stream, dq, dc, dout, nc, (uint32_t)nq, dim, span, \
                       metric);                                                                                        \
    break;                                                                                                             \
  }
    switch (nblk) {
      SDB_K1_STREAM(1) SDB_K1_STREAM(2) SDB_K1_STREAM(3) SDB_K1_STREAM(4) SDB_K1_STREAM(5) SDB_K1_STREAM(6) SDB_K1_STREAM(7)
      SDB_K1_STREAM(8) SDB_K1_STREAM(9) SDB_K1_STREAM(10) SDB_K1_STREAM(11) SDB_K1_STREAM(12) SDB_K1_STREAM(13)
      SDB_K1_STREAM(14) SDB_K1_STREAM(15) SDB_K1_STREAM(16) SDB_K1_STREAM(17) SDB_K1_STREAM(18) SDB_K1_STREAM(19)
      SDB_K1_STREAM(20) SDB_K1_STREAM(21) SDB_K1_STREAM(22) SDB_K1_STREAM(23) SDB_K1_STREAM(24) SDB_K1_STREAM(25)
      SDB_K1_STREAM(26) SDB_K1_STREAM(27) SDB_K1_STREAM(28) SDB_K1_STREAM(29) SDB_K1_STREAM(30) SDB_K1_STREAM(31)
      SDB_K1_STREAM(32)
    }
#undef SDB_K1_STREAM
    if (hipGetLastError() != hipSuccess) return -fail(SDB_ERR_DEVICE, "k_k1_stream_mfma launch failed");
    return 1;
  }
  const uint32_t ngroups = (uint32_t)((nq + 15) / 16);
  const uint32_t grp_floats = nblk * 512 + (tail ? kK1TailImgFloats : 0);
  const size_t lds = (size_t)2 * grp_floats * 4 + (tail ? (size_t)kK1Rows * kK1TailPitch * 4 : 0);
  if (lds > 160 * 1024) return 0;
  const uint64_t total = (uint64_t)ngroups * grp_floats;
  if (total > 0xFFFFFFFFull) return 0;
  // the queries in operand order: stream-ordered scratch (the pool keeps it for the next call)
  float *qsw = nullptr;
  if (hipMallocAsync(reinterpret_cast<void **>(&qsw), total * 4, stream) != hipSuccess)
    return -fail(SDB_ERR_DEVICE, "out of device memory for the query operands");
  hipLaunchKernelGGL(k_k1_swizzle_queries, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dq, qsw,
                     (uint32_t)nq, dim, nblk, tail, (uint32_t)total);
  const int vec_store = ((nc & 3) == 0 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0) ? 1 : 0;
  int rc = k1_mfma_launch(tail != 0, grid, lds, stream, dc, qsw, dout, nc, (uint32_t)nq, dim, nblk,
                          tail, metric, vec_store);
  (void)hipFreeAsync(qsw, stream);
  return rc == SDB_OK ? 1 : -rc;
}

}  // namespace sdb
