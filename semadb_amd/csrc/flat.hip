// flat.hip -- IndexFlat.Search (shard/index/flat/flat.go:76-132): exact scan of the vector store with the
// same distance closure the graph index uses (vecStore.DistanceFromFloat).  SURVEY 8f-4: it doubles as the
// exact-kNN ground truth for recall.
//
// The reference walks ItemCache.ForEach -- Go map order, i.e. unspecified -- and keeps the `limit`
// closest points with `dist >= tail -> skip` (:104) and a stable bubble (:121-123).  Here the walk is in
// slot order, which is one of the orders the reference may take: among equal distances the first seen
// stays.  Phase 1 computes the [nq][chunk] distance block (bit-identical arithmetic, dist_core.h);
// phase 2 folds it into the per-query top list, one wavefront per query.
#include <cfloat>

// streaming scan shape: waves per workgroup x query pairs per pass, 1 024 x 1M x 384 euclidean.  Query groups handed
// out from a counter (SDB_SCAN_DYN): 16 x 1 (four waves per SIMD) 16.2 - 17.3 ms depending on the box, 12 x 1 17.4,
// 8 x 2 (half the LDS reads per FMA, two waves per SIMD) 22.5, 8 x 1 23.0; with a fixed share per wave 16 x 1 took
// 18.9 - 19.5 ms (3.1 of 4 wave slots occupied on average: the SIMD favours its oldest wave, the youngest finishes alone)
#ifndef SDB_SCAN_DYN
#define SDB_SCAN_DYN 1
#endif
#ifndef SDB_SCAN_WAVES
#define SDB_SCAN_WAVES 16
#define SDB_SCAN_PAIRS 1
#endif

#include "pq.h"
#include "search_kernel.h"

namespace sdb {

void keep_pool_memory(int device);  // distance.hip

// ---- phase 1: distances of `rows` slab rows (first_row ..) or of listed slots to every query -------------
// grid (ceil(rows / 64), nq), block 256 = 8 half-waves, 8 candidates each
template <bool L2>
__global__ __launch_bounds__(256) void k_flat_dist(const float *__restrict__ slab, const float *__restrict__ queries,
                                                   const uint32_t *__restrict__ slots, const uint32_t *__restrict__ slot_off,
                                                   uint32_t first_row, uint32_t rows, float *__restrict__ out,
                                                   uint32_t out_stride, uint32_t dim, uint32_t nblk, uint32_t ng,
                                                   uint32_t tail, uint32_t ld, int metric) {
  extern __shared__ __attribute__((aligned(16))) float qs[];
  const uint32_t q = blockIdx.y;
  const float *qv = queries + (size_t)q * dim;
  for (uint32_t i = threadIdx.x; i < ng * 128; i += blockDim.x) {
    uint32_t g = i / 128, r = i % 128;
    qs[i] = q_elem(qv, nblk, g, r % 4, (int)(r / 4));
  }
  if (tail && threadIdx.x < 32) qs[ng * 128 + threadIdx.x] = threadIdx.x < tail ? qv[nblk * 32 + threadIdx.x] : 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, L = lane & 31;
  const int hw = threadIdx.x >> 5;
  // filtered: this query's candidate list is slots[slot_off[q] .. slot_off[q+1])
  uint32_t nrows = rows;
  const uint32_t *myslots = nullptr;
  if (slots) {
    myslots = slots + slot_off[q];
    nrows = slot_off[q + 1] - slot_off[q];
  }
  constexpr int U = 4;
  const uint32_t c0 = blockIdx.x * 64 + hw * 8;  // this half-wave's 8 candidates; the wave handles 16 = 2*U*... pairs
  // a wave = two half-waves -> pairs (c0 + u) for half 0 and half 1 handled by their own half: use chunk_dist_lds
  // with per-lane slots: lanes of half h take candidate c0 + u (their own c0)
  uint32_t slot[U], slot2[U];
  bool live[U], live2[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const uint32_t c = c0 + u, c2 = c0 + U + u;
    live[u] = c < nrows, live2[u] = c2 < nrows;
    // a candidate past the end of the list is computed and dropped: on row 0, never on whatever follows the list
    // (a query whose filter names no stored id has an EMPTY list; found by tools/fuzz_parity.py, seed 31337 trial 233)
    slot[u] = live[u] ? (myslots ? myslots[c] : first_row + c) : 0u;
    slot2[u] = live2[u] ? (myslots ? myslots[c2] : first_row + c2) : 0u;
  }
  float res[U], res2[U];
  chunk_dist_lds<L2, U>(slab, ld, ng, tail, qs, slot, res, lane);
  chunk_dist_lds<L2, U>(slab, ld, ng, tail, qs, slot2, res2, lane);
#pragma unroll
  for (int u = 0; u < U; u++) {
    if (L == 0 && live[u]) out[(size_t)q * out_stride + c0 + u] = metric_finish(res[u], metric);
    if (L == 0 && live2[u]) out[(size_t)q * out_stride + c0 + U + u] = metric_finish(res2[u], metric);
  }
}

// phase 1 over a quantized store: vecStore.DistanceFromFloat is the LUT distance (product.go:250-277),
// sum over the sub-quantizers in index order.  Thread per row, the query's table read through the cache.
__global__ __launch_bounds__(256) void k_flat_dist_pq(const float *__restrict__ lut, const uint8_t *__restrict__ codes,
                                                      const uint32_t *__restrict__ slots,
                                                      const uint32_t *__restrict__ slot_off, uint32_t first_row,
                                                      uint32_t rows, float *__restrict__ out, uint32_t out_stride,
                                                      uint32_t M, uint32_t K) {
  const uint32_t q = blockIdx.y;
  uint32_t nrows = rows;
  const uint32_t *myslots = nullptr;
  if (slots) {
    myslots = slots + slot_off[q];
    nrows = slot_off[q + 1] - slot_off[q];
  }
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nrows) return;
  const uint32_t slot = myslots ? myslots[c] : first_row + c;
  const float *l = lut + (size_t)q * M * K;
  const uint8_t *cd = codes + (size_t)slot * M;
  float dist = 0.0f;
  for (uint32_t i = 0; i < M; i++) dist += l[i * K + cd[i]];
  out[(size_t)q * out_stride + c] = dist;
}

// ---- phase 2: fold a distance block into the running top list (flat.go:98-124), one wave per query -------
struct FlatFoldArgs {
  const float *dists;       // [nq][stride]
  uint32_t stride;
  uint32_t count;           // values per query in this block (unfiltered)
  const uint32_t *slot_off; // filtered: per-query counts come from here
  const uint32_t *slots;    // filtered: slot of value i of query q
  uint32_t first_row;       // unfiltered: slot of value i is first_row + i
  uint32_t skip_slot;       // the graph's start node never belongs to a flat result (it is not a point)
  const uint64_t *ids;      // slot -> id, 0 = deleted
  uint32_t limit;
  uint32_t *top_slot;       // [nq][128] running state
  float *top_dist;
  uint32_t *top_len;        // [nq]
};

__global__ __launch_bounds__(64) void k_flat_fold(const FlatFoldArgs a) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  uint32_t cid[2];
  float cd[2];
  cid[0] = a.top_slot[(size_t)q * 128 + lane], cid[1] = a.top_slot[(size_t)q * 128 + 64 + lane];
  cd[0] = a.top_dist[(size_t)q * 128 + lane], cd[1] = a.top_dist[(size_t)q * 128 + 64 + lane];
  int len = (int)a.top_len[q];
  const int cap = (int)a.limit;
  uint32_t n = a.count;
  const uint32_t *myslots = nullptr;
  if (a.slots) {
    myslots = a.slots + a.slot_off[q];
    n = a.slot_off[q + 1] - a.slot_off[q];
  }
  const float *d = a.dists + (size_t)q * a.stride;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t i = base + lane;
    const bool has = i < n;
    const float dist = has ? d[i] : 0.0f;
    const uint32_t slot = has ? (myslots ? myslots[i] : a.first_row + i) : kNoSlot;
    // deleted rows (tombstones, id 0) are not in the store any more
    uint64_t pd = __ballot(has && slot != a.skip_slot && a.ids[slot] != 0);
    while (pd) {
      bool ok = true;
      // :104 `dist >= tail -> skip`, literally: a NaN on either side compares false, so a NaN row REPLACES the tail of
      // a full list and any row replaces a NaN tail -- the walk in storage order decides, and this loop is that walk
      if (len == cap) ok = !(dist >= list_tail(cd, cap));
      const uint64_t am = __ballot(ok) & pd;
      if (!am) break;
      const int j = __ffsll((unsigned long long)am) - 1;
      const float dj = rlf(dist, j);
      const uint32_t sj = rl(slot, j);
      pd = (j == 63) ? 0ull : ((pd >> (j + 1)) << (j + 1));
      list_insert(cid, cd, len, cap, sj, dj, lane);  // :116-123 append or overwrite the tail, bubble with '<'
    }
  }
  a.top_slot[(size_t)q * 128 + lane] = cid[0], a.top_slot[(size_t)q * 128 + 64 + lane] = cid[1];
  a.top_dist[(size_t)q * 128 + lane] = cd[0], a.top_dist[(size_t)q * 128 + 64 + lane] = cd[1];
  if (lane == 0) a.top_len[q] = (uint32_t)len;
}

__global__ void k_flat_emit(const uint32_t *__restrict__ top_slot, const float *__restrict__ top_dist,
                            const uint32_t *__restrict__ top_len, const uint64_t *__restrict__ ids, uint32_t limit,
                            uint64_t *__restrict__ out_ids, float *__restrict__ out_dists, uint32_t *__restrict__ out_counts) {
  const uint32_t q = blockIdx.x;
  const uint32_t len = top_len[q];
  for (uint32_t i = threadIdx.x; i < limit; i += blockDim.x) {
    if (i < len) {
      out_ids[(size_t)q * limit + i] = ids[top_slot[(size_t)q * 128 + i]];
      out_dists[(size_t)q * limit + i] = top_dist[(size_t)q * 128 + i];
    } else {  // short rows end in zeros
      out_ids[(size_t)q * limit + i] = 0;
      out_dists[(size_t)q * limit + i] = 0.0f;
    }
  }
  if (threadIdx.x == 0) out_counts[q] = len;
}


// ---- the streaming scan: every slab row read ONCE, against all queries ---------------------------------------
// k_flat_dist above recomputes nothing but re-reads everything: a block is one query x 64 rows, so the slab
// streams through the cache hierarchy once per query (1.5 TB for 1 024 queries over 1M x 384) and the
// [nq][chunk] distance block makes a round trip through HBM before the fold.  k_flat_scan turns the loop around:
// a workgroup stages 64 rows in LDS (padded so that lane r reads row r without bank conflicts), lane r of every
// wave OWNS row r, and the waves walk the queries -- whose elements are wave-uniform, i.e. scalar operands -- four
// at a time: 32 partial sums per (row, query) in registers, FMA chains over the blocks in order, the reduce tree
// as plain adds in the lane (dot.s:45-53).  Same arithmetic, same bits as dist_core.h; no cross-lane traffic at
// all.  A (row, query) pair whose distance is not above the query's current threshold (an upper bound of its
// k-th best distance, taken from the rows scanned so far) is appended to the query's candidate list; k_flat_merge
// folds the few candidates into the running top list under (distance, slot) order -- the order the reference's
// walk in storage order with `dist >= tail -> skip` produces (flat.go:104,121-123).
constexpr uint32_t kScanRows = 64;  // rows per workgroup tile = lanes of a wave

struct FlatScanArgs {
  const float *slab;     // [n][ld] permuted rows
  const float *queries;  // [nq][dim] the queries as the caller passed them (original layout)
  const uint64_t *ids;   // slot -> id, 0 = deleted
  const float *thr;      // [nq] upper bound of the query's limit-th best distance so far (+inf: list not full)
  uint32_t *cnt;         // [nq] candidates appended in this launch
  uint2 *cand;           // [nq][cap] (slot, distance bits)
  uint32_t cap;
  uint32_t first, rows;  // slab rows [first, first + rows)
  uint32_t nq, ld, dim, nblk, tail, skip_slot;
  int metric;
};

// LDS holds only the row tile, [64][32 nblk + 4] floats in the ORIGINAL element order (the slab's permuted rows are
// turned back while staging; +16 B per row: lane r starts at bank 4r, so the lanes' ds_read_b128 never collide).
// The queries are read where the caller left them (original layout) through the scalar cache: element pair
// (32b + 2j, 32b + 2j + 1) of a query is one SGPR pair, of the lane's row one VGPR pair, and one v_pk_fma_f32
// advances the reference's partial sums 2j and 2j + 1 of block b -- each half is the reference's own fused
// multiply-add (VFMADD231PS, dot.s:24-27), blocks in the reference's order.  A wave takes the queries two at a
// time (64 SGPRs of operands per block); kWaves waves per workgroup share the tile.
typedef float f2v __attribute__((ext_vector_type(2)));
template <bool L2>
__device__ __forceinline__ f2v chain1_pk(f2v acc, f2v x, f2v y) {
  if constexpr (L2) {
    const f2v d = x - y;  // separately rounded, like VSUBPS (euclidean.s:27)
    return __builtin_elementwise_fma(d, d, acc);
  } else {
    return __builtin_elementwise_fma(x, y, acc);
  }
}
constexpr int kScanWaves = SDB_SCAN_WAVES;
constexpr int kScanPairs = SDB_SCAN_PAIRS;
template <bool L2>
__global__ __launch_bounds__(kScanWaves * 64) void k_flat_scan(const float *__restrict__ slab,
                                                               const float *__restrict__ queries,
                                                               const FlatScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [64][kstride]
  const uint32_t nblk = a.nblk, kstride = nblk * 32 + 4;
  float *next_group = tile + (size_t)kScanRows * kstride;  // the waves' work counter, behind the tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform, and the compiler may know it
  if (tid == 0) *reinterpret_cast<uint32_t *>(next_group) = 0;
  (void)wave;
  const uint32_t row0 = a.first + blockIdx.x * kScanRows;
  const uint32_t nrows = min(kScanRows, a.first + a.rows - row0);
  // ---- stage the tile: consecutive threads read consecutive 16 B of a slab row; float4 c = 32 g + L of it holds
  // blocks 4g..4g+3 of partial sum L, i.e. original elements 32 (4g + k) + L
  const uint32_t row_f4 = a.ld / 4;
  for (uint32_t i = tid; i < kScanRows * row_f4; i += kScanWaves * 64) {
    const uint32_t r = i / row_f4, c = i % row_f4;
    const uint32_t rr = r < nrows ? r : nrows - 1;
    const float4 v = reinterpret_cast<const float4 *>(slab + (size_t)(row0 + rr) * a.ld)[c];
    const uint32_t g = c >> 5, Lx = c & 31;
    float *d = tile + (size_t)r * kstride + 128 * g + Lx;
    if (4 * g + 0 < nblk) d[0] = v.x;
    if (4 * g + 1 < nblk) d[32] = v.y;
    if (4 * g + 2 < nblk) d[64] = v.z;
    if (4 * g + 3 < nblk) d[96] = v.w;
  }
  __syncthreads();
  const uint32_t slot = row0 + (uint32_t)lane;
  const bool live = (uint32_t)lane < nrows && slot != a.skip_slot && a.ids[(uint32_t)lane < nrows ? slot : row0] != 0;
  const float4 *myrow = reinterpret_cast<const float4 *>(tile + (size_t)lane * kstride);
  // kScanPairs query pairs per pass: the lane's row block is read from LDS once and used for 2 * kScanPairs queries
  const uint32_t ngroups = (a.nq + 2 * kScanPairs - 1) / (2 * kScanPairs);
  // The query groups are handed out as the waves come for them: the SIMD favours its oldest wave, so with a fixed
  // share per wave the four waves of a SIMD finish one after another and the last runs alone with its loads exposed
  // (3.1 of 4 wave slots occupied on average, SQ_WAVE_CYCLES); taken from a counter, all waves finish within one group
#if SDB_SCAN_DYN
  for (;;) {
    const uint32_t grp = next_query_group(next_group, lane);
    if (grp >= ngroups) break;
#else
  for (uint32_t grp = (uint32_t)wave; grp < ngroups; grp += kScanWaves) {
#endif
    uniform_float *xq[2 * kScanPairs];
#pragma unroll
    for (int k = 0; k < 2 * kScanPairs; k++) {
      const uint32_t q = grp * 2 * kScanPairs + k;
      xq[k] = as_uniform(queries) + (size_t)(q < a.nq ? q : a.nq - 1) * a.dim;  // past the end: the last query again, dropped
    }
    f2v acc[2 * kScanPairs][16];
#pragma unroll
    for (int k = 0; k < 2 * kScanPairs; k++)
#pragma unroll
      for (int j = 0; j < 16; j++) acc[k][j] = f2v{0.0f, 0.0f};
#pragma unroll 1
    for (uint32_t b = 0; b < nblk; b++) {
      float4 y[8];
#pragma unroll
      for (int i = 0; i < 8; i++) y[i] = myrow[b * 8 + i];
#pragma unroll
      for (int pp = 0; pp < kScanPairs; pp++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {  // float4 i of the block: pairs j = 2i, 2i + 1
          const uniform_f4v u = reinterpret_cast<uniform_float4 *>(xq[2 * pp] + b * 32)[i];  // wave-uniform: scalar loads
          const uniform_f4v w = reinterpret_cast<uniform_float4 *>(xq[2 * pp + 1] + b * 32)[i];
          acc[2 * pp][2 * i] = chain1_pk<L2>(acc[2 * pp][2 * i], f2v{u.x, u.y}, f2v{y[i].x, y[i].y});
          acc[2 * pp][2 * i + 1] = chain1_pk<L2>(acc[2 * pp][2 * i + 1], f2v{u.z, u.w}, f2v{y[i].z, y[i].w});
          acc[2 * pp + 1][2 * i] = chain1_pk<L2>(acc[2 * pp + 1][2 * i], f2v{w.x, w.y}, f2v{y[i].x, y[i].y});
          acc[2 * pp + 1][2 * i + 1] = chain1_pk<L2>(acc[2 * pp + 1][2 * i + 1], f2v{w.z, w.w}, f2v{y[i].z, y[i].w});
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * kScanPairs; k++) {
      // the reduce tree of dot.s:45-53 / euclidean.s:55-63 over acc[L] = acc[k][L / 2][L % 2] (the tail vector
      // {t,0,0,0} is all zero here)
      auto A = [&](int L) { return acc[k][L >> 1][L & 1]; };
      float r4[4];
#pragma unroll
      for (int l = 0; l < 4; l++) {
        const float s0 = ((A(l) + A(8 + l)) + A(16 + l)) + A(24 + l);
        const float s1 = ((A(l + 4) + A(12 + l)) + A(20 + l)) + A(28 + l);
        r4[l] = (s0 + s1) + 0.0f;
      }
      const float dist = metric_finish((r4[0] + r4[1]) + (r4[2] + r4[3]), a.metric);
      const uint32_t q = grp * 2 * kScanPairs + k;
      if (q < a.nq && live && !(dist > a.thr[q])) {  // rare: a handful per query and million rows
        const uint32_t at = atomicAdd(a.cnt + q, 1u);
        if (at < a.cap) a.cand[(size_t)q * a.cap + at] = make_uint2(slot, __float_as_uint(dist));
      }
    }
  }
}

// ---- the same scan on the matrix cores (dot and cosine) ------------------------------------------------------
// The scan is a GEMM with a prescribed summation order: pair (row, query) keeps the reference's 32 partial sums,
// partial sum L = 8a + t taking elements 32b + L in the order of b, each step one fused multiply-add
// (VFMADD231PS, dot.s:24-27).  v_mfma_f32_16x16x1_4b_f32 is exactly that step for 4 x 16 x 16 independent
// accumulators at once: D[blk][i][j] = fma(A[blk][i], B[blk][j], D[blk][i][j]) -- one product, one rounding, denormals
// kept (tools/probes/mfma_exact.hip: 5.2 M outputs against fmaf(), none different; 154 TFLOP/s, the packed-FP32 rate,
// with two register operands per 1 024 FMAs instead of LDS/scalar operands per 128).  The four blocks of the
// instruction are given to the four quarters t mod 4 = blk of the partial sums, so that a wave's tile is 16 rows x 16
// queries and the whole reduce tree of dot.s:45-53 stays inside the lane:
//   A lane 16 blk + i = row i,   element 32b + 8a + 4tt + blk        accumulator set k = 2a + tt (8 sets x 16 registers)
//   B lane 16 blk + j = query j, the same element                    D register 4 blk + i4, lane l: row 4 (l / 16) + i4,
//                                                                     query l % 16, partial sum t = blk + 4tt
//   S_tt = ((c(0,tt) + c(1,tt)) + c(2,tt)) + c(3,tt)   r4[blk] = (S_0 + S_1) + 0   dist = (r4[0] + r4[1]) + (r4[2] + r4[3])
// A wave keeps its 16 rows in registers for the whole launch (32 floats per slab group and lane: the slab's permuted
// layout puts elements 32 (4g + kk) + L, kk = 0..3, side by side, one 16-byte load) and walks the queries 16 at a time;
// the queries arrive through LDS in the operand order (k_flat_swizzle_queries lays them out once per call), double
// buffered, shared by the 4 waves = 64 rows of the workgroup.  Euclidean is not of this shape ((x - y) is rounded per
// pair before the multiply, euclidean.s:27) and stays on k_flat_scan.
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr uint32_t kMfmaMaxGroups = 8;  // slab groups of 4 blocks a wave holds in registers: d <= 1024

// A query group's image: [b][h][l][c] = query 16 G + l % 16, element 32 b + 8 (2h + c / 2) + 4 (c % 2) + l / 16 for the
// nblk whole blocks; then, when the rows have a tail of d % 32 elements, [j][kTailPitch]: the tail elements of query
// 16 G + j in order, zero padded (768 floats = three 1 KB pieces)
constexpr uint32_t kTailPitch = 36;        // floats per row / query of tail elements in LDS: 16-byte aligned, lane j at bank 4j
constexpr uint32_t kTailImgFloats = 768;   // 16 x 36 = 576, rounded up to whole 1 KB pieces
__global__ void k_flat_swizzle_queries(const float *__restrict__ q, float *__restrict__ out, uint32_t nq, uint32_t dim,
                                       uint32_t nblk, uint32_t tail, uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t grp = nblk * 512 + (tail ? kTailImgFloats : 0);
  const uint32_t G = i / grp, o = i % grp;
  if (o >= nblk * 512) {  // the tail section
    const uint32_t t = o - nblk * 512, j = t / kTailPitch, m = t % kTailPitch, qi = 16 * G + j;
    out[i] = (j < 16 && m < tail && qi < nq) ? q[(size_t)qi * dim + 32 * nblk + m] : 0.0f;
    return;
  }
  const uint32_t c = o & 3, l = (o >> 2) & 63, h = (o >> 8) & 1, b = o >> 9;
  const uint32_t k = 4 * h + c, e = 32 * b + 8 * (k >> 1) + 4 * (k & 1) + (l >> 4), qi = 16 * G + (l & 15);
  out[i] = qi < nq ? q[(size_t)qi * dim + e] : 0.0f;
}

// Rows of up to 12 blocks leave room for two workgroups per CU (256 registers per wave: 128 accumulators, 8 NBLK row
// operands); longer rows run one workgroup per CU.
// TAIL: rows of d % 32 != 0.  The tail elements form one more chain per pair (dot.s:35-43: t = fma(x, y, t) over them in
// order), added to r[0] of the reduce tree (dot.s:50): a ninth accumulator set whose block 0 holds the pairs' tail chains
// (lanes of the other blocks multiply zeros: fma(0, 0, 0) = +0, the {t, 0, 0, 0} vector of the reference), one matrix
// instruction per tail element, operands from LDS ([row][36] and [query][36], four elements per 16-byte read).
template <int NBLK, bool TAIL>
__global__ __launch_bounds__(256, (NBLK <= 12 && !TAIL) ? 2 : 1) void k_flat_scan_mfma(const float *__restrict__ slab, const float *__restrict__ qsw,
                                                        const FlatScanArgs a) {
  constexpr int NG = (NBLK + 3) / 4;
  constexpr uint32_t kGrpF4 = NBLK * 128 + (TAIL ? kTailImgFloats / 4 : 0);  // float4 per query group
  constexpr int kPieces = (int)(kGrpF4 / 64);                               // 1 KB pieces of a group's image
  extern __shared__ __attribute__((aligned(16))) float bs[];  // [2][group image], [64][kTailPitch] row tails, the nq thresholds
  float *rowtail = bs + 2 * kGrpF4 * 4;
  float *thr_s = rowtail + (TAIL ? kScanRows * kTailPitch : 0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t row0 = a.first + blockIdx.x * kScanRows, end = a.first + a.rows;
  // ---- the wave's 16 rows: lane 16 blk + i holds, per slab group g and accumulator set k, the float4 at
  // 128 g + 4 (8a + 4tt + blk) = elements 32 (4g + kk) + 8a + 4tt + blk, kk = 0..3
  f4v A[NG][8];
  {
    const uint32_t r = row0 + 16 * wave + (lane & 15);
    const float *src = slab + (size_t)(r < end ? r : end - 1) * a.ld + 4 * (lane >> 4);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float *p = src + 128 * g + 4 * (8 * (k >> 1) + 4 * (k & 1));
        constexpr int kLast = NBLK - 4 * (NG - 1);  // blocks in the last slab group: its other components are padding that
        if (g + 1 < NG || kLast == 4) {              // no matrix instruction reads -- loaded, they held 8 registers each
          A[g][k] = *reinterpret_cast<const f4v *>(p);
        } else {
          A[g][k] = f4v{p[0], kLast > 1 ? p[1] : 0.0f, kLast > 2 ? p[2] : 0.0f, 0.0f};
        }
      }
  }
  // rows 4 (lane / 16) + i4 of the wave's tile are the ones this lane emits
  const uint32_t slot0 = row0 + 16 * wave + 4 * (lane >> 4);
  uint32_t live = 0;
#pragma unroll
  for (int i4 = 0; i4 < 4; i4++)
    if (slot0 + i4 < end && slot0 + i4 != a.skip_slot && a.ids[slot0 + i4] != 0) live |= 1u << i4;
  const uint32_t ngroups = (a.nq + 15) / 16;
  // a query group's operands (2 NBLK KB) go from the swizzled image straight into LDS (global_load_lds_dwordx4: a wave
  // instruction moves 64 x 16 B = 1 KB, no registers in between); a group past the end: the last one again
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto dma = [&](uint32_t G) __attribute__((always_inline)) {
    const char *src = reinterpret_cast<const char *>(qsw) + (size_t)min(G, ngroups - 1) * (kGrpF4 * 16) + wave * 1024;
    {
      // The group's base stays a scalar pair of ITS iteration (opaque to the optimiser): otherwise the loop-invariant
      // part of the address is hoisted together with the lane's offset into a 64-bit VGPR pair that lives through the
      // whole loop -- at 11 and 12 blocks, where two waves per SIMD leave exactly 256 registers, that pair (and what it
      // pushed out) went to scratch and came back once per query group.  With a scalar base the load takes the
      // saddr + 32-bit offset form and the lane's part is one register.
      uint32_t lo = (uint32_t)reinterpret_cast<uintptr_t>(src), hi = (uint32_t)(reinterpret_cast<uintptr_t>(src) >> 32);
      lo = __builtin_amdgcn_readfirstlane(lo), hi = __builtin_amdgcn_readfirstlane(hi);
      asm volatile("" : "+s"(lo), "+s"(hi));
      src = reinterpret_cast<const char *>(((uintptr_t)hi << 32) | lo);
    }
    char *dst = reinterpret_cast<char *>(bs) + (size_t)(G & 1) * (kGrpF4 * 16) + wave * 1024;
    uint32_t lane_off = lane * 16;  // the only per-lane part of the address ...
    asm volatile("" : "+v"(lane_off));  // ... made here, per group, from the lane id: one shift instead of a live 64-bit pair
#pragma unroll
    for (int piece = 0; piece < (kPieces + 3) / 4; piece++)
      if (4 * piece + wave < kPieces)
        __builtin_amdgcn_global_load_lds((glb_void *)(src + piece * 4096 + lane_off), (lds_void *)(dst + piece * 4096), 16, 0, 0);
  };
  // 8 NBLK matrix instructions: the wave's 16 rows against query group G, operands of block b + 1 on their way from
  // LDS while block b is multiplied
  auto multiply = [&](f16v (&acc)[8], f16v &T, uint32_t G) __attribute__((always_inline)) {
    const f4v *bq = reinterpret_cast<const f4v *>(bs) + (size_t)(G & 1) * kGrpF4 + lane;
    if constexpr (TAIL) {  // the tail chains first: block 0 of the instruction, lanes 0..15 = row i / query j
#pragma unroll
      for (int r = 0; r < 16; r++) T[r] = 0.0f;
      const float *ta = rowtail + (16 * wave + (lane & 15)) * kTailPitch;
      const float *tb = bs + (size_t)(G & 1) * kGrpF4 * 4 + NBLK * 512 + (lane & 15) * kTailPitch;
      const bool low = lane < 16;
      for (uint32_t m = 0; m < a.tail; m += 4) {
        const f4v x4 = *reinterpret_cast<const f4v *>(ta + m), y4 = *reinterpret_cast<const f4v *>(tb + m);
#pragma unroll
        for (int c = 0; c < 4; c++)
          if (m + c < a.tail)  // uniform: exactly `tail` steps, like the reference's loop
            T = __builtin_amdgcn_mfma_f32_16x16x1f32(low ? x4[c] : 0.0f, low ? y4[c] : 0.0f, T, 0, 0, 0);
      }
    }
    f4v b0 = bq[0], b1 = bq[64];
#pragma unroll
    for (int b = 0; b < NBLK; b++) {
      f4v n0 = b0, n1 = b1;
      if (b + 1 < NBLK) n0 = bq[(b + 1) * 128], n1 = bq[(b + 1) * 128 + 64];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        f16v c = acc[k];
        if (b == 0)
#pragma unroll
          for (int r = 0; r < 16; r++) c[r] = 0.0f;
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x1f32(A[b / 4][k][b % 4], k < 4 ? b0[k & 3] : b1[k & 3], c, 0, 0, 0);
      }
      b0 = n0, b1 = n1;
    }
  };
  // the reduce tree of dot.s:45-53 in the lane, then the threshold test of the lane's 4 (row, query) pairs
  auto reduce = [&](const f16v (&acc)[8], const f16v &T, float (&dist)[4]) __attribute__((always_inline)) {
    const f16v s0 = ((acc[0] + acc[2]) + acc[4]) + acc[6];
    const f16v s1 = ((acc[1] + acc[3]) + acc[5]) + acc[7];
    f16v r4;
    if constexpr (TAIL) r4 = (s0 + s1) + T;  // r[0] + tail chain, r[1..3] + 0 (dot.s:50)
    else r4 = (s0 + s1) + 0.0f;
#pragma unroll
    for (int i4 = 0; i4 < 4; i4++) {
      dist[i4] = (r4[i4] + r4[4 + i4]) + (r4[8 + i4] + r4[12 + i4]);  // metric_finish in emit
      asm volatile("" : "+v"(dist[i4]));  // keeps the adds here, next to the matrix instructions, not behind emit's branch
    }
  };
  auto threshold = [&](uint32_t G) __attribute__((always_inline)) { return thr_s[min(16 * G + (lane & 15), a.nq - 1)]; };
  // distance.go:19-25 without a branch per pair: cosine 1 - x, dot -x (the metric is the same for the whole launch)
  const bool cosine = a.metric == SDB_METRIC_COSINE;
  auto emit = [&](const float (&dist)[4], uint32_t G, float thr) __attribute__((always_inline)) {
    float d[4];
    uint32_t hit = 0;
#pragma unroll
    for (int i4 = 0; i4 < 4; i4++) {
      const float one_minus = 1.0f - dist[i4], neg = -dist[i4];
      d[i4] = cosine ? one_minus : neg;
      hit |= (!(d[i4] > thr) ? 1u : 0u) << i4;
    }
    hit &= live;
    if (16 * G + (lane & 15) >= a.nq) hit = 0;
    if (hit == 0) return;  // nearly always
    uint32_t *cnt_g = a.cnt + 16 * G;                       // uniform base, the lane's part is a 32-bit offset
    uint2 *cand_g = a.cand + (size_t)(16 * G) * a.cap;
    uint32_t j = lane & 15;
    asm volatile("" : "+v"(j));  // j * cap is made HERE, by the rare lane that has a hit, not kept through the loop
#pragma unroll
    for (int i4 = 0; i4 < 4; i4++)
      if ((hit >> i4) & 1u) {
        const uint32_t at = atomicAdd(cnt_g + j, 1u);
        if (at < a.cap) cand_g[j * a.cap + at] = make_uint2(slot0 + i4, __float_as_uint(d[i4]));
      }
  };
  // the order of issue: block b + 1's operands are asked for after the second matrix instruction of block b, six
  // instructions (192 cycles of the matrix pipe) before the first one that needs them
  auto pipeline = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // block 0's operands and the group's threshold
#pragma unroll
    for (int b = 0; b < NBLK; b++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (k == 2 && b + 1 < NBLK) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
  };
  dma(0);
  for (uint32_t i = tid; i < a.nq; i += 256) thr_s[i] = a.thr[i];
  if constexpr (TAIL) {  // the 64 rows' tail elements: slab floats [ld - 32, ld - 32 + tail)
    for (uint32_t i = tid; i < kScanRows * 32; i += 256) {
      const uint32_t r = i >> 5, m = i & 31, row = row0 + r;
      rowtail[r * kTailPitch + m] = m < a.tail ? slab[(size_t)(row < end ? row : end - 1) * a.ld + (a.ld - 32) + m] : 0.0f;
    }
    for (uint32_t i = tid; i < kScanRows * 4; i += 256) rowtail[(i >> 2) * kTailPitch + 32 + (i & 3)] = 0.0f;
  }
  __syncthreads();  // waits for this wave's DMAs (vmcnt) and for everybody else's
  f16v acc[8], T;
  float dist[4];
  // One barrier per group: LDS buffer (G + 1) & 1 was read by multiply(G - 1) before the previous barrier, is refilled
  // with group G + 1 now, and is read by multiply(G + 1) after this one.  Two workgroups share a CU (256 registers
  // per wave): while one wave reduces and emits, the matrix pipe of its SIMD works for the other.
  for (uint32_t G = 0; G < ngroups; G++) {
    dma(G + 1);
    const float thr = threshold(G);
    multiply(acc, T, G);
    if constexpr (!TAIL) pipeline();
    reduce(acc, T, dist);
    emit(dist, G, thr);
    __syncthreads();
  }
}

constexpr int kMergeThreads = 256;
// fold a launch's candidates into the running top list of one query (256 threads): the `limit` smallest of
// (list U candidates) under (distance, slot) order; then the new threshold.  cnt > cap: the list was cut -- flag it.
__global__ __launch_bounds__(kMergeThreads) void k_flat_merge(const uint32_t *__restrict__ cnt, const uint2 *__restrict__ cand,
                                                   uint32_t cap, uint32_t limit, uint32_t *__restrict__ top_slot,
                                                   float *__restrict__ top_dist, uint32_t *__restrict__ top_len,
                                                   float *__restrict__ thr, uint32_t *__restrict__ overflow) {
  extern __shared__ __attribute__((aligned(8))) uint2 items[];  // [128 + cap]
  const uint32_t q = blockIdx.x;
  const int t = threadIdx.x;
  const uint32_t len = top_len[q];
  uint32_t c = cnt[q];
  if (c > cap) {
    if (t == 0) atomicOr(overflow, 1u);
    c = cap;
  }
  // A NaN distance in the list or among the candidates: the reference's loop is not a sort then (flat.go:104-123 with a
  // NaN on either side of `>=` / `<`) and only the walk in storage order reproduces it -- the call starts over on the
  // block path, like a list that overflowed
  bool nan = false;
  for (uint32_t i = t; i < len; i += kMergeThreads) nan |= top_dist[(size_t)q * 128 + i] != top_dist[(size_t)q * 128 + i];
  for (uint32_t i = t; i < c; i += kMergeThreads) nan |= (cand[(size_t)q * cap + i].y & 0x7fffffffu) > 0x7f800000u;
  if (nan) atomicOr(overflow, 1u);
  if (c == 0) {
    if (t == 0) thr[q] = len >= limit ? top_dist[(size_t)q * 128 + limit - 1] : __int_as_float(0x7f800000);
    return;
  }
  for (uint32_t i = t; i < len; i += kMergeThreads) items[i] = make_uint2(top_slot[(size_t)q * 128 + i], __float_as_uint(top_dist[(size_t)q * 128 + i]));
  for (uint32_t i = t; i < c; i += kMergeThreads) items[len + i] = cand[(size_t)q * cap + i];
  __syncthreads();
  const uint32_t m = len + c;
  for (uint32_t i = t; i < m; i += kMergeThreads) {
    const float d = __uint_as_float(items[i].y);
    const uint32_t s = items[i].x;
    uint32_t rank = 0;
    for (uint32_t j = 0; j < m; j++) {
      const float dj = __uint_as_float(items[j].y);
      rank += (dj < d || (dj == d && items[j].x < s)) ? 1u : 0u;
    }
    if (rank < limit) top_slot[(size_t)q * 128 + rank] = s, top_dist[(size_t)q * 128 + rank] = d;
  }
  __syncthreads();
  if (t == 0) {
    const uint32_t nl = m < limit ? m : limit;
    top_len[q] = nl;
  }
  __syncthreads();
  if (t == 0) {
    __threadfence();
    thr[q] = (m >= limit) ? top_dist[(size_t)q * 128 + limit - 1] : __int_as_float(0x7f800000);
  }
}

template <bool L2>
static int launch_flat_scan(const FlatScanArgs &a, hipStream_t stream) {
  const dim3 grid((a.rows + kScanRows - 1) / kScanRows);
  const size_t lds = (size_t)kScanRows * (a.nblk * 32 + 4) * sizeof(float) + 16;  // the tile and the group counter
  static std::atomic<uint64_t> attr{0};
  if (first_use_on_this_device(attr))
    SDB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_flat_scan<L2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
  hipLaunchKernelGGL((k_flat_scan<L2>), grid, dim3(kScanWaves * 64), lds, stream, a.slab, a.queries, a);
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}

static size_t flat_mfma_lds_bytes(uint32_t nblk, uint32_t tail, uint64_t nq) {
  return (size_t)2 * (nblk * 512 + (tail ? kTailImgFloats : 0)) * sizeof(float) +
         (tail ? (size_t)kScanRows * kTailPitch * sizeof(float) : 0) + (size_t)nq * sizeof(float);
}
template <int NBLK, bool TAIL>
static int launch_flat_scan_mfma_nbt(const FlatScanArgs &a, const float *qsw, hipStream_t stream) {
  const dim3 grid((a.rows + kScanRows - 1) / kScanRows);
  const size_t lds = flat_mfma_lds_bytes(NBLK, TAIL ? 1 : 0, a.nq);
  static std::atomic<uint64_t> attr{0};
  if (first_use_on_this_device(attr))
    SDB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_flat_scan_mfma<NBLK, TAIL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((k_flat_scan_mfma<NBLK, TAIL>), grid, dim3(256), lds, stream, a.slab, qsw, a);
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}
template <int NBLK>
static int launch_flat_scan_mfma_nb(const FlatScanArgs &a, const float *qsw, hipStream_t stream) {
  return a.tail ? launch_flat_scan_mfma_nbt<NBLK, true>(a, qsw, stream) : launch_flat_scan_mfma_nbt<NBLK, false>(a, qsw, stream);
}
static int launch_flat_scan_mfma(const FlatScanArgs &a, const float *qsw, hipStream_t stream) {
  switch (a.nblk) {
#define SDB_MFMA_CASE(NB) \
  case NB:                \
    return launch_flat_scan_mfma_nb<NB>(a, qsw, stream);
    SDB_MFMA_CASE(1) SDB_MFMA_CASE(2) SDB_MFMA_CASE(3) SDB_MFMA_CASE(4) SDB_MFMA_CASE(5) SDB_MFMA_CASE(6) SDB_MFMA_CASE(7)
    SDB_MFMA_CASE(8) SDB_MFMA_CASE(9) SDB_MFMA_CASE(10) SDB_MFMA_CASE(11) SDB_MFMA_CASE(12) SDB_MFMA_CASE(13)
    SDB_MFMA_CASE(14) SDB_MFMA_CASE(15) SDB_MFMA_CASE(16) SDB_MFMA_CASE(17) SDB_MFMA_CASE(18) SDB_MFMA_CASE(19)
    SDB_MFMA_CASE(20) SDB_MFMA_CASE(21) SDB_MFMA_CASE(22) SDB_MFMA_CASE(23) SDB_MFMA_CASE(24) SDB_MFMA_CASE(25)
    SDB_MFMA_CASE(26) SDB_MFMA_CASE(27) SDB_MFMA_CASE(28) SDB_MFMA_CASE(29) SDB_MFMA_CASE(30) SDB_MFMA_CASE(31)
    SDB_MFMA_CASE(32)
#undef SDB_MFMA_CASE
  }
  return fail(SDB_ERR_INVALID, "row too long for the matrix-core scan");
}

}  // namespace sdb

using namespace sdb;

extern "C" int sdb_index_flat_search(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit,
                                     const uint64_t *filter_offsets, const uint64_t *filter_ids, uint64_t *out_ids,
                                     float *out_dists, uint32_t *out_counts, int mem, void *stream_) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (nq == 0) return SDB_OK;
  if (!queries || !out_ids || !out_dists || !out_counts) return fail(SDB_ERR_INVALID, "NULL argument");
  if (limit < 1 || limit > 128) return fail(SDB_ERR_INVALID, "limit must be between 1 and 128, got %u", limit);
  if (nq > 65535) return fail(SDB_ERR_INVALID, "at most 65535 queries per call");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  const bool filtered = filter_offsets != nullptr;
  if (filtered && filter_offsets[nq] && !filter_ids) return fail(SDB_ERR_INVALID, "filter_ids is NULL");
  DeviceGuard dg(ix->P.device);
  hipStream_t stream = as_stream(stream_);
  const RowLayout &l = ix->lay;
  // the scan sees the last committed version like a graph search does (index.h graph versions); the shared lock
  // is kept until everything is enqueued
  std::shared_lock<sdb::ViewMutex> rl(ix->view_mu);
  const sdb_index::View vw = ix->view;
  const uint32_t n = vw.n;
  // ---- filtered: per-query candidate slots, ascending (filter.Contains(point.Id()), flat.go:100)
  std::vector<uint32_t> f_off, f_slots;
  uint32_t max_f = 0;
  if (filtered) {
    f_off.assign(nq + 1, 0);
    for (uint64_t q = 0; q < nq; q++) {
      const size_t f0 = f_slots.size();
      for (uint64_t i = filter_offsets[q]; i < filter_offsets[q + 1]; i++) {
        const int64_t s = ix->slot_of_committed(filter_ids[i], vw.n);
        if (s >= 0 && s != ix->start_slot) f_slots.push_back((uint32_t)s);
      }
      if (!std::is_sorted(f_slots.begin() + f0, f_slots.end())) std::sort(f_slots.begin() + f0, f_slots.end());
      f_off[q + 1] = (uint32_t)f_slots.size();
      max_f = std::max<uint32_t>(max_f, (uint32_t)(f_slots.size() - f0));
    }
  }
  // ---- buffers: staged queries/outputs for host callers, running top lists, one distance block
  uint32_t chunk = filtered ? std::max<uint32_t>(max_f, 1)
                            : (uint32_t)std::min<uint64_t>(std::max<uint32_t>(n, 1), (1ull << 28) / nq);
  // the streaming scans: plain store, no filter, rows of whole 32-float blocks, a table worth streaming.  Dot and
  // cosine rows of up to 1 024 floats run on the matrix cores (k_flat_scan_mfma; its query operands and thresholds
  // must fit LDS), euclidean rows -- and the others when the matrix-core scan is switched off -- of up to 608 floats
  // on the packed-FMA kernel (k_flat_scan).  The first rows still go through the block path: they seed the thresholds.
  constexpr uint32_t kSeedRows = 4096, kMinSegment = 32768, kCandCap = 8192;
  const bool streamable = !filtered && !ix->pq && l.nblk >= 1 && n >= kMinSegment && nq <= 8192;
  const bool mfma = streamable && ix->P.metric != SDB_METRIC_EUCLIDEAN && l.nblk <= 4 * kMfmaMaxGroups && !ix->tune_no_mfma &&
                    flat_mfma_lds_bytes(l.nblk, l.tail, nq) <= 160 * 1024;
  const bool fast = mfma || (streamable && l.tail == 0 && l.nblk <= 19);
  if (fast) chunk = std::min<uint32_t>(chunk, kSeedRows);  // the block path only sees the seed rows
  const uint32_t stride = (chunk + 63) & ~63u;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_q = carve(mem == SDB_MEM_HOST ? nq * l.dim * 4 : 0);
  const size_t o_oi = carve(mem == SDB_MEM_HOST ? nq * limit * 8 : 0), o_od = carve(mem == SDB_MEM_HOST ? nq * limit * 4 : 0);
  const size_t o_oc = carve(mem == SDB_MEM_HOST ? nq * 4 : 0);
  const size_t o_ts = carve(nq * 128 * 4), o_td = carve(nq * 128 * 4), o_tl = carve(nq * 4);
  const size_t o_fo = carve(filtered ? (nq + 1) * 4 : 0), o_fs = carve(filtered ? f_slots.size() * 4 : 0);
  const size_t o_d = carve((size_t)nq * stride * 4);
  const sdb_pq *pq = ix->pq;
  const size_t o_lut = carve(pq ? (size_t)nq * pq->M * pq->K * 4 : 0);
  const size_t o_thr = carve(fast ? nq * 4 : 0);
  const size_t o_cnt = carve(fast ? nq * 4 + 256 : 0), o_cand = carve(fast ? (size_t)nq * kCandCap * 8 : 0);
  const uint32_t qsw_floats = mfma ? (uint32_t)((nq + 15) / 16) * (l.nblk * 512 + (l.tail ? kTailImgFloats : 0)) : 0;
  const size_t o_qsw = carve((size_t)qsw_floats * 4);
  // The workspace carries the event a commit waits for before it hands the copy this scan reads to the writer.  It
  // stays this call's own until the stream has been synchronised (`fr` below is destroyed first): released earlier, a
  // scan on another stream could take it and re-record the event, and the commit would wait for that scan only.
  struct WsHold {
    const sdb_index *ix;
    Workspace *ws;
    hipStream_t s;
    ~WsHold() { ix->release_ws(ws, s, false); }
  } hold{ix, ix->acquire_ws(stream, false), stream};
  // stream-ordered scratch from the device's pool (it keeps what it is given back: distance.hip), not a hipMalloc /
  // hipFree pair of ~85 MB per call
  char *buf = nullptr;
  keep_pool_memory(ix->P.device);
  SDB_HIP(hipMallocAsync(reinterpret_cast<void **>(&buf), off, stream));
  struct Free {
    char *p;
    hipStream_t s;
    ~Free() {
      (void)hipFreeAsync(p, s);
      (void)hipStreamSynchronize(s);
    }
  } fr{buf, stream};
  const float *dq = queries;
  uint64_t *d_oi = out_ids;
  float *d_od = out_dists;
  uint32_t *d_oc = out_counts;
  if (mem == SDB_MEM_HOST) {
    SDB_HIP(hipMemcpyAsync(buf + o_q, queries, nq * l.dim * 4, hipMemcpyHostToDevice, stream));
    dq = (const float *)(buf + o_q);
    d_oi = (uint64_t *)(buf + o_oi), d_od = (float *)(buf + o_od), d_oc = (uint32_t *)(buf + o_oc);
    SDB_HIP(hipMemsetAsync(buf + o_oi, 0, o_ts - o_oi, stream));
  }
  uint32_t *top_slot = (uint32_t *)(buf + o_ts), *top_len = (uint32_t *)(buf + o_tl);
  float *top_dist = (float *)(buf + o_td);
  SDB_HIP(hipMemsetAsync(top_slot, 0xFF, nq * 128 * 4, stream));
  SDB_HIP(hipMemsetAsync(top_dist, 0, nq * 128 * 4, stream));
  SDB_HIP(hipMemsetAsync(top_len, 0, nq * 4, stream));
  uint32_t *d_fo = nullptr, *d_fs = nullptr;
  if (filtered) {
    d_fo = (uint32_t *)(buf + o_fo), d_fs = (uint32_t *)(buf + o_fs);
    SDB_HIP(hipMemcpyAsync(d_fo, f_off.data(), (nq + 1) * 4, hipMemcpyHostToDevice, stream));
    if (!f_slots.empty())
      SDB_HIP(hipMemcpyAsync(d_fs, f_slots.data(), f_slots.size() * 4, hipMemcpyHostToDevice, stream));
    SDB_HIP(hipStreamSynchronize(stream));
  }
  float *d_dist = (float *)(buf + o_d);
  const size_t lds = (size_t)(l.ng * 128 + 32) * 4;
  const uint32_t total = filtered ? max_f : n;
  auto block_path = [&](uint32_t begin, uint32_t end) -> int {
  for (uint32_t first = begin; first < end; first += chunk) {
    const uint32_t rows = std::min<uint32_t>(chunk, end - first);
    dim3 grid((rows + 63) / 64, (unsigned)nq);
    if (pq) {
      if (first == 0) SDB_TRY(pq_build_lut(pq, dq, nq, (float *)(buf + o_lut), stream));
      hipLaunchKernelGGL(k_flat_dist_pq, dim3((rows + 255) / 256, (unsigned)nq), dim3(256), 0, stream,
                         (const float *)(buf + o_lut), ix->d_codes, d_fs, d_fo, first, rows, d_dist, stride, pq->M, pq->K);
    } else if (ix->P.metric == SDB_METRIC_EUCLIDEAN)
      hipLaunchKernelGGL(k_flat_dist<true>, grid, dim3(256), lds, stream, ix->d_slab, dq, d_fs, d_fo, first, rows, d_dist,
                         stride, l.dim, l.nblk, l.ng, l.tail, l.ld, (int)ix->P.metric);
    else
      hipLaunchKernelGGL(k_flat_dist<false>, grid, dim3(256), lds, stream, ix->d_slab, dq, d_fs, d_fo, first, rows, d_dist,
                         stride, l.dim, l.nblk, l.ng, l.tail, l.ld, (int)ix->P.metric);
    SDB_HIP(hipGetLastError());
    FlatFoldArgs fa{};
    fa.dists = d_dist, fa.stride = stride, fa.count = rows, fa.slot_off = d_fo, fa.slots = d_fs, fa.first_row = first;
    fa.skip_slot = ix->start_slot >= 0 ? (uint32_t)ix->start_slot : kNoSlot;
    fa.ids = vw.ids;
    fa.limit = limit, fa.top_slot = top_slot, fa.top_dist = top_dist, fa.top_len = top_len;
    hipLaunchKernelGGL(k_flat_fold, dim3((unsigned)nq), dim3(64), 0, stream, fa);
    SDB_HIP(hipGetLastError());
  }
  return SDB_OK;
  };
  if (!fast) {
    SDB_TRY(block_path(0, total));
  } else {
    const uint32_t seed = std::min<uint32_t>(kSeedRows, n);
    SDB_TRY(block_path(0, seed));
    float *d_thr = (float *)(buf + o_thr);
    uint32_t *d_cnt = (uint32_t *)(buf + o_cnt), *d_over = d_cnt + nq;
    uint2 *d_cand = (uint2 *)(buf + o_cand);
    SDB_HIP(hipMemsetAsync(d_cnt, 0, nq * 4 + 4, stream));
    const size_t merge_lds = (size_t)(128 + kCandCap) * sizeof(uint2);
    static std::atomic<uint64_t> merge_attr{0};
    if (first_use_on_this_device(merge_attr))
      SDB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_flat_merge), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)merge_lds));
    // thresholds from the seed rows (no candidates yet: the merge only derives thr from the lists)
    hipLaunchKernelGGL(k_flat_merge, dim3((unsigned)nq), dim3(kMergeThreads), merge_lds, stream, d_cnt, d_cand, kCandCap, limit, top_slot,
                       top_dist, top_len, d_thr, d_over);
    SDB_HIP(hipGetLastError());
    // A row passes its query's threshold with probability ~ limit / rows-scanned-so-far, so a segment as long as
    // everything before it brings about `limit` candidates per query.  The first segment is 32 768 rows (one
    // workgroup tile for each of the 512 resident workgroups; ~8 limit candidates after the 4 096 seed rows), then
    // the segments double: 6 launches for 1M rows, candidate lists far under kCandCap, merges that cost nothing
    for (uint32_t first = seed, seg = kMinSegment - seed; first < n; first += seg, seg = first) {
      FlatScanArgs sa{};
      sa.slab = ix->d_slab, sa.queries = dq, sa.ids = vw.ids, sa.thr = d_thr, sa.cnt = d_cnt;
      sa.cand = d_cand, sa.cap = kCandCap, sa.first = first, sa.rows = std::min<uint32_t>(seg, n - first);
      sa.nq = (uint32_t)nq, sa.ld = l.ld, sa.dim = l.dim, sa.nblk = l.nblk, sa.tail = l.tail, sa.skip_slot = ix->start_slot >= 0 ? (uint32_t)ix->start_slot : kNoSlot;
      sa.metric = (int)ix->P.metric;
      if (mfma) {
        if (first == seed) {
          hipLaunchKernelGGL(k_flat_swizzle_queries, dim3((qsw_floats + 255) / 256), dim3(256), 0, stream, dq,
                             (float *)(buf + o_qsw), (uint32_t)nq, l.dim, l.nblk, l.tail, qsw_floats);
          SDB_HIP(hipGetLastError());
        }
        SDB_TRY(launch_flat_scan_mfma(sa, (const float *)(buf + o_qsw), stream));

      } else if (ix->P.metric == SDB_METRIC_EUCLIDEAN) SDB_TRY(launch_flat_scan<true>(sa, stream));
      else SDB_TRY(launch_flat_scan<false>(sa, stream));
      hipLaunchKernelGGL(k_flat_merge, dim3((unsigned)nq), dim3(kMergeThreads), merge_lds, stream, d_cnt, d_cand, kCandCap, limit,
                         top_slot, top_dist, top_len, d_thr, d_over);
      SDB_HIP(hipGetLastError());
      SDB_HIP(hipMemsetAsync(d_cnt, 0, nq * 4, stream));
    }
    // a candidate list that overflowed (a sea of equal distances) lost entries: start over on the block path
    uint32_t over = 0;
    SDB_HIP(hipMemcpyAsync(&over, d_over, 4, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipStreamSynchronize(stream));
    if (over) {
      SDB_HIP(hipMemsetAsync(top_slot, 0xFF, nq * 128 * 4, stream));
      SDB_HIP(hipMemsetAsync(top_dist, 0, nq * 128 * 4, stream));
      SDB_HIP(hipMemsetAsync(top_len, 0, nq * 4, stream));
      SDB_TRY(block_path(0, total));
    }
  }
  hipLaunchKernelGGL(k_flat_emit, dim3((unsigned)nq), dim3(64), 0, stream, top_slot, top_dist, top_len, vw.ids, limit,
                     d_oi, d_od, d_oc);
  SDB_HIP(hipGetLastError());
  {  // a commit must wait for this scan before it hands the copy it reads to the writer
    Workspace *ws = hold.ws;
    if (!ws->launched) (void)hipEventCreateWithFlags(&ws->launched, hipEventDisableTiming);
    if (ws->launched && hipEventRecord(ws->launched, stream) == hipSuccess) ws->launched_valid = true;
    else (void)hipStreamSynchronize(stream);
  }
  rl.unlock();
  if (mem == SDB_MEM_HOST) {
    SDB_HIP(hipMemcpyAsync(out_ids, d_oi, nq * limit * 8, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipMemcpyAsync(out_dists, d_od, nq * limit * 4, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipMemcpyAsync(out_counts, d_oc, nq * 4, hipMemcpyDeviceToHost, stream));
  }
  return SDB_OK;  // `fr` synchronises and frees
}
SDB_API_CATCH("sdb_index_flat_search")

// vecStore.Set for a flat index (flat.go:46-49): store vectors without touching the graph.
namespace sdb {
int store_rows_public(sdb_index *ix, uint32_t first, uint32_t n, const float *dev_vectors, hipStream_t stream);
}
namespace sdb {
// vecStore.Delete on an index without a graph: the row becomes a tombstone (id 0), like a deleted graph node's
__global__ void k_flat_tombstone(uint64_t *ids, uint8_t *dirty, const uint32_t *dead, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ids[dead[i]] = 0, dirty[dead[i]] = 1;
}
}  // namespace sdb

// rows of a flat index (no graph) leave the store; part of the open transaction (the caller commits)
// *changed: set once device rows or host tables have begun to change (a failure before that leaves the index as it was)
static int flat_tombstone(sdb_index *ix, std::vector<uint32_t> &slots, bool *changed) {
  if (slots.empty()) return SDB_OK;
  std::sort(slots.begin(), slots.end());
  slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
  // ---- the host tables' memory first (a dense id table gets its hash map -- a dense table ignores the map, filling it
  // changes no answer --, the removed ids enter the transaction's record): running out of it must leave the rows alone
  struct HostPrep {
    sdb_index *ix;
    bool done = false, was_dense = false;
    std::vector<uint64_t> recorded;
    ~HostPrep() {
      if (done) return;
      std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);
      for (uint64_t id : recorded) ix->tx_deleted.erase(id);
      if (was_dense) ix->id2slot.clear();
    }
  } prep{ix};
  {
    std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);  // searches translate filter ids with these tables
    prep.recorded.reserve(slots.size());
    if (ix->dense_ids) {
      prep.was_dense = true;
      ix->id2slot.clear();
      ix->id2slot.reserve((size_t)ix->n * 2);
      for (uint32_t s = 0; s < ix->n; s++)
        if (ix->h_ids[s] != 0) ix->id2slot.emplace(ix->h_ids[s], s);
    }
    // until commit a search on the committed rows still finds a removed id; a row this transaction appended itself was
    // never visible, and an id keeps the committed row it had when the transaction began (first record wins)
    for (uint32_t s : slots)
      if (s < ix->tx_n0 && ix->tx_deleted.emplace(ix->h_ids[s], s).second) prep.recorded.push_back(ix->h_ids[s]);
  }
  uint32_t *d_dead = nullptr;
  SDB_HIP(hipMalloc(&d_dead, slots.size() * 4));
  hipError_t e = hipMemcpy(d_dead, slots.data(), slots.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    *changed = true;
    hipLaunchKernelGGL(sdb::k_flat_tombstone, dim3((unsigned)((slots.size() + 255) / 256)), dim3(256), 0, nullptr, ix->d_ids,
                       ix->d_dirty, d_dead, (uint32_t)slots.size());
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  (void)hipFree(d_dead);
  prep.done = *changed;  // the device rows are marked (or half-marked: the caller declares the handle unusable)
  if (e != hipSuccess) return fail(SDB_ERR_DEVICE, "tombstone failed: %s", hipGetErrorString(e));
  std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);
  ix->dense_ids = false;  // the dense id -> slot shortcut does not survive holes (the map was filled above)
  for (uint32_t s : slots) {
    ix->id2slot.erase(ix->h_ids[s]);
    ix->h_ids[s] = 0;
  }
  ix->n_dead += (uint32_t)slots.size();
  return SDB_OK;
}

// vecStore.Delete as IndexFlat.InsertUpdateDelete calls it for a point without a vector (flat.go:50-52): ids that are
// not stored are skipped (ItemCache.Delete of a missing key is not an error).  Only for an index without a graph
// (no start node); a graph index deletes through sdb_index_delete_batch.
extern "C" int sdb_index_remove_vectors(sdb_index *ix, uint64_t n, const uint64_t *ids) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (n == 0) return SDB_OK;
  if (!ids) return fail(SDB_ERR_INVALID, "ids is NULL");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  if (ix->start_slot >= 0) return fail(SDB_ERR_STATE, "the index has a graph: delete through sdb_index_delete_batch");
  std::vector<uint32_t> slots;
  for (uint64_t i = 0; i < n; i++) {
    const int64_t s = ix->slot_of(ids[i]);
    if (s >= 0) slots.push_back((uint32_t)s);
  }
  if (slots.empty()) return SDB_OK;
  DeviceGuard dg(ix->P.device);
  SDB_TRY(ix->begin_write());
  int trc;
  bool changed = false;
  const bool was_dirty = ix->tx_dirty;
  ix->tx_dirty = true;
  try {
    trc = flat_tombstone(ix, slots, &changed);
  } catch (...) {
    trc = sdb::on_exception("sdb_index_remove_vectors");
  }
  if (trc != SDB_OK) {
    if (changed) ix->broken = true;  // device rows / host tables half-marked: no way back (index.h `broken`)
    else {  // nothing was touched: the transaction this call opened for itself closes again
      ix->tx_dirty = was_dirty;
      if (!ix->tx_explicit && !was_dirty) ix->in_tx = false;
    }
    return trc;
  }
  if (!ix->tx_explicit) {
    SDB_TRY(ix->commit(nullptr));
    SDB_HIP(hipDeviceSynchronize());
  }
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_remove_vectors")

extern "C" int sdb_index_set_vectors(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors, int mem) try {
  if (!ix || !vectors) return fail(SDB_ERR_INVALID, "NULL argument");
  if (n == 0) return SDB_OK;
  if ((uint64_t)ix->n + n >= 0x7FFFFFFFull) return fail(SDB_ERR_INVALID, "too many nodes");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  std::vector<uint64_t> new_ids(n);
  std::vector<uint32_t> replaced;  // rows of ids that are stored already: vecStore.Set overwrites (plain.go:58-65)
  for (uint64_t i = 0; i < n; i++) {
    new_ids[i] = ids ? ids[i] : std::max<uint64_t>(ix->max_node_id, SDB_STARTID) + 1 + i;
    if (new_ids[i] == 0) return fail(SDB_ERR_INVALID, "invalid point id: 0");
    const int64_t s = ix->slot_of(new_ids[i]);
    if (s >= 0) {
      // on a graph index a stored id is an update, which is delete_batch + insert_batch; a flat index simply
      // replaces the vector: the old row becomes a tombstone, the new one is appended (rows are never rewritten
      // in place -- a search that runs meanwhile reads whole rows)
      if (ix->start_slot >= 0)
        return fail(SDB_ERR_EXISTS, "point %llu exists: updates are not on the device path", (unsigned long long)new_ids[i]);
      replaced.push_back((uint32_t)s);
    }
  }
  if (ids) {  // the same id twice in one call: the last vector wins, like consecutive Set calls
    std::unordered_map<uint64_t, uint64_t> last;
    for (uint64_t i = 0; i < n; i++) last[new_ids[i]] = i;
    if (last.size() != n) return fail(SDB_ERR_INVALID, "an id appears twice in one set_vectors call");
  }
  DeviceGuard dg(ix->P.device);
  const RowLayout &l = ix->lay;
  const uint32_t n0 = ix->n;
  SDB_TRY(ix->reserve(n0 + (uint32_t)n));
  float *staging = nullptr;
  const float *dvec = vectors;
  if (mem == SDB_MEM_HOST) {
    SDB_HIP(hipMalloc(&staging, n * l.dim * 4));
    hipError_t e = hipMemcpy(staging, vectors, n * l.dim * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(staging);
      return fail(SDB_ERR_DEVICE, "H2D copy failed: %s", hipGetErrorString(e));
    }
    dvec = staging;
  }
  int rc = store_rows_public(ix, n0, (uint32_t)n, dvec, nullptr);
  hipError_t e = hipMemcpy(ix->d_ids + n0, new_ids.data(), n * 8, hipMemcpyHostToDevice);
  (void)hipDeviceSynchronize();
  if (staging) (void)hipFree(staging);
  if (rc != SDB_OK) return rc;
  if (e != hipSuccess) return fail(SDB_ERR_DEVICE, "id copy failed: %s", hipGetErrorString(e));
  SDB_TRY(ix->begin_write());  // appended rows become visible to searches at commit
  const bool was_dirty = ix->tx_dirty;
  ix->tx_dirty = true;
  bool changed = false;
  auto tables = [&]() -> int {
  SDB_TRY(flat_tombstone(ix, replaced, &changed));
  std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);
  ix->h_ids.reserve(ix->h_ids.size() + n);
  changed = true;  // from here on the id tables move
  bool dense = ix->dense_ids;
  for (uint64_t i = 0; i < n; i++) {
    if (dense && !ix->h_ids.empty() && new_ids[i] != ix->h_ids[0] + ix->h_ids.size()) {
      dense = false;
      ix->id2slot.reserve((ix->h_ids.size() + n) * 2);
      for (size_t s = 0; s < ix->h_ids.size(); s++) ix->id2slot.emplace(ix->h_ids[s], (uint32_t)s);
    }
    if (!dense) ix->id2slot.emplace(new_ids[i], (uint32_t)ix->h_ids.size());
    ix->h_ids.push_back(new_ids[i]);
    if (new_ids[i] > ix->max_node_id) ix->max_node_id = new_ids[i];
  }
  ix->dense_ids = dense;
  ix->n = n0 + (uint32_t)n;
  return SDB_OK;
  };
  int trc;
  try {
    trc = tables();
  } catch (...) {  // the id tables are half-way between two states
    trc = sdb::on_exception("sdb_index_set_vectors");
  }
  if (trc != SDB_OK) {
    if (changed) ix->broken = true;  // the id tables are half-way between two states
    else if (!ix->tx_explicit && !was_dirty) ix->tx_dirty = false, ix->in_tx = false;  // (the stored rows lie past n: invisible)
    return trc;
  }
  if (!ix->tx_explicit) {
    SDB_TRY(ix->commit(nullptr));
    SDB_HIP(hipDeviceSynchronize());
  }
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_set_vectors")
