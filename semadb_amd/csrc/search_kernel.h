// search_kernel.h -- K2: GPU-resident greedySearch, one wavefront per query.
//
// Restates shard/index/vamana/search.go:9-102 (greedySearch) on top of
// shard/index/vamana/distset.go:166-200 (DistSet.AddWithLimit) with bit-identical distances
// (dist_core.h), so result ids, distances, visit order, n_dist and n_hop equal the reference's.
//
// Wave-level mapping (64 lanes, one wave per workgroup, no workgroup barriers):
//   * candidate set S (cap = searchSize): a sorted array held in VGPRs, entry e in lane e%64 of
//     register e/64; the `visited` flag rides in bit 31 of the slot word.
//   * one hop: wave-uniform pick of the first unvisited entry -> one coalesced 256-byte read of
//     its adjacency row (lane j = edge j, edge order preserved) -> per-lane test-and-set in the
//     query's visited set (an exact hash set in LDS, or the HBM bitset; the ids of one row are
//     distinct) -> distances for the new neighbours, two candidates per wave instruction (one per
//     32-lane half), 16-byte row loads, up to U pairs of rows in flight -> AddWithLimit for the
//     neighbours that beat the current tail, in edge order (the tail only shrinks, so a neighbour
//     that fails the current threshold can never pass a later one).
//   * with one wave per SIMD every instruction of any kind costs an issue slot, so the hop is written
//     for instruction count: the reduce tree is DPP adds (dist_core.h), row slots reach the lanes
//     through a rank-compacted LDS list, a row address is one 64-bit mad (PlainDist::hop_fast).
#pragma once
#include "dist_core.h"
#include "index.h"

namespace sdb {

struct SearchArgs {
  const float *slab;
  const uint32_t *adj;
  const uint64_t *ids;
  uint32_t *bitsets;
  uint32_t words_per_query;
  const float *queries;  // [nq][dim] original layout
  uint32_t dim, nblk, ng, tail, ld;
  uint32_t start_slot;
  // the start node's edges beyond its 64-entry row (index.h h_start_ext), kNoSlot-padded to whole chunks of 64
  const uint32_t *start_ext;
  uint32_t start_ext_n;
  uint32_t search_size;
  uint32_t limit;
  int metric;
  uint64_t *out_ids;
  float *out_dists;
  uint32_t *out_counts;
  uint32_t *tr_ndist, *tr_nhop, *tr_nedges;
  uint64_t *tr_visit;
  uint32_t visit_cap;
  // build path: the visit log as (slot, dist), in visit order
  uint32_t *vis_slots;
  float *vis_dists;
  uint32_t *vis_count;
  uint32_t vis_cap;
  // product-quantized store (product.go:238-277): per-query LUT [nq][M*K] and per-slot codes [n][M]
  const float *pq_lut;
  const uint8_t *pq_codes;
  uint32_t pq_M, pq_K;
  // != NULL: the code rows of every node's neighbours behind its adjacency row, [rows][64][M] (index.h d_adjcodes), for
  // the copy of the graph `adj` points at
  const uint8_t *adj_codes;
  uint32_t adj_rows;  // rows of `adj` (what tells an adjacency row from a chunk of the start node's overflow list)
  uint32_t pq_lut_in_lds;  // != 0: the kernel copies its LUT into LDS first
  uint32_t pq_narrow;      // 1: never a multi-wave walk (k_greedy_search_pqw, k_greedy_search_pq2); 2: the one-query-per-CU variant of the former for M = 192

  // filtered search (search.go:33-51,93-95): per query CSR of seeds (<= searchSize slots, ascending id
  // order) and of the whole filter as ascending slots; rbitsets = the result set's own visited set
  const uint32_t *seed_off, *seeds, *filt_off, *filt_slots;
  // filter lists resolved on the device (index.hip k_filter_resolve): the counts are not offset differences then
  // (segments keep the caller's offsets, unknown ids leave gaps at their ends); NULL: offset differences
  const uint32_t *seed_cnt, *filt_cnt;
  // != NULL: Contains (:93) is answered from the filter's IDS -- the uploaded lists, strictly ascending, at filt_off's
  // offsets -- with the id of the expanded node (ids[slot]), for tables where ascending ids are not ascending slots
  // (an id that was deleted and inserted again sits behind larger ids); filt_slots is not read then
  const uint64_t *filt_ids;
  uint32_t *rbitsets;
  // build path, full-precision store: every evaluated (slot, distance) also goes into the query's
  // direct-mapped table of 2^(32 - dcache_shift) entries (last writer wins); the back-edge prunes of the
  // round look their pair distances up there (build.hip BuildArgs::dcache).  NULL: not collected.
  uint2 *dcache;
  uint32_t dcache_shift;
  unsigned long long *totals;  // build path: [0] += n_dist, [1] += n_edges of every query (sdb_index_build_stats)
  uint32_t prefer_bitset;  // != 0: never use the LDS hash visited set (large build rounds)
  uint32_t wide_hash;      // != 0: quantized store keeps the 32-bit-cell set (HashVisited) instead of HashVisited16
  uint32_t hash16_probes;  // test knob: buckets a key of HashVisited16 may try (0 = all 15); fewer make the `stuck` spill common
  uint32_t hash_limit;     // ids the LDS hash set may hold before the query falls back to its bitset
  uint32_t wide_mode;      // the workgroup-per-query walk: 0 = calls of up to kWideMaxQueries queries, 1 = never, 2 = always
  uint32_t wide_pull;      // 2: that walk's other waves work ahead on the row the walk expands next (PlainWideDist); 0: they only share a hop's rows
  // Two-precision hop (index.h d_sketch; SDB_TUNE_SKETCH): a float16 copy of the slab, same element order, rows of
  // `ld` halves.  A neighbour whose float16 distance lies above the candidate array's last distance by more than the
  // bound on |float16 distance - the reference's float32 distance| is discarded by AddWithLimit whatever its exact
  // distance is (distset.go:184; the distance is never looked at again), so only the others are read in float32.
  const uint16_t *sketch;
  const float *sketch_norm;          // [rows] ||y16||^2 of every row (euclidean tables: d16 = ||q16||^2 + ||y16||^2 - 2 q16.y16)
  float sk_emax, sk_ymax;            // max over the rows of ||y - y16|| and of ||y16|| (k_sketch_rows), rounded up
  uint32_t sk_audit;                 // != 0: evaluate everything exactly as well and count decisions the exact distance contradicts
  unsigned long long *sk_counters;   // [0] += neighbours discarded on their float16 distance, [1] += contradicted ones (audit)
};

// pairs of candidate rows a wave keeps in flight per chunk.  DEEP (one wave per SIMD, the batch-search
// configuration: 4 waves per CU, the whole 512-entry register file available): ~190 VGPRs of loads;
// otherwise ~96 so that three waves per SIMD still fit (large build rounds).
template <int NG, bool DEEP>
struct ChunkPairs {
  static constexpr int base = NG <= 3 ? 8 : NG <= 4 ? 6 : NG <= 6 ? 4 : NG <= 8 ? 3 : NG <= 12 ? 2 : 1;
#ifndef SDB_DEEP3_PAIRS
#define SDB_DEEP3_PAIRS 16  // measurement builds: pairs per round of the d = 384 batch walk (tools/kernel_ab.py)
#endif
  static constexpr int value = DEEP ? (NG == 3 ? SDB_DEEP3_PAIRS : 2 * base) : base;  // DEEP: 32 / 24 / 16 / 12 / 8 / 4 rows in flight
};

// The kernel's own arguments once more, through a pointer the optimiser cannot see through.  Every walk kernel takes
// its SearchArgs by value as its first parameter, i.e. at offset 0 of the kernarg segment; a field read through `a` is
// loaded in the kernel's first instructions and -- with ~60 fields against ~100 SGPRs -- parked in a VGPR lane until it
// is used.  A field that only the epilogue reads is read through this view: one scalar load at the point of use.
__device__ __forceinline__ const SearchArgs &cold_args(const SearchArgs &) {
  typedef const __attribute__((address_space(4))) SearchArgs *kernarg_view;
  uintptr_t p = (uintptr_t)(kernarg_view)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const SearchArgs *)p;
}

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ float rlf(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Ordering point for LDS traffic inside ONE wavefront (the search kernels run one wave per workgroup): the
// LDS queue is in order per wave, so a ds_write is visible to the wave's later ds_reads without a barrier;
// only the compiler has to keep the order (and nothing waits for outstanding global loads, as a
// __syncthreads() would).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// U pairs of (query, candidate) raw distances; slot[u] is this lane's candidate row (same for the
// 32 lanes of a half).  res[u] is valid in lanes 0 and 32.
template <int NG, bool L2, int U>
__device__ __forceinline__ void chunk_dist(const float *__restrict__ slab, uint32_t ld, uint32_t tail,
                                           const float4 (&xq)[NG > 0 ? NG : 1], float xt,
                                           const uint32_t (&slot)[U], float (&res)[U], int lane
#ifdef SDB_STAMPS
                                           , unsigned long long *st = nullptr
#endif
) {
  const int L = lane & 31;
  float4 y[U][NG > 0 ? NG : 1];
  float yt[U];
#ifdef SDB_STAMPS
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll
  for (int u = 0; u < U; u++) {
    const float *row = slab + (size_t)slot[u] * ld;
    const float4 *r4 = reinterpret_cast<const float4 *>(row) + L;
#pragma unroll
    for (int g = 0; g < NG; g++) y[u][g] = r4[g * 32];
    yt[u] = tail ? row[NG * 128 + L] : 0.0f;
  }
#ifdef SDB_STAMPS
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll
  for (int u = 0; u < U; u++) {
    float acc = 0.0f;
#pragma unroll
    for (int g = 0; g < NG; g++) acc = chain4<L2>(acc, xq[g], y[u][g]);
    float t = tail ? tail_chain<L2>(xt, yt[u], tail, lane) : 0.0f;
    res[u] = asm_reduce(acc, t, lane);
  }
#ifdef SDB_STAMPS
  asm volatile("" ::"v"(res[U - 1]));
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (st) st[0] += t1 - t0, st[1] += t2 - t1, st[2] += t3 - t2;  // issue, wait, compute
#endif
}

// Generic-dimension variant: the permuted query tile lives in LDS (qs, ng*32 float4 + 32 tail floats).
template <bool L2, int U>
__device__ __forceinline__ void chunk_dist_lds(const float *__restrict__ slab, uint32_t ld, uint32_t ng,
                                               uint32_t tail, const float *qs, const uint32_t (&slot)[U],
                                               float (&res)[U], int lane) {
  const int L = lane & 31;
  float acc[U];
  const float4 *r4[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    acc[u] = 0.0f;
    r4[u] = reinterpret_cast<const float4 *>(slab + (size_t)slot[u] * ld) + L;
  }
  const float4 *q4 = reinterpret_cast<const float4 *>(qs) + L;
  // four 128-float groups of every row per step: 4 * U row loads in flight instead of U; the partial sums
  // still take their groups in ascending order
  constexpr int GC = 4;
  for (uint32_t g0 = 0; g0 < ng; g0 += GC) {
    float4 y[U][GC];
#pragma unroll
    for (int k = 0; k < GC; k++)
      if (g0 + k < ng) {
#pragma unroll
        for (int u = 0; u < U; u++) y[u][k] = r4[u][(g0 + k) * 32];
      }
#pragma unroll
    for (int k = 0; k < GC; k++)
      if (g0 + k < ng) {
        const float4 x = q4[(g0 + k) * 32];
#pragma unroll
        for (int u = 0; u < U; u++) acc[u] = chain4<L2>(acc[u], x, y[u][k]);
      }
  }
  float xt = tail ? qs[ng * 128 + L] : 0.0f;
#pragma unroll
  for (int u = 0; u < U; u++) {
    float yt = tail ? slab[(size_t)slot[u] * ld + ng * 128 + L] : 0.0f;
    float t = tail ? tail_chain<L2>(xt, yt, tail, lane) : 0.0f;
    res[u] = asm_reduce(acc[u], t, lane);
  }
}

// ---- distance policies: what vecStore.DistanceFromFloat(query) binds (plain.go:76-85 / product.go:238-277)

// Full-precision store.  NG >= 0: compile-time group count, query in registers.  NG == -1: run-time
// ng, query tile in LDS.
typedef _Float16 sk_h2 __attribute__((ext_vector_type(2)));
// float -> float16 as the sketch stores it: round to nearest, magnitudes below the smallest NORMAL half become 0 (no
// denormal half is ever an operand of v_dot2_f32_f16, whatever the mode register says about them); what this loses is
// part of the measured error ||v - v16||, not an assumption
__device__ __forceinline__ _Float16 sk_half(float v) { return fabsf(v) < 6.103515625e-5f ? (_Float16)0.0f : (_Float16)v; }

template <int NG, bool L2, bool DEEP = false, int UPAIRS = 0, bool SK = false>  // UPAIRS != 0: that many pairs of rows per chunk; SK: two-precision hop
struct PlainDist {
  static constexpr bool kSketch = SK;
  static constexpr bool kHasStamps = true;
  static constexpr bool kPointDistances = true;  // dist(query, row) is distFn between two stored vectors
  // search_body: nothing is worked ahead on between the hops.  (Round 5 tried the walker's naming of the next hop's row
  // here too -- its adjacency row asked for under AddWithLimit, a round trip less per hop in the batch's tail: 0.983
  // against 0.975 ms per batch, the naming's ~300 instructions per hop cost a lone wave more than the hidden latency.
  // The cheap half -- no naming, only the adjacency row of the array's first unvisited entry asked for before
  // AddWithLimit and taken from the register when the walk does go there -- made no difference at all: 0.955 / 0.958 /
  // 0.957 against 0.957 / 0.954 / 0.973 ms in alternating runs; the batch walk is bound by bytes, not by a wave's round trips.)
#ifndef SDB_SKETCH_SPECULATE
#define SDB_SKETCH_SPECULATE 0  // measurement builds (tools/sketch_ab.py): 1 measured 0.716 against 0.722 ms (min of 10) -- within the noise, off
#endif
  // ... with the two-precision hop the walk IS bound by its round trips; naming the next hop's adjacency row and asking
  // for it before AddWithLimit runs (search_body's kSpeculate path without the small-call kernel's marker wave) was
  // measured there too and moved the kernel by 1 %: the naming costs about what the hidden round trip saves
  static constexpr bool kSpeculate = SK && SDB_SKETCH_SPECULATE;
  static constexpr bool marked = false;  // (no marker wave: nothing is tested ahead)
  __device__ __forceinline__ void take_marks(int, uint64_t &, uint32_t &) {}
  static constexpr int NGR = NG > 0 ? NG : 1;
  static constexpr int U = UPAIRS ? UPAIRS : (NG >= 0 ? ChunkPairs<NG, DEEP>::value : 4);
  // dynamic LDS of the policy: NG == -1 the query tile; NG >= 0 the hop scratch -- pending slots by rank
  // [kHopSlots], raw distances by rank [kHopSlots], a U-word dump
  // (two-precision hop: pairs of float16 rows in flight per round -- all of a hop's new neighbours in ONE round where
  // the registers allow it, 2 NG registers per row: a round is a dependent memory round trip of the hop)
#ifndef SDB_SKETCH_ROWS
#define SDB_SKETCH_ROWS 32  // measurement builds: tools/sketch_ab.py
#endif
  static constexpr int kSkBudget = L2 ? 72 : 96;  // (the euclidean exact stage holds differences as well)
  static constexpr int US = !SK ? U : (SDB_SKETCH_ROWS < kSkBudget / NGR ? SDB_SKETCH_ROWS : kSkBudget / NGR);
  static constexpr int UX = U > US ? U : US;
  static constexpr uint32_t kHopSlots = 64 + UX;  // ranks 0..63 and the overrun of the last half-wave run
  static constexpr size_t kLdsBytes = NG >= 0 ? (2 * kHopSlots + UX) * sizeof(uint32_t) : 0;
  float4 xq[NGR];
  float xt;
  float *qs;
  uint32_t *hs;
  sk_h2 qh[SK ? NGR : 1][2];  // the query in float16, in xq's element order
  float sk_eps;               // bound on |float16 distance - the reference's float32 distance| for this query, any row
  float sk_qq, sk_delta;      // euclidean: ||q16||^2, and ||q - q16|| + max ||y - y16|| (rounded up)
  float sk_yy;                // euclidean, rows asked for ahead: ||y16||^2 of this lane's edge
  // Rows of up to 384 floats: the float16 rows of ALL 64 edges of an adjacency row fit the register file (2 NG registers
  // per pair of rows), so they are asked for as soon as the edge ids are there -- BEFORE the visited-set test, whose
  // LDS round trips (~2 100 cycles) then run under the rows' flight instead of in front of it.  Rows of edges that turn
  // out to be visited already were read for nothing (a quarter more float16 bytes; the walk is not bound by bytes).
#ifndef SDB_SKETCH_AHEAD
#define SDB_SKETCH_AHEAD 1  // measurement builds: 0 = rows asked for after the test, compacted (tools/sketch_ab.py)
#endif
  static constexpr bool kSketchAhead = SK && SDB_SKETCH_AHEAD && NG >= 1 && NG <= 3;
  uint2 sky[kSketchAhead ? 32 : 1][kSketchAhead ? NGR : 1];  // pair u: edge 2u (lanes 0..31) and edge 2u + 1 (lanes 32..63)
  bool sk_go = false;        // search_body: this hop's rows are asked for ahead (the array is full, the copy is there)
  bool sk_loaded = false;
#ifdef SDB_STAMPS
  unsigned long long st[3] = {0, 0, 0};  // issue, wait, compute
#endif

  // The bound.  With q16, y16 the float16 copies (exact in float32): q.y - q16.y16 = (q - q16).y16 + q.(y - y16), so
  // |q.y - q16.y16| <= ||q - q16|| Ymax16 + ||q|| Emax (Cauchy-Schwarz; Emax, Ymax16 measured over all rows by
  // k_sketch_rows, ||q - q16|| and ||q|| measured here).  The two computed sums differ from the exact ones by their
  // rounding: the reference's 32 chains of NG x 4 / 32 ... fused multiply-adds and its 6-level tree, at most 4 NG + 6
  // roundings of relative size 2^-24 on sums bounded by ||q|| ||y||; the sketch's 2 NG v_dot2_f32_f16 (products of
  // halves are exact in float32, three additions each) and the same tree: 6 NG + 6.  For NG <= 8 that is below
  // 110 x 2^-24 = 6.6e-6; 2e-5 ||q|| (Ymax16 + Emax) is charged.  Norms are inflated by 1e-4 for their own rounding; the
  // final 1 - dot / -dot and the subtraction of the bound are covered per comparison (sketch_keep).  A query or a table
  // with a non-finite or float16-overflowing element makes the bound infinite or NaN: nothing is discarded then.
  __device__ __forceinline__ void init_sketch(const SearchArgs &a, int lane) {
    float e2 = 0.0f, n2 = 0.0f;
#pragma unroll
    for (int g = 0; g < NGR; g++) {
      const float v[4] = {xq[g].x, xq[g].y, xq[g].z, xq[g].w};
      _Float16 h[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        h[k] = sk_half(v[k]);
        const float dv = v[k] - (float)h[k];
        e2 = __builtin_fmaf(dv, dv, e2), n2 = __builtin_fmaf(v[k], v[k], n2);
      }
      qh[g][0] = sk_h2{h[0], h[1]}, qh[g][1] = sk_h2{h[2], h[3]};
    }
    // (both halves of the wave hold the same query: the sum over one half's 32 lanes)
    e2 = rlf(asm_reduce(e2, 0.0f, lane), 0), n2 = rlf(asm_reduce(n2, 0.0f, lane), 0);
    const float qerr = __builtin_sqrtf(e2) * 1.0001f, qn = __builtin_sqrtf(n2) * 1.0001f;
    sk_eps = (qerr * a.sk_ymax + qn * a.sk_emax + 2e-5f * qn * (a.sk_ymax + a.sk_emax)) * 1.0001f;
    if constexpr (L2) {
      float hh = 0.0f;
#pragma unroll
      for (int g = 0; g < NGR; g++) {
        const float h0 = (float)qh[g][0][0], h1 = (float)qh[g][0][1], h2 = (float)qh[g][1][0], h3 = (float)qh[g][1][1];
        hh = __builtin_fmaf(h0, h0, hh), hh = __builtin_fmaf(h1, h1, hh), hh = __builtin_fmaf(h2, h2, hh), hh = __builtin_fmaf(h3, h3, hh);
      }
      sk_qq = rlf(asm_reduce(hh, 0.0f, lane), 0);
      sk_delta = (qerr + a.sk_emax) * 1.0001f;
    }
  }

  // Squared euclidean distance.  With D16 = ||q16 - y16||^2 (exact) and delta >= ||q - q16|| + ||y - y16||:
  // | sqrt(D) - sqrt(D16) | <= delta, so D >= D16 - 2 delta sqrt(D16) (when sqrt(D16) < delta the right side is
  // negative and proves nothing, as it should).  D16 is computed as ||q16||^2 + ||y16||^2 - 2 q16.y16 from three rounded
  // sums: its error is below 1e-5 (||q16||^2 + ||y16||^2 + 2 |q16.y16|) (the same roundings as the dot form, 2^-24 each for
  // the two norms, three more for the combination -- well under 100 x 2^-24 = 6e-6).  The reference's own sum of rounded
  // squares of rounded differences is within (NG x 4 + 10) x 2^-24 < 1e-5 of D, relatively (all terms are >= 0).
  __device__ __forceinline__ bool sketch_out(const SearchArgs &a, float sum16, float yy, float tail_d) const {
    if constexpr (L2) {
      const float d16 = (sk_qq + yy) - 2.0f * sum16;
      const float err = 1e-5f * (sk_qq + yy + 2.0f * fabsf(sum16)) + 1e-30f;
      const float up = d16 + err;  // D16 <= up
      const float root = __builtin_sqrtf(up > 0.0f ? up : 0.0f) * 1.00001f;  // >= sqrt(D16)
      const float lower = ((d16 - err) - 2.0f * sk_delta * root) * (1.0f - 2e-5f);  // <= the reference's distance, rounding of this line included below
      return lower - 1e-6f * (fabsf(d16) + err + sk_delta * root) > tail_d;  // (the roundings of the two lines above; a NaN anywhere: false)
    } else {
      const float d16 = metric_finish(sum16, a.metric);
      // (1 - dot, -dot and the subtraction below round once each: 3 x 2^-24 of magnitudes below 1 + |d16| + eps)
      const float slack = sk_eps + 4e-7f * (1.0f + fabsf(d16) + sk_eps);
      return d16 - slack > tail_d;
    }
  }

  // float16 dot products of the pending rows by rank, two rows per wave instruction like rows_range
  __device__ __forceinline__ void sketch_range(const SearchArgs &a, const uint32_t *s_slot, float *s_res, int cnt, int lane) {
    const int L = lane & 31, half = lane >> 5;
    const char *baseL = reinterpret_cast<const char *>(a.sketch) + L * 8;
    const uint32_t row_bytes = a.ld * 2u;
    for (int c0 = 0; c0 < cnt; c0 += 2 * US) {
      const int m = cnt - c0 < 2 * US ? cnt - c0 : 2 * US;
      const int h0 = (m + 1) >> 1;
      const int base = c0 + (half ? h0 : 0);
      uint32_t sl[US];
#pragma unroll
      for (int u = 0; u < US; u++) sl[u] = s_slot[base + u];
      uint2 y[US][NGR];
#pragma unroll
      for (int u = 0; u < US; u++)
        if (u < h0) {
          const char *r = baseL + (uint64_t)sl[u] * row_bytes;
#pragma unroll
          for (int g = 0; g < NG; g++) y[u][g] = *reinterpret_cast<const uint2 *>(r + g * 256);
        }
      float *wp = (L == 0) ? s_res + base : s_res + kHopSlots;  // the other lanes write to a dump
#pragma unroll
      for (int u = 0; u < US; u++)
        if (u < h0) {
          float acc = 0.0f;
#pragma unroll
          for (int g = 0; g < NG; g++) {
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(sk_h2, y[u][g].x), qh[g][0], acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(sk_h2, y[u][g].y), qh[g][1], acc, false);
          }
          wp[u] = asm_reduce(acc, 0.0f, lane);
        }
    }
  }

  // the pending neighbours that AddWithLimit may keep: `out` gets the ones whose float16 distance is above `tail_d` by
  // more than the bound (every comparison with a NaN is false: such a neighbour is kept for the exact evaluation)
  __device__ __forceinline__ uint64_t sketch_keep(const SearchArgs &a, uint32_t nb, uint64_t pend, int lane, float tail_d,
                                                  uint64_t &out) {
    if constexpr (kSketchAhead) {
      if (sk_loaded) {  // the rows are in registers, by edge position
        // 32 sums per half-wave (pair u: edge 2u in lanes 0..31, edge 2u + 1 in lanes 32..63), each spread over the 32
        // lanes.  Instead of 32 reductions that each end in one lane (and a trip through LDS to get edge j's sum to lane
        // j), one transposing butterfly: at stride s a lane keeps the rows whose bit s equals its own and hands the
        // others to lane ^ s -- 16 + 8 + 4 + 2 + 1 exchanges instead of 32 x 5, and lane L of a half ends with the
        // complete sum of pair L.  One ds_bpermute then puts edge j's sum into lane j.  (Any order of additions is within
        // the bound; no LDS memory is touched.)
        float w[32];
#pragma unroll
        for (int u = 0; u < 32; u++) {
          float acc = 0.0f;
#pragma unroll
          for (int g = 0; g < NG; g++) {
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(sk_h2, sky[u][g].x), qh[g][0], acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(sk_h2, sky[u][g].y), qh[g][1], acc, false);
          }
          w[u] = acc;
        }
#define SDB_SK_STAGE(S)                                                                                         \
  {                                                                                                             \
    const bool up = (lane & (S)) != 0;                                                                          \
    _Pragma("unroll") for (int i = 0; i < (S); i++) {                                                           \
      const float keep = up ? w[i + (S)] : w[i], send = up ? w[i] : w[i + (S)];                                 \
      w[i] = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), ((S) << 10) | 0x1F));      \
    }                                                                                                           \
  }
        SDB_SK_STAGE(16)
        SDB_SK_STAGE(8)
        SDB_SK_STAGE(4)
        SDB_SK_STAGE(2)
        SDB_SK_STAGE(1)
#undef SDB_SK_STAGE
        // lane 32 h + L holds edge 2 L + h: lane j takes its own from lane 32 (j & 1) + (j >> 1)
        const float mysum = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & 1) * 32 + (lane >> 1)) * 4, __float_as_int(w[0])));
        const bool mine = (pend >> lane) & 1ull;
        out = __ballot(mine && sketch_out(a, mysum, L2 ? sk_yy : 0.0f, tail_d));
        return pend & ~out;
      }
    }
    const int cnt = __popcll(pend);
    const bool mine = (pend >> lane) & 1ull;
    float yy_me = 0.0f;
    if constexpr (L2)
      if (mine) yy_me = a.sketch_norm[nb];  // (in flight while the rows are summed)
    const uint32_t rank =
        __builtin_amdgcn_mbcnt_hi((uint32_t)(pend >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pend, 0u));
    uint32_t *s_slot = hs;
    float *s_res = reinterpret_cast<float *>(hs + kHopSlots);
    if (mine) {
      s_slot[rank] = nb;
      if ((int)rank == cnt - 1) s_slot[cnt] = nb;  // the spare entry
    }
    wave_lds_sync();
    sketch_range(a, s_slot, s_res, cnt, lane);
    wave_lds_sync();
    out = __ballot(mine && sketch_out(a, s_res[rank], yy_me, tail_d));
    wave_lds_sync();  // hop() compacts into the same scratch
    return pend & ~out;
  }

  __device__ __forceinline__ void init(const SearchArgs &a, uint32_t q, int lane, float *lds) {
    const int L = lane & 31;
    const float *__restrict__ qv = a.queries + (size_t)q * a.dim;
    qs = lds;
    hs = reinterpret_cast<uint32_t *>(lds);
    xt = 0.0f;
    if constexpr (NG >= 0) {
#pragma unroll
      for (int g = 0; g < NG; g++)
        xq[g] = make_float4(q_elem(qv, a.nblk, g, 0, L), q_elem(qv, a.nblk, g, 1, L),
                            q_elem(qv, a.nblk, g, 2, L), q_elem(qv, a.nblk, g, 3, L));
      if (NG == 0) xq[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      xt = (a.tail && (uint32_t)L < a.tail) ? qv[a.nblk * 32 + L] : 0.0f;
      if constexpr (SK) init_sketch(a, lane);
    } else {
      for (uint32_t i = lane; i < a.ng * 128; i += 64) {
        uint32_t g = i / 128, r = i % 128;
        qs[i] = q_elem(qv, a.nblk, g, r % 4, (int)(r / 4));
      }
      if (a.tail && lane < 32) qs[a.ng * 128 + lane] = (uint32_t)lane < a.tail ? qv[a.nblk * 32 + lane] : 0.0f;
      __syncthreads();
    }
  }

  __device__ __forceinline__ void chunk(const SearchArgs &a, const uint32_t (&slot)[U], float (&res)[U], int lane) {
#ifdef SDB_STAMPS
    if constexpr (NG >= 0) chunk_dist<NG, L2, U>(a.slab, a.ld, a.tail, xq, xt, slot, res, lane, st);
    else chunk_dist_lds<L2, U>(a.slab, a.ld, a.ng, a.tail, qs, slot, res, lane);
#else
    if constexpr (NG >= 0) chunk_dist<NG, L2, U>(a.slab, a.ld, a.tail, xq, xt, slot, res, lane);
    else chunk_dist_lds<L2, U>(a.slab, a.ld, a.ng, a.tail, qs, slot, res, lane);
#endif
  }

  // distance to one row (wave-uniform result)
  __device__ __forceinline__ float one(const SearchArgs &a, uint32_t s, int lane) {
    uint32_t slot[U];
    float res[U];
#pragma unroll
    for (int u = 0; u < U; u++) slot[u] = s;
    chunk(a, slot, res, lane);
    return metric_finish(rlf(res[0], 0), a.metric);
  }

  __device__ __forceinline__ void prefetch(const SearchArgs &a, uint32_t nb, bool valid) {  // float32 rows are fetched in hop()
    if constexpr (kSketchAhead) {
      sk_loaded = false;
      if (!sk_go) return;
      const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const int L = lane & 31;
      const bool hi = lane >= 32;
      const char *baseL = reinterpret_cast<const char *>(a.sketch) + L * 8;
      const uint32_t row_bytes = a.ld * 2u;
      const uint32_t safe = valid ? nb : a.start_slot;  // an edge that is not there: any row (its result is not looked at)
      if constexpr (L2) sk_yy = a.sketch_norm[safe];
#pragma unroll
      for (int u = 0; u < 32; u++) {
        const uint32_t s0 = rl(safe, 2 * u), s1 = rl(safe, 2 * u + 1);
        const char *r = baseL + (uint64_t)(hi ? s1 : s0) * row_bytes;
#pragma unroll
        for (int g = 0; g < NG; g++) sky[u][g] = *reinterpret_cast<const uint2 *>(r + g * 256);
      }
      sk_loaded = true;
    }
  }
  __device__ __forceinline__ void speculation(bool, const uint32_t *) {}
  __device__ __forceinline__ void ahead(const SearchArgs &, const uint32_t *, int, bool) {}
  // hooks of the multi-wave quantized walk (PQWideDist); nothing to do for a one-wave policy
  __device__ __forceinline__ void begin_row(const SearchArgs &, const uint32_t *, int) {}
  __device__ __forceinline__ void skip(int) {}

  // The hop with the per-row instruction count cut to the arithmetic: the pending slots are compacted into LDS by rank (mbcnt), each
  // half-wave takes a contiguous run of them (so a lane fetches its 16 slots with plain ds_reads, no bit
  // scans or readlanes), a row address is ONE v_mad_u64_u32 (slot * row bytes + [slab + 16 * L]), the raw
  // distance leaves lane 0 of its half through one ds_write at the row's rank, and every pending lane reads
  // its own rank back.  With one wave per SIMD an instruction of any kind costs an issue slot, so this is
  // where the kernel's time was.  Rows beyond an odd count are read twice (one spare entry behind the
  // list); their results land behind the ranks anybody reads.
  // TAIL (dim % 32 != 0, e.g. 100, 784): the sequential scalar chain over the tail elements (dot.s:35-43)
  // runs in lane 0 of each half -- the query element comes from a readlane (once per element, shared by all
  // rows of the chunk), the row's elements walk down to lane 0 with one wave_shl DPP move per step.
  // raw distances of the pending rows of ranks [first, first + count) of the list in s_slot (one spare entry behind its
  // last rank), two rows per wave instruction, into s_res by rank.  `first` is even.
  // ALL: every one of the U row slots of a round is loaded and summed, the ones past the range as copies of its last row
  // (their sums go to the dump) -- no branch between the rows, so that all loads of a round are in flight at once.  (With
  // the rows conditional, as the one-wave kernels have them, the compiler here -- inside the multi-wave kernels -- placed
  // each row's wait and arithmetic right behind its three loads: one memory round trip PER ROW, three in sequence for a
  // wave's six rows ahead.)
  template <bool TAIL, bool ALL = false>
  __device__ __forceinline__ void rows_range(const SearchArgs &a, const uint32_t *s_slot, float *s_res, int first, int count,
                                             int lane) {
    const int L = lane & 31, half = lane >> 5;
    const char *baseL = reinterpret_cast<const char *>(a.slab) + L * 16;
    const uint32_t row_bytes = a.ld * 4u;
    const int cnt = first + count;
    for (int c0 = first; c0 < cnt; c0 += 2 * U) {
      const int m = cnt - c0 < 2 * U ? cnt - c0 : 2 * U;
      const int h0 = (m + 1) >> 1;  // rows of half 0; half 1 takes the other m - h0 (+ the spare when m is odd)
      const int base = c0 + (half ? h0 : 0);
#ifdef SDB_STAMPS
      unsigned long long t0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      uint32_t sl[U];
#pragma unroll
      for (int u = 0; u < U; u++) sl[u] = s_slot[base + (ALL ? (u < h0 ? u : h0 - 1) : u)];
      float4 y[U][NGR];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (ALL || u < h0) {
          const float4 *r4 = reinterpret_cast<const float4 *>(baseL + (uint64_t)sl[u] * row_bytes);
#pragma unroll
          for (int g = 0; g < NG; g++) y[u][g] = r4[g * 32];
        }
      float yt[TAIL ? U : 1];
      if constexpr (TAIL) {
        const char *baseT = reinterpret_cast<const char *>(a.slab) + (NG * 128 + L) * 4;
#pragma unroll
        for (int u = 0; u < U; u++) {
          yt[u] = 0.0f;
          if (ALL || u < h0) yt[u] = *reinterpret_cast<const float *>(baseT + (uint64_t)sl[u] * row_bytes);
        }
      }
#ifdef SDB_STAMPS
      unsigned long long t1 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      unsigned long long t2 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      float *wp = (L == 0) ? s_res + base : s_res + kHopSlots;  // the other lanes write to a dump
      if constexpr (!TAIL) {
#pragma unroll
        for (int u = 0; u < U; u++)
          if (ALL || u < h0) {
            float acc = 0.0f;
#pragma unroll
            for (int g = 0; g < NG; g++) acc = chain4<L2>(acc, xq[g], y[u][g]);
            const float r = asm_reduce(acc, 0.0f, lane);
            if constexpr (ALL) (u < h0 ? wp : s_res + kHopSlots)[u] = r;
            else wp[u] = r;
          }
      } else {
        float acc[U], t[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          acc[u] = 0.0f, t[u] = 0.0f;
          if (ALL || u < h0) {
#pragma unroll
            for (int g = 0; g < NG; g++) acc[u] = chain4<L2>(acc[u], xq[g], y[u][g]);
          }
        }
        for (uint32_t i = 0; i < a.tail; i++) {
          const float xi = rlf(xt, (int)i);  // both halves hold the query's tail in lanes 0..31
#pragma unroll
          for (int u = 0; u < U; u++) {
            t[u] = chain1<L2>(t[u], xi, yt[u]);  // lane 0 of each half: element i of its row
            yt[u] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(yt[u]), 0x130, 0xf, 0xf, true));
          }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
          if (ALL || u < h0) {
            const float r = asm_reduce(acc[u], t[u], lane);
            if constexpr (ALL) (u < h0 ? wp : s_res + kHopSlots)[u] = r;
            else wp[u] = r;
          }
      }
#ifdef SDB_STAMPS
      unsigned long long t3 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      st[0] += t1 - t0, st[1] += t2 - t1, st[2] += t3 - t2;  // issue, wait, compute
#endif
    }
  }

  template <bool TAIL>
  __device__ __forceinline__ float hop_fast(const SearchArgs &a, uint32_t nb, uint64_t pend, int lane) {
    const int cnt = __popcll(pend);
    const bool mine = (pend >> lane) & 1ull;
    const uint32_t rank =
        __builtin_amdgcn_mbcnt_hi((uint32_t)(pend >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pend, 0u));
    uint32_t *s_slot = hs;
    float *s_res = reinterpret_cast<float *>(hs + kHopSlots);
    if (mine) {
      s_slot[rank] = nb;
      if ((int)rank == cnt - 1) s_slot[cnt] = nb;  // the spare entry
    }
    wave_lds_sync();
    rows_range<TAIL>(a, s_slot, s_res, 0, cnt, lane);
    wave_lds_sync();
    return mine ? metric_finish(s_res[rank], a.metric) : 0.0f;
  }

  // distances of the new neighbours of one hop: lane j (bit j of pend) gets dist(query, row nb_j)
  __device__ __forceinline__ float hop(const SearchArgs &a, uint32_t nb, uint64_t pend, int lane) {
    if constexpr (NG >= 0) return a.tail ? hop_fast<true>(a, nb, pend, lane) : hop_fast<false>(a, nb, pend, lane);
    float mydist = 0.0f;
    uint64_t todo = pend;
    while (todo) {
      int jj[2 * U];
#pragma unroll
      for (int i = 0; i < 2 * U; i++) {
        if (todo) {
          jj[i] = __ffsll((unsigned long long)todo) - 1;
          todo &= todo - 1;
        } else {
          jj[i] = jj[i > 0 ? i - 1 : 0];
        }
      }
      uint32_t slot[U];
      float res[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        uint32_t s0 = rl(nb, jj[2 * u]), s1 = rl(nb, jj[2 * u + 1]);
        slot[u] = lane < 32 ? s0 : s1;
      }
      chunk(a, slot, res, lane);
#pragma unroll
      for (int u = 0; u < U; u++) {
        float d0 = metric_finish(rlf(res[u], 0), a.metric);
        float d1 = metric_finish(rlf(res[u], 32), a.metric);
        if (lane == jj[2 * u]) mydist = d0;
        if (lane == jj[2 * u + 1]) mydist = d1;
      }
    }
    return mydist;
  }
};

// The same store walked by a WORKGROUP per query, for calls with few queries (a single REST request is one query,
// vamana.go:278-310): one wave per query leaves most of the chip idle and a hop costs two dependent memory round trips
// plus the issue time of ~50 rows' loads on one SIMD.  Here wave 0 is the walker -- search_body as it stands: candidate
// array, visited set, AddWithLimit -- and a hop's pending rows are split over all W waves of the workgroup: the walker
// publishes the rank-compacted slot list (the one hop_fast builds anyway), every wave takes a contiguous share of it,
// two rows per instruction, and leaves the raw sums at the rows' ranks; the walker reads them back.  Same pairs, same
// arithmetic, same bits, same visit order; two workgroup barriers per hop.
constexpr uint32_t kWideAheadPad = 24;  // spare entry + rows_range's dump behind the 64 positions (kHopSlots = 64 + U, U <= 8)
// The walker talks to the other waves in WORDS: it fills the fields, then raises `word` by one; a wave that is through
// with the last word spins on it.  The walker sends a word only when every other wave has taken the one before -- it
// has waited for their completion counts by then -- so a field is never overwritten under a reader.
constexpr uint32_t kWordShare = 1u;  // `cnt` pending rows are listed: take your share (one barrier behind it)
constexpr uint32_t kWordAhead = 2u;  // `ahead` is the adjacency row of the candidate the walk expands next: work ahead on it
constexpr uint32_t kWordDone = 3u;   // the walk is over
struct WideShared {
  uint32_t word;       // words sent, ever (monotonic)
  uint32_t kind;       // what the last one says
  uint32_t cnt;        // kWordShare: pending rows of the hop
  uint32_t mark_on;    // kWordAhead: != 0: the marker wave runs the visited-set test of that row's neighbours
  unsigned long long ahead;  // kWordAhead: the row
  uint32_t ahead_done; // rows ahead the computing waves are through with, times their number (monotonic)
  uint32_t mark_seq;   // kWordAhead words the marker is through with (monotonic)
  unsigned long long mark_pend;  // the neighbours (by edge position) that were new to the visited set
  uint32_t mark_cell[64];        // where each of those went in the table
  uint32_t ahead_slot[64 + kWideAheadPad];  // that row's slots by edge position
  float ahead_res[64 + kWideAheadPad];      // raw sums of distFn(query, neighbour) by edge position
};

// pairs of rows per round of loads: a wave's share of a hop is at most 64 / W rows (rounded up to even), so 32 / W pairs
// hold it -- but its share of the distances AHEAD is 6 rows at W = 16 (14 waves compute): three pairs where the registers
// allow it (rows of up to 512 floats; 128 registers per wave at 16 waves)
// -- euclidean from 384 floats up stays at two: its separately rounded differences are live beside the rows, and the
// third pair spilled (1 register at 384 floats, 33 .. 69 at 512; tools/kernel_table.py --check holds every walk kernel
// to zero scratch)
template <int NG, int W, bool L2>
struct WidePairs {
  static constexpr int kMin = 32 / W > 0 ? 32 / W : 1;
  static constexpr int value = (kMin < 3 && (NG <= 2 || (NG <= 4 && !L2))) ? 3 : kMin;
};
template <int NG, bool L2, int W>
struct PlainWideDist : PlainDist<NG, L2, true, WidePairs<NG, W, L2>::value> {
  using Base = PlainDist<NG, L2, true, WidePairs<NG, W, L2>::value>;
  // A call of few queries has bandwidth to spare and a dependent chain to shorten.  Once a hop's distances are known, so
  // is the candidate the walk expands next (search_body, Dist::kSpeculate: 99 % of the hops expand exactly it) -- BEFORE
  // the hop's points are inserted.  The walker names it there (a.wide_pull = 2), and while it inserts
  //   * the marker (wave 1) runs the visited-set test of that candidate's neighbours on the walker's table,
  //   * the computing waves (2 .. W-1) COMPUTE the raw distances to all of its neighbours, by edge position,
  // so that the next hop starts with its adjacency row in a register, its CheckAndVisit verdict and its distances in LDS:
  // what is left of it is AddWithLimit.  Nothing is decided on a guess: should the walk go elsewhere (a tie, a NaN, the
  // start node's overflow list), the marks are taken back, the rows are shared out the plain way (kWordShare), and the
  // guess has cost traffic.  Same pairs, same arithmetic, same bits, same visit order.
  static constexpr bool kSpeculate = true;
  static constexpr size_t kLdsBytes = Base::kLdsBytes + sizeof(WideShared);
  // Wave 1 is the MARKER (W >= 4): CheckAndVisit marks before any distance is looked at (distset.go:174) and nothing else
  // touches the set between the naming and the next hop, so the marks are exactly the ones the walker's own test would
  // have made.  Waves kFirstAhead .. W-1 compute the distances ahead.
  static constexpr int kFirstAhead = W >= 4 ? 2 : 1;
  static constexpr int kAheadWaves = W - kFirstAhead;
  static constexpr int kAheadPer = ((64 + kAheadWaves - 1) / kAheadWaves + 1) & ~1;  // rows ahead per computing wave (even)
  WideShared *sh;
  int wave;
  bool hit_cur;         // walker: this hop expands the candidate named one hop ago, and its distances were computed ahead
  bool ahead_computed;  // walker: the last hop named a row
  bool marked;          // walker: ... and the marker was sent through it
  uint32_t ahead_expect;  // walker: value of sh->ahead_done once every computing wave is through with what was named
  uint32_t mark_expect;   // walker: value of sh->mark_seq once the marker is
  uint32_t words;         // words sent (walker) / taken (the others)
  SearchArgs const *args_;
#ifdef SDB_STAMPS
  unsigned long long t_named = 0;
#endif
  __device__ __forceinline__ void init_wave(const SearchArgs &a, uint32_t q, int lane, int w, float *lds) {
    Base::init(a, q, lane, lds);  // every wave keeps the query in its registers
    sh = reinterpret_cast<WideShared *>(reinterpret_cast<char *>(lds) + Base::kLdsBytes);
    wave = w, args_ = &a;
    hit_cur = false, ahead_computed = false, marked = false, ahead_expect = 0, mark_expect = 0, words = 0;
    if (w == 0 && lane == 0) sh->word = 0, sh->ahead_done = 0, sh->mark_seq = 0, sh->mark_on = 0;
  }
  __device__ __forceinline__ void speculation(bool use, const uint32_t *) { hit_cur = use && ahead_computed; }

  // ---- the walker (wave 0)
  // every other wave is through with the last word (and spins for the next)
  __device__ __forceinline__ void quiesce() {
    while (__atomic_load_n(&sh->ahead_done, __ATOMIC_RELAXED) != ahead_expect) __builtin_amdgcn_s_sleep(1);
    if (kFirstAhead == 2)
      while (__atomic_load_n(&sh->mark_seq, __ATOMIC_RELAXED) != mark_expect) __builtin_amdgcn_s_sleep(1);
  }
  __device__ __forceinline__ void send(uint32_t kind, int lane) {  // the fields are written (lane 0)
    words++;
    if (lane == 0) {
      sh->kind = kind;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __atomic_store_n(&sh->word, words, __ATOMIC_RELAXED);
    }
  }
  // The walker's word on the row to work ahead on, once per hop and AFTER the hop's distances are known (search_body).
  __device__ __forceinline__ void ahead(const SearchArgs &a, const uint32_t *rowp, int lane, bool markable) {
    ahead_computed = a.wide_pull == 2 && rowp != nullptr;
    marked = ahead_computed && markable && kFirstAhead == 2;
    if (!ahead_computed) return;
    quiesce();
#ifdef SDB_STAMPS
    t_named = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    ahead_expect += (uint32_t)kAheadWaves;
    if (kFirstAhead == 2) mark_expect++;
    if (lane == 0) sh->mark_on = marked ? 1u : 0u, sh->ahead = reinterpret_cast<unsigned long long>(rowp);
    send(kWordAhead, lane);
  }
  // what the marker found for the row named last: the neighbours that were new to the set, and where they went
  __device__ __forceinline__ void take_marks(int lane, uint64_t &mask, uint32_t &cell) {
    while (__atomic_load_n(&sh->mark_seq, __ATOMIC_RELAXED) != mark_expect) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    mask = sh->mark_pend;
    cell = sh->mark_cell[lane];
    marked = false;
  }
  // this wave's share of `cnt` pending rows: contiguous ranks, an even number per wave so that only the list's last row
  // can be the odd one out (its spare entry sits behind the list)
  __device__ __forceinline__ void share(const SearchArgs &a, int cnt, int lane) {
    const int per = (((cnt + W - 1) / W) + 1) & ~1;
    const int first = wave * per;
    const int count = cnt - first < per ? cnt - first : per;
    if (count <= 0) return;
    uint32_t *s_slot = this->hs;
    float *s_res = reinterpret_cast<float *>(this->hs + Base::kHopSlots);
    if (a.tail) this->template rows_range<true, true>(a, s_slot, s_res, first, count, lane);
    else this->template rows_range<false, true>(a, s_slot, s_res, first, count, lane);
  }
  __device__ __forceinline__ float hop(const SearchArgs &a, uint32_t nb, uint64_t pend, int lane) {
    const int cnt = __popcll(pend);
    const bool mine = (pend >> lane) & 1ull;
    if (hit_cur) {  // this row's distances were computed while the last hop's points were inserted: by edge position
      hit_cur = false;
#ifdef SDB_STAMPS  // st[2]: from naming the row to needing its distances (the walker's own work); st[0]: waiting for them
      const unsigned long long w0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      this->st[2] += w0 - t_named;
#endif
      while (__atomic_load_n(&sh->ahead_done, __ATOMIC_RELAXED) != ahead_expect) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const float raw = sh->ahead_res[lane];
#ifdef SDB_STAMPS
      asm volatile("" ::"v"(raw));
      const unsigned long long w1 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      this->st[0] += w1 - w0;
#endif
      return mine ? metric_finish(raw, a.metric) : 0.0f;
    }
    const uint32_t rank =
        __builtin_amdgcn_mbcnt_hi((uint32_t)(pend >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pend, 0u));
    uint32_t *s_slot = this->hs;
    float *s_res = reinterpret_cast<float *>(this->hs + Base::kHopSlots);
    quiesce();  // the others may still be at a row the walk did not go to
    if (mine) {
      s_slot[rank] = nb;
      if ((int)rank == cnt - 1) s_slot[cnt] = nb;  // the spare entry
    }
    if (lane == 0) sh->cnt = (uint32_t)cnt;
    send(kWordShare, lane);
    share(a, cnt, lane);
    __syncthreads();  // every share has been written
    return mine ? metric_finish(s_res[rank], a.metric) : 0.0f;
  }
  __device__ __forceinline__ void skip(int) { hit_cur = false; }  // a chunk without a new neighbour: nothing to share
  __device__ __forceinline__ void finish(int lane) {
    quiesce();
    send(kWordDone, lane);
  }

  // ---- waves 1 .. W-1
  // a computing wave's share of the distances to the neighbours of the candidate ahead
  __device__ __forceinline__ void compute_ahead(const SearchArgs &a, const uint32_t *rowp, int lane) {
    const uint32_t nb = rowp[lane];
    const uint64_t m = __ballot(nb != kNoSlot);
    const int deg = __popcll(m);  // rows are padded with kNoSlot behind their edges
    sh->ahead_slot[lane] = nb != kNoSlot ? nb : 0u;  // every computing wave writes the same words
    if (lane == deg - 1) sh->ahead_slot[deg] = nb;  // the spare entry behind the list
    wave_lds_sync();
    const int first = (wave - kFirstAhead) * kAheadPer;
    const int count = deg - first < kAheadPer ? deg - first : kAheadPer;
    if (count > 0) {
      if (a.tail) this->template rows_range<true, true>(a, sh->ahead_slot, sh->ahead_res, first, count, lane);
      else this->template rows_range<false, true>(a, sh->ahead_slot, sh->ahead_res, first, count, lane);
    }
    wave_lds_sync();  // the sums before the count
    if (lane == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      atomicAdd(&sh->ahead_done, 1u);
    }
  }
  template <class HV>  // HV: the walker's visited set (its table is `tab`)
  __device__ __forceinline__ void serve(const SearchArgs &a, int lane, uint32_t *tab) {
#ifdef SDB_STAMPS  // per wave: waiting for a word, the shares, the work ahead
    unsigned long long hs_word = 0, hs_share = 0, hs_work = 0, hs_n = 0;
#define SDB_HSTAMP(acc)                                         \
  {                                                             \
    unsigned long long _t = __builtin_amdgcn_s_memtime();       \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    acc += _t - hs_t0;                                          \
    hs_t0 = _t;                                                 \
  }
    unsigned long long hs_t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define SDB_HSTAMP(acc)
#endif
    for (;;) {
      words++;
      while (__atomic_load_n(&sh->word, __ATOMIC_RELAXED) != words) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      SDB_HSTAMP(hs_word)
      const uint32_t kind = sh->kind;
      if (kind == kWordDone) {
#ifdef SDB_STAMPS
        if (lane == 0 && a.tr_visit && a.visit_cap >= 44) a.tr_visit[(size_t)blockIdx.x * a.visit_cap + 28 + wave] = hs_work;
        if (lane == 0 && (wave == 1 || wave == 2) && a.tr_visit && a.visit_cap >= 24) {
          uint64_t *o = a.tr_visit + (size_t)blockIdx.x * a.visit_cap + (wave == 1 ? 12 : 18);
          o[0] = 0, o[1] = hs_share, o[2] = hs_word, o[3] = hs_work, o[4] = hs_n, o[5] = 0;
        }
#endif
        return;
      }
      if (kind == kWordShare) {
        share(a, (int)sh->cnt, lane);
        __syncthreads();
        SDB_HSTAMP(hs_share)
        continue;
      }
#ifdef SDB_STAMPS
      hs_n++;
#endif
      const uint32_t *ahead = reinterpret_cast<const uint32_t *>(sh->ahead);
      if (kFirstAhead == 2 && wave == 1) {
        if (sh->mark_on) {
          const uint32_t nb = ahead[lane];
          uint32_t cell;
          const bool isnew = HV::claim(tab, nb != kNoSlot, nb, cell);
          const uint64_t pend = __ballot(isnew);
          sh->mark_cell[lane] = cell;
          if (lane == 0) sh->mark_pend = pend;
          wave_lds_sync();
        }
        n_ahead++;
        if (lane == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __atomic_store_n(&sh->mark_seq, n_ahead, __ATOMIC_RELAXED);
        }
      } else {
        compute_ahead(a, ahead, lane);
      }
      SDB_HSTAMP(hs_work)
    }
  }
  uint32_t n_ahead = 0;  // marker: kWordAhead words it is through with
};

// Fitted product quantizer: dist = sum_i lut[i*K + code_i], plain fp32 adds in index order
// (product.go:271-275).  One lane per neighbour: all new neighbours of a hop in one pass.
struct PQDist {
  // (the same naming of the next hop's row, with its code rows fetched ahead: 0.419 against 0.359 ms per batch at
  // 4M x 768, M = 8 -- a one-wave hop is bound by its instruction count, not by the fetch; measured and removed)
  static constexpr bool kSpeculate = false;
  static constexpr bool kHasStamps = false;
  static constexpr bool kPointDistances = false;  // LUT distance != the centroid-pair distance of the prunes
  const float *lut;  // this query's [M][K] table, in LDS or in global memory
  __device__ __forceinline__ void init(const SearchArgs &a, uint32_t q, int lane, float *lds) {
    const float *g = a.pq_lut + (size_t)q * a.pq_M * a.pq_K;
    if (a.pq_lut_in_lds) {
      for (uint32_t i = lane; i < a.pq_M * a.pq_K; i += 64) lds[i] = g[i];
      __syncthreads();
      lut = lds;
    } else {
      lut = g;
    }
  }
  // dist = 0; dist += lut[i][code_i] for i = 0 .. M-1, plain fp32 adds in index order (product.go:271-275), over the
  // code row at `c` (M bytes; rows are M apart from an aligned base, so whole words when M is a multiple of four)
  __device__ __forceinline__ float sum_row(const SearchArgs &a, const uint8_t *__restrict__ c) const {
    float dist = 0.0f;
    const uint32_t K = a.pq_K;
    if ((a.pq_M & 3u) == 0) {
      const uint32_t *__restrict__ w = reinterpret_cast<const uint32_t *>(c);
      for (uint32_t i = 0; i < a.pq_M; i += 4) {
        const uint32_t v = w[i >> 2];
        dist += lut[(i + 0) * K + (v & 0xFF)];
        dist += lut[(i + 1) * K + ((v >> 8) & 0xFF)];
        dist += lut[(i + 2) * K + ((v >> 16) & 0xFF)];
        dist += lut[(i + 3) * K + (v >> 24)];
      }
    } else {
      for (uint32_t i = 0; i < a.pq_M; i++) dist += lut[i * K + c[i]];
    }
    return dist;
  }
  __device__ __forceinline__ float sum(const SearchArgs &a, uint32_t slot) const {
    return sum_row(a, a.pq_codes + (size_t)slot * a.pq_M);
  }
  __device__ __forceinline__ float one(const SearchArgs &a, uint32_t s, int lane) { return sum(a, s); }
  __device__ __forceinline__ void speculation(bool, const uint32_t *) {}
  __device__ __forceinline__ void ahead(const SearchArgs &, const uint32_t *, int, bool) {}
  __device__ __forceinline__ void skip(int) {}
  // The chunk's code rows.  With the neighbours' codes stored behind the adjacency row (SearchArgs::adj_codes) lane j's
  // row is at a fixed place next to edge j: it is asked for TOGETHER with the row of ids -- one round trip per hop, 64 M
  // contiguous bytes -- instead of after it, by slot (64 scattered M-byte reads, a 64-byte sector each).  The start
  // node's overflow chunks and the filter's seeds are no adjacency rows: those gather by slot.
  const uint8_t *row_codes = nullptr;  // this lane's code row of the chunk being expanded, or NULL: by slot
  __device__ __forceinline__ void begin_row(const SearchArgs &a, const uint32_t *rowp, int lane) {
    row_codes = nullptr;
    pre_ready = false;
    if (a.adj_codes && rowp >= a.adj && rowp < a.adj + (size_t)a.adj_rows * kAdjStride) {  // not a chunk of the overflow list
      const size_t e0 = (size_t)(rowp - a.adj);  // = row * 64
      row_codes = a.adj_codes + (e0 + (size_t)lane) * a.pq_M;
      if (a.pq_M == 8) {
        pre = *reinterpret_cast<const uint2 *>(row_codes), pre_ready = true;
      } else if (a.pq_M == 16) {
        prew[0] = *reinterpret_cast<const uint4 *>(row_codes), pre_ready = true;
      } else if (a.pq_M == 32) {
        prew[0] = reinterpret_cast<const uint4 *>(row_codes)[0], prew[1] = reinterpret_cast<const uint4 *>(row_codes)[1];
        pre_ready = true;
      }
    }
  }
  uint4 prew[2];  // M = 16 / 32: the lane's code row, asked for with the row of ids
  __device__ __forceinline__ float add4(float dist, uint32_t v, uint32_t i, uint32_t K) const {
    dist += lut[(i + 0) * K + (v & 0xFF)];
    dist += lut[(i + 1) * K + ((v >> 8) & 0xFF)];
    dist += lut[(i + 2) * K + ((v >> 16) & 0xFF)];
    dist += lut[(i + 3) * K + (v >> 24)];
    return dist;
  }
  __device__ __forceinline__ float sum16(float dist, const uint4 &v, uint32_t i, uint32_t K) const {
    dist = add4(dist, v.x, i, K), dist = add4(dist, v.y, i + 4, K);
    dist = add4(dist, v.z, i + 8, K), dist = add4(dist, v.w, i + 12, K);
    return dist;
  }
  // M == 8 (the documented configuration): the 8 code bytes of every neighbour are fetched as one 8-byte
  // load BEFORE the visited-set test, so the code fetch and the test-and-set round trip overlap; codes of
  // neighbours that turn out to be already visited are simply not used (8 bytes each).
  uint2 pre;
  bool pre_ready = false;
  __device__ __forceinline__ void prefetch(const SearchArgs &a, uint32_t nb, bool valid) {
    if (pre_ready) return;  // came with the row
    pre = make_uint2(0u, 0u);
    if (a.pq_M == 8 && valid && !row_codes) pre = *reinterpret_cast<const uint2 *>(a.pq_codes + (size_t)nb * 8);
  }
  // M == 8 with the code row at hand (it came with the row of ids): the eight table entries can be asked for AHEAD of
  // the visited-set test, whose first LDS round trip then covers theirs, and summed after it (k_greedy_search_pq2)
  __device__ __forceinline__ bool ahead8(const SearchArgs &a) const { return a.pq_M == 8 && pre_ready; }
  __device__ __forceinline__ void load8(const SearchArgs &a, bool valid, float (&t)[8]) const {
    const uint32_t K = a.pq_K;
    const uint32_t x = valid ? pre.x : 0u, y = valid ? pre.y : 0u;
    t[0] = lut[0 * K + (x & 0xFF)], t[1] = lut[1 * K + ((x >> 8) & 0xFF)];
    t[2] = lut[2 * K + ((x >> 16) & 0xFF)], t[3] = lut[3 * K + (x >> 24)];
    t[4] = lut[4 * K + (y & 0xFF)], t[5] = lut[5 * K + ((y >> 8) & 0xFF)];
    t[6] = lut[6 * K + ((y >> 16) & 0xFF)], t[7] = lut[7 * K + (y >> 24)];
  }
  __device__ __forceinline__ float sum8(const float (&t)[8], uint64_t pend, int lane) {
    row_codes = nullptr, pre_ready = false;
    if (!((pend >> lane) & 1ull)) return 0.0f;
    float dist = 0.0f;  // the same sequential adds in index order (product.go:271-275)
#pragma unroll
    for (int i = 0; i < 8; i++) dist += t[i];
    return dist;
  }
  __device__ __forceinline__ float hop(const SearchArgs &a, uint32_t nb, uint64_t pend, int lane) {
    const uint8_t *rc = row_codes;
    const bool ready = pre_ready;
    row_codes = nullptr, pre_ready = false;  // one chunk's worth (the seeds of a filtered search call hop without begin_row)
    if (!((pend >> lane) & 1ull)) return 0.0f;
    if (a.pq_M != 8) {
      if (ready && a.pq_M == 16) return sum16(0.0f, prew[0], 0, a.pq_K);
      if (ready && a.pq_M == 32) return sum16(sum16(0.0f, prew[0], 0, a.pq_K), prew[1], 16, a.pq_K);
      return rc ? sum_row(a, rc) : sum(a, nb);
    }
    float dist = 0.0f;  // same sequential adds in index order (product.go:271-275)
    const uint32_t K = a.pq_K;
    dist += lut[0 * K + (pre.x & 0xFF)];
    dist += lut[1 * K + ((pre.x >> 8) & 0xFF)];
    dist += lut[2 * K + ((pre.x >> 16) & 0xFF)];
    dist += lut[3 * K + (pre.x >> 24)];
    dist += lut[4 * K + (pre.y & 0xFF)];
    dist += lut[5 * K + ((pre.y >> 8) & 0xFF)];
    dist += lut[6 * K + ((pre.y >> 16) & 0xFF)];
    dist += lut[7 * K + (pre.y >> 24)];
    return dist;
  }
};

// Fitted product quantizer whose per-query table does not fit beside a one-wave walk (M x K x 4 bytes > 64 KB:
// M = 192 at d = 768 is 192 KB).  Round 2 left that table in global memory and every lookup was a 4-byte gather from
// a 200 MB working set: 125 k QPS at 10M x 768 against 450 k for the full-precision walk.  Here ONE query owns a
// workgroup of four waves, one per SIMD, and with it a CU's LDS and register file:
//   * the table of sub-quantizer i (K floats) lives in LDS, or in the registers of one of the four waves (four
//     registers per table: lane l holds entries l, 64 + l, 128 + l, 192 + l; a lookup is four ds_bpermute and a
//     select -- the register file as a second, larger LDS);
//   * wave w owns the contiguous index range [w M/4, (w + 1) M/4): its first NL tables in LDS, its last RT in its
//     registers (M = 4 (NL + RT)).  Wave 0 is the walker: it runs search_body exactly as the one-wave kernels do
//     (candidate array, visited set, AddWithLimit).  Per adjacency chunk it publishes the row pointer, every wave
//     reads the row (lane j = edge j) and fetches its range of the neighbours' code bytes while the walker runs the
//     visited-set test; then all four look their table entries up in parallel;
//   * the sum is the reference's: dist = 0; dist += lut[i][code_i] for i = 0 .. M-1, plain fp32 adds in index order
//     (product.go:271-275).  The waves take turns -- wave 0 adds its M/4 values to 0 and hands the partial sums on
//     through LDS, wave 1 adds its own, ... -- so the adds happen in exactly the sequential order.
// Barriers per chunk (all four waves): B0 row published, B1 pending mask published, then one per wave's turn.
struct PQWideShared {
  unsigned long long rowp;  // mode 1: the adjacency chunk to expand
  uint32_t mode;            // 0: the walk is over, 1: expand rowp, 2: the single point `slot` (start node), 3: (split form) the merger inserts what it holds and names the next node
  uint32_t slot;
  unsigned long long pend;  // lanes whose neighbour passed CheckAndVisit
  // the split form (k_greedy_search_pqw<..., SPLIT>: the candidate array lives in a helper wave, the merger)
  uint32_t named;  // walker -> merger, with mode 1 / 3: the node the walk went to after the chunk the merger still holds, kNoSlot: the merger picks
  uint32_t next;   // merger -> walker, mode 3: the node it picked (marked), kNoSlot: no unvisited entry left
  uint32_t f1;     // merger -> walker: the first unvisited entry behind the node the walk went to, kNoSlot: none
  float f1d;
  float psum[64];           // partial sums on their way from wave to wave, by lane
};
constexpr uint32_t kPqwSharedWords = sizeof(PQWideShared) / 4;

// NL / RT: tables per wave in LDS / in registers (NL + RT a multiple of 16); W: waves per query, M = W (NL + RT).
// W = 8 (two waves per SIMD, 256 registers each) carries M = 384 with the per-wave layout of M = 192: round 3's
// four-wave form of it kept 64 tables = 256 registers per wave and spilled 80 more.
template <int NREG, int FEW = 2>  // (defined with the candidate array's other operations below)
__device__ __forceinline__ void add_with_limit_merge(uint32_t (&cid)[NREG], float (&cd)[NREG], int &len, int cap, uint32_t idreg,
                                                     float mydist, uint64_t pd, int lane, uint32_t *scratch
#ifdef SDB_STAMPS
                                                     , unsigned long long *mst = nullptr
#endif
);
template <int NL, int RT, int W = 4>
struct PQWideDist {
  // Fetching ahead (search_body, Dist::kSpeculate) was built for this walk too -- every wave fetched the likely next
  // row at the start of a hop and its neighbours' codes at the end of it, so that 70 % of the hops started with both in
  // registers -- and measured no gain at two queries per CU (M = 192, 1M x 768: 1.089 ms per batch against 1.04 without,
  // twice the code traffic): the other query's waves already fill the waits.  Removed.
  static constexpr bool kSpeculate = false;
  static constexpr bool kHasStamps = false;
  static constexpr bool kPointDistances = false;
  static constexpr int MS = NL + RT;  // sub-quantizers per wave; M = 4 MS
  static constexpr int RTR = RT > 0 ? RT : 1;
  PQWideShared *sh;
  const float *lds_lut;  // this wave's LDS-resident tables, [NL][K]
  uint32_t K, lo;
  int wave;
  float T[RTR][4];    // register-resident tables
  // (a native vector type: copies of HIP's uint4 struct between members become 16-byte memcpys that keep the whole
  // policy object -- register tables included -- in scratch memory)
  typedef uint32_t code16 __attribute__((ext_vector_type(4)));
  code16 cw[MS / 16];  // the wave's range of the lane's neighbour's code bytes

  // All four waves: tables in, from the query's [M][K] block that pq_build_lut left in global memory.  (Computing the
  // entries here instead -- no 200 MB block written and read back per batch -- was built and measured: the loads and
  // chains of the build share the walk's register budget, and the two-per-CU variant went from 1.04 to 1.23 ms per
  // batch with a loop per entry, to 1.89 ms with the loads batched; removed.  The block costs 0.12 ms per batch.)
  // `before`: tables of the waves in front of this one in the workgroup's LDS block (w * NL when every wave has the same
  // split; the walker of the eight-wave form keeps more of its tables in LDS, k_greedy_search_pqw)
  __device__ __forceinline__ void init_wave(const SearchArgs &a, uint32_t q, int lane, int w, float *lut_lds, PQWideShared *shared,
                                            uint32_t before) {
    sh = shared, wave = w, K = a.pq_K, lo = (uint32_t)w * MS;

    float *dst = lut_lds + (size_t)before * K;
    lds_lut = dst;
    const float *g = a.pq_lut + ((size_t)q * a.pq_M + lo) * K;
    for (uint32_t i = lane; i < NL * K; i += 64) dst[i] = g[i];
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const uint32_t e = c * 64 + lane;
        T[t][c] = e < K ? g[(size_t)(NL + t) * K + e] : 0.0f;
      }
  }
  __device__ __forceinline__ void load_codes(const SearchArgs &a, uint32_t nb) {
    const uint32_t s = nb == kNoSlot ? 0u : nb;  // lanes without a neighbour read row 0 and are never looked at
    const code16 *cp = reinterpret_cast<const code16 *>(a.pq_codes + (size_t)s * a.pq_M + lo);
#pragma unroll
    for (int j = 0; j < MS / 16; j++) cw[j] = cp[j];
  }
  template <int I>
  __device__ __forceinline__ uint32_t code_at() const {  // code byte I of the wave's range
    const code16 v = cw[I / 16];
    constexpr int wsel = (I / 4) % 4;
    const uint32_t word = wsel == 0 ? v.x : wsel == 1 ? v.y : wsel == 2 ? v.z : v.w;
    return (word >> (8 * (I % 4))) & 0xFFu;
  }
  template <int I>
  __device__ __forceinline__ float lookup() const {
    const uint32_t e = code_at<I>();
    if constexpr (I < NL) {
      return lds_lut[(size_t)I * K + e];
    } else {
      constexpr int t = I - NL;
      const int addr = (int)((e & 63u) << 2);
      const float v0 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(T[t][0])));
      const float v1 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(T[t][1])));
      const float v2 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(T[t][2])));
      const float v3 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(T[t][3])));
      const uint32_t c = e >> 6;
      return c == 0 ? v0 : c == 1 ? v1 : c == 2 ? v2 : v3;
    }
  }
  template <int I>
  __device__ __forceinline__ void lookups(float (&val)[MS]) const {
    if constexpr (I < MS) {
      val[I] = lookup<I>();
      lookups<I + 1>(val);
    }
  }
  // The turns: wave w adds its MS values, in index order, on top of what the waves before it left.  Returns the
  // finished sums (valid after the last turn) by lane.
  __device__ __forceinline__ float turns(const float (&val)[MS]) const {
    const int lane_ = threadIdx.x & 63;
#pragma unroll
    for (int stage = 0; stage < W; stage++) {
      if (stage == wave) {
        float acc = stage == 0 ? 0.0f : sh->psum[lane_];
#pragma unroll
        for (int i = 0; i < MS; i++) acc += val[i];  // product.go:271-275: dist += dists[i*K + code[i]]
        sh->psum[lane_] = acc;
      }
      __syncthreads();  // B2 + stage
    }
    return sh->psum[lane_];
  }

  // ---- the walker's side (wave 0): the policy interface search_body calls
  __device__ __forceinline__ void speculation(bool, const uint32_t *) {}
  __device__ __forceinline__ void ahead(const SearchArgs &, const uint32_t *, int, bool) {}
  uint32_t post_named = kNoSlot;  // split form: what begin_row tells the merger about the chunk before this one
  __device__ __forceinline__ void begin_row(const SearchArgs &, const uint32_t *rowp, int lane) {
    if (lane == 0) sh->rowp = reinterpret_cast<unsigned long long>(rowp), sh->mode = 1u, sh->named = post_named;
    __syncthreads();  // B0
  }
  // split form: the merger inserts the chunk it holds, picks the first unvisited entry and says which (B0, B1)
  __device__ __forceinline__ uint32_t ask_next(int lane) {
    if (lane == 0) sh->mode = 3u, sh->named = kNoSlot;
    __syncthreads();  // B0
    __syncthreads();  // B1: the answer is there
    return sh->next;
  }
  __device__ __forceinline__ void prefetch(const SearchArgs &a, uint32_t nb, bool) { load_codes(a, nb); }
  __device__ __forceinline__ void skip(int lane) {
    if (lane == 0) sh->pend = 0ull;
    __syncthreads();  // B1: nobody passed CheckAndVisit, the chunk ends here for every wave
  }
  __device__ __forceinline__ float hop(const SearchArgs &, uint32_t, uint64_t pend, int lane) {
    if (lane == 0) sh->pend = pend;
    __syncthreads();  // B1
    float val[MS];
    lookups<0>(val);  // the helpers did theirs while this wave ran the visited-set test
    return turns(val);
  }
  __device__ __forceinline__ float one(const SearchArgs &a, uint32_t s, int lane) {
    if (lane == 0) sh->slot = s, sh->mode = 2u, sh->named = kNoSlot;
    __syncthreads();  // B0
    load_codes(a, lane == 0 ? s : kNoSlot);
    return rlf(hop(a, s, 1ull, lane), 0);
  }
  __device__ __forceinline__ void finish(int lane) {
    if (lane == 0) sh->mode = 0u;
    __syncthreads();  // B0: the helpers leave
  }
  // ---- waves 1..3
  // MERGER (split form): this helper also owns the candidate array.  It keeps the chunk it has just helped to sum -- the
  // row's slots, the finished sums, the pending mask: every wave has them -- and runs AddWithLimit over it one round
  // LATER, after the next row's codes have been asked for and while the walker runs that row's visited-set test; then
  // it marks the node the walker went to and leaves the first unvisited entry behind it in sh->f1 before B1, which is
  // when the walker needs it: after that round's sums.  No polling: the walk's own barriers order every word.
  // (Pulling that entry's adjacency row through L2 here, as k_greedy_search_pq2's merger does, cost more than it hid:
  // 1.027 against 0.989 ms per batch at 2M x 768, M = 192 -- this wave is the one the round's second barrier waits for.)
  template <bool MERGER = false>
  __device__ __forceinline__ void serve(const SearchArgs &a, int lane, const uint32_t q = 0, uint32_t *scratch = nullptr) {
    constexpr int NREG = 2;
    uint32_t cid[MERGER ? NREG : 1];
    float cd[MERGER ? NREG : 1];
    int len = 0;
    const int cap = (int)a.search_size;
    uint32_t h_nb = kNoSlot;  // the chunk held back
    float h_d = 0.0f;
    uint64_t h_pend = 0;
    bool held = false;
    if constexpr (MERGER) {
#pragma unroll
      for (int r = 0; r < NREG; r++) cid[r] = kNoSlot, cd[r] = 0.0f;
    }
    // AddWithLimit over the chunk held back (distset.go:184-198), then the node the walk went to -- named by the walker,
    // or the first unvisited entry (search.go:66-71) -- is marked :74; answers in sh->next / f1 / f1d
    auto settle = [&](uint32_t named) {
      if constexpr (MERGER) {
        if (h_pend) add_with_limit_merge(cid, cd, len, cap, h_nb, h_d, h_pend, lane, scratch);
        held = false, h_pend = 0;
        uint64_t um[NREG];
#pragma unroll
        for (int r = 0; r < NREG; r++) um[r] = __ballot((r * 64 + lane) < len && !(cid[r] & kVisBit));
        int sel = -1;
        if (named != kNoSlot) {
#pragma unroll
          for (int r = 0; r < NREG; r++) {
            const uint64_t m = __ballot((r * 64 + lane) < len && cid[r] == named);
            if (m) sel = r * 64 + __ffsll((unsigned long long)m) - 1;
          }
        } else {
#pragma unroll
          for (int r = NREG - 1; r >= 0; r--)
            if (um[r]) sel = r * 64 + __ffsll((unsigned long long)um[r]) - 1;
        }
        uint32_t next = kNoSlot;
#pragma unroll
        for (int r = 0; r < NREG; r++)
          if (sel >= 0 && (sel >> 6) == r) {
            next = rl(cid[r], sel & 63);
            if (lane == (sel & 63)) cid[r] |= kVisBit;
            um[r] &= ~(1ull << (sel & 63));
          }
        uint32_t f1 = kNoSlot;
        float f1d = 0.0f;
#pragma unroll
        for (int r = NREG - 1; r >= 0; r--)
          if (um[r]) {
            const int s2 = __ffsll((unsigned long long)um[r]) - 1;
            f1 = rl(cid[r], s2), f1d = rlf(cd[r], s2);
          }
        if (lane == 0) sh->next = next, sh->f1 = f1, sh->f1d = f1d;
      }
    };
    for (;;) {
      __syncthreads();  // B0
      const uint32_t mode = sh->mode;
      if (mode == 0u) break;
      if (mode == 3u) {
        if constexpr (MERGER) settle(kNoSlot);
        __syncthreads();  // B1
        continue;
      }
      const uint32_t named = sh->named;
      uint32_t nb;
      if (mode == 1u) nb = reinterpret_cast<const uint32_t *>(sh->rowp)[lane];
      else nb = lane == 0 ? sh->slot : kNoSlot;
      load_codes(a, nb);
      float val[MS];
      // one query per CU (512 registers per wave): the lookups run now, for every lane, pending or not, under the
      // walker's CheckAndVisit (1.59 -> 1.47 ms per batch at 1M x 768).  The two-per-CU variant has no registers to
      // keep 48 values across the barrier -- there it cost 1.38 -> 1.89 ms in spills -- and looks up after it.
      if constexpr (NL >= 16) lookups<0>(val);
      if constexpr (MERGER)
        if (held) settle(named);
      __syncthreads();  // B1
      const uint64_t pend = sh->pend;
      if constexpr (MERGER) held = true, h_nb = nb, h_pend = pend, h_d = 0.0f;
      if (pend == 0ull) continue;
      if constexpr (NL < 16) lookups<0>(val);
      const float sum = turns(val);
      if constexpr (MERGER) h_d = sum;
    }
    if constexpr (MERGER) {
      // ---- IndexVamana.Search result copy vamana.go:293-307
      if (a.out_ids) {
        int base = 0;
#pragma unroll
        for (int r = 0; r < NREG; r++) {
          const uint32_t s = cid[r] & ~kVisBit;
          const bool ok = (r * 64 + lane) < len && s != a.start_slot;  // :294-296
          const uint64_t m = __ballot(ok);
          const int rank = base + __popcll(m & ((1ull << lane) - 1));
          if (ok && rank < (int)a.limit) {  // :297-299
            a.out_ids[(size_t)q * a.limit + rank] = a.ids[s];
            a.out_dists[(size_t)q * a.limit + rank] = cd[r];
          }
          base += __popcll(m);
        }
        const int got = base < (int)a.limit ? base : (int)a.limit;
        if (lane == 0) a.out_counts[q] = (uint32_t)got;
        for (int i = got + lane; i < (int)a.limit; i += 64)
          a.out_ids[(size_t)q * a.limit + i] = 0, a.out_dists[(size_t)q * a.limit + i] = 0.0f;
      }
    }
  }
};

// ---- the candidate array (DistSet.items, distset.go:133-138) held in registers -------------------------
// entry e lives in lane e % 64 of register e / 64; bit 31 of the slot word is the `visited` flag.

// bubble position + shift of DistSet.AddWithLimit (distset.go:189-198) for one accepted point
template <int NREG>
__device__ __forceinline__ void list_insert(uint32_t (&cid)[NREG], float (&cd)[NREG], int &len, int cap, uint32_t id,
                                            float d, int lane) {
  const bool full = (len == cap);
  const int newlen = full ? len : len + 1;  // :189-194 append, or overwrite the tail
  const int range = newlen - 1;             // entries the bubble loop compares against
  int pos = 0;                              // :196-198 stops at the first i with !(d < items[i-1])
#pragma unroll
  for (int r = 0; r < NREG; r++) {
    uint64_t m = __ballot((r * 64 + lane) < range && !(d < cd[r]));
    if (m) pos = r * 64 + 64 - __clzll(m);
  }
#pragma unroll
  for (int r = NREG - 1; r >= 0; r--) {
    // entry e-1 -> e: one v_mov_b32_dpp wave_shr:1 per register; lane 0 takes the carry (lane 63 of the
    // previous register) through the DPP `old` operand
    uint32_t c_id = 0;
    float c_d = 0.0f;
    if (r > 0) c_id = rl(cid[r - 1], 63), c_d = rlf(cd[r - 1], 63);
    const uint32_t up_id = (uint32_t)__builtin_amdgcn_update_dpp((int)c_id, (int)cid[r], 0x138, 0xf, 0xf, false);
    const float up_d =
        __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(c_d), __float_as_int(cd[r]), 0x138, 0xf, 0xf, false));
    const int e = r * 64 + lane;
    if (e > pos && e < newlen) cid[r] = up_id, cd[r] = up_d;
    else if (e == pos) cid[r] = id, cd[r] = d;
  }
  len = newlen;
}

template <int NREG>
__device__ __forceinline__ float list_tail(const float (&cd)[NREG], int cap) {
  float t = 0.0f;
#pragma unroll
  for (int r = 0; r < NREG; r++)
    if (((cap - 1) >> 6) == r) t = rlf(cd[r], (cap - 1) & 63);
  return t;
}

// AddWithLimit (distset.go:184-198) over the lanes in `pd`, IN LANE ORDER, each lane holding (idreg,
// mydist).  Only points that beat the current tail are replayed: the tail only shrinks, so a point that
// fails the current threshold can never pass a later one.
template <int NREG>
__device__ __forceinline__ void add_with_limit_lanes(uint32_t (&cid)[NREG], float (&cd)[NREG], int &len, int cap,
                                                     uint32_t idreg, float mydist, uint64_t pd, int lane) {
  while (pd) {
    bool ok = true;
    if (len == cap) ok = !(mydist > list_tail(cd, cap));  // :184 strict '>'
    const uint64_t am = __ballot(ok) & pd;
    if (!am) break;
    const int j = __ffsll((unsigned long long)am) - 1;
    const float d = rlf(mydist, j);
    const uint32_t id = rl(idreg, j);
    pd = (j == 63) ? 0ull : ((pd >> (j + 1)) << (j + 1));
    list_insert(cid, cd, len, cap, id, d, lane);
  }
}

// The same AddWithLimit over several points at once, for a FULL and SORTED candidate array (the unfiltered
// search after its first hops).  With no two equal distances among the array and the points that beat the
// tail, replaying them one by one ends in exactly one state: the `cap` smallest of (array U points), in
// order -- so that state is computed directly: every array entry moves up by the number of points below
// it, every point lands at (#entries below it + #points below it), entries pushed past `cap` vanish.  The
// scatter goes through a 2*NREG*64-word LDS scratch.  Any tie, NaN, or a not-yet-full array falls back to
// the one-by-one replay, which is the specification.  FEW: fewer candidates than this are replayed one by one as well
// (the two-precision hop hands over ~5 points per hop, not ~50: the scatter's fixed cost is not worth it then).
template <int NREG, int FEW>
__device__ __forceinline__ void add_with_limit_merge(uint32_t (&cid)[NREG], float (&cd)[NREG], int &len, int cap,
                                                     uint32_t idreg, float mydist, uint64_t pd, int lane,
                                                     uint32_t *scratch
#ifdef SDB_STAMPS
                                                     , unsigned long long *mst
#endif
) {
#ifdef SDB_STAMPS
  unsigned long long m_t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define SDB_MST(i)                                              \
  if (mst) {                                                    \
    unsigned long long _t = __builtin_amdgcn_s_memtime();       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
    mst[i] += _t - m_t0;                                        \
    m_t0 = _t;                                                  \
  }
#else
#define SDB_MST(i)
#endif
  if (len != cap) {
    return add_with_limit_lanes(cid, cd, len, cap, idreg, mydist, pd, lane);
  }
  const float tail0 = list_tail(cd, cap);
  const uint64_t cm = __ballot(!(mydist > tail0)) & pd;  // the points the replay would look at first
  SDB_MST(0)
  if (__popcll(cm) < FEW || (__ballot(mydist != mydist) & pd)) {
    add_with_limit_lanes(cid, cd, len, cap, idreg, mydist, pd, lane);
#ifdef SDB_STAMPS
    asm volatile("" ::"v"(cd[0]));
#endif
    SDB_MST(1)
    return;
  }
  const bool inC = (cm >> lane) & 1ull;
  // The pass over the points is the hot part (about a dozen points per hop): per point one readlane, and per
  // array register one compare-and-count each way.  Ties are not looked for here; they show up afterwards
  // as two elements on one position (checked below), which costs nothing per point.
  float cdx[NREG];    // the array's distances, +inf in the lanes past its end
  uint32_t up[NREG];  // per array entry: points strictly below it
#pragma unroll
  for (int r = 0; r < NREG; r++) up[r] = 0, cdx[r] = (r * 64 + lane) < cap ? cd[r] : __int_as_float(0x7f800000);
  uint32_t rl_me = 0, rc_me = 0;  // per point lane: entries below it, points below it
  for (uint64_t t = cm; t; t &= t - 1) {
    const int j = __ffsll((unsigned long long)t) - 1;
    const float dj = rlf(mydist, j);
    uint32_t below = 0;
#pragma unroll
    for (int r = 0; r < NREG; r++) {
      up[r] += dj < cdx[r] ? 1u : 0u;
      below += (uint32_t)__popcll(__ballot(cdx[r] < dj));
    }
    if (lane == j) rl_me = below;
    rc_me += dj < mydist ? 1u : 0u;
  }
#ifdef SDB_STAMPS
  asm volatile("" ::"v"(rc_me), "v"(up[0]));
#endif
  SDB_MST(2)
  // positions; with no two equal distances they are a permutation and exactly `cap` of them lie below `cap`
  const uint32_t np_me = rl_me + rc_me;
  uint32_t kept = (uint32_t)__popcll(__ballot(inC && np_me < (uint32_t)cap));
  uint32_t *s_id = scratch;
  float *s_d = reinterpret_cast<float *>(scratch + NREG * 64);
#pragma unroll
  for (int r = 0; r < NREG; r++) {
    const int e = r * 64 + lane;
    kept += (uint32_t)__popcll(__ballot(e < cap && (uint32_t)e + up[r] < (uint32_t)cap));
    if (e < cap) s_id[e] = kNoSlot;  // a position nobody lands on stays marked
  }
  if (kept != (uint32_t)cap) {
    return add_with_limit_lanes(cid, cd, len, cap, idreg, mydist, pd, lane);
  }
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < NREG; r++) {
    const int e = r * 64 + lane;
    const uint32_t np = (uint32_t)e + up[r];
    if (e < cap && np < (uint32_t)cap) s_id[np] = cid[r], s_d[np] = cd[r];
  }
  if (inC && np_me < (uint32_t)cap) s_id[np_me] = idreg, s_d[np_me] = mydist;
  wave_lds_sync();
  uint32_t nid[NREG];
  float nd[NREG];
  bool hole = false;
#pragma unroll
  for (int r = 0; r < NREG; r++) {
    const int e = r * 64 + lane;
    nid[r] = e < cap ? s_id[e] : cid[r], nd[r] = e < cap ? s_d[e] : cd[r];
    hole |= e < cap && nid[r] == kNoSlot;
  }
  wave_lds_sync();
  // an unfilled position = two elements shared another one = equal distances: replay one by one
  if (__ballot(hole)) {
    return add_with_limit_lanes(cid, cd, len, cap, idreg, mydist, pd, lane);
  }
#pragma unroll
  for (int r = 0; r < NREG; r++) cid[r] = nid[r], cd[r] = nd[r];
#ifdef SDB_STAMPS
  asm volatile("" ::"v"(cd[0]));
#endif
  SDB_MST(3)
}

// roaring Contains on this query's ascending slot list: 64-ary search, all lanes probe at once
template <typename T>  // uint32_t slots or uint64_t ids
__device__ __forceinline__ bool filter_contains(const T *__restrict__ arr, uint32_t n, T target, int lane) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 64) {
    const uint32_t step = (hi - lo + 63) / 64;
    const uint32_t idx = lo + (uint32_t)lane * step;
    const T v = idx < hi ? arr[idx] : ~(T)0;
    const uint64_t m = __ballot(idx < hi && v <= target);
    if (!m) return false;  // target below the first pivot
    const uint32_t k = (uint32_t)__popcll(m);
    lo = lo + (k - 1) * step;
    hi = (lo + step < hi) ? lo + step : hi;
  }
  const T v = (lo + (uint32_t)lane < hi) ? arr[lo + lane] : ~(T)0;
  return __ballot(lo + (uint32_t)lane < hi && v == target) != 0ull;
}

// ---- visited-set policies (the visitedSet interface, distset.go:66-69) -----------------------------------

// VisitedBitSet (distset.go:89-116): one bit per slot in HBM, test-and-set with a returning atomicOr.
struct BitVisited {
  uint32_t *__restrict__ bits;
  __device__ __forceinline__ bool test_and_set(bool active, uint32_t slot, int lane) {
    if (!active) return false;
    const uint32_t bit = 1u << (slot & 31);
    return !(atomicOr(&bits[slot >> 5], bit) & bit);
  }
};

// VisitedMap (distset.go:71-87) as an exact open-addressing hash set in LDS: a query marks a few thousand
// ids, so the set fits next to the wave and CheckAndVisit costs an LDS atomic instead of an HBM round
// trip.  Keys inserted by one instruction are distinct (rows are deduplicated), so the CAS loop only
// resolves slot collisions.  When the table fills past `limit` the set SPILLS: the wave clears its HBM
// bitset, replays every stored key into it and carries on there -- the set stays exact and the walk is
// not repeated (the reference makes the same kind of switch by id range, distset.go:140-153).
constexpr uint32_t kHashCap = 8192;    // plain store: 32 KB per wave -> 4 waves per CU
constexpr uint32_t kHashLimit = 6000;  // keys a table may hold before it spills (any CAP: limit * CAP / 8192)
constexpr uint32_t kHashCapPQ = 7417;  // quantized store: a prime, 29 KB, leaves room for the 8 KB LUT (M = 8)
#ifndef SDB_HASH_PROBES
#define SDB_HASH_PROBES 4  // 6 and 8 measured the same (tools/kernel_ab.py: 0.947 / 0.947 / 0.946 ms plain, 0.315 / 0.312 / 0.317 ms quantized)
#endif
constexpr int kProbes = SDB_HASH_PROBES;  // probe positions a round of the visited-set test reads at once
// CAP is a power of two (mask) or a prime (conditional subtract): either way every probe stride in
// [1, CAP) reaches every slot.
template <uint32_t CAP>
struct HashVisited {
  static constexpr bool kPow2 = (CAP & (CAP - 1)) == 0;
  static constexpr uint32_t kWords = (CAP + 3) & ~3u;  // LDS words reserved (16-byte multiple)
  uint32_t *tab;
  uint32_t *bits;
  uint32_t words, count, limit;
  bool spilled;
  __device__ __forceinline__ void init_nosync(uint32_t *lds, uint32_t *bitset, uint32_t nwords, int lane, uint32_t lim) {
    tab = lds, bits = bitset, words = nwords;
    count = 0, spilled = false;
    limit = (uint32_t)(((uint64_t)lim * CAP) >> 13);
    uint4 *t4 = reinterpret_cast<uint4 *>(lds);
    for (uint32_t i = lane; i < kWords / 4; i += 64) t4[i] = make_uint4(kNoSlot, kNoSlot, kNoSlot, kNoSlot);
  }
  __device__ __forceinline__ void init(uint32_t *lds, uint32_t *bitset, uint32_t nwords, int lane, uint32_t lim) {
    init_nosync(lds, bitset, nwords, lane, lim);
    __syncthreads();
  }
  __device__ __forceinline__ void spill(int lane) {
    for (uint32_t i = lane; i < words; i += 64) bits[i] = 0u;  // ClearAll distset.go:101
    __threadfence();                                           // the clears land before the atomics below
    for (uint32_t i = lane; i < CAP; i += 64) {
      const uint32_t k = tab[i];
      if (k != kNoSlot) atomicOr(&bits[k >> 5], 1u << (k & 31));
    }
    __threadfence();
    spilled = true;
  }
  __device__ __forceinline__ bool test_and_set(bool active, uint32_t slot, int lane) {
    if (spilled) {
      if (!active) return false;
      const uint32_t bit = 1u << (slot & 31);
      return !(atomicOr(&bits[slot >> 5], bit) & bit);
    }
    bool isnew = false, done = !active;
    // double hashing: probe chains of different keys do not pile up the way linear probing clusters.
    // start = hash1 * CAP >> 32 in [0, CAP); stride in [1, CAP), odd for the power-of-two table
    uint32_t h = (uint32_t)(((uint64_t)(slot * 2654435761u) * CAP) >> 32);
    uint32_t step = 1u + (uint32_t)(((uint64_t)(slot * 0x9E3779B1u) * (CAP - 1)) >> 32);
    if (kPow2) step |= 1u;
    auto next = [&](uint32_t x) {
      x += step;
      if (kPow2) x &= CAP - 1;
      else if (x >= CAP) x -= CAP;
      return x;
    };
    // A round reads kProbes consecutive probe positions at once (one LDS round trip), takes the first that is
    // empty or already holds the key -- nothing is ever removed, so a key that is present sits before the
    // first empty position of its chain -- and only then spends the atomic: almost every lane finishes in one
    // round, where a compare-and-swap per probe made the whole wave wait for its longest chain.
    while (__ballot(!done)) {
      uint32_t hp[kProbes], kp[kProbes];
      hp[0] = h;
#pragma unroll
      for (int i = 1; i < kProbes; i++) hp[i] = next(hp[i - 1]);
#pragma unroll
      for (int i = 0; i < kProbes; i++) kp[i] = tab[hp[i]];
      bool any = false;
      uint32_t hs = hp[kProbes - 1], ks = kp[kProbes - 1];
#pragma unroll
      for (int i = kProbes - 1; i >= 0; i--) {
        const bool si = kp[i] == slot || kp[i] == kNoSlot;
        hs = si ? hp[i] : hs, ks = si ? kp[i] : ks;
        any |= si;
      }
      if (!done) {
        if (!any) {
          h = next(hp[kProbes - 1]);
        } else if (ks == slot) {
          done = true;
        } else {
          const uint32_t old = atomicCAS(&tab[hs], kNoSlot, slot);
          if (old == kNoSlot) isnew = true, done = true;
          else if (old == slot) done = true;
          else h = hs;  // another lane's key landed there in this round: carry on from it
        }
      }
    }
    count += (uint32_t)__popcll(__ballot(isnew));
    if (count > limit) spill(lane);  // at most 64 keys past the limit: the table never fills
    return isnew;
  }
  // test_and_set's probe-and-claim as ANOTHER wave of the workgroup runs it on the walker's table, ahead of the walk
  // (PlainWideDist's marker wave): no count, no spill; `cell` is where a new key went, so that the walker can take the
  // marks back should the walk not go there after all (unmark).  The walker stays away from the table meanwhile.
  static __device__ __forceinline__ bool claim(uint32_t *tab, bool active, uint32_t slot, uint32_t &cell) {
    bool isnew = false, done = !active;
    uint32_t h = (uint32_t)(((uint64_t)(slot * 2654435761u) * CAP) >> 32);
    uint32_t step = 1u + (uint32_t)(((uint64_t)(slot * 0x9E3779B1u) * (CAP - 1)) >> 32);
    if (kPow2) step |= 1u;
    auto next = [&](uint32_t x) {
      x += step;
      if (kPow2) x &= CAP - 1;
      else if (x >= CAP) x -= CAP;
      return x;
    };
    cell = 0;
    while (__ballot(!done)) {
      uint32_t hp[kProbes], kp[kProbes];
      hp[0] = h;
#pragma unroll
      for (int i = 1; i < kProbes; i++) hp[i] = next(hp[i - 1]);
#pragma unroll
      for (int i = 0; i < kProbes; i++) kp[i] = tab[hp[i]];
      bool any = false;
      uint32_t hs = hp[kProbes - 1], ks = kp[kProbes - 1];
#pragma unroll
      for (int i = kProbes - 1; i >= 0; i--) {
        const bool si = kp[i] == slot || kp[i] == kNoSlot;
        hs = si ? hp[i] : hs, ks = si ? kp[i] : ks;
        any |= si;
      }
      if (!done) {
        if (!any) {
          h = next(hp[kProbes - 1]);
        } else if (ks == slot) {
          done = true;
        } else {
          const uint32_t old = atomicCAS(&tab[hs], kNoSlot, slot);
          if (old == kNoSlot) isnew = true, done = true, cell = hs;
          else if (old == slot) done = true;
          else h = hs;
        }
      }
    }
    return isnew;
  }
  __device__ __forceinline__ bool markable() const { return !spilled; }
  // marks made by claim() on the walker's behalf: now part of the set ...
  __device__ __forceinline__ void credit(uint32_t n, int lane) {
    count += n;
    if (count > limit) spill(lane);  // at most 128 keys past the limit
  }
  // ... or taken back (nothing has been added since they were made: the table is as it was before them)
  __device__ __forceinline__ void unmark(bool mine, uint32_t cell) {
    if (mine) tab[cell] = kNoSlot;
    wave_lds_sync();
  }
};

// The same exact set in half the LDS, for stores of up to 2^24 rows: 16-bit cells.  h = slot * odd mod 2^24 is a
// bijection of the 24-bit universe; its top 12 bits name a bucket -- one 32-bit LDS word, two cells -- and its low
// 12 bits are the remainder a cell stores, next to the number (0..14) of the probe that placed the key: probe i of
// a key looks at bucket (b + i * (2 rem + 1)) mod 4096, double hashing over buckets with a stride the cell itself
// gives back.  Position and content of a cell therefore identify its key exactly -- nothing is a fingerprint --
// and spill() rebuilds every key.  The cells of a bucket fill low half first and nothing is ever removed, so a key
// that is present sits before the first bucket of its sequence that is not full.  A key that finds 15 full buckets
// (probability ~1e-7 at 60 % load), or a table past its limit, spills to the HBM bitset like HashVisited does.
// 8 192 cells = 16 KB: with the 8 KB LUT of M = 8 six walks fit a CU instead of four.
constexpr uint32_t inverse24(uint32_t a) {  // a * x = 1 mod 2^24, Newton steps double the correct bits
  uint32_t x = a;
  for (int i = 0; i < 5; i++) x *= 2u - a * x;
  return x & 0xFFFFFFu;
}
constexpr uint32_t kHash16Mul = 0x9E3779B1u & 0xFFFFFFu, kHash16Inv = inverse24(kHash16Mul);
static_assert(((kHash16Mul * kHash16Inv) & 0xFFFFFFu) == 1u, "kHash16Inv must invert kHash16Mul mod 2^24");
struct HashVisited16 {
  static constexpr uint32_t kBuckets = 4096, kCells = 2 * kBuckets;
  static constexpr uint32_t kWords = kBuckets;  // 32-bit LDS words
  static constexpr uint32_t kMaxProbe = 14;     // (probe 15, remainder 0xFFF) is the empty cell
  uint32_t *tab;
  uint32_t *bits;
  uint32_t words, count, limit, maxp;
  bool spilled;
  __device__ __forceinline__ void init_nosync(uint32_t *lds, uint32_t *bitset, uint32_t nwords, int lane, uint32_t lim, uint32_t probes = 0) {
    tab = lds, bits = bitset, words = nwords;
    count = 0, spilled = false;
    maxp = (probes >= 1 && probes <= kMaxProbe) ? probes - 1 : kMaxProbe;  // last probe number a key may use
    limit = (uint32_t)(((uint64_t)lim * kCells) >> 13);  // 6 000 of 8 192 cells by default
    uint4 *t4 = reinterpret_cast<uint4 *>(lds);
    for (uint32_t i = lane; i < kWords / 4; i += 64) t4[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
  }
  __device__ __forceinline__ void init(uint32_t *lds, uint32_t *bitset, uint32_t nwords, int lane, uint32_t lim, uint32_t probes = 0) {
    init_nosync(lds, bitset, nwords, lane, lim, probes);
    __syncthreads();
  }
  __device__ __forceinline__ void spill(int lane) {
    for (uint32_t i = lane; i < words; i += 64) bits[i] = 0u;  // ClearAll distset.go:101
    __threadfence();
    for (uint32_t c = lane; c < kCells; c += 64) {
      const uint32_t v = (tab[c >> 1] >> ((c & 1u) * 16u)) & 0xFFFFu;
      if (v != 0xFFFFu) {
        const uint32_t rem = v & 0xFFFu, probe = v >> 12;
        const uint32_t b = ((c >> 1) - probe * (2u * rem + 1u)) & (kBuckets - 1);
        const uint32_t k = (((b << 12) | rem) * kHash16Inv) & 0xFFFFFFu;
        atomicOr(&bits[k >> 5], 1u << (k & 31));
      }
    }
    __threadfence();
    spilled = true;
  }
  __device__ __forceinline__ bool bit_test_and_set(bool active, uint32_t slot) {
    if (!active) return false;
    const uint32_t bit = 1u << (slot & 31);
    return !(atomicOr(&bits[slot >> 5], bit) & bit);
  }
  __device__ __forceinline__ bool test_and_set(bool active, uint32_t slot, int lane) {
    if (spilled) return bit_test_and_set(active, slot);
    const uint32_t h = (slot * kHash16Mul) & 0xFFFFFFu;
    const uint32_t rem = h & 0xFFFu, step = 2u * rem + 1u;
    uint32_t bucket = h >> 12, probe = 0;
    bool isnew = false, done = !active, stuck = false;
    // a round reads kProbes buckets of the sequence at once and takes the first that holds the key or has room
    while (__ballot(!done)) {
      uint32_t bp[kProbes], wp[kProbes];
      bp[0] = bucket;
#pragma unroll
      for (int i = 1; i < kProbes; i++) bp[i] = (bp[i - 1] + step) & (kBuckets - 1);
#pragma unroll
      for (int i = 0; i < kProbes; i++) wp[i] = tab[bp[i]];
      int hit = -1;  // first bucket of this round that ends the search
      bool present = false;
#pragma unroll
      for (int i = kProbes - 1; i >= 0; i--) {
        const uint32_t target = rem | ((probe + i) << 12);
        const bool has = (wp[i] & 0xFFFFu) == target || (wp[i] >> 16) == target;
        const bool room = (wp[i] >> 16) == 0xFFFFu;
        if ((has || room) && probe + i <= maxp) hit = i, present = has;
      }
      if (!done) {
        if (hit < 0) {
          probe += kProbes;
          if (probe > maxp) done = stuck = true;
          else bucket = (bp[kProbes - 1] + step) & (kBuckets - 1);
        } else if (present) {
          done = true;
        } else {
          uint32_t old = wp[0], at = bp[0];
#pragma unroll
          for (int i = 1; i < kProbes; i++)
            if (hit == i) old = wp[i], at = bp[i];
          const uint32_t target = rem | ((probe + (uint32_t)hit) << 12);
          const uint32_t neu = (old & 0xFFFFu) == 0xFFFFu ? ((old & 0xFFFF0000u) | target) : ((old & 0xFFFFu) | (target << 16));
          if (atomicCAS(tab + at, old, neu) == old) isnew = true, done = true;
          else probe += (uint32_t)hit, bucket = at;  // another lane's key took a cell of that bucket: look at it again
        }
      }
    }
    count += (uint32_t)__popcll(__ballot(isnew));
    const uint64_t st = __ballot(stuck);
    if (st || count > limit) {
      spill(lane);
      if (stuck) isnew = bit_test_and_set(true, slot);  // after the replay: the set is the bitset now
    }
    return isnew;
  }
};

// the unfiltered search has no result set of its own
struct NoVisited {
  __device__ __forceinline__ bool test_and_set(bool, uint32_t, int) { return false; }
};
// The filtered search's resultSet has a visited set of its own (search.go:37: NewDistSet(k, ...)): it sees the seeds
// (<= searchSize ids) and the expanded nodes that pass the filter -- a few hundred ids at most, so a 1 024-entry table
// (4 KB) next to the search set's; past 750 ids it spills to its HBM bitset like the big one does.
constexpr uint32_t kHashCapResult = 1024;

// does the distance policy have the two-precision stage (PlainDist<..., SK = true>)?
template <class D, class = void>
struct sketch_policy { static constexpr bool value = false; };
template <class D>
struct sketch_policy<D, decltype((void)D::kSketch)> { static constexpr bool value = D::kSketch; };

// greedySearch for one query by one wavefront.  RVis: the visited set of the filtered search's result set.
template <class Dist, int NREG, bool FILT, class Visited, class RVis>
__device__ __forceinline__ void search_body(const SearchArgs &a, const uint32_t q, const int lane, Dist &dist,
                                            Visited &vis, RVis &rvis) {
  uint32_t cid[NREG];
  float cd[NREG];
#pragma unroll
  for (int r = 0; r < NREG; r++) cid[r] = kNoSlot, cd[r] = 0.0f;
  int len = 0;
  const int cap = (int)a.search_size;
  uint32_t n_dist = 0, n_hop = 0, n_edges = 0;
  uint32_t n_sk_out = 0;  // two-precision hop: neighbours discarded on their float16 distance
  __shared__ uint32_t s_scatter[2 * NREG * 64];  // add_with_limit_merge scratch

  // filtered search (search.go:33-51): resultSet = DistSet(cap k) with its own visited set
  uint32_t rid[FILT ? NREG : 1];
  float rd[FILT ? NREG : 1];
  int rlen = 0;
  const int rcap = (int)a.limit;
  const uint32_t *__restrict__ fsorted = nullptr;
  const uint64_t *__restrict__ fids = nullptr;
  uint32_t nfilt = 0;
  if constexpr (FILT) {
#pragma unroll
    for (int r = 0; r < NREG; r++) rid[r] = kNoSlot, rd[r] = 0.0f;
    fsorted = a.filt_slots + a.filt_off[q];
    nfilt = a.filt_cnt ? a.filt_cnt[q] : a.filt_off[q + 1] - a.filt_off[q];
    if (a.filt_ids) fids = a.filt_ids + a.filt_off[q], nfilt = a.filt_off[q + 1] - a.filt_off[q];
    // seeds: the first <= searchSize filter ids in ascending id order that exist (:41-48)
    const uint32_t s0 = a.seed_off[q], ns = a.seed_cnt ? a.seed_cnt[q] : a.seed_off[q + 1] - s0;
    for (uint32_t base = 0; base < ns; base += 64) {
      const uint32_t j = base + lane;
      const bool has = j < ns;
      const uint32_t slot = has ? a.seeds[s0 + j] : kNoSlot;
      // searchSet.Add (:49): CheckAndVisit, distance, plain append -- NOT sorted
      dist.prefetch(a, slot, has);
      const bool isnew = vis.test_and_set(has, slot, lane);
      const uint64_t pend = __ballot(isnew);
      n_dist += (uint32_t)__popcll(pend);
      const float mydist = dist.hop(a, slot, pend, lane);
      for (uint64_t t = pend; t; t &= t - 1) {
        const int j0 = __ffsll((unsigned long long)t) - 1;
        const uint32_t id = rl(slot, j0);
        const float d = rlf(mydist, j0);
#pragma unroll
        for (int r = 0; r < NREG; r++)
          if ((len >> 6) == r && lane == (len & 63)) cid[r] = id, cd[r] = d;
        len++;
      }
      // resultSet.AddWithLimit(filterPoints...) (:50): its own CheckAndVisit, distances evaluated again
      const bool rnew = rvis.test_and_set(has, slot, lane);
      const uint64_t rpend = __ballot(rnew);
      n_dist += (uint32_t)__popcll(rpend);
      add_with_limit_lanes(rid, rd, rlen, rcap, slot, mydist, rpend, lane);
    }
  }

  // ---- searchSet.AddWithLimit(startNode)  search.go:57-61
  {
    const uint32_t s = a.start_slot;
    const bool snew = vis.test_and_set(lane == 0, s, lane);
    if (__ballot(snew)) {
      const float d = dist.one(a, s, lane);
      n_dist++;
      if (!(len == cap && d > list_tail(cd, cap))) list_insert(cid, cd, len, cap, s, d, lane);
    }
  }

#ifdef SDB_STAMPS  // diagnostic build only: where does a hop spend its cycles (never in the shipped library)
  unsigned long long st_adj = 0, st_atom = 0, st_vec = 0, st_ins = 0, st_t0 = 0;
  unsigned long long st_m[4] = {0, 0, 0, 0};  // inside the merge: preamble, <2-candidates path, per-point pass, scatter
  unsigned long long st_w[4] = {0, 0, 0, 0};  // Dist::kSpeculate: pick + guess, the marker's verdict, the late guess, naming the row
#define SDB_STAMP(acc)                                              \
  {                                                                 \
    unsigned long long _t = __builtin_amdgcn_s_memtime();           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
    acc += _t - st_t0;                                              \
    st_t0 = _t;                                                     \
  }
#else
#define SDB_STAMP(acc)
#endif
  uint32_t spec_pid = kNoSlot, spec_nb = kNoSlot;  // Dist::kSpeculate: the row fetched ahead, and whose it is
  float spec_d = 0.0f;                             // ... and that candidate's distance
  bool have_marks = false;                         // ... and whether its neighbours have been through the visited set
  uint64_t mark_mask = 0;
#ifdef SDB_SPEC_STATS
  uint32_t n_spec_hit = 0;  // measurement builds: hops that found their row fetched ahead, reported in place of n_edges
#endif
  // ---- main loop search.go:65-98
  while (true) {
#ifdef SDB_STAMPS
    st_t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    int sel = -1;
#pragma unroll
    for (int r = 0; r < NREG; r++) {
      uint64_t m = __ballot((r * 64 + lane) < len && !(cid[r] & kVisBit));
      if (sel < 0 && m) sel = r * 64 + __ffsll((unsigned long long)m) - 1;
    }
    if (sel < 0) break;
    uint32_t pid = 0;
    float pdist = 0.0f;
#pragma unroll
    for (int r = 0; r < NREG; r++)
      if ((sel >> 6) == r) {
        pid = rl(cid[r], sel & 63);
        pdist = rlf(cd[r], sel & 63);
        if (lane == (sel & 63)) cid[r] |= kVisBit;  // :74
      }
    if (lane == 0) {  // visitedSet.AddAlreadyUnique :73
#ifndef SDB_STAMPS
      if (a.tr_visit && n_hop < a.visit_cap) a.tr_visit[(size_t)q * a.visit_cap + n_hop] = a.ids[pid];
#endif
      if (a.vis_slots && n_hop < a.vis_cap) {
        a.vis_slots[(size_t)q * a.vis_cap + n_hop] = pid;
        a.vis_dists[(size_t)q * a.vis_cap + n_hop] = pdist;
      }
    }
    n_hop++;
    uint64_t pid_id = 0;  // FILT by ids: asked for now, looked at after the hop
    if constexpr (FILT)
      if (fids) pid_id = a.ids[pid];

    // node.neighbours in edge order :77-91.  One pass per 64 edges: every node has one, except a start node
    // that carries an overflow list (index.h h_start_ext) -- its chunks follow in the same edge order, which
    // is all AddWithLimit(neighbours...) depends on.
    const uint32_t *__restrict__ rowp = a.adj + (size_t)pid * kAdjStride;
    uint32_t ext_left = pid == a.start_slot ? a.start_ext_n : 0u, ext_done = 0;
    // Dist::kSpeculate: is this hop's row the one named during the last hop (then its adjacency row is in spec_nb), and
    // who is first in line now -- this hop's own entry is marked already -- with its distance: the hop names either it
    // or a nearer new point once its distances are known (below)
    bool use_spec = false;
    const uint32_t *spec_rowp = nullptr;
    if constexpr (Dist::kSpeculate) {
      use_spec = pid == spec_pid;
      int sel2 = -1;
#pragma unroll
      for (int r = 0; r < NREG; r++) {
        const uint64_t m2 = __ballot((r * 64 + lane) < len && !(cid[r] & kVisBit));
        if (sel2 < 0 && m2) sel2 = r * 64 + __ffsll((unsigned long long)m2) - 1;
      }
      spec_pid = kNoSlot;
      spec_d = __int_as_float(0x7f800000);
#pragma unroll
      for (int r = 0; r < NREG; r++)
        if (sel2 >= 0 && (sel2 >> 6) == r) spec_pid = rl(cid[r], sel2 & 63) & ~kVisBit, spec_d = rlf(cd[r], sel2 & 63);
      if (spec_pid != kNoSlot) spec_rowp = a.adj + (size_t)spec_pid * kAdjStride;
      dist.speculation(use_spec, spec_rowp);
      SDB_STAMP(st_w[0])
      // the visited-set test of the row named last has been run ahead (PlainWideDist's marker): its verdict, or back out
      have_marks = false;
      if (dist.marked) {
        uint32_t cell;
        dist.take_marks(lane, mark_mask, cell);
        if (use_spec) have_marks = true, vis.credit((uint32_t)__popcll(mark_mask), lane);
        else vis.unmark((mark_mask >> lane) & 1ull, cell);
      }
      SDB_STAMP(st_w[1])
    }
    bool first_chunk = true;
    while (true) {
      dist.begin_row(a, rowp, lane);
      uint32_t nb;
      if (Dist::kSpeculate && first_chunk && use_spec) {
        nb = spec_nb;  // fetched during the hop before this one
#ifdef SDB_SPEC_STATS
        n_spec_hit++;
#endif
      } else {
        nb = rowp[lane];
      }
      const bool valid = nb != kNoSlot;
      n_edges += (uint32_t)__popcll(__ballot(valid));
#ifdef SDB_STAMPS
      if (!Dist::kSpeculate) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // charge the adjacency round trip to st_adj
#endif
      SDB_STAMP(st_adj)
      if constexpr (sketch_policy<Dist>::value) dist.sk_go = len == cap && a.sketch != nullptr;
      dist.prefetch(a, nb, valid);
      // CheckAndVisit distset.go:174 -- marks before any distance test
      bool isnew;
      if (Dist::kSpeculate && have_marks && first_chunk) isnew = (mark_mask >> lane) & 1ull;
      else isnew = vis.test_and_set(valid, nb, lane);
      uint64_t pend = __ballot(isnew);
      SDB_STAMP(st_atom)
      if (pend) {
        n_dist += (uint32_t)__popcll(pend);
        if constexpr (sketch_policy<Dist>::value) {
          // two-precision hop: with the array full, the neighbours whose float16 distance is provably above its last
          // distance -- as it is NOW: it only falls while this row's points are inserted -- are discarded here;
          // AddWithLimit would discard them one by one (distset.go:184) and nothing else ever reads their distance
          if (len == cap) {
            const float tail_d = list_tail(cd, cap);
            uint64_t out = 0;
            const uint64_t keep = dist.sketch_keep(a, nb, pend, lane, tail_d, out);
            if (a.sk_audit) {  // (rare path: counted at once, not carried in a register through the walk)
              const float dx = dist.hop(a, nb, pend, lane);
              const uint64_t bad = __ballot(((out >> lane) & 1ull) && !(dx > tail_d));
              if (bad && lane == 0 && a.sk_counters) atomicAdd(a.sk_counters + 1, (unsigned long long)__popcll(bad));
            }
            n_sk_out += (uint32_t)__popcll(out);
            pend = keep;
          }
        }
        const float mydist = pend ? dist.hop(a, nb, pend, lane) : 0.0f;  // lane j: distance of edge j
        if constexpr (Dist::kPointDistances)
          if (a.dcache && ((pend >> lane) & 1ull))
            a.dcache[((size_t)q << (32 - a.dcache_shift)) + ((nb * 2654435761u) >> a.dcache_shift)] =
                make_uint2(nb, __float_as_uint(mydist));
#ifdef SDB_STAMPS
        asm volatile("" ::"v"(mydist));
#endif
        SDB_STAMP(st_vec)
        if constexpr (Dist::kSpeculate) {
          // With this hop's distances the next pick is no guess any more: it is the nearest of the new points when that
          // one is nearer than the candidate first in line (and gets into the array), else that candidate.  Known
          // BEFORE the points are inserted -- so the row to fetch and work ahead on is named now, and the insert below
          // runs beside that work.  (A tie or a NaN can still make it wrong; the next hop checks pid == spec_pid.)
          if (first_chunk) {
            if (ext_left == 0) {
              uint64_t nearer = __ballot(((pend >> lane) & 1ull) && mydist < spec_d);
              if (nearer) {
                int bj = -1;
                float bd = spec_d;
                for (; nearer; nearer &= nearer - 1) {
                  const int j = __ffsll((unsigned long long)nearer) - 1;
                  const float dj = rlf(mydist, j);
                  if (dj < bd) bd = dj, bj = j;
                }
                if (bj >= 0 && (len < cap || !(bd > list_tail(cd, cap)))) {
                  spec_pid = rl(nb, bj);
                  spec_rowp = a.adj + (size_t)spec_pid * kAdjStride;
                }
              }
            }
            // (the marker only when no overflow chunk follows: the walker's own tests of those must have the table to themselves)
            // its row, for the next hop: asked for HERE and not when the hop began -- a load in flight is waited for at
            // every workgroup barrier (the fence of __syncthreads), and the hop's handshake would stand still for it
            if (spec_rowp) spec_nb = spec_rowp[lane];
            SDB_STAMP(st_w[2])
            dist.ahead(a, spec_rowp, lane, vis.markable() && ext_left == 0);
            SDB_STAMP(st_w[3])
          }
        }
        // AddWithLimit over the new neighbours, in edge order distset.go:184-198
        if (!sketch_policy<Dist>::value || pend) {  // (the two-precision hop may have discarded every new neighbour)
          if constexpr (FILT) add_with_limit_lanes(cid, cd, len, cap, nb, mydist, pend, lane);  // array may be unsorted
#ifndef SDB_SKETCH_FEW
#define SDB_SKETCH_FEW 2  // measurement builds (tools/sketch_ab.py): 7 measured 0.756 against 0.729 ms
#endif
#ifdef SDB_STAMPS
          else add_with_limit_merge<NREG, sketch_policy<Dist>::value ? SDB_SKETCH_FEW : 2>(cid, cd, len, cap, nb, mydist, pend, lane, s_scatter, st_m);
#else
          else add_with_limit_merge<NREG, sketch_policy<Dist>::value ? SDB_SKETCH_FEW : 2>(cid, cd, len, cap, nb, mydist, pend, lane, s_scatter);
#endif
        }
        SDB_STAMP(st_ins)
      } else {
        dist.skip(lane);
        if constexpr (Dist::kSpeculate)
          if (first_chunk) {  // no new point: the candidate first in line stays it
            if (spec_rowp) spec_nb = spec_rowp[lane];
            dist.ahead(a, spec_rowp, lane, vis.markable() && ext_left == 0);
          }
      }
      if constexpr (Dist::kSpeculate) first_chunk = false;
      if (__builtin_expect(ext_left == 0, 1)) break;
      rowp = a.start_ext + ext_done;
      ext_done += 64;
      ext_left = ext_left > 64 ? ext_left - 64 : 0;
    }
    if constexpr (FILT) {  // :93-95 resultSet.AddWithLimit(distElem.Point) when the node passes the filter
      if (fids ? filter_contains(fids, nfilt, pid_id, lane) : filter_contains(fsorted, nfilt, pid, lane)) {
        const bool rnew = rvis.test_and_set(lane == 0, pid, lane);  // CheckAndVisit of the result set
        if (__ballot(rnew)) {
          n_dist++;  // distFn is evaluated again by the reference; same inputs, same bits as pdist
          if (!(rlen == rcap && pdist > list_tail(rd, rcap))) list_insert(rid, rd, rlen, rcap, pid, pdist, lane);
        }
      }
    }
  }
#ifdef SDB_STAMPS
  if (lane == 0 && a.tr_visit && a.visit_cap >= 4) {
    a.tr_visit[(size_t)q * a.visit_cap + 0] = st_adj;
    a.tr_visit[(size_t)q * a.visit_cap + 1] = st_atom;
    a.tr_visit[(size_t)q * a.visit_cap + 2] = st_vec;
    a.tr_visit[(size_t)q * a.visit_cap + 3] = st_ins;
    if constexpr (!Dist::kHasStamps)
      if (a.visit_cap >= 8)
        for (int i = 0; i < 4; i++) a.tr_visit[(size_t)q * a.visit_cap + 4 + i] = st_m[i];
    if constexpr (Dist::kHasStamps)
      if (a.visit_cap >= 8) {
        a.tr_visit[(size_t)q * a.visit_cap + 4] = dist.st[0];
        a.tr_visit[(size_t)q * a.visit_cap + 5] = dist.st[1];
        a.tr_visit[(size_t)q * a.visit_cap + 6] = dist.st[2];
        if (a.visit_cap >= 12)
          for (int i = 0; i < 4; i++) a.tr_visit[(size_t)q * a.visit_cap + 8 + i] = st_m[i];
        if (a.visit_cap >= 28)
          for (int i = 0; i < 4; i++) a.tr_visit[(size_t)q * a.visit_cap + 24 + i] = st_w[i];
      }
  }
#endif

  // ---- IndexVamana.Search result copy vamana.go:293-307 (from resultSet when filtered, search.go:36)
  // (what only this epilogue reads -- outputs, trace, counters -- comes through cold_args(): loaded here, once, instead
  // of being carried, spilled into VGPR lanes, from the kernel's first instruction through the whole walk)
  const SearchArgs &e = cold_args(a);
  if (e.out_ids) {
    int base = 0;
    const int olen = FILT ? rlen : len;
    const int limit = (int)e.limit;
    uint64_t *const o_ids = e.out_ids + (size_t)q * limit;
    float *const o_d = e.out_dists + (size_t)q * limit;
    const uint64_t *const ids = e.ids;
#pragma unroll
    for (int r = 0; r < NREG; r++) {
      const uint32_t s = (FILT ? rid[FILT ? r : 0] : cid[r]) & ~kVisBit;
      const float dd = FILT ? rd[FILT ? r : 0] : cd[r];
      const bool ok = (r * 64 + lane) < olen && s != a.start_slot;  // :294-296
      const uint64_t m = __ballot(ok);
      const int rank = base + __popcll(m & ((1ull << lane) - 1));
      if (ok && rank < limit) {  // :297-299
        o_ids[rank] = ids[s];
        o_d[rank] = dd;
      }
      base += __popcll(m);
    }
    const int got = base < limit ? base : limit;
    if (lane == 0) e.out_counts[q] = (uint32_t)got;
    for (int i = got + lane; i < limit; i += 64)  // rows shorter than `limit` end in zeros, not in
      o_ids[i] = 0, o_d[i] = 0.0f;                // whatever was there
  }
  if (lane == 0) {
    if (e.tr_ndist) e.tr_ndist[q] = n_dist;
    if (e.tr_nhop) e.tr_nhop[q] = n_hop;
#ifdef SDB_SPEC_STATS
    if (e.tr_nedges) e.tr_nedges[q] = Dist::kSpeculate ? n_spec_hit : n_edges;
#else
    if (e.tr_nedges) e.tr_nedges[q] = n_edges;
#endif
    if (e.vis_count) e.vis_count[q] = n_hop;
    if constexpr (sketch_policy<Dist>::value)
      if (e.sk_counters && n_sk_out) atomicAdd(e.sk_counters, (unsigned long long)n_sk_out);
    if (e.totals) {  // one of 64 copies of the counters (index.h kStatCopies)
      unsigned long long *t = e.totals + (q & 63u) * 16u;
      atomicAdd(t, (unsigned long long)n_dist), atomicAdd(t + 1, (unsigned long long)n_edges);
    }
  }
}


// HASH: visited set in LDS, spilling to the HBM bitset when it fills (plain store, unfiltered); otherwise the
// HBM bitset from the start.
// HCAP != 0: capacity of the LDS hash visited set (it sits first in dynamic LDS, the distance policy's
// tile / LUT after it), kHash16: the 16-bit-cell set; HCAP == 0: HBM bitset from the start.
constexpr uint32_t kHash16 = 0xFFFFFFFFu;  // HCAP value that selects HashVisited16
template <class Dist, int NREG, bool FILT, uint32_t HCAP>
__global__ __launch_bounds__(64) void k_greedy_search(const SearchArgs a) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  Dist dist;
  uint32_t *bits = a.bitsets + (size_t)q * a.words_per_query;
  // dynamic LDS: [search set's hash table][filtered: result set's hash table][distance policy's tile / LUT / scratch]
  constexpr uint32_t kRWords = FILT ? HashVisited<kHashCapResult>::kWords : 0;
  if constexpr (HCAP == kHash16) {
    dist.init(a, q, lane, lds_f + HashVisited16::kWords + kRWords);
    HashVisited16 hv;
    hv.init(reinterpret_cast<uint32_t *>(lds_f), bits, a.words_per_query, lane, a.hash_limit, a.hash16_probes);
    if constexpr (FILT) {
      HashVisited<kHashCapResult> rv;
      rv.init(reinterpret_cast<uint32_t *>(lds_f) + HashVisited16::kWords, a.rbitsets + (size_t)q * a.words_per_query,
              a.words_per_query, lane, a.hash_limit);
      search_body<Dist, NREG, FILT>(a, q, lane, dist, hv, rv);
    } else {
      NoVisited rv;
      search_body<Dist, NREG, FILT>(a, q, lane, dist, hv, rv);
    }
  } else if constexpr (HCAP != 0) {
    dist.init(a, q, lane, lds_f + HashVisited<HCAP>::kWords + kRWords);
    HashVisited<HCAP> hv;
    hv.init(reinterpret_cast<uint32_t *>(lds_f), bits, a.words_per_query, lane, a.hash_limit);
    if constexpr (FILT) {
      HashVisited<kHashCapResult> rv;
      rv.init(reinterpret_cast<uint32_t *>(lds_f) + HashVisited<HCAP>::kWords, a.rbitsets + (size_t)q * a.words_per_query,
              a.words_per_query, lane, a.hash_limit);
      search_body<Dist, NREG, FILT>(a, q, lane, dist, hv, rv);
    } else {
      NoVisited rv;
      search_body<Dist, NREG, FILT>(a, q, lane, dist, hv, rv);
    }
  } else {
    dist.init(a, q, lane, lds_f);
    BitVisited bv{bits};
    if constexpr (FILT) {
      BitVisited rv{a.rbitsets + (size_t)q * a.words_per_query};
      search_body<Dist, NREG, FILT>(a, q, lane, dist, bv, rv);
    } else {
      NoVisited rv;
      search_body<Dist, NREG, FILT>(a, q, lane, dist, bv, rv);
    }
  }
}

// ---- the one-wave quantized walk on two waves -------------------------------------------------------------
// A quantizer whose table sits in LDS beside the walk (M x K <= 2 048) makes a hop a matter of instructions, not of
// bytes: at 10M x 768, M = 8 a hop is ~7 200 cycles of ONE wave's instruction stream -- 1 550 the row and its code
// rows, 2 100 the visited-set test, 500 the table sums, 3 050 AddWithLimit (profiles/r05_stamps_c4.txt) -- and a batch
// of 1 024 queries is four lone waves per CU.  Here a query is two waves, and the stream is cut where its only
// dependency allows:
//   * the WALKER (wave 0) fetches the row, runs the visited-set test and the sums -- and names the next hop's node
//     itself, from one fact about the candidate array (its first unvisited entry F1 after the last insertions) and
//     this hop's distances: a new point lands in front of F1 iff its distance is smaller (distset.go:196-198 moves a
//     point left while it is `<` its neighbour, so it stops behind every entry `<=` it), the smallest such point --
//     if it is the only one at that distance -- is then the array's first unvisited entry (it is always inserted: at
//     its turn the array's tail is F1 or a point not smaller than it; nothing inserted later is in front of it or
//     overwrites it), and if there is none F1 stays where it is;
//   * the MERGER (wave 1) owns the candidate array: it takes the hop's (slot, distance) points from LDS, runs
//     AddWithLimit exactly as the one-wave kernel does, marks the named node, and answers with the next F1 -- while the
//     walker is already at the named node's row.
// (The merger answering AHEAD of its insertions -- the same rule one step on, from the array's first two unvisited
// entries and the batch's two smallest points; 79 of a walk's 83 answers, none wrong -- was built and measured: 0.308
// against 0.306 ms per batch.  The walker's own stream, ~3 700 cycles of fetch, visited-set test and sums plus ~800 of
// naming and hand-over, is what a hop takes; the merger is idle a third of the time either way.  Removed.)
// Nothing is guessed: whenever the rule above does not apply -- no unvisited entry left, a point equal to F1, two points
// sharing the smallest distance, a NaN among
// the distances now or earlier (a NaN entry stops every later point behind it, distset.go:197) -- the walker names nothing, waits for the
// merge and is told the next node (or that the walk is over) by the merger.
// a 64-bit LDS word read / written as ONE ds instruction (a volatile access through a generic pointer compiles to
// flat_load / flat_store with system scope: the aperture check and the vector-memory path cost a mailbox poll ~600 cycles)
typedef __attribute__((address_space(3))) volatile unsigned long long lds_u64_t;
__device__ __forceinline__ unsigned long long lds_load_u64(const void *p) {
  return *(lds_u64_t *)p;
}
__device__ __forceinline__ void lds_store_u64(void *p, unsigned long long v) { *(lds_u64_t *)p = v; }
// the mailboxes' payload goes through volatile LDS accesses as well: the compiler keeps volatile accesses in program
// order among themselves, so the payload's stores stay in front of the store of the word that carries the sequence
// number and its loads stay behind the load that saw it -- the hardware performs one wave's LDS instructions in issue
// order, and the emitted instructions are the same ds_read / ds_write as before
typedef __attribute__((address_space(3))) volatile uint32_t lds_u32_t;
__device__ __forceinline__ uint32_t lds_load_u32(const void *p) { return *(lds_u32_t *)p; }
__device__ __forceinline__ void lds_store_u32(void *p, uint32_t v) { *(lds_u32_t *)p = v; }

struct Pq2Shared {
  // Both mailboxes are written by one lane with volatile LDS stores, the word that carries the sequence number last: a
  // wave's LDS instructions are performed in the order it issued them (and volatile accesses are issued in program
  // order), so whoever sees the number sees what was written before it -- no s_waitcnt on either side (a workgroup-scope
  // release would wait for every global load the walker has in flight: the next hop's rows).
  uint2 pts_word;  // x: the node the walker goes to next, or kNoSlot: the merger says; y: batches posted (monotonic)
  uint2 pts_mask;  // lanes of the batch that hold a point
  uint2 ans_word;  // x: pts_word.x == kNoSlot: the first unvisited entry after the insertions (marked now), kNoSlot: none left; y: batches merged
  uint2 ans_f1;    // the first unvisited entry behind the node the walk goes to (x: slot or kNoSlot, y: its distance's bits)
  uint32_t pad[8];
  uint32_t pts_id[64];
  float pts_d[64];
};
constexpr uint32_t kPq2SharedWords = sizeof(Pq2Shared) / 4;

template <class Visited>
__device__ __forceinline__ void pq2_walker(const SearchArgs &a, const uint32_t q, const int lane, PQDist &dist, Visited &vis,
                                           Pq2Shared *sh) {
  uint32_t n_dist = 0, n_hop = 0, n_edges = 0, seq = 0;
  bool seen_nan = false;
#ifdef SDB_PQ2_STATS
  uint32_t n_slow = 0;
  unsigned long long w_t0 = __builtin_amdgcn_s_memtime(), w_tot0 = w_t0, w_acc[4] = {0, 0, 0, 0};  // front part, wait, naming + post, told
#define SDB_PQ2_W(i)                                          \
  {                                                           \
    unsigned long long _t = __builtin_amdgcn_s_memtime();     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    w_acc[i] += _t - w_t0;                                    \
    w_t0 = _t;                                                \
  }
#else
#define SDB_PQ2_W(i)
#endif
  // one batch of points to the merger; `named`: where the walk goes next, if the walker knows
  auto post = [&](uint64_t pend, uint32_t id, float d, uint32_t named) {
    if ((pend >> lane) & 1ull) lds_store_u32(&sh->pts_id[lane], id), lds_store_u32(&sh->pts_d[lane], __float_as_uint(d));
    seq++;
    if (lane == 0) lds_store_u64(&sh->pts_mask, (unsigned long long)pend);
    wave_lds_sync();
    if (lane == 0) lds_store_u64(&sh->pts_word, (unsigned long long)named | ((unsigned long long)seq << 32));
#ifdef SDB_PQ2_STATS  // hand-over timeline of query 0, sequence numbers 17 .. 24 (tools/pq2_stats.py, visit_cap >= 40)
    if (q == 0 && lane == 0 && seq >= 17 && seq < 25 && a.tr_visit && a.visit_cap >= 40)
      a.tr_visit[8 + (seq - 17) * 4 + 0] = __builtin_amdgcn_s_memtime();
#endif
  };
  uint32_t told = kNoSlot;  // the merger's word on where to go (read by answered())
  auto answered = [&]() {   // the merger is through with everything posted
    unsigned long long w;
    while (w = lds_load_u64(&sh->ans_word), (uint32_t)(w >> 32) != seq) __builtin_amdgcn_s_sleep(1);
    told = (uint32_t)w;
    wave_lds_sync();
#ifdef SDB_PQ2_STATS
    if (q == 0 && lane == 0 && seq >= 17 && seq < 25 && a.tr_visit && a.visit_cap >= 40)
      a.tr_visit[8 + (seq - 17) * 4 + 1] = __builtin_amdgcn_s_memtime();
#endif
  };
  // ---- searchSet.AddWithLimit(startNode)  search.go:57-61: a batch of one point, the merger names the first node
  {
    const uint32_t s = a.start_slot;
    const bool snew = vis.test_and_set(lane == 0, s, lane);
    const uint64_t pend = __ballot(snew);
    float d = 0.0f;
    if (pend) d = dist.one(a, s, lane), n_dist++;
    seen_nan = d != d;
    post(pend, s, d, kNoSlot);
  }
  answered();
  uint32_t pid = told;
  // ---- main loop search.go:65-98
  while (pid != kNoSlot) {
#ifndef SDB_PQ2_STATS
    if (lane == 0 && a.tr_visit && n_hop < a.visit_cap) a.tr_visit[(size_t)q * a.visit_cap + n_hop] = a.ids[pid];  // :73
#endif
    n_hop++;
    const uint32_t *__restrict__ rowp = a.adj + (size_t)pid * kAdjStride;
    dist.begin_row(a, rowp, lane);
    const uint32_t nb = rowp[lane];  // node.neighbours in edge order :77-91
    const bool valid = nb != kNoSlot;
    n_edges += (uint32_t)__popcll(__ballot(valid));
    dist.prefetch(a, nb, valid);
    const bool ahead8 = dist.ahead8(a);
    float t8[8];
    if (ahead8) dist.load8(a, valid, t8);
    const bool isnew = vis.test_and_set(valid, nb, lane);  // CheckAndVisit distset.go:174
    const uint64_t pend = __ballot(isnew);
    float mydist = 0.0f;
    if (pend) {
      n_dist += (uint32_t)__popcll(pend);
      mydist = ahead8 ? dist.sum8(t8, pend, lane) : dist.hop(a, nb, pend, lane);
    } else {
      dist.skip(lane);
    }
    const bool mine = (pend >> lane) & 1ull;
    seen_nan = seen_nan || (__ballot(mine && mydist != mydist) != 0ull);
#ifdef SDB_PQ2_STATS
    asm volatile("" ::"v"(mydist));
#endif
    SDB_PQ2_W(0)
    // the array after the LAST hop's insertions (the merger has had this hop's fetch, test and sums for them)
    answered();
    SDB_PQ2_W(1)
    const unsigned long long f1w64 = lds_load_u64(&sh->ans_f1);
    const uint2 f1w = make_uint2((uint32_t)f1w64, (uint32_t)(f1w64 >> 32));
    const uint32_t f1 = f1w.x;
    const float f1d = __uint_as_float(f1w.y);
    uint32_t named = kNoSlot;
    // (a point EQUAL to F1 goes behind it -- unless F1 is the full array's tail, which the point overwrites before it
    // moves, distset.go:189-194: rare enough to leave to the merger as well)
    if (!seen_nan && f1 != kNoSlot && !__ballot(mine && mydist == f1d)) {
      uint32_t b1 = f1;
      float b1d = f1d, b2d = f1d;  // the two smallest distances in front of F1, edge order among equals
      for (uint64_t t = __ballot(mine && mydist < f1d); t; t &= t - 1) {
        const int j = __ffsll((unsigned long long)t) - 1;
        const float dj = rlf(mydist, j);
        if (dj < b1d) b2d = b1d, b1d = dj, b1 = rl(nb, j);
        else if (dj < b2d) b2d = dj;
      }
      // two points share the smallest distance: the first in edge order is in front -- unless it is the full array's
      // tail when the second arrives and is overwritten by it (distset.go:189-194); left to the merger
      if (b1 == f1 || b2d != b1d) named = b1;
    }
    post(pend, nb, mydist, named);
    SDB_PQ2_W(2)
    if (named == kNoSlot) {
#ifdef SDB_PQ2_STATS  // measurement builds: hops the merger named, reported in place of n_edges
      n_slow++;
#endif
      answered();
      named = told;
      SDB_PQ2_W(3)
    }
    pid = named;
  }
  if (lane == 0) {
    if (a.tr_ndist) a.tr_ndist[q] = n_dist;
    if (a.tr_nhop) a.tr_nhop[q] = n_hop;
#ifdef SDB_PQ2_STATS
    if (a.tr_nedges) a.tr_nedges[q] = n_slow;
    if (a.tr_visit && a.visit_cap >= 8) {
      for (int i = 0; i < 4; i++) a.tr_visit[(size_t)q * a.visit_cap + i] = w_acc[i];
      a.tr_visit[(size_t)q * a.visit_cap + 4] = __builtin_amdgcn_s_memtime() - w_tot0;
    }
#else
    if (a.tr_nedges) a.tr_nedges[q] = n_edges;
#endif
    if (a.vis_count) a.vis_count[q] = n_hop;
    if (a.totals) {  // one of 64 copies of the counters (index.h kStatCopies)
      unsigned long long *t = a.totals + (q & 63u) * 16u;
      atomicAdd(t, (unsigned long long)n_dist), atomicAdd(t + 1, (unsigned long long)n_edges);
    }
  }
}

__device__ __forceinline__ void pq2_merger(const SearchArgs &a, const uint32_t q, const int lane, Pq2Shared *sh, uint32_t *scratch) {
  constexpr int NREG = 2;
  uint32_t cid[NREG];
  float cd[NREG];
#pragma unroll
  for (int r = 0; r < NREG; r++) cid[r] = kNoSlot, cd[r] = 0.0f;
  int len = 0;
  const int cap = (int)a.search_size;
  uint32_t pulling = 0, pulled = 0;  // words of the rows pulled ahead (see below)
#ifdef SDB_PQ2_STATS
  unsigned long long m_t0 = __builtin_amdgcn_s_memtime(), m_acc[3] = {0, 0, 0};  // waiting for points, AddWithLimit, mark + answer
#define SDB_PQ2_M(i)                                          \
  {                                                           \
    unsigned long long _t = __builtin_amdgcn_s_memtime();     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    m_acc[i] += _t - m_t0;                                    \
    m_t0 = _t;                                                \
  }
#else
#define SDB_PQ2_M(i)
#endif
  for (uint32_t seq = 1;; seq++) {
    unsigned long long w;
    while (w = lds_load_u64(&sh->pts_word), (uint32_t)(w >> 32) != seq) __builtin_amdgcn_s_sleep(1);
    wave_lds_sync();
#ifdef SDB_PQ2_STATS
    if (q == 0 && lane == 0 && seq >= 17 && seq < 25 && a.tr_visit && a.visit_cap >= 40)
      a.tr_visit[8 + (seq - 17) * 4 + 2] = __builtin_amdgcn_s_memtime();
#endif
    SDB_PQ2_M(0)
    const uint32_t named = (uint32_t)w;
    const unsigned long long pm64 = lds_load_u64(&sh->pts_mask);
    const uint2 pm = make_uint2((uint32_t)pm64, (uint32_t)(pm64 >> 32));
    const uint64_t pend = (uint64_t)pm.x | ((uint64_t)pm.y << 32);
    const bool mine = (pend >> lane) & 1ull;
    const uint32_t id = mine ? lds_load_u32(&sh->pts_id[lane]) : kNoSlot;
    const float d = mine ? __uint_as_float(lds_load_u32(&sh->pts_d[lane])) : 0.0f;
    // AddWithLimit over the new neighbours, in edge order distset.go:184-198
    if (pend) add_with_limit_merge(cid, cd, len, cap, id, d, pend, lane, scratch);
#ifdef SDB_PQ2_STATS
    asm volatile("" ::"v"(cd[0]));
#endif
    SDB_PQ2_M(1)
    // the node the walk goes to -- named by the walker, or the first unvisited entry (search.go:66-71) -- is marked :74
    uint64_t um[NREG];  // the array's unvisited entries, by position
#pragma unroll
    for (int r = 0; r < NREG; r++) um[r] = __ballot((r * 64 + lane) < len && !(cid[r] & kVisBit));
    int sel = -1;
    if (named != kNoSlot) {
#pragma unroll
      for (int r = 0; r < NREG; r++) {
        const uint64_t m = __ballot((r * 64 + lane) < len && cid[r] == named);
        if (m) sel = r * 64 + __ffsll((unsigned long long)m) - 1;
      }
    } else {
#pragma unroll
      for (int r = NREG - 1; r >= 0; r--)
        if (um[r]) sel = r * 64 + __ffsll((unsigned long long)um[r]) - 1;
    }
    uint32_t next = kNoSlot;
#pragma unroll
    for (int r = 0; r < NREG; r++)
      if (sel >= 0 && (sel >> 6) == r) {
        next = rl(cid[r], sel & 63);
        if (lane == (sel & 63)) cid[r] |= kVisBit;
        um[r] &= ~(1ull << (sel & 63));
      }
    // the first unvisited entry behind it: what the walker names the hop after this one with
    uint32_t f1 = kNoSlot;
    float f1d = 0.0f;
#pragma unroll
    for (int r = NREG - 1; r >= 0; r--)
      if (um[r]) {
        const int s2 = __ffsll((unsigned long long)um[r]) - 1;
        f1 = rl(cid[r], s2), f1d = rlf(cd[r], s2);
      }
    if (lane == 0) lds_store_u64(&sh->ans_f1, (unsigned long long)f1 | ((unsigned long long)__float_as_uint(f1d) << 32));
    wave_lds_sync();
    if (lane == 0) lds_store_u64(&sh->ans_word, (unsigned long long)next | ((unsigned long long)seq << 32));
#ifdef SDB_PQ2_STATS
    if (q == 0 && lane == 0 && seq >= 17 && seq < 25 && a.tr_visit && a.visit_cap >= 40)
      a.tr_visit[8 + (seq - 17) * 4 + 3] = __builtin_amdgcn_s_memtime();
#endif
    // That entry is where the walk goes next unless the hop under way finds a nearer point: its adjacency row and the
    // code rows behind it are pulled through L2 now, a whole hop before the walker asks for them (the values are not
    // used; they are waited for one batch later, when they have long arrived): 0.321 -> 0.306 ms per batch at 4M x 768
    pulled ^= pulling;
    pulling = 0;
    if (f1 != kNoSlot && a.adj_codes) {
      pulling = a.adj[(size_t)f1 * kAdjStride + lane];
      const size_t row_bytes = (size_t)64 * a.pq_M;  // one 64-byte line per lane
      if ((uint32_t)lane * 64u < row_bytes) pulling ^= *reinterpret_cast<const uint32_t *>(a.adj_codes + (size_t)f1 * row_bytes + (size_t)lane * 64);
    }
    SDB_PQ2_M(2)
    if (named == kNoSlot && next == kNoSlot) break;  // no unvisited entry left :66-71 -- the walker has been waiting for this
  }
#ifdef SDB_PQ2_STATS
  if (lane == 0 && a.tr_visit && a.visit_cap >= 8)
    for (int i = 0; i < 3; i++) a.tr_visit[(size_t)q * a.visit_cap + 5 + i] = m_acc[i];
#endif
  if ((pulled ^ pulling) == 0x9e3779b9u && a.limit == 0xFFFFFFFFu) a.out_counts[q] = pulled;  // (keeps the pulls alive; never true)
  // ---- IndexVamana.Search result copy vamana.go:293-307
  if (a.out_ids) {
    int base = 0;
#pragma unroll
    for (int r = 0; r < NREG; r++) {
      const uint32_t s = cid[r] & ~kVisBit;
      const bool ok = (r * 64 + lane) < len && s != a.start_slot;  // :294-296
      const uint64_t m = __ballot(ok);
      const int rank = base + __popcll(m & ((1ull << lane) - 1));
      if (ok && rank < (int)a.limit) {  // :297-299
        a.out_ids[(size_t)q * a.limit + rank] = a.ids[s];
        a.out_dists[(size_t)q * a.limit + rank] = cd[r];
      }
      base += __popcll(m);
    }
    const int got = base < (int)a.limit ? base : (int)a.limit;
    if (lane == 0) a.out_counts[q] = (uint32_t)got;
    for (int i = got + lane; i < (int)a.limit; i += 64)
      a.out_ids[(size_t)q * a.limit + i] = 0, a.out_dists[(size_t)q * a.limit + i] = 0.0f;
  }
}

// Dynamic LDS: [visited set's table][the query's M x K table][Pq2Shared].  Unfiltered searches with searchSize <= 128,
// no visit log for a build, no overflow list on the start node (index.hip launch_greedy_search decides).
// (A workgroup of four waves with the walker on wave (blockIdx / 256) % 4 and the merger two SIMDs on -- the placement
// trick of k_greedy_search_pqw -- was slower: 0.370 against 0.323 ms at 4M x 768.)
template <uint32_t HCAP>
__global__ __launch_bounds__(128) void k_greedy_search_pq2(const SearchArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t q = blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  __shared__ uint32_t s_scatter2[2 * 2 * 64];  // add_with_limit_merge scratch
  constexpr uint32_t kVisWords = HCAP == kHash16 ? HashVisited16::kWords : HashVisited<HCAP == kHash16 ? 4u : HCAP>::kWords;
  float *lut = lds_f + kVisWords;
  Pq2Shared *sh = reinterpret_cast<Pq2Shared *>(lut + ((a.pq_M * a.pq_K + 3u) & ~3u));
  uint32_t *bits = a.bitsets + (size_t)q * a.words_per_query;
  if (wave == 1) {
    const float *g = a.pq_lut + (size_t)q * a.pq_M * a.pq_K;
    for (uint32_t i = lane; i < a.pq_M * a.pq_K; i += 64) lut[i] = g[i];
    if (lane == 0) sh->pts_word = make_uint2(kNoSlot, 0u), sh->ans_word = make_uint2(kNoSlot, 0u);
    __syncthreads();  // table, visited set and mailbox in place
    return pq2_merger(a, q, lane, sh, s_scatter2);
  }
  PQDist dist;
  dist.lut = lut;
  if constexpr (HCAP == kHash16) {
    HashVisited16 hv;
    hv.init_nosync(reinterpret_cast<uint32_t *>(lds_f), bits, a.words_per_query, lane, a.hash_limit, a.hash16_probes);
    __syncthreads();
    pq2_walker(a, q, lane, dist, hv, sh);
  } else {
    HashVisited<HCAP == kHash16 ? 4u : HCAP> hv;
    hv.init_nosync(reinterpret_cast<uint32_t *>(lds_f), bits, a.words_per_query, lane, a.hash_limit);
    __syncthreads();
    pq2_walker(a, q, lane, dist, hv, sh);
  }
}

// The walker of the multi-wave quantized walk in its split form: k_greedy_search_pq2's division of labour inside
// k_greedy_search_pqw.  The candidate array lives in a helper wave (PQWideDist::serve<true>, the merger), which inserts
// a hop's points one round late, under the next row's code fetch and visited-set test; this wave names the next node
// from the array's first unvisited entry and the hop's sums, by the rule and with the exceptions described above
// pq2_walker -- when the rule does not apply it asks (a round of two barriers) and the merger picks.
template <class Dist, class Visited>
__device__ __forceinline__ void pqw_split_walker(const SearchArgs &a, const uint32_t q, const int lane, Dist &dist, Visited &vis) {
  uint32_t n_dist = 0, n_hop = 0, n_edges = 0;
  bool seen_nan = false;
  // ---- searchSet.AddWithLimit(startNode)  search.go:57-61
  {
    const uint32_t s = a.start_slot;
    const bool snew = vis.test_and_set(lane == 0, s, lane);
    if (__ballot(snew)) {
      const float d = dist.one(a, s, lane);
      n_dist++;
      seen_nan = d != d;
    }
  }
  uint32_t pid = dist.ask_next(lane);
  dist.post_named = kNoSlot;
  // ---- main loop search.go:65-98
  while (pid != kNoSlot) {
    if (lane == 0 && a.tr_visit && n_hop < a.visit_cap) a.tr_visit[(size_t)q * a.visit_cap + n_hop] = a.ids[pid];  // :73
    n_hop++;
    const uint32_t *__restrict__ rowp = a.adj + (size_t)pid * kAdjStride;
    dist.begin_row(a, rowp, lane);
    const uint32_t nb = rowp[lane];  // node.neighbours in edge order :77-91
    const bool valid = nb != kNoSlot;
    n_edges += (uint32_t)__popcll(__ballot(valid));
    dist.prefetch(a, nb, valid);
    const bool isnew = vis.test_and_set(valid, nb, lane);  // CheckAndVisit distset.go:174
    const uint64_t pend = __ballot(isnew);
    float mydist = 0.0f;
    if (pend) {
      n_dist += (uint32_t)__popcll(pend);
      mydist = dist.hop(a, nb, pend, lane);
    } else {
      dist.skip(lane);
    }
    const bool mine = (pend >> lane) & 1ull;
    seen_nan = seen_nan || (__ballot(mine && mydist != mydist) != 0ull);
    // the array after the LAST hop's insertions and with this hop's node marked: written by the merger before this
    // round's second barrier
    const uint32_t f1 = dist.sh->f1;
    const float f1d = dist.sh->f1d;
    uint32_t named = kNoSlot;
    if (!seen_nan && f1 != kNoSlot && !__ballot(mine && mydist == f1d)) {
      uint32_t b1 = f1;
      float b1d = f1d, b2d = f1d;  // the two smallest distances in front of F1, edge order among equals
      for (uint64_t t = __ballot(mine && mydist < f1d); t; t &= t - 1) {
        const int j = __ffsll((unsigned long long)t) - 1;
        const float dj = rlf(mydist, j);
        if (dj < b1d) b2d = b1d, b1d = dj, b1 = rl(nb, j);
        else if (dj < b2d) b2d = dj;
      }
      if (b1 == f1 || b2d != b1d) named = b1;
    }
    dist.post_named = named;  // (with the next row; kNoSlot: the merger has settled by then)
    pid = named != kNoSlot ? named : dist.ask_next(lane);
  }
  if (lane == 0) {
    if (a.tr_ndist) a.tr_ndist[q] = n_dist;
    if (a.tr_nhop) a.tr_nhop[q] = n_hop;
    if (a.tr_nedges) a.tr_nedges[q] = n_edges;
    if (a.vis_count) a.vis_count[q] = n_hop;
    if (a.totals) {  // one of 64 copies of the counters (index.h kStatCopies)
      unsigned long long *t = a.totals + (q & 63u) * 16u;
      atomicAdd(t, (unsigned long long)n_dist), atomicAdd(t + 1, (unsigned long long)n_edges);
    }
  }
}

// The multi-wave quantized walk: one query per workgroup of four waves (PQWideDist above).  Dynamic LDS: the visited
// set's table, the command area, the LDS-resident tables [4][NL][K].
// NL < 16: the variant meant to run two queries per CU (half the LDS each) -- its registers are capped accordingly
// NLW: tables the WALKER keeps in LDS (the rest of its NL + RT in registers).  The walker carries search_body's state
// (candidate array, visited set, ~100 registers) on top of what every wave holds; at eight waves per query (M = 384,
// 256 registers per wave) that state plus 33 register tables plus the 48 looked-up values did not fit -- 43 registers
// spilled.  There the walker takes 24 of its 48 tables from LDS and 24 from registers; the helpers keep 15 + 33.
// SPLIT: the candidate array lives in the helper next to the walker (pqw_split_walker, PQWideDist::serve<true>)
template <int NL, int RT, uint32_t HCAP, int W = 4, int NLW = NL, bool SPLIT = false>
__global__ __launch_bounds__(64 * W, (NL < 16 && W == 4) ? 2 : 1) void k_greedy_search_pqw(const SearchArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t q = blockIdx.x;
  // Which wave walks.  Every wave owns its index range whatever its role; the walker additionally runs search_body
  // (visited set, candidate array), about twice a helper's instruction count, on ITS SIMD.  Two queries share a CU in
  // the NL < 16 variant, and the k-th wave of a workgroup lands on the k-th SIMD: with wave 0 walking in both, one
  // SIMD carried both walkers.  Workgroups that share a CU are 256 (or a multiple) apart in the grid, so the role
  // rotates with blockIdx / 256.  Any choice is correct; this one balances.
  const int walker = (NL < 16 && W == 4) ? (int)((blockIdx.x >> 8) & 3u) : 0;
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  constexpr uint32_t kVisWords = HCAP == kHash16 ? HashVisited16::kWords : HashVisited<HCAP == kHash16 ? 4u : HCAP>::kWords;
  PQWideShared *sh = reinterpret_cast<PQWideShared *>(lds_f + kVisWords);
  float *lut_lds = lds_f + kVisWords + kPqwSharedWords;
  static_assert(NLW == NL || W == 8, "a walker with its own split is wave 0 of the eight-wave form");
  using Walker = PQWideDist<NLW, NL + RT - NLW, W>;
  using Helper = PQWideDist<NL, RT, W>;
  uint32_t *bits = a.bitsets + (size_t)q * a.words_per_query;
  __shared__ uint32_t s_scatter_m[SPLIT ? 2 * 2 * 64 : 1];  // the merger's add_with_limit_merge scratch
  if (wave != walker) {
    Helper dist;
    // (NLW != NL: the walker is wave 0 and its NLW tables come first in the block)
    dist.init_wave(a, q, lane, wave, lut_lds, sh, NLW == NL ? (uint32_t)wave * NL : (uint32_t)(NLW + (wave - 1) * NL));
    __syncthreads();  // tables and visited set in place
    if constexpr (SPLIT)
      if (wave == ((walker + 1) & (W - 1))) return dist.template serve<true>(a, lane, q, s_scatter_m);
    return dist.serve(a, lane);
  }
  Walker dist;
  dist.init_wave(a, q, lane, wave, lut_lds, sh, NLW == NL ? (uint32_t)wave * NL : 0u);
  NoVisited rv;
  if constexpr (HCAP == kHash16) {
    HashVisited16 hv;
    hv.init_nosync(reinterpret_cast<uint32_t *>(lds_f), bits, a.words_per_query, lane, a.hash_limit, a.hash16_probes);
    __syncthreads();  // tables and visited set in place
    if constexpr (SPLIT) pqw_split_walker(a, q, lane, dist, hv);
    else search_body<Walker, 2, false>(a, q, lane, dist, hv, rv);
  } else {
    HashVisited<HCAP == kHash16 ? 4u : HCAP> hv;
    hv.init_nosync(reinterpret_cast<uint32_t *>(lds_f), bits, a.words_per_query, lane, a.hash_limit);
    __syncthreads();
    if constexpr (SPLIT) pqw_split_walker(a, q, lane, dist, hv);
    else search_body<Walker, 2, false>(a, q, lane, dist, hv, rv);
  }
  dist.finish(lane);
}

// The workgroup-per-query walk of a plain store (PlainWideDist): W waves, wave 0 walks.  Dynamic LDS: the visited set's
// table, then the policy's hop scratch and command word.
template <int NG, bool L2, int W, bool FILT>
__global__ __launch_bounds__(64 * W) void k_greedy_search_wide(const SearchArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t q = blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  // dynamic LDS: [search set's hash table][filtered: the result set's][the policy's hop scratch and command words]
  constexpr uint32_t kRWords = FILT ? HashVisited<kHashCapResult>::kWords : 0;
  PlainWideDist<NG, L2, W> dist;
  dist.init_wave(a, q, lane, wave, lds_f + HashVisited<kHashCap>::kWords + kRWords);
  HashVisited<kHashCap> hv;
  HashVisited<kHashCapResult> rv;
  if (wave == 0) {
    hv.init_nosync(reinterpret_cast<uint32_t *>(lds_f), a.bitsets + (size_t)q * a.words_per_query, a.words_per_query, lane,
                   a.hash_limit);
    if constexpr (FILT)
      rv.init_nosync(reinterpret_cast<uint32_t *>(lds_f) + HashVisited<kHashCap>::kWords,
                     a.rbitsets + (size_t)q * a.words_per_query, a.words_per_query, lane, a.hash_limit);
  }
  __syncthreads();
  if (wave != 0) return dist.template serve<HashVisited<kHashCap>>(a, lane, reinterpret_cast<uint32_t *>(lds_f));
  if constexpr (FILT) {
    search_body<PlainWideDist<NG, L2, W>, 2, true>(a, q, lane, dist, hv, rv);
  } else {
    NoVisited nv;
    search_body<PlainWideDist<NG, L2, W>, 2, false>(a, q, lane, dist, hv, nv);
  }
  dist.finish(lane);
}

// host-side launcher: picks the instantiation for (ng, metric, search_size)
int launch_greedy_search(const SearchArgs &a, uint32_t nq, hipStream_t stream);

// true when launch_greedy_search will use the LDS hash visited set (then the bitsets need no clearing)
bool search_uses_hash(const SearchArgs &a, uint32_t nq);

}  // namespace sdb
