// build.hip -- K4: batched Vamana insert (greedy search -> RobustPrune -> back-edges) on device.
//
// Restates insertSinglePoint (shard/index/vamana/insert.go:16-68) and robustPrune
// (shard/index/vamana/search.go:106-138).  The reference runs NumCPU-1 insertSinglePoint workers
// concurrently (vamana.go:190-196) so its graph depends on goroutine interleaving; here inserts are
// applied in deterministic rounds: all points of a round search the same frozen snapshot, prune
// their own candidate lists in parallel, and then every back-edge target B is updated by exactly
// one wavefront, in insert order.  A round of one point is exactly a sequential insertSinglePoint.
//
// All distances use the reference's arithmetic (dist_core.h), so with round_size = 1 the graph is
// identical, edge for edge, to the oracle's sequential build.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <type_traits>

#include "pq.h"
#include "search_kernel.h"

#ifndef SDB_DCACHE_BITS
#define SDB_DCACHE_BITS 13
#endif

namespace sdb {

struct BuildArgs {
  const float *slab;
  uint32_t *adj;
  uint32_t *deg;
  uint32_t *clean;  // [n] leading edges of each row that its last robustPrune produced (see robust_prune_wave)
  float *adjdist;    // [n][64] distFn(row, edge) for the first dcount[row] edges (index.h)
  uint32_t *dcount;  // [n]
  uint32_t dim, nblk, ng, tail, ld;
  int metric;
  float alpha;
  uint32_t R;
  // round
  uint32_t first_slot;  // slot of the round's first new node
  uint32_t nnew;
  const uint32_t *vis_slots;
  const float *vis_dists;
  const uint32_t *vis_count;
  uint32_t vis_cap;
  uint64_t *keys_in;   // [nnew*64] (target slot << 32 | a_idx << 6 | edge position)
  uint64_t *keys_sorted;
  // quantized store (NG == kQuantized): point-to-point distances are sums over the centroid-pair table
  const uint8_t *pq_codes;  // [n][M]
  const float *pq_cdists;   // [M][K][K]
  uint32_t pq_M, pq_K;
  // distances the round's searches evaluated: per new point a direct-mapped table of (slot, distance bits),
  // 2^(32 - dcache_shift) entries, written by K2 (SearchArgs::dcache); NULL when not collected
  const uint2 *dcache;
  uint32_t dcache_shift;
  // targets with at least big_min requests in this round are not handled by their k_backedges wave but listed
  // (position of the first request, count, target, its degree) for the chip-wide prune (bigprune.inc)
  uint32_t *big_count;
  uint4 *big_list;
  uint32_t big_min, big_cap;
  // the start node's overflow edges (index.h h_start_ext): while there are any, the start node takes the chip-wide
  // prune for whatever requests it gets -- the candidate set is its row + the overflow + the requests
  uint32_t start_slot;
  const uint32_t *start_ext;
  uint32_t start_ext_n;
  uint32_t no_tile;           // != 0: new nodes are pruned by k_prune_new (rows from global memory), a test knob
  uint32_t *prune_done;       // [nnew] 0: left to k_prune_new, 2: pair table ready for k_prune_select, 1: pruned; NULL: k_prune_new takes all
  const uint32_t *self_list;  // k_prune_new_tiled for nodes that are not this round's new ones (delete.inc): the node of list q, or NULL
  float *pair_tab;            // [nnew][kTileMaxCand^2] the tiled kernel's pair table of a node (row stride = its list length)
  uint32_t *pair_slots;       // [nnew][kTileMaxCand] its sorted visit list
  float *pair_dists;
  uint8_t *dirty;             // [n] set for every row whose adjacency this call writes (index.h graph versions)
  unsigned long long *stats;  // sdb_index_build_stats counters (index.h d_bstats), or NULL
  uint32_t *flags;            // [0] bit 0: a search's visit log did not fit vis_cap
  // Back-edge targets whose re-prune cannot be settled from a few rows of pair distances -- a node that overflows for
  // the first time (its edges have never been pruned against each other), or more than kMaxDirty candidates that arrived
  // since its last prune -- are DEFERRED by k_backedges: it leaves the candidate list (slots + distances from the node)
  // here, and the LDS-tiled prune (k_prune_new_tiled with a node list, what the delete path uses) takes them in one
  // launch behind it: the candidates' rows are read ONCE into LDS (~65 rows) where the wave's pick-by-pick walk read
  // every pair's row from L2 / HBM (~600 rows per such node).  NULL: every re-prune runs in k_backedges.
  // Pair distances of APPENDED edges, kept from the round that appended them.  A target with room takes a new point
  // without a prune (insert.go:62); when it overflows later, that edge is a candidate that "arrived since the last
  // prune" and robust_prune_wave needs its distances to all other candidates -- a whole row of pair distances, 65 rows
  // read from HBM, for a point whose own search had evaluated nearly all of them in the round it arrived (the target
  // was expanded by that search, so its neighbours were looked at) and whose table is gone by now.  So the append
  // copies them out of the table while it is there: pairc[node][p - clean][e] = distFn(edge p, edge e) for e < p, for
  // the up to kMaxDirty positions p behind the node's clean prefix; all-ones bits: not known.  Full-precision store
  // only; lives for one insert_batch call (a cache: what is missing is computed from rows).  NULL: none.
  float *pairc;               // [rows][kMaxDirty][64]
  uint32_t *def_count;        // [0] targets deferred this round (may exceed def_cap: the excess was pruned in place)
  uint32_t def_cap;
  uint32_t *def_self;         // [def_cap] the target's slot
  uint32_t *def_nc;           // [def_cap] its candidates (0: nothing here)
  uint32_t *def_slots;        // [def_cap][kTileMaxCand]
  float *def_dists;           // [def_cap][kTileMaxCand]
  uint32_t *def_done;         // [def_cap] the tiled kernel's per-list state (prune_done)
};

// sdb_index_build_stats slots
enum { kStSearchDist = 0, kStSearchEdges, kStPrunePairs, kStBackPairs, kStBackCached, kStRequests, kStReprunes,
       kStAppends, kStStagedRows, kStRounds, kStHubs };

__device__ __forceinline__ void stat_add(const BuildArgs &a, int slot, unsigned long long v, int lane) {
  if (a.stats && lane == 0 && v) atomicAdd(a.stats + (blockIdx.x & 63u) * 16u + slot, v);  // index.h kStatCopies
}

constexpr int kQuantized = -2;  // value of the NG template parameter for a fitted product quantizer

// productQuantizer.DistanceFromPoint (product.go:296-304): sequential fp32 adds over the sub-quantizers
__device__ __forceinline__ float pq_sym_dist(const BuildArgs &a, const uint8_t *__restrict__ cx,
                                             const uint8_t *__restrict__ cy) {
  float dist = 0.0f;
  for (uint32_t i = 0; i < a.pq_M; i++) dist += a.pq_cdists[((size_t)i * a.pq_K + cx[i]) * a.pq_K + cy[i]];
  return dist;
}

constexpr uint64_t kNoKey = ~0ull;

// Distance source for a bound point p (vecStore.DistanceFromPoint, plain.go:87-97): p's slab row in
// registers (NG >= 0) or in LDS (NG == -1).
template <int NG>
struct PointRow {
  float4 xq[NG > 0 ? NG : 1];
  float xt;
};

// robustPrune (search.go:106-138) by one wavefront.
//   in_slot/in_dist [nc]   candidates in arrival order (LDS)
//   s_*                    LDS scratch for the sorted copy
// The candidate list is first sorted by distance, stable (DistSet.Sort distset.go:223-238: insertion
// sort with strict '<', so equal distances keep arrival order).  Then for each surviving candidate in
// order: add it as an edge (:118), stop at DegreeBound (:119-121), and mark every later candidate j
// with alpha * dist(p*, c_j) < c_j.Distance as removed (:132).
constexpr int kPairMax = 80;  // largest candidate set of the few-new-candidates mode (rows of D)

constexpr int kMaxDirty = 16;  // "few new candidates" mode of robust_prune_wave
// rows of up to 768 floats keep the pair distances of appended edges (BuildArgs::pairc).  At 768 floats the look-ups'
// registers do not fit k_backedges' budget of three waves per SIMD (1 spilled): those rows run two waves per SIMD, which
// the rows not read more than pay for (1M x 768 build 3.05 - 3.26 -> 2.86 - 3.07 s)
template <int NG>
constexpr bool kPairRecords = NG >= 0 && NG <= 6;

// D: LDS scratch of kMaxDirty rows of kPairMax pair distances for the few-new-candidates mode, or nullptr.
//
// n_clean: the first n_clean input candidates are this node's edges as its LAST robustPrune left them.  For two
// such candidates i < j (sorted order) that prune evaluated alpha * dist(c_i, c_j) < c_j.Distance and found it
// false -- c_j would not be an edge otherwise -- and it would find the same again: same vectors, same
// arithmetic, same order (the key dist(node, .) never changes and ties keep edge order).  So only pairs with
// a candidate that arrived since (appended edges, the new points) are evaluated: one row of distances per
// such candidate instead of the whole triangle.  The graph that results is bit-for-bit the one the full
// evaluation gives; the sequential-build parity tests cover it.
template <int NG, bool L2>
__device__ void robust_prune_wave(const BuildArgs &a, uint32_t self_slot, int nc, const uint32_t *in_slot,
                                  const float *in_dist, uint32_t *s_slot, float *s_dist, uint32_t *s_rem,
                                  float *qs, int lane, float *D = nullptr, int n_clean = 0,
                                  bool dists_are_point_to_point = true, uint32_t *n_eval = nullptr,
                                  uint32_t *n_cached = nullptr, const float *pairc_row = nullptr, int n_row = 0) {
  // pairc_row / n_row: the node's record of pair distances of appended edges (BuildArgs::pairc) and how many of the
  // input candidates are its row's entries, in edge order (the rest are this round's new points)
  uint32_t ev = 0, ca = 0;  // pair distances evaluated / taken from the searches' tables (sdb_index_build_stats)
  uint32_t *s_org = (D && nc <= kPairMax) ? reinterpret_cast<uint32_t *>(D + kMaxDirty * kPairMax + kMaxDirty) : nullptr;
  constexpr int U = NG >= 0 ? ChunkPairs<NG, false>::value : 4;
  const int L = lane & 31;
  const bool any_nan = wave_any_nan(in_dist, nc, lane);
  for (int i = lane; i < nc; i += 64) {
    const float d = in_dist[i];
    const int rank = dist_sort_rank(in_dist, nc, i, any_nan);
    s_slot[rank] = in_slot[i];
    s_dist[rank] = d;
    s_rem[rank] = i < n_clean ? 2u : 0u;  // bit 0: pruneRemoved (distset.go:124), bit 1: clean
    if (s_org) s_org[rank] = (uint32_t)i;  // where the candidate stood in the input (a row entry's edge position)
  }
  __syncthreads();
  // sparse mode: few dirty candidates and room for their distance rows in D
  bool sparse = false;
  uint32_t *dord = const_cast<uint32_t *>(in_slot);  // input arrays are free after the sort
  uint32_t my_out = kNoSlot;  // lane e holds edge e of the new row
  float my_outd = 0.0f;       // and its distance from `self`
  int cnt = 0;
  bool decided = false;  // the sparse mode below settles the whole prune without walking the neighbours
  if constexpr (NG >= 0) {
    if (D && n_clean > 0 && nc - n_clean <= kMaxDirty && nc <= kPairMax) {
      sparse = true;
      int nd = 0;
      // positions of the dirty candidates, in LDS behind D's rows (sixteen scalar registers and a select chain per
      // look-up otherwise -- the kernel sat two registers over its three-waves-per-SIMD budget)
      uint32_t *dpos = reinterpret_cast<uint32_t *>(D + kMaxDirty * kPairMax);
      for (int base = 0; base < nc; base += 64) {
        const int j = base + lane;
        const bool dirty = j < nc && !(s_rem[j] & 2u);
        const uint64_t m = __ballot(dirty);
        const uint32_t ord = (uint32_t)(nd + __popcll(m & ((1ull << lane) - 1)));
        if (j < nc) dord[j] = dirty ? ord : 0xFFFFFFFFu;
        if (dirty) dpos[ord] = (uint32_t)j;
        nd += __popcll(m);
      }
      __syncthreads();
      // one row of pair distances per dirty candidate: D[k][j] = distFn(c_dirty_k, c_j).  When the dirty
      // candidate is a point of this round, its own search evaluated most of these a moment ago (the node
      // being re-pruned was expanded there, so its neighbours were looked at): BuildArgs::dcache holds what
      // survived in that search's direct-mapped table -- exact values, same arithmetic (bitwise symmetric) --
      // and only the misses are computed from rows.
      uint32_t *miss = reinterpret_cast<uint32_t *>(const_cast<float *>(in_dist));  // free after the sort
      for (int k = 0; k < nd; k++) {
        const int dk = __builtin_amdgcn_readfirstlane((int)dpos[k]);
        const uint32_t sd = s_slot[dk];
        const uint2 *tab = nullptr;
        if (a.dcache && sd >= a.first_slot && sd - a.first_slot < a.nnew)
          tab = a.dcache + ((size_t)(sd - a.first_slot) << (32 - a.dcache_shift));
        // ... or an edge that was appended in an earlier round: its pair distances to the row's other entries were
        // copied out of its search table then (BuildArgs::pairc), and this round's new points have theirs to it
        const int ok = __builtin_amdgcn_readfirstlane((int)s_org[dk]);
        const bool k_in_row = kPairRecords<NG> && pairc_row && ok < n_row;
        int nmiss = 0;
        for (int base = 0; base < nc; base += 64) {
          const int j = base + lane;
          bool need = j < nc && j != dk;  // (a candidate's distance to itself is never asked for)
          if (need && tab) {
            const uint32_t cs = s_slot[j];
            const uint2 e = tab[(cs * 2654435761u) >> a.dcache_shift];
            if (e.x == cs) D[k * kPairMax + j] = __uint_as_float(e.y), need = false;
          } else if (need && k_in_row) {
            const int oj = (int)s_org[j];
            if (oj < n_row) {  // both are row entries: the record of the later one holds the pair
              const int hi = oj > ok ? oj : ok, lo = oj > ok ? ok : oj, rec = hi - n_clean;
              if (rec >= 0 && rec < kMaxDirty) {
                const uint32_t v = __float_as_uint(pairc_row[rec * 64 + lo]);
                if (v != 0xFFFFFFFFu) D[k * kPairMax + j] = __uint_as_float(v), need = false;
              }
            } else if (a.dcache) {  // a new point of this round: its search's table, looked up for the appended edge
              const uint32_t cs = s_slot[j];
              if (cs >= a.first_slot && cs - a.first_slot < a.nnew) {
                const uint2 e = (a.dcache + ((size_t)(cs - a.first_slot) << (32 - a.dcache_shift)))[(sd * 2654435761u) >> a.dcache_shift];
                if (e.x == sd) D[k * kPairMax + j] = __uint_as_float(e.y), need = false;
              }
            }
          }
          const uint64_t mm = __ballot(need);
          if (need) miss[nmiss + __popcll(mm & ((1ull << lane) - 1))] = (uint32_t)j;
          nmiss += __popcll(mm);
        }
        __syncthreads();
        ev += (uint32_t)nmiss, ca += (uint32_t)(nc - nmiss);
        if (nmiss) {
          PointRow<NG> pr;
          const float *prow = a.slab + (size_t)sd * a.ld;
#pragma unroll
          for (int g = 0; g < NG; g++) pr.xq[g] = reinterpret_cast<const float4 *>(prow)[g * 32 + L];
          if (NG == 0) pr.xq[0] = make_float4(0.f, 0.f, 0.f, 0.f);
          pr.xt = a.tail ? prow[NG * 128 + L] : 0.0f;
          for (int m0 = 0; m0 < nmiss; m0 += 2 * U) {
            // each half-wave its own candidate of a pair: one index per pair and lane (not two wave-wide ones), and the
            // sum leaves lane 0 of its half straight to its place
            uint32_t slot[U];
            float res[U];
            int cidx[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
              const int at = m0 + 2 * u + (lane >> 5);
              cidx[u] = (int)miss[at < nmiss ? at : nmiss - 1];
              slot[u] = s_slot[cidx[u]];
            }
            chunk_dist<NG, L2, U>(a.slab, a.ld, a.tail, pr.xq, pr.xt, slot, res, lane);
            if (L == 0) {
#pragma unroll
              for (int u = 0; u < U; u++) D[k * kPairMax + cidx[u]] = metric_finish(res[u], a.metric);
            }
          }
        }
        __syncthreads();  // `miss` is reused by the next dirty candidate
      }
      __syncthreads();
      // With the clean pairs settled, a candidate is removed only by a dirty candidate before it, or -- if it
      // is dirty itself -- by any selected candidate before it.  So the dirty candidates are taken in order
      // (there are at most kMaxDirty): each is tested against everything alive before it (one ballot over
      // its row of D), and if it survives it strikes out what it beats further on (one lane-parallel step).
      // What is alive at the end, in order, cut at the degree bound, is the row: the candidate at which the
      // sequential loop would break (:119-121) and everything after it are not among the first R alive, and
      // removals only ever reach forward, so nothing past the break can change what comes before it.
      {
        const int j0 = lane, j1 = lane + 64;  // nc <= kPairMax <= 128
        const bool v0 = j0 < nc, v1 = j1 < nc;
        const uint32_t sl0 = v0 ? s_slot[j0] : kNoSlot, sl1 = v1 ? s_slot[j1] : kNoSlot;
        const float sd0 = v0 ? s_dist[j0] : 0.0f, sd1 = v1 ? s_dist[j1] : 0.0f;
        const bool ok0 = v0 && sl0 != self_slot, ok1 = v1 && sl1 != self_slot;  // :115-117
        bool rem0 = false, rem1 = false;
        for (int k = 0; k < nd; k++) {
          const int pd = __builtin_amdgcn_readfirstlane((int)dpos[k]);
          const float dpd = s_dist[pd];
          const float e0 = v0 ? D[k * kPairMax + j0] : 0.0f, e1 = v1 ? D[k * kPairMax + j1] : 0.0f;
          const bool kills0 = ok0 && !rem0 && j0 < pd && a.alpha * e0 < dpd;  // :132 seen from the victim
          const bool kills1 = ok1 && !rem1 && j1 < pd && a.alpha * e1 < dpd;
          if (__ballot(kills0 || kills1) || s_slot[pd] == self_slot) {
            rem0 |= j0 == pd, rem1 |= j1 == pd;
          } else {
            rem0 |= v0 && j0 > pd && a.alpha * e0 < sd0;  // :132
            rem1 |= v1 && j1 > pd && a.alpha * e1 < sd1;
          }
        }
        const bool al0 = ok0 && !rem0, al1 = ok1 && !rem1;
        const uint64_t m0 = __ballot(al0), m1 = __ballot(al1);
        const uint32_t r0 = (uint32_t)__popcll(m0 & ((1ull << lane) - 1));
        const uint32_t r1 = (uint32_t)(__popcll(m0) + __popcll(m1 & ((1ull << lane) - 1)));
        const int alive = __popcll(m0) + __popcll(m1);
        cnt = alive < (int)a.R ? alive : (int)a.R;
        uint32_t *o_slot = dord;  // both input arrays are free now
        float *o_dist = const_cast<float *>(in_dist);
        __syncthreads();
        if (al0 && r0 < a.R) o_slot[r0] = sl0, o_dist[r0] = sd0;
        if (al1 && r1 < a.R) o_slot[r1] = sl1, o_dist[r1] = sd1;
        __syncthreads();
        if (lane < cnt) my_out = o_slot[lane], my_outd = o_dist[lane];
        decided = true;
      }
    } else {
      D = nullptr;  // the scratch holds kMaxDirty rows, not a full matrix: pairs are computed as they come
    }
  }
  int i = 0;
  while (!decided && i < nc) {
    int found = -1;
    for (int base = i & ~63; base < nc && found < 0; base += 64) {
      const int j = base + lane;
      const bool ok = j >= i && j < nc && !(s_rem[j] & 1u) && s_slot[j] != self_slot;  // :115-117
      const uint64_t m = __ballot(ok);
      if (m) found = base + __ffsll((unsigned long long)m) - 1;
    }
    if (found < 0) break;
    const uint32_t p = s_slot[found];
    if (lane == cnt) my_out = p, my_outd = s_dist[found];  // node.AddNeighbour :118
    cnt++;
    if (cnt >= (int)a.R) break;  // :119-121
    if (sparse) {  // only pairs with a dirty candidate can prune; their distances are rows of D
      const uint32_t fo = dord[found];
      for (int base = (found + 1) & ~63; base < nc; base += 64) {
        const int j = base + lane;
        if (j > found && j < nc && !(s_rem[j] & 1u)) {
          const uint32_t jo = dord[j];
          if (fo != 0xFFFFFFFFu || jo != 0xFFFFFFFFu) {
            // distFn(c_found, c_j); the arithmetic is bitwise symmetric in its two arguments
            const float d = fo != 0xFFFFFFFFu ? D[fo * kPairMax + j] : D[jo * kPairMax + found];
            if (a.alpha * d < s_dist[j]) s_rem[j] |= 1u;  // :132
          }
        }
      }
      __syncthreads();
      i = found + 1;
      continue;
    }
    if (D) {  // pair distances are in LDS: the sweep is a lookup
      for (int base = (found + 1) & ~63; base < nc; base += 64) {
        const int j = base + lane;
        if (j > found && j < nc && !(s_rem[j] & 1u) && a.alpha * D[found * kPairMax + j] < s_dist[j]) s_rem[j] |= 1u;  // :132
      }
      __syncthreads();
      i = found + 1;
      continue;
    }
    if constexpr (NG == kQuantized) {  // one candidate per lane: M table lookups each
      const uint8_t *cp = a.pq_codes + (size_t)p * a.pq_M;
      for (int base = (found + 1) & ~63; base < nc; base += 64) {
        const int j = base + lane;
        const bool lv = j > found && j < nc && !(s_rem[j] & 1u);
        ev += (uint32_t)__popcll(__ballot(lv));
        if (lv) {
          const float d = pq_sym_dist(a, cp, a.pq_codes + (size_t)s_slot[j] * a.pq_M);
          if (a.alpha * d < s_dist[j]) s_rem[j] |= 1u;  // :132
        }
      }
      __syncthreads();
      i = found + 1;
      continue;
    }
    // bind p (DistanceFromPoint :124)
    PointRow<NG> pr;
    const float *prow = a.slab + (size_t)p * a.ld;
    if constexpr (NG >= 0) {
#pragma unroll
      for (int g = 0; g < NG; g++) pr.xq[g] = reinterpret_cast<const float4 *>(prow)[g * 32 + L];
      if (NG == 0) pr.xq[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      pr.xt = a.tail ? prow[NG * 128 + L] : 0.0f;
    } else {
      __syncthreads();
      for (uint32_t t = lane; t < a.ld; t += 64) qs[t] = prow[t];
      __syncthreads();
    }
    // The candidates still alive behind `found`, 64 positions at a time: their slots go into a list by rank (the input
    // arrays are free after the sort), each half-wave takes its candidate of a pair from there, and the sums come back
    // through the list's twin -- no per-pair scalar bookkeeping (sixteen lane numbers in scalar registers, a readlane
    // per row and result: the kernel sat two registers over its budget of three waves per SIMD and spilled them).
    uint32_t *lst = const_cast<uint32_t *>(in_slot);
    float *lres = const_cast<float *>(in_dist);
    for (int base = (found + 1) & ~63; base < nc; base += 64) {
      const int j = base + lane;
      const bool live = j > found && j < nc && !(s_rem[j] & 1u);
      const uint64_t todo = __ballot(live);
      const int nl = __popcll(todo);
      if (nl == 0) continue;
      ev += (uint32_t)nl;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
      if (live) lst[rank] = s_slot[j];
      __syncthreads();
      for (int m0 = 0; m0 < nl; m0 += 2 * U) {
        uint32_t slot[U];
        float res[U];
        int at[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int x = m0 + 2 * u + (lane >> 5);
          at[u] = x < nl ? x : nl - 1;
          slot[u] = lst[at[u]];
        }
        if constexpr (NG >= 0) chunk_dist<NG, L2, U>(a.slab, a.ld, a.tail, pr.xq, pr.xt, slot, res, lane);
        else chunk_dist_lds<L2, U>(a.slab, a.ld, a.ng, a.tail, qs, slot, res, lane);
        if (L == 0) {
#pragma unroll
          for (int u = 0; u < U; u++) lres[at[u]] = metric_finish(res[u], a.metric);
        }
      }
      __syncthreads();
      if (live && a.alpha * lres[rank] < s_dist[j]) s_rem[j] |= 1u;  // :132
      __syncthreads();  // the list is written again for the next 64 positions
    }
    __syncthreads();
    i = found + 1;
  }
  // node.edges of `self`, kNoSlot padded; every edge of a freshly pruned row is "clean"
  a.adj[(size_t)self_slot * kAdjStride + lane] = lane < cnt ? my_out : kNoSlot;
  a.adjdist[(size_t)self_slot * kAdjStride + lane] = my_outd;
  if (lane == 0) {
    if (a.dirty) a.dirty[self_slot] = 1;
    a.deg[self_slot] = (uint32_t)cnt;
    a.clean[self_slot] = (uint32_t)cnt;
    // the candidates' distances are distFn(self, .) except for a new node of a quantized store, whose
    // candidate list carries the search's LUT distances (DistanceFromFloat), not DistanceFromPoint
    a.dcount[self_slot] = dists_are_point_to_point ? (uint32_t)cnt : 0u;
  }
  if (n_eval) *n_eval += ev;
  if (n_cached) *n_cached += ca;
}

// dynamic LDS carve shared by both prune kernels
struct PruneLds {
  uint32_t *in_slot;
  float *in_dist;
  uint32_t *s_slot;
  float *s_dist;
  uint32_t *s_rem;
  float *qs;
  float *D;  // [kMaxDirty][kPairMax], only carved by k_backedges
  __device__ PruneLds(char *base, uint32_t cap, bool with_pairs = false) {
    D = with_pairs ? reinterpret_cast<float *>(base + (size_t)cap * 20) : nullptr;
    base_init(base, cap);
    if (with_pairs) qs = D + kMaxDirty * kPairMax + kMaxDirty + kPairMax;  // (behind D's rows: the dirty candidates' positions, the candidates' input positions)
  }
  __device__ void base_init(char *base, uint32_t cap) {
    in_slot = reinterpret_cast<uint32_t *>(base);
    in_dist = reinterpret_cast<float *>(in_slot + cap);
    s_slot = reinterpret_cast<uint32_t *>(in_dist + cap);
    s_dist = reinterpret_cast<float *>(s_slot + cap);
    s_rem = reinterpret_cast<uint32_t *>(s_dist + cap);
    qs = reinterpret_cast<float *>(s_rem + cap);
  }
};

// in_dist[c] = DistanceFromPoint(point)(in_slot[c]) for the nc candidates staged in LDS (plain.go:87-97 /
// product.go:279-305).  Callers have synchronised after filling in_slot; synchronises before returning.
template <int NG, bool L2>
__device__ void dists_from_point(const BuildArgs &a, uint32_t point, uint32_t nc, const PruneLds &l, int lane) {
  if constexpr (NG == kQuantized) {
    const uint8_t *cp = a.pq_codes + (size_t)point * a.pq_M;
    for (uint32_t c = lane; c < nc; c += 64) l.in_dist[c] = pq_sym_dist(a, cp, a.pq_codes + (size_t)l.in_slot[c] * a.pq_M);
  } else {
    constexpr int U = NG >= 0 ? ChunkPairs<NG, false>::value : 4;
    const int L = lane & 31;
    PointRow<NG> pr;
    const float *arow = a.slab + (size_t)point * a.ld;
    if constexpr (NG >= 0) {
#pragma unroll
      for (int g = 0; g < NG; g++) pr.xq[g] = reinterpret_cast<const float4 *>(arow)[g * 32 + L];
      if (NG == 0) pr.xq[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      pr.xt = a.tail ? arow[NG * 128 + L] : 0.0f;
    } else {
      for (uint32_t x = lane; x < a.ld; x += 64) l.qs[x] = arow[x];
      __syncthreads();
    }
    for (uint32_t c0 = 0; c0 < nc; c0 += 2 * U) {
      uint32_t slot[U];
      float res[U];
      uint32_t cidx[2 * U];
#pragma unroll
      for (int k2 = 0; k2 < 2 * U; k2++) cidx[k2] = (c0 + k2 < nc) ? c0 + k2 : nc - 1;
#pragma unroll
      for (int u = 0; u < U; u++) slot[u] = lane < 32 ? l.in_slot[cidx[2 * u]] : l.in_slot[cidx[2 * u + 1]];
      if constexpr (NG >= 0) chunk_dist<NG, L2, U>(a.slab, a.ld, a.tail, pr.xq, pr.xt, slot, res, lane);
      else chunk_dist_lds<L2, U>(a.slab, a.ld, a.ng, a.tail, l.qs, slot, res, lane);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const float d0 = metric_finish(rlf(res[u], 0), a.metric);
        const float d1 = metric_finish(rlf(res[u], 32), a.metric);
        if (lane == 0) l.in_dist[cidx[2 * u]] = d0, l.in_dist[cidx[2 * u + 1]] = d1;
      }
    }
  }
  __syncthreads();
}

static size_t prune_lds_bytes(uint32_t cap, int NG, uint32_t ld, bool with_pairs = false) {
  return (size_t)cap * 20 + (with_pairs ? ((size_t)kMaxDirty * (kPairMax + 1) + kPairMax) * 4 : 0) + (NG == -1 ? (size_t)ld * 4 + 16 : 0);
}

// robustPrune(nodeA, visitedSet) for every new node of the round (insert.go:29-31), then emit the
// back-edge requests (insert.go:36) as sortable keys.
template <int NG, bool L2>
__global__ __launch_bounds__(64) void k_prune_new(const BuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  PruneLds l(lds_raw, a.vis_cap);
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  if (a.prune_done && a.prune_done[q]) return;  // k_prune_new_tiled has pruned this node
  const uint32_t self = a.first_slot + q;
  uint32_t nc = a.vis_count[q];
  if (nc > a.vis_cap) {  // the reference's visited list is unbounded (AddAlreadyUnique): never prune a cut one silently
    if (lane == 0) atomicOr(a.flags, 1u);
    nc = a.vis_cap;
  }
  for (uint32_t i = lane; i < nc; i += 64) {
    l.in_slot[i] = a.vis_slots[(size_t)q * a.vis_cap + i];
    l.in_dist[i] = a.vis_dists[(size_t)q * a.vis_cap + i];
  }
  __syncthreads();
  uint32_t n_eval = 0;
  robust_prune_wave<NG, L2>(a, self, (int)nc, l.in_slot, l.in_dist, l.s_slot, l.s_dist, l.s_rem, l.qs, lane, nullptr, 0,
                            NG != kQuantized, &n_eval);
  stat_add(a, kStPrunePairs, n_eval, lane);
  const uint32_t nb = a.adj[(size_t)self * kAdjStride + lane];  // this lane wrote it
  a.keys_in[(size_t)q * 64 + lane] =
      nb == kNoSlot ? kNoKey : ((uint64_t)nb << 32) | ((uint64_t)q << 6) | (uint64_t)lane;
}


// ---- robustPrune of the new nodes with the candidate rows staged in LDS ---------------------------------------
// The robustPrune of a new node walks its sorted visit list (~79 entries at searchSize 75), selects ~52 edges and
// on the way evaluates ~1 850 pair distances among those 79 rows: every row is an operand ~23 times.  k_prune_new
// fetches the row from global memory each time (2.8 TB per 1M inserts, served mostly by L2).  Here one 256-thread
// workgroup per new node gathers the list's rows ONCE into an LDS tile (79 x 1 536 B = 121 KB of the CU's 160 KB
// at d = 384) and works in two phases:
//   A. every pair distance of the list, d(c_i, c_j) for i < j, by all four waves -- a half-wave per pair, row i in
//      registers, row j from the tile, U pairs in flight.  The selection below needs about 60 % of them (which
//      ones is only known as it goes); computing all of them removes every dependency from the arithmetic, and
//      with one workgroup per CU (the tile fills its LDS) a dependent LDS round trip per selected edge is what the
//      chip cannot hide.  Same 32-chain arithmetic (dist_core.h), so the same bits the one-by-one walk computes.
//   B. the selection loop (search.go:113-137) by one wave with the list in registers: the sweep of a selected
//      candidate is one LDS read of its row of the pair table per lane.
// The graph is the one k_prune_new builds, edge for edge (tests/test_gpu_build.py).  The tile holds the first rows
// of the SORTED list; what does not fit (wide rows) stays in global memory and is read from there in phase A.
// Lists longer than kTileMaxCand are left to k_prune_new (BuildArgs::prune_done tells it which).
constexpr uint32_t kTileLdsBytes = 160 * 1024;
constexpr int kTileMaxCand = 128;
constexpr uint32_t kTileFixedBytes = kTileMaxCand * 16;  // s_slot, s_dist, in_slot, in_dist

template <int NG, bool L2, bool TAIL, int NW>
__global__ __launch_bounds__(NW * 64) void k_prune_new_tiled(const BuildArgs a) {
  static_assert(NG >= 1, "register-row kernels only");
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  uint32_t *s_slot = reinterpret_cast<uint32_t *>(lds_raw);
  float *s_dist = reinterpret_cast<float *>(s_slot + kTileMaxCand);
  uint32_t *in_slot = reinterpret_cast<uint32_t *>(s_dist + kTileMaxCand);
  float *in_dist = reinterpret_cast<float *>(in_slot + kTileMaxCand);
  float *D = reinterpret_cast<float *>(lds_raw + kTileFixedBytes);  // [nc][nc], upper triangle used
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, L = lane & 31, half = lane >> 5;
  const uint32_t q = blockIdx.x;
  const uint32_t self = a.self_list ? a.self_list[q] : a.first_slot + q;
  const uint32_t nc_raw = a.vis_count[q];
  if (nc_raw > (uint32_t)kTileMaxCand || nc_raw == 0) {  // a long list: the one-wave kernel prunes it (an empty one: it has)
    if (tid == 0) a.prune_done[q] = 0u;
    return;
  }
  const int nc = (int)nc_raw;
  // ---- DistSet.Sort (distset.go:223-238): stable by distance, rank by counting
  if (tid < nc) {
    in_slot[tid] = a.vis_slots[(size_t)q * a.vis_cap + tid];
    in_dist[tid] = a.vis_dists[(size_t)q * a.vis_cap + tid];
  }
  __syncthreads();
  const bool any_nan = wave_any_nan(in_dist, nc, lane);  // every wave looks at the whole list: no LDS word to share
  if (tid < nc) {
    const float d = in_dist[tid];
    const int rank = dist_sort_rank(in_dist, nc, tid, any_nan);
    s_slot[rank] = in_slot[tid], s_dist[rank] = d;
  }
  __syncthreads();
  // ---- gather the first nt rows of the sorted list into the tile: a half-wave per row, a batch of loads in
  // flight before the first LDS write
  const uint32_t d_bytes = ((uint32_t)(nc * nc) * 4u + 15u) & ~15u;
  float *tile = reinterpret_cast<float *>(lds_raw + kTileFixedBytes + d_bytes);
  // rows sit 16 bytes further apart than they are long: a column of 16-byte pieces (what the matrix-core phase A
  // reads: the same piece of 16 rows) then falls on different banks instead of all on one
  const uint32_t tpitch = a.ld * 4u + 16u;
  const int fit = (int)((kTileLdsBytes - kTileFixedBytes - d_bytes - 1024u) / tpitch);  // 1 KB: DMA overhang
  const int nt = nc < fit ? nc : fit;
  {
    // LDS-DMA (global_load_lds_dwordx4): a wave instruction moves 64 x 16 B straight into 1 KB of the tile, no
    // registers in between, so a wave keeps all its ~30 pieces in flight at once.  The destination is lane-linear;
    // the source is per lane -- a piece may straddle two rows, each lane finds its own (row, column).  Row slots
    // come from LDS: an ordinary global load in this loop would make the compiler drain the DMAs at every use.
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const uint32_t rowb = a.ld * 4u, total = (uint32_t)nt * tpitch;
    char *tile_b = reinterpret_cast<char *>(tile);
    for (uint32_t piece = (uint32_t)wave; piece * 1024u < total; piece += NW) {
      uint32_t o = piece * 1024u + (uint32_t)lane * 16u;
      if (o >= total) o = total - 16u;  // the last piece's overhang lands in unused LDS behind the tile
      const uint32_t r = o / tpitch;
      uint32_t c = o - r * tpitch;
      if (c >= rowb) c = 0;  // the 16 bytes between two rows: any valid source
      const char *src = reinterpret_cast<const char *>(a.slab) + (size_t)s_slot[r] * rowb + c;
      __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(tile_b + piece * 1024u), 16, 0, 0);
    }
  }
  __syncthreads();  // waits for this wave's DMAs (vmcnt) and for everybody else's
  // ---- phase A: D[i][j] = distFn(c_i, c_j), i < j.  Rows i are dealt to the four waves in a zigzag (row i has
  // nc - 1 - i pairs: 0,7,8,15,.. / 1,6,9,14,.. come to the same total within a row's length); the two halves of
  // a wave share the row (the bound point of DistanceFromPoint, plain.go:87-97) and take its pairs alternately.
  constexpr int U = NG <= 3 ? 4 : NG <= 6 ? 2 : 1;  // pairs in flight per half-wave
  const uint32_t rowb = a.ld * 4u;
  const char *tileL = reinterpret_cast<const char *>(tile) + L * 16;
  const char *slabL = reinterpret_cast<const char *>(a.slab) + L * 16;
  auto row_pairs = [&](auto all_in_tile, int i, const float4(&xq)[NG], float xt) {
    constexpr bool ALLT = decltype(all_in_tile)::value;
    const int n = nc - 1 - i;
    for (int t0 = 0; 2 * t0 < n; t0 += U) {
      int jj[U];
      bool ok[U];
      float4 y[U][NG];
      float yt[TAIL ? U : 1];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int k = 2 * (t0 + u) + half;
        ok[u] = k < n;
        jj[u] = i + 1 + (ok[u] ? k : 0);  // past the end: the row's first pair again, computed and dropped
        if (ALLT || jj[u] < nt) {
          const char *row = tileL + (uint32_t)jj[u] * tpitch;
#pragma unroll
          for (int g = 0; g < NG; g++) y[u][g] = *reinterpret_cast<const float4 *>(row + g * 512);
          if constexpr (TAIL) yt[u] = *reinterpret_cast<const float *>(row + NG * 512 - L * 12);
        } else {
          const char *row = slabL + (size_t)s_slot[jj[u]] * rowb;
#pragma unroll
          for (int g = 0; g < NG; g++) y[u][g] = *reinterpret_cast<const float4 *>(row + g * 512);
          if constexpr (TAIL) yt[u] = *reinterpret_cast<const float *>(row + NG * 512 - L * 12);
        }
      }
      // all U reduce trees first, stores afterwards: with nothing conditional in between, the compiler interleaves
      // the trees (a DPP add has to wait two slots for its operand; another tree's add fills them)
      float raw[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        float acc = 0.0f;
#pragma unroll
        for (int g = 0; g < NG; g++) acc = chain4<L2>(acc, xq[g], y[u][g]);
        float t = 0.0f;
        if constexpr (TAIL) t = tail_chain<L2>(xt, yt[u], a.tail, lane);
        raw[u] = asm_reduce(acc, t, lane);
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (L == 0 && ok[u]) D[i * nc + jj[u]] = raw[u];  // the raw sum; phase B applies the metric (distance.go:19-25)
    }
  };
  // dot / cosine rows of whole blocks, all in the tile: phase A on the matrix cores.  d(c_i, c_j) for a 16 x 16 block of
  // pairs is the same product of partial-sum chains as the exact scan's (flat.hip, k_flat_scan_mfma): one
  // v_mfma_f32_16x16x1 = one fused multiply-add of every chain of 256 pairs, its four blocks the quarters t mod 4 of
  // the 32 partial sums, the reduce tree in the lane.  Both operands come from the tile (the same 16-byte piece of 16
  // rows per read); blocks of pairs (I <= J) are dealt to the waves in turn.
  bool by_mfma = false;
  if constexpr (!L2 && !TAIL && NG <= 4) by_mfma = nt == nc;
  if (by_mfma) {
    if constexpr (!L2 && !TAIL && NG <= 4) {
      typedef float f16v __attribute__((ext_vector_type(16)));
      typedef float f4v __attribute__((ext_vector_type(4)));
      const int nb16 = (nc + 15) >> 4;
      const uint32_t nblk = a.nblk;
      const uint32_t lane_off = 16u * (uint32_t)(lane >> 4);  // the quarter's 16-byte piece inside a (a, tt) pair of pieces
      int p = 0;
      for (int I = 0; I < nb16; I++)
        for (int J = I; J < nb16; J++, p++) {
          if (p % NW != wave) continue;  // wave-uniform
          const int ri = min(16 * I + (lane & 15), nc - 1), rj = min(16 * J + (lane & 15), nc - 1);
          const char *pa = reinterpret_cast<const char *>(tile) + (uint32_t)ri * tpitch + lane_off;
          const char *pb = reinterpret_cast<const char *>(tile) + (uint32_t)rj * tpitch + lane_off;
          f16v acc[8];
#pragma unroll
          for (int k = 0; k < 8; k++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[k][r] = 0.0f;
#pragma unroll
          for (int g = 0; g < NG; g++) {
            f4v A[8], B[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {  // accumulator set k = 2a + tt: partial sums 8a + 4tt + quarter
              A[k] = *reinterpret_cast<const f4v *>(pa + g * 512 + (k >> 1) * 128 + (k & 1) * 64);
              B[k] = *reinterpret_cast<const f4v *>(pb + g * 512 + (k >> 1) * 128 + (k & 1) * 64);
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
              if ((uint32_t)(4 * g + kk) < nblk) {  // uniform
#pragma unroll
                for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_mfma_f32_16x16x1f32(A[k][kk], B[k][kk], acc[k], 0, 0, 0);
              }
          }
          const f16v s0 = ((acc[0] + acc[2]) + acc[4]) + acc[6];
          const f16v s1 = ((acc[1] + acc[3]) + acc[5]) + acc[7];
          const f16v r4 = (s0 + s1) + 0.0f;
#pragma unroll
          for (int i4 = 0; i4 < 4; i4++) {
            const int row_i = 16 * I + 4 * (lane >> 4) + i4, col_j = 16 * J + (lane & 15);
            const float raw = (r4[i4] + r4[4 + i4]) + (r4[8 + i4] + r4[12 + i4]);
            if (row_i < col_j && col_j < nc) D[row_i * nc + col_j] = raw;  // the raw sum; phase B applies the metric
          }
        }
    }
  } else
  for (int b = 0; NW * b < nc - 1; b++) {
    const int i = NW * b + ((b & 1) ? NW - 1 - wave : wave);
    if (i >= nc - 1) continue;  // wave-uniform
    float4 xq[NG];
    float xt = 0.0f;
    {
      const char *prow = i < nt ? tileL + (uint32_t)i * tpitch : slabL + (size_t)s_slot[i] * rowb;
      if (i < nt) {
#pragma unroll
        for (int g = 0; g < NG; g++) xq[g] = *reinterpret_cast<const float4 *>(tileL + (uint32_t)i * tpitch + g * 512);
        if constexpr (TAIL) xt = *reinterpret_cast<const float *>(tileL + (uint32_t)i * tpitch + NG * 512 - L * 12);
      } else {
#pragma unroll
        for (int g = 0; g < NG; g++) xq[g] = *reinterpret_cast<const float4 *>(prow + g * 512);
        if constexpr (TAIL) xt = *reinterpret_cast<const float *>(prow + NG * 512 - L * 12);
      }
    }
    if (nt == nc) row_pairs(std::true_type{}, i, xq, xt);
    else row_pairs(std::false_type{}, i, xq, xt);
  }
  __syncthreads();
  if (a.pair_tab) {
    // phase B runs in a kernel of its own (k_prune_select): one wave per node and no LDS, so that its 52 dependent
    // steps per node overlap across a few thousand nodes instead of holding a whole CU's LDS for one
    float *gD = a.pair_tab + (size_t)q * (kTileMaxCand * kTileMaxCand);
    for (int x = tid; x < nc * nc; x += NW * 64) gD[x] = D[x];
    if (tid < nc) a.pair_slots[(size_t)q * kTileMaxCand + tid] = s_slot[tid], a.pair_dists[(size_t)q * kTileMaxCand + tid] = s_dist[tid];
    if (tid == 0) a.prune_done[q] = 2u;
    if (wave == 0) stat_add(a, kStStagedRows, (unsigned long long)nt, lane);
    return;
  }
  if (wave != 0) return;
  // ---- phase B: the selection loop, candidate j = 64 c + lane in registers
  constexpr int NCH = kTileMaxCand / 64;
  float sd[NCH];
  uint32_t ss[NCH];
  bool rem[NCH];  // pruneRemoved (distset.go:124); lanes past the list count as removed
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int j = c * 64 + lane;
    sd[c] = j < nc ? s_dist[j] : 0.0f, ss[c] = j < nc ? s_slot[j] : kNoSlot, rem[c] = j >= nc;
  }
  uint32_t my_out = kNoSlot;
  float my_outd = 0.0f;
  int cnt = 0, i = 0;
  uint32_t n_eval = 0;
  while (true) {
    int found = -1;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const uint64_t m = __ballot(c * 64 + lane >= i && !rem[c] && ss[c] != self);  // :115-117
      if (found < 0 && m) found = c * 64 + __ffsll((unsigned long long)m) - 1;
    }
    if (found < 0) break;
    uint32_t p = 0;
    float pd = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; c++)
      if ((found >> 6) == c) p = rl(ss[c], found & 63), pd = rlf(sd[c], found & 63);
    if (lane == cnt) my_out = p, my_outd = pd;  // node.AddNeighbour :118
    cnt++;
    if (cnt >= (int)a.R) break;  // :119-121
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const int j = c * 64 + lane;
      const bool live = j > found && !rem[c];
      n_eval += (uint32_t)__popcll(__ballot(live));  // the pairs the reference's walk evaluates
      if (live && a.alpha * metric_finish(D[found * nc + j], a.metric) < sd[c]) rem[c] = true;  // :132
    }
    i = found + 1;
  }
  // ---- node.edges of the new node, kNoSlot padded; a freshly pruned row is clean and carries its distances
  a.adj[(size_t)self * kAdjStride + lane] = lane < cnt ? my_out : kNoSlot;
  a.adjdist[(size_t)self * kAdjStride + lane] = my_outd;
  if (lane == 0) {
    a.deg[self] = (uint32_t)cnt, a.clean[self] = (uint32_t)cnt, a.dcount[self] = (uint32_t)cnt;
    a.prune_done[q] = 1u;
    if (a.dirty) a.dirty[self] = 1;
  }
  if (a.keys_in)
    a.keys_in[(size_t)q * 64 + lane] =  // back-edge requests (insert.go:36)
        (lane < cnt) ? ((uint64_t)my_out << 32) | ((uint64_t)q << 6) | (uint64_t)lane : kNoKey;
  stat_add(a, kStPrunePairs, n_eval, lane);
  stat_add(a, kStStagedRows, (unsigned long long)nt, lane);
}


// Phase B of the tiled prune for every node whose pair table k_prune_new_tiled left in BuildArgs::pair_tab: the
// selection loop of robustPrune (search.go:113-137), one wave per node, the list in registers, one row of the table per
// selected candidate straight from global memory (written a moment ago: L2 / Infinity Cache).
__global__ __launch_bounds__(64) void k_prune_select(const BuildArgs a) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  if (a.prune_done[q] != 2u) return;
  const uint32_t self = a.first_slot + q;
  const int nc = (int)a.vis_count[q];
  const float *__restrict__ D = a.pair_tab + (size_t)q * (kTileMaxCand * kTileMaxCand);
  constexpr int NCH = kTileMaxCand / 64;
  float sd[NCH];
  uint32_t ss[NCH];
  bool rem[NCH];  // pruneRemoved (distset.go:124); lanes past the list count as removed
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int j = c * 64 + lane;
    sd[c] = j < nc ? a.pair_dists[(size_t)q * kTileMaxCand + j] : 0.0f;
    ss[c] = j < nc ? a.pair_slots[(size_t)q * kTileMaxCand + j] : kNoSlot;
    rem[c] = j >= nc;
  }
  uint32_t my_out = kNoSlot;
  float my_outd = 0.0f;
  int cnt = 0, i = 0;
  uint32_t n_eval = 0;
  while (true) {
    int found = -1;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const uint64_t m = __ballot(c * 64 + lane >= i && !rem[c] && ss[c] != self);  // :115-117
      if (found < 0 && m) found = c * 64 + __ffsll((unsigned long long)m) - 1;
    }
    if (found < 0) break;
    uint32_t p = 0;
    float pd = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; c++)
      if ((found >> 6) == c) p = rl(ss[c], found & 63), pd = rlf(sd[c], found & 63);
    if (lane == cnt) my_out = p, my_outd = pd;  // node.AddNeighbour :118
    cnt++;
    if (cnt >= (int)a.R) break;  // :119-121
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const int j = c * 64 + lane;
      const bool live = j > found && !rem[c];
      n_eval += live ? 1u : 0u;  // the pairs the reference's walk evaluates
      if (live && a.alpha * metric_finish(D[found * nc + j], a.metric) < sd[c]) rem[c] = true;  // :132
    }
    i = found + 1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n_eval += __shfl_xor(n_eval, o);
  // ---- node.edges of the new node, kNoSlot padded; a freshly pruned row is clean and carries its distances
  a.adj[(size_t)self * kAdjStride + lane] = lane < cnt ? my_out : kNoSlot;
  a.adjdist[(size_t)self * kAdjStride + lane] = my_outd;
  if (lane == 0) {
    a.deg[self] = (uint32_t)cnt, a.clean[self] = (uint32_t)cnt, a.dcount[self] = (uint32_t)cnt;
    a.prune_done[q] = 1u;
    if (a.dirty) a.dirty[self] = 1;
  }
  a.keys_in[(size_t)q * 64 + lane] =  // back-edge requests (insert.go:36)
      (lane < cnt) ? ((uint64_t)my_out << 32) | ((uint64_t)q << 6) | (uint64_t)lane : kNoKey;
  stat_add(a, kStPrunePairs, n_eval, lane);
}

}  // namespace sdb

#include "bigprune.inc"

namespace sdb {

// One wavefront per sorted key; only the first key of each target B proceeds and applies B's requests
// (insert.go:36-65).  Requests are taken in insert order, as many at a time as fit the candidate buffer
// (kBackCap - degree): if they all fit under the degree bound they are appended (:62), otherwise B is
// re-pruned once over its neighbours plus those new points (:47-58: candidateSet.Add(neighbours...),
// Add(A), Sort, robustPrune).  With one request per target -- always the case for round_size = 1 -- this is
// exactly the reference's per-edge rule; with several it is the same rule applied to the group, which
// spares a hub node one full re-prune per incoming edge.
constexpr uint32_t kBackCap = 256;
// amdgpu_waves_per_eu(3): the kernel came out at 169 VGPRs, one over the limit for three waves per SIMD; held to 168 it
// runs 12 waves per CU instead of 8 and 5 - 13 % faster (it lives on rows in flight); four waves (128 VGPRs) spill: 1.3 x slower.

#ifndef SDB_BACK_WAVES
#define SDB_BACK_WAVES 3
#endif
// Rows of 1 536 floats and more (NG >= 12: the query row alone is 48+ registers, a pair of candidate rows as many again)
// do not fit that cap: NG = 12 spilled 20 registers, NG = 24 more than 180 (404 bytes of scratch per lane).  They run two
// waves per SIMD with 256 registers each -- since round 5 also rows of 768 and 1 024 floats and the run-time row length
// (NG = 6 with its pair records; NG = 8, -1: 7 and 10 registers spilled at three waves once the deferral of full
// re-prunes had joined the kernel).  Rows of 3 072 floats (NG = 24: the query row is 96 registers) take the whole file --
// one wave per SIMD -- since round 6: at 256 registers 5 of them went to scratch.
template <int NG, bool L2>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NG == 24 ? 1 : (NG >= 6 || NG == -1) ? 2 : SDB_BACK_WAVES))) void k_backedges(const BuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  PruneLds l(lds_raw, kBackCap, NG >= 0);
  const int lane = threadIdx.x, L = lane & 31;
  const size_t pos = blockIdx.x;
  const size_t total = (size_t)a.nnew * 64;
#ifdef SDB_BACK_PROFILE
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  const uint64_t key = a.keys_sorted[pos];
  if (key == kNoKey) return;
  const uint32_t b = (uint32_t)(key >> 32);
  if (pos > 0 && (uint32_t)(a.keys_sorted[pos - 1] >> 32) == b) return;  // not the head of B's segment
#ifdef SDB_BACK_U
  constexpr int U = NG == 3 ? SDB_BACK_U : (NG >= 0 ? ChunkPairs<NG, false>::value : 4);
#else
  constexpr int U = NG >= 0 ? ChunkPairs<NG, false>::value : 4;
#endif
  // segment length m: requests for B, already in insert order
  size_t m = 1;
  while (pos + m < total) {
    const uint64_t kk = a.keys_sorted[pos + m];
    if (kk == kNoKey || (uint32_t)(kk >> 32) != b) break;
    m++;
  }
  if (a.big_count && (m >= a.big_min || (a.start_ext_n && b == a.start_slot))) {  // a hub: the whole chip prunes it afterwards (bigprune.inc)
    if (lane == 0) {
      const uint32_t at = atomicAdd(a.big_count, 1u);
      if (at < a.big_cap) a.big_list[at] = make_uint4((uint32_t)pos, (uint32_t)m, b, a.deg[b]);
    }
    stat_add(a, kStRequests, m, lane);
    stat_add(a, kStHubs, 1, lane);
    return;
  }
  uint32_t row = a.adj[(size_t)b * kAdjStride + lane];
  float rowd = a.adjdist[(size_t)b * kAdjStride + lane];  // cached distFn(B, edge), valid for lanes < dc
  uint32_t deg = a.deg[b];
  uint32_t ncl0 = 0;  // the clean prefix the appends below stand behind (rows that keep pair records only)
  if constexpr (kPairRecords<NG>) ncl0 = a.clean[b] < deg ? a.clean[b] : deg;
  uint32_t dc = a.dcount[b] < deg ? a.dcount[b] : deg;
  bool row_dirty = false;
  auto req_slot = [&](size_t r) { return a.first_slot + (uint32_t)((a.keys_sorted[pos + r] & 0xFFFFFFFFull) >> 6); };
  // distFn(B, A) for request r: A's own row holds distFn(A, B) at the edge position the key carries, and the
  // arithmetic is bitwise symmetric in its two arguments -- valid when A's row has its distances cached
  auto req_dist = [&](size_t r, bool *valid) {
    const uint32_t anew = req_slot(r);
    const uint32_t epos = (uint32_t)(a.keys_sorted[pos + r] & 63ull);
    *valid = a.dcount[anew] > epos;
    return a.adjdist[(size_t)anew * kAdjStride + epos];
  };
  size_t done = 0;
  uint32_t st_eval = 0, st_cached = 0, st_reprune = 0, st_append = 0;
  while (done < m) {
    size_t t = m - done;
    if (t > kBackCap - deg) t = kBackCap - deg;
    if (deg + t <= a.R) {  // insert.go:62 nodeB.AddNeighbour(vecA), t times
      for (size_t r = 0; r < t; r++) {
        const uint32_t anew = req_slot(done + r);
        bool have;
        const float d = req_dist(done + r, &have);
        if (kPairRecords<NG> && a.pairc && a.dcache) {  // the new edge's distances to the edges in front of it, out of its own search's table
          const int rec = (int)deg - (int)ncl0;
          if (rec >= 0 && rec < kMaxDirty && lane < (int)deg) {
            const uint2 e = (a.dcache + ((size_t)(anew - a.first_slot) << (32 - a.dcache_shift)))[(row * 2654435761u) >> a.dcache_shift];
            a.pairc[((size_t)b * kMaxDirty + rec) * 64 + lane] = __uint_as_float(e.x == row ? e.y : 0xFFFFFFFFu);
          }
        }
        if (lane == (int)deg) row = anew, rowd = d;
        if (dc == deg && have) dc = deg + 1;  // the cached prefix grows only without a gap
        deg++;
      }
      row_dirty = true;
      done += t;
      st_append += (uint32_t)t;
      continue;
    }
    // B overflows: distances from B (distFn = DistanceFromPoint(nB) :49) to its neighbours and the t new points
    const int nc = (int)deg + (int)t;
    if constexpr (NG == kQuantized) {
      __syncthreads();
      for (int c = lane; c < nc; c += 64)
        l.in_slot[c] = c >= (int)deg ? req_slot(done + (size_t)(c - (int)deg)) : kNoSlot;
      if (lane < (int)deg) l.in_slot[lane] = row;
      __syncthreads();
      const uint8_t *cb = a.pq_codes + (size_t)b * a.pq_M;
      for (int c = lane; c < nc; c += 64) l.in_dist[c] = pq_sym_dist(a, cb, a.pq_codes + (size_t)l.in_slot[c] * a.pq_M);
    } else {
      // candidate c: c < deg -> row entry c (edge order), then the new points in insert order (:55-56; Add
      // dedupes, and a new node can not already be a neighbour).  Cached distances are taken as they are;
      // the indices of the others are collected in s_slot (free until the prune sorts into it).
      __syncthreads();
      int nmiss = 0;
      for (int base = 0; base < nc; base += 64) {
        const int c = base + lane;
        bool have = false;
        if (c < (int)deg) {
          l.in_slot[c] = row, l.in_dist[c] = rowd, have = c < (int)dc;
        } else if (c < nc) {
          l.in_slot[c] = req_slot(done + (size_t)(c - (int)deg));
          l.in_dist[c] = req_dist(done + (size_t)(c - (int)deg), &have);
        }
        const uint64_t mm = __ballot(c < nc && !have);
        if (c < nc && !have) l.s_slot[nmiss + __popcll(mm & ((1ull << lane) - 1))] = (uint32_t)c;
        nmiss += __popcll(mm);
      }
      __syncthreads();
      st_eval += (uint32_t)nmiss, st_cached += (uint32_t)(nc - nmiss);
      if (nmiss) {
        PointRow<NG> pr;
        const float *brow = a.slab + (size_t)b * a.ld;
        if constexpr (NG >= 0) {
#pragma unroll
          for (int g = 0; g < NG; g++) pr.xq[g] = reinterpret_cast<const float4 *>(brow)[g * 32 + L];
          if (NG == 0) pr.xq[0] = make_float4(0.f, 0.f, 0.f, 0.f);
          pr.xt = a.tail ? brow[NG * 128 + L] : 0.0f;
        } else {
          for (uint32_t x = lane; x < a.ld; x += 64) l.qs[x] = brow[x];
          __syncthreads();
        }
        for (int m0 = 0; m0 < nmiss; m0 += 2 * U) {
          uint32_t slot[U];
          float res[U];
          int cidx[U];  // this half-wave's candidate of each pair
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int at = m0 + 2 * u + (lane >> 5);
            cidx[u] = (int)l.s_slot[at < nmiss ? at : nmiss - 1];
            slot[u] = l.in_slot[cidx[u]];
          }
          if constexpr (NG >= 0) chunk_dist<NG, L2, U>(a.slab, a.ld, a.tail, pr.xq, pr.xt, slot, res, lane);
          else chunk_dist_lds<L2, U>(a.slab, a.ld, a.ng, a.tail, l.qs, slot, res, lane);
          if (L == 0) {
#pragma unroll
            for (int u = 0; u < U; u++) l.in_dist[cidx[u]] = metric_finish(res[u], a.metric);
          }
        }
      }
    }
    __syncthreads();
    // the first clean[b] candidates are B's edges as its last prune left them (edges appended since and the
    // new points are not)
    uint32_t ncl;
    if constexpr (kPairRecords<NG>) ncl = ncl0 < deg ? ncl0 : deg;  // (= clean[b]: kept up to date across this wave's own prunes)
    else ncl = a.clean[b] < deg ? a.clean[b] : deg;
    if constexpr (NG >= 1) {
      // (kTileMaxCand is declared with the tiled kernel above: 128 candidates)
      const bool few_new = nc <= kPairMax && ncl > 0 && nc - (int)ncl <= kMaxDirty;  // robust_prune_wave's sparse mode
      if (a.def_count && !few_new && nc <= kTileMaxCand && done + t == m) {
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(a.def_count, 1u);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (at < a.def_cap) {
          for (int c = lane; c < nc; c += 64) {
            a.def_slots[(size_t)at * kTileMaxCand + c] = l.in_slot[c];
            a.def_dists[(size_t)at * kTileMaxCand + c] = l.in_dist[c];
          }
          if (lane == 0) a.def_self[at] = b, a.def_nc[at] = (uint32_t)nc;
          st_reprune++;
          row_dirty = false;  // the tiled prune writes the row
          done += t;
          break;
        }
      }
    }
#ifdef SDB_BACK_COUNT  // measurement builds: how the re-prunes split into sparse ones and full ones, and the rows each kind reads
    const uint32_t ev0 = st_eval;
    const bool sparse_kind = NG >= 0 && nc <= kPairMax && ncl > 0 && nc - (int)ncl <= kMaxDirty;
#endif
    robust_prune_wave<NG, L2>(a, b, nc, l.in_slot, l.in_dist, l.s_slot, l.s_dist, l.s_rem, l.qs, lane,
                              (NG >= 0 && nc <= kPairMax) ? l.D : nullptr, (int)ncl, true, &st_eval,
                              &st_cached, a.pairc ? a.pairc + (size_t)b * kMaxDirty * 64 : nullptr, (int)deg);  // :57-58
#ifdef SDB_BACK_COUNT
    stat_add(a, sparse_kind ? 13 : 11, 1, lane);
    stat_add(a, sparse_kind ? 14 : 12, st_eval - ev0, lane);
    if (!sparse_kind && ncl == 0) stat_add(a, 15, 1, lane);  // ... of which: rows that had never been pruned
#endif
    st_reprune++;
    __syncthreads();
    row = a.adj[(size_t)b * kAdjStride + lane];  // written by this lane just above
    rowd = a.adjdist[(size_t)b * kAdjStride + lane];
    deg = (uint32_t)__popcll(__ballot(row != kNoSlot));
    dc = deg;  // every edge of a freshly pruned row carries its distance
    if constexpr (kPairRecords<NG>) ncl0 = deg;  // ... and the whole row is clean: later appends of this wave stand behind it
    row_dirty = false;
    done += t;
  }
  if (row_dirty) {
    a.adj[(size_t)b * kAdjStride + lane] = row;
    a.adjdist[(size_t)b * kAdjStride + lane] = rowd;
    if (lane == 0) {
      a.deg[b] = deg, a.dcount[b] = dc;
      if (a.dirty) a.dirty[b] = 1;
    }
  }
#ifdef SDB_BACK_PROFILE  // measurement builds (tools/backprof.py): the wave's cycles by requests per target, spare stat slots
  {
    const unsigned long long cyc = __builtin_readcyclecounter() - t_begin;
    stat_add(a, m == 1 ? 11 : (m <= 4 ? 12 : (m <= 16 ? 13 : 14)), cyc, lane);
    stat_add(a, 15, cyc > 200000ull ? cyc : 0ull, lane);
  }
#endif
  stat_add(a, kStBackPairs, st_eval, lane);
  stat_add(a, kStBackCached, st_cached, lane);
  stat_add(a, kStRequests, m, lane);
  stat_add(a, kStReprunes, st_reprune, lane);
  stat_add(a, kStAppends, st_append, lane);
}

template <int NG, bool L2>
static int launch_round(const BuildArgs &a, hipStream_t stream, void *sort_tmp, size_t sort_tmp_bytes,
                        int sort_end_bit, BigScratch *big, bool *start_pruned) {
  BuildArgs a1 = a;
  a1.prune_done = nullptr;
  bool tiled_ok = false;  // the LDS-tiled prune serves this row shape (and is not switched off by the test knob)
  if constexpr (NG >= 1) {
    // candidate rows staged in LDS when a useful share of a visit list fits beside its pair table
    // (always at d <= 1024 and searchSize 75); what the tiled kernel leaves is pruned by k_prune_new
    const uint32_t worst = kTileFixedBytes + 80 * 80 * 4 + 1024;
    if (a.no_tile != 1 && a.prune_done && worst + 24 * a.ld * 4 <= kTileLdsBytes) {
      static std::atomic<uint64_t> attr_set{0};  // per instantiation, a bit per device
      auto setattr = [](const void *f) {
        return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileLdsBytes);
      };
      if (first_use_on_this_device(attr_set)) {
        SDB_HIP(setattr(reinterpret_cast<const void *>(&k_prune_new_tiled<NG, L2, false, 4>)));
        SDB_HIP(setattr(reinterpret_cast<const void *>(&k_prune_new_tiled<NG, L2, true, 4>)));
        SDB_HIP(setattr(reinterpret_cast<const void *>(&k_prune_new_tiled<NG, L2, false, 8>)));
      }
      // 8 waves (two per SIMD) for the common no-tail rows: one wave's LDS waits run under the other's arithmetic
      // (1M x 384 build: 2.25 s with the one-wave kernel, 2.10 s tiled with 4 waves, 1.98 s with 8)
      if (a.tail) hipLaunchKernelGGL((k_prune_new_tiled<NG, L2, true, 4>), dim3(a.nnew), dim3(256), kTileLdsBytes, stream, a);
      else if (a.no_tile == 2) hipLaunchKernelGGL((k_prune_new_tiled<NG, L2, false, 4>), dim3(a.nnew), dim3(256), kTileLdsBytes, stream, a);
      else hipLaunchKernelGGL((k_prune_new_tiled<NG, L2, false, 8>), dim3(a.nnew), dim3(512), kTileLdsBytes, stream, a);
      SDB_HIP(hipGetLastError());
      if (a.pair_tab) {
        hipLaunchKernelGGL(k_prune_select, dim3(a.nnew), dim3(64), 0, stream, a);
        SDB_HIP(hipGetLastError());
      }
      a1.prune_done = a.prune_done;
      tiled_ok = true;
    }
  }
  {
    const size_t lds1 = prune_lds_bytes(a.vis_cap, NG, a.ld);
    hipLaunchKernelGGL((k_prune_new<NG, L2>), dim3(a.nnew), dim3(64), lds1, stream, a1);
  }
  SDB_HIP(hipGetLastError());
  size_t tmp = sort_tmp_bytes;
  SDB_HIP(hipcub::DeviceRadixSort::SortKeys(sort_tmp, tmp, a.keys_in, a.keys_sorted, (int)((size_t)a.nnew * 64), 0,
                                            sort_end_bit, stream));
  const size_t lds2 = prune_lds_bytes(kBackCap, NG, a.ld, NG >= 0);
  if (a.big_count) SDB_HIP(hipMemsetAsync(a.big_count, 0, 4, stream));  // word 1 = BuildArgs::flags, kept
  BuildArgs ab = a;  // k_backedges' view: deferral only where the tiled prune can take what is deferred
  if (!tiled_ok || a.no_tile) ab.def_count = nullptr;
  if (ab.def_count) {
    // (one workgroup per list in the launch behind k_backedges, most of them empty -- 94 k of a 1M build's 22.7 M
    // re-prunes are deferred -- but a list that finds no room is pruned in place by a wave that then reads 1 600 rows
    // and holds its launch up: a grid of nnew / 2 cost the build 4 %, so the room is generous)
    ab.def_cap = std::min<uint32_t>(a.def_cap, 2 * a.nnew + 64);
    SDB_HIP(hipMemsetAsync(ab.def_count, 0, 4, stream));
    SDB_HIP(hipMemsetAsync(ab.def_nc, 0, (size_t)ab.def_cap * 4, stream));
  }
  hipLaunchKernelGGL((k_backedges<NG, L2>), dim3(a.nnew * 64), dim3(64), lds2, stream, ab);
  SDB_HIP(hipGetLastError());
  if constexpr (NG >= 1) {
    if (ab.def_count) {  // the re-prunes k_backedges deferred: candidate rows staged in LDS once, selection in the kernel
      BuildArgs d = ab;
      d.self_list = a.def_self, d.vis_slots = a.def_slots, d.vis_dists = a.def_dists, d.vis_count = a.def_nc;
      d.vis_cap = kTileMaxCand, d.nnew = ab.def_cap, d.prune_done = a.def_done;
      d.pair_tab = nullptr, d.pair_slots = nullptr, d.pair_dists = nullptr, d.keys_in = nullptr, d.def_count = nullptr;
      if (a.tail) hipLaunchKernelGGL((k_prune_new_tiled<NG, L2, true, 4>), dim3(d.nnew), dim3(256), kTileLdsBytes, stream, d);
      else hipLaunchKernelGGL((k_prune_new_tiled<NG, L2, false, 8>), dim3(d.nnew), dim3(512), kTileLdsBytes, stream, d);
      SDB_HIP(hipGetLastError());
    }
  }
  // the hubs of this round, if any (bigprune.inc).  A target gets at most one request per new node, so a round of
  // fewer than big_min points cannot have one and the host need not wait for the count (the early rounds -- two
  // thirds of all rounds of a 1M build -- then run without a host round trip each)
  if (a.big_count && (a.nnew >= a.big_min || a.start_ext_n)) {
    uint32_t nbig = 0;
    SDB_HIP(hipMemcpyAsync(&nbig, a.big_count, 4, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipStreamSynchronize(stream));
    if (nbig > a.big_cap) return fail(SDB_ERR_DEVICE, "hub list overflow (%u > %u)", nbig, a.big_cap);
    if (nbig) {
      std::vector<uint4> list(nbig);
      SDB_HIP(hipMemcpy(list.data(), a.big_list, (size_t)nbig * sizeof(uint4), hipMemcpyDeviceToHost));
      std::sort(list.begin(), list.end(), [](const uint4 &x, const uint4 &y) { return x.x < y.x; });
      for (const uint4 &e : list) {
        const size_t pos = e.x;
        const uint32_t deg = e.w;
        const uint32_t ext = e.z == a.start_slot ? a.start_ext_n : 0u;
        SDB_TRY((big_prune<NG, L2>(a, e.z, deg + ext + e.y, *big, stream, [&](const BigArgs &g, unsigned tb) {
          hipLaunchKernelGGL(k_big_fill_round, dim3(tb), dim3(256), 0, stream, g, pos, deg, ext);
        })));
        if (ext) *start_pruned = true;  // robustPrune leaves at most DegreeBound edges: the overflow list is gone
      }
    }
  }
  return SDB_OK;
}

template <bool L2>
static int launch_round_ng(const BuildArgs &a, hipStream_t s, void *t, size_t tb, int eb, BigScratch *big, bool *sp) {
  switch (a.ng) {
    case 0: return launch_round<0, L2>(a, s, t, tb, eb, big, sp);
    case 1: return launch_round<1, L2>(a, s, t, tb, eb, big, sp);
    case 2: return launch_round<2, L2>(a, s, t, tb, eb, big, sp);
    case 3: return launch_round<3, L2>(a, s, t, tb, eb, big, sp);
    case 4: return launch_round<4, L2>(a, s, t, tb, eb, big, sp);
    case 6: return launch_round<6, L2>(a, s, t, tb, eb, big, sp);
    case 8: return launch_round<8, L2>(a, s, t, tb, eb, big, sp);
    case 12: return launch_round<12, L2>(a, s, t, tb, eb, big, sp);  // 1536
    case 16: return launch_round<16, L2>(a, s, t, tb, eb, big, sp);  // 2048
    case 24: return launch_round<24, L2>(a, s, t, tb, eb, big, sp);  // 3072
    default: return launch_round<-1, L2>(a, s, t, tb, eb, big, sp);
  }
}

__global__ void k_fill_u64(uint64_t *p, uint64_t v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace sdb

using namespace sdb;

// defined in index.hip
extern "C" int sdb_index_stats(const sdb_index *ix, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *max_node_id);

namespace sdb {
int store_rows_public(sdb_index *ix, uint32_t first, uint32_t n, const float *dev_vectors, hipStream_t stream);
}

extern "C" int sdb_index_insert_batch(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors, int mem,
                                      uint32_t round_size, void *stream_) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (n == 0) return SDB_OK;
  if (!vectors) return fail(SDB_ERR_INVALID, "vectors is NULL");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  if (ix->start_slot < 0) return fail(SDB_ERR_STATE, "failed to get start point");  // search.go:57-60
  if ((uint64_t)ix->n + n >= 0x7FFFFFFFull) return fail(SDB_ERR_INVALID, "too many nodes");
  // ---- ids: vamana.go:150-157 rejects 0 and the start id; an existing id would be an update
  std::vector<uint64_t> new_ids(n);
  for (uint64_t i = 0; i < n; i++) {
    uint64_t id = ids ? ids[i] : std::max<uint64_t>(ix->max_node_id, SDB_STARTID) + 1 + i;
    if (id == SDB_STARTID) return fail(SDB_ERR_INVALID, "cannot modify point with start id: %llu", SDB_STARTID);
    if (id == 0) return fail(SDB_ERR_INVALID, "invalid point id: 0");
    new_ids[i] = id;
  }
  {
    // duplicates inside the batch or against the index
    std::vector<uint64_t> sorted(new_ids);
    std::sort(sorted.begin(), sorted.end());
    for (uint64_t i = 1; i < n; i++)
      if (sorted[i] == sorted[i - 1]) return fail(SDB_ERR_EXISTS, "duplicate id %llu in batch", (unsigned long long)sorted[i]);
    for (uint64_t i = 0; i < n; i++)
      if (ix->slot_of(new_ids[i]) >= 0)
        return fail(SDB_ERR_EXISTS, "point %llu exists: updates are not on the device path", (unsigned long long)new_ids[i]);
  }
  DeviceGuard dg(ix->P.device);
  hipStream_t stream = as_stream(stream_);
  const RowLayout &l = ix->lay;
  const uint32_t n0 = ix->n;
  const sdb_pq *pq = ix->pq;
  struct Cleanup {
    std::vector<void *> ptrs;
    hipStream_t s;
    ~Cleanup() {
      (void)hipStreamSynchronize(s);
      for (void *p : ptrs)
        if (p) (void)hipFree(p);
    }
  } cleanup{{}, stream};
  // ---- every allocation of the call comes first: an out-of-memory failure must leave the index as it was
  SDB_TRY(ix->reserve(n0 + (uint32_t)n));
  // ... the host tables included: the ids of the new points go into the id -> slot map NOW, at the slots they will
  // have (slots past ix->n resolve to nothing, sdb_index::slot_of), so that the per-round bookkeeping (`commit` below)
  // only moves counters and cannot run out of memory with rounds already applied
  uint64_t done = 0;  // points whose rounds have been applied
  struct PreMapped {
    sdb_index *ix;
    const std::vector<uint64_t> *ids;
    const uint64_t *done;
    bool active = false, was_dense = false;
    ~PreMapped() {  // whatever was not applied leaves the map again (erase does not allocate)
      if (!active || *done == ids->size()) return;
      std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);
      if (was_dense && *done == 0) ix->id2slot.clear();
      else
        for (uint64_t i = *done; i < ids->size(); i++) ix->id2slot.erase((*ids)[i]);
    }
  } premap{ix, &new_ids, &done};
  {
    std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);  // a rehash moves what the searches' id lookups read
    ix->h_ids.reserve((size_t)n0 + n);
    bool stays_dense = ix->dense_ids && ix->h_ids.size() > 0;
    for (uint64_t i = 0; i < n && stays_dense; i++) stays_dense = new_ids[i] == ix->h_ids[0] + ix->h_ids.size() + i;
    if (!stays_dense) {
      premap.was_dense = ix->dense_ids;
      premap.active = true;  // from here on a failure erases what got in
      if (ix->dense_ids) ix->id2slot.clear();  // (a dense table ignores the map: filling it changes no answer)
      ix->id2slot.reserve(((size_t)n0 + n) * 2);
      if (ix->dense_ids)
        for (size_t sl = 0; sl < ix->h_ids.size(); sl++)
          if (ix->h_ids[sl] != 0) ix->id2slot.emplace(ix->h_ids[sl], (uint32_t)sl);
      for (uint64_t i = 0; i < n; i++) ix->id2slot.emplace(new_ids[i], n0 + (uint32_t)i);
    }
  }
  static_assert(SDB_BUILD_STATS <= sdb_index::kStatStride, "stat slots per copy");
  const size_t bstats_bytes = (size_t)sdb_index::kStatCopies * sdb_index::kStatStride * sizeof(uint64_t);
  if (!ix->d_bstats) SDB_HIP(hipMalloc(&ix->d_bstats, bstats_bytes));
  float *staging = nullptr;
  if (mem == SDB_MEM_HOST) {
    SDB_HIP(hipMalloc(&staging, n * l.dim * sizeof(float)));
    cleanup.ptrs.push_back(staging);
  }
  // per-round buffers, sized for the largest round
  const uint32_t L = ix->P.search_size;
  const uint32_t vis_cap = std::max<uint32_t>(1024, 4 * L);
  uint32_t max_round = round_size ? round_size : 16384;
  if (max_round > n) max_round = (uint32_t)n;
  const size_t lut_row = pq ? (size_t)pq->M * pq->K * sizeof(float) : 0;
  if (pq && (size_t)max_round * lut_row > ((size_t)1 << 30)) max_round = (uint32_t)std::max<size_t>(1, ((size_t)1 << 30) / lut_row);
  const uint32_t total_rows = n0 + (uint32_t)n;
  const uint32_t words = ((total_rows + 31) / 32 + 31) & ~31u;
  uint32_t *bitsets = nullptr, *vis_slots = nullptr, *vis_count = nullptr;
  float *vis_dists = nullptr;
  uint64_t *keys_in = nullptr, *keys_sorted = nullptr;
  void *sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int end_bit = 64;
  {
    int b = 32;
    while (b < 64 && (total_rows >> (b - 32)) != 0) b++;
    end_bit = b;
  }
  SDB_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                            (int)((size_t)max_round * 64), 0, end_bit, stream));
  SDB_HIP(hipMalloc(&bitsets, (size_t)max_round * words * 4));
  cleanup.ptrs.push_back(bitsets);
  SDB_HIP(hipMalloc(&vis_slots, (size_t)max_round * vis_cap * 4));
  cleanup.ptrs.push_back(vis_slots);
  SDB_HIP(hipMalloc(&vis_dists, (size_t)max_round * vis_cap * 4));
  cleanup.ptrs.push_back(vis_dists);
  SDB_HIP(hipMalloc(&vis_count, (size_t)max_round * 4));
  cleanup.ptrs.push_back(vis_count);
  SDB_HIP(hipMalloc(&keys_in, (size_t)max_round * 64 * 8));
  cleanup.ptrs.push_back(keys_in);
  SDB_HIP(hipMalloc(&keys_sorted, (size_t)max_round * 64 * 8));
  cleanup.ptrs.push_back(keys_sorted);
  SDB_HIP(hipMalloc(&sort_tmp, sort_tmp_bytes ? sort_tmp_bytes : 16));
  cleanup.ptrs.push_back(sort_tmp);
  float *lut = nullptr;
  if (pq) {
    SDB_HIP(hipMalloc(&lut, (size_t)max_round * lut_row));
    cleanup.ptrs.push_back(lut);
  }
  // hubs: targets with this many requests in one round go to the chip-wide prune (bigprune.inc)
  const uint32_t big_min = ix->tune_hub_min;  // 512 unless a test set it (sdb_index_set_tuning)
  const uint32_t big_cap = (uint32_t)((size_t)max_round * 64 / big_min + 2);
  uint32_t *big_count = nullptr;  // [0] hubs listed this round, [1] flags (BuildArgs::flags)
  uint4 *big_list = nullptr;
  SDB_HIP(hipMalloc(&big_count, 8));
  cleanup.ptrs.push_back(big_count);
  SDB_HIP(hipMalloc(&big_list, (size_t)big_cap * sizeof(uint4)));
  cleanup.ptrs.push_back(big_list);
  uint32_t *prune_done = nullptr;
  SDB_HIP(hipMalloc(&prune_done, (size_t)max_round * 4));
  cleanup.ptrs.push_back(prune_done);
  // the tiled prune's pair tables, handed from its all-pairs phase to the selection kernel (64 KB per point of a round)
  float *pair_tab = nullptr, *pair_dists = nullptr;
  uint32_t *pair_slots = nullptr;
  if (!pq && ix->tune_no_tile != 3) {
    SDB_HIP(hipMalloc(&pair_tab, (size_t)max_round * kTileMaxCand * kTileMaxCand * 4));
    cleanup.ptrs.push_back(pair_tab);
    SDB_HIP(hipMalloc(&pair_slots, (size_t)max_round * kTileMaxCand * 4));
    cleanup.ptrs.push_back(pair_slots);
    SDB_HIP(hipMalloc(&pair_dists, (size_t)max_round * kTileMaxCand * 4));
    cleanup.ptrs.push_back(pair_dists);
  }
  // back-edge targets whose re-prune is left to the LDS-tiled kernel (BuildArgs::def_*): full-precision store only
  uint32_t *def_words = nullptr, *def_slots = nullptr;
  float *def_dists = nullptr;
  const uint32_t def_cap = 2 * max_round + 64;
  if (!pq) {
    SDB_HIP(hipMalloc(&def_words, ((size_t)3 * def_cap + 4) * 4));  // [count, pad x3][self][nc][done]
    cleanup.ptrs.push_back(def_words);
    SDB_HIP(hipMalloc(&def_slots, (size_t)def_cap * kTileMaxCand * 4));
    cleanup.ptrs.push_back(def_slots);
    SDB_HIP(hipMalloc(&def_dists, (size_t)def_cap * kTileMaxCand * 4));
    cleanup.ptrs.push_back(def_dists);
  }
  BigScratch big_scratch;
  // per new point: the (slot, distance) pairs its search evaluates, direct-mapped (SearchArgs::dcache)
  constexpr uint32_t kDcacheBits = SDB_DCACHE_BITS;  // 8 192 entries = 64 KB per point: ~4 000 evaluations, ~80 % survive
  uint2 *dcache = nullptr;
  if (!pq) {
    SDB_HIP(hipMalloc(&dcache, ((size_t)max_round << kDcacheBits) * sizeof(uint2)));
    cleanup.ptrs.push_back(dcache);
  }
  // pair distances of appended edges (BuildArgs::pairc): 4 KB per row for the length of this call -- bulk builds only
  // (a call that adds at least a quarter to the table).  A CACHE: it is asked for LAST, after every buffer the call
  // cannot do without, and only when it leaves the device room for what is allocated later -- the chip-wide prune's
  // scratch and its radix sort (bigprune.inc, mid-build: a failure there leaves the handle unusable), the workspaces
  // and quantizer tables of searches that run meanwhile: 4 GB or a sixteenth of the device, whichever is more (at
  // 10M rows the cache is 41 GB; on a device near capacity it used to be taken first and starve the rest).
  float *pairc = nullptr;
  if (!pq && l.ng <= 6 && n >= 256 && n >= (uint64_t)n0 / 4 && !ix->tune_no_defer) {
    const size_t bytes = (size_t)total_rows * kMaxDirty * 64 * sizeof(float);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = total_b = 0, (void)hipGetLastError();
    const size_t headroom = std::max<size_t>((size_t)4 << 30, total_b / 16);
    if (free_b >= bytes + headroom && hipMalloc(&pairc, bytes) == hipSuccess) {
      cleanup.ptrs.push_back(pairc);
      if (hipMemsetAsync(pairc, 0xFF, bytes, stream) != hipSuccess) pairc = nullptr;
    } else {
      (void)hipGetLastError();
      pairc = nullptr;
    }
  }

  // ---- from here on the call writes, into the writer's copy of the graph (index.h graph versions): searches
  // issued meanwhile keep walking the last committed version.  Rows beyond ix->n first.
  SDB_TRY(ix->begin_write());
  // Every exit below this line goes through one of two doors.  write_failed: nothing of this call has reached the
  // graph or the host tables yet when done == 0 (rows past ix->n are invisible) -- the transaction this call opened
  // for itself closes again and the index is as it was; with rounds already applied the rows they appended and the
  // back-edges they wrote have no committed counterpart and no rollback, so the handle becomes unusable (index.h
  // `broken`), never a half-open transaction that wedges the next writer.  round_failed: the round's kernels had
  // started, which is the same as "applied".
  auto round_failed = [&](int rc) {
    ix->broken = true;
    ix->tx_dirty = true;
    return rc;
  };
  auto write_failed = [&](int rc) {
    if (done > 0) return round_failed(rc);
    if (!ix->tx_explicit && !ix->tx_dirty) ix->in_tx = false;
    return rc;
  };
#define SDB_W_TRY(expr)                          \
  do {                                           \
    int _rc = (expr);                            \
    if (_rc != SDB_OK) return write_failed(_rc); \
  } while (0)
#define SDB_W_HIP(expr)                                                                                         \
  do {                                                                                                          \
    hipError_t _e = (expr);                                                                                     \
    if (_e != hipSuccess)                                                                                       \
      return write_failed(sdb::fail(SDB_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                                    __LINE__));                                                                 \
  } while (0)
  bool in_round = false;  // a round's prune / back-edge kernels may have started
  auto run = [&]() -> int {
  SDB_W_HIP(hipMemsetAsync(ix->d_bstats, 0, bstats_bytes, stream));
  SDB_W_HIP(hipMemsetAsync(big_count, 0, 8, stream));
  const float *dvec = vectors;  // the vectors (original layout) on device; they double as the search queries
  if (staging) {
    SDB_W_HIP(hipMemcpyAsync(staging, vectors, n * l.dim * sizeof(float), hipMemcpyHostToDevice, stream));
    dvec = staging;
  }
  // vecStore.Set for the whole batch (insert.go:17): rows are unreachable until they get in-edges
  SDB_W_TRY(store_rows_public(ix, n0, (uint32_t)n, dvec, stream));
  // a fitted quantizer encodes on Set (product.go:161-169); from here on every distance of the insert is a
  // table distance: LUT for the search (DistanceFromFloat), centroid pairs for the prunes (DistanceFromPoint)
  if (pq) SDB_W_TRY(pq_encode_device(pq, dvec, n, ix->d_codes + (size_t)n0 * pq->M, stream));
  SDB_W_HIP(hipMemcpyAsync(ix->d_ids + n0, new_ids.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
  // host-side id bookkeeping of the points of one completed round: h_ids / id2slot / max_node_id move together
  // with ix->n, so that an error return never leaves ids that resolve to slots past the rows in use
  auto commit = [&](uint64_t from, uint64_t to) {
    std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);  // searches translate filter ids with these tables
    ix->tx_dirty = true;
    bool dense = ix->dense_ids;
    for (uint64_t i = from; i < to; i++) {
      if (dense && new_ids[i] != ix->h_ids[0] + ix->h_ids.size()) dense = false;  // the map has been filled (above)
      ix->h_ids.push_back(new_ids[i]);                                           // (room reserved above)
      if (new_ids[i] > ix->max_node_id) ix->max_node_id = new_ids[i];  // vamana.go:166-168
    }
    ix->dense_ids = dense;
    ix->n = n0 + (uint32_t)to;
  };
  auto check_flags = [&]() -> int {
    uint32_t fl = 0;
    SDB_HIP(hipMemcpyAsync(&fl, big_count + 1, 4, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipStreamSynchronize(stream));
    if (fl & 1u)
      return fail(SDB_ERR_INVALID, "an insert's greedy search expanded more than %u nodes: the visit list does not fit "
                                   "the device log (searchSize %u)", vis_cap, L);
    return SDB_OK;
  };

  uint64_t n_rounds = 0;
  while (done < n) {
    const uint32_t cur = n0 + (uint32_t)done;  // rows in storage so far = slot of the round's first point
    // a round never exceeds 2 % of the nodes already in the graph -- live ones, deleted rows do not count --
    // (points of one round do not see each other, so early rounds are sequential and rounds grow with the
    // graph); round_size caps it
    uint32_t rs = (uint32_t)((double)(cur - ix->n_dead) * 0.02);
    if (rs < 1) rs = 1;
    if (rs > max_round) rs = max_round;
    if (rs > n - done) rs = (uint32_t)(n - done);
    // ---- greedySearch(vec, 1, SearchSize, nil) for every point of the round (insert.go:22)
    SearchArgs sa{};
    sa.slab = ix->d_slab, sa.adj = ix->d_adj, sa.ids = ix->d_ids;
    sa.bitsets = bitsets, sa.words_per_query = words;
    sa.queries = dvec + done * l.dim;
    sa.dim = l.dim, sa.nblk = l.nblk, sa.ng = l.ng, sa.tail = l.tail, sa.ld = l.ld;
    sa.start_slot = (uint32_t)ix->start_slot;
    sa.start_ext = ix->d_start_ext, sa.start_ext_n = (uint32_t)ix->h_start_ext.size();
    sa.search_size = L, sa.limit = 1, sa.metric = (int)ix->P.metric;
    sa.vis_slots = vis_slots, sa.vis_dists = vis_dists, sa.vis_count = vis_count, sa.vis_cap = vis_cap;
    sa.hash_limit = ix->tune_hash_limit, sa.prefer_bitset = ix->tune_no_hash ? 1u : 0u;
    sa.wide_hash = ix->tune_wide_hash ? 1u : 0u, sa.hash16_probes = ix->tune_hash16_probes;
    sa.pq_narrow = ix->tune_pq_narrow;
    // the warm-up rounds (a round is at most 2 % of the graph: ~330 rounds of up to 512 points before the graph holds
    // 25 600) are small calls like a REST request's and take the workgroup-per-query walk (0.32 instead of 0.54 ms per
    // round); the big rounds one wave per query
    sa.wide_mode = (rs <= 512 && ix->tune_wide_walk != 1) ? 0u : 1u;
    sa.totals = reinterpret_cast<unsigned long long *>(ix->d_bstats);  // [0] n_dist, [1] n_edges
    if (dcache) {
      SDB_W_HIP(hipMemsetAsync(dcache, 0xFF, ((size_t)rs << kDcacheBits) * sizeof(uint2), stream));  // no slot is ~0
      sa.dcache = dcache, sa.dcache_shift = 32 - kDcacheBits;
    }
    if (pq) {
      SDB_W_TRY(pq_build_lut(pq, sa.queries, rs, lut, stream));
      sa.pq_lut = lut, sa.pq_codes = ix->d_codes, sa.pq_M = pq->M, sa.pq_K = pq->K;
      sa.pq_lut_in_lds = (lut_row <= 64 * 1024) ? 1u : 0u;
    }
    if (!search_uses_hash(sa, rs)) SDB_W_HIP(hipMemsetAsync(bitsets, 0, (size_t)rs * words * 4, stream));
    SDB_W_TRY(launch_greedy_search(sa, rs, stream));  // read-only on the graph: a failure here adds nothing to what the rounds before it did
    // ---- robustPrune + back-edges
    BuildArgs ba{};
    ba.slab = ix->d_slab, ba.adj = ix->d_adj, ba.deg = ix->d_deg, ba.clean = ix->d_clean;
    ba.adjdist = ix->d_adjdist, ba.dcount = ix->d_dcount;
    ba.dim = l.dim, ba.nblk = l.nblk, ba.ng = l.ng, ba.tail = l.tail, ba.ld = l.ld;
    ba.metric = (int)ix->P.metric, ba.alpha = ix->P.alpha, ba.R = ix->P.degree_bound;
    ba.first_slot = cur, ba.nnew = rs;
    ba.vis_slots = vis_slots, ba.vis_dists = vis_dists, ba.vis_count = vis_count, ba.vis_cap = vis_cap;
    ba.keys_in = keys_in, ba.keys_sorted = keys_sorted;
    if (pq) ba.pq_codes = ix->d_codes, ba.pq_cdists = pq->d_cdists, ba.pq_M = pq->M, ba.pq_K = pq->K;
    ba.dcache = dcache, ba.dcache_shift = 32 - kDcacheBits;
    ba.big_count = big_count, ba.big_list = big_list, ba.big_min = big_min, ba.big_cap = big_cap;
    ba.start_slot = (uint32_t)ix->start_slot;
    ba.start_ext = ix->d_start_ext, ba.start_ext_n = (uint32_t)ix->h_start_ext.size();
    ba.stats = reinterpret_cast<unsigned long long *>(ix->d_bstats), ba.flags = big_count + 1;
    ba.no_tile = ix->tune_no_tile, ba.prune_done = prune_done, ba.dirty = ix->d_dirty;
    ba.pair_tab = pair_tab, ba.pair_slots = pair_slots, ba.pair_dists = pair_dists;
    ba.pairc = pairc;
    if (def_words && !ix->tune_no_defer)
      ba.def_count = def_words, ba.def_cap = def_cap, ba.def_self = def_words + 4, ba.def_nc = def_words + 4 + def_cap,
      ba.def_done = def_words + 4 + 2 * (size_t)def_cap, ba.def_slots = def_slots, ba.def_dists = def_dists;
    bool start_pruned = false;
    in_round = true;
    int rc = pq ? launch_round<kQuantized, false>(ba, stream, sort_tmp, sort_tmp_bytes, end_bit, &big_scratch, &start_pruned)
         : ix->P.metric == SDB_METRIC_EUCLIDEAN
             ? launch_round_ng<true>(ba, stream, sort_tmp, sort_tmp_bytes, end_bit, &big_scratch, &start_pruned)
             : launch_round_ng<false>(ba, stream, sort_tmp, sort_tmp_bytes, end_bit, &big_scratch, &start_pruned);
    if (rc != SDB_OK) return round_failed(rc);
    if (start_pruned) ix->h_start_ext.clear();  // the device copy is simply no longer referenced (count 0)
    commit(done, done + rs);
    in_round = false;
    done += rs;
    n_rounds++;
  }
  {
    const unsigned long long nr = n_rounds;
    SDB_W_HIP(hipMemcpyAsync(ix->d_bstats + kStRounds, &nr, 8, hipMemcpyHostToDevice, stream));
  }
  if (int rc = check_flags()) return round_failed(rc);
  if (!ix->tx_explicit) {  // the call is its own transaction: publish
    if (int rc = ix->commit(stream)) return round_failed(rc);
    SDB_W_HIP(hipStreamSynchronize(stream));
  }
  return SDB_OK;
  };
  // a C++ exception past this point (a hub list that finds no host memory) takes the same two doors
  try {
    return run();
  } catch (...) {
    const int rc = sdb::on_exception("sdb_index_insert_batch");
    return in_round ? round_failed(rc) : write_failed(rc);
  }
#undef SDB_W_TRY
#undef SDB_W_HIP
}
SDB_API_CATCH("sdb_index_insert_batch")

#include "delete.inc"
