// search_kernel.h -- K2: GPU-resident greedySearch, one wavefront per query.
//
// Restates shard/index/vamana/search.go:9-102 (greedySearch) on top of
// shard/index/vamana/distset.go:166-200 (DistSet.AddWithLimit) with bit-identical distances
// (dist_core.h), so result ids, distances, visit order, n_dist and n_hop equal the reference's.
//
// Wave-level mapping (64 lanes, no LDS, no barriers):
//   * candidate set S (cap = searchSize): a sorted array held in VGPRs, entry e in lane e%64 of
//     register e/64; the `visited` flag rides in bit 31 of the slot word.
//   * one hop: wave-uniform pick of the first unvisited entry -> one coalesced 256-byte read of
//     its adjacency row (lane j = edge j, edge order preserved) -> per-lane test-and-set in the
//     query's visited bitset (atomicOr; the ids of one row are distinct) -> distances for the new
//     neighbours, two candidates per wave instruction (one per 32-lane half), 16-byte row loads,
//     up to U pairs of rows in flight -> AddWithLimit replayed in edge order for the neighbours
//     that beat the current tail (the tail only shrinks, so a neighbour that fails the current
//     threshold can never pass a later one).
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
  uint32_t pq_lut_in_lds;  // != 0: the kernel copies its LUT into LDS first
};

template <int NG>
struct ChunkPairs {
  // pairs of rows kept in flight per wave: ~96 VGPRs of loads
  static constexpr int value = NG <= 1 ? 8 : (NG <= 3 ? 8 : (NG <= 4 ? 6 : (NG <= 6 ? 4 : 3)));
};

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ float rlf(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// U pairs of (query, candidate) raw distances; slot[u] is this lane's candidate row (same for the
// 32 lanes of a half).  res[u] is valid in lanes 0 and 32.
template <int NG, bool L2, int U>
__device__ __forceinline__ void chunk_dist(const float *__restrict__ slab, uint32_t ld, uint32_t tail,
                                           const float4 (&xq)[NG > 0 ? NG : 1], float xt,
                                           const uint32_t (&slot)[U], float (&res)[U], int lane) {
  const int L = lane & 31;
  float4 y[U][NG > 0 ? NG : 1];
  float yt[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const float *row = slab + (size_t)slot[u] * ld;
    const float4 *r4 = reinterpret_cast<const float4 *>(row) + L;
#pragma unroll
    for (int g = 0; g < NG; g++) y[u][g] = r4[g * 32];
    yt[u] = tail ? row[NG * 128 + L] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    float acc = 0.0f;
#pragma unroll
    for (int g = 0; g < NG; g++) acc = chain4<L2>(acc, xq[g], y[u][g]);
    float t = tail ? tail_chain<L2>(xt, yt[u], tail, lane) : 0.0f;
    res[u] = asm_reduce(acc, t, lane);
  }
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
  for (uint32_t g = 0; g < ng; g++) {
    float4 x = q4[g * 32];
    float4 y[U];
#pragma unroll
    for (int u = 0; u < U; u++) y[u] = r4[u][g * 32];
#pragma unroll
    for (int u = 0; u < U; u++) acc[u] = chain4<L2>(acc[u], x, y[u]);
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
template <int NG, bool L2>
struct PlainDist {
  static constexpr int NGR = NG > 0 ? NG : 1;
  static constexpr int U = NG >= 0 ? ChunkPairs<NG>::value : 4;
  float4 xq[NGR];
  float xt;
  float *qs;

  __device__ __forceinline__ void init(const SearchArgs &a, uint32_t q, int lane, float *lds) {
    const int L = lane & 31;
    const float *__restrict__ qv = a.queries + (size_t)q * a.dim;
    qs = lds;
    xt = 0.0f;
    if constexpr (NG >= 0) {
#pragma unroll
      for (int g = 0; g < NG; g++)
        xq[g] = make_float4(q_elem(qv, a.nblk, g, 0, L), q_elem(qv, a.nblk, g, 1, L),
                            q_elem(qv, a.nblk, g, 2, L), q_elem(qv, a.nblk, g, 3, L));
      if (NG == 0) xq[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      xt = (a.tail && (uint32_t)L < a.tail) ? qv[a.nblk * 32 + L] : 0.0f;
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
    if constexpr (NG >= 0) chunk_dist<NG, L2, U>(a.slab, a.ld, a.tail, xq, xt, slot, res, lane);
    else chunk_dist_lds<L2, U>(a.slab, a.ld, a.ng, a.tail, qs, slot, res, lane);
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

  // distances of the new neighbours of one hop: lane j (bit j of pend) gets dist(query, row nb_j)
  __device__ __forceinline__ float hop(const SearchArgs &a, uint32_t nb, uint64_t pend, int lane) {
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

// Fitted product quantizer: dist = sum_i lut[i*K + code_i], plain fp32 adds in index order
// (product.go:271-275).  One lane per neighbour: all new neighbours of a hop in one pass.
struct PQDist {
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
  __device__ __forceinline__ float sum(const SearchArgs &a, uint32_t slot) const {
    const uint8_t *__restrict__ c = a.pq_codes + (size_t)slot * a.pq_M;
    float dist = 0.0f;
    for (uint32_t i = 0; i < a.pq_M; i++) dist += lut[i * a.pq_K + c[i]];
    return dist;
  }
  __device__ __forceinline__ float one(const SearchArgs &a, uint32_t s, int lane) { return sum(a, s); }
  __device__ __forceinline__ float hop(const SearchArgs &a, uint32_t nb, uint64_t pend, int lane) {
    return ((pend >> lane) & 1ull) ? sum(a, nb) : 0.0f;
  }
};

template <class Dist, int NREG>
__global__ __launch_bounds__(64) void k_greedy_search(const SearchArgs a) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  Dist dist;
  dist.init(a, q, lane, lds_f);

  uint32_t *__restrict__ bits = a.bitsets + (size_t)q * a.words_per_query;
  uint32_t cid[NREG];
  float cd[NREG];
#pragma unroll
  for (int r = 0; r < NREG; r++) cid[r] = kNoSlot, cd[r] = 0.0f;
  int len = 0;
  const int cap = (int)a.search_size;
  uint32_t n_dist = 0, n_hop = 0, n_edges = 0;

  // DistSet.AddWithLimit for one point whose distance is known (distset.go:184-198).
  auto insert = [&](uint32_t id, float d) {
    const bool full = (len == cap);
    const int newlen = full ? len : len + 1;  // :189-194 append, or overwrite the tail
    const int range = newlen - 1;             // entries that the bubble loop compares against
    int pos = 0;                              // :196-198 stop at the first i with !(d < items[i-1])
#pragma unroll
    for (int r = 0; r < NREG; r++) {
      uint64_t m = __ballot((r * 64 + lane) < range && !(d < cd[r]));
      if (m) pos = r * 64 + 64 - __clzll(m);
    }
#pragma unroll
    for (int r = NREG - 1; r >= 0; r--) {
      uint32_t up_id = __shfl_up(cid[r], 1, 64);
      float up_d = __shfl_up(cd[r], 1, 64);
      if (r > 0) {
        uint32_t c_id = rl(cid[r - 1], 63);
        float c_d = rlf(cd[r - 1], 63);
        if (lane == 0) up_id = c_id, up_d = c_d;
      }
      const int e = r * 64 + lane;
      if (e > pos && e < newlen) cid[r] = up_id, cd[r] = up_d;
      else if (e == pos) cid[r] = id, cd[r] = d;
    }
    len = newlen;
  };

  // ---- searchSet.AddWithLimit(startNode)  search.go:57-61
  {
    const uint32_t s = a.start_slot;
    if (lane == 0) atomicOr(&bits[s >> 5], 1u << (s & 31));
    const float d = dist.one(a, s, lane);
    n_dist = 1;
    insert(s, d);
  }

  // ---- main loop search.go:65-98
  while (true) {
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
      if (a.tr_visit && n_hop < a.visit_cap) a.tr_visit[(size_t)q * a.visit_cap + n_hop] = a.ids[pid];
      if (a.vis_slots && n_hop < a.vis_cap) {
        a.vis_slots[(size_t)q * a.vis_cap + n_hop] = pid;
        a.vis_dists[(size_t)q * a.vis_cap + n_hop] = pdist;
      }
    }
    n_hop++;

    // node.neighbours in edge order :77-91
    const uint32_t nb = a.adj[(size_t)pid * kAdjStride + lane];
    const bool valid = nb != kNoSlot;
    n_edges += (uint32_t)__popcll(__ballot(valid));
    bool isnew = false;
    if (valid) {  // CheckAndVisit distset.go:174 -- marks before any distance test
      const uint32_t bit = 1u << (nb & 31);
      const uint32_t old = atomicOr(&bits[nb >> 5], bit);
      isnew = !(old & bit);
    }
    const uint64_t pend = __ballot(isnew);
    if (!pend) continue;
    n_dist += (uint32_t)__popcll(pend);

    const float mydist = dist.hop(a, nb, pend, lane);  // lane j: distance of edge j

    // AddWithLimit over the new neighbours, in edge order distset.go:184-198
    uint64_t pd = pend;
    while (pd) {
      bool ok = true;
      if (len == cap) {
        float tail_d = 0.0f;
#pragma unroll
        for (int r = 0; r < NREG; r++)
          if (((cap - 1) >> 6) == r) tail_d = rlf(cd[r], (cap - 1) & 63);
        ok = !(mydist > tail_d);  // :184 strict '>'
      }
      const uint64_t am = __ballot(ok) & pd;
      if (!am) break;
      const int j = __ffsll((unsigned long long)am) - 1;
      const float d = rlf(mydist, j);
      const uint32_t id = rl(nb, j);
      pd = (j == 63) ? 0ull : ((pd >> (j + 1)) << (j + 1));
      insert(id, d);
    }
  }

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
    if (lane == 0) a.out_counts[q] = (uint32_t)(base < (int)a.limit ? base : (int)a.limit);
  }
  if (lane == 0) {
    if (a.tr_ndist) a.tr_ndist[q] = n_dist;
    if (a.tr_nhop) a.tr_nhop[q] = n_hop;
    if (a.tr_nedges) a.tr_nedges[q] = n_edges;
    if (a.vis_count) a.vis_count[q] = n_hop;
  }
}

// host-side launcher: picks the instantiation for (ng, metric, search_size)
int launch_greedy_search(const SearchArgs &a, uint32_t nq, hipStream_t stream);

}  // namespace sdb
