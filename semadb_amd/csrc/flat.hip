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

#include "pq.h"
#include "search_kernel.h"

namespace sdb {

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
    const uint32_t cc = live[u] ? c : 0, cc2 = live2[u] ? c2 : 0;
    slot[u] = myslots ? myslots[cc] : first_row + cc;
    slot2[u] = myslots ? myslots[cc2] : first_row + cc2;
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
      if (len == cap) ok = dist < list_tail(cd, cap);  // :104 `dist >= tail -> skip`
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

}  // namespace sdb

using namespace sdb;

extern "C" int sdb_index_flat_search(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit,
                                     const uint64_t *filter_offsets, const uint64_t *filter_ids, uint64_t *out_ids,
                                     float *out_dists, uint32_t *out_counts, int mem, void *stream_) {
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
  std::shared_lock<std::shared_mutex> rl(ix->view_mu);
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
      std::sort(f_slots.begin() + f0, f_slots.end());
      f_off[q + 1] = (uint32_t)f_slots.size();
      max_f = std::max<uint32_t>(max_f, (uint32_t)(f_slots.size() - f0));
    }
  }
  // ---- buffers: staged queries/outputs for host callers, running top lists, one distance block
  const uint32_t chunk = filtered ? std::max<uint32_t>(max_f, 1)
                                  : (uint32_t)std::min<uint64_t>(std::max<uint32_t>(n, 1), (1ull << 28) / nq);
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
  char *buf = nullptr;
  SDB_HIP(hipMalloc(&buf, off));
  struct Free {
    char *p;
    hipStream_t s;
    ~Free() {
      (void)hipStreamSynchronize(s);
      (void)hipFree(p);
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
  for (uint32_t first = 0; first < total; first += chunk) {
    const uint32_t rows = std::min<uint32_t>(chunk, total - first);
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
  hipLaunchKernelGGL(k_flat_emit, dim3((unsigned)nq), dim3(64), 0, stream, top_slot, top_dist, top_len, vw.ids, limit,
                     d_oi, d_od, d_oc);
  SDB_HIP(hipGetLastError());
  {  // a commit must wait for this scan before it hands the copy it reads to the writer
    Workspace *ws = ix->acquire_ws(stream, false);
    if (!ws->launched) (void)hipEventCreateWithFlags(&ws->launched, hipEventDisableTiming);
    if (ws->launched && hipEventRecord(ws->launched, stream) == hipSuccess) ws->launched_valid = true;
    else (void)hipStreamSynchronize(stream);
    ix->release_ws(ws, stream, false);
  }
  rl.unlock();
  if (mem == SDB_MEM_HOST) {
    SDB_HIP(hipMemcpyAsync(out_ids, d_oi, nq * limit * 8, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipMemcpyAsync(out_dists, d_od, nq * limit * 4, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipMemcpyAsync(out_counts, d_oc, nq * 4, hipMemcpyDeviceToHost, stream));
  }
  return SDB_OK;  // `fr` synchronises and frees
}

// vecStore.Set for a flat index (flat.go:46-49): store vectors without touching the graph.
namespace sdb {
int store_rows_public(sdb_index *ix, uint32_t first, uint32_t n, const float *dev_vectors, hipStream_t stream);
}
extern "C" int sdb_index_set_vectors(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors, int mem) {
  if (!ix || !vectors) return fail(SDB_ERR_INVALID, "NULL argument");
  if (n == 0) return SDB_OK;
  if ((uint64_t)ix->n + n >= 0x7FFFFFFFull) return fail(SDB_ERR_INVALID, "too many nodes");
  std::vector<uint64_t> new_ids(n);
  for (uint64_t i = 0; i < n; i++) {
    new_ids[i] = ids ? ids[i] : std::max<uint64_t>(ix->max_node_id, SDB_STARTID) + 1 + i;
    if (ix->slot_of(new_ids[i]) >= 0)
      return fail(SDB_ERR_EXISTS, "point %llu exists: updates are not on the device path", (unsigned long long)new_ids[i]);
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
  std::unique_lock<std::shared_mutex> wl(ix->view_mu);
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
  wl.unlock();
  if (!ix->tx_explicit) {
    SDB_TRY(ix->commit(nullptr));
    SDB_HIP(hipDeviceSynchronize());
  }
  return SDB_OK;
}
