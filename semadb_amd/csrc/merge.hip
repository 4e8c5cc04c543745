// merge.hip -- the shard fan-out merge of ClusterNode.SearchPoints (cluster/actions.go:291-376).
//
// In the 8-GPU layout every GPU holds one shard and answers every query; the per-shard top-k
// lists are all-gathered over xGMI (RCCL, driven by the host process) and merged here.
#include "common.h"
#include "exchange.h"

namespace sdb {

constexpr int kMergeMaxItems = 2048;  // n_shards * per_shard

// The tags of all gathered blocks against rank 0's (semadb_amd.h "Collective calls, order and failure").  Every
// workgroup of the merge runs it -- n_shards x 64 bytes out of L2 -- so that a failed check yields no answer for ANY
// query; workgroup 0 also leaves the verdict where the host reads it (pinned host memory).
__device__ __forceinline__ bool tags_agree(uint32_t n_shards, const char *tags_b, size_t tag_stride, ExchangeVerdict *verdict,
                                           bool report) {
  __shared__ uint32_t s_bad_rank, s_fields, s_status;
  const int t = threadIdx.x;
  if (t == 0) s_bad_rank = 0xFFFFFFFFu, s_fields = 0, s_status = 0;
  __syncthreads();
  const sdb_block_tag *t0 = reinterpret_cast<const sdb_block_tag *>(tags_b);
  if ((uint32_t)t < n_shards) {
    const sdb_block_tag *tg = reinterpret_cast<const sdb_block_tag *>(tags_b + (size_t)t * tag_stride);
    uint32_t f = 0;
    if (tg->magic != SDB_BLOCK_MAGIC) f |= kTagMagic;
    if (tg->seq != t0->seq) f |= kTagSeq;
    if (tg->ticket != t0->ticket) f |= kTagTicket;
    if (tg->nq != t0->nq) f |= kTagNq;
    if (tg->per_shard != t0->per_shard) f |= kTagPerShard;
    if (tg->limit != t0->limit) f |= kTagLimit;
    if (tg->query_hash != t0->query_hash) f |= kTagQueryHash;
    if (tg->rank != (uint32_t)t) f |= kTagRank;
    const uint32_t st = tg->status;
    if (f || st) {
      atomicMin(&s_bad_rank, (uint32_t)t);
      atomicOr(&s_fields, f);
    }
    if (st) atomicMax(&s_status, st | ((uint32_t)t << 8));  // any failed shard (the highest-numbered one is named)
  }
  __syncthreads();
  const bool ok = s_bad_rank == 0xFFFFFFFFu;
  if (report && t == 0 && verdict) {
    verdict->bad_rank = ok ? 0 : s_bad_rank;
    verdict->fields = s_fields;
    verdict->status = s_status & 0xFFu;
    verdict->status_rank = s_status >> 8;
    verdict->seq = t0->seq;
    verdict->ticket = t0->ticket;
    __threadfence_system();
    verdict->state = ok ? kVerdictOk : (s_status ? kVerdictShardFailed : kVerdictMismatch);
    __threadfence_system();
  }
  return ok;
}

// one 64-thread block per query; rank-by-counting under the total order (dist, shard, id, position) -- the
// position only matters for a caller that hands in the same (dist, shard, id) twice: every item still gets a
// rank of its own
//
// The three per-shard arrays are addressed as base + shard * stride (bytes): separate shard-major arrays
// (stride = the array's own per-shard size) or the packed blocks an all-gather delivers back to back
// (stride = the block size for all three, cluster.hip).
__global__ __launch_bounds__(64) void k_topk_merge(uint32_t n_shards, uint64_t nq, uint32_t per_shard,
                                                   const char *__restrict__ ids_b, size_t ids_stride,
                                                   const char *__restrict__ dists_b, size_t dists_stride,
                                                   const char *__restrict__ counts_b, size_t counts_stride,
                                                   uint32_t limit, uint64_t *__restrict__ out_ids,
                                                   float *__restrict__ out_dists, uint32_t *__restrict__ out_shards,
                                                   uint32_t *__restrict__ out_counts, const char *__restrict__ tags_b,
                                                   size_t tag_stride, ExchangeVerdict *verdict) {
  __shared__ float s_d[kMergeMaxItems];
  __shared__ uint64_t s_id[kMergeMaxItems];
  __shared__ uint16_t s_sh[kMergeMaxItems];
  __shared__ uint32_t s_off[65];
  const uint64_t q = blockIdx.x;
  const int t = threadIdx.x;
  if (tags_b && !tags_agree(n_shards, tags_b, tag_stride, verdict, q == 0)) {
    if (t == 0) out_counts[q] = 0;  // no answer rather than the merge of two different requests
    return;
  }
  if (t == 0) {
    uint32_t o = 0;
    for (uint32_t s = 0; s < n_shards; s++) {
      s_off[s] = o;
      uint32_t c = reinterpret_cast<const uint32_t *>(counts_b + (size_t)s * counts_stride)[q];
      o += c < per_shard ? c : per_shard;
    }
    s_off[n_shards] = o;
  }
  __syncthreads();
  const uint32_t total = s_off[n_shards];
  for (uint32_t s = 0; s < n_shards; s++) {
    const uint32_t c = s_off[s + 1] - s_off[s];
    for (uint32_t i = t; i < c; i += 64) {
      const size_t src = (size_t)q * per_shard + i;
      s_d[s_off[s] + i] = reinterpret_cast<const float *>(dists_b + (size_t)s * dists_stride)[src];
      s_id[s_off[s] + i] = reinterpret_cast<const uint64_t *>(ids_b + (size_t)s * ids_stride)[src];
      s_sh[s_off[s] + i] = (uint16_t)s;
    }
  }
  __syncthreads();
  // HybridScore = -dist * weight descending (actions.go:362-364) == dist ascending for weight > 0
  for (uint32_t i = t; i < total; i += 64) {
    const float d = s_d[i];
    const uint64_t id = s_id[i];
    const uint16_t sh = s_sh[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < total; j++) {
      const float dj = s_d[j];
      // a single shard is not re-sorted at all (actions.go:357: `if len(col.ShardIds) > 1`)
      const bool before = n_shards == 1 ? (j < i)
                                        : (dj < d || (dj == d && (s_sh[j] < sh || (s_sh[j] == sh && (s_id[j] < id || (s_id[j] == id && j < i))))));
      rank += before ? 1u : 0u;
    }
    if (rank < limit) {  // truncate to the original limit (actions.go:372-374)
      out_ids[q * limit + rank] = id;
      out_dists[q * limit + rank] = d;
      if (out_shards) out_shards[q * limit + rank] = sh;
    }
  }
  if (t == 0) out_counts[q] = total < limit ? total : limit;
}

int launch_topk_merge(uint32_t n_shards, uint64_t nq, uint32_t per_shard, const void *ids, size_t ids_stride,
                      const void *dists, size_t dists_stride, const void *counts, size_t counts_stride, uint32_t limit,
                      uint64_t *out_ids, float *out_dists, uint32_t *out_shards, uint32_t *out_counts,
                      hipStream_t stream, const void *tags, size_t tag_stride, ExchangeVerdict *verdict) {
  hipLaunchKernelGGL(k_topk_merge, dim3((unsigned)nq), dim3(64), 0, stream, n_shards, nq, per_shard,
                     static_cast<const char *>(ids), ids_stride, static_cast<const char *>(dists), dists_stride,
                     static_cast<const char *>(counts), counts_stride, limit, out_ids, out_dists, out_shards,
                     out_counts, static_cast<const char *>(tags), tag_stride, verdict);
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}

int check_merge_shape(uint32_t n_shards, uint32_t per_shard, uint32_t limit) {
  if (n_shards == 0 || n_shards > 64) return fail(SDB_ERR_INVALID, "n_shards must be 1..64, got %u", n_shards);
  if (limit == 0 || per_shard == 0) return fail(SDB_ERR_INVALID, "limit and per_shard must be positive");
  if ((uint64_t)n_shards * per_shard > kMergeMaxItems)
    return fail(SDB_ERR_INVALID, "n_shards * per_shard = %llu exceeds %d", (unsigned long long)n_shards * per_shard,
                kMergeMaxItems);
  return SDB_OK;
}

}  // namespace sdb

using namespace sdb;

extern "C" {

int sdb_shard_limit(uint32_t limit, uint32_t n_shards, uint32_t max_search_limit, uint32_t *out) try {
  if (!out || n_shards == 0) return fail(SDB_ERR_INVALID, "bad argument");
  // actions.go:291-299: int(float32(limit) * (1/float32(nShards)) * 1.42 + 10)
  int target = (int)((float)limit * (1.0f / (float)n_shards) * 1.42f + 10.0f);
  if (target > (int)max_search_limit) target = (int)max_search_limit;
  if (target > (int)limit) target = (int)limit;
  *out = (uint32_t)target;
  return SDB_OK;
}
SDB_API_CATCH("sdb_shard_limit")

int sdb_topk_merge(uint32_t n_shards, uint64_t nq, uint32_t per_shard, const uint64_t *ids, const float *dists,
                   const uint32_t *counts, uint32_t limit, uint64_t *out_ids, float *out_dists,
                   uint32_t *out_shards, uint32_t *out_counts, int mem, int device, void *stream_) try {
  if (nq == 0) return SDB_OK;
  if (!ids || !dists || !counts || !out_ids || !out_dists || !out_counts) return fail(SDB_ERR_INVALID, "NULL argument");
  SDB_TRY(check_merge_shape(n_shards, per_shard, limit));
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  if (device < 0 || device >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", device);
  DeviceGuard dg(device);
  hipStream_t stream = as_stream(stream_);
  const size_t items = (size_t)n_shards * nq * per_shard;
  const size_t st_i = (size_t)nq * per_shard * 8, st_d = (size_t)nq * per_shard * 4, st_c = (size_t)nq * 4;
  if (mem == SDB_MEM_DEVICE)
    return launch_topk_merge(n_shards, nq, per_shard, ids, st_i, dists, st_d, counts, st_c, limit, out_ids, out_dists,
                             out_shards, out_counts, stream);
  char *buf = nullptr;
  const size_t b_ids = items * 8, b_d = items * 4, b_c = (size_t)n_shards * nq * 4;
  const size_t b_oi = nq * limit * 8, b_od = nq * limit * 4, b_os = nq * limit * 4, b_oc = nq * 4;
  SDB_HIP(hipMalloc(&buf, b_ids + b_d + b_c + b_oi + b_od + b_os + b_oc));
  uint64_t *d_ids = (uint64_t *)buf;
  uint64_t *d_oi = (uint64_t *)(buf + b_ids);
  float *d_d = (float *)(buf + b_ids + b_oi);
  uint32_t *d_c = (uint32_t *)(buf + b_ids + b_oi + b_d);
  float *d_od = (float *)(buf + b_ids + b_oi + b_d + b_c);
  uint32_t *d_os = (uint32_t *)(buf + b_ids + b_oi + b_d + b_c + b_od);
  uint32_t *d_oc = (uint32_t *)(buf + b_ids + b_oi + b_d + b_c + b_od + b_os);
  hipError_t e = hipMemcpyAsync(d_ids, ids, b_ids, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_d, dists, b_d, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_c, counts, b_c, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemsetAsync(d_oi, 0, b_oi, stream);
  if (e == hipSuccess) e = hipMemsetAsync(d_od, 0, b_od + b_os + b_oc, stream);
  if (e == hipSuccess &&
      launch_topk_merge(n_shards, nq, per_shard, d_ids, st_i, d_d, st_d, d_c, st_c, limit, d_oi, d_od, d_os, d_oc,
                        stream) != SDB_OK)
    e = hipErrorLaunchFailure;
  if (e == hipSuccess) e = hipMemcpyAsync(out_ids, d_oi, b_oi, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out_dists, d_od, b_od, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess && out_shards) e = hipMemcpyAsync(out_shards, d_os, b_os, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out_counts, d_oc, b_oc, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(buf);
  if (e != hipSuccess) return fail(SDB_ERR_DEVICE, "topk_merge failed: %s", hipGetErrorString(e));
  return SDB_OK;
}
SDB_API_CATCH("sdb_topk_merge")

}  // extern "C"
