// cluster.hip -- the shard fan-out exchange of ClusterNode.SearchPoints (cluster/actions.go:275-379) behind the
// C ABI: one rank = one shard = one MI355X; the gather step (actions.go:316-351, msgpack net/rpc in the reference)
// is one RCCL all-gather of the fixed-size per-shard result blocks over xGMI, the merge (:357-376) is
// k_topk_merge (merge.hip) on every rank.
//
// Streams.  A cluster handle owns an exchange stream.  A call records an event on the caller's stream (the search
// that produced the block is enqueued there), makes the exchange stream wait for it, and enqueues all-gather +
// merge on the exchange stream -- the caller's stream is free for the next batch's graph walk at once.  The
// message is 124 KB per rank at 1024 x 10: latency-bound, one step over the direct xGMI links, so there is no
// bucket or ring tuning to do; what matters is that it never sits on the search stream.
#include <rccl/rccl.h>

#include <algorithm>

#include "index.h"

namespace sdb {
int launch_topk_merge(uint32_t n_shards, uint64_t nq, uint32_t per_shard, const void *ids, size_t ids_stride,
                      const void *dists, size_t dists_stride, const void *counts, size_t counts_stride, uint32_t limit,
                      uint64_t *out_ids, float *out_dists, uint32_t *out_shards, uint32_t *out_counts,
                      hipStream_t stream);
int check_merge_shape(uint32_t n_shards, uint32_t per_shard, uint32_t limit);

struct BlockLayout {
  size_t off_d, off_c, bytes;
  BlockLayout(uint64_t nq, uint32_t per) {
    off_d = (size_t)nq * per * 8;
    off_c = off_d + (size_t)nq * per * 4;
    bytes = (off_c + (size_t)nq * 4 + 15) & ~(size_t)15;  // blocks sit back to back in the gathered buffer
  }
};
}  // namespace sdb

#define SDB_NCCL(expr)                                                                                  \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess)                                                                              \
      return sdb::fail(SDB_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, \
                       __LINE__);                                                                       \
  } while (0)

struct sdb_cluster {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t xs = nullptr;       // the exchange stream
  hipEvent_t produced = nullptr;  // caller's stream -> exchange stream
  hipEvent_t finished = nullptr;  // last exchange enqueued so far
  bool any = false;
  char *gathered = nullptr;  // [world][block bytes]
  size_t gathered_bytes = 0;
  static constexpr int kRing = 4;  // blocks search_batch may have in flight
  struct Slot {
    char *block = nullptr;
    size_t bytes = 0;
    hipEvent_t consumed = nullptr;  // its all-gather has read it
    bool used = false;
  } ring[kRing];
  unsigned next = 0;
  // staging for SDB_MEM_HOST callers
  char *stage = nullptr;
  size_t stage_bytes = 0;
  std::mutex mu;

  int ensure(char **p, size_t *have, size_t want) {
    if (want <= *have) return SDB_OK;
    SDB_HIP(hipStreamSynchronize(xs));  // nothing may still read the old buffer
    if (*p) (void)hipFree(*p);
    *p = nullptr, *have = 0;
    SDB_HIP(hipMalloc(p, want));
    *have = want;
    return SDB_OK;
  }
};

using namespace sdb;

static int cluster_finish_init(sdb_cluster *c) {
  DeviceGuard dg(c->device);
  SDB_HIP(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
  SDB_HIP(hipEventCreateWithFlags(&c->produced, hipEventDisableTiming));
  SDB_HIP(hipEventCreateWithFlags(&c->finished, hipEventDisableTiming));
  for (auto &s : c->ring) SDB_HIP(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
  return SDB_OK;
}

// enqueue all-gather + merge of `block` on the exchange stream, after what `stream` holds now.  Device outputs.
static int exchange(sdb_cluster *c, uint64_t nq, uint32_t per_shard, const void *block, uint32_t limit,
                    uint64_t *out_ids, float *out_dists, uint32_t *out_shards, uint32_t *out_counts,
                    hipStream_t stream, hipEvent_t consumed) {
  const BlockLayout bl(nq, per_shard);
  SDB_TRY(c->ensure(&c->gathered, &c->gathered_bytes, bl.bytes * (size_t)c->world));
  SDB_HIP(hipEventRecord(c->produced, stream));
  SDB_HIP(hipStreamWaitEvent(c->xs, c->produced, 0));
  SDB_NCCL(ncclAllGather(block, c->gathered, bl.bytes, ncclUint8, c->comm, c->xs));
  if (consumed) SDB_HIP(hipEventRecord(consumed, c->xs));
  SDB_TRY(launch_topk_merge((uint32_t)c->world, nq, per_shard, c->gathered, bl.bytes, c->gathered + bl.off_d, bl.bytes,
                            c->gathered + bl.off_c, bl.bytes, limit, out_ids, out_dists, out_shards, out_counts, c->xs));
  SDB_HIP(hipEventRecord(c->finished, c->xs));
  c->any = true;
  return SDB_OK;
}

extern "C" {

int sdb_cluster_unique_id(uint8_t *id) {
  if (!id) return fail(SDB_ERR_INVALID, "id is NULL");
  static_assert(sizeof(ncclUniqueId) == SDB_CLUSTER_ID_BYTES, "unique id size");
  ncclUniqueId u;
  SDB_NCCL(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return SDB_OK;
}

int sdb_cluster_create(int rank, int world, const uint8_t *id, int device, sdb_cluster **out) {
  if (!id || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (world < 1 || world > 64 || rank < 0 || rank >= world)
    return fail(SDB_ERR_INVALID, "rank %d / world %d out of range (1..64 shards)", rank, world);
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  if (device < 0 || device >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", device);
  DeviceGuard dg(device);
  if (!dg.ok) return fail(SDB_ERR_DEVICE, "hipSetDevice(%d) failed", device);
  auto *c = new sdb_cluster();
  c->rank = rank, c->world = world, c->device = device;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(SDB_ERR_DEVICE, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device,
                ncclGetErrorString(r));
  }
  int rc = cluster_finish_init(c);
  if (rc != SDB_OK) {
    sdb_cluster_destroy(c);
    return rc;
  }
  *out = c;
  return SDB_OK;
}

int sdb_cluster_create_local(int n, const int *devices, sdb_cluster **out) {
  if (!out) return fail(SDB_ERR_INVALID, "out is NULL");
  for (int i = 0; i < n; i++) out[i] = nullptr;
  if (n < 1 || n > 64) return fail(SDB_ERR_INVALID, "shard count %d out of range (1..64)", n);
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  std::vector<int> devs(n);
  for (int i = 0; i < n; i++) {
    devs[i] = devices ? devices[i] : i;
    if (devs[i] < 0 || devs[i] >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", devs[i]);
    for (int j = 0; j < i; j++)
      if (devs[j] == devs[i]) return fail(SDB_ERR_INVALID, "device %d named twice: one shard per GPU", devs[i]);
  }
  std::vector<ncclComm_t> comms(n, nullptr);
  SDB_NCCL(ncclCommInitAll(comms.data(), n, devs.data()));
  int rc = SDB_OK;
  for (int i = 0; i < n; i++) {
    auto *c = new sdb_cluster();
    c->rank = i, c->world = n, c->device = devs[i], c->comm = comms[i];
    out[i] = c;
    if (rc == SDB_OK) rc = cluster_finish_init(c);
  }
  if (rc != SDB_OK)
    for (int i = 0; i < n; i++) {
      sdb_cluster_destroy(out[i]);
      out[i] = nullptr;
    }
  return rc;
}

int sdb_cluster_destroy(sdb_cluster *c) {
  if (!c) return SDB_OK;
  DeviceGuard dg(c->device);
  if (c->xs) (void)hipStreamSynchronize(c->xs);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->gathered) (void)hipFree(c->gathered);
  if (c->stage) (void)hipFree(c->stage);
  for (auto &s : c->ring) {
    if (s.block) (void)hipFree(s.block);
    if (s.consumed) (void)hipEventDestroy(s.consumed);
  }
  if (c->produced) (void)hipEventDestroy(c->produced);
  if (c->finished) (void)hipEventDestroy(c->finished);
  if (c->xs) (void)hipStreamDestroy(c->xs);
  delete c;
  return SDB_OK;
}

int sdb_cluster_info(const sdb_cluster *c, int *rank, int *world, int *device) {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (device) *device = c->device;
  return SDB_OK;
}

int sdb_cluster_block_layout(uint64_t nq, uint32_t per_shard, size_t *off_dists, size_t *off_counts, size_t *bytes) {
  if (per_shard == 0) return fail(SDB_ERR_INVALID, "per_shard must be positive");
  const BlockLayout bl(nq, per_shard);
  if (off_dists) *off_dists = bl.off_d;
  if (off_counts) *off_counts = bl.off_c;
  if (bytes) *bytes = bl.bytes;
  return SDB_OK;
}

int sdb_cluster_wait(sdb_cluster *c, void *stream) {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  std::lock_guard<std::mutex> g(c->mu);
  if (!c->any) return SDB_OK;
  DeviceGuard dg(c->device);
  SDB_HIP(hipStreamWaitEvent(as_stream(stream), c->finished, 0));
  return SDB_OK;
}

int sdb_cluster_synchronize(sdb_cluster *c) {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  DeviceGuard dg(c->device);
  SDB_HIP(hipStreamSynchronize(c->xs));
  return SDB_OK;
}

// host outputs: merged block staged in device memory, copied back on the exchange stream, synchronised
static int merged_to_host(sdb_cluster *c, uint64_t nq, uint32_t limit, char *stage, uint64_t *out_ids, float *out_dists,
                          uint32_t *out_shards, uint32_t *out_counts) {
  const size_t b_i = nq * limit * 8, b_d = nq * limit * 4;
  SDB_HIP(hipMemcpyAsync(out_ids, stage, b_i, hipMemcpyDeviceToHost, c->xs));
  SDB_HIP(hipMemcpyAsync(out_dists, stage + b_i, b_d, hipMemcpyDeviceToHost, c->xs));
  if (out_shards) SDB_HIP(hipMemcpyAsync(out_shards, stage + b_i + b_d, b_d, hipMemcpyDeviceToHost, c->xs));
  SDB_HIP(hipMemcpyAsync(out_counts, stage + b_i + 2 * b_d, nq * 4, hipMemcpyDeviceToHost, c->xs));
  SDB_HIP(hipStreamSynchronize(c->xs));
  return SDB_OK;
}

int sdb_cluster_allgather_merge(sdb_cluster *c, uint64_t nq, uint32_t per_shard, const void *block, uint32_t limit,
                                uint64_t *out_ids, float *out_dists, uint32_t *out_shards, uint32_t *out_counts,
                                int mem, void *stream_) {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  if (nq == 0) return SDB_OK;
  if (!block || !out_ids || !out_dists || !out_counts) return fail(SDB_ERR_INVALID, "NULL argument");
  SDB_TRY(check_merge_shape((uint32_t)c->world, per_shard, limit));
  std::lock_guard<std::mutex> g(c->mu);
  DeviceGuard dg(c->device);
  hipStream_t stream = as_stream(stream_);
  if (mem == SDB_MEM_DEVICE)
    return exchange(c, nq, per_shard, block, limit, out_ids, out_dists, out_shards, out_counts, stream, nullptr);
  const size_t b_i = nq * limit * 8, b_d = nq * limit * 4;
  SDB_TRY(c->ensure(&c->stage, &c->stage_bytes, b_i + 2 * b_d + nq * 4));
  char *st = c->stage;
  SDB_TRY(exchange(c, nq, per_shard, block, limit, (uint64_t *)st, (float *)(st + b_i), (uint32_t *)(st + b_i + b_d),
                   (uint32_t *)(st + b_i + 2 * b_d), stream, nullptr));
  return merged_to_host(c, nq, limit, st, out_ids, out_dists, out_shards, out_counts);
}

int sdb_cluster_search_batch(sdb_cluster *c, sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit,
                             uint32_t search_size, uint64_t *out_ids, float *out_dists, uint32_t *out_shards,
                             uint32_t *out_counts, int mem, void *stream_) {
  if (!c || !ix) return fail(SDB_ERR_INVALID, "NULL handle");
  if (nq == 0) return SDB_OK;
  if (!queries || !out_ids || !out_dists || !out_counts) return fail(SDB_ERR_INVALID, "NULL argument");
  if (limit < 1) return fail(SDB_ERR_INVALID, "invalid limit %u", limit);
  if (search_size < limit)  // search.go:23-25, checked against the query's own limit
    return fail(SDB_ERR_INVALID, "searchSize (%u) must be greater than k (%u)", search_size, limit);
  if (ix->P.device != c->device) return fail(SDB_ERR_INVALID, "index lives on device %d, cluster rank on %d", ix->P.device, c->device);
  uint32_t per_shard = 0;
  SDB_TRY(sdb_shard_limit(limit, (uint32_t)c->world, 75, &per_shard));  // actions.go:291-299, MaxSearchLimit 75
  SDB_TRY(check_merge_shape((uint32_t)c->world, per_shard, limit));
  std::lock_guard<std::mutex> g(c->mu);
  DeviceGuard dg(c->device);
  hipStream_t stream = as_stream(stream_);
  const BlockLayout bl(nq, per_shard);
  sdb_cluster::Slot &slot = c->ring[c->next % sdb_cluster::kRing];
  c->next++;
  if (slot.bytes < bl.bytes) {
    if (slot.used) SDB_HIP(hipEventSynchronize(slot.consumed));
    if (slot.block) (void)hipFree(slot.block);
    slot.block = nullptr, slot.bytes = 0;
    SDB_HIP(hipMalloc(&slot.block, bl.bytes));
    slot.bytes = bl.bytes;
    SDB_HIP(hipMemset(slot.block, 0, bl.bytes));  // the padding travels too
  }
  // the walk may overwrite the block only after the all-gather that last used it has read it
  if (slot.used) SDB_HIP(hipStreamWaitEvent(stream, slot.consumed, 0));
  slot.used = true;
  const float *dq = queries;
  char *st = nullptr;
  const size_t b_i = nq * limit * 8, b_d = nq * limit * 4, b_out = b_i + 2 * b_d + nq * 4;
  if (mem == SDB_MEM_HOST) {
    // the staged queries follow the merged block in the staging buffer
    const size_t b_q = nq * (size_t)ix->lay.dim * 4;
    SDB_TRY(c->ensure(&c->stage, &c->stage_bytes, ((b_out + 255) & ~(size_t)255) + b_q));
    st = c->stage;
    float *sq = (float *)(st + ((b_out + 255) & ~(size_t)255));
    SDB_HIP(hipMemcpyAsync(sq, queries, b_q, hipMemcpyHostToDevice, stream));
    dq = sq;
  }
  // IndexVamana.Search on this shard, straight into the message.  The shard truncates to the per-shard limit
  // (sr.Limit, actions.go:299); a smaller limit is a prefix of a larger one, so searching at per_shard is the same.
  SDB_TRY(sdb_index_search_batch(ix, nq, dq, per_shard, search_size, nullptr, nullptr, (uint64_t *)slot.block,
                                 (float *)(slot.block + bl.off_d), (uint32_t *)(slot.block + bl.off_c), nullptr,
                                 SDB_MEM_DEVICE, stream));
  if (mem == SDB_MEM_DEVICE)
    return exchange(c, nq, per_shard, slot.block, limit, out_ids, out_dists, out_shards, out_counts, stream, slot.consumed);
  SDB_TRY(exchange(c, nq, per_shard, slot.block, limit, (uint64_t *)st, (float *)(st + b_i), (uint32_t *)(st + b_i + b_d),
                   (uint32_t *)(st + b_i + 2 * b_d), stream, slot.consumed));
  return merged_to_host(c, nq, limit, st, out_ids, out_dists, out_shards, out_counts);
}

}  // extern "C"
