// cluster.hip -- the shard fan-out exchange of ClusterNode.SearchPoints (cluster/actions.go:275-379) behind the
// C ABI: one rank = one shard; the gather step (actions.go:316-351, msgpack net/rpc in the reference) is one
// all-gather of the fixed-size per-shard result blocks, the merge (:357-376) is k_topk_merge (merge.hip) on every rank.
//
// Transports.  (1) RCCL: one rank per MI355X, ncclAllGather over xGMI -- the production shape.  (2) shared device:
// all ranks of one process on ONE GPU (several shards of a collection on one GPU; also how a one-GPU box runs the
// whole N-shard protocol): the blocks move by device-to-device copies behind a host rendezvous -- the last rank to
// arrive for a sequence number enqueues, for every rank, the copies of all blocks into that rank's gathered buffer
// and its merge.  Everything around the gather is the same code for both: ticket order, tags, error propagation.
//
// Streams.  A cluster handle owns an exchange stream.  A call records an event on the stream that carries the search
// (the caller's for device memory, a stream of the ring slot for host memory), makes the exchange stream wait for it,
// and enqueues all-gather + merge there -- the search stream is free for the next batch's graph walk at once.  The
// message is 124 KB per rank at 1024 x 10: latency-bound, one step over the direct xGMI links, so there is no
// bucket or ring tuning to do; what matters is that it never sits on the search stream.
//
// Order and failure (include/semadb_amd.h "Collective calls, order and failure"): calls enter the exchange in ticket
// order; every block carries a tag that the merge compares across ranks; a rank whose shard search failed enters
// anyway with its status in the tag.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <map>
#include <set>

#include "exchange.h"
#include "index.h"
#include "turnstile.h"

namespace sdb {

struct BlockLayout {
  size_t off_d, off_c, off_t, bytes;
  BlockLayout(uint64_t nq, uint32_t per) {
    off_d = (size_t)nq * per * 8;
    off_c = off_d + (size_t)nq * per * 4;
    off_t = (off_c + (size_t)nq * 4 + 15) & ~(size_t)15;
    bytes = off_t + SDB_BLOCK_TAG_BYTES;  // blocks sit back to back in the gathered buffer
  }
};

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30, x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27, x *= 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

// Writes a block's tag.  The hash of the queries is a sum of per-element mixes of (position, bit pattern): order
// independent across threads, position dependent across the batch -- two ranks that were handed different queries,
// or the same queries in a different order, disagree.  The tag was zeroed by a memset on the same stream.
__global__ __launch_bounds__(256) void k_stamp_tag(sdb_block_tag *tag, uint64_t nq, uint32_t per_shard, uint32_t limit,
                                                   uint32_t rank, uint64_t seq, uint64_t ticket, uint32_t status,
                                                   const uint32_t *__restrict__ qbits, uint64_t n_words) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    tag->magic = SDB_BLOCK_MAGIC, tag->status = status, tag->seq = seq, tag->ticket = ticket, tag->nq = nq;
    tag->per_shard = per_shard, tag->limit = limit, tag->rank = rank;
  }
  if (!qbits) return;
  uint64_t h = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
    h += mix64((i << 32) ^ (uint64_t)qbits[i] ^ 0x9e3779b97f4a7c15ull);
  for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o, 64);
  if ((threadIdx.x & 63) == 0 && h) atomicAdd(reinterpret_cast<unsigned long long *>(&tag->query_hash), (unsigned long long)h);
}

static int stamp_tag(void *block, const BlockLayout &bl, uint64_t nq, uint32_t per_shard, uint32_t limit, uint32_t rank,
                     uint64_t seq, uint64_t ticket, uint32_t status, const float *queries, uint32_t dim, hipStream_t stream) {
  sdb_block_tag *tag = reinterpret_cast<sdb_block_tag *>(static_cast<char *>(block) + bl.off_t);
  SDB_HIP(hipMemsetAsync(tag, 0, SDB_BLOCK_TAG_BYTES, stream));
  const uint64_t words = queries ? nq * dim : 0;
  const unsigned grid = words ? (unsigned)std::min<uint64_t>(64, (words + 1023) / 1024) : 1;
  hipLaunchKernelGGL(k_stamp_tag, dim3(grid), dim3(256), 0, stream, tag, nq, per_shard, limit, rank, seq, ticket, status,
                     reinterpret_cast<const uint32_t *>(queries), words);
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}

static std::string describe(const ExchangeVerdict &v, int world) {
  char buf[512];
  if (v.state == kVerdictShardFailed) {
    snprintf(buf, sizeof(buf), "shard exchange %llu (ticket %llu): the search on shard %u of %d failed with status %u; no "
             "answer was produced for this request", (unsigned long long)v.seq, (unsigned long long)v.ticket,
             v.status_rank, world, v.status);
    return buf;
  }
  std::string f;
  const struct { uint32_t bit; const char *name; } names[] = {
      {kTagMagic, "magic"}, {kTagSeq, "sequence number"}, {kTagTicket, "ticket"}, {kTagNq, "nq"},
      {kTagPerShard, "per_shard"}, {kTagLimit, "limit"}, {kTagQueryHash, "query hash"}, {kTagRank, "rank"}};
  for (auto &n : names)
    if (v.fields & n.bit) f += (f.empty() ? "" : ", ") + std::string(n.name);
  snprintf(buf, sizeof(buf), "shard exchange %llu (ticket %llu): the blocks gathered from the %d ranks belong to different "
           "requests -- rank %u disagrees with rank 0 on: %s; no answer was produced (collective calls must be issued in the "
           "same order on every rank: pass tickets)", (unsigned long long)v.seq, (unsigned long long)v.ticket, world,
           v.bad_rank, f.c_str());
  return buf;
}

struct Group;

}  // namespace sdb

#define SDB_NCCL(expr)                                                                                  \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess)                                                                              \
      return sdb::fail(SDB_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, \
                       __LINE__);                                                                       \
  } while (0)

// (the order of a rank's calls -- tickets, sequence numbers, ring-slot flags, its lock -- is sdb::OrderState /
// sdb::SlotState of turnstile.h, which compile and are stress-tested without a GPU)
struct sdb_cluster : sdb::OrderState {
  int world = 1, device = 0;
  ncclComm_t comm = nullptr;    // RCCL transport
  sdb::Group *group = nullptr;  // shared-device transport
  hipStream_t xs = nullptr;       // the exchange stream
  hipEvent_t finished = nullptr;  // last exchange enqueued so far
  hipEvent_t copied = nullptr;    // shared transport: this rank's copies of one exchange have read all blocks
  bool any = false;
  char *gathered = nullptr;  // [world][block bytes]
  size_t gathered_bytes = 0;
  static constexpr int kRing = SDB_CLUSTER_RING;  // exchanges one rank may have in flight
  struct Slot : sdb::SlotState {
    char *block = nullptr;  // the shard's result block (search_batch)
    size_t bytes = 0;
    char *stage = nullptr;  // host-memory callers: staged queries + merged outputs on the device ...
    size_t stage_bytes = 0;
    char *hstage = nullptr;  // ... and the merged outputs in pinned host memory
    size_t hstage_bytes = 0;
    hipStream_t hs = nullptr;       // the stream a host-memory caller's search runs on
    hipEvent_t produced = nullptr;  // search stream -> exchange stream
    hipEvent_t done = nullptr;      // every read of the block, the merge and the copies back of this exchange have run
  } ring[kRing];
  // verdicts of the exchanges in flight (pinned host memory, written by the merge kernels)
  static constexpr int kVerdicts = 64;
  sdb::ExchangeVerdict *verdicts = nullptr;
  hipEvent_t vdone[kVerdicts] = {};
  bool vused[kVerdicts] = {};
  bool vhost[kVerdicts] = {};  // a host-memory call in flight will read this one itself
  std::string sticky;  // first failure of a device-memory exchange since the last synchronize
};

namespace sdb {

// one exchange as a rank registered it with the group
struct Arrival {
  sdb_cluster *c = nullptr;
  OrderState *owner = nullptr;  // = c, as the rendezvous knows it
  sdb_cluster::Slot *slot = nullptr;
  const char *block = nullptr;
  hipEvent_t produced = nullptr;
  uint64_t nq = 0;
  uint32_t per_shard = 0, limit = 0;
  uint64_t *o_ids = nullptr;  // device outputs of the merge (the caller's, or the slot's staging)
  float *o_d = nullptr;
  uint32_t *o_s = nullptr, *o_c = nullptr;
  bool host = false;
  bool copy_back = false;  // a host caller that wants this rank's copy of the merged answer
  int vi = 0;
  uint64_t seq = 0, ticket = 0;
};

struct Group : GroupState<Arrival> {};

}  // namespace sdb

using namespace sdb;

static int cluster_finish_init(sdb_cluster *c) {
  DeviceGuard dg(c->device);
  SDB_HIP(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
  SDB_HIP(hipEventCreateWithFlags(&c->finished, hipEventDisableTiming));
  SDB_HIP(hipEventCreateWithFlags(&c->copied, hipEventDisableTiming));
  for (auto &s : c->ring) {
    SDB_HIP(hipEventCreateWithFlags(&s.produced, hipEventDisableTiming));
    SDB_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  }
  SDB_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->verdicts), sizeof(ExchangeVerdict) * sdb_cluster::kVerdicts, hipHostMallocDefault));
  memset((void *)c->verdicts, 0, sizeof(ExchangeVerdict) * sdb_cluster::kVerdicts);
  for (auto &e : c->vdone) SDB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return SDB_OK;
}

static int ensure_dev(char **p, size_t *have, size_t want, hipStream_t drain) {
  if (want <= *have) return SDB_OK;
  if (drain) SDB_HIP(hipStreamSynchronize(drain));  // nothing may still read the old buffer
  if (*p) (void)hipFree(*p);
  *p = nullptr, *have = 0;
  SDB_HIP(hipMalloc(p, want));
  *have = want;
  return SDB_OK;
}

// merged outputs of one exchange behind each other: ids | dists | shards | counts
struct OutLayout {
  size_t b_i, b_d, bytes;
  OutLayout(uint64_t nq, uint32_t limit) : b_i(nq * limit * 8), b_d(nq * limit * 4), bytes(b_i + 2 * b_d + nq * 4) {}
};

// the merge (with the tag check) + the copy back of a host caller's outputs, on the rank's exchange stream
static int enqueue_merge(sdb_cluster *c, const Arrival &a, const BlockLayout &bl) {
  ExchangeVerdict *v = &c->verdicts[a.vi];
  SDB_TRY(launch_topk_merge((uint32_t)c->world, a.nq, a.per_shard, c->gathered, bl.bytes, c->gathered + bl.off_d, bl.bytes,
                            c->gathered + bl.off_c, bl.bytes, a.limit, a.o_ids, a.o_d, a.o_s, a.o_c, c->xs,
                            c->gathered + bl.off_t, bl.bytes, v));
  SDB_HIP(hipEventRecord(c->vdone[a.vi], c->xs));
  if (a.copy_back) {
    const OutLayout ol(a.nq, a.limit);
    SDB_HIP(hipMemcpyAsync(a.slot->hstage, a.slot->stage, ol.bytes, hipMemcpyDeviceToHost, c->xs));
  }
  SDB_HIP(hipEventRecord(a.slot->done, c->xs));
  SDB_HIP(hipEventRecord(c->finished, c->xs));
  c->any = true;
  return SDB_OK;
}

// RCCL transport: all-gather + merge behind the search
static int exchange_rccl(sdb_cluster *c, const Arrival &a) {
  const BlockLayout bl(a.nq, a.per_shard);
  SDB_HIP(hipStreamWaitEvent(c->xs, a.produced, 0));
  SDB_NCCL(ncclAllGather(a.block, c->gathered, bl.bytes, ncclUint8, c->comm, c->xs));
  return enqueue_merge(c, a, bl);
}

// An exchange that cannot run (shapes differ; a rank of the group is gone): the verdict is written from the host,
// behind whatever this rank's block was waiting for, and the outputs say "no answer".  Group mutex held.
static int fail_arrival(const Arrival &a, uint32_t state, uint32_t fields, uint32_t status, uint32_t status_rank) {
  sdb_cluster *c = a.c;
  ExchangeVerdict *v = &c->verdicts[a.vi];
  int rc = SDB_OK;
  auto chk = [&](hipError_t e, const char *what) {
    if (e != hipSuccess && rc == SDB_OK) rc = fail(SDB_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(e));
  };
  chk(hipStreamWaitEvent(c->xs, a.produced, 0), "stream wait");
  chk(hipMemsetAsync(a.o_c, 0, a.nq * 4, c->xs), "clearing the counts");
  // the stream has nothing of this exchange on it that writes the verdict: a plain host store, ordered before the
  // event the reader waits for
  v->bad_rank = 0, v->fields = fields, v->status = status, v->status_rank = status_rank;
  v->seq = a.seq, v->ticket = a.ticket;
  v->state = state;
  chk(hipEventRecord(c->vdone[a.vi], c->xs), "event record");
  if (a.copy_back) chk(hipMemcpyAsync(a.slot->hstage, a.slot->stage, OutLayout(a.nq, a.limit).bytes, hipMemcpyDeviceToHost, c->xs), "copy back");
  chk(hipEventRecord(a.slot->done, c->xs), "event record");
  chk(hipEventRecord(c->finished, c->xs), "event record");
  c->any = true;
  return rc;
}

// shared-device transport: called by the last rank to arrive for a sequence number, group mutex held
static int exchange_shared(std::vector<Arrival> &arr) {
  bool same = true;
  for (auto &a : arr) same &= a.nq == arr[0].nq && a.per_shard == arr[0].per_shard;
  int rc = SDB_OK;
  if (!same) {
    // ranks that disagree on the shape cannot even be gathered: the verdict is written here, no answer for anybody
    for (auto &a : arr) {
      sdb_cluster *c = a.c;
      if (fail_arrival(a, kVerdictMismatch, kTagNq | kTagPerShard, 0, 0) != SDB_OK) rc = SDB_ERR_DEVICE;
      c->any = true;
    }
  } else {
    const BlockLayout bl(arr[0].nq, arr[0].per_shard);
    for (auto &a : arr) {  // every rank gathers all blocks on its own exchange stream
      sdb_cluster *c = a.c;
      for (auto &b : arr) {
        if (rc == SDB_OK && hipStreamWaitEvent(c->xs, b.produced, 0) != hipSuccess) rc = fail(SDB_ERR_DEVICE, "stream wait failed");
        if (rc == SDB_OK && hipMemcpyAsync(c->gathered + (size_t)b.c->rank * bl.bytes, b.block, bl.bytes, hipMemcpyDeviceToDevice, c->xs) != hipSuccess)
          rc = fail(SDB_ERR_DEVICE, "block copy failed");
      }
      if (rc == SDB_OK && hipEventRecord(c->copied, c->xs) != hipSuccess) rc = fail(SDB_ERR_DEVICE, "event record failed");
    }
    for (auto &a : arr) {  // a block may be overwritten once EVERY rank's copy of it has run: `done` stands for that too
      sdb_cluster *c = a.c;
      for (auto &b : arr)
        if (rc == SDB_OK && b.c != c && hipStreamWaitEvent(c->xs, b.c->copied, 0) != hipSuccess) rc = fail(SDB_ERR_DEVICE, "stream wait failed");
      if (rc == SDB_OK) rc = enqueue_merge(c, a, bl);
    }
  }
  for (auto &a : arr) a.slot->pending = false;
  return rc;
}

// hipEventSynchronize with the handle's deadline: 0 = the event has fired, 1 = deadline passed, -1 = the device failed
static int wait_event(sdb_cluster *c, hipEvent_t ev) {
  if (c->deadline_ms == 0) return hipEventSynchronize(ev) == hipSuccess ? 0 : -1;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) return -1;
    const auto waited = std::chrono::steady_clock::now() - t0;
    if (waited > std::chrono::milliseconds(c->deadline_ms)) return 1;
    // an exchange takes a millisecond or two: yield through that, then stop burning the core
    if (waited < std::chrono::milliseconds(4)) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
}

static int wait_event_or_fail(sdb_cluster *c, hipEvent_t ev, const char *what) {
  const int w = wait_event(c, ev);
  if (w == 0) return SDB_OK;
  if (w < 0) return fail(SDB_ERR_DEVICE, "waiting for %s failed", what);
  c->desync = true;  // something this rank enqueued never ran: a peer did not join its collective
  return fail(SDB_ERR_STATE, "rank %d waited %u ms for %s: a peer never joined that exchange; the cluster handle is out of "
              "step with its peers, recreate it", c->rank, c->deadline_ms, what);
}

static int verdict_slot(sdb_cluster *c, uint64_t seq, int *vi) {
  const int i = (int)(seq % sdb_cluster::kVerdicts);
  if (c->vused[i]) {
    SDB_TRY(wait_event_or_fail(c, c->vdone[i], "an earlier exchange"));  // only when kVerdicts exchanges are in flight
    if (c->verdicts[i].state >= kVerdictMismatch && c->sticky.empty()) c->sticky = describe(c->verdicts[i], c->world);
  }
  c->verdicts[i].state = kVerdictNone;
  c->vused[i] = true;
  c->vhost[i] = false;
  *vi = i;
  return SDB_OK;
}

// Common body of the two collective calls.  ix == nullptr: the caller's own block (allgather_merge).
// how far a collective call had got when a C++ exception (out of host memory) ended it: collective() below puts the
// handle into the state the same point reaches through its error returns
struct CallProgress {
  sdb_cluster::Slot *slot = nullptr;
  int vi = -1;
  bool numbered = false;  // took a sequence number ...
  bool entered = false;   // ... and its block is in the exchange
};

static int collective_impl(sdb_cluster *c, sdb_index *ix, uint64_t ticket, uint64_t nq, const float *queries, uint32_t per_shard,
                           void *user_block, uint32_t limit, uint32_t search_size, uint64_t *out_ids, float *out_dists,
                           uint32_t *out_shards, uint32_t *out_counts, int mem, hipStream_t user_stream, bool skip,
                           CallProgress &prog);

static int collective(sdb_cluster *c, sdb_index *ix, uint64_t ticket, uint64_t nq, const float *queries, uint32_t per_shard,
                      void *user_block, uint32_t limit, uint32_t search_size, uint64_t *out_ids, float *out_dists,
                      uint32_t *out_shards, uint32_t *out_counts, int mem, hipStream_t user_stream, bool skip = false) {
  CallProgress prog;
  try {
    return collective_impl(c, ix, ticket, nq, queries, per_shard, user_block, limit, search_size, out_ids, out_dists, out_shards,
                           out_counts, mem, user_stream, skip, prog);
  } catch (...) {
    const int rc = on_exception("shard exchange");
    std::lock_guard<std::mutex> g(*c->mu);  // the call's own lock went with the unwinding (after its turn was given on)
    if (prog.numbered && !prog.entered) c->desync = true;  // a number was spent and the peers wait for its block
    if (prog.slot) prog.slot->busy = false, prog.slot->pending = prog.entered && prog.slot->pending;
    if (prog.vi >= 0 && c->vhost[prog.vi]) c->vhost[prog.vi] = false, c->verdicts[prog.vi].state = kVerdictNone;
    c->cv->notify_all();
    if (mem == SDB_MEM_HOST && out_counts && nq) memset(out_counts, 0, nq * 4);
    return rc;
  }
}

static int collective_impl(sdb_cluster *c, sdb_index *ix, uint64_t ticket, uint64_t nq, const float *queries, uint32_t per_shard,
                           void *user_block, uint32_t limit, uint32_t search_size, uint64_t *out_ids, float *out_dists,
                           uint32_t *out_shards, uint32_t *out_counts, int mem, hipStream_t user_stream, bool skip,
                           CallProgress &prog) {
  const bool host = mem == SDB_MEM_HOST;
  std::unique_lock<std::mutex> lk(*c->mu);
  Turn turn{c, ticket};
  SDB_TRY(turn.enter(lk));
  // ---- what every rank decides alike (same arguments everywhere): no sequence number is spent on these
  if (nq == 0) return SDB_OK;
  if (c->group && c->group->gone >= 0)
    return fail(SDB_ERR_STATE, "rank %d of this shard group has been destroyed: no exchange can complete, recreate the group", c->group->gone);
  if (c->desync) return fail(SDB_ERR_STATE, "this rank left an earlier exchange half-way: the cluster handle is out of step with its peers, recreate it");
  if (limit < 1) return fail(SDB_ERR_INVALID, "invalid limit %u", limit);
  if (ix && search_size < limit)  // search.go:23-25, checked against the query's own limit
    return fail(SDB_ERR_INVALID, "searchSize (%u) must be greater than k (%u)", search_size, limit);
  if (ix || (skip && per_shard == 0)) SDB_TRY(sdb_shard_limit(limit, (uint32_t)c->world, 75, &per_shard));  // actions.go:291-299, MaxSearchLimit 75
  SDB_TRY(check_merge_shape((uint32_t)c->world, per_shard, limit));
  DeviceGuard dg(c->device);
  const BlockLayout bl(nq, per_shard);
  const OutLayout ol(nq, limit);
  // ---- buffers first: an allocation that fails here leaves this rank outside the exchange (its peers' only cure is
  // their own timeout), so nothing that can fail for another reason comes before the rank is sure to get in
  sdb_cluster::Slot *slot = take_slot(c, c->ring, lk);
  if (slot->used) SDB_TRY(wait_event_or_fail(c, slot->done, "this ring slot's previous exchange"));  // staging, events
  int vi = 0;
  SDB_TRY(verdict_slot(c, c->seq, &vi));
  SDB_TRY(ensure_dev(&c->gathered, &c->gathered_bytes, bl.bytes * (size_t)c->world, c->xs));
  char *block = static_cast<char *>(user_block);
  if (ix || skip) {
    if (slot->bytes < bl.bytes) {
      SDB_TRY(ensure_dev(&slot->block, &slot->bytes, bl.bytes, nullptr));
      SDB_HIP(hipMemset(slot->block, 0, bl.bytes));  // the padding travels too
    }
    block = slot->block;
  }
  hipStream_t stream = user_stream;
  size_t o_q = 0;
  if (host) {
    o_q = (ol.bytes + 255) & ~(size_t)255;  // the staged queries follow the merged outputs
    const size_t b_q = ix ? nq * (size_t)ix->lay.dim * 4 : 0;
    SDB_TRY(ensure_dev(&slot->stage, &slot->stage_bytes, o_q + b_q, nullptr));
    if (slot->hstage_bytes < ol.bytes) {
      if (slot->hstage) (void)hipHostFree(slot->hstage);
      slot->hstage = nullptr, slot->hstage_bytes = 0;
      SDB_HIP(hipHostMalloc(reinterpret_cast<void **>(&slot->hstage), ol.bytes, hipHostMallocDefault));
      slot->hstage_bytes = ol.bytes;
    }
    if (!slot->hs) SDB_HIP(hipStreamCreateWithFlags(&slot->hs, hipStreamNonBlocking));
    // a search of the library's own runs on the slot's stream; the caller's own block (allgather_merge) was written
    // by work on the CALLER's stream, and the tag / the `produced` event must be ordered behind that work
    if (ix || skip) stream = slot->hs;
  }
  // ---- from here on this rank WILL enter the exchange; a failure of its own goes into the tag
  // (the group's arrival list for this sequence number gets its room first: nothing between taking the number and
  // registering the arrival may need host memory -- a rank that took a number and then stayed out is out of step)
  if (c->group) c->group->reserve(c->seq);
  const uint64_t seq = c->seq++;
  prog.numbered = true, prog.slot = slot, prog.vi = vi;
  int local_rc = SDB_OK;
  char local_msg[kErrBytes] = {0};
  auto note = [&](int rc) {
    if (rc != SDB_OK && local_rc == SDB_OK) local_rc = rc, memcpy(local_msg, last_error_buf(), kErrBytes);
  };
  const float *dq = queries;
  if (skip) {  // the fan-out gave this request up for this rank: an empty answer under an error flag, so that the
               // peers that did enter are not left inside the all-gather (guard 3)
    note(fail(SDB_ERR_STATE, "ticket %llu was skipped on rank %d", (unsigned long long)ticket, c->rank));
    (void)hipMemsetAsync(block + bl.off_c, 0, nq * 4, stream);
  }
  if (ix) {
    if (ix->P.device != c->device) note(fail(SDB_ERR_INVALID, "index lives on device %d, cluster rank on %d", ix->P.device, c->device));
    if (host && local_rc == SDB_OK) {
      float *sq = reinterpret_cast<float *>(slot->stage + o_q);
      if (hipMemcpyAsync(sq, queries, nq * (size_t)ix->lay.dim * 4, hipMemcpyHostToDevice, stream) != hipSuccess)
        note(fail(SDB_ERR_DEVICE, "H2D copy of the queries failed"));
      dq = sq;
    }
    // IndexVamana.Search on this shard, straight into the message.  The shard truncates to the per-shard limit
    // (sr.Limit, actions.go:299); a smaller limit is a prefix of a larger one, so searching at per_shard is the same.
    if (local_rc == SDB_OK)
      note(sdb_index_search_batch(ix, nq, dq, per_shard, search_size, nullptr, nullptr, (uint64_t *)block,
                                  (float *)(block + bl.off_d), (uint32_t *)(block + bl.off_c), nullptr, SDB_MEM_DEVICE, stream));
    if (local_rc != SDB_OK) (void)hipMemsetAsync(block + bl.off_c, 0, nq * 4, stream);  // an empty answer under the error flag
  }
  bool entered = true;
  const float *hash_src = (ix && (local_rc == SDB_OK || !host)) ? dq : nullptr;
  if (stamp_tag(block, bl, nq, per_shard, limit, (uint32_t)c->rank, seq, ticket, (uint32_t)local_rc, hash_src,
                ix ? ix->lay.dim : 0, stream) != SDB_OK)
    entered = false;
  if (entered && hipEventRecord(slot->produced, stream) != hipSuccess) entered = false;
  Arrival a;
  a.c = c, a.owner = c, a.slot = slot, a.block = block, a.produced = slot->produced, a.nq = nq, a.per_shard = per_shard, a.limit = limit;
  a.host = host, a.vi = vi, a.seq = seq, a.ticket = ticket;
  a.copy_back = host && out_ids != nullptr;
  if (host) {
    a.o_ids = (uint64_t *)slot->stage, a.o_d = (float *)(slot->stage + ol.b_i);
    a.o_s = (uint32_t *)(slot->stage + ol.b_i + ol.b_d), a.o_c = (uint32_t *)(slot->stage + ol.b_i + 2 * ol.b_d);
  } else {
    a.o_ids = out_ids, a.o_d = out_dists, a.o_s = out_shards, a.o_c = out_counts;
  }
  slot->used = true;
  if (entered) {
    if (c->group) {
      slot->pending = true;
      std::vector<Arrival> all;
      if (c->group->arrive(seq, a, &all)) {  // the last rank enqueues for everybody
        if (exchange_shared(all) != SDB_OK) entered = false;
        c->cv->notify_all();
      }
    } else if (exchange_rccl(c, a) != SDB_OK) {
      entered = false;
    }
  }
  prog.entered = entered;
  if (!entered) {  // the device refused an enqueue: this rank is out of step from now on
    c->desync = true;
    slot->pending = false;
    c->cv->notify_all();
    return SDB_ERR_DEVICE;  // message set by the failing call
  }
  if (!host) {
    // asynchronous: the verdict arrives with sdb_cluster_synchronize; a failure of this rank's own search is known now
    if (local_rc != SDB_OK) {
      memcpy(last_error_buf(), local_msg, kErrBytes);
      return local_rc;
    }
    return SDB_OK;
  }
  // ---- host memory: give the turn on, let the next call of this rank enqueue, and wait for this exchange alone
  slot->busy = true;
  c->vhost[vi] = true;
  turn.pass();
  // shared transport: until the last rank has enqueued it
  auto enqueued = [&] { return !slot->pending; };
  if (c->deadline_ms == 0) {
    c->cv->wait(lk, enqueued);
  } else if (!c->cv->wait_for(lk, std::chrono::milliseconds(c->deadline_ms), enqueued)) {
    // the peers never presented this request: take the arrival back.  Nothing of it is on any stream yet, so if it was
    // this rank's latest sequence number the handle is exactly where it was before the call
    c->group->withdraw(seq, c);
    slot->pending = false, slot->busy = false;
    c->vhost[vi] = false, c->verdicts[vi].state = kVerdictNone;
    c->cv->notify_all();
    if (out_counts) memset(out_counts, 0, nq * 4);
    return fail(SDB_ERR_STATE, "shard exchange %llu (ticket %llu): the other ranks did not join within %u ms; the request was "
                "withdrawn on rank %d%s", (unsigned long long)seq, (unsigned long long)ticket, c->deadline_ms, c->rank,
                c->desync ? " and the handle is out of step with its peers, recreate it" : "");
  }
  lk.unlock();
  int rc = SDB_OK;
  {
    const int w = wait_event(c, slot->done);
    if (w < 0) rc = fail(SDB_ERR_DEVICE, "waiting for the exchange failed");
    if (w > 0) {
      lk.lock();
      c->desync = true;
      lk.unlock();
      rc = fail(SDB_ERR_STATE, "shard exchange %llu (ticket %llu): rank %d waited %u ms inside the all-gather, a peer never "
                "joined; the cluster handle is out of step, recreate it", (unsigned long long)seq, (unsigned long long)ticket,
                c->rank, c->deadline_ms);
    }
  }
  if (skip) {  // nothing to deliver: the peers have been told
    if (rc == SDB_OK && c->verdicts[vi].state < kVerdictMismatch)
      rc = fail(SDB_ERR_DEVICE, "the merge of the skipped exchange %llu left no failure verdict", (unsigned long long)seq);
    lk.lock();
    c->verdicts[vi].state = kVerdictNone;
    c->vhost[vi] = false;
    slot->busy = false;
    c->cv->notify_all();
    return rc;
  }
  if (rc == SDB_OK) {
    const ExchangeVerdict v = c->verdicts[vi];
    if (v.state >= kVerdictMismatch) {
      if (local_rc != SDB_OK) rc = fail(local_rc, "%s", local_msg);  // this shard's own failure, in its own words
      else rc = fail(SDB_ERR_STATE, "%s", describe(v, c->world).c_str());
    } else if (v.state != kVerdictOk) {
      rc = fail(SDB_ERR_DEVICE, "the merge of exchange %llu left no verdict", (unsigned long long)seq);
    }
  }
  if (rc == SDB_OK && out_ids) {
    memcpy(out_ids, slot->hstage, ol.b_i);
    memcpy(out_dists, slot->hstage + ol.b_i, ol.b_d);
    if (out_shards) memcpy(out_shards, slot->hstage + ol.b_i + ol.b_d, ol.b_d);
    memcpy(out_counts, slot->hstage + ol.b_i + 2 * ol.b_d, nq * 4);
  } else if (out_counts) {
    memset(out_counts, 0, nq * 4);
  }
  lk.lock();
  c->verdicts[vi].state = kVerdictNone;  // delivered
  c->vhost[vi] = false;
  slot->busy = false;
  c->cv->notify_all();
  return rc;
}

extern "C" {

int sdb_cluster_unique_id(uint8_t *id) try {
  if (!id) return fail(SDB_ERR_INVALID, "id is NULL");
  static_assert(sizeof(ncclUniqueId) == SDB_CLUSTER_ID_BYTES, "unique id size");
  ncclUniqueId u;
  SDB_NCCL(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return SDB_OK;
}
SDB_API_CATCH("sdb_cluster_unique_id")

int sdb_cluster_create(int rank, int world, const uint8_t *id, int device, sdb_cluster **out) try {
  if (!id || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (world < 1 || world > 64 || rank < 0 || rank >= world)
    return fail(SDB_ERR_INVALID, "rank %d / world %d out of range (1..64 shards)", rank, world);
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  if (device < 0 || device >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", device);
  DeviceGuard dg(device);
  if (!dg.ok) return fail(SDB_ERR_DEVICE, "hipSetDevice(%d) failed", device);
  auto *c = new sdb_cluster();  // (std::bad_alloc here: nothing to undo)
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
SDB_API_CATCH("sdb_cluster_create")

int sdb_cluster_create_local(int n, const int *devices, sdb_cluster **out) try {
  if (!out) return fail(SDB_ERR_INVALID, "out is NULL");
  for (int i = 0; i < n; i++) out[i] = nullptr;
  if (n < 1 || n > 64) return fail(SDB_ERR_INVALID, "shard count %d out of range (1..64)", n);
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  std::vector<int> devs(n);
  int repeats = 0;
  for (int i = 0; i < n; i++) {
    devs[i] = devices ? devices[i] : i;
    if (devs[i] < 0 || devs[i] >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", devs[i]);
    for (int j = 0; j < i; j++)
      if (devs[j] == devs[i]) {
        repeats++;
        break;
      }
  }
  const bool shared = n > 1 && repeats == n - 1;  // every rank on the same GPU
  if (repeats && !shared)
    return fail(SDB_ERR_INVALID, "devices must be all distinct (one shard per GPU, RCCL) or all the same (shards sharing one GPU)");
  std::vector<ncclComm_t> comms(n, nullptr);
  Group *g = nullptr;
  if (shared) {
    g = new Group();
    g->world = n, g->device = devs[0], g->alive = n;
  } else {
    SDB_NCCL(ncclCommInitAll(comms.data(), n, devs.data()));
  }
  int rc = SDB_OK;
  for (int i = 0; i < n && rc == SDB_OK; i++) {
    auto *c = new (std::nothrow) sdb_cluster();
    if (!c) {
      rc = fail(SDB_ERR_DEVICE, "out of host memory for rank %d's handle", i);
      break;
    }
    c->rank = i, c->world = n, c->device = devs[i], c->comm = comms[i], c->group = g;
    if (g) c->mu = &g->mu, c->cv = &g->cv;
    out[i] = c;
    rc = cluster_finish_init(c);
  }
  if (rc != SDB_OK) {
    char msg[kErrBytes];
    memcpy(msg, last_error_buf(), kErrBytes);
    int made = 0;
    for (int i = 0; i < n; i++) {
      if (out[i]) made++, sdb_cluster_destroy(out[i]);  // the last of the group's ranks frees the group
      else if (comms[i]) (void)ncclCommDestroy(comms[i]);
      out[i] = nullptr;
    }
    if (g && made < n) {  // ranks that were never made cannot be the last to leave
      g->alive -= n - made;
      if (made == 0 || g->alive <= 0) delete g;
    }
    memcpy(last_error_buf(), msg, kErrBytes);
  }
  return rc;
}
SDB_API_CATCH("sdb_cluster_create_local")

int sdb_cluster_destroy(sdb_cluster *c) try {
  if (!c) return SDB_OK;
  DeviceGuard dg(c->device);
  {
    std::unique_lock<std::mutex> lk(*c->mu);
    // Exchanges registered with the group that have not run yet can never run without this rank: its own arrivals go
    // with the handle, the peers' are failed (verdict "shard failed", no answer) instead of leaving their callers
    // waiting for a last rank that will not come; later calls on the peers are refused (group->gone).
    if (c->group) {
      c->group->gone = c->rank;
      for (auto &kv : c->group->rv)
        for (auto &a : kv.second) {
          if (a.c != c) (void)fail_arrival(a, kVerdictShardFailed, 0, (uint32_t)SDB_ERR_STATE, (uint32_t)c->rank);
          a.slot->pending = false;
        }
      c->group->rv.clear();
      c->cv->notify_all();
    }
  }
  if (c->comm && c->desync) {
    (void)ncclCommAbort(c->comm);  // an all-gather the peers never joined sits on the exchange stream: abort, don't drain
    c->comm = nullptr;
  }
  if (c->xs) (void)hipStreamSynchronize(c->xs);
  for (auto &s : c->ring)
    if (s.hs) (void)hipStreamSynchronize(s.hs);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->gathered) (void)hipFree(c->gathered);
  for (auto &s : c->ring) {
    if (s.block) (void)hipFree(s.block);
    if (s.stage) (void)hipFree(s.stage);
    if (s.hstage) (void)hipHostFree(s.hstage);
    if (s.hs) (void)hipStreamDestroy(s.hs);
    if (s.produced) (void)hipEventDestroy(s.produced);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  for (auto &e : c->vdone)
    if (e) (void)hipEventDestroy(e);
  if (c->verdicts) (void)hipHostFree((void *)c->verdicts);
  if (c->finished) (void)hipEventDestroy(c->finished);
  if (c->copied) (void)hipEventDestroy(c->copied);
  if (c->xs) (void)hipStreamDestroy(c->xs);
  if (c->group) {
    bool last;
    {
      std::lock_guard<std::mutex> g(c->group->mu);
      last = --c->group->alive == 0;
    }
    if (last) delete c->group;
  }
  delete c;
  return SDB_OK;
}
SDB_API_CATCH("sdb_cluster_destroy")

int sdb_cluster_info(const sdb_cluster *c, int *rank, int *world, int *device) try {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (device) *device = c->device;
  return SDB_OK;
}
SDB_API_CATCH("sdb_cluster_info")

int sdb_cluster_set_deadline(sdb_cluster *c, uint32_t milliseconds) try {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  std::lock_guard<std::mutex> g(*c->mu);
  c->deadline_ms = milliseconds;
  c->cv->notify_all();
  return SDB_OK;
}
SDB_API_CATCH("sdb_cluster_set_deadline")

int sdb_cluster_transport(const sdb_cluster *c, char *buf, size_t cap) try {
  if (!c || !buf || cap == 0) return fail(SDB_ERR_INVALID, "NULL argument");
  if (c->group) {
    snprintf(buf, cap, "shared-device: %d ranks on GPU %d, device-to-device copies behind a host rendezvous", c->world, c->device);
    return SDB_OK;
  }
  int ver = 0, count = 0, urank = -1;
  (void)ncclGetVersion(&ver);
  (void)ncclCommCount(c->comm, &count);
  (void)ncclCommUserRank(c->comm, &urank);
  Dl_info di{};
  const char *path = dladdr(reinterpret_cast<const void *>(&ncclAllGather), &di) && di.dli_fname ? di.dli_fname : "?";
  snprintf(buf, cap, "rccl %d.%d.%d (%s): ncclAllGather, communicator of %d ranks, this is rank %d on GPU %d", ver / 10000,
           (ver / 100) % 100, ver % 100, path, count, urank, c->device);
  return SDB_OK;
}
SDB_API_CATCH("sdb_cluster_transport")

int sdb_cluster_skip_ticket(sdb_cluster *c, uint64_t ticket, uint64_t nq, uint32_t per_shard, uint32_t limit) try {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  if (!ticket) return fail(SDB_ERR_INVALID, "ticket 0 is not a ticket");
  if (nq == 0) {  // no rank has entered or will enter for this ticket: the turn passes over it
    std::lock_guard<std::mutex> g(*c->mu);
    return skip_unentered(c, ticket);
  }
  // other ranks may be inside this request's exchange: stand in for it with an empty answer under an error flag
  return collective(c, nullptr, ticket, nq, nullptr, per_shard, nullptr, limit, 0, nullptr, nullptr, nullptr, nullptr,
                    SDB_MEM_HOST, nullptr, true);
}
SDB_API_CATCH("sdb_cluster_skip_ticket")

int sdb_cluster_block_layout(uint64_t nq, uint32_t per_shard, size_t *off_dists, size_t *off_counts, size_t *off_tag,
                             size_t *bytes) try {
  if (per_shard == 0) return fail(SDB_ERR_INVALID, "per_shard must be positive");
  const BlockLayout bl(nq, per_shard);
  if (off_dists) *off_dists = bl.off_d;
  if (off_counts) *off_counts = bl.off_c;
  if (off_tag) *off_tag = bl.off_t;
  if (bytes) *bytes = bl.bytes;
  return SDB_OK;
}
SDB_API_CATCH("sdb_cluster_block_layout")

int sdb_cluster_next_ticket(const sdb_cluster *c, uint64_t *ticket) try {
  if (!c || !ticket) return fail(SDB_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(*c->mu);
  *ticket = c->next_ticket;
  return SDB_OK;
}
SDB_API_CATCH("sdb_cluster_next_ticket")

// shared transport: the exchanges this rank registered are on its stream only once the last rank has arrived
static void wait_enqueued(sdb_cluster *c, std::unique_lock<std::mutex> &lk) {
  if (!c->group) return;
  c->cv->wait(lk, [&] {
    for (auto &s : c->ring)
      if (s.pending) return false;
    return true;
  });
}

int sdb_cluster_wait(sdb_cluster *c, void *stream) try {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  std::unique_lock<std::mutex> lk(*c->mu);
  wait_enqueued(c, lk);
  if (!c->any) return SDB_OK;
  DeviceGuard dg(c->device);
  SDB_HIP(hipStreamWaitEvent(as_stream(stream), c->finished, 0));
  return SDB_OK;
}
SDB_API_CATCH("sdb_cluster_wait")

int sdb_cluster_synchronize(sdb_cluster *c) try {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  DeviceGuard dg(c->device);
  std::unique_lock<std::mutex> lk(*c->mu);
  wait_enqueued(c, lk);
  if (c->any) SDB_TRY(wait_event_or_fail(c, c->finished, "the exchanges in flight"));
  // verdicts of the device-memory exchanges since the last call (host-memory calls took theirs with them)
  std::string first = c->sticky;
  uint64_t first_seq = ~0ull;
  c->sticky.clear();
  for (int i = 0; i < sdb_cluster::kVerdicts; i++) {
    ExchangeVerdict &v = c->verdicts[i];
    if (c->vhost[i] || v.state < kVerdictMismatch) continue;
    if (first.empty() || v.seq < first_seq) first = describe(v, c->world), first_seq = v.seq;
    v.state = kVerdictNone;
  }
  if (!first.empty()) return fail(SDB_ERR_STATE, "%s", first.c_str());
  return SDB_OK;
}
SDB_API_CATCH("sdb_cluster_synchronize")

int sdb_cluster_allgather_merge(sdb_cluster *c, uint64_t ticket, uint64_t nq, uint32_t per_shard, void *block,
                                uint32_t limit, uint64_t *out_ids, float *out_dists, uint32_t *out_shards,
                                uint32_t *out_counts, int mem, void *stream_) try {
  if (!c) return fail(SDB_ERR_INVALID, "cluster is NULL");
  if (nq && (!block || !out_ids || !out_dists || !out_counts)) return fail(SDB_ERR_INVALID, "NULL argument");
  if (nq && per_shard == 0) return fail(SDB_ERR_INVALID, "per_shard must be positive");
  return collective(c, nullptr, ticket, nq, nullptr, per_shard, block, limit, 0, out_ids, out_dists, out_shards, out_counts,
                    mem, as_stream(stream_));
}
SDB_API_CATCH("sdb_cluster_allgather_merge")

int sdb_cluster_search_batch(sdb_cluster *c, sdb_index *ix, uint64_t ticket, uint64_t nq, const float *queries,
                             uint32_t limit, uint32_t search_size, uint64_t *out_ids, float *out_dists,
                             uint32_t *out_shards, uint32_t *out_counts, int mem, void *stream_) try {
  if (!c || !ix) return fail(SDB_ERR_INVALID, "NULL handle");
  const bool no_out = mem == SDB_MEM_HOST && !out_ids && !out_dists && !out_shards && !out_counts;  // answer not wanted here
  if (nq && (!queries || (!no_out && (!out_ids || !out_dists || !out_counts)))) return fail(SDB_ERR_INVALID, "NULL argument");
  return collective(c, ix, ticket, nq, queries, 0, nullptr, limit, search_size, out_ids, out_dists, out_shards, out_counts,
                    mem, as_stream(stream_));
}
SDB_API_CATCH("sdb_cluster_search_batch")

int sdb_cluster_stamp_block(void *block, uint64_t nq, uint32_t per_shard, uint32_t limit, uint32_t rank, uint64_t seq,
                            uint64_t ticket, uint32_t status, const float *queries, uint32_t dim, int device,
                            void *stream_) try {
  if (!block || per_shard == 0 || nq == 0) return fail(SDB_ERR_INVALID, "bad argument");
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  if (device < 0 || device >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", device);
  DeviceGuard dg(device);
  return stamp_tag(block, BlockLayout(nq, per_shard), nq, per_shard, limit, rank, seq, ticket, status, queries, dim,
                   as_stream(stream_));
}
SDB_API_CATCH("sdb_cluster_stamp_block")

int sdb_cluster_merge_gathered(uint32_t world, uint64_t nq, uint32_t per_shard, const void *gathered, uint32_t limit,
                               uint64_t *out_ids, float *out_dists, uint32_t *out_shards, uint32_t *out_counts,
                               int device, void *stream_) try {
  if (nq == 0) return SDB_OK;
  if (!gathered || !out_ids || !out_dists || !out_counts) return fail(SDB_ERR_INVALID, "NULL argument");
  SDB_TRY(check_merge_shape(world, per_shard, limit));
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  if (device < 0 || device >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", device);
  DeviceGuard dg(device);
  hipStream_t stream = as_stream(stream_);
  const BlockLayout bl(nq, per_shard);
  ExchangeVerdict *v = nullptr;
  SDB_HIP(hipHostMalloc(reinterpret_cast<void **>(&v), sizeof(ExchangeVerdict), hipHostMallocDefault));
  memset((void *)v, 0, sizeof(*v));
  const char *g = static_cast<const char *>(gathered);
  int rc = launch_topk_merge(world, nq, per_shard, g, bl.bytes, g + bl.off_d, bl.bytes, g + bl.off_c, bl.bytes, limit,
                             out_ids, out_dists, out_shards, out_counts, stream, g + bl.off_t, bl.bytes, v);
  if (rc == SDB_OK && hipStreamSynchronize(stream) != hipSuccess) rc = fail(SDB_ERR_DEVICE, "merge failed");
  if (rc == SDB_OK && v->state >= kVerdictMismatch) rc = fail(SDB_ERR_STATE, "%s", describe(*v, (int)world).c_str());
  else if (rc == SDB_OK && v->state != kVerdictOk) rc = fail(SDB_ERR_DEVICE, "the merge left no verdict");
  (void)hipHostFree((void *)v);
  return rc;
}
SDB_API_CATCH("sdb_cluster_merge_gathered")

}  // extern "C"
