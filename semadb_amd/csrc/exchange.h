// exchange.h -- what merge.hip and cluster.hip share about the shard exchange: the verdict a merge leaves behind
// after it has compared the tags of the gathered blocks (include/semadb_amd.h sdb_block_tag).
#pragma once
#include "common.h"

namespace sdb {

static_assert(sizeof(sdb_block_tag) == SDB_BLOCK_TAG_BYTES, "the tag is part of the all-gather message");

// which fields of a tag differ from rank 0's
enum : uint32_t {
  kTagMagic = 1u, kTagSeq = 2u, kTagTicket = 4u, kTagNq = 8u, kTagPerShard = 16u, kTagLimit = 32u,
  kTagQueryHash = 64u, kTagRank = 128u
};
enum : uint32_t { kVerdictNone = 0, kVerdictOk = 1, kVerdictMismatch = 2, kVerdictShardFailed = 3 };

// one per exchange in flight, in pinned host memory; written by workgroup 0 of the merge, read by the host after
// the exchange stream has passed the merge
struct ExchangeVerdict {
  volatile uint32_t state;  // kVerdict*
  uint32_t bad_rank;        // lowest rank whose tag differs from rank 0's or whose shard failed
  uint32_t fields;          // kTag* of every difference seen
  uint32_t status;          // sdb_status of a failed shard search (kVerdictShardFailed)
  uint32_t status_rank;     // whose
  uint32_t pad;
  uint64_t seq, ticket;     // of rank 0's tag
};

int launch_topk_merge(uint32_t n_shards, uint64_t nq, uint32_t per_shard, const void *ids, size_t ids_stride,
                      const void *dists, size_t dists_stride, const void *counts, size_t counts_stride, uint32_t limit,
                      uint64_t *out_ids, float *out_dists, uint32_t *out_shards, uint32_t *out_counts,
                      hipStream_t stream, const void *tags = nullptr, size_t tag_stride = 0,
                      ExchangeVerdict *verdict = nullptr);
int check_merge_shape(uint32_t n_shards, uint32_t per_shard, uint32_t limit);

}  // namespace sdb
