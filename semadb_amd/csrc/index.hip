// index.hip -- K3 (HBM slab + adjacency loader) and the host side of K2 (search_batch).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <new>
#include <stdexcept>

#include "pq.h"
#include "search_kernel.h"

namespace sdb {

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
char *last_error_buf() noexcept {
  static thread_local char e[kErrBytes] = {0};
  return e;
}

int fail(int code, const char *fmt, ...) noexcept {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), kErrBytes, fmt, ap);
  va_end(ap);
  return code;
}

// the one place where exceptions end (common.h SDB_API_CATCH)
int on_exception(const char *fn) noexcept {
  try {
    throw;
  } catch (const std::bad_alloc &) {
    return fail(SDB_ERR_DEVICE, "%s: out of host memory", fn);
  } catch (const std::exception &e) {
    return fail(SDB_ERR_STATE, "%s: %s", fn, e.what());
  } catch (...) {
    return fail(SDB_ERR_STATE, "%s: unknown C++ exception", fn);
  }
}

// ------------------------------------------------------------------------------------------
// workspaces
// ------------------------------------------------------------------------------------------
int Workspace::ensure_bitsets(size_t bytes) {
  if (bytes <= bitset_bytes) return SDB_OK;
  if (bitsets) SDB_HIP(hipFree(bitsets));
  bitsets = nullptr, bitset_bytes = 0;
  SDB_HIP(hipMalloc(&bitsets, bytes));
  bitset_bytes = bytes;
  return SDB_OK;
}

int Workspace::ensure_scratch(size_t bytes) {
  if (bytes <= scratch_bytes) return SDB_OK;
  if (scratch) SDB_HIP(hipFree(scratch));
  scratch = nullptr, scratch_bytes = 0;
  SDB_HIP(hipMalloc(&scratch, bytes));
  scratch_bytes = bytes;
  return SDB_OK;
}

int Workspace::ensure_filter(size_t bytes) {
  if (bytes <= filter_bytes) return SDB_OK;
  if (filter) SDB_HIP(hipFree(filter));
  filter = nullptr, filter_bytes = 0;
  SDB_HIP(hipMalloc(&filter, bytes));
  filter_bytes = bytes;
  return SDB_OK;
}

int Workspace::ensure_filter_aux(size_t bytes) {
  if (bytes <= filter_aux_bytes) return SDB_OK;
  if (filter_aux) SDB_HIP(hipFree(filter_aux));
  filter_aux = nullptr, filter_aux_bytes = 0;
  SDB_HIP(hipMalloc(&filter_aux, bytes));
  filter_aux_bytes = bytes;
  return SDB_OK;
}

int Workspace::ensure_filter_host(size_t bytes) {
  if (bytes <= filter_host_bytes) return SDB_OK;
  if (filter_host) SDB_HIP(hipHostFree(filter_host));
  filter_host = nullptr, filter_host_bytes = 0;
  bytes += bytes / 4;  // grows in steps: pinning is the expensive part
  SDB_HIP(hipHostMalloc(&filter_host, bytes, hipHostMallocDefault));
  filter_host_bytes = bytes;
  return SDB_OK;
}

int Workspace::ensure_lut(size_t bytes) {
  if (bytes <= lut_bytes) return SDB_OK;
  if (lut) SDB_HIP(hipFree(lut));
  lut = nullptr, lut_bytes = 0;
  SDB_HIP(hipMalloc(&lut, bytes));
  lut_bytes = bytes;
  return SDB_OK;
}

void Workspace::release() {
  if (filter) (void)hipFree(filter);
  filter = nullptr;
  if (filter_host) (void)hipHostFree(filter_host);
  filter_host = nullptr, filter_host_bytes = 0;
  if (filter_aux) (void)hipFree(filter_aux);
  filter_aux = nullptr, filter_aux_bytes = 0;
  if (lut) (void)hipFree(lut);
  lut = nullptr;
  if (bitsets) (void)hipFree(bitsets);
  if (scratch) (void)hipFree(scratch);
  if (own_stream) (void)hipStreamDestroy(own_stream);
  if (done) (void)hipEventDestroy(done);
  done = nullptr;
  if (launched) (void)hipEventDestroy(launched);
  launched = nullptr, launched_valid = false;
  bitsets = nullptr, scratch = nullptr, own_stream = nullptr;
}

// ------------------------------------------------------------------------------------------
// layout kernels
// ------------------------------------------------------------------------------------------
// original row-major [n][dim]  ->  slab rows [n][ld] (RowLayout in common.h)
__global__ void k_permute_rows(const float *__restrict__ src, float *__restrict__ dst, uint32_t n,
                               uint32_t dim, uint32_t nblk, uint32_t ng, uint32_t tail, uint32_t ld) {
  const uint32_t row = blockIdx.x;
  if (row >= n) return;
  const float *s = src + (size_t)row * dim;
  float *d = dst + (size_t)row * ld;
  for (uint32_t t = threadIdx.x; t < ld; t += blockDim.x) {
    float v = 0.0f;
    if (t < ng * 128) {
      uint32_t g = t / 128, r = t % 128, Lx = r / 4, k = r % 4, b = 4 * g + k;
      if (b < nblk) v = s[32 * b + Lx];
    } else {
      uint32_t i = t - ng * 128;
      if (i < tail) v = s[nblk * 32 + i];
    }
    d[t] = v;
  }
}

__global__ void k_unpermute_rows(const float *__restrict__ src, float *__restrict__ dst, uint32_t n,
                                 uint32_t dim, uint32_t nblk, uint32_t ng, uint32_t ld) {
  const uint32_t row = blockIdx.x;
  if (row >= n) return;
  const float *s = src + (size_t)row * ld;
  float *d = dst + (size_t)row * dim;
  for (uint32_t e = threadIdx.x; e < dim; e += blockDim.x) {
    uint32_t b = e / 32, Lx = e % 32;
    d[e] = b < nblk ? s[(b / 4) * 128 + Lx * 4 + (b % 4)] : s[ng * 128 + Lx];
  }
}

// slab rows by slot list -> original-layout rows (kNoSlot: a zero row)
__global__ void k_gather_rows(const float *__restrict__ slab, const uint32_t *__restrict__ slots, float *__restrict__ dst,
                              uint32_t n, uint32_t dim, uint32_t nblk, uint32_t ng, uint32_t ld) {
  const uint32_t i = blockIdx.x;
  if (i >= n) return;
  const uint32_t slot = slots[i];
  const float *s = slab + (size_t)slot * ld;
  float *d = dst + (size_t)i * dim;
  for (uint32_t e = threadIdx.x; e < dim; e += blockDim.x) {
    const uint32_t b = e / 32, Lx = e % 32;
    d[e] = slot == kNoSlot ? 0.0f : (b < nblk ? s[(b / 4) * 128 + Lx * 4 + (b % 4)] : s[ng * 128 + Lx]);
  }
}

// sdb_index_compact: new row j <- old row live[j], adjacency renumbered through map[]
__global__ __launch_bounds__(64) void k_compact_rows(const uint32_t *__restrict__ live, const uint32_t *__restrict__ map,
                                                     uint32_t ld, const float *__restrict__ slab, float *__restrict__ nslab,
                                                     const uint32_t *__restrict__ adj, uint32_t *__restrict__ nadj,
                                                     const float *__restrict__ adjdist, float *__restrict__ nadjdist,
                                                     const uint32_t *__restrict__ deg, uint32_t *__restrict__ ndeg,
                                                     const uint32_t *__restrict__ clean, uint32_t *__restrict__ nclean,
                                                     const uint32_t *__restrict__ dcount, uint32_t *__restrict__ ndcount,
                                                     const uint64_t *__restrict__ ids, uint64_t *__restrict__ nids,
                                                     const uint8_t *__restrict__ codes, uint8_t *__restrict__ ncodes,
                                                     uint32_t M, uint32_t *__restrict__ lost) {
  const uint32_t j = blockIdx.x, s = live[j];
  const int lane = threadIdx.x;
  const float4 *src = reinterpret_cast<const float4 *>(slab + (size_t)s * ld);
  float4 *dst = reinterpret_cast<float4 *>(nslab + (size_t)j * ld);
  for (uint32_t i = lane; i < ld / 4; i += 64) dst[i] = src[i];
  const uint32_t e = adj[(size_t)s * kAdjStride + lane];
  const uint32_t ne = e == kNoSlot ? kNoSlot : map[e];
  if (e != kNoSlot && ne == kNoSlot) atomicAdd(lost, 1u);  // an edge into a tombstone: delete_batch never leaves one
  nadj[(size_t)j * kAdjStride + lane] = ne;
  nadjdist[(size_t)j * kAdjStride + lane] = adjdist[(size_t)s * kAdjStride + lane];
  if (lane == 0) ndeg[j] = deg[s], nclean[j] = clean[s], ndcount[j] = dcount[s], nids[j] = ids[s];
  if (codes)
    for (uint32_t i = lane; i < M; i += 64) ncodes[(size_t)j * M + i] = codes[(size_t)s * M + i];
}

// The filter of a search is a set of node ids (a roaring bitmap in the reference, search.go:33-51,93); the walk wants
// slots: per query the slots of the ids that exist, ascending (Contains, :93), and how many of the first searchSize ids
// exist (the seeds, :41-48).  One wave per query resolves its ids in place -- the compacted slots at the start of the
// query's own segment.  MAP = false: a table whose ids are consecutive (id = base + slot: resolving is a subtraction,
// and ascending ids are ascending slots).  MAP = true: any other table -- a probe of the committed view's id -> slot
// table (sdb_index::IdMap); rows are appended in the order they arrive, so ascending ids USUALLY are ascending slots,
// and a query for which they are not raises flags[2]: the walk then answers Contains from the ids themselves
// (SearchArgs::filt_ids); the seeds are the first slots in id order either way.
// flags[0]: != 0 when some query's ids are not strictly ascending; flags[1]: one such query.
__device__ __forceinline__ uint32_t idmap_hash(uint64_t id) { return (uint32_t)((id * 0x9E3779B97F4A7C15ull) >> 32); }
__global__ void k_idmap_build(const uint64_t *__restrict__ ids, uint32_t n, uint64_t *__restrict__ keys,
                              uint32_t *__restrict__ vals, uint32_t mask) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint64_t id = ids[s];
  if (id == 0) return;  // tombstone
  uint32_t h = idmap_hash(id) & mask;
  while (true) {
    const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(keys + h), 0ull, (unsigned long long)id);
    if (old == 0ull || old == id) {
      vals[h] = s;
      return;
    }
    h = (h + 1) & mask;
  }
}
template <bool MAP>
__global__ __launch_bounds__(64) void k_filter_resolve(const uint64_t *__restrict__ ids, const uint32_t *__restrict__ off,
                                                       uint64_t base_id, uint32_t view_n, uint32_t search_size,
                                                       uint32_t *__restrict__ slots, uint32_t *__restrict__ fcnt,
                                                       uint32_t *__restrict__ scnt, uint32_t *__restrict__ flags,
                                                       const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                       uint32_t mask) {
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t b = off[q], e = off[q + 1];
  uint32_t pos = 0, ns = 0;
  bool bad = false, unsorted = false;
  int64_t last = -1;  // MAP: the slot of the last id that resolved (wave-uniform)
  for (uint32_t base = b; base < e; base += 64) {
    const uint32_t i = base + lane;
    const bool has = i < e;
    const uint64_t id = has ? ids[i] : 0;
    if (has && i > b && id <= ids[i - 1]) bad = true;
    uint64_t s = id - base_id;
    bool ok = has && id >= base_id && s < (uint64_t)view_n;  // rows past the committed count do not exist yet
    if constexpr (MAP) {
      ok = false;
      if (has && id != 0) {
        uint32_t h = idmap_hash(id) & mask;
        while (true) {
          const uint64_t k = keys[h];
          if (k == id) {
            s = vals[h], ok = true;
            break;
          }
          if (k == 0) break;
          h = (h + 1) & mask;
        }
      }
    }
    const uint64_t m = __ballot(ok);
    if constexpr (MAP) {
      if (m) {
        const uint64_t below = m & ((1ull << lane) - 1);
        const int prev_lane = below ? 63 - __clzll((long long)below) : lane;
        const uint32_t sp = (uint32_t)__shfl((int)(uint32_t)s, prev_lane, 64);
        const int64_t prev = below ? (int64_t)sp : last;
        if (ok && (int64_t)s <= prev) unsorted = true;
        last = (int64_t)(uint32_t)__shfl((int)(uint32_t)s, 63 - __clzll((long long)m), 64);
      }
    }
    if (ok) slots[b + pos + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = (uint32_t)s;
    if (base - b < search_size) {  // GetMany(first searchSize ids) skips the unknown ones (itemcache.go:109-128)
      const uint32_t left = search_size - (base - b);
      ns += (uint32_t)__popcll(left >= 64 ? m : (m & ((1ull << left) - 1)));
    }
    pos += (uint32_t)__popcll(m);
  }
  if (__ballot(bad) && lane == 0) atomicOr(flags, 1u), flags[1] = q;
  if (MAP && __ballot(unsorted) && lane == 0) flags[2] = 1u;
  if (lane == 0) fcnt[q] = pos, scnt[q] = ns;
}

// The same for a filter handed over as a BITMAP (sdb_index_search_batch_bitmap): bit i of query q's words is the id
// first_id[q] + i.  A roaring bitmap keeps its dense chunks in exactly this form, and at 100 000 ids out of a million it is
// an eighth of the bytes of the id list -- the upload is what a large filter costs.  Pass 1 (this kernel, one wave per
// query, a lane per 64-bit word): how many of its bits name rows that exist (fcnt) and how many of its FIRST searchSize
// bits do (scnt: the seeds are GetMany(first searchSize ids), unknown ids skipped, search.go:41-48).
__device__ __forceinline__ uint64_t bitmap_valid_mask(uint64_t id0, uint64_t base_id, uint32_t view_n) {
  // bits b of a word whose first id is id0 with base_id <= id0 + b < base_id + view_n  (ids wrap nowhere near 2^64 here)
  uint64_t m = ~0ull;
  if (id0 < base_id) {
    const uint64_t skip = base_id - id0;
    m = skip >= 64 ? 0ull : (m << skip);
  }
  const uint64_t end = base_id + view_n;  // first id past the table
  if (id0 >= end) return 0ull;
  const uint64_t room = end - id0;
  if (room < 64) m &= (1ull << room) - 1;
  return m;
}
__device__ __forceinline__ uint64_t lowest_set_bits(uint64_t w, uint32_t k) {  // the k lowest set bits of w (k < popc(w))
  uint64_t out = 0;
  for (uint32_t i = 0; i < k; i++) {
    const uint64_t b = w & (0 - w);
    out |= b, w ^= b;
  }
  return out;
}
template <bool ALL>  // ALL: every bit counts (the ids are resolved afterwards, k_filter_resolve<true>); scnt is not written
__global__ __launch_bounds__(64) void k_filter_bitmap_count(const uint64_t *__restrict__ words, const uint32_t *__restrict__ woff,
                                                            const uint64_t *__restrict__ first_id, uint64_t base_id,
                                                            uint32_t view_n, uint32_t search_size, uint32_t *__restrict__ fcnt,
                                                            uint32_t *__restrict__ scnt) {
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t b = woff[q], e = woff[q + 1];
  const uint64_t f0 = first_id[q];
  uint32_t total = 0, seeds = 0, seen = 0;  // seen: set bits before this round of 64 words
  for (uint32_t base = b; base < e; base += 64) {
    const uint32_t w = base + lane;
    const uint64_t word = w < e ? words[w] : 0ull;
    const uint64_t valid = ALL ? word : (word & bitmap_valid_mask(f0 + (uint64_t)(w - b) * 64u, base_id, view_n));
    uint32_t pc = (uint32_t)__popcll(word), incl = pc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {  // inclusive scan of the words' populations over the lanes
      const uint32_t up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    const uint32_t before = seen + incl - pc;  // set bits of the filter in front of this word
    if (before < search_size) {
      const uint32_t left = search_size - before;
      seeds += (uint32_t)__popcll(left >= pc ? valid : (valid & lowest_set_bits(word, left)));
    }
    total += (uint32_t)__popcll(valid);
    seen += __shfl(incl, 63, 64);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o, 64), seeds += __shfl_down(seeds, o, 64);
  if (lane == 0) {
    fcnt[q] = total;
    if (!ALL) scnt[q] = seeds;
  }
}
// exclusive scan of the per-query counts -> where each query's slot list starts (one workgroup; nq is a batch size)
__global__ __launch_bounds__(1024) void k_filter_offsets(const uint32_t *__restrict__ cnt, uint32_t nq, uint32_t *__restrict__ off,
                                                          uint32_t *__restrict__ overflow) {
  __shared__ uint64_t s_part[1024];
  const uint32_t t = threadIdx.x, per = (nq + 1023) / 1024;
  uint64_t sum = 0;
  for (uint32_t i = t * per; i < min(nq, (t + 1) * per); i++) sum += cnt[i];
  s_part[t] = sum;
  __syncthreads();
  if (t == 0) {
    uint64_t run = 0;
    for (uint32_t i = 0; i < 1024; i++) {
      const uint64_t v = s_part[i];
      s_part[i] = run, run += v;
    }
    off[nq] = (uint32_t)run;
    if (run > 0xFFFFFFFFull) *overflow = 1u;
  }
  __syncthreads();
  uint64_t run = s_part[t];
  for (uint32_t i = t * per; i < min(nq, (t + 1) * per); i++) off[i] = (uint32_t)run, run += cnt[i];
}
// pass 2: the slots, ascending, at the query's place in the list
template <bool IDS>  // IDS: every bit, as the id it stands for (64-bit), for k_filter_resolve<true>
__global__ __launch_bounds__(64) void k_filter_bitmap_expand(const uint64_t *__restrict__ words, const uint32_t *__restrict__ woff,
                                                             const uint64_t *__restrict__ first_id, uint64_t base_id,
                                                             uint32_t view_n, const uint32_t *__restrict__ off,
                                                             uint32_t *__restrict__ slots, uint64_t *__restrict__ ids_out) {
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t b = woff[q], e = woff[q + 1];
  const uint64_t f0 = first_id[q];
  uint32_t pos = off[q];
  for (uint32_t base = b; base < e; base += 64) {
    const uint32_t w = base + lane;
    const uint64_t id0 = f0 + (uint64_t)(w - b) * 64u;
    uint64_t valid = w < e ? (IDS ? words[w] : (words[w] & bitmap_valid_mask(id0, base_id, view_n))) : 0ull;
    const uint32_t pc = (uint32_t)__popcll(valid);
    uint32_t incl = pc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    uint32_t at = pos + incl - pc;
    while (valid) {
      const int bit = __ffsll((unsigned long long)valid) - 1;
      valid &= valid - 1;
      if constexpr (IDS) ids_out[at++] = id0 + (uint64_t)bit;
      else slots[at++] = (uint32_t)(id0 + (uint64_t)bit - base_id);
    }
    pos += __shfl(incl, 63, 64);
  }
}

__global__ void k_fill_u32(uint32_t *p, uint32_t v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------
// search launcher
// ------------------------------------------------------------------------------------------
// LDS hash visited set: plain store with the query in registers, unfiltered, reference-range search size.
// Any batch size: four resident waves per CU already saturate the memory system with this kernel, and a long
// batch keeps them resident back to back (batch 16 384: 1.18 M QPS, 7.3 TB/s, against 0.88 M QPS on the
// bitset variant at 16 waves per CU).
// the multi-wave quantized walk (search_kernel.h PQWideDist): which instantiation serves (M, K), or -1
static int pq_wide_shape(const SearchArgs &a) {
  if (!a.pq_codes || a.pq_narrow == 1 || a.filt_off || a.prefer_bitset) return -1;
  if (a.search_size > 96 || a.pq_K > 256 || a.pq_K % 32) return -1;
  switch (a.pq_M) {
    case 128: return 1;  // NL 15, RT 17 (two queries per CU)
    case 192: return 2;  // NL 15, RT 33 (two queries per CU)
    case 256: return 3;  // NL 32, RT 32
    case 384: return 4;  // NL 15, RT 33, eight waves
    default: return -1;  // M <= 64: the table fits beside a one-wave walk
  }
}

// SPLIT (the candidate array in a helper wave, search_kernel.h pqw_split_walker): plain searches on a start node
// without an overflow list; SDB_TUNE_PQ_NARROW = 3 keeps the walker-does-everything form for comparison
template <int NL, int RT, int W, int NLW, bool SPLIT>
static int launch_pqw_form(const SearchArgs &a, uint32_t nq, hipStream_t stream) {
  const bool h16 = (uint64_t)a.words_per_query * 32 <= (1u << 24) && !a.wide_hash;
  const size_t lut = (size_t)((W - 1) * NL + NLW) * a.pq_K * sizeof(float);
  static std::atomic<uint64_t> at16{0}, at32{0};
  if (h16) {
    const size_t lds = HashVisited16::kWords * 4 + sizeof(PQWideShared) + lut;
    if (first_use_on_this_device(at16))
      SDB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_greedy_search_pqw<NL, RT, kHash16, W, NLW, SPLIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    hipLaunchKernelGGL((k_greedy_search_pqw<NL, RT, kHash16, W, NLW, SPLIT>), dim3(nq), dim3(64 * W), lds, stream, a);
  } else {
    const size_t lds = HashVisited<kHashCapPQ>::kWords * 4 + sizeof(PQWideShared) + lut;
    if (first_use_on_this_device(at32))
      SDB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_greedy_search_pqw<NL, RT, kHashCapPQ, W, NLW, SPLIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    hipLaunchKernelGGL((k_greedy_search_pqw<NL, RT, kHashCapPQ, W, NLW, SPLIT>), dim3(nq), dim3(64 * W), lds, stream, a);
  }
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}
template <int NL, int RT, int W = 4, int NLW = NL>
static int launch_pqw(const SearchArgs &a, uint32_t nq, hipStream_t stream) {
  if constexpr (W == 4 && NL == 15)  // M = 128 / 192, two queries per CU (M = 192 at 2M x 768: 1.050 -> 0.989 ms per batch)
    if (a.pq_narrow != 3 && !a.vis_slots && !a.dcache && !a.start_ext_n) return launch_pqw_form<NL, RT, W, NLW, true>(a, nq, stream);
  return launch_pqw_form<NL, RT, W, NLW, false>(a, nq, stream);
}

bool search_uses_hash(const SearchArgs &a, uint32_t nq) {
  // filtered searches too (round 3): their search set takes the same table, their result set a 4 KB one beside it
  // (search_kernel.h kHashCapResult) -- no 2 x 128 MB bitset clear per batch, no HBM atomic per edge
  if (a.prefer_bitset) return false;
  if (a.search_size > 96) return false;
  // quantized store: only with the small LUT of M*K <= 2048 entries next to the table (M = 32's 32 KB LUT leaves
  // room for three walks per CU beside the 16 KB table: 0.95 M QPS against 1.31 M on the bitset at five)
  if (a.pq_codes) return pq_wide_shape(a) >= 0 || (a.pq_lut_in_lds && (size_t)a.pq_M * a.pq_K <= 2048);
  switch (a.ng) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 12: case 16: case 24: return true;
    default: return false;
  }
}

// The quantized walk with its table in LDS, on two waves per query (search_kernel.h k_greedy_search_pq2: the walker and
// the merger): plain unfiltered searches of the reference's searchSize range; a build's searches (visit log), a start
// node with an overflow list and SDB_TUNE_PQ_NARROW = 1 keep the one-wave kernel.
static bool pq_two_waves(const SearchArgs &a, uint32_t nq) {
  if (!a.pq_codes || !a.pq_lut_in_lds || (size_t)a.pq_M * a.pq_K > 2048 || a.pq_narrow == 1) return false;
  if (a.filt_off || a.vis_slots || a.dcache || a.start_ext_n || a.search_size > 96) return false;
  return search_uses_hash(a, nq);
}
template <uint32_t HCAP>
static int launch_pq2(const SearchArgs &a, uint32_t nq, hipStream_t stream) {
  constexpr uint32_t vis = HCAP == kHash16 ? HashVisited16::kWords : HashVisited<HCAP == kHash16 ? 4u : HCAP>::kWords;
  const size_t lds = ((size_t)vis + (((size_t)a.pq_M * a.pq_K + 3) & ~(size_t)3) + kPq2SharedWords) * sizeof(uint32_t);
  hipLaunchKernelGGL((k_greedy_search_pq2<HCAP>), dim3(nq), dim3(128), lds, stream, a);
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}

// HCAP: 0 = bitset, else the LDS hash set's capacity (it precedes the policy's own LDS of `lds` bytes)
template <class Dist, uint32_t HCAP>
static int launch_nreg(const SearchArgs &a, uint32_t nq, hipStream_t stream, size_t lds) {
  const bool filt = a.filt_off != nullptr;
  const size_t rwords = filt ? HashVisited<kHashCapResult>::kWords * sizeof(uint32_t) : 0;
  if constexpr (HCAP == kHash16) {
    const size_t total = HashVisited16::kWords * sizeof(uint32_t) + rwords + lds;
    if (filt) hipLaunchKernelGGL((k_greedy_search<Dist, 2, true, HCAP>), dim3(nq), dim3(64), total, stream, a);
    else hipLaunchKernelGGL((k_greedy_search<Dist, 2, false, HCAP>), dim3(nq), dim3(64), total, stream, a);
  } else if constexpr (HCAP != 0) {
    const size_t total = HashVisited<HCAP>::kWords * sizeof(uint32_t) + rwords + lds;
    if (filt) hipLaunchKernelGGL((k_greedy_search<Dist, 2, true, HCAP>), dim3(nq), dim3(64), total, stream, a);
    else hipLaunchKernelGGL((k_greedy_search<Dist, 2, false, HCAP>), dim3(nq), dim3(64), total, stream, a);
  } else if (a.search_size <= 128) {
    if (filt) hipLaunchKernelGGL((k_greedy_search<Dist, 2, true, 0>), dim3(nq), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((k_greedy_search<Dist, 2, false, 0>), dim3(nq), dim3(64), lds, stream, a);
  } else {
    if (filt) hipLaunchKernelGGL((k_greedy_search<Dist, 8, true, 0>), dim3(nq), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((k_greedy_search<Dist, 8, false, 0>), dim3(nq), dim3(64), lds, stream, a);
  }
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}

// The workgroup-per-query walk (search_kernel.h PlainWideDist) for calls with few queries: a 256-query call is one
// workgroup per CU.  Waves per workgroup, measured at 1M x 384 (tools/bench_latency.py, whole call of 1 / 256 queries;
// one wave per query: 0.538 / 0.598 ms): 2 waves 0.78 / 0.86, 4 waves 0.536 / 0.607, 8 waves 0.410 / 0.469, 16 waves
// 0.361 / 0.416 ms -- a hop's ~50 rows are 2 pairs per wave then, one short burst of loads each.  With the helpers
// pulling the likely next hop's rows through L2 (PlainWideDist::pull_ahead): 0.336 / 0.419 ms; 0.371 at 128 queries
// (0.408 without); with the helpers COMPUTING that hop's distances ahead (compute_ahead): 0.316 / 0.405 ms, 0.362 at 128.
// Past 256 queries eight waves per query (0.58 ms at 512 against 0.66 for one wave per query, 0.80 for sixteen).  Plain store, query in registers; the build's warm-up rounds (up to 512 points) included.
constexpr uint32_t kWideMaxQueries = 256;
#ifndef SDB_WIDE_AHEAD
#define SDB_WIDE_AHEAD 256
#endif
constexpr uint32_t kWideAheadQueries = SDB_WIDE_AHEAD;
#ifndef SDB_WIDE_WAVES
#define SDB_WIDE_WAVES 16
#endif
constexpr int kWideWaves = SDB_WIDE_WAVES;
static bool wide_walk(const SearchArgs &a, uint32_t nq) {
  // (the build's warm-up rounds come here too -- rounds of up to 512 points while the graph is small, ~330 of a 1M
  // build's ~580 rounds: their visit logs and distance tables are written by the walker like any other wave's)
  if (a.wide_mode == 1 || a.pq_codes || !search_uses_hash(a, nq)) return false;
  // 257 .. 512 queries (eight waves per query): faster than one wave per query for rows of up to 384 floats only
  // (d = 384: 0.56 against 0.65 ms; 512: 0.98 against 0.78; 768: 1.20 against 1.05; profiles/r05_latency.json)
  return a.wide_mode == 2 || nq <= (a.ng <= 3 ? 2 : 1) * kWideMaxQueries;
}
template <int NG, bool L2, int W>
static int launch_wide_w(const SearchArgs &a, uint32_t nq, hipStream_t stream) {
  using D = PlainWideDist<NG, L2, W>;
  const size_t lds = HashVisited<kHashCap>::kWords * sizeof(uint32_t) + D::kLdsBytes;
  SearchArgs b = a;
  // calls of up to 256 queries (a workgroup per CU at most): the other waves work ahead on the row the walk expands next
  b.wide_pull = ((W == 16 || NG == 8) && nq <= kWideAheadQueries) ? 2u : 0u;
  if (a.filt_off)  // the filtered walk (search.go:33-51,93-95): a hybrid REST query is one query with a filter
    hipLaunchKernelGGL((k_greedy_search_wide<NG, L2, W, true>), dim3(nq), dim3(64 * W),
                       lds + HashVisited<kHashCapResult>::kWords * sizeof(uint32_t), stream, b);
  else
    hipLaunchKernelGGL((k_greedy_search_wide<NG, L2, W, false>), dim3(nq), dim3(64 * W), lds, stream, b);
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}
// up to 256 queries: 16 waves per query (one workgroup per CU); up to 512: 8 waves (two per CU: 0.56 ms per call against
// 0.66 for one wave per query and 0.78 for 16 waves)
// rows of 1 024 floats: at sixteen waves (128 registers each) the query and two pairs of rows do not fit -- 17 .. 57
// registers spilled -- so every small call of that shape takes the eight-wave workgroup (214 registers, none spilled)
template <int NG, bool L2>
static int launch_wide(const SearchArgs &a, uint32_t nq, hipStream_t stream) {
  if constexpr (NG != 8)
    if (nq <= kWideMaxQueries || kWideWaves != 16) return launch_wide_w<NG, L2, kWideWaves>(a, nq, stream);
  return launch_wide_w<NG, L2, 8>(a, nq, stream);
}

template <int NG, bool L2>
static int launch_plain(const SearchArgs &a, uint32_t nq, hipStream_t stream) {
  if constexpr (NG >= 1 && NG <= 8)
    if (wide_walk(a, nq)) return launch_wide<NG, L2>(a, nq, stream);
  if (search_uses_hash(a, nq)) {
    // the two-precision hop (SearchArgs::sketch): batch walks, plain searches
    if constexpr (NG == 1 || NG == 2 || NG == 3 || NG == 4 || NG == 6)
      if (a.sketch && !a.filt_off && !a.vis_slots && !a.dcache && a.tail == 0 && a.search_size <= 128) {
        // (few float32 rows survive the first stage: four pairs of them in flight per round leave the registers to the float16 rows)
        using SkDist = PlainDist<NG, L2, true, 4, true>;
        hipLaunchKernelGGL((k_greedy_search<SkDist, 2, false, kHashCap>), dim3(nq), dim3(64),
                           HashVisited<kHashCap>::kWords * sizeof(uint32_t) + SkDist::kLdsBytes, stream, a);
        SDB_HIP(hipGetLastError());
        return SDB_OK;
      }
    return launch_nreg<PlainDist<NG, L2, true>, kHashCap>(a, nq, stream, PlainDist<NG, L2, true>::kLdsBytes);
  }
  return launch_nreg<PlainDist<NG, L2, false>, 0>(a, nq, stream, PlainDist<NG, L2, false>::kLdsBytes);
}

template <bool L2>
static int launch_ng(const SearchArgs &a, uint32_t nq, hipStream_t stream) {
  const size_t lds = (size_t)(a.ng * 128 + 32) * sizeof(float);
  switch (a.ng) {
    case 0: return launch_plain<0, L2>(a, nq, stream);
    case 1: return launch_plain<1, L2>(a, nq, stream);
    case 2: return launch_plain<2, L2>(a, nq, stream);
    case 3: return launch_plain<3, L2>(a, nq, stream);
    case 4: return launch_plain<4, L2>(a, nq, stream);
    case 6: return launch_plain<6, L2>(a, nq, stream);
    case 8: return launch_plain<8, L2>(a, nq, stream);
    case 12: return launch_plain<12, L2>(a, nq, stream);  // 1536
    case 16: return launch_plain<16, L2>(a, nq, stream);  // 2048
    case 24: return launch_plain<24, L2>(a, nq, stream);  // 3072
    default: return launch_nreg<PlainDist<-1, L2>, false>(a, nq, stream, lds);
  }
}

int launch_greedy_search(const SearchArgs &a_in, uint32_t nq, hipStream_t stream) {
  if (nq == 0) return SDB_OK;
  SearchArgs a = a_in;
  if (a.hash_limit == 0 || a.hash_limit > kHashLimit) a.hash_limit = kHashLimit;
  if (a.search_size == 0 || a.search_size > 512)
    return fail(SDB_ERR_INVALID, "searchSize %u not supported on device (1..512)", a.search_size);
  if (a.pq_codes) {  // fitted product quantizer attached (product.go:250-277)
    switch (pq_wide_shape(a)) {  // tables too large to sit beside a one-wave walk: one query per four waves
      case 1: return a.pq_narrow == 2 ? launch_pqw<32, 0>(a, nq, stream) : launch_pqw<15, 17>(a, nq, stream);  // two per CU
      // M = 192: 15 tables per wave in LDS and 33 in registers leave room for TWO queries per CU (643 k against
      // 570 k QPS at 10M x 768 with 32 + 16, one query per CU: profiles/r03_c4_10Mx768_pq.log); SDB_TUNE_PQ_NARROW = 2
      // selects the latter for comparison
      case 2: return a.pq_narrow == 2 ? launch_pqw<32, 16>(a, nq, stream) : launch_pqw<15, 33>(a, nq, stream);
      case 3: return launch_pqw<32, 32>(a, nq, stream);
      // M = 384: eight waves with M = 192's per-wave layout, the walker 24 + 24 (129 tables in LDS: with the 32-bit-cell
      // visited set 162 064 of the 162 816 bytes a workgroup may ask for)
      case 4: return launch_pqw<15, 33, 8, 24>(a, nq, stream);
      default: break;
    }
    const size_t lds = a.pq_lut_in_lds ? (size_t)a.pq_M * a.pq_K * sizeof(float) : 0;
    if (pq_two_waves(a, nq)) {
      if ((uint64_t)a.words_per_query * 32 <= (1u << 24) && !a.wide_hash) return launch_pq2<kHash16>(a, nq, stream);
      return launch_pq2<kHashCapPQ>(a, nq, stream);
    }
    if (search_uses_hash(a, nq)) {
      // up to 2^24 rows: the 16-bit-cell set (16 KB, six walks per CU); beyond: 32-bit cells
      if ((uint64_t)a.words_per_query * 32 <= (1u << 24) && !a.wide_hash) return launch_nreg<PQDist, kHash16>(a, nq, stream, lds);
      return launch_nreg<PQDist, kHashCapPQ>(a, nq, stream, lds);
    }
    return launch_nreg<PQDist, 0>(a, nq, stream, lds);
  }
  if (a.metric == SDB_METRIC_EUCLIDEAN) return launch_ng<true>(a, nq, stream);
  return launch_ng<false>(a, nq, stream);
}

}  // namespace sdb

using namespace sdb;

// ------------------------------------------------------------------------------------------
// sdb_index methods
// ------------------------------------------------------------------------------------------
int64_t sdb_index::slot_of(uint64_t id) const {
  if (n == 0) return -1;
  if (dense_ids) {
    uint64_t base = h_ids[0];
    if (id < base || id - base >= n) return -1;
    return (int64_t)(id - base);
  }
  auto it = id2slot.find(id);
  // (an insert in progress has entered its ids at the slots they will have: rows past n are not there yet)
  return it == id2slot.end() || it->second >= n ? -1 : (int64_t)it->second;
}

int sdb_index::reserve(uint32_t rows) {
  if (rows <= cap) return SDB_OK;
  uint32_t ncap = cap ? cap : 1024;
  while (ncap < rows) ncap = ncap < (1u << 30) ? ncap * 2 : rows;
  // every new buffer first; if one allocation fails the ones before it are returned and the index is as it was
  struct Fresh {
    std::vector<void *> p;
    bool keep = false;
    ~Fresh() {
      if (!keep)
        for (void *x : p)
          if (x) (void)hipFree(x);
    }
    int get(void **out, size_t bytes) {
      SDB_HIP(hipMalloc(out, bytes));
      p.push_back(*out);
      return SDB_OK;
    }
    // two buffers of a CACHE (the neighbours' code rows behind both adjacency copies): both or neither, and no room
    // for them is no error -- the index works without (searches gather code rows by slot)
    bool get_pair_if_room(void **a, void **b, size_t bytes) {
      *a = *b = nullptr;
      if (hipMalloc(a, bytes) == hipSuccess && hipMalloc(b, bytes) == hipSuccess) {
        p.push_back(*a), p.push_back(*b);
        return true;
      }
      (void)hipGetLastError();
      if (*a) (void)hipFree(*a);
      *a = *b = nullptr;
      return false;
    }
  } fresh;
  float *nslab = nullptr, *nad = nullptr;
  uint32_t *nadj = nullptr, *nradj = nullptr, *ndeg = nullptr, *nclean = nullptr, *ndc = nullptr;
  uint64_t *nids = nullptr, *nrids = nullptr;
  uint8_t *ndirty = nullptr, *ncodes = nullptr;
  SDB_TRY(fresh.get((void **)&nslab, (size_t)ncap * lay.ld * sizeof(float)));
  SDB_TRY(fresh.get((void **)&nadj, (size_t)ncap * kAdjStride * sizeof(uint32_t)));
  SDB_TRY(fresh.get((void **)&nradj, (size_t)ncap * kAdjStride * sizeof(uint32_t)));
  SDB_TRY(fresh.get((void **)&ndeg, (size_t)ncap * sizeof(uint32_t)));
  SDB_TRY(fresh.get((void **)&nclean, (size_t)ncap * sizeof(uint32_t)));
  SDB_TRY(fresh.get((void **)&nids, (size_t)ncap * sizeof(uint64_t)));
  SDB_TRY(fresh.get((void **)&nrids, (size_t)ncap * sizeof(uint64_t)));
  SDB_TRY(fresh.get((void **)&ndirty, (size_t)ncap));
  SDB_TRY(fresh.get((void **)&nad, (size_t)ncap * kAdjStride * sizeof(float)));  // edge-distance cache of the write path (index.h)
  SDB_TRY(fresh.get((void **)&ndc, (size_t)ncap * sizeof(uint32_t)));
  if (pq) SDB_TRY(fresh.get((void **)&ncodes, (size_t)ncap * pq->M));  // the code rows of a quantized store grow with it
  uint8_t *nacw = nullptr, *nacr = nullptr;  // ... and the neighbours' code rows behind both adjacency copies
  const bool had_ac = has_adjcodes();
  size_t ac_row = had_ac ? (size_t)kAdjStride * pq->M : 0;
  // (2 x 64 M bytes per row: 41 GB at 10M rows and M = 32, more than the vectors of d = 768.  A table that grows past
  // the room for them drops them instead of failing the insert -- alloc_adjcodes treats them as optional the same way)
  if (ac_row && !fresh.get_pair_if_room((void **)&nacw, (void **)&nacr, (size_t)ncap * ac_row)) ac_row = 0;
  // the old buffers are freed below: nothing may still be walking them (searches run on streams of their own)
  if (cap) SDB_HIP(hipDeviceSynchronize());
  SDB_HIP(hipMemset(nadj, 0xFF, (size_t)ncap * kAdjStride * sizeof(uint32_t)));
  SDB_HIP(hipMemset(nradj, 0xFF, (size_t)ncap * kAdjStride * sizeof(uint32_t)));
  SDB_HIP(hipMemset(ndeg, 0, (size_t)ncap * sizeof(uint32_t)));
  SDB_HIP(hipMemset(nclean, 0, (size_t)ncap * sizeof(uint32_t)));  // loaded / appended edges are not "clean"
  SDB_HIP(hipMemset(ndirty, 0, (size_t)ncap));
  if (n) {
    SDB_HIP(hipMemcpy(nslab, d_slab, (size_t)n * lay.ld * sizeof(float), hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemcpy(nadj, d_adj, (size_t)n * kAdjStride * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemcpy(nradj, r_adj, (size_t)n * kAdjStride * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemcpy(ndeg, d_deg, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemcpy(nclean, d_clean, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemcpy(nids, d_ids, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemcpy(nrids, r_ids, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemcpy(ndirty, d_dirty, (size_t)n, hipMemcpyDeviceToDevice));
  }
  SDB_HIP(hipMemset(ndc, 0, (size_t)ncap * sizeof(uint32_t)));  // a row without cached distances has d_dcount 0
  if (n) {
    SDB_HIP(hipMemcpy(nad, d_adjdist, (size_t)n * kAdjStride * sizeof(float), hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemcpy(ndc, d_dcount, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    if (pq) SDB_HIP(hipMemcpy(ncodes, d_codes, (size_t)n * pq->M, hipMemcpyDeviceToDevice));
    if (ac_row) {
      SDB_HIP(hipMemcpy(nacw, d_adjcodes, (size_t)n * ac_row, hipMemcpyDeviceToDevice));
      SDB_HIP(hipMemcpy(nacr, r_adjcodes, (size_t)n * ac_row, hipMemcpyDeviceToDevice));
    }
  }
  SDB_HIP(hipDeviceSynchronize());
  {
    std::unique_lock<sdb::ViewMutex> wl(view_mu);  // searches pick their pointers up under this lock
    for (void *p : {(void *)d_slab, (void *)d_adj, (void *)r_adj, (void *)d_deg, (void *)d_clean, (void *)d_ids,
                    (void *)r_ids, (void *)d_dirty, (void *)d_adjdist, (void *)d_dcount})
      if (p) (void)hipFree(p);
    if (pq && d_codes) (void)hipFree(d_codes);
    if (had_ac) (void)hipFree(d_adjcodes), (void)hipFree(r_adjcodes), d_adjcodes = nacw, r_adjcodes = nacr;  // (NULL: dropped)
    d_slab = nslab, d_adj = nadj, r_adj = nradj, d_deg = ndeg, d_clean = nclean, d_ids = nids, r_ids = nrids;
    d_dirty = ndirty, d_adjdist = nad, d_dcount = ndc;
    if (pq) d_codes = ncodes;
    fresh.keep = true;
    cap = ncap;
    view.adj = r_adj, view.ids = r_ids, view.adj_codes = r_adjcodes;
    view_gen++;
  }
  return SDB_OK;
}

// ------------------------------------------------------------------------------------------
// graph versions
// ------------------------------------------------------------------------------------------
namespace sdb {
// rows the transaction wrote (dirty flags) and rows it appended: committed copy -> the writer's stale copy; with the
// neighbours' code rows behind them (index.h d_adjcodes; code_bytes = 64 M per row, 0: none)
__global__ void k_sync_rows(const uint32_t *__restrict__ src_adj, uint32_t *__restrict__ dst_adj,
                            const uint64_t *__restrict__ src_ids, uint64_t *__restrict__ dst_ids,
                            uint8_t *__restrict__ dirty, uint32_t n, uint32_t first_new,
                            const uint8_t *__restrict__ src_codes, uint8_t *__restrict__ dst_codes, uint32_t code_bytes) {
  const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  if (row < first_new && !dirty[row]) return;
  dst_adj[(size_t)row * kAdjStride + lane] = src_adj[(size_t)row * kAdjStride + lane];
  if (code_bytes) {  // a multiple of 64: whole words when M is a multiple of four, bytes otherwise
    if ((code_bytes & 255u) == 0) {
      const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src_codes + (size_t)row * code_bytes);
      uint32_t *d4 = reinterpret_cast<uint32_t *>(dst_codes + (size_t)row * code_bytes);
      for (uint32_t i = lane; i < code_bytes / 4; i += 64) d4[i] = s4[i];
    } else {
      for (uint32_t i = lane; i < code_bytes; i += 64) dst_codes[(size_t)row * code_bytes + i] = src_codes[(size_t)row * code_bytes + i];
    }
  }
  if (lane == 0) dst_ids[row] = src_ids[row], dirty[row] = 0;
}
// The neighbours' code rows behind the adjacency rows (index.h d_adjcodes): out[row][e] = codes[adj[row][e]], zeros behind
// the edges.  dirty != NULL: only the rows a transaction wrote or appended (first_new: rows at its start).
__global__ __launch_bounds__(256) void k_adjcodes_rows(const uint32_t *__restrict__ adj, const uint8_t *__restrict__ codes,
                                                       uint8_t *__restrict__ out, const uint8_t *__restrict__ dirty, uint32_t n,
                                                       uint32_t first_new, uint32_t M) {
  const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  if (dirty && row < first_new && !dirty[row]) return;
  const uint32_t nb = adj[(size_t)row * kAdjStride + lane];
  const size_t at = ((size_t)row * kAdjStride + lane) * M;
  if ((M & 7u) == 0) {
    const uint2 *src = reinterpret_cast<const uint2 *>(codes + (size_t)(nb == kNoSlot ? 0u : nb) * M);
    uint2 *dst = reinterpret_cast<uint2 *>(out + at);
    for (uint32_t i = 0; i < M / 8; i++) dst[i] = nb == kNoSlot ? make_uint2(0u, 0u) : src[i];
  } else {
    for (uint32_t i = 0; i < M; i++) out[at + i] = nb == kNoSlot ? (uint8_t)0 : codes[(size_t)nb * M + i];
  }
}
// sdb_index_abort_write: the rows the transaction wrote take the committed copy back, the rows it appended become
// empty again.  What only the write path keeps per row -- degree, clean prefix, cached edge distances -- is rebuilt
// conservatively: the degree from the row, no clean prefix, no cached distances (both only ever save work: a prune
// that finds none evaluates everything and arrives at the same row, build.hip).
__global__ void k_restore_rows(const uint32_t *__restrict__ c_adj, uint32_t *__restrict__ w_adj,
                               const uint64_t *__restrict__ c_ids, uint64_t *__restrict__ w_ids, uint32_t *__restrict__ deg,
                               uint32_t *__restrict__ clean, uint32_t *__restrict__ dcount, uint8_t *__restrict__ dirty,
                               uint32_t n_now, uint32_t n_committed) {
  const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_now) return;
  const bool appended = row >= n_committed;
  if (!appended && !dirty[row]) return;
  const uint32_t e = appended ? kNoSlot : c_adj[(size_t)row * kAdjStride + lane];
  w_adj[(size_t)row * kAdjStride + lane] = e;
  const uint32_t d = (uint32_t)__popcll(__ballot(e != kNoSlot));
  if (lane == 0) {
    w_ids[row] = appended ? 0ull : c_ids[row];
    deg[row] = d, clean[row] = 0, dcount[row] = 0, dirty[row] = 0;
  }
}
// test support: rows on which the two copies differ
__global__ void k_count_version_diff(const uint32_t *a_adj, const uint32_t *b_adj, const uint64_t *a_ids,
                                     const uint64_t *b_ids, uint32_t n, unsigned long long *out) {
  const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const bool diff = a_adj[(size_t)row * kAdjStride + lane] != b_adj[(size_t)row * kAdjStride + lane] ||
                    (lane == 0 && a_ids[row] != b_ids[row]);
  if (__ballot(diff) && lane == 0) atomicAdd(out, 1ull);
}
}  // namespace sdb

int sdb_index::begin_write() {
  if (!in_tx) in_tx = true, tx_n0 = n, tx_dead0 = n_dead, tx_max_id0 = max_node_id, tx_dirty = false;
  return SDB_OK;
}

// ---- the float16 copy of the slab (two-precision hop) ----------------------------------------------------------
namespace sdb {
// one wave per row: element t of the row -> half t of the copy (sk_half: round to nearest, no denormals), and the
// row's ||y - y16||, ||y16|| into the table-wide maxima (non-negative floats order like their bit patterns; a NaN's
// pattern is above every number's, so a row with a NaN makes the bound NaN and the stage discards nothing)
__global__ __launch_bounds__(256) void k_sketch_rows(const float *__restrict__ slab, uint16_t *__restrict__ sk,
                                                      float *__restrict__ sk_norm, uint32_t n, uint32_t ld,
                                                      uint32_t *__restrict__ stats) {
  const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float *s = slab + (size_t)row * ld;
  uint16_t *d = sk + (size_t)row * ld;
  double e2 = 0.0, n2 = 0.0;
  for (uint32_t t = lane; t < ld; t += 64) {
    const float v = s[t];
    const _Float16 h = sk_half(v);
    d[t] = __builtin_bit_cast(uint16_t, h);
    const double dv = (double)v - (double)(float)h, hv = (double)(float)h;
    e2 += dv * dv, n2 += hv * hv;
  }
  for (int o = 32; o; o >>= 1) e2 += __shfl_xor(e2, o), n2 += __shfl_xor(n2, o);
  if (lane == 0) {
    sk_norm[row] = (float)n2;  // ||y16||^2 (the sum in double: one rounding)
    const float e = (float)(sqrt(e2) * 1.0001), y = (float)(sqrt(n2) * 1.0001);  // rounded up past their own rounding
    atomicMax(stats, __float_as_uint(e < 0.0f ? 0.0f : e));
    atomicMax(stats + 1, __float_as_uint(y < 0.0f ? 0.0f : y));
  }
}
}  // namespace sdb

bool sdb_index::sketch_supported() const {
  if (lay.tail != 0 || pq) return false;
  return lay.ng == 1 || lay.ng == 2 || lay.ng == 3 || lay.ng == 4 || lay.ng == 6;
}

void sdb_index::drop_sketch() {
  if (d_sketch) (void)hipFree(d_sketch);
  if (d_sketch_norm) (void)hipFree(d_sketch_norm);
  d_sketch = nullptr, d_sketch_norm = nullptr, sketch_cap = 0, sketch_gen = 0;
}

// `from` > 0: the copy is current for rows [0, from) -- a Vamana table never rewrites a committed row, it appends
// (updates are delete + insert, vamana.go:170-251) -- and only the rows behind them are converted; the two maxima
// carry on from their values (they stay upper bounds when rows are deleted).
int sdb_index::build_sketch(hipStream_t stream, uint32_t from) {
  const bool carry = from > 0 && from <= n && d_sketch && sketch_cap >= cap && sketch_gen != 0;
  if (!carry) from = 0;
  sketch_gen = 0;
  if (!tune_sketch || !sketch_supported() || n == 0) return SDB_OK;
  if (sketch_cap < cap) {
    drop_sketch();
    if (hipMalloc(&d_sketch, (size_t)cap * lay.ld * sizeof(uint16_t)) != hipSuccess ||
        hipMalloc(&d_sketch_norm, (size_t)cap * sizeof(float)) != hipSuccess) {  // a cache: without room for it the walk reads float32 rows
      (void)hipGetLastError();
      drop_sketch();
      return SDB_OK;
    }
    sketch_cap = cap;
  }
  if (!d_sk_counters) {
    if (hipMalloc(&d_sk_counters, 4 * sizeof(unsigned long long)) != hipSuccess) {  // (the copy is optional: so is this)
      (void)hipGetLastError();
      d_sk_counters = nullptr;
      drop_sketch();
      return SDB_OK;
    }
    SDB_HIP(hipMemset(d_sk_counters, 0, 4 * sizeof(unsigned long long)));
  }
  uint32_t *stats = reinterpret_cast<uint32_t *>(d_sk_counters + 2);
  uint32_t h[2] = {0, 0};
  if (from) memcpy(&h[0], &sk_emax, 4), memcpy(&h[1], &sk_ymax, 4);
  SDB_HIP(hipMemcpyAsync(stats, h, 8, hipMemcpyHostToDevice, stream));
  if (n > from)
    hipLaunchKernelGGL(sdb::k_sketch_rows, dim3((n - from + 3) / 4), dim3(256), 0, stream, d_slab + (size_t)from * lay.ld,
                       d_sketch + (size_t)from * lay.ld, d_sketch_norm + from, n - from, lay.ld, stats);
  SDB_HIP(hipGetLastError());
  SDB_HIP(hipMemcpyAsync(h, stats, 8, hipMemcpyDeviceToHost, stream));
  SDB_HIP(hipStreamSynchronize(stream));
  memcpy(&sk_emax, &h[0], 4), memcpy(&sk_ymax, &h[1], 4);
  sketch_gen = view_gen;
  return SDB_OK;
}

int sdb_index::ensure_idmap(const View &vw, hipStream_t stream) const {
  std::lock_guard<std::mutex> g(idmap.mu);
  if (idmap.gen == view_gen) return SDB_OK;
  // every reader of the previous table held the shared view lock over its probe kernel and the wait behind it, and
  // the view has been published (exclusively) since: nothing reads the old cells any more
  uint32_t cells = 1024;
  while (cells < 2ull * vw.n && cells < (1u << 31)) cells <<= 1;
  if (cells < 2ull * vw.n) return fail(SDB_ERR_INVALID, "too many rows for the id table");
  idmap.gen = 0;
  if (cells > idmap.cells) {
    if (idmap.keys) (void)hipFree(idmap.keys);
    if (idmap.vals) (void)hipFree(idmap.vals);
    idmap.keys = nullptr, idmap.vals = nullptr, idmap.cells = 0;
    if (hipMalloc(&idmap.keys, (size_t)cells * 8) != hipSuccess || hipMalloc(&idmap.vals, (size_t)cells * 4) != hipSuccess) {
      (void)hipGetLastError();
      if (idmap.keys) (void)hipFree(idmap.keys);
      idmap.keys = nullptr;
      return fail(SDB_ERR_DEVICE, "out of device memory for the id table (%u cells)", cells);
    }
    idmap.cells = cells;
  }
  SDB_HIP(hipMemsetAsync(idmap.keys, 0, (size_t)idmap.cells * 8, stream));
  hipLaunchKernelGGL(sdb::k_idmap_build, dim3((vw.n + 255) / 256), dim3(256), 0, stream, vw.ids, vw.n, idmap.keys, idmap.vals,
                     idmap.cells - 1);
  SDB_HIP(hipGetLastError());
  SDB_HIP(hipStreamSynchronize(stream));  // other searches probe it from their own streams
  idmap.gen = view_gen;
  return SDB_OK;
}

int64_t sdb_index::slot_of_committed(uint64_t id, uint32_t view_n) const {
  int64_t s = slot_of(id);
  if ((s < 0 || (uint32_t)s >= view_n) && in_tx) {  // removed or replaced by the open transaction: the committed row is still there for a search on the committed graph
    auto it = tx_deleted.find(id);
    if (it != tx_deleted.end()) s = (int64_t)it->second;
  }
  return (s >= 0 && (uint32_t)s < view_n) ? s : -1;  // rows past the committed count belong to the transaction
}

int sdb_index::alloc_adjcodes() {
  if (d_adjcodes) (void)hipFree(d_adjcodes);
  if (r_adjcodes) (void)hipFree(r_adjcodes);
  d_adjcodes = r_adjcodes = nullptr;
  view.adj_codes = nullptr;
  if (!pq || pq->M > kAdjCodesMaxM) return SDB_OK;
  const size_t bytes = (size_t)cap * kAdjStride * pq->M;
  // a cache of what the code rows hold: without room for it the walk gathers by slot, as it does for larger M
  if (hipMalloc(&d_adjcodes, bytes) != hipSuccess || hipMalloc(&r_adjcodes, bytes) != hipSuccess) {
    (void)hipGetLastError();
    if (d_adjcodes) (void)hipFree(d_adjcodes);
    d_adjcodes = r_adjcodes = nullptr;
  }
  return SDB_OK;
}

int sdb_index::rebuild_adjcodes(hipStream_t stream) {
  if (!has_adjcodes()) return SDB_OK;
  const uint32_t M = pq->M;
  if (n) hipLaunchKernelGGL(sdb::k_adjcodes_rows, dim3((n + 3) / 4), dim3(256), 0, stream, d_adj, d_codes, d_adjcodes, nullptr, n, 0u, M);
  if (view.n)
    hipLaunchKernelGGL(sdb::k_adjcodes_rows, dim3((view.n + 3) / 4), dim3(256), 0, stream, r_adj, d_codes, r_adjcodes, nullptr, view.n,
                       0u, M);
  SDB_HIP(hipGetLastError());
  SDB_HIP(hipStreamSynchronize(stream));
  view.adj_codes = r_adjcodes;
  return SDB_OK;
}

int sdb_index::commit(hipStream_t stream) {
  if (!in_tx) return SDB_OK;
  const bool sk_current = d_sketch && sketch_gen == view_gen && sketch_cap >= cap;  // the float16 copy describes the rows below tx_n0
  const uint32_t need = (uint32_t)((h_start_ext.size() + 63) / 64 * 64);
  const uint32_t ac_bytes = has_adjcodes() ? kAdjStride * pq->M : 0;
  if (ac_bytes && n) {
    // the writer's copy of the neighbours' code rows catches up with what the transaction did to the adjacency rows --
    // before the copies change hands, while the searches still walk the other one
    hipLaunchKernelGGL(sdb::k_adjcodes_rows, dim3((n + 3) / 4), dim3(256), 0, stream, d_adj, d_codes, d_adjcodes, d_dirty, n,
                       tx_n0 < n ? tx_n0 : n, pq->M);
    SDB_HIP(hipGetLastError());
    SDB_HIP(hipStreamSynchronize(stream));
  }
  {
    std::unique_lock<sdb::ViewMutex> wl(view_mu);
    // every search that took the old view has enqueued its kernels and recorded its event by now (it held the
    // shared lock until then): the writer's stream waits for them before it touches the copy they walk
    {
      std::lock_guard<std::mutex> g(mu);
      for (auto *w : pool)
        if (w->launched_valid) SDB_HIP(hipStreamWaitEvent(stream, w->launched, 0));
    }
    std::swap(d_adj, r_adj);
    std::swap(d_adjcodes, r_adjcodes);
    std::swap(d_ids, r_ids);
    std::swap(d_start_ext, r_start_ext);
    std::swap(start_ext_cap, r_start_ext_cap);
    view.n = n, view.adj = r_adj, view.ids = r_ids, view.start_ext = r_start_ext;
    view.start_ext_n = (uint32_t)h_start_ext.size();
    view.adj_codes = r_adjcodes;
    view_gen++;
    tx_deleted.clear();
    in_tx = false, tx_explicit = false, tx_dirty = false;
  }
  // the writer's copy is now the one the last version's searches walked: bring it up to date
  if (n) {
    hipLaunchKernelGGL(sdb::k_sync_rows, dim3((n + 3) / 4), dim3(256), 0, stream, r_adj, d_adj, r_ids, d_ids, d_dirty, n,
                       tx_n0 < n ? tx_n0 : n, r_adjcodes, d_adjcodes, ac_bytes);
    SDB_HIP(hipGetLastError());
  }
  if (need) {
    if (need > start_ext_cap) {
      SDB_HIP(hipStreamSynchronize(stream));
      if (d_start_ext) (void)hipFree(d_start_ext);
      d_start_ext = nullptr, start_ext_cap = 0;
      SDB_HIP(hipMalloc(&d_start_ext, (size_t)need * 2 * 4));
      start_ext_cap = need * 2;
    }
    SDB_HIP(hipMemcpyAsync(d_start_ext, r_start_ext, (size_t)need * 4, hipMemcpyDeviceToDevice, stream));
  }
  // the float16 copy follows: on `stream`, which has waited for the searches of the old view (above); searches of the
  // new one read float32 rows until sketch_gen says the copy is theirs
  if (tune_sketch) SDB_TRY(build_sketch(stream, sk_current && start_slot >= 0 ? tx_n0 : 0));
  return SDB_OK;
}

int sdb_index::rollback() {
  std::unique_lock<sdb::ViewMutex> wl(view_mu);  // searches translate filter ids with the tables changed below
  SDB_HIP(hipDeviceSynchronize());               // the transaction's kernels are done
  if (n)
    hipLaunchKernelGGL(sdb::k_restore_rows, dim3((n + 3) / 4), dim3(256), 0, nullptr, r_adj, d_adj, r_ids, d_ids, d_deg,
                       d_clean, d_dcount, d_dirty, n, tx_n0);
  SDB_HIP(hipGetLastError());
  // The writer's copy of the neighbours' code rows follows the restored adjacency rows.  commit() brings it to the
  // transaction's state BEFORE the copies change hands, so a commit that failed after that step -- or a transaction that
  // is simply aborted after its rows were written -- would leave restored rows carrying the transaction's code rows:
  // edge e walked with another node's codes, silently.  All committed rows (a rollback is rare; 64 M bytes per row).
  if (has_adjcodes() && tx_n0) {
    hipLaunchKernelGGL(sdb::k_adjcodes_rows, dim3((tx_n0 + 3) / 4), dim3(256), 0, nullptr, d_adj, d_codes, d_adjcodes, nullptr,
                       tx_n0, 0u, pq->M);
    SDB_HIP(hipGetLastError());
  }
  // the start node's overflow list as committed
  const uint32_t ext_n = view.start_ext_n, need = (ext_n + 63) / 64 * 64;
  h_start_ext.assign(ext_n, 0);
  if (ext_n) {
    SDB_HIP(hipMemcpy(h_start_ext.data(), r_start_ext, (size_t)ext_n * 4, hipMemcpyDeviceToHost));
    if (need > start_ext_cap) {
      if (d_start_ext) (void)hipFree(d_start_ext);
      d_start_ext = nullptr, start_ext_cap = 0;
      SDB_HIP(hipMalloc(&d_start_ext, (size_t)need * 2 * 4));
      start_ext_cap = need * 2;
    }
    SDB_HIP(hipMemcpy(d_start_ext, r_start_ext, (size_t)need * 4, hipMemcpyDeviceToDevice));
  }
  SDB_HIP(hipDeviceSynchronize());
  // host tables: appended rows go, rows the transaction tombstoned get their ids back
  h_ids.resize(tx_n0);
  for (auto &kv : tx_deleted)
    if (kv.second < tx_n0) h_ids[kv.second] = kv.first;
  n = tx_n0, n_dead = tx_dead0, max_node_id = tx_max_id0;
  bool dense = n > 0;
  for (uint32_t i = 1; i < n && dense; i++) dense = h_ids[i] == h_ids[0] + i;
  if (n && h_ids[0] == 0) dense = false;
  dense_ids = n == 0 ? true : dense;
  id2slot.clear();
  if (!dense_ids) {
    id2slot.reserve((size_t)n * 2);
    for (uint32_t sl = 0; sl < n; sl++)
      if (h_ids[sl] != 0) id2slot.emplace(h_ids[sl], sl);
  }
  tx_deleted.clear();
  in_tx = false, tx_explicit = false, tx_dirty = false;
  if (tune_sketch) SDB_TRY(build_sketch(nullptr));  // (exclusive lock held, device idle)
  return SDB_OK;
}

int sdb_index::publish_full() {
  SDB_HIP(hipDeviceSynchronize());
  std::unique_lock<sdb::ViewMutex> wl(view_mu);
  if (n) {
    SDB_HIP(hipMemcpy(r_adj, d_adj, (size_t)n * kAdjStride * 4, hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemcpy(r_ids, d_ids, (size_t)n * 8, hipMemcpyDeviceToDevice));
    SDB_HIP(hipMemset(d_dirty, 0, n));
    if (has_adjcodes()) {
      hipLaunchKernelGGL(sdb::k_adjcodes_rows, dim3((n + 3) / 4), dim3(256), 0, nullptr, d_adj, d_codes, d_adjcodes, nullptr, n, 0u, pq->M);
      SDB_HIP(hipGetLastError());
      SDB_HIP(hipMemcpy(r_adjcodes, d_adjcodes, (size_t)n * kAdjStride * pq->M, hipMemcpyDeviceToDevice));
    }
  }
  const uint32_t need = (uint32_t)((h_start_ext.size() + 63) / 64 * 64);
  if (need) {
    if (need > r_start_ext_cap) {
      if (r_start_ext) (void)hipFree(r_start_ext);
      r_start_ext = nullptr, r_start_ext_cap = 0;
      SDB_HIP(hipMalloc(&r_start_ext, (size_t)need * 2 * 4));
      r_start_ext_cap = need * 2;
    }
    SDB_HIP(hipMemcpy(r_start_ext, d_start_ext, (size_t)need * 4, hipMemcpyDeviceToDevice));
  }
  view.n = n, view.adj = r_adj, view.ids = r_ids, view.start_ext = r_start_ext;
  view.start_ext_n = (uint32_t)h_start_ext.size();
  view.adj_codes = r_adjcodes;
  view_gen++;
  tx_deleted.clear();
  in_tx = false, tx_explicit = false, tx_dirty = false;
  if (tune_sketch) {
    SDB_HIP(hipDeviceSynchronize());  // searches that slipped in between the first wait and the lock
    SDB_TRY(build_sketch(nullptr));
  }
  return SDB_OK;
}

// A workspace (visited bitsets, staging, LUTs) belongs to one in-flight batch.  Host-memory calls hold it
// until they have synchronised.  Device-memory calls return before the kernel runs, so the workspace
// stays bound to the caller's stream: it is handed out again only to a call on the same stream (stream
// order then serialises the reuse) or once its completion event has fired.
Workspace *sdb_index::acquire_ws(hipStream_t stream, bool async) const {
  std::lock_guard<std::mutex> g(mu);
  for (auto *w : pool) {
    if (w->busy) continue;
    if (w->pending) {
      if (async && w->bound_stream == stream) {
        w->busy = true;
        w->launched_is_tail = false;
        return w;
      }
      if (hipEventQuery(w->done_is_launched ? w->launched : w->done) != hipSuccess) continue;
      w->pending = false;
    }
    w->busy = true;
    w->launched_is_tail = false;
    return w;
  }
  std::unique_ptr<Workspace> w(new Workspace());
  w->device = P.device;
  w->busy = true;
  pool.push_back(w.get());
  return w.release();
}

void sdb_index::release_ws(Workspace *ws, hipStream_t stream, bool async) const {
  std::lock_guard<std::mutex> g(mu);
  if (async) {
    if (ws->launched_is_tail && ws->launched_valid) {  // recorded on `stream` behind everything this call enqueued
      ws->pending = true, ws->bound_stream = stream, ws->done_is_launched = true;
    } else {
      ws->done_is_launched = false;
      if (!ws->done) (void)hipEventCreateWithFlags(&ws->done, hipEventDisableTiming);
      if (ws->done && hipEventRecord(ws->done, stream) == hipSuccess) {
        ws->pending = true;
        ws->bound_stream = stream;
      }
    }
  }
  ws->busy = false;
}

// uploads the start node's overflow edges, padded with kNoSlot to whole 64-entry chunks (search_kernel.h reads them
// like adjacency rows).  Writers are exclusive (the reference holds the shard's write lock), so nothing reads the
// old buffer once the device is idle.
int sdb_index::sync_start_ext() {
  const uint32_t need = (uint32_t)((h_start_ext.size() + 63) / 64 * 64);
  if (need > start_ext_cap) {
    SDB_HIP(hipDeviceSynchronize());
    if (d_start_ext) (void)hipFree(d_start_ext);
    d_start_ext = nullptr, start_ext_cap = 0;
    const uint32_t ncap = need * 2;
    SDB_HIP(hipMalloc(&d_start_ext, (size_t)ncap * 4));
    start_ext_cap = ncap;
  }
  if (need) {
    std::vector<uint32_t> padded(need, kNoSlot);
    std::copy(h_start_ext.begin(), h_start_ext.end(), padded.begin());
    SDB_HIP(hipMemcpy(d_start_ext, padded.data(), (size_t)need * 4, hipMemcpyHostToDevice));
  }
  return SDB_OK;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
// Page-locked host memory the kernels can address in place.  Blocks from sdb_host_alloc are kept in a table (exact,
// and dropped by sdb_host_free before the memory goes back); any other pointer is asked of the runtime -- memory the
// caller page-locked itself (hipHostMalloc, hipHostRegister; torch's pin_memory) says so, a pageable pointer does not.
struct PinnedRange {
  size_t bytes;
  char *dev;  // the device's address of the block's first byte (NULL: not mapped)
};
static std::shared_mutex g_pinned_mu;
static std::map<uintptr_t, PinnedRange> g_pinned;

static bool device_view_of_host(const void *p, size_t bytes, void **dev) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  {
    std::shared_lock<std::shared_mutex> g(g_pinned_mu);
    auto it = g_pinned.upper_bound(a);
    if (it != g_pinned.begin()) {
      --it;
      if (a >= it->first && a + bytes <= it->first + it->second.bytes) {
        if (!it->second.dev) return false;
        *dev = it->second.dev + (a - it->first);
        return true;
      }
    }
  }
  hipPointerAttribute_t attr{};
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // a pageable pointer is not an error of this call
    return false;
  }
  if (attr.type != hipMemoryTypeHost || !attr.devicePointer) return false;
  *dev = attr.devicePointer;
  return true;
}

extern "C" {

const char *sdb_last_error(void) { return last_error_buf(); }

int sdb_abi_version(void) { return SDB_ABI_VERSION; }

int sdb_device_count(int *count) try {
  if (!count) return fail(SDB_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    *count = 0;
    return fail(SDB_ERR_DEVICE, "no HIP device visible: %s", hipGetErrorString(e));
  }
  *count = n;
  return SDB_OK;
}
SDB_API_CATCH("sdb_device_count")

int sdb_host_alloc(size_t bytes, void **out) try {
  if (!out) return fail(SDB_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (bytes == 0) return SDB_OK;
  void *p = nullptr;
  SDB_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  void *dev = nullptr;
  if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) dev = nullptr, (void)hipGetLastError();
  try {
    std::unique_lock<std::shared_mutex> g(g_pinned_mu);
    g_pinned[reinterpret_cast<uintptr_t>(p)] = PinnedRange{bytes, static_cast<char *>(dev)};
  } catch (...) {  // no room for the note: the block is still good memory, searches just stage through it
  }
  *out = p;
  return SDB_OK;
}
SDB_API_CATCH("sdb_host_alloc")

int sdb_host_free(void *p) try {
  if (!p) return SDB_OK;
  {
    std::unique_lock<std::shared_mutex> g(g_pinned_mu);
    g_pinned.erase(reinterpret_cast<uintptr_t>(p));
  }
  SDB_HIP(hipHostFree(p));
  return SDB_OK;
}
SDB_API_CATCH("sdb_host_free")

int sdb_index_create(const sdb_index_params *p, sdb_index **out) try {
  if (!p || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (p->dim < 1 || p->dim > 4096)  // models/index.go:285-287
    return fail(SDB_ERR_INVALID, "vector size must be between 1 and 4096, got %u", p->dim);
  if (p->metric > SDB_METRIC_DOT) return fail(SDB_ERR_INVALID, "unknown distance metric %u", p->metric);
  if (p->degree_bound < 1 || p->degree_bound > kAdjStride)
    return fail(SDB_ERR_INVALID, "degree bound must be between 1 and %u, got %u", kAdjStride, p->degree_bound);
  if (p->search_size < 1 || p->search_size > 512)
    return fail(SDB_ERR_INVALID, "search size must be between 1 and 512, got %u", p->search_size);
  if (p->strict) {  // models/index.go:299-307
    if (p->search_size < 25 || p->search_size > 75)
      return fail(SDB_ERR_INVALID, "search size must be between 25 and 75, got %u", p->search_size);
    if (p->degree_bound < 32 || p->degree_bound > 64)
      return fail(SDB_ERR_INVALID, "degree bound must be between 32 and 64, got %u", p->degree_bound);
    if (p->alpha < 1.1f || p->alpha > 1.5f)
      return fail(SDB_ERR_INVALID, "alpha must be between 1.1 and 1.5, got %f", (double)p->alpha);
  }
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  if (p->device < 0 || p->device >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", p->device);
  DeviceGuard dg(p->device);
  if (!dg.ok) return fail(SDB_ERR_DEVICE, "hipSetDevice(%d) failed", p->device);
  std::unique_ptr<sdb_index> ix(new sdb_index());
  ix->P = *p;
  ix->lay = RowLayout(p->dim);
  SDB_TRY(ix->reserve((uint32_t)std::max<uint64_t>(p->capacity ? p->capacity : 1024, 16)));
  *out = ix.release();
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_create")

int sdb_index_destroy(sdb_index *ix) try {
  if (!ix) return SDB_OK;
  DeviceGuard dg(ix->P.device);
  (void)hipDeviceSynchronize();
  if (ix->d_slab) (void)hipFree(ix->d_slab);
  ix->drop_sketch();
  if (ix->d_sk_counters) (void)hipFree(ix->d_sk_counters);
  if (ix->d_adj) (void)hipFree(ix->d_adj);
  if (ix->d_deg) (void)hipFree(ix->d_deg);
  if (ix->d_clean) (void)hipFree(ix->d_clean);
  if (ix->d_adjdist) (void)hipFree(ix->d_adjdist);
  if (ix->d_dcount) (void)hipFree(ix->d_dcount);
  if (ix->d_ids) (void)hipFree(ix->d_ids);
  if (ix->r_adj) (void)hipFree(ix->r_adj);
  if (ix->r_ids) (void)hipFree(ix->r_ids);
  if (ix->idmap.keys) (void)hipFree(ix->idmap.keys);
  if (ix->idmap.vals) (void)hipFree(ix->idmap.vals);
  if (ix->r_start_ext) (void)hipFree(ix->r_start_ext);
  if (ix->d_dirty) (void)hipFree(ix->d_dirty);
  if (ix->d_start_ext) (void)hipFree(ix->d_start_ext);
  if (ix->d_codes) (void)hipFree(ix->d_codes);
  if (ix->d_adjcodes) (void)hipFree(ix->d_adjcodes);
  if (ix->r_adjcodes) (void)hipFree(ix->r_adjcodes);
  if (ix->d_bstats) (void)hipFree(ix->d_bstats);
  for (auto e : ix->ev0)
    if (e) (void)hipEventDestroy(e);
  for (auto e : ix->ev1)
    if (e) (void)hipEventDestroy(e);
  for (auto *w : ix->pool) {
    w->release();
    delete w;
  }
  delete ix;
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_destroy")

// copies n original-layout rows (host or device) into slab rows [first, first+n)
static int store_rows(sdb_index *ix, uint32_t first, uint32_t n, const float *vectors, int mem, hipStream_t stream);
static int store_rows(sdb_index *ix, uint32_t first, uint32_t n, const float *vectors, int mem,
                      hipStream_t stream) {
  if (n == 0) return SDB_OK;
  const RowLayout &l = ix->lay;
  const float *src = vectors;
  float *staging = nullptr;
  if (mem == SDB_MEM_HOST) {
    SDB_HIP(hipMalloc(&staging, (size_t)n * l.dim * sizeof(float)));
    hipError_t e = hipMemcpyAsync(staging, vectors, (size_t)n * l.dim * sizeof(float), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) {
      (void)hipFree(staging);
      return fail(SDB_ERR_DEVICE, "H2D copy failed: %s", hipGetErrorString(e));
    }
    src = staging;
  }
  hipLaunchKernelGGL(k_permute_rows, dim3(n), dim3(128), 0, stream, src, ix->d_slab + (size_t)first * l.ld, n,
                     l.dim, l.nblk, l.ng, l.tail, l.ld);
  hipError_t e = hipGetLastError();
  if (staging) {
    (void)hipStreamSynchronize(stream);
    (void)hipFree(staging);
  }
  if (e != hipSuccess) return fail(SDB_ERR_DEVICE, "permute launch failed: %s", hipGetErrorString(e));
  return SDB_OK;
}

int sdb_index_set_start(sdb_index *ix, const float *vec, int mem) try {
  if (!ix || !vec) return fail(SDB_ERR_INVALID, "NULL argument");
  if (ix->start_slot >= 0) return SDB_OK;  // vamana.go:95-97: already there
  if (ix->n != 0) return fail(SDB_ERR_STATE, "start node must be the first node of an empty index");
  DeviceGuard dg(ix->P.device);
  SDB_TRY(ix->reserve(1));
  SDB_TRY(store_rows(ix, 0, 1, vec, mem, nullptr));
  uint64_t id = SDB_STARTID;
  SDB_HIP(hipMemcpy(ix->d_ids, &id, sizeof(id), hipMemcpyHostToDevice));
  SDB_HIP(hipDeviceSynchronize());
  ix->h_ids.assign(1, id);
  ix->dense_ids = true;
  ix->n = 1;
  ix->start_slot = 0;
  return ix->publish_full();
}
SDB_API_CATCH("sdb_index_set_start")

int sdb_index_load(sdb_index *ix, uint64_t n, const uint64_t *ids, const float *vectors,
                   const uint64_t *offsets, const uint64_t *edges, int mem) try {
  if (!ix || !vectors || !offsets) return fail(SDB_ERR_INVALID, "NULL argument");
  if (ix->n != 0) return fail(SDB_ERR_STATE, "index is not empty");
  if (n == 0 || n >= 0x7FFFFFFFull) return fail(SDB_ERR_INVALID, "node count %llu out of range", (unsigned long long)n);
  if (offsets[n] && !edges) return fail(SDB_ERR_INVALID, "edges is NULL");
  DeviceGuard dg(ix->P.device);
  SDB_TRY(ix->reserve((uint32_t)n));
  // whatever ends this call early -- an error return, a host allocation that fails (the adjacency staging is 3.2 GB at
  // 12.5 M rows) -- leaves the index empty, as it was
  struct Undo {
    sdb_index *ix;
    bool keep = false;
    uint32_t wrote = 0;  // device rows that may hold part of the graph: empty again (rows past n are assumed empty)
    ~Undo() {
      if (keep) return;
      ix->n = 0, ix->start_slot = -1, ix->max_node_id = 0, ix->dense_ids = true;
      ix->h_ids.clear(), ix->id2slot.clear(), ix->h_start_ext.clear();
      if (wrote) {
        (void)hipMemset(ix->d_adj, 0xFF, (size_t)wrote * kAdjStride * 4);
        (void)hipMemset(ix->d_deg, 0, (size_t)wrote * 4);
        (void)hipDeviceSynchronize();
      }
    }
  } undo{ix};
  // id table
  ix->h_ids.resize(n);
  bool dense = true;
  for (uint64_t i = 0; i < n; i++) {
    ix->h_ids[i] = ids ? ids[i] : i + 1;
    if (i && ix->h_ids[i] != ix->h_ids[0] + i) dense = false;
  }
  ix->dense_ids = dense;
  ix->id2slot.clear();
  ix->start_slot = -1;
  ix->max_node_id = 0;
  if (!dense) ix->id2slot.reserve(n * 2);
  for (uint64_t i = 0; i < n; i++) {
    uint64_t id = ix->h_ids[i];
    if (id == 0) {
      ix->h_ids.clear();
      return fail(SDB_ERR_INVALID, "invalid point id: 0");
    }
    if (!dense && !ix->id2slot.emplace(id, (uint32_t)i).second) {
      ix->h_ids.clear();
      return fail(SDB_ERR_INVALID, "duplicate node id %llu", (unsigned long long)id);
    }
    if (id == SDB_STARTID) ix->start_slot = (int64_t)i;
    else if (id > ix->max_node_id) ix->max_node_id = id;
  }
  if (ix->start_slot < 0) {
    ix->h_ids.clear();
    ix->id2slot.clear();
    return fail(SDB_ERR_INVALID, "start node (id %llu) is not among the loaded ids", SDB_STARTID);
  }
  ix->n = (uint32_t)n;  // slot_of works from here on
  ix->h_start_ext.clear();
  // adjacency rows: ids -> slots, unknown ids dropped, first occurrence kept, edge order kept
  std::vector<uint32_t> adj((size_t)n * kAdjStride, kNoSlot), deg(n, 0);
  for (uint64_t i = 0; i < n; i++) {
    uint32_t *row = adj.data() + (size_t)i * kAdjStride;
    uint32_t dcnt = 0;
    for (uint64_t e = offsets[i]; e < offsets[i + 1]; e++) {
      int64_t s = ix->slot_of(edges[e]);
      if (s < 0) continue;  // itemcache.go:109-128
      bool dup = false;
      for (uint32_t k = 0; k < dcnt; k++) dup |= (row[k] == (uint32_t)s);
      if (dup) continue;
      if (dcnt == kAdjStride) {
        if ((int64_t)i == ix->start_slot) {  // the start node has no bound (node.go:73-80): the rest is its overflow list
          if (std::find(ix->h_start_ext.begin(), ix->h_start_ext.end(), (uint32_t)s) == ix->h_start_ext.end())
            ix->h_start_ext.push_back((uint32_t)s);
          continue;
        }
        ix->n = 0;
        ix->h_start_ext.clear();
        return fail(SDB_ERR_INVALID, "node %llu has more than %u edges", (unsigned long long)ix->h_ids[i], kAdjStride);
      }
      row[dcnt++] = (uint32_t)s;
    }
    deg[i] = dcnt;
  }
  SDB_TRY(ix->sync_start_ext());
  undo.wrote = (uint32_t)n;
  SDB_HIP(hipMemcpy(ix->d_adj, adj.data(), adj.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  SDB_HIP(hipMemcpy(ix->d_deg, deg.data(), deg.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  SDB_HIP(hipMemcpy(ix->d_ids, ix->h_ids.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
  SDB_TRY(store_rows(ix, 0, (uint32_t)n, vectors, mem, nullptr));
  SDB_TRY(ix->publish_full());
  undo.keep = true;
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_load")

// IndexVamana.InsertUpdateDelete is ONE write transaction made of several calls here (inserts, one delete scan,
// re-inserts of the updated points); the shard runs it under its write lock while searches keep being served --
// by a cold index built from the bucket when the cached one is locked (shard/cache/manager.go:159-181).  Here the
// searches simply keep walking the last committed graph: between begin_write and commit every insert_batch /
// delete_batch changes the writer's copy only.  Without begin_write each such call is a transaction by itself.
int sdb_index_begin_write(sdb_index *ix) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  if (ix->in_tx && ix->tx_explicit) return fail(SDB_ERR_STATE, "a write transaction is already open");
  SDB_TRY(ix->begin_write());
  ix->tx_explicit = true;
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_begin_write")

int sdb_index_commit(sdb_index *ix, void *stream_) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (!ix->in_tx) return fail(SDB_ERR_STATE, "no write transaction is open");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  DeviceGuard dg(ix->P.device);
  hipStream_t stream = as_stream(stream_);
  SDB_TRY(ix->commit(stream));
  SDB_HIP(hipStreamSynchronize(stream));
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_commit")

// The way out of a transaction that will not be committed (a host that found a bad point after begin_write, a failed
// call, a cancelled request).  Searches never saw the transaction: they walk the committed copy of the graph (index.h
// graph versions), which is exactly what a rollback needs -- the rows the transaction wrote take the committed copy
// back, the rows it appended are dropped, the host's id tables follow.  The reference has no such thing: after an
// error inside a transaction its cache manager scraps the shard's cache and rebuilds it from the bucket
// (shard/cache/manager.go:231-240); here the index is what it was at begin_write, in milliseconds.  Only a handle that
// a device failure left half-written (index.h `broken`) cannot be brought back.
int sdb_index_abort_write(sdb_index *ix) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  if (!ix->in_tx) return SDB_OK;
  if (!ix->tx_dirty) {  // nothing to undo
    std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);
    ix->tx_deleted.clear();
    ix->in_tx = false, ix->tx_explicit = false;
    return SDB_OK;
  }
  DeviceGuard dg(ix->P.device);
  int rc;
  try {
    rc = ix->rollback();
  } catch (...) {
    rc = on_exception("sdb_index_abort_write");
  }
  if (rc != SDB_OK) ix->broken = true;  // a device error (or no host memory for the id tables) half-way through the restore
  return rc;
}
SDB_API_CATCH("sdb_index_abort_write")

// test support: the number of rows on which the two graph copies differ (0 whenever no transaction is open)
int sdb_index_version_diff(const sdb_index *ix, uint64_t *rows) try {
  if (!ix || !rows) return fail(SDB_ERR_INVALID, "NULL argument");
  *rows = 0;
  if (ix->n == 0) return SDB_OK;
  DeviceGuard dg(ix->P.device);
  unsigned long long *d = nullptr;
  SDB_HIP(hipMalloc(&d, 8));
  SDB_HIP(hipMemset(d, 0, 8));
  SDB_HIP(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_count_version_diff, dim3((ix->n + 3) / 4), dim3(256), 0, nullptr, ix->d_adj, ix->r_adj, ix->d_ids,
                     ix->r_ids, ix->n, d);
  unsigned long long h = 0;
  hipError_t e = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(SDB_ERR_DEVICE, "version check failed: %s", hipGetErrorString(e));
  *rows = h;
  if (ix->h_start_ext.size()) {
    const size_t m = ix->h_start_ext.size();
    std::vector<uint32_t> a(m), b(m);
    SDB_HIP(hipMemcpy(a.data(), ix->d_start_ext, m * 4, hipMemcpyDeviceToHost));
    SDB_HIP(hipMemcpy(b.data(), ix->r_start_ext, m * 4, hipMemcpyDeviceToHost));
    if (a != b) *rows += 1;
  }
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_version_diff")

// a filter handed over as bitmaps (sdb_index_search_batch_bitmap)
struct BitmapFilters {
  const uint64_t *first_id, *word_offsets, *words;
};
constexpr int kBitmapNeedsIds = -7001;  // search_batch_impl: this table cannot take bitmaps on the device; the caller expands them

static int search_batch_impl(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit,
                             uint32_t search_size, const uint64_t *filter_offsets,
                             const uint64_t *filter_ids, const BitmapFilters *bm, uint64_t *out_ids, float *out_dists,
                             uint32_t *out_counts, const sdb_search_trace *trace, int mem, void *stream_) {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (nq == 0) return SDB_OK;
  if (!queries || !out_ids || !out_dists || !out_counts) return fail(SDB_ERR_INVALID, "NULL argument");
  if (limit < 1) return fail(SDB_ERR_INVALID, "invalid limit %u for vector query", limit);
  if (search_size < limit)  // search.go:23-25
    return fail(SDB_ERR_INVALID, "searchSize (%u) must be greater than k (%u)", search_size, limit);
  if (ix->P.strict && (search_size < 25 || search_size > 75 || limit > 75))  // models/search.go:287-297
    return fail(SDB_ERR_INVALID, "invalid searchSize %u / limit %u for vector query, expected 25-75 / 1-75",
                search_size, limit);
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  if (ix->start_slot < 0) return fail(SDB_ERR_STATE, "failed to get start point");  // search.go:57-60
  const bool filtered = filter_offsets != nullptr || bm != nullptr;
  if (filter_offsets && filter_offsets[nq] && !filter_ids) return fail(SDB_ERR_INVALID, "filter_ids is NULL");
  if (nq > 0x7FFFFFFFull) return fail(SDB_ERR_INVALID, "too many queries");
  DeviceGuard dg(ix->P.device);
  hipStream_t stream = as_stream(stream_);
  const bool async = mem == SDB_MEM_DEVICE;
  Workspace *ws = ix->acquire_ws(stream, async);
  struct Rel {
    const sdb_index *ix;
    Workspace *ws;
    hipStream_t stream;
    bool async;
    ~Rel() { ix->release_ws(ws, stream, async); }
  } rel{ix, ws, stream, async};
  if (mem == SDB_MEM_HOST) {
    if (!ws->own_stream) SDB_HIP(hipStreamCreateWithFlags(&ws->own_stream, hipStreamNonBlocking));
    stream = ws->own_stream;
  }
  const RowLayout &l = ix->lay;
  // the graph version this batch walks: the last committed one, whatever a writer is doing meanwhile.  The shared
  // lock is held until the kernels are enqueued and their event recorded (sdb_index::commit counts on that).
  std::shared_lock<sdb::ViewMutex> rl(ix->view_mu);
  const sdb_index::View vw = ix->view;
  const uint32_t words = ((vw.n + 31) / 32 + 31) & ~31u;  // per-query bitset, 128-byte multiple
  const size_t bs_bytes = (size_t)nq * words * sizeof(uint32_t);
  SDB_TRY(ws->ensure_bitsets(filtered ? 2 * bs_bytes : bs_bytes));

  SearchArgs a{};
  a.slab = ix->d_slab, a.adj = vw.adj, a.ids = vw.ids;
  a.adj_codes = vw.adj_codes, a.adj_rows = vw.n;
  a.bitsets = ws->bitsets, a.words_per_query = words;
  if (bm) {
    // bitmaps: two passes on the device (count, expand) turn them into the same ascending slot lists; only for a table
    // with consecutive ids, where id -> slot is a subtraction
    if (ix->n == 0 || vw.n == 0 || ix->tune_host_filters) return kBitmapNeedsIds;
    // a table whose ids are not consecutive: the bitmaps become id lists ON THE DEVICE (every bit, as the id it stands
    // for) and those are resolved through the committed view's id -> slot table like uploaded id lists are
    const bool by_map = !ix->dense_ids;
    if (by_map && ix->ensure_idmap(vw, stream) != SDB_OK) return kBitmapNeedsIds;
    for (uint64_t q = 0; q < nq; q++)
      if (bm->word_offsets[q + 1] < bm->word_offsets[q]) return fail(SDB_ERR_INVALID, "filter_word_offsets must be non-decreasing");
    const uint64_t total_words = bm->word_offsets[nq] - bm->word_offsets[0];
    if (total_words >= (1ull << 32)) return fail(SDB_ERR_INVALID, "filter bitmaps of one batch are limited to 2^32 words");
    const uint64_t base_id = ix->h_ids[0];
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_off = up((nq + 1) * 4), b_first = up(nq * 8), b_words = up(total_words * 8), b_cnt = up(nq * 4);
    SDB_TRY(ws->ensure_filter_host(b_off + 256));
    SDB_TRY(ws->ensure_filter_aux(2 * b_off + b_first + b_words + 2 * b_cnt + 256));
    char *fb = static_cast<char *>(ws->filter_aux);
    uint32_t *d_woff = (uint32_t *)fb, *d_off = (uint32_t *)(fb + b_off);
    uint64_t *d_first = (uint64_t *)(fb + 2 * b_off), *d_words = (uint64_t *)(fb + 2 * b_off + b_first);
    uint32_t *d_fc = (uint32_t *)(fb + 2 * b_off + b_first + b_words), *d_sc = (uint32_t *)((char *)d_fc + b_cnt);
    uint32_t *d_flags = (uint32_t *)((char *)d_sc + b_cnt);
    uint32_t *h_off = static_cast<uint32_t *>(ws->filter_host);
    uint32_t *h_back = h_off + (nq + 1) + 2;  // [0] total slots, [1] overflow
    for (uint64_t q = 0; q <= nq; q++) h_off[q] = (uint32_t)(bm->word_offsets[q] - bm->word_offsets[0]);
    SDB_HIP(hipMemcpyAsync(d_woff, h_off, (nq + 1) * 4, hipMemcpyHostToDevice, stream));
    SDB_HIP(hipMemcpyAsync(d_first, bm->first_id, nq * 8, hipMemcpyHostToDevice, stream));
    if (total_words)
      SDB_HIP(hipMemcpyAsync(d_words, bm->words + bm->word_offsets[0], total_words * 8, hipMemcpyHostToDevice, stream));
    SDB_HIP(hipMemsetAsync(d_flags, 0, 16, stream));
    if (by_map)
      hipLaunchKernelGGL(k_filter_bitmap_count<true>, dim3((unsigned)nq), dim3(64), 0, stream, d_words, d_woff, d_first, base_id, vw.n,
                         search_size, d_fc, d_sc);
    else
      hipLaunchKernelGGL(k_filter_bitmap_count<false>, dim3((unsigned)nq), dim3(64), 0, stream, d_words, d_woff, d_first, base_id, vw.n,
                         search_size, d_fc, d_sc);
    hipLaunchKernelGGL(k_filter_offsets, dim3(1), dim3(1024), 0, stream, d_fc, (uint32_t)nq, d_off, d_flags);
    SDB_HIP(hipGetLastError());
    SDB_HIP(hipMemcpyAsync(h_back, d_off + nq, 4, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipMemcpyAsync(h_back + 1, d_flags, 4, hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipStreamSynchronize(stream));  // the slot list's size; the caller's arrays are free again
    if (h_back[1]) return fail(SDB_ERR_INVALID, "the filters of one batch name more than 2^32 stored ids");
    const size_t b_sl = up((size_t)h_back[0] * 4);
    SDB_TRY(ws->ensure_filter(b_sl + (by_map ? (size_t)h_back[0] * 8 : 0) + 256));
    uint32_t *d_sl = static_cast<uint32_t *>(ws->filter);
    if (by_map) {
      uint64_t *d_ids64 = reinterpret_cast<uint64_t *>(static_cast<char *>(ws->filter) + b_sl);
      hipLaunchKernelGGL(k_filter_bitmap_expand<true>, dim3((unsigned)nq), dim3(64), 0, stream, d_words, d_woff, d_first, base_id, vw.n,
                         d_off, d_sl, d_ids64);
      // d_flags[0] is the offsets' overflow word (zero here); the resolve kernel's three words follow it
      hipLaunchKernelGGL(k_filter_resolve<true>, dim3((unsigned)nq), dim3(64), 0, stream, d_ids64, d_off, base_id, vw.n, search_size, d_sl,
                         d_fc, d_sc, d_flags + 1, ix->idmap.keys, ix->idmap.vals, ix->idmap.cells - 1);
      SDB_HIP(hipGetLastError());
      SDB_HIP(hipMemcpyAsync(h_back, d_flags + 3, 4, hipMemcpyDeviceToHost, stream));
      SDB_HIP(hipStreamSynchronize(stream));
      if (h_back[0]) a.filt_ids = d_ids64;  // ids and slots disagree on the order somewhere: Contains by id (search_kernel.h)
    } else {
      hipLaunchKernelGGL(k_filter_bitmap_expand<false>, dim3((unsigned)nq), dim3(64), 0, stream, d_words, d_woff, d_first, base_id, vw.n,
                         d_off, d_sl, nullptr);
      SDB_HIP(hipGetLastError());
    }
    a.seed_off = d_off, a.filt_off = d_off, a.seeds = d_sl, a.filt_slots = d_sl, a.seed_cnt = d_sc, a.filt_cnt = d_fc;
    a.rbitsets = ws->bitsets + (size_t)nq * words;
  } else if (filtered) {
    // search.go:41-48: seeds = the first <= searchSize filter ids (ascending) that exist; Contains (:93) is
    // answered from the whole filter as ascending slots.  Filter arrays are host memory (header).
    const uint64_t total_ids = filter_offsets[nq] - filter_offsets[0];
    bool on_device = false;
    {
      // Consecutive ids (the bulk-loaded and append-only table: id = first id + slot, no holes): the ids go up as they
      // are and one wave per query turns them into slots on the device.  What the host did for this -- a hash lookup per
      // id on up to 16 threads, two passes, an upload of the slots -- cost 1.2 ms per batch at 1 000 ids per query next
      // to a 1.5 ms walk, 26 ms at 100 000.  The ids of the open transaction's appended rows resolve to slots past the
      // committed count and are dropped; deletes end the table's consecutiveness (the table below takes over).
      // (the id tables only change under the view lock, which this call holds shared)
      on_device = ix->n > 0 && vw.n > 0 && total_ids < (1ull << 32) && !ix->tune_host_filters;
    }
    // Any other table (deletes left holes, arbitrary ids): the same kernel probes the committed view's id -> slot
    // table, built on the device by the first filtered search of the view (sdb_index::IdMap).
    const bool by_map = on_device && !ix->dense_ids;
    if (by_map && ix->ensure_idmap(vw, stream) != SDB_OK) on_device = false;  // no memory for the table: the host's map
    if (on_device) {
      for (uint64_t q = 0; q < nq; q++)
        if (filter_offsets[q + 1] < filter_offsets[q]) return fail(SDB_ERR_INVALID, "filter_offsets must be non-decreasing");
      const uint64_t base_id = ix->h_ids[0];
      const size_t b_off = ((nq + 1) * 4 + 255) & ~(size_t)255, b_ids = (total_ids * 8 + 255) & ~(size_t)255;
      const size_t b_sl = (total_ids * 4 + 255) & ~(size_t)255, b_cnt = (nq * 4 + 255) & ~(size_t)255;
      SDB_TRY(ws->ensure_filter(b_off + b_ids + b_sl + 2 * b_cnt + 256));
      SDB_TRY(ws->ensure_filter_host(b_off + 256));
      char *fb = static_cast<char *>(ws->filter);
      uint32_t *d_off = (uint32_t *)fb;
      uint64_t *d_raw = (uint64_t *)(fb + b_off);
      uint32_t *d_sl = (uint32_t *)(fb + b_off + b_ids);
      uint32_t *d_fc = (uint32_t *)(fb + b_off + b_ids + b_sl), *d_sc = (uint32_t *)(fb + b_off + b_ids + b_sl + b_cnt);
      uint32_t *d_flags = (uint32_t *)(fb + b_off + b_ids + b_sl + 2 * b_cnt);
      uint32_t *h_off = static_cast<uint32_t *>(ws->filter_host);
      uint32_t *h_flags = h_off + (nq + 1) + 2;
      for (uint64_t q = 0; q <= nq; q++) h_off[q] = (uint32_t)(filter_offsets[q] - filter_offsets[0]);
      SDB_HIP(hipMemcpyAsync(d_off, h_off, (nq + 1) * 4, hipMemcpyHostToDevice, stream));
      // straight from the caller's array: one DMA when it is pinned (sdb_host_alloc), a staged copy otherwise
      if (total_ids) SDB_HIP(hipMemcpyAsync(d_raw, filter_ids + filter_offsets[0], total_ids * 8, hipMemcpyHostToDevice, stream));
      SDB_HIP(hipMemsetAsync(d_flags, 0, 12, stream));
      if (by_map)
        hipLaunchKernelGGL(k_filter_resolve<true>, dim3((unsigned)nq), dim3(64), 0, stream, d_raw, d_off, base_id, vw.n, search_size, d_sl,
                           d_fc, d_sc, d_flags, ix->idmap.keys, ix->idmap.vals, ix->idmap.cells - 1);
      else
        hipLaunchKernelGGL(k_filter_resolve<false>, dim3((unsigned)nq), dim3(64), 0, stream, d_raw, d_off, base_id, vw.n, search_size, d_sl,
                           d_fc, d_sc, d_flags, nullptr, nullptr, 0u);
      SDB_HIP(hipGetLastError());
      SDB_HIP(hipMemcpyAsync(h_flags, d_flags, 12, hipMemcpyDeviceToHost, stream));
      SDB_HIP(hipStreamSynchronize(stream));  // the caller's arrays are free again; an invalid filter is an error, not a search
      if (h_flags[0])
        return fail(SDB_ERR_INVALID, "filter ids of query %llu are not strictly ascending", (unsigned long long)h_flags[1]);
      a.seed_off = d_off, a.filt_off = d_off, a.seeds = d_sl, a.filt_slots = d_sl, a.seed_cnt = d_sc, a.filt_cnt = d_fc;
      // ids and slots disagree on the order somewhere (an id deleted and inserted again sits behind larger ids): the
      // seeds are the first slots in ID order either way; Contains is answered from the ids themselves then
      if (h_flags[2]) a.filt_ids = d_raw;
      a.rbitsets = ws->bitsets + (size_t)nq * words;
    }
    std::vector<uint32_t> off_seed(on_device ? 0 : nq + 1, 0), off_filt(on_device ? 0 : nq + 1, 0);
    if (!on_device) {
    // ids -> slots is a hash lookup per id; a batch of 1 024 queries with 1 000-id filters carries a million of them
    // (5 ms on one core), so big batches are split over a few host threads.  Pass 1 validates and counts, pass 2
    // fills the two CSR arrays in place.
    // one thread per 128 k ids, at most 16 (and what the machine has): a million ids take 5 ms on one core, a thread ~50 us to start
    const unsigned nthr = (unsigned)std::min<uint64_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())),
                                                       std::max<uint64_t>(1, total_ids >> 17));
    std::vector<uint32_t> n_seed(nq, 0), n_filt(nq, 0);
    std::atomic<int> bad{0};  // 1: offsets decrease, 2: ids not ascending
    std::atomic<uint64_t> bad_q{0};
    auto for_queries = [&](auto &&fn) {
      if (nthr == 1) {
        for (uint64_t q = 0; q < nq; q++) fn(q);
        return;
      }
      // a thread that cannot be had (std::system_error, or no memory for its state) must not leave joinable threads
      // behind -- their destructors would end the process: whatever did start is joined, the rest of the queries are
      // done here
      std::vector<std::thread> pool;
      unsigned started = 0;
      try {
        pool.reserve(nthr);
        for (; started < nthr; started++)
          pool.emplace_back([&, t = started] {
            for (uint64_t q = (uint64_t)nq * t / nthr; q < (uint64_t)nq * (t + 1) / nthr; q++) fn(q);
          });
      } catch (...) {
      }
      for (uint64_t q = (uint64_t)nq * started / nthr; q < nq; q++) fn(q);
      for (auto &th : pool) th.join();
    };
    for_queries([&](uint64_t q) {
      const uint64_t b = filter_offsets[q], e = filter_offsets[q + 1];
      if (e < b) {
        bad = 1;
        return;
      }
      uint32_t ns = 0, nf = 0;
      for (uint64_t i = b; i < e; i++) {
        if (i > b && filter_ids[i] <= filter_ids[i - 1]) {
          bad = 2, bad_q = q;
          return;
        }
        if (ix->slot_of_committed(filter_ids[i], vw.n) < 0) continue;  // GetMany skips unknown ids (itemcache.go:109-128)
        if (i - b < search_size) ns++;
        nf++;
      }
      n_seed[q] = ns, n_filt[q] = nf;
    });
    if (bad == 1) return fail(SDB_ERR_INVALID, "filter_offsets must be non-decreasing");
    if (bad == 2) return fail(SDB_ERR_INVALID, "filter ids of query %llu are not strictly ascending", (unsigned long long)bad_q.load());
    for (uint64_t q = 0; q < nq; q++) off_seed[q + 1] = off_seed[q] + n_seed[q], off_filt[q + 1] = off_filt[q] + n_filt[q];
    // the two lists are written where the DMA engine can take them from (pinned, kept with the workspace): a pageable
    // vector of 10^8 slots costs its zero-fill and a staged copy on top of the translation
    const size_t n_seeds = off_seed[nq], n_fslots = off_filt[nq];
    const size_t h_seeds = (n_seeds * 4 + 255) & ~(size_t)255;
    std::unique_ptr<char[]> pageable;  // when the pinned buffer cannot be had (the limit on locked memory): an ordinary one
    char *hbase = nullptr;
    if (ws->ensure_filter_host(h_seeds + n_fslots * 4 + 256) == SDB_OK) {
      hbase = static_cast<char *>(ws->filter_host);
    } else {
      (void)hipGetLastError();
      pageable.reset(new (std::nothrow) char[h_seeds + n_fslots * 4 + 256]);
      if (!pageable) return fail(SDB_ERR_DEVICE, "out of host memory for the filter lists");
      hbase = pageable.get();
    }
    uint32_t *seeds_h = reinterpret_cast<uint32_t *>(hbase);
    uint32_t *fslots_h = reinterpret_cast<uint32_t *>(hbase + h_seeds);
    for_queries([&](uint64_t q) {
      const uint64_t b = filter_offsets[q], e = filter_offsets[q + 1];
      uint32_t *sp = seeds_h + off_seed[q], *fp = fslots_h + off_filt[q];
      bool ascending = true;  // rows stored in id order (the usual case) translate to ascending slots: nothing to sort
      int64_t last = -1;
      for (uint64_t i = b; i < e; i++) {
        const int64_t s = ix->slot_of_committed(filter_ids[i], vw.n);
        if (s < 0) continue;
        if (i - b < search_size) *sp++ = (uint32_t)s;  // search.go:41-48: the first <= searchSize filter ids that exist
        *fp++ = (uint32_t)s;
        ascending &= s > last;
        last = s;
      }
      if (!ascending) std::sort(fslots_h + off_filt[q], fp);  // Contains (:93) is answered from ascending slots
    });
    const size_t b_off = (nq + 1) * 4, b_seeds = n_seeds * 4, b_f = n_fslots * 4;
    SDB_TRY(ws->ensure_filter(2 * ((b_off + 255) & ~(size_t)255) + ((b_seeds + 255) & ~(size_t)255) + b_f + 256));
    char *fb = static_cast<char *>(ws->filter);
    uint32_t *d_so = (uint32_t *)fb;
    uint32_t *d_fo = (uint32_t *)(fb + ((b_off + 255) & ~(size_t)255));
    uint32_t *d_seeds = (uint32_t *)(fb + 2 * ((b_off + 255) & ~(size_t)255));
    uint32_t *d_f = (uint32_t *)((char *)d_seeds + ((b_seeds + 255) & ~(size_t)255));
    SDB_HIP(hipMemcpyAsync(d_so, off_seed.data(), b_off, hipMemcpyHostToDevice, stream));
    SDB_HIP(hipMemcpyAsync(d_fo, off_filt.data(), b_off, hipMemcpyHostToDevice, stream));
    if (b_seeds) SDB_HIP(hipMemcpyAsync(d_seeds, seeds_h, b_seeds, hipMemcpyHostToDevice, stream));
    if (b_f) SDB_HIP(hipMemcpyAsync(d_f, fslots_h, b_f, hipMemcpyHostToDevice, stream));
    SDB_HIP(hipStreamSynchronize(stream));  // the staging vectors die with this scope
    a.seed_off = d_so, a.filt_off = d_fo, a.seeds = d_seeds, a.filt_slots = d_f;
    a.rbitsets = ws->bitsets + (size_t)nq * words;
    }
  }
  a.dim = l.dim, a.nblk = l.nblk, a.ng = l.ng, a.tail = l.tail, a.ld = l.ld;
  a.start_slot = (uint32_t)ix->start_slot;
  a.start_ext = vw.start_ext, a.start_ext_n = vw.start_ext_n;
  a.search_size = search_size, a.limit = limit, a.metric = (int)ix->P.metric;
  a.hash_limit = ix->tune_hash_limit, a.prefer_bitset = ix->tune_no_hash ? 1u : 0u;
  a.wide_hash = ix->tune_wide_hash ? 1u : 0u, a.hash16_probes = ix->tune_hash16_probes;
  a.pq_narrow = ix->tune_pq_narrow;
  a.wide_mode = ix->tune_wide_walk;
  // two-precision hop: only with the float16 copy of exactly this view's rows, outside a write transaction
  if (ix->tune_sketch && ix->d_sketch && ix->sketch_gen == ix->view_gen && !ix->in_tx && !filtered)
    a.sketch = ix->d_sketch, a.sketch_norm = ix->d_sketch_norm, a.sk_emax = ix->sk_emax, a.sk_ymax = ix->sk_ymax, a.sk_audit = ix->tune_sketch == 2 ? 1u : 0u,
    a.sk_counters = ix->d_sk_counters;

  const uint32_t vcap = trace ? trace->visit_cap : 0;
  auto launch = [&]() -> int {
    if (ix->pq) {  // fitted quantizer: DistanceFromFloat builds the M x K table first (product.go:255-263)
      const sdb_pq *pq = ix->pq;
      SDB_TRY(ws->ensure_lut((size_t)nq * pq->M * pq->K * sizeof(float)));
      SDB_TRY(pq_build_lut(pq, a.queries, nq, ws->lut, stream));
      a.pq_lut = ws->lut, a.pq_codes = ix->d_codes, a.pq_M = pq->M, a.pq_K = pq->K;
      a.pq_lut_in_lds = ((size_t)pq->M * pq->K * sizeof(float) <= 64 * 1024) ? 1u : 0u;
    }
    // ClearAll (distset.go:101); the LDS hash variant clears a bitset only for a query that overflows it
    if (!search_uses_hash(a, (uint32_t)nq))
      SDB_HIP(hipMemsetAsync(ws->bitsets, 0, filtered ? 2 * bs_bytes : bs_bytes, stream));
    const bool prof = ix->profiling && !ix->ev0.empty();
    const uint32_t slot = (uint32_t)(ix->prof_count % sdb_index::kProfRing);
    if (prof) (void)hipEventRecord(ix->ev0[slot], stream);
    int rc = launch_greedy_search(a, (uint32_t)nq, stream);
    if (prof) {
      (void)hipEventRecord(ix->ev1[slot], stream);
      ix->prof_count++;
    }
    // from here on a commit may hand the copy this batch walks to the writer: it waits for this event first
    if (!ws->launched) (void)hipEventCreateWithFlags(&ws->launched, hipEventDisableTiming);
    if (ws->launched && hipEventRecord(ws->launched, stream) == hipSuccess) ws->launched_valid = true, ws->launched_is_tail = mem == SDB_MEM_DEVICE;
    else if (rc == SDB_OK) (void)hipStreamSynchronize(stream);  // no event: finish before letting go of the version
    rl.unlock();
    return rc;
  };
  if (mem == SDB_MEM_DEVICE) {
    a.queries = queries;
    a.out_ids = out_ids, a.out_dists = out_dists, a.out_counts = out_counts;
    if (trace) {
      a.tr_ndist = trace->n_dist, a.tr_nhop = trace->n_hop, a.tr_nedges = trace->n_edges;
      a.tr_visit = trace->visit_ids, a.visit_cap = trace->visit_ids ? vcap : 0;
    }
    return launch();
  }
  // host memory.  Page-locked buffers (sdb_host_alloc, or locked by the caller) are read and written IN PLACE by the
  // walk: a wave reads its query once, when it starts (PlainDist::init), and writes its <= limit results when it ends,
  // so what crosses the bus is what a copy would have moved, without the copy packets around the kernel and without a
  // staging buffer -- the call is one launch and one wait.  (Not with a quantizer: the table kernel reads a query's
  // sub-vectors once per centroid block, and host memory is not cached on the device.  Not with a trace: test calls.)
  const bool zc_ok = !trace && !ix->tune_no_zero_copy;
  void *z_q = nullptr, *z_i = nullptr, *z_d = nullptr, *z_c = nullptr;
  const bool zc_out = zc_ok && device_view_of_host(out_ids, nq * limit * sizeof(uint64_t), &z_i) &&
                      device_view_of_host(out_dists, nq * limit * sizeof(float), &z_d) &&
                      device_view_of_host(out_counts, nq * sizeof(uint32_t), &z_c);
  const bool zc_q = zc_ok && !ix->pq && device_view_of_host(queries, nq * l.dim * sizeof(float), &z_q);
  if (zc_out && zc_q) {
    a.queries = static_cast<const float *>(z_q);
    a.out_ids = static_cast<uint64_t *>(z_i), a.out_dists = static_cast<float *>(z_d), a.out_counts = static_cast<uint32_t *>(z_c);
    SDB_TRY(launch());
    SDB_HIP(hipStreamSynchronize(stream));
    return SDB_OK;
  }
  // otherwise: stage through one scratch allocation (what is page-locked still skips its copy)
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_q = carve(nq * l.dim * sizeof(float));
  const size_t o_ids = carve(nq * limit * sizeof(uint64_t));
  const size_t o_d = carve(nq * limit * sizeof(float));
  const size_t o_c = carve(nq * sizeof(uint32_t));
  const size_t o_nd = carve(nq * sizeof(uint32_t)), o_nh = carve(nq * sizeof(uint32_t)),
               o_ne = carve(nq * sizeof(uint32_t));
  const size_t o_v = carve((trace && trace->visit_ids) ? nq * vcap * sizeof(uint64_t) : 0);
  SDB_TRY(ws->ensure_scratch(off));
  char *base = static_cast<char *>(ws->scratch);
  if (zc_q) {
    a.queries = static_cast<const float *>(z_q);
  } else {
    SDB_HIP(hipMemcpyAsync(base + o_q, queries, nq * l.dim * sizeof(float), hipMemcpyHostToDevice, stream));
    a.queries = reinterpret_cast<float *>(base + o_q);
  }
  a.out_ids = reinterpret_cast<uint64_t *>(base + o_ids);
  a.out_dists = reinterpret_cast<float *>(base + o_d);
  a.out_counts = reinterpret_cast<uint32_t *>(base + o_c);
  if (zc_out) a.out_ids = static_cast<uint64_t *>(z_i), a.out_dists = static_cast<float *>(z_d), a.out_counts = static_cast<uint32_t *>(z_c);
  if (trace) {
    a.tr_ndist = reinterpret_cast<uint32_t *>(base + o_nd);
    a.tr_nhop = reinterpret_cast<uint32_t *>(base + o_nh);
    a.tr_nedges = reinterpret_cast<uint32_t *>(base + o_ne);
    if (trace->visit_ids) a.tr_visit = reinterpret_cast<uint64_t *>(base + o_v), a.visit_cap = vcap;
  }
  SDB_HIP(hipMemsetAsync(base + o_ids, 0, o_c - o_ids, stream));
  SDB_TRY(launch());
  if (!zc_out) {
    SDB_HIP(hipMemcpyAsync(out_ids, a.out_ids, nq * limit * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipMemcpyAsync(out_dists, a.out_dists, nq * limit * sizeof(float), hipMemcpyDeviceToHost, stream));
    SDB_HIP(hipMemcpyAsync(out_counts, a.out_counts, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  }
  if (trace) {
    if (trace->n_dist) SDB_HIP(hipMemcpyAsync(trace->n_dist, a.tr_ndist, nq * 4, hipMemcpyDeviceToHost, stream));
    if (trace->n_hop) SDB_HIP(hipMemcpyAsync(trace->n_hop, a.tr_nhop, nq * 4, hipMemcpyDeviceToHost, stream));
    if (trace->n_edges) SDB_HIP(hipMemcpyAsync(trace->n_edges, a.tr_nedges, nq * 4, hipMemcpyDeviceToHost, stream));
    if (trace->visit_ids)
      SDB_HIP(hipMemcpyAsync(trace->visit_ids, a.tr_visit, nq * vcap * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  }
  SDB_HIP(hipStreamSynchronize(stream));
  return SDB_OK;
}

int sdb_index_search_batch(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit,
                           uint32_t search_size, const uint64_t *filter_offsets,
                           const uint64_t *filter_ids, uint64_t *out_ids, float *out_dists,
                           uint32_t *out_counts, const sdb_search_trace *trace, int mem, void *stream_) try {
  return search_batch_impl(ix, nq, queries, limit, search_size, filter_offsets, filter_ids, nullptr, out_ids, out_dists, out_counts,
                           trace, mem, stream_);
}
SDB_API_CATCH("sdb_index_search_batch")

int sdb_index_search_batch_bitmap(sdb_index *ix, uint64_t nq, const float *queries, uint32_t limit, uint32_t search_size,
                                  const uint64_t *filter_first_id, const uint64_t *filter_word_offsets,
                                  const uint64_t *filter_words, uint64_t *out_ids, float *out_dists, uint32_t *out_counts,
                                  const sdb_search_trace *trace, int mem, void *stream_) try {
  if (!filter_first_id || !filter_word_offsets) return fail(SDB_ERR_INVALID, "NULL filter argument");
  if (nq && filter_word_offsets[nq] > filter_word_offsets[0] && !filter_words) return fail(SDB_ERR_INVALID, "filter_words is NULL");
  const BitmapFilters bm{filter_first_id, filter_word_offsets, filter_words};
  const int rc = search_batch_impl(ix, nq, queries, limit, search_size, nullptr, nullptr, &bm, out_ids, out_dists, out_counts, trace,
                                   mem, stream_);
  if (rc != kBitmapNeedsIds) return rc;
  // a table whose ids are not consecutive (deletes, arbitrary ids): the host's hash map resolves ids, so the bitmaps
  // become id lists here -- the reference iterates its roaring bitmap the same way (search.go:41-48)
  std::vector<uint64_t> off(nq + 1, 0), ids;
  for (uint64_t q = 0; q < nq; q++) {
    if (filter_word_offsets[q + 1] < filter_word_offsets[q]) return fail(SDB_ERR_INVALID, "filter_word_offsets must be non-decreasing");
    for (uint64_t w = filter_word_offsets[q]; w < filter_word_offsets[q + 1]; w++)
      for (uint64_t word = filter_words[w]; word; word &= word - 1)
        ids.push_back(filter_first_id[q] + (w - filter_word_offsets[q]) * 64 + (uint64_t)__builtin_ctzll(word));
    off[q + 1] = ids.size();
  }
  if (ids.empty()) ids.push_back(0);
  return search_batch_impl(ix, nq, queries, limit, search_size, off.data(), ids.data(), nullptr, out_ids, out_dists, out_counts, trace,
                           mem, stream_);
}
SDB_API_CATCH("sdb_index_search_batch_bitmap")

int sdb_index_set_tuning(sdb_index *ix, int key, uint64_t value) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  switch (key) {
    case SDB_TUNE_HUB_MIN:
      if (value < 2 || value > 0xFFFFFFFFull) return fail(SDB_ERR_INVALID, "hub threshold must be at least 2");
      ix->tune_hub_min = (uint32_t)value;
      return SDB_OK;
    case SDB_TUNE_HASH_LIMIT:
      if (value > kHashLimit) return fail(SDB_ERR_INVALID, "hash limit is at most %u", kHashLimit);
      ix->tune_hash_limit = (uint32_t)value;  // 0 = default
      return SDB_OK;
    case SDB_TUNE_NO_HASH:
      ix->tune_no_hash = value != 0;
      return SDB_OK;
    case SDB_TUNE_NO_TILE:
      ix->tune_no_tile = (uint32_t)(value > 3 ? 1 : value);  // 3: tiled, selection loop inside the tiled kernel (no k_prune_select)
      return SDB_OK;
    case SDB_TUNE_NO_MFMA:
      ix->tune_no_mfma = value != 0;
      return SDB_OK;
    case SDB_TUNE_WIDE_HASH:
      ix->tune_wide_hash = value != 0;
      return SDB_OK;
    case SDB_TUNE_PQ_NARROW:
      ix->tune_pq_narrow = (uint32_t)value;
      return SDB_OK;
    case SDB_TUNE_WIDE_WALK:
      if (value > 2) return fail(SDB_ERR_INVALID, "wide walk: 0 = few queries, 1 = never, 2 = always");
      ix->tune_wide_walk = (uint32_t)value;
      return SDB_OK;
    case SDB_TUNE_HOST_FILTERS:
      ix->tune_host_filters = value != 0;
      return SDB_OK;
    case SDB_TUNE_NO_ZERO_COPY:
      ix->tune_no_zero_copy = value != 0;
      return SDB_OK;
    case SDB_TUNE_SKETCH: {
      if (value > 2) return fail(SDB_ERR_INVALID, "sketch: 0 = off, 1 = on, 2 = on with audit");
      DeviceGuard dg(ix->P.device);
      std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);
      SDB_HIP(hipDeviceSynchronize());  // walks that read the copy
      ix->tune_sketch = (uint32_t)value;
      if (!value) {
        ix->drop_sketch();
        return SDB_OK;
      }
      if (ix->d_sk_counters) SDB_HIP(hipMemset(ix->d_sk_counters, 0, 2 * sizeof(unsigned long long)));
      return ix->in_tx ? SDB_OK : ix->build_sketch(nullptr);  // inside a transaction: its commit builds it
    }
    case SDB_TUNE_NO_DEFER:
      ix->tune_no_defer = value != 0;
      return SDB_OK;
    case SDB_TUNE_HASH16_PROBES:
      if (value > 15) return fail(SDB_ERR_INVALID, "at most 15 probes");
      ix->tune_hash16_probes = (uint32_t)value;
      return SDB_OK;
    default:
      return fail(SDB_ERR_INVALID, "unknown tuning key %d", key);
  }
}
SDB_API_CATCH("sdb_index_set_tuning")

int sdb_index_sketch_stats(sdb_index *ix, uint64_t out[3]) try {
  if (!ix || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  DeviceGuard dg(ix->P.device);
  out[0] = out[1] = out[2] = 0;
  std::shared_lock<sdb::ViewMutex> rl(ix->view_mu);
  if (ix->d_sk_counters) {
    SDB_HIP(hipDeviceSynchronize());
    unsigned long long h[2] = {0, 0};
    SDB_HIP(hipMemcpy(h, ix->d_sk_counters, sizeof(h), hipMemcpyDeviceToHost));
    out[0] = h[0], out[1] = h[1];
  }
  out[2] = (ix->tune_sketch && ix->d_sketch && ix->sketch_gen == ix->view_gen && !ix->in_tx) ? 1 : 0;
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_sketch_stats")

int sdb_index_build_stats(const sdb_index *ix, uint64_t *out, uint32_t cap) try {
  if (!ix || !out) return fail(SDB_ERR_INVALID, "NULL argument");
#ifdef SDB_BACK_PROFILE  // measurement builds: the five spare slots carry k_backedges' cycle counters (tools/backprof.py)
  const uint32_t n = cap < sdb_index::kStatStride ? cap : sdb_index::kStatStride;
#else
  const uint32_t n = cap < SDB_BUILD_STATS ? cap : SDB_BUILD_STATS;
#endif
  for (uint32_t i = 0; i < n; i++) out[i] = 0;
  if (!ix->d_bstats || n == 0) return SDB_OK;
  DeviceGuard dg(ix->P.device);
  SDB_HIP(hipDeviceSynchronize());
  std::vector<uint64_t> all((size_t)sdb_index::kStatCopies * sdb_index::kStatStride);
  SDB_HIP(hipMemcpy(all.data(), ix->d_bstats, all.size() * 8, hipMemcpyDeviceToHost));
  for (uint32_t c = 0; c < sdb_index::kStatCopies; c++)
    for (uint32_t i = 0; i < n; i++) out[i] += all[(size_t)c * sdb_index::kStatStride + i];
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_build_stats")

int sdb_index_get_vectors(const sdb_index *ix, uint64_t n, const uint64_t *ids, float *out, uint8_t *found) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (n == 0) return SDB_OK;
  if (!ids || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  if (n > 0x7FFFFFFFull) return fail(SDB_ERR_INVALID, "too many ids");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  DeviceGuard dg(ix->P.device);
  const RowLayout &l = ix->lay;
  std::vector<uint32_t> slots(n);
  struct Bufs {
    uint32_t *slots = nullptr;
    float *out = nullptr;
    ~Bufs() {
      if (slots) (void)hipFree(slots);
      if (out) (void)hipFree(out);
    }
  } b;
  SDB_HIP(hipMalloc(&b.slots, n * 4));
  SDB_HIP(hipMalloc(&b.out, n * l.dim * 4));
  // the shared lock is held until the rows are on the host: compact and reserve replace the slab (and renumber the
  // rows) under the exclusive lock, and a slot resolved before that must not be read after it
  std::shared_lock<sdb::ViewMutex> rl(ix->view_mu);
  for (uint64_t i = 0; i < n; i++) {
    const int64_t s = ix->slot_of_committed(ids[i], ix->view.n);
    slots[i] = s < 0 ? kNoSlot : (uint32_t)s;
    if (found) found[i] = s < 0 ? 0 : 1;
  }
  SDB_HIP(hipMemcpy(b.slots, slots.data(), n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)n), dim3(128), 0, nullptr, ix->d_slab, b.slots, b.out, (uint32_t)n, l.dim,
                     l.nblk, l.ng, l.ld);
  SDB_HIP(hipGetLastError());
  SDB_HIP(hipMemcpy(out, b.out, n * l.dim * 4, hipMemcpyDeviceToHost));
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_get_vectors")

int sdb_index_exists_batch(const sdb_index *ix, uint64_t n, const uint64_t *ids, uint8_t *out) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (n == 0) return SDB_OK;
  if (!ids || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  std::shared_lock<sdb::ViewMutex> rl(ix->view_mu);  // a writer rehashes the id tables under the exclusive lock
  for (uint64_t i = 0; i < n; i++) out[i] = ix->slot_of(ids[i]) >= 0 ? 1 : 0;
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_exists_batch")

int sdb_index_set_profiling(sdb_index *ix, int enabled) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  DeviceGuard dg(ix->P.device);
  if (enabled && ix->ev0.empty()) {
    ix->ev0.resize(sdb_index::kProfRing, nullptr);
    ix->ev1.resize(sdb_index::kProfRing, nullptr);
    for (uint32_t i = 0; i < sdb_index::kProfRing; i++) {
      SDB_HIP(hipEventCreate(&ix->ev0[i]));
      SDB_HIP(hipEventCreate(&ix->ev1[i]));
    }
  }
  ix->profiling = enabled != 0;
  ix->prof_count = 0;
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_set_profiling")

int sdb_index_profile_read(sdb_index *ix, float *ms, uint32_t cap, uint32_t *n) try {
  if (!ix || !ms || !n) return fail(SDB_ERR_INVALID, "NULL argument");
  *n = 0;
  if (ix->ev0.empty()) return fail(SDB_ERR_STATE, "profiling was never enabled");
  DeviceGuard dg(ix->P.device);
  uint64_t have = std::min<uint64_t>(ix->prof_count, sdb_index::kProfRing);
  if (have > cap) have = cap;
  for (uint64_t k = 0; k < have; k++) {
    const uint32_t slot = (uint32_t)((ix->prof_count - have + k) % sdb_index::kProfRing);
    SDB_HIP(hipEventSynchronize(ix->ev1[slot]));
    SDB_HIP(hipEventElapsedTime(&ms[k], ix->ev0[slot], ix->ev1[slot]));
  }
  *n = (uint32_t)have;
  ix->prof_count = 0;
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_profile_read")

int sdb_index_last_search_ms(sdb_index *ix, float *ms) try {
  if (!ix || !ms) return fail(SDB_ERR_INVALID, "NULL argument");
  if (!ix->profiling || ix->prof_count == 0) return fail(SDB_ERR_STATE, "no profiled search_batch yet");
  DeviceGuard dg(ix->P.device);
  const uint32_t slot = (uint32_t)((ix->prof_count - 1) % sdb_index::kProfRing);
  SDB_HIP(hipEventSynchronize(ix->ev1[slot]));
  SDB_HIP(hipEventElapsedTime(ms, ix->ev0[slot], ix->ev1[slot]));
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_last_search_ms")

int sdb_index_size_in_memory(const sdb_index *ix, int64_t *bytes) try {
  if (!ix || !bytes) return fail(SDB_ERR_INVALID, "NULL argument");
  // vecStore.SizeInMemory + nodeStore.SizeInMemory (vamana.go:83-85), as held in HBM
  // slab row + adjacency row + its distance cache + degree / clean / cached counters + id (+ code row)
  // (+ the second adjacency / id copy of the graph versions, + the neighbours' code rows behind both adjacency copies)
  *bytes = (int64_t)ix->cap * (ix->lay.ld * 4 + 3 * kAdjStride * 4 + 3 * 4 + 2 * 8 + (ix->pq ? ix->pq->M : 0) +
                               (ix->has_adjcodes() ? 2 * kAdjStride * ix->pq->M : 0)) +
           (int64_t)ix->sketch_cap * (ix->lay.ld * 2 + 4);  // (+ the float16 copy of the rows and their norms, SDB_TUNE_SKETCH)
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_size_in_memory")

int sdb_index_stats(const sdb_index *ix, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *max_node_id) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (n_nodes) *n_nodes = ix->n - ix->n_dead;  // live nodes (start node included)
  if (max_node_id) *max_node_id = ix->max_node_id;
  if (n_edges) {
    DeviceGuard dg(ix->P.device);
    std::vector<uint32_t> deg(ix->n);
    if (ix->n) SDB_HIP(hipMemcpy(deg.data(), ix->d_deg, (size_t)ix->n * 4, hipMemcpyDeviceToHost));
    uint64_t t = 0;
    for (uint32_t d : deg) t += d;
    *n_edges = t + ix->h_start_ext.size();
  }
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_stats")

int sdb_index_row_usage(const sdb_index *ix, uint64_t *rows, uint64_t *dead) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (rows) *rows = ix->n;
  if (dead) *dead = ix->n_dead;
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_row_usage")

int sdb_index_compact(sdb_index *ix) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  if (ix->in_tx) return fail(SDB_ERR_STATE, "a write transaction is open");
  if (ix->n_dead == 0) return SDB_OK;
  DeviceGuard dg(ix->P.device);
  std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);  // searches wait: every buffer they read is replaced
  SDB_HIP(hipDeviceSynchronize());
  const uint32_t n = ix->n, cap = ix->cap, ld = ix->lay.ld;
  const uint32_t M = ix->pq ? ix->pq->M : 0;
  std::vector<uint32_t> map(n, kNoSlot), live;
  live.reserve(n - ix->n_dead);
  for (uint32_t s = 0; s < n; s++)
    if (ix->h_ids[s] != 0) map[s] = (uint32_t)live.size(), live.push_back(s);
  const uint32_t nn = (uint32_t)live.size();
  // ---- every allocation first: a failure leaves the index as it was
  struct Bufs {
    std::vector<void *> p;
    bool keep = false;
    ~Bufs() {
      if (!keep)
        for (void *x : p)
          if (x) (void)hipFree(x);
    }
    int get(void **out, size_t bytes) {
      SDB_HIP(hipMalloc(out, bytes));
      p.push_back(*out);
      return SDB_OK;
    }
  } nb, tmp;
  float *nslab = nullptr, *nad = nullptr;
  uint32_t *nadj = nullptr, *nradj = nullptr, *ndeg = nullptr, *nclean = nullptr, *ndc = nullptr;
  uint64_t *nids = nullptr, *nrids = nullptr;
  uint8_t *ncodes = nullptr;
  uint32_t *d_live = nullptr, *d_map = nullptr, *d_lost = nullptr;
  SDB_TRY(nb.get((void **)&nslab, (size_t)cap * ld * 4));
  SDB_TRY(nb.get((void **)&nadj, (size_t)cap * kAdjStride * 4));
  SDB_TRY(nb.get((void **)&nradj, (size_t)cap * kAdjStride * 4));
  SDB_TRY(nb.get((void **)&nad, (size_t)cap * kAdjStride * 4));
  SDB_TRY(nb.get((void **)&ndeg, (size_t)cap * 4));
  SDB_TRY(nb.get((void **)&nclean, (size_t)cap * 4));
  SDB_TRY(nb.get((void **)&ndc, (size_t)cap * 4));
  SDB_TRY(nb.get((void **)&nids, (size_t)cap * 8));
  SDB_TRY(nb.get((void **)&nrids, (size_t)cap * 8));
  if (M) SDB_TRY(nb.get((void **)&ncodes, (size_t)cap * M));
  uint8_t *nacw = nullptr, *nacr = nullptr;  // the neighbours' code rows follow the renumbered adjacency
  const bool had_ac = ix->has_adjcodes();
  if (had_ac && (hipMalloc((void **)&nacw, (size_t)cap * kAdjStride * M) != hipSuccess ||
                 hipMalloc((void **)&nacr, (size_t)cap * kAdjStride * M) != hipSuccess)) {
    // a cache (reserve, alloc_adjcodes): without room for its second copy the compacted index goes on without it
    (void)hipGetLastError();
    if (nacw) (void)hipFree(nacw);
    nacw = nacr = nullptr;
  }
  struct AcGuard {  // until they change hands below
    uint8_t *&a, *&b;
    bool keep = false;
    ~AcGuard() {
      if (!keep) {
        if (a) (void)hipFree(a);
        if (b) (void)hipFree(b);
      }
    }
  } ac_guard{nacw, nacr};
  SDB_TRY(tmp.get((void **)&d_live, (size_t)nn * 4 + 4));
  SDB_TRY(tmp.get((void **)&d_map, (size_t)n * 4));
  SDB_TRY(tmp.get((void **)&d_lost, 4));
  SDB_HIP(hipMemcpy(d_live, live.data(), (size_t)nn * 4, hipMemcpyHostToDevice));
  SDB_HIP(hipMemcpy(d_map, map.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  SDB_HIP(hipMemset(d_lost, 0, 4));
  SDB_HIP(hipMemset(nadj, 0xFF, (size_t)cap * kAdjStride * 4));
  SDB_HIP(hipMemset(ndeg, 0, (size_t)cap * 4));
  SDB_HIP(hipMemset(nclean, 0, (size_t)cap * 4));
  SDB_HIP(hipMemset(ndc, 0, (size_t)cap * 4));
  hipLaunchKernelGGL(k_compact_rows, dim3(nn), dim3(64), 0, nullptr, d_live, d_map, ld, ix->d_slab, nslab, ix->d_adj, nadj,
                     ix->d_adjdist, nad, ix->d_deg, ndeg, ix->d_clean, nclean, ix->d_dcount, ndc, ix->d_ids, nids,
                     ix->d_codes, ncodes, M, d_lost);
  SDB_HIP(hipGetLastError());
  SDB_HIP(hipMemcpy(nradj, nadj, (size_t)cap * kAdjStride * 4, hipMemcpyDeviceToDevice));
  SDB_HIP(hipMemcpy(nrids, nids, (size_t)nn * 8, hipMemcpyDeviceToDevice));
  if (nacw && nn) {
    hipLaunchKernelGGL(k_adjcodes_rows, dim3((nn + 3) / 4), dim3(256), 0, nullptr, nadj, ncodes, nacw, nullptr, nn, 0u, M);
    SDB_HIP(hipGetLastError());
    SDB_HIP(hipMemcpy(nacr, nacw, (size_t)nn * kAdjStride * M, hipMemcpyDeviceToDevice));
  }
  uint32_t lost = 0;
  SDB_HIP(hipMemcpy(&lost, d_lost, 4, hipMemcpyDeviceToHost));
  if (lost) return fail(SDB_ERR_STATE, "%u edges point at deleted rows: the graph is inconsistent, nothing was changed", lost);
  // the host tables of the compacted index, complete before anything changes hands (their memory may not be there)
  std::vector<uint64_t> h(nn);
  bool dense = true;
  for (uint32_t j = 0; j < nn; j++) {
    h[j] = ix->h_ids[live[j]];
    if (j && h[j] != h[0] + j) dense = false;
  }
  std::unordered_map<uint64_t, uint32_t> nmap;
  if (!dense) {
    nmap.reserve((size_t)nn * 2);
    for (uint32_t j = 0; j < nn; j++) nmap.emplace(h[j], j);
  }
  std::vector<uint32_t> padded((ix->h_start_ext.size() + 63) / 64 * 64, kNoSlot);
  // ---- swap in (nothing below can fail)
  for (void *x : {(void *)ix->d_slab, (void *)ix->d_adj, (void *)ix->r_adj, (void *)ix->d_adjdist, (void *)ix->d_deg,
                  (void *)ix->d_clean, (void *)ix->d_dcount, (void *)ix->d_ids, (void *)ix->r_ids})
    (void)hipFree(x);
  if (M) (void)hipFree(ix->d_codes);
  if (had_ac) (void)hipFree(ix->d_adjcodes), (void)hipFree(ix->r_adjcodes), ix->d_adjcodes = nacw, ix->r_adjcodes = nacr;  // (NULL: dropped)
  ac_guard.keep = true;
  nb.keep = true;
  ix->d_slab = nslab, ix->d_adj = nadj, ix->r_adj = nradj, ix->d_adjdist = nad, ix->d_deg = ndeg, ix->d_clean = nclean;
  ix->d_dcount = ndc, ix->d_ids = nids, ix->r_ids = nrids;
  if (M) ix->d_codes = ncodes;
  (void)hipMemset(ix->d_dirty, 0, cap);
  ix->h_ids.swap(h);
  ix->id2slot.swap(nmap);
  ix->dense_ids = dense;
  if (ix->start_slot >= 0) ix->start_slot = (int64_t)map[(uint32_t)ix->start_slot];  // a flat index has none
  for (auto &t : ix->h_start_ext) t = map[t];
  ix->n = nn, ix->n_dead = 0;
  const uint32_t need = (uint32_t)((ix->h_start_ext.size() + 63) / 64 * 64);
  if (need) {  // both copies of the overflow list, renumbered
    std::copy(ix->h_start_ext.begin(), ix->h_start_ext.end(), padded.begin());
    (void)hipMemcpy(ix->d_start_ext, padded.data(), (size_t)need * 4, hipMemcpyHostToDevice);
    if (need > ix->r_start_ext_cap) {
      if (ix->r_start_ext) (void)hipFree(ix->r_start_ext);
      ix->r_start_ext = nullptr, ix->r_start_ext_cap = 0;
      if (hipMalloc(&ix->r_start_ext, (size_t)need * 2 * 4) == hipSuccess) ix->r_start_ext_cap = need * 2;
    }
    if (ix->r_start_ext) (void)hipMemcpy(ix->r_start_ext, padded.data(), (size_t)need * 4, hipMemcpyHostToDevice);
  }
  ix->view.n = nn, ix->view.adj = ix->r_adj, ix->view.ids = ix->r_ids, ix->view.start_ext = ix->r_start_ext;
  ix->view.start_ext_n = (uint32_t)ix->h_start_ext.size();
  ix->view.adj_codes = ix->r_adjcodes;
  ix->view_gen++;
  (void)hipDeviceSynchronize();
  if (ix->tune_sketch) (void)ix->build_sketch(nullptr);  // rows have moved (a stale copy is never used: sketch_gen)
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_compact")

int sdb_index_export(const sdb_index *ix, uint64_t *ids, float *vectors, uint64_t *offsets, uint64_t *edges) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (ix->broken) return fail(SDB_ERR_STATE, "index is unusable after a failed write; reload it from the bucket");
  DeviceGuard dg(ix->P.device);
  SDB_HIP(hipDeviceSynchronize());
  const uint32_t n = ix->n;
  // deleted nodes (tombstones: id 0) are gone from the bucket (node.go:129-134); live rows keep their order
  if (ids) {
    uint64_t k = 0;
    for (uint32_t i = 0; i < n; i++)
      if (ix->h_ids[i] != 0) ids[k++] = ix->h_ids[i];
  }
  if (offsets || edges) {
    std::vector<uint32_t> adj((size_t)n * kAdjStride), deg(n);
    if (n) {
      SDB_HIP(hipMemcpy(adj.data(), ix->d_adj, adj.size() * 4, hipMemcpyDeviceToHost));
      SDB_HIP(hipMemcpy(deg.data(), ix->d_deg, deg.size() * 4, hipMemcpyDeviceToHost));
    }
    uint64_t o = 0, k = 0;
    for (uint32_t i = 0; i < n; i++) {
      if (ix->h_ids[i] == 0) continue;
      if (offsets) offsets[k] = o;
      for (uint32_t e = 0; e < deg[i]; e++, o++)
        if (edges) edges[o] = ix->h_ids[adj[(size_t)i * kAdjStride + e]];
      if ((int64_t)i == ix->start_slot)
        for (uint32_t t : ix->h_start_ext) {
          if (edges) edges[o] = ix->h_ids[t];
          o++;
        }
      k++;
    }
    if (offsets) offsets[k] = o;
  }
  if (vectors && n) {
    const RowLayout &l = ix->lay;
    float *tmp = nullptr;
    SDB_HIP(hipMalloc(&tmp, (size_t)n * l.dim * sizeof(float)));
    hipLaunchKernelGGL(k_unpermute_rows, dim3(n), dim3(128), 0, nullptr, ix->d_slab, tmp, n, l.dim, l.nblk, l.ng, l.ld);
    hipError_t e = hipSuccess;
    if (ix->n_dead == 0) {
      e = hipMemcpy(vectors, tmp, (size_t)n * l.dim * sizeof(float), hipMemcpyDeviceToHost);
    } else {
      std::vector<float> all((size_t)n * l.dim);
      e = hipMemcpy(all.data(), tmp, all.size() * sizeof(float), hipMemcpyDeviceToHost);
      uint64_t k = 0;
      for (uint32_t i = 0; i < n && e == hipSuccess; i++)
        if (ix->h_ids[i] != 0) memcpy(vectors + (k++) * l.dim, all.data() + (size_t)i * l.dim, l.dim * sizeof(float));
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(SDB_ERR_DEVICE, "D2H copy failed: %s", hipGetErrorString(e));
  }
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_export")

}  // extern "C"

namespace sdb {
int store_rows_public(sdb_index *ix, uint32_t first, uint32_t n, const float *dev_vectors, hipStream_t stream) {
  return store_rows(ix, first, n, dev_vectors, SDB_MEM_DEVICE, stream);
}
}  // namespace sdb

namespace sdb {
int unpermute_rows_public(const sdb_index *ix, uint32_t first, uint32_t n, float *dst, hipStream_t stream) {
  if (n == 0) return SDB_OK;
  const RowLayout &l = ix->lay;
  hipLaunchKernelGGL(k_unpermute_rows, dim3(n), dim3(128), 0, stream, ix->d_slab + (size_t)first * l.ld, dst, n, l.dim,
                     l.nblk, l.ng, l.ld);
  SDB_HIP(hipGetLastError());
  return SDB_OK;
}
}  // namespace sdb

// productQuantizer.Set -> encode for every stored vector (product.go:161-169), then searches use the
// LUT distance (product.go:250-277).
// The write path keeps two things per row from its last robustPrune: how many leading edges came out of it
// (d_clean: those do not dominate each other, build.hip) and the distances to them (d_dcount / d_adjdist).  Both
// are statements about the store's distance function.  When that changes -- the store switches to the quantizer's
// table distances, or centroid ids are overwritten -- they no longer hold and every row starts over.
static int forget_prune_state(sdb_index *ix) {
  if (ix->n == 0) return SDB_OK;
  SDB_HIP(hipMemset(ix->d_clean, 0, (size_t)ix->n * sizeof(uint32_t)));
  SDB_HIP(hipMemset(ix->d_dcount, 0, (size_t)ix->n * sizeof(uint32_t)));
  return SDB_OK;
}

extern "C" int sdb_index_attach_pq(sdb_index *ix, const sdb_pq *pq, void *stream_) try {
  if (!ix || !pq) return fail(SDB_ERR_INVALID, "NULL argument");
  if (!pq->fitted) return fail(SDB_ERR_STATE, "quantizer is not fitted");
  if (pq->dim != ix->lay.dim) return fail(SDB_ERR_INVALID, "quantizer dim %u != index dim %u", pq->dim, ix->lay.dim);
  if (pq->device != ix->P.device) return fail(SDB_ERR_INVALID, "quantizer and index live on different devices");
  {
    // the reference swaps cosine for euclidean inside the quantizer only (product.go:52-61); any other
    // mismatch between the index metric and the quantizer metric is a configuration error
    const int want = ix->P.metric == SDB_METRIC_COSINE ? SDB_METRIC_EUCLIDEAN : (int)ix->P.metric;
    if (pq->metric != want) return fail(SDB_ERR_INVALID, "quantizer metric does not match the index metric");
  }
  if (ix->in_tx) return fail(SDB_ERR_STATE, "a write transaction is open");
  DeviceGuard dg(ix->P.device);
  hipStream_t stream = as_stream(stream_);
  // The hosts run Fit after a commit while their batcher keeps searching.  The new code rows are encoded into a
  // buffer of their own while searches go on with what they have (full precision, or the previous quantizer's
  // codes); then (quantizer, codes) change hands together under the exclusive lock, once the walks that were
  // launched with the old pair have drained -- a search sees either pair whole, never the new tables over
  // half-written codes, never freed rows.
  uint8_t *ncodes = nullptr;
  SDB_HIP(hipMalloc(&ncodes, (size_t)ix->cap * pq->M));
  const uint32_t chunk = 1u << 18;
  float *tmp = nullptr;
  if (hipMalloc(&tmp, (size_t)std::min<uint32_t>(chunk, std::max<uint32_t>(ix->n, 1)) * ix->lay.dim * sizeof(float)) != hipSuccess) {
    (void)hipFree(ncodes);
    return fail(SDB_ERR_DEVICE, "out of device memory for the encode staging");
  }
  int rc = SDB_OK;
  for (uint32_t first = 0; first < ix->n && rc == SDB_OK; first += chunk) {
    const uint32_t m = std::min<uint32_t>(chunk, ix->n - first);
    rc = unpermute_rows_public(ix, first, m, tmp, stream);
    if (rc == SDB_OK) rc = pq_encode_device(pq, tmp, m, ncodes + (size_t)first * pq->M, stream);
  }
  (void)hipStreamSynchronize(stream);
  (void)hipFree(tmp);
  if (rc != SDB_OK) {
    (void)hipFree(ncodes);
    return rc;
  }
  std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);
  (void)hipDeviceSynchronize();  // walks launched with the old (quantizer, codes) pair
  if (ix->d_codes) (void)hipFree(ix->d_codes);
  ix->d_codes = ncodes;
  ix->pq = pq;
  // the neighbours' code rows behind the adjacency rows, for the new codes (M <= 32; index.h d_adjcodes)
  SDB_TRY(ix->alloc_adjcodes());
  SDB_TRY(ix->rebuild_adjcodes(stream));
  return forget_prune_state(ix);
}
SDB_API_CATCH("sdb_index_attach_pq")

// Centroid ids that do NOT come from encode(): the k-means labels productQuantizer.Fit leaves on its
// training points (product.go:216-218) and the codes a bucket persisted under 'q' (product.go:349-383).
extern "C" int sdb_index_set_codes(sdb_index *ix, uint64_t n, const uint64_t *ids, const uint8_t *codes) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (!ix->pq) return fail(SDB_ERR_STATE, "no quantizer attached");
  // (code rows exist once, not per graph version: rewritten inside a transaction they could not be rolled back with it)
  if (ix->in_tx) return fail(SDB_ERR_STATE, "a write transaction is open: centroid ids are set outside it");
  if (n == 0) return SDB_OK;
  if (!ids || !codes) return fail(SDB_ERR_INVALID, "NULL argument");
  const uint32_t M = ix->pq->M;
  DeviceGuard dg(ix->P.device);
  for (uint64_t i = 0; i < n; i++)  // nothing is written unless every id resolves
    if (ix->slot_of(ids[i]) < 0) return fail(SDB_ERR_NOT_FOUND, "point %llu not found", (unsigned long long)ids[i]);
  // code rows are rewritten in place: searches stand aside (exclusive lock) and the walks in flight drain first
  std::unique_lock<sdb::ViewMutex> wl(ix->view_mu);
  SDB_HIP(hipDeviceSynchronize());
  // group runs of consecutive slots into one copy each (ids in storage order give one run)
  uint64_t i = 0;
  while (i < n) {
    const int64_t s0 = ix->slot_of(ids[i]);
    uint64_t j = i + 1;
    while (j < n && ix->slot_of(ids[j]) == s0 + (int64_t)(j - i)) j++;
    SDB_HIP(hipMemcpy(ix->d_codes + (size_t)s0 * M, codes + i * M, (j - i) * M, hipMemcpyHostToDevice));
    i = j;
  }
  SDB_TRY(ix->rebuild_adjcodes(nullptr));  // every node that has one of these as a neighbour carries a copy of its code row
  return forget_prune_state(ix);
}
SDB_API_CATCH("sdb_index_set_codes")

extern "C" int sdb_index_get_codes(const sdb_index *ix, uint64_t n, const uint64_t *ids, uint8_t *codes) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (!ix->pq) return fail(SDB_ERR_STATE, "no quantizer attached");
  if (n == 0) return SDB_OK;
  if (!ids || !codes) return fail(SDB_ERR_INVALID, "NULL argument");
  const uint32_t M = ix->pq->M;
  DeviceGuard dg(ix->P.device);
  uint64_t i = 0;
  while (i < n) {
    const int64_t s0 = ix->slot_of(ids[i]);
    if (s0 < 0) return fail(SDB_ERR_NOT_FOUND, "point %llu not found", (unsigned long long)ids[i]);
    uint64_t j = i + 1;
    while (j < n && ix->slot_of(ids[j]) == s0 + (int64_t)(j - i)) j++;
    SDB_HIP(hipMemcpy(codes + i * M, ix->d_codes + (size_t)s0 * M, (j - i) * M, hipMemcpyDeviceToHost));
    i = j;
  }
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_get_codes")
