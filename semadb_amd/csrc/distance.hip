// distance.hip -- K1: batched query x candidate distance (L2 / cosine / dot).
//
// Replaces distance.FloatDistFunc (distance/distance.go:11,19-25,70-83) on its AVX2 path
// (distance/distance_amd64.go:19-27 -> distance/asm/dot.s, euclidean.s).  The query tile is
// staged in LDS, candidate rows are read with coalesced 128-byte (original layout) or 512-byte
// (slab layout) half-wave loads, and the 32 partial sums are reduced with wave shuffles in the
// assembly's order (dist_core.h), so every distance is bit-identical to the reference's.
#include <cfloat>

#include "search_kernel.h"

namespace sdb {

// ---- original-layout inputs: out[q][c] = dist(queries[q], cands[c]) -------------------------
// grid (ctiles, nq), block 256 = 8 half-waves; half-wave h takes candidates h, h+8, ... of the tile.
constexpr int kK1CandPerBlock = 64;

template <bool L2>
__global__ __launch_bounds__(256) void k_distance_batch(const float *__restrict__ queries,
                                                        const float *__restrict__ cands, float *__restrict__ out,
                                                        uint32_t dim, uint64_t nc, int metric) {
  extern __shared__ float qs[];  // the query tile: dim floats
  const uint32_t q = blockIdx.y;
  const float *qv = queries + (size_t)q * dim;
  for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) qs[i] = qv[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, L = lane & 31;
  const int hw = threadIdx.x >> 5;  // 0..7
  const uint32_t nblk = dim / 32, tail = dim % 32;
  const uint64_t c0 = (uint64_t)blockIdx.x * kK1CandPerBlock;
  for (int i = hw; i < kK1CandPerBlock; i += 8) {
    uint64_t c = c0 + i;
    const bool live = c < nc;  // uniform per half-wave; shuffles below need the whole wave
    if (!live) c = nc - 1;
    const float *__restrict__ y = cands + c * dim;
    float acc = 0.0f;
    uint32_t b = 0;
    for (; b + 4 <= nblk; b += 4) {  // four independent loads in flight per lane
      float y0 = y[32 * b + L], y1 = y[32 * (b + 1) + L], y2 = y[32 * (b + 2) + L], y3 = y[32 * (b + 3) + L];
      acc = chain1<L2>(acc, qs[32 * b + L], y0);
      acc = chain1<L2>(acc, qs[32 * (b + 1) + L], y1);
      acc = chain1<L2>(acc, qs[32 * (b + 2) + L], y2);
      acc = chain1<L2>(acc, qs[32 * (b + 3) + L], y3);
    }
    for (; b < nblk; b++) acc = chain1<L2>(acc, qs[32 * b + L], y[32 * b + L]);
    float t = 0.0f;
    if (tail) {
      float xt = (uint32_t)L < tail ? qs[nblk * 32 + L] : 0.0f;
      float yt = (uint32_t)L < tail ? y[nblk * 32 + L] : 0.0f;
      t = tail_chain<L2>(xt, yt, tail, lane);
    }
    float r = asm_reduce(acc, t, lane);
    if (L == 0 && live) out[(size_t)q * nc + c] = metric_finish(r, metric);
  }
}

// ---- slab-layout candidates by slot: plainStore.DistanceFromFloat (plain.go:76-85) batched ------
// one wave per (query, chunk of 2*U candidates)
template <bool L2>
__global__ __launch_bounds__(64) void k_index_distance(const float *__restrict__ slab,
                                                       const float *__restrict__ queries,
                                                       const uint32_t *__restrict__ slots, float *__restrict__ out,
                                                       uint32_t dim, uint32_t nblk, uint32_t ng, uint32_t tail,
                                                       uint32_t ld, uint64_t nc, int metric) {
  constexpr int U = 4;
  extern __shared__ float qs[];
  const int lane = threadIdx.x, L = lane & 31;
  const uint32_t q = blockIdx.y;
  const float *qv = queries + (size_t)q * dim;
  for (uint32_t i = lane; i < ng * 128; i += 64) {
    uint32_t g = i / 128, r = i % 128;
    qs[i] = q_elem(qv, nblk, g, r % 4, (int)(r / 4));
  }
  if (tail && lane < 32) qs[ng * 128 + lane] = (uint32_t)lane < tail ? qv[nblk * 32 + lane] : 0.0f;
  __syncthreads();
  const uint64_t c0 = (uint64_t)blockIdx.x * (2 * U);
  uint32_t slot[U];
  bool known[U];
  uint64_t cidx[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    uint64_t c = c0 + 2 * u + (lane >> 5);
    cidx[u] = c;
    uint32_t s = c < nc ? slots[(size_t)q * nc + c] : kNoSlot;
    known[u] = s != kNoSlot;
    slot[u] = known[u] ? s : 0;
  }
  float res[U];
  chunk_dist_lds<L2, U>(slab, ld, ng, tail, qs, slot, res, lane);
#pragma unroll
  for (int u = 0; u < U; u++)
    if (L == 0 && cidx[u] < nc)  // unknown point -> math.MaxFloat32 (plain.go:78-82)
      out[(size_t)q * nc + cidx[u]] = known[u] ? metric_finish(res[u], metric) : FLT_MAX;
}

}  // namespace sdb

namespace sdb {
int launch_k1_tiles(int metric, uint32_t dim, const float *dq, uint64_t nq, const float *dc, uint64_t nc, float *dout,
                    hipStream_t stream);

// Stream-ordered scratch (hipMallocAsync) instead of a hipMalloc / hipFree pair per call: the device's default pool
// is told to keep what it is given back, so that a serving process pays for staging memory once.
void keep_pool_memory(int device) {
  static std::atomic<uint64_t> done{0};
  const uint64_t bit = 1ull << (device & 63);
  if (done.fetch_or(bit) & bit) return;
  hipMemPool_t pool = nullptr;
  if (hipDeviceGetDefaultMemPool(&pool, device) != hipSuccess || !pool) return;
  uint64_t keep = 1ull << 30;  // up to 1 GB of idle scratch stays with the process
  (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
}
}  // namespace sdb

using namespace sdb;

extern "C" {

int sdb_distance_batch(int metric, uint32_t dim, const float *queries, uint64_t nq, const float *candidates,
                       uint64_t nc, float *out, int mem, int device, void *stream_) try {
  if (metric < 0 || metric > SDB_METRIC_DOT) return fail(SDB_ERR_INVALID, "unknown float32 distance function: %d", metric);
  if (dim < 1 || dim > 4096) return fail(SDB_ERR_INVALID, "vector size must be between 1 and 4096, got %u", dim);
  if (nq == 0 || nc == 0) return SDB_OK;
  if (!queries || !candidates || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  if (nq > 65535) return fail(SDB_ERR_INVALID, "at most 65535 queries per call, got %llu", (unsigned long long)nq);
  int ndev = 0;
  SDB_TRY(sdb_device_count(&ndev));
  if (device < 0 || device >= ndev) return fail(SDB_ERR_INVALID, "device %d out of range", device);
  DeviceGuard dg(device);
  hipStream_t stream = as_stream(stream_);
  keep_pool_memory(device);
  const float *dq = queries, *dc = candidates;
  float *dout = out;
  float *buf = nullptr;
  if (mem == SDB_MEM_HOST) {
    // staging from the stream-ordered pool (no hipMalloc / hipFree per call); the three parts start on 256-byte
    // boundaries so that the tile kernels' 16-byte loads are aligned whatever the shapes
    const size_t bq = (nq * dim * sizeof(float) + 255) & ~(size_t)255, bc = (nc * dim * sizeof(float) + 255) & ~(size_t)255;
    const size_t bo = nq * nc * sizeof(float);
    SDB_HIP(hipMallocAsync(reinterpret_cast<void **>(&buf), bq + bc + bo, stream));
    float *pq = buf, *pc = reinterpret_cast<float *>(reinterpret_cast<char *>(buf) + bq);
    float *po = reinterpret_cast<float *>(reinterpret_cast<char *>(buf) + bq + bc);
    hipError_t e = hipMemcpyAsync(pq, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pc, candidates, nc * dim * sizeof(float), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) {
      (void)hipFreeAsync(buf, stream);
      return fail(SDB_ERR_DEVICE, "H2D copy failed: %s", hipGetErrorString(e));
    }
    dq = pq, dc = pc, dout = po;
  }
  // row reuse (distance_tile.hip): a workgroup keeps 64 candidate rows and walks all queries over them; shapes it does
  // not cover (a single query, rows shorter than a block or not whole float4s, unaligned callers) take the first kernel
  hipError_t e = hipSuccess;
  const int tiled = launch_k1_tiles(metric, dim, dq, nq, dc, nc, dout, stream);
  if (tiled < 0) {
    if (buf) (void)hipFreeAsync(buf, stream);
    return -tiled;
  }
  if (tiled == 0) {
    dim3 grid((unsigned)((nc + kK1CandPerBlock - 1) / kK1CandPerBlock), (unsigned)nq);
    size_t lds = dim * sizeof(float);
    if (metric == SDB_METRIC_EUCLIDEAN)
      hipLaunchKernelGGL(k_distance_batch<true>, grid, dim3(256), lds, stream, dq, dc, dout, dim, nc, metric);
    else
      hipLaunchKernelGGL(k_distance_batch<false>, grid, dim3(256), lds, stream, dq, dc, dout, dim, nc, metric);
    e = hipGetLastError();
  }
  if (e == hipSuccess && mem == SDB_MEM_HOST) {
    e = hipMemcpyAsync(out, dout, nq * nc * sizeof(float), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  if (buf) (void)hipFreeAsync(buf, stream);
  if (e != hipSuccess) return fail(SDB_ERR_DEVICE, "distance_batch failed: %s", hipGetErrorString(e));
  return SDB_OK;
}
SDB_API_CATCH("sdb_distance_batch")

int sdb_index_distance_batch(sdb_index *ix, uint64_t nq, const float *queries, uint64_t nc,
                             const uint64_t *cand_ids, float *out, int mem, void *stream_) try {
  if (!ix) return fail(SDB_ERR_INVALID, "index is NULL");
  if (nq == 0 || nc == 0) return SDB_OK;
  if (!queries || !cand_ids || !out) return fail(SDB_ERR_INVALID, "NULL argument");
  if (nq > 65535) return fail(SDB_ERR_INVALID, "at most 65535 queries per call");
  DeviceGuard dg(ix->P.device);
  hipStream_t stream = as_stream(stream_);
  const RowLayout &l = ix->lay;
  std::vector<uint32_t> slots(nq * nc);
  // a read like a search: ids resolve against the committed state (index.h graph versions); the slab rows
  // themselves never change once written.  The lock is kept until the kernel is enqueued (reserve() swaps
  // the slab pointer under it).
  std::shared_lock<sdb::ViewMutex> rl(ix->view_mu);
  const uint32_t view_n = ix->view.n;
  for (size_t i = 0; i < slots.size(); i++) {
    int64_t s = ix->slot_of_committed(cand_ids[i], view_n);
    slots[i] = s < 0 ? kNoSlot : (uint32_t)s;
  }
  uint32_t *dslots = nullptr;
  float *dq = nullptr, *dout = nullptr;
  SDB_HIP(hipMalloc(&dslots, slots.size() * 4));
  hipError_t e = hipMemcpyAsync(dslots, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, stream);
  const float *q = queries;
  float *o = out;
  if (e == hipSuccess && mem == SDB_MEM_HOST) {
    e = hipMalloc(&dq, nq * l.dim * 4);
    if (e == hipSuccess) e = hipMalloc(&dout, nq * nc * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(dq, queries, nq * l.dim * 4, hipMemcpyHostToDevice, stream);
    q = dq, o = dout;
  }
  if (e == hipSuccess) {
    dim3 grid((unsigned)((nc + 7) / 8), (unsigned)nq);
    size_t lds = (size_t)(l.ng * 128 + 32) * sizeof(float);
    if (ix->P.metric == SDB_METRIC_EUCLIDEAN)
      hipLaunchKernelGGL(k_index_distance<true>, grid, dim3(64), lds, stream, ix->d_slab, q, dslots, o, l.dim,
                         l.nblk, l.ng, l.tail, l.ld, nc, (int)ix->P.metric);
    else
      hipLaunchKernelGGL(k_index_distance<false>, grid, dim3(64), lds, stream, ix->d_slab, q, dslots, o, l.dim,
                         l.nblk, l.ng, l.tail, l.ld, nc, (int)ix->P.metric);
    e = hipGetLastError();
  }
  if (e == hipSuccess && mem == SDB_MEM_HOST)
    e = hipMemcpyAsync(out, dout, nq * nc * 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);  // dslots is freed below
  (void)hipFree(dslots);
  if (dq) (void)hipFree(dq);
  if (dout) (void)hipFree(dout);
  if (e != hipSuccess) return fail(SDB_ERR_DEVICE, "index_distance_batch failed: %s", hipGetErrorString(e));
  return SDB_OK;
}
SDB_API_CATCH("sdb_index_distance_batch")

}  // extern "C"
