// gather_probe.hip -- measurement only (not part of libsemadb_amd.so): what the memory system delivers for
// the search kernel's access shape with every dependency removed.  Each wavefront reads `iters` chunks of
// 2*U random slab rows (half-wave per row, float4 per lane and 128-float group, exactly K2's chunk_dist
// shape) and folds them into one float so the loads stay live.  No adjacency fetch, no visited set, no
// candidate list: the result is the practical ceiling for "gather random rows of NG*512 bytes".
#include <hip/hip_runtime.h>
#include <stdint.h>

// think: idle time after each chunk in units of 64 clocks (a wave that does something else between chunks);
// shared_pct: percentage of chunks whose rows are the same for EVERY wave of the launch (all queries of a
// batch walking through the same hub nodes at about the same time).
__device__ uint32_t g_think = 0, g_shared_pct = 0;

template <int NG, int U>
__global__ __launch_bounds__(64) void k_gather_probe(const float *__restrict__ slab, uint32_t n_rows, uint32_t ld,
                                                     uint32_t iters, float *__restrict__ out) {
  const uint32_t think = g_think, shared_pct = g_shared_pct;
  const int lane = threadIdx.x, L = lane & 31, half = lane >> 5;
  uint32_t state = (blockIdx.x + 1) * 2654435761u;
  float acc = 0.0f;
  for (uint32_t it = 0; it < iters; it++) {
    float4 y[U][NG];
    const bool shared = ((it * 2654435761u) >> 16) % 100u < shared_pct;
    uint32_t sstate = (it + 1) * 0x85EBCA6Bu;
#pragma unroll
    for (int u = 0; u < U; u++) {
      state = state * 1664525u + 1013904223u;  // wave-uniform LCG; the two halves take different rows
      sstate = sstate * 1664525u + 1013904223u;
      const uint32_t st = shared ? sstate : state;
      uint32_t x = st + half * 0x9E3779B9u;  // full 32-bit mix (murmur3 finalizer): rows uniform over the slab
      x ^= x >> 16, x *= 0x85EBCA6Bu, x ^= x >> 13, x *= 0xC2B2AE35u, x ^= x >> 16;
      const uint32_t r = (uint32_t)(((uint64_t)x * n_rows) >> 32);
      const float4 *row = reinterpret_cast<const float4 *>(slab + (size_t)r * ld);
#pragma unroll
      for (int g = 0; g < NG; g++) y[u][g] = row[g * 32 + L];
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int g = 0; g < NG; g++) acc += (y[u][g].x + y[u][g].y) + (y[u][g].z + y[u][g].w);
    for (uint32_t t = 0; t < think; t++) __builtin_amdgcn_s_sleep(1);
  }
  if (acc == 12345.678f) out[blockIdx.x * 64 + lane] = acc;  // never true for real data; keeps the loads
}

template <int NG>
static int launch(const float *slab, uint32_t n_rows, uint32_t ld, uint32_t waves, uint32_t iters, int U, float *out,
                  hipStream_t s) {
  if (U == 16) hipLaunchKernelGGL((k_gather_probe<NG, 16>), dim3(waves), dim3(64), 0, s, slab, n_rows, ld, iters, out);
  else if (U == 8) hipLaunchKernelGGL((k_gather_probe<NG, 8>), dim3(waves), dim3(64), 0, s, slab, n_rows, ld, iters, out);
  else hipLaunchKernelGGL((k_gather_probe<NG, 4>), dim3(waves), dim3(64), 0, s, slab, n_rows, ld, iters, out);
  return (int)hipGetLastError();
}

extern "C" int gather_probe_config(uint32_t think, uint32_t shared_pct) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_think), &think, 4) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_shared_pct), &shared_pct, 4) != hipSuccess) return -1;
  return 0;
}

// returns the kernel time in ms (HIP events on `stream`), < 0 on error; rows read = waves * iters * 2 * U
extern "C" float gather_probe(const float *slab, uint32_t n_rows, uint32_t ld, uint32_t ng, uint32_t waves,
                              uint32_t iters, int U, float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0f;
  (void)hipEventRecord(e0, s);
  int rc;
  switch (ng) {
    case 1: rc = launch<1>(slab, n_rows, ld, waves, iters, U, out, s); break;
    case 2: rc = launch<2>(slab, n_rows, ld, waves, iters, U, out, s); break;
    case 3: rc = launch<3>(slab, n_rows, ld, waves, iters, U, out, s); break;
    case 6: rc = launch<6>(slab, n_rows, ld, waves, iters, U > 8 ? 8 : U, out, s); break;
    default: return -2.0f;
  }
  (void)hipEventRecord(e1, s);
  if (rc != 0 || hipEventSynchronize(e1) != hipSuccess) return -3.0f;
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  return ms;
}

// ---- dependent-fetch probe: what ONE dependent HBM round trip costs a lone lane ------------------------------------
// A graph walk's hop begins with a fetch whose address the previous hop produced.  Lane 0 of each wave chases through
// `buf`: the next address is a hash of the word just loaded, so no two loads overlap.  One wave = the unloaded figure;
// many waves = the same under the load of a batch.  Returns nanoseconds per dependent load (s_memrealtime: the constant
// 100 MHz counter -- s_memtime counts shader clocks on gfx950), < 0 on error.
__global__ __launch_bounds__(64) void k_chase(const uint32_t *__restrict__ buf, uint64_t words, uint32_t steps,
                                              unsigned long long *__restrict__ ticks, uint32_t *__restrict__ sink) {
  if (threadIdx.x != 0) return;
  uint64_t at = ((uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull) % words;
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < steps; i++) {
    const uint32_t v = __builtin_nontemporal_load(buf + at);
    acc += v;
    uint64_t x = ((uint64_t)v << 32 | i) + at;
    x ^= x >> 30, x *= 0xbf58476d1ce4e5b9ull, x ^= x >> 27, x *= 0x94d049bb133111ebull, x ^= x >> 31;
    at = (x % words) & ~15ull;  // 64-byte aligned, anywhere in the buffer
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  ticks[blockIdx.x] = t1 - t0;
  sink[blockIdx.x] = acc;
}

extern "C" double chase_probe(const void *buf, uint64_t bytes, uint32_t steps, uint32_t waves, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!buf || bytes < 4096 || !steps || !waves) return -2.0;
  unsigned long long *ticks = nullptr;
  uint32_t *sink = nullptr;
  if (hipMalloc(&ticks, (size_t)waves * 8) != hipSuccess || hipMalloc(&sink, (size_t)waves * 4) != hipSuccess) return -1.0;
  hipLaunchKernelGGL(k_chase, dim3(waves), dim3(64), 0, s, (const uint32_t *)buf, bytes / 4, steps, ticks, sink);
  double ns = -1.0;
  if (hipStreamSynchronize(s) == hipSuccess) {
    unsigned long long *h = new unsigned long long[waves];
    if (hipMemcpy(h, ticks, (size_t)waves * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      double sum = 0;
      for (uint32_t i = 0; i < waves; i++) sum += (double)h[i];
      ns = sum / waves / steps * 10.0;  // 100 MHz
    }
    delete[] h;
  }
  (void)hipFree(ticks), (void)hipFree(sink);
  return ns;
}
