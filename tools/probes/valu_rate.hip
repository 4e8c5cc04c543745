// Probe (not product code): issue cost of the vector instructions the euclidean scan is made of, per SIMD, at 1 / 2 /
// 4 waves per SIMD: v_pk_fma_f32, v_pk_add_f32 with a scalar-pair or a vector operand, and their one-float forms.
// No memory traffic; cycles from s_memtime around the loop of the longest-running wave.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

enum { PK_FMA = 0, PK_SUBS_FMA, PK_SUBV_FMA, F_SUBS_FMA, F_FMA, PK_SUBS_FMA_2ROW, PK_MUL_S, PK_ADD_S, PK_ADD_V, PK_FMA_DD, PK_FMASUB_FMA, NVAR };
static const char *names[NVAR] = {"v_pk_fma_f32 v,v,v",
                                  "v_pk_add_f32 s,-v ; v_pk_fma_f32 d,d,acc",
                                  "v_pk_add_f32 v,-v ; v_pk_fma_f32 d,d,acc",
                                  "v_sub_f32 s,v ; v_fma_f32 d,d,acc (x2 per packed pair)",
                                  "v_fma_f32 v,v,v",
                                  "v_pk_add_f32 s,-v x2 rows ; v_pk_fma_f32 x2",
                                  "v_pk_fma_f32 v,s,v",
                                  "v_pk_add_f32 s,-v alone",
                                  "v_pk_add_f32 v,-v alone",
                                  "v_pk_fma_f32 d,d,acc alone",
                                  "v_pk_fma_f32 y,m1,s (the difference) ; v_pk_fma_f32 d,d,acc"};

template <int V>
__global__ void k_rate(uint64_t *cycles, float *sink, int iters, f2 sx, float seed) {
  f2 acc[16], y[8];
  for (int i = 0; i < 16; i++) acc[i] = f2{seed + i, seed - i};
  for (int i = 0; i < 8; i++) y[i] = f2{seed * (threadIdx.x + i), seed + threadIdx.x};
  f2 s0 = sx, s1 = sx + 1.0f;
  s0.x = __builtin_amdgcn_readfirstlane(s0.x);
  s0.y = __builtin_amdgcn_readfirstlane(s0.y);
  s1.x = __builtin_amdgcn_readfirstlane(s1.x);
  s1.y = __builtin_amdgcn_readfirstlane(s1.y);
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if constexpr (V == PK_FMA) {
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(y[i & 7]), "v"(y[(i + 1) & 7]));
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(y[(i + 2) & 7]), "v"(y[(i + 3) & 7]));
    } else if constexpr (V == PK_SUBS_FMA) {
      f2 d[16];
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d[i]) : "s"(i & 1 ? s0 : s1), "v"(y[i & 7]));
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(d[i]));
    } else if constexpr (V == PK_SUBV_FMA) {
      f2 d[16];
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d[i]) : "v"(y[(i + 3) & 7]), "v"(y[i & 7]));
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(d[i]));
    } else if constexpr (V == F_SUBS_FMA) {
      float d[16];
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d[i]) : "s"(i & 1 ? s0.x : s1.y), "v"(y[i & 7].x));
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[i].x) : "v"(d[i]));
    } else if constexpr (V == F_FMA) {
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(y[i & 7].x), "v"(y[(i + 1) & 7].y));
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(y[i & 7].y), "v"(y[(i + 1) & 7].x));
    } else if constexpr (V == PK_SUBS_FMA_2ROW) {
      f2 d[16];
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d[i]) : "s"((i >> 1) & 1 ? s0 : s1), "v"(y[i & 7]));
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(d[i]));
    } else if constexpr (V == PK_ADD_S) {
#pragma unroll
      for (int i = 0; i < 32; i++)
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(acc[i & 15]) : "s"(i & 1 ? s0 : s1), "v"(y[i & 7]));
    } else if constexpr (V == PK_ADD_V) {
#pragma unroll
      for (int i = 0; i < 32; i++)
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(acc[i & 15]) : "v"(y[(i + 3) & 7]), "v"(y[i & 7]));
    } else if constexpr (V == PK_FMA_DD) {
#pragma unroll
      for (int i = 0; i < 32; i++) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i & 15]) : "v"(y[i & 7]));
    } else if constexpr (V == PK_FMASUB_FMA) {
      f2 d[16];
      const f2 m1 = f2{-1.0f, -1.0f};
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d[i]) : "v"(y[i & 7]), "v"(m1), "s"(i & 1 ? s0 : s1));
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(d[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(y[i & 7]), "s"(i & 1 ? s0 : s1));
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(y[(i + 2) & 7]), "s"(i & 1 ? s1 : s0));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  f2 s = acc[0];
  for (int i = 1; i < 16; i++) s += acc[i];
  if (s.x == 12345.678f) sink[0] = s.y;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int V>
static void run(uint64_t *dc, float *ds, int wps, int blocks = 256, int iters = 4000) {
  const int threads = 256 * wps;
  std::vector<uint64_t> h((size_t)blocks * threads / 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_rate<V><<<blocks, threads>>>(dc, ds, 100, f2{1.5f, 2.5f}, 0.001f);
  hipEventRecord(e0);
  k_rate<V><<<blocks, threads>>>(dc, ds, iters, f2{1.5f, 2.5f}, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (auto c : h) mx = c > mx ? c : mx;
  const double instr = 32.0 * iters * wps;  // vector instructions per SIMD
  printf("  %-58s %d waves/SIMD, %3d CUs: %.2f ticks/instr/SIMD (s_memtime), %.3f ms = %.2f ns/instr/SIMD, %.2f G ticks/s\n", names[V], wps, blocks,
         (double)mx / instr, ms, ms * 1e6 / instr, (double)mx / ms / 1e6);
}

int main() {
  uint64_t *dc;
  float *ds;
  hipMalloc(&dc, 8 * 256 * 16);
  hipMalloc(&ds, 64);
  for (int wps : {2, 4}) {
    run<PK_FMA>(dc, ds, wps);
    run<PK_SUBS_FMA>(dc, ds, wps);
    run<PK_SUBV_FMA>(dc, ds, wps);
    run<F_SUBS_FMA>(dc, ds, wps);
    run<F_FMA>(dc, ds, wps);
    run<PK_SUBS_FMA_2ROW>(dc, ds, wps);
    run<PK_MUL_S>(dc, ds, wps);
    run<PK_ADD_S>(dc, ds, wps);
    run<PK_ADD_V>(dc, ds, wps);
    run<PK_FMA_DD>(dc, ds, wps);
    run<PK_FMASUB_FMA>(dc, ds, wps);
  }
  // the same stream on a part of the chip: is the rate the clock's (power) or the pipeline's?
  for (int blocks : {16, 64, 128, 256}) {
    run<PK_SUBS_FMA>(dc, ds, 4, blocks);
    run<PK_FMASUB_FMA>(dc, ds, 4, blocks);
    run<F_SUBS_FMA>(dc, ds, 4, blocks);
  }
  // long launches: what clock does the chip sustain under this stream?
  for (int iters : {4000, 40000, 160000, 160000, 160000}) run<PK_SUBS_FMA>(dc, ds, 4, 256, iters);
  for (int iters : {160000, 160000}) run<F_SUBS_FMA>(dc, ds, 4, 256, iters);
  for (int iters : {160000, 160000}) run<PK_FMA>(dc, ds, 4, 256, iters);
  for (int iters : {160000, 160000}) run<PK_FMA>(dc, ds, 1, 256, iters);
  return 0;
}
