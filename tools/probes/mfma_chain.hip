// Probe (not product code): does a chain of v_mfma_f32_16x16x1_4b_f32 on ONE accumulator (each instruction's C operand is
// the previous one's result) run at the matrix pipe's rate, or does it need several independent accumulators in rotation?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k_chain(float *out, int iters, float x, float y) {
  f16 a[NACC];
  for (int i = 0; i < NACC; i++)
    for (int r = 0; r < 16; r++) a[i][r] = 0.0f;
  const float xa = x + threadIdx.x, ya = y + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 16 / NACC; j++)
#pragma unroll
      for (int i = 0; i < NACC; i++) a[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(xa, ya, a[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; i++)
    for (int r = 0; r < 16; r++) s += a[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
static void run(float *d, int waves_per_simd) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k_chain<NACC><<<256, 256 * waves_per_simd>>>(d, 100, 1.0f, 2.0f);
  (void)hipEventRecord(e0);
  k_chain<NACC><<<256, 256 * waves_per_simd>>>(d, iters, 1.0f, 2.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = 16.0 * iters * waves_per_simd;  // matrix instructions per SIMD
  printf("  %d accumulator(s) in rotation, %d wave(s) per SIMD: %.2f ns per instruction and SIMD (%.1f TFLOP/s)\n", NACC,
         waves_per_simd, ms * 1e6 / n, n * 1024 * 2048.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float *d;
  (void)hipMalloc(&d, 64);
  for (int w : {1, 2}) {
    run<1>(d, w);
    run<2>(d, w);
    run<4>(d, w);
    run<8>(d, w);
  }
  return 0;
}
