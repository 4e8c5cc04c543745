// Probe (not product code): is v_mfma_f32_{4x4x1,16x16x1} bit-identical to one fused v_fma_f32 per output element --
// same rounding, denormals kept?  If so a k=1 MFMA advances one step of a reference FMA chain for a whole tile of
// (row, query) pairs.  Prints mismatch counts against fmaf() and against the unfused (a*b)+c.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

__global__ void k4(const float *a, const float *b, const float *c, float *d) {
  const int l = threadIdx.x, t = blockIdx.x;
  f4 acc;
  for (int i = 0; i < 4; i++) acc[i] = c[(t * 64 + l) * 4 + i];
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t * 64 + l], b[t * 64 + l], acc, 0, 0, 0);
  for (int i = 0; i < 4; i++) d[(t * 64 + l) * 4 + i] = acc[i];
}
__global__ void k16(const float *a, const float *b, const float *c, float *d) {
  const int l = threadIdx.x, t = blockIdx.x;
  f16 acc;
  for (int i = 0; i < 16; i++) acc[i] = c[(t * 64 + l) * 16 + i];
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[t * 64 + l], b[t * 64 + l], acc, 0, 0, 0);
  for (int i = 0; i < 16; i++) d[(t * 64 + l) * 16 + i] = acc[i];
}

// throughput: 8 independent accumulators per wave, 4 waves per workgroup, no memory traffic
__global__ __launch_bounds__(256) void k_rate16(float *out, int iters, float x, float y) {
  f16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  const float xa = x + threadIdx.x, ya = y + threadIdx.x;
  for (int it = 0; it < iters; it++) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x1f32(xa, ya, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x1f32(xa, ya, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x1f32(xa, ya, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x1f32(xa, ya, a3, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x1f32(ya, xa, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x1f32(ya, xa, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x1f32(ya, xa, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x1f32(ya, xa, a3, 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_rate4(float *out, int iters, float x, float y) {
  f4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0}, a4 = {0}, a5 = {0}, a6 = {0}, a7 = {0};
  const float xa = x + threadIdx.x, ya = y + threadIdx.x;
  for (int it = 0; it < iters; it++) {
    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa, ya, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa, ya, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa, ya, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa, ya, a3, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(ya, xa, a4, 0, 0, 0);
    a5 = __builtin_amdgcn_mfma_f32_4x4x1f32(ya, xa, a5, 0, 0, 0);
    a6 = __builtin_amdgcn_mfma_f32_4x4x1f32(ya, xa, a6, 0, 0, 0);
    a7 = __builtin_amdgcn_mfma_f32_4x4x1f32(ya, xa, a7, 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; i++) s += a0[i] + a1[i] + a2[i] + a3[i] + a4[i] + a5[i] + a6[i] + a7[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_ratepk(float *out, int iters, float x, float y) {
  f2 acc[32];
  for (int k = 0; k < 32; k++) acc[k] = f2{(float)threadIdx.x, (float)k};
  f2 q = {x, y}, r = {y, x};
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int k = 0; k < 32; k++) acc[k] = __builtin_elementwise_fma(q, r, acc[k]);
  float s = 0;
  for (int k = 0; k < 32; k++) s += acc[k][0] + acc[k][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static uint32_t bitsof(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
  const int T = 4096;
  std::mt19937 rng(1);
  auto gen = [&](int mode) {
    uint32_t u = rng();
    float f;
    switch (mode % 5) {
      case 0: f = std::ldexp((float)(int32_t)rng() / 2147483648.f, (int)(rng() % 40) - 20); break;  // ordinary
      case 1: u &= 0x807FFFFFu; memcpy(&f, &u, 4); break;                                          // denormal
      case 2: f = std::ldexp((float)(int32_t)rng() / 2147483648.f, -(int)(rng() % 30) - 60); break; // products underflow
      case 3: f = (float)((int)(rng() % 7) - 3); break;                                            // small ints, zeros
      default: memcpy(&f, &u, 4); if (rng() % 16 == 0) f = (rng() & 1) ? INFINITY : -INFINITY; break;  // any bits: NaNs and infinities too
    }
    return f;
  };
  for (int shape = 0; shape < 2; shape++) {
    const int R = shape == 0 ? 4 : 16;
    std::vector<float> a(T * 64), b(T * 64), c(T * 64 * R), d(T * 64 * R);
    for (int t = 0; t < T; t++)
      for (int l = 0; l < 64; l++) {
        a[t * 64 + l] = gen(t), b[t * 64 + l] = gen(t / 5);
        for (int i = 0; i < R; i++) c[(t * 64 + l) * R + i] = gen(t / 25);
      }
    float *da, *db, *dc, *dd;
    hipMalloc(&da, a.size() * 4), hipMalloc(&db, b.size() * 4), hipMalloc(&dc, c.size() * 4), hipMalloc(&dd, d.size() * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    if (shape == 0) hipLaunchKernelGGL(k4, dim3(T), dim3(64), 0, 0, da, db, dc, dd);
    else hipLaunchKernelGGL(k16, dim3(T), dim3(64), 0, 0, da, db, dc, dd);
    hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
    long bad_fused = 0, bad_unfused = 0, total = 0, differ = 0, bad_den = 0, den_total = 0, nan_class = 0;
    for (int t = 0; t < T; t++)
      for (int l = 0; l < 64; l++)
        for (int r = 0; r < R; r++) {
          int ia, ib;  // which lane's a / b feeds output (lane l, reg r)
          if (shape == 0) ia = 4 * (l / 4) + r, ib = l;                     // 16 blocks of 4x4: D[blk][i=r][j=l%4]
          else ia = 16 * (r / 4) + 4 * (l / 16) + r % 4, ib = 16 * (r / 4) + l % 16;  // 4 blocks of 16x16
          const float x = a[t * 64 + ia], y = b[t * 64 + ib], z = c[(t * 64 + l) * R + r];
          const float want = std::fmaf(x, y, z);
          volatile float p = x * y;
          const float unf = p + z;
          const float got = d[(t * 64 + l) * R + r];
          total++;
          if (bitsof(want) != bitsof(unf)) differ++;
          if (std::isnan(got) != std::isnan(want)) nan_class++;
          if (bitsof(got) != bitsof(want) && !(std::isnan(got) && std::isnan(want))) {
            bad_fused++;
            if (bad_fused <= 5) std::printf("  shape %d: a=%a b=%a c=%a got=%a fmaf=%a\n", R, x, y, z, got, want);
          }
          if (bitsof(got) != bitsof(unf) && !(std::isnan(got) && std::isnan(unf))) bad_unfused++;
          const bool den = (std::fpclassify(x) == FP_SUBNORMAL) || (std::fpclassify(y) == FP_SUBNORMAL) ||
                           (std::fpclassify(z) == FP_SUBNORMAL) || (std::fpclassify(want) == FP_SUBNORMAL);
          if (den) { den_total++; if (bitsof(got) != bitsof(want)) bad_den++; }
        }
    std::printf("  (a NaN where fmaf has none or the other way round: %ld)\n", nan_class);
    std::printf("mfma_f32_%dx%dx1: %ld outputs, %ld where fused != unfused; mismatches vs fmaf %ld (of them with denormals %ld / %ld), vs unfused %ld\n",
                R, R, total, differ, bad_fused, bad_den, den_total, bad_unfused);
  }
  {
    float *o;
    hipMalloc(&o, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int which = 0; which < 3; which++)
      for (int rep = 0; rep < 3; rep++) {
        const int iters = 20000, wgs = 2048;
        hipEventRecord(e0, 0);
        if (which == 0) hipLaunchKernelGGL(k_rate16, dim3(wgs), dim3(256), 0, 0, o, iters, 1.0f, 1e-9f);
        else if (which == 2) hipLaunchKernelGGL(k_rate4, dim3(wgs), dim3(256), 0, 0, o, iters, 1.0f, 1e-9f);
        else hipLaunchKernelGGL(k_ratepk, dim3(wgs), dim3(256), 0, 0, o, iters * 4, 1.0f, 1e-9f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double fma = which == 0 ? (double)wgs * 4 * iters * 8 * 1024.0 : which == 2 ? (double)wgs * 4 * iters * 8 * 256.0 : (double)wgs * 4 * iters * 4 * 32 * 128.0;
        std::printf("%s: %.2f ms, %.1f TFLOP/s\n", which == 0 ? "mfma_f32_16x16x1 x4 acc" : which == 2 ? "mfma_f32_4x4x1 x8 acc" : "v_pk_fma_f32 x32 chains", ms,
                    2 * fma / ms * 1e-9);
      }
  }
  return 0;
}
