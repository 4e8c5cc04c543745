// Probe: what do hipMalloc / first kernel touch / hipFree of a scratch buffer of this size cost?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(float *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 1024 < n) p[i * 1024] = 1.0f;
}
int main() {
  for (size_t gb10 : {1, 5, 13, 26}) {
    const size_t bytes = gb10 * 100ull << 20;
    for (int rep = 0; rep < 2; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      float *p;
      (void)hipMalloc(&p, bytes);
      auto t1 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(touch, dim3((unsigned)(bytes / 4 / 1024 / 256 + 1)), dim3(256), 0, 0, p, bytes / 4);
      (void)hipDeviceSynchronize();
      auto t2 = std::chrono::steady_clock::now();
      (void)hipFree(p);
      auto t3 = std::chrono::steady_clock::now();
      auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      std::printf("%5zu MB: malloc %.2f ms, touch %.2f ms, free %.2f ms\n", bytes >> 20, ms(t0, t1), ms(t1, t2), ms(t2, t3));
    }
  }
}
