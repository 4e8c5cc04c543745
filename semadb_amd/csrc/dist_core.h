// dist_core.h -- device-side distance arithmetic shared by every kernel.
//
// The reference's distance is asm.Dot / asm.SquaredEuclideanDistance (distance/asm/dot.s:7-55,
// distance/asm/euclidean.s:7-65): 32 partial sums (4 YMM registers x 8 lanes), each an FMA chain
// over the 32-float blocks in order, a sequential scalar chain for the n % 32 tail, and a fixed
// reduce tree.  Identical result ids and visit order need identical float compares, so the
// kernels reproduce exactly that arithmetic: a 32-lane half-wave owns one (query, candidate)
// pair, lane L owns partial sum L, and the reduce tree is replayed with wave shuffles.
// Compile with -ffp-contract=off: every fusion below is explicit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/semadb_amd.h"

namespace sdb {

constexpr uint32_t kNoSlot = 0xFFFFFFFFu;  // adjacency padding / "unknown id"
constexpr uint32_t kVisBit = 0x80000000u;  // DistSetElem.visited (distset.go:123) packed into the slot word

// one link of partial sum L: VFMADD231PS (dot.s:24-27) or VSUBPS + VFMADD231PS (euclidean.s:27-34)
template <bool L2>
__device__ __forceinline__ float chain1(float acc, float x, float y) {
  if constexpr (L2) {
    float d = x - y;  // separately rounded subtract
    return __builtin_fmaf(d, d, acc);
  } else {
    return __builtin_fmaf(x, y, acc);
  }
}

// four consecutive blocks (b = 4g..4g+3) of partial sum L: the slab layout puts them in one float4
template <bool L2>
__device__ __forceinline__ float chain4(float acc, const float4 &x, const float4 &y) {
  acc = chain1<L2>(acc, x.x, y.x);
  acc = chain1<L2>(acc, x.y, y.y);
  acc = chain1<L2>(acc, x.z, y.z);
  acc = chain1<L2>(acc, x.w, y.w);
  return acc;
}

// Sequential tail chain (dot.s:35-43 / euclidean.s:44-53).  Lane L of each half holds tail element L of x
// and of y; x is the bound point (the same in both halves), y the half's own row.  The chain runs in lane 0
// of each half -- the only lanes whose result asm_reduce uses: element i of x comes from a readlane, the
// row's elements walk down to lane 0 with one wave_shl DPP move per step (no LDS round trips).
template <bool L2>
__device__ __forceinline__ float tail_chain(float xt, float yt, uint32_t tail, int lane) {
  float t = 0.0f;
  for (uint32_t i = 0; i < tail; i++) {
    const float xi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xt), (int)i));
    t = chain1<L2>(t, xi, yt);
    yt = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(yt), 0x130, 0xf, 0xf, true));  // wave_shl:1
  }
  return t;
}

// The reduce tree (dot.s:45-53 / euclidean.s:55-63) inside each 32-lane half:
//   s[l] = ((acc[l] + acc[8+l]) + acc[16+l]) + acc[24+l]   l = 0..7   three VADDPS
//   r[l] = s[l] + s[l+4]                                   l = 0..3   VEXTRACTF128 + VADDPS
//   r[l] = r[l] + t[l]   (t = {tail, 0, 0, 0})                        VADDPS X0, X4, X0
//   result = (r[0] + r[1]) + (r[2] + r[3])                            two VHADDPS
// The result is valid in lane 0 of each half (wave lanes 0 and 32).
//
// Cross-lane moves are DPP row shifts (a modifier on the add itself, no LDS round trip) inside the 16-lane
// rows, and one v_permlane16_swap to bring lanes 16..31 of each half next to lanes 0..15.  Lanes that do
// not take part compute garbage that nobody reads.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {  // row_shl:n = 0x100 + n: lane i reads lane i + n of its row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float asm_reduce(float acc, float t, int lane) {
  const int L = lane & 31;
  // second result: even rows hold what the odd rows held -- lane l < 16 of each half gets acc[16 + l]
  const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
  const float hi = __uint_as_float(sw[1]);
  float s = ((acc + dpp_row<0x108>(acc)) + hi) + dpp_row<0x108>(hi);  // lanes 0..7
  float r = s + dpp_row<0x104>(s);                                    // lanes 0..3
  r = r + (L == 0 ? t : 0.0f);
  float u = r + dpp_row<0x101>(r);  // lane 0: r0 + r1, lane 2: r2 + r3
  return u + dpp_row<0x102>(u);
}

// distance.go:19-25: euclidean -> as is, cosine -> 1 - dot, dot -> -dot (one more fp32 rounding)
__device__ __forceinline__ float metric_finish(float raw, int metric) {
  if (metric == SDB_METRIC_COSINE) return 1.0f - raw;
  if (metric == SDB_METRIC_DOT) return -raw;
  return raw;
}

// Query element for group g, slot k of lane L, read from an ORIGINAL-layout vector.
__device__ __forceinline__ float q_elem(const float *__restrict__ q, uint32_t nblk, uint32_t g, uint32_t k,
                                        int L) {
  uint32_t b = 4 * g + k;
  return b < nblk ? q[32 * b + L] : 0.0f;
}

}  // namespace sdb
