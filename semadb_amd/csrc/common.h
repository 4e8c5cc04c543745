// common.h -- shared host-side plumbing for the C ABI (error reporting, device guards).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/semadb_amd.h"

namespace sdb {

// thread-local last error (the C ABI never throws and never aborts).  A fixed buffer: reporting "out of host memory"
// must not itself allocate.
constexpr size_t kErrBytes = 1024;
char *last_error_buf() noexcept;
int fail(int code, const char *fmt, ...) noexcept;

// The C ABI's "never aborts the process" (CONTRIBUTING.md:150: no panics; the reference returns an `error`): no C++
// exception may leave an extern "C" function -- through cgo it would reach std::terminate and take the whole database
// down.  Every entry point is a function-try-block that ends in SDB_API_CATCH: std::bad_alloc (a 3 GB host staging
// vector at 12.5 M rows) -> SDB_ERR_DEVICE "out of host memory", std::system_error (no thread to be had) and anything
// else -> SDB_ERR_STATE with what(); locks, workspaces and device buffers are released by the destructors on the way.
// Entry points that change an index decide in their own handlers what the index is afterwards (as it was / unusable).
int on_exception(const char *fn) noexcept;  // inside a catch (...) handler only
#define SDB_API_CATCH(fn) \
  catch (...) { return sdb::on_exception(fn); }

#define SDB_HIP(expr)                                                                        \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      (void)hipGetLastError(); /* reported here: a later launch check must not find it again */ \
      return sdb::fail(SDB_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                                  \
    }                                                                                        \
  } while (0)

#define SDB_TRY(expr)            \
  do {                           \
    int _rc = (expr);            \
    if (_rc != SDB_OK) return _rc; \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// Row layout of the HBM vector slab (see DESIGN.md "Data layout"):
//   [ ng groups of 128 floats ][ 32 tail floats if dim % 32 != 0 ]
// group g, float L*4+k  <->  original element 32*(4g+k) + L   (zero when 4g+k >= dim/32)
// so that a 16-byte load at lane L of a 32-lane half-wave yields four consecutive links of the
// FMA chain of partial sum L of distance/asm/dot.s (block b = 4g+k, lane L).
struct RowLayout {
  uint32_t dim = 0, nblk = 0, ng = 0, tail = 0, ld = 0;
  explicit RowLayout(uint32_t d = 0) {
    dim = d;
    nblk = d / 32;
    ng = (nblk + 3) / 4;
    // The kernels that keep the query in registers exist for 1, 2, 3, 4, 6, 8, 12, 16 and 24 groups; a row
    // with a group count in between is zero-padded up to the next of those when that costs at most a third
    // more bytes (640 -> 768 floats, 1280 -> 1536): fma(0, 0, acc) = acc, so no sum changes, and the padded
    // row on the fast kernels beats the exact row on the generic one (4.3 -> 5.7 TB/s useful at d = 640).
    for (uint32_t t : {1u, 2u, 3u, 4u, 6u, 8u, 12u, 16u, 24u})
      if (t >= ng) {
        if (3 * t <= 4 * ng) ng = t;
        break;
      }
    tail = d % 32;
    ld = ng * 128 + (tail ? 32 : 0);
  }
};

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute applies to the function on the CURRENT device: a process that drives several GPUs
// (sdb_cluster_create_local) has to set a kernel's attribute once per device, not once per process.
inline bool first_use_on_this_device(std::atomic<uint64_t> &seen) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return !(seen.fetch_or(bit) & bit);
}

#ifdef __HIPCC__
// Read-only kernel inputs that a wave reads at wave-uniform addresses (the queries of the scans): through the constant
// address space such a load is a scalar load whatever else the kernel stores; as a global load the compiler makes it
// one only when it can prove that no store in the kernel may have written the location first
typedef const __attribute__((address_space(4))) float uniform_float;
typedef float uniform_f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) uniform_f4v uniform_float4;
__device__ __forceinline__ uniform_float *as_uniform(const float *p) { return (uniform_float *)p; }

// Position of element i of d[0..n) after DistSet.Sort (distset.go:223-238): an insertion sort that moves an element left
// while it is strictly `<` its left neighbour.  For ordinary and infinite distances that is a stable sort -- rank by
// counting under (distance, arrival index).  A NaN compares false both ways: it is never moved and nothing moves past
// it, so the NaNs stay where they are and cut the list into runs that are sorted each on its own.  (Counting without
// that rule gives every NaN rank 0 and leaves other positions unwritten -- whatever lay in the scratch then went into
// the graph as a slot number.)  `any_nan`: whether d[] holds a NaN at all, the same for every caller of one list.
__device__ __forceinline__ int dist_sort_rank(const float *d, int n, int i, bool any_nan) {
  const float di = d[i];
  int rank = 0;
  if (!any_nan) {
    for (int j = 0; j < n; j++) {
      const float dj = d[j];
      rank += (dj < di || (dj == di && j < i)) ? 1 : 0;
    }
    return rank;
  }
  if (di != di) return i;
  int lo = -1, hi = n;  // the NaNs next to i on either side
  for (int j = 0; j < n; j++) {
    const float dj = d[j];
    if (dj != dj) {
      if (j < i) lo = j;
      else if (j < hi) hi = j;
    }
  }
  rank = lo + 1;
  for (int j = lo + 1; j < hi; j++) {
    const float dj = d[j];
    rank += (dj < di || (dj == di && j < i)) ? 1 : 0;
  }
  return rank;
}
// does d[0..n) hold a NaN?  One wave, every lane gets the answer.
__device__ __forceinline__ bool wave_any_nan(const float *d, int n, int lane) {
  bool f = false;
  for (int j = lane; j < n; j += 64) f |= d[j] != d[j];
  return __ballot(f) != 0;
}

// A workgroup's waves take turns at a work counter in LDS: one ds_add_rtn by lane 0, the answer broadcast
__device__ __forceinline__ uint32_t next_query_group(void *lds_counter, int lane) {
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const uint32_t addr = (uint32_t)(uintptr_t)(lds_u32 *)lds_counter;
  uint32_t got = 0;
  if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(addr), "v"(1u));
  return __builtin_amdgcn_readfirstlane(got);
}
#endif

}  // namespace sdb
