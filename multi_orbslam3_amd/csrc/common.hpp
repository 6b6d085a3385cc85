// Shared host-side helpers for liborbgpu (HIP runtime only; no torch, no third-party deps).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/orbgpu.h"

#define ORBG_HIP(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      if (getenv("ORBG_VERBOSE"))                                                               \
        fprintf(stderr, "[orbgpu] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? ORBG_NO_DEVICE : ORBG_HIP_ERROR; \
    }                                                                                           \
  } while (0)

namespace orbg {

constexpr int kEdge = 19;        // EDGE_THRESHOLD  S/ORBextractor.cc:72
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE S/ORBextractor.cc:71
constexpr int kPatch = 31;       // PATCH_SIZE      S/ORBextractor.cc:70

// Growable device buffer (never shrinks); all allocations happen outside the timed/launch path once sizes settle.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return ORBG_OK;
    if (p) { ORBG_HIP(hipFree(p)); p = nullptr; cap = 0; }
    size_t want = n + n / 4 + 64;
    ORBG_HIP(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return ORBG_OK;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
};

// Pinned host buffer that the device can address directly (zero-copy hand-off of small, latency-bound records).
template <typename T>
struct PinnedBuf {
  T* h = nullptr;
  T* d = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return ORBG_OK;
    if (h) { ORBG_HIP(hipHostFree(h)); h = nullptr; d = nullptr; cap = 0; }
    size_t want = n + n / 4 + 64;
    ORBG_HIP(hipHostMalloc((void**)&h, want * sizeof(T), hipHostMallocMapped));
    ORBG_HIP(hipHostGetDevicePointer((void**)&d, h, 0));
    cap = want;
    return ORBG_OK;
  }
  void release() { if (h) { (void)hipHostFree(h); h = nullptr; d = nullptr; cap = 0; } }
};

inline int select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return ORBG_NO_DEVICE;
  if (device < 0 || device >= n) return ORBG_BAD_ARG;
  ORBG_HIP(hipSetDevice(device));
  return ORBG_OK;
}

}  // namespace orbg
