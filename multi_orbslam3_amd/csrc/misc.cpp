// liborbgpu -- version / error strings / device probing.
#include "common.hpp"

extern "C" const char* orbg_version(void) { return "orbgpu 0.1.0 (gfx950)"; }

extern "C" const char* orbg_strerror(int code) {
  switch (code) {
    case ORBG_OK: return "ok";
    case ORBG_EMPTY: return "empty image";
    case ORBG_BAD_ARG: return "bad argument";
    case ORBG_CAP_EXCEEDED: return "capacity exceeded";
    case ORBG_HIP_ERROR: return "HIP runtime error (set ORBG_VERBOSE=1 for details)";
    case ORBG_NO_DEVICE: return "no usable HIP device (the library has no CPU fallback)";
    case ORBG_INTERNAL: return "internal error";
    default: return "unknown error";
  }
}

extern "C" int orbg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- StreamSignal (common.hpp)
#include <time.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace orbg {

__global__ void orbg_signal_kernel(volatile unsigned* flag, unsigned seq) {
  *flag = seq;
  __threadfence_system();
}

int StreamSignal::post(hipStream_t st) {
  if (!word.h) { int rc = init(); if (rc) return rc; }
  seq++;
  hipLaunchKernelGGL(orbg_signal_kernel, dim3(1), dim3(1), 0, st, (volatile unsigned*)word.d, seq);
  ORBG_HIP(hipGetLastError());
  return ORBG_OK;
}

int StreamSignal::wait(hipStream_t st) {
  if (!getenv("ORBG_NO_POLL")) {
    volatile unsigned* w = word.h;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0;; spins++) {
      if (*w == seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return ORBG_OK; }
#if defined(__x86_64__)
      _mm_pause();
#endif
      if ((spins & 0x3FFF) == 0x3FFF) {           // every ~16k spins: give up on polling after 50 ms (error or very long kernel)
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > 50.0) break;
      }
    }
  }
  ORBG_HIP(hipStreamSynchronize(st));
  return ORBG_OK;
}

}  // namespace orbg
