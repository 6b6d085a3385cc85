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
