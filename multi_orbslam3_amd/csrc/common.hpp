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

// ORBG_POISON=1 fills every new buffer with 0xA5 so that a kernel consuming memory nobody wrote shows up in the parity
// tests (fresh HIP allocations are often zero, recycled ones are not).  Test aid; allocations are outside the timed path.
inline bool poison_allocations() { static const bool on = getenv("ORBG_POISON") != nullptr; return on; }

// Wait policy per host-thread ROLE (include/orbgpu.h, "Host threads"): a wait either SPINS on a completion word in pinned memory or
// goes through the runtime's blocking waits / a condition variable.  Roles: the caller's threads (Tracking), the local-BA worker of a
// handle, the process's image-ingest thread.  ORBG_NO_POLL in the environment gives the start-up policy ("1" / "all": every role
// blocks; a comma list of caller / lba / ingest: those roles block); orbg_set_wait_policy() changes it at run time.  Implemented in
// misc.cpp (one state for all translation units); poll_allowed() is what every wait site asks -- an atomic load + a thread-local.
enum { kRoleCaller = 0, kRoleLbaWorker = 1, kRoleIngest = 2, kRoleCount = 3 };
bool poll_allowed();                 // may the CALLING thread spin?
void set_thread_role(int role);      // called once by the library's own threads when they start

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
    // (the fill runs on the null stream and hipMemset may return before it has finished; the library's streams are non-blocking
    // since round 4 and do not join it: without the synchronisation the fill raced with the first kernels that write the buffer)
    if (poison_allocations()) { ORBG_HIP(hipMemset(p, 0xA5, want * sizeof(T))); ORBG_HIP(hipDeviceSynchronize()); }
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
    if (poison_allocations()) memset(h, 0xA5, want * sizeof(T));
    cap = want;
    return ORBG_OK;
  }
  void release() { if (h) { (void)hipHostFree(h); h = nullptr; d = nullptr; cap = 0; } }
};

// Completion signal without the runtime's wake-up path: a one-thread kernel at the tail of a stream writes a sequence number
// into mapped pinned memory (everything launched before it on that stream has completed and is visible by then), the host
// spins on that word.  Falls back to hipStreamSynchronize if the word does not arrive.
struct StreamSignal {
  PinnedBuf<unsigned> word;
  unsigned seq = 0;
  int init() { int rc = word.reserve(16); if (rc) return rc; word.h[0] = 0; return ORBG_OK; }
  void release() { word.release(); }
  int post(hipStream_t st);      // enqueue the signal kernel (misc.cpp)
  // for kernels that post the signal themselves (DoneSig below): next sequence number + the word's device address
  int arm(unsigned* seq_out, volatile unsigned** flag_out) {
    if (!word.h) { int rc = init(); if (rc) return rc; }
    *seq_out = ++seq; *flag_out = (volatile unsigned*)word.d;
    return ORBG_OK;
  }
  int wait(hipStream_t st);      // spin until the posted signal arrives (misc.cpp)
  int sync(hipStream_t st) { int rc = post(st); return rc ? rc : wait(st); }
};

// Streams of the library's handles.  HIP multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 unless the
// environment says otherwise): the first four streams get a queue each, every further one joins the queue with the fewest users,
// ties broken by the queues' ADDRESSES -- i.e. differently from run to run.  Streams that share a hardware queue are serialised: with
// one stream per handle (two extractors, two frames, the local map, the local BA, PoseOptimization = 7) the local BA landed on the
// queue of an extractor in about one process out of three: 0.79 instead of 0.59 ms per solve and 5600 instead of 8400 frames/s for
// the whole run, same cores, no host noise.  The library therefore owns FOUR non-blocking streams per device,
// created together by the first handle so that each gets a hardware queue of its own:
//   "lba"                              -> L        (local BA handles; nothing else ever)
//   "ex"                               -> E0 / E1  (extractor handles, alternating: Frame(t+1) is built next to the searches on frame t)
//   "fr"                               -> M in a process that has extractors (a frame built by the constructor is on its extractor's
//                                         stream until that constructor has been collected, then back on M: matcher.hip,
//                                         orbm_internal_set_n); M, E0, E1 in turn in a process without (the server's KeyFrame matchers)
//   "map", "po", "bow", "db", "misc"   -> M        (map uploads, PoseOptimization, vocabulary / database work, stand-alone utilities)
// None of them is the legacy null stream: nothing the library launches joins (or is joined by) the blocking streams of the
// application it is embedded in.  An application that itself keeps streams busy should run with GPU_MAX_HW_QUEUES >= 4 + its own
// (include/orbgpu.h, "Streams"), or hand the library its streams: every handle type has a *_set_stream entry point.
// Handles that share a stream stay correct -- every completion signal is enqueued behind the handle's own work on an in-order
// stream -- they merely do not overlap.  ORBG_STREAM_POOL=0 gives every handle a stream of its own again.  Implemented in misc.cpp
// (one registry for all translation units).
hipError_t create_stream(hipStream_t* st, const char* role);
void release_stream(hipStream_t st);       // destroys a stream of its own; pool streams live as long as the process
bool is_library_stream(hipStream_t st);    // one of the pool's streams (shared between handles: never capture on it, never destroy it)

// *_set_stream (include/orbgpu.h, "Streams"): the handle's work goes to the caller's stream from now on; NULL = back to the library's
// stream of the role.  The handle must be idle; its old stream is drained first.  A caller's stream is never destroyed by the library.
inline int swap_stream(hipStream_t* slot, bool* external, void* user, const char* role) {
  if (*slot) ORBG_HIP(hipStreamSynchronize(*slot));
  if (!*external) release_stream(*slot);
  *slot = nullptr;
  if (user) { *slot = (hipStream_t)user; *external = true; }
  else { *external = false; ORBG_HIP(create_stream(slot, role)); }
  return ORBG_OK;
}

// the M stream of the current device for a stand-alone utility call (a stream of its own, destroyed on scope exit, when the pool is off)
struct MiscStream {
  hipStream_t s = nullptr;
  int open() { ORBG_HIP(create_stream(&s, "misc")); return ORBG_OK; }
  ~MiscStream() { if (s) { (void)hipStreamSynchronize(s); release_stream(s); } }
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
