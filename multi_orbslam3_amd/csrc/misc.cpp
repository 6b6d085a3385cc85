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

#include <atomic>
#include <mutex>

namespace orbg {

// ---- wait policy per thread role (common.hpp)
namespace {
unsigned block_mask_from_env() {
  const char* e = getenv("ORBG_NO_POLL");
  if (!e) return 0u;
  if (!*e || !strcmp(e, "1") || !strcmp(e, "all")) return (1u << kRoleCount) - 1u;
  unsigned m = 0;
  if (strstr(e, "caller")) m |= 1u << kRoleCaller;
  if (strstr(e, "lba")) m |= 1u << kRoleLbaWorker;
  if (strstr(e, "ingest")) m |= 1u << kRoleIngest;
  return m ? m : (1u << kRoleCount) - 1u;                  // anything else that is set: the round-4 meaning (every wait blocks)
}
std::atomic<unsigned>& block_mask() { static std::atomic<unsigned> m{block_mask_from_env()}; return m; }
thread_local int t_role = kRoleCaller;
}  // namespace
bool poll_allowed() { return !((block_mask().load(std::memory_order_relaxed) >> t_role) & 1u); }
void set_thread_role(int role) { if (role >= 0 && role < kRoleCount) t_role = role; }

namespace {
struct DevicePool {
  bool made = false; hipStream_t L = nullptr, E[2] = {nullptr, nullptr}, M = nullptr; unsigned n_ex = 0, n_fr = 0;
  // orbg_quiesce's completion words: per DEVICE (pinned words belong to the device that was current when they were allocated), one
  // caller at a time (q_mu); they live as long as the pool
  std::mutex q_mu; StreamSignal q_sig[4];
};
std::mutex g_pool_mu;
DevicePool g_pool[64];
bool pool_enabled() { static const bool on = [] { const char* e = getenv("ORBG_STREAM_POOL"); return !(e && e[0] == '0'); }(); return on; }

hipError_t create_own(hipStream_t* st, const char* /*role*/) { return hipStreamCreateWithFlags(st, hipStreamNonBlocking); }

// the four streams of a device, created together (g_pool_mu held)
hipError_t make_pool(DevicePool& P) {
  if (P.made) return hipSuccess;
  hipStream_t s4[4] = {nullptr, nullptr, nullptr, nullptr};
  // (stream priorities were measured in rounds 3-4: they open further hardware queues and were slower in every combination)
  for (int i = 0; i < 4; i++) {
    const hipError_t e = hipStreamCreateWithFlags(&s4[i], hipStreamNonBlocking);
    if (e != hipSuccess) {
      for (int j = 0; j < i; j++) (void)hipStreamDestroy(s4[j]);
      return e;
    }
  }
  P.L = s4[0]; P.E[0] = s4[1]; P.E[1] = s4[2]; P.M = s4[3]; P.made = true;
  return hipSuccess;
}
}  // namespace

hipError_t create_stream(hipStream_t* st, const char* role) {
  if (!pool_enabled()) return create_own(st, role);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return create_own(st, role);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  DevicePool& P = g_pool[dev];
  if ((e = make_pool(P)) != hipSuccess) return e;
  if (!strcmp(role, "lba")) *st = P.L;
  else if (!strcmp(role, "ex")) *st = P.E[P.n_ex++ & 1];
  else if (!strcmp(role, "fr")) {
    // a frame's OWN stream (host-built frames: KeyFrames on the server, uploaded frames of the glue).  In a process with
    // extractors -- an agent -- E0 / E1 belong to the constructor chains and the searches of the tracking thread: other frames
    // go to M.  A process without extractors (the server's matcher threads) spreads its frames over M, E0, E1.
    const unsigned k = P.n_ex ? 0 : P.n_fr++ % 3;
    *st = k == 0 ? P.M : P.E[k - 1];
  }
  else *st = P.M;                                       // "map", "po", "bow", "db", "misc"
  return hipSuccess;
}

void release_stream(hipStream_t st) {
  if (!st) return;
  if (pool_enabled()) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (const DevicePool& P : g_pool)
      if (P.made && (st == P.L || st == P.E[0] || st == P.E[1] || st == P.M)) return;
  }
  (void)hipStreamDestroy(st);
}

bool is_library_stream(hipStream_t st) {
  if (!st || !pool_enabled()) return false;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (const DevicePool& P : g_pool)
    if (P.made && (st == P.L || st == P.E[0] || st == P.E[1] || st == P.M)) return true;
  return false;
}

}  // namespace orbg

extern "C" int orbg_set_wait_policy(int role, int spin) {
  if (role < 0 || role >= orbg::kRoleCount) return ORBG_BAD_ARG;
  if (spin) orbg::block_mask().fetch_and(~(1u << role), std::memory_order_relaxed);
  else orbg::block_mask().fetch_or(1u << role, std::memory_order_relaxed);
  return ORBG_OK;
}
extern "C" int orbg_get_wait_policy(int role) {
  if (role < 0 || role >= orbg::kRoleCount) return ORBG_BAD_ARG;
  return ((orbg::block_mask().load(std::memory_order_relaxed) >> role) & 1u) ? 0 : 1;
}

// Waits -- spinning on completion words, like every other wait of the library -- until everything enqueued so far on the library's
// pooled streams of `device` has completed.  For a caller that brackets a region with the runtime's own device synchronisation
// (bench.py: torch.cuda.synchronize()): after this call that synchronisation finds nothing outstanding and returns at once, instead
// of blocking in the runtime, whose wake-up path runs through helper threads the caller has not pinned (on a host whose other
// cores are busy a blocked wait was measured at 4-80 ms for work that had finished long before).
extern "C" int orbg_quiesce(int device) {
  using namespace orbg;
  if (!pool_enabled()) { ORBG_HIP(hipSetDevice(device)); ORBG_HIP(hipDeviceSynchronize()); return ORBG_OK; }
  if (device < 0 || device >= 64) return ORBG_BAD_ARG;
  hipStream_t st[4];
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    const DevicePool& P = g_pool[device];
    if (!P.made) return ORBG_OK;
    st[0] = P.L; st[1] = P.E[0]; st[2] = P.E[1]; st[3] = P.M;
  }
  ORBG_HIP(hipSetDevice(device));
  DevicePool& Pq = g_pool[device];
  std::lock_guard<std::mutex> qlk(Pq.q_mu);
  StreamSignal* sig = Pq.q_sig;
  int rc;
  for (int i = 0; i < 4; i++) if ((rc = sig[i].post(st[i]))) return rc;
  for (int i = 0; i < 4; i++) if ((rc = sig[i].wait(st[i]))) return rc;
  // let the runtime retire what has completed (non-blocking, ~3 us per stream): its own device synchronisation then has no command
  // left to look at (measured: 57 -> 9 us inside the synchronisation for 13 us more here)
  for (int i = 0; i < 4; i++) (void)hipStreamQuery(st[i]);
  return ORBG_OK;
}

namespace orbg {

// (the word is ordered behind the work it signals by the kernel boundary in front of this launch; a fence BEHIND the store orders
// nothing, and a full system fence invalidates the XCD's L2 under the kernels of the other streams)
__global__ void orbg_signal_kernel(volatile unsigned* flag, unsigned seq) {
  *flag = seq;
}

int StreamSignal::post(hipStream_t st) {
  if (!word.h) { int rc = init(); if (rc) return rc; }
  seq++;
  // (hipStreamWriteValue32 instead of a one-thread kernel: measured +2-3 us per search, round 4)
  hipLaunchKernelGGL(orbg_signal_kernel, dim3(1), dim3(1), 0, st, (volatile unsigned*)word.d, seq);
  ORBG_HIP(hipGetLastError());
  return ORBG_OK;
}

int StreamSignal::wait(hipStream_t st) {
  if (poll_allowed()) {
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
