// Hand-over latency between two workgroups through memory: what a multi-workgroup LDL^T would pay per step.
// 64 one-wavefront workgroups are launched; each reports the XCD it runs on (HW_REG_XCC_ID); two of them -- on the SAME XCD or on
// DIFFERENT ones -- play ping-pong with a 2 KB payload + a flag word:
//   mode 0  agent-scope release store / acquire load of the flag (L2 write-back + invalidate: correct across XCDs)
//   mode 1  payload stores, s_waitcnt vmcnt(0), relaxed agent-scope flag store; flag and payload read with agent-scope relaxed atomic
//           loads (sc1: miss in the CU's vector L1, served by the XCD's L2) -- valid only when both workgroups share an L2
// Prints cycles (s_memtime, 100 MHz) and nanoseconds per one-way hand-over.   Build: tools/micro/build.sh xcd_handover
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

__global__ __launch_bounds__(64) void k_where(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// wgA / wgB: the two participating workgroups; everybody else leaves
template <int MODE>
__global__ __launch_bounds__(64) void k_pingpong(int wgA, int wgB, int iters, double* payA, double* payB, unsigned* flagA, unsigned* flagB,
                                                long long* cycles, int* bad) {
  const int me = (int)blockIdx.x == wgA ? 0 : (int)blockIdx.x == wgB ? 1 : -1;
  if (me < 0) return;
  const int lane = threadIdx.x;
  double* mine = me == 0 ? payA : payB;
  double* theirs = me == 0 ? payB : payA;
  unsigned* my_flag = me == 0 ? flagA : flagB;
  unsigned* their_flag = me == 0 ? flagB : flagA;
  int errors = 0;
  const long long t0 = wall_clock64();
  for (int it = 1; it <= iters; it++) {
    if (me == 1) {
      // wait for A's payload of this round
      if (MODE == 0) { while (__hip_atomic_load(their_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)it) {} }
      else { while (__hip_atomic_load(their_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)it) {} }
      for (int q = 0; q < 4; q++) {
        const double v = MODE == 0 ? theirs[4 * lane + q] : __hip_atomic_load(&theirs[4 * lane + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        errors += v != (double)it;
      }
    }
    for (int q = 0; q < 4; q++) mine[4 * lane + q] = (double)it;
    if (MODE == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // every lane's stores (the flag store below is lane 0's)
      if (lane == 0) __hip_atomic_store(my_flag, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(my_flag, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (me == 0) {
      if (MODE == 0) { while (__hip_atomic_load(their_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)it) {} }
      else { while (__hip_atomic_load(their_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)it) {} }
      for (int q = 0; q < 4; q++) {
        const double v = MODE == 0 ? theirs[4 * lane + q] : __hip_atomic_load(&theirs[4 * lane + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        errors += v != (double)it;
      }
    }
  }
  const long long t1 = wall_clock64();
  if (lane == 0 && me == 0) cycles[0] = t1 - t0;
  if (errors) atomicAdd(bad, errors);
}

int main() {
  const int G = 64, iters = 2000;
  unsigned* d_where; CK(hipMalloc(&d_where, G * sizeof(unsigned)));
  hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, 0, d_where);
  std::vector<unsigned> where(G);
  CK(hipMemcpy(where.data(), d_where, G * sizeof(unsigned), hipMemcpyDeviceToHost));
  printf("XCC_ID of workgroups 0..%d:", G - 1);
  for (int i = 0; i < G; i++) printf(" %u", where[i]);
  printf("\n");
  int same = -1, other = -1;
  for (int i = 1; i < G && (same < 0 || other < 0); i++) {
    if (where[i] == where[0] && same < 0) same = i;
    if (where[i] != where[0] && other < 0) other = i;
  }
  double *pa, *pb; unsigned *fa, *fb; long long* cyc; int* bad;
  CK(hipMalloc(&pa, 2048)); CK(hipMalloc(&pb, 2048)); CK(hipMalloc(&fa, 256)); CK(hipMalloc(&fb, 256)); CK(hipMalloc(&cyc, 64)); CK(hipMalloc(&bad, 4));
  int freq_khz = 0;
  CK(hipDeviceGetAttribute(&freq_khz, hipDeviceAttributeWallClockRate, 0));
  struct Case { const char* name; int mode; int partner; } cases[] = {
      {"agent-scope release / acquire, SAME XCD", 0, same}, {"agent-scope release / acquire, OTHER XCD", 0, other},
      {"vmcnt(0) + relaxed sc1 accesses,  SAME XCD", 1, same}, {"vmcnt(0) + relaxed sc1 accesses,  OTHER XCD (not coherent: errors expected)", 1, other}};
  for (const Case& c : cases) {
    if (c.partner < 0) { printf("%s: no such partner\n", c.name); continue; }
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(fa, 0, 256)); CK(hipMemset(fb, 0, 256)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(pa, 0, 2048)); CK(hipMemset(pb, 0, 2048));
      CK(hipDeviceSynchronize());
      if (c.mode == 0) hipLaunchKernelGGL(k_pingpong<0>, dim3(G), dim3(64), 0, 0, 0, c.partner, iters, pa, pb, fa, fb, cyc, bad);
      else hipLaunchKernelGGL(k_pingpong<1>, dim3(G), dim3(64), 0, 0, 0, c.partner, iters, pa, pb, fa, fb, cyc, bad);
      CK(hipDeviceSynchronize());
    }
    long long cy = 0; int nb = 0;
    CK(hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
    const double ns = 1e6 * (double)cy / (double)freq_khz / (2.0 * iters);
    printf("%-80s workgroups 0 <-> %2d: %8.1f ns per one-way hand-over of 2 KB (%d payload mismatches)\n", c.name, c.partner, ns, nb);
  }
  return 0;
}
