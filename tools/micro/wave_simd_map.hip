// Which SIMD of the compute unit does wavefront w of a workgroup run on?  (HW_REG_HW_ID: WAVE_ID[3:0] SIMD_ID[5:4] CU_ID[11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(2); } } while (0)
__global__ void k_map(unsigned* out) {
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = id;
}
int main() {
  unsigned* d; CK(hipMalloc(&d, 64));
  for (int nt : {256, 512, 1024}) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_map, dim3(1), dim3(nt), 0, 0, d);
      CK(hipDeviceSynchronize());
      unsigned h[16]; CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
      printf("%4d threads: wavefront -> simd:", nt);
      for (int w = 0; w < nt / 64; w++) printf(" %u", (h[w] >> 4) & 3);
      printf("   (slot:");
      for (int w = 0; w < nt / 64; w++) printf(" %u", h[w] & 15);
      printf(")  cu %u\n", (h[0] >> 8) & 15);
    }
  }
  return 0;
}
