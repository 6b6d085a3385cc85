// hipStreamLegacy as an explicit handle of the null stream (ROCm 7.2.0, MI355X): launches, copies and hipEventRecord work,
// hipStreamWaitEvent on an event recorded on it segfaults inside the runtime -- the library passes a null pointer instead.
//   hipcc --offload-arch=gfx950 -O2 -o stream_legacy_probe stream_legacy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* p) { p[0] = 7; }
#define T(x) do { printf("%s ... ", #x); fflush(stdout); hipError_t e = (x); printf("%s\n", hipGetErrorString(e)); } while (0)
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t s = hipStreamLegacy, own; int *d, *h; hipEvent_t ev;
  T(hipMalloc(&d, 64)); T(hipHostMalloc(&h, 64, hipHostMallocMapped)); T(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  T(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
  T(hipMemsetAsync(d, 0, 64, s));
  T(hipMemcpyAsync(d, h, 64, hipMemcpyHostToDevice, s));
  printf("launch ... "); hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s, d); printf("%s\n", hipGetErrorString(hipGetLastError()));
  T(hipEventRecord(ev, s));
  T(hipStreamWaitEvent(own, ev, 0));
  T(hipEventRecord(ev, own));
  T(hipStreamWaitEvent(s, ev, 0));
  T(hipStreamQuery(s));
  T(hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s));
  T(hipStreamSynchronize(s));
  printf("h[0] = %d\n", h[0]);
  T(hipMemcpy2DAsync(d, 16, h, 16, 16, 2, hipMemcpyHostToDevice, s));
  hipGraph_t g; T(hipStreamBeginCapture(own, hipStreamCaptureModeThreadLocal)); hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, own, d); T(hipStreamEndCapture(own, &g));
  void* args[] = {&d};
  T(hipLaunchKernel((const void*)k, dim3(1), dim3(1), args, 0, s));
  T(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
