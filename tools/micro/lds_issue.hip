// Micro-benchmark: LDS instruction issue cost (cycles per instruction per wavefront) for the access shapes the block
// LDL^T uses, with 1 and with 10 wavefronts of one workgroup hammering the LDS of one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 64
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k_lds(long long* cyc, double* out, int stride_dw) {
  extern __shared__ double lds[];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int i = t; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  // per-lane base (in doubles): MODE-specific
  int base;
  if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4 || MODE == 5) base = wv * 512 + lane * (stride_dw / 2);   // distinct addresses
  else base = wv * 512;                                                          // broadcast
  __syncthreads();
  long long t0 = clock64();
#pragma unroll 8
  for (int i = 0; i < N_IT; i++) {
    const int o = base + ((i & 7) << 1);
    if (MODE == 0 || MODE == 1) {          // ds_read_b64 x4
      double a, b, c, d;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:2048\n ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:6144\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(o * 8) : "memory");
      acc0 += a; acc1 += b; acc2 += c; acc3 += d;
    } else if (MODE == 2) {   // ds_read2_b64 x4 (8 doubles)
      d2 a, b, c, d;
      asm volatile("ds_read2_b64 %0, %4 offset1:1\n ds_read2_b64 %1, %4 offset0:2 offset1:3\n ds_read2_b64 %2, %4 offset0:4 offset1:5\n ds_read2_b64 %3, %4 offset0:6 offset1:7\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(o * 8) : "memory");
      acc0 += a.x + a.y; acc1 += b.x + b.y; acc2 += c.x + c.y; acc3 += d.x + d.y;
    } else if (MODE == 3) {   // ds_read_b128 x4 (8 doubles)
      d2 a, b, c, d;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"((o & ~1) * 8) : "memory");
      acc0 += a.x + a.y; acc1 += b.x + b.y; acc2 += c.x + c.y; acc3 += d.x + d.y;
    } else if (MODE == 4) {   // ds_write_b64 x4
      asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %1 offset:6144\n s_waitcnt lgkmcnt(0)"
                   : : "v"(o * 8), "v"(acc0) : "memory");
    } else if (MODE == 5) {   // ds_write_b128 x4
      d2 v; v.x = acc0; v.y = acc1;
      asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:16\n ds_write_b128 %0, %1 offset:32\n ds_write_b128 %0, %1 offset:48\n s_waitcnt lgkmcnt(0)"
                   : : "v"((o & ~1) * 8), "v"(v) : "memory");
    }
  }
  long long t1 = clock64();
  if (lane == 0) cyc[wv] = t1 - t0;
  out[t] = acc0 + acc1 + acc2 + acc3;
}
template <int MODE>
void run(const char* name, int threads, int stride_dw) {
  long long* cyc; double* out;
  hipMalloc(&cyc, 16 * 8); hipMalloc(&out, 1024 * 8);
  hipLaunchKernelGGL(k_lds<MODE>, dim3(1), dim3(threads), 65536, 0, cyc, out, stride_dw);
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  long long mx = 0; for (int i = 0; i < threads / 64; i++) mx = h[i] > mx ? h[i] : mx;
  printf("%-34s waves %2d stride %2d dw: %.1f cycles per group of 4 instr per wave (max over waves), %.1f per instr aggregated\n", name, threads / 64, stride_dw,
         (double)mx / N_IT, (double)mx / N_IT / 4 / (threads / 64));
  hipFree(cyc); hipFree(out);
}
int main() {
  for (int th : {64, 640}) {
    run<0>("ds_read_b64 distinct", th, 2);
    run<0>("ds_read_b64 distinct stride 18dw", th, 18);
    run<1>("ds_read_b64 broadcast", th, 2);
    run<2>("ds_read2_b64 distinct", th, 16);
    run<3>("ds_read_b128 distinct", th, 16);
    run<3>("ds_read_b128 distinct stride 4dw", th, 4);
    run<4>("ds_write_b64 distinct", th, 2);
    run<5>("ds_write_b128 distinct", th, 4);
  }
  return 0;
}
