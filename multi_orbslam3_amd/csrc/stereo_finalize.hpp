// Pieces shared by extractor.hip and matcher.hip: the tail of the stereo Frame constructor (median-of-SAD rejection of
// ComputeStereoMatches, S/Frame.cc:949-962) runs either as a kernel of its own or as the second workgroup of the grid build.
#pragma once

#include <hip/hip_runtime.h>

#include "wave.hpp"

using orbg::wave_incl_scan_add;

// Packed (iniTh count | minTh count << 16) exclusive scan over a 256-thread workgroup.
__device__ __forceinline__ unsigned block_excl_scan_256(unsigned v, unsigned* total, unsigned* wsum /*LDS[4]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned inc = wave_incl_scan_add(v);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const unsigned s = wsum[w];
    if (w < wave) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}

// arguments of stereo_finalize_body for the kernel that runs it next to the grid build (matcher.hip)
struct StereoFinalizeArgs {
  float* uright; float* depth; const int* best_sad; int nl; const int* d_nkp; float* host_out;
  unsigned* ticket;                  // device word, zero between launches: the workgroup that finishes second posts the completion word
};

// median-of-SAD outlier rejection (:949-962), one 256-thread workgroup: the element vDistIdx[size/2].first of the
// sorted list is found by a two-level (high byte / low byte) histogram select -- SAD <= 121*510 < 2^16 -- instead
// of sorting; then every match whose SAD is not below 1.5f*1.4f*median is dropped.
// Round 4: the three passes over (best_sad, uright, depth) read them ONCE -- up to kSfKeep values per thread stay in registers
// (frames of up to 256 * kSfKeep left keypoints; larger ones re-read the rest) -- so the body is one trip to memory, two LDS
// histograms and one store pass instead of three dependent trips.
constexpr int kSfKeep = 8;
__device__ __forceinline__ void stereo_finalize_body(float* __restrict__ uright, float* __restrict__ depth,
                                                     const int* __restrict__ best_sad, int nl, const int* __restrict__ d_nkp,
                                                     float* __restrict__ host_out) {
  __shared__ unsigned hist[256];
  if (d_nkp) nl = d_nkp[0];
  __shared__ unsigned wsum[4];
  __shared__ int s_bin, s_before;
  const int tid = threadIdx.x;
  int sv[kSfKeep];
  float su[kSfKeep], sd[kSfKeep];
#pragma unroll
  for (int q = 0; q < kSfKeep; q++) {
    const int i = tid + 256 * q;
    const bool in = i < nl;
    sv[q] = in ? best_sad[i] : -1;
    su[q] = in ? uright[i] : -1.0f;
    sd[q] = in ? depth[i] : -1.0f;
  }
  // ---- pass 0: histogram of the high byte -> bin holding the element of rank kth = n/2
  hist[tid] = 0;
  if (tid == 0) { s_bin = -1; s_before = 0; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kSfKeep; q++) if (sv[q] >= 0) atomicAdd(&hist[(sv[q] >> 8) & 255], 1u);
  for (int i = tid + 256 * kSfKeep; i < nl; i += 256) {
    const int v = best_sad[i];
    if (v >= 0) atomicAdd(&hist[(v >> 8) & 255], 1u);
  }
  __syncthreads();
  unsigned total;
  unsigned mine = hist[tid];
  unsigned excl = block_excl_scan_256(mine, &total, wsum);
  if (total == 0) {                                 // no stereo match at all (uniform exit)
    if (host_out) for (int i = tid; i < nl; i += 256) { host_out[i] = uright[i]; host_out[nl + i] = depth[i]; }
    return;
  }
  const unsigned kth = total / 2;
  if (excl <= kth && kth < excl + mine) { s_bin = tid; s_before = (int)excl; }
  __syncthreads();
  const int hi = s_bin;
  const unsigned before = (unsigned)s_before;
  __syncthreads();
  // ---- pass 1: histogram of the low byte inside that bin
  hist[tid] = 0;
  if (tid == 0) s_bin = -1;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kSfKeep; q++) if (sv[q] >= 0 && (sv[q] >> 8) == hi) atomicAdd(&hist[sv[q] & 255], 1u);
  for (int i = tid + 256 * kSfKeep; i < nl; i += 256) {
    const int v = best_sad[i];
    if (v >= 0 && (v >> 8) == hi) atomicAdd(&hist[v & 255], 1u);
  }
  __syncthreads();
  mine = hist[tid];
  excl = block_excl_scan_256(mine, &total, wsum);
  if (before + excl <= kth && kth < before + excl + mine) s_bin = tid;
  __syncthreads();
  const int median = (hi << 8) | s_bin;
  const float thDist = 1.5f * 1.4f * (float)median;
#pragma unroll
  for (int q = 0; q < kSfKeep; q++) {
    const int i = tid + 256 * q;
    if (i < nl) {
      float u = su[q], d = sd[q];
      if (sv[q] >= 0 && !((float)sv[q] < thDist)) { u = -1; d = -1; uright[i] = u; depth[i] = d; }
      if (host_out) { host_out[i] = u; host_out[nl + i] = d; }     // mirror into mapped pinned memory: [uRight | depth]
    }
  }
  for (int i = tid + 256 * kSfKeep; i < nl; i += 256) {
    const int v = best_sad[i];
    float u = uright[i], d = depth[i];
    if (v >= 0 && !((float)v < thDist)) { u = -1; d = -1; uright[i] = u; depth[i] = d; }
    if (host_out) { host_out[i] = u; host_out[nl + i] = d; }
  }
}
