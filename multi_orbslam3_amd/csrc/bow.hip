// liborbgpu -- bag-of-words side of the hot path, hand-written HIP for gfx950:
//   * DBoW2::TemplatedVocabulary<FORB>::transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1260): every
//     descriptor walks the vocabulary tree (first minimum of FORB::distance among the children, FORB.cpp:83-103) to its
//     word; BowVector / FeatureVector are assembled from the per-feature (word, node, weight) on the host, in the
//     reference's insertion order;
//   * MapPoint::ComputeDistinctiveDescriptors (S/MapPoint.cc:448-522) for a batch of map points: N x N Hamming
//     distances per point, median of every row, first row with the least median.
#include "common.hpp"
#include "wave.hpp"

#include <algorithm>
#include <cmath>
#include <map>

using namespace orbg;

int orbm_internal_features(orbm_frame* f, const uint8_t** d_desc, int* n, hipStream_t* stream);

namespace {

__device__ __forceinline__ int hamming32(const uint32_t* a, const uint32_t* b) {
  int d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d += __popc(a[i] ^ b[i]);
  return d;
}

// One thread per feature: the walk is a chain of L dependent "k children" comparisons -- pure latency, so the only
// thing that matters is that all features walk concurrently.
__global__ __launch_bounds__(256) void vocab_transform_kernel(const uint8_t* __restrict__ feat, int n, const int* __restrict__ child_start,
                                                             const int* __restrict__ child_ids, const uint8_t* __restrict__ node_desc,
                                                             const int* __restrict__ node_word, const double* __restrict__ node_weight,
                                                             int nid_level, int32_t* __restrict__ word_id, int32_t* __restrict__ node_id,
                                                             double* __restrict__ weight) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t f[8];
  {
    const uint4 f0 = *reinterpret_cast<const uint4*>(feat + (size_t)i * 32), f1 = *reinterpret_cast<const uint4*>(feat + (size_t)i * 32 + 16);
    f[0] = f0.x; f[1] = f0.y; f[2] = f0.z; f[3] = f0.w; f[4] = f1.x; f[5] = f1.y; f[6] = f1.z; f[7] = f1.w;
  }
  int nid = 0;                    // root when nid_level <= 0 (:1224); also the pinned value when a leaf comes earlier
  int final_id = 0, level = 0;
  int cs = child_start[0], ce = child_start[1];
  while (ce > cs) {               // do { ... } while (!isLeaf()) -- the root of a usable vocabulary has children
    ++level;
    int best = child_ids[cs];
    int best_d;
    {
      const uint4 d0 = *reinterpret_cast<const uint4*>(node_desc + (size_t)best * 32), d1 = *reinterpret_cast<const uint4*>(node_desc + (size_t)best * 32 + 16);
      const uint32_t nd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      best_d = hamming32(f, nd);
    }
    for (int c = cs + 1; c < ce; c++) {
      const int id = child_ids[c];
      const uint4 d0 = *reinterpret_cast<const uint4*>(node_desc + (size_t)id * 32), d1 = *reinterpret_cast<const uint4*>(node_desc + (size_t)id * 32 + 16);
      const uint32_t nd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      const int d = hamming32(f, nd);
      if (d < best_d) { best_d = d; best = id; }      // strict '<': first minimum wins (:1240-1244)
    }
    final_id = best;
    if (level == nid_level) nid = final_id;
    cs = child_start[final_id]; ce = child_start[final_id + 1];
  }
  word_id[i] = node_word[final_id];
  node_id[i] = nid;
  weight[i] = node_weight[final_id];
}

// One wavefront per map point.  Rows are kept in LDS (u16) while N <= kDdMax; the median of row i is found by rank
// counting (no sort): the element whose rank interval [#smaller, #smaller + #equal) holds k = (int)(0.5*(N-1)).
constexpr int kDdMax = 128;

__global__ __launch_bounds__(64) void distinctive_kernel(const uint8_t* __restrict__ desc, const int* __restrict__ start, int m,
                                                        int32_t* __restrict__ best_out) {
  __shared__ unsigned short dist[kDdMax * kDdMax];
  const int p = blockIdx.x, lane = threadIdx.x;
  if (p >= m) return;
  const int s = start[p], N = start[p + 1] - s;
  if (N <= 0) { if (lane == 0) best_out[p] = -1; return; }
  const uint32_t* D = reinterpret_cast<const uint32_t*>(desc) + (size_t)s * 8;
  const int k = (int)(0.5 * (double)(N - 1));
  unsigned best_key = 0xFFFFFFFFu;                 // (median << 16) | row : the minimum = least median, first row
  if (N <= kDdMax) {
    for (int e = lane; e < N * N; e += 64) {
      const int i = e / N, j = e - i * N;
      dist[i * kDdMax + j] = (unsigned short)(i == j ? 0 : hamming32(D + 8 * (size_t)i, D + 8 * (size_t)j));
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    for (int i = lane; i < N; i += 64) {
      const unsigned short* row = dist + i * kDdMax;
      int median = 0;
      for (int j = 0; j < N; j++) {
        const int v = row[j];
        int less = 0, eq = 0;
        for (int l = 0; l < N; l++) { less += row[l] < v; eq += row[l] == v; }
        if (less <= k && k < less + eq) { median = v; break; }
      }
      best_key = min(best_key, ((unsigned)median << 16) | (unsigned)i);
    }
  } else {
    // more observations than the LDS block holds: distances are recomputed on the fly (rare, O(N^3))
    for (int i = lane; i < N; i += 64) {
      int median = 0;
      for (int j = 0; j < N; j++) {
        const int v = i == j ? 0 : hamming32(D + 8 * (size_t)i, D + 8 * (size_t)j);
        int less = 0, eq = 0;
        for (int l = 0; l < N; l++) {
          const int w = i == l ? 0 : hamming32(D + 8 * (size_t)i, D + 8 * (size_t)l);
          less += w < v; eq += w == v;
        }
        if (less <= k && k < less + eq) { median = v; break; }
      }
      best_key = min(best_key, ((unsigned)median << 16) | (unsigned)(i & 0xFFFF));
    }
  }
  best_key = wave_min(best_key);
  if (lane == 0) best_out[p] = (int)(best_key & 0xFFFFu);
}

// double L1Scoring::score(v1, v2) (ScoringObject.cpp:23-68) of ONE query BowVector against m candidates (CSR), one thread per
// candidate: the merge walk and the accumulation order are the reference's, so the doubles are too.
__global__ __launch_bounds__(64) void bow_score_l1_kernel(const int* __restrict__ qw, const double* __restrict__ qv, int nq,
                                                         const int* __restrict__ cstart, const int* __restrict__ cw,
                                                         const double* __restrict__ cv, int m, double* __restrict__ score) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= m) return;
  int i = 0, j = cstart[c];
  const int je = cstart[c + 1];
  double s = 0;
  while (i < nq && j < je) {
    const int a = qw[i], b = cw[j];
    if (a == b) { const double vi = qv[i], wi = cv[j]; s += fabs(vi - wi) - fabs(vi) - fabs(wi); ++i; ++j; }
    else if (a < b) ++i;        // lower_bound(b) on a sorted list = advance to the first element >= b
    else ++j;
  }
  score[c] = -s / 2.0;
}


// ---------------------------------------------------------------------------------------------------------------------
// KeyFrameDatabase::DetectNBestCandidates (S/KeyFrameDatabase.cc:594-761) on the flattened database.
// db_walk_kernel: one workgroup per query word walks mvInvertedFile[word]: common-word count per keyframe (integer atomics:
// the result does not depend on the order) and the position of the keyframe's FIRST encounter (query word rank, list
// position) -- the order of lKFsSharingWords, which decides ties of the final stable sort.
__global__ __launch_bounds__(256) void db_walk_kernel(const int* __restrict__ qw, int nq, const int* __restrict__ inv_start,
                                                     const int* __restrict__ inv_kf, int n_words, int* __restrict__ words,
                                                     unsigned long long* __restrict__ first_key) {
  const int r = blockIdx.x;
  const int w = qw[r];
  if (w < 0 || w >= n_words) return;
  const int b0 = inv_start[w], e0 = inv_start[w + 1];
  for (int p = b0 + threadIdx.x; p < e0; p += 256) {
    const int kf = inv_kf[p];
    atomicAdd(&words[kf], 1);
    atomicMin(&first_key[kf], ((unsigned long long)r << 32) | (unsigned)(p - b0));
  }
}

// one workgroup: maxCommonWords over the keyframes that are not connected to the query (:636-644), minCommonWords =
// int(max * 0.8f) (:646), selection flags
__global__ __launch_bounds__(1024) void db_select_kernel(int n_kfs, const int* __restrict__ words, const uint8_t* __restrict__ connected,
                                                        uint8_t* __restrict__ shares, uint8_t* __restrict__ sel, int* __restrict__ out2) {
  __shared__ int s_max[16];
  int m = 0;
  for (int i = threadIdx.x; i < n_kfs; i += 1024)
    if (!connected[i] && words[i] > m) m = words[i];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
  __syncthreads();
  int mx = 0;
  for (int w = 0; w < 16; w++) mx = max(mx, s_max[w]);
  const int minw = (int)(mx * 0.8f);
  for (int i = threadIdx.x; i < n_kfs; i += 1024) {
    const bool sh = !connected[i] && words[i] > 0;
    const bool se = sh && words[i] > minw;
    shares[i] = sh; sel[i] = se;
  }
  if (threadIdx.x == 0) { out2[0] = mx; out2[1] = minw; }
}

// float si = mpVoc->score(pKF->mBowVec, pKFi->mBowVec) for the selected keyframes (:652-663), the walk of bow_score_l1_kernel
__global__ __launch_bounds__(64) void db_score_kernel(int n_kfs, const uint8_t* __restrict__ sel, const int* __restrict__ qw,
                                                     const double* __restrict__ qv, int nq, const int* __restrict__ bstart,
                                                     const int* __restrict__ bw, const double* __restrict__ bv, float* __restrict__ place_score) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= n_kfs || !sel[c]) return;
  int i = 0, j = bstart[c];
  const int je = bstart[c + 1];
  double s = 0;
  while (i < nq && j < je) {
    const int a = qw[i], b = bw[j];
    if (a == b) { const double vi = qv[i], wi = bv[j]; s += fabs(vi - wi) - fabs(vi) - fabs(wi); ++i; ++j; }
    else if (a < b) ++i;
    else ++j;
  }
  place_score[c] = (float)(-s / 2.0);
}

// covisibility accumulation (:673-701): the keyframe's own score plus those of its best-covisibility neighbours that share a
// word with the query (float sums in list order); the best-scoring keyframe of the group represents it
__global__ __launch_bounds__(64) void db_accumulate_kernel(int n_kfs, const uint8_t* __restrict__ sel, const uint8_t* __restrict__ shares,
                                                          const int* __restrict__ cstart, const int* __restrict__ ckf,
                                                          const float* __restrict__ place_score, float* __restrict__ acc_out,
                                                          int* __restrict__ best_out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= n_kfs || !sel[c]) return;
  float best = place_score[c], acc = best;
  int bk = c;
  for (int p = cstart[c]; p < cstart[c + 1]; p++) {
    const int k2 = ckf[p];
    if (!shares[k2]) continue;                        // pKF2->mnPlaceRecognitionQuery != pKF->mnId
    const float s2 = place_score[k2];
    acc += s2;
    if (s2 > best) { bk = k2; best = s2; }
  }
  acc_out[c] = acc; best_out[c] = bk;
}

}  // namespace

struct orbv_vocab {
  int device = 0;
  hipStream_t stream = nullptr;
  bool ext_stream = false;
  int n_nodes = 0, L = 0, weighting = 0, scoring_norm = 1;
  DevBuf<int> d_child_start, d_child_ids, d_word;
  DevBuf<uint8_t> d_desc;
  DevBuf<double> d_weight;
  DevBuf<uint8_t> d_feat;                 // staging for host descriptors
  DevBuf<int> d_out_word, d_out_node;
  DevBuf<double> d_out_weight;
  PinnedBuf<uint8_t> pin;
};

extern "C" int orbv_vocab_create(int device, const orbv_vocab_view* v, orbv_vocab** out) {
  if (!v || !out || v->n_nodes < 1 || !v->child_start || !v->desc || !v->weight || !v->word_id) return ORBG_BAD_ARG;
  const int nn = v->n_nodes;
  const int nc = v->child_start[nn];
  if (v->child_start[0] != 0 || nc < 0 || (nc > 0 && !v->child_ids)) return ORBG_BAD_ARG;
  for (int i = 0; i < nn; i++) if (v->child_start[i + 1] < v->child_start[i]) return ORBG_BAD_ARG;
  for (int i = 0; i < nn; i++)                  // children are created after their parent (m_nodes.push_back): ids grow downwards,
    for (int c = v->child_start[i]; c < v->child_start[i + 1]; c++)   // which also guarantees that every walk terminates
      if (v->child_ids[c] <= i || v->child_ids[c] >= nn) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  orbv_vocab* h = new orbv_vocab();
  h->device = device; h->n_nodes = nn; h->L = v->L; h->weighting = v->weighting; h->scoring_norm = v->scoring_norm;
  if (orbg::create_stream(&h->stream, "bow") != hipSuccess) { delete h; return ORBG_HIP_ERROR; }
  if ((rc = h->d_child_start.reserve(nn + 1)) || (rc = h->d_child_ids.reserve(std::max(nc, 1))) || (rc = h->d_word.reserve(nn)) ||
      (rc = h->d_desc.reserve((size_t)nn * 32)) || (rc = h->d_weight.reserve(nn))) { orbv_vocab_destroy(h); return rc; }
  ORBG_HIP(hipMemcpy(h->d_child_start.p, v->child_start, (size_t)(nn + 1) * 4, hipMemcpyHostToDevice));
  if (nc > 0) ORBG_HIP(hipMemcpy(h->d_child_ids.p, v->child_ids, (size_t)nc * 4, hipMemcpyHostToDevice));
  ORBG_HIP(hipMemcpy(h->d_word.p, v->word_id, (size_t)nn * 4, hipMemcpyHostToDevice));
  ORBG_HIP(hipMemcpy(h->d_desc.p, v->desc, (size_t)nn * 32, hipMemcpyHostToDevice));
  ORBG_HIP(hipMemcpy(h->d_weight.p, v->weight, (size_t)nn * 8, hipMemcpyHostToDevice));
  *out = h;
  return ORBG_OK;
}

extern "C" int orbv_vocab_set_stream(orbv_vocab* h, void* hip_stream) {
  if (!h) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  return orbg::swap_stream(&h->stream, &h->ext_stream, hip_stream, "bow");
}

extern "C" int orbv_vocab_destroy(orbv_vocab* h) {
  if (!h) return ORBG_BAD_ARG;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (!h->ext_stream) orbg::release_stream(h->stream);
  h->d_child_start.release(); h->d_child_ids.release(); h->d_word.release(); h->d_desc.release(); h->d_weight.release();
  h->d_feat.release(); h->d_out_word.release(); h->d_out_node.release(); h->d_out_weight.release(); h->pin.release();
  delete h;
  return ORBG_OK;
}

static int transform_common(orbv_vocab* h, const uint8_t* d_feat, int n, int levelsup, hipStream_t wait_on, int32_t* word_id,
                            int32_t* node_id, double* weight) {
  int rc;
  if ((rc = h->d_out_word.reserve(std::max(n, 1))) || (rc = h->d_out_node.reserve(std::max(n, 1))) ||
      (rc = h->d_out_weight.reserve(std::max(n, 1))) || (rc = h->pin.reserve((size_t)std::max(n, 1) * 16)))
    return rc;
  if (wait_on && wait_on != h->stream) ORBG_HIP(hipStreamSynchronize(wait_on));     // features produced on another handle's stream
  hipLaunchKernelGGL(vocab_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, d_feat, n, h->d_child_start.p,
                     h->d_child_ids.p, h->d_desc.p, h->d_word.p, h->d_weight.p, h->L - levelsup, h->d_out_word.p, h->d_out_node.p,
                     h->d_out_weight.p);
  ORBG_HIP(hipGetLastError());
  uint8_t* P = h->pin.h;
  ORBG_HIP(hipMemcpyAsync(P, h->d_out_weight.p, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
  ORBG_HIP(hipMemcpyAsync(P + (size_t)n * 8, h->d_out_word.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
  ORBG_HIP(hipMemcpyAsync(P + (size_t)n * 12, h->d_out_node.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
  ORBG_HIP(hipStreamSynchronize(h->stream));
  memcpy(weight, P, (size_t)n * 8);
  memcpy(word_id, P + (size_t)n * 8, (size_t)n * 4);
  memcpy(node_id, P + (size_t)n * 12, (size_t)n * 4);
  return ORBG_OK;
}

extern "C" int orbv_transform(orbv_vocab* h, const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id, double* weight) {
  if (!h || n < 0 || (n > 0 && (!desc || !word_id || !node_id || !weight))) return ORBG_BAD_ARG;
  if (n == 0) return ORBG_OK;
  int rc = select_device(h->device);
  if (rc) return rc;
  if ((rc = h->d_feat.reserve((size_t)n * 32))) return rc;
  ORBG_HIP(hipMemcpyAsync(h->d_feat.p, desc, (size_t)n * 32, hipMemcpyHostToDevice, h->stream));
  return transform_common(h, h->d_feat.p, n, levelsup, nullptr, word_id, node_id, weight);
}

extern "C" int orbv_transform_frame(orbv_vocab* h, orbm_frame* f, int levelsup, int32_t* word_id, int32_t* node_id, double* weight) {
  if (!h || !f || !word_id || !node_id || !weight) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  const uint8_t* d_desc = nullptr; int n = 0; hipStream_t fs = nullptr;
  if ((rc = orbm_internal_features(f, &d_desc, &n, &fs))) return rc;
  if (n == 0) return ORBG_OK;
  return transform_common(h, d_desc, n, levelsup, fs, word_id, node_id, weight);
}

// transform(features, BowVector&, FeatureVector&, levelsup) bookkeeping (TemplatedVocabulary.h:1127-1199) on the host: the
// std::map insertions of the reference replayed in feature order, so sums and ordering are the reference's.
extern "C" int orbv_bow_assemble(const orbv_vocab* h, const int32_t* word_id, const int32_t* node_id, const double* weight, int n,
                                 int32_t* bow_word, double* bow_value, int32_t* n_words, uint32_t* fv_node, uint32_t* fv_start,
                                 uint32_t* fv_feat, int32_t* n_fv_nodes) {
  if (!h || n < 0 || !n_words || !n_fv_nodes || (n > 0 && (!word_id || !node_id || !weight || !bow_word || !bow_value || !fv_node ||
                                                           !fv_start || !fv_feat)))
    return ORBG_BAD_ARG;
  std::map<int32_t, double> v;
  std::map<uint32_t, std::vector<uint32_t>> fv;
  const bool tf = h->weighting == ORBV_TF_IDF || h->weighting == ORBV_TF;
  for (int i = 0; i < n; i++) {
    const double w = weight[i];
    if (!(w > 0)) continue;                                         // stopped word (:1159)
    if (tf) v[word_id[i]] += w;                                     // BowVector::addWeight
    else v.emplace(word_id[i], w);                                  // BowVector::addIfNotExist
    fv[(uint32_t)node_id[i]].push_back((uint32_t)i);                // FeatureVector::addFeature
  }
  const bool must = h->scoring_norm != ORBV_NORM_NONE;
  if (tf && !v.empty() && !must) {
    const double nd = (double)v.size();
    for (auto& kv : v) kv.second /= nd;
  }
  if (must) {                                                       // BowVector::normalize (BowVector.cpp:62-84)
    double norm = 0.0;
    if (h->scoring_norm == ORBV_NORM_L1) for (auto& kv : v) norm += std::fabs(kv.second);
    else { for (auto& kv : v) norm += kv.second * kv.second; norm = std::sqrt(norm); }
    if (norm > 0.0) for (auto& kv : v) kv.second /= norm;
  }
  int k = 0;
  for (auto& kv : v) { bow_word[k] = kv.first; bow_value[k] = kv.second; k++; }
  *n_words = k;
  int nn = 0; uint32_t off = 0;
  for (auto& kv : fv) {
    fv_node[nn] = kv.first; fv_start[nn] = off;
    for (uint32_t fi : kv.second) fv_feat[off++] = fi;
    nn++;
  }
  if (n > 0) fv_start[nn] = off;
  *n_fv_nodes = nn;
  return ORBG_OK;
}

extern "C" int orbm_distinctive_descriptors(int device, const uint8_t* desc, const int32_t* start, int m, int32_t* best) {
  if (m < 0 || (m > 0 && (!start || !best))) return ORBG_BAD_ARG;
  if (m == 0) return ORBG_OK;
  const int total = start[m];
  if (start[0] != 0 || total < 0 || (total > 0 && !desc)) return ORBG_BAD_ARG;
  for (int i = 0; i < m; i++) if (start[i + 1] < start[i] || start[i + 1] - start[i] > 65535) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  struct Scratch { DevBuf<uint8_t> d_desc; DevBuf<int> d_start, d_best; int device = -1; };
  static thread_local Scratch sc;
  if (sc.device != device) { sc.d_desc.release(); sc.d_start.release(); sc.d_best.release(); sc.device = device; }
  if ((rc = sc.d_desc.reserve((size_t)std::max(total, 1) * 32)) || (rc = sc.d_start.reserve(m + 1)) || (rc = sc.d_best.reserve(m))) return rc;
  orbg::MiscStream ms;                                 // the library's M stream (never the legacy null stream)
  if ((rc = ms.open())) return rc;
  if (total > 0) ORBG_HIP(hipMemcpyAsync(sc.d_desc.p, desc, (size_t)total * 32, hipMemcpyHostToDevice, ms.s));
  ORBG_HIP(hipMemcpyAsync(sc.d_start.p, start, (size_t)(m + 1) * 4, hipMemcpyHostToDevice, ms.s));
  hipLaunchKernelGGL(distinctive_kernel, dim3(m), dim3(64), 0, ms.s, sc.d_desc.p, sc.d_start.p, m, sc.d_best.p);
  ORBG_HIP(hipGetLastError());
  ORBG_HIP(hipMemcpyAsync(best, sc.d_best.p, (size_t)m * 4, hipMemcpyDeviceToHost, ms.s));
  ORBG_HIP(hipStreamSynchronize(ms.s));
  return ORBG_OK;
}

extern "C" int orbv_score_l1(int device, const int32_t* q_word, const double* q_value, int nq, const int32_t* cand_start,
                             const int32_t* cand_word, const double* cand_value, int m, double* score) {
  if (nq < 0 || m < 0 || (m > 0 && (!cand_start || !score)) || (nq > 0 && (!q_word || !q_value))) return ORBG_BAD_ARG;
  if (m == 0) return ORBG_OK;
  const int total = cand_start[m];
  if (cand_start[0] != 0 || total < 0 || (total > 0 && (!cand_word || !cand_value))) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  struct Scratch { DevBuf<int> qw, cs, cw; DevBuf<double> qv, cv, sc; int device = -1; };
  static thread_local Scratch t;
  if (t.device != device) { t.qw.release(); t.cs.release(); t.cw.release(); t.qv.release(); t.cv.release(); t.sc.release(); t.device = device; }
  if ((rc = t.qw.reserve(std::max(nq, 1))) || (rc = t.qv.reserve(std::max(nq, 1))) || (rc = t.cs.reserve(m + 1)) ||
      (rc = t.cw.reserve(std::max(total, 1))) || (rc = t.cv.reserve(std::max(total, 1))) || (rc = t.sc.reserve(m)))
    return rc;
  orbg::MiscStream ms;
  if ((rc = ms.open())) return rc;
  if (nq > 0) {
    ORBG_HIP(hipMemcpyAsync(t.qw.p, q_word, (size_t)nq * 4, hipMemcpyHostToDevice, ms.s));
    ORBG_HIP(hipMemcpyAsync(t.qv.p, q_value, (size_t)nq * 8, hipMemcpyHostToDevice, ms.s));
  }
  ORBG_HIP(hipMemcpyAsync(t.cs.p, cand_start, (size_t)(m + 1) * 4, hipMemcpyHostToDevice, ms.s));
  if (total > 0) {
    ORBG_HIP(hipMemcpyAsync(t.cw.p, cand_word, (size_t)total * 4, hipMemcpyHostToDevice, ms.s));
    ORBG_HIP(hipMemcpyAsync(t.cv.p, cand_value, (size_t)total * 8, hipMemcpyHostToDevice, ms.s));
  }
  hipLaunchKernelGGL(bow_score_l1_kernel, dim3((m + 63) / 64), dim3(64), 0, ms.s, t.qw.p, t.qv.p, nq, t.cs.p, t.cw.p, t.cv.p, m, t.sc.p);
  ORBG_HIP(hipGetLastError());
  ORBG_HIP(hipMemcpyAsync(score, t.sc.p, (size_t)m * 8, hipMemcpyDeviceToHost, ms.s));
  ORBG_HIP(hipStreamSynchronize(ms.s));
  return ORBG_OK;
}


// ---------------------------------------------------------------- place recognition database
struct orbd_database {
  int device = 0;
  hipStream_t stream = nullptr;
  bool ext_stream = false;
  int n_kfs = 0, n_words = 0;
  DevBuf<int> inv_start, inv_kf, bow_start, bow_word, covis_start, covis_kf, words, best, out2, qw;
  DevBuf<double> bow_value, qv;
  DevBuf<uint8_t> connected, shares, sel;
  DevBuf<unsigned long long> first_key;
  DevBuf<float> place_score, acc;
  std::vector<int> map_id;
  std::vector<uint8_t> bad, map_bad;
  std::vector<int> h_words, h_best;
  std::vector<unsigned long long> h_key;
  std::vector<uint8_t> h_sel;
  std::vector<float> h_acc;
};

extern "C" int orbd_database_create(int device, const orbd_database_view* v, orbd_database** out) {
  if (!v || !out || v->n_kfs < 0 || v->n_words < 0) return ORBG_BAD_ARG;
  const int K = v->n_kfs, Wn = v->n_words;
  if ((Wn > 0 && !v->inv_start) || (K > 0 && (!v->bow_start || !v->covis_start || !v->map_id || !v->bad || !v->map_bad))) return ORBG_BAD_ARG;
  const int n_inv = Wn > 0 ? v->inv_start[Wn] : 0, n_bow = K > 0 ? v->bow_start[K] : 0, n_cov = K > 0 ? v->covis_start[K] : 0;
  if (n_inv < 0 || n_bow < 0 || n_cov < 0 || (n_inv > 0 && !v->inv_kf) || (n_bow > 0 && (!v->bow_word || !v->bow_value)) || (n_cov > 0 && !v->covis_kf))
    return ORBG_BAD_ARG;
  for (int i = 0; i < n_inv; i++) if (v->inv_kf[i] < 0 || v->inv_kf[i] >= K) return ORBG_BAD_ARG;
  for (int i = 0; i < n_cov; i++) if (v->covis_kf[i] < 0 || v->covis_kf[i] >= K) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  orbd_database* d = new orbd_database;
  d->device = device; d->n_kfs = K; d->n_words = Wn;
  if (orbg::create_stream(&d->stream, "db") != hipSuccess) { delete d; return ORBG_HIP_ERROR; }
  auto up = [&](auto& buf, const auto* src, size_t n) -> int {
    int r = buf.reserve(std::max<size_t>(n, 1));
    if (r) return r;
    if (n > 0) ORBG_HIP(hipMemcpyAsync(buf.p, src, n * sizeof(*src), hipMemcpyHostToDevice, d->stream));
    return ORBG_OK;
  };
  if ((rc = up(d->inv_start, v->inv_start, Wn > 0 ? (size_t)Wn + 1 : 0)) || (rc = up(d->inv_kf, v->inv_kf, n_inv)) ||
      (rc = up(d->bow_start, v->bow_start, K > 0 ? (size_t)K + 1 : 0)) || (rc = up(d->bow_word, v->bow_word, n_bow)) ||
      (rc = up(d->bow_value, v->bow_value, n_bow)) || (rc = up(d->covis_start, v->covis_start, K > 0 ? (size_t)K + 1 : 0)) ||
      (rc = up(d->covis_kf, v->covis_kf, n_cov)) || (rc = d->words.reserve(std::max(K, 1))) || (rc = d->best.reserve(std::max(K, 1))) ||
      (rc = d->out2.reserve(4)) || (rc = d->connected.reserve(std::max(K, 1))) || (rc = d->shares.reserve(std::max(K, 1))) ||
      (rc = d->sel.reserve(std::max(K, 1))) || (rc = d->first_key.reserve(std::max(K, 1))) || (rc = d->place_score.reserve(std::max(K, 1))) ||
      (rc = d->acc.reserve(std::max(K, 1)))) {
    orbd_database_destroy(d);
    return rc;
  }
  d->map_id.assign(v->map_id, v->map_id + K); d->bad.assign(v->bad, v->bad + K); d->map_bad.assign(v->map_bad, v->map_bad + K);
  if (hipStreamSynchronize(d->stream) != hipSuccess) { orbd_database_destroy(d); return ORBG_HIP_ERROR; }
  *out = d;
  return ORBG_OK;
}

extern "C" int orbd_database_set_stream(orbd_database* d, void* hip_stream) {
  if (!d) return ORBG_BAD_ARG;
  int rc = select_device(d->device);
  if (rc) return rc;
  return orbg::swap_stream(&d->stream, &d->ext_stream, hip_stream, "db");
}

extern "C" int orbd_database_destroy(orbd_database* d) {
  if (!d) return ORBG_OK;
  (void)hipSetDevice(d->device);
  d->inv_start.release(); d->inv_kf.release(); d->bow_start.release(); d->bow_word.release(); d->covis_start.release(); d->covis_kf.release();
  d->words.release(); d->best.release(); d->out2.release(); d->qw.release(); d->bow_value.release(); d->qv.release(); d->connected.release();
  d->shares.release(); d->sel.release(); d->first_key.release(); d->place_score.release(); d->acc.release();
  (void)hipStreamSynchronize(d->stream);
  if (!d->ext_stream) orbg::release_stream(d->stream);
  delete d;
  return ORBG_OK;
}

extern "C" int orbd_detect_n_best_candidates(orbd_database* d, const int32_t* q_word, const double* q_value, int nq, const uint8_t* connected,
                                             int32_t query_map_id, int n_candidates, float* place_score, int32_t* loop_cand, int32_t* n_loop,
                                             int32_t* merge_cand, int32_t* n_merge) {
  if (!d || nq < 0 || (nq > 0 && (!q_word || !q_value)) || n_candidates < 0 || !n_loop || !n_merge || (n_candidates > 0 && (!loop_cand || !merge_cand)))
    return ORBG_BAD_ARG;
  const int K = d->n_kfs;
  if (K > 0 && (!connected || !place_score)) return ORBG_BAD_ARG;
  *n_loop = 0; *n_merge = 0;
  if (K == 0 || nq == 0) return ORBG_OK;
  int rc = select_device(d->device);
  if (rc) return rc;
  hipStream_t st = d->stream;
  if ((rc = d->qw.reserve(nq)) || (rc = d->qv.reserve(nq))) return rc;
  ORBG_HIP(hipMemcpyAsync(d->qw.p, q_word, (size_t)nq * 4, hipMemcpyHostToDevice, st));
  ORBG_HIP(hipMemcpyAsync(d->qv.p, q_value, (size_t)nq * 8, hipMemcpyHostToDevice, st));
  ORBG_HIP(hipMemcpyAsync(d->connected.p, connected, (size_t)K, hipMemcpyHostToDevice, st));
  ORBG_HIP(hipMemcpyAsync(d->place_score.p, place_score, (size_t)K * 4, hipMemcpyHostToDevice, st));
  ORBG_HIP(hipMemsetAsync(d->words.p, 0, (size_t)K * 4, st));
  ORBG_HIP(hipMemsetAsync(d->first_key.p, 0xFF, (size_t)K * 8, st));
  hipLaunchKernelGGL(db_walk_kernel, dim3(nq), dim3(256), 0, st, d->qw.p, nq, d->inv_start.p, d->inv_kf.p, d->n_words, d->words.p, d->first_key.p);
  hipLaunchKernelGGL(db_select_kernel, dim3(1), dim3(1024), 0, st, K, d->words.p, d->connected.p, d->shares.p, d->sel.p, d->out2.p);
  hipLaunchKernelGGL(db_score_kernel, dim3((K + 63) / 64), dim3(64), 0, st, K, d->sel.p, d->qw.p, d->qv.p, nq, d->bow_start.p, d->bow_word.p,
                     d->bow_value.p, d->place_score.p);
  hipLaunchKernelGGL(db_accumulate_kernel, dim3((K + 63) / 64), dim3(64), 0, st, K, d->sel.p, d->shares.p, d->covis_start.p, d->covis_kf.p,
                     d->place_score.p, d->acc.p, d->best.p);
  ORBG_HIP(hipGetLastError());
  d->h_sel.resize(K); d->h_acc.resize(K); d->h_best.resize(K); d->h_key.resize(K);
  ORBG_HIP(hipMemcpyAsync(d->h_sel.data(), d->sel.p, (size_t)K, hipMemcpyDeviceToHost, st));
  ORBG_HIP(hipMemcpyAsync(d->h_acc.data(), d->acc.p, (size_t)K * 4, hipMemcpyDeviceToHost, st));
  ORBG_HIP(hipMemcpyAsync(d->h_best.data(), d->best.p, (size_t)K * 4, hipMemcpyDeviceToHost, st));
  ORBG_HIP(hipMemcpyAsync(d->h_key.data(), d->first_key.p, (size_t)K * 8, hipMemcpyDeviceToHost, st));
  ORBG_HIP(hipMemcpyAsync(place_score, d->place_score.p, (size_t)K * 4, hipMemcpyDeviceToHost, st));
  ORBG_HIP(hipStreamSynchronize(st));
  // lScoreAndMatch order = order of first encounter in the inverted-file walk; list::sort(compFirst) is a stable sort
  struct Ent { unsigned long long key; float acc; int best; };
  std::vector<Ent> ents;
  for (int i = 0; i < K; i++)
    if (d->h_sel[i]) ents.push_back(Ent{d->h_key[i], d->h_acc[i], d->h_best[i]});
  std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.key < b.key; });
  std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.acc > b.acc; });
  std::vector<uint8_t> added(K, 0);
  for (const Ent& e : ents) {
    if (!(*n_loop < n_candidates || *n_merge < n_candidates)) break;              // :714
    const int k = e.best;
    if (d->bad[k]) continue;                                                       // pinned: skipped (see orbgpu.h)
    if (added[k]) continue;
    if (d->map_id[k] == query_map_id && *n_loop < n_candidates) loop_cand[(*n_loop)++] = k;
    else if (d->map_id[k] != query_map_id && *n_merge < n_candidates && !d->map_bad[k]) merge_cand[(*n_merge)++] = k;
    added[k] = 1;
  }
  return ORBG_OK;
}
