// ORACLE (test infrastructure, not product code) -- see orb_oracle.h for status: parity unpinned.
//
// CPU restatement of the bag-of-words pieces next to the hot path:
//   DBoW2::TemplatedVocabulary<FORB>::transform  (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1260, FORB.cpp:83-103,
//   BowVector.cpp:26-84, FeatureVector.cpp:31-45) and MapPoint::ComputeDistinctiveDescriptors (S/MapPoint.cc:448-522).
#include "orb_oracle.h"

#include <algorithm>
#include <list>
#include <set>
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

// void transform(const TDescriptor &feature, WordId &word_id, WordValue &weight, NodeId *nid, int levelsup) -- :1214-1260
extern "C" int oracle_vocab_transform(const orbv_vocab_view* v, const uint8_t* desc, int n, int levelsup, int32_t* word_id,
                                      int32_t* node_id, double* weight) {
  if (!v || n < 0) return ORBG_BAD_ARG;
  const int nid_level = v->L - levelsup;
  for (int i = 0; i < n; i++) {
    const uint8_t* f = desc + 32 * (size_t)i;
    int nid = 0;                                   // "if(nid_level <= 0 && nid != NULL) *nid = 0" (:1224); pinned to 0 otherwise too
    int final_id = 0, current_level = 0;
    while (v->child_start[final_id + 1] > v->child_start[final_id]) {
      ++current_level;
      const int cs = v->child_start[final_id], ce = v->child_start[final_id + 1];
      final_id = v->child_ids[cs];
      double best_d = oracle_hamming(f, v->desc + 32 * (size_t)final_id);
      for (int c = cs + 1; c < ce; c++) {
        const int id = v->child_ids[c];
        const double d = oracle_hamming(f, v->desc + 32 * (size_t)id);
        if (d < best_d) { best_d = d; final_id = id; }
      }
      if (current_level == nid_level) nid = final_id;
    }
    word_id[i] = v->word_id[final_id];
    weight[i] = v->weight[final_id];
    node_id[i] = nid;
  }
  return ORBG_OK;
}

// void transform(const vector<TDescriptor>& features, BowVector &v, FeatureVector &fv, int levelsup) -- :1127-1199
extern "C" int oracle_vocab_bow(const orbv_vocab_view* voc, const uint8_t* desc, int n, int levelsup, int32_t* bow_word,
                                double* bow_value, int32_t* n_words, uint32_t* fv_node, uint32_t* fv_start, uint32_t* fv_feat,
                                int32_t* n_fv_nodes) {
  std::vector<int32_t> wid(std::max(n, 1)), nid(std::max(n, 1));
  std::vector<double> w(std::max(n, 1));
  int rc = oracle_vocab_transform(voc, desc, n, levelsup, wid.data(), nid.data(), w.data());
  if (rc) return rc;
  std::map<int32_t, double> v;                              // BowVector: std::map<WordId, WordValue>
  std::map<uint32_t, std::vector<uint32_t>> fv;             // FeatureVector: std::map<NodeId, std::vector<unsigned int>>
  const bool tf = voc->weighting == ORBV_TF_IDF || voc->weighting == ORBV_TF;
  for (int i = 0; i < n; i++) {
    if (!(w[i] > 0)) continue;                              // "if(w > 0) // not stopped"
    if (tf) {                                               // BowVector::addWeight
      auto it = v.lower_bound(wid[i]);
      if (it != v.end() && !(v.key_comp()(wid[i], it->first))) it->second += w[i];
      else v.insert(it, std::make_pair(wid[i], w[i]));
    } else {                                                // BowVector::addIfNotExist
      auto it = v.lower_bound(wid[i]);
      if (it == v.end() || v.key_comp()(wid[i], it->first)) v.insert(it, std::make_pair(wid[i], w[i]));
    }
    fv[(uint32_t)nid[i]].push_back((uint32_t)i);            // FeatureVector::addFeature
  }
  const bool must = voc->scoring_norm != ORBV_NORM_NONE;
  if (tf && !v.empty() && !must) {
    const double nd = (double)v.size();
    for (auto& kv : v) kv.second /= nd;
  }
  if (must) {
    double norm = 0.0;
    if (voc->scoring_norm == ORBV_NORM_L1) { for (auto& kv : v) norm += std::fabs(kv.second); }
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
  fv_start[nn] = off;
  *n_fv_nodes = nn;
  return ORBG_OK;
}

// MapPoint::ComputeDistinctiveDescriptors -- S/MapPoint.cc:448-522, batched over m points (CSR lists of descriptors)
extern "C" int oracle_distinctive_descriptors(const uint8_t* desc, const int32_t* start, int m, int32_t* best) {
  for (int p = 0; p < m; p++) {
    const int N = start[p + 1] - start[p];
    if (N <= 0) { best[p] = -1; continue; }                 // vDescriptors.empty(): return
    const uint8_t* D = desc + 32 * (size_t)start[p];
    std::vector<float> Distances((size_t)N * N);            // float Distances[N][N]
    for (int i = 0; i < N; i++) {
      Distances[(size_t)i * N + i] = 0;
      for (int j = i + 1; j < N; j++) {
        const int dij = oracle_hamming(D + 32 * (size_t)i, D + 32 * (size_t)j);
        Distances[(size_t)i * N + j] = dij; Distances[(size_t)j * N + i] = dij;
      }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (int i = 0; i < N; i++) {
      std::vector<int> vDists(Distances.begin() + (size_t)i * N, Distances.begin() + (size_t)(i + 1) * N);
      std::sort(vDists.begin(), vDists.end());
      const int median = vDists[(size_t)(0.5 * (N - 1))];
      if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    best[p] = BestIdx;
  }
  return ORBG_OK;
}

// double L1Scoring::score(const BowVector &v1, const BowVector &v2) -- Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68
extern "C" int oracle_score_l1(const int32_t* q_word, const double* q_value, int nq, const int32_t* cand_start,
                               const int32_t* cand_word, const double* cand_value, int m, double* score_out) {
  for (int c = 0; c < m; c++) {
    const int32_t* w2 = cand_word + cand_start[c];
    const double* v2 = cand_value + cand_start[c];
    const int n2 = cand_start[c + 1] - cand_start[c];
    int i = 0, j = 0;
    double score = 0;
    while (i < nq && j < n2) {
      const double vi = q_value[i], wi = v2[j];
      if (q_word[i] == w2[j]) { score += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi); ++i; ++j; }
      else if (q_word[i] < w2[j]) i = (int)(std::lower_bound(q_word + i, q_word + nq, w2[j]) - q_word);
      else j = (int)(std::lower_bound(w2 + j, w2 + n2, q_word[i]) - w2);
    }
    score_out[c] = -score / 2.0;
  }
  return ORBG_OK;
}

// KF wire block: Converter::toCvKeyPointMsg / fromCvKeyPointMsg (S/Converter.cc:217-245), R/msg/CvKeyPoint.msg, R/msg/Descriptor.msg
extern "C" int oracle_wire_pack(const orbx_keypoint* kps, const uint8_t* desc, int n, uint8_t* wire) {
  for (int i = 0; i < n; i++) {
    uint8_t* p = wire + (size_t)i * 15;
    std::memcpy(p, &kps[i].x, 4); std::memcpy(p + 4, &kps[i].y, 4);
    p[8] = (uint8_t)kps[i].size;                  // Msg.size = (u_int8_t)kp.size
    std::memcpy(p + 9, &kps[i].angle, 4);
    p[13] = (uint8_t)kps[i].response;             // Msg.response = (u_int8_t)kp.response
    p[14] = (uint8_t)(int8_t)kps[i].octave;       // int8 octave
    std::memcpy(wire + (size_t)n * 15 + (size_t)i * 32, desc + (size_t)i * 32, 32);
  }
  return ORBG_OK;
}

extern "C" int oracle_wire_unpack(const uint8_t* wire, int n, orbx_keypoint* kps, uint8_t* desc) {
  for (int i = 0; i < n; i++) {
    const uint8_t* p = wire + (size_t)i * 15;
    std::memcpy(&kps[i].x, p, 4); std::memcpy(&kps[i].y, p + 4, 4);
    kps[i].size = (float)p[8];
    std::memcpy(&kps[i].angle, p + 9, 4);
    kps[i].response = (float)p[13];
    kps[i].octave = (int)(int8_t)p[14];
    std::memcpy(desc + (size_t)i * 32, wire + (size_t)n * 15 + (size_t)i * 32, 32);
  }
  return ORBG_OK;
}


// void KeyFrameDatabase::DetectNBestCandidates(KeyFrame* pKF, vector<KeyFrame*>& vpLoopCand, vector<KeyFrame*>& vpMergeCand,
//                                              int nNumCandidates) -- S/KeyFrameDatabase.cc:594-761, on the flattened database
// (orbd_database_view, keyframes = indices).  Per-keyframe members of the reference become arrays: mnPlaceRecognitionQuery
// -> `queried` (every query has a fresh id), mnPlaceRecognitionWords -> `words`, mPlaceRecognitionScore -> place_score
// (in/out: it survives from query to query in the reference and the covisibility accumulation reads stale values, :688).
extern "C" int oracle_detect_n_best_candidates(const orbd_database_view* v, const int32_t* q_word, const double* q_value, int nq,
                                               const uint8_t* connected, int32_t query_map_id, int n_candidates, float* place_score,
                                               int32_t* loop_cand, int32_t* n_loop, int32_t* merge_cand, int32_t* n_merge) {
  *n_loop = 0; *n_merge = 0;
  const int K = v->n_kfs;
  std::vector<uint8_t> queried(K, 0);
  std::vector<int> words(K, 0);
  std::list<int> lKFsSharingWords;
  for (int r = 0; r < nq; r++) {                                                  // :605-633
    const int w = q_word[r];
    if (w < 0 || w >= v->n_words) continue;
    for (int p = v->inv_start[w]; p < v->inv_start[w + 1]; p++) {
      const int kf = v->inv_kf[p];
      if (!queried[kf]) {
        words[kf] = 0;
        if (!connected[kf]) { queried[kf] = 1; lKFsSharingWords.push_back(kf); }
      }
      words[kf]++;
    }
  }
  if (lKFsSharingWords.empty()) return ORBG_OK;
  int maxCommonWords = 0;                                                          // :636-644
  for (int kf : lKFsSharingWords) if (words[kf] > maxCommonWords) maxCommonWords = words[kf];
  const int minCommonWords = (int)(maxCommonWords * 0.8f);                         // :646
  std::list<std::pair<float, int>> lScoreAndMatch;
  for (int kf : lKFsSharingWords) {                                                // :652-663
    if (words[kf] > minCommonWords) {
      double sc = 0;
      const int one[2] = {0, v->bow_start[kf + 1] - v->bow_start[kf]};
      oracle_score_l1(q_word, q_value, nq, one, v->bow_word + v->bow_start[kf], v->bow_value + v->bow_start[kf], 1, &sc);
      const float si = (float)sc;
      place_score[kf] = si;
      lScoreAndMatch.push_back(std::make_pair(si, kf));
    }
  }
  if (lScoreAndMatch.empty()) return ORBG_OK;
  std::list<std::pair<float, int>> lAccScoreAndMatch;
  for (const auto& it : lScoreAndMatch) {                                          // :673-701
    const int kf = it.second;
    float bestScore = it.first, accScore = bestScore;
    int best = kf;
    for (int p = v->covis_start[kf]; p < v->covis_start[kf + 1]; p++) {
      const int k2 = v->covis_kf[p];
      if (!queried[k2]) continue;
      accScore += place_score[k2];
      if (place_score[k2] > bestScore) { best = k2; bestScore = place_score[k2]; }
    }
    lAccScoreAndMatch.push_back(std::make_pair(accScore, best));
  }
  lAccScoreAndMatch.sort([](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first; });   // :705, compFirst :588-591
  std::set<int> added;
  for (const auto& it : lAccScoreAndMatch) {                                       // :713-735
    if (!(*n_loop < n_candidates || *n_merge < n_candidates)) break;
    const int kf = it.second;
    if (v->bad[kf]) continue;       // pinned: the reference's `continue` (:718-719) skips the iterator increment and spins forever
    if (!added.count(kf)) {
      if (v->map_id[kf] == query_map_id && *n_loop < n_candidates) loop_cand[(*n_loop)++] = kf;
      else if (v->map_id[kf] != query_map_id && *n_merge < n_candidates && !v->map_bad[kf]) merge_cand[(*n_merge)++] = kf;
      added.insert(kf);
    }
  }
  return ORBG_OK;
}
