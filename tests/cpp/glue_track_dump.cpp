// Host only (the glue over the CPU oracle's entry points): a synthetic Frame and local map go through include/orbgpu_dropin.hpp's
// SearchLocalPoints; the scene before the call and everything the call left behind -- F.mvpMapPoints, every point's mnLastFrameSeen /
// visible count / mbTrackInView, the return value -- are printed as JSON.  tests/test_reference_formulas.py rebuilds the scene as
// Python stand-ins and runs Tracking::SearchLocalPoints' own text (with Frame::isInFrustum and ORBmatcher::SearchByProjection,
// transliterated) on it.
//   g++ -O1 -std=c++17 -I include -I tests/cpp tests/cpp/glue_track_dump.cpp -L multi_orbslam3_amd -lorbgpu -L oracle -loracle -o glue_track_dump
#include <cstdio>
#include <cstring>
#include <vector>

#include "scenario.hpp"
#include "oracle_ops.hpp"

static void dump_floats(const char* name, const float* v, size_t n) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < n; i++) std::printf("%s%.9g", i ? ", " : "", v[i]);
  std::printf("]");
}
static void dump_bytes(const char* name, const uint8_t* v, size_t n) {
  std::printf("\"%s\": \"", name);
  for (size_t i = 0; i < n; i++) std::printf("%02x", v[i]);
  std::printf("\"");
}

int main() {
  for (int scene = 0; scene < 3; scene++) {
    g_seed = 777u + scene;
    const int N = 900, M = scene == 2 ? 40 : 800;
    Agent A;
    std::unique_ptr<Frame> Fp(new Frame); Frame& F = *Fp;
    F.mnId = 4321 + scene; F.N = N; F.mnMinX = 0; F.mnMaxX = W; F.mnMinY = 0; F.mnMaxY = H; F.fx = FX; F.fy = FX; F.cx = CX; F.cy = CY; F.mbf = BF; F.mb = BB;
    F.mvKeysUn.resize(N); F.mDescriptors = Mat(N, 32, 1); F.mvuRight.resize(N); F.mvDepth.resize(N);
    std::vector<float> depth(N);
    for (int i = 0; i < N; i++) {
      F.mvKeysUn[i] = KeyPoint{{(float)(5 + 630 * urand()), (float)(5 + 470 * urand())}, 31.f, (float)(360 * urand()), 20.f, (int)(rnd() % 8)};
      for (int b = 0; b < 32; b++) F.mDescriptors.ptr<uint8_t>(i)[b] = (uint8_t)(rnd() & 255);
      depth[i] = (float)(2 + 18 * urand());
      const bool stereo = urand() < 0.7;
      F.mvuRight[i] = stereo ? F.mvKeysUn[i].pt.x - BF / depth[i] : -1.f;
      F.mvDepth[i] = stereo ? depth[i] : -1.f;
    }
    F.mvKeys = F.mvKeysUn;
    F.mvpMapPoints.assign(N, nullptr); F.mvbOutlier.assign(N, false);
    double T[16]; pose_of(3, T);
    F.mTcw = mat44(T);
    float sf[8]; sf[0] = 1.f; for (int l = 1; l < 8; l++) sf[l] = sf[l - 1] * 1.2f;
    // local map: most points re-project next to a feature and carry its descriptor with a few bits flipped
    std::vector<MapPoint*> local;
    std::vector<int> target(M);
    for (int j = 0; j < M; j++) {
      const int t = (int)(rnd() % N); target[j] = t;
      const float z = depth[t] * (float)(0.97 + 0.06 * urand());
      const float u = F.mvKeysUn[t].pt.x + (float)(8 * urand() - 4), v = F.mvKeysUn[t].pt.y + (float)(8 * urand() - 4);
      float zz = z;
      if (j % 23 == 5) zz = -z;                                     // behind the camera
      const double Pc[3] = {(u - CX) * zz / FX, (v - CY) * zz / FX, zz};
      std::unique_ptr<MapPoint> p(new MapPoint);
      float* X = p->mWorldPos.ptr<float>(0);
      double Ow[3];
      for (int a = 0; a < 3; a++) { X[a] = (float)(T[a] * (Pc[0] - T[3]) + T[4 + a] * (Pc[1] - T[7]) + T[8 + a] * (Pc[2] - T[11])); Ow[a] = -(T[a] * T[3] + T[4 + a] * T[7] + T[8 + a] * T[11]); }
      double PO[3] = {X[0] - Ow[0], X[1] - Ow[1], X[2] - Ow[2]};
      const double dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
      for (int a = 0; a < 3; a++) p->mNormalVector.ptr<float>(0)[a] = (float)(PO[a] / dist + (j % 7 == 3 ? 1.5 : 0.1) * nrand());
      { float* nv = p->mNormalVector.ptr<float>(0); const float nn = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]); for (int a = 0; a < 3; a++) nv[a] /= nn; }
      const int oct = std::min(7, std::max(0, F.mvKeysUn[t].octave + (int)(rnd() % 2)));
      p->mfMaxDistance = (float)(dist * sf[oct] * 0.93) * (j % 11 == 7 ? 0.4f : 1.f); p->mfMinDistance = p->mfMaxDistance / sf[7];
      for (int b = 0; b < 32; b++) p->mDescriptor.ptr<uint8_t>(0)[b] = F.mDescriptors.ptr<uint8_t>(t)[b] ^ (uint8_t)(rnd() & rnd() & rnd() & 255);
      p->mnId = 100 + j; p->nObs = (int)(rnd() % 4); p->mpMap = &A.map; p->mbBad = (j % 29 == 11);
      local.push_back(p.get());
      A.points.push_back(std::move(p));
    }
    // the frame already holds some points (the motion-model search): a few of the local map, a few others, one of them bad
    for (int k = 0; k < 120; k++) {
      const int i = (int)(rnd() % N);
      if (F.mvpMapPoints[i]) continue;
      if (k % 3 == 0) { F.mvpMapPoints[i] = local[rnd() % M]; continue; }
      std::unique_ptr<MapPoint> p(new MapPoint);
      p->mnId = 50000 + k; p->nObs = (int)(rnd() % 3); p->mpMap = &A.map; p->mbBad = (k % 17 == 4);
      F.mvpMapPoints[i] = p.get();
      A.points.push_back(std::move(p));
    }
    const float th = scene == 1 ? 5.f : 1.f; const bool far_pts = scene == 1; const float th_far = 12.f;
    // ---- scene before the call
    std::printf("{\"scene\": %d, \"frame\": {\"id\": %lu, \"N\": %d, \"th\": %.9g, \"far\": %d, \"th_far\": %.9g, \"fx\": %.9g, \"fy\": %.9g, \"cx\": %.9g, \"cy\": %.9g, \"mbf\": %.9g, \"mb\": %.9g, ", scene, F.mnId, N, th,
                (int)far_pts, th_far, F.fx, F.fy, F.cx, F.cy, F.mbf, F.mb);
    dump_floats("Tcw", F.mTcw.ptr<float>(0), 16);
    std::printf(", \"keys\": [");
    for (int i = 0; i < N; i++) std::printf("%s[%.9g, %.9g, %d]", i ? ", " : "", F.mvKeysUn[i].pt.x, F.mvKeysUn[i].pt.y, F.mvKeysUn[i].octave);
    std::printf("], ");
    dump_floats("uRight", F.mvuRight.data(), N);
    std::printf(", ");
    dump_bytes("desc", F.mDescriptors.ptr<uint8_t>(0), (size_t)N * 32);
    std::printf(", \"held_before\": [");
    for (int i = 0; i < N; i++) std::printf("%s%ld", i ? ", " : "", F.mvpMapPoints[i] ? (long)F.mvpMapPoints[i]->mnId : -1L);
    std::printf("]},\n \"points\": [");
    bool first = true;
    for (auto& up : A.points) {
      MapPoint* p = up.get();
      std::printf("%s\n  {\"id\": %lu, \"local\": %d, \"bad\": %d, \"nobs\": %d, \"visible\": %d, \"mind\": %.9g, \"maxd\": %.9g, ", first ? "" : ",", p->mnId, (int)(p->mnId < 50000), (int)p->isBad(), p->nObs,
                  p->mnVisible, p->mfMinDistance, p->mfMaxDistance);
      dump_floats("pos", p->mWorldPos.ptr<float>(0), 3); std::printf(", "); dump_floats("normal", p->mNormalVector.ptr<float>(0), 3); std::printf(", ");
      dump_bytes("desc", p->mDescriptor.ptr<uint8_t>(0), 32);
      std::printf("}");
      first = false;
    }
    std::printf("],\n");
    // ---- the glue
    const int n = od::SearchLocalPoints<OracleOps>(F, local, th, far_pts, th_far);
    std::printf(" \"result\": {\"matches\": %d, \"held_after\": [", n);
    for (int i = 0; i < N; i++) std::printf("%s%ld", i ? ", " : "", F.mvpMapPoints[i] ? (long)F.mvpMapPoints[i]->mnId : -1L);
    std::printf("], \"points_after\": [");
    first = true;
    for (auto& up : A.points) {
      MapPoint* p = up.get();
      std::printf("%s[%lu, %ld, %d, %d]", first ? "" : ", ", p->mnId, p->mnLastFrameSeen == ~0ul ? -1L : (long)p->mnLastFrameSeen, p->mnVisible, (int)p->mbTrackInView);
      first = false;
    }
    std::printf("]}}\n");
  }
  return 0;
}
