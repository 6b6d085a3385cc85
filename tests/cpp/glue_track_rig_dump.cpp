// Host only (the glue over the CPU oracle's entry points): a two-camera Frame (Nleft != -1) of the two-fisheye rig and a local map go
// through include/orbgpu_dropin.hpp's SearchLocalPoints; the scene before the call and everything the call left behind --
// F.mvpMapPoints, every point's mnLastFrameSeen / visible count / track flags, levels and fields of both cameras, the return value --
// are printed as JSON.  tests/test_reference_formulas.py rebuilds the scene as Python stand-ins and runs Tracking::SearchLocalPoints'
// own text (with Frame::isInFrustum / isInFrustumChecks, KannalaBrandt8::project and ORBmatcher::SearchByProjection, transliterated).
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
static void dump_keys(const char* name, const std::vector<KeyPoint>& k) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < k.size(); i++) std::printf("%s[%.9g, %.9g, %d]", i ? ", " : "", k[i].pt.x, k[i].pt.y, k[i].octave);
  std::printf("]");
}
static void dump_ints(const char* name, const std::vector<int>& v) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); i++) std::printf("%s%d", i ? ", " : "", v[i]);
  std::printf("]");
}

int main() {
  for (int scene = 0; scene < 3; scene++) {
    Agent A;
    RigTrack S = build_rig_track_scene(A, 900u + scene, scene == 1 ? 300 : 800, scene == 1 ? 60 : 200);
    Frame mono;                                                   // scene 2: the left camera alone -- a MONOCULAR fisheye Frame (Nleft == -1, mpCamera a KannalaBrandt8)
    if (scene == 2) {
      const Frame& C = *S.cur; const int nl = C.Nleft;
      mono = C; mono.Nleft = -1; mono.Nright = -1; mono.N = nl; mono.mpCamera2 = nullptr;
      mono.mvKeys.resize(nl); mono.mvKeysUn = mono.mvKeys; mono.mvKeysRight.clear(); mono.mvuRight.assign(nl, -1.f); mono.mvDepth.assign(nl, -1.f);
      mono.mvpMapPoints.resize(nl); mono.mvbOutlier.resize(nl);
      Mat d(nl, 32, 1); std::memcpy(d.ptr<uint8_t>(0), C.mDescriptors.ptr<uint8_t>(0), (size_t)nl * 32); mono.mDescriptors = d;
      mono.mvLeftToRightMatch.assign(nl, -1); mono.mvRightToLeftMatch.clear();
    }
    Frame& F = scene == 2 ? mono : *S.cur;
    const float th = scene == 1 ? 5.f : 1.f; const bool far_pts = scene == 1; const float th_far = 6.f;
    std::printf("{\"scene\": %d, \"frame\": {\"id\": %lu, \"N\": %d, \"Nleft\": %d, \"th\": %.9g, \"far\": %d, \"th_far\": %.9g, \"size\": %.9g, \"mb\": %.9g, ", scene, F.mnId, F.N, F.Nleft, th,
                (int)far_pts, th_far, F.mnMaxX, F.mb);
    dump_floats("Tcw", F.mTcw.ptr<float>(0), 16); std::printf(", "); dump_floats("Trl", F.mTrl.ptr<float>(0), 12); std::printf(", "); dump_floats("Tlr", F.mTlr.ptr<float>(0), 12);
    std::printf(", "); dump_floats("cam_left", KB8_L, 8); std::printf(", "); dump_floats("cam_right", KB8_R, 8); std::printf(", ");
    dump_keys("keys", F.mvKeys); std::printf(", "); dump_keys("keysRight", F.mvKeysRight); std::printf(", ");
    dump_ints("l2r", F.mvLeftToRightMatch); std::printf(", "); dump_ints("r2l", F.mvRightToLeftMatch); std::printf(", ");
    dump_bytes("desc", F.mDescriptors.ptr<uint8_t>(0), (size_t)F.N * 32);
    std::printf(", \"held_before\": [");
    for (int i = 0; i < F.N; i++) std::printf("%s%ld", i ? ", " : "", F.mvpMapPoints[i] ? (long)F.mvpMapPoints[i]->mnId : -1L);
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
    const int n = od::SearchLocalPoints<OracleOps>(F, S.local, th, far_pts, th_far);
    std::printf(" \"result\": {\"matches\": %d, \"held_after\": [", n);
    for (int i = 0; i < F.N; i++) std::printf("%s%ld", i ? ", " : "", F.mvpMapPoints[i] ? (long)F.mvpMapPoints[i]->mnId : -1L);
    std::printf("], \"points_after\": [");
    first = true;
    for (auto& up : A.points) {
      MapPoint* p = up.get();
      std::printf("%s[%lu, %ld, %d, %d, %d, %d, %d, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", first ? "" : ", ", p->mnId, p->mnLastFrameSeen == ~0ul ? -1L : (long)p->mnLastFrameSeen,
                  p->mnVisible, (int)p->mbTrackInView, (int)p->mbTrackInViewR, p->mnTrackScaleLevel, p->mnTrackScaleLevelR, p->mTrackProjX, p->mTrackProjY, p->mTrackDepth,
                  p->mTrackViewCos, p->mTrackProjXR, p->mTrackProjYR, p->mTrackDepthR, p->mTrackViewCosR);
      first = false;
    }
    std::printf("]}}\n");
  }
  return 0;
}
