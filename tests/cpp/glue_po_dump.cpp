// Host only: Frames of mock objects go through include/orbgpu_dropin.hpp's PoseOptimization with an entry-point set that RECORDS the
// flattened pose_opt_problem and answers with a synthetic result (every third correspondence an outlier, the pose shifted); the scene,
// the recorded problem and what the glue wrote back into the Frame are printed as JSON for tests/test_reference_formulas.py, which
// runs the reference's own edge collection (S/Optimizer.cc:964-1161, transliterated) on Python stand-ins of the same Frames.
#include <cstdio>
#include <cstring>
#include <vector>

#include "scenario.hpp"

namespace od = orbgpu::dropin;

struct Cap { std::vector<float> Xw, u, v, ur, w; float cam[5]; float Tcw[16]; bool has_rig, has_right; } g_cap;

struct RecordOps {
  static constexpr bool kUsesResidentFrame = false;
  static int pose_opt(const pose_opt_problem& p, pose_opt_result& r) {
    g_cap.Xw.assign(p.Xw, p.Xw + 3 * (size_t)p.n); g_cap.u.assign(p.u, p.u + p.n); g_cap.v.assign(p.v, p.v + p.n); g_cap.ur.assign(p.ur, p.ur + p.n);
    g_cap.w.assign(p.inv_sigma2, p.inv_sigma2 + p.n);
    g_cap.cam[0] = p.fx; g_cap.cam[1] = p.fy; g_cap.cam[2] = p.cx; g_cap.cam[3] = p.cy; g_cap.cam[4] = p.bf;
    std::memcpy(g_cap.Tcw, p.Tcw, 64); g_cap.has_rig = p.rig != nullptr; g_cap.has_right = p.rig && p.rig->has_right;
    int nbad = 0;
    for (int k = 0; k < p.n; k++) { r.outlier[k] = (k % 3) == 0; nbad += r.outlier[k]; }
    std::memcpy(r.Tcw, p.Tcw, 64); r.Tcw[3] += 0.25f;
    r.n_bad = nbad; r.n_inliers = p.n - nbad;
    return ORBG_OK;
  }
};

static void dump_floats(const char* name, const float* v, size_t n) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < n; i++) std::printf("%s%.9g", i ? ", " : "", v[i]);
  std::printf("]");
}
static void dump_keys(const char* name, const std::vector<KeyPoint>& k) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < k.size(); i++) std::printf("%s[%.9g, %.9g, %d]", i ? ", " : "", k[i].pt.x, k[i].pt.y, k[i].octave);
  std::printf("]");
}

int main() {
  for (int scene = 0; scene < 3; scene++) {
    Agent A;
    Frame* F;
    if (scene == 2) {
      F = build_rig_frame(A, 180, 140, 0.1, 99);
      F->fx = KB8_L[0]; F->fy = KB8_L[1]; F->cx = KB8_L[2]; F->cy = KB8_L[3]; F->mbf = 0.f;
    } else {
      g_seed = 500u + scene;
      std::unique_ptr<Frame> Fp(new Frame); F = Fp.get();
      const int N = scene == 1 ? 2 : 400;                           // (scene 1: fewer than three correspondences: returns 0, nothing is written)
      F->N = N; F->fx = FX; F->fy = FX; F->cx = CX; F->cy = CY; F->mbf = BF;
      float isig[8]; { float s = 1.f; for (int l = 0; l < 8; l++) { isig[l] = 1.f / (s * s); s *= 1.2f; } }
      F->mvInvLevelSigma2.assign(isig, isig + 8);
      F->mvKeysUn.resize(N); F->mvuRight.resize(N); F->mvpMapPoints.assign(N, nullptr); F->mvbOutlier.assign(N, true);
      for (int i = 0; i < N; i++) {
        F->mvKeysUn[i] = KeyPoint{{(float)(640 * urand()), (float)(480 * urand())}, 31.f, 0.f, 20.f, (int)(rnd() % 8)};
        F->mvuRight[i] = urand() < 0.3 ? -1.f : F->mvKeysUn[i].pt.x - (float)(40 * urand());
        if (i % 5 == 2) continue;
        std::unique_ptr<MapPoint> p(new MapPoint);
        p->mnId = 100 + i; p->mpMap = &A.map;
        for (int a = 0; a < 3; a++) p->mWorldPos.ptr<float>(0)[a] = (float)(10 * urand() - 5);
        F->mvpMapPoints[i] = p.get();
        A.points.push_back(std::move(p));
      }
      F->mvKeys = F->mvKeysUn;
      double T[16]; pose_of(5, T); F->mTcw = mat44(T);
      A.frames.push_back(std::move(Fp));
    }
    std::printf("{\"scene\": %d, \"N\": %d, \"Nleft\": %d, \"camera2\": %d, \"fx\": %.9g, \"fy\": %.9g, \"cx\": %.9g, \"cy\": %.9g, \"mbf\": %.9g, ", scene, F->N, F->Nleft, F->mpCamera2 ? 1 : 0,
                F->fx, F->fy, F->cx, F->cy, F->mbf);
    dump_floats("Tcw", F->mTcw.ptr<float>(0), 16); std::printf(", ");
    dump_keys("keysUn", F->mvKeysUn); std::printf(", "); dump_keys("keys", F->mvKeys); std::printf(", "); dump_keys("keysRight", F->mvKeysRight); std::printf(", ");
    dump_floats("uRight", F->mvuRight.data(), F->mvuRight.size()); std::printf(", ");
    dump_floats("invSigma2", F->mvInvLevelSigma2.data(), F->mvInvLevelSigma2.size());
    std::printf(", \"points\": [");
    for (int i = 0; i < F->N; i++) {
      if (F->mvpMapPoints[i]) { const float* X = F->mvpMapPoints[i]->mWorldPos.ptr<float>(0); std::printf("%s[%.9g, %.9g, %.9g]", i ? ", " : "", X[0], X[1], X[2]); }
      else std::printf("%snull", i ? ", " : "");
    }
    std::printf("], \"outlier_before\": [");
    for (int i = 0; i < F->N; i++) std::printf("%s%d", i ? ", " : "", (int)F->mvbOutlier[i]);
    g_cap = Cap();
    const int ret = od::PoseOptimization<RecordOps>(F);
    std::printf("], \"ret\": %d, \"outlier_after\": [", ret);
    for (int i = 0; i < F->N; i++) std::printf("%s%d", i ? ", " : "", (int)F->mvbOutlier[i]);
    std::printf("], "); dump_floats("Tcw_after", F->mTcw.ptr<float>(0), 16);
    std::printf(", \"problem\": {"); dump_floats("Xw", g_cap.Xw.data(), g_cap.Xw.size()); std::printf(", "); dump_floats("u", g_cap.u.data(), g_cap.u.size()); std::printf(", ");
    dump_floats("v", g_cap.v.data(), g_cap.v.size()); std::printf(", "); dump_floats("ur", g_cap.ur.data(), g_cap.ur.size()); std::printf(", ");
    dump_floats("w", g_cap.w.data(), g_cap.w.size()); std::printf(", "); dump_floats("cam", g_cap.cam, 5); std::printf(", "); dump_floats("Tcw", g_cap.Tcw, 16);
    std::printf(", \"has_rig\": %d, \"has_right\": %d}}\n", (int)g_cap.has_rig, (int)g_cap.has_right);
  }
  return 0;
}
