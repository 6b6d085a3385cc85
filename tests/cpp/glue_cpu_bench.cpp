// Host-only timing of the reference-side glue around Optimizer::LocalBundleAdjustment (include/orbgpu_dropin.hpp): graph collection,
// flattening and write-back over the mock objects with an entry-point set that returns at once (the solve itself is timed on the GPU
// by tests/cpp/dropin_bench).  Eight consecutive windows, each for a new keyframe that sees a third of the last one's points; with
// the window cache (the mocks carry MapPoint::mnChangeStamp) and without.  Runs without a GPU:
//   g++ -O2 -std=c++17 -I include -I tests/cpp tests/cpp/glue_cpu_bench.cpp -L multi_orbslam3_amd -lorbgpu
#include <chrono>
#include <cstdio>
#include <cstring>

#include "scenario.hpp"

namespace od = orbgpu::dropin;

template <bool NoCache>
struct NullOps {
  static constexpr bool kUsesResidentFrame = false;
  static constexpr bool kNoLbaCache = NoCache;
  static int lba(const lba_problem& p, const volatile bool*, lba_result& r) {
    std::memcpy(r.poses, p.poses, sizeof(float) * 16 * (size_t)p.n_poses);
    std::memcpy(r.points, p.points, sizeof(float) * 3 * (size_t)p.n_points);
    for (int k = 0; k < p.n_edges; k++) { r.edge_outlier[k] = (k % 133) == 0; r.edge_depth_pos[k] = 1; r.edge_chi2[k] = 1.0; }
    r.status = LBA_APPLIED;
    return ORBG_OK;
  }
};

template <class Ops>
static double run(const char* what) {
  od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
  Agent B;
  KeyFrame* cur = build_lba_scene(B, 21, 10, 2000, 0.03, 99);
  bool stop = false; int nf = 0;
  double sum = 0, best = 1e30; int n = 0;
  for (int w = 0; w < 8; w++) {
    const auto t0 = std::chrono::steady_clock::now();
    od::LocalBundleAdjustment<Ops>(cur, &stop, &B.map, nf, 0);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (w >= 2) { sum += us; n++; best = std::min(best, us); }
    cur = next_keyframe(B, cur, w);
  }
  std::printf("LocalBundleAdjustment glue alone, %s (windows 3-8 of 8; 20 + 10 keyframes, 2000 points): mean %.1f us, best %.1f us\n", what, sum / n, best);
  return sum / n;
}

int main() {
  const double a = run<NullOps<false>>("window cache");
  const double b = run<NullOps<true>>("every point re-read");
  std::printf("{\"lba_glue_cached_us\": %.1f, \"lba_glue_uncached_us\": %.1f}\n", a, b);
  return 0;
}
