// Host-only timing of the reference-side glue around Optimizer::LocalBundleAdjustment (include/orbgpu_dropin.hpp): graph collection,
// flattening and write-back over the mock objects with an entry-point set that returns at once (the solve itself is timed on the GPU
// by tests/cpp/dropin_bench).  Runs without a GPU:  g++ -O2 -I include -I tests/cpp tests/cpp/glue_cpu_bench.cpp -L multi_orbslam3_amd -lorbgpu
#include <chrono>
#include <cstdio>
#include <cstring>

#include "scenario.hpp"

namespace od = orbgpu::dropin;

struct NullOps {
  static constexpr bool kUsesResidentFrame = false;
  static int lba(const lba_problem& p, const volatile bool*, lba_result& r) {
    std::memcpy(r.poses, p.poses, sizeof(float) * 16 * (size_t)p.n_poses);
    std::memcpy(r.points, p.points, sizeof(float) * 3 * (size_t)p.n_points);
    for (int k = 0; k < p.n_edges; k++) { r.edge_outlier[k] = (k % 33) == 0; r.edge_depth_pos[k] = 1; r.edge_chi2[k] = 1.0; }
    r.status = LBA_APPLIED;
    return ORBG_OK;
  }
};

int main() {
  double best = 1e30, sum = 0;
  const int reps = 12;
  for (int i = 0; i < reps; i++) {
    Agent B;
    KeyFrame* cur = build_lba_scene(B, 21, 10, 2000, 0.03, 99);
    bool stop = false; int nf = 0;
    const auto t0 = std::chrono::steady_clock::now();
    od::LocalBundleAdjustment<NullOps>(cur, &stop, &B.map, nf, 0);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (i >= 2) { best = std::min(best, us); sum += us; }
  }
  std::printf("LocalBundleAdjustment glue alone (20 + 10 keyframes, 2000 points): mean %.1f us, best %.1f us\n", sum / (reps - 2), best);
  return 0;
}
