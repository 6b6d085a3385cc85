// Compile / link / run check of the C++ adapters (include/orbgpu_adapters.hpp) against liborbgpu.so.
// Without a GPU the adapters must fail loudly (std::runtime_error from ORBG_NO_DEVICE): exit code 3.
// With a GPU: extract a synthetic image, check determinism and the getters: exit code 0.
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ORBGPU_DROPIN_NAMESPACE
#include "orbgpu_adapters.hpp"

int main() {
  const int W = 320, H = 240;
  std::vector<uint8_t> img((size_t)W * H);
  unsigned s = 12345;
  for (auto& p : img) { s = s * 1664525u + 1013904223u; p = (uint8_t)(s >> 24); }
  for (int y = 60; y < 180; y++)
    for (int x = 80; x < 240; x++) img[(size_t)y * W + x] = (uint8_t)(((x / 16 + y / 16) & 1) ? 230 : 25);
  try {
    ORB_SLAM3::ORBextractor ex(500, 1.2f, 8, 20, 7, W, H);
    std::vector<orbx_keypoint> k1, k2;
    std::vector<uint8_t> d1, d2;
    std::vector<int> lap = {0, 0};
    const int m1 = ex(img.data(), W, H, W, k1, d1, lap);
    const int m2 = ex(img.data(), W, H, W, k2, d2, lap);
    if (m1 != (int)k1.size() || k1.size() < 50 || k1.size() != k2.size() || d1 != d2) { std::printf("extract mismatch\n"); return 1; }
    if (ex.GetLevels() != 8 || ex.GetScaleFactors().size() != 8 || ex.GetScaleFactors()[1] != 1.2f) { std::printf("getters\n"); return 1; }
    std::vector<uint8_t> none;
    std::vector<orbx_keypoint> k0;
    if (ex(nullptr, 0, 0, 0, k0, none, lap) != -1) { std::printf("empty image must return -1\n"); return 1; }
    std::printf("adapter smoke ok: %zu keypoints\n", k1.size());
    return 0;
  } catch (const std::runtime_error& e) {
    std::printf("runtime_error: %s\n", e.what());
    return 3;
  }
}
