// The HAVE_OPENCV branches of include/orbgpu_adapters.hpp (ORBextractor::operator() with the reference's exact signature,
// I/ORBextractor.h:61-63) and include/orbgpu_dropin.hpp (mat_f32 / mat_u8 / make_mat over cv::Mat) compiled and run against the
// signature-only stub in tests/cpp/opencv_stub (this image has no OpenCV).  Exit code 3 + "no usable HIP device" without a GPU.
#define HAVE_OPENCV 1
#include <cstdio>
#include <cstring>
#include <vector>

#include "orbgpu_dropin.hpp"

int main() {
  try {
    const int W = 320, H = 240;
    cv::Mat im(H, W, CV_8UC1);
    unsigned s = 12345u;
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) { s = s * 1664525u + 1013904223u; im.ptr<uint8_t>(y)[x] = (uint8_t)(((x / 16 + y / 16) & 1) ? 40 + (s >> 28) : 200 - (s >> 28)); }
    orbgpu::ORBextractor ex(500, 1.2f, 8, 20, 7, W, H);
    std::vector<cv::KeyPoint> kps; cv::Mat desc; std::vector<int> lap{0, 0};
    const int mono = ex(im, cv::Mat(), kps, desc, lap);                 // the reference's call (S/Frame.cc:396-399)
    std::vector<orbx_keypoint> k2; std::vector<uint8_t> d2;
    const int mono2 = ex(im.data, im.cols, im.rows, (int)im.step, k2, d2, lap);
    bool same = mono == mono2 && kps.size() == k2.size() && desc.rows == (int)k2.size() && desc.cols == 32 && !kps.empty();
    for (size_t i = 0; same && i < k2.size(); i++)
      same = kps[i].pt.x == k2[i].x && kps[i].pt.y == k2[i].y && kps[i].angle == k2[i].angle && kps[i].octave == k2[i].octave &&
             std::memcmp(desc.ptr<uint8_t>((int)i), &d2[32 * i], 32) == 0;
    std::vector<cv::KeyPoint> none; cv::Mat nodesc;
    const int empty_rc = ex(cv::Mat(), cv::Mat(), none, nodesc, lap);   // empty image: -1 (S/ORBextractor.cc:1072-1073)
    const float T[16] = {1, 0, 0, 1.5f, 0, 1, 0, -2.f, 0, 0, 1, 0.25f, 0, 0, 0, 1};
    cv::Mat M; orbgpu::dropin::make_mat(M, 4, 4, T);
    const bool mats = M.rows == 4 && M.cols == 4 && std::memcmp(orbgpu::dropin::mat_f32(M), T, 64) == 0 && M.data != (const uint8_t*)T &&
                      orbgpu::dropin::mat_u8(desc, 1) == desc.ptr<uint8_t>(1);
    std::printf("opencv branches: %zu keypoints, monoIndex %d, identical to the pointer overload: %d, empty image -> %d, mat helpers: %d\n", kps.size(), mono,
                (int)same, empty_rc, (int)mats);
    return same && empty_rc == -1 && mats ? 0 : 1;
  } catch (const std::runtime_error& e) {
    std::printf("runtime_error: %s\n", e.what());
    return 3;
  }
}
