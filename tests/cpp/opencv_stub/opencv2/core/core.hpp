// SIGNATURE-ONLY stand-in for the handful of OpenCV declarations the HAVE_OPENCV branches of include/orbgpu_adapters.hpp and
// include/orbgpu_dropin.hpp use -- TEST INFRASTRUCTURE, never shipped, never part of the oracle or of any reference build.  OpenCV is
// not installed in this image; without this file those ~25 lines would never see a compiler.  It pins NOTHING about OpenCV's
// behaviour: cv::Mat here is a plain row-major byte buffer.  A deployment compiles the same lines against the real headers.
#pragma once
#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(expr) assert(expr)
namespace cv {
class Mat {
 public:
  int rows = 0, cols = 0; uint8_t* data = nullptr; size_t step = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* ext) : rows(r), cols(c), data((uint8_t*)ext), step((size_t)c * esz(type)), type_(type) {}   // no copy, as in OpenCV
  void create(int r, int c, int type) { rows = r; cols = c; type_ = type; step = (size_t)c * esz(type); own_.reset(new uint8_t[(size_t)r * step + 1]()); data = own_.get(); }
  void release() { own_.reset(); data = nullptr; rows = cols = 0; step = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return type_; }
  Mat clone() const { Mat m; if (!empty()) { m.create(rows, cols, type_); for (int r = 0; r < rows; r++) std::memcpy(m.data + r * m.step, data + r * step, m.step); } return m; }
  template <class T> T* ptr(int row = 0) { return reinterpret_cast<T*>(data + (size_t)row * step); }
  template <class T> const T* ptr(int row = 0) const { return reinterpret_cast<const T*>(data + (size_t)row * step); }
 private:
  static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
  std::shared_ptr<uint8_t[]> own_; int type_ = CV_8U;
};
struct Point2f { float x = 0, y = 0; };
struct KeyPoint {
  Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : size(s), angle(a), response(r), octave(o), class_id(c) { pt.x = x; pt.y = y; }
};
// _InputArray / _OutputArray reduced to "a Mat, by reference"
class _InputArray { public: _InputArray() {} _InputArray(const Mat& m) : m_(&m) {} bool empty() const { return !m_ || m_->empty(); } Mat getMat() const { return m_ ? *m_ : Mat(); } private: const Mat* m_ = nullptr; };
class _OutputArray { public: _OutputArray(Mat& m) : m_(&m) {} void create(int r, int c, int t) const { m_->create(r, c, t); } void release() const { m_->release(); } Mat& getMat() const { return *m_; } private: Mat* m_; };
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
