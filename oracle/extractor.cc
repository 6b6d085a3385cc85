// ORACLE (test infrastructure, not product code) -- see orb_oracle.h for status: parity unpinned.
//
// CPU restatement of ORB_SLAM3::ORBextractor (S/ORBextractor.cc) and of the OpenCV 3.2-era image
// primitives it calls.  Every function cites the reference lines it follows.
// Compile with -ffp-contract=off: float expressions must round exactly like an SSE2 build of the
// reference.

#include "orb_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr int kPatch = 31;       // PATCH_SIZE       S/ORBextractor.cc:70
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE  S/ORBextractor.cc:71
constexpr int kEdge = 19;        // EDGE_THRESHOLD   S/ORBextractor.cc:72

const int8_t kPattern[1024] = {
#include "orb_pattern_data.inc"
};

// cvRound: round-half-to-even (SSE2 cvtsd2si / lrint under the default rounding mode), Appendix A-6.
inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_round_f(float v) { return (int)std::nearbyintf(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

inline int reflect101(int p, int n) {
  // BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba  (cv::borderInterpolate)
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    else p = 2 * n - 2 - p;
  }
  return p;
}

}  // namespace

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 -- Appendix A-1; called at
// S/ORBextractor.cc:1165.  OpenCV: imgproc/src/imgwarp.cpp resizeGeneric_/HResizeLinear/VResizeLinear
// with INTER_RESIZE_COEF_BITS = 11.
extern "C" void oracle_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst,
                                     int dw, int dh, int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> alpha(2 * dw), beta(2 * dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    alpha[2 * dx] = sat_short(cv_round_f((1.f - fx) * 2048));
    alpha[2 * dx + 1] = sat_short(cv_round_f(fx * 2048));
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    beta[2 * dy] = sat_short(cv_round_f((1.f - fy) * 2048));
    beta[2 * dy + 1] = sat_short(cv_round_f(fy * 2048));
  }
  std::vector<int> row0(dw), row1(dw);
  auto hresize = [&](int sy, std::vector<int>& out) {
    const uint8_t* S = src + (size_t)sy * sstride;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      int sx1 = sx + 1 < sw ? sx + 1 : sx;   // right tap never read past the row (its weight is 0 there)
      out[dx] = S[sx] * alpha[2 * dx] + S[sx1] * alpha[2 * dx + 1];
    }
  };
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    hresize(sy0, row0);
    hresize(sy1, row1);
    int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// cv::copyMakeBorder(src, dst, b,b,b,b, BORDER_REFLECT_101) -- S/ORBextractor.cc:1167-1173.
// dst points at the top-left of the (w+2b) x (h+2b) bordered buffer.
extern "C" void oracle_border_reflect101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst,
                                         int border, int dstride) {
  for (int y = -border; y < h + border; y++) {
    const uint8_t* S = src + (size_t)reflect101(y, h) * sstride;
    uint8_t* D = dst + (size_t)(y + border) * dstride;
    for (int x = -border; x < w + border; x++) D[x + border] = S[reflect101(x, w)];
  }
}

namespace {
// FAST-9/16 Bresenham circle, Appendix A-2 (OpenCV features2d/src/fast_score.cpp makeOffsets).
const int kCircleDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kCircleDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
}  // namespace

// cornerScore<16> (OpenCV fast_score.cpp): max over the 16 arcs of 9 contiguous circle pixels of
// min(v - c) (bright centre) and of min(c - v) (dark centre), minus 1.  Equals the largest threshold t
// for which the pixel passes the segment test (c < v - t  or  c > v + t on >= 9 contiguous pixels).
extern "C" int oracle_fast_score(const uint8_t* img, int stride, int x, int y) {
  const uint8_t* p = img + (size_t)y * stride + x;
  int v = p[0];
  int d[25];
  for (int k = 0; k < 25; k++) d[k] = v - p[kCircleDy[k & 15] * stride + kCircleDx[k & 15]];
  int best = -256;
  for (int s = 0; s < 16; s++) {
    int mn = d[s], mx = d[s];
    for (int k = 1; k < 9; k++) {
      mn = std::min(mn, d[s + k]);
      mx = std::max(mx, d[s + k]);
    }
    best = std::max(best, std::max(mn, -mx));
  }
  return best - 1;   // >= t  <=>  corner at threshold t (t >= 0); negative => not a corner even at t = 0
}

// cv::FAST(image, keypoints, threshold, nonmaxSuppression) -- FAST_t<16> in OpenCV
// features2d/src/fast.cpp.  Detection area = [3, w-3) x [3, h-3); score buffer holds 0 for
// non-corners and for everything outside the detection area; NMS keeps strict 8-neighbour maxima;
// output order row-major.  Returns the number of keypoints (writes at most cap).
extern "C" int oracle_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax,
                                  int32_t* xys, int cap) {
  threshold = std::min(std::max(threshold, 0), 255);
  if (w < 7 || h < 7) return 0;
  std::vector<int> score((size_t)w * h, 0);
  std::vector<uint8_t> is_corner((size_t)w * h, 0);
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      // quick rejection (same role as the threshold_tab pre-test in FAST_t): a 9-arc of the 16-circle
      // always contains one pixel of every opposite pair, so a pair with both pixels inside
      // [v-t, v+t] rules the pixel out.
      const uint8_t* p = img + (size_t)y * stride + x;
      const int v = p[0];
      auto inside = [&](int k) { int c = p[kCircleDy[k] * stride + kCircleDx[k]]; return c >= v - threshold && c <= v + threshold; };
      if ((inside(0) && inside(8)) || (inside(4) && inside(12))) continue;
      int s = oracle_fast_score(img, stride, x, y);
      if (s >= threshold) {
        is_corner[(size_t)y * w + x] = 1;
        score[(size_t)y * w + x] = s;   // (uchar)cornerScore, 0..254
      }
    }
  int n = 0;
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      if (!is_corner[(size_t)y * w + x]) continue;
      int s = score[(size_t)y * w + x];
      bool keep = true;
      if (nonmax) {
        for (int dy = -1; dy <= 1 && keep; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            if (!dx && !dy) continue;
            if (!(s > score[(size_t)(y + dy) * w + (x + dx)])) { keep = false; break; }
          }
      }
      if (keep) {
        if (n < cap) { xys[3 * n] = x; xys[3 * n + 1] = y; xys[3 * n + 2] = s; }
        n++;
      }
    }
  return n;
}

// cv::GaussianBlur(m, m, Size(7,7), 2, 2, BORDER_REFLECT_101) on CV_8UC1 -- Appendix A-4, OpenCV <= 3.4.1
// fixed-point separable path: Q8 taps [18,34,49,55,49,34,18] (sum 257), row pass in int32,
// column pass (sum + 32768) >> 16, saturated.  Called at S/ORBextractor.cc:1115.
// taps4 = outer-to-centre half of the kernel (orbx_config.gauss_taps): {18,34,49,55} or, for OpenCV >= 4.5, {18,34,48,56}
extern "C" void oracle_gaussian_blur7_taps(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, const int* taps4) {
  const int K[7] = {taps4[0], taps4[1], taps4[2], taps4[3], taps4[2], taps4[1], taps4[0]};
  std::vector<int> tmp((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src + (size_t)y * sstride;
    for (int x = 0; x < w; x++) {
      int acc = 0;
      for (int k = -3; k <= 3; k++) acc += K[k + 3] * S[reflect101(x + k, w)];
      tmp[(size_t)y * w + x] = acc;
    }
  }
  for (int y = 0; y < h; y++) {
    uint8_t* D = dst + (size_t)y * dstride;
    for (int x = 0; x < w; x++) {
      int acc = 0;
      for (int k = -3; k <= 3; k++) acc += K[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
      D[x] = sat_u8((acc + 32768) >> 16);
    }
  }
}

extern "C" void oracle_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  static const int K4[4] = {18, 34, 49, 55};
  oracle_gaussian_blur7_taps(src, w, h, sstride, dst, dstride, K4);
}

// cv::fastAtan2(y, x) in degrees -- Appendix A-5 (OpenCV 3.x core/src/mathfuncs_core.cpp, scalar path).
extern "C" float oracle_fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ------------------------------------------------------------------------------------------------
// Extractor

struct OracleCand { int x, y, score; };   // relative to (minBorderX, minBorderY)

struct oracle_extractor {
  orbx_config cfg;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> feats_per_level;
  int umax[kHalfPatch + 1];
  // pyramid: bordered buffers, level image starts at (kEdge,kEdge)
  std::vector<std::vector<uint8_t>> pyr;
  std::vector<int> lw, lh, lstride;
  std::vector<std::vector<OracleCand>> cands;   // per level, vToDistributeKeys
  const uint8_t* level_ptr(int l) const { return pyr[l].data() + (size_t)kEdge * lstride[l] + kEdge; }
};

// ORBextractor::ORBextractor -- S/ORBextractor.cc:408-468.
extern "C" int oracle_extractor_create(const orbx_config* cfg, oracle_extractor** out) {
  if (!cfg || !out || cfg->n_levels < 1 || cfg->n_levels > ORBG_MAX_LEVELS || cfg->n_features < 1) return ORBG_BAD_ARG;
  auto* e = new oracle_extractor();
  e->cfg = *cfg;
  if (e->cfg.gauss_taps[0] == 0 && e->cfg.gauss_taps[1] == 0 && e->cfg.gauss_taps[2] == 0 && e->cfg.gauss_taps[3] == 0) {
    e->cfg.gauss_taps[0] = 18; e->cfg.gauss_taps[1] = 34; e->cfg.gauss_taps[2] = 49; e->cfg.gauss_taps[3] = 55;
  }
  const int nl = cfg->n_levels;
  e->scale.resize(nl); e->inv_scale.resize(nl); e->sigma2.resize(nl); e->inv_sigma2.resize(nl);
  e->scale[0] = 1.0f; e->sigma2[0] = 1.0f;
  for (int i = 1; i < nl; i++) {                          // :417-421
    e->scale[i] = e->scale[i - 1] * cfg->scale_factor;
    e->sigma2[i] = e->scale[i] * e->scale[i];
  }
  for (int i = 0; i < nl; i++) {                          // :425-429
    e->inv_scale[i] = 1.0f / e->scale[i];
    e->inv_sigma2[i] = 1.0f / e->sigma2[i];
  }
  e->feats_per_level.resize(nl);
  float factor = 1.0f / cfg->scale_factor;                // :434-444
  float desired = cfg->n_features * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    e->feats_per_level[l] = cv_round(desired);
    sum += e->feats_per_level[l];
    desired *= factor;
  }
  e->feats_per_level[nl - 1] = std::max(cfg->n_features - sum, 0);
  // umax: :452-467
  int v, v0, vmax = cv_floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
  int vmin = cv_ceil(kHalfPatch * std::sqrt(2.f) / 2);
  const double hp2 = kHalfPatch * kHalfPatch;
  for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round(std::sqrt(hp2 - v * v));
  for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
    e->umax[v] = v0;
    ++v0;
  }
  e->pyr.resize(nl); e->lw.resize(nl); e->lh.resize(nl); e->lstride.resize(nl); e->cands.resize(nl);
  *out = e;
  return ORBG_OK;
}

extern "C" int oracle_extractor_destroy(oracle_extractor* e) { delete e; return ORBG_OK; }

extern "C" int oracle_get_tables(const oracle_extractor* e, float* scale, float* inv_scale, float* sigma2,
                                 float* inv_sigma2, int32_t* fpl) {
  if (!e) return ORBG_BAD_ARG;
  for (int i = 0; i < e->cfg.n_levels; i++) {
    if (scale) scale[i] = e->scale[i];
    if (inv_scale) inv_scale[i] = e->inv_scale[i];
    if (sigma2) sigma2[i] = e->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
    if (fpl) fpl[i] = e->feats_per_level[i];
  }
  return ORBG_OK;
}

extern "C" int oracle_get_umax(const oracle_extractor* e, int32_t* umax16) {
  for (int i = 0; i <= kHalfPatch; i++) umax16[i] = e->umax[i];
  return ORBG_OK;
}

// ORBextractor::ComputePyramid -- S/ORBextractor.cc:1152-1177: level l resized from level l-1
// (cascade), then a 19-px REFLECT_101 border around every level.
static void compute_pyramid(oracle_extractor* e, const uint8_t* img, int w, int h, int stride) {
  for (int l = 0; l < e->cfg.n_levels; l++) {
    float s = e->inv_scale[l];
    int lw = cv_round((float)w * s), lh = cv_round((float)h * s);   // :1157
    e->lw[l] = lw; e->lh[l] = lh; e->lstride[l] = lw + 2 * kEdge;
    e->pyr[l].assign((size_t)(lw + 2 * kEdge) * (lh + 2 * kEdge), 0);
    uint8_t* interior = e->pyr[l].data() + (size_t)kEdge * e->lstride[l] + kEdge;
    if (l == 0) {
      for (int y = 0; y < lh; y++) std::memcpy(interior + (size_t)y * e->lstride[l], img + (size_t)y * stride, lw);
    } else {
      oracle_resize_linear(e->level_ptr(l - 1), e->lw[l - 1], e->lh[l - 1], e->lstride[l - 1], interior, lw, lh,
                           e->lstride[l]);
    }
    // border from the level's own interior (BORDER_ISOLATED); done via a temporary copy of the interior
    std::vector<uint8_t> tmp((size_t)lw * lh);
    for (int y = 0; y < lh; y++) std::memcpy(tmp.data() + (size_t)y * lw, interior + (size_t)y * e->lstride[l], lw);
    oracle_border_reflect101(tmp.data(), lw, lh, lw, e->pyr[l].data(), kEdge, e->lstride[l]);
  }
}

// ------------------------------------------------------------------------------------------------
// DistributeOctTree -- S/ORBextractor.cc:479-761.  The reference keeps nodes in a std::list (children
// pushed to the FRONT, parent erased) and, in the "close to N" phase, splits nodes in the order given by
// sort(pair<int size, ExtractorNode*>) walked from the back.  The pointer tie-break is heap-address
// dependent; this restatement pins it to (size, creation sequence), i.e. among
// equal sizes the most recently created node is split first.

namespace {

struct OctNode {
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  std::vector<int> keys;    // indices into the candidate array, in arrival order
  bool no_more = false;
  int prev = -1, next = -1; // list links
  int seq = 0;
};

struct OctList {
  std::vector<OctNode> pool;
  int head = -1, tail = -1, size = 0, seq = 0;
  int push_front(OctNode&& n) {
    n.seq = seq++;
    pool.push_back(std::move(n));
    int id = (int)pool.size() - 1;
    pool[id].prev = -1; pool[id].next = head;
    if (head >= 0) pool[head].prev = id; else tail = id;
    head = id; size++;
    return id;
  }
  int push_back(OctNode&& n) {
    n.seq = seq++;
    pool.push_back(std::move(n));
    int id = (int)pool.size() - 1;
    pool[id].next = -1; pool[id].prev = tail;
    if (tail >= 0) pool[tail].next = id; else head = id;
    tail = id; size++;
    return id;
  }
  int erase(int id) {   // returns next
    int p = pool[id].prev, n = pool[id].next;
    if (p >= 0) pool[p].next = n; else head = n;
    if (n >= 0) pool[n].prev = p; else tail = p;
    size--;
    return n;
  }
};

// ExtractorNode::DivideNode -- S/ORBextractor.cc:479-535
void divide_node(const OctNode& p, const std::vector<OracleCand>& c, OctNode ch[4]) {
  const int halfX = (int)std::ceil(static_cast<float>(p.URx - p.ULx) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(p.BRy - p.ULy) / 2);
  OctNode &n1 = ch[0], &n2 = ch[1], &n3 = ch[2], &n4 = ch[3];
  n1.ULx = p.ULx; n1.ULy = p.ULy; n1.URx = p.ULx + halfX; n1.URy = p.ULy;
  n1.BLx = p.ULx; n1.BLy = p.ULy + halfY; n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
  n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = p.URx; n2.URy = p.URy;
  n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
  n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy;
  n3.BLx = p.BLx; n3.BLy = p.BLy; n3.BRx = n1.BRx; n3.BRy = p.BLy;
  n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy;
  n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = p.BRx; n4.BRy = p.BRy;
  for (int k : p.keys) {
    const float kx = (float)c[k].x, ky = (float)c[k].y;
    if (kx < n1.URx) {
      if (ky < n1.BRy) n1.keys.push_back(k); else n3.keys.push_back(k);
    } else if (ky < n1.BRy) n2.keys.push_back(k);
    else n4.keys.push_back(k);
  }
  for (int i = 0; i < 4; i++) if (ch[i].keys.size() == 1) ch[i].no_more = true;
}

std::vector<int> distribute_octree(const std::vector<OracleCand>& c, int minX, int maxX, int minY, int maxY, int N, bool oldest_first = false) {
  std::vector<int> result;
  if (c.empty()) return result;
  int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));   // :541
  if (nIni < 1) nIni = 1;   // reference divides by zero for portrait images; pinned to 1 root
  const float hX = static_cast<float>(maxX - minX) / nIni;
  OctList L;
  L.pool.reserve(c.size() * 4 + 16);
  std::vector<int> ini(nIni);
  for (int i = 0; i < nIni; i++) {                                                // :550-561
    OctNode n;
    n.ULx = (int)(hX * static_cast<float>(i)); n.ULy = 0;
    n.URx = (int)(hX * static_cast<float>(i + 1)); n.URy = 0;
    n.BLx = n.ULx; n.BLy = maxY - minY;
    n.BRx = n.URx; n.BRy = maxY - minY;
    ini[i] = L.push_back(std::move(n));
  }
  for (size_t i = 0; i < c.size(); i++) {                                         // :564-568
    int r = (int)((float)c[i].x / hX);
    if (r >= nIni) r = nIni - 1;
    L.pool[ini[r]].keys.push_back((int)i);
  }
  for (int it = L.head; it >= 0;) {                                               // :572-583
    if (L.pool[it].keys.size() == 1) { L.pool[it].no_more = true; it = L.pool[it].next; }
    else if (L.pool[it].keys.empty()) it = L.erase(it);
    else it = L.pool[it].next;
  }
  bool finish = false;
  std::vector<std::pair<int, int>> size_and_node;   // (size, node id); node seq == creation order
  auto add_children = [&](OctNode ch[4], int* nToExpand) {
    for (int i = 0; i < 4; i++) {
      if (ch[i].keys.empty()) continue;
      int sz = (int)ch[i].keys.size();
      int id = L.push_front(std::move(ch[i]));
      if (sz > 1) {
        if (nToExpand) (*nToExpand)++;
        size_and_node.push_back({sz, id});
      }
    }
  };
  while (!finish) {                                                               // :592-737
    int prevSize = L.size;
    int nToExpand = 0;
    size_and_node.clear();
    for (int it = L.head; it >= 0;) {
      if (L.pool[it].no_more) { it = L.pool[it].next; continue; }
      OctNode ch[4];
      divide_node(L.pool[it], c, ch);
      add_children(ch, &nToExpand);
      it = L.erase(it);
    }
    if (L.size >= N || L.size == prevSize) {
      finish = true;
    } else if (L.size + nToExpand * 3 > N) {
      while (!finish) {
        prevSize = L.size;
        std::vector<std::pair<int, int>> prev = size_and_node;
        size_and_node.clear();
        // sort(pair<int, ExtractorNode*>): pinned tie-break = creation sequence (== pool id order)
        std::sort(prev.begin(), prev.end(), [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {
          if (a.first != b.first) return a.first < b.first;
          return oldest_first ? L.pool[a.second].seq > L.pool[b.second].seq : L.pool[a.second].seq < L.pool[b.second].seq;
        });
        for (int j = (int)prev.size() - 1; j >= 0; j--) {
          OctNode ch[4];
          divide_node(L.pool[prev[j].second], c, ch);
          add_children(ch, nullptr);
          L.erase(prev[j].second);
          if (L.size >= N) break;
        }
        if (L.size >= N || L.size == prevSize) finish = true;
      }
    }
  }
  result.reserve(L.size);
  for (int it = L.head; it >= 0; it = L.pool[it].next) {                          // :742-758
    const std::vector<int>& ks = L.pool[it].keys;
    int best = ks[0];
    for (size_t k = 1; k < ks.size(); k++)
      if (c[ks[k]].score > c[best].score) best = ks[k];
    result.push_back(best);
  }
  return result;
}

}  // namespace

extern "C" int oracle_distribute_octree(const int32_t* xys, int n, int min_x, int max_x, int min_y, int max_y,
                                        int n_target, int32_t* out_xys, int cap) {
  std::vector<OracleCand> c(n);
  for (int i = 0; i < n; i++) c[i] = {xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]};
  std::vector<int> keep = distribute_octree(c, min_x, max_x, min_y, max_y, n_target);
  for (size_t i = 0; i < keep.size() && (int)i < cap; i++) {
    out_xys[3 * i] = c[keep[i]].x; out_xys[3 * i + 1] = c[keep[i]].y; out_xys[3 * i + 2] = c[keep[i]].score;
  }
  return (int)keep.size();
}

// The cells cv::FAST is run on, level by level: ComputeKeyPointsOctTree's tiling, S/ORBextractor.cc:771-804 (W = 30, +6 px of overlap,
// the float / int mix of the reference kept).  [x0, x1) x [y0, y1) in level coordinates; (i, j) = the cell's row and column.
struct FastCell { int x0, x1, y0, y1, i, j; };
static bool level_cells(int lw, int lh, std::vector<FastCell>& out, int* w_cell, int* h_cell) {
  const float W = 30;
  const int minBorderX = kEdge - 3, minBorderY = minBorderX;
  const int maxBorderX = lw - kEdge + 3;
  const int maxBorderY = lh - kEdge + 3;
  const float fw = (float)(maxBorderX - minBorderX), fh = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(fw / W), nRows = (int)(fh / W);
  if (nCols < 1 || nRows < 1) return false;
  const int wCell = (int)std::ceil(fw / nCols), hCell = (int)std::ceil(fh / nRows);
  *w_cell = wCell; *h_cell = hCell;
  for (int i = 0; i < nRows; i++) {
    const float iniY = (float)(minBorderY + i * hCell);
    float maxY = iniY + hCell + 6;
    if (iniY >= maxBorderY - 3) continue;
    if (maxY > maxBorderY) maxY = (float)maxBorderY;
    for (int j = 0; j < nCols; j++) {
      const float iniX = (float)(minBorderX + j * wCell);
      float maxX = iniX + wCell + 6;
      if (iniX >= maxBorderX - 6) continue;
      if (maxX > maxBorderX) maxX = (float)maxBorderX;
      out.push_back(FastCell{(int)iniX, (int)maxX, (int)iniY, (int)maxY, i, j});
    }
  }
  return true;
}

// IC_Angle -- S/ORBextractor.cc:75-102
static float ic_angle(const uint8_t* center, int step, const int* umax) {
  int m_01 = 0, m_10 = 0;
  for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m_10 += u * center[u];
  for (int v = 1; v <= kHalfPatch; ++v) {
    int v_sum = 0;
    int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return oracle_fast_atan2((float)m_01, (float)m_10);
}

// computeOrbDescriptor -- S/ORBextractor.cc:105-145
static void orb_descriptor(float kp_angle, const uint8_t* center, int step, uint8_t* desc) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float angle = kp_angle * factorPI;
  float a = (float)std::cos(angle), b = (float)std::sin(angle);   // cosf / sinf overloads
  const int8_t* pat = kPattern;
  auto get = [&](int idx) -> int {
    int px = pat[2 * idx], py = pat[2 * idx + 1];
    int r = cv_round((double)(px * b + py * a));
    int c = cv_round((double)(px * a - py * b));
    return center[r * step + c];
  };
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int k = 0; k < 8; k++) {
      int t0 = get(2 * k), t1 = get(2 * k + 1);
      val |= (t0 < t1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

// the FAST cells of a level, row by row (ComputeKeyPointsOctTree, S/ORBextractor.cc:771-804), for the known-answer tests
extern "C" int oracle_fast_cell_grid(int level_w, int level_h, int32_t* rects6, int cap, int32_t* w_cell, int32_t* h_cell) {
  std::vector<FastCell> cells;
  int wc = 0, hc = 0;
  if (!level_cells(level_w, level_h, cells, &wc, &hc)) return 0;
  if (w_cell) *w_cell = wc;
  if (h_cell) *h_cell = hc;
  for (size_t k = 0; k < cells.size() && (int)k < cap; k++) {
    const FastCell& c = cells[k];
    rects6[6 * k] = c.x0; rects6[6 * k + 1] = c.x1; rects6[6 * k + 2] = c.y0; rects6[6 * k + 3] = c.y1; rects6[6 * k + 4] = c.i; rects6[6 * k + 5] = c.j;
  }
  return (int)cells.size();
}

// IC_Angle of one keypoint on a caller-supplied level image, for the known-answer tests: the radius-15 disc must lie inside
extern "C" float oracle_ic_angle(const oracle_extractor* e, const uint8_t* img, int stride, int x, int y) {
  return ic_angle(img + (size_t)y * stride + x, stride, e->umax);
}

// one descriptor on a caller-supplied (already blurred) image, for the known-answer tests: the patch around (x, y) must lie inside
extern "C" void oracle_orb_descriptor(float kp_angle_deg, const uint8_t* img, int stride, int x, int y, uint8_t* desc32) {
  orb_descriptor(kp_angle_deg, img + (size_t)y * stride + x, stride, desc32);
}

// ORBextractor::operator() -- S/ORBextractor.cc:1068-1150 (ComputeKeyPointsOctTree :763-878 inlined).
extern "C" int oracle_extract(oracle_extractor* e, const uint8_t* img, int width, int height, int stride,
                              int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                              int* n_mono_out) {
  if (!e || !n_out) return ORBG_BAD_ARG;
  if (!img || width <= 0 || height <= 0) return ORBG_EMPTY;       // :1072-1073
  const int nl = e->cfg.n_levels;
  compute_pyramid(e, img, width, height, stride);

  struct LevelKp { float x, y, angle, response; };
  std::vector<std::vector<LevelKp>> all(nl);
  for (int level = 0; level < nl; ++level) {                       // :769-873
    const int minBorderX = kEdge - 3, minBorderY = minBorderX;
    const int maxBorderX = e->lw[level] - kEdge + 3;
    const int maxBorderY = e->lh[level] - kEdge + 3;
    std::vector<OracleCand>& cand = e->cands[level];
    cand.clear();
    int wCell = 0, hCell = 0;
    std::vector<FastCell> cells;
    if (!level_cells(e->lw[level], e->lh[level], cells, &wCell, &hCell)) continue;   // level too small for a single cell (reference would divide by 0)
    const uint8_t* base = e->level_ptr(level);
    const int ls = e->lstride[level];
    std::vector<int32_t> cell(3 * 4096);
    for (const FastCell& c : cells) {
      const int x0 = c.x0, y0 = c.y0, cw = c.x1 - x0, ch = c.y1 - y0;
      const uint8_t* sub = base + (size_t)y0 * ls + x0;
      int nk = oracle_fast_detect(sub, cw, ch, ls, e->cfg.ini_th_fast, 1, cell.data(), 4096);   // :808
      if (nk == 0) nk = oracle_fast_detect(sub, cw, ch, ls, e->cfg.min_th_fast, 1, cell.data(), 4096);   // :825-828
      for (int k = 0; k < nk; k++)                                                              // :845-850
        cand.push_back({cell[3 * k] + c.j * wCell, cell[3 * k + 1] + c.i * hCell, cell[3 * k + 2]});
    }
    std::vector<int> keep = distribute_octree(cand, minBorderX, maxBorderX, minBorderY, maxBorderY, e->feats_per_level[level],
                                              e->cfg.octree_oldest_first != 0);
    all[level].reserve(keep.size());
    for (int k : keep) {                                                                          // :862-872
      LevelKp kp;
      kp.x = (float)(cand[k].x + minBorderX);
      kp.y = (float)(cand[k].y + minBorderY);
      kp.response = (float)cand[k].score;
      kp.angle = 0;
      all[level].push_back(kp);
    }
  }
  for (int level = 0; level < nl; ++level)                                                        // :876-877
    for (auto& kp : all[level])
      kp.angle = ic_angle(e->level_ptr(level) + (size_t)cv_round(kp.y) * e->lstride[level] + cv_round(kp.x),
                          e->lstride[level], e->umax);

  int nkeypoints = 0;
  for (int level = 0; level < nl; ++level) nkeypoints += (int)all[level].size();
  *n_out = nkeypoints;
  if (nkeypoints > cap) return ORBG_CAP_EXCEEDED;
  int monoIndex = 0, stereoIndex = nkeypoints - 1;                                                // :1104
  for (int level = 0; level < nl; ++level) {
    if (all[level].empty()) continue;
    const int lw = e->lw[level], lh = e->lh[level];
    // workingMat = mvImagePyramid[level].clone(); GaussianBlur(...)  :1114-1115
    std::vector<uint8_t> work((size_t)lw * lh), blur((size_t)lw * lh);
    for (int y = 0; y < lh; y++)
      std::memcpy(work.data() + (size_t)y * lw, e->level_ptr(level) + (size_t)y * e->lstride[level], lw);
    oracle_gaussian_blur7_taps(work.data(), lw, lh, lw, blur.data(), lw, e->cfg.gauss_taps);
    const float scale = e->scale[level];
    const int scaledPatchSize = (int)(kPatch * e->scale[level]);                                  // :862
    for (auto& kp : all[level]) {
      uint8_t d[32];
      orb_descriptor(kp.angle, blur.data() + (size_t)cv_round(kp.y) * lw + cv_round(kp.x), lw, d);
      float px = kp.x, py = kp.y;
      if (level != 0) { px *= scale; py *= scale; }                                               // :1131-1133
      orbx_keypoint o;
      o.x = px; o.y = py; o.size = (float)scaledPatchSize; o.angle = kp.angle; o.response = kp.response; o.octave = level;
      int dst;
      if (px >= (float)lap0 && px <= (float)lap1) dst = stereoIndex--;                             // :1135-1144
      else dst = monoIndex++;
      kps[dst] = o;
      std::memcpy(desc + (size_t)dst * 32, d, 32);
    }
  }
  if (n_mono_out) *n_mono_out = monoIndex;
  return ORBG_OK;
}

extern "C" int oracle_get_level(oracle_extractor* e, int level, uint8_t* host_out, int* width, int* height) {
  if (!e || level < 0 || level >= e->cfg.n_levels || e->pyr[level].empty()) return ORBG_BAD_ARG;
  if (width) *width = e->lw[level];
  if (height) *height = e->lh[level];
  if (host_out)
    for (int y = 0; y < e->lh[level]; y++)
      std::memcpy(host_out + (size_t)y * e->lw[level], e->level_ptr(level) + (size_t)y * e->lstride[level], e->lw[level]);
  return ORBG_OK;
}

extern "C" int oracle_get_level_bordered(oracle_extractor* e, int level, uint8_t* host_out, int* width, int* height) {
  if (!e || level < 0 || level >= e->cfg.n_levels || e->pyr[level].empty()) return ORBG_BAD_ARG;
  if (width) *width = e->lw[level];
  if (height) *height = e->lh[level];
  const int bw = e->lw[level] + 2 * kEdge, bh = e->lh[level] + 2 * kEdge;
  if (host_out)
    for (int y = 0; y < bh; y++) std::memcpy(host_out + (size_t)y * bw, e->pyr[level].data() + (size_t)y * e->lstride[level], bw);
  return ORBG_OK;
}

extern "C" int oracle_get_candidates(oracle_extractor* e, int level, int32_t* xys, int cap, int* n) {
  if (!e || level < 0 || level >= e->cfg.n_levels || !n) return ORBG_BAD_ARG;
  *n = (int)e->cands[level].size();
  for (int i = 0; i < *n && i < cap; i++) {
    xys[3 * i] = e->cands[level][i].x; xys[3 * i + 1] = e->cands[level][i].y; xys[3 * i + 2] = e->cands[level][i].score;
  }
  return *n > cap ? ORBG_CAP_EXCEEDED : ORBG_OK;
}

// Accessors used by stereo.cc
extern "C++" {
const uint8_t* oracle_level_ptr(const oracle_extractor* e, int level, int* w, int* h, int* stride) {
  *w = e->lw[level]; *h = e->lh[level]; *stride = e->lstride[level];
  return e->level_ptr(level);
}
const std::vector<float>& oracle_scale(const oracle_extractor* e) { return e->scale; }
const std::vector<float>& oracle_inv_scale(const oracle_extractor* e) { return e->inv_scale; }
}
