// Stand-alone check + timing of csrc/ldlt_mfma.hpp (the FP64-MFMA LDL^T of the reduced camera system).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ldlt_mfma_test ldlt_mfma_test.hip && ./ldlt_mfma_test
// 1. probes the operand / accumulator lane layout of v_mfma_f64_16x16x4_f64 with asymmetric integer data,
// 2. solves random SPD systems of the sizes the local BA produces and compares with a long-double host LDL^T,
// 3. checks the zero-pivot failure flag, 4. times the kernel (HIP events over back-to-back launches).
#include <hip/hip_runtime.h>

#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#ifndef NO_PROFILE
#define LDLTM_PROFILE 1
#endif
#ifdef XPROFILE
#define LDLTX_PROFILE 1
#endif
#ifdef XWATCHDOG
#define LDLTX_WATCHDOG 1
#endif
#include "../../multi_orbslam3_amd/csrc/ldlt_mfma.hpp"
#include "../../multi_orbslam3_amd/csrc/ldlt_xcd.hpp"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void k_probe(const double* A /*16x4*/, const double* B /*4x16*/, double* D /*16x16*/) {
  const int l = threadIdx.x;
  ldltm::d4 c = {0, 0, 0, 0};
  c = ldltm::mfma(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c);
  for (int g = 0; g < 4; g++) D[((l >> 4) + 4 * g) * 16 + (l & 15)] = c[g];
}

static bool host_solve(int n, const std::vector<double>& S, const std::vector<double>& b, std::vector<double>& x) {
  std::vector<long double> a(S.begin(), S.end()), y(b.begin(), b.end());
  for (int j = 0; j < n; j++) {
    const long double d = a[(size_t)j * n + j];
    if (d == 0) return false;
    for (int i = j + 1; i < n; i++) {
      const long double l = a[(size_t)i * n + j] / d;
      for (int c = j + 1; c <= i; c++) a[(size_t)i * n + c] -= l * a[(size_t)c * n + j];   // rows c >= j+1 still unscaled below
    }
    for (int i = j + 1; i < n; i++) a[(size_t)i * n + j] /= d;
  }
  for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= a[(size_t)i * n + j] * y[j];
  for (int i = 0; i < n; i++) y[i] /= a[(size_t)i * n + i];
  for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= a[(size_t)j * n + i] * y[j];
  x.assign(n, 0.0);
  for (int i = 0; i < n; i++) x[i] = (double)y[i];
  return true;
}

int main(int argc, char** argv) {
  int fails = 0;
  {   // ---- 1. layout probe
    std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i * 7 + k * 3;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 + k * 11 + j * 5 + (j * j) % 3;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; i++) bad += D[i] != ref[i];
    printf("mfma_f64_16x16x4 layout probe: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    fails += bad != 0;
  }
  if (argc > 2 && !strcmp(argv[1], "stress")) {
    // ./ldlt_mfma_test stress N: N launches of the eight-workgroup kernel on two alternating systems (150 and 300 unknowns, other
    // XCD every 64 launches), every result compared bit for bit with the first one of its size
    const int N = atoi(argv[2]);
    std::mt19937_64 rs(777);
    std::normal_distribution<double> G01(0.0, 1.0);
    struct Sys { int n; double* dS; double* dx; int* dok; std::vector<double> first; } sys[2] = {{150}, {300}};
    for (auto& y : sys) {
      const int n = y.n;
      std::vector<double> M((size_t)n * n), S((size_t)n * n), b(n);
      for (auto& v : M) v = G01(rs);
      for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double a = 0; for (int k = 0; k < n; k++) a += M[(size_t)i * n + k] * M[(size_t)j * n + k]; if (i == j) a += 0.05 * n; S[(size_t)i * n + j] = S[(size_t)j * n + i] = a; }
      for (auto& v : b) v = G01(rs);
      std::vector<double> im(ldltm::tile_image_doubles(n), 0.0);
      for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) { const int pos = ldltm::tile_image_pos(r, c); if (pos >= 0) im[pos] = S[(size_t)r * n + c]; }
      for (int r = 0; r < n; r++) ldltm::image_put_rhs(im.data(), n, r, b[r]);
      CK(hipMalloc(&y.dS, im.size() * 8)); CK(hipMalloc(&y.dx, n * 8)); CK(hipMalloc(&y.dok, 4));
      CK(hipMemcpy(y.dS, im.data(), im.size() * 8, hipMemcpyHostToDevice));
      CK(ldltm::launch_image_pad(n, y.dS, 0));
    }
    ldltx::Context cx[2];
    int bad = 0;
    for (int it = 0; it < N; it++) {
      Sys& y = sys[it & 1];
      ldltx::Context& c = cx[it & 1];
      c.pick = (it >> 6) & 7;
      if (it == N / 2 || it == N / 2 + 1) c.epoch = 0xFFFFFDu;          // the launch counter wraps three launches later (flags cleared, counter restarts)
      CK(ldltx::launch(c, y.n, y.dS, y.dx, y.dok, 0));
      if (it < 2 || it % 997 == 0 || it >= N - 2 || (it >= N / 2 && it < N / 2 + 12)) {
        std::vector<double> x(y.n); int ok = 0;
        CK(hipMemcpy(x.data(), y.dx, y.n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ok, y.dok, 4, hipMemcpyDeviceToHost));
        if (y.first.empty()) y.first = x;
        if (ok != 1 || memcmp(x.data(), y.first.data(), y.n * 8)) { printf("launch %d (n=%d): ok=%d, result differs from the first\n", it, y.n, ok); bad++; }
      }
    }
    CK(hipDeviceSynchronize());
    printf("stress: %d launches, %s\n", N, bad ? "FAILED" : "ALL OK");
    return bad ? 1 : 0;
  }
  std::mt19937_64 rng(12345);
  std::normal_distribution<double> N01(0.0, 1.0);
  const int sizes_all[] = {6, 12, 18, 60, 96, 114, 120, 126, 132, 138, 150, 204, 222, 240, 270, 300};
  const int sizes_q[] = {120};   // quick mode: the C2 window only
  const bool quick = argc > 1 && strcmp(argv[1], "stress");
  const int* sizes = quick ? sizes_q : sizes_all;
  const int nsizes = quick ? 1 : 16;
  for (int si = 0; si < nsizes; si++) {
    const int n = sizes[si];
    if (!ldltm::supports(n)) { printf("n=%d unsupported\n", n); continue; }
    std::vector<double> M((size_t)n * n), S((size_t)n * n), b(n), x(n), xr;
    for (auto& v : M) v = N01(rng);
    for (int i = 0; i < n; i++)
      for (int j = 0; j <= i; j++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += M[(size_t)i * n + k] * M[(size_t)j * n + k];
        if (i == j) s += 0.05 * n;
        S[(size_t)i * n + j] = S[(size_t)j * n + i] = s;
      }
    for (auto& v : b) v = N01(rng);
    host_solve(n, S, b, xr);
    const ldltm::Geo g = ldltm::make_geo(n);
    double *dS, *db, *dx, *dw; int* dok;
    CK(hipMalloc(&dS, S.size() * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dok, 4));
    CK(hipMalloc(&dw, ldltm::wglob_doubles(g) * 8));
    auto to_image = [&](const std::vector<double>& M2) {     // the kernels read the matrix as a tile image; poison the rest
      std::vector<double> im(ldltm::tile_image_doubles(n), std::nan(""));
      for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) { const int pos = ldltm::tile_image_pos(r, c); if (pos >= 0) im[pos] = M2[(size_t)r * n + c]; }
      for (int r = 0; r < n; r++) ldltm::image_put_rhs(im.data(), n, r, b[r]);      // the padding is left to k_image_pad
      return im;
    };
    std::vector<double> im = to_image(S);
    CK(hipFree(dS)); CK(hipMalloc(&dS, im.size() * 8));
    CK(hipMemcpy(dS, im.data(), im.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    CK(ldltm::launch_image_pad(n, dS, 0));
    CK(hipMemset(dx, 0, n * 8)); CK(hipMemset(dok, 0xFF, 4));
    CK(ldltm::launch(n, dS, dx, dok, dw, 0));
    CK(hipDeviceSynchronize());
    int ok = -7;
    CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost));
    double err = 0, mx = 0;
    for (int i = 0; i < n; i++) { err = std::max(err, std::fabs(x[i] - xr[i])); mx = std::max(mx, std::fabs(xr[i])); }
    if (n == 6 && getenv("LDLT_DEBUG")) { for (int i = 0; i < n; i++) printf("   x[%d] = % .6e  ref % .6e\n", i, x[i], xr[i]); }
    const bool good = ok == 1 && err <= 1e-10 * std::max(mx, 1.0);
    // timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    for (int i = 0; i < 10; i++) CK(ldltm::launch(n, dS, dx, dok, dw, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) CK(ldltm::launch(n, dS, dx, dok, dw, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
#ifdef LDLTM_PROFILE
    if (n == 120 || n == 300) {
      long long pr[512];
      CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(ldltm::g_prof), sizeof(pr)));
      printf("  prof n=%d (cycles from kernel start): loaded %lld  factor_done %lld  sync %lld  end %lld", n, pr[1] - pr[0], pr[2] - pr[0], pr[3] - pr[0], pr[4] - pr[0]);
      if (n == 120) printf("  (back-substitution: first column loaded %lld, pairs %lld, store %lld)", pr[5] - pr[3], pr[6] - pr[5], pr[4] - pr[6]);
      printf("\n");
      for (int k = 0; k < g.Tp; k++) {
        const long long* e = pr + 8 + 8 * k;
        printf("   row %2d: factor start %7lld  pivots %6lld (first 8: %5lld) publish %5lld | panel(k,k+1) start %7lld dur %5lld | upd(k+1,k+1) done %7lld\n", k, e[0] - pr[0], e[1] - e[0], e[6] - e[0], e[2] - e[1], e[3] - pr[0], e[4] - e[3], e[5] - pr[0]);
      }
    }
    if (getenv("LDLT_WAVES") && atoi(getenv("LDLT_WAVES")) == n) {
      long long pr[2048];
      CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(ldltm::g_prof), sizeof(pr)));
      printf("  n=%d: loaded %lld factor_done %lld end %lld\n", n, pr[1] - pr[0], pr[2] - pr[0], pr[4] - pr[0]);
      printf("   wave 0 items (kind*100+row : cycles to next):");
      for (int i = 0; i + 1 < 400 && pr[1100 + 2 * i + 2] > 0; i++) printf(" %lld:%lld", pr[1101 + 2 * i], pr[1100 + 2 * i + 2] - pr[1100 + 2 * i]);
      printf("\n");
      for (int k = -1; k < g.Tp; k++) {
        printf("   row %2d:", k);
        for (int w = 0; w < 4; w++) {
          const long long* e = pr + 512 + ((k + 1) * 4 + w) * 8;
          printf(" w%d[%lld d%lld n%lld w%lld b%lld | p@%lld +%lld]", w, e[0] - pr[0], e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3], e[5] - pr[0], e[6] - e[5]);
        }
        printf("\n");
      }
    }
    if (n == 120) {
      long long pr[512];
      CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(ldltm::g_prof), sizeof(pr)));
      printf("   columns loaded (start -> ready):");
      for (int w = 0; w < 8; w++) printf(" %lld->%lld", pr[310 + w] - pr[0], pr[300 + w] - pr[0]);
      printf("\n");
      for (int w = 1; w < 8; w++) {
        printf("   wave %d:", w);
        for (int k = 0; k < w; k++) printf(" r%d[%lld %lld %lld]", k, pr[80 + w * 24 + 3 * k] - pr[0], pr[80 + w * 24 + 3 * k + 1] - pr[0], k + 1 < w ? pr[80 + w * 24 + 3 * k + 2] - pr[0] : 0);
        printf("\n");
      }
    }
#endif
    printf("n=%3d T=%2d ok=%d max|dx|=%.3e (max|x|=%.3e) %s   %.2f us/launch\n", n, g.T, ok, err, mx, good ? "ok" : "FAIL", 1000.0 * ms / reps);
    fails += !good;
    for (int np = ldltx::kMaxP; np <= ldltx::kMaxP && ldltx::supports(n); np += 4) {   // the same system on four / eight compute units of one XCD
      static ldltx::Context cx;
      cx.pick = si & 7;                      // (every size on another XCD)
      if (!ldltx::plan_fits(n, np, 4)) continue;
      CK(hipMemset(dx, 0, n * 8)); CK(hipMemset(dok, 0xFF, 4));
      CK(ldltx::launch(cx, n, dS, dx, dok, 0, np));
      CK(hipDeviceSynchronize());
      int ok2 = -7;
#ifdef LDLTX_WATCHDOG
      {
        int dg[16];
        CK(hipMemcpyFromSymbol(dg, HIP_SYMBOL(ldltx::g_xdog), sizeof(dg)));
        if (dg[0]) { printf("   WATCHDOG n=%d np=%d: %d waits gave up; first: where %d wave %d a %d b %d\n", n, np, dg[0], dg[1], dg[2], dg[3], dg[4]); return 3; }
      }
#endif
      CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ok2, dok, 4, hipMemcpyDeviceToHost));
      double err2 = 0;
      for (int i = 0; i < n; i++) err2 = std::max(err2, std::fabs(x[i] - xr[i]));
      const bool good2 = ok2 == 1 && err2 <= 1e-10 * std::max(mx, 1.0);
      for (int i = 0; i < 10; i++) CK(ldltx::launch(cx, n, dS, dx, dok, 0, np));
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < reps; i++) CK(ldltx::launch(cx, n, dS, dx, dok, 0, np));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ok2, dok, 4, hipMemcpyDeviceToHost));
      double err3 = 0;
      for (int i = 0; i < n; i++) err3 = std::max(err3, std::fabs(x[i] - xr[i]));
      const bool good3 = ok2 == 1 && err3 <= 1e-10 * std::max(mx, 1.0);
      printf("   xcd (%d workgroups, %d slots): ok=%d max|dx|=%.3e %s; after %d launches %.3e %s   %.2f us/launch\n", np, ldltx::plan_max_slots(cx.plan), ok2, err2,
             good2 ? "ok" : "FAIL", reps + 10, err3, good3 ? "ok" : "FAIL", 1000.0 * ms / reps);
      fails += !good2 + !good3;
      {   // the cross-XCD-safe hand-overs (agent-scope release / acquire), forced
        CK(hipMemset(dx, 0, n * 8));
        CK(ldltx::launch(cx, n, dS, dx, dok, 0, np, true));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; i++) CK(ldltx::launch(cx, n, dS, dx, dok, 0, np, true));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ok2, dok, 4, hipMemcpyDeviceToHost));
        double err4 = 0;
        for (int i = 0; i < n; i++) err4 = std::max(err4, std::fabs(x[i] - xr[i]));
        const bool good4 = ok2 == 1 && err4 <= 1e-10 * std::max(mx, 1.0);
        printf("   xcd, agent-scope fences forced: ok=%d max|dx|=%.3e %s   %.2f us/launch\n", ok2, err4, good4 ? "ok" : "FAIL", 1000.0 * ms / 20);
        fails += !good4;
      }
#ifdef LDLTX_PROFILE
      {
        long long z[1024] = {0}, pr[1024];
        for (int q = 430; q < 460; q++) z[q] = 0x7fffffffffffffffll;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(ldltx::g_xprof), z, sizeof(z)));
        CK(ldltx::launch(cx, n, dS, dx, dok, 0, np));
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(ldltx::g_xprof), sizeof(pr)));
        long long t0 = pr[560];
        for (int q = 1; q < np; q++) t0 = std::min(t0, pr[560 + q]);
        auto us = [&](long long t) { return (t - t0) * 0.01; };
        printf("   xcd prof: participants %lld safe %lld; placed %.2f; all waves done %.2f; end %.2f us\n", pr[1], pr[12], us(pr[0]), us(pr[2]), us(pr[3]));
        printf("   back-substitution (start, own block from, to):");
        for (int w = 0; w < 5; w++) printf("  w%d %.2f %.2f %.2f", w, us(pr[300 + 4 * w]), us(pr[301 + 4 * w]), us(pr[302 + 4 * w]));
        printf("\n   own block in clock64() ticks:");
        for (int w = 0; w < 5; w++) printf(" %lld", pr[330 + w]);
        printf("\n");
        printf("   wave done:");
        for (int w = 0; w < 8 * np; w++) printf(" %.1f", us(pr[600 + w]));
        printf("\n");
        for (int k = 0; k < g.Tp; k++)
          if (np == 8) printf("    row %2d: pivots start %.2f done %.2f published %.2f | chain panel: seen %.2f G loaded %.2f, computed %.2f published %.2f done %.2f\n", k, us(pr[16 + 8 * k + 0]), us(pr[16 + 8 * k + 1]), us(pr[16 + 8 * k + 2]), us(pr[16 + 8 * k + 6]), us(pr[16 + 8 * k + 3]), us(pr[16 + 8 * k + 4]), us(pr[16 + 8 * k + 5]), us(pr[16 + 8 * k + 7]));
        printf("   generic panel tiles of a row published (first .. last) after its G: ");
        for (int k = 0; k + 4 < g.Tp; k++) printf(" %.1f..%.1f", us(pr[430 + k]) - us(pr[16 + 8 * k + 2]), us(pr[400 + k]) - us(pr[16 + 8 * k + 2]));
        printf("\n");
      }
#endif
#ifdef LDLTX_WATCHDOG
      if (n == 300 && np == 8) {   // a participant that is never placed (a grid one participant short): every wait gives up, the launch reports "timed out" (-2, not 0 = "not positive definite"), no hang
        const auto t_w0 = std::chrono::steady_clock::now();
        CK(ldltx::launch(cx, n, dS, dx, dok, 0, np, false, /*one_short*/ true));
        CK(hipDeviceSynchronize());
        const double wd_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w0).count();
        int ok3 = -7, dg[16];
        CK(hipMemcpy(&ok3, dok, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpyFromSymbol(dg, HIP_SYMBOL(ldltx::g_xdog), sizeof(dg)));
        const bool wd_ok = ok3 == ldltx::kOkTimedOut && dg[0] > 0 && dg[1] == 7 && wd_s > 0.2 && wd_s < 2.0;    // (the bound is wall-clock time: 0.25 s in this build)
        printf("   xcd with a participant missing: ok=%d after %.2f s, %d waits gave up (first: where %d) %s\n", ok3, wd_s, dg[0], dg[1], wd_ok ? "ok" : "FAIL");
        fails += !wd_ok;
        // ... and the next launch of the same context is a good one (nothing of the failed launch sticks)
        CK(ldltx::launch(cx, n, dS, dx, dok, 0, np));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&ok3, dok, 4, hipMemcpyDeviceToHost));
        printf("   xcd after the timed-out launch: ok=%d %s\n", ok3, ok3 == 1 ? "ok" : "FAIL");
        fails += ok3 != 1;
        int z[16] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(ldltx::g_xdog), z, sizeof(z)));
      }
#endif
      if (n == 300 && np == 8) {
        std::vector<double> S2 = S;
        for (int i = 0; i < n; i++) S2[i] = S2[(size_t)i * n] = 0.0;
        std::vector<double> im2 = to_image(S2);
        CK(hipMemcpy(dS, im2.data(), im2.size() * 8, hipMemcpyHostToDevice));
        CK(ldltm::launch_image_pad(n, dS, 0));
        CK(ldltx::launch(cx, n, dS, dx, dok, 0, np));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&ok2, dok, 4, hipMemcpyDeviceToHost));
        printf("   xcd zero pivot: ok=%d %s\n", ok2, ok2 == 0 ? "ok" : "FAIL");
        fails += ok2 != 0;
      }
    }
    if (n == 120) {   // zero pivot -> ok = 0
      std::vector<double> S2 = S;
      for (int i = 0; i < n; i++) S2[i] = S2[(size_t)i * n] = 0.0;
      std::vector<double> im2 = to_image(S2);
      CK(hipMemcpy(dS, im2.data(), im2.size() * 8, hipMemcpyHostToDevice));
      CK(ldltm::launch_image_pad(n, dS, 0));
      CK(ldltm::launch(n, dS, dx, dok, dw, 0));
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost));
      printf("zero pivot: ok=%d %s\n", ok, ok == 0 ? "ok" : "FAIL");
      fails += ok != 0;
    }
    CK(hipFree(dS)); CK(hipFree(db)); CK(hipFree(dx)); CK(hipFree(dok)); CK(hipFree(dw));
  }
  printf(fails ? "FAILED (%d)\n" : "ALL OK\n", fails);
  return fails ? 1 : 0;
}
