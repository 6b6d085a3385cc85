// Dense LDL^T + solve of the reduced camera system on EIGHT workgroups of ONE XCD (round 5).  The C4 window (50 free poses, 300
// unknowns, 19 tile rows) takes 111 us alone and 140 us next to the tracking streams on the one compute unit of
// ldltm::k_ldlt_big48; here 84 / 92-100 us.  The 190 tiles of 16 x 16 live in the registers of 64 wavefronts on 8 compute units that share
// an L2.  What ldltm::k_ldlt_mfma hands from wavefront to wavefront through LDS -- G = L_kk^-1 and D^-1 of a diagonal tile, the
// -R / W images of a panel tile, the "published" flags -- goes through global memory that stays in that L2:
//   * stores are plain (the vector L1 writes through), s_waitcnt vmcnt(0), then a relaxed agent-scope store of the flag word;
//   * flags and operands are read with relaxed agent-scope atomic loads (sc1: they miss in the CU's vector L1 and are served by the
//     XCD's L2): 0.45 us from the flag store to the reader seeing it, 0.16 .. 0.6 us per load round trip (quiet .. 64 wavefronts
//     polling); tools/micro/xcd_handover measured 0.59 us per 2 KB hand-over against 0.93 / 1.25 us for an agent-scope release /
//     acquire pair inside / across XCDs.  Coherent ONLY because all participants share one L2 -- so:
// Placement: a grid of up to 64 blocks; blocks x, x + 8, .., x + 56 take part (the dispatcher deals a grid's workgroups round-robin
// over the 8 XCDs), the others leave at once.  x differs from user to user (process id + handle number): the participants spin
// until all eight are resident, so solves that run at the same time must not crowd one XCD's 32 compute units -- with the
// users spread over the XCDs that takes more than 32 concurrent solves of 21+ free poses on one GPU (one agent per GPU has one).  Every participant posts HW_REG_XCC_ID; if the ids differ (another partition mode, a changed
// dispatcher) every hand-over becomes an agent-scope release / acquire pair: slower (147 us), correct, tested (ORBG_LDLT_XCD=safe).
// Flags carry the launch's number (epoch): nothing is cleared between launches.
// G and D^-1 of a diagonal tile -- the one hand-over per tile row on the critical path -- travel without a flag, as self-validating
// (value, value ^ tag) pairs: the producer does not wait for its stores and the consumer's poll is its load (see st_pair / get_G).
// Schedule: column j's last four tiles (j-3, j) .. (j, j) sit on one "chain" wavefront, which per row k = j-3 .. j-1 spins on
// G_k's flag, solves (k, j), updates the tiles below from registers / one L2 image, and for k = j-1 runs the pivots of (j, j) at
// once: ONE hand-over per tile row on the critical path.  The other tiles, <= 4 per wavefront, are taken one after the other (every
// row above the tile, then G of its row, the solve, the publication).  Same arithmetic per tile as ldltm::k_ldlt_mfma (pivot pairs as
// rank-2 matrix instructions, G collected on an identity copy) in one fixed order: bitwise reproducible.  The back-substitution
// runs on five wavefronts of workgroup 0, one per 64-row block, x posted through LDS.  DESIGN.md section 8 has the timeline.
#pragma once

#include <time.h>
#include <unistd.h>

#include <atomic>
#include <type_traits>

#include "ldlt_mfma.hpp"

namespace ldltx {

using ldltm::d4;
using ldltm::Geo;
using ldltm::kGld;
using ldltm::make_geo;
using ldltm::mfma;
using ldltm::rcp1;
using ldltm::rdlane;
using ldltm::row_even_to_odd;

constexpr int kMaxP = 8;                    // participating workgroups (compute units of one XCD)
constexpr int kWgWaves = 8, kThreads = 64 * kWgWaves;
constexpr int kMaxW = kMaxP * kWgWaves;     // wavefronts that hold tiles
constexpr int kChain = 4;                   // tiles of a column, from the diagonal up, that its chain wavefront holds
constexpr int kMaxNS = 8;                   // tile slots per wavefront in the plan (the kernel uses 4)
constexpr int kMaxT = ldltm::kMaxT;
constexpr int kNY = 5;                      // 64-lane groups of the solution vector (n_pad <= 320)
// flag words
constexpr int kFDiag = 0 /* (unused since G travels as self-validating pairs) */, kFPanel = 32, kFWave = kFPanel + kMaxT * kMaxT, kFElect = kFWave + 128, kFBad = kFElect + 16, kFDog = kFBad + 1;
constexpr int kFlagStride = 640;             // one copy of the flags per participant (its wavefronts poll that copy only)
constexpr int kFlagWords = kFlagStride * kMaxP;
// scratch (doubles)
constexpr size_t kPanOff = 0, kPanDoubles = (size_t)kMaxT * kMaxT * 512;
constexpr size_t kGbOff = kPanOff + kPanDoubles, kGbDoubles = 2 * (size_t)kMaxT * 16 * kGld;      // (value, value ^ tag) pairs
constexpr size_t kDvOff = kGbOff + kGbDoubles, kDvDoubles = 2 * (size_t)kMaxT * 16;
constexpr size_t kWOff = kDvOff + kDvDoubles;
// (the factor's store is sized for all 64 * kNY columns: the back-substitution preloads the 64 columns of a wavefront's block
// whether or not the system reaches them -- masked where used, but read)
__host__ inline size_t scratch_doubles() {
  const size_t w = ldltm::wglob_doubles(make_geo(16 * kMaxT - 20)), full = (size_t)(64 * kNY) * (64 * kNY + 1) / 2;
  return kWOff + (w > full ? w : full) + 1024;
}
// *ok_flag of a launch whose waits gave up (LDLTX_DOG): neither "solved" (1) nor "not positive definite" (0)
constexpr int kOkTimedOut = -2;

struct Plan { short tile[kMaxW][kMaxNS]; signed char chain[kMaxW]; int np, ns, force_safe, pick; unsigned long long nonce; };       // tile index j(j+1)/2 + i per wavefront slot (-1: none), in processing order

__host__ inline bool plan_fits(int n, int np, int ns);
// The systems this kernel takes: 9 .. 19 tile rows (21 .. 50 free poses).  Inside a local BA it is ahead of the one-workgroup kernels
// at every such size (tools/lba_sizes.py, same box: the whole solve 0.688 vs 0.703 ms at 21 free poses, 0.842 vs 0.913 at 29, 0.998
// vs 1.112 at 36, 1.10 vs 1.26 at 40, 1.36 vs 1.68 at 50); alone (tools/micro/ldlt_mfma_test) 32 / 38 / 52 / 65 / 77 / 84 us against
// 34 / 42 / 63 / 80 / 99 / 111 us at 9 / 10 / 13 / 16 / 18 / 19 rows.  8 tile rows (the C2 window) stay on ldltm::k_ldlt_cols: 19.6 us.
__host__ inline bool supports(int n) { const Geo g = make_geo(n); return n >= 1 && g.T >= 9 && g.T <= kMaxT - 1 && g.n_pad <= 64 * kNY && plan_fits(n, kMaxP, 4); }
__host__ inline bool pays(int n) { return supports(n); }

// Tiles to wavefronts.  Column j's last kChain tiles go to one "chain" wavefront (consecutive columns on different compute units)
// that holds nothing else; the other tiles are dealt round the other wavefronts in (row, column) order.
__host__ inline Plan make_plan(int n, int np, int ns) {
  const Geo g = make_geo(n);
  const int W = np * kWgWaves;
  Plan P;
  P.np = np; P.ns = ns; P.force_safe = 0; P.pick = 0; P.nonce = 0;
  int cnt[kMaxW];
  bool chain[kMaxW];
  for (int w = 0; w < kMaxW; w++) { cnt[w] = 0; chain[w] = false; for (int s = 0; s < kMaxNS; s++) P.tile[w][s] = -1; }
  auto put = [&](int w, int i, int j) { P.tile[w][cnt[w]++] = (short)ldltm::tile_index(i, j); };
  auto chain_wave = [&](int j) { return (j % np) * kWgWaves + (j / np) % kWgWaves; };
  for (int j = 0; j < g.T; j++) {           // (j-3, j) .. (j, j) in slots 0 .. 3 (-1 where there is none)
    const int w = chain_wave(j);
    chain[w] = true;
    cnt[w] = kChain;
    for (int q = 0; q < kChain; q++) P.tile[w][q] = j - (kChain - 1) + q >= 0 ? (short)ldltm::tile_index(j - (kChain - 1) + q, j) : (short)-1;
  }
  // the other tiles in (row, column) order, dealt round the other wavefronts: a wavefront takes its tiles one after the other, so
  // its tiles should be rows apart (with 45 wavefronts for 120 tiles: rows 0-3, 3-7, 8+) -- the second one then has caught up
  // on its rows long before its G arrives.  (Dealt by column to the least loaded wavefront, tiles of neighbouring rows met on one
  // wavefront and the later one came out 4-8 us after its G; four tiles of one ROW per wavefront was slower still.)
  {
    int order[kMaxW], no = 0;                // the other wavefronts, walking the workgroups round-robin
    for (int q = 0; q < kWgWaves; q++) for (int p2 = 0; p2 < np; p2++) if (!chain[p2 * kWgWaves + q]) order[no++] = p2 * kWgWaves + q;
    int t = 0;
    for (int i = 0; i + kChain < g.T; i++)
      for (int j = i + kChain; j < g.T; j++) { put(order[t % no], i, j); t++; }
  }
  for (int w = 0; w < kMaxW; w++) P.chain[w] = chain[w];
  for (int w = 0; w < W; w++) {             // order by (row, column); insertion sort of <= 8 entries
    if (P.chain[w]) continue;
    auto key = [&](short t) { int j = 0; while ((j + 1) * (j + 2) / 2 <= t) j++; const int i = t - j * (j + 1) / 2; return i * 64 + j; };
    for (int a = 1; a < cnt[w]; a++) {
      const short t = P.tile[w][a];
      int b = a - 1;
      while (b >= 0 && key(P.tile[w][b]) > key(t)) { P.tile[w][b + 1] = P.tile[w][b]; b--; }
      P.tile[w][b + 1] = t;
    }
  }
  return P;
}
__host__ inline bool plan_fits(int n, int np, int ns) {
  const Geo g = make_geo(n);
  int ct = 0;
  for (int j = 0; j < g.T; j++) ct += j + 1 < kChain ? j + 1 : kChain;
  return g.T <= np * kWgWaves && ns >= kChain && g.ntiles - ct <= (np * kWgWaves - g.T) * ns;
}
__host__ inline int plan_max_slots(const Plan& P) { int m = 0; for (int w = 0; w < kMaxW; w++) for (int s = 0; s < kMaxNS; s++) if (P.tile[w][s] >= 0 && s + 1 > m) m = s + 1; return m; }

#ifdef LDLTX_PROFILE
__device__ long long g_xprof[1024];
#define LDLTX_T(i) do { if (lane == 0) g_xprof[(i)] = wall_clock64(); } while (0)
#else
#define LDLTX_T(i) do { } while (0)
#endif

// Every in-kernel wait is bounded IN TIME: a wavefront that has been in the kernel for kDogTicks of the 100 MHz wall clock (2 s; a
// launch takes 0.1 ms, a hand-over a microsecond) and is still waiting gives up, marks the launch as TIMED OUT (kFDog; *ok_flag =
// kOkTimedOut -- distinct from a non-positive pivot, which is an LM verdict: lba.hip re-solves a timed-out window on the
// one-workgroup kernel and counts it) and lets the kernel end -- a participant that the dispatcher never placed must not turn into a
// GPU that never comes back.  The clock is read once per 1024 polls of a wait (a poll is 0.1 .. 1 us).  Time, not a poll count:
// a live launch whose participants are placed late behind long kernels of other streams must not trip it.  -DLDLTX_WATCHDOG
// (micro-benchmark builds) gives up after 0.25 s and records which wait it was.
#ifdef LDLTX_WATCHDOG
__device__ int g_xdog[16];
constexpr long long kDogTicks = 25000000ll;
#define LDLTX_DOG_RECORD(where, a, b) do { if (lane == 0 && atomicAdd(&g_xdog[0], 1) == 0) { g_xdog[1] = (where); g_xdog[2] = gw_; g_xdog[3] = (a); g_xdog[4] = (b); } } while (0)
#else
constexpr long long kDogTicks = 200000000ll;
#define LDLTX_DOG_RECORD(where, a, b) do { } while (0)
#endif
#define LDLTX_DOG(where, a, b) do { if ((++dog_ & 1023) == 0 && (long long)wall_clock64() - t_dog0_ > kDogTicks) { LDLTX_DOG_RECORD(where, a, b); if (lane == 0) { __hip_atomic_store(flags + kFBad, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(flags + kFDog, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } bail_ = true; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}
__device__ __forceinline__ double ld_l2(const double* p) {
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ unsigned ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ __launch_bounds__(kThreads) void k_ldlt_xcd(int n, const double* __restrict__ St, double* __restrict__ x, int* __restrict__ ok_flag,
                                                       double* scr, unsigned* flags, unsigned epoch, Plan plan) {
#pragma clang fp contract(fast)        // the solve is tolerance-checked (1e-4), not bit-compared
  __shared__ __attribute__((aligned(32))) double s_x[64 * kNY];        // the solution, posted block by block during the back-substitution
  __shared__ int s_xready;                 // column groups (from the last column down) whose x_J are posted
  constexpr int kNS = 4;
  const int kP = plan.np, kW = kP * kWgWaves;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  int dog_ = 0; [[maybe_unused]] const int gw_ = (int)(blockIdx.x >> 3) * kWgWaves + wv;
  const long long t_dog0_ = (long long)wall_clock64();      // the wavefront's start: the time base of every wait's bound
  bool bail_ = false;                      // a wait gave up (LDLTX_DOG): every later wait of this wavefront returns at once
#ifdef LDLTX_PROFILE
  const long long t_enter = wall_clock64();
#endif
  // ---- which workgroups take part: block 8 r is participant r (the dispatcher deals a grid's workgroups round-robin over the 8
  // XCDs, so these share one); the others leave at once.  Every participant posts its XCC id; if they differ after all (another
  // partition mode, a changed dispatcher) `safe` turns the hand-overs into agent-scope release / acquire pairs -- slower, correct.
  if ((blockIdx.x & 7u) != (unsigned)plan.pick) return;
  const int rank = blockIdx.x >> 3;
  const unsigned my_xcc = xcc_id() & 0xFFu, ep = epoch & 0xFFFFFFu;
  if (tid == 0) {
    __hip_atomic_store(flags + kFElect + rank, (ep << 8) | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_xready = 0;
  }
  __syncthreads();
  const int gw = rank * kWgWaves + wv;
#ifdef LDLTX_PROFILE
  if (tid == 0) { atomicAdd((unsigned long long*)&g_xprof[1], 1ull); g_xprof[560 + rank] = t_enter; }
#endif
  const Geo G = make_geo(n);
  const int T = G.T, n_pad = G.n_pad, cb = G.cb;
  double* const Pan = scr + kPanOff;      // [T][T][2][256]: -R and W of panel tile (k, j), operand layout = accumulator layout
  double* const Gb = scr + kGbOff;        // [T][16 * kGld] self-validating pairs
  double* const Dv = scr + kDvOff;        // [T][16] self-validating pairs
  double* const Wg = scr + kWOff;         // the unit upper factor, column-packed: entry (I, J), I < J, at J (J - 1) / 2 + I
  // (every participant polls its own copy of the G / panel flags: sixty-four wavefronts polling one cache line queue up in the
  // L2 channel that holds it -- an idle sweep took 3 us; the publisher writes the eight copies with one store instruction)
  unsigned* const f_panel = flags + rank * kFlagStride + kFPanel;
  auto wm_store = [&](int I, int J, double v) { Wg[J * (J - 1) / 2 + I] = v; };

  // ---- this wavefront's tiles
  d4 acc[kNS];
  int ti0[kNS], tj0[kNS];
#pragma unroll
  for (int s = 0; s < kNS; s++) {
    const int t = __builtin_amdgcn_readfirstlane((int)plan.tile[gw][s]);
    int i = 1 << 20, j = 1 << 20;
    if (t >= 0) {
      j = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while ((j + 1) * (j + 2) / 2 <= t) j++;
      while (j * (j + 1) / 2 > t) j--;
      i = t - j * (j + 1) / 2;
    }
    ti0[s] = i; tj0[s] = j;
    const d4 v = *reinterpret_cast<const d4*>(St + (size_t)max(t, 0) * 256 + 4 * lane);
#pragma unroll
    for (int g = 0; g < 4; g++) acc[s][g] = t >= 0 ? v[g] : 0.0;
  }

  bool safe;
  for (;;) {
    const unsigned f = ld_flag(flags + kFElect + min(lane, kP - 1));
    if (__builtin_amdgcn_ballot_w64((f >> 8) != ep) == 0) { safe = plan.force_safe || __builtin_amdgcn_ballot_w64((f & 0xFFu) != my_xcc) != 0; break; }
    __builtin_amdgcn_s_sleep(1);
    LDLTX_DOG(7, 0, 0);                    // (a participant that was never placed)
    if (bail_) { safe = true; break; }
  }
#ifdef LDLTX_PROFILE
  if (gw == 0) { LDLTX_T(0); if (lane == 0) g_xprof[12] = safe; }
#endif
  auto publish = [&](int idx, bool copies) { // every store of this wavefront has reached the L2 (or, `safe`, the memory); then the flag
    if (safe) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < (copies ? kMaxP : 1)) __hip_atomic_store(flags + lane * kFlagStride + idx, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto acquire = [&]() { if (safe) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); };
  // G and D^-1 of a diagonal tile travel WITHOUT a flag: every value v is stored as the pair (v, v ^ tag), tag a 64-bit hash of
  // the launch number, with two agent-scope 8-byte stores; a reader loads both words and takes the value when their XOR is the tag
  // -- a word of an earlier launch, or one of the two not yet arrived, fails the test (two unrelated doubles XOR to the tag with
  // probability 2^-64).  The producer does not wait for its stores (0.25 us per tile row on the chain), the consumer's poll IS its
  // load (one L2 round trip instead of flag + data), and the words are coherent across XCDs as they are (agent-scope accesses).
  // (plan.nonce: fresh per Context::bind and per wrap of the launch counter -- a recycled allocation or a reused launch number
  // cannot carry pairs that pass for this launch's)
  const unsigned long long tag = ((0x9E3779B97F4A7C15ull * (unsigned long long)(epoch + 1u)) ^ plan.nonce) | 1ull;
  auto st_pair = [&](double* p, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    if (safe) {                              // across XCDs: write-through stores
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, b ^ tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {                                 // one L2: plain stores (the vector L1 writes through to it)
      typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<ull2*>(p) = ull2{b, b ^ tag};
    }
  };
  // G_k (as the panel's A operand chunks) and D_k^-1 for this lane; `eager`: the caller's next step is this G (a chain wavefront):
  // poll with the full load; otherwise poll one word first (sixteen wavefronts polling 8 KB each would be 400 GB/s of L2 traffic)
  auto get_G = [&](int k, double (&Gf)[4], double (&dv4)[4], bool eager) {
    const double* const gk = Gb + (size_t)k * 32 * kGld;
    const double* const dk = Dv + (size_t)k * 32;
    if (!eager) {
      for (;;) {
        const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(dk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(dk) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane((unsigned)((a ^ b) == tag)) || bail_) break;
        LDLTX_DOG(1, k, 0);
      }
    }
    for (;;) {
      unsigned long long a[8], b[8];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned long long* const pg = reinterpret_cast<const unsigned long long*>(gk + 2 * ((4 * q + lr) * kGld + lc));
        const unsigned long long* const pd = reinterpret_cast<const unsigned long long*>(dk + 2 * (lr + 4 * q));
        a[q] = __hip_atomic_load(pg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b[q] = __hip_atomic_load(pg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a[4 + q] = __hip_atomic_load(pd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b[4 + q] = __hip_atomic_load(pd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      bool good = true;
#pragma unroll
      for (int q = 0; q < 8; q++) good = good && (a[q] ^ b[q]) == tag;
      if (__builtin_amdgcn_ballot_w64(!good) == 0 || bail_) {
#pragma unroll
        for (int q = 0; q < 4; q++) { Gf[q] = __longlong_as_double((long long)a[q]); dv4[q] = __longlong_as_double((long long)a[4 + q]); }
        break;
      }
      LDLTX_DOG(1, k, 1);
    }
  };

  // ---- the pivots of diagonal tile k, two per matrix instruction (ldltm::k_ldlt_mfma's `factor`), G and D^-1 to the L2
  auto factor = [&](int k) {
    LDLTX_T(16 + 8 * k + 0);
    d4 C = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < kNS; s++)
      if (ti0[s] == k && tj0[s] == k) C = acc[s];
    d4 E, Gc, Wc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; g++) E[g] = (lr + 4 * g == lc) ? 1.0 : 0.0;
    Gc = E;
    double dvv = 1.0;
    const int npiv = min(16, n_pad - 16 * k);          // a multiple of 4
    double rlast = 1.0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (4 * g < npiv) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int q0 = 2 * h, p0 = 4 * g + q0, p1 = p0 + 1;
          double u = C[g];
          asm volatile("" : "+v"(u));
          const double c00 = rdlane(u, q0 * 16 + p0), c01 = rdlane(u, q0 * 16 + p1), c11 = rdlane(u, (q0 + 1) * 16 + p1);
          const double det = __builtin_fma(c00, c11, -(c01 * c01));
          const double r0 = rcp1(c00);
          const double rdet = rcp1(det);
          const double r1 = c00 * rdet;
          const double nl10 = -(c01 * r0);
          const double u0b = row_even_to_odd(u);
          const double u1 = __builtin_fma(nl10, u0b, u);
          const bool in0 = lr == q0, in1 = lr == q0 + 1;
          const double bv = in1 ? u1 : u;
          const double av = in0 ? u * -r0 : in1 ? u1 * -r1 : 0.0;
          if (p1 < 15) C = mfma(av, bv, C);
          double eg = E[g];
          asm volatile("" : "+v"(eg));
          const double e0b = row_even_to_odd(eg);
          const double erow = in1 ? __builtin_fma(nl10, e0b, eg) : eg;
          Gc[g] = (in0 || in1) ? erow : Gc[g];
          if (p1 < 15) E = mfma(av, erow, E);
          Wc[g] -= av;
          dvv = lane == p0 ? r0 : lane == p1 ? r1 : dvv;
          if (h == 1) rlast = r0 + r1;
        }
      }
    }
    const bool good = fabs(rlast) < INFINITY;
    LDLTX_T(16 + 8 * k + 1);
    double* const gk = Gb + (size_t)k * 32 * kGld;
#pragma unroll
    for (int g = 0; g < 4; g++) st_pair(gk + 2 * (lc * kGld + lr + 4 * g), Gc[g]);
    if (lane < 16) st_pair(Dv + (size_t)k * 32 + 2 * lane, dvv);
    if (!good && lane == 0) __hip_atomic_store(flags + kFBad, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    LDLTX_T(16 + 8 * k + 2);
#pragma unroll
    for (int g = 0; g < 4; g++) {              // the factor's rows are not needed before the back-substitution
      const int I = 16 * k + lr + 4 * g, J = 16 * k + lc;
      if (I < J && I < n_pad && J <= cb) wm_store(I, J, Wc[g]);
    }
  };
  // ---- the schedule.  A chain wavefront (tiles (j-3, j) .. (j, j) in slots 0 .. 3) first applies the rows above its tiles, row by
  // row: wait for the row's panel images of its four tiles, all operand images in flight together, update.  Then per row k = j-3,
  // j-2, j-1: it spins on G_k's flag, solves (k, j), updates the diagonal tile from registers and the tiles between with the -R
  // images of (k, k+1) .., which the chain wavefronts of those columns publish at about the same time; for k = j - 1 the pivots
  // of (j, j) follow at once.  One hand-over per tile row on the critical path.  The other wavefronts: see below.  Every wait is
  // for something of a row above, published by wavefronts that have only rows above behind them: no cycle
  // (tests/cpp/ldlt_xcd_plan_check.hip replays the schedule for every size).
  const bool chain_wave = plan.chain[gw] != 0;
  // (a chain wavefront shares its SIMD with one that updates up to four tiles, 16 matrix instructions back to back: it goes first)
  if (chain_wave) __builtin_amdgcn_s_setprio(3);
  bool on[kNS], isdiag[kNS];
  const unsigned* fpa[kNS]; const unsigned* fpb[kNS];
  unsigned offa[kNS], offw[kNS];
  int r_end = 0;                           // rows the row loop runs over
#pragma unroll
  for (int s = 0; s < kNS; s++) {
    on[s] = ti0[s] < (1 << 20);
    const int i = on[s] ? ti0[s] : 0, j = on[s] ? tj0[s] : 0;
    isdiag[s] = on[s] && i == j;
    fpa[s] = f_panel + i; fpb[s] = f_panel + j;
    offa[s] = (unsigned)i * 512u; offw[s] = (unsigned)j * 512u + 256u;
    if (on[s]) r_end = max(r_end, i + 1);
  }
  if (chain_wave) {                        // the first row its chain steps take over
    r_end = 1 << 20;
#pragma unroll
    for (int s = 0; s < kNS; s++) if (on[s]) r_end = min(r_end, ti0[s]);
  }
  const unsigned row_words = (unsigned)T, row_doubles = (unsigned)T * 512u;
  auto spin = [&](const unsigned* f, int where, int a2) {
    while (__builtin_amdgcn_readfirstlane(ld_flag(f)) != epoch && !bail_) { LDLTX_DOG(where, a2, 0); }
  };
  // ---- a wavefront that holds no chain tiles takes its tiles ONE AFTER THE OTHER, in row order: every row above the tile (up to
  // kRB rows' operand images in flight together: rows published long ago are caught up on at the matrix-instruction rate), then G of
  // the tile's row, the solve, the publication.  Walking all its tiles row by row (the loop below, still used by the chain
  // wavefronts for the rows above their four tiles) made a tile wait for the latest flag of the OTHER tiles' rows, and a row's
  // panel tiles came out up to 7 us after its G.  Waits are for rows above the tile only and tiles are taken in row order: no cycle.
  if (!chain_wave) {
    constexpr int kRB = 4;
    auto take = [&](d4 c, const int i, const int j) __attribute__((always_inline)) {
      const unsigned* fa = f_panel + i; const unsigned* fb = f_panel + j;
      unsigned oa_off = (unsigned)i * 512u, ow_off = (unsigned)j * 512u + 256u;
      int r = 0;
      while (r < i) {
        const int nb = min(kRB, i - r);
        int m = 0;
        for (;;) {
          unsigned f1[kRB], f2[kRB];
#pragma unroll
          for (int u = 0; u < kRB; u++) { const unsigned d = (unsigned)min(u, nb - 1) * row_words; f1[u] = ld_flag(fa + d); f2[u] = ld_flag(fb + d); }
          bool run = true;
          m = 0;
#pragma unroll
          for (int u = 0; u < kRB; u++) {
            run = run && u < nb && __builtin_amdgcn_readfirstlane(f1[u]) == epoch && __builtin_amdgcn_readfirstlane(f2[u]) == epoch;
            m += run;
          }
          if (m > 0 || bail_) break;
          LDLTX_DOG(3, r, i * 100 + j);
        }
        if (bail_) break;
        acquire();
        double oa[kRB][4], ow[kRB][4];
#pragma unroll
        for (int u = 0; u < kRB; u++) {            // (rows past the ready ones re-read the last ready row: no branch around loads)
          const unsigned d = (unsigned)min(u, m - 1) * row_doubles;
          const double* const pa = Pan + oa_off + d + lane;
          const double* const pw = Pan + ow_off + d + lane;
#pragma unroll
          for (int q = 0; q < 4; q++) { oa[u][q] = ld_l2(pa + q * 64); ow[u][q] = ld_l2(pw + q * 64); }
        }
#pragma unroll
        for (int u = 0; u < kRB; u++) {
          if (u < m) {
#pragma unroll
            for (int q = 0; q < 4; q++) c = mfma(oa[u][q], ow[u][q], c);
          }
        }
        r += m; fa += (unsigned)m * row_words; fb += (unsigned)m * row_words; oa_off += (unsigned)m * row_doubles; ow_off += (unsigned)m * row_doubles;
      }
      // the tile against G_i
      double Gf[4], dv4[4];
      get_G(i, Gf, dv4, false);
      d4 R0 = {0.0, 0.0, 0.0, 0.0}, R1 = {0.0, 0.0, 0.0, 0.0};
      R0 = mfma(Gf[0], c[0], R0);
      R1 = mfma(Gf[2], c[2], R1);
      R0 = mfma(Gf[1], c[1], R0);
      R1 = mfma(Gf[3], c[3], R1);
      double* const pb = Pan + ((size_t)(i * T + j) * 2) * 256 + lane;
      double w4[4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const double rr = R0[g] + R1[g];
        w4[g] = rr * dv4[g];
        pb[g * 64] = -rr;
        pb[256 + g * 64] = w4[g];
      }
      publish(kFPanel + i * T + j, true);
#ifdef LDLTX_PROFILE
      if (lane == 0) { atomicMax((unsigned long long*)&g_xprof[400 + i], (unsigned long long)wall_clock64()); atomicMin((unsigned long long*)&g_xprof[430 + i], (unsigned long long)wall_clock64()); }
#endif
      const int J = 16 * j + lc;
      if (J <= cb) {
#pragma unroll
        for (int g = 0; g < 4; g++) wm_store(16 * i + lr + 4 * g, J, w4[g]);
      }
    };
#pragma unroll
    for (int s = 0; s < kNS; s++) {
      int i = ti0[s], j = tj0[s];
      asm volatile("" : "+s"(i), "+s"(j));
      if (on[s]) take(acc[s], i, j);
    }
    r_end = 0;
  }
  for (int r = 0; r < r_end; r++) {
    // (opaque copies: everything derived from a slot's tile position is loop-invariant, and hoisted out of this loop it costs 64
    // vector registers of store addresses and spills)
    int ti[kNS], tj[kNS];
#pragma unroll
    for (int s = 0; s < kNS; s++) { ti[s] = ti0[s]; tj[s] = tj0[s]; asm volatile("" : "+s"(ti[s]), "+s"(tj[s])); }
    // ---- tiles below row r: all their flags, then all their operand images, then the updates
    bool need[kNS], any = false;
#pragma unroll
    for (int s = 0; s < kNS; s++) { need[s] = on[s] && ti[s] > r; any = any || need[s]; }
    if (any) {
      for (;;) {
        unsigned fa[kNS], fb[kNS];
#pragma unroll
        for (int s = 0; s < kNS; s++) { fa[s] = ld_flag(fpa[s]); fb[s] = ld_flag(fpb[s]); }
        bool all = true;
#pragma unroll
        for (int s = 0; s < kNS; s++)
          all = all && (!need[s] || (__builtin_amdgcn_readfirstlane(fa[s]) == epoch && __builtin_amdgcn_readfirstlane(fb[s]) == epoch));
        if (all || bail_) break;
        LDLTX_DOG(3, r, ti0[0] * 100 + tj0[0]);
      }
      acquire();
      double oa[kNS][4], ow[kNS][4];
#pragma unroll
      for (int s = 0; s < kNS; s++) {            // unconditional loads (a branch would drain the memory counter per slot)
        const double* const pa = Pan + offa[s] + lane;
        const double* const pw = Pan + offw[s] + lane;
#pragma unroll
        for (int q = 0; q < 4; q++) { oa[s][q] = ld_l2(pa + q * 64); ow[s][q] = ld_l2(pw + q * 64); }
      }
#pragma unroll
      for (int s = 0; s < kNS; s++) {
        if (need[s]) {
          d4 c = acc[s];
#pragma unroll
          for (int q = 0; q < 4; q++) c = mfma(oa[s][q], ow[s][q], c);
          acc[s] = c;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < kNS; s++) { fpa[s] += row_words; fpb[s] += row_words; offa[s] += row_doubles; offw[s] += row_doubles; }
  }
  if (chain_wave) {
    int ti[kNS], tj[kNS];
#pragma unroll
    for (int s = 0; s < kNS; s++) { ti[s] = ti0[s]; tj[s] = tj0[s]; asm volatile("" : "+s"(ti[s]), "+s"(tj[s])); }
    constexpr int D = kChain - 1;                // the diagonal tile's slot
    auto chain_step = [&](auto SC) {
      constexpr int S = decltype(SC)::value;
      if (!on[S]) return;
      const int k = ti[S], j = tj[S];
      double Gf[4], dv4[4];
      get_G(k, Gf, dv4, true);
      if (S == D - 1) { LDLTX_T(16 + 8 * k + 6); LDLTX_T(16 + 8 * k + 3); }
      const d4 X = acc[S];
      d4 R0 = {0.0, 0.0, 0.0, 0.0}, R1 = {0.0, 0.0, 0.0, 0.0};
      R0 = mfma(Gf[0], X[0], R0);
      R1 = mfma(Gf[2], X[2], R1);
      R0 = mfma(Gf[1], X[1], R0);
      R1 = mfma(Gf[3], X[3], R1);
      double* const pb = Pan + ((size_t)(k * T + j) * 2) * 256 + lane;
      double nR[4], w4[4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const double rr = R0[g] + R1[g];
        nR[g] = -rr;
        w4[g] = rr * dv4[g];
        pb[g * 64] = nR[g];
        pb[256 + g * 64] = w4[g];
      }
      if (S == D - 1) LDLTX_T(16 + 8 * k + 4);
      publish(kFPanel + k * T + j, true);
      if (S == D - 1) LDLTX_T(16 + 8 * k + 5);
      {                                            // (the same order of operations as every other tile update: results do not
        d4 c = acc[D];                             // depend on who applied a row)
#pragma unroll
        for (int q = 0; q < 4; q++) c = mfma(nR[q], w4[q], c);
        acc[D] = c;
      }
      if (S == D - 1) {
        LDLTX_T(16 + 8 * k + 7);
        if (k + 1 < G.Tp) factor(k + 1);
      }
#pragma unroll
      for (int s1 = S + 1; s1 < D; s1++) {         // tile (i, j), k < i < j: -R of (k, i) from the L2, W of (k, j) from registers
        spin(fpa[s1], 6, k);
        acquire();
        const double* const pa = Pan + offa[s1] + lane;
        double oa[4];
#pragma unroll
        for (int q = 0; q < 4; q++) oa[q] = ld_l2(pa + q * 64);
        d4 c = acc[s1];
#pragma unroll
        for (int q = 0; q < 4; q++) c = mfma(oa[q], w4[q], c);
        acc[s1] = c;
      }
      const int J = 16 * j + lc;
      if (J <= cb) {
#pragma unroll
        for (int g = 0; g < 4; g++) wm_store(16 * k + lr + 4 * g, J, w4[g]);
      }
#pragma unroll
      for (int s = 0; s < kNS; s++) { fpa[s] += row_words; offa[s] += row_doubles; }
    };
    if (!on[D - 1] && on[D] && ti[D] == 0 && G.Tp > 0) factor(0);      // column 0: the first pivots wait for nothing
    chain_step(std::integral_constant<int, 0>{});
    chain_step(std::integral_constant<int, 1>{});
    chain_step(std::integral_constant<int, 2>{});
  }
  // ---- every wavefront reports; participant 0 back-substitutes  L^T x = y.  Wavefront w owns rows 64 w .. 64 w + 63 (one per
  // lane): it streams its rows of the factor's columns from the L2, four columns a group from the last column down, kDepth groups
  // in flight; for the columns right of its block it subtracts W(I, J) x_J with x_J read from LDS, then it solves its own block
  // column by column (v_readlane broadcast, ldltm::k_ldlt_mfma's loop restricted to one row group) and posts each group's four
  // x_J.  The dependent chain is one readlane + one FMA per column and moves from wavefront to wavefront at block boundaries.
  publish(kFWave + gw, false);
#ifdef LDLTX_PROFILE
  LDLTX_T(600 + gw);
#endif
  if (rank != 0 || wv >= kNY || 64 * wv >= n_pad) return;
  for (;;) {
    const unsigned f = lane < kW ? ld_flag(&flags[kFWave + lane]) : epoch, f2 = lane + 64 < kW ? ld_flag(&flags[kFWave + 64 + lane]) : epoch;
    if (__builtin_amdgcn_ballot_w64(f != epoch || f2 != epoch) == 0 || bail_) break;
    __builtin_amdgcn_s_sleep(1);
    LDLTX_DOG(4, 0, 0);
  }
  acquire();
  const int ok = ld_flag(flags + kFBad) != epoch;
  const bool timed_out = ld_flag(flags + kFDog) == epoch || bail_;
  if (wv == 0) LDLTX_T(2);
  if (ok) {
    const int I = 64 * wv + lane, lo = 64 * wv;
    // (a column's start is wave-uniform and moves by J - 1 doubles from column J to J - 1: two scalar instructions and one load
    // per value; entries on or below the diagonal are masked where the value is USED -- masked at the load, the select drags the
    // wait for the load up to it)
    const unsigned Iu = (unsigned)I;
    auto col_ptr = [&](int J) -> const double* { return Wg + (size_t)(J * (J - 1) / 2); };
    double y = I < n_pad ? ld_l2(col_ptr(cb) + Iu) : 0.0;
    // Columns in groups of four from the last one down: group g = columns n_pad - 4 g - 4 .. n_pad - 4 g - 1.
    const int hi = min(n_pad, lo + 64);
    const int g_off = (n_pad - hi) >> 2;            // groups right of this block
    const int ng_own = (hi - lo) >> 2;
    // this block's own columns: in registers before anything else (the block's solve then is the bare readlane / FMA chain;
    // streamed it ran at 130 cycles a column: branches, address arithmetic, waits)
    double wown[64];
    {
      const double* cp = col_ptr(lo);
#pragma unroll
      for (int c = 0; c < 64; c++) { wown[c] = ld_l2(cp + Iu); cp += lo + c; }
    }
    // ---- the columns right of this block, eight a batch: one look at the count of posted groups, two 32-byte reads of x from
    // LDS, eight FMAs; the next batch's loads in flight meanwhile
    constexpr int kB = 2;                           // groups per batch
    double bufA[4 * kB], bufB[4 * kB];
    int Jld = n_pad - 1;                            // next column to load, going down
    int avail = 0;                                  // posted groups, as last read (a lagging wavefront looks once per several batches)
    const double* pld = col_ptr(Jld);
    auto load_batch = [&](double (&b)[4 * kB]) {
#pragma unroll
      for (int q = 0; q < kB; q++)
#pragma unroll
        for (int c = 3; c >= 0; c--) { b[4 * q + c] = ld_l2(pld + Iu); Jld--; pld -= Jld; }
    };
    auto run_batch = [&](const double (&b)[4 * kB], int gs) {
      if (gs >= g_off) return;
      const int need = min(gs + kB, g_off);
      // (no fences: a workgroup-scope fence also waits for the factor loads in flight; LDS operations of a wavefront execute in
      // order, which is all the ordering the count and the values need)
      while (avail < need && !bail_) { avail = __hip_atomic_load(&s_xready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); LDLTX_DOG(2, gs, 0); }
      asm volatile("" ::: "memory");
      double ya = 0.0, yb = 0.0;
#pragma unroll
      for (int q = 0; q < kB; q++) {
        if (gs + q < g_off) {
          const int J0 = n_pad - 4 - 4 * (gs + q);
          const d4 xv = *reinterpret_cast<const d4*>(&s_x[J0]);
          ya += b[4 * q + 3] * xv[3]; yb += b[4 * q + 2] * xv[2];       // (every row of this block is above these columns: no mask)
          ya += b[4 * q + 1] * xv[1]; yb += b[4 * q + 0] * xv[0];
        }
      }
      y -= ya + yb;
    };
    load_batch(bufA);
    load_batch(bufB);
    LDLTX_T(300 + 4 * wv);
    for (int g0 = 0; g0 < g_off; g0 += 2 * kB) {
      run_batch(bufA, g0);
      load_batch(bufA);
      run_batch(bufB, g0 + kB);
      load_batch(bufB);
    }
    // ---- this block: one readlane + one FMA per column; the four x_J of a group are posted at once
    LDLTX_T(300 + 4 * wv + 1);
#ifdef LDLTX_PROFILE
    if (lane == 0) g_xprof[330 + wv] = clock64();
#endif
#pragma unroll
    for (int q = 15; q >= 0; q--) {
      if (q < ng_own) {
#pragma unroll
        for (int c = 3; c >= 0; c--) y -= (lane < 4 * q + c ? wown[4 * q + c] : 0.0) * rdlane(y, 4 * q + c);
        if ((lane >> 2) == q) s_x[lo + lane] = y;
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(&s_xready, g_off + ng_own - q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    LDLTX_T(300 + 4 * wv + 2);
#ifdef LDLTX_PROFILE
    if (lane == 0) g_xprof[330 + wv] = clock64() - g_xprof[330 + wv];
#endif
    if (I < n) x[I] = y;
  }
  if (wv == 0) { LDLTX_T(3); if (lane == 0) *ok_flag = timed_out ? kOkTimedOut : ok; }
}

// Host side: scratch + flags of one user (an lba handle); the epoch advances with every launch.  The memory is the caller's
// (bind) or the context's own (ensure: the micro-benchmark).
// What the scratch may hold when a context starts on it: the flags must be zero (stale launch numbers would pass for published
// tiles); the G / D^-1 pair region (kGbOff .. kWOff) may hold anything -- a pair is accepted only under this context's tag, a hash of
// (nonce, launch number), and the nonce is fresh per bind / ensure and per wrap of the 24-bit launch counter, so pairs left by a
// destroyed handle in a recycled allocation, or by this context 16 M launches ago, fail the test like any two unrelated words
// (2^-63).  lba.hip zeroes the region next to the flags all the same.
__host__ inline unsigned long long fresh_nonce() {
  static std::atomic<unsigned long long> counter{0};
  timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  unsigned long long z = (counter.fetch_add(1) + 1) * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec) ^ ((unsigned long long)getpid() << 40);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;      // splitmix64 finaliser
  return z ^ (z >> 31);
}
struct Context {
  double* scr = nullptr; unsigned* flags = nullptr; unsigned epoch = 0; int plan_n = -1; Plan plan; bool owned = false;
  int pick = 0;                             // which XCD's blocks take part (0 .. 7)
  unsigned long long nonce = 0;
  void bind(double* scratch /* scratch_doubles() */, unsigned* zeroed_flags /* kFlagWords */) { scr = scratch; flags = zeroed_flags; owned = false; epoch = 0; nonce = fresh_nonce(); }
  hipError_t ensure() {
    if (scr) return hipSuccess;
    hipError_t e = hipMalloc((void**)&scr, scratch_doubles() * sizeof(double));
    if (e != hipSuccess) return e;
    if ((e = hipMalloc((void**)&flags, kFlagWords * sizeof(unsigned))) != hipSuccess) return e;
    if ((e = hipMemset(flags, 0, kFlagWords * sizeof(unsigned))) != hipSuccess) return e;
    owned = true;
    nonce = fresh_nonce();
    return hipDeviceSynchronize();
  }
  void release() { if (owned) { if (scr) (void)hipFree(scr); if (flags) (void)hipFree(flags); } scr = nullptr; flags = nullptr; }
};
// one_short (test hook): the grid ends one participant early, as if the dispatcher never placed it -- every wait of the others runs
// into its bound and the launch reports kOkTimedOut
__host__ inline hipError_t launch(Context& c, int n, const double* St, double* x, int* ok, hipStream_t st, int np = 8, bool force_safe = false, bool one_short = false) {
  hipError_t e = c.ensure();
  if (e != hipSuccess) return e;
  const int ns = 4;
  if (np != kMaxP || !plan_fits(n, np, ns)) return hipErrorInvalidValue;
  if (c.plan_n != n) { c.plan = make_plan(n, np, ns); c.plan_n = n; }
  c.plan.force_safe = force_safe;
  c.plan.pick = c.pick & 7;
  c.epoch = (c.epoch + 1) & 0xFFFFFFu;
  if (c.epoch == 0) {                        // once per 16 M launches: stale flags of the same epoch value must not survive the wrap
    if ((e = hipMemsetAsync(c.flags, 0, kFlagWords * sizeof(unsigned), st)) != hipSuccess) return e;
    c.epoch = 1;
    c.nonce = fresh_nonce();                 // ... nor pairs tagged with a launch number that comes round again
  }
  c.plan.nonce = c.nonce;
  hipLaunchKernelGGL(k_ldlt_xcd, dim3(8 * (np - 1 - (one_short ? 1 : 0)) + (c.pick & 7) + 1), dim3(kThreads), 0, st, n, St, x, ok, c.scr, c.flags, c.epoch, c.plan);
  return hipGetLastError();
}

}  // namespace ldltx
