// Host-only check of csrc/ldlt_xcd.hpp's tile plan and schedule for every system size the kernel takes (no GPU: only the __host__
// plan builder runs).  (1) every tile of the upper triangle sits in exactly one wavefront slot; a chain wavefront holds exactly the
// last kChain tiles of its column, in row order, diagonal last; the others hold at most four tiles in (row, column) order.
// (2) the schedule of the kernel (chain wavefronts: the rows above their tiles, then their chain steps; the others: tile after tile),
// replayed as a set of programs of WAIT / PUBLISH steps over the same flags, runs to the
// end from any interleaving (a fixed point over "who can advance"): no wait is for something only a later step of a waiting
// wavefront would publish.     hipcc -O1 -o ldlt_xcd_plan_check ldlt_xcd_plan_check.hip && ./ldlt_xcd_plan_check
#include <cstdio>
#include <vector>

#include "../../multi_orbslam3_amd/csrc/ldlt_xcd.hpp"

using namespace ldltx;

struct Step { bool publish; int flag; };                 // flag: diag k -> k, panel (k, j) -> 64 + k * 32 + j

int main() {
  int checked = 0, bad = 0;
  for (int n = 1; n <= 16 * (kMaxT - 1) - 4; n++) {
    if (!supports(n)) continue;
    const Geo g = make_geo(n);
    const Plan P = make_plan(n, kMaxP, 4);
    const int W = kMaxP * kWgWaves;
    std::vector<int> owner(g.ntiles, -1);
    std::vector<std::vector<Step>> prog(W);
    for (int w = 0; w < W; w++) {
      int ti[4], tj[4], cnt = 0; bool on[4];
      for (int s = 0; s < 4; s++) {
        const int t = P.tile[w][s];
        on[s] = t >= 0;
        ti[s] = tj[s] = -1;
        if (t < 0) continue;
        if (t >= g.ntiles || owner[t] >= 0) { std::printf("n=%d: tile %d of wavefront %d is out of range or assigned twice\n", n, t, w); bad++; continue; }
        owner[t] = w;
        int j = 0; while ((j + 1) * (j + 2) / 2 <= t) j++;
        ti[s] = t - j * (j + 1) / 2; tj[s] = j; cnt++;
      }
      for (int s = 4; s < kMaxNS; s++) if (P.tile[w][s] >= 0) { std::printf("n=%d: wavefront %d uses slot %d\n", n, w, s); bad++; }
      if (P.chain[w]) {
        const int j = tj[kChain - 1];
        if (j < 0 || ti[kChain - 1] != j) { std::printf("n=%d: chain wavefront %d does not end on a diagonal tile\n", n, w); bad++; continue; }
        for (int s = 0; s < kChain; s++) {
          const int want = j - (kChain - 1) + s;
          if ((want >= 0) != on[s] || (on[s] && (ti[s] != want || tj[s] != j))) { std::printf("n=%d: chain wavefront %d slot %d\n", n, w, s); bad++; }
        }
        // the kernel's program: rows above its tiles, then the chain steps
        int r_end = 1 << 20;
        for (int s = 0; s < 4; s++) if (on[s] && ti[s] < r_end) r_end = ti[s];
        for (int r = 0; r < r_end; r++)
          for (int s = 0; s < 4; s++) if (on[s]) { prog[w].push_back({false, 64 + r * 32 + ti[s]}); prog[w].push_back({false, 64 + r * 32 + tj[s]}); }
        if (!on[kChain - 2] && j == 0 && g.Tp > 0) prog[w].push_back({true, 0});
        for (int S = 0; S + 1 < kChain; S++) {
          if (!on[S]) continue;
          const int k = ti[S];
          prog[w].push_back({false, k});
          prog[w].push_back({true, 64 + k * 32 + j});
          if (S == kChain - 2 && k + 1 < g.Tp) prog[w].push_back({true, k + 1});
          for (int s1 = S + 1; s1 < kChain - 1; s1++) prog[w].push_back({false, 64 + k * 32 + ti[s1]});
        }
      } else {
        if (cnt > 4) { std::printf("n=%d: wavefront %d holds %d tiles\n", n, w, cnt); bad++; }
        for (int s = 1; s < 4; s++)
          if (on[s] && (!on[s - 1] || ti[s - 1] * 64 + tj[s - 1] >= ti[s] * 64 + tj[s])) { std::printf("n=%d: wavefront %d slots out of (row, column) order\n", n, w); bad++; }
        // the kernel's program: tile after tile -- every row above it, then G of its row, then its publication
        for (int s = 0; s < 4; s++) {
          if (!on[s]) continue;
          if (ti[s] == tj[s]) { std::printf("n=%d: wavefront %d holds diagonal tile %d without being a chain wavefront\n", n, w, ti[s]); bad++; }
          for (int r = 0; r < ti[s]; r++) { prog[w].push_back({false, 64 + r * 32 + ti[s]}); prog[w].push_back({false, 64 + r * 32 + tj[s]}); }
          prog[w].push_back({false, ti[s]});
          prog[w].push_back({true, 64 + ti[s] * 32 + tj[s]});
        }
      }
    }
    for (int t = 0; t < g.ntiles; t++) if (owner[t] < 0) { std::printf("n=%d: tile %d has no wavefront\n", n, t); bad++; }
    // replay
    std::vector<char> set(64 + 32 * 32, 0);
    std::vector<int> published(64 + 32 * 32, 0);
    std::vector<size_t> pc(W, 0);
    for (bool moved = true; moved;) {
      moved = false;
      for (int w = 0; w < W; w++)
        while (pc[w] < prog[w].size()) {
          const Step& st = prog[w][pc[w]];
          if (st.publish) { set[st.flag] = 1; published[st.flag]++; }
          else if (!set[st.flag]) break;
          pc[w]++; moved = true;
        }
    }
    for (int w = 0; w < W; w++) if (pc[w] < prog[w].size()) { std::printf("n=%d: wavefront %d is stuck at step %zu (flag %d)\n", n, w, pc[w], prog[w][pc[w]].flag); bad++; break; }
    for (int k = 0; k < g.Tp; k++) if (published[k] != 1) { std::printf("n=%d: G of tile row %d published %d times\n", n, k, published[k]); bad++; }
    for (int j = 1; j < g.T; j++) for (int k = 0; k < j; k++) if (published[64 + k * 32 + j] != 1) { std::printf("n=%d: panel (%d, %d) published %d times\n", n, k, j, published[64 + k * 32 + j]); bad++; }
    checked++;
  }
  std::printf("%d system sizes checked: %s\n", checked, bad ? "FAILED" : "ALL OK");
  return bad ? 1 : 0;
}
