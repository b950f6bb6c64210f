// Host-side schedule and table builders of libg3hip: pure C++ (no HIP types, no device code), so the
// same text is compiled into the library AND into the CPU sanitizer harness (tests/host_asan/, g++
// -fsanitize=address,undefined) that exercises the index arithmetic without a GPU.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#define G3H_LB 128   // = G3_LB: width of the diagonal block one kernel factors

static inline int64_t g3h_roundup(int64_t n, int64_t m) { return (n + m - 1) / m * m; }

static inline int g3h_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Panel and super-panel boundaries of the look-ahead sweep over an n x n matrix (n a multiple of 128).
//   bnd: first column of every panel, then n.  NB-wide panels while the trailing matrix is large, narrower
//        ones near the end, where the bulk stream runs out of work and the latency of the critical-path chain
//        (diagonal-block kernels, small GEMMs) is what is left -- narrow panels shorten that chain exactly as
//        they do for a small stand-alone problem (G3_NB_TAIL=0 keeps NB throughout; G3_NB_MIN the floor)
//   grp: index of the first panel of every super-panel (G consecutive panels), then the panel count
static inline void g3h_panel_bounds(int64_t n, int64_t NB, int G, int batch, std::vector<int64_t>* bnd, std::vector<int>* grp) {
  const int taper = g3h_env_int("G3_NB_TAIL", 10);   // halve the width while the remaining size is <= taper * width
  const int wmin = g3h_env_int("G3_NB_MIN", 128);
  const int64_t wlo = batch > 1 && wmin < 256 ? 256 : wmin;   // batched sweeps: work per launch matters more
  bnd->clear();
  grp->clear();
  if (NB < G3H_LB) NB = G3H_LB;
  if (G < 1) G = 1;
  int64_t r0 = 0;
  while (r0 < n) {
    int64_t w = NB;
    if (taper) {
      const int64_t rem = n - r0;
      while (w > wlo && rem <= (int64_t)taper * w) w /= 2;
    }
    w = g3h_roundup(w, G3H_LB);
    bnd->push_back(r0);
    r0 += w;
  }
  bnd->push_back(n);
  const int nblk = (int)bnd->size() - 1;
  for (int k = 0; k < nblk; k += G) grp->push_back(k);
  grp->push_back(nblk);
}
