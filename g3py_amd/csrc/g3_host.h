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

// ---- staircase launches of the multi-GPU sweep.  One launch describes at most G3H_STAIR_MAX row segments and
// G3H_STAIR_MAX blocks of the B operand (the raster table travels in the kernel arguments, g3_gemm.hip::RasterTab):
// a longer staircase -- N / nb > 160 row blocks -- is cut into row chunks and column chunks.
#define G3H_STAIR_MAX 160
struct G3hStairChunk {
  int64_t row0, col0;              // first row / column of the chunk, relative to the staircase
  std::vector<int64_t> rows, cols; // segment rows / columns inside the chunk
  int blk0, nblk;                  // blocks of B the chunk multiplies with: [blk0, blk0 + nblk) of the block table
};
static inline int g3h_stair_limit() {
  const int v = g3h_env_int("G3_STAIR_MAX", G3H_STAIR_MAX);   // tests lower it to exercise the chunking
  return v < 1 ? 1 : (v > G3H_STAIR_MAX ? G3H_STAIR_MAX : v);
}
// seg_rows / seg_cols: the staircase (entries multiples of 128; columns multiples of block_rows); nperm: blocks in
// the B table (columns beyond nperm * block_rows do not exist)
static inline void g3h_stair_chunks(const std::vector<int64_t>& seg_rows, const std::vector<int64_t>& seg_cols, int64_t block_rows,
                                    int nperm, std::vector<G3hStairChunk>* out) {
  out->clear();
  const int limit = g3h_stair_limit();
  const int nseg = (int)seg_rows.size();
  int64_t width = 0;
  for (int s = 0; s < nseg; ++s) width = seg_cols[s] > width ? seg_cols[s] : width;
  if (width > (int64_t)nperm * block_rows) width = (int64_t)nperm * block_rows;
  const int64_t cstep = (int64_t)limit * block_rows;
  int64_t r0 = 0;
  for (int s0 = 0; s0 < nseg; s0 += limit) {
    const int s1 = s0 + limit < nseg ? s0 + limit : nseg;
    int64_t rsum = 0;
    for (int s = s0; s < s1; ++s) rsum += seg_rows[s];
    for (int64_t c0 = 0; c0 < width && rsum > 0; c0 += cstep) {
      G3hStairChunk ch;
      ch.row0 = r0;
      ch.col0 = c0;
      int64_t cmaxw = 0;
      for (int s = s0; s < s1; ++s) {
        int64_t c = seg_cols[s] - c0;
        if (c < 0) c = 0;
        if (c > cstep) c = cstep;
        if (c0 + c > width) c = width - c0;
        ch.rows.push_back(seg_rows[s]);
        ch.cols.push_back(c);
        if (c > cmaxw && seg_rows[s] > 0) cmaxw = c;
      }
      if (cmaxw <= 0) continue;
      ch.blk0 = (int)(c0 / block_rows);
      ch.nblk = (int)((cmaxw + block_rows - 1) / block_rows);
      out->push_back(ch);
    }
    r0 += rsum;
  }
}
