// Host-side schedule and table builders of libg3hip: pure C++ (no HIP types, no device code), so the
// same text is compiled into the library AND into the CPU sanitizer harness (tests/host_asan/, g++
// -fsanitize=address,undefined) that exercises the index arithmetic without a GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "g3hip.h"

#define G3H_LB 128   // = G3_LB: width of the diagonal block one kernel factors

static inline int64_t g3h_roundup(int64_t n, int64_t m) { return (n + m - 1) / m * m; }

static inline int g3h_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Tuning knobs of the library.  They are read from the environment ONCE, when a context is created (g3_ctx_create), and
// travel with the context: no launch path looks at the environment, so contexts on different threads never race on a
// lazily initialised static (include/g3hip.h: distinct contexts may run concurrently).  Defaults are what was measured
// on MI355X; every knob selects another schedule of the SAME arithmetic.
struct G3hTune {
  int64_t nb;             // G3_NB          panel width of the look-ahead sweeps (0: by matrix size)
  int sb;                 // G3_SB          panels per super-panel (1)
  int nb_tail;            // G3_NB_TAIL     halve the panel width while the remaining size is <= nb_tail x width (10; 0: never)
  int nb_min;             // G3_NB_MIN      narrowest panel of the taper (128)
  int nb_head;            // G3_NB_HEAD     leading panels of half / quarter width (-1: one for n <= 16384, else none)
  int side_lds;           // G3_SIDE_LDS    least LDS a bulk-stream GEMM launch asks for (54000; 0: what it needs)
  int64_t gemm_big_min;   // G3_GEMM_BIG_MIN  least number of 128 x 128 tiles for the big tile (4096)
  int64_t gemm_big_min_k; // G3_GEMM_BIG_MIN_K  ... when K >= 1024 (1024): a rank's staircase at P = 8 has 1/8 of the tiles
  int64_t trsm_thin_max;  // G3_TRSM_THIN_MAX panel rows up to which the stripe solve uses 16-row stripes (2048)
  int64_t trsm_wide_min;  // G3_TRSM_WIDE_MIN panel rows from which it uses 64-row stripes (12288; 0: never)
  int64_t trsm_split_min; // G3_TRSM_SPLIT_MIN panel rows from which the solve is split at the launch level (0: never)
  int64_t trsm_split_n;   // G3_TRSM_SPLIT_N   widest block one stripe launch solves once the split applies (512)
  int stair_max;          // G3_STAIR_MAX   row segments / B blocks per staircase launch (160; tests lower it)
  int gram_interpret;     // G3_GRAM_NOFAST   1: always the interpreted Gram kernel (A/B measurements)
  int grad_interpret;     // G3_GRAD_GENERIC  1: always the interpreted Gram-gradient kernel
  int gram_jit;           // G3_GRAM_JIT      0: never generate a Gram kernel for an expression (g3_gram_jit.hip); default 1
  int coop_max_n;         // G3_COOP_MAX_N    chains: largest padded N whose members are factored by a workgroup group in one launch (1024; 0: never)
  int coop_min_batch;     // G3_COOP_MIN_BATCH  ... and the shortest batch it is used for (200: below that the lock-step sweep is faster)
  int coop_group;         // G3_COOP_GROUP    workgroups per member of that kernel (0: by batch size, 1 ... 8)
  int probe;              // G3_PROBE         stream-placement probe of the two-stream sweeps (1; 0: off; 2: also print what it found)
};
#define G3H_STAIR_MAX 160
static inline G3hTune g3h_tune_from_env() {
  G3hTune t;
  const char* e = getenv("G3_NB");
  t.nb = e ? atoll(e) : 0;
  t.sb = g3h_env_int("G3_SB", 1);
  t.nb_tail = g3h_env_int("G3_NB_TAIL", 10);
  t.nb_min = g3h_env_int("G3_NB_MIN", 128);
  t.nb_head = g3h_env_int("G3_NB_HEAD", -1);
  t.side_lds = g3h_env_int("G3_SIDE_LDS", 54000);
  e = getenv("G3_GEMM_BIG_MIN");
  t.gemm_big_min = e ? atoll(e) : 4096;
  e = getenv("G3_GEMM_BIG_MIN_K");
  t.gemm_big_min_k = e ? atoll(e) : 1024;
  e = getenv("G3_TRSM_THIN_MAX");
  t.trsm_thin_max = e ? atoll(e) : 2048;
  e = getenv("G3_TRSM_WIDE_MIN");
  t.trsm_wide_min = e ? atoll(e) : 12288;
  e = getenv("G3_TRSM_SPLIT_MIN");
  t.trsm_split_min = e ? atoll(e) : 0;
  e = getenv("G3_TRSM_SPLIT_N");
  t.trsm_split_n = e ? atoll(e) : 512;
  const int sm = g3h_env_int("G3_STAIR_MAX", G3H_STAIR_MAX);
  t.stair_max = sm < 1 ? 1 : (sm > G3H_STAIR_MAX ? G3H_STAIR_MAX : sm);
  t.gram_interpret = g3h_env_int("G3_GRAM_NOFAST", 0) ? 1 : 0;
  t.grad_interpret = g3h_env_int("G3_GRAD_GENERIC", 0) ? 1 : 0;
  t.gram_jit = g3h_env_int("G3_GRAM_JIT", 1) ? 1 : 0;
  t.coop_max_n = g3h_env_int("G3_COOP_MAX_N", 1024);
  t.coop_min_batch = g3h_env_int("G3_COOP_MIN_BATCH", 200);
  t.coop_group = g3h_env_int("G3_COOP_GROUP", 0);
  t.probe = g3h_env_int("G3_PROBE", 1);
  return t;
}

// Panel and super-panel boundaries of the look-ahead sweep over an n x n matrix (n a multiple of 128).
//   bnd: first column of every panel, then n.  NB-wide panels while the trailing matrix is large, narrower
//        ones near the end, where the bulk stream runs out of work and the latency of the critical-path chain
//        (diagonal-block kernels, small GEMMs) is what is left -- narrow panels shorten that chain exactly as
//        they do for a small stand-alone problem (G3_NB_TAIL=0 keeps NB throughout; G3_NB_MIN the floor)
//   grp: index of the first panel of every super-panel (G consecutive panels), then the panel count
static inline void g3h_panel_bounds(int64_t n, int64_t NB, int G, int batch, const G3hTune& tune, std::vector<int64_t>* bnd, std::vector<int>* grp) {
  const int taper = tune.nb_tail;   // halve the width while the remaining size is <= taper * width
  const int wmin = tune.nb_min;
  const int64_t wlo = batch > 1 && wmin < 256 ? 256 : wmin;   // batched sweeps: work per launch matters more
  bnd->clear();
  grp->clear();
  if (NB < G3H_LB) NB = G3H_LB;
  if (G < 1) G = 1;
  // G3_NB_HEAD=h: the first panels ramp up NB / 2^h ... NB / 2 (one each): the bulk stream has nothing to do until the
  // first panel is final, a narrow one gets it started sooner.  Measured (round 3, two A/B rounds on one box): one
  // half-width panel first is worth 1.0 - 1.3 % up to n = 16384 (8192: 7.22 -> 7.14 ms, 16384: 33.0 -> 32.6), nothing at
  // 20480 and costs 0.3 - 0.4 % from 24576 on (its K = NB / 2 bulk update is a slower launch): default 1 up to 16384
  const int head = tune.nb_head >= 0 ? tune.nb_head : (n <= 16384 ? 1 : 0);
  int64_t r0 = 0;
  int hstep = head;
  while (r0 < n) {
    int64_t w = NB;
    if (hstep > 0 && n >= 8 * NB) {
      w = NB >> hstep;
      if (w < G3H_LB) w = G3H_LB;
      --hstep;
    }
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

// ---- tile raster.  The host describes the ACTIVE tiles of a launch as a list of row groups (a few
// consecutive row tiles each) with a column-tile count per group; the table travels by value in the
// kernel arguments.  This one mechanism covers dense products, lower-triangular / trapezoidal
// updates of the factorisation (column count grows with the row, `diag_off` masks the elements above
// the shifted diagonal) and the multi-GPU "staircase" (a rank's row blocks, each with its own
// width, updated by ONE launch).  Tiles that are not wanted are never launched.
// row tiles per raster group.  Measured on a 30720^2 x 1024 lower-triangular update (rocprofv3 --pmc
// FETCH_SIZE, profiles/r02_summary.md): 2 -> 66.8 TFLOP/s, 23.0 GB fetched; 4 -> 66.3, 22.6 GB;
// 8 -> 65.9, 24.7 GB; 16 -> 64.8, 28.2 GB
constexpr int GROUP_M = 4;
constexpr int G3_RASTER_MAX = 160;
constexpr int G3_DENSE_OFF = 1 << 30;
struct RasterTab {
  int ngroups;
  int diag_off;                             // element (row, col) is wanted iff col <= row + diag_off
  // per group, packed so that the search and the payload of a small launch share one cache line
  // (the table is read with dependent scalar loads at the start of every workgroup)
  struct Group {
    int prefix;                             // first virtual tile id of the group; g[ngroups].prefix = grid size
    unsigned short row0;                    // first row tile
    unsigned short nrows;                   // row tiles in the group (its tiles are ordered column-major)
  } g[G3_RASTER_MAX + 1];
  // optional row-block permutation of B: logical block s (b_nb rows) lives at physical block
  // b_blk[s] -- the gathered panel of the multi-GPU sweep arrives rank-major, not in global order
  int b_nb;                                 // 0: B is in logical order
  unsigned short b_blk[G3_RASTER_MAX];
  // B is LOWER triangular (row j of B is zero beyond column j): the column tile at n0 only needs K = n0 + BN.  The product
  // against an explicitly inverted diagonal factor, X L^-T = X V^T with V = L^-1 (multi-GPU panel solve, g3_dist.hip)
  int k_tri;
};


// ---- host side: which elements of C a launch produces
struct GemmShape {
  int kind;                 // 0 dense, 1 trapezoid (col <= row + off), 2 staircase (row segments with own widths)
  int64_t m, n;             // extent of C in elements (staircase: m = sum of segment rows, n = widest segment)
  int64_t off;              // trapezoid: diagonal offset in elements
  int nseg;                 // staircase
  const int64_t* seg_rows;
  const int64_t* seg_cols;
  int64_t b_nb;             // rows per permuted block of B (0: none)
  const int32_t* b_perm;    // physical block of logical block s
  int nperm;
  // staircase: seg_diag[s] != 0 says the LAST seg_rows[s] columns of segment s are its square diagonal block, of
  // which only the lower triangle is wanted: tiles entirely above it are not launched (the elements above the
  // diagonal inside a launched tile are still written -- the strict upper part of a diagonal block is scratch)
  const int64_t* seg_diag;
  int k_tri;                // dense only: B is lower triangular, column tile n0 stops at K = n0 + BN
};

static inline double shape_elems(const GemmShape& sh) {
  if (sh.kind == 0) return (double)sh.m * (double)sh.n;
  if (sh.kind == 2) {
    double e = 0;
    for (int s = 0; s < sh.nseg; ++s) {
      e += (double)sh.seg_rows[s] * (double)sh.seg_cols[s];
      if (sh.seg_diag && sh.seg_diag[s] && sh.seg_cols[s] >= sh.seg_rows[s])   // algorithmic count: the lower triangle only
        e -= 0.5 * (double)sh.seg_rows[s] * ((double)sh.seg_rows[s] - 1.0);
    }
    return e;
  }
  // sum_{i < m} clamp(i + off + 1, 0, n)
  int64_t i0 = sh.off < 0 ? -sh.off : 0;           // first row with a wanted element: i + off + 1 >= 1
  if (i0 > sh.m) i0 = sh.m;
  int64_t i1 = sh.n - 1 - sh.off;                  // first row that is full width
  if (i1 < i0) i1 = i0;
  if (i1 > sh.m) i1 = sh.m;
  const double cnt = (double)(i1 - i0);
  const double tri = cnt * ((double)i0 + (double)sh.off + 1.0) + 0.5 * cnt * (cnt - 1.0);
  return tri + (double)(sh.m - i1) * (double)sh.n;
}

// Build the raster table for BM x BN tiles.  Returns the grid size, or -1 when the launch needs more
// than G3_RASTER_MAX groups (staircases with very many segments: the caller splits the launch).
template <int BM, int BN>
static inline long long build_raster(const GemmShape& sh, RasterTab* tab) {
  const int64_t tiles_m = sh.m / BM;
  tab->diag_off = sh.kind == 1 ? (int)sh.off : G3_DENSE_OFF;
  int ng = 0;
  long long total = 0;
  auto push = [&](int64_t row_tile0, int64_t rows, int64_t nc) -> bool {
    if (rows <= 0 || nc <= 0) return true;
    if (ng >= G3_RASTER_MAX) return false;
    tab->g[ng].prefix = (int)total;
    tab->g[ng].row0 = (unsigned short)row_tile0;
    tab->g[ng].nrows = (unsigned short)rows;
    total += rows * nc;
    ++ng;
    return true;
  };
  if (sh.kind == 2) {
    int64_t rt = 0;
    // group height: GROUP_M row tiles unless that needs too many groups
    int64_t ngroups_min = 0;
    for (int s = 0; s < sh.nseg; ++s) ngroups_min += (sh.seg_rows[s] / BM + GROUP_M - 1) / GROUP_M;
    const int64_t gh = ngroups_min <= G3_RASTER_MAX ? GROUP_M : GROUP_M * ((ngroups_min + G3_RASTER_MAX - 1) / G3_RASTER_MAX + 1);
    for (int s = 0; s < sh.nseg; ++s) {
      const int64_t st = sh.seg_rows[s] / BM, nc = sh.seg_cols[s] / BN;
      const bool dg = sh.seg_diag && sh.seg_diag[s] && sh.seg_cols[s] >= sh.seg_rows[s];
      const int64_t dcol0 = sh.seg_cols[s] - sh.seg_rows[s];      // first column of the diagonal block
      for (int64_t t = 0; t < st; t += gh) {
        // groups never span segments; a group's row tiles are consecutive, so an empty segment in
        // between simply starts a new group
        const int64_t rows = st - t < gh ? st - t : gh;
        int64_t ncg = nc;
        if (dg) {     // columns up to the diagonal element of the group's last row
          ncg = (dcol0 + (t + rows) * BM - 1) / BN + 1;
          if (ncg > nc) ncg = nc;
        }
        if (!push(rt + t, rows, ncg)) return -1;
      }
      rt += st;
    }
  } else {
    const int64_t tiles_n = sh.n / BN;
    int64_t gh = GROUP_M;
    if ((tiles_m + gh - 1) / gh > G3_RASTER_MAX) gh = (tiles_m + G3_RASTER_MAX - 1) / G3_RASTER_MAX;
    if (gh > 65535) return -1;
    for (int64_t t = 0; t < tiles_m; t += gh) {
      const int64_t rows = tiles_m - t < gh ? tiles_m - t : gh;
      int64_t nc = tiles_n;
      if (sh.kind == 1) {
        const int64_t lim = (t + rows) * BM - 1 + sh.off;   // last wanted column of the group's last row
        nc = lim < 0 ? 0 : lim / BN + 1;
        if (nc > tiles_n) nc = tiles_n;
      }
      if (!push(t, rows, nc)) return -1;
    }
  }
  tab->ngroups = ng;
  tab->g[ng].prefix = (int)total;
  tab->g[ng].row0 = 0;
  tab->g[ng].nrows = 1;
  tab->k_tri = (sh.kind == 0 && sh.k_tri) ? 1 : 0;
  tab->b_nb = 0;
  if (sh.b_nb > 0 && sh.b_perm) {
    if (sh.nperm > G3_RASTER_MAX || sh.b_nb % BN) return -1;
    tab->b_nb = (int)sh.b_nb;
    for (int i = 0; i < sh.nperm; ++i) tab->b_blk[i] = (unsigned short)sh.b_perm[i];
  }
  return total;
}


// ---- stripe-local triangular solve: the op list of one launch (g3_gemm.hip::trsm_stripe_kernel)
constexpr int G3_TRSM_MAXOPS = 20;          // n <= 1024: 8 leaves + 12 update tiles of 128 columns
struct TrsmOps {
  int nops;
  struct Op {      // one 32 x 128 output tile
    int col;       // first column of the tile written
    int acol;      // first column of the left operand (leaf: col)
    int k;         // reduction length
    int brow, bcol;  // leaf: index of the 128 x 128 inverse block (bcol unused); update: row / column of the L block
    int leaf;
  } op[G3_TRSM_MAXOPS];
};


static inline int64_t trsm_split(int64_t n) {     // must mirror split_point() of g3_potrf.hip (same recursion shape)
  int64_t g = G3H_LB;
  while (g * 2 <= n / 4 && g < 2048) g *= 2;
  int64_t n1 = g3h_roundup(n / 2, g);
  if (n1 >= n) n1 = n - G3H_LB;
  return n1;
}

static inline void trsm_ops_rec(TrsmOps* ops, int64_t c0, int64_t n) {
  if (n == G3H_LB) {
    auto& o = ops->op[ops->nops++];
    o.col = (int)c0; o.acol = (int)c0; o.k = G3H_LB; o.brow = (int)(c0 / G3H_LB); o.bcol = 0; o.leaf = 1;
    return;
  }
  const int64_t n1 = trsm_split(n), n2 = n - n1;
  trsm_ops_rec(ops, c0, n1);
  for (int64_t t = 0; t < n2; t += G3H_LB) {       // X[:, c0+n1+t .. +128) -= X[:, c0 .. c0+n1) L[c0+n1+t .., c0 ..)^T
    auto& o = ops->op[ops->nops++];
    o.col = (int)(c0 + n1 + t); o.acol = (int)c0; o.k = (int)n1; o.brow = (int)(c0 + n1 + t); o.bcol = (int)c0; o.leaf = 0;
  }
  trsm_ops_rec(ops, c0 + n1, n2);
}


// ---- dealing of the multi-GPU driver (g3_dist.hip): which rank owns row block I of the covariance.  The work of block I
// over the sweep grows like I^2 (its trailing updates) + a few I (its column updates and panel solves), so the blocks are
// dealt FROM THE TOP in rounds of P -- every rank gets one block per round, the least loaded rank so far the heaviest
// block of the round.  Counts differ by at most one (memory, and the padding of the panel all-gathers); the ranks' total
// work is within 2 % of the mean at P = 8, nblk = 32, where the plain boustrophedon 0..P-1, P-1..0 of rounds 1-4 left
// rank 0 (it owned the top block of every period) at +10 %, and a ragged count (nblk = 33: +70 %) is dealt as well as
// a full one.  Round-robin would leave the last rank at 1.4x.  Deterministic (ties: lowest rank): every rank computes
// the same table, and g3py_amd/distributed.py::deal_blocks mirrors it line by line.
#include <algorithm>
static inline void g3h_deal(int P, int nblk, std::vector<int>* owner, int snake = 0) {
  owner->assign(nblk > 0 ? nblk : 0, 0);
  if (P < 1) return;
  if (snake) {            // G3_DIST_DEAL=snake: the boustrophedon of rounds 1-4 (A/B measurements)
    for (int I = 0; I < nblk; ++I) {
      const int r = I % (2 * P);
      (*owner)[I] = r < P ? r : 2 * P - 1 - r;
    }
    return;
  }
  std::vector<int64_t> load(P, 0);
  std::vector<int> order(P);
  for (int top = nblk - 1; top >= 0; top -= P) {
    for (int q = 0; q < P; ++q) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return load[a] < load[b]; });   // least loaded first
    for (int i = 0; i < P && top - i >= 0; ++i) {
      const int I = top - i, r = order[i];
      (*owner)[I] = r;
      load[r] += (int64_t)I * I + 6 * (int64_t)I + 1;
    }
  }
}
// position of the blocks lo .. hi in a rank-major padded all-gather of them (every rank contributes `return value`
// block slots, its own blocks in ascending order first): idx[I - lo] = owner * count + (number of the owner's earlier blocks)
static inline int g3h_gather_table(const std::vector<int>& owner, int P, int lo, int hi, std::vector<int32_t>* idx) {
  std::vector<int> cnt(P, 0), seen(P, 0);
  if (lo < 0) lo = 0;
  for (int I = lo; I <= hi; ++I) cnt[owner[I]]++;
  int c = 0;
  for (int q = 0; q < P; ++q) c = cnt[q] > c ? cnt[q] : c;
  idx->clear();
  for (int I = lo; I <= hi; ++I) {
    const int q = owner[I];
    idx->push_back(q * c + seen[q]++);
  }
  return c;
}

// ---- staircase launches of the multi-GPU sweep.  One launch describes at most G3H_STAIR_MAX row segments and
// G3H_STAIR_MAX blocks of the B operand (the raster table travels in the kernel arguments, g3_gemm.hip::RasterTab):
// a longer staircase -- N / nb > 160 row blocks -- is cut into row chunks and column chunks.
struct G3hStairChunk {
  int64_t row0, col0;              // first row / column of the chunk, relative to the staircase
  std::vector<int64_t> rows, cols; // segment rows / columns inside the chunk
  std::vector<int64_t> diag;       // != 0: the segment's trailing square block inside this chunk is its diagonal block
  int blk0, nblk;                  // blocks of B the chunk multiplies with: [blk0, blk0 + nblk) of the block table
};
// seg_rows / seg_cols: the staircase (entries multiples of 128; columns multiples of block_rows); nperm: blocks in
// the B table (columns beyond nperm * block_rows do not exist)
static inline void g3h_stair_chunks(const std::vector<int64_t>& seg_rows, const std::vector<int64_t>& seg_cols, int64_t block_rows,
                                    int nperm, std::vector<G3hStairChunk>* out, const std::vector<int64_t>* seg_diag = nullptr,
                                    int limit = G3H_STAIR_MAX) {
  out->clear();
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
        if (seg_rows[s] <= 0) c = 0;       // an empty segment asks for nothing (and must not widen the launch)
        ch.rows.push_back(seg_rows[s]);
        ch.cols.push_back(c);
        // the diagonal block (block aligned, so never cut by a chunk boundary) belongs to the chunk that holds
        // the segment's last column
        const bool dg = seg_diag && (*seg_diag)[s] && seg_cols[s] > c0 && seg_cols[s] <= c0 + cstep && c >= seg_rows[s];
        ch.diag.push_back(dg ? 1 : 0);
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


// ---- CholeskyRobust's jitter schedule (g3py/libs/tensors.py:203-213): dK = mean(diag) * 1e-6 (float32 constants, as
// the reference's Theano graph has them), non-positive diagonals lifted by mean * 1e-6 - min, then up to 20 retries
// with dK *= 10.  One definition for the single-GPU, batched and multi-GPU callers.
struct G3hJitter {
  double dK, lift;
  int tries;
  G3hJitter(double diag_mean, double diag_min) : dK(diag_mean * (double)1e-6f), lift(0.0), tries(0) {
    if (diag_min <= 0.0) lift = diag_mean * (double)1e-6f - diag_min;
  }
  double value() const { return lift + dK; }        // what is added to the diagonal in this attempt
  bool usable() const { const double v = lift + dK; return v == v; }   // NaN statistics: every retry fails
  void next() { dK *= (double)10.0f; ++tries; }
  static int max_tries() { return 20; }
};

// ---- ring of kernel-program slots (pinned + device copies): which slot the next upload takes and whether the
// host must first wait for the launch that read it.  busy[] / next / last live in the context.
static inline int g3h_ring_take(bool* busy, int nslots, int* next, int* last, bool* must_wait, int* mark_busy) {
  // the launch that read the previous slot is already enqueued: an event recorded now sits behind it
  *mark_busy = *last;                     // slot to record an event for (-1: none)
  if (*last >= 0) busy[*last] = true;
  const int s = *next;
  *next = (s + 1) % nslots;
  *must_wait = busy[s];
  busy[s] = false;
  *last = s;
  return s;
}

// ---- Gram fast paths: recognise  var * k(x[:, 0:d]) [+ pvar * COS(x[:, 0:d])] [+ Noise]  (compile-time kernels of
// g3_gram.hip).  D and the kinds are template / run-time outputs; see g3_gram.hip for the device side.
template <typename T, int D>
struct SeParams {
  T w[D];   // ARD_L2 kinds: 0.5 * rate^2 ; OU (ARD_L1): rate
  T var, noise, alpha;
  // optional second term, a periodic leaf on the same columns (kernels.py:466-487): f = 2 pi freq,
  // pr = its rate vector (SIN: rate; SM: rate^2 * 2 pi^2; COS: unused)
  T f[D];
  T pr[D];
  T pvar;
  int mul;     // 0: stationary + periodic (KernelSum), 1: stationary * periodic (KernelProd, kernels.py:225-226)
};
#define G3H_PI 3.14159265358979323846
// returns the stationary leaf's kind (-1: no match) and the periodic term's kind in *pk (-1: none)
template <typename T, int D>
static inline int g3h_match_fast(const g3_kernel_prog* p, int d, SeParams<T, D>* out, int* pk) {
  *pk = -1;
  if (d != D || p->shift != 0.0 || p->nprod < 1 || p->nprod > 3) return -1;
  int se = -1, noise = -1, per = -1;       // se / noise / per: PRODUCT indices; sel / perl: leaf indices
  int sel = -1, perl = -1, mul = 0;
  auto is_stat = [](int kd) { return kd == G3_K_SE || kd == G3_K_OU || kd == G3_K_MAT32 || kd == G3_K_MAT52 || kd == G3_K_RQ; };
  auto is_per = [](int kd) { return kd == G3_K_COS || kd == G3_K_SIN || kd == G3_K_SM; };
  for (int q = 0; q < p->nprod; ++q) {
    // (coef * var) applied once; identical to var * k only when coef == 1 (the un-scaled kernel)
    if (p->prod[q].coef != 1.0) return -1;
    if (p->prod[q].nfac == 2) {             // stationary * periodic: one product term with two factors
      const int l0 = p->prod[q].fac[0], l1 = p->prod[q].fac[1];
      if (l0 < 0 || l0 >= p->nleaf || l1 < 0 || l1 >= p->nleaf || l0 >= G3_MAXLEAF || l1 >= G3_MAXLEAF) return -1;
      const int k0 = p->leaf[l0].kind, k1 = p->leaf[l1].kind;
      if (se >= 0 || per >= 0) return -1;
      if (is_stat(k0) && is_per(k1)) { sel = l0; perl = l1; }
      else if (is_stat(k1) && is_per(k0)) { sel = l1; perl = l0; }
      else return -1;
      se = per = q;
      mul = 1;
      continue;
    }
    if (p->prod[q].nfac != 1) return -1;
    const int li = p->prod[q].fac[0];
    if (li < 0 || li >= p->nleaf || li >= G3_MAXLEAF) return -1;
    const g3_leaf& lf = p->leaf[li];
    if (is_stat(lf.kind) && se < 0) { se = q; sel = li; }
    else if (lf.kind == G3_K_NOISE && noise < 0) noise = q;
    else if (is_per(lf.kind) && per < 0) { per = q; perl = li; }
    else return -1;
  }
  if (se < 0) return -1;
  const g3_leaf& lf = p->leaf[sel];
  if (lf.ndims != D) return -1;
  for (int k = 0; k < D; ++k) {
    if (lf.dims[k] != k) return -1;
    out->w[k] = lf.kind == G3_K_OU ? (T)lf.rate[k] : (T)(0.5 * lf.rate[k] * lf.rate[k]);
    out->f[k] = T(0);
    out->pr[k] = T(0);
  }
  out->var = (T)lf.var;
  out->alpha = (T)lf.alpha;
  out->noise = T(0);
  out->pvar = T(0);
  out->mul = mul;
  if (noise >= 0) out->noise = (T)p->leaf[p->prod[noise].fac[0]].var;
  if (per >= 0) {
    const g3_leaf& pl = p->leaf[perl];
    // instantiated for the stationary kinds and widths below (compile time); everything else is interpreted
    const bool have = (lf.kind == G3_K_SE || lf.kind == G3_K_MAT32 || lf.kind == G3_K_MAT52) &&
                      (D == 1 || D == 2 || D == 4 || D == 8);
    if (!have || pl.ndims != D) return -1;
    for (int k = 0; k < D; ++k) {
      if (pl.dims[k] != k) return -1;
      out->f[k] = T(2 * G3H_PI) * (T)pl.freq[k];     // the generic path's  (2 pi * freq) * x
      out->pr[k] = pl.kind == G3_K_SM ? (T)pl.rate[k] * (T)pl.rate[k] : (T)pl.rate[k];
    }
    out->pvar = (T)pl.var;
    *pk = pl.kind;
  }
  return lf.kind;
}

// validity of a kernel program for inputs with d columns (every index the device will follow)
static inline int g3h_validate_prog(const g3_kernel_prog* p, int d) {
  if (p->nleaf < 0 || p->nleaf > G3_MAXLEAF || p->nprod < 0 || p->nprod > G3_MAXPROD) return 1;
  for (int l = 0; l < p->nleaf; ++l) {
    const g3_leaf& lf = p->leaf[l];
    if (lf.kind < 0 || lf.kind > G3_K_WN) return 1;
    if (lf.ndims < 0 || lf.ndims > G3_MAXD) return 1;
    for (int k = 0; k < lf.ndims; ++k)
      if (lf.dims[k] < 0 || lf.dims[k] >= d) return 1;
  }
  for (int q = 0; q < p->nprod; ++q) {
    if (p->prod[q].nfac < 0 || p->prod[q].nfac > G3_MAXFAC) return 1;
    for (int f = 0; f < p->prod[q].nfac; ++f)
      if (p->prod[q].fac[f] < 0 || p->prod[q].fac[f] >= p->nleaf) return 1;
  }
  return 0;
}

// ---- chains: ONE template program plus the doubles that differ per member (g3_gp_factor_batched_fields,
// g3_gp_dlogp_batched_fields).  A byte offset is valid iff it names one of the double members of g3_kernel_prog
// (shift, a leaf's var / alpha / rate / freq, a product's coef) -- never a count, a kind or an index the device follows.
static inline int g3h_field_offset_ok(int32_t off) {
  if (off < 0 || off % 8 || (size_t)off + 8 > sizeof(g3_kernel_prog)) return 0;
  if ((size_t)off == offsetof(g3_kernel_prog, shift)) return 1;
  const size_t l0 = offsetof(g3_kernel_prog, leaf), p0 = offsetof(g3_kernel_prog, prod);
  if ((size_t)off >= l0 && (size_t)off < l0 + sizeof(g3_leaf) * G3_MAXLEAF)
    return ((size_t)off - l0) % sizeof(g3_leaf) >= offsetof(g3_leaf, var);
  if ((size_t)off >= p0) return ((size_t)off - p0) % sizeof(g3_prod) == offsetof(g3_prod, coef);
  return 0;
}
