// MFMA "NT" GEMM for gfx950:  C[m x n] = alpha * A[m x k] * B[n x k]^T + beta * C
// (row-major, k contiguous in both operands).  This one kernel carries every O(N^3) part
// of the path: the SYRK/GEMM trailing updates of the blocked Cholesky (replacing dpotrf,
// g3py/libs/tensors.py:198), the panel solves against inverted diagonal blocks, the
// multi-right-hand-side triangular solves of the posterior (g3py/processes/elliptical.py:
// 81-91) and the posterior covariance / sampling products.
//
// CDNA4 mapping
//   * v_mfma_f64_16x16x4_f64 (or v_mfma_f32_16x16x4_f32): one wave owns a WM x WN patch as
//     (WM/16) x (WN/16) accumulator tiles; A and B fragments are ONE scalar per lane
//     (A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]).
//   * The reduction index may be visited in any order as long as A and B agree, so a lane
//     fetches one 16-byte LDS chunk (2 f64 / 4 f32 consecutive k) per fragment row with
//     ds_read_b128 and feeds its elements to consecutive MFMAs.
//   * LDS tiles are [row][128 bytes] with the 16-byte chunk index XOR-swizzled by
//     (row >> 1) & 7, which makes every ds_read_b128 lane group hit 64 distinct banks.
//   * Global -> LDS by LDS-DMA (global_load_lds_dwordx4, swizzle applied on the source
//     address), double-buffered, one barrier per K tile; the f64 MFMA is 64 cycles per
//     instruction, so the loop is matrix-pipe bound (ablation: register staging + ds_write
//     cost 6-9 % of the MFMA rate).
#include "g3_internal.h"
#include <stdlib.h>

#include "g3_mfma.h"
#include "g3_host.h"

#include "g3_gemm_tile.h"


template <typename T, int BM, int BN, int WM, int WN, int NSTAGE>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, ((BM / WM) * (BN / WN) >= 16 ? 1 : 2))
gemm_nt_kernel(T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb,
               int K, T alpha, T beta, const int* __restrict__ info,
               int64_t bsC, int64_t bsA, int64_t bsB, const RasterTab tab) {
  // batch member (grid.y); strides are 0 for a single product
  C += (int64_t)blockIdx.y * bsC;
  A += (int64_t)blockIdx.y * bsA;
  B += (int64_t)blockIdx.y * bsB;
  if (info != nullptr) info += blockIdx.y;

  // ---- tile assignment: XCD-aware, grouped raster over the ACTIVE tiles only.
  // Workgroups are dealt round-robin over the 8 XCDs, so ids {x, x+8, ...} share an L2;
  // remap so that each XCD walks a contiguous range of "virtual" ids, and order virtual
  // ids group by group, column-major inside a group: the ~32 tiles an XCD works on at any
  // time then form a (group height) x 8 patch that shares 4 A panels and 8 B panels through
  // its L2 instead of streaming 32 distinct B panels from HBM.
  int bm, bn;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    int lo = 0, hi = tab.ngroups;            // largest g with prefix[g] <= v  (prefix[0] = 0, v < prefix[ngroups])
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tab.g[mid].prefix <= v) lo = mid; else hi = mid;
    }
    const int w = v - tab.g[lo].prefix;
    const int rows = (int)tab.g[lo].nrows;
    bn = w / rows;
    bm = (int)tab.g[lo].row0 + (w - bn * rows);
  }
  const int doff = tab.diag_off;
  const bool lower_only = doff < G3_DENSE_OFF;
  const int m0 = bm * BM, n0 = bn * BN;
  if (lower_only && n0 > m0 + BM - 1 + doff) return;    // a tile of a group above its own row's limit
  // a failed pivot earlier in the sweep turns every later launch into a no-op; the flag is requested
  // now and tested only before the first store, so its round trip hides under the first DMA
  const int failed = (info != nullptr) ? *info : 0;
  int64_t brow = n0;
  if (tab.b_nb > 0) {
    const int sblk = n0 / tab.b_nb;
    brow = (int64_t)tab.b_blk[sblk] * tab.b_nb + (n0 - sblk * tab.b_nb);
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // B lower triangular (k_tri): row j of B is zero beyond column j, so this column tile's reduction ends at n0 + BN
  const int Kt = (tab.k_tri && n0 + BN < K) ? n0 + BN : K;
  gemm_tile<T, BM, BN, WM, WN, NSTAGE>(C, ldc, A + (int64_t)m0 * lda, lda, B + brow * ldb, ldb, Kt, alpha, beta, m0, n0,
                                       lower_only, doff, failed, smem);
}

// ---- stripe-local triangular solve: X <- X L^-T for a tall panel X (m x n, n <= 1024) in ONE launch.
// The solve is independent row by row, so a workgroup owns a 32-row stripe of X and runs the whole
// recursion of blocked products on it, in place:  leaf  X_j <- X_j W_j^T  (W_j = inverse of the j-th
// 128 x 128 diagonal block; one 32 x 128 tile spans the block, so its rows are read before they are
// overwritten)  and  update  X_hi -= X_lo L_hi,lo^T.  Each step reads what earlier steps of the SAME
// workgroup wrote; the barrier between steps carries workgroup-scope release / acquire.  Replaces the
// 2 n/128 - 1 launches of the recursive solve on the factorisation's critical path.
template <typename T, int BM>
__global__ void __launch_bounds__(256, 2)
trsm_stripe_kernel(T* X, int64_t ldx, const T* L, int64_t ldl, const T* W, const int* __restrict__ info,
                   int64_t bsX, int64_t bsL, int64_t bsW, const TrsmOps ops) {
  X += (int64_t)blockIdx.y * bsX;
  L += (int64_t)blockIdx.y * bsL;
  W += (int64_t)blockIdx.y * bsW;
  if (info != nullptr) info += blockIdx.y;
  const int failed = (info != nullptr) ? *info : 0;
  if (failed != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int m0 = blockIdx.x * BM;
  for (int i = 0; i < ops.nops; ++i) {
    const auto o = ops.op[i];
    const T* gB = o.leaf ? W + (int64_t)o.brow * G3_LB * G3_LB : L + (int64_t)o.brow * ldl + o.bcol;
    gemm_tile<T, BM, 128, BM, 32, STAGES>(X, ldx, X + (int64_t)m0 * ldx + o.acol, ldx, gB, o.leaf ? (int64_t)G3_LB : ldl, o.k,
                                          o.leaf ? T(1) : T(-1), o.leaf ? T(0) : T(1), m0, o.col, false, G3_DENSE_OFF, 0, smem);
    __syncthreads();          // this tile's stores are visible to the workgroup's next step
  }
}


template <typename T, int BM, int BN, int WM, int WN, int NSTAGE>
static int launch_cfg(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                      int64_t ldb, int64_t k, double alpha, double beta, const GemmShape& sh) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr int LDS = NSTAGE * (BM + BN) * ROWB;
  auto kern = gemm_nt_kernel<T, BM, BN, WM, WN, NSTAGE>;
  // the attribute belongs to the (function, device) pair: cached per device.  Setting it twice from
  // two threads is harmless (same value), so the flag needs no lock.
  static bool attr_set[G3_MAX_DEVICES] = {};
  const int dev_slot = ctx->device & (G3_MAX_DEVICES - 1);
  if (!attr_set[dev_slot]) {
    G3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[dev_slot] = true;
  }
  RasterTab tab;
  const long long nv = build_raster<BM, BN>(sh, &tab);
  if (nv < 0) {
    snprintf(ctx->err, sizeof(ctx->err), "GEMM raster needs more than %d row groups", G3_RASTER_MAX);
    return G3_ERR_HIP;
  }
  if (nv == 0) return G3_OK;
  dim3 grid((unsigned)nv, (unsigned)g3_nbatch(ctx));
  // Launches on the bulk stream ask for at least 54 000 B of LDS: the 64 x 64 tile (32 KiB, five workgroups
  // per CU otherwise) is then resident three per CU, which leaves the LDS and registers a critical-path
  // kernel needs free on every CU instead of making it wait for several bulk workgroups to retire together.
  // The bulk stream has slack wherever the small tile is used.  A/B on one box, N = 4096 ... 32768:
  // -0.3 ... -3.5 % per step (12288: 19.4 -> 18.7 ms), never slower.  G3_SIDE_LDS overrides (0 = off).
  int lds_req = LDS;
  if (ctx->tune.side_lds > LDS && (g3_on_bulk_stream(ctx) || ctx->bulk_role)) lds_req = ctx->tune.side_lds;
  // algorithmic flops: 2 k per output element that is wanted.  Profiling
  // tag: launches of the 128 x 128 tile with >= 4096 tiles are the bulk panel updates
  // (>= 4096 tiles: the P2b launches, as in rounds 1-2 when nothing smaller used this tile; the column updates that take it
  //  since round 3 -- 1024 .. 4095 tiles -- are tagged MID so that the bulk figure stays comparable)
  const int tag = (BM == 128 && BN == 128) ? (nv >= 4096 ? G3_TAG_GEMM_BIG : G3_TAG_GEMM_MID) : G3_TAG_GEMM_SMALL;
  const double kmean = sh.k_tri ? 0.5 * ((double)k + (double)BN) : (double)k;     // triangular B: K grows with the column tile
  const int pr = g3i_prof_begin(ctx, tag, 2.0 * shape_elems(sh) * kmean);
  if (FILE* lg = ctx->gemm_log)   // G3_GEMM_LOG=<file>: one line per launch, joined with a kernel trace by scripts/launch_table.py
    fprintf(lg, "gemm %d %d %d %lld %lld %lld %d %lld %.9e %d\n", BM, BN, NT / 64, (long long)sh.m, (long long)sh.n, (long long)k, sh.kind,
            nv, 2.0 * shape_elems(sh) * kmean, g3_on_bulk_stream(ctx) ? 1 : 0);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds_req, ctx->stream, (T*)C, ldc, (const T*)A, lda,
                     (const T*)B, ldb, (int)k, (T)alpha, (T)beta, ctx->d_info,
                     g3_bstride_of(ctx, C), g3_bstride_of(ctx, A), g3_bstride_of(ctx, B), tab);
  g3i_prof_end(ctx, pr);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

template <typename T>
static int launch_t(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                    int64_t ldb, int64_t k, double alpha, double beta, const GemmShape& sh, int wide) {
  // tile choice: big tiles when they still fill the chip, small tiles for the narrow
  // panel / leaf operations on the critical path of the factorisation
  const int64_t blocks128 = (int64_t)(shape_elems(sh) / (128.0 * 128.0)) * g3_nbatch(ctx);
  bool all128 = sh.m % 128 == 0 && sh.n % 128 == 0;
  if (sh.kind == 2)
    for (int s = 0; s < sh.nseg; ++s) all128 = all128 && sh.seg_rows[s] % 128 == 0 && sh.seg_cols[s] % 128 == 0;
  // Tile choice (measured on MI355X, scripts/gemm_bench.py):
  //  * in-place panel solves (`wide`, C aliases A, n = 128): thin 32 x 128 tiles always -- one
  //    tile must span the 128 output columns, and 4x more workgroups beat 128 x 128 tiles from
  //    m = 1024 (13 vs 32 us) to m = 31744 (34 vs 45 us);
  //  * >= big_tile_min() tiles of 128 x 128: the big tile (two blocks per CU, 65-68 TFLOP/s);
  //  * everything in between: 64 x 64 tiles -- finer work quanta balance better over the 256 CUs
  //    (lower SYRK 4096^2 x 1024: 51 vs 40 TFLOP/s; 2048 x 1024 x 1024: 50 vs 27).
  // (a 256 x 128 tile, one block per CU, was 4-5 % slower than 128 x 128 everywhere.)
  if (wide) {
    if (sh.kind == 0 && sh.n % 128 == 0 && sh.m % 32 == 0)
      return launch_cfg<T, 32, 128, 32, 32, STAGES>(ctx, C, ldc, A, lda, B, ldb, k, alpha, beta, sh);
    snprintf(ctx->err, sizeof(ctx->err), "in-place panel GEMM needs n %% 128 == 0 and m %% 32 == 0 (m=%lld n=%lld)",
             (long long)sh.m, (long long)sh.n);
    return G3_ERR_HIP;
  }
  // The 128 x 128 tile runs with EIGHT waves (64 x 32 each, 118 VGPRs) since round 3: two tiles per CU are then
  // four waves per SIMD instead of two, and the matrix pipe finds a ready wave more often -- stand-alone
  // 30720^2 x 1024 lower update 66.3 -> 68.0 TFLOP/s, x 2048: 68.2 -> 69.4, 16384^2 x 512: 60.2 -> 64.4; the
  // N = 32768 step 208.3 -> 203.3 ms, config 3 33.7 -> 33.4 (profiles/r03_gemm_variants.md; the four-wave tile and
  // five other shapes that were measured and rejected are in that table, not in the library any more).
  // The big tile from gemm_big_min tiles on, and from 1024 tiles on when K >= 1024 (round 3, scripts/r3_sweep2.sh: N = 32768
  // 204.0 -> 203.3 ms, config 3 33.54 -> 33.43; at K = 512 -- config 2 -- the small tile stays better: 7.25 vs 7.31 ms)
  const int64_t big_min = ctx->tune.gemm_big_min;
  if (all128 && (blocks128 >= big_min || (blocks128 >= ctx->tune.gemm_big_min_k && k >= 1024 && big_min == 4096)))
    return launch_cfg<T, 128, 128, 64, 32, STAGES>(ctx, C, ldc, A, lda, B, ldb, k, alpha, beta, sh);
  return launch_cfg<T, 64, 64, 32, 32, STAGES>(ctx, C, ldc, A, lda, B, ldb, k, alpha, beta, sh);
}

static int launch_dt(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                     int64_t ldb, int64_t k, double alpha, double beta, g3_dtype dt, const GemmShape& sh, int wide) {
  if (dt == G3_F64) return launch_t<double>(ctx, C, ldc, A, lda, B, ldb, k, alpha, beta, sh, wide);
  return launch_t<float>(ctx, C, ldc, A, lda, B, ldb, k, alpha, beta, sh, wide);
}

int g3i_gemm_nt_ex(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                   int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                   g3_dtype dt, int lower_only, int wide) {
  if (m == 0 || n == 0) return G3_OK;
  const GemmShape sh{lower_only ? 1 : 0, m, n, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0};
  return launch_dt(ctx, C, ldc, A, lda, B, ldb, k, alpha, beta, dt, sh, wide);
}

int g3i_gemm_nt(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                g3_dtype dt, int lower_only) {
  return g3i_gemm_nt_ex(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, dt, lower_only, 0);
}

// C[m x n] = alpha * A[m x n] * V^T + beta * C with V (n x n, row-major) LOWER triangular: the panel solve X L^-T as ONE
// product against the explicitly inverted factor V = L^-1; the column tile at n0 reduces over K = n0 + BN only (half the
// flops of the dense product).  C must not alias A.
int g3i_gemm_nt_ktri(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* V, int64_t ldv, int64_t m, int64_t n,
                     double alpha, double beta, g3_dtype dt) {
  if (m == 0 || n == 0) return G3_OK;
  const GemmShape sh{0, m, n, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 1};
  return launch_dt(ctx, C, ldc, A, lda, V, ldv, n, alpha, beta, dt, sh, 0);
}

// trapezoid: only elements with col <= row + diag_off are produced (diag_off = 0: lower triangle
// of a C whose top-left corner lies on the diagonal; > 0: C starts diag_off columns left of it)
int g3i_gemm_nt_trap(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                     int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                     g3_dtype dt, int64_t diag_off) {
  if (m == 0 || n == 0) return G3_OK;
  const GemmShape sh{1, m, n, diag_off, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0};
  return launch_dt(ctx, C, ldc, A, lda, B, ldb, k, alpha, beta, dt, sh, 0);
}

// ---- stripe-local solve X <- X L^-T (X: m x n rows of a panel, n <= 1024, m % 32 == 0), one launch
template <typename T, int BM>
static int trsm_stripe_t(g3_ctx* ctx, const T* L, int64_t n, int64_t ldl, T* X, int64_t m, int64_t ldx, const T* W) {
  constexpr int LDS = STAGES * (BM + 128) * ROWB;
  auto kern = trsm_stripe_kernel<T, BM>;
  static bool attr_set[G3_MAX_DEVICES] = {};
  const int dev_slot = ctx->device & (G3_MAX_DEVICES - 1);
  if (!attr_set[dev_slot]) {
    G3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set[dev_slot] = true;
  }
  TrsmOps ops;
  ops.nops = 0;
  trsm_ops_rec(&ops, 0, n);
  const int pr = g3i_prof_begin(ctx, G3_TAG_GEMM_SMALL, (double)m * (double)n * (double)n);
  if (FILE* lg = ctx->gemm_log)
    fprintf(lg, "trsm %d 128 4 %lld %lld %lld 0 %lld %.9e %d\n", BM, (long long)m, (long long)n, (long long)n, (long long)(m / BM),
            (double)m * (double)n * (double)n, g3_on_bulk_stream(ctx) ? 1 : 0);
  hipLaunchKernelGGL(kern, dim3((unsigned)(m / BM), (unsigned)g3_nbatch(ctx)), dim3(256), LDS, ctx->stream, X, ldx, L, ldl, W,
                     ctx->d_info, g3_bstride_of(ctx, X), g3_bstride_of(ctx, L), g3_bstride_of(ctx, W), ops);
  g3i_prof_end(ctx, pr);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

// L: n x n lower (its diagonal blocks' inverses in W, 128 x 128 each), X: m x n, in place.  Returns 1 when
// the shape is outside what one launch covers (the caller recurses), 0 when done, < 0 on error.
int g3i_trsm_stripe(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, void* X, int64_t m, int64_t ldx, const void* W,
                    g3_dtype dt) {
  if (n > 1024 || n % G3_LB || m % 32 || m <= 0) return 1;
  const int64_t al = 16 / (int64_t)g3_esize(dt);
  if (ldx % al || ldl % al || (((uintptr_t)X | (uintptr_t)L | (uintptr_t)W) & 15)) return 1;
  // A stripe's workgroup is bound by the share of its CU's matrix pipes it gets beside the bulk workgroups: a
  // short panel (few stripes) is solved in 16-row stripes, twice the workgroups with half the products each.
  const int64_t thin_max = ctx->tune.trsm_thin_max;
  const bool thin = m * (int64_t)g3_nbatch(ctx) <= thin_max;
  // a tall panel is bound by the L2 -> LDS traffic of its stripes (every stripe re-reads the triangle): 64-row stripes
  // move 1.7x fewer bytes per flop.  Measured (same box, twice): N = 32768 205.8 -> 204.5 ms, 24576 93.3 -> 92.7; below
  // m ~ 12000 there are too few stripes to fill the chip and 32 rows win (N = 8192: 7.09 -> 7.40 ms with 64).
  const int64_t wide_min = ctx->tune.trsm_wide_min;
  if (wide_min > 0 && m >= wide_min && m % 64 == 0 && g3_nbatch(ctx) == 1) {
    if (dt == G3_F64) return trsm_stripe_t<double, 64>(ctx, (const double*)L, n, ldl, (double*)X, m, ldx, (const double*)W);
    return trsm_stripe_t<float, 64>(ctx, (const float*)L, n, ldl, (float*)X, m, ldx, (const float*)W);
  }
  if (dt == G3_F64)
    return thin ? trsm_stripe_t<double, 16>(ctx, (const double*)L, n, ldl, (double*)X, m, ldx, (const double*)W)
                : trsm_stripe_t<double, 32>(ctx, (const double*)L, n, ldl, (double*)X, m, ldx, (const double*)W);
  return thin ? trsm_stripe_t<float, 16>(ctx, (const float*)L, n, ldl, (float*)X, m, ldx, (const float*)W)
              : trsm_stripe_t<float, 32>(ctx, (const float*)L, n, ldl, (float*)X, m, ldx, (const float*)W);
}

// staircase: row segment s (seg_rows[s] rows, stacked) gets its first seg_cols[s] columns
int g3i_gemm_nt_stair(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                      int64_t ldb, int64_t k, const int64_t* seg_rows, const int64_t* seg_cols, int nseg,
                      double alpha, double beta, g3_dtype dt, int64_t b_nb, const int32_t* b_perm, int nperm,
                      const int64_t* seg_diag) {
  int64_t m = 0, n = 0;
  for (int s = 0; s < nseg; ++s) {
    m += seg_rows[s];
    if (seg_cols[s] > n) n = seg_cols[s];
  }
  if (m == 0 || n == 0) return G3_OK;
  const GemmShape sh{2, m, n, 0, nseg, seg_rows, seg_cols, b_nb, b_perm, nperm, seg_diag, 0};
  return launch_dt(ctx, C, ldc, A, lda, B, ldb, k, alpha, beta, dt, sh, 0);
}

extern "C" int g3_gemm_nt(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda,
                          const void* B, int64_t ldb, int64_t m, int64_t n, int64_t k,
                          double alpha, double beta, g3_dtype dt, int lower_only) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!C) return -2;
  if (!A) return -4;
  if (!B) return -6;
  const int64_t bk = ROWB / (int64_t)g3_esize(dt);
  if (m < 0 || m % 64) return -8;
  if (n < 0 || n % 64) return -9;
  if (k < 0 || k % bk) return -10;
  const int64_t al = 16 / (int64_t)g3_esize(dt);
  if (ldc < n) return -3;
  if (lda < k || lda % al) return -5;
  if (ldb < k || ldb % al) return -7;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return -4;
  if (k == 0) {
    // degenerate: C = beta * C is not needed anywhere on the path
    return -10;
  }
  int rc = g3i_reset_info(ctx);
  if (rc) return rc;
  return g3i_gemm_nt(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, dt, lower_only);
}

extern "C" int g3_gemm_nt_stair(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda,
                                const void* B, int64_t ldb, int64_t k, const int64_t* seg_rows,
                                const int64_t* seg_cols, int nseg, double alpha, double beta, g3_dtype dt,
                                int64_t b_block_rows, const int32_t* b_perm, int nperm, const int64_t* seg_diag) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!C) return -2;
  if (!A) return -4;
  if (!B) return -6;
  if (!seg_rows) return -9;
  if (!seg_cols) return -10;
  if (nseg < 0 || nseg > 4096) return -11;
  const int64_t bk = ROWB / (int64_t)g3_esize(dt);
  if (k <= 0 || k % bk) return -8;
  const int64_t al = 16 / (int64_t)g3_esize(dt);
  int64_t m = 0, n = 0;
  for (int s = 0; s < nseg; ++s) {
    if (seg_rows[s] < 0 || seg_rows[s] % 128) return -9;
    if (seg_cols[s] < 0 || seg_cols[s] % 128) return -10;
    if (seg_diag && seg_diag[s] && seg_cols[s] < seg_rows[s]) return -18;   // a diagonal block needs seg_rows[s] columns
    m += seg_rows[s];
    if (seg_cols[s] > n) n = seg_cols[s];
  }
  if (m >= 65535 * 64) return -9;
  if (b_perm) {
    if (b_block_rows <= 0 || b_block_rows % 128) return -15;
    if (nperm * b_block_rows < n || nperm > G3_RASTER_MAX) return -17;
    for (int i = 0; i < nperm; ++i)
      if (b_perm[i] < 0 || b_perm[i] > 65535) return -16;
  }
  if (ldc < n) return -3;
  if (lda < k || lda % al) return -5;
  if (ldb < k || ldb % al) return -7;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return -4;
  int rc = g3i_reset_info(ctx);
  if (rc) return rc;
  return g3i_gemm_nt_stair(ctx, C, ldc, A, lda, B, ldb, k, seg_rows, seg_cols, nseg, alpha, beta, dt,
                           b_perm ? b_block_rows : 0, b_perm, b_perm ? nperm : 0, seg_diag);
}
