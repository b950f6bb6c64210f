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

constexpr int ROWB = 128;  // bytes of K per LDS row and per stage
constexpr int GROUP_M = 4; // row-blocks per raster group

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, 2)
gemm_nt_kernel(T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb,
               int K, T alpha, T beta, int lower_only, const int* __restrict__ info, int tiles_m, int tiles_n,
               int64_t bsC, int64_t bsA, int64_t bsB) {
  using M = MfmaT<T>;
  // batch member (grid.y); strides are 0 for a single product
  C += (int64_t)blockIdx.y * bsC;
  A += (int64_t)blockIdx.y * bsA;
  B += (int64_t)blockIdx.y * bsB;
  if (info != nullptr) info += blockIdx.y;
  using chunk_t = typename M::chunk_t;
  using acc_t = typename M::acc_t;
  constexpr int EPC = M::EPC;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int NWN = BN / WN;
  constexpr int NT = (BM / WM) * NWN * 64;
  constexpr int TM = WM / 16, TN = WN / 16;

  // ---- tile assignment: XCD-aware, grouped raster over the ACTIVE tiles only.
  // Workgroups are dealt round-robin over the 8 XCDs, so ids {x, x+8, ...} share an L2;
  // remap so that each XCD walks a contiguous range of "virtual" ids, and order virtual
  // ids in groups of GROUP_M row-blocks, column-major inside a group: the ~32 tiles an XCD
  // works on at any time then form a GROUP_M x 8 patch that shares 4 A panels and 8 B
  // panels through its L2 instead of streaming 32 distinct B panels from HBM.
  int bm, bn;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    // groups of GROUP_M row-blocks; group g has nc(g) active column-blocks
    int g = 0, base = 0;
    if (!lower_only) {
      const int per = GROUP_M * tiles_n;
      g = v / per;
      base = g * per;
    } else {
      // nc(g) = min(tiles_n, RATIO * GROUP_M * (g + 1)), RATIO = BM / BN column tiles per row tile;
      // quadratic prefix until saturation at g_sat, linear after
      constexpr int RATIO = (BM >= BN) ? BM / BN : 1;
      const int step = RATIO * GROUP_M;                       // nc grows by `step` per group
      const int g_sat = (tiles_n + step - 1) / step - 1;      // first group with nc == tiles_n (maybe)
      // prefix(g) = GROUP_M * step * g (g + 1) / 2 for g <= g_sat
      const long long pre_sat = (long long)GROUP_M * step * g_sat * (g_sat + 1) / 2;
      if (v < pre_sat) {
        // solve GROUP_M*step*g(g+1)/2 <= v
        const double a = 0.5 * GROUP_M * step;
        g = (int)((-a + sqrt(a * a + 4.0 * a * (double)v)) / (2.0 * a));
        while ((long long)GROUP_M * step * (g + 1) * (g + 2) / 2 <= v) ++g;
        while ((long long)GROUP_M * step * g * (g + 1) / 2 > v) --g;
        base = (int)((long long)GROUP_M * step * g * (g + 1) / 2);
      } else {
        const int per = GROUP_M * tiles_n;
        g = g_sat + (int)((v - pre_sat) / per);
        base = (int)(pre_sat + (long long)(g - g_sat) * per);
      }
    }
    const int w = v - base;                 // index inside the group, column-major
    const int rows_in_group = min(GROUP_M, tiles_m - g * GROUP_M);
    bn = w / GROUP_M;
    bm = g * GROUP_M + (w - bn * GROUP_M);
    if (rows_in_group < GROUP_M) {          // ragged last group: re-derive with its own height
      bn = w / rows_in_group;
      bm = g * GROUP_M + (w - bn * rows_in_group);
    }
    if (bm >= tiles_m || bn >= tiles_n) return;
  }
  const int m0 = bm * BM, n0 = bn * BN;
  if (lower_only && n0 >= m0 + BM) return;
  if (info != nullptr && *info != 0) return;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                      // [2][BM][ROWB]
  char* sB = smem + 2 * BM * ROWB;      // [2][BN][ROWB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;

  acc_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  // ---- staging: LDS-DMA (global_load_lds_dwordx4).  One wave-instruction moves 64 x 16 B =
  // 8 rows x 128 B straight from global memory into LDS (no VGPR round trip, no ds_write).
  // The LDS destination is linear (wave-uniform base + lane * 16), so the XOR swizzle is
  // applied to the per-lane SOURCE address: physical chunk c of row r receives logical chunk
  // c ^ ((r >> 1) & 7), which is what the fragment reads below expect.
  constexpr int NW = NT / 64;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "rows per wave-instruction");
  const T* gA = A + (int64_t)m0 * lda;
  const T* gB = B + (int64_t)n0 * ldb;
  const int sr = lane >> 3, sc = lane & 7;
  auto stage = [&](int buf, int k0) {
    char* a = sA + buf * BM * ROWB;
    char* b = sB + buf * BN * ROWB;
#pragma unroll
    for (int i = 0; i < BM / 8 / NW; ++i) {
      const int rb = (i * NW + wave) * 8, row = rb + sr;
      const T* src = gA + (int64_t)row * lda + k0 + ((sc ^ ((row >> 1) & 7)) * EPC);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(a + rb * ROWB), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BN / 8 / NW; ++i) {
      const int rb = (i * NW + wave) * 8, row = rb + sr;
      const T* src = gB + (int64_t)row * ldb + k0 + ((sc ^ ((row >> 1) & 7)) * EPC);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(b + rb * ROWB), 16, 0, 0);
    }
  };

  const int frow = lane & 15, kq = lane >> 4, swz = (frow >> 1) & 7;
  auto compute = [&](int buf, int s) {   // s = 0, 1: the two 8-deep halves of a K tile
    const char* a = sA + buf * BM * ROWB + (wm + frow) * ROWB;
    const char* b = sB + buf * BN * ROWB + (wn + frow) * ROWB;
    {
      const int off = (((s * 4 + kq) ^ swz) << 4);
      chunk_t fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const chunk_t*>(a + i * 16 * ROWB + off);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const chunk_t*>(b + j * 16 * ROWB + off);
#pragma unroll
      for (int e = 0; e < EPC; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = M::mfma(fa[i][e], fb[j][e], acc[i][j]);
    }
  };

  // two LDS buffers: the DMA of tile kt+1 is in flight during all 64 MFMAs of tile kt; the
  // vmcnt(0) + barrier at the end of an iteration both publishes tile kt+1 and guarantees
  // everyone has finished reading the buffer the next DMA will overwrite
  const int KT = K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // epilogue geometry (see below).  (Requesting the first chunk of C before the last K tile would
  // hide its latency too, but costs 25 VGPRs: the kernel must stay <= 224 so that the
  // critical-path diagonal kernel still fits beside one of these workgroups.)
  constexpr int RC = 32;                                   // rows per chunk
  constexpr int PITCH = BN * (int)sizeof(T) + 128;         // +128 B: rows r, r+1 land in different bank halves
  constexpr int VPR = BN * (int)sizeof(T) / 16;            // 16-byte vectors per row
  constexpr int RPP = NT / VPR;                            // rows per pass of the whole workgroup
  constexpr int NPASS = RC / RPP;
  constexpr int NCH = BM / RC;
  static_assert(RC * PITCH <= 2 * (BM + BN) * ROWB, "epilogue chunk must fit in the staging LDS");
  static_assert(NT % VPR == 0 && RC % RPP == 0 && BM % RC == 0, "epilogue tiling");
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (ldc % EPC == 0);
  const bool rmw = vec_ok && beta != T(0);
  const int vrow = tid / VPR, vcol = (tid % VPR) * EPC;    // this thread's row (within a pass) and first column
  chunk_t cnext[NPASS];
  auto fetch_c = [&](int c) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int row = m0 + c * RC + p * RPP + vrow, col = n0 + vcol;
      cnext[p] = chunk_t{};
      if (!(lower_only && col > row)) cnext[p] = *reinterpret_cast<const chunk_t*>(C + (int64_t)row * ldc + col);
    }
  };
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) stage(buf ^ 1, (kt + 1) * BK);
    compute(buf, 0);
    compute(buf, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue.  The accumulators are in MFMA layout (a lane holds 4 rows x 1 column of each
  // 16 x 16 tile): written straight to C that is 128-byte pieces scattered over 64 rows per
  // instruction, and with beta != 0 the same pattern is read first -- measured 6 % of a K = 1024
  // launch for the read alone.  Instead the tile goes through the (now idle) staging LDS in
  // 32-row chunks and every global access is a full row segment: 64 consecutive 16-byte vectors,
  // BN * sizeof(T) contiguous bytes per row (1 KiB for the 128-column fp64 tile).
  if (vec_ok) {
    const int col_l = lane & 15;
    if (rmw) fetch_c(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // this chunk's C vectors were requested one step ago; request the next chunk's now
      chunk_t cold[NPASS];
#pragma unroll
      for (int p = 0; p < NPASS; ++p) cold[p] = cnext[p];
      if (rmw && c + 1 < NCH) fetch_c(c + 1);
      // accumulators of the waves that own rows of this chunk -> LDS (row-major)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if ((wm + i * 16) / RC == c) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int rl = (wm + i * 16) % RC + M::row(lane, r);
              *reinterpret_cast<T*>(smem + rl * PITCH + (wn + j * 16 + col_l) * (int)sizeof(T)) = acc[i][j][r];
            }
        }
      }
      __syncthreads();
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int rl = p * RPP + vrow;
        const int row = m0 + c * RC + rl, col = n0 + vcol;
        if (lower_only && col > row) continue;
        chunk_t v = *reinterpret_cast<const chunk_t*>(smem + rl * PITCH + vcol * (int)sizeof(T));
        v = v * alpha;
        if (rmw) v = v + cold[p] * beta;
        T* dst = C + (int64_t)row * ldc + col;
        if (!lower_only || col + EPC - 1 <= row) {
          *reinterpret_cast<chunk_t*>(dst) = v;
        } else {                                           // the vector straddles the diagonal
#pragma unroll
          for (int e = 0; e < EPC; ++e)
            if (col + e <= row) dst[e] = v[e];
        }
      }
      __syncthreads();
    }
    return;
  }
  // fallback (C not 16-byte aligned): element-wise from the MFMA layout
  const int col_l = lane & 15;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + j * 16 + col_l;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + M::row(lane, r);
        if (lower_only && col > row) continue;
        T* p = C + (int64_t)row * ldc + col;
        T v = alpha * acc[i][j][r];
        if (beta != T(0)) v += beta * (*p);
        *p = v;
      }
    }
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_cfg(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                      int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                      int lower_only) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr int LDS = 2 * (BM + BN) * ROWB;
  auto kern = gemm_nt_kernel<T, BM, BN, WM, WN>;
  static bool attr_set = false;
  if (!attr_set) {
    G3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  // number of virtual tile ids (see the kernel's raster): groups of GROUP_M row-blocks, each
  // with nc(g) column-blocks; the last group may be ragged
  const int tiles_m = (int)(m / BM), tiles_n = (int)(n / BN);
  long long nv = 0;
  {
    constexpr int RATIO = (BM >= BN) ? BM / BN : 1;
    const int ngroups = (tiles_m + GROUP_M - 1) / GROUP_M;
    for (int g = 0; g < ngroups; ++g) {
      const int rows = (g == ngroups - 1) ? tiles_m - g * GROUP_M : GROUP_M;
      long long nc = tiles_n;
      if (lower_only) {
        const long long lim = (long long)RATIO * GROUP_M * (g + 1);
        if (lim < nc) nc = lim;
      }
      // the kernel's prefix formula assumes full groups except the last one
      nv += (g == ngroups - 1) ? (long long)rows * nc : (long long)GROUP_M * nc;
    }
  }
  dim3 grid((unsigned)nv, (unsigned)g3_nbatch(ctx));
  // algorithmic flops: 2 k per output element that is wanted.  Profiling
  // tag: launches of the 128 x 128 tile with >= 1024 tiles are the bulk panel updates
  const int tag = (BM == 128 && BN == 128) ? (nv >= 1024 ? G3_TAG_GEMM_BIG : G3_TAG_GEMM_MID) : G3_TAG_GEMM_SMALL;
  // lower-only: 2k flops for every element on or below the diagonal (m >= n: m n - n(n-1)/2 of them)
  const double elems = lower_only ? ((double)m * n - 0.5 * (double)n * (n - 1) - (m < n ? 0.5 * (double)(n - m) * (n - m + 1) : 0.0))
                                  : (double)m * n;
  const int pr = g3i_prof_begin(ctx, tag, 2.0 * elems * (double)k);
  hipLaunchKernelGGL(kern, grid, dim3(NT), LDS, ctx->stream, (T*)C, ldc, (const T*)A, lda,
                     (const T*)B, ldb, (int)k, (T)alpha, (T)beta, lower_only, ctx->d_info, tiles_m, tiles_n,
                     g3_bstride_of(ctx, C), g3_bstride_of(ctx, A), g3_bstride_of(ctx, B));
  g3i_prof_end(ctx, pr);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

template <typename T>
static int launch_t(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                    int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                    int lower_only, int wide) {
  // tile choice: big tiles when they still fill the chip, small tiles for the narrow
  // panel / leaf operations on the critical path of the factorisation
  const int64_t blocks128 = (m / 128) * (n / 128) / (lower_only ? 2 : 1) * g3_nbatch(ctx);
  static int forced = -1;   // G3_GEMM_CFG: development override of the tile choice
  if (forced < 0) {
    const char* e = getenv("G3_GEMM_CFG");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 1 && m % 256 == 0 && n % 128 == 0)
    return launch_cfg<T, 256, 128, 64, 64>(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, lower_only);
  if (forced == 2 && m % 128 == 0 && n % 128 == 0)
    return launch_cfg<T, 128, 128, 64, 64>(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, lower_only);
  if (forced == 3) return launch_cfg<T, 64, 64, 32, 32>(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, lower_only);
  if (forced == 4 && n % 128 == 0)
    return launch_cfg<T, 32, 128, 32, 32>(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, lower_only);
  // Tile choice (measured on MI355X, scripts/gemm_bench.py):
  //  * in-place panel solves (`wide`, C aliases A, n = 128): thin 32 x 128 tiles always -- one
  //    tile must span the 128 output columns, and 4x more workgroups beat 128 x 128 tiles from
  //    m = 1024 (13 vs 32 us) to m = 31744 (34 vs 45 us);
  //  * >= 4096 tiles of 128 x 128: the big tile (two blocks per CU, 65-68 TFLOP/s);
  //  * everything in between: 64 x 64 tiles -- finer work quanta balance better over the 256 CUs
  //    (lower SYRK 4096^2 x 1024: 51 vs 40 TFLOP/s; 2048 x 1024 x 1024: 50 vs 27).
  // (a 256 x 128 tile, one block per CU, was 4-5 % slower than 128 x 128 everywhere.)
  if (wide) {
    if (n % 128 == 0 && m % 32 == 0)
      return launch_cfg<T, 32, 128, 32, 32>(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, lower_only);
    snprintf(ctx->err, sizeof(ctx->err), "in-place panel GEMM needs n %% 128 == 0 and m %% 32 == 0 (m=%lld n=%lld)",
             (long long)m, (long long)n);
    return G3_ERR_HIP;
  }
  if (m % 128 == 0 && n % 128 == 0 && blocks128 >= 4096)
    return launch_cfg<T, 128, 128, 64, 64>(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, lower_only);
  return launch_cfg<T, 64, 64, 32, 32>(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, lower_only);
}

int g3i_gemm_nt_ex(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                   int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                   g3_dtype dt, int lower_only, int wide) {
  if (m == 0 || n == 0) return G3_OK;
  if (dt == G3_F64)
    return launch_t<double>(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, lower_only, wide);
  return launch_t<float>(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, lower_only, wide);
}

int g3i_gemm_nt(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                g3_dtype dt, int lower_only) {
  return g3i_gemm_nt_ex(ctx, C, ldc, A, lda, B, ldb, m, n, k, alpha, beta, dt, lower_only, 0);
}

extern "C" int g3_gemm_nt(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda,
                          const void* B, int64_t ldb, int64_t m, int64_t n, int64_t k,
                          double alpha, double beta, g3_dtype dt, int lower_only) {
  if (!ctx) return -1;
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
