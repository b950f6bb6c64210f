// Blocked Cholesky and triangular solves for gfx950.
//
// Replaces: scipy.linalg.lapack.dpotrf inside CholeskyRobust (g3py/libs/tensors.py:197-222),
// solve_lower_triangular (tensors.py:265-270; g3py/processes/gaussian.py:212) and, through
// the Cholesky factor, the LU `tsl.solve` calls of the posterior (elliptical.py:81-91).
//
// Structure (all row-major, lower triangle):
//   potrf(A[n x n])      = potrf(A11); A21 <- A21 L11^-T; A22 -= A21 A21^T; potrf(A22)
//   trsm_rlt(B, L[n x n]) = trsm(B1, L11); B2 -= B1 L21^T; trsm(B2, L22)
// Leaves are 64 x 64: ONE workgroup factors the diagonal block and, in the same sweep,
// forms its inverse W = L^-1 (forward elimination on [A | I]); every panel solve is then a
// GEMM against W^T, so all O(N^3) work runs in the MFMA GEMM of g3_gemm.hip.
#include "g3_internal.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------
// 64 x 64 leaf: FACTOR = true  : A <- chol(A) (lower, strict upper zeroed), W <- inv(L)
//               FACTOR = false : A holds L already;                         W <- inv(L)
// W is written at stride ldw and, transposed, into Wt at stride ldwt.
// 256 threads as a 16 x 16 grid, each owning a 4 x 4 register micro-tile of A and of W.
// Per elimination step only column j of A and row j of W travel through LDS (double
// buffered: one barrier per step).
__device__ __forceinline__ double fast_rsqrt(double p) {
  double y = __builtin_amdgcn_rsq(p);
  y = y * fma(-0.5 * p * y, y, 1.5);
  y = y * fma(-0.5 * p * y, y, 1.5);
  return y;
}
__device__ __forceinline__ float fast_rsqrt(float p) {
  float y = __builtin_amdgcn_rsqf(p);
  y = y * fmaf(-0.5f * p * y, y, 1.5f);
  return y;
}
__device__ __forceinline__ double fast_rcp(double p) {
  double y = __builtin_amdgcn_rcp(p);
  y = y * fma(-p, y, 2.0);
  y = y * fma(-p, y, 2.0);
  return y;
}
__device__ __forceinline__ float fast_rcp(float p) {
  float y = __builtin_amdgcn_rcpf(p);
  y = y * fmaf(-p, y, 2.0f);
  return y;
}

template <typename T, bool FACTOR>
__global__ void __launch_bounds__(256)
leaf64_kernel(T* A, int64_t ld, int64_t a_stride, T* W, int64_t ldw, T* Wt, int64_t ldwt,
              int64_t w_stride, int* info, int64_t row_base) {
  if (FACTOR && *info != 0) return;
  A += (int64_t)blockIdx.x * a_stride;
  W += (int64_t)blockIdx.x * w_stride;
  Wt += (int64_t)blockIdx.x * w_stride;
  __shared__ T colbuf[2][G3_LEAF];
  __shared__ T rowbuf[2][G3_LEAF];
  const int tid = threadIdx.x, ti = tid >> 4, tk = tid & 15;
  T a[4][4], w[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = 4 * ti + r, col = 4 * tk + c;
      a[r][c] = (row >= col) ? A[(int64_t)row * ld + col] : T(0);
      w[r][c] = (row == col) ? T(1) : T(0);
    }
  for (int jj = 0; jj < 16; ++jj) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int j = 4 * jj + s;
      const int buf = s & 1;
      if (tk == jj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) colbuf[buf][4 * ti + r] = a[r][s];
      }
      if (ti == jj) {
#pragma unroll
        for (int c = 0; c < 4; ++c) rowbuf[buf][4 * tk + c] = w[s][c];
      }
      __syncthreads();
      T p = colbuf[buf][j];
      T rp, dg;
      if (FACTOR) {
        if (!(p > T(0))) {  // also catches NaN
          if (tid == 0) atomicCAS(info, 0, (int)(row_base + (int64_t)blockIdx.x * G3_LEAF + j + 1));
          p = T(1);
        }
        // 1/sqrt(p) from the hardware estimate + two Newton steps (full precision, no fp64
        // divide / sqrt sequences on the critical path); sqrt(p) = p * rp with one correction
        rp = fast_rsqrt(p);
        dg = p * rp;
        dg = fma(T(0.5) * rp, fma(-dg, dg, p), dg);
      } else {
        dg = p;
        rp = fast_rcp(p);
      }
      T li[4], lk[4], wj[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        li[r] = colbuf[buf][4 * ti + r];
        lk[r] = colbuf[buf][4 * tk + r];
        wj[r] = rowbuf[buf][4 * tk + r] * rp;
        if (FACTOR) {
          li[r] *= rp;
          lk[r] *= rp;
        }
      }
      // masked operands instead of per-element predicates: rows <= j and columns <= j of the
      // trailing update contribute exactly zero
      T lim[4], lkm[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lim[r] = ((4 * ti + r) > j) ? li[r] : T(0);
        lkm[r] = ((4 * tk + r) > j) ? lk[r] : T(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (FACTOR) a[r][c] = fma(-lim[r], lkm[c], a[r][c]);
          w[r][c] = fma(-lim[r], wj[c], w[r][c]);
        }
      if (FACTOR && tk == jj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * ti + r;
          if (row > j) a[r][s] = li[r];
          else if (row == j) a[r][s] = dg;
        }
      }
      if (ti == jj) {
#pragma unroll
        for (int c = 0; c < 4; ++c) w[s][c] = wj[c];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = 4 * ti + r, col = 4 * tk + c;
      const T wv = (row >= col) ? w[r][c] : T(0);
      if (FACTOR) A[(int64_t)row * ld + col] = (row >= col) ? a[r][c] : T(0);
      W[(int64_t)row * ldw + col] = wv;
      Wt[(int64_t)col * ldwt + row] = wv;
    }
}

constexpr int64_t LB = G3_LB;
constexpr int64_t LF = G3_LEAF;

static int64_t split_point(int64_t n, int64_t unit) {
  // n is a multiple of `unit` and > unit; split near the middle on a coarse power-of-two
  // grid so that large sub-problems keep tile-friendly sizes
  int64_t g = unit;
  while (g * 2 <= n / 4 && g < 2048) g *= 2;
  int64_t n1 = g3_roundup(n / 2, g);
  if (n1 >= n) n1 = n - unit;
  return n1;
}

// ---- inside one LB x LB diagonal block: 64-wide recursion against the 64 x 64 inverses that
// sit on the diagonal of that block's W (stride LB)
template <typename T>
static int trsm_rec64(g3_ctx* ctx, const T* L, int64_t n, int64_t ldl, T* B, int64_t m, int64_t ldb,
                      const T* W, g3_dtype dt) {
  if (n == LF)  // B <- B W^T in place: the 64 x 64 tile spans all 64 output columns
    return g3i_gemm_nt(ctx, B, ldb, B, ldb, W, LB, m, LF, LF, 1.0, 0.0, dt, 0);
  const int64_t n1 = split_point(n, LF), n2 = n - n1;
  int rc = trsm_rec64<T>(ctx, L, n1, ldl, B, m, ldb, W, dt);
  if (rc) return rc;
  rc = g3i_gemm_nt(ctx, B + n1, ldb, B, ldb, L + n1 * ldl, ldl, m, n2, n1, -1.0, 1.0, dt, 0);
  if (rc) return rc;
  return trsm_rec64<T>(ctx, L + n1 * ldl + n1, n2, ldl, B + n1, m, ldb, W + n1 * (LB + 1), dt);
}

template <typename T>
static int potrf_rec64(g3_ctx* ctx, T* A, int64_t n, int64_t ld, T* W, T* Wt, int64_t row_base, g3_dtype dt) {
  if (n == LF) {
    const int pr = g3i_prof_begin(ctx, G3_TAG_LEAF, 64.0 * 64.0 * 64.0 / 3.0);
    hipLaunchKernelGGL((leaf64_kernel<T, true>), dim3(1), dim3(256), 0, ctx->stream, A, ld, (int64_t)0, W, LB,
                       Wt, LB, (int64_t)0, ctx->d_info, row_base);
    g3i_prof_end(ctx, pr);
    G3_LAUNCH_CHECK();
    return G3_OK;
  }
  const int64_t n1 = split_point(n, LF), n2 = n - n1;
  int rc = potrf_rec64<T>(ctx, A, n1, ld, W, Wt, row_base, dt);
  if (rc) return rc;
  T* A21 = A + n1 * ld;
  T* A22 = A21 + n1;
  rc = trsm_rec64<T>(ctx, A, n1, ld, A21, n2, ld, W, dt);
  if (rc) return rc;
  rc = g3i_gemm_nt(ctx, A22, ld, A21, ld, A21, ld, n2, n2, n1, -1.0, 1.0, dt, 1);
  if (rc) return rc;
  return potrf_rec64<T>(ctx, A22, n2, ld, W + n1 * (LB + 1), Wt + n1 * (LB + 1), row_base + n1, dt);
}

// W (LB x LB) <- inv(L) given the 64 x 64 diagonal inverses in W and their transposes in Wt:
// inv([[L11,0],[L21,L22]]) = [[W1,0],[-W2 L21 W1, W2]], first on 128-blocks, then on the block.
template <typename T>
static int merge_inverse(g3_ctx* ctx, const T* L, int64_t ld, T* W, T* Wt, T* Tt, g3_dtype dt) {
  int rc;
  for (int64_t o = 0; o < LB; o += 128) {
    const T* L21 = L + (o + 64) * ld + o;
    T* W2 = W + (o + 64) * (LB + 1);
    rc = g3i_gemm_nt(ctx, Tt, 128, Wt + o * (LB + 1), LB, L21, ld, 64, 64, 64, 1.0, 0.0, dt, 0);   // (L21 W1)^T
    if (rc) return rc;
    rc = g3i_gemm_nt(ctx, W + (o + 64) * LB + o, LB, W2, LB, Tt, 128, 64, 64, 64, -1.0, 0.0, dt, 0);  // W21
    if (rc) return rc;
    rc = g3i_gemm_nt(ctx, Wt + o * LB + o + 64, LB, Tt, 128, W2, LB, 64, 64, 64, -1.0, 0.0, dt, 0);   // W21^T
    if (rc) return rc;
  }
  rc = g3i_gemm_nt(ctx, Tt, 128, Wt, LB, L + 128 * ld, ld, 128, 128, 128, 1.0, 0.0, dt, 0);
  if (rc) return rc;
  return g3i_gemm_nt(ctx, W + 128 * LB, LB, W + 128 * (LB + 1), LB, Tt, 128, 128, 128, 128, -1.0, 0.0, dt, 0);
}

template <typename T>
static int potrf_diag(g3_ctx* ctx, T* A, int64_t ld, T* W, int64_t row_base, g3_dtype dt) {
  T* Wt = (T*)ctx->wscr;
  T* Tt = Wt + LB * LB;
  int rc = potrf_rec64<T>(ctx, A, LB, ld, W, Wt, row_base, dt);
  if (rc) return rc;
  return merge_inverse<T>(ctx, A, ld, W, Wt, Tt, dt);
}

// ---- across LB blocks
template <typename T>
static int trsm_rec(g3_ctx* ctx, const T* L, int64_t n, int64_t ldl, T* B, int64_t m, int64_t ldb,
                    const T* W, g3_dtype dt) {
  if (n == LB) {
    // B <- B W^T with W lower triangular, in place: 128-column output tiles in DESCENDING
    // order, each reading only the columns at or left of itself (one tile spans its 128
    // output columns, so a workgroup has read its rows before it overwrites them)
    int rc = g3i_gemm_nt_ex(ctx, B + 128, ldb, B, ldb, W + 128 * LB, LB, m, 128, 256, 1.0, 0.0, dt, 0, 1);
    if (rc) return rc;
    return g3i_gemm_nt_ex(ctx, B, ldb, B, ldb, W, LB, m, 128, 128, 1.0, 0.0, dt, 0, 1);
  }
  const int64_t n1 = split_point(n, LB), n2 = n - n1;
  int rc = trsm_rec<T>(ctx, L, n1, ldl, B, m, ldb, W, dt);
  if (rc) return rc;
  rc = g3i_gemm_nt(ctx, B + n1, ldb, B, ldb, L + n1 * ldl, ldl, m, n2, n1, -1.0, 1.0, dt, 0);
  if (rc) return rc;
  return trsm_rec<T>(ctx, L + n1 * ldl + n1, n2, ldl, B + n1, m, ldb, W + (n1 / LB) * LB * LB, dt);
}

template <typename T>
static int potrf_rec(g3_ctx* ctx, T* A, int64_t n, int64_t ld, T* W, int64_t row_base, g3_dtype dt) {
  if (n == LB) return potrf_diag<T>(ctx, A, ld, W, row_base, dt);
  const int64_t n1 = split_point(n, LB), n2 = n - n1;
  int rc = potrf_rec<T>(ctx, A, n1, ld, W, row_base, dt);
  if (rc) return rc;
  T* A21 = A + n1 * ld;
  T* A22 = A21 + n1;
  rc = trsm_rec<T>(ctx, A, n1, ld, A21, n2, ld, W, dt);
  if (rc) return rc;
  rc = g3i_gemm_nt(ctx, A22, ld, A21, ld, A21, ld, n2, n2, n1, -1.0, 1.0, dt, 1);
  if (rc) return rc;
  return potrf_rec<T>(ctx, A22, n2, ld, W + (n1 / LB) * LB * LB, row_base + n1, dt);
}

int g3i_reset_info(g3_ctx* ctx) {
  G3_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
  return G3_OK;
}

int g3i_ensure_invd(g3_ctx* ctx, int64_t n, g3_dtype dt) {
  if (!ctx->wscr) {
    const size_t wb = (size_t)(LB * LB + 128 * 128) * sizeof(double);
    G3_HIP(hipMalloc(&ctx->wscr, wb));
    G3_HIP(hipMemset(ctx->wscr, 0, wb));   // the never-written blocks of Wt must read as zero
  }
  size_t need = (size_t)(n / LB) * LB * LB * g3_esize(dt);
  if (ctx->invd_bytes >= need) return G3_OK;
  if (ctx->invd) {
    G3_HIP(hipStreamSynchronize(ctx->stream));
    G3_HIP(hipFree(ctx->invd));
    ctx->invd = nullptr;
    ctx->invd_bytes = 0;
  }
  G3_HIP(hipMalloc(&ctx->invd, need));
  ctx->invd_bytes = need;
  return G3_OK;
}

int g3i_ensure_work(g3_ctx* ctx, size_t bytes) {
  if (ctx->work_bytes >= bytes) return G3_OK;
  if (ctx->work) {
    G3_HIP(hipStreamSynchronize(ctx->stream));
    G3_HIP(hipFree(ctx->work));
    ctx->work = nullptr;
    ctx->work_bytes = 0;
  }
  G3_HIP(hipMalloc(&ctx->work, bytes));
  ctx->work_bytes = bytes;
  return G3_OK;
}

static int zero_wt_scratch(g3_ctx* ctx) {
  // Wt is typed by the caller; zero the whole scratch so stale data of another dtype or of a
  // previous block never leaks into the blocks the merges read but nobody writes
  G3_HIP(hipMemsetAsync(ctx->wscr, 0, (size_t)(LB * LB + 128 * 128) * sizeof(double), ctx->stream));
  return G3_OK;
}

// ---- flat right-looking sweep over NB-wide panels with one-panel look-ahead.
// Stream A (the context's stream) carries the critical path: for panel k+1 the update of its
// own block column (U^a), the recursive factorisation of its diagonal block and its panel
// solve.  Stream B (low priority) carries the bulk trailing update with panel k (U^b), so the
// latency-bound panel work of k+1 overlaps the throughput-bound update of k.  U^b(k) is two
// launches: block column k+2 first (its completion event is what stream A waits for before
// it touches that column), then the rest.
template <typename T>
static int potrf_lookahead(g3_ctx* ctx, T* A, int64_t n, int64_t ld, T* W, int64_t NB, g3_dtype dt) {
  const int nblk = (int)((n + NB - 1) / NB);
  if (ctx->la_nev < 2 * nblk) {
    if (ctx->la_ev) {
      for (int i = 0; i < ctx->la_nev; ++i) (void)hipEventDestroy(ctx->la_ev[i]);
      free(ctx->la_ev);
    }
    ctx->la_nev = 2 * nblk;
    ctx->la_ev = (hipEvent_t*)calloc(ctx->la_nev, sizeof(hipEvent_t));
    if (!ctx->la_ev) return G3_ERR_NOMEM;
    for (int i = 0; i < ctx->la_nev; ++i) G3_HIP(hipEventCreateWithFlags(&ctx->la_ev[i], hipEventDisableTiming));
  }
  hipEvent_t* evP = ctx->la_ev;           // panel k final (stream A)
  hipEvent_t* evB = ctx->la_ev + nblk;    // block column k+2 carries the update with panel k (stream B)
  hipStream_t sA = ctx->stream, sB = ctx->side_stream;
  auto r = [&](int k) { return (int64_t)k * NB < n ? (int64_t)k * NB : n; };
  auto nbk = [&](int k) { return r(k + 1) - r(k); };
  int rc = G3_OK;
  // B must not start before everything already queued on A (Gram, memsets) is done
  G3_HIP(hipEventRecord(evB[nblk - 1], sA));
  G3_HIP(hipStreamWaitEvent(sB, evB[nblk - 1], 0));
  auto panel = [&](int k) -> int {   // D_k and P_k on stream A
    T* Akk = A + r(k) * ld + r(k);
    int e = potrf_rec<T>(ctx, Akk, nbk(k), ld, W + (r(k) / LB) * LB * LB, r(k), dt);
    if (e) return e;
    if (r(k + 1) < n) e = trsm_rec<T>(ctx, Akk, nbk(k), ld, A + r(k + 1) * ld + r(k), n - r(k + 1), ld,
                                      W + (r(k) / LB) * LB * LB, dt);
    return e;
  };
  rc = panel(0);
  if (rc) return rc;
  G3_HIP(hipEventRecord(evP[0], sA));
  for (int k = 0; k + 1 < nblk; ++k) {
    const int64_t r0 = r(k), r1 = r(k + 1), r2 = r(k + 2), r3 = r(k + 3), kk = nbk(k);
    // stream B: bulk update with panel k
    G3_HIP(hipStreamWaitEvent(sB, evP[k], 0));
    ctx->stream = sB;
    if (r2 < n) {
      rc = g3i_gemm_nt(ctx, A + r2 * ld + r2, ld, A + r2 * ld + r0, ld, A + r2 * ld + r0, ld, n - r2, r3 - r2, kk,
                       -1.0, 1.0, dt, 1);
      if (!rc && hipEventRecord(evB[k], sB) != hipSuccess) rc = G3_ERR_HIP;
      if (!rc && r3 < n)
        rc = g3i_gemm_nt(ctx, A + r3 * ld + r3, ld, A + r3 * ld + r0, ld, A + r3 * ld + r0, ld, n - r3, n - r3, kk,
                         -1.0, 1.0, dt, 1);
    } else if (hipEventRecord(evB[k], sB) != hipSuccess) {
      rc = G3_ERR_HIP;
    }
    ctx->stream = sA;
    if (rc) return rc;
    // stream A: look-ahead on block column k+1
    if (k >= 1) G3_HIP(hipStreamWaitEvent(sA, evB[k - 1], 0));
    rc = g3i_gemm_nt(ctx, A + r1 * ld + r1, ld, A + r1 * ld + r0, ld, A + r1 * ld + r0, ld, n - r1, r2 - r1, kk,
                     -1.0, 1.0, dt, 1);
    if (rc) return rc;
    rc = panel(k + 1);
    if (rc) return rc;
    G3_HIP(hipEventRecord(evP[k + 1], sA));
  }
  // join: A continues only after B has drained
  G3_HIP(hipEventRecord(evB[nblk - 1], sB));
  G3_HIP(hipStreamWaitEvent(sA, evB[nblk - 1], 0));
  return G3_OK;
}

int g3i_potrf(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, void* invd) {
  G3_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
  if (n == 0) return G3_OK;
  // strictly-upper 64-blocks of every W are read as zeros by the merges and the panel GEMMs
  G3_HIP(hipMemsetAsync(invd, 0, (size_t)(n / LB) * LB * LB * g3_esize(dt), ctx->stream));
  int rc = zero_wt_scratch(ctx);
  if (rc) return rc;
  int64_t NB = ctx->nb_lookahead;
  if (NB <= 0) {
    const char* e = getenv("G3_NB");
    NB = e ? atoll(e) : 2048;
  }
  NB = g3_roundup(NB < LB ? LB : NB, LB);
  if (n >= 3 * NB) {
    if (dt == G3_F64) return potrf_lookahead<double>(ctx, (double*)A, n, ld, (double*)invd, NB, dt);
    return potrf_lookahead<float>(ctx, (float*)A, n, ld, (float*)invd, NB, dt);
  }
  if (dt == G3_F64) return potrf_rec<double>(ctx, (double*)A, n, ld, (double*)invd, 0, dt);
  return potrf_rec<float>(ctx, (float*)A, n, ld, (float*)invd, 0, dt);
}

int g3i_trsm_rlt(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, void* B, int64_t m,
                 int64_t ldb, g3_dtype dt, const void* invd) {
  if (n == 0 || m == 0) return G3_OK;
  if (dt == G3_F64)
    return trsm_rec<double>(ctx, (const double*)L, n, ldl, (double*)B, m, ldb, (const double*)invd, dt);
  return trsm_rec<float>(ctx, (const float*)L, n, ldl, (float*)B, m, ldb, (const float*)invd, dt);
}

template <typename T>
static int trtri_t(g3_ctx* ctx, const T* L, int64_t n, int64_t ldl, T* W, g3_dtype dt) {
  T* Wt = (T*)ctx->wscr;
  T* Tt = Wt + LB * LB;
  for (int64_t b = 0; b < n / LB; ++b) {
    const T* Lb = L + b * LB * (ldl + 1);
    T* Wb = W + b * LB * LB;
    hipLaunchKernelGGL((leaf64_kernel<T, false>), dim3((unsigned)(LB / LF)), dim3(256), 0, ctx->stream,
                       const_cast<T*>(Lb), ldl, LF * (ldl + 1), Wb, LB, Wt, LB, LF * (LB + 1), ctx->d_info,
                       (int64_t)0);
    G3_LAUNCH_CHECK();
    int rc = merge_inverse<T>(ctx, Lb, ldl, Wb, Wt, Tt, dt);
    if (rc) return rc;
  }
  return G3_OK;
}

int g3i_trtri_blocks(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, g3_dtype dt, void* invd) {
  if (n == 0) return G3_OK;
  G3_HIP(hipMemsetAsync(invd, 0, (size_t)(n / LB) * LB * LB * g3_esize(dt), ctx->stream));
  int rc = zero_wt_scratch(ctx);
  if (rc) return rc;
  if (dt == G3_F64) return trtri_t<double>(ctx, (const double*)L, n, ldl, (double*)invd, dt);
  return trtri_t<float>(ctx, (const float*)L, n, ldl, (float*)invd, dt);
}

static int read_info(g3_ctx* ctx, int* info_host) {
  G3_HIP(hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  G3_HIP(hipStreamSynchronize(ctx->stream));
  *info_host = *ctx->h_info;
  return G3_OK;
}

extern "C" int g3_potrf(g3_ctx* ctx, void* A_dev, int64_t n, int64_t ld, g3_dtype dt,
                        void* invd_dev, int* info_host) {
  if (!ctx) return -1;
  if (!A_dev) return -2;
  if (n < 0 || n % G3_LB) return -3;
  if (ld < n || ld % (16 / (int64_t)g3_esize(dt))) return -4;
  if (!info_host) return -7;
  if (!invd_dev) {
    int rc = g3i_ensure_invd(ctx, n, dt);
    if (rc) return rc;
    invd_dev = ctx->invd;
  }
  int rc = g3i_potrf(ctx, A_dev, n, ld, dt, invd_dev);
  if (rc) return rc;
  return read_info(ctx, info_host);
}

extern "C" int g3_trsm_rlt(g3_ctx* ctx, const void* L_dev, int64_t n, int64_t ldl, void* B_dev,
                           int64_t m, int64_t ldb, g3_dtype dt, const void* invd_dev) {
  if (!ctx) return -1;
  if (!L_dev) return -2;
  if (n < 0 || n % G3_LB) return -3;
  const int64_t al = 16 / (int64_t)g3_esize(dt);
  if (ldl < n || ldl % al) return -4;
  if (!B_dev) return -5;
  if (m < 0 || m % 128) return -6;
  if (ldb < n || ldb % al) return -7;
  int rc0 = g3i_reset_info(ctx);
  if (rc0) return rc0;
  if (!invd_dev) {
    int rc = g3i_ensure_invd(ctx, n, dt);
    if (rc) return rc;
    rc = g3i_trtri_blocks(ctx, L_dev, n, ldl, dt, ctx->invd);
    if (rc) return rc;
    invd_dev = ctx->invd;
  }
  return g3i_trsm_rlt(ctx, L_dev, n, ldl, B_dev, m, ldb, dt, invd_dev);
}

// ---------------------------------------------------------------------------------------
// CholeskyRobust (tensors.py:197-222)
template <typename T>
__global__ void copy_lower_pad_kernel(T* L, int64_t ldl, const T* K, int64_t ldk, int64_t n,
                                      int64_t npad, T diag_add) {
  // L[i][j] = K[i][j] (+diag_add on the diagonal) for j <= i < n ; 0 above the diagonal;
  // identity in the padding rows/cols
  const int64_t i = blockIdx.y;
  for (int64_t j = blockIdx.x * blockDim.x + threadIdx.x; j < npad; j += (int64_t)gridDim.x * blockDim.x) {
    T v = T(0);
    if (i < n && j < n) {
      if (j <= i) v = K[i * ldk + j];
      if (j == i) v += diag_add;
    } else if (i == j) {
      v = T(1);
    }
    L[i * ldl + j] = v;
  }
}

template <typename T>
__global__ void set_scaled_eye_kernel(T* L, int64_t ld, int64_t n, T v) {
  const int64_t i = blockIdx.y;
  for (int64_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
    L[i * ld + j] = (i == j) ? v : T(0);
}

template <typename T>
__global__ void count_nonfinite_lower_kernel(const T* K, int64_t ld, int64_t n, unsigned long long* out) {
  const int64_t i = blockIdx.y;
  unsigned long long c = 0;
  for (int64_t j = blockIdx.x * blockDim.x + threadIdx.x; j <= i; j += (int64_t)gridDim.x * blockDim.x) {
    T v = K[i * ld + j];
    if (!(fabs((double)v) <= 1.7976931348623157e308)) ++c;
  }
  if (c) atomicAdd(out, c);
}

template <typename T>
static int robust_t(g3_ctx* ctx, const T* K, int64_t ldk, T* L, int64_t ldl, int64_t n, g3_dtype dt,
                    int maxtries, int* tries_host, int* fallback_host, double* jitter_host) {
  const int64_t np = g3_roundup(n, G3_LB);
  const bool direct = (np == n) && (ldl % (16 / (int64_t)sizeof(T)) == 0) && (((uintptr_t)L & 15) == 0);
  T* F = L;
  int64_t ldf = ldl;
  if (!direct) {
    int rc = g3i_ensure_work(ctx, (size_t)np * np * sizeof(T));
    if (rc) return rc;
    F = (T*)ctx->work;
    ldf = np;
  }
  int rc = g3i_ensure_invd(ctx, np, dt);
  if (rc) return rc;
  const dim3 grid((unsigned)((np + 255) / 256 > 64 ? 64 : (np + 255) / 256), (unsigned)np);
  auto attempt = [&](double add, int* info) -> int {
    hipLaunchKernelGGL((copy_lower_pad_kernel<T>), grid, dim3(256), 0, ctx->stream, F, ldf, K, ldk, n, np, (T)add);
    G3_LAUNCH_CHECK();
    int r = g3i_potrf(ctx, F, np, ldf, dt, ctx->invd);
    if (r) return r;
    return read_info(ctx, info);
  };
  int info = 0, tries = 0, fallback = 0;
  double jitter = 0.0;
  rc = attempt(0.0, &info);
  if (rc) return rc;
  if (info != 0) {
    double st[3];
    rc = g3_diag_stats(ctx, K, n, ldk, dt, st);
    if (rc) return rc;
    const double c6 = (double)1e-6f, c10 = (double)10.0f;
    double dK = st[1] * c6;
    double lift = 0.0;
    if (st[0] <= 0.0) lift = st[1] * c6 - st[0];
    // sp.linalg.cholesky(check_finite=True) raises on NaN/Inf input: every retry then fails
    unsigned long long* cnt = (unsigned long long*)ctx->d_stats;
    G3_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), ctx->stream));
    const dim3 g2((unsigned)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL((count_nonfinite_lower_kernel<T>), g2, dim3(256), 0, ctx->stream, K, ldk, n, cnt);
    G3_LAUNCH_CHECK();
    G3_HIP(hipMemcpyAsync(ctx->h_stats, cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    G3_HIP(hipStreamSynchronize(ctx->stream));
    const bool finite = (*(unsigned long long*)ctx->h_stats == 0) && (st[1] == st[1]) &&
                        (fabs(st[1]) <= 1.7976931348623157e308);
    bool ok = false;
    for (int t = 0; t < maxtries; ++t) {
      ++tries;
      if (finite && (lift + dK) == (lift + dK)) {
        rc = attempt(lift + dK, &info);
        if (rc) return rc;
        if (info == 0) {
          ok = true;
          jitter = lift + dK;
          break;
        }
      }
      dK *= c10;
    }
    if (!ok) {
      fallback = 1;
      hipLaunchKernelGGL((set_scaled_eye_kernel<T>), grid, dim3(256), 0, ctx->stream, F, ldf, np, (T)1e-10f);
      G3_LAUNCH_CHECK();
    }
  }
  rc = g3i_reset_info(ctx);
  if (rc) return rc;
  if (!direct) {
    rc = g3_copy2d(ctx, L, ldl, F, ldf, n, n, dt);
    if (rc) return rc;
  }
  G3_HIP(hipStreamSynchronize(ctx->stream));
  if (tries_host) *tries_host = tries;
  if (fallback_host) *fallback_host = fallback;
  if (jitter_host) *jitter_host = jitter;
  return G3_OK;
}

extern "C" int g3_potrf_robust(g3_ctx* ctx, const void* K_dev, int64_t ldk, void* L_dev, int64_t ldl,
                               int64_t n, g3_dtype dt, int maxtries, int* tries_host,
                               int* fallback_host, double* jitter_host) {
  if (!ctx) return -1;
  if (!K_dev) return -2;
  if (ldk < n) return -3;
  if (!L_dev) return -4;
  if (ldl < n) return -5;
  if (n < 0) return -6;
  if (maxtries < 0) return -8;
  if (n == 0) {
    if (tries_host) *tries_host = 0;
    if (fallback_host) *fallback_host = 0;
    if (jitter_host) *jitter_host = 0;
    return G3_OK;
  }
  if (dt == G3_F64)
    return robust_t<double>(ctx, (const double*)K_dev, ldk, (double*)L_dev, ldl, n, dt, maxtries,
                            tries_host, fallback_host, jitter_host);
  return robust_t<float>(ctx, (const float*)K_dev, ldk, (float*)L_dev, ldl, n, dt, maxtries,
                         tries_host, fallback_host, jitter_host);
}
