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

// ---------------------------------------------------------------------------------------
// 64 x 64 leaf: FACTOR = true  : A <- chol(A) (lower, strict upper zeroed), W <- inv(L)
//               FACTOR = false : A holds L already;                         W <- inv(L)
// 256 threads as a 16 x 16 grid, each owning a 4 x 4 register micro-tile of A and of W.
// Per elimination step only column j of A and row j of W travel through LDS (double
// buffered: one barrier per step).
template <typename T, bool FACTOR>
__global__ void __launch_bounds__(256)
leaf64_kernel(T* A, int64_t ld, T* W, int* info, int64_t row_base, int64_t a_stride) {
  if (FACTOR && *info != 0) return;
  A += (int64_t)blockIdx.x * a_stride;
  W += (int64_t)blockIdx.x * (G3_LEAF * G3_LEAF);
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
        dg = sqrt(p);
        rp = T(1) / dg;
      } else {
        dg = p;
        rp = T(1) / p;
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
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool below = (4 * ti + r) > j;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (FACTOR) {
            if (below && (4 * tk + c) > j) a[r][c] -= li[r] * lk[c];
          }
          if (below) w[r][c] -= li[r] * wj[c];
        }
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
      if (FACTOR) A[(int64_t)row * ld + col] = (row >= col) ? a[r][c] : T(0);
      W[row * G3_LEAF + col] = (row >= col) ? w[r][c] : T(0);
    }
}

template <typename T>
static int leaf_factor(g3_ctx* ctx, T* A, int64_t ld, T* W, int64_t row_base) {
  hipLaunchKernelGGL((leaf64_kernel<T, true>), dim3(1), dim3(256), 0, ctx->stream, A, ld, W,
                     ctx->d_info, row_base, (int64_t)0);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

static int64_t split_point(int64_t n) {
  // n is a multiple of 64 and > 64; split near the middle on a coarse power-of-two grid so
  // that large sub-problems keep tile-friendly sizes
  int64_t g = 64;
  while (g * 2 <= n / 4 && g < 2048) g *= 2;
  int64_t n1 = g3_roundup(n / 2, g);
  if (n1 >= n) n1 = n - 64;
  return n1;
}

template <typename T>
static int trsm_rec(g3_ctx* ctx, const T* L, int64_t n, int64_t ldl, T* B, int64_t m, int64_t ldb,
                    const T* W, g3_dtype dt) {
  if (n == G3_LEAF)  // B <- B W^T, in place: one tile column covers all 64 output columns
    return g3i_gemm_nt(ctx, B, ldb, B, ldb, W, G3_LEAF, m, G3_LEAF, G3_LEAF, 1.0, 0.0, dt, 0);
  const int64_t n1 = split_point(n), n2 = n - n1;
  int rc = trsm_rec<T>(ctx, L, n1, ldl, B, m, ldb, W, dt);
  if (rc) return rc;
  rc = g3i_gemm_nt(ctx, B + n1, ldb, B, ldb, L + n1 * ldl, ldl, m, n2, n1, -1.0, 1.0, dt, 0);
  if (rc) return rc;
  return trsm_rec<T>(ctx, L + n1 * ldl + n1, n2, ldl, B + n1, m, ldb, W + (n1 / G3_LEAF) * G3_LEAF * G3_LEAF, dt);
}

template <typename T>
static int potrf_rec(g3_ctx* ctx, T* A, int64_t n, int64_t ld, T* W, int64_t row_base, g3_dtype dt) {
  if (n == G3_LEAF) return leaf_factor<T>(ctx, A, ld, W, row_base);
  const int64_t n1 = split_point(n), n2 = n - n1;
  int rc = potrf_rec<T>(ctx, A, n1, ld, W, row_base, dt);
  if (rc) return rc;
  T* A21 = A + n1 * ld;
  T* A22 = A21 + n1;
  rc = trsm_rec<T>(ctx, A, n1, ld, A21, n2, ld, W, dt);
  if (rc) return rc;
  rc = g3i_gemm_nt(ctx, A22, ld, A21, ld, A21, ld, n2, n2, n1, -1.0, 1.0, dt, 1);
  if (rc) return rc;
  return potrf_rec<T>(ctx, A22, n2, ld, W + (n1 / G3_LEAF) * G3_LEAF * G3_LEAF, row_base + n1, dt);
}

int g3i_reset_info(g3_ctx* ctx) {
  G3_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
  return G3_OK;
}

int g3i_ensure_invd(g3_ctx* ctx, int64_t n, g3_dtype dt) {
  size_t need = (size_t)(n / G3_LEAF) * G3_LEAF * G3_LEAF * g3_esize(dt);
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

int g3i_potrf(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, void* invd) {
  G3_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
  if (n == 0) return G3_OK;
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

int g3i_trtri_blocks(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, g3_dtype dt, void* invd) {
  if (n == 0) return G3_OK;
  const unsigned nb = (unsigned)(n / G3_LEAF);
  const int64_t stride = G3_LEAF * (ldl + 1);
  if (dt == G3_F64)
    hipLaunchKernelGGL((leaf64_kernel<double, false>), dim3(nb), dim3(256), 0, ctx->stream,
                       (double*)L, ldl, (double*)invd, ctx->d_info, (int64_t)0, stride);
  else
    hipLaunchKernelGGL((leaf64_kernel<float, false>), dim3(nb), dim3(256), 0, ctx->stream,
                       (float*)L, ldl, (float*)invd, ctx->d_info, (int64_t)0, stride);
  G3_LAUNCH_CHECK();
  return G3_OK;
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
  if (n < 0 || n % G3_LEAF) return -3;
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
  if (n < 0 || n % G3_LEAF) return -3;
  const int64_t al = 16 / (int64_t)g3_esize(dt);
  if (ldl < n || ldl % al) return -4;
  if (!B_dev) return -5;
  if (m < 0 || m % G3_LEAF) return -6;
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
  const int64_t np = g3_roundup(n, G3_LEAF);
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
