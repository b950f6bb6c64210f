// Blocked Cholesky and triangular solves for gfx950.
//
// Replaces: scipy.linalg.lapack.dpotrf inside CholeskyRobust (g3py/libs/tensors.py:197-222),
// solve_lower_triangular (tensors.py:265-270; g3py/processes/gaussian.py:212) and, through
// the Cholesky factor, the LU `tsl.solve` calls of the posterior (elliptical.py:81-91).
//
// Structure (all row-major, lower triangle):
//   potrf(A[n x n])      = potrf(A11); A21 <- A21 L11^-T; A22 -= A21 A21^T; potrf(A22)
//   trsm_rlt(B, L[n x n]) = trsm(B1, L11); B2 -= B1 L21^T; trsm(B2, L22)
// Leaves are 128 x 128: ONE workgroup factors the diagonal block and forms its inverse
// W = L^-1 with both matrices resident in MFMA accumulators; every panel solve is then a GEMM
// against W^T, so all O(N^3) work runs on the matrix pipe (g3_gemm.hip).  Large matrices use a
// flat right-looking sweep over panels with one-panel look-ahead on two streams.
#include "g3_internal.h"
#include <vector>
#include "g3_host.h"
#include "g3_mfma.h"
#include "g3_gemm_tile.h"
#include <stdlib.h>

#include "g3_diag.h"

template <typename T>
__global__ void __launch_bounds__(512, 4)   // <= 128 VGPRs: must fit beside a bulk GEMM workgroup (2 x 128 + 224 <= 512 per SIMD)
potrf256_kernel(T* A, int64_t ld, T* W, int* info, int64_t row_base, int64_t a_batch, int64_t w_batch) {
  info += blockIdx.y;                       // batch member (grid.y)
  if (*info != 0) return;
  A += (int64_t)blockIdx.y * a_batch;
  W += (int64_t)blockIdx.y * w_batch;
  __shared__ DiagLds<T> S;
  const int lane = threadIdx.x & 63;
  // block row of this wave.  Waves v and v + 4 share a SIMD and block row W carries work in proportion to W + 1:
  // the rows are dealt 0 1 2 3 | 7 6 5 4 so that every SIMD gets the same total (rows 3 + 7 on one SIMD: 12 of 36)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = wv < 4 ? wv : 11 - wv;
  switch (w) {
    case 0: potrf256_wave<T, 0>(A, ld, W, info, row_base, S, lane); break;
    case 1: potrf256_wave<T, 1>(A, ld, W, info, row_base, S, lane); break;
    case 2: potrf256_wave<T, 2>(A, ld, W, info, row_base, S, lane); break;
    case 3: potrf256_wave<T, 3>(A, ld, W, info, row_base, S, lane); break;
    case 4: potrf256_wave<T, 4>(A, ld, W, info, row_base, S, lane); break;
    case 5: potrf256_wave<T, 5>(A, ld, W, info, row_base, S, lane); break;
    case 6: potrf256_wave<T, 6>(A, ld, W, info, row_base, S, lane); break;
    default: potrf256_wave<T, 7>(A, ld, W, info, row_base, S, lane); break;
  }
}

// ---- one WORKGROUP per chain member, N <= 256 (the reference's caller is the Python loop of
// stochastic.py:515-564 over the rows of a chain, at N = 30 ... 125 in its notebooks).  The batched sweep pushes such a
// batch through the large-N launch sequence (diagonal kernel, stripe solve of the right-hand-side rows, copies, a
// reduction kernel: five launches with the batch in grid.y, four passes over HBM); here ONE launch does, per member:
//   L = chol(K_b) in place (the fused 256-wide factorisation, or the 128-wide one), its block inverses W_b,
//   a_b = L^-1 delta_b through the inverses (a_0 = W_0 d_0;  a_1 = W_1 (d_1 - L_10 a_0)),
//   sum log L_ii, a^T a, the counts of non-finite entries -- four doubles per member.
// The covariance comes from the one batched Gram launch before it (grid.z = batch; generated per expression since round
// 4: 0.20 ms per 4096 members at N = 128).  Building it INSIDE this kernel was tried twice in round 5 and measured slower:
// the member's program interpreted per pair (g3_kernel_eval.h::prog_eval) costs 63 us per member, a written-out stationary
// kernel 27 us, against the 37 us of the whole factorisation -- one 512-thread workgroup per CU has two waves per SIMD and
// hides neither the loads of the inputs nor the fp64 exp; the full-occupancy Gram kernel does (profiles/r05_chain_small.txt).
#ifdef G3_SMALL_VGPR     // measurement builds: force the register budget (two workgroups per CU need <= 128)
#define G3_SMALL_ATTR __attribute__((amdgpu_num_vgpr(G3_SMALL_VGPR)))
#else
#define G3_SMALL_ATTR
#endif
template <typename T>
__global__ void __launch_bounds__(512, 4) G3_SMALL_ATTR
small_factor_kernel(T* K, int64_t ld, int64_t kstride, T* Wall, int64_t wstride, const T* delta, int64_t ldd, T* a_out, int64_t astride,
                    double* stats, int* info, int n, int np) {
  const int b = blockIdx.x;
  info += b;
  T* A = K + (int64_t)b * kstride;
  T* W = Wall + (int64_t)b * wstride;
  __shared__ DiagLds<T> S;
  // the solve's scratch lives in the memory the factorisation has finished with: the kernel then asks for 78 KB of LDS, and
  // TWO workgroups fit on a CU (2 x 78 <= 160 KB) when the register budget allows it as well
  T* const vec = reinterpret_cast<T*>(&S);                  // [2 * 128]  the right-hand side, then a
  T* const part = vec + 2 * G3_LB;                          // [4 * 128]  partial dot products (4 column quarters x 128 rows)
  double (*const red)[8] = reinterpret_cast<double (*)[8]>(part + 4 * G3_LB);   // [4][8]
  static_assert(sizeof(DiagLds<T>) >= 6 * G3_LB * sizeof(T) + 32 * sizeof(double), "solve scratch must fit in the factorisation's LDS");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int w = wv < 4 ? wv : 11 - wv;
  if (np == 2 * G3_LB) {
    switch (w) {
      case 0: potrf256_wave<T, 0>(A, ld, W, info, 0, S, lane); break;
      case 1: potrf256_wave<T, 1>(A, ld, W, info, 0, S, lane); break;
      case 2: potrf256_wave<T, 2>(A, ld, W, info, 0, S, lane); break;
      case 3: potrf256_wave<T, 3>(A, ld, W, info, 0, S, lane); break;
      case 4: potrf256_wave<T, 4>(A, ld, W, info, 0, S, lane); break;
      case 5: potrf256_wave<T, 5>(A, ld, W, info, 0, S, lane); break;
      case 6: potrf256_wave<T, 6>(A, ld, W, info, 0, S, lane); break;
      default: potrf256_wave<T, 7>(A, ld, W, info, 0, S, lane); break;
    }
  } else {
    switch (w) {
      case 0: diag128_wave<T, true, 0>(A, ld, W, (int64_t)G3_LB, info, 0, S, lane); break;
      case 1: diag128_wave<T, true, 1>(A, ld, W, (int64_t)G3_LB, info, 0, S, lane); break;
      case 2: diag128_wave<T, true, 2>(A, ld, W, (int64_t)G3_LB, info, 0, S, lane); break;
      case 3: diag128_wave<T, true, 3>(A, ld, W, (int64_t)G3_LB, info, 0, S, lane); break;
      case 4: diag128_wave<T, true, 4>(A, ld, W, (int64_t)G3_LB, info, 0, S, lane); break;
      case 5: diag128_wave<T, true, 5>(A, ld, W, (int64_t)G3_LB, info, 0, S, lane); break;
      case 6: diag128_wave<T, true, 6>(A, ld, W, (int64_t)G3_LB, info, 0, S, lane); break;
      default: diag128_wave<T, true, 7>(A, ld, W, (int64_t)G3_LB, info, 0, S, lane); break;
    }
  }
  // the factor and the inverses were written by other waves of this workgroup: make them visible to all of it
  __threadfence_block();
  __syncthreads();
  for (int i = tid; i < np; i += 512) vec[i] = i < n ? delta[(int64_t)b * ldd + i] : T(0);
  __syncthreads();
  // y = M x for a 128 x 128 row-major block M (leading dimension ldm): thread (row r = tid & 127, quarter q = tid >> 7)
  // takes 32 consecutive columns; the four partial sums of a row are added in a fixed order
  const int r = tid & (G3_LB - 1), q = tid >> 7;
  auto matvec = [&](const T* M, int64_t ldm, const T* x) {
    const T* row = M + (int64_t)r * ldm + 32 * q;
    T acc = T(0);
#pragma unroll 8
    for (int j = 0; j < 32; ++j) acc = fma(row[j], x[32 * q + j], acc);
    part[q * G3_LB + r] = acc;
    __syncthreads();
  };
  matvec(W, G3_LB, vec);                                   // a_0 = W_0 d_0
  if (tid < G3_LB) vec[tid] = (part[tid] + part[G3_LB + tid]) + (part[2 * G3_LB + tid] + part[3 * G3_LB + tid]);
  __syncthreads();
  if (np == 2 * G3_LB) {
    matvec(A + (int64_t)G3_LB * ld, ld, vec);              // L_10 a_0
    if (tid < G3_LB)
      vec[G3_LB + tid] -= (part[tid] + part[G3_LB + tid]) + (part[2 * G3_LB + tid] + part[3 * G3_LB + tid]);
    __syncthreads();
    matvec(W + G3_LB * G3_LB, G3_LB, vec + G3_LB);         // a_1 = W_1 (d_1 - L_10 a_0)
    T a1 = T(0);
    if (tid < G3_LB) a1 = (part[tid] + part[G3_LB + tid]) + (part[2 * G3_LB + tid] + part[3 * G3_LB + tid]);
    __syncthreads();
    if (tid < G3_LB) vec[G3_LB + tid] = a1;
    __syncthreads();
  }
  // a and the four scalars of logp_terms_kernel (g3_api.hip): sum log L_ii, a^T a, #non-finite a, #bad diagonal
  double ld_sum = 0, ss = 0, nf = 0, bd = 0;
  for (int i = tid; i < np; i += 512) {
    const T v = vec[i];
    a_out[(int64_t)b * astride + i] = v;
    if (i < n) {
      const double dg = (double)A[(int64_t)i * ld + i];
      ld_sum += log(dg);
      if (!(dg > 0.0) || __builtin_isinf(dg)) bd += 1;
      const double dv = (double)v;
      ss += dv * dv;
      if (dv != dv || __builtin_isinf(dv)) nf += 1;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    ld_sum += __shfl_down(ld_sum, off);
    ss += __shfl_down(ss, off);
    nf += __shfl_down(nf, off);
    bd += __shfl_down(bd, off);
  }
  if (lane == 0) { red[0][wv] = ld_sum; red[1][wv] = ss; red[2][wv] = nf; red[3][wv] = bd; }
  __syncthreads();
  if (tid < 4) {
    double t = 0;
    for (int k = 0; k < 8; ++k) t += red[tid][k];
    stats[4 * b + tid] = t;
  }
}

// the fused small-N evaluation of a batch (np = 128 or 256 padded rows per member); the members' covariances are in K
int g3i_small_factor_batched(g3_ctx* ctx, void* K, int64_t ld, int64_t kstride, void* W, int64_t wstride, const void* delta, int64_t ldd,
                             void* a, int64_t astride, double* dstats, int batch, int64_t n, int64_t np, g3_dtype dt) {
  if (np != G3_LB && np != 2 * G3_LB) return -1;
  G3_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * batch, ctx->stream));
  ctx->info_clean = false;
  if (dt == G3_F64)
    hipLaunchKernelGGL((small_factor_kernel<double>), dim3((unsigned)batch), dim3(512), 0, ctx->stream, (double*)K, ld, kstride, (double*)W,
                       wstride, (const double*)delta, ldd, (double*)a, astride, dstats, ctx->d_info, (int)n, (int)np);
  else
    hipLaunchKernelGGL((small_factor_kernel<float>), dim3((unsigned)batch), dim3(512), 0, ctx->stream, (float*)K, ld, kstride, (float*)W,
                       wstride, (const float*)delta, ldd, (float*)a, astride, dstats, ctx->d_info, (int)n, (int)np);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

template <typename T, bool FACTOR>
__global__ void __launch_bounds__(512)
diag128m_kernel(T* A, int64_t ld, int64_t a_stride, T* W, int64_t ldw, int64_t w_stride, int* info,
                int64_t row_base, int64_t a_batch, int64_t w_batch) {
  info += blockIdx.y;                       // batch member (grid.y)
  if (FACTOR && *info != 0) return;
  A += (int64_t)blockIdx.x * a_stride + (int64_t)blockIdx.y * a_batch;
  W += (int64_t)blockIdx.x * w_stride + (int64_t)blockIdx.y * w_batch;
  row_base += (int64_t)blockIdx.x * G3_LB;
  __shared__ DiagLds<T> S;
  const int lane = threadIdx.x & 63;
  // wave index as a scalar: the eight wave programs below are selected by a uniform branch, so a
  // wave only ever executes (and counts the barriers of) its own program
  // block row of this wave.  Waves v and v + 4 share a SIMD and block row W carries work in proportion to W + 1:
  // the rows are dealt 0 1 2 3 | 7 6 5 4 so that every SIMD gets the same total (rows 3 + 7 on one SIMD: 12 of 36)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = wv < 4 ? wv : 11 - wv;
  switch (w) {
    case 0: diag128_wave<T, FACTOR, 0>(A, ld, W, ldw, info, row_base, S, lane); break;
    case 1: diag128_wave<T, FACTOR, 1>(A, ld, W, ldw, info, row_base, S, lane); break;
    case 2: diag128_wave<T, FACTOR, 2>(A, ld, W, ldw, info, row_base, S, lane); break;
    case 3: diag128_wave<T, FACTOR, 3>(A, ld, W, ldw, info, row_base, S, lane); break;
    case 4: diag128_wave<T, FACTOR, 4>(A, ld, W, ldw, info, row_base, S, lane); break;
    case 5: diag128_wave<T, FACTOR, 5>(A, ld, W, ldw, info, row_base, S, lane); break;
    case 6: diag128_wave<T, FACTOR, 6>(A, ld, W, ldw, info, row_base, S, lane); break;
    default: diag128_wave<T, FACTOR, 7>(A, ld, W, ldw, info, row_base, S, lane); break;
  }
}

constexpr int64_t LB = G3_LB;

static int64_t split_point(int64_t n, int64_t unit) {
  // n is a multiple of `unit` and > unit; split near the middle on a coarse power-of-two
  // grid so that large sub-problems keep tile-friendly sizes
  int64_t g = unit;
  while (g * 2 <= n / 4 && g < 2048) g *= 2;
  int64_t n1 = g3_roundup(n / 2, g);
  if (n1 >= n) n1 = n - unit;
  return n1;
}

template <typename T>
static int potrf_diag(g3_ctx* ctx, T* A, int64_t ld, T* W, int64_t row_base, g3_dtype dt) {
  const int pr = g3i_prof_begin(ctx, G3_TAG_LEAF, 128.0 * 128.0 * 128.0 / 3.0);
  hipLaunchKernelGGL((diag128m_kernel<T, true>), dim3(1, (unsigned)g3_nbatch(ctx)), dim3(512), 0, ctx->stream, A, ld,
                     (int64_t)0, W, LB, (int64_t)0, ctx->d_info, row_base, g3_bstride_of(ctx, A), g3_bstride_of(ctx, W));
  g3i_prof_end(ctx, pr);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

// ---- across LB blocks
template <typename T>
static int trsm_rec(g3_ctx* ctx, const T* L, int64_t n, int64_t ldl, T* B, int64_t m, int64_t ldb,
                    const T* W, g3_dtype dt) {
  // a tall panel takes the top of the recursion as launches (its X_hi -= X_lo L_hi,lo^T products are then tiles of the bulk
  // GEMM, not 64 x 128 products of a stripe's workgroup) and only blocks up to trsm_split_n as stripe launches
  const bool split = ctx->tune.trsm_split_min > 0 && m >= ctx->tune.trsm_split_min && n > ctx->tune.trsm_split_n &&
                     g3_nbatch(ctx) == 1;
  if (n > LB && n <= 1024 && !split) {
    // the whole recursion below this point in one launch: a workgroup per 32-row stripe of B
    const int rc = g3i_trsm_stripe(ctx, L, n, ldl, B, m, ldb, W, dt);
    if (rc <= 0) return rc;
  }
  if (n == LB)
    // B <- B W^T in place: one tile spans the 128 output columns, so a workgroup has read
    // its rows before it overwrites them
    return g3i_gemm_nt_ex(ctx, B, ldb, B, ldb, W, LB, m, LB, LB, 1.0, 0.0, dt, 0, 1);
  const int64_t n1 = split_point(n, LB), n2 = n - n1;
  int rc = trsm_rec<T>(ctx, L, n1, ldl, B, m, ldb, W, dt);
  if (rc) return rc;
  rc = g3i_gemm_nt(ctx, B + n1, ldb, B, ldb, L + n1 * ldl, ldl, m, n2, n1, -1.0, 1.0, dt, 0);
  if (rc) return rc;
  return trsm_rec<T>(ctx, L + n1 * ldl + n1, n2, ldl, B + n1, m, ldb, W + (n1 / LB) * LB * LB, dt);
}

template <typename T>
static int potrf_diag256(g3_ctx* ctx, T* A, int64_t ld, T* W, int64_t row_base) {
  const int pr = g3i_prof_begin(ctx, G3_TAG_LEAF, 256.0 * 256.0 * 256.0 / 3.0);
  hipLaunchKernelGGL((potrf256_kernel<T>), dim3(1, (unsigned)g3_nbatch(ctx)), dim3(512), 0, ctx->stream, A, ld, W,
                     ctx->d_info, row_base, g3_bstride_of(ctx, A), g3_bstride_of(ctx, W));
  g3i_prof_end(ctx, pr);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

template <typename T>
static int potrf_rec(g3_ctx* ctx, T* A, int64_t n, int64_t ld, T* W, int64_t row_base, g3_dtype dt) {
  if (n == LB) return potrf_diag<T>(ctx, A, ld, W, row_base, dt);
  if (n == 2 * LB && ctx->fuse256) return potrf_diag256<T>(ctx, A, ld, W, row_base);
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

// the pivot flag is known to be zero for work queued on the context's current stream
bool g3i_info_known_zero(const g3_ctx* ctx) {
  return ctx->info_clean && g3_nbatch(ctx) == 1 && (ctx->info_sync || ctx->info_stream == ctx->stream);
}

// GEMM and solve kernels only READ the pivot flag (a failed pivot turns what follows into no-ops).  Once it has been cleared
// on this stream and no factorisation has been queued since, clearing it again is a 5 us stream operation for nothing --
// the multi-GPU driver makes ~450 such calls per evaluation.  Everything that can SET the flag marks it unknown first.
int g3i_reset_info(g3_ctx* ctx) {
  if (g3i_info_known_zero(ctx)) return G3_OK;
  G3_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * g3_nbatch(ctx), ctx->stream));
  if (g3_nbatch(ctx) == 1) {
    ctx->info_clean = true;
    ctx->info_sync = false;
    ctx->info_stream = ctx->stream;
  }
  return G3_OK;
}

int g3i_ensure_invd(g3_ctx* ctx, int64_t n, g3_dtype dt) {
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


// ---- flat right-looking sweep over NB-wide panels, grouped into super-panels of G panels, with one
// super-panel of look-ahead on two streams.
//
// The chain (stream A, the context's stream) factors panel after panel; WITHIN a super-panel a finished
// panel is applied at once to the remaining columns of its super-panel only (U_in, K = NB).  Everything to
// the right of the super-panel receives the update of all G panels together, K = G * NB, on the bulk
// stream B (low priority): the dominant launches of the factorisation run at twice / four times the K of the
// chain's granularity -- fewer read-modify-write passes over C per flop and half / a quarter as many
// launches -- while the chain keeps the narrow panels that make it short.  G = 1 is the plain
// look-ahead sweep of rounds 1-2.  Per super-panel s (columns [c0, c1), panels g0 .. g1-1):
//
//   A:  wait B2a(s-1) . Ua(s-1): first panel column of s  -= S_{s-1} S_{s-1}^T   (K of super-panel s-1)
//       panel(g0) . [wait B1(s-1)] . U_in(g0) . panel(g0+1) . U_in(g0+1) ...      -> event G(s)
//   B:  wait G(s) . P1(s):  the other columns of super-panel s+1               -> event B1(s)
//                   P2a(s): first panel column of super-panel s+2               -> event B2a(s)
//                   P2b(s): everything to the right of that                     (the bulk)
//
// so the chain of super-panel s+1 runs beside P2b(s) (and P2b(s-1) while it lasts).  The appended E
// right-hand-side rows are simply more rows of every update and panel solve.
template <typename T>
static int potrf_lookahead(g3_ctx* ctx, T* A, int64_t n, int64_t ld, T* W, int64_t NB, int G, g3_dtype dt, int64_t E) {
  const int64_t R = n + E;   // rows: the square part plus E appended right-hand-side rows
  std::vector<int64_t> bnd;
  std::vector<int> gb;
  g3h_panel_bounds(n, NB, G, ctx->batch, ctx->tune, &bnd, &gb);
  const int nblk = (int)bnd.size() - 1, ngrp = (int)gb.size() - 1;
  const int nev = 4 * ngrp + 1;
  if (ctx->la_nev < nev) {
    if (ctx->la_ev) {
      for (int i = 0; i < ctx->la_nev; ++i) (void)hipEventDestroy(ctx->la_ev[i]);
      free(ctx->la_ev);
    }
    ctx->la_nev = nev;
    ctx->la_ev = (hipEvent_t*)calloc(ctx->la_nev, sizeof(hipEvent_t));
    if (!ctx->la_ev) return G3_ERR_NOMEM;
    for (int i = 0; i < ctx->la_nev; ++i) G3_HIP(hipEventCreateWithFlags(&ctx->la_ev[i], hipEventDisableTiming));
  }
  hipEvent_t* evG = ctx->la_ev;                // super-panel s final (stream A)
  hipEvent_t* evB1 = ctx->la_ev + ngrp;        // P1(s) done (stream B)
  hipEvent_t* evB2a = ctx->la_ev + 2 * ngrp;   // P2a(s) done (stream B)
  hipEvent_t evJoin = ctx->la_ev[4 * ngrp];
  { const int rs = g3i_ensure_side_stream(ctx); if (rs) return rs; }
  hipStream_t sA = ctx->stream, sB = ctx->side_stream;
  auto r = [&](int k) { return k < nblk ? bnd[k] : n; };                 // first column of panel k
  auto c = [&](int s) { return s < ngrp ? bnd[gb[s]] : n; };             // first column of super-panel s
  int rc = G3_OK;
  // B must not start before everything already queued on A (Gram, memsets) is done
  G3_HIP(hipEventRecord(evJoin, sA));
  G3_HIP(hipStreamWaitEvent(sB, evJoin, 0));
  auto panel = [&](int k) -> int {   // D_k and P_k on stream A
    T* Akk = A + r(k) * ld + r(k);
    const int64_t w = r(k + 1) - r(k);
    int e = potrf_rec<T>(ctx, Akk, w, ld, W + (r(k) / LB) * LB * LB, r(k), dt);
    if (e) return e;
    if (r(k + 1) < R) e = trsm_rec<T>(ctx, Akk, w, ld, A + r(k + 1) * ld + r(k), R - r(k + 1), ld,
                                      W + (r(k) / LB) * LB * LB, dt);
    return e;
  };
  // C[rows >= row0, cols [col0, col1)] -= A[rows, [k0, k1)] A[[col0, col1), [k0, k1)]^T, lower trapezoid
  // (col <= row; the block starts on the diagonal when row0 == col0)
  auto update = [&](int64_t row0, int64_t col0, int64_t col1, int64_t k0, int64_t k1) -> int {
    if (col1 <= col0 || row0 >= R || k1 <= k0) return G3_OK;
    return g3i_gemm_nt_trap(ctx, A + row0 * ld + col0, ld, A + row0 * ld + k0, ld, A + col0 * ld + k0, ld, R - row0,
                            col1 - col0, k1 - k0, -1.0, 1.0, dt, row0 - col0);
  };
  for (int s = 0; s < ngrp; ++s) {
    const int g0 = gb[s], g1 = gb[s + 1];
    // ---- stream A: the chain of super-panel s
    if (s >= 1) {
      if (s >= 2) G3_HIP(hipStreamWaitEvent(sA, evB2a[s - 2], 0));
      rc = update(r(g0), r(g0), r(g0 + 1), c(s - 1), c(s));          // Ua(s-1)
      if (rc) return rc;
    }
    for (int j = g0; j < g1; ++j) {
      rc = panel(j);
      if (rc) return rc;
      if (j + 1 < g1) {
        if (j == g0 && s >= 1) G3_HIP(hipStreamWaitEvent(sA, evB1[s - 1], 0));
        // U_in(j): the rest of this super-panel.  The next panel's column first (it is all the chain
        // needs to go on), the columns behind it in a second launch
        rc = update(r(j + 1), r(j + 1), r(j + 2), r(j), r(j + 1));
        if (!rc && j + 2 < g1) rc = update(r(j + 2), r(j + 2), c(s + 1), r(j), r(j + 1));
        if (rc) return rc;
      }
    }
    G3_HIP(hipEventRecord(evG[s], sA));
    // ---- stream B: the update with super-panel s of everything to its right
    if (c(s + 1) < n) {
      G3_HIP(hipStreamWaitEvent(sB, evG[s], 0));
      ctx->stream = sB;
      const int h0 = gb[s + 1];                                       // first panel of super-panel s+1
      rc = update(r(h0 + 1), r(h0 + 1), c(s + 2), c(s), c(s + 1));    // P1(s)
      if (!rc && hipEventRecord(evB1[s], sB) != hipSuccess) rc = G3_ERR_HIP;
      if (!rc && c(s + 2) < n) {
        const int q0 = gb[s + 2];
        rc = update(c(s + 2), c(s + 2), r(q0 + 1), c(s), c(s + 1));   // P2a(s)
        if (!rc && hipEventRecord(evB2a[s], sB) != hipSuccess) rc = G3_ERR_HIP;
        if (!rc) rc = update(r(q0 + 1), r(q0 + 1), n, c(s), c(s + 1));   // P2b(s)
      } else if (!rc && hipEventRecord(evB2a[s], sB) != hipSuccess) {
        rc = G3_ERR_HIP;
      }
      ctx->stream = sA;
      if (rc) return rc;
    }
  }
  // join: A continues only after B has drained
  G3_HIP(hipEventRecord(evJoin, sB));
  G3_HIP(hipStreamWaitEvent(sA, evJoin, 0));
  return G3_OK;
}

#ifdef G3_CHAIN_SERVER   // measurement variant only (scripts/variants/chain_server.inc, scripts/build_variant.sh)
#include "../../scripts/variants/chain_server.inc"
#endif


int g3i_potrf(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, void* invd) {
  return g3i_potrf_tall(ctx, A, n, ld, dt, invd, 0);
}


static int64_t g3i_panel_width(g3_ctx* ctx, int64_t n, int* G) {
  int64_t NB = ctx->nb_lookahead;
  *G = ctx->tune.sb;                 // panels per super-panel
  if (NB <= 0) {
    // measured on MI355X (fp64): narrow panels shorten the latency-bound chain of diagonal-block
    // kernels that dominates small problems, wide panels give the bulk updates more K
    const bool forced = ctx->tune.nb > 0;
    NB = forced ? ctx->tune.nb : (n <= 4096 ? 128 : (n <= 6144 ? 256 : (n <= 12288 ? 512 : 1024)));
    // a batched sweep is bound by work per launch, not by the chain: wider panels again
    if (!forced && ctx->batch > 1 && NB < 256) NB = 256;
  }
  if (*G < 1) *G = 1;
  if (*G > 8) *G = 8;
  return g3_roundup(NB < LB ? LB : NB, LB);
}

// Cholesky of the leading n x n block of a tall (n + E) x n matrix whose last E rows are
// right-hand sides B: on return those rows hold B L^-T (the forward substitution rides along
// with the panel solves and trailing updates of the factorisation -- no separate trsm pass).
int g3i_potrf_tall(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, void* invd, int64_t E) {
  if (!g3i_info_known_zero(ctx))
    G3_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * g3_nbatch(ctx), ctx->stream));
  ctx->info_clean = false;
  if (n == 0) return G3_OK;
  // The one-launch 256-wide diagonal kernel shortens the dependency chain (N = 8192: 8.9 -> 8.2 ms when it was
  // introduced); since the round-2 rewrite of the diagonal kernels it is no slower at any size (N = 24576: 95.2 ->
  // 95.0 ms, 32768: equal), so it is always used.
  ctx->fuse256 = true;

  int G = 1;
  int64_t NB = g3i_panel_width(ctx, n, &G);
  if (n >= 3 * NB) {
#ifdef G3_CHAIN_SERVER
    if (g3i_chain_usable(ctx, n, NB, G)) {
      if (dt == G3_F64) return potrf_lookahead_chain<double>(ctx, (double*)A, n, ld, (double*)invd, NB, dt, E);
      return potrf_lookahead_chain<float>(ctx, (float*)A, n, ld, (float*)invd, NB, dt, E);
    }
#endif
    if (dt == G3_F64) return potrf_lookahead<double>(ctx, (double*)A, n, ld, (double*)invd, NB, G, dt, E);
    return potrf_lookahead<float>(ctx, (float*)A, n, ld, (float*)invd, NB, G, dt, E);
  }
  int rc;
  if (dt == G3_F64) rc = potrf_rec<double>(ctx, (double*)A, n, ld, (double*)invd, 0, dt);
  else rc = potrf_rec<float>(ctx, (float*)A, n, ld, (float*)invd, 0, dt);
  if (rc || E == 0) return rc;
  return g3i_trsm_rlt(ctx, A, n, ld, (char*)A + (size_t)n * ld * g3_esize(dt), E, ld, dt, invd);
}

// ---- X <- X L^-T for FEW right-hand-side rows against a LARGE factor (the posterior's cross solve after a
// factorisation: m = M test points, n = N; elliptical.py:81-91): right-looking over 1024-wide column blocks with one
// block of look-ahead on two streams.  The recursion above runs its n / 1024 leaf solves -- each the same ~20 dependent
// tile steps whatever m is, ~0.3 ms with only m / 16 workgroups -- and its updates one after the other; here the leaf of
// block j+1 and its own column update (chain, stream A) run beside the update of everything to the right of it with
// block j (bulk, stream B, K = 1024).  N = 32768: M = 1024 24.3 -> 20.6 ms, M = 128 12.0 -> 10.9 ms (what is left there is the
// leaves' own latency: twenty dependent tile steps through global memory, ~0.33 ms each; a four-buffer DMA pipeline for
// them was measured and did not help -- 11.4 ms -- the steps are bound by their fixed round trips, not by the K loop).
template <typename T>
static int trsm_lookahead(g3_ctx* ctx, const T* L, int64_t n, int64_t ldl, T* B, int64_t m, int64_t ldb, const T* W, g3_dtype dt) {
  const int64_t NBK = 1024;
  const int nblk = (int)((n + NBK - 1) / NBK);
  if (ctx->la_nev < 2 * nblk + 1) {
    if (ctx->la_ev) {
      for (int i = 0; i < ctx->la_nev; ++i) (void)hipEventDestroy(ctx->la_ev[i]);
      free(ctx->la_ev);
    }
    ctx->la_nev = 2 * nblk + 1;
    ctx->la_ev = (hipEvent_t*)calloc(ctx->la_nev, sizeof(hipEvent_t));
    if (!ctx->la_ev) return G3_ERR_NOMEM;
    for (int i = 0; i < ctx->la_nev; ++i) G3_HIP(hipEventCreateWithFlags(&ctx->la_ev[i], hipEventDisableTiming));
  }
  hipEvent_t* evX = ctx->la_ev;            // block j of X solved (stream A)
  hipEvent_t* evU = ctx->la_ev + nblk;     // everything right of block j+1 carries block j (stream B)
  hipEvent_t evJoin = ctx->la_ev[2 * nblk];
  { const int rs = g3i_ensure_side_stream(ctx); if (rs) return rs; }
  hipStream_t sA = ctx->stream, sB = ctx->side_stream;
  auto c = [&](int j) { return j < nblk ? (int64_t)j * NBK : n; };
  G3_HIP(hipEventRecord(evJoin, sA));
  G3_HIP(hipStreamWaitEvent(sB, evJoin, 0));
  auto leaf = [&](int j) -> int {
    return trsm_rec<T>(ctx, L + c(j) * ldl + c(j), c(j + 1) - c(j), ldl, B + c(j), m, ldb, W + (c(j) / LB) * LB * LB, dt);
  };
  // B[:, col0:col1) -= X_j L[col0:col1, block j]^T
  auto update = [&](int j, int64_t col0, int64_t col1) -> int {
    if (col1 <= col0) return G3_OK;
    return g3i_gemm_nt(ctx, B + col0, ldb, B + c(j), ldb, L + col0 * ldl + c(j), ldl, m, col1 - col0, c(j + 1) - c(j), -1.0, 1.0, dt, 0);
  };
  int rc = leaf(0);
  if (rc) return rc;
  G3_HIP(hipEventRecord(evX[0], sA));
  for (int j = 0; j + 1 < nblk; ++j) {
    // stream B: block j applied to the blocks right of j+1
    if (c(j + 2) < n) {
      G3_HIP(hipStreamWaitEvent(sB, evX[j], 0));
      ctx->stream = sB;
      rc = update(j, c(j + 2), n);
      ctx->stream = sA;
      if (rc) return rc;
    }
    G3_HIP(hipEventRecord(evU[j], sB));
    // stream A: block j applied to block j+1 (which carries the blocks before j once U_{j-1} has fired), then its leaf
    if (j >= 1) G3_HIP(hipStreamWaitEvent(sA, evU[j - 1], 0));
    rc = update(j, c(j + 1), c(j + 2));
    if (!rc) rc = leaf(j + 1);
    if (rc) return rc;
    G3_HIP(hipEventRecord(evX[j + 1], sA));
  }
  G3_HIP(hipEventRecord(evJoin, sB));
  G3_HIP(hipStreamWaitEvent(sA, evJoin, 0));
  return G3_OK;
}

int g3i_trsm_rlt(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, void* B, int64_t m,
                 int64_t ldb, g3_dtype dt, const void* invd) {
  if (n == 0 || m == 0) return G3_OK;
  // few rows against a long factor, outside batch mode and not from inside a two-stream sweep (side stream free)
  if (n >= 4096 && m <= 4096 && ctx->batch <= 1 && !g3_on_bulk_stream(ctx)) {
    if (dt == G3_F64) return trsm_lookahead<double>(ctx, (const double*)L, n, ldl, (double*)B, m, ldb, (const double*)invd, dt);
    return trsm_lookahead<float>(ctx, (const float*)L, n, ldl, (float*)B, m, ldb, (const float*)invd, dt);
  }
  if (dt == G3_F64)
    return trsm_rec<double>(ctx, (const double*)L, n, ldl, (double*)B, m, ldb, (const double*)invd, dt);
  return trsm_rec<float>(ctx, (const float*)L, n, ldl, (float*)B, m, ldb, (const float*)invd, dt);
}

int g3i_trtri_blocks(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, g3_dtype dt, void* invd) {
  if (n == 0) return G3_OK;
  ctx->info_clean = false;       // the inversion kernel takes the flag read-write

  const unsigned nb = (unsigned)(n / LB);
  if (dt == G3_F64)
    hipLaunchKernelGGL((diag128m_kernel<double, false>), dim3(nb, (unsigned)g3_nbatch(ctx)), dim3(512), 0, ctx->stream,
                       (double*)const_cast<void*>(L), ldl, LB * (ldl + 1), (double*)invd, LB, LB * LB,
                       ctx->d_info, (int64_t)0, g3_bstride_of(ctx, L), g3_bstride_of(ctx, invd));
  else
    hipLaunchKernelGGL((diag128m_kernel<float, false>), dim3(nb, (unsigned)g3_nbatch(ctx)), dim3(512), 0, ctx->stream,
                       (float*)const_cast<void*>(L), ldl, LB * (ldl + 1), (float*)invd, LB, LB * LB,
                       ctx->d_info, (int64_t)0, g3_bstride_of(ctx, L), g3_bstride_of(ctx, invd));
  G3_LAUNCH_CHECK();
  return G3_OK;
}

// ---- the FULL inverse V = L^-1 of a factored n x n diagonal block (n = 128 * 2^q <= 2048), from the inverses W of its
// 128 x 128 diagonal blocks, by recursive doubling: with L = [L11 0; L21 L22] and V11, V22 known,
//   V21 = -V22 L21 V11.
// Both V and its transpose Vt are carried (the GEMM tile is "NT": C = A B^T with both operands k-contiguous), so every
// product is one: Ut = Vt11 L21^T (= (L21 V11)^T);  V21 = -V22 Ut^T;  Vt12 = V21^T = -Ut V22^T.  Two dependent launches per
// level, all pairs of a level in one launch (grid.y), triangular operands cut the reduction range per tile.  The multi-GPU
// driver broadcasts V instead of (L, W): a rank's panel solve X L^-T is then ONE product X V^T on the matrix pipe instead of
// a 20-step stripe recursion (g3_dist.hip).  Error: |V - L^-1| ~ kappa(L) eps per level.
// All four matrices are compact n x n (leading dimension n); V's blocks above and Vt's blocks below the block diagonal are
// never written and never read.
template <typename T>
__global__ void __launch_bounds__(256) trinv_seed_kernel(T* __restrict__ V, T* __restrict__ Vt, const T* __restrict__ W, int64_t ld) {
  const T* w = W + (int64_t)blockIdx.x * G3_LB * G3_LB;
  T* v = V + (int64_t)blockIdx.x * G3_LB * (ld + 1);
  T* vt = Vt + (int64_t)blockIdx.x * G3_LB * (ld + 1);
  for (int e = threadIdx.x + 256 * blockIdx.y; e < G3_LB * G3_LB; e += 256 * gridDim.y) {
    const int i = e >> 7, j = e & (G3_LB - 1);
    const T x = w[e];
    v[(int64_t)i * ld + j] = x;
    vt[(int64_t)j * ld + i] = x;
  }
}

template <typename T>
__global__ void __launch_bounds__(256, 2)
trinv_mul_kernel(T* V, T* Vt, T* U, const T* L, int64_t ld, int h, int step, const int* __restrict__ info) {
  const int failed = *info;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = h / 64;
  const int ti = (int)blockIdx.x / nt, tj = (int)blockIdx.x % nt;
  const int m0 = ti * 64, n0 = tj * 64;
  const int64_t o = (int64_t)blockIdx.y * 2 * h;                 // first row / column of this pair
  T* Up = U + o * ld + o;                                        // Ut of the pair (h x h)
  if (step == 0) {
    // Ut[a][b] = sum_{k >= a} Vt11[a][k] L21[b][k]
    const T* A = Vt + o * ld + o;
    const T* B = L + (o + h) * ld + o;
    const int k0 = m0;
    gemm_tile<T, 64, 64, 32, 32, STAGES>(Up, ld, A + (int64_t)m0 * ld + k0, ld, B + (int64_t)n0 * ld + k0, ld, h - k0, T(1), T(0), m0, n0,
                                         false, G3_DENSE_OFF, failed, smem);
  } else if (blockIdx.z == 0) {
    // V21[i][j] = -sum_{k <= i} V22[i][k] Ut[j][k]
    const T* A = V + (o + h) * ld + (o + h);
    const int K = m0 + 64 < h ? m0 + 64 : h;
    gemm_tile<T, 64, 64, 32, 32, STAGES>(V + (o + h) * ld + o, ld, A + (int64_t)m0 * ld, ld, Up + (int64_t)n0 * ld, ld, K, T(-1), T(0), m0, n0,
                                         false, G3_DENSE_OFF, failed, smem);
  } else {
    // Vt12[j][i] = V21[i][j] = -sum_{k <= i} Ut[j][k] V22[i][k]
    const T* B = V + (o + h) * ld + (o + h);
    const int K = n0 + 64 < h ? n0 + 64 : h;
    gemm_tile<T, 64, 64, 32, 32, STAGES>(Vt + o * ld + (o + h), ld, Up + (int64_t)m0 * ld, ld, B + (int64_t)n0 * ld, ld, K, T(-1), T(0), m0, n0,
                                         false, G3_DENSE_OFF, failed, smem);
  }
}

template <typename T>
static int trtri_full_t(g3_ctx* ctx, const T* L, int64_t n, const T* W, T* V, T* Vt, T* U) {
  constexpr int LDS = STAGES * (64 + 64) * ROWB;
  hipLaunchKernelGGL((trinv_seed_kernel<T>), dim3((unsigned)(n / LB), 8), dim3(256), 0, ctx->stream, V, Vt, W, n);
  G3_LAUNCH_CHECK();
  for (int64_t h = LB; 2 * h <= n; h *= 2) {
    const unsigned tiles = (unsigned)((h / 64) * (h / 64)), pairs = (unsigned)(n / (2 * h));
    const int pr = g3i_prof_begin(ctx, G3_TAG_GEMM_SMALL, (double)pairs * 2.0 * (double)h * h * h);
    hipLaunchKernelGGL((trinv_mul_kernel<T>), dim3(tiles, pairs, 1), dim3(256), LDS, ctx->stream, V, Vt, U, L, n, (int)h, 0, ctx->d_info);
    G3_LAUNCH_CHECK();
    hipLaunchKernelGGL((trinv_mul_kernel<T>), dim3(tiles, pairs, 2 * h < n ? 2 : 1), dim3(256), LDS, ctx->stream, V, Vt, U, L, n, (int)h, 1,
                       ctx->d_info);
    g3i_prof_end(ctx, pr);
    G3_LAUNCH_CHECK();
  }
  return G3_OK;
}

// n = 128 * 2^q <= 2048; L, V, Vt, U compact n x n; W the 128 x 128 block inverses (n / 128 of them).  A failed pivot flag
// (d_info of this context, left by the factorisation just before) turns the launches into no-ops.
int g3i_trtri_full(g3_ctx* ctx, const void* L, int64_t n, const void* W, void* V, void* Vt, void* U, g3_dtype dt) {
  if (n < LB || n > 2048 || (n & (n - 1)) != 0) return -3;
  if (dt == G3_F64) return trtri_full_t<double>(ctx, (const double*)L, n, (const double*)W, (double*)V, (double*)Vt, (double*)U);
  return trtri_full_t<float>(ctx, (const float*)L, n, (const float*)W, (float*)V, (float*)Vt, (float*)U);
}

extern "C" int g3_trtri_full(g3_ctx* ctx, const void* L_dev, int64_t n, const void* invd_dev, void* V_dev, void* Vt_dev, void* U_dev,
                             g3_dtype dt) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!L_dev) return -2;
  if (n < G3_LB || n > 2048 || (n & (n - 1)) != 0) return -3;
  if (!invd_dev) return -4;
  if (!V_dev) return -5;
  if (!Vt_dev) return -6;
  if (!U_dev) return -7;
  if (((uintptr_t)L_dev | (uintptr_t)V_dev | (uintptr_t)Vt_dev | (uintptr_t)U_dev) & 15) return -2;
  int rc = g3i_reset_info(ctx);
  if (rc) return rc;
  return g3i_trtri_full(ctx, L_dev, n, invd_dev, V_dev, Vt_dev, U_dev, dt);
}

extern "C" int g3_trsm_full(g3_ctx* ctx, const void* V_dev, int64_t n, int64_t ldv, const void* B_dev, int64_t m, int64_t ldb,
                            void* X_dev, int64_t ldx, g3_dtype dt) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!V_dev) return -2;
  if (n < 0 || n % G3_LB) return -3;
  const int64_t al = 16 / (int64_t)g3_esize(dt);
  if (ldv < n || ldv % al) return -4;
  if (!B_dev) return -5;
  if (m < 0 || m % 64) return -6;
  if (ldb < n || ldb % al) return -7;
  if (!X_dev || X_dev == B_dev) return -8;
  if (ldx < n) return -9;
  if (((uintptr_t)V_dev | (uintptr_t)B_dev) & 15) return -2;
  int rc = g3i_reset_info(ctx);
  if (rc) return rc;
  return g3i_gemm_nt_ktri(ctx, X_dev, ldx, B_dev, ldb, V_dev, ldv, m, n, 1.0, 0.0, dt);
}

static int read_info(g3_ctx* ctx, int* info_host) {
  G3_HIP(hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  G3_HIP(hipStreamSynchronize(ctx->stream));
  *info_host = *ctx->h_info;
#ifdef G3_CHAIN_SERVER
  if (g3i_chain_gave_up(ctx, *info_host)) {
    // the matrix was being factored in place: the caller must supply it again (the server is off from now on)
    (void)g3i_reset_info(ctx);
    snprintf(ctx->err, sizeof(ctx->err), "the resident chain workgroups gave up; the factorisation is incomplete -- call again");
    return G3_ERR_HIP;
  }
#endif
  return G3_OK;
}

extern "C" int g3_potrf(g3_ctx* ctx, void* A_dev, int64_t n, int64_t ld, g3_dtype dt,
                        void* invd_dev, int* info_host) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
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

// the first non-zero info of a sequence of factorisations, kept on the device
__global__ void info_merge_kernel(const int* info, int* accum) {
  if (*accum == 0 && *info != 0) *accum = *info;
}

// g3_potrf without the host synchronisation: the caller supplies a 4-byte device accumulator that
// keeps the first non-zero info of all calls made with it (read it once, at the end of a sweep).
extern "C" int g3_potrf_nowait(g3_ctx* ctx, void* A_dev, int64_t n, int64_t ld, g3_dtype dt, void* invd_dev,
                               int* info_accum_dev) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!A_dev) return -2;
  if (n < 0 || n % G3_LB) return -3;
  if (ld < n || ld % (16 / (int64_t)g3_esize(dt))) return -4;
  if (!invd_dev) return -6;
  if (!info_accum_dev) return -7;
  int rc;
  if (n <= 2048) {
    // a diagonal block of the multi-GPU sweep: it is factored beside bulk updates that fill the chip, exactly like
    // the diagonal block of a panel of the one-GPU sweep -- same code path (one stream, recursion down to the fused
    // 256-wide kernel) instead of the two-stream sweep tuned for a stand-alone small matrix
    G3_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * g3_nbatch(ctx), ctx->stream));
    ctx->info_clean = false;
    ctx->fuse256 = true;
    if (n == 0) rc = G3_OK;
    else if (dt == G3_F64) rc = potrf_rec<double>(ctx, (double*)A_dev, n, ld, (double*)invd_dev, 0, dt);
    else rc = potrf_rec<float>(ctx, (float*)A_dev, n, ld, (float*)invd_dev, 0, dt);
  } else {
#ifdef G3_CHAIN_SERVER
    const bool broken = ctx->chain_broken;
    ctx->chain_broken = true;              // (the flag of this call is read much later: no resident workgroups here)
#endif
    rc = g3i_potrf(ctx, A_dev, n, ld, dt, invd_dev);
#ifdef G3_CHAIN_SERVER
    ctx->chain_broken = broken;
#endif
  }
  if (rc) return rc;
  hipLaunchKernelGGL(info_merge_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_info, info_accum_dev);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

extern "C" int g3_trsm_rlt(g3_ctx* ctx, const void* L_dev, int64_t n, int64_t ldl, void* B_dev,
                           int64_t m, int64_t ldb, g3_dtype dt, const void* invd_dev) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
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
#ifdef G3_CHAIN_SERVER
    const bool chain_was_on = !ctx->chain_broken;
#endif
    r = read_info(ctx, info);
#ifdef G3_CHAIN_SERVER
    if (r == G3_ERR_HIP && chain_was_on && ctx->chain_broken) {        // once: the copy above is simply made again
      hipLaunchKernelGGL((copy_lower_pad_kernel<T>), grid, dim3(256), 0, ctx->stream, F, ldf, K, ldk, n, np, (T)add);
      G3_LAUNCH_CHECK();
      r = g3i_potrf(ctx, F, np, ldf, dt, ctx->invd);
      if (!r) r = read_info(ctx, info);
    }
#endif
    return r;
  };
  int info = 0, tries = 0, fallback = 0;
  double jitter = 0.0;
  rc = attempt(0.0, &info);
  if (rc) return rc;
  if (info != 0) {
    double st[3];
    rc = g3_diag_stats(ctx, K, n, ldk, dt, st);
    if (rc) return rc;
    G3hJitter jit(st[1], st[0]);
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
      if (finite && jit.usable()) {
        rc = attempt(jit.value(), &info);
        if (rc) return rc;
        if (info == 0) {
          ok = true;
          jitter = jit.value();
          break;
        }
      }
      jit.next();
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
  g3_dev_guard _dg(ctx);
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

#ifdef G3_PROBE   // measurement build only: the round-4 feasibility probes (scripts/variants/probe_resident.inc)
#include "../../scripts/variants/probe_resident.inc"
#endif
