// Gradient of the GP log marginal likelihood with respect to the kernel hyper-parameters.
//
// Reference: StochasticProcess.th_dlogp = gradient(th_logp) (g3py/processes/stochastic.py:308-309,
// g3py/libs/tensors.py:11-22) -- Theano differentiates logp_cho (gaussian.py:208-224) through
// CholeskyRobust.grad (tensors.py:224-260, Murray 2016).  In closed form that chain is
//     d logp / d theta = 1/2 * sum_ij G_ij * dK_ij/dtheta ,   G = alpha alpha^T - K^-1 ,  alpha = K^-1 delta
// with K the matrix that was actually factored (jitter included, treated as a constant, exactly
// as grad() re-uses the jittered factor).  Here:
//   g3i_potri        Y = L^-T by a right-looking sweep with one-panel look-ahead (the same
//                    two-stream schedule as the factorisation), K^-1 = Y Y^T accumulated panel by
//                    panel as SYRKs on the bulk stream -- 2 N^3 / 3 flops, all in the MFMA GEMM;
//   gram_grad_kernel one pass over the lower triangle of K^-1 (HBM-read bound) that re-evaluates
//                    the kernel expression per pair and accumulates G_ij * d k_ij / d theta for
//                    every parameter of every leaf; deterministic two-stage reduction.
#include "g3_internal.h"
#include <vector>
#include <stdlib.h>

#define GG_T 64            // pair tile edge
#define GG_THREADS 256
#define GG_PI 3.14159265358979323846

// ------------------------------------------------------------------------------------------
// K^-1 from the factor
// ------------------------------------------------------------------------------------------
static int ensure_events(g3_ctx* ctx, int need) {
  if (ctx->la_nev >= need) return G3_OK;
  if (ctx->la_ev) {
    for (int i = 0; i < ctx->la_nev; ++i) (void)hipEventDestroy(ctx->la_ev[i]);
    free(ctx->la_ev);
    ctx->la_ev = nullptr;
    ctx->la_nev = 0;
  }
  ctx->la_ev = (hipEvent_t*)calloc((size_t)need, sizeof(hipEvent_t));
  if (!ctx->la_ev) return G3_ERR_NOMEM;
  for (int i = 0; i < need; ++i) G3_HIP(hipEventCreateWithFlags(&ctx->la_ev[i], hipEventDisableTiming));
  ctx->la_nev = need;
  return G3_OK;
}

// Y (n x n, upper triangular on return) = L^-T and C (lower triangle) = K^-1 = Y Y^T.
// Right-looking over NB-wide panels of columns: panel k of Y is final after the solve against
// L_kk; it then updates the columns to its right (rows 0..r1 only: Y is upper triangular, so
// no flop is spent on structural zeros) and adds its outer product to C.
// rows x row_bytes zeros at `pitch` for every batch member (grid.z), members `bstride` bytes apart
__global__ void __launch_bounds__(256) zero2d_batched_kernel(char* __restrict__ p, size_t pitch, size_t row_bytes, int64_t rows, size_t bstride) {
  const size_t c = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (c >= row_bytes) return;
  p += (size_t)blockIdx.z * bstride + c;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) *reinterpret_cast<uint4*>(p + r * pitch) = make_uint4(0, 0, 0, 0);
}

int g3i_potri(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, const void* invd, g3_dtype dt, void* Y,
              int64_t ldy, void* C, int64_t ldc) {
  if (n == 0) return G3_OK;
  const size_t es = g3_esize(dt);
  int64_t NB = ctx->nb_lookahead;
  if (NB <= 0) {
    // the inverse's chain is light (one panel solve per step): wide panels from mid sizes on
    // (measured, dlogp: N=8192 9.4 -> 8.4 ms, 16384 55.7 -> 55.0 ms)
    NB = ctx->tune.nb > 0 ? ctx->tune.nb : (n <= 4096 ? 128 : (n <= 6144 ? 256 : (n <= 12288 ? 512 : 1024)));
  }
  NB = g3_roundup(NB < G3_LB ? G3_LB : NB, G3_LB);
  const int nblk = (int)((n + NB - 1) / NB);
  int rc = ensure_events(ctx, 2 * nblk + 2);
  if (rc) return rc;
  hipEvent_t* evP = ctx->la_ev;
  hipEvent_t* evB = ctx->la_ev + nblk;
  const bool two = nblk >= 3;   // small problems: everything on the caller's stream
  if (two) { const int rs = g3i_ensure_side_stream(ctx); if (rs) return rs; }
  hipStream_t sA = ctx->stream, sB = ctx->side_stream;
  if (!two) sB = sA;

  if (g3_nbatch(ctx) > 1 && ((n * es) % 16 == 0) && ((ldy * es) % 16 == 0) && ((ldc * es) % 16 == 0) &&
      ((((uintptr_t)Y | (uintptr_t)C) & 15) == 0) && ((g3_bstride_of(ctx, Y) * es) % 16 == 0) && ((g3_bstride_of(ctx, C) * es) % 16 == 0)) {
    // batch mode: every member's Y and K^-1 in two launches (one memset per member is 8192 stream operations for a chain
    // of 4096 members -- 15 ms of device time and 80 ms of host time for 2.5 ms of kernels)
    const unsigned gx = (unsigned)((n * es / 16 + 255) / 256);
    const unsigned gy = (unsigned)(n < 64 ? n : 64);
    const dim3 grid(gx, gy, (unsigned)g3_nbatch(ctx));
    hipLaunchKernelGGL(zero2d_batched_kernel, grid, dim3(256), 0, sA, (char*)Y, (size_t)ldy * es, (size_t)n * es, n, (size_t)g3_bstride_of(ctx, Y) * es);
    hipLaunchKernelGGL(zero2d_batched_kernel, grid, dim3(256), 0, sA, (char*)C, (size_t)ldc * es, (size_t)n * es, n, (size_t)g3_bstride_of(ctx, C) * es);
    G3_LAUNCH_CHECK();
  } else {
    for (int b = 0; b < g3_nbatch(ctx); ++b) {
      G3_HIP(hipMemset2DAsync((char*)Y + (size_t)b * g3_bstride_of(ctx, Y) * es, (size_t)ldy * es, 0, (size_t)n * es, (size_t)n, sA));
      G3_HIP(hipMemset2DAsync((char*)C + (size_t)b * g3_bstride_of(ctx, C) * es, (size_t)ldc * es, 0, (size_t)n * es, (size_t)n, sA));
    }
  }
  rc = g3i_diag_add(ctx, Y, n, ldy, dt, 1.0);
  if (rc) return rc;
  if (two) {
    G3_HIP(hipEventRecord(evB[nblk], sA));
    G3_HIP(hipStreamWaitEvent(sB, evB[nblk], 0));
  }
  auto r = [&](int k) { return (int64_t)k * NB < n ? (int64_t)k * NB : n; };
  auto Lp = [&](int64_t i, int64_t j) { return (const char*)L + ((size_t)i * ldl + j) * es; };
  auto Yp = [&](int64_t j) { return (char*)Y + (size_t)j * es; };   // row 0, column j
  const char* Wb = (const char*)invd;
  auto panel = [&](int k) -> int {   // Y[0:r(k+1), panel k] <- . L_kk^-T
    return g3i_trsm_rlt(ctx, Lp(r(k), r(k)), r(k + 1) - r(k), ldl, Yp(r(k)), r(k + 1), ldy, dt,
                        Wb + (size_t)(r(k) / G3_LB) * G3_LB * G3_LB * es);
  };
  auto update = [&](int k, int64_t c0, int64_t c1) -> int {   // columns [c0, c1) of Y with panel k
    return g3i_gemm_nt(ctx, Yp(c0), ldy, Yp(r(k)), ldy, Lp(c0, r(k)), ldl, r(k + 1), c1 - c0, r(k + 1) - r(k),
                       -1.0, 1.0, dt, 0);
  };
  auto syrk = [&](int k) -> int {
    return g3i_gemm_nt(ctx, C, ldc, Yp(r(k)), ldy, Yp(r(k)), ldy, r(k + 1), r(k + 1), r(k + 1) - r(k), 1.0, 1.0,
                       dt, 1);
  };
  rc = panel(0);
  if (rc) return rc;
  if (two) G3_HIP(hipEventRecord(evP[0], sA));
  for (int k = 0; k + 1 < nblk; ++k) {
    const int64_t r2 = r(k + 2), r3 = r(k + 3);
    // bulk stream: columns beyond the next panel, then this panel's share of K^-1
    if (two) G3_HIP(hipStreamWaitEvent(sB, evP[k], 0));
    ctx->stream = sB;
    if (r2 < n) {
      rc = update(k, r2, r3);
      if (!rc && two && hipEventRecord(evB[k], sB) != hipSuccess) rc = G3_ERR_HIP;
      if (!rc && r3 < n) rc = update(k, r3, n);
    } else if (two && hipEventRecord(evB[k], sB) != hipSuccess) {
      rc = G3_ERR_HIP;
    }
    if (!rc) rc = syrk(k);
    ctx->stream = sA;
    if (rc) return rc;
    // critical path: next panel
    if (two && k >= 1) G3_HIP(hipStreamWaitEvent(sA, evB[k - 1], 0));
    rc = update(k, r(k + 1), r2);
    if (rc) return rc;
    rc = panel(k + 1);
    if (rc) return rc;
    if (two) G3_HIP(hipEventRecord(evP[k + 1], sA));
  }
  if (two) G3_HIP(hipStreamWaitEvent(sB, evP[nblk - 1], 0));
  ctx->stream = sB;
  rc = syrk(nblk - 1);
  ctx->stream = sA;
  if (rc) return rc;
  if (two) {
    G3_HIP(hipEventRecord(evB[nblk], sB));
    G3_HIP(hipStreamWaitEvent(sA, evB[nblk], 0));
  }
  return G3_OK;
}

extern "C" int g3_potri(g3_ctx* ctx, const void* L_dev, int64_t n, int64_t ldl, const void* invd_dev, g3_dtype dt,
                        void* Y_dev, int64_t ldy, void* Kinv_dev, int64_t ldc) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!L_dev) return -2;
  if (n < 0 || n % G3_LB) return -3;
  const int64_t al = 16 / (int64_t)g3_esize(dt);
  if (ldl < n || ldl % al) return -4;
  if (!Y_dev) return -7;
  if (ldy < n || ldy % al) return -8;
  if (!Kinv_dev) return -9;
  if (ldc < n || ldc % al) return -10;
  int rc = g3i_reset_info(ctx);
  if (rc) return rc;
  if (!invd_dev) {
    rc = g3i_ensure_invd(ctx, n, dt);
    if (rc) return rc;
    rc = g3i_trtri_blocks(ctx, L_dev, n, ldl, dt, ctx->invd);
    if (rc) return rc;
    invd_dev = ctx->invd;
  }
  return g3i_potri(ctx, L_dev, n, ldl, invd_dev, dt, Y_dev, ldy, Kinv_dev, ldc);
}

// ------------------------------------------------------------------------------------------
// sum_ij G_ij dK_ij / dtheta
// ------------------------------------------------------------------------------------------
extern "C" int g3_grad_layout(const g3_kernel_prog* prog, g3_grad_map* map) {
  if (!prog) return -1;
  if (!map) return -2;
  if (prog->nleaf < 0 || prog->nleaf > G3_MAXLEAF) return -1;
  int s = 0;
  for (int l = 0; l < G3_MAXLEAF; ++l) map->var[l] = map->alpha[l] = map->rate[l] = map->freq[l] = -1;
  for (int l = 0; l < prog->nleaf; ++l) {
    const g3_leaf& lf = prog->leaf[l];
    const int nd = lf.ndims;
    map->var[l] = s++;
    switch (lf.kind) {
      case G3_K_SE: case G3_K_OU: case G3_K_MAT32: case G3_K_MAT52:
        map->rate[l] = s; s += nd; break;
      case G3_K_RQ:
        map->alpha[l] = s++; map->rate[l] = s; s += nd; break;
      case G3_K_COS: case G3_K_SINC:
        map->freq[l] = s; s += nd; break;
      case G3_K_SIN: case G3_K_SM:
        map->freq[l] = s; s += nd; map->rate[l] = s; s += nd; break;
      default: break;
    }
  }
  map->nslots = s;
  return G3_OK;
}

// thread-private accumulators in LDS: slot-major so that a wave touches 64 consecutive doubles
struct SlotAcc {
  double* acc;     // [window][GG_THREADS]
  int lo, width, tid;
  __device__ __forceinline__ void operator()(int slot, double v) const {
    const unsigned s = (unsigned)(slot - lo);
    if (s < (unsigned)width) acc[s * GG_THREADS + tid] += v;
  }
};

// unit-variance value of a leaf for the pair and, through `add`, w * d(var * k)/d(param) for
// every parameter of the leaf (w = G_ij * dK/d(leaf value)).  Formulas follow the leaf
// definitions in g3_gram.hip (kernels.py:388-487, metrics.py:89-102).
template <typename Acc>
__device__ __forceinline__ void leaf_grad(const g3_leaf& lf, int l, const g3_grad_map& map, const double* xi,
                                          const double* xj, bool diag, double w, const Acc& add) {
  const int nd = lf.ndims;
  const double var = lf.var;
  const double wv = w * var;
  switch (lf.kind) {
    case G3_K_NOISE:
    case G3_K_WN:
      if (diag) add(map.var[l], w);
      return;
    case G3_K_SE:
    case G3_K_MAT32:
    case G3_K_MAT52:
    case G3_K_RQ: {
      double D = 0.0;
      for (int k = 0; k < nd; ++k) {
        const double dx = xi[lf.dims[k]] - xj[lf.dims[k]];
        D += (dx * dx) * (0.5 * lf.rate[k] * lf.rate[k]);
      }
      double kv, dkdD;
      if (lf.kind == G3_K_SE) {
        kv = exp(-D); dkdD = -kv;
      } else if (lf.kind == G3_K_MAT32) {
        const double s = sqrt(3.0 * D), e = exp(-s);
        kv = (1.0 + s) * e; dkdD = -1.5 * e;
      } else if (lf.kind == G3_K_MAT52) {
        const double s = sqrt(5.0 * D), e = exp(-s);
        kv = (1.0 + s + 5.0 * D / 3.0) * e; dkdD = -(5.0 / 6.0) * (1.0 + s) * e;
      } else {
        const double al = lf.alpha, b = 1.0 + D / al;
        kv = pow(b, -al); dkdD = -kv / b;
        add(map.alpha[l], wv * kv * (-log(b) + D / (al + D)));
      }
      add(map.var[l], w * kv);
      const double c = wv * dkdD;
      for (int k = 0; k < nd; ++k) {
        const double dx = xi[lf.dims[k]] - xj[lf.dims[k]];
        add(map.rate[l] + k, c * lf.rate[k] * (dx * dx));
      }
      return;
    }
    case G3_K_OU: {
      double D = 0.0;
      for (int k = 0; k < nd; ++k) D += fabs(xi[lf.dims[k]] - xj[lf.dims[k]]) * lf.rate[k];
      const double kv = exp(-D);
      add(map.var[l], w * kv);
      for (int k = 0; k < nd; ++k) add(map.rate[l] + k, -wv * kv * fabs(xi[lf.dims[k]] - xj[lf.dims[k]]));
      return;
    }
    case G3_K_SIN: {
      double s = 0.0;
      for (int k = 0; k < nd; ++k) {
        const double v = sin(GG_PI * (xi[lf.dims[k]] - xj[lf.dims[k]]) * lf.freq[k]);
        s += (v * v) * lf.rate[k];
      }
      const double kv = exp(2.0 * s);
      add(map.var[l], w * kv);
      for (int k = 0; k < nd; ++k) {
        const double dx = xi[lf.dims[k]] - xj[lf.dims[k]];
        const double v = sin(GG_PI * dx * lf.freq[k]);
        add(map.rate[l] + k, wv * kv * 2.0 * (v * v));
        add(map.freq[l] + k, wv * kv * (2.0 * GG_PI) * lf.rate[k] * dx * sin(2.0 * GG_PI * dx * lf.freq[k]));
      }
      return;
    }
    case G3_K_COS:
    case G3_K_SINC:
    case G3_K_SM: {
      // value = env * prod_k f_k ; d/dfreq_m = env * f'_m * prod_{k != m} f_k
      const bool sinc = lf.kind == G3_K_SINC;
      const double cs = sinc ? 2.0 * GG_PI * GG_PI : 2.0 * GG_PI;
      double p = 1.0, s = 0.0;
      for (int k = 0; k < nd; ++k) {
        const double dx = xi[lf.dims[k]] - xj[lf.dims[k]];
        const double th = cs * dx * lf.freq[k];
        p *= sinc ? (dx != 0.0 ? sin(th) / th : 1.0) : cos(th);
        s += (dx * dx) * (lf.rate[k] * lf.rate[k]);
      }
      const double env = lf.kind == G3_K_SM ? exp(-2.0 * GG_PI * GG_PI * s) : 1.0;
      const double kv = env * p;
      add(map.var[l], w * kv);
      for (int m = 0; m < nd; ++m) {
        const double dx = xi[lf.dims[m]] - xj[lf.dims[m]];
        const double th = cs * dx * lf.freq[m];
        double fm, dfm;
        if (sinc) {
          fm = dx != 0.0 ? sin(th) / th : 1.0;
          dfm = dx != 0.0 ? (cos(th) - fm) / lf.freq[m] : 0.0;
        } else {
          fm = cos(th);
          dfm = -cs * dx * sin(th);
        }
        double others;
        if (fm != 0.0) {
          others = p / fm;
        } else {
          others = 1.0;
          for (int k = 0; k < nd; ++k) {
            if (k == m) continue;
            const double dk = xi[lf.dims[k]] - xj[lf.dims[k]];
            const double tk = cs * dk * lf.freq[k];
            others *= sinc ? (dk != 0.0 ? sin(tk) / tk : 1.0) : cos(tk);
          }
        }
        add(map.freq[l] + m, wv * env * dfm * others);
        if (lf.kind == G3_K_SM) add(map.rate[l] + m, wv * kv * (-4.0 * GG_PI * GG_PI) * (dx * dx) * lf.rate[m]);
      }
      return;
    }
    default:
      return;
  }
}

// value of a leaf (variance included) for the pair, in fp64
__device__ __forceinline__ double leaf_value(const g3_leaf& lf, const double* xi, const double* xj, bool diag) {
  const int nd = lf.ndims;
  switch (lf.kind) {
    case G3_K_NOISE:
    case G3_K_WN:
      return diag ? lf.var : 0.0;
    case G3_K_SE: case G3_K_MAT32: case G3_K_MAT52: case G3_K_RQ: {
      double D = 0.0;
      for (int k = 0; k < nd; ++k) {
        const double dx = xi[lf.dims[k]] - xj[lf.dims[k]];
        D += (dx * dx) * (0.5 * lf.rate[k] * lf.rate[k]);
      }
      if (lf.kind == G3_K_SE) return lf.var * exp(-D);
      if (lf.kind == G3_K_MAT32) { const double s = sqrt(3.0 * D); return lf.var * ((1.0 + s) * exp(-s)); }
      if (lf.kind == G3_K_MAT52) { const double s = sqrt(5.0 * D); return lf.var * ((1.0 + s + 5.0 * D / 3.0) * exp(-s)); }
      return lf.var * pow(1.0 + D / lf.alpha, -lf.alpha);
    }
    case G3_K_OU: {
      double D = 0.0;
      for (int k = 0; k < nd; ++k) D += fabs(xi[lf.dims[k]] - xj[lf.dims[k]]) * lf.rate[k];
      return lf.var * exp(-D);
    }
    case G3_K_SIN: {
      double s = 0.0;
      for (int k = 0; k < nd; ++k) {
        const double v = sin(GG_PI * (xi[lf.dims[k]] - xj[lf.dims[k]]) * lf.freq[k]);
        s += (v * v) * lf.rate[k];
      }
      return lf.var * exp(2.0 * s);
    }
    case G3_K_COS: case G3_K_SINC: case G3_K_SM: {
      const bool sinc = lf.kind == G3_K_SINC;
      const double cs = sinc ? 2.0 * GG_PI * GG_PI : 2.0 * GG_PI;
      double p = 1.0, s = 0.0;
      for (int k = 0; k < nd; ++k) {
        const double dx = xi[lf.dims[k]] - xj[lf.dims[k]];
        const double th = cs * dx * lf.freq[k];
        p *= sinc ? (dx != 0.0 ? sin(th) / th : 1.0) : cos(th);
        s += (dx * dx) * (lf.rate[k] * lf.rate[k]);
      }
      return lf.var * (lf.kind == G3_K_SM ? exp(-2.0 * GG_PI * GG_PI * s) * p : p);
    }
    default:
      return 0.0;
  }
}

// One workgroup walks 64 x 64 tiles of the lower triangle (grid-stride).  Thread t owns column
// t & 63 and rows (t >> 6) + 4 r of the tile; G rows are read 512 B at a time per wave.
template <typename T>
__global__ void __launch_bounds__(GG_THREADS)
gram_grad_kernel(const g3_kernel_prog* __restrict__ prog, g3_grad_map map, const T* __restrict__ X, int64_t N,
                 int64_t ldx, int d, const T* __restrict__ G, int64_t ldg, const T* __restrict__ alpha,
                 double* __restrict__ partial, int lo, int width, int64_t row0, int64_t row1, int64_t gstride, int64_t astride) {
  // rows [row0, row1) of the lower triangle (row0 a multiple of the tile edge); G holds those rows only.
  // grid.y = batch member (chains of hyper-parameter vectors on the same inputs): its own program, K^-1, alpha and partial sums
  prog += blockIdx.y;
  G += (int64_t)blockIdx.y * gstride;
  alpha += (int64_t)blockIdx.y * astride;
  partial += (size_t)blockIdx.y * gridDim.x * width;
  extern __shared__ __attribute__((aligned(16))) char smem_gg[];
  const int dp = d | 1;
  double* xi_s = (double*)smem_gg;             // GG_T x dp
  double* xj_s = xi_s + GG_T * dp;             // GG_T x dp
  double* ai_s = xj_s + GG_T * dp;             // GG_T
  double* aj_s = ai_s + GG_T;                  // GG_T
  double* lv_s = aj_s + GG_T;                  // G3_MAXLEAF x GG_THREADS
  double* acc_s = lv_s + G3_MAXLEAF * GG_THREADS;   // width x GG_THREADS
  const int tid = threadIdx.x;
  for (int s = 0; s < width; ++s) acc_s[s * GG_THREADS + tid] = 0.0;
  const SlotAcc add{acc_s, lo, width, tid};
  // the program is read many times per pair with data-dependent indices: keep it in LDS
  // (scalar loads from global memory in the inner loops would stall every wave)
  __shared__ g3_kernel_prog sp;
  __shared__ double qconst[G3_MAXLEAF];
  __shared__ int need_lv;
  for (int e = tid; e < (int)(sizeof(g3_kernel_prog) / 4); e += GG_THREADS)
    reinterpret_cast<int*>(&sp)[e] = reinterpret_cast<const int*>(prog)[e];
  __syncthreads();
  const int nleaf = sp.nleaf, nprod = sp.nprod;
  if (tid == 0) {
    // dK/d(leaf l) is the constant sum of its coefficients when every product has one factor
    int multi = 0;
    for (int p = 0; p < nprod; ++p) multi |= sp.prod[p].nfac > 1;
    need_lv = multi;
    for (int l = 0; l < G3_MAXLEAF; ++l) {
      double q = 0.0;
      for (int p = 0; p < nprod; ++p)
        if (sp.prod[p].nfac == 1 && sp.prod[p].fac[0] == l) q += sp.prod[p].coef;
      qconst[l] = q;
    }
  }
  __syncthreads();
  const bool multi = need_lv != 0;
  const int64_t bi0 = row0 / GG_T, bi1 = (row1 + GG_T - 1) / GG_T;
  const int64_t id0 = bi0 * (bi0 + 1) / 2, ntiles = bi1 * (bi1 + 1) / 2 - id0;
  for (int64_t idl = blockIdx.x; idl < ntiles; idl += gridDim.x) {
    const int64_t id = id0 + idl;
    int64_t bi = (int64_t)((sqrt(1.0 + 8.0 * (double)id) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= id) ++bi;
    while (bi * (bi + 1) / 2 > id) --bi;
    const int64_t bj = id - bi * (bi + 1) / 2;
    const int64_t i0 = bi * GG_T, j0 = bj * GG_T;
    __syncthreads();
    for (int e = tid; e < GG_T * d; e += GG_THREADS) {
      const int r = e / d, c = e - r * d;
      xi_s[r * dp + c] = i0 + r < row1 ? (double)X[(i0 + r) * ldx + c] : 0.0;
      xj_s[r * dp + c] = j0 + r < N ? (double)X[(j0 + r) * ldx + c] : 0.0;
    }
    if (tid < GG_T) ai_s[tid] = i0 + tid < row1 ? (double)alpha[i0 + tid] : 0.0;
    else if (tid < 2 * GG_T) aj_s[tid - GG_T] = j0 + tid - GG_T < N ? (double)alpha[j0 + tid - GG_T] : 0.0;
    __syncthreads();
    const int c = tid & (GG_T - 1);
    const int64_t j = j0 + c;
    const double* xj = xj_s + c * dp;
    for (int rr = tid >> 6; rr < GG_T; rr += GG_THREADS / GG_T) {
      const int64_t i = i0 + rr;
      if (i >= row1 || j > i) continue;
      const bool diag = i == j;
      const double* xi = xi_s + rr * dp;
      // G_ij with the symmetric pair (j, i) folded in
      const double g = (diag ? 0.5 : 1.0) * (ai_s[rr] * aj_s[c] - (double)G[(i - row0) * ldg + j]);
      if (!multi) {
        for (int l = 0; l < nleaf; ++l)
          if (qconst[l] != 0.0) leaf_grad(sp.leaf[l], l, map, xi, xj, diag, g * qconst[l], add);
        continue;
      }
      for (int l = 0; l < nleaf; ++l) lv_s[l * GG_THREADS + tid] = leaf_value(sp.leaf[l], xi, xj, diag);
      for (int l = 0; l < nleaf; ++l) {
        // dK/d(leaf l) = sum over the products that contain it of coef * the other factors
        double q = 0.0;
        for (int p = 0; p < nprod; ++p) {
          const g3_prod& pr = sp.prod[p];
          bool has = false;
          double v = pr.coef;
          for (int f = 0; f < pr.nfac; ++f) {
            if (pr.fac[f] == l && !has) has = true;
            else v *= lv_s[pr.fac[f] * GG_THREADS + tid];
          }
          if (has) q += v;
        }
        if (q != 0.0) leaf_grad(sp.leaf[l], l, map, xi, xj, diag, g * q, add);
      }
    }
  }
  __syncthreads();
  // block reduction: wave w sums slots w, w + 4, ...
  const int lane = tid & 63, wv = tid >> 6;
  for (int s = wv; s < width; s += GG_THREADS / 64) {
    double v = 0.0;
    for (int q = 0; q < GG_THREADS / 64; ++q) v += acc_s[s * GG_THREADS + q * 64 + lane];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) partial[(size_t)blockIdx.x * width + s] = v;
  }
}

__global__ void __launch_bounds__(256)
grad_reduce_kernel(const double* __restrict__ partial, int nblocks, int width, double* __restrict__ out, int ostride) {
  __shared__ double red[256];
  const int s = blockIdx.x;
  partial += (size_t)blockIdx.y * nblocks * width;      // batch member
  out += (size_t)blockIdx.y * ostride;
  double v = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) v += partial[(size_t)b * width + s];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[s] = red[0];
}

// ---- fast path: prog == var * k(all d columns in order) [+ noise on the diagonal] with k one stationary kernel
// (SE, OU, MAT32, MAT52, RQ); D and the kind are compile-time.  Accumulators live in registers (var, noise, D rates,
// alpha); one exp (or pow) per pair; for SE the pass is bound by reading the lower triangle of K^-1.  Slot order of
// the partial sums: [var, noise, alpha, rate_0..rate_{D-1}].  Derivatives (kernels.py:388-436, metrics.py:89-102):
//   d = sum_k w_k dx_k^2, w = rate^2 / 2 (OU: d = sum_k rate_k |dx_k|);  dk/drate_k = dk/dd * rate_k dx_k^2  (OU: * |dx_k|)
//   SE, OU  dk/dd = -k            MAT32  -3/2 e^-s, s = sqrt(3 d)      MAT52  -5/6 (1 + s) e^-s, s = sqrt(5 d)
//   RQ      dk/dd = -(1 + d/alpha)^(-alpha-1),  dk/dalpha = k (d / (alpha + d) - log(1 + d/alpha))
template <int D>
struct SeGradParams {
  double w[D];      // 0.5 * rate^2 (OU: rate)
  double rate[D];
  double var, alpha;
  // optional periodic term (COS, SIN or SM: kernels.py:466-467, 471-472, 486-487), added to the stationary term or
  // multiplied with it (mul): f = 2 pi freq, pr = the periodic leaf's rate
  double f[D], pr[D];
  double pvar;
  int mul;
};

// PK >= 0 adds the periodic term p = pvar * k2, t_k = 2 pi freq_k dx_k, with cos / sin of the angle DIFFERENCE from per-tile
// tables (as the Gram fast path):
//   COS  k2 = prod_k cos t_k                         dk2/dfreq_k = -2 pi dx_k sin t_k prod_{k' != k} cos t_k'
//   SM   k2 = exp(-2 pi^2 sum dx_k^2 r_k^2) prod cos dk2/dfreq_k as COS times the envelope, dk2/dr_k = -4 pi^2 dx_k^2 r_k k2
//   SIN  k2 = exp(2 sum_k r_k sin^2(t_k / 2))        dk2/dr_k = (1 - cos t_k) k2,  dk2/dfreq_k = 2 pi r_k dx_k sin t_k k2
// se.mul: K = (var k1)(pvar k2) -- each term's derivatives carry the other term's value -- instead of var k1 + pvar k2.
// Slots: [var, noise, alpha, rate_0..D-1, pvar, freq_0..D-1, prate_0..D-1 (SIN, SM)]
template <typename T, int D, int FK, int PK>
__global__ void __launch_bounds__(GG_THREADS)
gram_grad_se_kernel(SeGradParams<D> se, const T* __restrict__ X, int64_t N, int64_t ldx, const T* __restrict__ G,
                    int64_t ldg, const T* __restrict__ alpha, double* __restrict__ partial, int64_t row0, int64_t row1,
                    const SeGradParams<D>* __restrict__ sev, int64_t gstride, int64_t astride) {
  constexpr bool PER = PK >= 0;
  constexpr bool PRATE = PK == G3_K_SIN || PK == G3_K_SM;
  constexpr int NS = !PER ? D + 3 : PRATE ? 3 * D + 4 : 2 * D + 4;
  // grid.y = batch member: its own parameters (sev, device array), K^-1, alpha and partial sums
  if (sev != nullptr) se = sev[blockIdx.y];
  G += (int64_t)blockIdx.y * gstride;
  alpha += (int64_t)blockIdx.y * astride;
  partial += (size_t)blockIdx.y * gridDim.x * NS;
  constexpr int TS2 = 2 * D + 1;     // trig row stride (odd)
  __shared__ double xi_s[GG_T * (D | 1)], xj_s[GG_T * (D | 1)], ai_s[GG_T], aj_s[GG_T];
  __shared__ double ti_s[PER ? GG_T * TS2 : 1], tj_s[PER ? GG_T * TS2 : 1];
  __shared__ double red[NS * (GG_THREADS / 64)];
  constexpr int dp = D | 1;
  const int tid = threadIdx.x;
  double g_var = 0.0, g_noise = 0.0, g_alpha = 0.0, g_rate[D], g_pvar = 0.0, g_freq[D], g_prate[PRATE ? D : 1];
#pragma unroll
  for (int k = 0; k < D; ++k) { g_rate[k] = 0.0; g_freq[k] = 0.0; }
#pragma unroll
  for (int k = 0; k < (PRATE ? D : 1); ++k) g_prate[k] = 0.0;
  const int64_t bi0 = row0 / GG_T, bi1 = (row1 + GG_T - 1) / GG_T;
  const int64_t id0 = bi0 * (bi0 + 1) / 2, ntiles = bi1 * (bi1 + 1) / 2 - id0;
  for (int64_t idl = blockIdx.x; idl < ntiles; idl += gridDim.x) {
    const int64_t id = id0 + idl;
    int64_t bi = (int64_t)((sqrt(1.0 + 8.0 * (double)id) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= id) ++bi;
    while (bi * (bi + 1) / 2 > id) --bi;
    const int64_t bj = id - bi * (bi + 1) / 2;
    const int64_t i0 = bi * GG_T, j0 = bj * GG_T;
    __syncthreads();
    for (int e = tid; e < GG_T * D; e += GG_THREADS) {
      const int r = e / D, c = e - r * D;
      xi_s[r * dp + c] = i0 + r < row1 ? (double)X[(i0 + r) * ldx + c] : 0.0;
      xj_s[r * dp + c] = j0 + r < N ? (double)X[(j0 + r) * ldx + c] : 0.0;
    }
    if (tid < GG_T) ai_s[tid] = i0 + tid < row1 ? (double)alpha[i0 + tid] : 0.0;
    else if (tid < 2 * GG_T) aj_s[tid - GG_T] = j0 + tid - GG_T < N ? (double)alpha[j0 + tid - GG_T] : 0.0;
    __syncthreads();
    if constexpr (PER) {
      for (int e = tid; e < 2 * GG_T * D; e += GG_THREADS) {
        const int side = e / (GG_T * D), q = e - side * (GG_T * D), r = q / D, k = q - r * D;
        const double th = se.f[k] * (side ? xj_s[r * dp + k] : xi_s[r * dp + k]);
        double* t = (side ? tj_s : ti_s) + r * TS2 + 2 * k;
        t[0] = cos(th);
        t[1] = sin(th);
      }
      __syncthreads();
    }
    const int c = tid & (GG_T - 1);
    const int64_t j = j0 + c;
    double xj[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xj[k] = xj_s[c * dp + k];
    double cj[PER ? D : 1], sj[PER ? D : 1];
    if constexpr (PER) {
#pragma unroll
      for (int k = 0; k < D; ++k) { cj[k] = tj_s[c * TS2 + 2 * k]; sj[k] = tj_s[c * TS2 + 2 * k + 1]; }
    }
    const double aj = aj_s[c];
#pragma unroll 4
    for (int rr = tid >> 6; rr < GG_T; rr += GG_THREADS / GG_T) {
      const int64_t i = i0 + rr;
      if (i >= row1 || j > i) continue;
      const bool diag = i == j;
      const double g = (diag ? 0.5 : 1.0) * (ai_s[rr] * aj - (double)G[(i - row0) * ldg + j]);
      double dm[D], dxs[D], dd = 0.0;       // dm: dx^2 (|dx| for OU)
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double dx = xi_s[rr * dp + k] - xj[k];
        dxs[k] = dx;
        dm[k] = FK == G3_K_OU ? fabs(dx) : dx * dx;
        dd = fma(dm[k], se.w[k], dd);
      }
      double kv, dkdd;              // unit-variance kernel value and its derivative with respect to d
      if constexpr (FK == G3_K_MAT32) {
        const double s3 = sqrt(3.0 * dd), e = exp(-s3);
        kv = (1.0 + s3) * e;
        dkdd = -1.5 * e;
      } else if constexpr (FK == G3_K_MAT52) {
        const double s5 = sqrt(5.0 * dd), e = exp(-s5);
        kv = (1.0 + s5 + 5.0 * dd / 3.0) * e;
        dkdd = -(5.0 / 6.0) * (1.0 + s5) * e;
      } else if constexpr (FK == G3_K_RQ) {
        const double b = 1.0 + dd / se.alpha;
        kv = pow(b, -se.alpha);
        dkdd = -kv / b;
        g_alpha = fma(g * se.var * kv, dd / (se.alpha + dd) - log(b), g_alpha);
      } else {
        kv = exp(-dd);
        dkdd = -kv;
      }
      double ms = 1.0;              // multiplier of the stationary term's derivatives (product form: pvar * k2)
      if constexpr (PER) {
        const double mp = se.mul ? se.var * kv : 1.0;     // ... and of the periodic term's
        double cs[D], sn[D], pre[D + 1];
        double q2 = 0.0;            // SM: sum dx^2 r^2;  SIN: sum r (1 - cos t) / 2
        pre[0] = 1.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const double ci = ti_s[rr * TS2 + 2 * k], si = ti_s[rr * TS2 + 2 * k + 1];
          cs[k] = ci * cj[k] + si * sj[k];      // cos(theta_i - theta_j)
          sn[k] = si * cj[k] - ci * sj[k];      // sin(theta_i - theta_j)
          pre[k + 1] = pre[k] * cs[k];
          if constexpr (PK == G3_K_SM) q2 = fma(dxs[k] * dxs[k], se.pr[k] * se.pr[k], q2);
          if constexpr (PK == G3_K_SIN) q2 = fma(0.5 * (1.0 - cs[k]), se.pr[k], q2);
        }
        const double gm = g * mp;
        if constexpr (PK == G3_K_SIN) {
          const double k2 = exp(2.0 * q2);
          g_pvar = fma(gm, k2, g_pvar);
          const double gp = gm * se.pvar * k2;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            g_prate[k] = fma(gp, 1.0 - cs[k], g_prate[k]);
            g_freq[k] = fma(gp * (2.0 * GG_PI) * se.pr[k], dxs[k] * sn[k], g_freq[k]);
          }
          if (se.mul) ms = se.pvar * k2;
        } else {
          const double env = PK == G3_K_SM ? exp(-2.0 * GG_PI * GG_PI * q2) : 1.0;
          const double k2 = env * pre[D];
          g_pvar = fma(gm, k2, g_pvar);
          double suf = 1.0;
          const double gp = -gm * se.pvar * env * (2.0 * GG_PI);
#pragma unroll
          for (int k = D - 1; k >= 0; --k) {
            g_freq[k] = fma(gp * dxs[k] * sn[k], pre[k] * suf, g_freq[k]);
            suf *= cs[k];
          }
          if constexpr (PK == G3_K_SM) {
            const double gr = gm * se.pvar * k2 * (-4.0 * GG_PI * GG_PI);
#pragma unroll
            for (int k = 0; k < D; ++k) g_prate[k] = fma(gr * se.pr[k], dxs[k] * dxs[k], g_prate[k]);
          }
          if (se.mul) ms = se.pvar * k2;
        }
      }
      g_var = fma(g * ms, kv, g_var);
      if (diag) g_noise += g;
      const double gv = g * ms * dkdd * se.var;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        if constexpr (FK == G3_K_OU) g_rate[k] = fma(gv, dm[k], g_rate[k]);
        else g_rate[k] = fma(gv * se.rate[k], dm[k], g_rate[k]);
      }
    }
  }
  // block reduction of the NS sums
  const int lane = tid & 63, wv = tid >> 6;
  double vals[NS];
  vals[0] = g_var; vals[1] = g_noise; vals[2] = g_alpha;
#pragma unroll
  for (int k = 0; k < D; ++k) vals[3 + k] = g_rate[k];
  if constexpr (PER) {
    vals[D + 3] = g_pvar;
#pragma unroll
    for (int k = 0; k < D; ++k) vals[D + 4 + k] = g_freq[k];
    if constexpr (PRATE) {
#pragma unroll
      for (int k = 0; k < D; ++k) vals[2 * D + 4 + k] = g_prate[k];
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double v = vals[s];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[s * (GG_THREADS / 64) + wv] = v;
  }
  __syncthreads();
  if (tid < NS) {
    double v = 0.0;
    for (int q = 0; q < GG_THREADS / 64; ++q) v += red[tid * (GG_THREADS / 64) + q];
    partial[(size_t)blockIdx.x * NS + tid] = v;
  }
}

template <int D>
static int match_se_grad(const g3_kernel_prog* p, int d, SeGradParams<D>* out, int* leaf_se, int* leaf_noise, int* leaf_per) {
  // the shapes of the Gram fast path (g3h_match_fast): one stationary leaf, optionally one periodic leaf added to it or
  // multiplied with it, optionally white noise.  Returns the stationary leaf's kind, or -1; *leaf_per = the periodic leaf, or -1
  *leaf_per = -1;
  if (d != D || p->nprod < 1 || p->nprod > 3 || p->shift != 0.0) return -1;
  int se = -1, noise = -1, per = -1, mul = 0;
  auto is_stat = [](int kd) { return kd == G3_K_SE || kd == G3_K_OU || kd == G3_K_MAT32 || kd == G3_K_MAT52 || kd == G3_K_RQ; };
  auto is_per = [](int kd) { return kd == G3_K_COS || kd == G3_K_SIN || kd == G3_K_SM; };
  for (int q = 0; q < p->nprod; ++q) {
    if (p->prod[q].coef != 1.0) return -1;
    if (p->prod[q].nfac == 2) {
      const int l0 = p->prod[q].fac[0], l1 = p->prod[q].fac[1];
      if (se >= 0 || per >= 0) return -1;
      if (is_stat(p->leaf[l0].kind) && is_per(p->leaf[l1].kind)) { se = l0; per = l1; }
      else if (is_stat(p->leaf[l1].kind) && is_per(p->leaf[l0].kind)) { se = l1; per = l0; }
      else return -1;
      mul = 1;
      continue;
    }
    if (p->prod[q].nfac != 1) return -1;
    const int l = p->prod[q].fac[0];
    const int kd = p->leaf[l].kind;
    if (is_stat(kd) && se < 0) se = l;
    else if (kd == G3_K_NOISE && noise < 0) noise = l;
    else if (is_per(kd) && per < 0) per = l;
    else return -1;
  }
  if (se < 0) return -1;
  out->mul = mul;
  if (per >= 0) {     // instantiated for SE / MAT32 / MAT52 and d in {1, 2, 4, 8}, as the Gram fast path
    const g3_leaf& pl = p->leaf[per];
    const int kd = p->leaf[se].kind;
    if (!((kd == G3_K_SE || kd == G3_K_MAT32 || kd == G3_K_MAT52) && (D == 1 || D == 2 || D == 4 || D == 8)) || pl.ndims != D) return -1;
    for (int k = 0; k < D; ++k) {
      if (pl.dims[k] != k) return -1;
      out->f[k] = 2.0 * GG_PI * pl.freq[k];
      out->pr[k] = pl.rate[k];
    }
    out->pvar = pl.var;
    *leaf_per = per;
  } else {
    for (int k = 0; k < D; ++k) { out->f[k] = 0.0; out->pr[k] = 0.0; }
    out->pvar = 0.0;
  }
  const g3_leaf& lf = p->leaf[se];
  if (lf.ndims != D) return -1;
  for (int k = 0; k < D; ++k) {
    if (lf.dims[k] != k) return -1;
    out->w[k] = lf.kind == G3_K_OU ? lf.rate[k] : 0.5 * lf.rate[k] * lf.rate[k];
    out->rate[k] = lf.rate[k];
  }
  out->var = lf.var;
  out->alpha = lf.alpha;
  *leaf_se = se;
  *leaf_noise = noise;
  return lf.kind;
}

template <int D>
static int gram_grad_se(g3_ctx* ctx, const g3_kernel_prog* progs, int batch, const g3_grad_map* map, const void* X, int64_t N,
                        int64_t ldx, g3_dtype dt, const void* G, int64_t ldg, int64_t gstride, const void* alpha, int64_t astride,
                        double* out_host, bool* handled, int64_t row0, int64_t row1) {
  // `batch` members with programs progs[0 .. batch) of ONE structure: member b reads G + b gstride, alpha + b astride and
  // writes out_host + b nslots.  All members must match the same compile-time shape, else nothing is done here.
  SeGradParams<D> se;
  int lse = -1, lnoise = -1, lper = -1;
  const int kind = match_se_grad<D>(&progs[0], D, &se, &lse, &lnoise, &lper);
  *handled = kind >= 0;
  if (!*handled) return G3_OK;
  const int pkind = lper >= 0 ? progs[0].leaf[lper].kind : -1;
  std::vector<SeGradParams<D>> sev;
  if (batch > 1) {
    sev.resize(batch);
    sev[0] = se;
    for (int b = 1; b < batch; ++b) {
      int l0 = -1, l1 = -1, l2 = -1;
      const int kb = match_se_grad<D>(&progs[b], D, &sev[b], &l0, &l1, &l2);
      if (kb != kind || l0 != lse || l1 != lnoise || l2 != lper || sev[b].mul != se.mul ||
          (lper >= 0 && progs[b].leaf[lper].kind != pkind)) {
        *handled = false;
        return G3_OK;
      }
    }
  }
  const bool prate = pkind == G3_K_SIN || pkind == G3_K_SM;
  const int ns = lper < 0 ? D + 3 : prate ? 3 * D + 4 : 2 * D + 4;
  const int64_t bi0 = row0 / GG_T, bi1 = (row1 + GG_T - 1) / GG_T;
  const int64_t ntiles = bi1 * (bi1 + 1) / 2 - bi0 * (bi0 + 1) / 2;
  int nblocks = (int)(ntiles < 4096 ? ntiles : 4096);
  if (batch > 1) {     // enough workgroups over the whole batch; a member's tiles are then walked by fewer of them
    const int per = (int)((8192 + batch - 1) / batch);
    if (nblocks > per) nblocks = per < 1 ? 1 : per;
  }
  const size_t pbytes = (size_t)batch * nblocks * ns * sizeof(double);
  const size_t obytes = (size_t)batch * 64 * sizeof(double);
  const size_t sbytes = batch > 1 ? (size_t)batch * sizeof(SeGradParams<D>) : 0;
  int rc = g3i_ensure_work(ctx, pbytes + obytes + sbytes);
  if (rc) return rc;
  double* partial = (double*)ctx->work;
  double* dout = (double*)((char*)ctx->work + pbytes);
  SeGradParams<D>* dsev = nullptr;
  if (batch > 1) {
    dsev = (SeGradParams<D>*)((char*)ctx->work + pbytes + obytes);
    G3_HIP(hipMemcpyAsync(dsev, sev.data(), sbytes, hipMemcpyHostToDevice, ctx->stream));
  }
  const dim3 grid((unsigned)nblocks, (unsigned)batch);
  int rec = g3i_prof_begin(ctx, G3_TAG_GRAM, (double)batch * (row0 == 0 && row1 == N ? (double)N * (N + 1) / 2 : (double)ntiles * GG_T * GG_T) * g3_esize(dt));
#define G3_GRAD_FAST(KIND, PKIND)                                                                                   \
  do {                                                                                                              \
    if (dt == G3_F64)                                                                                               \
      hipLaunchKernelGGL((gram_grad_se_kernel<double, D, KIND, PKIND>), grid, dim3(GG_THREADS), 0, ctx->stream, se,  \
                         (const double*)X, N, ldx, (const double*)G, ldg, (const double*)alpha, partial, row0, row1, \
                         dsev, gstride, astride);                                                                   \
    else                                                                                                            \
      hipLaunchKernelGGL((gram_grad_se_kernel<float, D, KIND, PKIND>), grid, dim3(GG_THREADS), 0, ctx->stream, se,   \
                         (const float*)X, N, ldx, (const float*)G, ldg, (const float*)alpha, partial, row0, row1,   \
                         dsev, gstride, astride);                                                                   \
  } while (0)
  if (lper >= 0) {
    if constexpr (D == 1 || D == 2 || D == 4 || D == 8) {
#define G3_GRAD_FAST_PK(PKIND)                                  \
  switch (kind) {                                               \
    case G3_K_SE: G3_GRAD_FAST(G3_K_SE, PKIND); break;          \
    case G3_K_MAT32: G3_GRAD_FAST(G3_K_MAT32, PKIND); break;    \
    default: G3_GRAD_FAST(G3_K_MAT52, PKIND); break;            \
  }
      if (pkind == G3_K_SIN) { G3_GRAD_FAST_PK(G3_K_SIN) }
      else if (pkind == G3_K_SM) { G3_GRAD_FAST_PK(G3_K_SM) }
      else { G3_GRAD_FAST_PK(G3_K_COS) }
#undef G3_GRAD_FAST_PK
    }
  } else {
    switch (kind) {
      case G3_K_SE: G3_GRAD_FAST(G3_K_SE, -1); break;
      case G3_K_OU: G3_GRAD_FAST(G3_K_OU, -1); break;
      case G3_K_MAT32: G3_GRAD_FAST(G3_K_MAT32, -1); break;
      case G3_K_MAT52: G3_GRAD_FAST(G3_K_MAT52, -1); break;
      default: G3_GRAD_FAST(G3_K_RQ, -1); break;
    }
  }
#undef G3_GRAD_FAST
  G3_LAUNCH_CHECK();
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(ns, (unsigned)batch), dim3(256), 0, ctx->stream, partial, nblocks, ns, dout, 64);
  G3_LAUNCH_CHECK();
  g3i_prof_end(ctx, rec);
  std::vector<double> hv((size_t)batch * 64);
  G3_HIP(hipMemcpyAsync(hv.data(), dout, obytes, hipMemcpyDeviceToHost, ctx->stream));
  G3_HIP(hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < batch; ++b) {
    const double* h = hv.data() + (size_t)b * 64;
    double* o = out_host + (size_t)b * map->nslots;
    for (int s = 0; s < map->nslots; ++s) o[s] = 0.0;
    if (map->var[lse] >= 0) o[map->var[lse]] = h[0];
    if (lnoise >= 0 && map->var[lnoise] >= 0) o[map->var[lnoise]] = h[1];
    if (kind == G3_K_RQ && map->alpha[lse] >= 0) o[map->alpha[lse]] = h[2];
    if (map->rate[lse] >= 0)
      for (int k = 0; k < D; ++k) o[map->rate[lse] + k] = h[3 + k];
    if (lper >= 0) {
      if (map->var[lper] >= 0) o[map->var[lper]] = h[D + 3];
      if (map->freq[lper] >= 0)
        for (int k = 0; k < D; ++k) o[map->freq[lper] + k] = h[D + 4 + k];
      if (prate && map->rate[lper] >= 0)
        for (int k = 0; k < D; ++k) o[map->rate[lper] + k] = h[2 * D + 4 + k];
    }
  }
  return G3_OK;
}

static bool same_shape(const g3_kernel_prog* a, const g3_kernel_prog* b) {
  if (a->nleaf != b->nleaf || a->nprod != b->nprod) return false;
  for (int l = 0; l < a->nleaf; ++l) {
    if (a->leaf[l].kind != b->leaf[l].kind || a->leaf[l].ndims != b->leaf[l].ndims) return false;
    for (int k = 0; k < a->leaf[l].ndims; ++k)
      if (a->leaf[l].dims[k] != b->leaf[l].dims[k]) return false;
  }
  for (int q = 0; q < a->nprod; ++q) {
    if (a->prod[q].nfac != b->prod[q].nfac) return false;
    for (int f = 0; f < a->prod[q].nfac; ++f)
      if (a->prod[q].fac[f] != b->prod[q].fac[f]) return false;
  }
  return true;
}

// The gradient kernel generated for the members' common structure (g3_gram_jit.hip::g3_grad_jit): compiled at first use,
// cached; it accumulates g3_grad_layout's standard slots, routed to the caller's map here.  *handled = false: none
// (no hipRTC, too many slots, members of different structure) -- the caller interprets.
static int gram_grad_generated(g3_ctx* ctx, const g3_kernel_prog* progs, int batch, const g3_grad_map* map, const void* X, int64_t N,
                               int64_t ldx, int d, g3_dtype dt, const void* G, int64_t ldg, int64_t gstride, const void* alpha,
                               int64_t astride, double* out_host, bool* handled, int64_t row0, int64_t row1) {
  *handled = false;
  int ns = 0;
  hipFunction_t fn = g3i_grad_jit_function(ctx, &progs[0], d, dt, &ns);
  if (!fn) return G3_OK;
  for (int b = 1; b < batch; ++b)
    if (!same_shape(&progs[0], &progs[b])) return G3_OK;
  const int64_t bi0 = row0 / GG_T, bi1 = (row1 + GG_T - 1) / GG_T;
  const int64_t ntiles = bi1 * (bi1 + 1) / 2 - bi0 * (bi0 + 1) / 2;
  int nblocks = (int)(ntiles < 4096 ? ntiles : 4096);
  if (batch > 1) {
    const int per = (int)((8192 + batch - 1) / batch);
    if (nblocks > per) nblocks = per < 1 ? 1 : per;
  }
  const int ostride = (int)g3_roundup(ns, 32);
  const size_t pbytes = (size_t)batch * nblocks * ns * sizeof(double);
  const size_t obytes = (size_t)batch * ostride * sizeof(double);
  const size_t gbytes = batch > 1 ? (size_t)batch * sizeof(g3_kernel_prog) : 0;
  int rc = g3i_ensure_work(ctx, pbytes + obytes + gbytes);
  if (rc) return rc;
  double* partial = (double*)ctx->work;
  double* dout = (double*)((char*)ctx->work + pbytes);
  const g3_kernel_prog* dprog = nullptr;
  if (batch > 1) {
    g3_kernel_prog* dp_all = (g3_kernel_prog*)((char*)ctx->work + pbytes + obytes);
    G3_HIP(hipMemcpyAsync(dp_all, progs, gbytes, hipMemcpyHostToDevice, ctx->stream));
    dprog = dp_all;
  } else {
    rc = g3i_upload_prog(ctx, progs, 0, &dprog);
    if (rc) return rc;
  }
  struct {
    const void* prog; const void* X; long long N, ldx; const void* G; long long ldg; const void* alpha; double* partial;
    long long row0, row1, gstride, astride;
  } args = {dprog, X, N, ldx, G, ldg, alpha, partial, row0, row1, gstride, astride};
  size_t asz = sizeof(args);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
  int rec = g3i_prof_begin(ctx, G3_TAG_GRAM, (double)batch * (row0 == 0 && row1 == N ? (double)N * (N + 1) / 2 : (double)ntiles * GG_T * GG_T) * g3_esize(dt));
  if (hipModuleLaunchKernel(fn, (unsigned)nblocks, (unsigned)batch, 1, GG_THREADS, 1, 1, 0, ctx->stream, nullptr, cfg) != hipSuccess) {
    (void)hipGetLastError();
    g3i_prof_end(ctx, rec);
    return G3_OK;                  // not handled: the interpreter takes it
  }
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(ns, (unsigned)batch), dim3(256), 0, ctx->stream, partial, nblocks, ns, dout, ostride);
  G3_LAUNCH_CHECK();
  g3i_prof_end(ctx, rec);
  std::vector<double> hv((size_t)batch * ostride);
  G3_HIP(hipMemcpyAsync(hv.data(), dout, obytes, hipMemcpyDeviceToHost, ctx->stream));
  G3_HIP(hipStreamSynchronize(ctx->stream));
  ctx->grad_paths[1] += 1;
  // standard layout -> the caller's map (slot -1: not wanted)
  g3_grad_map stdm;
  if (g3_grad_layout(&progs[0], &stdm) != 0 || stdm.nslots != ns) return G3_ERR_HIP;
  for (int b = 0; b < batch; ++b) {
    const double* h = hv.data() + (size_t)b * ostride;
    double* o = out_host + (size_t)b * map->nslots;
    for (int s2 = 0; s2 < map->nslots; ++s2) o[s2] = 0.0;
    for (int l = 0; l < progs[0].nleaf; ++l) {
      const int nd = progs[0].leaf[l].ndims;
      if (map->var[l] >= 0 && stdm.var[l] >= 0) o[map->var[l]] = h[stdm.var[l]];
      if (map->alpha[l] >= 0 && stdm.alpha[l] >= 0) o[map->alpha[l]] = h[stdm.alpha[l]];
      if (map->rate[l] >= 0 && stdm.rate[l] >= 0)
        for (int k = 0; k < nd; ++k) o[map->rate[l] + k] = h[stdm.rate[l] + k];
      if (map->freq[l] >= 0 && stdm.freq[l] >= 0)
        for (int k = 0; k < nd; ++k) o[map->freq[l] + k] = h[stdm.freq[l] + k];
    }
  }
  *handled = true;
  return G3_OK;
}

// `batch` members (programs of one structure, G and alpha gstride / astride elements apart, out_host batch x nslots) in
// launches that carry the member in grid.y and ONE copy back: a chain row costs no launch and no host round trip of its own.
int g3i_gram_grad_batched(g3_ctx* ctx, const g3_kernel_prog* progs, int batch, const g3_grad_map* map, const void* X, int64_t N,
                          int64_t ldx, int d, g3_dtype dt, const void* G, int64_t ldg, int64_t gstride, const void* alpha,
                          int64_t astride, double* out_host, int64_t row0, int64_t nrows) {
  // nrows < 0: the whole lower triangle; otherwise rows [row0, row0 + nrows) of it, G = those rows (multi-GPU driver)
  const int nslots = map->nslots;
  const int64_t row1 = nrows < 0 ? N : (row0 + nrows < N ? row0 + nrows : N);
  if (nrows < 0) row0 = 0;
  if (nslots == 0 || N == 0 || row1 <= row0) {
    for (size_t s = 0; s < (size_t)nslots * batch; ++s) out_host[s] = 0.0;
    return G3_OK;
  }
  const int generic_only = ctx->tune.grad_interpret;   // G3_GRAD_GENERIC=1 at context creation: always the sum-of-products kernel
  if (!generic_only) {   // var * SE(+ noise) on all columns: register fast path
    bool done = false;
    int r = G3_OK;
#define G3_SE_CASE(DD) case DD: r = gram_grad_se<DD>(ctx, progs, batch, map, X, N, ldx, dt, G, ldg, gstride, alpha, astride, out_host, &done, row0, row1); break
    switch (d) {
      G3_SE_CASE(1);
      G3_SE_CASE(2);
      G3_SE_CASE(3);
      G3_SE_CASE(4);
      G3_SE_CASE(8);
      G3_SE_CASE(16);
      default: break;
    }
#undef G3_SE_CASE
    if (!r && done) ctx->grad_paths[0] += 1;
    if (r || done) return r;
  }
  {   // a kernel generated for this expression's structure (register accumulators, folded leaf formulas)
    bool done = false;
    const int r = gram_grad_generated(ctx, progs, batch, map, X, N, ldx, d, dt, G, ldg, gstride, alpha, astride, out_host, &done, row0, row1);
    if (r || done) return r;
  }
  ctx->grad_paths[2] += 1;
  const int dp = d | 1;
  const size_t fixed = ((size_t)2 * GG_T * dp + 2 * GG_T + (size_t)G3_MAXLEAF * GG_THREADS) * sizeof(double);
  const size_t lds_max = 160 * 1024;
  int window = (int)((lds_max - fixed) / (GG_THREADS * sizeof(double)));
  if (window > 32) window = 32;      // keeps two workgroups per CU for the usual d
  if (window > nslots) window = nslots;
  if (window < 1) return G3_ERR_NOMEM;
  const int64_t bi0 = row0 / GG_T, bi1 = (row1 + GG_T - 1) / GG_T;
  const int64_t ntiles = bi1 * (bi1 + 1) / 2 - bi0 * (bi0 + 1) / 2;
  int nblocks = (int)(ntiles < 2048 ? ntiles : 2048);
  if (batch > 1) {
    const int per = (int)((8192 + batch - 1) / batch);
    if (nblocks > per) nblocks = per < 1 ? 1 : per;
  }
  const size_t pbytes = (size_t)batch * nblocks * window * sizeof(double);
  const int ostride = (int)g3_roundup(nslots, 32);
  const size_t obytes = (size_t)batch * ostride * sizeof(double);
  const size_t gbytes = batch > 1 ? (size_t)batch * sizeof(g3_kernel_prog) : 0;
  int rc = g3i_ensure_work(ctx, pbytes + obytes + gbytes);
  if (rc) return rc;
  double* partial = (double*)ctx->work;
  double* dout = (double*)((char*)ctx->work + pbytes);
  const g3_kernel_prog* dprog = nullptr;
  if (batch > 1) {
    g3_kernel_prog* dp_all = (g3_kernel_prog*)((char*)ctx->work + pbytes + obytes);
    G3_HIP(hipMemcpyAsync(dp_all, progs, gbytes, hipMemcpyHostToDevice, ctx->stream));
    dprog = dp_all;
  } else {
    rc = g3i_upload_prog(ctx, progs, 0, &dprog);
    if (rc) return rc;
  }
  const size_t lds = fixed + (size_t)window * GG_THREADS * sizeof(double);
  if (dt == G3_F64)
    G3_HIP(hipFuncSetAttribute((const void*)gram_grad_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  else
    G3_HIP(hipFuncSetAttribute((const void*)gram_grad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int rec = g3i_prof_begin(ctx, G3_TAG_GRAM, (double)batch * (row0 == 0 && row1 == N ? (double)N * (N + 1) / 2 : (double)ntiles * GG_T * GG_T) * g3_esize(dt));
  const dim3 grid((unsigned)nblocks, (unsigned)batch);
  for (int lo = 0; lo < nslots; lo += window) {
    const int width = nslots - lo < window ? nslots - lo : window;
    if (dt == G3_F64)
      hipLaunchKernelGGL((gram_grad_kernel<double>), grid, dim3(GG_THREADS), lds, ctx->stream, dprog, *map,
                         (const double*)X, N, ldx, d, (const double*)G, ldg, (const double*)alpha, partial, lo, width, row0, row1,
                         gstride, astride);
    else
      hipLaunchKernelGGL((gram_grad_kernel<float>), grid, dim3(GG_THREADS), lds, ctx->stream, dprog, *map,
                         (const float*)X, N, ldx, d, (const float*)G, ldg, (const float*)alpha, partial, lo, width, row0, row1,
                         gstride, astride);
    G3_LAUNCH_CHECK();
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(width, (unsigned)batch), dim3(256), 0, ctx->stream, partial, nblocks, width, dout + lo, ostride);
    G3_LAUNCH_CHECK();
  }
  g3i_prof_end(ctx, rec);
  std::vector<double> hv((size_t)batch * ostride);
  G3_HIP(hipMemcpyAsync(hv.data(), dout, obytes, hipMemcpyDeviceToHost, ctx->stream));
  G3_HIP(hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < batch; ++b)
    for (int s2 = 0; s2 < nslots; ++s2) out_host[(size_t)b * nslots + s2] = hv[(size_t)b * ostride + s2];
  return G3_OK;
}

int g3i_gram_grad(g3_ctx* ctx, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X, int64_t N,
                  int64_t ldx, int d, g3_dtype dt, const void* G, int64_t ldg, const void* alpha, double* out_host,
                  int64_t row0, int64_t nrows) {
  return g3i_gram_grad_batched(ctx, prog, 1, map, X, N, ldx, d, dt, G, ldg, 0, alpha, 0, out_host, row0, nrows);
}

static int check_map(const g3_kernel_prog* prog, const g3_grad_map* map) {
  if (map->nslots < 0 || map->nslots > G3_GRAD_MAXSLOTS) return 1;
  for (int l = 0; l < prog->nleaf; ++l) {
    const int nd = prog->leaf[l].ndims;
    const int32_t v[4] = {map->var[l], map->alpha[l], map->rate[l], map->freq[l]};
    const int w[4] = {1, 1, nd, nd};
    for (int q = 0; q < 4; ++q)
      if (v[q] < -1 || (v[q] >= 0 && v[q] + w[q] > map->nslots)) return 1;
  }
  return 0;
}

extern "C" int g3_gram_grad(g3_ctx* ctx, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X_dev,
                            int64_t N, int64_t ldx, int d, g3_dtype dt, const void* G_dev, int64_t ldg,
                            const void* alpha_dev, double* out_host) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!prog || prog->nleaf < 0 || prog->nleaf > G3_MAXLEAF || prog->nprod < 0 || prog->nprod > G3_MAXPROD) return -2;
  if (!map || check_map(prog, map)) return -3;
  if (!X_dev && N > 0) return -4;
  if (N < 0) return -5;
  if (d < 1 || d > G3_MAXCOLS) return -7;
  if (ldx < d) return -6;
  if (!G_dev && N > 0) return -9;
  if (ldg < N) return -10;
  if (!alpha_dev && N > 0) return -11;
  if (!out_host) return -12;
  return g3i_gram_grad(ctx, prog, map, X_dev, N, ldx, d, dt, G_dev, ldg, alpha_dev, out_host);
}

// rows [row0, row0 + nrows) of the same sum: G_rows_dev holds those rows of K^-1 (columns 0 .. row0 + nrows are read).
// The partial sums of disjoint row ranges add up to g3_gram_grad's; the multi-GPU driver calls this per row block.
extern "C" int g3_gram_grad_rows(g3_ctx* ctx, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X_dev,
                                 int64_t N, int64_t ldx, int d, g3_dtype dt, int64_t row0, int64_t nrows,
                                 const void* G_rows_dev, int64_t ldg, const void* alpha_dev, double* out_host) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!prog || prog->nleaf < 0 || prog->nleaf > G3_MAXLEAF || prog->nprod < 0 || prog->nprod > G3_MAXPROD) return -2;
  if (!map || check_map(prog, map)) return -3;
  if (!X_dev && N > 0) return -4;
  if (N < 0) return -5;
  if (d < 1 || d > G3_MAXCOLS) return -7;
  if (ldx < d) return -6;
  if (row0 < 0 || row0 % GG_T) return -9;
  if (nrows < 0 || row0 + nrows > N) return -10;
  if (!G_rows_dev && nrows > 0) return -11;
  if (ldg < row0 + nrows) return -12;
  if (!alpha_dev && N > 0) return -13;
  if (!out_host) return -14;
  return g3i_gram_grad(ctx, prog, map, X_dev, N, ldx, d, dt, G_rows_dev, ldg, alpha_dev, out_host, row0, nrows);
}

// Fused: K^-1 and alpha from a factor produced by g3_gp_factor, then the hyper-parameter sums.
extern "C" int g3_gp_dlogp(g3_ctx* ctx, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X_dev,
                           int64_t N, int64_t ldx, int d, const void* L_dev, int64_t ldl, const void* invd_dev,
                           const void* a_dev, g3_dtype dt, void* Y_dev, int64_t ldy, void* Kinv_dev, int64_t ldc,
                           void* alpha_dev, double* out_host) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!prog || prog->nleaf < 0 || prog->nleaf > G3_MAXLEAF || prog->nprod < 0 || prog->nprod > G3_MAXPROD) return -2;
  if (!map || check_map(prog, map)) return -3;
  if (!X_dev) return -4;
  if (N <= 0) return -5;
  if (d < 1 || d > G3_MAXCOLS) return -7;
  if (ldx < d) return -6;
  const int64_t Np = g3_roundup(N, G3_LB), al = 16 / (int64_t)g3_esize(dt);
  if (!L_dev) return -8;
  if (ldl < Np || ldl % al) return -9;
  if (!invd_dev) return -10;
  if (!a_dev) return -11;
  if (!Y_dev) return -13;
  if (ldy < Np || ldy % al) return -14;
  if (!Kinv_dev) return -15;
  if (ldc < Np || ldc % al) return -16;
  if (!alpha_dev) return -17;
  if (!out_host) return -18;
  int rc = g3i_reset_info(ctx);
  if (rc) return rc;
  int rec = g3i_prof_begin(ctx, G3_TAG_POTRF, 2.0 * (double)N * N * N / 3.0);
  rc = g3i_potri(ctx, L_dev, Np, ldl, invd_dev, dt, Y_dev, ldy, Kinv_dev, ldc);
  g3i_prof_end(ctx, rec);
  if (rc) return rc;
  // alpha = L^-T a = Y a  (row i of Y dotted with a)
  rc = g3_rows_dot_ss(ctx, Y_dev, Np, Np, ldy, a_dev, dt, alpha_dev, nullptr);
  if (rc) return rc;
  return g3i_gram_grad(ctx, prog, map, X_dev, N, ldx, d, dt, Kinv_dev, ldc, alpha_dev, out_host);
}


// Batched dlogp: `batch` factorisations produced by ONE g3_gp_factor_batched sweep (members kstride
// elements apart in L_dev, diagonal-block inverses Npad*128 apart, a_dev batch x Npad) -- the caller
// pattern of fixed_dlogp / find_MAP restarts (g3py/processes/stochastic.py:554-564: a Python loop
// over single gradients in the reference).  K^-1 of every member comes out of one batched potri
// sweep (grid.y = member in every MFMA GEMM and diagonal-block launch); the O(N^2) kernel-parameter
// sums then run member by member.  Y_dev / Kinv_dev: batch members, kstride apart, ld = ldl.
extern "C" int g3_gp_dlogp_batched(g3_ctx* ctx, const g3_kernel_prog* progs, int batch, const g3_grad_map* map,
                                   const void* X_dev, int64_t N, int64_t ldx, int d, const void* L_dev, int64_t ldl,
                                   int64_t kstride, const void* invd_dev, const void* a_dev, g3_dtype dt, void* Y_dev,
                                   void* Kinv_dev, void* alpha_dev, double* out_host) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!progs) return -2;
  if (batch < 1 || batch > G3_MAX_BATCH) return -3;
  if (!map) return -4;
  for (int b = 0; b < batch; ++b) {
    if (progs[b].nleaf < 0 || progs[b].nleaf > G3_MAXLEAF || progs[b].nprod < 0 || progs[b].nprod > G3_MAXPROD) return -2;
    if (check_map(&progs[b], map)) return -4;
  }
  if (!X_dev) return -5;
  if (N <= 0) return -6;
  if (d < 1 || d > G3_MAXCOLS) return -8;
  if (ldx < d) return -7;
  const int64_t Np = g3_roundup(N, G3_LB), al = 16 / (int64_t)g3_esize(dt);
  const size_t es = g3_esize(dt);
  if (!L_dev) return -9;
  if (ldl < Np || ldl % al) return -10;
  if (kstride < Np * ldl || kstride % al) return -11;
  if (!invd_dev) return -12;
  if (!a_dev) return -13;
  if (!Y_dev) return -15;
  if (!Kinv_dev) return -16;
  if (!alpha_dev) return -17;
  if (!out_host) return -18;
  if (batch > 1) {
    ctx->batch = batch;
    ctx->bstride = kstride;
    ctx->bstride_w = Np * G3_LB;
    ctx->bw_base = (const char*)invd_dev;
    ctx->bw_bytes = (size_t)batch * Np * G3_LB * es;
  }
  // in batch mode: EVERY member's pivot flag (a member whose first factorisation failed was re-run through the jitter
  // schedule on its own, and its flag of the batched sweep must not turn this sweep's launches into no-ops for it)
  int rc = g3i_reset_info(ctx);
  if (rc) { ctx->batch = 0; ctx->bw_base = nullptr; return rc; }
  const int rec = g3i_prof_begin(ctx, G3_TAG_POTRF, 2.0 * (double)batch * (double)N * N * N / 3.0);
  rc = g3i_potri(ctx, L_dev, Np, ldl, invd_dev, dt, Y_dev, ldl, Kinv_dev, ldl);
  g3i_prof_end(ctx, rec);
  ctx->batch = 0;
  ctx->bw_base = nullptr;
  if (rc) return rc;
  // alpha_b = L_b^-T a_b for every member in one launch, then the kernel-parameter sums of all members in launches that
  // carry the member in grid.y and ONE copy back (a chain of 4096 members: 3 launches instead of 12 288 and 4096 host waits)
  rc = g3i_rows_dot_ss_batched(ctx, Y_dev, Np, Np, ldl, a_dev, dt, alpha_dev, nullptr, batch, kstride, Np, Np);
  if (rc) return rc;
  return g3i_gram_grad_batched(ctx, progs, batch, map, X_dev, N, ldx, d, dt, Kinv_dev, ldl, kstride, alpha_dev, Np, out_host);
}

// g3_gp_dlogp_batched with the members given as in g3_gp_factor_batched_fields: one template program plus the doubles that
// differ per member.  (The gradient kernels need each member's parameters on the host side -- fast-path matching -- so the
// programs are expanded here, 6 KB per member, instead of being packed by the binding.)
extern "C" int g3_gp_dlogp_batched_fields(g3_ctx* ctx, const g3_kernel_prog* tmpl, int batch, const double* fields,
                                          const int32_t* offsets, int nfield, const g3_grad_map* map, const void* X_dev,
                                          int64_t N, int64_t ldx, int d, const void* L_dev, int64_t ldl, int64_t kstride,
                                          const void* invd_dev, const void* a_dev, g3_dtype dt, void* Y_dev, void* Kinv_dev,
                                          void* alpha_dev, double* out_host) {
  if (!ctx) return -1;
  if (!tmpl) return -2;
  if (batch < 1 || batch > G3_MAX_BATCH) return -3;
  if (nfield < 0 || nfield > G3_MAX_FIELDS) return -6;
  if (nfield && (!fields || !offsets)) return -4;
  for (int i = 0; i < nfield; ++i)
    if (!g3h_field_offset_ok(offsets[i])) return -5;      // the same rule as g3_gp_factor_batched_fields (g3_host.h)
  std::vector<g3_kernel_prog> progs((size_t)batch, *tmpl);
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < nfield; ++i)
      memcpy((char*)&progs[b] + offsets[i], &fields[(size_t)b * nfield + i], sizeof(double));
  const int rc = g3_gp_dlogp_batched(ctx, progs.data(), batch, map, X_dev, N, ldx, d, L_dev, ldl, kstride, invd_dev, a_dev, dt,
                                     Y_dev, Kinv_dev, alpha_dev, out_host);
  return (rc <= -4 && rc >= -18) ? rc - 3 : rc;
}

extern "C" int g3_grad_path_stats(g3_ctx* ctx, double out_host[3]) {
  if (!ctx) return -1;
  if (!out_host) return -2;
  for (int i = 0; i < 3; ++i) out_host[i] = (double)ctx->grad_paths[i];
  return G3_OK;
}
