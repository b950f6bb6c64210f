// Gram-matrix assembly for g3py's stationary kernels on gfx950.
//
// Replaces Metric.gram + Kernel.cov (g3py/processes/hypers/metrics.py:11-13,
// kernels.py:96-110,192-244,360-487): the reference materialises an n1 x n2 x d broadcast
// tensor and then reduces it; here one 64 x 64 output tile is produced per workgroup from
// two 64 x d input tiles staged in LDS (coalesced loads of the N x d input), every thread
// evaluates the whole kernel expression for its pairs and writes K exactly once, row-wise
// coalesced (HBM-write bound).  tt_to_num (tensors.py:90-92) is fused into the store.
#include "g3_internal.h"
#include "g3_host.h"
#include <stdlib.h>

#define GT 64           // output tile rows

#include "g3_kernel_eval.h"

// number of (leaf, dimension) pairs that need a cos / sin table entry
static __host__ __device__ inline int prog_trig_pairs(const g3_kernel_prog* p) {
  int tt = 0;
  for (int l = 0; l < p->nleaf; ++l)
    if (leaf_has_trig(p->leaf[l].kind)) tt += p->leaf[l].ndims;
  return tt;
}

// Fast path: prog == var * k(all d columns in order) [+ noise on the square diagonal] with k one stationary
// kernel on an ARD metric -- SE, OU, MAT32, MAT52 or RQ, the expressions the reference's examples are built from.
// D and the kind are compile-time: x_j lives in registers, x_i is an LDS broadcast, nothing is interpreted
// (the run-time interpreter costs 36 % on top of the arithmetic).
// (SeParams and the host-side matcher g3h_match_fast live in g3_host.h)
// One workgroup (256 threads) writes a 64-row x 128-column tile; a thread owns two adjacent
// columns (one 16-byte store per row for fp64: a wave writes 1 KiB of one row per instruction)
// and every fourth row.
// Round 5 (profiles/r05_gram.md): the store pattern alone streams at 5.5 TB/s (a dense fill of the same bytes: 5.75);
// what held the kernel at 3.7 TB/s was the per-element control flow of the loop (see the interior loop below), not the
// tile shape: 128-row tiles and non-temporal stores were built and measured (+-3 % in the sweep, noisier) and removed.
#define GTN 128
// FK: g3_kind of the fast path's stationary term, -1 = generic program; PK: g3_kind of its periodic second term
// (G3_K_COS, G3_K_SIN, G3_K_SM) or -1.  The sum  stationary + periodic (+ noise)  is the shape of BASELINE config 3's kernel.
template <typename T, int D, int FK, int PK = -1>
__global__ void __launch_bounds__(256)
gram_kernel(const g3_kernel_prog* __restrict__ prog, SeParams<T, D> se, const T* __restrict__ X1,
            int64_t n1, int64_t ldx1, const T* __restrict__ X2, int64_t n2, int64_t ldx2, int d,
            T* __restrict__ K, int64_t ldk, int64_t n1pad, int64_t n2pad, unsigned flags, int sym, int ntrig,
            int64_t kstride, int64_t diag_off) {
  // grid.z = batch member: its own program (hyper-parameters) and output matrix, same inputs
  constexpr bool SE_FAST = FK >= 0;
  if constexpr (!SE_FAST) prog += blockIdx.z;
  K += (int64_t)blockIdx.z * kstride;
  int64_t bi = blockIdx.y, bj = blockIdx.x;
  constexpr int gt = GT;
  if (flags & G3_GRAM_LOWER) {
    // 1-D grid over the tiles on or below the diagonal only: row-block b has b/2 + 1 column
    // tiles (64-row, 128-column tiles), prefix q(q+1) + r(q+1) for b = 2q + r
    const int64_t id = blockIdx.x;
    int64_t q = (int64_t)((sqrt(1.0 + 4.0 * (double)id) - 1.0) * 0.5);
    while ((q + 1) * (q + 2) <= id) ++q;
    while (q * (q + 1) > id) --q;
    int64_t rem = id - q * (q + 1);
    const int64_t r = rem >= q + 1 ? 1 : 0;
    if (r) rem -= q + 1;
    bi = 2 * q + r;
    bj = rem;
  }
  const int64_t i0 = bi * gt, j0 = bj * GTN;
  if (i0 >= n1pad || j0 >= n2pad) return;
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  const int dp = d | 1;  // odd row stride: conflict-free column-varying reads
  T* xi_s = reinterpret_cast<T*>(smem_g);
  T* xj_s = xi_s + gt * dp;
  const int tid = threadIdx.x;
  for (int e = tid; e < gt * d; e += 256) {
    const int r = e / d, c = e - r * d;
    xi_s[r * dp + c] = (i0 + r < n1) ? X1[(i0 + r) * ldx1 + c] : T(0);
  }
  for (int e = tid; e < GTN * d; e += 256) {
    const int r = e / d, c = e - r * d;
    xj_s[r * dp + c] = (j0 + r < n2) ? X2[(j0 + r) * ldx2 + c] : T(0);
  }
  // cos / sin tables of the periodic leaves for the 64 + 128 points of this tile
  T* trig_s = xj_s + GTN * dp;
  const int tstride = 2 * ntrig + 1;   // odd stride: conflict-free column-varying reads
  __shared__ int toff_s[G3_MAXLEAF];
  if constexpr (!SE_FAST) if (ntrig > 0) {
    if (tid == 0) {
      int o = 0;
      for (int l = 0; l < prog->nleaf; ++l) {
        const bool h = leaf_has_trig(prog->leaf[l].kind);
        toff_s[l] = h ? o : -1;
        if (h) o += prog->leaf[l].ndims;
      }
      for (int l = prog->nleaf; l < G3_MAXLEAF; ++l) toff_s[l] = -1;
    }
    __syncthreads();
    for (int e = tid; e < (gt + GTN) * ntrig; e += 256) {
      const int pnt = e / ntrig, t = e - pnt * ntrig;
      int l = 0;
      while (l + 1 < prog->nleaf && !(toff_s[l] >= 0 && t >= toff_s[l] && t < toff_s[l] + prog->leaf[l].ndims)) ++l;
      const g3_leaf& lf = prog->leaf[l];
      const int k = t - toff_s[l];
      const T x = (pnt < gt ? xi_s[pnt * dp + lf.dims[k]] : xj_s[(pnt - gt) * dp + lf.dims[k]]);
      const T scale = (lf.kind == G3_K_SINC) ? T(2 * G3_PI * G3_PI) : T(2 * G3_PI);
      const T th = scale * (T)lf.freq[k] * x;
      const T sn = sin(th), cs = cos(th);
      trig_s[pnt * tstride + 2 * t] = cs;
      trig_s[pnt * tstride + 2 * t + 1] = sn;
    }
  }
  constexpr int FTS = 2 * D + 1;        // fast path: [cos, sin] of 2 pi freq_k x_k per point, odd row stride
  if constexpr (SE_FAST && PK >= 0) {
    __syncthreads();
    for (int e = tid; e < (gt + GTN) * D; e += 256) {
      const int pnt = e / D, k = e - pnt * D;
      const T x = pnt < gt ? xi_s[pnt * dp + k] : xj_s[(pnt - gt) * dp + k];
      const T th = se.f[k] * x;
      trig_s[pnt * FTS + 2 * k] = cos(th);
      trig_s[pnt * FTS + 2 * k + 1] = sin(th);
    }
  }
  __syncthreads();
  const int tx = tid & 63, ty = tid >> 6;  // column pair within tile, row phase
  const int64_t ja = j0 + 2 * tx;
  if (ja >= n2pad) return;
  const bool two = (ja + 1 < n2pad);
  const T* xja = xj_s + (2 * tx) * dp;
  const T* xjb = xja + dp;
  T xra[D], xrb[D];
  T cja[D], sja[D], cjb[D], sjb[D];
  if (SE_FAST) {
#pragma unroll
    for (int c = 0; c < D; ++c) { xra[c] = xja[c]; xrb[c] = xjb[c]; }
    if constexpr (PK >= 0) {
      const T* ta = trig_s + (gt + 2 * tx) * FTS;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        cja[c] = ta[2 * c]; sja[c] = ta[2 * c + 1];
        cjb[c] = ta[FTS + 2 * c]; sjb[c] = ta[FTS + 2 * c + 1];
      }
    }
  }
  const bool scr = (flags & G3_GRAM_SCRUB) != 0;
  const bool eye = (flags & G3_GRAM_PAD_EYE) != 0;
  const bool vec_ok = two && ((ldk & 1) == 0) && ((reinterpret_cast<uintptr_t>(K) & (2 * sizeof(T) - 1)) == 0);
  typedef T vec2 __attribute__((ext_vector_type(2)));
  // the fast path's value for (tile row rr, column ja + q): var * k(d) [+ | *] the periodic term; no noise, no scrub
  auto fast_value = [&](int rr, int q) -> T {
    const T* xi = xi_s + rr * dp;
    T dd = T(0);
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const T dx = xi[c] - (q ? xrb[c] : xra[c]);
      if constexpr (FK == G3_K_OU) dd += fabs(dx) * se.w[c];      // metrics.py:89-91
      else dd += (dx * dx) * se.w[c];                             // metrics.py:100-102
    }
    T kv;
    if constexpr (FK == G3_K_MAT32) {
      const T s3 = sqrt(T(3) * dd);
      kv = (T(1) + s3) * g3_exp(-s3);
    } else if constexpr (FK == G3_K_MAT52) {
      const T s5 = sqrt(T(5) * dd);
      kv = (T(1) + s5 + dd * T(5.0 / 3.0)) * g3_exp(-s5);
    } else if constexpr (FK == G3_K_RQ) {
      kv = pow(T(1) + dd / se.alpha, -se.alpha);
    } else {
      kv = g3_exp(-dd);
    }
    T val = se.var * kv;
    if constexpr (PK >= 0) {
      const T* ti = trig_s + rr * FTS;
      T pr = T(1), sm = T(0);
#pragma unroll
      for (int c = 0; c < D; ++c) {   // cos(theta_i - theta_j), angle-difference identity (as the generic path)
        const T cd = ti[2 * c] * (q ? cjb[c] : cja[c]) + ti[2 * c + 1] * (q ? sjb[c] : sja[c]);
        if constexpr (PK == G3_K_SIN) {
          sm += (T(0.5) * (T(1) - cd)) * se.pr[c];            // sin^2(pi f dx) = (1 - cos(2 pi f dx)) / 2, kernels.py:471-472
        } else {
          pr *= cd;                                           // COS, SM: kernels.py:466-467, 486-487
          if constexpr (PK == G3_K_SM) {
            const T dx = xi[c] - (q ? xrb[c] : xra[c]);
            sm += (dx * dx) * se.pr[c];
          }
        }
      }
      T pv;
      if constexpr (PK == G3_K_COS) pv = se.pvar * pr;
      else if constexpr (PK == G3_K_SIN) pv = se.pvar * exp(T(2) * sm);   // positive exponent, as written in the reference
      else pv = se.pvar * (exp(T(-2 * G3_PI * G3_PI) * sm) * pr);
      val = se.mul ? val * pv : val + pv;       // KernelProd (the locally periodic form) or KernelSum
    }
    return val;
  };
  if constexpr (SE_FAST) {
    // Interior tiles (every row and column inside the matrix: all but the last tile row / column) take a loop without
    // per-element control flow -- up to four rows x two columns of independent exp chains in flight, the noise added by a
    // select and only in tiles the diagonal crosses, tt_to_num as ONE test per eight values (non-finite values are the
    // exception).  The per-element branches of the general loop below serialise those chains: SE, d = 4, N = 32768 in the
    // headline step 1.15 ms (3.7 TB/s) -> 0.86 ms (5.0 TB/s; profiles/r05_gram.md).  Same formulas in the same order (the compiler places its fused multiply-adds per loop: equal to rounding).
    if (vec_ok && i0 + gt <= n1 && j0 + GTN <= n2) {
      const int64_t dlo = i0 + diag_off;
      const bool touches = sym && dlo < j0 + GTN && j0 < dlo + gt;      // (uniform)
      // rows in flight per thread: four, fewer where the periodic term's cos / sin registers (4 D) leave no room
      constexpr int UR = PK < 0 ? 4 : (D <= 2 ? 2 : 1);
      for (int r0 = ty; r0 < gt; r0 += 4 * UR) {
        T v[UR][2];
#pragma unroll
        for (int u = 0; u < UR; ++u)
#pragma unroll
          for (int q = 0; q < 2; ++q) v[u][q] = fast_value(r0 + 4 * u, q);
        if (touches) {
#pragma unroll
          for (int u = 0; u < UR; ++u)
#pragma unroll
            for (int q = 0; q < 2; ++q) v[u][q] += (dlo + r0 + 4 * u == ja + q) ? se.noise : T(0);
        }
        if (scr) {
          bool bad = false;
#pragma unroll
          for (int u = 0; u < UR; ++u)
#pragma unroll
            for (int q = 0; q < 2; ++q) bad |= !__builtin_isfinite(v[u][q]);
          if (bad) {
#pragma unroll
            for (int u = 0; u < UR; ++u)
#pragma unroll
              for (int q = 0; q < 2; ++q) v[u][q] = scrub(v[u][q]);
          }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
          *reinterpret_cast<vec2*>(K + (i0 + r0 + 4 * u) * ldk + ja) = vec2{v[u][0], v[u][1]};
        }
      }
      return;
    }
  }
#pragma unroll 4
  for (int rr = ty; rr < gt; rr += 4) {
    const int64_t i = i0 + rr;
    if (i >= n1pad) break;
    T v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int64_t j = ja + q;
      if (i < n1 && j < n2) {
        const bool dg = sym && (i + diag_off == j);   // diag_off: row offset of a row block of a square matrix
        if (SE_FAST) {
          v[q] = fast_value(rr, q);
          if (dg) v[q] += se.noise;
        } else {
          if (ntrig > 0)
            v[q] = prog_eval<T>(prog, xi_s + rr * dp, q ? xjb : xja, dg, sym != 0, trig_s + rr * tstride,
                                trig_s + (gt + 2 * tx + q) * tstride, toff_s);
          else
            v[q] = prog_eval<T>(prog, xi_s + rr * dp, q ? xjb : xja, dg, sym != 0);
        }
        if (scr) v[q] = scrub(v[q]);
      } else {
        v[q] = (eye && i + diag_off == j) ? T(1) : T(0);
      }
    }
    T* p = K + i * ldk + ja;
    if (vec_ok) {
      *reinterpret_cast<vec2*>(p) = vec2{v[0], v[1]};
    } else {
      p[0] = v[0];
      if (two) p[1] = v[1];
    }
  }
}

template <typename T>
__global__ void gram_diag_kernel(const g3_kernel_prog* __restrict__ prog, const T* __restrict__ X,
                                 int64_t n, int64_t ldx, int d, T* __restrict__ out) {
  // diag(Kernel.cov(X)): the square-case diagonal, so NOISE / WN contribute
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  T* xs = reinterpret_cast<T*>(smem_g);
  const int dp = d | 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  T* xi = xs + threadIdx.x * dp;
  if (i < n)
    for (int c = 0; c < d; ++c) xi[c] = X[i * ldx + c];
  if (i < n) out[i] = prog_eval<T>(prog, xi, xi, true, true);
}

int g3i_upload_prog(g3_ctx* ctx, const g3_kernel_prog* prog, int /*slot*/, const g3_kernel_prog** dptr) {
  // A ring of pinned + device slots keeps the host ahead of the stream: the launch that read the
  // previous slot is already enqueued when the next program arrives, so an event recorded now sits
  // behind it, and a slot is only waited for when the ring has wrapped onto work still in flight.
  bool must_wait = false;
  int mark = -1;
  const int s = g3h_ring_take(ctx->prog_busy, G3_PROG_SLOTS, &ctx->prog_next, &ctx->prog_last, &must_wait, &mark);
  if (mark >= 0) G3_HIP(hipEventRecord(ctx->prog_ev[mark], ctx->prog_stream));
  if (must_wait) G3_HIP(hipEventSynchronize(ctx->prog_ev[s]));
  memcpy(&ctx->h_prog[s], prog, sizeof(g3_kernel_prog));
  G3_HIP(hipMemcpyAsync(&ctx->d_prog[s], &ctx->h_prog[s], sizeof(g3_kernel_prog),
                        hipMemcpyHostToDevice, ctx->stream));
  ctx->prog_stream = ctx->stream;
  *dptr = &ctx->d_prog[s];
  return G3_OK;
}

int g3i_validate_prog(const g3_kernel_prog* p, int d);
static int validate_prog(const g3_kernel_prog* p, int d) { return g3i_validate_prog(p, d); }
int g3i_validate_prog(const g3_kernel_prog* p, int d) { return g3h_validate_prog(p, d); }

static dim3 gram_grid(int64_t n1pad, int64_t n2pad, unsigned flags) {
  const int64_t tr = (n1pad + GT - 1) / GT, tc = (n2pad + GTN - 1) / GTN;
  if (flags & G3_GRAM_LOWER) {   // tiles on or below the diagonal: sum_b (b/2 + 1), b < tr
    const int64_t q = tr / 2, r = tr % 2;
    return dim3((unsigned)(q * (q + 1) + r * (q + 1)));
  }
  return dim3((unsigned)tc, (unsigned)tr);
}

// row offset of the block being built by g3_gram_rows (0 for every other entry point)
static inline int64_t ctx_diag_off(const g3_ctx* ctx) { return ctx->gram_diag_off; }

template <typename T, int D>
static int launch_gram_fast(g3_ctx* ctx, int kind, int pk, const SeParams<T, D>& se, const T* X1, int64_t n1, int64_t ldx1,
                            const T* X2, int64_t n2, int64_t ldx2, T* K, int64_t ldk, int64_t n1pad,
                            int64_t n2pad, unsigned flags, int sym) {
  dim3 grid = gram_grid(n1pad, n2pad, flags);
  ctx->gram_paths[0] += 1;
  const size_t lds = (GT + GTN) * ((D | 1) + (pk >= 0 ? 2 * D + 1 : 0)) * sizeof(T);
#define G3_GRAM_FAST_LAUNCH(KIND, PKIND)                                                                            \
  hipLaunchKernelGGL((gram_kernel<T, D, KIND, PKIND>), grid, dim3(256), lds, ctx->stream, (const g3_kernel_prog*)nullptr, \
                     se, X1, n1, ldx1, X2, n2, ldx2, D, K, ldk, n1pad, n2pad, flags, sym, 0, (int64_t)0,           \
                     ctx_diag_off(ctx))
  if (pk >= 0) {
    if constexpr (D == 1 || D == 2 || D == 4 || D == 8) {
#define G3_GRAM_FAST_PK(PKIND)                                             \
      switch (kind) {                                                        \
        case G3_K_SE: G3_GRAM_FAST_LAUNCH(G3_K_SE, PKIND); break;            \
        case G3_K_MAT32: G3_GRAM_FAST_LAUNCH(G3_K_MAT32, PKIND); break;      \
        default: G3_GRAM_FAST_LAUNCH(G3_K_MAT52, PKIND); break;              \
      }
      if (pk == G3_K_SIN) { G3_GRAM_FAST_PK(G3_K_SIN) }
      else if (pk == G3_K_SM) { G3_GRAM_FAST_PK(G3_K_SM) }
      else { G3_GRAM_FAST_PK(G3_K_COS) }
#undef G3_GRAM_FAST_PK
    }
  } else {
    switch (kind) {
      case G3_K_SE: G3_GRAM_FAST_LAUNCH(G3_K_SE, -1); break;
      case G3_K_OU: G3_GRAM_FAST_LAUNCH(G3_K_OU, -1); break;
      case G3_K_MAT32: G3_GRAM_FAST_LAUNCH(G3_K_MAT32, -1); break;
      case G3_K_MAT52: G3_GRAM_FAST_LAUNCH(G3_K_MAT52, -1); break;
      default: G3_GRAM_FAST_LAUNCH(G3_K_RQ, -1); break;
    }
  }
#undef G3_GRAM_FAST_LAUNCH
  G3_LAUNCH_CHECK();
  return G3_OK;
}

template <typename T>
static int gram_t(g3_ctx* ctx, const g3_kernel_prog* prog, const T* X1, int64_t n1, int64_t ldx1,
                  const T* X2, int64_t n2, int64_t ldx2, int d, T* K, int64_t ldk, int64_t n1pad,
                  int64_t n2pad, unsigned flags, int sym) {
  {
    const int nofast = ctx->tune.gram_interpret;      // G3_GRAM_NOFAST=1 at context creation: always interpret (A/B runs)
    int fk, pk;
    SeParams<T, 1> s1; SeParams<T, 2> s2; SeParams<T, 3> s3; SeParams<T, 4> s4; SeParams<T, 8> s8; SeParams<T, 16> s16;
#define G3_TRY_FAST(DD, SS)                                                                                   \
    if (!nofast && (fk = g3h_match_fast<T, DD>(prog, d, &SS, &pk)) >= 0)                                          \
      return launch_gram_fast<T, DD>(ctx, fk, pk, SS, X1, n1, ldx1, X2, n2, ldx2, K, ldk, n1pad, n2pad, flags, sym)
    G3_TRY_FAST(1, s1);
    G3_TRY_FAST(2, s2);
    G3_TRY_FAST(3, s3);
    G3_TRY_FAST(4, s4);
    G3_TRY_FAST(8, s8);
    G3_TRY_FAST(16, s16);
#undef G3_TRY_FAST
  }
  const g3_kernel_prog* dprog;
  int rc = g3i_upload_prog(ctx, prog, 0, &dprog);
  if (rc) return rc;
  dim3 grid = gram_grid(n1pad, n2pad, flags);
  // a kernel generated for this expression's structure (compiled at first use, g3_gram_jit.hip) ...
  if (g3i_gram_jit(ctx, prog, dprog, 1, X1, n1, ldx1, X2, n2, ldx2, d, sizeof(T) == 8 ? G3_F64 : G3_F32, K, ldk, n1pad, n2pad, flags, sym,
                   (int64_t)0, ctx_diag_off(ctx), grid) == 0)
    return G3_OK;
  // ... or the interpreter.  Periodic leaves: per-tile cos / sin tables unless there are too many for the LDS
  ctx->gram_paths[2] += 1;
  int ntrig = prog_trig_pairs(prog);
  if (ntrig > 16) ntrig = 0;
  const size_t lds = (GT + GTN) * ((d | 1) + (ntrig ? 2 * ntrig + 1 : 0)) * sizeof(T);
  SeParams<T, 1> dummy{};
  hipLaunchKernelGGL((gram_kernel<T, 1, -1, -1>), grid, dim3(256), lds, ctx->stream, dprog, dummy, X1, n1,
                     ldx1, X2, n2, ldx2, d, K, ldk, n1pad, n2pad, flags, sym, ntrig, (int64_t)0, ctx_diag_off(ctx));
  G3_LAUNCH_CHECK();
  return G3_OK;
}

// `batch` square Gram matrices of the same inputs, member b from dprogs[b] (device array, all of
// one structure) into K + b * kstride: the generic kernel with grid.z = batch
int g3i_gram_batched(g3_ctx* ctx, const g3_kernel_prog* dprogs, const g3_kernel_prog* first_host, int batch,
                     const void* X, int64_t n, int64_t ldx, int d, g3_dtype dt, void* K, int64_t ldk,
                     int64_t kstride, int64_t npad, unsigned flags) {
  dim3 grid = gram_grid(npad, npad, flags);
  grid.z = (unsigned)batch;
  if (g3i_gram_jit(ctx, first_host, dprogs, batch, X, n, ldx, X, n, ldx, d, dt, K, ldk, npad, npad, flags, 1, kstride, (int64_t)0, grid) == 0)
    return G3_OK;
  ctx->gram_paths[2] += 1;
  int ntrig = prog_trig_pairs(first_host);
  if (ntrig > 16) ntrig = 0;
  if (dt == G3_F64) {
    const size_t lds = (GT + GTN) * ((d | 1) + (ntrig ? 2 * ntrig + 1 : 0)) * sizeof(double);
    SeParams<double, 1> dummy{};
    hipLaunchKernelGGL((gram_kernel<double, 1, -1, -1>), grid, dim3(256), lds, ctx->stream, dprogs, dummy,
                       (const double*)X, n, ldx, (const double*)X, n, ldx, d, (double*)K, ldk, npad, npad, flags, 1,
                       ntrig, kstride, (int64_t)0);
  } else {
    const size_t lds = (GT + GTN) * ((d | 1) + (ntrig ? 2 * ntrig + 1 : 0)) * sizeof(float);
    SeParams<float, 1> dummy{};
    hipLaunchKernelGGL((gram_kernel<float, 1, -1, -1>), grid, dim3(256), lds, ctx->stream, dprogs, dummy,
                       (const float*)X, n, ldx, (const float*)X, n, ldx, d, (float*)K, ldk, npad, npad, flags, 1,
                       ntrig, kstride, (int64_t)0);
  }
  G3_LAUNCH_CHECK();
  return G3_OK;
}

int g3i_validate_prog(const g3_kernel_prog* p, int d);

extern "C" int g3_gram(g3_ctx* ctx, const g3_kernel_prog* prog, const void* X1, int64_t n1, int64_t ldx1,
                       const void* X2, int64_t n2, int64_t ldx2, int d, g3_dtype dt, void* K, int64_t ldk,
                       int64_t n1pad, int64_t n2pad, unsigned flags) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!prog) return -2;
  if (!X1) return -3;
  if (n1 < 0) return -4;
  const int sym = (X2 == nullptr);
  if (sym) { X2 = X1; n2 = n1; ldx2 = ldx1; }
  if (n2 < 0) return -7;
  if (d < 1 || d > G3_MAXCOLS) return -9;
  if (ldx1 < d) return -5;
  if (ldx2 < d) return -8;
  if (validate_prog(prog, d)) return -2;
  if (!K) return -11;
  if (n1pad < n1) return -13;
  if (n2pad < n2 && !sym) return -14;   // symmetric case: n2pad < n1 limits the columns written
  if (ldk < n2pad) return -12;
  if (!sym && (flags & G3_GRAM_LOWER)) return -15;
  if (n1pad == 0 || n2pad == 0) return G3_OK;
  if (dt == G3_F64)
    return gram_t<double>(ctx, prog, (const double*)X1, n1, ldx1, (const double*)X2, n2, ldx2, d, (double*)K,
                          ldk, n1pad, n2pad, flags, sym);
  return gram_t<float>(ctx, prog, (const float*)X1, n1, ldx1, (const float*)X2, n2, ldx2, d, (float*)K, ldk,
                       n1pad, n2pad, flags, sym);
}

// Rows [row0, row0 + nrows) and columns [0, row0 + nrows) of the SQUARE covariance of X (the
// row-block layout of the multi-GPU driver): NOISE / WN act on the true diagonal i + row0 == j,
// rows / columns beyond N carry the identity (G3_GRAM_PAD_EYE) or zero.
extern "C" int g3_gram_rows(g3_ctx* ctx, const g3_kernel_prog* prog, const void* X, int64_t N, int64_t ldx, int d,
                            int64_t row0, int64_t nrows, g3_dtype dt, void* K, int64_t ldk, unsigned flags) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!prog) return -2;
  if (!X) return -3;
  if (N < 0) return -4;
  if (d < 1 || d > G3_MAXCOLS) return -6;
  if (ldx < d) return -5;
  if (validate_prog(prog, d)) return -2;
  if (row0 < 0) return -7;
  if (nrows < 0) return -8;
  if (!K) return -10;
  if (ldk < row0 + nrows) return -11;
  if (flags & G3_GRAM_LOWER) return -12;
  if (nrows == 0) return G3_OK;
  const size_t es = g3_esize(dt);
  const int64_t n1 = row0 < N ? (N - row0 < nrows ? N - row0 : nrows) : 0;
  const int64_t n2 = N < row0 + nrows ? N : row0 + nrows;
  const void* X1 = (const char*)X + (size_t)(row0 < N ? row0 : 0) * ldx * es;
  ctx->gram_diag_off = row0;
  int rc;
  if (dt == G3_F64)
    rc = gram_t<double>(ctx, prog, (const double*)X1, n1, ldx, (const double*)X, n2, ldx, d, (double*)K, ldk, nrows,
                        row0 + nrows, flags, 1);
  else
    rc = gram_t<float>(ctx, prog, (const float*)X1, n1, ldx, (const float*)X, n2, ldx, d, (float*)K, ldk, nrows,
                       row0 + nrows, flags, 1);
  ctx->gram_diag_off = 0;
  return rc;
}

extern "C" int g3_gram_diag(g3_ctx* ctx, const g3_kernel_prog* prog, const void* X, int64_t n, int64_t ldx,
                            int d, g3_dtype dt, void* diag) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!prog) return -2;
  if (!X) return -3;
  if (n < 0) return -4;
  if (d < 1 || d > G3_MAXCOLS) return -6;
  if (ldx < d) return -5;
  if (validate_prog(prog, d)) return -2;
  if (!diag) return -8;
  if (n == 0) return G3_OK;
  const g3_kernel_prog* dprog;
  int rc = g3i_upload_prog(ctx, prog, 1, &dprog);
  if (rc) return rc;
  const unsigned nb = (unsigned)((n + 63) / 64);
  if (dt == G3_F64)
    hipLaunchKernelGGL((gram_diag_kernel<double>), dim3(nb), dim3(64), 64 * (d | 1) * sizeof(double),
                       ctx->stream, dprog, (const double*)X, n, ldx, d, (double*)diag);
  else
    hipLaunchKernelGGL((gram_diag_kernel<float>), dim3(nb), dim3(64), 64 * (d | 1) * sizeof(float),
                       ctx->stream, dprog, (const float*)X, n, ldx, d, (float*)diag);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

extern "C" int g3_gram_path_stats(g3_ctx* ctx, double out_host[3]) {
  if (!ctx) return -1;
  if (!out_host) return -2;
  for (int i = 0; i < 3; ++i) out_host[i] = (double)ctx->gram_paths[i];
  return G3_OK;
}
