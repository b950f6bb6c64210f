// Gram kernels generated for a kernel EXPRESSION at first use (round 4).
//
// g3_gram.hip carries compile-time variants for the shapes the reference's examples are built from (one stationary
// kernel (+ noise) (+ or x one periodic term), a table of (kind, d) instantiations) and an interpreter for everything
// else -- KernelSum / KernelProd / KernelScale / KernelShift trees of any leaves (kernels.py:192-244), column subsets
// (hypers/__init__.py:55-83), SINC, WN.  Interpreting costs 2-3x on the Gram (310 vector + 226 scalar instructions per
// entry, most of them the per-dimension address arithmetic of the walk through the program).  What is
// fast is a kernel in which the STRUCTURE of the expression -- leaf kinds, the columns each leaf uses, which leaves
// multiply in which product term -- is compile-time and only the hyper-parameters are data.  A run-time expression has
// no such kernel ahead of time; so it is compiled when the expression is first seen: the source below (the same tile
// geometry, store logic and leaf formulas as g3_gram.hip) is specialised by a generated header of constexpr tables,
// compiled for gfx950 by hipRTC (dlopen'ed: a build without libhiprtc simply keeps interpreting), loaded as a module and
// cached per (device, structure, dtype).  No second backend: the same HIP source language, the same device.
#include "g3_internal.h"
#include "g3_host.h"

#include <dlfcn.h>

#include <map>
#include <mutex>
#include <string>

namespace {

const char* kJitSource = R"JIT(
typedef JT T;
typedef long long i64;
#define GT 64
#define GTN 128
#define G3_PI 3.14159265358979323846
enum { K_SE = 0, K_OU = 1, K_MAT32 = 2, K_MAT52 = 3, K_RQ = 4, K_COS = 5, K_SIN = 6, K_SINC = 7, K_SM = 8, K_NOISE = 9, K_WN = 10 };
struct jleaf { int kind; int ndims; int dims[JMAXD]; double var; double alpha; double rate[JMAXD]; double freq[JMAXD]; };
struct jprod { double coef; int nfac; int fac[JMAXFAC]; int pad[3]; };
struct jprog { int nleaf; int nprod; double shift; jleaf leaf[JMAXLEAF]; jprod prod[JMAXPROD]; };
static_assert(sizeof(jprog) == JPROG_BYTES, "g3_kernel_prog layout");

JTABLES

__device__ __forceinline__ T scrub(T v) {        // tt_to_num: NaN -> 0, +-Inf -> 1e10 (tensors.py:90-92)
  if (v != v) return T(0);
  if (__builtin_isinf(v)) return (T)1e10f;
  return v;
}

// one leaf, structure compile-time (L), hyper-parameters from the program; ti / tj: the [cos, sin] rows of the two points
template <int L>
__device__ __forceinline__ T leaf_eval(const jleaf& lf, const T* xi, const T* xj, bool diag_sym, bool sym, const T* ti, const T* tj) {
  constexpr int kind = jkind[L], nd = jnd[L], to = jtoff[L];
  const T var = (T)lf.var;
  if constexpr (kind == K_NOISE) {
    return diag_sym ? var : T(0);
  } else if constexpr (kind == K_WN) {
    if (sym) return diag_sym ? var : T(0);
    T cnt = T(0);
#pragma unroll
    for (int k = 0; k < nd; ++k) cnt += (xi[jdims[L][k]] - xj[jdims[L][k]] == T(0)) ? T(1) : T(0);
    return var * cnt;
  } else if constexpr (kind == K_SE || kind == K_MAT32 || kind == K_MAT52 || kind == K_RQ) {
    T d = T(0);
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const T dx = xi[jdims[L][k]] - xj[jdims[L][k]];
      const T r = (T)lf.rate[k];
      d += (dx * dx) * (T(0.5) * r * r);                       // ARD_L2, metrics.py:100-102
    }
    if constexpr (kind == K_SE) return var * exp(-d);
    if constexpr (kind == K_MAT32) { const T s = sqrt(T(3) * d); return var * ((T(1) + s) * exp(-s)); }
    if constexpr (kind == K_MAT52) { const T s = sqrt(T(5) * d); return var * ((T(1) + s + T(5) * d / T(3)) * exp(-s)); }
    const T al = (T)lf.alpha;
    return var * pow(T(1) + d / al, -al);
  } else if constexpr (kind == K_OU) {
    T d = T(0);
#pragma unroll
    for (int k = 0; k < nd; ++k) d += fabs(xi[jdims[L][k]] - xj[jdims[L][k]]) * (T)lf.rate[k];   // ARD_L1, metrics.py:89-91
    return var * exp(-d);
  } else if constexpr (kind == K_COS) {
    T p = T(1);
#pragma unroll
    for (int k = 0; k < nd; ++k) p *= ti[2 * (to + k)] * tj[2 * (to + k)] + ti[2 * (to + k) + 1] * tj[2 * (to + k) + 1];
    return var * p;
  } else if constexpr (kind == K_SIN) {
    T s = T(0);
#pragma unroll
    for (int k = 0; k < nd; ++k)      // sin^2(pi f dx) = (1 - cos(2 pi f dx)) / 2
      s += (T(0.5) * (T(1) - (ti[2 * (to + k)] * tj[2 * (to + k)] + ti[2 * (to + k) + 1] * tj[2 * (to + k) + 1]))) * (T)lf.rate[k];
    return var * exp(T(2) * s);       // positive exponent, as written at kernels.py:472
  } else if constexpr (kind == K_SINC) {
    T p = T(1);
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const T dx = xi[jdims[L][k]] - xj[jdims[L][k]];
      const T f = (T)lf.freq[k];
      const T v = sin(T(2 * G3_PI * G3_PI) * dx * f) / (T(2 * G3_PI * G3_PI) * f * dx);   // direct: theta's rounding would be amplified by 1 / dx
      p *= (dx != T(0)) ? v : T(1);
    }
    return var * p;
  } else if constexpr (kind == K_SM) {
    T s = T(0), p = T(1);
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const T dx = xi[jdims[L][k]] - xj[jdims[L][k]];
      const T r = (T)lf.rate[k];
      s += (dx * dx) * (r * r);
      p *= ti[2 * (to + k)] * tj[2 * (to + k)] + ti[2 * (to + k) + 1] * tj[2 * (to + k) + 1];
    }
    return var * (exp(T(-2 * G3_PI * G3_PI) * s) * p);
  } else {
    return T(0);
  }
}

// the value of leaf L for this pair, by compile-time dispatch on the leaf index of a product's factor
template <int L>
__device__ __forceinline__ T leaf_by_index(int l, const jprog* prog, const T* xi, const T* xj, bool dg, bool sym, const T* ti, const T* tj) {
  if constexpr (L >= JNL) {
    return T(0);
  } else {
    if (l == L) return leaf_eval<L>(prog->leaf[L], xi, xj, dg, sym, ti, tj);
    return leaf_by_index<L + 1>(l, prog, xi, xj, dg, sym, ti, tj);
  }
}

__device__ __forceinline__ T prog_eval(const jprog* __restrict__ prog, const T* xi, const T* xj, bool dg, bool sym, const T* ti, const T* tj) {
  T acc = (T)prog->shift;
#pragma unroll
  for (int p = 0; p < JNP; ++p) {
    T v = (T)prog->prod[p].coef;
#pragma unroll
    for (int f = 0; f < JMAXFAC; ++f)
      if (f < jnfac[p]) v *= leaf_by_index<0>(jfac[p][f], prog, xi, xj, dg, sym, ti, tj);    // jfac is constexpr: the chain folds
    acc += v;
  }
  return acc;
}

extern "C" __global__ void __launch_bounds__(256)
g3_gram_jit(const jprog* __restrict__ prog, const T* __restrict__ X1, i64 n1, i64 ldx1, const T* __restrict__ X2, i64 n2, i64 ldx2,
            T* __restrict__ K, i64 ldk, i64 n1pad, i64 n2pad, unsigned flags, int sym, i64 kstride, i64 diag_off) {
  prog += blockIdx.z;                       // grid.z = batch member: its own hyper-parameters and output, same inputs
  K += (i64)blockIdx.z * kstride;
  i64 bi = blockIdx.y, bj = blockIdx.x;
  if (flags & 1u) {                         // G3_GRAM_LOWER: 1-D grid over the tiles on or below the diagonal (g3_gram.hip)
    const i64 id = blockIdx.x;
    i64 q = (i64)((sqrt(1.0 + 4.0 * (double)id) - 1.0) * 0.5);
    while ((q + 1) * (q + 2) <= id) ++q;
    while (q * (q + 1) > id) --q;
    i64 rem = id - q * (q + 1);
    const i64 r = rem >= q + 1 ? 1 : 0;
    if (r) rem -= q + 1;
    bi = 2 * q + r;
    bj = rem;
  }
  const i64 i0 = bi * GT, j0 = bj * GTN;
  if (i0 >= n1pad || j0 >= n2pad) return;
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  constexpr int dp = JD | 1;                // odd row stride: conflict-free column-varying reads
  constexpr int tstride = 2 * JNTRIG + 1;
  T* xi_s = reinterpret_cast<T*>(smem_g);
  T* xj_s = xi_s + GT * dp;
  T* trig_s = xj_s + GTN * dp;
  const int tid = threadIdx.x;
  for (int e = tid; e < GT * JD; e += 256) {
    const int r = e / JD, c = e - r * JD;
    xi_s[r * dp + c] = (i0 + r < n1) ? X1[(i0 + r) * ldx1 + c] : T(0);
  }
  for (int e = tid; e < GTN * JD; e += 256) {
    const int r = e / JD, c = e - r * JD;
    xj_s[r * dp + c] = (j0 + r < n2) ? X2[(j0 + r) * ldx2 + c] : T(0);
  }
  if constexpr (JNTRIG > 0) {
    __syncthreads();
    // [cos, sin] of 2 pi freq x for every (periodic leaf, dimension) pair and the 64 + 128 points of the tile
    for (int e = tid; e < (GT + GTN) * JNTRIG; e += 256) {
      const int pnt = e / JNTRIG, t = e - pnt * JNTRIG;
      const int l = jtleaf[t], k = jtk[t];
      const T x = pnt < GT ? xi_s[pnt * dp + jtcol[t]] : xj_s[(pnt - GT) * dp + jtcol[t]];
      const T th = T(2 * G3_PI) * (T)prog->leaf[l].freq[k] * x;
      trig_s[pnt * tstride + 2 * t] = cos(th);
      trig_s[pnt * tstride + 2 * t + 1] = sin(th);
    }
  }
  __syncthreads();
  const int tx = tid & 63, ty = tid >> 6;   // column pair within the tile, row phase
  const i64 ja = j0 + 2 * tx;
  if (ja >= n2pad) return;
  const bool two = (ja + 1 < n2pad);
  T xra[JD], xrb[JD];                       // the thread's two points x_j in registers
#pragma unroll
  for (int c = 0; c < JD; ++c) { xra[c] = xj_s[(2 * tx) * dp + c]; xrb[c] = xj_s[(2 * tx + 1) * dp + c]; }
  const bool scr = (flags & 2u) != 0, eye = (flags & 4u) != 0;
  const bool vec_ok = two && ((ldk & 1) == 0) && ((reinterpret_cast<unsigned long long>(K) & (2 * sizeof(T) - 1)) == 0);
  typedef T vec2 __attribute__((ext_vector_type(2)));
  if (vec_ok && i0 + GT <= n1 && j0 + GTN <= n2) {
    // interior tile: no per-element control flow (g3_gram.hip::gram_kernel, round 5) -- two rows x two columns of
    // independent evaluations in flight, the diagonal test only in tiles the diagonal crosses, tt_to_num as one test
    // per four values.  Same formulas as the general loop below (equal to rounding).
    const i64 dlo = i0 + diag_off;
    const bool touches = sym && dlo < j0 + GTN && j0 < dlo + GT;        // (uniform)
    auto rows = [&](const bool on_diag) __attribute__((always_inline)) {
      for (int r0 = ty; r0 < GT; r0 += 8) {
        T v[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int rr = r0 + 4 * u;
            const bool dg = on_diag && (dlo + rr == ja + q);
            v[u][q] = prog_eval(prog, xi_s + rr * dp, q ? xrb : xra, dg, sym != 0, trig_s + rr * tstride, trig_s + (GT + 2 * tx + q) * tstride);
          }
        if (scr) {
          bool bad = false;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 2; ++q) bad |= !__builtin_isfinite(v[u][q]);
          if (bad) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int q = 0; q < 2; ++q) v[u][q] = scrub(v[u][q]);
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<vec2*>(K + (i0 + r0 + 4 * u) * ldk + ja) = vec2{v[u][0], v[u][1]};
      }
    };
    if (touches) rows(true); else rows(false);
    return;
  }
  for (int rr = ty; rr < GT; rr += 4) {
    const i64 i = i0 + rr;
    if (i >= n1pad) break;
    T v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const i64 j = ja + q;
      if (i < n1 && j < n2) {
        const bool dg = sym && (i + diag_off == j);
        v[q] = prog_eval(prog, xi_s + rr * dp, q ? xrb : xra, dg, sym != 0, trig_s + rr * tstride, trig_s + (GT + 2 * tx + q) * tstride);
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
)JIT";

// ---- the kernel-parameter sums of the gradient (g3_grad.hip::gram_grad_kernel) for ONE expression structure: leaf kinds,
// dimensions and the product table are compile-time, so every slot index is a constant and the accumulators are registers
// (the interpreter keeps them in LDS: one read-modify-write per slot and pair), and the leaf formulas are straight-line code.
const char* kJitGradSource = R"JIT(
typedef JT T;
typedef long long i64;
#define GG_T 64
#define GG_THREADS 256
#define GG_PI 3.14159265358979323846
enum { K_SE = 0, K_OU = 1, K_MAT32 = 2, K_MAT52 = 3, K_RQ = 4, K_COS = 5, K_SIN = 6, K_SINC = 7, K_SM = 8, K_NOISE = 9, K_WN = 10 };
struct jleaf { int kind; int ndims; int dims[JMAXD]; double var; double alpha; double rate[JMAXD]; double freq[JMAXD]; };
struct jprod { double coef; int nfac; int fac[JMAXFAC]; int pad[3]; };
struct jprog { int nleaf; int nprod; double shift; jleaf leaf[JMAXLEAF]; jprod prod[JMAXPROD]; };
static_assert(sizeof(jprog) == JPROG_BYTES, "g3_kernel_prog layout");

JTABLES

// the standard slot layout of g3_grad_layout: per leaf var, [alpha], [freq...], [rate...]
constexpr int slots_of(int kind, int nd) {
  return (kind == K_NOISE || kind == K_WN) ? 1 : kind == K_RQ ? 2 + nd : (kind == K_SIN || kind == K_SM) ? 1 + 2 * nd : 1 + nd;
}
constexpr int slot_base(int L) {
  int s = 0;
  for (int l = 0; l < L; ++l) s += slots_of(jkind[l], jnd[l]);
  return s;
}
constexpr int JNS = slot_base(JNL);
constexpr int JNSA = JNS > 0 ? JNS : 1;
constexpr int JNLA = JNL > 0 ? JNL : 1;
static_assert(JNS == JNSLOTS, "slot layout");
constexpr bool any_multi() {
  for (int p = 0; p < JNP; ++p) if (jnfac[p] > 1) return true;
  return false;
}
constexpr bool JMULTI = any_multi();

// value of leaf L (variance included) for the pair -- g3_grad.hip::leaf_value with the kind folded
template <int L>
__device__ __forceinline__ double leaf_val(const jleaf& lf, const double* xi, const double* xj, bool diag) {
  constexpr int kind = jkind[L], nd = jnd[L];
  if constexpr (kind == K_NOISE || kind == K_WN) {
    return diag ? lf.var : 0.0;
  } else if constexpr (kind == K_SE || kind == K_MAT32 || kind == K_MAT52 || kind == K_RQ) {
    double D = 0.0;
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const double dx = xi[jdims[L][k]] - xj[jdims[L][k]];
      D += (dx * dx) * (0.5 * lf.rate[k] * lf.rate[k]);
    }
    if constexpr (kind == K_SE) return lf.var * exp(-D);
    if constexpr (kind == K_MAT32) { const double s = sqrt(3.0 * D); return lf.var * ((1.0 + s) * exp(-s)); }
    if constexpr (kind == K_MAT52) { const double s = sqrt(5.0 * D); return lf.var * ((1.0 + s + 5.0 * D / 3.0) * exp(-s)); }
    return lf.var * pow(1.0 + D / lf.alpha, -lf.alpha);
  } else if constexpr (kind == K_OU) {
    double D = 0.0;
#pragma unroll
    for (int k = 0; k < nd; ++k) D += fabs(xi[jdims[L][k]] - xj[jdims[L][k]]) * lf.rate[k];
    return lf.var * exp(-D);
  } else if constexpr (kind == K_SIN) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const double v = sin(GG_PI * (xi[jdims[L][k]] - xj[jdims[L][k]]) * lf.freq[k]);
      s += (v * v) * lf.rate[k];
    }
    return lf.var * exp(2.0 * s);
  } else {     // COS, SINC, SM
    constexpr bool sinc = kind == K_SINC;
    constexpr double cs = sinc ? 2.0 * GG_PI * GG_PI : 2.0 * GG_PI;
    double p = 1.0, s = 0.0;
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const double dx = xi[jdims[L][k]] - xj[jdims[L][k]];
      const double th = cs * dx * lf.freq[k];
      p *= sinc ? (dx != 0.0 ? sin(th) / th : 1.0) : cos(th);
      s += (dx * dx) * (lf.rate[k] * lf.rate[k]);
    }
    const double env = kind == K_SM ? exp(-2.0 * GG_PI * GG_PI * s) : 1.0;
    return lf.var * (env * p);
  }
}

// w * d(var * k)/d(param) for every parameter of leaf L into its slots -- g3_grad.hip::leaf_grad with the kind folded
template <int L>
__device__ __forceinline__ void leaf_grad(const jleaf& lf, const double* xi, const double* xj, bool diag, double w, double (&acc)[JNSA]) {
  constexpr int kind = jkind[L], nd = jnd[L], s0 = slot_base(L);
  const double var = lf.var;
  const double wv = w * var;
  if constexpr (kind == K_NOISE || kind == K_WN) {
    if (diag) acc[s0] += w;
  } else if constexpr (kind == K_SE || kind == K_MAT32 || kind == K_MAT52 || kind == K_RQ) {
    constexpr int srate = s0 + (kind == K_RQ ? 2 : 1);
    double D = 0.0;
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const double dx = xi[jdims[L][k]] - xj[jdims[L][k]];
      D += (dx * dx) * (0.5 * lf.rate[k] * lf.rate[k]);
    }
    double kv, dkdD;
    if constexpr (kind == K_SE) {
      kv = exp(-D); dkdD = -kv;
    } else if constexpr (kind == K_MAT32) {
      const double s = sqrt(3.0 * D), e = exp(-s);
      kv = (1.0 + s) * e; dkdD = -1.5 * e;
    } else if constexpr (kind == K_MAT52) {
      const double s = sqrt(5.0 * D), e = exp(-s);
      kv = (1.0 + s + 5.0 * D / 3.0) * e; dkdD = -(5.0 / 6.0) * (1.0 + s) * e;
    } else {
      const double al = lf.alpha, b = 1.0 + D / al;
      kv = pow(b, -al); dkdD = -kv / b;
      acc[s0 + 1] += wv * kv * (-log(b) + D / (al + D));
    }
    acc[s0] += w * kv;
    const double c = wv * dkdD;
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const double dx = xi[jdims[L][k]] - xj[jdims[L][k]];
      acc[srate + k] += c * lf.rate[k] * (dx * dx);
    }
  } else if constexpr (kind == K_OU) {
    double D = 0.0;
#pragma unroll
    for (int k = 0; k < nd; ++k) D += fabs(xi[jdims[L][k]] - xj[jdims[L][k]]) * lf.rate[k];
    const double kv = exp(-D);
    acc[s0] += w * kv;
#pragma unroll
    for (int k = 0; k < nd; ++k) acc[s0 + 1 + k] += -wv * kv * fabs(xi[jdims[L][k]] - xj[jdims[L][k]]);
  } else if constexpr (kind == K_SIN) {
    constexpr int sfreq = s0 + 1, srate = s0 + 1 + nd;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const double v = sin(GG_PI * (xi[jdims[L][k]] - xj[jdims[L][k]]) * lf.freq[k]);
      s += (v * v) * lf.rate[k];
    }
    const double kv = exp(2.0 * s);
    acc[s0] += w * kv;
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const double dx = xi[jdims[L][k]] - xj[jdims[L][k]];
      const double v = sin(GG_PI * dx * lf.freq[k]);
      acc[srate + k] += wv * kv * 2.0 * (v * v);
      acc[sfreq + k] += wv * kv * (2.0 * GG_PI) * lf.rate[k] * dx * sin(2.0 * GG_PI * dx * lf.freq[k]);
    }
  } else {     // COS, SINC, SM: value = env * prod_k f_k ; d/dfreq_m = env * f'_m * prod_{k != m} f_k
    constexpr bool sinc = kind == K_SINC;
    constexpr double cs = sinc ? 2.0 * GG_PI * GG_PI : 2.0 * GG_PI;
    constexpr int sfreq = s0 + 1, srate = s0 + 1 + nd;
    double p = 1.0, s = 0.0;
#pragma unroll
    for (int k = 0; k < nd; ++k) {
      const double dx = xi[jdims[L][k]] - xj[jdims[L][k]];
      const double th = cs * dx * lf.freq[k];
      p *= sinc ? (dx != 0.0 ? sin(th) / th : 1.0) : cos(th);
      s += (dx * dx) * (lf.rate[k] * lf.rate[k]);
    }
    const double env = kind == K_SM ? exp(-2.0 * GG_PI * GG_PI * s) : 1.0;
    const double kv = env * p;
    acc[s0] += w * kv;
#pragma unroll
    for (int m = 0; m < nd; ++m) {
      const double dx = xi[jdims[L][m]] - xj[jdims[L][m]];
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
#pragma unroll
        for (int k = 0; k < nd; ++k) {
          if (k == m) continue;
          const double dk = xi[jdims[L][k]] - xj[jdims[L][k]];
          const double tk = cs * dk * lf.freq[k];
          others *= sinc ? (dk != 0.0 ? sin(tk) / tk : 1.0) : cos(tk);
        }
      }
      acc[sfreq + m] += wv * env * dfm * others;
      if constexpr (kind == K_SM) acc[srate + m] += wv * kv * (-4.0 * GG_PI * GG_PI) * (dx * dx) * lf.rate[m];
    }
  }
}

template <int L>
__device__ __forceinline__ void all_vals(const jprog* __restrict__ prog, const double* xi, const double* xj, bool diag, double (&lv)[JNLA]) {
  if constexpr (L < JNL) {
    lv[L] = leaf_val<L>(prog->leaf[L], xi, xj, diag);
    all_vals<L + 1>(prog, xi, xj, diag, lv);
  }
}

// dK/d(leaf L) = sum over the products that contain it of coef * the other factors (first occurrence removed)
template <int L>
__device__ __forceinline__ double dk_dleaf(const jprog* __restrict__ prog, const double (&lv)[JNLA]) {
  double q = 0.0;
#pragma unroll
  for (int p = 0; p < JNP; ++p) {
    bool has = false;
    double v = prog->prod[p].coef;
#pragma unroll
    for (int f = 0; f < JMAXFAC; ++f)
      if (f < jnfac[p]) {
        if (jfac[p][f] == L && !has) has = true;
        else v *= lv[jfac[p][f]];
      }
    if (has) q += v;
  }
  return q;
}

template <int L>
__device__ __forceinline__ void all_grads(const jprog* __restrict__ prog, const double* xi, const double* xj, bool diag, double g,
                                          const double (&lv)[JNLA], double (&acc)[JNSA]) {
  if constexpr (L < JNL) {
    const double q = dk_dleaf<L>(prog, lv);
    if (q != 0.0) leaf_grad<L>(prog->leaf[L], xi, xj, diag, g * q, acc);
    all_grads<L + 1>(prog, xi, xj, diag, g, lv, acc);
  }
}

extern "C" __global__ void __launch_bounds__(GG_THREADS)
g3_grad_jit(const jprog* __restrict__ prog, const T* __restrict__ X, i64 N, i64 ldx, const T* __restrict__ G, i64 ldg,
            const T* __restrict__ alpha, double* __restrict__ partial, i64 row0, i64 row1, i64 gstride, i64 astride) {
  // rows [row0, row1) of the lower triangle; grid.y = batch member (its program, K^-1, alpha, partial sums)
  prog += blockIdx.y;
  G += (i64)blockIdx.y * gstride;
  alpha += (i64)blockIdx.y * astride;
  partial += (size_t)blockIdx.y * gridDim.x * JNSA;
  constexpr int dp = JD | 1;
  __shared__ double xi_s[GG_T * dp], xj_s[GG_T * dp], ai_s[GG_T], aj_s[GG_T];
  __shared__ double red[JNSA * (GG_THREADS / 64)];
  const int tid = threadIdx.x;
  double acc[JNSA];
#pragma unroll
  for (int s = 0; s < JNSA; ++s) acc[s] = 0.0;
  const i64 bi0 = row0 / GG_T, bi1 = (row1 + GG_T - 1) / GG_T;
  const i64 id0 = bi0 * (bi0 + 1) / 2, ntiles = bi1 * (bi1 + 1) / 2 - id0;
  for (i64 idl = blockIdx.x; idl < ntiles; idl += gridDim.x) {
    const i64 id = id0 + idl;
    i64 bi = (i64)((sqrt(1.0 + 8.0 * (double)id) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= id) ++bi;
    while (bi * (bi + 1) / 2 > id) --bi;
    const i64 bj = id - bi * (bi + 1) / 2;
    const i64 i0 = bi * GG_T, j0 = bj * GG_T;
    __syncthreads();
    for (int e = tid; e < GG_T * JD; e += GG_THREADS) {
      const int r = e / JD, c = e - r * JD;
      xi_s[r * dp + c] = i0 + r < row1 ? (double)X[(i0 + r) * ldx + c] : 0.0;
      xj_s[r * dp + c] = j0 + r < N ? (double)X[(j0 + r) * ldx + c] : 0.0;
    }
    if (tid < GG_T) ai_s[tid] = i0 + tid < row1 ? (double)alpha[i0 + tid] : 0.0;
    else if (tid < 2 * GG_T) aj_s[tid - GG_T] = j0 + tid - GG_T < N ? (double)alpha[j0 + tid - GG_T] : 0.0;
    __syncthreads();
    const int c = tid & (GG_T - 1);
    const i64 j = j0 + c;
    double xj[JD];
#pragma unroll
    for (int q = 0; q < JD; ++q) xj[q] = xj_s[c * dp + q];
    for (int rr = tid >> 6; rr < GG_T; rr += GG_THREADS / GG_T) {
      const i64 i = i0 + rr;
      if (i >= row1 || j > i) continue;
      const bool diag = i == j;
      const double* xi = xi_s + rr * dp;
      const double g = (diag ? 0.5 : 1.0) * (ai_s[rr] * aj_s[c] - (double)G[(i - row0) * ldg + j]);
      double lv[JNLA];
      if constexpr (JMULTI) all_vals<0>(prog, xi, xj, diag, lv);
      else {
#pragma unroll
        for (int l = 0; l < JNLA; ++l) lv[l] = 1.0;
      }
      all_grads<0>(prog, xi, xj, diag, g, lv, acc);
    }
  }
  // block reduction
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int s = 0; s < JNS; ++s) {
    double v = acc[s];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[s * (GG_THREADS / 64) + wv] = v;
  }
  __syncthreads();
  if (tid < JNS) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < GG_THREADS / 64; ++q) v += red[tid * (GG_THREADS / 64) + q];
    partial[(size_t)blockIdx.x * JNSA + tid] = v;
  }
}
)JIT";

// ---- hipRTC, loaded on demand
struct Rtc {
  void* h = nullptr;
  int (*CreateProgram)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*CompileProgram)(void*, int, const char**) = nullptr;
  int (*GetProgramLogSize)(void*, size_t*) = nullptr;
  int (*GetProgramLog)(void*, char*) = nullptr;
  int (*GetCodeSize)(void*, size_t*) = nullptr;
  int (*GetCode)(void*, char*) = nullptr;
  int (*DestroyProgram)(void**) = nullptr;
};

std::mutex g_mu;
Rtc g_rtc;
int g_rtc_state = 0;                     // 0 untried, 1 ok, -1 unavailable
struct Entry { hipFunction_t fn = nullptr; int ntrig = 0; bool failed = false; };
std::map<std::string, Entry> g_cache;    // key: device | dtype | d | structure

Rtc* rtc() {      // (g_mu held)
  if (g_rtc_state == 1) return &g_rtc;
  if (g_rtc_state == -1) return nullptr;
  const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
  for (const char* n : names) {
    g_rtc.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rtc.h) break;
  }
  g_rtc_state = -1;
  if (!g_rtc.h) return nullptr;
#define G3_RSYM(f)                                                  \
  g_rtc.f = (decltype(g_rtc.f))dlsym(g_rtc.h, "hiprtc" #f);         \
  if (!g_rtc.f) return nullptr;
  G3_RSYM(CreateProgram) G3_RSYM(CompileProgram) G3_RSYM(GetProgramLogSize) G3_RSYM(GetProgramLog) G3_RSYM(GetCodeSize)
  G3_RSYM(GetCode) G3_RSYM(DestroyProgram)
#undef G3_RSYM
  g_rtc_state = 1;
  return &g_rtc;
}

// the structure of a program: everything the generated kernel is specialised on, as the cache key AND as the header
// of constexpr tables.  Hyper-parameters (var, rate, freq, alpha, coef, shift) are data and not part of it.
std::string structure_tables(const g3_kernel_prog* p, int d, int* ntrig_out) {
  std::string s;
  char b[256];
  auto arr = [&](const char* name, int n, auto get) {
    s += std::string("constexpr int ") + name + "[" + std::to_string(n > 0 ? n : 1) + "] = {";
    for (int i = 0; i < (n > 0 ? n : 1); ++i) { snprintf(b, sizeof(b), "%s%d", i ? ", " : "", i < n ? get(i) : 0); s += b; }
    s += "};\n";
  };
  const int nl = p->nleaf, np = p->nprod;
  snprintf(b, sizeof(b), "#define JD %d\n#define JNL %d\n#define JNP %d\n", d, nl, np);
  s += b;
  arr("jkind", nl, [&](int i) { return (int)p->leaf[i].kind; });
  arr("jnd", nl, [&](int i) { return (int)p->leaf[i].ndims; });
  s += "constexpr int jdims[" + std::to_string(nl > 0 ? nl : 1) + "][JMAXD] = {";
  for (int l = 0; l < (nl > 0 ? nl : 1); ++l) {
    s += l ? ", {" : "{";
    for (int k = 0; k < G3_MAXD; ++k) { snprintf(b, sizeof(b), "%s%d", k ? "," : "", (l < nl && k < p->leaf[l].ndims) ? (int)p->leaf[l].dims[k] : 0); s += b; }
    s += "}";
  }
  s += "};\n";
  // trig table: one [cos, sin] pair per (periodic leaf, dimension)
  std::string tl, tk, tc;
  int nt = 0;
  std::string toff;
  for (int l = 0; l < nl; ++l) {
    const int kd = p->leaf[l].kind;
    const bool per = kd == G3_K_COS || kd == G3_K_SIN || kd == G3_K_SM;
    snprintf(b, sizeof(b), "%s%d", l ? ", " : "", per ? nt : -1);
    toff += b;
    if (per)
      for (int k = 0; k < p->leaf[l].ndims; ++k, ++nt) {
        snprintf(b, sizeof(b), "%s%d", nt ? ", " : "", l); tl += b;
        snprintf(b, sizeof(b), "%s%d", nt ? ", " : "", k); tk += b;
        snprintf(b, sizeof(b), "%s%d", nt ? ", " : "", (int)p->leaf[l].dims[k]); tc += b;
      }
  }
  if (nl == 0) toff = "-1";
  s += "constexpr int jtoff[" + std::to_string(nl > 0 ? nl : 1) + "] = {" + toff + "};\n";
  snprintf(b, sizeof(b), "#define JNTRIG %d\n", nt);
  s += b;
  if (nt == 0) { tl = tk = tc = "0"; }
  s += "constexpr int jtleaf[" + std::to_string(nt > 0 ? nt : 1) + "] = {" + tl + "};\n";
  s += "constexpr int jtk[" + std::to_string(nt > 0 ? nt : 1) + "] = {" + tk + "};\n";
  s += "constexpr int jtcol[" + std::to_string(nt > 0 ? nt : 1) + "] = {" + tc + "};\n";
  arr("jnfac", np, [&](int i) { return (int)p->prod[i].nfac; });
  s += "constexpr int jfac[" + std::to_string(np > 0 ? np : 1) + "][JMAXFAC] = {";
  for (int q = 0; q < (np > 0 ? np : 1); ++q) {
    s += q ? ", {" : "{";
    for (int f = 0; f < G3_MAXFAC; ++f) { snprintf(b, sizeof(b), "%s%d", f ? "," : "", (q < np && f < p->prod[q].nfac) ? (int)p->prod[q].fac[f] : 0); s += b; }
    s += "}";
  }
  s += "};\n";
  *ntrig_out = nt;
  return s;
}

// compile the source specialised by `tables` for gfx950; 0 = ok (code object in *code), else the compiler's log   (g_mu held)
int compile_structure(const std::string& tables, g3_dtype dt, std::string* code, std::string* log, int grad_slots = -1) {
  Rtc* r = rtc();
  if (!r) { *log = "libhiprtc not available"; return -1; }
  std::string src = grad_slots >= 0 ? kJitGradSource : kJitSource;     // grad_slots >= 0: the gradient kernel, with its slot count
  src.replace(src.find("JTABLES"), 7, tables);
  void* pr = nullptr;
  if (r->CreateProgram(&pr, src.c_str(), "g3_gram_jit.hip", 0, nullptr, nullptr) != 0) { *log = "hiprtcCreateProgram failed"; return -2; }
  char o1[64], o2[64], o3[64], o4[64], o5[64], o6[64], o7[64];
  snprintf(o7, sizeof(o7), "-DJNSLOTS=%d", grad_slots >= 0 ? grad_slots : 0);
  snprintf(o1, sizeof(o1), "-DJT=%s", dt == G3_F64 ? "double" : "float");
  snprintf(o2, sizeof(o2), "-DJMAXD=%d", G3_MAXD);
  snprintf(o3, sizeof(o3), "-DJMAXLEAF=%d", G3_MAXLEAF);
  snprintf(o4, sizeof(o4), "-DJMAXPROD=%d", G3_MAXPROD);
  snprintf(o5, sizeof(o5), "-DJMAXFAC=%d", G3_MAXFAC);
  snprintf(o6, sizeof(o6), "-DJPROG_BYTES=%d", (int)sizeof(g3_kernel_prog));
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", o1, o2, o3, o4, o5, o6, o7};
  const int cr = r->CompileProgram(pr, 10, opts);
  size_t ls = 0;
  r->GetProgramLogSize(pr, &ls);
  if (ls > 1) { log->assign(ls, '\0'); r->GetProgramLog(pr, &(*log)[0]); }
  size_t cs = 0;
  int rc = cr;
  if (cr == 0 && r->GetCodeSize(pr, &cs) == 0 && cs > 0) {
    code->assign(cs, '\0');
    if (r->GetCode(pr, &(*code)[0]) != 0) rc = -3;
  } else if (cr == 0) {
    rc = -4;
  }
  r->DestroyProgram(&pr);
  return rc;
}

// the number of slots of g3_grad_layout's standard map for this program (what the generated gradient kernel accumulates)
int std_grad_slots(const g3_kernel_prog* p) {
  int s = 0;
  for (int l = 0; l < p->nleaf; ++l) {
    const int kd = p->leaf[l].kind, nd = p->leaf[l].ndims;
    s += (kd == G3_K_NOISE || kd == G3_K_WN) ? 1 : kd == G3_K_RQ ? 2 + nd : (kd == G3_K_SIN || kd == G3_K_SM) ? 1 + 2 * nd : 1 + nd;
  }
  return s;
}

}  // namespace

extern "C" int g3_grad_jit_check(const g3_kernel_prog* prog, int d, g3_dtype dt, int64_t* code_bytes, char* log, int64_t log_bytes) {
  if (!prog) return -1;
  if (g3i_validate_prog(prog, d)) return -2;
  int ntrig = 0;
  const std::string tables = structure_tables(prog, d, &ntrig);
  std::string code, lg;
  std::lock_guard<std::mutex> lk(g_mu);
  const int rc = compile_structure(tables, dt, &code, &lg, std_grad_slots(prog));
  if (code_bytes) *code_bytes = (int64_t)code.size();
  if (log && log_bytes > 0) snprintf(log, (size_t)log_bytes, "%s", lg.c_str());
  return rc;
}

// The generated gradient kernel (g3_grad_jit) for (prog's structure, d, dtype) on the context's device, compiled at first use
// and cached; nullptr when there is none (no hipRTC, more than G3_GRAD_JIT_MAXSLOTS register accumulators, a failed
// compilation: the caller interprets).  *nslots: the standard slot count the kernel accumulates.
hipFunction_t g3i_grad_jit_function(g3_ctx* ctx, const g3_kernel_prog* prog_host, int d, g3_dtype dt, int* nslots) {
  if (ctx->tune.grad_interpret || !ctx->tune.gram_jit) return nullptr;
  const int ns = std_grad_slots(prog_host);
  *nslots = ns;
  if (ns < 1 || ns > G3_GRAD_JIT_MAXSLOTS) return nullptr;
  int ntrig = 0;
  const std::string tables = structure_tables(prog_host, d, &ntrig);
  const std::string key = std::to_string(ctx->device) + (dt == G3_F64 ? "|f64|grad|" : "|f32|grad|") + tables;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_cache.find(key);
  if (it != g_cache.end()) return it->second.failed ? nullptr : it->second.fn;
  Entry ent;
  ent.failed = true;
  std::string code, log;
  if (compile_structure(tables, dt, &code, &log, ns) == 0) {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    if (hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(&fn, mod, "g3_grad_jit") == hipSuccess) {
      ent.fn = fn;
      ent.failed = false;
    }
  } else if (getenv("G3_JIT_VERBOSE")) {
    fprintf(stderr, "libg3hip: generated gradient kernel did not compile:\n%s\n", log.c_str());
  }
  g_cache[key] = ent;
  return ent.failed ? nullptr : ent.fn;
}

// No GPU needed: does the kernel generated for this expression compile for gfx950?  Returns 0 and the code object's size,
// or the compiler's status with its log (tests/test_host.py runs the whole kernel zoo through it on the build host).
extern "C" int g3_gram_jit_check(const g3_kernel_prog* prog, int d, g3_dtype dt, int64_t* code_bytes, char* log, int64_t log_bytes) {
  if (!prog) return -1;
  if (g3i_validate_prog(prog, d)) return -2;
  int ntrig = 0;
  const std::string tables = structure_tables(prog, d, &ntrig);
  std::string code, lg;
  std::lock_guard<std::mutex> lk(g_mu);
  const int rc = compile_structure(tables, dt, &code, &lg);
  if (code_bytes) *code_bytes = (int64_t)code.size();
  if (log && log_bytes > 0) snprintf(log, (size_t)log_bytes, "%s", lg.c_str());
  return rc;
}

// The generated kernel for (prog's structure, d, dtype) on the context's device: compiled at first use, then cached.
// Returns 1 when there is none (no hipRTC, too many trig pairs for the LDS, a failed compilation: the caller interprets).
int g3i_gram_jit(g3_ctx* ctx, const g3_kernel_prog* prog_host, const g3_kernel_prog* prog_dev, int batch, const void* X1, int64_t n1,
                 int64_t ldx1, const void* X2, int64_t n2, int64_t ldx2, int d, g3_dtype dt, void* K, int64_t ldk, int64_t n1pad,
                 int64_t n2pad, unsigned flags, int sym, int64_t kstride, int64_t diag_off, dim3 grid) {
  if (ctx->tune.gram_interpret || !ctx->tune.gram_jit) return 1;
  int ntrig = 0;
  const std::string tables = structure_tables(prog_host, d, &ntrig);
  if (ntrig > 24) return 1;
  const std::string key = std::to_string(ctx->device) + (dt == G3_F64 ? "|f64|" : "|f32|") + tables;
  Entry ent;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) {
      ent = it->second;
    } else {
      ent.ntrig = ntrig;
      ent.failed = true;
      std::string code, log;
      if (compile_structure(tables, dt, &code, &log) == 0) {
        hipModule_t mod = nullptr;
        hipFunction_t fn = nullptr;
        if (hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(&fn, mod, "g3_gram_jit") == hipSuccess) {
          (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
          ent.fn = fn;
          ent.failed = false;
        }
      } else if (getenv("G3_JIT_VERBOSE")) {
        fprintf(stderr, "libg3hip: generated Gram kernel did not compile:\n%s\n", log.c_str());
      }
      g_cache[key] = ent;
    }
  }
  if (ent.failed || !ent.fn) return 1;
  const size_t es = g3_esize(dt);
  const size_t lds = (size_t)(64 + 128) * ((d | 1) + (ntrig ? 2 * ntrig + 1 : 0)) * es;
  if (lds > 96 * 1024) return 1;
  grid.z = (unsigned)(batch > 1 ? batch : 1);
  struct {
    const void* prog; const void* X1; long long n1, ldx1; const void* X2; long long n2, ldx2; void* K; long long ldk, n1pad, n2pad;
    unsigned flags; int sym; long long kstride, diag_off;
  } args = {prog_dev, X1, n1, ldx1, X2, n2, ldx2, K, ldk, n1pad, n2pad, flags, sym, kstride, diag_off};
  size_t asz = sizeof(args);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
  if (hipModuleLaunchKernel(ent.fn, grid.x, grid.y, grid.z, 256, 1, 1, (unsigned)lds, ctx->stream, nullptr, cfg) != hipSuccess) {
    (void)hipGetLastError();
    return 1;
  }
  ctx->gram_paths[1] += 1;
  return G3_OK;
}
