// Evaluation of a kernel expression (g3_kernel_prog: shift + sum of products of leaves) for ONE pair of points, as device
// functions: the interpreter of the Gram kernel (g3_gram.hip) and of the one-workgroup-per-member chain kernel
// (g3_potrf.hip::small_factor_kernel, which builds its member's covariance itself).  Formulas and their reference lines:
// g3py/processes/hypers/metrics.py:30-35,89-102; kernels.py:360-487.
#pragma once
#include <hip/hip_runtime.h>

#include "g3hip.h"

#define G3_PI 3.14159265358979323846

// Periodic leaves depend on the pair only through cos / sin of theta_i - theta_j with
// theta = scale * freq_k * x_k.  When `ti`, `tj` are given they hold, per point,
// [cos(theta_k), sin(theta_k)] pairs prepared once per tile, and the pair value costs two
// FMAs per dimension instead of one fp64 cos/sin (the angle-difference identities are exact;
// the only deviation from the direct formula is the rounding of theta itself).
// (SINC divides sin(theta_i - theta_j) by dx: the absolute rounding of theta would be amplified
//  for close points, so SINC keeps the direct formula.)
__host__ __device__ __forceinline__ bool leaf_has_trig(int kind) {
  return kind == G3_K_COS || kind == G3_K_SIN || kind == G3_K_SM;
}

// exp(x) for the compile-time fast paths.  Round 3 built an own fp64 exp -- n = rint(x / ln 2), r = x - n ln 2 in two
// pieces, degree-13 Taylor polynomial of exp(r) (truncation 4e-18), one v_ldexp_f64; <= 4 ulp, NaN / +-Inf and the
// subnormal range handled -- and MEASURED it against the library routine (scripts/gram_bench.py, profiles/r03_gram.md):
// SE d=4 N=32768 1.645 vs 1.599 ms, MAT52+COS d=8 N=16384 0.790 vs 0.766 ms: the device library's exp is already a short
// branch-free sequence and wins by 2-3 %.  The library routine is used; the hand-written one was removed in round 4 (its A/B numbers stay in profiles/r03_gram.md).
__device__ __forceinline__ double g3_exp(double x) { return exp(x); }
__device__ __forceinline__ float g3_exp(float x) { return exp(x); }

template <typename T>
__device__ __forceinline__ T leaf_eval(const g3_leaf& lf, const T* xi, const T* xj, bool diag_sym,
                                       bool sym, const T* ti, const T* tj) {
  // xi, xj: LDS rows (all d columns of the two points); ti, tj: trig rows of this leaf or null
  const int nd = lf.ndims;
  const T var = (T)lf.var;
  switch (lf.kind) {
    case G3_K_NOISE:
      return diag_sym ? var : T(0);
    case G3_K_WN: {
      if (sym) return diag_sym ? var : T(0);
      T cnt = T(0);
      for (int k = 0; k < nd; ++k) {
        const int c = lf.dims[k];
        cnt += (xi[c] - xj[c] == T(0)) ? T(1) : T(0);
      }
      return var * cnt;
    }
    case G3_K_SE:
    case G3_K_MAT32:
    case G3_K_MAT52:
    case G3_K_RQ: {
      T d = T(0);
      for (int k = 0; k < nd; ++k) {
        const int c = lf.dims[k];
        const T dx = xi[c] - xj[c];
        const T r = (T)lf.rate[k];
        d += (dx * dx) * (T(0.5) * r * r);  // ARD_L2, metrics.py:100-102
      }
      if (lf.kind == G3_K_SE) return var * exp(-d);
      if (lf.kind == G3_K_MAT32) {
        const T s = sqrt(T(3) * d);
        return var * ((T(1) + s) * exp(-s));
      }
      if (lf.kind == G3_K_MAT52) {
        const T s = sqrt(T(5) * d);
        return var * ((T(1) + s + T(5) * d / T(3)) * exp(-s));
      }
      const T al = (T)lf.alpha;
      return var * pow(T(1) + d / al, -al);
    }
    case G3_K_OU: {
      T d = T(0);
      for (int k = 0; k < nd; ++k) {
        const int c = lf.dims[k];
        d += fabs(xi[c] - xj[c]) * (T)lf.rate[k];  // ARD_L1, metrics.py:89-91
      }
      return var * exp(-d);
    }
    case G3_K_COS: {
      T p = T(1);
      if (ti) {
        for (int k = 0; k < nd; ++k) p *= ti[2 * k] * tj[2 * k] + ti[2 * k + 1] * tj[2 * k + 1];
        return var * p;
      }
      for (int k = 0; k < nd; ++k) {
        const int c = lf.dims[k];
        p *= cos(T(2 * G3_PI) * (xi[c] - xj[c]) * (T)lf.freq[k]);
      }
      return var * p;
    }
    case G3_K_SIN: {
      T s = T(0);
      if (ti) {   // sin^2(pi f dx) = (1 - cos(2 pi f dx)) / 2
        for (int k = 0; k < nd; ++k)
          s += (T(0.5) * (T(1) - (ti[2 * k] * tj[2 * k] + ti[2 * k + 1] * tj[2 * k + 1]))) * (T)lf.rate[k];
        return var * exp(T(2) * s);
      }
      for (int k = 0; k < nd; ++k) {
        const int c = lf.dims[k];
        const T v = sin(T(G3_PI) * (xi[c] - xj[c]) * (T)lf.freq[k]);
        s += (v * v) * (T)lf.rate[k];
      }
      return var * exp(T(2) * s);  // positive exponent, as written at kernels.py:472
    }
    case G3_K_SINC: {
      T p = T(1);
      for (int k = 0; k < nd; ++k) {
        const int c = lf.dims[k];
        const T dx = xi[c] - xj[c];
        const T f = (T)lf.freq[k];
        const T sn = ti ? (ti[2 * k + 1] * tj[2 * k] - ti[2 * k] * tj[2 * k + 1])   // sin(theta_i - theta_j)
                        : sin(T(2 * G3_PI * G3_PI) * dx * f);
        const T v = sn / (T(2 * G3_PI * G3_PI) * f * dx);
        p *= (dx != T(0)) ? v : T(1);
      }
      return var * p;
    }
    case G3_K_SM: {
      T s = T(0), p = T(1);
      for (int k = 0; k < nd; ++k) {
        const int c = lf.dims[k];
        const T dx = xi[c] - xj[c];
        const T r = (T)lf.rate[k];
        s += (dx * dx) * (r * r);
        p *= ti ? (ti[2 * k] * tj[2 * k] + ti[2 * k + 1] * tj[2 * k + 1]) : cos(T(2 * G3_PI) * dx * (T)lf.freq[k]);
      }
      return var * (exp(T(-2 * G3_PI * G3_PI) * s) * p);
    }
    default:
      return T(0);
  }
}

template <typename T>
__device__ __forceinline__ T prog_eval(const g3_kernel_prog* __restrict__ prog, const T* xi,
                                       const T* xj, bool diag_sym, bool sym, const T* trig_i = nullptr,
                                       const T* trig_j = nullptr, const int* toff = nullptr) {
  T acc = (T)prog->shift;
  const int np = prog->nprod;
  for (int p = 0; p < np; ++p) {
    T v = (T)prog->prod[p].coef;
    const int nf = prog->prod[p].nfac;
    for (int f = 0; f < nf; ++f) {
      const int l = prog->prod[p].fac[f];
      const bool tr = trig_i != nullptr && toff[l] >= 0;
      v *= leaf_eval<T>(prog->leaf[l], xi, xj, diag_sym, sym, tr ? trig_i + 2 * toff[l] : (const T*)nullptr,
                        tr ? trig_j + 2 * toff[l] : (const T*)nullptr);
    }
    acc += v;
  }
  return acc;
}

template <typename T>
__device__ __forceinline__ T scrub(T v) {
  // tt_to_num: NaN -> 0, +-Inf -> 1e10 (tensors.py:90-92)
  if (v != v) return T(0);
  if (__builtin_isinf(v)) return (T)1e10f;
  return v;
}

