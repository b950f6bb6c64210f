// Diagonal-block device code of libg3hip: the 16 x 16 tile routines, the 128 x 128 factor-and-invert block program of one
// wave (diag128_core / diag128_wave) and the fused 256-wide factorisation (potrf256_wave) -- as device functions, shared by the
// diagonal kernels of the blocked Cholesky (g3_potrf.hip) and by the cooperative chain kernel (g3_chainb.hip), in which a
// small group of workgroups factors one member of a batch.  Replaces the dpotrf leaf of g3py/libs/tensors.py:197-222.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "g3_internal.h"
#include "g3_mfma.h"

// ---------------------------------------------------------------------------------------
// v_rsq_f64 is good to 5e-8 (24 bits); ONE third-order step  y (1 + e/2 + 3 e^2/8), e = 1 - p y^2, takes that to
// full precision (max relative error 1.4e-16 over 2^20 arguments, scripts/rsq_probe.hip -- two Newton steps: 2.4e-16)
// in four dependent operations instead of six: this sits on the pivot chain of every diagonal tile.
__device__ __forceinline__ double fast_rsqrt(double p) {
  const double y = __builtin_amdgcn_rsq(p);
  const double e = fma(-(p * y), y, 1.0);
  return fma(y * e, fma(0.375, e, 0.5), y);
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

// =======================================================================================
// MFMA-blocked diagonal kernel: one workgroup (8 waves) factors a 128 x 128 block and forms its
// inverse with BOTH matrices resident in MFMA accumulators.  The block is an 8 x 8 grid of
// 16 x 16 tiles; a wave owns one block row (tiles (w, 0..w)) of A and of W.  Per block step k:
//   wave k      : 16 x 16 Cholesky + triangular inverse of its diagonal tile (diag16s for fp64: the tile
//                 stays in one accumulator, 4-column block steps on the matrix pipe; diag16 for fp32: a
//                 column sweep with v_readlane broadcasts) -- the only sequential part;
//                 afterwards W(k, :) <- W_dd W(k, :) by MFMA
//   waves i > k : L(i,k) = A(i,k) W_dd^T (MFMA), published through LDS;
//                 A(i,j) -= L(i,k) L(j,k)^T for k < j <= i and W(i,j) -= L(i,k) W_dd W(k,j) for j <= k
// Two barriers per block step (16 per block) instead of one per column (128), all O(n^3) work on the matrix
// pipe, and the next diagonal wave is never kept waiting by anything but its own panel tile (see
// diag128_core).  FACTOR = false: A already holds L, only W is formed.
constexpr int TS = 17;   // LDS tile row stride in elements (16 + 1: conflict-free fragment reads)
template <typename T>
struct DiagScratch {};      // fp64: the tile routine works on S.D / S.Wd directly
template <>
struct DiagScratch<float> { // fp32: the 16 x 16 diagonal tile is factored in double (as the reference's dpotrf does,
  double D[16 * TS];        // tensors.py:198,219) by the same routine as the fp64 path
  double W[16 * TS];
};
template <typename T>
struct DiagLds {
  DiagScratch<T> x;
  T D[2][16 * TS];        // [step parity] diagonal tile, then L_dd
  T Wd[2][16 * TS];       // [step parity] W_dd = inv(L_dd)
  T P[8][16 * TS];        // panel tiles L(i,k), i = block row (also scratch for the raw A(i,k))
  T Tt[8][16 * TS];       // T_i = L(i,k) W_dd, private to wave i
  T Wr[2][8][16 * TS];    // [step parity] row k of W BEFORE its scaling by W_dd: tiles W(k, j), j < k
};

__device__ __forceinline__ double readlane_t(double v, int l) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
  return u.d;
}
__device__ __forceinline__ float readlane_t(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <typename T>
struct TileOps {
  using M = MfmaT<T>;
  using acc_t = typename M::acc_t;
  // accumulator (C/D layout) <-> LDS tile
  static __device__ __forceinline__ void store(T* t, const acc_t& a, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) t[M::row(lane, r) * TS + (lane & 15)] = a[r];
  }
  static __device__ __forceinline__ acc_t load(const T* t, int lane) {
    acc_t a;
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = t[M::row(lane, r) * TS + (lane & 15)];
    return a;
  }
  // acc += sign * X * Y^T   (X, Y LDS tiles): A[i][k] = X[i][k], B[k][j] = Y[j][k]
  static __device__ __forceinline__ acc_t mul_nt(const T* X, const T* Y, acc_t acc, T sign, int lane) {
    const int o = (lane & 15) * TS + (lane >> 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = M::mfma(sign * X[o + 4 * s], Y[o + 4 * s], acc);
    return acc;
  }
  // acc += sign * X * Y     (X, Y LDS tiles): A[i][k] = X[i][k], B[k][j] = Y[k][j]
  static __device__ __forceinline__ acc_t mul_nn(const T* X, const T* Y, acc_t acc, T sign, int lane) {
    const int oa = (lane & 15) * TS + (lane >> 4);
    const int ob = (lane >> 4) * TS + (lane & 15);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = M::mfma(sign * X[oa + 4 * s], Y[ob + 4 * s * TS], acc);
    return acc;
  }
};

// 16 x 16 Cholesky + triangular inverse by ONE wave: lane i < 16 holds row i of the tile.
// In: D (LDS, lower triangle valid).  Out: D <- L_dd (upper zero), Wd <- inv(L_dd).
template <typename T, bool FACTOR>
__device__ __noinline__ void diag16(T* D, T* Wd, int lane, int* info, int64_t base) {
  const int i = lane & 15;
  T row[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) row[c] = D[i * TS + c];
  // Factor and invert in ONE column sweep.  Lane i holds row i of the tile and column i of
  // W = inv(L): w[r] = W[r][i] = (delta_ri - sum_{k<r} L[r][k] W[k][i]) / L[r][r].  Once column j
  // of L is final, the scalars L[c][j] (c > j) that the trailing update broadcasts are exactly
  // the ones the forward substitution for W needs, so both use the same v_readlane.
  T w[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) w[r] = (r == i) ? T(1) : T(0);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    T p = readlane_t(row[j], j);
    T rpj, lj;
    if (FACTOR) {
      if (!(p > T(0))) {   // also catches NaN
        if (lane == 0) atomicCAS(info, 0, (int)(base + j + 1));
        p = T(1);
      }
      rpj = fast_rsqrt(p);
      T dg = p * rpj;
      dg = fma(T(0.5) * rpj, fma(-dg, dg, p), dg);
      lj = (i > j) ? row[j] * rpj : ((i == j) ? dg : T(0));
    } else {
      rpj = fast_rcp(p);
      lj = (i >= j) ? row[j] : T(0);
    }
    row[j] = lj;
    const T wj = w[j] * rpj;          // row j of W is final
    w[j] = wj;
#pragma unroll
    for (int c = j + 1; c < 16; ++c) {
      const T s = readlane_t(lj, c);  // L[c][j]
      if (FACTOR) row[c] = fma(-lj, s, row[c]);
      w[c] = fma(-s, wj, w[c]);
    }
  }
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c) D[i * TS + c] = row[c];
#pragma unroll
    for (int r = 0; r < 16; ++r) Wd[r * TS + i] = (r >= i) ? w[r] : T(0);
  }
}

#ifdef G3_DIAG_TIMING   // measurement build only: 100 MHz timestamps of the block steps (wave 7 and the diagonal waves)
__device__ unsigned long long g3_dbg_ts[128];
extern "C" int g3_dbg_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g3_dbg_ts), sizeof(g3_dbg_ts));
}
#define G3_TS(slot) do { if (lane == 0) g3_dbg_ts[slot] = wall_clock64(); } while (0)
// shader-cycle counter (20 bits, no waitcnt side effects) inside the 16 x 16 tile routine, first tile only
#define G3_CY(slot) do { if (base == 0 && lane == 0) g3_dbg_ts[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define G3_TS(slot) do { } while (0)
#define G3_CY(slot) do { } while (0)
#endif

// fp64 variant of diag16 that keeps the O(16^3) part on the matrix pipe.  The 16 x 16 tile is
// processed in four 4-column block steps; per step every lane redundantly factors and inverts the
// 4 x 4 diagonal block from LDS broadcast reads (the only sequential arithmetic: 4 pivots instead
// of 16), forms ITS element of the 16 x 4 panel directly in MFMA operand layout (lane = (row, k)),
// and one v_mfma_f64_16x16x4 applies the rank-4 trailing update to the whole tile.  The inverse is
// carried along the same way (W rows of the block by one MFMA, rows below by another); for f64
// accumulator register q holds exactly tile rows 4q..4q+3, so those MFMAs take their B operand
// straight from the accumulator with no cross-lane traffic.
template <bool FACTOR>
__device__ __noinline__ void diag16m(double* D, double* Wd, int lane, int* info, int64_t base) {
  using M = MfmaT<double>;
  using acc_t = typename M::acc_t;
  const int row = lane & 15, kq = lane >> 4;      // this lane's (row, k) in MFMA A-operand layout
  acc_t tA = TileOps<double>::load(D, lane);      // the tile, C layout
  acc_t aW;
  double pvq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) aW[r] = (M::row(lane, r) == (lane & 15)) ? 1.0 : 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c0 = 4 * q;
    G3_CY(96 + 8 * q + 0);
    // 1. the 4 x 4 diagonal block: Cholesky factor l and inverse w4 (every lane, uniform data)
    double d[4][4], l[4][4], w4[4][4], rp[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        d[a][b] = (b <= a) ? D[(c0 + a) * TS + c0 + b] : 0.0;
        l[a][b] = 0.0;
        w4[a][b] = 0.0;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double p = d[j][j];
      if (FACTOR) {
        if (!(p > 0.0)) {   // also catches NaN
          if (lane == 0) atomicCAS(info, 0, (int)(base + c0 + j + 1));
          p = 1.0;
        }
        rp[j] = fast_rsqrt(p);
        double dg = p * rp[j];
        dg = fma(0.5 * rp[j], fma(-dg, dg, p), dg);
        l[j][j] = dg;
#pragma unroll
        for (int a = j + 1; a < 4; ++a) l[a][j] = d[a][j] * rp[j];
#pragma unroll
        for (int a = j + 1; a < 4; ++a)
#pragma unroll
          for (int b = j + 1; b <= a; ++b) d[a][b] = fma(-l[a][j], l[b][j], d[a][b]);
      } else {
        rp[j] = fast_rcp(p);
#pragma unroll
        for (int a = j; a < 4; ++a) l[a][j] = d[a][j];
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      w4[b][b] = rp[b];
#pragma unroll
      for (int a = b + 1; a < 4; ++a) {
        double acc = 0.0;
#pragma unroll
        for (int m = b; m < a; ++m) acc = fma(l[a][m], w4[m][b], acc);
        w4[a][b] = -acc * rp[a];
      }
    }
    G3_CY(96 + 8 * q + 1);
    // 2. this lane's panel element P[row][kq] = sum_{m <= kq} T[row][c0 + m] * w4[kq][m]
    double pv;
    const int ra = row - c0;                      // row inside the diagonal block when 0 <= ra < 4
    if (FACTOR) {
      double t[4], cf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        t[m] = D[row * TS + c0 + m];
        cf[m] = 0.0;
#pragma unroll
        for (int k = m; k < 4; ++k) cf[m] = (kq == k) ? w4[k][m] : cf[m];
      }
      pv = t[0] * cf[0];
#pragma unroll
      for (int m = 1; m < 4; ++m) pv = fma(t[m], cf[m], pv);
      if (ra < 4) {                               // the diagonal block itself (exact factor), zero above it
        double lv = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) lv = (ra == a && kq == b) ? l[a][b] : lv;
        pv = lv;
      }
    } else {
      pv = (ra >= 4 || (ra >= 0 && kq <= ra)) ? D[row * TS + c0 + kq] : 0.0;
    }
    G3_CY(96 + 8 * q + 2);
    // 3. rank-4 trailing update of the whole tile, back to LDS for the next block step
    if (FACTOR) {
      tA = M::mfma(-pv, pv, tA);
      TileOps<double>::store(D, tA, lane);
    }
    // 4. the finished columns c0..c0+3 of L: kept until the sweep is over (the tile stores of the
    //    later block steps overwrite the whole tile)
    pvq[q] = pv;
    G3_CY(96 + 8 * q + 3);
    // 5. inverse: rows of this block, then the rows below
    double ah = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) ah = (ra == a && kq == b) ? w4[a][b] : ah;
    const acc_t R = M::mfma(ah, aW[q], acc_t{0, 0, 0, 0});
    const double pb = (ra >= 4) ? pv : 0.0;
    aW = M::mfma(-pb, R[q], aW);
    aW[q] = R[q];
    G3_CY(96 + 8 * q + 4);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) D[row * TS + 4 * q + kq] = pvq[q];
  TileOps<double>::store(Wd, aW, lane);
}

// Register-resident successor of diag16m<true> (fp64, factor + inverse).  The tile is kept SYMMETRIC in one
// accumulator, and that removes every LDS round trip from the block steps:
//   * C layout: register q of lane (g, c) is T[4q + g][c]; by symmetry that is T[c][4q + g], i.e. exactly the
//     element lane (row = c, k = g) must supply as the A operand -- or lane (col = c, k = g) as the B operand --
//     of a 16x16x4 product with the column block T[:, 4q .. 4q+3].  The accumulator register IS the operand.
//   * the 4 x 4 diagonal block of step q sits in register q of lanes (a, 4q + b): ten v_readlane pairs.
//   * the panel P = T[:, c0..c0+3] w4^T comes out of ONE MFMA as P^T = w4 T[:, c0..]^T: rows 0..3 of the
//     result are register 0 of lane (i, n) = P[n][i] -- the (row, k) operand layout the rank-4 update
//     needs for both of its operands, again without moving anything.
// Per block step: readlanes -> 4 x 4 factor (uniform, every lane) with the lane's own column of the inverse
// carried along -> MFMA (panel) -> MFMA (rank-4 update); the two inverse MFMAs are off the chain.
__device__ __noinline__ void diag16s(double* D, double* Wd, int lane, int* info, int64_t base) {
  using M = MfmaT<double>;
  using acc_t = typename M::acc_t;
  const int row = lane & 15, kq = lane >> 4;
  const int r3 = row & 3, rq = row >> 2;
  acc_t tS, aW;
#pragma unroll
  for (int r = 0; r < 4; ++r) {                   // symmetric copy of the tile (its lower triangle is valid)
    const int i = 4 * r + kq, c = row;
    tS[r] = (i >= c) ? D[i * TS + c] : D[c * TS + i];
    aW[r] = (i == c) ? 1.0 : 0.0;
  }
  double e[4];                                    // column kq of the 4 x 4 identity: this lane inverts against it
#pragma unroll
  for (int a = 0; a < 4; ++a) e[a] = (kq == a) ? 1.0 : 0.0;
  double pvq[4];
  int first_bad = 0;                              // 1-based column of the first failing pivot (uniform)
  acc_t R = acc_t{0, 0, 0, 0};                    // rows of the previous block step's inverse, not yet applied below
  double pb = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c0 = 4 * q;
    G3_CY(96 + 8 * q + 0);
    // 1. the 4 x 4 diagonal block (uniform values, every lane): factor l; column kq of its inverse in x
    double d[4][4], l[4][4], rp[4], x[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        d[a][b] = (b <= a) ? readlane_t(tS[q], a * 16 + c0 + b) : 0.0;
        l[a][b] = 0.0;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double p = d[j][j];
      // a non-positive (or NaN) pivot is only RECORDED here (uniform compare, scalar bookkeeping) and reported
      // after the tile: the pivot chain carries no select and no branch.  The NaNs it then produces stay in
      // this matrix, whose factorisation is discarded (every later kernel of the sweep tests the flag).
      if (!(p > 0.0) && first_bad == 0) first_bad = c0 + j + 1;
      rp[j] = fast_rsqrt(p);
#pragma unroll
      for (int a = j + 1; a < 4; ++a) l[a][j] = d[a][j] * rp[j];
#pragma unroll
      for (int a = j + 1; a < 4; ++a)
#pragma unroll
        for (int b = j + 1; b <= a; ++b) d[a][b] = fma(-l[a][j], l[b][j], d[a][b]);
      // forward substitution l x = e_kq, one row per pivot: x[j] = (e[j] - sum_{m<j} l[j][m] x[m]) / l[j][j].
      // Lanes of group kq get column kq of the inverse (zeros above the diagonal come out by themselves), so
      // the entry (r3, kq) a lane must supply as an MFMA operand is x[r3] -- no 10-way select.
      double acc = e[j];
#pragma unroll
      for (int m = 0; m < j; ++m) acc = fma(-l[j][m], x[m], acc);
      x[j] = acc * rp[j];
    }
    G3_CY(96 + 8 * q + 1);
    // 2. the previous block step's inverse rows go into the rows below them now: the matrix pipe is idle
    //    during the factor, and issuing this product any earlier would stall the instruction stream on R
    if (q > 0) {
      aW = M::mfma(-pb, R[q - 1], aW);
      aW[q - 1] = R[q - 1];
    }
    // 3. w4 placed on the rows of this block: the A operand of both products below
    const double wsel = (r3 < 2) ? (r3 == 0 ? x[0] : x[1]) : (r3 == 2 ? x[2] : x[3]);
    const double ah = (rq == q) ? wsel : 0.0;
    // 4. panel: rows c0..c0+3 of w4 T[c0..c0+3, :] are P^T, i.e. register q of the product is P[row][kq].  For
    //    the rows of the diagonal block itself that is w4 (l l^T) = l^T transposed back: the factor l, to rounding
    const acc_t pt = M::mfma(ah, tS[q], acc_t{0, 0, 0, 0});
    const double pv = (rq < q || (rq == q && kq > r3)) ? 0.0 : pt[q];
    G3_CY(96 + 8 * q + 2);
    // 5. rank-4 update of the (symmetric) tile; columns c0..c0+3 of L are final
    tS = M::mfma(-pv, pv, tS);
    pvq[q] = pv;
    G3_CY(96 + 8 * q + 3);
    // 6. inverse rows of this block (W[c0.., :] <- w4 W[c0.., :]); applied to the rows below in the next step
    R = M::mfma(ah, aW[q], acc_t{0, 0, 0, 0});
    pb = (rq > q) ? pv : 0.0;
    __builtin_amdgcn_sched_barrier(0);            // keep the order above: the scheduler must not move the next
    G3_CY(96 + 8 * q + 4);                        // step's readlanes behind a product that waits for R
  }
  aW[3] = R[3];                                   // (no rows below the last block)
  if (first_bad != 0 && lane == 0) atomicCAS(info, 0, (int)(base + first_bad));
#pragma unroll
  for (int q = 0; q < 4; ++q) D[row * TS + 4 * q + kq] = pvq[q];
  TileOps<double>::store(Wd, aW, lane);
}

// fp32 front end of diag16s: the tile goes through double (16 x 16 elements: four conversions per lane each way).
// CholeskyRobust factors float32 covariances in float64 and casts back (tensors.py:198,219); doing that at least
// for the diagonal tiles -- where the pivots are -- follows it more closely than an fp32 sweep, and the fp64 tile
// routine is the faster one (2.65 us against 3.4).
__device__ __forceinline__ void diag16s_f32(float* D, float* Wd, double* Dd, double* Wdd, int lane, int* info, int64_t base) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = lane + 64 * r, i = e >> 4, c = e & 15;
    Dd[i * TS + c] = (double)D[i * TS + c];
  }
  diag16s(Dd, Wdd, lane, info, base);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = lane + 64 * r, i = e >> 4, c = e & 15;
    D[i * TS + c] = (float)Dd[i * TS + c];
    Wd[i * TS + c] = (float)Wdd[i * TS + c];
  }
}

// The program of wave W (block row W) of the diagonal-block kernel, with W a compile-time constant:
// every "does this wave take part" test folds away, so the register allocator sees the true
// lifetime of each accumulator tile instead of the union over all waves.
//
// Register budget: the kernel sits on the critical path while the bulk GEMM keeps every CU busy
// with two 4-wave workgroups of ~224 VGPRs.  It must fit next to ONE of them (the half CU that a
// retiring GEMM workgroup leaves behind): 8 waves x <= 144 VGPRs and <= 96 KiB of LDS.  Otherwise
// it waits for a whole CU to drain -- measured 494 us per launch instead of 43.  So accumulator
// tiles exist only while they are live: A(W, j) is written out the moment it is final (step j),
// W(W, j) comes into existence at step j, and nothing is live across the call that factors the
// diagonal tile (wave W publishes its old W row before it, and rebuilds the row from LDS after).
// diag128_core works on the caller's accumulator tiles aA (tiles (W, 0..W) of the block, lower part):
// diag128_wave loads them from A; the fused 256-wide kernel below hands over the second diagonal block
// straight from the registers it has just updated it in.  WT != nullptr: the finished inverse is also left
// in LDS as 16 x 16 tiles (tile (i, j), j <= i, at index i (i + 1) / 2 + j), aliasing S -- the core
// synchronises before and after writing them.
template <typename T, bool FACTOR, int W>
__device__ __forceinline__ void diag128_core(typename TileOps<T>::acc_t (&aA)[8], T* A, int64_t ld, T* Wg, int64_t ldw,
                                             int* info, int64_t row_base, DiagLds<T>& S, int lane, T* WT) {
  using TO = TileOps<T>;
  using acc_t = typename TO::acc_t;
  using M = MfmaT<T>;
  acc_t aW[8];
  // Per block step k (two barriers):
  //   before B1   wave k: 16 x 16 factor + inverse of its diagonal tile (the only sequential part);
  //               waves < k (finished): bring row k of W, parked in LDS, up to date with step k-1
  //   B1 .. B2    waves > k: panel tile L(i,k); wave k+1 also finishes ITS diagonal tile from its own
  //               panel tile, hands it over in S.D and parks its row of W
  //   after B2    wave k+1 goes straight into the next tile routine; waves > k+1: T_i = L(i,k) W_dd and the
  //               trailing updates of A and W; wave k: W(k,:) <- W_dd W_old(k,:)
  // Everything written for step k+1 while step k's readers may still run lives in the other-parity buffers
  // (S.D, S.Wd, S.Wr), and each S.P / S.Tt tile has a single writer, so two barriers per step are enough.
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int par = k & 1;
    if (W == 7) G3_TS(8 * k + 0);
    if (W == k) {
      G3_TS(64 + 2 * k);
      if (k == 0) TO::store(S.D[par], aA[k], lane);      // (k > 0: handed over in the previous step, below)
      if constexpr (sizeof(T) == 8 && FACTOR)          // fp64 factor + inverse: the register-resident symmetric tile routine
        diag16s((double*)S.D[par], (double*)S.Wd[par], lane, info, row_base + 16 * k);
      else if constexpr (sizeof(T) == 8)               // fp64, inverse of an already factored block
        diag16m<FACTOR>((double*)S.D[par], (double*)S.Wd[par], lane, info, row_base + 16 * k);
      else if constexpr (FACTOR)                       // fp32 factor: through the fp64 routine (tensors.py:198,219)
        diag16s_f32((float*)S.D[par], (float*)S.Wd[par], S.x.D, S.x.W, lane, info, row_base + 16 * k);
      else
        diag16<T, FACTOR>(S.D[par], S.Wd[par], lane, info, row_base + 16 * k);
      if (FACTOR) {                         // the diagonal tile of L is final: write it out
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = M::row(lane, r), col = lane & 15;
          if (row >= col) A[(int64_t)(16 * k + row) * ld + 16 * k + col] = S.D[par][row * TS + col];
        }
      }
      G3_TS(64 + 2 * k + 1);
    } else if (W < k) {
      // A finished wave helps the diagonal wave, which is busy with the 16 x 16 tile: row k of W was parked in
      // S.Wr[par] one update short (step k-1's); wave W brings tile (k, W) up to date.
      //   T = L(k,k-1) W_dd(k-1);  W(k,j) -= T W_old(k-1,j) for j < k-1;  W(k,k-1) = -T
      const acc_t t = TO::mul_nn(S.P[k], S.Wd[par ^ 1], acc_t{0, 0, 0, 0}, T(1), lane);
      if (W == k - 1) {
        TO::store(S.Wr[par][W], -t, lane);
      } else {
        TO::store(S.Tt[W], t, lane);        // (this wave's own buffer: free since its block row is finished)
        const acc_t w = TO::mul_nn(S.Tt[W], S.Wr[par ^ 1][W], TO::load(S.Wr[par][W], lane), T(-1), lane);
        TO::store(S.Wr[par][W], w, lane);
      }
    }
    __syncthreads();                       // B1: W_dd and the old row k of W are published
    if (W == 7) G3_TS(8 * k + 1);
    if (W > k) {
      TO::store(S.P[W], aA[k], lane);
      if (FACTOR) {
        const acc_t lk = TO::mul_nt(S.P[W], S.Wd[par], acc_t{0, 0, 0, 0}, T(1), lane);   // L(W,k) = A(W,k) W_dd^T
        TO::store(S.P[W], lk, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)         // final: write it out, its registers are free from here on
          A[(int64_t)(16 * W + M::row(lane, r)) * ld + 16 * k + (lane & 15)] = lk[r];
      }
      if (W == k + 1) {
        // The NEXT diagonal wave: all its diagonal tile still needs is its own panel tile, so the tile is
        // finished and handed over here, before B2, and the wave starts the next 16 x 16 factorisation the
        // moment B2 falls.  Its row of W is parked in LDS (nothing stays live across the tile routine) one
        // update short; the finished waves complete it meanwhile (above).
        if (FACTOR) aA[k + 1] = TO::mul_nt(S.P[W], S.P[W], aA[k + 1], T(-1), lane);
        TO::store(S.D[par ^ 1], aA[k + 1], lane);
#pragma unroll
        for (int j = 0; j < k; ++j) TO::store(S.Wr[par ^ 1][j], aW[j], lane);
      }
    }
    if (W == 7) G3_TS(8 * k + 2);
    __syncthreads();                       // B2: the panel L(:, k) is published
    if (W == 7) G3_TS(8 * k + 3);
    if (W > k + 1) {
      // T_W = L(W,k) W_dd: W(W,j) -= T_W W_old(k,j) for j < k, and W(W,k) = -T_W (W_old(k,k) = I)
      const acc_t t = TO::mul_nn(S.P[W], S.Wd[par], acc_t{0, 0, 0, 0}, T(1), lane);
      aW[k] = -t;
      TO::store(S.Tt[W], t, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (FACTOR && j > k && j <= W) aA[j] = TO::mul_nt(S.P[W], S.P[j], aA[j], T(-1), lane);
        if (j < k) aW[j] = TO::mul_nn(S.Tt[W], S.Wr[par][j], aW[j], T(-1), lane);
        if (sizeof(T) == 8 && (j & 1)) __builtin_amdgcn_sched_barrier(0);   // operand loads at most two tiles ahead (144-VGPR budget)
      }
    } else if (W == k) {          // off the critical path: W(k, :) <- W_dd W_old(k, :)
#pragma unroll
      for (int j = 0; j < k; ++j) aW[j] = TO::mul_nn(S.Wd[par], S.Wr[par][j], acc_t{0, 0, 0, 0}, T(1), lane);
      aW[k] = TO::load(S.Wd[par], lane);
    }
    if (W == 7) G3_TS(8 * k + 4);
  }
  // write back W (full 128 x 128 block row, upper part zero); L went out tile by tile
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * W + M::row(lane, r), col = 16 * j + (lane & 15);
      T v = T(0);
      if (j <= W) v = (row >= col) ? aW[j <= W ? j : 0][r] : T(0);
      Wg[(int64_t)row * ldw + col] = v;
    }
  }
  if (WT != nullptr) {
    __syncthreads();                       // every wave is done with S: its memory becomes the tile array
#pragma unroll
    for (int j = 0; j <= W; ++j) {
      acc_t t = aW[j];
      if (j == W) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = (M::row(lane, r) >= (lane & 15)) ? t[r] : T(0);
      }
      TO::store(WT + (W * (W + 1) / 2 + j) * (16 * TS), t, lane);
    }
    __syncthreads();
  }
}

template <typename T, bool FACTOR, int W>
__device__ __forceinline__ void diag128_wave(T* A, int64_t ld, T* Wg, int64_t ldw, int* info, int64_t row_base,
                                             DiagLds<T>& S, int lane, T* WT = nullptr) {
  using M = MfmaT<T>;
  typename TileOps<T>::acc_t aA[8];
#pragma unroll
  for (int j = 0; j <= W; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * W + M::row(lane, r), col = 16 * j + (lane & 15);
      aA[j][r] = (row >= col) ? A[(int64_t)row * ld + col] : T(0);
    }
  }
  diag128_core<T, FACTOR, W>(aA, A, ld, Wg, ldw, info, row_base, S, lane, WT);
}

// ---- fused factorisation of a 256 x 256 diagonal block by ONE workgroup:
//   diag(A00) -> L10 = A10 W0^T -> A11 -= L10 L10^T -> diag(A11)
// i.e. the four launches of potrf_rec(256) (two diagonal kernels, one in-place panel product, one SYRK)
// as one: on the critical path of every factorisation each launch costs its ~12 us of fixed latency --
// 2-4x that while a bulk update streams through the same CUs -- for 4 MFLOP of work.  The two products
// run on the same eight waves: W0 stays in LDS as tiles (in the memory the diagonal phase has
// finished with), A10 is read straight into MFMA operand registers, L10 goes through LDS in two
// 64-column halves for the SYRK, and the updated A11 never leaves the accumulators it is factored from.
template <typename T, int W>
__device__ __forceinline__ void potrf256_wave(T* A, int64_t ld, T* Wg, int* info, int64_t row_base, DiagLds<T>& S, int lane) {
  using TO = TileOps<T>;
  using acc_t = typename TO::acc_t;
  using M = MfmaT<T>;
  constexpr int TSZ = 16 * TS;
  T* tiles = reinterpret_cast<T*>(&S);                                  // aliases S between the diagonal phases
  static_assert(sizeof(DiagLds<T>) >= 36 * TSZ * sizeof(T), "tile array must fit in the diagonal kernel's LDS");
  if (W == 7) G3_TS(88);
  diag128_wave<T, true, W>(A, ld, Wg, G3_LB, info, row_base, S, lane, tiles);
  if (W == 7) G3_TS(89);
  T* A10 = A + (int64_t)G3_LB * ld;
  T* A11 = A10 + G3_LB;
  const int frow = lane & 15, kq = lane >> 4;
  // ---- L10(W, j) = sum_{c <= j} A10(W, c) W0(j, c)^T
  acc_t L[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) L[j] = acc_t{0, 0, 0, 0};
  T an[4];                                 // operand of the NEXT column tile: its global round trip hides under this one's products
#pragma unroll
  for (int q = 0; q < 4; ++q) an[q] = A10[(int64_t)(16 * W + frow) * ld + 4 * q + kq];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    T a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = an[q];
    if (c + 1 < 8) {
#pragma unroll
      for (int q = 0; q < 4; ++q) an[q] = A10[(int64_t)(16 * W + frow) * ld + 16 * (c + 1) + 4 * q + kq];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = c; j < 8; ++j) {
      const T* y = tiles + (j * (j + 1) / 2 + c) * TSZ + frow * TS + kq;
#pragma unroll
      for (int q = 0; q < 4; ++q) L[j] = M::mfma(a[q], y[4 * q], L[j]);
      __builtin_amdgcn_sched_barrier(0);   // one tile product at a time: the scheduler must not hoist later
    }                                      // operand loads into this one's registers (144-VGPR budget)
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) A10[(int64_t)(16 * W + M::row(lane, r)) * ld + 16 * j + frow] = L[j][r];
  if (W == 7) G3_TS(90);
  // ---- A11(W, j) -= sum_c L10(W, c) L10(j, c)^T, j <= W, in two halves of c through LDS.  (A11 is
  // loaded only after the first half of L10 has left the registers: the kernel must stay within the 144
  // VGPRs that let it run next to a bulk GEMM workgroup, like the 128-wide diagonal kernel.)
  acc_t aA[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                       // the W0 tiles (h = 0) / the first half (h = 1) are no longer read
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) TO::store(tiles + (W * 4 + cc) * TSZ, L[4 * h + cc], lane);
    __syncthreads();
    if (h == 0) {
#pragma unroll
      for (int j = 0; j <= W; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * W + M::row(lane, r), col = 16 * j + frow;
          aA[j][r] = (row >= col) ? A11[(int64_t)row * ld + col] : T(0);
        }
    }
#pragma unroll
    for (int j = 0; j <= W; ++j)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        aA[j] = TO::mul_nt(tiles + (W * 4 + cc) * TSZ, tiles + (j * 4 + cc) * TSZ, aA[j], T(-1), lane);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) aA[W][r] = (M::row(lane, r) >= frow) ? aA[W][r] : T(0);   // strict upper part of the diagonal tile
  __syncthreads();                         // S is about to be used as DiagLds again
  if (W == 7) G3_TS(91);
  diag128_core<T, true, W>(aA, A11, ld, Wg + G3_LB * G3_LB, G3_LB, info, row_base + G3_LB, S, lane, nullptr);
  if (W == 7) G3_TS(92);
}

