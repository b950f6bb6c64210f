// Microbenchmark for the Gram store rate: the SE fast path of g3_gram.hip re-stated with
// knobs, lower triangle of an N x N fp64 matrix, d = 4.  Separates the store pattern from the arithmetic:
//   MATH 0: ocml exp (what the library used up to round 4)   1: no exp (store pattern alone)   2: written-out exp
//   NT   0: ordinary stores   1: non-temporal stores
//   TR      rows of a workgroup's tile (64 / 128),  RP = row phases (4: 256 threads)
// hipcc --offload-arch=gfx950 -O3 scripts/gram_rate.hip -o /tmp/gram_rate && /tmp/gram_rate [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

__device__ __forceinline__ double exp_fast(double x) {
  // exp for finite or -inf arguments: k = rint(x log2 e), r = x - k ln 2 (two-term), degree-13 Taylor on |r| <= 0.347
  const double xc = fmin(fmax(x, -1000.0), 710.0);
  const double k = rint(xc * 1.4426950408889634);
  double r = fma(-k, 6.93147180369123816490e-01, xc);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double v = ldexp(p, (int)k);
  return x != x ? x : v;
}

template <int MATH, int NT, int TR, int NTH = 256, int UNR = 4>
__global__ void __launch_bounds__(NTH) gram_se(const double* __restrict__ X, int64_t n, double* __restrict__ K, int64_t ldk,
                                               double w0, double w1, double w2, double w3, double var, double noise) {
  constexpr int D = 4, GTN = 128, DP = 5;
  constexpr int RB = TR / 64;   // 64-row units per tile: a tile on the diagonal band spans RB * 64 rows
  // 1-D grid over tiles on or below the diagonal: tile row b (TR rows) has floor(b * TR / GTN) + 1 column tiles
  const int64_t id = blockIdx.x;
  int64_t bi, bj;
  if (TR == 64) {
    int64_t q = (int64_t)((sqrt(1.0 + 4.0 * (double)id) - 1.0) * 0.5);
    while ((q + 1) * (q + 2) <= id) ++q;
    while (q * (q + 1) > id) --q;
    int64_t rem = id - q * (q + 1);
    const int64_t r = rem >= q + 1 ? 1 : 0;
    if (r) rem -= q + 1;
    bi = 2 * q + r; bj = rem;
  } else if (TR == 128) {   // row b has b + 1 tiles
    int64_t q = (int64_t)((sqrt(1.0 + 8.0 * (double)id) - 1.0) * 0.5);
    while ((q + 1) * (q + 2) / 2 <= id) ++q;
    while (q * (q + 1) / 2 > id) --q;
    bi = q; bj = id - q * (q + 1) / 2;
  } else {                  // TR == 256: row b has 2 b + 2 tiles
    int64_t q = (int64_t)((sqrt(1.0 + 4.0 * (double)id) - 1.0) * 0.5);
    while ((q + 1) * (q + 2) <= id) ++q;
    while (q * (q + 1) > id) --q;
    bi = q; bj = id - q * (q + 1);
  }
  const int64_t i0 = bi * TR, j0 = bj * GTN;
  __shared__ double xi_s[TR * DP], xj_s[GTN * DP];
  const int tid = threadIdx.x;
  for (int e = tid; e < TR * D; e += NTH) { const int r = e / D, c = e - r * D; xi_s[r * DP + c] = X[(i0 + r) * D + c]; }
  for (int e = tid; e < GTN * D; e += NTH) { const int r = e / D, c = e - r * D; xj_s[r * DP + c] = X[(j0 + r) * D + c]; }
  __syncthreads();
  const int tx = tid & 63, ty = tid >> 6;
  const int64_t ja = j0 + 2 * tx;
  double xa[D], xb[D];
  const double w[4] = {w0, w1, w2, w3};
#pragma unroll
  for (int c = 0; c < D; ++c) { xa[c] = xj_s[2 * tx * DP + c]; xb[c] = xj_s[(2 * tx + 1) * DP + c]; }
#pragma unroll UNR
  for (int rr = ty; rr < TR; rr += NTH / 64) {
    const int64_t i = i0 + rr;
    double v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      double dd = 0;
#pragma unroll
      for (int c = 0; c < D; ++c) { const double dx = xi_s[rr * DP + c] - (q ? xb[c] : xa[c]); dd += (dx * dx) * w[c]; }
      double kv;
      if (MATH == 0) kv = exp(-dd);
      else if (MATH == 1) kv = 1.0 - dd;
      else kv = exp_fast(-dd);
      v[q] = var * kv;
      if (i == ja + q) v[q] += noise;
    }
    typedef double vec2 __attribute__((ext_vector_type(2)));
    vec2* p = reinterpret_cast<vec2*>(K + i * ldk + ja);
    if (NT) __builtin_nontemporal_store(vec2{v[0], v[1]}, p);
    else *p = vec2{v[0], v[1]};
  }
  (void)RB;
}

// store-only streaming: full dense rows, the plainest pattern (how fast can this chip write 4.3 GB?)
template <int NT>
__global__ void __launch_bounds__(256) fill(double* __restrict__ K, int64_t nvec) {
  typedef double vec2 __attribute__((ext_vector_type(2)));
  vec2* p = reinterpret_cast<vec2*>(K);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(vec2{1.0, 2.0}, p + i); else p[i] = vec2{1.0, 2.0};
  }
}

template <typename F>
static double time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 32768;
  if (n % 128) { fprintf(stderr, "N must be a multiple of 128\n"); return 1; }
  std::vector<double> hx(n * 4);
  srand(1);
  const double side = pow((double)n, 0.25);
  for (auto& v : hx) v = side * (rand() / (double)RAND_MAX);
  double *X, *K;
  hipMalloc(&X, hx.size() * 8);
  hipMalloc(&K, (size_t)n * n * 8);
  hipMemcpy(X, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
  const double gb = (0.5 * n * (n + 1) * 8 + n * 4 * 8) / 1e9;
  const int64_t t64 = n / 64, q = t64 / 2, tiles64 = q * (q + 1);
  const int64_t t128 = n / 128, tiles128 = t128 * (t128 + 1) / 2;
  printf("N = %lld, lower triangle %.3f GB; 64x128 tiles %lld, 128x128 tiles %lld\n", (long long)n, gb, (long long)tiles64, (long long)tiles128);
#define RUN4(M, NTV, TRV, NTHV, UNRV)                                                                                  \
  {                                                                                                                   \
    const int64_t g = (TRV == 64) ? tiles64 : (TRV == 128 ? tiles128 : (n / 256) * (n / 256 + 1));                     \
    const double ms = time_ms([&] { hipLaunchKernelGGL((gram_se<M, NTV, TRV, NTHV, UNRV>), dim3((unsigned)g), dim3(NTHV), 0, 0, X, n, K, n, \
                                                       0.5, 0.4, 0.3, 0.2, 1.3, 0.1); }, 10);                          \
    printf("math %d nt %d rows %3d threads %d unroll %d: %.3f ms  %.2f TB/s\n", M, NTV, TRV, NTHV, UNRV, ms, gb / ms);  \
  }
#define RUN(M, NTV, TRV) RUN4(M, NTV, TRV, 256, 4)
  RUN(0, 0, 64) RUN(1, 0, 64) RUN(2, 0, 64)
  RUN(0, 1, 64) RUN(1, 1, 64) RUN(2, 1, 64)
  RUN(0, 0, 128) RUN(1, 0, 128) RUN(2, 0, 128)
  RUN(0, 1, 128) RUN(1, 1, 128) RUN(2, 1, 128)
  RUN(0, 0, 256) RUN(0, 1, 256) RUN(1, 1, 256)
  RUN4(0, 0, 128, 512, 4) RUN4(0, 1, 128, 512, 4) RUN4(0, 0, 256, 512, 4) RUN4(0, 1, 256, 512, 4)
  RUN4(0, 0, 128, 256, 8) RUN4(0, 1, 128, 256, 8) RUN4(0, 0, 128, 256, 2) RUN4(0, 1, 128, 256, 2)
  RUN4(0, 0, 256, 256, 8) RUN4(0, 1, 256, 256, 8) RUN4(0, 0, 128, 128, 4) RUN4(0, 1, 128, 128, 4)
  const int64_t nvec = (int64_t)(0.5 * n * (n + 1)) / 2;
  for (int blocks : {1024, 4096, 16384}) {
    double ms = time_ms([&] { hipLaunchKernelGGL((fill<0>), dim3(blocks), dim3(256), 0, 0, K, nvec); }, 10);
    printf("dense fill of the same bytes, %5d blocks, plain: %.3f ms  %.2f TB/s\n", blocks, ms, nvec * 16 / 1e9 / ms);
    ms = time_ms([&] { hipLaunchKernelGGL((fill<1>), dim3(blocks), dim3(256), 0, 0, K, nvec); }, 10);
    printf("dense fill of the same bytes, %5d blocks, nt   : %.3f ms  %.2f TB/s\n", blocks, ms, nvec * 16 / 1e9 / ms);
  }
  // accuracy of the written-out exp against the library's on the values the kernel sees
  hipLaunchKernelGGL((gram_se<0, 0, 64>), dim3((unsigned)tiles64), dim3(256), 0, 0, X, n, K, n, 0.5, 0.4, 0.3, 0.2, 1.3, 0.1);
  std::vector<double> r0(512 * 512), r2(512 * 512);
  hipMemcpy2D(r0.data(), 512 * 8, K + (n - 512) * n, n * 8, 512 * 8, 512, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL((gram_se<2, 0, 64>), dim3((unsigned)tiles64), dim3(256), 0, 0, X, n, K, n, 0.5, 0.4, 0.3, 0.2, 1.3, 0.1);
  hipMemcpy2D(r2.data(), 512 * 8, K + (n - 512) * n, n * 8, 512 * 8, 512, hipMemcpyDeviceToHost);
  double worst = 0;
  for (size_t i = 0; i < r0.size(); ++i)
    if (r0[i] > 1e-300) worst = fmax(worst, fabs(r2[i] - r0[i]) / r0[i]);
  printf("written-out exp vs library exp, worst relative difference over a 512 x 512 block: %.3g\n", worst);
  return 0;
}
