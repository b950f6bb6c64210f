// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (no memory traffic).
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// RANDOM: operands with full random mantissas that change every iteration (power-realistic)
template <int NACC, bool RANDOM = false>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  unsigned long long sa = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + blockIdx.x * 977), sb = sa ^ 0xD1B54A32D192ED03ull;
  for (int it = 0; it < iters; ++it) {
    if (RANDOM) {
      sa = sa * 6364136223846793005ull + 1442695040888963407ull;
      sb = sb * 6364136223846793005ull + 1442695040888963407ull;
      a = __longlong_as_double((sa >> 12) | 0x3FF0000000000000ull) - 1.5;   // uniform in [-0.5, 0.5)
      b = __longlong_as_double((sb >> 12) | 0x3FF0000000000000ull) - 1.5;
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool RANDOM = false>
void run(int blocks_per_cu, int threads) {
  int ncu = 256;
  double* out;
  hipMalloc(&out, sizeof(double) * ncu * blocks_per_cu * threads);
  int iters = 40000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, RANDOM>), dim3(ncu * blocks_per_cu), dim3(threads), 0, 0, out, 100, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, RANDOM>), dim3(ncu * blocks_per_cu), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 2048.0 * NACC * iters * (threads / 64) * ncu * blocks_per_cu;
  printf("%s NACC=%d blocks/CU=%d threads=%d: %.2f ms  %.2f TFLOP/s\n", RANDOM ? "random  " : "constant", NACC, blocks_per_cu, threads, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(1, 256); run<16>(1, 256); run<16>(2, 256); run<8>(4, 256);
  run<16, true>(1, 256); run<16, true>(2, 256); run<16, true>(2, 256); run<16>(2, 256);
  return 0;
}
