// Accuracy of v_rsq_f64 / v_rcp_f64 and of one / two Newton steps on gfx950 (development aid).
// hipcc --offload-arch=gfx950 -O3 scripts/rsq_probe.hip -o /tmp/rsq_probe && /tmp/rsq_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* p, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = p[i];
  double y0 = __builtin_amdgcn_rsq(x);
  double y1 = y0 * fma(-0.5 * x * y0, y0, 1.5);
  double y2 = y1 * fma(-0.5 * x * y1, y1, 1.5);
  double g = x * y1;                       // sqrt after one step, then the residual correction
  g = fma(0.5 * y1, fma(-g, g, x), g);
  // one third-order step: y (1 + e/2 + 3 e^2 / 8), e = 1 - x y^2  (24 -> 72 bits: a single step suffices)
  double e = fma(-(x * y0), y0, 1.0);
  double y3 = fma(y0 * e, fma(0.375, e, 0.5), y0);
  double r0 = __builtin_amdgcn_rcp(x);
  double r1 = r0 * fma(-x, r0, 2.0);
  out[7 * i + 0] = y0; out[7 * i + 1] = y1; out[7 * i + 2] = y2; out[7 * i + 3] = g; out[7 * i + 4] = r0; out[7 * i + 5] = r1; out[7 * i + 6] = y3;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double u = (s >> 11) * (1.0 / 9007199254740992.0);
    h[i] = std::exp((u - 0.5) * 60.0);
  }
  double *dp, *dout;
  hipMalloc(&dp, n * 8); hipMalloc(&dout, n * 56);
  hipMemcpy(dp, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dp, dout, n);
  std::vector<double> o(7 * n);
  hipMemcpy(o.data(), dout, n * 56, hipMemcpyDeviceToHost);
  double e[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    long double x = h[i], rs = 1.0L / sqrtl(x), sq = sqrtl(x), rc = 1.0L / x;
    long double ref[7] = {rs, rs, rs, sq, rc, rc, rs};
    for (int j = 0; j < 7; ++j) {
      double rel = (double)fabsl((o[7 * i + j] - ref[j]) / ref[j]);
      if (rel > e[j]) e[j] = rel;
    }
  }
  printf("max relative error (2^-53 = 1.1e-16): rsq %.3g, +1 Newton %.3g, +2 Newton %.3g, sqrt(1 Newton + residual) %.3g, rcp %.3g, rcp +1 Newton %.3g, rsq + one third-order step %.3g\n",
         e[0], e[1], e[2], e[3], e[4], e[5], e[6]);
  return 0;
}
