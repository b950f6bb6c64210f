// Development probe: CU-mask bit -> (XCC, SE, SH, CU) on gfx950.  Mask bit i is expected to select
// XCC i % 8, logical CU i / 8 (scripts/cumask_probe.hip); this prints, for every logical CU index c,
// the hardware ids a stream masked to {bits c*8 .. c*8+7} dispatches to.
// build: hipcc --offload-arch=gfx950 -O2 scripts/cumask_map.hip -o /tmp/cumask_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
#include <string>

__global__ void probe(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32, nb = 512;
  unsigned* d; hipMalloc(&d, nb * 2 * sizeof(unsigned));
  for (int c = 0; c < ncu / 8; ++c) {
    std::vector<uint32_t> m(words, 0u);
    for (int x = 0; x < 8; ++x) { int b = c * 8 + x; m[b >> 5] |= 1u << (b & 31); }
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, words, m.data()) != hipSuccess) { printf("c=%d: create failed\n", c); continue; }
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, s, d, 20000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::set<std::string> seen;
    for (int b = 0; b < nb; ++b) {
      unsigned hw = h[2 * b];
      char buf[64];
      snprintf(buf, sizeof buf, "se%u.sh%u.cu%u", (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
      seen.insert(buf);
    }
    std::set<unsigned> xccs;
    for (int b = 0; b < nb; ++b) xccs.insert(h[2 * b + 1] & 0xf);
    printf("logical cu %2d: xccs %zu ids:", c, xccs.size());
    for (auto& s2 : seen) printf(" %s", s2.c_str());
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
