// Development probe: which physical CUs (XCC, SE, SH, CU) does a CU-masked stream dispatch to on gfx950?
// build: hipcc --offload-arch=gfx950 -O2 scripts/cumask_probe.hip -o /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <set>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char* name, hipStream_t s) {
  const int nb = 8192;
  unsigned* d; hipMalloc(&d, nb * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, s, d, 200000);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(nb * 2);
  hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<int, std::set<int>> per_xcc;
  for (int b = 0; b < nb; ++b) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc].insert(se * 100 + sh * 16 + cu);
  }
  int match = 0;
  for (int b = 0; b < nb; ++b) match += ((h[2 * b + 1] & 0xf) == (unsigned)(b & 7));
  printf("%-22s rr-match %5.1f%% ", name, 100.0 * match / nb);
  int tot = 0;
  for (auto& kv : per_xcc) { printf(" xcc%d:%2zu", kv.first, kv.second.size()); tot += kv.second.size(); }
  printf("  total %d\n", tot);
  hipFree(d);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
  printf("CUs %d\n", ncu);
  hipStream_t s0; hipStreamCreate(&s0); run("unmasked", s0);
  struct { const char* n; int mode, cnt; } tests[] = {{"clear first 8", 0, 8}, {"clear first 16", 0, 16}, {"clear first 32", 0, 32},
    {"clear stride32 x8", 1, 8}, {"clear bits 0..7 of w0-7 (64)", 2, 64}, {"only first 8 bits set", 3, 8}, {"only bits%8==0 set", 4, 0}};
  for (auto& t : tests) {
    std::vector<uint32_t> m(words, 0xFFFFFFFFu);
    if (t.mode == 0) for (int b = 0; b < t.cnt; ++b) m[b >> 5] &= ~(1u << (b & 31));
    if (t.mode == 1) for (int q = 0; q < t.cnt; ++q) { int b = q * 32; m[b >> 5] &= ~(1u << (b & 31)); }
    if (t.mode == 2) for (int w = 0; w < words; ++w) m[w] &= ~0xFFu;
    if (t.mode == 3) { for (auto& x : m) x = 0; m[0] = 0xFF; }
    if (t.mode == 4) { for (auto& x : m) x = 0x01010101u; }
    hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, words, m.data());
    if (e != hipSuccess) { printf("%s: create failed %s\n", t.n, hipGetErrorString(e)); continue; }
    run(t.n, s);
    hipStreamDestroy(s);
  }
  return 0;
}
