// Do the diagonal-block kernels of libg3hip slow each other down when they share a CU -- and the chip?  (round 5: the medium-N
// chain kernel's ablation found two fused 256-wide factorisations per CU 2.4x slower each; the look-ahead sweeps see the same
// kernels 1.4 - 2.7x slower beside bulk GEMM workgroups.)  The library's own device code (g3_diag.h), batch in grid.y,
// B = 1 ... 2048 members: time per launch, and what that means per CU.  Also: the same launch with a 64 KiB instruction
// footprint (diag128: 42 KB of code) against the 147 KB of potrf256.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ig3py_amd/csrc -Wno-pass-failed scripts/diag_corun.hip -o build/diag_corun
#include "g3_internal.h"
#include "g3_mfma.h"
#include "g3_diag.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

template <typename T>
__global__ void __launch_bounds__(512, 4) k_potrf256(T* A, int64_t ld, T* W, int* info, int64_t a_batch, int64_t w_batch) {
  info += blockIdx.y;
  A += (int64_t)blockIdx.y * a_batch;
  W += (int64_t)blockIdx.y * w_batch;
  __shared__ DiagLds<T> S;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = wv < 4 ? wv : 11 - wv;
  switch (w) {
    case 0: potrf256_wave<T, 0>(A, ld, W, info, 0, S, lane); break;
    case 1: potrf256_wave<T, 1>(A, ld, W, info, 0, S, lane); break;
    case 2: potrf256_wave<T, 2>(A, ld, W, info, 0, S, lane); break;
    case 3: potrf256_wave<T, 3>(A, ld, W, info, 0, S, lane); break;
    case 4: potrf256_wave<T, 4>(A, ld, W, info, 0, S, lane); break;
    case 5: potrf256_wave<T, 5>(A, ld, W, info, 0, S, lane); break;
    case 6: potrf256_wave<T, 6>(A, ld, W, info, 0, S, lane); break;
    default: potrf256_wave<T, 7>(A, ld, W, info, 0, S, lane); break;
  }
}
template <typename T>
__global__ void __launch_bounds__(512, 4) k_diag128(T* A, int64_t ld, T* W, int* info, int64_t a_batch, int64_t w_batch) {
  info += blockIdx.y;
  A += (int64_t)blockIdx.y * a_batch;
  W += (int64_t)blockIdx.y * w_batch;
  __shared__ DiagLds<T> S;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = wv < 4 ? wv : 11 - wv;
  switch (w) {
    case 0: diag128_wave<T, true, 0>(A, ld, W, G3_LB, info, 0, S, lane); break;
    case 1: diag128_wave<T, true, 1>(A, ld, W, G3_LB, info, 0, S, lane); break;
    case 2: diag128_wave<T, true, 2>(A, ld, W, G3_LB, info, 0, S, lane); break;
    case 3: diag128_wave<T, true, 3>(A, ld, W, G3_LB, info, 0, S, lane); break;
    case 4: diag128_wave<T, true, 4>(A, ld, W, G3_LB, info, 0, S, lane); break;
    case 5: diag128_wave<T, true, 5>(A, ld, W, G3_LB, info, 0, S, lane); break;
    case 6: diag128_wave<T, true, 6>(A, ld, W, G3_LB, info, 0, S, lane); break;
    default: diag128_wave<T, true, 7>(A, ld, W, G3_LB, info, 0, S, lane); break;
  }
}

// the SAME 128-block program run R times inside one launch (fresh data each time): the first pass fetches its 43 KB of
// instructions from memory, later passes find them in the instruction cache -- is instruction fetch part of the 29 us?
template <typename T>
__global__ void __launch_bounds__(512, 4) k_diag128_rep(T* A, int64_t ld, T* W, int* info, int64_t a_batch, int64_t w_batch, int reps, unsigned long long* ts) {
  __shared__ DiagLds<T> S;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = wv < 4 ? wv : 11 - wv;
#pragma nounroll
  for (int r = 0; r < reps; ++r) {
    __syncthreads();
    if (threadIdx.x == 0) ts[2 * r] = wall_clock64();
    T* Ar = A + (int64_t)r * a_batch;
    T* Wr = W + (int64_t)r * w_batch;
    switch (w) {
      case 0: diag128_wave<T, true, 0>(Ar, ld, Wr, G3_LB, info, 0, S, lane); break;
      case 1: diag128_wave<T, true, 1>(Ar, ld, Wr, G3_LB, info, 0, S, lane); break;
      case 2: diag128_wave<T, true, 2>(Ar, ld, Wr, G3_LB, info, 0, S, lane); break;
      case 3: diag128_wave<T, true, 3>(Ar, ld, Wr, G3_LB, info, 0, S, lane); break;
      case 4: diag128_wave<T, true, 4>(Ar, ld, Wr, G3_LB, info, 0, S, lane); break;
      case 5: diag128_wave<T, true, 5>(Ar, ld, Wr, G3_LB, info, 0, S, lane); break;
      case 6: diag128_wave<T, true, 6>(Ar, ld, Wr, G3_LB, info, 0, S, lane); break;
      default: diag128_wave<T, true, 7>(Ar, ld, Wr, G3_LB, info, 0, S, lane); break;
    }
    __syncthreads();
    if (threadIdx.x == 0) ts[2 * r + 1] = wall_clock64();
  }
}

int main() {
  const int n = 256, BMAX = 2048;
  std::vector<double> h((size_t)n * n);
  srand(3);
  std::vector<double> Bm((size_t)n * 64);
  for (auto& v : Bm) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = 0; k < 64; ++k) s += Bm[i * 64 + k] * Bm[j * 64 + k] / 64.0;
      h[(size_t)i * n + j] = s;
    }
  double *A0, *A, *W;
  int* info;
  hipMalloc(&A0, (size_t)n * n * 8);
  hipMalloc(&A, (size_t)BMAX * n * n * 8);
  hipMalloc(&W, (size_t)BMAX * 2 * 128 * 128 * 8);
  hipMalloc(&info, BMAX * sizeof(int));
  hipMemcpy(A0, h.data(), (size_t)n * n * 8, hipMemcpyHostToDevice);
  hipMemset(info, 0, BMAX * sizeof(int));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    printf("%s\n", which == 0 ? "potrf256 (147 KB of code, 8 wave programs)" : "diag128 (43 KB of code)");
    for (int B : {1, 8, 64, 128, 256, 384, 512, 768, 1024, 2048}) {
      double best = 1e30;
      for (int rep = 0; rep < 6; ++rep) {
        for (int b = 0; b < B; ++b) hipMemcpyAsync(A + (size_t)b * n * n, A0, (size_t)n * n * 8, hipMemcpyDeviceToDevice, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (which == 0)
          hipLaunchKernelGGL(k_potrf256<double>, dim3(1, B), dim3(512), 0, 0, A, (int64_t)n, W, info, (int64_t)n * n, (int64_t)2 * 128 * 128);
        else
          hipLaunchKernelGGL(k_diag128<double>, dim3(1, B), dim3(512), 0, 0, A, (int64_t)n, W, info, (int64_t)n * n, (int64_t)2 * 128 * 128);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      int hinfo = 0;
      hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost);
      printf("  B = %4d: %8.1f us per launch  (%.2f workgroups per CU; %.1f us per member and CU)  info %d\n", B, best * 1e3, B / 256.0,
             best * 1e3 / (B > 256 ? B / 256.0 : 1.0), hinfo);
    }
  }
  {
    const int R = 6;
    unsigned long long* ts;
    hipMalloc(&ts, 2 * R * sizeof(unsigned long long));
    for (int trial = 0; trial < 3; ++trial) {
      for (int b = 0; b < R; ++b) hipMemcpyAsync(A + (size_t)b * n * n, A0, (size_t)n * n * 8, hipMemcpyDeviceToDevice, 0);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(k_diag128_rep<double>, dim3(1), dim3(512), 0, 0, A, (int64_t)n, W, info, (int64_t)n * n, (int64_t)2 * 128 * 128, R, ts);
      hipDeviceSynchronize();
      unsigned long long h[2 * R];
      hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost);
      printf("diag128 program run %d times inside one launch (100 MHz clock), us per pass:", R);
      for (int r = 0; r < R; ++r) printf(" %.2f", (double)(h[2 * r + 1] - h[2 * r]) * 0.01);
      printf("\n");
    }
  }
  return 0;
}
