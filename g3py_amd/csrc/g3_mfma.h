// MFMA traits shared by the GEMM and the fused diagonal-block kernel.
#pragma once
#include <hip/hip_runtime.h>

template <typename T>
struct MfmaT;
template <>
struct MfmaT<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  typedef double chunk_t __attribute__((ext_vector_type(2)));
  static constexpr int EPC = 2;  // elements per 16-byte chunk
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct MfmaT<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  typedef float chunk_t __attribute__((ext_vector_type(4)));
  static constexpr int EPC = 4;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

