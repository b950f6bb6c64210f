// The MFMA "NT" tile routine of libg3hip: one output tile C[m0.., n0..] = alpha * A_rows * B_rows^T + beta * C as a
// device function.  Included by g3_gemm.hip (one tile per workgroup, the stripe-local solve) and by g3_potrf.hip (the
// resident chain workgroups run sequences of tiles).  See g3_gemm.hip for the CDNA4 mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "g3_mfma.h"

constexpr int ROWB = 128;  // bytes of K per LDS row and per stage
// LDS buffers per tile configuration.  Two everywhere: deeper pipelines (four buffers) were measured on
// the small tiles and LOST -- 64 / 80 KiB of LDS per workgroup no longer fits into the slot a retiring
// bulk workgroup leaves behind, so the critical-path launches wait for several to retire at once
// (N = 16384: 36.2 -> 38.1 ms; 8192: 8.8 -> 9.2 ms), while the uncontended latency of a tiny product
// barely moves (12.0 -> 11.6 us).  The kernel keeps the general NSTAGE loop.
constexpr int STAGES = 2;

// One output tile: C[m0.., n0..] = alpha * A_rows * B_rows^T + beta * C, with gA / gB the first row of the
// tile's A / B operand (k contiguous).  Shared by the GEMM kernel (one tile per workgroup) and the
// stripe-local triangular solve (a workgroup runs a sequence of tiles on its own rows).  Ends with the
// workgroup synchronised and the staging LDS free again.
template <typename T, int BM, int BN, int WM, int WN, int NSTAGE>
__device__ __forceinline__ void gemm_tile(T* C, int64_t ldc, const T* gA, int64_t lda, const T* gB, int64_t ldb,
                                          int K, T alpha, T beta, int m0, int n0, bool lower_only, int doff,
                                          int failed, char* smem) {
  using M = MfmaT<T>;
  using chunk_t = typename M::chunk_t;
  using acc_t = typename M::acc_t;
  constexpr int EPC = M::EPC;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int NWN = BN / WN;
  constexpr int NT = (BM / WM) * NWN * 64;
  constexpr int TM = WM / 16, TN = WN / 16;
  char* sA = smem;                           // [NSTAGE][BM][ROWB]
  char* sB = smem + NSTAGE * BM * ROWB;      // [NSTAGE][BN][ROWB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;

  acc_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  // ---- staging: LDS-DMA (global_load_lds_dwordx4).  One wave-instruction moves 64 x 16 B =
  // 8 rows x 128 B straight from global memory into LDS (no VGPR round trip, no ds_write).
  // The LDS destination is linear (wave-uniform base + lane * 16), so the XOR swizzle is
  // applied to the per-lane SOURCE address: physical chunk c of row r receives logical chunk
  // c ^ ((r >> 1) & 7), which is what the fragment reads below expect.
  constexpr int NW = NT / 64;
  static_assert(((BM / 8) % NW == 0 || BM / 8 < NW) && (BN / 8) % NW == 0, "rows per wave-instruction");
  const int sr = lane >> 3, sc = lane & 7;
  auto stage = [&](int buf, int k0) {
    char* a = sA + buf * BM * ROWB;
    char* b = sB + buf * BN * ROWB;
#pragma unroll
    for (int i = 0; i < (BM / 8 + NW - 1) / NW; ++i) {
      if (BM / 8 < NW && wave >= BM / 8) break;      // a tile with fewer 8-row groups than waves (16-row stripes)
      const int rb = (i * NW + wave) * 8, row = rb + sr;
      const T* src = gA + (int64_t)row * lda + k0 + ((sc ^ ((row >> 1) & 7)) * EPC);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(a + rb * ROWB), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BN / 8 / NW; ++i) {
      const int rb = (i * NW + wave) * 8, row = rb + sr;
      const T* src = gB + (int64_t)row * ldb + k0 + ((sc ^ ((row >> 1) & 7)) * EPC);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(b + rb * ROWB), 16, 0, 0);
    }
  };

  const int frow = lane & 15, kq = lane >> 4, swz = (frow >> 1) & 7;
  // s = 0, 1: the two 8-deep halves of a K tile.  Fragment reads and MFMAs are separate steps so that the
  // K loop can put the reads of a freshly published tile in front of the DMA issue for the next one
  auto load_frags = [&](int buf, int s, chunk_t (&fa)[TM], chunk_t (&fb)[TN]) {
    const char* a = sA + buf * BM * ROWB + (wm + frow) * ROWB;
    const char* b = sB + buf * BN * ROWB + (wn + frow) * ROWB;
    const int off = (((s * 4 + kq) ^ swz) << 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const chunk_t*>(a + i * 16 * ROWB + off);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const chunk_t*>(b + j * 16 * ROWB + off);
  };
  auto mfma_frags = [&](const chunk_t (&fa)[TM], const chunk_t (&fb)[TN]) {
#pragma unroll
    for (int e = 0; e < EPC; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = M::mfma(fa[i][e], fb[j][e], acc[i][j]);
  };

  // NSTAGE LDS buffers, NSTAGE - 1 K tiles of DMA in flight.  The big tile is matrix-pipe bound
  // (64 MFMAs = 4096 cycles per K tile and wave, twice that with the partner workgroup): two
  // buffers hide a memory round trip completely.  (See STAGES above for why the small tiles stay
  // at two buffers as well.)
  const int KT = K / BK;
  constexpr int LPS = (BM / 8 + NW - 1) / NW + BN / 8 / NW;   // DMA instructions per wave and stage (upper bound)
  static_assert(NSTAGE == 2 || BM / 8 >= NW, "the vmcnt bookkeeping of deeper pipelines needs the same count on every wave");
  static_assert((NSTAGE - 2) * LPS <= 63, "vmcnt range");
  // epilogue geometry (see below).  (Requesting the first chunk of C before the last K tile would
  // hide its latency too, but costs 25 VGPRs: the kernel must stay <= 224 so that the
  // critical-path diagonal kernel still fits beside one of these workgroups.)
  constexpr int RC = BM < 32 ? BM : 32;                    // rows per chunk
  constexpr int PITCH = BN * (int)sizeof(T) + 128;         // +128 B: rows r, r+1 land in different bank halves
  constexpr int VPR = BN * (int)sizeof(T) / 16;            // 16-byte vectors per row
  constexpr int RPP = NT / VPR;                            // rows per pass of the whole workgroup
  constexpr int NPASS = RC / RPP;
  constexpr int NCH = BM / RC;
  static_assert(RC * PITCH <= NSTAGE * (BM + BN) * ROWB, "epilogue chunk must fit in the staging LDS");
  static_assert(NT % VPR == 0 && RC % RPP == 0 && BM % RC == 0, "epilogue tiling");
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (ldc % EPC == 0);
  const bool rmw = vec_ok && beta != T(0);
  const int vrow = tid / VPR, vcol = (tid % VPR) * EPC;    // this thread's row (within a pass) and first column
  chunk_t cnext[NPASS];
  auto fetch_c = [&](int c) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int row = m0 + c * RC + p * RPP + vrow, col = n0 + vcol;
      cnext[p] = chunk_t{};
      if (!(lower_only && col > row + doff)) cnext[p] = *reinterpret_cast<const chunk_t*>(C + (int64_t)row * ldc + col);
    }
  };
  // small tiles have registers to spare: request the first chunk of C before anything else
  constexpr bool EARLY_C = NSTAGE > 2;
  if (EARLY_C && rmw) fetch_c(0);
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < KT) stage(s, s * BK);
  int buf = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // K tile kt has landed once at most the NSTAGE - 2 younger stages are outstanding (loads retire
    // in order); near the end fewer stages were issued, so wait for everything
    if (NSTAGE > 2 && kt + NSTAGE - 2 < KT) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LPS) : "memory");   // (the C prefetch is older: covered)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();          // tile kt is published; everyone has finished tile kt - 1, whose buffer is refilled next
    const int nxt = kt + NSTAGE - 1;
    chunk_t fa[TM], fb[TN];
    // the first fragments are requested before the ~40 scalar / vector instructions that issue the next
    // tile's DMA: the LDS round trip of the one hides under the address arithmetic of the other (round 3:
    // +0.9 ... +4 % stand-alone against the opposite order, profiles/r03_gemm_variants.md)
    load_frags(buf, 0, fa, fb);
    if (nxt < KT) stage(buf == 0 ? NSTAGE - 1 : buf - 1, nxt * BK);
    mfma_frags(fa, fb);
    load_frags(buf, 1, fa, fb);
    mfma_frags(fa, fb);
    buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
  }
  __syncthreads();            // the epilogue re-uses the staging LDS
  if (failed != 0) return;    // uniform: every thread of every workgroup reads the same flag

  // ---- epilogue.  The accumulators are in MFMA layout (a lane holds 4 rows x 1 column of each
  // 16 x 16 tile): written straight to C that is 128-byte pieces scattered over 64 rows per
  // instruction, and with beta != 0 the same pattern is read first -- measured 6 % of a K = 1024
  // launch for the read alone.  Instead the tile goes through the (now idle) staging LDS in
  // 32-row chunks and every global access is a full row segment: 64 consecutive 16-byte vectors,
  // BN * sizeof(T) contiguous bytes per row (1 KiB for the 128-column fp64 tile).
  if (vec_ok) {
    const int col_l = lane & 15;
    if (rmw && !EARLY_C) fetch_c(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // this chunk's C vectors were requested one step ago; request the next chunk's now
      chunk_t cold[NPASS];
#pragma unroll
      for (int p = 0; p < NPASS; ++p) cold[p] = cnext[p];
      if (rmw && c + 1 < NCH) fetch_c(c + 1);
      // accumulators of the waves that own rows of this chunk -> LDS (row-major)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if ((wm + i * 16) / RC == c) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int rl = (wm + i * 16) % RC + M::row(lane, r);
              *reinterpret_cast<T*>(smem + rl * PITCH + (wn + j * 16 + col_l) * (int)sizeof(T)) = acc[i][j][r];
            }
        }
      }
      __syncthreads();
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int rl = p * RPP + vrow;
        const int row = m0 + c * RC + rl, col = n0 + vcol;
        if (lower_only && col > row + doff) continue;
        chunk_t v = *reinterpret_cast<const chunk_t*>(smem + rl * PITCH + vcol * (int)sizeof(T));
        v = v * alpha;
        if (rmw) v = v + cold[p] * beta;
        T* dst = C + (int64_t)row * ldc + col;
        if (!lower_only || col + EPC - 1 <= row + doff) {
          *reinterpret_cast<chunk_t*>(dst) = v;
        } else {                                           // the vector straddles the diagonal
#pragma unroll
          for (int e = 0; e < EPC; ++e)
            if (col + e <= row + doff) dst[e] = v[e];
        }
      }
      __syncthreads();
    }
    return;
  }
  // fallback (C not 16-byte aligned): element-wise from the MFMA layout; no LDS involved
  const int col_l = lane & 15;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + j * 16 + col_l;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + M::row(lane, r);
        if (lower_only && col > row + doff) continue;
        T* p = C + (int64_t)row * ldc + col;
        T v = alpha * acc[i][j][r];
        if (beta != T(0)) v += beta * (*p);
        *p = v;
      }
    }
}
