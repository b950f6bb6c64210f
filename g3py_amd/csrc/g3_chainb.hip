// Chains of MEDIUM problems (256 < N <= 1024): a small GROUP of workgroups factors one member of a batch, all members in
// ONE launch.  The reference's callers evaluate a chain one hyper-parameter vector at a time
// (g3py/processes/stochastic.py:515-531 "TODO: Vectorized", :740-771); each evaluation is CholeskyRobust's dpotrf
// (g3py/libs/tensors.py:197-222) plus the triangular solve of gaussian.py:212 on an N x N covariance.
//
// Round 2 pushed such a batch through the large-N launch sequence with the batch in grid.y: at N = 512 that is ~25 dependent
// launches per sweep, each a few microseconds of work per member -- 243 k evaluations / s, 0.14 of the FP64 matrix peak -- and
// no knob of the sweep moved it (profiles/r04_midchain.txt).  Here the sweep of ONE member is a loop inside a workgroup
// group: right-looking over 256-wide steps,
//   D  workgroup 0 of the group   the fused 256-wide factorisation of the diagonal block (g3_diag.h::potrf256_wave)
//   T  all G workgroups           rows below it:  X <- X L_jj^-T  in 64-row stripes through the block inverses
//   S  all G workgroups           trailing update  A22 -= X X^T  in 128 x 128 tiles, the next diagonal block first
// and the appended right-hand-side rows (delta in row 0 of a 16-row block) ride along, so a = L^-1 delta comes out of the
// same loop.  G = 1 for long chains (4096 members keep every CU busy with whole members: no synchronisation at all),
// G up to 8 when the batch is short (96 members x 5 workgroups).  Group members find each other through per-member counters
// in device memory with the agent-scope release / acquire recipe of MI355X_MICROARCH.md ("inter-workgroup visibility");
// member and role come from a ticket taken when a workgroup STARTS, one ticket counter per XCD so that a group shares an
// L2, which also makes the waits deadlock-free whatever fits on the chip at once: a workgroup only ever waits for
// workgroups that took earlier or adjacent tickets.  Every wait has a wall-clock limit; a group that runs into it marks its
// member failed, and the host re-evaluates that member alone (the path a failed pivot takes anyway).
#include "g3_internal.h"
#include "g3_host.h"
#include "g3_mfma.h"
#include "g3_gemm_tile.h"
#include "g3_diag.h"

namespace {

constexpr unsigned long long CB_TIMEOUT = 400000000ull;   // 4 s of the 100 MHz wall clock without progress
enum { CB_ABORT = 8, CB_HDR = 32, CB_PER = 32 };           // words: 8 ticket counters, abort flag; then 128 B per member
enum { CB_DIAG = 0, CB_T = 1, CB_S = 2, CB_SD = 3 };       // a member's counters
#define G3_INFO_COOP 0x40000002                            // pivot-flag value: the group gave up (never a pivot index)

__device__ __forceinline__ unsigned cb_load(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// whole workgroup: wait until *p >= target, then acquire.  `dead` is a sticky LDS word.
__device__ __forceinline__ bool cb_wait(unsigned* ctl, int* info, const unsigned* p, unsigned target, int* dead) {
  if (threadIdx.x == 0) {
    if (cb_load(p) < target) {
      const unsigned long long t0 = wall_clock64();
      unsigned spins = 0;
      while (cb_load(p) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 31u) == 0) {
          if (cb_load(ctl + CB_ABORT) != 0) { *dead = 1; break; }
          if (wall_clock64() - t0 > CB_TIMEOUT) {
            __hip_atomic_store(ctl + CB_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicCAS(info, 0, G3_INFO_COOP);
            *dead = 1;
            break;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return *dead == 0;
}
// whole workgroup: everything it has stored becomes visible to the agent, then *p += 1 (add) or *p = v
__device__ __forceinline__ void cb_signal(unsigned* p, bool add, unsigned v) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (add) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// one BM x 128 tile of a stripe: C = alpha A B^T + beta C (C may alias A: the tile spans the 128 output columns)
// (eight waves: 16- and 32-row stripes as 1 x 8 waves of BM x 16, 64-row stripes as 2 x 4 waves of 32 x 32)
template <typename T, int BM>
__device__ __noinline__ void cb_stripe_tile(T* C, const T* gA, const T* gB, int64_t ld, int64_t ldb, int K, T alpha, T beta, char* smem) {
  gemm_tile<T, BM, 128, (BM > 32 ? 32 : BM), (BM > 32 ? 32 : 16), 2>(C, ld, gA, ld, gB, ldb, K, alpha, beta, 0, 0, false, 0, 0, smem);
  __syncthreads();
}
// one TS_ x TS_ tile of the lower-triangular update C[m0.., n0..] -= A_rows B_rows^T.  The members of a long chain stream
// their operands from HBM (4096 members x 2 MB do not fit any cache): 128 x 128 tiles read each operand byte half as often
// as 64 x 64 ones
constexpr int CB_TS = 128;
template <typename T>
__device__ __noinline__ void cb_syrk_tile(T* C, const T* gA, const T* gB, int64_t ld, int K, int m0, int n0, char* smem) {
  gemm_tile<T, CB_TS, CB_TS, CB_TS / 2, CB_TS / 4, 2>(C, ld, gA, ld, gB, ld, K, T(-1), T(1), m0, n0, true, 0, 0, smem);
}
// X <- X L^-T for one BM-row stripe X against the nblk 128-column blocks of L (inverses of its diagonal blocks in Winv):
// X_b -= X_{<b} L_{b,<b}^T, then X_b <- X_b W_b^T
template <typename T, int BM>
__device__ __forceinline__ void cb_solve_stripe(T* X, int64_t ld, const T* L, const T* Winv, int nblk, char* smem) {
  for (int b = 0; b < nblk; ++b) {
    if (b > 0) cb_stripe_tile<T, BM>(X + b * G3_LB, X, L + (int64_t)b * G3_LB * ld, ld, ld, b * G3_LB, T(-1), T(1), smem);
    cb_stripe_tile<T, BM>(X + b * G3_LB, X + b * G3_LB, Winv + (int64_t)b * G3_LB * G3_LB, ld, (int64_t)G3_LB, G3_LB, T(1), T(0), smem);
  }
}

// the diagonal step: the fused 256-wide factorisation, or the 128-wide one for the last step of an odd block count
// (not inlined: inside the step loop the compiler hoists the factorisation's per-lane index arithmetic out of the loop and keeps
//  it live across the tile phases -- 354 spilled registers at the 128 the kernel must stay within; as a function of its own it
//  spills what the stand-alone potrf256_kernel spills)
template <typename T>
__device__ __noinline__ void cb_diag_step(T* Ajj, int64_t ld, T* Wj, int* info, int64_t rb, int wj, DiagLds<T>& S) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = wv < 4 ? wv : 11 - wv;       // block rows dealt 0 1 2 3 | 7 6 5 4 (g3_potrf.hip::potrf256_kernel)
  if (wj == 2 * G3_LB) {
    switch (w) {
      case 0: potrf256_wave<T, 0>(Ajj, ld, Wj, info, rb, S, lane); break;
      case 1: potrf256_wave<T, 1>(Ajj, ld, Wj, info, rb, S, lane); break;
      case 2: potrf256_wave<T, 2>(Ajj, ld, Wj, info, rb, S, lane); break;
      case 3: potrf256_wave<T, 3>(Ajj, ld, Wj, info, rb, S, lane); break;
      case 4: potrf256_wave<T, 4>(Ajj, ld, Wj, info, rb, S, lane); break;
      case 5: potrf256_wave<T, 5>(Ajj, ld, Wj, info, rb, S, lane); break;
      case 6: potrf256_wave<T, 6>(Ajj, ld, Wj, info, rb, S, lane); break;
      default: potrf256_wave<T, 7>(Ajj, ld, Wj, info, rb, S, lane); break;
    }
  } else {
    switch (w) {
      case 0: diag128_wave<T, true, 0>(Ajj, ld, Wj, (int64_t)G3_LB, info, rb, S, lane); break;
      case 1: diag128_wave<T, true, 1>(Ajj, ld, Wj, (int64_t)G3_LB, info, rb, S, lane); break;
      case 2: diag128_wave<T, true, 2>(Ajj, ld, Wj, (int64_t)G3_LB, info, rb, S, lane); break;
      case 3: diag128_wave<T, true, 3>(Ajj, ld, Wj, (int64_t)G3_LB, info, rb, S, lane); break;
      case 4: diag128_wave<T, true, 4>(Ajj, ld, Wj, (int64_t)G3_LB, info, rb, S, lane); break;
      case 5: diag128_wave<T, true, 5>(Ajj, ld, Wj, (int64_t)G3_LB, info, rb, S, lane); break;
      case 6: diag128_wave<T, true, 6>(Ajj, ld, Wj, (int64_t)G3_LB, info, rb, S, lane); break;
      default: diag128_wave<T, true, 7>(Ajj, ld, Wj, (int64_t)G3_LB, info, rb, S, lane); break;
    }
  }
}

constexpr int CB_RHS = 16;      // rows of the right-hand-side block that take part (row 0 carries delta, the rest is zero)
constexpr int CB_SR = 64;       // rows per stripe of the panel solve (every stripe re-reads the diagonal block and its inverses)

#ifndef G3_COOP_WAVES
#define G3_COOP_WAVES 4         // waves per SIMD = two workgroups per CU (registers: 128 per lane, VGPRs and AGPRs together; LDS:
#endif                          // 2 x 78 KB): a group's phases are latency-bound tile sequences, a second resident workgroup fills the gaps
template <typename T>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(G3_COOP_WAVES, G3_COOP_WAVES)))
coop_factor_kernel(T* K, int64_t ld, int64_t kstride, T* Wall, int64_t wstride, int* info_all, unsigned* ctl, int np, int G, int batch) {
  extern __shared__ __attribute__((aligned(16))) char cb_smem[];
  __shared__ int dead;
  __shared__ unsigned tk;
  // member and role: a ticket per XCD (workgroup ids are dealt round-robin over the 8 XCDs), taken when the workgroup starts
  const unsigned xcd = blockIdx.x & 7u;
  if (threadIdx.x == 0) {
    dead = 0;
    tk = __hip_atomic_fetch_add(ctl + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int m = (int)(xcd + 8u * (tk / (unsigned)G)), me = (int)(tk % (unsigned)G);
  if (m >= batch) return;
  T* const A = K + (int64_t)m * kstride;
  T* const W = Wall + (int64_t)m * wstride;
  int* const info = info_all + m;
  unsigned* const c = ctl + CB_HDR + (int64_t)m * CB_PER;
  DiagLds<T>& S = *reinterpret_cast<DiagLds<T>*>(cb_smem);
  const bool coop = G > 1;                    // a group of one needs no counters and no fences
  const int nsteps = (np + 2 * G3_LB - 1) / (2 * G3_LB);
  auto failed_now = [&]() { return __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; };
  for (int j = 0; j < nsteps; ++j) {
    const int cj = 2 * G3_LB * j, wj = (np - cj < 2 * G3_LB) ? np - cj : 2 * G3_LB;
    // ---- D: the diagonal block (workgroup 0 of the group); it carries the updates of all earlier steps once every
    // workgroup has finished its share of the previous step's diagonal-block tiles
    if (me == 0) {
      if (coop && j > 0 && !cb_wait(ctl, info, c + CB_SD, (unsigned)G * (unsigned)j, &dead)) return;
#ifndef G3_COOP_SKIP_D     // (measurement builds: phase costs)
      if (!failed_now())
#else
      if (false)
#endif
        cb_diag_step<T>(A + (int64_t)cj * ld + cj, ld, W + (int64_t)(cj / G3_LB) * G3_LB * G3_LB, info, (int64_t)cj, wj, S);
      if (coop) cb_signal(c + CB_DIAG, false, (unsigned)(j + 1));
      else { __threadfence_block(); __syncthreads(); }
    }
    if (coop && !cb_wait(ctl, info, c + CB_DIAG, (unsigned)(j + 1), &dead)) return;
    const int r0 = cj + wj;                   // first row below the step's diagonal block
#ifndef G3_COOP_SKIP_TS
    const bool failed = failed_now();
#else
    const bool failed = true;
#endif
    // ---- T: X <- X L_jj^-T on the rows [r0, R): 32-row stripes of the square part, then the 16 right-hand-side rows
    const T* Ljj = A + (int64_t)cj * ld + cj;
    const T* Wj = W + (int64_t)(cj / G3_LB) * G3_LB * G3_LB;
    const int nstripe = (np - r0) / CB_SR;
    if (!failed) {
      for (int t = me; t < nstripe; t += G)
        cb_solve_stripe<T, CB_SR>(A + (int64_t)(r0 + CB_SR * t) * ld + cj, ld, Ljj, Wj, wj / G3_LB, cb_smem);
      if (nstripe % G == me) cb_solve_stripe<T, CB_RHS>(A + (int64_t)np * ld + cj, ld, Ljj, Wj, wj / G3_LB, cb_smem);
    }
    if (r0 >= np) break;                      // the last step: only the right-hand-side rows were left
    if (coop) {
      cb_signal(c + CB_T, true, 1u);
      if (!cb_wait(ctl, info, c + CB_T, (unsigned)G * (unsigned)(j + 1), &dead)) return;
    } else {
      __threadfence_block();
      __syncthreads();
    }
    // ---- S: A[r0 : R, r0 : np] -= X X^T with X = A[r0 : R, cj : cj + wj]: lower triangle of the square part in 64 x 64
    // tiles -- the next step's diagonal block first, so that workgroup 0 can go on -- then the right-hand-side rows
    const int rows_in = np - r0;
    const int wn = rows_in < 2 * G3_LB ? rows_in : 2 * G3_LB;
    T* C = A + (int64_t)r0 * ld + r0;
    const T* X = A + (int64_t)r0 * ld + cj;
    int id = 0;
    if (!failed)
      for (int ti = 0; ti < wn / CB_TS; ++ti)
        for (int tj = 0; tj <= ti; ++tj, ++id)
          if (id % G == me) cb_syrk_tile<T>(C, X + (int64_t)ti * CB_TS * ld, X + (int64_t)tj * CB_TS * ld, ld, wj, ti * CB_TS, tj * CB_TS, cb_smem);
    if (coop) cb_signal(c + CB_SD, true, 1u);
    if (!failed) {
      for (int ti = wn / CB_TS; ti < rows_in / CB_TS; ++ti)
        for (int tj = 0; tj <= ti; ++tj, ++id)
          if (id % G == me) cb_syrk_tile<T>(C, X + (int64_t)ti * CB_TS * ld, X + (int64_t)tj * CB_TS * ld, ld, wj, ti * CB_TS, tj * CB_TS, cb_smem);
      const T* Xr = A + (int64_t)np * ld + cj;                 // the right-hand-side rows of the panel
      for (int q = 0; q < rows_in / G3_LB; ++q, ++id)
        if (id % G == me)
          cb_stripe_tile<T, CB_RHS>(A + (int64_t)np * ld + r0 + q * G3_LB, Xr, X + (int64_t)q * G3_LB * ld, ld, ld, wj, T(-1), T(1), cb_smem);
    }
    if (coop) {
      cb_signal(c + CB_S, true, 1u);
      if (!cb_wait(ctl, info, c + CB_S, (unsigned)G * (unsigned)(j + 1), &dead)) return;
    } else {
      __threadfence_block();
      __syncthreads();
    }
  }
}

}  // namespace

size_t g3i_coop_ctl_bytes(int batch) { return ((size_t)CB_HDR + (size_t)batch * CB_PER) * sizeof(unsigned); }

// workgroups per member: fill the chip's ~256 single-workgroup slots when the batch is short, whole members when it is long
int g3i_coop_group(const g3_ctx* ctx, int batch, int64_t np) {
  if (ctx->tune.coop_group > 0) return ctx->tune.coop_group > 8 ? 8 : ctx->tune.coop_group;
  int g = (int)(480 / (batch > 0 ? batch : 1));      // ~512 workgroup slots (two per CU)
  const int gmax = (int)(np / 128);          // no more workgroups than 128-row slabs of work in the first step
  if (g > gmax) g = gmax;
  return g < 1 ? 1 : (g > 8 ? 8 : g);
}

// members of np = 384 ... 1024 padded rows (np % 128 == 0): every member's matrix K_b (lower, identity padded, the
// right-hand-side block behind it with delta in its row 0) is factored in place, the block inverses go to W_b, and the
// right-hand-side rows become [a_b; 0].  ctl: g3i_coop_ctl_bytes(batch) of device scratch.  Pivot failures (and groups that
// gave up) are left in ctx->d_info[b].
int g3i_coop_factor_batched(g3_ctx* ctx, void* K, int64_t ld, int64_t kstride, void* W, int64_t wstride, unsigned* ctl, int batch,
                            int64_t np, g3_dtype dt) {
  if (np % G3_LB || np < 2 * G3_LB || np > 2048 || batch < 1 || batch > G3_MAX_BATCH) return -1;
  const int G = g3i_coop_group(ctx, batch, np);
  G3_HIP(hipMemsetAsync(ctl, 0, g3i_coop_ctl_bytes(batch), ctx->stream));
  G3_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * batch, ctx->stream));
  ctx->info_clean = false;
  const unsigned grid = 8u * (unsigned)G * (unsigned)((batch + 7) / 8);
  static bool attr_set[G3_MAX_DEVICES][2] = {};
  const int dev_slot = ctx->device & (G3_MAX_DEVICES - 1);
  if (dt == G3_F64) {
    constexpr int LDS = (int)sizeof(DiagLds<double>) > 65536 ? (int)sizeof(DiagLds<double>) : 65536;
    if (!attr_set[dev_slot][0]) {
      G3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(coop_factor_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      attr_set[dev_slot][0] = true;
    }
    hipLaunchKernelGGL((coop_factor_kernel<double>), dim3(grid), dim3(512), LDS, ctx->stream, (double*)K, ld, kstride, (double*)W, wstride,
                       ctx->d_info, ctl, (int)np, G, batch);
  } else {
    constexpr int LDS = (int)sizeof(DiagLds<float>) > 65536 ? (int)sizeof(DiagLds<float>) : 65536;
    if (!attr_set[dev_slot][1]) {
      G3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(coop_factor_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      attr_set[dev_slot][1] = true;
    }
    hipLaunchKernelGGL((coop_factor_kernel<float>), dim3(grid), dim3(512), LDS, ctx->stream, (float*)K, ld, kstride, (float*)W, wstride,
                       ctx->d_info, ctl, (int)np, G, batch);
  }
  G3_LAUNCH_CHECK();
  return G3_OK;
}
