// Internal declarations shared by the libg3hip translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "g3hip.h"
#include "g3_host.h"

#define G3_LB 128   // diagonal block factored + inverted by ONE fused kernel; matrices are padded to it

#define G3_PROG_SLOTS 8
struct g3_ctx {
  int device;
  hipStream_t stream;      // stream work is enqueued on
  hipStream_t own_stream;  // created by the context
  hipStream_t side_stream; // low-priority stream carrying the bulk trailing updates (look-ahead)
  hipStream_t side_for;    // the stream the side stream's placement was probed against (g3i_ensure_side_stream)
  hipEvent_t* la_ev;       // look-ahead events (2 per panel)
  int la_nev;
  int64_t nb_lookahead;    // panel width of the flat right-looking sweep (0 = default)
  G3hTune tune;            // tuning knobs, read from the environment once at g3_ctx_create
  unsigned long long gram_paths[3];   // Gram launches so far: compile-time table, generated at first use, interpreted
  unsigned long long grad_paths[3];   // the same for the gradient's kernel-parameter sums
  FILE* gemm_log;          // G3_GEMM_LOG=<file>: one line per MFMA GEMM / stripe-solve launch (scripts/launch_table.py)
  bool info_clean;         // d_info is known to be zero: left so by a synchronised evaluation (info_stream == nullptr and
                           // info_sync), or cleared on info_stream with no factorisation queued since (g3i_reset_info)
  bool info_sync;
  hipStream_t info_stream;
  bool fuse256;            // factor 256-wide diagonal blocks with the one-launch kernel (chain-bound sizes)
  bool adopted;            // stream belongs to the caller
  bool bulk_role;          // this context's stream carries bulk updates beside another context's chain (multi-GPU driver)
#ifdef G3_CHAIN_SERVER   // measurement variant only (scripts/variants/chain_server.inc): the chain of a sweep on resident workgroups
  unsigned* chain_ctl;     // device: counters and per-panel flags of the running server
  hipStream_t chain_sA, chain_sB;            // the chain sweep's own chain / bulk streams (created with the two below)
  hipStream_t chain_stream, chain_stream2;   // streams of the server's two kernels (diagonal workgroup, workers)
  hipEvent_t chain_ev, chain_ev2, chain_ev3; // bracket of the caller's stream, end of the two server kernels
  int chain_wgs;           // workgroups of the server (< 2: off)
  int chain_lds;           // LDS bytes a server workgroup asks for (0: what it needs)
  int64_t chain_min_n, chain_max_n;   // matrices the server is used for
  bool chain_broken;       // a server gave up (wall-clock limit): launches per kernel from then on
#endif
  // batch mode (g3_gp_factor_batched): every MFMA GEMM and diagonal-block launch of a sweep acts on
  // `batch` matrices at once (grid.y); operands inside the block-inverse buffer [bw_base, +bw_bytes)
  // are `bstride_w` elements apart, everything else `bstride` elements
  int batch;
  int64_t bstride, bstride_w;
  const char* bw_base;
  size_t bw_bytes;
  int64_t gram_diag_off;   // row offset of the block g3_gram_rows is building (0 otherwise)
  void* bbuf;              // device scratch of the batched entry points (programs, statistics)
  size_t bbuf_bytes;
  // small device scratch
  int* d_info;             // potrf info flag (one per batch member, G3_MAX_BATCH of them)
  double* d_stats;         // 64 doubles of reduction outputs
  g3_kernel_prog* d_prog;  // kernel programs (device copies, a ring of G3_PROG_SLOTS)
  hipEvent_t prog_ev[G3_PROG_SLOTS];   // recorded behind the launch that read the slot
  bool prog_busy[G3_PROG_SLOTS];
  int prog_next, prog_last;
  hipStream_t prog_stream;   // stream the last slot was uploaded / consumed on
  // pinned host mirrors
  int* h_info;
  double* h_stats;
  g3_kernel_prog* h_prog;
  // block inverses of the last factorisation
  void* invd;
  size_t invd_bytes;
  // padded workspace for g3_potrf_robust / g3_trsm on ragged sizes
  void* work;
  size_t work_bytes;
  // optional profiling: HIP-event pairs around tagged regions (g3_prof_*)
  bool prof_on;
  int prof_level;
  unsigned prof_skip;      // sampling counter of level 3
  hipEvent_t* prof_ev;
  int prof_cap, prof_n;          // events allocated / used
  struct { int e0, e1, tag; double work; }* prof_rec;
  int prof_nrec;
  char err[512];
};

// profiling tags
enum { G3_TAG_GEMM_BIG = 0, G3_TAG_GRAM = 1, G3_TAG_POTRF = 2, G3_TAG_TRSV = 3, G3_TAG_CROSS_GRAM = 4,
       G3_TAG_TRSM = 5, G3_TAG_REDUCE = 6, G3_TAG_GEMM_MID = 7, G3_TAG_GEMM_SMALL = 8, G3_TAG_LEAF = 9,
       G3_NTAGS = 10 };
int g3i_prof_begin(g3_ctx* ctx, int tag, double work);   // returns record index or -1
void g3i_prof_end(g3_ctx* ctx, int rec);

#define G3_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      snprintf(ctx->err, sizeof(ctx->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call,       \
               hipGetErrorString(_e));                                                        \
      return G3_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

#define G3_LAUNCH_CHECK()                                                                     \
  do {                                                                                        \
    hipError_t _e = hipGetLastError();                                                        \
    if (_e != hipSuccess) {                                                                   \
      snprintf(ctx->err, sizeof(ctx->err), "%s:%d launch -> %s", __FILE__, __LINE__,          \
               hipGetErrorString(_e));                                                        \
      return G3_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

// Every extern "C" entry runs with the context's device current and restores the caller's device
// on return, so contexts on different GPUs can be used from one process (one thread per context).
struct g3_dev_guard {
  int prev = -1;
  bool switched = false;
  explicit g3_dev_guard(const g3_ctx* ctx) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != ctx->device) switched = (hipSetDevice(ctx->device) == hipSuccess);
  }
  ~g3_dev_guard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
  g3_dev_guard(const g3_dev_guard&) = delete;
  g3_dev_guard& operator=(const g3_dev_guard&) = delete;
};
#define G3_MAX_DEVICES 64   // per-device caches of function attributes

#define G3_MAX_BATCH 4096
static inline int g3_nbatch(const g3_ctx* ctx) { return ctx->batch > 1 ? ctx->batch : 1; }
// element stride between batch members for an operand at p (0 outside batch mode)
static inline int64_t g3_bstride_of(const g3_ctx* ctx, const void* p) {
  if (ctx->batch <= 1) return 0;
  const char* c = (const char*)p;
  return (ctx->bw_base && c >= ctx->bw_base && c < ctx->bw_base + ctx->bw_bytes) ? ctx->bstride_w : ctx->bstride;
}

// launches issued now go to the low-priority bulk stream of the look-ahead sweep
static inline bool g3_on_bulk_stream(const g3_ctx* ctx) {
#ifdef G3_CHAIN_SERVER
  if (ctx->chain_sB && ctx->stream == ctx->chain_sB) return true;
#endif
  return ctx->stream == ctx->side_stream;
}

static inline size_t g3_esize(g3_dtype dt) { return dt == G3_F64 ? 8 : 4; }
static inline int64_t g3_roundup(int64_t n, int64_t m) { return (n + m - 1) / m * m; }

// a context on the caller's stream that creates no stream of its own; the low-priority side stream on first need
int g3i_ctx_create_on(int device, hipStream_t stream, g3_ctx** out);
int g3i_ensure_side_stream(g3_ctx* ctx);
void g3i_ctx_forget_stream(g3_ctx* ctx, hipStream_t s);     // s (synchronised) is about to be destroyed by its owner
// latency (us) of a one-wave kernel submitted on `a` while a dispatch-bound launch (its duration in *long_us) runs on `b`
bool g3i_probe_pair(hipStream_t a, hipStream_t b, unsigned* scratch, double* tiny_us, double* long_us, int reps);

// internal launchers (stream-ordered, no host sync)
int g3i_gemm_nt(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                g3_dtype dt, int lower_only);
// wide = 1: force a tile that spans 128 output columns (C may then alias A when n == 128)
int g3i_gemm_nt_ex(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                   int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                   g3_dtype dt, int lower_only, int wide);
// C = alpha A V^T + beta C, V (n x n) lower triangular: column tile n0 reduces over K = n0 + BN only (C must not alias A)
int g3i_gemm_nt_ktri(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* V, int64_t ldv, int64_t m, int64_t n,
                     double alpha, double beta, g3_dtype dt);
// only elements with col <= row + diag_off (trapezoid; 0 = lower triangle of a diagonal-anchored C)
int g3i_gemm_nt_trap(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                     int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                     g3_dtype dt, int64_t diag_off);
// staircase: stacked row segments, segment s updates its first seg_cols[s] columns (one launch)
int g3i_gemm_nt_stair(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda, const void* B,
                      int64_t ldb, int64_t k, const int64_t* seg_rows, const int64_t* seg_cols, int nseg,
                      double alpha, double beta, g3_dtype dt, int64_t b_nb, const int32_t* b_perm, int nperm,
                      const int64_t* seg_diag = nullptr);
// one-launch stripe-local solve X <- X L^-T (n <= 1024); 1 = shape not covered, 0 = done
int g3i_trsm_stripe(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, void* X, int64_t m, int64_t ldx, const void* W,
                    g3_dtype dt);
int g3i_potrf(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, void* invd);
int g3i_potrf_tall(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, void* invd, int64_t E);
int g3i_trsm_rlt(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, void* B, int64_t m,
                 int64_t ldb, g3_dtype dt, const void* invd);
int g3i_trtri_blocks(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, g3_dtype dt, void* invd);
// V = L^-1 (n x n lower, n = 128 * 2^q <= 2048) from the 128-block inverses W; Vt and U are n x n scratch (Vt = V^T on return
// except for the last level); all compact (leading dimension n)
int g3i_trtri_full(g3_ctx* ctx, const void* L, int64_t n, const void* W, void* V, void* Vt, void* U, g3_dtype dt);
int g3i_reset_info(g3_ctx* ctx);
bool g3i_info_known_zero(const g3_ctx* ctx);
// g3_diag_stats / g3_logp_terms / g3_rows_dot_ss results left in device memory (no host synchronisation): 3 / 4 doubles
int g3i_diag_stats_dev(g3_ctx* ctx, const void* A, int64_t n, int64_t ld, g3_dtype dt, double* out_dev);
int g3i_logp_terms_dev(g3_ctx* ctx, const void* L, int64_t n, int64_t ld, const void* a, g3_dtype dt, double* out_dev);
// batch members of at most 256 padded rows: factor + block inverses + a = L^-1 delta + the four logp scalars, one
// workgroup per member, one launch (g3_potrf.hip)
int g3i_small_factor_batched(g3_ctx* ctx, void* K, int64_t ld, int64_t kstride, void* W, int64_t wstride, const void* delta, int64_t ldd,
                             void* a, int64_t astride, double* dstats, int batch, int64_t n, int64_t np, g3_dtype dt);
// batch members of 384 ... 1024 padded rows: a group of workgroups per member factors it (and solves its right-hand-side rows)
// in ONE launch for the whole batch (g3_chainb.hip); ctl: g3i_coop_ctl_bytes(batch) of device scratch
size_t g3i_coop_ctl_bytes(int batch);
int g3i_coop_group(const g3_ctx* ctx, int batch, int64_t np);
int g3i_coop_factor_batched(g3_ctx* ctx, void* K, int64_t ld, int64_t kstride, void* W, int64_t wstride, unsigned* ctl, int batch,
                            int64_t np, g3_dtype dt);
#ifdef G3_CHAIN_SERVER
// true when `info` says the chain server gave up: it is switched off for this context (one line on stderr)
bool g3i_chain_gave_up(g3_ctx* ctx, int info);
#define G3_INFO_CHAIN 0x40000000   // pivot-flag value: the chain server gave up (never a pivot index)
#endif
int g3i_diag_add(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, double value);
// A[0:rows, 0:cols) *= factor (stream-ordered)
int g3i_scale(g3_ctx* ctx, void* A, int64_t rows, int64_t cols, int64_t ld, g3_dtype dt, double factor);
int g3i_gram_grad(g3_ctx* ctx, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X, int64_t N, int64_t ldx,
                  int d, g3_dtype dt, const void* G, int64_t ldg, const void* alpha, double* out_host,
                  int64_t row0 = 0, int64_t nrows = -1);
// generated gradient kernel (g3_gram_jit.hip); at most this many register accumulators per thread
#define G3_GRAD_JIT_MAXSLOTS 40
hipFunction_t g3i_grad_jit_function(g3_ctx* ctx, const g3_kernel_prog* prog_host, int d, g3_dtype dt, int* nslots);
int g3i_gram_grad_batched(g3_ctx* ctx, const g3_kernel_prog* progs, int batch, const g3_grad_map* map, const void* X, int64_t N,
                          int64_t ldx, int d, g3_dtype dt, const void* G, int64_t ldg, int64_t gstride, const void* alpha,
                          int64_t astride, double* out_host, int64_t row0 = 0, int64_t nrows = -1);
int g3i_rows_dot_ss_batched(g3_ctx* ctx, const void* V, int64_t m, int64_t n, int64_t ld, const void* a, g3_dtype dt, void* dot,
                            void* ss, int batch, int64_t vstride, int64_t astride, int64_t ostride);
int g3i_ensure_invd(g3_ctx* ctx, int64_t n, g3_dtype dt);
int g3i_ensure_work(g3_ctx* ctx, size_t bytes);
int g3i_upload_prog(g3_ctx* ctx, const g3_kernel_prog* prog, int slot, const g3_kernel_prog** dptr);
int g3i_gram_batched(g3_ctx* ctx, const g3_kernel_prog* dprogs, const g3_kernel_prog* first_host, int batch,
                     const void* X, int64_t n, int64_t ldx, int d, g3_dtype dt, void* K, int64_t ldk,
                     int64_t kstride, int64_t npad, unsigned flags);
int g3i_validate_prog(const g3_kernel_prog* p, int d);
// the Gram kernel generated for prog's structure (g3_gram_jit.hip); 0 = launched, 1 = none (the caller interprets)
int g3i_gram_jit(g3_ctx* ctx, const g3_kernel_prog* prog_host, const g3_kernel_prog* prog_dev, int batch, const void* X1, int64_t n1,
                 int64_t ldx1, const void* X2, int64_t n2, int64_t ldx2, int d, g3_dtype dt, void* K, int64_t ldk, int64_t n1pad,
                 int64_t n2pad, unsigned flags, int sym, int64_t kstride, int64_t diag_off, dim3 grid);
int g3i_potri(g3_ctx* ctx, const void* L, int64_t n, int64_t ldl, const void* invd, g3_dtype dt, void* Y,
              int64_t ldy, void* C, int64_t ldc);
