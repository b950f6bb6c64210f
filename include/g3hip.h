/*
 * g3hip.h -- C ABI of libg3hip.so: the MI355X (gfx950) implementation of g3py's GP
 * inference hot path (Gram assembly + Cholesky + triangular solves for the log marginal
 * likelihood and the posterior mean / variance / covariance / draws).
 *
 * The reference (griosd/g3py) is pure Python on Theano; it has no FFI.  Each entry point
 * below names the reference seam it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes stub a g3py maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types cross the boundary.
 *   - Every function returns an int status: 0 = success; negative -i = argument i is
 *     invalid; G3_ERR_HIP = a HIP runtime error (text via g3_last_error); a positive
 *     LAPACK-style `info` (1-based order of the first non-positive pivot) is reported
 *     through the `info` out-parameter, never as the return value.
 *   - Nothing throws or aborts across the ABI.
 *   - All matrices are row-major (NumPy C order) with an explicit leading dimension `ld`
 *     counted in elements.  "dev" pointers are HIP device pointers (e.g. from g3_malloc
 *     or torch.Tensor.data_ptr()); "host" pointers are ordinary host memory, borrowed
 *     for the duration of the call only.
 *   - Work is enqueued on the context's HIP stream (g3_ctx_set_stream adopts a caller
 *     stream such as torch's current stream).  Functions with host out-parameters
 *     synchronise the stream before returning; functions with only device outputs are
 *     asynchronous.
 *   - One g3_ctx is single-threaded; distinct contexts may be used concurrently.
 */
#ifndef G3HIP_H
#define G3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G3_OK 0
#define G3_ERR_HIP (-1000)
#define G3_ERR_NOMEM (-1001)

typedef struct g3_ctx g3_ctx;

typedef enum { G3_F64 = 0, G3_F32 = 1 } g3_dtype;

/* ---- kernel description -------------------------------------------------------------
 * A covariance function is passed as a sum of products of stationary "leaf" kernels
 *      K(x1, x2) = shift + sum_p coef_p * prod_{t in p} leaf_t(x1, x2)
 * which is what any tree of g3py's KernelSum / KernelProd / KernelScale / KernelShift
 * (g3py/processes/hypers/kernels.py:192-244) over leaf kernels expands to.
 * Leaf formulas follow kernels.py:360-487 and metrics.py:30-35,59-61,89-91,100-102; all
 * hyper-parameters are in NATURAL space (the host applies exp() to `*_log_` values).
 */
#define G3_MAXD 32     /* max columns one leaf may use            */
#define G3_MAXCOLS 40  /* max columns d of the N x d input        */
#define G3_MAXLEAF 8   /* max distinct leaves in one program      */
#define G3_MAXPROD 16  /* max product terms                       */
#define G3_MAXFAC 4    /* max factors in one product term         */

typedef enum {
  G3_K_SE = 0,    /* var*exp(-sum_k 0.5*rate_k^2*dx_k^2)                 kernels.py:434-436, metrics.py:100-102 */
  G3_K_OU = 1,    /* var*exp(-sum_k rate_k*|dx_k|)                       kernels.py:429-431, metrics.py:89-91   */
  G3_K_MAT32 = 2, /* d=ARD_L2; s=sqrt(3d); var*(1+s)exp(-s)              kernels.py:406-412 */
  G3_K_MAT52 = 3, /* s=sqrt(5d); var*(1+s+5d/3)exp(-s)                   kernels.py:415-421 */
  G3_K_RQ = 4,    /* var*(1+d/alpha)^(-alpha)                            kernels.py:388-403 */
  G3_K_COS = 5,   /* var*prod_k cos(2 pi dx_k f_k)                       kernels.py:462-467 */
  G3_K_SIN = 6,   /* var*exp(+2 sum_k rate_k sin^2(pi dx_k f_k))         kernels.py:470-472 (sign as written) */
  G3_K_SINC = 7,  /* var*prod_k [dx!=0 ? sin(2pi^2 dx f)/(2pi^2 f dx):1] kernels.py:475-482 */
  G3_K_SM = 8,    /* var*exp(-2pi^2 sum dx^2 rate^2)*prod cos(2pi dx f)  kernels.py:485-487 */
  G3_K_NOISE = 9, /* square: var*I ; cross: 0                            kernels.py:360-371 */
  G3_K_WN = 10    /* square: var*I ; cross: var*#{k: dx_k==0}            kernels.py:374-385, metrics.py:30-35 */
} g3_kind;

typedef struct {
  int32_t kind;             /* g3_kind */
  int32_t ndims;            /* number of input columns used (x[:, dims], hypers/__init__.py:55-83) */
  int32_t dims[G3_MAXD];    /* the column indices */
  double var;
  double alpha;             /* RQ only */
  double rate[G3_MAXD];     /* per used column */
  double freq[G3_MAXD];     /* periodic family */
} g3_leaf;

typedef struct {
  double coef;
  int32_t nfac;
  int32_t fac[G3_MAXFAC];   /* leaf indices */
  int32_t _pad[3];
} g3_prod;

typedef struct {
  int32_t nleaf;
  int32_t nprod;
  double shift;
  g3_leaf leaf[G3_MAXLEAF];
  g3_prod prod[G3_MAXPROD];
} g3_kernel_prog;

/* g3_gram flags */
#define G3_GRAM_LOWER 1u     /* symmetric case: tiles strictly above the diagonal are not written */
#define G3_GRAM_SCRUB 2u     /* fuse tt_to_num: NaN->0, +-Inf->1e10 (g3py/libs/tensors.py:90-92) */
#define G3_GRAM_PAD_EYE 4u   /* rows/cols in [n, npad) get the identity (keeps a padded factor exact) */

/* ---- context and device memory ------------------------------------------------------ */
int g3_ctx_create(int device, g3_ctx** out);
int g3_ctx_destroy(g3_ctx* ctx);
int g3_ctx_set_stream(g3_ctx* ctx, void* hip_stream /* hipStream_t, NULL = context's own */);
int g3_ctx_sync(g3_ctx* ctx);
const char* g3_last_error(g3_ctx* ctx);
int g3_version(void);

int g3_malloc(g3_ctx* ctx, size_t bytes, void** dev);
int g3_free(g3_ctx* ctx, void* dev);
int g3_memcpy_h2d(g3_ctx* ctx, void* dev, const void* host, size_t bytes);
int g3_memcpy_d2h(g3_ctx* ctx, void* host, const void* dev, size_t bytes);
int g3_memcpy_d2d(g3_ctx* ctx, void* dst, const void* src, size_t bytes);
int g3_memset(g3_ctx* ctx, void* dev, int byte, size_t bytes);
/* strided 2-D copy dev->dev (rows x cols elements), used to pad / unpad matrices */
int g3_copy2d(g3_ctx* ctx, void* dst, int64_t ldd, const void* src, int64_t lds,
              int64_t rows, int64_t cols, g3_dtype dt);

/* ---- Gram assembly --------------------------------------------------------------------
 * Replaces Kernel.cov(x1, x2=None) (kernels.py:48-49,106-110) + Metric.gram
 * (metrics.py:11-13): K[i][j] = prog(X1[i], X2[j]); X2 == NULL selects the symmetric
 * (square) case, which is what turns on the NOISE / WN diagonal.  The n1 x n2 x d broadcast
 * intermediate of the reference is never formed.  K is (n1pad x n2pad), written for
 * i < n1pad, j < n2pad; entries outside n1 x n2 are 0 (or the identity with PAD_EYE).  In the
 * symmetric case n2pad may be smaller than n1: only the first n2pad columns are then written (a
 * block column of the covariance, as the multi-GPU path stores it). */
int g3_gram(g3_ctx* ctx, const g3_kernel_prog* prog_host,
            const void* X1_dev, int64_t n1, int64_t ldx1,
            const void* X2_dev, int64_t n2, int64_t ldx2, int d,
            g3_dtype dt, void* K_dev, int64_t ldk, int64_t n1pad, int64_t n2pad, unsigned flags);
/* diag(Kernel.cov(X)) without forming the matrix (for the posterior variance,
 * g3py/processes/elliptical.py:94-97) */
int g3_gram_diag(g3_ctx* ctx, const g3_kernel_prog* prog_host, const void* X_dev, int64_t n,
                 int64_t ldx, int d, g3_dtype dt, void* diag_dev);

/* tt_to_cov's diagonal lift (tensors.py:95-98), given a matrix already scrubbed by
 * G3_GRAM_SCRUB: m = min diag; if m <= 0 add (1e-6f - m) to the diagonal. */
int g3_cov_lift(g3_ctx* ctx, void* K_dev, int64_t n, int64_t ld, g3_dtype dt);
/* Which Gram kernel the launches of this context used so far: out_host = [compile-time table (one stationary kernel
 * (+ noise) (+ or x one periodic term)), generated for the expression at first use (hipRTC, g3_gram_jit.hip: any
 * KernelSum / KernelProd / KernelScale / KernelShift tree, kernels.py:192-244), interpreted].  G3_GRAM_JIT=0 or a
 * missing libhiprtc leaves the interpreter; all three evaluate the same formulas. */
int g3_gram_path_stats(g3_ctx* ctx, double out_host[3]);
/* The same three counters for the kernel-parameter sums of the gradient (g3_gram_grad, g3_gp_dlogp*): compile-time table,
 * generated at first use (g3_gram_jit.hip::g3_grad_jit), interpreted. */
int g3_grad_path_stats(g3_ctx* ctx, double out_host[3]);
/* Build-host check (no GPU, no context): does the kernel generated for this expression compile for gfx950?  0 and the
 * size of the code object, or hipRTC's status and its log (-1: libhiprtc not available). */
int g3_gram_jit_check(const g3_kernel_prog* prog_host, int d, g3_dtype dt, int64_t* code_bytes, char* log, int64_t log_bytes);
/* The same for the generated kernel of the kernel-parameter sums of the gradient (g3_gram_grad and the dlogp entry points). */
int g3_grad_jit_check(const g3_kernel_prog* prog, int d, g3_dtype dt, int64_t* code_bytes, char* log, int64_t log_bytes);
/* tt_to_num over a dense n1 x n2 matrix (tensors.py:90-92) */
int g3_scrub(g3_ctx* ctx, void* A_dev, int64_t n1, int64_t n2, int64_t ld, g3_dtype dt);

/* ---- dense kernels ---------------------------------------------------------------------
 * C[m x n] = alpha * A[m x k] * B[n x k]^T + beta * C   (all row-major, k contiguous).
 * m, n multiples of 64, k a multiple of 16 (f64) / 32 (f32).  lower_only: only tiles on
 * or below the diagonal are touched and elements above the diagonal are left unchanged. */
int g3_gemm_nt(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda,
               const void* B, int64_t ldb, int64_t m, int64_t n, int64_t k,
               double alpha, double beta, g3_dtype dt, int lower_only);
/* Staircase product in ONE launch: C is a stack of nseg row segments (seg_rows[s] rows each, in
 * order); segment s gets its first seg_cols[s] columns:
 *   C[rows_s, 0:seg_cols[s]) = alpha * A[rows_s, 0:k) * B[0:seg_cols[s], 0:k)^T + beta * C.
 * seg_rows / seg_cols are HOST arrays (borrowed for the call), entries multiples of 128 (0 allowed).
 * b_perm (HOST, may be NULL): B is stored as blocks of b_block_rows rows (a multiple of 128) and
 * logical block s -- the rows that multiply columns [s, s+1) * b_block_rows of C -- is physical
 * block b_perm[s]; nperm entries.  seg_diag (HOST, may be NULL): seg_diag[s] != 0 says the last seg_rows[s]
 * columns of segment s are its square DIAGONAL block, of which only the lower triangle is wanted -- tiles
 * entirely above it are skipped (elements above the diagonal inside a computed tile are still written: the strict
 * upper part of a diagonal block is scratch).  This is the trailing update of one rank of the multi-GPU
 * factorisation: its row blocks of the lower triangle have different widths and the all-gathered
 * panel arrives rank-major (g3py/libs/tensors.py:198 is one dpotrf on one host). */
int g3_gemm_nt_stair(g3_ctx* ctx, void* C, int64_t ldc, const void* A, int64_t lda,
                     const void* B, int64_t ldb, int64_t k, const int64_t* seg_rows,
                     const int64_t* seg_cols, int nseg, double alpha, double beta, g3_dtype dt,
                     int64_t b_block_rows, const int32_t* b_perm, int nperm, const int64_t* seg_diag);

/* In-place lower Cholesky of the lower triangle of A (n x n, n a multiple of 128; the
 * strict upper triangle is neither read nor written).  Replaces the dpotrf call at
 * g3py/libs/tensors.py:198.  Blocked right-looking sweep over 1024-wide panels with one-panel
 * look-ahead on two HIP streams (recursive inside a panel): every 128x128 diagonal block is
 * factored AND inverted by one fused workgroup kernel, panels are solved by in-place GEMM
 * against the block inverses, trailing updates are MFMA SYRK/GEMM.  `invd_dev` (n/128 blocks
 * of 128x128, may be NULL to use the context's own buffer) receives inv(L_kk) for every
 * diagonal block, for reuse by g3_trsm_rlt.  *info_host = 0 or the 1-based index of the first
 * non-positive (or NaN) pivot. */
int g3_potrf(g3_ctx* ctx, void* A_dev, int64_t n, int64_t ld, g3_dtype dt, void* invd_dev,
             int* info_host);

/* g3_potrf without the host synchronisation (look-ahead streams of the multi-GPU driver):
 * info_accum_dev is a 4-byte DEVICE integer, zeroed by the caller, that keeps the first non-zero
 * info of all calls made with it; invd_dev is required. */
int g3_potrf_nowait(g3_ctx* ctx, void* A_dev, int64_t n, int64_t ld, g3_dtype dt, void* invd_dev,
                    int* info_accum_dev);

/* CholeskyRobust.perform (tensors.py:197-222): non-destructive, never fails.
 * L (lower, strict upper zeroed) <- chol(K); on info != 0 runs the reference's jitter
 * schedule (dK = mean(diag)*1e-6, lift non-positive diagonals, up to `maxtries` (20)
 * retries with dK *= 10) and finally the 1e-10*I fallback.  K and L are n x n with any n
 * (padding is internal).  tries_host / fallback_host / jitter_host report what happened. */
int g3_potrf_robust(g3_ctx* ctx, const void* K_dev, int64_t ldk, void* L_dev, int64_t ldl,
                    int64_t n, g3_dtype dt, int maxtries, int* tries_host, int* fallback_host,
                    double* jitter_host);

/* Solve X * L^T = B in place (B is m x n: each ROW of B is one right-hand side, i.e.
 * B^T <- solve_lower_triangular(L, B^T); tensors.py:265-270, gaussian.py:212).
 * n a multiple of 128, m a multiple of 128.  invd_dev: block inverses from g3_potrf, or NULL to
 * compute them from L. */
int g3_trsm_rlt(g3_ctx* ctx, const void* L_dev, int64_t n, int64_t ldl, void* B_dev, int64_t m,
                int64_t ldb, g3_dtype dt, const void* invd_dev);

/* The FULL inverse V = L^-1 of a factored n x n lower block, n = 128 * 2^q <= 2048, by recursive doubling from the
 * inverses of its 128 x 128 diagonal blocks (invd_dev, as g3_potrf leaves them): V21 = -V22 L21 V11 level by level, two
 * small MFMA launches per level.  L, V and the two scratch matrices Vt (= V^T on return, except for the last level) and U
 * are COMPACT n x n (leading dimension n); only the lower block triangle of V is written.  The multi-GPU driver broadcasts
 * V instead of (L, invd): a rank's panel solve -- the `solve_lower_triangular` of tensors.py:265-270 / gaussian.py:212 --
 * is then one product (g3_trsm_full).  |V - L^-1| ~ kappa(L) eps per level. */
int g3_trtri_full(g3_ctx* ctx, const void* L_dev, int64_t n, const void* invd_dev, void* V_dev, void* Vt_dev, void* U_dev,
                  g3_dtype dt);

/* X_out = B V^T for V = L^-1 from g3_trtri_full (n x n, leading dimension ldv): X L^T = B solved as ONE K-triangular MFMA
 * product, out of place (X_out must not overlap B).  B, X_out: m x n, m a multiple of 64, n a multiple of 128. */
int g3_trsm_full(g3_ctx* ctx, const void* V_dev, int64_t n, int64_t ldv, const void* B_dev, int64_t m, int64_t ldb, void* X_dev,
                 int64_t ldx, g3_dtype dt);

/* ---- reductions (device in, host out; synchronising) ----------------------------------- */
/* out[0] = sum_i log L[i][i], out[1] = sum_i a[i]^2, out[2] = #non-finite in a,
 * out[3] = #non-finite or <= 0 on diag(L)   -- the pieces of logp_cho, gaussian.py:208-241 */
int g3_logp_terms(g3_ctx* ctx, const void* L_dev, int64_t n, int64_t ld, const void* a_dev,
                  g3_dtype dt, double out_host[4]);
/* out = [min, mean, max] of diag(A) (jitter schedule inputs, tensors.py:203-206) */
int g3_diag_stats(g3_ctx* ctx, const void* A_dev, int64_t n, int64_t ld, g3_dtype dt,
                  double out_host[3]);
int g3_diag_add(g3_ctx* ctx, void* A_dev, int64_t n, int64_t ld, g3_dtype dt, double value);
/* For V (m x n, rows are L^-1 k_*): dot[i] = sum_j V[i][j]*a[j]; ss[i] = sum_j V[i][j]^2
 * -- posterior mean and variance pieces (elliptical.py:81-97) in one pass over V. */
int g3_rows_dot_ss(g3_ctx* ctx, const void* V_dev, int64_t m, int64_t n, int64_t ld,
                   const void* a_dev, g3_dtype dt, void* dot_dev, void* ss_dev);

/* ---- fused hot path (device-resident inputs) -------------------------------------------
 * One evaluation of the GP log marginal likelihood pieces for
 *     K = tt_to_cov(prog(X, X))   (prog already contains the Noise term, elliptical.py:26-31,70-71)
 *     L = cholesky_robust(K); a = L^-1 delta          (gaussian.py:208-224)
 * X_dev: N x d (row stride ldx), delta_dev: N (= T^-1(y) - m(X), computed by the host layer).
 * K_dev: workspace / output, (Npad + 128) x Npad with Npad = roundup(N, 128), ld = ldk; on return
 * the lower triangle of its first Npad rows holds L; the 128 trailing rows are the right-hand-side
 * block that carries delta THROUGH the factorisation (the panel solves and trailing updates that
 * factor K also perform the forward substitution -- there is no separate trsv pass).  invd_dev (Npad/128 blocks of 128x128) receives the diagonal-block
 * inverses that g3_gp_cross needs.  a_dev (Npad) receives L^-1 delta.
 * out_host[0] = sum log L_ii, [1] = a^T a, [2] = #non-finite in a, [3] = jitter tries,
 * [4] = 1 if the 1e-10*I fallback was taken, [5] = potrf info of the first attempt. */
int g3_gp_factor(g3_ctx* ctx, const g3_kernel_prog* prog_host, const void* X_dev, int64_t N,
                 int64_t ldx, int d, const void* delta_dev, g3_dtype dt, void* K_dev, int64_t ldk,
                 void* invd_dev, void* a_dev, double out_host[6]);

/* g3_gp_factor plus the posterior pieces of M test points in ONE sweep: K_dev has
 * Npad + 128 + Mpad rows (Mpad = roundup(M, 128)); rows [Npad+128, +Mpad) receive
 * V = K(Xs, X) L^-T, carried through the factorisation like the delta row.
 * mu[i] = V[i,:] . a, ss[i] = |V[i,:]|^2 (mean = m(Xs) + mu, var = diag K** - ss; elliptical.py:81-97).
 * prog_cross: kernel of the cross covariance (elliptical.py:78-79). */
int g3_gp_factor_predict(g3_ctx* ctx, const g3_kernel_prog* prog_host, const g3_kernel_prog* prog_cross,
                         const void* X_dev, int64_t N, int64_t ldx, int d, const void* delta_dev,
                         const void* Xs_dev, int64_t M, int64_t ldxs, g3_dtype dt, void* K_dev, int64_t ldk,
                         void* invd_dev, void* a_dev, void* mu_dev, void* ss_dev, double out_host[6]);

/* Posterior location / variance pieces at M test points given the factor from g3_gp_factor:
 *     V = K(Xs, X) L^-T  (Mpad x Npad with Mpad = roundup(M, 128), in V_dev, ldv);
 *     mu[i] = V[i,:] . a;  ss[i] = |V[i,:]|^2
 * prog_cross: the kernel used for the cross covariance (with or without the Noise term,
 * elliptical.py:78-79 -- Noise contributes 0 to a cross block either way). */
int g3_gp_cross(g3_ctx* ctx, const g3_kernel_prog* prog_cross, const void* Xs_dev, int64_t M,
                int64_t ldxs, const void* X_dev, int64_t N, int64_t ldx, int d, const void* L_dev,
                int64_t ldl, const void* invd_dev, const void* a_dev, g3_dtype dt, void* V_dev, int64_t ldv,
                void* mu_dev, void* ss_dev);

/* Row block of the square covariance (the layout of the multi-GPU driver, g3py_amd/distributed.py):
 * K_dev (nrows x (row0 + nrows), row stride ldk) <- rows [row0, row0 + nrows), columns
 * [0, row0 + nrows) of Kernel.cov(X) with the SQUARE-case semantics of kernels.py:360-385 (NOISE /
 * WN on the true diagonal only).  Rows / columns beyond N: identity with G3_GRAM_PAD_EYE, else 0.
 * flags: G3_GRAM_SCRUB, G3_GRAM_PAD_EYE (G3_GRAM_LOWER is refused). */
int g3_gram_rows(g3_ctx* ctx, const g3_kernel_prog* prog, const void* X_dev, int64_t N, int64_t ldx, int d,
                 int64_t row0, int64_t nrows, g3_dtype dt, void* K_dev, int64_t ldk, unsigned flags);

/* ---- batched evaluation -----------------------------------
 * `batch` independent g3_gp_factor evaluations on the SAME inputs X with different kernel
 * hyper-parameters and right-hand sides: the caller pattern of logp_chain / fixed_logp /
 * find_MAP restarts (g3py/processes/stochastic.py:515-564, 740-771 -- a Python loop in the
 * reference).  progs_host[b] must all have the same structure (leaf kinds, dims, products);
 * delta_dev is batch x N with row stride ldd; K_dev holds the members kstride elements apart,
 * each laid out as for g3_gp_factor ((roundup(N,128)+128) x ldk); invd_dev is batch x
 * roundup(N,128) x 128; a_dev is batch x roundup(N,128); out_host is batch x 6 (as g3_gp_factor).
 * One Gram launch (grid.z = batch) and ONE factorisation sweep whose MFMA-GEMM and
 * diagonal-block launches carry the batch in grid.y; members whose first factorisation fails are
 * re-run one at a time through the jitter schedule (tensors.py:203-222).  batch <= 4096. */
int g3_gp_factor_batched(g3_ctx* ctx, const g3_kernel_prog* progs_host, int batch, const void* X_dev, int64_t N,
                         int64_t ldx, int d, const void* delta_dev, int64_t ldd, g3_dtype dt, void* K_dev,
                         int64_t ldk, int64_t kstride, void* invd_dev, void* a_dev, double* out_host);

/* The same sweep with the members' programs given as ONE template plus what differs: member b is
 * the template with the double at byte offset offsets_host[i] of g3_kernel_prog replaced by
 * fields_host[b * nfield + i] (i < nfield <= G3_MAX_FIELDS).  Offsets must name double members
 * (shift, leaf var / alpha / rate[k] / freq[k], product coef).  The members are expanded on the
 * device, so a chain row costs nfield doubles of packing and PCIe instead of a 6 KB program --
 * what lets logp_chain / fixed_logp (stochastic.py:515-532) keep up with the one-workgroup-per-
 * member evaluation at N <= 256.  Members that fail the first factorisation are rebuilt on the
 * host and re-run through the jitter schedule exactly as above.  Returns as g3_gp_factor_batched
 * (argument numbers follow this signature). */
#define G3_MAX_FIELDS (1 + G3_MAXLEAF * (2 + 2 * G3_MAXD) + G3_MAXPROD)
int g3_gp_factor_batched_fields(g3_ctx* ctx, const g3_kernel_prog* tmpl_host, int batch, const double* fields_host,
                                const int32_t* offsets_host, int nfield, const void* X_dev, int64_t N, int64_t ldx,
                                int d, const void* delta_dev, int64_t ldd, g3_dtype dt, void* K_dev, int64_t ldk,
                                int64_t kstride, void* invd_dev, void* a_dev, double* out_host);

/* ---- gradient of logp w.r.t. the kernel hyper-parameters ----
 * Reference: StochasticProcess.th_dlogp = gradient(th_logp) (g3py/processes/stochastic.py:308-309;
 * g3py/libs/tensors.py:11-22), i.e. Theano's reverse mode through logp_cho (gaussian.py:208-224)
 * and CholeskyRobust.grad (tensors.py:224-260).  Closed form used here:
 *     d logp / d theta = 1/2 sum_ij G_ij dK_ij/dtheta,   G = alpha alpha^T - K^-1,  alpha = K^-1 delta
 * with K the matrix that was factored (grad() re-uses the possibly jittered factor the same way).
 *
 * g3_grad_map tells the device where each parameter of each leaf of a g3_kernel_prog goes in
 * the output vector: slot index, or -1 to skip; `rate` / `freq` name the first of `ndims`
 * consecutive slots.  g3_grad_layout fills the standard map (every parameter of every leaf, in
 * leaf order: var, [alpha], [freq...], [rate...] as the leaf kind has them). */
#define G3_GRAD_MAXSLOTS (G3_MAXLEAF * (2 + 2 * G3_MAXD))
typedef struct g3_grad_map {
  int32_t nslots;
  int32_t var[G3_MAXLEAF];
  int32_t alpha[G3_MAXLEAF];
  int32_t rate[G3_MAXLEAF];
  int32_t freq[G3_MAXLEAF];
} g3_grad_map;
int g3_grad_layout(const g3_kernel_prog* prog, g3_grad_map* map);

/* From a lower Cholesky factor L (n x n, n a multiple of 128, identity padded) and the inverses
 * of its 128 x 128 diagonal blocks (invd_dev, NULL = computed into the context's scratch):
 * Y_dev <- L^-T (upper triangular, row-major) and the lower triangle of Kinv_dev <- (L L^T)^-1
 * = Y Y^T.  2 n^3 / 3 flops in the MFMA GEMM (LAPACK dpotri's work), look-ahead on two streams. */
int g3_potri(g3_ctx* ctx, const void* L_dev, int64_t n, int64_t ldl, const void* invd_dev, g3_dtype dt,
             void* Y_dev, int64_t ldy, void* Kinv_dev, int64_t ldc);

/* out_host[slot] = 1/2 sum_{i,j < N} (alpha_i alpha_j - Kinv_ij) * d prog(x_i, x_j) / d param(slot)
 * over the FULL symmetric index range (only the lower triangle of Kinv_dev is read).  Parameters
 * are the natural-space `var`, `alpha`, `rate[k]`, `freq[k]` fields of the leaves; NOISE / WN
 * leaves contribute on the diagonal (square case, kernels.py:360-385).  One pass over Kinv
 * (HBM-read bound); partial sums are combined in a fixed order (bitwise reproducible). */
int g3_gram_grad(g3_ctx* ctx, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X_dev,
                 int64_t N, int64_t ldx, int d, g3_dtype dt, const void* Kinv_dev, int64_t ldc,
                 const void* alpha_dev, double* out_host);

/* The same sum restricted to rows [row0, row0 + nrows) of the lower triangle (row0 a multiple of 64, row0 + nrows <= N):
 * Kinv_rows_dev holds THOSE rows of K^-1 (row r of the buffer = global row row0 + r; columns 0 .. row0 + nrows - 1 are
 * read), alpha_dev all N entries.  Disjoint row ranges add up to g3_gram_grad's result: a rank of the multi-GPU
 * driver calls this once per row block it owns (g3_dist_gp_dlogp).  No counterpart in the reference (its gradient is
 * Theano autodiff through CholeskyRobust.grad, g3py/libs/tensors.py:224-260). */
int g3_gram_grad_rows(g3_ctx* ctx, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X_dev,
                      int64_t N, int64_t ldx, int d, g3_dtype dt, int64_t row0, int64_t nrows,
                      const void* Kinv_rows_dev, int64_t ldc, const void* alpha_dev, double* out_host);

/* Fused: after g3_gp_factor (L_dev = its K_dev, invd_dev, a_dev = L^-1 delta, roundup(N,128) long):
 * g3_potri, alpha_dev <- L^-T a (roundup(N,128) entries), then g3_gram_grad.  Y_dev and Kinv_dev
 * are roundup(N,128)-square workspaces / outputs. */
int g3_gp_dlogp(g3_ctx* ctx, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X_dev,
                int64_t N, int64_t ldx, int d, const void* L_dev, int64_t ldl, const void* invd_dev,
                const void* a_dev, g3_dtype dt, void* Y_dev, int64_t ldy, void* Kinv_dev, int64_t ldc,
                void* alpha_dev, double* out_host);

/* Batched g3_gp_dlogp after ONE g3_gp_factor_batched sweep (same member layout: factors kstride
 * elements apart in L_dev with leading dimension ldl, block inverses roundup(N,128)*128 apart,
 * a_dev batch x roundup(N,128)).  Replaces the reference's Python loop over single gradients
 * (fixed_dlogp, g3py/processes/stochastic.py:554-564).  Y_dev and Kinv_dev hold `batch` members
 * kstride apart (ld = ldl); alpha_dev is batch x roundup(N,128); out_host is batch x map->nslots.
 * All programs must share one structure (one g3_grad_map). */
int g3_gp_dlogp_batched(g3_ctx* ctx, const g3_kernel_prog* progs, int batch, const g3_grad_map* map,
                        const void* X_dev, int64_t N, int64_t ldx, int d, const void* L_dev, int64_t ldl,
                        int64_t kstride, const void* invd_dev, const void* a_dev, g3_dtype dt, void* Y_dev,
                        void* Kinv_dev, void* alpha_dev, double* out_host);

/* The same with the members' programs given as g3_gp_factor_batched_fields takes them (template + per-member doubles at byte
 * offsets of g3_kernel_prog): what the binding of dlogp_chain / fixed_dlogp (stochastic.py:554-564) packs per chain block.
 * Argument numbers in negative return codes follow this signature. */
int g3_gp_dlogp_batched_fields(g3_ctx* ctx, const g3_kernel_prog* tmpl_host, int batch, const double* fields_host,
                               const int32_t* offsets_host, int nfield, const g3_grad_map* map, const void* X_dev, int64_t N,
                               int64_t ldx, int d, const void* L_dev, int64_t ldl, int64_t kstride, const void* invd_dev,
                               const void* a_dev, g3_dtype dt, void* Y_dev, void* Kinv_dev, void* alpha_dev, double* out_host);

/* Latent draws of the sampler (g3py/processes/gaussian.py:89-95, before the mapping):
 *     out[i][s] = loc[i] + sum_j L[i][j] Z[j][s]
 * L_dev: lower Cholesky factor of the prior / posterior covariance of the M query points, stored
 * roundup(M,128)-square and zero outside its M x M lower triangle (what g3_potrf_robust writes into
 * a zeroed buffer); loc_host: M values; Z_host: the caller's standard normals, M x S row-major
 * (np.random.randn(len(space), samples), gaussian.py:91); out_host: M x S row-major.  The product
 * runs in the MFMA GEMM; host buffers are borrowed for the call. */
int g3_gp_sample(g3_ctx* ctx, const void* L_dev, int64_t M, int64_t ldl, const void* loc_host,
                 const void* Z_host, int64_t S, g3_dtype dt, void* out_host);

/* ---- profiling (bench.py's live roofline measurement) -------------------------------------
 * When enabled, HIP-event pairs are recorded ON THE CONTEXT'S STREAM around every launch of
 * the MFMA GEMM (one tag per tile configuration) and around the phases of g3_gp_factor /
 * g3_gp_cross.  g3_prof_collect synchronises and returns, per tag t (G3_PROF_NTAGS of them):
 * out[3t] = number of regions, out[3t+1] = summed milliseconds, out[3t+2] = summed
 * algorithmic work (flops, or bytes for the Gram tags).  Tags: 0 MFMA GEMM, 128x128 tile,
 * launches with >= 4096 tiles (the bulk panel updates), 1 Gram, 2 potrf (whole factorisation),
 * 3 trsv (L^-1 delta), 4 cross Gram, 5 trsm (predict), 6 reductions; and with on = 2 also
 * 7 other 128x128-tile GEMM launches, 8 small-tile GEMM launches, 9 fused diagonal-block kernels;
 * on = 3 times only every 16th of those small launches (a sample, for launch-bound callers). */
#define G3_PROF_NTAGS 10
int g3_prof_enable(g3_ctx* ctx, int on);
int g3_prof_reset(g3_ctx* ctx);
int g3_prof_collect(g3_ctx* ctx, double* out_host /* 3 * G3_PROF_NTAGS */);

/* ---- multi-GPU: the N x N covariance block-partitioned over the GPUs of one node ------------
 * One process per GPU; the library owns the RCCL communicators (librccl is dlopen'ed by g3_dist_create only) and the
 * whole per-panel loop -- a caller needs neither torch nor a Python loop.  The reference has no distributed code
 * (its only parallelism is a process pool, g3py/processes/stochastic.py:773-783); what is distributed here is the
 * algebra of tensors.py:197-222 (CholeskyRobust incl. the jitter schedule and the 1e-10*I fallback),
 * gaussian.py:208-224 (logp pieces) and elliptical.py:81-97 (posterior location / variance), SURVEY.md 8(b)/(e).
 * Layout, schedule and byte counts: DESIGN.md section 6 (row-block-cyclic; diagonal-factor broadcast on its own
 * communicator and stream, panel all-gather, bulk staircase updates on a third stream).
 *
 * Bootstrap: rank 0 calls g3_dist_unique_id twice and ships the 2 x G3_DIST_ID_BYTES to every rank by any channel
 * (MPI, a file, torch.distributed over gloo ...); every rank then calls g3_dist_create (collective). */
typedef struct g3_dist g3_dist;
#define G3_DIST_ID_BYTES 128
int g3_dist_unique_id(void* id_out /* G3_DIST_ID_BYTES */);
int g3_dist_create(g3_ctx* ctx, const void* id_gather, const void* id_bcast, int rank, int world, g3_dist** out);
/* Test transport: the three collectives are supplied by the caller as blocking host callbacks on DEVICE buffers
 * (return 0 on success).  Lets several ranks share ONE GPU (RCCL refuses that) so the schedule can be checked for
 * world > 1 on a one-GPU box; never used by the product path.  The panel all-gathers are issued IN PLACE, as RCCL
 * defines it: send_dev == recv_dev + rank * bytes_per_rank; a callback must read its block before it writes recv_dev. */
typedef struct {
  void* user;
  int (*bcast)(void* user, void* buf_dev, size_t bytes, int root);
  int (*allgather)(void* user, const void* send_dev, void* recv_dev, size_t bytes_per_rank);
  int (*allreduce)(void* user, double* vals_host, int n, int op /* 0 sum, 1 min, 2 max */);
} g3_dist_callbacks;
int g3_dist_create_callbacks(g3_ctx* ctx, const g3_dist_callbacks* cb, int rank, int world, g3_dist** out);
/* ASYNCHRONOUS test transport (round 5): the same three collectives as host callbacks, but on HOST staging buffers and
 * served by two worker threads the library owns -- one for the panel all-gathers and scalar all-reduces, one for the
 * diagonal-factor broadcasts, i.e. the product's two communicators.  A collective is stream-ordered exactly like an RCCL
 * call: the library records an event on the stream, hands the job to the worker and makes the STREAM (not the host) wait
 * for it; the worker waits for the event, stages device -> pinned host, calls the callback, stages back and releases the
 * stream.  The host thread runs ahead over the whole sweep, collectives of both kinds are in flight together beside the
 * three streams' kernels, and a cross-rank ordering or buffer-reuse fault shows -- which the blocking transport above
 * (a hipStreamSynchronize before every collective) serialises away.  `bcast` is called from the broadcast worker,
 * `allgather` and `allreduce` from the gather worker, each in the order the library issues them (the same on every rank):
 * serve the two kinds from two independent communicators (e.g. two gloo process groups).  allgather: host_recv holds
 * world x bytes_per_rank, this rank's part already at host_recv + rank * bytes_per_rank (= host_send). */
typedef struct {
  void* user;
  int (*bcast)(void* user, void* host_buf, size_t bytes, int root);
  int (*allgather)(void* user, const void* host_send, void* host_recv, size_t bytes_per_rank);
  int (*allreduce)(void* user, double* vals_host, int n, int op /* 0 sum, 1 min, 2 max */);
} g3_dist_host_callbacks;
int g3_dist_create_callbacks_async(g3_ctx* ctx, const g3_dist_host_callbacks* cb, int rank, int world, g3_dist** out);
/* Replay transport (measurement): ONE rank of a `world`-rank evaluation, alone on one GPU.  `reference` is a world-1
 * driver (either transport) for which g3_dist_set_keep(reference, 1) was called BEFORE g3_dist_plan and which has since
 * evaluated the same problem: its local matrix is the whole factor, its block inverses were kept.  In the replayed
 * rank every collective is a device-to-device copy of exactly the bytes that rank would receive (the diagonal factors it
 * does not own, the other ranks' blocks of every panel, a = L^-1 delta) and the scalar all-reduces return the rank's own
 * contribution: g3_dist_gp_factor_predict then times everything of a P-rank evaluation except the fabric, and returns
 * this rank's share of the log-determinant, of a^T a and of the posterior means / sums of squares.  The plan must repeat
 * the reference's (N, M, nb, dtype); gradient mode, the posterior covariance and the jitter schedule are not replayed;
 * the reference must outlive the replay object.  There is nothing of this in the reference package (it has no
 * multi-device code, stochastic.py:773-783); it exists because a one-GPU box cannot run RCCL with P > 1. */
int g3_dist_set_keep(g3_dist* D, int on);
int g3_dist_create_replay(g3_ctx* ctx, g3_dist* reference, int rank, int world, g3_dist** out);
int g3_dist_destroy(g3_dist* D);
const char* g3_dist_last_error(g3_dist* D);
/* Problem shape: N observations in d columns, M test points, nb-row blocks (multiple of 128).  Allocates the
 * rank's rows of the covariance ((its blocks + its right-hand-side chunks) x roundup(N, nb)) and the panel buffers. */
int g3_dist_plan(g3_dist* D, int64_t N, int d, int64_t M, int64_t nb, g3_dtype dt);
/* One evaluation (the multi-GPU g3_gp_factor_predict): every rank passes the SAME replicated device inputs
 * X (N x d), delta (N), Xs (M x d).  out_host = [sum log L_ii, a^T a, potrf info of the first attempt, jitter tries,
 * 1 if the 1e-10*I fallback was taken]; mean_host[i] = V_i . a, ss_host[i] = |V_i|^2 (all ranks get all M). */
int g3_dist_gp_factor_predict(g3_dist* D, const g3_kernel_prog* prog_noise, const g3_kernel_prog* prog_cross,
                              const void* X_dev, int64_t ldx, const void* delta_dev, const void* Xs_dev, int64_t ldxs,
                              double out_host[5], double* mean_host, double* ss_host);
/* The posterior covariance itself, K(Xs, Xs) - V V^T (elliptical.py:86-91; prog with the Noise term for the noisy
 * covariance, without for the f process -- NOISE / WN leaves act on the true diagonal as in kernel.cov(Xs)), from the
 * last g3_dist_gp_factor_predict at the same Xs: every rank forms the lower part of the rows of its chunks, the matrix
 * is all-gathered and mirrored, and EVERY rank's cov_dev (roundup(M,128) square, row stride ldc, zero outside M x M)
 * receives the full symmetric matrix.  The driver remembers the prediction points of its last evaluation: any other
 * Xs is refused (-3, text in g3_dist_last_error) instead of being combined with cross-solve rows that belong elsewhere. */
int g3_dist_posterior_cov(g3_dist* D, const g3_kernel_prog* prog, const void* Xs_dev, int64_t ldxs, void* cov_dev,
                          int64_t ldc);

/* After g3_dist_gp_factor_predict: posterior covariance K_f(Xs, Xs) - V V^T, its robust Cholesky and the latent draws
 * loc + L_post Z (gaussian.py:75-97, elliptical.py:86-92).  loc_host (M), Z_host (M x S), out_host (M x S) are host
 * arrays of the plan's dtype; every rank returns the same draws. */
int g3_dist_posterior_draws(g3_dist* D, const g3_kernel_prog* prog_f, const void* Xs_dev, int64_t ldxs,
                            const void* loc_host, const void* Z_host, int64_t S, void* out_host, int* tries_host,
                            int* fallback_host);
/* Gradient mode.  g3_dist_set_grad(D, 1): every later g3_dist_gp_factor_predict also carries the IDENTITY as
 * right-hand-side rows, dealt like the matrix's row blocks, which the sweep turns into the rank's rows of L^-T (N^3 / 3
 * more flops over all ranks; the local matrix doubles; a planned driver is re-planned).  g3_dist_gp_dlogp then returns
 * what g3_gp_dlogp returns on one GPU -- slots_host[map->nslots] = 1/2 sum_ij (alpha_i alpha_j - K^-1_ij) dK_ij/dparam
 * and alpha_host[N] = alpha_scale * K^-1 delta (NULL: not wanted), identical on every rank -- from the last
 * factorisation: alpha = L^-T a; K^-1 = L^-T L^-1 by one all-gathered panel of L^-T + one staircase GEMM per column
 * block (no dependency between the steps); g3_gram_grad_rows over the rank's row blocks; one all-reduce of the sums.
 * prog must be the (noise-wrapped) program the factorisation was built from.  alpha_scale: sqrt(s) of a density whose
 * d logp / d|L^-1 delta|^2 is -s/2 (1 for the Gaussian).  Replaces gradient(th_logp) (g3py/processes/stochastic.py:
 * 308-309) through CholeskyRobust.grad (g3py/libs/tensors.py:224-260) for a covariance no single GPU holds. */
int g3_dist_set_grad(g3_dist* D, int on);
int g3_dist_gp_dlogp(g3_dist* D, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X_dev, int64_t ldx,
                     double alpha_scale, double* slots_host, double* alpha_host);

/* out_host[3k .. 3k+2] for k = 0 broadcast, 1 all-gather, 2 all-reduce: calls, bytes sent + received by this rank,
 * device milliseconds inside the collective calls (HIP events on the stream each ran on).  Resets the counters. */
int g3_dist_comm_stats(g3_dist* D, double out_host[9]);
/* As of the last g3_dist_comm_stats: out_host = [diagonal blocks this rank updated and factored (its turns on the
 * critical chain, tensors.py:198 per block), device ms of those (update + factorisation + copies); panel solves of its
 * rows, device ms of those] -- HIP events on the streams they ran on. */
int g3_dist_phase_stats(g3_dist* D, double out_host[4]);
/* g3_prof_enable / g3_prof_collect for the driver's bulk stream (its staircase MFMA-GEMM launches), same layout */
int g3_dist_prof_enable(g3_dist* D, int on);
int g3_dist_prof_collect(g3_dist* D, double* out_host /* 3 * G3_PROF_NTAGS */);
int g3_dist_local_rows(g3_dist* D, int64_t* rows_mat, int64_t* rows_rhs, int64_t* ld);

#ifdef __cplusplus
}
#endif
#endif /* G3HIP_H */
