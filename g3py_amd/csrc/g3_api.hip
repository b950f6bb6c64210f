// Context, device memory, element-wise / reduction kernels and the fused hot-path entry
// points of libg3hip (see include/g3hip.h for the contract of each function).
#include "g3_internal.h"
#include "g3_host.h"
#include <stdlib.h>

// ----------------------------------------------------------------------------- context
extern "C" int g3_version(void) { return 100; }

static int ctx_create_impl(int device, hipStream_t on_stream, g3_ctx** out);
extern "C" int g3_ctx_create(int device, g3_ctx** out) { return ctx_create_impl(device, nullptr, out); }
// A context that works on the caller's stream and creates NO stream of its own (the side stream of a two-stream sweep is
// created the first time one is needed, g3i_ensure_side_stream).  The multi-GPU driver's look-ahead and bulk contexts are
// made this way: HIP maps streams onto a few hardware queues per priority, and every idle stream a process holds shifts
// which of the streams that matter end up sharing one.
int g3i_ctx_create_on(int device, hipStream_t stream, g3_ctx** out) { return stream ? ctx_create_impl(device, stream, out) : -3; }
// ---- stream placement probe (shared with the multi-GPU driver, g3_dist.hip::pick_streams).  Two HIP streams whose hardware
// queues sit on the same compute pipe disturb each other: while the pipe dispatches a CU-filling launch of one, every small
// kernel of the other waits ~100 us instead of ~35 (and most of the launch when they share the QUEUE).  HIP decides the
// placement from the process's stream history and offers no way to ask, so it is measured.
#include <time.h>
__global__ void probe_long_kernel(int iters) {
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
}
__global__ void probe_tiny_kernel(unsigned* out) {
  if (threadIdx.x == 0 && out) *out = 1u;
}
static double probe_now_us() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}
// latency (us) of a one-wave kernel submitted on `a` while the dispatch-bound kernel runs on `b`, and that kernel's duration
bool g3i_probe_pair(hipStream_t a, hipStream_t b, unsigned* scratch, double* tiny_us, double* long_us, int reps) {
  double best = 1e30, lbest = 1e30;
  for (int rep = 0; rep < reps; ++rep) {
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    const double t0 = probe_now_us();
    hipLaunchKernelGGL(probe_long_kernel, dim3(32768), dim3(256), 0, b, 12);
    // let the long kernel get going, then submit the small one and wait for it
    while (probe_now_us() - t0 < 100.0) {}
    const double t1 = probe_now_us();
    hipLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, a, scratch);
    if (hipStreamSynchronize(a) != hipSuccess) return false;
    const double t2 = probe_now_us();
    if (hipStreamSynchronize(b) != hipSuccess) return false;
    const double t3 = probe_now_us();
    if (t2 - t1 < best) best = t2 - t1;
    if (t3 - t0 < lbest) lbest = t3 - t0;
  }
  *tiny_us = best;
  *long_us = lbest;
  return hipGetLastError() == hipSuccess;
}


// The low-priority side stream of the two-stream sweeps: (re)chosen, by the probe above, the first time a sweep needs it with
// the stream the context currently works on -- four low-priority candidates (the context's own side stream among them), the
// first whose queue does not share a pipe with the chain's stream wins.  ~7 ms once per (context, stream); G3_PROBE=0 keeps
// whatever the runtime dealt.
int g3i_ensure_side_stream(g3_ctx* ctx) {
  if (ctx->side_stream && (ctx->side_for == ctx->stream || !ctx->tune.probe)) return G3_OK;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (!ctx->side_stream) G3_HIP(hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, lo));
  ctx->side_for = ctx->stream;
  if (!ctx->tune.probe) return G3_OK;
  const int NL = 4;
  hipStream_t L[NL] = {ctx->side_stream, nullptr, nullptr, nullptr};
  for (int i = 1; i < NL; ++i)
    if (hipStreamCreateWithPriority(&L[i], hipStreamNonBlocking, lo) != hipSuccess) L[i] = nullptr;
  double t[NL], base = 1e30, tl = 0;
  bool ok = true;
  unsigned* scratch = (unsigned*)(ctx->d_stats + 63);   // (the last word of the reduction scratch: no reduction uses it)
  for (int i = 0; i < NL && ok; ++i) {
    t[i] = 1e30;
    if (!L[i]) continue;
    ok = g3i_probe_pair(ctx->stream, L[i], scratch, &t[i], &tl, 2);
    if (t[i] < base) base = t[i];
  }
  int pick = 0;
  if (ok) {
    const double limit = base * 2.0 > base + 40.0 ? base * 2.0 : base + 40.0;
    for (int i = 0; i < NL; ++i)
      if (L[i] && t[i] <= limit) { pick = i; break; }
    if (ctx->tune.probe > 1)
      fprintf(stderr, "libg3hip placement: side stream candidates %.0f %.0f %.0f %.0f us beside the chain's stream -> %d\n", t[0], t[1], t[2], t[3], pick);
  }
  for (int i = 0; i < NL; ++i)
    if (L[i] && i != pick) (void)hipStreamDestroy(L[i]);
  ctx->side_stream = L[pick];
  return G3_OK;
}

static int ctx_create_impl(int device, hipStream_t on_stream, g3_ctx** out) {
  if (!out) return -2;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return G3_ERR_HIP;
  if (device < 0 || device >= ndev) return -1;
  g3_ctx* ctx = new (std::nothrow) g3_ctx();
  if (!ctx) return G3_ERR_NOMEM;
  memset(ctx, 0, sizeof(*ctx));
  ctx->device = device;
  int prev_dev = -1;                       // like every other entry: the caller's current device is restored
  if (hipGetDevice(&prev_dev) != hipSuccess) prev_dev = -1;
  hipError_t e = hipSetDevice(device);
  int lo = 0, hi = 0;
  if (e == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least priority, hi = greatest
  // the chain runs on the context's stream at the greatest priority, the bulk updates on the side stream at the least:
  // measured, a bulk stream WITHOUT the low priority costs 2 % (N = 8192) to 10 % (N = 32768) of the step
  if (!on_stream) {
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, lo);
  }
  ctx->tune = g3h_tune_from_env();
  {
    const char* lg = getenv("G3_GEMM_LOG");
    ctx->gemm_log = (lg && *lg) ? fopen(lg, "a") : nullptr;
  }
#ifdef G3_CHAIN_SERVER   // measurement variant (scripts/variants/chain_server.inc): G3_CHAIN=1 puts the chain of a sweep on resident workgroups
  ctx->chain_wgs = g3h_env_int("G3_CHAIN", 0) ? g3h_env_int("G3_CHAIN_WGS", 16) : 0;
  ctx->chain_lds = g3h_env_int("G3_CHAIN_LDS", 0);
  ctx->chain_min_n = g3h_env_int("G3_CHAIN_MIN_N", 0);
  ctx->chain_max_n = g3h_env_int("G3_CHAIN_MAX_N", 10240);
#endif
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_info, G3_MAX_BATCH * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_stats, 64 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_prog, G3_PROG_SLOTS * sizeof(g3_kernel_prog));
  if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_info, G3_MAX_BATCH * sizeof(int), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_stats, 64 * sizeof(double), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_prog, G3_PROG_SLOTS * sizeof(g3_kernel_prog), hipHostMallocDefault);
  for (int i = 0; i < G3_PROG_SLOTS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ctx->prog_ev[i], hipEventDisableTiming);
  ctx->prog_last = -1;
  if (e == hipSuccess) e = hipMemset(ctx->d_info, 0, G3_MAX_BATCH * sizeof(int));
  if (e != hipSuccess) {
    g3_ctx_destroy(ctx);
    if (prev_dev >= 0 && prev_dev != device) (void)hipSetDevice(prev_dev);
    return G3_ERR_HIP;
  }
  ctx->stream = on_stream ? on_stream : ctx->own_stream;
  ctx->adopted = on_stream != nullptr;
  *out = ctx;
  if (prev_dev >= 0 && prev_dev != device) (void)hipSetDevice(prev_dev);
  return G3_OK;
}

extern "C" int g3_ctx_destroy(g3_ctx* ctx) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_info) (void)hipFree(ctx->d_info);
  if (ctx->d_stats) (void)hipFree(ctx->d_stats);
  if (ctx->d_prog) (void)hipFree(ctx->d_prog);
  for (int i = 0; i < G3_PROG_SLOTS; ++i) if (ctx->prog_ev[i]) (void)hipEventDestroy(ctx->prog_ev[i]);
  if (ctx->h_info) (void)hipHostFree(ctx->h_info);
  if (ctx->h_stats) (void)hipHostFree(ctx->h_stats);
  if (ctx->h_prog) (void)hipHostFree(ctx->h_prog);
  if (ctx->invd) (void)hipFree(ctx->invd);
  if (ctx->work) (void)hipFree(ctx->work);
  if (ctx->bbuf) (void)hipFree(ctx->bbuf);
  if (ctx->prof_ev) {
    for (int i = 0; i < ctx->prof_cap; ++i) if (ctx->prof_ev[i]) (void)hipEventDestroy(ctx->prof_ev[i]);
    free(ctx->prof_ev);
  }
  if (ctx->prof_rec) free(ctx->prof_rec);
  if (ctx->la_ev) {
    for (int i = 0; i < ctx->la_nev; ++i) (void)hipEventDestroy(ctx->la_ev[i]);
    free(ctx->la_ev);
  }
#ifdef G3_CHAIN_SERVER
  for (hipStream_t* st : {&ctx->chain_stream, &ctx->chain_stream2, &ctx->chain_sA, &ctx->chain_sB})
    if (*st) {
      (void)hipStreamSynchronize(*st);
      (void)hipStreamDestroy(*st);
    }
  for (hipEvent_t* ev : {&ctx->chain_ev, &ctx->chain_ev2, &ctx->chain_ev3})
    if (*ev) (void)hipEventDestroy(*ev);
  if (ctx->chain_ctl) (void)hipFree(ctx->chain_ctl);
#endif
  if (ctx->gemm_log) fclose(ctx->gemm_log);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return G3_OK;
}

// A stream the context has worked on is about to be destroyed by its owner (the multi-GPU driver's chain stream): the caller
// has synchronised it; forget every reference -- the program ring records its events on the stream a slot was consumed on
void g3i_ctx_forget_stream(g3_ctx* ctx, hipStream_t s) {
  if (!ctx || !s) return;
  if (ctx->prog_stream == s) {
    for (int i = 0; i < G3_PROG_SLOTS; ++i) ctx->prog_busy[i] = false;
    ctx->prog_last = -1;
    ctx->prog_stream = nullptr;
  }
  if (ctx->info_stream == s) {
    ctx->info_clean = false;
    ctx->info_stream = nullptr;
  }
  if (ctx->side_for == s) ctx->side_for = nullptr;
}

extern "C" int g3_ctx_set_stream(g3_ctx* ctx, void* s) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  G3_HIP(hipStreamSynchronize(ctx->stream));
  // the old stream is drained: every slot of the program ring is free, and nothing may be recorded on (or
  // waited for from) a stream the caller is about to destroy
  for (int i = 0; i < G3_PROG_SLOTS; ++i) ctx->prog_busy[i] = false;
  ctx->prog_last = -1;
  ctx->prog_stream = nullptr;
  ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
  ctx->adopted = (s != nullptr);
  return G3_OK;
}

// ----------------------------------------------------------------------------- profiling
extern "C" int g3_prof_enable(g3_ctx* ctx, int on) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (on && !ctx->prof_ev) {
    ctx->prof_cap = 16384;
    ctx->prof_ev = (hipEvent_t*)calloc(ctx->prof_cap, sizeof(hipEvent_t));
    ctx->prof_rec = (decltype(ctx->prof_rec))calloc(ctx->prof_cap / 2, sizeof(*ctx->prof_rec));
    if (!ctx->prof_ev || !ctx->prof_rec) return G3_ERR_NOMEM;
    for (int i = 0; i < ctx->prof_cap; ++i) G3_HIP(hipEventCreate(&ctx->prof_ev[i]));
  }
  ctx->prof_on = on != 0;
  ctx->prof_level = on;
  return G3_OK;
}
extern "C" int g3_prof_reset(g3_ctx* ctx) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  G3_HIP(hipStreamSynchronize(ctx->stream));
  ctx->prof_n = 0;
  ctx->prof_nrec = 0;
  return G3_OK;
}
int g3i_prof_begin(g3_ctx* ctx, int tag, double work) {
  if (!ctx->prof_on || ctx->prof_n + 2 > ctx->prof_cap) return -1;
  // level 1 (default): phases and the large GEMM launches only; level 2 adds every small launch
  const bool minor = tag == G3_TAG_GEMM_SMALL || tag == G3_TAG_LEAF || tag == G3_TAG_GEMM_MID;
  if (ctx->prof_level < 2 && minor) return -1;
  // level 3: like 2, but only every 16th of the small launches is timed (an unbiased sample that
  // keeps the event traffic out of launch-bound loops such as the multi-GPU driver's)
  if (ctx->prof_level == 3 && minor && (ctx->prof_skip++ & 15) != 0) return -1;
  const int r = ctx->prof_nrec++;
  ctx->prof_rec[r].e0 = ctx->prof_n++;
  ctx->prof_rec[r].e1 = ctx->prof_n++;
  ctx->prof_rec[r].tag = tag;
  ctx->prof_rec[r].work = work;
  (void)hipEventRecord(ctx->prof_ev[ctx->prof_rec[r].e0], ctx->stream);
  return r;
}
void g3i_prof_end(g3_ctx* ctx, int rec) {
  if (rec < 0) return;
  (void)hipEventRecord(ctx->prof_ev[ctx->prof_rec[rec].e1], ctx->stream);
}
extern "C" int g3_prof_collect(g3_ctx* ctx, double* out) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!out) return -2;
  G3_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 3 * G3_PROF_NTAGS; ++i) out[i] = 0.0;
  for (int r = 0; r < ctx->prof_nrec; ++r) {
    float ms = 0.f;
    G3_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[ctx->prof_rec[r].e0], ctx->prof_ev[ctx->prof_rec[r].e1]));
    const int t = ctx->prof_rec[r].tag;
    out[3 * t] += 1.0;
    out[3 * t + 1] += (double)ms;
    out[3 * t + 2] += ctx->prof_rec[r].work;
  }
  return G3_OK;
}

extern "C" int g3_ctx_sync(g3_ctx* ctx) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  G3_HIP(hipStreamSynchronize(ctx->stream));
  return G3_OK;
}

extern "C" const char* g3_last_error(g3_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int g3_malloc(g3_ctx* ctx, size_t bytes, void** dev) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!dev) return -3;
  *dev = nullptr;
  if (bytes == 0) return G3_OK;
  G3_HIP(hipSetDevice(ctx->device));
  G3_HIP(hipMalloc(dev, bytes));
  return G3_OK;
}
extern "C" int g3_free(g3_ctx* ctx, void* dev) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!dev) return G3_OK;
  G3_HIP(hipStreamSynchronize(ctx->stream));
  G3_HIP(hipFree(dev));
  return G3_OK;
}
extern "C" int g3_memcpy_h2d(g3_ctx* ctx, void* dev, const void* host, size_t bytes) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (bytes == 0) return G3_OK;
  if (!dev) return -2;
  if (!host) return -3;
  G3_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  G3_HIP(hipStreamSynchronize(ctx->stream));  // host buffer is borrowed only for the call
  return G3_OK;
}
extern "C" int g3_memcpy_d2h(g3_ctx* ctx, void* host, const void* dev, size_t bytes) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (bytes == 0) return G3_OK;
  if (!host) return -2;
  if (!dev) return -3;
  G3_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  G3_HIP(hipStreamSynchronize(ctx->stream));
  return G3_OK;
}
extern "C" int g3_memcpy_d2d(g3_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (bytes == 0) return G3_OK;
  if (!dst) return -2;
  if (!src) return -3;
  G3_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return G3_OK;
}
extern "C" int g3_memset(g3_ctx* ctx, void* dev, int byte, size_t bytes) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (bytes == 0) return G3_OK;
  if (!dev) return -2;
  G3_HIP(hipMemsetAsync(dev, byte, bytes, ctx->stream));
  return G3_OK;
}
// 16 bytes per thread, rows dealt over grid.y: the runtime's rectangular copy moves an 8 MB block at 0.5 TB/s
// (30 us, measured in the multi-GPU driver's trace), this one is bound by HBM
__global__ void __launch_bounds__(256) copy2d_kernel(char* __restrict__ dst, size_t dpitch, const char* __restrict__ src, size_t spitch,
                                                     size_t row_bytes, int64_t rows) {
  const size_t c = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (c >= row_bytes) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y)
    *reinterpret_cast<uint4*>(dst + r * dpitch + c) = *reinterpret_cast<const uint4*>(src + r * spitch + c);
}

extern "C" int g3_copy2d(g3_ctx* ctx, void* dst, int64_t ldd, const void* src, int64_t lds,
                         int64_t rows, int64_t cols, g3_dtype dt) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (rows == 0 || cols == 0) return G3_OK;
  if (!dst) return -2;
  if (!src) return -4;
  if (rows < 0) return -6;
  if (cols < 0 || cols > ldd || cols > lds) return -7;
  const size_t es = g3_esize(dt);
  const size_t rb = (size_t)cols * es, dp = (size_t)ldd * es, sp = (size_t)lds * es;
  if (((rb | dp | sp | (uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
    const unsigned gx = (unsigned)((rb / 16 + 255) / 256);
    int64_t gy = 8192 / gx;
    gy = gy < 1 ? 1 : (gy > rows ? rows : gy);
    hipLaunchKernelGGL(copy2d_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, ctx->stream, (char*)dst, dp, (const char*)src, sp, rb, rows);
    G3_LAUNCH_CHECK();
    return G3_OK;
  }
  G3_HIP(hipMemcpy2DAsync(dst, dp, src, sp, rb, (size_t)rows, hipMemcpyDeviceToDevice, ctx->stream));
  return G3_OK;
}

// ----------------------------------------------------------------------------- small kernels
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T u = __shfl_down(v, o, 64);
    v = (u < v || u != u) ? u : v;  // NaN propagates like numpy.min
  }
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T u = __shfl_down(v, o, 64);
    v = (u > v || u != u) ? u : v;
  }
  return v;
}

// one workgroup (1024 threads): out = [min, mean, max] of diag(A); optionally applies
// tt_to_cov's lift  A_ii += (1e-6f - min) when min <= 0  (tensors.py:95-98)
template <typename T>
__global__ void __launch_bounds__(1024)
diag_stats_kernel(T* A, int64_t n, int64_t ld, double* out, int lift, int64_t bstride) {
  A += (int64_t)blockIdx.x * bstride;            // batch member (grid.x)
  if (out) out += 3 * blockIdx.x;
  __shared__ double s_min[16], s_max[16], s_sum[16];
  __shared__ double s_m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double mn = 1.0 / 0.0, mx = -1.0 / 0.0, sm = 0.0;
  bool first = true;
  for (int64_t i = tid; i < n; i += 1024) {
    const double v = (double)A[i * ld + i];
    if (first) { mn = v; mx = v; first = false; }
    else { mn = (v < mn || v != v) ? v : mn; mx = (v > mx || v != v) ? v : mx; }
    sm += v;
  }
  mn = wave_min(mn); mx = wave_max(mx); sm = wave_sum(sm);
  if (lane == 0) { s_min[wave] = mn; s_max[wave] = mx; s_sum[wave] = sm; }
  __syncthreads();
  if (tid == 0) {
    double a = s_min[0], b = s_max[0], c = 0.0;
    for (int w = 0; w < 16; ++w) {
      a = (s_min[w] < a || s_min[w] != s_min[w]) ? s_min[w] : a;
      b = (s_max[w] > b || s_max[w] != s_max[w]) ? s_max[w] : b;
      c += s_sum[w];
    }
    if (out) { out[0] = a; out[1] = c / (double)n; out[2] = b; }
    s_m = a;
  }
  __syncthreads();
  if (lift) {
    const T m = (T)s_m;
    if (!(m > T(0))) {
      const T add = (T)1e-6f - m;
      for (int64_t i = tid; i < n; i += 1024) A[i * ld + i] += add;
    }
  }
}

template <typename T>
__global__ void diag_add_kernel(T* A, int64_t n, int64_t ld, T v, int64_t bstride) {
  A += (int64_t)blockIdx.y * bstride;            // batch member (grid.y)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[i * ld + i] += v;
}

template <typename T>
__global__ void scrub_kernel(T* A, int64_t n2, int64_t ld) {
  const int64_t i = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n2; j += (int64_t)gridDim.x * blockDim.x) {
    T v = A[i * ld + j];
    if (v != v) A[i * ld + j] = T(0);
    else if (__builtin_isinf(v)) A[i * ld + j] = (T)1e10f;
  }
}

// out[0] = sum log L_ii ; out[1] = sum a_i^2 ; out[2] = #non-finite a ; out[3] = #bad diag
template <typename T>
__global__ void __launch_bounds__(1024)
logp_terms_kernel(const T* L, int64_t n, int64_t ld, const T* a, double* out, int64_t bstride, int64_t astride) {
  L += (int64_t)blockIdx.x * bstride;            // batch member (grid.x)
  if (a) a += (int64_t)blockIdx.x * astride;
  out += 4 * blockIdx.x;
  __shared__ double s0[16], s1[16], s2[16], s3[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double ld_sum = 0, ss = 0, nf = 0, bd = 0;
  for (int64_t i = tid; i < n; i += 1024) {
    const double d = (double)L[i * ld + i];
    ld_sum += log(d);
    if (!(d > 0.0) || __builtin_isinf(d)) bd += 1;
    if (a) {
      const double v = (double)a[i];
      ss += v * v;
      if (v != v || __builtin_isinf(v)) nf += 1;
    }
  }
  ld_sum = wave_sum(ld_sum); ss = wave_sum(ss); nf = wave_sum(nf); bd = wave_sum(bd);
  if (lane == 0) { s0[wave] = ld_sum; s1[wave] = ss; s2[wave] = nf; s3[wave] = bd; }
  __syncthreads();
  if (tid == 0) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int w = 0; w < 16; ++w) { a0 += s0[w]; a1 += s1[w]; a2 += s2[w]; a3 += s3[w]; }
    out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
  }
}

// The tail of one evaluation in ONE launch (small problems are bound by the number of launches, ~5 us each):
// a_dst <- a_src (the solved right-hand-side row, npad entries), the four scalars of logp_terms_kernel over its
// first n entries, and the pivot flag handed over and cleared: out[4] = *info, *info = 0.
template <typename T>
__global__ void __launch_bounds__(1024)
logp_finish_kernel(const T* L, int64_t n, int64_t npad, int64_t ld, const T* __restrict__ a_src, T* __restrict__ a_dst,
                   double* out, int* info) {
  __shared__ double s0[16], s1[16], s2[16], s3[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double ld_sum = 0, ss = 0, nf = 0, bd = 0;
  for (int64_t i = tid; i < npad; i += 1024) {
    const T av = a_src[i];
    a_dst[i] = av;
    if (i < n) {
      const double d = (double)L[i * ld + i];
      ld_sum += log(d);
      if (!(d > 0.0) || __builtin_isinf(d)) bd += 1;
      const double v = (double)av;
      ss += v * v;
      if (v != v || __builtin_isinf(v)) nf += 1;
    }
  }
  ld_sum = wave_sum(ld_sum); ss = wave_sum(ss); nf = wave_sum(nf); bd = wave_sum(bd);
  if (lane == 0) { s0[wave] = ld_sum; s1[wave] = ss; s2[wave] = nf; s3[wave] = bd; }
  __syncthreads();
  if (tid == 0) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int w = 0; w < 16; ++w) { a0 += s0[w]; a1 += s1[w]; a2 += s2[w]; a3 += s3[w]; }
    out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
    out[4] = (double)*info;
    *info = 0;
  }
}

// one workgroup (256 threads) per row of V: dot with a, and sum of squares
template <typename T>
__global__ void __launch_bounds__(256)
rows_dot_ss_kernel(const T* __restrict__ V, int64_t n, int64_t ld, const T* __restrict__ a,
                   T* __restrict__ dot, T* __restrict__ ss, int64_t vstride, int64_t astride, int64_t ostride) {
  __shared__ double sd[4], sq[4];
  // grid.y = batch member: V, a and the outputs vstride / astride / ostride elements apart (0, 0, 0 for one problem)
  V += (int64_t)blockIdx.y * vstride;
  if (a) a += (int64_t)blockIdx.y * astride;
  if (dot) dot += (int64_t)blockIdx.y * ostride;
  if (ss) ss += (int64_t)blockIdx.y * ostride;
  const T* v = V + (int64_t)blockIdx.x * ld;
  double d = 0, q = 0;
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    const double x = (double)v[j];
    if (a) d += x * (double)a[j];
    q += x * x;
  }
  d = wave_sum(d); q = wave_sum(q);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sd[wave] = d; sq[wave] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (dot) dot[blockIdx.x] = (T)(sd[0] + sd[1] + sd[2] + sd[3]);
    if (ss) ss[blockIdx.x] = (T)(sq[0] + sq[1] + sq[2] + sq[3]);
  }
}

static int fetch_stats(g3_ctx* ctx, double* out, int cnt) {
  G3_HIP(hipMemcpyAsync(ctx->h_stats, ctx->d_stats, cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  G3_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < cnt; ++i) out[i] = ctx->h_stats[i];
  return G3_OK;
}

static int diag_stats_launch(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, double* dout, int lift) {
  if (dt == G3_F64)
    hipLaunchKernelGGL((diag_stats_kernel<double>), dim3(1), dim3(1024), 0, ctx->stream, (double*)A, n, ld, dout, lift, (int64_t)0);
  else
    hipLaunchKernelGGL((diag_stats_kernel<float>), dim3(1), dim3(1024), 0, ctx->stream, (float*)A, n, ld, dout, lift, (int64_t)0);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

extern "C" int g3_cov_lift(g3_ctx* ctx, void* K, int64_t n, int64_t ld, g3_dtype dt) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!K) return -2;
  if (n < 0) return -3;
  if (ld < n) return -4;
  if (n == 0) return G3_OK;
  return diag_stats_launch(ctx, K, n, ld, dt, nullptr, 1);
}

extern "C" int g3_diag_stats(g3_ctx* ctx, const void* A, int64_t n, int64_t ld, g3_dtype dt, double out[3]) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!A) return -2;
  if (n <= 0) return -3;
  if (ld < n) return -4;
  if (!out) return -6;
  int rc = diag_stats_launch(ctx, const_cast<void*>(A), n, ld, dt, ctx->d_stats + 8, 0);
  if (rc) return rc;
  G3_HIP(hipMemcpyAsync(ctx->h_stats + 8, ctx->d_stats + 8, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  G3_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 3; ++i) out[i] = ctx->h_stats[8 + i];
  return G3_OK;
}

int g3i_diag_stats_dev(g3_ctx* ctx, const void* A, int64_t n, int64_t ld, g3_dtype dt, double* out_dev) {
  return diag_stats_launch(ctx, const_cast<void*>(A), n, ld, dt, out_dev, 0);
}

extern "C" int g3_diag_add(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, double value) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!A) return -2;
  if (n < 0) return -3;
  if (ld < n) return -4;
  if (n == 0) return G3_OK;
  return g3i_diag_add(ctx, A, n, ld, dt, value);
}

template <typename T>
__global__ void scale_kernel(T* A, int64_t rows, int64_t cols, int64_t ld, T f) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j < cols && i < rows) A[i * ld + j] *= f;
}
int g3i_scale(g3_ctx* ctx, void* A, int64_t rows, int64_t cols, int64_t ld, g3_dtype dt, double factor) {
  if (rows <= 0 || cols <= 0) return G3_OK;
  const dim3 grid((unsigned)((cols + 255) / 256), (unsigned)rows);
  if (dt == G3_F64) hipLaunchKernelGGL((scale_kernel<double>), grid, dim3(256), 0, ctx->stream, (double*)A, rows, cols, ld, factor);
  else hipLaunchKernelGGL((scale_kernel<float>), grid, dim3(256), 0, ctx->stream, (float*)A, rows, cols, ld, (float)factor);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

// stream-ordered; in batch mode every member gets the same increment
int g3i_diag_add(g3_ctx* ctx, void* A, int64_t n, int64_t ld, g3_dtype dt, double value) {
  const dim3 grid((unsigned)((n + 255) / 256), (unsigned)g3_nbatch(ctx));
  if (dt == G3_F64)
    hipLaunchKernelGGL((diag_add_kernel<double>), grid, dim3(256), 0, ctx->stream, (double*)A, n, ld, value,
                       g3_bstride_of(ctx, A));
  else
    hipLaunchKernelGGL((diag_add_kernel<float>), grid, dim3(256), 0, ctx->stream, (float*)A, n, ld, (float)value,
                       g3_bstride_of(ctx, A));
  G3_LAUNCH_CHECK();
  return G3_OK;
}

extern "C" int g3_scrub(g3_ctx* ctx, void* A, int64_t n1, int64_t n2, int64_t ld, g3_dtype dt) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!A) return -2;
  if (n1 < 0) return -3;
  if (n2 < 0) return -4;
  if (ld < n2) return -5;
  if (n1 == 0 || n2 == 0) return G3_OK;
  const dim3 grid((unsigned)((n2 + 255) / 256 > 64 ? 64 : (n2 + 255) / 256), (unsigned)n1);
  if (dt == G3_F64)
    hipLaunchKernelGGL((scrub_kernel<double>), grid, dim3(256), 0, ctx->stream, (double*)A, n2, ld);
  else
    hipLaunchKernelGGL((scrub_kernel<float>), grid, dim3(256), 0, ctx->stream, (float*)A, n2, ld);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

static int logp_terms_launch(g3_ctx* ctx, const void* L, int64_t n, int64_t ld, const void* a, g3_dtype dt) {
  if (dt == G3_F64)
    hipLaunchKernelGGL((logp_terms_kernel<double>), dim3(1), dim3(1024), 0, ctx->stream, (const double*)L, n, ld,
                       (const double*)a, ctx->d_stats, (int64_t)0, (int64_t)0);
  else
    hipLaunchKernelGGL((logp_terms_kernel<float>), dim3(1), dim3(1024), 0, ctx->stream, (const float*)L, n, ld,
                       (const float*)a, ctx->d_stats, (int64_t)0, (int64_t)0);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

int g3i_logp_terms_dev(g3_ctx* ctx, const void* L, int64_t n, int64_t ld, const void* a, g3_dtype dt, double* out_dev) {
  if (dt == G3_F64)
    hipLaunchKernelGGL((logp_terms_kernel<double>), dim3(1), dim3(1024), 0, ctx->stream, (const double*)L, n, ld,
                       (const double*)a, out_dev, (int64_t)0, (int64_t)0);
  else
    hipLaunchKernelGGL((logp_terms_kernel<float>), dim3(1), dim3(1024), 0, ctx->stream, (const float*)L, n, ld,
                       (const float*)a, out_dev, (int64_t)0, (int64_t)0);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

extern "C" int g3_logp_terms(g3_ctx* ctx, const void* L, int64_t n, int64_t ld, const void* a,
                             g3_dtype dt, double out[4]) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!L) return -2;
  if (n <= 0) return -3;
  if (ld < n) return -4;
  if (!out) return -7;
  int rc = logp_terms_launch(ctx, L, n, ld, a, dt);
  if (rc) return rc;
  return fetch_stats(ctx, out, 4);
}

// batch of SMALL problems (a chain of hyper-parameter vectors at N <= 1024): one workgroup per member walks its m rows, a
// wave per row -- m x batch workgroups of one row each cost more in dispatch than in arithmetic (0.58 ms for 4096 x 128 rows)
template <typename T>
__global__ void __launch_bounds__(256)
rows_dot_ss_member_kernel(const T* __restrict__ V, int64_t m, int64_t n, int64_t ld, const T* __restrict__ a,
                          T* __restrict__ dot, T* __restrict__ ss, int64_t vstride, int64_t astride, int64_t ostride) {
  V += (int64_t)blockIdx.x * vstride;
  if (a) a += (int64_t)blockIdx.x * astride;
  if (dot) dot += (int64_t)blockIdx.x * ostride;
  if (ss) ss += (int64_t)blockIdx.x * ostride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = wave; r < m; r += 4) {
    const T* v = V + r * ld;
    double d = 0, q = 0;
    for (int64_t j = lane; j < n; j += 64) {
      const double x = (double)v[j];
      if (a) d += x * (double)a[j];
      q += x * x;
    }
    d = wave_sum(d); q = wave_sum(q);
    if (lane == 0) {
      if (dot) dot[r] = (T)d;
      if (ss) ss[r] = (T)q;
    }
  }
}

static int rows_dot_ss_launch(g3_ctx* ctx, const void* V, int64_t m, int64_t n, int64_t ld, const void* a,
                              g3_dtype dt, void* dot, void* ss, int batch = 1, int64_t vstride = 0, int64_t astride = 0,
                              int64_t ostride = 0) {
  if (m == 0) return G3_OK;
  if (batch > 1 && m <= 1024) {
    if (dt == G3_F64)
      hipLaunchKernelGGL((rows_dot_ss_member_kernel<double>), dim3((unsigned)batch), dim3(256), 0, ctx->stream,
                         (const double*)V, m, n, ld, (const double*)a, (double*)dot, (double*)ss, vstride, astride, ostride);
    else
      hipLaunchKernelGGL((rows_dot_ss_member_kernel<float>), dim3((unsigned)batch), dim3(256), 0, ctx->stream,
                         (const float*)V, m, n, ld, (const float*)a, (float*)dot, (float*)ss, vstride, astride, ostride);
    G3_LAUNCH_CHECK();
    return G3_OK;
  }
  if (dt == G3_F64)
    hipLaunchKernelGGL((rows_dot_ss_kernel<double>), dim3((unsigned)m, (unsigned)batch), dim3(256), 0, ctx->stream,
                       (const double*)V, n, ld, (const double*)a, (double*)dot, (double*)ss, vstride, astride, ostride);
  else
    hipLaunchKernelGGL((rows_dot_ss_kernel<float>), dim3((unsigned)m, (unsigned)batch), dim3(256), 0, ctx->stream,
                       (const float*)V, n, ld, (const float*)a, (float*)dot, (float*)ss, vstride, astride, ostride);
  G3_LAUNCH_CHECK();
  return G3_OK;
}

int g3i_rows_dot_ss_batched(g3_ctx* ctx, const void* V, int64_t m, int64_t n, int64_t ld, const void* a, g3_dtype dt, void* dot,
                            void* ss, int batch, int64_t vstride, int64_t astride, int64_t ostride) {
  return rows_dot_ss_launch(ctx, V, m, n, ld, a, dt, dot, ss, batch, vstride, astride, ostride);
}

extern "C" int g3_rows_dot_ss(g3_ctx* ctx, const void* V, int64_t m, int64_t n, int64_t ld,
                              const void* a, g3_dtype dt, void* dot, void* ss) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!V) return -2;
  if (m < 0) return -3;
  if (n < 0) return -4;
  if (ld < n) return -5;
  return rows_dot_ss_launch(ctx, V, m, n, ld, a, dt, dot, ss);
}

// ----------------------------------------------------------------------------- fused path
template <typename T>
__global__ void pad_row_kernel(T* dst, int64_t ld, const T* src, int64_t n, int64_t npad, int64_t rows,
                               int64_t dstride, int64_t sstride) {
  // dst is rows x npad (row stride ld): row 0 = [src, 0...], other rows 0; grid.y = batch member
  dst += (int64_t)blockIdx.y * dstride;
  src += (int64_t)blockIdx.y * sstride;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= npad) return;
  for (int64_t r = 0; r < rows; ++r) dst[r * ld + j] = (r == 0 && j < n) ? src[j] : T(0);
}

// Shared implementation of g3_gp_factor / g3_gp_factor_predict.  K_dev is a tall buffer:
//   rows [0, Np)            the covariance, then its factor (lower triangle)
//   rows [Np, Np+128)       right-hand-side block whose first row is delta -> a = L^-1 delta
//   rows [Np+128, +Mp)      K(Xs, X) -> V = K(Xs, X) L^-T            (only when M > 0)
// The right-hand-side rows are carried through the factorisation itself (g3i_potrf_tall): the
// panel solves and trailing updates that factor K also perform the forward substitutions.
static int gp_factor_impl(g3_ctx* ctx, const g3_kernel_prog* prog, const void* X, int64_t N, int64_t ldx, int d,
                          const void* delta, g3_dtype dt, void* K, int64_t ldk, void* invd, void* a, double out[6],
                          const g3_kernel_prog* prog_cross, const void* Xs, int64_t M, int64_t ldxs, void* mu,
                          void* ss) {
  const int64_t Np = g3_roundup(N, G3_LB), RB = 128, Mp = M > 0 ? g3_roundup(M, 128) : 0;
  const int64_t E = RB + Mp;
  const size_t es = g3_esize(dt);
  const double es_d = (double)es;
  char* rhs = (char*)K + (size_t)Np * ldk * es;            // delta block
  char* Vp = rhs + (size_t)RB * ldk * es;                  // cross-covariance rows
  int rc = G3_OK;
  const unsigned gflags = G3_GRAM_LOWER | G3_GRAM_SCRUB | G3_GRAM_PAD_EYE;
  auto build_rhs = [&]() -> int {
    const unsigned nb = (unsigned)((Np + 255) / 256);
    if (dt == G3_F64)
      hipLaunchKernelGGL((pad_row_kernel<double>), dim3(nb), dim3(256), 0, ctx->stream, (double*)rhs, ldk,
                         (const double*)delta, N, Np, RB, (int64_t)0, (int64_t)0);
    else
      hipLaunchKernelGGL((pad_row_kernel<float>), dim3(nb), dim3(256), 0, ctx->stream, (float*)rhs, ldk,
                         (const float*)delta, N, Np, RB, (int64_t)0, (int64_t)0);
    G3_LAUNCH_CHECK();
    if (M > 0) {
      // V = tt_to_num(cov(Xs, X))  (elliptical.py:78-79)
      const int pr = g3i_prof_begin(ctx, G3_TAG_CROSS_GRAM, (double)(N + M) * d * es_d + (double)N * M * es_d);
      int r = g3_gram(ctx, prog_cross, Xs, M, ldxs, X, N, ldx, d, dt, Vp, ldk, Mp, Np, G3_GRAM_SCRUB);
      g3i_prof_end(ctx, pr);
      if (r) return r;
    }
    return G3_OK;
  };
  // K = tt_to_cov(cov(X))  (elliptical.py:70-71), lower triangle only
  auto build = [&]() -> int {
    // algorithmic bytes of the lower-triangle Gram: N d s read + N(N+1)/2 s written
    const int pr = g3i_prof_begin(ctx, G3_TAG_GRAM, (double)N * d * es_d + 0.5 * (double)N * (N + 1) * es_d);
    int r = g3_gram(ctx, prog, X, N, ldx, nullptr, 0, 0, d, dt, K, ldk, Np, Np, gflags);
    if (r) return r;
    r = g3_cov_lift(ctx, K, N, ldk, dt);
    g3i_prof_end(ctx, pr);
    if (r) return r;
    return build_rhs();
  };
  auto factor = [&](int* info) -> int {
    const int pr = g3i_prof_begin(ctx, G3_TAG_POTRF, (double)N * N * N / 3.0 + (double)N * N * (1 + M));
    int r = g3i_potrf_tall(ctx, K, Np, ldk, dt, invd, E);
    g3i_prof_end(ctx, pr);
    if (r) return r;
    G3_HIP(hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    G3_HIP(hipStreamSynchronize(ctx->stream));
    *info = *ctx->h_info;
#ifdef G3_CHAIN_SERVER
    if (g3i_chain_gave_up(ctx, *info)) {   // (cannot happen twice: the server is off after the first time)
      snprintf(ctx->err, sizeof(ctx->err), "chain server gave up inside a jitter retry");
      return G3_ERR_HIP;
    }
#endif
    return G3_OK;
  };
  // scalars of the evaluation: log-determinant, quadratic form, guards (and mean / sum of squares per query)
  double st4[5];
  auto finish = [&]() -> int {
    const int pr = g3i_prof_begin(ctx, G3_TAG_REDUCE, 2.0 * N * M);
    if (dt == G3_F64)
      hipLaunchKernelGGL((logp_finish_kernel<double>), dim3(1), dim3(1024), 0, ctx->stream, (const double*)K, N, Np, ldk,
                         (const double*)rhs, (double*)a, ctx->d_stats, ctx->d_info);
    else
      hipLaunchKernelGGL((logp_finish_kernel<float>), dim3(1), dim3(1024), 0, ctx->stream, (const float*)K, N, Np, ldk,
                         (const float*)rhs, (float*)a, ctx->d_stats, ctx->d_info);
    G3_LAUNCH_CHECK();
    int r = G3_OK;
    if (M > 0 && (mu || ss)) r = rows_dot_ss_launch(ctx, Vp, M, N, ldk, a, dt, mu, ss);
    g3i_prof_end(ctx, pr);
    if (r) return r;
    r = fetch_stats(ctx, st4, 5);        // the one host synchronisation of a successful evaluation
    if (!r) { ctx->info_clean = true; ctx->info_sync = true; }      // the kernel above left the pivot flag cleared, and the host has waited for it
    return r;
  };
  rc = build();
  if (rc) return rc;
  int info = 0;
  {
    // first attempt: the pivot flag comes back with the reductions -- one round trip per evaluation instead of two
    // (the reductions of a failed factorisation are simply discarded)
    const int pr = g3i_prof_begin(ctx, G3_TAG_POTRF, (double)N * N * N / 3.0 + (double)N * N * (1 + M));
    rc = g3i_potrf_tall(ctx, K, Np, ldk, dt, invd, E);
    g3i_prof_end(ctx, pr);
    if (rc) return rc;
    rc = finish();
    if (rc) return rc;
    info = (int)st4[4];
#ifdef G3_CHAIN_SERVER
    if (g3i_chain_gave_up(ctx, info)) {
      // the resident chain workgroups ran into their wall-clock limit (they are off from now on): this evaluation is
      // redone with the launch-per-kernel sweep
      rc = g3i_reset_info(ctx);
      if (!rc) rc = build();
      if (rc) return rc;
      const int pr2 = g3i_prof_begin(ctx, G3_TAG_POTRF, (double)N * N * N / 3.0 + (double)N * N * (1 + M));
      rc = g3i_potrf_tall(ctx, K, Np, ldk, dt, invd, E);
      g3i_prof_end(ctx, pr2);
      if (!rc) rc = finish();
      if (rc) return rc;
      info = (int)st4[4];
    }
#endif
  }
  double tries = 0, fallback = 0;
  const int info0 = info;
  if (info != 0) {
    // jitter schedule of CholeskyRobust._cholesky (tensors.py:203-213); the Gram is rebuilt
    // instead of kept: it costs one HBM pass, a copy would cost the same plus 8 N^2 bytes
    rc = build();
    if (rc) return rc;
    double st[3];
    rc = g3_diag_stats(ctx, K, N, ldk, dt, st);
    if (rc) return rc;
    G3hJitter jit(st[1], st[0]);
    bool ok = false;
    for (int t = 0; t < G3hJitter::max_tries(); ++t) {
      tries += 1;
      if (t > 0) {
        rc = build();
        if (rc) return rc;
      }
      rc = g3_diag_add(ctx, K, N, ldk, dt, jit.value());
      if (rc) return rc;
      rc = factor(&info);
      if (rc) return rc;
      if (info == 0) { ok = true; break; }
      jit.next();
    }
    if (!ok) {
      fallback = 1;
      // 1e-10 * I (tensors.py:221); right-hand sides are solved against it separately
      G3_HIP(hipMemset2DAsync(K, (size_t)ldk * es, 0, (size_t)Np * es, (size_t)Np, ctx->stream));
      rc = g3_diag_add(ctx, K, N, ldk, dt, (double)1e-10f);
      if (rc) return rc;
      if (Np > N) {
        rc = g3_diag_add(ctx, (char*)K + (size_t)N * (ldk + 1) * es, Np - N, ldk, dt, 1.0);
        if (rc) return rc;
      }
      rc = g3i_reset_info(ctx);
      if (rc) return rc;
      rc = g3i_trtri_blocks(ctx, K, Np, ldk, dt, invd);
      if (rc) return rc;
      rc = build_rhs();
      if (rc) return rc;
      rc = g3i_trsm_rlt(ctx, K, Np, ldk, rhs, E, ldk, dt, invd);
      if (rc) return rc;
    }
  }
  if (info0 != 0) {          // the jitter path produced a new factor: its scalars replace the discarded ones
    rc = finish();
    if (rc) return rc;
  }
  out[0] = st4[0];
  out[1] = st4[1];
  out[2] = st4[2];
  out[3] = tries;
  out[4] = fallback;
  out[5] = (double)info0;
  return G3_OK;
}

// ---------------------------------------------------------------------------------------
// Batched evaluation: the caller pattern of logp_chain / fixed_logp / find_MAP restarts
// (stochastic.py:515-564, 740-771: a Python loop, "TODO: Vectorized" in the reference) --
// many hyper-parameter vectors on the same inputs.  All `batch` covariances are built by one
// Gram launch (grid.z) and factored by ONE sweep whose GEMM / diagonal-block launches carry the
// batch in grid.y, so a problem too small to fill 256 CUs on its own still does.
static int same_structure(const g3_kernel_prog* a, const g3_kernel_prog* b) {
  if (a->nleaf != b->nleaf || a->nprod != b->nprod) return 0;
  for (int l = 0; l < a->nleaf; ++l) {
    if (a->leaf[l].kind != b->leaf[l].kind || a->leaf[l].ndims != b->leaf[l].ndims) return 0;
    for (int k = 0; k < a->leaf[l].ndims; ++k)
      if (a->leaf[l].dims[k] != b->leaf[l].dims[k]) return 0;
  }
  for (int q = 0; q < a->nprod; ++q) {
    if (a->prod[q].nfac != b->prod[q].nfac) return 0;
    for (int f = 0; f < a->prod[q].nfac; ++f)
      if (a->prod[q].fac[f] != b->prod[q].fac[f]) return 0;
  }
  return 1;
}

static int ensure_bbuf(g3_ctx* ctx, size_t bytes) {
  if (ctx->bbuf_bytes >= bytes) return G3_OK;
  if (ctx->bbuf) (void)hipFree(ctx->bbuf);
  ctx->bbuf = nullptr;
  ctx->bbuf_bytes = 0;
  G3_HIP(hipMalloc(&ctx->bbuf, bytes));
  ctx->bbuf_bytes = bytes;
  return G3_OK;
}

// Where the members' programs come from: `batch` complete host programs, or ONE template plus, per member, the doubles
// that differ (hyper-parameter values) and the byte offsets in g3_kernel_prog they go to -- a chain of 4096 members is
// then 4096 x nfield doubles to pack and to copy instead of 4096 x 6 KB.
struct MemberProgs {
  const g3_kernel_prog* progs = nullptr;
  const g3_kernel_prog* tmpl = nullptr;
  const double* fields = nullptr;
  const int32_t* offs = nullptr;
  int nfield = 0;
  void member(int b, g3_kernel_prog* out) const {
    if (progs) { *out = progs[b]; return; }
    *out = *tmpl;
    for (int i = 0; i < nfield; ++i) memcpy((char*)out + offs[i], &fields[(size_t)b * nfield + i], sizeof(double));
  }
};

__global__ void __launch_bounds__(256) expand_progs_kernel(g3_kernel_prog* dst, const g3_kernel_prog* tmpl,
                                                          const double* fields, const int32_t* offs, int nfield) {
  const uint32_t* s = (const uint32_t*)tmpl;
  uint32_t* o = (uint32_t*)(dst + blockIdx.x);
  for (unsigned i = threadIdx.x; i < sizeof(g3_kernel_prog) / 4; i += 256) o[i] = s[i];
  __syncthreads();
  for (int i = threadIdx.x; i < nfield; i += 256)
    *(double*)((char*)o + offs[i]) = fields[(size_t)blockIdx.x * nfield + i];
}

static int gp_factor_batched_impl(g3_ctx* ctx, const MemberProgs& mp, int batch, const void* X, int64_t N,
                                  int64_t ldx, int d, const void* delta, int64_t ldd, g3_dtype dt, void* K,
                                  int64_t ldk, int64_t kstride, void* invd, void* a, double* out) {
  if (batch < 1 || batch > G3_MAX_BATCH) return -3;
  if (!X) return -4;
  if (N <= 0) return -5;
  if (d < 1 || d > G3_MAXCOLS) return -7;
  if (ldx < d) return -6;
  if (!delta) return -8;
  if (ldd < N) return -9;
  if (!K) return -11;
  const int64_t Np = g3_roundup(N, G3_LB), RB = 128;
  const size_t es = g3_esize(dt);
  const int64_t al = 16 / (int64_t)es;
  if (ldk < Np || ldk % al) return -12;
  if (kstride < (Np + RB) * ldk || kstride % al) return -13;
  if (!invd) return -14;
  if (!a) return -15;
  if (!out) return -16;
  g3_kernel_prog first;
  mp.member(0, &first);
  if (mp.progs) {
    for (int b = 0; b < batch; ++b) {
      if (g3i_validate_prog(&mp.progs[b], d)) return -2;
      if (!same_structure(&mp.progs[0], &mp.progs[b])) return -2;
    }
  } else {
    if (g3i_validate_prog(mp.tmpl, d) || g3i_validate_prog(&first, d)) return -2;
  }
  const g3_kernel_prog* progs = &first;   // the structure every member shares
  int rc = G3_OK;
  if (batch == 1)
    return gp_factor_impl(ctx, progs, X, N, ldx, d, delta, dt, K, ldk, invd, a, out, nullptr, nullptr, 0, 0, nullptr,
                          nullptr);
  // device copies of the programs, then the per-member statistics
  const size_t pbytes = (size_t)batch * sizeof(g3_kernel_prog);
  const size_t sbytes = (size_t)batch * 4 * sizeof(double);
  const size_t fbytes = mp.progs ? 0 : (size_t)batch * mp.nfield * sizeof(double);
  const size_t obytes = mp.progs ? 0 : (((size_t)mp.nfield * sizeof(int32_t) + 15) & ~(size_t)15);
  const size_t cbytes = g3i_coop_ctl_bytes(batch);
  const size_t head = (pbytes + sbytes + (mp.progs ? 0 : sizeof(g3_kernel_prog)) + fbytes + obytes + 255) & ~(size_t)255;
  rc = ensure_bbuf(ctx, head + cbytes);
  if (rc) return rc;
  unsigned* coop_ctl = (unsigned*)((char*)ctx->bbuf + head);
  g3_kernel_prog* dprogs = (g3_kernel_prog*)ctx->bbuf;
  double* dstats = (double*)((char*)ctx->bbuf + pbytes);
  if (mp.progs) {
    G3_HIP(hipMemcpyAsync(dprogs, mp.progs, pbytes, hipMemcpyHostToDevice, ctx->stream));
  } else {
    g3_kernel_prog* dtmpl = (g3_kernel_prog*)((char*)dstats + sbytes);
    double* dfields = (double*)(dtmpl + 1);
    int32_t* doffs = (int32_t*)((char*)dfields + fbytes);
    G3_HIP(hipMemcpyAsync(dtmpl, mp.tmpl, sizeof(g3_kernel_prog), hipMemcpyHostToDevice, ctx->stream));
    if (mp.nfield) {
      G3_HIP(hipMemcpyAsync(dfields, mp.fields, fbytes, hipMemcpyHostToDevice, ctx->stream));
      G3_HIP(hipMemcpyAsync(doffs, mp.offs, (size_t)mp.nfield * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(expand_progs_kernel, dim3(batch), dim3(256), 0, ctx->stream, dprogs, dtmpl, dfields, doffs, mp.nfield);
    G3_LAUNCH_CHECK();
  }
  const int64_t wstride = Np * G3_LB;
  // K_b = tt_to_cov(cov(X)) (elliptical.py:70-71), right-hand-side block = [delta_b; 0]
  const unsigned nb = (unsigned)((Np + 255) / 256);
  char* rhs = (char*)K + (size_t)Np * ldk * es;
  const bool small = Np <= 2 * G3_LB;       // one workgroup per member does the whole evaluation (g3i_small_factor_batched)
  int pr = g3i_prof_begin(ctx, G3_TAG_GRAM, (double)batch * ((double)N * d + 0.5 * (double)N * (N + 1)) * es);
  rc = g3i_gram_batched(ctx, dprogs, progs, batch, X, N, ldx, d, dt, K, ldk, kstride, Np,
                        G3_GRAM_LOWER | G3_GRAM_SCRUB | G3_GRAM_PAD_EYE);
  if (rc) return rc;
  if (small) {
    if (dt == G3_F64)
      hipLaunchKernelGGL((diag_stats_kernel<double>), dim3(batch), dim3(1024), 0, ctx->stream, (double*)K, N, ldk, (double*)nullptr, 1, kstride);
    else
      hipLaunchKernelGGL((diag_stats_kernel<float>), dim3(batch), dim3(1024), 0, ctx->stream, (float*)K, N, ldk, (double*)nullptr, 1, kstride);
    g3i_prof_end(ctx, pr);
    G3_LAUNCH_CHECK();
    pr = g3i_prof_begin(ctx, G3_TAG_POTRF, (double)batch * ((double)N * N * N / 3.0 + (double)N * N));
    rc = g3i_small_factor_batched(ctx, K, ldk, kstride, invd, wstride, delta, ldd, a, Np, dstats, batch, N, Np, dt);
    g3i_prof_end(ctx, pr);
    if (!rc && hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int) * batch, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = G3_ERR_HIP;
    if (!rc && hipMemsetAsync(ctx->d_info, 0, sizeof(int) * batch, ctx->stream) != hipSuccess) rc = G3_ERR_HIP;   // every member's flag
    if (rc) return rc;
  } else if (dt == G3_F64) {
    hipLaunchKernelGGL((diag_stats_kernel<double>), dim3(batch), dim3(1024), 0, ctx->stream, (double*)K, N, ldk,
                       (double*)nullptr, 1, kstride);
    hipLaunchKernelGGL((pad_row_kernel<double>), dim3(nb, batch), dim3(256), 0, ctx->stream, (double*)rhs, ldk,
                       (const double*)delta, N, Np, RB, kstride, ldd);
  } else {
    hipLaunchKernelGGL((diag_stats_kernel<float>), dim3(batch), dim3(1024), 0, ctx->stream, (float*)K, N, ldk,
                       (double*)nullptr, 1, kstride);
    hipLaunchKernelGGL((pad_row_kernel<float>), dim3(nb, batch), dim3(256), 0, ctx->stream, (float*)rhs, ldk,
                       (const float*)delta, N, Np, RB, kstride, ldd);
  }
  if (!small) {
  g3i_prof_end(ctx, pr);
  G3_LAUNCH_CHECK();
  pr = g3i_prof_begin(ctx, G3_TAG_POTRF, (double)batch * ((double)N * N * N / 3.0 + (double)N * N));
  if (Np <= ctx->tune.coop_max_n && batch >= ctx->tune.coop_min_batch) {
    // long chains of medium members: a group of workgroups per member, the whole batch in ONE launch (g3_chainb.hip): 1.2 -
    // 1.3x the lock-step sweep below at N <= 512, 1.08x at 768 - 1024; short batches (< ~200 members) stay with the sweep,
    // whose launches are at least as wide as the batch (profiles/r05_chain_medium.txt)
    rc = g3i_coop_factor_batched(ctx, K, ldk, kstride, invd, wstride, coop_ctl, batch, Np, dt);
    g3i_prof_end(ctx, pr);
    if (!rc && hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int) * batch, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = G3_ERR_HIP;
    if (!rc && hipMemsetAsync(ctx->d_info, 0, sizeof(int) * batch, ctx->stream) != hipSuccess) rc = G3_ERR_HIP;   // every member's flag
    if (rc) return rc;
  } else {
    // one sweep factors every member; a member whose pivot fails only stops its own launches
    ctx->batch = batch;
    ctx->bstride = kstride;
    ctx->bstride_w = wstride;
    ctx->bw_base = (const char*)invd;
    ctx->bw_bytes = (size_t)batch * wstride * es;
    rc = g3i_potrf_tall(ctx, K, Np, ldk, dt, invd, RB);
    g3i_prof_end(ctx, pr);
    if (!rc && hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int) * batch, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = G3_ERR_HIP;
    if (!rc) rc = g3i_reset_info(ctx);
    ctx->batch = 0;
    ctx->bw_base = nullptr;
    if (rc) return rc;
  }
  // a_b = first row of the solved right-hand-side block; log det and a^T a per member
  G3_HIP(hipMemcpy2DAsync(a, (size_t)Np * es, rhs, (size_t)kstride * es, (size_t)Np * es, (size_t)batch,
                          hipMemcpyDeviceToDevice, ctx->stream));
  if (dt == G3_F64)
    hipLaunchKernelGGL((logp_terms_kernel<double>), dim3(batch), dim3(1024), 0, ctx->stream, (const double*)K, N, ldk,
                       (const double*)a, dstats, kstride, Np);
  else
    hipLaunchKernelGGL((logp_terms_kernel<float>), dim3(batch), dim3(1024), 0, ctx->stream, (const float*)K, N, ldk,
                       (const float*)a, dstats, kstride, Np);
  G3_LAUNCH_CHECK();
  }
  double* hst = (double*)malloc(sbytes);
  if (!hst) return G3_ERR_NOMEM;
  if (hipMemcpyAsync(hst, dstats, sbytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    free(hst);
    snprintf(ctx->err, sizeof(ctx->err), "g3_gp_factor_batched: statistics copy failed");
    return G3_ERR_HIP;
  }
  int* infos = (int*)malloc(sizeof(int) * batch);
  if (!infos) { free(hst); return G3_ERR_NOMEM; }
  memcpy(infos, ctx->h_info, sizeof(int) * batch);
  for (int b = 0; b < batch && !rc; ++b) {
    double* o = out + 6 * b;
    if (infos[b] == 0) {
      o[0] = hst[4 * b]; o[1] = hst[4 * b + 1]; o[2] = hst[4 * b + 2];
      o[3] = 0; o[4] = 0; o[5] = 0;
    } else {
      // CholeskyRobust's jitter schedule (tensors.py:203-222) for this member alone
      g3_kernel_prog mine;
      mp.member(b, &mine);
      rc = gp_factor_impl(ctx, &mine, X, N, ldx, d, (const char*)delta + (size_t)b * ldd * es, dt,
                          (char*)K + (size_t)b * kstride * es, ldk, (char*)invd + (size_t)b * wstride * es,
                          (char*)a + (size_t)b * Np * es, o, nullptr, nullptr, 0, 0, nullptr, nullptr);
    }
  }
  free(hst);
  free(infos);
  return rc;
}

extern "C" int g3_gp_factor_batched(g3_ctx* ctx, const g3_kernel_prog* progs, int batch, const void* X, int64_t N,
                                    int64_t ldx, int d, const void* delta, int64_t ldd, g3_dtype dt, void* K,
                                    int64_t ldk, int64_t kstride, void* invd, void* a, double* out) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!progs) return -2;
  MemberProgs mp;
  mp.progs = progs;
  return gp_factor_batched_impl(ctx, mp, batch, X, N, ldx, d, delta, ldd, dt, K, ldk, kstride, invd, a, out);
}

extern "C" int g3_gp_factor_batched_fields(g3_ctx* ctx, const g3_kernel_prog* tmpl, int batch, const double* fields,
                                           const int32_t* offsets, int nfield, const void* X, int64_t N, int64_t ldx,
                                           int d, const void* delta, int64_t ldd, g3_dtype dt, void* K, int64_t ldk,
                                           int64_t kstride, void* invd, void* a, double* out) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!tmpl) return -2;
  if (nfield < 0 || nfield > G3_MAX_FIELDS) return -6;
  if (nfield && (!fields || !offsets)) return -4;
  for (int i = 0; i < nfield; ++i)
    if (!g3h_field_offset_ok(offsets[i])) return -5;
  MemberProgs mp;
  mp.tmpl = tmpl;
  mp.fields = fields;
  mp.offs = offsets;
  mp.nfield = nfield;
  // the shared checks number their arguments as g3_gp_factor_batched does: X and later sit three places further here
  int rc = gp_factor_batched_impl(ctx, mp, batch, X, N, ldx, d, delta, ldd, dt, K, ldk, kstride, invd, a, out);
  return (rc <= -4 && rc >= -16) ? rc - 3 : rc;
}

extern "C" int g3_gp_factor(g3_ctx* ctx, const g3_kernel_prog* prog, const void* X, int64_t N,
                            int64_t ldx, int d, const void* delta, g3_dtype dt, void* K, int64_t ldk,
                            void* invd, void* a, double out[6]) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!prog) return -2;
  if (!X) return -3;
  if (N <= 0) return -4;
  if (!delta) return -7;
  if (!K) return -9;
  const int64_t Np = g3_roundup(N, G3_LB);
  if (ldk < Np || ldk % (16 / (int64_t)g3_esize(dt))) return -10;
  if (!invd) return -11;
  if (!a) return -12;
  if (!out) return -13;
  return gp_factor_impl(ctx, prog, X, N, ldx, d, delta, dt, K, ldk, invd, a, out, nullptr, nullptr, 0, 0, nullptr,
                        nullptr);
}

extern "C" int g3_gp_factor_predict(g3_ctx* ctx, const g3_kernel_prog* prog, const g3_kernel_prog* prog_cross,
                                    const void* X, int64_t N, int64_t ldx, int d, const void* delta,
                                    const void* Xs, int64_t M, int64_t ldxs, g3_dtype dt, void* K, int64_t ldk,
                                    void* invd, void* a, void* mu, void* ss, double out[6]) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!prog) return -2;
  if (!prog_cross) return -3;
  if (!X) return -4;
  if (N <= 0) return -5;
  if (!delta) return -8;
  if (!Xs) return -9;
  if (M <= 0) return -10;
  if (!K) return -13;
  const int64_t Np = g3_roundup(N, G3_LB);
  if (ldk < Np || ldk % (16 / (int64_t)g3_esize(dt))) return -14;
  if (!invd) return -15;
  if (!a) return -16;
  if (!out) return -19;
  return gp_factor_impl(ctx, prog, X, N, ldx, d, delta, dt, K, ldk, invd, a, out, prog_cross, Xs, M, ldxs, mu, ss);
}

extern "C" int g3_gp_cross(g3_ctx* ctx, const g3_kernel_prog* prog, const void* Xs, int64_t M,
                           int64_t ldxs, const void* X, int64_t N, int64_t ldx, int d, const void* L,
                           int64_t ldl, const void* invd, const void* a, g3_dtype dt, void* V, int64_t ldv,
                           void* mu, void* ss) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!prog) return -2;
  if (!Xs) return -3;
  if (M <= 0) return -4;
  if (!X) return -6;
  if (N <= 0) return -7;
  if (!L) return -10;
  const int64_t Np = g3_roundup(N, G3_LB), Mp = g3_roundup(M, 128);
  const int64_t al = 16 / (int64_t)g3_esize(dt);
  if (ldl < Np || ldl % al) return -11;
  if (!V) return -14;
  if (ldv < Np || ldv % al) return -15;
  if (!invd) return -12;
  // V = tt_to_num(cov(Xs, X))  (elliptical.py:78-79), then V <- V L^-T
  const double es_d = (double)g3_esize(dt);
  int pr = g3i_prof_begin(ctx, G3_TAG_CROSS_GRAM, (double)(N + M) * d * es_d + (double)N * M * es_d);
  int rc = g3_gram(ctx, prog, Xs, M, ldxs, X, N, ldx, d, dt, V, ldv, Mp, Np, G3_GRAM_SCRUB);
  g3i_prof_end(ctx, pr);
  if (rc) return rc;
  rc = g3i_reset_info(ctx);
  if (rc) return rc;
  pr = g3i_prof_begin(ctx, G3_TAG_TRSM, (double)N * N * M);
  rc = g3i_trsm_rlt(ctx, L, Np, ldl, V, Mp, ldv, dt, invd);
  g3i_prof_end(ctx, pr);
  if (rc) return rc;
  if (mu || ss) {
    pr = g3i_prof_begin(ctx, G3_TAG_REDUCE, 2.0 * N * M);
    rc = rows_dot_ss_launch(ctx, V, M, N, ldv, a, dt, mu, ss);
    g3i_prof_end(ctx, pr);
  }
  return rc;
}


// draws = loc + L Z  (gaussian.py:92-95).  Z^T and the product live in the context's workspace.
template <typename T>
static int gp_sample_t(g3_ctx* ctx, const T* L, int64_t M, int64_t ldl, const T* loc, const T* Z, int64_t S, T* out) {
  const int64_t Mp = g3_roundup(M, 128), Sp = g3_roundup(S, 64);
  const size_t zt_bytes = (size_t)Sp * Mp * sizeof(T), c_bytes = (size_t)Mp * Sp * sizeof(T);
  int rc = g3i_ensure_work(ctx, zt_bytes + c_bytes);
  if (rc) return rc;
  T* Zt_dev = (T*)ctx->work;
  T* C_dev = (T*)((char*)ctx->work + zt_bytes);
  // Z^T, zero padded, assembled on the host (M x S is small next to the M x M factor)
  T* zt = (T*)calloc((size_t)Sp * Mp, sizeof(T));
  if (!zt) return G3_ERR_NOMEM;
  for (int64_t i = 0; i < M; ++i)
    for (int64_t s = 0; s < S; ++s) zt[s * Mp + i] = Z[i * S + s];
  hipError_t e = hipMemcpyAsync(Zt_dev, zt, zt_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  free(zt);
  if (e != hipSuccess) {
    snprintf(ctx->err, sizeof(ctx->err), "g3_gp_sample: upload of Z failed: %s", hipGetErrorString(e));
    return G3_ERR_HIP;
  }
  rc = g3i_reset_info(ctx);
  if (rc) return rc;
  const g3_dtype dt = sizeof(T) == 8 ? G3_F64 : G3_F32;
  rc = g3i_gemm_nt(ctx, C_dev, Sp, L, ldl, Zt_dev, Mp, Mp, Sp, Mp, 1.0, 0.0, dt, 0);      // C = L (Z^T)^T
  if (rc) return rc;
  T* c = (T*)malloc(c_bytes);
  if (!c) return G3_ERR_NOMEM;
  e = hipMemcpyAsync(c, C_dev, c_bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    free(c);
    snprintf(ctx->err, sizeof(ctx->err), "g3_gp_sample: download failed: %s", hipGetErrorString(e));
    return G3_ERR_HIP;
  }
  for (int64_t i = 0; i < M; ++i)
    for (int64_t s = 0; s < S; ++s) out[i * S + s] = loc[i] + c[i * Sp + s];
  free(c);
  return G3_OK;
}

extern "C" int g3_gp_sample(g3_ctx* ctx, const void* L_dev, int64_t M, int64_t ldl, const void* loc_host,
                            const void* Z_host, int64_t S, g3_dtype dt, void* out_host) {
  if (!ctx) return -1;
  g3_dev_guard _dg(ctx);
  if (!L_dev) return -2;
  if (M <= 0) return -3;
  const int64_t Mp = g3_roundup(M, 128), al = 16 / (int64_t)g3_esize(dt);
  if (ldl < Mp || ldl % al) return -4;
  if (!loc_host) return -5;
  if (!Z_host) return -6;
  if (S <= 0) return -7;
  if (!out_host) return -9;
  if (dt == G3_F64)
    return gp_sample_t<double>(ctx, (const double*)L_dev, M, ldl, (const double*)loc_host, (const double*)Z_host, S,
                               (double*)out_host);
  return gp_sample_t<float>(ctx, (const float*)L_dev, M, ldl, (const float*)loc_host, (const float*)Z_host, S,
                            (float*)out_host);
}
