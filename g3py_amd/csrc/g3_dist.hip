// Multi-GPU GP hot path behind the C ABI: the N x N covariance block-partitioned over the GPUs of one
// node, one process per GPU, RCCL over xGMI owned by the library (dlopen: a single-GPU user never needs
// librccl).  The reference has no distributed code at all (its only parallelism is a process pool,
// g3py/processes/stochastic.py:773-783); the algebra is the Gram + Cholesky + triangular solves of
// g3py/libs/tensors.py:197-222, g3py/processes/gaussian.py:208-224 and elliptical.py:81-97.
//
// Layout: ROW-block-cyclic, blocks dealt from the top in load-balanced rounds (g3h_deal).  A rank stacks its nb-row blocks in
// one full-width local matrix, followed by its share of the right-hand-side rows [delta^T; K(Xs, X)]
// (128-row chunks), which ride through the factorisation as on one GPU.  Only the nb x nb diagonal factor
// travels on the critical path (broadcast, on its OWN communicator and on the look-ahead stream); a panel is
// solved by all ranks at once and exchanged by an all-gather in which every GPU talks to all peers -- the
// pattern a point-to-point xGMI mesh is good at.
//
// Three HIP streams per rank, the schedule of the one-GPU sweep (g3_potrf.hip::potrf_lookahead) with the
// collectives in it.  With G_k the gathered panel k, step k is
//   bulk stream   d1. block column k+2 of everything the rank owns from block k+2 down -= P_k G_k^T   -> event B_k
//                 d2. block columns >= k+3 of its blocks >= k+3 and of the right-hand-side rows (ONE staircase launch)
//   chain stream  a.  (after B_{k-1}) block column k+1 of its blocks >= k+2 -= P_k G_k^T
//   (the ctx's)   b.  (after the broadcast of L_{k+1,k+1}) solve its rows of panel k+1; all-gather them
//   look-ahead    c.  (after B_k) the owner of block k+2 applies panel k+1 to its diagonal block from its own
//   stream            rows, factors it and broadcasts factor + 128 x 128 block inverses
//
// Transports: RCCL (the product) or caller-supplied host callbacks (tests: several ranks sharing ONE GPU over
// gloo -- RCCL refuses two ranks on one device -- so the schedule is checked for P > 1 on a one-GPU box).
#include "g3_internal.h"
#include "g3_host.h"

#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <time.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------- transports
enum { G3_COLL_BCAST = 0, G3_COLL_ALLGATHER = 1, G3_COLL_ALLREDUCE = 2, G3_NCOLL = 3, G3_PH_DIAG = 3, G3_PH_SOLVE = 4, G3_NKIND = 5 };

struct Transport {
  virtual ~Transport() {}
  // stream-ordered after everything queued on s; the result is visible to work queued on s afterwards
  virtual int bcast(void* buf, size_t bytes, int root, hipStream_t s) = 0;
  virtual int allgather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t s) = 0;
  // host values in / out; synchronises s.  op: 0 sum, 1 min, 2 max
  virtual int allreduce(double* host, int n, int op, hipStream_t s) = 0;
  virtual const char* name() const = 0;
  // what the NEXT collective carries (G3_HINT_*, index of the block / panel): only the replay transport needs to know
  virtual void hint(int what, int index) { (void)what; (void)index; }
  char err[256] = {0};
};
enum { G3_HINT_NONE = 0, G3_HINT_DIAG = 1, G3_HINT_PANEL = 2, G3_HINT_AVEC = 3 };

struct RcclApi {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;      // optional
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static RcclApi* rccl_api(char* err, size_t errlen) {
  static RcclApi api;
  static int state = 0;   // 0 untried, 1 ok, -1 failed
  if (state == 1) return &api;
  if (state == -1) return nullptr;
  // a copy already loaded into the process (e.g. the one PyTorch ships) is reused; G3_RCCL_PATH overrides
  const char* forced = getenv("G3_RCCL_PATH");
  const bool only = forced && *forced;          // a path given explicitly is the ONLY candidate: a wrong one fails loudly
  const char* names[] = {forced, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2 && !api.h; ++pass)
    for (const char* n : names) {
      if (!n || !*n) continue;
      if (only && n != forced) break;
      api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (api.h) break;
    }
  if (!api.h) {
    snprintf(err, errlen, "librccl not found (set G3_RCCL_PATH): %s", dlerror());
    state = -1;
    return nullptr;
  }
#define G3_SYM(f)                                                        \
  api.f = (decltype(api.f))dlsym(api.h, "nccl" #f);                      \
  if (!api.f) {                                                          \
    snprintf(err, errlen, "librccl lacks nccl" #f);                      \
    state = -1;                                                          \
    return nullptr;                                                      \
  }
  G3_SYM(GetUniqueId) G3_SYM(CommInitRank) G3_SYM(CommDestroy) G3_SYM(Broadcast) G3_SYM(AllGather) G3_SYM(AllReduce)
  G3_SYM(GetErrorString)
#undef G3_SYM
  api.CommAbort = (decltype(api.CommAbort))dlsym(api.h, "ncclCommAbort");
  state = 1;
  return &api;
}

struct RcclTransport : Transport {
  RcclApi* api = nullptr;
  ncclComm_t comm_gather = nullptr;   // panel all-gathers, scalar all-reduces (chain stream)
  ncclComm_t comm_bcast = nullptr;    // diagonal-factor broadcasts (look-ahead stream): never queues behind a gather
  double* scratch = nullptr;          // device scratch of the scalar all-reduces
  double* hscratch = nullptr;         // pinned
  static const int SCR = 16384;
  bool failed = false;                // an evaluation returned an error on this rank: peers may be inside a collective
  ~RcclTransport() override {
    // after a local error the communicators are ABORTED, not destroyed: ncclCommDestroy waits for outstanding
    // collectives, which the peers of a rank that left the sweep will never complete.  (A failed rank must end the job:
    // the other ranks are by then blocked on the GPU in a collective this rank did not enter.)
    auto drop = [&](ncclComm_t c) { if (failed && api->CommAbort) api->CommAbort(c); else api->CommDestroy(c); };
    if (api && comm_gather) drop(comm_gather);
    if (api && comm_bcast) drop(comm_bcast);
    if (scratch) (void)hipFree(scratch);
    if (hscratch) (void)hipHostFree(hscratch);
  }
  int chk(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return G3_OK;
    snprintf(err, sizeof(err), "%s: %s", what, api->GetErrorString(r));
    return G3_ERR_HIP;
  }
  int bcast(void* buf, size_t bytes, int root, hipStream_t s) override {
    return chk(api->Broadcast(buf, buf, bytes, ncclChar, root, comm_bcast, s), "ncclBroadcast");
  }
  int allgather(const void* send, void* recv, size_t bytes, hipStream_t s) override {
    return chk(api->AllGather(send, recv, bytes, ncclChar, comm_gather, s), "ncclAllGather");
  }
  int allreduce(double* host, int n, int op, hipStream_t s) override {
    for (int off = 0; off < n; off += SCR) {
      const int c = n - off < SCR ? n - off : SCR;
      memcpy(hscratch, host + off, c * sizeof(double));
      if (hipMemcpyAsync(scratch, hscratch, c * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) return G3_ERR_HIP;
      const ncclRedOp_t o = op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax);
      int rc = chk(api->AllReduce(scratch, scratch, c, ncclDouble, o, comm_gather, s), "ncclAllReduce");
      if (rc) return rc;
      if (hipMemcpyAsync(hscratch, scratch, c * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return G3_ERR_HIP;
      if (hipStreamSynchronize(s) != hipSuccess) return G3_ERR_HIP;
      memcpy(host + off, hscratch, c * sizeof(double));
    }
    return G3_OK;
  }
  const char* name() const override { return "rccl"; }
};

struct CallbackTransport : Transport {
  g3_dist_callbacks cb;
  int bcast(void* buf, size_t bytes, int root, hipStream_t s) override {
    if (hipStreamSynchronize(s) != hipSuccess) return G3_ERR_HIP;
    return cb.bcast(cb.user, buf, bytes, root) ? G3_ERR_HIP : G3_OK;
  }
  int allgather(const void* send, void* recv, size_t bytes, hipStream_t s) override {
    if (hipStreamSynchronize(s) != hipSuccess) return G3_ERR_HIP;
    return cb.allgather(cb.user, send, recv, bytes) ? G3_ERR_HIP : G3_OK;
  }
  int allreduce(double* host, int n, int op, hipStream_t s) override {
    if (hipStreamSynchronize(s) != hipSuccess) return G3_ERR_HIP;
    return cb.allreduce(cb.user, host, n, op) ? G3_ERR_HIP : G3_OK;
  }
  const char* name() const override { return "callbacks"; }
};

// ---- asynchronous host-callback transport: the collectives as jobs of two worker threads (gather / all-reduce, and
// broadcast: the product's two communicators), stream-ordered like RCCL calls.
//   issue (host thread):  record E on the stream -> queue the job -> launch a one-wave kernel on the stream that waits for
//                         the worker's ticket (a word in pinned host memory) -> return: the host runs ahead
//   worker:               wait for E -> device -> pinned staging (its own copy stream) -> callback -> staging -> device
//                         -> publish the ticket: the stream goes on
// The wait kernel has a wall-clock limit (G3_DIST_ASYNC_TIMEOUT_S, default 120 s): a stream never hangs on a dead peer, the
// limit sets a device error word that every later transport call reports.  The workers' copy streams have NORMAL priority:
// HIP keeps separate hardware-queue pools per priority, so they never sit behind a waiting high- or low-priority stream of
// the driver.
__global__ void host_ticket_wait_kernel(const unsigned* ticket, unsigned want, unsigned long long limit, unsigned* err) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();
  unsigned spins = 0;
  while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    __builtin_amdgcn_s_sleep(8);
    if ((++spins & 63u) == 0 && wall_clock64() - t0 > limit) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

struct AsyncCallbackTransport : Transport {
  g3_dist_host_callbacks cb;
  int device = 0, rank = 0, world = 1;
  struct Job {
    int kind = -1;                 // G3_COLL_*; -1: quit
    hipEvent_t ev = nullptr;
    void* buf = nullptr;           // bcast buffer / all-gather receive buffer (device)
    const void* send = nullptr;    // all-gather: this rank's part (device)
    size_t bytes = 0;
    int root = 0;
    double* vals = nullptr;        // all-reduce (host, the caller waits)
    int n = 0, op = 0;
    unsigned seq = 0;
  };
  struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::deque<Job> q;
    unsigned issued = 0, done = 0;       // tickets (done is mirrored in *ticket for the device)
    unsigned* ticket = nullptr;          // pinned, mapped
    unsigned* ticket_dev = nullptr;
    hipStream_t cs = nullptr;            // copy stream
    char* stage = nullptr;               // pinned staging
    size_t stage_bytes = 0;
    std::vector<char*> retired;          // outgrown staging buffers: released at teardown (hipHostFree synchronises the
                                         // DEVICE -- called while a stream waits for this worker it would never return)
    int failed = 0;
    char msg[200] = {0};
  } w[2];                                // 0: all-gather + all-reduce, 1: broadcast
  unsigned* err_host = nullptr;          // pinned: a wait kernel ran into its limit
  unsigned* err_dev = nullptr;
  unsigned long long limit = 12000000000ull;
  std::vector<hipEvent_t> evpool;
  std::mutex evmu;

  int start() {
    if (hipHostMalloc((void**)&err_host, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return G3_ERR_HIP;
    *err_host = 0;
    if (hipHostGetDevicePointer((void**)&err_dev, err_host, 0) != hipSuccess) return G3_ERR_HIP;
    const int lim = g3h_env_int("G3_DIST_ASYNC_TIMEOUT_S", 120);
    limit = (unsigned long long)(lim > 0 ? lim : 120) * 100000000ull;      // wall_clock64 ticks at 100 MHz
    for (int i = 0; i < 2; ++i) {
      if (hipHostMalloc((void**)&w[i].ticket, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return G3_ERR_HIP;
      *w[i].ticket = 0;
      if (hipHostGetDevicePointer((void**)&w[i].ticket_dev, w[i].ticket, 0) != hipSuccess) return G3_ERR_HIP;
      if (hipStreamCreateWithFlags(&w[i].cs, hipStreamNonBlocking) != hipSuccess) return G3_ERR_HIP;
      w[i].th = std::thread([this, i]() { run(i); });
    }
    return G3_OK;
  }
  ~AsyncCallbackTransport() override {
    for (int i = 0; i < 2; ++i) {
      if (w[i].th.joinable()) {
        { std::lock_guard<std::mutex> lk(w[i].mu); w[i].q.push_back(Job()); }
        w[i].cv.notify_all();
        w[i].th.join();
      }
      if (w[i].cs) { (void)hipStreamSynchronize(w[i].cs); (void)hipStreamDestroy(w[i].cs); }
      if (w[i].stage) (void)hipHostFree(w[i].stage);
      for (char* p : w[i].retired) (void)hipHostFree(p);
      if (w[i].ticket) (void)hipHostFree(w[i].ticket);
    }
    for (hipEvent_t e : evpool) (void)hipEventDestroy(e);
    if (err_host) (void)hipHostFree(err_host);
  }
  hipEvent_t take_event() {
    std::lock_guard<std::mutex> lk(evmu);
    if (!evpool.empty()) { hipEvent_t e = evpool.back(); evpool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
  }
  void give_event(hipEvent_t e) { std::lock_guard<std::mutex> lk(evmu); evpool.push_back(e); }
  bool ensure_stage(Worker& W, size_t bytes) {
    if (W.stage_bytes >= bytes) return true;
    if (W.stage) W.retired.push_back(W.stage);       // (not freed here: see `retired`)
    W.stage = nullptr; W.stage_bytes = 0;
    bytes += bytes / 4;                              // head room: plans of one driver differ by a few blocks
    if (hipHostMalloc((void**)&W.stage, bytes, hipHostMallocDefault) != hipSuccess) return false;
    W.stage_bytes = bytes;
    return true;
  }
  void fail(Worker& W, const char* what) {
    W.failed = 1;
    snprintf(W.msg, sizeof(W.msg), "%s", what);
  }
  void run(int i) {
    Worker& W = w[i];
    (void)hipSetDevice(device);
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(W.mu);
        W.cv.wait(lk, [&]() { return !W.q.empty(); });
        j = W.q.front();
        W.q.pop_front();
      }
      if (j.kind < 0) return;
      bool ok = !W.failed;
      if (j.ev) {
        if (hipEventSynchronize(j.ev) != hipSuccess) { ok = false; fail(W, "waiting for the stream failed"); }
        give_event(j.ev);
      }
      if (ok && j.kind == G3_COLL_BCAST) {
        ok = ensure_stage(W, j.bytes);
        if (ok && rank == j.root) ok = hipMemcpyAsync(W.stage, j.buf, j.bytes, hipMemcpyDeviceToHost, W.cs) == hipSuccess && hipStreamSynchronize(W.cs) == hipSuccess;
        if (ok) ok = cb.bcast(cb.user, W.stage, j.bytes, j.root) == 0;
        if (ok && rank != j.root) ok = hipMemcpyAsync(j.buf, W.stage, j.bytes, hipMemcpyHostToDevice, W.cs) == hipSuccess && hipStreamSynchronize(W.cs) == hipSuccess;
        if (!ok && !W.failed) fail(W, "broadcast failed (staging copy or callback)");
      } else if (ok && j.kind == G3_COLL_ALLGATHER) {
        ok = ensure_stage(W, j.bytes * (size_t)world);
        char* mine = W.stage + (size_t)rank * j.bytes;
        if (ok) ok = hipMemcpyAsync(mine, j.send, j.bytes, hipMemcpyDeviceToHost, W.cs) == hipSuccess && hipStreamSynchronize(W.cs) == hipSuccess;
        if (ok) ok = cb.allgather(cb.user, mine, W.stage, j.bytes) == 0;
        // everything but this rank's own part goes back (the in-place gather leaves that where it is; an out-of-place
        // one -- send outside the receive buffer -- gets the local copy as well)
        const bool inplace = (const char*)j.send == (const char*)j.buf + (size_t)rank * j.bytes;
        for (int q = 0; q < world && ok; ++q)
          if (q != rank || !inplace)
            ok = hipMemcpyAsync((char*)j.buf + (size_t)q * j.bytes, W.stage + (size_t)q * j.bytes, j.bytes, hipMemcpyHostToDevice, W.cs) == hipSuccess;
        if (ok) ok = hipStreamSynchronize(W.cs) == hipSuccess;
        if (!ok && !W.failed) fail(W, "all-gather failed (staging copy or callback)");
      } else if (ok && j.kind == G3_COLL_ALLREDUCE) {
        ok = cb.allreduce(cb.user, j.vals, j.n, j.op) == 0;
        if (!ok && !W.failed) fail(W, "all-reduce callback failed");
      }
      // the ticket is published even after a failure: a stream must never be left waiting (the failure is reported by
      // the next transport call and ends the evaluation with an error)
      {
        std::lock_guard<std::mutex> lk(W.mu);
        W.done = j.seq;
        __atomic_store_n(W.ticket, j.seq, __ATOMIC_RELEASE);
      }
      W.cv_done.notify_all();
    }
  }
  int check() {
    for (int i = 0; i < 2; ++i)
      if (w[i].failed) { snprintf(err, sizeof(err), "%s", w[i].msg); return G3_ERR_HIP; }
    if (*err_host) { snprintf(err, sizeof(err), "a stream waited longer than the limit for a collective (peer gone?)"); return G3_ERR_HIP; }
    return G3_OK;
  }
  int issue(int wi, Job j, hipStream_t s) {
    int rc = check();
    if (rc) return rc;
    Worker& W = w[wi];
    j.ev = take_event();
    if (!j.ev || hipEventRecord(j.ev, s) != hipSuccess) { snprintf(err, sizeof(err), "event record failed"); return G3_ERR_HIP; }
    {
      std::lock_guard<std::mutex> lk(W.mu);
      j.seq = ++W.issued;
      W.q.push_back(j);
    }
    W.cv.notify_all();
    hipLaunchKernelGGL(host_ticket_wait_kernel, dim3(1), dim3(64), 0, s, (const unsigned*)W.ticket_dev, j.seq, limit, err_dev);
    if (hipGetLastError() != hipSuccess) { snprintf(err, sizeof(err), "wait-kernel launch failed"); return G3_ERR_HIP; }
    return G3_OK;
  }
  int bcast(void* buf, size_t bytes, int root, hipStream_t s) override {
    Job j; j.kind = G3_COLL_BCAST; j.buf = buf; j.bytes = bytes; j.root = root;
    return issue(1, j, s);
  }
  int allgather(const void* send, void* recv, size_t bytes, hipStream_t s) override {
    Job j; j.kind = G3_COLL_ALLGATHER; j.send = send; j.buf = recv; j.bytes = bytes;
    return issue(0, j, s);
  }
  int allreduce(double* host, int n, int op, hipStream_t s) override {
    // behind everything queued on s and behind every gather issued so far; the caller needs the values: wait here
    Job j; j.kind = G3_COLL_ALLREDUCE; j.vals = host; j.n = n; j.op = op;
    int rc = issue(0, j, s);
    if (rc) return rc;
    Worker& W = w[0];
    unsigned want;
    { std::lock_guard<std::mutex> lk(W.mu); want = W.issued; }
    { std::unique_lock<std::mutex> lk(W.mu); W.cv_done.wait(lk, [&]() { return W.done >= want; }); }
    if (hipStreamSynchronize(s) != hipSuccess) return G3_ERR_HIP;
    return check();
  }
  const char* name() const override { return "callbacks-async"; }
};

}  // namespace

// ---------------------------------------------------------------------------------------- the driver object
struct g3_dist {
  g3_ctx* ctx = nullptr;        // chain stream (the caller's context)
  g3_ctx* ctx_look = nullptr;   // look-ahead stream: its own context, so no scratch / info flag is shared
  g3_ctx* ctx_bulk = nullptr;   // bulk stream (the one the current plan uses: plain or CU-masked, see g3_dist_plan)
  hipStream_t s_look = nullptr, s_bulk = nullptr;
  // From six ranks on the driver holds TWO bulk streams, one barred from 32 CUs (pick_streams) and one plain, each with its
  // context; a plan takes the masked one only where the serial diagonal chain would otherwise outlast the rank's own step
  hipStream_t s_bulk_alt[2] = {nullptr, nullptr};      // [0] plain low-priority, [1] CU-masked (nullptr: not created)
  g3_ctx* ctx_bulk_alt[2] = {nullptr, nullptr};
  int bulk_mask_mode = 0;                              // G3_DIST_BULK_MASK_MODE: 0 by plan (default), 1 always, -1 never
  int bulk_masked = 0;                                 // what the current plan took
  // The chain runs on a stream of the DRIVER (not the caller's), created back to back with the other two: which hardware
  // queue -- and which of the chip's compute pipes -- a stream lands on is decided by creation order, and the same replayed
  // rank measured 36.9 or 42.1 ms (config 4, P = 8, one box) depending on how many streams the process had created before
  // the driver's.  Three consecutive creations keep the three streams of the sweep on distinct queues whatever the caller
  // did; its stream only brackets an entry point (ChainScope).  G3_DIST_OWN_CHAIN=0: the chain on the caller's stream.
  hipStream_t s_chain = nullptr;
  hipEvent_t ev_bracket = nullptr;
  std::vector<hipStream_t> pad_streams;   // G3_DIST_PAD_STREAMS=n (measurements): n idle streams created in front of the three
  int rank = 0, world = 1;
  Transport* tr = nullptr;
  // plan
  bool planned = false;
  int64_t N = 0, M = 0, nb = 0, Np = 0, Mp = 0;
  int d = 0, nblk = 0, nchunk = 0;
  g3_dtype dt = G3_F64;
  size_t es = 8;
  std::vector<int> owner;       // owner[I]: the rank that holds row block I (g3h_deal, the same table on every rank)
  std::vector<int> my_blocks, my_chunks;
  std::vector<int64_t> loff;    // local row offset of an owned block, -1 otherwise
  int64_t rows_mat = 0, rows_rhs = 0, cmax = 1;
  char* A = nullptr;            // (rows_mat + rows_rhs) x Np local rows, full width
  char* dbuf[2] = {nullptr, nullptr};   // nb x nb diagonal factor + nb x 128 block inverses
  char* send[2] = {nullptr, nullptr};
  char* gath[3] = {nullptr, nullptr, nullptr};
  char* avec = nullptr;         // 1 x Np: a = L^-1 delta, broadcast for the mean
  char* dots = nullptr;         // 2 x 128 scratch of rows_dot_ss, one pair per right-hand-side chunk of this rank
  double* blkstats = nullptr;   // 4 doubles per owned block: per-block reductions stay on the device until ONE copy fetches them
  std::vector<char> xs_seen;    // the prediction points of the last evaluation (M x d, packed): the cross-solve rows V in the
                                // driver belong to THEM, and posterior_cov / draws refuse any other Xs
  int* info_dev = nullptr;
  double phase_calls[2] = {0, 0}, phase_ms[2] = {0, 0};   // as of the last g3_dist_comm_stats
  bool serial_coll = false;     // G3_DIST_SERIAL_COLL=1 (read at creation): the two communicators are never in flight together
  bool keep = false;            // g3_dist_set_keep: every block's inverses are kept (the reference of a replay)
  char* wstore = nullptr;       // Np x 128: block inverses of all diagonal blocks, in block order
  char* vstore = nullptr;       // Np x nb: the full inverses V_k = L_kk^-1 of all diagonal blocks (keep && fullinv)
  // Panel solve as ONE product (round 5): the owner of diagonal block k inverts the whole nb x nb factor on its look-ahead
  // stream (g3i_trtri_full: recursive doubling from the 128-block inverses, 7 small launches at nb = 1024) and broadcasts
  // V_k = L_kk^-1 -- nb x nb, less than the (L, W) pair it replaces -- and every rank's rows of panel k become X V_k^T: one
  // K-triangular MFMA GEMM instead of a 20-step stripe recursion that ran at ~4 TFLOP/s on a rank's few stripes.
  // G3_DIST_FULLINV=0 (read at creation) or a block height that is not 128 * 2^q <= 2048 keeps the stripe solve.
  bool want_fullinv = true, fullinv = false;
  char* vt = nullptr;           // nb x nb scratch of the inversion: V^T
  char* ubuf = nullptr;         // nb x nb scratch of the inversion: (L21 V11)^T per pair
  char* rbuf = nullptr;         // (rows_rhs + rows_inv) x nb: the right-hand-side rows of a panel, solved out of place
  int deal_snake = 0;           // G3_DIST_DEAL=snake (read at creation): the round-4 dealing, for A/B measurements
  int plan_gen = 0;             // bumped by every g3_dist_plan: a replay notices a re-planned reference
  int replay_refs = 0;          // replay drivers that read this driver's factor: it refuses re-plan / destroy meanwhile
  // phases timed with the collectives' event machinery: 3 = a diagonal block's update + factorisation, 4 = a panel solve
  // gradient mode (g3_dist_set_grad): N / P more right-hand-side rows -- the identity, which the sweep turns into the
  // rank's rows of L^-T -- and the rank's rows of K^-1 = L^-T L^-1
  bool grad = false, have_inv = false;
  int64_t rows_inv = 0, cmax_all = 1;
  char* Kinv = nullptr;         // rows_mat x Np
  char* alpha_dev = nullptr;    // 1 x Np: alpha = K^-1 delta (scaled), global order
  char* agath = nullptr;        // world x cmax_all x nb: gathered alpha pieces
  std::vector<hipEvent_t> ev;   // B_k events + stream joins
  // accounting: per collective kind calls, bytes (sent + received by this rank), device milliseconds
  double n_calls[G3_NKIND] = {0, 0, 0, 0, 0}, n_bytes[G3_NKIND] = {0, 0, 0, 0, 0};
  std::vector<hipEvent_t> tev;  // timing event pairs
  std::vector<int> tkind;
  size_t tused = 0;
  bool timing_now = false;
  // outcome of the last evaluation
  int last_info = 0, last_tries = 0, last_fallback = 0;
  char err[512] = {0};
};

namespace {
// ---- replay transport: ONE rank of a P-rank evaluation, alone on the GPU.  The reference is a world-1 driver that has
// just evaluated the same problem with the same block height and kept its block inverses (g3_dist_set_keep): its local
// matrix is the whole factor.  Every collective becomes a device-to-device copy of exactly the bytes this rank would
// receive -- the diagonal factor it does not own, the other ranks' blocks of a panel, a = L^-1 delta -- and the scalar
// all-reduces return the rank's own contribution.  What this times is everything except the fabric.
struct ReplayTransport : Transport {
  g3_dist* self = nullptr;
  g3_dist* ref = nullptr;
  int ref_gen = -1;               // plan generation of the reference this replay was planned against
  int what = G3_HINT_NONE, index = -1;
  bool ref_ok() const;            // the reference still holds the factor this replay was planned against
  void hint(int w, int i) override { what = w; index = i; }
  int fail(const char* m) { snprintf(err, sizeof(err), "%s", m); return G3_ERR_HIP; }
  int bcast(void* buf, size_t bytes, int root, hipStream_t s) override;
  int allgather(const void* send, void* recv, size_t bytes, hipStream_t s) override;
  int allreduce(double* host, int n, int op, hipStream_t s) override {
    (void)host; (void)n; (void)op;                   // the rank's own contribution stays as it is
    return hipStreamSynchronize(s) == hipSuccess ? G3_OK : G3_ERR_HIP;
  }
  const char* name() const override { return "replay"; }
};
}  // namespace

#define G3D_HIP(call)                                                                              \
  do {                                                                                             \
    hipError_t _e = (call);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      snprintf(D->err, sizeof(D->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      return G3_ERR_HIP;                                                                           \
    }                                                                                              \
  } while (0)
// a failing library call: keep its context's message
#define G3D_RC(cx, call)                                                                           \
  do {                                                                                             \
    int _rc = (call);                                                                              \
    if (_rc) {                                                                                     \
      snprintf(D->err, sizeof(D->err), "%s:%d %s -> %d %s", __FILE__, __LINE__, #call, _rc, (cx) ? (cx)->err : ""); \
      return _rc < 0 ? _rc : G3_ERR_HIP;                                                           \
    }                                                                                              \
  } while (0)

// An entry point's work runs on the driver's own chain stream: it starts behind everything the caller has queued on its
// stream, and the caller's stream continues behind it.  (The context's program ring records its events on the stream a slot
// was consumed on, so switching the stream of a context between calls is safe.)
struct ChainScope {
  g3_dist* D;
  hipStream_t user;
  explicit ChainScope(g3_dist* D_) : D(D_), user(D_->ctx->stream) {
    if (!D->s_chain) return;
    (void)hipEventRecord(D->ev_bracket, user);
    (void)hipStreamWaitEvent(D->s_chain, D->ev_bracket, 0);
    D->ctx->stream = D->s_chain;
  }
  ~ChainScope() {
    if (!D->s_chain) return;
    (void)hipEventRecord(D->ev_bracket, D->s_chain);
    (void)hipStreamWaitEvent(user, D->ev_bracket, 0);
    D->ctx->stream = user;
  }
  ChainScope(const ChainScope&) = delete;
  ChainScope& operator=(const ChainScope&) = delete;
};

// a sweep that ended in an error on this rank: its RCCL communicators are aborted (not destroyed) at teardown
static int mark_failed(g3_dist* D, int rc) {
  if (RcclTransport* t = dynamic_cast<RcclTransport*>(D->tr)) t->failed = true;
  return rc;
}

static int owner_of(const g3_dist* D, int I) { return D->owner[I]; }   // g3h_deal (g3_host.h)

// blocks per rank in the padded gather of step k, and the position of global block k+1+s in the rank-major
// gather buffer
static int perm_of(const g3_dist* D, int k, std::vector<int32_t>* idx) { return g3h_gather_table(D->owner, D->world, k + 1, D->nblk - 1, idx); }

static inline char* Aat(const g3_dist* D, int64_t row, int64_t col) { return D->A + ((size_t)row * D->Np + col) * D->es; }
static inline char* Lof(const g3_dist* D, int k) { return D->dbuf[k % 2]; }
static inline char* Wof(const g3_dist* D, int k) { return D->dbuf[k % 2] + (size_t)D->nb * D->nb * D->es; }
static inline char* Vof(const g3_dist* D, int k) { return D->dbuf[k % 2] + ((size_t)D->nb * D->nb + (size_t)D->nb * 128) * D->es; }
// one broadcast buffer: L (nb x nb), its 128-block inverses (nb x 128) and, with the full inverse, V (nb x nb)
static inline size_t dbuf_bytes(const g3_dist* D) { return ((size_t)D->nb * D->nb * (D->fullinv ? 2 : 1) + (size_t)D->nb * 128) * D->es; }
// what travels: V alone with the full inverse (the owner keeps L in its matrix), else L and the block inverses
static inline char* bc_ptr(const g3_dist* D, int k) { return D->fullinv ? Vof(D, k) : D->dbuf[k % 2]; }
static inline size_t bc_bytes(const g3_dist* D) { return (D->fullinv ? (size_t)D->nb * D->nb : (size_t)D->nb * D->nb + (size_t)D->nb * 128) * D->es; }

// all blocks of a panel that other ranks would send, in ONE launch (a hipMemcpy2DAsync per 2 MB block costs ~20 us each:
// 70 ms per evaluation at nb = 512, which would be the replay's own artefact, not the schedule's)
struct ReplayPanelTab {
  int nblk;                       // blocks k+1 .. of the panel
  int slot[G3_RASTER_MAX + 96];   // destination slot in the rank-major gather buffer, -1: this rank's own (not copied)
};
__global__ void replay_panel_kernel(char* __restrict__ recv, const char* __restrict__ src0, size_t src_ld_bytes, size_t row_bytes,
                                    int nb, const ReplayPanelTab tab) {
  const int b = blockIdx.x;
  const int slot = tab.slot[b];
  if (slot < 0) return;
  const char* src = src0 + (size_t)b * nb * src_ld_bytes;
  char* dst = recv + (size_t)slot * nb * row_bytes;
  for (int r = blockIdx.y; r < nb; r += gridDim.y) {
    const uint4* sp = reinterpret_cast<const uint4*>(src + (size_t)r * src_ld_bytes);
    uint4* dp = reinterpret_cast<uint4*>(dst + (size_t)r * row_bytes);
    for (size_t c = threadIdx.x; c < row_bytes / 16; c += blockDim.x) dp[c] = sp[c];
  }
}

bool ReplayTransport::ref_ok() const {
  return ref && ref->planned && ref->keep && ref->A && ref->wstore && (!self->fullinv || ref->vstore) && ref->plan_gen == ref_gen;
}

int ReplayTransport::bcast(void* buf, size_t bytes, int root, hipStream_t s) {
  const g3_dist* D = self;
  if (root == D->rank) return G3_OK;
  if (!ref_ok()) return fail("replay: the reference driver was re-planned or released after this replay was planned");
  const size_t es = D->es;
  if (what == G3_HINT_DIAG) {
    if (index < 0 || index >= D->nblk || bytes != bc_bytes(D)) return fail("replay: bad diagonal-factor broadcast");
    if (D->fullinv) {              // V_jj = L_jj^-1 as the reference's owner computed it
      if (hipMemcpyAsync(buf, ref->vstore + (size_t)index * D->nb * D->nb * es, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail("replay: copy of an inverted diagonal factor failed");
      return G3_OK;
    }
    // L_jj (nb x nb, leading dimension nb) out of the reference's full factor, then its nb x 128 block inverses
    const char* src = ref->A + ((size_t)index * D->nb * ref->Np + (size_t)index * D->nb) * es;
    if (hipMemcpy2DAsync(buf, (size_t)D->nb * es, src, (size_t)ref->Np * es, (size_t)D->nb * es, (size_t)D->nb, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return fail("replay: copy of a diagonal factor failed");
    if (hipMemcpyAsync((char*)buf + (size_t)D->nb * D->nb * es, ref->wstore + (size_t)index * D->nb * 128 * es, (size_t)D->nb * 128 * es,
                       hipMemcpyDeviceToDevice, s) != hipSuccess)
      return fail("replay: copy of the block inverses failed");
    return G3_OK;
  }
  if (what == G3_HINT_AVEC) {
    // a = L^-1 delta: row 0 of the reference's right-hand-side chunk 0
    if (hipMemcpyAsync(buf, ref->A + (size_t)ref->rows_mat * ref->Np * es, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return fail("replay: copy of a failed");
    return G3_OK;
  }
  return fail("replay transport: this broadcast is not part of g3_dist_gp_factor_predict");
}

int ReplayTransport::allgather(const void* send, void* recv, size_t bytes, hipStream_t s) {
  const g3_dist* D = self;
  if (what != G3_HINT_PANEL || index < 0 || index >= D->nblk - 1)
    return fail("replay transport: this all-gather is not part of g3_dist_gp_factor_predict");
  if (!ref_ok()) return fail("replay: the reference driver was re-planned or released after this replay was planned");
  const int k = index;
  const size_t es = D->es, blk = (size_t)D->nb * D->nb * es;
  std::vector<int32_t> idx;
  const int cnt = g3h_gather_table(D->owner, D->world, k + 1, D->nblk - 1, &idx);
  if ((size_t)cnt * blk != bytes) return fail("replay: panel all-gather of an unexpected size");
  // this rank's own slots, as the collective would place them
  if ((const char*)send != (char*)recv + (size_t)D->rank * bytes &&
      hipMemcpyAsync((char*)recv + (size_t)D->rank * bytes, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return fail("replay: local copy failed");
  // the other ranks' blocks of panel k: block (I, k) of the reference factor, all in one launch
  const int nbk = D->nblk - 1 - k;
  for (int b0 = 0; b0 < nbk; b0 += G3_RASTER_MAX + 96) {
    ReplayPanelTab tab;
    tab.nblk = nbk - b0 < G3_RASTER_MAX + 96 ? nbk - b0 : G3_RASTER_MAX + 96;
    for (int b = 0; b < tab.nblk; ++b) {
      const int I = k + 1 + b0 + b;
      tab.slot[b] = D->owner[I] == D->rank ? -1 : idx[I - k - 1];
    }
    const char* src0 = ref->A + ((size_t)(k + 1 + b0) * D->nb * ref->Np + (size_t)k * D->nb) * es;
    hipLaunchKernelGGL(replay_panel_kernel, dim3((unsigned)tab.nblk, 32), dim3(256), 0, s, (char*)recv, src0, (size_t)ref->Np * es,
                       (size_t)D->nb * es, (int)D->nb, tab);
    if (hipGetLastError() != hipSuccess) return fail("replay: panel copy launch failed");
  }
  return G3_OK;
}

static const size_t G3D_MAX_TIMED = 16384;   // event pairs kept between two g3_dist_comm_stats calls; beyond that only counts
static int coll_begin(g3_dist* D, int kind, hipStream_t s, double bytes) {
  D->n_calls[kind] += 1;
  D->n_bytes[kind] += bytes;
  D->timing_now = D->tused + 2 <= 2 * G3D_MAX_TIMED;
  if (!D->timing_now) return G3_OK;        // a caller that never collects the statistics must not grow the event pool
  if (D->tused + 2 > D->tev.size()) {
    for (int i = 0; i < 64; ++i) {
      hipEvent_t e;
      G3D_HIP(hipEventCreate(&e));
      D->tev.push_back(e);
    }
  }
  if (D->tused / 2 >= D->tkind.size()) D->tkind.resize(D->tused / 2 + 64);
  D->tkind[D->tused / 2] = kind;
  G3D_HIP(hipEventRecord(D->tev[D->tused], s));
  return G3_OK;
}
static int coll_end(g3_dist* D, hipStream_t s) {
  if (!D->timing_now) return G3_OK;
  G3D_HIP(hipEventRecord(D->tev[D->tused + 1], s));
  D->tused += 2;
  return G3_OK;
}
#define G3D_TR(call)                                                                        \
  do {                                                                                      \
    int _rc = (call);                                                                       \
    if (_rc) {                                                                              \
      snprintf(D->err, sizeof(D->err), "%s transport: %s", D->tr->name(), D->tr->err);      \
      return _rc;                                                                           \
    }                                                                                       \
  } while (0)

static int do_bcast(g3_dist* D, void* buf, size_t bytes, int root, hipStream_t s, int what = G3_HINT_NONE, int index = -1) {
  int rc = coll_begin(D, G3_COLL_BCAST, s, D->world > 1 ? (double)bytes : 0.0);
  if (rc) return rc;
  D->tr->hint(what, index);
  G3D_TR(D->tr->bcast(buf, bytes, root, s));
  return coll_end(D, s);
}
static int do_allgather(g3_dist* D, const void* sendb, void* recvb, size_t bytes, hipStream_t s, int what = G3_HINT_NONE, int index = -1) {
  int rc = coll_begin(D, G3_COLL_ALLGATHER, s, 2.0 * (D->world - 1) * (double)bytes);
  if (rc) return rc;
  D->tr->hint(what, index);
  G3D_TR(D->tr->allgather(sendb, recvb, bytes, s));
  return coll_end(D, s);
}
static int do_allreduce(g3_dist* D, double* host, int n, int op) {
  int rc = coll_begin(D, G3_COLL_ALLREDUCE, D->ctx->stream, 2.0 * (D->world - 1) / D->world * n * 8.0);
  if (rc) return rc;
  G3D_TR(D->tr->allreduce(host, n, op, D->ctx->stream));
  return coll_end(D, D->ctx->stream);
}

// ---------------------------------------------------------------------------------------- stream placement probe
// Where a HIP stream's hardware queue lands is decided by the runtime from the process's stream history, and it matters:
// when the bulk stream's queue shares a compute pipe with the chain's or the look-ahead's, their small kernels wait for the
// pipe to finish DISPATCHING a CU-filling bulk launch -- the same replayed rank (config 4, P = 8) took 33.5 or 42 ms with
// nothing changed but the number of idle streams created before the driver's (profiles/r05_placement.txt).  HIP offers no
// way to ask; so the driver measures: it creates four high- and four low-priority candidates, and for every pair runs a
// dispatch-bound kernel (32768 short workgroups, ~0.7 ms) on one while a one-wave kernel is submitted on the other.  Sharing
// shows as the small kernel's latency jumping from ~20 us to a large part of the long kernel's duration.  The three streams
// of the sweep are a triple with no sharing; the rest are destroyed.  ~60 ms once per driver; G3_DIST_PROBE=0 takes the
// first of each (G3_DIST_PROBE_LOG=1 prints the matrix).
// picks (chain, look, bulk) out of freshly created candidates; on any failure falls back to the first of each
static int pick_streams(g3_dist* D, int lo, int hi, bool own_chain) {
  const int NH = 4, NL = 8, NLP = 4;      // low candidates 0 .. NLP-1: plain low priority; NLP .. NL-1: CU-masked (if any)
  hipStream_t H[NH] = {nullptr, nullptr, nullptr, nullptr}, L[NL] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int i = 0; i < NH && e == hipSuccess; ++i) e = hipStreamCreateWithPriority(&H[i], hipStreamNonBlocking, hi);
  // From six ranks on a bulk stream exists that may not use 32 of the 256 CUs (mask bit i = CU i div 8 of XCC i mod 8: the
  // first four CUs of every XCC; G3_DIST_BULK_MASK=n overrides, 0 = none): the chain's and the look-ahead's latency-bound
  // kernels then find CUs without bulk workgroups.  At P = 8 a rank's own step is ~34 ms (config 4) but the SERIAL chain of
  // diagonal blocks -- every block's update + factorisation + inversion on its owner, which no rank can overlap with the next
  // block's -- summed to 48 ms: that, not the rank, bounds the P-rank step.  With the reservation: rank 35.5 ms, chain 36.3 ms
  // (replay, profiles/r05_replay_mask.txt; 48 or 64 CUs cost the bulk more than the chain gains; at P = 4 the rank is the
  // bound and the reservation costs 7 %).  Where the rank's step is far longer than the chain (config 5's shape: 137 against
  // 59 ms) the reservation only costs, so a PLAN chooses (g3_dist_plan) between this stream and a plain one.  A CU-masked
  // stream cannot carry the low priority; it is still below the two high ones.
  const int reserve = g3h_env_int("G3_DIST_BULK_MASK", D->world >= 6 ? 32 : 0);
  for (int i = 0; i < NLP && e == hipSuccess; ++i) e = hipStreamCreateWithPriority(&L[i], hipStreamNonBlocking, lo);
  if (reserve > 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, D->ctx->device) != hipSuccess) e = hipErrorUnknown;
    uint32_t mask[16];
    for (int i = 0; i < 16; ++i) mask[i] = 0xffffffffu;
    const uint32_t words = (uint32_t)((prop.multiProcessorCount + 31) / 32);
    if (prop.multiProcessorCount % 32) mask[words - 1] = (1u << (prop.multiProcessorCount % 32)) - 1u;
    for (int b = 0; b < reserve && b < prop.multiProcessorCount; ++b) mask[b / 32] &= ~(1u << (b % 32));
    for (int i = NLP; i < NL && e == hipSuccess; ++i) e = hipExtStreamCreateWithCUMask(&L[i], words, mask);
  }
  unsigned* scratch = nullptr;
  if (e == hipSuccess) e = hipMalloc((void**)&scratch, sizeof(unsigned));
  auto cleanup = [&](int kc, int kl, int kb, int kb2) {
    for (int i = 0; i < NH; ++i) if (H[i] && i != kc && i != kl) (void)hipStreamDestroy(H[i]);
    for (int i = 0; i < NL; ++i) if (L[i] && i != kb && i != kb2) (void)hipStreamDestroy(L[i]);
    if (scratch) (void)hipFree(scratch);
  };
  if (e != hipSuccess) { cleanup(-1, -1, -1, -1); return G3_ERR_HIP; }
  int kc = 0, kl = 1, kb = 0, kbm = reserve > 0 ? NLP : -1;     // kb: the plain bulk stream; kbm: the masked one
  const bool probe = g3h_env_int("G3_DIST_PROBE", 1) != 0, log = g3h_env_int("G3_DIST_PROBE_LOG", 0) != 0;
  hipStream_t user = D->ctx->stream;
  if (probe) {
    // th[i][j]: latency of a one-wave kernel on high candidate i (index NH = the caller's stream) beside a dispatch-bound
    // launch on low candidate j; tt[i][j]: the same beside one on high candidate j.  Two signatures: the SAME hardware
    // queue (the small kernel waits for the whole launch: most of its duration), and two queues on the same compute
    // PIPE (the pipe serves its queues in turns: ~100 us instead of ~35 for every dependent kernel of a chain)
    double th[NH + 1][NL], tt[NH + 1][NH + 1], lmean = 0, base = 1e30;
    bool hl[NH + 1][NL], hh[NH + 1][NH + 1];
    bool ok = true;
    double t_us = 0, l_us = 0;
    int nl = 0;
    for (int i = 0; i <= NH && ok; ++i) {
      hipStream_t a = i < NH ? H[i] : user;
      for (int j = 0; j < NL && ok; ++j) {
        if (!L[j]) { th[i][j] = 0; continue; }
        ok = g3i_probe_pair(a, L[j], scratch, &t_us, &l_us, 2);
        th[i][j] = t_us;
        lmean += l_us; ++nl;
        if (t_us < base) base = t_us;
      }
      for (int j = 0; j <= NH && ok; ++j) {
        tt[i][j] = 0;
        if (j == i) continue;
        hipStream_t b = j < NH ? H[j] : user;
        ok = g3i_probe_pair(a, b, scratch, &t_us, &l_us, 1);
        tt[i][j] = t_us;
        if (t_us < base) base = t_us;
      }
    }
    lmean /= nl > 0 ? nl : 1;
    const double limit = base * 2.0 > base + 40.0 ? base * 2.0 : base + 40.0;
    for (int i = 0; i <= NH && ok; ++i) {
      for (int j = 0; j < NL; ++j) {
        hl[i][j] = L[j] != nullptr && th[i][j] > limit;
        if (log && L[j]) fprintf(stderr, "libg3hip placement: high %d / low %d: small kernel %.0f us beside a %.0f us dispatch-bound launch%s\n", i, j,
                         th[i][j], lmean, hl[i][j] ? (th[i][j] > 0.5 * lmean ? "  <- same queue" : "  <- same pipe") : "");
      }
      for (int j = 0; j <= NH; ++j) {
        hh[i][j] = j != i && tt[i][j] > limit;
        if (log && hh[i][j]) fprintf(stderr, "libg3hip placement: high %d / high %d: %.0f us%s\n", i, j, tt[i][j],
                                     tt[i][j] > 0.5 * lmean ? "  <- same queue" : "  <- same pipe");
      }
    }
    if (ok) {
      // chain candidates: the driver's own (0 .. NH-1) or the caller's stream (NH); fewest conflicts wins, first in order on ties
      // the two high streams together with the best plain AND the best masked bulk candidate for them: a plan may take either
      int bestc = 1 << 30;
      for (int c = own_chain ? 0 : NH; c <= (own_chain ? NH - 1 : NH); ++c)
        for (int l = 0; l < NH; ++l) {
          if (l == c) continue;
          int bp = 0, cp = 1 << 30, bm = -1, cm = 0;
          for (int b = 0; b < NLP; ++b) {
            const int k = 4 * (int)hl[c][b] + 4 * (int)hl[l][b];
            if (k < cp) { cp = k; bp = b; }
          }
          if (reserve > 0) {
            cm = 1 << 30;
            for (int b = NLP; b < NL; ++b) {
              const int k = 4 * (int)hl[c][b] + 4 * (int)hl[l][b];
              if (k < cm) { cm = k; bm = b; }
            }
          }
          const int conflicts = cp + cm + (int)(hh[c][l] || hh[l][c]);
          if (conflicts < bestc) { bestc = conflicts; kc = c; kl = l; kb = bp; kbm = bm; }
        }
      if (log) fprintf(stderr, "libg3hip placement: chain = high %d%s, look-ahead = high %d, bulk = low %d, masked bulk = %d (%d conflicts)\n", kc,
                       kc == NH ? " (caller's stream)" : "", kl, kb, kbm, bestc);
    }
  }
  if (!own_chain) kc = NH;
  D->s_chain = kc < NH ? H[kc] : nullptr;
  D->s_look = H[kl];
  D->s_bulk_alt[0] = L[kb];
  D->s_bulk_alt[1] = kbm >= 0 ? L[kbm] : nullptr;
  D->s_bulk = D->s_bulk_alt[0];
  cleanup(kc, kl, kb, kbm);
  return G3_OK;
}

// ---------------------------------------------------------------------------------------- create / destroy
extern "C" int g3_dist_unique_id(void* id_out) {
  if (!id_out) return -1;
  char err[256];
  RcclApi* api = rccl_api(err, sizeof(err));
  if (!api) return G3_ERR_HIP;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return G3_ERR_HIP;
  static_assert(sizeof(id) == G3_DIST_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &id, sizeof(id));
  return G3_OK;
}

static int dist_common(g3_ctx* ctx, int rank, int world, g3_dist** out) {
  g3_dist* D = new (std::nothrow) g3_dist();
  if (!D) return G3_ERR_NOMEM;
  D->ctx = ctx;
  D->rank = rank;
  D->world = world;
  g3_dev_guard _dg(ctx);
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least priority, hi = greatest
  // The driver's own two streams, and NOTHING else: the look-ahead and bulk contexts work on them and create no streams of
  // their own.  HIP maps streams onto a few hardware queues per priority; with the four idle streams two ordinary contexts
  // would bring, whether the chain stream and the look-ahead stream share a queue (and then serialise) depended on what
  // else the process had created before (the same replay measured 45 or 37 ms at P = 8).
  hipError_t e = hipSuccess;
  for (int i = 0, n = g3h_env_int("G3_DIST_PAD_STREAMS", 0); i < n && e == hipSuccess; ++i) {
    hipStream_t ps = nullptr;
    e = hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, (i & 1) ? lo : hi);
    if (e == hipSuccess) D->pad_streams.push_back(ps);
  }
  const bool own_chain = g3h_env_int("G3_DIST_OWN_CHAIN", 1) != 0;
  if (e == hipSuccess && pick_streams(D, lo, hi, own_chain) != G3_OK) e = hipErrorUnknown;
  if (e == hipSuccess && D->s_chain) e = hipEventCreateWithFlags(&D->ev_bracket, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&D->info_dev, sizeof(int));
  int rc = e == hipSuccess ? G3_OK : G3_ERR_HIP;
  if (!rc) rc = g3i_ctx_create_on(ctx->device, D->s_look, &D->ctx_look);
  for (int i = 0; i < 2 && !rc; ++i)
    if (D->s_bulk_alt[i]) rc = g3i_ctx_create_on(ctx->device, D->s_bulk_alt[i], &D->ctx_bulk_alt[i]);
  D->ctx_bulk = D->ctx_bulk_alt[0];
  if (rc) {
    if (D->ctx_look) g3_ctx_destroy(D->ctx_look);
    for (int i = 0; i < 2; ++i) if (D->ctx_bulk_alt[i]) g3_ctx_destroy(D->ctx_bulk_alt[i]);
    if (D->s_look) (void)hipStreamDestroy(D->s_look);
    for (int i = 0; i < 2; ++i) if (D->s_bulk_alt[i]) (void)hipStreamDestroy(D->s_bulk_alt[i]);
    if (D->s_chain) (void)hipStreamDestroy(D->s_chain);
    if (D->ev_bracket) (void)hipEventDestroy(D->ev_bracket);
    for (hipStream_t ps : D->pad_streams) (void)hipStreamDestroy(ps);
    if (D->info_dev) (void)hipFree(D->info_dev);
    delete D;
    return rc;
  }
  D->serial_coll = g3h_env_int("G3_DIST_SERIAL_COLL", 0) != 0;
  D->want_fullinv = g3h_env_int("G3_DIST_FULLINV", 1) != 0;
  {
    const char* dl = getenv("G3_DIST_DEAL");
    D->deal_snake = (dl && !strcmp(dl, "snake")) ? 1 : 0;
  }
  for (int i = 0; i < 2; ++i)
    if (D->ctx_bulk_alt[i]) D->ctx_bulk_alt[i]->bulk_role = true;     // its small-tile launches leave room on every CU for the chain's kernels (g3_gemm.hip)
  {
    const char* mm = getenv("G3_DIST_BULK_MASK_MODE");
    D->bulk_mask_mode = !mm ? 0 : (!strcmp(mm, "always") ? 1 : (!strcmp(mm, "never") ? -1 : 0));
  }
  *out = D;
  return G3_OK;
}

extern "C" int g3_dist_create(g3_ctx* ctx, const void* id_gather, const void* id_bcast, int rank, int world, g3_dist** out) {
  if (!ctx) return -1;
  if (!id_gather) return -2;
  if (!id_bcast) return -3;
  if (world < 1 || rank < 0 || rank >= world) return -4;
  if (!out) return -6;
  *out = nullptr;
  RcclApi* api = rccl_api(ctx->err, sizeof(ctx->err));
  if (!api) return G3_ERR_HIP;
  g3_dist* D = nullptr;
  int rc = dist_common(ctx, rank, world, &D);
  if (rc) return rc;
  g3_dev_guard _dg(ctx);
  RcclTransport* t = new (std::nothrow) RcclTransport();
  if (!t) { g3_dist_destroy(D); return G3_ERR_NOMEM; }
  t->api = api;
  D->tr = t;
  ncclUniqueId a, b;
  memcpy(&a, id_gather, sizeof(a));
  memcpy(&b, id_bcast, sizeof(b));
  // two communicators (same order on every rank): the broadcast of a diagonal factor must never wait in the
  // queue of the communicator that carries a 100 MB panel all-gather
  ncclResult_t r = api->CommInitRank(&t->comm_gather, world, a, rank);
  if (r == ncclSuccess) r = api->CommInitRank(&t->comm_bcast, world, b, rank);
  if (r != ncclSuccess) {
    snprintf(ctx->err, sizeof(ctx->err), "ncclCommInitRank: %s", api->GetErrorString(r));
    g3_dist_destroy(D);
    return G3_ERR_HIP;
  }
  if (hipMalloc((void**)&t->scratch, RcclTransport::SCR * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&t->hscratch, RcclTransport::SCR * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    g3_dist_destroy(D);
    return G3_ERR_HIP;
  }
  *out = D;
  return G3_OK;
}

extern "C" int g3_dist_create_callbacks(g3_ctx* ctx, const g3_dist_callbacks* cb, int rank, int world, g3_dist** out) {
  if (!ctx) return -1;
  if (!cb || !cb->bcast || !cb->allgather || !cb->allreduce) return -2;
  if (world < 1 || rank < 0 || rank >= world) return -3;
  if (!out) return -5;
  *out = nullptr;
  g3_dist* D = nullptr;
  int rc = dist_common(ctx, rank, world, &D);
  if (rc) return rc;
  CallbackTransport* t = new (std::nothrow) CallbackTransport();
  if (!t) { g3_dist_destroy(D); return G3_ERR_NOMEM; }
  t->cb = *cb;
  D->tr = t;
  *out = D;
  return G3_OK;
}

extern "C" int g3_dist_create_callbacks_async(g3_ctx* ctx, const g3_dist_host_callbacks* cb, int rank, int world, g3_dist** out) {
  if (!ctx) return -1;
  if (!cb || !cb->bcast || !cb->allgather || !cb->allreduce) return -2;
  if (world < 1 || rank < 0 || rank >= world) return -3;
  if (!out) return -5;
  *out = nullptr;
  g3_dist* D = nullptr;
  int rc = dist_common(ctx, rank, world, &D);
  if (rc) return rc;
  g3_dev_guard _dg(ctx);
  AsyncCallbackTransport* t = new (std::nothrow) AsyncCallbackTransport();
  if (!t) { g3_dist_destroy(D); return G3_ERR_NOMEM; }
  t->cb = *cb;
  t->device = ctx->device;
  t->rank = rank;
  t->world = world;
  D->tr = t;
  rc = t->start();
  if (rc) { g3_dist_destroy(D); return rc; }
  *out = D;
  return G3_OK;
}

extern "C" int g3_dist_set_keep(g3_dist* D, int on) {
  if (!D) return -1;
  if (D->replay_refs > 0) {
    snprintf(D->err, sizeof(D->err), "g3_dist_set_keep: %d replay driver(s) still read this driver's factor", D->replay_refs);
    return -2;
  }
  D->keep = on != 0;
  D->planned = false;          // the store of the block inverses is part of the plan: re-plan
  return G3_OK;
}

extern "C" int g3_dist_create_replay(g3_ctx* ctx, g3_dist* reference, int rank, int world, g3_dist** out) {
  if (!ctx) return -1;
  if (!reference || reference->world != 1 || !reference->keep || !reference->wstore || !reference->planned) return -2;
  if (world < 1 || rank < 0 || rank >= world) return -3;
  if (!out) return -5;
  *out = nullptr;
  g3_dist* D = nullptr;
  int rc = dist_common(ctx, rank, world, &D);
  if (rc) return rc;
  ReplayTransport* t = new (std::nothrow) ReplayTransport();
  if (!t) { g3_dist_destroy(D); return G3_ERR_NOMEM; }
  t->self = D;
  t->ref = reference;
  reference->replay_refs += 1;       // released by g3_dist_destroy of the replay
  D->want_fullinv = reference->fullinv;   // the replayed rank must run the schedule the reference ran
  D->tr = t;
  *out = D;
  return G3_OK;
}

static void free_plan(g3_dist* D) {
  void* bufs[] = {D->A, D->dbuf[0], D->dbuf[1], D->send[0], D->send[1], D->gath[0], D->gath[1], D->gath[2], D->avec, D->dots,
                  D->Kinv, D->alpha_dev, D->agath, D->wstore, D->blkstats, D->vstore, D->vt, D->ubuf, D->rbuf};
  D->blkstats = nullptr;
  D->vstore = D->vt = D->ubuf = D->rbuf = nullptr;
  for (void* b : bufs) if (b) (void)hipFree(b);
  D->A = D->dbuf[0] = D->dbuf[1] = D->send[0] = D->send[1] = D->gath[0] = D->gath[1] = D->gath[2] = D->avec = D->dots = nullptr;
  D->Kinv = D->alpha_dev = D->agath = nullptr;
  D->wstore = nullptr;
  D->have_inv = false;
  for (hipEvent_t e : D->ev) (void)hipEventDestroy(e);
  D->ev.clear();
  D->planned = false;
}

extern "C" int g3_dist_destroy(g3_dist* D) {
  if (!D) return -1;
  if (D->replay_refs > 0) {          // a replay copies out of this driver's buffers in every collective
    snprintf(D->err, sizeof(D->err), "g3_dist_destroy: %d replay driver(s) still read this driver's factor; destroy them first", D->replay_refs);
    return -2;
  }
  if (ReplayTransport* rt = dynamic_cast<ReplayTransport*>(D->tr))
    if (rt->ref && rt->ref->replay_refs > 0) rt->ref->replay_refs -= 1;
  g3_dev_guard _dg(D->ctx);
  (void)hipStreamSynchronize(D->ctx->stream);
  if (D->s_look) (void)hipStreamSynchronize(D->s_look);
  for (int i = 0; i < 2; ++i) if (D->s_bulk_alt[i]) (void)hipStreamSynchronize(D->s_bulk_alt[i]);
  delete D->tr;            // communicators first: their kernels are done
  free_plan(D);
  for (hipEvent_t e : D->tev) (void)hipEventDestroy(e);
  if (D->info_dev) (void)hipFree(D->info_dev);
  if (D->ctx_look) g3_ctx_destroy(D->ctx_look);
  for (int i = 0; i < 2; ++i) if (D->ctx_bulk_alt[i]) g3_ctx_destroy(D->ctx_bulk_alt[i]);
  if (D->s_look) (void)hipStreamDestroy(D->s_look);
  for (int i = 0; i < 2; ++i) if (D->s_bulk_alt[i]) (void)hipStreamDestroy(D->s_bulk_alt[i]);
  if (D->s_chain) {
    (void)hipStreamSynchronize(D->s_chain);
    g3i_ctx_forget_stream(D->ctx, D->s_chain);      // the caller's context worked on it inside every entry point
    (void)hipStreamDestroy(D->s_chain);
  }
  if (D->ev_bracket) (void)hipEventDestroy(D->ev_bracket);
  for (hipStream_t ps : D->pad_streams) (void)hipStreamDestroy(ps);
  delete D;
  return G3_OK;
}

extern "C" const char* g3_dist_last_error(g3_dist* D) { return D ? D->err : "null g3_dist"; }

extern "C" int g3_dist_plan(g3_dist* D, int64_t N, int d, int64_t M, int64_t nb, g3_dtype dt) {
  if (!D) return -1;
  if (N <= 0) return -2;
  if (d < 1 || d > G3_MAXCOLS) return -3;
  if (M < 0) return -4;
  if (nb < 128 || nb % 128) return -5;
  if (D->replay_refs > 0) {
    snprintf(D->err, sizeof(D->err), "g3_dist_plan: %d replay driver(s) still read this driver's factor", D->replay_refs);
    return -6;
  }
  g3_dev_guard _dg(D->ctx);
  G3D_HIP(hipStreamSynchronize(D->ctx->stream));
  free_plan(D);
  D->plan_gen += 1;
  D->fullinv = D->want_fullinv && nb <= 2048 && ((nb / 128) & (nb / 128 - 1)) == 0;
  D->N = N; D->d = d; D->M = M; D->nb = nb; D->dt = dt; D->es = g3_esize(dt);
  D->Np = g3_roundup(N, nb);
  D->nblk = (int)(D->Np / nb);
  {
    // Which bulk stream: the CU-masked one (32 CUs left to the chain's kernels) where the serial diagonal chain -- N / 1024
    // times ~1.5 ms (fp64) / ~1.2 ms (fp32) beside an unmasked bulk stream (measured at nb = 1024) -- would outlast the rank's own
    // N^3 / (3 P) flops at ~52 (fp64) / ~100 (fp32) TFLOP/s: config 4 on 8 ranks yes (48 against 34 ms), config 5's shape no
    // (59 against 137 ms).  Both figures from the replays of profiles/r05_replay_*; G3_DIST_BULK_MASK_MODE=always|never overrides.
    const double chain_s = (double)D->Np / 1024.0 * (dt == G3_F64 ? 1.5e-3 : 1.2e-3);   // (latency-bound: ~ proportional to nb per block)
    const double rank_s = (double)N * (double)N * (double)N / (3.0 * D->world) / (dt == G3_F64 ? 52e12 : 100e12);
    const bool want = D->bulk_mask_mode > 0 || (D->bulk_mask_mode == 0 && chain_s > rank_s);
    const int pick = (want && D->s_bulk_alt[1] && D->bulk_mask_mode >= 0) ? 1 : 0;
    D->s_bulk = D->s_bulk_alt[pick];
    D->ctx_bulk = D->ctx_bulk_alt[pick];
    D->bulk_masked = pick;
  }
  D->Mp = g3_roundup(M, 128);
  D->nchunk = 1 + (int)(D->Mp / 128);
  D->my_blocks.clear(); D->my_chunks.clear();
  g3h_deal(D->world, D->nblk, &D->owner, D->deal_snake);
  D->loff.assign(D->nblk, -1);
  for (int I = 0; I < D->nblk; ++I)
    if (owner_of(D, I) == D->rank) { D->loff[I] = (int64_t)D->my_blocks.size() * nb; D->my_blocks.push_back(I); }
  for (int c = 0; c < D->nchunk; ++c)
    if (c % D->world == D->rank) D->my_chunks.push_back(c);
  D->rows_mat = (int64_t)D->my_blocks.size() * nb;
  D->rows_rhs = (int64_t)D->my_chunks.size() * 128;
  std::vector<int32_t> idx;
  D->cmax = D->nblk > 1 ? perm_of(D, 0, &idx) : 1;
  if (D->cmax < 1) D->cmax = 1;
  D->rows_inv = 0;
  D->cmax_all = 1;
  if (D->grad) {
    D->rows_inv = D->rows_mat;
    D->cmax_all = perm_of(D, -1, &idx);       // all blocks: the panels of L^-T reach from block 0 down
    if (D->cmax < D->cmax_all) D->cmax = D->cmax_all;
  }
  const size_t rows = (size_t)(D->rows_mat + D->rows_rhs + D->rows_inv);
  G3D_HIP(hipMalloc((void**)&D->A, (rows ? rows : 1) * D->Np * D->es));
  G3D_HIP(hipMemsetAsync(D->A, 0, (rows ? rows : 1) * D->Np * D->es, D->ctx->stream));
  for (int i = 0; i < 2; ++i) {
    G3D_HIP(hipMalloc((void**)&D->dbuf[i], dbuf_bytes(D)));
    // (the blocks of V above the block diagonal are never written: keep them zero for whoever reads V densely)
    G3D_HIP(hipMemsetAsync(D->dbuf[i], 0, dbuf_bytes(D), D->ctx->stream));
    G3D_HIP(hipMalloc((void**)&D->send[i], (size_t)D->cmax * nb * nb * D->es));
    G3D_HIP(hipMemsetAsync(D->send[i], 0, (size_t)D->cmax * nb * nb * D->es, D->ctx->stream));
  }
  // three gather buffers: the bulk update with panel k may still be reading its buffer while panel k + 2 arrives
  for (int i = 0; i < 3; ++i) {
    G3D_HIP(hipMalloc((void**)&D->gath[i], (size_t)D->world * D->cmax * nb * nb * D->es));
    // ranks write their rows straight into their slot: the padding of a slot is never written, keep it finite
    G3D_HIP(hipMemsetAsync(D->gath[i], 0, (size_t)D->world * D->cmax * nb * nb * D->es, D->ctx->stream));
  }
  G3D_HIP(hipMalloc((void**)&D->avec, (size_t)D->Np * D->es));
  G3D_HIP(hipMalloc((void**)&D->dots, (D->my_chunks.size() + 1) * 2 * 128 * D->es));
  G3D_HIP(hipMalloc((void**)&D->blkstats, (D->my_blocks.size() + 1) * 4 * sizeof(double)));
  if (D->keep) G3D_HIP(hipMalloc((void**)&D->wstore, (size_t)D->Np * 128 * D->es));
  if (D->fullinv) {
    if (D->keep) G3D_HIP(hipMalloc((void**)&D->vstore, (size_t)D->Np * nb * D->es));
    G3D_HIP(hipMalloc((void**)&D->vt, (size_t)nb * nb * D->es));
    G3D_HIP(hipMalloc((void**)&D->ubuf, (size_t)nb * nb * D->es));
    G3D_HIP(hipMemsetAsync(D->vt, 0, (size_t)nb * nb * D->es, D->ctx->stream));
    G3D_HIP(hipMemsetAsync(D->ubuf, 0, (size_t)nb * nb * D->es, D->ctx->stream));
    const size_t rr = (size_t)(D->rows_rhs + D->rows_inv);
    G3D_HIP(hipMalloc((void**)&D->rbuf, (rr ? rr : 1) * nb * D->es));
  }
  if (D->grad) {
    G3D_HIP(hipMalloc((void**)&D->Kinv, (size_t)(D->rows_mat ? D->rows_mat : 1) * D->Np * D->es));
    G3D_HIP(hipMalloc((void**)&D->alpha_dev, (size_t)D->Np * D->es));
    G3D_HIP(hipMalloc((void**)&D->agath, (size_t)D->world * D->cmax_all * nb * D->es));
  }
  if (ReplayTransport* rt = dynamic_cast<ReplayTransport*>(D->tr)) {
    if (rt->ref->Np != D->Np || rt->ref->nb != D->nb || rt->ref->dt != D->dt || rt->ref->M != D->M || D->grad ||
        rt->ref->fullinv != D->fullinv || !rt->ref->planned || !rt->ref->keep) {
      snprintf(D->err, sizeof(D->err), "replay: the plan must repeat the reference's (N, M, nb, dtype, panel-solve mode) and gradient "
               "mode is not replayed");
      return -6;
    }
    rt->ref_gen = rt->ref->plan_gen;
  }
  D->ev.resize(D->nblk + 6);
  for (auto& e : D->ev) G3D_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  G3D_HIP(hipStreamSynchronize(D->ctx->stream));
  D->planned = true;
  return G3_OK;
}

// ---------------------------------------------------------------------------------------- build
// stream `to` waits for everything queued so far on `from`
static int stream_after(g3_dist* D, hipStream_t to, hipStream_t from, hipEvent_t ev) {
  G3D_HIP(hipEventRecord(ev, from));
  G3D_HIP(hipStreamWaitEvent(to, ev, 0));
  return G3_OK;
}

static int rhs_rows(g3_dist* D, int t, int chunk, const g3_kernel_prog* prog_cross, const void* X, int64_t ldx, const void* Xs,
                    int64_t ldxs, const void* delta) {
  // chunk 0 = [delta; 0 ...]; chunk c >= 1 = tt_to_num(K(Xs[(c-1)*128 : c*128], X)), zero padded
  char* out = Aat(D, D->rows_mat + (int64_t)t * 128, 0);
  G3D_HIP(hipMemsetAsync(out, 0, (size_t)128 * D->Np * D->es, D->ctx->stream));
  if (chunk == 0) {
    G3D_HIP(hipMemcpyAsync(out, delta, (size_t)D->N * D->es, hipMemcpyDeviceToDevice, D->ctx->stream));
    return G3_OK;
  }
  const int64_t s0 = (int64_t)(chunk - 1) * 128;
  const int64_t m = D->M - s0 < 128 ? D->M - s0 : 128;
  if (m > 0)
    G3D_RC(D->ctx, g3_gram(D->ctx, prog_cross, (const char*)Xs + (size_t)s0 * ldxs * D->es, m, ldxs, X, D->N, ldx, D->d, D->dt, out,
                           D->Np, m, D->N, G3_GRAM_SCRUB));
  return G3_OK;
}

static int build(g3_dist* D, const g3_kernel_prog* prog, const g3_kernel_prog* prog_cross, const void* X, int64_t ldx,
                 const void* Xs, int64_t ldxs, const void* delta, double jitter) {
  const int64_t nb = D->nb;
  double lmin = INFINITY;
  {
    // the diagonal minimum of each owned block stays on the device; one copy fetches them all
    int cnt = 0;
    for (int I : D->my_blocks) {
      G3D_RC(D->ctx, g3_gram_rows(D->ctx, prog, X, D->N, ldx, D->d, (int64_t)I * nb, nb, D->dt, Aat(D, D->loff[I], 0), D->Np,
                                  G3_GRAM_SCRUB | G3_GRAM_PAD_EYE));
      const int64_t nv = D->N - (int64_t)I * nb < nb ? D->N - (int64_t)I * nb : nb;
      if (nv > 0) {
        G3D_RC(D->ctx, g3i_diag_stats_dev(D->ctx, Aat(D, D->loff[I], (int64_t)I * nb), nv, D->Np, D->dt, D->blkstats + 4 * cnt));
        ++cnt;
      }
    }
    if (cnt > 0) {
      std::vector<double> hs((size_t)4 * cnt);
      G3D_HIP(hipMemcpyAsync(hs.data(), D->blkstats, hs.size() * sizeof(double), hipMemcpyDeviceToHost, D->ctx->stream));
      G3D_HIP(hipStreamSynchronize(D->ctx->stream));
      for (int i = 0; i < cnt; ++i)
        if (hs[4 * i] < lmin || hs[4 * i] != hs[4 * i]) lmin = hs[4 * i];
    }
  }
  for (size_t t = 0; t < D->my_chunks.size(); ++t) {
    int rc = rhs_rows(D, (int)t, D->my_chunks[t], prog_cross, X, ldx, Xs, ldxs, delta);
    if (rc) return rc;
  }
  if (D->rows_inv > 0) {     // the identity, row block I of it stored like block I of the matrix
    G3D_HIP(hipMemsetAsync(Aat(D, D->rows_mat + D->rows_rhs, 0), 0, (size_t)D->rows_inv * D->Np * D->es, D->ctx->stream));
    for (int I : D->my_blocks)
      G3D_RC(D->ctx, g3_diag_add(D->ctx, Aat(D, D->rows_mat + D->rows_rhs + D->loff[I], (int64_t)I * nb), nb, D->Np, D->dt, 1.0));
  }
  // tt_to_cov (tensors.py:95-98): min over the WHOLE diagonal
  double g = isfinite(lmin) ? lmin : 1e300;
  int rc = do_allreduce(D, &g, 1, 1);
  if (rc) return rc;
  double add = jitter;
  if (!(g > 0)) add += (double)1e-6f - g;
  if (add != 0.0)
    for (int I : D->my_blocks) {
      const int64_t nv = D->N - (int64_t)I * nb < nb ? D->N - (int64_t)I * nb : nb;
      if (nv > 0) G3D_RC(D->ctx, g3_diag_add(D->ctx, Aat(D, D->loff[I], (int64_t)I * nb), nv, D->Np, D->dt, add));
    }
  return G3_OK;
}

// ---------------------------------------------------------------------------------------- the sweep
static int factor_block(g3_dist* D, g3_ctx* cx, int k) {
  // owner only: factor diagonal block k into the broadcast buffer; keep the factor in place for the log-determinant
  char* Dk = Aat(D, D->loff[k], (int64_t)k * D->nb);
  G3D_RC(cx, g3_copy2d(cx, Lof(D, k), D->nb, Dk, D->Np, D->nb, D->nb, D->dt));
  G3D_RC(cx, g3_potrf_nowait(cx, Lof(D, k), D->nb, D->nb, D->dt, Wof(D, k), D->info_dev));
  G3D_RC(cx, g3_copy2d(cx, Dk, D->Np, Lof(D, k), D->nb, D->nb, D->nb, D->dt));
  // V_k = L_kk^-1 for everybody's panel solve (a failed pivot leaves the flag of cx set: the launches below are no-ops)
  if (D->fullinv) G3D_RC(cx, g3i_trtri_full(cx, Lof(D, k), D->nb, Wof(D, k), Vof(D, k), D->vt, D->ubuf, D->dt));
  if (D->keep && D->wstore)
    G3D_HIP(hipMemcpyAsync(D->wstore + (size_t)k * D->nb * 128 * D->es, Wof(D, k), (size_t)D->nb * 128 * D->es, hipMemcpyDeviceToDevice,
                           cx->stream));
  if (D->keep && D->vstore)
    G3D_HIP(hipMemcpyAsync(D->vstore + (size_t)k * D->nb * D->nb * D->es, Vof(D, k), (size_t)D->nb * D->nb * D->es, hipMemcpyDeviceToDevice,
                           cx->stream));
  return G3_OK;
}

static int64_t rows_done(const g3_dist* D, int k) {   // local rows of the blocks <= k
  int64_t c = 0;
  for (int I : D->my_blocks) if (I <= k) c += D->nb;
  return c;
}

// gradient mode: identity row block I is zero left of column block I, so it joins the sweep at panel I -- the active
// identity rows at panel k are the local rows of the blocks <= k, a prefix of the identity region
static int64_t inv_active(const g3_dist* D, int k) { return D->rows_inv > 0 ? rows_done(D, k) : 0; }

// panel k: solve my rows below block k (right-hand-side rows included) against L_kk, then all-gather them into
// gath[k % 3] -- all on the chain stream, which has nothing else to do until the panel is there
static int solve_and_gather(g3_dist* D, int k, hipEvent_t ev_solved) {
  const int64_t nb = D->nb, c0 = (int64_t)k * nb;
  const int64_t r_lo = rows_done(D, k);
  const int64_t m = D->rows_mat + D->rows_rhs - r_lo + inv_active(D, k);
  // The look-ahead of block k+1 needs ITS rows of the panel, not the gathered panel and not my other rows: the owner of
  // block k+1 solves those nb rows first and releases the look-ahead (update + factorisation + broadcast of block k+1, the
  // serial cycle every rank waits on) before it solves the rest -- the one-GPU sweep orders its panel solve the same way.
  // Row-wise independent solves: same arithmetic per row whichever launch carries it.
  const bool own_next = k + 1 < D->nblk && owner_of(D, k + 1) == D->rank;
  const int64_t head = (own_next && m > nb) ? nb : 0;
  bool released = false;
  const int64_t mine = D->rows_mat - r_lo;          // my matrix rows below block k (m - mine right-hand-side rows follow them)
  const bool gathers = D->nblk - 1 - k > 0;
  std::vector<int32_t> idx;
  const int cnt = gathers ? perm_of(D, k, &idx) : 0;
  // my rows go straight into my slot of the gather buffer and the all-gather runs in place (send = recv + rank * bytes):
  // no staging copy, and RCCL skips the local block -- at one rank the collective moves nothing at all
  const size_t gbytes = (size_t)cnt * nb * nb * D->es;
  char* slot = gathers ? D->gath[k % 3] + (size_t)D->rank * gbytes : nullptr;
  if (m > 0) {
    int rcp = coll_begin(D, G3_PH_SOLVE, D->ctx->stream, 0.0);
    if (rcp) return rcp;
    if (D->fullinv) {
      // X <- X V_k^T, out of place: the matrix rows land in my slot of the gather buffer (and are copied back into the
      // local matrix, which the staircase launches and the look-ahead read), the right-hand-side rows go through rbuf
      const char* V = Vof(D, k);
      if (head > 0) {
        G3D_RC(D->ctx, g3i_gemm_nt_ktri(D->ctx, slot, nb, Aat(D, r_lo, c0), D->Np, V, nb, head, nb, 1.0, 0.0, D->dt));
        G3D_RC(D->ctx, g3_copy2d(D->ctx, Aat(D, r_lo, c0), D->Np, slot, nb, head, nb, D->dt));
        G3D_HIP(hipEventRecord(ev_solved, D->ctx->stream));
        released = true;
      }
      if (mine - head > 0)
        G3D_RC(D->ctx, g3i_gemm_nt_ktri(D->ctx, slot + (size_t)head * nb * D->es, nb, Aat(D, r_lo + head, c0), D->Np, V, nb, mine - head, nb,
                                        1.0, 0.0, D->dt));
      if (m - mine > 0)
        G3D_RC(D->ctx, g3i_gemm_nt_ktri(D->ctx, D->rbuf, nb, Aat(D, D->rows_mat, c0), D->Np, V, nb, m - mine, nb, 1.0, 0.0, D->dt));
      if (mine - head > 0)
        G3D_RC(D->ctx, g3_copy2d(D->ctx, Aat(D, r_lo + head, c0), D->Np, slot + (size_t)head * nb * D->es, nb, mine - head, nb, D->dt));
      if (m - mine > 0) G3D_RC(D->ctx, g3_copy2d(D->ctx, Aat(D, D->rows_mat, c0), D->Np, D->rbuf, nb, m - mine, nb, D->dt));
    } else {
      if (head > 0) {
        G3D_RC(D->ctx, g3_trsm_rlt(D->ctx, Lof(D, k), nb, nb, Aat(D, r_lo, c0), head, D->Np, D->dt, Wof(D, k)));
        G3D_HIP(hipEventRecord(ev_solved, D->ctx->stream));
        released = true;
      }
      G3D_RC(D->ctx, g3_trsm_rlt(D->ctx, Lof(D, k), nb, nb, Aat(D, r_lo + head, c0), m - head, D->Np, D->dt, Wof(D, k)));
      if (mine > 0) G3D_RC(D->ctx, g3_copy2d(D->ctx, slot, nb, Aat(D, r_lo, c0), D->Np, mine, nb, D->dt));
    }
    rcp = coll_end(D, D->ctx->stream);
    if (rcp) return rcp;
  }
  // (ranks that do not own block k+1: the event only orders the broadcast buffer's reuse behind this solve)
  if (!released) G3D_HIP(hipEventRecord(ev_solved, D->ctx->stream));
  if (!gathers) return G3_OK;
  int rc = do_allgather(D, slot, D->gath[k % 3], gbytes, D->ctx->stream, G3_HINT_PANEL, k);
  if (rc) return rc;
  if (D->serial_coll) G3D_HIP(hipEventRecord(D->ev[D->nblk + 5], D->ctx->stream));    // "panel k is gathered"
  return G3_OK;
}

// diagonal block j on the look-ahead stream: its owner applies the update with panel j-1 from its own panel rows
// (the earlier panels arrived with the bulk stream's column launches: event `after`), factors and broadcasts;
// the others post the receive.  `ev_solved`: panel j-1 is solved on this rank (recorded by solve_and_gather BEFORE
// its all-gather: the factorisation of block j overlaps the exchange of panel j-1; it also says that the chain has
// finished reading the broadcast buffer this block will overwrite).  `joined`: event recorded behind the broadcast,
// the chain waits for it before it uses dbuf[j % 2]
static int lookahead(g3_dist* D, int j, hipEvent_t after, hipEvent_t ev_solved, hipEvent_t joined) {
  const int64_t nb = D->nb;
  int rc = G3_OK;
  G3D_HIP(hipStreamWaitEvent(D->s_look, ev_solved, 0));
  if (owner_of(D, j) == D->rank) {
    if (after) G3D_HIP(hipStreamWaitEvent(D->s_look, after, 0));
    const int64_t lo = D->loff[j];
    char* Pn = Aat(D, lo, (int64_t)(j - 1) * nb);
    rc = coll_begin(D, G3_PH_DIAG, D->s_look, 0.0);
    if (rc) return rc;
    G3D_RC(D->ctx_look, g3_gemm_nt(D->ctx_look, Aat(D, lo, (int64_t)j * nb), D->Np, Pn, D->Np, Pn, D->Np, nb, nb, nb, -1.0, 1.0, D->dt, 1));
    rc = factor_block(D, D->ctx_look, j);
    if (rc) return rc;
    rc = coll_end(D, D->s_look);
    if (rc) return rc;
  }
  // Default: the broadcast of block j (its own communicator, this stream) is in flight together with the all-gather of
  // panel j-1 (the other communicator, chain stream) -- what hides the exchange behind the factorisation.  Two RCCL
  // communicators used concurrently from one device have never run with more than one rank here (a one-GPU box);
  // G3_DIST_SERIAL_COLL=1 is the conservative schedule: the broadcast waits until the gather of panel j-1 has been
  // issued AND completed on this rank, so every rank issues the two communicators strictly one after the other.
  if (D->serial_coll && j >= 1 && D->nblk - j > 0) G3D_HIP(hipStreamWaitEvent(D->s_look, D->ev[D->nblk + 5], 0));
  rc = do_bcast(D, bc_ptr(D, j), bc_bytes(D), owner_of(D, j), D->s_look, G3_HINT_DIAG, j);
  if (rc) return rc;
  G3D_HIP(hipEventRecord(joined, D->s_look));
  return G3_OK;
}

static int stair_ptr(g3_dist* D, g3_ctx* cx, char* C, const char* A, int64_t lda, const char* G, const std::vector<int64_t>& seg_rows,
                     const std::vector<int64_t>& seg_cols, const int32_t* perm, int nperm, const std::vector<int64_t>* seg_diag,
                     double alpha) {
  // C[rows_s, 0 : seg_cols[s]) += alpha * A[rows_s, 0 : nb) G[block table]^T (C full-width local rows, ld = Np; A with leading
  // dimension lda), cut into launches of at most G3H_STAIR_MAX row segments / blocks of G (g3_host.h)
  std::vector<G3hStairChunk> ch;
  g3h_stair_chunks(seg_rows, seg_cols, D->nb, nperm, &ch, seg_diag, D->ctx->tune.stair_max);
  for (const auto& c : ch) {
    G3D_RC(cx, g3_gemm_nt_stair(cx, C + ((size_t)c.row0 * D->Np + c.col0) * D->es, D->Np, A + (size_t)c.row0 * lda * D->es, lda, G,
                                D->nb, D->nb, c.rows.data(), c.cols.data(), (int)c.rows.size(), alpha, 1.0, D->dt, D->nb, perm + c.blk0,
                                c.nblk, seg_diag ? c.diag.data() : nullptr));
  }
  return G3_OK;
}
// C[rows_s, col0 : col0 + seg_cols[s]) -= A[rows_s, kcol : kcol + nb) G[block table]^T inside the local matrix.
// (Measured and not kept, round 4: taking A from this rank's slot of the gathered panel -- both operands from one compact
//  buffer, as the one-GPU sweep takes both from the panel column -- gains 0.8 ms of 210 at N = 32768, but the right-hand-side
//  rows are not part of the exchange and need their own launch then, which costs 3.5 ms.)
static int stair(g3_dist* D, g3_ctx* cx, int64_t row0, int64_t col0, int64_t kcol, const char* G, const std::vector<int64_t>& seg_rows,
                 const std::vector<int64_t>& seg_cols, const int32_t* perm, int nperm, const std::vector<int64_t>* seg_diag = nullptr) {
  return stair_ptr(D, cx, Aat(D, row0, col0), Aat(D, row0, kcol), D->Np, G, seg_rows, seg_cols, perm, nperm, seg_diag, -1.0);
}

static int factor(g3_dist* D, const g3_kernel_prog* prog, const g3_kernel_prog* prog_cross, const void* X, int64_t ldx, const void* Xs,
                  int64_t ldxs, const void* delta, double jitter, int* info_out) {
  const int64_t nb = D->nb;
  const int nblk = D->nblk;
  hipStream_t sA = D->ctx->stream, sB = D->s_bulk;
  hipEvent_t ev_tmp = D->ev[nblk], ev_join = D->ev[nblk + 1], ev_bc[2] = {D->ev[nblk + 2], D->ev[nblk + 3]}, ev_sv = D->ev[nblk + 4];
  int rc = build(D, prog, prog_cross, X, ldx, Xs, ldxs, delta, jitter);
  if (rc) return rc;
  G3D_HIP(hipMemsetAsync(D->info_dev, 0, sizeof(int), sA));
  if (owner_of(D, 0) == D->rank) {
    rc = coll_begin(D, G3_PH_DIAG, sA, 0.0);
    if (!rc) rc = factor_block(D, D->ctx, 0);
    if (!rc) rc = coll_end(D, sA);
    if (rc) return rc;
  }
  rc = do_bcast(D, bc_ptr(D, 0), bc_bytes(D), owner_of(D, 0), sA, G3_HINT_DIAG, 0);
  if (rc) return rc;
  rc = solve_and_gather(D, 0, ev_sv);
  if (rc) return rc;
  if (nblk > 1) {
    rc = lookahead(D, 1, nullptr, ev_sv, ev_bc[1]);
    if (rc) return rc;
  }
  hipEvent_t ev_prev = nullptr;
  std::vector<int32_t> perm;
  std::vector<int64_t> seg_rows, seg_cols, seg_diag;
  for (int k = 0; k + 1 < nblk; ++k) {
    const int64_t c0 = (int64_t)k * nb, c1 = c0 + nb, c2 = c1 + nb, c3 = c2 + nb;
    perm_of(D, k, &perm);
    const char* G = D->gath[k % 3];
    hipEvent_t ev_k = nullptr;
    const int64_t rr = D->rows_rhs + inv_active(D, k);     // right-hand-side rows panel k applies to
    // ---- bulk stream: everything beyond block column k+1 (after the gather of panel k, queued on the chain)
    rc = stream_after(D, sB, sA, ev_tmp);
    if (rc) return rc;
    if (k + 2 < nblk) {
      // d1. block column k+2 (block k+2's diagonal block included: the look-ahead adds panel k+1 only)
      seg_rows.clear(); seg_cols.clear(); seg_diag.clear();
      int64_t lo = -1;
      for (int I : D->my_blocks) if (I >= k + 2) { if (lo < 0) lo = D->loff[I]; seg_rows.push_back(nb); seg_cols.push_back(nb); seg_diag.push_back(I == k + 2); }
      if (rr > 0) { if (lo < 0) lo = D->rows_mat; seg_rows.push_back(rr); seg_cols.push_back(nb); seg_diag.push_back(0); }
      if (!seg_rows.empty()) {
        rc = stair(D, D->ctx_bulk, lo, c2, c0, G, seg_rows, seg_cols, perm.data() + 1, 1, &seg_diag);
        if (rc) return rc;
      }
      ev_k = D->ev[k];
      G3D_HIP(hipEventRecord(ev_k, sB));
      // d2. the rest: block columns >= k+3 of my blocks >= k+3 and of the right-hand-side rows
      seg_rows.clear(); seg_cols.clear(); seg_diag.clear();
      lo = -1;
      for (int I : D->my_blocks) if (I >= k + 3) { if (lo < 0) lo = D->loff[I]; seg_rows.push_back(nb); seg_cols.push_back((int64_t)(I - k - 2) * nb); seg_diag.push_back(1); }
      if (rr > 0 && nblk - k - 3 > 0) { if (lo < 0) lo = D->rows_mat; seg_rows.push_back(rr); seg_cols.push_back((int64_t)(nblk - k - 3) * nb); seg_diag.push_back(0); }
      if (!seg_rows.empty()) {
        rc = stair(D, D->ctx_bulk, lo, c3, c0, G, seg_rows, seg_cols, perm.data() + 2, (int)perm.size() - 2, &seg_diag);
        if (rc) return rc;
      }
    }
    // ---- chain: a. block column k+1 (it carries the updates up to panel k-1 once B_{k-1} has fired)
    if (ev_prev) G3D_HIP(hipStreamWaitEvent(sA, ev_prev, 0));
    {
      seg_rows.clear(); seg_cols.clear();
      int64_t lo = -1;
      for (int I : D->my_blocks) if (I >= k + 2) { if (lo < 0) lo = D->loff[I]; seg_rows.push_back(nb); seg_cols.push_back(nb); }
      if (rr > 0) { if (lo < 0) lo = D->rows_mat; seg_rows.push_back(rr); seg_cols.push_back(nb); }
      if (!seg_rows.empty()) {
        rc = stair(D, D->ctx, lo, c1, c0, G, seg_rows, seg_cols, perm.data(), 1);
        if (rc) return rc;
      }
    }
    // b. panel k+1: needs the broadcast factor
    G3D_HIP(hipStreamWaitEvent(sA, ev_bc[(k + 1) % 2], 0));
    rc = solve_and_gather(D, k + 1, ev_sv);
    if (rc) return rc;
    // c. diagonal block k+2
    if (k + 2 < nblk) {
      rc = lookahead(D, k + 2, ev_k, ev_sv, ev_bc[k % 2]);
      if (rc) return rc;
    }
    ev_prev = ev_k;
  }
  rc = stream_after(D, sA, sB, ev_join);
  if (rc) return rc;
  rc = stream_after(D, sA, D->s_look, ev_join);
  if (rc) return rc;
  int info = 0;
  G3D_HIP(hipMemcpyAsync(D->ctx->h_info, D->info_dev, sizeof(int), hipMemcpyDeviceToHost, sA));
  G3D_HIP(hipStreamSynchronize(sA));
  info = D->ctx->h_info[0];
  double v = (double)info;
  rc = do_allreduce(D, &v, 1, 2);
  if (rc) return rc;
  *info_out = (int)v;
  return G3_OK;
}

// CholeskyRobust's schedule (tensors.py:197-222) around the distributed factorisation
static int factor_robust(g3_dist* D, const g3_kernel_prog* prog, const g3_kernel_prog* prog_cross, const void* X, int64_t ldx,
                         const void* Xs, int64_t ldxs, const void* delta) {
  int info = 0;
  int rc = factor(D, prog, prog_cross, X, ldx, Xs, ldxs, delta, 0.0, &info);
  if (rc) return rc;
  int tries = 0, fallback = 0;
  if (info != 0) {
    // jitter from the diagonal of the (lifted) covariance: rebuild and reduce mean / min
    rc = build(D, prog, prog_cross, X, ldx, Xs, ldxs, delta, 0.0);
    if (rc) return rc;
    double acc[2] = {0.0, 0.0}, mn = INFINITY;
    for (int I : D->my_blocks) {
      const int64_t nv = D->N - (int64_t)I * D->nb < D->nb ? D->N - (int64_t)I * D->nb : D->nb;
      if (nv > 0) {
        double st[3];
        G3D_RC(D->ctx, g3_diag_stats(D->ctx, Aat(D, D->loff[I], (int64_t)I * D->nb), nv, D->Np, D->dt, st));
        if (st[0] < mn) mn = st[0];
        acc[0] += st[1] * (double)nv;
        acc[1] += (double)nv;
      }
    }
    rc = do_allreduce(D, acc, 2, 0);
    if (rc) return rc;
    double gmin = isfinite(mn) ? mn : 1e300;
    rc = do_allreduce(D, &gmin, 1, 1);
    if (rc) return rc;
    const double mean = acc[0] / (acc[1] > 1.0 ? acc[1] : 1.0);
    G3hJitter jit(mean, gmin);
    bool ok = false;
    for (int t = 0; t < G3hJitter::max_tries(); ++t) {
      ++tries;
      int inf2 = 0;
      rc = factor(D, prog, prog_cross, X, ldx, Xs, ldxs, delta, jit.value(), &inf2);
      if (rc) return rc;
      if (inf2 == 0) { ok = true; break; }
      jit.next();
    }
    if (!ok) {
      // CholeskyRobust.perform never raises: the factor becomes 1e-10 * I (tensors.py:215-222), identity on the
      // padding; the right-hand-side rows are rebuilt and solved against it (a division)
      fallback = 1;
      const double c = (double)1e-10f;
      if (D->rows_mat > 0) G3D_HIP(hipMemsetAsync(D->A, 0, (size_t)D->rows_mat * D->Np * D->es, D->ctx->stream));
      for (int I : D->my_blocks) {
        const int64_t nv0 = D->N - (int64_t)I * D->nb;
        const int64_t nv = nv0 < 0 ? 0 : (nv0 < D->nb ? nv0 : D->nb);
        char* Dg = Aat(D, D->loff[I], (int64_t)I * D->nb);
        if (nv > 0) G3D_RC(D->ctx, g3_diag_add(D->ctx, Dg, nv, D->Np, D->dt, c));
        if (nv < D->nb) G3D_RC(D->ctx, g3_diag_add(D->ctx, Dg + ((size_t)nv * D->Np + nv) * D->es, D->nb - nv, D->Np, D->dt, 1.0));
      }
      for (size_t t = 0; t < D->my_chunks.size(); ++t) {
        rc = rhs_rows(D, (int)t, D->my_chunks[t], prog_cross, X, ldx, Xs, ldxs, delta);
        if (rc) return rc;
      }
      if (D->rows_rhs > 0)
        G3D_RC(D->ctx, g3i_scale(D->ctx, Aat(D, D->rows_mat, 0), D->rows_rhs, D->N, D->Np, D->dt, 1.0 / c));
      if (D->rows_inv > 0) {     // L^-T of the fallback factor: 1e10 on the valid diagonal, 1 on the padding
        G3D_HIP(hipMemsetAsync(Aat(D, D->rows_mat + D->rows_rhs, 0), 0, (size_t)D->rows_inv * D->Np * D->es, D->ctx->stream));
        for (int I : D->my_blocks) {
          const int64_t nv0 = D->N - (int64_t)I * D->nb;
          const int64_t nv = nv0 < 0 ? 0 : (nv0 < D->nb ? nv0 : D->nb);
          char* Dg = Aat(D, D->rows_mat + D->rows_rhs + D->loff[I], (int64_t)I * D->nb);
          if (nv > 0) G3D_RC(D->ctx, g3_diag_add(D->ctx, Dg, nv, D->Np, D->dt, 1.0 / c));
          if (nv < D->nb) G3D_RC(D->ctx, g3_diag_add(D->ctx, Dg + ((size_t)nv * D->Np + nv) * D->es, D->nb - nv, D->Np, D->dt, 1.0));
        }
      }
    }
  }
  D->have_inv = D->grad;
  D->last_info = info;
  D->last_tries = tries;
  D->last_fallback = fallback;
  return G3_OK;
}

// (logdet, quad, mean pieces[M], ss[M]) summed over all ranks
static int stats(g3_dist* D, double* logdet, double* quad, double* mean, double* ss) {
  const int64_t nb = D->nb, M = D->M, Np = D->Np;
  std::vector<double> acc(2 + 2 * M, 0.0);
  int nstat = 0;
  for (int I : D->my_blocks) {
    const int64_t nv0 = D->N - (int64_t)I * nb;
    const int64_t nv = nv0 < 0 ? 0 : (nv0 < nb ? nv0 : nb);
    if (nv > 0) {
      G3D_RC(D->ctx, g3i_logp_terms_dev(D->ctx, Aat(D, D->loff[I], (int64_t)I * nb), nv, Np, nullptr, D->dt, D->blkstats + 4 * nstat));
      ++nstat;
    }
  }
  std::vector<double> hstat((size_t)4 * nstat + 1);
  if (nstat > 0)      // fetched by the synchronisation below (rank 0: with a; the others: with the first chunk or the all-reduce)
    G3D_HIP(hipMemcpyAsync(hstat.data(), D->blkstats, (size_t)4 * nstat * sizeof(double), hipMemcpyDeviceToHost, D->ctx->stream));
  // a = L^-1 delta is row 0 of right-hand-side chunk 0 (rank 0 holds it); everyone needs it for V a
  const int own0 = 0 % D->world;
  if (D->rank == own0) G3D_HIP(hipMemcpyAsync(D->avec, Aat(D, D->rows_mat, 0), (size_t)Np * D->es, hipMemcpyDeviceToDevice, D->ctx->stream));
  int rc = do_bcast(D, D->avec, (size_t)Np * D->es, own0, D->ctx->stream, G3_HINT_AVEC, 0);
  if (rc) return rc;
  if (D->rank == own0) {     // a^T a in double on the host (Np <= a few 10^5 values)
    std::vector<char> ha((size_t)Np * D->es);
    G3D_HIP(hipMemcpyAsync(ha.data(), D->avec, (size_t)Np * D->es, hipMemcpyDeviceToHost, D->ctx->stream));
    G3D_HIP(hipStreamSynchronize(D->ctx->stream));
    double q = 0.0;
    for (int64_t i = 0; i < Np; ++i) {
      const double v = D->dt == G3_F64 ? ((const double*)ha.data())[i] : (double)((const float*)ha.data())[i];
      q += v * v;
    }
    acc[1] = q;
  }
  // V a and the row sums of squares of every chunk of mine, then ONE copy and ONE synchronisation for all of them
  const size_t pair = 2 * 128 * D->es;
  std::vector<char> hb((D->my_chunks.size() + 1) * pair);
  for (size_t t = 0; t < D->my_chunks.size(); ++t) {
    const int c = D->my_chunks[t];
    if (c == 0) continue;
    const int64_t s0 = (int64_t)(c - 1) * 128;
    const int64_t m = M - s0 < 128 ? M - s0 : 128;
    if (m <= 0) continue;
    char* dd = D->dots + t * pair;
    G3D_RC(D->ctx, g3_rows_dot_ss(D->ctx, Aat(D, D->rows_mat + (int64_t)t * 128, 0), m, Np, Np, D->avec, D->dt, dd, dd + 128 * D->es));
  }
  if (!D->my_chunks.empty())
    G3D_HIP(hipMemcpyAsync(hb.data(), D->dots, D->my_chunks.size() * pair, hipMemcpyDeviceToHost, D->ctx->stream));
  G3D_HIP(hipStreamSynchronize(D->ctx->stream));
  for (int i = 0; i < nstat; ++i) acc[0] += hstat[4 * i];
  for (size_t t = 0; t < D->my_chunks.size(); ++t) {
    const int c = D->my_chunks[t];
    if (c == 0) continue;
    const int64_t s0 = (int64_t)(c - 1) * 128;
    const int64_t m = M - s0 < 128 ? M - s0 : 128;
    const char* h = hb.data() + t * pair;
    for (int64_t i = 0; i < m; ++i) {
      if (D->dt == G3_F64) {
        acc[2 + s0 + i] += ((const double*)h)[i];
        acc[2 + M + s0 + i] += ((const double*)h)[128 + i];
      } else {
        acc[2 + s0 + i] += (double)((const float*)h)[i];
        acc[2 + M + s0 + i] += (double)((const float*)h)[128 + i];
      }
    }
  }
  rc = do_allreduce(D, acc.data(), (int)acc.size(), 0);
  if (rc) return rc;
  *logdet = acc[0];
  *quad = acc[1];
  for (int64_t i = 0; i < M; ++i) { mean[i] = acc[2 + i]; ss[i] = acc[2 + M + i]; }
  return G3_OK;
}

// packed host copy of the M x d prediction points (a few tens of KB)
static int fetch_xs(g3_dist* D, const void* Xs_dev, int64_t ldxs, std::vector<char>* out) {
  out->resize((size_t)D->M * D->d * D->es);
  if (out->empty()) return G3_OK;
  G3D_HIP(hipMemcpy2DAsync(out->data(), (size_t)D->d * D->es, Xs_dev, (size_t)ldxs * D->es, (size_t)D->d * D->es, (size_t)D->M,
                           hipMemcpyDeviceToHost, D->ctx->stream));
  G3D_HIP(hipStreamSynchronize(D->ctx->stream));
  return G3_OK;
}
static int same_xs_as_evaluated(g3_dist* D, const void* Xs_dev, int64_t ldxs) {
  std::vector<char> now;
  int rc = fetch_xs(D, Xs_dev, ldxs, &now);
  if (rc) return rc;
  if (now.size() != D->xs_seen.size() || (now.size() && memcmp(now.data(), D->xs_seen.data(), now.size()) != 0)) {
    snprintf(D->err, sizeof(D->err), "the cross-solve rows in the driver belong to other prediction points: evaluate "
             "g3_dist_gp_factor_predict with these Xs first");
    return -3;
  }
  return G3_OK;
}

extern "C" int g3_dist_gp_factor_predict(g3_dist* D, const g3_kernel_prog* prog, const g3_kernel_prog* prog_cross, const void* X_dev,
                                         int64_t ldx, const void* delta_dev, const void* Xs_dev, int64_t ldxs, double out_host[5],
                                         double* mean_host, double* ss_host) {
  if (!D) return -1;
  if (!D->planned) return -1;
  if (!prog) return -2;
  if (!prog_cross) return -3;
  if (!X_dev) return -4;
  if (ldx < D->d) return -5;
  if (!delta_dev) return -6;
  if (D->M > 0 && !Xs_dev) return -7;
  if (D->M > 0 && ldxs < D->d) return -8;
  if (!out_host) return -9;
  if (D->M > 0 && (!mean_host || !ss_host)) return -10;
  if (g3i_validate_prog(prog, D->d) || g3i_validate_prog(prog_cross, D->d)) return -2;
  g3_dev_guard _dg(D->ctx);
  ChainScope _cs(D);
  int rc = factor_robust(D, prog, prog_cross, X_dev, ldx, Xs_dev, ldxs, delta_dev);
  if (rc) return mark_failed(D, rc);
  std::vector<double> mm(D->M > 0 ? D->M : 1), sv(D->M > 0 ? D->M : 1);
  double logdet = 0, quad = 0;
  rc = stats(D, &logdet, &quad, mm.data(), sv.data());
  if (rc) return mark_failed(D, rc);
  out_host[0] = logdet;
  out_host[1] = quad;
  out_host[2] = (double)D->last_info;
  out_host[3] = (double)D->last_tries;
  out_host[4] = (double)D->last_fallback;
  for (int64_t i = 0; i < D->M; ++i) { mean_host[i] = mm[i]; ss_host[i] = sv[i]; }
  if (D->M > 0) {
    rc = fetch_xs(D, Xs_dev, ldxs, &D->xs_seen);
    if (rc) return mark_failed(D, rc);
  }
  return G3_OK;
}

// ---------------------------------------------------------------------------------------- posterior covariance + draws
// BASELINE config 5's extra work: K(Xs, Xs) - V V^T (elliptical.py:86-91), its robust Cholesky (elliptical.py:88,92;
// tensors.py:197-222) and loc + L_post Z (gaussian.py:75-97, before the mapping).  V = K(Xs, X) L^-T sits in the
// right-hand-side chunks, full rows per chunk.  All-gather V (M x N: 1 GiB at config 5), every rank forms the LOWER part
// of the covariance rows of ITS chunks with one staircase launch against the gathered V (rank-major: block table) -- the
// prior part from g3_gram_rows, i.e. with the square-case semantics of NOISE / WN leaves (kernels.py:360-385) -- the
// M x M covariance is all-gathered (64 MiB), mirrored, and (draws) factored redundantly on every rank -- M^3/3 flops,
// nothing to exchange -- so all ranks hold the same matrix and the same draws.
template <typename T>
__global__ void __launch_bounds__(256) mirror_lower_kernel(T* __restrict__ A, int64_t n, int64_t ld) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j < n && j > i) A[i * ld + j] = A[j * ld + i];
}

static int posterior_cov(g3_dist* D, const g3_kernel_prog* prog, const void* Xs_dev, int64_t ldxs, char* cov, int64_t ldc) {
  const int P = D->world;
  const int64_t M = D->M, Mp = D->Mp, Np = D->Np, pad = 128;
  const int nch = (int)(Mp / pad);
  std::vector<int> own(nch), cntq(P, 0), mine;
  for (int c = 0; c < nch; ++c) { own[c] = (c + 1) % P; cntq[own[c]]++; if (own[c] == D->rank) mine.push_back(c); }
  int cmax = 1;
  for (int q = 0; q < P; ++q) cmax = cntq[q] > cmax ? cntq[q] : cmax;
  hipStream_t s = D->ctx->stream;
  char *sendb = nullptr, *Vall = nullptr, *rows = nullptr, *call = nullptr;
  int rc = G3_OK;
  auto cleanup = [&]() {
    void* b[] = {sendb, Vall, rows, call};
    (void)hipStreamSynchronize(s);
    for (void* p : b) if (p) (void)hipFree(p);
  };
#define G3D_TRY(x) do { rc = (x); if (rc) { cleanup(); return rc; } } while (0)
#define G3D_TRYH(x) do { if ((x) != hipSuccess) { snprintf(D->err, sizeof(D->err), "%s:%d %s", __FILE__, __LINE__, #x); cleanup(); return G3_ERR_HIP; } } while (0)
  const size_t es = D->es;
  G3D_TRYH(hipMalloc((void**)&sendb, (size_t)cmax * pad * Np * es));
  G3D_TRYH(hipMalloc((void**)&Vall, (size_t)P * cmax * pad * Np * es));
  G3D_TRYH(hipMalloc((void**)&rows, (size_t)cmax * pad * Mp * es));
  G3D_TRYH(hipMalloc((void**)&call, (size_t)P * cmax * pad * Mp * es));
  G3D_TRYH(hipMemsetAsync(sendb, 0, (size_t)cmax * pad * Np * es, s));
  G3D_TRYH(hipMemsetAsync(rows, 0, (size_t)cmax * pad * Mp * es, s));
  // my V chunks, in chunk order (chunk c of Xs = right-hand-side chunk c + 1)
  for (size_t i = 0; i < mine.size(); ++i) {
    size_t t = 0;
    while (t < D->my_chunks.size() && D->my_chunks[t] != mine[i] + 1) ++t;
    if (t == D->my_chunks.size()) {     // the two dealings (plan: c % P; here: (c + 1) % P of the Xs chunk) must agree
      snprintf(D->err, sizeof(D->err), "posterior covariance: right-hand-side chunk %d is not on rank %d", mine[i] + 1, D->rank);
      cleanup();
      return G3_ERR_HIP;
    }
    G3D_TRYH(hipMemcpyAsync(sendb + i * pad * Np * es, Aat(D, D->rows_mat + (int64_t)t * pad, 0), (size_t)pad * Np * es,
                            hipMemcpyDeviceToDevice, s));
  }
  G3D_TRY(do_allgather(D, sendb, Vall, (size_t)cmax * pad * Np * es, s));
  std::vector<int32_t> perm(nch);
  {
    std::vector<int> seen(P, 0);
    for (int c = 0; c < nch; ++c) perm[c] = own[c] * cmax + seen[own[c]]++;
  }
  // lower part of the covariance rows of my chunks: K(Xs)[rows of c, columns <= the chunk's last row] - V_c Vall^T
  // (prior part: the plain kernel.cov(Xs), not scrubbed)
  std::vector<int64_t> sr, sc, sd;
  for (size_t i = 0; i < mine.size(); ++i) {
    const int64_t c = mine[i];
    const int64_t m = M - c * pad < pad ? M - c * pad : pad;
    if (m > 0)
      G3D_TRY(g3_gram_rows(D->ctx, prog, Xs_dev, M, ldxs, D->d, c * pad, m, D->dt, rows + i * pad * Mp * es, Mp, 0));
    sr.push_back(pad);
    sc.push_back((c + 1) * pad);
    sd.push_back(1);
  }
  if (!mine.empty()) {
    std::vector<G3hStairChunk> ch;
    g3h_stair_chunks(sr, sc, pad, nch, &ch, &sd, D->ctx->tune.stair_max);
    for (const auto& cc : ch)
      G3D_TRY(g3_gemm_nt_stair(D->ctx, rows + ((size_t)cc.row0 * Mp + cc.col0) * es, Mp, sendb + (size_t)cc.row0 * Np * es, Np, Vall, Np, Np,
                               cc.rows.data(), cc.cols.data(), (int)cc.rows.size(), -1.0, 1.0, D->dt, pad, perm.data() + cc.blk0, cc.nblk,
                               cc.diag.data()));
  }
  G3D_TRY(do_allgather(D, rows, call, (size_t)cmax * pad * Mp * es, s));
  for (int c = 0; c < nch; ++c)
    G3D_TRYH(hipMemcpy2DAsync(cov + (size_t)c * pad * ldc * es, (size_t)ldc * es, call + (size_t)perm[c] * pad * Mp * es, (size_t)Mp * es,
                              (size_t)Mp * es, pad, hipMemcpyDeviceToDevice, s));
  {
    const dim3 grid((unsigned)((Mp + 255) / 256), (unsigned)Mp);
    if (D->dt == G3_F64) hipLaunchKernelGGL(mirror_lower_kernel<double>, grid, dim3(256), 0, s, (double*)cov, Mp, ldc);
    else hipLaunchKernelGGL(mirror_lower_kernel<float>, grid, dim3(256), 0, s, (float*)cov, Mp, ldc);
    G3D_TRYH(hipGetLastError());
  }
  cleanup();
#undef G3D_TRY
#undef G3D_TRYH
  return G3_OK;
}

extern "C" int g3_dist_posterior_cov(g3_dist* D, const g3_kernel_prog* prog, const void* Xs_dev, int64_t ldxs, void* cov_dev, int64_t ldc) {
  if (!D) return -1;
  if (!D->planned || D->M <= 0) return -1;
  if (!prog || g3i_validate_prog(prog, D->d)) return -2;
  if (!Xs_dev) return -3;
  if (ldxs < D->d) return -4;
  if (!cov_dev) return -5;
  if (ldc < D->Mp) return -6;
  g3_dev_guard _dg(D->ctx);
  ChainScope _cs(D);
  int rcx = same_xs_as_evaluated(D, Xs_dev, ldxs);
  if (rcx) return rcx;
  return posterior_cov(D, prog, Xs_dev, ldxs, (char*)cov_dev, ldc);
}

extern "C" int g3_dist_posterior_draws(g3_dist* D, const g3_kernel_prog* prog_f, const void* Xs_dev, int64_t ldxs,
                                       const void* loc_host, const void* Z_host, int64_t S, void* out_host, int* tries_host,
                                       int* fallback_host) {
  if (!D) return -1;
  if (!D->planned || D->M <= 0) return -1;
  if (!prog_f) return -2;
  if (!Xs_dev) return -3;
  if (ldxs < D->d) return -4;
  if (!loc_host) return -5;
  if (!Z_host) return -6;
  if (S <= 0) return -7;
  if (!out_host) return -8;
  if (g3i_validate_prog(prog_f, D->d)) return -2;
  g3_dev_guard _dg(D->ctx);
  ChainScope _cs(D);
  {
    const int rcx = same_xs_as_evaluated(D, Xs_dev, ldxs);
    if (rcx) return rcx;
  }
  const int64_t M = D->M, Mp = D->Mp;
  const size_t es = D->es;
  hipStream_t s = D->ctx->stream;
  char *cov = nullptr, *Lp = nullptr;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(s);
    if (cov) (void)hipFree(cov);
    if (Lp) (void)hipFree(Lp);
  };
  if (hipMalloc((void**)&cov, (size_t)Mp * Mp * es) != hipSuccess || hipMalloc((void**)&Lp, (size_t)Mp * Mp * es) != hipSuccess ||
      hipMemsetAsync(Lp, 0, (size_t)Mp * Mp * es, s) != hipSuccess) {
    snprintf(D->err, sizeof(D->err), "posterior_draws: out of device memory for two %lld x %lld matrices", (long long)Mp, (long long)Mp);
    cleanup();
    return G3_ERR_NOMEM;
  }
  int rc = posterior_cov(D, prog_f, Xs_dev, ldxs, cov, Mp);
  int tries = 0, fb = 0;
  double jit = 0;
  if (!rc) {
    rc = g3_potrf_robust(D->ctx, cov, Mp, Lp, Mp, M, D->dt, 20, &tries, &fb, &jit);
    if (rc) snprintf(D->err, sizeof(D->err), "posterior_draws: g3_potrf_robust -> %d %s", rc, D->ctx->err);
  }
  if (!rc) {
    rc = g3_gp_sample(D->ctx, Lp, M, Mp, loc_host, Z_host, S, D->dt, out_host);
    if (rc) snprintf(D->err, sizeof(D->err), "posterior_draws: g3_gp_sample -> %d %s", rc, D->ctx->err);
  }
  if (tries_host) *tries_host = tries;
  if (fallback_host) *fallback_host = fb;
  cleanup();
  return rc;
}

// ---------------------------------------------------------------------------------------- gradient of logp
// dlogp / dtheta = 1/2 sum_ij (alpha_i alpha_j - K^-1_ij) dK_ij / dtheta,  alpha = K^-1 delta  (the reference differentiates
// through CholeskyRobust.grad, tensors.py:224-260; stochastic.py:308-309).  In gradient mode the factorisation carries the
// identity as extra right-hand-side rows, so every rank ends up with ITS rows of X = L^-T (upper triangular: row block I
// starts at column block I).  K^-1 = X X^T: for every column block k the ranks all-gather the panel X[:, k] (rows of the
// blocks <= k) and add X[I, k] X[J, k]^T to the rows I <= k they own, columns J <= I -- the staircase launch of the
// factorisation run backwards, N^3 / 3 flops over all ranks, no dependency between the steps (gather k + 1 overlaps the
// product k).  alpha = X a; then one pass of the gradient kernel over the rank's rows of K^-1, and an all-reduce of the
// parameter sums.
template <typename T>
__global__ void __launch_bounds__(256) rows_dot_kernel(const T* __restrict__ A, int64_t rows, int64_t cols, int64_t ld,
                                                      const T* __restrict__ v, T* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (int64_t j = lane; j < cols; j += 64) acc = fma((double)A[r * ld + j], (double)v[j], acc);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) out[r] = (T)acc;
}

static int perm_upto(const g3_dist* D, int k, std::vector<int32_t>* idx) {   // blocks 0 .. k in the rank-major gather
  const int c = g3h_gather_table(D->owner, D->world, 0, k, idx);
  return c < 1 ? 1 : c;
}

extern "C" int g3_dist_set_grad(g3_dist* D, int on) {
  if (!D) return -1;
  const bool want = on != 0;
  if (D->grad == want) return G3_OK;
  D->grad = want;
  D->have_inv = false;
  if (!D->planned) return G3_OK;
  return g3_dist_plan(D, D->N, D->d, D->M, D->nb, D->dt);     // the local matrix changes size
}

extern "C" int g3_dist_gp_dlogp(g3_dist* D, const g3_kernel_prog* prog, const g3_grad_map* map, const void* X_dev, int64_t ldx,
                                double alpha_scale, double* slots_host, double* alpha_host) {
  if (!D) return -1;
  if (!D->planned || !D->grad || !D->have_inv) {
    if (D) snprintf(D->err, sizeof(D->err), "g3_dist_gp_dlogp: g3_dist_set_grad(1) and g3_dist_gp_factor_predict first");
    return -1;
  }
  if (!prog || g3i_validate_prog(prog, D->d)) return -2;
  if (!map || map->nslots < 0 || map->nslots > G3_GRAD_MAXSLOTS) return -3;
  if (!X_dev) return -4;
  if (ldx < D->d) return -5;
  if (!slots_host) return -7;
  g3_dev_guard _dg(D->ctx);
  ChainScope _cs(D);
  const int64_t nb = D->nb, Np = D->Np;
  const size_t es = D->es;
  const int nblk = D->nblk;
  hipStream_t sA = D->ctx->stream, sB = D->s_bulk;
  hipEvent_t ev_tmp = D->ev[nblk], ev_join = D->ev[nblk + 1];
  const char* Xinv = Aat(D, D->rows_mat + D->rows_rhs, 0);
  // ---- alpha = X a (a = L^-1 delta: the broadcast copy stats() left in avec), gathered and put in global order
  std::vector<int32_t> perm;
  const int call = perm_upto(D, nblk - 1, &perm);
  const int64_t apr = (int64_t)call * nb;               // alpha entries per rank in the padded gather
  char* asend = D->send[0];                             // cmax >= cmax_all blocks of nb x nb: room for cmax_all * nb values
  G3D_HIP(hipMemsetAsync(asend, 0, (size_t)apr * es, sA));
  if (D->rows_inv > 0) {
    const unsigned grid = (unsigned)((D->rows_inv + 3) / 4);
    if (D->dt == G3_F64)
      hipLaunchKernelGGL(rows_dot_kernel<double>, dim3(grid), dim3(256), 0, sA, (const double*)Xinv, D->rows_inv, Np, Np,
                         (const double*)D->avec, (double*)asend);
    else
      hipLaunchKernelGGL(rows_dot_kernel<float>, dim3(grid), dim3(256), 0, sA, (const float*)Xinv, D->rows_inv, Np, Np,
                         (const float*)D->avec, (float*)asend);
    G3D_HIP(hipGetLastError());
  }
  int rc = do_allgather(D, asend, D->agath, (size_t)apr * es, sA);
  if (rc) return rc;
  std::vector<char> hg((size_t)D->world * apr * es), ha((size_t)Np * es);
  G3D_HIP(hipMemcpyAsync(hg.data(), D->agath, hg.size(), hipMemcpyDeviceToHost, sA));
  G3D_HIP(hipStreamSynchronize(sA));
  for (int I = 0; I < nblk; ++I)
    for (int64_t r = 0; r < nb; ++r) {
      const size_t src = (size_t)perm[I] * nb + r, dst = (size_t)I * nb + r;
      if (D->dt == G3_F64) {
        const double v = ((const double*)hg.data())[src] * alpha_scale;
        ((double*)ha.data())[dst] = v;
        if (alpha_host && (int64_t)dst < D->N) alpha_host[dst] = v;
      } else {
        const float v = (float)(((const float*)hg.data())[src] * alpha_scale);
        ((float*)ha.data())[dst] = v;
        if (alpha_host && (int64_t)dst < D->N) alpha_host[dst] = (double)v;
      }
    }
  G3D_HIP(hipMemcpyAsync(D->alpha_dev, ha.data(), ha.size(), hipMemcpyHostToDevice, sA));
  // ---- my rows of K^-1 = X X^T, lower part
  if (D->rows_mat > 0) G3D_HIP(hipMemsetAsync(D->Kinv, 0, (size_t)D->rows_mat * Np * es, sA));
  std::vector<int64_t> seg_rows, seg_cols, seg_diag;
  for (int k = 0; k < nblk; ++k) {
    const int cnt = perm_upto(D, k, &perm);
    const int64_t act = rows_done(D, k);
    if (k >= 3) G3D_HIP(hipStreamWaitEvent(sA, D->ev[k - 3], 0));     // the product that read gath[k % 3] is done
    char* slot = D->gath[k % 3] + (size_t)D->rank * cnt * nb * nb * es;      // in place, as in solve_and_gather
    if (act > 0) G3D_RC(D->ctx, g3_copy2d(D->ctx, slot, nb, Xinv + (size_t)k * nb * es, Np, act, nb, D->dt));
    rc = do_allgather(D, slot, D->gath[k % 3], (size_t)cnt * nb * nb * es, sA);
    if (rc) return rc;
    rc = stream_after(D, sB, sA, ev_tmp);
    if (rc) return rc;
    seg_rows.clear(); seg_cols.clear(); seg_diag.clear();
    for (int I : D->my_blocks) if (I <= k) { seg_rows.push_back(nb); seg_cols.push_back((int64_t)(I + 1) * nb); seg_diag.push_back(1); }
    if (!seg_rows.empty()) {
      rc = stair_ptr(D, D->ctx_bulk, D->Kinv, Xinv + (size_t)k * nb * es, Np, D->gath[k % 3], seg_rows, seg_cols, perm.data(), k + 1,
                     &seg_diag, 1.0);
      if (rc) return rc;
    }
    G3D_HIP(hipEventRecord(D->ev[k], sB));
  }
  rc = stream_after(D, sA, sB, ev_join);
  if (rc) return rc;
  // ---- the parameter sums over my rows, then over the ranks
  const int ns = map->nslots;
  std::vector<double> acc(ns > 0 ? ns : 1, 0.0), part(ns > 0 ? ns : 1, 0.0);
  for (int I : D->my_blocks) {
    const int64_t r0 = (int64_t)I * nb;
    const int64_t nv = D->N - r0 < nb ? D->N - r0 : nb;
    if (nv <= 0 || ns == 0) continue;
    G3D_RC(D->ctx, g3i_gram_grad(D->ctx, prog, map, X_dev, D->N, ldx, D->d, D->dt, D->Kinv + (size_t)D->loff[I] * Np * es, Np,
                                 D->alpha_dev, part.data(), r0, nv));
    for (int q = 0; q < ns; ++q) acc[q] += part[q];
  }
  G3D_HIP(hipStreamSynchronize(sA));
  if (ns > 0) {
    rc = do_allreduce(D, acc.data(), ns, 0);
    if (rc) return rc;
  }
  for (int q = 0; q < ns; ++q) slots_host[q] = acc[q];
  return G3_OK;
}

// per kind (broadcast, all-gather, all-reduce): calls, bytes sent + received by this rank, device milliseconds of
// the collective calls themselves (HIP events on the stream each was enqueued on); counters reset
extern "C" int g3_dist_comm_stats(g3_dist* D, double out_host[9]) {
  if (!D) return -1;
  if (!out_host) return -2;
  g3_dev_guard _dg(D->ctx);
  G3D_HIP(hipStreamSynchronize(D->ctx->stream));
  if (D->s_chain) G3D_HIP(hipStreamSynchronize(D->s_chain));
  G3D_HIP(hipStreamSynchronize(D->s_look));
  G3D_HIP(hipStreamSynchronize(D->s_bulk));
  double ms[G3_NKIND] = {0, 0, 0, 0, 0};
  for (size_t i = 0; i + 1 < D->tused; i += 2) {
    float t = 0;
    if (hipEventElapsedTime(&t, D->tev[i], D->tev[i + 1]) == hipSuccess) ms[D->tkind[i / 2]] += t;
  }
  for (int k = 0; k < G3_NCOLL; ++k) {
    out_host[3 * k] = D->n_calls[k];
    out_host[3 * k + 1] = D->n_bytes[k];
    out_host[3 * k + 2] = ms[k];
  }
  // the two timed phases of the chain, read by g3_dist_phase_stats (same reset point)
  for (int k = 0; k < 2; ++k) {
    D->phase_calls[k] = D->n_calls[G3_PH_DIAG + k];
    D->phase_ms[k] = ms[G3_PH_DIAG + k];
  }
  for (int k = 0; k < G3_NKIND; ++k) D->n_calls[k] = D->n_bytes[k] = 0;
  D->tused = 0;
  return G3_OK;
}

// HIP-event profiling of the bulk stream's MFMA GEMM launches (the staircase updates: where the flops of a rank
// are), same tags and output layout as g3_prof_collect
extern "C" int g3_dist_phase_stats(g3_dist* D, double out_host[4]) {
  if (!D) return -1;
  if (!out_host) return -2;
  out_host[0] = D->phase_calls[0];   // diagonal blocks this rank updated + factored on its look-ahead stream
  out_host[1] = D->phase_ms[0];      // ... and the device time of that (gemm + factorisation + copies), summed
  out_host[2] = D->phase_calls[1];   // panel solves of this rank's rows
  out_host[3] = D->phase_ms[1];
  return G3_OK;
}

extern "C" int g3_dist_prof_enable(g3_dist* D, int on) {
  if (!D) return -1;
  int rc = G3_OK;
  for (int i = 0; i < 2 && !rc; ++i)
    if (D->ctx_bulk_alt[i]) {
      rc = g3_prof_enable(D->ctx_bulk_alt[i], on);
      if (!rc) rc = g3_prof_reset(D->ctx_bulk_alt[i]);
    }
  return rc;
}
extern "C" int g3_dist_prof_collect(g3_dist* D, double* out_host) {
  if (!D) return -1;
  if (!out_host) return -2;
  return g3_prof_collect(D->ctx_bulk, out_host);
}

// rows x cols view of the rank's local matrix (tests / debugging): copies A[row0 : row0 + rows, 0 : cols) to the host
extern "C" int g3_dist_local_rows(g3_dist* D, int64_t* rows_mat, int64_t* rows_rhs, int64_t* ld) {
  if (!D || !D->planned) return -1;
  if (rows_mat) *rows_mat = D->rows_mat;
  if (rows_rhs) *rows_rhs = D->rows_rhs;
  if (ld) *ld = D->Np;
  return G3_OK;
}
