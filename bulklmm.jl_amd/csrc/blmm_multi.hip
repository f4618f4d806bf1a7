// blmm_multi.hip -- the multi-GPU form of the bulkscan entry point behind the C ABI (include/bulklmm_hip.h).
//
// Traits shard (every column of Y is independent given (Xt, lambda); the reference blocks the same way over threads,
// src/bulkscan.jl:263-309): device r of R scans the contiguous column block [r*ceil(m/R), min(m, (r+1)*ceil(m/R))) of Y
// and owns that block of the column-major p x m LOD matrix.  G, K, the covariates and the weights are replicated and
// the n x n eigen problem is solved redundantly on every device (it is latency bound on a few CUs; broadcasting U
// would cost a synchronisation of all devices in the middle of the call).  One host worker thread per device drives
// its own blmm_ctx (contexts are not thread-safe, threads never share one); the data path has NO collective.
//
// gather_mode decides where the blocks end up:
//   host_shards  every device copies its block straight into the caller's L_out -- R PCIe links in parallel (the
//                drop-in default: the reference returns L in host memory);
//   none         the blocks stay in HBM (blmm_multi_device_result gives the pointers);
//   allgather    RCCL ncclAllGather over xGMI assembles the full matrix on EVERY device (north_star's optional final
//                step, for device-side consumers); librccl.so is loaded with dlopen on first use.  With duplicated
//                device ids (the 2-shards-on-one-GPU rehearsal of the tests) or BLMM_ALLGATHER=peer the same exchange
//                runs as R-1 direct hipMemcpyPeerAsync copies per device instead.
#include "blmm_internal.h"
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <mutex>
#include <set>
#include <thread>

namespace {

struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false;
  int rc = 0;
};

}  // namespace

struct blmm_multi {
  int ndev = 0;
  std::vector<int> dev;
  std::vector<blmm_ctx*> ctx;
  std::vector<Worker*> wk;
  std::string err;
  // device-resident results of the last call (gather none / allgather)
  std::vector<blmm::DevBuf> dY, dG, dK, dCov, dW, dL, dH;
  int64_t last_m = 0, last_p = 0, last_block = 0;
  int last_gather = -1, last_method = 0;
  // RCCL (dlopen)
  void* nccl_lib = nullptr;
  bool nccl_tried = false;
  std::vector<void*> comms;
  int (*ncclCommInitAll)(void**, int, const int*) = nullptr;
  int (*ncclCommDestroy)(void*) = nullptr;
  int (*ncclGroupStart)() = nullptr;
  int (*ncclGroupEnd)() = nullptr;
  int (*ncclAllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*ncclGetErrorString)(int) = nullptr;
};

namespace {

// The multi-device entry points visit every device from the CALLER's thread (context creation, the all-gather loops, the
// copy-out); the caller's current HIP device is put back on every exit path, so a torch caller keeps allocating and
// launching where it was.
struct DeviceRestore {
  int d = -1;
  DeviceRestore() { if (hipGetDevice(&d) != hipSuccess) d = -1; }
  ~DeviceRestore() { if (d >= 0) (void)hipSetDevice(d); }
};

int mfail(blmm_multi* mc, int code, const std::string& msg) {
  if (mc) mc->err = msg;
  return code;
}

void worker_main(Worker* w) {
  for (;;) {
    std::function<int()> job;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->has_job || w->quit; });
      if (w->quit) return;
      job = w->job;
    }
    const int rc = job();
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->rc = rc; w->has_job = false; w->done = true;
    }
    w->cv.notify_all();
  }
}

void post(Worker* w, std::function<int()> f) {
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->job = std::move(f); w->has_job = true; w->done = false;
  }
  w->cv.notify_all();
}

int wait(Worker* w) {
  std::unique_lock<std::mutex> lk(w->mu);
  w->cv.wait(lk, [&] { return w->done; });
  return w->rc;
}

// Runs f(r) on every device's worker thread and returns the first failure (every worker is always joined).
int on_all(blmm_multi* mc, const std::function<int(int)>& f) {
  for (int r = 0; r < mc->ndev; ++r) post(mc->wk[r], [&f, r] { return f(r); });
  int first = BLMM_OK, who = -1;
  for (int r = 0; r < mc->ndev; ++r) {
    const int rc = wait(mc->wk[r]);
    if (rc != BLMM_OK && first == BLMM_OK) { first = rc; who = r; }
  }
  if (first != BLMM_OK) {
    const char* m = blmm_last_error(mc->ctx[who]);
    mc->err = "device " + std::to_string(mc->dev[who]) + ": " + ((m && *m) ? m : blmm_err_string(first));
  }
  return first;
}

bool distinct_devices(const blmm_multi* mc) {
  std::set<int> s(mc->dev.begin(), mc->dev.end());
  return (int)s.size() == mc->ndev;
}

int load_rccl(blmm_multi* mc) {
  if (mc->nccl_tried) return mc->ncclAllGather ? BLMM_OK : BLMM_ERR_UNSUPPORTED;
  mc->nccl_tried = true;
  mc->nccl_lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!mc->nccl_lib) mc->nccl_lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!mc->nccl_lib) mc->nccl_lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!mc->nccl_lib) return mfail(mc, BLMM_ERR_UNSUPPORTED, "gather_mode allgather needs librccl.so (dlopen failed)");
  mc->ncclCommInitAll = reinterpret_cast<int (*)(void**, int, const int*)>(dlsym(mc->nccl_lib, "ncclCommInitAll"));
  mc->ncclCommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(mc->nccl_lib, "ncclCommDestroy"));
  mc->ncclGroupStart = reinterpret_cast<int (*)()>(dlsym(mc->nccl_lib, "ncclGroupStart"));
  mc->ncclGroupEnd = reinterpret_cast<int (*)()>(dlsym(mc->nccl_lib, "ncclGroupEnd"));
  mc->ncclGetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(mc->nccl_lib, "ncclGetErrorString"));
  auto ag = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(mc->nccl_lib, "ncclAllGather"));
  if (!mc->ncclCommInitAll || !mc->ncclCommDestroy || !mc->ncclGroupStart || !mc->ncclGroupEnd || !ag)
    return mfail(mc, BLMM_ERR_UNSUPPORTED, "librccl.so lacks an expected symbol");
  mc->comms.assign(mc->ndev, nullptr);
  const int st = mc->ncclCommInitAll(mc->comms.data(), mc->ndev, mc->dev.data());
  if (st != 0) {
    mc->comms.clear();
    return mfail(mc, BLMM_ERR_HIP, std::string("ncclCommInitAll: ") + (mc->ncclGetErrorString ? mc->ncclGetErrorString(st) : "failed"));
  }
  mc->ncclAllGather = ag;
  return BLMM_OK;
}

}  // namespace

extern "C" {

void blmm_multi_shard(int64_t m, int rank, int ndev, int64_t* lo, int64_t* hi) {
  const int64_t blk = ndev > 0 ? (m + ndev - 1) / ndev : m;
  int64_t a = (int64_t)rank * blk, b = a + blk;
  if (a > m) a = m;
  if (b > m) b = m;
  if (lo) *lo = a;
  if (hi) *hi = b;
}

int blmm_create_multi(const int* device_ids, int ndev, blmm_multi** out) {
  if (!out) return BLMM_ERR_INVALID;
  *out = nullptr;
  const int avail = blmm_device_count();
  if (avail <= 0) return BLMM_ERR_NO_DEVICE;
  if (ndev <= 0) ndev = avail;            // ndev <= 0: every visible device
  if (ndev > 64) return BLMM_ERR_INVALID;
  DeviceRestore keep_device;
  blmm_multi* mc = new blmm_multi();
  mc->ndev = ndev;
  mc->dY.resize(ndev); mc->dG.resize(ndev); mc->dK.resize(ndev); mc->dCov.resize(ndev); mc->dW.resize(ndev);
  mc->dL.resize(ndev); mc->dH.resize(ndev);
  for (int r = 0; r < ndev; ++r) {
    const int d = device_ids ? device_ids[r] : r;
    blmm_ctx* c = nullptr;
    const int rc = blmm_create(d, nullptr, &c);   // a private stream per device
    if (rc != BLMM_OK) { blmm_destroy_multi(mc); return rc; }
    mc->dev.push_back(d);
    mc->ctx.push_back(c);
  }
  for (int r = 0; r < ndev; ++r) {
    Worker* w = new Worker();
    w->th = std::thread(worker_main, w);
    mc->wk.push_back(w);
  }
  *out = mc;
  return BLMM_OK;
}

void blmm_destroy_multi(blmm_multi* mc) {
  if (!mc) return;
  DeviceRestore keep_device;
  for (Worker* w : mc->wk) {
    { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  if (mc->ncclCommDestroy) for (void* c : mc->comms) if (c) mc->ncclCommDestroy(c);
  for (int r = 0; r < (int)mc->ctx.size(); ++r) {
    if (!mc->ctx[r]) continue;
    (void)hipSetDevice(mc->dev[r]);
    (void)hipStreamSynchronize(mc->ctx[r]->stream);
    for (blmm::DevBuf* b : {&mc->dY[r], &mc->dG[r], &mc->dK[r], &mc->dCov[r], &mc->dW[r], &mc->dL[r], &mc->dH[r]})
      if (b->p) (void)hipFree(b->p);
    blmm_destroy(mc->ctx[r]);
  }
  delete mc;
}

int blmm_multi_ndev(const blmm_multi* mc) { return mc ? mc->ndev : 0; }
const char* blmm_multi_last_error(const blmm_multi* mc) { return mc ? mc->err.c_str() : "multi context is NULL"; }

void blmm_default_multi_opts(blmm_multi_opts* o) {
  if (!o) return;
  o->gather_mode = BLMM_GATHER_HOST_SHARDS;
  o->reserved = 0;
}

int blmm_bulkscan_multi(blmm_multi* mc, const blmm_opts* opts, const blmm_multi_opts* mopts, const double* Y, int64_t n,
                        int64_t m, const double* G, int64_t p, const double* Covar, int64_t ncov, const double* K,
                        const double* weights, const double* h2_grid, int64_t ngrid, double* L_out, double* h2_out,
                        blmm_status* status) {
  using namespace blmm;
  if (!mc) return BLMM_ERR_INVALID;
  if (!opts) return mfail(mc, BLMM_ERR_INVALID, "opts is NULL");
  const int gather = mopts ? mopts->gather_mode : BLMM_GATHER_HOST_SHARDS;
  if (gather != BLMM_GATHER_NONE && gather != BLMM_GATHER_HOST_SHARDS && gather != BLMM_GATHER_ALLGATHER)
    return mfail(mc, BLMM_ERR_INVALID, "unknown gather_mode");
  if (!Y || !G || !K) return mfail(mc, BLMM_ERR_INVALID, "bulkscan: NULL buffer");
  // host_shards with L_out == NULL: every device keeps its block in its context's workspace (blmm_bulkscan with L_out == NULL) and
  // blmm_multi_last_colmax / blmm_multi_last_lod_threshold reduce the blocks where they are
  if (gather == BLMM_GATHER_HOST_SHARDS && !h2_out && opts->method != BLMM_ALT_GRID) return mfail(mc, BLMM_ERR_INVALID, "bulkscan: NULL output buffer");
  if (n < 1 || m < 0 || p < 0) return mfail(mc, BLMM_ERR_DIM, "Dimension mismatch.");
  DeviceRestore keep_device;
  const int R = mc->ndev;
  const bool alt = opts->method == BLMM_ALT_GRID;
  const int64_t blk = (m + R - 1) / R;          // columns per device; the last shard may be shorter (or empty)
  if (status) std::memset(status, 0, sizeof(blmm_status) * R);

  if (gather == BLMM_GATHER_HOST_SHARDS) {
    // every worker runs the host-pointer entry point on its column block of the caller's matrices
    const int rc = on_all(mc, [&](int r) -> int {
      int64_t lo, hi;
      blmm_multi_shard(m, r, R, &lo, &hi);
      return blmm_bulkscan(mc->ctx[r], opts, Y + (size_t)lo * n, n, hi - lo, G, p, Covar, ncov, K, weights, h2_grid, ngrid,
                           L_out ? L_out + (size_t)lo * p : nullptr, !h2_out ? nullptr : (alt ? h2_out + (size_t)lo * p : h2_out + lo),
                           status ? status + r : nullptr);
    });
    mc->last_gather = gather; mc->last_m = m; mc->last_p = p; mc->last_block = blk; mc->last_method = opts->method;
    if (getenv("BLMM_MULTI_LOG")) fprintf(stderr, "blmm_bulkscan_multi: %d devices, gather host_shards%s, rc %d\n", R, L_out ? "" : " (L stays in HBM)", rc);
    return rc;
  }

  // device-resident results: every device gets a buffer for the WHOLE (padded) matrix when it is to be gathered,
  // for its own block otherwise; its scan writes at its block's place
  const bool full = gather == BLMM_GATHER_ALLGATHER;
  const int64_t cols_alloc = full ? blk * R : blk;
  int rc = on_all(mc, [&](int r) -> int {
    blmm_ctx* ctx = mc->ctx[r];
    BLMM_HIP(hipSetDevice(ctx->device));
    int64_t lo, hi;
    blmm_multi_shard(m, r, R, &lo, &hi);
    const int64_t mr = hi - lo;
    int e;
    if ((e = ensure(ctx, mc->dY[r], sizeof(double) * n * (mr > 0 ? mr : 1)))) return e;
    if ((e = ensure(ctx, mc->dG[r], sizeof(double) * n * (p > 0 ? p : 1)))) return e;
    if ((e = ensure(ctx, mc->dK[r], sizeof(double) * n * n))) return e;
    if ((e = ensure(ctx, mc->dL[r], sizeof(double) * (size_t)(p > 0 ? p : 1) * (cols_alloc > 0 ? cols_alloc : 1)))) return e;
    if ((e = ensure(ctx, mc->dH[r], sizeof(double) * (alt ? (size_t)(p > 0 ? p : 1) : 1) * (cols_alloc > 0 ? cols_alloc : 1)))) return e;
    if (mr > 0) BLMM_HIP(hipMemcpyAsync(mc->dY[r].p, Y + (size_t)lo * n, sizeof(double) * n * mr, hipMemcpyHostToDevice, ctx->stream));
    if (p > 0) BLMM_HIP(hipMemcpyAsync(mc->dG[r].p, G, sizeof(double) * n * p, hipMemcpyHostToDevice, ctx->stream));
    BLMM_HIP(hipMemcpyAsync(mc->dK[r].p, K, sizeof(double) * n * n, hipMemcpyHostToDevice, ctx->stream));
    const double* dCov = nullptr; const double* dW = nullptr;
    if (Covar && ncov > 0) {
      if ((e = ensure(ctx, mc->dCov[r], sizeof(double) * n * ncov))) return e;
      BLMM_HIP(hipMemcpyAsync(mc->dCov[r].p, Covar, sizeof(double) * n * ncov, hipMemcpyHostToDevice, ctx->stream));
      dCov = ptr<double>(mc->dCov[r]);
    }
    if (weights) {
      if ((e = ensure(ctx, mc->dW[r], sizeof(double) * n))) return e;
      BLMM_HIP(hipMemcpyAsync(mc->dW[r].p, weights, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
      dW = ptr<double>(mc->dW[r]);
    }
    const int64_t off = full ? lo : 0;
    double* dL = ptr<double>(mc->dL[r]) + (size_t)off * p;
    double* dH = ptr<double>(mc->dH[r]) + (alt ? (size_t)off * p : (size_t)off);
    e = blmm_bulkscan_dev(ctx, opts, ptr<double>(mc->dY[r]), n, mr, ptr<double>(mc->dG[r]), p, dCov, dCov ? ncov : 0,
                          ptr<double>(mc->dK[r]), dW, h2_grid, ngrid, dL, p > 0 ? p : 1, dH, status ? status + r : nullptr);
    if (e) { (void)hipStreamSynchronize(ctx->stream); return e; }
    // the host inputs may be released; the blocks are complete; a device-side failure of this call made without a status
    // (grid-barrier timeout, eigensolver abort) is reported by this call
    return blmm_synchronize(ctx);
  });
  if (rc) return rc;
  mc->last_m = m; mc->last_p = p; mc->last_block = blk; mc->last_gather = gather; mc->last_method = opts->method;

  const char* env = dev_env("BLMM_ALLGATHER");   // "peer": direct copies; "rccl": RCCL even for a single device (tests)
  const bool force_rccl = env && std::strcmp(env, "rccl") == 0;
  if (full && (R > 1 || force_rccl) && blk > 0 && p > 0) {
    const size_t cntL = (size_t)blk * p, cntH = alt ? (size_t)blk * p : (size_t)blk;
    const bool want_peer = (env && std::strcmp(env, "peer") == 0) || !distinct_devices(mc);
    if (getenv("BLMM_MULTI_LOG")) fprintf(stderr, "blmm_bulkscan_multi: %d devices, all-gather through %s (%zu doubles per block)\n", R, want_peer ? "hipMemcpyPeerAsync" : "RCCL ncclAllGather", cntL);
    if (!want_peer) {
      if ((rc = load_rccl(mc))) return rc;   // no silent fallback: a missing / failing RCCL is the caller's to know
      // in-place all-gather: device r's block already sits at slot r of its own full-size buffer
      int st = mc->ncclGroupStart();
      for (int r = 0; r < R && st == 0; ++r) {
        (void)hipSetDevice(mc->dev[r]);
        double* Lr = blmm::ptr<double>(mc->dL[r]); double* Hr = blmm::ptr<double>(mc->dH[r]);
        st = mc->ncclAllGather(Lr + (size_t)r * cntL, Lr, cntL, /*ncclDouble*/ 8, mc->comms[r], mc->ctx[r]->stream);
        if (st == 0) st = mc->ncclAllGather(Hr + (size_t)r * cntH, Hr, cntH, 8, mc->comms[r], mc->ctx[r]->stream);
      }
      const int st2 = mc->ncclGroupEnd();
      if (st == 0) st = st2;
      if (st != 0) return mfail(mc, BLMM_ERR_HIP, std::string("ncclAllGather: ") + (mc->ncclGetErrorString ? mc->ncclGetErrorString(st) : "failed"));
    } else {
      // direct exchange: device r pulls the R-1 foreign blocks, one peer copy each (fully connected xGMI: every copy
      // has its own link)
      for (int r = 0; r < R; ++r) {
        (void)hipSetDevice(mc->dev[r]);
        double* Lr = blmm::ptr<double>(mc->dL[r]); double* Hr = blmm::ptr<double>(mc->dH[r]);
        for (int s = 0; s < R; ++s) {
          if (s == r) continue;
          const double* Ls = blmm::ptr<double>(mc->dL[s]); const double* Hs = blmm::ptr<double>(mc->dH[s]);
          hipError_t e1 = hipMemcpyPeerAsync(Lr + (size_t)s * cntL, mc->dev[r], Ls + (size_t)s * cntL, mc->dev[s], sizeof(double) * cntL, mc->ctx[r]->stream);
          hipError_t e2 = hipMemcpyPeerAsync(Hr + (size_t)s * cntH, mc->dev[r], Hs + (size_t)s * cntH, mc->dev[s], sizeof(double) * cntH, mc->ctx[r]->stream);
          if (e1 != hipSuccess || e2 != hipSuccess) return mfail(mc, BLMM_ERR_HIP, std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        }
      }
    }
    for (int r = 0; r < R; ++r) {
      (void)hipSetDevice(mc->dev[r]);
      if (hipStreamSynchronize(mc->ctx[r]->stream) != hipSuccess) return mfail(mc, BLMM_ERR_HIP, "all-gather: stream synchronisation failed");
    }
  }
  // optional copy-out for callers that passed host buffers with a device-resident mode: device 0's view
  if (L_out && h2_out && p > 0 && m > 0) {
    if (full) {
      blmm_ctx* ctx = mc->ctx[0];
      (void)hipSetDevice(ctx->device);
      if (hipMemcpy(L_out, mc->dL[0].p, sizeof(double) * (size_t)p * m, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(h2_out, mc->dH[0].p, sizeof(double) * (alt ? (size_t)p * m : (size_t)m), hipMemcpyDeviceToHost) != hipSuccess)
        return mfail(mc, BLMM_ERR_HIP, "copy-out of the gathered matrix failed");
    } else {
      rc = on_all(mc, [&](int r) -> int {
        blmm_ctx* ctx = mc->ctx[r];
        BLMM_HIP(hipSetDevice(ctx->device));
        int64_t lo, hi;
        blmm_multi_shard(m, r, R, &lo, &hi);
        if (hi > lo) {
          BLMM_HIP(hipMemcpy(L_out + (size_t)lo * p, mc->dL[r].p, sizeof(double) * (size_t)p * (hi - lo), hipMemcpyDeviceToHost));
          BLMM_HIP(hipMemcpy(alt ? h2_out + (size_t)lo * p : h2_out + lo, mc->dH[r].p,
                             sizeof(double) * (alt ? (size_t)p * (hi - lo) : (size_t)(hi - lo)), hipMemcpyDeviceToHost));
        }
        return BLMM_OK;
      });
      if (rc) return rc;
    }
  }
  return BLMM_OK;
}

// ---- consumers of the blocks where they are (no gather, nothing p x m crosses PCIe): per-trait maxima and threshold triplets
// of the LAST blmm_bulkscan_multi call, whatever its gather mode.  Column / trait indices are global (0 .. m-1).
static int block_source(blmm_multi* mc, int r, const double** dL, int64_t* lo, int64_t* hi) {
  blmm_multi_shard(mc->last_m, r, mc->ndev, lo, hi);
  if (mc->last_gather == BLMM_GATHER_HOST_SHARDS) {
    *dL = mc->ctx[r]->last_L;
    if (*hi > *lo && (!*dL || mc->ctx[r]->last_m != *hi - *lo || mc->ctx[r]->last_p != mc->last_p)) return BLMM_ERR_INVALID;
  } else {
    const int64_t off = mc->last_gather == BLMM_GATHER_ALLGATHER ? *lo : 0;
    *dL = blmm::ptr<double>(mc->dL[r]) + (size_t)off * mc->last_p;
  }
  return BLMM_OK;
}

int blmm_multi_last_colmax(blmm_multi* mc, double* max_out, int64_t* argmax_out) {
  using namespace blmm;
  if (!mc || !max_out) return BLMM_ERR_INVALID;
  if (mc->last_gather < 0) return mfail(mc, BLMM_ERR_INVALID, "multi_last_colmax: no previous blmm_bulkscan_multi call");
  DeviceRestore keep_device;
  return on_all(mc, [&](int r) -> int {
    blmm_ctx* ctx = mc->ctx[r];
    const double* dL; int64_t lo, hi;
    if (block_source(mc, r, &dL, &lo, &hi)) return fail(ctx, BLMM_ERR_INVALID, "multi_last_colmax: the block of the last call is no longer resident");
    if (hi <= lo) return BLMM_OK;
    BLMM_HIP(hipSetDevice(ctx->device));
    int e;
    if ((e = ensure(ctx, ctx->tmpA, sizeof(double) * (size_t)(hi - lo)))) return e;
    if ((e = ensure(ctx, ctx->tmpB, sizeof(int64_t) * (size_t)(hi - lo)))) return e;
    if ((e = launch_colmax(ctx, dL, mc->last_p, hi - lo, mc->last_p > 0 ? mc->last_p : 1, ptr<double>(ctx->tmpA), ptr<int64_t>(ctx->tmpB)))) return e;
    BLMM_HIP(hipMemcpyAsync(max_out + lo, ctx->tmpA.p, sizeof(double) * (size_t)(hi - lo), hipMemcpyDeviceToHost, ctx->stream));
    if (argmax_out) BLMM_HIP(hipMemcpyAsync(argmax_out + lo, ctx->tmpB.p, sizeof(int64_t) * (size_t)(hi - lo), hipMemcpyDeviceToHost, ctx->stream));
    BLMM_HIP(hipStreamSynchronize(ctx->stream));
    return BLMM_OK;
  });
}

int blmm_multi_last_lod_threshold(blmm_multi* mc, double thr, int64_t cap, int32_t* i_out, int32_t* j_out, double* lod_out, int64_t* count_out) {
  using namespace blmm;
  if (!mc || !count_out || cap < 0 || (cap > 0 && (!i_out || !j_out || !lod_out))) return BLMM_ERR_INVALID;
  if (mc->last_gather < 0) return mfail(mc, BLMM_ERR_INVALID, "multi_last_lod_threshold: no previous blmm_bulkscan_multi call");
  DeviceRestore keep_device;
  const int R = mc->ndev;
  // every device filters its block with the whole cap; the host concatenates in device order up to cap
  std::vector<std::vector<int32_t>> vi(R), vj(R); std::vector<std::vector<double>> vl(R); std::vector<int64_t> cnt(R, 0);
  int rc = on_all(mc, [&](int r) -> int {
    blmm_ctx* ctx = mc->ctx[r];
    const double* dL; int64_t lo, hi;
    if (block_source(mc, r, &dL, &lo, &hi)) return fail(ctx, BLMM_ERR_INVALID, "multi_last_lod_threshold: the block of the last call is no longer resident");
    if (hi <= lo || mc->last_p <= 0) return BLMM_OK;
    BLMM_HIP(hipSetDevice(ctx->device));
    const int64_t c1 = cap > 0 ? cap : 1;
    int e;
    if ((e = ensure(ctx, ctx->redtrip, (sizeof(double) + 2 * sizeof(int32_t)) * (size_t)c1 + 64))) return e;
    int64_t* dcount = ptr<int64_t>(ctx->redtrip);
    double* dl = reinterpret_cast<double*>(dcount + 8);
    int32_t* di = reinterpret_cast<int32_t*>(dl + c1); int32_t* dj = di + c1;
    if ((e = launch_threshold(ctx, dL, mc->last_p, hi - lo, mc->last_p, thr, cap, di, dj, dl, dcount))) return e;
    BLMM_HIP(hipMemcpyAsync(&cnt[r], dcount, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    BLMM_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t got = cnt[r] < cap ? cnt[r] : cap;
    if (got > 0) {
      vi[r].resize(got); vj[r].resize(got); vl[r].resize(got);
      BLMM_HIP(hipMemcpyAsync(vl[r].data(), dl, sizeof(double) * (size_t)got, hipMemcpyDeviceToHost, ctx->stream));
      BLMM_HIP(hipMemcpyAsync(vi[r].data(), di, sizeof(int32_t) * (size_t)got, hipMemcpyDeviceToHost, ctx->stream));
      BLMM_HIP(hipMemcpyAsync(vj[r].data(), dj, sizeof(int32_t) * (size_t)got, hipMemcpyDeviceToHost, ctx->stream));
      BLMM_HIP(hipStreamSynchronize(ctx->stream));
      for (auto& j : vj[r]) j += (int32_t)lo;
    }
    return BLMM_OK;
  });
  if (rc) return rc;
  int64_t total = 0, at = 0;
  for (int r = 0; r < R; ++r) {
    total += cnt[r];
    for (size_t k = 0; k < vi[r].size() && at < cap; ++k, ++at) { i_out[at] = vi[r][k]; j_out[at] = vj[r][k]; lod_out[at] = vl[r][k]; }
  }
  *count_out = total;
  return BLMM_OK;
}

int blmm_multi_device_result(blmm_multi* mc, int rank, double** dL, int64_t* ldL, int64_t* col_lo, int64_t* col_hi, double** dh2) {
  if (!mc || rank < 0 || rank >= mc->ndev) return BLMM_ERR_INVALID;
  if (mc->last_gather != BLMM_GATHER_NONE && mc->last_gather != BLMM_GATHER_ALLGATHER)
    return mfail(mc, BLMM_ERR_INVALID, "no device-resident result: the last call used gather_mode host_shards");
  int64_t lo, hi;
  blmm_multi_shard(mc->last_m, rank, mc->ndev, &lo, &hi);
  const bool full = mc->last_gather == BLMM_GATHER_ALLGATHER;
  if (dL) *dL = blmm::ptr<double>(mc->dL[rank]);
  if (dh2) *dh2 = blmm::ptr<double>(mc->dH[rank]);
  if (ldL) *ldL = mc->last_p > 0 ? mc->last_p : 1;
  if (col_lo) *col_lo = full ? 0 : lo;
  if (col_hi) *col_hi = full ? mc->last_m : hi;
  return BLMM_OK;
}

}  // extern "C"
