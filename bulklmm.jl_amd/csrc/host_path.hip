// host_path.hip -- the device -> host leg of the host-pointer entry points (blmm_bulkscan, blmm_scan_perms, ...).
//
// The drop-in API hands L back in caller memory (the reference returns a Julia Array).  At BXD size that is 2.08 GB over
// ONE PCIe Gen5 x16 link: >= 33 ms at the 63 GB/s of the link against 2.3 ms for the whole scan, so this leg IS the
// end-to-end time.  Two routes:
//   * destination pinned (blmm_host_alloc, or the caller's own array after blmm_host_register): one asynchronous copy
//     straight into it at link rate;
//   * destination pageable: hipMemcpy would stage through the runtime's single bounce buffer (measured 22 GB/s).  Here
//     the matrix moves in 32 MB pieces through a ring of pinned buffers; while piece c+1 .. c+3 are in flight on the link,
//     a small pool of host threads copies piece c from the ring into the caller's pages.
#include "blmm_internal.h"
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace blmm {

// A fixed pool of host threads that split one memcpy; job hand-off by generation counter under one mutex.
struct CopyPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  uint64_t gen = 0;
  int pending = 0;
  bool quit = false;
  char* dst = nullptr; const char* src = nullptr; size_t bytes = 0;
  int nthreads = 0;

  explicit CopyPool(int n) : nthreads(n) {
    for (int i = 0; i < n; ++i) th.emplace_back([this, i] { run(i); });
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv_go.notify_all();
    for (auto& t : th) if (t.joinable()) t.join();
  }
  static void part(char* d, const char* s, size_t bytes, int i, int n) {
    // 4 KB aligned slices: no two threads write the same page
    const size_t per = ((bytes / (size_t)n) + 4095) & ~(size_t)4095;
    const size_t lo = per * (size_t)i;
    if (lo >= bytes) return;
    const size_t len = (lo + per > bytes) ? bytes - lo : per;
    std::memcpy(d + lo, s + lo, len);
  }
  void run(int i) {
    uint64_t seen = 0;
    for (;;) {
      char* d; const char* s; size_t b;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_go.wait(lk, [&] { return quit || gen != seen; });
        if (quit) return;
        seen = gen; d = dst; s = src; b = bytes;
      }
      part(d, s, b, i + 1, nthreads + 1);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }
  // the calling thread takes slice 0
  void copy(void* d, const void* s, size_t b) {
    {
      std::lock_guard<std::mutex> lk(mu);
      dst = static_cast<char*>(d); src = static_cast<const char*>(s); bytes = b; pending = nthreads; ++gen;
    }
    cv_go.notify_all();
    part(static_cast<char*>(d), static_cast<const char*>(s), b, 0, nthreads + 1);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
};

struct HostStage {
  static constexpr int NB = 4;
  static constexpr size_t CH = (size_t)32 << 20;
  void* buf[NB] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev[NB] = {nullptr, nullptr, nullptr, nullptr};
  CopyPool* pool = nullptr;
};

void destroy_host_stage(HostStage* hs) {
  if (!hs) return;
  delete hs->pool;
  for (int i = 0; i < HostStage::NB; ++i) {
    if (hs->buf[i]) (void)hipHostFree(hs->buf[i]);
    if (hs->ev[i]) (void)hipEventDestroy(hs->ev[i]);
  }
  delete hs;
}

static bool is_pinned(const void* p) {
  hipPointerAttribute_t at;
  std::memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

// Device -> caller memory on ctx->stream's order; returns when the bytes are in place.
int copy_to_host(blmm_ctx* ctx, void* dst, const void* dsrc, size_t bytes) {
  if (bytes == 0) return BLMM_OK;
  const char* mode = dev_env("BLMM_D2H");   // "plain": one hipMemcpyAsync whatever the destination (A/B timing)
  if (bytes < ((size_t)8 << 20) || is_pinned(dst) || (mode && std::strcmp(mode, "plain") == 0)) {
    BLMM_HIP(hipMemcpyAsync(dst, dsrc, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BLMM_HIP(hipStreamSynchronize(ctx->stream));
    return BLMM_OK;
  }
  if (!ctx->hstage) {
    HostStage* hs = new HostStage();
    for (int i = 0; i < HostStage::NB; ++i) {
      if (hipHostMalloc(&hs->buf[i], HostStage::CH, hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&hs->ev[i], hipEventDisableTiming) != hipSuccess) {
        destroy_host_stage(hs);
        return fail(ctx, BLMM_ERR_ALLOC, "pinned staging ring: hipHostMalloc failed");
      }
    }
    unsigned hw = std::thread::hardware_concurrency();
    int nt = hw >= 16 ? 7 : (hw >= 8 ? 3 : 1);
    if (const char* e = dev_env("BLMM_D2H_THREADS")) nt = std::max(0, atoi(e) - 1);
    hs->pool = new CopyPool(nt);
    ctx->hstage = hs;
  }
  HostStage* hs = ctx->hstage;
  const size_t CH = HostStage::CH;
  const size_t nch = (bytes + CH - 1) / CH;
  auto issue = [&](size_t c) -> hipError_t {
    const size_t off = c * CH, len = (off + CH > bytes) ? bytes - off : CH;
    hipError_t e = hipMemcpyAsync(hs->buf[c % HostStage::NB], static_cast<const char*>(dsrc) + off, len, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return e;
    return hipEventRecord(hs->ev[c % HostStage::NB], ctx->stream);
  };
  for (size_t c = 0; c < nch && c < (size_t)HostStage::NB; ++c) BLMM_HIP(issue(c));
  for (size_t c = 0; c < nch; ++c) {
    BLMM_HIP(hipEventSynchronize(hs->ev[c % HostStage::NB]));
    const size_t off = c * CH, len = (off + CH > bytes) ? bytes - off : CH;
    hs->pool->copy(static_cast<char*>(dst) + off, hs->buf[c % HostStage::NB], len);
    if (c + HostStage::NB < nch) BLMM_HIP(issue(c + HostStage::NB));
  }
  return BLMM_OK;
}

}  // namespace blmm

extern "C" {

int blmm_host_register(void* p, uint64_t bytes) {
  if (!p || !bytes) return BLMM_ERR_INVALID;
  return hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault) == hipSuccess ? BLMM_OK : BLMM_ERR_HIP;
}

int blmm_host_unregister(void* p) {
  if (!p) return BLMM_ERR_INVALID;
  return hipHostUnregister(p) == hipSuccess ? BLMM_OK : BLMM_ERR_HIP;
}

void* blmm_host_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void blmm_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

}  // extern "C"
