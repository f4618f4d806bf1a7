// blmm_api.hip -- C ABI (include/bulklmm_hip.h) and the host orchestration of the bulkscan pipeline:
//   design -> eigen (device Jacobi) -> rotation GEMM -> null-model h2 (Brent / grid) -> panels -> LOD kernels.
// Mirrors bulkscan / bulkscan_null / bulkscan_null_grid / bulkscan_alt_grid (src/bulkscan.jl) and
// scan_perms_lite (src/scan.jl:485-557) of BulkLMM.jl; nothing here falls back to a CPU path.
#include "blmm_internal.h"
#include "fastmath.h"
#include <cmath>
#include <cstddef>
#include <chrono>
#include <cstring>
#include <limits>
#include <cstdlib>
#include <mutex>

using namespace blmm;

namespace blmm {

int fail(blmm_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

int ensure(blmm_ctx* ctx, DevBuf& b, size_t bytes) {
  // whoever asks for the output buffer is about to overwrite (or reallocate) it: the blmm_last_* consumers must not see the
  // previous call's matrix through it.  The entry points set last_L again once their own result is in place.
  if (&b == &ctx->outL) ctx->last_L = nullptr;
  if (&b == &ctx->outL || &b == &ctx->outP) ctx->last_P = nullptr;
  if (bytes == 0) bytes = 8;
  if (b.cap >= bytes) return BLMM_OK;
  if (b.p) {
    // outstanding work may still use the old buffer
    hipStreamSynchronize(ctx->stream);
    hipFree(b.p);
    b.p = nullptr; b.cap = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    return fail(ctx, BLMM_ERR_ALLOC, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
  }
  b.cap = want;
  return BLMM_OK;
}

namespace {
std::mutex g_grid_mu[64];           // one per device: held across wait -> launch -> record (GridKernelGuard)
hipEvent_t g_grid_ev[64];
bool g_grid_has[64];
}  // namespace

GridKernelGuard::GridKernelGuard(blmm_ctx* c) : ctx(c), dev(c->device & 63) {
  g_grid_mu[dev].lock();
  if (g_grid_has[dev] && hipStreamWaitEvent(ctx->stream, g_grid_ev[dev], 0) != hipSuccess)
    rc = fail(ctx, BLMM_ERR_HIP, "hipStreamWaitEvent (grid-kernel order) failed");
}

int GridKernelGuard::record() {
  if (!g_grid_has[dev]) {
    if (hipEventCreateWithFlags(&g_grid_ev[dev], hipEventDisableTiming) != hipSuccess)
      return fail(ctx, BLMM_ERR_HIP, "hipEventCreate (grid-kernel order) failed");
    g_grid_has[dev] = true;
  }
  if (hipEventRecord(g_grid_ev[dev], ctx->stream) != hipSuccess)
    return fail(ctx, BLMM_ERR_HIP, "hipEventRecord (grid-kernel order) failed");
  return BLMM_OK;
}

GridKernelGuard::~GridKernelGuard() { g_grid_mu[dev].unlock(); }

__global__ void k_fill(double* p, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// column-major n x ncols  ->  row-major npad x ld (zero padded)
__global__ void k_to_rowmajor(const double* __restrict__ In, int n, int64_t ncols, double* __restrict__ Out, int npad, int64_t ld) {
  __shared__ double tile[32][33];
  const int64_t c0 = (int64_t)blockIdx.x * 32;
  const int r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int cc = ty; cc < 32; cc += 8) {
    const int64_t col = c0 + cc; const int row = r0 + tx;
    tile[cc][tx] = (row < n && col < ncols) ? In[col * n + row] : 0.0;
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int row = r0 + rr; const int64_t col = c0 + tx;
    if (row < npad && col < ld) Out[(int64_t)row * ld + col] = tile[tx][rr];
  }
}

// Copies the device-side "gave up" conditions of a call into the context's pinned host word (system-scope store):
// bit 0: the multi-workgroup weight-basis kernel timed out at its grid barrier (stat[8] < 0);
// bit 1: the eigensolver gave up (stat[11]: -7 grid barrier of the tridiagonalisation timed out, -8 QL iteration limit).
__global__ void k_sticky(const int64_t* __restrict__ stat, int64_t* hflag) {
  int64_t f = 0;
  if (stat[8] < 0) f |= 1;
  if (stat[11] != 0) { f |= 2; hflag[1] = stat[11]; }      // the code itself: -7 / -8, the own solver's aborts
  if (f) __hip_atomic_fetch_or(hflag, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace blmm

namespace {

// A device-side failure of an EARLIER call made without a blmm_status (nothing synchronised then) surfaces here.
int check_sticky(blmm_ctx* ctx) {
  if (!ctx->hflag) return BLMM_OK;
  const int64_t f = *ctx->hflag;
  if (!f) return BLMM_OK;
  *ctx->hflag = 0;
  if (f & 1) return fail(ctx, BLMM_ERR_HIP, "a call failed on the device: the weight-basis kernel timed out at its grid barrier (its LOD output is NaN)");
  const long long code = (long long)ctx->hflag[1];
  return fail(ctx, BLMM_ERR_HIP, "a call failed on the device: the eigensolver did not converge (code " + std::to_string(code) +
                                 ": -7 grid barrier of the tridiagonalisation timed out, -8 QL iteration limit)");
}

struct Timer {
  blmm_ctx* ctx; blmm_ctx::EvSet* set = nullptr;
  explicit Timer(blmm_ctx* c) : ctx(c) {
    if (!ctx->timing) return;
    if (ctx->ev_used >= 4096) ctx->ev_used = 0;  // nobody is reading: recycle
    if (ctx->ev_used == ctx->evsets.size()) {
      blmm_ctx::EvSet s; s.n = 0;
      for (auto& e : s.e) (void)hipEventCreate(&e);
      ctx->evsets.push_back(s);
    }
    set = &ctx->evsets[ctx->ev_used++];
    set->n = 0;
  }
  void mark() { if (set && set->n < 8) (void)hipEventRecord(set->e[set->n++], ctx->stream); }
};

// phase times of one event set; marks: 0 start, 1 eigen done, 2 rotate done, 3 h2 done, 4 prep done, 5 scan done
void phase_times(const blmm_ctx::EvSet& s, double out[6]) {
  for (int i = 0; i < 6; ++i) out[i] = 0.0;
  for (int i = 0; i + 1 < s.n && i < 5; ++i) { float ms = 0; (void)hipEventElapsedTime(&ms, s.e[i], s.e[i + 1]); out[i] = ms; }
  if (s.n >= 2) { float tot = 0; (void)hipEventElapsedTime(&tot, s.e[0], s.e[s.n - 1]); out[5] = tot; }
}

int to_rowmajor(blmm_ctx* ctx, const double* In, int n, int64_t ncols, double* Out, int npad, int64_t ld) {
  dim3 grid((unsigned)((ld + 31) / 32), (unsigned)((npad + 31) / 32));
  hipLaunchKernelGGL(k_to_rowmajor, grid, dim3(256), 0, ctx->stream, In, n, ncols, Out, npad, ld);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ctx, BLMM_ERR_HIP, std::string("k_to_rowmajor: ") + hipGetErrorString(e));
  return BLMM_OK;
}

int fill(blmm_ctx* ctx, double* p, int64_t n, double v) {
  if (n <= 0) return BLMM_OK;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, p, n, v);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ctx, BLMM_ERR_HIP, std::string("k_fill: ") + hipGetErrorString(e));
  return BLMM_OK;
}

int check_opts(blmm_ctx* ctx, const blmm_opts* o) {
  if (!o) return fail(ctx, BLMM_ERR_INVALID, "opts is NULL");
  if (o->decomp_scheme != BLMM_EIGEN && o->decomp_scheme != BLMM_SVD)
    return fail(ctx, BLMM_ERR_DECOMP, "Please choose either `eigen` or `svd` for decomposition of the kinship matrix.");
  return BLMM_OK;
}

int reset_stat(blmm_ctx* ctx, int64_t** stat) {
  int rc = ensure(ctx, ctx->stat, sizeof(int64_t) * NSTAT);
  if (rc) return rc;
  *stat = ptr<int64_t>(ctx->stat);
  ctx->audit_ran = false;
  ctx->brent_cnt_used = false;
  BLMM_HIP(hipMemsetAsync(*stat, 0, sizeof(int64_t) * NSTAT, ctx->stream));
  return BLMM_OK;
}

// Synchronises and fills *status (only when the caller asked for it).
int finish_status(blmm_ctx* ctx, blmm_status* st, Timer* tm) {
  if (!st) return BLMM_OK;
  std::memset(st, 0, sizeof(*st));
  int64_t h[NSTAT];
  BLMM_HIP(hipMemcpyAsync(h, ctx->stat.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  st->n_neg_eig = h[ST_NEG_EIG];
  st->n_nonpos_weight = h[ST_NONPOS_W];
  st->n_zero_norm = h[ST_ZERO_NORM];
  st->n_nan_lod = h[ST_NAN_LOD];
  st->n_brent_maxiter = h[ST_BRENT_MAXIT];
  st->jacobi_sweeps = h[ST_JACOBI_SWEEPS];
  st->jacobi_cycles = h[6]; st->jacobi_ticks_100mhz = h[7];
  st->lowrank_rank = h[8];
  st->lowrank_fallback = h[10];
  st->lowrank_shared = h[12] + h[14];
  st->n_h2_boundary = h[ST_H2_BOUNDARY];
  st->n_h2_multimodal = ctx->audit_ran ? h[ST_H2_MULTIMODAL] : -1;
  st->n_illcond_rescan = h[ST_ILLCOND];
  if (h[8] < 0) return fail(ctx, BLMM_ERR_HIP, "weight-basis kernel: a workgroup timed out at the grid barrier");
  if (h[11] != 0) return fail(ctx, BLMM_ERR_HIP, "the eigensolver did not converge (code " + std::to_string((long long)h[11]) +
                              ": -7 grid barrier of the tridiagonalisation timed out, -8 QL iteration limit)");
  if (ctx->hflag && *ctx->hflag) return check_sticky(ctx);
  { double r2; std::memcpy(&r2, &h[9], sizeof(double)); st->lowrank_resid = std::sqrt(r2 < 0 ? 0.0 : r2); }
  if (tm && tm->set && tm->set->n >= 2) {
    double t[6];
    phase_times(*tm->set, t);
    st->t_eigen_ms = t[0]; st->t_rotate_ms = t[1]; st->t_h2_ms = t[2]; st->t_prep_ms = t[3]; st->t_scan_ms = t[4];
    st->t_total_ms = t[5];
  }
  return BLMM_OK;
}

// Every bulkscan / scan call ends here.  Large-n calls (multi-workgroup weight basis, vendor eigensolver) first copy
// their device-side failure conditions into the sticky host word, so that they are reported even without a status.
int end_call(blmm_ctx* ctx, const Pipe& P, blmm_status* st, Timer* tm) {
  if (P.big && ctx->hflag && P.stat) {
    hipLaunchKernelGGL(k_sticky, dim3(1), dim3(1), 0, ctx->stream, P.stat, const_cast<int64_t*>(ctx->hflag));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, BLMM_ERR_HIP, std::string("k_sticky: ") + hipGetErrorString(e));
  }
  return finish_status(ctx, st, tm);
}

// design -> eigen (prepare_eigen) -> rotation of Y and G (prepare_rotate).  centered = 1: the rotation also removes the
// unweighted projection on the null covariates (kernels_prep.hip:k_post_eigen).
int prepare_eigen(blmm_ctx* ctx, const blmm_opts* o, int64_t n, const double* dCovar, int64_t ncov, const double* dK,
                  const double* dweights, int centered, Pipe& P, Timer& tm) {
  // whatever blmm_prepare_dev left in this context (Rp, Z0, lambda, the status block) is about to be overwritten or reallocated:
  // the *_prerotated entry points must not run on it afterwards (blmm_prepare_dev sets the flag again when IT got here)
  ctx->prep_valid = false;
  if (n < 1 || ncov < 0) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  if (n > 2048) return fail(ctx, BLMM_ERR_UNSUPPORTED, "more than 2048 individuals: the device eigensolver (tridiagonalisation + divide and conquer) stops at n = 2048");
  int add_int = o->add_intercept ? 1 : 0;
  if (ncov == 0 || !dCovar) { add_int = 1; ncov = 0; dCovar = nullptr; }  // bulkscan(Y,G,K): intercept-only null model
  const int c = (int)ncov + add_int;
  if (c < 1 || c > CMAX) return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
  if (c >= n) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  P.n = (int)n; P.c = c; P.npad = (int)round_up(n, 8); P.ldr = (int)round_up(P.npad, 16);   // K padded to 8: even K-step count
  int rc;
  if ((rc = ensure(ctx, ctx->Ks, sizeof(double) * n * n))) return rc;
  if ((rc = ensure(ctx, ctx->V, sizeof(double) * (n * n + 4 * n + 16)))) return rc;
  if ((rc = ensure(ctx, ctx->lraw, sizeof(double) * n))) return rc;
  if ((rc = ensure(ctx, ctx->U, sizeof(double) * n * n))) return rc;
  if ((rc = ensure(ctx, ctx->lam, sizeof(double) * n))) return rc;
  if ((rc = ensure(ctx, ctx->Zs, sizeof(double) * n * c))) return rc;
  if ((rc = ensure(ctx, ctx->Z0, sizeof(double) * n * c))) return rc;
  if ((rc = ensure(ctx, ctx->Rp, sizeof(double) * (size_t)P.npad * P.ldr))) return rc;
  if ((rc = reset_stat(ctx, &P.stat))) return rc;
  P.Z0 = ptr<double>(ctx->Z0); P.lam = ptr<double>(ctx->lam);
  tm.mark();
  if ((rc = launch_design(ctx, dK, dCovar, (int)ncov, add_int, dweights, (int)n, ptr<double>(ctx->Ks), ptr<double>(ctx->Zs)))) return rc;
  const double* evec = ptr<double>(ctx->V);
  // n <= 124: LDS Jacobi.  Beyond: the own tridiagonalisation + divide-and-conquer solver (kernels_eig.hip) up to n = 2048 (its
  // reduction keeps the matrix in LDS up to ~1450 and in L2-resident global memory beyond).  No vendor library: rocSOLVER's first
  // use in a process takes MINUTES on this image (code-object load) and its path could not be part of the default test run.
  // BLMM_EIGEN = dc | jacobi overrides the choice (A/B timing and tests; "dc" also for n <= 124; "jacobi" beyond 124 is the
  // single-workgroup global-memory Jacobi: 0.74 s at n = 333).
  const char* eig_env = dev_env("BLMM_EIGEN");                 // tuning key "eigen_solver": 1 = "jacobi", 2 = "dc"
  if (!eig_env && ctx->tune.eigen_solver == 1) eig_env = "jacobi";
  if (!eig_env && ctx->tune.eigen_solver == 2) eig_env = "dc";
  const bool big = n > jacobi_lds_max_n();
  P.big = big;
  bool done = false, post_done = false;
  const bool want_dc = eig_env ? std::strcmp(eig_env, "dc") == 0 : big;
  if (!done && want_dc && n >= 3) {
    rc = launch_eig_dc(ctx, ptr<double>(ctx->Ks), (int)n, ptr<double>(ctx->lraw), ptr<double>(ctx->V), P.stat);
    if (rc == BLMM_OK) { done = true; P.big = true; }
    else if (rc != BLMM_ERR_UNSUPPORTED) return rc;
  }
  if (!done) {
    // n <= 124: the fast path first (tridiagonalisation, Sturm multi-section, twisted factorisation; kernels_eig.hip: k_eigf_*).
    // It checks its own result on the device; the Jacobi behind it is a no-op when the checks passed and the whole solver when
    // they did not (numerically repeated eigenvalues).  BLMM_EIGEN=jacobi: the Jacobi alone (A/B timing, tests).
    const bool want_fast = !(eig_env && std::strcmp(eig_env, "jacobi") == 0) && n >= 3 && n <= eig_fast_max_n();
    if (want_fast) {
      rc = launch_eig_fast(ctx, ptr<double>(ctx->Ks), (int)n, ptr<double>(ctx->lraw), ptr<double>(ctx->V), P.stat);
      if (rc != BLMM_OK && rc != BLMM_ERR_UNSUPPORTED) return rc;
    }
    // (the post-eigen work rides in the tail of this launch where its LDS fits: one dependent-launch boundary less)
    if ((rc = launch_jacobi_post(ctx, ptr<double>(ctx->Ks), ptr<double>(ctx->V), (int)n, ptr<double>(ctx->lraw), P.stat, ptr<double>(ctx->Zs),
                                 dweights, c, P.npad, P.ldr, o->decomp_scheme, centered, P.lam, ptr<double>(ctx->U), P.Z0, ptr<double>(ctx->Rp),
                                 &post_done))) return rc;
  }
  if (!post_done && (rc = launch_post_eigen(ctx, ptr<double>(ctx->lraw), evec, ptr<double>(ctx->Zs), dweights, (int)n, c,
                              P.npad, P.ldr, o->decomp_scheme, centered, P.lam, ptr<double>(ctx->U), P.Z0,
                              ptr<double>(ctx->Rp), P.stat))) return rc;
  tm.mark();
  return BLMM_OK;
}

// the host entry points' deferred uploads (blmm_ctx::up_pending): on the copy stream, behind nothing; ev_in says when they are there
static int flush_upload(blmm_ctx* ctx) {
  if (!ctx->up_pending) return BLMM_OK;
  ctx->up_pending = false;
  for (int i = 0; i < 2; ++i) {
    if (ctx->up_bytes[i]) BLMM_HIP(hipMemcpyAsync(ctx->up_dst[i], ctx->up_src[i], ctx->up_bytes[i], hipMemcpyHostToDevice, ctx->copy));
    if (i == 0) BLMM_HIP(hipEventRecord(ctx->ev_inY, ctx->copy));     // the traits first: the main stream's rotation is the critical one
  }
  BLMM_HIP(hipEventRecord(ctx->ev_in, ctx->copy));
  ctx->in_wait = true;
  return BLMM_OK;
}

// null-exact, low-rank weights form: the basis of the weight family needs only the sorted eigenvalues, so it starts on the side
// stream right behind the eigen-decomposition, beside the rotation and the Brent search (joined before k_lr_panels)
int rotate_markers(blmm_ctx* ctx, Pipe& P, const double* dG, int64_t p);
// dG != nullptr: the marker rotation goes to the side stream as well, in front of the basis (the h2 search on the main stream needs
// only the rotated traits; the marker-side products that follow the basis on the side stream and -- behind ev_join -- the scan
// need Xt): 12 us less in front of k_brent at the BXD shape
int start_wbasis(blmm_ctx* ctx, Pipe& P, const double* dG = nullptr, int64_t p = 0, bool xt_recorded = false) {
  int rc;
  const int64_t n = P.n;
  const LrSeg seg = lr_segments(ctx, P.n);
  if ((rc = ensure(ctx, ctx->wbQ, sizeof(double) * (size_t)seg.S * P.npad * n))) return rc;
  if ((rc = ensure(ctx, ctx->wbW, sizeof(double) * (size_t)n * (256 + 16)))) return rc;
  if ((rc = ensure(ctx, ctx->wbRk, sizeof(int) * 4 * LR_SEG_MAX))) return rc;
  hipStream_t main_stream = ctx->stream;
  if (!xt_recorded) BLMM_HIP(hipEventRecord(ctx->ev_xt, main_stream));     // (else: recorded by the eigen phase's last kernel itself)
  BLMM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_xt, 0));
  ctx->wb_on_side2 = false;
  if (dG) {
    // The basis (a handful of 1024-thread workgroups, 70 us of dependent steps) goes to the SECOND side stream, beside the marker
    // rotation instead of behind it: queued behind the rotation it became dispatchable at the same moment as the h2 search,
    // whose 2,200 waves then held every CU until the first of them retired -- 126 us instead of 70, and with the marker-side
    // products behind it the critical path of the front whenever nothing else delayed the main stream (a caller that does not
    // ask for phase timings: +35 us per call; profiles/r04_timeline_notiming_*.txt).  The marker-side products follow the basis
    // on ITS stream (lr_begin: no cross-queue wait between the two, each of which costs 14-20 us here) and wait there for
    // the rotation (ev_wb), which is done long before.
    BLMM_HIP(hipStreamWaitEvent(ctx->side2, ctx->ev_xt, 0));
    ctx->stream = ctx->side2;
    rc = launch_wbasis(ctx, P.lam, (int)n, P.npad, seg, ptr<double>(ctx->wbW), ptr<double>(ctx->wbQ), ptr<int>(ctx->wbRk), P.stat);
    ctx->stream = ctx->side;                                         // the launchers enqueue on ctx->stream
    if (!rc && ctx->in_wait && hipStreamWaitEvent(ctx->side, ctx->ev_in, 0) != hipSuccess) rc = fail(ctx, BLMM_ERR_HIP, "hipStreamWaitEvent failed");
    if (!rc) rc = rotate_markers(ctx, P, dG, p);
    if (!rc && hipEventRecord(ctx->ev_wb, ctx->side) != hipSuccess) rc = fail(ctx, BLMM_ERR_HIP, "hipEventRecord failed");   // ev_wb: the rotated markers
    ctx->wb_on_side2 = true;
  } else {
    ctx->stream = ctx->side;
    rc = launch_wbasis(ctx, P.lam, (int)n, P.npad, seg, ptr<double>(ctx->wbW), ptr<double>(ctx->wbQ), ptr<int>(ctx->wbRk), P.stat);
  }
  ctx->stream = main_stream;
  return rc;
}

// The own rotation kernels sum every output element in a fixed order, so a trait's (or marker's) rotated column does not depend on
// how many OTHER columns are in the call (the sharding contract: a column block scanned alone is bit-identical,
// tests/test_gpu_configs.py; a vendor GEMM picks its kernel -- tile shape, split-K -- from the problem shape and broke exactly
// that in round 1).
int rotate_traits(blmm_ctx* ctx, Pipe& P, const double* dY, int64_t m) {
  if (m < 0) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  P.m = m; P.ldy = round_up(m > 0 ? m : 1, 128);
  int rc = ensure(ctx, ctx->Yt, sizeof(double) * (size_t)P.npad * P.ldy);
  if (rc) return rc;
  P.Yt = ptr<double>(ctx->Yt);
  return launch_rotate(ctx, ptr<double>(ctx->Rp), P.ldr, P.n, P.npad, dY, m, P.Yt, P.ldy, P.ldy);
}
int rotate_markers(blmm_ctx* ctx, Pipe& P, const double* dG, int64_t p) {
  if (p < 0) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  P.p = p; P.ldx = round_up(p > 0 ? p : 1, 128);
  int rc = ensure(ctx, ctx->Xt, sizeof(double) * (size_t)P.npad * P.ldx);
  if (rc) return rc;
  P.Xt = ptr<double>(ctx->Xt);
  return launch_rotate(ctx, ptr<double>(ctx->Rp), P.ldr, P.n, P.npad, dG, p, P.Xt, P.ldx, P.ldx);
}

int prepare(blmm_ctx* ctx, const blmm_opts* o, const double* dY, int64_t n, int64_t m, const double* dG, int64_t p,
            const double* dCovar, int64_t ncov, const double* dK, const double* dweights, int centered, Pipe& P, Timer& tm,
            bool early_wbasis = false, bool grid_side = false, bool skip_markers = false) {
  if (m < 0 || p < 0) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  // the side streams fork where the eigen phase ends: its last kernel records the fork event itself (BLMM_LAUNCH_STOP)
  const bool forks = m > 0 && p > 0 && (early_wbasis || grid_side);
  ctx->stop_event_used = false;
  ctx->stop_event_next = forks ? ctx->ev_xt : nullptr;
  int rc = prepare_eigen(ctx, o, n, dCovar, ncov, dK, dweights, centered, P, tm);
  const bool xt_recorded = ctx->stop_event_used;
  ctx->stop_event_next = nullptr; ctx->stop_event_used = false;
  if (rc) { ctx->up_pending = false; return rc; }
  // (host entry points) the eigen phase is queued: now the traits and the markers go up, beside it; this stream reads them next
  if (ctx->up_pending) {
    if ((rc = flush_upload(ctx))) return rc;
    BLMM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_inY, 0));
  }
  const bool up_flight = ctx->in_wait;
  // (BLMM_ROTATE_SIDE=0: the marker rotation stays on the main stream behind the traits' -- A/B testing)
  static const bool rot_side = !(dev_env("BLMM_ROTATE_SIDE") && dev_env("BLMM_ROTATE_SIDE")[0] == '0');
  // only where the rotation is the small latency-bound kernel (n <= 160): the GEMM of larger n fills the chip by itself, and behind it
  // the basis and the marker-side products come later (n = 500 shard: 6.32 against 6.29 ms; BXD: 1.714 against 1.734 ms, 4 A/B rounds)
  const bool side_g = early_wbasis && m > 0 && p > 0 && rot_side && n <= 160;
  if (early_wbasis && m > 0 && p > 0 && (rc = start_wbasis(ctx, P, side_g ? dG : nullptr, p, xt_recorded))) return rc;
  P.xt_side = side_g;
  // the grid methods (grid_side): the marker rotation and, later, the marker norms of every grid point (launch_isx) on the side stream,
  // beside the traits' rotation, the grid log-likelihoods and the trait panels on the main stream; joined in front of the scan
  if (grid_side && !early_wbasis && m > 0 && p > 0 && rot_side && n <= 160) {
    hipStream_t main_stream = ctx->stream;
    if (!xt_recorded) BLMM_HIP(hipEventRecord(ctx->ev_xt, main_stream));
    BLMM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_xt, 0));
    if (ctx->in_wait) BLMM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_in, 0));
    ctx->stream = ctx->side;
    rc = rotate_markers(ctx, P, dG, p);
    ctx->stream = main_stream;
    if (rc) return rc;
    P.xt_side = true;
  }
  if ((rc = rotate_traits(ctx, P, dY, m))) return rc;
  if (skip_markers) { P.p = p; P.ldx = round_up(p > 0 ? p : 1, 128); P.Xt = nullptr; }   // the caller rotates them itself (fp32 permutation path)
  else if (!P.xt_side) {
    if (up_flight) BLMM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_in, 0));     // (the markers on this stream: they went up second)
    if ((rc = rotate_markers(ctx, P, dG, p))) return rc;
  }
  ctx->in_wait = false;                     // every reader of the uploaded inputs is queued behind ev_in
  tm.mark();
  return BLMM_OK;
}

NullModel null_model(const Pipe& P, const blmm_opts* o) {
  NullModel nm;
  nm.n = P.n; nm.c = P.c; nm.npad = P.npad; nm.reml = o->reml ? 1 : 0;
  nm.optim_interval = o->optim_interval < 1 ? 1 : o->optim_interval;
  nm.prior_a = o->prior_variance; nm.prior_b = o->prior_sample_size;
  return nm;
}

int grid_to_device(blmm_ctx* ctx, const double* h2_grid_host, int64_t ngrid, double** out) {
  if (!h2_grid_host || ngrid < 1) return fail(ctx, BLMM_ERR_INVALID, "h2 grid is empty");
  for (int64_t g = 0; g < ngrid; ++g) {
    const double h = h2_grid_host[g];
    if (std::isinf(h / (1.0 - h))) return fail(ctx, BLMM_ERR_H2_ONE, "Heritability of 1 is not allowed.");
  }
  int rc = ensure(ctx, ctx->gridd, sizeof(double) * ngrid);
  if (rc) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->gridd.p, h2_grid_host, sizeof(double) * ngrid, hipMemcpyHostToDevice, ctx->stream));
  // the source is caller memory: make sure the copy has left it before we return
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  *out = ptr<double>(ctx->gridd);
  return BLMM_OK;
}

ScanArgs scan_args(blmm_ctx* ctx, const Pipe& P, const double* panels, int64_t ldp, double* L, int64_t ldL, int64_t m) {
  ScanArgs a;
  a.Xt = P.Xt; a.ldx = P.ldx; a.P = panels; a.ldp = ldp; a.pstride = (int64_t)P.npad * ldp;
  a.ks = P.npad / 4; a.n = P.n; a.p = P.p; a.m = m; a.L = L; a.ldL = ldL;
  a.isx = nullptr; a.ld_isx = 0; a.bin = nullptr; a.stat = P.stat; a.logtab = ptr<double>(ctx->logtab); a.lodtab = ptr<double>(ctx->lodtab);
  a.Pv = ctx->pv_cur; a.ldPv = ctx->pv_cur_ld; a.pvtab = ptr<double>(ctx->pvtab);
  a.red = ctx->red_cur;                    // blmm_bulkscan_reduced: the scan kernels reduce in their epilogues, L == nullptr
  a.c = P.c;
  lod_poly5_host(-0.5 * (double)P.n, a.lodc);
  return a;
}

// ---- null-exact LOD scan in the low-rank weights form (kernels_lowrank.hip), shared by bulkscan(null-exact) and the
// liteqtl_given_h2 seam.  lr_begin: weight basis (unless prepare() already started it) and the marker-side products on
// the side stream; the caller may then enqueue the h2 search on the main stream; lr_finish: per-trait panels, the
// MFMA scan, the all-trait residual guard beside it, and the full-rank re-scan of the flagged traits.
double lr_tolerance(const blmm_ctx* ctx) {
  // relative residual |w_j - Q Q'w_j| / |w_j| above which a trait's column is recomputed from the full-length sums: tuning key
  // "lr_tol" (tests: 0 flags every trait, so the re-scan kernel is compared with the oracle as a whole)
  const char* e = dev_env("BLMM_LR_TOL");
  return e ? atof(e) : ctx->tune.lr_tol;
}

// Width of one region of the panel arrays (LrRegion): k_lr_classify fills it from both ends, so it needs a whole padding
// tile beyond the traits.  The arrays hold two regions (leading dimension 2 * lr_ldq): the second one is used when the
// h2 search is split and its second kernel runs beside the scan of the traits the first kernel finished.
int64_t lr_ldq(const Pipe& P) { return P.ldy + 128 + LR_TILE * LR_SEG_MAX; }   // + a partly filled tile per weight-basis segment
// the per-region device counters of the class / segment layout (blmm_internal.h: NSTAT)
static LrRegion lr_region(const Pipe& P, int r) {
  LrRegion rg;
  rg.col0 = r * lr_ldq(P); rg.ncol = lr_ldq(P); rg.counts = P.stat + 12 + 2 * r; rg.segcnt = P.stat + 24 + 20 * r;
  return rg;
}

int lr_begin(blmm_ctx* ctx, const Pipe& P, bool wbasis_started) {
  int rc;
  ctx->lr_last_ldq = lr_ldq(P); ctx->lr_last_m = P.m;            // blmm_lowrank_columns
  const int64_t ldp = 2 * lr_ldq(P), tstride = (int64_t)P.npad * P.ldx;
  if ((rc = ensure(ctx, ctx->lrPerm, sizeof(int) * (size_t)ldp))) return rc;
  if ((rc = ensure(ctx, ctx->lrDen0, sizeof(double) * (size_t)P.ldx))) return rc;
  const LrSeg seg = lr_segments(ctx, P.n);
  if ((rc = ensure(ctx, ctx->lrT, sizeof(double) * (size_t)seg.S * (1 + P.c) * tstride))) return rc;
  if ((rc = ensure(ctx, ctx->lrC, sizeof(double) * (size_t)P.npad * ldp))) return rc;
  if ((rc = ensure(ctx, ctx->lrL, sizeof(double) * (size_t)(P.c * (P.c + 1) / 2) * ldp))) return rc;
  if ((rc = ensure(ctx, ctx->lrFlag, sizeof(int) * (size_t)ldp))) return rc;
  if ((rc = ensure(ctx, ctx->lrPart, sizeof(double) * 2 * (size_t)((P.n + 63) / 64) * ldp))) return rc;
  if ((rc = ensure(ctx, ctx->panels, sizeof(double) * (size_t)P.npad * ldp))) return rc;
  if ((rc = ensure(ctx, ctx->wbQ, sizeof(double) * (size_t)seg.S * P.npad * P.n))) return rc;
  if ((rc = ensure(ctx, ctx->wbW, sizeof(double) * (size_t)P.n * (256 + 16)))) return rc;
  if ((rc = ensure(ctx, ctx->wbRk, sizeof(int) * 4 * LR_SEG_MAX))) return rc;
  int* rk = ptr<int>(ctx->wbRk);
  hipStream_t main_stream = ctx->stream;
  if (!(wbasis_started && P.xt_side)) {                            // (the side stream rotated the markers itself: nothing of the main stream to wait for)
    BLMM_HIP(hipEventRecord(ctx->ev_fork, main_stream));          // rotated operands are ready
    BLMM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
  }
  // (start_wbasis put the basis on the second side stream: the marker-side products follow it THERE, behind the rotation's event)
  const bool on2 = wbasis_started && P.xt_side && ctx->wb_on_side2;
  hipStream_t lr_side = on2 ? ctx->side2 : ctx->side;
  ctx->stream = lr_side;                                           // the launchers enqueue on ctx->stream
  rc = BLMM_OK;
  if (!wbasis_started) rc = launch_wbasis(ctx, P.lam, P.n, P.npad, seg, ptr<double>(ctx->wbW), ptr<double>(ctx->wbQ), rk, P.stat);
  if (!rc && hipEventRecord(ctx->ev_q, lr_side) != hipSuccess) rc = fail(ctx, BLMM_ERR_HIP, "hipEventRecord failed");   // what the panels need
  if (!rc && on2 && hipStreamWaitEvent(lr_side, ctx->ev_wb, 0) != hipSuccess) rc = fail(ctx, BLMM_ERR_HIP, "hipStreamWaitEvent failed");
  if (!rc) rc = launch_lr_tpanels(ctx, P.Xt, P.ldx, P.p, P.n, P.c, P.npad, P.Z0, ptr<double>(ctx->wbQ), rk, seg, ptr<double>(ctx->lrT), tstride,
                                  ptr<double>(ctx->lrDen0));
  ctx->stream = main_stream;
  if (rc) return rc;
  BLMM_HIP(hipEventRecord(ctx->ev_join, lr_side));                  // ... and what the scan needs on top
  // (the column order's preset, -1 = padding, is written by the count pass of k_lr_classify)
  return BLMM_OK;
}

namespace {
// the shared-weights class uses the tolerance of the expansion guard; BLMM_LR_SHARED=0 switches the class off (A/B testing)
double lr_shared_tol(const blmm_ctx* ctx) {
  // tuning key "lr_shared" = 0: no class (bench.py times the same process with and without it: ms_per_step_all_rank_form)
  const char* e = dev_env("BLMM_LR_SHARED");
  const bool on = e ? e[0] != '0' : ctx->tune.lr_shared != 0;
  return on ? lr_tolerance(ctx) : 0.0;
}
LrArgs lr_args(blmm_ctx* ctx, const Pipe& P, const LrRegion& rg, double* dL, int64_t ldL) {
  const int64_t ldp = 2 * lr_ldq(P);
  LrArgs la;
  // a.m sizes the grid only: a region holds at most m traits, in at most m/64 + 3 trait tiles
  la.s = scan_args(ctx, P, ptr<double>(ctx->panels), ldp, dL, ldL, P.m + 192 < rg.ncol ? P.m + 192 : rg.ncol);
  la.Cp = ptr<double>(ctx->lrC); la.T = ptr<double>(ctx->lrT); la.tstride = (int64_t)P.npad * P.ldx; la.Ls = ptr<double>(ctx->lrL);
  la.rk = ptr<int>(ctx->wbRk); la.c = P.c; la.perm = ptr<int>(ctx->lrPerm); la.rg = rg; la.den0 = ptr<double>(ctx->lrDen0);
  la.skip_shared = 0;
  la.seg = lr_segments(ctx, P.n);
  return la;
}
// LOD scan of one region on the current stream: the shared-weights class through the table kernel (one bin, 4 waves per
// SIMD), the other traits through k_scan_lr.  BLMM_LR_LEAN=0: both classes in k_scan_lr (A/B testing).
int lr_region_scan(blmm_ctx* ctx, const Pipe& P, const LrRegion& rg, double* dL, int64_t ldL) {
  static const bool lean = !(dev_env("BLMM_LR_LEAN") && dev_env("BLMM_LR_LEAN")[0] == '0');
  // diagnostic: the scan kernels alone on the chip (their rocprofv3 durations are then free of the side streams' kernels)
  static const bool serial = dev_env("BLMM_LR_SERIAL") && dev_env("BLMM_LR_SERIAL")[0] == '1';
  if (serial) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamSynchronize(ctx->side2); (void)hipStreamSynchronize(ctx->stream); }
  LrArgs la = lr_args(ctx, P, rg, dL, ldL);
  int rc;
  if (lean) {
    ScanArgs a = la.s;
    a.isx = la.den0; a.ld_isx = P.ldx; a.bin = nullptr;
    a.perm = la.perm; a.col0 = rg.col0; a.count = rg.counts;
    if ((rc = launch_scan_shared(ctx, a))) return rc;
    la.skip_shared = 1;
  }
  return launch_scan_lr(ctx, la);
}
// panels of one region on the current stream
int lr_region_panels(blmm_ctx* ctx, const Pipe& P, const NullModel& nm, const double* dh2, const LrRegion& rg) {
  return launch_lr_panels(ctx, nm, P.Yt, P.ldy, P.m, P.Z0, P.lam, dh2, ptr<double>(ctx->wbQ), ptr<int>(ctx->wbRk), lr_segments(ctx, P.n), ptr<int>(ctx->lrPerm),
                          rg, ptr<double>(ctx->panels), ptr<double>(ctx->lrC), ptr<double>(ctx->lrL), 2 * lr_ldq(P), P.stat);
}
int lr_region_resid(blmm_ctx* ctx, const Pipe& P, const NullModel& nm, const double* dh2, const LrRegion& rg) {
  return launch_lr_resid(ctx, nm, P.m, lr_tolerance(ctx), P.lam, dh2, ptr<double>(ctx->wbQ), ptr<int>(ctx->wbRk), lr_segments(ctx, P.n), ptr<int>(ctx->lrPerm), rg,
                         ptr<double>(ctx->lrC), 2 * lr_ldq(P), ptr<int>(ctx->lrFlag), ptr<double>(ctx->lrPart), P.stat);
}
int lr_fix(blmm_ctx* ctx, const Pipe& P, const NullModel& nm, const double* dh2, double* dL, int64_t ldL) {
  // flagged traits (normally none: the kernel reads the count on the device and returns): full-length sums
  if (ctx->red_cur.pmax) return BLMM_OK;   // reduce-in-epilogue: there is no L to patch -- reduced_impl() reads the count and re-runs
  return launch_scan_fix(ctx, nm, P.Xt, P.ldx, P.p, ptr<double>(ctx->panels), ptr<double>(ctx->lrL), 2 * lr_ldq(P), P.Z0, P.lam, dh2,
                         ptr<int>(ctx->lrFlag), ptr<int>(ctx->lrPerm), dL, ldL, P.stat);
}
}  // namespace

// Conditioning guard (kernels_dyn.hip): traits whose weighted null design is nearly collinear (several covariates, h2 -> 1)
// get their LOD columns recomputed with an orthogonalised projection; a no-op for c = 1.  Runs on the current stream after
// the scan kernels that wrote dL.
int illcond_rescan(blmm_ctx* ctx, const Pipe& P, const NullModel& nm, int64_t m, const double* dh2, double* dL, int64_t ldL) {
  if (P.c < 2 || m <= 0 || P.p <= 0) return BLMM_OK;
  int rc = ensure(ctx, ctx->illList, sizeof(int) * (size_t)m);
  if (rc) return rc;
  if ((rc = launch_illcond_flag(ctx, nm, m, P.Z0, P.lam, dh2, ptr<int>(ctx->illList), P.stat))) return rc;
  if (ctx->red_cur.pmax) return BLMM_OK;   // (as lr_fix)
  return launch_scan_qr(ctx, nm, P.Yt, P.ldy, P.Xt, P.ldx, P.p, P.Z0, P.lam, dh2, ptr<int>(ctx->illList), dL, ldL, P.stat);
}

// every trait's h2 is final: one region
int lr_finish(blmm_ctx* ctx, const Pipe& P, const NullModel& nm, const double* dh2, double* dL, int64_t ldL, Timer& tm) {
  int rc;
  const LrRegion rg = lr_region(P, 0);
  const LrSeg seg = lr_segments(ctx, P.n);
  hipStream_t main_stream = ctx->stream;
  BLMM_HIP(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
  if ((rc = launch_lr_classify(ctx, P.n, P.m, lr_shared_tol(ctx), P.lam, dh2, nullptr, nullptr, nullptr, ptr<int>(ctx->lrPerm), rg, seg))) return rc;
  if ((rc = lr_region_panels(ctx, P, nm, dh2, rg))) return rc;
  tm.mark();
  // residual guard of the weight basis, every trait: side stream, beside the scan kernel; joined below
  BLMM_HIP(hipEventRecord(ctx->ev_fork, main_stream));
  BLMM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
  ctx->stream = ctx->side;
  rc = lr_region_resid(ctx, P, nm, dh2, rg);
  ctx->stream = main_stream;
  if (rc) return rc;
  BLMM_HIP(hipEventRecord(ctx->ev_join, ctx->side));
  if ((rc = lr_region_scan(ctx, P, rg, dL, ldL))) return rc;
  BLMM_HIP(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
  if ((rc = lr_fix(ctx, P, nm, dh2, dL, ldL))) return rc;
  if ((rc = illcond_rescan(ctx, P, nm, P.m, dh2, dL, ldL))) return rc;
  tm.mark();
  return BLMM_OK;
}

// The h2 search was split (launch_brent phase 1 came back with sp.active): the traits k_brent finished (fin[j] == 1) form
// region 0 and are scanned at once; k_brent2, the classification and the panels of its traits (region 1) run on a
// second side stream beside that scan, and region 1 is scanned after it.
int lr_finish_split(blmm_ctx* ctx, const Pipe& P, const NullModel& nm, double* dh2, double* dL, int64_t ldL, Timer& tm,
                    const BrentSplit& sp) {
  int rc;
  const LrRegion r0 = lr_region(P, 0), r1 = lr_region(P, 1);
  const LrSeg seg = lr_segments(ctx, P.n);
  hipStream_t main_stream = ctx->stream;
  // ---- main stream: region 0's classification and panels (the weight basis is ready at ev_q, the marker-side products at ev_join) ...
  if ((rc = launch_lr_classify(ctx, P.n, P.m, lr_shared_tol(ctx), P.lam, dh2, sp.fin, nullptr, nullptr, ptr<int>(ctx->lrPerm), r0, seg))) return rc;
  // ONE wait: ev_join is recorded on the side stream behind ev_q (the basis) and behind the marker-side products, which are done
  // before the panels kernel can start anyway (profiles/r03_timeline_bxd_step.txt: 410 us against 425) -- every barrier packet
  // between two dependent kernels of this stream costs ~10 us
  BLMM_HIP(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
  ctx->stop_event_used = false;
  ctx->stop_event_next = ctx->ev_b1;                                 // recorded by the panels kernel itself
  rc = lr_region_panels(ctx, P, nm, dh2, r0);
  const bool b1_recorded = ctx->stop_event_used;
  ctx->stop_event_next = nullptr; ctx->stop_event_used = false;
  if (rc) return rc;
  // ---- ... and only then the second side stream: the rest of the h2 search, then region 1's columns.  Forked right behind
  //      k_brent (rounds 2-3a) k_brent2's older waves won the issue arbitration against the 16-lane panels kernel on the
  //      critical path (45 us beside it, 28 alone); its chain has ~0.5 ms of slack before region 1's scan needs it
  if (!b1_recorded) BLMM_HIP(hipEventRecord(ctx->ev_b1, main_stream));
  BLMM_HIP(hipStreamWaitEvent(ctx->side2, ctx->ev_b1, 0));
  ctx->stream = ctx->side2;
  BrentSplit sp2 = sp;
  rc = launch_brent(ctx, nm, P.Yt, P.ldy, P.m, P.Z0, P.lam, dh2, nullptr, nullptr, P.stat, 2, &sp2);
  if (!rc && hipStreamWaitEvent(ctx->side2, ctx->ev_q, 0) != hipSuccess) rc = fail(ctx, BLMM_ERR_HIP, "hipStreamWaitEvent failed");
  if (!rc) rc = launch_lr_classify(ctx, P.n, P.m, lr_shared_tol(ctx), P.lam, dh2, nullptr, sp.list, sp.cnt, ptr<int>(ctx->lrPerm), r1, seg);
  if (!rc) rc = lr_region_panels(ctx, P, nm, dh2, r1);
  ctx->stream = main_stream;
  if (rc) return rc;
  BLMM_HIP(hipEventRecord(ctx->ev_b2, ctx->side2));
  tm.mark();
  BLMM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_b1, 0));            // region 0's panels are done (the same point the second side stream forks at)
  ctx->stream = ctx->side;
  rc = lr_region_resid(ctx, P, nm, dh2, r0);
  ctx->stream = main_stream;
  if (rc) return rc;
  if ((rc = lr_region_scan(ctx, P, r0, dL, ldL))) return rc;
  // ---- region 1: its guard on the first side stream (behind region 0's), its scan on the main stream
  BLMM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_b2, 0));
  ctx->stream = ctx->side;
  rc = lr_region_resid(ctx, P, nm, dh2, r1);
  ctx->stream = main_stream;
  if (rc) return rc;
  BLMM_HIP(hipEventRecord(ctx->ev_join, ctx->side));
  BLMM_HIP(hipStreamWaitEvent(main_stream, ctx->ev_b2, 0));
  if ((rc = lr_region_scan(ctx, P, r1, dL, ldL))) return rc;
  BLMM_HIP(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
  if ((rc = lr_fix(ctx, P, nm, dh2, dL, ldL))) return rc;
  if ((rc = illcond_rescan(ctx, P, nm, P.m, dh2, dL, ldL))) return rc;
  tm.mark();
  return BLMM_OK;
}

}  // namespace

extern "C" {

int blmm_version(void) { return BLMM_VERSION; }

int blmm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* blmm_err_string(int code) {
  switch (code) {
    case BLMM_OK: return "ok";
    case BLMM_ERR_INVALID: return "invalid argument";
    case BLMM_ERR_DIM: return "Dimension mismatch.";
    case BLMM_ERR_H2_ONE: return "Heritability of 1 is not allowed.";
    case BLMM_ERR_DECOMP: return "Please choose either `eigen` or `svd` for decomposition of the kinship matrix.";
    case BLMM_ERR_METHOD: return "unknown bulkscan method";
    case BLMM_ERR_ONE_TRAIT: return "Can only handle one trait.";
    case BLMM_ERR_NO_INTERCEPT: return "Intercept has to be added when no other covariate is given.";
    case BLMM_ERR_ZERO_NORM: return "Dividing by zeros: the input vector can not contain any zeros!";
    case BLMM_ERR_NPERMS: return "The required number of permutations must be a positive integer.";
    case BLMM_ERR_UNSUPPORTED: return "unsupported configuration";
    case BLMM_ERR_NO_DEVICE: return "no usable HIP device";
    case BLMM_ERR_HIP: return "HIP runtime error";
    case BLMM_ERR_ALLOC: return "device allocation failed";
  }
  return "unknown error";
}

int blmm_create(int device_id, void* hip_stream, blmm_ctx** out) {
  if (!out) return BLMM_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return BLMM_ERR_NO_DEVICE;
  if (hipSetDevice(device_id) != hipSuccess) return BLMM_ERR_NO_DEVICE;
  blmm_ctx* ctx = new blmm_ctx();
  ctx->device = device_id;
  if (hip_stream == BLMM_STREAM_NULL) {
    ctx->stream = nullptr;   // the legacy default stream (handle 0)
  } else if (hip_stream) {
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return BLMM_ERR_HIP; }
    ctx->own_stream = true;
  }
  if (hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_xt, hipEventDisableTiming) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_b1, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_b2, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_q, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_m, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_wb, hipEventDisableTiming) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_inY, hipEventDisableTiming) != hipSuccess) {
    blmm_destroy(ctx);
    return BLMM_ERR_HIP;
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess) ctx->num_cus = cus;
    void* hp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess) { blmm_destroy(ctx); return BLMM_ERR_HIP; }
    std::memset(hp, 0, 64);
    ctx->hflag = reinterpret_cast<volatile int64_t*>(hp);
  }
  if (ensure(ctx, ctx->logtab, sizeof(blmm_log_table_host)) != BLMM_OK ||
      hipMemcpy(ctx->logtab.p, blmm_log_table_host, sizeof(blmm_log_table_host), hipMemcpyHostToDevice) != hipSuccess ||
      ensure(ctx, ctx->lodtab, sizeof(blmm_lod_table_host)) != BLMM_OK ||
      hipMemcpy(ctx->lodtab.p, blmm_lod_table_host, sizeof(blmm_lod_table_host), hipMemcpyHostToDevice) != hipSuccess ||
      ensure(ctx, ctx->pvtab, sizeof(blmm_pv_table_host)) != BLMM_OK ||
      hipMemcpy(ctx->pvtab.p, blmm_pv_table_host, sizeof(blmm_pv_table_host), hipMemcpyHostToDevice) != hipSuccess) {
    blmm_destroy(ctx);
    return BLMM_ERR_HIP;
  }
  *out = ctx;
  return BLMM_OK;
}

void blmm_destroy(blmm_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  DevBuf* bufs[] = {&ctx->Ks, &ctx->V, &ctx->lam, &ctx->U, &ctx->Zs, &ctx->Z0, &ctx->Rp, &ctx->Yt, &ctx->Xt, &ctx->panels,
                    &ctx->iyy, &ctx->h2, &ctx->h2idx, &ctx->sig2, &ctx->ell, &ctx->isx, &ctx->stat, &ctx->gridd, &ctx->misc,
                    &ctx->EllTab, &ctx->inY, &ctx->inG, &ctx->inK, &ctx->inCov, &ctx->inW, &ctx->outL, &ctx->outH2,
                    &ctx->tmpA, &ctx->tmpB, &ctx->tmpC, &ctx->perm, &ctx->r0, &ctx->altbuf, &ctx->logtab, &ctx->lraw,
                    &ctx->wbQ, &ctx->wbW, &ctx->wbRk, &ctx->lrT, &ctx->lrC, &ctx->lrL, &ctx->lrFlag, &ctx->lrPart, &ctx->lrPerm, &ctx->lrDen0, &ctx->eigW, &ctx->xf32, &ctx->pf32, &ctx->brSt, &ctx->brList, &ctx->illList, &ctx->qrSlab, &ctx->lodtab, &ctx->dynFac, &ctx->pvtab, &ctx->outP, &ctx->redbuf, &ctx->redtrip, &ctx->altC, &ctx->rf32, &ctx->btG};
  for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
  for (auto& s : ctx->evsets) for (auto& e : s.e) (void)hipEventDestroy(e);
  if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_xt) (void)hipEventDestroy(ctx->ev_xt);
  if (ctx->ev_b1) (void)hipEventDestroy(ctx->ev_b1);
  if (ctx->ev_b2) (void)hipEventDestroy(ctx->ev_b2);
  if (ctx->ev_q) (void)hipEventDestroy(ctx->ev_q);
  if (ctx->ev_m) (void)hipEventDestroy(ctx->ev_m);
  if (ctx->ev_wb) (void)hipEventDestroy(ctx->ev_wb);
  if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
  if (ctx->ev_inY) (void)hipEventDestroy(ctx->ev_inY);
  if (ctx->copy) { (void)hipStreamSynchronize(ctx->copy); (void)hipStreamDestroy(ctx->copy); }
  if (ctx->side2) { (void)hipStreamSynchronize(ctx->side2); (void)hipStreamDestroy(ctx->side2); }
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  if (ctx->hflag) (void)hipHostFree(const_cast<int64_t*>(ctx->hflag));
  destroy_host_stage(ctx->hstage);
  delete ctx;
}

const char* blmm_last_error(const blmm_ctx* ctx) { return ctx ? ctx->err.c_str() : "ctx is NULL"; }

int blmm_set_stream(blmm_ctx* ctx, void* hip_stream) {
  if (!ctx) return BLMM_ERR_INVALID;
  hipStreamSynchronize(ctx->stream);
  if (ctx->own_stream) { hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
  if (hip_stream == BLMM_STREAM_NULL) ctx->stream = nullptr;
  else if (hip_stream) ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return fail(ctx, BLMM_ERR_HIP, "hipStreamCreate");
    ctx->own_stream = true;
  }
  return BLMM_OK;
}

int blmm_set_timing(blmm_ctx* ctx, int on) {
  if (!ctx) return BLMM_ERR_INVALID;
  ctx->timing = on != 0;
  return BLMM_OK;
}

int blmm_read_timings(blmm_ctx* ctx, double* sums_ms, int64_t* ncalls) {
  if (!ctx || !sums_ms || !ncalls) return BLMM_ERR_INVALID;
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 6; ++i) sums_ms[i] = 0.0;
  for (size_t k = 0; k < ctx->ev_used; ++k) {
    double t[6];
    phase_times(ctx->evsets[k], t);
    for (int i = 0; i < 6; ++i) sums_ms[i] += t[i];
  }
  *ncalls = (int64_t)ctx->ev_used;
  ctx->ev_used = 0;
  return BLMM_OK;
}

int blmm_lowrank_profile(blmm_ctx* ctx, int64_t* out) {
  if (!ctx || !out) return BLMM_ERR_INVALID;
  for (int i = 0; i < 2 + 2 * LR_SEG_MAX; ++i) out[i] = 0;
  if (!ctx->stat.p || !ctx->wbRk.p) return BLMM_OK;           // no null-exact call yet
  BLMM_HIP(hipSetDevice(ctx->device));
  int64_t h[NSTAT];
  int rk[4 * LR_SEG_MAX];
  BLMM_HIP(hipMemcpyAsync(h, ctx->stat.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(rk, ctx->wbRk.p, sizeof(rk), hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  // the segments in use: those whose rank the basis kernel wrote (a single basis: segment 0 only, the other counts are zero)
  int S = 0;
  for (int s = 0; s < LR_SEG_MAX; ++s) {
    const int64_t cnt = h[24 + s] + h[44 + s];
    if (cnt > 0) { S = s + 1; out[2 + 2 * s] = cnt; out[3 + 2 * s] = rk[4 * s]; }
  }
  out[0] = S; out[1] = h[12] + h[14];
  return BLMM_OK;
}

// Diagnostic for tests that report WHERE in the data-dependent panel layout a trait sat (k_lr_classify: two regions split by the
// h2 search's hand-over, in each the shared-weights class from the front and the weight-basis segments from the back): col_out[j] =
// panel column of trait j in the last null-exact call (-1: none), *region_width = columns per region (region = col / width),
// counts_out[4] = {shared-weights traits, columns of the other class} of region 0, then of region 1.
int blmm_lowrank_columns(blmm_ctx* ctx, int64_t m, int32_t* col_out, int64_t* region_width, int64_t* counts_out) {
  if (!ctx || !col_out || m < 0) return BLMM_ERR_INVALID;
  for (int64_t j = 0; j < m; ++j) col_out[j] = -1;
  if (region_width) *region_width = ctx->lr_last_ldq;
  if (counts_out) for (int i = 0; i < 4; ++i) counts_out[i] = 0;
  if (!ctx->lrPerm.p || !ctx->stat.p || ctx->lr_last_ldq <= 0) return BLMM_OK;
  BLMM_HIP(hipSetDevice(ctx->device));
  const int64_t ldq = ctx->lr_last_ldq;
  std::vector<int> perm((size_t)(2 * ldq));
  int64_t h[NSTAT];
  BLMM_HIP(hipMemcpyAsync(perm.data(), ctx->lrPerm.p, sizeof(int) * perm.size(), hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(h, ctx->stat.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  for (int r = 0; r < 2; ++r) {
    const int64_t nsh = h[12 + 2 * r], noth = h[13 + 2 * r];
    if (counts_out) { counts_out[2 * r] = nsh; counts_out[2 * r + 1] = noth; }
    for (int64_t cidx = 0; cidx < ldq; ++cidx) {
      if (!(cidx < nsh || cidx >= ldq - noth)) continue;
      const int t = perm[(size_t)(r * ldq + cidx)];
      if (t >= 0 && t < m) col_out[t] = (int32_t)(r * ldq + cidx);
    }
  }
  return BLMM_OK;
}

int blmm_synchronize(blmm_ctx* ctx) {
  if (!ctx) return BLMM_ERR_INVALID;
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return check_sticky(ctx);
}

void blmm_default_opts(blmm_opts* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->method = BLMM_NULL_GRID; o->reml = 0; o->add_intercept = 1; o->decomp_scheme = BLMM_EIGEN;
  o->optim_interval = 1; o->compat_flags = 0; o->prior_variance = 1.0; o->prior_sample_size = 0.0;
}

// What selects another ARITHMETIC path is a property of the context, not of the caller's environment (blmm_internal.h: Tuning).
static const struct { const char* key; int kind; size_t off; double lo, hi; } kTune[] = {
  {"lr_tol", 0, offsetof(blmm::Tuning, lr_tol), 0.0, 1.0},
  {"illcond_rho", 0, offsetof(blmm::Tuning, illcond_rho), 0.0, 1e300},
  {"exact_full_rank", 1, offsetof(blmm::Tuning, exact_full_rank), 0, 1},
  {"pval_libm", 1, offsetof(blmm::Tuning, pval_libm), 0, 1},
  {"pval_fused", 1, offsetof(blmm::Tuning, pval_fused), 0, 1},
  {"lr_segments", 1, offsetof(blmm::Tuning, lr_segments), 0, LR_SEG_MAX},
  {"lr_shared", 1, offsetof(blmm::Tuning, lr_shared), 0, 1},
  {"lr_split", 1, offsetof(blmm::Tuning, lr_split), -1, 1},
  {"eigen_solver", 1, offsetof(blmm::Tuning, eigen_solver), 0, 2},
  {"f32_rotation", 1, offsetof(blmm::Tuning, f32_rotation), 0, 1},
};
int blmm_set_tuning(blmm_ctx* ctx, const char* key, double value) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!key) return fail(ctx, BLMM_ERR_INVALID, "set_tuning: key is NULL");
  if (std::strcmp(key, "defaults") == 0) { ctx->tune = blmm::Tuning(); return BLMM_OK; }
  for (const auto& t : kTune)
    if (std::strcmp(key, t.key) == 0) {
      if (!(value >= t.lo && value <= t.hi) || (t.kind == 1 && value != std::floor(value)))
        return fail(ctx, BLMM_ERR_INVALID, std::string("set_tuning: value out of range for ") + key);
      char* base = reinterpret_cast<char*>(&ctx->tune) + t.off;
      if (t.kind == 0) *reinterpret_cast<double*>(base) = value; else *reinterpret_cast<int*>(base) = (int)value;
      return BLMM_OK;
    }
  return fail(ctx, BLMM_ERR_INVALID, std::string("set_tuning: unknown key ") + key);
}
int blmm_get_tuning(const blmm_ctx* ctx, const char* key, double* value) {
  if (!ctx || !key || !value) return BLMM_ERR_INVALID;
  for (const auto& t : kTune)
    if (std::strcmp(key, t.key) == 0) {
      const char* base = reinterpret_cast<const char*>(&ctx->tune) + t.off;
      *value = t.kind == 0 ? *reinterpret_cast<const double*>(base) : (double)*reinterpret_cast<const int*>(base);
      return BLMM_OK;
    }
  return BLMM_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------------
int blmm_kinship_dev(blmm_ctx* ctx, const double* dG, int64_t n, int64_t p, double* dK_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!dG || !dK_out || n < 1 || p < 1) return fail(ctx, BLMM_ERR_INVALID, "calcKinship: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc = ensure(ctx, ctx->tmpA, sizeof(double) * (size_t)64 * n * n);
  if (rc) return rc;
  return launch_kinship(ctx, dG, n, p, dK_out, ptr<double>(ctx->tmpA));
}

int blmm_kinship(blmm_ctx* ctx, const double* G, int64_t n, int64_t p, double* K_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!G || !K_out || n < 1 || p < 1) return fail(ctx, BLMM_ERR_INVALID, "calcKinship: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->inG, sizeof(double) * n * p))) return rc;
  if ((rc = ensure(ctx, ctx->inK, sizeof(double) * n * n))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->inG.p, G, sizeof(double) * n * p, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = blmm_kinship_dev(ctx, ptr<double>(ctx->inG), n, p, ptr<double>(ctx->inK)))) return rc;
  BLMM_HIP(hipMemcpyAsync(K_out, ctx->inK.p, sizeof(double) * n * n, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return BLMM_OK;
}

__global__ void k_round_digits(double* __restrict__ v, int64_t cnt, double scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cnt) v[i] = rint(v[i] * scale) / scale;
}

int blmm_kinship_rounded(blmm_ctx* ctx, const double* G, int64_t n, int64_t p, int64_t digits, double* K_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!G || !K_out || n < 1 || p < 1 || digits > 300) return fail(ctx, BLMM_ERR_INVALID, "calcKinship: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->inG, sizeof(double) * n * p))) return rc;
  if ((rc = ensure(ctx, ctx->inK, sizeof(double) * n * n))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->inG.p, G, sizeof(double) * n * p, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = blmm_kinship_dev(ctx, ptr<double>(ctx->inG), n, p, ptr<double>(ctx->inK)))) return rc;
  if (digits >= 0 && digits <= 17)   // beyond 17 digits rounding a double changes nothing (and 10^digits would overflow)
    hipLaunchKernelGGL(k_round_digits, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, ctx->stream, ptr<double>(ctx->inK), n * n, std::pow(10.0, (double)digits));
  BLMM_HIP(hipMemcpyAsync(K_out, ctx->inK.p, sizeof(double) * n * n, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return BLMM_OK;
}

// ---------------------------------------------------------------------------------------------------
int blmm_lod_colmax_dev(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, double* dmax_out, int64_t* dargmax_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!dL || !dmax_out || p < 1 || m < 0 || ldL < p) return fail(ctx, BLMM_ERR_INVALID, "lod_colmax: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  return launch_colmax(ctx, dL, p, m, ldL, dmax_out, dargmax_out);
}

int blmm_lod_colmax(blmm_ctx* ctx, const double* L, int64_t p, int64_t m, double* max_out, int64_t* argmax_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!L || !max_out || p < 1 || m < 0) return fail(ctx, BLMM_ERR_INVALID, "lod_colmax: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * (size_t)p * (m > 0 ? m : 1)))) return rc;
  if ((rc = ensure(ctx, ctx->tmpA, sizeof(double) * (size_t)(m > 0 ? m : 1)))) return rc;
  if ((rc = ensure(ctx, ctx->tmpB, sizeof(int64_t) * (size_t)(m > 0 ? m : 1)))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->outL.p, L, sizeof(double) * (size_t)p * m, hipMemcpyHostToDevice, ctx->stream));
  if (m > 0) { ctx->last_L = ptr<double>(ctx->outL); ctx->last_p = p; ctx->last_m = m; ctx->last_f32 = false; }
  if ((rc = launch_colmax(ctx, ptr<double>(ctx->outL), p, m, p, ptr<double>(ctx->tmpA), ptr<int64_t>(ctx->tmpB)))) return rc;
  BLMM_HIP(hipMemcpyAsync(max_out, ctx->tmpA.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
  if (argmax_out) BLMM_HIP(hipMemcpyAsync(argmax_out, ctx->tmpB.p, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return BLMM_OK;
}

// host-pointer forms of the consumers (upload, reduce on the device, download the small result)
int blmm_lod2log10p(blmm_ctx* ctx, const double* L, int64_t p, int64_t m, int64_t chisq_df, double* P_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!L || !P_out || p < 0 || m < 0 || chisq_df < 1) return fail(ctx, BLMM_ERR_INVALID, "lod2log10p: bad arguments");
  if ((size_t)p * m == 0) return BLMM_OK;
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->altbuf, sizeof(double) * (size_t)p * m * 2))) return rc;
  double* dL = ptr<double>(ctx->altbuf);
  double* dP = dL + (size_t)p * m;
  BLMM_HIP(hipMemcpyAsync(dL, L, sizeof(double) * (size_t)p * m, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = launch_lod2log10p(ctx, dL, p, m, p, (int)chisq_df, dP, p))) return rc;
  if ((rc = copy_to_host(ctx, P_out, dP, sizeof(double) * (size_t)p * m))) return rc;
  return BLMM_OK;
}

int blmm_lod_threshold(blmm_ctx* ctx, const double* L, int64_t p, int64_t m, double thr, int64_t cap, int32_t* i_out,
                       int32_t* j_out, double* lod_out, int64_t* count_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!L || p < 1 || m < 1) return fail(ctx, BLMM_ERR_INVALID, "lod_threshold: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * (size_t)p * m))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->outL.p, L, sizeof(double) * (size_t)p * m, hipMemcpyHostToDevice, ctx->stream));
  ctx->last_L = ptr<double>(ctx->outL); ctx->last_p = p; ctx->last_m = m; ctx->last_f32 = false;
  return blmm_last_lod_threshold(ctx, thr, cap, i_out, j_out, lod_out, count_out);
}

int blmm_get_thresholds(blmm_ctx* ctx, const double* Lperms, int64_t p, int64_t nperms, const double* probs, int64_t nprobs,
                        double* thrs_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!Lperms || p < 1 || nperms < 1) return fail(ctx, BLMM_ERR_INVALID, "get_thresholds: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->altbuf, sizeof(double) * (size_t)p * nperms))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->altbuf.p, Lperms, sizeof(double) * (size_t)p * nperms, hipMemcpyHostToDevice, ctx->stream));
  return blmm_get_thresholds_dev(ctx, ptr<double>(ctx->altbuf), p, nperms, p, probs, nprobs, thrs_out);
}

// ---------------------------------------------------------------------------------------------------
// Everything behind the rotations: the h2 search / grid log-likelihoods, the panels and the LOD kernels of one method, then the
// status.  Shared by blmm_bulkscan_dev and blmm_bulkscan_prerotated_dev.
// `output_pvals` inside the scan (blmm_set_log10p_output): resolves the armed request for THIS call.  One degree of freedom and a
// null-* method: the scan kernels write -log10 p from their epilogues (ScanArgs::Pv, set through ctx->pv_cur while they run);
// otherwise (alt-grid: the LOD is only final after the last grid point; other degrees of freedom: incomplete gamma function)
// the column pass of kernels_post.hip runs over the finished L, still inside the call.
extern "C++" {
namespace {
// The request of blmm_set_log10p_output, TAKEN out of the context by the first statement of every bulkscan entry point: whatever
// that call then does -- fail its argument checks, fail in prepare, succeed -- the context is disarmed, so a later unrelated call
// can never write to a stale pointer.
struct PvReq { bool armed = false; double* out = nullptr; int64_t ld = 0, df = 1; };
PvReq pv_take(blmm_ctx* ctx) {
  PvReq r; r.armed = ctx->pv_armed; r.out = ctx->pv_out; r.ld = ctx->pv_ld; r.df = ctx->pv_df;
  ctx->pv_armed = false; ctx->pv_out = nullptr; ctx->pv_ld = 0; ctx->pv_df = 1;
  return r;
}
void pv_hand_over(blmm_ctx* ctx, const PvReq& r) {   // host-pointer entry point -> the *_dev call it makes next
  ctx->pv_armed = r.armed; ctx->pv_out = r.out; ctx->pv_ld = r.ld; ctx->pv_df = r.df;
}
struct PvCall {
  blmm_ctx* ctx; PvReq req; double* P = nullptr; int64_t ld = 0, df = 1; bool fused = false, owned = false;
  PvCall(blmm_ctx* c, const PvReq& r) : ctx(c), req(r) {}
  ~PvCall() { ctx->pv_cur = nullptr; ctx->pv_cur_ld = 0; }
  int begin(const Pipe& Pp, const blmm_opts* o) {
    if (!req.armed) return BLMM_OK;
    if (Pp.p <= 0 || Pp.m <= 0) return BLMM_OK;
    df = req.df; P = req.out; ld = req.ld;
    if (P && ld < Pp.p) { P = nullptr; return fail(ctx, BLMM_ERR_INVALID, "log10p output: ldP < p"); }
    if (!P) {
      int rc = ensure(ctx, ctx->outP, sizeof(double) * (size_t)Pp.p * (size_t)Pp.m);
      if (rc) return rc;
      P = ptr<double>(ctx->outP); ld = Pp.p; owned = true;
    }
    const char* pf = dev_env("BLMM_PVAL_FUSED");             // tuning key "pval_fused"
    fused = df == 1 && o->method != BLMM_ALT_GRID && (pf ? pf[0] != '0' : ctx->tune.pval_fused != 0);
    if (fused) { ctx->pv_cur = P; ctx->pv_cur_ld = ld; }
    return BLMM_OK;
  }
  int finish(const Pipe& Pp, const double* dL, int64_t ldL) {
    ctx->pv_cur = nullptr; ctx->pv_cur_ld = 0;
    if (!P) return BLMM_OK;
    if (!fused) { int rc = launch_lod2log10p(ctx, dL, Pp.p, Pp.m, ldL, (int)df, P, ld); if (rc) return rc; }
    if (owned) { ctx->last_P = P; ctx->last_P_ld = ld; ctx->last_P_df = df; }
    return BLMM_OK;
  }
};
}  // namespace
}  // extern "C++"

// marker norms of every grid point; on the side stream behind the marker rotation when prepare() put that there (P.xt_side), then
// joined into the main stream: the scan that follows needs both
static int isx_maybe_side(blmm_ctx* ctx, const Pipe& P, const NullModel& nm, const double* dgrid, int ngrid) {
  int rc;
  if ((rc = ensure(ctx, ctx->isx, sizeof(double) * (size_t)ngrid * P.ldx))) return rc;
  hipStream_t main_stream = ctx->stream;
  if (P.xt_side) ctx->stream = ctx->side;
  rc = launch_isx(ctx, nm, P.Xt, P.ldx, P.p, P.Z0, P.lam, dgrid, ngrid, ptr<double>(ctx->isx), P.ldx, P.stat);
  ctx->stream = main_stream;
  if (rc) return rc;
  if (P.xt_side) {
    BLMM_HIP(hipEventRecord(ctx->ev_join, ctx->side));
    BLMM_HIP(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
  }
  return BLMM_OK;
}

static int scan_pipeline(blmm_ctx* ctx, const blmm_opts* opts, Pipe& P, Timer& tm, bool lowrank, bool wbasis_started, double* dgrid,
                         const double* h2_grid_host, int64_t ngrid, double* dL_out, int64_t ldL, double* dh2_out, blmm_status* status,
                         const PvReq& pvreq) {
  int rc;
  const int64_t m = P.m, p = P.p;
  const NullModel nm = null_model(P, opts);
  const int64_t ldp = P.ldy;
  PvCall pvc(ctx, pvreq);
  if ((rc = pvc.begin(P, opts))) return rc;
  if (m == 0) { tm.mark(); tm.mark(); tm.mark(); return end_call(ctx, P, status, &tm); }
  if (p == 0) {
    // no markers: only the per-trait null model (h2_null_list does not depend on G); alt-grid's h2_panel is p x m = empty
    if (opts->method == BLMM_NULL_EXACT) {
      if ((rc = launch_brent(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dh2_out, nullptr, nullptr, P.stat))) return rc;
    } else if (opts->method == BLMM_NULL_GRID) {
      if ((rc = ensure(ctx, ctx->h2idx, sizeof(int) * (size_t)m))) return rc;
      if ((rc = launch_loglik_grid(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dgrid, (int)ngrid, nullptr, ptr<int>(ctx->h2idx), dh2_out, P.stat))) return rc;
    }
    tm.mark(); tm.mark(); tm.mark();
    return end_call(ctx, P, status, &tm);
  }

  if (opts->method == BLMM_NULL_EXACT) {
    if (lowrank) {
      // the basis (started in prepare) and the marker-side products (Q, Xt) run on the side stream beside the
      // per-trait Brent search
      if ((rc = lr_begin(ctx, P, wbasis_started))) return rc;
      // the second kernel of a split h2 search runs beside the scan of the traits the first one finished (BLMM_LR_SPLIT=0: A/B)
      // ... when there are enough traits for two regions (BLMM_LR_SPLIT unset): below ~8 k each region's scan is a few dispatch rounds
      // with their ramps, and one region wins (m = 4445, a rank's share of the BXD problem on 8 GPUs: 0.651 against 0.677 ms per
      // step; m = 8889: 0.777 against 0.776; m = 35554: the split is worth 2.7 %).  BLMM_LR_SPLIT=1: always
      const char* split_env = dev_env("BLMM_LR_SPLIT");                // tuning key "lr_split" (tests hold the two forms against each other)
      const bool split_on = split_env ? split_env[0] != '0' : (ctx->tune.lr_split < 0 ? m >= 8192 : ctx->tune.lr_split != 0);
      BrentSplit sp;
      if ((rc = launch_brent(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dh2_out, nullptr, nullptr, P.stat, split_on ? 1 : 0, &sp))) return rc;
      tm.mark();
      if (sp.active) { if ((rc = lr_finish_split(ctx, P, nm, dh2_out, dL_out, ldL, tm, sp))) return rc; }
      else if ((rc = lr_finish(ctx, P, nm, dh2_out, dL_out, ldL, tm))) return rc;
    } else {
      if ((rc = launch_brent(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dh2_out, nullptr, nullptr, P.stat))) return rc;
      tm.mark();
      if ((rc = ensure(ctx, ctx->panels, sizeof(double) * (size_t)(2 + P.c) * P.npad * ldp))) return rc;
      if ((rc = launch_panels(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dh2_out, 1, ptr<double>(ctx->panels), ldp, P.stat))) return rc;
      tm.mark();
      ScanArgs a = scan_args(ctx, P, ptr<double>(ctx->panels), ldp, dL_out, ldL, m);
      if ((rc = launch_scan_exact(ctx, a, P.c))) return rc;
      if ((rc = illcond_rescan(ctx, P, nm, m, dh2_out, dL_out, ldL))) return rc;
      tm.mark();
    }
    if (opts->compat_flags & BLMM_FLAG_H2_AUDIT) {
      // opt-in diagnostic: the profile log-likelihood of every trait on the grid 0, 1/16, .., 15/16 -> n_h2_multimodal
      double gridh[16];
      for (int g = 0; g < 16; ++g) gridh[g] = g / 16.0;
      double* dg = nullptr;
      if ((rc = grid_to_device(ctx, gridh, 16, &dg))) return rc;
      if ((rc = ensure(ctx, ctx->EllTab, sizeof(double) * (size_t)16 * m))) return rc;
      if ((rc = launch_loglik_grid(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dg, 16, ptr<double>(ctx->EllTab), nullptr, nullptr, P.stat))) return rc;
      if ((rc = launch_h2_audit(ctx, ptr<double>(ctx->EllTab), 16, m, P.stat))) return rc;
      ctx->audit_ran = true;
    }
  } else if (opts->method == BLMM_NULL_GRID) {
    if ((rc = ensure(ctx, ctx->h2idx, sizeof(int) * (size_t)m))) return rc;
    if ((rc = launch_loglik_grid(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dgrid, (int)ngrid, nullptr, ptr<int>(ctx->h2idx), dh2_out, P.stat))) return rc;
    tm.mark();
    if ((rc = ensure(ctx, ctx->panels, sizeof(double) * (size_t)P.npad * ldp))) return rc;
    if ((rc = launch_panels(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dh2_out, 0, ptr<double>(ctx->panels), ldp, P.stat))) return rc;
    if ((rc = isx_maybe_side(ctx, P, nm, dgrid, (int)ngrid))) return rc;
    tm.mark();
    ScanArgs a = scan_args(ctx, P, ptr<double>(ctx->panels), ldp, dL_out, ldL, m);
    a.isx = ptr<double>(ctx->isx); a.ld_isx = P.ldx; a.bin = ptr<int>(ctx->h2idx);
    if ((rc = launch_scan_table(ctx, a))) return rc;
    tm.mark();
  } else {  // alt-grid
    if ((rc = ensure(ctx, ctx->EllTab, sizeof(double) * (size_t)ngrid * m))) return rc;
    if ((rc = launch_loglik_grid(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dgrid, (int)ngrid, ptr<double>(ctx->EllTab), nullptr, nullptr, P.stat))) return rc;
    tm.mark();
    if ((rc = ensure(ctx, ctx->panels, sizeof(double) * (size_t)ngrid * P.npad * ldp))) return rc;
    if ((rc = ensure(ctx, ctx->h2, sizeof(double) * (size_t)m))) return rc;
    if (nm.c <= CTPL) {
      // every grid point's panel in one launch (grid point on blockIdx.y)
      if ((rc = launch_panels(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, nullptr, 0, ptr<double>(ctx->panels), ldp, P.stat, dgrid, (int)ngrid))) return rc;
    } else {
      for (int64_t g = 0; g < ngrid; ++g) {
        if ((rc = fill(ctx, ptr<double>(ctx->h2), m, h2_grid_host[g]))) return rc;
        if ((rc = launch_panels(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, ptr<double>(ctx->h2), 0,
                                ptr<double>(ctx->panels) + (size_t)g * P.npad * ldp, ldp, P.stat))) return rc;
      }
    }
    if ((rc = isx_maybe_side(ctx, P, nm, dgrid, (int)ngrid))) return rc;
    if ((rc = ensure(ctx, ctx->altC, sizeof(double) * (size_t)ngrid * m))) return rc;
    if ((rc = launch_alt_ctab(ctx, ptr<double>(ctx->EllTab), (int)ngrid, m, P.n, ptr<double>(ctx->altC)))) return rc;
    tm.mark();
    AltArgs aa;
    aa.s = scan_args(ctx, P, ptr<double>(ctx->panels), ldp, dL_out, ldL, m);
    aa.s.isx = ptr<double>(ctx->isx); aa.s.ld_isx = P.ldx;
    aa.ngrid = (int)ngrid; aa.EllTab = ptr<double>(ctx->EllTab); aa.grid_dev = dgrid; aa.H2 = dh2_out; aa.ldH = p;
    aa.Ctab = ptr<double>(ctx->altC);
    aa.counter_quirk = (opts->compat_flags & BLMM_COMPAT_ALT_COUNTER) ? 1 : 0;
    if ((rc = launch_scan_alt(ctx, aa))) return rc;
    tm.mark();
  }
  if ((rc = pvc.finish(P, dL_out, ldL))) return rc;
  return end_call(ctx, P, status, &tm);
}

int blmm_set_log10p_output(blmm_ctx* ctx, double* dP_out, int64_t ldP, int64_t chisq_df) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (chisq_df < 0 || chisq_df > 1000000 || (dP_out && ldP < 1)) return fail(ctx, BLMM_ERR_INVALID, "set_log10p_output: bad arguments");
  ctx->pv_armed = chisq_df > 0;
  ctx->pv_out = dP_out; ctx->pv_ld = ldP; ctx->pv_df = chisq_df > 0 ? chisq_df : 1;
  return BLMM_OK;
}

// null-exact runs the low-rank weights form (kernels_lowrank.hip) unless BLMM_EXACT=full (A/B testing), c >= 4 (the
// kernel would spill) or n is beyond what the basis kernel keeps in LDS
static bool exact_full(const blmm_ctx* ctx) {   // tuning key "exact_full_rank"
  const char* e = dev_env("BLMM_EXACT");
  return e ? std::strcmp(e, "full") == 0 : ctx->tune.exact_full_rank != 0;
}
static bool wants_lowrank(const blmm_ctx* ctx, const blmm_opts* opts, int64_t n, const double* dCovar, int64_t ncov) {
  const int c_eff = (int)((ncov == 0 || !dCovar) ? 1 : ncov + (opts->add_intercept ? 1 : 0));
  return opts->method == BLMM_NULL_EXACT && !exact_full(ctx) && c_eff <= 3 && n <= 6000;
}

// dL_out == nullptr: only with ctx->red_cur set (blmm_bulkscan_reduced: the scan kernels reduce in their epilogues)
static int bulkscan_dev_impl(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t n, int64_t m, const double* dG,
                             int64_t p, const double* dCovar, int64_t ncov, const double* dK, const double* dweights,
                             const double* h2_grid_host, int64_t ngrid, double* dL_out, int64_t ldL, double* dh2_out,
                             blmm_status* status, const PvReq& pvreq) {
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (!dY || !dG || !dK || (!dL_out && !ctx->red_cur.pmax) || !dh2_out) return fail(ctx, BLMM_ERR_INVALID, "bulkscan: NULL buffer");
  if (ldL < p) return fail(ctx, BLMM_ERR_INVALID, "bulkscan: ldL < p");
  if (opts->method != BLMM_NULL_EXACT && opts->method != BLMM_NULL_GRID && opts->method != BLMM_ALT_GRID)
    return fail(ctx, BLMM_ERR_METHOD, "Unknown method; choose null-exact, null-grid or alt-grid.");
  BLMM_HIP(hipSetDevice(ctx->device));
  if ((rc = check_sticky(ctx))) return rc;
  Timer tm(ctx);
  Pipe P;
  double* dgrid = nullptr;
  if (opts->method != BLMM_NULL_EXACT) {
    if ((rc = grid_to_device(ctx, h2_grid_host, ngrid, &dgrid))) return rc;
  }
  const bool lowrank = wants_lowrank(ctx, opts, n, dCovar, ncov);
  if ((rc = prepare(ctx, opts, dY, n, m, dG, p, dCovar, ncov, dK, dweights, 1, P, tm, lowrank, opts->method != BLMM_NULL_EXACT))) return rc;
  return scan_pipeline(ctx, opts, P, tm, lowrank, /*wbasis_started*/ lowrank && m > 0 && p > 0, dgrid, h2_grid_host, ngrid, dL_out, ldL, dh2_out, status, pvreq);
}

int blmm_bulkscan_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t n, int64_t m, const double* dG,
                      int64_t p, const double* dCovar, int64_t ncov, const double* dK, const double* dweights,
                      const double* h2_grid_host, int64_t ngrid, double* dL_out, int64_t ldL, double* dh2_out,
                      blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  const PvReq pvreq = pv_take(ctx);
  ctx->red_cur = RedArgs();
  return bulkscan_dev_impl(ctx, opts, dY, n, m, dG, p, dCovar, ncov, dK, dweights, h2_grid_host, ngrid, dL_out, ldL, dh2_out, status, pvreq);
}

// ---------------------------------------------------------------------------------------------------
// bulkscan without the LOD matrix (include/bulklmm_hip.h: blmm_bulkscan_reduced).  `out` holds DEVICE pointers here.
// Native route (null-grid; null-exact in the low-rank weights form): the scan kernels' reduce-in-epilogue instantiations write
// per-(trait, 64-marker slot) partial maxima and the triplets, k_red_final finishes the maxima -- L is never written.  The rare
// per-trait re-scans (k_scan_fix: expansion residual of the weight basis; k_scan_qr: ill-conditioned weighted covariates) patch
// columns of a stored L, which does not exist here: the route SPECULATES that no trait is flagged, reads the two device counts at
// the end, and when one is non-zero -- or the method / covariate count has no fused instantiation (alt-grid, c >= 4, BLMM_EXACT=full)
// -- the call runs once more into the context's resident L and reduces it with k_colmax / k_threshold (the same values by
// construction).  Synchronises the stream before it returns.
static int reduced_impl(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t n, int64_t m, const double* dG, int64_t p,
                        const double* dCovar, int64_t ncov, const double* dK, const double* dweights, const double* h2_grid_host,
                        int64_t ngrid, const blmm_reduced* out, double* dh2_out, blmm_status* status, int* route_out) {
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (!out || !dY || !dG || !dK || (!dh2_out && opts->method != BLMM_ALT_GRID)) return fail(ctx, BLMM_ERR_INVALID, "bulkscan_reduced: NULL buffer");
  if (out->cap < 0 || (out->cap > 0 && (!out->ti || !out->tj || !out->tlod)) || (out->want_triplets && !out->count))
    return fail(ctx, BLMM_ERR_INVALID, "bulkscan_reduced: triplet buffers");
  if (opts->method != BLMM_NULL_EXACT && opts->method != BLMM_NULL_GRID && opts->method != BLMM_ALT_GRID)
    return fail(ctx, BLMM_ERR_METHOD, "Unknown method; choose null-exact, null-grid or alt-grid.");
  if (n < 1 || m < 0 || p < 0 || p > 0x7fffffffLL || m > 0x7fffffffLL) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  BLMM_HIP(hipSetDevice(ctx->device));
  const bool native = p > 0 && m > 0 && (opts->method == BLMM_NULL_GRID || wants_lowrank(ctx, opts, n, dCovar, ncov));
  if (route_out) *route_out = 0;
  if (native) {
    const int nslot = 2 * (int)((p + 127) / 128);
    const int64_t ldm = round_up(m, 64);
    if ((rc = ensure(ctx, ctx->redbuf, (sizeof(double) + sizeof(int)) * (size_t)nslot * (size_t)ldm))) return rc;
    RedArgs r;
    r.pmax = ptr<double>(ctx->redbuf); r.parg = reinterpret_cast<int*>(r.pmax + (size_t)nslot * ldm); r.ldm = ldm;
    r.want_trip = out->want_triplets ? 1 : 0; r.thr = out->thr; r.cap = out->cap;
    r.ti = out->ti; r.tj = out->tj; r.tl = out->tlod; r.cnt = reinterpret_cast<unsigned long long*>(out->count);
    if (out->count) BLMM_HIP(hipMemsetAsync(out->count, 0, sizeof(int64_t), ctx->stream));
    ctx->red_cur = r;
    rc = bulkscan_dev_impl(ctx, opts, dY, n, m, dG, p, dCovar, ncov, dK, dweights, h2_grid_host, ngrid, nullptr, p, dh2_out, status, PvReq());
    ctx->red_cur = RedArgs();
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if ((rc = launch_red_final(ctx, r, nslot, m, out->colmax, out->argmax))) return rc;
    int64_t h[NSTAT];
    BLMM_HIP(hipMemcpyAsync(h, ctx->stat.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    BLMM_HIP(hipStreamSynchronize(ctx->stream));
    if ((rc = check_sticky(ctx))) return rc;
    if (h[10] == 0 && h[ST_ILLCOND] == 0) { if (route_out) *route_out = 1; return BLMM_OK; }
  }
  // through a resident L
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * (size_t)(p > 0 ? p : 1) * (size_t)(m > 0 ? m : 1)))) return rc;
  double* dL = ptr<double>(ctx->outL);
  double* dH = dh2_out;
  if (opts->method == BLMM_ALT_GRID) {      // h2_panel (p x m) is not part of the reduced result: into the workspace
    if ((rc = ensure(ctx, ctx->altbuf, sizeof(double) * (size_t)(p > 0 ? p : 1) * (size_t)(m > 0 ? m : 1)))) return rc;
    dH = ptr<double>(ctx->altbuf);
  }
  if ((rc = bulkscan_dev_impl(ctx, opts, dY, n, m, dG, p, dCovar, ncov, dK, dweights, h2_grid_host, ngrid, dL, p > 0 ? p : 1, dH, status, PvReq()))) {
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  if (m > 0) { ctx->last_L = dL; ctx->last_p = p; ctx->last_m = m; ctx->last_f32 = false; }
  if (out->colmax && m > 0 && (rc = launch_colmax(ctx, dL, p, m, p > 0 ? p : 1, out->colmax, out->argmax))) return rc;
  if (out->want_triplets && (rc = launch_threshold(ctx, dL, p, m, p > 0 ? p : 1, out->thr, out->cap, out->ti, out->tj, out->tlod, out->count))) return rc;
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  if (route_out) *route_out = 2;
  return check_sticky(ctx);
}

int blmm_bulkscan_reduced_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t n, int64_t m, const double* dG,
                              int64_t p, const double* dCovar, int64_t ncov, const double* dK, const double* dweights,
                              const double* h2_grid_host, int64_t ngrid, const blmm_reduced* out, double* dh2_out,
                              blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  (void)pv_take(ctx);
  return reduced_impl(ctx, opts, dY, n, m, dG, p, dCovar, ncov, dK, dweights, h2_grid_host, ngrid, out, dh2_out, status, &ctx->last_reduced_route);
}

// ---------------------------------------------------------------------------------------------------
// One process per GPU (torch.distributed / MPI hosts): the pipeline in three calls, so that the marker rotation -- replicated
// work when every rank runs blmm_bulkscan_dev on the full G -- is SHARDED.  At n >= 500 it is 10-25 % of a rank's step
// (configs[4]: 3.8 ms of 16.5), against ~0.7 ms for an all-gather of the rotated blocks over xGMI; the eigen-decomposition stays
// replicated (broadcasting U would synchronise all devices in the middle of the call).
//   blmm_prepare_dev               design, eigen, post-eigen: the context then holds U, lambda, Z0 and the rotation matrix
//   blmm_rotate_block_dev          rank r rotates ITS column block of G into a k-major block (rows = blmm_rotated_rows())
//   [the host all-gathers the blocks: RCCL ncclAllGather / torch.distributed.all_gather_into_tensor]
//   blmm_bulkscan_prerotated_dev   assembles Xt from the gathered blocks, rotates this rank's traits, runs the method
// A marker's rotated column is the same bits whichever rank and block shape produced it (fixed summation order), so the result
// equals blmm_bulkscan_dev's bit for bit (tests/test_gpu_configs.py).
int blmm_prepare_dev(blmm_ctx* ctx, const blmm_opts* opts, int64_t n, const double* dCovar, int64_t ncov, const double* dK,
                     const double* dweights, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (!dK) return fail(ctx, BLMM_ERR_INVALID, "prepare: NULL buffer");
  BLMM_HIP(hipSetDevice(ctx->device));
  if ((rc = check_sticky(ctx))) return rc;
  ctx->prep_valid = false;
  Timer tm(ctx);
  Pipe P;
  if ((rc = prepare_eigen(ctx, opts, n, dCovar, ncov, dK, dweights, 1, P, tm))) return rc;
  ctx->prep = P;
  ctx->prep_valid = true;
  return end_call(ctx, P, status, &tm);
}

// The state blmm_prepare_dev left, with the device pointers taken from the workspace as it is NOW (never from the copy)
static Pipe prepared_pipe(blmm_ctx* ctx) {
  Pipe P = ctx->prep;
  P.Z0 = ptr<double>(ctx->Z0); P.lam = ptr<double>(ctx->lam); P.stat = ptr<int64_t>(ctx->stat);
  P.Yt = nullptr; P.Xt = nullptr;
  return P;
}

int64_t blmm_rotated_rows(const blmm_ctx* ctx) { return (ctx && ctx->prep_valid) ? ctx->prep.npad : 0; }

int blmm_rotate_block_dev(blmm_ctx* ctx, const double* dG_block, int64_t pb, double* dXt_block, int64_t ld) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!ctx->prep_valid) return fail(ctx, BLMM_ERR_INVALID, "rotate_block: blmm_prepare_dev has not run on this context");
  if (!dG_block || !dXt_block || pb < 0 || ld < pb) return fail(ctx, BLMM_ERR_INVALID, "rotate_block: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  const Pipe P = prepared_pipe(ctx);
  return launch_rotate(ctx, ptr<double>(ctx->Rp), P.ldr, P.n, P.npad, dG_block, pb, dXt_block, ld, ld);
}

// Xt (k-major, npad x ldx) from the gathered blocks [nblocks][npad][block_ld]: block b holds the columns [b block_cols, ..)
static int assemble_prerotated(blmm_ctx* ctx, Pipe& P, int64_t p, const double* dXt_blocks, int64_t nblocks, int64_t block_cols,
                               int64_t block_ld) {
  int rc;
  P.p = p; P.ldx = round_up(p > 0 ? p : 1, 128);
  if ((rc = ensure(ctx, ctx->Xt, sizeof(double) * (size_t)P.npad * P.ldx))) return rc;
  P.Xt = ptr<double>(ctx->Xt);
  if (P.ldx > p) BLMM_HIP(hipMemset2DAsync(P.Xt + p, sizeof(double) * P.ldx, 0, sizeof(double) * (P.ldx - p), P.npad, ctx->stream));
  for (int64_t b = 0; b < nblocks; ++b) {
    const int64_t lo = b * block_cols, hi = (lo + block_cols < p) ? lo + block_cols : p;
    if (hi <= lo) break;
    BLMM_HIP(hipMemcpy2DAsync(P.Xt + lo, sizeof(double) * P.ldx, dXt_blocks + (size_t)b * P.npad * block_ld, sizeof(double) * block_ld,
                              sizeof(double) * (hi - lo), P.npad, hipMemcpyDeviceToDevice, ctx->stream));
  }
  return BLMM_OK;
}

int blmm_bulkscan_prerotated_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t m, int64_t p,
                                 const double* dXt_blocks, int64_t nblocks, int64_t block_cols, int64_t block_ld,
                                 const double* h2_grid_host, int64_t ngrid, double* dL_out, int64_t ldL, double* dh2_out,
                                 blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  const PvReq pvreq = pv_take(ctx);
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (!ctx->prep_valid) return fail(ctx, BLMM_ERR_INVALID, "bulkscan_prerotated: blmm_prepare_dev has not run on this context");
  if (!dY || !dXt_blocks || !dL_out || !dh2_out) return fail(ctx, BLMM_ERR_INVALID, "bulkscan: NULL buffer");
  if (m < 0 || p < 0 || nblocks < 1 || block_cols < 1 || block_ld < block_cols || nblocks * block_cols < p)
    return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  if (ldL < p) return fail(ctx, BLMM_ERR_INVALID, "bulkscan: ldL < p");
  if (opts->method != BLMM_NULL_EXACT && opts->method != BLMM_NULL_GRID && opts->method != BLMM_ALT_GRID)
    return fail(ctx, BLMM_ERR_METHOD, "Unknown method; choose null-exact, null-grid or alt-grid.");
  BLMM_HIP(hipSetDevice(ctx->device));
  if ((rc = check_sticky(ctx))) return rc;
  Timer tm(ctx);
  Pipe P = prepared_pipe(ctx);
  double* dgrid = nullptr;
  if (opts->method != BLMM_NULL_EXACT) {
    if ((rc = grid_to_device(ctx, h2_grid_host, ngrid, &dgrid))) return rc;
  }
  // the counters of this scan start from zero except what the eigen-decomposition left (negative eigenvalues, its clocks)
  BLMM_HIP(hipMemsetAsync(P.stat + 1, 0, sizeof(int64_t) * 4, ctx->stream));
  BLMM_HIP(hipMemsetAsync(P.stat + 8, 0, sizeof(int64_t) * (NSTAT - 8), ctx->stream));
  ctx->audit_ran = false;
  ctx->brent_cnt_used = false;
  tm.mark(); tm.mark();
  const bool lowrank = opts->method == BLMM_NULL_EXACT && !exact_full(ctx) && P.c <= 3;
  if (lowrank && m > 0 && p > 0 && (rc = start_wbasis(ctx, P))) return rc;
  if ((rc = rotate_traits(ctx, P, dY, m))) return rc;
  if ((rc = assemble_prerotated(ctx, P, p, dXt_blocks, nblocks, block_cols, block_ld))) return rc;
  tm.mark();
  return scan_pipeline(ctx, opts, P, tm, lowrank, lowrank && m > 0 && p > 0, dgrid, h2_grid_host, ngrid, dL_out, ldL, dh2_out, status, pvreq);
}

// host inputs of a bulkscan call -> the context's input buffers (asynchronous on the context's stream)
static int upload_bulk_inputs(blmm_ctx* ctx, const double* Y, int64_t n, int64_t m, const double* G, int64_t p, const double* Covar,
                              int64_t ncov, const double* K, const double* weights, const double** dCov, const double** dW) {
  int rc;
  if ((rc = ensure(ctx, ctx->inY, sizeof(double) * n * (m > 0 ? m : 1)))) return rc;
  if ((rc = ensure(ctx, ctx->inG, sizeof(double) * n * (p > 0 ? p : 1)))) return rc;
  if ((rc = ensure(ctx, ctx->inK, sizeof(double) * n * n))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->inK.p, K, sizeof(double) * n * n, hipMemcpyHostToDevice, ctx->stream));
  // Y and G: left for prepare(), which copies them on ctx->copy once the eigen phase is queued (see blmm_ctx::up_pending)
  ctx->up_src[0] = Y; ctx->up_dst[0] = ctx->inY.p; ctx->up_bytes[0] = sizeof(double) * (size_t)n * (size_t)m;
  ctx->up_src[1] = G; ctx->up_dst[1] = ctx->inG.p; ctx->up_bytes[1] = sizeof(double) * (size_t)n * (size_t)p;
  ctx->up_pending = true; ctx->in_wait = false;
  *dCov = nullptr; *dW = nullptr;
  if (Covar && ncov > 0) {
    if ((rc = ensure(ctx, ctx->inCov, sizeof(double) * n * ncov))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->inCov.p, Covar, sizeof(double) * n * ncov, hipMemcpyHostToDevice, ctx->stream));
    *dCov = ptr<double>(ctx->inCov);
  }
  if (weights) {
    if ((rc = ensure(ctx, ctx->inW, sizeof(double) * n))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->inW.p, weights, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    *dW = ptr<double>(ctx->inW);
  }
  return BLMM_OK;
}

// L_out == NULL: the matrix stays in HBM (2.08 GB at BXD size: 36 of the call's 39 ms are its trip over PCIe) and the blmm_last_*
// consumers serve it -- peaks, LOD > t triplets, permutation quantiles, -log10 p, single columns (README.md:246-255, 354-359;
// src/analysis_helpers/single_trait_analysis.jl:13-23 are what the reference's users do with L).  alt-grid: h2_out (the p x m
// h2_panel) may be NULL likewise.
int blmm_bulkscan(blmm_ctx* ctx, const blmm_opts* opts, const double* Y, int64_t n, int64_t m, const double* G, int64_t p,
                  const double* Covar, int64_t ncov, const double* K, const double* weights, const double* h2_grid,
                  int64_t ngrid, double* L_out, double* h2_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  const PvReq pvreq = pv_take(ctx);
  if (!opts) return fail(ctx, BLMM_ERR_INVALID, "opts is NULL");
  const bool alt = opts->method == BLMM_ALT_GRID;
  if (!Y || !G || !K || (!h2_out && !alt)) return fail(ctx, BLMM_ERR_INVALID, "bulkscan: NULL buffer");
  if (n < 1 || m < 0 || p < 0) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  const size_t h2_elems = alt ? (size_t)p * m : (size_t)m;
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * (size_t)p * m))) return rc;
  if ((rc = ensure(ctx, ctx->outH2, sizeof(double) * h2_elems))) return rc;
  // BLMM_HOST_PROF=1: wall-clock of the call's legs on stderr (diagnostic: it synchronises between them)
  static const bool hprof = getenv("BLMM_HOST_PROF") && getenv("BLMM_HOST_PROF")[0] == '1';
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double hp0 = hprof ? now() : 0.0;
  const double* dCov = nullptr; const double* dW = nullptr;
  if ((rc = upload_bulk_inputs(ctx, Y, n, m, G, p, Covar, ncov, K, weights, &dCov, &dW))) return rc;
  if (hprof) (void)hipStreamSynchronize(ctx->stream);
  const double hp1 = hprof ? now() : 0.0;
  pv_hand_over(ctx, pvreq);
  rc = blmm_bulkscan_dev(ctx, opts, ptr<double>(ctx->inY), n, m, ptr<double>(ctx->inG), p, dCov, dCov ? ncov : 0,
                         ptr<double>(ctx->inK), dW, h2_grid, ngrid, ptr<double>(ctx->outL), p, ptr<double>(ctx->outH2), status);
  ctx->up_pending = false; ctx->in_wait = false;        // (a call refused before prepare() never uploaded them)
  if (rc) { hipStreamSynchronize(ctx->stream); return rc; }
  const double hp2 = hprof ? now() : 0.0;
  if (hprof) (void)hipStreamSynchronize(ctx->stream);
  const double hp3 = hprof ? now() : 0.0;
  ctx->last_L = ptr<double>(ctx->outL); ctx->last_p = p; ctx->last_m = m; ctx->last_f32 = false;
  if (L_out && (size_t)p * m > 0 && (rc = copy_to_host(ctx, L_out, ctx->outL.p, sizeof(double) * (size_t)p * m))) return rc;
  const double hp4 = hprof ? now() : 0.0;
  if (h2_out && h2_elems > 0 && (rc = copy_to_host(ctx, h2_out, ctx->outH2.p, sizeof(double) * h2_elems))) return rc;
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  if (hprof)
    fprintf(stderr, "blmm_bulkscan legs (ms): uploads %.2f | enqueue %.2f | device %.2f | L to host %.2f | h2 to host + sync %.2f\n", hp1 - hp0, hp2 - hp1,
            hp3 - hp2, hp4 - hp3, now() - hp4);
  return check_sticky(ctx);   // a device-side failure of THIS call (no status passed): reported now, not by the next call
}

// host-pointer form of the reduce-in-epilogue scan: `out` holds HOST pointers; the small results come back, nothing p x m moves
int blmm_bulkscan_reduced(blmm_ctx* ctx, const blmm_opts* opts, const double* Y, int64_t n, int64_t m, const double* G, int64_t p,
                          const double* Covar, int64_t ncov, const double* K, const double* weights, const double* h2_grid,
                          int64_t ngrid, const blmm_reduced* out, double* h2_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  (void)pv_take(ctx);
  if (!opts) return fail(ctx, BLMM_ERR_INVALID, "opts is NULL");
  const bool alt = opts->method == BLMM_ALT_GRID;
  if (!out || !Y || !G || !K || (!h2_out && !alt)) return fail(ctx, BLMM_ERR_INVALID, "bulkscan_reduced: NULL buffer");
  if (out->cap < 0 || (out->cap > 0 && (!out->ti || !out->tj || !out->tlod)) || (out->want_triplets && !out->count))
    return fail(ctx, BLMM_ERR_INVALID, "bulkscan_reduced: triplet buffers");
  if (n < 1 || m < 0 || p < 0) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  const int64_t cap = out->cap > 0 ? out->cap : 0, mm = m > 0 ? m : 1;
  // device side of `out`: maxima / arg-maxima (tmpA / tmpB), triplets + count (redtrip), h2 (outH2)
  if ((rc = ensure(ctx, ctx->tmpA, sizeof(double) * (size_t)mm))) return rc;
  if ((rc = ensure(ctx, ctx->tmpB, sizeof(int64_t) * (size_t)mm))) return rc;
  if ((rc = ensure(ctx, ctx->redtrip, (sizeof(double) + 2 * sizeof(int32_t)) * (size_t)(cap > 0 ? cap : 1) + 64))) return rc;
  if ((rc = ensure(ctx, ctx->outH2, sizeof(double) * (size_t)mm))) return rc;
  blmm_reduced d = *out;
  d.colmax = (out->colmax || out->argmax) ? ptr<double>(ctx->tmpA) : nullptr;
  d.argmax = out->argmax ? ptr<int64_t>(ctx->tmpB) : nullptr;
  d.count = ptr<int64_t>(ctx->redtrip);
  d.tlod = reinterpret_cast<double*>(d.count + 8);
  d.ti = reinterpret_cast<int32_t*>(d.tlod + (cap > 0 ? cap : 1));
  d.tj = d.ti + (cap > 0 ? cap : 1);
  const double* dCov = nullptr; const double* dW = nullptr;
  if ((rc = upload_bulk_inputs(ctx, Y, n, m, G, p, Covar, ncov, K, weights, &dCov, &dW))) return rc;
  rc = reduced_impl(ctx, opts, ptr<double>(ctx->inY), n, m, ptr<double>(ctx->inG), p, dCov, dCov ? ncov : 0, ptr<double>(ctx->inK), dW,
                    h2_grid, ngrid, &d, ptr<double>(ctx->outH2), status, &ctx->last_reduced_route);
  ctx->up_pending = false; ctx->in_wait = false;
  if (rc) return rc;
  if (m > 0) {
    if (out->colmax) BLMM_HIP(hipMemcpyAsync(out->colmax, d.colmax, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    if (out->argmax) BLMM_HIP(hipMemcpyAsync(out->argmax, d.argmax, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    if (h2_out && !alt) BLMM_HIP(hipMemcpyAsync(h2_out, ctx->outH2.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (out->want_triplets) {
    BLMM_HIP(hipMemcpyAsync(out->count, d.count, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    BLMM_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t got = *out->count < cap ? *out->count : cap;
    if (got > 0) {
      BLMM_HIP(hipMemcpyAsync(out->tlod, d.tlod, sizeof(double) * (size_t)got, hipMemcpyDeviceToHost, ctx->stream));
      BLMM_HIP(hipMemcpyAsync(out->ti, d.ti, sizeof(int32_t) * (size_t)got, hipMemcpyDeviceToHost, ctx->stream));
      BLMM_HIP(hipMemcpyAsync(out->tj, d.tj, sizeof(int32_t) * (size_t)got, hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return check_sticky(ctx);
}

int blmm_last_reduced_route(const blmm_ctx* ctx) { return ctx ? ctx->last_reduced_route : 0; }

// ---------------------------------------------------------------------------------------------------
// dLperms_out (fp64) or dLperms32_out (fp32, kernels_scan_f32.hip): exactly one of them when nperms > 0
static int perms_pipeline(blmm_ctx* ctx, const blmm_opts* opts, Pipe& P, Timer& tm, int64_t nperms, uint64_t seed,
                          const int32_t* dperm_idx, double* dscalars_out, double* dlod_out, double* dLperms_out,
                          float* dLperms32_out, blmm_status* status, const double* dG_raw = nullptr);
// fp32 permutation path with its own fp32 rotation (kernels_scan_f32.hip: k_rotate_f32): intercept-only null model (the
// conditioning guard of more covariates re-scans from the fp64 rotated markers), tuning key "f32_rotation"
static bool f32_rotation_route(const blmm_ctx* ctx, const blmm_opts* o, const double* dCovar, int64_t ncov, int64_t nperms, int64_t p, bool f32) {
  const int c_eff = (int)((ncov == 0 || !dCovar) ? 1 : ncov + (o->add_intercept ? 1 : 0));
  return f32 && nperms > 0 && p > 0 && c_eff == 1 && ctx->tune.f32_rotation != 0;
}

static int scan_perms_impl(blmm_ctx* ctx, const blmm_opts* opts, const double* dy, int64_t n, const double* dG, int64_t p,
                           const double* dCovar, int64_t ncov, const double* dK, const double* dweights, int64_t nperms,
                           uint64_t seed, const int32_t* dperm_idx, double* dscalars_out, double* dlod_out,
                           double* dLperms_out, float* dLperms32_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (nperms < 0) return fail(ctx, BLMM_ERR_NPERMS, "The required number of permutations must be a positive integer.");
  if (!dy || !dG || !dK || !dscalars_out || !dlod_out || (nperms > 0 && !dLperms_out && !dLperms32_out))
    return fail(ctx, BLMM_ERR_INVALID, "scan_perms: NULL buffer");
  BLMM_HIP(hipSetDevice(ctx->device));
  if ((rc = check_sticky(ctx))) return rc;
  Timer tm(ctx);
  Pipe P;
  // the library's own permutation indices depend on nothing in the call: generated on the side stream, beside the eigen-decomposition
  // (the multi-kernel panel form, n > 256, consumes them; ev_m orders them in front of the panels)
  ctx->perm_ready = false;
  bool perm_side = false;
  if (!dperm_idx && nperms > 0 && n > 256 && n <= 65535) {
    hipStream_t main_stream = ctx->stream;
    BLMM_HIP(hipEventRecord(ctx->ev_xt, main_stream));
    BLMM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_xt, 0));
    ctx->stream = ctx->side;
    rc = launch_perm_gen(ctx, (int)n, nperms, seed);
    ctx->stream = main_stream;
    if (rc) return rc;
    BLMM_HIP(hipEventRecord(ctx->ev_m, ctx->side));
    perm_side = true;
  }
  const bool own_rot = f32_rotation_route(ctx, opts, dCovar, ncov, nperms, p, dLperms32_out != nullptr);
  rc = prepare(ctx, opts, dy, n, 1, dG, p, dCovar, ncov, dK, dweights, 1, P, tm, false, false, /*skip_markers*/ own_rot);
  if (perm_side) BLMM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_m, 0));      // (also on the error path: the side stream joins the call)
  if (rc) { ctx->perm_ready = false; return rc; }
  return perms_pipeline(ctx, opts, P, tm, nperms, seed, dperm_idx, dscalars_out, dlod_out, dLperms_out, dLperms32_out, status, own_rot ? dG : nullptr);
}

// Everything of the permutation test behind the rotations (shared with blmm_scan_perms_prerotated_dev)
static int perms_pipeline(blmm_ctx* ctx, const blmm_opts* opts, Pipe& P, Timer& tm, int64_t nperms, uint64_t seed,
                          const int32_t* dperm_idx, double* dscalars_out, double* dlod_out, double* dLperms_out,
                          float* dLperms32_out, blmm_status* status, const double* dG_raw) {
  int rc;
  const int64_t p = P.p;
  const NullModel nm = null_model(P, opts);
  // scalars: [sigma2_e, h2_null]
  if ((rc = launch_brent(ctx, nm, P.Yt, P.ldy, 1, P.Z0, P.lam, dscalars_out + 1, dscalars_out, nullptr, P.stat))) return rc;
  tm.mark();
  const int64_t ldp0 = 128, ldp1 = round_up(nperms > 0 ? nperms : 1, 128);
  if ((rc = ensure(ctx, ctx->panels, sizeof(double) * (size_t)P.npad * (ldp0 + ldp1)))) return rc;
  double* pan0 = ptr<double>(ctx->panels);
  double* pan1 = pan0 + (size_t)P.npad * ldp0;
  if ((rc = launch_perm_panel(ctx, nm, P.Yt, P.ldy, P.Z0, P.lam, dscalars_out + 1, nullptr, 0, seed, 1, pan0, ldp0, P.stat))) return rc;
  if (nperms > 0)
    if ((rc = launch_perm_panel(ctx, nm, P.Yt, P.ldy, P.Z0, P.lam, dscalars_out + 1, dperm_idx, nperms, seed, 0, pan1, ldp1, P.stat))) return rc;
  if ((rc = ensure(ctx, ctx->isx, sizeof(double) * (size_t)P.ldx))) return rc;
  if (dG_raw) {
    // ---- fp32 from the rotation on (round 4): XF = R G on the fp32 matrix cores straight into k_scan_f32's operand layout (no fp64
    //      rotated markers, no conversion pass), the marker norms from XF (fp64 sums), and the original trait's LOD vector with an
    //      fp64 numerator taken from G itself (g_i' R'a0) -- it agrees with the fp64 path to ~1e-7 relative (the norms carry the
    //      fp32 rounding of the rotated markers, averaged over n)
    const int64_t ldxf = round_up(p, 256), ldpf = ldp1;
    const int kpad = (int)round_up(P.npad, 128), ldrr = (int)round_up(P.n, 16);
    if ((rc = ensure(ctx, ctx->xf32, sizeof(float) * (size_t)P.npad * ldxf))) return rc;
    if ((rc = ensure(ctx, ctx->pf32, sizeof(float) * (size_t)P.npad * ldpf))) return rc;
    if ((rc = ensure(ctx, ctx->rf32, sizeof(float) * (size_t)kpad * ldrr + sizeof(double) * ((size_t)ldrr + (size_t)p) + 64))) return rc;
    float* RF = ptr<float>(ctx->rf32);
    double* vwork = reinterpret_cast<double*>(RF + (((size_t)kpad * ldrr + 3) & ~(size_t)3));
    double* numv = vwork + ldrr;
    if ((rc = launch_rotate_f32(ctx, ptr<double>(ctx->Rp), P.ldr, P.n, P.npad, dG_raw, p, RF, ptr<float>(ctx->xf32), ldxf, pan0, ldp0, vwork, numv))) return rc;
    if ((rc = launch_isx_f32(ctx, nm, ptr<float>(ctx->xf32), ldxf, p, P.Z0, P.lam, dscalars_out + 1, ptr<double>(ctx->isx), P.ldx, P.stat))) return rc;
    tm.mark();
    if ((rc = launch_lod_from_num(ctx, numv, ptr<double>(ctx->isx), P.n, p, dlod_out, P.stat))) return rc;
    if ((rc = launch_cvt_f32(ctx, pan1, ldp1, P.n, nperms, ptr<float>(ctx->pf32), ldpf, P.npad / 8))) return rc;
    if ((rc = launch_scan_f32(ctx, ptr<float>(ctx->xf32), ldxf, ptr<float>(ctx->pf32), ldpf, P.npad, P.n, p, nperms,
                              ptr<double>(ctx->isx), dLperms32_out, p, P.stat))) return rc;
    tm.mark();
    return end_call(ctx, P, status, &tm);
  }
  if ((rc = launch_isx(ctx, nm, P.Xt, P.ldx, p, P.Z0, P.lam, dscalars_out + 1, 1, ptr<double>(ctx->isx), P.ldx, P.stat))) return rc;
  tm.mark();
  if (p > 0) {
    ScanArgs a = scan_args(ctx, P, pan0, ldp0, dlod_out, p, 1);
    a.isx = ptr<double>(ctx->isx); a.ld_isx = P.ldx;
    if ((rc = launch_scan_table(ctx, a))) return rc;
    // the trait's own LOD vector (scan_null's result) gets the conditioning guard; the permutation matrix keeps the Cholesky form
    if ((rc = illcond_rescan(ctx, P, nm, 1, dscalars_out + 1, dlod_out, p))) return rc;
    if (nperms > 0 && dLperms32_out) {
      // fp32 path: fp32 fragment-major copies of the rotated markers and of the permutation panel, fp32 MFMA, fp32 L
      const int64_t ldxf = round_up(p, 256), ldpf = ldp1;
      if ((rc = ensure(ctx, ctx->xf32, sizeof(float) * (size_t)P.npad * ldxf))) return rc;
      if ((rc = ensure(ctx, ctx->pf32, sizeof(float) * (size_t)P.npad * ldpf))) return rc;
      if ((rc = launch_cvt_f32(ctx, P.Xt, P.ldx, P.n, p, ptr<float>(ctx->xf32), ldxf, P.npad / 8))) return rc;
      if ((rc = launch_cvt_f32(ctx, pan1, ldp1, P.n, nperms, ptr<float>(ctx->pf32), ldpf, P.npad / 8))) return rc;
      if ((rc = launch_scan_f32(ctx, ptr<float>(ctx->xf32), ldxf, ptr<float>(ctx->pf32), ldpf, P.npad, P.n, p, nperms,
                                ptr<double>(ctx->isx), dLperms32_out, p, P.stat))) return rc;
    } else if (nperms > 0) {
      ScanArgs b = scan_args(ctx, P, pan1, ldp1, dLperms_out, p, nperms);
      b.isx = ptr<double>(ctx->isx); b.ld_isx = P.ldx;
      if ((rc = launch_scan_table(ctx, b))) return rc;
    }
  }
  tm.mark();
  return end_call(ctx, P, status, &tm);
}

int blmm_scan_perms_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dy, int64_t n, const double* dG, int64_t p,
                        const double* dCovar, int64_t ncov, const double* dK, const double* dweights, int64_t nperms,
                        uint64_t seed, const int32_t* dperm_idx, double* dscalars_out, double* dlod_out,
                        double* dLperms_out, blmm_status* status) {
  if (ctx && nperms > 0 && !dLperms_out) return fail(ctx, BLMM_ERR_INVALID, "scan_perms: NULL buffer");
  return scan_perms_impl(ctx, opts, dy, n, dG, p, dCovar, ncov, dK, dweights, nperms, seed, dperm_idx, dscalars_out,
                         dlod_out, dLperms_out, nullptr, status);
}

int blmm_scan_perms_f32_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dy, int64_t n, const double* dG, int64_t p,
                            const double* dCovar, int64_t ncov, const double* dK, const double* dweights, int64_t nperms,
                            uint64_t seed, const int32_t* dperm_idx, double* dscalars_out, double* dlod_out,
                            float* dLperms_out, blmm_status* status) {
  if (ctx && nperms > 0 && !dLperms_out) return fail(ctx, BLMM_ERR_INVALID, "scan_perms: NULL buffer");
  return scan_perms_impl(ctx, opts, dy, n, dG, p, dCovar, ncov, dK, dweights, nperms, seed, dperm_idx, dscalars_out,
                         dlod_out, nullptr, dLperms_out, status);
}

// One process per GPU (include/bulklmm_hip.h: the pipeline in three calls): the permutation test on marker blocks rotated by the
// ranks and gathered by the host -- this rank's permutations (nperms, seed / dperm_idx) against all p markers.  Exactly one of
// dLperms_out (fp64) / dLperms32_out (fp32 matrix cores) is given.  Bit-identical to blmm_scan_perms[_f32]_dev.
int blmm_scan_perms_prerotated_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dy, int64_t p, const double* dXt_blocks,
                                   int64_t nblocks, int64_t block_cols, int64_t block_ld, int64_t nperms, uint64_t seed,
                                   const int32_t* dperm_idx, double* dscalars_out, double* dlod_out, double* dLperms_out,
                                   float* dLperms32_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (!ctx->prep_valid) return fail(ctx, BLMM_ERR_INVALID, "scan_perms_prerotated: blmm_prepare_dev has not run on this context");
  if (nperms < 0) return fail(ctx, BLMM_ERR_NPERMS, "The required number of permutations must be a positive integer.");
  if (!dy || !dXt_blocks || !dscalars_out || !dlod_out || (nperms > 0 && !dLperms_out == !dLperms32_out))
    return fail(ctx, BLMM_ERR_INVALID, "scan_perms_prerotated: NULL buffer (exactly one of the fp64 / fp32 permutation matrices)");
  if (p < 0 || nblocks < 1 || block_cols < 1 || block_ld < block_cols || nblocks * block_cols < p) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  BLMM_HIP(hipSetDevice(ctx->device));
  if ((rc = check_sticky(ctx))) return rc;
  Timer tm(ctx);
  Pipe P = prepared_pipe(ctx);
  BLMM_HIP(hipMemsetAsync(P.stat + 1, 0, sizeof(int64_t) * 4, ctx->stream));
  BLMM_HIP(hipMemsetAsync(P.stat + 8, 0, sizeof(int64_t) * (NSTAT - 8), ctx->stream));
  ctx->audit_ran = false;
  ctx->brent_cnt_used = false;
  ctx->perm_ready = false;
  tm.mark(); tm.mark();
  if ((rc = rotate_traits(ctx, P, dy, 1))) return rc;
  if ((rc = assemble_prerotated(ctx, P, p, dXt_blocks, nblocks, block_cols, block_ld))) return rc;
  tm.mark();
  return perms_pipeline(ctx, opts, P, tm, nperms, seed, dperm_idx, dscalars_out, dlod_out, dLperms_out, dLperms32_out, status);
}

static int scan_perms_host(blmm_ctx* ctx, const blmm_opts* opts, const double* y, int64_t n, const double* G, int64_t p,
                           const double* Covar, int64_t ncov, const double* K, const double* weights, int64_t nperms,
                           uint64_t seed, const int32_t* perm_idx, double* scalars_out, double* lod_out, void* Lperms_out,
                           bool f32, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!opts) return fail(ctx, BLMM_ERR_INVALID, "opts is NULL");
  if (nperms < 0) return fail(ctx, BLMM_ERR_NPERMS, "The required number of permutations must be a positive integer.");
  if (!y || !G || !K || !scalars_out || !lod_out || (nperms > 0 && !Lperms_out)) return fail(ctx, BLMM_ERR_INVALID, "scan_perms: NULL buffer");
  if (n < 1 || p < 0) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  BLMM_HIP(hipSetDevice(ctx->device));
  const size_t esz = f32 ? sizeof(float) : sizeof(double);
  int rc;
  if ((rc = ensure(ctx, ctx->inY, sizeof(double) * n))) return rc;
  if ((rc = ensure(ctx, ctx->inG, sizeof(double) * n * (p > 0 ? p : 1)))) return rc;
  if ((rc = ensure(ctx, ctx->inK, sizeof(double) * n * n))) return rc;
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * (size_t)p + esz * (size_t)p * nperms))) return rc;
  if ((rc = ensure(ctx, ctx->outH2, sizeof(double) * 2))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->inY.p, y, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(ctx->inG.p, G, sizeof(double) * n * p, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(ctx->inK.p, K, sizeof(double) * n * n, hipMemcpyHostToDevice, ctx->stream));
  const double* dCov = nullptr; const double* dW = nullptr; const int32_t* dperm = nullptr;
  if (Covar && ncov > 0) {
    if ((rc = ensure(ctx, ctx->inCov, sizeof(double) * n * ncov))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->inCov.p, Covar, sizeof(double) * n * ncov, hipMemcpyHostToDevice, ctx->stream));
    dCov = ptr<double>(ctx->inCov);
  }
  if (weights) {
    if ((rc = ensure(ctx, ctx->inW, sizeof(double) * n))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->inW.p, weights, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    dW = ptr<double>(ctx->inW);
  }
  if (perm_idx && nperms > 0) {
    if ((rc = ensure(ctx, ctx->tmpC, sizeof(int32_t) * (size_t)n * nperms))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->tmpC.p, perm_idx, sizeof(int32_t) * (size_t)n * nperms, hipMemcpyHostToDevice, ctx->stream));
    dperm = ptr<int32_t>(ctx->tmpC);
  }
  double* dL = ptr<double>(ctx->outL);
  void* dLp = dL + p;                       // 8-byte aligned; the fp32 kernel needs 4
  rc = scan_perms_impl(ctx, opts, ptr<double>(ctx->inY), n, ptr<double>(ctx->inG), p, dCov, dCov ? ncov : 0,
                       ptr<double>(ctx->inK), dW, nperms, seed, dperm, ptr<double>(ctx->outH2), dL,
                       f32 ? nullptr : reinterpret_cast<double*>(dLp), f32 ? reinterpret_cast<float*>(dLp) : nullptr, status);
  if (rc) { hipStreamSynchronize(ctx->stream); return rc; }
  ctx->last_L = reinterpret_cast<const double*>(dLp); ctx->last_p = p; ctx->last_m = nperms; ctx->last_f32 = f32;
  BLMM_HIP(hipMemcpyAsync(scalars_out, ctx->outH2.p, sizeof(double) * 2, hipMemcpyDeviceToHost, ctx->stream));
  if (p > 0) BLMM_HIP(hipMemcpyAsync(lod_out, dL, sizeof(double) * p, hipMemcpyDeviceToHost, ctx->stream));
  if (p > 0 && nperms > 0 && (rc = copy_to_host(ctx, Lperms_out, dLp, esz * (size_t)p * nperms))) return rc;
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return check_sticky(ctx);
}

int blmm_scan_perms(blmm_ctx* ctx, const blmm_opts* opts, const double* y, int64_t n, const double* G, int64_t p,
                    const double* Covar, int64_t ncov, const double* K, const double* weights, int64_t nperms, uint64_t seed,
                    const int32_t* perm_idx, double* scalars_out, double* lod_out, double* Lperms_out, blmm_status* status) {
  return scan_perms_host(ctx, opts, y, n, G, p, Covar, ncov, K, weights, nperms, seed, perm_idx, scalars_out, lod_out,
                         Lperms_out, false, status);
}

int blmm_scan_perms_f32(blmm_ctx* ctx, const blmm_opts* opts, const double* y, int64_t n, const double* G, int64_t p,
                        const double* Covar, int64_t ncov, const double* K, const double* weights, int64_t nperms, uint64_t seed,
                        const int32_t* perm_idx, double* scalars_out, double* lod_out, float* Lperms_out, blmm_status* status) {
  return scan_perms_host(ctx, opts, y, n, G, p, Covar, ncov, K, weights, nperms, seed, perm_idx, scalars_out, lod_out,
                         Lperms_out, true, status);
}

// ---------------------------------------------------------------------------------------------------
// scan(...; assumption = "alt") -> scan_alt (src/scan.jl:397-453): scalars [sigma2_e, h2_null], lod p, h2_each_marker p
int blmm_scan_alt_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dy, int64_t n, const double* dG, int64_t p,
                      const double* dCovar, int64_t ncov, const double* dK, const double* dweights, double* dscalars_out,
                      double* dlod_out, double* dh2_each_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (!dy || !dG || !dK || !dscalars_out || !dlod_out || !dh2_each_out) return fail(ctx, BLMM_ERR_INVALID, "scan_alt: NULL buffer");
  BLMM_HIP(hipSetDevice(ctx->device));
  if ((rc = check_sticky(ctx))) return rc;
  Timer tm(ctx);
  Pipe P;
  if ((rc = prepare(ctx, opts, dy, n, 1, dG, p, dCovar, ncov, dK, dweights, 1, P, tm))) return rc;
  const NullModel nm = null_model(P, opts);
  if (P.c + 1 >= P.n) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  if ((rc = launch_brent(ctx, nm, P.Yt, P.ldy, 1, P.Z0, P.lam, dscalars_out + 1, dscalars_out, nullptr, P.stat))) return rc;
  tm.mark();
  if ((rc = launch_alt_brent(ctx, nm, P.Yt, P.ldy, P.Xt, P.ldx, p, P.Z0, P.lam, dscalars_out + 1,
                             (opts->compat_flags & BLMM_COMPAT_ALT_TRUE_WEIGHTS) ? 1 : 0, dlod_out, dh2_each_out, P.stat))) return rc;
  tm.mark();
  return end_call(ctx, P, status, &tm);
}

// Bulk form of scan_alt (second half; the reference has only the single-trait function, src/scan.jl:397-453, and
// the grid approximation bulkscan_alt_grid): for EVERY (trait, marker) the exact heritability under the alternative -- one Brent
// search per test on the design [Z0 x_i] -- and the LOD against the trait's null model.  The per-trait null searches run in bulk
// (k_brent), the per-test searches as k_alt_brent with the trait on blockIdx.y: bit-identical, column by column, to
// blmm_scan_alt_dev on that trait.  ~0.02 us per test at n = 79 (64 traits x 7321 markers: 8.6 ms host to host; the whole BXD
// matrix would take ~5 s against the grid kernel's 15 ms): meant for trait subsets.
int blmm_bulkscan_alt_exact_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t n, int64_t m, const double* dG,
                                int64_t p, const double* dCovar, int64_t ncov, const double* dK, const double* dweights,
                                double* dL_out, int64_t ldL, double* dh2_panel_out, int64_t ldH, double* dh2_null_out,
                                double* dsigma2_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (!dY || !dG || !dK || !dL_out || !dh2_panel_out || !dh2_null_out) return fail(ctx, BLMM_ERR_INVALID, "bulkscan_alt_exact: NULL buffer");
  if (ldL < p || ldH < p) return fail(ctx, BLMM_ERR_INVALID, "bulkscan_alt_exact: leading dimension < p");
  BLMM_HIP(hipSetDevice(ctx->device));
  if ((rc = check_sticky(ctx))) return rc;
  Timer tm(ctx);
  Pipe P;
  if ((rc = prepare(ctx, opts, dY, n, m, dG, p, dCovar, ncov, dK, dweights, 1, P, tm))) return rc;
  const NullModel nm = null_model(P, opts);
  if (P.c + 1 >= P.n) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  if (m > 0) {
    double* dsig = dsigma2_out;
    if (!dsig) { if ((rc = ensure(ctx, ctx->sig2, sizeof(double) * (size_t)m))) return rc; dsig = ptr<double>(ctx->sig2); }
    if ((rc = launch_brent(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dh2_null_out, dsig, nullptr, P.stat))) return rc;
  }
  tm.mark();
  if ((rc = launch_alt_brent(ctx, nm, P.Yt, P.ldy, P.Xt, P.ldx, p, P.Z0, P.lam, dh2_null_out,
                             (opts->compat_flags & BLMM_COMPAT_ALT_TRUE_WEIGHTS) ? 1 : 0, dL_out, dh2_panel_out, P.stat, m, ldL, ldH))) return rc;
  tm.mark(); tm.mark();
  return end_call(ctx, P, status, &tm);
}

int blmm_bulkscan_alt_exact(blmm_ctx* ctx, const blmm_opts* opts, const double* Y, int64_t n, int64_t m, const double* G, int64_t p,
                            const double* Covar, int64_t ncov, const double* K, const double* weights, double* L_out,
                            double* h2_panel_out, double* h2_null_out, double* sigma2_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!opts) return fail(ctx, BLMM_ERR_INVALID, "opts is NULL");
  if (!Y || !G || !K || !L_out || !h2_panel_out || !h2_null_out) return fail(ctx, BLMM_ERR_INVALID, "bulkscan_alt_exact: NULL buffer");
  if (n < 1 || p < 1 || m < 1) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->inY, sizeof(double) * n * m))) return rc;
  if ((rc = ensure(ctx, ctx->inG, sizeof(double) * n * p))) return rc;
  if ((rc = ensure(ctx, ctx->inK, sizeof(double) * n * n))) return rc;
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * 2 * (size_t)p * m))) return rc;
  if ((rc = ensure(ctx, ctx->outH2, sizeof(double) * 2 * (size_t)m))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->inY.p, Y, sizeof(double) * n * m, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(ctx->inG.p, G, sizeof(double) * n * p, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(ctx->inK.p, K, sizeof(double) * n * n, hipMemcpyHostToDevice, ctx->stream));
  const double* dCov = nullptr; const double* dW = nullptr;
  if (Covar && ncov > 0) {
    if ((rc = ensure(ctx, ctx->inCov, sizeof(double) * n * ncov))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->inCov.p, Covar, sizeof(double) * n * ncov, hipMemcpyHostToDevice, ctx->stream));
    dCov = ptr<double>(ctx->inCov);
  }
  if (weights) {
    if ((rc = ensure(ctx, ctx->inW, sizeof(double) * n))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->inW.p, weights, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    dW = ptr<double>(ctx->inW);
  }
  double* dL = ptr<double>(ctx->outL);
  double* dH = dL + (size_t)p * m;
  double* dh2 = ptr<double>(ctx->outH2);
  rc = blmm_bulkscan_alt_exact_dev(ctx, opts, ptr<double>(ctx->inY), n, m, ptr<double>(ctx->inG), p, dCov, dCov ? ncov : 0,
                                   ptr<double>(ctx->inK), dW, dL, p, dH, p, dh2, dh2 + m, status);
  if (rc) { hipStreamSynchronize(ctx->stream); return rc; }
  ctx->last_L = dL; ctx->last_p = p; ctx->last_m = m; ctx->last_f32 = false;
  if ((rc = copy_to_host(ctx, L_out, dL, sizeof(double) * (size_t)p * m))) return rc;
  if ((rc = copy_to_host(ctx, h2_panel_out, dH, sizeof(double) * (size_t)p * m))) return rc;
  BLMM_HIP(hipMemcpyAsync(h2_null_out, dh2, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
  if (sigma2_out) BLMM_HIP(hipMemcpyAsync(sigma2_out, dh2 + m, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return check_sticky(ctx);
}

int blmm_scan_alt(blmm_ctx* ctx, const blmm_opts* opts, const double* y, int64_t n, const double* G, int64_t p,
                  const double* Covar, int64_t ncov, const double* K, const double* weights, double* scalars_out,
                  double* lod_out, double* h2_each_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!opts) return fail(ctx, BLMM_ERR_INVALID, "opts is NULL");
  if (!y || !G || !K || !scalars_out || !lod_out || !h2_each_out) return fail(ctx, BLMM_ERR_INVALID, "scan_alt: NULL buffer");
  if (n < 1 || p < 1) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->inY, sizeof(double) * n))) return rc;
  if ((rc = ensure(ctx, ctx->inG, sizeof(double) * n * p))) return rc;
  if ((rc = ensure(ctx, ctx->inK, sizeof(double) * n * n))) return rc;
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * 2 * (size_t)p))) return rc;
  if ((rc = ensure(ctx, ctx->outH2, sizeof(double) * 2))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->inY.p, y, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(ctx->inG.p, G, sizeof(double) * n * p, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(ctx->inK.p, K, sizeof(double) * n * n, hipMemcpyHostToDevice, ctx->stream));
  const double* dCov = nullptr; const double* dW = nullptr;
  if (Covar && ncov > 0) {
    if ((rc = ensure(ctx, ctx->inCov, sizeof(double) * n * ncov))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->inCov.p, Covar, sizeof(double) * n * ncov, hipMemcpyHostToDevice, ctx->stream));
    dCov = ptr<double>(ctx->inCov);
  }
  if (weights) {
    if ((rc = ensure(ctx, ctx->inW, sizeof(double) * n))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->inW.p, weights, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    dW = ptr<double>(ctx->inW);
  }
  double* dL = ptr<double>(ctx->outL);
  rc = blmm_scan_alt_dev(ctx, opts, ptr<double>(ctx->inY), n, ptr<double>(ctx->inG), p, dCov, dCov ? ncov : 0,
                         ptr<double>(ctx->inK), dW, ptr<double>(ctx->outH2), dL, dL + p, status);
  if (rc) { hipStreamSynchronize(ctx->stream); return rc; }
  ctx->last_L = dL; ctx->last_p = p; ctx->last_m = 1; ctx->last_f32 = false;
  BLMM_HIP(hipMemcpyAsync(scalars_out, ctx->outH2.p, sizeof(double) * 2, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(lod_out, dL, sizeof(double) * p, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(h2_each_out, dL + p, sizeof(double) * p, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return check_sticky(ctx);
}

// ---------------------------------------------------------------------------------------------------
// lower-level seams (host pointers)
// ---------------------------------------------------------------------------------------------------
int blmm_rotate(blmm_ctx* ctx, const blmm_opts* opts, const double* Y, int64_t n, int64_t m, const double* G, int64_t p,
                const double* Covar, int64_t ncov, const double* K, double* Y0_out, double* X0_out, double* lambda_out,
                blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (!Y || !G || !K || !Y0_out || !X0_out || !lambda_out) return fail(ctx, BLMM_ERR_INVALID, "rotate: NULL buffer");
  if (n < 1 || m < 1 || p < 1) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  BLMM_HIP(hipSetDevice(ctx->device));
  if ((rc = ensure(ctx, ctx->inY, sizeof(double) * n * m))) return rc;
  if ((rc = ensure(ctx, ctx->inG, sizeof(double) * n * p))) return rc;
  if ((rc = ensure(ctx, ctx->inK, sizeof(double) * n * n))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->inY.p, Y, sizeof(double) * n * m, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(ctx->inG.p, G, sizeof(double) * n * p, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(ctx->inK.p, K, sizeof(double) * n * n, hipMemcpyHostToDevice, ctx->stream));
  const double* dCov = nullptr;
  if (Covar && ncov > 0) {
    if ((rc = ensure(ctx, ctx->inCov, sizeof(double) * n * ncov))) return rc;
    BLMM_HIP(hipMemcpyAsync(ctx->inCov.p, Covar, sizeof(double) * n * ncov, hipMemcpyHostToDevice, ctx->stream));
    dCov = ptr<double>(ctx->inCov);
  }
  Timer tm(ctx);
  Pipe P;
  if ((rc = prepare(ctx, opts, ptr<double>(ctx->inY), n, m, ptr<double>(ctx->inG), p, dCov, dCov ? ncov : 0,
                    ptr<double>(ctx->inK), nullptr, 0, P, tm))) return rc;
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * n * (size_t)(m > p ? m : p)))) return rc;
  double* tmp = ptr<double>(ctx->outL);
  if ((rc = launch_untranspose(ctx, P.Yt, P.ldy, (int)n, m, tmp))) return rc;
  BLMM_HIP(hipMemcpyAsync(Y0_out, tmp, sizeof(double) * n * m, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  if ((rc = launch_untranspose(ctx, P.Xt, P.ldx, (int)n, p, tmp))) return rc;
  BLMM_HIP(hipMemcpyAsync(X0_out + (size_t)n * P.c, tmp, sizeof(double) * n * p, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(X0_out, P.Z0, sizeof(double) * n * P.c, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(lambda_out, P.lam, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return finish_status(ctx, status, nullptr);
}

namespace {
// uploads rotated host inputs into the pipeline's internal layouts (no eigen / rotation)
int upload_rotated(blmm_ctx* ctx, const double* Y0, int64_t n, int64_t m, const double* Z0, int64_t c, const double* X0m,
                   int64_t p, const double* lambda, Pipe& P) {
  if (n < 1 || m < 1 || c < 1 || c > CMAX || c >= n) return fail(ctx, BLMM_ERR_DIM, "Dimension mismatch.");
  ctx->prep_valid = false;   // Z0 / lambda / the status block of a blmm_prepare_dev are overwritten below
  P.n = (int)n; P.c = (int)c; P.npad = (int)round_up(n, 8); P.ldr = (int)round_up(P.npad, 16);
  P.m = m; P.p = p; P.ldy = round_up(m, 128); P.ldx = round_up(p > 0 ? p : 1, 128);
  int rc;
  if ((rc = ensure(ctx, ctx->inY, sizeof(double) * n * m))) return rc;
  if ((rc = ensure(ctx, ctx->Yt, sizeof(double) * (size_t)P.npad * P.ldy))) return rc;
  if ((rc = ensure(ctx, ctx->Z0, sizeof(double) * n * c))) return rc;
  if ((rc = ensure(ctx, ctx->lam, sizeof(double) * n))) return rc;
  if ((rc = reset_stat(ctx, &P.stat))) return rc;
  P.Yt = ptr<double>(ctx->Yt); P.Z0 = ptr<double>(ctx->Z0); P.lam = ptr<double>(ctx->lam);
  BLMM_HIP(hipMemcpyAsync(ctx->inY.p, Y0, sizeof(double) * n * m, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(P.Z0, Z0, sizeof(double) * n * c, hipMemcpyHostToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(P.lam, lambda, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = to_rowmajor(ctx, ptr<double>(ctx->inY), (int)n, m, P.Yt, P.npad, P.ldy))) return rc;
  if (X0m && p > 0) {
    if ((rc = ensure(ctx, ctx->inG, sizeof(double) * n * p))) return rc;
    if ((rc = ensure(ctx, ctx->Xt, sizeof(double) * (size_t)P.npad * P.ldx))) return rc;
    P.Xt = ptr<double>(ctx->Xt);
    BLMM_HIP(hipMemcpyAsync(ctx->inG.p, X0m, sizeof(double) * n * p, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = to_rowmajor(ctx, ptr<double>(ctx->inG), (int)n, p, P.Xt, P.npad, P.ldx))) return rc;
  }
  return BLMM_OK;
}
}  // namespace

int blmm_null_h2_brent(blmm_ctx* ctx, const blmm_opts* opts, const double* Y0, int64_t n, int64_t m, const double* Z0,
                       int64_t c, const double* lambda, double* h2_out, double* sigma2_out, double* ell_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!opts || !Y0 || !Z0 || !lambda || !h2_out) return fail(ctx, BLMM_ERR_INVALID, "null_h2_brent: NULL buffer");
  BLMM_HIP(hipSetDevice(ctx->device));
  Pipe P;
  int rc = upload_rotated(ctx, Y0, n, m, Z0, c, nullptr, 0, lambda, P);
  if (rc) return rc;
  const NullModel nm = null_model(P, opts);
  if ((rc = ensure(ctx, ctx->h2, sizeof(double) * m))) return rc;
  if ((rc = ensure(ctx, ctx->sig2, sizeof(double) * m))) return rc;
  if ((rc = ensure(ctx, ctx->ell, sizeof(double) * m))) return rc;
  if ((rc = launch_brent(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, ptr<double>(ctx->h2), ptr<double>(ctx->sig2), ptr<double>(ctx->ell), P.stat))) return rc;
  BLMM_HIP(hipMemcpyAsync(h2_out, ctx->h2.p, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  if (sigma2_out) BLMM_HIP(hipMemcpyAsync(sigma2_out, ctx->sig2.p, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  if (ell_out) BLMM_HIP(hipMemcpyAsync(ell_out, ctx->ell.p, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return finish_status(ctx, status, nullptr);
}

int blmm_null_loglik_grid(blmm_ctx* ctx, const blmm_opts* opts, const double* Y0, int64_t n, int64_t m, const double* Z0,
                          int64_t c, const double* lambda, const double* h2_grid, int64_t ngrid, double* Ell_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!opts || !Y0 || !Z0 || !lambda || !Ell_out) return fail(ctx, BLMM_ERR_INVALID, "null_loglik_grid: NULL buffer");
  BLMM_HIP(hipSetDevice(ctx->device));
  double* dgrid = nullptr;
  int rc = grid_to_device(ctx, h2_grid, ngrid, &dgrid);
  if (rc) return rc;
  Pipe P;
  if ((rc = upload_rotated(ctx, Y0, n, m, Z0, c, nullptr, 0, lambda, P))) return rc;
  const NullModel nm = null_model(P, opts);
  if ((rc = ensure(ctx, ctx->EllTab, sizeof(double) * (size_t)ngrid * m))) return rc;
  if ((rc = launch_loglik_grid(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, dgrid, (int)ngrid, ptr<double>(ctx->EllTab), nullptr, nullptr, P.stat))) return rc;
  BLMM_HIP(hipMemcpyAsync(Ell_out, ctx->EllTab.p, sizeof(double) * (size_t)ngrid * m, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return finish_status(ctx, status, nullptr);
}

int blmm_weighted_liteqtl(blmm_ctx* ctx, const double* Y0, int64_t n, int64_t m, const double* X0, int64_t c, int64_t p,
                          const double* lambda, double hsq, double* LOD_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!Y0 || !X0 || !lambda || !LOD_out || p < 1) return fail(ctx, BLMM_ERR_INVALID, "weighted_liteqtl: bad arguments");
  if (std::isinf(hsq / (1.0 - hsq))) return fail(ctx, BLMM_ERR_H2_ONE, "Heritability of 1 is not allowed.");
  BLMM_HIP(hipSetDevice(ctx->device));
  Pipe P;
  int rc = upload_rotated(ctx, Y0, n, m, X0, c, X0 + (size_t)n * c, p, lambda, P);
  if (rc) return rc;
  blmm_opts o; blmm_default_opts(&o);
  const NullModel nm = null_model(P, &o);
  if ((rc = ensure(ctx, ctx->h2, sizeof(double) * m))) return rc;
  if ((rc = fill(ctx, ptr<double>(ctx->h2), m, hsq))) return rc;
  if ((rc = ensure(ctx, ctx->panels, sizeof(double) * (size_t)P.npad * P.ldy))) return rc;
  if ((rc = launch_panels(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, ptr<double>(ctx->h2), 0, ptr<double>(ctx->panels), P.ldy, P.stat))) return rc;
  if ((rc = ensure(ctx, ctx->isx, sizeof(double) * (size_t)P.ldx))) return rc;
  if ((rc = launch_isx(ctx, nm, P.Xt, P.ldx, p, P.Z0, P.lam, ptr<double>(ctx->h2), 1, ptr<double>(ctx->isx), P.ldx, P.stat))) return rc;
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * (size_t)p * m))) return rc;
  ScanArgs a = scan_args(ctx, P, ptr<double>(ctx->panels), P.ldy, ptr<double>(ctx->outL), p, m);
  a.isx = ptr<double>(ctx->isx); a.ld_isx = P.ldx;
  if ((rc = launch_scan_table(ctx, a))) return rc;
  BLMM_HIP(hipMemcpyAsync(LOD_out, ctx->outL.p, sizeof(double) * (size_t)p * m, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return finish_status(ctx, status, nullptr);
}

int blmm_liteqtl_given_h2(blmm_ctx* ctx, const double* Y0, int64_t n, int64_t m, const double* X0, int64_t c, int64_t p,
                          const double* lambda, const double* h2, double* LOD_out, blmm_status* status) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!Y0 || !X0 || !lambda || !h2 || !LOD_out || p < 1) return fail(ctx, BLMM_ERR_INVALID, "liteqtl_given_h2: bad arguments");
  for (int64_t j = 0; j < m; ++j)
    if (std::isinf(h2[j] / (1.0 - h2[j]))) return fail(ctx, BLMM_ERR_H2_ONE, "Heritability of 1 is not allowed.");
  BLMM_HIP(hipSetDevice(ctx->device));
  Pipe P;
  int rc = upload_rotated(ctx, Y0, n, m, X0, c, X0 + (size_t)n * c, p, lambda, P);
  if (rc) return rc;
  blmm_opts o; blmm_default_opts(&o);
  const NullModel nm = null_model(P, &o);
  if ((rc = ensure(ctx, ctx->h2, sizeof(double) * m))) return rc;
  BLMM_HIP(hipMemcpyAsync(ctx->h2.p, h2, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * (size_t)p * m))) return rc;
  // the same kernel choice as bulkscan(method = null-exact): low-rank weights form with its residual guard unless
  // BLMM_EXACT=full, c = 4 or n beyond the basis kernel
  if (!exact_full(ctx) && P.c <= 3 && n <= 6000) {
    Timer tm(ctx);
    if ((rc = lr_begin(ctx, P, /*wbasis_started*/ false))) return rc;
    if ((rc = lr_finish(ctx, P, nm, ptr<double>(ctx->h2), ptr<double>(ctx->outL), p, tm))) return rc;
  } else {
    if ((rc = ensure(ctx, ctx->panels, sizeof(double) * (size_t)(2 + P.c) * P.npad * P.ldy))) return rc;
    if ((rc = launch_panels(ctx, nm, P.Yt, P.ldy, m, P.Z0, P.lam, ptr<double>(ctx->h2), 1, ptr<double>(ctx->panels), P.ldy, P.stat))) return rc;
    ScanArgs a = scan_args(ctx, P, ptr<double>(ctx->panels), P.ldy, ptr<double>(ctx->outL), p, m);
    if ((rc = launch_scan_exact(ctx, a, P.c))) return rc;
    if ((rc = illcond_rescan(ctx, P, nm, m, ptr<double>(ctx->h2), ptr<double>(ctx->outL), p))) return rc;
  }
  BLMM_HIP(hipMemcpyAsync(LOD_out, ctx->outL.p, sizeof(double) * (size_t)p * m, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return finish_status(ctx, status, nullptr);
}

}  // extern "C"
