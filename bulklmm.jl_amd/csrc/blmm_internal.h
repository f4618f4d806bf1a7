// Internal declarations shared by the HIP translation units of libbulklmm_hip.so.
// Not part of the public ABI (that is include/bulklmm_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "../../include/bulklmm_hip.h"

namespace blmm {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int CMAX = 32;         // null covariates (incl. intercept) the library takes: 1 .. CTPL through kernels that are templates over c,
constexpr int CTPL = 8;          // CTPL + 1 .. CMAX through the run-time-c kernels of kernels_dyn.hip (the reference has no cap: src/wls.jl:27-60)
constexpr int CFAST = 4;         // ... with the tuned forms (LDS-resident evaluators, single-pass exact scan); beyond: the generic evaluators and a covariate-chunked scan
#define BLMM_C_ERR "number of null covariates (incl. intercept) must be 1..32"
// one `case C: M(C); break;` per instantiated covariate count (1 .. CTPL)
#define BLMM_FOR_EACH_C(M) \
  case 1: M(1); break; case 2: M(2); break; case 3: M(3); break; case 4: M(4); break; \
  case 5: M(5); break; case 6: M(6); break; case 7: M(7); break; case 8: M(8); break;
constexpr int TILE_T = 64;       // traits per workgroup tile of the scan kernels
constexpr int TILE_I = 128;      // markers per workgroup tile of the scan kernels
constexpr int NSTAT = 64;        // device status counters ([24 + 20 r ..]: per panel region r, 8 counts of traits per weight-basis segment and the 9 placement cursors of k_lr_classify; [6],[7]: eigensolver clocks, [8]: weight-basis rank, [9]: its residual, [10]: traits re-scanned full rank, [11]: eigensolver abort code, [12..15]: shared-weights traits / the others of the two panel regions (k_lr_classify), [16..18]: StatIdx below)

enum StatIdx { ST_NEG_EIG = 0, ST_NONPOS_W = 1, ST_ZERO_NORM = 2, ST_NAN_LOD = 3, ST_BRENT_MAXIT = 4, ST_JACOBI_SWEEPS = 5,
               ST_H2_BOUNDARY = 16, ST_H2_MULTIMODAL = 17, ST_ILLCOND = 18,
               ST_EIG_FAST = 19 /* 1: the fast eigen path ran */, ST_EIG_BAD = 20 /* its largest check / bound, bits of a double: accepted up to 1.0 */,
               ST_BRENT_CNT = 21 /* [21..22]: hand-over counter of the split h2 search (kernels_prep.hip: launch_brent_t) */,
               ST_EIG_DONE = 23 /* workgroups of k_jacobi_lds that have finished (the last one does the post-eigen work) */ };

inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

// BLMM_* environment switches are DEVELOPER switches (A/B timing, diagnostics): the library reads them only when BLMM_DEV_ENV=1 is
// set as well, so that a caller's environment can never silently select another code path.  What changes the arithmetic of a
// result is a property of the context instead: blmm_set_tuning (include/bulklmm_hip.h); under BLMM_DEV_ENV=1 the old variable
// names still override it (tools/*.sh).
inline const char* dev_env(const char* name) {
  const char* on = getenv("BLMM_DEV_ENV");
  return (on && on[0] == '1') ? getenv(name) : nullptr;
}
// blmm_set_tuning / blmm_get_tuning (keys = the field names)
struct Tuning {
  double lr_tol = 1e-13;        // residual |w - QQ'w| / |w| above which a trait's column is re-scanned full rank; tolerance of the shared-weights class (0: every trait flagged, class empty)
  double illcond_rho = 1e-4;    // pivot-share threshold of the conditioning guard (0: off; 2: every trait with c >= 2 re-scanned)
  int exact_full_rank = 0;      // 1: null-exact through the full-rank kernel k_scan<NX = 1 + c> instead of the low-rank weights form
  int pval_libm = 0;            // 1: -log10 p through erfc / erfcx / log instead of the bucketed polynomials (df = 1)
  int pval_fused = 1;           // 0: output_pvals as a column pass over the finished L instead of the scan epilogues
  int lr_segments = 0;          // 0: default (six segments of the heritability axis for n <= 80); 1: one weight basis; 2..8: that many equal segments
  int lr_shared = 1;            // 0: no shared-weights class (every trait through the rank-R form)
  int lr_split = -1;            // -1: split the h2 search into two panel regions from 8192 traits on; 0 / 1: never / always
  int f32_rotation = 1;         // fp32 permutation path (blmm_scan_perms_f32, c <= 3): 1 = rotate G on the fp32 matrix cores straight into k_scan_f32's operand layout; 0 = fp64 rotation + conversion (round 3)
  int eigen_solver = 0;         // 0: by n (fast path + Jacobi up to 124, tridiagonalisation + divide and conquer beyond); 1: Jacobi; 2: divide and conquer
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// What the stages of a call hand to one another: shapes, the rotated operands and the null model's rotated covariates.
struct Pipe {
  int n = 0, c = 0, npad = 0, ldr = 0;
  int64_t m = 0, p = 0, ldy = 0, ldx = 0;
  double *Yt = nullptr, *Xt = nullptr, *Z0 = nullptr, *lam = nullptr;
  int64_t* stat = nullptr;
  bool big = false;                // n beyond the LDS Jacobi: the call ends with k_sticky
  bool xt_side = false;            // the marker rotation was enqueued on the side stream (in front of the weight basis): Xt is ordered there
};

// Reduce-in-epilogue output of the scan kernels (blmm_bulkscan_reduced; SURVEY.md N1: "the matrix never has to leave HBM" -- here
// it is never WRITTEN): instead of a lane's four LODs of one trait going to L, the 16 lanes of the MFMA row reduce them to the
// trait's (maximum, marker) over the wave's 64 markers -> pmax / parg [slot = first marker / 64][ldm], finished by k_red_final;
// and every LOD > thr becomes a (marker, trait, LOD) triplet behind a device counter.  Same values, same tie rule (lowest marker)
// and same NaN rule (never the maximum, never > thr) as k_colmax / k_threshold on a stored L: the results are bit-identical.
struct RedArgs {
  double* pmax = nullptr; int* parg = nullptr; int64_t ldm = 0;
  int want_trip = 0; double thr = 0.0; int64_t cap = 0;
  int32_t* ti = nullptr; int32_t* tj = nullptr; double* tl = nullptr; unsigned long long* cnt = nullptr;
};
}  // namespace blmm

namespace blmm { struct HostStage; }

struct blmm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool timing = false;
  std::string err;
  // grow-only workspace
  blmm::DevBuf Ks, V, lam, U, Zs, Z0, Rp, Yt, Xt, panels, iyy, h2, h2idx, sig2, ell, isx, stat, gridd, misc, EllTab,
      inY, inG, inK, inCov, inW, outL, outH2, tmpA, tmpB, tmpC, perm, r0, altbuf, logtab, lraw, wbQ, wbW, wbRk, lrT, lrC, lrL, lrFlag, lrPart, lrPerm, lrDen0, eigW, xf32, pf32, brSt, brList, illList, qrSlab, lodtab, dynFac, pvtab, outP, redbuf, redtrip, altC, rf32, btG;
  // event sets: one per timed call since the last blmm_read_timings (grown on demand, reused afterwards)
  struct EvSet { hipEvent_t e[8]; int n; };
  std::vector<EvSet> evsets;
  size_t ev_used = 0;
  // side stream: work that only depends on the eigenvalues / rotated markers runs beside the per-trait Brent search
  hipStream_t side = nullptr, side2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_xt = nullptr, ev_b1 = nullptr, ev_b2 = nullptr, ev_q = nullptr, ev_m = nullptr, ev_wb = nullptr;
  // An event to be recorded BY the next launch that honours it (BLMM_LAUNCH_STOP: hipExtLaunchKernel's stop event = the kernel's own
  // completion signal) instead of by a marker packet behind that kernel: a hipEventRecord between two dependent kernels of the main
  // stream cost 10-12 us of its critical path (profiles/r04_timeline_notiming_*.txt).  stop_event_used: the launch took it.
  hipEvent_t stop_event_next = nullptr; bool stop_event_used = false;
  bool wb_on_side2 = false;            // start_wbasis: the weight basis went to the second side stream (lr_begin follows it there)
  // Host entry points (blmm_bulkscan, blmm_bulkscan_reduced): K, the covariates and the weights go up first, the eigen phase is
  // queued, and only then Y and G are copied -- on their own stream, beside the eigen kernels (0.23 ms at n = 79 that the caller
  // used to wait for behind 27 MB of uploads).  up_pending: upload_bulk_inputs left them for prepare(); in_wait: the copies are
  // in flight, whoever reads inY / inG waits for ev_in first.
  hipStream_t copy = nullptr; hipEvent_t ev_in = nullptr, ev_inY = nullptr;   // ev_inY: the traits are there (the main stream's need), ev_in: the markers too
  bool up_pending = false, in_wait = false;
  const void* up_src[2] = {nullptr, nullptr}; void* up_dst[2] = {nullptr, nullptr}; size_t up_bytes[2] = {0, 0};
  int num_cus = 0;                 // multiProcessorCount of the device (bounds every co-resident grid)
  // sticky device-side abort word in pinned, device-mapped host memory: a kernel that gives up (bounded spin of the
  // multi-workgroup weight-basis kernel) is reported by the NEXT API call / blmm_synchronize even when the failing
  // call was made without a blmm_status (blmm_api.hip: check_sticky)
  volatile int64_t* hflag = nullptr;
  // the LOD matrix of the last host-pointer call, still resident in the workspace (kernels_post.hip: blmm_last_*)
  const double* last_L = nullptr; int64_t last_p = 0, last_m = 0; bool last_f32 = false;
  // blmm_set_log10p_output: the next bulkscan call also writes -log10 p (pv_out == nullptr: into outP, dense); pv_cur is set
  // while that call's scan kernels run (blmm_api.hip: scan_args).  last_P: the matrix that call left (blmm_last_log10p).
  bool pv_armed = false; double* pv_out = nullptr; int64_t pv_ld = 0, pv_df = 1;
  double* pv_cur = nullptr; int64_t pv_cur_ld = 0;
  const double* last_P = nullptr; int64_t last_P_ld = 0, last_P_df = 0;
  // blmm_bulkscan_reduced: set while that call's scan kernels run (blmm_api.hip: scan_args); last_reduced_route: 1 = the
  // reduce-in-epilogue kernels stood, 2 = through a resident L (no fused instantiation, or a trait needed a re-scan)
  blmm::RedArgs red_cur; int last_reduced_route = 0;
  blmm::Tuning tune;                   // blmm_set_tuning
  bool perm_ready = false; int perm_ready_n = 0; int64_t perm_ready_nperms = 0; uint64_t perm_ready_seed = 0;   // launch_perm_gen -> launch_perm_panel
  int64_t lr_last_ldq = 0, lr_last_m = 0;   // panel-region width / trait count of the last low-rank null-exact scan (blmm_lowrank_columns)
  blmm::Pipe prep; bool prep_valid = false;   // state left by blmm_prepare_dev for blmm_rotate_block_dev / blmm_bulkscan_prerotated_dev
  bool brent_cnt_used = false;         // the current call has run a split h2 search already (its counter in the status block is spent)
  bool audit_ran = false;              // the current call ran the BLMM_FLAG_H2_AUDIT pass (finish_status: n_h2_multimodal, else -1)
  int eig_plan_n = -1;                 // n whose merge tree sits in eigW (kernels_eig.hip)
  blmm::HostStage* hstage = nullptr;   // pinned staging ring + copy threads of the host-pointer entry points (host_path.hip)
};

namespace blmm {

// ---- error helpers -------------------------------------------------------------------------------
int fail(blmm_ctx* ctx, int code, const std::string& msg);
#define BLMM_HIP(call)                                                                              \
  do {                                                                                              \
    hipError_t e__ = (call);                                                                        \
    if (e__ != hipSuccess)                                                                          \
      return blmm::fail(ctx, BLMM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));     \
  } while (0)

int ensure(blmm_ctx* ctx, DevBuf& b, size_t bytes);
// Kernels whose workgroups meet at a grid barrier (k_sytrd, k_wbasis_mw) need ALL their workgroups resident at once.  Two of
// them launched from different contexts / streams on one device could each hold part of the CUs and starve the other (their
// bounded spins would then fail the calls).  A GridKernelGuard makes the stream wait for the previous such kernel on that
// device and, after the launch, record() marks this one: they run one after the other on the device, with no host
// synchronisation.  The guard holds the device's mutex from the wait to its destruction, so that wait -> launch -> record is
// one step against the other contexts' threads (blmm_bulkscan_multi with repeated device ids drives several contexts of one
// device from different threads; with begin / end as two separately locked calls both could pass the wait before either
// recorded).
struct GridKernelGuard {
  blmm_ctx* ctx; int dev; int rc = 0;
  explicit GridKernelGuard(blmm_ctx* c);
  int record();
  ~GridKernelGuard();
  GridKernelGuard(const GridKernelGuard&) = delete;
  GridKernelGuard& operator=(const GridKernelGuard&) = delete;
};
// host_path.hip: device -> caller memory in stream order; returns when the bytes are in place
int copy_to_host(blmm_ctx* ctx, void* dst, const void* dsrc, size_t bytes);
void destroy_host_stage(HostStage* hs);
template <typename T>
inline T* ptr(DevBuf& b) { return reinterpret_cast<T*>(b.p); }

// ---- kernel launchers (kernels_prep.hip) ----------------------------------------------------------
struct Design {
  int n = 0, c = 0, npad = 0;   // npad: n rounded up to 4 (K dimension of the f64 MFMA)
  int ldr = 0;                  // leading dimension of Rp (npad rounded up to 16)
};

// Builds Zs = wd .* [1 Covar] (n x c) and Ks = wd_i wd_j K_ij (n x n).
int launch_design(blmm_ctx* ctx, const double* dK, const double* dCovar, int ncov, int add_intercept,
                  const double* dweights, int n, double* Ks, double* Zs);
// One-sided Jacobi eigen-decomposition of the symmetric n x n matrix in A (destroyed); V gets the eigenvectors
// (unsorted), then post_eigen sorts/derives everything the rotation needs.
struct PostEigenArgs;
int launch_jacobi(blmm_ctx* ctx, double* A, double* V, int n, double* lraw, int64_t* stat, const PostEigenArgs* pe = nullptr, bool* fused = nullptr);
// ... with launch_post_eigen's work in the tail of the same launch where its LDS fits (*fused)
int launch_jacobi_post(blmm_ctx* ctx, double* A, double* V, int n, double* lraw, int64_t* stat, const double* Zs, const double* dweights,
                       int c, int npad, int ldr, int decomp, int centered, double* lam, double* U, double* Z0, double* Rp, bool* fused);
// kernels_eig.hip: tridiagonalisation + divide and conquer for n beyond the LDS Jacobi; A is not modified
int launch_eig_dc(blmm_ctx* ctx, const double* A, int n, double* lraw, double* evec, int64_t* stat);
int eig_dc_max_n(const blmm_ctx* ctx);
int launch_eig_fast(blmm_ctx* ctx, const double* A, int n, double* lraw, double* evec, int64_t* stat);
int eig_fast_max_n();
int jacobi_lds_max_n();
// lambda (ascending, or |lambda| descending for svd), U sorted, Z0 = U' Zs, Rp = (centered ? Q U' Wd : U' Wd)'
int launch_post_eigen(blmm_ctx* ctx, const double* lraw, const double* V, const double* Zs, const double* dweights, int n,
                      int c, int npad, int ldr, int decomp, int centered, double* lam, double* U, double* Z0, double* Rp,
                      int64_t* stat);
// Out (row-major, npad x ldo) = R * In  (In column-major n x ncols); pads with zeros up to ncols_pad / npad.
int launch_rotate(blmm_ctx* ctx, const double* Rp, int ldr, int n, int npad, const double* In, int64_t ncols,
                  double* Out, int64_t ldo, int64_t ncols_pad);
// row-major (npad x ld) -> column-major (n x ncols)
int launch_untranspose(blmm_ctx* ctx, const double* In, int64_t ld, int n, int64_t ncols, double* Out);

struct NullModel {
  int n, c, npad, reml, optim_interval;
  double prior_a, prior_b;
};
// per-trait Brent h2 (fitlmm) from centred/rotated Yt; outputs m each (sigma2/ell may be null).
// phase 0: the whole search.  phase 1: the first kernel only -- when sp->active comes back true, the traits with
// fin[j] == 1 are final and the other `*cnt` traits (list[]) are finished by a later phase-2 call with the same
// arguments (possibly on another stream); when false the search is already complete.
struct BrentSplit {
  bool active = false;
  const int* fin = nullptr;
  const int* list = nullptr;
  const unsigned int* cnt = nullptr;
};
int launch_brent(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                 const double* lam, double* h2, double* sigma2, double* ell, int64_t* stat, int phase = 0,
                 BrentSplit* sp = nullptr);
// scan_alt: per-marker fitlmm on [Z0 x_i] and the LOD against the null model (the trait is column 0 of Yt)
int launch_alt_brent(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Xt, int64_t ldx, int64_t p,
                     const double* Z0, const double* lam, const double* h2null, int true_w, double* lod, double* h2each,
                     int64_t* stat, int64_t m = 1, int64_t ldL = 0, int64_t ldH = 0);
// Ell[g, j] for every grid point and first-argmax; EllTab (ngrid x m, ld = ngrid) may be null
int launch_loglik_grid(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                       const double* lam, const double* grid_dev, int ngrid, double* EllTab, int* h2idx, double* h2,
                       int64_t* stat);
// BLMM_FLAG_H2_AUDIT: counts the traits whose grid profile (EllTab, ngrid x m) has two or more local maxima
int launch_h2_audit(blmm_ctx* ctx, const double* EllTab, int ngrid, int64_t m, int64_t* stat);
// A-side panels for the scan kernels from per-trait h2: panel 0 = w.*resid/sqrt(yy); if full: panel 1 = w,
// panels 2..1+c = w .* (Z0 Linv')_q.  panels: [np][npad][ldp]
int launch_panels(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                  const double* lam, const double* h2, int full, double* panels, int64_t ldp, int64_t* stat,
                  const double* gridv = nullptr, int ngrid = 0);   // gridv (c <= CTPL, !full): panel g = every trait at h2 = gridv[g], one launch
// isx[g][i] = 1/||P_g sqrt(w_g) x_i|| for every grid point
int launch_isx(blmm_ctx* ctx, const NullModel& nm, const double* Xt, int64_t ldx, int64_t p, const double* Z0,
               const double* lam, const double* grid_dev, int ngrid, double* isx, int64_t ld_isx, int64_t* stat);
int launch_kinship(blmm_ctx* ctx, const double* dG, int64_t n, int64_t p, double* dK, double* partial);
int launch_colmax(blmm_ctx* ctx, const double* L, int64_t p, int64_t m, int64_t ldL, double* mx, int64_t* arg);
// kernels_post.hip
int launch_lod2log10p(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, int df, double* dP, int64_t ldP);
// permutation panel: column b = sqrt(w) .* P_w( pi_b(r0) ) / ||r0||  etc.  (see kernels_prep.hip)
int launch_perm_gen(blmm_ctx* ctx, int n, int64_t nperms, uint64_t seed);   // the library's own permutation indices into ctx->perm, ahead of launch_perm_panel
int launch_perm_panel(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Z0,
                      const double* lam, const double* h2, const int32_t* perm_idx, int64_t nperms, uint64_t seed,
                      int orig, double* panel, int64_t ldp, int64_t* stat);

// ---- kernel launchers (kernels_scan.hip) ----------------------------------------------------------
struct ScanArgs {
  const double* Xt; int64_t ldx;           // markers, row-major npad x ldx (unweighted, centred, rotated)
  const double* P; int64_t ldp; int64_t pstride;  // A-side panels [np][npad][ldp]
  int ks;                                  // npad / 4
  int n;                                   // sample size (LOD scale = -n/2)
  int64_t p, m;
  double* L; int64_t ldL;
  const double* isx; int64_t ld_isx; const int* bin;  // table mode
  // permuted-column mode of the table kernel (the shared-weights class of the low-rank form): panel column -> trait, the
  // region's first column, and the device count of the class (columns [col0, col0 + *count))
  const int* perm = nullptr; int64_t col0 = 0; const int64_t* count = nullptr;
  int c = 1;                               // null covariates (exact mode beyond CFAST: panels 2 + CFAST .. 1 + c are folded in chunks)
  const double* logtab;                    // device copy of log_table.h
  const double* lodtab;                    // ... of its directly indexed LOD table (fastmath.h: fast_lod5)
  double lodc[6];                          // {scale = -n/2, c1..c5}: the LOD polynomial of fast_lod5, computed on the host so that
                                           // the kernels hold it in SGPRs (gfx950 has no scalar fp64 arithmetic: derived in the
                                           // kernel the five coefficients cost 10 VGPRs in kernels at the register limit)
  int64_t* stat;
  // optional second output of the epilogue: -log10 p, one degree of freedom (blmm_set_log10p_output); pvtab = device copy of
  // pval_table.h.  Kernels without the fused form (alt-grid, fp32 permutations, the rare per-trait re-scans) leave Pv to a
  // column pass over the finished L (launch_lod2log10p / launch_pv_list).
  double* Pv = nullptr; int64_t ldPv = 0; const double* pvtab = nullptr;
  RedArgs red;                             // red.pmax != nullptr: the reduce-in-epilogue instantiation runs and L is not touched
};
// kernels_post.hip: threshold triplets of a resident L; the second pass of the reduce-in-epilogue scan (RedArgs partials -> per trait)
int launch_threshold(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, double thr, int64_t cap,
                     int32_t* di, int32_t* dj, double* dlod, int64_t* dcount);
int launch_red_final(blmm_ctx* ctx, const RedArgs& r, int nslot, int64_t m, double* mx, int64_t* arg);
int launch_scan_exact(blmm_ctx* ctx, const ScanArgs& a, int c);
// A region of the panel arrays of the low-rank form: columns [col0, col0 + ncol), the shared-weights class at its front
// (counts[0] traits) and the other class at its back (counts[1]); counts live on the device.
constexpr int LR_TILE = 64;   // a multiple of every trait-tile width of k_scan_lr (32 * MB); regions are multiples of it
struct LrRegion {
  int64_t col0 = 0, ncol = 0;
  int64_t* counts = nullptr;        // {shared-weights traits, columns of the other class (every segment's run rounded up to LR_TILE)}
  int64_t* segcnt = nullptr;        // [LR_SEG_MAX] traits of the other class per weight-basis segment; segcnt + 8: the placement cursors {shared, segment 0, ..}
};
// Segments of the heritability axis, each with its OWN weight basis (kernels_lowrank.hip): the family { w(h2) : h2 in a segment }
// has a far smaller numerical rank than the whole family (BXD spectrum: 11-12 per segment of the six below against 23), so the
// rank-R phase of k_scan_lr runs 3 K steps where the single basis needs 6.  The other class of a region is laid out segment by
// segment from the back of the region, every segment's run rounded up to LR_TILE columns: a tile of the scan never mixes segments.
constexpr int LR_SEG_MAX = 8;
struct LrSeg {
  int S = 1;
  double edge[LR_SEG_MAX + 1] = {0.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0};   // segment s: edge[s] <= h2 < edge[s + 1] (the last one takes the rest)
};
// columns of the other class in front of segment s's run, counted from the END of the region, and the run's width
__host__ __device__ inline int64_t lr_seg_width(int64_t cnt) { return (cnt + LR_TILE - 1) / LR_TILE * LR_TILE; }
// segment of the column `dist` columns before the region's end (dist = 0: the last column); S - 1 beyond every run (padding)
__host__ __device__ inline int lr_seg_of(int64_t dist, const int64_t* segcnt, int S) {
  for (int s = 0; s + 1 < S; ++s) {
    const int64_t w = lr_seg_width(segcnt[s]);
    if (dist < w) return s;
    dist -= w;
  }
  return S - 1;
}
struct LrArgs {
  ScanArgs s;                       // s.P = panel 0 only
  const double* Cp;                 // weight-basis coefficients [4*KR][ldp]
  const double* T; int64_t tstride; // marker-side basis products [1+c][4*KR][ldx]
  const double* Ls;                 // packed L_j^-1 [c(c+1)/2][ldp]
  const int* rk;                    // {R, KR, -, -} per segment on the device
  LrSeg seg;                        // T, Q of segment s: s * (1 + c) * tstride, s * npad * n doubles further on
  const int* perm;                  // panel column -> trait (k_lr_classify; -1: padding)
  LrRegion rg;                      // the region of the panel arrays this launch scans
  const double* den0;               // the shared-weights class's 1/sqrt(Sxx - |u|^2), per marker (= isx of the unweighted model)
  int skip_shared;                  // 1: the class's tiles were scanned by launch_scan_shared
  int c;
};
int launch_scan_lr(blmm_ctx* ctx, const LrArgs& la);
int launch_lr_resid(blmm_ctx* ctx, const NullModel& nm, int64_t m, double tol, const double* lam, const double* h2,
                    const double* Q, const int* rk, const LrSeg& seg, const int* perm, const LrRegion& rg, const double* Cp, int64_t ldp,
                    int* flag_list, double* part, int64_t* stat);
int launch_scan_fix(blmm_ctx* ctx, const NullModel& nm, const double* Xt, int64_t ldx, int64_t p, const double* P0,
                    const double* Ls, int64_t ldp, const double* Z0, const double* lam, const double* h2,
                    const int* flag_list, const int* perm, double* L, int64_t ldL, int64_t* stat);
// shared-weights class: column order of the panels (perm, info) and the per-marker denominators of the unweighted model
int launch_lr_classify(blmm_ctx* ctx, int n, int64_t m, double tol, const double* lam, const double* h2, const int* fin,
                       const int* list, const unsigned int* list_cnt, int* perm, const LrRegion& rg, const LrSeg& seg);
// kernels_scan_f32.hip: fp32 permutation LOD kernel and the fp64 k-major -> fp32 fragment-major conversion
int launch_cvt_f32(blmm_ctx* ctx, const double* M, int64_t ld_in, int rows_valid, int64_t cols_valid, float* F,
                   int64_t ld_out, int kblocks);
// fp32 rotation of the fp32 permutation path: XF (fragment-major) = R G from the caller's fp64 column-major G; the marker norms
// from XF; the original trait's fp64 LOD from G itself (kernels_scan_f32.hip, kernels_prep.hip)
int launch_rotate_f32(blmm_ctx* ctx, const double* Rp, int ldr, int n, int npad, const double* dG, int64_t p, float* RF, float* XF, int64_t ldxf,
                      const double* a0, int64_t lda, double* v_work /* n rounded up to 16 */, double* num /* p: g_i' R'a0 in fp64 */);
int launch_lod_from_num(blmm_ctx* ctx, const double* num, const double* isx, int n, int64_t p, double* lod, int64_t* stat);
int launch_isx_f32(blmm_ctx* ctx, const NullModel& nm, const float* XF, int64_t ldxf, int64_t p, const double* Z0, const double* lam,
                   const double* h2_dev, double* isx, int64_t ld_isx, int64_t* stat);
int launch_scan_f32(blmm_ctx* ctx, const float* XF, int64_t ldxf, const float* PF, int64_t ldpf, int npad, int n,
                    int64_t p, int64_t m, const double* isx, float* L, int64_t ldL, int64_t* stat);
// kernels_dyn.hip: run-time covariate counts (c = CTPL + 1 .. CMAX): the counterparts of launch_brent / launch_loglik_grid /
// launch_panels / launch_isx / launch_perm_panel, which route there
int launch_dyn_brent(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0, const double* lam,
                     double* h2, double* sigma2, double* ell, int64_t* stat);
// scan_alt's per-marker searches on the design [Z0 x_i] (c + 1 <= CMAX columns), the counterpart of launch_alt_brent
int launch_dyn_alt_brent(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Xt, int64_t ldx, int64_t p,
                         const double* Z0, const double* lam, const double* h2null, int true_w, double* lod, double* h2each,
                         int64_t* stat, int64_t m, int64_t ldL, int64_t ldH);
int launch_dyn_loglik_grid(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                           const double* lam, const double* grid_dev, int ngrid, double* EllTab, int* h2idx, double* h2, int64_t* stat);
int launch_dyn_panels(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0, const double* lam,
                      const double* h2, int full, double* panels, int64_t ldp, int64_t* stat);
int launch_dyn_isx(blmm_ctx* ctx, const NullModel& nm, const double* Xt, int64_t ldx, int64_t p, const double* Z0, const double* lam,
                   const double* grid_dev, int ngrid, double* isx, int64_t ld_isx, int64_t* stat);
int launch_dyn_perm(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Z0, const double* lam, const double* h2,
                    const int32_t* perm, int64_t ncols, int orig, double* r0, double* panel, int64_t ldp, int64_t* stat);
// kernels_dyn.hip: conditioning guard of the null-exact scan (c >= 2) and the QR-grade re-scan of the flagged traits
int launch_illcond_flag(blmm_ctx* ctx, const NullModel& nm, int64_t m, const double* Z0, const double* lam, const double* h2,
                        int* list, int64_t* stat);
int launch_scan_qr(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Xt, int64_t ldx, int64_t p,
                   const double* Z0, const double* lam, const double* h2, const int* list, double* L, int64_t ldL, int64_t* stat);
// kernels_lowrank.hip
// the segments a call with n individuals uses (one when the basis comes from the multi-workgroup / LDS kernels: n > 80)
LrSeg lr_segments(const blmm_ctx* ctx, int n);
int launch_wbasis(blmm_ctx* ctx, const double* lam, int n, int npad, const LrSeg& seg, double* Wk, double* Q, int* rk, int64_t* stat);
int launch_lr_tpanels(blmm_ctx* ctx, const double* Xt, int64_t ldx, int64_t p, int n, int c, int npad, const double* Z0,
                      const double* Q, const int* rk, const LrSeg& seg, double* T, int64_t tstride, double* den0);
int launch_lr_panels(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                     const double* lam, const double* h2, const double* Q, const int* rk, const LrSeg& seg, const int* perm, const LrRegion& rg,
                     double* P0, double* Cp, double* Ls, int64_t ldp, int64_t* stat);
int launch_scan_table(blmm_ctx* ctx, const ScanArgs& a);
// the shared-weights class of one panel region through the table kernel (isx = 1/sqrt(den0), one bin, stores through perm)
int launch_scan_shared(blmm_ctx* ctx, const ScanArgs& a);
struct AltArgs {
  ScanArgs s;
  int ngrid; const double* EllTab; /* ngrid x m */ const double* grid_dev; double* H2; int64_t ldH; int counter_quirk;
  const double* Ctab;    // exp(-(2/n)(Ell[g, j] - max_g Ell[g, j])), ngrid x m (launch_alt_ctab): the fold's monotone image of logL1
};
int launch_scan_alt(blmm_ctx* ctx, const AltArgs& a);
int launch_alt_ctab(blmm_ctx* ctx, const double* EllTab, int ngrid, int64_t m, int n, double* C);

}  // namespace blmm

// launch on ctx->stream; a pending ctx->stop_event_next is recorded by this kernel's completion (see blmm_ctx)
#define BLMM_LAUNCH_STOP(ctx_, kernel_, grid_, block_, lds_, ...)                                                          \
  do {                                                                                                                   \
    if ((ctx_)->stop_event_next) {                                                                                       \
      hipExtLaunchKernelGGL(kernel_, grid_, block_, (std::uint32_t)(lds_), (ctx_)->stream, nullptr, (ctx_)->stop_event_next, 0, __VA_ARGS__); \
      (ctx_)->stop_event_next = nullptr; (ctx_)->stop_event_used = true;                                                 \
    } else hipLaunchKernelGGL(kernel_, grid_, block_, lds_, (ctx_)->stream, __VA_ARGS__);                                 \
  } while (0)
