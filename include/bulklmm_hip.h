/*
 * bulklmm_hip.h -- C ABI of libbulklmm_hip.so: the MI355X (gfx950) bulkscan engine.
 *
 * This is the drop-in boundary for the `bulkscan` hot path of senresearch/BulkLMM.jl v1.2.0
 * (pure Julia; it has no FFI of its own, so the seam is placed under its exported API,
 * src/BulkLMM.jl:9-47).  A Julia host binds these entry points with `ccall`
 * (bulklmm.jl_amd/julia/BulkLMMHIP.jl); the Python host mirror
 * (bulklmm.jl_amd/api.py) binds the same symbols with ctypes.
 *
 * Conventions
 *   - every matrix is dense float64, column-major, leading dimension = row count unless an
 *     explicit `ld` argument is given (Julia Array{Float64,2} / NumPy order='F');
 *   - sizes are int64_t (Julia Int64);
 *   - the caller owns every buffer; the library never retains a caller pointer after return;
 *   - functions return 0 on success or a negative blmm_err code; blmm_last_error(ctx) gives the
 *     message (the reference's own message strings are used where the reference throws);
 *   - `*_dev` entry points take DEVICE pointers (HBM-resident operands, e.g. torch tensors) and
 *     enqueue on the context's stream without synchronising it; the un-suffixed entry points take
 *     HOST pointers, copy in/out and return when the outputs are complete in caller memory;
 *   - a blmm_ctx is bound to one GPU and is not thread-safe (one call at a time per ctx).
 */
#ifndef BULKLMM_HIP_H
#define BULKLMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLMM_VERSION 210 /* 0.2.3: L_out == NULL keeps the matrix in HBM (blmm_bulkscan, blmm_bulkscan_multi), blmm_last_lod_colmax /
                            blmm_last_lod_columns / blmm_last_dims, blmm_bulkscan_reduced[_dev] (no L at all), blmm_tuning;
                            205 (0.2.2): blmm_status.n_h2_boundary / n_h2_multimodal / n_illcond_rescan (appended), BLMM_FLAG_H2_AUDIT;
                            201: lowrank_shared, readers, blmm_scan_alt; 200: lowrank_fallback, BLMM_STREAM_NULL, multi-GPU */

typedef struct blmm_ctx blmm_ctx;

enum blmm_err {
  BLMM_OK = 0,
  BLMM_ERR_INVALID = -1,       /* bad argument */
  BLMM_ERR_DIM = -2,           /* "Dimension mismatch."                      src/transform_helpers.jl:9-11 */
  BLMM_ERR_H2_ONE = -3,        /* "Heritability of 1 is not allowed."        src/lmm.jl:19-21 */
  BLMM_ERR_DECOMP = -4,        /* "Please choose either `eigen` or `svd`..." src/transform_helpers.jl:51 */
  BLMM_ERR_METHOD = -5,        /* unknown bulkscan method                    src/bulkscan.jl:126-154 */
  BLMM_ERR_ONE_TRAIT = -6,     /* "Can only handle one trait."               src/scan.jl:496-498 */
  BLMM_ERR_NO_INTERCEPT = -7,  /* "Intercept has to be added when no other covariate is given." src/scan.jl:167-169 */
  BLMM_ERR_ZERO_NORM = -8,     /* "Dividing by zeros: the input vector can not contain any zeros!" src/util.jl:69-71 */
  BLMM_ERR_NPERMS = -9,        /* "The required number of permutations must be a positive integer." src/scan.jl:528-530 */
  BLMM_ERR_UNSUPPORTED = -10,  /* e.g. more null covariates than the kernels are instantiated for */
  BLMM_ERR_NO_DEVICE = -11,    /* no usable gfx950 device */
  BLMM_ERR_HIP = -12,          /* a HIP runtime call failed */
  BLMM_ERR_ALLOC = -13
};

enum blmm_method { BLMM_NULL_EXACT = 0, BLMM_NULL_GRID = 1, BLMM_ALT_GRID = 2 };
enum blmm_decomp { BLMM_EIGEN = 0, BLMM_SVD = 1 };

/* compat_flags bits */
#define BLMM_COMPAT_ALT_COUNTER 1 /* B2: h2_panel indexed by an improvement counter, src/bulkscan_helpers.jl:342-343 */
#define BLMM_COMPAT_ALT_TRUE_WEIGHTS 2 /* blmm_scan_alt: evaluate the closing log-likelihoods at makeweights(h2); the default
                                          restates src/scan.jl:431-436, which hands wls the square roots of the weights */
/* Not a compat switch but carried in the same word: null-exact only, opt-in diagnostic.  After the h2 search the profile
 * log-likelihood of EVERY trait is evaluated on the 16-point grid 0, 1/16, .., 15/16 and blmm_status.n_h2_multimodal counts
 * the traits whose grid profile has two or more local maxima.  Brent (src/gridbrent.jl:9-24, one run over [0, 1] when
 * optim_interval = 1) is a LOCAL method: on such a trait rounding-level differences decide which maximum a run ends in --
 * for the reference's arithmetic as much as for this library's -- so these are the traits whose h2 (and LOD column) may
 * legitimately differ between two correct implementations; raising optim_interval resolves them.  Costs one grid
 * log-likelihood pass (~0.06 ms at BXD size). */
#define BLMM_FLAG_H2_AUDIT 4

/* Mirrors the keyword arguments of bulkscan()/scan() 1:1 (src/bulkscan.jl:81-92, src/scan.jl:94-109).
 * `nb` and `nt_blas` (thread blocking knobs of the CPU reference) have no meaning here. */
typedef struct blmm_opts {
  int32_t method;         /* blmm_method; bulkscan(...; method=)           */
  int32_t reml;           /* reml::Bool                                     */
  int32_t add_intercept;  /* addIntercept::Bool (Covar given)               */
  int32_t decomp_scheme;  /* blmm_decomp; decomp_scheme::String             */
  int32_t optim_interval; /* optim_interval::Int64 (null-exact, scan)       */
  int32_t compat_flags;   /* BLMM_COMPAT_*                                  */
  double prior_variance;    /* prior_variance::Float64                      */
  double prior_sample_size; /* prior_sample_size::Float64                   */
} blmm_opts;

/* Counters behind the reference's warnings / exceptions, plus per-phase device timings (ms)
 * measured with HIP events when blmm_set_timing(ctx, 1) is on (0 otherwise). */
typedef struct blmm_status {
  int64_t n_neg_eig;       /* eigenvalues < -1e-7           -> warning, src/transform_helpers.jl:27-30 */
  int64_t n_nonpos_weight; /* weights <= 0                  -> warning, src/wls.jl:35-37               */
  int64_t n_zero_norm;     /* |column norm| <= eps          -> error,   src/util.jl:47-71              */
  int64_t n_nan_lod;       /* r^2 > 1 (DomainError in the reference, NaN here), src/bulkscan_helpers.jl:23 */
  int64_t n_brent_maxiter; /* traits whose Brent search hit 1000 iterations                            */
  int64_t jacobi_sweeps;   /* sweeps of the LDS Jacobi; 0: n <= 124 and the fast path's result stood, or n > 124 */
  int64_t jacobi_cycles;   /* shader cycles / 100 MHz ticks spent inside the eigensolver (diagnostic)  */
  int64_t jacobi_ticks_100mhz;
  int64_t lowrank_rank;    /* rank R of the weight-family basis used by the null-exact kernel (kernels_lowrank.hip) */
  int64_t lowrank_fallback;/* traits whose expansion residual exceeded 1e-13: their LOD columns were recomputed from
                              the full-length sums (k_scan_fix), so every returned LOD is either guarded or exact      */
  int64_t lowrank_shared;  /* traits whose weights are 1 to within the same tolerance (likelihood peaks at h2 = 0): their
                              denominators are the per-marker constants of the unweighted model, no basis needed          */
  double lowrank_resid;    /* largest relative residual |w_j - Q Q'w_j| / |w_j| over all traits of the rank-R class (before the
                              re-scan); the shared-weights class is bounded by its own criterion, the same tolerance         */
  double t_eigen_ms, t_rotate_ms, t_h2_ms, t_prep_ms, t_scan_ms, t_total_ms;
  int64_t n_h2_boundary;   /* null-exact / scan: traits whose h2 estimate sits on a boundary of [0, 1] (<= 1e-6 or >= 1 - 1e-6):
                              the likelihood is one-sided there and x_tol shrinks with x, so these are the long Brent runs  */
  int64_t n_h2_multimodal; /* BLMM_FLAG_H2_AUDIT: traits whose profile log-likelihood has >= 2 local maxima on the 16-point
                              grid (optimiser-sensitive: see the flag); -1 when the audit was not requested                  */
  int64_t n_illcond_rescan;/* traits whose weighted null design sqrt(w) .* Z0 is ill conditioned (Cholesky pivot ratio of
                              Z0'WZ0 above 1e4, i.e. cond(sqrt(W) Z0) > 100: h2 -> 1 with several covariates): their LOD
                              columns were recomputed with an orthogonalised (MGS2, QR-grade) projection, as the
                              reference's `resid` does by Householder QR (src/wls.jl:221-241)                               */
} blmm_status;

/* ---- library / context ------------------------------------------------------------------ */
int blmm_version(void);
int blmm_device_count(void);
/* Creates a context on HIP device `device_id`.
 *   hip_stream == NULL             : the library creates a PRIVATE non-blocking stream; nothing the caller enqueues on
 *                                    any other stream (the legacy default stream included) is ordered against the calls.
 *   hip_stream == BLMM_STREAM_NULL : adopt the legacy default ("null") stream, handle 0 -- what torch.cuda's default
 *                                    stream is.  The handle 0 itself cannot be passed because it reads as NULL.
 *   otherwise                      : a hipStream_t of the caller; every *_dev call enqueues on it. */
#define BLMM_STREAM_NULL ((void*)(intptr_t)-1)
int blmm_create(int device_id, void* hip_stream, blmm_ctx** out);
void blmm_destroy(blmm_ctx* ctx);
const char* blmm_last_error(const blmm_ctx* ctx);
const char* blmm_err_string(int code);
int blmm_set_stream(blmm_ctx* ctx, void* hip_stream);
/* on = 1: HIP events are recorded on the context's stream at every phase boundary of each bulkscan / scan call
 * (no synchronisation).  blmm_status then carries the LAST call's phase times, and blmm_read_timings() returns the
 * SUM over all calls since the previous read together with their count (it synchronises the stream). */
int blmm_set_timing(blmm_ctx* ctx, int on);
/* sums_ms[6] = {eigen, rotate, h2, prep, scan, total}; *ncalls = calls accumulated. */
int blmm_read_timings(blmm_ctx* ctx, double* sums_ms, int64_t* ncalls);
/* What the last null-exact call EXECUTED in its low-rank weights form (kernels_lowrank.hip; it synchronises the stream):
 * out[18] = {segments of the heritability axis with traits, traits of the shared-weights class (no basis: 2 n flop per test),
 * then per segment s: traits, rank R_s of its weight basis (2 (n + (1 + c) 4 ceil(R_s / 4)) flop per test)}.  A diagnostic for
 * benchmarks that price the executed arithmetic (bench.py); the reference has no counterpart. */
int blmm_lowrank_profile(blmm_ctx* ctx, int64_t* out);
/* Where each trait sat in that call's data-dependent panel layout (two regions split by the h2 search's hand-over; in each the
 * shared-weights class from the front, the weight-basis segments from the back): col_out[j] = panel column of trait j (-1: none),
 * *region_width = columns per region (region = col / width), counts_out[4] = {shared-weights traits, columns of the other class}
 * of region 0, then of region 1.  For tests that report the class / region of their worst entry. */
int blmm_lowrank_columns(blmm_ctx* ctx, int64_t m, int32_t* col_out, int64_t* region_width, int64_t* counts_out);
int blmm_synchronize(blmm_ctx* ctx);
/* ---- tuning: the switches that select another ARITHMETIC path are properties of the context (they were BLMM_* environment
 * variables up to 0.2.2; the environment is now read only under BLMM_DEV_ENV=1, for A/B timing by developers).  Keys and defaults:
 *   "lr_tol"          1e-13  relative residual of the weight-basis expansion above which a trait's LOD column is recomputed from
 *                            the full-length sums, and the tolerance of the shared-weights class (0: every trait re-scanned)
 *   "illcond_rho"     1e-4   pivot-share threshold of the conditioning guard (0: off; 2: every trait with >= 2 covariates re-scanned)
 *   "exact_full_rank" 0      1: null-exact through the full-rank kernel (2n(2+c) flop per test) instead of the low-rank weights form
 *   "pval_libm"       0      1: -log10 p through erfc / erfcx / log instead of the bucketed polynomials (chisq_df = 1)
 *   "pval_fused"      1      0: output_pvals as a pass over the finished L instead of a second output of the scan epilogues
 *   "lr_segments"     0      weight bases per heritability axis: 0 = default (six for n <= 80, else one), 1, or 2..8 equal segments
 *   "lr_shared"       1      0: no shared-weights class (traits with h2 = 0 go through the rank-R form like the others)
 *   "lr_split"        -1     split h2 search / two panel regions: -1 = from 8192 traits on, 0 never, 1 always
 *   "eigen_solver"    0      0 = by n; 1 = Jacobi; 2 = tridiagonalisation + divide and conquer
 *   "f32_rotation"    1      blmm_scan_perms_f32 with an intercept-only null model: 1 = the marker rotation runs on the fp32 matrix
 *                            cores as well; 0 = fp64 rotation, converted (0.2.2)
 *   "defaults"               (set only) every key back to its default
 * Every setting gives results within the library's stated tolerances; they exist for tests and for A/B measurements. */
int blmm_set_tuning(blmm_ctx* ctx, const char* key, double value);
int blmm_get_tuning(const blmm_ctx* ctx, const char* key, double* value);
void blmm_default_opts(blmm_opts* o); /* bulkscan() defaults: null-grid, ML, prior (1.0, 0.0), eigen */

/* ---- pinned host memory for the outputs of the host-pointer entry points ------------------------------------------
 * L is p x m doubles (2.08 GB at BXD size) and has to cross ONE PCIe link: into a pinned destination it moves at link
 * rate in one asynchronous copy; into pageable memory the library pipelines 32 MB pieces through its own pinned ring
 * and copies them out with a few host threads.  A Julia caller either wraps blmm_host_alloc memory (unsafe_wrap) or
 * registers the Array it already has; both are optional. */
int blmm_host_register(void* p, uint64_t bytes);
int blmm_host_unregister(void* p);
void* blmm_host_alloc(uint64_t bytes);
void blmm_host_free(void* p);

/* ---- readers for the file formats the reference reads (host code, no GPU needed) ----------------------------------
 * blmm_read_csv: numeric CSV; `skip_lines` leading lines are dropped, then of every line the fields first_col, first_col +
 * col_step, ... (0-based) up to the last `drop_last` fields are parsed: readGenoProb = (1, 1, 1, 0),
 * readGenoProb_ExcludeComplements = (1, 1, 2, 0), readBXDpheno = (1, 1, 1, 1), readBXDgeno = (1, 1, 2, 0)
 * (src/readData.jl:41-96, 159-165).  blmm_read_he: Helium .he (test/kinship_test.jl:5).  The table is column-major;
 * blmm_table_copy writes it into the caller's rows x cols buffer (which may be pinned: blmm_host_alloc). */
typedef struct blmm_table blmm_table;
int blmm_read_csv(const char* path, int64_t skip_lines, int64_t first_col, int64_t col_step, int64_t drop_last, blmm_table** out);
int blmm_read_he(const char* path, blmm_table** out);
int64_t blmm_table_rows(const blmm_table* t);
int64_t blmm_table_cols(const blmm_table* t);
int blmm_table_copy(const blmm_table* t, double* dst);
void blmm_table_free(blmm_table* t);

/* ---- calcKinship(G)  (src/kinship.jl:4-14) ----------------------------------------------- */
int blmm_kinship(blmm_ctx* ctx, const double* G, int64_t n, int64_t p, double* K_out);
int blmm_kinship_dev(blmm_ctx* ctx, const double* dG, int64_t n, int64_t p, double* dK_out);
/* round.(calcKinship(G), digits = d) on the device, the convention of README.md:176-181 and test/generate_test_bxdData.jl:14
 * (Julia / NumPy: round(x * 10^d) / 10^d, ties to even); digits < 0: no rounding */
int blmm_kinship_rounded(blmm_ctx* ctx, const double* G, int64_t n, int64_t p, int64_t digits, double* K_out);

/* ---- bulkscan(Y, G, [Covar], K; ...)  (src/bulkscan.jl:81-162, 188-314, 321-397, 428-526) --
 * Y n x m, G n x p, Covar n x ncov (NULL/0 = none: the intercept is the only null covariate),
 * K n x n, weights n (NULL = missing), h2_grid ngrid doubles in HOST memory for both variants
 * (ignored by null-exact).
 * Outputs: L p x m (column j = trait j, leading dimension ldL >= p); h2_out: m doubles
 * (h2_null_list; null-exact / null-grid) or p x m, ld = p (h2_panel; alt-grid).
 * blmm_bulkscan with L_out == NULL: the matrix is not copied to the host; it stays in the context's workspace, where
 * blmm_last_lod_colmax / blmm_last_lod_threshold / blmm_last_get_thresholds / blmm_last_log10p / blmm_last_lod_columns serve it
 * until the next call that produces a matrix (alt-grid: h2_out may be NULL likewise). */
int blmm_bulkscan(blmm_ctx* ctx, const blmm_opts* opts, const double* Y, int64_t n, int64_t m, const double* G,
                  int64_t p, const double* Covar, int64_t ncov, const double* K, const double* weights,
                  const double* h2_grid, int64_t ngrid, double* L_out, double* h2_out, blmm_status* status);
int blmm_bulkscan_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t n, int64_t m,
                      const double* dG, int64_t p, const double* dCovar, int64_t ncov, const double* dK,
                      const double* dweights, const double* h2_grid_host, int64_t ngrid, double* dL_out,
                      int64_t ldL, double* dh2_out, blmm_status* status);

/* ---- bulkscan WITHOUT the LOD matrix.  What the reference's users do with L is reduce it: the peak LOD of every
 * trait and its marker, the (marker, trait) pairs above a threshold (README.md:246-255, 354-359;
 * src/analysis_helpers/single_trait_analysis.jl:13-23).  Here the scan kernels do that in their epilogues and L is never
 * written (2.08 GB of HBM writes and 36 ms of PCIe at BXD size):
 *   colmax[j]  = max_i L[i, j],  argmax[j] = the lowest such i (0-based; -1: no finite-comparable entry)   -- m each, or NULL
 *   want_triplets != 0: every (i, j) with L[i, j] > thr as (ti, tj, tlod), order unspecified; *count = how many exist, the first
 *   `cap` of them are stored (call again with a larger cap when *count > cap)
 * bit-identical to blmm_lod_colmax_dev / blmm_lod_threshold_dev on the matrix blmm_bulkscan_dev writes.  The pointers inside
 * `out` are HOST pointers for blmm_bulkscan_reduced and DEVICE pointers for blmm_bulkscan_reduced_dev; h2_out as in
 * blmm_bulkscan (null-exact / null-grid: m; alt-grid: not written, may be NULL).  null-grid, and null-exact with up to 3 null
 * covariates, run fused; alt-grid, more covariates, or a call in which a trait needs one of the per-trait re-scans
 * (blmm_status.lowrank_fallback / n_illcond_rescan > 0) go through a matrix that stays in the context's workspace -- same results,
 * and the blmm_last_* consumers then serve that matrix.  blmm_last_reduced_route: 1 fused, 2 through the resident matrix.
 * Both forms return when the results are complete (they synchronise the stream). */
typedef struct blmm_reduced {
  double* colmax;
  int64_t* argmax;
  int64_t want_triplets;
  double thr;
  int64_t cap;
  int32_t* ti;
  int32_t* tj;
  double* tlod;
  int64_t* count;
} blmm_reduced;
int blmm_bulkscan_reduced(blmm_ctx* ctx, const blmm_opts* opts, const double* Y, int64_t n, int64_t m, const double* G,
                          int64_t p, const double* Covar, int64_t ncov, const double* K, const double* weights,
                          const double* h2_grid, int64_t ngrid, const blmm_reduced* out, double* h2_out, blmm_status* status);
int blmm_bulkscan_reduced_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t n, int64_t m, const double* dG,
                              int64_t p, const double* dCovar, int64_t ncov, const double* dK, const double* dweights,
                              const double* h2_grid_host, int64_t ngrid, const blmm_reduced* out, double* dh2_out,
                              blmm_status* status);
int blmm_last_reduced_route(const blmm_ctx* ctx);

/* ---- the pipeline in three calls, for hosts that run ONE PROCESS PER GPU (torch.distributed, MPI; bench.py --gpus N):
 * blmm_bulkscan_dev on every rank repeats the rotation of the whole G (10-25 % of a rank's step at n >= 500).  Instead every
 * rank prepares (design, eigen-decomposition, rotation matrix: replicated, as the reference's transform_rotation is one call,
 * src/transform_helpers.jl:21-34), rotates ITS column block of G, the host all-gathers the k-major blocks (RCCL over xGMI), and
 * the scan takes the gathered blocks.  Bit-identical to blmm_bulkscan_dev.
 *   blmm_prepare_dev              K n x n, Covar n x ncov (NULL/0: intercept only), weights n (NULL: missing)
 *   blmm_rotated_rows             rows of a rotated block (n rounded up to 8); 0 before blmm_prepare_dev
 *   blmm_rotate_block_dev         dG_block n x pb (column-major) -> dXt_block rows x ld (k-major: row k contiguous), ld >= pb
 *   blmm_bulkscan_prerotated_dev  dXt_blocks = nblocks consecutive blocks of rows x block_ld doubles, block b = the markers
 *                                 [b block_cols, min(p, (b+1) block_cols)); dY n x m = this rank's traits; outputs as blmm_bulkscan_dev
 *   blmm_scan_perms_prerotated_dev  the permutation test (blmm_scan_perms[_f32]_dev) on the same gathered blocks: this rank's
 *                                 nperms permutations against all p markers; exactly one of dLperms_out (fp64) / dLperms32_out (fp32) */
int blmm_prepare_dev(blmm_ctx* ctx, const blmm_opts* opts, int64_t n, const double* dCovar, int64_t ncov, const double* dK,
                     const double* dweights, blmm_status* status);
int64_t blmm_rotated_rows(const blmm_ctx* ctx);
int blmm_rotate_block_dev(blmm_ctx* ctx, const double* dG_block, int64_t pb, double* dXt_block, int64_t ld);
int blmm_bulkscan_prerotated_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t m, int64_t p,
                                 const double* dXt_blocks, int64_t nblocks, int64_t block_cols, int64_t block_ld,
                                 const double* h2_grid_host, int64_t ngrid, double* dL_out, int64_t ldL, double* dh2_out,
                                 blmm_status* status);
int blmm_scan_perms_prerotated_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dy, int64_t p, const double* dXt_blocks,
                                   int64_t nblocks, int64_t block_cols, int64_t block_ld, int64_t nperms, uint64_t seed,
                                   const int32_t* dperm_idx, double* dscalars_out, double* dlod_out, double* dLperms_out,
                                   float* dLperms32_out, blmm_status* status);

/* ---- the same call over several GPUs of one node (north_star: traits shard across the GPUs) -----------------------
 * Replaces the reference's thread blocking over contiguous trait ranges (src/bulkscan.jl:263-309): device r of R scans
 * the column block [r*ceil(m/R), min(m, (r+1)*ceil(m/R))) (blmm_multi_shard) and owns that block of the column-major
 * L.  One host worker thread per device; G/K/Covar/weights are replicated; no collective on the data path.
 *   gather_mode  BLMM_GATHER_HOST_SHARDS (default): every device copies its block straight into the caller's L_out /
 *                  h2_out over its own PCIe link -- the reference's result, L in host memory; L_out == NULL: the blocks stay in
 *                  HBM and blmm_multi_last_colmax / blmm_multi_last_lod_threshold reduce them there;
 *                BLMM_GATHER_NONE: the blocks stay in HBM (blmm_multi_device_result); L_out / h2_out may be NULL
 *                  (when given they are filled as well);
 *                BLMM_GATHER_ALLGATHER: an RCCL all-gather over xGMI leaves the FULL p x m matrix (ld = p, columns
 *                  padded to R*ceil(m/R)) on every device; librccl.so is loaded on first use.
 * status: NULL or an array of blmm_multi_ndev() entries, one per device.
 * device_ids == NULL or ndev <= 0: every visible device.  A device id may be repeated (several shards on one GPU). */
typedef struct blmm_multi blmm_multi;
enum blmm_gather { BLMM_GATHER_NONE = 0, BLMM_GATHER_HOST_SHARDS = 1, BLMM_GATHER_ALLGATHER = 2 };
typedef struct blmm_multi_opts {
  int32_t gather_mode; /* blmm_gather */
  int32_t reserved;
} blmm_multi_opts;
int blmm_create_multi(const int* device_ids, int ndev, blmm_multi** out);
void blmm_destroy_multi(blmm_multi* mc);
int blmm_multi_ndev(const blmm_multi* mc);
const char* blmm_multi_last_error(const blmm_multi* mc);
void blmm_default_multi_opts(blmm_multi_opts* o);
void blmm_multi_shard(int64_t m, int rank, int ndev, int64_t* lo, int64_t* hi);
int blmm_bulkscan_multi(blmm_multi* mc, const blmm_opts* opts, const blmm_multi_opts* mopts, const double* Y, int64_t n,
                        int64_t m, const double* G, int64_t p, const double* Covar, int64_t ncov, const double* K,
                        const double* weights, const double* h2_grid, int64_t ngrid, double* L_out, double* h2_out,
                        blmm_status* status);
/* Consumers of the last blmm_bulkscan_multi call's blocks WHERE THEY ARE (any gather mode; with host_shards and L_out == NULL the
 * matrix never leaves the devices): per-trait maxima (max_out m, argmax_out m or NULL) and LOD > thr triplets with global trait
 * indices, every device reducing its own block (the rules of blmm_lod_colmax / blmm_lod_threshold). */
int blmm_multi_last_colmax(blmm_multi* mc, double* max_out, int64_t* argmax_out);
int blmm_multi_last_lod_threshold(blmm_multi* mc, double thr, int64_t cap, int32_t* i_out, int32_t* j_out, double* lod_out,
                                  int64_t* count_out);
/* Device-resident result of the last blmm_bulkscan_multi with gather_mode none / allgather on device `rank`:
 * *dL (ld *ldL) holds the columns [*col_lo, *col_hi) of L, *dh2 the matching h2 entries. */
int blmm_multi_device_result(blmm_multi* mc, int rank, double** dL, int64_t* ldL, int64_t* col_lo, int64_t* col_hi,
                             double** dh2);

/* ---- scan(y, G, [Covar], K; permutation_test=true)  (src/scan.jl:485-557) ------------------
 * perm_idx: n x nperms int32, 0-based, column b = permutation b (r0perm[:, b+1] = r0[perm_idx[:, b]]);
 * NULL = the library draws them from its own counter-based generator seeded by `seed`
 * (Julia's MersenneTwister stream is not reproducible outside Julia).
 * Outputs: scalars[0] = sigma2_e, scalars[1] = h2_null; lod_out p; Lperms_out p x nperms (ld = p). */
int blmm_scan_perms(blmm_ctx* ctx, const blmm_opts* opts, const double* y, int64_t n, const double* G, int64_t p,
                    const double* Covar, int64_t ncov, const double* K, const double* weights, int64_t nperms,
                    uint64_t seed, const int32_t* perm_idx, double* scalars_out, double* lod_out,
                    double* Lperms_out, blmm_status* status);
int blmm_scan_perms_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dy, int64_t n, const double* dG,
                        int64_t p, const double* dCovar, int64_t ncov, const double* dK, const double* dweights,
                        int64_t nperms, uint64_t seed, const int32_t* dperm_idx, double* dscalars_out,
                        double* dlod_out, double* dLperms_out, blmm_status* status);

/* fp32 permutation matrix (BASELINE.json configs[4]): the null model (eigen-decomposition, h2, residuals, panel construction)
 * stays fp64; the marker rotation and the p x nperms contraction run on the fp32 matrix cores and Lperms_out is float
 * (p x nperms, ld = p).  Expected agreement with the fp64 path: |d| <= 1e-3 |ref| + 1e-4.  The original trait's lod_out keeps an
 * fp64 numerator (taken from G itself); its marker norms come from the fp32-rotated markers, so it agrees with blmm_scan_perms'
 * lod_out to ~1e-7 relative, not bit for bit.  With null covariates beyond the intercept (or tuning "f32_rotation" = 0) the
 * rotation is the fp64 one, converted, and lod_out is blmm_scan_perms' bit for bit. */
int blmm_scan_perms_f32(blmm_ctx* ctx, const blmm_opts* opts, const double* y, int64_t n, const double* G, int64_t p,
                        const double* Covar, int64_t ncov, const double* K, const double* weights, int64_t nperms,
                        uint64_t seed, const int32_t* perm_idx, double* scalars_out, double* lod_out,
                        float* Lperms_out, blmm_status* status);
int blmm_scan_perms_f32_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dy, int64_t n, const double* dG,
                            int64_t p, const double* dCovar, int64_t ncov, const double* dK, const double* dweights,
                            int64_t nperms, uint64_t seed, const int32_t* dperm_idx, double* dscalars_out,
                            double* dlod_out, float* dLperms_out, blmm_status* status);

/* ---- scan(y, G, [Z], K; assumption = "alt") -> scan_alt (src/scan.jl:397-453): the variance components are re-estimated for
 * every marker (fitlmm on [Z g_i], src/lmm.jl:56-86, one Brent search per marker on the device).
 * scalars_out = [sigma2_e, h2_null]; lod_out p; h2_each_out p (`h2_each_marker`).  opts->compat_flags:
 * BLMM_COMPAT_ALT_TRUE_WEIGHTS. */
int blmm_scan_alt(blmm_ctx* ctx, const blmm_opts* opts, const double* y, int64_t n, const double* G, int64_t p,
                  const double* Covar, int64_t ncov, const double* K, const double* weights, double* scalars_out,
                  double* lod_out, double* h2_each_out, blmm_status* status);
int blmm_scan_alt_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dy, int64_t n, const double* dG, int64_t p,
                      const double* dCovar, int64_t ncov, const double* dK, const double* dweights, double* dscalars_out,
                      double* dlod_out, double* dh2_each_out, blmm_status* status);
/* The bulk form of it (not in the reference, which has the single-trait scan_alt and the grid approximation
 * bulkscan_alt_grid): for EVERY (trait, marker) the exact heritability under the alternative (one Brent search per test) and
 * the LOD against the trait's null model.  L_out, h2_panel_out: p x m column-major (leading dimensions ldL, ldH in the _dev
 * form); h2_null_out m; sigma2_out m or NULL.  Column j equals blmm_scan_alt on trait j bit for bit.  ~0.02 us per test at
 * n = 79 (64 traits x 7321 markers: 8.6 ms host to host) -- the whole BXD matrix would take ~5 s where the 16-point grid of
 * bulkscan_alt_grid takes 15 ms: meant for subsets of traits.  At most 31 null covariates (the per-marker design [Z0 x] has c + 1 <= 32 columns; beyond 8 the run-time-c kernel k_dyn_alt_brent). */
int blmm_bulkscan_alt_exact(blmm_ctx* ctx, const blmm_opts* opts, const double* Y, int64_t n, int64_t m, const double* G, int64_t p,
                            const double* Covar, int64_t ncov, const double* K, const double* weights, double* L_out,
                            double* h2_panel_out, double* h2_null_out, double* sigma2_out, blmm_status* status);
int blmm_bulkscan_alt_exact_dev(blmm_ctx* ctx, const blmm_opts* opts, const double* dY, int64_t n, int64_t m, const double* dG,
                                int64_t p, const double* dCovar, int64_t ncov, const double* dK, const double* dweights,
                                double* dL_out, int64_t ldL, double* dh2_panel_out, int64_t ldH, double* dh2_null_out,
                                double* dsigma2_out, blmm_status* status);

/* ---- on-device consumer of L: column maxima (per-trait / per-permutation peak LOD and its marker, 0-based) -------
 * The reduction behind get_thresholds (src/analysis_helpers/single_trait_analysis.jl:13-23); argmax_out may be NULL. */
int blmm_lod_colmax(blmm_ctx* ctx, const double* L, int64_t p, int64_t m, double* max_out, int64_t* argmax_out);
int blmm_lod_colmax_dev(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, double* dmax_out,
                        int64_t* dargmax_out);

/* ---- -log10 p-values: lod2log10p.(L, chisq_df)  (src/util.jl:199-206; `output_pvals`, src/bulkscan.jl:154-157,
 * src/scan.jl:353-355).  df = 1: LOD + x w(x), x = sqrt(LOD ln 10), w = -log10(erfcx(x)) / x from bucketed polynomials
 * (4e-15 relative; BLMM_PVAL_LIBM=1: erfc / erfcx / log instead); general df through ln Q(df/2, .) in log space. */
int blmm_lod2log10p(blmm_ctx* ctx, const double* L, int64_t p, int64_t m, int64_t chisq_df, double* P_out);
int blmm_lod2log10p_dev(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, int64_t chisq_df, double* dP_out,
                        int64_t ldP);
/* `output_pvals = true` of bulkscan (src/bulkscan.jl:154-157) INSIDE the scan: the next blmm_bulkscan / blmm_bulkscan_dev /
 * blmm_bulkscan_prerotated_dev call of this context also writes -log10 p, p x m with leading dimension ldP, into the DEVICE
 * buffer dP_out -- or, dP_out == NULL, into a buffer of the context that blmm_last_log10p then hands out without computing
 * anything.  chisq_df = 1 with a null-* method: a second output of the scan kernels' epilogues (the LOD matrix is not read
 * back from HBM for it); alt-grid or another chisq_df: the column pass above, run inside the call.  One-shot: the request is
 * consumed by that call; chisq_df = 0 withdraws it. */
int blmm_set_log10p_output(blmm_ctx* ctx, double* dP_out, int64_t ldP, int64_t chisq_df);
/* ---- threshold filter: every (marker, trait) with LOD > thr as a sparse triplet (0-based int32 indices), the count on
 * the device (README.md:354-359).  At most `cap` triplets are stored, *count is the total found; order unspecified. */
int blmm_lod_threshold(blmm_ctx* ctx, const double* L, int64_t p, int64_t m, double thr, int64_t cap, int32_t* i_out,
                       int32_t* j_out, double* lod_out, int64_t* count_out);
int blmm_lod_threshold_dev(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, double thr, int64_t cap,
                           int32_t* di_out, int32_t* dj_out, double* dlod_out, int64_t* dcount_out);
/* ---- get_thresholds (src/analysis_helpers/single_trait_analysis.jl:13-23): quantiles (Julia's default, linear
 * interpolation) at `probs` (HOST array, nprobs <= 64) of the per-permutation maxima; column maxima, sort and
 * interpolation on the device, thrs_out (nprobs doubles) in HOST memory. */
int blmm_get_thresholds(blmm_ctx* ctx, const double* Lperms, int64_t p, int64_t nperms, const double* probs, int64_t nprobs,
                        double* thrs_out);
int blmm_get_thresholds_dev(blmm_ctx* ctx, const double* dLperms, int64_t p, int64_t nperms, int64_t ld, const double* probs,
                            int64_t nprobs, double* thrs_out);
/* ---- the same consumers on the LOD matrix of the LAST host-pointer call of this context (blmm_bulkscan: L;
 * blmm_scan_perms: L_perms), which is still resident in HBM: nothing is uploaded again. */
int blmm_last_log10p(blmm_ctx* ctx, int64_t chisq_df, double* P_out);
int blmm_last_lod_threshold(blmm_ctx* ctx, double thr, int64_t cap, int32_t* i_out, int32_t* j_out, double* lod_out,
                            int64_t* count_out);
int blmm_last_get_thresholds(blmm_ctx* ctx, const double* probs, int64_t nprobs, double* thrs_out);
/* shape of the resident matrix (0 x 0 and BLMM_ERR_INVALID when there is none); its column maxima (max_out m, argmax_out m or
 * NULL; the rule of blmm_lod_colmax); selected columns: out is p x ncols, column k = L[:, cols[k]] (0-based) */
int blmm_last_dims(const blmm_ctx* ctx, int64_t* p_out, int64_t* m_out);
int blmm_last_lod_colmax(blmm_ctx* ctx, double* max_out, int64_t* argmax_out);
int blmm_last_lod_columns(blmm_ctx* ctx, const int64_t* cols, int64_t ncols, double* out);

/* ---- lower-level seams (1:1 with the reference's internal functions; used by the parity tests) ---- */
/* transform_rotation(y, [Z G], K)  (src/transform_helpers.jl:1-54): Y0 n x m, X0 n x (c+p) (first c
 * columns = rotated null covariates, intercept first when add_intercept), lambda n. */
int blmm_rotate(blmm_ctx* ctx, const blmm_opts* opts, const double* Y, int64_t n, int64_t m, const double* G,
                int64_t p, const double* Covar, int64_t ncov, const double* K, double* Y0_out, double* X0_out,
                double* lambda_out, blmm_status* status);
/* fitlmm over every column of Y0 (src/lmm.jl:56-86, src/gridbrent.jl:9-24): h2, sigma2, ell: m each. */
int blmm_null_h2_brent(blmm_ctx* ctx, const blmm_opts* opts, const double* Y0, int64_t n, int64_t m,
                       const double* Z0, int64_t c, const double* lambda, double* h2_out, double* sigma2_out,
                       double* ell_out, blmm_status* status);
/* wls_multivar(Y0, Z0, makeweights(h2_g), prior).Ell for every grid point (src/wls.jl:103-176,
 * src/bulkscan_helpers.jl:267-269): Ell_out ngrid x m (ld = ngrid). */
int blmm_null_loglik_grid(blmm_ctx* ctx, const blmm_opts* opts, const double* Y0, int64_t n, int64_t m,
                          const double* Z0, int64_t c, const double* lambda, const double* h2_grid, int64_t ngrid,
                          double* Ell_out, blmm_status* status);
/* weighted_liteqtl(Y0, X0, lambda, hsq; num_of_covar)  (src/bulkscan_helpers.jl:175-201): X0 n x (c+p). */
int blmm_weighted_liteqtl(blmm_ctx* ctx, const double* Y0, int64_t n, int64_t m, const double* X0, int64_t c,
                          int64_t p, const double* lambda, double hsq, double* LOD_out, blmm_status* status);
/* univar_liteqtl over every column of Y0 with the per-trait h2 supplied by the caller
 * (src/bulkscan_helpers.jl:138-146): the exact-weights LOD kernel on its own. */
int blmm_liteqtl_given_h2(blmm_ctx* ctx, const double* Y0, int64_t n, int64_t m, const double* X0, int64_t c,
                          int64_t p, const double* lambda, const double* h2, double* LOD_out, blmm_status* status);

#ifdef __cplusplus
}
#endif
#endif /* BULKLMM_HIP_H */
