// kernels_dyn.hip -- the conditioning guard of the null-exact scan and its QR-grade re-scan.
//
// The scan kernels project the weighted markers and traits off the weighted null covariates through the Cholesky factor of
// A = Z0'WZ0 (closed-form WLS): its error grows with cond(A) = cond(sqrt(W) Z0)^2, where the reference's
// `resid` (src/wls.jl:221-241: Householder QR, `X \ y`) loses only cond(sqrt(W) Z0).  With one covariate the two coincide;
// with several covariates and an h2 estimate at the h2 -> 1 boundary the weights span 8-9 orders of magnitude, the weighted
// columns become nearly collinear (cond(sqrt(W) Z0) ~ 2e4 in the case tools/fuzz_parity.py found: n = 13, 8 covariates) and
// the Cholesky form misses the 1e-6 parity bound.  So, per trait:
//   k_illcond_flag   rho_j = min_q d_q / A_qq over the Cholesky pivots d_q of A (the share of column q's weighted norm that is
//                    left after the columns before it: scale invariant, 1 for orthogonal columns).  rho_j < 1e-4
//                    (cond(sqrt(W) Z0) beyond ~100) puts the trait on a device list; count in stat[ST_ILLCOND].
//   k_scan_qr        re-computes the LOD columns of the listed traits the way computeR_LMM does (src/bulkscan_helpers.jl:47-64)
//                    with an ORTHONORMAL basis of span(sqrt(W) Z0) built by Gram-Schmidt with re-orthogonalisation (backward
//                    stable like Householder QR: error ~ cond, not cond^2) and explicit residuals, plain fp64 + libm log10.
// Both are no-ops for c = 1 and for data without such traits (a fixed grid reads the count on the device).
#include "blmm_internal.h"
#include "fastmath.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace blmm {

#define KCHECK()                                                                                      \
  do {                                                                                                \
    hipError_t e__ = hipGetLastError();                                                               \
    if (e__ != hipSuccess) return fail(ctx, BLMM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e__)); \
  } while (0)

// ---- conditioning of the weighted null design, one thread per trait ---------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) k_illcond_flag(int n, int64_t m, const double* __restrict__ Z0,
                                                      const double* __restrict__ lam, const double* __restrict__ h2v,
                                                      double rho_min, int* __restrict__ list, int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* sLam = sh;
  double* sZ = sh + n;
  for (int e = threadIdx.x; e < n; e += blockDim.x) sLam[e] = lam[e];
  for (int e = threadIdx.x; e < n * C; e += blockDim.x) sZ[e] = Z0[e];
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  constexpr int NA = C * (C + 1) / 2;
  const double h2 = h2v[j];
  const double delta = h2 / (1.0 - h2);
  double A[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
  for (int k = 0; k < n; ++k) {
    const double w = fabs(1.0 / fma(delta, sLam[k], 1.0));
#pragma unroll
    for (int q = 0; q < C; ++q) {
      const double wz = w * sZ[q * n + k];
#pragma unroll
      for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(wz, sZ[r * n + k], A[q * (q + 1) / 2 + r]);
    }
  }
  double L[NA], rho = 1.0;
#pragma unroll
  for (int q = 0; q < C; ++q) {
#pragma unroll
    for (int r = 0; r <= q; ++r) {
      double s = A[q * (q + 1) / 2 + r];
#pragma unroll
      for (int u = 0; u < r; ++u) s = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], s);
      if (r == q) {
        const double share = s / A[q * (q + 1) / 2 + q];
        rho = (share < rho || !(share == share)) ? share : rho;   // NaN sticks
        L[q * (q + 1) / 2 + q] = sqrt(s);
      } else {
        L[q * (q + 1) / 2 + r] = s / L[r * (r + 1) / 2 + r];
      }
    }
  }
  if (!(rho >= rho_min)) {
    const unsigned long long slot = atomicAdd((unsigned long long*)&stat[ST_ILLCOND], 1ull);
    list[slot] = (int)j;
  }
}

// ---- block-wide sums of NV values per thread (256 threads), fixed summation order ---------------------------------------
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* s_red /* [4][NV] */) {
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = group_sum<64>(v[q]);
  const int w = threadIdx.x >> 6;
  __syncthreads();                              // the readers of the previous reduction are done with s_red
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) s_red[w * NV + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = (s_red[q] + s_red[NV + q]) + (s_red[2 * NV + q] + s_red[3 * NV + q]);
}

// Removes from column `tgt` (n doubles, thread t owns the rows t, t + 256, ..) its components along the orthonormal columns
// Qb[0 .. nq): eight coefficients per block reduction (classical Gram-Schmidt inside a chunk, modified across chunks).
__device__ __forceinline__ void project_out(double* tgt, const double* Qb, int nq, int n, double* s_red) {
  for (int r0 = 0; r0 < nq; r0 += 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) {
      const double v = tgt[k];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + u < nq) t[u] = fma(Qb[(size_t)(r0 + u) * n + k], v, t[u]);
    }
    block_sum<8>(t, s_red);
    for (int k = threadIdx.x; k < n; k += 256) {
      double v = tgt[k];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + u < nq) v = fma(-t[u], Qb[(size_t)(r0 + u) * n + k], v);
      tgt[k] = v;
    }
  }
}

// CQ: compile-time bound of the per-marker coefficient arrays (c <= CQ).  buf: (c + 2) * n doubles per workgroup -- the
// weights' square roots S, the orthonormal basis Qb (c columns) and the normalised trait residual yb -- in LDS when it fits
// (`slab` == nullptr) and in a per-workgroup slab of global memory otherwise.
template <int CQ>
__global__ void __launch_bounds__(256) k_scan_qr(int n, int c, const double* __restrict__ Yt, int64_t ldy,
                                                 const double* __restrict__ Xt, int64_t ldx, int64_t p,
                                                 const double* __restrict__ Z0, const double* __restrict__ lam,
                                                 const double* __restrict__ h2v, const int* __restrict__ list,
                                                 double* slab, double* __restrict__ L, int64_t ldL, int64_t* stat,
                                                 double* __restrict__ Pv, int64_t ldPv, const double* __restrict__ pvtab) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ double s_red[4 * 8];
  const int64_t cnt = stat[ST_ILLCOND];
  if (cnt <= 0) return;
  double* buf = slab ? slab + (size_t)blockIdx.x * (size_t)(c + 2) * n : sh;
  double* S = buf;
  double* Qb = buf + n;
  double* yb = buf + (size_t)(1 + c) * n;
  const double scale = -0.5 * (double)n;
  for (int64_t item = blockIdx.x; item < cnt; item += gridDim.x) {
    const int64_t j = list[item];
    const double h2 = h2v[j];
    const double delta = h2 / (1.0 - h2);
    __syncthreads();                            // the marker loop of the previous trait has finished reading buf
    // sqrt.(abs.(makeweights(h2, lambda))) and the weighted columns (src/bulkscan_helpers.jl:138-141); every thread works
    // on its own rows k = t, t + 256, .. until the marker loop
    for (int k = threadIdx.x; k < n; k += 256) {
      const double s = sqrt(fabs(1.0 / fma(delta, lam[k], 1.0)));
      S[k] = s;
      for (int q = 0; q < c; ++q) Qb[(size_t)q * n + k] = s * Z0[(size_t)q * n + k];
      yb[k] = s * Yt[(int64_t)k * ldy + j];
    }
    for (int q = 0; q < c; ++q) {
      double* col = Qb + (size_t)q * n;
      project_out(col, Qb, q, n, s_red);
      project_out(col, Qb, q, n, s_red);        // "twice is enough": orthogonal to rounding
      double nn[1] = {0.0};
      for (int k = threadIdx.x; k < n; k += 256) nn[0] = fma(col[k], col[k], nn[0]);
      block_sum<1>(nn, s_red);
      const double inv = 1.0 / sqrt(nn[0]);
      for (int k = threadIdx.x; k < n; k += 256) col[k] *= inv;
    }
    project_out(yb, Qb, c, n, s_red);
    project_out(yb, Qb, c, n, s_red);
    {
      double nn[1] = {0.0};
      for (int k = threadIdx.x; k < n; k += 256) nn[0] = fma(yb[k], yb[k], nn[0]);
      block_sum<1>(nn, s_red);
      if (threadIdx.x == 0 && !(sqrt(nn[0]) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
      const double inv = 1.0 / sqrt(nn[0]);
      for (int k = threadIdx.x; k < n; k += 256) yb[k] *= inv;
    }
    __syncthreads();                            // basis and trait residual complete: from here every thread reads all rows
    for (int64_t i0 = 0; i0 < p; i0 += 256) {
      const int64_t i = i0 + threadIdx.x;
      if (i >= p) continue;
      double t[CQ], t2[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) { t[q] = 0.0; t2[q] = 0.0; }
      for (int k = 0; k < n; ++k) {
        const double x = S[k] * Xt[(int64_t)k * ldx + i];
#pragma unroll
        for (int q = 0; q < CQ; ++q)
          if (q < c) t[q] = fma(Qb[(size_t)q * n + k], x, t[q]);
      }
      for (int k = 0; k < n; ++k) {             // second projection pass: coefficients of the first residual
        double xp = S[k] * Xt[(int64_t)k * ldx + i];
#pragma unroll
        for (int q = 0; q < CQ; ++q)
          if (q < c) xp = fma(-t[q], Qb[(size_t)q * n + k], xp);
#pragma unroll
        for (int q = 0; q < CQ; ++q)
          if (q < c) t2[q] = fma(Qb[(size_t)q * n + k], xp, t2[q]);
      }
      double xx = 0.0, num = 0.0;
      for (int k = 0; k < n; ++k) {
        double xp = S[k] * Xt[(int64_t)k * ldx + i];
#pragma unroll
        for (int q = 0; q < CQ; ++q)
          if (q < c) xp = fma(-(t[q] + t2[q]), Qb[(size_t)q * n + k], xp);
        xx = fma(xp, xp, xx);
        num = fma(xp, yb[k], num);
      }
      if (!(sqrt(xx) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
      const double r = num / sqrt(xx);
      const double u1 = 1.0 - r * r;            // r2lod, src/bulkscan_helpers.jl:22-24
      double lod = scale * log10(u1);
      if (!(u1 > 0.0)) lod = (u1 == 0.0) ? INFINITY : NAN;
      L[j * ldL + i] = lod;
      if (Pv) Pv[j * ldPv + i] = fast_log10p1(lod, reinterpret_cast<const dpair*>(pvtab));   // the fused `output_pvals` column
    }
  }
}

// =================================================================================================================
// Run-time covariate counts: c = CTPL + 1 .. CMAX (9 .. 32) null covariates.  The tuned kernels of kernels_prep.hip are templates
// over c with the c x c normal equations in REGISTERS (instantiated for 1 .. CTPL); the reference has no cap (src/wls.jl:27-60,
// src/bulkscan_helpers.jl:187-193 take any c).  Here the normal equations A = Z0'WZ0, their Cholesky factor L and L^-1 live in
// LDS and a group of NT threads (a whole workgroup, or one 64-thread workgroup = one wave per trait) builds them together.
// Same arithmetic as the templates (closed-form WLS): these kernels produce the operands the SAME scan
// kernels consume -- the covariate-chunked exact kernel (k_scan<.., MORE>) and the table / alt kernels take any c.
// Packed lower-triangular storage: element (q, r), r <= q, at q (q + 1) / 2 + r.
// =================================================================================================================
constexpr int DYN_NA = CMAX * (CMAX + 1) / 2;

// Builds, from the weights already in sW[0 .. n): sA <- Cholesky factor L of A = Z0' diag(sW) Z0 (packed), sLi <- L^-1 (packed;
// skipped when null), sScal[0] = ln det A, sScal[1] = min_q d_q / A_qq (the conditioning guard's pivot share).
// Every thread of the workgroup (NT threads, all of them in the call) takes part; Z0 is read from global memory (n x c,
// column-major, L1 / L2-resident).
// The design of the run-time-c kernels: columns 0 .. cz - 1 are Z0's (global memory, n x cz column-major); a column beyond is
// `x` (n contiguous values; scan_alt's design [Z0 x_i], k_dyn_alt_brent).  x == nullptr: Z0 alone.
struct DynCols {
  const double* Z0; const double* x; int cz; int n;
  __device__ __forceinline__ const double* col(int q) const { return q < cz ? Z0 + (size_t)q * n : x; }
};
template <int NT>
__device__ __forceinline__ void dyn_factor(int n, int c, const DynCols dc, const double* sW, double* sA, double* sLi,
                                           double* sScal) {
  const int t = threadIdx.x, na = c * (c + 1) / 2;
  __shared__ double s_diag[CMAX];
  for (int e = t; e < na; e += NT) {
    int q = 0;
    while ((q + 1) * (q + 2) / 2 <= e) ++q;
    const int r = e - q * (q + 1) / 2;
    const double* zq = dc.col(q);
    const double* zr = dc.col(r);
    double acc = 0.0;
    for (int k = 0; k < n; ++k) acc = fma(sW[k] * zq[k], zr[k], acc);
    sA[e] = acc;
    if (r == q) s_diag[q] = acc;
  }
  __syncthreads();
  // right-looking Cholesky, one column per step: the column below the pivot, then the trailing update
  double logdet = 0.0, rho = 1.0;
  for (int j = 0; j < c; ++j) {
    const double piv = sA[j * (j + 1) / 2 + j];
    const double share = piv / s_diag[j];
    rho = (share < rho || !(share == share)) ? share : rho;
    logdet += log(piv);
    const double ljj = sqrt(piv);
    __syncthreads();                                      // everybody has read the pivot
    for (int i = j + t; i < c; i += NT) sA[i * (i + 1) / 2 + j] = (i == j) ? ljj : sA[i * (i + 1) / 2 + j] / ljj;
    __syncthreads();
    const int rem = c - j - 1;                            // trailing rows / columns j + 1 .. c - 1
    for (int e = t; e < rem * (rem + 1) / 2; e += NT) {
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= e) ++a;
      const int b = e - a * (a + 1) / 2;
      const int i = j + 1 + a, k = j + 1 + b;
      sA[i * (i + 1) / 2 + k] = fma(-sA[i * (i + 1) / 2 + j], sA[k * (k + 1) / 2 + j], sA[i * (i + 1) / 2 + k]);
    }
    __syncthreads();
  }
  if (t == 0) { sScal[0] = logdet; sScal[1] = rho; }
  if (sLi) {
    // column r of L^-1 by forward substitution, one thread per column
    for (int r = t; r < c; r += NT) {
      for (int q = r; q < c; ++q) {
        double sacc = (q == r) ? 1.0 : 0.0;
        for (int u = r; u < q; ++u) sacc = fma(-sA[q * (q + 1) / 2 + u], sLi[u * (u + 1) / 2 + r], sacc);
        sLi[q * (q + 1) / 2 + r] = sacc / sA[q * (q + 1) / 2 + q];
      }
    }
  }
  __syncthreads();
}

template <int NT>
__device__ __forceinline__ void dyn_factor(int n, int c, const double* __restrict__ Z0, const double* sW, double* sA, double* sLi,
                                           double* sScal) {
  dyn_factor<NT>(n, c, DynCols{Z0, nullptr, c, n}, sW, sA, sLi, sScal);
}

// weights of one h2 into sW; returns (to every thread) sum_k ln(delta lambda_k + 1) and sets *nonpos when a weight is <= 0
// (`absw`: the scan's sqrt.(abs.(w)) convention, src/bulkscan_helpers.jl:138; the likelihood uses w itself, src/wls.jl:40)
// SQW: the square roots of the weights (scan_alt's closing wls calls, src/scan.jl:431-437); the caller halves the returned sum
template <int NT, bool SQW = false>
__device__ __forceinline__ double dyn_weights(int n, double h2, const double* __restrict__ lam, double* sW, bool absw, int* nonpos,
                                              double* s_red /* NT / 64 + 1 */) {
  const double delta = h2 / (1.0 - h2);
  double ls = 0.0;
  int bad = 0;
  for (int k = threadIdx.x; k < n; k += NT) {
    const double tk = fma(delta, lam[k], 1.0);
    const double w = 1.0 / tk;
    bad |= !(w > 0.0);
    sW[k] = SQW ? sqrt(w) : (absw ? fabs(w) : w);
    ls += log(tk);
  }
  ls = group_sum<64>(ls);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = ls;
  if (__any(bad) && nonpos) *nonpos = 1;                  // every thread of a wave that saw one (64-thread groups: every thread)
  __syncthreads();
  double tot = 0.0;
  for (int w = 0; w < NT / 64; ++w) tot += s_red[w];
  return tot;
}

// ell of one trait (sY, n values in LDS) under the factor in sA (weights sW): v = Z0'Wy (one thread per covariate), t = L^-1 v
// (every thread, redundantly), the reference's formula (src/wls.jl:69-88).  sV: c doubles of LDS.  Returns ell / sigma2 / rss.
template <int NT>
__device__ __forceinline__ void dyn_ell(int n, int c, const DynCols dc, const double* sW, const double* sY, const double* sA,
                                        double logdet, double logsum, double prior_a, double prior_b, int reml, double* sV,
                                        double* ell_out, double* sigma2_out, double* rss_out) {
  for (int q = threadIdx.x; q <= c; q += NT) {
    double acc = 0.0;
    if (q < c) { const double* zq = dc.col(q); for (int k = 0; k < n; ++k) acc = fma(sW[k] * sY[k], zq[k], acc); }
    else { for (int k = 0; k < n; ++k) acc = fma(sW[k] * sY[k], sY[k], acc); }
    sV[q] = acc;                                          // sV[c] = y'Wy
  }
  __syncthreads();
  double tt = 0.0;
  {
    double tq[CMAX];
    for (int q = 0; q < c; ++q) {
      double sacc = sV[q];
      for (int u = 0; u < q; ++u) sacc = fma(-sA[q * (q + 1) / 2 + u], tq[u], sacc);
      tq[q] = sacc / sA[q * (q + 1) / 2 + q];
      tt = fma(tq[q], tq[q], tt);
    }
  }
  const double rss = sV[c] - tt;
  const double prior_df = prior_b > 0.0 ? prior_b + 2.0 : prior_b;
  const double num = rss + prior_a * prior_b;
  const double sigma2 = num / ((reml ? (double)(n - c) : (double)n) + prior_df);
  const double ls = log(sigma2);
  double ell = -0.5 * (((double)n + prior_b) * ls + logsum + num / sigma2);
  if (reml) ell += 0.5 * ((double)c * ls - logdet);
  *ell_out = ell; *sigma2_out = sigma2; *rss_out = rss;
  __syncthreads();                                        // sV may be overwritten by the next evaluation
}

template <int NT>
__device__ __forceinline__ void dyn_ell(int n, int c, const double* __restrict__ Z0, const double* sW, const double* sY, const double* sA,
                                        double logdet, double logsum, double prior_a, double prior_b, int reml, double* sV,
                                        double* ell_out, double* sigma2_out, double* rss_out) {
  dyn_ell<NT>(n, c, DynCols{Z0, nullptr, c, n}, sW, sY, sA, logdet, logsum, prior_a, prior_b, reml, sV, ell_out, sigma2_out, rss_out);
}

// ---- fitlmm for every trait: one 64-thread workgroup (= one wave) per trait; brent_search is the template of kernels_prep.hip's
// k_brent restated here because that file's copy is file-local ------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ double dyn_brent_search(F& f, int nint, int* hit_max) {
  const double golden = 0.5 * (3.0 - sqrt(5.0));
  const double rel_tol = 1.4901161193847656e-08, abs_tol = 2.220446049250313e-16;
  double best_x = 0.0, best_f = INFINITY;
  for (int iv = 0; iv < nint; ++iv) {
    double x_lower = (double)iv / (double)nint, x_upper = (iv + 1 == nint) ? 1.0 : (double)(iv + 1) / (double)nint;
    double new_minimizer = x_lower + golden * (x_upper - x_lower);
    double new_minimum = f(new_minimizer);
    double step = 0.0, old_step = 0.0;
    double old_minimizer = new_minimizer, old_old_minimizer = new_minimizer;
    double old_minimum = new_minimum, old_old_minimum = new_minimum;
    bool done = false;
    int it = 0;
    for (; it < 1000; ++it) {
      double p = 0.0, q = 0.0;
      const double x_tol = rel_tol * fabs(new_minimizer) + abs_tol;
      const double x_mid = (x_upper + x_lower) / 2;
      if (fabs(new_minimizer - x_mid) <= 2 * x_tol - (x_upper - x_lower) / 2) { done = true; break; }
      if (fabs(old_step) > x_tol) {
        const double r = (new_minimizer - old_minimizer) * (new_minimum - old_old_minimum);
        q = (new_minimizer - old_old_minimizer) * (new_minimum - old_minimum);
        p = (new_minimizer - old_old_minimizer) * q - (new_minimizer - old_minimizer) * r;
        q = 2 * (q - r);
        if (q > 0) p = -p; else q = -q;
      }
      double nstep, nold;
      if (fabs(p) < fabs(q * old_step / 2) && p < q * (x_upper - new_minimizer) && p < q * (new_minimizer - x_lower)) {
        nold = step;
        nstep = p / q;
        const double x_temp = new_minimizer + nstep;
        if ((x_temp - x_lower) < 2 * x_tol || (x_upper - x_temp) < 2 * x_tol) nstep = (new_minimizer < x_mid) ? x_tol : -x_tol;
      } else {
        nold = (new_minimizer < x_mid) ? x_upper - new_minimizer : x_lower - new_minimizer;
        nstep = golden * nold;
      }
      const double new_x = (fabs(nstep) >= x_tol) ? new_minimizer + nstep : new_minimizer + ((nstep > 0) ? x_tol : -x_tol);
      const double new_f = f(new_x);
      old_step = nold; step = nstep;
      if (new_f < new_minimum) {
        if (new_x < new_minimizer) x_upper = new_minimizer; else x_lower = new_minimizer;
        old_old_minimizer = old_minimizer; old_old_minimum = old_minimum;
        old_minimizer = new_minimizer; old_minimum = new_minimum;
        new_minimizer = new_x; new_minimum = new_f;
      } else {
        if (new_x < new_minimizer) x_lower = new_x; else x_upper = new_x;
        if (new_f <= old_minimum || old_minimizer == new_minimizer) {
          old_old_minimizer = old_minimizer; old_old_minimum = old_minimum;
          old_minimizer = new_x; old_minimum = new_f;
        } else if (new_f <= old_old_minimum || old_old_minimizer == new_minimizer || old_old_minimizer == old_minimizer) {
          old_old_minimizer = new_x; old_old_minimum = new_f;
        }
      }
    }
    if (it >= 1000 && !done) *hit_max = 1;
    if (new_minimum < best_f || iv == 0) { best_f = new_minimum; best_x = new_minimizer; }
  }
  return best_x;
}

__global__ void __launch_bounds__(64) k_dyn_brent(NullModel nm, const double* __restrict__ Yt, int64_t ldy, int64_t m,
                                                  const double* __restrict__ Z0, const double* __restrict__ lam,
                                                  double* __restrict__ h2out, double* __restrict__ s2out, double* __restrict__ ellout,
                                                  int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ double sA[DYN_NA], sV[CMAX + 1], sScal[2], s_red[2];
  const int n = nm.n, c = nm.c;
  double* sW = sh;
  double* sY = sh + n;
  const int64_t j = blockIdx.x;                                  // one workgroup (one wave) per trait: every thread runs the
  for (int k = threadIdx.x; k < n; k += 64) sY[k] = Yt[(int64_t)k * ldy + j];   // same (scalar) search on the same values
  __syncthreads();
  int nonpos = 0, hit_max = 0;
  double e_ell = 0.0, e_s2 = 0.0, e_rss = 0.0;
  auto f = [&](double h2) {
    const double logsum = dyn_weights<64>(n, h2, lam, sW, false, &nonpos, s_red);
    dyn_factor<64>(n, c, Z0, sW, sA, nullptr, sScal);
    dyn_ell<64>(n, c, Z0, sW, sY, sA, sScal[0], logsum, nm.prior_a, nm.prior_b, nm.reml, sV, &e_ell, &e_s2, &e_rss);
    return -e_ell;
  };
  const double best = dyn_brent_search(f, nm.optim_interval < 1 ? 1 : nm.optim_interval, &hit_max);
  (void)f(best);                                                 // the final wls at the minimiser (src/lmm.jl:84)
  wave_count(&stat[ST_H2_BOUNDARY], threadIdx.x == 0 && h2_on_boundary(best));
  if (threadIdx.x == 0) {
    h2out[j] = best;
    if (s2out) s2out[j] = e_s2;
    if (ellout) ellout[j] = e_ell;
    if (hit_max) atomicAdd((unsigned long long*)&stat[ST_BRENT_MAXIT], 1ull);
    if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
  }
}

// ---- scan_alt with a run-time covariate count (src/scan.jl:397-453; the counterpart of kernels_prep.hip's k_alt_brent, which is a
// template over c <= 8): one 64-thread workgroup per (marker, trait); the design [Z0 x_i] has c + 1 <= CMAX columns, x_i staged in LDS.
// The search on the likelihood of fitlmm(y0, [Z0 x_i], lambda; reml, optim_interval), then the reference's two closing wls calls
// (ML, the SQUARE ROOTS of the weights handed over as weights; true_w: makeweights(h2) itself, BLMM_COMPAT_ALT_TRUE_WEIGHTS).
__global__ void __launch_bounds__(64) k_dyn_alt_brent(NullModel nm, const double* __restrict__ Yt, int64_t ldy,
                                                      const double* __restrict__ Xt, int64_t ldx, int64_t p,
                                                      const double* __restrict__ Z0, const double* __restrict__ lam,
                                                      const double* __restrict__ h2null, int true_w, double* __restrict__ lod,
                                                      double* __restrict__ h2each, int64_t* stat, int64_t trait0, int64_t ldL, int64_t ldH) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ double sA[DYN_NA], sV[CMAX + 1], sScal[2], s_red[2];
  const int n = nm.n, c = nm.c, D = c + 1;
  double* sW = sh;
  double* sY = sh + n;
  double* sX = sh + 2 * n;
  const int64_t tr = trait0 + blockIdx.y;
  const int64_t i = blockIdx.x;
  lod += tr * ldL; h2each += tr * ldH;
  for (int k = threadIdx.x; k < n; k += 64) { sY[k] = Yt[(int64_t)k * ldy + tr]; sX[k] = Xt[(int64_t)k * ldx + i]; }
  __syncthreads();
  const DynCols alt{Z0, sX, c, n}, nul{Z0, nullptr, c, n};
  int nonpos = 0, hit_max = 0;
  double e_ell = 0.0, e_s2 = 0.0, e_rss = 0.0;
  auto f = [&](double h2) {
    const double logsum = dyn_weights<64>(n, h2, lam, sW, false, &nonpos, s_red);
    dyn_factor<64>(n, D, alt, sW, sA, nullptr, sScal);
    dyn_ell<64>(n, D, alt, sW, sY, sA, sScal[0], logsum, nm.prior_a, nm.prior_b, nm.reml, sV, &e_ell, &e_s2, &e_rss);
    return -e_ell;
  };
  const double hx = dyn_brent_search(f, nm.optim_interval < 1 ? 1 : nm.optim_interval, &hit_max);
  const double h0 = h2null[tr];
  auto closing = [&](double h2, const DynCols& dc, int d) {
    double logsum;
    if (true_w) logsum = dyn_weights<64, false>(n, h2, lam, sW, false, nullptr, s_red);
    else logsum = 0.5 * dyn_weights<64, true>(n, h2, lam, sW, false, nullptr, s_red);
    dyn_factor<64>(n, d, dc, sW, sA, nullptr, sScal);
    dyn_ell<64>(n, d, dc, sW, sY, sA, sScal[0], logsum, nm.prior_a, nm.prior_b, 0, sV, &e_ell, &e_s2, &e_rss);
    return e_ell;
  };
  const double e1 = closing(hx, alt, D);
  const double e0 = closing(h0, nul, c);
  if (threadIdx.x == 0) {
    lod[i] = (e1 - e0) / log(10.0);
    h2each[i] = hx;
    if (hit_max) atomicAdd((unsigned long long*)&stat[ST_BRENT_MAXIT], 1ull);
    if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
  }
}

// ---- factors of a LIST of h2 values (the grid of null-grid / alt-grid, the single h2 of the permutation test): one workgroup
// per value.  fac[g]: L (DYN_NA) | L^-1 (DYN_NA) | {ln det A, sum ln t, nonpos, rho}  ---------------------------------------------
constexpr int DYN_FS = 2 * DYN_NA + 8;
__global__ void __launch_bounds__(256) k_dyn_factor(int n, int c, const double* __restrict__ Z0, const double* __restrict__ lam,
                                                    const double* __restrict__ h2list, int absw, double* __restrict__ fac) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ double sA[DYN_NA], sLi[DYN_NA], sScal[2], s_red[5];
  double* sW = sh;
  int nonpos = 0;
  const double logsum = dyn_weights<256>(n, h2list[blockIdx.x], lam, sW, absw != 0, &nonpos, s_red);
  dyn_factor<256>(n, c, Z0, sW, sA, sLi, sScal);
  double* o = fac + (size_t)blockIdx.x * DYN_FS;
  const int na = c * (c + 1) / 2;
  for (int e = threadIdx.x; e < na; e += 256) { o[e] = sA[e]; o[DYN_NA + e] = sLi[e]; }
  const int anybad = __syncthreads_or(nonpos);
  if (threadIdx.x == 0) { o[2 * DYN_NA] = sScal[0]; o[2 * DYN_NA + 1] = logsum; o[2 * DYN_NA + 2] = anybad ? 1.0 : 0.0; o[2 * DYN_NA + 3] = sScal[1]; }
}

// Ell[g, j] over the grid and the first arg-max (wls_multivar, find_optim_h2): one wave per trait, the factors from k_dyn_factor
// (likelihood weights: absw = 0)
__global__ void __launch_bounds__(64) k_dyn_grid(NullModel nm, const double* __restrict__ Yt, int64_t ldy, int64_t m,
                                                 const double* __restrict__ Z0, const double* __restrict__ lam,
                                                 const double* __restrict__ grid, int ngrid, const double* __restrict__ fac,
                                                 double* __restrict__ EllTab, int* __restrict__ h2idx, double* __restrict__ h2out,
                                                 int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ double sA[DYN_NA], sV[CMAX + 1], s_red[2];
  const int n = nm.n, c = nm.c, na = c * (c + 1) / 2;
  double* sW = sh;
  double* sY = sh + n;
  const int64_t j = blockIdx.x;
  for (int k = threadIdx.x; k < n; k += 64) sY[k] = Yt[(int64_t)k * ldy + j];
  int best = 0, nonpos = 0;
  double bestv = -INFINITY;
  for (int g = 0; g < ngrid; ++g) {
    const double* fg = fac + (size_t)g * DYN_FS;
    __syncthreads();
    for (int e = threadIdx.x; e < na; e += 64) sA[e] = fg[e];
    (void)dyn_weights<64>(n, grid[g], lam, sW, false, &nonpos, s_red);
    double ell, s2, rss;
    dyn_ell<64>(n, c, Z0, sW, sY, sA, fg[2 * DYN_NA], fg[2 * DYN_NA + 1], nm.prior_a, nm.prior_b, nm.reml, sV, &ell, &s2, &rss);
    if (threadIdx.x == 0 && EllTab) EllTab[j * (int64_t)ngrid + g] = ell;
    if (g == 0 || ell > bestv) { bestv = ell; best = g; }
  }
  if (threadIdx.x == 0) {
    if (h2idx) h2idx[j] = best;
    if (h2out) h2out[j] = grid[best];
    if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
  }
}

// A-side panels of one trait at its own h2 (k_panels restated: panel 0 = w (y - Z0 beta) / sqrt(yy); full: panel 1 = w, panels
// 2 + q = w (Z0 L^-T)_q), one wave per panel column; padding columns are zero-filled.  flag_list != nullptr: the conditioning
// guard's verdict on the way (rho below rho_min -> the trait goes on the list, count in stat[ST_ILLCOND]).
__global__ void __launch_bounds__(64) k_dyn_panels(NullModel nm, const double* __restrict__ Yt, int64_t ldy, int64_t m,
                                                   const double* __restrict__ Z0, const double* __restrict__ lam,
                                                   const double* __restrict__ h2v, int full, double* __restrict__ P, int64_t ldp,
                                                   int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ double sA[DYN_NA], sLi[DYN_NA], sV[CMAX + 1], sB[CMAX], sScal[2], s_red[2];
  const int n = nm.n, npad = nm.npad, c = nm.c;
  const int64_t j = blockIdx.x;
  const int64_t pstride = (int64_t)npad * ldp;
  const int np = full ? 2 + c : 1;
  if (j >= m) {
    for (int e = threadIdx.x; e < np * npad; e += 64) P[(int64_t)(e / npad) * pstride + (int64_t)(e % npad) * ldp + j] = 0.0;
    return;
  }
  double* sW = sh;
  double* sY = sh + n;
  for (int k = threadIdx.x; k < n; k += 64) sY[k] = Yt[(int64_t)k * ldy + j];
  __syncthreads();
  (void)dyn_weights<64>(n, h2v[j], lam, sW, true, nullptr, s_red);
  dyn_factor<64>(n, c, Z0, sW, sA, sLi, sScal);
  double ell, s2, rss;
  dyn_ell<64>(n, c, Z0, sW, sY, sA, 0.0, 0.0, 0.0, 0.0, 0, sV, &ell, &s2, &rss);     // only rss = yy is used; leaves sV = Z0'Wy
  // beta = L^-T L^-1 v (every thread, redundantly; sV was released by dyn_ell's closing barrier only for WRITING: it still holds v)
  {
    double tq[CMAX];
    for (int q = 0; q < c; ++q) {
      double sacc = 0.0;
      for (int r = 0; r <= q; ++r) sacc = fma(sLi[q * (q + 1) / 2 + r], sV[r], sacc);
      tq[q] = sacc;
    }
    for (int q = threadIdx.x; q < c; q += 64) {
      double sacc = 0.0;
      for (int u = q; u < c; ++u) sacc = fma(sLi[u * (u + 1) / 2 + q], tq[u], sacc);
      sB[q] = sacc;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && !(sqrt(fabs(rss)) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
  const double isy = 1.0 / sqrt(rss);
  for (int k = threadIdx.x; k < npad; k += 64) {
    double p0 = 0.0, w = 0.0;
    if (k < n) {
      w = sW[k];
      double res = sY[k];
      for (int q = 0; q < c; ++q) res = fma(-sB[q], Z0[(size_t)q * n + k], res);
      p0 = w * res * isy;
    }
    P[(int64_t)k * ldp + j] = p0;
    if (full) {
      P[pstride + (int64_t)k * ldp + j] = w;
      for (int q = 0; q < c; ++q) {
        double sacc = 0.0;
        if (k < n) for (int r = 0; r <= q; ++r) sacc = fma(sLi[q * (q + 1) / 2 + r], Z0[(size_t)r * n + k], sacc);
        P[(int64_t)(2 + q) * pstride + (int64_t)k * ldp + j] = w * sacc;
      }
    }
  }
}

// isx[g][i] = 1 / || P_g (sqrt(w_g) x_i) || (k_isx restated): grid (markers / 256, grid points), the factors from k_dyn_factor
// with the scan's |w| convention (absw = 1)
__global__ void __launch_bounds__(256) k_dyn_isx(int n, int c, const double* __restrict__ Xt, int64_t ldx, int64_t p,
                                                 const double* __restrict__ Z0, const double* __restrict__ lam,
                                                 const double* __restrict__ grid, const double* __restrict__ fac,
                                                 double* __restrict__ isx, int64_t ld_isx, int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ double sLi[DYN_NA];
  double* sW = sh;
  const int g = blockIdx.y, na = c * (c + 1) / 2;
  const double h2 = grid[g];
  const double delta = h2 / (1.0 - h2);
  for (int k = threadIdx.x; k < n; k += 256) sW[k] = fabs(1.0 / fma(delta, lam[k], 1.0));
  const double* fg = fac + (size_t)g * DYN_FS + DYN_NA;
  for (int e = threadIdx.x; e < na; e += 256) sLi[e] = fg[e];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ld_isx) return;
  double out = 0.0;
  if (i < p) {
    double sxx = 0.0, sxz[CMAX];
#pragma unroll
    for (int q = 0; q < CMAX; ++q) sxz[q] = 0.0;
    for (int k = 0; k < n; ++k) {
      const double x = Xt[(int64_t)k * ldx + i];
      const double wx = sW[k] * x;
      sxx = fma(wx, x, sxx);
#pragma unroll
      for (int q = 0; q < CMAX; ++q)
        if (q < c) sxz[q] = fma(wx, Z0[(size_t)q * n + k], sxz[q]);
    }
    double uu = 0.0;
#pragma unroll
    for (int q = 0; q < CMAX; ++q)
      if (q < c) {
        double u = 0.0;
#pragma unroll
        for (int r = 0; r < CMAX; ++r)
          if (r <= q) u = fma(sLi[q * (q + 1) / 2 + r], sxz[r], u);
        uu = fma(u, u, uu);
      }
    const double xx = sxx - uu;
    if (!(sqrt(fabs(xx)) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
    out = 1.0 / sqrt(xx);
  }
  isx[(int64_t)g * ld_isx + i] = out;
}

// Permutation panels (k_perm_r0 / k_perm_coef / k_perm_fill restated; src/scan.jl:521-542): the factor of the fitted h2 comes
// from k_dyn_factor (ONE factor serves every permutation: A does not depend on the column).
//   mode 0: r0 = sqrt(w) (y - Z0 beta_w) of the trait in Yt[:, 0]           (one workgroup)
//   mode 1: panel column b = sqrt(w) (v - sqrt(w) Z0 beta_b) / |v|,  v = pi_b(r0) (orig: the identity), one wave per column
__global__ void __launch_bounds__(64) k_dyn_perm(int mode, NullModel nm, const double* __restrict__ Yt, int64_t ldy,
                                                 const double* __restrict__ Z0, const double* __restrict__ lam,
                                                 const double* __restrict__ h2p, const double* __restrict__ fac,
                                                 const int32_t* __restrict__ perm, int64_t ncols, int orig, double* __restrict__ r0buf,
                                                 double* __restrict__ P, int64_t ldp, int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ double sLi[DYN_NA], sV[CMAX], sB[CMAX];
  const int n = nm.n, npad = nm.npad, c = nm.c, na = c * (c + 1) / 2;
  double* sS = sh;          // sqrt(w)
  double* sX = sh + n;      // the vector being projected
  const int64_t b = blockIdx.x;
  if (mode == 1 && b >= ncols) {
    for (int k = threadIdx.x; k < npad; k += 64) P[(int64_t)k * ldp + b] = 0.0;
    return;
  }
  const double h2 = h2p[0];
  const double delta = h2 / (1.0 - h2);
  for (int e = threadIdx.x; e < na; e += 64) sLi[e] = fac[DYN_NA + e];
  for (int k = threadIdx.x; k < n; k += 64) {
    sS[k] = sqrt(1.0 / fma(delta, lam[k], 1.0));
    sX[k] = (mode == 0) ? Yt[(int64_t)k * ldy] : r0buf[orig ? k : perm[b * (int64_t)n + k]];
  }
  __syncthreads();
  // g_q = (sqrt(w) z_q)' (sqrt(w) y) [mode 0: the weighted normal equations of y]  or  (sqrt(w) z_q)' v [mode 1]
  for (int q = threadIdx.x; q < c; q += 64) {
    const double* zq = Z0 + (size_t)q * n;
    double acc = 0.0;
    for (int k = 0; k < n; ++k) acc = fma(sS[k] * zq[k], (mode == 0 ? sS[k] : 1.0) * sX[k], acc);
    sV[q] = acc;
  }
  __syncthreads();
  {
    double tq[CMAX];
    for (int q = 0; q < c; ++q) {
      double sacc = 0.0;
      for (int r = 0; r <= q; ++r) sacc = fma(sLi[q * (q + 1) / 2 + r], sV[r], sacc);
      tq[q] = sacc;
    }
    for (int q = threadIdx.x; q < c; q += 64) {
      double sacc = 0.0;
      for (int u = q; u < c; ++u) sacc = fma(sLi[u * (u + 1) / 2 + q], tq[u], sacc);
      sB[q] = sacc;
    }
  }
  __syncthreads();
  if (mode == 0) {
    for (int k = threadIdx.x; k < n; k += 64) {
      double v = sX[k];
      for (int q = 0; q < c; ++q) v = fma(-sB[q], Z0[(size_t)q * n + k], v);
      r0buf[k] = sS[k] * v;
    }
    return;
  }
  double rr = 0.0;
  for (int k = threadIdx.x; k < n; k += 64) rr = fma(sX[k], sX[k], rr);
  rr = group_sum<64>(rr);
  if (orig && threadIdx.x == 0 && !(sqrt(rr) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
  const double inr = 1.0 / sqrt(rr);
  for (int k = threadIdx.x; k < npad; k += 64) {
    double out = 0.0;
    if (k < n) {
      double v = sX[k];
      for (int q = 0; q < c; ++q) v = fma(-sB[q], sS[k] * Z0[(size_t)q * n + k], v);
      out = sS[k] * v * inr;
    }
    P[(int64_t)k * ldp + b] = out;
  }
}

// the conditioning guard's flag for run-time c: one wave per trait
__global__ void __launch_bounds__(64) k_dyn_illcond(int n, int c, int64_t m, const double* __restrict__ Z0, const double* __restrict__ lam,
                                                    const double* __restrict__ h2v, double rho_min, int* __restrict__ list, int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ double sA[DYN_NA], sScal[2], s_red[2];
  const int64_t j = blockIdx.x;
  (void)dyn_weights<64>(n, h2v[j], lam, sh, true, nullptr, s_red);
  dyn_factor<64>(n, c, Z0, sh, sA, nullptr, sScal);
  if (threadIdx.x == 0 && !(sScal[1] >= rho_min)) {
    const unsigned long long slot = atomicAdd((unsigned long long*)&stat[ST_ILLCOND], 1ull);
    list[slot] = (int)j;
  }
}

// ---- launchers (kernels_prep.hip's launch_* route c > CTPL here) -----------------------------------------------------------------
static int dyn_lds(const NullModel& nm, int vecs) { return (int)(sizeof(double) * (size_t)vecs * nm.n); }
#define DYN_LDS_ATTR(kern, bytes) do { if ((bytes) > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); } while (0)

int launch_dyn_brent(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0, const double* lam,
                     double* h2, double* sigma2, double* ell, int64_t* stat) {
  if (m <= 0) return BLMM_OK;
  const int lds = dyn_lds(nm, 2);
  DYN_LDS_ATTR(k_dyn_brent, lds);
  hipLaunchKernelGGL(k_dyn_brent, dim3((unsigned)m), dim3(64), lds, ctx->stream, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat);
  KCHECK();
  return BLMM_OK;
}

int launch_dyn_alt_brent(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Xt, int64_t ldx, int64_t p,
                         const double* Z0, const double* lam, const double* h2null, int true_w, double* lod, double* h2each,
                         int64_t* stat, int64_t m, int64_t ldL, int64_t ldH) {
  if (p <= 0 || m <= 0) return BLMM_OK;
  if (nm.c + 1 > CMAX) return fail(ctx, BLMM_ERR_UNSUPPORTED, "scan(...; assumption = \"alt\"): at most 31 null covariates (incl. intercept): the per-marker design [Z0 x] has c + 1 <= 32 columns");
  if (p > 0x7fffffffLL) return fail(ctx, BLMM_ERR_INVALID, "problem too large for one launch");
  const int lds = dyn_lds(nm, 3);
  if (lds > 150 * 1024) return fail(ctx, BLMM_ERR_UNSUPPORTED, "scan_alt: n too large for the LDS-resident trait");
  DYN_LDS_ATTR(k_dyn_alt_brent, lds);
  for (int64_t t0 = 0; t0 < m; t0 += 65535) {       // gridDim.y <= 65535
    const int64_t mt = (m - t0 < 65535) ? m - t0 : 65535;
    hipLaunchKernelGGL(k_dyn_alt_brent, dim3((unsigned)p, (unsigned)mt), dim3(64), lds, ctx->stream, nm, Yt, ldy, Xt, ldx, p, Z0, lam,
                       h2null, true_w, lod, h2each, stat, t0, ldL, ldH);
    KCHECK();
  }
  return BLMM_OK;
}

// factors of ngrid h2 values (device list) into ctx->dynFac
static int dyn_factors(blmm_ctx* ctx, const NullModel& nm, const double* Z0, const double* lam, const double* h2list_dev, int ngrid,
                       int absw, double** fac) {
  int rc = ensure(ctx, ctx->dynFac, sizeof(double) * (size_t)DYN_FS * (size_t)(ngrid > 0 ? ngrid : 1) * 2);
  if (rc) return rc;
  *fac = ptr<double>(ctx->dynFac) + (absw ? (size_t)DYN_FS * ngrid : 0);     // two halves: likelihood weights | scan weights
  const int lds = dyn_lds(nm, 1);
  DYN_LDS_ATTR(k_dyn_factor, lds);
  hipLaunchKernelGGL(k_dyn_factor, dim3((unsigned)ngrid), dim3(256), lds, ctx->stream, nm.n, nm.c, Z0, lam, h2list_dev, absw, *fac);
  KCHECK();
  return BLMM_OK;
}

int launch_dyn_loglik_grid(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                           const double* lam, const double* grid_dev, int ngrid, double* EllTab, int* h2idx, double* h2, int64_t* stat) {
  if (m <= 0) return BLMM_OK;
  double* fac = nullptr;
  int rc = dyn_factors(ctx, nm, Z0, lam, grid_dev, ngrid, 0, &fac);
  if (rc) return rc;
  const int lds = dyn_lds(nm, 2);
  DYN_LDS_ATTR(k_dyn_grid, lds);
  hipLaunchKernelGGL(k_dyn_grid, dim3((unsigned)m), dim3(64), lds, ctx->stream, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, fac, EllTab, h2idx, h2, stat);
  KCHECK();
  return BLMM_OK;
}

int launch_dyn_panels(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0, const double* lam,
                      const double* h2, int full, double* panels, int64_t ldp, int64_t* stat) {
  const int lds = dyn_lds(nm, 2);
  DYN_LDS_ATTR(k_dyn_panels, lds);
  hipLaunchKernelGGL(k_dyn_panels, dim3((unsigned)ldp), dim3(64), lds, ctx->stream, nm, Yt, ldy, m, Z0, lam, h2, full, panels, ldp, stat);
  KCHECK();
  return BLMM_OK;
}

int launch_dyn_isx(blmm_ctx* ctx, const NullModel& nm, const double* Xt, int64_t ldx, int64_t p, const double* Z0, const double* lam,
                   const double* grid_dev, int ngrid, double* isx, int64_t ld_isx, int64_t* stat) {
  double* fac = nullptr;
  int rc = dyn_factors(ctx, nm, Z0, lam, grid_dev, ngrid, 1, &fac);
  if (rc) return rc;
  const int lds = dyn_lds(nm, 1);
  DYN_LDS_ATTR(k_dyn_isx, lds);
  hipLaunchKernelGGL(k_dyn_isx, dim3((unsigned)((ld_isx + 255) / 256), (unsigned)ngrid), dim3(256), lds, ctx->stream, nm.n, nm.c, Xt, ldx, p, Z0,
                     lam, grid_dev, fac, isx, ld_isx, stat);
  KCHECK();
  return BLMM_OK;
}

int launch_dyn_perm(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Z0, const double* lam, const double* h2,
                    const int32_t* perm, int64_t ncols, int orig, double* r0, double* panel, int64_t ldp, int64_t* stat) {
  double* fac = nullptr;
  int rc = dyn_factors(ctx, nm, Z0, lam, h2, 1, 0, &fac);          // sqrt(w): the weights themselves (src/scan.jl:521-527)
  if (rc) return rc;
  const int lds = dyn_lds(nm, 2);
  DYN_LDS_ATTR(k_dyn_perm, lds);
  if (orig) hipLaunchKernelGGL(k_dyn_perm, dim3(1), dim3(64), lds, ctx->stream, 0, nm, Yt, ldy, Z0, lam, h2, fac, perm, ncols, orig, r0, panel, ldp, stat);
  hipLaunchKernelGGL(k_dyn_perm, dim3((unsigned)ldp), dim3(64), lds, ctx->stream, 1, nm, Yt, ldy, Z0, lam, h2, fac, perm, ncols, orig, r0, panel, ldp, stat);
  KCHECK();
  return BLMM_OK;
}

double illcond_rho_min(const blmm_ctx* ctx) {
  // tuning key "illcond_rho" (tests: 2 flags every trait with c >= 2, so that the QR-grade kernel is compared with the oracle
  // as a whole; 0 switches the guard off)
  const char* e = dev_env("BLMM_ILLCOND_RHO");
  return e ? atof(e) : ctx->tune.illcond_rho;
}

int launch_illcond_flag(blmm_ctx* ctx, const NullModel& nm, int64_t m, const double* Z0, const double* lam, const double* h2,
                        int* list, int64_t* stat) {
  if (m <= 0 || nm.c < 2) return BLMM_OK;
  const double rho = illcond_rho_min(ctx);
  if (!(rho > 0.0)) return BLMM_OK;
  const unsigned blocks = (unsigned)((m + 255) / 256);
  const size_t lds = sizeof(double) * (size_t)nm.n * (1 + nm.c);
  if (nm.c <= CTPL && lds > 140 * 1024) return BLMM_OK;   // (1 + c) n doubles beyond one CU's LDS: the guard does not apply
  if (nm.c > CTPL) {
    const int l1 = dyn_lds(nm, 1);
    DYN_LDS_ATTR(k_dyn_illcond, l1);
    hipLaunchKernelGGL(k_dyn_illcond, dim3((unsigned)m), dim3(64), l1, ctx->stream, nm.n, nm.c, m, Z0, lam, h2, rho, list, stat);
    KCHECK();
    return BLMM_OK;
  }
#define IF(C) do { if (lds > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_illcond_flag<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k_illcond_flag<C>, dim3(blocks), dim3(256), lds, ctx->stream, nm.n, m, Z0, lam, h2, rho, list, stat); } while (0)
  switch (nm.c) {
    BLMM_FOR_EACH_C(IF)
    default: return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
  }
#undef IF
  KCHECK();
  return BLMM_OK;
}

int launch_scan_qr(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Xt, int64_t ldx, int64_t p,
                   const double* Z0, const double* lam, const double* h2, const int* list, double* L, int64_t ldL,
                   int64_t* stat) {
  if (p <= 0 || nm.c < 2 || !(illcond_rho_min(ctx) > 0.0)) return BLMM_OK;
  const size_t per = (size_t)(nm.c + 2) * nm.n;
  const unsigned grid = (unsigned)(2 * (ctx->num_cus > 0 ? ctx->num_cus : 256));
  double* slab = nullptr;
  size_t lds = sizeof(double) * per;
  if (lds > 64 * 1024) {
    int rc = ensure(ctx, ctx->qrSlab, sizeof(double) * per * grid);
    if (rc) return rc;
    slab = ptr<double>(ctx->qrSlab);
    lds = 0;
  }
  if (nm.c <= 8) {
    if (lds > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan_qr<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_scan_qr<8>, dim3(grid), dim3(256), lds, ctx->stream, nm.n, nm.c, Yt, ldy, Xt, ldx, p, Z0, lam, h2, list, slab, L, ldL, stat, ctx->pv_cur, ctx->pv_cur_ld, ptr<double>(ctx->pvtab));
  } else {
    if (lds > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan_qr<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_scan_qr<32>, dim3(grid), dim3(256), lds, ctx->stream, nm.n, nm.c, Yt, ldy, Xt, ldx, p, Z0, lam, h2, list, slab, L, ldL, stat, ctx->pv_cur, ctx->pv_cur_ld, ptr<double>(ctx->pvtab));
  }
  KCHECK();
  return BLMM_OK;
}

}  // namespace blmm
