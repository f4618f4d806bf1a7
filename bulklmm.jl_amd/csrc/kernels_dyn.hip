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
                                                 double* slab, double* __restrict__ L, int64_t ldL, int64_t* stat) {
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
    }
  }
}

double illcond_rho_min() {
  // BLMM_ILLCOND_RHO overrides the threshold (tests: 2 flags every trait with c >= 2, so that the QR-grade kernel is compared
  // with the oracle as a whole; 0 switches the guard off)
  const char* e = getenv("BLMM_ILLCOND_RHO");
  return e ? atof(e) : 1e-4;
}

int launch_illcond_flag(blmm_ctx* ctx, const NullModel& nm, int64_t m, const double* Z0, const double* lam, const double* h2,
                        int* list, int64_t* stat) {
  if (m <= 0 || nm.c < 2) return BLMM_OK;
  const double rho = illcond_rho_min();
  if (!(rho > 0.0)) return BLMM_OK;
  const unsigned blocks = (unsigned)((m + 255) / 256);
  const size_t lds = sizeof(double) * (size_t)nm.n * (1 + nm.c);
  if (lds > 140 * 1024) return BLMM_OK;         // (1 + c) n doubles beyond one CU's LDS: the guard does not apply
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
  if (p <= 0 || nm.c < 2 || !(illcond_rho_min() > 0.0)) return BLMM_OK;
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
    hipLaunchKernelGGL(k_scan_qr<8>, dim3(grid), dim3(256), lds, ctx->stream, nm.n, nm.c, Yt, ldy, Xt, ldx, p, Z0, lam, h2, list, slab, L, ldL, stat);
  } else {
    if (lds > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan_qr<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_scan_qr<32>, dim3(grid), dim3(256), lds, ctx->stream, nm.n, nm.c, Yt, ldy, Xt, ldx, p, Z0, lam, h2, list, slab, L, ldL, stat);
  }
  KCHECK();
  return BLMM_OK;
}

}  // namespace blmm
