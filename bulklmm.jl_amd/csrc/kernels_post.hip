// kernels_post.hip -- on-device consumers of the LOD matrix, so that the p x m matrix (2 GB at BXD size) does not have
// to cross PCIe before it is reduced to what the user looks at:
//   * -log10 p-values           lod2log10p, src/util.jl:199-206 (bulkscan / scan `output_pvals`, src/bulkscan.jl:154-157)
//   * threshold filter          LOD > t -> sparse (marker, trait, LOD) triplets (README.md:354-359, plot_eQTL threshold)
//   * permutation thresholds    quantiles of the per-permutation maxima, get_thresholds,
//                               src/analysis_helpers/single_trait_analysis.jl:13-23
#include "blmm_internal.h"
#include "fastmath.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace blmm {

#define KCHECK()                                                                                      \
  do {                                                                                                \
    hipError_t e__ = hipGetLastError();                                                               \
    if (e__ != hipSuccess) return fail(ctx, BLMM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e__)); \
  } while (0)

// ln of the upper regularised incomplete gamma function Q(a, z), a > 0, z >= 0: logccdf(Chisq(df), x) = lnQ(df/2, x/2).
// Series for z < a + 1 (Q = 1 - P), modified Lentz continued fraction beyond, both assembled in log space so that tiny
// tail probabilities (LOD of several hundred) do not underflow.
__device__ double ln_gamma_q(double a, double z) {
  if (!(z > 0.0)) return (z == z) ? 0.0 : z;       // Q(a, 0) = 1; NaN stays NaN
  if (isinf(z)) return -INFINITY;
  const double lg = lgamma(a);
  if (z < a + 1.0) {
    double ap = a, del = 1.0 / a, sum = del;
    for (int it = 0; it < 500; ++it) {
      ap += 1.0; del *= z / ap; sum += del;
      if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    const double lnP = -z + a * log(z) - lg + log(sum);
    return log1p(-exp(lnP));
  }
  const double tiny = 1e-300;
  double b = z + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 500; ++i) {
    const double an = -(double)i * ((double)i - a);
    b += 2.0;
    d = an * d + b; if (fabs(d) < tiny) d = tiny;
    c = b + an / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return -z + a * log(z) - lg + log(h);
}

// -log10 p of a LOD score under chi^2_df (src/util.jl:199-206): lrs = lod * 2 ln 10; -logccdf(Chisq(df), lrs) / ln 10.
// df = 1: ccdf = erfc(sqrt(lrs / 2)), taken through erfcx beyond 1 so that the logarithm never sees an underflowed erfc.
__device__ __forceinline__ double lod_to_log10p(double lod, int df) {
  const double ln10 = 2.302585092994046;
  if (df == 1) {
    const double t = lod * ln10;                 // lrs / 2
    if (!(t > 0.0)) return (t == t) ? 0.0 : t;
    const double x = sqrt(t);
    const double lnp = (x < 1.0) ? log(erfc(x)) : (log(erfcx(x)) - t);
    return -lnp / ln10;
  }
  return -ln_gamma_q(0.5 * (double)df, lod * ln10) / ln10;
}

__global__ void __launch_bounds__(256) k_lod2log10p(const double* __restrict__ L, int64_t p, int64_t m, int64_t ldL, int df,
                                                    double* __restrict__ P, int64_t ldP) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p) return;
  for (int64_t j = blockIdx.y; j < m; j += gridDim.y) P[j * ldP + i] = lod_to_log10p(L[j * ldL + i], df);
}

// One degree of freedom (the default of `output_pvals`): the bucketed-polynomial form of fastmath.h (table in LDS), ~20 fp64
// operations per value -- the pass is bound by its 16 B/value of HBM traffic, where the erfc / erfcx / log route above costs ~150.
__global__ void __launch_bounds__(256) k_lod2log10p1(const double* __restrict__ L, int64_t p, int64_t m, int64_t ldL,
                                                     const double* __restrict__ pvtab, double* __restrict__ P, int64_t ldP) {
  __shared__ dpair s_pv[BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2)];
  const dpair* g = reinterpret_cast<const dpair*>(pvtab);
  for (int i = threadIdx.x; i < BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2); i += 256) s_pv[i] = g[i];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p) return;
  for (int64_t j = blockIdx.y; j < m; j += gridDim.y) P[j * ldP + i] = fast_log10p1(L[j * ldL + i], s_pv);
}

int launch_lod2log10p(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, int df, double* dP, int64_t ldP) {
  if (p <= 0 || m <= 0) return BLMM_OK;
  const char* le = dev_env("BLMM_PVAL_LIBM");                         // tuning key "pval_libm": erfc / erfcx / log for df = 1 too
  const bool exact_route = le ? le[0] == '1' : ctx->tune.pval_libm != 0;
  if (df == 1 && ctx->pvtab.p && !exact_route) {
    // few, long columns walks: the table is staged once per workgroup
    dim3 grid1((unsigned)((p + 255) / 256), (unsigned)(m < 64 ? m : 64));
    hipLaunchKernelGGL(k_lod2log10p1, grid1, dim3(256), 0, ctx->stream, dL, p, m, ldL, ptr<double>(ctx->pvtab), dP, ldP);
    KCHECK();
    return BLMM_OK;
  }
  dim3 grid((unsigned)((p + 255) / 256), (unsigned)(m < 4096 ? m : 4096));
  hipLaunchKernelGGL(k_lod2log10p, grid, dim3(256), 0, ctx->stream, dL, p, m, ldL, df, dP, ldP);
  KCHECK();
  return BLMM_OK;
}

// Threshold filter: every (marker i, trait j) with L[i, j] > thr (NaN never passes) as a triplet.  One wave-aggregated
// atomic per wave instruction reserves the output slots; entries beyond `cap` are counted but not stored, so the caller
// can size a retry from *count.  The order of the triplets is unspecified (the host wrappers sort by (trait, marker)).
__global__ void __launch_bounds__(256) k_threshold(const double* __restrict__ L, int64_t p, int64_t m, int64_t ldL, double thr,
                                                   int64_t cap, int32_t* __restrict__ oi, int32_t* __restrict__ oj,
                                                   double* __restrict__ ol, unsigned long long* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (int64_t j = blockIdx.y; j < m; j += gridDim.y) {
    const double v = (i < p) ? L[j * ldL + i] : -INFINITY;
    const bool hit = v > thr;
    const unsigned long long mask = __ballot(hit);
    if (mask == 0ull) continue;
    unsigned long long base = 0;
    if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(count, (unsigned long long)__builtin_popcountll(mask));
    base = __shfl(base, (int)__builtin_ctzll(mask), 64);
    if (hit) {
      const unsigned long long slot = base + (unsigned long long)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
      if ((int64_t)slot < cap) { oi[slot] = (int32_t)i; oj[slot] = (int32_t)j; ol[slot] = v; }
    }
  }
}

int launch_threshold(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, double thr, int64_t cap,
                     int32_t* di, int32_t* dj, double* dlod, int64_t* dcount) {
  BLMM_HIP(hipMemsetAsync(dcount, 0, sizeof(int64_t), ctx->stream));
  if (p <= 0 || m <= 0) return BLMM_OK;
  dim3 grid((unsigned)((p + 255) / 256), (unsigned)(m < 4096 ? m : 4096));
  hipLaunchKernelGGL(k_threshold, grid, dim3(256), 0, ctx->stream, dL, p, m, ldL, thr, cap, di, dj, dlod,
                     reinterpret_cast<unsigned long long*>(dcount));
  KCHECK();
  return BLMM_OK;
}

// Second pass of the reduce-in-epilogue scan (RedArgs, blmm_internal.h): the scan kernels left, per trait and 64-marker slot, the
// slot's maximum LOD and its marker; one thread per trait walks the slots in marker order with k_colmax's rule (strictly larger,
// or equal at the lower marker; NaN never), so (max, argmax) equal k_colmax's on the stored matrix bit for bit.
__global__ void __launch_bounds__(256) k_red_final(const double* __restrict__ pmax, const int* __restrict__ parg, int64_t ldm, int nslot,
                                                   int64_t m, double* __restrict__ mx, int64_t* __restrict__ arg) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  double best = -INFINITY; int64_t bi = -1;
  for (int s = 0; s < nslot; ++s) {
    const double v = pmax[(int64_t)s * ldm + j];
    const int64_t i = parg[(int64_t)s * ldm + j];
    if (v > best || (v == best && i >= 0 && (bi < 0 || i < bi))) { best = v; bi = i; }
  }
  if (mx) mx[j] = best;
  if (arg) arg[j] = bi;
}

int launch_red_final(blmm_ctx* ctx, const RedArgs& r, int nslot, int64_t m, double* mx, int64_t* arg) {
  if (m <= 0 || (!mx && !arg)) return BLMM_OK;
  hipLaunchKernelGGL(k_red_final, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, r.pmax, r.parg, r.ldm, nslot, m, mx, arg);
  KCHECK();
  return BLMM_OK;
}

// column j of the resident matrix -> out column k (blmm_last_lod_columns)
__global__ void __launch_bounds__(256) k_gather_cols(const double* __restrict__ L, int64_t p, int64_t ldL, const int64_t* __restrict__ cols,
                                                     double* __restrict__ out) {
  const int64_t j = cols[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p; i += (int64_t)gridDim.x * 256) out[(int64_t)blockIdx.y * p + i] = L[j * ldL + i];
}

// ---- quantiles of a device vector (the per-permutation maxima): bitonic sort + linear interpolation ------------------
// NaN sorts last.  npow = the next power of two >= count; the tail is padded with +inf.
__device__ __forceinline__ bool key_less(double a, double b) {   // total order with NaN as the largest key
  if (a != a) return false;
  if (b != b) return true;
  return a < b;
}
__global__ void __launch_bounds__(1024) k_bitonic_lds(double* __restrict__ v, int npow) {   // npow <= 16384: one workgroup
  extern __shared__ double sv[];
  for (int e = threadIdx.x; e < npow; e += blockDim.x) sv[e] = v[e];
  __syncthreads();
  for (int k = 2; k <= npow; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = threadIdx.x; e < npow; e += blockDim.x) {
        const int x = e ^ j;
        if (x > e) {
          const double a = sv[e], b = sv[x];
          const bool up = (e & k) == 0;
          if (up ? key_less(b, a) : key_less(a, b)) { sv[e] = b; sv[x] = a; }
        }
      }
      __syncthreads();
    }
  for (int e = threadIdx.x; e < npow; e += blockDim.x) v[e] = sv[e];
}
__global__ void __launch_bounds__(256) k_bitonic_step(double* __restrict__ v, int64_t npow, int64_t k, int64_t j) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= npow) return;
  const int64_t x = e ^ j;
  if (x > e) {
    const double a = v[e], b = v[x];
    const bool up = (e & k) == 0;
    if (up ? key_less(b, a) : key_less(a, b)) { v[e] = b; v[x] = a; }
  }
}
__global__ void k_pad_inf(double* __restrict__ v, int64_t count, int64_t npow) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count && e < npow) v[e] = INFINITY;
}
// Julia's default quantile (type 7): h = (n - 1) q; v[floor(h)] + (h - floor(h)) (v[floor(h) + 1] - v[floor(h)]).
__global__ void k_quantiles(const double* __restrict__ sorted, int64_t count, const double* __restrict__ probs, int nprobs,
                            double* __restrict__ out) {
  const int t = threadIdx.x;
  if (t >= nprobs) return;
  if (count <= 0) { out[t] = NAN; return; }
  double q = probs[t];
  q = q < 0.0 ? 0.0 : (q > 1.0 ? 1.0 : q);
  const double h = (double)(count - 1) * q;
  const int64_t lo = (int64_t)floor(h);
  const int64_t hi = lo + 1 < count ? lo + 1 : count - 1;
  const double a = sorted[lo], b = sorted[hi];
  out[t] = a + (h - (double)lo) * (b - a);
}

// sorts `work` (count values, capacity npow) in place and evaluates the quantiles; probs/out are device arrays
int launch_quantiles(blmm_ctx* ctx, double* work, int64_t count, int64_t npow, const double* dprobs, int nprobs, double* dout) {
  if (npow > count) {
    hipLaunchKernelGGL(k_pad_inf, dim3((unsigned)((npow + 255) / 256)), dim3(256), 0, ctx->stream, work, count, npow);
    KCHECK();
  }
  if (npow <= 16384) {
    const size_t lds = sizeof(double) * (size_t)npow;
    if (lds > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bitonic_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (npow >= 2) { hipLaunchKernelGGL(k_bitonic_lds, dim3(1), dim3(1024), lds, ctx->stream, work, (int)npow); KCHECK(); }
  } else {
    for (int64_t k = 2; k <= npow; k <<= 1)
      for (int64_t j = k >> 1; j > 0; j >>= 1)
        hipLaunchKernelGGL(k_bitonic_step, dim3((unsigned)((npow + 255) / 256)), dim3(256), 0, ctx->stream, work, npow, k, j);
    KCHECK();
  }
  hipLaunchKernelGGL(k_quantiles, dim3(1), dim3(64), 0, ctx->stream, work, count, dprobs, nprobs, dout);
  KCHECK();
  return BLMM_OK;
}

}  // namespace blmm

using namespace blmm;

extern "C" {

int blmm_lod2log10p_dev(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, int64_t chisq_df, double* dP_out,
                        int64_t ldP) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!dL || !dP_out || p < 0 || m < 0 || ldL < p || ldP < p || chisq_df < 1 || chisq_df > 1000000)
    return fail(ctx, BLMM_ERR_INVALID, "lod2log10p: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  return launch_lod2log10p(ctx, dL, p, m, ldL, (int)chisq_df, dP_out, ldP);
}

int blmm_lod_threshold_dev(blmm_ctx* ctx, const double* dL, int64_t p, int64_t m, int64_t ldL, double thr, int64_t cap,
                           int32_t* di_out, int32_t* dj_out, double* dlod_out, int64_t* dcount_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!dL || !dcount_out || p < 0 || m < 0 || ldL < p || cap < 0 || (cap > 0 && (!di_out || !dj_out || !dlod_out)) ||
      p > 0x7fffffffLL || m > 0x7fffffffLL)
    return fail(ctx, BLMM_ERR_INVALID, "lod_threshold: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  return launch_threshold(ctx, dL, p, m, ldL, thr, cap, di_out, dj_out, dlod_out, dcount_out);
}

int blmm_get_thresholds_dev(blmm_ctx* ctx, const double* dLperms, int64_t p, int64_t nperms, int64_t ld, const double* probs,
                            int64_t nprobs, double* thrs_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!dLperms || !probs || !thrs_out || p < 1 || nperms < 1 || ld < p || nprobs < 1 || nprobs > 64)
    return fail(ctx, BLMM_ERR_INVALID, "get_thresholds: bad arguments");
  BLMM_HIP(hipSetDevice(ctx->device));
  int64_t npow = 1;
  while (npow < nperms) npow <<= 1;
  int rc;
  if ((rc = ensure(ctx, ctx->tmpA, sizeof(double) * (size_t)npow))) return rc;
  if ((rc = ensure(ctx, ctx->misc, sizeof(double) * 128))) return rc;
  double* dprobs = ptr<double>(ctx->misc);
  double* dout = dprobs + 64;
  BLMM_HIP(hipMemcpyAsync(dprobs, probs, sizeof(double) * (size_t)nprobs, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = launch_colmax(ctx, dLperms, p, nperms, ld, ptr<double>(ctx->tmpA), nullptr))) return rc;
  if ((rc = launch_quantiles(ctx, ptr<double>(ctx->tmpA), nperms, npow, dprobs, (int)nprobs, dout))) return rc;
  BLMM_HIP(hipMemcpyAsync(thrs_out, dout, sizeof(double) * (size_t)nprobs, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return BLMM_OK;
}

// ---- the same consumers applied to the LOD matrix of the LAST host-pointer call of this context, which is still
// resident in the context's workspace (no second trip of L over PCIe) ------------------------------------------------
int blmm_last_log10p(blmm_ctx* ctx, int64_t chisq_df, double* P_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!P_out || chisq_df < 1) return fail(ctx, BLMM_ERR_INVALID, "last_log10p: bad arguments");
  if (!ctx->last_L || ctx->last_f32) return fail(ctx, BLMM_ERR_INVALID, "last_log10p: no fp64 LOD matrix of a previous host-pointer call is resident");
  BLMM_HIP(hipSetDevice(ctx->device));
  const int64_t p = ctx->last_p, m = ctx->last_m;
  int rc;
  if (ctx->last_P && ctx->last_P_df == chisq_df && ctx->last_P_ld == p) {
    // the call that left L also wrote -log10 p (blmm_set_log10p_output with a library-owned buffer): nothing to compute
    if ((rc = copy_to_host(ctx, P_out, const_cast<double*>(ctx->last_P), sizeof(double) * (size_t)p * m))) return rc;
    BLMM_HIP(hipStreamSynchronize(ctx->stream));
    return BLMM_OK;
  }
  if ((rc = ensure(ctx, ctx->altbuf, sizeof(double) * (size_t)(p > 0 ? p : 1) * (size_t)(m > 0 ? m : 1)))) return rc;
  if ((rc = launch_lod2log10p(ctx, ctx->last_L, p, m, p, (int)chisq_df, ptr<double>(ctx->altbuf), p))) return rc;
  if ((rc = copy_to_host(ctx, P_out, ctx->altbuf.p, sizeof(double) * (size_t)p * m))) return rc;
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return BLMM_OK;
}

int blmm_last_lod_threshold(blmm_ctx* ctx, double thr, int64_t cap, int32_t* i_out, int32_t* j_out, double* lod_out,
                            int64_t* count_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!count_out || cap < 0 || (cap > 0 && (!i_out || !j_out || !lod_out))) return fail(ctx, BLMM_ERR_INVALID, "last_lod_threshold: bad arguments");
  if (!ctx->last_L || ctx->last_f32) return fail(ctx, BLMM_ERR_INVALID, "last_lod_threshold: no fp64 LOD matrix of a previous host-pointer call is resident");
  BLMM_HIP(hipSetDevice(ctx->device));
  const int64_t p = ctx->last_p, m = ctx->last_m;
  int rc;
  const size_t per = sizeof(int32_t) * 2 + sizeof(double);
  if ((rc = ensure(ctx, ctx->altbuf, per * (size_t)(cap > 0 ? cap : 1) + 64))) return rc;
  double* dl = ptr<double>(ctx->altbuf);
  int32_t* di = reinterpret_cast<int32_t*>(dl + (cap > 0 ? cap : 1));
  int32_t* dj = di + (cap > 0 ? cap : 1);
  if ((rc = ensure(ctx, ctx->misc, sizeof(double) * 128))) return rc;
  int64_t* dcount = ptr<int64_t>(ctx->misc);
  if ((rc = launch_threshold(ctx, ctx->last_L, p, m, p, thr, cap, di, dj, dl, dcount))) return rc;
  BLMM_HIP(hipMemcpyAsync(count_out, dcount, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t got = *count_out < cap ? *count_out : cap;
  if (got > 0) {
    BLMM_HIP(hipMemcpyAsync(lod_out, dl, sizeof(double) * (size_t)got, hipMemcpyDeviceToHost, ctx->stream));
    BLMM_HIP(hipMemcpyAsync(i_out, di, sizeof(int32_t) * (size_t)got, hipMemcpyDeviceToHost, ctx->stream));
    BLMM_HIP(hipMemcpyAsync(j_out, dj, sizeof(int32_t) * (size_t)got, hipMemcpyDeviceToHost, ctx->stream));
    BLMM_HIP(hipStreamSynchronize(ctx->stream));
  }
  return BLMM_OK;
}

int blmm_last_dims(const blmm_ctx* ctx, int64_t* p_out, int64_t* m_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (p_out) *p_out = ctx->last_L ? ctx->last_p : 0;
  if (m_out) *m_out = ctx->last_L ? ctx->last_m : 0;
  return ctx->last_L ? BLMM_OK : BLMM_ERR_INVALID;
}

int blmm_last_lod_colmax(blmm_ctx* ctx, double* max_out, int64_t* argmax_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!max_out) return fail(ctx, BLMM_ERR_INVALID, "last_lod_colmax: bad arguments");
  if (!ctx->last_L || ctx->last_f32) return fail(ctx, BLMM_ERR_INVALID, "last_lod_colmax: no fp64 LOD matrix of a previous host-pointer call is resident");
  BLMM_HIP(hipSetDevice(ctx->device));
  const int64_t p = ctx->last_p, m = ctx->last_m;
  if (m <= 0) return BLMM_OK;
  int rc;
  if ((rc = ensure(ctx, ctx->tmpA, sizeof(double) * (size_t)m))) return rc;
  if ((rc = ensure(ctx, ctx->tmpB, sizeof(int64_t) * (size_t)m))) return rc;
  if ((rc = launch_colmax(ctx, ctx->last_L, p, m, p, ptr<double>(ctx->tmpA), ptr<int64_t>(ctx->tmpB)))) return rc;
  BLMM_HIP(hipMemcpyAsync(max_out, ctx->tmpA.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
  if (argmax_out) BLMM_HIP(hipMemcpyAsync(argmax_out, ctx->tmpB.p, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return BLMM_OK;
}

int blmm_last_lod_columns(blmm_ctx* ctx, const int64_t* cols, int64_t ncols, double* out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (ncols < 0 || (ncols > 0 && (!cols || !out))) return fail(ctx, BLMM_ERR_INVALID, "last_lod_columns: bad arguments");
  if (!ctx->last_L || ctx->last_f32) return fail(ctx, BLMM_ERR_INVALID, "last_lod_columns: no fp64 LOD matrix of a previous host-pointer call is resident");
  const int64_t p = ctx->last_p, m = ctx->last_m;
  for (int64_t k = 0; k < ncols; ++k)
    if (cols[k] < 0 || cols[k] >= m) return fail(ctx, BLMM_ERR_INVALID, "last_lod_columns: column index out of range");
  if (ncols == 0 || p == 0) return BLMM_OK;
  BLMM_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->altbuf, sizeof(double) * (size_t)p * (size_t)ncols + sizeof(int64_t) * (size_t)ncols + 64))) return rc;
  double* dout = ptr<double>(ctx->altbuf);
  int64_t* dcols = reinterpret_cast<int64_t*>(dout + (size_t)p * ncols);
  BLMM_HIP(hipMemcpyAsync(dcols, cols, sizeof(int64_t) * (size_t)ncols, hipMemcpyHostToDevice, ctx->stream));
  for (int64_t k0 = 0; k0 < ncols; k0 += 32768) {
    const int64_t nk = ncols - k0 < 32768 ? ncols - k0 : 32768;
    hipLaunchKernelGGL(k_gather_cols, dim3((unsigned)((p + 255) / 256 < 64 ? (p + 255) / 256 : 64), (unsigned)nk), dim3(256), 0, ctx->stream, ctx->last_L, p, p,
                       dcols + k0, dout + (size_t)k0 * p);
  }
  KCHECK();
  if ((rc = copy_to_host(ctx, out, dout, sizeof(double) * (size_t)p * (size_t)ncols))) return rc;
  BLMM_HIP(hipStreamSynchronize(ctx->stream));
  return BLMM_OK;
}

int blmm_last_get_thresholds(blmm_ctx* ctx, const double* probs, int64_t nprobs, double* thrs_out) {
  if (!ctx) return BLMM_ERR_INVALID;
  if (!ctx->last_L || ctx->last_f32) return fail(ctx, BLMM_ERR_INVALID, "last_get_thresholds: no fp64 LOD matrix of a previous host-pointer call is resident");
  return blmm_get_thresholds_dev(ctx, ctx->last_L, ctx->last_p, ctx->last_m, ctx->last_p, probs, nprobs, thrs_out);
}

}  // extern "C"
