// kernels_lowrank.hip -- low-rank form of the per-trait weights for the exact (null-exact) LOD kernel.
//
// Per-trait weights enter the marker-side sums only through  w_j = 1 ./ (delta_j*lambda + 1):
//     Sxx[i,j] = sum_k x_ik^2 w_jk ,      s_q[i,j] = sum_k x_ik z_qk w_jk          
// and the one-parameter family { w(delta) : delta >= 0 } has a numerical rank R far below n (BXD kinship spectrum:
// 23 at 1e-15 of 79; tools/ and DESIGN.md §4.3).  With an orthonormal basis Q (n x R) of that family,
// w_j = Q c_j (c_j = Q'w_j) to rounding, hence
//     Sxx = (Q'(X.^2))' C ,   s_q = (Q'(X.*z_q))' C            -- contractions of length R instead of n.
// Only  num = x' a0  (a0 depends on the trait's y) keeps length n: 2n + 2(1+c)R flops per test instead of 2n(2+c).
// R is found on the device by a greedy pivoted Gram-Schmidt over 256 sampled deltas (stops at a 1e-15 relative
// residual, or at R = n where the form is exact by construction), so one code path serves every kinship; the
// per-trait approximation residual is reported in blmm_status.lowrank_resid.
// The rank of the family over a SEGMENT of the heritability axis is about half that of the whole family (BXD: 11-12 for each of
// six segments), so for n <= 80 the traits of the rank-R class are grouped by segment (LrSeg; k_lr_classify lays a region's
// columns out class by class, segment by segment, tile aligned), every segment has its own Q, T and rank, and a tile of the scan
// kernel runs ceil(R_s / 4) = 3 K steps per accumulator where the single basis needs 6.
#include "blmm_internal.h"
#include <vector>
#include <algorithm>
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

constexpr int WB_NS = 256;  // sampled deltas (one per thread) when the sample columns live in global memory
constexpr int WB_QCAP = 48; // at most this many basis vectors are mirrored in LDS for the re-orthogonalisation
                            // (fewer when n is large; further ones are read from global memory)

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// Greedy pivoted Gram-Schmidt over NS = blockDim.x / TPS sampled deltas (TPS adjacent lanes share one sample column and
// split its rows k = part, part + TPS, ...).  Q: r-major, Q[r*n + k]; rk[0] = R, rk[1] = KR.
// WLDS: the n x NS sample columns (k-major) live in LDS (n*NS*8 bytes), else in the global workspace Wg.
template <bool WLDS, int TPS>
__global__ void __launch_bounds__(WB_NS * TPS) k_wbasis(const double* __restrict__ lam, int n, double* __restrict__ Wg,
                                                        double* __restrict__ Q, int* __restrict__ rk, int64_t* stat,
                                                        int qcap) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int NT = blockDim.x, NS = NT / TPS;
  double* sq = sh;              // n : the pivot column / new basis vector
  double* sd = sh + n;          // n : re-orthogonalisation coefficients
  double* Ql = sh + 2 * n;      // qcap x n : LDS mirror of the first basis vectors
  double* Wl = Ql + qcap * n;   // n x NS (WLDS only)
  __shared__ double s_red[WB_NS];
  __shared__ int s_arg[WB_NS];
  __shared__ int s_neg;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwave = NT >> 6;
  const int s = t / TPS, part = t % TPS;
  // a kinship with negative eigenvalues leaves the smooth family (poles at delta = -1/lambda): use the identity basis,
  // for which the low-rank form is the full-rank form
  if (t == 0) s_neg = 0;
  __syncthreads();
  for (int k = t; k < n; k += NT) if (lam[k] < -1e-12) s_neg = 1;
  __syncthreads();
  if (s_neg) {
    for (int e = t; e < n * n; e += NT) Q[e] = ((e / n) == (e % n)) ? 1.0 : 0.0;
    if (t == 0) { rk[0] = n; rk[1] = (n + 3) / 4; stat[8] = n; }
    return;
  }
  // leading dimension of the sample columns: NS + 16 doubles.  In LDS the TPS row groups of a half-wave then fall into
  // disjoint banks; in global memory it avoids the 2 KB row stride that maps every row to the same L2 channel / L1 set
  // (measured: 140 us instead of ~15 us per basis vector at n = 500)
  const int ldw = NS + 16;
#define WK(k) (WLDS ? Wl[(k) * ldw + s] : Wg[(size_t)(k) * ldw + s])
#define TPS_SUM(x)                                                    \
  do {                                                                \
    _Pragma("unroll") for (int o__ = 1; o__ < TPS; o__ <<= 1) x += __shfl_xor(x, o__, 64); \
  } while (0)
  // delta_0 = 0 (w = 1), then log-spaced over [1e-6, 1e9]: h2 from 1e-6 to 1 - 1e-9
  const double delta = (s == 0) ? 0.0 : exp(2.302585092994046 * (-6.0 + 15.0 * (double)(s - 1) / (double)(NS - 2)));
  double nrm = 0.0;
  for (int k = part; k < n; k += TPS) { const double w = 1.0 / fma(delta, fabs(lam[k]), 1.0); WK(k) = w; nrm = fma(w, w, nrm); }
  TPS_SUM(nrm);
  const double inv = 1.0 / sqrt(nrm);
  for (int k = part; k < n; k += TPS) WK(k) *= inv;
  double res2 = 1.0;
  // stop at a ~4e-15 relative residual (squared; scaled with n: the rounding floor of the deflated columns grows with it)
  const double tol2 = 2e-31 * (double)n;
  int R = 0;
  for (; R < n; ++R) {
    // arg-max of the residual norms (first maximum wins)
    if (part == 0) { s_red[s] = res2; s_arg[s] = s; }
    __syncthreads();
    for (int cnt = NS; cnt > 1;) {             // NS may be 128, 192 or 256: halve with round-up
      const int o = (cnt + 1) >> 1;
      if (t + o < cnt && (s_red[t + o] > s_red[t] || (s_red[t + o] == s_red[t] && s_arg[t + o] < s_arg[t]))) { s_red[t] = s_red[t + o]; s_arg[t] = s_arg[t + o]; }
      cnt = o;
      __syncthreads();
    }
    const double mx = s_red[0];
    const int piv = s_arg[0];
    __syncthreads();
    if (!(mx > tol2)) break;
    for (int k = t; k < n; k += NT) sq[k] = WLDS ? Wl[k * ldw + piv] : Wg[(size_t)k * ldw + piv];
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {   // classical Gram-Schmidt twice: Q stays orthonormal to rounding even for
                                              // the last, noise-dominated pivots
      for (int r = wave; r < R; r += nwave) {  // one wave per existing basis vector (LDS copy; the rest: global)
        // two loops, not one pointer chosen at run time: a pointer that may be LDS or global loses its address space and
        // every access through it becomes a flat load
        double d = 0.0;
        if (r < qcap) { for (int k = lane; k < n; k += 64) d = fma(Ql[r * n + k], sq[k], d); }
        else { for (int k = lane; k < n; k += 64) d = fma(Q[(size_t)r * n + k], sq[k], d); }
        d = wave_sum(d);
        if (lane == 0) sd[r] = d;
      }
      __syncthreads();
      for (int k = t; k < n; k += NT) {
        double v = sq[k];
        const int rl = R < qcap ? R : qcap;
        for (int r = 0; r < rl; ++r) v = fma(-sd[r], Ql[r * n + k], v);
        for (int r = rl; r < R; ++r) v = fma(-sd[r], Q[(size_t)r * n + k], v);
        sq[k] = v;
      }
      __syncthreads();
    }
    double pn = 0.0;
    for (int k = t; k < n; k += NT) pn = fma(sq[k], sq[k], pn);
    pn = wave_sum(pn);
    if (lane == 0) s_red[wave] = pn;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < nwave; ++w) tot += s_red[w];
    const double qn = 1.0 / sqrt(tot);
    __syncthreads();
    // a pivot that loses > 99.99 % of its norm in the re-orthogonalisation was rounding noise: the family is exhausted
    if (!(tot > 1e-8 * mx)) break;
    for (int k = t; k < n; k += NT) { const double v = sq[k] * qn; sq[k] = v; Q[(size_t)R * n + k] = v; if (R < qcap) Ql[R * n + k] = v; }
    __syncthreads();
    // deflate every sample column (loads batched 8 deep: the global-memory variant is latency bound otherwise, and the
    // compiler may not move a load of W above the preceding store to W)
    double c = 0.0;
    {
      int k = part;
      for (; k + 7 * TPS < n; k += 8 * TPS) {
        double wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = WK(k + u * TPS);
#pragma unroll
        for (int u = 0; u < 8; ++u) c = fma(sq[k + u * TPS], wv[u], c);
      }
      for (; k < n; k += TPS) c = fma(sq[k], WK(k), c);
    }
    TPS_SUM(c);
    double r2 = 0.0;
    {
      int k = part;
      for (; k + 7 * TPS < n; k += 8 * TPS) {
        double wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = WK(k + u * TPS);
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double v = fma(-c, sq[k + u * TPS], wv[u]); WK(k + u * TPS) = v; r2 = fma(v, v, r2); }
      }
      for (; k < n; k += TPS) { const double v = fma(-c, sq[k], WK(k)); WK(k) = v; r2 = fma(v, v, r2); }
    }
    TPS_SUM(r2);
    res2 = r2;
    __syncthreads();
  }
#undef WK
#undef TPS_SUM
  if (t == 0) { rk[0] = R; rk[1] = (R + 3) / 4; stat[8] = R; }
}

// Register-resident variant for n <= 4 * NKR: the 256 sample columns live in the registers of their 4 lanes (NKR values
// each), so the deflation of every column -- the bulk of an iteration -- touches LDS only for the new basis vector.  Same
// greedy sequence as k_wbasis (same samples when that one runs with 256 of them, same pivot rule).
// blockIdx.x = segment of the heritability axis (LrSeg): its own samples, its own basis Q + blockIdx.x * qstride, rk + 4 * blockIdx.x.
template <int NKR>
__global__ void __launch_bounds__(1024) k_wbasis_reg(const double* __restrict__ lam, int n, double* __restrict__ Q,
                                                     int* __restrict__ rk, int64_t* stat, int qcap, LrSeg seg, int64_t qstride) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  constexpr int TPS = 4, NS = 256, NT = 1024;
  const int sg = blockIdx.x;
  Q += (int64_t)sg * qstride; rk += 4 * sg;
  double* sq = sh;              // n : the pivot column / new basis vector
  double* sd = sh + n;          // n : re-orthogonalisation coefficients
  double* Ql = sh + 2 * n;      // qcap x n : LDS mirror of the first basis vectors
  __shared__ double s_red[NS];
  __shared__ int s_arg[NS];
  __shared__ int s_neg;
  __shared__ double s_piv_val;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwave = NT >> 6;
  const int s = t / TPS, part = t % TPS;
  if (t == 0) s_neg = 0;
  __syncthreads();
  for (int k = t; k < n; k += NT) if (lam[k] < -1e-12) s_neg = 1;
  __syncthreads();
  if (s_neg) {
    for (int e = t; e < n * n; e += NT) Q[e] = ((e / n) == (e % n)) ? 1.0 : 0.0;
    if (t == 0) { rk[0] = n; rk[1] = (n + 3) / 4; stat[8] = n; }
    return;
  }
  // one segment: delta_0 = 0 (w = 1), then log-spaced over [1e-6, 1e9] (h2 from 1e-6 to 1 - 1e-9).  Several: sample 0 is the
  // segment's lower edge, the others log-spaced from there (1e-6 in the first segment) to its upper edge (1e9 in the last)
  double delta;
  if (seg.S <= 1) delta = (s == 0) ? 0.0 : exp(2.302585092994046 * (-6.0 + 15.0 * (double)(s - 1) / (double)(NS - 2)));
  else {
    const double ha = seg.edge[sg], hb = seg.edge[sg + 1];
    const double da = ha / (1.0 - ha), db = (sg + 1 < seg.S && hb < 1.0) ? hb / (1.0 - hb) : 1e9;
    const double lo = log(fmax(da, 1e-6)), hi = log(fmin(fmax(db, 2e-6), 1e9));
    delta = (s == 0) ? da : exp(lo + (hi - lo) * (double)(s - 1) / (double)(NS - 2));
  }
  double wv[NKR];
  double nrm = 0.0;
#pragma unroll
  for (int i = 0; i < NKR; ++i) {
    const int k = part + TPS * i;
    wv[i] = (k < n) ? 1.0 / fma(delta, fabs(lam[k]), 1.0) : 0.0;
    nrm = fma(wv[i], wv[i], nrm);
  }
  nrm += __shfl_xor(nrm, 1, 64); nrm += __shfl_xor(nrm, 2, 64);
  const double inv = 1.0 / sqrt(nrm);
#pragma unroll
  for (int i = 0; i < NKR; ++i) wv[i] *= inv;
  double res2 = 1.0;
  const double tol2 = 2e-31 * (double)n;
  int R = 0;
  for (; R < n; ++R) {
    if (part == 0) s_red[s] = res2;
    __syncthreads();
    // arg-max of the residual norms, first maximum wins: one wave, lane l owns the samples 4l .. 4l+3, so the first maximum
    // is the first lane holding the wave's maximum (a tree over LDS cost eight workgroup barriers per basis vector)
    if (wave == 0) {
      double bv = s_red[4 * lane];
      int bi = 4 * lane;
#pragma unroll
      for (int u = 1; u < 4; ++u) { const double v = s_red[4 * lane + u]; if (v > bv) { bv = v; bi = 4 * lane + u; } }
      double wm = bv;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wm = fmax(wm, __shfl_xor(wm, o, 64));
      const unsigned long long who = __ballot(bv == wm);
      const int first = who ? __ffsll((long long)who) - 1 : 0;      // who == 0: NaN residuals, the loop ends below
      const int pi = __shfl(bi, first, 64);
      if (lane == 0) { s_arg[0] = pi; s_piv_val = wm; }
    }
    __syncthreads();
    const double mx = s_piv_val;
    const int piv = s_arg[0];
    if (!(mx > tol2)) break;
    if (s == piv) {
#pragma unroll
      for (int i = 0; i < NKR; ++i) { const int k = part + TPS * i; if (k < n) sq[k] = wv[i]; }
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {     // classical Gram-Schmidt twice (as k_wbasis)
      for (int r = wave; r < R; r += nwave) {
        // two loops, not one pointer chosen at run time: a pointer that may be LDS or global loses its address space and
        // every access through it becomes a flat load
        double d = 0.0;
        if (r < qcap) { for (int k = lane; k < n; k += 64) d = fma(Ql[r * n + k], sq[k], d); }
        else { for (int k = lane; k < n; k += 64) d = fma(Q[(size_t)r * n + k], sq[k], d); }
        d = wave_sum(d);
        if (lane == 0) sd[r] = d;
      }
      __syncthreads();
      for (int k = t; k < n; k += NT) {
        double v = sq[k];
        const int rl = R < qcap ? R : qcap;
        for (int r = 0; r < rl; ++r) v = fma(-sd[r], Ql[r * n + k], v);
        for (int r = rl; r < R; ++r) v = fma(-sd[r], Q[(size_t)r * n + k], v);
        sq[k] = v;
      }
      __syncthreads();
    }
    double pn = 0.0;
    for (int k = t; k < n; k += NT) pn = fma(sq[k], sq[k], pn);
    pn = wave_sum(pn);
    if (lane == 0) s_red[wave] = pn;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < nwave; ++w) tot += s_red[w];
    const double qn = 1.0 / sqrt(tot);
    __syncthreads();
    if (!(tot > 1e-8 * mx)) break;             // noise pivot: the family is exhausted
    for (int k = t; k < n; k += NT) { const double v = sq[k] * qn; sq[k] = v; Q[(size_t)R * n + k] = v; if (R < qcap) Ql[R * n + k] = v; }
    __syncthreads();
    // deflate this lane group's sample column (registers) against the new basis vector (LDS, 4 addresses per wave)
    double qv[NKR];
    double c = 0.0;
#pragma unroll
    for (int i = 0; i < NKR; ++i) {
      const int k = part + TPS * i;
      qv[i] = (k < n) ? sq[k] : 0.0;
      c = fma(qv[i], wv[i], c);
    }
    c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64);
    double r2 = 0.0;
#pragma unroll
    for (int i = 0; i < NKR; ++i) { wv[i] = fma(-c, qv[i], wv[i]); r2 = fma(wv[i], wv[i], r2); }
    r2 += __shfl_xor(r2, 1, 64); r2 += __shfl_xor(r2, 2, 64);
    res2 = r2;
    __syncthreads();
  }
  if (t == 0) { rk[0] = R; rk[1] = (R + 3) / 4; atomicMax((unsigned long long*)&stat[8], (unsigned long long)R); }   // stat[8]: the largest rank over the segments
}

// Multi-workgroup variant for n beyond the single-workgroup LDS budget.  The 256 sample columns are dealt S per
// workgroup and stay in LDS (one wave per column: rows across the lanes); every workgroup runs the SAME greedy
// iteration and builds the SAME Q (bitwise: identical inputs, identical instruction sequence), so the only exchange per
// basis vector is "which column is the pivot": each workgroup publishes its best residual and that column, ONE grid
// barrier (monotonic counter, agent-scope release/acquire), then all read the G candidates and take the same winner.
// G <= 64 workgroups of one CU each: co-resident on any MI355X partition that runs the scan at all; a workgroup that
// waits longer than 2^22 polls (seconds) gives up (rk = {-1, 0}: the scan writes NaN and blmm_status reports the failure)
// rather than hang the device.
struct WbExch {                 // global exchange area, double-buffered by iteration parity
  double* val;                  // [2][G]   best residual norm^2 of the workgroup
  double* col;                  // [2][G][n] the corresponding (deflated) sample column
  unsigned int* cnt;            // arrival counter (zeroed by the launcher)
};

__global__ void __launch_bounds__(1024) k_wbasis_mw(const double* __restrict__ lam, int n, int S, int qcap,
                                                    double* __restrict__ Q, int* __restrict__ rk, int64_t* stat,
                                                    WbExch x) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int G = gridDim.x, g = blockIdx.x, NS = G * S;
  double* sq = sh;                       // n
  double* sd = sh + n;                   // n
  double* Ql = sh + 2 * n;               // qcap x n
  double* Wl = Ql + (size_t)qcap * n;    // S x n, sample-major
  __shared__ double s_res[16], s_red[16], s_val[64];
  __shared__ int s_i0, s_i1;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, NT = blockDim.x, nwave = NT >> 6;
  // negative eigenvalues: identity basis (see k_wbasis); every workgroup takes the same branch, no barrier involved
  if (t == 0) s_i0 = 0;
  __syncthreads();
  for (int k = t; k < n; k += NT) if (lam[k] < -1e-12) s_i0 = 1;
  __syncthreads();
  if (s_i0) {
    for (size_t e = (size_t)g * NT + t; e < (size_t)n * n; e += (size_t)G * NT) Q[e] = ((e / n) == (e % n)) ? 1.0 : 0.0;
    if (g == 0 && t == 0) { rk[0] = n; rk[1] = (n + 3) / 4; stat[8] = n; }
    return;
  }
  // sample columns of this workgroup: global sample index s = g * S + ss (the log-spaced deltas of k_wbasis)
  for (int ss = wave; ss < S; ss += nwave) {
    const int s = g * S + ss;
    const double delta = (s == 0) ? 0.0 : exp(2.302585092994046 * (-6.0 + 15.0 * (double)(s - 1) / (double)(NS - 2)));
    double nrm = 0.0;
    for (int k = lane; k < n; k += 64) { const double w = 1.0 / fma(delta, fabs(lam[k]), 1.0); Wl[(size_t)ss * n + k] = w; nrm = fma(w, w, nrm); }
    nrm = wave_sum(nrm);
    const double inv = 1.0 / sqrt(nrm);
    for (int k = lane; k < n; k += 64) Wl[(size_t)ss * n + k] *= inv;
    if (lane == 0) s_res[ss] = 1.0;
  }
  const double tol2 = 2e-31 * (double)n;
  int R = 0;
  bool aborted = false;
  for (; R < n; ++R) {
    const int par = R & 1;
    __syncthreads();
    // publish this workgroup's best column
    if (t == 0) {
      int b = 0;
      for (int ss = 1; ss < S; ++ss) if (s_res[ss] > s_res[b]) b = ss;
      s_i0 = b;
      x.val[par * G + g] = s_res[b];
    }
    __syncthreads();
    {
      const double* wb = Wl + (size_t)s_i0 * n;
      double* dst = x.col + ((size_t)par * G + g) * n;
      for (int k = t; k < n; k += NT) dst[k] = wb[k];
    }
    __threadfence();
    __syncthreads();
    if (t == 0) {
      __hip_atomic_fetch_add(x.cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int target = (unsigned int)G * (unsigned int)(R + 1);
      int ok = 0;
      for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(x.cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      s_i1 = ok;
    }
    __syncthreads();
    if (!s_i1) { aborted = true; break; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // the same winner in every workgroup: largest residual, lowest workgroup (= lowest sample index) on ties
    if (t < G) s_val[t] = x.val[par * G + t];
    __syncthreads();
    if (t == 0) {
      int b = 0;
      for (int w = 1; w < G; ++w) if (s_val[w] > s_val[b]) b = w;
      s_i0 = b;
    }
    __syncthreads();
    const double mx = s_val[s_i0];
    if (!(mx > tol2)) break;
    {
      const double* src = x.col + ((size_t)par * G + s_i0) * n;
      for (int k = t; k < n; k += NT) sq[k] = src[k];
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {     // classical Gram-Schmidt twice (as k_wbasis)
      for (int r = wave; r < R; r += nwave) {
        // two loops, not one pointer chosen at run time: a pointer that may be LDS or global loses its address space and
        // every access through it becomes a flat load
        double d = 0.0;
        if (r < qcap) { for (int k = lane; k < n; k += 64) d = fma(Ql[(size_t)r * n + k], sq[k], d); }
        else { for (int k = lane; k < n; k += 64) d = fma(Q[(size_t)r * n + k], sq[k], d); }
        d = wave_sum(d);
        if (lane == 0) sd[r] = d;
      }
      __syncthreads();
      for (int k = t; k < n; k += NT) {
        double v = sq[k];
        const int rl = R < qcap ? R : qcap;
        for (int r = 0; r < rl; ++r) v = fma(-sd[r], Ql[(size_t)r * n + k], v);
        for (int r = rl; r < R; ++r) v = fma(-sd[r], Q[(size_t)r * n + k], v);
        sq[k] = v;
      }
      __syncthreads();
    }
    double pn = 0.0;
    for (int k = t; k < n; k += NT) pn = fma(sq[k], sq[k], pn);
    pn = wave_sum(pn);
    if (lane == 0) s_red[wave] = pn;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < nwave; ++w) tot += s_red[w];
    const double qn = 1.0 / sqrt(tot);
    if (!(tot > 1e-8 * mx)) break;             // noise pivot: the family is exhausted (uniform: same data everywhere)
    // every workgroup writes the same bytes of Q[R] (benign) and reads back only what its own threads wrote
    for (int k = t; k < n; k += NT) { const double v = sq[k] * qn; sq[k] = v; Q[(size_t)R * n + k] = v; if (R < qcap) Ql[(size_t)R * n + k] = v; }
    __syncthreads();
    for (int ss = wave; ss < S; ss += nwave) {   // deflate this workgroup's sample columns, one wave per column
      double* w = Wl + (size_t)ss * n;
      double c = 0.0;
      for (int k = lane; k < n; k += 64) c = fma(sq[k], w[k], c);
      c = wave_sum(c);
      double r2 = 0.0;
      for (int k = lane; k < n; k += 64) { const double v = fma(-c, sq[k], w[k]); w[k] = v; r2 = fma(v, v, r2); }
      r2 = wave_sum(r2);
      if (lane == 0) s_res[ss] = r2;
    }
  }
  if (g == 0 && t == 0) {
    if (aborted) { rk[0] = -1; rk[1] = 0; stat[8] = -1; }
    else { rk[0] = R; rk[1] = (R + 3) / 4; stat[8] = R; }
  }
}

// Segments of the heritability axis (LrSeg): only the register-resident basis kernel builds several bases (n <= 80: the BXD case);
// BLMM_LR_SEGMENTS=1: a single basis everywhere (A/B testing), =2..8: that many equal segments.  The default edges keep every
// segment's rank at 11-12 on the BXD kinship spectrum (tools: see DESIGN §4.1).
LrSeg lr_segments(const blmm_ctx* ctx, int n) {
  LrSeg sg;
  static const char* mw_env = dev_env("BLMM_WBASIS");
  static const char* wide_env = dev_env("BLMM_LR_PANELS_WIDE");
  static const int nb_env = dev_env("BLMM_LR_PANELS_BATCH") ? atoi(dev_env("BLMM_LR_PANELS_BATCH")) : 0;
  const char* se = dev_env("BLMM_LR_SEGMENTS");       // read per call: tests compare segmented and single-basis results in one process
  const int want = se ? atoi(se) : ctx->tune.lr_segments;   // tuning key "lr_segments"
  if (n > 80 || n < 8 || want == 1 || (mw_env && std::strcmp(mw_env, "lds") == 0) || (wide_env && wide_env[0] == '0') || nb_env > 1) return sg;
  if (want >= 2 && want <= LR_SEG_MAX) {
    sg.S = want;
    for (int i = 0; i <= LR_SEG_MAX; ++i) sg.edge[i] = (i < want) ? (double)i / want : 2.0;
    return sg;
  }
  static const double def[7] = {0.0, 0.25, 0.5, 0.7, 0.85, 0.95, 2.0};
  sg.S = 6;
  for (int i = 0; i <= LR_SEG_MAX; ++i) sg.edge[i] = (i < 7) ? def[i] : 2.0;
  return sg;
}

int launch_wbasis(blmm_ctx* ctx, const double* lam, int n, int npad, const LrSeg& seg, double* Wk, double* Q, int* rk, int64_t* stat) {
  // LDS budget (156 KB dynamic): two work vectors, the sample columns when 256, 192 or 128 of them fit (padded to
  // ns + 16 per row) beside at least 16 mirrored basis vectors, then as many mirrored basis vectors as fit (<= WB_QCAP)
  const size_t budget = 156 * 1024, work = sizeof(double) * (size_t)2 * n, row = sizeof(double) * (size_t)n;
  if (work + row > budget) return fail(ctx, BLMM_ERR_UNSUPPORTED, "n too large for the weight-basis kernel");
  int ns = 0;
  for (int cand : {256, 192, 128})
    if (work + row * (size_t)(cand + 16) + row * 16 <= budget) { ns = cand; break; }
  static const char* mw_env = dev_env("BLMM_WBASIS");   // "single": never take the multi-workgroup variant; "lds": never the register one (A/B testing)
  if (n <= 80 && !(mw_env && std::strcmp(mw_env, "lds") == 0)) {
    // sample columns in registers (20 per lane; 32 for n <= 128 would spill at 1024 threads)
    const int qcap = (int)std::min<size_t>(WB_QCAP, (budget - work) / row);
    const size_t lds = work + row * qcap;
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wbasis_reg<20>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_wbasis_reg<20>, dim3((unsigned)seg.S), dim3(1024), lds, ctx->stream, lam, n, Q, rk, stat, qcap, seg, (int64_t)npad * n);
    KCHECK();
    return BLMM_OK;
  }
  if (seg.S != 1) return fail(ctx, BLMM_ERR_INVALID, "launch_wbasis: several segments need the register-resident basis kernel");
  if (ns) {
    const size_t wbytes = row * (size_t)(ns + 16);
    const int qcap = (int)std::min<size_t>(WB_QCAP, (budget - work - wbytes) / row);
    const size_t lds = work + row * qcap + wbytes;
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wbasis<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_wbasis<true, 4>), dim3(1), dim3(ns * 4), lds, ctx->stream, lam, n, Wk, Q, rk, stat, qcap);
    KCHECK();
    return BLMM_OK;
  }
  // multi-workgroup variant: S = 16, 8 or 4 sample columns per workgroup (G = 16, 32, 64), at least 2 mirrored basis rows
  int S = 0;
  for (int cand : {16, 8, 4})
    if (work + row * (size_t)cand + row * 2 <= budget) { S = cand; break; }
  // every workgroup of the multi-workgroup kernel occupies one CU (1024 threads, > 80 KB of LDS) and they meet at a grid
  // barrier: G must not exceed the CUs of this device / partition (CPX: 32), or the surplus workgroups never become resident
  while (S && S < 16 && 256 / S > (ctx->num_cus > 0 ? ctx->num_cus : 256) && work + row * (size_t)(2 * S) + row * 2 <= budget) S *= 2;
  if (S && 256 / S <= (ctx->num_cus > 0 ? ctx->num_cus : 256) && !(mw_env && std::strcmp(mw_env, "single") == 0)) {
    const int G = 256 / S;
    const int qcap = (int)std::min<size_t>(WB_QCAP, (budget - work - row * S) / row);
    const size_t lds = work + row * qcap + row * S;
    // exchange area inside the sample workspace Wk (n x 272 doubles >= 2 G (n + 1) doubles + the counter)
    WbExch x;
    x.val = Wk; x.col = Wk + 2 * G; x.cnt = reinterpret_cast<unsigned int*>(Wk + 2 * G + (size_t)2 * G * n);
    BLMM_HIP(hipMemsetAsync(x.cnt, 0, sizeof(unsigned int) * 2, ctx->stream));
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wbasis_mw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GridKernelGuard gk(ctx);
    if (gk.rc) return gk.rc;
    hipLaunchKernelGGL(k_wbasis_mw, dim3(G), dim3(1024), lds, ctx->stream, lam, n, S, qcap, Q, rk, stat, x);
    KCHECK();
    return gk.record();
  }
  {
    // sample columns in global memory (L2-resident), one workgroup: slow (~0.1 ms per basis vector) but has no
    // co-residency requirement
    const int qcap = (int)std::min<size_t>(WB_QCAP, (budget - work) / row);
    const size_t lds = work + row * qcap;
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wbasis<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_wbasis<false, 4>), dim3(1), dim3(WB_NS * 4), lds, ctx->stream, lam, n, Wk, Q, rk, stat, qcap);
  }
  KCHECK();
  return BLMM_OK;
}

// T[q][r][i]: q = 0: sum_k Q[r][k] x_ik^2 ; q = 1..c: sum_k Q[r][k] x_ik z_(q-1)k.   Rows r >= R (up to 4*KR) are zero.
// grid = (ldx/256, ceil(4*KRmax/16)); the kernel reads R from rk and returns early for chunks beyond it.
// The first chunk also leaves den0_i = 1 / sqrt(Sxx - |L0^-1 s|^2) of the unweighted model (Sxx = sum_k x_ik^2, s_q = sum_k x_ik z_qk,
// Z0'Z0 = L0 L0'): the denominators of the shared-weights class, from the same pass over Xt.
template <int C, bool STAGE>
__global__ void __launch_bounds__(256) k_lr_tpanels(const double* __restrict__ Xt, int64_t ldx, int64_t p, int n,
                                                    const double* __restrict__ Z0, const double* __restrict__ Q,
                                                    const int* __restrict__ rk, double* __restrict__ T, int64_t tstride,
                                                    double* __restrict__ den0, int nchunk, int64_t qstride) {
  // The chunk's 16 basis rows and Z0 are staged in LDS once (STAGE; beyond the LDS budget, n > ~1000, they are read from
  // L2 -- a template parameter, not a run-time pointer choice: a pointer that may be either loses its address space and
  // every access becomes a flat load), and a thread fetches its marker's x eight
  // individuals at a time before using them: with the loads next to their uses every k cost an L2 round trip (94 us at the
  // BXD shape for ~9 us of arithmetic -- on the critical path of the scan).
  extern __shared__ __attribute__((aligned(16))) double sh[];
  // blockIdx.y = segment * nchunk + chunk of 16 basis rows (segment s: Q + s * qstride, rk + 4 s, T + s * (1 + C) * tstride)
  const int sgi = blockIdx.y / nchunk;
  Q += (int64_t)sgi * qstride; rk += 4 * sgi; T += (int64_t)sgi * (1 + C) * tstride;
  const int R = rk[0], R4 = rk[1] * 4;
  const int r0 = (blockIdx.y - sgi * nchunk) * 16;
  const bool first = r0 == 0 && sgi == 0;        // this chunk also leaves den0 (the unweighted model)
  if (r0 >= R4 && !first) return;                // workgroup-uniform
  if constexpr (STAGE) {
    for (int e = threadIdx.x; e < 16 * n; e += blockDim.x) { const int t = e / n; sh[e] = (r0 + t < R) ? Q[(size_t)(r0 + t) * n + (e % n)] : 0.0; }
    for (int e = threadIdx.x; e < C * n; e += blockDim.x) sh[16 * n + e] = Z0[e];
    __syncthreads();
  }
  auto qval = [&](int t, int k) -> double {       // basis row r0 + t at individual k
    if constexpr (STAGE) return sh[t * n + k];
    else return (r0 + t < R) ? Q[(size_t)(r0 + t) * n + k] : 0.0;
  };
  auto zval = [&](int q, int k) -> double {
    if constexpr (STAGE) return sh[16 * n + q * n + k];
    else return Z0[q * n + k];
  };
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ldx) return;
  constexpr int NA = C * (C + 1) / 2;
  double un[1 + C], A0[NA];                      // unweighted sums (first chunk only)
#pragma unroll
  for (int q = 0; q <= C; ++q) un[q] = 0.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) A0[a] = 0.0;
  double acc[1 + C][16];
#pragma unroll
  for (int q = 0; q <= C; ++q)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[q][t] = 0.0;
  if (i < p) {
    // the next eight individuals are requested before the current eight are used (one wave per SIMD here: every batch was a full
    // memory round trip in front of its arithmetic)
    double xn[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xn[u] = (u < n) ? Xt[(int64_t)u * ldx + i] : 0.0;
    for (int k0 = 0; k0 < n; k0 += 8) {
      double xs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xs[u] = xn[u];
#pragma unroll
      for (int u = 0; u < 8; ++u) xn[u] = (k0 + 8 + u < n) ? Xt[(int64_t)(k0 + 8 + u) * ldx + i] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u;
        if (k < n) {                             // workgroup-uniform
          const double x = xs[u];
          double xv[1 + C];
          xv[0] = x * x;
#pragma unroll
          for (int q = 0; q < C; ++q) xv[1 + q] = x * zval(q, k);
          if (first) {                           // workgroup-uniform
#pragma unroll
            for (int q = 0; q <= C; ++q) un[q] += xv[q];
#pragma unroll
            for (int q = 0; q < C; ++q)
#pragma unroll
              for (int r = 0; r <= q; ++r) A0[q * (q + 1) / 2 + r] = fma(zval(q, k), zval(r, k), A0[q * (q + 1) / 2 + r]);
          }
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const double qv = qval(t, k);          // one address per wave: a broadcast
#pragma unroll
            for (int q = 0; q <= C; ++q) acc[q][t] = fma(qv, xv[q], acc[q][t]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (r0 + t < R4)
#pragma unroll
      for (int q = 0; q <= C; ++q) T[q * tstride + (int64_t)(r0 + t) * ldx + i] = acc[q][t];
  if (first) {
    double xx = 1.0;
    if (i < p) {
      double L[NA];
      xx = un[0];
#pragma unroll
      for (int q = 0; q < C; ++q) {
#pragma unroll
        for (int r = 0; r <= q; ++r) {
          double v = A0[q * (q + 1) / 2 + r];
#pragma unroll
          for (int u = 0; u < r; ++u) v = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], v);
          L[q * (q + 1) / 2 + r] = (r == q) ? sqrt(v) : v / L[r * (r + 1) / 2 + r];
        }
        double u = un[1 + q];                    // forward substitution: u_q = (L0^-1 s)_q
#pragma unroll
        for (int r = 0; r < q; ++r) u = fma(-L[q * (q + 1) / 2 + r], un[1 + r], u);
        u /= L[q * (q + 1) / 2 + q];
        un[1 + q] = u;
        xx = fma(-u, u, xx);
      }
    }
    den0[i] = 1.0 / sqrt(xx);    // isx of the unweighted model: what the shared-weights epilogue multiplies the numerator by
  }
}

int launch_lr_tpanels(blmm_ctx* ctx, const double* Xt, int64_t ldx, int64_t p, int n, int c, int npad, const double* Z0,
                      const double* Q, const int* rk, const LrSeg& seg, double* T, int64_t tstride, double* den0) {
  const int nchunk = (npad + 15) / 16;
  const int64_t qstride = (int64_t)npad * n;
  dim3 grid((unsigned)((ldx + 255) / 256), (unsigned)(nchunk * seg.S));
  size_t lds = sizeof(double) * (size_t)n * (16 + c);
  const int stage = lds <= 150 * 1024;
  if (!stage) lds = 0;
#define TP(C) do { if (stage) { \
    if (lds > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lr_tpanels<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_lr_tpanels<C, true>), grid, dim3(256), lds, ctx->stream, Xt, ldx, p, n, Z0, Q, rk, T, tstride, den0, nchunk, qstride); \
  } else hipLaunchKernelGGL((k_lr_tpanels<C, false>), grid, dim3(256), 0, ctx->stream, Xt, ldx, p, n, Z0, Q, rk, T, tstride, den0, nchunk, qstride); } while (0)
  switch (c) {
    case 1: TP(1); break;
    case 2: TP(2); break;
    case 3: TP(3); break;
    case 4: TP(4); break;
    default: return fail(ctx, BLMM_ERR_UNSUPPORTED, "number of null covariates (incl. intercept) must be 1..4");
  }
#undef TP
  KCHECK();
  return BLMM_OK;
}

// Per trait (one thread each), from h2_j:  panel0[k][j] = w_k (y - Z0 beta_w)_k / sqrt(yy)   (as k_panels),
// Cp[r][j] = (Q' w_j)_r for r < R (zero up to 4*KR),  Ls[e][j] = packed lower-triangular L_j^-1 (A_j = Z0'W_jZ0 = L L').
template <int C>
__global__ void __launch_bounds__(256) k_lr_panels(NullModel nm, const double* __restrict__ Yt, int64_t ldy, int64_t m,
                                                   const double* __restrict__ Z0, const double* __restrict__ lam,
                                                   const double* __restrict__ h2v, const double* __restrict__ Q,
                                                   const int* __restrict__ rk, int qcap, const int* __restrict__ perm,
                                                   int64_t col0, int64_t ncol, double* __restrict__ P0,
                                                   double* __restrict__ Cp, double* __restrict__ Ls, int64_t ldp,
                                                   int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int n = nm.n, npad = nm.npad;
  double* sLam = sh;
  double* sZ = sh + n;
  double* sQ = sZ + n * C;                     // min(R, qcap) x n : basis rows (the rest is read from global memory)
  const int R = rk[0], R4 = rk[1] * 4;
  const int rl = R < qcap ? R : qcap;
  for (int e = threadIdx.x; e < n; e += blockDim.x) sLam[e] = lam[e];
  for (int e = threadIdx.x; e < n * C; e += blockDim.x) sZ[e] = Z0[e];
  for (int e = threadIdx.x; e < rl * n; e += blockDim.x) sQ[e] = Q[e];
  __syncthreads();
  // column jc of the panels belongs to trait j = perm[jc] (k_lr_classify: shared-weights traits first); -1: padding
  const int64_t jc = col0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (jc >= col0 + ncol) return;
  constexpr int NA = C * (C + 1) / 2;
  const int64_t j = perm[jc];
  if (j < 0 || j >= m) {  // padding columns: zeroed where a trait tile of k_scan_lr reaches them -- a tile of the class filled
    // from the front holds a trait in its first column, one of the class filled from the back in its last
    const int64_t tb = jc & ~(int64_t)(LR_TILE - 1);
    if (perm[tb] < 0 && perm[tb + LR_TILE - 1] < 0) return;
    for (int k = 0; k < npad; ++k) P0[(int64_t)k * ldp + jc] = 0.0;
    for (int r = 0; r < R4; ++r) Cp[(int64_t)r * ldp + jc] = 0.0;
    for (int e = 0; e < NA; ++e) Ls[(int64_t)e * ldp + jc] = 0.0;
    return;
  }
  const double h2 = h2v[j];
  const double delta = h2 / (1.0 - h2);
  double A[NA], v[C], syy = 0.0, ww = 0.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
#pragma unroll
  for (int q = 0; q < C; ++q) v[q] = 0.0;
  // y is read eight rows at a time (independent loads first): one thread walks a whole column, and a load placed next
  // to its use costs a global round trip per row
  for (int k0 = 0; k0 < n; k0 += 8) {
    double yv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) yv[u] = (k0 + u < n) ? Yt[(int64_t)(k0 + u) * ldy + j] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      if (k < n) {
        const double w = fabs(fast_rcp(fma(delta, sLam[k], 1.0)));  // sqrt.(abs.(makeweights)) squared, src/bulkscan_helpers.jl:138
        ww = fma(w, w, ww);
        const double y = yv[u];
        const double wy = w * y;
        syy = fma(wy, y, syy);
#pragma unroll
        for (int q = 0; q < C; ++q) {
          const double zq = sZ[q * n + k];
          v[q] = fma(wy, zq, v[q]);
          const double wz = w * zq;
#pragma unroll
          for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(wz, sZ[r * n + k], A[q * (q + 1) / 2 + r]);
        }
      }
    }
  }
  double L[NA], Li[NA], t[C], beta[C], tt = 0.0;
#pragma unroll
  for (int q = 0; q < C; ++q) {
#pragma unroll
    for (int r = 0; r <= q; ++r) {
      double s = A[q * (q + 1) / 2 + r];
#pragma unroll
      for (int u = 0; u < r; ++u) s = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], s);
      L[q * (q + 1) / 2 + r] = (r == q) ? sqrt(s) : s / L[r * (r + 1) / 2 + r];
    }
  }
#pragma unroll
  for (int q = 0; q < C; ++q) {
#pragma unroll
    for (int r = 0; r <= q; ++r) {
      double s = (r == q) ? 1.0 : 0.0;
#pragma unroll
      for (int u = r; u < q; ++u) s = fma(-L[q * (q + 1) / 2 + u], Li[u * (u + 1) / 2 + r], s);
      Li[q * (q + 1) / 2 + r] = s / L[q * (q + 1) / 2 + q];
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r <= q; ++r) s = fma(Li[q * (q + 1) / 2 + r], v[r], s);
    t[q] = s;
    tt = fma(s, s, tt);
  }
#pragma unroll
  for (int q = 0; q < C; ++q) {
    double s = 0.0;
#pragma unroll
    for (int u = q; u < C; ++u) s = fma(Li[u * (u + 1) / 2 + q], t[u], s);
    beta[q] = s;
  }
  const double yy = syy - tt;
  if (!(sqrt(fabs(yy)) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
  const double isy = 1.0 / sqrt(yy);
  for (int k0 = 0; k0 < npad; k0 += 8) {
    double yv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) yv[u] = (k0 + u < n) ? Yt[(int64_t)(k0 + u) * ldy + j] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      if (k < npad) {
        double p0 = 0.0;
        if (k < n) {
          const double w = fabs(fast_rcp(fma(delta, sLam[k], 1.0)));
          double res = yv[u];
#pragma unroll
          for (int q = 0; q < C; ++q) res = fma(-beta[q], sZ[q * n + k], res);
          p0 = w * res * isy;
        }
        P0[(int64_t)k * ldp + jc] = p0;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < NA; ++e) Ls[(int64_t)e * ldp + jc] = Li[e];
  // coefficients in the weight basis, 8 at a time (the weights are recomputed per chunk: rcp + 2 Newton steps).
  // Rows below `rl` come from the LDS copy of Q; the (rare) rest from global memory in a separate loop so that the
  // fast loop carries no global load at all.
  const int rl8 = (rl / 8) * 8;
  for (int rb = 0; rb < rl8; rb += 8) {
    double c8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c8[u] = 0.0;
    for (int k = 0; k < n; ++k) {
      const double w = fabs(fast_rcp(fma(delta, sLam[k], 1.0)));
#pragma unroll
      for (int u = 0; u < 8; ++u) c8[u] = fma(sQ[(rb + u) * n + k], w, c8[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) Cp[(int64_t)(rb + u) * ldp + jc] = c8[u];
  }
  for (int r = rl8; r < R4; ++r) {
    double c = 0.0;
    if (r < R) {
      const double* qr = (r < rl) ? sQ + r * n : Q + (size_t)r * n;
      for (int k = 0; k < n; ++k) c = fma(qr[k], fabs(fast_rcp(fma(delta, sLam[k], 1.0))), c);
    }
    Cp[(int64_t)r * ldp + jc] = c;
  }
}

// The same panels with LPT lanes per trait (the default): one thread walking a trait's n individuals (twice, plus R dot
// products of length n) is a long dependent chain on few waves -- 0.4 ms at n = 500 with a few thousand traits per shard,
// 86 us on the critical path at the BXD shape.
// Lane `sub` of a trait's group owns the individuals k = sub, sub + LPT, ...; sums are butterflied over the group, the
// small per-trait algebra is done redundantly by every lane.  Basis rows come from L2 (every group of a workgroup reads
// the same addresses, consecutive lanes consecutive k).
#ifndef PW_PRIO
#define PW_PRIO 3
#endif
#ifdef PW_DIAG
__device__ unsigned long long g_pw_diag[3 * 8192];   // per workgroup: start / after staging / end (100 MHz ticks)
#endif
template <int C, int LPT>
__global__ void __launch_bounds__(256) k_lr_panels_w(NullModel nm, const double* __restrict__ Yt, int64_t ldy, int64_t m,
                                                     const double* __restrict__ Z0, const double* __restrict__ lam,
                                                     const double* __restrict__ h2v, const double* __restrict__ Q,
                                                     const int* __restrict__ rk, int qcap, const int* __restrict__ perm,
                                                     int64_t col0, int64_t ncol, const int64_t* __restrict__ counts, int nbatch, int hiprio,
                                                     double* __restrict__ P0,
                                                     double* __restrict__ Cp, double* __restrict__ Ls, int64_t ldp,
                                                     int64_t* stat, const int64_t* __restrict__ segcnt, int nseg, int64_t qstride) {
#ifdef PW_DIAG
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_pw_diag[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) double sh[];
  // On the critical path, often beside k_brent2 (second half of the h2 search, which has slack): its older waves win the
  // oldest-first issue arbitration of a SIMD unless these waves carry a higher priority (28 us alone, 66 us beside it).
  if (hiprio) __builtin_amdgcn_s_setprio(PW_PRIO);   // (the launch on a side stream, region 1's, runs beside the scan and keeps priority 0)
  const int n = nm.n, npad = nm.npad;
  double* sLam = sh;
  double* sZ = sh + n;
  double* sQ = sZ + n * C;                     // min(R, qcap) x n : basis rows (the rest is read from L2)
  // the workgroup's columns (16 consecutive ones, nbatch = 1 whenever there are several segments) lie in ONE segment's run: its basis
  if (nseg > 1) {
    const int64_t jc0 = col0 + (int64_t)blockIdx.x * nbatch * (256 / LPT);
    const int sgi = lr_seg_of(col0 + ncol - 1 - jc0, segcnt, nseg);
    Q += (int64_t)sgi * qstride; rk += 4 * sgi;
  }
  const int R = rk[0], R4 = rk[1] * 4;
  const int rl = R < qcap ? R : qcap;
  for (int e = threadIdx.x; e < n; e += blockDim.x) sLam[e] = lam[e];
  for (int e = threadIdx.x; e < n * C; e += blockDim.x) sZ[e] = Z0[e];
  for (int e = threadIdx.x; e < rl * n; e += blockDim.x) sQ[e] = Q[e];
  __syncthreads();
#ifdef PW_DIAG
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_pw_diag[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int TPB = 256 / LPT;
  const int sub = threadIdx.x % LPT;
  constexpr int NA = C * (C + 1) / 2;
  // A workgroup walks `nbatch` batches of TPB columns (BLMM_LR_PANELS_BATCH; default 1: fewer, longer workgroups did not
  // help -- the kernel stays at 45-65 us for ~9 k wave-cycles of work per wave at ~5 % average wave-slot occupancy).
  auto do_column = [&](int64_t jc) {
  if (jc >= col0 + ncol) return;                 // whole lane groups leave together
  const int64_t j = perm[jc];
  if (j < 0 || j >= m) {   // padding column (see k_lr_panels)
    const int64_t tb = jc & ~(int64_t)(LR_TILE - 1);
    if (perm[tb] < 0 && perm[tb + LR_TILE - 1] < 0) return;
    for (int k = sub; k < npad; k += LPT) P0[(int64_t)k * ldp + jc] = 0.0;
    for (int r = sub; r < R4; r += LPT) Cp[(int64_t)r * ldp + jc] = 0.0;
    for (int e = sub; e < NA; e += LPT) Ls[(int64_t)e * ldp + jc] = 0.0;
    return;
  }
  auto gsum = [](double x) { return group_sum<LPT>(x); };
#ifdef PW_DIAG
  unsigned long long pw_t[6]; pw_t[0] = __builtin_amdgcn_s_memtime();
#endif
  const double h2 = h2v[j];
  const double delta = h2 / (1.0 - h2);
  double A[NA], v[C], syy = 0.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
#pragma unroll
  for (int q = 0; q < C; ++q) v[q] = 0.0;
  // the lane's y values, fetched together up front (a trait's rows are 8 bytes each in different sectors: next to their
  // uses every one of them cost a memory round trip) and kept for the second pass when they fit (n <= YK * LPT)
  constexpr int YK = 8;
  const bool yreg = n <= YK * LPT;
  double yv[YK];
#pragma unroll
  for (int i = 0; i < YK; ++i) { const int k = sub + LPT * i; yv[i] = (yreg && k < n) ? Yt[(int64_t)k * ldy + j] : 0.0; }
  double wreg[YK];                               // the lane's weights, kept for the second pass and the coefficients (yreg)
#pragma unroll
  for (int i = 0; i < YK; ++i) wreg[i] = 0.0;
  auto pass1 = [&](int k, double y, double& wkeep, bool live = true) {
    double w = fabs(fast_rcp(fma(delta, sLam[k], 1.0)));  // sqrt.(abs.(makeweights)) squared, src/bulkscan_helpers.jl:138
    if (!live) w = 0.0;                            // a select: a lane past n works on a clamped element at weight zero
    wkeep = w;
    const double wy = w * y;
    syy = fma(wy, y, syy);
#pragma unroll
    for (int q = 0; q < C; ++q) {
      const double zq = sZ[q * n + k];
      v[q] = fma(wy, zq, v[q]);
      const double wz = w * zq;
#pragma unroll
      for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(wz, sZ[r * n + k], A[q * (q + 1) / 2 + r]);
    }
  };
  if (yreg) {
#pragma unroll
    for (int i = 0; i < YK; ++i)
      if (LPT * i < n) { const int k = sub + LPT * i; pass1(k < n ? k : n - 1, yv[i], wreg[i], k < n); }   // wave-uniform test
  } else {
    double wdummy;
    for (int k = sub; k < n; k += LPT) pass1(k, Yt[(int64_t)k * ldy + j], wdummy);
  }
#ifdef PW_DIAG
  pw_t[1] = __builtin_amdgcn_s_memtime();
#endif
  syy = gsum(syy);
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = gsum(A[a]);
#pragma unroll
  for (int q = 0; q < C; ++q) v[q] = gsum(v[q]);
  double L[NA], Li[NA], t[C], beta[C], tt = 0.0;
#pragma unroll
  for (int q = 0; q < C; ++q) {
#pragma unroll
    for (int r = 0; r <= q; ++r) {
      double s = A[q * (q + 1) / 2 + r];
#pragma unroll
      for (int u = 0; u < r; ++u) s = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], s);
      L[q * (q + 1) / 2 + r] = (r == q) ? sqrt(s) : s / L[r * (r + 1) / 2 + r];
    }
  }
#pragma unroll
  for (int q = 0; q < C; ++q) {
#pragma unroll
    for (int r = 0; r <= q; ++r) {
      double s = (r == q) ? 1.0 : 0.0;
#pragma unroll
      for (int u = r; u < q; ++u) s = fma(-L[q * (q + 1) / 2 + u], Li[u * (u + 1) / 2 + r], s);
      Li[q * (q + 1) / 2 + r] = s / L[q * (q + 1) / 2 + q];
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r <= q; ++r) s = fma(Li[q * (q + 1) / 2 + r], v[r], s);
    t[q] = s;
    tt = fma(s, s, tt);
  }
#pragma unroll
  for (int q = 0; q < C; ++q) {
    double s = 0.0;
#pragma unroll
    for (int u = q; u < C; ++u) s = fma(Li[u * (u + 1) / 2 + q], t[u], s);
    beta[q] = s;
  }
#ifdef PW_DIAG
  pw_t[2] = __builtin_amdgcn_s_memtime();
#endif
  const double yy = syy - tt;
  if (sub == 0 && !(sqrt(fabs(yy)) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
  const double isy = 1.0 / sqrt(yy);
  auto pass2 = [&](int k, double y, double wk) {   // wk < 0: recompute the weight
    double p0 = 0.0;
    if (k < n) {
      const double w = (wk >= 0.0) ? wk : fabs(fast_rcp(fma(delta, sLam[k], 1.0)));
      double res = y;
#pragma unroll
      for (int q = 0; q < C; ++q) res = fma(-beta[q], sZ[q * n + k], res);
      p0 = w * res * isy;
    }
    P0[(int64_t)k * ldp + jc] = p0;
  };
  if (yreg) {
    // registers: weight 0 and y = 0 past n, so the arithmetic runs branch-free on a clamped element; only the store is masked
#pragma unroll
    for (int i = 0; i < YK; ++i)
      if (LPT * i < npad) {                        // wave-uniform
        const int k = sub + LPT * i, kc = k < n ? k : n - 1;
        double res = yv[i];
#pragma unroll
        for (int q = 0; q < C; ++q) res = fma(-beta[q], sZ[q * n + kc], res);
        const double p0 = (k < n) ? wreg[i] * res * isy : 0.0;
        if (k < npad) P0[(int64_t)k * ldp + jc] = p0;
      }
    for (int k = sub + LPT * YK; k < npad; k += LPT) pass2(k, 0.0, 0.0);    // padding rows beyond the registers (k >= n)
  } else {
    for (int k = sub; k < npad; k += LPT) pass2(k, k < n ? Yt[(int64_t)k * ldy + j] : 0.0, -1.0);
  }
#ifdef PW_DIAG
  pw_t[3] = __builtin_amdgcn_s_memtime();
#endif
  if (sub == 0) {
#pragma unroll
    for (int e = 0; e < NA; ++e) Ls[(int64_t)e * ldp + jc] = Li[e];
  }
  // A column of the shared-weights class (the front of the region) needs no coefficients: neither scan kernel reads them.
  if (jc - col0 < counts[0]) return;
  // coefficients in the weight basis, 8 rows at a time over the lane's own individuals.  With the weights in registers
  // (yreg) the loop over individuals is unrolled: its LDS reads are issued together instead of one round trip per
  // individual, and no reciprocal is recomputed (s_memtime: this phase was 12-14 k of a wave's ~19 k cycles).
  for (int rb = 0; rb < R4; rb += 8) {
    double c8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c8[u] = 0.0;
    if (yreg && rb + 8 <= rl) {                   // workgroup-uniform: weights in registers, the chunk's rows all in LDS
      // branch-free over the lanes (a lane past n reads a clamped element against its zero weight): inside a lane-conditional
      // block hipcc waits for each block's LDS reads before it issues the next block's
#pragma unroll
      for (int i = 0; i < YK; ++i) {
        if (LPT * i < n) {                        // wave-uniform
          const int k = sub + LPT * i;
          const int kc = k < n ? k : n - 1;
#pragma unroll
          for (int u = 0; u < 8; ++u) c8[u] = fma(sQ[(rb + u) * n + kc], wreg[i], c8[u]);
        }
      }
    } else {
      for (int k = sub; k < n; k += LPT) {
        const double w = fabs(fast_rcp(fma(delta, sLam[k], 1.0)));
        if (rb + 8 <= rl) {                       // workgroup-uniform: the chunk's rows are all in LDS
#pragma unroll
          for (int u = 0; u < 8; ++u) c8[u] = fma(sQ[(rb + u) * n + k], w, c8[u]);
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (rb + u < R) c8[u] = fma((rb + u < rl) ? sQ[(rb + u) * n + k] : Q[(size_t)(rb + u) * n + k], w, c8[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double c = gsum(c8[u]);
      if (rb + u < R4 && sub == ((rb + u) % LPT)) Cp[(int64_t)(rb + u) * ldp + jc] = c;
    }
  }
#ifdef PW_DIAG
  pw_t[4] = __builtin_amdgcn_s_memtime();
#ifdef PW_DIAG_PRINT
  if ((threadIdx.x & 63) == 0 && blockIdx.x % 97 == 0) printf("pw wg %d wave %d: loads+pass1 %llu  sums+chol %llu  pass2 %llu  coeffs %llu cycles\n", (int)blockIdx.x, (int)(threadIdx.x >> 6), pw_t[1] - pw_t[0], pw_t[2] - pw_t[1], pw_t[3] - pw_t[2], pw_t[4] - pw_t[3]);
#endif
  (void)pw_t;
#endif
  };
  for (int b = 0; b < nbatch; ++b) do_column(col0 + ((int64_t)blockIdx.x * nbatch + b) * TPB + threadIdx.x / LPT);
#ifdef PW_DIAG
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_pw_diag[3 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Guard of the low-rank form: relative residual |w_j - Q c_j| / |w_j| of the weight-basis expansion of EVERY trait,
// evaluated directly (one thread per trait).  The largest squared value goes to stat[9]; a trait whose squared residual
// exceeds tol2 is appended to flag_list (count in stat[10]) and its LOD column is recomputed by k_scan_fix from the full
// length-n sums, so no LOD leaves the library that rests on an unchecked expansion.
// Stage 1: block (x: 256 traits, y: a 64-row slice of the n individuals) -> partial sums of |w - Qc|^2 and |w|^2 of the
// slice, one thread per trait (coalesced reads of its coefficients Cp[r][j]; the slice of the basis rows in LDS).
// Stage 2 adds the slices in a fixed order (the flag decision must not depend on an atomic's arrival order).
constexpr int LRR_KS = 64;
constexpr int LRR_QC = 96;    // basis rows mirrored in LDS (48 KB); the (rare) rest is read from global memory
__global__ void __launch_bounds__(256) k_lr_resid(int n, int64_t m, const double* __restrict__ lam,
                                                  const double* __restrict__ h2v, const double* __restrict__ Q,
                                                  const int* __restrict__ rk, const int* __restrict__ perm, int64_t col0,
                                                  int64_t ncol, const int64_t* __restrict__ counts,
                                                  const double* __restrict__ Cp, int64_t ldp,
                                                  double* __restrict__ part /* [nslice][2][ldp] */, const int64_t* __restrict__ segcnt,
                                                  int nseg, int64_t qstride) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  // several segments: blocks of LR_TILE columns (one tile of the scan = one segment = one basis)
  if (nseg > 1) {
    const int sgi = lr_seg_of(ncol - 1 - (int64_t)blockIdx.x * blockDim.x, segcnt, nseg);
    Q += (int64_t)sgi * qstride; rk += 4 * sgi;
  }
  const int R = rk[0];
  if (R < 0) return;                       // the basis kernel gave up: the call fails as a whole (stat[8] < 0)
  const int k0 = blockIdx.y * LRR_KS, kc = (n - k0 < LRR_KS) ? (n - k0) : LRR_KS;
  double* sLam = sh;                       // LRR_KS
  double* sQ = sh + LRR_KS;                // R x LRR_KS
  for (int e = threadIdx.x; e < kc; e += blockDim.x) sLam[e] = lam[k0 + e];
  const int rl = R < LRR_QC ? R : LRR_QC;
  for (int e = threadIdx.x; e < rl * LRR_KS; e += blockDim.x) {
    const int r = e / LRR_KS, u = e % LRR_KS;
    sQ[e] = (u < kc) ? Q[(size_t)r * n + k0 + u] : 0.0;
  }
  __syncthreads();
  const int64_t j = col0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // panel column; trait perm[j]
  if (j >= col0 + ncol) return;
  if (j - col0 < counts[0]) return;        // shared-weights class: no expansion to check (its criterion IS the bound)
  const int64_t jt = perm[j];
  if (jt < 0 || jt >= m) return;
  const double h2 = h2v[jt];
  const double delta = h2 / (1.0 - h2);
  double rr = 0.0, ww = 0.0;
  constexpr int KC = 16;
  for (int u0 = 0; u0 < kc; u0 += KC) {
    double v[KC];
#pragma unroll
    for (int u = 0; u < KC; ++u) {
      const double w = (u0 + u < kc) ? fabs(1.0 / fma(delta, sLam[u0 + u], 1.0)) : 0.0;
      v[u] = w; ww = fma(w, w, ww);
    }
    for (int r = 0; r < rl; ++r) {
      const double c = Cp[(int64_t)r * ldp + j];
      const double* qr = sQ + r * LRR_KS + u0;     // zero padded beyond kc
#pragma unroll
      for (int u = 0; u < KC; ++u) v[u] = fma(-qr[u], c, v[u]);
    }
    for (int r = rl; r < R; ++r) {
      const double c = Cp[(int64_t)r * ldp + j];
      const double* qr = Q + (size_t)r * n + k0 + u0;
#pragma unroll
      for (int u = 0; u < KC; ++u) if (u0 + u < kc) v[u] = fma(-qr[u], c, v[u]);
    }
#pragma unroll
    for (int u = 0; u < KC; ++u) rr = fma(v[u], v[u], rr);
  }
  part[((size_t)blockIdx.y * 2 + 0) * ldp + j] = rr;
  part[((size_t)blockIdx.y * 2 + 1) * ldp + j] = ww;
}

__global__ void __launch_bounds__(256) k_lr_resid2(int nslice, int64_t m, double tol2, const double* __restrict__ part,
                                                   int64_t ldp, const int* __restrict__ rk, const int* __restrict__ perm,
                                                   int64_t col0, int64_t ncol, const int64_t* __restrict__ counts,
                                                   int* __restrict__ flag_list, int64_t* stat) {
  if (rk[0] < 0) return;
  const int64_t j = col0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // panel column
  double rel2 = 0.0;
  if (j < col0 + ncol && j - col0 >= counts[0] && perm[j] >= 0 && perm[j] < m) {
    double rr = 0.0, ww = 0.0;
    for (int s = 0; s < nslice; ++s) { rr += part[((size_t)s * 2 + 0) * ldp + j]; ww += part[((size_t)s * 2 + 1) * ldp + j]; }
    rel2 = rr / ww;
    if (!(rel2 >= 0.0)) rel2 = INFINITY;   // NaN counts as a failure of the expansion
    if (!(rel2 <= tol2)) {
      const unsigned long long slot = atomicAdd((unsigned long long*)&stat[10], 1ull);
      flag_list[slot] = (int)j;            // panel column; order of the list is immaterial: k_scan_fix recomputes whole columns
    }
  }
  // largest squared residual of the block -> stat[9] (bit pattern of a non-negative double orders like an integer)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rel2 = fmax(rel2, __shfl_xor(rel2, o, 64));
  if ((threadIdx.x & 63) == 0 && rel2 > 0.0) atomicMax((unsigned long long*)&stat[9], (unsigned long long)__double_as_longlong(rel2));
}

// Full-rank recomputation of the LOD columns of the flagged traits (univar_liteqtl, src/bulkscan_helpers.jl:138-146):
//   num = x_i' a0_j ,  Sxx = sum_k x_ik^2 w_jk ,  s_q = sum_k x_ik z_qk w_jk ,  u = L_j^-1 s ,
//   r^2 = num^2 / (Sxx - |u|^2) ,  LOD = -(n/2) log10(1 - r^2)
// with plain fp64 VALU arithmetic and libm log10.  A fixed grid walks (flagged trait, 256-marker tile) pairs; it reads
// the count on the device (no host round trip) and exits at once when nothing was flagged (the normal case).
template <int C>
__global__ void __launch_bounds__(256) k_scan_fix(NullModel nm, const double* __restrict__ Xt, int64_t ldx, int64_t p,
                                                   const double* __restrict__ P0, const double* __restrict__ Ls, int64_t ldp,
                                                   const double* __restrict__ Z0, const double* __restrict__ lam,
                                                   const double* __restrict__ h2v, const int* __restrict__ flag_list,
                                                   const int* __restrict__ perm, double* __restrict__ L, int64_t ldL,
                                                   int64_t* stat, double* __restrict__ Pv, int64_t ldPv,
                                                   const double* __restrict__ pvtab) {
  constexpr int KC = 256, NL = C * (C + 1) / 2;
  __shared__ double s_a0[KC], s_w[KC], s_wz[C][KC];
  const int64_t cnt = stat[10];
  if (cnt <= 0) return;
  const int n = nm.n;
  const int64_t ntile = (p + 255) / 256;
  const double scale = -0.5 * (double)n;
  int nnan = 0;
  for (int64_t item = blockIdx.x; item < cnt * ntile; item += gridDim.x) {
    const int64_t j = flag_list[item / ntile];      // panel column
    const int64_t jt = perm[j];                     // its trait
    const int64_t i = (item % ntile) * 256 + threadIdx.x;
    const double h2 = h2v[jt];
    const double delta = h2 / (1.0 - h2);
    double num = 0.0, sxx = 0.0, sq[C];
#pragma unroll
    for (int q = 0; q < C; ++q) sq[q] = 0.0;
    for (int k0 = 0; k0 < n; k0 += KC) {
      __syncthreads();
      const int k = k0 + threadIdx.x;
      if (k < n) {
        const double w = fabs(1.0 / fma(delta, lam[k], 1.0));
        s_a0[threadIdx.x] = P0[(int64_t)k * ldp + j];
        s_w[threadIdx.x] = w;
#pragma unroll
        for (int q = 0; q < C; ++q) s_wz[q][threadIdx.x] = w * Z0[q * n + k];
      }
      __syncthreads();
      if (i < p) {
        const int kc = (n - k0 < KC) ? (n - k0) : KC;
        for (int kk = 0; kk < kc; ++kk) {
          const double x = Xt[(int64_t)(k0 + kk) * ldx + i];
          num = fma(x, s_a0[kk], num);
          sxx = fma(x * x, s_w[kk], sxx);
#pragma unroll
          for (int q = 0; q < C; ++q) sq[q] = fma(x, s_wz[q][kk], sq[q]);
        }
      }
    }
    if (i < p) {
      double li[NL];
#pragma unroll
      for (int e = 0; e < NL; ++e) li[e] = Ls[(int64_t)e * ldp + j];
      double xx = sxx;
#pragma unroll
      for (int q = 0; q < C; ++q) {
        double u = 0.0;
#pragma unroll
        for (int e = 0; e <= q; ++e) u = fma(li[q * (q + 1) / 2 + e], sq[e], u);
        xx = fma(-u, u, xx);
      }
      const double r2 = (num * num) / xx;
      const double u1 = 1.0 - r2;
      double lod = scale * log10(u1);
      if (!(u1 > 0.0)) { lod = (u1 == 0.0) ? INFINITY : NAN; nnan += (u1 != 0.0); }
      L[jt * ldL + i] = lod;
      if (Pv) Pv[jt * ldPv + i] = fast_log10p1(lod, reinterpret_cast<const dpair*>(pvtab));   // the fused `output_pvals` column
    }
  }
  if (nnan) atomicAdd((unsigned long long*)&stat[ST_NAN_LOD], (unsigned long long)nnan);
}

// ------------------------------------------------------------------------------------------------
// Shared-weights class.  A trait whose weights are all 1 to within the tolerance of the expansion guard,
//     |w_j - 1|_2 / |1|_2 <= delta_j sqrt(sum lambda^2 / n) <= tol          (w_jk = 1/(delta_j lambda_k + 1)),
// needs no weight basis: its denominators Sxx - |u|^2 are the per-marker constants den0_i of the unweighted model (the
// numerator panel still carries the trait's own weights).  On eQTL-like data that is every trait whose likelihood peaks
// at the h2 = 0 boundary -- half of the BXD-shaped bench workload -- and k_scan_lr skips the rank-R phase for their
// tiles.  k_lr_classify orders the panel columns: the class fills perm[] from the front, the other traits from the back
// (perm was preset to -1 = padding; ldq - m >= 128 keeps a whole padding tile between the two), counts in counts[0] /
// counts[1].  A workgroup owns 1024 consecutive traits and keeps their order; the order of the workgroups' ranges follows
// the arrival of two atomics -- immaterial, every LOD is written through perm and its arithmetic does not depend on the
// column it sits in.  The panel arrays hold two such regions (LrRegion): the traits k_brent finished, and the ones
// k_brent2 finishes while the first region is already being scanned.
// ------------------------------------------------------------------------------------------------
// Two passes (MODE 0: count, MODE 1: place), one trait per thread.  Class 0 = shared weights, class 1 + s = segment s of the
// heritability axis (LrSeg).  Per wave and class ONE atomic (lane c adds the wave's count of class c: nine addresses in a single
// instruction).  Pass 0 -> counts[0], segcnt[s]; pass 1 places: the shared class from the front of the region, segment s from
// the back behind the runs of the segments before it, every run rounded up to LR_TILE columns (cursors: segcnt[8 + class]);
// counts[1] = the sum of the rounded runs (what the scan's tile arithmetic needs).
template <int MODE>
__global__ void __launch_bounds__(256) k_lr_classify(int n, int64_t m, double tol, const double* __restrict__ lam,
                                                     const double* __restrict__ h2v, const int* __restrict__ fin,
                                                     const int* __restrict__ list, const unsigned int* __restrict__ list_cnt,
                                                     int* __restrict__ perm, int64_t col0, int64_t ldq,
                                                     int64_t* __restrict__ counts, int64_t* __restrict__ segcnt, LrSeg seg) {
  // The traits of this pass: list[0 .. *list_cnt) when a list is given (the traits k_brent2 finished), otherwise every
  // trait j < m with fin[j] == 1 (fin == nullptr: all of them).  Its panel columns are [col0, col0 + ldq).
  __shared__ double s_red[4];
  const int t = threadIdx.x, lane = t & 63;
  const int64_t E = list ? (int64_t)*list_cnt : m;
  // pass 0 also presets the region's column order to -1 = padding (every workgroup of the launch takes part, before any of them
  // leaves): a fill on a side stream with its event pair on the main stream cost more than these stores
  if (MODE == 0)
    for (int64_t e = (int64_t)blockIdx.x * 256 + t; e < ldq; e += (int64_t)gridDim.x * 256) perm[col0 + e] = -1;
  if (MODE == 1 && blockIdx.x == 0 && t == 0) {
    int64_t tot = 0;
    for (int s = 0; s < seg.S; ++s) tot += lr_seg_width(segcnt[s]);
    counts[1] = tot;
  }
  if ((int64_t)blockIdx.x * 256 >= E) return;               // workgroup-uniform
  double s2 = 0.0;
  for (int k = t; k < n; k += 256) s2 = fma(lam[k], lam[k], s2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
  if (lane == 0) s_red[t >> 6] = s2;
  __syncthreads();
  const double rms = sqrt((s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (double)n);
  const double thr = (tol > 0.0) ? ((rms > 0.0) ? tol / rms : INFINITY) : 0.0;
  const int64_t e = (int64_t)blockIdx.x * 256 + t;
  int cls = -1;
  int64_t jt = 0;
  if (e < E) {
    jt = list ? (int64_t)list[e] : e;
    const bool inc = list ? true : (fin ? fin[jt] == 1 : true);
    if (inc) {
      const double h2 = h2v[jt];
      if (fabs(h2 / (1.0 - h2)) <= thr) cls = 0;            // false for NaN
      else {
        int sg = 0;
        while (sg + 1 < seg.S && !(h2 < seg.edge[sg + 1])) ++sg;   // NaN: the last segment (its expansion fails the guard)
        cls = 1 + sg;
      }
    }
  }
  // the wave's count per class, and this lane's rank inside its class
  int mycount = 0, myrank = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int c = 0; c <= seg.S; ++c) {
    const unsigned long long mask = __ballot(cls == c);
    if (lane == c) mycount = __popcll(mask);
    if (cls == c) myrank = __popcll(mask & below);
  }
  long long base = 0;
  if (lane <= seg.S && mycount > 0) {
    int64_t* dst = (MODE == 0) ? (lane == 0 ? &counts[0] : &segcnt[lane - 1]) : &segcnt[8 + lane];
    base = (long long)atomicAdd((unsigned long long*)dst, (unsigned long long)mycount);
  }
  if (MODE == 1) {
    const long long b = __shfl(base, cls < 0 ? 0 : cls, 64);
    if (cls == 0) perm[col0 + b + myrank] = (int)jt;
    else if (cls > 0) {
      int64_t off = 0;
      for (int s = 0; s + 1 < cls; ++s) off += lr_seg_width(segcnt[s]);
      perm[col0 + ldq - 1 - off - (b + myrank)] = (int)jt;
    }
  }
}

// (Round 4 built the same layout in ONE launch -- 1024 entries per workgroup, class counts exchanged as data-tagged words, every
// workgroup waiting for all of them, deterministic placement without atomics -- and measured it WORSE: region 0's launch 24.7 us
// against 12 + 11 for these two passes, and region 1's, which runs beside the first region's scan, 533 us: its 1024-thread
// workgroups only become resident as scan workgroups retire, the resident ones spin meanwhile, and the scan itself slowed from
// 0.70 to 0.90 ms.  DESIGN.md "tried, not adopted".)
int launch_lr_classify(blmm_ctx* ctx, int n, int64_t m, double tol, const double* lam, const double* h2, const int* fin,
                       const int* list, const unsigned int* list_cnt, int* perm, const LrRegion& rg, const LrSeg& seg) {
  if (m > 0x7ffffff0LL) return fail(ctx, BLMM_ERR_INVALID, "too many traits for one launch");
  if (m <= 0) return BLMM_OK;
  hipLaunchKernelGGL(k_lr_classify<0>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, n, m, tol, lam, h2, fin, list,
                     list_cnt, perm, rg.col0, rg.ncol, rg.counts, rg.segcnt, seg);
  hipLaunchKernelGGL(k_lr_classify<1>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, n, m, tol, lam, h2, fin, list,
                     list_cnt, perm, rg.col0, rg.ncol, rg.counts, rg.segcnt, seg);
  KCHECK();
  return BLMM_OK;
}

int launch_lr_panels(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                     const double* lam, const double* h2, const double* Q, const int* rk, const LrSeg& seg, const int* perm, const LrRegion& rg,
                     double* P0, double* Cp, double* Ls, int64_t ldp, int64_t* stat) {
  // basis rows in LDS up to 56 KB, staged by every block: 64 threads per block give more blocks than CUs at m ~ 35k, 128
  // halve the staging per trait (BLMM_LR_PANELS_NT: A/B testing)
  // 16 lanes per trait at every n (BXD shape: prep 0.143 -> 0.122 ms, step -2.7 %; n = 500 shard: 0.42 -> 0.23 ms for the
  // kernel); BLMM_LR_PANELS_WIDE=0: one thread per trait (A/B testing)
  static const char* wide_env = dev_env("BLMM_LR_PANELS_WIDE");
  if (!(wide_env && wide_env[0] == '0')) {
    constexpr int LPT = 16;
    const int64_t wgroups = (rg.ncol + (256 / LPT) - 1) / (256 / LPT);
    static const int nb_env = dev_env("BLMM_LR_PANELS_BATCH") ? atoi(dev_env("BLMM_LR_PANELS_BATCH")) : 0;
    const int ncu = ctx->num_cus > 0 ? ctx->num_cus : 256;
    (void)ncu;
    const int nbatch = (nb_env > 0 && seg.S == 1) ? nb_env : 1;   // BXD shape, prep phase: 0.109 ms at 1, 0.106 at 2, 0.113 at 4, 0.164 at 8
    const unsigned wblocks = (unsigned)((wgroups + nbatch - 1) / nbatch);
    const int hiprio = (ctx->stream != ctx->side && ctx->stream != ctx->side2) ? 1 : 0;   // main stream = critical path
    const size_t wbase = sizeof(double) * (size_t)nm.n * (1 + nm.c);
    // rows of the basis mirrored in LDS: the rank is only known on the device (20-24 on kinship spectra); 32 rows keep the
    // workgroup at 22 KB at n = 80 (7 per CU; n rows = 52 KB held it at 3), rows beyond come from L2
    static const int qrows_env = dev_env("BLMM_LR_PANELS_QROWS") ? atoi(dev_env("BLMM_LR_PANELS_QROWS")) : 0;
    const size_t qrows = qrows_env > 0 ? (size_t)qrows_env : 32;
    const int wqcap = (wbase + sizeof(double) * nm.n <= 60 * 1024) ? (int)std::min<size_t>(std::min<size_t>((size_t)nm.n, qrows), (60 * 1024 - wbase) / (sizeof(double) * (size_t)nm.n)) : 0;
    const size_t wlds = wbase + sizeof(double) * (size_t)wqcap * nm.n;
#define LPW(C) do { if (wlds > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lr_panels_w<C, LPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds)); \
    BLMM_LAUNCH_STOP(ctx, (k_lr_panels_w<C, LPT>), dim3(wblocks), dim3(256), wlds, nm, Yt, ldy, m, Z0, lam, h2, Q, rk, wqcap, perm, rg.col0, rg.ncol, rg.counts, nbatch, hiprio, P0, Cp, Ls, ldp, stat, rg.segcnt, seg.S, (int64_t)nm.npad * nm.n); } while (0)
    switch (nm.c) {
      case 1: LPW(1); break;
      case 2: LPW(2); break;
      case 3: LPW(3); break;
      case 4: LPW(4); break;
      default: return fail(ctx, BLMM_ERR_UNSUPPORTED, "number of null covariates (incl. intercept) must be 1..4");
    }
#undef LPW
#ifdef PW_DIAG
    {
      (void)hipStreamSynchronize(ctx->stream);
      static unsigned long long z[3 * 8192];
      (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_pw_diag), sizeof(z));
      const unsigned nb = wblocks < 8192 ? wblocks : 8192;
      unsigned long long t0 = ~0ull, t1 = 0;
      std::vector<double> st, dur, stage;
      for (unsigned b = 0; b < nb; ++b) { if (z[3 * b] < t0) t0 = z[3 * b]; if (z[3 * b + 2] > t1) t1 = z[3 * b + 2]; }
      for (unsigned b = 0; b < nb; ++b) { st.push_back((z[3 * b] - t0) * 0.01); stage.push_back((z[3 * b + 1] - z[3 * b]) * 0.01); dur.push_back((z[3 * b + 2] - z[3 * b]) * 0.01); }
      std::sort(st.begin(), st.end()); std::sort(dur.begin(), dur.end()); std::sort(stage.begin(), stage.end());
      fprintf(stderr, "panels_w diag: %u wgs, span %.1f us | start p10 %.1f p50 %.1f p90 %.1f max %.1f | staging p50 %.1f p90 %.1f | duration p10 %.1f p50 %.1f p90 %.1f max %.1f us\n",
              nb, (t1 - t0) * 0.01, st[nb / 10], st[nb / 2], st[nb * 9 / 10], st[nb - 1], stage[nb / 2], stage[nb * 9 / 10], dur[nb / 10], dur[nb / 2], dur[nb * 9 / 10], dur[nb - 1]);
    }
#endif
    KCHECK();
    return BLMM_OK;
  }
  if (seg.S != 1) return fail(ctx, BLMM_ERR_INVALID, "launch_lr_panels: the one-thread-per-trait kernel takes a single weight basis");
  static const int nt_env = dev_env("BLMM_LR_PANELS_NT") ? atoi(dev_env("BLMM_LR_PANELS_NT")) : 0;
  const int nthr = (nt_env == 64 || nt_env == 128 || nt_env == 256) ? nt_env : 64;
  const unsigned blocks = (unsigned)((rg.ncol + nthr - 1) / nthr);
  const int qcap = (int)std::min<size_t>((size_t)nm.n, (56 * 1024) / (sizeof(double) * (size_t)nm.n));
  const size_t lds = sizeof(double) * ((size_t)nm.n * (1 + nm.c) + (size_t)qcap * nm.n);
#define LP(C) hipLaunchKernelGGL(k_lr_panels<C>, dim3(blocks), dim3(nthr), lds, ctx->stream, nm, Yt, ldy, m, Z0, lam, h2, Q, rk, qcap, perm, rg.col0, rg.ncol, P0, Cp, Ls, ldp, stat)
  switch (nm.c) {
    case 1: LP(1); break;
    case 2: LP(2); break;
    case 3: LP(3); break;
    case 4: LP(4); break;
    default: return fail(ctx, BLMM_ERR_UNSUPPORTED, "number of null covariates (incl. intercept) must be 1..4");
  }
#undef LP
  KCHECK();
  return BLMM_OK;
}

// The guard (all traits): the caller runs it on the side stream beside the scan kernel, then launch_scan_fix.
int launch_lr_resid(blmm_ctx* ctx, const NullModel& nm, int64_t m, double tol, const double* lam, const double* h2,
                    const double* Q, const int* rk, const LrSeg& seg, const int* perm, const LrRegion& rg, const double* Cp, int64_t ldp,
                    int* flag_list, double* part, int64_t* stat) {
  if (m <= 0) return BLMM_OK;
  const int nslice = (nm.n + LRR_KS - 1) / LRR_KS;
  const size_t lds = sizeof(double) * ((size_t)LRR_KS * (1 + (size_t)LRR_QC));
  if (lds > 48 * 1024)
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lr_resid), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int rthreads = seg.S > 1 ? LR_TILE : 256;
  hipLaunchKernelGGL(k_lr_resid, dim3((unsigned)((rg.ncol + rthreads - 1) / rthreads), (unsigned)nslice), dim3(rthreads), lds, ctx->stream, nm.n, m, lam, h2, Q,
                     rk, perm, rg.col0, rg.ncol, rg.counts, Cp, ldp, part, rg.segcnt, seg.S, (int64_t)nm.npad * nm.n);
  hipLaunchKernelGGL(k_lr_resid2, dim3((unsigned)((rg.ncol + 255) / 256)), dim3(256), 0, ctx->stream, nslice, m, tol * tol, part, ldp, rk,
                     perm, rg.col0, rg.ncol, rg.counts, flag_list, stat);
  KCHECK();
  return BLMM_OK;
}

int launch_scan_fix(blmm_ctx* ctx, const NullModel& nm, const double* Xt, int64_t ldx, int64_t p, const double* P0,
                    const double* Ls, int64_t ldp, const double* Z0, const double* lam, const double* h2,
                    const int* flag_list, const int* perm, double* L, int64_t ldL, int64_t* stat) {
  if (p <= 0) return BLMM_OK;
  const unsigned grid = (unsigned)(8 * (ctx->num_cus > 0 ? ctx->num_cus : 256));
#define FX(C) hipLaunchKernelGGL(k_scan_fix<C>, dim3(grid), dim3(256), 0, ctx->stream, nm, Xt, ldx, p, P0, Ls, ldp, Z0, lam, h2, flag_list, perm, L, ldL, stat, ctx->pv_cur, ctx->pv_cur_ld, ptr<double>(ctx->pvtab))
  switch (nm.c) {
    case 1: FX(1); break;
    case 2: FX(2); break;
    case 3: FX(3); break;
    default: return fail(ctx, BLMM_ERR_UNSUPPORTED, "number of null covariates (incl. intercept) must be 1..3 in the low-rank form");
  }
#undef FX
  KCHECK();
  return BLMM_OK;
}

}  // namespace blmm
