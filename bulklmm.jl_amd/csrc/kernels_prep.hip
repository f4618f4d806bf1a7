// kernels_prep.hip -- everything that runs once per scan before the LOD kernels:
//   design/weights, device eigensolver (one-sided Jacobi), rotation matrix, rotation GEMM (f64 MFMA),
//   per-trait null-model h2 (Optim-style Brent / grid), A-side panels, per-grid marker norms, kinship.
// gfx950 only.  Reference anchors are cited per kernel (paths relative to the BulkLMM.jl checkout).
#include "blmm_internal.h"
#include "fastmath.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace blmm {

#define KCHECK()                                                                                      \
  do {                                                                                                \
    hipError_t e__ = hipGetLastError();                                                               \
    if (e__ != hipSuccess) return fail(ctx, BLMM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e__)); \
  } while (0)

// ------------------------------------------------------------------------------------------------
// design: Zs = wd .* [1 Covar], Ks = wd wd' .* K        (src/bulkscan.jl:231-250, src/scan.jl:201-221)
// ------------------------------------------------------------------------------------------------
__global__ void k_design(const double* __restrict__ K, const double* __restrict__ Covar, int ncov, int add_intercept,
                         const double* __restrict__ wd, int n, double* __restrict__ Ks, double* __restrict__ Zs) {
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = tid; e < (int64_t)n * n; e += stride) {
    const int i = (int)(e % n), j = (int)(e / n);
    double v = K[e];
    if (wd) v *= wd[i] * wd[j];
    Ks[e] = v;
  }
  const int c = ncov + (add_intercept ? 1 : 0);
  for (int64_t e = tid; e < (int64_t)n * c; e += stride) {
    const int i = (int)(e % n), q = (int)(e / n);
    double v;
    if (add_intercept) v = (q == 0) ? 1.0 : Covar[(int64_t)(q - 1) * n + i];
    else v = Covar[(int64_t)q * n + i];
    if (wd) v *= wd[i];
    Zs[e] = v;
  }
}

int launch_design(blmm_ctx* ctx, const double* dK, const double* dCovar, int ncov, int add_intercept,
                  const double* dweights, int n, double* Ks, double* Zs) {
  int blocks = (int)std::min<int64_t>(1024, ((int64_t)n * n + 255) / 256);
  hipLaunchKernelGGL(k_design, dim3(blocks), dim3(256), 0, ctx->stream, dK, dCovar, ncov, add_intercept, dweights, n, Ks, Zs);
  KCHECK();
  return BLMM_OK;
}

// ------------------------------------------------------------------------------------------------
// Eigen-decomposition of the symmetric kinship: two-sided cyclic Jacobi with the round-robin parallel
// ordering, one workgroup, A and V resident in LDS (n <= ~100) or in global memory (larger n).
// Replaces LAPACK `eigen(K)` / `svd(K)` of src/transform_helpers.jl:21-49.  LODs do not depend on the
// eigenbasis chosen, only on K = V diag(lambda) V' holding to rounding.
// Per round (N/2 disjoint pairs): (1) one lane per pair computes (c, s) from a_pp, a_qq, a_pq;
// (2) columns p,q of A and V are rotated; (3) rows p,q of A are rotated.  No cross-lane reductions.
// The work-item -> (pair slot, matrix, row) map is the same in every round and is kept in registers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double nr_rsqrt(double x) {  // 1/sqrt(x), x > 0 normal: v_rsq_f64 + two Newton steps
  double y = __builtin_amdgcn_rsq(x);
  double h = 0.5 * x * y;
  y = fma(y, fma(-h, y, 0.5), y);
  h = 0.5 * x * y;
  return fma(y, fma(-h, y, 0.5), y);
}
__device__ __forceinline__ double nr_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

// LDS variant (n <= JAC_NMAX): every workgroup of the launch runs the SAME deterministic iteration on its own LDS
// copy of A (bitwise identical rotations, so no inter-workgroup synchronisation is ever needed) and accumulates
// the rotations only into its own slice of the rows of V (rows of V are independent under column rotations).
//
// Brent-Luk form of the round-robin ordering: the pairs of a round are always the ADJACENT positions (2s, 2s+1);
// instead of changing the index pairs, every round moves the data by the fixed permutation
//     pi: top_0 stays, top_s -> top_{s+1}, top_last -> bottom_last, bottom_s -> bottom_{s-1}, bottom_0 -> top_1
// (rows and columns of A, columns of V), ping-ponging between two LDS copies.  One thread per 2x2 block between
// pair slots (s1 < s2): B' = J1' B J2, four 8-byte reads and four 8-byte writes to round-independent addresses.
// One lane per slot computes (c, s) from its 2x2 diagonal block, no reductions.
//
// LDS layout of A ("planar blocks", upper triangle only): the four entries of the off-diagonal block (s1, s2) live in
// four planes at the SAME index tri(s1, s2) = s2 (s2 - 1)/2 + s1 -- which is also the index of the thread that owns
// the block -- and the diagonal blocks in three arrays (a_pp, a_qq, a_pq per slot).  Consecutive lanes therefore
// read consecutive 8-byte words of each plane, and since pi moves tops up and bottoms down by one slot, they also
// write consecutive words of the destination planes: reads and writes are bank-conflict free (the previous
// column-major layout, 16-byte lane stride, ran the round at ~2.7x the LDS-throughput bound, which is what limits it).
constexpr int JAC_NMAX = 124;  // NP2 = 62: 1891 off-diagonal blocks on 2 x 960 worker threads; ~140 KB of LDS
constexpr int JAC_NWG = 20;    // >= 20: at n = 79 the V items + the blocks are one item per worker thread

__device__ __forceinline__ int jac_pi(int pos, int NP2) {
  if (NP2 <= 1 || pos == 0) return pos;
  const int s = pos >> 1;
  if (pos & 1) return (s == 0) ? 2 : 2 * (s - 1) + 1;        // bottoms move down the slot index, bottom_0 -> top_1
  return (s == NP2 - 1) ? 2 * s + 1 : 2 * (s + 1);            // tops move up, the last top -> last bottom
}

// ------------------------------------------------------------------------------------------------
// post-eigen: eigenvalues, ordering, Z0 = U' Zs, and the rotation matrix
//   R = Q U' Wd,  Q = I - Z0 (Z0'Z0)^-1 Z0'  (centered = 1; the LOD statistic and the null likelihood are
//   invariant to this unweighted projection -- SURVEY.md A.4 -- it only tames cancellation), or
//   R = U' Wd (centered = 0; literal transform_rotation, src/transform_helpers.jl:34).
// Rp[i*ldr + k] = R[k, i], zero padded to npad x ldr (the A-operand layout of k_rotate).
// One workgroup.  A kernel of its own (k_post_eigen) and -- round 4, where its LDS fits beside nothing else -- the TAIL of
// k_jacobi_lds: that launch sat between k_backtransform and k_post_eigen as a 6 us no-op whenever the fast eigen path's result
// stood (every kinship of full rank), one more dependent-launch boundary on the critical path of the step; now its first workgroup
// does this work instead of returning, and when the Jacobi has to run the LAST workgroup to finish does it.
// ------------------------------------------------------------------------------------------------
struct PostEigenArgs {
  const double* Zs; const double* wd; int c, npad, ldr, decomp, centered;
  double* lam; double* U; double* Z0; double* Rp; double* tmp;
};
template <bool VLDS>
__device__ __forceinline__ void post_eigen_body(const double* __restrict__ lraw_in, const double* __restrict__ Vg, int n, const PostEigenArgs& pa,
                                                int64_t* stat, double* shv, double* Ginv /* CMAX x CMAX */, double* Gw /* CMAX x 2 CMAX */,
                                                int* s_neg) {
  const double* __restrict__ Zs_g = pa.Zs; const double* __restrict__ wd = pa.wd;
  const int c = pa.c, npad = pa.npad, ldr = pa.ldr, decomp = pa.decomp, centered = pa.centered;
  double* __restrict__ lam = pa.lam; double* __restrict__ Ug = pa.U; double* __restrict__ Z0g = pa.Z0; double* __restrict__ Rp = pa.Rp;
  double* __restrict__ tmp = pa.tmp;
  const int tid = threadIdx.x, nt = blockDim.x;
  // VLDS: V, U, Zs, Z0, Bq and the raw eigenvalues live in LDS (the loops below walk them with stride n and would
  // otherwise pay a global-memory round trip per phase); results are also written to their global buffers.
  double* Vl = shv;                    // n x n
  double* Ul = shv + (size_t)n * n;    // n x n
  double* Zsl = Ul + (size_t)n * n;    // n x c
  double* Z0l = Zsl + (size_t)n * c;   // n x c
  double* Bql = Z0l + (size_t)n * c;   // c x n
  double* lrl = Bql + (size_t)n * c;   // n
  if (VLDS) {
    for (int e = tid; e < n * n; e += nt) Vl[e] = Vg[e];
    for (int e = tid; e < n * c; e += nt) Zsl[e] = Zs_g[e];
  }
  const double* V = VLDS ? Vl : Vg;
  const double* Zs = VLDS ? Zsl : Zs_g;
  double* U = VLDS ? Ul : Ug;
  double* Z0 = VLDS ? Z0l : Z0g;
  double* lraw = VLDS ? lrl : tmp;          // n
  double* Bq = VLDS ? Bql : tmp + n;        // c x n : Ginv * (Zs' Wd)
  if (tid == 0) *s_neg = 0;
  for (int i = tid; i < n; i += nt) lraw[i] = (decomp == BLMM_SVD) ? fabs(lraw_in[i]) : lraw_in[i];
  __syncthreads();
  for (int i = tid; i < n; i += nt) {
    const double li = lraw[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const double lj = lraw[j];
      if (decomp == BLMM_SVD) rank += (lj > li) || (lj == li && j < i);
      else rank += (lj < li) || (lj == li && j < i);
    }
    lam[rank] = li;
    if (li < -1e-7) atomicAdd(s_neg, 1);
    // deterministic sign: the largest-magnitude component of every eigenvector is positive (LAPACK leaves the
    // sign unspecified; only the permutation test depends on it)
    double big = 0.0;
    for (int k = 0; k < n; ++k) { const double v = V[(size_t)i * n + k]; if (fabs(v) > fabs(big)) big = v; }
    const double sg = (big < 0.0) ? -1.0 : 1.0;
    for (int k = 0; k < n; ++k) U[(size_t)rank * n + k] = sg * V[(size_t)i * n + k];
  }
  __syncthreads();
  if (tid == 0 && *s_neg) stat[ST_NEG_EIG] += *s_neg;
  if (VLDS) { for (int e = tid; e < n * n; e += nt) Ug[e] = Ul[e]; }
  // Z0[k,q] = sum_i U[i,k] Zs[i,q]
  for (int e = tid; e < n * c; e += nt) {
    const int k = e % n, q = e / n;
    double s = 0;
    for (int i = 0; i < n; ++i) s = fma(U[(size_t)k * n + i], Zs[(size_t)q * n + i], s);
    Z0[e] = s;
    if (VLDS) Z0g[e] = s;
  }
  __syncthreads();
  if (tid == 0) {
    // Gram = Z0'Z0, inverse by Gauss-Jordan (c <= CMAX)
    auto G = [&](int r_, int c_) -> double& { return Gw[r_ * (2 * CMAX) + c_]; };   // (LDS: 16 KB at CMAX = 32 -- as a local array it would be scratch memory of every thread)
    for (int a = 0; a < c; ++a)
      for (int b = 0; b < c; ++b) {
        double s = 0;
        for (int k = 0; k < n; ++k) s = fma(Z0[(size_t)a * n + k], Z0[(size_t)b * n + k], s);
        G(a, b) = s; G(a, c + b) = (a == b) ? 1.0 : 0.0;
      }
    for (int a = 0; a < c; ++a) {
      int piv = a;
      for (int r = a + 1; r < c; ++r) if (fabs(G(r, a)) > fabs(G(piv, a))) piv = r;
      if (piv != a) for (int b = 0; b < 2 * c; ++b) { double t = G(a, b); G(a, b) = G(piv, b); G(piv, b) = t; }
      const double d = 1.0 / G(a, a);
      for (int b = 0; b < 2 * c; ++b) G(a, b) *= d;
      for (int r = 0; r < c; ++r) if (r != a) { const double f = G(r, a); for (int b = 0; b < 2 * c; ++b) G(r, b) -= f * G(a, b); }
    }
    for (int a = 0; a < c; ++a) for (int b = 0; b < c; ++b) Ginv[a * CMAX + b] = G(a, c + b);
  }
  __syncthreads();
  // Bq[q,i] = sum_r Ginv[q,r] * (Z0' U' Wd)[r,i] = sum_r Ginv[q,r] * Zs[i,r] * wd_i   (U Z0 = Zs)
  for (int e = tid; e < n * c; e += nt) {
    const int i = e % n, q = e / n;
    double s = 0;
    for (int r = 0; r < c; ++r) s = fma(Ginv[q * CMAX + r], Zs[(size_t)r * n + i], s);
    Bq[(size_t)q * n + i] = s * (wd ? wd[i] : 1.0);
  }
  __syncthreads();
  for (int e = tid; e < npad * ldr; e += nt) {
    const int k = e % ldr, i = e / ldr;
    double v = 0.0;
    if (i < n && k < n) {
      v = U[(size_t)k * n + i] * (wd ? wd[i] : 1.0);
      if (centered)
        for (int q = 0; q < c; ++q) v = fma(-Z0[(size_t)q * n + k], Bq[(size_t)q * n + i], v);
    }
    Rp[e] = v;
  }
}

template <bool VLDS>
__global__ void __launch_bounds__(1024) k_post_eigen(const double* __restrict__ lraw_in, const double* __restrict__ Vg, int n, PostEigenArgs pa,
                                                     int64_t* stat) {
  __shared__ double Ginv[CMAX * CMAX];
  __shared__ double Gw[CMAX * 2 * CMAX];
  __shared__ int s_neg;
  extern __shared__ __attribute__((aligned(16))) double shv[];
  post_eigen_body<VLDS>(lraw_in, Vg, n, pa, stat, shv, Ginv, Gw, &s_neg);
}

__global__ void __launch_bounds__(1024) k_jacobi_lds(const double* __restrict__ Ag, double* __restrict__ Vg, int n,
                                                     double* __restrict__ lraw, int64_t* stat, double stop2, PostEigenArgs pa, int fuse_post,
                                                     int pe_off /* doubles: where the post-eigen work arrays start in smem */) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ double s_anorm;
  __shared__ int s_pe_neg, s_pe_last;
  // the fast path (kernels_eig.hip: k_eigf_*) ran ahead of this launch and its checks passed: no Jacobi (workgroup-uniform).
  // fuse_post: the first workgroup then does the post-eigen work (what the next launch used to do) instead of returning.
  if (stat[ST_EIG_FAST] == 1 && __longlong_as_double((long long)stat[ST_EIG_BAD]) <= 1.0) {
    if (fuse_post && blockIdx.x == 0) post_eigen_body<true>(lraw, Vg, n, pa, stat, smem, smem + pe_off, smem + pe_off + CMAX * CMAX, &s_pe_neg);
    return;
  }
  const int tid = threadIdx.x, nt = blockDim.x;
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
  const int N = n + (n & 1), NP2 = N / 2, ld = N + 1;       // ld: leading dimension of the V slices (odd)
  const int rows_per = (n + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per, r1 = min(n, r0 + rows_per), nr = max(0, r1 - r0);
  // All buffers are addressed as smem[offset] so that hipcc keeps them in the LDS address space (a runtime-selected
  // pointer array degrades every access to flat_load/flat_store).
  //   A ping/pong: planar blocks (see above);  V slices: row-major Vs[r*ld + pos]
  const int PL = (NP2 * (NP2 - 1) / 2 + 1) & ~1;      // plane size (off-diagonal blocks), even
  const int DG = 4 * PL;                              // a_pp[NP2], a_qq[NP2], a_pq[NP2] behind the planes
  const int AO = (4 * PL + 3 * NP2 + 1) & ~1;         // size of one A copy
  const int VO = rows_per * ld;                       // size of one V slice copy
  const int V0 = 2 * AO;                              // first V copy
  const int REC = (2 * AO + 2 * VO + 1) & ~1;         // NP2 x 2 : (c, s) per slot, 16-byte aligned
  const int REL = REC + 4 * NP2;                      // NP2 (behind the two (c, s) tables)
  // offset of element (r, c) of the symmetric matrix inside one A copy
  auto aidx = [&](int r, int c) -> int {
    if (r > c) { const int t = r; r = c; c = t; }
    const int sr = r >> 1, sc = c >> 1;
    if (sr == sc) return DG + ((r == c) ? ((r & 1) ? NP2 + sr : sr) : 2 * NP2 + sr);
    return ((r & 1) * 2 + (c & 1)) * PL + sc * (sc - 1) / 2 + sr;
  };
  for (int e = tid; e < 2 * AO + 2 * VO; e += nt) smem[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < n * n; e += nt) { const int i = e % n, j = e / n; if (i <= j) smem[aidx(i, j)] = Ag[e]; }
  for (int r = tid; r < nr; r += nt) smem[V0 + r * ld + (r0 + r)] = 1.0;
  __syncthreads();
  if (tid == 0) {
    double m = 0.0;
    for (int i = 0; i < n; ++i) m = fmax(m, fabs(smem[aidx(i, i)]));
    s_anorm = m;
  }
  __syncthreads();
  const double eps = 2.220446049250313e-16;
  const double tol2 = (4.0 * eps) * (4.0 * eps);
  const double floor2 = (eps * s_anorm) * (eps * s_anorm);
  // ---- fixed thread -> work maps (identical in every round) -------------------------------------------
  const int nblk = NP2 * (NP2 - 1) / 2;                 // off-diagonal slot pairs s1 < s2 (diagonal blocks: phase 1)
  constexpr int MAXB = 2;                               // blocks per thread: NP2 <= 62 -> nblk <= 1891 <= 2 x 960
  const int NW = nt - 64;                               // worker threads (the last wave is reserved for the angle lanes)
  int b_s1[MAXB], b_s2[MAXB], b_src[MAXB], b_dst[MAXB][4];
  bool b_ok[MAXB];
#pragma unroll
  for (int u = 0; u < MAXB; ++u) {
    const int b = (tid < NW) ? tid + u * NW : nblk;
    // a lane without a block still runs the branch-free loads and arithmetic (on slot 0 / block 0) and skips only the stores
    b_s1[u] = 0; b_s2[u] = 0; b_src[u] = 0; b_ok[u] = false;
#pragma unroll
    for (int w = 0; w < 4; ++w) b_dst[u][w] = 0;
    if (b < nblk) {
      b_ok[u] = true;
      int row = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
      while (row * (row + 1) / 2 > b) --row;
      while ((row + 1) * (row + 2) / 2 <= b) ++row;
      const int s2 = row + 1, s1 = b - row * (row + 1) / 2;   // s1 < s2
      b_s1[u] = s1; b_s2[u] = s2;
      b_src[u] = b;                                       // = tri(s1, s2): the block's index in every plane
      const int rp = jac_pi(2 * s1, NP2), rq = jac_pi(2 * s1 + 1, NP2), cp = jac_pi(2 * s2, NP2), cq = jac_pi(2 * s2 + 1, NP2);
      b_dst[u][0] = aidx(rp, cp); b_dst[u][1] = aidx(rp, cq);
      b_dst[u][2] = aidx(rq, cp); b_dst[u][3] = aidx(rq, cq);
    }
  }
  // V items (slot, local row), dealt from the last thread downwards
  const int nvit = NP2 * nr;
  const int nvthr = NW;
  const int vitem = (tid < nvthr) ? nvthr - 1 - tid : -1;
  int v_slot = 0, v_src = 0, v_d0 = 0, v_d1 = 0;
  bool v_ok = false;
  if (vitem >= 0 && vitem < nvit) {
    v_ok = true;
    v_slot = vitem / nr; const int r = vitem % nr;
    v_src = r * ld + 2 * v_slot; v_d0 = r * ld + jac_pi(2 * v_slot, NP2); v_d1 = r * ld + jac_pi(2 * v_slot + 1, NP2);
  }
  // ---- angle lanes: the last wave; lane `at` owns pair slot `at` ------------------------------------------
  // In round r lane `at` (i) rotates its own diagonal block with its (c, s)_r and (ii) ALREADY derives (c, s)_{r+1}:
  // the next diagonal block of slot `at` is made of three elements of A_{r+1} = pi(J' A_r J) -- the rotated diagonal
  // entries of the two slots its positions come from and one element of the rotated block between them -- which it
  // recomputes from A_r and (c, s)_r with the very expressions the owning threads use.  Blocks, V and angles thus
  // all read round-r data only: ONE barrier per round.
  const int at = tid - (nt - 64);
  const bool angle_lane = at >= 0 && at < NP2;
  // the angle wave runs one long dependent fp64 chain per round and shares its SIMD with three block waves: let it win
  // the issue arbitration (cdna_hip_programming.md T5, static form)
  if (__builtin_amdgcn_readfirstlane(tid >> 6) == (nt >> 6) - 1) __builtin_amdgcn_s_setprio(3);
  int d_pp = 0, d_qq = 0, d_pq = 0;                     // where slot at's diagonal block goes
  int o_pp = 0, o_qq = 0, o_pq = 0;                     // ... and where it sits now
  int sa = 0, sb = 0, ia = 0, ib = 0, ix = 0, iy = 0, xb = 0;
  if (angle_lane) {
    const int rp = jac_pi(2 * at, NP2), rq = jac_pi(2 * at + 1, NP2);
    d_pp = aidx(rp, rp); d_qq = aidx(rq, rq); d_pq = aidx(rp, rq);
    o_pp = DG + at; o_qq = DG + NP2 + at; o_pq = DG + 2 * NP2 + at;
    int pa = 0, pb = 0;                                 // pre-images of positions 2*at, 2*at+1 under pi
    for (int pos = 0; pos < N; ++pos) { const int q = jac_pi(pos, NP2); if (q == 2 * at) pa = pos; if (q == 2 * at + 1) pb = pos; }
    sa = pa >> 1; ia = pa & 1; sb = pb >> 1; ib = pb & 1;
    int sx, sy;
    if (sa < sb) { sx = sa; ix = ia; sy = sb; iy = ib; } else { sx = sb; ix = ib; sy = sa; iy = ia; }
    xb = sy * (sy - 1) / 2 + sx;                        // the block between the two source slots, in every plane
  }
  const int REC1 = REC + 2 * NP2;                       // second (c, s) table (ping-pong with the A copies)
  auto angle_of = [&](double app, double aqq, double apq, double& c, double& sn, double& rel) {
    c = 1.0; sn = 0.0;
    const double pp = fmax(fabs(app * aqq), floor2), a2 = apq * apq;
    if (a2 > tol2 * pp) {
      // cos 2phi = |d|/h, sin 2phi = sign(d) 2 apq / h, |phi| <= pi/4
      const double d = aqq - app;
      const double ih = nr_rsqrt(fma(d, d, 4.0 * a2));
      const double c2 = fma(0.5 * fabs(d), ih, 0.5);
      const double ic = nr_rsqrt(c2);
      c = c2 * ic;
      sn = copysign(apq * ih, apq * copysign(1.0, d)) * ic;
      // ~ (relative off-diagonal)^2; only gates the stop rule.  A LARGE-angle rotation also keeps the iteration going: between
      // (numerically) equal diagonal entries any off-diagonal above the threshold rotates by 45 degrees, which re-mixes the
      // cross terms this sweep had already annihilated -- the sweep is then not the final, quadratically convergent one
      // (four 16-fold eigenvalues: K - U diag U' stayed at 3e-9 |K|; tools/fuzz_eig.py).  On distinct spectra the last
      // sweep's angles are ~1e-9 and nothing changes.
      rel = fmax(rel, a2 * __builtin_amdgcn_rcp(pp));
      if (fabs(sn) > 1e-3) rel = fmax(rel, 1.0);
    }
  };
  // One round of a worker lane, branch-free up to the stores: EVERY LDS read first (the blocks' angles and entries, the V
  // item's angle and pair), then the arithmetic, then the writes.  A and An (and the V copies) alias as far as the compiler
  // can tell, so a read placed behind a write waits for it, and a lane with two items would walk two LDS round trips per
  // round.  U1 / HV (wave-uniform): some lane of the wave owns a second block / a V item; lanes that own none load slot 0.
#ifdef JAC_PROF
  __shared__ unsigned long long s_prof[16][5];
  bool prof = false;
#endif
  using TrueT = std::integral_constant<bool, true>;
  using FalseT = std::integral_constant<bool, false>;
  const bool w_u1 = __any(b_ok[1]) != 0, w_v = __any(v_ok) != 0;
  auto worker_round = [&](auto U1c, auto HVc, const double* A, double* An, const double* rec, const double* Vc, double* Vn) {
    constexpr bool U1 = decltype(U1c)::value, HV = decltype(HVc)::value;
    constexpr int NU = U1 ? 2 : 1;
    dpair cs1[NU], cs2[NU], x0[NU], x1[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      cs1[u] = *reinterpret_cast<const dpair*>(rec + 2 * b_s1[u]);
      cs2[u] = *reinterpret_cast<const dpair*>(rec + 2 * b_s2[u]);
      x0[u] = (dpair){A[b_src[u]], A[2 * PL + b_src[u]]};            // (b00, b10): column 2*s2
      x1[u] = (dpair){A[PL + b_src[u]], A[3 * PL + b_src[u]]};       // (b01, b11): column 2*s2+1
    }
    dpair csv = (dpair){1.0, 0.0}, xy = (dpair){0.0, 0.0};
    if constexpr (HV) {
      csv = *reinterpret_cast<const dpair*>(rec + 2 * v_slot);
      xy = (dpair){Vc[v_src], Vc[v_src + 1]};
    }
    // A: off-diagonal blocks B' = J1' B J2 written through pi.  The empty asm statements pin the arithmetic (and with it the
    // loads) in front of the lane-conditional stores: hipcc otherwise sinks a block's loads into its `if`, behind the
    // previous block's stores.
    double o[NU][4];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const double c1 = cs1[u][0], sn1 = cs1[u][1], c2 = cs2[u][0], sn2 = cs2[u][1];
      const double t00 = fma(c2, x0[u][0], -sn2 * x1[u][0]), t01 = fma(sn2, x0[u][0], c2 * x1[u][0]);
      const double t10 = fma(c2, x0[u][1], -sn2 * x1[u][1]), t11 = fma(sn2, x0[u][1], c2 * x1[u][1]);
      o[u][0] = fma(c1, t00, -sn1 * t10); o[u][1] = fma(c1, t01, -sn1 * t11);
      o[u][2] = fma(sn1, t00, c1 * t10); o[u][3] = fma(sn1, t01, c1 * t11);
      asm volatile("" : "+v"(o[u][0]), "+v"(o[u][1]), "+v"(o[u][2]), "+v"(o[u][3]));
    }
#ifdef JAC_PROF
    if (prof) s_prof[tid >> 6][1] = __builtin_amdgcn_s_memtime();
#endif
    double v0 = 0.0, v1 = 0.0;
    if constexpr (HV) {   // V slice: columns (2s, 2s+1) of one local row
      v0 = fma(csv[0], xy[0], -csv[1] * xy[1]); v1 = fma(csv[1], xy[0], csv[0] * xy[1]);
      asm volatile("" : "+v"(v0), "+v"(v1));
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (b_ok[u]) { An[b_dst[u][0]] = o[u][0]; An[b_dst[u][1]] = o[u][1]; An[b_dst[u][2]] = o[u][2]; An[b_dst[u][3]] = o[u][3]; }
    if constexpr (HV) {
      if (v_ok) { Vn[v_d0] = v0; Vn[v_d1] = v1; }
    }
  };
  const double jac_stop2 = stop2;
  const bool is_angle_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == (nt >> 6) - 1;
  int cur = 0, sweep = 0, dpos = N - 1;                 // dpos: where the zero pad row/col of an odd n currently sits
  double mc = 1.0, ms = 0.0, myrel = 0.0;               // this lane's (c, s) for the current round
  if (angle_lane) {                                     // prologue: angles of round 0 straight from A_0
    angle_of(smem[o_pp], smem[o_qq], smem[o_pq], mc, ms, myrel);
    *reinterpret_cast<dpair*>(smem + REC + 2 * at) = (dpair){mc, ms};
  }
  __syncthreads();
  for (; sweep < 30; ++sweep) {
    for (int round = 0; round < N - 1; ++round) {
#ifdef JAC_PROF
      prof = sweep == 2 && round == 10 && blockIdx.x == 0 && (tid & 63) == 0;
      if (prof) s_prof[tid >> 6][0] = __builtin_amdgcn_s_memtime();
#endif
      const double* A = smem + (cur ? AO : 0);
      double* An = smem + (cur ? 0 : AO);
      const double* rec = smem + (cur ? REC1 : REC);
      double* recn = smem + (cur ? REC : REC1);
      // The angle wave and the worker waves take wave-uniform (scalar) branches: a wave must not walk the other role's
      // code with an empty exec mask (its LDS instructions would still take issue slots on the critical path).
      if (!is_angle_wave) {
        const double* Vc = smem + V0 + (cur ? VO : 0);
        double* Vn = smem + V0 + (cur ? 0 : VO);
        if (w_u1) { if (w_v) worker_round(TrueT{}, TrueT{}, A, An, rec, Vc, Vn); else worker_round(TrueT{}, FalseT{}, A, An, rec, Vc, Vn); }
        else      { if (w_v) worker_round(FalseT{}, TrueT{}, A, An, rec, Vc, Vn); else worker_round(FalseT{}, FalseT{}, A, An, rec, Vc, Vn); }
        if (vitem >= 0) for (int item = vitem + nvthr; item < nvit; item += nvthr) {   // only when NP2 * rows_per exceeds the lanes left
          const int slot = item / nr, r = item % nr;
          const dpair csw = *reinterpret_cast<const dpair*>(rec + 2 * slot);
          const dpair xw = (dpair){Vc[r * ld + 2 * slot], Vc[r * ld + 2 * slot + 1]};
          Vn[r * ld + jac_pi(2 * slot, NP2)] = fma(csw[0], xw[0], -csw[1] * xw[1]);
          Vn[r * ld + jac_pi(2 * slot + 1, NP2)] = fma(csw[1], xw[0], csw[0] * xw[1]);
        }
      } else if (angle_lane) {
      // ---- angle lanes: own diagonal block of round r, then (c, s) of round r+1 -----------------------------------
        auto rot_diag = [&](double c, double sn, double app, double aqq, double apq, double& npp, double& nqq) {
          const double cc = c * c, ss = sn * sn, xx = 2.0 * c * sn * apq;
          npp = fma(cc, app, fma(ss, aqq, -xx));
          nqq = fma(ss, app, fma(cc, aqq, xx));
        };
        // every LDS read first (one round trip; a read placed after the writes below would have to wait for them:
        // A and An alias as far as the compiler can tell), then the critical path -- the angles of round r+1 -- and the
        // lane's own diagonal block last
        const double app = A[o_pp], aqq = A[o_qq], apq = A[o_pq];
        const dpair ca = *reinterpret_cast<const dpair*>(rec + 2 * sa);
        const dpair cb = *reinterpret_cast<const dpair*>(rec + 2 * sb);
        const double a0 = A[DG + sa], a1 = A[DG + NP2 + sa], a2 = A[DG + 2 * NP2 + sa];
        const double e0 = A[DG + sb], e1 = A[DG + NP2 + sb], e2 = A[DG + 2 * NP2 + sb];
        const double b00 = A[xb], b01 = A[PL + xb], b10 = A[2 * PL + xb], b11 = A[3 * PL + xb];
        double a_pp, a_qq, b_pp, b_qq;
        rot_diag(ca[0], ca[1], a0, a1, a2, a_pp, a_qq);
        rot_diag(cb[0], cb[1], e0, e1, e2, b_pp, b_qq);
        const double napp = ia ? a_qq : a_pp, naqq = ib ? b_qq : b_pp;
        const dpair cx = (sa < sb) ? ca : cb, cy = (sa < sb) ? cb : ca;
        const double c1 = cx[0], sn1 = cx[1], c2 = cy[0], sn2 = cy[1];
        const double t0 = iy ? fma(sn2, b00, c2 * b01) : fma(c2, b00, -sn2 * b01);
        const double t1 = iy ? fma(sn2, b10, c2 * b11) : fma(c2, b10, -sn2 * b11);
        const double napq = ix ? fma(sn1, t0, c1 * t1) : fma(c1, t0, -sn1 * t1);
        const double omc = mc, oms = ms;                  // this round's (c, s) of the lane's own pair
        angle_of(napp, naqq, napq, mc, ms, myrel);
        *reinterpret_cast<dpair*>(recn + 2 * at) = (dpair){mc, ms};
        double npp, nqq;
        rot_diag(omc, oms, app, aqq, apq, npp, nqq);
        An[d_pp] = npp; An[d_qq] = nqq; An[d_pq] = (oms != 0.0) ? 0.0 : apq;
      }
      dpos = jac_pi(dpos, NP2);
      cur ^= 1;
#ifdef JAC_PROF
      if (prof) s_prof[tid >> 6][2] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (prof) s_prof[tid >> 6][3] = __builtin_amdgcn_s_memtime();
#endif
      __syncthreads();
#ifdef JAC_PROF
      if (prof) s_prof[tid >> 6][4] = __builtin_amdgcn_s_memtime();
#endif
    }
    // sweep verdict: the largest relative off-diagonal (squared) any pair met while its angles were derived
    if (angle_lane) { smem[REL + at] = myrel; myrel = 0.0; }
    __syncthreads();
    double mx = 0.0;
    for (int i = 0; i < NP2; ++i) mx = fmax(mx, smem[REL + i]);
    __syncthreads();
    // Jacobi converges quadratically: once every relative off-diagonal met in a sweep was below ~3e-8 (stop2 = 1e-15 on
    // the squared value) the sweep left them at rounding level, and a further (verification) sweep would not rotate
    // anything (tests/test_gpu_parity.py::test_eigensolver_accuracy)
    if (mx < jac_stop2) break;
  }
#ifdef JAC_PROF
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {
    const unsigned long long b = s_prof[0][0];
    for (int w = 0; w < (nt >> 6); ++w)
      printf("jac wave %2d start %5lld computed %5lld issued %5lld lds-done %5lld after-barrier %5lld\n", w, (long long)(s_prof[w][0] - b),
             (long long)(s_prof[w][1] - b), (long long)(s_prof[w][2] - b), (long long)(s_prof[w][3] - b), (long long)(s_prof[w][4] - b));
  }
#endif
  // positions -> compact output (skip the pad position of an odd n): eigenvalue = diagonal, eigenvector = V column
  const double* A = smem + (cur ? AO : 0);
  const double* Vc = smem + V0 + (cur ? VO : 0);
  const bool odd = (n & 1) != 0;
  if (blockIdx.x == 0) {
    for (int pos = tid; pos < N; pos += nt) {
      if (odd && pos == dpos) continue;
      const int o = (odd && pos > dpos) ? pos - 1 : pos;
      lraw[o] = A[aidx(pos, pos)];
    }
    if (tid == 0) {
      stat[ST_JACOBI_SWEEPS] = sweep + 1;
      stat[6] = (int64_t)(__builtin_amdgcn_s_memtime() - dbg_t0);      // shader cycles spent in the eigensolver
      stat[7] = (int64_t)(__builtin_amdgcn_s_memrealtime() - dbg_r0);  // 100 MHz ticks
    }
  }
  for (int e = tid; e < nr * N; e += nt) {
    const int r = e % nr, pos = e / nr;
    if (odd && pos == dpos) continue;
    const int o = (odd && pos > dpos) ? pos - 1 : pos;
    Vg[(size_t)o * n + (r0 + r)] = Vc[r * ld + pos];      // compact n x n column-major: Vg[col*n + row]
  }
  if (!fuse_post) return;
  // the last workgroup to get here has every slice of V (and workgroup 0's eigenvalues) in front of it: post-eigen work
  __threadfence();
  __syncthreads();
  if (tid == 0) s_pe_last = (atomicAdd((unsigned long long*)&stat[ST_EIG_DONE], 1ull) == (unsigned long long)gridDim.x - 1ull) ? 1 : 0;
  __syncthreads();
  if (!s_pe_last) return;
  __threadfence();
  post_eigen_body<true>(lraw, Vg, n, pa, stat, smem, smem + pe_off, smem + pe_off + CMAX * CMAX, &s_pe_neg);
}

// Global-memory variant for larger n (one workgroup; A, V in L2): the straightforward column/row form.
__global__ void __launch_bounds__(1024) k_jacobi_glb(double* __restrict__ A, double* __restrict__ V, int n,
                                                     double* __restrict__ lraw, int64_t* stat) {
  __shared__ int s_rot;
  __shared__ double s_anorm;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int N = n + (n & 1), NP2 = N / 2, ld = n;
  double* cs = V + (size_t)n * n;                        // NP2 x 2 (the caller allocates n*n + 4n + 16)
  int* pq = reinterpret_cast<int*>(cs + 2 * NP2);       // NP2 x 2
  if (tid == 0) { s_rot = 0; s_anorm = 0.0; }
  for (int e = tid; e < n * n; e += nt) V[e] = ((e % n) == (e / n)) ? 1.0 : 0.0;
  __syncthreads();
  if (tid == 0) {
    double m = 0.0;
    for (int i = 0; i < n; ++i) m = fmax(m, fabs(A[(size_t)i * ld + i]));
    s_anorm = m;
  }
  __syncthreads();
  const double eps = 2.220446049250313e-16;
  const double tol2 = (4.0 * eps) * (4.0 * eps);
  const double floor2 = (eps * s_anorm) * (eps * s_anorm);
  int sweep = 0;
  for (; sweep < 30; ++sweep) {
    for (int round = 0; round < N - 1; ++round) {
      for (int slot = tid; slot < NP2; slot += nt) {
        int p, q;
        if (slot == 0) { p = N - 1; q = round; }
        else { p = (round + slot) % (N - 1); q = (round - slot + (N - 1)) % (N - 1); }
        if (p > q) { const int t = p; p = q; q = t; }
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double app = A[(size_t)p * ld + p], aqq = A[(size_t)q * ld + q], apq = A[(size_t)q * ld + p];
          const double pp = fabs(app * aqq), a2 = apq * apq;
          if (a2 > tol2 * fmax(pp, floor2)) {
            const double d = aqq - app;
            const double h = sqrt(fma(d, d, 4.0 * a2));
            const double t = copysign(2.0 * apq, apq * copysign(1.0, d)) / (fabs(d) + h);
            c = 1.0 / sqrt(fma(t, t, 1.0));
            s = t * c;
            s_rot = 1;
          }
        } else { q = p; }
        cs[2 * slot] = c; cs[2 * slot + 1] = s;
        pq[2 * slot] = p; pq[2 * slot + 1] = q;
      }
      __threadfence_block();
      __syncthreads();
      for (int item = tid; item < NP2 * n * 2; item += nt) {
        const int slot = item / (2 * n), off = item % (2 * n);
        const double s = cs[2 * slot + 1];
        if (s != 0.0) {
          const double c = cs[2 * slot];
          const int p = pq[2 * slot], q = pq[2 * slot + 1];
          double* M = (off >= n) ? V : A;
          const int k = (off >= n) ? off - n : off;
          const double x = M[(size_t)p * ld + k], y = M[(size_t)q * ld + k];
          M[(size_t)p * ld + k] = fma(c, x, -s * y);
          M[(size_t)q * ld + k] = fma(s, x, c * y);
        }
      }
      __threadfence_block();
      __syncthreads();
      for (int item = tid; item < NP2 * n; item += nt) {
        const int slot = item / n, j = item - slot * n;
        const double s = cs[2 * slot + 1];
        if (s != 0.0) {
          const double c = cs[2 * slot];
          const int p = pq[2 * slot], q = pq[2 * slot + 1];
          const double x = A[(size_t)j * ld + p], y = A[(size_t)j * ld + q];
          A[(size_t)j * ld + p] = fma(c, x, -s * y);
          A[(size_t)j * ld + q] = fma(s, x, c * y);
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    const int rot = s_rot;
    __syncthreads();
    if (tid == 0) s_rot = 0;
    __syncthreads();
    if (!rot) break;
  }
  for (int i = tid; i < n; i += nt) lraw[i] = A[(size_t)i * ld + i];
  if (tid == 0) stat[ST_JACOBI_SWEEPS] = sweep + 1;
}

int jacobi_lds_max_n() { return JAC_NMAX; }

// pe != nullptr: the caller wants the post-eigen work done as well; *fused says whether this launch did it (LDS permitting)
int launch_jacobi(blmm_ctx* ctx, double* A, double* V, int n, double* lraw, int64_t* stat, const PostEigenArgs* pe, bool* fused) {
  if (fused) *fused = false;
  if (n <= JAC_NMAX) {
    const int N = n + (n & 1), NP2 = N / 2, ld = N + 1;
    static const int nwg_env = dev_env("BLMM_JAC_NWG") ? atoi(dev_env("BLMM_JAC_NWG")) : 0;
    const int nwg = (nwg_env >= 1 && nwg_env <= 64) ? nwg_env : JAC_NWG;
    const int rows_per = (n + nwg - 1) / nwg;
    const int PL = (NP2 * (NP2 - 1) / 2 + 1) & ~1, AO = (4 * PL + 3 * NP2 + 1) & ~1;
    size_t lds = sizeof(double) * ((size_t)2 * AO + (size_t)2 * rows_per * ld + 5 * NP2 + 2) + 64;
    // post-eigen work in the same launch: its arrays (2 n^2 + 3 n c + n doubles) and the Gauss-Jordan work arrays behind them
    static const bool fuse_on = !(dev_env("BLMM_EIG_FUSE_POST") && dev_env("BLMM_EIG_FUSE_POST")[0] == '0');   // A/B: the separate launch of round 3
    PostEigenArgs pa{};
    int fuse = 0, pe_off = 0;
    if (pe && fuse_on) {
      pe_off = (int)(((size_t)2 * n * n + (size_t)3 * n * pe->c + n + 1) & ~(size_t)1);
      const size_t lds_pe = sizeof(double) * ((size_t)pe_off + 3 * CMAX * CMAX) + 64;
      if (lds_pe <= 150 * 1024) { fuse = 1; pa = *pe; if (lds_pe > lds) lds = lds_pe; }
    }
    if (fused) *fused = fuse != 0;
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_jacobi_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const double stop2 = dev_env("BLMM_JAC_STOP2") ? atof(dev_env("BLMM_JAC_STOP2")) : 1e-15;
    // 768 threads (11 worker waves + the angle wave) when two items per worker thread cover the blocks: fewer waves at
    // the barrier and in the LDS queue measured ~8 % faster than 1024 at n = 79.  BLMM_JAC_NT overrides.
    static const int nt_env = dev_env("BLMM_JAC_NT") ? atoi(dev_env("BLMM_JAC_NT")) : 0;
    const int nblk = NP2 * (NP2 - 1) / 2;
    int nthr = (nblk <= 2 * (768 - 64)) ? 768 : 1024;
    if (nt_env >= 256 && nt_env <= 1024 && nt_env % 64 == 0 && nblk <= 2 * (nt_env - 64)) nthr = nt_env;
    if (fuse) BLMM_LAUNCH_STOP(ctx, k_jacobi_lds, dim3(nwg), dim3(nthr), lds, A, V, n, lraw, stat, stop2, pa, fuse, pe_off);   // (the eigen phase's last launch)
    else hipLaunchKernelGGL(k_jacobi_lds, dim3(nwg), dim3(nthr), lds, ctx->stream, A, V, n, lraw, stat, stop2, pa, fuse, pe_off);
  } else {
    hipLaunchKernelGGL(k_jacobi_glb, dim3(1), dim3(1024), 0, ctx->stream, A, V, n, lraw, stat);
  }
  KCHECK();
  return BLMM_OK;
}

// Jacobi launch (a no-op on the device when the fast eigen path's result stood) with the post-eigen work in its tail when the LDS
// allows; *fused = false: the caller launches launch_post_eigen behind it as before
int launch_jacobi_post(blmm_ctx* ctx, double* A, double* V, int n, double* lraw, int64_t* stat, const double* Zs, const double* dweights,
                       int c, int npad, int ldr, int decomp, int centered, double* lam, double* U, double* Z0, double* Rp, bool* fused) {
  const PostEigenArgs pa{Zs, dweights, c, npad, ldr, decomp, centered, lam, U, Z0, Rp, nullptr};
  return launch_jacobi(ctx, A, V, n, lraw, stat, &pa, fused);
}

// ---- the same steps as k_post_eigen for n beyond its LDS budget, spread over the chip (k_post_eigen<false>, one
//      workgroup walking n^2 strided global accesses, took 0.9 ms at n = 500 and 3.9 ms at n = 1000) ---------------------
// (1) rank of every eigenvalue (ascending; svd: |lambda| descending; ties by index) -> lam[rank], rankof[i]
__global__ void __launch_bounds__(256) k_pe_rank(const double* __restrict__ lraw_in, int n, int decomp, double* __restrict__ lam,
                                                 int* __restrict__ rankof, int64_t* stat) {
  // the values every thread compares against go through LDS, 256 at a time (read straight from memory the loop was one
  // dependent scalar load per iteration: 55 us at n = 500, 109 us at n = 1000)
  __shared__ double s_l[256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool svd = decomp == BLMM_SVD;
  const double li = (i < n) ? (svd ? fabs(lraw_in[i]) : lraw_in[i]) : 0.0;
  // The own eigensolvers hand the eigenvalues over in ascending order: every workgroup checks that (each thread a strided share of the
  // adjacent pairs, all of them -- 4 n loads in total at n = 1000) and, where it holds, the stable rank of entry i IS i; the
  // counting loop below (n iterations per thread: 47 us at n = 1000) is for the SVD order and for a solver that did not sort.
  int unsorted = svd ? 1 : 0;
  if (!svd)
    for (int j = threadIdx.x; j + 1 < n; j += blockDim.x) unsorted |= !(lraw_in[j] <= lraw_in[j + 1]);
  if (!__syncthreads_or(unsorted)) {
    if (i >= n) return;
    lam[i] = li;
    rankof[i] = i;
    if (li < -1e-7) atomicAdd((unsigned long long*)&stat[ST_NEG_EIG], 1ull);
    return;
  }
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int jn = (n - j0 < 256) ? n - j0 : 256;
    __syncthreads();
    if ((int)threadIdx.x < jn) { const double v = lraw_in[j0 + threadIdx.x]; s_l[threadIdx.x] = svd ? fabs(v) : v; }
    __syncthreads();
    for (int u = 0; u < jn; ++u) {
      const double lj = s_l[u];
      const int j = j0 + u;
      if (svd) rank += (lj > li) || (lj == li && j < i);
      else rank += (lj < li) || (lj == li && j < i);
    }
  }
  if (i >= n) return;
  lam[rank] = li;
  rankof[i] = rank;
  if (li < -1e-7) atomicAdd((unsigned long long*)&stat[ST_NEG_EIG], 1ull);
}
// (2) one workgroup per eigenvector i: sign (largest-magnitude component positive, first one on ties), U[rank] = sg V[i],
//     Z0[q][rank] = <U[rank], Zs[q]>
__global__ void __launch_bounds__(256) k_pe_cols(const double* __restrict__ V, const int* __restrict__ rankof,
                                                 const double* __restrict__ Zs, int n, int c, double* __restrict__ U,
                                                 double* __restrict__ Z0) {
  __shared__ double s_v[256];
  __shared__ int s_k[256];
  __shared__ double s_z[CMAX][4];
  const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double* v = V + (size_t)i * n;
  double big = 0.0; int bk = n;
  for (int k = t; k < n; k += 256) { const double x = v[k]; if (fabs(x) > fabs(big)) { big = x; bk = k; } }
  s_v[t] = big; s_k[t] = bk;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      const double a = s_v[t], b = s_v[t + o];
      if (fabs(b) > fabs(a) || (fabs(b) == fabs(a) && s_k[t + o] < s_k[t])) { s_v[t] = b; s_k[t] = s_k[t + o]; }
    }
    __syncthreads();
  }
  const double sg = (s_v[0] < 0.0) ? -1.0 : 1.0;
  const int rank = rankof[i];
  double zacc[CMAX] = {};
  for (int k = t; k < n; k += 256) {
    const double u = sg * v[k];
    U[(size_t)rank * n + k] = u;
    for (int q = 0; q < c; ++q) zacc[q] = fma(u, Zs[(size_t)q * n + k], zacc[q]);
  }
  for (int q = 0; q < c; ++q) {
    double z = zacc[q];
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
    if (lane == 0) s_z[q][wave] = z;
  }
  __syncthreads();
  if (t < c) Z0[(size_t)t * n + rank] = (s_z[t][0] + s_z[t][1]) + (s_z[t][2] + s_z[t][3]);
}
// (3) Ginv = (Z0'Z0)^-1 and Bq = Ginv (Zs' Wd)   (one workgroup; c <= CMAX)
__global__ void __launch_bounds__(256) k_pe_bq(const double* __restrict__ Z0, const double* __restrict__ Zs,
                                               const double* __restrict__ wd, int n, int c, double* __restrict__ Bq) {
  __shared__ double Ginv[CMAX * CMAX], s_g[CMAX * CMAX][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int ab = 0; ab < c * c; ++ab) {
    const int a = ab / c, b = ab % c;
    double g = 0.0;
    for (int k = t; k < n; k += 256) g = fma(Z0[(size_t)a * n + k], Z0[(size_t)b * n + k], g);
    for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
    if (lane == 0) s_g[ab][wave] = g;
  }
  __syncthreads();
  if (t == 0) {
    __shared__ double G[CMAX][2 * CMAX];        // (LDS: 16 KB at CMAX = 32 -- as a local array it would be scratch memory of every thread)
    for (int a = 0; a < c; ++a)
      for (int b = 0; b < c; ++b) {
        const int ab = a * c + b;
        G[a][b] = (s_g[ab][0] + s_g[ab][1]) + (s_g[ab][2] + s_g[ab][3]);
        G[a][c + b] = (a == b) ? 1.0 : 0.0;
      }
    for (int a = 0; a < c; ++a) {
      int piv = a;
      for (int r = a + 1; r < c; ++r) if (fabs(G[r][a]) > fabs(G[piv][a])) piv = r;
      if (piv != a) for (int b = 0; b < 2 * c; ++b) { double tt = G[a][b]; G[a][b] = G[piv][b]; G[piv][b] = tt; }
      const double d = 1.0 / G[a][a];
      for (int b = 0; b < 2 * c; ++b) G[a][b] *= d;
      for (int r = 0; r < c; ++r) if (r != a) { const double f = G[r][a]; for (int b = 0; b < 2 * c; ++b) G[r][b] -= f * G[a][b]; }
    }
    for (int a = 0; a < c; ++a) for (int b = 0; b < c; ++b) Ginv[a * CMAX + b] = G[a][c + b];
  }
  __syncthreads();
  for (int e = t; e < n * c; e += 256) {
    const int i = e % n, q = e / n;
    double sacc = 0;
    for (int r = 0; r < c; ++r) sacc = fma(Ginv[q * CMAX + r], Zs[(size_t)r * n + i], sacc);
    Bq[(size_t)q * n + i] = sacc * (wd ? wd[i] : 1.0);
  }
}
// (4) Rp[i][k] = U[k][i] wd_i - sum_q Z0[q][k] Bq[q][i]   (32 x 32 tiles through LDS: both sides coalesced)
__global__ void __launch_bounds__(256) k_pe_rp(const double* __restrict__ U, const double* __restrict__ Z0,
                                               const double* __restrict__ Bq, const double* __restrict__ wd, int n, int c,
                                               int npad, int ldr, int centered, double* __restrict__ Rp) {
  __shared__ double tile[32][33];
  const int k0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {                        // U[k0 + r][i0 + tx]
    const int k = k0 + r, i = i0 + tx;
    tile[r][tx] = (k < n && i < n) ? U[(size_t)k * n + i] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {                        // Rp[i0 + r][k0 + tx]
    const int i = i0 + r, k = k0 + tx;
    if (i >= npad || k >= ldr) continue;
    double v = 0.0;
    if (i < n && k < n) {
      v = tile[tx][r] * (wd ? wd[i] : 1.0);
      if (centered)
        for (int q = 0; q < c; ++q) v = fma(-Z0[(size_t)q * n + k], Bq[(size_t)q * n + i], v);
    }
    Rp[(size_t)i * ldr + k] = v;
  }
}

int launch_post_eigen(blmm_ctx* ctx, const double* lraw, const double* V, const double* Zs, const double* dweights, int n,
                      int c, int npad, int ldr, int decomp, int centered, double* lam, double* U, double* Z0, double* Rp,
                      int64_t* stat) {
  int rc = ensure(ctx, ctx->misc, sizeof(double) * ((size_t)n + (size_t)c * n + 64));
  if (rc) return rc;
  const size_t lds = sizeof(double) * ((size_t)2 * n * n + (size_t)3 * n * c + n);
  // the kernel's static LDS (the CMAX-sized Gauss-Jordan work arrays) counts against the 160 KB of the CU as well
  static const size_t lds_static = [] {
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_post_eigen<true>)) == hipSuccess ? (size_t)fa.sharedSizeBytes : (size_t)32768;
  }();
  const PostEigenArgs pa{Zs, dweights, c, npad, ldr, decomp, centered, lam, U, Z0, Rp, ptr<double>(ctx->misc)};
  if (lds + lds_static <= 158 * 1024) {
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_post_eigen<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BLMM_LAUNCH_STOP(ctx, (k_post_eigen<true>), dim3(1), dim3(1024), lds, lraw, V, n, pa, stat);
  } else {
    double* Bq = ptr<double>(ctx->misc);                        // c x n
    int* rankof = reinterpret_cast<int*>(Bq + (size_t)c * n);   // n ints (the workspace holds n more doubles)
    hipLaunchKernelGGL(k_pe_rank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, lraw, n, decomp, lam, rankof, stat);
    hipLaunchKernelGGL(k_pe_cols, dim3((unsigned)n), dim3(256), 0, ctx->stream, V, rankof, Zs, n, c, U, Z0);
    hipLaunchKernelGGL(k_pe_bq, dim3(1), dim3(256), 0, ctx->stream, Z0, Zs, dweights, n, c, Bq);
    hipLaunchKernelGGL(k_pe_rp, dim3((unsigned)((ldr + 31) / 32), (unsigned)((npad + 31) / 32)), dim3(256), 0, ctx->stream, U, Z0,
                       Bq, dweights, n, c, npad, ldr, centered, Rp);
  }
  KCHECK();
  return BLMM_OK;
}

// ------------------------------------------------------------------------------------------------
// rotation GEMM on the f64 matrix cores:  Out[k][j] = sum_i R[k,i] * In[i,j]
//   (Ut*y, Ut*X of src/transform_helpers.jl:34, with the centring folded into R).
// One wave = 16 input columns x (MBLK x 16) output rows; v_mfma_f64_16x16x4_f64:
//   A frag: lane l holds A[row = l&15][k = l>>4];  B frag: B[k = l>>4][col = l&15];
//   D: col = l&15, row = (l>>4) + 4*reg   (cdna_hip_programming.md §3, verified by tools/mb_f64.hip).
// In is column-major n x ncols; Out is row-major (npad rows) with leading dimension ldo.
// ------------------------------------------------------------------------------------------------
template <int MBLK>
__global__ void __launch_bounds__(256) k_rotate(const double* __restrict__ Rp, int ldr, int n, int npad,
                                                const double* __restrict__ In, int64_t ncols,
                                                double* __restrict__ Out, int64_t ldo, int64_t ncols_pad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t col = ((int64_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
  const int rb0 = blockIdx.y * MBLK;  // first 16-row block of this wave
  const int kk = lane >> 4;
  d4 acc[MBLK];
#pragma unroll
  for (int b = 0; b < MBLK; ++b) acc[b] = (d4){0, 0, 0, 0};
  const bool colok = col < ncols;
  const double* pin = In + (colok ? col : 0) * (int64_t)n;
  // four K steps per trip, every operand load of the trip issued before its first MFMA: a load placed next to its use
  // exposes one L2 round trip per K step (the kernel runs at ~2 waves per SIMD)
  for (int i0 = 0; i0 < npad; i0 += 16) {
    double bv[4], av[4][MBLK];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 4 * u + kk;
      bv[u] = (colok && i < n) ? pin[i] : 0.0;
      const double* pr = Rp + (size_t)(i < npad ? i : 0) * ldr + (lane & 15);
#pragma unroll
      for (int b = 0; b < MBLK; ++b) {
        const int r0 = (rb0 + b) * 16;
        av[u][b] = (r0 < ldr && i < npad) ? pr[r0] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int b = 0; b < MBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][b], bv[u], acc[b], 0, 0, 0);
  }
  if (col < ncols_pad) {
#pragma unroll
    for (int b = 0; b < MBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (rb0 + b) * 16 + kk + 4 * r;
        if (row < npad) Out[(int64_t)row * ldo + col] = acc[b][r];
      }
  }
}

// A handful of columns at large n (the single trait of scan / scan_perms): one matrix-vector product per column, rows of
// the output across the lanes, the contraction split over the four waves (k_rotate would walk n/4 dependent steps on a
// single wave column: 0.41 ms at n = 1000).
__global__ void __launch_bounds__(256) k_rotate_vec(const double* __restrict__ Rp, int ldr, int n, int npad,
                                                    const double* __restrict__ In, int64_t col, double* __restrict__ Out,
                                                    int64_t ldo) {
  __shared__ double s_acc[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const double* x = In + col * (int64_t)n;
  // eight rows per trip, their loads in flight together (one load pair per trip was one L2 round trip per row: 77 us at n = 1000)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int kc = (k < npad) ? k : 0;
  int i = wave;
  for (; i + 28 < n; i += 32) {
    double r[8], xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { r[u] = Rp[(size_t)(i + 4 * u) * ldr + kc]; xv[u] = x[i + 4 * u]; }
    a0 = fma(r[0], xv[0], a0); a1 = fma(r[1], xv[1], a1); a2 = fma(r[2], xv[2], a2); a3 = fma(r[3], xv[3], a3);
    a0 = fma(r[4], xv[4], a0); a1 = fma(r[5], xv[5], a1); a2 = fma(r[6], xv[6], a2); a3 = fma(r[7], xv[7], a3);
  }
  for (; i < n; i += 4) a0 = fma(Rp[(size_t)i * ldr + kc], x[i], a0);
  const double acc = (k < npad) ? (a0 + a1) + (a2 + a3) : 0.0;
  s_acc[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && k < npad) Out[(int64_t)k * ldo + col] = (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]);
}

// Rotation for large n (K dimension = n in the hundreds or thousands): a real GEMM.  Workgroup = 4 waves (2 x 2), tile
// 128 rows of Out (k) x 32 NB columns; wave tile 64 x 16 NB = 4 x NB MFMA blocks.  No LDS: a 64-cycle f64 MFMA needs few operand
// bytes, so fragments come straight from L2 -- but In is column-major (a column's n values contiguous), which makes the
// natural fragment (16 lanes = 16 columns at ONE i) a 16-way strided gather.  The contraction order inside a trip of 32 i
// is therefore permuted: lane group gq = lane >> 4 takes i = i0 + 8 gq + s at MFMA step s, so a lane reads 8 CONSECUTIVE
// doubles of its column (64 bytes, four 16-byte loads) per trip, and the Rp fragment of a step is 16 consecutive doubles of
// one row of Rp.  Any fixed assignment of i to (step, lane group) is a valid order of the sum, and it is the same for
// every output element: a column's rotated values do not depend on what else is in the call (the sharding contract).
// Block order: an XCD (blockIdx % 8) walks whole column tiles, all k tiles of one after the other, so In is fetched from
// HBM once and re-read from that XCD's L2.
typedef double d2ua __attribute__((ext_vector_type(2), aligned(8)));
// TR: consecutive rows a lane reads of its column per trip (a trip = 4 TR rows).  TR = 8: 64-byte pieces, the next trip's
// B fragments fetched after the trip's MFMAs (2 waves per SIMD cover each other).  TR = 4: 32-byte pieces, half the fragment
// registers -- room to prefetch the next trip's fragments under the current trip's 64 MFMAs at NB = 4.
template <int NB, bool PREF, int TR = 8>
__global__ void __launch_bounds__(256, 2) k_rotate_big(const double* __restrict__ Rp, int ldr, int n, int npad,
                                                       const double* __restrict__ In, int64_t ncols,
                                                       double* __restrict__ Out, int64_t ldo, int64_t ncols_pad, int nkt,
                                                       int64_t nct) {
  constexpr int MB = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, gq = lane >> 4;
  // XCD-aware tile walk
  const int64_t bid = blockIdx.x;
  const int64_t xcd = bid & 7, idx = bid >> 3;
  const int64_t ctile = (idx / nkt) * 8 + xcd;
  const int ktile = (int)(idx % nkt);
  if (ctile >= nct) return;
  const int k0 = ktile * 128 + (wave >> 1) * 64;
  const int64_t col0 = ctile * (32 * NB) + (wave & 1) * (16 * NB);
  d4 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (d4){0, 0, 0, 0};
  const double* pcol[NB];
  bool cok[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int64_t col = col0 + 16 * b + c16;
    cok[b] = col < ncols;
    pcol[b] = In + (cok[b] ? col : 0) * (int64_t)n;
  }
  auto loadB = [&](double (&bv)[NB][TR], int i0) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int ib = i0 + TR * gq;
      if (cok[b] && ib + TR <= n) {
#pragma unroll
        for (int h = 0; h < TR / 2; ++h) { const d2ua v = *reinterpret_cast<const d2ua*>(pcol[b] + ib + 2 * h); bv[b][2 * h] = v[0]; bv[b][2 * h + 1] = v[1]; }
      } else {
#pragma unroll
        for (int s = 0; s < TR; ++s) bv[b][s] = (cok[b] && ib + s < n) ? pcol[b][ib + s] : 0.0;
      }
    }
  };
  auto loadA = [&](double (&av)[MB], int i) {     // Rp has npad rows (zero beyond n), ldr columns
    const double* pr = Rp + (size_t)(i < npad ? i : 0) * ldr + c16;
#pragma unroll
    for (int a = 0; a < MB; ++a) { const int k = k0 + 16 * a; av[a] = (i < npad && k + c16 < ldr) ? pr[k] : 0.0; }
  };
  // Full trips (i0 + 32 <= n) load through buffer descriptors with real bounds: a fragment of Rp beyond its npad rows and
  // a column of In beyond ncols read as zero without a branch, the per-lane part of every address is one 32-bit offset
  // fixed for the whole loop and the trip enters through the scalar offset.  (The guarded loads above cost ~270 scalar /
  // vector instructions per 128 MFMAs, 43 of them exec-mask branches: the matrix pipe was busy 70 % of the time.)  The
  // last, partial trip keeps the guarded loads -- a row beyond n of a column must not read the next column's values.
  typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2r __attribute__((ext_vector_type(2)));
  const uint64_t bytesA = (uint64_t)npad * (uint64_t)ldr * 8u, bytesB = (uint64_t)ncols * (uint64_t)n * 8u;
  const bool fast = bytesA < 0xffffff00ull && bytesB < 0xffffff00ull;
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Rp), 0, (unsigned)(fast ? bytesA : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(In), 0, (unsigned)(fast ? bytesB : 0), 0x00020000);
  // (the scalar offset is not part of the hardware's range check: nothing below relies on it for an out-of-range access.
  // Rows of Rp stay below npad in a full trip; a column index beyond ldr -- rows of Out beyond npad, never stored -- is clamped)
  uint32_t voffA[MB];
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    const int kc = k0 + 16 * a + c16;
    voffA[a] = (uint32_t)((((int64_t)TR * gq) * ldr + (kc < ldr ? kc : ldr - 1)) * 8);
  }
  uint32_t voffB[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int64_t col = col0 + 16 * b + c16;
    // a column beyond ncols: an offset past the end of the buffer (reads as zero); clamped so that it stays in 32 bits
    voffB[b] = (col < ncols) ? (uint32_t)((col * (int64_t)n + TR * gq) * 8) : 0xffffff00u;
  }
  auto loadB_fast = [&](double (&bv)[NB][TR], int i0) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int h = 0; h < TR / 2; ++h) {
        const u32x4r v = __builtin_amdgcn_raw_buffer_load_b128(srdB, voffB[b] + 16 * h, (unsigned)i0 * 8u, 0);
        const d2 w = __builtin_bit_cast(d2, v);
        bv[b][2 * h] = w[0]; bv[b][2 * h + 1] = w[1];
      }
  };
  auto loadA_fast = [&](double (&av)[MB], int irow) {   // irow: the trip's row i0 + s (wave-uniform); the lane adds TR gq rows
#pragma unroll
    for (int a = 0; a < MB; ++a) {
      const u32x2r v = __builtin_amdgcn_raw_buffer_load_b64(srdA, voffA[a], (unsigned)irow * (unsigned)ldr * 8u, 0);
      av[a] = __builtin_bit_cast(double, v);
    }
  };
  constexpr int TRIP = 4 * TR;
  double bcur[NB][TR], bnext[PREF ? NB : 1][TR];
  const int nfull = fast ? (n / TRIP) * TRIP : 0;    // rows covered by full trips
  // one trip of 32 rows: FT = a full trip through the descriptors; the B fragments of the trip are in bcur
  auto trip = [&](auto FTc, int i0) {
    constexpr bool FT = decltype(FTc)::value;
    if constexpr (PREF) { if (i0 + TRIP < n) { if (i0 + TRIP < nfull) loadB_fast(bnext, i0 + TRIP); else loadB(bnext, i0 + TRIP); } }
    double a0[MB], a1[MB];
    if constexpr (FT) loadA_fast(a0, i0); else loadA(a0, i0 + TR * gq);
#pragma unroll
    for (int s = 0; s < TR; s += 2) {
      if constexpr (FT) loadA_fast(a1, i0 + s + 1); else loadA(a1, i0 + TR * gq + s + 1);
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[a], bcur[b][s], acc[a][b], 0, 0, 0);
      if (s + 2 < TR) { if constexpr (FT) loadA_fast(a0, i0 + s + 2); else loadA(a0, i0 + TR * gq + s + 2); }
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[a], bcur[b][s + 1], acc[a][b], 0, 0, 0);
    }
    if (i0 + TRIP < n) {
      if constexpr (PREF) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int s = 0; s < TR; ++s) bcur[b][s] = bnext[b][s];
      } else {
        if (i0 + TRIP < nfull) loadB_fast(bcur, i0 + TRIP); else loadB(bcur, i0 + TRIP);
      }
    }
  };
  if (nfull > 0) loadB_fast(bcur, 0); else loadB(bcur, 0);
  int i0 = 0;
  for (; i0 < nfull; i0 += TRIP) trip(std::true_type{}, i0);
  for (; i0 < n; i0 += TRIP) trip(std::false_type{}, i0);
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + 16 * a + gq + 4 * r;
      if (k >= npad) continue;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int64_t col = col0 + 16 * b + c16;
        if (col < ncols_pad) Out[(int64_t)k * ldo + col] = acc[a][b][r];
      }
    }
}

int launch_rotate(blmm_ctx* ctx, const double* Rp, int ldr, int n, int npad, const double* In, int64_t ncols,
                  double* Out, int64_t ldo, int64_t ncols_pad) {
  if (ncols > 0 && ncols <= 4 && n > 256) {
    BLMM_HIP(hipMemsetAsync(Out, 0, sizeof(double) * (size_t)npad * ldo, ctx->stream));   // pad columns read as zero
    for (int64_t col = 0; col < ncols; ++col)
      hipLaunchKernelGGL(k_rotate_vec, dim3((unsigned)((npad + 63) / 64)), dim3(256), 0, ctx->stream, Rp, ldr, n, npad, In, col, Out, ldo);
    KCHECK();
    return BLMM_OK;
  }
  if (n > 160 && ncols >= 64 && !(dev_env("BLMM_ROTATE") && std::strcmp(dev_env("BLMM_ROTATE"), "small") == 0)) {
    const int nkt = (npad + 127) / 128;
    // BLMM_ROTATE_TILE (A/B timing): "64": 128 x 64 tiles with the B fragments prefetched; "128p": 128 x 128 tiles, 32-byte
    // pieces, the next trip's B fragments prefetched
    const char* rv = dev_env("BLMM_ROTATE_TILE");
    const bool wide = !(rv && std::strcmp(rv, "64") == 0);     // "128": the plain 128 x 128 form at every n
    // default: the prefetching form from n = 900 (n = 1000, p = 1e5: 3.75 against 3.87 ms, 53.4 TF; n = 700: 1.32 against 1.31;
    // n = 500: 0.78 against 0.75); the choice depends on n only, so a column's bits do not depend on the call's width
    const bool pref4 = rv ? std::strcmp(rv, "128p") == 0 : (n >= 900);
    const int64_t nct = (ncols_pad + (wide ? 127 : 63)) / (wide ? 128 : 64);
    const int64_t nblk = ((nct + 7) / 8) * 8 * nkt;           // every XCD walks ceil(nct / 8) column tiles
    if (nblk <= 0x7fffffffLL) {
      if (wide && pref4) hipLaunchKernelGGL((k_rotate_big<4, true, 4>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, Rp, ldr, n, npad, In, ncols,
                                            Out, ldo, ncols_pad, nkt, nct);
      else if (wide) hipLaunchKernelGGL((k_rotate_big<4, false>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, Rp, ldr, n, npad, In, ncols, Out, ldo,
                                   ncols_pad, nkt, nct);
      else hipLaunchKernelGGL((k_rotate_big<2, true>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, Rp, ldr, n, npad, In, ncols, Out, ldo,
                         ncols_pad, nkt, nct);
      KCHECK();
      return BLMM_OK;
    }
  }
  constexpr int MBLK = 5;
  const int nrb = (npad + 15) / 16;
  dim3 grid((unsigned)((ncols_pad + 63) / 64), (unsigned)((nrb + MBLK - 1) / MBLK));
  hipLaunchKernelGGL(k_rotate<MBLK>, grid, dim3(256), 0, ctx->stream, Rp, ldr, n, npad, In, ncols, Out, ldo, ncols_pad);
  KCHECK();
  return BLMM_OK;
}

__global__ void k_untranspose(const double* __restrict__ In, int64_t ld, int n, int64_t ncols, double* __restrict__ Out) {
  __shared__ double tile[32][33];
  const int64_t c0 = (int64_t)blockIdx.x * 32;
  const int r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int row = r0 + r; const int64_t col = c0 + tx;
    tile[r][tx] = (row < n && col < ncols) ? In[(int64_t)row * ld + col] : 0.0;
  }
  __syncthreads();
  for (int cc = ty; cc < 32; cc += 8) {
    const int64_t col = c0 + cc; const int row = r0 + tx;
    if (row < n && col < ncols) Out[col * n + row] = tile[tx][cc];
  }
}

int launch_untranspose(blmm_ctx* ctx, const double* In, int64_t ld, int n, int64_t ncols, double* Out) {
  dim3 grid((unsigned)((ncols + 31) / 32), (unsigned)((n + 31) / 32));
  hipLaunchKernelGGL(k_untranspose, grid, dim3(256), 0, ctx->stream, In, ld, n, ncols, Out);
  KCHECK();
  return BLMM_OK;
}

// ------------------------------------------------------------------------------------------------
// Null model: -ell(h2) for one trait, LPT lanes per trait            (src/wls.jl:27-97, src/lmm.jl:15-33)
//   w_k = 1/(delta lambda_k + 1);  A = Z0'WZ0, v = Z0'Wy, Syy = y'Wy;  rss = Syy - v'A^-1 v
//   sigma2 = (rss + a b)/(n [-c] + b_df);  ell = -1/2 [(n+b) ln sigma2 - sum ln w + (rss + a b)/sigma2]
//   REML: + 1/2 [c ln sigma2 - ln det A]
// ------------------------------------------------------------------------------------------------
struct EllOut { double ell, sigma2, yy; };

template <int C, int LPT>
__device__ __forceinline__ EllOut null_ell(double h2, const double* __restrict__ ycol, int64_t ystride, int sub, int n,
                                           const double* __restrict__ sZ /* [C][n] */, const double* __restrict__ sLam,
                                           double prior_a, double prior_b, int reml, int* nonpos) {
  constexpr int NA = C * (C + 1) / 2;
  const double delta = h2 / (1.0 - h2);
  double A[NA], v[C], syy = 0.0, logsum = 0.0, prod = 1.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
#pragma unroll
  for (int q = 0; q < C; ++q) v[q] = 0.0;
  int cnt = 0, bad = 0;
  for (int k = sub; k < n; k += LPT) {
    const double t = fma(delta, sLam[k], 1.0);
    const double w = 1.0 / t;
    bad |= !(w > 0.0);
    prod *= t;
    if (++cnt == 8) { logsum += log(prod); prod = 1.0; cnt = 0; }
    const double y = ycol[(int64_t)k * ystride];
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
  logsum += log(prod);
  syy = group_sum<LPT>(syy); logsum = group_sum<LPT>(logsum);
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = group_sum<LPT>(A[a]);
#pragma unroll
  for (int q = 0; q < C; ++q) v[q] = group_sum<LPT>(v[q]);
  if (bad && nonpos) *nonpos = 1;
  // Cholesky A = L L', t = L^-1 v
  double L[NA], t[C], logdet = 0.0, tt = 0.0;
#pragma unroll
  for (int q = 0; q < C; ++q) {
#pragma unroll
    for (int r = 0; r <= q; ++r) {
      double s = A[q * (q + 1) / 2 + r];
#pragma unroll
      for (int u = 0; u < r; ++u) s = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], s);
      if (r == q) { L[q * (q + 1) / 2 + q] = sqrt(s); logdet += log(s); }
      else L[q * (q + 1) / 2 + r] = s / L[r * (r + 1) / 2 + r];
    }
    double s = v[q];
#pragma unroll
    for (int u = 0; u < q; ++u) s = fma(-L[q * (q + 1) / 2 + u], t[u], s);
    t[q] = s / L[q * (q + 1) / 2 + q];
    tt = fma(t[q], t[q], tt);
  }
  const double rss = syy - tt;
  const double prior_df = prior_b > 0.0 ? prior_b + 2.0 : prior_b;
  const double num = rss + prior_a * prior_b;
  const double sigma2 = num / ((reml ? (double)(n - C) : (double)n) + prior_df);
  const double ls = log(sigma2);
  double ell = -0.5 * (((double)n + prior_b) * ls + logsum + num / sigma2);  // -sum ln w = +sum ln t
  if (reml) ell += 0.5 * ((double)C * ls - logdet);
  EllOut o; o.ell = ell; o.sigma2 = sigma2; o.yy = rss;
  return o;
}

// ---- register-resident variant -----------------------------------------------------------------------
// LPT lanes share one trait; lane `sub` owns individuals k = sub + LPT*i, i < NK, held in registers
// (lambda, y and the C covariate columns; padding rows have lambda = y = z = 0 and contribute nothing).
// All divisions are rcp + 2 Newton steps and all logarithms the table-driven fast_log (fastmath.h).
// build knobs of k_brent (A/B testing: make EXTRA="-DBRENT_MINW=2 -DBRENT_UNROLL=5")
#ifndef BRENT_MINW
#define BRENT_MINW 3
#endif
#ifndef BRENT_UNROLL2
#define BRENT_UNROLL2 5    // ... in k_brent2 (about one wave per SIMD: latency bound, wants the ILP)
#endif
#ifndef BRENT_MINW2
#define BRENT_MINW2 2
#endif
#ifndef BRENT_UNROLL
#define BRENT_UNROLL 1     // quads of the evaluator loop unrolled together
#endif
constexpr int NULL_NK = 20;

template <int LPT>
__device__ __forceinline__ double lpt_sum(double x) { return group_sum<LPT>(x); }

template <int C, int LPT>
struct NullRegs {
  // Operands of the register-free evaluator, all in LDS (the name is historical: they used to be 60 (1 + ..) VGPRs per
  // lane, which held the kernel at one wave per SIMD):
  //   base[yo + i * 256]                   : y_k of this lane's trait, k = sub + LPT * i   (thread-major: conflict-free)
  //   base[lo + i * (1 + C)] = {lambda_k, z_0k, ..} : trait-independent, one table per lane-group position (16-lane broadcasts)
  const double* base;   // the kernel's dynamic LDS array (kept un-laundered: the compiler must see the LDS address space)
  int yo, lo;           // offsets of yp / lz in it
};
// Stages one trait's y into the workgroup's LDS slab: thread t owns column t of sY[NULL_NK][256].
template <int LPT, int NK = NULL_NK>
__device__ __forceinline__ void stage_null_y(double* sY, const double* __restrict__ ycol, int64_t ystride, int sub, int n, bool valid) {
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const int k = sub + LPT * i;
    sY[i * 256 + threadIdx.x] = (k < n && valid) ? ycol[(int64_t)k * ystride] : 0.0;
  }
}
// fills the LDS table of NullRegs (all threads of the workgroup; the caller synchronises)
template <int C, int LPT, int NK = NULL_NK>
__device__ __forceinline__ void stage_null_lz(double* lzbase, int n, const double* __restrict__ Z0, const double* __restrict__ lamv) {
  for (int e = threadIdx.x; e < LPT * NK; e += blockDim.x) {
    const int sub = e / NK, i = e % NK, k = sub + LPT * i;
    double* d = lzbase + (size_t)e * (1 + C);
    d[0] = (k < n) ? lamv[k] : 0.0;
#pragma unroll
    for (int q = 0; q < C; ++q) d[1 + q] = (k < n) ? Z0[q * n + k] : 0.0;
  }
}

template <int C, int LPT, int UNR = BRENT_UNROLL, int NK = NULL_NK>
__device__ __forceinline__ EllOut null_ell_reg(double h2, const NullRegs<C, LPT>& R, int n, double prior_a, double prior_b,
                                               int reml, const dpair* __restrict__ s_ln, int* nonpos) {
  constexpr int NA = C * (C + 1) / 2;
  const double delta = h2 * fast_rcp(1.0 - h2);
  double A[NA], v[C], syy = 0.0, p1 = 1.0, p2 = 1.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
#pragma unroll
  for (int q = 0; q < C; ++q) v[q] = 0.0;
  int bad = 0;
  static_assert(NK % 4 == 0, "weights are inverted four at a time");
  // w = 1/t four at a time from ONE reciprocal (of the product): 22 instruction slots per four elements instead of 40
  // (v_rcp_f64 is quarter rate); the product also feeds sum ln t = ln(prod t), kept as two partial products that stay
  // far from overflow.  Each w carries ~3 extra roundings (4e-16 relative), below the rounding of the sums it enters.
  // the pointers are laundered once per evaluation: otherwise the loads are loop-invariant for the Brent iteration and
  // hipcc hoists all of them back into registers
  // (the OFFSETS are laundered, not the pointers: a laundered pointer loses its address space and every access
  // becomes a flat_load -- several hundred cycles each on the critical path of a latency-bound kernel)
  int lo = R.lo, yo = R.yo;
  asm volatile("" : "+v"(lo), "+v"(yo));
  const double* lzp = R.base + lo;
  const double* yp = R.base + yo;
#pragma unroll UNR
  for (int i0 = 0; i0 < NK; i0 += 4) {
    double lz[4][1 + C], yv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      yv[j] = yp[(i0 + j) * 256];
      if constexpr (C == 1) {
        const dpair e = *reinterpret_cast<const dpair*>(lzp + (i0 + j) * 2);
        lz[j][0] = e[0]; lz[j][1] = e[1];
      } else {
#pragma unroll
        for (int q = 0; q <= C; ++q) lz[j][q] = lzp[(i0 + j) * (1 + C) + q];
      }
    }
    const double t0 = fma(delta, lz[0][0], 1.0), t1 = fma(delta, lz[1][0], 1.0);
    const double t2 = fma(delta, lz[2][0], 1.0), t3 = fma(delta, lz[3][0], 1.0);
    bad |= !(t0 > 0.0) | !(t1 > 0.0) | !(t2 > 0.0) | !(t3 > 0.0);
    const double ab = t0 * t1, cd = t2 * t3, q4 = ab * cd;
    const double rq = fast_rcp(q4);
    const double rab = rq * cd, rcd = rq * ab;
    const double wv[4] = {rab * t1, rab * t0, rcd * t3, rcd * t2};
    if (i0 < NK / 2) p1 *= q4; else p2 *= q4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double w = wv[j];
      const double wy = w * yv[j];
      syy = fma(wy, yv[j], syy);
#pragma unroll
      for (int q = 0; q < C; ++q) {
        v[q] = fma(wy, lz[j][1 + q], v[q]);
        const double wz = w * lz[j][1 + q];
#pragma unroll
        for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(wz, lz[j][1 + r], A[q * (q + 1) / 2 + r]);
      }
    }
  }
  double logsum = fast_log<false>(p1, s_ln) + fast_log<false>(p2, s_ln);
  syy = lpt_sum<LPT>(syy); logsum = lpt_sum<LPT>(logsum);
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = lpt_sum<LPT>(A[a]);
#pragma unroll
  for (int q = 0; q < C; ++q) v[q] = lpt_sum<LPT>(v[q]);
  if (bad && nonpos) *nonpos = 1;
  double L[NA], t[C], detA = 1.0, tt = 0.0;
#pragma unroll
  for (int q = 0; q < C; ++q) {
#pragma unroll
    for (int r = 0; r <= q; ++r) {
      double s = A[q * (q + 1) / 2 + r];
#pragma unroll
      for (int u = 0; u < r; ++u) s = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], s);
      if (r == q) { detA *= s; L[q * (q + 1) / 2 + q] = nr_rsqrt(s); }   // store 1/L_qq
      else L[q * (q + 1) / 2 + r] = s * L[r * (r + 1) / 2 + r];
    }
    double s = v[q];
#pragma unroll
    for (int u = 0; u < q; ++u) s = fma(-L[q * (q + 1) / 2 + u], t[u], s);
    t[q] = s * L[q * (q + 1) / 2 + q];
    tt = fma(t[q], t[q], tt);
  }
  const double rss = syy - tt;
  const double prior_df = prior_b > 0.0 ? prior_b + 2.0 : prior_b;
  const double num = rss + prior_a * prior_b;
  const double den = (reml ? (double)(n - C) : (double)n) + prior_df;
  const double sigma2 = num * fast_rcp(den);
  const double ls = (sigma2 > 0.0) ? fast_log<false>(sigma2, s_ln) : log(sigma2);
  double ell = -0.5 * (((double)n + prior_b) * ls + logsum + den);  // (rss + a b)/sigma2 = den
  if (reml) ell += 0.5 * ((double)C * ls - ((detA > 0.0) ? fast_log<false>(detA, s_ln) : log(detA)));
  EllOut o; o.ell = ell; o.sigma2 = sigma2; o.yy = rss;
  return o;
}

constexpr int BRENT_LPT = 4;
#ifndef BRENT_PHASE1_IT
// build knob (tools/brent_phase1.sh).  26 (rounds 2-3a) kept every trait that converges in the interior in the first kernel; at 20 the
// slowest of them join the boundary traits in k_brent2 (which runs beside the scan of the first region) and the scan starts 20 us
// earlier: step 1.704-1.715 ms against 1.716-1.755 at 26, 1.708-1.722 at 22, 1.718-1.746 at 18, 1.703-1.725 at 16 (one box, two runs each)
#define BRENT_PHASE1_IT 20
#endif
constexpr int BRENT_PHASE1 = BRENT_PHASE1_IT;   // iterations before the unfinished traits of a workgroup are repacked (k_brent)

// Optim.jl Brent() restated (third-party; see oracle/bulklmm_oracle.py:brent_optim and SURVEY.md A.3) on the
// gridbrent sub-intervals (src/gridbrent.jl:9-24).  Every lane of the wave runs the same number of function
// evaluations (converged lanes keep evaluating at their minimiser and discard the value), so the cross-lane
// reductions inside `f` stay convergent.  Returns the minimiser of the best sub-interval (first wins).
template <typename F>
__device__ __forceinline__ double brent_search(F& f, int nint, bool valid, int* hit_max) {
  const double golden = 0.5 * (3.0 - sqrt(5.0));
  const double rel_tol = 1.4901161193847656e-08, abs_tol = 2.220446049250313e-16;
  double best_x = 0.0, best_f = INFINITY;
  for (int iv = 0; iv < nint; ++iv) {
    // points = range(0, 1, length = nint+1)
    double x_lower = (double)iv / (double)nint, x_upper = (iv + 1 == nint) ? 1.0 : (double)(iv + 1) / (double)nint;
    double new_minimizer = x_lower + golden * (x_upper - x_lower);
    double new_minimum = f(new_minimizer);
    double step = 0.0, old_step = 0.0;
    double old_minimizer = new_minimizer, old_old_minimizer = new_minimizer;
    double old_minimum = new_minimum, old_old_minimum = new_minimum;
    bool done = !valid;
    int it = 0;
    for (; it < 1000; ++it) {
      double p = 0.0, q = 0.0;
      const double x_tol = rel_tol * fabs(new_minimizer) + abs_tol;
      const double x_mid = (x_upper + x_lower) / 2;
      if (fabs(new_minimizer - x_mid) <= 2 * x_tol - (x_upper - x_lower) / 2) done = true;
      if (__all(done)) break;
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
      const double new_f = f(done ? new_minimizer : new_x);
      if (!done) {
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
    }
    if (it >= 1000 && !done) *hit_max = 1;
    if (new_minimum < best_f || iv == 0) { best_f = new_minimum; best_x = new_minimizer; }  // argmin: first wins
  }
  return best_x;
}

// The same iteration as a resumable state machine (one sub-interval): brent_run advances every lane of the wave by the
// same number of iterations (until all lanes are done or `max_it` more iterations), so a trait can be moved to another
// lane group between two calls.
struct BrentState {
  double xl, xu, x, fx, step, old_step, ox, oox, ofx, oofx;
  int done;
};
template <typename F>
__device__ __forceinline__ void brent_init(F& f, BrentState& S, double a, double b, bool valid) {
  const double golden = 0.5 * (3.0 - sqrt(5.0));
  S.xl = a; S.xu = b;
  S.x = a + golden * (b - a);
  S.fx = f(S.x);
  S.step = 0.0; S.old_step = 0.0;
  S.ox = S.x; S.oox = S.x; S.ofx = S.fx; S.oofx = S.fx;
  S.done = !valid;
}
template <typename F>
__device__ __forceinline__ int brent_run(F& f, BrentState& S, int max_it) {
  const double golden = 0.5 * (3.0 - sqrt(5.0));
  const double rel_tol = 1.4901161193847656e-08, abs_tol = 2.220446049250313e-16;
  int it = 0;
  for (; it < max_it; ++it) {
    double p = 0.0, q = 0.0;
    const double x_tol = rel_tol * fabs(S.x) + abs_tol;
    const double x_mid = (S.xu + S.xl) / 2;
    if (fabs(S.x - x_mid) <= 2 * x_tol - (S.xu - S.xl) / 2) S.done = 1;
    if (__all(S.done)) break;
    if (fabs(S.old_step) > x_tol) {
      const double r = (S.x - S.ox) * (S.fx - S.oofx);
      q = (S.x - S.oox) * (S.fx - S.ofx);
      p = (S.x - S.oox) * q - (S.x - S.ox) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p; else q = -q;
    }
    double nstep, nold;
    if (fabs(p) < fabs(q * S.old_step / 2) && p < q * (S.xu - S.x) && p < q * (S.x - S.xl)) {
      nold = S.step;
      nstep = p / q;
      const double x_temp = S.x + nstep;
      if ((x_temp - S.xl) < 2 * x_tol || (S.xu - x_temp) < 2 * x_tol) nstep = (S.x < x_mid) ? x_tol : -x_tol;
    } else {
      nold = (S.x < x_mid) ? S.xu - S.x : S.xl - S.x;
      nstep = golden * nold;
    }
    const double new_x = (fabs(nstep) >= x_tol) ? S.x + nstep : S.x + ((nstep > 0) ? x_tol : -x_tol);
    const double new_f = f(S.done ? S.x : new_x);
    if (!S.done) {
      S.old_step = nold; S.step = nstep;
      if (new_f < S.fx) {
        if (new_x < S.x) S.xu = S.x; else S.xl = S.x;
        S.oox = S.ox; S.oofx = S.ofx;
        S.ox = S.x; S.ofx = S.fx;
        S.x = new_x; S.fx = new_f;
      } else {
        if (new_x < S.x) S.xl = new_x; else S.xu = new_x;
        if (new_f <= S.ofx || S.ox == S.x) {
          S.oox = S.ox; S.oofx = S.ofx;
          S.ox = new_x; S.ofx = new_f;
        } else if (new_f <= S.oofx || S.oox == S.x || S.oox == S.ox) {
          S.oox = new_x; S.oofx = new_f;
        }
      }
    }
  }
  return it;
}

// fitlmm for every trait (src/lmm.jl:56-86): Brent search, then the final wls at the minimiser (:84).
// REG: the register-resident evaluator (n <= LPT * NULL_NK); otherwise operands are re-read every evaluation.
// Continuation area of the two-kernel form: traits unfinished after BRENT_PHASE1 iterations are appended to `list`
// with their Brent state (st[f * m + j], f < 12: the ten doubles of BrentState, the iteration count, the non-positive
// weight flag) and finished by k_brent2 in densely packed lane groups.  list == nullptr: repack inside the workgroup.
struct BrentCont {
  double* st;
  int* list;
  unsigned int* cnt;
  int* fin;            // fin[j] = 1: trait j finished in k_brent (its h2 is final), 0: it is on the list
};

template <int C, int LPT, bool REG>
__global__ void __launch_bounds__(256, BRENT_MINW) k_brent(NullModel nm, const double* __restrict__ Yt, int64_t ldy, int64_t m,
                                               const double* __restrict__ Z0, const double* __restrict__ lam,
                                               const double* __restrict__ logtab, double* __restrict__ h2out,
                                               double* __restrict__ s2out, double* __restrict__ ellout, int64_t* stat,
                                               BrentCont cont) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ dpair s_ln[BLMM_LOG_TABLE_N];
  const int n = nm.n;
  double* sLam = sh;       // n      (generic evaluator only)
  double* sZ = sh + n;     // C*n
  if (!REG) {
    for (int e = threadIdx.x; e < n; e += blockDim.x) sLam[e] = lam[e];
    for (int e = threadIdx.x; e < n * C; e += blockDim.x) sZ[e] = Z0[e];
  } else {
    stage_null_lz<C, LPT>(sh, n, Z0, lam);      // sh: LPT * NULL_NK * (1 + C) doubles, then sY[NULL_NK][256]
  }
  stage_log_table<false>(s_ln, logtab);
  __syncthreads();
  const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPT;
  const int sub = threadIdx.x % LPT;
  const bool valid = j < m;
  const double* ycol = Yt + (valid ? j : 0);
  int nonpos = 0, hit_max = 0;
  const int nint = nm.optim_interval < 1 ? 1 : nm.optim_interval;
  double best_x;
  EllOut fin;
  double* sY = sh + LPT * NULL_NK * (1 + C);
  if constexpr (REG) {
    stage_null_y<LPT>(sY, ycol, ldy, sub, n, valid);   // own column only: no barrier needed before the lane reads it back
    __syncthreads();                                   // (the repacking path reads other lanes' columns later)
  }
  if constexpr (REG && (64 / LPT) > 1) {
    if (nint == 1) {
      // Two phases with a repack in between.  On eQTL-like data the evaluation count is bimodal: about half of the
      // traits converge in 12-25 evaluations, the other half (minimum at the h2 = 0 boundary, where x_tol shrinks
      // with x) needs 72-77; a wave runs until its slowest trait is done, so with 16 traits per wave nearly every wave
      // ran ~76.  After BRENT_PHASE1 iterations the unfinished traits of the workgroup are packed into as few waves
      // as possible and the other waves retire.  Every trait still sees exactly the Optim.jl iteration sequence.
      constexpr int TPW = 256 / LPT;            // traits per workgroup
      constexpr int TPV = 64 / LPT;             // traits per wave
      __shared__ double s_bst[10][TPW];
      __shared__ int s_bit[TPW], s_blist[TPW], s_bnp[TPW], s_bcnt;
      NullRegs<C, LPT> R;
      R.base = sh; R.lo = sub * NULL_NK * (1 + C); R.yo = LPT * NULL_NK * (1 + C) + threadIdx.x;
      auto f = [&](double h2) { return -null_ell_reg<C, LPT>(h2, R, n, nm.prior_a, nm.prior_b, nm.reml, s_ln, &nonpos).ell; };
      BrentState S;
      brent_init(f, S, 0.0, 1.0, valid);
      const int it1 = brent_run(f, S, BRENT_PHASE1);
      fin = null_ell_reg<C, LPT>(S.x, R, n, nm.prior_a, nm.prior_b, nm.reml, s_ln, &nonpos);
      const int ts = threadIdx.x / LPT;
      wave_count(&stat[ST_H2_BOUNDARY], sub == 0 && valid && S.done && h2_on_boundary(S.x));
      if (sub == 0) {
        if (valid && cont.fin) cont.fin[j] = S.done ? 1 : 0;
        if (valid && S.done) {
          h2out[j] = S.x;
          if (s2out) s2out[j] = fin.sigma2;
          if (ellout) ellout[j] = fin.ell;
          if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
        }
        s_bst[0][ts] = S.xl; s_bst[1][ts] = S.xu; s_bst[2][ts] = S.x; s_bst[3][ts] = S.fx; s_bst[4][ts] = S.step;
        s_bst[5][ts] = S.old_step; s_bst[6][ts] = S.ox; s_bst[7][ts] = S.oox; s_bst[8][ts] = S.ofx; s_bst[9][ts] = S.oofx;
        s_bit[ts] = S.done ? -1 : it1;
        s_bnp[ts] = nonpos;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        int cnt = 0;
        for (int e = 0; e < TPW; ++e) if (s_bit[e] >= 0) s_blist[cnt++] = e;
        s_bcnt = cnt;
      }
      __syncthreads();
      const int U = s_bcnt;
      if (cont.list) {
        // two-kernel form: hand the unfinished traits over (one atomic per workgroup; the order of the list depends on
        // the scheduling, the per-trait results do not)
        __shared__ unsigned int s_bbase;
        if (threadIdx.x == 0 && U > 0) s_bbase = atomicAdd(cont.cnt, (unsigned int)U);
        __syncthreads();
        for (int e = threadIdx.x; e < U; e += blockDim.x) cont.list[s_bbase + e] = (int)(blockIdx.x * TPW + s_blist[e]);
        for (int e = threadIdx.x; e < U * 12; e += blockDim.x) {
          const int fi = e / U, src = s_blist[e % U];
          const int64_t jt = (int64_t)blockIdx.x * TPW + src;
          cont.st[(int64_t)fi * m + jt] = (fi < 10) ? s_bst[fi][src] : (fi == 10 ? (double)s_bit[src] : (double)s_bnp[src]);
        }
        return;
      }
      // the surviving waves rotate with the workgroup index: the four waves of a workgroup sit on the four SIMDs of its
      // CU, and always keeping waves 0.. would leave all of phase 2 on SIMDs 0 and 1
      const int wave = ((threadIdx.x >> 6) + 4 - (int)(blockIdx.x & 3)) & 3;
      if (wave * TPV >= U) return;                // wave-uniform: this wave has nothing left
      const int qslot = wave * TPV + (threadIdx.x & 63) / LPT;
      const bool valid2 = qslot < U;
      const int src = s_blist[valid2 ? qslot : 0];
      const int64_t j2 = (int64_t)blockIdx.x * TPW + src;
      R.yo = LPT * NULL_NK * (1 + C) + src * LPT + sub;   // the trait's y is already in the workgroup's slab
      S.xl = s_bst[0][src]; S.xu = s_bst[1][src]; S.x = s_bst[2][src]; S.fx = s_bst[3][src]; S.step = s_bst[4][src];
      S.old_step = s_bst[5][src]; S.ox = s_bst[6][src]; S.oox = s_bst[7][src]; S.ofx = s_bst[8][src]; S.oofx = s_bst[9][src];
      S.done = !valid2;
      nonpos = s_bnp[src];
      const int it0 = s_bit[src];
      const int it2 = brent_run(f, S, 1000 - it0);
      fin = null_ell_reg<C, LPT>(S.x, R, n, nm.prior_a, nm.prior_b, nm.reml, s_ln, &nonpos);
      wave_count(&stat[ST_H2_BOUNDARY], valid2 && sub == 0 && h2_on_boundary(S.x));
      if (valid2 && sub == 0) {
        h2out[j2] = S.x;
        if (s2out) s2out[j2] = fin.sigma2;
        if (ellout) ellout[j2] = fin.ell;
        if (it0 + it2 >= 1000 && !S.done) atomicAdd((unsigned long long*)&stat[ST_BRENT_MAXIT], 1ull);
        if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
      }
      return;
    }
  }
  if constexpr (REG) {
    NullRegs<C, LPT> R;
    R.base = sh; R.lo = sub * NULL_NK * (1 + C); R.yo = LPT * NULL_NK * (1 + C) + threadIdx.x;
    auto f = [&](double h2) { return -null_ell_reg<C, LPT>(h2, R, n, nm.prior_a, nm.prior_b, nm.reml, s_ln, &nonpos).ell; };
    best_x = brent_search(f, nint, valid, &hit_max);
    fin = null_ell_reg<C, LPT>(best_x, R, n, nm.prior_a, nm.prior_b, nm.reml, s_ln, &nonpos);
  } else {
    auto f = [&](double h2) { return -null_ell<C, LPT>(h2, ycol, ldy, sub, n, sZ, sLam, nm.prior_a, nm.prior_b, nm.reml, &nonpos).ell; };
    best_x = brent_search(f, nint, valid, &hit_max);
    fin = null_ell<C, LPT>(best_x, ycol, ldy, sub, n, sZ, sLam, nm.prior_a, nm.prior_b, nm.reml, &nonpos);
  }
  wave_count(&stat[ST_H2_BOUNDARY], valid && sub == 0 && h2_on_boundary(best_x));
  if (valid && sub == 0) {
    h2out[j] = best_x;
    if (s2out) s2out[j] = fin.sigma2;
    if (ellout) ellout[j] = fin.ell;
    if (hit_max) atomicAdd((unsigned long long*)&stat[ST_BRENT_MAXIT], 1ull);
    if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
  }
}

// Second kernel of the two-kernel form: finishes the traits k_brent handed over, LPT lanes per list entry, NK individuals
// per lane (the state is per trait and y is re-staged, so the split may differ from k_brent's: BLMM_BRENT2_LPT).
template <int C, int LPT, int NK = NULL_NK>
__global__ void __launch_bounds__(256, BRENT_MINW2) k_brent2(NullModel nm, const double* __restrict__ Yt, int64_t ldy, int64_t m,
                                                const double* __restrict__ Z0, const double* __restrict__ lam,
                                                const double* __restrict__ logtab, double* __restrict__ h2out,
                                                double* __restrict__ s2out, double* __restrict__ ellout, int64_t* stat,
                                                BrentCont cont) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ dpair s_ln[BLMM_LOG_TABLE_N];
  constexpr int TPW = 256 / LPT;
  const unsigned int U = *cont.cnt;
  if ((unsigned int)blockIdx.x * TPW >= U) return;          // workgroup-uniform
  const int n = nm.n;
  stage_null_lz<C, LPT, NK>(sh, n, Z0, lam);
  stage_log_table<false>(s_ln, logtab);
  double* sY = sh + LPT * NK * (1 + C);
  const unsigned int q = (unsigned int)blockIdx.x * TPW + threadIdx.x / LPT;
  const int sub = threadIdx.x % LPT;
  const bool valid = q < U;
  const int64_t j = cont.list[valid ? q : (unsigned int)blockIdx.x * TPW];
  stage_null_y<LPT, NK>(sY, Yt + j, ldy, sub, n, true);
  __syncthreads();
  int nonpos = (int)cont.st[(int64_t)11 * m + j];
  NullRegs<C, LPT> R;
  R.base = sh; R.lo = sub * NK * (1 + C); R.yo = LPT * NK * (1 + C) + threadIdx.x;
  auto f = [&](double h2) { return -null_ell_reg<C, LPT, (NK / 4 < BRENT_UNROLL2 ? NK / 4 : BRENT_UNROLL2), NK>(h2, R, n, nm.prior_a, nm.prior_b, nm.reml, s_ln, &nonpos).ell; };
  BrentState S;
  S.xl = cont.st[j]; S.xu = cont.st[m + j]; S.x = cont.st[2 * m + j]; S.fx = cont.st[3 * m + j];
  S.step = cont.st[4 * m + j]; S.old_step = cont.st[5 * m + j]; S.ox = cont.st[6 * m + j]; S.oox = cont.st[7 * m + j];
  S.ofx = cont.st[8 * m + j]; S.oofx = cont.st[9 * m + j];
  S.done = !valid;
  const int it0 = (int)cont.st[(int64_t)10 * m + j];
  const int it2 = brent_run(f, S, 1000 - it0);
  const EllOut fin = null_ell_reg<C, LPT, (NK / 4 < BRENT_UNROLL2 ? NK / 4 : BRENT_UNROLL2), NK>(S.x, R, n, nm.prior_a, nm.prior_b, nm.reml, s_ln, &nonpos);
  wave_count(&stat[ST_H2_BOUNDARY], valid && sub == 0 && h2_on_boundary(S.x));
  if (valid && sub == 0) {
    h2out[j] = S.x;
    if (s2out) s2out[j] = fin.sigma2;
    if (ellout) ellout[j] = fin.ell;
    if (it0 + it2 >= 1000 && !S.done) atomicAdd((unsigned long long*)&stat[ST_BRENT_MAXIT], 1ull);
    if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
  }
}

// phase 0: the whole search; 1: k_brent only (sp->active tells whether k_brent2 is pending; if not, the search is complete);
// 2: k_brent2 (after a phase 1 with sp->active, same arguments)
template <int C, int LPT, bool REG>
static int launch_brent_t(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                          const double* lam, double* h2, double* sigma2, double* ell, int64_t* stat, int phase, BrentSplit* sp) {
  const int64_t threads = m * LPT;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  const size_t lds = REG ? sizeof(double) * ((size_t)LPT * NULL_NK * (1 + C) + (size_t)NULL_NK * 256) : sizeof(double) * (size_t)nm.n * (1 + C);
  if (lds > 48 * 1024)
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_brent<C, LPT, REG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  BrentCont cont{nullptr, nullptr, nullptr, nullptr};
  static const char* two_env = dev_env("BLMM_BRENT_TWO");   // "0": keep the single-kernel form (A/B testing)
  const bool two = REG && (64 / LPT) > 1 && nm.optim_interval <= 1 && m >= 1024 && !(two_env && two_env[0] == '0');
  if (two) {
    int rc = ensure(ctx, ctx->brSt, sizeof(double) * (size_t)12 * m);
    if (!rc) rc = ensure(ctx, ctx->brList, sizeof(int) * 2 * (size_t)m + 16);
    if (rc) return rc;
    cont.st = ptr<double>(ctx->brSt);
    // the hand-over counter lives in the status block (stat[21..22]), which the memset at the head of the call has zeroed: the
    // first search of a call needs no fill of its own (a 5 us runtime kernel with a ~10 us gap in front of k_brent, on the
    // critical path of the step); a second search inside the same call zeroes it again
    cont.cnt = reinterpret_cast<unsigned int*>(stat + ST_BRENT_CNT);
    cont.list = ptr<int>(ctx->brList) + 4;
    cont.fin = cont.list + m;
    if (phase != 2) {
      if (ctx->brent_cnt_used) BLMM_HIP(hipMemsetAsync(cont.cnt, 0, 16, ctx->stream));
      ctx->brent_cnt_used = true;
    }
  }
  if (sp) { sp->active = two && phase == 1; sp->fin = cont.fin; sp->list = cont.list; sp->cnt = cont.cnt; }
  if (phase != 2) {
    hipLaunchKernelGGL((k_brent<C, LPT, REG>), dim3(blocks), dim3(256), lds, ctx->stream, nm, Yt, ldy, m, Z0, lam,
                       ptr<double>(ctx->logtab), h2, sigma2, ell, stat, cont);
    KCHECK();
  }
  if (phase == 1 && two) return BLMM_OK;
  if (phase == 2 && !two) return BLMM_OK;
  if constexpr (REG) {
    if (two) {
      // A/B testing: 8 / 16 spread a trait of the second kernel over more lanes than k_brent's 4 (measured at BXD size:
      // h2 phase 0.250 ms at 4, 0.259 at 8, 0.372 at 16 -- the wider cross-lane sums cost more than the shorter loop saves)
      static const int lpt2_env = dev_env("BLMM_BRENT2_LPT") ? atoi(dev_env("BLMM_BRENT2_LPT")) : 0;
      auto launch2 = [&](auto kern, int lpt2, int nk2) -> int {
        const size_t lds2 = sizeof(double) * ((size_t)lpt2 * nk2 * (1 + C) + (size_t)nk2 * 256);
        if (lds2 > 48 * 1024)
          BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        const unsigned blocks2 = (unsigned)((m * lpt2 + 255) / 256);
        hipLaunchKernelGGL(kern, dim3(blocks2), dim3(256), lds2, ctx->stream, nm, Yt, ldy, m, Z0, lam, ptr<double>(ctx->logtab), h2,
                           sigma2, ell, stat, cont);
        KCHECK();
        return BLMM_OK;
      };
      if constexpr (LPT == 4) {
        if (lpt2_env == 16 && nm.n <= 16 * 8) return launch2(&k_brent2<C, 16, 8>, 16, 8);
        if (lpt2_env == 8 && nm.n <= 8 * 12) return launch2(&k_brent2<C, 8, 12>, 8, 12);
      }
      return launch2(&k_brent2<C, LPT, NULL_NK>, LPT, NULL_NK);
    }
  }
  return BLMM_OK;
}

template <int C>
static int launch_brent_c(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                          const double* lam, double* h2, double* sigma2, double* ell, int64_t* stat, int phase, BrentSplit* sp) {
  const int n = nm.n;
  static const int lpt_env = dev_env("BLMM_BRENT_LPT") ? atoi(dev_env("BLMM_BRENT_LPT")) : 0;
  if (lpt_env == 8 && n <= 8 * NULL_NK) return launch_brent_t<C, 8, true>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
  if (lpt_env == 16 && n <= 16 * NULL_NK) return launch_brent_t<C, 16, true>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
  if (n <= 4 * NULL_NK) return launch_brent_t<C, 4, true>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
  if (n <= 8 * NULL_NK) return launch_brent_t<C, 8, true>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
  if (n <= 16 * NULL_NK) return launch_brent_t<C, 16, true>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
  if (n <= 32 * NULL_NK) return launch_brent_t<C, 32, true>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
  if (n <= 64 * NULL_NK) return launch_brent_t<C, 64, true>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
  return launch_brent_t<C, BRENT_LPT, false>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
}

int launch_brent(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                 const double* lam, double* h2, double* sigma2, double* ell, int64_t* stat, int phase, BrentSplit* sp) {
  if (nm.c > CTPL && nm.c <= CMAX) {       // run-time covariate count: one kernel, never split
    if (sp) sp->active = false;
    return phase == 2 ? BLMM_OK : launch_dyn_brent(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat);
  }
  switch (nm.c) {
    case 1: return launch_brent_c<1>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
    case 2: return launch_brent_c<2>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
    case 3: return launch_brent_c<3>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
    case 4: return launch_brent_c<4>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp);
    // beyond CFAST: the generic evaluator (operands re-read per evaluation, libm log / IEEE division), one instantiation each
#define BG(C) return launch_brent_t<C, 16, false>(ctx, nm, Yt, ldy, m, Z0, lam, h2, sigma2, ell, stat, phase, sp)
    case 5: BG(5);
    case 6: BG(6);
    case 7: BG(7);
    case 8: BG(8);
#undef BG
  }
  return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
}

// ------------------------------------------------------------------------------------------------
// scan_alt (src/scan.jl:397-453): one trait, the variance components re-estimated for every marker.
//   for marker i:  X = [Z0 x_i];  out11 = fitlmm(y0, X, lambda, prior; reml, optim_interval)          (:425-428)
//                  lod_i = (wls(y0, X, sqrt.(w(h2_i)), prior).ell - wls(y0, Z0, sqrt.(w(h2_null)), prior).ell) / ln 10   (:431-437)
// The two closing wls calls hand wls the SQUARE ROOTS of the weights as its weights and always use ML (no `reml` keyword):
// SQW = true restates exactly that; `true_w` (BLMM_COMPAT_ALT_TRUE_WEIGHTS) evaluates both at makeweights(h2) instead.
// ell of y on the design [Z0 (C columns, LDS)  x (X = 1: one strided column from memory)], LPT lanes per marker; y, Z0, lambda
// are shared by all markers and sit in LDS.
// ------------------------------------------------------------------------------------------------
template <int C, int X, int LPT, bool SQW>
__device__ __forceinline__ EllOut alt_ell(double h2, const double* __restrict__ xcol, int64_t xstride, int sub, int n,
                                          const double* __restrict__ sY, const double* __restrict__ sZ,
                                          const double* __restrict__ sLam, double prior_a, double prior_b, int reml, int* nonpos) {
  constexpr int D = C + X, NA = D * (D + 1) / 2;
  const double delta = h2 / (1.0 - h2);
  double A[NA], v[D], syy = 0.0, logsum = 0.0, prod = 1.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
#pragma unroll
  for (int q = 0; q < D; ++q) v[q] = 0.0;
  int cnt = 0, bad = 0;
  for (int k = sub; k < n; k += LPT) {
    const double t = fma(delta, sLam[k], 1.0);
    double w = 1.0 / t;
    bad |= !(w > 0.0);
    if (SQW) w = sqrt(w);
    prod *= t;
    if (++cnt == 8) { logsum += log(prod); prod = 1.0; cnt = 0; }
    const double y = sY[k];
    const double wy = w * y;
    syy = fma(wy, y, syy);
    double col[D];
#pragma unroll
    for (int q = 0; q < C; ++q) col[q] = sZ[q * n + k];
    if (X) col[D - 1] = xcol[(int64_t)k * xstride];
#pragma unroll
    for (int q = 0; q < D; ++q) {
      v[q] = fma(wy, col[q], v[q]);
      const double wz = w * col[q];
#pragma unroll
      for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(wz, col[r], A[q * (q + 1) / 2 + r]);
    }
  }
  logsum += log(prod);
  if (SQW) logsum *= 0.5;                       // sum ln t of the weights actually used, sqrt(w) = t^(-1/2)
  syy = group_sum<LPT>(syy); logsum = group_sum<LPT>(logsum);
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = group_sum<LPT>(A[a]);
#pragma unroll
  for (int q = 0; q < D; ++q) v[q] = group_sum<LPT>(v[q]);
  if (bad && nonpos) *nonpos = 1;
  double L[NA], t[D], logdet = 0.0, tt = 0.0;
#pragma unroll
  for (int q = 0; q < D; ++q) {
#pragma unroll
    for (int r = 0; r <= q; ++r) {
      double s = A[q * (q + 1) / 2 + r];
#pragma unroll
      for (int u = 0; u < r; ++u) s = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], s);
      if (r == q) { L[q * (q + 1) / 2 + q] = sqrt(s); logdet += log(s); }
      else L[q * (q + 1) / 2 + r] = s / L[r * (r + 1) / 2 + r];
    }
    double s = v[q];
#pragma unroll
    for (int u = 0; u < q; ++u) s = fma(-L[q * (q + 1) / 2 + u], t[u], s);
    t[q] = s / L[q * (q + 1) / 2 + q];
    tt = fma(t[q], t[q], tt);
  }
  const double rss = syy - tt;
  const double prior_df = prior_b > 0.0 ? prior_b + 2.0 : prior_b;
  const double num = rss + prior_a * prior_b;
  const double sigma2 = num / ((reml ? (double)(n - D) : (double)n) + prior_df);
  const double ls = log(sigma2);
  double ell = -0.5 * (((double)n + prior_b) * ls + logsum + num / sigma2);
  if (reml) ell += 0.5 * ((double)D * ls - logdet);
  EllOut o; o.ell = ell; o.sigma2 = sigma2; o.yy = rss;
  return o;
}

template <int C, int LPT>
__global__ void __launch_bounds__(256) k_alt_brent(NullModel nm, const double* __restrict__ Yt, int64_t ldy,
                                                   const double* __restrict__ Xt, int64_t ldx, int64_t p, const double* __restrict__ Z0,
                                                   const double* __restrict__ lam, const double* __restrict__ h2null,
                                                   int true_w, double* __restrict__ lod, double* __restrict__ h2each,
                                                   int64_t* stat, int64_t trait0, int64_t ldL, int64_t ldH) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int n = nm.n;
  // blockIdx.y: the trait (bulk form, blmm_bulkscan_alt_exact: trait0 + blockIdx.y; outputs are columns of p x m matrices)
  const int64_t tr = trait0 + blockIdx.y;
  lod += tr * ldL; h2each += tr * ldH;
  double* sLam = sh;            // n
  double* sY = sh + n;          // n   (the trait: column tr of Yt)
  double* sZ = sh + 2 * n;      // C*n
  for (int e = threadIdx.x; e < n; e += blockDim.x) { sLam[e] = lam[e]; sY[e] = Yt[(int64_t)e * ldy + tr]; }
  for (int e = threadIdx.x; e < n * C; e += blockDim.x) sZ[e] = Z0[e];
  __syncthreads();
  const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPT;
  const int sub = threadIdx.x % LPT;
  const bool valid = j < p;
  const double* xcol = Xt + (valid ? j : 0);
  int nonpos = 0, hit_max = 0;
  const int nint = nm.optim_interval < 1 ? 1 : nm.optim_interval;
  auto f = [&](double h2) { return -alt_ell<C, 1, LPT, false>(h2, xcol, ldx, sub, n, sY, sZ, sLam, nm.prior_a, nm.prior_b, nm.reml, &nonpos).ell; };
  const double hx = brent_search(f, nint, valid, &hit_max);
  const double h0 = h2null[tr];
  double e1, e0;
  if (true_w) {
    e1 = alt_ell<C, 1, LPT, false>(hx, xcol, ldx, sub, n, sY, sZ, sLam, nm.prior_a, nm.prior_b, 0, nullptr).ell;
    e0 = alt_ell<C, 0, LPT, false>(h0, xcol, ldx, sub, n, sY, sZ, sLam, nm.prior_a, nm.prior_b, 0, nullptr).ell;
  } else {
    e1 = alt_ell<C, 1, LPT, true>(hx, xcol, ldx, sub, n, sY, sZ, sLam, nm.prior_a, nm.prior_b, 0, nullptr).ell;
    e0 = alt_ell<C, 0, LPT, true>(h0, xcol, ldx, sub, n, sY, sZ, sLam, nm.prior_a, nm.prior_b, 0, nullptr).ell;
  }
  if (valid && sub == 0) {
    lod[j] = (e1 - e0) / log(10.0);
    h2each[j] = hx;
    if (hit_max) atomicAdd((unsigned long long*)&stat[ST_BRENT_MAXIT], 1ull);
    if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
  }
}

template <int C, int LPT>
static int launch_alt_brent_t(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Xt, int64_t ldx, int64_t p,
                              const double* Z0, const double* lam, const double* h2null, int true_w, double* lod,
                              double* h2each, int64_t* stat, int64_t m, int64_t ldL, int64_t ldH) {
  const size_t lds = sizeof(double) * (size_t)nm.n * (2 + C);
  if (lds > 160 * 1024 - 512) return fail(ctx, BLMM_ERR_UNSUPPORTED, "scan_alt: n too large for the LDS-resident trait");
  if (lds > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_alt_brent<C, LPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t threads = p * LPT;
  for (int64_t t0 = 0; t0 < m; t0 += 65535) {       // gridDim.y <= 65535
    const int64_t mt = (m - t0 < 65535) ? m - t0 : 65535;
    hipLaunchKernelGGL((k_alt_brent<C, LPT>), dim3((unsigned)((threads + 255) / 256), (unsigned)mt), dim3(256), lds, ctx->stream, nm, Yt, ldy, Xt, ldx, p,
                       Z0, lam, h2null, true_w, lod, h2each, stat, t0, ldL, ldH);
    KCHECK();
  }
  return BLMM_OK;
}

int launch_alt_brent(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Xt, int64_t ldx, int64_t p,
                     const double* Z0, const double* lam, const double* h2null, int true_w, double* lod, double* h2each,
                     int64_t* stat, int64_t m, int64_t ldL, int64_t ldH) {
  if (p < 1 || m < 1) return BLMM_OK;
#define AB(C) (nm.n <= 160 ? launch_alt_brent_t<C, 4>(ctx, nm, Yt, ldy, Xt, ldx, p, Z0, lam, h2null, true_w, lod, h2each, stat, m, ldL, ldH) \
                           : launch_alt_brent_t<C, 16>(ctx, nm, Yt, ldy, Xt, ldx, p, Z0, lam, h2null, true_w, lod, h2each, stat, m, ldL, ldH))
  switch (nm.c) {
    case 1: return AB(1);
    case 2: return AB(2);
    case 3: return AB(3);
    case 4: return AB(4);
#define AG(C) return launch_alt_brent_t<C, 16>(ctx, nm, Yt, ldy, Xt, ldx, p, Z0, lam, h2null, true_w, lod, h2each, stat, m, ldL, ldH)
    case 5: AG(5);
    case 6: AG(6);
    case 7: AG(7);
    case 8: AG(8);
#undef AG
  }
#undef AB
  // beyond the instantiated counts: the run-time-c form (kernels_dyn.hip: one wave per marker, factorisations in LDS)
  return launch_dyn_alt_brent(ctx, nm, Yt, ldy, Xt, ldx, p, Z0, lam, h2null, true_w, lod, h2each, stat, m, ldL, ldH);
}

// Ell[g, j] = wls_multivar(Y0, Z0, makeweights(grid[g]), prior).Ell  (src/bulkscan_helpers.jl:267-269),
// per-trait first arg-max (find_optim_h2, src/bulkscan_helpers.jl:204-211).
template <int C>
__global__ void __launch_bounds__(256) k_loglik_grid(NullModel nm, const double* __restrict__ Yt, int64_t ldy, int64_t m,
                                                     const double* __restrict__ Z0, const double* __restrict__ lam,
                                                     const double* __restrict__ grid, int ngrid,
                                                     double* __restrict__ EllTab, int* __restrict__ h2idx,
                                                     double* __restrict__ h2out, int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int n = nm.n;
  double* sLam = sh;
  double* sZ = sh + n;
  for (int e = threadIdx.x; e < n; e += blockDim.x) sLam[e] = lam[e];
  for (int e = threadIdx.x; e < n * C; e += blockDim.x) sZ[e] = Z0[e];
  __syncthreads();
  constexpr int LPT = BRENT_LPT;
  const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPT;
  const int sub = threadIdx.x % LPT;
  const bool valid = j < m;
  const double* ycol = Yt + (valid ? j : 0);
  int nonpos = 0, best = 0;
  double bestv = -INFINITY;
  for (int g = 0; g < ngrid; ++g) {
    const EllOut e = null_ell<C, LPT>(grid[g], ycol, ldy, sub, n, sZ, sLam, nm.prior_a, nm.prior_b, nm.reml, &nonpos);
    if (valid && sub == 0 && EllTab) EllTab[j * (int64_t)ngrid + g] = e.ell;
    if (g == 0 || e.ell > bestv) { bestv = e.ell; best = g; }
  }
  if (valid && sub == 0) {
    if (h2idx) h2idx[j] = best;
    if (h2out) h2out[j] = grid[best];
    if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
  }
}

// The same with the operands staged in LDS once per trait and the batched-reciprocal evaluator of k_brent (the generic
// evaluator above re-reads y from global memory for every grid point: one exposed round trip per element and point).
template <int C, int LPT>
__global__ void __launch_bounds__(256, BRENT_MINW) k_loglik_grid_lds(NullModel nm, const double* __restrict__ Yt, int64_t ldy,
                                                         int64_t m, const double* __restrict__ Z0,
                                                         const double* __restrict__ lam, const double* __restrict__ logtab,
                                                         const double* __restrict__ grid, int ngrid,
                                                         double* __restrict__ EllTab, int* __restrict__ h2idx,
                                                         double* __restrict__ h2out, int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ dpair s_ln[BLMM_LOG_TABLE_N];
  const int n = nm.n;
  stage_null_lz<C, LPT>(sh, n, Z0, lam);
  stage_log_table<false>(s_ln, logtab);
  double* sY = sh + LPT * NULL_NK * (1 + C);
  const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPT;
  const int sub = threadIdx.x % LPT;
  const bool valid = j < m;
  stage_null_y<LPT>(sY, Yt + (valid ? j : 0), ldy, sub, n, valid);
  __syncthreads();
  NullRegs<C, LPT> R;
  R.base = sh; R.lo = sub * NULL_NK * (1 + C); R.yo = LPT * NULL_NK * (1 + C) + threadIdx.x;
  int nonpos = 0, best = 0;
  double bestv = -INFINITY;
  for (int g = 0; g < ngrid; ++g) {
    const EllOut e = null_ell_reg<C, LPT>(grid[g], R, n, nm.prior_a, nm.prior_b, nm.reml, s_ln, &nonpos);
    if (valid && sub == 0 && EllTab) EllTab[j * (int64_t)ngrid + g] = e.ell;
    if (g == 0 || e.ell > bestv) { bestv = e.ell; best = g; }
  }
  if (valid && sub == 0) {
    if (h2idx) h2idx[j] = best;
    if (h2out) h2out[j] = grid[best];
    if (nonpos) atomicAdd((unsigned long long*)&stat[ST_NONPOS_W], 1ull);
  }
}

template <int C, int LPT>
static int launch_loglik_grid_lds(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                                  const double* lam, const double* grid_dev, int ngrid, double* EllTab, int* h2idx,
                                  double* h2, int64_t* stat) {
  const int64_t threads = m * LPT;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  const size_t lds = sizeof(double) * ((size_t)LPT * NULL_NK * (1 + C) + (size_t)NULL_NK * 256);
  if (lds > 48 * 1024)
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loglik_grid_lds<C, LPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_loglik_grid_lds<C, LPT>), dim3(blocks), dim3(256), lds, ctx->stream, nm, Yt, ldy, m, Z0, lam,
                     ptr<double>(ctx->logtab), grid_dev, ngrid, EllTab, h2idx, h2, stat);
  KCHECK();
  return BLMM_OK;
}
template <int C>
static int launch_loglik_grid_c(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                                const double* lam, const double* grid_dev, int ngrid, double* EllTab, int* h2idx, double* h2,
                                int64_t* stat, bool* done) {
  const int n = nm.n;
  *done = true;
  if (n <= 4 * NULL_NK) return launch_loglik_grid_lds<C, 4>(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat);
  if (n <= 8 * NULL_NK) return launch_loglik_grid_lds<C, 8>(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat);
  if (n <= 16 * NULL_NK) return launch_loglik_grid_lds<C, 16>(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat);
  if (n <= 32 * NULL_NK) return launch_loglik_grid_lds<C, 32>(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat);
  if (n <= 64 * NULL_NK) return launch_loglik_grid_lds<C, 64>(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat);
  *done = false;
  return BLMM_OK;
}

int launch_loglik_grid(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                       const double* lam, const double* grid_dev, int ngrid, double* EllTab, int* h2idx, double* h2,
                       int64_t* stat) {
  if (nm.c > CTPL && nm.c <= CMAX) return launch_dyn_loglik_grid(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat);
  static const char* gen_env = dev_env("BLMM_LOGLIK_GENERIC");   // "1": the generic evaluator for every n (A/B testing)
  if (!(gen_env && gen_env[0] == '1')) {
    bool done = false;
    int rc = BLMM_OK;
    switch (nm.c) {
      case 1: rc = launch_loglik_grid_c<1>(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat, &done); break;
      case 2: rc = launch_loglik_grid_c<2>(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat, &done); break;
      case 3: rc = launch_loglik_grid_c<3>(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat, &done); break;
      case 4: rc = launch_loglik_grid_c<4>(ctx, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat, &done); break;
      case 5: case 6: case 7: case 8: break;   // beyond CFAST: the generic evaluator below
      default: return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
    }
    if (rc || done) return rc;
  }
  const int64_t threads = m * BRENT_LPT;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  const size_t lds = sizeof(double) * (size_t)nm.n * (1 + nm.c);
#define LG(C) hipLaunchKernelGGL(k_loglik_grid<C>, dim3(blocks), dim3(256), lds, ctx->stream, nm, Yt, ldy, m, Z0, lam, grid_dev, ngrid, EllTab, h2idx, h2, stat)
  switch (nm.c) {
    BLMM_FOR_EACH_C(LG)
    default: return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
  }
#undef LG
  KCHECK();
  return BLMM_OK;
}

// BLMM_FLAG_H2_AUDIT: local maxima of every trait's profile log-likelihood on a grid (EllTab: ngrid x m from
// launch_loglik_grid).  Grid point g is a local maximum when it exceeds both neighbours by more than 1e-9 |Ell| (the end
// points have one neighbour); two or more -> the trait is counted in stat[ST_H2_MULTIMODAL]: a local optimiser (Brent,
// src/gridbrent.jl:9-24) may end in either one.
__global__ void __launch_bounds__(256) k_h2_audit(const double* __restrict__ EllTab, int ngrid, int64_t m, int64_t* stat) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int nmax = 0;
  if (j < m && ngrid >= 2) {
    const double* e = EllTab + j * (int64_t)ngrid;
    for (int g = 0; g < ngrid; ++g) {
      const double v = e[g], slack = 1e-9 * fabs(v);
      const bool up = (g == 0) || v > e[g - 1] + slack;
      const bool dn = (g == ngrid - 1) || v > e[g + 1] + slack;
      nmax += (up && dn) ? 1 : 0;
    }
  }
  wave_count(&stat[ST_H2_MULTIMODAL], nmax >= 2);
}

int launch_h2_audit(blmm_ctx* ctx, const double* EllTab, int ngrid, int64_t m, int64_t* stat) {
  if (m <= 0) return BLMM_OK;
  hipLaunchKernelGGL(k_h2_audit, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, EllTab, ngrid, m, stat);
  KCHECK();
  return BLMM_OK;
}

// ------------------------------------------------------------------------------------------------
// A-side panels for the LOD kernels (src/bulkscan_helpers.jl:138-145 and :182-196 rewritten as GEMM
// operands).  With w = makeweights(h2_j), beta = (Z0'WZ0)^-1 Z0'Wy, A = L L':
//   panel 0 [k][j] = w_k (y - Z0 beta)_k / sqrt(yy),  yy = y'Wy - |L^-1 Z0'Wy|^2    -> num  = x' panel0
//   panel 1 [k][j] = w_k                                                              -> Sxx  = (x.^2)' panel1
//   panel 2+q [k][j] = w_k (Z0 L^-T)_kq                                               -> u_q  = x' panel(2+q)
// so that r = num / sqrt(Sxx - sum_q u_q^2)  is the correlation of computeR_LMM (src/bulkscan_helpers.jl:47-64).
// One thread per trait; rows of the panels are written coalesced.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) k_panels(NullModel nm, const double* __restrict__ Yt, int64_t ldy, int64_t m,
                                                const double* __restrict__ Z0, const double* __restrict__ lam,
                                                const double* __restrict__ h2v, int full, double* __restrict__ P,
                                                int64_t ldp, int64_t* stat, const double* __restrict__ gridv) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int n = nm.n, npad = nm.npad;
  // gridv (alt-grid): blockIdx.y = grid point g, every trait at h2 = gridv[g], panel g of the output -- the sixteen launches
  // (and sixteen fills of a constant h2 vector) of rounds 1-3a were 0.57 ms of launch-bound prep per call
  if (gridv) P += (int64_t)blockIdx.y * npad * ldp;
  double* sLam = sh;
  double* sZ = sh + n;
  for (int e = threadIdx.x; e < n; e += blockDim.x) sLam[e] = lam[e];
  for (int e = threadIdx.x; e < n * C; e += blockDim.x) sZ[e] = Z0[e];
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ldp) return;
  const int64_t pstride = (int64_t)npad * ldp;
  const int np = full ? 2 + C : 1;
  if (j >= m) {  // padding columns
    for (int q = 0; q < np; ++q)
      for (int k = 0; k < npad; ++k) P[q * pstride + (int64_t)k * ldp + j] = 0.0;
    return;
  }
  constexpr int NA = C * (C + 1) / 2;
  const double h2 = gridv ? gridv[blockIdx.y] : h2v[j];
  const double delta = h2 / (1.0 - h2);
  double A[NA], v[C], syy = 0.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
#pragma unroll
  for (int q = 0; q < C; ++q) v[q] = 0.0;
  for (int k = 0; k < n; ++k) {
    const double w = fabs(1.0 / fma(delta, sLam[k], 1.0));  // sqrt.(abs.(makeweights)) squared, src/bulkscan_helpers.jl:138
    const double y = Yt[(int64_t)k * ldy + j];
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
  // L (lower Cholesky), Linv (lower), t = L^-1 v, beta = L^-T t
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
    // row q of Linv
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
  for (int k = 0; k < npad; ++k) {
    double p0 = 0.0, w = 0.0;
    double zl[C];
#pragma unroll
    for (int q = 0; q < C; ++q) zl[q] = 0.0;
    if (k < n) {
      w = fabs(1.0 / fma(delta, sLam[k], 1.0));
      double res = Yt[(int64_t)k * ldy + j];
#pragma unroll
      for (int q = 0; q < C; ++q) res = fma(-beta[q], sZ[q * n + k], res);
      p0 = w * res * isy;
#pragma unroll
      for (int q = 0; q < C; ++q) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r <= q; ++r) s = fma(Li[q * (q + 1) / 2 + r], sZ[r * n + k], s);
        zl[q] = w * s;
      }
    }
    P[(int64_t)k * ldp + j] = p0;
    if (full) {
      P[pstride + (int64_t)k * ldp + j] = w;
#pragma unroll
      for (int q = 0; q < C; ++q) P[(2 + q) * pstride + (int64_t)k * ldp + j] = zl[q];
    }
  }
}

int launch_panels(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, int64_t m, const double* Z0,
                  const double* lam, const double* h2, int full, double* panels, int64_t ldp, int64_t* stat, const double* gridv,
                  int ngrid) {
  if (nm.c > CTPL && nm.c <= CMAX) {
    if (gridv) return fail(ctx, BLMM_ERR_INVALID, "launch_panels: the batched grid form has no run-time-c kernel");
    return launch_dyn_panels(ctx, nm, Yt, ldy, m, Z0, lam, h2, full, panels, ldp, stat);
  }
  const unsigned blocks = (unsigned)((ldp + 255) / 256);
  const unsigned gy = (unsigned)(gridv ? ngrid : 1);
  if (gridv && (full || ngrid < 1)) return fail(ctx, BLMM_ERR_INVALID, "launch_panels: the batched grid form writes panel 0 only");
  const size_t lds = sizeof(double) * (size_t)nm.n * (1 + nm.c);
#define PN(C) hipLaunchKernelGGL(k_panels<C>, dim3(blocks, gy), dim3(256), lds, ctx->stream, nm, Yt, ldy, m, Z0, lam, h2, full, panels, ldp, stat, gridv)
  switch (nm.c) {
    BLMM_FOR_EACH_C(PN)
    default: return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
  }
#undef PN
  KCHECK();
  return BLMM_OK;
}

// ------------------------------------------------------------------------------------------------
// isx[g][i] = 1 / || P_g (sqrt(w_g) .* x_i) ||  for every grid point g and marker i: the marker-side
// normalisation of computeR_LMM (src/bulkscan_helpers.jl:51,55,58) under the shared weights of
// weighted_liteqtl (src/bulkscan_helpers.jl:182-193).  grid = (p/256, ngrid).
// ------------------------------------------------------------------------------------------------
// XF32: the markers are the fragment-major fp32 matrix of kernels_scan_f32.hip (F[kb][h][col][j] = x[8 kb + 2 j + h][col], ld = ldx) --
// the fp32 permutation path never forms the fp64 rotated markers; sums in fp64 either way
template <int C, bool XF32 = false>
__global__ void __launch_bounds__(256) k_isx(NullModel nm, const double* __restrict__ Xt, int64_t ldx, int64_t p,
                                             const double* __restrict__ Z0, const double* __restrict__ lam,
                                             const double* __restrict__ grid, double* __restrict__ isx, int64_t ld_isx,
                                             int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int n = nm.n;
  double* sW = sh;        // n
  double* sZ = sh + n;    // C*n
  __shared__ double sLi[CMAX * CMAX];
  const int g = blockIdx.y;
  const double h2 = grid[g];
  const double delta = h2 / (1.0 - h2);
  for (int e = threadIdx.x; e < n; e += blockDim.x) sW[e] = fabs(1.0 / fma(delta, lam[e], 1.0));
  for (int e = threadIdx.x; e < n * C; e += blockDim.x) sZ[e] = Z0[e];
  __syncthreads();
  constexpr int NA = C * (C + 1) / 2;
  __shared__ double sA[CMAX * (CMAX + 1) / 2];
  if constexpr (XF32) {
    // (the n-long sums of Z'WZ by wave 0 -- lanes over k, one wave reduction per entry -- instead of by thread 0 alone: 7 us per
    //  workgroup at n = 1000, in front of every workgroup's marker loop; the fp64 instantiations keep thread 0's summation order)
    if (threadIdx.x < 64) {
#pragma unroll
      for (int q = 0; q < C; ++q)
#pragma unroll
        for (int r = 0; r <= q; ++r) {
          double a = 0.0;
          for (int k = threadIdx.x; k < n; k += 64) a = fma(sW[k] * sZ[q * n + k], sZ[r * n + k], a);
          for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
          if (threadIdx.x == 0) sA[q * (q + 1) / 2 + r] = a;
        }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double A[NA], L[NA], Li[NA];
    for (int a = 0; a < NA; ++a) A[a] = 0.0;
    if constexpr (XF32) { for (int a = 0; a < NA; ++a) A[a] = sA[a]; }
    else
    for (int k = 0; k < n; ++k)
      for (int q = 0; q < C; ++q)
        for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(sW[k] * sZ[q * n + k], sZ[r * n + k], A[q * (q + 1) / 2 + r]);
    for (int q = 0; q < C; ++q)
      for (int r = 0; r <= q; ++r) {
        double s = A[q * (q + 1) / 2 + r];
        for (int u = 0; u < r; ++u) s = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], s);
        L[q * (q + 1) / 2 + r] = (r == q) ? sqrt(s) : s / L[r * (r + 1) / 2 + r];
      }
    for (int q = 0; q < C; ++q)
      for (int r = 0; r <= q; ++r) {
        double s = (r == q) ? 1.0 : 0.0;
        for (int u = r; u < q; ++u) s = fma(-L[q * (q + 1) / 2 + u], Li[u * (u + 1) / 2 + r], s);
        Li[q * (q + 1) / 2 + r] = s / L[q * (q + 1) / 2 + q];
      }
    for (int q = 0; q < C; ++q)
      for (int r = 0; r < C; ++r) sLi[q * CMAX + r] = (r <= q) ? Li[q * (q + 1) / 2 + r] : 0.0;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ld_isx) return;
  double out = 0.0;
  if (i < p) {
    double sxx = 0.0, sxz[C];
#pragma unroll
    for (int q = 0; q < C; ++q) sxz[q] = 0.0;
    auto term = [&](int k, double x) {
      const double wx = sW[k] * x;
      sxx = fma(wx, x, sxx);
#pragma unroll
      for (int q = 0; q < C; ++q) sxz[q] = fma(wx, sZ[q * n + k], sxz[q]);
    };
    if constexpr (XF32) {
      typedef float f4x __attribute__((ext_vector_type(4)));
      const f4x* F = reinterpret_cast<const f4x*>(Xt);
      // two accumulator sets (h = 0 / 1 pieces), four pieces' loads per trip: the single fma chain of n terms behind one load per
      // trip took 187 us at n = 1000, p = 1e5
      double sxx1 = 0.0, sxz1[C];
#pragma unroll
      for (int q = 0; q < C; ++q) sxz1[q] = 0.0;
      auto term1 = [&](int k, double x) {
        const double wx = sW[k] * x;
        sxx1 = fma(wx, x, sxx1);
#pragma unroll
        for (int q = 0; q < C; ++q) sxz1[q] = fma(wx, sZ[q * n + k], sxz1[q]);
      };
      const int KH = 2 * ((n + 7) / 8);                     // kh = kb * 2 + h holds k = 8 kb + 2 j + h, j = 0 .. 3
      int kh = 0;
      for (; kh + 4 <= KH; kh += 4) {
        f4x v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = F[(int64_t)(kh + u) * ldx + i];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kb = (kh + u) >> 1, h = (kh + u) & 1;
#pragma unroll
          for (int j = 0; j < 4; ++j) { const int k = 8 * kb + 2 * j + h; if (k < n) { if (u & 1) term1(k, (double)v[u][j]); else term(k, (double)v[u][j]); } }
        }
      }
      for (; kh < KH; ++kh) {
        const f4x v = F[(int64_t)kh * ldx + i];
        const int kb = kh >> 1, h = kh & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = 8 * kb + 2 * j + h; if (k < n) term(k, (double)v[j]); }
      }
      sxx += sxx1;
#pragma unroll
      for (int q = 0; q < C; ++q) sxz[q] += sxz1[q];
    } else {
      for (int k = 0; k < n; ++k) term(k, Xt[(int64_t)k * ldx + i]);
    }
    double uu = 0.0;
#pragma unroll
    for (int q = 0; q < C; ++q) {
      double u = 0.0;
#pragma unroll
      for (int r = 0; r <= q; ++r) u = fma(sLi[q * CMAX + r], sxz[r], u);
      uu = fma(u, u, uu);
    }
    const double xx = sxx - uu;
    if (!(sqrt(fabs(xx)) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
    out = 1.0 / sqrt(xx);
  }
  isx[(int64_t)g * ld_isx + i] = out;
}

int launch_isx(blmm_ctx* ctx, const NullModel& nm, const double* Xt, int64_t ldx, int64_t p, const double* Z0,
               const double* lam, const double* grid_dev, int ngrid, double* isx, int64_t ld_isx, int64_t* stat) {
  if (nm.c > CTPL && nm.c <= CMAX) return launch_dyn_isx(ctx, nm, Xt, ldx, p, Z0, lam, grid_dev, ngrid, isx, ld_isx, stat);
  dim3 grid((unsigned)((ld_isx + 255) / 256), (unsigned)ngrid);
  const size_t lds = sizeof(double) * (size_t)nm.n * (1 + nm.c);
#define IX(C) hipLaunchKernelGGL(k_isx<C>, grid, dim3(256), lds, ctx->stream, nm, Xt, ldx, p, Z0, lam, grid_dev, isx, ld_isx, stat)
  switch (nm.c) {
    BLMM_FOR_EACH_C(IX)
    default: return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
  }
#undef IX
  KCHECK();
  return BLMM_OK;
}

// marker norms from the fragment-major fp32 markers XF (ldxf >= ld_isx); c <= 3 (the fp32 rotation path is taken for those)
int launch_isx_f32(blmm_ctx* ctx, const NullModel& nm, const float* XF, int64_t ldxf, int64_t p, const double* Z0, const double* lam,
                   const double* h2_dev, double* isx, int64_t ld_isx, int64_t* stat) {
  dim3 grid((unsigned)((ld_isx + 255) / 256), 1u);
  const size_t lds = sizeof(double) * (size_t)nm.n * (1 + nm.c);
  const double* X = reinterpret_cast<const double*>(XF);
  switch (nm.c) {
    case 1: hipLaunchKernelGGL((k_isx<1, true>), grid, dim3(256), lds, ctx->stream, nm, X, ldxf, p, Z0, lam, h2_dev, isx, ld_isx, stat); break;
    case 2: hipLaunchKernelGGL((k_isx<2, true>), grid, dim3(256), lds, ctx->stream, nm, X, ldxf, p, Z0, lam, h2_dev, isx, ld_isx, stat); break;
    case 3: hipLaunchKernelGGL((k_isx<3, true>), grid, dim3(256), lds, ctx->stream, nm, X, ldxf, p, Z0, lam, h2_dev, isx, ld_isx, stat); break;
    default: return fail(ctx, BLMM_ERR_UNSUPPORTED, "isx_f32: c <= 3");
  }
  KCHECK();
  return BLMM_OK;
}

// ------------------------------------------------------------------------------------------------
// calcKinship (src/kinship.jl:4-14): K = 2 (G-1/2)(G-1/2)'/p + 1/2, diag = 1.
// Stage 1: each block sums a slice of markers for a 32x32 output tile (deterministic, no atomics);
// stage 2: sums the slices in fixed order and applies the affine map.
// ------------------------------------------------------------------------------------------------
constexpr int KIN_SPLITS = 64;
__global__ void __launch_bounds__(256) k_kinship_partial(const double* __restrict__ G, int64_t n, int64_t p,
                                                         double* __restrict__ part) {
  __shared__ double sa[32][33], sb[32][33];
  const int ti = blockIdx.x, tj = blockIdx.y, sp = blockIdx.z;
  if (tj > ti) return;  // symmetric: lower tiles only
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t chunk = (p + KIN_SPLITS - 1) / KIN_SPLITS;
  const int64_t k0 = sp * chunk, k1 = (k0 + chunk < p) ? k0 + chunk : p;
  double acc[4] = {0, 0, 0, 0};
  for (int64_t kb = k0; kb < k1; kb += 32) {
    for (int r = ty; r < 32; r += 8) {
      const int64_t k = kb + r;
      const int64_t ia = (int64_t)ti * 32 + tx, ib = (int64_t)tj * 32 + tx;
      sa[r][tx] = (k < k1 && ia < n) ? G[k * n + ia] - 0.5 : 0.0;
      sb[r][tx] = (k < k1 && ib < n) ? G[k * n + ib] - 0.5 : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const double a = sa[r][tx];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = fma(a, sb[r][ty + 8 * u], acc[u]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = (int64_t)ti * 32 + tx, j = (int64_t)tj * 32 + ty + 8 * u;
    if (i < n && j < n) part[((int64_t)sp * n + j) * n + i] = acc[u];
  }
}
__global__ void k_kinship_final(const double* __restrict__ part, int64_t n, int64_t p, double* __restrict__ K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * n) return;
  int64_t i = e % n, j = e / n;
  if (i == j) { K[e] = 1.0; return; }
  if (i / 32 < j / 32) { const int64_t t = i; i = j; j = t; }  // only lower tiles were computed
  double s = 0.0;
  for (int sp = 0; sp < KIN_SPLITS; ++sp) s += part[((int64_t)sp * n + j) * n + i];
  K[e] = 2.0 * s / (double)p + 0.5;
}

int launch_kinship(blmm_ctx* ctx, const double* dG, int64_t n, int64_t p, double* dK, double* partial) {
  const unsigned nt = (unsigned)((n + 31) / 32);
  hipLaunchKernelGGL(k_kinship_partial, dim3(nt, nt, KIN_SPLITS), dim3(256), 0, ctx->stream, dG, n, p, partial);
  KCHECK();
  hipLaunchKernelGGL(k_kinship_final, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, ctx->stream, partial, n, p, dK);
  KCHECK();
  return BLMM_OK;
}

// ------------------------------------------------------------------------------------------------
// Permutation panel for scan_perms_lite (src/scan.jl:521-542, src/transform_helpers.jl:57-102):
//   r0 = sqrt(w) .* (y0 - Z0 beta_w)  (weighted-LS residual),  r0perm_b = pi_b(r0),  column 0 = original.
//   L[:, b] = X00' r0perm_b / (||X00_i|| ||r0||),  X00 = P (sqrt(w) .* X0m)
//            = sum_k x_ik * [ sqrt(w_k) (P r0perm_b)_k ] / (...)          (P symmetric idempotent)
// so the A-side panel column b is  sqrt(w) .* (P pi_b(r0)) / ||r0||  and the marker side stays the shared,
// unweighted Xt with the per-marker scale isx (k_isx at the fitted h2).
// One thread per permutation column.  perm_idx == nullptr: Fisher-Yates from a splitmix64 counter stream.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <int C>
__global__ void __launch_bounds__(64) k_perm_panel(NullModel nm, const double* __restrict__ Yt, int64_t ldy,
                                                   const double* __restrict__ Z0, const double* __restrict__ lam,
                                                   const double* __restrict__ h2p, const int32_t* __restrict__ perm_idx,
                                                   int64_t nperms, uint64_t seed, int orig, double* __restrict__ P,
                                                   int64_t ldp, double* __restrict__ r0buf /* n */,
                                                   int32_t* __restrict__ permbuf, int64_t* stat) {
  const int n = nm.n, npad = nm.npad;
  // orig = 1: column 0 = the un-permuted residual, every other column zero;
  // orig = 0: column b = permutation b (b < nperms), zero beyond.
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= ldp) return;
  if ((orig && b > 0) || (!orig && b >= nperms)) {
    for (int k = 0; k < npad; ++k) P[(int64_t)k * ldp + b] = 0.0;
    return;
  }
  constexpr int NA = C * (C + 1) / 2;
  const double h2 = h2p[0];
  const double delta = h2 / (1.0 - h2);
  // weighted normal equations for the permuted residual vector v = pi_b(r0): coefficients of sqrt(w) Z0
  double A[NA], g[C];
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
  for (int q = 0; q < C; ++q) g[q] = 0.0;
  int32_t* myperm = permbuf ? permbuf + b * (int64_t)n : nullptr;
  if (!orig && !perm_idx) {
    uint64_t s = seed * 0xD1342543DE82EF95ull + (uint64_t)(b + 1);
    for (int k = 0; k < n; ++k) myperm[k] = k;
    for (int k = n - 1; k > 0; --k) {
      const int r = (int)(splitmix64(s) % (uint64_t)(k + 1));
      const int32_t t = myperm[k]; myperm[k] = myperm[r]; myperm[r] = t;
    }
  }
  double rr = 0.0;
  for (int k = 0; k < n; ++k) {
    const double w = 1.0 / fma(delta, lam[k], 1.0);
    const double sw = sqrt(w);
    const int src = orig ? k : (perm_idx ? perm_idx[b * (int64_t)n + k] : myperm[k]);
    const double v = r0buf[src];
    rr = fma(v, v, rr);
    for (int q = 0; q < C; ++q) {
      const double zq = sw * Z0[q * n + k];
      g[q] = fma(zq, v, g[q]);
      for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(zq, sw * Z0[r * n + k], A[q * (q + 1) / 2 + r]);
    }
  }
  // solve A beta = g by Cholesky
  double L[NA], t[C], beta[C];
  for (int q = 0; q < C; ++q) {
    for (int r = 0; r <= q; ++r) {
      double s = A[q * (q + 1) / 2 + r];
      for (int u = 0; u < r; ++u) s = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], s);
      L[q * (q + 1) / 2 + r] = (r == q) ? sqrt(s) : s / L[r * (r + 1) / 2 + r];
    }
    double s = g[q];
    for (int u = 0; u < q; ++u) s = fma(-L[q * (q + 1) / 2 + u], t[u], s);
    t[q] = s / L[q * (q + 1) / 2 + q];
  }
  for (int q = C - 1; q >= 0; --q) {
    double s = t[q];
    for (int u = q + 1; u < C; ++u) s = fma(-L[u * (u + 1) / 2 + q], beta[u], s);
    beta[q] = s / L[q * (q + 1) / 2 + q];
  }
  const double inr = 1.0 / sqrt(rr);
  if (orig && !(sqrt(rr) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
  for (int k = 0; k < npad; ++k) {
    double out = 0.0;
    if (k < n) {
      const double w = 1.0 / fma(delta, lam[k], 1.0);
      const double sw = sqrt(w);
      const int src = orig ? k : (perm_idx ? perm_idx[b * (int64_t)n + k] : myperm[k]);
      double v = r0buf[src];
      for (int q = 0; q < C; ++q) v = fma(-beta[q], sw * Z0[q * n + k], v);
      out = sw * v * inr;
    }
    P[(int64_t)k * ldp + b] = out;
  }
}

// r0 = sqrt(w) .* (y0 - Z0 beta_w) for the single trait in Yt[:, 0]; scalars[0] = sigma2 (given), [1] = h2 (given)
template <int C>
__global__ void k_perm_r0(NullModel nm, const double* __restrict__ Yt, int64_t ldy, const double* __restrict__ Z0,
                          const double* __restrict__ lam, const double* __restrict__ h2p, double* __restrict__ r0buf) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int n = nm.n;
  constexpr int NA = C * (C + 1) / 2;
  const double h2 = h2p[0];
  const double delta = h2 / (1.0 - h2);
  double A[NA], g[C];
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
  for (int q = 0; q < C; ++q) g[q] = 0.0;
  for (int k = 0; k < n; ++k) {
    const double w = 1.0 / fma(delta, lam[k], 1.0);
    const double y = Yt[(int64_t)k * ldy];
    for (int q = 0; q < C; ++q) {
      const double wz = w * Z0[q * n + k];
      g[q] = fma(wz, y, g[q]);
      for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(wz, Z0[r * n + k], A[q * (q + 1) / 2 + r]);
    }
  }
  double L[NA], t[C], beta[C];
  for (int q = 0; q < C; ++q) {
    for (int r = 0; r <= q; ++r) {
      double s = A[q * (q + 1) / 2 + r];
      for (int u = 0; u < r; ++u) s = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], s);
      L[q * (q + 1) / 2 + r] = (r == q) ? sqrt(s) : s / L[r * (r + 1) / 2 + r];
    }
    double s = g[q];
    for (int u = 0; u < q; ++u) s = fma(-L[q * (q + 1) / 2 + u], t[u], s);
    t[q] = s / L[q * (q + 1) / 2 + q];
  }
  for (int q = C - 1; q >= 0; --q) {
    double s = t[q];
    for (int u = q + 1; u < C; ++u) s = fma(-L[u * (u + 1) / 2 + q], beta[u], s);
    beta[q] = s / L[q * (q + 1) / 2 + q];
  }
  for (int k = 0; k < n; ++k) {
    const double w = 1.0 / fma(delta, lam[k], 1.0);
    double v = Yt[(int64_t)k * ldy];
    for (int q = 0; q < C; ++q) v = fma(-beta[q], Z0[q * n + k], v);
    r0buf[k] = sqrt(w) * v;
  }
}

// ---- the same panel for large n x nperms, spread over the chip.  The one-thread-per-column kernels above walk n
//      rows serially (0.7 ms per launch at n = 1000 whatever the number of columns, 0.36 ms for r0 on a single thread). ----
template <int C>
__global__ void __launch_bounds__(256) k_perm_r0_wg(NullModel nm, const double* __restrict__ Yt, int64_t ldy,
                                                    const double* __restrict__ Z0, const double* __restrict__ lam,
                                                    const double* __restrict__ h2p, double* __restrict__ r0buf) {
  constexpr int NA = C * (C + 1) / 2;
  __shared__ double s_part[NA + C][4], s_beta[C];
  const int n = nm.n, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double h2 = h2p[0];
  const double delta = h2 / (1.0 - h2);
  double A[NA], g[C];
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
  for (int q = 0; q < C; ++q) g[q] = 0.0;
  for (int k = t; k < n; k += 256) {
    const double w = 1.0 / fma(delta, lam[k], 1.0);
    const double y = Yt[(int64_t)k * ldy];
    for (int q = 0; q < C; ++q) {
      const double wz = w * Z0[q * n + k];
      g[q] = fma(wz, y, g[q]);
      for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(wz, Z0[r * n + k], A[q * (q + 1) / 2 + r]);
    }
  }
  for (int a = 0; a < NA + C; ++a) {
    double v = (a < NA) ? A[a] : g[a - NA];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) s_part[a][wave] = v;
  }
  __syncthreads();
  if (t == 0) {
    for (int a = 0; a < NA; ++a) A[a] = (s_part[a][0] + s_part[a][1]) + (s_part[a][2] + s_part[a][3]);
    for (int q = 0; q < C; ++q) g[q] = (s_part[NA + q][0] + s_part[NA + q][1]) + (s_part[NA + q][2] + s_part[NA + q][3]);
    double L[NA], tt[C], beta[C];
    for (int q = 0; q < C; ++q) {
      for (int r = 0; r <= q; ++r) {
        double sacc = A[q * (q + 1) / 2 + r];
        for (int u = 0; u < r; ++u) sacc = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], sacc);
        L[q * (q + 1) / 2 + r] = (r == q) ? sqrt(sacc) : sacc / L[r * (r + 1) / 2 + r];
      }
      double sacc = g[q];
      for (int u = 0; u < q; ++u) sacc = fma(-L[q * (q + 1) / 2 + u], tt[u], sacc);
      tt[q] = sacc / L[q * (q + 1) / 2 + q];
    }
    for (int q = C - 1; q >= 0; --q) {
      double sacc = tt[q];
      for (int u = q + 1; u < C; ++u) sacc = fma(-L[u * (u + 1) / 2 + q], beta[u], sacc);
      beta[q] = sacc / L[q * (q + 1) / 2 + q];
      s_beta[q] = beta[q];
    }
  }
  __syncthreads();
  for (int k = t; k < n; k += 256) {
    const double w = 1.0 / fma(delta, lam[k], 1.0);
    double v = Yt[(int64_t)k * ldy];
    for (int q = 0; q < C; ++q) v = fma(-s_beta[q], Z0[q * n + k], v);
    r0buf[k] = sqrt(w) * v;
  }
}

// Fisher-Yates per permutation column with the index array in LDS (16-bit entries, k-major), the SAME splitmix64 stream and
// swap sequence as k_perm_panel; writes permbuf[b * n + k].
__global__ void __launch_bounds__(64) k_perm_gen(int n, int64_t nperms, uint64_t seed, int cols, int32_t* __restrict__ permbuf) {
  extern __shared__ unsigned short s_perm[];
  const int t = threadIdx.x;
  const int64_t b = (int64_t)blockIdx.x * cols + t;
  if (t >= cols || b >= nperms) return;
  for (int k = 0; k < n; ++k) s_perm[k * cols + t] = (unsigned short)k;
  uint64_t st = seed * 0xD1342543DE82EF95ull + (uint64_t)(b + 1);
  for (int k = n - 1; k > 0; --k) {
    const int r = (int)(splitmix64(st) % (uint64_t)(k + 1));
    const unsigned short tmp = s_perm[k * cols + t]; s_perm[k * cols + t] = s_perm[r * cols + t]; s_perm[r * cols + t] = tmp;
  }
  for (int k = 0; k < n; ++k) permbuf[b * (int64_t)n + k] = (int32_t)s_perm[k * cols + t];
}

// One wave per column b: beta_b (coefficients of sqrt(w) Z0 for v = pi_b(r0)) and 1 / ||v||  -> coef[b][C + 1]
template <int C>
__global__ void __launch_bounds__(256) k_perm_coef(NullModel nm, const double* __restrict__ Z0, const double* __restrict__ lam,
                                                   const double* __restrict__ h2p, const int32_t* __restrict__ perm,
                                                   int64_t ncols, int orig, const double* __restrict__ r0buf,
                                                   double* __restrict__ coef, int64_t* stat) {
  constexpr int NA = C * (C + 1) / 2;
  const int n = nm.n, lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= ncols) return;
  const double h2 = h2p[0];
  const double delta = h2 / (1.0 - h2);
  double A[NA], g[C], rr = 0.0;
  for (int a = 0; a < NA; ++a) A[a] = 0.0;
  for (int q = 0; q < C; ++q) g[q] = 0.0;
  for (int k = lane; k < n; k += 64) {
    const double w = 1.0 / fma(delta, lam[k], 1.0);
    const double sw = sqrt(w);
    const int src = orig ? k : perm[b * (int64_t)n + k];
    const double v = r0buf[src];
    rr = fma(v, v, rr);
    for (int q = 0; q < C; ++q) {
      const double zq = sw * Z0[q * n + k];
      g[q] = fma(zq, v, g[q]);
      for (int r = 0; r <= q; ++r) A[q * (q + 1) / 2 + r] = fma(zq, sw * Z0[r * n + k], A[q * (q + 1) / 2 + r]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    rr += __shfl_xor(rr, o, 64);
    for (int a = 0; a < NA; ++a) A[a] += __shfl_xor(A[a], o, 64);
    for (int q = 0; q < C; ++q) g[q] += __shfl_xor(g[q], o, 64);
  }
  if (lane == 0) {
    double L[NA], tt[C], beta[C];
    for (int q = 0; q < C; ++q) {
      for (int r = 0; r <= q; ++r) {
        double sacc = A[q * (q + 1) / 2 + r];
        for (int u = 0; u < r; ++u) sacc = fma(-L[q * (q + 1) / 2 + u], L[r * (r + 1) / 2 + u], sacc);
        L[q * (q + 1) / 2 + r] = (r == q) ? sqrt(sacc) : sacc / L[r * (r + 1) / 2 + r];
      }
      double sacc = g[q];
      for (int u = 0; u < q; ++u) sacc = fma(-L[q * (q + 1) / 2 + u], tt[u], sacc);
      tt[q] = sacc / L[q * (q + 1) / 2 + q];
    }
    for (int q = C - 1; q >= 0; --q) {
      double sacc = tt[q];
      for (int u = q + 1; u < C; ++u) sacc = fma(-L[u * (u + 1) / 2 + q], beta[u], sacc);
      beta[q] = sacc / L[q * (q + 1) / 2 + q];
      coef[b * (C + 1) + q] = beta[q];
    }
    coef[b * (C + 1) + C] = 1.0 / sqrt(rr);
    if (orig && !(sqrt(rr) > 2.220446049250313e-16)) atomicAdd((unsigned long long*)&stat[ST_ZERO_NORM], 1ull);
  }
}

// P[k][b] = sqrt(w_k) (v_k - sum_q beta_bq sqrt(w_k) z_qk) / ||v||, v = pi_b(r0); zero beyond ncols / n.  Threads along b.
template <int C>
__global__ void __launch_bounds__(256) k_perm_fill(NullModel nm, const double* __restrict__ Z0, const double* __restrict__ lam,
                                                   const double* __restrict__ h2p, const int32_t* __restrict__ perm,
                                                   int64_t ncols, int orig, const double* __restrict__ r0buf,
                                                   const double* __restrict__ coef, double* __restrict__ P, int64_t ldp) {
  const int n = nm.n, npad = nm.npad;
  const int64_t b = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int k0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 16;
  if (b >= ldp) return;
  const double h2 = h2p[0];
  const double delta = h2 / (1.0 - h2);
  const bool live = b < ncols;
  double beta[C], inr = 0.0;
  for (int q = 0; q < C; ++q) beta[q] = live ? coef[b * (C + 1) + q] : 0.0;
  if (live) inr = coef[b * (C + 1) + C];
  for (int k = k0; k < k0 + 16 && k < npad; ++k) {
    double out = 0.0;
    if (live && k < n) {
      const double w = 1.0 / fma(delta, lam[k], 1.0);
      const double sw = sqrt(w);
      const int src = orig ? k : perm[b * (int64_t)n + k];
      double v = r0buf[src];
      for (int q = 0; q < C; ++q) v = fma(-beta[q], sw * Z0[q * n + k], v);
      out = sw * v * inr;
    }
    P[(int64_t)k * ldp + b] = out;
  }
}

// The library's own permutations (Fisher-Yates from the splitmix64 counter stream of `seed`) depend on nothing but (n, nperms,
// seed): the permutation test starts this on the side stream at the head of the call, beside the eigen-decomposition (0.38 ms at
// n = 1000, 1250 permutations, a few latency-bound workgroups; k_sytrd leaves 150 CUs idle meanwhile); launch_perm_panel then
// finds the indices in ctx->perm (ctx->perm_ready) instead of generating them on the critical path.
int launch_perm_gen(blmm_ctx* ctx, int n, int64_t nperms, uint64_t seed) {
  if (nperms <= 0 || n > 65535) return BLMM_OK;
  int rc = ensure(ctx, ctx->perm, sizeof(int32_t) * (size_t)n * (size_t)(nperms + 1));
  if (rc) return rc;
  int cols = (int)std::min<size_t>(64, (150 * 1024) / (sizeof(unsigned short) * (size_t)n));
  if (cols < 1) cols = 1;
  const size_t lds = sizeof(unsigned short) * (size_t)n * cols;
  BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_perm_gen), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_perm_gen, dim3((unsigned)((nperms + cols - 1) / cols)), dim3(64), lds, ctx->stream, n, nperms, seed, cols, ptr<int32_t>(ctx->perm));
  KCHECK();
  ctx->perm_ready_n = n; ctx->perm_ready_nperms = nperms; ctx->perm_ready_seed = seed; ctx->perm_ready = true;
  return BLMM_OK;
}

int launch_perm_panel(blmm_ctx* ctx, const NullModel& nm, const double* Yt, int64_t ldy, const double* Z0,
                      const double* lam, const double* h2, const int32_t* perm_idx, int64_t nperms, uint64_t seed,
                      int orig, double* panel, int64_t ldp, int64_t* stat) {
  int rc = ensure(ctx, ctx->r0, sizeof(double) * (size_t)nm.n);
  if (rc) return rc;
  // indices generated ahead of this call (launch_perm_gen, ordered in front of this stream position by the caller)
  const bool pregen = !perm_idx && !orig && ctx->perm_ready && ctx->perm_ready_n == nm.n && ctx->perm_ready_nperms == nperms &&
                      ctx->perm_ready_seed == seed;
  if (!orig) ctx->perm_ready = false;
  int32_t* permbuf = nullptr;
  if (!perm_idx && !orig) {
    rc = ensure(ctx, ctx->perm, sizeof(int32_t) * (size_t)nm.n * (size_t)(nperms + 1));
    if (rc) return rc;
    permbuf = ptr<int32_t>(ctx->perm);
  }
  double* r0 = ptr<double>(ctx->r0);
  if (nm.c > CTPL && nm.c <= CMAX) {
    // run-time covariate count (kernels_dyn.hip): permutations from the same generator (k_perm_gen), one wave per column
    if (nm.n > 65535) return fail(ctx, BLMM_ERR_UNSUPPORTED, "permutation test: n > 65535");
    const int64_t ncols = orig ? 1 : nperms;
    const int32_t* pidx = perm_idx;
    if (pregen) pidx = permbuf;
    else if (!orig && !perm_idx && nperms > 0) {
      int cols = (int)std::min<size_t>(64, (150 * 1024) / (sizeof(unsigned short) * (size_t)nm.n));
      if (cols < 1) cols = 1;
      const size_t lds = sizeof(unsigned short) * (size_t)nm.n * cols;
      BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_perm_gen), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_perm_gen, dim3((unsigned)((nperms + cols - 1) / cols)), dim3(64), lds, ctx->stream, nm.n, nperms, seed, cols, permbuf);
      KCHECK();
      pidx = permbuf;
    }
    return launch_dyn_perm(ctx, nm, Yt, ldy, Z0, lam, h2, pidx, ncols, orig, r0, panel, ldp, stat);
  }
  // large n: the multi-kernel form (BLMM_PERM_PATH=old|new forces one; read per call so that a test can compare them)
  const char* path_env = dev_env("BLMM_PERM_PATH");
  const bool newpath = path_env ? std::strcmp(path_env, "new") == 0 : nm.n > 256;
  if (newpath && nm.n <= 65535) {
    const int64_t ncols = orig ? 1 : nperms;
    rc = ensure(ctx, ctx->tmpB, sizeof(double) * (size_t)(ncols > 0 ? ncols : 1) * (CMAX + 1) + 64);
    if (rc) return rc;
    double* coef = ptr<double>(ctx->tmpB);
    const int32_t* pidx = perm_idx;
    if (pregen) pidx = permbuf;
    else if (!orig && !perm_idx && nperms > 0) {
      int cols = (int)std::min<size_t>(64, (150 * 1024) / (sizeof(unsigned short) * (size_t)nm.n));
      if (cols < 1) cols = 1;
      const size_t lds = sizeof(unsigned short) * (size_t)nm.n * cols;
      BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_perm_gen), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_perm_gen, dim3((unsigned)((nperms + cols - 1) / cols)), dim3(64), lds, ctx->stream, nm.n, nperms, seed, cols, permbuf);
      KCHECK();
      pidx = permbuf;
    }
    dim3 fgrid((unsigned)((ldp + 63) / 64), (unsigned)((nm.npad + 63) / 64));
#define PN(C)                                                                                                              \
    if (orig) hipLaunchKernelGGL(k_perm_r0_wg<C>, dim3(1), dim3(256), 0, ctx->stream, nm, Yt, ldy, Z0, lam, h2, r0);       \
    if (ncols > 0) hipLaunchKernelGGL(k_perm_coef<C>, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, ctx->stream, nm, Z0, lam, h2, pidx, ncols, orig, r0, coef, stat); \
    hipLaunchKernelGGL(k_perm_fill<C>, fgrid, dim3(256), 0, ctx->stream, nm, Z0, lam, h2, pidx, ncols, orig, r0, coef, P_, ldp)
    double* P_ = panel;
    switch (nm.c) {
      BLMM_FOR_EACH_C(PN)
      default: return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
    }
#undef PN
    KCHECK();
    return BLMM_OK;
  }
  const unsigned blocks = (unsigned)((ldp + 63) / 64);
#define PP(C)                                                                                                     \
  if (orig) hipLaunchKernelGGL(k_perm_r0<C>, dim3(1), dim3(64), 0, ctx->stream, nm, Yt, ldy, Z0, lam, h2, r0);   \
  hipLaunchKernelGGL(k_perm_panel<C>, dim3(blocks), dim3(64), 0, ctx->stream, nm, Yt, ldy, Z0, lam, h2, perm_idx, \
                     nperms, seed, orig, panel, ldp, r0, permbuf, stat)
  switch (nm.c) {
    BLMM_FOR_EACH_C(PP)
    default: return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
  }
#undef PP
  KCHECK();
  return BLMM_OK;
}

// ------------------------------------------------------------------------------------------------
// Column maxima of an LOD matrix (p x m, ld = ldL): per-trait / per-permutation peak and the marker where it sits.
// The consumer behind get_thresholds (src/analysis_helpers/single_trait_analysis.jl:13-23) and the usual
// "max LOD per trait" summary, so the 2 GB matrix need not leave HBM.  One wave per column,
// 16-byte loads, first maximum wins; NaNs are ignored (a column of NaNs gives -inf, marker -1).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_colmax(const double* __restrict__ L, int64_t p, int64_t m, int64_t ldL,
                                                double* __restrict__ mx, int64_t* __restrict__ arg) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= m) return;
  const double* col = L + j * ldL;
  double best = -INFINITY;
  int64_t bi = -1;
  for (int64_t i = lane; i < p; i += 64) {
    const double v = col[i];
    if (v > best) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_xor(best, o, 64);
    const int64_t oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi >= 0 && (bi < 0 || oi < bi))) { best = ob; bi = oi; }
  }
  if (lane == 0) { mx[j] = best; if (arg) arg[j] = bi; }
}

int launch_colmax(blmm_ctx* ctx, const double* L, int64_t p, int64_t m, int64_t ldL, double* mx, int64_t* arg) {
  if (m <= 0) return BLMM_OK;
  hipLaunchKernelGGL(k_colmax, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, L, p, m, ldL, mx, arg);
  KCHECK();
  return BLMM_OK;
}

}  // namespace blmm
