// kernels_eig.hip -- symmetric eigensolver for the kinship matrix beyond the LDS Jacobi's range (n > 124), replacing
// LAPACK `eigen(K)` of transform_rotation (src/transform_helpers.jl:21-34) without leaving the GPU and without a vendor
// library on the hot path (rocSOLVER dsyevd took 13 / 23 ms at n = 500 / 1000, 70-80 % of a shard's step).
//
//   1. k_sytrd          Householder tridiagonalisation K = H T H'.  The matrix lives in the LDS of G workgroups (rows
//                       dealt cyclically, full symmetric storage), so a step costs ONE exchange: every workgroup
//                       publishes its rows of p = tau A v, the owner of row k+1 publishes that row, one grid barrier,
//                       then every workgroup forms w, updates its rows and derives the next column on its own.
//   2. k_tql_leaves     implicit QL on leaf blocks (<= 32) of T, one wave per leaf.
//   3. k_dc_*           Cuppen's divide and conquer up the tree: deflation (as LAPACK dlaed2), secular roots relative to
//                       the nearer pole, Gu-Eisenstat z-hat for orthogonal vectors, the update Q <- Q W as an f64-MFMA GEMM.
//   4. k_backtransform  U = H Z, reflectors applied to column slabs held in registers.
//
// tools/dc_prototype.py is the NumPy model of exactly this data flow (same conventions, same tolerances).
// Eigenvalues come out ascending; eigenvector i is evec[i*n .. i*n+n) (what k_post_eigen expects).
#include "blmm_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace blmm {

#define KCHECK()                                                                                      \
  do {                                                                                                \
    hipError_t e__ = hipGetLastError();                                                               \
    if (e__ != hipSuccess) return fail(ctx, BLMM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e__)); \
  } while (0)

namespace {

constexpr double EPS = 2.220446049250313e-16;
constexpr int LEAF = 32;          // largest leaf block of T (LDS arrays of k_tql_leaves); the plan aims lower at small n, see launch_eig_dc

// Wave-wide reductions on the DPP path (quad_perm / row_half_mirror / row_mirror inside a row of 16 lanes, then the four
// row totals through v_readlane): ~60 cycles for a double, against ~700 for the six ds_bpermute pairs of a __shfl_xor
// butterfly -- these sit on the critical path of every row of k_sytrd and every reflector of k_backtransform.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_bcast(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wsum(double x) {
  x += dpp_mov<0xB1>(x);    // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);    // quad_perm [2,3,0,1]
  x += dpp_mov<0x141>(x);   // row_half_mirror
  x += dpp_mov<0x140>(x);   // row_mirror: every lane holds its row's total
  return (lane_bcast(x, 0) + lane_bcast(x, 16)) + (lane_bcast(x, 32) + lane_bcast(x, 48));
}
__device__ __forceinline__ double wmax(double x) {
  x = fmax(x, dpp_mov<0xB1>(x)); x = fmax(x, dpp_mov<0x4E>(x)); x = fmax(x, dpp_mov<0x141>(x)); x = fmax(x, dpp_mov<0x140>(x));
  return fmax(fmax(lane_bcast(x, 0), lane_bcast(x, 16)), fmax(lane_bcast(x, 32), lane_bcast(x, 48)));
}
__device__ __forceinline__ double wprod(double x) {
  x *= dpp_mov<0xB1>(x); x *= dpp_mov<0x4E>(x); x *= dpp_mov<0x141>(x); x *= dpp_mov<0x140>(x);
  return (lane_bcast(x, 0) * lane_bcast(x, 16)) * (lane_bcast(x, 32) * lane_bcast(x, 48));
}
// Four wave-wide sums for the price of (about) two: the first two butterfly stages fold the four values into ONE per lane (lane l
// then carries the quad's partial sum of d_(l & 3)), two row rotations and the gfx950 row / half-wave swaps finish all four at
// once -- 45 instructions against 4 x 23 of four wsum()s.  Every lane returns all four totals.
__device__ __forceinline__ double swap_rows_sum(double x, bool half) {      // x + (x of the partner row / partner half-wave)
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  const v2u a = half ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const v2u b = half ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
}
__device__ __forceinline__ void wsum4(double& d0, double& d1, double& d2, double& d3) {
  const int lane = threadIdx.x & 63;
  const bool b0 = lane & 1, b1 = lane & 2;
  const double x01 = (b0 ? d1 : d0) + dpp_mov<0xB1>(b0 ? d0 : d1);     // even lanes: d0 over the pair, odd lanes: d1
  const double x23 = (b0 ? d3 : d2) + dpp_mov<0xB1>(b0 ? d2 : d3);     //             d2                       d3
  double x = (b1 ? x23 : x01) + dpp_mov<0x4E>(b1 ? x01 : x23);         // lane l: d_(l & 3) over its quad
  x += dpp_mov<0x124>(x);                                              // row_ror:4
  x += dpp_mov<0x128>(x);                                              // row_ror:8 -> over its row of 16
  x = swap_rows_sum(x, false);                                         // rows 0+1, 2+3
  x = swap_rows_sum(x, true);                                          // both half-waves
  d0 = lane_bcast(x, 0); d1 = lane_bcast(x, 1); d2 = lane_bcast(x, 2); d3 = lane_bcast(x, 3);
}
// 1/sqrt(h) from v_rsq_f64 (good to ~5e-8) and two Newton steps (~1 ulp), for h > 0: the QL rotations take one of these
// instead of an IEEE sqrt followed by an IEEE division (~400 dependent cycles)
__device__ __forceinline__ double fast_rsqrt(double h) {
  double y = __builtin_amdgcn_rsq(h);
  double e = fma(-h * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-h * y, y, 1.0);
  return fma(0.5 * y, e, y);
}
// sum over the workgroup; every thread gets the result.  red: >= 16 doubles of LDS.  Two barriers.
__device__ __forceinline__ double block_sum(double x, double* red) {
  x = wsum(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = x;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}
__device__ __forceinline__ double block_max(double x, double* red) {
  x = wmax(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = x;
  __syncthreads();
  double s = red[0];
  for (int w = 1; w < nw; ++w) s = fmax(s, red[w]);
  return s;
}

// agent-scope (sc1) 8-byte accesses for the exchange of k_sytrd (MI355X_MICROARCH.md, inter-workgroup visibility:
// every handed-off byte is stored and loaded with sc1, the storing waves drain before ONE lane signals)
__device__ __forceinline__ void st_sc1(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}

struct SytrdEx {          // global exchange area of k_sytrd
  unsigned long long* gr; // [2 parities][n rows][4 granules]: {p lo, p hi, a lo, a hi}, each (tag << 32) | 32 data bits
  int* abort;             // set when a workgroup gave up waiting
  const unsigned long long* anorm;   // ABSMAX_PARTS partial maxima of |A_ij| as bit patterns (k_sytrd_prep)
};
constexpr int ABSMAX_PARTS = 32;

// Everything k_sytrd needs set up, in one launch (was: memset, k_absmax with an atomic, memset -- three dependent launches
// in front of every decomposition): block b's maximum of |A_ij| -> part[b] (bit pattern of a non-negative double; k_sytrd
// takes the maximum of the ABSMAX_PARTS), the abort word and the exchange granules zeroed.
__global__ void __launch_bounds__(256) k_sytrd_prep(const double* __restrict__ A, int64_t cnt, unsigned long long* __restrict__ part,
                                                    int* __restrict__ abort8, unsigned long long* __restrict__ gr, int64_t ngr) {
  __shared__ double s_m[4];
  double m = 0.0;
  for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < cnt; e0 += (int64_t)gridDim.x * 256) m = fmax(m, fabs(A[e0]));
  m = wmax(m);
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < ngr; e0 += (int64_t)gridDim.x * 256) gr[e0] = 0ull;
  if (blockIdx.x == 0 && threadIdx.x < 8) abort8[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (unsigned long long)__double_as_longlong(fmax(fmax(s_m[0], s_m[1]), fmax(s_m[2], s_m[3])));
}

// ---------------------------------------------------------------------------------------------------------------------
// 1. tridiagonalisation.  Workgroup g of G owns the rows i = g, g + G, ... (local row li = i / G), all n columns of
// each, in LDS.  Outputs: d (n), e (n - 1), tau (n - 2), V[k*n + j] = v_k[j] for j > k (v_k[k+1] = 1).
//
// Step k, every workgroup: Householder vector v of the current column x (known to all) -> p_i = tau (A v)_i for its rows.
// It publishes p_i together with its own entries a_i = A[i][k+1] (by symmetry the union over the workgroups is row k+1,
// which nobody would otherwise have in full) and gathers everybody's: ONE exchange per step.  Then all form
// w = p - (tau/2)(p'v) v and derive the next column x = a - w - w[k+1] v on their own.
// The rank-2 update of the rows is DEFERRED by one step and applied while the next exchange is in flight: step k's
// products use the rows with updates <= k-2 plus the correction  - vp (wp'v) - wp (vp'v)  for the pending pair (vp, wp).
// Four workgroup barriers per step (they cost ~500 cycles each at 512 threads: the per-step floor besides the exchange).
// Exchange protocol: data-tagged granules (MI355X_MICROARCH.md, persistent-kernel price list: "Granule = one naturally
// aligned 8-byte {data, tag} written by ONE sc1 store", the form that needs no ordering at all).  A double travels as two
// granules (tag = step + 1 in the upper word, 32 data bits in the lower); the consumer spins with sc1 loads on exactly
// the granules it needs until their tags match: no drain, no flag, no counter (64-256 atomics on one word cost 1-3 us per
// step by themselves).  Buffers alternate by step parity (a workgroup can be at most one step ahead of the slowest: its
// step k+1 payload needs everybody's step k payload); spins are bounded (abort word).
// ---------------------------------------------------------------------------------------------------------------------
// GLB: the workgroup's rows live in a slab of global memory (Aw: L2-resident, a workgroup only ever touches its own rows) instead
// of LDS -- n beyond the LDS budget (~1450 on a full MI355X) up to the 2048 the rest of the solver takes.  Same algorithm, same
// exchange; a step then re-reads the workgroup's rows from L2 (n = 2000, 10 rows: 160 KB per step).  A row is written by one
// wave (the deferred update) and read by another a step later, with workgroup barriers in between.
template <bool GLB>
__global__ void __launch_bounds__(512) k_sytrd(const double* __restrict__ A, int n, int nloc_max, double* __restrict__ d,
                                                double* __restrict__ e, double* __restrict__ tau, double* __restrict__ V,
                                                SytrdEx ex, int64_t* stat, double* Aw) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int G = gridDim.x, g = blockIdx.x, t = threadIdx.x, NT = blockDim.x;
  const int lane = t & 63, wave = t >> 6, nwave = NT >> 6;
  double* sx = sh;                  // n : current column x (alternates with sa)
  double* sa = sx + n;              // n : gathered column k+1, turned into the next x in place
  double* sp = sa + n;              // n : gathered p
  double* svc = sp + n;             // n : v of this step        (svc/swc alternate with svp/swp)
  double* swc = svc + n;            // n : w of this step
  double* svp = swc + n;            // n : v of the pending (previous) step
  double* swp = svp + n;            // n : w of the pending step
  double* red = swp + n;            // 2 parities x 4 sums x 16 waves
  double* Al = GLB ? Aw + (size_t)g * nloc_max * n : red + 128;   // nloc_max x n
  const int nloc = (n - g + G - 1) / G;
  for (int li = 0; li < nloc; ++li) {
    const int i = g + li * G;
    for (int j = t; j < n; j += NT) Al[(size_t)li * n + j] = A[(size_t)j * n + i];   // symmetric: row i = column i
  }
  for (int j = t; j < n; j += NT) { sx[j] = (j >= 1) ? A[j] : 0.0; svp[j] = 0.0; swp[j] = 0.0; }   // column 0; no pending update yet
  double akk = A[0];
  __syncthreads();
  // per-wave partial sums of the three reductions a step opens with, written at the END of the previous step (here: for
  // step 0) into rdA[parity]: |x|^2, vp'x, wp'x over j >= k+2
  double* rdA = red;            // [2][3][16]
  double* rdB = red + 96;       // [16] partial p'x of the gather
  {
    double s1 = 0.0;
    for (int j = 2 + t; j < n; j += NT) s1 = fma(sx[j], sx[j], s1);
    s1 = wsum(s1);
    if (lane == 0) { rdA[wave] = s1; rdA[16 + wave] = 0.0; rdA[32 + wave] = 0.0; }
  }
  __syncthreads();
  bool aborted = false;
  // A column whose norm is negligible against |A| gets NO reflector (as an exactly zero one): with tau from a norm whose
  // square is in the denormal range the reflector is not orthogonal -- on a rank-2 kinship the trailing matrix is the
  // rounding residue of the rounding residue ..., 1e-163 after a few steps, and the eigenvectors of the zero cluster came
  // out 0.19 off orthogonality (found by the fuzzer; LAPACK's dlarfg rescales such columns instead).
  double anorm = 0.0;
  for (int b = 0; b < ABSMAX_PARTS; ++b) anorm = fmax(anorm, __longlong_as_double((long long)ex.anorm[b]));
  const double s1_negl = (EPS * anorm) * (EPS * anorm);
#ifdef SYTRD_PROF
  long long pf[6] = {0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
#define PSTAMP(i) do { const long long t1__ = __builtin_amdgcn_s_memtime(); pf[i] += t1__ - pt0; pt0 = t1__; } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
  for (int k = 0; k + 2 < n; ++k) {
    const int par = k & 1;
    // (a) the step's scalars from the partial sums (every thread, redundantly: no barrier)
    // (the 3 x 16 per-wave partials: ONE read per wave -- lane l takes slot l, the three groups are the three DPP rows of 16 -- and a
    //  row reduction; the plain `for (q < nwave) s += rd[q]` compiled to read -> wait -> add chains, and as 48 reads per thread it
    //  was LDS-issue bound: 512 threads x 48 broadcast reads)
    double s1, s2, s3;
    {
      const double* rd = rdA + par * 48;
      double v = rd[(lane < 48) ? lane : 0];
      v = (lane < 48 && (lane & 15) < nwave) ? v : 0.0;
      v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);     // every lane: its row's total
      s1 = lane_bcast(v, 0); s2 = lane_bcast(v, 16); s3 = lane_bcast(v, 32);
    }
    const double alpha = sx[k + 1];
    double beta, tk, sc;
    if (s1 <= s1_negl) { beta = alpha; tk = 0.0; sc = 0.0; }
    else {
      // one v_rsq_f64 + Newton and one v_rcp_f64 + Newton where an IEEE square root and two IEEE divisions stood on the critical
      // path of every step (as in small_sytrd): tau = (beta - alpha) / beta = 1 + |alpha| / |x|, 1 / (alpha - beta) = sign(alpha) / (|alpha| + |x|)
      const double hh = fma(alpha, alpha, s1), rs = fast_rsqrt(hh), nrm = hh * rs;
      beta = -copysign(nrm, alpha);
      tk = fma(fabs(alpha), rs, 1.0);
      double y = __builtin_amdgcn_rcp(fabs(alpha) + nrm);
      y = fma(y, fma(-(fabs(alpha) + nrm), y, 1.0), y);
      y = fma(y, fma(-(fabs(alpha) + nrm), y, 1.0), y);
      sc = copysign(y, alpha);
    }
    const double vpk1 = svp[k + 1], wpk1 = swp[k + 1];
    const double vpv = fma(sc, s2, vpk1), wpv = fma(sc, s3, wpk1);      // vp'v and wp'v with v = (1, sc x[k+2:])
    PSTAMP(0);
    // (b) owned rows i > k:  p_i = tau ((A v)_i - vp_i (wp'v) - wp_i (vp'v)),  a_i = A[i][k+1] with the pending update applied
    const int li0 = (k + 1 > g) ? (k + 1 - g + G - 1) / G : 0;    // first local row with global index > k
    // Two rows per trip (with ~10 rows and 8 waves the first waves own two; they used to take them one after the other: product,
    // wave sum, publish, twice): the products share the loads of x, both sums go through one wsum4, and lanes 0 .. 7 publish the
    // eight granules of the two rows with one store each.
    for (int li = li0 + wave; li < nloc; li += 2 * nwave) {
      const int liB = li + nwave;
      const bool hasB = liB < nloc;
      const double* rowA = Al + (size_t)li * n;
      const double* rowB = Al + (size_t)(hasB ? liB : li) * n;
      double accA = 0.0, accB = 0.0;
      for (int j = k + 2 + lane; j < n; j += 64) { const double x = sx[j]; accA = fma(rowA[j], x, accA); accB = fma(rowB[j], x, accB); }
      { double z0 = 0.0, z1 = 0.0; wsum4(accA, accB, z0, z1); }
      if (lane < 8) {
        const int r = lane >> 2, q = lane & 3;
        if (r == 0 || hasB) {
          const int i = g + (r ? liB : li) * G;
          const double* row = r ? rowB : rowA;
          const double acc = r ? accB : accA;
          const double vpi = svp[i], wpi = swp[i], rk1 = row[k + 1];
          const double pi = tk * (fma(sc, acc, rk1) - vpi * wpv - wpi * vpv);
          const double ai = rk1 - vpi * wpk1 - wpi * vpk1;
          if (G > 1) {
            const unsigned long long tg = (unsigned long long)(unsigned int)(k + 1) << 32;
            const unsigned long long pay = (unsigned long long)__double_as_longlong((q < 2) ? pi : ai);
            unsigned long long* gq = ex.gr + ((size_t)par * n + i) * 4;
            __hip_atomic_store(gq + q, tg | ((q & 1) ? (pay >> 32) : (pay & 0xffffffffull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else if (q == 0) { sp[i] = pi; sa[i] = ai; }
        }
      }
    }
    PSTAMP(1);
    // (c) the pending rank-2 update (step k-1) of the same rows this wave just used, while the exchange is in flight
    if (k > 0) {
      for (int li = li0 + wave; li < nloc; li += nwave) {
        const int i = g + li * G;
        double* row = Al + (size_t)li * n;
        const double vi = svp[i], wi = swp[i];
        for (int j = k + 1 + lane; j < n; j += 64) row[j] = fma(-vi, swp[j], fma(-wi, svp[j], row[j]));
      }
    }
    PSTAMP(2);
    // (d) gather p and a of every row j > k; partial p'x on the way
    double s4 = 0.0;
    bool bad = false;
    if (G > 1) {
      const unsigned int want = (unsigned int)(k + 1);
      for (int j = k + 1 + t; j < n && !bad; j += NT) {
        const unsigned long long* gq = ex.gr + ((size_t)par * n + j) * 4;
        unsigned long long q0, q1, q2, q3;
        int spin = 0;
        for (;;) {
          // the four granules of a row (32 aligned bytes) with two 16-byte sc1 loads: every 8-byte granule validates itself by
          // its tag, so a set read half old, half new only sends the loop round again
          {
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            u64x2 lo2, hi2;
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(lo2), "=&v"(hi2) : "v"(gq) : "memory");
            q0 = lo2.x; q1 = lo2.y; q2 = hi2.x; q3 = hi2.y;
          }
          if ((unsigned int)(q0 >> 32) == want && (unsigned int)(q1 >> 32) == want && (unsigned int)(q2 >> 32) == want &&
              (unsigned int)(q3 >> 32) == want) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spin > (1 << 22) || ((spin & 1023) == 0 && __hip_atomic_load(ex.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { bad = true; break; }
        }
        const double pj = __longlong_as_double((long long)((q1 << 32) | (q0 & 0xffffffffull)));
        sp[j] = pj;
        sa[j] = __longlong_as_double((long long)((q3 << 32) | (q2 & 0xffffffffull)));
        if (j > k + 1) s4 = fma(pj, sx[j], s4);
      }
#ifdef TQL_DIAG
      if (bad && (t & 63) == 0) printf("sytrd wg %d/%d thread %d gave up at step %d (n %d, nloc %d)\n", g, G, t, k, n, nloc);
#endif
      if (bad) __hip_atomic_store(ex.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __syncthreads();     // G = 1: sp / sa were written by the row waves of this workgroup
      for (int j = k + 2 + t; j < n; j += NT) s4 = fma(sp[j], sx[j], s4);
    }
    s4 = wsum(s4);
    if (lane == 0) rdB[wave] = s4;
    if (__syncthreads_or(bad ? 1 : 0)) { aborted = true; break; }        // barrier X
    PSTAMP(3);
    // (e) p'v, w = p - (tau/2)(p'v) v; (f) this step's pair, the next column in place, and the next step's three sums
    {
      double v = rdB[lane & 15];
      v = ((lane & 15) < nwave) ? v : 0.0;
      v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
      s4 = lane_bcast(v, 0);
    }
    const double pk1 = sp[k + 1];
    const double cw = 0.5 * tk * fma(sc, s4, pk1);
    const double wk1 = pk1 - cw;                                         // v[k+1] = 1
    const bool vown = (g == k % G);
    double n1 = 0.0, n2 = 0.0, n3 = 0.0;
    for (int j = k + 1 + t; j < n; j += NT) {
      const double v = (j == k + 1) ? 1.0 : sx[j] * sc;
      const double wj = fma(-cw, v, sp[j]);
      svc[j] = v; swc[j] = wj;
      if (vown) V[(size_t)k * n + j] = v;
      if (j > k + 1) {
        const double xn = sa[j] - wj - wk1 * v;                          // the next column, in place
        sa[j] = xn;
        if (j > k + 2) { n1 = fma(xn, xn, n1); n2 = fma(v, xn, n2); n3 = fma(wj, xn, n3); }
      }
    }
    { double n4 = 0.0; wsum4(n1, n2, n3, n4); }                          // (three wave sums in one pass)
    if (lane == 0) { double* rn = rdA + (par ^ 1) * 48; rn[wave] = n1; rn[16 + wave] = n2; rn[32 + wave] = n3; }
    if (vown && t == 0) { d[k] = akk; e[k] = beta; tau[k] = tk; }
    akk = sa[k + 1] - 2.0 * wk1;                                         // sa[k+1] is not written above
    __syncthreads();                                                     // barrier Y
    { double* tmp = sx; sx = sa; sa = tmp; }
    { double* tmp = svc; svc = svp; svp = tmp; tmp = swc; swc = swp; swp = tmp; }   // this step's pair is now the pending one
    PSTAMP(4);
  }
#ifdef SYTRD_PROF
  if (t == 0 && (g == 0 || g == G - 1))
    printf("sytrd prof wg %d/%d nt %d n %d: scalars %lld symv+publish %lld pending update %lld wait+gather %lld w/next/sums %lld (cycles/step)\n", g, G, NT, n,
           pf[0] / (n - 2), pf[1] / (n - 2), pf[2] / (n - 2), pf[3] / (n - 2), pf[4] / (n - 2));
#endif
  if (aborted) {
    if (g == 0 && t == 0) stat[11] = -7;   // reported as an eigensolver failure (blmm_api.hip: finish_status / k_sticky)
    return;
  }
  if (g == 0 && t == 0) { d[n - 2] = akk; e[n - 2] = sx[n - 1]; }
  // d[n-1]: the last row still carries the pending update of the final step
  if (g == (n - 1) % G && t == 0) d[n - 1] = Al[(size_t)((n - 1) / G) * n + (n - 1)] - 2.0 * svp[n - 1] * swp[n - 1];
}

// Inner products of the reflectors of one group of the back-transformation (BT_RB = 4 reflectors applied together, section 4):
// group grp = reflectors k_b = n - 3 - 4 grp - b (b = 0 .. 3, the order they are applied in); G[8 grp + ..] = v_k1'v_k0, v_k2'v_k0,
// v_k2'v_k1, v_k3'v_k0, v_k3'v_k1, v_k3'v_k2.  One wave per group, spare workgroups of k_tql_leaves / k_eigf_pairs (they need only
// the reflectors, which are complete when those kernels start): off the critical path, no launch of their own.
constexpr int BT_RB = 4;
__device__ __forceinline__ void bt_gram_wave(const double* __restrict__ V, int n, int grp, double* __restrict__ G) {
  const int lane = threadIdx.x & 63, nref = n - 2;
  int k[BT_RB];
#pragma unroll
  for (int b = 0; b < BT_RB; ++b) k[b] = nref - 1 - BT_RB * grp - b;
  double c[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int j = lane; j < n; j += 64) {
    double v[BT_RB];
#pragma unroll
    for (int b = 0; b < BT_RB; ++b) v[b] = (k[b] >= 0 && j > k[b]) ? V[(size_t)k[b] * n + j] : 0.0;
    c[0] = fma(v[1], v[0], c[0]); c[1] = fma(v[2], v[0], c[1]); c[2] = fma(v[2], v[1], c[2]);
    c[3] = fma(v[3], v[0], c[3]); c[4] = fma(v[3], v[1], c[4]); c[5] = fma(v[3], v[2], c[5]);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) c[i] = wsum(c[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) G[(size_t)8 * grp + i] = c[i];
  }
}
__host__ __device__ inline int bt_groups(int n) { return (n - 2 + BT_RB - 1) / BT_RB; }

// ---------------------------------------------------------------------------------------------------------------------
// 2. leaves: implicit QL with eigenvectors (EISPACK tql2) on T[lo:hi, lo:hi] with the rank-one corrections of the splits
// taken off its two end diagonals; one wave per leaf, lane r owns row r of Z.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_tql_leaves(const double* __restrict__ d, const double* __restrict__ e, int n,
                                                   const int* __restrict__ bounds, double* __restrict__ lam,
                                                   double* __restrict__ Q, int64_t* stat, double* __restrict__ tnorm_out,
                                                   int nleaves, const double* __restrict__ V, double* __restrict__ G) {
  __shared__ double sd[LEAF + 1], se[LEAF + 1], Z[LEAF][LEAF + 1];
  __shared__ int sidx[LEAF];
  if ((int)blockIdx.x >= nleaves) { bt_gram_wave(V, n, (int)blockIdx.x - nleaves, G); return; }   // (see bt_gram_wave)
  const int lo = bounds[blockIdx.x], hi = bounds[blockIdx.x + 1], N = hi - lo, r = threadIdx.x;
  if (r < N) {
    double dv = d[lo + r];
    if (r == 0 && lo > 0) dv -= fabs(e[lo - 1]);
    if (r == N - 1 && hi < n) dv -= fabs(e[hi - 1]);
    sd[r] = dv;
    se[r] = (r < N - 1) ? e[lo + r] : 0.0;
    for (int c = 0; c < N; ++c) Z[r][c] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  // the QL iteration is the same scalar sequence in every lane (d, e in LDS); only the row of Z differs.
  // An off-diagonal is negligible relative to its two diagonal neighbours (LAPACK's test) OR relative to the norm of T:
  // on a kinship with a block of (numerically) zero eigenvalues -- d = 0 +- rounding, e = rounding noise -- the first test
  // alone never fires and the iteration burns its limit resolving noise (found by the fuzzer: K from ONE 0/1 marker,
  // eigenvalues {0 x 123, 60, 65}).  The second test costs eps * |T| of absolute accuracy, the backward-stable bound.
  // (the norm of the WHOLE tridiagonal matrix: on a rank-2 kinship the trailing part of T is the rounding residue of the
  // rounding residue ..., entries of 1e-163 whose squares underflow -- negligible against |T|, not against each other)
  double tnorm = 0.0;
  for (int j = threadIdx.x; j < n; j += 64) tnorm = fmax(tnorm, fmax(fabs(d[j]), (j < n - 1) ? fabs(e[j]) : 0.0));
  tnorm = wmax(tnorm);
  if (blockIdx.x == 0 && threadIdx.x == 0) *tnorm_out = tnorm;     // the merges' deflation tolerance is relative to it
  const double abs_small = EPS * tnorm;
  bool failed = false;
  for (int l = 0; l < N && !failed; ++l) {
    int iter = 0;
    for (;;) {
      // first m >= l with a negligible off-diagonal e[m] (m = N-1 if none): every lane tests its own index, one ballot
      const bool small = (r >= N - 1) || (fabs(se[r]) <= EPS * (fabs(sd[r]) + fabs(sd[r + 1]))) || (fabs(se[r]) <= abs_small);
      const unsigned long long mask = __ballot(small) >> l;
      const int m = l + (int)__builtin_ctzll(mask | (1ull << (N - 1 - l)));
      if (m == l) break;
      if (++iter > 80) { failed = true; break; }
      double gg = (sd[l + 1] - sd[l]) / (2.0 * se[l]);
      double rr = sqrt(fma(gg, gg, 1.0));
      gg = sd[m] - sd[l] + se[l] / (gg + copysign(rr, gg));
      double s = 1.0, c = 1.0, p = 0.0;
      int i = m - 1;
      bool under = false;
      // one wave: LDS accesses complete in program order, so the scalar recurrence (identical in every lane) needs no
      // barrier; r = sqrt(f^2 + g^2) and its reciprocal from one fast_rsqrt (kinship-scale data keeps f^2 + g^2 far
      // from over/underflow; an exact zero takes the reference algorithm's underflow branch)
      double ei = se[i], di = sd[i], di1 = sd[i + 1];
      // the lane's row of Z: column i + 1 is carried in a register from one rotation to the next (instead of a store -> load round trip
      // through LDS on every step; 8 us of the kernel's 334 at n = 500: the scalar recurrence is what bounds it), column i is requested one step early like d and e
      const int rz = r < N ? r : 0;
      double zc = Z[rz][m], z0n = Z[rz][i];
      for (; i >= l; --i) {
        const double f = s * ei, b = c * ei;
        const double h = fma(f, f, gg * gg);
        const double ein = (i > l) ? se[i - 1] : 0.0, din = (i > l) ? sd[i - 1] : 0.0;   // next iteration's operands, early
        const double z0 = z0n;
        z0n = (i > l) ? Z[rz][i - 1] : 0.0;
        if (h == 0.0) {
          if (r == 0) { se[i + 1] = 0.0; sd[i + 1] = di1 - p; se[m] = 0.0; }
          if (r < N) Z[r][i + 1] = zc;
          under = true;
          break;
        }
        const double ir = fast_rsqrt(h);
        rr = h * ir;
        if (r == 0) se[i + 1] = rr;
        s = f * ir; c = gg * ir;
        gg = di1 - p;
        rr = (di - gg) * s + 2.0 * c * b;
        p = s * rr;
        if (r == 0) sd[i + 1] = gg + p;
        gg = c * rr - b;
        if (r < N) Z[r][i + 1] = s * z0 + c * zc;
        zc = c * z0 - s * zc;
        di1 = di; di = din; ei = ein;
      }
      if (under) continue;
      if (r < N) Z[r][l] = zc;
      if (r == 0) { sd[l] -= p; se[l] = gg; se[m] = 0.0; }
    }
  }
  __syncthreads();
  if (failed && r == 0) stat[11] = -8;
#ifdef TQL_DIAG
  if (failed && r == 0) {
    printf("tql leaf %d [%d,%d) failed; tnorm %g\n", (int)blockIdx.x, lo, hi, tnorm);
    for (int j = 0; j < N; ++j) printf("  j %d d %.17g e %.17g  (orig d %.17g e %.17g)\n", j, sd[j], se[j], d[lo + j], (lo + j < n - 1) ? e[lo + j] : 0.0);
  }
#endif
  // ascending order (stable rank)
  if (r < N) {
    int rank = 0;
    const double v = sd[r];
    for (int j = 0; j < N; ++j) rank += (sd[j] < v) || (sd[j] == v && j < r);
    sidx[rank] = r;
  }
  __syncthreads();
  if (r < N) {
    lam[lo + r] = sd[sidx[r]];
    for (int c = 0; c < N; ++c) Q[(size_t)(lo + c) * n + lo + r] = Z[r][sidx[c]];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. divide and conquer.  Node (lo, mid, hi) joins the solved halves [lo, mid) and [mid, hi):
//      T_node = diag(T1', T2') + |beta| u u',  beta = e[mid-1],  u = e_(mid-1) + sign(beta) e_mid.
// Per-node scratch lives at offset lo of length-n arrays; K x K matrices on the node's diagonal block of n x n buffers.
// ---------------------------------------------------------------------------------------------------------------------
struct DcWs {
  int n;
  const double* e;          // off-diagonal of T
  const double* lamIn; double* lamOut;
  double* Qin; double* Qout;            // n x n, eigenvector c of a node = column c: Q[c*n + r]
  double* Dm; double* Wt;               // n x n
  double *dl, *zl, *zh, *defld, *lamnew, *rotc, *rots, *rho;
  int *colidx, *deflcol, *rota, *rotb, *posn, *posd, *info;   // info[4*node + {0: K, 1: ndefl, 2: nrot}]
  const int* nodes;         // [3*node + {lo, mid, hi}]
  const double* tnorm;      // max-norm of T (k_tql_leaves)
  int serial_deflate;       // 1: always take dlaed2's serial scan (BLMM_DC_DEFLATE=serial; tests compare it with the parallel path)
};

__global__ void __launch_bounds__(1024) k_dc_deflate(DcWs w) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int node = blockIdx.x, lo = w.nodes[3 * node], mid = w.nodes[3 * node + 1], hi = w.nodes[3 * node + 2];
  const int N = hi - lo, n = w.n, t = threadIdx.x, NT = blockDim.x;
  double* sd = sh;            // N
  double* sz = sd + N;        // N
  double* red = sz + N;       // 16
  int* ord = reinterpret_cast<int*>(red + 16);   // N
  const double beta = w.e[mid - 1];
  const double sgn = (beta < 0.0) ? -1.0 : 1.0;
  const double rho = 2.0 * fabs(beta);
  for (int j = t; j < N; j += NT) {
    sd[j] = w.lamIn[lo + j];
    const int col = lo + j;
    const double zr = (col < mid) ? w.Qin[(size_t)col * n + mid - 1] : sgn * w.Qin[(size_t)col * n + mid];
    sz[j] = zr * 0.7071067811865476;
  }
  __syncthreads();
  double dm = 0.0, zm = 0.0;
  for (int j = t; j < N; j += NT) { dm = fmax(dm, fabs(sd[j])); zm = fmax(zm, fabs(sz[j])); }
  dm = block_max(dm, red);
  zm = block_max(zm, red);
  // dlaed2's tolerance 8 eps max(|d|, |z|) is meant for a matrix scaled to unit norm (dstedc scales T first): z has unit norm
  // whatever the scale of T, so on the unscaled matrix the z term carries |T| (a kinship scaled by 1e-150 deflated every
  // merge completely: found by tools/fuzz_eig.py)
  const double tol = 8.0 * EPS * fmax(dm, zm * *w.tnorm);
  // sorted copies (ds, zs, ord) so that the serial scan below walks consecutive LDS words
  double* ds = red + 16 + (N + 1) / 2;   // past ord (N ints)
  double* zs = ds + N;
  for (int j = t; j < N; j += NT) {
    const double v = sd[j];
    int rank = 0;
    for (int i = 0; i < N; ++i) rank += (sd[i] < v) || (sd[i] == v && i < j);
    ord[rank] = j; ds[rank] = v; zs[rank] = sz[j];
  }
  __syncthreads();
  // Fast path, all threads: without a close pair of poles among the entries that survive the z test, dlaed2's scan keeps
  // exactly those entries, in sorted order, and deflates the others in sorted order -- two stream compactions.  One thread
  // walking the list costs ~360 cycles per entry (75 of the kernel's 97 us at N = 500); the serial scan below stays for
  // the merges that do rotate (clustered spectra), where its order of operations is LAPACK's.
  __shared__ int s_wcnt[2][16], s_any;
  int* cidx = reinterpret_cast<int*>(sd);          // surviving entries in sorted order (sd / sz are free after the sort)
  if (t == 0) s_any = 0;
  if (rho * zm > tol && N <= 2 * NT && !w.serial_deflate) {
    const int lane = t & 63, wv = t >> 6, nwv = NT >> 6;
    int keep[2], pre[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int sI = t + h * NT;
      keep[h] = (sI < N) && !(rho * fabs(zs[sI < N ? sI : 0]) <= tol);
      const unsigned long long bal = __ballot(keep[h]);
      pre[h] = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wcnt[h][wv] = __popcll(bal);
    }
    __syncthreads();
    int base[2] = {0, 0}, tot0 = 0, tot1 = 0;
    for (int q = 0; q < nwv; ++q) { if (q < wv) { base[0] += s_wcnt[0][q]; base[1] += s_wcnt[1][q]; } tot0 += s_wcnt[0][q]; tot1 += s_wcnt[1][q]; }
    base[1] += tot0;
    const int K = tot0 + tot1;
#pragma unroll
    for (int h = 0; h < 2; ++h) if (keep[h]) cidx[base[h] + pre[h]] = t + h * NT;
    __syncthreads();
    int pass = 0;
    for (int k = 1 + t; k < K; k += NT) {
      const int a = cidx[k - 1], b = cidx[k];
      const double tt = ds[b] - ds[a], zn = zs[b], zp = zs[a];
      pass |= fabs(tt * zn * zp) <= tol * fma(zn, zn, zp * zp);
    }
    if (pass) s_any = 1;
    __syncthreads();
    if (!s_any) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int sI = t + h * NT;
        if (sI >= N) continue;
        if (keep[h]) {
          const int k = base[h] + pre[h];
          w.dl[lo + k] = ds[sI]; w.zl[lo + k] = zs[sI]; w.colidx[lo + k] = lo + ord[sI];
        } else {
          const int q = sI - (base[h] + pre[h]);       // deflated entries before it = entries before it - kept before it
          w.defld[lo + q] = ds[sI]; w.deflcol[lo + q] = lo + ord[sI];
        }
      }
      if (t == 0) { w.info[4 * node] = K; w.info[4 * node + 1] = N - K; w.info[4 * node + 2] = 0; w.rho[node] = rho; }
      return;
    }
  }
  __syncthreads();
  if (t == 0) {
    int K = 0, nd = 0, nr = 0;
    if (rho * zm <= tol) {
      for (int s = 0; s < N; ++s) { w.defld[lo + nd] = ds[s]; w.deflcol[lo + nd] = lo + ord[s]; ++nd; }
    } else {
      // LAPACK dlaed2's scan in sorted order; (dp, zp, cp) = the pending candidate (value, z, column)
      int havep = 0, cp = 0;
      double dp = 0.0, zp = 0.0;
      // the next entry is requested before the current one is worked on (every path of the body ends in a branch; read at
      // the top of the loop, each iteration began with an LDS round trip)
      double dnx = ds[0], znx = zs[0];
      int cnx = ord[0];
      for (int s = 0; s < N; ++s) {
        const double dn = dnx, zn = znx;
        const int cn = cnx;
        if (s + 1 < N) { dnx = ds[s + 1]; znx = zs[s + 1]; cnx = ord[s + 1]; }
        if (rho * fabs(zn) <= tol) { w.defld[lo + nd] = dn; w.deflcol[lo + nd] = lo + cn; ++nd; continue; }
        if (!havep) { havep = 1; dp = dn; zp = zn; cp = cn; continue; }
        const double tt = dn - dp, tau2 = fma(zn, zn, zp * zp);
        // |t c s| <= tol with c, s normalised by tau, tested without the division (the common case keeps both)
        if (fabs(tt * zn * zp) <= tol * tau2) {
          const double tau = sqrt(tau2);
          const double cs = zn / tau, sn = -zp / tau;
          w.rota[lo + nr] = lo + cp; w.rotb[lo + nr] = lo + cn; w.rotc[lo + nr] = cs; w.rots[lo + nr] = sn; ++nr;
          const double dpn = dp * cs * cs + dn * sn * sn;
          const double dnn = dp * sn * sn + dn * cs * cs;
          w.defld[lo + nd] = dpn; w.deflcol[lo + nd] = lo + cp; ++nd;
          dp = dnn; zp = tau; cp = cn;
        } else {
          w.dl[lo + K] = dp; w.zl[lo + K] = zp; w.colidx[lo + K] = lo + cp; ++K;
          dp = dn; zp = zn; cp = cn;
        }
      }
      if (havep) { w.dl[lo + K] = dp; w.zl[lo + K] = zp; w.colidx[lo + K] = lo + cp; ++K; }
    }
    w.info[4 * node] = K; w.info[4 * node + 1] = nd; w.info[4 * node + 2] = nr;
    w.rho[node] = rho;
  }
}

// Givens rotations of the close-pole deflations, applied to the node's rows of Qin in list order (rows are independent)
__device__ __forceinline__ void dc_rot(const DcWs& w, int by) {
  const int node = blockIdx.x, lo = w.nodes[3 * node], hi = w.nodes[3 * node + 2], n = w.n;
  const int nr = w.info[4 * node + 2];
  const int r = lo + by * 256 + threadIdx.x;
  if (nr == 0 || r >= hi) return;
  for (int q = 0; q < nr; ++q) {
    const int a = w.rota[lo + q], b = w.rotb[lo + q];
    const double c = w.rotc[lo + q], s = w.rots[lo + q];
    const double x = w.Qin[(size_t)a * n + r], y = w.Qin[(size_t)b * n + r];
    w.Qin[(size_t)a * n + r] = c * x + s * y;
    w.Qin[(size_t)b * n + r] = c * y - s * x;
  }
}

// secular equation 1 + rho sum_j z_j^2 / (d_j - lam) = 0, root i in (d_i, d_(i+1)), one wave per root.  The root is
// found relative to the nearer pole (the differences d_j - lam keep full relative accuracy), Newton steps safeguarded by
// the bracket, bisection otherwise (the function is monotone between two poles: the bracket always holds the root).
#ifdef SEC_DIAG
__device__ unsigned int g_sec_hist[64];   // iterations per root (diagnostic build; read by launch_eig_dc)
#endif
__device__ __forceinline__ void dc_secular(const DcWs& w, int by, double* sh) {
  const int node = blockIdx.x, lo = w.nodes[3 * node], n = w.n;
  const int K = w.info[4 * node];
  if ((int)by * 4 >= K) return;
  double* dl = sh;
  double* z2 = sh + K;
  for (int j = threadIdx.x; j < K; j += blockDim.x) { dl[j] = w.dl[lo + j]; const double z = w.zl[lo + j]; z2[j] = z * z; }
  __syncthreads();
  const int lane = threadIdx.x & 63, i = by * 4 + (threadIdx.x >> 6);
  if (i >= K) return;
  const double rho = w.rho[node];
  int org;
  double a, b;
  if (i < K - 1) {
    const double left = dl[i], mid = 0.5 * (dl[i + 1] - left);
    double f = 0.0;
    for (int j = lane; j < K; j += 64) f += z2[j] / ((dl[j] - left) - mid);
    f = 1.0 + rho * wsum(f);
    if (f > 0.0) { org = i; a = 0.0; b = mid; } else { org = i + 1; a = -mid; b = 0.0; }
    if (f == 0.0) { org = i; a = mid; b = mid; }
  } else {
    double s = 0.0;
    for (int j = lane; j < K; j += 64) s += z2[j];
    org = K - 1; a = 0.0; b = rho * wsum(s);
  }
  const double dorg = dl[org];
  double tcur = 0.5 * (a + b);
  // Bunch-Nielsen-Sorensen iteration: psi (poles left of the root) and phi (poles right of it) are each replaced by
  // r + s / (pole - t) matching value and slope at the iterate; the root of the resulting two-pole equation (a quadratic)
  // is the next iterate: monotone and quadratic from anywhere inside the interval.  The sign of f keeps a bracket, and a
  // candidate outside it (rounding, or the degenerate ends) falls back to bisection.
  const double poleL = dl[i] - dorg, poleR = (i < K - 1) ? dl[i + 1] - dorg : 0.0;
#ifdef SEC_DIAG
  int sec_it = 0;
#endif
  for (int it = 0; it < 100 && a < b; ++it) {
#ifdef SEC_DIAG
    sec_it = it + 1;
#endif
    double ps = 0.0, psp = 0.0, ph = 0.0, php = 0.0;
    for (int j = lane; j < K; j += 64) {
      const double q = 1.0 / ((dl[j] - dorg) - tcur);
      const double zq = z2[j] * q;
      if (j <= i) { ps += zq; psp = fma(zq, q, psp); } else { ph += zq; php = fma(zq, q, php); }
    }
    ps = rho * wsum(ps); psp = rho * wsum(psp); ph = rho * wsum(ph); php = rho * wsum(php);
    const double f = 1.0 + ps + ph;
    // LAPACK dlaed4's test: |f| within the rounding error of its own evaluation (here in units of f = rho w: the sum of the
    // terms' magnitudes ph - ps, the constant, and the slope times the offset).  Without it one root in ten went on bisecting
    // on the noise of the sign of f for 25-54 iterations (4-8 otherwise), and the slowest root is the kernel's duration.
    if (fabs(f) <= EPS * (16.0 * (ph - ps) + 2.0 + 3.0 * fabs(tcur) * (psp + php))) break;
    if (f > 0.0) b = tcur; else a = tcur;
    double tn;
    const double dL = poleL - tcur;
    const double sps = psp * dL * dL, rps = ps - psp * dL;
    if (i < K - 1) {
      const double dR = poleR - tcur;
      const double sph = php * dR * dR, rph = ph - php * dR;
      const double c = 1.0 + rps + rph;
      const double a2 = c, a1 = -(c * (poleL + poleR) + sps + sph), a0 = c * poleL * poleR + sps * poleR + sph * poleL;
      const double disc = a1 * a1 - 4.0 * a2 * a0;
      tn = tcur;
      if (disc >= 0.0) {
        const double qq = -0.5 * (a1 + copysign(sqrt(disc), a1));
        const double r1 = (a2 != 0.0) ? qq / a2 : a, r2 = (qq != 0.0) ? a0 / qq : a;
        tn = (a < r1 && r1 < b) ? r1 : r2;
      }
    } else {
      const double c = 1.0 + rps;
      tn = poleL + sps / c;
    }
    if (!(a < tn && tn < b)) tn = 0.5 * (a + b);
    if (tn == tcur || b - a <= 2.0 * EPS * fmax(fabs(a), fabs(b)) || fabs(tn - tcur) <= EPS * fabs(tn)) { tcur = tn; break; }
    tcur = tn;
  }
#ifdef SEC_DIAG
  if (lane == 0) atomicAdd(&g_sec_hist[sec_it < 63 ? sec_it : 63], 1u);
#endif
  if (lane == 0) w.lamnew[lo + i] = dorg + tcur;
  double* drow = w.Dm + (size_t)(lo + i) * n + lo;
  for (int j = lane; j < K; j += 64) drow[j] = (dl[j] - dorg) - tcur;
}

// Gu-Eisenstat: the z-hat for which the computed roots are the exact eigenvalues,
//   zh_j^2 = | prod_i (d_j - lam_i) / prod_(i != j) (d_j - d_i) |,  sign from z_j.   One wave per pole j.
__device__ __forceinline__ void dc_zhat(const DcWs& w, int by) {
  const int node = blockIdx.x, lo = w.nodes[3 * node], n = w.n;
  const int K = w.info[4 * node];
  const int lane = threadIdx.x & 63, j = by * 4 + (threadIdx.x >> 6);
  if (j >= K) return;
  const double dj = w.dl[lo + j];
  double prod = 1.0;
  for (int i = lane; i < K; i += 64) {
    const double dij = w.Dm[(size_t)(lo + i) * n + lo + j];
    prod *= (i == j) ? dij : dij / (dj - w.dl[lo + i]);
  }
  prod = wprod(prod);
  if (lane == 0) w.zh[lo + j] = copysign(sqrt(fabs(prod)), w.zl[lo + j]);
}

// eigenvector of root i in the basis of the kept columns: Wt[i][j] = zh_j / (d_j - lam_i), normalised.  One wave per root.
__device__ __forceinline__ void dc_wt(const DcWs& w, int by) {
  const int node = blockIdx.x, lo = w.nodes[3 * node], n = w.n;
  const int K = w.info[4 * node];
  const int lane = threadIdx.x & 63, i = by * 4 + (threadIdx.x >> 6);
  if (i >= K) return;
  const double* drow = w.Dm + (size_t)(lo + i) * n + lo;
  double* wrow = w.Wt + (size_t)(lo + i) * n + lo;
  double s = 0.0;
  for (int j = lane; j < K; j += 64) { const double v = w.zh[lo + j] / drow[j]; wrow[j] = v; s = fma(v, v, s); }
  s = 1.0 / sqrt(wsum(s));
  for (int j = lane; j < K; j += 64) wrow[j] *= s;
}

// positions of the node's N eigenvalues (K new roots, then the deflated ones) in ascending order; sorted values out
__device__ __forceinline__ void dc_finalize(const DcWs& w, int by, double* sh) {
  // block `by` of a node ranks the entries [256 by, 256 by + 256) against all N values (staged in LDS by every block)
  const int node = blockIdx.x, lo = w.nodes[3 * node], hi = w.nodes[3 * node + 2], N = hi - lo;
  const int K = w.info[4 * node], t = threadIdx.x, NT = blockDim.x;
  if (by * 256 >= N) return;                                   // workgroup-uniform
  double* val = sh;
  for (int j = t; j < N; j += NT) val[j] = (j < K) ? w.lamnew[lo + j] : w.defld[lo + j - K];
  __syncthreads();
  const int j = by * 256 + t;
  if (j >= N) return;
  const double v = val[j];
  int rank = 0;
  for (int i = 0; i < N; ++i) rank += (val[i] < v) || (val[i] == v && i < j);
  if (j < K) w.posn[lo + j] = lo + rank; else w.posd[lo + j - K] = lo + rank;
  w.lamOut[lo + rank] = v;
}

__device__ __forceinline__ void dc_copy(const DcWs& w, int by, int bz, int nz) {
  const int node = blockIdx.x, lo = w.nodes[3 * node], hi = w.nodes[3 * node + 2], n = w.n;
  const int nd = w.info[4 * node + 1];
  const int r = lo + by * 256 + threadIdx.x;
  if (r >= hi) return;
  for (int q = bz; q < nd; q += nz)
    w.Qout[(size_t)w.posd[lo + q] * n + r] = w.Qin[(size_t)w.deflcol[lo + q] * n + r];
}

// The merge's small steps, three launches instead of six (a launch is ~4.7 us, six levels of them per decomposition): blocks
// past the first kernel's range do the work of an independent second one.
//   secular roots  |  Givens rotations of the deflated pairs on Qin (read again only by the copy and the GEMM)
//   z-hat          |  ranks of the new and the deflated eigenvalues (needs the roots, not z-hat)
//   eigenvectors   |  deflated columns to their places (needs the ranks)
constexpr int DC_COPY_Z = 16;
__global__ void __launch_bounds__(256) k_dc_secular_rot(DcWs w, int nsec) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  if ((int)blockIdx.y < nsec) dc_secular(w, (int)blockIdx.y, sh); else dc_rot(w, (int)blockIdx.y - nsec);
}
__global__ void __launch_bounds__(256) k_dc_zhat_fin(DcWs w, int nz) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  if ((int)blockIdx.y < nz) dc_zhat(w, (int)blockIdx.y); else dc_finalize(w, (int)blockIdx.y - nz, sh);
}
__global__ void __launch_bounds__(256) k_dc_wt_copy(DcWs w, int nw) {
  if ((int)blockIdx.y < nw) { dc_wt(w, (int)blockIdx.y); return; }
  const int c = (int)blockIdx.y - nw;
  dc_copy(w, c / DC_COPY_Z, c % DC_COPY_Z, DC_COPY_Z);
}

// Qout[:, posn[i]] = sum_j Wt[i][j] Qin[:, colidx[j]]  on the f64 matrix cores: D (16 roots x 16 rows of Q) per block,
// A = Wt (root i, k = j), B = Qin' (k = j, row r).  Inside a trip of 16 k the lane group g = lane >> 4 takes
// k = k0 + 4 g + s at MFMA step s (any fixed assignment is a valid order of the contraction): a lane reads 4 consecutive
// Wt entries, and 16 lanes read 16 consecutive rows of one column of Qin.
typedef double d4v __attribute__((ext_vector_type(4)));
template <int MB, int NB>
__global__ void __launch_bounds__(256) k_dc_gemm(DcWs w, int tiles_r) {
  __shared__ __attribute__((aligned(16))) int s_col[2048];         // the node's column list (K <= n <= 2048)
  const int node = blockIdx.x, lo = w.nodes[3 * node], hi = w.nodes[3 * node + 2], n = w.n;
  const int K = w.info[4 * node];
  const int tile_i = blockIdx.y / tiles_r, tile_r = blockIdx.y % tiles_r;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = tile_i * (32 * MB) + (wave >> 1) * (16 * MB);     // roots
  const int r0 = lo + tile_r * (32 * NB) + (wave & 1) * (16 * NB); // rows of Q
  if (tile_i * (32 * MB) >= K || lo + tile_r * (32 * NB) >= hi) return;     // (uniform over the workgroup)
  const int c16 = lane & 15, g = lane >> 4;
  // The K loop had two DEPENDENT global round trips per step of 16 -- the column index, then the column of Q it names -- each
  // inside its own `(k < K) ? load : 0` branch: ~1.5 us per step against 0.2 us of MFMAs (30 us per level at n = 1000).  Now the
  // column list sits in LDS, every load takes a clamped address (the W operand is zeroed past K, rows / roots past the edge are
  // computed on a copy and not stored), and the operands of steps it+2 / it+3 are in flight while step it / it+1 multiply.
  for (int k = threadIdx.x; k < K; k += 256) s_col[k] = w.colidx[lo + k];
  __syncthreads();
  d4v acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (d4v){0, 0, 0, 0};
  const double* wr[MB];
  const double* qr[NB];
#pragma unroll
  for (int a = 0; a < MB; ++a) { const int i = i0 + 16 * a + c16; wr[a] = w.Wt + (size_t)(lo + (i < K ? i : K - 1)) * n + lo; }
#pragma unroll
  for (int b = 0; b < NB; ++b) { const int r = r0 + 16 * b + c16; qr[b] = w.Qin + (r < hi ? r : hi - 1); }
  struct __attribute__((packed, aligned(8))) D4 { double x[4]; };   // a lane's four consecutive k of W: one 32-byte access
  auto load = [&](int it, double (&av)[MB][4], double (&bv)[NB][4]) {
    if (16 * it + 16 <= K) {                                          // (uniform) a whole step: no clamps
      const int kb = 16 * it + 4 * g;
      const int4 cl = *reinterpret_cast<const int4*>(&s_col[kb]);
      const int col[4] = {cl.x, cl.y, cl.z, cl.w};
#pragma unroll
      for (int a = 0; a < MB; ++a) {
        const D4 x = *reinterpret_cast<const D4*>(wr[a] + kb);
#pragma unroll
        for (int s = 0; s < 4; ++s) av[a][s] = x.x[s];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int b = 0; b < NB; ++b) bv[b][s] = qr[b][(size_t)col[s] * n];
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 16 * it + 4 * g + s, kc = (k < K) ? k : K - 1;
        const int col = s_col[kc];
#pragma unroll
        for (int a = 0; a < MB; ++a) { const double x = wr[a][kc]; av[a][s] = (k < K) ? x : 0.0; }
#pragma unroll
        for (int b = 0; b < NB; ++b) bv[b][s] = qr[b][(size_t)col * n];
      }
    }
  };
  auto mult = [&](const double (&av)[MB][4], const double (&bv)[NB][4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a][s], bv[b][s], acc[a][b], 0, 0, 0);
  };
  const int KT = (K + 15) >> 4;
  double a0[MB][4], b0[NB][4], a1[MB][4], b1[NB][4];
  load(0, a0, b0);
  load(1, a1, b1);                                                  // (a step past the end loads valid addresses and multiplies zeros)
  for (int it = 0; it < KT; it += 2) {
    mult(a0, b0); load(it + 2, a0, b0);
    mult(a1, b1); load(it + 3, a1, b1);
  }
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = i0 + 16 * a + g + 4 * reg;
      if (i >= K) continue;
      double* dst = w.Qout + (size_t)w.posn[lo + i] * n;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int r = r0 + 16 * b + c16;
        if (r < hi) dst[r] = acc[a][b][reg];
      }
    }
}

// "badness" of a check as a monotone bit pattern: v / bound, NaN or negative counted as +Inf
__device__ __forceinline__ void eigf_note(int64_t* stat, double v, double bound) {
  const double q = (v >= 0.0 && bound > 0.0) ? v / bound : ((v == 0.0) ? 0.0 : INFINITY);
  atomicMax(reinterpret_cast<unsigned long long*>(&stat[ST_EIG_BAD]), (unsigned long long)__double_as_longlong(q));
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. back-transformation U = H_0 H_1 ... H_(n-3) Z: reflectors applied in reverse order, each wave owns CPW columns of Z
// in registers (rows lane, lane + 64, ...); the reflectors are staged in LDS for the workgroup, BT_RB = 4 at a time.
//
// The four reflectors of a group are applied TOGETHER (the compact WY form written out for four): with d_b = v_b'z taken from the
// column as it stands BEFORE the group,
//   s_0 = tau_0 d_0,   s_b = tau_b (d_b - sum_(a < b) s_a v_b'v_a),   z <- z - sum_b s_b v_b
// is H_3 H_2 H_1 H_0 z exactly; the four dot products and their wave reductions are independent chains that overlap, where the
// one-at-a-time form of rounds 2-3 ran LDS read -> NR dependent FMAs -> reduction -> update four times in sequence.  The inner
// products v_b'v_a come from bt_gram_wave.
//
// NO conditional memory access in the loops: every `(r < n) ? load : 0` of the earlier form became its own basic block (exec-mask
// branch, s_waitcnt at the join) -- 60+ of them per group at n = 1000.  Reflectors are staged at a padded length LDR = 64 NR with
// zeros beyond n (and at j <= k), global loads take a clamped index and a select, Z's LDS copy carries 128 doubles of padding.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int div_small(int e, int n, float rn) {      // e / n for 0 <= e < 2^22, n <= 2048, rn = 1.0f / n
  int q = (int)((float)e * rn);
  q -= (q * n > e);
  q += ((q + 1) * n <= e);
  return q;
}
// LDS bytes of k_backtransform<NR, ..>: plain (the multi-workgroup solver), with the orthogonality check (n <= 124), and with
// everything staged (check + all reflectors, tau, inner products: where it fits)
__host__ __device__ inline size_t bt_lds_plain(int NR) { return sizeof(double) * 2 * ((size_t)BT_RB * 64 * NR + 16); }
__host__ __device__ inline size_t bt_lds_chk(int NR, int n) { return bt_lds_plain(NR) + sizeof(double) * ((size_t)n * (n | 1) + 128); }
__host__ __device__ inline size_t bt_lds_staged(int NR, int n) {
  return sizeof(double) * ((size_t)n * (n | 1) + 128 + (size_t)BT_RB * bt_groups(n) * 64 * NR + 16 * (size_t)bt_groups(n));
}

template <int NR, int CPW, int NT>
__global__ void __launch_bounds__(NT) k_backtransform(const double* __restrict__ V, const double* __restrict__ tau,
                                                       const double* __restrict__ G, int n, double* __restrict__ Z, const double* __restrict__ Zc_chk,
                                                       int64_t* __restrict__ stat_chk, int stage_v) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  constexpr int RB = BT_RB, LDR = 64 * NR;
  constexpr size_t BUF = (size_t)RB * LDR + 16;     // one group: RB padded reflectors, then tau_0..3 and the six inner products
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = NT >> 6;
  const int c0 = (blockIdx.x * nwave + wave) * CPW;
  const int nref = n - 2, ngrp = (nref + RB - 1) / RB;
  const bool staged = NR == 2 && Zc_chk && stage_v;  // (uniform over the grid)
  // LDS map.  plain: [2 BUF].  check: [2 BUF][Zs].  staged: [Zs][Vs: RB ngrp rows of LDR][Ts: 16 per group]
  double* Zs = sh + (staged ? 0 : 2 * BUF);
  const int lz = n | 1;                              // odd column stride: lane <-> column reads spread over the banks
  double* Vs = Zs + (size_t)n * lz + 128;
  double* Ts = Vs + (size_t)RB * ngrp * LDR;
  double z[CPW][NR];
#ifdef EIGF_PROF      // tools/eigf_prof.sh: cycles of the phases (first and last workgroup, wave 0)
  long long bpf[6] = {0, 0, 0, 0, 0, 0}, bpt0 = __builtin_amdgcn_s_memtime();
#define BPSTAMP(i) do { const long long t1__ = __builtin_amdgcn_s_memtime(); bpf[i] += t1__ - bpt0; bpt0 = t1__; } while (0)
#else
#define BPSTAMP(i) do { } while (0)
#endif
  // Behind k_eigf_pairs (NR = 2, CPW = 1, n <= 128): the whole uncorrected Z (Zc_chk, column-major, written by workgroups on
  // every XCD: a dependent chain of reads from it would pay the cross-XCD round trip each time -- 52 us for this kernel, 35 with
  // the staging) is copied into LDS with one batch of loads.  Row a of E = Z'Z - I first -- its largest entry / 3e-8 ->
  // stat[ST_EIG_BAD] --, then the symmetric first-order correction z_a <- z_a - (1/2) sum_b E_ab z_b: every workgroup corrects
  // from the same uncorrected Z, together Z (I - E / 2), orthogonal to ~|E|^2.  On kinships |E| is ~4e-14 and the correction only
  // takes the vectors from there to rounding level; it is what lets near-repeated eigenvalues through (|E| up to 3e-8).
  if (Zc_chk) {
    // Loads in batches of 16 per thread: Zc_chk and V were written by workgroups on other XCDs, a load costs 1-2 us, and batches of
    // four (the unrolled loop of the earlier form) paid that 15 times: 13 of this kernel's 34 us at n = 79.
    const int nn = n * n;
    const float rn = 1.0f / (float)n;
    const int zpad = (n - 1) * lz + n;                 // 128 doubles of padding behind the last column (zeroed)
    const int ziters = (nn + 128 + NT - 1) / NT;
    constexpr int SB = 16;                             // loads in flight per thread (a fully unrolled copy -- 61 + 44 loads -- spent
                                                       // more time fetching its straight-line code, cold on every launch, than it saved)
    for (int it0 = 0; it0 < ziters; it0 += SB) {
      double rg[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) { const int e = threadIdx.x + NT * (it0 + u); rg[u] = Zc_chk[(e < nn) ? e : 0]; }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int e = threadIdx.x + NT * (it0 + u), ee = (e < nn) ? e : 0;
        const int cb = div_small(ee, n, rn);
        Zs[(e < nn) ? cb * lz + (ee - cb * n) : zpad + min(e - nn, 127)] = (e < nn) ? rg[u] : 0.0;      // (past the end: zeros into the padding)
      }
    }
    // staged (the launcher: it fits, n <= 90): ALL reflectors (row RB g + b = reflector n - 3 - RB g - b, zero rows past the first
    // reflector), their tau and the groups' inner products go to LDS as well -- the group loop below then touches neither global
    // memory (a group of four two-register reflectors is shorter than the L2 round trip of its successor's prefetch) nor a barrier.
    // (Issuing these loads ahead of the check and the correction, which need only Z, and storing them afterwards measured no
    // better: the copy is bound by the misses a CU can keep in flight against ~1.5 us of cross-XCD latency, not by their order.)
    if (staged) {
      const int vtot = RB * ngrp * LDR;                // (a multiple of NT)
      for (int e0 = 0; e0 < vtot; e0 += NT * SB) {
        double rg[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int e = e0 + threadIdx.x + NT * u;
          const int i = e / LDR, r = e - i * LDR, kk = nref - 1 - i;
          rg[u] = V[(e < vtot && kk >= 0 && r > kk && r < n) ? (size_t)kk * n + r : 0];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int e = e0 + threadIdx.x + NT * u;
          const int i = e / LDR, r = e - i * LDR, kk = nref - 1 - i;
          if (e < vtot) Vs[e] = (kk >= 0 && r > kk && r < n) ? rg[u] : 0.0;
        }
      }
      for (int e = threadIdx.x; e < 16 * ngrp; e += NT) {
        const int gg = e >> 4, i = e & 15, kk = nref - 1 - RB * gg - i;
        const bool isT = i < RB && kk >= 0, isG = i >= RB && i < RB + 6;
        const double tv = tau[isT ? kk : 0], gv = G[isG ? (size_t)8 * gg + (i - RB) : 0];
        Ts[e] = isT ? tv : (isG ? gv : 0.0);
      }
    }
    __syncthreads();
    BPSTAMP(0);
    const int a = (c0 < n) ? c0 : n - 1;               // (a wave past the last column works on a copy of it and stores nothing)
    const double* za = Zs + (size_t)a * lz;
    const double* zb0 = Zs + (size_t)((lane < n) ? lane : 0) * lz;
    const double* zb1 = Zs + (size_t)((lane + 64 < n) ? lane + 64 : 0) * lz;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 4
    for (int i = 0; i < n; ++i) { const double v = za[i]; acc0 = fma(v, zb0[i], acc0); acc1 = fma(v, zb1[i], acc1); }
    const double e0 = (lane < n) ? acc0 - ((lane == a) ? 1.0 : 0.0) : 0.0;
    const double e1 = (lane + 64 < n) ? acc1 - ((lane + 64 == a) ? 1.0 : 0.0) : 0.0;
    double bad = fmax(fabs(e0), fabs(e1));
    if (!(e0 == e0) || !(e1 == e1)) bad = INFINITY;
    bad = wmax(bad);
    if (lane == 0 && c0 < n) eigf_note(stat_chk, bad, 3e-8);
    BPSTAMP(1);
    double c0v = 0.0, c1v = 0.0;
    auto corr = [&](int b) {
      const double eab = lane_bcast((b < 64) ? e0 : e1, b & 63);     // (b is wave-uniform)
      const double* zb = Zs + (size_t)b * lz;
      c0v = fma(eab, zb[lane], c0v);                                 // (lanes past n read the next column / the padding; masked below)
      c1v = fma(eab, zb[lane + 64], c1v);
    };
    int b4 = 0;                                                      // (unrolled by hand: v_readlane is convergent, the optimizer
    for (; b4 + 4 <= n; b4 += 4) { corr(b4); corr(b4 + 1); corr(b4 + 2); corr(b4 + 3); }   //  will not unroll a run-time trip count around it)
    for (; b4 < n; ++b4) corr(b4);
    const double z0 = (lane < n) ? fma(-0.5, c0v, za[lane]) : 0.0;
    const double z1 = (lane + 64 < n) ? fma(-0.5, c1v, za[lane + 64]) : 0.0;
    z[0][0] = z0;
    if constexpr (NR > 1) z[0][1] = z1;
  } else {
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const int r = lane + 64 * q;
        const bool ok = c0 + c < n && r < n;
        const double val = Z[ok ? (size_t)(c0 + c) * n + r : 0];
        z[c][q] = ok ? val : 0.0;
      }
  }
  BPSTAMP(2);
  // Streamed form (not staged): group g = reflectors k = nref-1-RB*g ... (descending) is read from global memory two iterations
  // before its use (into registers) and written to LDS one iteration before, so neither the L2 round trip nor the LDS write sits
  // between two barriers.  One barrier per group.
  constexpr int PF = (LDR + NT - 1) / NT;         // elements per thread of one padded reflector
  double pre[RB][PF];
  double pret = 0.0;              // threads 0 .. 9 (of every 16): tau_b / inner product of the group, staged with the vectors (a
                                  // scalar load of tau[k] next to its use put an L2 round trip into every reflector's chain)
  auto fetch = [&](int grp) {
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const int k = nref - 1 - RB * grp - b;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int j = threadIdx.x + NT * u;
        const bool ok = k >= 0 && j > k && j < n;
        const double val = V[ok ? (size_t)k * n + j : 0];
        pre[b][u] = ok ? val : 0.0;
      }
    }
    const int i = threadIdx.x & 15, kk = nref - 1 - RB * grp - i;
    const bool isT = i < RB && kk >= 0, isG = i >= RB && i < RB + 6 && RB * grp < nref;
    const double tv = tau[isT ? kk : 0], gv = G[isG ? (size_t)8 * grp + (i - RB) : 0];
    pret = isT ? tv : (isG ? gv : 0.0);
  };
  auto stash = [&](double* dst) {
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int u = 0; u < PF; ++u) { const int j = threadIdx.x + NT * u; if (LDR % NT == 0 || j < LDR) dst[(size_t)b * LDR + j] = pre[b][u]; }
    if (threadIdx.x < 16) dst[(size_t)RB * LDR + threadIdx.x] = pret;
  };
  if (!staged) {
    fetch(0); stash(sh);
    fetch(1);
    __syncthreads();
  }
  double vr[RB][NR], tgr[10];                        // the group's reflectors (zero past n, and whole zero rows past the first
                                                     // reflector, tau = 0: no special case), tau_0..3, v_1'v_0, v_2'v_0, v_2'v_1, v_3'v_0, v_3'v_1, v_3'v_2
  auto load_group = [&](const double* vb, const double* tg, double (&v)[RB][NR], double (&t)[10]) {
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int q = 0; q < NR; ++q) v[b][q] = vb[(size_t)b * LDR + lane + 64 * q];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = tg[i];
  };
  if (staged) load_group(Vs, Ts, vr, tgr);
  for (int grp = 0; grp < ngrp; ++grp) {
    if (!staged) {
      const double* vb = sh + (size_t)(grp & 1) * BUF;
      stash(sh + (size_t)((grp + 1) & 1) * BUF);      // group grp+1 (fetched during the previous iteration)
      fetch(grp + 2);                                 // lands during this iteration and the next
      load_group(vb, vb + (size_t)RB * LDR, vr, tgr);
    }
    double vn[(NR == 2) ? RB : 1][NR], tn[10];        // staged: the next group's operands are read while this one is applied
    if constexpr (NR == 2) {
      if (staged) { const int gn = min(grp + 1, ngrp - 1); load_group(Vs + (size_t)gn * RB * LDR, Ts + 16 * gn, vn, tn); }
    }
    const double t0 = tgr[0], t1 = tgr[1], t2 = tgr[2], t3 = tgr[3];
    const double c10 = tgr[4], c20 = tgr[5], c21 = tgr[6], c30 = tgr[7], c31 = tgr[8], c32 = tgr[9];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      double d[RB];
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int q = 0; q < NR; q += 2) { a0 = fma(vr[b][q], z[c][q], a0); if (q + 1 < NR) a1 = fma(vr[b][q + 1], z[c][q + 1], a1); }
        d[b] = a0 + a1;
      }
      wsum4(d[0], d[1], d[2], d[3]);
      const double s0 = t0 * d[0];
      const double s1 = t1 * fma(-s0, c10, d[1]);
      const double s2 = t2 * fma(-s1, c21, fma(-s0, c20, d[2]));
      const double s3 = t3 * fma(-s2, c32, fma(-s1, c31, fma(-s0, c30, d[3])));
#pragma unroll
      for (int q = 0; q < NR; ++q)
        z[c][q] = fma(-s3, vr[3][q], fma(-s2, vr[2][q], fma(-s1, vr[1][q], fma(-s0, vr[0][q], z[c][q]))));
    }
    if constexpr (NR == 2) {
      if (staged) {
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
          for (int q = 0; q < NR; ++q) vr[b][q] = vn[b][q];
#pragma unroll
        for (int i = 0; i < 10; ++i) tgr[i] = tn[i];
      }
    }
    if (!staged) __syncthreads();
  }
  BPSTAMP(3);
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int r = lane + 64 * q;
      if (c0 + c < n && r < n) Z[(size_t)(c0 + c) * n + r] = z[c][q];
    }
  BPSTAMP(4);
#ifdef EIGF_PROF
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
    printf("backtransform prof wg %d n %d staged %d: stage %lld | check %lld | correct %lld | %d groups %lld | store %lld cycles\n", (int)blockIdx.x, n, (int)staged,
           bpf[0], bpf[1], bpf[2], ngrp, bpf[3], bpf[4]);
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// 5. n <= 124: the tridiagonalisation inside ONE workgroup, matrix resident in LDS (the BXD case, n = 79, where the
// eigen-decomposition is a serial front of the call that sharding traits over GPUs does not shorten).  Round 2 continued in the
// same workgroup with QL leaves, the merge tree and the back-transformation (`k_eig_small`: 0.89 ms, slower than the Jacobi);
// round 3 keeps only the reduction and hands T to the parallel fast path of section 6.
// ---------------------------------------------------------------------------------------------------------------------
struct SmallWs {   // LDS carve-up of the reduction (doubles); M2 optional (zeroed when given); svp / swp: two parities of n each
  double *M1, *M2, *sx, *sv, *svp, *swp, *part, *d, *e, *tau, *red;
};

// Householder tridiagonalisation of the n x n matrix A inside ONE workgroup of 1024 threads (n <= 128; k_eigf_reduce): A is
// copied to w.M1 (LDS, full symmetric storage), on return w.d / w.e / w.tau hold the diagonal, the
// off-diagonal and the reflector scalars of T = H' A H and Vg[k * n + j] the reflector vectors (v[k + 1] = 1).
//
// A step is two phases with one barrier behind each (round 4; rounds 2-3 applied the rank-2 update inside the product pass -- five
// LDS operations per element on the critical path -- and left fifteen waves idle during wave 0's vector work):
//   pass  all 16 waves: y_j = sum_(i > k) M1[i][j] v_k[i], column form (thread (j, rs) sums its row subset: no cross-lane
//         reduction), M1 carrying the updates of the steps <= k-2 only: two LDS operations per element.
//   B     wave 0: p = tau (y - v_(k-1) (w_(k-1)'v_k) - w_(k-1) (v_(k-1)'v_k)) -- the pending pair's correction, as in k_sytrd --,
//         w_k, row k+1 with the pending update applied by hand, the next column and its reflector v_(k+1), the two inner
//         products the next correction needs (one wsum4 with |x|^2); v_(k-1), w_(k-1), v_k, tau live in its registers.
//         waves 1-15 MEANWHILE: the rank-2 update of step k-1 on rows and columns >= k+2 (the pair is read from the parity
//         buffer wave 0 wrote a step ago; wave 0 writes this step's pair to the other one): off the critical path.
__device__ void small_sytrd(const SmallWs& w, const double* __restrict__ A, int n, double* __restrict__ Vg, double* s_sc) {
  const int t = threadIdx.x, NT = blockDim.x, lane = t & 63, wave = t >> 6;
  double amax = 0.0;
  for (int e0 = t; e0 < n * n; e0 += NT) { const double a = A[e0]; w.M1[e0] = a; if (w.M2) w.M2[e0] = 0.0; amax = fmax(amax, fabs(a)); }
  for (int j = t; j < 2 * n; j += NT) { w.svp[j] = 0.0; w.swp[j] = 0.0; }
  amax = block_max(amax, w.red);                 // (two barriers inside)
  const double s1_negl = (EPS * amax) * (EPS * amax);   // see k_sytrd: no reflector for a column negligible against |A|
  __syncthreads();
  // wave 0's state across the steps (lane <-> index j = lane + 64 h): v_k, the pending pair (v_(k-1), w_(k-1)), tau_k and the two
  // inner products of the pending pair with v_k
  double vcur[2] = {0.0, 0.0}, vprv[2] = {0.0, 0.0}, wprv[2] = {0.0, 0.0}, tkc = 0.0, wpvc = 0.0, vpvc = 0.0;
  if (wave == 0) {
    // step 0: x = column 0 (= row 0), its Householder vector
    for (int j = lane; j < n; j += 64) w.sx[j] = (j >= 1) ? w.M1[j] : 0.0;
    double s1 = 0.0;
    for (int j = 2 + lane; j < n; j += 64) s1 = fma(w.sx[j], w.sx[j], s1);
    s1 = wsum(s1);
    const double alpha = w.sx[1];
    double beta, tk, sc;
    if (s1 <= s1_negl) { beta = alpha; tk = 0.0; sc = 0.0; }
    else { beta = -copysign(sqrt(fma(alpha, alpha, s1)), alpha); tk = (beta - alpha) / beta; sc = 1.0 / (alpha - beta); }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = lane + 64 * h;
      const double v = (j == 1) ? 1.0 : ((j > 1 && j < n) ? w.sx[j] * sc : 0.0);
      vcur[h] = v;
      if (j >= 1 && j < n) { w.sv[j] = v; Vg[j] = v; }
    }
    if (lane == 0) { w.d[0] = w.M1[0]; w.e[0] = beta; w.tau[0] = tk; s_sc[0] = tk; }
    tkc = tk;
  }
  __syncthreads();
#ifdef EIGF_PROF      // tools/eigf_prof.sh: cycles per step of the phases, threads 0 (wave 0) and 64 (wave 1)
  long long spf[4] = {0, 0, 0, 0}, spt0 = __builtin_amdgcn_s_memtime();
#define SPSTAMP(i) do { const long long t1__ = __builtin_amdgcn_s_memtime(); spf[i] += t1__ - spt0; spt0 = t1__; } while (0)
#else
#define SPSTAMP(i) do { } while (0)
#endif
  for (int k = 0; k + 2 < n; ++k) {
    // ---- pass: y = M1 v_k over the rows and columns > k ----
    {
      const int j = t & 127, rs = t >> 7;
      double acc = 0.0;
      if (j > k && j < n)
        for (int i = k + 1 + rs; i < n; i += 8) acc = fma(w.M1[i * n + j], w.sv[i], acc);
      w.part[rs * 128 + j] = acc;
    }
    SPSTAMP(0);
    __syncthreads();
    SPSTAMP(1);
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);        // the step's critical path; the three waves that share its SIMD only run the update
      const double tk = tkc;
      double arow[2], pj[2], vj[2], pv = 0.0;
#pragma unroll
      for (int h = 0; h < 2; ++h) { const int j = lane + 64 * h; arow[h] = w.M1[(k + 1) * n + ((j < n) ? j : 0)]; }   // (loads first, no branch)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        // (columns outside (k, n) were summed as zeros by the pass: no branch around the eight reads; pairwise sums)
        const double q0 = w.part[j], q1 = w.part[128 + j], q2 = w.part[256 + j], q3 = w.part[384 + j];
        const double q4 = w.part[512 + j], q5 = w.part[640 + j], q6 = w.part[768 + j], q7 = w.part[896 + j];
        const double y = ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7));
        const bool in = j > k && j < n;
        const double p = in ? tk * fma(-wprv[h], vpvc, fma(-vprv[h], wpvc, y)) : 0.0;
        pj[h] = p; vj[h] = in ? vcur[h] : 0.0;
        pv = fma(p, vj[h], pv);
      }
      pv = wsum(pv);
      const double cw = 0.5 * tk * pv;
      double wj[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) { wj[h] = fma(-cw, vj[h], pj[h]); }
      const int hk1 = (k + 1) >> 6, lk1 = (k + 1) & 63;
      const double wk1 = lane_bcast(wj[hk1], lk1);
      const double vpk1 = lane_bcast(vprv[hk1], lk1), wpk1 = lane_bcast(wprv[hk1], lk1);
      // row k+1: the pending update (step k-1) by hand, then this step's; the next column x (rows k+2 ..) stays in registers for
      // its reflector: one reciprocal square root and one reciprocal instead of an IEEE square root and two IEEE divisions
      double akk = 0.0, xj[2];
      const int par = k & 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        const bool in = j > k && j < n;
        const double a = fma(-wpk1, vprv[h], fma(-vpk1, wprv[h], arow[h]));
        akk = (in && j == k + 1) ? a - 2.0 * wk1 : akk;
        xj[h] = (in && j != k + 1) ? a - wj[h] - wk1 * vj[h] : 0.0;
        if (j < n) { w.svp[par * n + j] = vj[h]; w.swp[par * n + j] = wj[h]; }     // this step's pair: the other waves apply it a step later
      }
      akk = lane_bcast(akk, lk1);   // held by the lane that owns j = k+1 ... in half (k+1) >> 6
      // (read here, with every lane active: v_readlane ignores EXEC, and inside the lane-0 region below the compiler is free to
      //  evaluate xj for lane 0 only -- the last off-diagonal entry came out 0)
      const double xlast = lane_bcast(xj[(n - 1) >> 6], (n - 1) & 63);
      if (k + 3 < n) {
        const int k1 = k + 1, hk2 = (k1 + 1) >> 6, lk2 = (k1 + 1) & 63;
        const double alpha = lane_bcast(xj[hk2], lk2);
        double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = lane + 64 * h;
          const double xm = (j > k1 + 1) ? xj[h] : 0.0;             // (xj is zero at and beyond n)
          s1 = fma(xm, xm, s1); s2 = fma(wj[h], xm, s2); s3 = fma(vj[h], xm, s3);
        }
        wsum4(s1, s2, s3, s4);
        double beta, tk1, sc;
        if (s1 <= s1_negl) { beta = alpha; tk1 = 0.0; sc = 0.0; }
        else {
          const double hh = fma(alpha, alpha, s1), rs = fast_rsqrt(hh), nrm = hh * rs;
          beta = -copysign(nrm, alpha);
          tk1 = fma(fabs(alpha), rs, 1.0);                          // (beta - alpha) / beta
          double y = __builtin_amdgcn_rcp(fabs(alpha) + nrm);       // 1 / (alpha - beta) = sign(alpha) / (|alpha| + |x|)
          y = fma(y, fma(-(fabs(alpha) + nrm), y, 1.0), y);
          y = fma(y, fma(-(fabs(alpha) + nrm), y, 1.0), y);
          sc = copysign(y, alpha);
        }
        // w_k'v_(k+1) and v_k'v_(k+1) with v_(k+1) = (1 at k+2, sc x beyond)
        wpvc = fma(sc, s2, lane_bcast(wj[hk2], lk2));
        vpvc = fma(sc, s3, lane_bcast(vj[hk2], lk2));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = lane + 64 * h;
          const double v = (j == k1 + 1) ? 1.0 : ((j > k1 + 1) ? xj[h] * sc : 0.0);
          vcur[h] = v;
          if (j > k1 && j < n) { w.sv[j] = v; Vg[k1 * n + j] = v; }
        }
        if (lane == 0) { w.d[k1] = akk; w.e[k1] = beta; w.tau[k1] = tk1; s_sc[0] = tk1; }
        tkc = tk1;
      } else if (lane == 0) {
        // the reduction ends: the last off-diagonal entry and the last but one diagonal entry (d[n-1]: behind the loop)
        w.d[n - 2] = akk; w.e[n - 2] = xlast;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) { vprv[h] = vj[h]; wprv[h] = wj[h]; }
      __builtin_amdgcn_s_setprio(0);
    } else if (k > 0) {
      // waves 1 .. 15: the rank-2 update of step k-1 on rows / columns >= k+2 (row k+1 is wave 0's, above; nothing below is read again)
      const double* vpb = w.svp + ((k - 1) & 1) * n;
      const double* wpb = w.swp + ((k - 1) & 1) * n;
      double vpj[2], wpj[2];
      int jc[2]; bool cin[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        cin[h] = j >= k + 2 && j < n; jc[h] = cin[h] ? j : k + 2;
        vpj[h] = vpb[jc[h]]; wpj[h] = wpb[jc[h]];
      }
      const int w1 = __builtin_amdgcn_readfirstlane(wave) - 1;
      for (int i0 = k + 2 + w1; i0 < n; i0 += 45) {                 // three rows per trip: i0, i0 + 15, i0 + 30
        double a[3][2], vi[3], wi[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int i = i0 + 15 * r, ic = (i < n) ? i : n - 1;
          vi[r] = vpb[ic]; wi[r] = wpb[ic];
#pragma unroll
          for (int h = 0; h < 2; ++h) a[r][h] = w.M1[ic * n + jc[h]];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int i = i0 + 15 * r;
          if (i < n) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
              if (cin[h]) w.M1[i * n + lane + 64 * h] = fma(-wi[r], vpj[h], fma(-vi[r], wpj[h], a[r][h]));
          }
        }
      }
    }
    SPSTAMP(2);
    __syncthreads();
    SPSTAMP(3);
  }
#ifdef EIGF_PROF
  if (t == 0 || t == 64)
    printf("small_sytrd prof thread %d n %d: pass %lld | barrier %lld | %s %lld | barrier %lld cycles/step\n", t, n, spf[0] / (n - 2), spf[1] / (n - 2),
           t == 0 ? "vector work" : "update", spf[2] / (n - 2), spf[3] / (n - 2));
#endif
  // d[n-1]: the last diagonal entry carries the updates <= n-4 (applied above) and, by hand, the final step's
  if (wave == 0) {
    const int hl = (n - 1) >> 6, ll = (n - 1) & 63;
    const double vl = lane_bcast(vprv[hl], ll), wl = lane_bcast(wprv[hl], ll);
    if (lane == 0) w.d[n - 1] = w.M1[(n - 1) * n + (n - 1)] - 2.0 * vl * wl;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 6. n <= 124, the FAST path (round 3), three launches:
//   k_eigf_reduce (one workgroup)  Householder tridiagonalisation in LDS (small_sytrd); T cleaned of negligible off-diagonals,
//                                  a copy scaled into [-1, 1], its Gershgorin interval -> a small global workspace
//   k_eigf_pairs  (n workgroups)   workgroup k: eigenvalue k by multi-section on Sturm counts (512 points per round, 7-8 rounds
//                                  down to the grid of doubles), then ITS eigenvector of T by the twisted factorisation -- no
//                                  iteration, no orthogonalisation --, and the residual of the pair
//   k_backtransform (checked form) first the orthogonality of the wave's column against all others and the symmetric first-order
//                                  correction Z (I - (Z'Z - I) / 2) (Z staged in LDS), then U = H Z
// Every step past the reduction is parallel over the eigenpairs, the one sequential dimension being the n of a Sturm count or of
// a factorisation (a few thousand cycles), against the Jacobi's 9 x 79 dependent rounds on one CU.  The twisted vectors are as
// good as the eigenvalues are separated: on full-rank kinships (BXD: smallest relative gap 4e-5) they come out orthogonal to
// 4e-14.  Numerically repeated eigenvalues (rank-deficient K, duplicated individuals, identity, Wilkinson pairs) give parallel
// vectors: the checks see it -- stat[ST_EIG_BAD] holds max(|z_a' z_b - delta_ab| / 3e-8, |T z - lambda z| / (2e-14 |T|)) as
// the bit pattern of a double -- and the Jacobi launched behind the three kernels, a no-op otherwise, does the whole job.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double quot_fast(double num, double den) {   // num / den, ~2 ulp; den = +-Inf -> 0
  const double y = __builtin_amdgcn_rcp(den);
  const double q = num * y;
  const double r = fma(-den, q, num);
  const double t = fma(r, y, q);
  return (fabs(den) == INFINITY) ? 0.0 : t;
}
// Number of eigenvalues of the (scaled) tridiagonal matrix below x0 / x1: sign changes of the leading principal minors
//   p_0 = 1, p_1 = d_0 - x, p_(i+1) = (d_i - x) p_i - e_(i-1)^2 p_(i-1)
// (a vanishing p takes the sign opposite to its predecessor's).  Three fp64 operations per step where the quotient form of
// LAPACK's dlaebz needs a division; the pair (p_(i-1), p_i) is rescaled by a power of two every 8 steps, and the matrix is
// scaled to |d_i - x| <= 2 by the caller, so nothing overflows; ds, es2 in LDS (broadcast reads).
__device__ __forceinline__ void sturm2(const double* __restrict__ ds, const double* __restrict__ es2, int n, double x0, double x1,
                                       int& c0, int& c1) {
  // (p, q) = two consecutive minors, the NEWER one alternating between the two registers (no moves); a minor that vanishes
  // exactly needs no care unless the off-diagonal in front of it vanishes too (a decoupled block: the recurrence would stay at
  // zero), which the rescaling block looks at every 8 steps -- counts that are off for a few steps on such matrices only cost
  // the fast path its check (repeated eigenvalues: the Jacobi's job anyway)
  double p0 = 1.0, p1 = 1.0, q0 = ds[0] - x0, q1 = ds[0] - x1;
  unsigned s0 = (unsigned)__double2hiint(q0) >> 31, s1 = (unsigned)__double2hiint(q1) >> 31;
  int i = 1;
  for (; i + 1 < n; i += 2) {
    const double da = ds[i], ea = es2[i - 1], db = ds[i + 1], eb = es2[i];
    p0 = fma(da - x0, q0, -ea * p0); p1 = fma(da - x1, q1, -ea * p1);
    s0 += (unsigned)(__double2hiint(p0) ^ __double2hiint(q0)) >> 31;
    s1 += (unsigned)(__double2hiint(p1) ^ __double2hiint(q1)) >> 31;
    q0 = fma(db - x0, p0, -eb * q0); q1 = fma(db - x1, p1, -eb * q1);
    s0 += (unsigned)(__double2hiint(q0) ^ __double2hiint(p0)) >> 31;
    s1 += (unsigned)(__double2hiint(q1) ^ __double2hiint(p1)) >> 31;
    if ((i & 7) == 7) {
      if (q0 == 0.0) q0 = -1e-100 * p0;
      if (q1 == 0.0) q1 = -1e-100 * p1;
      const int e0 = __builtin_amdgcn_frexp_exp(fmax(fabs(p0), fabs(q0))), e1 = __builtin_amdgcn_frexp_exp(fmax(fabs(p1), fabs(q1)));
      p0 = __builtin_amdgcn_ldexp(p0, -e0); q0 = __builtin_amdgcn_ldexp(q0, -e0);
      p1 = __builtin_amdgcn_ldexp(p1, -e1); q1 = __builtin_amdgcn_ldexp(q1, -e1);
    }
  }
  if (i < n) {
    const double da = ds[i], ea = es2[i - 1];
    p0 = fma(da - x0, q0, -ea * p0); p1 = fma(da - x1, q1, -ea * p1);
    s0 += (unsigned)(__double2hiint(p0) ^ __double2hiint(q0)) >> 31;
    s1 += (unsigned)(__double2hiint(p1) ^ __double2hiint(q1)) >> 31;
  }
  c0 = (int)s0; c1 = (int)s1;
}
// global workspace behind the reflectors: tau | d | e | ds | es2 | {|T|, scale, lo, hi, ...} | Zc (n x n, column-major: eigenvector
// k of T in Zc[k * n ..])
struct EigfWs { double *tau, *d, *e, *ds, *es2, *par, *Zc; };
__host__ __device__ inline EigfWs eigf_ws(double* base, int n) {
  EigfWs q;
  q.tau = base; q.d = base + n; q.e = base + 2 * n; q.ds = base + 3 * n; q.es2 = base + 4 * n; q.par = base + 5 * n; q.Zc = base + 5 * n + 8;
  return q;
}

__global__ void __launch_bounds__(1024) k_eigf_reduce(const double* __restrict__ A, int n, double* __restrict__ Vg, double* __restrict__ wsb,
                                                      int64_t* stat) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  __shared__ double s_sc[4];
  SmallWs w = {};
  {
    double* q = sh;
    w.M1 = q; q += n * n;
    w.sx = q; q += n; w.sv = q; q += n; w.svp = q; q += 2 * n; w.swp = q; q += 2 * n;
    w.part = q; q += 8 * 128;
    w.d = q; q += n; w.e = q; q += n; w.tau = q; q += n; w.red = q; q += 16;
  }
  small_sytrd(w, A, n, Vg, s_sc);
#ifdef SMALL_SYTRD_DBG      // tools/dbg_small_sytrd.sh: the tridiagonal matrix as the reduction left it
  if (t == 0) { for (int j = 0; j < n; ++j) printf("sytrd_dbg n %d j %d d %.17g e %.17g tau %.17g\n", n, j, w.d[j], (j < n - 1) ? w.e[j] : 0.0, (j < n - 2) ? w.tau[j] : 0.0); }
#endif
  const EigfWs g = eigf_ws(wsb, n);
  if (wave == 0) {
    double tn = 0.0;
    for (int j = lane; j < n; j += 64) { tn = fmax(tn, fabs(w.d[j])); if (j < n - 1) tn = fmax(tn, fabs(w.e[j])); }
    tn = wmax(tn);
    for (int j = lane; j < n; j += 64) {
      double ej = (j < n - 1) ? w.e[j] : 0.0;
      if (fabs(ej) <= EPS * tn) ej = 0.0;        // negligible against |T|: the blocks decouple exactly
      w.e[j] = ej;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    double gl = INFINITY, gu = -INFINITY;
    for (int j = lane; j < n; j += 64) {
      const double rad = ((j > 0) ? fabs(w.e[j - 1]) : 0.0) + fabs(w.e[j]);
      gl = fmin(gl, w.d[j] - rad); gu = fmax(gu, w.d[j] + rad);
    }
    gl = -wmax(-gl); gu = wmax(gu);
    // the Sturm counts work on T / bn (entries and arguments within [-1, 1]); bn = 0: T = 0, every eigenvalue 0
    const double bn = fmax(fabs(gl), fabs(gu));
    const double sinv = (bn > 0.0) ? 1.0 / bn : 0.0;
    for (int j = lane; j < n; j += 64) {
      const double es = w.e[j] * sinv;
      g.d[j] = w.d[j]; g.e[j] = w.e[j]; g.ds[j] = w.d[j] * sinv; g.es2[j] = es * es; g.tau[j] = (j < n - 2) ? w.tau[j] : 0.0;
    }
    if (lane == 0) { g.par[0] = tn; g.par[1] = bn; g.par[2] = gl * sinv - 4.0 * EPS * n; g.par[3] = gu * sinv + 4.0 * EPS * n; stat[ST_EIG_FAST] = 1; }
  }
}

template <int NT>
__global__ void __launch_bounds__(NT) k_eigf_pairs(int n, const double* __restrict__ wsb, double* __restrict__ lam_out,
                                                    int64_t* stat, const double* __restrict__ Vg, double* __restrict__ G) {
  __shared__ double sd[128], se[128], sds[128], ses2[128], sDp[128], sDm[128];
  __shared__ double s_red[8];
  __shared__ int s_cnt[2][NT / 64];
  __shared__ int s_r;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, k = blockIdx.x;
  if (k >= n) {                                        // spare workgroups: the back-transformation's group inner products (bt_gram_wave)
    const int grp = (k - n) * (NT / 64) + wave;
    if (grp < bt_groups(n)) bt_gram_wave(Vg, n, grp, G);
    return;
  }
  const EigfWs g = eigf_ws(const_cast<double*>(wsb), n);
  if (t < n) { sd[t] = g.d[t]; se[t] = g.e[t]; sds[t] = g.ds[t]; ses2[t] = g.es2[t]; }
  const double tn = g.par[0], bscale = g.par[1];
  double lo = g.par[2], hi = g.par[3];
#ifdef EIGF_PROF      // tools/eigf_prof.sh: cycles of the phases of workgroups 0, n/2, n-1
  long long pf[4] = {0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
  int prounds = 0;
#define EPSTAMP(i) do { const long long t1__ = __builtin_amdgcn_s_memtime(); pf[i] += t1__ - pt0; pt0 = t1__; } while (0)
#else
#define EPSTAMP(i) do { } while (0)
#endif
  __syncthreads();
  EPSTAMP(0);
  // ---- eigenvalue k: 2 NT interior points per round (two per thread) -----------------------------------------------------
  constexpr int NPT = 2 * NT;
  for (int round = 0; round < 8; ++round) {          // (2 NT + 1)^7 > 2^63 for NT >= 256
    const double wd = hi - lo;
    const double x0 = lo + wd * ((2 * t + 1) * (1.0 / (NPT + 1))), x1 = lo + wd * ((2 * t + 2) * (1.0 / (NPT + 1)));
    int c0, c1;
    sturm2(sds, ses2, n, x0, x1, c0, c1);
    const int mle = (int)wsum((double)((c0 <= k) + (c1 <= k)));       // points with at most k eigenvalues below them
    if (lane == 0) s_cnt[round & 1][wave] = mle;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int q = 0; q < NT / 64; ++q) tot += s_cnt[round & 1][q];
    const double nlo = (tot == 0) ? lo : lo + wd * (tot * (1.0 / (NPT + 1)));
    const double nhi = (tot == NPT) ? hi : lo + wd * ((tot + 1) * (1.0 / (NPT + 1)));
    lo = nlo; hi = nhi;
#ifdef EIGF_PROF
    ++prounds;
#endif
    if (!(hi - lo > 4.0 * EPS * fmax(fabs(lo), fabs(hi)))) break;     // to 2 ulp of the midpoint; workgroup-uniform
  }
  EPSTAMP(1);
  const double l = (0.5 * (lo + hi)) * bscale;
  // ---- its eigenvector of T: twisted factorisation; thread 0 runs the forward factor D+, thread 64 the backward D- ---------
  const double tiny = fmax(EPS * EPS * tn, 1e-300);
  if (t == 0) {
    double dp = sd[0] - l;
    for (int i = 0; i < n - 1; ++i) {
      if (fabs(dp) < tiny) dp = copysign(tiny, dp);
      sDp[i] = dp;
      dp = (sd[i + 1] - l) - quot_fast(se[i] * se[i], dp);
    }
    if (fabs(dp) < tiny) dp = copysign(tiny, dp);
    sDp[n - 1] = dp;
  } else if (t == 64) {
    double dm = sd[n - 1] - l;
    for (int i = n - 1; i >= 1; --i) {
      if (fabs(dm) < tiny) dm = copysign(tiny, dm);
      sDm[i] = dm;
      dm = (sd[i - 1] - l) - quot_fast(se[i - 1] * se[i - 1], dm);
    }
    if (fabs(dm) < tiny) dm = copysign(tiny, dm);
    sDm[0] = dm;
  }
  __syncthreads();
  // twist index r = argmin |gamma_i| (the first one), gamma_i = D+_i + D-_i - (d_i - lambda): waves 0 and 1 hold the n <= 128
  // values one per lane; an arg-min by value, ties to the smaller index
  if (wave < 2) {
    const double gam = (t < n) ? fabs(sDp[t] + sDm[t] - (sd[t] - l)) : INFINITY;
    const double gmin = -wmax(-gam);
    unsigned long long m = __ballot(gam == gmin);
    const int first = m ? (wave * 64 + (int)__builtin_ctzll(m)) : 0x7fffffff;
    if (lane == 0) { s_red[2 + wave] = gmin; s_cnt[0][wave] = first; }
  }
  __syncthreads();
  if (t == 0) s_r = (s_red[2] <= s_red[3]) ? s_cnt[0][0] : s_cnt[0][1];
  __syncthreads();
  const int r = s_r;
  // z_r = 1, z_i = -(e_i / D+_i) z_(i+1) upwards, z_(i+1) = -(e_i / D-_(i+1)) z_i downwards.  The quotients do not depend on z:
  // every lane takes its own, and the two recurrences are a suffix / a prefix PRODUCT around r -- seven doubling steps over LDS
  // instead of two chains of up to n dependent quotient-and-multiply steps on two threads (13 k of this kernel's 112 k cycles).
  double fct = 1.0;
  if (t < n) { if (t < r) fct = -quot_fast(se[t], sDp[t]); else if (t > r) fct = -quot_fast(se[t - 1], sDm[t]); }
  __syncthreads();                                     // (every quotient is taken before the factors are overwritten)
  double* za = sDp; double* zb = sDm;
  if (t < 128) za[t] = fct;
  __syncthreads();
  for (int s_ = 1; s_ < n; s_ <<= 1) {
    if (t < n) {
      double v = za[t];
      if (t > r && t - s_ > r) v *= za[t - s_]; else if (t < r && t + s_ < r) v *= za[t + s_];
      zb[t] = v;
    }
    __syncthreads();
    double* tmp = za; za = zb; zb = tmp;
  }
  if (wave < 2) {
    const double zt = (t < n) ? za[t] : 0.0;
    const double nr = wsum(zt * zt);
    if (lane == 0) s_red[wave] = nr;
  }
  __syncthreads();
  EPSTAMP(2);
  const double inv = fast_rsqrt(s_red[0] + s_red[1]);
  double res = 0.0;
  if (t < n) {
    auto zat = [&](int i) { return za[i] * inv; };
    const double zi = zat(t);
    g.Zc[(size_t)k * n + t] = zi;
    double rr = (sd[t] - l) * zi;
    if (t > 0) rr = fma(se[t - 1], zat(t - 1), rr);
    if (t < n - 1) rr = fma(se[t], zat(t + 1), rr);
    res = fabs(rr);
    if (!(rr == rr)) res = INFINITY;
  }
  if (t == 0) lam_out[k] = l;
  if (wave < 2) {
    res = wmax(res);
    if (lane == 0) eigf_note(stat, res, 2e-14 * tn);
  }
  EPSTAMP(3);
#ifdef EIGF_PROF
  if (t == 0 && (k == 0 || k == n / 2 || k == n - 1))
    printf("eigf_pairs prof k %d: load %lld | %d rounds %lld | twisted %lld | finish %lld cycles\n", k, pf[0], prounds, pf[1], pf[2], pf[3]);
#endif
}

}  // namespace

// Largest n the solver takes: the back-transformation holds a column in 32 registers per lane of a wave (n <= 2048), and the
// seven vectors of k_sytrd fit the LDS up to there; its rows move to global memory where the LDS budget ends (~1450).
int eig_dc_max_n(const blmm_ctx*) { return 2048; }

// The fast path for 3 <= n <= 124 (k_eigf_reduce, k_eigf_pairs, k_backtransform with the orthogonality check).  The device decides
// whether the result stands (stat[ST_EIG_FAST] = 1: it ran; stat[ST_EIG_BAD]: largest check / bound as the bits of a double, accepted
// up to 1.0); the caller launches the Jacobi behind it, which returns at once when it does.
int eig_fast_max_n() { return 124; }
int launch_eig_fast(blmm_ctx* ctx, const double* A, int n, double* lraw, double* evec, int64_t* stat) {
  if (n < 3 || n > eig_fast_max_n()) return BLMM_ERR_UNSUPPORTED;
  int rc;
  if ((rc = ensure(ctx, ctx->eigW, sizeof(double) * ((size_t)2 * n * n + (size_t)5 * n + 8) + 256))) return rc;
  const size_t lds = sizeof(double) * ((size_t)n * n + (size_t)9 * n + 8 * 128 + 16) + 64;
  if (lds > 158 * 1024) return BLMM_ERR_UNSUPPORTED;
  double* Vg = ptr<double>(ctx->eigW);
  double* wsb = Vg + (size_t)n * n;
  const EigfWs g = eigf_ws(wsb, n);
  BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eigf_reduce), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_eigf_reduce, dim3(1), dim3(1024), lds, ctx->stream, A, n, Vg, wsb, stat);
  KCHECK();
  // (256 threads: 512 / 1024 per workgroup -- more points per round, fewer rounds -- measured 0.265 / 0.286 ms of eigen phase against
  // 0.256: the waves sharing a SIMD slow each other's dependent chains)
  if ((rc = ensure(ctx, ctx->btG, sizeof(double) * 8 * (size_t)bt_groups(n)))) return rc;
  double* btG = ptr<double>(ctx->btG);
  hipLaunchKernelGGL(k_eigf_pairs<256>, dim3(n + (bt_groups(n) + 3) / 4), dim3(256), 0, ctx->stream, n, (const double*)wsb, lraw, stat, (const double*)Vg, btG);
  KCHECK();
  static const bool bt_stage_off = dev_env("BLMM_BT_STAGE") && dev_env("BLMM_BT_STAGE")[0] == '0';
  const int stage_v = (bt_lds_staged(2, n) <= 158 * 1024 && !bt_stage_off) ? 1 : 0;
  const size_t lds_bt = stage_v ? bt_lds_staged(2, n) : bt_lds_chk(2, n);
  BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_backtransform<2, 1, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bt));
  hipLaunchKernelGGL((k_backtransform<2, 1, 256>), dim3((n + 3) / 4), dim3(256), lds_bt, ctx->stream, Vg, g.tau, (const double*)btG, n, evec, (const double*)g.Zc, stat, stage_v);
  KCHECK();
  ctx->eig_plan_n = -1;    // the workspace was reused: a cached merge tree of the multi-workgroup solver is gone
  return BLMM_OK;
}

// Eigen-decomposition of the symmetric n x n matrix A (device, not modified): lraw ascending, evec[i*n + r].
int launch_eig_dc(blmm_ctx* ctx, const double* A, int n, double* lraw, double* evec, int64_t* stat) {
  if (n < 3) return fail(ctx, BLMM_ERR_UNSUPPORTED, "eig_dc: n < 3");
  if (n > eig_dc_max_n(ctx)) return BLMM_ERR_UNSUPPORTED;
  const size_t nn = (size_t)n * n;
  // ---- plan: leaves and the merge tree (host) ----
  // leaf blocks: implicit QL on one wave is a scalar recurrence (334 us for leaves of 31 rows, 102 us at 16, whatever n), while
  // a merge level of tiny nodes costs ~50 us since the secular solver stops on LAPACK's test.  The number of leaves is a power
  // of two; tools/sweep_leaf.sh over n = 130 .. 1400 with bounds 6 .. 32: leaves of 6-12 rows are best or tied everywhere
  // (n = 500: eigen 2.83 ms at <= 12, 2.86 at <= 24, 3.02 at <= 32; n = 300: 1.74 / 1.78 / 1.77) (BLMM_EIG_LEAF: A/B testing)
  static const int leaf_env = dev_env("BLMM_EIG_LEAF") ? atoi(dev_env("BLMM_EIG_LEAF")) : 0;
  const int leaf = (leaf_env >= 2 && leaf_env <= LEAF) ? leaf_env : 12;
  int nl = 1;
  while ((n + nl - 1) / nl > leaf) nl *= 2;
  std::vector<int> bounds(nl + 1);
  for (int i = 0; i <= nl; ++i) bounds[i] = (int)std::llround((double)i * n / nl);
  std::vector<std::vector<int>> levels;   // per level: flat [lo, mid, hi] triples
  {
    std::vector<int> cur(bounds);           // nl is a power of two: every level pairs its segments up exactly
    while (cur.size() > 2) {
      std::vector<int> nodes, nxt;
      for (size_t i = 0; i + 2 < cur.size(); i += 2) {
        nodes.push_back(cur[i]); nodes.push_back(cur[i + 1]); nodes.push_back(cur[i + 2]);
      }
      for (size_t i = 0; i < cur.size(); i += 2) nxt.push_back(cur[i]);
      levels.push_back(nodes);
      cur = nxt;
    }
  }
  size_t nnodes_total = 0;
  for (auto& l : levels) nnodes_total += l.size() / 3;
  // ---- workspace ----
  int rc;
  const size_t ints = (size_t)7 * n + 320 + 2 * ABSMAX_PARTS + 2 + 4 * nnodes_total + 3 * nnodes_total + (nl + 1) + 64;
  const size_t dbls = 5 * nn + (size_t)14 * n + nnodes_total + 64;
  if ((rc = ensure(ctx, ctx->eigW, sizeof(double) * dbls + sizeof(int) * ints + 256))) return rc;
  double* base = ptr<double>(ctx->eigW);
  double* V = base; double* Qa = V + nn; double* Qb = Qa + nn; double* Dm = Qb + nn; double* Wt = Dm + nn;
  double* vec = Wt + nn;
  double* d = vec; double* e = d + n; double* tau = e + n; double* lamA = tau + n; double* lamB = lamA + n;
  double* dl = lamB + n; double* zl = dl + n; double* zh = zl + n; double* defld = zh + n; double* lamnew = defld + n;
  double* rotc = lamnew + n; double* rots = rotc + n; double* pbuf = rots + n;   // pbuf: 2n, then rowbuf below
  double* rho = pbuf + 2 * n;   // nnodes_total (+ pad)
  // rowbuf shares Dm (unused until the first merge)
  double* rowbuf = Dm;
  int* ib = reinterpret_cast<int*>(rho + nnodes_total + 8);
  int* colidx = ib; int* deflcol = colidx + n; int* rota = deflcol + n; int* rotb = rota + n; int* posn = rotb + n;
  int* posd = posn + n; int* sync = posd + n;          // sync: n + 320 + 2 ABSMAX_PARTS ints reserved: [0] abort, [8 .. 8+G) flags, [n + 320 ..) partial maxima of |A|
  int* info = sync + n + 320 + 2 * ABSMAX_PARTS + 2; int* nodes_dev = info + 4 * nnodes_total; int* bounds_dev = nodes_dev + 3 * nnodes_total;
  // plan -> device (tiny; cached per n in the context: the copy is skipped when n repeats)
  if (ctx->eig_plan_n != n) {
    std::vector<int> flat;
    for (auto& l : levels) flat.insert(flat.end(), l.begin(), l.end());
    BLMM_HIP(hipMemcpyAsync(nodes_dev, flat.data(), sizeof(int) * flat.size(), hipMemcpyHostToDevice, ctx->stream));
    BLMM_HIP(hipMemcpyAsync(bounds_dev, bounds.data(), sizeof(int) * bounds.size(), hipMemcpyHostToDevice, ctx->stream));
    BLMM_HIP(hipStreamSynchronize(ctx->stream));   // the host vectors die at return
    ctx->eig_plan_n = n;
  }
  // ---- 1. tridiagonalisation ----
  {
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    // rows per workgroup: as many as the LDS takes beside the seven vectors (fewer workgroups = a cheaper barrier)
    if (sizeof(double) * ((size_t)7 * n + 128) + 64 > 156 * 1024) return BLMM_ERR_UNSUPPORTED;
    const size_t budget = 156 * 1024 - sizeof(double) * ((size_t)7 * n + 128) - 64;
    int nloc = (int)(budget / (sizeof(double) * (size_t)n));
    int G = nloc >= 1 ? (n + nloc - 1) / nloc : cus + 1;
    // the rows of a workgroup in global memory when the LDS cannot hold them with one workgroup per CU (BLMM_SYTRD_GLB=1: always)
    const bool glb = G > cus || (dev_env("BLMM_SYTRD_GLB") && dev_env("BLMM_SYTRD_GLB")[0] == '1');
    if (glb) G = 1;
    // ~10 rows per workgroup measured best (n = 500: 3.6 ms at G = 48 against 4.0 at the LDS minimum of 16; tools/sweep_sytrd.sh)
    const int Gmin = G;
    if ((n + 9) / 10 > G) G = std::min(cus, (n + 9) / 10);
    if (const char* ge = dev_env("BLMM_SYTRD_G")) { const int gv = atoi(ge); if (gv >= Gmin && gv <= cus) G = gv; }
    nloc = (n + G - 1) / G;
    const size_t lds = sizeof(double) * ((size_t)7 * n + 128 + (glb ? 0 : (size_t)nloc * n));
    double* Aw = Qa;                          // G * nloc * n <= n^2 + G n doubles: the two Q buffers are unused until the leaves
    SytrdEx ex; ex.gr = reinterpret_cast<unsigned long long*>(rowbuf); ex.abort = sync;      // 8 n granules in Dm (unused until the merges)
    unsigned long long* parts = reinterpret_cast<unsigned long long*>(sync + ((n + 320 + 1) & ~1));   // 8-byte aligned (sync is)
    ex.anorm = parts;
    hipLaunchKernelGGL(k_sytrd_prep, dim3(ABSMAX_PARTS), dim3(256), 0, ctx->stream, A, (int64_t)n * n, parts, sync,
                       reinterpret_cast<unsigned long long*>(rowbuf), (int64_t)8 * n);           // tags 0: nothing published
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sytrd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sytrd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nthr = 512;                                           // barriers at 1024 threads cost almost twice as much
    if (const char* te = dev_env("BLMM_SYTRD_NT")) { const int tv = atoi(te); if (tv == 256 || tv == 512) nthr = tv; }
    auto launch = [&]() {
      if (glb) hipLaunchKernelGGL(k_sytrd<true>, dim3(G), dim3(nthr), lds, ctx->stream, A, n, nloc, d, e, tau, V, ex, stat, Aw);
      else hipLaunchKernelGGL(k_sytrd<false>, dim3(G), dim3(nthr), lds, ctx->stream, A, n, nloc, d, e, tau, V, ex, stat, Aw);
    };
    if (G > 1) {
      GridKernelGuard gk(ctx);
      if (gk.rc) return gk.rc;
      launch();
      KCHECK();
      if ((rc = gk.record())) return rc;
    } else {
      launch();
      KCHECK();
    }
  }
  // ---- 2. leaves ----
  // a node reads the full square [lo, hi)^2 of its input Q: the blocks off the solved halves' diagonal must read as zero
  // (block-diagonal eigenvector matrix); both ping-pong buffers, because every level leaves such blocks unwritten
  BLMM_HIP(hipMemsetAsync(Qa, 0, sizeof(double) * 2 * nn, ctx->stream));
  if ((rc = ensure(ctx, ctx->btG, sizeof(double) * 8 * (size_t)bt_groups(n)))) return rc;
  double* btG = ptr<double>(ctx->btG);
  hipLaunchKernelGGL(k_tql_leaves, dim3(nl + bt_groups(n)), dim3(64), 0, ctx->stream, d, e, n, bounds_dev, lamA, Qa, stat, rho + nnodes_total,   // |T| -> the spare doubles behind rho
                     nl, (const double*)V, btG);
  KCHECK();
  // ---- 3. merges ----
  DcWs w;
  w.tnorm = rho + nnodes_total;
  { const char* de = dev_env("BLMM_DC_DEFLATE"); w.serial_deflate = (de && de[0] == 's') ? 1 : 0; }   // read per call: a test flips it
  w.n = n; w.e = e; w.Dm = Dm; w.Wt = Wt; w.dl = dl; w.zl = zl; w.zh = zh; w.defld = defld; w.lamnew = lamnew;
  w.rotc = rotc; w.rots = rots; w.colidx = colidx; w.deflcol = deflcol; w.rota = rota; w.rotb = rotb; w.posn = posn; w.posd = posd;
  double* lamIn = lamA; double* lamOut = lamB; double* Qin = Qa; double* Qout = Qb;
  size_t node_off = 0;
  for (auto& lvl : levels) {
    const int nnode = (int)(lvl.size() / 3);
    int Nmax = 0;
    for (int q = 0; q < nnode; ++q) Nmax = std::max(Nmax, lvl[3 * q + 2] - lvl[3 * q]);
    w.lamIn = lamIn; w.lamOut = lamOut; w.Qin = Qin; w.Qout = Qout;
    w.info = info + 4 * node_off; w.rho = rho + node_off; w.nodes = nodes_dev + 3 * node_off;
    const size_t lds_defl = sizeof(double) * ((size_t)4 * Nmax + 16 + (Nmax + 1) / 2 + 2);
    if (lds_defl > 48 * 1024) BLMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dc_deflate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_defl));
    hipLaunchKernelGGL(k_dc_deflate, dim3(nnode), dim3(1024), lds_defl, ctx->stream, w);
    const int nq4 = (Nmax + 3) / 4, nb256 = (Nmax + 255) / 256;
    const size_t lds_sec = sizeof(double) * (size_t)2 * Nmax;
    hipLaunchKernelGGL(k_dc_secular_rot, dim3(nnode, nq4 + nb256), dim3(256), lds_sec, ctx->stream, w, nq4);
    hipLaunchKernelGGL(k_dc_zhat_fin, dim3(nnode, nq4 + nb256), dim3(256), sizeof(double) * (size_t)Nmax, ctx->stream, w, nq4);
    hipLaunchKernelGGL(k_dc_wt_copy, dim3(nnode, nq4 + nb256 * DC_COPY_Z), dim3(256), 0, ctx->stream, w, nq4);
    {
      // 64 x 64 tiles per workgroup where they fill the chip, 32 x 32 where they would not (n = 500, top level: 64 workgroups
      // against 256; BLMM_DC_TILE=1 / 2 forces one: A/B testing)
      const int t2 = (Nmax + 63) / 64, t1 = (Nmax + 31) / 32;
      static const int tile_env = dev_env("BLMM_DC_TILE") ? atoi(dev_env("BLMM_DC_TILE")) : 0;
      const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
      const bool small = tile_env == 1 || (tile_env != 2 && (long)nnode * t2 * t2 < cus);
      if (small) hipLaunchKernelGGL((k_dc_gemm<1, 1>), dim3(nnode, t1 * t1), dim3(256), 0, ctx->stream, w, t1);
      else hipLaunchKernelGGL((k_dc_gemm<2, 2>), dim3(nnode, t2 * t2), dim3(256), 0, ctx->stream, w, t2);
    }
    KCHECK();
#ifdef SEC_DIAG
    {
      (void)hipStreamSynchronize(ctx->stream);
      unsigned int h[64];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sec_hist), sizeof(h));
      fprintf(stderr, "secular diag: level of %d nodes (N <= %d): iterations per root:", nnode, Nmax);
      for (int q = 0; q < 64; ++q) if (h[q]) fprintf(stderr, " %d:%u", q, h[q]);
      fprintf(stderr, "\n");
      unsigned int z[64] = {0};
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sec_hist), z, sizeof(z));
    }
#endif
    std::swap(lamIn, lamOut); std::swap(Qin, Qout);
    node_off += nnode;
  }
  // ---- 4. back-transformation (in place on the final Q), results out ----
  {
    const int nr = (n + 63) / 64;
    static const bool bt_wide = dev_env("BLMM_BT_WIDE") && dev_env("BLMM_BT_WIDE")[0] == '1';
    static const bool bt_cpw2 = dev_env("BLMM_BT_CPW") && dev_env("BLMM_BT_CPW")[0] == '2';
#define BT(NR)                                                                                                             \
  do {                                                                                                                     \
    /* one column per wave, four waves per workgroup: the kernel is VALU-issue bound per SIMD (~110 instructions per        \
       reflector and column), so the columns are spread over as many CUs as there are (n = 500: 125 workgroups instead of   \
       32 of eight two-column waves: 302 -> 228 us; n = 200: 95 -> 59; n = 1000: 721 -> 739; BLMM_BT_WIDE=1: the old shape) */  \
    constexpr int CPWW = (NR <= 8) ? 2 : 1;                                                                                \
    const size_t lds = bt_lds_plain(NR);                                                                                   \
    if (bt_wide && NR <= 16) {             /* (beyond: four reflectors x NR registers do not fit two waves per SIMD) */        \
      constexpr int NRW = (NR <= 16) ? NR : 16;                                                                            \
      const int cols_per_wg = 8 * CPWW;                                                                                    \
      if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_backtransform<NRW, CPWW, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((k_backtransform<NRW, CPWW, 512>), dim3((n + cols_per_wg - 1) / cols_per_wg), dim3(512), lds, ctx->stream, V, tau, (const double*)btG, n, Qin, (const double*)nullptr, (int64_t*)nullptr, 0); \
    } else if (bt_cpw2 && NR <= 16) {      /* two columns per wave, 8 per workgroup: half the LDS reads and L2 fetches per column */ \
      constexpr int NRW = (NR <= 16) ? NR : 16;                                                                            \
      if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_backtransform<NRW, 2, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((k_backtransform<NRW, 2, 256>), dim3((n + 7) / 8), dim3(256), lds, ctx->stream, V, tau, (const double*)btG, n, Qin, (const double*)nullptr, (int64_t*)nullptr, 0);   \
    } else {                                                                                                               \
      if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_backtransform<NR, 1, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((k_backtransform<NR, 1, 256>), dim3((n + 3) / 4), dim3(256), lds, ctx->stream, V, tau, (const double*)btG, n, Qin, (const double*)nullptr, (int64_t*)nullptr, 0);   \
    }                                                                                                                      \
  } while (0)
    if (nr <= 2) BT(2); else if (nr <= 4) BT(4); else if (nr <= 8) BT(8); else if (nr <= 16) BT(16); else if (nr <= 24) BT(24); else BT(32);
#undef BT
    KCHECK();
  }
  BLMM_HIP(hipMemcpyAsync(lraw, lamIn, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
  BLMM_HIP(hipMemcpyAsync(evec, Qin, sizeof(double) * nn, hipMemcpyDeviceToDevice, ctx->stream));
  return BLMM_OK;
}

}  // namespace blmm
