// kernels_scan.hip -- the LOD kernels: every (trait, marker) test of the bulkscan hot path.
//
//   L[i, j] = -(n/2) * log10(1 - r_ij^2)                                   (r2lod, src/bulkscan_helpers.jl:22-24)
//   r_ij    = <P x~_i, P y~_j> / (|P x~_i| |P y~_j|)                         (computeR_LMM, src/bulkscan_helpers.jl:47-64)
//
// rewritten as a markers x traits contraction over the n individuals on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64), with the projection / normalisation / r -> LOD map fused into the epilogue:
//
//   exact (per-trait weights; univar_liteqtl, src/bulkscan_helpers.jl:127-150):
//       num = x_i' a0_j,  Sxx = (x_i.^2)' a1_j,  u_q = x_i' a(2+q)_j,   r^2 = num^2 / (Sxx - sum_q u_q^2)
//   table (shared weights; weighted_liteqtl :175-201, scan_perms_lite src/scan.jl:542):
//       r = (x_i' a0_j) * isx[bin_j][i]
//   alt   (bulkscan_alt_grid, src/bulkscan.jl:495-522 with tmax!, src/bulkscan_helpers.jl:330-350):
//       running max over the h2 grid of  ln10 * LOD_g + Ell[g, j]
//
// Operand layout (both row-major "k-major", produced by k_rotate / k_panels):
//   Xt[k][i]  markers,  ld = ldx (padded to the tile);     P[q][k][j]  A-side panels, ld = ldp.
// MFMA roles: A (16 rows) = traits, B (16 cols) = markers, so that D's 16 lanes of a register hold 16
// marker slots of ONE trait column of L.  Marker <-> (block nb, col c) is permuted to i0 + NB*c + nb (mslot below) and
// trait <-> (block mb, row r) to tbase + MB*r + mb, so every lane owns NB consecutive markers (vector
// loads of Xt, 32-byte stores of L; 16 lanes = 512 contiguous bytes of one L column) and MB consecutive
// traits (vector loads of the panels).
// Workgroup = 4 waves (2 x 2) = (32*MB) traits x (32*NB) markers; no LDS: fragments come straight from
// L2/L1 (a 64-cycle f64 MFMA leaves the operand traffic at a few bytes/clk/CU).
#include "blmm_internal.h"
#include "fastmath.h"
#include <type_traits>
#include <cmath>
#include <cstdlib>

namespace blmm {

#define KCHECK()                                                                                      \
  do {                                                                                                \
    hipError_t e__ = hipGetLastError();                                                               \
    if (e__ != hipSuccess) return fail(ctx, BLMM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e__)); \
  } while (0)

template <int N> struct VecD;
template <> struct VecD<1> { typedef double type; };
template <> struct VecD<2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecD<4> { typedef double type __attribute__((ext_vector_type(4))); };

template <int N>
__device__ __forceinline__ void loadv(double (&dst)[N], const double* __restrict__ p) {
  if constexpr (N == 1) {
    dst[0] = p[0];
  } else if constexpr (N == 2) {
    const d2 v = *reinterpret_cast<const d2*>(p);
    dst[0] = v[0]; dst[1] = v[1];
  } else {
    const d2 v0 = *reinterpret_cast<const d2*>(p);
    const d2 v1 = *reinterpret_cast<const d2*>(p + 2);
    dst[0] = v0[0]; dst[1] = v0[1]; dst[2] = v1[0]; dst[3] = v1[1];
  }
}

// Fragment loads through buffer descriptors: the 128-bit SRD is built from wave-uniform scalars only (kernel
// arguments, blockIdx, the loop counter), every per-lane part sits in ONE 32-bit voffset, so the K loop carries two
// address VGPRs instead of a 64-bit pointer per panel (cdna_hip_programming.md T8/T20).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const double* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), /*stride*/ 0, /*bytes*/ 0xffffffffu, 0x00020000);
}

template <int N>
__device__ __forceinline__ void bufload(double (&dst)[N], __amdgpu_buffer_rsrc_t srd, uint32_t voff) {
  if constexpr (N == 1) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(srd, voff, 0, 0);
    dst[0] = __builtin_bit_cast(double, v);
  } else if constexpr (N == 2) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, 0, 0);
    const d2 w = __builtin_bit_cast(d2, v);
    dst[0] = w[0]; dst[1] = w[1];
  } else {
    const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, 0, 0);
    const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(srd, voff + 16, 0, 0);
    const d2 w0 = __builtin_bit_cast(d2, v0), w1 = __builtin_bit_cast(d2, v1);
    dst[0] = w0[0]; dst[1] = w0[1]; dst[2] = w1[0]; dst[3] = w1[1];
  }
}

// 8-byte-aligned 16-byte vector (columns of L start at arbitrary multiples of 8 bytes: ld = p is odd for BXD)
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

// ---- marker slots of a lane ----------------------------------------------------------------------------------------------
// A wave's 16 NB markers; MFMA block nb, column c (= lane & 15): column c owns the NB CONSECUTIVE markers i0 + NB c + nb (32
// contiguous bytes of Xt and of an L column for NB = 4, in two 16-byte instructions).  Measured and not adopted (round 3): the
// split map i0 + 32 (nb >> 1) + 2 c + (nb & 1), with which one 16-byte access per lane covers 256 contiguous bytes per 16 lanes
// (every 128-byte line written whole by ONE store instruction): null-exact scan 1.28-1.29 ms against 1.22-1.26, null-grid 0.93
// against 0.91-0.92 -- a lane's two accesses to the same 64-byte sector are worth more than dense quarter-waves.
template <int NB>
__device__ __forceinline__ int mslot(int c, int nb) { return NB * c + nb; }
template <int NB>
__device__ __forceinline__ uint32_t mvoff(int c) { return (uint32_t)(NB * c * 8); }   // byte offset of the lane's first slot
template <int NB>
__device__ __forceinline__ void bufload_m(double (&dst)[NB], __amdgpu_buffer_rsrc_t srd, uint32_t voff) { bufload<NB>(dst, srd, voff); }
// per-marker values (marker norms): p points at the wave's first marker
template <int NB>
__device__ __forceinline__ void loadv_m(double (&dst)[NB], const double* __restrict__ p, int c) { loadv<NB>(dst, p + NB * c); }
// a lane's NB results of one trait column: col points at L[i0, trait], i0 the wave's first marker, `valid` = p - i0 markers exist
template <int NB>
__device__ __forceinline__ void store_m(double* __restrict__ col, int c, const double (&out)[NB], int64_t valid) {
  double* dst = col + NB * c;
  if (NB * c + NB <= valid) {
    if constexpr (NB == 4) {
      __builtin_nontemporal_store((d2u){out[0], out[1]}, reinterpret_cast<d2u*>(dst));
      __builtin_nontemporal_store((d2u){out[2], out[3]}, reinterpret_cast<d2u*>(dst + 2));
    } else if constexpr (NB == 2) {
      __builtin_nontemporal_store((d2u){out[0], out[1]}, reinterpret_cast<d2u*>(dst));
    } else {
      __builtin_nontemporal_store(out[0], dst);
    }
  } else {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      if (NB * c + nb < valid) dst[nb] = out[nb];
  }
}

// ---- reduce-in-epilogue (RedArgs, blmm_internal.h): what replaces store_m when no L is written --------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_movi(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }
// the better of two (value, marker) candidates by k_colmax's rule: strictly larger value, or the same value at the lower marker;
// a NaN never wins (every comparison with it is false), -1 marks "no candidate yet"
__device__ __forceinline__ void red_comb(double& best, int& bi, double ob, int oi) {
  if (ob > best || (ob == best && oi >= 0 && (bi < 0 || oi < bi))) { best = ob; bi = oi; }
}
// a lane's NB LODs of one trait: lane c (= lane & 15) of the 16-lane DPP row holds the markers i0 + NB c + nb; vl = markers that
// exist from i0 on (at most 16 NB); all 16 lanes of a row are in the same trait, so the row is active or skipped as a whole
template <int NB>
__device__ __forceinline__ void red_row(const RedArgs& R, int64_t trait, int64_t i0, int c, int lane, const double (&out)[NB], int vl) {
  double best = -INFINITY; int bi = -1;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int idx = NB * c + nb;
    if (idx < vl && out[nb] > best) { best = out[nb]; bi = idx; }
  }
  { const double ob = blmm_dpp_mov<0xB1>(best); const int oi = dpp_movi<0xB1>(bi); red_comb(best, bi, ob, oi); }     // lane ^ 1
  { const double ob = blmm_dpp_mov<0x4E>(best); const int oi = dpp_movi<0x4E>(bi); red_comb(best, bi, ob, oi); }     // lane ^ 2
  { const double ob = blmm_dpp_mov<0x141>(best); const int oi = dpp_movi<0x141>(bi); red_comb(best, bi, ob, oi); }   // row_half_mirror
  { const double ob = blmm_dpp_mov<0x140>(best); const int oi = dpp_movi<0x140>(bi); red_comb(best, bi, ob, oi); }   // row_mirror
  if (c == 0) {
    const int64_t at = (i0 >> 6) * R.ldm + trait;
    R.pmax[at] = best;
    R.parg[at] = bi < 0 ? -1 : (int)i0 + bi;
  }
  if (R.want_trip) {                                           // kernel argument: a scalar branch
    bool h[NB]; int my = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { h[nb] = (NB * c + nb < vl) && out[nb] > R.thr; my += h[nb] ? 1 : 0; }
    const unsigned long long any = __ballot(my != 0);
    if (any != 0ull) {                                         // rare: one wave-aggregated atomic reserves the slots (as k_threshold)
      const unsigned long long b0 = __ballot((my & 1) != 0), b1 = __ballot((my & 2) != 0), b2 = __ballot((my & 4) != 0);
      const unsigned long long lt = (1ull << lane) - 1ull;
      const int pre = __builtin_popcountll(b0 & lt) + 2 * __builtin_popcountll(b1 & lt) + 4 * __builtin_popcountll(b2 & lt);
      const int tot = __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1) + 4 * __builtin_popcountll(b2);
      const int leader = (int)__builtin_ctzll(any);
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(R.cnt, (unsigned long long)tot);
      const unsigned int blo = __builtin_amdgcn_readlane((unsigned int)base, leader);
      const unsigned int bhi = __builtin_amdgcn_readlane((unsigned int)(base >> 32), leader);
      unsigned long long slot = (((unsigned long long)bhi << 32) | blo) + (unsigned long long)pre;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (h[nb]) {
          if ((int64_t)slot < R.cap) { R.ti[slot] = (int32_t)(i0 + NB * c + nb); R.tj[slot] = (int32_t)trait; R.tl[slot] = out[nb]; }
          ++slot;
        }
    }
  }
}
__device__ __forceinline__ int red_valid(int64_t p, int64_t i0, int span) { const int64_t v = p - i0; return v > span ? span : (int)v; }

__device__ __forceinline__ int64_t xcd_swizzle(int64_t bid, int64_t nwg) {
  // Workgroups are dealt round-robin over the 8 XCDs (bid % 8).  Give each XCD a contiguous range of tiles
  // so the A-side panels of a trait tile stay in ONE XCD's L2 (speed only; any placement is correct).
  const int64_t q = nwg >> 3;
  return (bid < (q << 3)) ? (bid & 7) * q + (bid >> 3) : bid;
}

// Tile walk inside an XCD's range: groups of GT trait tiles; inside a group the marker tile is the slow index and
// the trait tile the fast one.  The GT trait tiles' A-side panels (GT x ~120 KB) stay in the 4 MB L2 while each
// 80 KB marker tile of Xt is fetched once per group instead of once per trait tile (HBM/MALL fetch / ~GT).
constexpr int GT = 16;

__device__ __forceinline__ void tile_of(int64_t id, int64_t ntile_t, int ntile_i, int64_t& tile_t, int& tile_i) {
  const int64_t per_group = (int64_t)GT * ntile_i;
  const int64_t g = id / per_group, rem = id - g * per_group;
  const int64_t t_first = g * GT;
  const int gt = (int)((ntile_t - t_first < GT) ? (ntile_t - t_first) : GT);  // last group may be short
  tile_i = (int)(rem / gt);
  tile_t = t_first + (rem - (int64_t)tile_i * gt);
}

// the same walk in 32-bit arithmetic (launchers bound the workgroup count by 2^31): a 64-bit division is ~100 scalar
// instructions, and the walk has two of them at the head of every workgroup
template <int GTV = GT>
__device__ __forceinline__ void tile_of32(uint32_t id, uint32_t ntile_t, uint32_t ntile_i, uint32_t& tile_t, uint32_t& tile_i) {
  const uint32_t per_group = (uint32_t)GTV * ntile_i;
  const uint32_t g = id / per_group, rem = id - g * per_group;
  const uint32_t t_first = g * GTV;
  const uint32_t gt = (ntile_t - t_first < (uint32_t)GTV) ? (ntile_t - t_first) : (uint32_t)GTV;  // last group may be short
  tile_i = rem / gt;
  tile_t = t_first + (rem - tile_i * gt);
}

// PERM (table mode only): the trait tiles are the first ceil(*a.count / tile) tiles of the panel region starting at column
// a.col0, a panel column's trait is a.perm[column] (-1: padding) -- the shared-weights class of the low-rank form, which is
// exactly a one-bin table scan (102 VGPRs, 4 waves per SIMD, where k_scan_lr holds 2).  Every XCD takes an eighth of the items.
// MORE (exact mode, c > CFAST): the covariate panels beyond the first CFAST are contracted CFAST at a time ahead of the main
// loop and folded into the Sxx accumulator as -u_q^2 (an MFMA accumulates on top of whatever its accumulator holds), so the
// register budget does not grow with c; the marker tile is re-read from L2 once per chunk.
template <int NX, int MB, int NB, bool TABLE, int W2, bool PERM = false, bool MORE = false, bool PV = false, bool RED = false>
__global__ void __launch_bounds__(64 * W2 * W2, 2) k_scan(ScanArgs a, int ntile_i, int64_t nwg) {
  static_assert(!(RED && PV), "reduce-in-epilogue: no p-value output");
  constexpr int NP = 1 + NX;  // A-side panels consumed
  constexpr int NT = 64 * W2 * W2;
  static_assert(!PERM || (TABLE && NX == 0), "permuted columns: table mode");
  static_assert(!MORE || (!TABLE && NX == 1 + CFAST), "covariate chunks: exact mode with all CFAST in-loop panels");
  __shared__ dpair s_lod[BLMM_LOD_TABLE_N];
  __shared__ int s_perm[PERM ? 16 * W2 * MB : 1];
  // PV: -log10 p (one degree of freedom) as a second output of the epilogue (`output_pvals`, src/bulkscan.jl:154-157) -- the LOD
  // never travels to HBM and back for it (fastmath.h: fast_log10p1)
  __shared__ dpair s_pv[PV ? BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2) : 1];
  // the LOD table (read after the K loop; the barrier sits right before the epilogue): its loads go out first and are written
  // to LDS once the tile arithmetic is done and the first fragment loads are in flight -- one exposed round trip per workgroup
  // instead of two or three.  (64-thread workgroups, an A/B variant, would need 33 entries per thread: plain copy loop.)
  LodStage<NT> lst;
  if constexpr (NT >= 256) lod_stage_load<NT>(lst, a.lodtab);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t tile_t, tile_i;                  // 32-bit tile arithmetic: the launchers bound nwg by 2^31
  int perm_st = -1;
  if constexpr (PERM) {
    constexpr int TW = 16 * W2 * MB;
    const uint32_t q = (uint32_t)(nwg >> 3);
    if (blockIdx.x >= (q << 3)) return;                          // workgroup-uniform; before any barrier
    const uint32_t x = blockIdx.x & 7, local = blockIdx.x >> 3;
    const uint32_t ns = ((uint32_t)*a.count + TW - 1) / TW;
    const uint64_t NS = (uint64_t)ns * (uint32_t)ntile_i;
    const uint32_t sS = (uint32_t)((x * NS) >> 3), cS = (uint32_t)(((x + 1) * NS) >> 3) - sS;
    if (local >= cS) return;
    tile_of32(sS + local, ns, (uint32_t)ntile_i, tile_t, tile_i);
    tile_t += (uint32_t)(a.col0 / TW);
    if (threadIdx.x < TW) perm_st = a.perm[(int64_t)tile_t * TW + threadIdx.x];
  } else {
    const uint32_t bid = (uint32_t)xcd_swizzle(blockIdx.x, nwg);
    tile_of32(bid, (uint32_t)nwg / (uint32_t)ntile_i, (uint32_t)ntile_i, tile_t, tile_i);
  }
  const int wt = (W2 == 2) ? (wave >> 1) : 0, wi = (W2 == 2) ? (wave & 1) : 0;
  const int64_t t0 = (int64_t)tile_t * (16 * W2 * MB) + wt * (16 * MB);
  const int64_t i0 = (int64_t)tile_i * (16 * W2 * NB) + wi * (16 * NB);
  const int r = lane & 15, kk = lane >> 4;

  d4 acc[NP][MB][NB];
  auto stage_and_zero = [&]() {
    if constexpr (NT >= 256) {
      lod_stage_store<NT>(lst, s_lod, a.lodc[0]);
    } else {
      for (int i = threadIdx.x; i < BLMM_LOD_TABLE_N; i += NT)
        s_lod[i] = (dpair){a.lodtab[2 * i], a.lodc[0] * a.lodtab[2 * i + 1]};
    }
    if constexpr (PERM) {
      if (threadIdx.x < 16 * W2 * MB) s_perm[threadIdx.x] = perm_st;
    }
    if constexpr (PV) {
      const dpair* g = reinterpret_cast<const dpair*>(a.pvtab);
      for (int i = threadIdx.x; i < BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2); i += NT) s_pv[i] = g[i];
    }
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[q][mb][nb] = (d4){0, 0, 0, 0};
  };

  // uniform tile bases (blockIdx-derived) for the descriptors; per-lane byte offsets (wave, lane) in voffset
  const double* PA = a.P + (int64_t)tile_t * (16 * W2 * MB);
  const double* PB = a.Xt + (int64_t)tile_i * (16 * W2 * NB);
  const uint32_t voffA = (uint32_t)(((int64_t)kk * a.ldp + wt * (16 * MB) + MB * r) * 8);
  const uint32_t voffB = (uint32_t)(((int64_t)kk * a.ldx + wi * (16 * NB)) * 8) + mvoff<NB>(r);
  const int64_t sa = 4 * a.ldp, sb = 4 * a.ldx;

  // K loop, two fragment sets: the loads of step ks+1 are issued before the MFMAs of step ks and are only waited
  // for after them (sched_barrier keeps hipcc from sinking the loads below the MFMA block)
  auto load_set = [&](double (&A)[NP][MB], double (&B)[NB], int step) {
#pragma unroll
    for (int q = 0; q < NP; ++q) bufload<MB>(A[q], make_srd(PA + q * a.pstride + step * sa), voffA);
    bufload_m<NB>(B, make_srd(PB + step * sb), voffB);
  };
  auto mfma_set = [&](const double (&A)[NP][MB], const double (&B)[NB]) {
    double b2[NB];
    if constexpr (NX > 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) b2[nb] = B[nb] * B[nb];
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc[0][mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0][mb], B[nb], acc[0][mb][nb], 0, 0, 0);
        if constexpr (NX > 0) {
          acc[1][mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[1][mb], b2[nb], acc[1][mb][nb], 0, 0, 0);
#pragma unroll
          for (int q = 2; q < NP; ++q)
            acc[q][mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[q][mb], B[nb], acc[q][mb][nb], 0, 0, 0);
        }
      }
  };
  if constexpr (MORE) {
    stage_and_zero();
    for (int q0 = CFAST; q0 < a.c; q0 += CFAST) {
      const int nq = (a.c - q0 < CFAST) ? a.c - q0 : CFAST;     // wave-uniform
      for (int step = 0; step < a.ks; ++step) {
        double A[CFAST][MB], B[NB];
        bufload_m<NB>(B, make_srd(PB + step * sb), voffB);
#pragma unroll
        for (int q = 0; q < CFAST; ++q)
          if (q < nq) bufload<MB>(A[q], make_srd(PA + (2 + q0 + q) * a.pstride + step * sa), voffA);
#pragma unroll
        for (int q = 0; q < CFAST; ++q)
          if (q < nq) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[2 + q][mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[q][mb], B[nb], acc[2 + q][mb][nb], 0, 0, 0);
          }
      }
#pragma unroll
      for (int q = 0; q < CFAST; ++q)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
              acc[1][mb][nb][reg] = fma(-acc[2 + q][mb][nb][reg], acc[2 + q][mb][nb][reg], acc[1][mb][nb][reg]);
            acc[2 + q][mb][nb] = (d4){0, 0, 0, 0};
          }
    }
  }
  double a0[NP][MB], b0[NB], a1[NP][MB], b1[NB];
  load_set(a0, b0, 0);
  if constexpr (!MORE) {
    __builtin_amdgcn_sched_barrier(0);
    stage_and_zero();       // under the first fragments' round trip
    __builtin_amdgcn_sched_barrier(0);
  }
  int ks = 0;
  for (; ks + 2 <= a.ks; ks += 2) {
    load_set(a1, b1, ks + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_set(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    load_set(a0, b0, (ks + 2 < a.ks) ? ks + 2 : ks);  // unconditional (clamped): keeps the vmcnt bookkeeping static
    __builtin_amdgcn_sched_barrier(0);
    mfma_set(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (ks < a.ks) mfma_set(a0, b0);

  // ---- epilogue: projection, normalisation, r -> LOD, 32-byte stores ---------------------------------
  __syncthreads();
  const double scale = a.lodc[0];
  const LodPoly5 lp = lod_poly5_of(a.lodc);
  int nnan = 0;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      int64_t trait = t0 + MB * (kk + 4 * reg) + mb;
      if constexpr (PERM) {
        trait = s_perm[wt * (16 * MB) + MB * (kk + 4 * reg) + mb];
        if (trait < 0) continue;
      } else {
        if (trait >= a.m) continue;
      }
      // (prefetching these before the K loop costs 64 VGPRs = half the occupancy: 1.15 ms instead of 0.96 ms at BXD)
      double sc[NB];
      if constexpr (TABLE) {
        const int64_t b = a.bin ? (int64_t)a.bin[trait] : 0;
        loadv_m<NB>(sc, a.isx + b * a.ld_isx + i0, r);
      }
      // the NB outputs of a row are independent chains; u outside the LOD table's range (LOD beyond ~1.2 n / 2, r^2 >= 1 --
      // +Inf / DomainError in Julia, NaN here --, NaN) is rare and handled per ROW behind one branch
      double uv[NB], out[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const double num = acc[0][mb][nb][reg];
        double r2;
        if constexpr (TABLE) {
          const double rr = num * sc[nb];
          r2 = rr * rr;
        } else {
          double xx = acc[1][mb][nb][reg];
#pragma unroll
          for (int q = 2; q < NP; ++q) xx = fma(-acc[q][mb][nb][reg], acc[q][mb][nb][reg], xx);
          r2 = (num * num) * fast_rcp1(xx);
        }
        uv[nb] = 1.0 - r2;   // r2lod (src/bulkscan_helpers.jl:22-24): -(n/2) * log10(1.0 - r^2), same operation order
      }
      bool ok = true;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) { out[nb] = fast_lod5(uv[nb], s_lod, lp); ok = ok && lod_fast_ok(uv[nb]); }
      if (__builtin_expect(!ok, 0)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (!lod_fast_ok(uv[nb])) out[nb] = lod_out_of_range(uv[nb], s_lod, lp, scale, i0 + mslot<NB>(r, nb) < a.p, &nnan);
      }
      if constexpr (RED) red_row<NB>(a.red, trait, i0, r, lane, out, red_valid(a.p, i0, 16 * NB));
      else store_m<NB>(a.L + trait * a.ldL + i0, r, out, a.p - i0);
      if constexpr (PV) {
        double pv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) pv[nb] = fast_log10p1(out[nb], s_pv);
        store_m<NB>(a.Pv + trait * a.ldPv + i0, r, pv, a.p - i0);
      }
    }
  if (nnan) atomicAdd((unsigned long long*)&a.stat[ST_NAN_LOD], (unsigned long long)nnan);
}

template <int NX, bool TABLE, int MB, int W2 = 2, bool MORE = false>
static int launch_scan_t(blmm_ctx* ctx, const ScanArgs& a) {
  constexpr int NB = 4;
  static_assert(16 * W2 * MB <= 128 && 16 * W2 * NB <= TILE_I, "tile constants (operands are padded to 128)");
  const int64_t ntile_t = (a.m + 16 * W2 * MB - 1) / (16 * W2 * MB);
  const int64_t ntile_i = (a.p + 16 * W2 * NB - 1) / (16 * W2 * NB);
  const int64_t nwg = ntile_t * ntile_i;
  if (nwg <= 0) return BLMM_OK;
  if (nwg > 0x7fffffffLL) return fail(ctx, BLMM_ERR_INVALID, "problem too large for one launch");
  if (a.red.pmax) {
    // reduce-in-epilogue: instantiated for the table kernel at its default tile only (the callers route everything else through a
    // resident L and the column passes)
    if constexpr (TABLE && MB == 2 && W2 == 2 && !MORE)
      hipLaunchKernelGGL((k_scan<NX, MB, NB, TABLE, W2, false, MORE, false, true>), dim3((unsigned)nwg), dim3(64 * W2 * W2), 0, ctx->stream, a, (int)ntile_i, nwg);
    else
      return fail(ctx, BLMM_ERR_UNSUPPORTED, "reduce-in-epilogue: no such instantiation of k_scan");
  } else if (a.Pv)
    hipLaunchKernelGGL((k_scan<NX, MB, NB, TABLE, W2, false, MORE, true>), dim3((unsigned)nwg), dim3(64 * W2 * W2), 0, ctx->stream, a, (int)ntile_i, nwg);
  else
    hipLaunchKernelGGL((k_scan<NX, MB, NB, TABLE, W2, false, MORE>), dim3((unsigned)nwg), dim3(64 * W2 * W2), 0, ctx->stream, a, (int)ntile_i, nwg);
  KCHECK();
  return BLMM_OK;
}

int launch_scan_exact(blmm_ctx* ctx, const ScanArgs& a, int c) {
  switch (c) {
    // accumulators per lane: (2 + c) * MB * 4 blocks * 8 VGPRs; MB = 1 beyond c = 1 keeps 2 waves/SIMD spill-free
    case 1: {
      return launch_scan_t<2, false, 2, 2>(ctx, a);
    }
    case 2: return launch_scan_t<3, false, 1>(ctx, a);
    case 3: return launch_scan_t<4, false, 1>(ctx, a);
    case 4: return launch_scan_t<5, false, 1>(ctx, a);
  }
  if (c > CFAST && c <= CMAX && a.c == c) return launch_scan_t<1 + CFAST, false, 1, 2, true>(ctx, a);
  return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
}

// ------------------------------------------------------------------------------------------------
// exact kernel, low-rank weights form (kernels_lowrank.hip):
//   phase 1 (n long):   num  = x_i' a0_j                         A = P0 panel,  B = Xt
//   phase 2 (R long):   Sxx  = (Q'x_i.^2)' c_j ,  s_q = (Q'(x_i.*z_q))' c_j      A = Cp panel,  B = T[0], T[1+q]
//   epilogue:           u = L_j^-1 s ;  r^2 = num^2 / (Sxx - |u|^2) ;  LOD
// Same tiling, fragment maps and store path as k_scan.  KR = ceil(R/4) is read from device memory (rk[1]).
// Panel column t belongs to trait perm[t] (k_lr_classify).  The tiles at the front hold the shared-weights class
// (weights = 1 to within the guard's tolerance): phase 2 is skipped and Sxx - |u|^2 is the per-marker constant den0_i.
// ------------------------------------------------------------------------------------------------
#ifdef LR_DIAG
__device__ unsigned long long g_lr_diag[24];   // per class {sum of workgroup cycles, count}; per XCD last end tick
#endif
#ifdef LR_PHASE
// per class (0: shared-weights tiles, 1: rank-R tiles) sums over the waves of {K loops, barrier wait, epilogue} cycles and the count
__device__ unsigned long long g_lr_phase[8];
#endif
template <int C, int MB, int NB, bool PV = false, bool RED = false>
__global__ void __launch_bounds__(256, 2) k_scan_lr(LrArgs la, int ntile_i, int64_t nwg) {
  static_assert(!(RED && PV), "reduce-in-epilogue: no p-value output");
  const ScanArgs& a = la.s;
#ifdef LR_DIAG
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef LR_PHASE
  const unsigned long long ph_t0 = __builtin_amdgcn_s_memtime();
#endif
  constexpr int NACC = 2 + C;
  constexpr int NL = C * (C + 1) / 2;
  constexpr int TW = 32 * MB;
  __shared__ dpair s_lod[BLMM_LOD_TABLE_N];
  __shared__ double s_li[NL][TW];           // packed L_j^-1 of the tile's traits (read in the epilogue)
  __shared__ int s_perm[TW];                // their trait numbers (-1: padding column)
  __shared__ dpair s_pv[PV ? BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2) : 1];   // PV: -log10 p as a second output (see k_scan)
  // The LOD table's loads go out first: nothing below depends on them until the LDS stores in front of the K loops, so their
  // round trip runs under the tile arithmetic and the first fragment loads (round 2 staged the table, then L^-1 / perm, then
  // fetched the first fragments: three exposed round trips at the head of every workgroup).
  LodStage<256> lst;
  lod_stage_load<256>(lst, a.lodtab);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Trait tiles in use: ns at the front (the shared-weights class, 5/8 of the matrix work of a tile of the other class) and
  // nf from tile `fo` on.  Workgroups reach the CUs of an XCD in strict round-robin order (measured: a mix of the two
  // classes inside one dispatch round runs at the pace of the slower class, whatever the share of the faster one), so
  // each XCD (blockIdx % 8) is dealt a contiguous run of shared-weights work items followed by a contiguous run of the
  // others, an eighth of either class: rounds are homogeneous, the XCDs balanced, and the items of a run follow the
  // L2-friendly walk of tile_of.  (32-bit tile arithmetic: the launcher bounds nwg by 2^31; 64-bit divisions cost ~800 scalar
  // instructions at the head of every workgroup.)
  uint32_t tile_t; uint32_t tile_i;
  bool shared_w;
  {
    const uint32_t q = (uint32_t)(nwg >> 3);
    if (blockIdx.x >= (q << 3)) return;                          // workgroup-uniform; before any barrier
    const uint32_t x = blockIdx.x & 7, local = blockIdx.x >> 3;
    const uint32_t nsh = (uint32_t)la.rg.counts[0], noth = (uint32_t)la.rg.counts[1];
    const uint32_t tb = (uint32_t)(la.rg.col0 / TW);             // the region's first tile (col0, ncol: multiples of 64)
    const uint32_t ncolt = (uint32_t)(la.rg.ncol / TW);
    const uint32_t ns = (nsh + TW - 1) / TW, fo = ((uint32_t)la.rg.ncol - noth) / TW, nf = ncolt - fo;
    const uint64_t NS = la.skip_shared ? 0 : (uint64_t)ns * (uint32_t)ntile_i, NF = (uint64_t)nf * (uint32_t)ntile_i;
    const uint32_t sS = (uint32_t)((x * NS) >> 3), cS = (uint32_t)(((x + 1) * NS) >> 3) - sS;
    const uint32_t sF = (uint32_t)((x * NF) >> 3), cF = (uint32_t)(((x + 1) * NF) >> 3) - sF;
    if (local < cS) {
      shared_w = true;
      tile_of32(sS + local, ns, (uint32_t)ntile_i, tile_t, tile_i);
      tile_t += tb;
    } else if (local - cS < cF) {
      shared_w = false;
      tile_of32(sF + local - cS, nf, (uint32_t)ntile_i, tile_t, tile_i);
      tile_t += tb + fo;
    } else {
      return;
    }
  }
  // L^-1 and the trait numbers of the tile (Ls is padded to ldp, a multiple of the tile): fetched here, stored to LDS in front
  // of the K loops -- in the epilogue these loads would sit behind the stores of the previous row (possible aliasing)
  static_assert(NL * TW <= 2 * 256, "L^-1 staging: two elements per thread");
  double li_st[2]; int perm_st = -1;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = (int)threadIdx.x + 256 * u;
    li_st[u] = (e < NL * TW) ? la.Ls[(int64_t)(e / TW) * a.ldp + (int64_t)tile_t * TW + (e % TW)] : 0.0;
  }
  if (threadIdx.x < TW) perm_st = la.perm[(int64_t)tile_t * TW + threadIdx.x];
  const int wt = wave >> 1, wi = wave & 1;
  const int64_t i0 = (int64_t)tile_i * (32 * NB) + wi * (16 * NB);
  const int r = lane & 15, kk = lane >> 4;

  const double* PA = a.P + (int64_t)tile_t * TW;
  const double* PB = a.Xt + (int64_t)tile_i * (32 * NB);
  const double* PC = la.Cp + (int64_t)tile_t * TW;
  // the tile's segment of the heritability axis (its own basis: rank and marker-side products); runs are laid out from the region's end
  int sgi = 0;
  if (la.seg.S > 1 && !shared_w)
    sgi = lr_seg_of(la.rg.ncol - 1 - ((int64_t)tile_t * TW - la.rg.col0), la.rg.segcnt, la.seg.S);
  const double* PT = la.T + (int64_t)sgi * (1 + C) * la.tstride + (int64_t)tile_i * (32 * NB);
  const uint32_t voffA = (uint32_t)(((int64_t)kk * a.ldp + wt * (16 * MB) + MB * r) * 8);
  const uint32_t voffB = (uint32_t)(((int64_t)kk * a.ldx + wi * (16 * NB)) * 8) + mvoff<NB>(r);
  const int64_t sa = 4 * a.ldp, sb = 4 * a.ldx;

  d4 acc[NACC][MB][NB];

  // ---- phase 1: num, two K steps per fragment set (a.ks is even: the K dimension is padded to 8);
  //      phase 2: Sxx and s_q over the weight basis.  The first fragment set of phase 2 is fetched under the last
  //      MFMAs of phase 1, so the only exposed load latency of a tile is its very first set.
  auto load1 = [&](double (&A)[2][MB], double (&B)[2][NB], int step2) {
#ifdef LR_NOLOAD    // diagnostic: the matrix pipe alone (operands from registers)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int u = 0; u < MB; ++u) { A[h][u] = (double)(lane + step2 + u); asm volatile("" : "+v"(A[h][u])); }
#pragma unroll
      for (int u = 0; u < NB; ++u) { B[h][u] = (double)(lane - step2 + u); asm volatile("" : "+v"(B[h][u])); }
    }
    return;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bufload<MB>(A[h], make_srd(PA + (int64_t)(2 * step2 + h) * sa), voffA);
      bufload_m<NB>(B[h], make_srd(PB + (int64_t)(2 * step2 + h) * sb), voffB);
    }
  };
  auto mfma1 = [&](const double (&A)[2][MB], const double (&B)[2][NB]) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[0][mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[h][mb], B[h][nb], acc[0][mb][nb], 0, 0, 0);
  };
  auto load2 = [&](double (&A)[MB], double (&B)[1 + C][NB], int step) {
#ifdef LR_NOLOAD
#pragma unroll
    for (int u = 0; u < MB; ++u) { A[u] = (double)(lane + step + u); asm volatile("" : "+v"(A[u])); }
#pragma unroll
    for (int q = 0; q <= C; ++q)
#pragma unroll
      for (int u = 0; u < NB; ++u) { B[q][u] = (double)(lane - step + u + q); asm volatile("" : "+v"(B[q][u])); }
    return;
#endif
    bufload<MB>(A, make_srd(PC + (int64_t)step * sa), voffA);
#pragma unroll
    for (int q = 0; q <= C; ++q) bufload_m<NB>(B[q], make_srd(PT + q * la.tstride + (int64_t)step * sb), voffB);
  };
  auto mfma2 = [&](const double (&A)[MB], const double (&B)[1 + C][NB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q <= C; ++q)
          acc[1 + q][mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[mb], B[q][nb], acc[1 + q][mb][nb], 0, 0, 0);
  };
  const int K2 = a.ks / 2;
  const int KR = shared_w ? 0 : la.rk[4 * sgi + 1];
  double c0[MB], d0[1 + C][NB];
  {
    double a0[2][MB], b0[2][NB], a1[2][MB], b1[2][NB];
    load1(a0, b0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // the staged tables go to LDS while the first fragments are on their way (read after the barrier in front of the epilogue)
    lod_stage_store<256>(lst, s_lod, a.lodc[0]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = (int)threadIdx.x + 256 * u;
      if (e < NL * TW) s_li[e / TW][e % TW] = li_st[u];
    }
    if (threadIdx.x < TW) s_perm[threadIdx.x] = perm_st;
    if constexpr (PV) {
      const dpair* g = reinterpret_cast<const dpair*>(a.pvtab);
      for (int i = threadIdx.x; i < BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2); i += 256) s_pv[i] = g[i];
    }
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[q][mb][nb] = (d4){0, 0, 0, 0};
    __builtin_amdgcn_sched_barrier(0);
    int s2 = 0;
    while (s2 + 2 < K2) {   // a following pair exists
      load1(a1, b1, s2 + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma1(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load1(a0, b0, s2 + 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma1(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      s2 += 2;
    }
    if (K2 - s2 == 2) {
      load1(a1, b1, s2 + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma1(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (KR > 0) load2(c0, d0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma1(a1, b1);
    } else {
      if (KR > 0) load2(c0, d0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma1(a0, b0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    double c1[MB], d1[1 + C][NB];
    int ks = 0;
    for (; ks + 2 <= KR; ks += 2) {
      load2(c1, d1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma2(c0, d0);
      __builtin_amdgcn_sched_barrier(0);
      load2(c0, d0, (ks + 2 < KR) ? ks + 2 : ks);
      __builtin_amdgcn_sched_barrier(0);
      mfma2(c1, d1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ks < KR) mfma2(c0, d0);
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------
#ifdef LR_PHASE
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long ph_t2 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#endif
  __syncthreads();
#ifdef LR_PHASE
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long ph_t3 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#endif
  const double scale = a.lodc[0];
  const LodPoly5 lp = lod_poly5_of(a.lodc);
  int nnan = 0;
  // Two instances of the epilogue behind one workgroup-uniform branch.  A shared-weights tile multiplies by the per-marker
  // 1 / sqrt(Sxx - |u|^2) of the unweighted model (left by k_lr_tpanels): no L_j^-1, no reciprocal -- the epilogue is fp64
  // VALU time the matrix pipe cannot overlap.  (By default these tiles go through the leaner table kernel instead:
  // launch_scan_shared; this path serves BLMM_LR_LEAN=0.)
  // The NB outputs of a row are NB independent chains; u outside the LOD table's range (LOD beyond ~1.2 n / 2, r^2 >= 1, NaN)
  // is rare and handled per ROW behind one branch.
  auto epilogue = [&](auto SH) {
    constexpr bool SHW = decltype(SH)::value;
    double rd[NB];
    if constexpr (SHW) loadv_m<NB>(rd, la.den0 + i0, r);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t trait = s_perm[wt * (16 * MB) + MB * (kk + 4 * reg) + mb];
        if (trait < 0) continue;
        double li[NL];
        if constexpr (!SHW) {
#pragma unroll
          for (int e = 0; e < NL; ++e) li[e] = s_li[e][wt * (16 * MB) + MB * (kk + 4 * reg) + mb];
        }
        double uv[NB], out[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const double num = acc[0][mb][nb][reg];
          double r2;
          if constexpr (SHW) {
            const double rr = num * rd[nb];
            r2 = rr * rr;
          } else {
            double xx = acc[1][mb][nb][reg];
#pragma unroll
            for (int q = 0; q < C; ++q) {
              double u = 0.0;
#pragma unroll
              for (int e = 0; e <= q; ++e) u = fma(li[q * (q + 1) / 2 + e], acc[2 + e][mb][nb][reg], u);
              xx = fma(-u, u, xx);
            }
            r2 = (num * num) * fast_rcp1(xx);
          }
          uv[nb] = 1.0 - r2;   // r2lod (src/bulkscan_helpers.jl:22-24): -(n/2) * log10(1.0 - r^2)
        }
        bool ok = true;
#ifdef LR_NOEPI     // diagnostic: the K loops and the stores alone
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) out[nb] = acc[0][mb][nb][reg] + acc[1][mb][nb][reg] + acc[2][mb][nb][reg];
#else
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { out[nb] = fast_lod5(uv[nb], s_lod, lp); ok = ok && lod_fast_ok(uv[nb]); }
        if (__builtin_expect(!ok, 0)) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            if (!lod_fast_ok(uv[nb])) out[nb] = lod_out_of_range(uv[nb], s_lod, lp, scale, i0 + mslot<NB>(r, nb) < a.p, &nnan);
        }
#endif
#ifdef LR_NOSTORE   // diagnostic: everything but the stream of stores
        if (out[0] + out[1] + out[2] + out[3] != 1.2345e-300) continue;
#endif
        if constexpr (RED) red_row<NB>(a.red, trait, i0, r, lane, out, red_valid(a.p, i0, 16 * NB));
        else store_m<NB>(a.L + trait * a.ldL + i0, r, out, a.p - i0);
        if constexpr (PV) {
          double pv[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) pv[nb] = fast_log10p1(out[nb], s_pv);
          store_m<NB>(a.Pv + trait * a.ldPv + i0, r, pv, a.p - i0);
        }
      }
  };
  if (shared_w) epilogue(std::true_type{}); else epilogue(std::false_type{});
  if (nnan) atomicAdd((unsigned long long*)&a.stat[ST_NAN_LOD], (unsigned long long)nnan);
#ifdef LR_PHASE
  {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long ph_t4 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      unsigned long long* z = g_lr_phase + (shared_w ? 0 : 4);
      atomicAdd(&z[0], ph_t2 - ph_t0); atomicAdd(&z[1], ph_t3 - ph_t2); atomicAdd(&z[2], ph_t4 - ph_t3); atomicAdd(&z[3], 1ull);
    }
  }
#endif
#ifdef LR_DIAG
  if (threadIdx.x == 0) {
    atomicAdd(&g_lr_diag[shared_w ? 0 : 2], __builtin_amdgcn_s_memtime() - diag_t0);
    atomicAdd(&g_lr_diag[shared_w ? 1 : 3], 1ull);
    const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 15u;   // HW_REG_XCC_ID[3:0]
    atomicMax(&g_lr_diag[8 + (xcc & 7)], __builtin_amdgcn_s_memrealtime());
    atomicAdd(&g_lr_diag[16 + (xcc & 7)], 1ull);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_scan_lr3: the rank-R tiles of k_scan_lr<1, 2, 4> at THREE waves per SIMD (the default for c = 1 without the p-value output and
// with the shared-weights class in the table kernel; BLMM_LR3=0: k_scan_lr).  PV: -log10 p as a second output.  Phase 2 comes FIRST and in two halves over the wave's trait blocks: a half
// accumulates Sxx and s for four 16 x 16 blocks (64 VGPRs), converts them to 1 / (Sxx - u^2) (32 VGPRs) and lets them go; then
// phase 1 accumulates num (64 VGPRs) beside the 64 of the reciprocals -- 128 accumulator registers at the peak where k_scan_lr
// holds 192 -- and the epilogue is r^2 = num^2 * that reciprocal.  Same arithmetic per output, same bits.
// ------------------------------------------------------------------------------------------------
template <int C, bool PV, bool RED = false>
__global__ void __launch_bounds__(256, 3) k_scan_lr3(LrArgs la, int ntile_i, int64_t nwg) {
  static_assert(!(RED && PV), "reduce-in-epilogue: no p-value output");
  const ScanArgs& a = la.s;
  constexpr int MB = 2, NB = 4, TW = 64;
  constexpr int NL = C * (C + 1) / 2;
  // phase 2 goes chunk by chunk: (trait block mb) x (NBC marker blocks) with 1 + C accumulators per block -- four marker blocks for
  // c = 1 (64 registers), two for c = 2, 3 (48, 64) -- so that a chunk's sums and the reciprocals already made stay within three waves
  constexpr int NBC = (C == 1) ? 4 : 2;
  __shared__ dpair s_lod[BLMM_LOD_TABLE_N];
  __shared__ double s_li[NL][TW];
  __shared__ int s_perm[TW];
  __shared__ dpair s_pv[PV ? BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2) : 1];   // PV: -log10 p as a second output (see k_scan)
  LodStage<256> lst;
  lod_stage_load<256>(lst, a.lodtab);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t tile_t, tile_i;
  {
    const uint32_t q = (uint32_t)(nwg >> 3);
    if (blockIdx.x >= (q << 3)) return;
    const uint32_t x = blockIdx.x & 7, local = blockIdx.x >> 3;
    const uint32_t noth = (uint32_t)la.rg.counts[1];
    const uint32_t tb = (uint32_t)(la.rg.col0 / TW);
    const uint32_t ncolt = (uint32_t)(la.rg.ncol / TW);
    const uint32_t fo = ((uint32_t)la.rg.ncol - noth) / TW, nf = ncolt - fo;
    const uint64_t NF = (uint64_t)nf * (uint32_t)ntile_i;
    const uint32_t sF = (uint32_t)((x * NF) >> 3), cF = (uint32_t)(((x + 1) * NF) >> 3) - sF;
    if (local >= cF) return;
    tile_of32(sF + local, nf, (uint32_t)ntile_i, tile_t, tile_i);
    tile_t += tb + fo;
  }
  static_assert(NL * TW <= 2 * 256, "L^-1 staging: two elements per thread");
  double li_st[2]; int perm_st = -1;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = (int)threadIdx.x + 256 * u;
    li_st[u] = (e < NL * TW) ? la.Ls[(int64_t)(e / TW) * a.ldp + (int64_t)tile_t * TW + (e % TW)] : 0.0;
  }
  if (threadIdx.x < TW) perm_st = la.perm[(int64_t)tile_t * TW + threadIdx.x];
  const int wt = wave >> 1, wi = wave & 1;
  const int64_t i0 = (int64_t)tile_i * (32 * NB) + wi * (16 * NB);
  const int r = lane & 15, kk = lane >> 4;
  const double* PA = a.P + (int64_t)tile_t * TW;
  const double* PB = a.Xt + (int64_t)tile_i * (32 * NB);
  const double* PC = la.Cp + (int64_t)tile_t * TW;
  int sgi = 0;
  if (la.seg.S > 1) sgi = lr_seg_of(la.rg.ncol - 1 - ((int64_t)tile_t * TW - la.rg.col0), la.rg.segcnt, la.seg.S);
  const double* PT = la.T + (int64_t)sgi * (1 + C) * la.tstride + (int64_t)tile_i * (32 * NB);
  const uint32_t voffA = (uint32_t)(((int64_t)kk * a.ldp + wt * (16 * MB) + MB * r) * 8);
  const uint32_t voffB = (uint32_t)(((int64_t)kk * a.ldx + wi * (16 * NB)) * 8) + mvoff<NB>(r);
  const int64_t sa = 4 * a.ldp, sb = 4 * a.ldx;
  const int KR = la.rk[4 * sgi + 1];
  // first fragments of phase 2 on their way, then the staged tables to LDS
  double c0[MB], d0[1 + C][NBC];
  auto load2 = [&](double (&A)[MB], double (&B)[1 + C][NBC], int step, int nb0) {
    bufload<MB>(A, make_srd(PC + (int64_t)step * sa), voffA);
#pragma unroll
    for (int q = 0; q <= C; ++q) bufload<NBC>(B[q], make_srd(PT + q * la.tstride + (int64_t)step * sb), voffB + (uint32_t)(8 * nb0));
  };
  if (KR > 0) load2(c0, d0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  lod_stage_store<256>(lst, s_lod, a.lodc[0]);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = (int)threadIdx.x + 256 * u;
    if (e < NL * TW) s_li[e / TW][e % TW] = li_st[u];
  }
  if (threadIdx.x < TW) s_perm[threadIdx.x] = perm_st;
  if constexpr (PV) {
    const dpair* g = reinterpret_cast<const dpair*>(a.pvtab);
    for (int i = threadIdx.x; i < BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2); i += 256) s_pv[i] = g[i];
  }
  __syncthreads();
  d4 den[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb0 = 0; nb0 < NB; nb0 += NBC) {
      d4 sm[1 + C][NBC];
#pragma unroll
      for (int q = 0; q <= C; ++q)
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb) sm[q][nb] = (d4){0, 0, 0, 0};
      // one fragment set: the next step's loads go out right behind the MFMAs that read this step's (the matrix pipe drains them
      // meanwhile, and two other waves share the SIMD); behind a chunk's last step, the first set of the next chunk
      for (int ks = 0; ks < KR; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
          for (int q = 0; q <= C; ++q) sm[q][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(c0[mb], d0[q][nb], sm[q][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < KR) load2(c0, d0, ks + 1, nb0);
        else if (nb0 + NBC < NB) load2(c0, d0, 0, nb0 + NBC);
        else if (mb + 1 < MB) load2(c0, d0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        double li[NL];
#pragma unroll
        for (int e = 0; e < NL; ++e) li[e] = s_li[e][wt * (16 * MB) + MB * (kk + 4 * reg) + mb];
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb) {
          double xx = sm[0][nb][reg];
#pragma unroll
          for (int q = 0; q < C; ++q) {
            double u = 0.0;
#pragma unroll
            for (int e = 0; e <= q; ++e) u = fma(li[q * (q + 1) / 2 + e], sm[1 + e][nb][reg], u);
            xx = fma(-u, u, xx);
          }
          den[mb][nb0 + nb][reg] = fast_rcp1(xx);
        }
      }
      // pin the reciprocals HERE: left alone the compiler sinks the whole conversion into the epilogue, next to its uses, and keeps
      // the registers of Sxx and s alive through phase 1 (142 spilled dwords)
#pragma unroll
      for (int nb = 0; nb < NBC; ++nb) asm volatile("" : "+v"(den[mb][nb0 + nb]));
    }
  // ---- phase 1: num over n, one K step per fragment set
  d4 num[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) num[mb][nb] = (d4){0, 0, 0, 0};
  {
    double a0[MB], b0[NB];
    auto load1 = [&](double (&A)[MB], double (&B)[NB], int step) {
      bufload<MB>(A, make_srd(PA + (int64_t)step * sa), voffA);
      bufload_m<NB>(B, make_srd(PB + (int64_t)step * sb), voffB);
    };
    auto mf1 = [&](const double (&A)[MB], const double (&B)[NB]) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) num[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[mb], B[nb], num[mb][nb], 0, 0, 0);
    };
    const int K = a.ks;
    load1(a0, b0, 0);
    for (int s1 = 0; s1 < K; ++s1) {
      __builtin_amdgcn_sched_barrier(0);
      mf1(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (s1 + 1 < K) load1(a0, b0, s1 + 1);
    }
  }
  // ---- epilogue
  const double scale = a.lodc[0];
  const LodPoly5 lp = lod_poly5_of(a.lodc);
  int nnan = 0;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t trait = s_perm[wt * (16 * MB) + MB * (kk + 4 * reg) + mb];
      if (trait < 0) continue;
      double uv[NB], out[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const double nm = num[mb][nb][reg];
        const double r2 = (nm * nm) * den[mb][nb][reg];   // (its own statement, as in k_scan_lr: -ffp-contract=on would fuse `1 - a * b` into one fma)
        uv[nb] = 1.0 - r2;
      }
      bool ok = true;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) { out[nb] = fast_lod5(uv[nb], s_lod, lp); ok = ok && lod_fast_ok(uv[nb]); }
      if (__builtin_expect(!ok, 0)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (!lod_fast_ok(uv[nb])) out[nb] = lod_out_of_range(uv[nb], s_lod, lp, scale, i0 + mslot<NB>(r, nb) < a.p, &nnan);
      }
      if constexpr (RED) red_row<NB>(a.red, trait, i0, r, lane, out, red_valid(a.p, i0, 16 * NB));
      else store_m<NB>(a.L + trait * a.ldL + i0, r, out, a.p - i0);
      if constexpr (PV) {
        double pv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) pv[nb] = fast_log10p1(out[nb], s_pv);
        store_m<NB>(a.Pv + trait * a.ldPv + i0, r, pv, a.p - i0);
      }
    }
  if (nnan) atomicAdd((unsigned long long*)&a.stat[ST_NAN_LOD], (unsigned long long)nnan);
}

template <int C, int MB>
static int launch_scan_lr_t(blmm_ctx* ctx, const LrArgs& la) {
  constexpr int NB = 4;
  const ScanArgs& a = la.s;
  const int64_t ntile_t = (a.m + 32 * MB - 1) / (32 * MB);      // a.m: an upper bound of the panel columns in tiles that hold traits
  const int64_t ntile_i = (a.p + 32 * NB - 1) / (32 * NB);
  if (ntile_t * ntile_i <= 0) return BLMM_OK;
  const int64_t nwg = (ntile_t * ntile_i + 16 + 7) / 8 * 8;      // every XCD's share of either class rounds up: + 2 items each
  if (nwg > 0x7fffffffLL) return fail(ctx, BLMM_ERR_INVALID, "problem too large for one launch");
#ifdef LR_DIAG
  unsigned long long z[24] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lr_diag), z, sizeof(z));
  (void)hipStreamSynchronize(ctx->stream);
#endif
#ifdef LR_PHASE
  unsigned long long zp[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lr_phase), zp, sizeof(zp));
  (void)hipStreamSynchronize(ctx->stream);
#endif
  // k_scan_lr3 (three waves per SIMD) is the default for c = 1 and n <= 128; BLMM_LR3=0: k_scan_lr (A/B testing).
  // One box, four alternating rounds: scan 1.164-1.183 against 1.206-1.240 ms, step 1.636-1.654 against 1.674-1.714.
  static const bool lr3 = !(dev_env("BLMM_LR3") && dev_env("BLMM_LR3")[0] == '0');
  if (a.red.pmax && lr3 && C == 1 && MB == 2 && la.skip_shared && a.n <= 128)
    hipLaunchKernelGGL((k_scan_lr3<1, false, true>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, la, (int)ntile_i, nwg);
  else if (a.red.pmax)
    hipLaunchKernelGGL((k_scan_lr<C, MB, NB, false, true>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, la, (int)ntile_i, nwg);
  else if (a.Pv && lr3 && C == 1 && MB == 2 && la.skip_shared && a.n <= 128)
    hipLaunchKernelGGL((k_scan_lr3<1, true>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, la, (int)ntile_i, nwg);
  else if (a.Pv)
    hipLaunchKernelGGL((k_scan_lr<C, MB, NB, true>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, la, (int)ntile_i, nwg);
  else if (lr3 && C == 1 && MB == 2 && la.skip_shared && a.n <= 128)   // beyond: phase 1 is long and its single fragment set shows (n = 200: +1.7 %, n = 500: +4 %; n = 124: -1 %, n = 79: -3.8 % of the scan)
    hipLaunchKernelGGL((k_scan_lr3<1, false>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, la, (int)ntile_i, nwg);
  else
    hipLaunchKernelGGL((k_scan_lr<C, MB, NB>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, la, (int)ntile_i, nwg);
#ifdef LR_PHASE
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipMemcpyFromSymbol(zp, HIP_SYMBOL(g_lr_phase), sizeof(zp));
  for (int cls = 0; cls < 2; ++cls)
    if (zp[4 * cls + 3])
      fprintf(stderr, "lr phase: %s tiles: %llu waves, avg cycles K loops %.0f | barrier wait %.0f | epilogue %.0f\n", cls ? "rank-R" : "shared",
              zp[4 * cls + 3], (double)zp[4 * cls] / zp[4 * cls + 3], (double)zp[4 * cls + 1] / zp[4 * cls + 3], (double)zp[4 * cls + 2] / zp[4 * cls + 3]);
#endif
#ifdef LR_DIAG
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_lr_diag), sizeof(z));
  unsigned long long mn = ~0ull;
  for (int x = 0; x < 8; ++x) if (z[8 + x] && z[8 + x] < mn) mn = z[8 + x];
  fprintf(stderr, "lr diag: shared wgs %llu avg cycles %.0f | full wgs %llu avg cycles %.0f | xcd last-end ticks(100MHz) rel:", z[1],
          z[1] ? (double)z[0] / z[1] : 0.0, z[3], z[3] ? (double)z[2] / z[3] : 0.0);
  for (int x = 0; x < 8; ++x) fprintf(stderr, " %llu(%llu)", z[8 + x] - mn, z[16 + x]);
  fprintf(stderr, "\n");
#endif
  KCHECK();
  return BLMM_OK;
}

// c = 2, 3 through the three-wave kernel (64-trait tiles; k_scan_lr<C, 1> holds 16 x 64 per wave at two waves): same conditions as for
// c = 1 -- n <= 128, the shared-weights class in the table kernel, BLMM_LR3 != 0
template <int C>
static bool lr3_covariates(blmm_ctx*, const LrArgs& la) {
  static const bool lr3 = !(dev_env("BLMM_LR3") && dev_env("BLMM_LR3")[0] == '0');
  return lr3 && la.skip_shared && la.s.n <= 128;
}
template <int C>
static int launch_scan_lr3_c(blmm_ctx* ctx, const LrArgs& la) {
  const ScanArgs& a = la.s;
  const int64_t ntile_t = (a.m + 63) / 64, ntile_i = (a.p + 127) / 128;
  if (ntile_t * ntile_i <= 0) return BLMM_OK;
  const int64_t nwg = (ntile_t * ntile_i + 16 + 7) / 8 * 8;
  if (nwg > 0x7fffffffLL) return fail(ctx, BLMM_ERR_INVALID, "problem too large for one launch");
  if (a.red.pmax) hipLaunchKernelGGL((k_scan_lr3<C, false, true>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, la, (int)ntile_i, nwg);
  else if (a.Pv) hipLaunchKernelGGL((k_scan_lr3<C, true>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, la, (int)ntile_i, nwg);
  else hipLaunchKernelGGL((k_scan_lr3<C, false>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, la, (int)ntile_i, nwg);
  KCHECK();
  return BLMM_OK;
}

int launch_scan_lr(blmm_ctx* ctx, const LrArgs& la) {
  switch (la.c) {
    case 1: {
      static const int mb1 = dev_env("BLMM_LR_MB1") ? atoi(dev_env("BLMM_LR_MB1")) : 0;
      return mb1 ? launch_scan_lr_t<1, 1>(ctx, la) : launch_scan_lr_t<1, 2>(ctx, la);
    }
    case 2: return lr3_covariates<2>(ctx, la) ? launch_scan_lr3_c<2>(ctx, la) : launch_scan_lr_t<2, 1>(ctx, la);
    case 3: return lr3_covariates<3>(ctx, la) ? launch_scan_lr3_c<3>(ctx, la) : launch_scan_lr_t<3, 1>(ctx, la);
  }
  return fail(ctx, BLMM_ERR_UNSUPPORTED, BLMM_C_ERR);
}

int launch_scan_shared(blmm_ctx* ctx, const ScanArgs& a) {
  constexpr int MB = 2, NB = 4, W2 = 2;
  const int64_t ntile_t = (a.m + 16 * W2 * MB - 1) / (16 * W2 * MB);     // a.m: an upper bound of the class (sizes the grid)
  const int64_t ntile_i = (a.p + 16 * W2 * NB - 1) / (16 * W2 * NB);
  if (ntile_t * ntile_i <= 0) return BLMM_OK;
  const int64_t nwg = (ntile_t * ntile_i + 8 + 7) / 8 * 8;              // every XCD's share rounds up
  if (nwg > 0x7fffffffLL) return fail(ctx, BLMM_ERR_INVALID, "problem too large for one launch");
  if (a.red.pmax)
    hipLaunchKernelGGL((k_scan<0, MB, NB, true, W2, true, false, false, true>), dim3((unsigned)nwg), dim3(64 * W2 * W2), 0, ctx->stream, a, (int)ntile_i, nwg);
  else if (a.Pv)
    hipLaunchKernelGGL((k_scan<0, MB, NB, true, W2, true, false, true>), dim3((unsigned)nwg), dim3(64 * W2 * W2), 0, ctx->stream, a, (int)ntile_i, nwg);
  else
    hipLaunchKernelGGL((k_scan<0, MB, NB, true, W2, true>), dim3((unsigned)nwg), dim3(64 * W2 * W2), 0, ctx->stream, a, (int)ntile_i, nwg);
  KCHECK();
  return BLMM_OK;
}

int launch_scan_table(blmm_ctx* ctx, const ScanArgs& a) {
  static const int mb = dev_env("BLMM_TABLE_MB") ? atoi(dev_env("BLMM_TABLE_MB")) : 2;
  if (mb == 4) return launch_scan_t<0, true, 4>(ctx, a);
  if (mb == 1) return launch_scan_t<0, true, 1>(ctx, a);
  return launch_scan_t<0, true, 2>(ctx, a);
}

// ------------------------------------------------------------------------------------------------
// alt-grid: for every (trait, marker) the maximum over the h2 grid of  logL1_g = ln10*LOD_g + Ell[g, j];
//   L = (max_g logL1_g - max_g Ell[g, j]) / ln10,  h2_panel = grid value at the first arg-max
// (src/bulkscan.jl:495-522).  counter_quirk reproduces tmax!'s improvement counter.
// Panels: P[g][k][j] = panel 0 under h2 = grid[g].
// Round 4: the running maximum is kept in a MONOTONE IMAGE of logL1 that needs no logarithm per grid point.  With l0_j = max_g
// Ell[g, j] and c[g, j] = exp(-(2/n) (Ell[g, j] - l0_j)) (k_alt_ctab, G x m values),
//     logL1_g - l0_j = -(n/2) ln(1 - r_g^2) + (Ell[g, j] - l0_j) = -(n/2) ln( (1 - r_g^2) c[g, j] ),
// so  max_g logL1_g  <=>  min_g v_g,  v_g = (1.0 - r_g^2) c[g, j]  (strict <: the first extremum wins, as tmax!'s strict <), and
//     L = -(n/2) log10(v_min)  -- ONE table logarithm per test instead of one per grid point and test: the fold is five fp64
// operations where it was ~20 plus an LDS lookup, and every one of them is matrix-pipe time (fp64 VALU and MFMA exclude each other).
// Two grid points are ordered differently from the reference only when their logL1 agree to rounding (the tests' tie rule).
// ------------------------------------------------------------------------------------------------
#ifndef ALT_GT
#define ALT_GT 8
#endif
// BIDX8: the running arg-max (grid index, or the improvement counter of the compat quirk) of a lane's 4 NB outputs of one trait
// packed into the four bytes of ONE register (grids of at most 255 points; longer grids: the unpacked instantiation).  With 16
// index registers the kernel held 168 VGPRs + 10 spilled dwords at three waves per SIMD, and the spill traffic -- scratch lines
// pushed out of L2 by the stream of L / h2_panel stores -- showed in WRITE_SIZE (6.19 GB against 4.16 GB of output in round 3,
// 4.71 GB before the kernel ran three waves).
#ifndef ALT_MINW
#define ALT_MINW 3
#endif
template <int MB, int NB, bool BIDX8>
__global__ void __launch_bounds__(256, ALT_MINW) k_scan_alt(AltArgs aa, int ntile_i, int64_t nwg) {
  const ScanArgs& a = aa.s;
  __shared__ dpair s_lod[BLMM_LOD_TABLE_N];
  const double scale = a.lodc[0];         // -(n/2): the table of the other scan kernels, used once per test at the end
  {
    LodStage<256> lst;
    lod_stage_load<256>(lst, a.lodtab);
    lod_stage_store<256>(lst, s_lod, scale);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t bid = (uint32_t)xcd_swizzle(blockIdx.x, nwg);
  uint32_t tile_t, tile_i;
  // groups of ALT_GT trait tiles: a trait tile's panels are G x npad x 32 MB doubles (327 KB at BXD size with 16 grid points) -- 16 of
  // them (the group of the other scan kernels, whose tile has ONE panel set) are 5.2 MB and do not stay in an XCD's 4 MB L2: round 3
  // measured 11.5 GB of operand fetches per launch for 0.37 GB of panels
  tile_of32<ALT_GT>(bid, (uint32_t)nwg / (uint32_t)ntile_i, (uint32_t)ntile_i, tile_t, tile_i);
  const int64_t t0 = (int64_t)tile_t * (32 * MB) + (wave >> 1) * (16 * MB);
  const int64_t i0 = (int64_t)tile_i * (32 * NB) + (wave & 1) * (16 * NB);
  const int r = lane & 15, kk = lane >> 4;
  const LodPoly5 lp = lod_poly5_of(a.lodc);

  static_assert(!BIDX8 || NB == 4, "packed arg-max: four outputs per register");
  double best[MB][NB][4];
  int bidx[MB][BIDX8 ? 1 : NB][4];
  d4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      acc[mb][nb] = (d4){0, 0, 0, 0};
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) { best[mb][nb][reg] = 0.0; bidx[mb][BIDX8 ? 0 : nb][reg] = 0; }
    }

  // one flat loop over (grid point g, K step ks) with the same two-fragment-set prefetch as k_scan; the prefetch
  // runs across grid points, the running-max epilogue fires after the last K step of every g
  const double* PA = a.P + (int64_t)tile_t * (32 * MB);
  const double* PB = a.Xt + (int64_t)tile_i * (32 * NB);
  const uint32_t voffA = (uint32_t)(((int64_t)kk * a.ldp + (wave >> 1) * (16 * MB) + MB * r) * 8);
  const uint32_t voffB = (uint32_t)(((int64_t)kk * a.ldx + (wave & 1) * (16 * NB)) * 8) + mvoff<NB>(r);
  const int64_t sa = 4 * a.ldp, sb = 4 * a.ldx;
  const int KS = a.ks, G = aa.ngrid;
  int gl = 0, kl = 0;  // load cursor
  auto load_next = [&](double (&A)[MB], double (&B)[NB]) {
    bufload<MB>(A, make_srd(PA + (int64_t)gl * a.pstride + kl * sa), voffA);
    bufload_m<NB>(B, make_srd(PB + kl * sb), voffB);
    if (++kl == KS) { kl = 0; if (gl + 1 < G) ++gl; else kl = KS - 1; }   // clamp at the very end (harmless re-load)
  };
  auto mfma_set = [&](const double (&A)[MB], const double (&B)[NB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[mb], B[nb], acc[mb][nb], 0, 0, 0);
  };
  // operands of fold(g) -- the marker norms of grid point g and c[g, trait] of the lane's 4*MB traits -- are fetched
  // at the START of g's K loop: at fold time they used to cost one exposed global round trip per grid point
  double sc[NB], cv[MB][4];
  auto fold_fetch = [&](int g) {
    loadv_m<NB>(sc, a.isx + (int64_t)g * a.ld_isx + i0, r);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t trait = t0 + MB * (kk + 4 * reg) + mb;
        cv[mb][reg] = (trait < a.m) ? aa.Ctab[trait * (int64_t)G + g] : 1.0;
      }
  };
  auto fold = [&](int g) {  // v_g = (1 - r_g^2) c[g, j]; keep the running minimum = the running maximum of logL1 (tmax!, strict <)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const double cg = cv[mb][reg];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const double rr = acc[mb][nb][reg] * sc[nb];
          const double r2 = rr * rr;                 // (its own statement: -ffp-contract=on would fuse 1 - rr * rr into one fma)
          const double u = 1.0 - r2;                 // r2lod's operand, same operation order (src/bulkscan_helpers.jl:22-24)
          const double l1 = u * cg;
          const bool first = g == 0;
          const bool better = l1 < best[mb][nb][reg];
          if (first || better) best[mb][nb][reg] = l1;
          if constexpr (BIDX8) {
            const unsigned int w = (unsigned int)bidx[mb][0][reg], sh = 8u * (unsigned int)nb, msk = 0xffu << sh;
            const unsigned int nv = aa.counter_quirk ? ((w >> sh) & 0xffu) + 1u : (unsigned int)g;
            if (!first && better) bidx[mb][0][reg] = (int)((w & ~msk) | (nv << sh));
          } else {
            if (!first && better) bidx[mb][nb][reg] = aa.counter_quirk ? bidx[mb][nb][reg] + 1 : g;
          }
        }
      }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (d4){0, 0, 0, 0};
  };
  double a0[MB], b0[NB], a1[MB], b1[NB];
  load_next(a0, b0);
  for (int g = 0; g < G; ++g) {
    fold_fetch(g);
    __builtin_amdgcn_sched_barrier(0);
    int ks = 0;
    for (; ks + 2 <= KS; ks += 2) {
      load_next(a1, b1);                   // (g, ks+1)
      __builtin_amdgcn_sched_barrier(0);
      mfma_set(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load_next(a0, b0);                   // (g, ks+2), or the first step of g+1
      __builtin_amdgcn_sched_barrier(0);
      mfma_set(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ks < KS) {                         // odd KS: a0/b0 hold the last step of g; fetch (g+1, 0) and rotate the sets
      load_next(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_set(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a0[mb] = a1[mb];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) b0[nb] = b1[nb];
    }
    fold(g);
  }

  int nnan = 0;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t trait = t0 + MB * (kk + 4 * reg) + mb;
      if (trait >= a.m) continue;
      double lv[NB], hv[NB];
      bool ok = true;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) { lv[nb] = fast_lod5(best[mb][nb][reg], s_lod, lp); ok = ok && lod_fast_ok(best[mb][nb][reg]); }
      if (__builtin_expect(!ok, 0)) {   // LOD beyond ~1.2 n / 2; r^2 >= 1 at the winning grid point (+Inf / NaN); NaN
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (!lod_fast_ok(best[mb][nb][reg])) { int dummy = 0; lv[nb] = lod_out_of_range(best[mb][nb][reg], s_lod, lp, scale, false, &dummy); }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        hv[nb] = aa.grid_dev[BIDX8 ? (int)(((unsigned int)bidx[mb][0][reg] >> (8 * nb)) & 0xffu) : bidx[mb][BIDX8 ? 0 : nb][reg]];
        nnan += (lv[nb] != lv[nb]) && (i0 + mslot<NB>(r, nb) < a.p);
      }
      // 16-byte stores (8-byte aligned: ld = p may be odd), whole 128-byte lines per instruction (store_m)
      store_m<NB>(a.L + trait * a.ldL + i0, r, lv, a.p - i0);
      store_m<NB>(aa.H2 + trait * aa.ldH + i0, r, hv, a.p - i0);
    }
  if (nnan) atomicAdd((unsigned long long*)&a.stat[ST_NAN_LOD], (unsigned long long)nnan);
}

// c[g, j] = exp(-(2/n) (Ell[g, j] - max_g Ell[g, j])) for the fold of k_scan_alt (see there); one thread per trait
__global__ void __launch_bounds__(256) k_alt_ctab(const double* __restrict__ EllTab, int G, int64_t m, double two_over_n, double* __restrict__ C) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  double l0 = EllTab[j * G];
  for (int g = 1; g < G; ++g) l0 = fmax(l0, EllTab[j * G + g]);
  for (int g = 0; g < G; ++g) C[j * G + g] = exp(-two_over_n * (EllTab[j * G + g] - l0));
}
int launch_alt_ctab(blmm_ctx* ctx, const double* EllTab, int ngrid, int64_t m, int n, double* C) {
  if (m <= 0) return BLMM_OK;
  hipLaunchKernelGGL(k_alt_ctab, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, EllTab, ngrid, m, 2.0 / (double)n, C);
  KCHECK();
  return BLMM_OK;
}

#ifndef ALT_MB
#define ALT_MB 1
#endif
int launch_scan_alt(blmm_ctx* ctx, const AltArgs& aa) {
  // MB = 1 (16 traits x 64 markers per wave): accumulators + running max + arg-max stay in registers at 2+ waves/SIMD
  constexpr int MB = ALT_MB, NB = 4;
  const ScanArgs& a = aa.s;
  const int64_t ntile_t = (a.m + 32 * MB - 1) / (32 * MB);
  const int64_t ntile_i = (a.p + TILE_I - 1) / TILE_I;
  const int64_t nwg = ntile_t * ntile_i;
  if (nwg <= 0) return BLMM_OK;
  if (nwg > 0x7fffffffLL) return fail(ctx, BLMM_ERR_INVALID, "problem too large for one launch");
  if (aa.ngrid <= 255)
    hipLaunchKernelGGL((k_scan_alt<MB, NB, true>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, aa, (int)ntile_i, nwg);
  else
    hipLaunchKernelGGL((k_scan_alt<MB, NB, false>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, aa, (int)ntile_i, nwg);
  KCHECK();
  return BLMM_OK;
}

}  // namespace blmm
