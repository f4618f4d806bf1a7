// fastmath.h -- table-driven fp64 logarithms and Newton reciprocal for the gfx950 kernels.
// On MI355X the fp64 MFMA and fp64 VALU instructions execute exclusively of each other on a SIMD
// (tools/mb2_f64.hip), so every VALU instruction of an epilogue is paid in matrix-pipe time: the OCML
// log10 (~50 instruction slots) and IEEE division (~16) are replaced by a 128-entry table + degree-8
// polynomial (~20 slots, <= 1 ulp-ish, full relative accuracy as x -> 1) and rcp + 2 Newton steps.
// Table: log_table.h (tools/gen_log_table.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "log_table.h"

namespace blmm {

typedef double dpair __attribute__((ext_vector_type(2)));

// stage {invc, log10(c)} (LOG10 = true) or {invc, ln(c)} pairs into LDS: 128 x 16 bytes.
// gtab: device copy of blmm_log_table_host (blmm_ctx::logtab, uploaded by blmm_create)
template <bool LOG10>
__device__ __forceinline__ void stage_log_table(dpair* lds, const double* __restrict__ gtab) {
  for (int i = threadIdx.x; i < BLMM_LOG_TABLE_N; i += blockDim.x)
    lds[i] = (dpair){gtab[3 * i], gtab[3 * i + (LOG10 ? 2 : 1)]};
}

// log10(x) (LOG10) or ln(x) for a positive, finite, normal x.  lds: table staged by stage_log_table.
template <bool LOG10>
__device__ __forceinline__ double fast_log(double x, const dpair* __restrict__ lds) {
  const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
  const uint32_t tmp = hi - 0x3fe60000u;
  const int i = (int)((tmp >> 13) & 127u);
  const int k = (int)tmp >> 20;
  const double z = __hiloint2double((int)(hi - (tmp & 0xfff00000u)), (int)lo);
  const dpair e = lds[i];
  const double r = fma(z, e[0], -1.0);
  constexpr double S = LOG10 ? BLMM_INV_LN10 : 1.0;
  double p = -S / 8.0;
  p = fma(p, r, S / 7.0);
  p = fma(p, r, -S / 6.0);
  p = fma(p, r, S / 5.0);
  p = fma(p, r, -S / 4.0);
  p = fma(p, r, S / 3.0);
  p = fma(p, r, -S / 2.0);
  p = fma(p, r, S);
  const double t = fma((double)k, LOG10 ? BLMM_LOG10_2 : BLMM_LN2, e[1]);
  return fma(r, p, t);
}

// ---- LOD form: scale * log10(x) with `scale` folded into the table and the polynomial ---------------------
// lds[i] = {invc, scale * log10(c)} (stage_lod_table); one multiply and one polynomial term fewer per output than
// scale * fast_log<true>(x): relative error <= ~3e-16 on (0, 1].
__device__ __forceinline__ void stage_lod_table(dpair* lds, const double* __restrict__ gtab, double scale) {
  for (int i = threadIdx.x; i < BLMM_LOG_TABLE_N; i += blockDim.x)
    lds[i] = (dpair){gtab[3 * i], scale * gtab[3 * i + 2]};
}
struct LodPoly { double k2, c1, c2, c3, c4, c5, c6, c7; };
__device__ __forceinline__ LodPoly make_lod_poly(double scale) {
  const double s = scale * BLMM_INV_LN10;
  LodPoly p;
  p.k2 = scale * BLMM_LOG10_2;
  // reciprocals as constants: an IEEE division by 3, 5, 6, 7 costs ~25 instructions each, once per tile, for a last-bit
  // difference in coefficients whose terms are <= 2^-24 of the result
  p.c1 = s; p.c2 = -s * 0.5; p.c3 = s * (1.0 / 3.0); p.c4 = -s * 0.25; p.c5 = s * (1.0 / 5.0); p.c6 = -s * (1.0 / 6.0);
  p.c7 = s * (1.0 / 7.0);
  return p;
}
__device__ __forceinline__ double fast_lod(double x, const dpair* __restrict__ lds, const LodPoly& P) {
  const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
  const uint32_t tmp = hi - 0x3fe60000u;
  const int i = (int)((tmp >> 13) & 127u);
  const int k = (int)tmp >> 20;
  const double z = __hiloint2double((int)(hi - (tmp & 0xfff00000u)), (int)lo);
  const dpair e = lds[i];
  const double r = fma(z, e[0], -1.0);
  double p = P.c7;
  p = fma(p, r, P.c6);
  p = fma(p, r, P.c5);
  p = fma(p, r, P.c4);
  p = fma(p, r, P.c3);
  p = fma(p, r, P.c2);
  p = fma(p, r, P.c1);
  return fma(r, p, fma((double)k, P.k2, e[1]));
}

// 1/x to ~1 ulp: v_rcp_f64 seed + two Newton steps (x finite, non-zero, normal)
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

// Sum over the aligned groups of LPT lanes (LPT a power of two <= 64); every lane of a group gets the group's total.
// Steps inside a 16-lane row go through DPP (quad_perm, row_half_mirror, row_mirror: one v_mov_dpp pair per step, ~8
// cycles) -- the __shfl_xor butterfly they replace is a pair of ds_bpermute per step, ~100+ cycles of LDS round trip each,
// and was the bulk of a latency-bound evaluation's time.  Same pairing as the xor butterfly (addition commutes), so the
// sums are bit-identical to it.
template <int CTRL>
__device__ __forceinline__ double blmm_dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int LPT>
__device__ __forceinline__ double group_sum(double x) {
  if constexpr (LPT >= 2) x += blmm_dpp_mov<0xB1>(x);     // quad_perm [1,0,3,2]   = lane ^ 1
  if constexpr (LPT >= 4) x += blmm_dpp_mov<0x4E>(x);     // quad_perm [2,3,0,1]   = lane ^ 2
  if constexpr (LPT >= 8) x += blmm_dpp_mov<0x141>(x);    // row_half_mirror: lane i <-> 7 - i (quads hold equal sums)
  if constexpr (LPT >= 16) x += blmm_dpp_mov<0x140>(x);   // row_mirror: lane i <-> 15 - i
  if constexpr (LPT >= 32) x += __shfl_xor(x, 16, 64);
  if constexpr (LPT >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

// Adds the number of lanes of the wave with `flag` set to *cnt: one atomic per wave (every lane of the wave must call it).
__device__ __forceinline__ void wave_count(int64_t* cnt, bool flag) {
  const unsigned long long b = __ballot(flag);
  if (b != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)b) - 1)
    atomicAdd((unsigned long long*)cnt, (unsigned long long)__popcll(b));
}

// h2 estimates on a boundary of [0, 1] (blmm_status.n_h2_boundary)
__device__ __forceinline__ bool h2_on_boundary(double h2) { return h2 <= 1e-6 || h2 >= 1.0 - 1e-6; }

}  // namespace blmm
