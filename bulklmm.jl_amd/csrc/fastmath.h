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
#include "pval_table.h"

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

// ---- LOD through the DIRECTLY indexed table (log_table.h: blmm_lod_table_host; tools/gen_log_table.py) ----------------
// The argument of the LOD map is u = 1 - r^2 in (0, 1], almost always in [2^-4, 1] (u < 2^-4 is a LOD beyond ~1.2 n / 2): over
// those four octaves the table is indexed by the high word of u alone -- no exponent extraction, no mantissa re-assembly, no
// int -> double conversion, and |r| <= 2^-10 leaves a degree-5 polynomial: 6 fp64 and 4 integer instructions per output where
// fast_lod has 11 + 7 (every fp64 VALU instruction of an epilogue is paid in matrix-pipe time, see the file header).
// Relative error <= ~2e-16 on [2^-4, 1] including u -> 1 (the last entry has c = 1: r = u - 1 exactly); lod_fast_ok(u) tells
// whether u is in the table's range, lod_slow() serves the rest (u < 2^-4, u <= 0, NaN) with the reference's own operations.
struct LodPoly5 { double c1, c2, c3, c4, c5; };
// host side: {scale, c1..c5} for ScanArgs::lodc
inline void lod_poly5_host(double scale, double (&out)[6]) {
  const double s = scale * BLMM_INV_LN10;
  out[0] = scale; out[1] = s; out[2] = -s * 0.5; out[3] = s * (1.0 / 3.0); out[4] = -s * 0.25; out[5] = s * (1.0 / 5.0);
}
__device__ __forceinline__ LodPoly5 lod_poly5_of(const double (&c)[6]) { return LodPoly5{c[1], c[2], c[3], c[4], c[5]}; }
__device__ __forceinline__ LodPoly5 make_lod_poly5(double scale) {
  const double s = scale * BLMM_INV_LN10;
  LodPoly5 p;
  p.c1 = s; p.c2 = -s * 0.5; p.c3 = s * (1.0 / 3.0); p.c4 = -s * 0.25; p.c5 = s * (1.0 / 5.0);
  return p;
}
// Staging of the table into LDS (NT threads per workgroup), split in two so that the loads can be issued at the top of a kernel
// and their results written out only after other loads are in flight: lds[i] = {invc_i, scale * log10(c_i)}.
template <int NT>
struct LodStage { dpair v[(BLMM_LOD_TABLE_N + NT - 1) / NT]; };
template <int NT>
__device__ __forceinline__ void lod_stage_load(LodStage<NT>& st, const double* __restrict__ gtab) {
  constexpr int PER = (BLMM_LOD_TABLE_N + NT - 1) / NT;
  const dpair* g = reinterpret_cast<const dpair*>(gtab);
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = (int)threadIdx.x + NT * u;
    st.v[u] = g[i < BLMM_LOD_TABLE_N ? i : BLMM_LOD_TABLE_N - 1];
  }
}
template <int NT>
__device__ __forceinline__ void lod_stage_store(const LodStage<NT>& st, dpair* lds, double scale) {
  constexpr int PER = (BLMM_LOD_TABLE_N + NT - 1) / NT;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = (int)threadIdx.x + NT * u;
    if (i < BLMM_LOD_TABLE_N) lds[i] = (dpair){st.v[u][0], scale * st.v[u][1]};
  }
}
__device__ __forceinline__ bool lod_fast_ok(double u) {
  return (uint32_t)__double2hiint(u) - BLMM_LOD_HI0 <= 0x3ff00000u - BLMM_LOD_HI0;   // 2^-4 <= u <= 1 (false for NaN, u <= 0)
}
__device__ __forceinline__ double fast_lod5(double u, const dpair* __restrict__ lds, const LodPoly5& P) {
  uint32_t off = (((uint32_t)__double2hiint(u) + (0x400u - BLMM_LOD_HI0)) >> 7) & 0xfff0u;   // 16 * ((hi - HI0 + 2^10) >> 11)
  off = off < 16u * (BLMM_LOD_TABLE_N - 1) ? off : 16u * (BLMM_LOD_TABLE_N - 1);             // out-of-range u: any entry (lod_slow replaces the value)
  const dpair e = *reinterpret_cast<const dpair*>(reinterpret_cast<const char*>(lds) + off);
  const double r = fma(u, e[0], -1.0);
  double p = P.c5;
  p = fma(p, r, P.c4);
  p = fma(p, r, P.c3);
  p = fma(p, r, P.c2);
  p = fma(p, r, P.c1);
  return fma(r, p, e[1]);
}
// r2lod outside the table's range (src/bulkscan_helpers.jl:22-24: scale * log10(u)).  0 < u < 2^-4 is brought into the table by
// an exact power of 16: scale log10(u) = fast_lod5(u 16^sh) - sh (4 scale log10 2), u 16^sh in [2^-4, 1) -- no libm call (its
// log10 is ~200 instructions and ~40 registers at every call site of kernels that sit at the register limit).  u = 0 -> +Inf;
// u < 0 (r^2 > 1: DomainError in Julia) or NaN -> NaN, counted in *nnan when `counted`.
__device__ __forceinline__ double lod_out_of_range(double u, const dpair* __restrict__ lds, const LodPoly5& P, double scale,
                                                   bool counted, int* nnan) {
  if (u > 0.0) {
    const int e = __builtin_amdgcn_frexp_exp(u);          // u = m 2^e, m in [0.5, 1); here e <= -4
    const int sh = (-e) >> 2;                             // e + 4 sh in (-4, 0]
    const double us = __builtin_amdgcn_ldexp(u, 4 * sh);   // exact, subnormal u included
    return fma(-(double)sh, scale * (4.0 * BLMM_LOG10_2), fast_lod5(us, lds, P));
  }
  if (u == 0.0) return INFINITY;
  *nnan += counted ? 1 : 0;
  return NAN;
}

// ---- -log10 p of a LOD score, one degree of freedom, inside a scan epilogue (`output_pvals`, src/bulkscan.jl:154-157;
// lod2log10p, src/util.jl:199-206 with chisq_df = 1) --------------------------------------------------------------------------
//   -log10 erfc(x) = LOD + x w(x),  x = sqrt(LOD ln 10),  w(x) = -log10(erfcx(x)) / x
// w: bucketed degree-7 polynomials (pval_table.h, tools/gen_pval_table.py: 4.3e-15 relative on the whole range); both terms are
// non-negative.  LOD <= 0 -> 0 (the reference's logccdf of a non-positive statistic), NaN -> NaN, +Inf -> +Inf; LOD < 1e-290
// (p = 1 - 1e-145) -> 0.  ~20 fp64 operations + four 16-byte LDS reads per value, against ~150 of the erfc / erfcx / log route
// of kernels_post.hip (which stays the general-df path).
template <int NT>
struct PvStage { dpair v[(BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2) + NT - 1) / NT]; };
template <int NT>
__device__ __forceinline__ void pv_stage_load(PvStage<NT>& st, const double* __restrict__ gtab) {
  constexpr int PER = (BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2) + NT - 1) / NT;
  const dpair* g = reinterpret_cast<const dpair*>(gtab);
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = (int)threadIdx.x + NT * u;
    st.v[u] = g[i < BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2) ? i : BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2) - 1];
  }
}
template <int NT>
__device__ __forceinline__ void pv_stage_store(const PvStage<NT>& st, dpair* lds) {
  constexpr int PER = (BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2) + NT - 1) / NT;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = (int)threadIdx.x + NT * u;
    if (i < BLMM_PV_TABLE_N * (BLMM_PV_STRIDE / 2)) lds[i] = st.v[u];
  }
}
__device__ __forceinline__ double fast_log10p1(double lod, const dpair* __restrict__ pv) {
  const double t = lod * 2.302585092994046;
  if (!(t >= 1e-290)) return (t == t) ? 0.0 : t;         // <= 0, underflowing, NaN
  // sqrt: v_rsq_f64 seed, two Newton steps on x = t y
  const double y = __builtin_amdgcn_rsq(t);
  const double h = 0.5 * y;
  double x = t * y;
  x = fma(fma(-x, x, t), h, x);
  x = fma(fma(-x, x, t), h, x);
  if (!(x < 16384.0)) return lod;                         // beyond the table (LOD >= 1.2e8, +Inf): x w(x) < 1e-8 LOD
  const uint32_t hi = (uint32_t)__double2hiint(x);
  const bool first = hi < BLMM_PV_HI0;                    // x < 2^-12: bucket 0, polynomial in x itself
  const uint32_t b = first ? 0u : 1u + ((hi - BLMM_PV_HI0) >> BLMM_PV_SHIFT);
  const double c = first ? 0.0 : __hiloint2double((int)((hi & ~((1u << BLMM_PV_SHIFT) - 1u)) | (1u << (BLMM_PV_SHIFT - 1))), 0);
  const double s = x - c;
  const dpair* e = pv + (BLMM_PV_STRIDE / 2) * b;
  const dpair c01 = e[0], c23 = e[1], c45 = e[2], c67 = e[3];
  double w = c67[1];
  w = fma(w, s, c67[0]);
  w = fma(w, s, c45[1]);
  w = fma(w, s, c45[0]);
  w = fma(w, s, c23[1]);
  w = fma(w, s, c23[0]);
  w = fma(w, s, c01[1]);
  w = fma(w, s, c01[0]);
  return fma(x, w, lod);
}

// 1/x to ~20 ulp (2.2e-15 relative, tools/mb4_rcp.hip): v_rcp_f64 seed (4.6e-8) + ONE Newton step.  Used where the quotient
// feeds 1 - r^2 of a LOD: its error there is far below the rounding of the subtraction.
__device__ __forceinline__ double fast_rcp1(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, y, 1.0);
  return fma(y, e, y);
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
