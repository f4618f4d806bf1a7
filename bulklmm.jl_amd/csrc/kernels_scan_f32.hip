// kernels_scan_f32.hip -- fp32 LOD kernel for the permutation test (BASELINE.json configs[4]: n = 1000, p = 100000,
// 10000 permutations as pseudo-traits, fp32).
//
//   L_perms[i, b] = -(n/2) * log10(1 - r_ib^2),   r_ib = <x~_i, pi_b(r0)~>      (scan_perms_lite, src/scan.jl:534-552)
//
// The permutation copies share ONE weight vector, so this is the "table" form of kernels_scan.hip with a single bin:
// r = (x_i' a_b) * isx[i].  Here the contraction runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: 64 cycles per
// 32x32x2 tile per SIMD = twice the f64 MFMA rate, exact fp32 fma chain), the operands are fp32 copies of the rotated
// markers / the permutation panel, and L_perms is written as fp32 (half the HBM bytes of the f64 path).  The null
// model (eigen-decomposition, rotation, Brent h2, panel construction, marker norms) stays in fp64.
//
// Operand layout ("fragment-major", written by k_cvt_f32): F[kb][h][col][j] = M[8 kb + 2 j + h][col], i.e. one 16-byte
// vector holds the values a lane feeds to FOUR consecutive K steps (the 32x32x2 MFMA takes k = lane >> 5 from lane l).
// MFMA roles as in the f64 kernels: A (32 rows) = permutations, B (32 cols) = markers; D puts one marker column on a
// lane (col = lane & 31) and 16 permutation rows in its registers.  marker <-> (block nb, col c) = i0 + NB c + nb, so
// a lane owns NB = 4 consecutive markers: one 16-byte store per (register, permutation), 32 lanes = 512 contiguous
// bytes of one L_perms column.  Workgroup = 4 waves (2 x 2) = 128 permutations x 256 markers, no LDS.
#include "blmm_internal.h"
#include <cmath>

namespace blmm {

#define KCHECK()                                                                                      \
  do {                                                                                                \
    hipError_t e__ = hipGetLastError();                                                               \
    if (e__ != hipSuccess) return fail(ctx, BLMM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e__)); \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// M: k-major fp64 (rows x ld_in, rows beyond `rows_valid` and columns beyond `cols_valid` read as zero)
// F: fragment-major fp32 [kb][h][ld_out][4], kb < kblocks
__global__ void __launch_bounds__(256) k_cvt_f32(const double* __restrict__ M, int64_t ld_in, int rows_valid,
                                                 int64_t cols_valid, float* __restrict__ F, int64_t ld_out, int kblocks) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kh = blockIdx.y;                 // kb * 2 + h
  if (col >= ld_out) return;
  const int kb = kh >> 1, h = kh & 1;
  f4 v = (f4){0.f, 0.f, 0.f, 0.f};
  if (col < cols_valid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 8 * kb + 2 * j + h;
      if (k < rows_valid) v[j] = (float)M[(int64_t)k * ld_in + col];
    }
  }
  *reinterpret_cast<f4*>(F + ((int64_t)kh * ld_out + col) * 4) = v;
}

int launch_cvt_f32(blmm_ctx* ctx, const double* M, int64_t ld_in, int rows_valid, int64_t cols_valid, float* F,
                   int64_t ld_out, int kblocks) {
  if (kblocks <= 0 || ld_out <= 0) return BLMM_OK;
  dim3 grid((unsigned)((ld_out + 255) / 256), (unsigned)(2 * kblocks));
  hipLaunchKernelGGL(k_cvt_f32, grid, dim3(256), 0, ctx->stream, M, ld_in, rows_valid, cols_valid, F, ld_out, kblocks);
  KCHECK();
  return BLMM_OK;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t srd_f32(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), /*stride*/ 0, /*bytes*/ 0xffffffffu, 0x00020000);
}
__device__ __forceinline__ f4 bufload_f4(__amdgpu_buffer_rsrc_t srd, uint32_t voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, 0, 0);
  return __builtin_bit_cast(f4, v);
}

__device__ __forceinline__ int64_t xcd_swizzle32(int64_t bid, int64_t nwg) {
  const int64_t q = nwg >> 3;
  return (bid < (q << 3)) ? (bid & 7) * q + (bid >> 3) : bid;
}

struct ScanF32Args {
  const float* XF; int64_t ldxf;     // markers, fragment-major, ldxf a multiple of 256
  const float* PF; int64_t ldpf;     // permutation panel, fragment-major, ldpf a multiple of 128
  int kblocks;                       // npad / 8
  int n;
  int64_t p, m;
  const double* isx;                 // 1 / |P sqrt(w) x_i| (fp64, from k_isx)
  float* L; int64_t ldL;
  int64_t* stat;
};

template <int MB, int NB>
__global__ void __launch_bounds__(256, 2) k_scan_f32(ScanF32Args a, int ntile_i, int64_t nwg) {
  static_assert(NB == 4, "a lane owns four consecutive markers (16-byte stores)");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bid = xcd_swizzle32(blockIdx.x, nwg);
  // groups of 8 permutation tiles share a marker tile walk (as tile_of in kernels_scan.hip)
  constexpr int GTF = 8;
  const int64_t ntile_t = nwg / ntile_i;
  const int64_t per_group = (int64_t)GTF * ntile_i;
  const int64_t g = bid / per_group, rem = bid - g * per_group;
  const int64_t t_first = g * GTF;
  const int gt = (int)((ntile_t - t_first < GTF) ? (ntile_t - t_first) : GTF);
  const int tile_i = (int)(rem / gt);
  const int64_t tile_t = t_first + (rem - (int64_t)tile_i * gt);

  const int wt = wave >> 1, wi = wave & 1;
  const int c = lane & 31, h = lane >> 5;
  const int64_t t0 = tile_t * (64 * MB) + wt * (32 * MB);       // first permutation of this wave
  const int64_t i0 = (int64_t)tile_i * (64 * NB) + wi * (32 * NB);

  f16v acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f16v){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // uniform tile bases in the descriptors, per-lane byte offsets in ONE 32-bit voffset per operand
  const float* PA = a.PF + tile_t * (64 * MB) * 4;
  const float* PB = a.XF + (int64_t)tile_i * (64 * NB) * 4;
  const uint32_t voffA = (uint32_t)((((int64_t)h * a.ldpf + wt * (32 * MB) + MB * c) * 4) * 4);
  const uint32_t voffB = (uint32_t)((((int64_t)h * a.ldxf + wi * (32 * NB) + NB * c) * 4) * 4);
  const int64_t sa = 2 * a.ldpf * 4, sb = 2 * a.ldxf * 4;      // floats per K block

  auto load_set = [&](f4 (&A)[MB], f4 (&B)[NB], int kb) {
    const __amdgpu_buffer_rsrc_t ra = srd_f32(PA + (int64_t)kb * sa), rb = srd_f32(PB + (int64_t)kb * sb);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) A[mb] = bufload_f4(ra, voffA + 16 * mb);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) B[nb] = bufload_f4(rb, voffB + 16 * nb);
  };
  auto mfma_set = [&](const f4 (&A)[MB], const f4 (&B)[NB]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[mb][j], B[nb][j], acc[mb][nb], 0, 0, 0);
  };
  f4 a0[MB], b0[NB], a1[MB], b1[NB];
  load_set(a0, b0, 0);
  int kb = 0;
  for (; kb + 2 <= a.kblocks; kb += 2) {
    load_set(a1, b1, kb + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_set(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    load_set(a0, b0, (kb + 2 < a.kblocks) ? kb + 2 : kb);   // unconditional (clamped): static vmcnt bookkeeping
    __builtin_amdgcn_sched_barrier(0);
    mfma_set(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kb < a.kblocks) mfma_set(a0, b0);

  // ---- epilogue: r = num * isx, LOD = -(n/2) log10(1 - r^2), 16-byte stores ---------------------------------------
  // log(1 - x) in fp32 without losing small x: u = fl(1 - x), e = (1 - u) - x is the rounding error of u (exact), and
  // log(1 - x) = log(u) + log1p(e/u) ~ log(u) + e/u.  v_log_f32 / v_rcp_f32 (~1 ulp) are ample for the fp32 contract.
  const int64_t ibase = i0 + NB * c;
  float sc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) sc[nb] = (ibase + nb < a.p) ? (float)a.isx[ibase + nb] : 0.f;
  const float scale = (float)(-0.5 * (double)a.n / 2.302585092994046);
  const bool full = ibase + NB <= a.p;
  int nnan = 0;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      const int64_t trait = t0 + MB * row + mb;
      f4 out;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float rr = acc[mb][nb][reg] * sc[nb];
        const float r2 = rr * rr;
        const float u = 1.0f - r2;
        const float e = (1.0f - u) - r2;
        float lod = scale * fmaf(e, __builtin_amdgcn_rcpf(u), __logf(u));
        if (__builtin_expect(!(u > 0.0f), 0)) {   // r^2 = 1 -> +Inf; r^2 > 1 -> DomainError in Julia, NaN here
          lod = (u == 0.0f) ? INFINITY : NAN;
          nnan += (u != 0.0f) && (ibase + nb < a.p) && (trait < a.m);
        }
        out[nb] = lod;
      }
      if (trait < a.m) {
        float* dst = a.L + trait * a.ldL + ibase;
        if (full) {
          __builtin_nontemporal_store(out, reinterpret_cast<f4u*>(dst));
        } else {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            if (ibase + nb < a.p) dst[nb] = out[nb];
        }
      }
    }
  }
  if (nnan) atomicAdd((unsigned long long*)&a.stat[ST_NAN_LOD], (unsigned long long)nnan);
}

// XF: fragment-major markers (ldxf multiple of 256), PF: fragment-major panel (ldpf multiple of 128)
int launch_scan_f32(blmm_ctx* ctx, const float* XF, int64_t ldxf, const float* PF, int64_t ldpf, int npad, int n,
                    int64_t p, int64_t m, const double* isx, float* L, int64_t ldL, int64_t* stat) {
  constexpr int MB = 2, NB = 4;
  if (p <= 0 || m <= 0) return BLMM_OK;
  if (npad % 8 != 0 || ldxf % (64 * NB) != 0 || ldpf % (64 * MB) != 0)
    return fail(ctx, BLMM_ERR_INVALID, "scan_f32: operands are not padded to the tile");
  // per-lane byte offsets must fit the 32-bit voffset of a buffer load
  if ((uint64_t)(2 * ldxf) * 16 > 0xffffffffull || (uint64_t)(2 * ldpf) * 16 > 0xffffffffull)
    return fail(ctx, BLMM_ERR_UNSUPPORTED, "scan_f32: problem too large for one launch");
  const int64_t ntile_t = (m + 64 * MB - 1) / (64 * MB);
  const int64_t ntile_i = (p + 64 * NB - 1) / (64 * NB);
  const int64_t nwg = ntile_t * ntile_i;
  if (nwg > 0x7fffffffLL) return fail(ctx, BLMM_ERR_INVALID, "problem too large for one launch");
  ScanF32Args a;
  a.XF = XF; a.ldxf = ldxf; a.PF = PF; a.ldpf = ldpf; a.kblocks = npad / 8; a.n = n; a.p = p; a.m = m; a.isx = isx;
  a.L = L; a.ldL = ldL; a.stat = stat;
  hipLaunchKernelGGL((k_scan_f32<MB, NB>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, a, (int)ntile_i, nwg);
  KCHECK();
  return BLMM_OK;
}

}  // namespace blmm
