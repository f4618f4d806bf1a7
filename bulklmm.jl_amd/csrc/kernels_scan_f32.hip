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

// ------------------------------------------------------------------------------------------------
// fp32 ROTATION for the fp32 permutation path (round 4; BASELINE.json configs[4] asks the whole permutation config in fp32 -- round
// 3 rotated G on the fp64 matrix cores, 3.7 ms of a 14.6 ms shard step, and converted the result):
//     XF = fragment-major fp32 of  R G      (transform_rotation's Ut * X, src/transform_helpers.jl:34, centring folded into R)
// G arrives as the caller's fp64 column-major n x p matrix and is converted in registers; R as a row-major fp32 copy RF[k][r]
// (k_cvt_r32).  v_mfma_f32_32x32x2_f32: A (32 rows) = rotated index k, B (32 cols) = markers; the two k slots of an MFMA (lane >> 5)
// take the contraction indices r0 + 8 h + s, s = 0 .. 7 -- the order of a sum is free, and with this one a lane reads 64
// CONTIGUOUS bytes of its marker column per 16 contraction steps (and 32 contiguous bytes of its row of RF).  Workgroup = 4 waves
// (2 x 2), 128 rotated rows x 128 markers, four 32 x 32 accumulators per wave; the next chunk's raw operands are requested before
// the 32 MFMAs of the current one.  The output is written straight in the operand layout of k_scan_f32 (F[kb][h'][col][j], k = 8 kb +
// 2 j + h'): a lane's rows 8 g + 4 h + {t, t + 2} of a marker are two consecutive floats, and the two halves of the wave fill the
// 16-byte vector -- 512 dense bytes per store instruction, no conversion pass.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t xcd_swizzle32(int64_t bid, int64_t nwg);
typedef double d2v __attribute__((ext_vector_type(2), aligned(8)));
template <int CTRL>
__device__ __forceinline__ double blmm_dpp_mov8(double x) {      // quad_perm / row_half_mirror exchanges inside eight lanes
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
typedef float f2u __attribute__((ext_vector_type(2), aligned(8)));

// RF[k * ldrr + r] = (float) R[k, r] = (float) Rp[r * ldr + k], zero beyond n (rows up to kpad, columns up to ldrr)
__global__ void __launch_bounds__(256) k_cvt_r32(const double* __restrict__ Rp, int ldr, int n, float* __restrict__ RF, int kpad, int ldrr) {
  __shared__ float tile[32][33];                            // 32 x 32 transpose: reads run along k, writes along r
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, k = k0 + tx;
    tile[rr][tx] = (r < n && k < n) ? (float)Rp[(size_t)r * ldr + k] : 0.0f;
  }
  __syncthreads();
  for (int kk = ty; kk < 32; kk += 8) {
    const int k = k0 + kk, r = r0 + tx;
    if (k < kpad && r < ldrr) RF[(size_t)k * ldrr + r] = tile[tx][kk];
  }
}

struct RotF32Args {
  const double* G; int n; int64_t p;        // column-major n x p
  const float* RF; int ldrr;                // row-major kpad x ldrr
  float* XF; int64_t ldxf; int kblocks;     // fragment-major output, kblocks = npad / 8
  const double* v; double* num;             // optional: num[i] = g_i' v in fp64 (v: ldrr doubles, zero beyond n), by the row-tile-0 workgroups
};

// Operands through LDS (the first form read its fragments straight from global memory, a lane per marker column: 3072 64-byte
// requests per workgroup and 16 contraction steps -- the L1 / address path, not the matrix pipe, set its 3.2 ms at n = 1000,
// p = 1e5): per chunk of 16 contraction steps the workgroup fetches the 128 x 16 tile of G (16 KB of fp64, four lanes per 64-byte
// segment: 256 fully used requests) and the 128 x 16 tile of RF (8 KB, 128 requests), converts G, and lays both out in LDS as
// [h][row or marker][8 floats] -- exactly the eight values a lane feeds to the chunk's eight MFMA steps (two ds_read_b128).
// Double-buffered: the next chunk's global loads are in flight during the current chunk's 32 MFMAs per wave, one barrier per chunk.
__global__ void __launch_bounds__(256, 2) k_rotate_f32(RotF32Args a, int ntile_i) {
  // [buffer][h][steps 0-3 | 4-7][marker or rotated row][4]: consecutive lanes read consecutive 16-byte vectors (no bank conflicts)
  __shared__ __attribute__((aligned(16))) float s_b[2][2][2][128][4];
  __shared__ __attribute__((aligned(16))) float s_a[2][2][2][128][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wk = wave >> 1, wi = wave & 1, c = lane & 31, h = lane >> 5;
  // marker tile slow, row tile fast inside an XCD's range: the row tiles of a marker tile reuse its slice of G out of L2
  const int64_t nwg = gridDim.x;
  const int64_t bid = xcd_swizzle32(blockIdx.x, nwg);
  const int nkt = (int)(nwg / ntile_i);
  const int tile_k = (int)(bid % nkt);
  const int64_t tile_i = bid / nkt;
  const int k0 = tile_k * 128;
  const int64_t i0 = tile_i * 128;
  f16v acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f16v){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // staging map: G -- pass q (4): segment sg = 64 q + (t >> 2) = (marker = sg >> 1, half = sg & 1), piece = t & 3: two doubles
  //              RF -- pass q (2): row = 64 q + (t >> 2), piece = t & 3: four floats of the row's 16
  const int piece = t & 3;
  const double* gsrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int sg = 64 * q + (t >> 2);
    int64_t col = i0 + (sg >> 1);
    if (col >= a.p) col = a.p - 1;                       // pad columns: a finite duplicate (k_scan_f32 never stores them)
    gsrc[q] = a.G + col * (int64_t)a.n + 8 * (sg & 1) + 2 * piece;
  }
  const float* rsrc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) rsrc[q] = a.RF + (size_t)(k0 + 64 * q + (t >> 2)) * a.ldrr + 4 * piece;
  const int nch = (a.n + 15) / 16;
  d2v graw[4];
  f4 rraw[2];
  // the original trait's fp64 numerator rides along: this thread's two individuals of each of its four markers times v
  const bool do_num = a.v != nullptr && tile_k == 0;
  double pnum[4] = {0.0, 0.0, 0.0, 0.0};
  const double* vsrc = a.v + 8 * ((t >> 2) & 1) + 2 * piece;
  auto num_chunk = [&](int ch) {
    const d2v vv = *reinterpret_cast<const d2v*>(vsrc + 16 * ch);
#pragma unroll
    for (int q = 0; q < 4; ++q) pnum[q] = fma(graw[q][1], vv[1], fma(graw[q][0], vv[0], pnum[q]));
  };
  auto load_chunk = [&](int ch) {
    const int R0 = 16 * ch;
    if (R0 + 16 <= a.n) {
#pragma unroll
      for (int q = 0; q < 4; ++q) graw[q] = *reinterpret_cast<const d2v*>(gsrc[q] + R0);
    } else {                                             // the last, partial chunk: RF is zero there; G must stay inside its column
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sg = 64 * q + (t >> 2);
        const int r0 = R0 + 8 * (sg & 1) + 2 * piece, ra = r0 < a.n ? r0 : a.n - 1, rb = r0 + 1 < a.n ? r0 + 1 : a.n - 1;
        const double* base = gsrc[q] - (8 * (sg & 1) + 2 * piece);
        graw[q] = (d2v){base[ra], base[rb]};
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) rraw[q] = *reinterpret_cast<const f4*>(rsrc[q] + R0);
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int sg = 64 * q + (t >> 2);
      *reinterpret_cast<f2u*>(&s_b[buf][sg & 1][piece >> 1][sg >> 1][2 * (piece & 1)]) = (f2u){(float)graw[q][0], (float)graw[q][1]};
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<f4*>(&s_a[buf][piece >> 1][piece & 1][64 * q + (t >> 2)][0]) = rraw[q];
  };
  load_chunk(0);
  if (do_num) num_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) load_chunk(ch + 1);
    f4 bf[2][2], af[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      bf[nt][0] = *reinterpret_cast<const f4*>(&s_b[buf][h][0][64 * wi + 32 * nt + c][0]);
      bf[nt][1] = *reinterpret_cast<const f4*>(&s_b[buf][h][1][64 * wi + 32 * nt + c][0]);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      af[mt][0] = *reinterpret_cast<const f4*>(&s_a[buf][h][0][64 * wk + 32 * mt + c][0]);
      af[mt][1] = *reinterpret_cast<const f4*>(&s_a[buf][h][1][64 * wk + 32 * mt + c][0]);
    }
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][s8 >> 2][s8 & 3], bf[nt][s8 >> 2][s8 & 3], acc[mt][nt], 0, 0, 0);
    if (ch + 1 < nch) { if (do_num) num_chunk(ch + 1); store_chunk(buf ^ 1); }
    __syncthreads();
  }
  if (do_num) {                                          // the eight lanes t & 7 of a marker (two halves x four pieces), fixed order
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double x = pnum[q];
      x += blmm_dpp_mov8<0xB1>(x); x += blmm_dpp_mov8<0x4E>(x); x += blmm_dpp_mov8<0x141>(x);
      const int64_t col = i0 + ((64 * q + (t >> 2)) >> 1);
      if ((t & 7) == 0 && col < a.p) a.num[col] = x;
    }
  }
  // fragment-major stores: rows k = k0 + 64 wk + 32 mt + 8 g + 4 h + t of marker col; t = hp, hp + 2 are floats 2 h, 2 h + 1 of vector (kb, hp)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int kb = (k0 + 64 * wk + 32 * mt) / 8 + g;
      if (kb >= a.kblocks) continue;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int64_t col = i0 + 64 * wi + 32 * nt + c;
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          float* dst = a.XF + (((int64_t)(kb * 2 + hp) * a.ldxf + col) * 4 + 2 * h);
          *reinterpret_cast<f2u*>(dst) = (f2u){acc[mt][nt][4 * g + hp], acc[mt][nt][4 * g + hp + 2]};
        }
      }
    }
}

// The original trait's LOD vector of the fp32 permutation path, with an fp64 numerator that never needs the fp64 rotated markers:
//   num_i = x~_i' a0 = (R g_i)' a0 = g_i' v,  v = R' a0   (k_backproject: v[r] = sum_k Rp[r ldr + k] a0[k], zero from n to ldrr)
// The sum g_i' v rides along in k_rotate_f32, which has every element of G in registers as fp64 on its way to LDS.
__global__ void __launch_bounds__(256) k_backproject(const double* __restrict__ Rp, int ldr, int n, int ldrr, const double* __restrict__ a0,
                                                     int64_t lda, double* __restrict__ v) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per r: its row of Rp is contiguous
  if (r >= ldrr) return;
  double s = 0.0;
  if (r < n)
    for (int k = lane; k < n; k += 64) s = fma(Rp[(size_t)r * ldr + k], a0[(int64_t)k * lda], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) v[r] = s;
}

// XF (npad x ldxf, fragment-major fp32) = R G; scratch RF: kpad x ldrr floats, kpad = npad rounded up to 128, ldrr = n rounded up to 16
int launch_rotate_f32(blmm_ctx* ctx, const double* Rp, int ldr, int n, int npad, const double* dG, int64_t p, float* RF, float* XF, int64_t ldxf,
                      const double* a0, int64_t lda, double* v_work, double* num) {
  if (p <= 0) return BLMM_OK;
  if (npad % 8 != 0 || ldxf % 256 != 0) return fail(ctx, BLMM_ERR_INVALID, "rotate_f32: operands are not padded to the tile");
  const int kpad = (npad + 127) / 128 * 128, ldrr = (n + 15) / 16 * 16;
  hipLaunchKernelGGL(k_cvt_r32, dim3((unsigned)((ldrr + 31) / 32), (unsigned)((kpad + 31) / 32)), dim3(256), 0, ctx->stream, Rp, ldr, n, RF, kpad, ldrr);
  if (a0) hipLaunchKernelGGL(k_backproject, dim3((unsigned)((ldrr + 3) / 4)), dim3(256), 0, ctx->stream, Rp, ldr, n, ldrr, a0, lda, v_work);
  RotF32Args a;
  a.G = dG; a.n = n; a.p = p; a.RF = RF; a.ldrr = ldrr; a.XF = XF; a.ldxf = ldxf; a.kblocks = npad / 8;
  a.v = a0 ? v_work : nullptr; a.num = num;
  const int64_t ntile_i = ldxf / 128, nkt = kpad / 128;
  if (ntile_i * nkt > 0x7fffffffLL) return fail(ctx, BLMM_ERR_INVALID, "problem too large for one launch");
  hipLaunchKernelGGL(k_rotate_f32, dim3((unsigned)(ntile_i * nkt)), dim3(256), 0, ctx->stream, a, (int)ntile_i);
  KCHECK();
  return BLMM_OK;
}

// lod_i = -(n/2) log10(1 - (num_i isx_i)^2) of the original trait, num from k_rotate_f32's fp64 side sum (r2lod, src/bulkscan_helpers.jl:22-24)
__global__ void __launch_bounds__(256) k_lod_from_num(const double* __restrict__ num, const double* __restrict__ isx, int n, int64_t p,
                                                      double* __restrict__ lod, int64_t* stat) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p) return;
  const double rr = num[i] * isx[i];
  const double r2 = rr * rr;
  const double u = 1.0 - r2;
  if (!(u >= 0.0)) atomicAdd((unsigned long long*)&stat[ST_NAN_LOD], 1ull);
  lod[i] = -0.5 * (double)n * log10(u);
}
int launch_lod_from_num(blmm_ctx* ctx, const double* num, const double* isx, int n, int64_t p, double* lod, int64_t* stat) {
  if (p <= 0) return BLMM_OK;
  hipLaunchKernelGGL(k_lod_from_num, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, ctx->stream, num, isx, n, p, lod, stat);
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
