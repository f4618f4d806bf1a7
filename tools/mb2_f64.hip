// Microbenchmark 2: fp64 MFMA sustained rate vs waves/SIMD and operand data, with the in-kernel clock
// (s_memtime / s_memrealtime).  Build: hipcc --offload-arch=gfx950 -O3 -o mb2_f64 mb2_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// MODE 0: mfma only; 1: mfma + NV v_fma_f64 per mfma (same wave); 2: mfma + NV v_fma_f32 per mfma
template<int NACC, int MODE, int NV>
__global__ void __launch_bounds__(256) k_loop(double* out, unsigned long long* clk, int iters, const double* src) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0,0,0,0};
  double a = src[threadIdx.x], b = src[256 + threadIdx.x];
  double x[NV > 0 ? NV : 1]; float xf[NV > 0 ? NV : 1];
  for (int i = 0; i < NV; i++) { x[i] = a + i; xf[i] = (float)(a + i); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (MODE == 1) {
#pragma unroll
        for (int v = 0; v < NV; v++) x[v] = fma(x[v], 0.999, 1e-3);
      } else if (MODE == 2) {
#pragma unroll
        for (int v = 0; v < NV; v++) xf[v] = fmaf(xf[v], 0.999f, 1e-3f);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; i++) s += x[i] + xf[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0; clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; }
}

template<int NACC, int MODE, int NV>
void run(const char* name, int cus, int wps, int iters, double* out, unsigned long long* clk, const double* src, double valu_flops_per) {
  int grid = cus * wps;  // 256-thread blocks = 1 wave per SIMD each
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_loop<NACC, MODE, NV><<<grid, 256>>>(out, clk, iters, src); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0)); k_loop<NACC, MODE, NV><<<grid, 256>>>(out, clk, iters, src); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  std::vector<unsigned long long> h(2 * grid * 4);
  CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, mhz;
  for (int w = 0; w < grid * 4; w++) { cyc.push_back((double)h[2 * w]); mhz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0); }
  std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
  double nm = (double)grid * 4 * iters * NACC;
  double tf = nm * 2048 / best / 1e9;
  double vtf = nm * NV * valu_flops_per / best / 1e9;
  printf("%-34s waves/SIMD %d: %.3f ms  mfma %.1f TF  valu %.1f TF  cycles/mfma/wave(median) %.1f  clock(median) %.0f MHz\n",
         name, wps, best, tf, vtf, cyc[cyc.size() / 2] / (iters * NACC), mhz[mhz.size() / 2]);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double* out; CK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
  unsigned long long* clk; CK(hipMalloc(&clk, 16 * cus * 8 * 4));
  double *srcz, *srcr; CK(hipMalloc(&srcz, 4096)); CK(hipMalloc(&srcr, 4096));
  std::vector<double> hz(512, 0.0), hr(512);
  srand(1); for (auto& v : hr) v = (rand() / (double)RAND_MAX - 0.5) * 2;
  CK(hipMemcpy(srcz, hz.data(), 4096, hipMemcpyHostToDevice)); CK(hipMemcpy(srcr, hr.data(), 4096, hipMemcpyHostToDevice));
  int iters = 4000;
  for (int wps : {1, 2, 4}) {
    run<4, 0, 0>("mfma only, zeros", cus, wps, iters, out, clk, srcz, 0);
    run<4, 0, 0>("mfma only, random", cus, wps, iters, out, clk, srcr, 0);
  }
  for (int wps : {1, 2}) {
    run<4, 1, 2>("mfma + 2 v_fma_f64 each, random", cus, wps, iters, out, clk, srcr, 128);
    run<4, 1, 4>("mfma + 4 v_fma_f64 each, random", cus, wps, iters, out, clk, srcr, 128);
    run<4, 1, 8>("mfma + 8 v_fma_f64 each, random", cus, wps, iters, out, clk, srcr, 128);
    run<4, 2, 8>("mfma + 8 v_fma_f32 each, random", cus, wps, iters, out, clk, srcr, 128);
    run<4, 2, 16>("mfma + 16 v_fma_f32 each, random", cus, wps, iters, out, clk, srcr, 128);
  }
  return 0;
}
