// Microbenchmark 3: v_mfma_f64_16x16x4_f64 issue rate with the accumulator (SrcC/vDst) in VGPRs vs AGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template<int NACC, bool AG>
__global__ void __launch_bounds__(256) k_loop(double* out, unsigned long long* clk, int iters, const double* src) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0,0,0,0};
  double a = src[threadIdx.x], b = src[256 + threadIdx.x];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      if (AG) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      else    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0; clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; }
}

template<int NACC, bool AG>
void run(const char* name, int cus, int wps, int iters, double* out, unsigned long long* clk, const double* src) {
  int grid = cus * wps;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_loop<NACC, AG><<<grid, 256>>>(out, clk, iters, src); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0)); k_loop<NACC, AG><<<grid, 256>>>(out, clk, iters, src); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  std::vector<unsigned long long> h(2 * grid * 4);
  CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, mhz;
  for (int w = 0; w < grid * 4; w++) { cyc.push_back((double)h[2 * w]); mhz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0); }
  std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
  double nm = (double)grid * 4 * iters * NACC;
  // per SIMD: from the THROUGHPUT (1024 SIMDs at the measured clock), not from the launch's waves-per-SIMD figure -- with 4 waves
  // per SIMD requested the hardware may run two workgroups after one another (round 2 printed 32.0 there, against its own TF column)
  const double tf = nm * 2048 / best / 1e9;
  printf("%-26s nacc %d waves/SIMD %d: %.3f ms  %.1f TF  cycles/mfma/wave %.1f  (per SIMD, from the throughput: %.1f)  clock %.0f MHz\n", name, NACC, wps, best,
         tf, cyc[cyc.size() / 2] / (iters * NACC), 2048.0 * 1024.0 * mhz[mhz.size() / 2] * 1e6 / (tf * 1e12), mhz[mhz.size() / 2]);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double* out; CK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
  unsigned long long* clk; CK(hipMalloc(&clk, 16 * cus * 8 * 4));
  double* srcr; CK(hipMalloc(&srcr, 4096));
  std::vector<double> hr(512);
  srand(1); for (auto& v : hr) v = (rand() / (double)RAND_MAX - 0.5) * 2;
  CK(hipMemcpy(srcr, hr.data(), 4096, hipMemcpyHostToDevice));
  int iters = 3000;
  for (int wps : {1, 2, 4}) {
    run<4, false>("acc in VGPR", cus, wps, iters, out, clk, srcr);
    run<4, true>("acc in AGPR", cus, wps, iters, out, clk, srcr);
    run<12, false>("acc in VGPR", cus, wps, iters, out, clk, srcr);
    run<12, true>("acc in AGPR", cus, wps, iters, out, clk, srcr);
  }
  return 0;
}
