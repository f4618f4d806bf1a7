// Microbenchmark: fp64 MFMA / fp64 VALU issue rates on gfx950 and the f64 MFMA fragment layout.
// Build: hipcc --offload-arch=gfx950 -O3 -o mb_f64 mb_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template<int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0,0,0,0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double a0, double b0) {
  double x[16];
  for (int i = 0; i < 16; i++) x[i] = a0 + i + threadIdx.x * 1e-9;
  double m = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = fma(x[i], m, 1e-3);
  }
  double s = 0;
  for (int i = 0; i < 16; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// even waves MFMA, odd waves VALU fma (co-execution test); 512 threads = 2 waves per SIMD
__global__ void __launch_bounds__(512) k_mix(double* out, int iters, double a0, double b0) {
  int wave = threadIdx.x >> 6;
  double s = 0;
  if (wave < 4) {
    d4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (d4){0,0,0,0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double x[16];
    for (int i = 0; i < 16; i++) x[i] = a0 + i + threadIdx.x * 1e-9;
    double m = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = fma(x[i], m, 1e-3);
    }
    for (int i = 0; i < 16; i++) s += x[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_log10(double* out, int iters, double a0) {
  double x[8];
  for (int i = 0; i < 8; i++) x[i] = a0 + 0.01 * i + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = 0.5 + 0.01 * log10(x[i]) * log10(x[i]);
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_div(double* out, int iters, double a0) {
  double x[8];
  for (int i = 0; i < 8; i++) x[i] = a0 + 0.01 * i + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = 1.0 / (x[i] + 0.5);
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// layout check: D = A(16x4) * B(4x16), one wave
__global__ void k_layout(const double* A, const double* B, double* D) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];   // A[row=l&15][k=l>>4]
  double b = B[(l >> 4) * 16 + (l & 15)];  // B[k=l>>4][col=l&15]
  d4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
}

template<typename F> float timeit(F f, int reps=5) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
  double* out; CK(hipMalloc(&out, sizeof(double) * 1024 * 4096));
  // layout
  {
    std::vector<double> A(64), B(64), D(256);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = (i + 1) * 100 + k;      // asymmetric
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = (k + 1) * 7 + j * 3 + (j == 5);
    double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
      int row = (l >> 4) + 4 * r, col = l & 15;
      double ref = 0; for (int k = 0; k < 4; k++) ref += A[row * 4 + k] * B[k * 16 + col];
      if (ref != D[l * 4 + r]) bad++;
    }
    printf("f64 mfma layout check (row=(l>>4)+4r, col=l&15): %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  }
  int iters = 2000;
  int grid = cus * 8;
  for (int bs : {256}) {   // the kernels are built for 256-thread blocks (__launch_bounds__): a 512 launch fails
    {
      float ms = timeit([&]{ k_mfma<4><<<grid, bs>>>(out, iters, 1.0, 1e-3); });
      double nm = (double)grid * (bs / 64) * iters * 4; double fl = nm * 2048;
      printf("mfma_f64 acc4 block %d: %.3f ms  %.2f TFLOP/s  cycles/mfma/SIMD=%.1f (at %.0f MHz)\n", bs, ms, fl / ms / 1e9,
             ms * 1e-3 * clk / (nm / (cus * 4)), clk / 1e6);
    }
    {
      float ms = timeit([&]{ k_mfma<1><<<grid, bs>>>(out, iters * 4, 1.0, 1e-3); });
      double nm = (double)grid * (bs / 64) * iters * 4; double fl = nm * 2048;
      printf("mfma_f64 acc1(dep chain) block %d: %.3f ms  %.2f TFLOP/s cycles/mfma/SIMD=%.1f\n", bs, ms, fl / ms / 1e9, ms * 1e-3 * clk / (nm / (cus * 4)));
    }
    {
      float ms = timeit([&]{ k_fma<<<grid, bs>>>(out, iters * 4, 1.0, 0.999); });
      double ni = (double)grid * (bs / 64) * iters * 4 * 16; double fl = ni * 128;
      printf("v_fma_f64 block %d: %.3f ms  %.2f TFLOP/s cycles/inst/SIMD=%.2f\n", bs, ms, fl / ms / 1e9, ms * 1e-3 * clk / (ni / (cus * 4)));
    }
  }
  {
    float ms = timeit([&]{ k_mix<<<grid, 512>>>(out, iters, 1.0, 1e-3); });
    double nm = (double)grid * 4 * iters * 4; double ni = (double)grid * 4 * iters * 16;
    printf("mix (4 mfma waves + 4 fma waves per WG): %.3f ms  mfma %.2f TF + valu %.2f TF ; mfma-only time would be %.3f\n", ms, nm * 2048 / ms / 1e9, ni * 128 / ms / 1e9,
           nm / (cus * 4) * 64 / clk * 1e3);
  }
  {
    int it2 = 200;
    float ms = timeit([&]{ k_log10<<<grid, 256>>>(out, it2, 0.7); });
    double n = (double)grid * 256 * it2 * 8 * 2;
    printf("log10(double): %.3f ms  %.2f G log10/s ; cycles per wave-log10 per SIMD = %.1f\n", ms, n / ms / 1e6, ms * 1e-3 * clk / (n / 64 / (cus * 4)));
    ms = timeit([&]{ k_div<<<grid, 256>>>(out, it2 * 4, 0.7); });
    n = (double)grid * 256 * it2 * 4 * 8;
    printf("1.0/x (double): %.3f ms  %.2f G div/s ; cycles per wave-div per SIMD = %.1f\n", ms, n / ms / 1e6, ms * 1e-3 * clk / (n / 64 / (cus * 4)));
  }
  return 0;
}
