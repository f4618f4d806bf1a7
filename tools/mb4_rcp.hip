// accuracy of v_rcp_f64 / v_rsq_f64 (raw and after Newton steps) vs correctly rounded host results
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double v = x[i];
  double y = __builtin_amdgcn_rcp(v);
  o[i] = y;
  double e = fma(-v, y, 1.0); double y1 = fma(y, e, y); o[n + i] = y1;
  e = fma(-v, y1, 1.0); o[2 * n + i] = fma(y1, e, y1);
  double r = __builtin_amdgcn_rsq(v);
  o[3 * n + i] = r;
  double h = 0.5 * v * r; double r1 = fma(r, fma(-h, r, 0.5), r); o[4 * n + i] = r1;
  h = 0.5 * v * r1; o[5 * n + i] = fma(r1, fma(-h, r1, 0.5), r1);
}
int main() {
  int n = 1 << 20; std::vector<double> x(n), o(6 * n);
  srand(3); for (auto& v : x) v = exp((rand() / (double)RAND_MAX - 0.5) * 40.0);
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n); hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
  const char* names[6] = {"rcp raw", "rcp +1NR", "rcp +2NR", "rsq raw", "rsq +1NR", "rsq +2NR"};
  for (int t = 0; t < 6; t++) {
    double worst = 0;
    for (int i = 0; i < n; i++) {
      long double ref = t < 3 ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
      double rel = fabs((double)(((long double)o[t * n + i] - ref) / ref)); if (rel > worst) worst = rel;
    }
    printf("%-10s max rel err %.3e (%.2f ulp)\n", names[t], worst, worst / 1.11e-16);
  }
  return 0;
}
