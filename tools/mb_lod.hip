// accuracy of the LOD epilogue's logarithm (fastmath.h: fast_lod5 through the directly indexed table, lod_out_of_range below 2^-4)
// against a long-double log10 on the host: hipcc --offload-arch=gfx950 -O2 -I bulklmm.jl_amd/csrc tools/mb_lod.hip -o /tmp/mb_lod
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
#include "fastmath.h"
using namespace blmm;
__global__ void k(const double* x, double* o, int n, const double* gtab, double scale) {
  __shared__ dpair s_lod[BLMM_LOD_TABLE_N];
  LodStage<256> st;
  lod_stage_load<256>(st, gtab);
  lod_stage_store<256>(st, s_lod, scale);
  __syncthreads();
  const LodPoly5 lp = make_lod_poly5(scale);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double u = x[i];
    int nn = 0;
    o[i] = lod_fast_ok(u) ? fast_lod5(u, s_lod, lp) : lod_out_of_range(u, s_lod, lp, scale, true, &nn);
  }
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), o(n);
  srand(7);
  for (int i = 0; i < n; ++i) {
    const double t = rand() / (double)RAND_MAX;
    switch (i & 7) {
      case 0: x[i] = t; break;                                        // uniform (0, 1)
      case 1: x[i] = 1.0 - exp(-40.0 * t); break;                     // 1 - tiny .. 1 (r^2 -> 0: what null markers give)
      case 2: x[i] = 1.0 - 0.05 * t; break;                           // typical null tests at n ~ 79
      case 3: x[i] = exp(-30.0 * t); break;                           // down to 1e-13 (strong signals)
      case 4: x[i] = ldexp(0.5 + 0.5 * t, -(int)(1000 * t)); break;   // far below 2^-4, towards the subnormals
      case 5: x[i] = nextafter(ldexp(1.0, -(int)(8 * t)), i & 8 ? 0.0 : 2.0); break;   // around the octave boundaries
      case 6: x[i] = 0.0625 * (1.0 + 1e-3 * (t - 0.5)); break;        // around the table's lower edge 2^-4
      default: x[i] = 1.0 - ldexp(1.0, -(int)(53 * t)); break;        // 1 - 2^-k
    }
  }
  x[0] = 1.0; x[1] = 0.0625; x[2] = 4.9e-324; x[3] = 2.2250738585072014e-308;
  const double scale = -39.5;
  double *dx, *dout, *dt;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 8); hipMalloc(&dt, sizeof(blmm_lod_table_host));
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dt, blmm_lod_table_host, sizeof(blmm_lod_table_host), hipMemcpyHostToDevice);
  k<<<1024, 256>>>(dx, dout, n, dt, scale);
  hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
  double worst[8] = {0}; int hist[8] = {0};
  for (int i = 0; i < n; ++i) {
    const long double ref = (long double)scale * log10l((long double)x[i]);
    // u = 0 (case 7 with k = 0): +Inf on both sides is exact agreement, not NaN
    const double rel = (std::isinf((double)ref) && o[i] == (double)ref) ? 0.0 : ref == 0 ? fabs(o[i]) : fabs((double)(((long double)o[i] - ref) / ref));
    if (rel > worst[i & 7]) worst[i & 7] = rel;
    const double ulp = rel / 1.11e-16;
    hist[ulp <= 0.5 ? 0 : ulp <= 1 ? 1 : ulp <= 2 ? 2 : ulp <= 4 ? 3 : ulp <= 16 ? 4 : ulp <= 1e3 ? 5 : ulp <= 1e6 ? 6 : 7]++;
  }
  const char* nm[8] = {"uniform (0,1)", "1 - exp(-40 t)", "1 - 0.05 t", "exp(-30 t)", "2^-1000 t", "octave edges", "around 2^-4", "1 - 2^-k"};
  for (int c = 0; c < 8; ++c) printf("%-16s max rel err %.3e (%.2f ulp)\n", nm[c], worst[c], worst[c] / 1.11e-16);
  printf("ulp histogram (<=0.5, 1, 2, 4, 16, 1e3, 1e6, more):");
  for (int b = 0; b < 8; ++b) printf(" %d", hist[b]);
  printf("\nfirst: lod(1) = %g, lod(2^-4) = %.17g (ref %.17g), lod(denorm_min) = %.17g (ref %.17g)\n", o[0], o[1], -39.5 * log10(0.0625), o[2],
         (double)(-39.5L * log10l((long double)4.9e-324)));   // the DOUBLE denorm_min (a long double literal 4.9e-324L is another number)
  return 0;
}
