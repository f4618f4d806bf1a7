/* Plain-C caller of libbulklmm_hip.so (what a Julia ccall / any FFI host does): reads Y, G, K (column-major float64) from
 * a raw file written by the test, runs blmm_kinship + blmm_bulkscan (null-grid and null-exact) through the HOST-pointer
 * entry points declared in include/bulklmm_hip.h and writes the results back; then the 0.2.3 additions (L_out == NULL with the
 * blmm_last_* consumers, blmm_bulkscan_reduced with its blmm_reduced struct, blmm_set_tuning), checked inside this program against
 * the matrix it holds.  Compiled with gcc, no HIP, no Python.
 *   usage: c_abi_smoke <in.bin> <out.bin>
 *   in : int64 n, m, p, ngrid | Y n*m | G n*p | grid ngrid
 *   out: K n*n | L_grid p*m | h2_grid m | L_exact p*m | h2_exact m */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bulklmm_hip.h"

static void die(blmm_ctx* ctx, const char* what, int rc) {
  fprintf(stderr, "%s failed: %d (%s) %s\n", what, rc, blmm_err_string(rc), ctx ? blmm_last_error(ctx) : "");
  exit(2);
}

int main(int argc, char** argv) {
  if (argc != 3) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int64_t hdr[4];
  if (fread(hdr, sizeof(int64_t), 4, f) != 4) return 1;
  const int64_t n = hdr[0], m = hdr[1], p = hdr[2], ng = hdr[3];
  double* Y = malloc(sizeof(double) * n * m);
  double* G = malloc(sizeof(double) * n * p);
  double* grid = malloc(sizeof(double) * ng);
  if (fread(Y, sizeof(double), n * m, f) != (size_t)(n * m) || fread(G, sizeof(double), n * p, f) != (size_t)(n * p) ||
      fread(grid, sizeof(double), ng, f) != (size_t)ng) return 1;
  fclose(f);
  double* K = malloc(sizeof(double) * n * n);
  double* L1 = malloc(sizeof(double) * p * m);
  double* L2 = malloc(sizeof(double) * p * m);
  double* h1 = malloc(sizeof(double) * m);
  double* h2 = malloc(sizeof(double) * m);

  if (blmm_version() != BLMM_VERSION) { fprintf(stderr, "version mismatch\n"); return 2; }
  blmm_ctx* ctx = NULL;
  int rc = blmm_create(0, NULL, &ctx);
  if (rc) die(NULL, "blmm_create", rc);
  if ((rc = blmm_kinship(ctx, G, n, p, K))) die(ctx, "blmm_kinship", rc);

  blmm_opts o;
  blmm_default_opts(&o);
  blmm_status st;
  o.method = BLMM_NULL_GRID;
  if ((rc = blmm_bulkscan(ctx, &o, Y, n, m, G, p, NULL, 0, K, NULL, grid, ng, L1, h1, &st))) die(ctx, "blmm_bulkscan(null-grid)", rc);
  o.method = BLMM_NULL_EXACT;
  if ((rc = blmm_bulkscan(ctx, &o, Y, n, m, G, p, NULL, 0, K, NULL, NULL, 0, L2, h2, &st))) die(ctx, "blmm_bulkscan(null-exact)", rc);
  /* an error path: a grid point of 1 must be refused with the reference's message */
  double bad = 1.0;
  o.method = BLMM_NULL_GRID;
  rc = blmm_bulkscan(ctx, &o, Y, n, m, G, p, NULL, 0, K, NULL, &bad, 1, L1 + 0, h1 + 0, NULL);
  if (rc != BLMM_ERR_H2_ONE || strcmp(blmm_last_error(ctx), "Heritability of 1 is not allowed.") != 0) {
    fprintf(stderr, "expected BLMM_ERR_H2_ONE, got %d (%s)\n", rc, blmm_last_error(ctx));
    return 3;
  }
  /* (the failed call must not have touched the outputs: recompute the grid scan for the file) */
  if ((rc = blmm_bulkscan(ctx, &o, Y, n, m, G, p, NULL, 0, K, NULL, grid, ng, L1, h1, &st))) die(ctx, "blmm_bulkscan(null-grid, again)", rc);

  /* 0.2.3: the call without the matrix on the host.  (a) L_out == NULL: L stays in HBM and the blmm_last_* consumers serve it;
   * (b) blmm_bulkscan_reduced: the same reductions out of the scan kernels, L never written.  Both must reproduce what this
   * program computes from the L2 it already holds (null-exact), bit for bit. */
  {
    double* mx = malloc(sizeof(double) * m); int64_t* ax = malloc(sizeof(int64_t) * m); double* hk = malloc(sizeof(double) * m);
    double* mr = malloc(sizeof(double) * m); int64_t* ar = malloc(sizeof(int64_t) * m); double* hr = malloc(sizeof(double) * m);
    const int64_t cap = p * m;
    int32_t* ti = malloc(sizeof(int32_t) * cap); int32_t* tj = malloc(sizeof(int32_t) * cap); double* tl = malloc(sizeof(double) * cap);
    int64_t cnt = -1, pp = 0, mm = 0;
    const double thr = 1.0;
    o.method = BLMM_NULL_EXACT;
    if ((rc = blmm_bulkscan(ctx, &o, Y, n, m, G, p, NULL, 0, K, NULL, NULL, 0, NULL, hk, NULL))) die(ctx, "blmm_bulkscan(L_out = NULL)", rc);
    if ((rc = blmm_last_dims(ctx, &pp, &mm)) || pp != p || mm != m) die(ctx, "blmm_last_dims", rc);
    if ((rc = blmm_last_lod_colmax(ctx, mx, ax))) die(ctx, "blmm_last_lod_colmax", rc);
    blmm_reduced r;
    memset(&r, 0, sizeof(r));
    r.colmax = mr; r.argmax = ar; r.want_triplets = 1; r.thr = thr; r.cap = cap; r.ti = ti; r.tj = tj; r.tlod = tl; r.count = &cnt;
    if ((rc = blmm_bulkscan_reduced(ctx, &o, Y, n, m, G, p, NULL, 0, K, NULL, NULL, 0, &r, hr, NULL))) die(ctx, "blmm_bulkscan_reduced", rc);
    if (blmm_last_reduced_route(ctx) != 1) { fprintf(stderr, "reduced: expected the fused route\n"); return 4; }
    int64_t want = 0;
    for (int64_t j = 0; j < m; ++j) {
      int64_t bi = -1; double best = -1.0 / 0.0;
      for (int64_t i = 0; i < p; ++i) { const double v = L2[j * p + i]; if (v > best) { best = v; bi = i; } if (v > thr) ++want; }
      if (mx[j] != best || ax[j] != bi || mr[j] != best || ar[j] != bi || hk[j] != h2[j] || hr[j] != h2[j]) {
        fprintf(stderr, "reduced / resident consumers disagree with the stored matrix at trait %lld\n", (long long)j); return 4;
      }
    }
    if (cnt != want) { fprintf(stderr, "reduced: %lld triplets, expected %lld\n", (long long)cnt, (long long)want); return 4; }
    for (int64_t e = 0; e < cnt; ++e)
      if (tl[e] != L2[(int64_t)tj[e] * p + ti[e]] || !(tl[e] > thr)) { fprintf(stderr, "reduced: triplet %lld is wrong\n", (long long)e); return 4; }
    /* tuning keys: the switches that were environment variables up to 0.2.2 */
    double tv = 0.0;
    if ((rc = blmm_get_tuning(ctx, "lr_tol", &tv)) || tv != 1e-13) die(ctx, "blmm_get_tuning", rc);
    if ((rc = blmm_set_tuning(ctx, "exact_full_rank", 1.0))) die(ctx, "blmm_set_tuning", rc);
    if (blmm_set_tuning(ctx, "no_such_key", 1.0) == 0 || blmm_set_tuning(ctx, "lr_segments", 99.0) == 0) { fprintf(stderr, "set_tuning accepted nonsense\n"); return 4; }
    if ((rc = blmm_set_tuning(ctx, "defaults", 0.0))) die(ctx, "blmm_set_tuning(defaults)", rc);
    free(mx); free(ax); free(hk); free(mr); free(ar); free(hr); free(ti); free(tj); free(tl);
  }
  blmm_destroy(ctx);

  f = fopen(argv[2], "wb");
  if (!f) return 1;
  fwrite(K, sizeof(double), n * n, f);
  fwrite(L1, sizeof(double), p * m, f);
  fwrite(h1, sizeof(double), m, f);
  fwrite(L2, sizeof(double), p * m, f);
  fwrite(h2, sizeof(double), m, f);
  fclose(f);
  printf("c_abi_smoke ok: n=%lld m=%lld p=%lld rank=%lld\n", (long long)n, (long long)m, (long long)p, (long long)st.lowrank_rank);
  return 0;
}
