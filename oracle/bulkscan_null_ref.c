/*
 * bulkscan_null_ref.c -- C/OpenMP restatement of the reference's null-exact bulkscan, operation by operation.
 *
 * TEST INFRASTRUCTURE, like oracle/bulklmm_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may call it; the product path (bulklmm.jl_amd/) never does.  It exists (a) as the CPU baseline timed beside the GPU
 * on the GPU box's host cores (the reference is pure Julia and cannot run here), and (b) as a second,
 * independently written restatement that tests/test_oracle_kats.py holds against the NumPy one.
 *
 * Follows, per function:
 *   transform_rotation   src/transform_helpers.jl:1-54     eigen(K) (here: cyclic Jacobi), Ut*y, Ut*[Z G]
 *   makeweights          src/lmm.jl:15-33
 *   wls                  src/wls.jl:27-97                  Householder QR least squares, sigma2, ell (ML / REML, prior)
 *   fitlmm + gridbrent   src/lmm.jl:56-86, src/gridbrent.jl:9-24; Optim.jl Brent() restated from its published algorithm
 *   univar_liteqtl       src/bulkscan_helpers.jl:127-150   sqrt|w| row scaling of y0, X0_intercept, X0_covar (n x p, per trait)
 *   computeR_LMM         src/bulkscan_helpers.jl:47-64     resid() of both sides (src/wls.jl:221-241), norms, colDivide!, X00'y00
 *   r2lod                src/bulkscan_helpers.jl:22-24     -(n/2) log10(1.0 - r^2)
 *   bulkscan_null        src/bulkscan.jl:212-314           trait loop (Threads.@threads over blocks there, OpenMP here)
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC bulkscan_null_ref.c -o libblmm_oracle_c.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CMAXR 16

/* cyclic Jacobi eigen-decomposition of the symmetric n x n matrix A (column-major, destroyed); V: eigenvectors as columns;
 * eigenvalues ascending in lam (LAPACK's `eigen` order, src/transform_helpers.jl:23) */
static void sym_eigen(double* A, int n, double* lam, double* V) {
  for (int i = 0; i < n * n; ++i) V[i] = 0.0;
  for (int i = 0; i < n; ++i) V[i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) { if (i != j) off += A[j * n + i] * A[j * n + i]; else dia += A[j * n + i] * A[j * n + i]; }
    if (off <= 1e-32 * dia) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[q * n + p];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {   /* columns p, q */
          const double akp = A[p * n + k], akq = A[q * n + k];
          A[p * n + k] = c * akp - s * akq; A[q * n + k] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {   /* rows p, q */
          const double apk = A[k * n + p], aqk = A[k * n + q];
          A[k * n + p] = c * apk - s * aqk; A[k * n + q] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[p * n + k], vkq = V[q * n + k];
          V[p * n + k] = c * vkp - s * vkq; V[q * n + k] = s * vkp + c * vkq;
        }
      }
  }
  /* ascending order */
  int* idx = (int*)malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  for (int i = 1; i < n; ++i) { const int v = idx[i]; int j = i - 1; while (j >= 0 && A[idx[j] * n + idx[j]] > A[v * n + v]) { idx[j + 1] = idx[j]; --j; } idx[j + 1] = v; }
  double* Vs = (double*)malloc(sizeof(double) * n * n);
  for (int i = 0; i < n; ++i) { lam[i] = A[idx[i] * n + idx[i]]; memcpy(Vs + (size_t)i * n, V + (size_t)idx[i] * n, sizeof(double) * n); }
  memcpy(V, Vs, sizeof(double) * n * n);
  free(Vs); free(idx);
}

/* Householder QR of the n x c matrix X (column-major, overwritten by R above the diagonal and the reflectors below), as
 * Julia's qr(X); tau[c] */
static void house_qr(double* X, int n, int c, double* tau) {
  for (int k = 0; k < c; ++k) {
    double* x = X + (size_t)k * n;
    double nrm = 0.0;
    for (int i = k; i < n; ++i) nrm += x[i] * x[i];
    nrm = sqrt(nrm);
    if (nrm == 0.0) { tau[k] = 0.0; continue; }
    const double alpha = x[k], beta = -(alpha >= 0 ? nrm : -nrm);
    tau[k] = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    for (int i = k + 1; i < n; ++i) x[i] *= sc;
    x[k] = beta;
    for (int j = k + 1; j < c; ++j) {
      double* y = X + (size_t)j * n;
      double s = y[k];
      for (int i = k + 1; i < n; ++i) s += x[i] * y[i];
      s *= tau[k];
      y[k] -= s;
      for (int i = k + 1; i < n; ++i) y[i] -= s * x[i];
    }
  }
}
/* b = qr(X) \ y for one right-hand side (y is overwritten by Q'y) */
static void qr_solve(const double* QR, const double* tau, int n, int c, double* y, double* b) {
  for (int k = 0; k < c; ++k) {
    const double* x = QR + (size_t)k * n;
    double s = y[k];
    for (int i = k + 1; i < n; ++i) s += x[i] * y[i];
    s *= tau[k];
    y[k] -= s;
    for (int i = k + 1; i < n; ++i) y[i] -= s * x[i];
  }
  for (int k = c - 1; k >= 0; --k) {
    double s = y[k];
    for (int j = k + 1; j < c; ++j) s -= QR[(size_t)j * n + k] * b[j];
    b[k] = s / QR[(size_t)k * n + k];
  }
}

typedef struct { int n, c, reml; double pa, pb; const double* y; const double* X; const double* lam; double* ws; } NullFit;

/* wls(y, X, w, prior; reml).ell and sigma2 (src/wls.jl:27-97); ws: n*(c+2) + 2c doubles of scratch */
static double wls_ell(const NullFit* f, double h2, double* sigma2_out) {
  const int n = f->n, c = f->c;
  double* XX = f->ws; double* yy = XX + (size_t)n * c; double* yq = yy + n; double* tau = yq + n; double* b = tau + c;
  const double delta = h2 / (1.0 - h2);
  double sumlogw = 0.0;
  for (int k = 0; k < n; ++k) {
    const double w = 1.0 / (delta * f->lam[k] + 1.0);     /* makeweights */
    const double sw = sqrt(w);
    sumlogw += log(w);
    yy[k] = sw * f->y[k];
    for (int q = 0; q < c; ++q) XX[(size_t)q * n + k] = sw * f->X[(size_t)q * n + k];
  }
  double* QR = (double*)malloc(sizeof(double) * (size_t)n * c);
  memcpy(QR, XX, sizeof(double) * (size_t)n * c);
  house_qr(QR, n, c, tau);
  memcpy(yq, yy, sizeof(double) * n);
  qr_solve(QR, tau, n, c, yq, b);
  double logdet = 0.0;
  for (int q = 0; q < c; ++q) logdet += log(fabs(QR[(size_t)q * n + q]));
  logdet *= 2.0;
  free(QR);
  double rss0 = 0.0;
  for (int k = 0; k < n; ++k) {
    double fit = 0.0;
    for (int q = 0; q < c; ++q) fit += XX[(size_t)q * n + k] * b[q];
    const double r = yy[k] - fit;
    rss0 += r * r;
  }
  const double prior_df = f->pb > 0.0 ? f->pb + 2.0 : f->pb;
  const double s2 = (rss0 + f->pa * f->pb) / ((f->reml ? (n - c) : n) + prior_df);
  double ell = -0.5 * ((n + f->pb) * log(s2) - sumlogw + (rss0 + f->pa * f->pb) / s2);
  if (f->reml) ell += 0.5 * (c * log(s2) - logdet);
  if (sigma2_out) *sigma2_out = s2;
  return ell;
}

/* Optim.jl optimize(f, a, b, Brent()) on g(h2) = -ell(h2) */
static void brent_min(const NullFit* f, double x_lower, double x_upper, double* xmin, double* fmin) {
  const double GOLDEN = 0.5 * (3.0 - sqrt(5.0)), REL = sqrt(2.220446049250313e-16), ABS = 2.220446049250313e-16;
  double new_x0 = x_lower + GOLDEN * (x_upper - x_lower);
  double new_minimizer = new_x0, new_minimum = -wls_ell(f, new_x0, NULL);
  double step = 0.0, old_step = 0.0;
  double old_minimizer = new_minimizer, old_old_minimizer = new_minimizer, old_minimum = new_minimum, old_old_minimum = new_minimum;
  for (int iteration = 0; iteration < 1000;) {
    double p = 0.0, q = 0.0;
    const double x_tol = REL * fabs(new_minimizer) + ABS;
    const double x_mid = (x_upper + x_lower) / 2;
    if (fabs(new_minimizer - x_mid) <= 2 * x_tol - (x_upper - x_lower) / 2) break;
    ++iteration;
    if (fabs(old_step) > x_tol) {
      const double r = (new_minimizer - old_minimizer) * (new_minimum - old_old_minimum);
      q = (new_minimizer - old_old_minimizer) * (new_minimum - old_minimum);
      p = (new_minimizer - old_old_minimizer) * q - (new_minimizer - old_minimizer) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p; else q = -q;
    }
    if (fabs(p) < fabs(q * old_step / 2) && p < q * (x_upper - new_minimizer) && p < q * (new_minimizer - x_lower)) {
      old_step = step;
      step = p / q;
      const double x_temp = new_minimizer + step;
      if ((x_temp - x_lower) < 2 * x_tol || (x_upper - x_temp) < 2 * x_tol) step = new_minimizer < x_mid ? x_tol : -x_tol;
    } else {
      old_step = new_minimizer < x_mid ? (x_upper - new_minimizer) : (x_lower - new_minimizer);
      step = GOLDEN * old_step;
    }
    const double new_x = fabs(step) >= x_tol ? new_minimizer + step : new_minimizer + (step > 0 ? x_tol : -x_tol);
    const double new_f = -wls_ell(f, new_x, NULL);
    if (new_f < new_minimum) {
      if (new_x < new_minimizer) x_upper = new_minimizer; else x_lower = new_minimizer;
      old_old_minimizer = old_minimizer; old_old_minimum = old_minimum;
      old_minimizer = new_minimizer; old_minimum = new_minimum;
      new_minimizer = new_x; new_minimum = new_f;
    } else {
      if (new_x < new_minimizer) x_lower = new_x; else x_upper = new_x;
      if (new_f <= old_minimum || old_minimizer == new_minimizer) {
        old_old_minimizer = old_minimizer; old_old_minimum = old_minimum;
        old_minimizer = new_x; old_minimum = new_f;
      } else if (new_f <= old_old_minimum || old_old_minimizer == new_minimizer || old_old_minimizer == old_minimizer) {
        old_old_minimizer = new_x; old_old_minimum = new_f;
      }
    }
  }
  *xmin = new_minimizer; *fmin = new_minimum;
}

/* Y n x m, G n x p, Covar n x ncov (may be NULL), K n x n, all column-major.  add_intercept != 0: [1 Covar] is the null design.
 * L p x m (ld = p), h2 m.  Returns 0, or -1 on a bad argument. */
/* h2_override (NULL or m values): the LOD columns are evaluated at THESE heritabilities instead of the search's own (the tests
 * hand in the device's estimates, so that every entry of L can be held to 1e-6 without the optimiser's stopping rule in between:
 * oracle/bulklmm_oracle.py's bulkscan_null(..., h2_override=...) does the same); h2_out then still receives the restatement's OWN
 * Brent estimates unless skip_search != 0 (h2_out = the override then). */
static int ref_bulkscan_null(const double* Y, int64_t n64, int64_t m, const double* G, int64_t p, const double* Covar, int64_t ncov,
                             int add_intercept, const double* K, double prior_variance, double prior_sample_size, int reml,
                             int optim_interval, double* L, double* h2_out, int nthreads, const double* h2_override, int skip_search) {
  const int n = (int)n64;
  const int c = (int)ncov + (add_intercept ? 1 : 0);
  if (n < 2 || c < 1 || c > CMAXR || c >= n || m < 0 || p < 0) return -1;
  if (optim_interval < 1) optim_interval = 1;
  /* transform_rotation */
  double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* U = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* lam = (double*)malloc(sizeof(double) * n);
  memcpy(A, K, sizeof(double) * (size_t)n * n);
  sym_eigen(A, n, lam, U);   /* U[:, i] = U + i*n */
  free(A);
  double* Z0 = (double*)malloc(sizeof(double) * (size_t)n * c);     /* rotated null covariates */
  double* X0 = (double*)malloc(sizeof(double) * (size_t)n * (p > 0 ? p : 1));
  double* Y0 = (double*)malloc(sizeof(double) * (size_t)n * (m > 0 ? m : 1));
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  for (int q = 0; q < c; ++q)
    for (int k = 0; k < n; ++k) {
      double s = 0.0;
      for (int i = 0; i < n; ++i) {
        const double z = (add_intercept && q == 0) ? 1.0 : Covar[(size_t)(q - (add_intercept ? 1 : 0)) * n + i];
        s += U[(size_t)k * n + i] * z;
      }
      Z0[(size_t)q * n + k] = s;
    }
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < p; ++j)
    for (int k = 0; k < n; ++k) { double s = 0.0; for (int i = 0; i < n; ++i) s += U[(size_t)k * n + i] * G[(size_t)j * n + i]; X0[(size_t)j * n + k] = s; }
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < m; ++j)
    for (int k = 0; k < n; ++k) { double s = 0.0; for (int i = 0; i < n; ++i) s += U[(size_t)k * n + i] * Y[(size_t)j * n + i]; Y0[(size_t)j * n + k] = s; }
  int bad = 0;
#pragma omp parallel
  {
    double* ws = (double*)malloc(sizeof(double) * ((size_t)n * (c + 2) + 2 * c));
    double* sqrtw = (double*)malloc(sizeof(double) * n);
    double* wy = (double*)malloc(sizeof(double) * n);
    double* wZ = (double*)malloc(sizeof(double) * (size_t)n * c);
    double* QR = (double*)malloc(sizeof(double) * (size_t)n * c);
    double* wx = (double*)malloc(sizeof(double) * n);
    double* tmp = (double*)malloc(sizeof(double) * n);
    double* Qe = (double*)malloc(sizeof(double) * (size_t)n * c);
    double tau[CMAXR], b[CMAXR];
#pragma omp for schedule(dynamic, 8)
    for (int64_t j = 0; j < m; ++j) {
      /* fitlmm: gridbrent over optim_interval sub-intervals of [0, 1], first smallest minimum wins */
      NullFit f = {n, c, reml, prior_variance, prior_sample_size, Y0 + (size_t)j * n, Z0, lam, ws};
      double best_x = 0.0, best_f = INFINITY;
      if (!(h2_override && skip_search))
        for (int s = 0; s < optim_interval; ++s) {
          double x, fx;
          brent_min(&f, (double)s / optim_interval, (double)(s + 1) / optim_interval, &x, &fx);
          if (fx < best_f) { best_f = fx; best_x = x; }
        }
      else best_x = h2_override[j];
      const double h2 = h2_override ? h2_override[j] : best_x;
      h2_out[j] = best_x;
      /* univar_liteqtl: sqrtw = sqrt.(abs.(makeweights(h2, lambda))); row scaling; computeR_LMM; r2lod */
      const double delta = h2 / (1.0 - h2);
      for (int k = 0; k < n; ++k) sqrtw[k] = sqrt(fabs(1.0 / (delta * lam[k] + 1.0)));
      for (int k = 0; k < n; ++k) wy[k] = sqrtw[k] * Y0[(size_t)j * n + k];
      for (int q = 0; q < c; ++q) for (int k = 0; k < n; ++k) wZ[(size_t)q * n + k] = sqrtw[k] * Z0[(size_t)q * n + k];
      memcpy(QR, wZ, sizeof(double) * (size_t)n * c);
      house_qr(QR, n, c, tau);
      /* Y00 = resid(wy, wZ) */
      memcpy(tmp, wy, sizeof(double) * n);
      qr_solve(QR, tau, n, c, tmp, b);
      double ny = 0.0;
      for (int k = 0; k < n; ++k) { double fit = 0.0; for (int q = 0; q < c; ++q) fit += wZ[(size_t)q * n + k] * b[q]; wy[k] -= fit; ny += wy[k] * wy[k]; }
      ny = sqrt(ny);
      if (fabs(ny) <= 2.220446049250313e-16) bad = 1;     /* colDivide!: "Dividing by zeros" (src/util.jl:69-71) */
      for (int k = 0; k < n; ++k) wy[k] /= ny;
      /* explicit thin Q (n x c, orthonormal columns) of wZ = Q R: resid(x, wZ) = x - Q (Q'x), the same projection that
       * `x - wZ * (qr(wZ) \ x)` (src/wls.jl:231-239) computes, in a form the compiler vectorises over the n individuals */
      for (int q = 0; q < c; ++q) {
        for (int k = 0; k < n; ++k) tmp[k] = (k == q) ? 1.0 : 0.0;
        for (int kk = c - 1; kk >= 0; --kk) {      /* Q e_q = H_0 ... H_(c-1) e_q */
          const double* x = QR + (size_t)kk * n;
          double sdot = tmp[kk];
          for (int i = kk + 1; i < n; ++i) sdot += x[i] * tmp[i];
          sdot *= tau[kk];
          tmp[kk] -= sdot;
          for (int i = kk + 1; i < n; ++i) tmp[i] -= sdot * x[i];
        }
        memcpy(Qe + (size_t)q * n, tmp, sizeof(double) * n);
      }
      double* restrict Lj = L + (size_t)j * p;
      const double scale = -(double)n / 2.0;
      for (int64_t i = 0; i < p; ++i) {
        const double* restrict x0 = X0 + (size_t)i * n;
        double tq[CMAXR];
        for (int q = 0; q < c; ++q) tq[q] = 0.0;
#pragma omp simd
        for (int k = 0; k < n; ++k) wx[k] = sqrtw[k] * x0[k];
        for (int q = 0; q < c; ++q) {
          const double* restrict qq = Qe + (size_t)q * n;
          double sdot = 0.0;
#pragma omp simd reduction(+ : sdot)
          for (int k = 0; k < n; ++k) sdot += qq[k] * wx[k];
          tq[q] = sdot;
        }
        for (int q = 0; q < c; ++q) {
          const double* restrict qq = Qe + (size_t)q * n;
          const double t = tq[q];
#pragma omp simd
          for (int k = 0; k < n; ++k) wx[k] -= t * qq[k];
        }
        double nx = 0.0, dot = 0.0;
#pragma omp simd reduction(+ : nx, dot)
        for (int k = 0; k < n; ++k) { nx += wx[k] * wx[k]; dot += wx[k] * wy[k]; }
        nx = sqrt(nx);
        if (fabs(nx) <= 2.220446049250313e-16) bad = 1;
        const double r = dot / nx;
        Lj[i] = scale * log10(1.0 - r * r);
      }
    }
    free(ws); free(sqrtw); free(wy); free(wZ); free(QR); free(wx); free(tmp); free(Qe);
  }
  free(U); free(lam); free(Z0); free(X0); free(Y0);
  return bad ? -8 : 0;
}

int blmm_ref_bulkscan_null(const double* Y, int64_t n64, int64_t m, const double* G, int64_t p, const double* Covar, int64_t ncov,
                           int add_intercept, const double* K, double prior_variance, double prior_sample_size, int reml,
                           int optim_interval, double* L, double* h2_out, int nthreads) {
  return ref_bulkscan_null(Y, n64, m, G, p, Covar, ncov, add_intercept, K, prior_variance, prior_sample_size, reml, optim_interval, L,
                           h2_out, nthreads, NULL, 0);
}

int blmm_ref_bulkscan_null_at(const double* Y, int64_t n64, int64_t m, const double* G, int64_t p, const double* Covar, int64_t ncov,
                              int add_intercept, const double* K, double prior_variance, double prior_sample_size, int reml,
                              int optim_interval, double* L, double* h2_out, int nthreads, const double* h2_override, int skip_search) {
  return ref_bulkscan_null(Y, n64, m, G, p, Covar, ncov, add_intercept, K, prior_variance, prior_sample_size, reml, optim_interval, L,
                           h2_out, nthreads, h2_override, skip_search);
}

int blmm_ref_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
