/* ORACLE (test / benchmark infrastructure) -- plain-C restatement of the reference's Krylov inner loop.
 *
 * The reference solves the reduced system with `method(A_reduced, b_reduced; kwargs...)`
 * (/root/reference/src/solver.jl:173-183): IterativeSolvers 0.9.4 on a SparseMatrixCSC, i.e. a sparse
 * mat-vec plus BLAS-1 on ONE thread.  IterativeSolvers is a Julia package that is not vendored; this file
 * restates the published algorithms it implements (van der Vorst 1992 BiCGStab, Hestenes-Stiefel CG) with a
 * CSR mat-vec, and is used (1) by tests to check oracle/penguin_oracle.py::bicgstab_ref and (2) by
 * bench.py's `cpu_baseline` leg, timed on the GPU box's host cores.  nthreads = 1 mirrors the reference's
 * single-threaded path; nthreads > 1 uses OpenMP over rows.  The product never links or loads this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static void spmv(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) s += v[k] * x[ci[k]];
    y[r] = s;
  }
}

static double dot(int64_t n, const double* a, const double* b) {
  double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

void krylov_ref_spmv(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, const double* x, double* y,
                     int nthreads) {
#ifdef _OPENMP
  omp_set_num_threads(nthreads > 0 ? nthreads : 1);
#endif
  spmv(n, rp, ci, v, x, y);
}

/* returns iterations; x must hold n doubles (zero initial guess is applied here) */
int krylov_ref_bicgstab(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, const double* b, double* x,
                        double reltol, double abstol, int maxiter, int nthreads, double* resnorm_out) {
#ifdef _OPENMP
  omp_set_num_threads(nthreads > 0 ? nthreads : 1);
#endif
  double* r = (double*)malloc(sizeof(double) * n);
  double* rh = (double*)malloc(sizeof(double) * n);
  double* p = (double*)calloc(n, sizeof(double));
  double* vv = (double*)calloc(n, sizeof(double));
  double* s = (double*)malloc(sizeof(double) * n);
  double* t = (double*)malloc(sizeof(double) * n);
  for (int64_t i = 0; i < n; ++i) { x[i] = 0.0; r[i] = b[i]; rh[i] = b[i]; }
  const double bb = dot(n, b, b);
  double tol2 = reltol * reltol * bb;
  if (abstol * abstol > tol2) tol2 = abstol * abstol;
  double rr = bb, rho_old = 1.0, alpha = 1.0, omega = 1.0, rho = bb, rhat2 = bb;
  int it = 0, restart = 0;
  while (rr > tol2 && it < maxiter) {
    ++it;
    if (restart) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = r[i];
      restart = 0;
    } else {
      const double beta = (rho / rho_old) * (alpha / omega);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = r[i] + beta * (p[i] - omega * vv[i]);
    }
    spmv(n, rp, ci, v, p, vv);
    const double den = dot(n, rh, vv);
    const int force = den == 0.0;
    alpha = force ? 0.0 : rho / den; /* (rhat,Ap) == 0: minimal-residual half step, then restart */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) s[i] = r[i] - alpha * vv[i];
    spmv(n, rp, ci, v, s, t);
    const double tt = dot(n, t, t);
    omega = tt != 0.0 ? dot(n, t, s) / tt : 0.0;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      x[i] += alpha * p[i] + omega * s[i];
      r[i] = s[i] - omega * t[i];
    }
    rho_old = rho;
    rho = dot(n, rh, r);
    rr = dot(n, r, r);
    if (rr <= tol2) break;
    if (omega == 0.0 || force || rho * rho < 1e-20 * rhat2 * rr) { /* restart with rhat := r (same rule as pg_krylov.hip) */
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) rh[i] = r[i];
      rho = rhat2 = rr;
      alpha = omega = 1.0;
      restart = 1;
    }
  }
  if (resnorm_out) *resnorm_out = sqrt(rr);
  free(r); free(rh); free(p); free(vv); free(s); free(t);
  return it;
}

/* The iteration pg_krylov.hip runs where its Gershgorin test admits it (not something the reference does -- here so that
 * tests can check the algorithm on the host and bench.py can time the SAME algorithm on the host cores): BiCGStab on
 *     C y = b - A x0,   C = A q(A) = I - R(A),   x = x0 + q(A) y,
 * R(A) = prod_k (I - tau_k A) the Chebyshev residual polynomial of degree m on [1 - g, 1 + g] in product form (roots taken
 * from both ends of the interval in turn), convergence tested on the weighted residual ||w .* r|| <= reltol ||w .* b|| (w =
 * the row scaling S of the equilibrated system: the residual in units of x; NULL: unweighted) after both halves of an
 * iteration.  m = 0: the plain iteration with the same tests.  Returns iterations (an accepted half step counts as one);
 * *nmv = products with A. */
static void poly_taus(int m, double g, double* tau) {
  double lam[16];
  for (int k = 0; k < m; ++k) lam[k] = 1.0 + g * cos(M_PI * (2.0 * k + 1.0) / (2.0 * m));
  for (int k = 0, lo = 0, hi = m - 1; k < m; ++k) tau[k] = 1.0 / ((k & 1) ? lam[hi--] : lam[lo++]);
}

static double wdot(int64_t n, const double* w, const double* a) {
  double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (int64_t i = 0; i < n; ++i) { const double t = (w ? w[i] : 1.0) * a[i]; s += t * t; }
  return s;
}

/* out = C in = in - R(A) in; wa, wb: work vectors */
static void apply_c(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, int m, const double* tau, const double* in,
                    double* out, double* wa, double* wb, int64_t* nmv) {
  if (m == 0) { spmv(n, rp, ci, v, in, out); *nmv += 1; return; }
  const double* src = in;
  for (int k = 0; k < m; ++k) {
    double* dst = k + 1 == m ? out : ((k & 1) ? wb : wa);
    spmv(n, rp, ci, v, src, dst);
    *nmv += 1;
    const double tk = tau[k];
    if (k + 1 < m) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) dst[i] = src[i] - tk * dst[i];
    } else {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) dst[i] = in[i] - (src[i] - tk * dst[i]);
    }
    src = dst;
  }
}

int krylov_ref_bicgstab_poly(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, const double* b, double* x,
                             const double* x0, const double* wts, int m, double g, double reltol, double abstol, int maxiter,
                             int nthreads, double* resnorm_out, int64_t* nmv_out) {
#ifdef _OPENMP
  omp_set_num_threads(nthreads > 0 ? nthreads : 1);
#endif
  if (m < 2) m = 0;
  if (m > 16) m = 16;
  double tau[16];
  if (m) poly_taus(m, g, tau);
  double* r = (double*)malloc(sizeof(double) * n);
  double* rh = (double*)malloc(sizeof(double) * n);
  double* p = (double*)calloc(n, sizeof(double));
  double* vv = (double*)calloc(n, sizeof(double));
  double* t = (double*)malloc(sizeof(double) * n);
  double* y = (double*)calloc(n, sizeof(double));
  double* wa = (double*)malloc(sizeof(double) * n);
  double* wb = (double*)malloc(sizeof(double) * n);
  int64_t nmv = 0;
  if (x0) {
    spmv(n, rp, ci, v, x0, r);
    nmv += 1;
    for (int64_t i = 0; i < n; ++i) { x[i] = x0[i]; r[i] = b[i] - r[i]; rh[i] = r[i]; }
  } else {
    for (int64_t i = 0; i < n; ++i) { x[i] = 0.0; r[i] = b[i]; rh[i] = b[i]; }
  }
  double tol2 = reltol * reltol * wdot(n, wts, b);
  if (abstol * abstol > tol2) tol2 = abstol * abstol;
  double rrw = wdot(n, wts, r), rr = dot(n, r, r);
  double rho_old = 1.0, alpha = 1.0, omega = 1.0, rho = rr, rhat2 = rr;
  int it = 0, restart = 0;
  while (rrw > tol2 && it < maxiter) {
    ++it;
    if (it == 1 || restart) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = r[i];
      restart = 0;
    } else {
      const double beta = (rho / rho_old) * (alpha / omega);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = r[i] + beta * (p[i] - omega * vv[i]);
    }
    apply_c(n, rp, ci, v, m, tau, p, vv, wa, wb, &nmv);
    const double den = dot(n, rh, vv);
    const int force = den == 0.0;
    alpha = force ? 0.0 : rho / den;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) r[i] -= alpha * vv[i]; /* r holds s */
    rrw = wdot(n, wts, r);
    if (rrw <= tol2) { /* half step accepted */
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) y[i] += alpha * p[i];
      break;
    }
    apply_c(n, rp, ci, v, m, tau, r, t, wa, wb, &nmv);
    const double tt = dot(n, t, t);
    omega = tt != 0.0 ? dot(n, t, r) / tt : 0.0;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      y[i] += alpha * p[i] + omega * r[i];
      r[i] -= omega * t[i];
    }
    rho_old = rho;
    rho = dot(n, rh, r);
    rr = dot(n, r, r);
    rrw = wdot(n, wts, r);
    if (rrw <= tol2) break;
    if (omega == 0.0 || force || rho * rho < 1e-20 * rhat2 * rr) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) rh[i] = r[i];
      rho = rhat2 = rr;
      alpha = omega = 1.0;
      restart = 1;
    }
  }
  /* x = x0 + q(A) y,  q(A) y = sum_k tau_k w_(k-1),  w_0 = y,  w_k = w_(k-1) - tau_k A w_(k-1) */
  if (m == 0) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] += y[i];
  } else {
    const double* src = y;
    for (int k = 0; k < m; ++k) {
      const double tk = tau[k];
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) x[i] += tk * src[i];
      if (k + 1 == m) break;
      double* dst = (k & 1) ? wb : wa;
      spmv(n, rp, ci, v, src, dst);
      nmv += 1;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) dst[i] = src[i] - tk * dst[i];
      src = dst;
    }
  }
  if (resnorm_out) *resnorm_out = sqrt(rrw);
  if (nmv_out) *nmv_out = nmv;
  free(r); free(rh); free(p); free(vv); free(t); free(y); free(wa); free(wb);
  return it;
}

int krylov_ref_cg(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, const double* b, double* x,
                  double reltol, double abstol, int maxiter, int nthreads, double* resnorm_out) {
#ifdef _OPENMP
  omp_set_num_threads(nthreads > 0 ? nthreads : 1);
#endif
  double* r = (double*)malloc(sizeof(double) * n);
  double* p = (double*)malloc(sizeof(double) * n);
  double* q = (double*)malloc(sizeof(double) * n);
  for (int64_t i = 0; i < n; ++i) { x[i] = 0.0; r[i] = b[i]; p[i] = b[i]; }
  double rr = dot(n, r, r);
  double tol2 = reltol * reltol * rr;
  if (abstol * abstol > tol2) tol2 = abstol * abstol;
  int it = 0;
  while (rr > tol2 && it < maxiter) {
    ++it;
    spmv(n, rp, ci, v, p, q);
    const double alpha = rr / dot(n, p, q);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * q[i]; }
    const double rn = dot(n, r, r);
    const double beta = rn / rr;
    rr = rn;
    if (rr <= tol2) break;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];
  }
  if (resnorm_out) *resnorm_out = sqrt(rr);
  free(r); free(p); free(q);
  return it;
}
