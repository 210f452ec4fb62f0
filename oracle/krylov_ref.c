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

/* The iteration pg_krylov.hip runs where its Gershgorin test admits it: BiCGStab right-preconditioned with the Neumann
 * polynomial M^-1 = 2I - A (u = 2p - Ap; v = Au; ... x += alpha u + omega u_s).  Not something the reference does --
 * here so that bench.py can time the SAME algorithm on the host cores next to the plain iteration. */
int krylov_ref_bicgstab_neumann(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, const double* b, double* x,
                                double reltol, double abstol, int maxiter, int nthreads, double* resnorm_out) {
#ifdef _OPENMP
  omp_set_num_threads(nthreads > 0 ? nthreads : 1);
#endif
  double* r = (double*)malloc(sizeof(double) * n);
  double* rh = (double*)malloc(sizeof(double) * n);
  double* p = (double*)calloc(n, sizeof(double));
  double* vv = (double*)calloc(n, sizeof(double));
  double* t = (double*)malloc(sizeof(double) * n);
  double* u = (double*)malloc(sizeof(double) * n);
  double* us = (double*)malloc(sizeof(double) * n);
  double* w = (double*)malloc(sizeof(double) * n);
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
    spmv(n, rp, ci, v, p, w);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) u[i] = 2.0 * p[i] - w[i];
    spmv(n, rp, ci, v, u, vv);
    const double den = dot(n, rh, vv);
    const int force = den == 0.0;
    alpha = force ? 0.0 : rho / den;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) r[i] -= alpha * vv[i]; /* r holds s */
    spmv(n, rp, ci, v, r, w);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) us[i] = 2.0 * r[i] - w[i];
    spmv(n, rp, ci, v, us, t);
    const double tt = dot(n, t, t);
    omega = tt != 0.0 ? dot(n, t, r) / tt : 0.0;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      x[i] += alpha * u[i] + omega * us[i];
      r[i] -= omega * t[i];
    }
    rho_old = rho;
    rho = dot(n, rh, r);
    rr = dot(n, r, r);
    if (rr <= tol2) break;
    if (omega == 0.0 || force || rho * rho < 1e-20 * rhat2 * rr) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) rh[i] = r[i];
      rho = rhat2 = rr;
      alpha = omega = 1.0;
      restart = 1;
    }
  }
  if (resnorm_out) *resnorm_out = sqrt(rr);
  free(r); free(rh); free(p); free(vv); free(t); free(u); free(us); free(w);
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
