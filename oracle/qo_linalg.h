/*
 * qo_linalg.h -- tiny dense row-major helpers for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it.
 *
 * The reference does its small dense algebra with Eigen (not available in this
 * image); these are plain loops with the same mathematical meaning.
 */
#ifndef QO_LINALG_H_
#define QO_LINALG_H_

#include <math.h>
#include <string.h>

/* C(r x c) = A(r x k) * B(k x c), all row-major with leading dims lda.. */
static inline void qo_mm(int r, int k, int c, const double* A, int lda, const double* B, int ldb,
                         double* C, int ldc) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int t = 0; t < k; ++t) s += A[i * lda + t] * B[t * ldb + j];
      C[i * ldc + j] = s;
    }
}
/* C(r x c) = A(k x r)^T * B(k x c) */
static inline void qo_mtm(int r, int k, int c, const double* A, int lda, const double* B, int ldb,
                          double* C, int ldc) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int t = 0; t < k; ++t) s += A[t * lda + i] * B[t * ldb + j];
      C[i * ldc + j] = s;
    }
}
/* y(r) = A(r x c) x */
static inline void qo_mv(int r, int c, const double* A, int lda, const double* x, double* y) {
  for (int i = 0; i < r; ++i) {
    double s = 0.0;
    for (int j = 0; j < c; ++j) s += A[i * lda + j] * x[j];
    y[i] = s;
  }
}
/* y(c) = A(r x c)^T x */
static inline void qo_mtv(int r, int c, const double* A, int lda, const double* x, double* y) {
  for (int j = 0; j < c; ++j) {
    double s = 0.0;
    for (int i = 0; i < r; ++i) s += A[i * lda + j] * x[i];
    y[j] = s;
  }
}
static inline double qo_dot(int n, const double* a, const double* b) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}
/* In-place lower Cholesky of the n x n SPD matrix A (row-major, lda).
 * Returns 0 on success, 1 + pivot index when a pivot is <= 0 / non-finite. */
static inline int qo_chol(int n, double* A, int lda) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * lda + j];
    for (int t = 0; t < j; ++t) d -= A[j * lda + t] * A[j * lda + t];
    if (!(d > 0.0) || !isfinite(d)) return 1 + j;
    d = sqrt(d);
    A[j * lda + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * lda + j];
      for (int t = 0; t < j; ++t) s -= A[i * lda + t] * A[j * lda + t];
      A[i * lda + j] = s / d;
    }
  }
  return 0;
}
/* Solve L L^T X = B in place for nrhs columns of B (n x nrhs, row-major ldb). */
static inline void qo_chol_solve(int n, const double* L, int lda, double* B, int ldb, int nrhs) {
  for (int c = 0; c < nrhs; ++c) {
    for (int i = 0; i < n; ++i) {
      double s = B[i * ldb + c];
      for (int t = 0; t < i; ++t) s -= L[i * lda + t] * B[t * ldb + c];
      B[i * ldb + c] = s / L[i * lda + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = B[i * ldb + c];
      for (int t = i + 1; t < n; ++t) s -= L[t * lda + i] * B[t * ldb + c];
      B[i * ldb + c] = s / L[i * lda + i];
    }
  }
}

#endif /* QO_LINALG_H_ */
