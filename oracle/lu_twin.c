/* Partial-pivot LU on the CPU: the twin of the GPU algorithm.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * The reference (linalg_solver/linalg.py:548-567) swaps rows only on an exact
 * zero; the GPU path pivots on max |a| as BASELINE.json's north_star asks.
 * Both reach the same RREF / solution / inverse in exact arithmetic (the RREF
 * is unique), so this file is not a restatement of reference code but the
 * scalar model of the device algorithm: unblocked right-looking elimination,
 * row-major storage, pivot = first row attaining max |a| in the column,
 * unit-lower L stored below the diagonal.  Tests use it to check factors,
 * pivot vectors and solutions at sizes where the Python-speed reference
 * restatement would take hours.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* A: n x n row-major, lda >= n, overwritten by L\U.  ipiv[k] = 0-based row
 * exchanged with row k at step k (LAPACK convention, 0-based).
 * Returns info: 0 ok, k+1 if U[k][k] == 0 exactly (first such k). */
int orc_getrf_f64(double *A, int n, int lda, int32_t *ipiv)
{
    int info = 0;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(A[(size_t)k * lda + k]);
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(A[(size_t)i * lda + k]);
            if (v > best) { best = v; p = i; }     /* strict: lowest index wins ties */
        }
        ipiv[k] = p;
        if (p != k) {
            double *rk = A + (size_t)k * lda, *rp = A + (size_t)p * lda;
            memcpy(tmp, rk, sizeof(double) * (size_t)n);
            memcpy(rk, rp, sizeof(double) * (size_t)n);
            memcpy(rp, tmp, sizeof(double) * (size_t)n);
        }
        const double *prow = A + (size_t)k * lda;
        double piv = prow[k];
        if (piv == 0.0) { if (!info) info = k + 1; continue; }
        for (int i = k + 1; i < n; ++i) {
            double *ri = A + (size_t)i * lda;
            double l = ri[k] / piv;
            ri[k] = l;
            if (l != 0.0)
                for (int j = k + 1; j < n; ++j) ri[j] -= l * prow[j];
        }
    }
    free(tmp);
    return info;
}

/* Solve A X = B with the factors above.  B: n x nrhs row-major, in place. */
void orc_getrs_f64(const double *LU, int n, int lda, const int32_t *ipiv,
                   double *B, int nrhs, int ldb)
{
    for (int k = 0; k < n; ++k) {                /* apply P */
        int p = ipiv[k];
        if (p != k)
            for (int j = 0; j < nrhs; ++j) {
                double t = B[(size_t)k * ldb + j];
                B[(size_t)k * ldb + j] = B[(size_t)p * ldb + j];
                B[(size_t)p * ldb + j] = t;
            }
    }
    for (int i = 1; i < n; ++i) {                /* L y = Pb, unit diagonal */
        const double *li = LU + (size_t)i * lda;
        double *bi = B + (size_t)i * ldb;
        for (int k = 0; k < i; ++k) {
            double l = li[k];
            if (l != 0.0) {
                const double *bk = B + (size_t)k * ldb;
                for (int j = 0; j < nrhs; ++j) bi[j] -= l * bk[j];
            }
        }
    }
    for (int i = n - 1; i >= 0; --i) {           /* U x = y */
        const double *ui = LU + (size_t)i * lda;
        double *bi = B + (size_t)i * ldb;
        for (int k = i + 1; k < n; ++k) {
            double u = ui[k];
            if (u != 0.0) {
                const double *bk = B + (size_t)k * ldb;
                for (int j = 0; j < nrhs; ++j) bi[j] -= u * bk[j];
            }
        }
        double d = ui[i];
        for (int j = 0; j < nrhs; ++j) bi[j] /= d;
    }
}

/* sign in {-1,0,+1} and log|det| from the factors: det = sign * exp(logabs). */
void orc_slogdet_f64(const double *LU, int n, int lda, const int32_t *ipiv,
                     double *sign, double *logabs)
{
    double s = 1.0, l = 0.0;
    for (int k = 0; k < n; ++k) {
        double d = LU[(size_t)k * lda + k];
        if (ipiv[k] != k) s = -s;
        if (d == 0.0) { *sign = 0.0; *logabs = -INFINITY; return; }
        if (d < 0) { s = -s; d = -d; }
        l += log(d);
    }
    *sign = s;
    *logabs = l;
}
