/* fp64 restatement of the reference's Matrix.row_reduce in plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): checker and timed CPU
 * baseline; never linked into the product library.
 *
 * Follows /root/reference/linalg_solver/linalg.py:534-630 operation for
 * operation: first-non-zero pivot (not the largest), exact ==0 / ==1 tests,
 * division of the pivot row from the pivot column rightwards, `a - f*b` with a
 * separately rounded product.  Build with -ffp-contract=off so the compiler
 * never fuses that product into an FMA; CPython rounds twice.
 *
 * Parity status: pinned by tests/golden (outputs of the reference itself).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* step record kinds, mirrored in oracle/capi.py */
enum { ORC_SWAP = 0, ORC_NORM = 1, ORC_BELOW = 2, ORC_ABOVE = 3 };

typedef struct {
    int32_t kind; /* ORC_* */
    int32_t a;    /* 1-based row (swap: first row; norm: row) or 1-based column */
    int32_t b;    /* swap: second 1-based row, else 0 */
} orc_step;

/* A: m x n row-major with leading dimension lda, reduced in place.
 * bar_col <= 0 means n-1 (linalg.py:543, `bar_col or n - 1`).
 * pivots: capacity min(m,n) pairs (row, col), 0-based (linalg.py:607).
 * steps: optional (NULL to skip), capacity step_cap; *nsteps counts all steps
 * even when they do not fit.  Returns the number of pivots (= rank as the
 * reference sees it). */
int orc_row_reduce_f64(double *A, int m, int n, int lda, int bar_col,
                       int32_t *pivots, orc_step *steps, int step_cap, int *nsteps)
{
    int bar = bar_col > 0 ? bar_col : n - 1;
    int pi = 0, pj = 0, np = 0, ns = 0;
#define REC(k_, a_, b_)                                                         \
    do {                                                                        \
        if (steps && ns < step_cap) {                                           \
            steps[ns].kind = (k_); steps[ns].a = (a_); steps[ns].b = (b_);      \
        }                                                                       \
        ++ns;                                                                   \
    } while (0)

    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    while (pi < m && pj < bar) {                       /* :547 */
        double *prow = A + (size_t)pi * lda;
        if (prow[pj] == 0.0) {                         /* :548 */
            int hit = -1;
            for (int i = pi + 1; i < m; ++i)           /* :550 */
                if (A[(size_t)i * lda + pj] != 0.0) { hit = i; break; }
            if (hit < 0) { ++pj; continue; }           /* :565-567 */
            double *other = A + (size_t)hit * lda;     /* :552 whole-row swap */
            memcpy(tmp, prow, sizeof(double) * (size_t)n);
            memcpy(prow, other, sizeof(double) * (size_t)n);
            memcpy(other, tmp, sizeof(double) * (size_t)n);
            REC(ORC_SWAP, pi + 1, hit + 1);
        }
        double piv = prow[pj];
        int changed = 0;
        if (piv != 1.0) {                              /* :571 */
            for (int j = pj; j < n; ++j) {             /* :572 */
                double old = prow[j];
                double q = old / piv;
                prow[j] = q;
                changed |= (q != old);
            }
        }
        if (changed) REC(ORC_NORM, pi + 1, 0);         /* :576 */
        int touched = 0;
        changed = 0;
        for (int k = pi + 1; k < m; ++k) {             /* :587 */
            double *rk = A + (size_t)k * lda;
            double f = rk[pj];
            if (f == 0.0) continue;                    /* :589 */
            touched = 1;
            for (int j = pj; j < n; ++j) {             /* :593 */
                double old = rk[j];
                double prod = f * prow[j];             /* rounded product */
                double v = old - prod;                 /* then rounded difference */
                rk[j] = v;
                changed |= (v != old);
            }
        }
        if (touched && changed) REC(ORC_BELOW, pj + 1, 0); /* :597 */
        pivots[2 * np] = pi;                           /* :607 */
        pivots[2 * np + 1] = pj;
        ++np; ++pi; ++pj;
    }
    for (int idx = np - 1; idx >= 0; --idx) {          /* :611 */
        int r = pivots[2 * idx], c = pivots[2 * idx + 1];
        const double *prow = A + (size_t)r * lda;
        int changed = 0;
        for (int k = 0; k < r; ++k) {                  /* :614 */
            double *rk = A + (size_t)k * lda;
            double f = rk[c];
            if (f == 0.0) continue;
            for (int j = c; j < n; ++j) {              /* :618 */
                double old = rk[j];
                double prod = f * prow[j];
                double v = old - prod;
                rk[j] = v;
                changed |= (v != old);
            }
        }
        if (changed) REC(ORC_ABOVE, c + 1, 0);         /* :622 */
    }
    free(tmp);
    if (nsteps) *nsteps = ns;
#undef REC
    return np;
}

/* linalg.py:913-934.  reduced is m x (nvars+1...) row-major; returns 1-based
 * index of the first inconsistent row, 0 if consistent. */
int orc_inconsistent_row_f64(const double *R, int m, int lda, int nvars, int bar_col)
{
    for (int i = 0; i < m; ++i) {
        const double *row = R + (size_t)i * lda;
        int allzero = 1;
        for (int j = 0; j < nvars; ++j)
            if (!(row[j] == 0.0)) { allzero = 0; break; }
        if (allzero && row[bar_col] != 0.0) return i + 1;
    }
    return 0;
}

/* linalg.py:725-731: left n x n block of an n x 2n reduced matrix within 1e-12
 * of the identity.  (NaN compares false and therefore passes, as in Python.) */
int orc_left_is_identity_f64(const double *R, int n, int lda)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double x = R[(size_t)i * lda + j];
            double d = (i == j) ? x - 1.0 : x;
            if (d < 0) d = -d;
            if (d > 1e-12) return 0;
        }
    return 1;
}
