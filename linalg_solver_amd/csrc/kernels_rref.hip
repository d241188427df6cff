// General m x n reduced row echelon form (rank-revealing Gauss-Jordan).
//
// Covers what the blocked LU cannot: rectangular, rank-deficient and
// `bar_col`-split inputs of Matrix.row_reduce (linalg_solver/linalg.py:534-630),
// including the zero-column skip (linalg.py:565-567: pivot row stays, column
// advances) and the carried-along columns right of bar_col.
//
// Pivot rule (selectable):
//   - LSX_PIVOT_FIRST (default of the Python surface): the first row at or
//     below the pivot row whose entry is non-zero, exactly the reference's rule
//     (linalg.py:548-552).  This matters beyond style: with rank < m or
//     bar_col < n the carried-along columns depend on WHICH rows become pivot
//     rows, so only this rule reproduces the reference's numbers there.
//   - LSX_PIVOT_MAX: largest |a| (partial pivoting); same pivot positions and
//     same left block, better conditioned on large inputs.
// Difference from the reference, by design:
//   - "non-zero" means |a| > tol; a column with no such entry counts as a zero
//     column and its sub-pivot entries are set to exactly 0.  In exact
//     arithmetic this is the reference's test; in floating point it makes rank
//     the mathematical rank instead of a rounding artefact.
//   - elimination above and below the pivot happens in the same sweep
//     (Gauss-Jordan) instead of a separate backward pass (linalg.py:611-629).
//
// Two launches per column, state kept on the device (no host round trip):
//   rref_pivot  (1 workgroup)  advance state, arg-max, tolerance test, save the
//               normalised pivot row and the displaced row
//   rref_sweep  (many)         every other row  r -= r[pj] * pivot_row
// HBM-bound: each sweep reads and writes the live part of the matrix once.
#include "common.h"

namespace lsx {

struct RrefState {
    int pi;       // next pivot row
    int rank;
    int p;        // row chosen in the current column
    int pending;  // 1: a pivot was placed in the previous column, pi/rank not yet advanced
    int skip;     // 1: current column has no pivot
    int pad[3];
    double tol;        // fixed tolerance (user) or < 0: eps_scale * amax, re-evaluated per column
    double amax;       // running max |entry| of the working matrix (grows with elimination)
    double eps_scale;  // eps * max(m, n)
};

template <typename T>
__global__ __launch_bounds__(256) void rref_amax_kernel(int m, int ncols, const T *__restrict__ R,
                                                        int ldr, RrefState *st) {
    __shared__ double s[256];
    double v = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)m * ncols;
         e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / ncols), j = (int)(e % ncols);
        const double a = fabs((double)R[(size_t)i * ldr + j]);
        if (a > v) v = a;  // NaN never wins
    }
    s[threadIdx.x] = v;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        atomicMax((unsigned long long *)&st->amax, (unsigned long long)__double_as_longlong(s[0]));
}

__global__ void rref_init_kernel(RrefState *st, double tol, double eps_scale) {
    st->pi = 0; st->rank = 0; st->p = 0; st->pending = 0; st->skip = 1;
    st->tol = tol;
    st->eps_scale = eps_scale;
}

template <typename T>
__global__ __launch_bounds__(256) void rref_pivot_kernel(int m, int n, int pj, int pivot_rule,
                                                         T *__restrict__ R, int ldr, RrefState *st,
                                                         int32_t *__restrict__ pivots,
                                                         T *__restrict__ prow, T *__restrict__ orow) {
    __shared__ double s_v[256];
    __shared__ int s_i[256];
    const int tid = threadIdx.x;
    if (tid == 0 && st->pending) { st->pi += 1; st->rank += 1; st->pending = 0; }
    __syncthreads();
    const int pi = st->pi;
    if (pi >= m) {
        if (tid == 0) st->skip = 1;
        return;
    }
    // rounding noise scales with the largest magnitude the elimination has produced so far
    const double tol = st->tol >= 0 ? st->tol : st->eps_scale * st->amax;
    double v = -1;
    int vi = 0x7fffffff;
    for (int i = pi + tid; i < m; i += 256) {
        double a = fabs((double)R[(size_t)i * ldr + pj]);
        // FIRST rule: every non-zero entry scores the same, so the lowest row index wins below
        if (pivot_rule == LSX_PIVOT_FIRST) a = (a > tol) ? 1.0e300 : 0.0;
        if (a > v) { v = a; vi = i; }
    }
    s_v[tid] = v; s_i[tid] = vi;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) {
            const double ov = s_v[tid + k];
            const int oi = s_i[tid + k];
            if (ov > s_v[tid] || (ov == s_v[tid] && oi < s_i[tid])) { s_v[tid] = ov; s_i[tid] = oi; }
        }
        __syncthreads();
    }
    const double best = s_v[0];
    const int p = s_i[0];
    if (!(best > tol)) {  // zero column (or NaN): clear the sub-pivot entries, move on
        for (int i = pi + tid; i < m; i += 256) R[(size_t)i * ldr + pj] = T(0);
        if (tid == 0) st->skip = 1;
        return;
    }
    const T piv = R[(size_t)p * ldr + pj];
    for (int c = pj + tid; c < n; c += 256) {
        prow[c] = (c == pj) ? T(1) : R[(size_t)p * ldr + c] / piv;  // linalg.py:572-575
        orow[c] = R[(size_t)pi * ldr + c];
    }
    if (tid == 0) {
        st->p = p; st->skip = 0; st->pending = 1;
        pivots[2 * st->rank] = pi;      // linalg.py:607
        pivots[2 * st->rank + 1] = pj;
    }
}

// One wave per row.  Row pi receives the normalised pivot row; the row that
// held the pivot (p) receives the displaced row, eliminated; every other row
// with a non-zero entry in column pj is eliminated in place.
template <typename T>
__global__ __launch_bounds__(256) void rref_sweep_kernel(int m, int n, int pj, T *__restrict__ R,
                                                         int ldr, RrefState *st,
                                                         const T *__restrict__ prow,
                                                         const T *__restrict__ orow) {
    if (st->skip) return;
    const int pi = st->pi, p = st->p;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    T *row = R + (size_t)i * ldr;
    if (i == pi) {
        for (int c = pj + lane; c < n; c += 64) row[c] = prow[c];
        return;
    }
    const T *src = (i == p) ? orow : row;
    const T f = src[pj];
    if (f == T(0) && i != p) return;  // linalg.py:589
    double vmax = 0;
    for (int c = pj + lane; c < n; c += 64) {
        const T v = src[c] - f * prow[c];
        row[c] = v;
        vmax = fmax(vmax, fabs((double)v));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmax(vmax, __shfl_down(vmax, off, 64));
    if (lane == 0 && vmax > st->amax)
        atomicMax((unsigned long long *)&st->amax, (unsigned long long)__double_as_longlong(vmax));
}

__global__ void rref_finish_kernel(RrefState *st, int *d_rank) {
    if (st->pending) { st->pi += 1; st->rank += 1; st->pending = 0; }
    if (d_rank) *d_rank = st->rank;
}

// blocked form for large inputs with the largest-magnitude rule (kernels_rref_blk.hip); 1 = not applicable
template <typename T>
int rref_blocked(lsx_handle_t h, int m, int n, int bar, T *W, int ldw, int32_t *d_pivots, int *d_rank, double tol,
                 int pivot_rule);

template <typename T>
int launch_rref(lsx_handle_t h, int m, int n, int bar, T *R, int ldr, int32_t *d_pivots, int *d_rank,
                double tol, int pivot_rule) {
    if (h->rref_blocked) {
        const int rb = rref_blocked<T>(h, m, n, bar, R, ldr, d_pivots, d_rank, tol, pivot_rule);
        if (rb != 1) return rb;
    }
    ProfScope ps(h, LSX_PROF_OTHER);
    // scratch: state | prow[n] | orow[n]
    const size_t need = 256 + 2 * (size_t)n * sizeof(T);
    if (need > h->scratch_bytes) {
        set_error("rref: scratch too small (need %zu)", need);
        return LSX_ERR_INTERNAL;
    }
    RrefState *st = (RrefState *)h->scratch;
    T *prow = (T *)((char *)h->scratch + 256);
    T *orow = prow + n;
    LSX_HIP(hipMemsetAsync(st, 0, sizeof(RrefState), h->stream));
    if (tol < 0 && bar > 0)
        hipLaunchKernelGGL(rref_amax_kernel<T>, dim3(256), dim3(256), 0, h->stream, m, bar, R, ldr, st);
    // 32 eps max(m, n): the residue of a dependent column after k elimination steps is ~k eps times the running
    // maximum with a tail; a factor of 1 put exactly-rank-517 integer products of order 1000 at rank 518
    const double eps_scale = 32.0 * (double)Real<T>::eps * (double)(m > n ? m : n);
    hipLaunchKernelGGL(rref_init_kernel, dim3(1), dim3(1), 0, h->stream, st, tol, eps_scale);
    for (int pj = 0; pj < bar; ++pj) {
        hipLaunchKernelGGL(rref_pivot_kernel<T>, dim3(1), dim3(256), 0, h->stream, m, n, pj, pivot_rule,
                           R, ldr, st, d_pivots, prow, orow);
        hipLaunchKernelGGL(rref_sweep_kernel<T>, dim3((m + 3) / 4), dim3(256), 0, h->stream, m, n, pj,
                           R, ldr, st, prow, orow);
    }
    hipLaunchKernelGGL(rref_finish_kernel, dim3(1), dim3(1), 0, h->stream, st, d_rank);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

template int launch_rref<double>(lsx_handle_t, int, int, int, double *, int, int32_t *, int *, double,
                                 int);
template int launch_rref<float>(lsx_handle_t, int, int, int, float *, int, int32_t *, int *, double,
                                int);

}  // namespace lsx
