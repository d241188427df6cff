// Panel factorisation: partial-pivot LU of a tall m x jb block (row-major).
//
// Replaces, for jb consecutive pivots, the reference's pivot search + row swap +
// normalise + eliminate-below sequence (linalg_solver/linalg.py:548-596).  Two
// deliberate differences, both result-preserving in exact arithmetic (the RREF
// is unique): the pivot is the entry of largest magnitude (the reference takes
// the first non-zero one, linalg.py:550-552), and the multipliers are kept
// (unit-lower L) instead of normalising the pivot row (linalg.py:569-575).
//
// mode 0 ("per-column launches", this file's first half): two launches per
// column -- `panel_pivot` (finish the arg-max, record the interchange, swap the
// two panel rows) and `panel_update` (scale the column, rank-1 update of the
// remaining panel columns, and the per-workgroup arg-max partials of the NEXT
// column in the same pass).  Simple and robust; launch-latency bound.
#include "common.h"

namespace lsx {

constexpr int PROWS = 32;  // rows per workgroup in panel_update

// wave-level arg-max on (|v|, lowest row wins ties)
template <typename T>
__device__ __forceinline__ void wave_argmax(T &v, int &i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T ov = __shfl_down(v, off, 64);
        const int oi = __shfl_down(i, off, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// Finish the arg-max of column j from the per-workgroup partials, record the
// interchange, swap rows j <-> p inside the panel.  One workgroup.
template <typename T>
__global__ __launch_bounds__(256) void panel_pivot_kernel(int m, int jb, T *__restrict__ P, int ldp,
                                                          int row0, int col_global0, int j,
                                                          const T *__restrict__ cand_val,
                                                          const int *__restrict__ cand_idx, int ncand,
                                                          int32_t *__restrict__ ipiv,
                                                          int *__restrict__ info) {
    __shared__ T s_v[4];
    __shared__ int s_i[4];
    __shared__ int s_p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T v = T(-1);
    int i = 0x7fffffff;
    for (int c = tid; c < ncand; c += 256) {
        const T cv = cand_val[c];
        const int ci = cand_idx[c];
        if (cv > v || (cv == v && ci < i)) { v = cv; i = ci; }
    }
    wave_argmax(v, i);
    if (lane == 0) { s_v[wave] = v; s_i[wave] = i; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_v[w] > v || (s_v[w] == v && s_i[w] < i)) { v = s_v[w]; i = s_i[w]; }
        if (i < j || i >= m) i = j;  // no candidate (m == j+1 edge) -> keep the diagonal
        s_p = i;
        ipiv[j] = row0 + i;
        if (v == T(0) && info && *info == 0) *info = col_global0 + j + 1;
    }
    __syncthreads();
    const int p = s_p;
    if (p != j)
        for (int c = tid; c < jb; c += 256) {
            const T a = P[(size_t)j * ldp + c], b = P[(size_t)p * ldp + c];
            P[(size_t)j * ldp + c] = b;
            P[(size_t)p * ldp + c] = a;
        }
}

// Rows (j, m): l = a[i][j] / pivot stored in place, a[i][c] -= l * a[j][c] for
// the remaining panel columns, and the arg-max partial of column j+1 over this
// workgroup's rows.  j = -1 only computes the partials of column 0.
// One wave per row at a time: a row's jb entries are contiguous (coalesced).
template <typename T>
__global__ __launch_bounds__(256) void panel_update_kernel(int m, int jb, T *__restrict__ P, int ldp,
                                                           int j, T *__restrict__ cand_val,
                                                           int *__restrict__ cand_idx) {
    __shared__ T s_v[4];
    __shared__ int s_i[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = (j + 1) + blockIdx.x * PROWS;
    const int nc = j + 1;  // column whose arg-max we prepare
    T piv = T(1), rinv = T(0);
    if (j >= 0) {
        piv = P[(size_t)j * ldp + j];
        rinv = (piv != T(0)) ? fast_recip<T>(piv) : T(0);
    }
    T best = T(-1);
    int besti = 0x7fffffff;
    for (int rr = wave; rr < PROWS; rr += 4) {
        const int i = r0 + rr;
        if (i >= m) break;
        T *row = P + (size_t)i * ldp;
        T l = T(0);
        if (j >= 0) {
            l = row[j] * rinv;
            if (piv == T(0)) l = row[j];  // singular column: leave entries untouched (LAPACK)
        }
        for (int c0 = 0; c0 < jb; c0 += 64) {
            const int c = c0 + lane;
            if (c >= jb) break;
            T v = row[c];
            if (j >= 0 && piv != T(0)) {
                if (c == j) { v = l; row[c] = v; }
                else if (c > j) { v -= l * P[(size_t)j * ldp + c]; row[c] = v; }
            }
            if (c == nc && nc < jb) {
                const T av = v < 0 ? -v : v;
                if (av > best || (av == best && i < besti)) { best = av; besti = i; }
            }
        }
    }
    // the lane owning column nc carries this wave's candidate
    wave_argmax(best, besti);
    if (lane == 0) { s_v[wave] = best; s_i[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_v[w] > best || (s_v[w] == best && s_i[w] < besti)) { best = s_v[w]; besti = s_i[w]; }
        cand_val[blockIdx.x] = best;
        cand_idx[blockIdx.x] = besti;
    }
}

template <typename T>
int panel_percolumn(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0,
                    int32_t *d_ipiv, int *d_info) {
    // scratch layout: cand_val[ngrid] | cand_idx[ngrid]
    const int ngrid_max = (m + PROWS - 1) / PROWS + 1;
    T *cand_val = (T *)h->scratch;
    int *cand_idx = (int *)((char *)h->scratch + sizeof(double) * ngrid_max);
    if (sizeof(double) * ngrid_max + sizeof(int) * ngrid_max > h->scratch_bytes) {
        set_error("panel: scratch too small for m=%d", m);
        return LSX_ERR_INTERNAL;
    }
    int ncand = (m + PROWS - 1) / PROWS;
    hipLaunchKernelGGL(panel_update_kernel<T>, dim3(ncand), dim3(256), 0, h->stream, m, jb, P, ldp, -1,
                       cand_val, cand_idx);
    for (int j = 0; j < jb; ++j) {
        hipLaunchKernelGGL(panel_pivot_kernel<T>, dim3(1), dim3(256), 0, h->stream, m, jb, P, ldp, row0,
                           col0, j, cand_val, cand_idx, ncand, d_ipiv, d_info);
        const int rem = m - (j + 1);
        ncand = (rem + PROWS - 1) / PROWS;
        if (rem > 0)
            hipLaunchKernelGGL(panel_update_kernel<T>, dim3(ncand), dim3(256), 0, h->stream, m, jb, P,
                               ldp, j, cand_val, cand_idx);
    }
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

template <typename T>
int panel_cooperative(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0,
                      int32_t *d_ipiv, int *d_info);
template <typename T>
int panel_blocked(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv,
                  int *d_info);

template <typename T>
int panel_pipelined(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv,
                    int *d_info);

template <typename T>
int panel_xcd(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info);

template <typename T>
int panel_col(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info);

template <typename T>
int launch_panel(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int32_t *d_ipiv,
                 int *d_info) {
    if (m <= 0 || jb <= 0) return LSX_OK;
    h->moves_valid = false;
    ProfScope ps(h, LSX_PROF_PANEL, 0, 2.0 * sizeof(T) * m * (double)jb);
    // the panel's first column is global column row0 (square LU: panel starts on the diagonal)
    if (h->panel_mode == 4) {   // one XCD; taller panels than it holds take the device-scope kernel
        if (h->panel_col && !h->panel_debug) {   // columns distributed over the workgroups (kernels_panel_c.hip)
            const int rc = panel_col<T>(h, m, jb, P, ldp, row0, row0, d_ipiv, d_info);
            if (rc != 1) return rc;
        }
        const int r = panel_xcd<T>(h, m, jb, P, ldp, row0, row0, d_ipiv, d_info);   // rows distributed, pivot exchange in the L2
        if (r != 1) return r;
    }
    if (h->panel_mode >= 3) {
        const int r = panel_pipelined<T>(h, m, jb, P, ldp, row0, row0, d_ipiv, d_info);
        if (r != 1) return r;  // 1 = shape not supported: try the older cooperative kernel
    }
#ifdef LSX_DIAG_PANELS   // make DIAG=1: the superseded cooperative / blocked kernels as cross-checks
    if (h->panel_mode == 2) {
        const int r = panel_blocked<T>(h, m, jb, P, ldp, row0, row0, d_ipiv, d_info);
        if (r != 1) return r;  // 1 = shape not supported: try the unblocked cooperative kernel
    }
    if (h->panel_mode >= 1) {
        const int r = panel_cooperative<T>(h, m, jb, P, ldp, row0, row0, d_ipiv, d_info);
        if (r != 1) return r;  // 1 = shape not supported by the cooperative kernel
    }
#endif
    return panel_percolumn<T>(h, m, jb, P, ldp, row0, row0, d_ipiv, d_info);
}

template int launch_panel<double>(lsx_handle_t, int, int, double *, int, int, int32_t *, int *);
template int launch_panel<float>(lsx_handle_t, int, int, float *, int, int, int32_t *, int *);

}  // namespace lsx
