// Blocked rank-revealing reduced row echelon form (SURVEY 8f item 1 at scale).
//
// Same result as kernels_rref.hip with LSX_PIVOT_MAX -- Matrix.row_reduce's column-skip semantics
// (linalg_solver/linalg.py:547-567: a column without a usable pivot is skipped, the pivot row stays) with the
// largest-magnitude pivot and a tolerance -- but organised like the blocked LU so that the O(m n r) work runs on the
// MFMA tile instead of as one HBM sweep of the whole matrix per column:
//   phase 1  echelon form, 128 columns at a time.  Inside a block: per column an arg-max over the rows below the
//            current pivot row, the tolerance test (skip: clear the column's sub-pivot entries, the pivot row does
//            not advance), the row interchange and the elimination of the REST OF THE BLOCK only.  Then once per
//            block, for the q <= 128 pivots found: interchanges on the columns to the right, U12 = L11^-1 A12
//            (block solve) and A22 -= L21 U12 (trailing update, the same kernel as the LU's) with the multipliers
//            gathered from the q pivot columns, which are then cleared below their pivots.
//   phase 2  back substitution on the r pivot rows, 128 pivots at a time from the last to the first:
//            rows of the block <- (U_pp)^-1 rows (block solve, non-unit upper), rows above -= U_ap * rows (update).
//            Pivot columns come out as exact unit vectors, rows below the rank as exact zeros left of the bar.
// The number of pivots per block is data: the host reads it (4 bytes) after each block of phase 1.
// Tolerance: |a| <= tol is zero; tol < 0 selects eps * max(m, n) * (running max of the working matrix left of the
// bar, re-evaluated at every block: the unblocked kernel re-evaluates it at every column).
#include <algorithm>
#include <vector>

#include "common.h"

namespace lsx {

namespace {

constexpr int RB_W = 128;      // columns per block
constexpr int RRB_ROWS = 8;    // rows per workgroup of the per-column kernels (one arg-max candidate each): with 256 the
                               // update of an 8192-row block ran on 32 CUs and took 42 us per column, 77 % of the reduction

struct RrbState {
    int r;        // pivot row of the next pivot (global)
    int q;        // pivots found in the current block
    int skip;     // current column has no pivot
    int p;        // row chosen in the current column
    int adv;      // the last column got a pivot and q has not been advanced yet (done by the next rrb_pivot / rrb_advance)
    double tol;
    int pc[RB_W];   // pivot columns of the current block (global column index)
    int pr[RB_W];   // row each pivot was taken from (global): interchange t is rows (r0 + t) <-> pr[t]
};

template <typename T>
__device__ __forceinline__ void wave_argmax2(T &v, int &i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T ov = __shfl_down(v, off, 64);
        const int oi = __shfl_down(i, off, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rrb_amax_kernel(int m, int ncols, const T *__restrict__ R, int ldr, double *out) {
    __shared__ double s[256];
    double v = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)m * ncols; e += (size_t)gridDim.x * 256) {
        const double a = fabs((double)R[(e / ncols) * ldr + (e % ncols)]);
        if (a > v) v = a;
    }
    s[threadIdx.x] = v;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(s[0]));
}

__global__ void rrb_init_kernel(RrbState *st, double tol, double eps_scale, const double *amax) {
    st->r = 0; st->q = 0; st->skip = 1; st->p = 0; st->adv = 0;
    st->tol = tol >= 0 ? tol : eps_scale * amax[0];
}
// per block: the tolerance follows the running maximum of the working matrix (as the unblocked kernel's does per
// column): rounding noise in a "zero" column scales with the largest magnitude the elimination has produced
__global__ void rrb_block_begin_kernel(RrbState *st, double user_tol, double eps_scale, const double *amax) {
    st->q = 0;
    st->adv = 0;
    if (user_tol < 0) st->tol = eps_scale * amax[0];
}
__global__ void rrb_block_end_kernel(RrbState *st, int32_t *pivots, int *rank_out) {
    // pivots of this block: (row, column) pairs, rows are consecutive from st->r
    for (int t = threadIdx.x; t < st->q; t += blockDim.x) {
        pivots[2 * (st->r + t)] = st->r + t;
        pivots[2 * (st->r + t) + 1] = st->pc[t];
    }
    __syncthreads();
    if (threadIdx.x == 0) { st->r += st->q; *rank_out = st->r; }
}

// arg-max partials of column `col` over rows >= st->r + st->q (the first column of a block; later columns get
// theirs from rrb_update_kernel)
template <typename T>
__global__ __launch_bounds__(256) void rrb_cand_kernel(int m, const T *__restrict__ W, int ldw, int col,
                                                       const RrbState *st, T *cand_val, int *cand_idx) {
    __shared__ T s_v[4];
    __shared__ int s_i[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = st->r + st->q + blockIdx.x * RRB_ROWS;
    T v = T(-1);
    int i = 0x7fffffff;
    if (tid < RRB_ROWS && r0 + tid < m) { const T a = W[(size_t)(r0 + tid) * ldw + col]; v = a < 0 ? -a : a; if (!(v >= T(0))) v = T(-1); i = r0 + tid; }
    wave_argmax2(v, i);
    if (lane == 0) { s_v[wave] = v; s_i[wave] = i; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_v[w] > v || (s_v[w] == v && s_i[w] < i)) { v = s_v[w]; i = s_i[w]; }
        cand_val[blockIdx.x] = v;
        cand_idx[blockIdx.x] = i;
    }
}

// Column `col` (global) of the block [c0, c0 + w): finish the arg-max, tolerance test, record the pivot and swap
// the two rows inside the block's columns.  One workgroup.
template <typename T>
__global__ __launch_bounds__(256) void rrb_pivot_kernel(int m, T *__restrict__ W, int ldw, int c0, int w, int col,
                                                        RrbState *st, const T *__restrict__ cand_val,
                                                        const int *__restrict__ cand_idx, int ncand_max) {
    __shared__ T s_v[4];
    __shared__ int s_i[4];
    __shared__ int s_p;
    __shared__ int s_rc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {   // the previous column's pivot is counted here: one launch per column less (rrb_advance)
        if (st->adv) { st->q += 1; st->adv = 0; }
        s_rc = st->r + st->q;
    }
    __syncthreads();
    const int rc = s_rc;   // pivot row of this column if it gets one
    if (rc >= m) { if (tid == 0) st->skip = 1; return; }
    const int ncand = min(ncand_max, (m - rc + RRB_ROWS - 1) / RRB_ROWS);
    T v = T(-1);
    int i = 0x7fffffff;
    for (int c = tid; c < ncand; c += 256) {
        const T cv = cand_val[c];
        const int ci = cand_idx[c];
        if (cv > v || (cv == v && ci < i)) { v = cv; i = ci; }
    }
    wave_argmax2(v, i);
    if (lane == 0) { s_v[wave] = v; s_i[wave] = i; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; ++k)
            if (s_v[k] > v || (s_v[k] == v && s_i[k] < i)) { v = s_v[k]; i = s_i[k]; }
        const bool piv = (double)v > st->tol && i >= rc && i < m;
        s_p = piv ? i : -1;
        st->skip = piv ? 0 : 1;
        st->adv = piv ? 1 : 0;
        st->p = i;
        if (piv) { st->pc[st->q] = col; st->pr[st->q] = i; }
    }
    __syncthreads();
    const int p = s_p;
    if (p >= 0 && p != rc)
        for (int c = c0 + tid; c < c0 + w; c += 256) {
            const T a = W[(size_t)rc * ldw + c], b = W[(size_t)p * ldw + c];
            W[(size_t)rc * ldw + c] = b;
            W[(size_t)p * ldw + c] = a;
        }
}

// After rrb_pivot for column `col`.  Pivot: rows below the pivot row get l = a / pivot stored in place and are
// eliminated in the rest of the block; no pivot: the column's entries from the pivot row down are cleared.
// Either way the arg-max partials of the NEXT column (over the rows below the then-current pivot row) come out of
// the same pass.  One wave per row at a time; the block's columns of a row are contiguous.
template <typename T>
__global__ __launch_bounds__(256) void rrb_update_kernel(int m, T *__restrict__ W, int ldw, int c0, int w, int col,
                                                         const RrbState *st, T *cand_val, int *cand_idx) {
    __shared__ T s_v[4];
    __shared__ int s_i[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool skip = st->skip != 0;
    const int rc = st->r + st->q;              // this column's pivot row (if !skip)
    const int first = skip ? rc : rc + 1;      // rows that are candidates for the next column
    const int r0 = first + blockIdx.x * RRB_ROWS;   // rows per workgroup (cand index = blockIdx.x)
    const int nc = col + 1;
    T piv = T(1), rinv = T(0);
    if (!skip) {
        piv = W[(size_t)rc * ldw + col];
        rinv = fast_recip<T>(piv);
    }
    T best = T(-1);
    int besti = 0x7fffffff;
    for (int rr = wave; rr < RRB_ROWS; rr += 4) {
        const int i = r0 + rr;
        if (i >= m) break;
        T *row = W + (size_t)i * ldw;
        T l = T(0);
        if (!skip) l = row[col] * rinv;
        for (int cc = col - c0 + lane; cc < w; cc += 64) {   // columns col .. c0 + w - 1
            const int c = c0 + cc;
            T v = row[c];
            if (skip) {
                if (c == col) { v = T(0); row[c] = v; }
            } else if (c == col) {
                v = l; row[c] = v;
            } else {
                v -= l * W[(size_t)rc * ldw + c]; row[c] = v;
            }
            if (c == nc && nc < c0 + w) {
                T av = v < 0 ? -v : v;
                if (!(av >= T(0))) av = T(-1);   // NaN never wins
                if (av > best || (av == best && i < besti)) { best = av; besti = i; }
            }
        }
    }
    wave_argmax2(best, besti);
    if (lane == 0) { s_v[wave] = best; s_i[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; ++k)
            if (s_v[k] > best || (s_v[k] == best && s_i[k] < besti)) { best = s_v[k]; besti = s_i[k]; }
        cand_val[blockIdx.x] = best;
        cand_idx[blockIdx.x] = besti;
    }
}
// end of a block: count the last column's pivot
__global__ void rrb_advance_kernel(RrbState *st) { if (st->adv) { st->q += 1; st->adv = 0; } }

// the block's q interchanges, in order, on `ncols` columns starting at Wr (column-parallel)
template <typename T>
__global__ __launch_bounds__(256) void rrb_swap_kernel(int ncols, T *__restrict__ Wr, int ldw, const RrbState *st) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    const int r0 = st->r, q = st->q;
    for (int t = 0; t < q; ++t) {
        const int a = r0 + t, b = st->pr[t];
        if (a != b) {
            const T x = Wr[(size_t)a * ldw + c], y = Wr[(size_t)b * ldw + c];
            Wr[(size_t)a * ldw + c] = y;
            Wr[(size_t)b * ldw + c] = x;
        }
    }
}
// L (rows r .. m-1, q columns, leading dimension RB_W) <- the block's pivot columns: unit-lower in its first q rows
// (the multipliers), then the rows below; the multipliers are cleared in W (the echelon form has zeros there)
template <typename T>
__global__ __launch_bounds__(256) void rrb_gather_l_kernel(int m, T *__restrict__ W, int ldw, const RrbState *st,
                                                           T *__restrict__ L) {
    const int r0 = st->r, q = st->q;
    const int i = r0 + blockIdx.x * 2 + (threadIdx.x >> 7);
    const int t = threadIdx.x & 127;
    if (i >= m || t >= q) return;
    const int c = st->pc[t];
    const int rp = r0 + t;            // pivot row of pivot t
    T v = W[(size_t)i * ldw + c];
    if (i > rp) W[(size_t)i * ldw + c] = T(0);
    else if (i < rp) v = T(0);        // above the pivot inside the block's pivot rows: U entries, not multipliers
    else v = T(1);
    L[(size_t)(i - r0) * RB_W + t] = v;
}

// ---- phase 2
// G (rows 0 .. kb+jb-1, jb columns, leading dimension RB_W) <- columns pc[kb .. kb+jb) of the pivot rows
template <typename T>
__global__ __launch_bounds__(256) void rrb_gather_u_kernel(int nrows, int jb, const T *__restrict__ W, int ldw,
                                                           const int32_t *__restrict__ pivots, int kb, T *__restrict__ G) {
    const int i = blockIdx.x * 2 + (threadIdx.x >> 7);
    const int t = threadIdx.x & 127;
    if (i >= nrows || t >= jb) return;
    G[(size_t)i * RB_W + t] = W[(size_t)i * ldw + pivots[2 * (kb + t) + 1]];
}
// pivot columns -> exact unit vectors; rows >= rank -> exact zeros left of the bar
template <typename T>
__global__ __launch_bounds__(256) void rrb_finish_kernel(int m, int bar, T *__restrict__ W, int ldw,
                                                         const int32_t *__restrict__ pivots, int rank) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    for (int i = blockIdx.y; i < m; i += gridDim.y) {   // grid.y is capped at 65535 rows: the rest by stride
        if (i >= rank) { if (c < bar) W[(size_t)i * ldw + c] = T(0); continue; }
        if (c < rank) {   // thread c handles pivot column pc[c] of row i
            const int pcol = pivots[2 * c + 1];
            W[(size_t)i * ldw + pcol] = (c == i) ? T(1) : T(0);
        }
    }
}

}  // namespace

// Returns 1 when the shape / rule is not for this path (the caller takes kernels_rref.hip).
template <typename T>
int rref_blocked(lsx_handle_t h, int m, int n, int bar, T *W, int ldw, int32_t *d_pivots, int *d_rank, double tol,
                 int pivot_rule) {
    if (pivot_rule != LSX_PIVOT_MAX || (size_t)m * bar < (size_t)256 * 256 || bar < 1) return 1;
    ProfScope ps(h, LSX_PROF_OTHER);
    const int ncand = (m + RRB_ROWS - 1) / RRB_ROWS;
    // scratch: state | amax | cand_val | cand_idx ; workspace (ws3): L / G buffer m x 128, block inverses
    const size_t need = 2048 + (size_t)ncand * (sizeof(T) + sizeof(int)) + 256;
    if (need > h->scratch_bytes) { set_error("rref_blocked: scratch too small"); return LSX_ERR_INTERNAL; }
    RrbState *st = (RrbState *)h->scratch;
    double *amax = (double *)((char *)h->scratch + 1536);
    T *cand_val = (T *)((char *)h->scratch + 2048);
    int *cand_idx = (int *)((char *)h->scratch + 2048 + (((size_t)ncand * sizeof(T) + 15) & ~(size_t)15));
    const size_t lbytes = (((size_t)m * RB_W * sizeof(T)) + 255) & ~(size_t)255;
    const size_t tbytes = 2 * 4096 * sizeof(T);
    if (h->ws3_bytes < lbytes + tbytes) {
        LSX_HIP(hipStreamSynchronize(h->stream));
        if (h->ws3) (void)hipFree(h->ws3);
        h->ws3 = nullptr; h->ws3_bytes = 0;
        if (hipMalloc(&h->ws3, lbytes + tbytes) != hipSuccess) { set_error("rref_blocked: hipMalloc"); return LSX_ERR_ALLOC; }
        h->ws3_bytes = lbytes + tbytes;
    }
    T *L = (T *)h->ws3;
    T *Tinv = (T *)((char *)h->ws3 + lbytes);
    struct MfmaOnly { lsx_handle_t h; bool keep; ~MfmaOnly() { h->gemm_mfma_only = keep; } } mo{h, h->gemm_mfma_only};
    h->gemm_mfma_only = true;
    hipStream_t s = h->stream;
    LSX_HIP(hipMemsetAsync(h->scratch, 0, 2048, s));
    if (tol < 0) hipLaunchKernelGGL(rrb_amax_kernel<T>, dim3(256), dim3(256), 0, s, m, bar, W, ldw, amax);
    // 32 eps max(m, n): the residue of a dependent column after k elimination steps is ~k eps times the running
    // maximum with a tail; a factor of 1 put exactly-rank-517 integer products of order 1000 at rank 518
    const double eps_scale = 32.0 * (double)Real<T>::eps * (double)(m > n ? m : n);
    hipLaunchKernelGGL(rrb_init_kernel, dim3(1), dim3(1), 0, s, st, tol, eps_scale, amax);
    // ---------------- phase 1
    int r = 0;   // host copy of the pivot row
    for (int c0 = 0; c0 < bar && r < m; c0 += RB_W) {
        const int w = std::min(RB_W, bar - c0);
        const int rows_left = m - r;
        const int gc = (rows_left + RRB_ROWS - 1) / RRB_ROWS;
        if (tol < 0 && c0 > 0 && n - c0 > 0)   // running max over the live part (atomicMax: it never decreases)
            hipLaunchKernelGGL(rrb_amax_kernel<T>, dim3(256), dim3(256), 0, s, m - r, std::min(bar, n) - c0, W + (size_t)r * ldw + c0, ldw, amax);
        hipLaunchKernelGGL(rrb_block_begin_kernel, dim3(1), dim3(1), 0, s, st, tol, eps_scale, amax);
        hipLaunchKernelGGL(rrb_cand_kernel<T>, dim3(gc), dim3(256), 0, s, m, W, ldw, c0, st, cand_val, cand_idx);
        for (int j = 0; j < w; ++j) {
            hipLaunchKernelGGL(rrb_pivot_kernel<T>, dim3(1), dim3(256), 0, s, m, W, ldw, c0, w, c0 + j, st, cand_val, cand_idx, gc);
            hipLaunchKernelGGL(rrb_update_kernel<T>, dim3(gc), dim3(256), 0, s, m, W, ldw, c0, w, c0 + j, st, cand_val, cand_idx);
        }
        hipLaunchKernelGGL(rrb_advance_kernel, dim3(1), dim3(1), 0, s, st);
        int q = 0;
        LSX_HIP(hipMemcpyAsync(&q, &st->q, sizeof(int), hipMemcpyDeviceToHost, s));
        LSX_HIP(hipStreamSynchronize(s));
        const int right = n - (c0 + w);
        if (q > 0) {
            if (right > 0)
                hipLaunchKernelGGL(rrb_swap_kernel<T>, dim3((right + 255) / 256), dim3(256), 0, s, right, W + c0 + w, ldw, st);
            hipLaunchKernelGGL(rrb_gather_l_kernel<T>, dim3((rows_left + 1) / 2), dim3(256), 0, s, m, W, ldw, st, L);
            if (right > 0) {
                T *A12 = W + (size_t)r * ldw + c0 + w;
                LSX_TRY(launch_trtri<T>(h, 1, q, L, RB_W, Tinv));
                LSX_TRY(launch_trsm_block<T>(h, 1, q, right, L, RB_W, Tinv, A12, ldw));
                if (rows_left > q)
                    LSX_TRY(launch_gemm_sub<T>(h, rows_left - q, right, q, L + (size_t)q * RB_W, RB_W, A12, ldw,
                                               W + (size_t)(r + q) * ldw + c0 + w, ldw));
            }
        }
        hipLaunchKernelGGL(rrb_block_end_kernel, dim3(1), dim3(128), 0, s, st, d_pivots, d_rank);
        r += q;
    }
    const int rank = r;
    if (rank == 0) LSX_HIP(hipMemsetAsync(d_rank, 0, sizeof(int), s));
    // ---------------- phase 2: back substitution on the pivot rows
    std::vector<int32_t> hp(2 * (size_t)std::max(rank, 1));
    if (rank > 0) {
        LSX_HIP(hipMemcpyAsync(hp.data(), d_pivots, sizeof(int32_t) * 2 * rank, hipMemcpyDeviceToHost, s));
        LSX_HIP(hipStreamSynchronize(s));
    }
    const int last = rank > 0 ? ((rank - 1) / RB_W) * RB_W : -1;
    for (int kb = last; kb >= 0; kb -= RB_W) {
        const int jb = std::min(RB_W, rank - kb);
        const int col0 = (hp[2 * kb + 1] / 16) * 16;      // everything left of the block's first pivot column is zero here
        const int ncols = n - col0;
        hipLaunchKernelGGL(rrb_gather_u_kernel<T>, dim3((kb + jb + 1) / 2), dim3(256), 0, s, kb + jb, jb, W, ldw, d_pivots, kb, L);
        T *Ublk = L + (size_t)kb * RB_W;
        T *X = W + (size_t)kb * ldw + col0;
        LSX_TRY(launch_trtri<T>(h, 0, jb, Ublk, RB_W, Tinv));
        LSX_TRY(launch_trsm_block<T>(h, 0, jb, ncols, Ublk, RB_W, Tinv, X, ldw));
        if (kb > 0) LSX_TRY(launch_gemm_sub<T>(h, kb, ncols, jb, L, RB_W, X, ldw, W + col0, ldw));
    }
    hipLaunchKernelGGL(rrb_finish_kernel<T>, dim3((std::max(bar, rank) + 255) / 256, std::min(m, 65535)), dim3(256), 0, s, m, bar, W, ldw, d_pivots, rank);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// dst[i][k] = src[i][pivots[2 k + 1]] for rows i >= row_lo: the pivot columns of a row reduction as a dense m x r block
template <typename T>
__global__ __launch_bounds__(256) void gather_pivot_cols_kernel(int m, int r, int row_lo, const T *__restrict__ src, int lds,
                                                                const int32_t *__restrict__ pivots, T *__restrict__ dst, int ldd) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= r) return;
    const int pc = pivots[2 * k + 1];
    for (int i = row_lo + blockIdx.y; i < m; i += gridDim.y) dst[(size_t)i * ldd + k] = src[(size_t)i * lds + pc];
}
template <typename T>
int launch_gather_pivot_cols(lsx_handle_t h, int m, int r, int row_lo, const T *src, int lds, const int32_t *d_pivots, T *dst,
                             int ldd) {
    if (r <= 0 || m <= row_lo) return LSX_OK;
    hipLaunchKernelGGL(gather_pivot_cols_kernel<T>, dim3((r + 255) / 256, std::min(m - row_lo, 4096)), dim3(256), 0, h->stream, m, r,
                       row_lo, src, lds, d_pivots, dst, ldd);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}
// pivot columns -> exact unit vectors, rows >= rank -> exact zeros left of the bar (what the reference's arithmetic leaves)
template <typename T>
int launch_rref_finish(lsx_handle_t h, int m, int bar, T *W, int ldw, const int32_t *d_pivots, int rank) {
    hipLaunchKernelGGL(rrb_finish_kernel<T>, dim3((std::max(bar, rank) + 255) / 256, std::min(m, 65535)), dim3(256), 0, h->stream, m, bar,
                       W, ldw, d_pivots, rank);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}
template int launch_gather_pivot_cols<double>(lsx_handle_t, int, int, int, const double *, int, const int32_t *, double *, int);
template int launch_rref_finish<double>(lsx_handle_t, int, int, double *, int, const int32_t *, int);

template int rref_blocked<double>(lsx_handle_t, int, int, int, double *, int, int32_t *, int *, double, int);
template int rref_blocked<float>(lsx_handle_t, int, int, int, float *, int, int32_t *, int *, double, int);

}  // namespace lsx
