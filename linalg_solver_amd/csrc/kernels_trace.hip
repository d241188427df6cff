// Traced row reduction: the reference's Gauss-Jordan in the reference's own order, with its step
// log, on the device (step-log synthesis).
//
// Follows linalg_solver/linalg.py:534-630 operation for operation:
//   * pivot test `A[pi][pj] == 0` is EXACT; on zero the first lower row with a non-zero entry is
//     swapped in (S step, :548-564); no such row: the column is skipped, the pivot row stays (:565-567);
//   * the pivot row is divided by the pivot from column pj rightwards unless the pivot == 1; an N step
//     is logged only if some value changed (:569-586);
//   * every lower row with a non-zero factor gets `a - factor*p` from column pj rightwards; an E step is
//     logged only if such a row exists and some value changed (:587-607);
//   * after the sweep the entries ABOVE each pivot are eliminated in a separate pass, last pivot
//     first (:611-629).
// The arithmetic is the reference's (Python floats = IEEE binary64, one rounding per operation):
// the product of `a - f*p` is passed through an empty asm so that it is rounded on its own (the library is
// built with -ffp-contract=fast; neither the __dmul_rn / __dsub_rn intrinsics nor a contract(off) pragma
// kept hipcc from emitting an FMA here; this file is also compiled with -ffp-contract=off), so the
// reduced matrix, the pivots and the step labels are BIT-IDENTICAL to the reference's (tests/golden).
//
// A byte per entry carries the reference's entry TYPE along: the reference works on Python objects, so
// int - int*int stays an int and only division (or an operand that already is a float) produces a
// float.  The caller passes which entries are ints; the mask follows swaps,
// normalisation clears it, elimination ands the three operands' flags; the host turns flagged entries
// back into ints.  (Integer values are exact in binary64 up to 2^53.)
//
// This is the teaching/trace path for small and medium inputs, not the fast path: HBM-bound, several
// small launches per column, state on the device, no host round trip.  Optional snapshots of the
// matrix after every logged step (the reference's `intermediate_matrices`).
#include "common.h"

#pragma clang fp contract(off)

namespace lsx {

struct TraceState {
    int pi, pj;       // next pivot position
    int npiv;         // pivots placed so far
    int nsteps;       // steps logged so far
    int phase_step;   // the phase that just ran logged a step (the snapshot kernel keys on this)
    int active;       // the current column has a pivot: normalise + sweep run
    int anyfac;       // sweep: some row had a non-zero factor
    int changed;      // sweep / normalise: some value changed
    int done;         // forward pass finished
    int overflow;     // more steps than the log can hold
    int pad[6];
};

enum { TR_SWAP = 0, TR_NORM = 1, TR_ELIM_BELOW = 2, TR_ELIM_ABOVE = 3 };

__device__ __forceinline__ void trace_log(TraceState *st, int32_t *steps, int max_steps, int kind, int a, int b) {
    const int s = st->nsteps;
    if (s < max_steps) {
        steps[4 * s + 0] = kind;
        steps[4 * s + 1] = a;
        steps[4 * s + 2] = b;
        steps[4 * s + 3] = 0;
    } else {
        st->overflow = 1;
    }
    st->nsteps = s + 1;
    st->phase_step = 1;
}

__global__ void trace_init_kernel(TraceState *st) {
    if (threadIdx.x == 0) {
        st->pi = st->pj = st->npiv = st->nsteps = st->phase_step = st->active = 0;
        st->anyfac = st->changed = st->done = st->overflow = 0;
    }
}

// Phase A (one workgroup): skip zero columns, find the pivot row, swap it in.
__global__ __launch_bounds__(256) void trace_select_kernel(int m, int n, int bar, double *__restrict__ R,
                                                           int ldr, unsigned char *__restrict__ Tm,
                                                           TraceState *st, int32_t *steps, int max_steps) {
    __shared__ int s_first[256];
    __shared__ int s_pi, s_pj;
    const int tid = threadIdx.x;
    if (tid == 0) {
        st->phase_step = 0;
        st->active = 0;
        s_pi = st->pi;
        s_pj = st->pj;
    }
    __syncthreads();
    if (st->done) return;
    for (;;) {
        const int pi = s_pi, pj = s_pj;
        if (pi >= m || pj >= bar) {
            if (tid == 0) { st->done = 1; st->pi = pi; st->pj = pj; }
            return;
        }
        if (R[(size_t)pi * ldr + pj] != 0.0) {
            if (tid == 0) { st->pi = pi; st->pj = pj; st->active = 1; }
            return;
        }
        // first lower row with a non-zero entry in this column (linalg.py:550-552)
        int first = 0x7fffffff;
        for (int i = pi + 1 + tid; i < m; i += 256)
            if (R[(size_t)i * ldr + pj] != 0.0) { first = i; break; }
        s_first[tid] = first;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if (tid < k) s_first[tid] = min(s_first[tid], s_first[tid + k]);
            __syncthreads();
        }
        const int p = s_first[0];
        __syncthreads();
        if (p == 0x7fffffff) {  // zero column: advance the column only (:565-567)
            if (tid == 0) s_pj = pj + 1;
            __syncthreads();
            continue;
        }
        for (int j = tid; j < n; j += 256) {  // the reference swaps whole rows (:552)
            const double x = R[(size_t)pi * ldr + j];
            R[(size_t)pi * ldr + j] = R[(size_t)p * ldr + j];
            R[(size_t)p * ldr + j] = x;
            const unsigned char t = Tm[(size_t)pi * n + j];
            Tm[(size_t)pi * n + j] = Tm[(size_t)p * n + j];
            Tm[(size_t)p * n + j] = t;
        }
        if (tid == 0) {
            st->pi = pi; st->pj = pj; st->active = 1;
            trace_log(st, steps, max_steps, TR_SWAP, pi + 1, p + 1);
        }
        return;
    }
}

// Phase B (one workgroup): divide the pivot row by the pivot, from the pivot column rightwards.
__global__ __launch_bounds__(256) void trace_normalize_kernel(int n, double *__restrict__ R, int ldr,
                                                              unsigned char *__restrict__ Tm, TraceState *st,
                                                              int32_t *steps, int max_steps) {
    __shared__ int s_changed;
    const int tid = threadIdx.x;
    if (tid == 0) { st->phase_step = 0; s_changed = 0; }
    __syncthreads();
    if (!st->active) return;
    const int pi = st->pi, pj = st->pj;
    double *row = R + (size_t)pi * ldr;
    const double factor = row[pj];
    __syncthreads();  // everybody holds the pivot before it is overwritten
    if (factor != 1.0) {
        int ch = 0;
        for (int j = pj + tid; j < n; j += 256) {
            const double old = row[j];
            const double nw = old / factor;   // correctly rounded IEEE division
            row[j] = nw;
            Tm[(size_t)pi * n + j] = 0;   // true division: always a float
            ch |= (nw != old);
        }
        if (ch) atomicOr(&s_changed, 1);
    }
    __syncthreads();
    if (tid == 0 && s_changed) trace_log(st, steps, max_steps, TR_NORM, pi + 1, 0);
}

// Phase C: one workgroup per row (grid-stride): rows below the pivot (above == 0) or above the pivot
// `pivots[idx]` of the backward pass (above == 1).
__global__ __launch_bounds__(256) void trace_sweep_kernel(int m, int n, double *__restrict__ R, int ldr,
                                                          unsigned char *__restrict__ Tm, TraceState *st,
                                                          const int32_t *__restrict__ pivots, int above,
                                                          int back_it) {
    int prow, pcol, k0, k1;
    if (!above) {
        if (!st->active) return;
        prow = st->pi; pcol = st->pj; k0 = prow + 1; k1 = m;
    } else {
        const int idx = st->npiv - 1 - back_it;
        if (idx < 0) return;
        prow = pivots[2 * idx]; pcol = pivots[2 * idx + 1]; k0 = 0; k1 = prow;
    }
    const double *prow_p = R + (size_t)prow * ldr;
    const int tid = threadIdx.x;
    for (int k = k0 + blockIdx.x; k < k1; k += gridDim.x) {
        double *row = R + (size_t)k * ldr;
        const double factor = row[pcol];
        const unsigned char fint = Tm[(size_t)k * n + pcol];
        __syncthreads();  // the factor is read by every thread before column pcol is rewritten
        if (factor == 0.0) continue;   // workgroup-uniform
        int ch = 0;
        for (int j = pcol + tid; j < n; j += 256) {
            const double old = row[j];
            double prod = factor * prow_p[j];
            asm volatile("" : "+v"(prod));           // the product is rounded on its own: never an FMA
            const double nw = old - prod;
            row[j] = nw;
            Tm[(size_t)k * n + j] = Tm[(size_t)k * n + j] & fint & Tm[(size_t)prow * n + j];
            ch |= (nw != old);
        }
        if (tid == 0) atomicOr(&st->anyfac, 1);
        if (ch) atomicOr(&st->changed, 1);
        __syncthreads();
    }
}

// Phase D (one thread): log the elimination step, place the pivot, advance.
__global__ void trace_finalize_kernel(TraceState *st, int32_t *pivots, int32_t *steps, int max_steps, int above,
                                      int back_it) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->phase_step = 0;
    if (!above) {
        if (!st->active) return;
        // E step only if some lower row had a non-zero factor AND some value changed (:598)
        if (st->anyfac && st->changed) trace_log(st, steps, max_steps, TR_ELIM_BELOW, st->pj + 1, 0);
        pivots[2 * st->npiv] = st->pi;
        pivots[2 * st->npiv + 1] = st->pj;
        st->npiv += 1;
        st->pi += 1;
        st->pj += 1;
        st->active = 0;
    } else {
        const int idx = st->npiv - 1 - back_it;
        if (idx >= 0 && st->changed) trace_log(st, steps, max_steps, TR_ELIM_ABOVE, pivots[2 * idx + 1] + 1, 0);
    }
    st->anyfac = 0;
    st->changed = 0;
}

// Snapshot of the matrix after the step that was just logged (if any): snaps[step][m][n], dense.
__global__ __launch_bounds__(256) void trace_snap_kernel(int m, int n, const double *__restrict__ R, int ldr,
                                                         const unsigned char *__restrict__ Tm,
                                                         const TraceState *st, double *__restrict__ snaps,
                                                         unsigned char *__restrict__ snap_t, int max_snaps) {
    if (!st->phase_step) return;
    const int s = st->nsteps - 1;
    if (s >= max_snaps) return;
    double *dst = snaps + (size_t)s * m * n;
    unsigned char *dst_t = snap_t + (size_t)s * m * n;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)m * n; e += (size_t)gridDim.x * 256) {
        dst[e] = R[(e / n) * ldr + (e % n)];
        dst_t[e] = Tm[e];
    }
}

__global__ void trace_result_kernel(const TraceState *st, int *out) {
    if (threadIdx.x == 0) { out[0] = st->npiv; out[1] = st->nsteps; out[2] = st->overflow; }
}

// d_out: {number of pivots, number of steps, overflow flag}
// Tm: m x n bytes (dense): in = which entries are ints, out = which still are; d_snap_t: max_snaps x m x n.
int launch_rref_trace(lsx_handle_t h, int m, int n, int bar, double *R, int ldr, unsigned char *Tm,
                      int32_t *d_pivots, int32_t *d_steps, int max_steps, double *d_snaps,
                      unsigned char *d_snap_t, int max_snaps, int *d_out) {
    // the caller (api.hip) has made sure the handle's scratch holds at least 4 KB
    TraceState *st = (TraceState *)h->scratch;
    hipStream_t s = h->stream;
    const int iters = m < bar ? m : bar;
    const int rows_grid = m < 1024 ? (m > 0 ? m : 1) : 1024;
    const size_t elems = (size_t)m * n;
    const int snap_grid = (int)((elems + 255) / 256 < 2048 ? (elems + 255) / 256 : 2048);
    auto snap = [&]() {
        if (d_snaps && max_snaps > 0)
            hipLaunchKernelGGL(trace_snap_kernel, dim3(snap_grid > 0 ? snap_grid : 1), dim3(256), 0, s, m, n, R, ldr,
                               Tm, st, d_snaps, d_snap_t, max_snaps);
    };
    hipLaunchKernelGGL(trace_init_kernel, dim3(1), dim3(64), 0, s, st);
    // forward pass: at most min(m, bar) pivots; the state stops the chain early on its own
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(trace_select_kernel, dim3(1), dim3(256), 0, s, m, n, bar, R, ldr, Tm, st, d_steps, max_steps);
        snap();
        hipLaunchKernelGGL(trace_normalize_kernel, dim3(1), dim3(256), 0, s, n, R, ldr, Tm, st, d_steps, max_steps);
        snap();
        hipLaunchKernelGGL(trace_sweep_kernel, dim3(rows_grid), dim3(256), 0, s, m, n, R, ldr, Tm, st, d_pivots, 0, 0);
        hipLaunchKernelGGL(trace_finalize_kernel, dim3(1), dim3(64), 0, s, st, d_pivots, d_steps, max_steps, 0, 0);
        snap();
    }
    // backward pass: last pivot first
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(trace_sweep_kernel, dim3(rows_grid), dim3(256), 0, s, m, n, R, ldr, Tm, st, d_pivots, 1, it);
        hipLaunchKernelGGL(trace_finalize_kernel, dim3(1), dim3(64), 0, s, st, d_pivots, d_steps, max_steps, 1, it);
        snap();
    }
    hipLaunchKernelGGL(trace_result_kernel, dim3(1), dim3(64), 0, s, st, d_out);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

}  // namespace lsx
