// Few right-hand sides (nrhs <= 8): blocked triangular solves with ONE launch per 128-row block
// step -- the solve-latency path of BASELINE config #2 (reference: back-substitution loops
// linalg_solver/linalg.py:587-596 and 611-621 applied to the right-hand side only).
//
//   merge128   inv(T_kk) for every 128 x 128 diagonal block from its two 64 x 64 block inverses
//              (lower: X21 = -X22 * L21 * X11; upper: X12 = -X11 * U12 * X22), stored TRANSPOSED so
//              the matvec below reads it coalesced.
//   trsv_step  workgroup i owns rows [128 i, 128 i + 128):  b_i -= T[i,k] * x_k  from an LDS-staged
//              tile; the workgroup that owns the NEXT block then forms x_next = inv(T_next) * b_next
//              at once, so a step costs one launch, and the factors stream through HBM exactly once.
#include "common.h"

namespace lsx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T>
struct MfmaV;
template <>
struct MfmaV<double> {
    typedef d4 acc_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct MfmaV<float> {
    typedef f4 acc_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

constexpr int VB = 128;        // block edge of the solve
constexpr int VH = 64;         // half block
constexpr int VLD = VH + 2;

template <typename T>
__device__ void v_gemm64(const T *A, const T *B, T *D, T alpha, int wave, int lane) {
    // D (64x64) = alpha * A (64x64) * B (64x64), all in LDS with leading dimension VLD
    typedef typename MfmaV<T>::acc_t acc_t;
    const int lc = lane & 15, lq = lane >> 4;
    for (int tile = wave; tile < 16; tile += 4) {
        const int i0 = (tile >> 2) * 16, j0 = (tile & 3) * 16;
        acc_t acc = {0, 0, 0, 0};
        for (int k0 = 0; k0 < VH; k0 += 4)
            acc = MfmaV<T>::mma(A[(i0 + lc) * VLD + k0 + lq], B[(k0 + lq) * VLD + j0 + lc], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) D[(i0 + MfmaV<T>::crow(lane, r)) * VLD + j0 + lc] = alpha * acc[r];
    }
}

// inv128T[blk][c][r] = inv(T_blk)[r][c] for every 128-block of the n x n triangle.
template <typename T>
__global__ __launch_bounds__(256) void merge128_kernel(int lower, int n, const T *__restrict__ Tm, int ldt,
                                                       const T *__restrict__ inv64, T *__restrict__ inv128T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *X11 = (T *)smem, *X22 = X11 + VH * VLD, *OFF = X22 + VH * VLD, *W = OFF + VH * VLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x, r0 = blk * VB;
    const int nb64 = (n + VH - 1) / VH;
    for (int e = tid; e < VH * VH; e += 256) {
        const int i = e / VH, j = e % VH;
        X11[i * VLD + j] = inv64[(size_t)(2 * blk) * VH * VH + e];
        // a trailing half block that does not exist is the identity
        X22[i * VLD + j] = (2 * blk + 1 < nb64) ? inv64[(size_t)(2 * blk + 1) * VH * VH + e] : (i == j ? T(1) : T(0));
        const int gi = lower ? r0 + VH + i : r0 + i, gj = lower ? r0 + j : r0 + VH + j;
        OFF[i * VLD + j] = (gi < n && gj < n) ? Tm[(size_t)gi * ldt + gj] : T(0);
    }
    __syncthreads();
    if (lower) v_gemm64<T>(OFF, X11, W, T(1), wave, lane);   // L21 * X11
    else v_gemm64<T>(OFF, X22, W, T(1), wave, lane);         // U12 * X22
    __syncthreads();
    if (lower) v_gemm64<T>(X22, W, OFF, T(-1), wave, lane);  // -X22 * (L21 X11)
    else v_gemm64<T>(X11, W, OFF, T(-1), wave, lane);        // -X11 * (U12 X22)
    __syncthreads();
    T *out = inv128T + (size_t)blk * VB * VB;
    for (int e = tid; e < VB * VB; e += 256) {
        const int c = e / VB, r = e % VB;  // out[c][r] = inv[r][c]
        T v;
        if (r < VH && c < VH) v = X11[r * VLD + c];
        else if (r >= VH && c >= VH) v = X22[(r - VH) * VLD + c - VH];
        else if (lower) v = (r >= VH) ? OFF[(r - VH) * VLD + c] : T(0);
        else v = (r < VH) ? OFF[r * VLD + c - VH] : T(0);
        out[e] = v;
    }
}

// One block step k.  Every workgroup first forms x_k = inv(T_kk) * b_k itself (b_k is final: the
// previous launches updated it; the transposed inverse is read coalesced and is L2-resident after
// the first workgroup of each XCD touched it), then updates its own 32 rows:
//   lower:  b_i -= L[i, k] * x_k   for rows i below block k        (linalg.py:587-596)
//   upper:  b_i -= U[i, k] * x_k   for rows i above block k        (linalg.py:611-621)
// Workgroup 0 also stores x_k.  Grid = number of 32-row groups to update (>= 1; with nothing left
// to update the single workgroup only stores x_k).
constexpr int VR = 32;  // rows per workgroup

template <typename T, int NR>
__global__ __launch_bounds__(256) void trsv_step_kernel(int lower, int n, const T *__restrict__ LU, int lda,
                                                        const T *__restrict__ inv128T, int kblk, int row_lo,
                                                        int row_hi, T *__restrict__ B, int ldb,
                                                        T *__restrict__ X) {
    __shared__ T bk[VB][NR];
    __shared__ T xk[VB][NR];
    const int tid = threadIdx.x;
    const int c0 = kblk * VB;
    for (int e = tid; e < VB * NR; e += 256) {
        const int r = e / NR, q = e % NR;
        bk[r][q] = (c0 + r < n) ? B[(size_t)(c0 + r) * ldb + q] : T(0);
    }
    __syncthreads();
    {   // x_k: thread pair (r, half) sums half of row r of inv(T_kk) against b_k
        const T *inv = inv128T + (size_t)kblk * VB * VB;
        const int r = tid >> 1, hf = tid & 1;
        T acc[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) acc[q] = T(0);
#pragma unroll 8
        for (int c = hf; c < VB; c += 2) {
            const T t = inv[(size_t)c * VB + r];
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[q] += t * bk[c][q];
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) acc[q] += __shfl_xor(acc[q], 1, 64);
        if (hf == 0) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                xk[r][q] = acc[q];
                if (blockIdx.x == 0 && c0 + r < n) X[(size_t)(c0 + r) * NR + q] = acc[q];
            }
        }
    }
    __syncthreads();
    // own rows: 8 threads per row, 16 consecutive columns each (128 B), shuffle-reduce over the 8
    const int row = row_lo + blockIdx.x * VR + (tid >> 3);
    const int part = tid & 7;
    if (row < row_hi) {
        const T *src = LU + (size_t)row * lda + c0 + 16 * part;
        T acc[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) acc[q] = T(0);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const T t = (c0 + 16 * part + c < n) ? src[c] : T(0);
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[q] += t * xk[16 * part + c][q];
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            acc[q] += __shfl_xor(acc[q], 1, 64);
            acc[q] += __shfl_xor(acc[q], 2, 64);
            acc[q] += __shfl_xor(acc[q], 4, 64);
        }
        if (part == 0) {
#pragma unroll
            for (int q = 0; q < NR; ++q) B[(size_t)row * ldb + q] -= acc[q];
        }
    }
}

template <typename T, int NR>
static int trsv_run(lsx_handle_t h, int n, const T *LU, int lda, T *B, int ldb, T *X, const T *inv128L,
                    const T *inv128U) {
    const int nblk = (n + VB - 1) / VB;
    for (int k = 0; k < nblk; ++k) {  // forward: L y = b
        const int lo = (k + 1) * VB, hi = n;
        const int grid = hi > lo ? (hi - lo + VR - 1) / VR : 1;
        hipLaunchKernelGGL((trsv_step_kernel<T, NR>), dim3(grid), dim3(256), 0, h->stream, 1, n, LU, lda, inv128L, k,
                           lo, hi, B, ldb, X);
    }
    LSX_TRY(launch_copy2d<T>(h, n, NR, X, NR, B, ldb));  // y is the right-hand side of U x = y
    for (int k = nblk - 1; k >= 0; --k) {
        const int hi = k * VB;
        const int grid = hi > 0 ? (hi + VR - 1) / VR : 1;
        hipLaunchKernelGGL((trsv_step_kernel<T, NR>), dim3(grid), dim3(256), 0, h->stream, 0, n, LU, lda, inv128U, k,
                           0, hi, B, ldb, X);
    }
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// X (n x nrhs, dense) <- U^-1 L^-1 B for B already row-permuted; B is used as work space.
// nrhs must be 1, 2, 4 or 8 (the caller pads).
template <typename T>
int lu_solve_few_rhs(lsx_handle_t h, int n, int nrhs, const T *LU, int lda, T *B, int ldb, T *X, T *inv64L,
                     T *inv64U, T *inv128L, T *inv128U) {
    const int nblk = (n + VB - 1) / VB;
    ProfScope ps(h, LSX_PROF_TRSM, 2.0 * n * (double)n * nrhs, 2.0 * sizeof(T) * n * (double)n);
    const size_t shm_m = (size_t)4 * VH * VLD * sizeof(T);
    LSX_HIP(hipFuncSetAttribute((const void *)merge128_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_m));
    LSX_TRY(launch_trtri<T>(h, 1, n, LU, lda, inv64L));
    LSX_TRY(launch_trtri<T>(h, 0, n, LU, lda, inv64U));
    hipLaunchKernelGGL(merge128_kernel<T>, dim3(nblk), dim3(256), shm_m, h->stream, 1, n, LU, lda, inv64L, inv128L);
    hipLaunchKernelGGL(merge128_kernel<T>, dim3(nblk), dim3(256), shm_m, h->stream, 0, n, LU, lda, inv64U, inv128U);
    switch (nrhs) {
        case 1: return trsv_run<T, 1>(h, n, LU, lda, B, ldb, X, inv128L, inv128U);
        case 2: return trsv_run<T, 2>(h, n, LU, lda, B, ldb, X, inv128L, inv128U);
        case 4: return trsv_run<T, 4>(h, n, LU, lda, B, ldb, X, inv128L, inv128U);
        case 8: return trsv_run<T, 8>(h, n, LU, lda, B, ldb, X, inv128L, inv128U);
    }
    set_error("lu_solve_few_rhs: nrhs must be 1, 2, 4 or 8");
    return LSX_ERR_ARG;
}

template int lu_solve_few_rhs<double>(lsx_handle_t, int, int, const double *, int, double *, int, double *, double *,
                                      double *, double *, double *);
template int lu_solve_few_rhs<float>(lsx_handle_t, int, int, const float *, int, float *, int, float *, float *, float *,
                                     float *, float *);

}  // namespace lsx
