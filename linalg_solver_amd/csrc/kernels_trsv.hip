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
#include "panel_xchg.h"

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

// ---------------------------------------------------------------------------------------------
// The same solve in ONE launch per direction (option "trsv" = 1, default).  Workgroup w owns rows
// [64 w, 64 w + 64) for the whole sweep and keeps their right-hand sides on chip; all workgroups are
// co-resident (n <= 64 * #CUs).  Per 64-row step k the only cross-CU traffic is x_k: its owner forms
// it from its finished rows with the 64 x 64 diagonal-block inverse and publishes it as
// self-validating 16-byte granules {value, tag} (write-through stores, no flag, no drain); every
// workgroup still below (above, for the upper solve) polls the 64 granules and updates its rows from a
// factor block it loaded into registers before the poll.  One hop per step, no redundant work, the
// factors stream through HBM exactly once, every spin is bounded.
constexpr int CB = 64;   // rows per workgroup = step width of the cooperative solve

// sum over the 4 adjacent lanes of a quad by DPP (VALU rate; __shfl_xor goes through the LDS crossbar,
// ~100 cycles per dependent step, which at 8 right-hand sides was most of a solve step)
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_d<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(dpp_i<0xB1>(__float_as_int(v)));
    v += __int_as_float(dpp_i<0x4E>(__float_as_int(v)));
    return v;
}

template <typename T, int NR, bool LOWER>
__global__ __launch_bounds__(256) void trsv_coop_kernel(int n, const T *__restrict__ LU, int lda,
                                                        const T *__restrict__ inv64, const T *__restrict__ Bin,
                                                        int ldb, T *__restrict__ X, XGran *xb, int *status,
                                                        int spin_limit) {
    constexpr int LSD = CB + 2;            // LDS row stride of a staged factor block
    constexpr int NF = (CB * CB + 191) / 192;   // factor entries per loader thread (waves 1-3)
    __shared__ __attribute__((aligned(16))) T Ls[3][CB][LSD];   // factor blocks of steps i, i+1, i+2
    // x_k and the own right-hand sides: row r, column q at XI(r, q).  The four 16-row groups (one per
    // `part`) are 16 bytes out of step so that the four addresses of one broadcast read fall into
    // different banks (a plain [64][NR] layout puts them 1 KB apart at NR = 8: a 4-way conflict on
    // every read, which made a step three times longer at 8 right-hand sides)
    constexpr int XG = 16 * NR + 2;
    __shared__ T xk[4 * XG];
    __shared__ T bown[4 * XG];
    auto XI = [](const int r, const int q) __attribute__((always_inline)) { return (r >> 4) * XG + (r & 15) * NR + q; };
    __shared__ int s_fail;
    const int nblk = (n + CB - 1) / CB;
    // The upper solve runs from the last block row upwards: workgroups are numbered so that every dependency
    // points to a LOWER blockIdx, i.e. to a workgroup the dispatcher placed earlier -- a workgroup that is not
    // resident yet can then never be waited for by one that holds a CU.
    const int tid = threadIdx.x, w = LOWER ? (int)blockIdx.x : nblk - 1 - (int)blockIdx.x;
    const int r0 = w * CB;
    const int row_l = tid >> 2, part = tid & 3;   // 4 threads per row, 16 columns of the block each
    const int row = r0 + row_l;
    __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(xb, 0, nblk * CB * NR * (int)sizeof(XGran), 0x00020000);
    if (tid == 0) s_fail = 0;
    for (int e = tid; e < CB * NR; e += 256) {
        const int r = e / NR, q = e % NR;
        bown[XI(r, q)] = (r0 + r < n) ? Bin[(size_t)(r0 + r) * ldb + q] : T(0);
    }
    // this thread's 16 entries of its own row of inv(T_ww), used once at the end
    T dv[16];
    {
        const T *inv = inv64 + (size_t)w * CB * CB + (size_t)row_l * CB + 16 * part;
#pragma unroll
        for (int c = 0; c < 16; ++c) dv[c] = inv[c];
    }
    // Wave 0 only polls: vector-memory returns come back in issue order, so a poll queued behind
    // HBM loads would wait for them.  Waves 1-3 stream the factor blocks two steps ahead: global ->
    // registers at the top of a step, registers -> LDS at its end.
    const int k_first = LOWER ? 0 : nblk - 1;
    const int dk = LOWER ? 1 : -1;
    const int cnt = LOWER ? w : nblk - 1 - w;   // steps before my own
    T fr[NF];
    auto fetch = [&](const int i) __attribute__((always_inline)) {       // step index -> registers
        const int k = k_first + i * dk;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int e = (tid - 64) + 192 * j;
            const int r = e / CB, c = e % CB;
            const int gr = r0 + r, gc = k * CB + c;
            fr[j] = (e < CB * CB && gr < n && gc < n) ? LU[(size_t)gr * lda + gc] : T(0);
        }
    };
    auto stash = [&](const int i) __attribute__((always_inline)) {       // registers -> LDS buffer of step i
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int e = (tid - 64) + 192 * j;
            if (e < CB * CB) Ls[i % 3][e / CB][e % CB] = fr[j];
        }
    };
    if (tid >= 64) {
        if (cnt > 0) { fetch(0); stash(0); }
        if (cnt > 1) { fetch(1); stash(1); }
    }
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
        const int k = k_first + i * dk;
        if (tid >= 64) {
            if (i + 2 < cnt) fetch(i + 2);
        } else {
            // ---- x_k: CB x NR granules, wave 0: lane r takes row r, all its NR granules per shot
            const int gr = k * CB + tid;
            T val[NR];
#pragma unroll
            for (int q = 0; q < NR; ++q) val[q] = T(0);
            if (gr < n) {
                int spins = 0;
                for (;;) {
                    u4 g[NR];
                    const int oz = opaque_zero();
#pragma unroll
                    for (int q = 0; q < NR; ++q)
                        g[q] = __builtin_amdgcn_raw_buffer_load_b128(r_x, (gr * NR + q) * (int)sizeof(XGran), oz, 16);
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < NR; ++q) ok &= g[q].z == 1u;
                    if (ok) {
#pragma unroll
                        for (int q = 0; q < NR; ++q) {
                            if (sizeof(T) == 8)
                                val[q] = (T)__longlong_as_double((long long)(((unsigned long long)g[q].y << 32) | g[q].x));
                            else
                                val[q] = (T)__uint_as_float(g[q].x);
                        }
                        break;
                    }
                    if (s_fail || ++spins > spin_limit) { s_fail = 1; break; }
                }
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) xk[XI(tid, q)] = val[q];
        }
        __syncthreads();
        // ---- own rows: b_row -= T[row, block k] * x_k
        {
            const T *lrow = &Ls[i % 3][row_l][16 * part];
            T acc[NR];
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[q] = T(0);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const T l = lrow[c];
#pragma unroll
                for (int q = 0; q < NR; ++q) acc[q] += l * xk[part * XG + c * NR + q];
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[q] = quad_sum(acc[q]);
            if (part == 0) {
#pragma unroll
                for (int q = 0; q < NR; ++q) bown[XI(row_l, q)] -= acc[q];
            }
        }
        if (tid >= 64 && i + 2 < cnt) stash(i + 2);   // buffer (i+2)%3 was last read in step i-1
        __syncthreads();
    }
    // ---- my rows are final: x_w = inv(T_ww) b_w, published and stored
    {
        T acc[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) acc[q] = T(0);
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[q] += dv[c] * bown[part * XG + c * NR + q];
#pragma unroll
        for (int q = 0; q < NR; ++q) acc[q] = quad_sum(acc[q]);
        if (part == 0 && row < n) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                unsigned long long bits;
                if (sizeof(T) == 8) bits = (unsigned long long)__double_as_longlong((double)acc[q]);
                else bits = (unsigned long long)__float_as_uint((float)acc[q]);
                u4 v;
                v.x = (unsigned)bits; v.y = (unsigned)(bits >> 32); v.z = 1u; v.w = 0u;
                __builtin_amdgcn_raw_buffer_store_b128(v, r_x, (row * NR + q) * (int)sizeof(XGran), 0, 16);
                X[(size_t)row * NR + q] = acc[q];
            }
        }
    }
    // a time-out (x_k substituted by zero above) must not pass for a result: the host entry points read this word
    if (s_fail && tid == 0) atomicExch(status, 1);
}

// cooperative variant: true when it ran
template <typename T, int NR>
static int trsv_coop_run(lsx_handle_t h, int n, const T *LU, int lda, T *B, int ldb, T *X, const T *inv64L,
                         const T *inv64U) {
    const int wgs = (n + CB - 1) / CB;
    {
        // one launch per direction; exchange area (zeroed) + status word in the scratch
        const size_t xbytes = (size_t)wgs * CB * NR * sizeof(XGran);
        if (256 + 2 * xbytes > h->scratch_bytes) { set_error("trsv: scratch too small"); return LSX_ERR_INTERNAL; }
        int *status = h->dev_status + 1;   // persistent word: read (and cleared) by the host entry points / lsx_check_status
        const int spin_limit = h->spin_limit;
        XGran *xb0 = (XGran *)((char *)h->scratch + 256), *xb1 = (XGran *)((char *)h->scratch + 256 + xbytes);
        LSX_HIP(hipMemsetAsync(h->scratch, 0, 256 + 2 * xbytes, h->stream));
        hipLaunchKernelGGL((trsv_coop_kernel<T, NR, true>), dim3(wgs), dim3(256), 0, h->stream, n, LU, lda, inv64L,
                           (const T *)B, ldb, X, xb0, status, spin_limit);
        LSX_TRY(launch_copy2d<T>(h, n, NR, X, NR, B, ldb));  // y is the right-hand side of U x = y
        hipLaunchKernelGGL((trsv_coop_kernel<T, NR, false>), dim3(wgs), dim3(256), 0, h->stream, n, LU, lda, inv64U,
                           (const T *)B, ldb, X, xb1, status, spin_limit);
        LSX_HIP(hipGetLastError());
        return LSX_OK;
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
    LSX_TRY(launch_trtri_both<T>(h, n, LU, lda, inv64L, inv64U));
    // every workgroup must be resident at once; at 8 right-hand sides and large n the per-step path is
    // as fast (the LDS traffic of the 8-wide update dominates either way)
    if (h->trsv_mode == 1 && (n + CB - 1) / CB <= h->num_cu && (nrhs <= 4 || n <= 6144)) {
        switch (nrhs) {
            case 1: return trsv_coop_run<T, 1>(h, n, LU, lda, B, ldb, X, inv64L, inv64U);
            case 2: return trsv_coop_run<T, 2>(h, n, LU, lda, B, ldb, X, inv64L, inv64U);
            case 4: return trsv_coop_run<T, 4>(h, n, LU, lda, B, ldb, X, inv64L, inv64U);
            case 8: return trsv_coop_run<T, 8>(h, n, LU, lda, B, ldb, X, inv64L, inv64U);
        }
    }
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
