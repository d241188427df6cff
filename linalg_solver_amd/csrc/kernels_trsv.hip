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
    // D (64x64) = alpha * A (64x64) * B (64x64), all in LDS with leading dimension VLD.  Wave w owns tile row w: its
    // four 16 x 16 tiles share the A operand and run as four independent MFMA chains (one chain per tile, tile after
    // tile, was 64 dependent MFMA + LDS round trips per wave: most of the 27 us the preparation launch took)
    typedef typename MfmaV<T>::acc_t acc_t;
    const int lc = lane & 15, lq = lane >> 4;
    const int i0 = wave * 16;
    acc_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 4
    for (int k0 = 0; k0 < VH; k0 += 4) {
        const T a = A[(i0 + lc) * VLD + k0 + lq];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = MfmaV<T>::mma(a, B[(k0 + lq) * VLD + 16 * t + lc], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) D[(i0 + MfmaV<T>::crow(lane, r)) * VLD + 16 * t + lc] = alpha * acc[t][r];
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

// ---------------------------------------------------------------------------------------------
// Few right-hand sides, third form (option "trsv" = 2, default): 128-row steps and helper workgroups.
//
// What bounds the one-launch-per-direction form above is its chain of 64-row steps: n / 64 hand-overs from one
// workgroup to the next, each one device-scope hop plus two workgroup barriers (1.15 us per step: 74 us per
// direction at n = 4096, 2 x 74 of the 204 us of a solve -- the rest was seven small launches in front of and
// between the sweeps).  Here
//  * a step is 128 rows: half as many hand-overs;
//  * block row i belongs to H workgroups, blockIdx = ord(i) H + h.  h = 0 is the OWNER: it keeps inv(T_ii) in
//    registers, takes the block next to the diagonal only, and forms x_i.  h >= 1 are HELPERS: they stream the other
//    blocks of the row (every (H-1)-th one) as the x_j they need appear and hand their 128 partial sums to the owner.
//    With H = 8 (n <= 4096) the owners are the workgroups with blockIdx % 8 == 0, i.e. on one XCD, and all 256 CUs
//    stream the factors;
//  * inside a workgroup wave 0 only polls (vector-memory returns come back in issue order: a poll behind a block load
//    would wait for HBM) and hands x_j to the eight compute waves through LDS, ONE barrier per step; a compute
//    thread owns 32 columns of one row, and the lanes of a 16-lane DPP row share the column range, so x_j reaches
//    the multiply-adds as a DPP operand (row_newbcast) from two registers per thread instead of 32 LDS reads;
//  * the interchanges (ipiv -> permutation -> gather of B), the 128 x 128 diagonal-block inverses and nothing else
//    precede the sweeps: one launch (solve_prep_kernel) behind the 64 x 64 block inverses; the exchange areas are
//    validated by a per-call epoch instead of being cleared; the upper sweep writes the caller's B directly.
// Deterministic: every sum has a fixed order (columns ascending inside a thread, then the four column ranges, then
// the helpers in order).  Reference loops: linalg.py:587-596 and 611-621 on the right-hand side.
constexpr int SB = 128;                   // rows per step
constexpr int S2_THREADS = 64 + 512;      // wave 0 polls, waves 1..8 compute

template <typename T, bool SPLIT>
__device__ __forceinline__ void dpp_fma32(T &acc, const T xa, const T xb, const T (&blk)[32]) {
    // acc += sum_c blk[c] * xa(lane c of the row) + blk[16 + c] * xb(lane c of the row).  SPLIT (1 or 2 right-hand
    // sides): four independent chains of eight, combined in a fixed order -- a single chain of 32 dependent DPP
    // multiply-adds was a quarter of a solve step; with 4 or 8 right-hand sides their chains interleave by themselves
    // and the extra registers would spill.
    T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
    if (!SPLIT) {
#define S2_E(X, C, K)                                                                                                     \
    if (sizeof(T) == 8) {                                                                                                  \
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(X), "v"(blk[K]));    \
    } else {                                                                                                               \
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(X), "v"(blk[K]));    \
    }
        asm volatile("s_nop 1" : : "v"(xa), "v"(xb));
        S2_E(xa, 0, 0) S2_E(xa, 1, 1) S2_E(xa, 2, 2) S2_E(xa, 3, 3) S2_E(xa, 4, 4) S2_E(xa, 5, 5) S2_E(xa, 6, 6) S2_E(xa, 7, 7)
        S2_E(xa, 8, 8) S2_E(xa, 9, 9) S2_E(xa, 10, 10) S2_E(xa, 11, 11) S2_E(xa, 12, 12) S2_E(xa, 13, 13) S2_E(xa, 14, 14) S2_E(xa, 15, 15)
        S2_E(xb, 0, 16) S2_E(xb, 1, 17) S2_E(xb, 2, 18) S2_E(xb, 3, 19) S2_E(xb, 4, 20) S2_E(xb, 5, 21) S2_E(xb, 6, 22) S2_E(xb, 7, 23)
        S2_E(xb, 8, 24) S2_E(xb, 9, 25) S2_E(xb, 10, 26) S2_E(xb, 11, 27) S2_E(xb, 12, 28) S2_E(xb, 13, 29) S2_E(xb, 14, 30) S2_E(xb, 15, 31)
#undef S2_E
        return;
    }
#define S2_F(A, X, C, K)                                                                                                  \
    if (sizeof(T) == 8) {                                                                                                  \
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" : "+v"(A) : "v"(X), "v"(blk[K]));      \
    } else {                                                                                                               \
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" : "+v"(A) : "v"(X), "v"(blk[K]));      \
    }
    asm volatile("s_nop 1" : : "v"(xa), "v"(xb));   // a DPP read of a register written just before needs two wait states
    S2_F(a0, xa, 0, 0) S2_F(a1, xa, 1, 1) S2_F(a2, xa, 2, 2) S2_F(a3, xa, 3, 3)
    S2_F(a0, xa, 4, 4) S2_F(a1, xa, 5, 5) S2_F(a2, xa, 6, 6) S2_F(a3, xa, 7, 7)
    S2_F(a0, xa, 8, 8) S2_F(a1, xa, 9, 9) S2_F(a2, xa, 10, 10) S2_F(a3, xa, 11, 11)
    S2_F(a0, xa, 12, 12) S2_F(a1, xa, 13, 13) S2_F(a2, xa, 14, 14) S2_F(a3, xa, 15, 15)
    S2_F(a0, xb, 0, 16) S2_F(a1, xb, 1, 17) S2_F(a2, xb, 2, 18) S2_F(a3, xb, 3, 19)
    S2_F(a0, xb, 4, 20) S2_F(a1, xb, 5, 21) S2_F(a2, xb, 6, 22) S2_F(a3, xb, 7, 23)
    S2_F(a0, xb, 8, 24) S2_F(a1, xb, 9, 25) S2_F(a2, xb, 10, 26) S2_F(a3, xb, 11, 27)
    S2_F(a0, xb, 12, 28) S2_F(a1, xb, 13, 29) S2_F(a2, xb, 14, 30) S2_F(a3, xb, 15, 31)
#undef S2_F
    acc += (a0 + a1) + (a2 + a3);
}

// Sum over the four 16-lane rows of a wave, valid in lanes 0..15: gfx950's v_permlane32_swap / v_permlane16_swap move
// whole lane rows at VALU rate (the __shfl_xor form is four ds_bpermute round trips per double: ~0.2 us of a step).
__device__ __forceinline__ unsigned s2_rows_partner32(const unsigned v) {   // lanes 0..31: v of lane + 32
    return __builtin_amdgcn_permlane32_swap(v, v, false, false)[1];
}
__device__ __forceinline__ unsigned s2_rows_partner16(const unsigned v) {   // lanes 0..15 (and 32..47): v of lane + 16
    return __builtin_amdgcn_permlane16_swap(v, v, false, false)[1];
}
__device__ __forceinline__ double s2_row_sum(const double v) {
    const double p = __hiloint2double((int)s2_rows_partner32((unsigned)__double2hiint(v)), (int)s2_rows_partner32((unsigned)__double2loint(v)));
    const double s1 = v + p;
    const double q = __hiloint2double((int)s2_rows_partner16((unsigned)__double2hiint(s1)), (int)s2_rows_partner16((unsigned)__double2loint(s1)));
    return s1 + q;
}
__device__ __forceinline__ float s2_row_sum(const float v) {
    const float s1 = v + __uint_as_float(s2_rows_partner32(__float_as_uint(v)));
    return s1 + __uint_as_float(s2_rows_partner16(__float_as_uint(s1)));
}

template <typename T>
__device__ __forceinline__ void s2_store_gran(const __amdgpu_buffer_rsrc_t &r, const int idx, const T v, const unsigned tag) {
    unsigned long long bits;
    if (sizeof(T) == 8) bits = (unsigned long long)__double_as_longlong((double)v);
    else bits = (unsigned long long)__float_as_uint((float)v);
    u4 g;
    g.x = (unsigned)bits; g.y = (unsigned)(bits >> 32); g.z = tag; g.w = 0u;
    __builtin_amdgcn_raw_buffer_store_b128(g, r, idx * (int)sizeof(XGran), 0, 16);   // write-through: device scope
}
template <typename T>
__device__ __forceinline__ T s2_gran_value(const u4 &g) {
    if (sizeof(T) == 8) return (T)__longlong_as_double((long long)(((unsigned long long)g.y << 32) | g.x));
    return (T)__uint_as_float(g.x);
}

// xg: x_j granules, [block][row][q]; pg: helper partial sums, [ord * H + h][row][q]; both validated by `epoch`.
template <typename T, int NR, bool LOWER>
__global__ __launch_bounds__(S2_THREADS) void trsv2_kernel(int n, int H, const T *__restrict__ LU, int lda,
                                                          const T *__restrict__ inv128T, const T *__restrict__ Rhs,
                                                          T *__restrict__ Out, int ldo, int nout, XGran *xg, XGran *pg,
                                                          unsigned epoch, int *status, int spin_limit,
                                                          unsigned long long *dbg) {
    // dbg != nullptr (development, LSX_S2_DBG): 100 MHz stamps of the owner of every block row, 8 per row
#define S2_STAMP(k) if (dbg && owner) dbg[ord * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
    __shared__ T xs[2][SB * NR];          // x_j of the current / next step, [row][q]
    __shared__ T bs[SB * NR];             // owner: the finished right-hand side of its block
    __shared__ T ps[SB * NR];             // owner: sum of the helpers' partial sums
    __shared__ int s_fail;
    const int NB = (n + SB - 1) / SB;
    const int tid = threadIdx.x;
    const int ord = (int)blockIdx.x / H, h = (int)blockIdx.x % H;
    const int i = LOWER ? ord : NB - 1 - ord;             // block row
    auto src = [&](const int s) __attribute__((always_inline)) { return LOWER ? s : NB - 1 - s; };   // step -> source block
    // my steps: s = first, first + stride, ... < limit
    int first, stride, limit;
    if (h == 0) { first = (H == 1) ? 0 : ord - 1; stride = 1; limit = ord; }
    else { first = h - 1; stride = H - 1; limit = ord - 1; }
    if (first < 0) first = limit;                          // ord == 0: no steps
    const bool owner = h == 0;
    if (!owner && first >= limit) return;                  // a helper without blocks: the owner does not wait for it
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(xg, 0, NB * SB * NR * (int)sizeof(XGran), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_p = __builtin_amdgcn_make_buffer_rsrc(pg, 0, NB * H * SB * NR * (int)sizeof(XGran), 0x00020000);
    // (Tried: x_i a second time by plain stores, polled by the next owner beside the write-through copy -- with H = 8
    // all owners sit on one XCD.  The hop got LONGER, 0.93 instead of 0.76 us: the extra loads per poll cost more than
    // the L2-local copy gains.)
    if (tid == 0) s_fail = 0;
    __syncthreads();

    // helpers of this block row that have blocks: h' = 1 .. nhelp; the one that holds the block two steps back (step
    // ord - 2) is LATE -- its partial sums appear about when the owner's own last x_j does, and are polled with it
    const int nhelp = (H > 1 && ord >= 2) ? min(H - 1, ord - 1) : 0;
    const int h_late = nhelp > 0 ? 1 + (ord - 2) % (H - 1) : 0;
    if (tid < 64) {
        // ---------------- the polling wave
        const int lane = tid;
        T psum[2][NR];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < NR; ++q) psum[t][q] = T(0);
        auto poll_partial = [&](const int hh) __attribute__((always_inline)) {   // blocking; adds helper hh's sums to psum
            int spins = 0;
            for (;;) {
                u4 g[2][NR];
                const int oz = opaque_zero();
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < NR; ++q)
                        g[t][q] = __builtin_amdgcn_raw_buffer_load_b128(r_p, (((ord * H + hh) * SB + lane + 64 * t) * NR + q) * (int)sizeof(XGran), oz, 16);
                bool ok = true;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < NR; ++q) ok &= g[t][q].z == epoch;
                if (!__any(!ok)) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < NR; ++q) psum[t][q] += s2_gran_value<T>(g[t][q]);
                    return;
                }
                if (s_fail || ++spins > spin_limit) { s_fail = 1; return; }
            }
        };
        if (lane == 0) S2_STAMP(0);
        if (owner)   // the early helpers, in the order of h: a fixed order of summation, and all of them long there
            for (int hh = 1; hh <= nhelp; ++hh)
                if (hh != h_late) poll_partial(hh);
        if (lane == 0) S2_STAMP(1);
        for (int s = first; s < limit; s += stride) {
            const int j = src(s);
            // the owner's last step: the late helper's sums travel in the same shots as x_j (they are there by now, or
            // within the hop; a poll of their own behind x_j's would put a second round trip on the chain)
            // (1 or 2 right-hand sides; with more the two sets of granules do not fit the registers, the steps are
            // longer anyway, and the late sums are waited for in front of x_j)
            const bool last_own = owner && s + stride >= limit && h_late != 0;
            const bool with_late = NR <= 2 && last_own;
            if (NR > 2 && last_own) poll_partial(h_late);
            T val[2][NR], pv[2][NR <= 2 ? NR : 1];
            int spins = 0;
            for (;;) {
                u4 g[2][NR], gp[2][NR <= 2 ? NR : 1];
                const int oz = opaque_zero();
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < NR; ++q)
                        g[t][q] = __builtin_amdgcn_raw_buffer_load_b128(r_x, ((j * SB + lane + 64 * t) * NR + q) * (int)sizeof(XGran), oz, 16);
                bool ok = true;
                if constexpr (NR <= 2) {
                    if (with_late) {
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int q = 0; q < NR; ++q)
                                gp[t][q] = __builtin_amdgcn_raw_buffer_load_b128(r_p, (((ord * H + h_late) * SB + lane + 64 * t) * NR + q) * (int)sizeof(XGran), oz, 16);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int q = 0; q < NR; ++q) ok &= gp[t][q].z == epoch;
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < NR; ++q) ok &= g[t][q].z == epoch;
                if (!__any(!ok)) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < NR; ++q) {
                            val[t][q] = s2_gran_value<T>(g[t][q]);
                            if constexpr (NR <= 2) pv[t][q] = with_late ? s2_gran_value<T>(gp[t][q]) : T(0);
                        }
                    break;
                }
                if (s_fail || ++spins > spin_limit) {
                    s_fail = 1;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < NR; ++q) {
                            val[t][q] = T(0);
                            if constexpr (NR <= 2) pv[t][q] = T(0);
                        }
                    break;
                }
            }
            if (lane == 0 && s + stride >= limit) S2_STAMP(2);
            const int buf = ((s - first) / stride) & 1;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < NR; ++q) xs[buf][(lane + 64 * t) * NR + q] = val[t][q];
            if (owner && s + stride >= limit) {   // everything the compute waves need goes out under ONE barrier
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        T add = T(0);
                        if constexpr (NR <= 2) add = pv[t][q];
                        ps[(lane + 64 * t) * NR + q] = psum[t][q] + add;
                    }
            }
            __syncthreads();   // A(s): x_j is in LDS
        }
        if (owner) __syncthreads();   // B: bs is written
        if (s_fail && tid == 0) atomicExch(status, 1);
        return;
    }

    // ---------------- the compute waves: thread = (row 16 w + (l & 15), column range l >> 4)
    const int c = tid - 64, w = c >> 6, l = c & 63;
    const int rr = l & 15, part = l >> 4;
    const int row_l = 16 * w + rr;
    const int grow = i * SB + row_l;
    typedef T v2t __attribute__((ext_vector_type(2)));
    const bool vec_ok = (((size_t)LU % 16) == 0) && (lda % (16 / (int)sizeof(T)) == 0);
    auto load_block = [&](const int j, T (&blk)[32]) __attribute__((always_inline)) {
        const int col0 = j * SB + 32 * part;
        const T *p = LU + (size_t)(grow < n ? grow : 0) * lda + col0;
        if (vec_ok && (i + 1) * SB <= n && (j + 1) * SB <= n) {
            if (sizeof(T) == 8) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const v2t v = *(const v2t *)(p + 2 * t);
                    blk[2 * t] = v[0]; blk[2 * t + 1] = v[1];
                }
            } else {
                typedef T v4t __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const v4t v = *(const v4t *)(p + 4 * t);
                    blk[4 * t] = v[0]; blk[4 * t + 1] = v[1]; blk[4 * t + 2] = v[2]; blk[4 * t + 3] = v[3];
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 32; ++t) blk[t] = (grow < n && col0 + t < n) ? p[t] : T(0);
        }
    };
    T acc[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) acc[q] = T(0);
    auto apply = [&](const int buf, const T (&blk)[32]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const T xa = xs[buf][(32 * part + rr) * NR + q], xb = xs[buf][(32 * part + 16 + rr) * NR + q];
            dpp_fma32<T, (NR <= 2)>(acc[q], xa, xb, blk);
        }
    };

    if (!owner) {
        // ---- helper: stream my blocks, one register set in flight under the wait for the next x_j
        T blkA[32], blkB[32];
        load_block(src(first), blkA);
        int k = 0;
        for (int s = first; s < limit; s += 2 * stride, k += 2) {
            const bool more1 = s + stride < limit;
            if (more1) load_block(src(s + stride), blkB);
            __syncthreads();   // A(s)
            apply(k & 1, blkA);
            if (more1) {
                if (s + 2 * stride < limit) load_block(src(s + 2 * stride), blkA);
                __syncthreads();   // A(s + stride)
                apply((k + 1) & 1, blkB);
            }
        }
        // the four column ranges of a row sit in the four 16-lane rows of one wave: two cross-row exchanges, no LDS array
        // and no barrier; the lanes of the first range store
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const T v = s2_row_sum(acc[q]);
            if (part == 0) s2_store_gran<T>(r_p, ((int)blockIdx.x * SB + row_l) * NR + q, v, epoch);
        }
        return;
    }

    // ---- owner: inv(T_ii) in registers (transposed in memory: inv128T[c][r]), the block next to the diagonal
    T inv[32];
    {
        const T *ip = inv128T + (size_t)i * SB * SB + row_l;
#pragma unroll
        for (int t = 0; t < 32; ++t) inv[t] = ip[(size_t)(32 * part + t) * SB];
    }
    T rhs[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) rhs[q] = (part == 0 && grow < n) ? Rhs[(size_t)grow * NR + q] : T(0);
    {
        T blk[32];
        if (first < limit) load_block(src(first), blk);
        int k = 0;
        for (int s = first; s < limit; s += stride, ++k) {
            __syncthreads();   // A(s)
            if (c == 0 && s + stride >= limit) S2_STAMP(3);
            apply(k & 1, blk);
            if (s + stride < limit) load_block(src(s + stride), blk);   // H == 1 only: the owner streams its whole row
        }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const T v = s2_row_sum(acc[q]);
        if (part == 0) bs[row_l * NR + q] = rhs[q] - v - (nhelp > 0 ? ps[row_l * NR + q] : T(0));
    }
    __syncthreads();   // B
    if (c == 0) S2_STAMP(4);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        T v0 = T(0);
        const T xa = bs[(32 * part + rr) * NR + q], xb = bs[(32 * part + 16 + rr) * NR + q];
        dpp_fma32<T, (NR <= 2)>(v0, xa, xb, inv);
        const T v = s2_row_sum(v0);
        if (part == 0) {
            const T vv = grow < n ? v : T(0);
            s2_store_gran<T>(r_x, grow * NR + q, vv, epoch);
            if (grow < n && q < nout) Out[(size_t)grow * ldo + q] = v;
        }
    }
    if (c == 0) S2_STAMP(5);
#undef S2_STAMP
}

// One launch in front of the sweeps: blocks [0, 2 NB128): inv(T_kk) of every 128 x 128 diagonal block of L and of U
// from the 64 x 64 block inverses (merge128 above); the blocks behind them: the interchange list as a permutation and
// P * B gathered into Bp (n x NR, columns >= nrhs zero).  The permutation: position k is final after step k, and
//     perm[k] = content of position ipiv[k] just before step k,
//     content of position v before step `bound` = content of position c before step c for the LAST step c < bound with
//     ipiv[c] == v, or row v if there is none (kernels_misc.hip: perm_chase_kernel);
// the steps that target a position are found through buckets built with LDS atomics (count, exclusive scan, fill) --
// O(n) work in one workgroup instead of the n^2 / 2 compares of perm_index_kernel (21.6 us of a 204 us solve at 4096).
template <typename T>
__global__ __launch_bounds__(256) void solve_prep_kernel(int n, int nrhs, int NR, const T *__restrict__ LU, int lda,
                                                         const T *__restrict__ inv64L, const T *__restrict__ inv64U,
                                                         T *__restrict__ inv128L, T *__restrict__ inv128U,
                                                         const int32_t *__restrict__ ipiv, const T *__restrict__ B, int ldb,
                                                         T *__restrict__ Bp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NB = (n + VB - 1) / VB;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < 2 * NB) {
        const int lower = (int)blockIdx.x < NB ? 1 : 0;
        const int blk = lower ? (int)blockIdx.x : (int)blockIdx.x - NB;
        const T *inv64 = lower ? inv64L : inv64U;
        T *out128 = lower ? inv128L : inv128U;
        T *X11 = (T *)smem, *X22 = X11 + VH * VLD, *OFF = X22 + VH * VLD, *W = OFF + VH * VLD;
        const int lane = tid & 63, wave = tid >> 6;
        const int r0 = blk * VB;
        const int nb64 = (n + VH - 1) / VH;
        {   // all 48 loads of a thread in flight at once (one iteration at a time they were 16 serial memory latencies:
            // most of the 27 us this launch took at first)
            constexpr int NE = VH * VH / 256;
            T v11[NE], v22[NE], vof[NE];
#pragma unroll
            for (int t = 0; t < NE; ++t) {
                const int e = tid + 256 * t, ii = e / VH, jj = e % VH;
                v11[t] = inv64[(size_t)(2 * blk) * VH * VH + e];
                v22[t] = (2 * blk + 1 < nb64) ? inv64[(size_t)(2 * blk + 1) * VH * VH + e] : (ii == jj ? T(1) : T(0));
                const int gi = lower ? r0 + VH + ii : r0 + ii, gj = lower ? r0 + jj : r0 + VH + jj;
                vof[t] = (gi < n && gj < n) ? LU[(size_t)gi * lda + gj] : T(0);
            }
#pragma unroll
            for (int t = 0; t < NE; ++t) {
                const int e = tid + 256 * t, ii = e / VH, jj = e % VH;
                X11[ii * VLD + jj] = v11[t];
                X22[ii * VLD + jj] = v22[t];
                OFF[ii * VLD + jj] = vof[t];
            }
        }
        __syncthreads();
        if (lower) v_gemm64<T>(OFF, X11, W, T(1), wave, lane);
        else v_gemm64<T>(OFF, X22, W, T(1), wave, lane);
        __syncthreads();
        if (lower) v_gemm64<T>(X22, W, OFF, T(-1), wave, lane);
        else v_gemm64<T>(X11, W, OFF, T(-1), wave, lane);
        __syncthreads();
        T *out = out128 + (size_t)blk * VB * VB;
        for (int e = tid; e < VB * VB; e += 256) {
            const int cc = e / VB, r = e % VB;  // out[c][r] = inv[r][c]
            T v;
            if (r < VH && cc < VH) v = X11[r * VLD + cc];
            else if (r >= VH && cc >= VH) v = X22[(r - VH) * VLD + cc - VH];
            else if (lower) v = (r >= VH) ? OFF[(r - VH) * VLD + cc] : T(0);
            else v = (r < VH) ? OFF[r * VLD + cc - VH] : T(0);
            out[e] = v;
        }
        return;
    }
    // ---- the permutation block
    int *piv = (int *)smem, *cnt = piv + n, *start = cnt + n, *bucket = start + n;   // 4 n ints (host checks the size)
    __shared__ int s_part[256];
    for (int k = tid; k < n; k += 256) {
        const int p = ipiv[k];
        piv[k] = (p > k && p < n) ? p : k;     // anything else is no interchange (a self-swap, or not a valid list)
        cnt[k] = 0;
    }
    __syncthreads();
    for (int k = tid; k < n; k += 256)
        if (piv[k] != k) atomicAdd(&cnt[piv[k]], 1);
    __syncthreads();
    // exclusive scan of cnt -> start: every thread a contiguous chunk, then the chunk sums
    const int chunk = (n + 255) / 256;
    const int lo = tid * chunk, hi = min(n, lo + chunk);
    int sum = 0;
    for (int k = lo; k < hi; ++k) sum += cnt[k];
    {   // exclusive scan of the 256 chunk sums: inside each wave by shuffles, then the four wave totals
        int inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d *= 2) {
            const int o = __shfl_up(inc, d, 64);
            if ((tid & 63) >= d) inc += o;
        }
        if ((tid & 63) == 63) s_part[tid >> 6] = inc;
        __syncthreads();
        int base = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv) base += s_part[wv];
        __syncthreads();
        s_part[tid] = base + inc - sum;
    }
    __syncthreads();
    {
        int run = s_part[tid];
        for (int k = lo; k < hi; ++k) { const int v = cnt[k]; start[k] = run; run += v; cnt[k] = 0; }
    }
    __syncthreads();
    for (int k = tid; k < n; k += 256)
        if (piv[k] != k) bucket[start[piv[k]] + atomicAdd(&cnt[piv[k]], 1)] = k;
    __syncthreads();
    // every permutation block built the same buckets (cheap: a few passes over n ints); each chases its share of the rows
    const int pw = (int)gridDim.x - 2 * NB, pb = (int)blockIdx.x - 2 * NB;
    for (int k = pb * 256 + tid; k < n; k += 256 * pw) {
        int v = piv[k], bound = k;
        for (int hops = 0; hops < n; ++hops) {     // terminates: bound strictly decreases
            int best = -1;
            const int b0 = start[v], b1 = b0 + cnt[v];
            for (int e = b0; e < b1; ++e) {
                const int cc = bucket[e];
                if (cc < bound && cc > best) best = cc;
            }
            if (best < 0) break;
            v = best;
            bound = best;
        }
        for (int q = 0; q < NR; ++q) Bp[(size_t)k * NR + q] = q < nrhs ? B[(size_t)v * ldb + q] : T(0);
    }
}

// Returns 1 when the shape is outside what this form serves (the caller takes the older path).
template <typename T>
int lu_solve_few_rhs2(lsx_handle_t h, int n, int nrhs, int nr, const T *LU, int lda, const int32_t *d_ipiv, T *B, int ldb,
                      T *inv64L, T *inv64U, T *inv128L, T *inv128U, T *Bp, T *Y) {
    const int NB = (n + SB - 1) / SB;
    const size_t shm_merge = (size_t)4 * VH * VLD * sizeof(T);
    const size_t shm = shm_merge > (size_t)n * 16 ? shm_merge : (size_t)n * 16;   // the permutation block: 4 n ints
    if (n <= SB || NB > h->num_cu || shm > 150 * 1024) return 1;
    int H = 1;
    while (2 * H * NB <= h->num_cu && H < 8) H *= 2;     // every workgroup resident at once: one per CU (9 waves of 168 registers)
    if (nr >= 8 && H > 4) H = 4;                           // 8 right-hand sides: the helpers' partial sums are 8 x the traffic
    const size_t xbytes = (size_t)NB * SB * nr * sizeof(XGran), pbytes = xbytes * H;
    const size_t need = 2 * (xbytes + pbytes) + 256;
    if (need > h->xchg_bytes) {
        // a fresh, zeroed area: tags of another life of the memory must not pass for this handle's epochs
        LSX_HIP(hipStreamSynchronize(h->stream));
        if (h->xchg) (void)hipFree(h->xchg);
        h->xchg = nullptr; h->xchg_bytes = 0;
        LSX_HIP(hipMalloc(&h->xchg, need));
        LSX_HIP(hipMemsetAsync(h->xchg, 0, need, h->stream));
        h->xchg_bytes = need;
        h->xchg_epoch = 0;
    }
    if (++h->xchg_epoch == 0u) {   // wrapped: start over on a cleared area
        LSX_HIP(hipMemsetAsync(h->xchg, 0, h->xchg_bytes, h->stream));
        h->xchg_epoch = 1;
    }
    ProfScope ps(h, LSX_PROF_TRSM, 2.0 * n * (double)n * nrhs, 2.0 * sizeof(T) * n * (double)n);
    XGran *xgL = (XGran *)((char *)h->xchg + 256), *pgL = (XGran *)((char *)xgL + xbytes);
    XGran *xgU = (XGran *)((char *)pgL + pbytes), *pgU = (XGran *)((char *)xgU + xbytes);
    int *status = h->dev_status + 1;
    LSX_TRY(launch_trtri_both<T>(h, n, LU, lda, inv64L, inv64U));
    LSX_HIP(hipFuncSetAttribute((const void *)solve_prep_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    const int nb128 = (n + VB - 1) / VB;
    int perm_wgs = n >= 2048 ? 8 : (n + 255) / 256;
    if (const char *e = getenv("LSX_S2_PERM_WGS")) perm_wgs = atoi(e);   // development: timing of the two roles
    hipLaunchKernelGGL(solve_prep_kernel<T>, dim3(2 * nb128 + perm_wgs), dim3(256), shm, h->stream, n, nrhs, nr, LU, lda,
                       (const T *)inv64L, (const T *)inv64U, inv128L, inv128U, d_ipiv, (const T *)B, ldb, Bp);
    // development: LSX_S2_DBG=1 -> owner stamps in the handle's scratch (tools/solve_stamps.py reads them back)
    unsigned long long *dbgL = nullptr, *dbgU = nullptr;
    if (getenv("LSX_S2_DBG") && h->scratch_bytes >= (size_t)NB * 128 + 4096) {
        dbgL = (unsigned long long *)h->scratch;
        dbgU = dbgL + (size_t)NB * 8;
    }
#define S2_LAUNCH(NRV)                                                                                                    \
    hipLaunchKernelGGL((trsv2_kernel<T, NRV, true>), dim3(NB * H), dim3(S2_THREADS), 0, h->stream, n, H, LU, lda,          \
                       (const T *)inv128L, (const T *)Bp, Y, NRV, NRV, xgL, pgL, h->xchg_epoch, status, h->spin_limit, dbgL); \
    hipLaunchKernelGGL((trsv2_kernel<T, NRV, false>), dim3(NB * H), dim3(S2_THREADS), 0, h->stream, n, H, LU, lda,         \
                       (const T *)inv128U, (const T *)Y, B, ldb, nrhs, xgU, pgU, h->xchg_epoch, status, h->spin_limit, dbgU)
    switch (nr) {
        case 1: S2_LAUNCH(1); break;
        case 2: S2_LAUNCH(2); break;
        case 4: S2_LAUNCH(4); break;
        case 8: S2_LAUNCH(8); break;
        default: set_error("lu_solve_few_rhs2: nrhs must be padded to 1, 2, 4 or 8"); return LSX_ERR_ARG;
    }
#undef S2_LAUNCH
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

template int lu_solve_few_rhs2<double>(lsx_handle_t, int, int, int, const double *, int, const int32_t *, double *, int,
                                       double *, double *, double *, double *, double *, double *);
template int lu_solve_few_rhs2<float>(lsx_handle_t, int, int, int, const float *, int, const int32_t *, float *, int, float *,
                                      float *, float *, float *, float *, float *);

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
