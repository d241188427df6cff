// Supporting kernels: deterministic fills, row interchanges, triangular block
// inverses + block solves, determinant reduction, permutation helpers.
#include "common.h"
#include <type_traits>

namespace lsx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T>
struct MfmaS;
template <>
struct MfmaS<double> {
    typedef d4 acc_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct MfmaS<float> {
    typedef f4 acc_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// ------------------------------------------------------------------ fill
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename T>
__global__ void fill_kernel(int kind, uint64_t seed, int m, int n, T *A, int lda, int row_off,
                            int col_off) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n || i >= m) return;
    const uint64_t h = splitmix64(seed * 0x9E3779B97F4A7C15ull + ((uint64_t)(uint32_t)(i + row_off) << 32) +
                                  (uint64_t)(uint32_t)(j + col_off));
    double v;
    if (kind == LSX_FILL_INT5)
        v = (double)(int)(h % 11ull) - 5.0;  // integers -5..5 (random_matrix.py:104)
    else
        v = (double)(h >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;  // uniform [-1,1)
    A[(size_t)i * lda + j] = (T)v;
}

template <typename T>
int launch_fill(lsx_handle_t h, int kind, uint64_t seed, int m, int n, T *A, int lda, int row_off,
                int col_off) {
    if (m <= 0 || n <= 0) return LSX_OK;
    hipLaunchKernelGGL(fill_kernel<T>, dim3((n + 255) / 256, m), dim3(256), 0, h->stream, kind, seed,
                       m, n, A, lda, row_off, col_off);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// ------------------------------------------------------------------ laswp
// Applies the jb sequential interchanges (row0+k <-> ipiv[k]) to a column
// chunk as ONE gather: every affected destination row finds its source by
// walking the interchange list backwards (O(jb) per row, all rows in
// parallel), the chunk's source rows are staged in LDS, then written out.
// HBM traffic: each affected row segment read once and written once.
// (Reference: the O(1) list swap at linalg.py:552.)
template <typename T, int CW>
__global__ __launch_bounds__(256) void laswp_kernel(int ncols, T *__restrict__ A, int lda, int row0,
                                                    int jb, const int32_t *__restrict__ ipiv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *s_piv = (int *)smem;             // jb
    int *s_dst = s_piv + 256;             // 2*jb
    int *s_src = s_dst + 512;             // 2*jb
    T *tile = (T *)(smem + (256 + 512 + 512) * sizeof(int));  // [2*jb][CW]
    const int tid = threadIdx.x;
    for (int k = tid; k < jb; k += 256) s_piv[k] = ipiv[k];
    __syncthreads();
    const int nd = 2 * jb;
    for (int d = tid; d < nd; d += 256) {
        const int dst = d < jb ? row0 + d : s_piv[d - jb];
        int x = dst;
        for (int k = jb - 1; k >= 0; --k) {
            const int t = row0 + k, p = s_piv[k];
            x = (x == t) ? p : ((x == p) ? t : x);
        }
        // rows listed twice (d >= jb duplicates) keep only their first listing
        bool dup = false;
        if (d >= jb) {
            if (dst < row0 + jb) dup = true;  // already covered by the top block
            for (int e = 0; e < d - jb && !dup; ++e) dup = (s_piv[e] == dst);
        }
        s_dst[d] = (dup || x == dst) ? -1 : dst;
        s_src[d] = x;
    }
    __syncthreads();
    const int c0 = blockIdx.x * CW;
    const int tc = tid % CW, tr = tid / CW;
    constexpr int RP = 256 / CW;  // rows per pass
    const bool cok = c0 + tc < ncols;
    for (int d = tr; d < nd; d += RP)
        if (s_dst[d] >= 0 && cok) tile[d * CW + tc] = A[(size_t)s_src[d] * lda + c0 + tc];
    __syncthreads();
    for (int d = tr; d < nd; d += RP)
        if (s_dst[d] >= 0 && cok) A[(size_t)s_dst[d] * lda + c0 + tc] = tile[d * CW + tc];
}

template <typename T>
int launch_laswp(lsx_handle_t h, int ncols, T *A, int lda, int row0, int jb, const int32_t *d_ipiv) {
    if (ncols <= 0 || jb <= 0) return LSX_OK;
    if (jb > 256) {
        set_error("laswp: jb %d > 256", jb);
        return LSX_ERR_ARG;
    }
    constexpr int CW = 32;
    ProfScope ps(h, LSX_PROF_LASWP, 0, 4.0 * sizeof(T) * jb * (double)ncols);
    const size_t shm = (256 + 512 + 512) * sizeof(int) + (size_t)2 * jb * CW * sizeof(T);
    if (shm > 48 * 1024)
        LSX_HIP(hipFuncSetAttribute((const void *)laswp_kernel<T, CW>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((laswp_kernel<T, CW>), dim3((ncols + CW - 1) / CW), dim3(256), shm, h->stream,
                       ncols, A, lda, row0, jb, d_ipiv);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// Gather-list form: the cooperative panel kernel already knows every row's final position,
// so the column chunks only have to move data (no per-workgroup replay of the interchanges).
// Register-staged: a thread first issues ALL its loads (up to 32 independent 8-byte loads in
// flight), the workgroup synchronises (a row can be source and destination of different moves),
// then everything is stored; a workgroup owns a 32-column chunk for every move, so no other
// workgroup touches these columns.  Each row segment is a 256-byte run.
// VW = elements per lane (1, or 16 bytes' worth: 2 in fp64, 4 in fp32): with VW > 1 ncols, lda, hole_at and hole_w
// are counted in GROUPS of VW columns and A is read as such groups (the caller checks divisibility and alignment)
template <typename T, int CW, int VW>
__device__ __forceinline__ void laswp_moves_body(int bx, int ncols, T *__restrict__ A_, int lda, int row0,
                                                 const int2 *__restrict__ moves, int hole_at, int hole_w) {
    typedef T vw_t __attribute__((ext_vector_type(VW > 1 ? VW : 2)));
    typedef typename std::conditional<VW == 1, T, vw_t>::type vt;   // a 1-wide vector type ends up in scratch
    vt *__restrict__ A = (vt *)A_;
    __shared__ int s_dst[256], s_src[256];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    __syncthreads();
    {   // compact the valid moves (order is irrelevant: destinations are distinct)
        const int2 mv = moves[tid];
        if (mv.x >= 0) {
            const int slot = atomicAdd(&s_n, 1);
            s_dst[slot] = mv.x;
            s_src[slot] = mv.y;
        }
    }
    __syncthreads();
    const int nmv = s_n;
    const int c0 = bx * CW;
    const int tc = tid % CW, tr = tid / CW;
    constexpr int RP = 256 / CW;        // rows per pass
    constexpr int NP = 256 / RP;        // passes (max moves / RP)
    const bool cok = c0 + tc < ncols;
    // logical column -> matrix column: the hole_w columns at hole_at (the panel itself, already in
    // final order) are skipped, so the columns left and right of a panel take ONE launch
    const int col = c0 + tc + ((c0 + tc >= hole_at) ? hole_w : 0);
    vt v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int d = tr + RP * i;
        if (cok && d < nmv) v[i] = A[(size_t)(row0 + s_src[d]) * lda + col];
    }
    // a row may be the source of one move and the destination of another: every load of the
    // chunk must have returned before any thread of the workgroup stores
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int d = tr + RP * i;
        if (cok && d < nmv) A[(size_t)(row0 + s_dst[d]) * lda + col] = v[i];
    }
}

template <typename T, int CW, int VW>
__global__ __launch_bounds__(256) void laswp_moves_kernel(int ncols, T *__restrict__ A, int lda, int row0,
                                                          const int2 *__restrict__ moves, int hole_at,
                                                          int hole_w) {
    LSX_TS(8);
    laswp_moves_body<T, CW, VW>(blockIdx.x, ncols, A, lda, row0, moves, hole_at, hole_w);
}

// all columns of an n-column matrix except the hole_w columns starting at hole_at, one launch
template <typename T>
int launch_laswp_moves_around(lsx_handle_t h, int n, T *A, int lda, int row0, int hole_at, int hole_w) {
    const int ncols = n - hole_w;
    if (ncols <= 0) return LSX_OK;
    ProfScope ps(h, LSX_PROF_LASWP, 0, 4.0 * sizeof(T) * 128 * (double)ncols);
    constexpr int VW = 16 / (int)sizeof(T);   // 16 bytes per lane: the matrix as groups of 2 (fp64) or 4 (fp32) columns
    if (((size_t)A % 16 == 0) && lda % VW == 0 && n % VW == 0 && hole_at % VW == 0 && hole_w % VW == 0) {
        constexpr int CW = 32 / VW;   // the same 32 columns per workgroup, so the grid still covers the chip
        hipLaunchKernelGGL((laswp_moves_kernel<T, CW, VW>), dim3((ncols / VW + CW - 1) / CW), dim3(256), 0, h->stream,
                           ncols / VW, A, lda / VW, row0, (const int2 *)h->moves, hole_at / VW, hole_w / VW);
        LSX_HIP(hipGetLastError());
        return LSX_OK;
    }
    constexpr int CW = 32;
    hipLaunchKernelGGL((laswp_moves_kernel<T, CW, 1>), dim3((ncols + CW - 1) / CW), dim3(256), 0, h->stream,
                       ncols, A, lda, row0, (const int2 *)h->moves, hole_at, hole_w);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// Every panel's interchanges on the columns LEFT of it, for the whole factorisation in one launch (look-ahead
// driver with the XCD-scope panel: nothing may be launched beside a running panel, see api.hip, so the
// left-hand interchanges no longer trail each step).  Columns are independent: a workgroup owns a 32-column chunk
// and applies the gather lists of all panels right of it in factorisation order.  lists[s] (256 int2) belongs to
// the panel that starts at row/column k0 + s * nb.
template <typename T, int CW, int VW>
__global__ __launch_bounds__(256) void laswp_left_all_kernel(T *__restrict__ A, int lda, int k0, int nb, int nsteps,
                                                             const int2 *__restrict__ lists) {
    LSX_TS(9);
    const int c_end = (blockIdx.x + 1) * CW * VW;   // first column right of this chunk
    for (int s = 0; s < nsteps; ++s) {
        const int ks = k0 + s * nb;
        if (ks < c_end) continue;                    // the chunk is not entirely left of panel s (uniform)
        laswp_moves_body<T, CW, VW>(blockIdx.x, ks / VW, A, lda, ks, lists + 256 * s, 0x7fffffff, 0);
        __syncthreads();                             // the body's index arrays are rewritten next trip
    }
}

// columns [0, k_last) where k_last = k0 + (nsteps - 1) * nb is the last panel's first column
template <typename T>
int launch_laswp_left_all(lsx_handle_t h, T *A, int lda, int k0, int nb, int nsteps, const void *lists) {
    const int k_last = k0 + (nsteps - 1) * nb;
    if (nsteps <= 0 || k_last <= 0) return LSX_OK;
    ProfScope ps(h, LSX_PROF_LASWP, 0, 2.0 * sizeof(T) * 128 * (double)k_last * nsteps);
    constexpr int VW = 16 / (int)sizeof(T);
    if (((size_t)A % 16 == 0) && lda % VW == 0 && k0 % 32 == 0 && nb % 32 == 0) {
        constexpr int CW = 32 / VW;
        hipLaunchKernelGGL((laswp_left_all_kernel<T, CW, VW>), dim3((k_last + 31) / 32), dim3(256), 0, h->stream, A,
                           lda / VW, k0, nb, nsteps, (const int2 *)lists);
    } else {
        if (k0 % 32 || nb % 32) { set_error("laswp_left_all: panel starts must be multiples of 32"); return LSX_ERR_INTERNAL; }
        hipLaunchKernelGGL((laswp_left_all_kernel<T, 32, 1>), dim3((k_last + 31) / 32), dim3(256), 0, h->stream, A, lda,
                           k0, nb, nsteps, (const int2 *)lists);
    }
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// One small workgroup that waits (bounded) until *word >= target: the look-ahead driver puts it in front of the
// XCD-scope panel so that the trailing update launched beside it has had its workgroups dealt to the panel's XCD
// (where they leave at once) BEFORE the panel's workgroups fill that XCD's CUs -- a kernel launched once they
// are resident could not finish before the panel does, because an eighth of its workgroups is dealt to that XCD.
__global__ __launch_bounds__(64) void gate_kernel(const int *word, int target, int limit) {
    LSX_TS(7);
    if (threadIdx.x == 0) {
        for (int i = 0; i < limit; ++i) {
            if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
}
int launch_gate(lsx_handle_t h, const int *word, int target) {
    // ~0.25 us per poll: give up after ~0.5 ms (the update was not launched at all, or is far behind)
    hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, h->stream, word, target, 2000);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// One wave that waits until *word >= target and, unlike the gate, must not give up quietly: the look-ahead driver
// orders the panel chain behind the update's tile column 0 with it (gemm_sub_queue_kernel counts that column's
// finished tiles), in place of an event behind the whole update.  A time-out (~seconds) is recorded in `status`.
// and, as for a panel whose exchange timed out, as a negative value in the factorisation's info word: whatever is
// computed behind a time-out works on stale columns, and info < 0 is what every caller already checks.
__global__ __launch_bounds__(64) void wait_count_kernel(const int *word, int target, int limit, int *status, int *info) {
    LSX_TS(7);
    if (threadIdx.x == 0) {
        for (int i = 0; i < limit; ++i) {
            if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
            __builtin_amdgcn_s_sleep(4);
        }
        if (status) atomicMax(status, 1);
        if (info) atomicMin(info, -0x40000000);
    }
}
int launch_wait_count(lsx_handle_t h, const int *word, int target) {
    hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(64), 0, h->stream, word, target, h->chain_wait_limit, h->dev_status,
                       h->chain_info);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// the same for a plain column range
template <typename T>
int launch_laswp_moves(lsx_handle_t h, int ncols, T *A, int lda, int row0) {
    return launch_laswp_moves_around<T>(h, ncols, A, lda, row0, 0, 0);
}

// ------------------------------------------------------------------ triangular 64-block inverse
// One workgroup inverts one 64 x 64 diagonal block of a unit-lower (lower=1) or
// non-unit upper (lower=0) triangle: 16 x 16 blocks by substitution, then two
// levels of 2x2 block merges  X21 = -X22 * (T21 * X11)  on the MFMA pipe,
// everything LDS-resident.  Rows/columns past jb are treated as identity.
// The substitution below spells its multiply-adds as explicit fused operations.  Written as `s -= a * b` the
// contraction is hipcc's choice per inlining context, and it chose differently for the fp32 body inside
// chain_head_kernel (lower = 1 constant-propagated) than inside trtri64_kernel: both inverses were accurate to
// ~2e-6 but not the same bits, which is what the look-ahead driver's bit-identity tests caught in round 1
// (tools/chk_chain_head.py; test_chain_head_fused_equals_separate).
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

constexpr int TB = 64;        // diagonal block edge
constexpr int TLD = TB + 2;   // padded LDS leading dimension

template <typename T>
__device__ void lds_gemm_tile(int M, int N, int K, const T *A, int lda, const T *B, int ldb, T *D,
                              int ldd, T alpha, int wave, int nwaves, int lane) {
    typedef typename MfmaS<T>::acc_t acc_t;
    const int lc = lane & 15, lq = lane >> 4;
    const int tn = N / 16, nt = (M / 16) * tn;
    for (int tile = wave; tile < nt; tile += nwaves) {
        const int i0 = (tile / tn) * 16, j0 = (tile % tn) * 16;
        acc_t acc = {0, 0, 0, 0};
        for (int k0 = 0; k0 < K; k0 += 4)
            acc = MfmaS<T>::mma(A[(i0 + lc) * lda + k0 + lq], B[(k0 + lq) * ldb + j0 + lc], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) D[(i0 + MfmaS<T>::crow(lane, r)) * ldd + j0 + lc] = alpha * acc[r];
    }
}

// lower = 0 / 1: that triangle; lower = 2: blockIdx.y = 0 inverts the lower, 1 the upper triangle, the
// upper inverses going to Tinv + upper_off (both triangles of an LU in one launch)
template <typename T>
__device__ __forceinline__ void trtri64_work(int bx, int by, int lower, int jb, const T *__restrict__ Tm,
                                             int ldt, T *__restrict__ Tinv, size_t upper_off, T *X, T *W);
template <typename T>
__device__ __forceinline__ void trtri64_body(int bx, int by, int lower, int jb, const T *__restrict__ Tm,
                                             int ldt, T *__restrict__ Tinv, size_t upper_off) {
    // 16-byte aligned: the fast paths move pairs of T, and inside the fused chain-head kernel the arrays follow
    // other LDS variables
    __shared__ __attribute__((aligned(16))) T X[TB * TLD];   // triangle in, inverse out
    __shared__ __attribute__((aligned(16))) T W[32 * TLD];   // merge temporary
    trtri64_work<T>(bx, by, lower, jb, Tm, ldt, Tinv, upper_off, X, W);
}
// the same with the two LDS arrays handed in (X: TB * TLD, W: 32 * TLD elements, 16-byte aligned): the fused chain
// kernel carves them out of its dynamic allocation, which the block-solve workgroups of the same launch use otherwise
template <typename T>
__device__ __forceinline__ void trtri64_work(int bx, int by, int lower, int jb, const T *__restrict__ Tm,
                                             int ldt, T *__restrict__ Tinv, size_t upper_off, T *X, T *W) {
    if (lower == 2) {
        lower = by == 0;
        if (!lower) Tinv += upper_off;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = bx * TB;
    // load (identity outside the matrix / outside the triangle)
    typedef T v2t __attribute__((ext_vector_type(2)));
    const bool fast = (b0 + TB <= jb) && ((size_t)Tm % 16 == 0) && (ldt % 2 == 0) && ((size_t)Tinv % 16 == 0);
    if (fast) {   // complete block: 8 sixteen-byte loads per thread, all in flight, masked afterwards
        v2t r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = 2 * (tid + 256 * q);
            r[q] = *(const v2t *)(Tm + (size_t)(b0 + e / TB) * ldt + b0 + e % TB);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = 2 * (tid + 256 * q);
            const int i = e / TB, j = e % TB;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const bool inside = lower ? (j + d < i) : (j + d >= i);
                X[i * TLD + j + d] = inside ? r[q][d] : ((i == j + d) ? T(1) : T(0));
            }
        }
    } else {
        for (int e = tid; e < TB * TB; e += 256) {
            const int i = e / TB, j = e % TB;
            const int gi = b0 + i, gj = b0 + j;
            T v = (i == j) ? T(1) : T(0);
            if (gi < jb && gj < jb) {
                if (lower) {
                    if (j < i) v = Tm[(size_t)gi * ldt + gj];
                } else {
                    if (j >= i) v = Tm[(size_t)gi * ldt + gj];
                }
            }
            X[i * TLD + j] = v;
        }
    }
    __syncthreads();
    // 16 x 16 diagonal blocks: thread (blk, col) solves one column by substitution
    if (tid < 64) {
        const int blk = tid >> 4, c = tid & 15, o = blk * 16;
        T x[16];
        if (lower) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                T s = (i == c) ? T(1) : T(0);
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (t < i) s = fma_t(-X[(o + i) * TLD + o + t], (t >= c) ? x[t] : T(0), s);
                x[i] = (i >= c) ? s : T(0);  // unit diagonal
            }
        } else {
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) {
                const int i = 15 - ii;
                T s = (i == c) ? T(1) : T(0);
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (t > i) s = fma_t(-X[(o + i) * TLD + o + t], (t <= c) ? x[t] : T(0), s);
                x[i] = (i <= c) ? s / X[(o + i) * TLD + o + i] : T(0);
            }
        }
        // single wave: a lane's x[i] depends on every read of row i, so all lanes' reads of a row are issued before any
        // lane's write to it (LDS serves a wave's operations in order)
#pragma unroll
        for (int i = 0; i < 16; ++i) X[(o + i) * TLD + o + c] = x[i];
    }
    __syncthreads();
    // merges: s = 16 -> 32 -> 64.  lower: X21 = -X22 * (L21 * X11); upper: X12 = -X11 * (U12 * X22)
    for (int s = 16; s < TB; s *= 2) {
        for (int o = 0; o < TB; o += 2 * s) {
            const T *P11 = X + o * TLD + o;
            const T *P22 = X + (o + s) * TLD + o + s;
            T *Off = lower ? X + (o + s) * TLD + o : X + o * TLD + o + s;
            if (lower)
                lds_gemm_tile<T>(s, s, s, Off, TLD, P11, TLD, W, TLD, T(1), wave, 4, lane);
            else
                lds_gemm_tile<T>(s, s, s, Off, TLD, P22, TLD, W, TLD, T(1), wave, 4, lane);
            __syncthreads();
            if (lower)
                lds_gemm_tile<T>(s, s, s, P22, TLD, W, TLD, Off, TLD, T(-1), wave, 4, lane);
            else
                lds_gemm_tile<T>(s, s, s, P11, TLD, W, TLD, Off, TLD, T(-1), wave, 4, lane);
            __syncthreads();
        }
    }
    if (fast) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = 2 * (tid + 256 * q);
            v2t x;
            x[0] = X[(e / TB) * TLD + e % TB];
            x[1] = X[(e / TB) * TLD + e % TB + 1];
            *(v2t *)(Tinv + (size_t)bx * TB * TB + e) = x;
        }
        return;
    }
    for (int e = tid; e < TB * TB; e += 256) {
        const int i = e / TB, j = e % TB;
        Tinv[(size_t)bx * TB * TB + e] = X[i * TLD + j];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void trtri64_kernel(int lower, int jb, const T *__restrict__ Tm,
                                                      int ldt, T *__restrict__ Tinv, size_t upper_off) {
    trtri64_body<T>(blockIdx.x, blockIdx.y, lower, jb, Tm, ldt, Tinv, upper_off);
}

// Head of the look-ahead chain, one launch: the inverses of panel k's unit-lower diagonal blocks (workgroups
// 0 .. ntri-1) and panel k's interchanges on the next panel's column block (the others) are independent.
// wait_word != nullptr: every workgroup first waits (bounded, time-out -> *status) until *wait_word >= wait_target --
// the count of finished tiles of the previous update's column 0 (gemm_sub_queue_kernel), i.e. the columns this
// launch interchanges: the wait_count_kernel folded in, one launch less between two panels.
template <typename T, int CW, int VW>
__global__ __launch_bounds__(256) void chain_head_kernel(int ntri, int jb, const T *__restrict__ Tm, int ldt,
                                                         T *__restrict__ Tinv, int ncols, T *__restrict__ A, int lda,
                                                         int row0, const int2 *__restrict__ moves,
                                                         const int *wait_word, int wait_target, int wait_limit,
                                                         int *status, int *info) {
    LSX_TS(2);
    if (wait_word) {
        if (threadIdx.x == 0) {
            bool ok = false;
            for (int i = 0; i < wait_limit && !ok; ++i) {
                ok = __hip_atomic_load(wait_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= wait_target;
                if (!ok) __builtin_amdgcn_s_sleep(2);
            }
            // a time-out must reach the caller like a panel's: the columns behind it are stale (info < 0)
            if (!ok && status) atomicMax(status, 1);
            if (!ok && info && blockIdx.x == 0) atomicMin(info, -0x40000000);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);   // device scope: the counted tiles' stores are visible from here on
        }
        __syncthreads();
    }
    if ((int)blockIdx.x < ntri)
        trtri64_body<T>(blockIdx.x, 0, 1, jb, Tm, ldt, Tinv, 0);
    else
        laswp_moves_body<T, CW, VW>(blockIdx.x - ntri, ncols, A, lda, row0, moves, 0x7fffffff, 0);
}

// trtri(lower) of the jb x jb triangle at Tm  +  the gather-list interchanges on `ncols` columns at A.
// Returns 1 when the shapes do not allow the 16-byte path (the caller then issues the two launches).
template <typename T>
int launch_chain_head(lsx_handle_t h, int jb, const T *Tm, int ldt, T *Tinv, int ncols, T *A, int lda, int row0,
                      const int *wait_word, int wait_target) {
    constexpr int VW = 16 / (int)sizeof(T);
    constexpr int CW = 32 / VW;
    if (!h->moves_valid || jb <= 0 || ncols <= 0 || ((size_t)A % 16) || lda % VW || ncols % VW) return 1;
    const int ntri = (jb + TB - 1) / TB;
    ProfScope ps(h, LSX_PROF_TRSM);
    hipLaunchKernelGGL((chain_head_kernel<T, CW, VW>), dim3(ntri + (ncols / VW + CW - 1) / CW), dim3(256), 0, h->stream,
                       ntri, jb, Tm, ldt, Tinv, ncols / VW, A, lda / VW, row0, (const int2 *)h->moves, wait_word, wait_target,
                       h->chain_wait_limit, h->dev_status, h->chain_info);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}


// Diagnostics (tests/test_gpu_parity.py: the fused chain head against the separate launches): block inverses of
// the unit-lower triangle at Tm through chain_head_kernel with an all-void gather list on `ncols` columns of A.
template <typename T>
int diag_chain_head(lsx_handle_t h, int jb, const T *Tm, int ldt, T *Tinv, int ncols, T *A, int lda, int row0,
                    const void *moves) {
    constexpr int VW = 16 / (int)sizeof(T);
    constexpr int CW = 32 / VW;
    if (jb <= 0 || ncols <= 0 || ((size_t)A % 16) || lda % VW || ncols % VW) return LSX_ERR_ARG;
    const int ntri = (jb + TB - 1) / TB;
    hipLaunchKernelGGL((chain_head_kernel<T, CW, VW>), dim3(ntri + (ncols / VW + CW - 1) / CW), dim3(256), 0, h->stream,
                       ntri, jb, Tm, ldt, Tinv, ncols / VW, A, lda / VW, row0, (const int2 *)moves, (const int *)nullptr, 0,
                       0, (int *)nullptr, (int *)nullptr);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}
template int diag_chain_head<float>(lsx_handle_t, int, const float *, int, float *, int, float *, int, int, const void *);
template int diag_chain_head<double>(lsx_handle_t, int, const double *, int, double *, int, double *, int, int, const void *);

template <typename T>
int launch_trtri(lsx_handle_t h, int lower, int jb, const T *Tm, int ldt, T *Tinv) {
    if (jb <= 0) return LSX_OK;
    ProfScope ps(h, LSX_PROF_TRSM);
    hipLaunchKernelGGL(trtri64_kernel<T>, dim3((jb + TB - 1) / TB), dim3(256), 0, h->stream, lower,
                       jb, Tm, ldt, Tinv, (size_t)0);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// inverses of the 64 x 64 diagonal blocks of BOTH triangles of an LU factor, one launch
template <typename T>
int launch_trtri_both(lsx_handle_t h, int n, const T *LU, int lda, T *invL, T *invU) {
    if (n <= 0) return LSX_OK;
    ProfScope ps(h, LSX_PROF_TRSM);
    hipLaunchKernelGGL(trtri64_kernel<T>, dim3((n + TB - 1) / TB, 2), dim3(256), 0, h->stream, 2, n, LU, lda, invL,
                       (size_t)(invU - invL));
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// ------------------------------------------------------------------ block triangular solve
// B (jb x ncols) <- inv(Tri) * B for the jb x jb triangle at Tm, given the
// inverses of its 64 x 64 diagonal blocks (Tinv).  Each workgroup owns a chunk
// of CWT columns for ALL jb rows, so the block substitution needs no grid sync:
//   lower: for b = 0..nb-1:  B_b -= sum_{t<b} T_bt B_t ;  B_b = Tinv_b B_b
//   upper: for b = nb-1..0:  B_b -= sum_{t>b} T_bt B_t ;  B_b = Tinv_b B_b
// This is the U12 = L11^-1 A12 step of the blocked LU and the diagonal step of
// the blocked forward / backward solves (reference: the normalise + eliminate
// loops of linalg.py:569-596 and 611-621 restricted to the pivot block rows).
constexpr int CWT = 32;

template <typename T>
__global__ __launch_bounds__(256) void trsm_block_kernel(int lower, int jb, int ncols,
                                                         const T *__restrict__ Tm, int ldt,
                                                         const T *__restrict__ Tinv,
                                                         T *__restrict__ B, int ldb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nblk = (jb + TB - 1) / TB;
    const int BLD = CWT + 2;
    T *Bs = (T *)smem;                       // [nblk*TB][BLD]
    T *Ts = Bs + (size_t)nblk * TB * BLD;    // [TB][TLD] staged T_bt or Tinv_b
    T *Ws = Ts + TB * TLD;                   // [TB][BLD] product temporary
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * CWT;
    for (int e = tid; e < nblk * TB * CWT; e += 256) {
        const int i = e / CWT, j = e % CWT;
        Bs[i * BLD + j] = (i < jb && c0 + j < ncols) ? B[(size_t)i * ldb + c0 + j] : T(0);
    }
    __syncthreads();
    for (int bb = 0; bb < nblk; ++bb) {
        const int b = lower ? bb : nblk - 1 - bb;
        for (int tt = 0; tt < bb; ++tt) {
            const int t = lower ? tt : nblk - 1 - tt;
            // stage T_bt (64 x 64)
            for (int e = tid; e < TB * TB; e += 256) {
                const int i = e / TB, j = e % TB;
                const int gi = b * TB + i, gj = t * TB + j;
                Ts[i * TLD + j] = (gi < jb && gj < jb) ? Tm[(size_t)gi * ldt + gj] : T(0);
            }
            __syncthreads();
            lds_gemm_tile<T>(TB, CWT, TB, Ts, TLD, Bs + (size_t)t * TB * BLD, BLD, Ws, BLD, T(1), wave,
                             4, lane);
            __syncthreads();
            for (int e = tid; e < TB * CWT; e += 256) {
                const int i = e / CWT, j = e % CWT;
                Bs[(b * TB + i) * BLD + j] -= Ws[i * BLD + j];
            }
            __syncthreads();
        }
        for (int e = tid; e < TB * TB; e += 256) {
            const int i = e / TB, j = e % TB;
            Ts[i * TLD + j] = Tinv[(size_t)b * TB * TB + e];
        }
        __syncthreads();
        lds_gemm_tile<T>(TB, CWT, TB, Ts, TLD, Bs + (size_t)b * TB * BLD, BLD, Ws, BLD, T(1), wave, 4,
                         lane);
        __syncthreads();
        for (int e = tid; e < TB * CWT; e += 256) {
            const int i = e / CWT, j = e % CWT;
            Bs[(b * TB + i) * BLD + j] = Ws[i * BLD + j];
        }
        __syncthreads();
    }
    for (int e = tid; e < nblk * TB * CWT; e += 256) {
        const int i = e / CWT, j = e % CWT;
        if (i < jb && c0 + j < ncols) B[(size_t)i * ldb + c0 + j] = Bs[i * BLD + j];
    }
}

// Fast path of the block solve for jb <= 128 (two 64-row blocks): the three 64 x 64 blocks it needs
// (inv(T00), T10 or T01, inv(T11)) are fetched up front with all loads in flight at once, products
// stay in MFMA accumulators (no temporary tile), and B makes one trip through LDS.
template <typename T>
__device__ __forceinline__ void lds_gemm_regs(const T *A, int lda, const T *B, int ldb, typename MfmaS<T>::acc_t (&acc)[2],
                                              int wave, int lane) {
    // D (64 x 32) = A (64 x 64) * B (64 x 32): 4 x 2 tiles, wave w owns tile row w (2 tiles)
    const int lc = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        typename MfmaS<T>::acc_t a = {0, 0, 0, 0};
        for (int k0 = 0; k0 < 64; k0 += 4)
            a = MfmaS<T>::mma(A[(16 * wave + lc) * lda + k0 + lq], B[(k0 + lq) * ldb + 16 * t + lc], a);
        acc[t] = a;
    }
}

template <typename T>
__device__ __forceinline__ void trsm_block2_body(int bx, int lower, int jb, int ncols, const T *__restrict__ Tm, int ldt,
                                                 const T *__restrict__ Tinv, T *__restrict__ B, int ldb, char *smem);
template <typename T>
__global__ __launch_bounds__(256) void trsm_block2_kernel(int lower, int jb, int ncols,
                                                          const T *__restrict__ Tm, int ldt,
                                                          const T *__restrict__ Tinv, T *__restrict__ B,
                                                          int ldb) {
    LSX_TS(3);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    trsm_block2_body<T>(blockIdx.x, lower, jb, ncols, Tm, ldt, Tinv, B, ldb, smem);
}
template <typename T>
__device__ __forceinline__ void trsm_block2_body(int bx, int lower, int jb, int ncols, const T *__restrict__ Tm, int ldt,
                                                 const T *__restrict__ Tinv, T *__restrict__ B, int ldb, char *smem) {
    constexpr int BLD = CWT + 2;
    T *Bs = (T *)smem;              // [128][BLD]
    T *T0 = Bs + 128 * BLD;         // inverse of the first diagonal block to be applied
    T *T1 = T0 + TB * TLD;          // off-diagonal block
    T *T2 = T1 + TB * TLD;          // inverse of the second diagonal block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = bx * CWT;
    // order of the two 64-row blocks: lower = (0 then 1), upper = (1 then 0)
    const int b_first = lower ? 0 : 1, b_second = lower ? 1 : 0;
    typedef T v2t __attribute__((ext_vector_type(2)));
    const bool fast = (jb == 128) && (c0 + CWT <= ncols) && (((size_t)Tm | (size_t)Tinv | (size_t)B) % 16 == 0) &&
                      (ldt % 2 == 0) && (ldb % 2 == 0);
    if (fast) {
        // complete tile: all 32 sixteen-byte loads of a thread are in flight before the first LDS write
        v2t r0[8], r1[8], r2[8], rb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = 2 * (tid + 256 * q);          // element pair inside a 64 x 64 block
            const int i = e / TB, j = e % TB;
            r0[q] = *(const v2t *)(Tinv + (size_t)b_first * TB * TB + e);
            r2[q] = *(const v2t *)(Tinv + (size_t)b_second * TB * TB + e);
            r1[q] = *(const v2t *)(Tm + (size_t)(b_second * TB + i) * ldt + b_first * TB + j);
            const int eb = 2 * (tid + 256 * q);         // element pair inside the 128 x CWT chunk
            rb[q] = *(const v2t *)(B + (size_t)(eb / CWT) * ldb + c0 + eb % CWT);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = 2 * (tid + 256 * q);
            const int i = e / TB, j = e % TB;
            T0[i * TLD + j] = r0[q][0]; T0[i * TLD + j + 1] = r0[q][1];
            T2[i * TLD + j] = r2[q][0]; T2[i * TLD + j + 1] = r2[q][1];
            T1[i * TLD + j] = r1[q][0]; T1[i * TLD + j + 1] = r1[q][1];
            const int bi = e / CWT, bj = e % CWT;
            Bs[bi * BLD + bj] = rb[q][0]; Bs[bi * BLD + bj + 1] = rb[q][1];
        }
    } else {
        for (int e = tid; e < TB * TB; e += 256) {
            const int i = e / TB, j = e % TB;
            T0[i * TLD + j] = Tinv[(size_t)b_first * TB * TB + e];
            T2[i * TLD + j] = Tinv[(size_t)b_second * TB * TB + e];
            const int gi = b_second * TB + i, gj = b_first * TB + j;
            T1[i * TLD + j] = (gi < jb && gj < jb) ? Tm[(size_t)gi * ldt + gj] : T(0);
        }
        for (int e = tid; e < 128 * CWT; e += 256) {
            const int i = e / CWT, j = e % CWT;
            Bs[i * BLD + j] = (i < jb && c0 + j < ncols) ? B[(size_t)i * ldb + c0 + j] : T(0);
        }
    }
    __syncthreads();
    typename MfmaS<T>::acc_t acc[2];
    const int lc = lane & 15;
    T *B1 = Bs + (size_t)b_first * TB * BLD, *B2 = Bs + (size_t)b_second * TB * BLD;
    // X1 = inv(T11') * B1
    lds_gemm_regs<T>(T0, TLD, B1, BLD, acc, wave, lane);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) B1[(16 * wave + MfmaS<T>::crow(lane, r)) * BLD + 16 * t + lc] = acc[t][r];
    __syncthreads();
    // B2 -= T21 * X1
    lds_gemm_regs<T>(T1, TLD, B1, BLD, acc, wave, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) B2[(16 * wave + MfmaS<T>::crow(lane, r)) * BLD + 16 * t + lc] -= acc[t][r];
    __syncthreads();
    // X2 = inv(T22') * B2
    lds_gemm_regs<T>(T2, TLD, B2, BLD, acc, wave, lane);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) B2[(16 * wave + MfmaS<T>::crow(lane, r)) * BLD + 16 * t + lc] = acc[t][r];
    __syncthreads();
    if (fast) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = 2 * (tid + 256 * q);
            const int bi = e / CWT, bj = e % CWT;
            v2t x;
            x[0] = Bs[bi * BLD + bj]; x[1] = Bs[bi * BLD + bj + 1];
            *(v2t *)(B + (size_t)bi * ldb + c0 + bj) = x;
        }
        return;
    }
    for (int e = tid; e < 128 * CWT; e += 256) {
        const int i = e / CWT, j = e % CWT;
        if (i < jb && c0 + j < ncols) B[(size_t)i * ldb + c0 + j] = Bs[i * BLD + j];
    }
}

template <typename T>
int launch_trsm_block(lsx_handle_t h, int lower, int jb, int ncols, const T *Tm, int ldt,
                      const T *Tinv, T *B, int ldb) {
    if (jb <= 0 || ncols <= 0) return LSX_OK;
    const int nblk = (jb + TB - 1) / TB;
    const size_t shm = ((size_t)nblk * TB * (CWT + 2) + TB * TLD + TB * (CWT + 2)) * sizeof(T);
    if (shm > 160 * 1024) {
        set_error("trsm_block: jb %d too large for LDS", jb);
        return LSX_ERR_ARG;
    }
    ProfScope ps(h, LSX_PROF_TRSM, (double)jb * jb * ncols);
    if (jb > 64 && jb <= 128) {
        const size_t shm2 = ((size_t)128 * (CWT + 2) + 3 * TB * TLD) * sizeof(T);
        LSX_HIP(hipFuncSetAttribute((const void *)trsm_block2_kernel<T>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2));
        hipLaunchKernelGGL(trsm_block2_kernel<T>, dim3((ncols + CWT - 1) / CWT), dim3(256), shm2, h->stream,
                           lower, jb, ncols, Tm, ldt, Tinv, B, ldb);
        LSX_HIP(hipGetLastError());
        return LSX_OK;
    }
    if (shm > 48 * 1024)
        LSX_HIP(hipFuncSetAttribute((const void *)trsm_block_kernel<T>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(trsm_block_kernel<T>, dim3((ncols + CWT - 1) / CWT), dim3(256), shm, h->stream,
                       lower, jb, ncols, Tm, ldt, Tinv, B, ldb);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// The chain head AND the block solve of the next panel's column block in ONE launch (round 3: one launch, one event
// and ~7 us less between two panels).  Workgroups [0, 2): the two 64 x 64 block inverses of panel k (written to Tinv
// for the main stream's block solve, then counted in *ready behind a device-scope fence).  Workgroups [2, 6): one
// 32-column chunk of the next panel's columns each -- panel k's interchanges on the chunk (all moved rows), then, once
// both inverses are there, U12 of the chunk: the same three 64 x 64 x 32 MFMA products as trsm_block2_kernel, from the
// same operands, so the bits are those of the separate launches.  Lower block indices are dispatched first, so the
// consumers never wait for a workgroup that is not resident.  jb = ncols = 128 and the 16-byte paths only.
template <typename T, int CW, int VW>
__global__ __launch_bounds__(256) void chain_fused_kernel(int jb, const T *__restrict__ Tm, int ldt, T *__restrict__ Tinv,
                                                          T *__restrict__ A, int lda, int row0,
                                                          const int2 *__restrict__ moves, const int *wait_word, int wait_target,
                                                          int wait_limit, int *status, int *info, int *ready) {
    LSX_TS(2);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (wait_word) {
        if (threadIdx.x == 0) {
            bool ok = false;
            for (int i = 0; i < wait_limit && !ok; ++i) {
                ok = __hip_atomic_load(wait_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= wait_target;
                if (!ok) __builtin_amdgcn_s_sleep(2);
            }
            if (!ok && status) atomicMax(status, 1);
            if (!ok && info && blockIdx.x == 0) atomicMin(info, -0x40000000);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
    }
    constexpr int NTRI = 2;
    if ((int)blockIdx.x < NTRI) {
        T *X = (T *)smem, *W = X + TB * TLD;
        trtri64_work<T>(blockIdx.x, 0, 1, jb, Tm, ldt, Tinv, 0, X, W);
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            __hip_atomic_fetch_add(ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int chunk = (int)blockIdx.x - NTRI;
    laswp_moves_body<T, CW, VW>(chunk, 128 / VW, A, lda / VW, row0, moves, 0x7fffffff, 0);
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok = false;
        for (int i = 0; i < (1 << 22) && !ok; ++i)
            ok = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= NTRI;
        if (!ok) {
            if (status) atomicMax(status, 1);
            if (info) atomicMin(info, -0x40000000);
        }
    }
    __syncthreads();
    // device scope: the inverses of the other workgroups, and this workgroup's own interchanged rows (written through
    // to L2 by other threads of it; lines of those rows may sit in L1 from the gather), are read from L2
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    trsm_block2_body<T>(chunk, 1, jb, 128, Tm, ldt, Tinv, A + (size_t)row0 * lda, lda, smem);
}

// Returns 1 when the shapes do not allow it (the caller issues chain head and block solve separately).
template <typename T>
int launch_chain_fused(lsx_handle_t h, int jb, const T *Tm, int ldt, T *Tinv, int ncols, T *A, int lda, int row0,
                       const int *wait_word, int wait_target, int *ready) {
    constexpr int VW = 16 / (int)sizeof(T);
    constexpr int CW = 32 / VW;
    if (!h->chain_fused || !h->moves_valid || jb != 128 || ncols != 128 || ((size_t)A % 16) || lda % VW ||
        ((size_t)Tm % 16) || ((size_t)Tinv % 16) || ldt % 2 || !ready)
        return 1;
    const size_t shm = ((size_t)128 * (CWT + 2) + 3 * TB * TLD) * sizeof(T);
    ProfScope ps(h, LSX_PROF_TRSM, (double)jb * jb * ncols);
    LSX_HIP(hipFuncSetAttribute((const void *)chain_fused_kernel<T, CW, VW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((chain_fused_kernel<T, CW, VW>), dim3(2 + 128 / 32), dim3(256), shm, h->stream, jb, Tm, ldt, Tinv, A, lda,
                       row0, (const int2 *)h->moves, wait_word, wait_target, h->chain_wait_limit, h->dev_status, h->chain_info,
                       ready);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// ------------------------------------------------------------------ permutation helpers

// perm[i] = original row that ends at position i after all n interchanges.
// Each thread walks the interchange list backwards from its own position (the list is staged in LDS),
// so the conversion is O(n) deep and n-wide instead of an n-step serial chain.  Two facts keep it
// short: an interchange (t, p) has p >= t, so it cannot touch a walker whose position is below t --
// position i starts at step t = i, not n-1; four steps share one 16-byte LDS read; and a group of four
// that touches no lane of the wave costs five independent compares instead of an 8-deep select chain.  (At n = 4096 a
// plain full-length walk took 236 us: more than half of a one-right-hand-side solve.)
__global__ __launch_bounds__(256) void ipiv_to_perm_kernel(int n, const int32_t *__restrict__ ipiv,
                                                           int32_t *__restrict__ perm) {
    __shared__ __attribute__((aligned(16))) int s_piv[2048];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int imax = min(n - 1, (int)blockIdx.x * 256 + 255);   // highest position walked by this workgroup
    int x = i;
    for (int base = (imax / 2048) * 2048; base >= 0; base -= 2048) {
        const int cnt = min(2048, n - base);
        __syncthreads();
        // beyond the list: identity interchanges (t, t), which leave every walker where it is
        for (int k = threadIdx.x; k < 2048; k += 256) s_piv[k] = (k < cnt) ? ipiv[base + k] : base + k;
        __syncthreads();
        // eight steps per trip, the next trip's two 16-byte LDS reads already in flight
        const int top = min(2047, imax - base) | 7;
        int4 pa = *(const int4 *)&s_piv[top - 3], pb = *(const int4 *)&s_piv[top - 7];
        for (int k = top; k >= 7; k -= 8) {
            const int4 qa = pa, qb = pb;
            if (k >= 15) {
                pa = *(const int4 *)&s_piv[k - 11];
                pb = *(const int4 *)&s_piv[k - 15];
            }
            const int t = base + k;
            // the eight interchanges touch a walker only if it sits on one of their 16 rows: test that
            // with independent compares and skip the dependent chain when no lane of the wave is hit
            const bool hit = ((unsigned)(x - (t - 7)) < 8u) | (x == qa.w) | (x == qa.z) | (x == qa.y) | (x == qa.x) |
                             (x == qb.w) | (x == qb.z) | (x == qb.y) | (x == qb.x);
            if (!__any(hit)) continue;
            x = (x == t) ? qa.w : ((x == qa.w) ? t : x);
            x = (x == t - 1) ? qa.z : ((x == qa.z) ? t - 1 : x);
            x = (x == t - 2) ? qa.y : ((x == qa.y) ? t - 2 : x);
            x = (x == t - 3) ? qa.x : ((x == qa.x) ? t - 3 : x);
            x = (x == t - 4) ? qb.w : ((x == qb.w) ? t - 4 : x);
            x = (x == t - 5) ? qb.z : ((x == qb.z) ? t - 5 : x);
            x = (x == t - 6) ? qb.y : ((x == qb.y) ? t - 6 : x);
            x = (x == t - 7) ? qb.x : ((x == qb.x) ? t - 7 : x);
        }
    }
    if (i < n) perm[i] = x;
}

// D[i][:] = S[perm[i]][:]
template <typename T>
__global__ void gather_rows_kernel(int n, int ncols, const int32_t *__restrict__ perm,
                                   const T *__restrict__ S, int lds, T *__restrict__ D, int ldd) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < ncols) D[(size_t)i * ldd + j] = S[(size_t)perm[i] * lds + j];
}

// The same conversion without an O(n)-deep dependent chain.  Forward view: at step k the rows at
// positions k and p_k = ipiv[k] >= k swap; position k is final afterwards, so
//     perm[k] = content of position p_k just before step k,
// and the content of a position v just before step `bound` is the content position c had just before
// step c, where c is the LAST step < bound that targeted v (p_c == v) -- or row v itself if no step did.
// Two index arrays make every hop O(1) apart from skipping later steps that target the same position:
//     last_target[v] = largest c with p_c == v,       prev_same[c] = largest c' < c with p_c' == p_c.
// Building them is one pass of independent compares per entry (no loop-carried select chain across
// 16 dependent operations as in the walk above); the chase itself takes a few hops per row.
__global__ __launch_bounds__(256) void perm_index_kernel(int n, const int32_t *__restrict__ ipiv,
                                                         int32_t *__restrict__ last_target,
                                                         int32_t *__restrict__ prev_same) {
    __shared__ __attribute__((aligned(16))) int s_piv[2048];
    // four lanes per entry, each scanning every fourth 16-entry group; the quad's maximum is the answer
    const int idx = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
    const bool want_last = blockIdx.y == 0;     // y = 0: last_target[idx], y = 1: prev_same[idx]
    const int imax = min(n - 1, (int)blockIdx.x * 64 + 63);
    const int key = want_last ? idx : ((idx < n) ? ipiv[idx] : -2);   // the value looked for among p_c
    const int lim = want_last ? idx : idx - 1;                          // highest step that may count
    int best = -1;
    for (int base = 0; base <= imax; base += 2048) {
        const int cnt = min(2048, n - base);
        __syncthreads();
        for (int k = threadIdx.x; k < 2048; k += 256) s_piv[k] = (k < cnt) ? ipiv[base + k] : -1;
        __syncthreads();
        // 16 entries per trip: four independent 16-byte LDS reads in flight, then compares; ascending
        // order, the last match wins
        const int top = min(2047, imax - base) | 15;
        for (int k = 16 * sub; k <= top; k += 64) {
            const int4 p0 = *(const int4 *)&s_piv[k], p1 = *(const int4 *)&s_piv[k + 4];
            const int4 p2 = *(const int4 *)&s_piv[k + 8], p3 = *(const int4 *)&s_piv[k + 12];
            const int room = lim - (base + k);      // entries 0..room of this trip may count
            if (room < 0) break;
            int m = -1;                               // highest matching entry of the trip
            m = (p0.x == key) ? 0 : m;   m = (p0.y == key) ? 1 : m;   m = (p0.z == key) ? 2 : m;   m = (p0.w == key) ? 3 : m;
            m = (p1.x == key) ? 4 : m;   m = (p1.y == key) ? 5 : m;   m = (p1.z == key) ? 6 : m;   m = (p1.w == key) ? 7 : m;
            m = (p2.x == key) ? 8 : m;   m = (p2.y == key) ? 9 : m;   m = (p2.z == key) ? 10 : m;  m = (p2.w == key) ? 11 : m;
            m = (p3.x == key) ? 12 : m;  m = (p3.y == key) ? 13 : m;  m = (p3.z == key) ? 14 : m;  m = (p3.w == key) ? 15 : m;
            if (m >= 0) {
                if (m <= room) {
                    best = base + k + m;
                } else {   // rare: the trip straddles the limit -- redo it entry by entry
                    for (int e = 0; e <= room && e < 16; ++e)
                        if (s_piv[k + e] == key) best = base + k + e;
                }
            }
        }
    }
    best = max(best, __shfl_xor(best, 1, 64));
    best = max(best, __shfl_xor(best, 2, 64));
    if (idx < n && sub == 0) (want_last ? last_target : prev_same)[idx] = best;
}

__global__ __launch_bounds__(256) void perm_chase_kernel(int n, const int32_t *__restrict__ ipiv,
                                                         const int32_t *__restrict__ last_target,
                                                         const int32_t *__restrict__ prev_same,
                                                         int32_t *__restrict__ perm) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    int v = ipiv[k], bound = k;
    if (v < 0 || v >= n) { perm[k] = k; return; }   // not a valid interchange list: leave the row in place
    for (int hops = 0; hops < n; ++hops) {          // terminates: bound strictly decreases
        int c = last_target[v];
        while (c >= bound) c = prev_same[c];
        if (c < 0) break;
        v = c;
        bound = c;
    }
    perm[k] = v;
}

int launch_ipiv_to_perm(lsx_handle_t h, int n, const int32_t *d_ipiv, int32_t *d_perm) {
    if (n <= 0) return LSX_OK;
    const size_t need = 2 * sizeof(int32_t) * (size_t)n;
    if (h->scratch && h->scratch_bytes >= need) {
        // the two index arrays live in the handle's scratch (every later user of it is stream-ordered)
        int32_t *last_target = (int32_t *)h->scratch, *prev_same = last_target + n;
        hipLaunchKernelGGL(perm_index_kernel, dim3((n + 63) / 64, 2), dim3(256), 0, h->stream, n, d_ipiv, last_target,
                           prev_same);
        hipLaunchKernelGGL(perm_chase_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, d_ipiv, last_target,
                           prev_same, d_perm);
    } else {
        hipLaunchKernelGGL(ipiv_to_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, d_ipiv, d_perm);
    }
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

template <typename T>
int launch_gather_rows(lsx_handle_t h, int n, int ncols, const int32_t *d_perm, const T *S, int lds,
                       T *D, int ldd) {
    if (n <= 0 || ncols <= 0) return LSX_OK;
    ProfScope ps(h, LSX_PROF_OTHER);
    hipLaunchKernelGGL(gather_rows_kernel<T>, dim3((ncols + 255) / 256, n), dim3(256), 0, h->stream,
                       n, ncols, d_perm, S, lds, D, ldd);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// X (n x n) <- P (the row-permuted identity): X[i][perm[i]] = 1.  This is the
// right-hand side of the inverse, [A|I] at linalg.py:704-706, after pivoting.
// perm == nullptr: the identity itself
template <typename T>
__global__ void set_perm_identity_kernel(int n, const int32_t *__restrict__ perm, T *X, int ldx) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n) return;
    X[(size_t)i * ldx + j] = ((perm ? perm[i] : i) == j) ? T(1) : T(0);
}

// D[:, perm[c]] = S[:, c]: the column permutation that turns U^-1 L^-1 into U^-1 L^-1 P (structured inverse, api.hip).
// Reads are coalesced; the writes of a row stay inside that row's n * sizeof(T) bytes.
template <typename T>
__global__ __launch_bounds__(256) void scatter_cols_kernel(int n, const int32_t *__restrict__ perm, const T *__restrict__ S,
                                                           int lds, T *__restrict__ D, int ldd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const int pc = perm[c];
    for (int i = blockIdx.y; i < n; i += gridDim.y) D[(size_t)i * ldd + pc] = S[(size_t)i * lds + c];
}
template <typename T>
int launch_scatter_cols(lsx_handle_t h, int n, const int32_t *d_perm, const T *S, int lds, T *D, int ldd) {
    if (n <= 0) return LSX_OK;
    ProfScope ps(h, LSX_PROF_OTHER);
    hipLaunchKernelGGL(scatter_cols_kernel<T>, dim3((n + 255) / 256, n < 2048 ? n : 2048), dim3(256), 0, h->stream, n, d_perm, S, lds,
                       D, ldd);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

template <typename T>
int launch_set_identity_perm(lsx_handle_t h, int n, const int32_t *perm, T *X, int ldx) {
    if (n <= 0) return LSX_OK;
    ProfScope ps(h, LSX_PROF_OTHER);
    hipLaunchKernelGGL(set_perm_identity_kernel<T>, dim3((n + 255) / 256, n), dim3(256), 0, h->stream,
                       n, perm, X, ldx);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// ------------------------------------------------------------------ determinant
// det = sign * mant * 2^exp with mant in [0.5, 1): product of diag(U) kept as a
// (mantissa, exponent) pair so 8192 factors neither overflow nor underflow;
// sign also flips once per actual interchange.  One workgroup, tree reduction.
template <typename T>
__global__ __launch_bounds__(256) void det_kernel(int n, const T *__restrict__ LU, int lda,
                                                  const int32_t *__restrict__ ipiv, double *out) {
    __shared__ double s_m[256];
    __shared__ long long s_e[256];
    __shared__ int s_neg[256];
    __shared__ int s_zero[256];
    const int tid = threadIdx.x;
    double m = 1.0;
    long long e = 0;
    int neg = 0, zero = 0;
    for (int k = tid; k < n; k += 256) {
        double d = (double)LU[(size_t)k * lda + k];
        if (ipiv[k] != k) neg ^= 1;
        if (d == 0.0) { zero = 1; continue; }
        if (d < 0) { neg ^= 1; d = -d; }
        int ex;
        const double f = frexp(d, &ex);
        m *= f;
        e += ex;
        int ex2;
        m = frexp(m, &ex2);
        e += ex2;
    }
    s_m[tid] = m; s_e[tid] = e; s_neg[tid] = neg; s_zero[tid] = zero;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            int ex;
            s_m[tid] = frexp(s_m[tid] * s_m[tid + s], &ex);
            s_e[tid] += s_e[tid + s] + ex;
            s_neg[tid] ^= s_neg[tid + s];
            s_zero[tid] |= s_zero[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (s_zero[0]) {
            out[0] = 0.0; out[1] = 0.0; out[2] = 0.0;
        } else {
            out[0] = s_neg[0] ? -1.0 : 1.0;
            out[1] = s_m[0];
            out[2] = (double)s_e[0];
        }
    }
}

template <typename T>
int launch_det(lsx_handle_t h, int n, const T *LU, int lda, const int32_t *d_ipiv, double *d_out) {
    ProfScope ps(h, LSX_PROF_OTHER);
    hipLaunchKernelGGL(det_kernel<T>, dim3(1), dim3(256), 0, h->stream, n, LU, lda, d_ipiv, d_out);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// ------------------------------------------------------------------ conditioning probes
// out[0] = max |A_ij| over an m x n block (atomic max on the bit pattern of a
// non-negative double; out[0] must be zeroed first).
template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(int m, int n, const T *__restrict__ A, int lda,
                                                   double *out) {
    __shared__ double s[256];
    double v = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)m * n;
         e += (size_t)gridDim.x * 256) {
        const double a = fabs((double)A[(e / n) * (size_t)lda + (e % n)]);
        if (a > v) v = a;
    }
    s[threadIdx.x] = v;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(s[0]));
}

// out[1] = min_k |LU_kk| (one workgroup)
template <typename T>
__global__ __launch_bounds__(256) void diag_minabs_kernel(int n, const T *__restrict__ LU, int lda,
                                                          double *out) {
    __shared__ double s[256];
    double v = INFINITY;
    for (int k = threadIdx.x; k < n; k += 256) {
        const double a = fabs((double)LU[(size_t)k * lda + k]);
        if (!(a >= v)) v = a;  // NaN propagates as "smallest"
    }
    s[threadIdx.x] = v;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) {
            const double o = s[threadIdx.x + k];
            if (!(o >= s[threadIdx.x])) s[threadIdx.x] = o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[1] = s[0];
}

template <typename T>
int launch_amax(lsx_handle_t h, int m, int n, const T *A, int lda, double *d_out) {
    LSX_HIP(hipMemsetAsync(d_out, 0, sizeof(double), h->stream));
    if (m <= 0 || n <= 0) return LSX_OK;
    hipLaunchKernelGGL(amax_kernel<T>, dim3(512), dim3(256), 0, h->stream, m, n, A, lda, d_out);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

template <typename T>
int launch_diag_minabs(lsx_handle_t h, int n, const T *LU, int lda, double *d_out) {
    hipLaunchKernelGGL(diag_minabs_kernel<T>, dim3(1), dim3(256), 0, h->stream, n, LU, lda, d_out);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// ------------------------------------------------------------------ copies
template <typename T>
__global__ void copy2d_kernel(int m, int n, const T *__restrict__ S, int lds, T *__restrict__ D,
                              int ldd) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n && i < m) D[(size_t)i * ldd + j] = S[(size_t)i * lds + j];
}

template <typename T>
int launch_copy2d(lsx_handle_t h, int m, int n, const T *S, int lds, T *D, int ldd) {
    if (m <= 0 || n <= 0) return LSX_OK;
    hipLaunchKernelGGL(copy2d_kernel<T>, dim3((n + 255) / 256, m), dim3(256), 0, h->stream, m, n, S,
                       lds, D, ldd);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

// ------------------------------------------------------------------ mixed-precision refinement (fp32 factors)
// r = b - A x with A, b in fp32 and x, the products and the sum in fp64; the residual goes back in fp32 for the
// next correction solve.  One wave per row, lanes stride the columns (coalesced 256-byte pieces of the row), up to
// 8 right-hand sides per pass.  HBM-bound: A is read once per sweep (4 n^2 bytes).
__global__ __launch_bounds__(256) void resid_mixed_kernel(int n, int nrhs, const float *__restrict__ A, int lda,
                                                          const float *__restrict__ B, int ldb,
                                                          const double *__restrict__ X, int ldx,
                                                          float *__restrict__ R, int ldr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *a = A + (size_t)row * lda;
    for (int c0 = 0; c0 < nrhs; c0 += 8) {
        double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = lane; k < n; k += 64) {
            const double av = (double)a[k];
            const double *x = X + (size_t)k * ldx + c0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c0 + j < nrhs) part[j] = __builtin_fma(av, x[j], part[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double v = part[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && c0 + j < nrhs)
                R[(size_t)row * ldr + c0 + j] = (float)((double)B[(size_t)row * ldb + c0 + j] - v);
        }
    }
}
// X (fp64) <- D (fp32) [init] or X += D; Xf <- (float) X; out[0] = max|D|, out[1] = max|X| (as ordered ints: >= 0)
__global__ __launch_bounds__(256) void refine_apply_kernel(int n, int nrhs, int init, const float *__restrict__ D, int ldd,
                                                           double *__restrict__ X, int ldx, float *__restrict__ Xf,
                                                           int ldf, unsigned long long *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    double ad = 0, ax = 0;
    if (idx < n * nrhs) {
        const int i = idx / nrhs, j = idx % nrhs;
        const double d = (double)D[(size_t)i * ldd + j];
        const double x = init ? d : X[(size_t)i * ldx + j] + d;
        X[(size_t)i * ldx + j] = x;
        if (Xf) Xf[(size_t)i * ldf + j] = (float)x;
        ad = fabs(d);
        ax = fabs(x);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ad = fmax(ad, __shfl_down(ad, off, 64));
        ax = fmax(ax, __shfl_down(ax, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && out) {   // non-negative doubles order like their bit patterns
        atomicMax(&out[0], (unsigned long long)__double_as_longlong(ad));
        atomicMax(&out[1], (unsigned long long)__double_as_longlong(ax));
    }
}
int launch_resid_mixed(lsx_handle_t h, int n, int nrhs, const float *A, int lda, const float *B, int ldb, const double *X,
                       int ldx, float *R, int ldr) {
    hipLaunchKernelGGL(resid_mixed_kernel, dim3((n + 3) / 4), dim3(256), 0, h->stream, n, nrhs, A, lda, B, ldb, X, ldx, R, ldr);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}
int launch_refine_apply(lsx_handle_t h, int n, int nrhs, int init, const float *D, int ldd, double *X, int ldx, float *Xf,
                        int ldf, double *d_out2) {
    if (d_out2) LSX_HIP(hipMemsetAsync(d_out2, 0, 2 * sizeof(double), h->stream));
    hipLaunchKernelGGL(refine_apply_kernel, dim3((n * nrhs + 255) / 256), dim3(256), 0, h->stream, n, nrhs, init, D, ldd, X,
                       ldx, Xf, ldf, (unsigned long long *)d_out2);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

#define INST(T)                                                                                   \
    template int launch_fill<T>(lsx_handle_t, int, uint64_t, int, int, T *, int, int, int);       \
    template int launch_laswp<T>(lsx_handle_t, int, T *, int, int, int, const int32_t *);         \
    template int launch_laswp_moves<T>(lsx_handle_t, int, T *, int, int);                         \
    template int launch_laswp_left_all<T>(lsx_handle_t, T *, int, int, int, int, const void *);   \
    template int launch_laswp_moves_around<T>(lsx_handle_t, int, T *, int, int, int, int);        \
    template int launch_chain_head<T>(lsx_handle_t, int, const T *, int, T *, int, T *, int, int, const int *, int);  \
    template int launch_chain_fused<T>(lsx_handle_t, int, const T *, int, T *, int, T *, int, int, const int *, int, int *); \
    template int launch_trtri<T>(lsx_handle_t, int, int, const T *, int, T *);                    \
    template int launch_trtri_both<T>(lsx_handle_t, int, const T *, int, T *, T *);               \
    template int launch_trsm_block<T>(lsx_handle_t, int, int, int, const T *, int, const T *, T *, \
                                      int);                                                       \
    template int launch_set_identity_perm<T>(lsx_handle_t, int, const int32_t *, T *, int);       \
    template int launch_scatter_cols<T>(lsx_handle_t, int, const int32_t *, const T *, int, T *, int); \
    template int launch_det<T>(lsx_handle_t, int, const T *, int, const int32_t *, double *);     \
    template int launch_copy2d<T>(lsx_handle_t, int, int, const T *, int, T *, int);              \
    template int launch_gather_rows<T>(lsx_handle_t, int, int, const int32_t *, const T *, int, T *, int); \
    template int launch_amax<T>(lsx_handle_t, int, int, const T *, int, double *);                \
    template int launch_diag_minabs<T>(lsx_handle_t, int, const T *, int, double *);
INST(double)
INST(float)

}  // namespace lsx

LSX_TS_SETTER(misc)
