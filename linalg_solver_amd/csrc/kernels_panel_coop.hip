// Cooperative panel factorisation: ONE launch factors an m x jb panel (jb <= 128)
// with partial pivoting, instead of two launches per column.
//
// Reference loops covered: pivot search, row swap, scaling and elimination below
// the pivot for jb consecutive pivots (linalg_solver/linalg.py:548-596).
//
// Layout.  G = ceil(m / RB) workgroups of 256 threads, one per CU, all
// co-resident (G <= number of CUs).  Workgroup g owns panel rows
// [g*RB, (g+1)*RB), RB = 16*RT, and keeps them IN REGISTERS for the whole
// panel: thread (ty, tx) = (tid>>4, tid&15) holds rows {16*r + ty} x columns
// {8*tx + c}, an RT x 8 tile.  HBM traffic is the algorithmic minimum: every
// panel element is read once and written once.
//
// Per column j (the only cross-CU traffic is one small exchange):
//   A. threads owning column j put it in LDS and reduce to the workgroup's
//      candidate (largest |a| among rows not yet used as pivots).
//   B. the wave that holds the candidate row publishes it: 128-entry row +
//      16-byte header {|a|, row, epoch} with write-through (sc1) stores, header
//      after an `s_waitcnt vmcnt(0)` (guide G16 recipe R1, one storing wave).
//   C. wave 0 of every workgroup polls all G headers (one per lane, relaxed sc1
//      loads), all reduce to the same winner (largest |a|, lowest row on ties),
//      fetch the winner's row with sc1 loads, and put it in LDS.
//   D. every thread updates its tile: l = a[:,j]/pivot kept in place (unit-lower
//      L), a[:,c] -= l * u[c] for c > j (linalg.py:587-596).
// Rows never move during the loop ("implicit pivoting": used rows are frozen).
// The LAPACK interchange list ipiv and each row's final position are obtained by
// replaying the swaps on two jb-entry maps as the pivots are chosen; at the end
// every row is written straight to its final position.
//
// Every spin is bounded; a timeout sets *status and lets the grid drain.
#include "common.h"

namespace lsx {

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int PC_COLS = 128;   // column capacity (16 thread columns x 8)
constexpr int SPIN_LIMIT = 1 << 20;   // ~1 s of polling before a workgroup gives up

struct __attribute__((aligned(16))) XHdr {
    double val;          // |a| of the candidate, < 0: no candidate
    int idx;             // panel-local row of the candidate
    unsigned epochchk;   // (epoch << 16) | 16-bit fold of val's bits
};

__device__ __forceinline__ unsigned fold16(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    unsigned x = (unsigned)(b ^ (b >> 32));
    return (x ^ (x >> 16)) & 0xffffu;
}

template <typename T, int RT>
__global__ __launch_bounds__(256, 1) void panel_coop_kernel(int m, int jb, T *__restrict__ P, int ldp,
                                                            int row0, int col0,
                                                            int32_t *__restrict__ ipiv,
                                                            int *__restrict__ info, XHdr *hdr,
                                                            T *xrow, int *status) {
    constexpr int RB = 16 * RT;
    __shared__ T s_col[2][RB];
    __shared__ double s_cv[2][16];
    __shared__ int s_ci[2][16];
    __shared__ __attribute__((aligned(16))) T s_u[PC_COLS];
    __shared__ int s_win[4];      // [0]=winner workgroup, [1]=winner row (panel-local), [2]=valid
    __shared__ int s_topid[PC_COLS], s_postop[PC_COLS];
    __shared__ int s_order[RB];

    const int G = gridDim.x, g = blockIdx.x;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int lane = tid & 63, wave = tid >> 6;
    const int base = g * RB;

    // buffer descriptors for the exchange area (sc1 traffic only)
    __amdgpu_buffer_rsrc_t r_hdr =
        __builtin_amdgcn_make_buffer_rsrc(hdr, 0, 2 * G * (int)sizeof(XHdr), 0x00020000);
    __amdgpu_buffer_rsrc_t r_row =
        __builtin_amdgcn_make_buffer_rsrc(xrow, 0, 2 * G * PC_COLS * (int)sizeof(T), 0x00020000);

    // ---- load the slice (rows >= m and columns >= jb read as zero)
    T a[RT][8];
    const bool vec_ok = (jb == PC_COLS) && ((ldp * sizeof(T)) % 16 == 0) && (((size_t)P) % 16 == 0);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int gi = base + 16 * r + ty;
        const T *src = P + (size_t)gi * ldp + 8 * tx;
        if (gi < m && vec_ok) {
#pragma unroll
            for (int c = 0; c < 8; ++c) a[r][c] = src[c];
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) a[r][c] = (gi < m && 8 * tx + c < jb) ? src[c] : T(0);
        }
    }
    for (int t = tid; t < PC_COLS; t += 256) { s_topid[t] = t; s_postop[t] = t; }
    for (int t = tid; t < RB; t += 256) s_order[t] = -1;
    unsigned frozen = 0;  // bit r: local row 16*r+ty already used as a pivot (or outside the panel)
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (base + 16 * r + ty >= m) frozen |= 1u << r;
    bool failed = false;
    __syncthreads();

    for (int j = 0; j < jb; ++j) {
        const int par = j & 1;
        const int jt = j >> 3;
        const int jc = __builtin_amdgcn_readfirstlane(j & 7);
        // ---------------- A: column j to LDS + per-thread-row candidates
        if (tx == jt) {
            T colv[RT];
            switch (jc) {
#define PICK(k) case k: _Pragma("unroll") for (int r = 0; r < RT; ++r) colv[r] = a[r][k]; break;
                PICK(0) PICK(1) PICK(2) PICK(3) PICK(4) PICK(5) PICK(6) PICK(7)
#undef PICK
            }
            double bv = -1.0;
            int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                s_col[par][16 * r + ty] = colv[r];
                const double av = fabs((double)colv[r]);
                const int gi = base + 16 * r + ty;
                if (!((frozen >> r) & 1u) && (av > bv || (av == bv && gi < bi))) { bv = av; bi = gi; }
            }
            s_cv[par][ty] = bv;
            s_ci[par][ty] = bi;
        }
        __syncthreads();
        // every wave reduces the 16 thread-row candidates to the workgroup's candidate
        double wv = s_cv[par][lane & 15];
        int wi = s_ci[par][lane & 15];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double ov = __shfl_xor(wv, off, 64);
            const int oi = __shfl_xor(wi, off, 64);
            if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
        }
        const bool have = wv >= 0.0;
        const int cl = have ? wi - base : 0;          // slice-local row of the candidate
        const int cty = cl & 15;
        const int cr = __builtin_amdgcn_readfirstlane(cl >> 4);
        // ---------------- B: publish (the wave that holds the candidate row; wave 0 if none)
        const int pub_wave = have ? (cty >> 2) : 0;
        if (wave == pub_wave) {
            if (have && ty == cty) {
                T rowv[8];
                switch (cr) {
#define PICKR(k) case k: _Pragma("unroll") for (int c = 0; c < 8; ++c) rowv[c] = a[k < RT ? k : 0][c]; break;
                    PICKR(0) PICKR(1) PICKR(2) PICKR(3) PICKR(4) PICKR(5) PICKR(6) PICKR(7)
                    PICKR(8) PICKR(9) PICKR(10) PICKR(11) PICKR(12) PICKR(13) PICKR(14) PICKR(15)
#undef PICKR
                }
                if (G == 1) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) s_u[8 * tx + c] = rowv[c];
                } else {
                    const int off = ((par * G + g) * PC_COLS + 8 * tx) * (int)sizeof(T);
                    if (sizeof(T) == 8) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            u4 v;
                            const unsigned long long b0 = (unsigned long long)__double_as_longlong((double)rowv[2 * q]);
                            const unsigned long long b1 = (unsigned long long)__double_as_longlong((double)rowv[2 * q + 1]);
                            v.x = (unsigned)b0; v.y = (unsigned)(b0 >> 32); v.z = (unsigned)b1; v.w = (unsigned)(b1 >> 32);
                            __builtin_amdgcn_raw_buffer_store_b128(v, r_row, off + 16 * q, 0, 16);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            u4 v;
                            v.x = __float_as_uint((float)rowv[4 * q]); v.y = __float_as_uint((float)rowv[4 * q + 1]);
                            v.z = __float_as_uint((float)rowv[4 * q + 2]); v.w = __float_as_uint((float)rowv[4 * q + 3]);
                            __builtin_amdgcn_raw_buffer_store_b128(v, r_row, off + 16 * q, 0, 16);
                        }
                    }
                }
            }
            if (G > 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // row is out before the header
                if (lane == (have ? ((cty & 3) * 16) : 0)) {
                    const double hv = have ? wv : -1.0;
                    const unsigned long long vb = (unsigned long long)__double_as_longlong(hv);
                    u4 h;
                    h.x = (unsigned)vb; h.y = (unsigned)(vb >> 32);
                    h.z = (unsigned)(have ? wi : -1);
                    h.w = ((unsigned)(j + 1) << 16) | fold16(hv);
                    __builtin_amdgcn_raw_buffer_store_b128(h, r_hdr, (par * G + g) * (int)sizeof(XHdr), 0, 16);
                }
            }
        }
        if (G == 1) {
            if (tid == 0) { s_win[0] = 0; s_win[1] = have ? wi : -1; s_win[2] = have ? 1 : 0; }
        }
        // ---------------- C: wave 0 gathers all candidates, picks the winner, fetches its row
        if (wave == 0 && G > 1) {
            double bv = -2.0;
            int bi = 0x7fffffff, bg = 0;
            for (int q = lane; q < G && !failed; q += 64) {
                int spins = 0;
                for (;;) {
                    const u4 h = __builtin_amdgcn_raw_buffer_load_b128(r_hdr, (par * G + q) * (int)sizeof(XHdr), 0, 16);
                    const double hv = __longlong_as_double((long long)(((unsigned long long)h.y << 32) | h.x));
                    if ((h.w >> 16) == (unsigned)(j + 1) && (h.w & 0xffffu) == fold16(hv)) {
                        const int hi = (int)h.z;
                        if (hi >= 0 && (hv > bv || (hv == bv && hi < bi))) { bv = hv; bi = hi; bg = q; }
                        break;
                    }
                    if (++spins > SPIN_LIMIT) { failed = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(bv, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                const int og = __shfl_xor(bg, off, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; bg = og; }
            }
            const bool valid = bv >= 0.0;
            if (valid) {
                // winner's row: 64 lanes x 16 B covers 128 fp64 (fp32: lanes 0..31)
                const int nl = (int)(PC_COLS * sizeof(T) / 16);
                if (lane < nl) {
                    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(
                        r_row, (par * G + bg) * PC_COLS * (int)sizeof(T) + 16 * lane, 0, 16);
                    *((u4 *)((char *)s_u + 16 * lane)) = v;
                }
            }
            if (lane == 0) { s_win[0] = bg; s_win[1] = valid ? bi : -1; s_win[2] = valid ? 1 : 0; }
            if (__any(failed)) {  // sticky and wave-uniform: later columns skip the polling
                failed = true;
                if (lane == 0) atomicExch(status, 1);
            }
        }
        __syncthreads();
        // ---------------- bookkeeping + D: update
        const int wrow = s_win[1];
        const bool valid = s_win[2] != 0;
        if (!valid) continue;  // no row left (cannot happen for m >= jb); uniform
        const T piv = s_u[j];
        if (tid == 0) {
            // replay the interchange (top position j <-> current position of the winner)
            const int c = wrow;
            const int p = (c < jb) ? s_postop[c] : c;
            const int d = s_topid[j];
            if (p != j) {
                s_topid[j] = c;
                if (p < jb) s_topid[p] = d;
                s_postop[d] = p;
                if (c < jb) s_postop[c] = j;
            }
            if (g == 0) {
                ipiv[j] = row0 + p;
                if (piv == T(0) && info && *info == 0) *info = col0 + j + 1;
            }
        }
        if (s_win[0] == g) {
            const int wl = wrow - base;
            if (ty == (wl & 15)) {
                frozen |= 1u << (wl >> 4);
                if (tx == 0) s_order[wl] = j;
            }
        }
        if (piv != T(0)) {
            const T rinv = T(1) / piv;
            T l[RT], u[8];
#pragma unroll
            for (int r = 0; r < RT; ++r)
                l[r] = ((frozen >> r) & 1u) ? T(0) : s_col[par][16 * r + ty] * rinv;
#pragma unroll
            for (int c = 0; c < 8; ++c) u[c] = (8 * tx + c > j) ? s_u[8 * tx + c] : T(0);
            if (tx == jt) {
                switch (jc) {
#define PUT(k) case k: _Pragma("unroll") for (int r = 0; r < RT; ++r) if (!((frozen >> r) & 1u)) a[r][k] = l[r]; break;
                    PUT(0) PUT(1) PUT(2) PUT(3) PUT(4) PUT(5) PUT(6) PUT(7)
#undef PUT
                }
            }
            if (tx >= jt) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) a[r][c] -= l[r] * u[c];
            }
        }
    }
    __syncthreads();
    // ---- every row straight to its final (LAPACK-order) position
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int lr = 16 * r + ty;
        const int gi = base + lr;
        if (gi >= m) continue;
        const int ord = s_order[lr];
        const int dest = ord >= 0 ? ord : (gi < jb ? s_postop[gi] : gi);
        T *dst = P + (size_t)dest * ldp + 8 * tx;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (8 * tx + c < jb) dst[c] = a[r][c];
    }
}

template <typename T>
int panel_cooperative(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0,
                      int32_t *d_ipiv, int *d_info) {
    constexpr int RT = 8, RB = 16 * RT;
    const int G = (m + RB - 1) / RB;
    if (jb > PC_COLS || G > h->num_cu) return 1;  // caller falls back to the per-column path
    // exchange area in scratch: status | headers[2][G] | rows[2][G][128]
    const size_t hdr_bytes = (size_t)2 * G * sizeof(XHdr);
    const size_t need = 256 + hdr_bytes + (size_t)2 * G * PC_COLS * sizeof(T);
    if (need > h->scratch_bytes) {
        set_error("panel_coop: scratch too small (%zu > %zu)", need, h->scratch_bytes);
        return LSX_ERR_INTERNAL;
    }
    int *status = (int *)h->scratch;
    XHdr *hdr = (XHdr *)((char *)h->scratch + 256);
    T *xrow = (T *)((char *)h->scratch + 256 + hdr_bytes);
    // headers (and the status word) are zeroed before EVERY launch: epoch 0 never matches
    LSX_HIP(hipMemsetAsync(h->scratch, 0, 256 + hdr_bytes, h->stream));
    hipLaunchKernelGGL((panel_coop_kernel<T, RT>), dim3(G), dim3(256), 0, h->stream, m, jb, P, ldp, row0,
                       col0, d_ipiv, d_info, hdr, xrow, status);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

template int panel_cooperative<double>(lsx_handle_t, int, int, double *, int, int, int, int32_t *, int *);
template int panel_cooperative<float>(lsx_handle_t, int, int, float *, int, int, int, int32_t *, int *);

}  // namespace lsx
