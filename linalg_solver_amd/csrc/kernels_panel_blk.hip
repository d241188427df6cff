// Blocked cooperative panel factorisation (panel mode 2): ONE launch per m x jb panel,
// ONE cross-CU hop per column.
//
// Same mathematics and the same bits as kernels_panel_coop.hip (partial pivoting, unit-lower
// L, reference loops linalg_solver/linalg.py:548-596); what changes is the data flow:
//
//   * the panel is processed in blocks of 8 columns = ONE thread column of the register tile,
//     so inside a block every row's active entries sit in one thread and the rank-1 updates
//     of the block are thread-local;
//   * a candidate's 8 block values travel INSIDE the polled record (64 B of values + a
//     16-byte trailer {epoch, row, 64-bit hash}; the hash makes the 80-byte record
//     self-validating against torn / stale reads), so once a workgroup has seen all G records
//     it knows the pivot row's block values without a second, dependent fetch;
//   * the other 120 columns are brought up to date once per block: each pivot row's owner
//     publishes the full row when it wins (off the critical path), and at the end of the block
//     every workgroup fetches the 8 rows, forms U12 = L11^-1 R by substitution and applies the
//     rank-8 update to its tile -- the same fused-multiply-adds in the same order as eight
//     rank-1 updates, so results are bit-identical to the unblocked kernels.
//
// Layout: G = ceil(m / RB) workgroups, RB = NT/16 * RT rows each (default 512 threads x 4 rows
// = 128 rows), one per CU and all co-resident; thread (ty, tx) = (tid >> 4, tid & 15) holds rows
// {NTY*r + ty} x columns {8*tx + c}.  Rows never move during the loop (implicit pivoting);
// LAPACK-order positions are replayed on two jb-entry maps and rows are written straight to
// their final places, as in the unblocked kernel.  Every spin is bounded (status word).
#include <type_traits>

#include "common.h"

namespace lsx {

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int PB_COLS = 128;
constexpr int PB_SPIN = 1 << 20;
constexpr int PB_REC = 256;  // bytes reserved per workgroup record (80 used): one line each

__device__ __forceinline__ unsigned long long pb_mix(unsigned long long h, unsigned long long v) {
    return ((h << 7) | (h >> 57)) ^ v;
}

template <typename T>
__device__ __forceinline__ unsigned long long pb_bits(T v) {
    if (sizeof(T) == 8) return (unsigned long long)__double_as_longlong((double)v);
    return (unsigned long long)__float_as_uint((float)v);
}
template <typename T>
__device__ __forceinline__ T pb_val(unsigned long long b) {
    if (sizeof(T) == 8) return (T)__longlong_as_double((long long)b);
    return (T)__uint_as_float((unsigned)b);
}

struct __attribute__((aligned(16))) PbGran {
    unsigned long long bits;
    unsigned tag;
    unsigned pad;
};

template <typename T, int RT, int NT, bool DBG>
__global__ __launch_bounds__(NT, NT / 256) void panel_blk_kernel(int m, int jb, T *__restrict__ P, int ldp,
                                                                 int row0, int col0,
                                                                 int32_t *__restrict__ ipiv,
                                                                 int *__restrict__ info, char *recs,
                                                                 PbGran *rowbuf, int *status,
                                                                 int2 *__restrict__ moves,
                                                                 unsigned long long *dbg) {
    // DBG: stamped diagnostic build, wave 0 accumulates 100 MHz ticks per segment into dbg[g][0..7]
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = 0;
#define STAMP(i)                                                              \
    if (DBG && (threadIdx.x >> 6) == 0) {                                     \
        const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime();      \
        seg[i] += tn_ - tlast;                                                \
        tlast = tn_;                                                          \
    }
    constexpr int NTY = NT / 16;
    constexpr int RB = NTY * RT;
    __shared__ double s_cv[NTY];
    __shared__ int s_ci[NTY];
    __shared__ __attribute__((aligned(16))) T s_ub[8];  // winner's block values
    __shared__ int s_win[4];
    __shared__ int s_hist[PB_COLS], s_topid[PB_COLS], s_postop[PB_COLS];
    __shared__ int s_order[RB];
    __shared__ T s_L[RB][8];          // block multipliers per row (0 where the row was already used)
    __shared__ T s_R[8][PB_COLS];     // the block's pivot rows, full width

    // an earlier panel of this factorisation already failed (exchange time-out): do not spin again
    if (info && *info < 0) return;
    const int G = gridDim.x, g = blockIdx.x;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int lane = tid & 63, wave = tid >> 6;
    const int base = g * RB;

    __amdgpu_buffer_rsrc_t r_rec = __builtin_amdgcn_make_buffer_rsrc(recs, 0, 2 * G * PB_REC, 0x00020000);
    __amdgpu_buffer_rsrc_t r_row =
        __builtin_amdgcn_make_buffer_rsrc(rowbuf, 0, 2 * 8 * PB_COLS * (int)sizeof(PbGran), 0x00020000);

    T a[RT][8];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int gi = base + NTY * r + ty;
        const T *src = P + (size_t)gi * ldp + 8 * tx;
#pragma unroll
        for (int c = 0; c < 8; ++c) a[r][c] = (gi < m && 8 * tx + c < jb) ? src[c] : T(0);
    }
    for (int t = tid; t < PB_COLS; t += NT) { s_topid[t] = t; s_postop[t] = t; }
    for (int t = tid; t < RB; t += NT) s_order[t] = -1;
    unsigned frozen = 0;   // bit r: row NTY*r+ty used as a pivot (or outside the panel)
    int fk[RT];            // multipliers of the current block valid for k < fk[r]
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (base + NTY * r + ty >= m) frozen |= 1u << r;
    bool failed = false;
    __syncthreads();

    auto replay = [&](int jj) {
        const int c = s_hist[jj] & 0x3fffffff;
        const bool zero_piv = (s_hist[jj] >> 30) & 1;
        const int p = (c < jb) ? s_postop[c] : c;
        const int d = s_topid[jj];
        if (p != jj) {
            s_topid[jj] = c;
            if (p < jb) s_topid[p] = d;
            s_postop[d] = p;
            if (c < jb) s_postop[c] = jj;
        }
        if (g == 0) {
            ipiv[jj] = row0 + p;
            if (zero_piv && info && *info == 0) *info = col0 + jj + 1;
        }
    };

    // ---- one column; JC = j & 7 is a compile-time constant
    auto column = [&](auto JCt, const int j) {
        constexpr int JC = decltype(JCt)::value;
        const int par = j & 1;
        const int jt = j >> 3;
        const unsigned epoch = (unsigned)(j + 1);
        // 1: candidates of column j among this thread's rows (the block is up to date in registers)
        if (tx == jt) {
            double bv = -1.0;
            int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const double av = fabs((double)a[r][JC]);
                const int gi = base + NTY * r + ty;
                const bool better = (((frozen >> r) & 1u) == 0u) & ((av > bv) | ((av == bv) & (gi < bi)));
                bv = better ? av : bv;
                bi = better ? gi : bi;
            }
            s_cv[ty] = bv;
            s_ci[ty] = bi;
        }
        STAMP(0)
        __syncthreads();
        STAMP(1)
        constexpr int NCM = NTY < 64 ? NTY : 64;
        double wv = s_cv[lane & (NCM - 1)];
        int wi = s_ci[lane & (NCM - 1)];
#pragma unroll
        for (int off = NCM / 2; off > 0; off >>= 1) {
            const double ov = __shfl_xor(wv, off, 64);
            const int oi = __shfl_xor(wi, off, 64);
            const bool better = (ov > wv) | ((ov == wv) & (oi < wi));
            wv = better ? ov : wv;
            wi = better ? oi : wi;
        }
        const bool have = wv >= 0.0;
        const int cl = have ? wi - base : 0;
        const int cty = cl % NTY;
        const int cr = __builtin_amdgcn_readfirstlane(cl / NTY);
        // 2: publish the candidate's block values (one thread holds them all)
        if (tx == jt && ty == (have ? cty : 0)) {
            T v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = T(0);
            if (have) {
#pragma unroll
                for (int k = 0; k < RT; ++k)
                    if (cr == k) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = a[k][c];
                    }
            }
            if (G == 1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) s_ub[c] = v[c];
                s_win[0] = 0; s_win[1] = have ? wi : -1; s_win[2] = have ? 1 : 0;
            } else {
                unsigned long long h = epoch;
#pragma unroll
                for (int c = 0; c < 8; ++c) h = pb_mix(h, pb_bits<T>(v[c]));
                const int idx = have ? wi : -1;
                h = pb_mix(h, (unsigned long long)(unsigned)idx);
                const int off = (par * G + g) * PB_REC;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned long long b0 = pb_bits<T>(v[2 * q]), b1 = pb_bits<T>(v[2 * q + 1]);
                    u4 w;
                    w.x = (unsigned)b0; w.y = (unsigned)(b0 >> 32); w.z = (unsigned)b1; w.w = (unsigned)(b1 >> 32);
                    __builtin_amdgcn_raw_buffer_store_b128(w, r_rec, off + 16 * q, 0, 16);
                }
                u4 t;
                t.x = epoch; t.y = (unsigned)idx; t.z = (unsigned)h; t.w = (unsigned)(h >> 32);
                __builtin_amdgcn_raw_buffer_store_b128(t, r_rec, off + 64, 0, 16);
            }
        }
        if (tid == 64 && j > 0) replay(j - 1);
        STAMP(2)
        // 3: wave 0 reads every workgroup's record (80 B each), all reduce to the same winner
        if (wave == 0 && G > 1) {
            double bv = -2.0;
            int bi = 0x7fffffff, bg = 0;
            T bvals[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) bvals[c] = T(0);
            unsigned pend = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lane + 64 * k < G) pend |= 1u << k;
            int spins = 0;
            while (pend && !failed) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((pend >> k) & 1u) {
                        const int off = (par * G + lane + 64 * k) * PB_REC;
                        u4 w[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) w[q] = __builtin_amdgcn_raw_buffer_load_b128(r_rec, off + 16 * q, 0, 16);
                        const u4 t = __builtin_amdgcn_raw_buffer_load_b128(r_rec, off + 64, 0, 16);
                        if (t.x == epoch) {
                            unsigned long long h = epoch;
                            T v[8];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const unsigned long long b0 = ((unsigned long long)w[q].y << 32) | w[q].x;
                                const unsigned long long b1 = ((unsigned long long)w[q].w << 32) | w[q].z;
                                h = pb_mix(pb_mix(h, b0), b1);
                                v[2 * q] = pb_val<T>(b0);
                                v[2 * q + 1] = pb_val<T>(b1);
                            }
                            h = pb_mix(h, (unsigned long long)t.y);
                            if (h == (((unsigned long long)t.w << 32) | t.z)) {
                                const int hi = (int)t.y;
                                const double hv = fabs((double)v[JC]);
                                const bool better = (hi >= 0) & ((hv > bv) | ((hv == bv) & (hi < bi)));
                                bv = better ? hv : bv;
                                bi = better ? hi : bi;
                                bg = better ? lane + 64 * k : bg;
#pragma unroll
                                for (int c = 0; c < 8; ++c) bvals[c] = better ? v[c] : bvals[c];
                                pend &= ~(1u << k);
                            }
                        }
                    }
                if (pend) {
                    if (++spins > PB_SPIN) failed = true;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            STAMP(3)
            // wave arg-max; remember which lane holds the winner's values
            double rv = bv;
            int ri = bi, rl = lane;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(rv, off, 64);
                const int oi = __shfl_xor(ri, off, 64);
                const int ol = __shfl_xor(rl, off, 64);
                const bool better = (ov > rv) | ((ov == rv) & (oi < ri));
                rv = better ? ov : rv;
                ri = better ? oi : ri;
                rl = better ? ol : rl;
            }
            const bool valid = rv >= 0.0;
            if (lane == rl) {
#pragma unroll
                for (int c = 0; c < 8; ++c) s_ub[c] = bvals[c];
                s_win[0] = bg; s_win[1] = valid ? bi : -1; s_win[2] = valid ? 1 : 0;
            }
            if (__any(failed)) {
                failed = true;
                if (lane == 0) atomicExch(status, 1);
            }
        }
        STAMP(4)
        __syncthreads();
        STAMP(5)
        // 4: the pivot is known: multipliers + rank-1 update INSIDE the block, bookkeeping
        const int wrow = s_win[1];
        const bool valid = s_win[2] != 0;
        const T piv = valid ? s_ub[JC] : T(0);
        const bool act = valid & (piv != T(0));
        if (tid == 0) s_hist[j] = valid ? (wrow | ((piv == T(0)) ? (1 << 30) : 0)) : j;
        if (valid && s_win[0] == g) {
            const int wl = wrow - base;
            const int wr = __builtin_amdgcn_readfirstlane(wl / NTY);
            if (ty == (wl % NTY)) {
                frozen |= 1u << wr;
#pragma unroll
                for (int r = 0; r < RT; ++r) fk[r] = (r == wr) ? JC : fk[r];
                if (tx == 0) s_order[wl] = j;
                // the owner publishes the whole pivot row for the block-end update (off the critical path)
                T v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = T(0);
#pragma unroll
                for (int k = 0; k < RT; ++k)
                    if (wr == k) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = a[k][c];
                    }
                if (G == 1) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) s_R[JC][8 * tx + c] = v[c];
                } else {
                    const int off = (((jt & 1) * 8 + JC) * PB_COLS + 8 * tx) * (int)sizeof(PbGran);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const unsigned long long b = pb_bits<T>(v[c]);
                        u4 w;
                        w.x = (unsigned)b; w.y = (unsigned)(b >> 32); w.z = (unsigned)(jt + 1); w.w = 0u;
                        __builtin_amdgcn_raw_buffer_store_b128(w, r_row, off + 16 * c, 0, 16);
                    }
                }
            }
        }
        if (tx == jt && act) {
            const T rinv = fast_recip<T>(piv);
            T u[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) u[c] = s_ub[c];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const bool on = ((frozen >> r) & 1u) == 0u;
                const T l = on ? a[r][JC] * rinv : T(0);
                a[r][JC] = on ? l : a[r][JC];
#pragma unroll
                for (int c = JC + 1; c < 8; ++c) a[r][c] -= l * u[c];
            }
        }
        STAMP(6)
    };

    const int nblk = (jb + 7) / 8;
    if (DBG) tlast = __builtin_amdgcn_s_memrealtime();
    for (int jt = 0; jt < nblk; ++jt) {
        const int j0 = 8 * jt;
#pragma unroll
        for (int r = 0; r < RT; ++r) fk[r] = ((frozen >> r) & 1u) ? 0 : 8;
#define COL(k) if (j0 + k < jb) column(std::integral_constant<int, k>{}, j0 + k);
        COL(0) COL(1) COL(2) COL(3) COL(4) COL(5) COL(6) COL(7)
#undef COL
        const int nk = (jb - j0 < 8) ? jb - j0 : 8;   // pivots of this block
        if (8 * (jt + 1) >= jb) break;                 // nothing to the right of the last block
        // ---------------- block end: bring the columns right of the block up to date
        if (tx == jt) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) s_L[NTY * r + ty][k] = (k < fk[r]) ? a[r][k] : T(0);
        }
        if (G > 1) {
            // the block's pivot rows: 8 x 128 granules, re-read until the tag says "this block"
            for (int e = tid; e < 8 * PB_COLS; e += NT) {
                const int k = e / PB_COLS, col = e % PB_COLS;
                T val = T(0);
                if (k < nk && !failed) {
                    const int off = (((jt & 1) * 8 + k) * PB_COLS + col) * (int)sizeof(PbGran);
                    int spins = 0;
                    for (;;) {
                        const u4 w = __builtin_amdgcn_raw_buffer_load_b128(r_row, off, 0, 16);
                        if (w.z == (unsigned)(jt + 1)) {
                            val = pb_val<T>(((unsigned long long)w.y << 32) | w.x);
                            break;
                        }
                        if (++spins > PB_SPIN) { failed = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                s_R[k][col] = val;
            }
            if (__syncthreads_or(failed ? 1 : 0)) {
                failed = true;
                if (tid == 0) atomicExch(status, 1);
            }
        } else {
            __syncthreads();
            if (nk < 8)
                for (int e = tid; e < (8 - nk) * PB_COLS; e += NT) s_R[nk + e / PB_COLS][e % PB_COLS] = T(0);
            __syncthreads();
        }
        if (tx > jt) {
            T L11[8][8];  // only t < k is used
#pragma unroll
            for (int k = 1; k < 8; ++k)
#pragma unroll
                for (int t = 0; t < k; ++t) L11[k][t] = s_R[k][j0 + t];
            T l[RT][8];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) l[r][k] = s_L[NTY * r + ty][k];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                T U[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    T v = s_R[k][8 * tx + c];
#pragma unroll
                    for (int t = 0; t < k; ++t) v -= L11[k][t] * U[t];
                    U[k] = v;
                }
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[r][c] -= l[r][k] * U[k];
            }
        }
        STAMP(7)
    }
    if (DBG && tid == 0)
        for (int i = 0; i < 8; ++i) dbg[g * 8 + i] = seg[i];
#undef STAMP
    __syncthreads();
    if (tid == 0) replay(jb - 1);
    // a workgroup whose exchange timed out reports it through info (negative = protocol failure):
    // the host entry points turn that into LSX_ERR_INTERNAL instead of returning garbage factors
    if (failed && info && (tid & 63) == 0) atomicMin(info, -0x40000000);
    __syncthreads();
    if (g == 0 && moves) {
        for (int t = tid; t < 2 * PB_COLS; t += NT) {
            int dst = -1, src = -1;
            if (t < jb) {
                dst = t;
                src = s_hist[t] & 0x3fffffff;
            } else if (t >= PB_COLS && t - PB_COLS < jb) {
                const int d = t - PB_COLS;
                bool is_pivot = false;
                for (int q = 0; q < jb; ++q) is_pivot |= ((s_hist[q] & 0x3fffffff) == d);
                if (!is_pivot) { dst = s_postop[d]; src = d; }
            }
            if (dst == src) dst = src = -1;
            moves[t] = make_int2(dst, src);
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int lr = NTY * r + ty;
        const int gi = base + lr;
        if (gi < m) {
            const int ord = s_order[lr];
            const int dest = ord >= 0 ? ord : (gi < jb ? s_postop[gi] : gi);
            T *dst = P + (size_t)dest * ldp + 8 * tx;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (8 * tx + c < jb) dst[c] = a[r][c];
        }
    }
}

template <typename T, int RT, int NT>
static int panel_blk_launch(lsx_handle_t h, int G, int m, int jb, T *P, int ldp, int row0, int col0,
                            int32_t *d_ipiv, int *d_info) {
    // exchange area in scratch: status | records[2][G] | pivot-row granules[2][8][128]
    const size_t rec_bytes = (size_t)2 * G * PB_REC;
    const size_t row_bytes = (size_t)2 * 8 * PB_COLS * sizeof(PbGran);
    const size_t need = 256 + rec_bytes + row_bytes;
    if (need > h->scratch_bytes) {
        set_error("panel_blk: scratch too small (%zu > %zu)", need, h->scratch_bytes);
        return LSX_ERR_INTERNAL;
    }
    int *status = (int *)h->scratch;
    char *recs = (char *)h->scratch + 256;
    PbGran *rowbuf = (PbGran *)((char *)h->scratch + 256 + rec_bytes);
    LSX_HIP(hipMemsetAsync(h->scratch, 0, need, h->stream));  // epoch / tag 0 never matches
    if (h->panel_debug) {
        const size_t dbg_off = (need + 255) & ~(size_t)255;
        if (dbg_off + (size_t)G * 64 > h->scratch_bytes) { set_error("panel_blk: no room for stamps"); return LSX_ERR_INTERNAL; }
        hipLaunchKernelGGL((panel_blk_kernel<T, RT, NT, true>), dim3(G), dim3(NT), 0, h->stream, m, jb, P, ldp, row0,
                           col0, d_ipiv, d_info, recs, rowbuf, status, (int2 *)h->moves,
                           (unsigned long long *)((char *)h->scratch + dbg_off));
    } else {
        hipLaunchKernelGGL((panel_blk_kernel<T, RT, NT, false>), dim3(G), dim3(NT), 0, h->stream, m, jb, P, ldp, row0,
                           col0, d_ipiv, d_info, recs, rowbuf, status, (int2 *)h->moves, (unsigned long long *)nullptr);
    }
    LSX_HIP(hipGetLastError());
    h->moves_valid = true;
    return LSX_OK;
}

// Returns 1 when the shape is outside what this kernel supports (caller falls back).
template <typename T>
int panel_blocked(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv,
                  int *d_info) {
    if (jb > PB_COLS) return 1;
    // 512 threads x 4 rows = 128-row slices; 8 rows per thread once that needs more than 256 slices
    if ((m + 127) / 128 <= 256 && (m + 127) / 128 <= h->num_cu)
        return panel_blk_launch<T, 4, 512>(h, (m + 127) / 128, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    return 1;
}

template int panel_blocked<double>(lsx_handle_t, int, int, double *, int, int, int, int32_t *, int *);
template int panel_blocked<float>(lsx_handle_t, int, int, float *, int, int, int, int32_t *, int *);

}  // namespace lsx
