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
// Per column j the only cross-CU traffic is one exchange, and the bulk of the
// arithmetic is scheduled UNDER its latency:
//   1. apply the previous column's update to column j only; threads owning
//      column j put it in LDS and reduce to the workgroup's candidate (largest
//      |a| among rows not yet used as pivots).
//   2. the wave holding the candidate row brings that one row up to date and
//      publishes it as 128 self-validating 16-byte granules {value, epoch} plus a
//      16-byte header {|a|, row, epoch, checksum} -- write-through (sc1) stores,
//      NO drain and no flag: every granule carries its own tag (guide G16 R2).
//   3. all threads now apply the deferred rank-1 update of column j-1 to the rest
//      of their tile (linalg.py:587-596) while the exchange is in flight; one idle
//      lane replays the previous interchange on the position maps.
//   4. wave 0 polls the G headers (one per lane, relaxed sc1 loads), every
//      workgroup reduces to the same winner (largest |a|, lowest row on ties),
//      then reads the winner's granules until all 128 tags match, and puts the
//      row in LDS.
//   5. every thread derives its multipliers l = a[:,j]/pivot (kept in place:
//      unit-lower L) and keeps (l, u) in registers as the next deferred update.
// Rows never move during the loop ("implicit pivoting": used rows are frozen).
// The LAPACK interchange list ipiv and each row's final position are obtained by
// replaying the swaps on two jb-entry maps; at the end every row is written
// straight to its final position.
//
// Every spin is bounded; a timeout sets *status and lets the grid drain.
#include <type_traits>

#include "common.h"
#include "panel_xchg.h"

namespace lsx {

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

// DBG = true builds the stamped diagnostic variant: wave 0 accumulates, per segment of
// the column loop, 100 MHz wall-clock ticks into dbg[g][0..7] (never used for results).
//
// Code-generation notes (gfx950, ROCm 7.2): the column loop is unrolled by 8 through a
// templated step so that every register-tile index is a compile-time constant (a
// `switch` on the column made hipcc shuffle the tile through AGPRs: 1.3k v_accvgpr and
// 370 branches, 5 us per column); conditionals on per-row state are selects, not
// branches; the tile is RT x 8 with RT = 4 so the whole state stays in arch VGPRs.
template <typename T, int RT, int NT, bool DBG>
__global__ __launch_bounds__(NT, NT / 256) void panel_coop_kernel(int m, int jb, T *__restrict__ P, int ldp,
                                                            int row0, int col0,
                                                            int32_t *__restrict__ ipiv,
                                                            int *__restrict__ info, XHdr *hdr,
                                                            XGran *xrow, int *status,
                                                            unsigned long long *dbg,
                                                            int2 *__restrict__ moves) {
    constexpr int NTY = NT / 16;   // thread rows
    constexpr int RB = NTY * RT;   // panel rows per workgroup
    __shared__ T s_col[2][RB];
    __shared__ double s_cv[NTY];
    __shared__ int s_ci[NTY];
    __shared__ __attribute__((aligned(16))) T s_u[PC_COLS];
    __shared__ int s_win[4];      // [0]=winner workgroup, [1]=winner row (panel-local), [2]=valid
    __shared__ int s_hist[PC_COLS];           // winner row of every column (for the swap replay)
    __shared__ int s_topid[PC_COLS], s_postop[PC_COLS];
    __shared__ int s_order[RB];

    // latency-critical and tiny: when the look-ahead driver runs this kernel beside the trailing
    // update, its waves must win issue arbitration against the co-resident MFMA waves
    __builtin_amdgcn_s_setprio(3);
    // an earlier panel of this factorisation already failed (exchange time-out): do not spin again
    if (info && *info < 0) return;
    const int G = gridDim.x, g = blockIdx.x;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int lane = tid & 63, wave = tid >> 6;
    const int base = g * RB;

    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = 0;
#define STAMP(i)                                                              \
    if (DBG && wave == 0) {                                                   \
        const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime();      \
        seg[i] += tn_ - tlast;                                                \
        tlast = tn_;                                                          \
    }

    // buffer descriptors for the exchange area (sc1 traffic only)
    __amdgpu_buffer_rsrc_t r_hdr =
        __builtin_amdgcn_make_buffer_rsrc(hdr, 0, 2 * G * HDR_STRIDE, 0x00020000);
    __amdgpu_buffer_rsrc_t r_row =
        __builtin_amdgcn_make_buffer_rsrc(xrow, 0, 2 * G * PC_COLS * (int)sizeof(XGran), 0x00020000);

    // ---- load the slice (rows >= m and columns >= jb read as zero)
    T a[RT][8];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int gi = base + NTY * r + ty;
        const T *src = P + (size_t)gi * ldp + 8 * tx;
#pragma unroll
        for (int c = 0; c < 8; ++c) a[r][c] = (gi < m && 8 * tx + c < jb) ? src[c] : T(0);
    }
    for (int t = tid; t < PC_COLS; t += NT) { s_topid[t] = t; s_postop[t] = t; }
    for (int t = tid; t < RB; t += NT) s_order[t] = -1;
    unsigned frozen = 0;  // bit r: local row 16*r+ty already used as a pivot (or outside the panel)
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (base + NTY * r + ty >= m) frozen |= 1u << r;
    bool failed = false;
    // deferred rank-1 update of the previous column: a[r][c] -= lp[r] * up[c]
    T lp[RT], up[8];
#pragma unroll
    for (int r = 0; r < RT; ++r) lp[r] = T(0);
#pragma unroll
    for (int c = 0; c < 8; ++c) up[c] = T(0);
    __syncthreads();

    // replay of interchange jj on the position maps (one lane; LAPACK order bookkeeping)
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

    // one column; JC = j & 7 is a compile-time constant
    auto column = [&](auto JCt, const int j) {
        constexpr int JC = decltype(JCt)::value;
        const int par = j & 1;
        const int jt = j >> 3;
        // ---------------- 1: column j up to date, to LDS, per-thread-row candidates
        if (tx == jt) {
            double bv = -1.0;
            int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                a[r][JC] -= lp[r] * up[JC];
                const T v = a[r][JC];
                s_col[par][NTY * r + ty] = v;
                const double av = fabs((double)v);
                const int gi = base + NTY * r + ty;
                const bool better = (((frozen >> r) & 1u) == 0u) & ((av > bv) | ((av == bv) & (gi < bi)));
                bv = better ? av : bv;
                bi = better ? gi : bi;
            }
            up[JC] = T(0);
            s_cv[ty] = bv;
            s_ci[ty] = bi;
        }
        STAMP(0)
        __syncthreads();
        STAMP(1)
        // every wave reduces the 16 thread-row candidates to the workgroup's candidate
        constexpr int NCM = NTY < 64 ? NTY : 64;
        double wv = s_cv[lane & (NCM - 1)];
        int wi = s_ci[lane & (NCM - 1)];
        row16_argmax(wv, wi);
        if (NCM > 16) {  // thread rows beyond 16: combine the 16-lane rows through scalar registers
            double rv[NCM / 16];
            int ri[NCM / 16];
#pragma unroll
            for (int rr = 0; rr < NCM / 16; ++rr) {
                rv[rr] = readlane_d(wv, 16 * rr);
                ri[rr] = __builtin_amdgcn_readlane(wi, 16 * rr);
            }
            wv = rv[0];
            wi = ri[0];
#pragma unroll
            for (int rr = 1; rr < NCM / 16; ++rr) {
                const bool better = (rv[rr] > wv) | ((rv[rr] == wv) & (ri[rr] < wi));
                wv = better ? rv[rr] : wv;
                wi = better ? ri[rr] : wi;
            }
        }
        const bool have = wv >= 0.0;
        const int cl = have ? wi - base : 0;          // slice-local row of the candidate
        const int cty = cl % NTY;
        const int cr = __builtin_amdgcn_readfirstlane(cl / NTY);
        // ---------------- 2: publish (the wave that holds the candidate row; wave 0 if none)
        const int pub_wave = __builtin_amdgcn_readfirstlane(have ? (cty >> 2) : 0);
        if (wave == pub_wave) {
            if (have && ty == cty) {
                T rowv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) rowv[c] = T(0);
#pragma unroll
                for (int k = 0; k < RT; ++k)
                    if (cr == k) {  // scalar condition: one of RT short blocks runs
#pragma unroll
                        for (int c = 0; c < 8; ++c) rowv[c] = a[k][c] - lp[k] * up[c];
                    }
                if (G == 1) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) s_u[8 * tx + c] = rowv[c];
                } else {
                    const int off = ((par * G + g) * PC_COLS + 8 * tx) * (int)sizeof(XGran);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        unsigned long long bits;
                        if (sizeof(T) == 8) bits = (unsigned long long)__double_as_longlong((double)rowv[c]);
                        else bits = (unsigned long long)__float_as_uint((float)rowv[c]);
                        u4 v;
                        v.x = (unsigned)bits; v.y = (unsigned)(bits >> 32); v.z = (unsigned)(j + 1); v.w = 0u;
                        __builtin_amdgcn_raw_buffer_store_b128(v, r_row, off + 16 * c, 0, 16);
                    }
                }
            }
            if (G > 1 && lane == (have ? ((cty & 3) * 16) : 0)) {
                const double hv = have ? wv : -1.0;
                const unsigned long long vb = (unsigned long long)__double_as_longlong(hv);
                u4 h;
                h.x = (unsigned)vb; h.y = (unsigned)(vb >> 32);
                h.z = (unsigned)(have ? wi : -1);
                h.w = ((unsigned)(j + 1) << 16) | fold16(hv);
                __builtin_amdgcn_raw_buffer_store_b128(h, r_hdr, (par * G + g) * HDR_STRIDE, 0, 16);
            }
        }
        if (G == 1 && tid == 0) { s_win[0] = 0; s_win[1] = have ? wi : -1; s_win[2] = have ? 1 : 0; }
        STAMP(2)
        // ---------------- 3: deferred bulk update of column j-1, under the exchange latency
        if (tx >= jt) {  // up[] is zero for columns <= j-1
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) a[r][c] -= lp[r] * up[c];
        }
        if (tid == 64 && j > 0) replay(j - 1);
        STAMP(3)
        // ---------------- 4: wave 0 gathers all candidates, picks the winner, fetches its row
        if (wave == 0 && G > 1) {
            double bv = -2.0;
            int bi = 0x7fffffff, bg = 0;
            // lane q watches headers q, q+64, q+128, q+192: all of them in flight per poll round
            unsigned pend = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lane + 64 * k < G) pend |= 1u << k;
            int spins = 0;
            while (pend && !failed) {
                u4 h[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((pend >> k) & 1u)
                        h[k] = __builtin_amdgcn_raw_buffer_load_b128(r_hdr, (par * G + lane + 64 * k) * HDR_STRIDE, 0, 16);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((pend >> k) & 1u) {
                        const double hv = __longlong_as_double((long long)(((unsigned long long)h[k].y << 32) | h[k].x));
                        if ((h[k].w >> 16) == (unsigned)(j + 1) && (h[k].w & 0xffffu) == fold16(hv)) {
                            const int hi = (int)h[k].z;
                            const bool better = (hi >= 0) & ((hv > bv) | ((hv == bv) & (hi < bi)));
                            bv = better ? hv : bv;
                            bi = better ? hi : bi;
                            bg = better ? lane + 64 * k : bg;
                            pend &= ~(1u << k);
                        }
                    }
                if (pend) {
                    if (++spins > SPIN_LIMIT) failed = true;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            // 64-lane arg-max carrying the workgroup id: 16-lane rows by DPP, the 4 rows by readlane
            {
                // pack (row, workgroup) so one integer travels with the value; ties -> lowest row
                long long key = ((long long)bi << 20) | (long long)bg;  // bg < 2^20
                int klo = (int)(key & 0xffffffffll), khi = (int)(key >> 32);
#define LSX_DPP_STEP3(CTRL)                                                                   \
                {                                                                             \
                    const double ov_ = dpp_d<CTRL>(bv);                                       \
                    const int olo_ = dpp_i<CTRL>(klo), ohi_ = dpp_i<CTRL>(khi);               \
                    const long long ok_ = ((long long)ohi_ << 32) | (unsigned)olo_;            \
                    const long long mk_ = ((long long)khi << 32) | (unsigned)klo;              \
                    const bool b_ = (ov_ > bv) | ((ov_ == bv) & (ok_ < mk_));                  \
                    bv = b_ ? ov_ : bv;                                                       \
                    klo = b_ ? olo_ : klo;                                                    \
                    khi = b_ ? ohi_ : khi;                                                    \
                }
                LSX_DPP_STEP3(0xB1) LSX_DPP_STEP3(0x4E) LSX_DPP_STEP3(0x141) LSX_DPP_STEP3(0x140)
#undef LSX_DPP_STEP3
                double rv = readlane_d(bv, 0);
                long long rk = ((long long)__builtin_amdgcn_readlane(khi, 0) << 32) | (unsigned)__builtin_amdgcn_readlane(klo, 0);
#pragma unroll
                for (int rr = 1; rr < 4; ++rr) {
                    const double ov = readlane_d(bv, 16 * rr);
                    const long long ok = ((long long)__builtin_amdgcn_readlane(khi, 16 * rr) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane(klo, 16 * rr);
                    const bool b = (ov > rv) | ((ov == rv) & (ok < rk));
                    rv = b ? ov : rv;
                    rk = b ? ok : rk;
                }
                bv = rv;
                bi = (int)(rk >> 20);
                bg = (int)(rk & 0xfffff);
            }
            const bool valid = bv >= 0.0;
            STAMP(4)
            if (valid && !failed) {
                // winner's row: 128 granules, two per lane, re-read until both tags match
                const int roff = (par * G + bg) * PC_COLS * (int)sizeof(XGran);
                int spins = 0;
                for (;;) {
                    const u4 v0 = __builtin_amdgcn_raw_buffer_load_b128(r_row, roff + 16 * lane, 0, 16);
                    const u4 v1 = __builtin_amdgcn_raw_buffer_load_b128(r_row, roff + 16 * (lane + 64), 0, 16);
                    if (v0.z == (unsigned)(j + 1) && v1.z == (unsigned)(j + 1)) {
                        if (sizeof(T) == 8) {
                            s_u[lane] = (T)__longlong_as_double((long long)(((unsigned long long)v0.y << 32) | v0.x));
                            s_u[lane + 64] = (T)__longlong_as_double((long long)(((unsigned long long)v1.y << 32) | v1.x));
                        } else {
                            s_u[lane] = (T)__uint_as_float(v0.x);
                            s_u[lane + 64] = (T)__uint_as_float(v1.x);
                        }
                        break;
                    }
                    if (++spins > SPIN_LIMIT) { failed = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (lane == 0) { s_win[0] = bg; s_win[1] = valid ? bi : -1; s_win[2] = valid ? 1 : 0; }
            if (__any(failed)) {  // sticky and wave-uniform: later columns skip the polling
                failed = true;
                if (lane == 0) atomicExch(status, 1);
            }
        }
        STAMP(5)
        __syncthreads();
        STAMP(6)
        // ---------------- 5: multipliers of column j; (l, u) become the next deferred update
        const int wrow = s_win[1];
        const bool valid = s_win[2] != 0;
        const T piv = valid ? s_u[j] : T(0);
        if (tid == 0) s_hist[j] = valid ? (wrow | ((piv == T(0)) ? (1 << 30) : 0)) : j;
        if (valid && s_win[0] == g) {
            const int wl = wrow - base;
            if (ty == (wl % NTY)) {
                frozen |= 1u << (wl / NTY);
                if (tx == 0) s_order[wl] = j;
            }
        }
        const bool act = valid & (piv != T(0));
        const T rinv = act ? fast_recip<T>(piv) : T(0);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const T v = s_col[par][NTY * r + ty] * rinv;
            lp[r] = ((frozen >> r) & 1u) ? T(0) : v;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const T v = s_u[8 * tx + c];
            up[c] = (act & (8 * tx + c > j)) ? v : T(0);
        }
        if (tx == jt) {
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r][JC] = (act & (((frozen >> r) & 1u) == 0u)) ? lp[r] : a[r][JC];
        }
        STAMP(7)
    };

    if (DBG) tlast = __builtin_amdgcn_s_memrealtime();
    for (int j0 = 0; j0 < jb; j0 += 8) {
#define COL(k) if (j0 + k < jb) column(std::integral_constant<int, k>{}, j0 + k);
        COL(0) COL(1) COL(2) COL(3) COL(4) COL(5) COL(6) COL(7)
#undef COL
    }
    if (DBG && tid == 0)
        for (int i = 0; i < 8; ++i) dbg[g * 8 + i] = seg[i];
#undef STAMP
    // the last column's deferred update touches only columns > jb-1: nothing left inside the panel
    __syncthreads();
    if (tid == 0) replay(jb - 1);
    // a workgroup whose exchange timed out reports it through info (negative = protocol failure):
    // the host entry points turn that into LSX_ERR_INTERNAL instead of returning garbage factors
    if (failed && info && (tid & 63) == 0) atomicMin(info, -0x40000000);
    __syncthreads();
    // ---- the same permutation as a gather list for the columns outside the panel:
    // final[row0 + dst] = old[row0 + src]; slot j: pivot j, slot PC_COLS + d: displaced top row d
    if (g == 0 && moves) {
        for (int t = tid; t < 2 * PC_COLS; t += NT) {
            int dst = -1, src = -1;
            if (t < jb) {
                dst = t;
                src = s_hist[t] & 0x3fffffff;
            } else if (t >= PC_COLS && t - PC_COLS < jb) {
                const int d = t - PC_COLS;
                bool is_pivot = false;
                for (int q = 0; q < jb; ++q) is_pivot |= ((s_hist[q] & 0x3fffffff) == d);
                if (!is_pivot) { dst = s_postop[d]; src = d; }
            }
            if (dst == src) dst = src = -1;
            moves[t] = make_int2(dst, src);
        }
    }
    // ---- every row straight to its final (LAPACK-order) position
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
static int panel_coop_launch(lsx_handle_t h, int G, int m, int jb, T *P, int ldp, int row0, int col0,
                             int32_t *d_ipiv, int *d_info) {
    // exchange area in scratch: status | headers[2][G] (HDR_STRIDE apart) | granule rows[2][G][128]
    const size_t hdr_bytes = (size_t)2 * G * HDR_STRIDE;
    const size_t need = 256 + hdr_bytes + (size_t)2 * G * PC_COLS * sizeof(XGran);
    const size_t dbg_off = (need + 255) & ~(size_t)255;
    if (dbg_off + (h->panel_debug ? (size_t)G * 64 : 0) > h->scratch_bytes) {
        set_error("panel_coop: scratch too small (%zu > %zu)", dbg_off, h->scratch_bytes);
        return LSX_ERR_INTERNAL;
    }
    int *status = (int *)h->scratch;
    XHdr *hdr = (XHdr *)((char *)h->scratch + 256);
    XGran *xrow = (XGran *)((char *)h->scratch + 256 + hdr_bytes);
    // status word, headers AND granules are zeroed before EVERY launch: epoch 0 never matches
    LSX_HIP(hipMemsetAsync(h->scratch, 0, need, h->stream));
    if (h->panel_debug) {
        unsigned long long *dbg = (unsigned long long *)((char *)h->scratch + dbg_off);
        hipLaunchKernelGGL((panel_coop_kernel<T, RT, NT, true>), dim3(G), dim3(NT), 0, h->stream, m, jb, P, ldp,
                           row0, col0, d_ipiv, d_info, hdr, xrow, status, dbg, (int2 *)h->moves);
    } else {
        hipLaunchKernelGGL((panel_coop_kernel<T, RT, NT, false>), dim3(G), dim3(NT), 0, h->stream, m, jb, P, ldp,
                           row0, col0, d_ipiv, d_info, hdr, xrow, status, (unsigned long long *)nullptr,
                           (int2 *)h->moves);
    }
    LSX_HIP(hipGetLastError());
    h->moves_valid = true;
    return LSX_OK;
}

// Returns 1 when the shape is outside what the cooperative kernel supports (caller falls back).
template <typename T>
int panel_cooperative(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0,
                      int32_t *d_ipiv, int *d_info) {
    if (jb > PC_COLS) return 1;
    // shape of a workgroup: NT threads (16 thread-columns x NT/16 thread-rows), RT rows per thread.
    // More threads with small tiles keep each wave's instruction stream short while halving the
    // number of workgroups that take part in every exchange; all of them must be resident at once.
    int nt = h->panel_nt, rt = h->panel_rt;
    auto rows = [](int nt_, int rt_) { return nt_ / 16 * rt_; };
    if (h->panel_nt == 0) {  // auto: tall panels poll faster with half as many workgroups
        nt = m >= 6144 ? 512 : 256;
        rt = 4;
    }
    if ((m + rows(nt, rt) - 1) / rows(nt, rt) > h->num_cu) { nt = 512; rt = 8; }  // 256-row slices
    const int G = (m + rows(nt, rt) - 1) / rows(nt, rt);
    if (G > h->num_cu) return 1;
#define LSX_PC(RT_, NT_) \
    if (rt == RT_ && nt == NT_) return panel_coop_launch<T, RT_, NT_>(h, G, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    LSX_PC(4, 256) LSX_PC(8, 256) LSX_PC(2, 512) LSX_PC(4, 512) LSX_PC(8, 512) LSX_PC(2, 1024)
#undef LSX_PC
    set_error("panel_coop: unsupported workgroup shape nt=%d rt=%d", nt, rt);
    return LSX_ERR_ARG;
}

template int panel_cooperative<double>(lsx_handle_t, int, int, double *, int, int, int, int32_t *, int *);
template int panel_cooperative<float>(lsx_handle_t, int, int, float *, int, int, int, int32_t *, int *);

}  // namespace lsx
