// Panel factorisation, column-distributed inside one XCD ("panel = 4", option panel_col = 1, panels of up to 4096 rows).
//
// Reference loops covered: pivot search, row swap, scaling and elimination below the pivot for jb consecutive pivots
// (linalg_solver/linalg.py:548-596).  Same arithmetic per element as every other panel mode (multipliers from the same
// fast_recip, the same fused multiply-adds in the same order of columns), so the factors and the pivot sequence are
// bit-identical to theirs.
//
// Why another cut.  kernels_panel_x.hip distributes the panel's ROWS over the workgroups, so every column costs one
// all-gather of candidates across 32 CUs: round 3 measured 0.63 us from a workgroup's announcement to the moment it holds
// everybody's record, beside 0.6 us of one wave's dependent arithmetic -- 1.2-1.6 us per column whatever was done to
// the protocol.  Here the COLUMNS are distributed: workgroup g (256 threads, one wave per SIMD of one CU) holds
// columns 4 g .. 4 g + 3 of ALL rows in registers (4096 x 4 fp64 = 128 registers per lane).  The pivot search of a column is
// then a reduction inside ONE workgroup (per thread, DPP across the wave, four records in LDS, one barrier), and a workgroup
// runs its four columns back to back.  What crosses CUs is the multiplier vector of a finished column -- m values, written
// once into a buffer in the XCD's L2 and read by the workgroups to the right, which apply the column to their own four
// (left-looking in the panel): no agreement round, and the wait for it is off every chain but the hand-over from one
// workgroup to the next (once per four columns).
//
//   owner of column j (workgroup j / 4):  candidates -> arg-max (LDS, 1 barrier; a wave's record carries its candidate
//     row's entries in the owner's columns) -> flag[j] = {j + 1 | act << 16, pivot row} goes out at once -> reciprocal,
//     multipliers l = a[:, j] / pivot -> update of its columns right of j -> l to Lbuf[j][:] (coalesced) -> next column.
//   everybody to the right:  a wave polls flag[j .. j + 31] in one load; the thread that holds the pivot row puts its four
//     entries into LDS (1 barrier); l from Lbuf (L2), a[:, c] -= l * u[c] for its columns.  Lbuf is handed over filled
//     with all-ones words (a NaN nothing computes) and a value counts as soon as it is not that: nobody waits for the
//     acknowledgement of a store.  The multipliers of column j + 1 and the poll for it travel while column j is applied.
//   Four waves (one per SIMD), up to 16 rows per lane.  The per-column steps that do not shrink with the row count (two arg-max
//     reductions, record exchange, reciprocal, bookkeeping) are a dependent chain of ~2000 cycles whatever the number of
//     waves (measured with 16 waves x 8 rows and with 4 x 16: 1.1 and 0.95 us per own column at 256 rows); four waves keep
//     the barrier and the second reduction small.
//   A row that becomes a pivot is final: its thread writes it to its LAPACK position (row j of the panel) there and then and
//     zeroes its registers, so that no later step needs a "this row is done" mask -- a zero never wins the search (a column
//     whose largest entry is zero takes a slow path with the mask), its multiplier is 0 x 1/pivot = 0 and the update leaves
//     it alone.
// MEASURED SLOWER than kernels_panel_x.hip at every height (1.28 us per column at 256 rows, 3.1 at 4096, against 1.27 and
// 1.47; DESIGN 5 has the stamps): the second arg-max and the barrier cost what the all-gather cost, and the row work of a column
// (m multiplications, m stores through one CU's 64 bytes a cycle) is done by one CU instead of 32.  Off by default (option
// panel_col); kept as an independent second implementation that must reproduce the other's bits (tests).  Above 4096 rows
// the row-distributed kernel takes the panel.
// Placement as in kernels_panel_x.hip: 8 G workgroups are launched, those with blockIdx % 8 == 0 take part, the XCC ids are
// compared in a handshake and a panel whose participants do not share one runs with write-through stores.
// Bookkeeping (LAPACK-order positions, ipiv, info, the gather list for the columns outside the panel) is the one of
// kernels_panel_x.hip; every workgroup holds all rows, so every workgroup replays the whole sequence.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "panel_xchg.h"

namespace lsx {

constexpr int PCW = 4;                 // panel columns per workgroup
constexpr int PC_NT = 256;             // threads per workgroup: row t, t + 256, ...
constexpr int PC_LOGNT = 8;
constexpr int PC_MAXROWS = 16 * PC_NT; // 16 rows per lane
constexpr int PC_NONE = 0x7fffffff;
constexpr size_t PC_FLAG_BYTES = (size_t)PC_COLS * 8;   // flags[column]

// wave-wide arg-max (largest key, lowest idx on ties); idx == PC_NONE: no candidate (key must be 0 then).
__device__ __forceinline__ int pc_argmax64(unsigned khi, unsigned klo, int idx) {
    const unsigned mhi = rows_max_u32<4>(row16_max_u32(khi), 0);
    const bool top = (khi == mhi) & (idx != PC_NONE);
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(top);
    if (mask == 0ull) return PC_NONE;
    if ((mask & (mask - 1ull)) == 0ull)
        return __builtin_amdgcn_readlane(idx, __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask)));
    const unsigned mlo = rows_max_u32<4>(row16_max_u32(top ? klo : 0u), 0);
    return rows_min_i32<4>(row16_min_i32((top & (klo == mlo)) ? idx : PC_NONE), 0);
}

// the same over lanes 0..15 only (the per-wave records); the other lanes must hold idx == PC_NONE
__device__ __forceinline__ int pc_argmax16(unsigned khi, unsigned klo, int idx) {
    const unsigned mhi = (unsigned)__builtin_amdgcn_readlane((int)row16_max_u32(khi), 0);
    const bool top = (khi == mhi) & (idx != PC_NONE);
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(top);
    if (mask == 0ull) return PC_NONE;
    if ((mask & (mask - 1ull)) == 0ull)
        return __builtin_amdgcn_readlane(idx, __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask)));
    const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)row16_max_u32(top ? klo : 0u), 0);
    return __builtin_amdgcn_readlane(row16_min_i32((top & (klo == mlo)) ? idx : PC_NONE), 0);
}

__device__ __forceinline__ double pc_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float pc_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// exchange area: status, XCC handshake (zeroed) || flags[PC_COLS] (8 bytes each: {(j + 1) | act << 16, pivot row or -1}) |
// Lbuf[jb][mpad] (both handed over filled with 0xff bytes)
template <typename T, int RTC, bool XCD>
__device__ __forceinline__ void panel_c_body(const int G, const int g, const int m, const int jb, T *__restrict__ P, const int ldp,
                                             const int row0, const int col0, int32_t *__restrict__ ipiv, int *__restrict__ info,
                                             uint2 *flags, T *Lbuf, const int mpad, int *status, int2 *__restrict__ moves,
                                             const int spin_limit, unsigned long long *dbg) {
    constexpr int NT = PC_NT, NW = NT / 64;
    // per-wave candidate records of the owner: {|a| lo, |a| hi, row, -, the row's PCW entries}; two sets by column parity:
    // a fast wave writes the next column's record while a slow one still reads
    __shared__ __attribute__((aligned(16))) unsigned s_rec[2][NW][16];
    __shared__ int s_rec2[NW];                                   // slow path (largest entry zero): lowest live row per wave
    __shared__ __attribute__((aligned(16))) T s_u[2][PCW];       // the pivot row's entries in this workgroup's columns
    __shared__ int s_hist[PC_COLS], s_topid[PC_COLS], s_postop[PC_COLS];

    if (info && *info < 0) return;   // an earlier panel of this factorisation already failed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // development stamps (LSX_PC_DBG=1, tools/panelc_check.py): 100 MHz clock at the stages of workgroup g
    auto stamp = [&](const int k) __attribute__((always_inline)) {
        if (dbg && tid == 0) dbg[g * 16 + k] = wall_clock64();
    };
    int n_waits = 0;
    stamp(0);
    const int C0 = PCW * g;                            // first panel column of this workgroup
    const int NC = (jb - C0 < PCW) ? jb - C0 : PCW;    // >= 1 for every participant
    const int nrt = mpad >> PC_LOGNT;                  // register rows that exist in the multiplier buffer (<= RTC)

    // ---- the slice: rows tid + 256 i, columns C0 .. C0 + 3 (zero outside the panel)
    T a[RTC][PCW];
    typedef T v2t __attribute__((ext_vector_type(2)));
    const bool wide = (NC == PCW) && ((size_t)(P + C0) % 16 == 0) && (ldp % 2 == 0) && sizeof(T) == 8;
#pragma unroll
    for (int i = 0; i < RTC; ++i) {
        const int gi = tid + NT * i;
        const T *src = P + (size_t)(gi < m ? gi : 0) * ldp + C0;
        if (wide) {
            const v2t v0 = *(const v2t *)(src), v1 = *(const v2t *)(src + 2);
            a[i][0] = gi < m ? v0[0] : T(0); a[i][1] = gi < m ? v0[1] : T(0);
            a[i][2] = gi < m ? v1[0] : T(0); a[i][3] = gi < m ? v1[1] : T(0);
        } else {
#pragma unroll
            for (int c = 0; c < PCW; ++c) a[i][c] = (gi < m && c < NC) ? src[c] : T(0);
        }
    }
    for (int t = tid; t < PC_COLS; t += NT) { s_topid[t] = t; s_postop[t] = t; }
    unsigned frozen = 0;   // bit i: row tid + 256 i is out of the game (written to its final place and zeroed, or outside the panel)
#pragma unroll
    for (int i = 0; i < RTC; ++i)
        if (tid + NT * i >= m) frozen |= 1u << i;
    bool failed = false;
    __syncthreads();
    stamp(1);

    // flags[column], 8 bytes: {(column + 1) | act << 16, pivot row or -1}, published the moment the owner knows its pivot.
    const __amdgpu_buffer_rsrc_t r_flag = __builtin_amdgcn_make_buffer_rsrc(flags, 0, PC_COLS * 8, 0x00020000);
    // the multiplier buffer: loads always bypass L1 (the same addresses held another panel's multipliers three panels ago),
    // stores stay in this XCD's L2 (XCD) or go through to memory (participants on several XCDs)
    const __amdgpu_buffer_rsrc_t r_l = __builtin_amdgcn_make_buffer_rsrc(Lbuf, 0, PC_COLS * mpad * (int)sizeof(T), 0x00020000);
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    auto ld_l = [&](const int j, const int row) __attribute__((always_inline)) -> T {
        const int off = (j * mpad + row) * (int)sizeof(T);
        if (sizeof(T) == 8) {
            const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r_l, off, 0, 16);
            return (T)__longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
        }
        return (T)__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_l, off, 0, 16));
    };
    auto st_l = [&](const int j, const int row, const T v) __attribute__((always_inline)) {
        const int off = (j * mpad + row) * (int)sizeof(T);
        if (sizeof(T) == 8) {
            const unsigned long long b = (unsigned long long)__double_as_longlong((double)v);
            u2 w; w.x = (unsigned)b; w.y = (unsigned)(b >> 32);
            __builtin_amdgcn_raw_buffer_store_b64(w, r_l, off, 0, XCD ? 0 : 16);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)v), r_l, off, 0, XCD ? 0 : 16);
        }
    };
    auto unwritten = [&](const T v) __attribute__((always_inline)) -> bool {
        if (sizeof(T) == 8) return (unsigned)((unsigned long long)__double_as_longlong((double)v) >> 32) == 0xffffffffu;
        return __float_as_uint((float)v) == 0xffffffffu;
    };
    // replay of interchange jj on the position maps (one thread; LAPACK order bookkeeping), as in kernels_panel_x.hip
    auto replay = [&](int jj) __attribute__((always_inline)) {
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
    int rp = 0;   // the keeper (thread NT - 1): interchanges replayed so far; it catches up where nobody waits for it
    auto note_pivot = [&](const int j, const int wrow, const bool act) __attribute__((always_inline)) {
        if (tid == NT - 1) s_hist[j] = wrow >= 0 ? (wrow | (act ? 0 : (1 << 30))) : j;
    };
    // The pivot row of column j (wave-uniform; -1: none) is final: to LDS for the update (lds, may be null), to row j of
    // the panel, out of the game.
    auto freeze = [&](const int row, const int j, T *lds) __attribute__((always_inline)) {
        if (row < 0) return;
        if (((row >> 6) & (NW - 1)) != wave) return;
        const int ck = row >> PC_LOGNT;
        const bool mine = (row & (NT - 1)) == tid;
        T *dst = P + (size_t)j * ldp + C0;
#pragma unroll
        for (int k = 0; k < RTC; ++k)
            if (ck == k && mine) {
                if (lds) {
#pragma unroll
                    for (int c = 0; c < PCW; ++c) lds[c] = a[k][c];
                }
                if (wide) {
                    v2t v0, v1;
                    v0[0] = a[k][0]; v0[1] = a[k][1]; v1[0] = a[k][2]; v1[1] = a[k][3];
                    *(v2t *)(dst) = v0;
                    *(v2t *)(dst + 2) = v1;
                } else {
#pragma unroll
                    for (int c = 0; c < PCW; ++c)
                        if (c < NC) dst[c] = a[k][c];
                }
#pragma unroll
                for (int c = 0; c < PCW; ++c) a[k][c] = T(0);
                frozen |= 1u << k;
            }
    };
    // Flags of columns j, j + 1, ... (lane k: column j + k, 32 at a time, below lim): issue / evaluate apart, so that a
    // poll can travel under arithmetic.  count() = how many consecutive columns from j are published.
    auto poll_issue = [&](const int j, const int lim) __attribute__((always_inline)) -> u2 {
        const int c = j + lane;
        u2 f; f.x = 0u; f.y = 0u;
        if (lane < 32 && c < lim) f = __builtin_amdgcn_raw_buffer_load_b64(r_flag, c * 8, opaque_zero(), 16);
        return f;
    };
    auto poll_count = [&](const int j, const u2 f) __attribute__((always_inline)) -> int {
        const bool ok = (f.x & 0xffffu) == (unsigned)(j + lane + 1);   // lanes outside the range hold 0: never equal
        const unsigned long long nb = ~__builtin_amdgcn_ballot_w64(ok);
        return nb == 0ull ? 64 : __builtin_ctzll(nb);
    };
    // blocking form with the bounded spin; on a time-out the column counts as "no pivot" and the panel as failed
    auto poll_wait = [&](const int j, const int lim, u2 &f) __attribute__((always_inline)) -> int {
        int spins = failed ? spin_limit : 0;
        for (;;) {
            f = poll_issue(j, lim);
            const int cnt = poll_count(j, f);
            if (cnt > 0) return cnt;
            n_waits += 1;
            if (++spins > spin_limit) {
                if (!failed) {
                    failed = true;
                    if (lane == 0) atomicExch(status, 1);
                }
                f.x = 0u; f.y = 0xffffffffu;
                return 1;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto publish = [&](const int col, const int prow, const int act) __attribute__((always_inline)) {
        if (tid == 0) {
            u2 f; f.x = (unsigned)(col + 1) | ((unsigned)act << 16); f.y = (unsigned)prow;
            __builtin_amdgcn_raw_buffer_store_b64(f, r_flag, col * 8, 0, XCD ? 0 : 16);
        }
    };

    // ---------------- columns left of mine: apply them as they are published
    auto load_l = [&](const int j, T (&l)[RTC]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RTC; ++i) l[i] = (i < nrt) ? ld_l(j, tid + NT * i) : T(0);   // rows m .. mpad - 1: the owner's zeros
    };
    {
        T lA[RTC], lB[RTC];
        u2 fl; fl.x = 0u; fl.y = 0u;   // lane k: flag of column base + k
        int base = 0, avail = 0;         // columns [base, avail) are known to be published
        // one column; l: a (possibly incomplete) read of its multipliers when `have`, ln: where the next column's go
        auto left = [&](const int j, T (&l)[RTC], T (&ln)[RTC], const bool have) __attribute__((always_inline)) {
            if (j >= avail) {
                const int cnt = poll_wait(j, C0, fl);
                base = j; avail = j + cnt;
            }
            const unsigned w0 = __builtin_amdgcn_readlane(fl.x, j - base);
            const int wrow = (int)__builtin_amdgcn_readlane(fl.y, j - base);
            const bool act = (w0 >> 16) & 1u;
            if (!have && act) load_l(j, l);
            // the next column's multipliers travel while this one is applied, whether they are there yet or not; when
            // nothing is known about the next column, so does the poll for it
            bool polled = false;
            u2 fn; fn.x = 0u; fn.y = 0u;
            if (j + 1 < C0) {
                load_l(j + 1, ln);
                if (j + 1 >= avail) { fn = poll_issue(j + 1, C0); polled = true; }
            }
            freeze(wrow, j, &s_u[j & 1][0]);
            note_pivot(j, wrow, act);
            __syncthreads();
            if (tid == NT - 1 && j + 1 < C0)   // (not in front of my own columns: they are waited for)
                for (; rp < j; ++rp) replay(rp);
            if (act) {
                int spins = failed ? spin_limit : 0;
                for (;;) {   // every value of my rows written?
                    bool miss = false;
#pragma unroll
                    for (int i = 0; i < RTC; ++i) miss |= unwritten(l[i]);
                    if (!__any(miss)) break;
                    n_waits += 1;
                    if (++spins > spin_limit) {
                        if (!failed) {
                            failed = true;
                            if (lane == 0) atomicExch(status, 1);
                        }
                        break;
                    }
                    load_l(j, l);
                }
                T u[PCW];
#pragma unroll
                for (int c = 0; c < PCW; ++c) u[c] = s_u[j & 1][c];
#pragma unroll
                for (int i = 0; i < RTC; ++i) {   // rows out of the game: l = 0
#pragma unroll
                    for (int c = 0; c < PCW; ++c) a[i][c] = pc_fma(-u[c], l[i], a[i][c]);
                }
            }
            if (polled) {
                const int cnt = poll_count(j + 1, fn);
                if (cnt > 0) { fl = fn; base = j + 1; avail = j + 1 + cnt; }
            }
        };
        for (int j = 0; j < C0; j += 2) {   // C0 is a multiple of four
            left(j, lA, lB, j > 0);
            left(j + 1, lB, lA, true);
        }
    }

    stamp(2);
    if (dbg && tid == 0) dbg[g * 16 + 6] = (unsigned long long)n_waits;
    // ---------------- my columns
    auto own = [&](auto JCt) __attribute__((always_inline)) {
        constexpr int JC = decltype(JCt)::value;
        constexpr int PB = JC & 1;
        const int j = C0 + JC;
        long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0;
        if (dbg) tq0 = clock64();
        // candidates: largest |a|, lowest row on ties (rows of a thread ascend with i).  Rows out of the game hold zeros:
        // they cannot beat a positive maximum, and a maximum of zero is redone with the mask below.
        double nv = -1.0;
        int ni = PC_NONE;
#pragma unroll
        for (int i = 0; i < RTC; ++i) {
            const double av = fabs((double)a[i][JC]);
            const bool better = av > nv;
            nv = better ? av : nv;
            ni = better ? tid + NT * i : ni;
        }
        const unsigned long long kb = (ni != PC_NONE) ? (unsigned long long)__double_as_longlong(nv) : 0ull;
        const int win = pc_argmax64((unsigned)(kb >> 32), (unsigned)kb, ni);
        if (dbg) tq1 = clock64();
        if (win == PC_NONE) {
            if (lane == 0) { u4 k; k.x = 0u; k.y = 0u; k.z = (unsigned)PC_NONE; k.w = 0u; *(u4 *)&s_rec[PB][wave][0] = k; }
        } else {   // (win is in this wave: the candidate of wave w is one of its own rows)
            const int ck = win >> PC_LOGNT;
            const bool mine = (win & (NT - 1)) == tid;
#pragma unroll
            for (int k = 0; k < RTC; ++k)
                if (ck == k && mine) {
                    u4 kk; kk.x = (unsigned)kb; kk.y = (unsigned)(kb >> 32); kk.z = (unsigned)win; kk.w = 0u;
                    *(u4 *)&s_rec[PB][wave][0] = kk;
                    T *rv = (T *)&s_rec[PB][wave][4];
#pragma unroll
                    for (int c = 0; c < PCW; ++c) rv[c] = a[k][c];
                }
        }
        __syncthreads();
        if (dbg) tq2 = clock64();
        // the records -> the winner (every wave for itself: lane w holds the record of wave w)
        u4 rk; rk.x = 0u; rk.y = 0u; rk.z = (unsigned)PC_NONE; rk.w = 0u;
        if (lane < NW) rk = *(const u4 *)&s_rec[PB][lane][0];
        const bool cand = (int)rk.z != PC_NONE;
        int prow = pc_argmax16(cand ? rk.y : 0u, cand ? rk.x : 0u, (int)rk.z);
        T u[PCW];
#pragma unroll
        for (int c = 0; c < PCW; ++c) u[c] = T(0);
        if (__any(cand & ((rk.x | rk.y) != 0u))) {
            const int ww = (prow >> 6) & (NW - 1);   // the wave that holds the pivot row
            const T *rv = (const T *)&s_rec[PB][ww][4];
#pragma unroll
            for (int c = 0; c < PCW; ++c) u[c] = rv[c];
        } else {
            // the largest entry is zero (or there is no candidate): the pivot is the lowest row still in the game
            const unsigned live = ~frozen & ((1u << RTC) - 1u);
            const int ni2 = live ? tid + NT * (int)__builtin_ctz(live) : PC_NONE;
            const int w2 = pc_argmax64(0u, 0u, ni2);
            if (lane == 0) s_rec2[wave] = w2;
            __syncthreads();
            prow = pc_argmax16(0u, 0u, lane < NW ? s_rec2[lane] : PC_NONE);
        }
        const bool valid = prow != PC_NONE;
        const T piv = valid ? u[JC] : T(0);
        const bool act = valid & (piv != T(0));
        if (dbg) tq3 = clock64();
        publish(j, valid ? prow : -1, act ? 1 : 0);   // at once: the others fetch the pivot row while the multipliers are formed
        note_pivot(j, valid ? prow : -1, act);
        freeze(valid ? prow : -1, j, nullptr);
        if (act) {
            const T rabs = fast_recip<T>((T)fabs((double)piv));
            const T rinv = (piv < T(0)) ? -rabs : rabs;   // fast_recip is odd: the same bits as fast_recip(pivot)
            T l[RTC];
#pragma unroll
            for (int i = 0; i < RTC; ++i) {
                l[i] = a[i][JC] * rinv;   // rows out of the game: 0
                a[i][JC] = l[i];
            }
#pragma unroll
            for (int c = JC + 1; c < PCW; ++c) {
#pragma unroll
                for (int i = 0; i < RTC; ++i) a[i][c] = pc_fma(-u[c], l[i], a[i][c]);
            }
            if (g + 1 < G) {   // the multipliers for the workgroups to the right
#pragma unroll
                for (int i = 0; i < RTC; ++i)
                    if (i < nrt) st_l(j, tid + NT * i, l[i]);
            }
        }
        if (dbg) {
            tq4 = clock64();
            if (tid == 0) {
                dbg[g * 16 + 8] += (unsigned long long)(tq1 - tq0); dbg[g * 16 + 9] += (unsigned long long)(tq2 - tq1);
                dbg[g * 16 + 10] += (unsigned long long)(tq3 - tq2); dbg[g * 16 + 11] += (unsigned long long)(tq4 - tq3);
            }
        }
    };
    if (NC > 0) own(std::integral_constant<int, 0>{});
    if (NC > 1) own(std::integral_constant<int, 1>{});
    if (NC > 2) own(std::integral_constant<int, 2>{});
    if (NC > 3) own(std::integral_constant<int, 3>{});
    stamp(3);

    // ---------------- columns right of mine: their pivot rows leave the game (my columns of them hold multipliers: final).
    // No barrier: every wave follows the flags for its own rows, the keeper (alone with the maps; its LDS writes are in
    // order) catches up with the replay.
    {
        int j = C0 + NC;
        while (j < jb) {
            u2 fr;
            const int cnt = poll_wait(j, jb, fr);
            for (int k = 0; k < cnt; ++k) {
                const unsigned w0 = __builtin_amdgcn_readlane(fr.x, k);
                const int wrow = (int)__builtin_amdgcn_readlane(fr.y, k);
                freeze(wrow, j + k, nullptr);
                note_pivot(j + k, wrow, (w0 >> 16) & 1u);
                if (tid == NT - 1) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    for (; rp < j + k; ++rp) replay(rp);
                }
            }
            j += cnt;
        }
    }
    __syncthreads();
    stamp(4);
    if (tid == NT - 1)
        for (; rp < jb; ++rp) replay(rp);
    if (failed && info && lane == 0) atomicMin(info, -0x40000000);   // (any wave that met a time-out)
    __syncthreads();
    // ---- the same permutation as a gather list for the columns outside the panel (kernels_panel_x.hip)
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
    // ---- the rows that never were a pivot: those among the first jb move to the places the pivots left, the others stay
#pragma unroll
    for (int i = 0; i < RTC; ++i) {
        const int gi = tid + NT * i;
        if (((frozen >> i) & 1u) == 0u) {
            const int dest = gi < jb ? s_postop[gi] : gi;
            T *dst = P + (size_t)dest * ldp + C0;
            if (wide) {
                v2t v0, v1;
                v0[0] = a[i][0]; v0[1] = a[i][1]; v1[0] = a[i][2]; v1[1] = a[i][3];
                *(v2t *)(dst) = v0;
                *(v2t *)(dst + 2) = v1;
            } else {
#pragma unroll
                for (int c = 0; c < PCW; ++c)
                    if (c < NC) dst[c] = a[i][c];
            }
        }
    }
    stamp(5);
}

template <typename T, int RTC>
__global__ __launch_bounds__(PC_NT) void panel_c_kernel(int m, int jb, T *__restrict__ P, int ldp, int row0, int col0,
                                                       int32_t *__restrict__ ipiv, int *__restrict__ info, uint2 *flags,
                                                       T *Lbuf, int mpad, int *status, int2 *__restrict__ moves, int *xcc,
                                                       int *xcc_word, int spin_limit, int force_wt, unsigned long long *dbg) {
    if (blockIdx.x & 7) return;
    LSX_TS(1);
    const int G = gridDim.x >> 3, g = blockIdx.x >> 3;
    __shared__ int s_same;
    if (threadIdx.x < 64) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
        const int lane = threadIdx.x;
        if (lane == 0) {
            __hip_atomic_store(&xcc[g], (int)id + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g == 0 && xcc_word) __hip_atomic_store(xcc_word, (int)id + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool pend = lane < G, same = true;
        int spins = 0;
        while (__any(pend)) {
            const int v = __hip_atomic_load(&xcc[lane < G ? lane : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pend && v != 0) { pend = false; same = v == (int)id + 1; }
            if (++spins > spin_limit) { same = false; break; }
        }
        const bool all_same = !__any(!same) && !force_wt;
        if (lane == 0) s_same = all_same ? 1 : 0;
    }
    __syncthreads();
    if (s_same)
        panel_c_body<T, RTC, true>(G, g, m, jb, P, ldp, row0, col0, ipiv, info, flags, Lbuf, mpad, status, moves, spin_limit, dbg);
    else
        panel_c_body<T, RTC, false>(G, g, m, jb, P, ldp, row0, col0, ipiv, info, flags, Lbuf, mpad, status, moves, spin_limit, dbg);
}

// One exchange area for panels of up to m rows: [0, panel_c_ones_offset) is the row-distributed kernel's area (zeroed
// before use; both kernels keep their status word and XCC handshake at its start), behind it flags | Lbuf[128][mpad], which
// is handed over filled with 0xff bytes (panel_c_ones_bytes of them for a panel of m rows).
size_t panel_c_ones_offset(lsx_handle_t h, size_t elem) { return (panel_x_area_bytes(h, 1, elem) + 255) & ~(size_t)255; }
size_t panel_c_ones_bytes(int m, size_t elem) {
    if (m > PC_MAXROWS) m = PC_MAXROWS;
    if (m < 1) return 0;
    const size_t mpad = ((size_t)m + PC_NT - 1) & ~(size_t)(PC_NT - 1);
    return PC_FLAG_BYTES + (size_t)PC_COLS * mpad * elem;
}
size_t panel_c_area_bytes(lsx_handle_t h, int m, size_t elem) {
    const size_t off = panel_c_ones_offset(h, elem);
    return off == 0 ? 0 : (off + panel_c_ones_bytes(m, elem) + 255) & ~(size_t)255;
}

template <typename T, int RTC>
static int panel_col_rt(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info) {
    const int G = (jb + PCW - 1) / PCW;
    const int mpad = (m + PC_NT - 1) & ~(PC_NT - 1);
    const size_t off = panel_c_ones_offset(h, sizeof(T));
    if (off == 0) return 1;
    const size_t need = off + panel_c_ones_bytes(m, sizeof(T));
    const bool driver_clears = h->panel_area_stride > 0;
    const size_t base_off = driver_clears ? (size_t)h->panel_area * h->panel_area_stride : 0;
    if (base_off + need > h->scratch_bytes || (driver_clears && need > h->panel_area_stride)) return 1;   // the row-distributed kernel
    char *base = (char *)h->scratch + base_off;
    int *status = (int *)base;
    int *xcc = (int *)(base + 64);
    uint2 *flags = (uint2 *)(base + off);
    T *Lbuf = (T *)(base + off + PC_FLAG_BYTES);
    static const bool want_dbg = getenv("LSX_PC_DBG") != nullptr;   // stamps behind the multiplier buffer (standalone panels only)
    unsigned long long *dbg = nullptr;
    if (want_dbg && !driver_clears && need + 8192 <= h->scratch_bytes) dbg = (unsigned long long *)(base + ((need + 255) & ~(size_t)255));
    if (dbg) LSX_HIP(hipMemsetAsync(dbg, 0, 4096, h->stream));
    if (!driver_clears) {
        LSX_HIP(hipMemsetAsync(base, 0, 256, h->stream));
        LSX_HIP(hipMemsetAsync(base + off, 0xff, panel_c_ones_bytes(m, sizeof(T)), h->stream));
    }
    hipLaunchKernelGGL((panel_c_kernel<T, RTC>), dim3(8 * G), dim3(PC_NT), 0, h->stream, m, jb, P, ldp, row0, col0, d_ipiv,
                       d_info, flags, Lbuf, mpad, status, (int2 *)h->moves, xcc, h->panel_xcc_word, h->panel_spin_limit, h->panel_col_wt, dbg);
    LSX_HIP(hipGetLastError());
    h->panel_col_launches += 1;
    h->moves_valid = true;
    return LSX_OK;
}

// Returns 1 when the shape is outside what the kernel serves (caller: the row-distributed kernel).
template <typename T>
int panel_col(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info) {
    if (jb > PC_COLS || jb < 1 || m > PC_MAXROWS || m < 1) return 1;
    if (m <= 1 * PC_NT) return panel_col_rt<T, 1>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    if (m <= 2 * PC_NT) return panel_col_rt<T, 2>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    if (m <= 4 * PC_NT) return panel_col_rt<T, 4>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    if (m <= 8 * PC_NT) return panel_col_rt<T, 8>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    return panel_col_rt<T, 16>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
}

template int panel_col<double>(lsx_handle_t, int, int, double *, int, int, int, int32_t *, int *);
template int panel_col<float>(lsx_handle_t, int, int, float *, int, int, int, int32_t *, int *);

}  // namespace lsx

LSX_TS_SETTER(panelc)
