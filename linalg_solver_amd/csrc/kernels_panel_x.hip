// Panel factorisation with the pivot exchange at XCD scope ("panel = 4").
//
// Reference loops covered: pivot search, row swap, scaling and elimination below the pivot for jb
// consecutive pivots (linalg_solver/linalg.py:548-596).  Same arithmetic per element as every other panel
// mode (same multipliers from the same fast_recip, same fused multiply-adds in the same order), so the
// factors and the pivot sequence are bit-identical to theirs.
//
// What is different from kernels_panel_pipe.hip, and why (numbers: tools/kbench.py xchg / panelx):
//  * Scope.  At most 32 workgroups, dealt to ONE XCD (the launch has 8 G workgroups, those with
//    blockIdx % 8 == 0 take part; the others leave at once).  Exchange stores are plain stores that stay in
//    that XCD's L2 and the polling loads bypass L1 only: one hop is 0.26 us instead of 0.50-0.56 us through
//    the fabric.  Placement is verified by a device-scope handshake of XCC ids before column 0; a panel
//    whose participants do not share one id runs the same code with write-through stores.
//  * Cut.  A workgroup is 16 waves and wave w owns the eight panel columns [8w, 8w+8) of the workgroup's 64 RT rows
//    (lane l: rows 64 r + l; 128 registers per lane at four waves per SIMD, the tile takes 64).  PX_WC = 16 selects
//    the first form of this kernel, 8 waves x 16 columns with 256 registers per lane (measured slower).
//    The wave that owns the current column -- the owner wave -- runs the chain header -> winner ->
//    multipliers -> next candidate -> next header; the rank-1 update of the columns right of its own block belongs
//    to the other waves, which follow one barrier behind.
//  * Record.  A candidate is announced as ONE 128-byte record {header, 7 near granules}: the header carries
//    |a|, the row and the SIGN of the candidate, the near granules the candidate row's entries in the rest of
//    the current 8-column block.  With the sign in the header the owner forms 1/pivot and all its multipliers
//    while the winner's near granules are still in flight.  Everything right of the current block is "far":
//    published (one store per wave, the row's entries transposed across lanes through LDS) and fetched by the
//    waves that own those columns, one barrier behind (with PX_WC = 16 the owner wave's own second block included,
//    which it brings up to date after the barrier while its header shot for the next column is in flight).
//  * Arg-max.  The three-phase DPP arg-max runs its first phase only when the high word of |a| already
//    separates the candidates (ties fall back to the full form, so the choice is unchanged).
#include <type_traits>

#include "common.h"
#include "panel_xchg.h"

namespace lsx {

constexpr int PX_WC = 8;      // panel columns per wave: 8 (16 waves, 128 registers per lane) or 16 (8 waves, 256)
constexpr int PX_NT = 64 * (128 / PX_WC);
constexpr int PX_REC = 128;   // bytes of one record: granule 0 = header, granules 1..7 = near values
constexpr int PX_NONE = 0x7fffffff;
// Development builds only (-DLSX_PX_SEG=k, tools/seg_panel.sh): time ONE segment of the owner step -- between
// marks k and k+1 -- with two clock reads, so that the reads' own cost is the same for every segment.
#ifndef LSX_PX_SEG
#define LSX_PX_SEG -1
#endif
#define PX_MARK(k)                                                                  \
    if (LSX_PX_SEG >= 0) {                                                          \
        if (LSX_PX_SEG == (k)) seg_t0 = __builtin_amdgcn_s_memtime();               \
        if (LSX_PX_SEG + 1 == (k)) seg_acc += __builtin_amdgcn_s_memtime() - seg_t0; \
    }

// wave-wide arg-max (largest key, lowest idx on ties); idx == PX_NONE: no candidate (key must be 0 then).
// Returns the winning idx, wave-uniform, or PX_NONE.  All 64 lanes active.
__device__ __forceinline__ int argmax64_fast(unsigned khi, unsigned klo, int idx) {
    const unsigned mhi = rows_max_u32<4>(row16_max_u32(khi), 0);
    const bool top = (khi == mhi) & (idx != PX_NONE);
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(top);
    if (mask == 0ull) return PX_NONE;
    if ((mask & (mask - 1ull)) == 0ull)   // the high word decides
        return __builtin_amdgcn_readlane(idx, __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask)));
    const unsigned mlo = rows_max_u32<4>(row16_max_u32(top ? klo : 0u), 0);
    return rows_min_i32<4>(row16_min_i32((top & (klo == mlo)) ? idx : PX_NONE), 0);
}

template <typename T>
__device__ __forceinline__ unsigned long long value_bits(T v) {
    if (sizeof(T) == 8) return (unsigned long long)__double_as_longlong((double)v);
    return (unsigned long long)__float_as_uint((float)v);
}
template <typename T>
__device__ __forceinline__ T bits_value(unsigned lo, unsigned hi) {
    if (sizeof(T) == 8) return (T)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    return (T)__uint_as_float(lo);
}

// FIRST (the reference's pivot rule, linalg.py:548-552, for the rank-deficient / bar_col < n row reductions of api.hip's
// rref_first): the pivot of a column is not the largest entry but the FIRST row in the current row order whose entry
// exceeds `tol`.  Rows do not move inside this kernel, so every lane carries the current POSITION of its rows (an
// interchange j <-> p sends the row that sat at position j down to p); candidates are ordered by position, the
// position travels in bits 8..21 of the header's epoch word, and a column without a candidate is reported like an
// exactly zero pivot (info = column + 1).  Everything else -- record, exchange, bookkeeping, scatter -- is shared.
template <typename T, int RT, bool DBG, bool XCD, bool FIRST = false>
__device__ __forceinline__ void panel_x_body(const int G, const int g, int m, int jb, T *__restrict__ P, int ldp,
                                             int row0, int col0, int32_t *__restrict__ ipiv,
                                             int *__restrict__ info, char *rec, XGran *far, int *status,
                                             unsigned long long *dbg, int2 *__restrict__ moves, const int spin_limit,
                                             const double tol = -1.0) {
    constexpr int NT = PX_NT, WC = PX_WC;
    constexpr int NW = NT / 64;   // waves
    constexpr int RB = 64 * RT;   // panel rows per workgroup
    constexpr int NONE = PX_NONE;
    __shared__ T s_l[2][RB];                                   // multipliers of column j, by slice-local row
    // x: winner row (-1 none), y: bit0 act, bit1 failed, z: candidate of the next column (slice-local, -1 none)
    __shared__ __attribute__((aligned(16))) int4 s_info[2];
    __shared__ int s_cl2;                                      // candidate chosen at a wave boundary
    __shared__ int s_hist[PC_COLS], s_topid[PC_COLS], s_postop[PC_COLS];
    __shared__ int s_order[RB];
    __shared__ __attribute__((aligned(16))) T s_stage[NW][WC];   // per wave: a row's entries on their way across lanes

    // an earlier panel of this factorisation already failed (exchange time-out): do not spin again
    if (info && *info < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = g * RB;
    const int c0 = WC * wave;           // first panel column of this wave
    const bool has_cols = c0 < jb;
    const int nown = (jb + WC - 1) / WC;   // waves that own columns

    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // DBG: 100 MHz ticks per segment (slot 15 of dbg: XCC id)
    unsigned long long seg_t0 = 0, seg_acc = 0;             // LSX_PX_SEG builds: shader clocks of the chosen segment
    unsigned long long tlast = 0;
#define STAMP(i)                                                              \
    if (DBG && LSX_PX_SEG < 0) {                                              \
        const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime();      \
        seg[i] += tn_ - tlast;                                                \
        tlast = tn_;                                                          \
    }

    // one descriptor per exchange region, shared by the builtin loads and the inline-asm loads / stores
    // far granule rows: a ring of four columns.  The waves that consume them run one barrier behind their
    // owner wave, so a workgroup can still be reading column j - 1 while a faster one publishes column j + 1.
    const __amdgpu_buffer_rsrc_t r_rec = __builtin_amdgcn_make_buffer_rsrc(rec, 0, 2 * G * PX_REC, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_far =
        __builtin_amdgcn_make_buffer_rsrc(far, 0, 4 * G * PC_COLS * (int)sizeof(XGran), 0x00020000);
#define d_rec r_rec
#define d_far r_far

    // ---- load the slice (rows >= m and columns >= jb read as zero)
    T a[RT][WC];
    typedef T v2t __attribute__((ext_vector_type(2)));
    const bool wide = (jb == PC_COLS) && ((size_t)P % 16 == 0) && (ldp % 2 == 0);
    if (wide) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int gi = base + 64 * r + lane;
            const T *src = P + (size_t)(gi < m ? gi : 0) * ldp + c0;
#pragma unroll
            for (int c = 0; c < WC; c += 2) {
                const v2t v = *(const v2t *)(src + c);
                a[r][c] = gi < m ? v[0] : T(0);
                a[r][c + 1] = gi < m ? v[1] : T(0);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int gi = base + 64 * r + lane;
            const T *src = P + (size_t)gi * ldp + c0;
#pragma unroll
            for (int c = 0; c < WC; ++c) a[r][c] = (gi < m && c0 + c < jb) ? src[c] : T(0);
        }
    }
    for (int t = tid; t < PC_COLS; t += NT) { s_topid[t] = t; s_postop[t] = t; }
    for (int t = tid; t < RB; t += NT) s_order[t] = -1;
    unsigned frozen = 0;  // bit r: local row 64 r + lane already used as a pivot (or outside the panel)
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (base + 64 * r + lane >= m) frozen |= 1u << r;
    int pos[FIRST ? RT : 1];   // FIRST: current position of row r of this lane in the panel's row order
    if (FIRST) {
#pragma unroll
        for (int r = 0; r < RT; ++r) pos[r] = base + 64 * r + lane;
    }
    constexpr unsigned EMASK = FIRST ? 0xffu : 0x7fffffffu;   // epoch bits of a header's fourth word
    // the interchange of column j (winner row `win` sat at position wpos): whatever sat at position j goes to wpos
    auto track_swap = [&](const int j, const int win, const int wpos) __attribute__((always_inline)) {
        if (FIRST) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
                if (pos[r] == j && base + 64 * r + lane != win) pos[r] = wpos;
        }
    };
    bool failed = false;
    u4 hA = u4{0u, 0u, 0u, 0u}, hB = u4{0u, 0u, 0u, 0u};   // the two pre-issued header shots
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the tile is in registers before the column loop
    __syncthreads();

    // replay of interchange jj on the position maps (one lane; LAPACK order bookkeeping)
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

    // header shot at column jn, pre-issued (lane q reads the header of workgroup q): the compiler does not
    // know the load is in flight.  Shot A is issued before the step's barrier, shot B right after it and
    // nothing else in between or behind: "at most one outstanding" means A has landed.  B is drained (wait_b)
    // before its registers are read or given up -- a load in flight must never see its registers reassigned.
    auto shot_async = [&](u4 &h, const int jn) __attribute__((always_inline)) {
        const int off = ((jn & 1) * G + (lane < G ? lane : 0)) * PX_REC;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "+v"(h) : "v"(off), "s"(d_rec));
    };
    auto wait_a = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(1)" : "+v"(hA), "+v"(hB)); };
    auto wait_b = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(hA), "+v"(hB)); };
    // Exchange stores are inline asm.  (i) Given builtin stores inside the `row == k` ladders below, hipcc turns a
    // ladder into an indexed read of a stack copy of the whole tile (the tile then lives in scratch for the entire
    // kernel).  (ii) Given the 16-byte store operand as one register tuple, it builds that tuple in place around the
    // tile entry -- every fp64 entry then wants a four-register home and the tile no longer fits.  So the asm takes
    // the four dwords separately and assembles the tuple itself in the top four registers of the budget, which the
    // kernel gives up for this
    // (the budget at two waves per SIMD is 256 registers either way).  s_nop: a store wider than 64 bits reads its
    // data registers for two more cycles.
    auto store16 = [&](const unsigned lo, const unsigned hi, const unsigned z, const unsigned w, const int off,
                       const __amdgpu_buffer_rsrc_t &desc) __attribute__((always_inline)) {
#define LSX_ST16(R0, R1, R2, R3, POL)                                                                                   \
    asm volatile("v_mov_b32 v" #R0 ", %0\n\tv_mov_b32 v" #R1 ", %1\n\tv_mov_b32 v" #R2 ", %2\n\tv_mov_b32 v" #R3 ", %3\n\t"    \
                 "buffer_store_dwordx4 v[" #R0 ":" #R3 "], %4, %5, 0 offen" POL "\n\ts_nop 1"                             \
                 : : "v"(lo), "v"(hi), "s"(z), "v"(w), "v"(off), "s"(desc) : "v" #R0, "v" #R1, "v" #R2, "v" #R3)
        if (WC == 8) {   // 16 waves: the budget is 128 registers
            if (XCD) LSX_ST16(124, 125, 126, 127, ""); else LSX_ST16(124, 125, 126, 127, " sc1");
        } else {
            if (XCD) LSX_ST16(252, 253, 254, 255, ""); else LSX_ST16(252, 253, 254, 255, " sc1");
        }
#undef LSX_ST16
    };
    // header of column jn: {|a| of the candidate (fp64 bits), row (-1: none), epoch jn + 1 | sign << 31}
    auto store_header = [&](const int jn, const double val, const int row, const bool neg, const int rpos = 0)
                            __attribute__((always_inline)) {
        store16((unsigned)__double2loint(val), (unsigned)__double2hiint(val),
                (unsigned)__builtin_amdgcn_readfirstlane(row),
                (unsigned)(jn + 1) | (neg ? 0x80000000u : 0u) | (FIRST ? ((unsigned)rpos << 8) : 0u),
                ((jn & 1) * G + g) * PX_REC, d_rec);
    };
    // granule of column jn: {value bits, epoch jn + 1, 0}
    auto store_gran = [&](const __amdgpu_buffer_rsrc_t &desc, const int off, const T v, const int jn) __attribute__((always_inline)) {
        unsigned lo, hi;
        if (sizeof(T) == 8) { lo = (unsigned)__double2loint((double)v); hi = (unsigned)__double2hiint((double)v); }
        else { lo = __float_as_uint((float)v); hi = 0u; }
        store16(lo, hi, (unsigned)(jn + 1), 0u, off, desc);
    };
    // Owner wave, all lanes: candidates on tile column CN, arg-max, record of column jn = header + near granules
    // (the rest of CN's 8-column block).  Returns the slice-local candidate row (-1: none), wave-uniform.
    auto choose_and_announce = [&](auto CNt, const int jn) __attribute__((always_inline)) -> int {
        constexpr int CN = decltype(CNt)::value;
        double nv = -1.0;
        int ni = NONE;
        int np = NONE;   // FIRST: the candidate's position
#pragma unroll
        for (int r = 0; r < RT; ++r) {   // rows of a lane ascend with r: the first maximum is the lowest row
            const double av = fabs((double)a[r][CN]);
            bool better;
            if (FIRST) better = (((frozen >> r) & 1u) == 0u) & (av > tol) & (pos[r] < np);
            else better = (((frozen >> r) & 1u) == 0u) & (av > nv);
            nv = better ? av : nv;
            ni = better ? base + 64 * r + lane : ni;
            if (FIRST) np = better ? pos[r] : np;
        }
        const unsigned long long kb = (ni != NONE) ? (unsigned long long)__double_as_longlong(nv) : 0ull;
        // FIRST: the key is the position, inverted (the arg-max then finds the first row); positions are unique
        const int win = FIRST ? argmax64_fast((ni != NONE) ? (unsigned)(0x7fffffff - np) : 0u, 0u, ni)
                              : argmax64_fast((unsigned)(kb >> 32), (unsigned)kb, ni);
        const bool have = win != NONE;
        const int cl = have ? win - base : -1;
        if (have) {
            if (lane == (cl & 63)) {
                const int ck = cl >> 6;
                const int roff = ((jn & 1) * G + g) * PX_REC;
#pragma unroll
                for (int k = 0; k < RT; ++k)
                    if (ck == k) {
                        store_header(jn, fabs((double)a[k][CN]), win, a[k][CN] < T(0), FIRST ? pos[k] : 0);
#pragma unroll
                        for (int c = (CN & 7) + 1; c < 8; ++c)
                            store_gran(d_rec, roff + 16 * c, a[k][(CN & 8) + c], jn);
                    }
            }
        } else if (lane == 0) {
            store_header(jn, 0.0, -1, false);
        }
        return cl;
    };
    // Far granules of column jn: this wave's entries [CLO, CLO + NC) of slice-local row cl.  The lane that holds
    // the row writes them to the wave's LDS staging line, NC lanes read one each and store it: ONE store
    // instruction per wave instead of NC single-lane ones (LDS runs in order within a wave: no barrier).
    // All of it inline asm -- see store16 for what hipcc does to tile entries inside a `row == k` ladder.
    const unsigned stage_addr = (unsigned)(unsigned long long)(T __attribute__((address_space(3))) *)&s_stage[wave][0];
    auto publish_far = [&](auto CLOt, auto NCt, const int jn, const int cl_) __attribute__((always_inline)) {
        constexpr int CLO = decltype(CLOt)::value, NC = decltype(NCt)::value;
        const int cl = __builtin_amdgcn_readfirstlane(cl_);
        if (cl < 0 || !has_cols) return;
        if (lane == (cl & 63)) {
            const int ck = cl >> 6;
#pragma unroll
            for (int k = 0; k < RT; ++k)
                if (ck == k) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (sizeof(T) == 8)
                            asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(stage_addr), "v"(a[k][CLO + c]), "n"(8 * c) : "memory");
                        else
                            asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(stage_addr), "v"(a[k][CLO + c]), "n"(4 * c) : "memory");
                    }
                }
        }
        const int q = lane & (NC - 1);
        unsigned lo, hi = 0u;
        if (sizeof(T) == 8) {
            double v;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(stage_addr + 8u * q) : "memory");
            lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
        } else {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo) : "v"(stage_addr + 4u * q) : "memory");
        }
        if (lane < NC)
            store16(lo, hi, (unsigned)(jn + 1), 0u, (((jn & 3) * G + g) * PC_COLS + c0 + CLO + q) * (int)sizeof(XGran), d_far);
    };
    // Rank-1 update of this wave's tile columns [CLO, CLO + NC) for column j: multipliers from LDS, the pivot
    // row's entries from the far granules of the winner's workgroup bg.  false: the granules never came.
    auto far_update = [&](auto CLOt, auto NCt, const int j, const int bg) __attribute__((always_inline)) -> bool {
        constexpr int CLO = decltype(CLOt)::value, NC = decltype(NCt)::value;
        const int off = (((j & 3) * G + bg) * PC_COLS + c0 + CLO + (lane & (NC - 1))) * (int)sizeof(XGran);
        u4 v;
        int spins = failed ? spin_limit : 0;
        for (;;) {
            v = __builtin_amdgcn_raw_buffer_load_b128(r_far, off, opaque_zero(), 16);
            if (!__any(v.z != (unsigned)(j + 1))) break;
            if (++spins > spin_limit) return false;
        }
        T l[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) l[r] = s_l[j & 1][64 * r + lane];
        // a -= l * u with u taken straight from the lane that loaded it: a DPP operand (row_newbcast: lane c of the
        // reader's own 16-lane row; every row holds the NC granules) instead of a readlane into scalar registers per
        // entry.  (-u) * l + a is the same fused multiply-add as a - l * u.  s_nop: a DPP read of a register the
        // previous instruction wrote needs two wait states, which the assembler does not insert inside asm.
        const T un = -bits_value<T>(v.x, v.y);
        asm volatile("s_nop 1" : : "v"(un));
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (sizeof(T) == 8)
                    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                 : "+v"(a[r][CLO + c]) : "v"(un), "v"(l[r]), "n"(c));
                else
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                 : "+v"(a[r][CLO + c]) : "v"(un), "v"(l[r]), "n"(c));
            }
        }
        return true;
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 8> I8;
    typedef std::integral_constant<int, WC> IWC;
    // the step of a wave that does not own column j (owner wave ow): follow the owner one barrier behind
    auto follow = [&](const int j, const int ow) __attribute__((always_inline)) {
        const int par = j & 1;
        const bool stamp = DBG && wave == ow + 1;
        if (stamp) tlast = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if (stamp) STAMP(4)
        const int4 inf = s_info[par];
        const int wrow = inf.x;
        const bool act = (inf.y & 1) != 0;
        failed |= (inf.y & 2) != 0;
        const bool valid = wrow >= 0;
        const int bg = valid ? wrow / RB : 0;
        // bookkeeping on a wave that is not the owner: the last wave while wave 0 owns, wave 0 afterwards
        if (wave == (ow == 0 ? NW - 1 : 0) && lane == 63) {
            s_hist[j] = valid ? (wrow | (act ? 0 : (1 << 30))) : j;
            if (j > 0) replay(j - 1);
            if (valid && bg == g) s_order[wrow - base] = j;
        }
        if (valid && bg == g) {
            const int wl = wrow - base;
            if ((wl & 63) == lane) frozen |= 1u << (wl >> 6);
        }
        if (valid) track_swap(j, wrow, inf.w);
        bool ok = true;
        if (act && wave > ow && has_cols) ok = far_update(I0{}, IWC{}, j, bg);
        if (stamp) STAMP(5)
        if (!ok && !failed) {
            failed = true;
            if (lane == 0) atomicExch(status, 1);
        }
        if (j + 1 < jb) {
            if ((j & 7) < 7) {
                if (wave > ow) publish_far(I0{}, IWC{}, j + 1, inf.z);
            } else {
                // block boundary: the candidate of column j + 1 is chosen behind this step's update -- by the
                // owner wave itself in the middle of its sixteen columns, by wave ow + 1 at their end
                const bool hand_over = (j & (WC - 1)) == WC - 1;
                if (hand_over && wave == ow + 1) {
                    __builtin_amdgcn_s_setprio(3);
                    const int cl = choose_and_announce(I0{}, j + 1);
                    if (lane == 0) s_cl2 = cl;
                    shot_async(hA, j + 1);
                }
                __syncthreads();
                const int nown_ = hand_over ? ow + 1 : ow;
                if (hand_over && wave == ow + 1) {
                    if constexpr (WC == 16) publish_far(I8{}, I8{}, j + 1, s_cl2);
                    shot_async(hB, j + 1);
                }
                else if (wave > nown_) publish_far(I0{}, IWC{}, j + 1, s_cl2);
            }
        }
        if (stamp) STAMP(6)
    };

    // one column in its owner wave; CJ = j & 15 (the tile column) is a compile-time constant
    auto own = [&](auto CJt, const int j) __attribute__((always_inline)) {
        constexpr int CJ = decltype(CJt)::value;
        constexpr int JC = CJ & 7;        // column inside its 8-column block
        constexpr int CB = CJ & 8;        // tile column of the block's first column
        constexpr bool FIRSTB = WC == 16 && CJ < 8;   // the wave's first block: its second block is "far" and follows behind
        constexpr bool NEAR = JC < 7;     // the block has columns right of j
        const int par = j & 1;
        const bool more = j + 1 < jb;
        if (DBG && CJ == 0) tlast = __builtin_amdgcn_s_memrealtime();
        PX_MARK(0)
        // ---------------- O1: all headers of column j, the winner
        bool failed_now = failed;
        bool pend = lane < G;
        unsigned hlo = 0u, hhi = 0u, hneg = 0u;
        int hrow = NONE;
        int hpos = 0;
        auto absorb = [&](const u4 &h) __attribute__((always_inline)) {
            const bool ok = pend & ((h.w & EMASK) == (unsigned)(j + 1));
            hlo = ok ? h.x : hlo;
            hhi = ok ? h.y : hhi;
            hneg = ok ? (h.w >> 31) : hneg;
            if (FIRST) hpos = ok ? (int)((h.w >> 8) & 0x3fffu) : hpos;
            hrow = ok ? ((int)h.z >= 0 ? (int)h.z : NONE) : hrow;
            pend = ok ? false : pend;
        };
        // Shots in flight: in the second block and at a block start A (before the previous barrier) and B (after
        // it); inside the first block only A, and it has landed: the second block's far load behind it was waited for
        constexpr bool TWO = !FIRSTB || JC == 0;
        if (TWO) wait_a(); else wait_b();
        PX_MARK(1)
        absorb(hA);
        if (__any(pend)) {
            if (TWO) { wait_b(); absorb(hB); }
            int spins = failed ? spin_limit : 0;
            while (__any(pend)) {
                if (++spins > spin_limit) { failed_now = true; break; }
                const u4 h = __builtin_amdgcn_raw_buffer_load_b128(
                    r_rec, (par * G + (lane < G ? lane : 0)) * PX_REC, opaque_zero(), 16);
                absorb(h);
            }
        }
        const bool hv = hrow != NONE;
        const int win = FIRST ? argmax64_fast(hv ? (unsigned)(0x7fffffff - hpos) : 0u, 0u, hrow)
                              : argmax64_fast(hv ? hhi : 0u, hv ? hlo : 0u, hrow);
        const bool valid = (win != NONE) & !failed_now;
        const int bg = valid ? win / RB : 0;     // lane bg holds the winner's header
        const int wpos = FIRST ? __builtin_amdgcn_readlane(hpos, bg) : 0;
        const unsigned plo = (unsigned)__builtin_amdgcn_readlane((int)hlo, bg);
        const unsigned phi = (unsigned)__builtin_amdgcn_readlane((int)hhi, bg);
        const bool pneg = __builtin_amdgcn_readlane((int)hneg, bg) != 0;
        const double pabs = __longlong_as_double((long long)(((unsigned long long)phi << 32) | plo));
        PX_MARK(2)
        STAMP(0)
        // ---------------- O2: the winner's near granules -- issued now, consumed after the multipliers, which
        // need only |pivot| and its sign from the header
        const bool act = valid & (pabs != 0.0);
        const bool fetch = NEAR && more && act;
        u4 nq = u4{0u, 0u, 0u, 0u};
        const int noff = (par * G + bg) * PX_REC + 16 * (lane & 7);
        if (fetch) nq = __builtin_amdgcn_raw_buffer_load_b128(r_rec, noff, opaque_zero(), 16);
        const T rabs = act ? fast_recip<T>((T)pabs) : T(0);
        const T rinv = pneg ? -rabs : rabs;   // fast_recip is odd: the same bits as fast_recip(pivot)
        if (valid && bg == g) {
            const int wl = win - base;
            if ((wl & 63) == lane) frozen |= 1u << (wl >> 6);
        }
        if (valid) track_swap(j, win, wpos);
        T l[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const T v = a[r][CJ] * rinv;
            l[r] = ((frozen >> r) & 1u) ? T(0) : v;
            s_l[par][64 * r + lane] = l[r];
            a[r][CJ] = (act & (((frozen >> r) & 1u) == 0u)) ? l[r] : a[r][CJ];
        }
        PX_MARK(3)
        wait_b();   // shot B (unread when A was complete) and the granule load have landed
        bool upd = false;   // the granules are valid: the block takes the update of this column
        if (fetch) {
            int spins = 0;
            for (;;) {
                const int q = lane & 7;
                if (!__any((q > JC) & (nq.z != (unsigned)(j + 1)))) break;
                if (++spins > spin_limit) { failed_now = true; break; }
                nq = __builtin_amdgcn_raw_buffer_load_b128(r_rec, noff, opaque_zero(), 16);
            }
            upd = !failed_now;
        }
        STAMP(1)
        PX_MARK(4)
        if (upd) {   // the same DPP-operand form as far_update: lane c of every 16-lane row holds granule c
            const T un = -bits_value<T>(nq.x, nq.y);
            asm volatile("s_nop 1" : : "v"(un));
#pragma unroll
            for (int c = JC + 1; c < 8; ++c) {
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (sizeof(T) == 8)
                        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                     : "+v"(a[r][CB + c]) : "v"(un), "v"(l[r]), "n"(c));
                    else
                        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                     : "+v"(a[r][CB + c]) : "v"(un), "v"(l[r]), "n"(c));
                }
            }
        }
        PX_MARK(5)
        // ---------------- O3: candidate and record of the next column (at a block end: behind the barrier)
        int cl = -1;
        if (NEAR && more) cl = choose_and_announce(std::integral_constant<int, (CJ < WC - 1 ? CJ + 1 : WC - 1)>{}, j + 1);
        PX_MARK(6)
        const bool act2 = act & !failed_now;
        if (lane == 0) s_info[par] = make_int4(valid ? win : -1, (act2 ? 1 : 0) | (failed_now ? 2 : 0), cl, wpos);
        if (failed_now && !failed) {
            failed = true;
            if (lane == 0) atomicExch(status, 1);
        }
        if (NEAR && more) shot_async(hA, j + 1);
        STAMP(2)
        PX_MARK(7)
        __syncthreads();
        if (!FIRSTB && NEAR && more) shot_async(hB, j + 1);
        PX_MARK(8)
        STAMP(3)
        if constexpr (FIRSTB) { if (more) {
            // the second block follows behind the barrier like any far block; at the end of the first block it is
            // brought up to date BEFORE its first column's candidate is chosen (second barrier: see follow)
            bool ok = true;
            if (act2) ok = far_update(I8{}, I8{}, j, bg);
            if (!ok && !failed) {
                failed = true;
                if (lane == 0) atomicExch(status, 1);
            }
            if (NEAR) {
                publish_far(I8{}, I8{}, j + 1, cl);
            } else {
                const int cl8 = choose_and_announce(I8{}, j + 1);
                if (lane == 0) s_cl2 = cl8;
                shot_async(hA, j + 1);
                __syncthreads();
                shot_async(hB, j + 1);
            }
            STAMP(7)
        } } else if (!NEAR && more) {
            __syncthreads();   // end of the wave's columns: the hand-over barrier (see follow)
        }
    };

    // ---- column 0: wave 0 announces, the others publish their part of its candidate row
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        const int cl = choose_and_announce(I0{}, 0);
        if (lane == 0) s_cl2 = cl;
        shot_async(hA, 0);
    }
    __syncthreads();
    if (wave == 0) {
        if constexpr (WC == 16) publish_far(I8{}, I8{}, 0, s_cl2);
        shot_async(hB, 0);
    }
    else publish_far(I0{}, IWC{}, 0, s_cl2);

    for (int ow = 0; ow < nown; ++ow) {
        const int j0 = WC * ow;
        if (wave == ow) {
#define COL(k) if constexpr (k < WC) { if (j0 + k < jb) own(std::integral_constant<int, k>{}, j0 + k); }
            COL(0) COL(1) COL(2) COL(3) COL(4) COL(5) COL(6) COL(7)
            COL(8) COL(9) COL(10) COL(11) COL(12) COL(13) COL(14) COL(15)
#undef COL
            __builtin_amdgcn_s_setprio(0);
        } else {
            for (int k = 0; k < WC; ++k)
                if (j0 + k < jb) follow(j0 + k, ow);
        }
    }
    if (DBG && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&dbg[g * 16 + i], seg[i]);
    if (DBG && LSX_PX_SEG >= 0 && lane == 0) atomicAdd(&dbg[g * 16 + 12], seg_acc);
#undef STAMP
#undef d_rec
#undef d_far
    __syncthreads();
    if (tid == NT - 1) replay(jb - 1);   // the keeper recorded s_hist[jb - 1]; only its replay is left
    // a workgroup whose exchange timed out reports it through info (negative = protocol failure):
    // the host entry points turn that into LSX_ERR_INTERNAL instead of returning garbage factors
    if (failed && info && lane == 0) atomicMin(info, -0x40000000);
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
    if (has_cols) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int lr = 64 * r + lane;
            const int gi = base + lr;
            if (gi < m) {
                const int ord = s_order[lr];
                const int dest = ord >= 0 ? ord : (gi < jb ? s_postop[gi] : gi);
                T *dst = P + (size_t)dest * ldp + c0;
                if (wide) {
#pragma unroll
                    for (int c = 0; c < WC; c += 2) {
                        v2t v;
                        v[0] = a[r][c]; v[1] = a[r][c + 1];
                        *(v2t *)(dst + c) = v;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < WC; ++c)
                        if (c0 + c < jb) dst[c] = a[r][c];
                }
            }
        }
    }
}

// 8 G workgroups, the G with blockIdx.x % 8 == 0 take part.  `xcc` (G words, zero at launch) is the placement
// handshake: participant g stores 1 + its XCC id device-scope, everybody reads all G (bounded spin) and takes the
// XCD-scope protocol only if all ids agree -- the decision is a function of the same G words for every
// participant, so they all take the same branch.
template <typename T, int RT, bool DBG, bool FIRST = false>
__global__ __launch_bounds__(PX_NT, PX_NT / 256) void panel_x_kernel(int m, int jb, T *__restrict__ P, int ldp, int row0, int col0,
                                                        int32_t *__restrict__ ipiv, int *__restrict__ info,
                                                        char *rec, XGran *far, int *status, unsigned long long *dbg,
                                                        int2 *__restrict__ moves, int *xcc, int *xcc_word,
                                                        int spin_limit, double tol) {
    if (blockIdx.x & 7) return;
    LSX_TS(1);
    const int G = gridDim.x >> 3, g = blockIdx.x >> 3;
    if (spin_limit < 0) {   // fault injection (tests): the last participant shows up ~3 ms late, the others give up
        spin_limit = -spin_limit;
        if (g == G - 1 && G > 1)
            for (int i = 0; i < 900; ++i) __builtin_amdgcn_s_sleep(127);
    }
    __shared__ int s_same;
    if (threadIdx.x < 64) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
        const int lane = threadIdx.x;
        if (lane == 0) {
            __hip_atomic_store(&xcc[g], (int)id + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // for the trailing update that runs beside this panel (gemm_sub_queue_kernel stays off this XCD)
            if (g == 0 && xcc_word) __hip_atomic_store(xcc_word, (int)id + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool pend = lane < G, same = true;
        int spins = 0;
        while (__any(pend)) {
            const int v = __hip_atomic_load(&xcc[lane < G ? lane : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pend && v != 0) { pend = false; same = v == (int)id + 1; }
            if (++spins > spin_limit) { same = false; break; }
        }
        const bool all_same = !__any(!same);
        if (lane == 0) {
            s_same = all_same ? 1 : 0;
            if (DBG && dbg) dbg[g * 16 + 15] = ((unsigned long long)id << 8) | (all_same ? 1u : 0u);
        }
    }
    __syncthreads();
    if (s_same)
        panel_x_body<T, RT, DBG, true, FIRST>(G, g, m, jb, P, ldp, row0, col0, ipiv, info, rec, far, status, dbg, moves, spin_limit, tol);
    else
        panel_x_body<T, RT, DBG, false, FIRST>(G, g, m, jb, P, ldp, row0, col0, ipiv, info, rec, far, status, dbg, moves, spin_limit, tol);
}

// bytes of one exchange area for panels of up to m rows (0: not served)
size_t panel_x_area_bytes(lsx_handle_t h, int m, size_t elem) {
    const int rt = elem == 8 ? 4 : 8;
    if (h->panel_mode != 4 || h->panel_debug || h->nb > PC_COLS) return 0;
    if (m > 32 * 64 * rt) m = 32 * 64 * rt;   // taller panels take the device-scope kernel (its own area size)
    return ((size_t)256 + (size_t)32 * (2 * PX_REC + 4 * PC_COLS * sizeof(XGran)) + 255) & ~(size_t)255;
}

// Returns 1 when the shape is outside what the kernel serves (caller falls back to the device-scope kernel).
template <typename T, int RT>
static int panel_xcd_rt(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info,
                        const double first_tol = -1.0) {
    if (jb > PC_COLS) return 1;
    const int G = (m + 64 * RT - 1) / (64 * RT);
    if (G > 32 || 8 * G > 8 * h->num_cu) return 1;
    // exchange area in scratch: status (+ XCC handshake words at +64) | records[2][G] | far granule rows[4][G][128]
    const size_t rec_bytes = (size_t)2 * G * PX_REC;
    const size_t need = 256 + rec_bytes + (size_t)4 * G * PC_COLS * sizeof(XGran);
    const size_t dbg_off = (need + 255) & ~(size_t)255;
    const size_t total = dbg_off + (h->panel_debug ? (size_t)G * 128 : 0);
    const bool driver_clears = h->panel_area_stride > 0 && !h->panel_debug;
    const size_t base_off = driver_clears ? (size_t)h->panel_area * h->panel_area_stride : 0;
    if (base_off + total > h->scratch_bytes || (driver_clears && need > h->panel_area_stride)) {
        set_error("panel_xcd: scratch too small (%zu + %zu > %zu)", base_off, total, h->scratch_bytes);
        return LSX_ERR_INTERNAL;
    }
    char *base = (char *)h->scratch + base_off;
    int *status = (int *)base;
    int *xcc = (int *)(base + 64);
    char *rec = base + 256;
    XGran *far = (XGran *)(base + 256 + rec_bytes);
    if (!driver_clears) LSX_HIP(hipMemsetAsync(base, 0, h->panel_debug ? total : need, h->stream));
    unsigned long long *dbg = h->panel_debug ? (unsigned long long *)(base + dbg_off) : nullptr;
    if (first_tol >= 0.0) {   // the reference's first-non-zero rule (fp64, up to 8192 rows: callers check)
        if constexpr (sizeof(T) == 8 && RT <= 4)
            hipLaunchKernelGGL((panel_x_kernel<T, RT, false, true>), dim3(8 * G), dim3(PX_NT), 0, h->stream, m, jb, P, ldp, row0,
                               col0, d_ipiv, d_info, rec, far, status, dbg, (int2 *)h->moves, xcc, h->panel_xcc_word, h->panel_spin_limit, first_tol);
        else
            return 1;
    } else if (h->panel_debug)
        hipLaunchKernelGGL((panel_x_kernel<T, RT, true>), dim3(8 * G), dim3(PX_NT), 0, h->stream, m, jb, P, ldp, row0,
                           col0, d_ipiv, d_info, rec, far, status, dbg, (int2 *)h->moves, xcc, h->panel_xcc_word, h->panel_spin_limit, -1.0);
    else
        hipLaunchKernelGGL((panel_x_kernel<T, RT, false>), dim3(8 * G), dim3(PX_NT), 0, h->stream, m, jb, P, ldp, row0,
                           col0, d_ipiv, d_info, rec, far, status, dbg, (int2 *)h->moves, xcc, h->panel_xcc_word, h->panel_spin_limit, -1.0);
    LSX_HIP(hipGetLastError());
    h->moves_valid = true;
    return LSX_OK;
}

// Rows per lane (RT): as few as the panel's height allows with at most 32 workgroups of 64 RT rows -- 1 up to 2048 rows,
// 2 up to 4096, 4 up to 8192, and in fp32 (which has the registers) 8 up to 16384.  The owner wave's per-column work
// (multipliers, update of its block, choice of the next candidate) is proportional to RT and outweighs the dearer
// all-gather of more workgroups: 4 -> 2 -> 1 took 8192^2 fp64 from 17.1 to 16.6 to 16.3 ms and 4096^2 from 6.95 to
// 6.36 to 6.23 ms, 8 -> 4 took 8192^2 fp32 from 16.9 to 14.5 ms.  Same arithmetic per element, same pivots.
template <typename T>
int panel_xcd(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info) {
    if (m <= 32 * 64 * 1) return panel_xcd_rt<T, 1>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    if (m <= 32 * 64 * 2) return panel_xcd_rt<T, 2>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    if (sizeof(T) == 4 && m > 32 * 64 * 4) return panel_xcd_rt<T, sizeof(T) == 4 ? 8 : 4>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    return panel_xcd_rt<T, 4>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
}

// The same panel under the reference's pivot rule (first row in the current order with |a| > tol); fp64, at most 8192
// rows.  Returns 1 for anything else.
int panel_xcd_first(lsx_handle_t h, int m, int jb, double *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info,
                    double tol) {
    if (tol < 0.0 || m > 32 * 64 * 4) return 1;
    if (m <= 32 * 64 * 1) return panel_xcd_rt<double, 1>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info, tol);
    if (m <= 32 * 64 * 2) return panel_xcd_rt<double, 2>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info, tol);
    return panel_xcd_rt<double, 4>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info, tol);
}

template int panel_xcd<double>(lsx_handle_t, int, int, double *, int, int, int, int32_t *, int *);
template int panel_xcd<float>(lsx_handle_t, int, int, float *, int, int, int, int32_t *, int *);

}  // namespace lsx

LSX_TS_SETTER(panelx)
