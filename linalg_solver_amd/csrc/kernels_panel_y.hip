// Panel factorisation with the pivot exchange at XCD scope, second protocol ("panel = 4", option panel_proto = 1).
//
// Reference loops covered: pivot search, row swap, scaling and elimination below the pivot for jb consecutive
// pivots (linalg_solver/linalg.py:548-596).  Same arithmetic per element as every other panel mode (same
// multipliers from the same fast_recip, same fused multiply-adds in the same order), so the factors and the
// pivot sequence are bit-identical to theirs.
//
// Placement, cut of the tile over the waves, far granules, bookkeeping and the final scatter are those of
// kernels_panel_x.hip (read its header first).  What differs is the owner wave's chain, which round 2 measured as
// ONE wave's dependent instruction stream of ~3100 cycles per column with two exchange round trips inside it:
//  * Record.  A candidate is still one 128-byte line of eight 16-byte slots, one per column of the current 8-column
//    block -- the slot of the column itself is the header, the slots right of it the near granules -- but it is
//    written by ONE store instruction (one lane per slot; the candidate row's entries cross the lanes through the wave's
//    LDS staging line) and always in full, also by a workgroup without a candidate.
//  * Poll.  The owner wave fetches the WHOLE record of every workgroup, not the headers alone: lanes 0..31 and
//    32..63 of load i hold slots JC+2i and JC+2i+1 for workgroups 0..31, so 1..4 loads per column depending on the column's place JC in its 8-column block.  Once the winner is
//    known its row entries are already in registers (v_readlane into scalar registers, which the fused
//    multiply-adds take as operands): the second round trip of the first protocol (fetch the winner's near
//    granules) is gone.
//  * Order.  After the arg-max the wave does only what the NEXT announcement needs -- reciprocal, multipliers,
//    update of the rest of its block, choice of the next candidate, the record -- and only then what its own
//    workgroup needs (multipliers and winner into LDS, status, the barrier), so that this part runs while the
//    record travels.
#include <type_traits>

#include "common.h"
#include "panel_xchg.h"

namespace lsx {

constexpr int PY_WC = 8;      // panel columns per wave
constexpr int PY_NT = 64 * (128 / PY_WC);
constexpr int PY_REC = 128;   // bytes of one record: granule 0 = header, granules 1..7 = near values
constexpr int PY_NONE = 0x7fffffff;
// Development switches (tools/var_panel.sh builds one library per setting):
//  PY_POLL  0: the poll of the next column is issued in front of the step's barrier, 1: behind it,
//           2: both, into two register sets (the second is looked at only when the first came back incomplete)
//  PY_RCPALL 1: every lane forms the reciprocal of its header's |a| beside the arg-max (same bits: same function of
//           the same number), the winner's is picked by v_readlane -- takes the reciprocal off the chain
//  PY_TREE  1: the per-lane choice among the RT rows as a tree instead of a chain
#ifndef PY_POLL
#define PY_POLL 0
#endif
#ifndef PY_RCPALL
#define PY_RCPALL 0
#endif
#ifndef PY_TREE
#define PY_TREE 0
#endif
//  PY_OVL   1: the multipliers are written to LDS between the issue of the record's staging read and its wait
#ifndef PY_OVL
#define PY_OVL 0
#endif
// Development builds only (-DLSX_PX_SEG=k, tools/seg_panel.sh): time ONE segment of the owner step -- between
// marks k and k+1 -- with two clock reads, so that the reads' own cost is the same for every segment.
#ifndef LSX_PX_SEG
#define LSX_PX_SEG -1
#endif
#define PY_MARK(k)                                                                  \
    if (LSX_PX_SEG >= 0) {                                                          \
        if (LSX_PX_SEG == (k)) seg_t0 = __builtin_amdgcn_s_memtime();               \
        if (LSX_PX_SEG + 1 == (k)) seg_acc += __builtin_amdgcn_s_memtime() - seg_t0; \
    }

// arg-max over the first NROW 16-lane rows (largest key, lowest idx on ties); idx == PY_NONE: no candidate (key
// must be 0 then).  Returns the winning idx, wave-uniform, or PY_NONE.  All 64 lanes active; lanes outside the
// NROW rows must pass key 0 / PY_NONE.
template <int NROW>
__device__ __forceinline__ int argmax_fast(unsigned khi, unsigned klo, int idx) {
    const unsigned mhi = rows_max_u32<NROW>(row16_max_u32(khi), 0);
    const bool top = (khi == mhi) & (idx != PY_NONE);
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(top);
    if (mask == 0ull) return PY_NONE;
    if ((mask & (mask - 1ull)) == 0ull)   // the high word decides
        return __builtin_amdgcn_readlane(idx, __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask)));
    const unsigned mlo = rows_max_u32<NROW>(row16_max_u32(top ? klo : 0u), 0);
    return rows_min_i32<NROW>(row16_min_i32((top & (klo == mlo)) ? idx : PY_NONE), 0);
}

__device__ __forceinline__ double py_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float py_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T>
__device__ __forceinline__ T py_bits_value(unsigned lo, unsigned hi) {
    if (sizeof(T) == 8) return (T)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    return (T)__uint_as_float(lo);
}

template <typename T, int RT, bool DBG, bool XCD>
__device__ __forceinline__ void panel_y_body(const int G, const int g, int m, int jb, T *__restrict__ P, int ldp,
                                             int row0, int col0, int32_t *__restrict__ ipiv,
                                             int *__restrict__ info, char *rec, XGran *far, int *status,
                                             unsigned long long *dbg, int2 *__restrict__ moves, const int spin_limit) {
    constexpr int NT = PY_NT, WC = PY_WC;
    constexpr int NW = NT / 64;   // waves
    constexpr int RB = 64 * RT;   // panel rows per workgroup
    constexpr int NONE = PY_NONE;
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
    unsigned spin_total = 0;                                // DBG: re-polls of the owner steps
#define STAMP(i)                                                              \
    if (DBG && LSX_PX_SEG < 0) {                                              \
        const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime();      \
        seg[i] += tn_ - tlast;                                                \
        tlast = tn_;                                                          \
    }

    // one descriptor per exchange region.  Far granule rows: a ring of four columns -- the waves that consume them
    // run one barrier behind their owner wave, so a workgroup can still be reading column j - 1 while a faster one
    // publishes column j + 1.
    const __amdgpu_buffer_rsrc_t r_rec = __builtin_amdgcn_make_buffer_rsrc(rec, 0, 2 * G * PY_REC, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_far =
        __builtin_amdgcn_make_buffer_rsrc(far, 0, 4 * G * PC_COLS * (int)sizeof(XGran), 0x00020000);

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
    bool failed = false;
    // the pre-issued poll: entry t = 2 i + (lane >> 5) of the list {header, granule JC+1, ...} of workgroup lane & 31
    u4 R[4] = {u4{0u, 0u, 0u, 0u}, u4{0u, 0u, 0u, 0u}, u4{0u, 0u, 0u, 0u}, u4{0u, 0u, 0u, 0u}};
    u4 Q[4] = {u4{0u, 0u, 0u, 0u}, u4{0u, 0u, 0u, 0u}, u4{0u, 0u, 0u, 0u}, u4{0u, 0u, 0u, 0u}};   // PY_POLL == 2: the second shot
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

    // Poll of column jn whose place in its block is JCN: slots JCN..7 of every workgroup's record, two slots per
    // load (lanes 0..31 | 32..63: workgroup lane & 31), NL loads, in flight when this returns (builtin loads: the
    // compiler counts them and waits in front of their first use, which is behind the step's barrier).  One
    // per-lane base for all of them: slot and parity of the column go into the scalar offset (opaque, so that the
    // loads are re-issued on every trip of a spin loop).
    const int pq = lane & 31, phalf = lane >> 5;
    const bool pin = pq < G;
    const int poll_base = (pin ? pq : 0) * PY_REC + 16 * phalf;
    auto poll_load = [&](u4 (&D)[4], auto JCNt, auto It, const int jn) __attribute__((always_inline)) {
        constexpr int JCN = decltype(JCNt)::value, I = decltype(It)::value;
        // (the last load of an odd list reads one slot past the record in its upper half: never looked at, and in
        // bounds of the descriptor or answered with zeros)
        constexpr int IMM = 16 * JCN + 32 * I;
        int so = (jn & 1) * G * PY_REC + IMM;
        asm volatile("" : "+s"(so));
        D[I] = __builtin_amdgcn_raw_buffer_load_b128(r_rec, poll_base, so, 16);
    };
    auto issue_poll_to = [&](u4 (&D)[4], auto JCNt, const int jn) __attribute__((always_inline)) {
        constexpr int JCN = decltype(JCNt)::value;
        constexpr int NL = (8 - JCN + 1) / 2;
        poll_load(D, JCNt, std::integral_constant<int, 0>{}, jn);
        if constexpr (NL > 1) poll_load(D, JCNt, std::integral_constant<int, 1>{}, jn);
        if constexpr (NL > 2) poll_load(D, JCNt, std::integral_constant<int, 2>{}, jn);
        if constexpr (NL > 3) poll_load(D, JCNt, std::integral_constant<int, 3>{}, jn);
    };
    auto issue_poll = [&](auto JCNt, const int jn) __attribute__((always_inline)) { issue_poll_to(R, JCNt, jn); };
    // PY_POLL == 3: pipelined polling.  The loads are inline asm, so the compiler does not know they are in flight, and
    // every shot lands in the SAME registers R: a later shot read the records later, and a slot that was valid for
    // this column stays as it is until the column after next, so a landing shot can only replace valid entries by
    // identical ones.  Two shots are kept in flight about half a round trip apart (one issued in front of the step's
    // barrier, one behind it, then one per check): a record is noticed half a round trip after it became visible
    // at the latest, not a whole one -- with one outstanding poll every workgroup that just misses the slowest
    // participant's record pays a full round trip, and then IS the slowest participant of the next column.
    // (not a generic lambda: clang rejects asm operands that name captured variables inside one; jcn is a constant
    // at every call site and the lambda is inlined, so the loop folds)
    auto shot_n = [&](const int jcn, const int jn) __attribute__((always_inline)) {
        const int nl = (8 - jcn + 1) / 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nl) {
                const int so = __builtin_amdgcn_readfirstlane((jn & 1) * G * PY_REC + 16 * jcn + 32 * i);
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "+v"(R[i]) : "v"(poll_base), "s"(r_rec), "s"(so));
            }
    };
    auto shot = [&](auto JCNt, const int jn) __attribute__((always_inline)) { shot_n(decltype(JCNt)::value, jn); };
#define PY_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]))
    auto wait_but = [&](const int nl) __attribute__((always_inline)) {   // everything but the newest nl loads has landed
        if (nl == 4) PY_WAIT(4); else if (nl == 3) PY_WAIT(3); else if (nl == 2) PY_WAIT(2); else PY_WAIT(1);
    };
    // the poll around the step's barrier: `before` in front of it, `after` behind it
    auto poll_before = [&](auto JCNt, const int jn) __attribute__((always_inline)) {
        if (PY_POLL == 3) shot(JCNt, jn);
        else if (PY_POLL != 1) issue_poll_to(R, JCNt, jn);
    };
    auto poll_after = [&](auto JCNt, const int jn) __attribute__((always_inline)) {
        if (PY_POLL == 1) issue_poll_to(R, JCNt, jn);
        if (PY_POLL == 2) issue_poll_to(Q, JCNt, jn);
        if (PY_POLL == 3) shot(JCNt, jn);
    };

    // Exchange stores are inline asm with the 16-byte operand assembled in fixed registers: see kernels_panel_x.hip
    // (given builtin stores hipcc keeps the whole tile in 128-bit tuples).  s_nop: a store wider than 64 bits reads
    // its data registers for two more cycles.
    // The address is a per-lane part that never changes (voff) plus a wave-uniform part in a scalar register (soff):
    // per-column vector offsets were kept live by hipcc for both parities of every column and spilled.
    auto store16 = [&](const unsigned lo, const unsigned hi, const unsigned z, const unsigned w, const int voff,
                       const int soff_, const __amdgpu_buffer_rsrc_t &desc) __attribute__((always_inline)) {
        const int soff = __builtin_amdgcn_readfirstlane(soff_);
#define LSX_ST16(POL)                                                                                                   \
    asm volatile("v_mov_b32 v124, %0\n\tv_mov_b32 v125, %1\n\tv_mov_b32 v126, %2\n\tv_mov_b32 v127, %3\n\t"                \
                 "buffer_store_dwordx4 v[124:127], %4, %5, %6 offen" POL "\n\ts_nop 1"                                    \
                 : : "v"(lo), "v"(hi), "v"(z), "v"(w), "v"(voff), "s"(desc), "s"(soff) : "v124", "v125", "v126", "v127")
        if (XCD) LSX_ST16(""); else LSX_ST16(" sc1");
#undef LSX_ST16
    };
    const unsigned stage_addr = (unsigned)(unsigned long long)(T __attribute__((address_space(3))) *)&s_stage[wave][0];
    const int lane16 = 16 * lane;                                          // record slot of a lane < 8
    const int far_voff = (c0 + (lane & (WC - 1))) * (int)sizeof(XGran);   // this wave's far granule of lane & 7
    // Owner wave, all lanes: candidates on tile column CN, arg-max, record of column jn.  The lane that holds the
    // candidate row puts its entries in columns JN..7 of CN's 8-column block on the wave's staging line, lanes JN..7
    // pick one each and store the record with one instruction: slot JN (the column itself) is the header
    // {|a| as fp64 bits, epoch jn + 1 | sign << 31, row (-1: none)}, slots JN+1..7 are granules {value bits, epoch, 0}.
    // Returns the slice-local candidate row (-1: none), wave-uniform.
    // first half: choice, staging writes, the staging read ISSUED (v is not there before record_store's wait)
    auto choose_stage = [&](auto CNt, int &win, bool &have, T &v) __attribute__((always_inline)) -> int {
        constexpr int CN = decltype(CNt)::value;
        constexpr int JN = CN & 7;
        double nv = -1.0;
        int ni = NONE;
        if (PY_TREE && RT >= 2) {
            // pairs first, then pairs of pairs: half the depth of the chain.  Ties keep the lower row (strict >).
            double tv[RT];
            int ti[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const bool fr = ((frozen >> r) & 1u) != 0u;
                tv[r] = fr ? -1.0 : fabs((double)a[r][CN]);
                ti[r] = fr ? NONE : base + 64 * r + lane;
            }
#pragma unroll
            for (int st = 1; st < RT; st *= 2)
#pragma unroll
                for (int r = 0; r + st < RT; r += 2 * st) {
                    const bool better = tv[r + st] > tv[r];
                    tv[r] = better ? tv[r + st] : tv[r];
                    ti[r] = better ? ti[r + st] : ti[r];
                }
            nv = tv[0];
            ni = ti[0];
        } else {
#pragma unroll
            for (int r = 0; r < RT; ++r) {   // rows of a lane ascend with r: the first maximum is the lowest row
                const double av = fabs((double)a[r][CN]);
                const bool better = (((frozen >> r) & 1u) == 0u) & (av > nv);
                nv = better ? av : nv;
                ni = better ? base + 64 * r + lane : ni;
            }
        }
        const unsigned long long kb = (ni != NONE) ? (unsigned long long)__double_as_longlong(nv) : 0ull;
        win = argmax_fast<4>((unsigned)(kb >> 32), (unsigned)kb, ni);
        have = win != NONE;
        const int cl = have ? win - base : -1;
        if (have && lane == (cl & 63)) {
            const int ck = cl >> 6;
#pragma unroll
            for (int k = 0; k < RT; ++k)
                if (ck == k) {
#pragma unroll
                    for (int c = JN; c < 8; ++c) {
                        if (sizeof(T) == 8)
                            asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(stage_addr), "v"(a[k][(CN & 8) + c]), "n"(8 * c) : "memory");
                        else
                            asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(stage_addr), "v"(a[k][(CN & 8) + c]), "n"(4 * c) : "memory");
                    }
                }
        }
        const int q = lane & 7;
        if (sizeof(T) == 8)
            asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(stage_addr + 8u * q) : "memory");
        else
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(stage_addr + 4u * q) : "memory");
        return cl;
    };
    // second half: the staged entries have landed; assemble and store the record of column jn
    auto record_store = [&](auto CNt, const int jn, const int win, const bool have, T v) __attribute__((always_inline)) {
        constexpr int CN = decltype(CNt)::value;
        constexpr int JN = CN & 7;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory");
        if (!have) v = T(0);
        unsigned lo, hi, z = (unsigned)(jn + 1), w = 0u;
        if (lane == JN) {   // slot JN, the column itself: the header
            const double ab = fabs((double)v);
            lo = (unsigned)__double2loint(ab); hi = (unsigned)__double2hiint(ab);
            z |= (v < T(0)) ? 0x80000000u : 0u;
            w = (unsigned)(have ? win : -1);
        } else if (sizeof(T) == 8) {
            lo = (unsigned)__double2loint((double)v); hi = (unsigned)__double2hiint((double)v);
        } else {
            lo = __float_as_uint((float)v); hi = 0u;
        }
        if (lane >= JN && lane < 8) store16(lo, hi, z, w, lane16, ((jn & 1) * G + g) * PY_REC, r_rec);
    };
    auto choose_and_announce = [&](auto CNt, const int jn) __attribute__((always_inline)) -> int {
        int win; bool have; T v;
        const int cl = choose_stage(CNt, win, have, v);
        record_store(CNt, jn, win, have, v);
        return cl;
    };
    // Far granules of column jn: this wave's entries of slice-local row cl.  The lane that holds the row writes them
    // to the wave's LDS staging line, WC lanes read one each and store it: ONE store instruction per wave (LDS runs
    // in order within a wave: no barrier).  granule {value bits, epoch jn + 1, 0}
    auto publish_far = [&](const int jn, const int cl_) __attribute__((always_inline)) {
        const int cl = __builtin_amdgcn_readfirstlane(cl_);
        if (cl < 0 || !has_cols) return;
        if (lane == (cl & 63)) {
            const int ck = cl >> 6;
#pragma unroll
            for (int k = 0; k < RT; ++k)
                if (ck == k) {
#pragma unroll
                    for (int c = 0; c < WC; ++c) {
                        if (sizeof(T) == 8)
                            asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(stage_addr), "v"(a[k][c]), "n"(8 * c) : "memory");
                        else
                            asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(stage_addr), "v"(a[k][c]), "n"(4 * c) : "memory");
                    }
                }
        }
        const int q = lane & (WC - 1);
        unsigned lo, hi = 0u;
        if (sizeof(T) == 8) {
            double v;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(stage_addr + 8u * q) : "memory");
            lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
        } else {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo) : "v"(stage_addr + 4u * q) : "memory");
        }
        if (lane < WC)
            store16(lo, hi, (unsigned)(jn + 1), 0u, far_voff, ((jn & 3) * G + g) * PC_COLS * (int)sizeof(XGran), r_far);
    };
    // Rank-1 update of this wave's tile columns for column j: multipliers from LDS, the pivot row's entries from the
    // far granules of the winner's workgroup bg.  false: the granules never came.
    auto far_update = [&](const int j, const int bg) __attribute__((always_inline)) -> bool {
        u4 v;
        int spins = failed ? spin_limit : 0;
        for (;;) {
            int so = ((j & 3) * G + bg) * PC_COLS * (int)sizeof(XGran);
            asm volatile("" : "+s"(so));
            v = __builtin_amdgcn_raw_buffer_load_b128(r_far, far_voff, so, 16);
            if (!__any(v.z != (unsigned)(j + 1))) break;
            if (++spins > spin_limit) return false;
        }
        T l[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) l[r] = s_l[j & 1][64 * r + lane];
        // a -= l * u with u taken straight from the lane that loaded it: a DPP operand (row_newbcast: lane c of the
        // reader's own 16-lane row; every row holds the WC granules).  (-u) * l + a is the same fused multiply-add
        // as a - l * u.  s_nop: a DPP read of a register the previous instruction wrote needs two wait states.
        const T un = -py_bits_value<T>(v.x, v.y);
        asm volatile("s_nop 1" : : "v"(un));
#pragma unroll
        for (int c = 0; c < WC; ++c) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (sizeof(T) == 8)
                    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                 : "+v"(a[r][c]) : "v"(un), "v"(l[r]), "n"(c));
                else
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                 : "+v"(a[r][c]) : "v"(un), "v"(l[r]), "n"(c));
            }
        }
        return true;
    };
    typedef std::integral_constant<int, 0> I0;
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
        const int bg = __builtin_amdgcn_readfirstlane(valid ? wrow / RB : 0);
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
        bool ok = true;
        if (act && wave > ow && has_cols) ok = far_update(j, bg);
        if (stamp) STAMP(5)
        if (!ok && !failed) {
            failed = true;
            if (lane == 0) atomicExch(status, 1);
        }
        if (j + 1 < jb) {
            if ((j & 7) < 7) {
                if (wave > ow) publish_far(j + 1, inf.z);
            } else {
                // block boundary: the candidate of column j + 1 is chosen behind this step's update by wave ow + 1,
                // which owns that column
                if (wave == ow + 1) {
                    __builtin_amdgcn_s_setprio(3);
                    const int cl = choose_and_announce(I0{}, j + 1);
                    if (lane == 0) s_cl2 = cl;
                    poll_before(I0{}, j + 1);
                }
                __syncthreads();
                if (wave == ow + 1) poll_after(I0{}, j + 1);
                if (wave > ow + 1) publish_far(j + 1, s_cl2);
            }
        }
        if (stamp) STAMP(6)
    };

    // one column in its owner wave; CJ = j & 7 (the tile column) is a compile-time constant
    auto own = [&](auto CJt, const int j) __attribute__((always_inline)) {
        constexpr int CJ = decltype(CJt)::value;
        constexpr bool NEAR = CJ < 7;     // the block has columns right of j
        constexpr int NL = (8 - CJ + 1) / 2;
        const int par = j & 1;
        const bool more = j + 1 < jb;
        if (DBG && CJ == 0) tlast = __builtin_amdgcn_s_memrealtime();
        PY_MARK(0)
        // ---------------- O1: the records of column j (in flight since the previous step), the winner
        bool failed_now = failed;
        {
            int spins = failed ? spin_limit : 0;
            for (;;) {
                if (PY_POLL == 3) wait_but(NL);   // the older of the two shots in flight has landed
                bool bad = false;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const bool need = pin & (2 * i + phalf <= 7 - CJ);
                    bad |= need & ((R[i].z & 0x7fffffffu) != (unsigned)(j + 1));
                }
                if (!__any(bad)) break;
                if (PY_POLL == 3) {
                    if (++spins > spin_limit) { failed_now = true; break; }
                    shot(CJt, j);
                    continue;
                }
                if (PY_POLL == 2 && spins == (failed ? spin_limit : 0)) {   // first miss: the shot from behind the barrier
#pragma unroll
                    for (int i = 0; i < NL; ++i) R[i] = Q[i];
                    ++spins;
                    continue;
                }
                if (++spins > spin_limit) { failed_now = true; break; }
                issue_poll(CJt, j);
            }
            if (DBG) spin_total += (unsigned)(spins - (failed ? spin_limit : 0));
        }
        STAMP(0)
        PY_MARK(1)
        const bool hv = pin & (phalf == 0) & ((int)R[0].w >= 0) & !failed_now;   // slot CJ: the header
        T rall = T(0);
        if (PY_RCPALL) rall = fast_recip<T>((T)__longlong_as_double((long long)(((unsigned long long)R[0].y << 32) | R[0].x)));
        const int win = argmax_fast<2>(hv ? R[0].y : 0u, hv ? R[0].x : 0u, hv ? (int)R[0].w : NONE);
        const bool valid = win != NONE;
        const int bg = valid ? win / RB : 0;     // lane bg holds the winner's header, lane 32 + bg its next granule
        const unsigned plo = (unsigned)__builtin_amdgcn_readlane((int)R[0].x, bg);
        const unsigned phi = (unsigned)__builtin_amdgcn_readlane((int)R[0].y, bg);
        const bool pneg = ((unsigned)__builtin_amdgcn_readlane((int)R[0].z, bg) >> 31) != 0u;
        const double pabs = __longlong_as_double((long long)(((unsigned long long)phi << 32) | plo));
        PY_MARK(2)
        // ---------------- O2: multipliers, update of the rest of the block with the winner's entries from the poll
        const bool act = valid & (pabs != 0.0);
        T rabs;
        if (PY_RCPALL) rabs = act ? readlane_t(rall, bg) : T(0);
        else rabs = act ? fast_recip<T>((T)pabs) : T(0);
        const T rinv = pneg ? -rabs : rabs;   // fast_recip is odd: the same bits as fast_recip(pivot)
        if (valid && bg == g) {
            const int wl = win - base;
            if ((wl & 63) == lane) frozen |= 1u << (wl >> 6);
        }
        T l[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const T v = a[r][CJ] * rinv;
            l[r] = ((frozen >> r) & 1u) ? T(0) : v;
            a[r][CJ] = (act & (((frozen >> r) & 1u) == 0u)) ? l[r] : a[r][CJ];
        }
        PY_MARK(3)
        if (NEAR && more && act) {
#pragma unroll
            for (int c = CJ + 1; c < 8; ++c) {
                const int t = c - CJ, i = t >> 1, src = 32 * (t & 1) + bg;
                const T un = -py_bits_value<T>((unsigned)__builtin_amdgcn_readlane((int)R[i].x, src),
                                               (unsigned)__builtin_amdgcn_readlane((int)R[i].y, src));
#pragma unroll
                for (int r = 0; r < RT; ++r) a[r][c] = py_fma(un, l[r], a[r][c]);
            }
        }
        if (PY_POLL == 3) PY_WAIT(0);   // the shot still in flight has landed (long ago): R's registers may be given up
        STAMP(1)
        PY_MARK(4)
        // ---------------- O3: candidate and record of the next column (at a block end: behind the barrier)
        int cl = -1;
        typedef std::integral_constant<int, (CJ < WC - 1 ? CJ + 1 : WC - 1)> CNX;
        if (PY_OVL && NEAR && more) {
            // the multipliers go to LDS while the staged record entries come back from it
            int win2; bool have2; T v2;
            cl = choose_stage(CNX{}, win2, have2, v2);
#pragma unroll
            for (int r = 0; r < RT; ++r) s_l[par][64 * r + lane] = l[r];
            record_store(CNX{}, j + 1, win2, have2, v2);
        } else {
            if (NEAR && more) cl = choose_and_announce(CNX{}, j + 1);
        }
        PY_MARK(5)
        STAMP(2)
        // ---------------- O4: what this workgroup's other waves need, while the record travels
        if (!(PY_OVL && NEAR && more)) {
#pragma unroll
            for (int r = 0; r < RT; ++r) s_l[par][64 * r + lane] = l[r];
        }
        const bool act2 = act & !failed_now;
        if (lane == 0) s_info[par] = make_int4(valid ? win : -1, (act2 ? 1 : 0) | (failed_now ? 2 : 0), cl, 0);
        if (failed_now && !failed) {
            failed = true;
            if (lane == 0) atomicExch(status, 1);
        }
        PY_MARK(6)
        if (NEAR && more) poll_before(std::integral_constant<int, (CJ < WC - 1 ? CJ + 1 : WC - 1)>{}, j + 1);
        STAMP(3)
        PY_MARK(7)
        __syncthreads();
        if (NEAR && more) poll_after(std::integral_constant<int, (CJ < WC - 1 ? CJ + 1 : WC - 1)>{}, j + 1);
        PY_MARK(8)
        STAMP(7)
        if (!NEAR && more) __syncthreads();   // end of the wave's columns: the hand-over barrier (see follow)
    };

    // ---- column 0: wave 0 announces, the others publish their part of its candidate row
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        const int cl = choose_and_announce(I0{}, 0);
        if (lane == 0) s_cl2 = cl;
        poll_before(I0{}, 0);
    }
    __syncthreads();
    if (wave == 0) poll_after(I0{}, 0);
    if (wave != 0) publish_far(0, s_cl2);

    for (int ow = 0; ow < nown; ++ow) {
        const int j0 = WC * ow;
        if (wave == ow) {
#define COL(k) if (j0 + k < jb) own(std::integral_constant<int, k>{}, j0 + k);
            COL(0) COL(1) COL(2) COL(3) COL(4) COL(5) COL(6) COL(7)
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
    if (DBG && lane == 0) atomicAdd(&dbg[g * 16 + 8], (unsigned long long)spin_total);
#undef STAMP
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
template <typename T, int RT, bool DBG>
__global__ __launch_bounds__(PY_NT, PY_NT / 256) void panel_y_kernel(int m, int jb, T *__restrict__ P, int ldp, int row0, int col0,
                                                        int32_t *__restrict__ ipiv, int *__restrict__ info,
                                                        char *rec, XGran *far, int *status, unsigned long long *dbg,
                                                        int2 *__restrict__ moves, int *xcc, int *xcc_word,
                                                        int spin_limit) {
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
        panel_y_body<T, RT, DBG, true>(G, g, m, jb, P, ldp, row0, col0, ipiv, info, rec, far, status, dbg, moves, spin_limit);
    else
        panel_y_body<T, RT, DBG, false>(G, g, m, jb, P, ldp, row0, col0, ipiv, info, rec, far, status, dbg, moves, spin_limit);
}

// Returns 1 when the shape is outside what the kernel serves (caller falls back to the device-scope kernel).
// The exchange area is the one of the first protocol (panel_x_area_bytes): status | XCC handshake | records[2][G]
// | far granule rows[4][G][128].
template <typename T, int RT>
static int panel_ycd_rt(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info) {
    if (jb > PC_COLS) return 1;
    const int G = (m + 64 * RT - 1) / (64 * RT);
    if (G > 32 || 8 * G > 8 * h->num_cu) return 1;
    const size_t rec_bytes = (size_t)2 * G * PY_REC;
    const size_t need = 256 + rec_bytes + (size_t)4 * G * PC_COLS * sizeof(XGran);
    const size_t dbg_off = (need + 255) & ~(size_t)255;
    const size_t total = dbg_off + (h->panel_debug ? (size_t)G * 128 : 0);
    const bool driver_clears = h->panel_area_stride > 0 && !h->panel_debug;
    const size_t base_off = driver_clears ? (size_t)h->panel_area * h->panel_area_stride : 0;
    if (base_off + total > h->scratch_bytes || (driver_clears && need > h->panel_area_stride)) {
        set_error("panel_ycd: scratch too small (%zu + %zu > %zu)", base_off, total, h->scratch_bytes);
        return LSX_ERR_INTERNAL;
    }
    char *base = (char *)h->scratch + base_off;
    int *status = (int *)base;
    int *xcc = (int *)(base + 64);
    char *rec = base + 256;
    XGran *far = (XGran *)(base + 256 + rec_bytes);
    if (!driver_clears) LSX_HIP(hipMemsetAsync(base, 0, h->panel_debug ? total : need, h->stream));
    unsigned long long *dbg = h->panel_debug ? (unsigned long long *)(base + dbg_off) : nullptr;
    if (h->panel_debug)
        hipLaunchKernelGGL((panel_y_kernel<T, RT, true>), dim3(8 * G), dim3(PY_NT), 0, h->stream, m, jb, P, ldp, row0,
                           col0, d_ipiv, d_info, rec, far, status, dbg, (int2 *)h->moves, xcc, h->panel_xcc_word, h->panel_spin_limit);
    else
        hipLaunchKernelGGL((panel_y_kernel<T, RT, false>), dim3(8 * G), dim3(PY_NT), 0, h->stream, m, jb, P, ldp, row0,
                           col0, d_ipiv, d_info, rec, far, status, dbg, (int2 *)h->moves, xcc, h->panel_xcc_word, h->panel_spin_limit);
    LSX_HIP(hipGetLastError());
    h->moves_valid = true;
    return LSX_OK;
}

// Rows per lane (RT) as in the first protocol: as few as the panel's height allows with at most 32 workgroups.
template <typename T>
int panel_ycd(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info) {
    if (m <= 32 * 64 * 1) return panel_ycd_rt<T, 1>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    if (m <= 32 * 64 * 2) return panel_ycd_rt<T, 2>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    if (sizeof(T) == 4 && m > 32 * 64 * 4) return panel_ycd_rt<T, sizeof(T) == 4 ? 8 : 4>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    return panel_ycd_rt<T, 4>(h, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
}

template int panel_ycd<double>(lsx_handle_t, int, int, double *, int, int, int, int32_t *, int *);
template int panel_ycd<float>(lsx_handle_t, int, int, float *, int, int, int, int32_t *, int *);

}  // namespace lsx

LSX_TS_SETTER(panely)
