// Pipelined cooperative panel factorisation: the same one-launch, register-resident,
// implicit-pivoting scheme as kernels_panel_coop.hip, re-cut so that the chain that limits a
// column -- header visible -> winner known -> pivot row fetched -> multipliers -> next column
// updated -> next candidate -> next header stored -- runs inside ONE wave with no LDS round
// trip and no barrier on it.
//
// Reference loops covered: pivot search, row swap, scaling and elimination below the pivot for
// jb consecutive pivots (linalg_solver/linalg.py:548-596).
//
// Layout.  Workgroup g owns panel rows [g*RB, (g+1)*RB), RB = NTY*RT, in registers: thread
// (tx, ty) = (tid / NTY, tid % NTY) holds rows {NTY*r + ty} x columns {8*tx + c}.  The NTY
// threads that own column j (tx == j>>3) are CONSECUTIVE LANES OF ONE WAVE, the "owner wave".
//
// Per column j:
//   owner wave   reads the G headers of column j (one per lane; the first shot was issued at the end
//                of the previous step), reduces to the winner (largest |a|, lowest row on ties) by
//                fused-DPP integer reductions, fetches the winner's row (128
//                self-validating 16-byte granules), broadcasts the 8 values it needs by readlane,
//                forms the multipliers of its rows, applies them to the rest of its 8-column
//                block, picks the candidate of column j+1 among its lanes (DPP again), stores the
//                header of column j+1 (write-through), leaves row / multipliers / winner in LDS.
//   barrier      one per column.
//   all waves    apply the rank-1 update to the columns right of the owner block, freeze the pivot
//                row, and the thread row holding the new candidate publishes it as granules --
//                all of this while the next header exchange is in flight.
// Every eighth column the next column lives in another thread column.  If its lanes sit in the same
// wave the chain continues there (multipliers through LDS, which is in order within a wave); otherwise
// the new owner wave reads the multipliers from LDS after the barrier and starts the chain there (one
// extra barrier).
//
// Arithmetic per element is the same sequence of fused multiply-adds as the other panel modes
// (same multipliers from the same fast_recip), so the factors are bit-identical to theirs.
// Every spin is bounded; a time-out sets *status and lets the grid drain.
#include <type_traits>

#include "common.h"
#include "panel_xchg.h"

namespace lsx {

constexpr int XLOAD = 16;  // sc1: device scope
constexpr int PP_STAGGER = 4;  // s_sleep units (64 clk) between the two shots at a block start

// KS = header slots per polling lane (G <= 64 * KS)
// XCD = true: the exchange runs at XCD scope.  Only the workgroups with blockIdx.x % 8 == 0 take part (round-robin
// dispatch puts them on one XCD); their header / granule stores are PLAIN stores, which stay in that XCD's L2,
// and the polling loads bypass L1 only (`sc1`), so one hop costs an L2 round trip instead of a trip through
// the fabric.  Placement is verified, never assumed: every participant publishes its XCC id device-scope
// before column 0 and a panel whose participants do not share one id runs the device-scope protocol instead.
template <typename T, int RT, int NT, int KS, bool DBG, bool XCD>
__device__ __forceinline__ void panel_pipe_body(const int G, const int g, int m, int jb, T *__restrict__ P, int ldp,
                                                int row0, int col0, int32_t *__restrict__ ipiv,
                                                int *__restrict__ info, char *hdr, XGran *xrow, int *status,
                                                unsigned long long *dbg, int2 *__restrict__ moves) {
    constexpr int XSTORE = XCD ? 0 : 16;   // cache policy of the exchange stores: plain (L2) or sc1 (write-through)
    constexpr int NTY = NT / 16;   // thread rows; the owners of one column are NTY consecutive lanes
    constexpr int RB = NTY * RT;   // panel rows per workgroup
    constexpr int NONE = 0x7fffffff;
    static_assert(NTY == 16 || NTY == 32, "owner lanes must be one or two DPP rows of one wave");
    __shared__ __attribute__((aligned(16))) T s_u[2][PC_COLS];   // pivot row of column j
    __shared__ T s_l[2][RB];                                        // multipliers of column j
    // x: winner row (-1 none), y: bit0 act, bit1 failed, z: next candidate (slice-local row, -1 none)
    __shared__ __attribute__((aligned(16))) int4 s_info[2];
    __shared__ int s_cl2;          // candidate chosen at a block boundary (slice-local row, -1 none)
    __shared__ int s_hist[PC_COLS], s_topid[PC_COLS], s_postop[PC_COLS];
    __shared__ int s_order[RB];

    __builtin_amdgcn_s_setprio(3);
    // an earlier panel of this factorisation already failed (exchange time-out): do not spin again
    if (info && *info < 0) return;
    const int tid = threadIdx.x, ty = tid % NTY, tx = tid / NTY;
    const int lane = tid & 63, wave = tid >> 6;
    const int base = g * RB;

    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = 0;
#define STAMP(i)                                                              \
    if (DBG) {                                                                \
        const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime();      \
        seg[i] += tn_ - tlast;                                                \
        tlast = tn_;                                                          \
    }

    // The pre-issued header shots and the row-publication stores are inline asm with a hand-counted
    // s_waitcnt: hipcc's own wait insertion degrades to vmcnt(0) in this control flow, which would
    // stall the rank-1 update behind the shots.  Raw buffer descriptors for the asm operands:
    auto raw_desc = [](const void *p, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long b = (unsigned long long)p;
        u4 d;
        d.x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
        d.y = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffffu;
        d.z = (unsigned)__builtin_amdgcn_readfirstlane((int)bytes);
        d.w = 0x00020000u;
        return d;
    };
    const u4 d_hdr = raw_desc(hdr, 2u * G * HDR_STRIDE);
    const u4 d_row = raw_desc(xrow, 2u * G * PC_COLS * (unsigned)sizeof(XGran));
    __amdgpu_buffer_rsrc_t r_hdr = __builtin_amdgcn_make_buffer_rsrc(hdr, 0, 2 * G * HDR_STRIDE, 0x00020000);
    __amdgpu_buffer_rsrc_t r_row =
        __builtin_amdgcn_make_buffer_rsrc(xrow, 0, 2 * G * PC_COLS * (int)sizeof(XGran), 0x00020000);

    // ---- load the slice (rows >= m and columns >= jb read as zero)
    T a[RT][8];
    typedef T v2t __attribute__((ext_vector_type(2)));
    // full-width panel with 16-byte aligned rows: 16-byte accesses (a thread's 8 entries are contiguous)
    const bool wide = (jb == PC_COLS) && ((size_t)P % 16 == 0) && (ldp % 2 == 0);
    if (wide) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int gi = base + NTY * r + ty;
            const T *src = P + (size_t)(gi < m ? gi : 0) * ldp + 8 * tx;
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                const v2t v = *(const v2t *)(src + c);
                a[r][c] = gi < m ? v[0] : T(0);
                a[r][c + 1] = gi < m ? v[1] : T(0);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int gi = base + NTY * r + ty;
            const T *src = P + (size_t)gi * ldp + 8 * tx;
#pragma unroll
            for (int c = 0; c < 8; ++c) a[r][c] = (gi < m && 8 * tx + c < jb) ? src[c] : T(0);
        }
    }
    for (int t = tid; t < PC_COLS; t += NT) { s_topid[t] = t; s_postop[t] = t; }
    for (int t = tid; t < RB; t += NT) s_order[t] = -1;
    unsigned frozen = 0;  // bit r: local row NTY*r+ty already used as a pivot (or outside the panel)
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (base + NTY * r + ty >= m) frozen |= 1u << r;
    bool failed = false;
    // two header shots in flight for the column about to be processed, issued by its owner wave
    // during its previous step: A right after the barrier, B after the rank-1 update, both BEFORE the
    // row-publication stores (returns come back in issue order: a load queued behind write-through
    // stores would wait for their acknowledgements).  Both are always consumed.
    u4 hA[KS], hB[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) { hA[k] = u4{0u, 0u, 0u, 0u}; hB[k] = u4{0u, 0u, 0u, 0u}; }
    // explicit full drain of the tile loads: otherwise hipcc keeps the tile registers "possibly still
    // loading" across the column loop (entry state merged into every trip) and guards the rank-1
    // FMAs of EVERY column with vmcnt(0) -- which would wait for the pre-issued shots
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
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

    // one poll shot at the headers of column jn: lane q reads workgroups q, q+64, ...
    auto shot = [&](u4(&h)[KS], const int jn) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KS; ++k)
            h[k] = __builtin_amdgcn_raw_buffer_load_b128(r_hdr, ((jn & 1) * G + lane + 64 * k) * HDR_STRIDE,
                                                         opaque_zero(), XLOAD);
    };
    // the same shot, pre-issued: the compiler does not know the load is in flight; shots_wait() must
    // run before h is read.  "+v" ties input and output so no copy of h is ever made in between.
    auto shot_async = [&](u4(&h)[KS], const int jn) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int off = ((jn & 1) * G + lane + 64 * k) * HDR_STRIDE;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "+v"(h[k]) : "v"(off), "s"(d_hdr));
        }
    };
    // wait until the shots have landed: returns are in issue order and the operations this wave issued
    // after its last shot are exactly the 8 granule stores of publish_row (always 8, see there), so
    // "at most 8 outstanding" means both shots are complete without waiting for the stores'
    // acknowledgements.  Nothing touches the shot registers between the asm loads and this wait (the
    // "+v" operands tie them in place); the bit-identity tests against panel mode 1 on the GPU are
    // the regression check for this hand-scheduled section.
    auto shots_wait = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KS; ++k) asm volatile("s_waitcnt vmcnt(8)" : "+v"(hA[k]), "+v"(hB[k]));
    };
    // header of column jn: {|a| of the candidate (fp64 bits), row (-1: none), epoch jn+1}
    auto store_header = [&](const int jn, const double val, const int row) __attribute__((always_inline)) {
        const unsigned long long vb = (unsigned long long)__double_as_longlong(val);
        u4 h;
        h.x = (unsigned)vb; h.y = (unsigned)(vb >> 32); h.z = (unsigned)row; h.w = (unsigned)(jn + 1);
        __builtin_amdgcn_raw_buffer_store_b128(h, r_hdr, ((jn & 1) * G + g) * HDR_STRIDE, 0, XSTORE);
    };
    // the thread row holding slice-local row cl publishes it as granules of column jn.  ALWAYS
    // exactly 8 store instructions per wave (every wave holds every thread row): without a candidate
    // (cl < 0) row 0 is stored, which nobody fetches -- shots_wait() counts on the 8.
    auto publish_row = [&](const int jn, const int cl_) __attribute__((always_inline)) {
        const int cl = max(__builtin_amdgcn_readfirstlane(cl_), 0);
        const int cr = cl / NTY;
        if (ty == (cl % NTY)) {
            const int off = ((((jn & 1) * G + g) * PC_COLS) + 8 * tx) * (int)sizeof(XGran);
            u4 v;
            v.z = (unsigned)(jn + 1);
            v.w = 0u;
#pragma unroll
            for (int k = 0; k < RT; ++k)
                if (cr == k) {  // scalar branch; the tile values go through asm operands, which also
                                // keeps hipcc from turning the chain into a switch over a stack copy
                                // of the tile (tile in scratch for the whole kernel)
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        unsigned long long bits;
                        if (sizeof(T) == 8) bits = (unsigned long long)__double_as_longlong((double)a[k][c]);
                        else bits = (unsigned long long)__float_as_uint((float)a[k][c]);
                        v.x = (unsigned)bits;
                        v.y = (unsigned)(bits >> 32);
                        const int offc = off + 16 * c;
                        // s_nop: a store wider than 64 bits reads its data registers for two more
                        // cycles (hipcc pads its own stores; it does not look inside asm)
                        if (XCD)
                            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1"
                                         : : "v"(v), "v"(offc), "s"(d_row));
                        else
                            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1"
                                         : : "v"(v), "v"(offc), "s"(d_row));
                    }
                }
        }
    };
    // candidates of the owner lanes on register column CN, arg-max, header: runs in the owner wave
    // with all lanes active.  Returns the slice-local candidate row (-1: none), wave-uniform.
    auto choose_and_announce = [&](auto CNt, const int jn, const bool own, const int lane0)
                                   __attribute__((always_inline)) -> int {
        constexpr int CN = decltype(CNt)::value;
        double nv = -1.0;
        int ni = NONE;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const double av = fabs((double)a[r][CN]);
            const int gi = base + NTY * r + ty;
            const bool better = own & (((frozen >> r) & 1u) == 0u) & ((av > nv) | ((av == nv) & (gi < ni)));
            nv = better ? av : nv;
            ni = better ? gi : ni;
        }
        const bool mine = ni != NONE;
        const unsigned long long kb = mine ? (unsigned long long)__double_as_longlong(nv) : 0ull;
        const int win = argmax_rows<NTY / 16>((unsigned)(kb >> 32), (unsigned)kb, ni, lane0);
        const bool have = win != NONE;
        const int cl = have ? win - base : -1;
        // the lane that holds the winning row announces it (its nv is that row's |a|); lane0 if none
        if (have ? (own & (ni == win)) : (lane == lane0)) store_header(jn, have ? nv : 0.0, have ? win : -1);
        return cl;
    };
    // start of an 8-column block: column jn (jn & 7 == 0) is up to date in its owner threads
    auto block_start = [&](const int jn) __attribute__((always_inline)) {
        const int txn = jn >> 3;
        const int wn = (txn * NTY) >> 6, lane0 = (txn * NTY) & 63;
        if (wave == wn) {
            const int cl = choose_and_announce(std::integral_constant<int, 0>{}, jn, tx == txn, lane0);
            if (lane == 0) s_cl2 = cl;
        }
        __syncthreads();
        const int cl2 = s_cl2;
        if (wave == wn) { shot_async(hA, jn); __builtin_amdgcn_s_sleep(PP_STAGGER); shot_async(hB, jn); }
        publish_row(jn, cl2);
    };

    // one column; JC = j & 7 is a compile-time constant
    auto column = [&](auto JCt, const int j) __attribute__((always_inline)) {
        constexpr int JC = decltype(JCt)::value;
        const int par = j & 1;
        const int txo = j >> 3;
        const int wo = (txo * NTY) >> 6, lane0 = (txo * NTY) & 63;
        const bool more = j + 1 < jb;
        // block boundary inside one wave: column j + 1 belongs to the next thread column, whose lanes sit
        // in this very wave (3 boundaries in 4 at NT = 256, every other one at NT = 512) -- the chain
        // continues in the wave as within a block, no extra barrier
        const bool same_wave_next = (JC == 7) && more && ((((txo + 1) * NTY) >> 6) == wo);
        if (wave == wo) {
            if (DBG && JC == 0) tlast = __builtin_amdgcn_s_memrealtime();
            // ---------------- O1: all headers of column j (two shots are already in flight)
            double bv = -1.0;
            int bi = NONE;
            bool failed_now = failed;
            {
                // every shot is consumed on every path (also after a time-out): a load left pending
                // on ANY path makes hipcc guard later reuse of its registers with a full vmcnt(0),
                // which would stall the rank-1 update behind the shots of the next column
                unsigned pend = 0;
#pragma unroll
                for (int k = 0; k < KS; ++k)
                    if (lane + 64 * k < G) pend |= 1u << k;
                auto absorb = [&](u4(&h)[KS]) __attribute__((always_inline)) {
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        const bool ok = (((pend >> k) & 1u) != 0u) & (h[k].w == (unsigned)(j + 1));
                        const double hv = __longlong_as_double((long long)(((unsigned long long)h[k].y << 32) | h[k].x));
                        const int hi = (int)h[k].z;
                        const bool better = ok & (hi >= 0) & ((hv > bv) | ((hv == bv) & (hi < bi)));
                        bv = better ? hv : bv;
                        bi = better ? hi : bi;
                        pend = ok ? (pend & ~(1u << k)) : pend;
                    }
                };
                shots_wait();
                absorb(hA);
                absorb(hB);
                int spins = failed ? SPIN_LIMIT : 0;
                while (__any(pend != 0u)) {
                    if (++spins > SPIN_LIMIT) { failed_now = true; break; }
                    shot(hA, j);
                    absorb(hA);
                }
            }
            const unsigned long long kb = (bi != NONE) ? (unsigned long long)__double_as_longlong(bv) : 0ull;
            const int win = argmax_rows<4>((unsigned)(kb >> 32), (unsigned)kb, bi, 0);
            STAMP(0)
            // ---------------- O2: the winner's row (granules lane and lane + 64)
            const bool valid = (win != NONE) & !failed_now;
            const int bg = valid ? win / RB : 0;
            T u0 = T(0), u1 = T(0);
            if (valid) {
                const int roff = (par * G + bg) * PC_COLS * (int)sizeof(XGran);
                int spins = 0;
                for (;;) {
                    const int oz = opaque_zero();
                    const u4 v0 = __builtin_amdgcn_raw_buffer_load_b128(r_row, roff + 16 * lane, oz, XLOAD);
                    const u4 v1 = __builtin_amdgcn_raw_buffer_load_b128(r_row, roff + 16 * (lane + 64), oz, XLOAD);
                    if (!__any((v0.z != (unsigned)(j + 1)) | (v1.z != (unsigned)(j + 1)))) {
                        if (sizeof(T) == 8) {
                            u0 = (T)__longlong_as_double((long long)(((unsigned long long)v0.y << 32) | v0.x));
                            u1 = (T)__longlong_as_double((long long)(((unsigned long long)v1.y << 32) | v1.x));
                        } else {
                            u0 = (T)__uint_as_float(v0.x);
                            u1 = (T)__uint_as_float(v1.x);
                        }
                        break;
                    }
                    if (++spins > SPIN_LIMIT) { failed_now = true; break; }
                }
            }
            s_u[par][lane] = u0;
            s_u[par][lane + 64] = u1;
            STAMP(1)
            // ---------------- O3: multipliers, the rest of the owner block, next candidate, next header
            const int ub = (8 * txo) & 63;
            const T usrc = (txo >= 8) ? u1 : u0;
            T uu[8];
#pragma unroll
            for (int c = JC; c < 8; ++c) uu[c] = readlane_t(usrc, ub + c);
            const T piv = uu[JC];
            const bool act = valid & !failed_now & (piv != T(0));
            const T rinv = act ? fast_recip<T>(piv) : T(0);
            const bool own = tx == txo;
            if (own) {
                if (valid && bg == g) {
                    const int wl = win - base;
                    if ((wl % NTY) == ty) frozen |= 1u << ((wl / NTY) & 31);
                }
                T l[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const T v = a[r][JC] * rinv;
                    l[r] = ((frozen >> r) & 1u) ? T(0) : v;
                    s_l[par][NTY * r + ty] = l[r];
                    a[r][JC] = (act & (((frozen >> r) & 1u) == 0u)) ? l[r] : a[r][JC];
                }
                if (act) {
#pragma unroll
                    for (int c = JC + 1; c < 8; ++c)
#pragma unroll
                        for (int r = 0; r < RT; ++r) a[r][c] -= l[r] * uu[c];
                }
            }
            STAMP(2)
            int cl = -1;
            if (JC < 7 && more) {
                cl = choose_and_announce(std::integral_constant<int, (JC < 7 ? JC + 1 : 7)>{}, j + 1, own, lane0);
            } else if (same_wave_next) {
                // the lanes of thread column txo + 1: multipliers from LDS (written above by this wave:
                // LDS is in order within a wave), the pivot row's entry of column j + 1 by readlane
                const int txn = txo + 1;
                const bool own2 = tx == txn;
                const T un = readlane_t((txn >= 8) ? u1 : u0, (8 * txn) & 63);
                if (own2) {
                    if (valid && bg == g) {
                        const int wl = win - base;
                        if ((wl % NTY) == ty) frozen |= 1u << ((wl / NTY) & 31);
                    }
                    if (act) {
#pragma unroll
                        for (int r = 0; r < RT; ++r) a[r][0] -= s_l[par][NTY * r + ty] * un;
                    }
                }
                cl = choose_and_announce(std::integral_constant<int, 0>{}, j + 1, own2, (txn * NTY) & 63);
            }
            if (lane == 0) s_info[par] = make_int4(valid ? win : -1, (act ? 1 : 0) | (failed_now ? 2 : 0), cl, 0);
            if (failed_now && !failed && lane == 0) atomicExch(status, 1);
            STAMP(3)
        }
        __syncthreads();
        if (wave == wo) STAMP(4)
        // ---------------- everyone: bookkeeping, rank-1 update right of the owner block, row publication
        const int4 inf = s_info[par];
        const int wrow = inf.x;
        const bool act = (inf.y & 1) != 0;
        failed |= (inf.y & 2) != 0;
        const bool valid = wrow >= 0;
        const bool next_here = (JC < 7 && more) || same_wave_next;   // the next column has the same owner wave
        if (next_here && wave == wo) shot_async(hA, j + 1);
        // bookkeeping by a lane of the wave AFTER the owner wave: never on the critical wave, and its
        // global accesses (ipiv, info) never queue behind the owner wave's shots
        if (tid == ((((wo + 1) % (NT / 64)) << 6) | 63)) {
            s_hist[j] = valid ? (wrow | (act ? 0 : (1 << 30))) : j;
            if (j > 0) replay(j - 1);
        }
        if (valid && (wrow / RB) == g) {
            const int wl = wrow - base;
            if ((wl % NTY) == ty) {
                frozen |= 1u << ((wl / NTY) & 31);
                if (tx == 0) s_order[wl] = j;
            }
        }
        if (act && tx > txo) {
            T l[RT], u[8];
#pragma unroll
            for (int r = 0; r < RT; ++r) l[r] = s_l[par][NTY * r + ty];
#pragma unroll
            for (int c = 0; c < 8; ++c) u[c] = s_u[par][8 * tx + c];
            // column 0 of the next thread column is already up to date when its wave handled the boundary
            const bool skip0 = same_wave_next && tx == txo + 1;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                a[r][0] = skip0 ? a[r][0] : a[r][0] - l[r] * u[0];
#pragma unroll
                for (int c = 1; c < 8; ++c) a[r][c] -= l[r] * u[c];
            }
        }
        if (next_here && wave == wo) shot_async(hB, j + 1);
        if (next_here) publish_row(j + 1, inf.z);
        if (wave == wo) STAMP(5)
        if (JC == 7 && more && !same_wave_next) {
            block_start(j + 1);
            if (wave == wo) STAMP(6)
        }
    };

    block_start(0);
    for (int j0 = 0; j0 < jb; j0 += 8) {
#define COL(k) if (j0 + k < jb) column(std::integral_constant<int, k>{}, j0 + k);
        COL(0) COL(1) COL(2) COL(3) COL(4) COL(5) COL(6) COL(7)
#undef COL
    }
    if (DBG && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&dbg[g * 8 + i], seg[i]);
#undef STAMP
    __syncthreads();
    if (tid == NT - 1) replay(jb - 1);
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
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int lr = NTY * r + ty;
        const int gi = base + lr;
        if (gi < m) {
            const int ord = s_order[lr];
            const int dest = ord >= 0 ? ord : (gi < jb ? s_postop[gi] : gi);
            T *dst = P + (size_t)dest * ldp + 8 * tx;
            if (wide) {
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    v2t v;
                    v[0] = a[r][c]; v[1] = a[r][c + 1];
                    *(v2t *)(dst + c) = v;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (8 * tx + c < jb) dst[c] = a[r][c];
            }
        }
    }
}

template <typename T, int RT, int NT, int KS, bool DBG>
__global__ __launch_bounds__(NT, NT / 256) void panel_pipe_kernel(int m, int jb, T *__restrict__ P, int ldp,
                                                                  int row0, int col0,
                                                                  int32_t *__restrict__ ipiv,
                                                                  int *__restrict__ info, char *hdr,
                                                                  XGran *xrow, int *status,
                                                                  unsigned long long *dbg,
                                                                  int2 *__restrict__ moves) {
    panel_pipe_body<T, RT, NT, KS, DBG, false>(gridDim.x, blockIdx.x, m, jb, P, ldp, row0, col0, ipiv, info, hdr, xrow,
                                               status, dbg, moves);
}

// The XCD-scope launch: 8 * G workgroups, the G with blockIdx.x % 8 == 0 take part.  `xcc` (G words, zero at
// launch) is the placement handshake: participant g stores 1 + its XCC id write-through, everybody reads all G
// (bounded spin) and takes the XCD-scope protocol only if all ids agree -- the decision is a function of the
// same G words for every participant, so they all take the same branch.
template <typename T, int RT, int NT, int KS, bool DBG>
__global__ __launch_bounds__(NT, NT / 256) void panel_pipe_xcd_kernel(int m, int jb, T *__restrict__ P, int ldp,
                                                                      int row0, int col0,
                                                                      int32_t *__restrict__ ipiv,
                                                                      int *__restrict__ info, char *hdr,
                                                                      XGran *xrow, int *status,
                                                                      unsigned long long *dbg,
                                                                      int2 *__restrict__ moves, int *xcc) {
    if (blockIdx.x & 7) return;
    const int G = gridDim.x >> 3, g = blockIdx.x >> 3;
    __shared__ int s_same;
    if (threadIdx.x < 64) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(&xcc[g], (int)id + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool pend = lane < G, same = true;
        int spins = 0;
        while (__any(pend)) {
            const int v = __hip_atomic_load(&xcc[lane < G ? lane : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pend && v != 0) { pend = false; same = v == (int)id + 1; }
            if (++spins > SPIN_LIMIT) { same = false; break; }
        }
        const bool all_same = !__any(!same);
        if (lane == 0) {
            s_same = all_same ? 1 : 0;
            if (DBG && dbg) dbg[g * 8 + 7] = ((unsigned long long)id << 8) | (all_same ? 1u : 0u);
        }
    }
    __syncthreads();
    if (s_same)
        panel_pipe_body<T, RT, NT, KS, DBG, true>(G, g, m, jb, P, ldp, row0, col0, ipiv, info, hdr, xrow, status, dbg, moves);
    else
        panel_pipe_body<T, RT, NT, KS, DBG, false>(G, g, m, jb, P, ldp, row0, col0, ipiv, info, hdr, xrow, status, dbg, moves);
}

template <typename T, int RT, int NT, int KS, bool XCDL>
static int panel_pipe_launch(lsx_handle_t h, int G, int m, int jb, T *P, int ldp, int row0, int col0,
                             int32_t *d_ipiv, int *d_info) {
    // exchange area in scratch: status (+ the XCC handshake words at +64) | headers[2][G] (HDR_STRIDE apart) |
    // granule rows[2][G][128]
    const size_t hdr_bytes = (size_t)2 * G * HDR_STRIDE;
    const size_t need = 256 + hdr_bytes + (size_t)2 * G * PC_COLS * sizeof(XGran);
    const size_t dbg_off = (need + 255) & ~(size_t)255;
    const size_t total = dbg_off + (h->panel_debug ? (size_t)G * 64 : 0);
    const bool driver_clears = h->panel_area_stride > 0 && !h->panel_debug;
    const size_t base_off = driver_clears ? (size_t)h->panel_area * h->panel_area_stride : 0;
    if (base_off + total > h->scratch_bytes || (driver_clears && need > h->panel_area_stride)) {
        set_error("panel_pipe: scratch too small (%zu + %zu > %zu)", base_off, total, h->scratch_bytes);
        return LSX_ERR_INTERNAL;
    }
    char *base = (char *)h->scratch + base_off;
    int *status = (int *)base;
    int *xcc = (int *)(base + 64);
    char *hdr = base + 256;
    XGran *xrow = (XGran *)(base + 256 + hdr_bytes);
    // status word, headers AND granules are zero at EVERY launch (epoch 0 never matches): cleared here, or
    // by the look-ahead driver beside the previous panel so that the memset is not on the panel-to-panel chain
    if (!driver_clears) LSX_HIP(hipMemsetAsync(base, 0, h->panel_debug ? total : need, h->stream));
    unsigned long long *dbg = h->panel_debug ? (unsigned long long *)(base + dbg_off) : nullptr;
    if constexpr (XCDL) {
        if (h->panel_debug)
            hipLaunchKernelGGL((panel_pipe_xcd_kernel<T, RT, NT, KS, true>), dim3(8 * G), dim3(NT), 0, h->stream, m, jb, P,
                               ldp, row0, col0, d_ipiv, d_info, hdr, xrow, status, dbg, (int2 *)h->moves, xcc);
        else
            hipLaunchKernelGGL((panel_pipe_xcd_kernel<T, RT, NT, KS, false>), dim3(8 * G), dim3(NT), 0, h->stream, m, jb, P,
                               ldp, row0, col0, d_ipiv, d_info, hdr, xrow, status, dbg, (int2 *)h->moves, xcc);
    } else {
      if (h->panel_debug) {
        hipLaunchKernelGGL((panel_pipe_kernel<T, RT, NT, KS, true>), dim3(G), dim3(NT), 0, h->stream, m, jb, P, ldp,
                           row0, col0, d_ipiv, d_info, hdr, xrow, status, dbg, (int2 *)h->moves);
      } else {
        hipLaunchKernelGGL((panel_pipe_kernel<T, RT, NT, KS, false>), dim3(G), dim3(NT), 0, h->stream, m, jb, P, ldp,
                           row0, col0, d_ipiv, d_info, hdr, xrow, status, dbg, (int2 *)h->moves);
      }
    }
    LSX_HIP(hipGetLastError());
    h->moves_valid = true;
    return LSX_OK;
}

// Returns 1 when the shape is outside what the kernel supports (caller falls back).
template <typename T>
int panel_pipelined(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int col0, int32_t *d_ipiv,
                    int *d_info) {
    if (jb > PC_COLS) return 1;
    // workgroup shape: NT threads (16 thread columns x NT/16 thread rows), RT rows per thread.  One
    // wave per SIMD (NT = 256) keeps the critical wave's issue slots to itself; taller panels take
    // 512 threads so that every workgroup is still resident at once.
    int nt = h->panel_nt, rt = h->panel_rt;
    auto rows = [](int nt_, int rt_) { return nt_ / 16 * rt_; };
    auto wgs = [&](int nt_, int rt_) { return (m + rows(nt_, rt_) - 1) / rows(nt_, rt_); };
    // XCD-scope exchange (option panel_xcd): at most 32 workgroups, one per CU of one XCD -- 256-row slices
    // (512 threads x 8 rows) above 4096 rows, 128-row slices (512 x 4) down to 2048, 64-row slices below
    if (h->panel_xcd && m <= 32 * 256) {
        int xnt = 512, xrt = 8;
        if (m <= 32 * 64) { xnt = 256; xrt = 4; }
        else if (m <= 32 * 128) { xnt = 512; xrt = 4; }
        const int G = wgs(xnt, xrt);
#define LSX_PPX(RT_, NT_)                          \
        if (xrt == RT_ && xnt == NT_)              \
            return panel_pipe_launch<T, RT_, NT_, 1, true>(h, G, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
        LSX_PPX(8, 512) LSX_PPX(4, 512) LSX_PPX(4, 256)
#undef LSX_PPX
    }
    if (h->panel_nt == 0) {  // measured: one header per polling lane (<= 64 workgroups) wins
        nt = 256, rt = 4;
        if (wgs(nt, rt) > 64) nt = 512;
    }
    if (nt != 256 && nt != 512) return 1;
    if (wgs(nt, rt) > h->num_cu) { nt = 512; rt = 8; }  // 256-row slices
    const int G = wgs(nt, rt);
    if (G > h->num_cu || G > 256) return 1;
    const int ks = G <= 64 ? 1 : (G <= 128 ? 2 : 4);
#define LSX_PP(RT_, NT_, KS_)                     \
    if (rt == RT_ && nt == NT_ && ks == KS_)      \
        return panel_pipe_launch<T, RT_, NT_, KS_, false>(h, G, m, jb, P, ldp, row0, col0, d_ipiv, d_info);
    LSX_PP(4, 256, 1) LSX_PP(4, 256, 2) LSX_PP(4, 256, 4)
    LSX_PP(4, 512, 1) LSX_PP(4, 512, 2) LSX_PP(4, 512, 4)
    LSX_PP(8, 512, 1) LSX_PP(8, 512, 2) LSX_PP(8, 512, 4)
#undef LSX_PP
    return 1;
}

size_t panel_pipe_area_bytes(lsx_handle_t h, int m) {
    if (h->panel_mode < 3 || h->panel_debug || h->nb > PC_COLS) return 0;
    if (h->panel_nt != 0 && h->panel_nt != 256 && h->panel_nt != 512) return 0;
    if (!(h->panel_rt == 4 || (h->panel_rt == 8 && h->panel_nt == 512))) return 0;
    if (m > 256 * (h->num_cu < 256 ? h->num_cu : 256)) return 0;
    int G = (m + 127) / 128;        // 128-row slices above 4096 rows, never more than 64 slices below
    if (G < 64) G = 64;
    if (h->panel_nt == 256 && h->panel_rt == 4) G = (m + 63) / 64;
    if (G > 256) G = 256;
    return ((size_t)256 + (size_t)G * (2 * HDR_STRIDE + 2 * PC_COLS * sizeof(XGran)) + 255) & ~(size_t)255;
}

template int panel_pipelined<double>(lsx_handle_t, int, int, double *, int, int, int, int32_t *, int *);
template int panel_pipelined<float>(lsx_handle_t, int, int, float *, int, int, int, int32_t *, int *);

}  // namespace lsx
