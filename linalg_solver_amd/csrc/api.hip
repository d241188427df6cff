// C-ABI of liblsx.so (see include/lsx.h) and the host-side drivers of the
// blocked algorithms.  All device work is enqueued on the handle's stream with
// no host synchronisation inside a factorisation or solve: pivots, info and
// rank stay on the device, so a whole getrf is graph-capturable.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.h"

namespace lsx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int grow(void **p, size_t *have, size_t need) {
    if (need <= *have) return LSX_OK;
    if (*p) {
        LSX_HIP(hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    need = (need + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        *p = nullptr;
        return LSX_ERR_ALLOC;
    }
    *have = need;
    return LSX_OK;
}

int ensure_ws(lsx_handle_t h, size_t bytes) { return grow(&h->ws, &h->ws_bytes, bytes); }

// Every entry point runs with the handle's device current and puts the caller's back on the way out: with two
// handles in one process, or torch's current device elsewhere, workspaces and launches would land on the wrong GPU.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(lsx_handle_t h) {
        if (!h) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define LSX_DEVICE_GUARD(h) lsx::DeviceGuard device_guard_(h)

// The cooperative kernels (panel exchange, few-RHS solve) give up after a bounded spin and say so in a device word;
// a host-buffer entry point must not hand back what was computed after that.  Synchronises the handle's stream.
static int check_dev_status(lsx_handle_t h) {
    int st[3] = {0, 0, 0};
    LSX_HIP(hipStreamSynchronize(h->stream));
    LSX_HIP(hipMemcpy(st, h->dev_status, sizeof(st), hipMemcpyDeviceToHost));
    if (st[0] == 0 && st[1] == 0 && st[2] >= 0) return LSX_OK;
    LSX_HIP(hipMemset(h->dev_status, 0, sizeof(st)));
    set_error("%s timed out on the device (its workgroups were not all resident: another kernel holding the CUs?)",
              st[1] ? "the cooperative triangular solve" : "the panel exchange");
    return LSX_ERR_INTERNAL;
}

static int ensure_scratch(lsx_handle_t h, size_t bytes) {
    // growing frees the old block: make sure nothing queued still uses it
    if (bytes > h->scratch_bytes) LSX_HIP(hipStreamSynchronize(h->stream));
    return grow(&h->scratch, &h->scratch_bytes, bytes);
}

ProfScope::ProfScope(lsx_handle_t h_, int bucket, double flops, double bytes) : h(h_), st(h_->stream) {
    Prof &p = h->prof;
    if (!((p.mask >> bucket) & 1u)) return;
    if (p.sample > 1 && (p.seen[bucket]++ % p.sample) != 0) return;
    ProfEvent ev;
    ev.bucket = bucket;
    for (hipEvent_t *e : {&ev.a, &ev.b}) {
        if (!p.pool.empty()) {
            *e = p.pool.back();
            p.pool.pop_back();
        } else if (hipEventCreate(e) != hipSuccess) {
            return;
        }
    }
    (void)hipEventRecord(ev.a, st);
    p.pending.push_back(ev);
    idx = (int)p.pending.size() - 1;
    p.launches[bucket] += 1;
    p.flops[bucket] += flops;
    p.bytes[bucket] += bytes;
}

ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(h->prof.pending[idx].b, st);
}

// simple bump carving of a device block
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base((char *)b) {}
    template <typename U>
    U *take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        U *p = (U *)(base + off);
        off += count * sizeof(U);
        return p;
    }
};

static size_t pad256(size_t x) { return (x + 255) & ~(size_t)255; }

// row interchanges of the panel that was just factored, applied to `ncols` columns at Acols
template <typename T>
static int apply_panel_swaps(lsx_handle_t h, int ncols, T *Acols, int lda, int row0, int jb,
                             const int32_t *d_ipiv) {
    if (h->moves_valid) return launch_laswp_moves<T>(h, ncols, Acols, lda, row0);
    return launch_laswp<T>(h, ncols, Acols, lda, row0, jb, d_ipiv);
}

// ---------------------------------------------------------------- look-ahead LU driver
// Right-looking LU with look-ahead depth 1.  The update of step k is split: the columns of
// the NEXT panel are updated first, then that panel is factored, all on the high-priority side
// stream, while the main stream updates the rest of the trailing matrix.  The panel chain
// (latency-bound: one cross-CU exchange per column) thus runs underneath the MFMA work.
//   side:  panel(k) | trtri(k) | record P(k) | wait N(k-1) | laswp, trsm, gemm on the next panel's columns | panel(k+1)
//   main:                         wait P(k)  | laswp, trsm, gemm on the other columns | left-hand laswp | record N(k)
// (P(k) is recorded behind the next panel's column block when both streams share the CUs.)
// Disjointness: the side stream's step k touches columns [k+jb, k+2jb); the main stream's step k writes
// columns >= k+2jb and < k and reads L21 (columns [k,k+jb)), U12 (rows [k,k+jb)) and the block inverses.
// (A third variant confined the two streams to disjoint CU sets with hipExtStreamCreateWithCUMask.  A CU mask thins
// out the CUs of EVERY XCD alike and cannot keep a kernel off an XCD, so it never separated the kernels and was
// level with this one at best: removed in round 3, history in DESIGN 6.2.)
// On an error return inside a two-stream driver, work already queued on the side stream must not outlive the
// call: the caller's stream waits for it (the next call on the handle, or the caller reading A, would race otherwise).
struct JoinSide {
    lsx_handle_t h; hipStream_t side, main_s, caller; bool armed = true;
    ~JoinSide() {
        if (!armed) return;
        if (hipEventRecord(h->ev_done, side) == hipSuccess) (void)hipStreamWaitEvent(main_s, h->ev_done, 0);
        if (caller != main_s && hipEventRecord(h->ev_start, main_s) == hipSuccess) (void)hipStreamWaitEvent(caller, h->ev_start, 0);
        h->moves_valid = false;
    }
};

// k0 = first column handled here (columns < k0 were factored by the sequential driver).  k_stop > 0: stop in
// front of the panel that starts at column k_stop (trailing matrix fully updated, that panel not yet factored):
// the XCD-scope driver takes over from there.
template <typename T>
static int getrf_lookahead(lsx_handle_t h, int n, T *A, int lda, int32_t *d_ipiv, int *d_info, T *Tinv,
                           int k0, int k_stop = 0) {
    const int nb = h->nb;
    struct OnSide {  // launches inside this scope go to the given stream
        lsx_handle_t h; hipStream_t keep;
        OnSide(lsx_handle_t h_, hipStream_t s) : h(h_), keep(h_->stream) { h->stream = s; }
        ~OnSide() { h->stream = keep; }
    };
    hipStream_t caller = h->stream;
    hipStream_t main_s = h->stream, side = h->side_stream;
    struct Restore {  // the handle's stream is the update stream while this driver runs
        lsx_handle_t h; hipStream_t keep; int nt, rt, mode;
        ~Restore() { h->stream = keep; h->panel_nt = nt; h->panel_rt = rt; h->panel_mode = mode; h->panel_area_stride = 0; h->panel_area = 0; }
    } restore{h, caller, h->panel_nt, h->panel_rt, h->panel_mode};
    JoinSide join{h, side, main_s, caller};
    // this schedule shares the CUs between the panel and the update: the XCD-scope panel (which fills an XCD) has
    // its own driver, getrf_lookahead_x; here it would stall every launch beside it
    if (h->panel_mode == 4) h->panel_mode = 3;
    // Panels taller than one XCD holds (the phase in front of getrf_lookahead_x): 256-row slices, i.e. half as many
    // workgroups of the device-scope panel.  Its time per column does not depend on the slicing (2.5-2.6 us, kbench
    // panel3tall), but every CU it holds is one the update -- the bottleneck of this phase -- does not have:
    // 16384^2 fp64 77.8 -> 74.6 ms, 12288^2 38.5 -> 37.9 ms, same bits (tools/lu_tall.py).
    if (sizeof(T) == 8 && k_stop > 0 && h->panel_nt == 0 && n - k0 > 8192) { h->panel_nt = 512; h->panel_rt = 8; }
    // Exchange areas of the pipelined panel: two, used alternately, and cleared HERE on the main stream as
    // soon as their panel has finished -- a clear in front of every panel launch sits on the chain.
    size_t area = panel_pipe_area_bytes(h, n - k0);
    if (2 * area > h->scratch_bytes) area = 0;
    if (area) LSX_HIP(hipMemsetAsync(h->scratch, 0, 2 * area, main_s));
    // Counted first tile column of the big update (fp64, this driver as the phase in front of getrf_lookahead_x): the
    // chain of step k then waits, inside its first kernel, for update k-1 on the NEXT panel's columns only and runs
    // beside the rest of that update -- the update of step k follows update k-1 without the chain in between
    // (16384^2: the chain was ~45 us of every one of the 64 steps of this phase).
    const int nst = (n - k0 + nb - 1) / nb;
    int *c0words = nullptr;
    if (sizeof(T) == 8 && k_stop > 0 && area && 2 * area + pad256((size_t)nst * sizeof(int)) <= h->scratch_bytes && !h->x_events) {
        c0words = (int *)((char *)h->scratch + 2 * area);
        LSX_HIP(hipMemsetAsync(c0words, 0, (size_t)nst * sizeof(int), main_s));
    }
    struct ClearCol0 { lsx_handle_t h; ~ClearCol0() { h->gemm_col0_static = nullptr; } } clear_col0{h};
    bool prev_counted = false;
    int prev_tiles = 0;
    h->panel_area_stride = area;
    h->panel_area = 0;
    // the side stream starts after everything already queued on the main stream (info memset, fills)
    LSX_HIP(hipEventRecord(h->ev_start, main_s));
    LSX_HIP(hipStreamWaitEvent(side, h->ev_start, 0));
    // The gather list of panel p lives in moves_buf[p & 1]: panel p+1 (side stream) writes the other
    // buffer while the update stream still applies panel p's interchanges to the left-hand columns,
    // which nothing later depends on and which therefore run AFTER the big update, in its slack.
    struct KeepMoves { lsx_handle_t h; ~KeepMoves() { h->moves = h->moves_buf[0]; } } keep_moves{h};
    // The whole chain panel k -> panel k+1 stays on the side stream (no cross-stream hop on the critical
    // path): inverse of panel k's diagonal blocks, then interchanges / U12 / update of the next panel's own
    // column block, then panel k+1.  The main stream follows one event behind with the other columns.  The
    // side stream must not touch the next panel's columns before the main stream's update of step k-1 has
    // written them (ev_next, recorded long before it is needed: that update runs beside panel k); the block
    // inverses alternate between two buffers because the main stream still reads step k's while the side
    // stream writes step k+1's.
    int step = 0;
    T *Tinv2[2] = {Tinv, Tinv + (size_t)((nb + 63) / 64) * 4096};
    {
        OnSide g(h, side);
        const int jb0 = (n - k0) < nb ? (n - k0) : nb;
        h->moves = h->moves_buf[step & 1];
        LSX_TRY(launch_panel<T>(h, n - k0, jb0, A + (size_t)k0 * lda + k0, lda, k0, d_ipiv + k0, d_info));
    }
    bool have_update = false;   // ev_next holds the end of the previous step's update
    for (int k = k0; k < n; k += nb, ++step) {
        const int jb = (n - k < nb) ? n - k : nb;
        T *Akk = A + (size_t)k * lda + k;
        T *Ti = Tinv2[step & 1];
        const bool mv_valid = h->moves_valid;
        const int rest = n - k - jb;
        if (rest <= 0) {
            LSX_HIP(hipEventRecord(h->ev_panel, side));
            LSX_HIP(hipStreamWaitEvent(main_s, h->ev_panel, 0));  // the last panel is factored
            h->moves = h->moves_buf[step & 1];
            LSX_TRY(apply_panel_swaps<T>(h, k, A, lda, k, jb, d_ipiv + k));
            break;
        }
        T *A12 = A + (size_t)k * lda + k + jb;
        T *L21 = A + (size_t)(k + jb) * lda + k;
        T *A22 = A + (size_t)(k + jb) * lda + k + jb;
        if (k_stop > 0 && k + jb >= k_stop) {   // last step here: no panel ahead, the whole step on the main stream
            LSX_HIP(hipEventRecord(h->ev_panel, side));
            LSX_HIP(hipStreamWaitEvent(main_s, h->ev_panel, 0));
            h->moves = h->moves_buf[step & 1];
            h->moves_valid = mv_valid;
            LSX_TRY(launch_trtri<T>(h, 1, jb, Akk, lda, Ti));
            LSX_TRY(apply_panel_swaps<T>(h, rest, A + k + jb, lda, k, jb, d_ipiv + k));
            LSX_TRY(launch_trsm_block<T>(h, 1, jb, rest, Akk, lda, Ti, A12, lda));
            LSX_TRY(launch_gemm_sub<T>(h, rest, rest, jb, L21, lda, A12, lda, A22, lda));
            LSX_TRY(apply_panel_swaps<T>(h, k, A, lda, k, jb, d_ipiv + k));
            break;
        }
        const int jb2 = rest < nb ? rest : nb;  // width of the next panel
        bool next_valid;
        {
            OnSide g(h, side);
            h->moves = h->moves_buf[step & 1];
            // block inverses and the next block's interchanges in one launch (the main stream starts behind the
            // chain anyway: the CUs are shared)
            if (have_update && !prev_counted) LSX_HIP(hipStreamWaitEvent(side, h->ev_next, 0));
            const int fused = launch_chain_head<T>(h, jb, Akk, lda, Ti, jb2, A + k + jb, lda, k,
                                                   prev_counted ? c0words + (step - 1) : nullptr, prev_tiles);
            if (fused < 0) return fused;
            if (fused == 1 && prev_counted) LSX_TRY(launch_wait_count(h, c0words + (step - 1), prev_tiles));
            if (fused == 1) {
                LSX_TRY(launch_trtri<T>(h, 1, jb, Akk, lda, Ti));
                LSX_TRY(apply_panel_swaps<T>(h, jb2, A + k + jb, lda, k, jb, d_ipiv + k));
            }
            LSX_TRY(launch_trsm_block<T>(h, 1, jb, jb2, Akk, lda, Ti, A12, lda));
            LSX_TRY(launch_gemm_sub<T>(h, rest, jb2, jb, L21, lda, A12, lda, A22, lda));
            // Sharing the CUs, the big update would take the slots these small launches need (measured:
            // the chain doubles); the main stream therefore starts behind the chain.
            LSX_HIP(hipEventRecord(h->ev_panel, side));
            // panel k+1 writes the gather list and exchange area that step k-1 used: behind ALL of update k-1 and its
            // left-hand interchanges (with the counted column the chain above no longer waited for them)
            if (have_update && prev_counted) LSX_HIP(hipStreamWaitEvent(side, h->ev_next, 0));
            h->moves = h->moves_buf[(step + 1) & 1];
            h->panel_area = (step + 1) & 1;
            LSX_TRY(launch_panel<T>(h, rest, jb2, A22, lda, k + jb, d_ipiv + k + jb, d_info));
            next_valid = h->moves_valid;
        }
        LSX_HIP(hipStreamWaitEvent(main_s, h->ev_panel, 0));
        bool cur_counted = false;
        int cur_tiles = 0;
        // panel k is done with its exchange area; panel k+2 reuses it, behind ev_next below
        if (area) LSX_HIP(hipMemsetAsync((char *)h->scratch + (size_t)(step & 1) * area, 0, area, main_s));
        h->moves = h->moves_buf[step & 1];   // panel k's list
        h->moves_valid = mv_valid;
        if (rest > jb2) {
            LSX_TRY(apply_panel_swaps<T>(h, rest - jb2, A + k + jb + jb2, lda, k, jb, d_ipiv + k));
            LSX_TRY(launch_trsm_block<T>(h, 1, jb, rest - jb2, Akk, lda, Ti, A12 + jb2, lda));
            h->gemm_col0_static = c0words ? c0words + step : nullptr;
            const int rg = launch_gemm_sub<T>(h, rest, rest - jb2, jb, L21, lda, A12 + jb2, lda, A22 + jb2, lda);
            cur_counted = c0words && h->gemm_col0_complete;
            cur_tiles = h->gemm_col0_tiles;
            h->gemm_col0_static = nullptr;
            LSX_TRY(rg);
        }
        // panel k's interchanges on the columns left of it, in the slack after the update
        LSX_TRY(apply_panel_swaps<T>(h, k, A, lda, k, jb, d_ipiv + k));
        // recorded after the last reader of panel k's gather list: panel k+2 (launched by the side stream
        // behind its wait on this event) writes the same buffer
        LSX_HIP(hipEventRecord(h->ev_next, main_s));
        have_update = true;
        h->moves_valid = next_valid;
        prev_counted = cur_counted;
        prev_tiles = cur_tiles;
    }
    join.armed = false;   // both streams were joined by the last step
    return LSX_OK;
}

// ---------------------------------------------------------------- look-ahead LU driver, XCD-scope panel
// The panel of step k+1 (kernels_panel_x.hip) fills every CU of ONE XCD while it runs, and the hardware deals an
// eighth of EVERY kernel's workgroups to that XCD whatever the stream or CU mask (measured: lsx_diag_cu_mask_probe),
// so a kernel launched beside a running panel cannot finish before the panel does.  The schedule is built on that:
//   side:  [chain head k: block inverses + panel k's interchanges on the next panel's columns] -> record HEAD
//          -> U12 and update of the next panel's column block -> gate -> panel k+1
//   main:  wait HEAD -> clear panel k's exchange area -> interchanges, U12 and update of all other columns, the
//          update as ONE work-queue kernel whose workgroups on the panel's XCD leave at once and count themselves;
//          the gate in front of panel k+1 waits for that count, i.e. until the update no longer needs the XCD
//   nothing else is launched while a panel runs: the interchanges on the columns LEFT of the panels are applied
//   for the whole factorisation by one launch at the end (one gather list per panel is kept).
// Factors and pivots are bit-identical to the other drivers (same kernels per element, tests).
template <typename T>
static int getrf_lookahead_x(lsx_handle_t h, int n, T *A, int lda, int32_t *d_ipiv, int *d_info, T *Tinv, int k0) {
    const int nb = h->nb;
    const int nsteps = (n - k0 + nb - 1) / nb;
    hipStream_t main_s = h->stream, side = h->side_stream;
    struct OnSide {
        lsx_handle_t h; hipStream_t keep;
        OnSide(lsx_handle_t h_, hipStream_t s) : h(h_), keep(h_->stream) { h->stream = s; }
        ~OnSide() { h->stream = keep; }
    };
    // one exchange area serves either XCD panel: the row-distributed one clears and uses its first area_x bytes, the
    // column-distributed one needs its head (inside those bytes) cleared and the multiplier buffer behind it as it is
    const size_t area_x = panel_x_area_bytes(h, n - k0, sizeof(T));
    const size_t area_c = h->panel_col ? panel_c_area_bytes(h, n - k0, sizeof(T)) : 0;
    const size_t area = area_x == 0 ? 0 : (area_c > area_x ? area_c : area_x);
    const size_t pass_bytes = pad256(256 + (size_t)nsteps * sizeof(int)), ctr_bytes = pad256((size_t)nsteps * 8 * sizeof(int));
    const size_t col0_bytes = pad256((size_t)nsteps * 2 * sizeof(int));
    const size_t words = pass_bytes + ctr_bytes + col0_bytes + pad256((size_t)nsteps * sizeof(int));
    if (area == 0 || 3 * area + words > h->scratch_bytes) { set_error("getrf_lookahead_x: scratch"); return LSX_ERR_INTERNAL; }
    LSX_TRY(grow(&h->moves_all, &h->moves_all_bytes, (size_t)nsteps * 2048));
    struct Restore {
        lsx_handle_t h; hipStream_t keep;
        ~Restore() {
            h->stream = keep; h->panel_area_stride = 0; h->panel_area = 0; h->moves = h->moves_buf[0];
            h->gemm_queue = 0; h->gemm_counters = nullptr; h->gemm_avoid_word = nullptr; h->gemm_pass_word = nullptr;
            h->gemm_col0 = nullptr; h->panel_xcc_word = nullptr; h->chain_info = nullptr;
        }
    } restore{h, main_s};
    h->chain_info = d_info;
    JoinSide join{h, side, main_s, main_s};
    // three exchange areas in rotation (panel j uses area j % 3, cleared behind update j), then the words
    char *wbase = (char *)h->scratch + 3 * area;
    int *xcc_word = (int *)wbase;                    // 1 + XCC id of the panel's XCD (blocks = 0 mod 8 land on XCC 0)
    int *pass = (int *)(wbase + 256);                // per step: update workgroups that left the panel's XCD
    int *counters = (int *)(wbase + pass_bytes);     // per step: the update's eight strip queues
    int *col0 = (int *)(wbase + pass_bytes + ctr_bytes);   // per step: {ticket, finished tiles} of the update's tile column 0
    int *ready = (int *)(wbase + pass_bytes + ctr_bytes + col0_bytes);   // per step: block inverses finished (fused chain launch)
    for (int s = 0; s < 3; ++s) {
        LSX_HIP(hipMemsetAsync((char *)h->scratch + (size_t)s * area, 0, area_x, main_s));
        if (area_c && s < nsteps)   // the column-distributed kernel's flags and multiplier buffer: "unwritten"
            LSX_HIP(hipMemsetAsync((char *)h->scratch + (size_t)s * area + panel_c_ones_offset(h, sizeof(T)), 0xff,
                                   panel_c_ones_bytes(n - k0 - s * nb, sizeof(T)), main_s));
    }
    LSX_HIP(hipMemsetAsync(wbase, 0, words, main_s));
    LSX_HIP(hipMemsetD32Async((hipDeviceptr_t)xcc_word, 1, 1, main_s));
    h->panel_area_stride = area;
    h->panel_xcc_word = xcc_word;
    LSX_HIP(hipEventRecord(h->ev_start, main_s));
    LSX_HIP(hipStreamWaitEvent(side, h->ev_start, 0));
    auto list = [&](int s) { return (void *)((char *)h->moves_all + (size_t)s * 2048); };
    T *Tinv2[2] = {Tinv, Tinv + (size_t)((nb + 63) / 64) * 4096};
    {
        OnSide g(h, side);
        h->moves = list(0);
        h->panel_area = 0;
        LSX_TRY(launch_panel<T>(h, n - k0, (n - k0) < nb ? (n - k0) : nb, A + (size_t)k0 * lda + k0, lda, k0, d_ipiv + k0, d_info));
        if (!h->moves_valid) { set_error("getrf_lookahead_x: panel without a gather list"); return LSX_ERR_INTERNAL; }
    }
    // What the chain of step k needs from the main stream is update k-1 on the NEXT panel's columns only.  When that
    // update went through the work queue with no edge launches, its kernel does those columns (tile column 0) first
    // and counts their finished tiles; the chain then waits for the count inside a one-wave kernel: the panels run
    // ahead of a long update (the first ~15 steps of an 8192^2 LU are bound by the update, which then follows
    // back to back instead of alternating with the chain), and no gate is needed in front of a panel that leaves
    // CUs of its XCD free -- update workgroups held up behind a running panel only delay the END of the update
    // kernel, which nothing on the chain waits for any more.  Otherwise (ragged shapes): the event behind the whole
    // update, and the gate.  (What this cannot buy: the chain's small kernels overlapping a RUNNING update.  The
    // update's persistent workgroups fill every CU of seven XCDs and the hardware deals every kernel's workgroups
    // round-robin over all eight, so a chain kernel launched early waits for update workgroups to leave:
    // tools/ts_lu.py shows chain heads of 130 us in the first steps.)
    const bool use_col0 = !h->x_events;
    bool prev_col0 = false;
    int prev_tiles = 0;
    int step = 0;
    int left_done = 0;   // panels [0, left_done) have had their interchanges applied to the columns left of them
    for (int k = k0; k < n; k += nb, ++step) {
        const int jb = (n - k < nb) ? n - k : nb;
        const int rest = n - k - jb;
        if (rest <= 0) break;
        T *Akk = A + (size_t)k * lda + k;
        T *Ti = Tinv2[step & 1];
        T *A12 = A + (size_t)k * lda + k + jb;
        T *L21 = A + (size_t)(k + jb) * lda + k;
        T *A22 = A + (size_t)(k + jb) * lda + k + jb;
        const int jb2 = rest < nb ? rest : nb;  // width of the next panel
        bool fused_all = false;
        {
            OnSide g(h, side);
            // update k-1 wrote the next panel's columns: its column-0 count (waited for inside the chain head), or the
            // event behind all of it
            const bool counted = step > 0 && prev_col0;
            if (step > 0 && !counted) LSX_HIP(hipStreamWaitEvent(side, h->ev_next, 0));
            h->moves = list(step);
            h->moves_valid = true;
            // block inverses, interchanges and U12 of the next panel's columns in one launch where the shapes allow it
            const int all = launch_chain_fused<T>(h, jb, Akk, lda, Ti, jb2, A + k + jb, lda, k,
                                                  counted ? col0 + 2 * (step - 1) + 1 : nullptr, prev_tiles, ready + step);
            if (all < 0) return all;
            if (all == 1) {
                const int fused = launch_chain_head<T>(h, jb, Akk, lda, Ti, jb2, A + k + jb, lda, k,
                                                       counted ? col0 + 2 * (step - 1) + 1 : nullptr, prev_tiles);
                if (fused < 0) return fused;
                if (fused == 1) {
                    if (counted) LSX_TRY(launch_wait_count(h, col0 + 2 * (step - 1) + 1, prev_tiles));
                    LSX_TRY(launch_trtri<T>(h, 1, jb, Akk, lda, Ti));
                    LSX_TRY(launch_laswp_moves<T>(h, jb2, A + k + jb, lda, k));
                }
            }
            // HEAD: block inverses of panel k are there.  The fused launch counts them in a device word the main stream
            // waits for in a one-wave kernel: an event record here would be a marker packet BETWEEN the chain's launches
            // (6 us on the panel-to-panel path -- what the fusion itself saved)
            fused_all = all == LSX_OK;
            if (!fused_all) {
                LSX_HIP(hipEventRecord(h->ev_panel, side));
                LSX_TRY(launch_trsm_block<T>(h, 1, jb, jb2, Akk, lda, Ti, A12, lda));
            }
            LSX_TRY(launch_gemm_sub<T>(h, rest, jb2, jb, L21, lda, A12, lda, A22, lda));
        }
        // main stream, behind HEAD.  Measured alternatives (8192^2 / 4096^2, ms): this order 18.2 / 7.6; interchanges
        // ahead of HEAD and the area cleared by the update's idle workgroups, so that the update starts earlier,
        // 18.5 / 7.9 (its resident workgroups take the CUs the next panel's small update needs: 27 instead of 17 us);
        // the update ordered behind that small update by an event 19.0 / 8.0 (a cross-stream hop on the chain).
        if (fused_all) LSX_TRY(launch_wait_count(h, ready + step, 2));
        else LSX_HIP(hipStreamWaitEvent(main_s, h->ev_panel, 0));
        int queued = 0;
        bool cur_col0 = false;
        int cur_tiles = 0;
        if (rest > jb2) {
            h->moves = list(step);
            h->moves_valid = true;
            LSX_TRY(launch_laswp_moves<T>(h, rest - jb2, A + k + jb + jb2, lda, k));
            LSX_TRY(launch_trsm_block<T>(h, 1, jb, rest - jb2, Akk, lda, Ti, A12 + jb2, lda));
            h->gemm_queue = 1;
            h->gemm_counters = counters + 8 * step;
            h->gemm_counter_sets = 1;
            h->gemm_counter_set = 0;
            h->gemm_avoid_word = xcc_word;
            h->gemm_pass_word = pass + step;
            h->gemm_col0 = use_col0 ? col0 + 2 * step : nullptr;
            const int rq = launch_gemm_sub<T>(h, rest, rest - jb2, jb, L21, lda, A12 + jb2, lda, A22 + jb2, lda);
            queued = h->gemm_queue_used;
            cur_col0 = queued && h->gemm_col0_complete;
            cur_tiles = h->gemm_col0_tiles;
            h->gemm_queue = 0;
            h->gemm_counters = nullptr;
            h->gemm_col0 = nullptr;
            LSX_TRY(rq);
        }
        // panel k is done with its exchange area and panel k+3 reuses it: cleared behind the update, off the path
        // HEAD -> update start -> panel k+1; the chain of panel k+3 waits for (a part of) update k+1, behind this
        LSX_HIP(hipMemsetAsync((char *)h->scratch + (size_t)(step % 3) * area, 0, area_x, main_s));
        if (area_c && step + 3 < nsteps)   // ... and the multiplier buffer of the column-distributed kernel back to "unwritten"
            LSX_HIP(hipMemsetAsync((char *)h->scratch + (size_t)(step % 3) * area + panel_c_ones_offset(h, sizeof(T)), 0xff,
                                   panel_c_ones_bytes(n - k0 - (step + 3) * nb, sizeof(T)), main_s));
        LSX_HIP(hipEventRecord(h->ev_next, main_s));
        // panel k's interchanges on the columns LEFT of it: behind the update, where this stream only waits for the next
        // HEAD (one launch for all panels at the very end was 0.3 ms of an 8192^2 factorisation, on the critical path).
        // Its eighth of workgroups dealt to a busy panel XCD finishes when that panel does -- just before HEAD arrives.
        // Not in update-bound steps with a wide left-hand side, where this stream IS the critical path and the launch is not
        // small (16384^2: 8192+ columns from the first step of this driver, +0.4 ms): theirs are caught up in one launch at the
        // first step that has room.
        if (h->left_per_step && (rest <= 5632 || k <= 4096)) {
            if (left_done < step) {
                LSX_TRY(launch_laswp_left_all<T>(h, A, lda, k0 + left_done * nb, nb, step - left_done, list(left_done)));
                left_done = step;
            }
            if (k > 0) {
                h->moves = list(step);
                h->moves_valid = true;
                LSX_TRY(launch_laswp_moves<T>(h, k, A, lda, k));
            }
            left_done = step + 1;
        }
        {
            OnSide g(h, side);
            // The gate (panel k+1 not before update k has started) is still wanted while the panel takes nearly every
            // CU of its XCD: launched earlier, it starves the main stream's small kernels in front of the update of
            // their eighth of workgroups dealt to that XCD (8192^2, steps 2-7: 640 instead of 455 us per step).
            // (fp32 panels above 8192 rows keep 512 rows per workgroup: 25 of 32 CUs at 12800 rows)
            const bool tall = (rest > 6400 && (sizeof(T) == 8 || rest <= 8192)) || rest > 12800;
            if (queued && (!cur_col0 || tall)) LSX_TRY(launch_gate(h, pass + step, 2 * h->num_cu / 8));
            h->moves = list(step + 1);
            h->panel_area = (step + 1) % 3;
            LSX_TRY(launch_panel<T>(h, rest, jb2, A22, lda, k + jb, d_ipiv + k + jb, d_info));
            if (!h->moves_valid) { set_error("getrf_lookahead_x: panel without a gather list"); return LSX_ERR_INTERNAL; }
        }
        prev_col0 = cur_col0;
        prev_tiles = cur_tiles;
    }
    // the last panel is factored; every panel's interchanges on the columns left of it
    LSX_HIP(hipEventRecord(h->ev_panel, side));
    LSX_HIP(hipStreamWaitEvent(main_s, h->ev_panel, 0));
    join.armed = false;   // joined just above
    // what has not trailed its step: the last panel's (all of them with left_per_step = 0)
    return launch_laswp_left_all<T>(h, A, lda, k0 + left_done * nb, nb, nsteps - left_done, list(left_done));
}

// Workspaces of one factorisation of order n: scratch (panel partials, gather lists, three XCD-scope exchange areas
// and the driver's words) and ws2 (block inverses, two sets: the look-ahead driver alternates).  One computation for
// getrf_dev and the multi-device driver (mg.hip), so the two cannot drift apart.
int ensure_getrf_workspace(lsx_handle_t h, int n, size_t elem) {
    LSX_TRY(ensure_scratch(h, pad256(16 * ((size_t)n / 32 + 2)) + 2 * pad256(elem * 2 * (size_t)n) +
                                  ((size_t)n / 32 + 2) * 5248 + 8192 +
                                  3 * std::max(panel_x_area_bytes(h, n, elem), h->panel_col ? panel_c_area_bytes(h, n, elem) : (size_t)0) + 16384 +
                                  ((size_t)n / 16 + 2) * 40));
    const size_t tinv_elems = (size_t)((h->nb * h->kblock + 63) / 64) * 64 * 64;
    return grow(&h->ws2, &h->ws2_bytes, 2 * pad256(tinv_elems * elem));   // x2: the look-ahead driver alternates
}

// ---------------------------------------------------------------- blocked LU driver
template <typename T>
static int getrf_dev(lsx_handle_t h, int n, T *A, int lda, int32_t *d_ipiv, int *d_info) {
    LSX_ARG(n >= 0 && lda >= n && A && d_ipiv);
    if (n == 0) return LSX_OK;
    const int nb = h->nb;
    struct MfmaOnly { lsx_handle_t h; MfmaOnly(lsx_handle_t h_) : h(h_) { h->gemm_mfma_only = true; } ~MfmaOnly() { h->gemm_mfma_only = false; } } mfma_only(h);
    LSX_TRY(ensure_getrf_workspace(h, n, sizeof(T)));
    T *Tinv = (T *)h->ws2;
    if (!d_info) d_info = h->dev_status + 2;   // a time-out must be recorded somewhere: lsx_check_status reads it
    LSX_HIP(hipMemsetAsync(d_info, 0, sizeof(int), h->stream));
    // Look-ahead (panel k+1 on a side stream under the update of step k; bit-identical factors).  Measured
    // on MI355X with the pipelined panel, lookahead=1 against the sequential driver: n = 4096 -8 %,
    // 6144 +0.5 %, 7168 +5 %, 8192 +8 %, 10240..16384 +11..12 %, 20480 +10 %.  Below ~6500 the shorter update
    // no longer hides the panel and the split update costs more than it saves (fp32: see the variant choice below).
    int LOOKAHEAD_MIN = sizeof(T) == 8 ? 7168 : 10240;   // fp32: the update is half as long, break-even higher
    // XCD-scope panel and its own schedule.  Measured against the sequential driver: fp64 1536 0 %, 2048 +2 %, 2560 +4.5 %,
    // 3072 +7 %; fp32 (short updates) 3072 -1 %, 4096 0 %, 5120 +3 %, 6144 +7 %, 7168 +12 %.
    if (h->panel_mode == 4) LOOKAHEAD_MIN = sizeof(T) == 8 ? 2048 : 4096;
    if (h->lookahead_min > 0) LOOKAHEAD_MIN = h->lookahead_min;                  // option (tests, tuning)
    if (const char *e = getenv("LSX_LOOKAHEAD_MIN")) {   // diagnostics
        const int v = atoi(e);
        if (v > 0) LOOKAHEAD_MIN = v;
    }
    int k_end = n;  // the sequential driver below handles columns [0, k_end)
    if (h->lookahead && n >= LOOKAHEAD_MIN && h->kblock == 1) k_end = 0;
    const int W = nb * h->kblock;
    for (int k = 0; k < k_end; k += W) {
        const int w = (n - k < W) ? n - k : W;  // width of this super-block
        for (int j = 0; j < w; j += nb) {
            const int c = k + j;                      // first column of this panel
            const int jb = (w - j < nb) ? w - j : nb;
            T *Acc = A + (size_t)c * lda + c;
            LSX_TRY(launch_panel<T>(h, n - c, jb, Acc, lda, c, d_ipiv + c, d_info));
            if (h->moves_valid) {   // left and right of the panel in one launch
                LSX_TRY(launch_laswp_moves_around<T>(h, n, A, lda, c, c, jb));
            } else {
                LSX_TRY(apply_panel_swaps<T>(h, c, A, lda, c, jb, d_ipiv + c));                      // left
                LSX_TRY(apply_panel_swaps<T>(h, n - c - jb, A + c + jb, lda, c, jb, d_ipiv + c));  // right
            }
            const int inner = w - j - jb;  // columns of the super-block still to be factored
            if (inner > 0) {
                T *A12 = A + (size_t)c * lda + c + jb;
                LSX_TRY(launch_trtri<T>(h, 1, jb, Acc, lda, Tinv));
                LSX_TRY(launch_trsm_block<T>(h, 1, jb, inner, Acc, lda, Tinv, A12, lda));
                LSX_TRY(launch_gemm_sub<T>(h, n - c - jb, inner, jb, A + (size_t)(c + jb) * lda + c, lda, A12,
                                           lda, A + (size_t)(c + jb) * lda + c + jb, lda));
            }
        }
        const int rest = n - k - w;
        if (rest > 0) {
            T *Akk = A + (size_t)k * lda + k;
            T *A12 = A + (size_t)k * lda + k + w;
            LSX_TRY(launch_trtri<T>(h, 1, w, Akk, lda, Tinv));
            if (w > nb && nb % 64 == 0) {
                // U12 of a super-block one panel at a time: 128-row block solve, then the rows of the later
                // panels take the update of the earlier ones (a skinny MFMA update) before their own solve.
                // A single w-row block solve does the same flops inside one workgroup per 32 columns and
                // costs more than the deeper trailing update saves.
                for (int j = 0; j < w; j += nb) {
                    const int jb = (w - j < nb) ? w - j : nb;
                    if (j > 0)
                        LSX_TRY(launch_gemm_sub<T>(h, jb, rest, j, Akk + (size_t)j * lda, lda, A12, lda,
                                                   A12 + (size_t)j * lda, lda));
                    LSX_TRY(launch_trsm_block<T>(h, 1, jb, rest, Akk + (size_t)j * lda + j, lda,
                                                 Tinv + (size_t)(j / 64) * 4096, A12 + (size_t)j * lda, lda));
                }
            } else {
                LSX_TRY(launch_trsm_block<T>(h, 1, w, rest, Akk, lda, Tinv, A12, lda));
            }
            LSX_TRY(launch_gemm_sub<T>(h, rest, rest, w, A + (size_t)(k + w) * lda + k, lda, A12, lda,
                                       A + (size_t)(k + w) * lda + k + w, lda));
        }
    }
    if (k_end < n) {
        // XCD-scope panel (panel = 4): its own schedule, for the part of the matrix whose panels one XCD holds
        // (8192 rows fp64, 16384 fp32); the steps in front of that part run the shared-CU schedule with the
        // device-scope panel (12288^2 fp64: 46.4 -> see DESIGN 6).
        int xrows = 32 * 64 * (sizeof(T) == 8 ? 4 : 8);
        if (h->xrows_limit > 0 && h->xrows_limit < xrows) xrows = h->xrows_limit;   // tests: hand-over at small orders
        if (h->panel_mode == 4 && !h->panel_debug && nb % 32 == 0 && k_end % 32 == 0 &&
            panel_x_area_bytes(h, n - k_end, sizeof(T)) > 0) {
            int kx = k_end;
            if (n - kx > xrows) kx += (n - kx - xrows + nb - 1) / nb * nb;
            if (h->hybrid_off && kx > k_end) kx = n;
            if (kx < n) {
                if (kx > k_end) LSX_TRY(getrf_lookahead<T>(h, n, A, lda, d_ipiv, d_info, Tinv, k_end, kx));
                return getrf_lookahead_x<T>(h, n, A, lda, d_ipiv, d_info, Tinv, kx);
            }
        }
        return getrf_lookahead<T>(h, n, A, lda, d_ipiv, d_info, Tinv, k_end);
    }
    return LSX_OK;
}

// Block substitution sweeps shared by the multi-right-hand-side solve and the inverse: 128-row diagonal blocks (their
// inverses in TinvL / TinvU), the rows beyond a block updated on the MFMA tile.
// pairs: TWO blocks per trailing update -- the update of the rows beyond the pair has depth 256 (the tile then runs at 0.76 of
// the MFMA pipe instead of 0.68, profiles/r03_pmc_gemm_mfma.json, and there are half as many launches), the block row in
// between gets its own 128-deep update first.  Per element the same fused multiply-adds in the same order as one block at a
// time: the accumulators start from C and take k in sequence either way; going upwards, where the block that comes second
// in memory must be applied first, the tile rotates its k index (gemm_kshift).  Same bits (tools/getri_ab.py, tests).
// tri (forward sweep of the inverse): block row kb only has columns [0, kb + jb) to work on, the others are exact zeros.
template <typename T>
static int sweep_forward(lsx_handle_t h, int n, int ncols, bool tri, bool pairs, const T *LU, int lda, const T *TinvL, T *X, int ldx) {
    const int sb = 128;
    auto Lp = [&](int r, int c) { return LU + (size_t)r * lda + c; };
    auto Xp = [&](int r) { return X + (size_t)r * ldx; };
    for (int kb = 0; kb < n;) {
        const int jb = (n - kb < sb) ? n - kb : sb;
        const int nc = tri ? kb + jb : ncols;
        LSX_TRY(launch_trsm_block<T>(h, 1, jb, nc, Lp(kb, kb), lda, TinvL + (size_t)(kb / 64) * 4096, Xp(kb), ldx));
        const int below = n - kb - jb;
        if (below <= 0) break;
        if (!pairs || jb != sb) {
            LSX_TRY(launch_gemm_sub<T>(h, below, nc, jb, Lp(kb + jb, kb), lda, Xp(kb), ldx, Xp(kb + jb), ldx));
            kb += jb;
            continue;
        }
        const int jb2 = below < sb ? below : sb, nc2 = tri ? nc + jb2 : ncols;
        LSX_TRY(launch_gemm_sub<T>(h, jb2, nc, sb, Lp(kb + sb, kb), lda, Xp(kb), ldx, Xp(kb + sb), ldx));
        LSX_TRY(launch_trsm_block<T>(h, 1, jb2, nc2, Lp(kb + sb, kb + sb), lda, TinvL + (size_t)((kb + sb) / 64) * 4096, Xp(kb + sb), ldx));
        if (below > jb2)   // (tri: the first block's columns nc .. nc2 - 1 are exact zeros)
            LSX_TRY(launch_gemm_sub<T>(h, below - jb2, nc2, sb + jb2, Lp(kb + sb + jb2, kb), lda, Xp(kb), ldx, Xp(kb + sb + jb2), ldx));
        kb += sb + jb2;
    }
    return LSX_OK;
}

template <typename T>
static int sweep_backward(lsx_handle_t h, int n, int ncols, bool pairs, const T *LU, int lda, const T *TinvU, T *X, int ldx) {
    const int sb = 128;
    auto Up = [&](int r, int c) { return LU + (size_t)r * lda + c; };
    auto Xp = [&](int r) { return X + (size_t)r * ldx; };
    for (int kb = ((n - 1) / sb) * sb; kb >= 0;) {
        const int jb = (n - kb < sb) ? n - kb : sb;
        LSX_TRY(launch_trsm_block<T>(h, 0, jb, ncols, Up(kb, kb), lda, TinvU + (size_t)(kb / 64) * 4096, Xp(kb), ldx));
        if (kb == 0) break;
        const int lo = kb - sb;
        if (!pairs || jb % 16 != 0) {
            LSX_TRY(launch_gemm_sub<T>(h, kb, ncols, jb, Up(0, kb), lda, Xp(kb), ldx, X, ldx));
            kb -= sb;
            continue;
        }
        LSX_TRY(launch_gemm_sub<T>(h, sb, ncols, jb, Up(lo, kb), lda, Xp(kb), ldx, Xp(lo), ldx));   // the block row above only
        LSX_TRY(launch_trsm_block<T>(h, 0, sb, ncols, Up(lo, lo), lda, TinvU + (size_t)(lo / 64) * 4096, Xp(lo), ldx));
        if (lo > 0) {
            // rows above the pair: k runs over block kb first, then block lo -- the order of one block at a time
            h->gemm_kshift = sb;
            const int rc = launch_gemm_sub<T>(h, lo, ncols, sb + jb, Up(0, lo), lda, Xp(lo), ldx, X, ldx);
            h->gemm_kshift = 0;
            LSX_TRY(rc);
        }
        kb = lo - sb;
    }
    return LSX_OK;
}

// two blocks per update pay where the updates are large and regular (8192^2 inverse 17.8 -> 15.7 ms, 4096^2 3.18 -> 3.05;
// ragged or small orders lose 3-12 % to the extra launches and the edge kernels)
static bool sweep_pairs(lsx_handle_t h, int n, int ncols) {
    return h->getri_pairs && !h->gemm_queue && ncols >= 1024 && (n >= 7168 || (n % 128 == 0 && n >= 4096));
}

// X <- U^-1 L^-1 X for X already row-permuted (n x nrhs)
template <typename T>
static int lu_solve_permuted(lsx_handle_t h, int n, int nrhs, const T *LU, int lda, T *X, int ldx) {
    const size_t blk = (size_t)((n + 63) / 64) * 64 * 64;
    LSX_TRY(grow(&h->ws2, &h->ws2_bytes, 2 * pad256(blk * sizeof(T))));
    T *TinvL = (T *)h->ws2;
    T *TinvU = (T *)((char *)h->ws2 + pad256(blk * sizeof(T)));
    LSX_TRY(launch_trtri<T>(h, 1, n, LU, lda, TinvL));
    LSX_TRY(launch_trtri<T>(h, 0, n, LU, lda, TinvU));
    const bool pairs = sweep_pairs(h, n, nrhs);
    LSX_TRY(sweep_forward<T>(h, n, nrhs, false, pairs, LU, lda, TinvL, X, ldx));    // L y = P b   (linalg.py:587-596)
    LSX_TRY(sweep_backward<T>(h, n, nrhs, pairs, LU, lda, TinvU, X, ldx));          // U x = y     (linalg.py:611-621)
    return LSX_OK;
}

template <typename T>
static int getrs_dev(lsx_handle_t h, int n, int nrhs, const T *LU, int lda, const int32_t *d_ipiv,
                     T *B, int ldb) {
    LSX_ARG(n >= 0 && nrhs >= 0 && lda >= n && ldb >= nrhs && LU && d_ipiv && B);
    if (n == 0 || nrhs == 0) return LSX_OK;
    // few right-hand sides: the solve-latency path (kernels_trsv.hip) works on 1, 2, 4 or 8 columns
    const int nr = nrhs <= 1 ? 1 : nrhs <= 2 ? 2 : nrhs <= 4 ? 4 : 8;
    if (nrhs <= 8 && n > 128) {
        const size_t b64 = pad256((size_t)((n + 63) / 64) * 64 * 64 * sizeof(T));
        const size_t b128 = pad256((size_t)((n + 127) / 128) * 128 * 128 * sizeof(T));
        const size_t vec = pad256(sizeof(T) * (size_t)n * nr);
        LSX_TRY(grow(&h->ws3, &h->ws3_bytes, pad256(sizeof(int32_t) * n) + 3 * vec));
        LSX_TRY(grow(&h->ws2, &h->ws2_bytes, 2 * b64 + 2 * b128));
        LSX_TRY(ensure_scratch(h, 512 + 2 * (size_t)((n + 127) / 128) * 128 * nr * 16));  // the solve's exchange area
        int32_t *perm = (int32_t *)h->ws3;
        T *Bc = (T *)((char *)h->ws3 + pad256(sizeof(int32_t) * n));  // copy of B, zero-padded to nr columns
        T *Bp = (T *)((char *)Bc + vec);                               // P * B, work space of the solve
        T *X = (T *)((char *)Bp + vec);
        char *w2 = (char *)h->ws2;
        if (h->trsv_mode == 2) {   // 128-row steps, helper workgroups, interchanges + gather fused into the preparation launch
            const int r2 = lu_solve_few_rhs2<T>(h, n, nrhs, nr, LU, lda, d_ipiv, B, ldb, (T *)w2, (T *)(w2 + b64),
                                                (T *)(w2 + 2 * b64), (T *)(w2 + 2 * b64 + b128), Bp, X);
            if (r2 != 1) return r2;
        }
        LSX_TRY(launch_ipiv_to_perm(h, n, d_ipiv, perm));
        if (nr != nrhs) LSX_HIP(hipMemsetAsync(Bc, 0, sizeof(T) * (size_t)n * nr, h->stream));
        LSX_TRY(launch_copy2d<T>(h, n, nrhs, B, ldb, Bc, nr));
        LSX_TRY(launch_gather_rows<T>(h, n, nr, perm, Bc, nr, Bp, nr));
        char *w = (char *)h->ws2;
        LSX_TRY(lu_solve_few_rhs<T>(h, n, nr, LU, lda, Bp, nr, X, (T *)w, (T *)(w + b64), (T *)(w + 2 * b64),
                                    (T *)(w + 2 * b64 + b128)));
        return launch_copy2d<T>(h, n, nrhs, X, nr, B, ldb);
    }
    // perm + a copy of B for the row gather
    LSX_TRY(grow(&h->ws3, &h->ws3_bytes, pad256(sizeof(int32_t) * n) + pad256(sizeof(T) * (size_t)n * nrhs)));
    int32_t *perm = (int32_t *)h->ws3;
    T *Bc = (T *)((char *)h->ws3 + pad256(sizeof(int32_t) * n));
    LSX_TRY(ensure_scratch(h, 8 * (size_t)n + 256));   // index arrays of the permutation conversion
    LSX_TRY(launch_ipiv_to_perm(h, n, d_ipiv, perm));
    LSX_TRY(launch_copy2d<T>(h, n, nrhs, B, ldb, Bc, nrhs));
    LSX_TRY(launch_gather_rows<T>(h, n, nrhs, perm, Bc, nrhs, B, ldb));
    return lu_solve_permuted<T>(h, n, nrhs, LU, lda, B, ldb);
}

template <typename T>
static int getri_dev(lsx_handle_t h, int n, const T *LU, int lda, const int32_t *d_ipiv, T *Inv,
                     int ldi) {
    LSX_ARG(n >= 0 && lda >= n && ldi >= n && LU && d_ipiv && Inv);
    if (n == 0) return LSX_OK;
    LSX_TRY(grow(&h->ws3, &h->ws3_bytes, pad256(sizeof(int32_t) * n)));
    int32_t *perm = (int32_t *)h->ws3;
    LSX_TRY(ensure_scratch(h, 8 * (size_t)n + 256));
    LSX_TRY(launch_ipiv_to_perm(h, n, d_ipiv, perm));
    if (n >= 512 && !h->getri_plain) {
        // A^-1 = U^-1 L^-1 P with the permutation applied LAST: L^-1 of the identity is lower triangular, so the
        // forward substitution of block row kb only has columns [0, kb + jb) to work on -- n^3/3 flops instead
        // of n^3 (4/3 n^3 for the inverse, LAPACK's getri count, instead of 2 n^3).  The skipped operations are
        // multiplications by exact zeros, so every entry has the bits the plain form below gives it.
        const int ldw = (n + 15) & ~15;
        LSX_TRY(grow(&h->ws5, &h->ws5_bytes, sizeof(T) * (size_t)n * ldw));
        T *W = (T *)h->ws5;
        LSX_TRY(launch_set_identity_perm<T>(h, n, nullptr, W, ldw));
        const size_t blk = (size_t)((n + 63) / 64) * 64 * 64;
        LSX_TRY(grow(&h->ws2, &h->ws2_bytes, 2 * pad256(blk * sizeof(T))));
        T *TinvL = (T *)h->ws2;
        T *TinvU = (T *)((char *)h->ws2 + pad256(blk * sizeof(T)));
        LSX_TRY(launch_trtri<T>(h, 1, n, LU, lda, TinvL));
        LSX_TRY(launch_trtri<T>(h, 0, n, LU, lda, TinvU));
        const bool pairs = sweep_pairs(h, n, n);
        LSX_TRY(sweep_forward<T>(h, n, n, true, pairs, LU, lda, TinvL, W, ldw));    // L Y = I on the columns that can be non-zero
        LSX_TRY(sweep_backward<T>(h, n, n, pairs, LU, lda, TinvU, W, ldw));         // U Z = Y, all columns
        return launch_scatter_cols<T>(h, n, perm, W, ldw, Inv, ldi);   // Inv[:, perm[c]] = Z[:, c]
    }
    LSX_TRY(launch_set_identity_perm<T>(h, n, perm, Inv, ldi));  // P * I
    return lu_solve_permuted<T>(h, n, n, LU, lda, Inv, ldi);
}

// ---------------------------------------------------------------- host-buffer wrappers
template <typename T>
static int h2d(lsx_handle_t h, int m, int n, const T *src, int lds, T *dst, int ldd) {
    LSX_HIP(hipMemcpy2DAsync(dst, (size_t)ldd * sizeof(T), src, (size_t)lds * sizeof(T),
                             (size_t)n * sizeof(T), m, hipMemcpyHostToDevice, h->stream));
    return LSX_OK;
}
template <typename T>
static int d2h(lsx_handle_t h, int m, int n, const T *src, int lds, T *dst, int ldd) {
    LSX_HIP(hipMemcpy2DAsync(dst, (size_t)ldd * sizeof(T), src, (size_t)lds * sizeof(T),
                             (size_t)n * sizeof(T), m, hipMemcpyDeviceToHost, h->stream));
    return LSX_OK;
}

static int ld_for(int n) { return (n + 15) & ~15; }  // device leading dimension: 128-B rows (fp64)

// Host-buffer entry points: upload, factor, read the info word -- and when the cooperative panel's exchange timed out
// (its workgroups were not all resident at once: something else held the CUs for longer than the bounded spin), upload
// again and factor with panel mode 0 (two plain launches per column, no workgroup waits for another) instead of
// failing.  hinfo < 0 on return only if that also failed.  Synchronises the handle's stream.
template <typename T>
static int factor_from_host(lsx_handle_t h, int n, const T *A, int lda, T *dA, int ld, int32_t *dp, int *dinfo,
                            int *hinfo) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        LSX_TRY(h2d<T>(h, n, n, A, lda, dA, ld));
        int rc;
        if (attempt == 0) {
            rc = getrf_dev<T>(h, n, dA, ld, dp, dinfo);
        } else {
            const int mode = h->panel_mode, look = h->lookahead;
            h->panel_mode = 0;
            h->lookahead = 0;
            rc = getrf_dev<T>(h, n, dA, ld, dp, dinfo);
            h->panel_mode = mode;
            h->lookahead = look;
            h->panel_fallbacks += 1;
        }
        LSX_TRY(rc);
        LSX_HIP(hipMemcpyAsync(hinfo, dinfo, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        LSX_HIP(hipStreamSynchronize(h->stream));
        if (*hinfo >= 0) return LSX_OK;
        // the time-out also set the status word; this call reports (or repairs) it, so it must not be left for the
        // next, unrelated call on the handle to find
        LSX_HIP(hipMemsetAsync(h->dev_status, 0, 3 * sizeof(int), h->stream));
    }
    set_error("panel exchange timed out on the device, and so did the per-column fallback");
    return LSX_ERR_INTERNAL;
}

template <typename T>
static int getrf_host(lsx_handle_t h, int n, T *A, int lda, int32_t *ipiv, int *info) {
    LSX_ARG(h && n >= 0 && lda >= n && (n == 0 || (A && ipiv)));
    if (info) *info = 0;
    if (n == 0) return LSX_OK;
    const int ld = ld_for(n);
    LSX_TRY(ensure_ws(h, pad256(sizeof(T) * (size_t)n * ld) + pad256(sizeof(int32_t) * n) + 512));
    Carver c(h->ws);
    T *dA = c.take<T>((size_t)n * ld);
    int32_t *dp = c.take<int32_t>(n);
    int *dinfo = c.take<int>(1);
    int hinfo = 0;
    LSX_TRY(factor_from_host<T>(h, n, A, lda, dA, ld, dp, dinfo, &hinfo));
    LSX_TRY(d2h<T>(h, n, n, dA, ld, A, lda));
    LSX_HIP(hipMemcpyAsync(ipiv, dp, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipStreamSynchronize(h->stream));
    if (info) *info = hinfo;
    return LSX_OK;
}

template <typename T>
static int getrs_host(lsx_handle_t h, int n, int nrhs, const T *LU, int lda, const int32_t *ipiv,
                      T *B, int ldb) {
    LSX_ARG(h && n >= 0 && nrhs >= 0 && lda >= n && ldb >= nrhs);
    if (n == 0 || nrhs == 0) return LSX_OK;
    LSX_ARG(LU && ipiv && B);
    const int ld = ld_for(n), ldx = ld_for(nrhs);
    LSX_TRY(ensure_ws(h, pad256(sizeof(T) * (size_t)n * ld) + pad256(sizeof(T) * (size_t)n * ldx) +
                             pad256(sizeof(int32_t) * n) + 512));
    Carver c(h->ws);
    T *dA = c.take<T>((size_t)n * ld);
    T *dB = c.take<T>((size_t)n * ldx);
    int32_t *dp = c.take<int32_t>(n);
    LSX_TRY(h2d<T>(h, n, n, LU, lda, dA, ld));
    LSX_TRY(h2d<T>(h, n, nrhs, B, ldb, dB, ldx));
    LSX_HIP(hipMemcpyAsync(dp, ipiv, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
    LSX_TRY(getrs_dev<T>(h, n, nrhs, dA, ld, dp, dB, ldx));
    LSX_TRY(check_dev_status(h));
    LSX_TRY(d2h<T>(h, n, nrhs, dB, ldx, B, ldb));
    LSX_HIP(hipStreamSynchronize(h->stream));
    return LSX_OK;
}

template <typename T>
static int gesv_host(lsx_handle_t h, int n, int nrhs, const T *A, int lda, T *B, int ldb, int *info,
                     double *pivot_ratio) {
    LSX_ARG(h && n >= 0 && nrhs >= 0 && lda >= n && ldb >= nrhs);
    if (info) *info = 0;
    if (pivot_ratio) *pivot_ratio = 1.0;
    if (n == 0) return LSX_OK;
    LSX_ARG(A && (nrhs == 0 || B));
    const int ld = ld_for(n), ldx = ld_for(nrhs > 0 ? nrhs : 1);
    LSX_TRY(ensure_ws(h, pad256(sizeof(T) * (size_t)n * ld) + pad256(sizeof(T) * (size_t)n * ldx) +
                             pad256(sizeof(int32_t) * n) + 1024));
    Carver c(h->ws);
    T *dA = c.take<T>((size_t)n * ld);
    T *dB = c.take<T>((size_t)n * ldx);
    int32_t *dp = c.take<int32_t>(n);
    int *dinfo = c.take<int>(1);
    double *dprobe = c.take<double>(2);
    if (nrhs > 0) LSX_TRY(h2d<T>(h, n, nrhs, B, ldb, dB, ldx));
    LSX_TRY(h2d<T>(h, n, n, A, lda, dA, ld));
    LSX_TRY(launch_amax<T>(h, n, n, dA, ld, dprobe));
    int hinfo = 0;
    LSX_TRY(factor_from_host<T>(h, n, A, lda, dA, ld, dp, dinfo, &hinfo));
    LSX_TRY(launch_diag_minabs<T>(h, n, dA, ld, dprobe));
    double probe[2] = {0, 0};
    LSX_HIP(hipMemcpyAsync(probe, dprobe, sizeof(probe), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipStreamSynchronize(h->stream));
    if (info) *info = hinfo;
    if (pivot_ratio) *pivot_ratio = probe[0] > 0 ? probe[1] / probe[0] : 0.0;
    if (hinfo != 0 || nrhs == 0) return LSX_OK;  // singular: B is left untouched
    LSX_TRY(getrs_dev<T>(h, n, nrhs, dA, ld, dp, dB, ldx));
    LSX_TRY(check_dev_status(h));
    LSX_TRY(d2h<T>(h, n, nrhs, dB, ldx, B, ldb));
    LSX_HIP(hipStreamSynchronize(h->stream));
    return LSX_OK;
}

}  // namespace lsx

using namespace lsx;

// ================================================================== extern "C"
extern "C" {

const char *lsx_last_error(void) { return g_err; }

int lsx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lsx_create(lsx_handle_t *out, int device) {
    if (!out) { set_error("lsx_create: null out"); return LSX_ERR_ARG; }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible; liblsx has no CPU fallback");
        return LSX_ERR_NODEVICE;
    }
    LSX_ARG(device >= 0 && device < n);
    hipDeviceProp_t prop;
    LSX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; liblsx is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return LSX_ERR_NODEVICE;
    }
    struct RestoreDevice {   // the caller's (and torch's) current device is not ours to change
        int prev = -1;
        RestoreDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
        ~RestoreDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
    } restore_device;
    LSX_HIP(hipSetDevice(device));
    lsx_handle_t h = new lsx_handle_s();
    h->device = device;
    h->num_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete h;
        return LSX_ERR_HIP;
    }
    h->stream = h->own_stream;
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = most urgent
        if (hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, hi) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_panel, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_next, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming) != hipSuccess) {
            set_error("could not create the look-ahead stream / events");
            (void)hipStreamDestroy(h->own_stream);
            delete h;
            return LSX_ERR_HIP;
        }
    }
    int r = grow(&h->scratch, &h->scratch_bytes, 1 << 20);
    if (r == LSX_OK) {
        size_t ds = 0;
        void *p = nullptr;
        r = grow(&p, &ds, 256);
        h->dev_status = (int *)p;
        if (r == LSX_OK && hipMemset(p, 0, 256) != hipSuccess) r = LSX_ERR_HIP;
    }
    if (r == LSX_OK) {
        size_t mv = 0;
        r = grow(&h->moves_buf[0], &mv, 8192);
        h->moves_buf[1] = (char *)h->moves_buf[0] + 4096;
        h->moves = h->moves_buf[0];
    }
    if (r != LSX_OK) { (void)hipStreamDestroy(h->own_stream); delete h; return r; }
    *out = h;
    return LSX_OK;
}

int lsx_destroy(lsx_handle_t h) {
    if (!h) return LSX_OK;
    LSX_DEVICE_GUARD(h);
    (void)hipStreamSynchronize(h->stream);
    for (auto &ev : h->prof.pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto &e : h->prof.pool) (void)hipEventDestroy(e);
    if (h->ws) (void)hipFree(h->ws);
    if (h->ws2) (void)hipFree(h->ws2);
    if (h->ws3) (void)hipFree(h->ws3);
    if (h->ws4) (void)hipFree(h->ws4);
    if (h->ws6) (void)hipFree(h->ws6);
    if (h->xchg) (void)hipFree(h->xchg);
    if (h->ws5) (void)hipFree(h->ws5);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->moves_buf[0]) (void)hipFree(h->moves_buf[0]);
    if (h->moves_all) (void)hipFree(h->moves_all);
    if (h->dev_status) (void)hipFree(h->dev_status);
    if (h->ev_panel) (void)hipEventDestroy(h->ev_panel);
    if (h->ev_next) (void)hipEventDestroy(h->ev_next);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return LSX_OK;
}

int lsx_set_stream(lsx_handle_t h, void *hip_stream) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    LSX_HIP(hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)hip_stream;  // NULL = the default (null) stream
    return LSX_OK;
}

int lsx_use_own_stream(lsx_handle_t h) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    LSX_HIP(hipStreamSynchronize(h->stream));
    h->stream = h->own_stream;
    return LSX_OK;
}

int lsx_synchronize(lsx_handle_t h) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    LSX_HIP(hipStreamSynchronize(h->stream));
    return LSX_OK;
}

int lsx_check_status(lsx_handle_t h) {
    LSX_ARG(h);
    LSX_DEVICE_GUARD(h);
    return check_dev_status(h);
}

int lsx_set_option(lsx_handle_t h, const char *key, int value) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && key);
    if (!strcmp(key, "nb")) {
        LSX_ARG(value >= 16 && value <= 128 && value % 16 == 0);
        h->nb = value;
    } else if (!strcmp(key, "panel")) {
        LSX_ARG(value >= 0 && value <= 4);
#ifndef LSX_DIAG_PANELS
        if (value == 1 || value == 2) {
            set_error("panel modes 1 and 2 (superseded kernels) are only in the diagnostic build: make DIAG=1");
            return LSX_ERR_ARG;
        }
#endif
        h->panel_mode = value;
    } else if (!strcmp(key, "trsv")) {
        LSX_ARG(value >= 0 && value <= 2);
        h->trsv_mode = value;
    } else if (!strcmp(key, "gemm_tiles32")) {
        LSX_ARG(value == 0 || value == 1);
        h->gemm_no_tiles32 = !value;
    } else if (!strcmp(key, "gemm_queue_test")) {
        LSX_ARG(value >= 0 && value <= 3);
        h->gemm_queue_test = value;
    } else if (!strcmp(key, "gemm_stagger")) {
        LSX_ARG(value >= 0 && value <= 64);
        h->gemm_stagger = value;
    } else if (!strcmp(key, "gemm_waves")) {
        LSX_ARG(value == 0 || value == 4 || value == 8);
        h->gemm_waves = value;
    } else if (!strcmp(key, "kblock")) {
        LSX_ARG(value == 1 || value == 2);
        h->kblock = value;
    } else if (!strcmp(key, "panel_rt")) {
        LSX_ARG(value == 2 || value == 4 || value == 8);
        h->panel_rt = value;
    } else if (!strcmp(key, "panel_nt")) {
        LSX_ARG(value == 0 || value == 256 || value == 512 || value == 1024);
        h->panel_nt = value;
    } else if (!strcmp(key, "rref_blocked")) {
        LSX_ARG(value == 0 || value == 1);
        h->rref_blocked = value;
    } else if (!strcmp(key, "xrows_limit")) {   // tests: hand over to the XCD-scope driver below this many rows (0 = the kernel's limit)
        LSX_ARG(value >= 0);
        h->xrows_limit = value;
    } else if (!strcmp(key, "getri_structured")) {   // 0: the inverse as a plain solve of P*I (cross-check)
        LSX_ARG(value == 0 || value == 1);
        h->getri_plain = !value;
    } else if (!strcmp(key, "x_events")) {   // 1: the XCD-scope schedule orders its chain behind whole updates (events + gate)
        LSX_ARG(value == 0 || value == 1);
        h->x_events = value;
    } else if (!strcmp(key, "hybrid")) {   // 0: matrices taller than one XCD holds use the shared-CU schedule throughout
        LSX_ARG(value == 0 || value == 1);
        h->hybrid_off = !value;
    } else if (!strcmp(key, "prof_sample")) {   // bracket every value-th launch of a profiled bucket
        LSX_ARG(value >= 1);
        h->prof.sample = value;
    } else if (!strcmp(key, "panel_spin_limit")) {   // tests: a short limit turns a held-up panel into a time-out;
        LSX_ARG(value != 0);                           // negative: |value| and one participant arrives late
        h->panel_spin_limit = value;
    } else if (!strcmp(key, "trsv_spin_limit")) {   // tests: 0 makes the first unanswered poll a time-out
        LSX_ARG(value >= 0);
        h->spin_limit = value;
    } else if (!strcmp(key, "panel_xcd")) {
        LSX_ARG(value == 0 || value == 1);
        h->panel_xcd = value;
    } else if (!strcmp(key, "panel_col")) {   // XCD panel up to 4096 rows: 1 = columns over the workgroups (kernels_panel_c.hip; cross-check), 0 = rows
        LSX_ARG(value == 0 || value == 1);
        h->panel_col = value;
    } else if (!strcmp(key, "panel_col_wt")) {   // tests: the column-distributed panel as if its workgroups were on several XCDs
        LSX_ARG(value == 0 || value == 1);
        h->panel_col_wt = value;
    } else if (!strcmp(key, "getri_pairs")) {   // 0: the inverse with one 128-row block per trailing update (cross-check: same bits)
        LSX_ARG(value == 0 || value == 1);
        h->getri_pairs = value;
    } else if (!strcmp(key, "left_per_step")) {   // 0: every panel's interchanges on the columns left of it in one launch at the end (cross-check)
        LSX_ARG(value == 0 || value == 1);
        h->left_per_step = value;
    } else if (!strcmp(key, "chain_fused")) {   // 0: chain head and the next panel's block solve as separate launches (cross-check)
        LSX_ARG(value == 0 || value == 1);
        h->chain_fused = value;
    } else if (!strcmp(key, "rref_first_fast")) {   // 0: LSX_PIVOT_FIRST always through the per-column kernels (cross-check)
        LSX_ARG(value == 0 || value == 1);
        h->rref_first_fast = value;
    } else if (!strcmp(key, "chain_wait_limit")) {   // tests: 0 makes the chain's wait for the update's first tile column a time-out
        LSX_ARG(value >= 0);
        h->chain_wait_limit = value;
    } else if (!strcmp(key, "panel_debug")) {
        h->panel_debug = value != 0;
    } else if (!strcmp(key, "lookahead")) {
        LSX_ARG(value == 0 || value == 1);
        h->lookahead = value;
    } else if (!strcmp(key, "lookahead_min")) {
        LSX_ARG(value >= 0);
        h->lookahead_min = value;
    } else {
        set_error("unknown option '%s'", key);
        return LSX_ERR_ARG;
    }
    return LSX_OK;
}

int lsx_get_option(lsx_handle_t h, const char *key, int *value) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && key && value);
    if (!strcmp(key, "nb")) *value = h->nb;
    else if (!strcmp(key, "panel")) *value = h->panel_mode;
    else if (!strcmp(key, "lookahead")) *value = h->lookahead;
    else if (!strcmp(key, "lookahead_min")) *value = h->lookahead_min;
    else if (!strcmp(key, "panel_rt")) *value = h->panel_rt;
    else if (!strcmp(key, "kblock")) *value = h->kblock;
    else if (!strcmp(key, "gemm_waves")) *value = h->gemm_waves;
    else if (!strcmp(key, "gemm_stagger")) *value = h->gemm_stagger;
    else if (!strcmp(key, "trsv")) *value = h->trsv_mode;
    else if (!strcmp(key, "panel_nt")) *value = h->panel_nt;
    else if (!strcmp(key, "panel_xcd")) *value = h->panel_xcd;
    else if (!strcmp(key, "chain_wait_limit")) *value = h->chain_wait_limit;
    else if (!strcmp(key, "rref_first_fast")) *value = h->rref_first_fast;
    else if (!strcmp(key, "chain_fused")) *value = h->chain_fused;
    else if (!strcmp(key, "left_per_step")) *value = h->left_per_step;
    else if (!strcmp(key, "getri_pairs")) *value = h->getri_pairs;
    else if (!strcmp(key, "panel_col")) *value = h->panel_col;
    else if (!strcmp(key, "panel_col_launches")) *value = (int)(h->panel_col_launches & 0x7fffffff);
    else if (!strcmp(key, "panel_col_wt")) *value = h->panel_col_wt;
    else if (!strcmp(key, "rref_first_used")) *value = h->rref_first_used;
    else if (!strcmp(key, "panel_fallbacks")) *value = h->panel_fallbacks;
    else if (!strcmp(key, "diag_panels")) {
#ifdef LSX_DIAG_PANELS
        *value = 1;
#else
        *value = 0;
#endif
    }
    else if (!strcmp(key, "num_cu")) *value = h->num_cu;
    else { set_error("unknown option '%s'", key); return LSX_ERR_ARG; }
    return LSX_OK;
}

// ---- fp64 host
int lsx_getrf_f64(lsx_handle_t h, int n, double *A, int lda, int32_t *ipiv, int *info) {
    LSX_DEVICE_GUARD(h);
    return getrf_host<double>(h, n, A, lda, ipiv, info);
}
int lsx_getrs_f64(lsx_handle_t h, int n, int nrhs, const double *LU, int lda, const int32_t *ipiv,
                  double *B, int ldb) {
    LSX_DEVICE_GUARD(h);
    return getrs_host<double>(h, n, nrhs, LU, lda, ipiv, B, ldb);
}
int lsx_gesv_f64(lsx_handle_t h, int n, int nrhs, const double *A, int lda, double *B, int ldb,
                 int *info, double *pivot_ratio) {
    LSX_DEVICE_GUARD(h);
    return gesv_host<double>(h, n, nrhs, A, lda, B, ldb, info, pivot_ratio);
}
int lsx_getrf_f32(lsx_handle_t h, int n, float *A, int lda, int32_t *ipiv, int *info) {
    LSX_DEVICE_GUARD(h);
    return getrf_host<float>(h, n, A, lda, ipiv, info);
}
int lsx_getrs_f32(lsx_handle_t h, int n, int nrhs, const float *LU, int lda, const int32_t *ipiv,
                  float *B, int ldb) {
    LSX_DEVICE_GUARD(h);
    return getrs_host<float>(h, n, nrhs, LU, lda, ipiv, B, ldb);
}
int lsx_gesv_f32(lsx_handle_t h, int n, int nrhs, const float *A, int lda, float *B, int ldb,
                 int *info, double *pivot_ratio) {
    LSX_DEVICE_GUARD(h);
    return gesv_host<float>(h, n, nrhs, A, lda, B, ldb, info, pivot_ratio);
}

}  // extern "C"

namespace lsx {

template <typename T>
static int getri_host(lsx_handle_t h, int n, const T *A, int lda, T *Ainv, int ldi, int *info, double *pivot_ratio) {
    LSX_ARG(h && n >= 0 && lda >= n && ldi >= n);
    if (info) *info = 0;
    if (pivot_ratio) *pivot_ratio = 1.0;
    if (n == 0) return LSX_OK;
    LSX_ARG(A && Ainv);
    const int ld = ld_for(n);
    LSX_TRY(ensure_ws(h, 2 * pad256(sizeof(T) * (size_t)n * ld) + pad256(sizeof(int32_t) * n) + 1024));
    Carver c(h->ws);
    T *dA = c.take<T>((size_t)n * ld);
    T *dI = c.take<T>((size_t)n * ld);
    int32_t *dp = c.take<int32_t>(n);
    int *dinfo = c.take<int>(1);
    double *dprobe = c.take<double>(2);
    LSX_TRY(h2d<T>(h, n, n, A, lda, dA, ld));
    LSX_TRY(launch_amax<T>(h, n, n, dA, ld, dprobe));
    int hinfo = 0;
    LSX_TRY(factor_from_host<T>(h, n, A, lda, dA, ld, dp, dinfo, &hinfo));
    LSX_TRY(launch_diag_minabs<T>(h, n, dA, ld, dprobe));
    double probe[2] = {0, 0};
    LSX_HIP(hipMemcpyAsync(probe, dprobe, sizeof(probe), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipStreamSynchronize(h->stream));
    if (info) *info = hinfo;
    if (pivot_ratio) *pivot_ratio = probe[0] > 0 ? probe[1] / probe[0] : 0.0;
    if (hinfo != 0) return LSX_OK;  // exactly singular: caller reports NoSolution (linalg.py:737)
    LSX_TRY(getri_dev<T>(h, n, dA, ld, dp, dI, ld));
    LSX_TRY(d2h<T>(h, n, n, dI, ld, Ainv, ldi));
    LSX_HIP(hipStreamSynchronize(h->stream));
    return LSX_OK;
}

template <typename T>
static int det_host(lsx_handle_t h, int n, const T *A, int lda, double *sign, double *mant, int64_t *exp2) {
    LSX_ARG(h && n >= 0 && lda >= n && sign && mant && exp2);
    if (n == 0) { *sign = 1; *mant = 0.5; *exp2 = 1; return LSX_OK; }  // det([]) = 1 (linalg.py:197-199)
    LSX_ARG(A);
    const int ld = ld_for(n);
    LSX_TRY(ensure_ws(h, pad256(sizeof(T) * (size_t)n * ld) + pad256(sizeof(int32_t) * n) + 1024));
    Carver c(h->ws);
    T *dA = c.take<T>((size_t)n * ld);
    int32_t *dp = c.take<int32_t>(n);
    int *dinfo = c.take<int>(1);
    double *dout = c.take<double>(3);
    int hinfo = 0;   // a determinant of garbage factors must not be returned as a value: the status is read first
    LSX_TRY(factor_from_host<T>(h, n, A, lda, dA, ld, dp, dinfo, &hinfo));
    LSX_TRY(launch_det<T>(h, n, dA, ld, dp, dout));
    double out[3];
    LSX_HIP(hipMemcpyAsync(out, dout, sizeof(out), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipStreamSynchronize(h->stream));
    *sign = out[0]; *mant = out[1]; *exp2 = (int64_t)out[2];
    return LSX_OK;
}


// ---------------------------------------------------------------- first-rule row reduction, blocked
// The reference pivots on the FIRST row whose entry is non-zero (linalg.py:548-552).  With rank < m or bar < n that
// choice shows in the result: the carried-along columns of a pivot row are a combination of exactly the original rows
// that became pivot rows, and which rows those are depends on the rule.  The per-column kernels (kernels_rref.hip)
// reproduce the rule at one read + write of the live matrix per column -- ~8 TB for an 8192^2 matrix of rank 4096.
// This form gets the same result from three pieces of the fast machinery:
//   1. rank r and the pivot columns pc from the blocked max-rule reduction of a copy (the rule does not change them);
//   2. G = A[:, pc] (m x r, full column rank) factored P G = L U by the blocked LU with the panel kernel under the FIRST
//      rule: P and the r pivot rows in order are the reference's own interchanges (a swap there is a swap here);
//   3. with P A = [A1; A2]:  top rows  A1[:, pc]^-1 A1  (the unique rows spanned by the pivot rows whose pivot columns are
//      unit vectors),  other rows  A2 - A2[:, pc] * top  (MFMA update) -- then exact unit / zero entries as above.
//      The factors of step 2 only NAME the rows: without magnitude pivoting their multipliers grow (3e-7 relative error
//      in a 520 x 300 integer case of rank 130 when they were used for the values), so A1[:, pc] is factored once more
//      with partial pivoting and the values come from that solve.
// fp64, up to 8192 rows (one XCD's panel); returns 1 for anything else and when the factorisation of G meets a column
// without a candidate above the tolerance (the two rules then disagree on the rank: the per-column pass decides).
// Synchronises the stream twice (rank and max |a| are needed on the host for the launch shapes).
static int rref_first_fast(lsx_handle_t h, int m, int n, int bar, double *R, int ldr, int32_t *d_pivots, int *d_rank, double tol) {
    typedef double T;
    h->rref_first_used = 0;
    const bool dbg = getenv("LSX_RREF_DEBUG") != nullptr;   // development: say why the blocked form was not taken
    if (!h->rref_first_fast || !h->rref_blocked || h->panel_mode != 4 || m > 8192 || (size_t)m * bar < (size_t)256 * 256) {
        if (dbg) fprintf(stderr, "rref_first_fast: not applicable (fast %d blocked %d panel %d m %d bar %d)\n", h->rref_first_fast, h->rref_blocked, h->panel_mode, m, bar);
        return 1;
    }
    const int nb = h->nb;
    const int rmax = m < bar ? m : bar;
    const int ldw = ld_for(n), ldg = ld_for(rmax);
    const size_t wbytes = pad256(sizeof(T) * (size_t)m * ldw), gbytes = pad256(sizeof(T) * (size_t)m * ldg);
    LSX_TRY(grow(&h->ws6, &h->ws6_bytes, wbytes + gbytes + 2 * pad256(sizeof(int32_t) * (size_t)m) + 1024));
    Carver c(h->ws6);
    T *W = c.take<T>((size_t)m * ldw);
    T *Gm = c.take<T>((size_t)m * ldg);
    int32_t *ipiv = c.take<int32_t>(m);
    int32_t *ipiv2 = c.take<int32_t>(m);
    int *ginfo = c.take<int>(2);
    double *amax = c.take<double>(1);
    // 1. rank and pivot columns
    LSX_TRY(launch_copy2d<T>(h, m, n, R, ldr, W, ldw));
    const int rb = rref_blocked<T>(h, m, n, bar, W, ldw, d_pivots, d_rank, tol, LSX_PIVOT_MAX);
    if (dbg && rb != LSX_OK) fprintf(stderr, "rref_first_fast: rref_blocked returned %d\n", rb);
    if (rb != LSX_OK) return rb;   // 1: not for the blocked form either
    LSX_TRY(launch_amax<T>(h, m, bar, R, ldr, amax));
    int r = 0;
    double hmax = 0.0;
    LSX_HIP(hipMemcpyAsync(&r, d_rank, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipMemcpyAsync(&hmax, amax, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipStreamSynchronize(h->stream));
    if (r <= 0) { h->rref_first_used = 1; return LSX_OK; }      // nothing to pivot on: R is its own reduced form
    if (r == m) {   // every row is a pivot row: the result does not depend on the rule
        h->rref_first_used = 1;
        return launch_copy2d<T>(h, m, n, W, ldw, R, ldr);
    }
    // the same threshold the row reductions use (kernels_rref.hip: 32 eps max(m, n) max|a|), fixed at the input's scale
    const double ftol = tol >= 0 ? tol : 32.0 * 2.220446049250313e-16 * (double)(m > n ? m : n) * hmax;
    // 2. P G = L U under the first-non-zero rule
    LSX_TRY(launch_gather_pivot_cols<T>(h, m, r, 0, R, ldr, d_pivots, Gm, ldg));
    LSX_TRY(ensure_getrf_workspace(h, m, sizeof(T)));
    T *Tinv = (T *)h->ws2;
    LSX_HIP(hipMemsetAsync(ginfo, 0, sizeof(int), h->stream));
    struct MfmaOnly { lsx_handle_t h; MfmaOnly(lsx_handle_t h_) : h(h_) { h->gemm_mfma_only = true; } ~MfmaOnly() { h->gemm_mfma_only = false; } } mfma_only(h);
    for (int k = 0; k < r; k += nb) {
        const int jb = (r - k < nb) ? r - k : nb;
        T *Gkk = Gm + (size_t)k * ldg + k;
        h->moves_valid = false;
        const int pr = panel_xcd_first(h, m - k, jb, Gkk, ldg, k, k, ipiv + k, ginfo, ftol);
        if (dbg && pr != LSX_OK) fprintf(stderr, "rref_first_fast: panel at %d returned %d\n", k, pr);
        if (pr != LSX_OK) return pr;
        if (!h->moves_valid) { set_error("rref_first: panel without a gather list"); return LSX_ERR_INTERNAL; }
        LSX_TRY(launch_laswp_moves_around<T>(h, r, Gm, ldg, k, k, jb));         // the other columns of G
        const int rest = r - k - jb;
        if (rest > 0) {
            T *G12 = Gkk + jb;
            LSX_TRY(launch_trtri<T>(h, 1, jb, Gkk, ldg, Tinv));
            LSX_TRY(launch_trsm_block<T>(h, 1, jb, rest, Gkk, ldg, Tinv, G12, ldg));
            LSX_TRY(launch_gemm_sub<T>(h, m - k - jb, rest, jb, Gkk + (size_t)jb * ldg, ldg, G12, ldg, Gkk + (size_t)jb * ldg + jb, ldg));
        }
    }
    int hinfo = 0;
    LSX_HIP(hipMemcpyAsync(&hinfo, ginfo, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipStreamSynchronize(h->stream));
    if (dbg) fprintf(stderr, "rref_first_fast: rank %d, max|a| %g, tol %g, info of the first-rule LU %d\n", r, hmax, ftol, hinfo);
    if (hinfo != 0) {
        // a pivot column of the max rule without a first-rule candidate (or a time-out): R is still the input -- the
        // per-column pass decides
        if (hinfo < 0) (void)hipMemsetAsync(h->dev_status, 0, 3 * sizeof(int), h->stream);
        return 1;
    }
    // 3. P A, then the top rows A1[:, pc]^-1 A1 (a partial-pivot factorisation of the r x r block: the first-rule
    // factors above only named the rows) and the rest A2 - A2[:, pc] * top
    for (int k = 0; k < r; k += nb) LSX_TRY(launch_laswp<T>(h, n, R, ldr, k, (r - k < nb) ? r - k : nb, ipiv + k));
    LSX_TRY(launch_gather_pivot_cols<T>(h, r, r, 0, R, ldr, d_pivots, Gm, ldg));
    LSX_TRY(getrf_dev<T>(h, r, Gm, ldg, ipiv2, ginfo + 1));
    LSX_TRY(getrs_dev<T>(h, r, n, Gm, ldg, ipiv2, R, ldr));
    LSX_TRY(launch_gather_pivot_cols<T>(h, m, r, r, R, ldr, d_pivots, Gm, ldg));    // rows >= r of G <- (P A)[r:, pc]
    LSX_TRY(launch_gemm_sub<T>(h, m - r, n, r, Gm + (size_t)r * ldg, ldg, R, ldr, R + (size_t)r * ldr, ldr));
    LSX_TRY(launch_rref_finish<T>(h, m, bar, R, ldr, d_pivots, r));
    h->rref_first_used = 1;
    return LSX_OK;
}

// Row reduction under either rule: the blocked first-rule form where it applies, else kernels_rref.hip / _blk.hip.
template <typename T>
static int rref_any(lsx_handle_t h, int m, int n, int bar, T *R, int ldr, int32_t *d_pivots, int *d_rank, double tol, int pivot_rule) {
    if constexpr (sizeof(T) == 8) {
        if (pivot_rule == LSX_PIVOT_FIRST) {
            const int rc = rref_first_fast(h, m, n, bar, R, ldr, d_pivots, d_rank, tol);
            if (rc != 1) return rc;   // done, or a real error; 1: not applicable, R untouched
        }
    }
    return launch_rref<T>(h, m, n, bar, R, ldr, d_pivots, d_rank, tol, pivot_rule);
}

template <typename T>
static int rref_host(lsx_handle_t h, int m, int n, int bar_col, const T *A, int lda, T *R, int ldr, int32_t *pivots,
                     int *rank, double tol, int pivot_rule) {
    LSX_ARG(h && m >= 1 && n >= 1 && lda >= n && ldr >= n && A && R && pivots && rank);
    LSX_ARG(pivot_rule == LSX_PIVOT_FIRST || pivot_rule == LSX_PIVOT_MAX);
    const int bar = bar_col > 0 ? bar_col : n - 1;  // linalg.py:543
    LSX_ARG(bar <= n);
    const int ld = ld_for(n);
    const int np = m < n ? m : n;
    LSX_TRY(ensure_ws(h, pad256(sizeof(T) * (size_t)m * ld) + pad256(sizeof(int32_t) * 2 * np) + 512));
    LSX_TRY(ensure_scratch(h, 256 + 2 * sizeof(double) * (size_t)n + 4096 + 2 * (size_t)m));   // + the blocked form's row-group candidates
    Carver c(h->ws);
    T *dR = c.take<T>((size_t)m * ld);
    int32_t *dp = c.take<int32_t>(2 * (size_t)np);
    int *drank = c.take<int>(1);
    LSX_TRY(h2d<T>(h, m, n, A, lda, dR, ld));
    LSX_TRY(rref_any<T>(h, m, n, bar, dR, ld, dp, drank, tol, pivot_rule));
    LSX_TRY(d2h<T>(h, m, n, dR, ld, R, ldr));
    int hr = 0;
    LSX_HIP(hipMemcpyAsync(&hr, drank, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipStreamSynchronize(h->stream));
    *rank = hr;
    if (hr > 0) LSX_HIP(hipMemcpy(pivots, dp, sizeof(int32_t) * 2 * hr, hipMemcpyDeviceToHost));
    return LSX_OK;
}

// Mixed-precision solve (BASELINE config 5, "tolerance 1e-4" against the fp64 result): fp32 factors, residuals
// b - A x accumulated in fp64, `sweeps` corrections through the same factors.  dA (unfactored) and dB are read,
// dLU (n x n, ldl) receives the factors, dX64 (n x nrhs) the solution in fp64, dX32 (may be NULL) its fp32 rounding.
// d_stats (4 doubles, may be NULL): max|d| and max|x| of the last sweep, then of the first.  Asynchronous.
static int gesv_refined_dev(lsx_handle_t h, int n, int nrhs, const float *dA, int lda, float *dLU, int ldl, int32_t *d_ipiv,
                            int *d_info, const float *dB, int ldb, double *dX64, int ldx, float *dX32, int ldf, int sweeps,
                            double *d_stats) {
    LSX_ARG(n >= 1 && nrhs >= 1 && dA && dLU && d_ipiv && d_info && dB && dX64 && sweeps >= 0 && sweeps <= 16);
    LSX_TRY(launch_copy2d<float>(h, n, n, dA, lda, dLU, ldl));
    LSX_TRY(getrf_dev<float>(h, n, dLU, ldl, d_ipiv, d_info));
    // correction / right-hand-side buffer: n x nrhs fp32 (ws3 belongs to getrs)
    const int ldr = nrhs;
    LSX_TRY(grow(&h->ws4, &h->ws4_bytes, sizeof(float) * (size_t)n * ldr));
    float *dR = (float *)h->ws4;
    int rc = launch_copy2d<float>(h, n, nrhs, dB, ldb, dR, ldr);
    if (rc == LSX_OK) rc = getrs_dev<float>(h, n, nrhs, dLU, ldl, d_ipiv, dR, ldr);
    if (rc == LSX_OK) rc = launch_refine_apply(h, n, nrhs, 1, dR, ldr, dX64, ldx, dX32, ldf, d_stats ? d_stats + 2 : nullptr);
    for (int s = 0; s < sweeps && rc == LSX_OK; ++s) {
        rc = launch_resid_mixed(h, n, nrhs, dA, lda, dB, ldb, dX64, ldx, dR, ldr);
        if (rc == LSX_OK) rc = getrs_dev<float>(h, n, nrhs, dLU, ldl, d_ipiv, dR, ldr);
        if (rc == LSX_OK) rc = launch_refine_apply(h, n, nrhs, 0, dR, ldr, dX64, ldx, dX32, ldf, d_stats);
    }
    return rc;
}

}  // namespace lsx

extern "C" {

int lsx_getri_f64(lsx_handle_t h, int n, const double *A, int lda, double *Ainv, int ldi, int *info,
                  double *pivot_ratio) {
    LSX_DEVICE_GUARD(h);
    return getri_host<double>(h, n, A, lda, Ainv, ldi, info, pivot_ratio);
}
int lsx_getri_f32(lsx_handle_t h, int n, const float *A, int lda, float *Ainv, int ldi, int *info,
                  double *pivot_ratio) {
    LSX_DEVICE_GUARD(h);
    return getri_host<float>(h, n, A, lda, Ainv, ldi, info, pivot_ratio);
}
int lsx_det_f64(lsx_handle_t h, int n, const double *A, int lda, double *sign, double *mant, int64_t *exp2) {
    LSX_DEVICE_GUARD(h);
    return det_host<double>(h, n, A, lda, sign, mant, exp2);
}
int lsx_det_f32(lsx_handle_t h, int n, const float *A, int lda, double *sign, double *mant, int64_t *exp2) {
    LSX_DEVICE_GUARD(h);
    return det_host<float>(h, n, A, lda, sign, mant, exp2);
}

// fp32 factors + fp64 residuals: X (fp32) and / or X64 (fp64) <- A^-1 B to well below the fp32 forward error
int lsx_gesv_f32_refined(lsx_handle_t h, int n, int nrhs, const float *A, int lda, const float *B, int ldb, float *X,
                         int ldx, double *X64, int ldx64, int sweeps, int *info, double *pivot_ratio,
                         double *last_correction) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && n >= 0 && nrhs >= 0 && lda >= n && ldb >= nrhs && sweeps >= 0 && sweeps <= 16);
    if (info) *info = 0;
    if (pivot_ratio) *pivot_ratio = 1.0;
    if (last_correction) *last_correction = 0.0;
    if (n == 0 || nrhs == 0) return LSX_OK;
    LSX_ARG(A && B && (X || X64) && (!X || ldx >= nrhs) && (!X64 || ldx64 >= nrhs));
    const int ld = ld_for(n), lr = ld_for(nrhs);
    LSX_TRY(ensure_ws(h, 2 * pad256(sizeof(float) * (size_t)n * ld) + pad256(sizeof(float) * (size_t)n * lr) * 2 +
                             pad256(sizeof(double) * (size_t)n * lr) + pad256(sizeof(int32_t) * n) + 1024));
    Carver c(h->ws);
    float *dA = c.take<float>((size_t)n * ld);
    float *dLU = c.take<float>((size_t)n * ld);
    float *dB = c.take<float>((size_t)n * lr);
    float *dXf = c.take<float>((size_t)n * lr);
    double *dX = c.take<double>((size_t)n * lr);
    int32_t *dp = c.take<int32_t>(n);
    int *dinfo = c.take<int>(1);
    double *dstats = c.take<double>(6);
    LSX_TRY(h2d<float>(h, n, n, A, lda, dA, ld));
    LSX_TRY(h2d<float>(h, n, nrhs, B, ldb, dB, lr));
    LSX_TRY(launch_amax<float>(h, n, n, dA, ld, dstats + 4));
    LSX_TRY(gesv_refined_dev(h, n, nrhs, dA, ld, dLU, ld, dp, dinfo, dB, lr, dX, lr, dXf, lr, sweeps, dstats));
    LSX_TRY(launch_diag_minabs<float>(h, n, dLU, ld, dstats + 4));
    int hinfo = 0;
    double st[6] = {0, 0, 0, 0, 0, 0};
    LSX_HIP(hipMemcpyAsync(&hinfo, dinfo, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipMemcpyAsync(st, dstats, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipStreamSynchronize(h->stream));
    if (info) *info = hinfo;
    if (pivot_ratio) *pivot_ratio = st[4] > 0 ? st[5] / st[4] : 0.0;
    if (hinfo < 0) {
        set_error("panel exchange timed out on the device (workgroups not co-resident?)");
        return LSX_ERR_INTERNAL;
    }
    if (hinfo != 0) return LSX_OK;   // singular: X untouched
    LSX_TRY(check_dev_status(h));
    if (last_correction) *last_correction = st[1] > 0 ? st[0] / st[1] : 0.0;
    if (X) LSX_TRY(d2h<float>(h, n, nrhs, dXf, lr, X, ldx));
    if (X64) LSX_TRY(d2h<double>(h, n, nrhs, dX, lr, X64, ldx64));
    LSX_HIP(hipStreamSynchronize(h->stream));
    return LSX_OK;
}
int lsx_gesv_f32_refined_dev(lsx_handle_t h, int n, int nrhs, const float *dA, int lda, float *dLU, int ldl,
                             int32_t *d_ipiv, int *d_info, const float *dB, int ldb, double *dX64, int ldx, float *dX32,
                             int ldf, int sweeps, double *d_stats) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && lda >= n && ldl >= n && ldb >= nrhs && ldx >= nrhs && (!dX32 || ldf >= nrhs));
    return gesv_refined_dev(h, n, nrhs, dA, lda, dLU, ldl, d_ipiv, d_info, dB, ldb, dX64, ldx, dX32, ldf, sweeps, d_stats);
}

int lsx_matmul_f64(lsx_handle_t h, int m, int n, int k, const double *A, int lda, const double *B, int ldb,
                   double *C, int ldc) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && m >= 0 && n >= 0 && k >= 0 && lda >= k && ldb >= n && ldc >= n);
    if (m == 0 || n == 0) return LSX_OK;
    LSX_ARG(C && (k == 0 || (A && B)));
    const int la = ld_for(k > 0 ? k : 1), lb = ld_for(n), lc = ld_for(n);
    LSX_TRY(ensure_ws(h, pad256(sizeof(double) * (size_t)m * la) + pad256(sizeof(double) * (size_t)(k > 0 ? k : 1) * lb) +
                             pad256(sizeof(double) * (size_t)m * lc) + 512));
    Carver c(h->ws);
    double *dA = c.take<double>((size_t)m * la);
    double *dB = c.take<double>((size_t)(k > 0 ? k : 1) * lb);
    double *dC = c.take<double>((size_t)m * lc);
    LSX_HIP(hipMemsetAsync(dC, 0, sizeof(double) * (size_t)m * lc, h->stream));
    if (k > 0) {
        LSX_TRY(h2d<double>(h, m, k, A, lda, dA, la));
        LSX_TRY(h2d<double>(h, k, n, B, ldb, dB, lb));
        LSX_TRY(launch_gemm_acc<double>(h, 1, m, n, k, dA, la, dB, lb, dC, lc));
    }
    LSX_TRY(d2h<double>(h, m, n, dC, lc, C, ldc));
    LSX_HIP(hipStreamSynchronize(h->stream));
    return LSX_OK;
}

int lsx_rref_f64(lsx_handle_t h, int m, int n, int bar_col, const double *A, int lda, double *R,
                 int ldr, int32_t *pivots, int *rank, double tol, int pivot_rule) {
    LSX_DEVICE_GUARD(h);
    return rref_host<double>(h, m, n, bar_col, A, lda, R, ldr, pivots, rank, tol, pivot_rule);
}
int lsx_rref_f32(lsx_handle_t h, int m, int n, int bar_col, const float *A, int lda, float *R,
                 int ldr, int32_t *pivots, int *rank, double tol, int pivot_rule) {
    LSX_DEVICE_GUARD(h);
    return rref_host<float>(h, m, n, bar_col, A, lda, R, ldr, pivots, rank, tol, pivot_rule);
}

// ---- device-pointer entry points
int lsx_getrf_f64_dev(lsx_handle_t h, int n, double *dA, int lda, int32_t *d_ipiv, int *d_info) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    return getrf_dev<double>(h, n, dA, lda, d_ipiv, d_info);
}
int lsx_getrf_f32_dev(lsx_handle_t h, int n, float *dA, int lda, int32_t *d_ipiv, int *d_info) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    return getrf_dev<float>(h, n, dA, lda, d_ipiv, d_info);
}
int lsx_getrs_f64_dev(lsx_handle_t h, int n, int nrhs, const double *dLU, int lda,
                      const int32_t *d_ipiv, double *dB, int ldb) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    return getrs_dev<double>(h, n, nrhs, dLU, lda, d_ipiv, dB, ldb);
}
int lsx_getrs_f32_dev(lsx_handle_t h, int n, int nrhs, const float *dLU, int lda,
                      const int32_t *d_ipiv, float *dB, int ldb) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    return getrs_dev<float>(h, n, nrhs, dLU, lda, d_ipiv, dB, ldb);
}
int lsx_getri_f64_dev(lsx_handle_t h, int n, const double *dLU, int lda, const int32_t *d_ipiv,
                      double *dInv, int ldi) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    return getri_dev<double>(h, n, dLU, lda, d_ipiv, dInv, ldi);
}
int lsx_det_f64_dev(lsx_handle_t h, int n, const double *dLU, int lda, const int32_t *d_ipiv,
                    double *d_out) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && n >= 1 && dLU && d_ipiv && d_out);
    return launch_det<double>(h, n, dLU, lda, d_ipiv, d_out);
}
int lsx_getri_f32_dev(lsx_handle_t h, int n, const float *dLU, int lda, const int32_t *d_ipiv,
                      float *dInv, int ldi) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    return getri_dev<float>(h, n, dLU, lda, d_ipiv, dInv, ldi);
}
int lsx_det_f32_dev(lsx_handle_t h, int n, const float *dLU, int lda, const int32_t *d_ipiv,
                    double *d_out) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && n >= 1 && dLU && d_ipiv && d_out);
    return launch_det<float>(h, n, dLU, lda, d_ipiv, d_out);
}
int lsx_rref_trace_f64(lsx_handle_t h, int m, int n, int bar_col, const double *A, int lda, double *R, int ldr,
                       unsigned char *int_mask, int32_t *pivots, int *npivots, int32_t *steps, int max_steps,
                       int *nsteps, double *snaps, unsigned char *snap_int_mask, int max_snaps) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && m >= 1 && n >= 1 && lda >= n && ldr >= n && A && R && pivots && npivots && steps && nsteps);
    LSX_ARG(max_steps >= 1 && max_snaps >= 0 && (max_snaps == 0 || (snaps && snap_int_mask)));
    const int bar = bar_col > 0 ? bar_col : n - 1;  // linalg.py:543
    LSX_ARG(bar <= n);
    const int ld = ld_for(n);
    const int np = m < n ? m : n;
    const size_t snap_elems = (size_t)max_snaps * m * n;
    LSX_TRY(ensure_ws(h, pad256(sizeof(double) * (size_t)m * ld) + pad256(sizeof(int32_t) * 2 * np) +
                             pad256(sizeof(int32_t) * 4 * (size_t)max_steps) + pad256(sizeof(double) * snap_elems) +
                             pad256(snap_elems) + pad256((size_t)m * n) + 1024));
    LSX_TRY(ensure_scratch(h, 4096));
    Carver c(h->ws);
    double *dR = c.take<double>((size_t)m * ld);
    int32_t *dp = c.take<int32_t>(2 * (size_t)np);
    int32_t *ds = c.take<int32_t>(4 * (size_t)max_steps);
    double *dsn = snap_elems ? c.take<double>(snap_elems) : nullptr;
    unsigned char *dst = snap_elems ? c.take<unsigned char>(snap_elems) : nullptr;
    unsigned char *dT = c.take<unsigned char>((size_t)m * n);
    int *dout = c.take<int>(4);
    LSX_TRY(h2d<double>(h, m, n, A, lda, dR, ld));
    if (int_mask) LSX_HIP(hipMemcpyAsync(dT, int_mask, (size_t)m * n, hipMemcpyHostToDevice, h->stream));
    else LSX_HIP(hipMemsetAsync(dT, 0, (size_t)m * n, h->stream));
    LSX_TRY(launch_rref_trace(h, m, n, bar, dR, ld, dT, dp, ds, max_steps, dsn, dst, max_snaps, dout));
    LSX_TRY(d2h<double>(h, m, n, dR, ld, R, ldr));
    int out[3] = {0, 0, 0};
    LSX_HIP(hipMemcpyAsync(out, dout, sizeof(out), hipMemcpyDeviceToHost, h->stream));
    LSX_HIP(hipStreamSynchronize(h->stream));
    *npivots = out[0];
    *nsteps = out[1];
    if (out[2]) { set_error("rref_trace: more than %d steps", max_steps); return LSX_ERR_ARG; }
    if (out[0] > 0) LSX_HIP(hipMemcpy(pivots, dp, sizeof(int32_t) * 2 * out[0], hipMemcpyDeviceToHost));
    if (out[1] > 0) LSX_HIP(hipMemcpy(steps, ds, sizeof(int32_t) * 4 * out[1], hipMemcpyDeviceToHost));
    const int ns = out[1] < max_snaps ? out[1] : max_snaps;
    if (ns > 0) LSX_HIP(hipMemcpy(snaps, dsn, sizeof(double) * (size_t)ns * m * n, hipMemcpyDeviceToHost));
    if (ns > 0) LSX_HIP(hipMemcpy(snap_int_mask, dst, (size_t)ns * m * n, hipMemcpyDeviceToHost));
    if (int_mask) LSX_HIP(hipMemcpy(int_mask, dT, (size_t)m * n, hipMemcpyDeviceToHost));
    return LSX_OK;
}
int lsx_rref_f64_dev(lsx_handle_t h, int m, int n, int bar_col, double *dR, int ldr,
                     int32_t *d_pivots, int *d_rank, double tol, int pivot_rule) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && m >= 1 && n >= 1 && ldr >= n && dR && d_pivots);
    LSX_ARG(pivot_rule == LSX_PIVOT_FIRST || pivot_rule == LSX_PIVOT_MAX);
    const int bar = bar_col > 0 ? bar_col : n - 1;
    LSX_ARG(bar <= n);
    LSX_TRY(ensure_scratch(h, 256 + 2 * sizeof(double) * (size_t)n + 4096 + 2 * (size_t)m));   // + the blocked form's row-group candidates
    return rref_any<double>(h, m, n, bar, dR, ldr, d_pivots, d_rank, tol, pivot_rule);
}

int lsx_panel_f64_dev(lsx_handle_t h, int m, int jb, double *dP, int ldp, int row0, int32_t *d_ipiv,
                      int *d_info) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && m >= 1 && jb >= 1 && jb <= 256 && dP && d_ipiv && ldp >= jb);
    LSX_TRY(ensure_scratch(h, std::max({pad256(16 * ((size_t)m / 32 + 2)) + ((size_t)m / 32 + 2) * 5248 + 8192, panel_x_area_bytes(h, m, 8) + 4096,
                                        h->panel_col ? panel_c_area_bytes(h, m, 8) + 8192 : (size_t)0})));
    return launch_panel<double>(h, m, jb, dP, ldp, row0, d_ipiv, d_info);
}
int lsx_laswp_f64_dev(lsx_handle_t h, int ncols, double *dA, int lda, int row0, int jb,
                      const int32_t *d_ipiv) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dA && d_ipiv && jb >= 0 && jb <= 256);
    return launch_laswp<double>(h, ncols, dA, lda, row0, jb, d_ipiv);
}
int lsx_panel_moves_dev(lsx_handle_t h, int32_t *d_moves, int *valid) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && d_moves && valid);
    *valid = h->moves_valid ? 1 : 0;
    if (h->moves_valid)
        LSX_HIP(hipMemcpyAsync(d_moves, h->moves, 256 * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    return LSX_OK;
}
int lsx_laswp_moves_f64_dev(lsx_handle_t h, int ncols, double *dA, int lda, int row0, const int32_t *d_moves) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dA && d_moves);
    void *keep = h->moves;
    h->moves = (void *)d_moves;
    const int r = launch_laswp_moves<double>(h, ncols, dA, lda, row0);
    h->moves = keep;
    return r;
}
int lsx_trsm_lu_f64_dev(lsx_handle_t h, int jb, int ncols, const double *dL, int ldl, double *dB,
                        int ldb) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dL && dB && jb >= 1 && jb <= 256);
    const size_t tinv = (size_t)((jb + 63) / 64) * 4096 * sizeof(double);
    LSX_TRY(grow(&h->ws2, &h->ws2_bytes, pad256(tinv)));
    LSX_TRY(launch_trtri<double>(h, 1, jb, dL, ldl, (double *)h->ws2));
    return launch_trsm_block<double>(h, 1, jb, ncols, dL, ldl, (const double *)h->ws2, dB, ldb);
}
int lsx_gemm_sub_f64_dev(lsx_handle_t h, int m, int n, int k, const double *dA, int lda,
                         const double *dB, int ldb, double *dC, int ldc) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dA && dB && dC && lda >= k && ldb >= n && ldc >= n);
    if (h->gemm_queue_test) {   // measurements: the work-queue form alone (1: all XCDs, 2: XCD 0 left out as beside a panel)
        LSX_TRY(ensure_scratch(h, 4096));
        int *w = (int *)h->scratch;
        LSX_HIP(hipMemsetAsync(w, 0, 1024, h->stream));
        if (h->gemm_queue_test >= 2) LSX_HIP(hipMemsetD32Async((hipDeviceptr_t)w, 1, 1, h->stream));
        h->gemm_queue = 1; h->gemm_counters = w + 64; h->gemm_counter_sets = 1; h->gemm_counter_set = 0;
        h->gemm_avoid_word = h->gemm_queue_test >= 2 ? w : nullptr; h->gemm_pass_word = w + 1;
        h->gemm_col0 = h->gemm_queue_test == 3 ? w + 32 : nullptr;   // 3: with the column-0-first phase
        const int r = launch_gemm_sub<double>(h, m, n, k, dA, lda, dB, ldb, dC, ldc);
        h->gemm_queue = 0; h->gemm_counters = nullptr; h->gemm_avoid_word = nullptr; h->gemm_pass_word = nullptr; h->gemm_col0 = nullptr;
        return r;
    }
    return launch_gemm_sub<double>(h, m, n, k, dA, lda, dB, ldb, dC, ldc);
}
int lsx_gemm_add_f64_dev(lsx_handle_t h, int m, int n, int k, const double *dA, int lda,
                         const double *dB, int ldb, double *dC, int ldc) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dA && dB && dC && lda >= k && ldb >= n && ldc >= n);
    return launch_gemm_acc<double>(h, 1, m, n, k, dA, lda, dB, ldb, dC, ldc);
}
int lsx_gemm_sub_f32_dev(lsx_handle_t h, int m, int n, int k, const float *dA, int lda,
                         const float *dB, int ldb, float *dC, int ldc) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dA && dB && dC && lda >= k && ldb >= n && ldc >= n);
    return launch_gemm_sub<float>(h, m, n, k, dA, lda, dB, ldb, dC, ldc);
}

int lsx_fill_f64_dev(lsx_handle_t h, int kind, uint64_t seed, int m, int n, double *dA, int lda,
                     int row_off, int col_off) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dA && lda >= n && (kind == LSX_FILL_INT5 || kind == LSX_FILL_U11));
    return launch_fill<double>(h, kind, seed, m, n, dA, lda, row_off, col_off);
}
int lsx_fill_f32_dev(lsx_handle_t h, int kind, uint64_t seed, int m, int n, float *dA, int lda,
                     int row_off, int col_off) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dA && lda >= n && (kind == LSX_FILL_INT5 || kind == LSX_FILL_U11));
    return launch_fill<float>(h, kind, seed, m, n, dA, lda, row_off, col_off);
}

int lsx_diag_read_scratch(lsx_handle_t h, size_t offset, void *dst, size_t bytes) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dst && offset + bytes <= h->scratch_bytes);
    LSX_HIP(hipStreamSynchronize(h->stream));
    LSX_HIP(hipMemcpy(dst, (char *)h->scratch + offset, bytes, hipMemcpyDeviceToHost));
    return LSX_OK;
}

int lsx_diag_mfma_peak(lsx_handle_t h, int is_f32, int iters, int blocks_per_cu, double *tflops,
                       double *clock_mhz) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && tflops && iters > 0 && blocks_per_cu >= 1 && blocks_per_cu <= 8);
    return diag_mfma_peak(h, is_f32, iters, blocks_per_cu, tflops, clock_mhz);
}

int lsx_diag_xchg_probe(lsx_handle_t h, int mode, int G, int stride, int write_through, int epochs,
                        double *us_per_epoch, int *xcc_ids, int *nfail) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && us_per_epoch && nfail && G >= 2 && G <= 64 && (stride == 1 || stride == 8) && epochs >= 1);
    LSX_ARG(mode >= 0 && mode <= 2 && (G - 1) * stride + 1 <= 8 * h->num_cu);
    return diag_xchg_probe(h, mode, G, stride, write_through, epochs, us_per_epoch, xcc_ids, nfail);
}

int lsx_diag_occupy(lsx_handle_t h, int xcc, int wgs, int ms) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && xcc >= 0 && xcc < 8 && wgs >= 1 && wgs <= 32 && ms >= 1 && ms <= 10000);
    return diag_occupy(h, xcc, wgs, ms);
}

int lsx_diag_cu_mask_probe(lsx_handle_t h, const uint32_t *mask_words, int nwords, int nblocks, uint32_t *out) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && out && nblocks >= 1 && nwords >= 0 && (nwords == 0 || mask_words));
    return diag_cu_mask_probe(h, mask_words, nwords, nblocks, out);
}

// fused = 1: chain_head_kernel (inverses + gather list in one launch); 0: trtri64_kernel alone.  dTinv: ceil(jb/64) blocks
// of 64 x 64.  d_moves: 256 int2 (dst, src), -1 = void.  Asynchronous on the handle's stream.
int lsx_diag_chain_head_f32(lsx_handle_t h, int fused, int jb, const float *dT, int ldt, float *dTinv, int ncols,
                            float *dA, int lda, int row0, const int32_t *d_moves) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dT && dTinv && jb >= 1 && jb <= 128);
    if (!fused) return launch_trtri<float>(h, 1, jb, dT, ldt, dTinv);
    LSX_ARG(dA && d_moves);
    return diag_chain_head<float>(h, jb, dT, ldt, dTinv, ncols, dA, lda, row0, d_moves);
}
int lsx_diag_chain_head_f64(lsx_handle_t h, int fused, int jb, const double *dT, int ldt, double *dTinv, int ncols,
                            double *dA, int lda, int row0, const int32_t *d_moves) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && dT && dTinv && jb >= 1 && jb <= 128);
    if (!fused) return launch_trtri<double>(h, 1, jb, dT, ldt, dTinv);
    LSX_ARG(dA && d_moves);
    return diag_chain_head<double>(h, jb, dT, ldt, dTinv, ncols, dA, lda, row0, d_moves);
}

int lsx_getrf_mg_f64(lsx_handle_t *handles, int ndev, int n, double *const *dA, const int *lda,
                     int32_t *const *d_ipiv, int *const *d_info) {
    if (!handles || ndev < 1 || ndev > 64 || n < 1 || !dA || !lda || !d_ipiv || !d_info) {
        set_error("lsx_getrf_mg_f64: bad argument");
        return LSX_ERR_ARG;
    }
    for (int d = 0; d < ndev; ++d)
        if (!handles[d] || !d_ipiv[d] || !d_info[d]) { set_error("lsx_getrf_mg_f64: null handle or pointer for device %d", d); return LSX_ERR_ARG; }
    return getrf_mg_f64(handles, ndev, n, dA, lda, d_ipiv, d_info);
}

// ---- measurement
int lsx_prof_enable(lsx_handle_t h, int on) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    // 0 = off, 1 = every bucket, otherwise a bit mask (1 << LSX_PROF_*) shifted left by one:
    // e.g. 2 << LSX_PROF_GEMM brackets only the trailing-update launches
    h->prof.mask = on == 1 ? 0xffffffffu : (unsigned)on >> 1;
    return LSX_OK;
}

static int prof_drain(lsx_handle_t h) {
    LSX_HIP(hipStreamSynchronize(h->stream));
    Prof &p = h->prof;
    for (auto &ev : p.pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) p.ms[ev.bucket] += ms;
        p.pool.push_back(ev.a);
        p.pool.push_back(ev.b);
    }
    p.pending.clear();
    return LSX_OK;
}

int lsx_prof_reset(lsx_handle_t h) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h);
    LSX_TRY(prof_drain(h));
    for (int b = 0; b < LSX_PROF_NBUCKETS; ++b) {
        h->prof.ms[b] = 0; h->prof.launches[b] = 0; h->prof.flops[b] = 0; h->prof.bytes[b] = 0; h->prof.seen[b] = 0;
    }
    return LSX_OK;
}

int lsx_prof_read(lsx_handle_t h, int bucket, double *ms, long long *launches, double *flops,
                  double *bytes) {
    LSX_DEVICE_GUARD(h);
    LSX_ARG(h && bucket >= 0 && bucket < LSX_PROF_NBUCKETS);
    LSX_TRY(prof_drain(h));
    if (ms) *ms = h->prof.ms[bucket];
    if (launches) *launches = h->prof.launches[bucket];
    if (flops) *flops = h->prof.flops[bucket];
    if (bytes) *bytes = h->prof.bytes[bucket];
    return LSX_OK;
}

}  // extern "C"
