// Internal declarations shared by the HIP sources of liblsx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/lsx.h"

// Development build only (-DLSX_TSTAMP, tools/ts_lu.sh): device-clock stamps of kernel starts (first workgroup) and
// ends (every workgroup) without a profiler attached -- rocprofv3 stretches exactly the launch gaps one wants to see.
#ifdef LSX_TSTAMP
static __device__ long long *lsx_ts_buf;
struct TsScope {
    int id;
    __device__ void put(int tag) {
        long long *b = lsx_ts_buf;
        const int i = atomicAdd((int *)b, 1);
        if (i < (1 << 20)) { b[1 + 2 * i] = tag; b[2 + 2 * i] = wall_clock64(); }
    }
    __device__ TsScope(int id_) : id(id_) { if (lsx_ts_buf && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) put(id); }
    __device__ ~TsScope() { if (lsx_ts_buf && threadIdx.x == 0) put(id | 0x100); }
};
#define LSX_TS(id) TsScope ts_scope_(id)
#define LSX_TS_SETTER(name) extern "C" void lsx_ts_set_##name(long long *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(lsx_ts_buf), &p, sizeof p); }
#else
#define LSX_TS(id)
#define LSX_TS_SETTER(name)
#endif

namespace lsx {

void set_error(const char *fmt, ...);

#define LSX_HIP(call)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            lsx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                     \
            return LSX_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

#define LSX_TRY(call)              \
    do {                           \
        int r_ = (call);           \
        if (r_ != LSX_OK) return r_; \
    } while (0)

#define LSX_ARG(cond)                                                     \
    do {                                                                  \
        if (!(cond)) {                                                    \
            lsx::set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
            return LSX_ERR_ARG;                                           \
        }                                                                 \
    } while (0)

struct ProfEvent {
    hipEvent_t a, b;
    int bucket;
};

struct Prof {
    unsigned mask = 0;  // bit b set: launches of bucket b are bracketed by events
    int sample = 1;     // bracket every sample-th launch of a bucket only (events on the look-ahead chain cost time)
    long long seen[LSX_PROF_NBUCKETS] = {0};
    std::vector<ProfEvent> pending;
    std::vector<hipEvent_t> pool;
    double ms[LSX_PROF_NBUCKETS] = {0};
    long long launches[LSX_PROF_NBUCKETS] = {0};
    double flops[LSX_PROF_NBUCKETS] = {0};
    double bytes[LSX_PROF_NBUCKETS] = {0};
};

}  // namespace lsx

struct lsx_handle_s {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // stream in use (own or borrowed)
    hipStream_t side_stream = nullptr;  // high-priority stream for the look-ahead panel
    hipEvent_t ev_panel = nullptr, ev_next = nullptr, ev_start = nullptr, ev_done = nullptr;
    // look-ahead with a CU partition (lookahead = 2): the trailing update runs on a stream masked to
    // one set of CUs, the panel on a stream masked to the others, so both are resident at once
    // tunables
    int nb = 128;        // panel width (<= 128)
    int kblock = 1;      // panels per trailing update: update depth K = kblock * nb
    int panel_mode = 4;  // 0 = per-column launches, 1 = cooperative, 2 = blocked (experimental), 3 = pipelined (device-scope exchange), 4 = XCD-scope exchange on one XCD (taller panels than an XCD holds: 3)
    int lookahead = 1;   // 0: off; 1: panel k+1 on a high-priority side stream under the update of step k (fp64 n >= 7168, fp32 n >= 10240; bit-identical); 2: the variant with the update and the panel on disjoint CU sets; 3 = 1
    int lookahead_min = 0; // smallest n the look-ahead driver takes (0 = measured default: 7168 fp64, 10240 fp32)
    int panel_rt = 4;     // rows per thread in the cooperative panel
    int panel_nt = 0;     // threads per workgroup in the cooperative panel (0 = choose by panel height)
    int trsv_mode = 2;    // few-RHS solve: 2 = 128-row steps + helper workgroups (default), 1 = one cooperative launch per direction with 64-row steps, 0 = one launch per 128-row step
    int gemm_stagger = 0; // trailing update: start delay of every second resident workgroup, units of 8128 clocks
    int gemm_waves = 0;   // waves per workgroup in the trailing-update kernel (0 = auto; 4: 64x64 per wave, 8: 64x32)
    // trailing update through a work queue (look-ahead driver with the XCD-scope panel): see kernels_gemm.hip
    int gemm_queue = 0;           // 1: interior tiles are handed out by per-XCD counters
    const int *gemm_avoid_word = nullptr;  // device word: 1 + XCC id whose workgroups take no tiles (the panel's XCD)
    int *panel_xcc_word = nullptr;         // where the XCD-scope panel kernel records 1 + its XCC id
    int *gemm_counters = nullptr; // gemm_counter_sets x 8 ints in scratch, zeroed by the driver
    int gemm_counter_sets = 0, gemm_counter_set = 0;
    int *gemm_pass_word = nullptr;   // incremented by every workgroup that leaves because it sits on the avoided XCD
    long long panel_col_launches = 0;   // how many panels the column-distributed kernel took (tests: it really ran)
    int panel_col_wt = 0;            // tests: 1 = run it as if its workgroups were on several XCDs (write-through stores)
    int panel_col = 0;               // XCD panel up to 4096 rows: 1 = columns distributed over the workgroups (kernels_panel_c.hip: an independent
                                     // second implementation, slower -- DESIGN 5 -- kept as a cross-check), 0 = rows (kernels_panel_x.hip)
    int gemm_kshift = 0;             // next gemm launches: the k index starts at this offset and wraps (getri_dev)
    int getri_pairs = 1;             // inverse: two 128-row blocks per trailing update (K = 256), same bits
    int left_per_step = 1;           // XCD look-ahead driver: a panel's interchanges left of it trail its step (0: all at the end)
    int chain_fused = 1;             // 1: chain head and the next panel's block solve in one launch (option chain_fused)
    int *chain_info = nullptr;       // look-ahead driver: the factorisation's info word, for the chain's in-kernel waits (time-out -> negative)
    int chain_wait_limit = 1 << 21;  // polls of those waits before they give up (option chain_wait_limit: tests inject a time-out with 0)
    int *gemm_col0_static = nullptr; // shared-CU look-ahead driver: finished-tile count of the static grid's first tile column (zeroed by the driver)
    int *gemm_col0 = nullptr;        // look-ahead driver: {ticket, done} words of this update's tile column 0 (zeroed by the driver)
    int gemm_col0_tiles = 0;         // set with it: tiles in that column
    int x_events = 0;                // option x_events (measurements, tests): no column-0 ordering in the XCD-scope schedule
    bool gemm_col0_complete = false; // set by the last launch_gemm_*: the done word reaching tiles_m means column 0 is final
    int gemm_queue_used = 0;         // set by the last launch_gemm_*: 1 = its interior went through the queue
    void *moves_all = nullptr;       // look-ahead driver with the XCD-scope panel: one gather list per panel
    size_t moves_all_bytes = 0;
    int panel_xcd = 0;    // 1: pipelined panel with the exchange at XCD scope (<= 32 workgroups on one XCD)
    int panel_debug = 0;  // 1: stamped diagnostic panel kernel (tools/kbench.py)
    // set by the LU drivers: updates narrower than 16 columns also take the MFMA kernel, so that a column sees the
    // same summation order whichever driver (sequential / look-ahead) splits the trailing matrix around it
    bool gemm_mfma_only = false;
    // look-ahead driver: > 0 = the pipelined panel alternates between two exchange areas this far apart in
    // `scratch` and the DRIVER clears them (off the panel-to-panel chain); 0 = the launch clears its own
    size_t panel_area_stride = 0;
    int panel_area = 0;
    int num_cu = 256;
    // persistent device workspace (grown on demand, never shrunk)
    void *ws = nullptr;      // staging of caller matrices (host-buffer entry points)
    size_t ws_bytes = 0;
    void *ws2 = nullptr;     // triangular block inverses
    size_t ws2_bytes = 0;
    void *ws3 = nullptr;     // permutation vector + right-hand-side copy
    size_t ws3_bytes = 0;
    void *ws5 = nullptr;     // work matrix of the structured inverse (getri)
    size_t ws5_bytes = 0;
    int getri_plain = 0;     // option getri_structured=0: the inverse as a plain n-right-hand-side solve of P*I
    void *xchg = nullptr;    // few-RHS solve (kernels_trsv.hip, third form): exchange granules validated by an epoch, never cleared
    size_t xchg_bytes = 0;
    unsigned xchg_epoch = 0;
    void *ws6 = nullptr;     // first-rule row reduction: a copy of the matrix, its pivot columns, interchanges
    size_t ws6_bytes = 0;
    int rref_first_fast = 1; // 1: large fp64 inputs under LSX_PIVOT_FIRST take the blocked form (option rref_first_fast)
    int rref_first_used = 0; // read-only option: 1 if the last LSX_PIVOT_FIRST reduction took the blocked form
    void *ws4 = nullptr;     // residual / correction of the mixed-precision solve
    size_t ws4_bytes = 0;
    // small fixed device scratch: pivot search partials, flags, info words
    void *moves = nullptr;      // int2[256]: gather list emitted by the cooperative panel kernel (current buffer)
    void *moves_buf[2] = {nullptr, nullptr};  // the look-ahead driver alternates between two lists
    bool moves_valid = false;   // set by the last panel launch when `moves` describes its interchanges
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // device status words (never cleared by kernels): [0] panel exchange time-outs when the caller gave no info
    // word, [1] few-RHS solve time-outs, [2] the internal info word of a factorisation called without one
    int *dev_status = nullptr;
    int rref_blocked = 1;       // 1: large inputs with LSX_PIVOT_MAX take the blocked row reduction (option rref_blocked)
    int gemm_no_tiles32 = 0;    // option gemm_tiles32=0 (measurements)
    int gemm_queue_test = 0;    // option gemm_queue_test (measurements)
    int xrows_limit = 0;        // option xrows_limit (tests)
    int hybrid_off = 0;         // option hybrid=0 (measurements): no hand-over to the XCD-scope driver above its row limit
    int panel_fallbacks = 0;    // host-buffer factorisations redone with panel mode 0 after an exchange time-out
    int panel_spin_limit = 1 << 20;   // polls before the XCD-scope panel gives up on a neighbour (option: tests)
    int spin_limit = 1 << 20;   // polls before a cooperative solve gives up (option trsv_spin_limit: tests)
    lsx::Prof prof;
};

namespace lsx {

int ensure_ws(lsx_handle_t h, size_t bytes);
int ensure_getrf_workspace(lsx_handle_t h, int n, size_t elem);
size_t panel_c_area_bytes(lsx_handle_t h, int m, size_t elem);   // exchange area of the column-distributed XCD panel (contains the row-distributed one's)
size_t panel_c_ones_offset(lsx_handle_t h, size_t elem);          // where its 0xff-filled part starts ...
size_t panel_c_ones_bytes(int m, size_t elem);                    // ... and how long it is for a panel of m rows
// XCD-scope panel under the reference's first-non-zero pivot rule (kernels_panel_x.hip); 1 = shape not served
int panel_xcd_first(lsx_handle_t h, int m, int jb, double *P, int ldp, int row0, int col0, int32_t *d_ipiv, int *d_info, double tol);
template <typename T>
int launch_gather_pivot_cols(lsx_handle_t h, int m, int r, int row_lo, const T *src, int lds, const int32_t *d_pivots, T *dst, int ldd);
template <typename T>
int launch_rref_finish(lsx_handle_t h, int m, int bar, T *W, int ldw, const int32_t *d_pivots, int rank);
template <typename T>
int rref_blocked(lsx_handle_t h, int m, int n, int bar, T *W, int ldw, int32_t *d_pivots, int *d_rank, double tol, int pivot_rule);   // scratch + block-inverse workspace of one LU (api.hip)

// RAII-less profiling bracket: call begin() before a launch group, end() after.
struct ProfScope {
    lsx_handle_t h;
    hipStream_t st;
    int idx = -1;
    ProfScope(lsx_handle_t h_, int bucket, double flops = 0, double bytes = 0);
    ~ProfScope();
};

template <typename T>
struct Real;
template <>
struct Real<double> {
    static constexpr double eps = 2.220446049250313e-16;
};
template <>
struct Real<float> {
    static constexpr float eps = 1.1920929e-07f;
};

// 1/x for the pivot scaling of every panel kernel (the same function everywhere so the three
// panel modes stay bit-identical): hardware reciprocal + Newton, <= 1 ulp; an IEEE division
// costs ~40 instructions on the per-column critical path.
template <typename T>
__device__ __forceinline__ T fast_recip(T x) {
    if (sizeof(T) == 8) {
        double r = __builtin_amdgcn_rcp((double)x);
        r = r * (2.0 - (double)x * r);
        r = r * (2.0 - (double)x * r);
        return (T)r;
    }
    float r = __builtin_amdgcn_rcpf((float)x);
    r = r * (2.0f - (float)x * r);
    return (T)r;
}


// ---- kernel launchers (one per .hip file section); all asynchronous on h->stream ----
template <typename T>
int launch_fill(lsx_handle_t h, int kind, uint64_t seed, int m, int n, T *A, int lda, int row_off,
                int col_off);
template <typename T>
int launch_gemm_sub(lsx_handle_t h, int m, int n, int k, const T *A, int lda, const T *B, int ldb,
                    T *C, int ldc);
// C += A*B (plus = 1) or C -= A*B (plus = 0), same MFMA kernel
template <typename T>
int launch_gemm_acc(lsx_handle_t h, int plus, int m, int n, int k, const T *A, int lda, const T *B, int ldb,
                    T *C, int ldc);
// traced reference-order row reduction (kernels_trace.hip); d_out = {pivots, steps, overflow}
int launch_rref_trace(lsx_handle_t h, int m, int n, int bar, double *R, int ldr, unsigned char *Tm,
                      int32_t *d_pivots, int32_t *d_steps, int max_steps, double *d_snaps,
                      unsigned char *d_snap_t, int max_snaps, int *d_out);
template <typename T>
int launch_panel(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int32_t *d_ipiv,
                 int *d_info);
// bytes of one exchange area of the pipelined panel for panels of up to m rows; 0 = this handle's panel
// settings are not ones the pipelined kernel serves for every height <= m
size_t panel_pipe_area_bytes(lsx_handle_t h, int m);
size_t panel_x_area_bytes(lsx_handle_t h, int m, size_t elem);
template <typename T>
int launch_laswp(lsx_handle_t h, int ncols, T *A, int lda, int row0, int jb, const int32_t *d_ipiv);
// same interchanges from the gather list h->moves (written by the cooperative panel kernel)
template <typename T>
int launch_laswp_moves(lsx_handle_t h, int ncols, T *A, int lda, int row0);
template <typename T>
int launch_laswp_left_all(lsx_handle_t h, T *A, int lda, int k0, int nb, int nsteps, const void *lists);
int launch_gate(lsx_handle_t h, const int *word, int target);
int launch_wait_count(lsx_handle_t h, const int *word, int target);
int launch_resid_mixed(lsx_handle_t h, int n, int nrhs, const float *A, int lda, const float *B, int ldb, const double *X,
                       int ldx, float *R, int ldr);
int launch_refine_apply(lsx_handle_t h, int n, int nrhs, int init, const float *D, int ldd, double *X, int ldx, float *Xf,
                        int ldf, double *d_out2);
template <typename T>
int diag_chain_head(lsx_handle_t h, int jb, const T *Tm, int ldt, T *Tinv, int ncols, T *A, int lda, int row0,
                    const void *moves);
template <typename T>
int launch_laswp_moves_around(lsx_handle_t h, int n, T *A, int lda, int row0, int hole_at, int hole_w);
// Tinv (ceil(jb/64) blocks of 64x64) <- inverses of the 64x64 diagonal blocks of the
// unit-lower (lower=1) or non-unit upper (lower=0) triangle stored at T.
template <typename T>
int launch_trtri(lsx_handle_t h, int lower, int jb, const T *Tm, int ldt, T *Tinv);
// trtri(unit lower) + gather-list interchanges on a column block in one launch; 1 = not applicable
template <typename T>
int launch_chain_head(lsx_handle_t h, int jb, const T *Tm, int ldt, T *Tinv, int ncols, T *A, int lda, int row0,
                      const int *wait_word = nullptr, int wait_target = 0);
// chain head + block solve of the next panel's 128 columns in one launch (kernels_misc.hip); 1 = shapes not served
template <typename T>
int launch_chain_fused(lsx_handle_t h, int jb, const T *Tm, int ldt, T *Tinv, int ncols, T *A, int lda, int row0,
                       const int *wait_word, int wait_target, int *ready);
template <typename T>
int launch_trtri_both(lsx_handle_t h, int n, const T *LU, int lda, T *invL, T *invU);
// B (jb x ncols) <- inv(Tm) * B in place; Tinv = inverses of Tm's 64x64 diagonal blocks.
template <typename T>
int launch_trsm_block(lsx_handle_t h, int lower, int jb, int ncols, const T *Tm, int ldt,
                      const T *Tinv, T *B, int ldb);
int launch_ipiv_to_perm(lsx_handle_t h, int n, const int32_t *d_ipiv, int32_t *d_perm);
template <typename T>
int launch_gather_rows(lsx_handle_t h, int n, int ncols, const int32_t *d_perm, const T *S, int lds,
                       T *D, int ldd);
template <typename T>
int launch_set_identity_perm(lsx_handle_t h, int n, const int32_t *d_perm, T *X, int ldx);
template <typename T>
int launch_scatter_cols(lsx_handle_t h, int n, const int32_t *d_perm, const T *S, int lds, T *D, int ldd);
template <typename T>
int launch_det(lsx_handle_t h, int n, const T *LU, int lda, const int32_t *d_ipiv, double *d_out);
template <typename T>
int launch_rref(lsx_handle_t h, int m, int n, int bar, T *R, int ldr, int32_t *d_pivots,
                int *d_rank, double tol, int pivot_rule);
// d_out[0] = max|A_ij| ; d_out[1] = min_k |LU_kk|
template <typename T>
int launch_amax(lsx_handle_t h, int m, int n, const T *A, int lda, double *d_out);
template <typename T>
int launch_diag_minabs(lsx_handle_t h, int n, const T *LU, int lda, double *d_out);
// X (n x nrhs, ld = nrhs) <- U^-1 L^-1 B, nrhs <= 8, one launch per 128-row block step
template <typename T>
int lu_solve_few_rhs2(lsx_handle_t h, int n, int nrhs, int nr, const T *LU, int lda, const int32_t *d_ipiv, T *B, int ldb,
                      T *inv64L, T *inv64U, T *inv128L, T *inv128U, T *Bp, T *Y);
template <typename T>
int lu_solve_few_rhs(lsx_handle_t h, int n, int nrhs, const T *LU, int lda, T *B, int ldb, T *X, T *inv64L,
                     T *inv64U, T *inv128L, T *inv128U);
int diag_mfma_peak(lsx_handle_t h, int is_f32, int iters, int blocks_per_cu, double *tflops, double *clock_mhz);
int diag_cu_mask_probe(lsx_handle_t h, const uint32_t *mask_words, int nwords, int nblocks, unsigned *out_host);
int getrf_mg_f64(lsx_handle_t *hs, int P, int n, double *const *dA, const int *lda, int32_t *const *d_ipiv,
                 int *const *d_info);
int diag_occupy(lsx_handle_t h, int xcc, int wgs, int ms);
int diag_xchg_probe(lsx_handle_t h, int mode, int G, int stride, int wt, int epochs, double *us_per_epoch,
                    int *xcc_ids, int *nfail);
template <typename T>
int launch_copy2d(lsx_handle_t h, int m, int n, const T *S, int lds, T *D, int ldd);

}  // namespace lsx
