/* lsx.h -- C ABI of the MI355X-native dense LU / row-reduction library (liblsx.so)
 *
 * This is the drop-in boundary for the numeric core of koskja/linalg-solver's
 *   Matrix.row_reduce()        linalg_solver/linalg.py:534-630
 *   Matrix.find_preimage_of()  linalg_solver/linalg.py:632-680
 *   Matrix.inverse()           linalg_solver/linalg.py:682-743
 *   Matrix.determinant()       linalg_solver/linalg.py:183-207
 *   Matrix.rank()              linalg_solver/linalg.py:745-747
 * It takes the place of the reference's only native slot, the PyO3 module
 * `linalg_helper` (linalg-helper/src/lib.rs:122-143, built by
 * pyproject.toml:18-21): a shared library next to the Python package, entered
 * through plain C symbols (ctypes; see INTEGRATION.md for the binding).
 *
 * Conventions
 *   - Matrices are ROW-MAJOR (the reference stores list-of-rows), element (i,j)
 *     at base[i*ld + j]; ld >= number of columns.
 *   - Every call returns an int status: 0 = ok, <0 = bad argument / HIP error
 *     (text via lsx_last_error()).  Numerical outcomes (singular matrix, rank)
 *     are reported through out-parameters, never through the status, mirroring
 *     the reference's "results are values, not exceptions" rule
 *     (README.md:196-204, linalg.py:673,701,737).
 *   - Caller owns every buffer it passes; the library never frees or keeps them.
 *   - Host-buffer entry points copy to the device, compute there and copy back.
 *     The *_dev entry points work on device pointers already resident in HBM and
 *     are asynchronous on the handle's stream.
 *   - One handle = one GPU = one host thread at a time.  Handles are independent (every call makes the handle's device current for its duration and restores the caller's).
 *   - There is NO CPU fallback: without a usable gfx950 device lsx_create fails.
 */
#ifndef LSX_H
#define LSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSX_VERSION 100

typedef struct lsx_handle_s *lsx_handle_t;

enum {
    LSX_OK = 0,
    LSX_ERR_ARG = -1,     /* invalid argument */
    LSX_ERR_HIP = -2,     /* a HIP runtime call failed */
    LSX_ERR_NODEVICE = -3,/* no gfx950 device / wrong architecture */
    LSX_ERR_ALLOC = -4,   /* device allocation failed */
    LSX_ERR_INTERNAL = -5 /* in-kernel protocol timeout or invariant violated */
};

/* pivot rules of lsx_rref_* */
enum { LSX_PIVOT_FIRST = 0, LSX_PIVOT_MAX = 1 };

/* input generators: counter-based, identical on host and device */
enum { LSX_FILL_INT5 = 0, LSX_FILL_U11 = 1 };

/* profiling buckets for lsx_prof_read */
enum {
    LSX_PROF_PANEL = 0,  /* panel factorisation kernels */
    LSX_PROF_LASWP = 1,  /* row interchanges outside the panel */
    LSX_PROF_TRSM = 2,   /* triangular inverse + U12 / block solves */
    LSX_PROF_GEMM = 3,   /* trailing update C -= A*B (MFMA) */
    LSX_PROF_OTHER = 4,
    LSX_PROF_GEMM_SKINNY = 5, /* the same update on 64-row tiles (few-tile shapes: the next panel's column block) */
    LSX_PROF_NBUCKETS = 6
};

/* ---- lifetime ---------------------------------------------------------- */
int lsx_device_count(void);
int lsx_create(lsx_handle_t *out, int device);
int lsx_destroy(lsx_handle_t h);
/* Enqueue on an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).
 * NULL is a valid value: HIP's default (null) stream, which is torch's default too. */
int lsx_set_stream(lsx_handle_t h, void *hip_stream);
/* Go back to the private non-blocking stream created by lsx_create. */
int lsx_use_own_stream(lsx_handle_t h);
int lsx_synchronize(lsx_handle_t h);
/* Thread-local text of the last failure on this thread ("" if none). */
const char *lsx_last_error(void);
/* Tunables (defaults in brackets; every variant gives bit-identical factors):
 *   "nb"            panel width, <= 128 [128]
 *   "panel"         0 = per-column launches (the fallback after an exchange time-out), 3 = one launch per panel,
 *                   device-scope pivot exchange (panels taller than one XCD holds, multi-device driver),
 *                   4 = one launch per panel on ONE XCD with the exchange in that XCD's L2 [4]
 *                   (1, 2: superseded kernels, only in `make DIAG=1` builds)
 *   "lookahead"     0 = sequential driver, 1 = the next panel is factored on a side stream under the trailing
 *                   update [1]
 *   "lookahead_min" smallest n that takes the look-ahead driver; 0 = the measured break-even: 2048 in fp64 and
 *                   4096 in fp32 with panel = 4 (7168 / 10240 with panel = 3) [0]
 *   "kblock"        panels per trailing update [1]
 *   "trsv"          few-right-hand-side solve: 2 = 128-row steps with helper workgroups, one preparation launch [2],
 *                   1 = one cooperative launch per direction with 64-row steps, 0 = one launch per 128-row step
 *   "left_per_step" look-ahead driver: a panel's interchanges on the columns left of it trail its step [1]; 0 = all panels'
 *                   in one launch at the end
 *   "getri_pairs"   inverse / many-right-hand-side solve at large regular orders: 1 = two 128-row blocks per trailing
 *                   update (depth 256, same bits, 8192^2 inverse 17.6 -> 15.8 ms) [1], 0 = one
 *   "panel_col"     XCD panels of up to 4096 rows: 0 = rows distributed over the workgroups [0], 1 = columns distributed
 *                   (an independent second implementation, slower: kept as a cross-check; "panel_col_wt" = 1 runs it
 *                   with write-through stores, lsx_get_option "panel_col_launches" counts the panels it took)
 *   "gemm_waves", "gemm_stagger", "panel_rt", "panel_nt", "hybrid", "xrows_limit", "rref_blocked",
 *   "getri_structured"                      tuning / cross-check switches, see DESIGN.md
 *   "panel_spin_limit", "trsv_spin_limit", "chain_wait_limit"   bounded-spin limits (tests inject time-outs)
 * Returns LSX_ERR_ARG for unknown keys or values. */
int lsx_set_option(lsx_handle_t h, const char *key, int value);
/* The cooperative kernels (panel pivot exchange, few-right-hand-side solve) poll each other with a bounded spin; a
 * time-out -- their workgroups were not all resident, e.g. another kernel held the CUs -- is recorded in a device
 * word.  The host-buffer entry points check it themselves and return LSX_ERR_INTERNAL instead of a result; after
 * *_dev calls (asynchronous, no info word required) this synchronises the handle's stream and reports it:
 * LSX_OK, or LSX_ERR_INTERNAL once (the word is cleared). */
int lsx_check_status(lsx_handle_t h);
int lsx_get_option(lsx_handle_t h, const char *key, int *value);

/* ---- host-buffer entry points (fp64) ------------------------------------ */
/* In-place LU with partial pivoting, P*A = L*U, unit-lower L below the diagonal.
 * ipiv[k] = 0-based row exchanged with row k at step k.  *info = 0, or k+1 for
 * the first k whose pivot was exactly zero (factorisation continues). */
int lsx_getrf_f64(lsx_handle_t h, int n, double *A, int lda, int32_t *ipiv, int *info);
/* Solve A X = B from the factors; B (n x nrhs, row-major) is overwritten by X. */
int lsx_getrs_f64(lsx_handle_t h, int n, int nrhs, const double *LU, int lda,
                  const int32_t *ipiv, double *B, int ldb);
/* Factor + solve; A is not modified.  Replaces row_reduce([A|B], bar_col=n) for
 * square non-singular A (find_preimage_of, linalg.py:649-656). */
int lsx_gesv_f64(lsx_handle_t h, int n, int nrhs, const double *A, int lda,
                 double *B, int ldb, int *info, double *pivot_ratio);
/* Inverse; replaces row_reduce([A|I], bar_col=n) (inverse, linalg.py:704-711).
 * *info > 0: singular to working precision (caller returns NoSolution). */
int lsx_getri_f64(lsx_handle_t h, int n, const double *A, int lda, double *Ainv, int ldi,
                  int *info, double *pivot_ratio);
/* det(A) = sign * mant * 2^exp2 with mant in [0.5,1) (or 0): never overflows.
 * Replaces Matrix.determinant for numeric dense input (linalg.py:183-207). */
int lsx_det_f64(lsx_handle_t h, int n, const double *A, int lda, double *sign, double *mant,
                int64_t *exp2);
/* General m x n reduced row echelon form over columns [0,bar_col) with the
 * remaining columns carried along (row_reduce, linalg.py:534-630).  bar_col <= 0
 * means n-1 (linalg.py:543).  pivots holds *rank pairs (row, col), 0-based.
 * Pivot tolerance: |a| <= tol counts as zero; tol < 0 selects 32*eps*max(m,n)*max|working matrix|
 * (the running maximum, re-evaluated as the elimination proceeds).
 * pivot_rule LSX_PIVOT_FIRST takes the first row at or below the pivot row whose
 * entry is non-zero -- the reference's rule (linalg.py:548-552), which also fixes
 * the carried-along columns of rank-deficient / tall inputs; LSX_PIVOT_MAX takes
 * the largest magnitude (same pivot positions, better conditioning). */
int lsx_rref_f64(lsx_handle_t h, int m, int n, int bar_col, const double *A, int lda,
                 double *R, int ldr, int32_t *pivots, int *rank, double tol, int pivot_rule);
/* The fp32 forms of the three callers above (BASELINE config 5 computes in fp32): same meaning, fp32 MFMA tile. */
int lsx_getri_f32(lsx_handle_t h, int n, const float *A, int lda, float *Ainv, int ldi,
                  int *info, double *pivot_ratio);
int lsx_det_f32(lsx_handle_t h, int n, const float *A, int lda, double *sign, double *mant,
                int64_t *exp2);
int lsx_rref_f32(lsx_handle_t h, int m, int n, int bar_col, const float *A, int lda,
                 float *R, int ldr, int32_t *pivots, int *rank, double tol, int pivot_rule);
/* Mixed-precision solve: P A = L U in fp32, then `sweeps` (0..16, 2-3 suffice) corrections with the residual
 * b - A x accumulated in fp64 -- the forward error goes from cond(A) * eps32 to the fp32 rounding of the solution
 * of the fp64 system (config 5's "tolerance 1e-4 vs the fp64 result" is about the solution, which an unrefined fp32
 * solve of a random 8192 x 8192 system misses).  X (fp32, n x nrhs) and / or X64 (fp64) receive the solution;
 * last_correction = max|d| / max|x| of the last sweep (a convergence check: it should be ~eps32 or smaller). */
int lsx_gesv_f32_refined(lsx_handle_t h, int n, int nrhs, const float *A, int lda, const float *B, int ldb,
                         float *X, int ldx, double *X64, int ldx64, int sweeps, int *info,
                         double *pivot_ratio, double *last_correction);
/* Traced row reduction: Matrix.row_reduce in the reference's own operation order (exact-zero pivot
 * test with first-non-zero row swap, normalise, eliminate below, separate backward pass;
 * linalg.py:547-629) with one IEEE rounding per operation, so R, the pivots and the step log are
 * bit-identical to the reference's float arithmetic.  steps receives *nsteps records of 4 int32
 * {kind, a, b, 0}: kind 0 = S (rows a and b swapped, 1-based), 1 = N (pivot row a normalised),
 * 2 = E below the pivot in column a, 3 = E above the pivot in column a (linalg.py:556-606, :624-628).
 * If max_snaps > 0, snaps receives the dense m x n matrix after each of the first max_snaps steps
 * (the reference's intermediate_matrices, linalg.py:553-555 etc.).  max_steps >= 4*min(m,bar_col)
 * always suffices; LSX_ERR_ARG if the log is too small.  int_mask (m x n bytes, may be NULL = all
 * floats) says on entry which entries are Python ints and on return which still are: the reference
 * computes on Python objects, so int - int*int stays an int and only division or a float operand
 * makes a float; snap_int_mask (max_snaps x m x n bytes) is the same per
 * snapshot.  A host layer uses the masks to hand ints back.  The trace path is for small and medium inputs (several launches per column,
 * HBM-bound); the fast paths are lsx_gesv_f64 / lsx_rref_f64. */
int lsx_rref_trace_f64(lsx_handle_t h, int m, int n, int bar_col, const double *A, int lda, double *R,
                       int ldr, unsigned char *int_mask, int32_t *pivots, int *npivots, int32_t *steps,
                       int max_steps, int *nsteps, double *snaps, unsigned char *snap_int_mask,
                       int max_snaps);

/* C (m x n) = A (m x k) * B (k x n) on the MFMA tile of the trailing update.  Replaces the numeric
 * core of Matrix.__mul__ (linalg.py:101-158); used for residual checks A*x - b, A*inv(A) - I. */
int lsx_matmul_f64(lsx_handle_t h, int m, int n, int k, const double *A, int lda, const double *B, int ldb,
                   double *C, int ldc);

/* ---- host-buffer entry points (fp32; BASELINE config 5) ------------------- */
int lsx_getrf_f32(lsx_handle_t h, int n, float *A, int lda, int32_t *ipiv, int *info);
int lsx_getrs_f32(lsx_handle_t h, int n, int nrhs, const float *LU, int lda,
                  const int32_t *ipiv, float *B, int ldb);
int lsx_gesv_f32(lsx_handle_t h, int n, int nrhs, const float *A, int lda, float *B, int ldb,
                 int *info, double *pivot_ratio);

/* ---- device-pointer entry points (asynchronous on the handle's stream) ---- */
/* d_info: device int (may be NULL).  d_ipiv: device int32[n].
 * *d_info < 0 after the call means the in-kernel pivot exchange timed out (LSX_ERR_INTERNAL
 * in the host-buffer forms): the factors are not valid. */
int lsx_getrf_f64_dev(lsx_handle_t h, int n, double *dA, int lda, int32_t *d_ipiv, int *d_info);
int lsx_getrs_f64_dev(lsx_handle_t h, int n, int nrhs, const double *dLU, int lda,
                      const int32_t *d_ipiv, double *dB, int ldb);
int lsx_getri_f64_dev(lsx_handle_t h, int n, const double *dLU, int lda, const int32_t *d_ipiv,
                      double *dInv, int ldi);
/* d_out[0]=sign, d_out[1]=mant, d_out[2]=(double)exp2 */
/* fp32 forms of the two above, and the device-pointer mixed-precision solve: dA / dB are read, dLU receives the fp32
 * factors, dX64 (n x nrhs fp64) the solution, dX32 (may be NULL) its fp32 rounding, d_stats (4 doubles, may be
 * NULL) = max|d|, max|x| of the last sweep, then of the initial solve. */
int lsx_getri_f32_dev(lsx_handle_t h, int n, const float *dLU, int lda, const int32_t *d_ipiv,
                      float *dInv, int ldi);
int lsx_det_f32_dev(lsx_handle_t h, int n, const float *dLU, int lda, const int32_t *d_ipiv, double *d_out);
int lsx_gesv_f32_refined_dev(lsx_handle_t h, int n, int nrhs, const float *dA, int lda, float *dLU, int ldl,
                             int32_t *d_ipiv, int *d_info, const float *dB, int ldb, double *dX64, int ldx,
                             float *dX32, int ldf, int sweeps, double *d_stats);
int lsx_det_f64_dev(lsx_handle_t h, int n, const double *dLU, int lda, const int32_t *d_ipiv,
                    double *d_out);
int lsx_getrf_f32_dev(lsx_handle_t h, int n, float *dA, int lda, int32_t *d_ipiv, int *d_info);
int lsx_getrs_f32_dev(lsx_handle_t h, int n, int nrhs, const float *dLU, int lda,
                      const int32_t *d_ipiv, float *dB, int ldb);
int lsx_rref_f64_dev(lsx_handle_t h, int m, int n, int bar_col, double *dR, int ldr,
                     int32_t *d_pivots, int *d_rank, double tol, int pivot_rule);

/* Building blocks used by the multi-GPU driver (1-D block-cyclic columns,
 * SURVEY.md section 8e); all on device pointers, row-major. */
/* Factor the m x jb panel at dP (rows are global rows row0..row0+m-1); writes
 * d_ipiv[0..jb) as GLOBAL 0-based row indices and updates *d_info. */
int lsx_panel_f64_dev(lsx_handle_t h, int m, int jb, double *dP, int ldp, int row0,
                      int32_t *d_ipiv, int *d_info);
/* Apply the jb interchanges (k-th: row row0+k <-> d_ipiv[k]) to ncols columns of dA. */
int lsx_laswp_f64_dev(lsx_handle_t h, int ncols, double *dA, int lda, int row0, int jb,
                      const int32_t *d_ipiv);
/* The cooperative panel kernel also emits its interchanges as a gather list (256 x {dst, src}
 * row pairs relative to row0, -1 = unused).  lsx_panel_moves_dev copies the list of the LAST
 * lsx_panel_f64_dev call into d_moves (512 int32; *valid = 0 if that panel used the per-column
 * kernels and produced no list); lsx_laswp_moves_f64_dev applies such a list to ncols columns.
 * The multi-GPU driver ships the list inside the panel broadcast. */
int lsx_panel_moves_dev(lsx_handle_t h, int32_t *d_moves, int *valid);
int lsx_laswp_moves_f64_dev(lsx_handle_t h, int ncols, double *dA, int lda, int row0, const int32_t *d_moves);
/* dB (jb x ncols) <- inv(L11) * dB with L11 the unit-lower jb x jb block at dL. */
int lsx_trsm_lu_f64_dev(lsx_handle_t h, int jb, int ncols, const double *dL, int ldl, double *dB,
                        int ldb);
/* dC (m x n) -= dA (m x k) * dB (k x n) on the MFMA path. */
int lsx_gemm_sub_f64_dev(lsx_handle_t h, int m, int n, int k, const double *dA, int lda,
                         const double *dB, int ldb, double *dC, int ldc);
/* dC += dA * dB (same kernel, sign folded into the A operand). */
int lsx_gemm_add_f64_dev(lsx_handle_t h, int m, int n, int k, const double *dA, int lda,
                         const double *dB, int ldb, double *dC, int ldc);
int lsx_gemm_sub_f32_dev(lsx_handle_t h, int m, int n, int k, const float *dA, int lda,
                         const float *dB, int ldb, float *dC, int ldc);

/* Deterministic synthetic inputs written straight into HBM:
 * element (i,j) = f(splitmix64(seed*GOLDEN + (i+row_off)<<32 + (j+col_off))). */
int lsx_fill_f64_dev(lsx_handle_t h, int kind, uint64_t seed, int m, int n, double *dA, int lda,
                     int row_off, int col_off);
int lsx_fill_f32_dev(lsx_handle_t h, int kind, uint64_t seed, int m, int n, float *dA, int lda,
                     int row_off, int col_off);

/* ---- multi-GPU, one call ---------------------------------------------------------------
 * P A = L U of an n x n matrix distributed 1-D block-cyclic by columns over ndev devices of one node: column block
 * b (width nb = the handles' "nb" option, the same on all) lives on device b % ndev; dA[d] is device d's local
 * matrix, all n rows of its blocks side by side, row-major with leading dimension lda[d].  handles[d] was created on
 * device d (several handles on one device are allowed: rehearsal).  Per step the owner factors the panel and writes
 * it to every peer directly (hipMemcpyPeerAsync per link, in row chunks that the receivers consume as they land);
 * no torch.distributed, no collective library.  d_ipiv[d] (n entries, on device d) receives the full interchange
 * list on every device, d_info[d] the info word.  Synchronous: returns when every device has finished.  Factors and
 * pivots are bit-identical to lsx_getrf_f64_dev on one device. */
int lsx_getrf_mg_f64(lsx_handle_t *handles, int ndev, int n, double *const *dA, const int *lda,
                     int32_t *const *d_ipiv, int *const *d_info);

/* ---- measurement --------------------------------------------------------- */
/* When enabled, every kernel launch of the selected buckets is bracketed by HIP
 * events on the launch stream; lsx_prof_read sums them (synchronises).
 * on: 0 = off, 1 = all buckets, otherwise (bucket mask << 1), e.g. 2 << LSX_PROF_GEMM. */
int lsx_prof_enable(lsx_handle_t h, int on);
int lsx_prof_reset(lsx_handle_t h);
int lsx_prof_read(lsx_handle_t h, int bucket, double *ms, long long *launches, double *flops,
                  double *bytes);

/* Diagnostics: sustained MFMA rate of a register-resident loop (synchronous).
 * is_f32 = 0: v_mfma_f64_16x16x4_f64, 1: v_mfma_f32_16x16x4_f32. */
int lsx_diag_mfma_peak(lsx_handle_t h, int is_f32, int iters, int blocks_per_cu, double *tflops,
                       double *clock_mhz /* fp64 only: median in-kernel shader clock, may be NULL */);
/* Diagnostics: cost of one round of the cross-CU record exchange under the panel factorisation
 * (synchronous).  G participants = the workgroups with blockIdx % stride == 0 (stride 8: one XCD under
 * round-robin dispatch); mode 0 = header all-gather, 1 = header all-gather + the 128-granule row of a
 * rotating winner, 2 = ping-pong between two participants; write_through 1 = `sc1` stores (device scope),
 * 0 = plain stores (XCD scope: only meaningful when xcc_ids come back all equal).  xcc_ids[G] may be NULL. */
int lsx_diag_xchg_probe(lsx_handle_t h, int mode, int G, int stride, int write_through, int epochs,
                        double *us_per_epoch, int *xcc_ids, int *nfail);
/* Diagnostics (residency tests): hold `wgs` (1..32) CUs of XCD `xcc` for `ms` milliseconds with a filler kernel on
 * a stream of its own; returns at once. */
int lsx_diag_occupy(lsx_handle_t h, int xcc, int wgs, int ms);
/* Diagnostics: launch nblocks workgroups on a stream created with the given CU mask (nwords = 0: the handle's
 * stream) and report where each landed: out[2 b] = XCC id, out[2 b + 1] = HW_ID register.  Synchronous. */
int lsx_diag_cu_mask_probe(lsx_handle_t h, const uint32_t *mask_words, int nwords, int nblocks, uint32_t *out);
/* Diagnostics: the 64 x 64 block inverses of the unit-lower jb x jb triangle at dT, through the fused head of the
 * look-ahead chain (fused = 1: inverses + the gather list d_moves applied to ncols columns at dA in ONE launch) or
 * through their own launch (fused = 0).  Device pointers, asynchronous on the handle's stream. */
int lsx_diag_chain_head_f32(lsx_handle_t h, int fused, int jb, const float *dT, int ldt, float *dTinv, int ncols,
                            float *dA, int lda, int row0, const int32_t *d_moves);
int lsx_diag_chain_head_f64(lsx_handle_t h, int fused, int jb, const double *dT, int ldt, double *dTinv, int ncols,
                            double *dA, int lda, int row0, const int32_t *d_moves);
/* Copy a piece of the handle's device scratch to the host (stamped diagnostic builds). */
int lsx_diag_read_scratch(lsx_handle_t h, size_t offset, void *dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* LSX_H */
