// Trailing-submatrix update  C (m x n) -= A (m x k) * B (k x n)  on the MFMA pipe.
//
// Takes the place of the reference's elimination loop
// (linalg_solver/linalg.py:587-596: row_k -= f * row_p for every row below the
// pivot, one pivot at a time) applied for a whole block of `k` pivots at once:
// A = L21 (the multipliers f of k pivot columns), B = U12 (the k pivot rows).
//
// gfx950 design
//   - v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, 64-lane wavefronts.
//   - workgroup = 4 waves, macro tile 128 x 128, each wave a 64 x 64 quadrant
//     held as 4 x 4 MFMA accumulators (128 VGPRs in fp64); 2 workgroups per CU.
//   - the accumulators are initialised FROM C (one HBM read) and A is negated
//     on its way into LDS, so the k-loop computes C - A*B in place and C is
//     written exactly once: algorithmic HBM traffic = 2 * sizeof(T) * m * n.
//   - A / B k-slabs (BK = 16) are staged global -> registers -> LDS with one
//     register set in flight across the MFMA block (guide T14), two LDS
//     buffers, one barrier per slab.  LDS rows are padded so the 32-lane halves
//     of a ds_read_b64 land on disjoint banks.
//   - within a wave the 4 N-tiles interleave columns (column = 4*(lane&15)+t),
//     so each lane owns 4 consecutive C elements per row and C moves in 16/32-B
//     pieces per lane, 512 B contiguous per row per wave.
//   - 1-D grid with an XCD-aware, grouped tile order: blocks that share an XCD
//     (blockIdx % 8) walk a compact band of tiles, so L21 / U12 slabs are
//     served from that XCD's L2 instead of HBM.
#include "common.h"

namespace lsx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
    typedef d4 acc_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // C/D element r of a lane sits at row (lane>>4) + 4*r, column lane&15
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float> {
    typedef f4 acc_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // f32 form: row 4*(lane>>4) + r, column lane&15
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LPAD = 16;  // (BM + LPAD) * sizeof(double) / 4 == 32 (mod 64) banks
// A slab in LDS: k-rows in pairs, As[buf][k/2][(k%2) * (BM+LPAD) + m], each pair padded by 16 bytes.
// A ds_write_b64 is banked (addr/4) mod 32 over groups of 16 lanes, and the staging map puts 8 k-pairs
// x 2 rows in a group: with a pair stride of 0 (mod 32) dwords those were 8-way conflicts (47 % of all
// LDS cycles in the PMC run); a stride of 4 (mod 32) dwords makes them conflict-free.  The reads only
// see the (k%2) stride within a lane group, which stays 32 (mod 64) dwords.
template <typename T, int BM_>
struct ASlab {
    static constexpr int PAIR = 2 * (BM_ + LPAD) + 16 / (int)sizeof(T);
};
#define AS_AT(buf, k, m) As[buf][(k) >> 1][((k) & 1) * (BM_ + LPAD) + (m)]

// One 128 x 128 tile.  FULL = the tile lies inside the matrix, K is a multiple of BK and all
// three operands are 16-byte aligned with even leading dimensions: every load/store is an
// unpredicated 16-byte access.  Otherwise every element is bounds-checked (edge tiles, odd ld).
// NWN = waves along N (2 or 4): the workgroup has (BM_/WM)*NWN waves, each owning a WM x (128/NWN) piece, WM = 64 rows
// (32 for the 32-row tile).
// BM_ = tile height: 128; 64 or 32 for skinny updates (the next panel's column block: a K = 128 tile of 64 rows is
// 6.8 us of a CU's MFMA time and 126 of them leave half the chip idle; 252 tiles of 32 rows take half that each).
// ticket_ctr != nullptr (work-queue kernel): thread 0 draws the workgroup's NEXT ticket from that counter while the
// last slab is being multiplied and leaves it in *s_next before the C stores -- a device-scope atomic takes 2-3 us
// to return, which drawn between two tiles would be ~10 % of a tile with nothing to cover it.
template <typename T, bool FULL, int NWN, int BM_>
__device__ __forceinline__ void gemm_sub_tile(int M, int N, int K, const T *__restrict__ A, int lda,
                                              const T *__restrict__ B, int ldb, T *__restrict__ C, int ldc,
                                              int m0, int n0, T (*As)[BK / 2][ASlab<T, BM_>::PAIR], T (*Bs)[BK][BN + LPAD],
                                              int plus, int *ticket_ctr = nullptr, int *s_next = nullptr) {
    typedef typename Mfma<T>::acc_t acc_t;
    typedef T v2 __attribute__((ext_vector_type(2)));
    constexpr int WM = BM_ < 64 ? BM_ : 64;   // rows per wave
    constexpr int SR = WM / 16;               // 16-row MFMA tiles per wave
    constexpr int NT = (BM_ / WM) * NWN * 64; // threads
    constexpr int WN = BN / NWN;       // wave tile width: 64 or 32
    constexpr int TN = WN / 16;        // N-tiles per wave: 4 or 2
    constexpr int NLA = BM_ * 8 / NT;  // 16-byte staging loads per thread, A slab (BM_ x 16)
    constexpr int NL = 1024 / NT;      // the same for the B slab (16 x 128): 4 or 2
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // plus bits 16..27: the k index runs kshift, ..., K - 1, 0, ..., kshift - 1 instead of 0 ... K - 1 (a multiple of BK when
    // FULL).  One update of depth K then has the bits of two consecutive updates whose SECOND operand block comes first in
    // memory (block back-substitution upwards: api.hip, getri_dev).
    const int kshift = (plus >> 16) & 0xfff;
    plus &= 0xff;
    auto kmap = [&](int k) { const int s = k + kshift; return s >= K ? s - K : s; };
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int lc = lane & 15, lq = lane >> 4;

    // ---- staging maps
    // A slab 128 x 16: thread -> row (tid>>3) + (NT/8)*i, k-pair (tid&7)*2
    // B slab 16 x 128: thread -> k (tid>>6) + (NT/64)*i, column pair (tid&63)*2
    const int a_row = tid >> 3, a_k = (tid & 7) * 2;
    const int b_k = tid >> 6, b_n = (tid & 63) * 2;
    T ra0[NLA][2], rb0[NL][2];   // one staging set: slab kt+1 is in flight under the MFMAs of slab kt
    const T sgn = plus ? T(1) : T(-1);
    unsigned a_off[NLA], b_off[NL];
#pragma unroll
    for (int i = 0; i < NLA; ++i) a_off[i] = (unsigned)(a_row + (NT / 8) * i) * (unsigned)lda + a_k;
#pragma unroll
    for (int i = 0; i < NL; ++i) b_off[i] = (unsigned)(b_k + (NT / 64) * i) * (unsigned)ldb + b_n;

    auto load_slab = [&](int k0, T (&ra)[NLA][2], T (&rb)[NL][2]) __attribute__((always_inline)) {
        if (FULL) {
            // uniform 64-bit base + 32-bit lane offset: the addresses cost NL VGPRs per operand, not 2*NL per set
            const T *Au = A + (size_t)m0 * lda + kmap(k0);
            const T *Bu = B + (size_t)kmap(k0) * ldb + n0;
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                const v2 v = *(const v2 *)(Au + a_off[i]);
                ra[i][0] = v[0]; ra[i][1] = v[1];
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const v2 v = *(const v2 *)(Bu + b_off[i]);
                rb[i][0] = v[0]; rb[i][1] = v[1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                const int row = m0 + a_row + (NT / 8) * i;
                const int kk = k0 + a_k;
                const T *p = A + (size_t)row * lda;
                ra[i][0] = (row < M && kk < K) ? p[kmap(kk)] : T(0);
                ra[i][1] = (row < M && kk + 1 < K) ? p[kmap(kk + 1)] : T(0);
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int kk = k0 + b_k + (NT / 64) * i;
                const int col = n0 + b_n;
                const T *p = B + (size_t)(kk < K ? kmap(kk) : 0) * ldb + col;
                rb[i][0] = (kk < K && col < N) ? p[0] : T(0);
                rb[i][1] = (kk < K && col + 1 < N) ? p[1] : T(0);
            }
        }
    };
    // column n of the tile -> wave piece h = n/WN, c = (n%WN)/TN, t = n%TN -> LDS column WN*h + 16*t + c
    auto bperm = [](int n) { return (n / WN) * WN + 16 * (n % TN) + (n % WN) / TN; };
    const int bp0 = bperm(b_n), bp1 = bperm(b_n + 1);
    auto store_slab = [&](int buf, T (&ra)[NLA][2], T (&rb)[NL][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {  // sign folded into A: the k-loop then accumulates C -+ A*B
            AS_AT(buf, a_k, a_row + (NT / 8) * i) = sgn * ra[i][0];
            AS_AT(buf, a_k + 1, a_row + (NT / 8) * i) = sgn * ra[i][1];
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            Bs[buf][b_k + (NT / 64) * i][bp0] = rb[i][0];
            Bs[buf][b_k + (NT / 64) * i][bp1] = rb[i][1];
        }
    };

    // first slab goes out before the C loads so both latencies overlap
    const int nslab = (K + BK - 1) / BK;
    load_slab(0, ra0, rb0);

    // ---- accumulators <- C.  Lane owns columns wn + TN*lc + t (t = N-tile index).
    acc_t acc[SR][TN];
#pragma unroll
    for (int s = 0; s < SR; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm + 16 * s + Mfma<T>::crow(lane, r);
            const int col = n0 + wn + TN * lc;
            const T *p = C + (size_t)row * ldc + col;
            T v[TN];
#pragma unroll
            for (int t = 0; t < TN; ++t) v[t] = T(0);
            if (FULL) {
#pragma unroll
                for (int t = 0; t < TN; t += 2) {
                    const v2 x = *(const v2 *)(p + t);
                    v[t] = x[0]; v[t + 1] = x[1];
                }
            } else if (row < M) {
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    if (col + t < N) v[t] = p[t];
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) acc[s][t][r] = v[t];
        }

    // One slab of cover is enough: a 16-deep slab is 32 MFMAs of 64 cycles per wave, 1.7-3.4 us per
    // workgroup with two workgroups sharing the SIMDs, against ~1 us for an L2 hit under load (a second
    // register set in flight was tried: it spills at the 128-VGPR budget and cannot gain).
    store_slab(0, ra0, rb0);
    __syncthreads();
    int ticket = 0;
    for (int kt = 0; kt < nslab; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nslab) load_slab((kt + 1) * BK, ra0, rb0);
        else if (ticket_ctr && tid == 0) ticket = __hip_atomic_fetch_add(ticket_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            T a[SR], b[TN];
#pragma unroll
            for (int s = 0; s < SR; ++s) a[s] = AS_AT(buf, kk + lq, wm + 16 * s + lc);
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = Bs[buf][kk + lq][wn + 16 * t + lc];
#pragma unroll
            for (int s = 0; s < SR; ++s)
#pragma unroll
                for (int t = 0; t < TN; ++t) acc[s][t] = Mfma<T>::mma(a[s], b[t], acc[s][t]);
        }
        if (kt + 1 < nslab) store_slab(buf ^ 1, ra0, rb0);
        __syncthreads();
    }

    // ---- C <- accumulators.  The lane id is laundered so the row addresses are recomputed here
    // instead of being kept alive (and spilled to scratch) across the whole k-loop.
    if (ticket_ctr && tid == 0) *s_next = ticket;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lc_e = lane_e & 15;
#pragma unroll
    for (int s = 0; s < SR; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm + 16 * s + Mfma<T>::crow(lane_e, r);
            const int col = n0 + wn + TN * lc_e;
            T *p = C + (size_t)row * ldc + col;
            if (FULL) {
#pragma unroll
                for (int t = 0; t < TN; t += 2) {
                    v2 x;
                    x[0] = acc[s][t][r]; x[1] = acc[s][t + 1][r];
                    *(v2 *)(p + t) = x;
                }
            } else if (row < M) {
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    if (col + t < N) p[t] = acc[s][t][r];
            }
        }
}

// FULL = true: interior tiles only (grid covers tiles_m x tiles_n complete tiles, operands aligned);
// FULL = false: any tile, every access bounds-checked.  Two kernels rather than one branch so the
// interior kernel's register allocation is not set by the edge path (it spilled inside 128 VGPRs).
// (tm_off, tn_off) shift the tile grid so edge strips can be covered by separate launches.
template <typename T, int NWN, bool FULL, int BM_ = BM>
// waves per SIMD: 2 workgroups per CU, 3 for the fp32 8-wave form (70 VGPRs, 37 KB of LDS)
__global__ __launch_bounds__((BM_ < 64 ? 1 : BM_ / 64) * NWN * 64, (BM_ <= 64) ? 2 : (sizeof(T) == 4 && NWN == 4) ? 6 : NWN) void gemm_sub_kernel(int M, int N, int K,
                                                          const T *__restrict__ A, int lda,
                                                          const T *__restrict__ B, int ldb,
                                                          T *__restrict__ C, int ldc, int tiles_m,
                                                          int tiles_n, int tm_off, int tn_off, int plus,
                                                          int *__restrict__ col0_done = nullptr) {
    LSX_TS(4);
    __shared__ T As[2][BK / 2][ASlab<T, BM_>::PAIR];  // AS_AT(buf, k, m) = -A[m][k]
    __shared__ T Bs[2][BK][BN + LPAD];  // Bs[buf][k][n] permuted: n' = 16*t + c  <-  column 4*c + t

    // ---- optional phase stagger (option "gemm_stagger", default 0).  Two workgroups share a CU;
    // launched together they run their memory-bound phases (C read, C write) and their MFMA phases in
    // step.  Delaying the second resident wave of the grid by a fraction of a tile de-phases them;
    // measured gain 2-4 % at K = 128, inside run-to-run noise, so it is off by default.
    {
        const int stagger = (plus >> 8) & 0xff;
        if (stagger > 0 && ((blockIdx.x >> 8) & 1))
            for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    plus &= ~0xff00;   // (bits 16..27, the k rotation, go on to the tile)

    // col0_done != nullptr (shared-CU look-ahead driver, tall phase): tile column 0 -- the next panel's columns -- goes
    // to the FIRST tiles_m workgroups and every finished tile of it is counted behind a device-scope fence; the panel
    // chain waits for that count inside its first kernel instead of for this whole launch.
    if (col0_done) {
        if ((int)blockIdx.x < tiles_m) {
            gemm_sub_tile<T, FULL, NWN, BM_>(M, N, K, A, lda, B, ldb, C, ldc, ((int)blockIdx.x + tm_off) * BM_, tn_off * BN, As, Bs, plus);
            __syncthreads();
            if (threadIdx.x == 0) {
                __threadfence();
                __hip_atomic_fetch_add(col0_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
    // ---- XCD-aware grouped tile order
    const int first_col = col0_done ? 1 : 0;            // columns left to the grouped order
    if (col0_done) tiles_n -= 1;
    const int nwg = tiles_m * tiles_n;
    int bid = (int)blockIdx.x - (col0_done ? tiles_m : 0);
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int group_id = bid / per_group;
    const int first_m = group_id * GROUP;
    const int gsize = min(tiles_m - first_m, GROUP);
    const int tile_m = first_m + (bid % per_group) % gsize;
    const int tile_n = (bid % per_group) / gsize;
    const int m0 = (tile_m + tm_off) * BM_, n0 = (tile_n + first_col + tn_off) * BN;
    gemm_sub_tile<T, FULL, NWN, BM_>(M, N, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs, plus);
}

// The same interior tiles handed out by a work queue instead of one workgroup per tile: the look-ahead driver
// runs the next panel on ONE XCD while this update runs (kernels_panel_x.hip), so an eighth of the chip joins
// late or not at all and a static tile -> workgroup map would leave that eighth's tiles for the end.
//   grid = resident workgroups (2 per CU); a workgroup whose XCC id is `avoid_xcc` leaves at once (the panel's
//   XCD: its CUs stay free for the panel's workgroups whatever the dispatch order of the two kernels);
//   queues: one counter per active XCD over a contiguous share of the tile sequence (the XCD-aware band order
//   above, so an XCD's workgroups walk compact bands and L21 / U12 slabs stay in its L2), own share first,
//   then the others' (work stealing), so the tail is balanced.
// The XCC id is read from the hardware register, never inferred from blockIdx.
template <typename T, int NWN>
__global__ __launch_bounds__(BM * NWN, (sizeof(T) == 4 && NWN == 4) ? 6 : NWN) void gemm_sub_queue_kernel(
    int M, int N, int K, const T *__restrict__ A, int lda, const T *__restrict__ B, int ldb, T *__restrict__ C,
    int ldc, int tiles_m, int tiles_n, int plus, int *__restrict__ counters, const int *__restrict__ avoid_word,
    int *__restrict__ pass_word, int *__restrict__ col0) {
    LSX_TS(6);
    __shared__ T As[2][BK / 2][ASlab<T, BM>::PAIR];
    __shared__ T Bs[2][BK][BN + LPAD];
    __shared__ int s_tile;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    // *avoid_word = 1 + XCC id of the last XCD-scope panel launch (0: none yet); written by that kernel
    const int avoid_xcc = avoid_word ? __hip_atomic_load(avoid_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1 : -1;
    if ((int)xcc == avoid_xcc) {
        // seen by the gate in front of the panel launch (kernels_misc.hip: gate_kernel)
        if (pass_word && threadIdx.x == 0) __hip_atomic_fetch_add(pass_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int nx = avoid_xcc >= 0 ? 7 : 8;
    const int rank = (avoid_xcc >= 0 && (int)xcc > avoid_xcc) ? (int)xcc - 1 : (int)xcc;
    // col0 != nullptr (look-ahead driver): tile column 0 -- the NEXT panel's columns -- is a queue of its own that
    // every workgroup serves first (tickets from col0[0]), and every finished tile of it is counted in col0[1]
    // behind a device-scope fence: the panel chain waits for that count (wait_count_kernel) instead of for this
    // whole kernel.  The other columns are shared out as below.  (One loop nest, one copy of the tile body: a separate
    // loop for column 0 doubled the kernel, spilled, and its control flow came out of the compiler hanging.)
    //
    // An XCD's share is a vertical STRIP of tile columns, walked row by row: its U12 columns (strip width x K x
    // sizeof(T): ~1 MB) stay in that XCD's L2 for the whole update and every L21 row block is fetched once per
    // strip -- ~9 MB of slab traffic per XCD.  (Bands of 8 tile rows over all columns, the static kernel's order,
    // re-read all of U12 per band and XCD: 8 MB x 8 bands; profiles/r01_pmc_gemm.json saw it as 1.2x traffic.
    // In time the two orders are level -- 423 us either way at 8064^2 on all XCDs, 363 us for the static grid, of
    // which ~28 us are the two memsets of the measurement: tools/kbench.py gemmq.)
    const int c_first = col0 ? 1 : 0;
    const int tn = tiles_n - c_first;
    for (int q = -c_first; q < nx; ++q) {
        const int owner = (rank + max(q, 0)) % nx;
        int c_lo = c_first + (int)((long long)owner * tn / nx), c_hi = c_first + (int)((long long)(owner + 1) * tn / nx);
        int *ctr = &counters[owner];
        if (q < 0) { c_lo = 0; c_hi = 1; ctr = col0; }
        const int sw = c_hi - c_lo;
        const int hi = tiles_m * sw;
        if (sw <= 0) continue;
        // first ticket of this queue: drawn here; the following ones are drawn inside the tile, under its last slab
        if (threadIdx.x == 0) s_tile = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        for (;;) {
            const int bid = s_tile;
            if (bid >= hi) break;   // uniform: every thread read the same word
            const int tile_m = bid / sw;
            const int tile_n = c_lo + bid % sw;
            gemm_sub_tile<T, true, NWN, BM>(M, N, K, A, lda, B, ldb, C, ldc, tile_m * BM, tile_n * BN, As, Bs, plus, ctr, &s_tile);
            __syncthreads();   // the next ticket is in s_tile; the tile's last LDS reads are done, its C stores issued
            if (q < 0 && threadIdx.x == 0) {
                __threadfence();
                __hip_atomic_fetch_add(col0 + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();       // everybody has read the exhausted ticket before the next queue's first one lands
    }
}

// Small / skinny problems (n < 16, e.g. a single right-hand side): plain FMA,
// one thread per C element, A row and B column streamed from L2.
template <typename T>
__global__ __launch_bounds__(256) void gemm_sub_skinny_kernel(int M, int N, int K,
                                                              const T *__restrict__ A, int lda,
                                                              const T *__restrict__ B, int ldb,
                                                              T *__restrict__ C, int ldc, int plus) {
    // one wave per row of C: lanes split K, shuffle-reduce, lane j<N writes column j
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T *a = A + (size_t)row * lda;
    for (int j0 = 0; j0 < N; j0 += 8) {
        T part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = lane; k < K; k += 64) {
            const T av = a[k];
            const T *b = B + (size_t)k * ldb + j0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j0 + j < N) part[j] += av * b[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            T v = part[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && j0 + j < N) C[(size_t)row * ldc + j0 + j] += plus ? v : -v;
        }
    }
}

template <typename T>
int launch_gemm_acc(lsx_handle_t h, int plus, int m, int n, int k, const T *A, int lda, const T *B, int ldb,
                    T *C, int ldc) {
    if (m <= 0 || n <= 0 || k <= 0) return LSX_OK;
    h->gemm_queue_used = 0;
    h->gemm_col0_complete = false;
    const bool skinny = n < 16 && !h->gemm_mfma_only && !h->gemm_kshift;
    if (h->gemm_kshift && (h->gemm_queue || h->gemm_kshift >= k || h->gemm_kshift > 0xfff || h->gemm_kshift % BK || (k - h->gemm_kshift) % BK)) {
        set_error("gemm: k rotation %d not served here (k = %d)", h->gemm_kshift, k);
        return LSX_ERR_INTERNAL;
    }
    const bool tiles64 = !h->gemm_queue && n >= 16 && sizeof(T) == 8 && (h->gemm_waves == 0 || h->gemm_waves == 8) && m % 64 == 0 &&
                         n % BN == 0 && k % BK == 0 && ((m + BM - 1) / BM) * (n / BN) <= h->num_cu / 2 &&
                         ((size_t)A % 16 == 0) && ((size_t)B % 16 == 0) && ((size_t)C % 16 == 0) && lda % 2 == 0 &&
                         ldb % 2 == 0 && ldc % 2 == 0;
    ProfScope ps(h, tiles64 ? LSX_PROF_GEMM_SKINNY : LSX_PROF_GEMM, 2.0 * m * n * (double)k,
                 2.0 * sizeof(T) * m * (double)n);
    if (skinny) {
        hipLaunchKernelGGL(gemm_sub_skinny_kernel<T>, dim3((m + 3) / 4), dim3(256), 0, h->stream, m,
                           n, k, A, lda, B, ldb, C, ldc, plus);
    } else {
        const int tm = (m + BM - 1) / BM, tn = (n + BN - 1) / BN;
        const int elems16 = 16 / (int)sizeof(T);
        const bool aligned = ((size_t)A % 16 == 0) && ((size_t)B % 16 == 0) && ((size_t)C % 16 == 0) &&
                             (lda % elems16 == 0) && (ldb % elems16 == 0) && (ldc % elems16 == 0) && (k % BK == 0);
        // fp64: 8 waves (64x32 per wave, 4 waves/SIMD hide the C read); fp32: 4 waves (64x64 per wave) --
        // the 8-wave fp32 form at 3 workgroups/CU is +4 % standalone and equal over an LU, so it stays opt-in
        const int waves = h->gemm_waves ? h->gemm_waves : (sizeof(T) == 8 ? 8 : 4);
        const int fm = aligned ? m / BM : 0, fn = aligned ? n / BN : 0;  // complete tiles
        // static grid with the counted first tile column (shared-CU look-ahead driver): only when the interior launch
        // covers that column completely, i.e. no edge strip below touches it
        int *col0s = (h->gemm_col0_static && !h->gemm_queue && waves == 8 && fm == tm && fm > 0 && fn > 1) ? h->gemm_col0_static : nullptr;
        h->gemm_col0_complete = col0s != nullptr;
        h->gemm_col0_tiles = fm;
        auto go = [&](bool full, int gm, int gn, int om, int on) {
            if (gm <= 0 || gn <= 0) return;
            const dim3 grid(gm * gn);
            if (waves == 8) {
                if (full) hipLaunchKernelGGL((gemm_sub_kernel<T, 4, true>), grid, dim3(512), 0, h->stream, m, n, k, A, lda, B, ldb, C, ldc, gm, gn, om, on, plus | (h->gemm_stagger << 8) | (h->gemm_kshift << 16), col0s);
                else hipLaunchKernelGGL((gemm_sub_kernel<T, 4, false>), grid, dim3(512), 0, h->stream, m, n, k, A, lda, B, ldb, C, ldc, gm, gn, om, on, plus | (h->gemm_stagger << 8) | (h->gemm_kshift << 16));
            } else {
                if (full) hipLaunchKernelGGL((gemm_sub_kernel<T, 2, true>), grid, dim3(256), 0, h->stream, m, n, k, A, lda, B, ldb, C, ldc, gm, gn, om, on, plus | (h->gemm_stagger << 8) | (h->gemm_kshift << 16));
                else hipLaunchKernelGGL((gemm_sub_kernel<T, 2, false>), grid, dim3(256), 0, h->stream, m, n, k, A, lda, B, ldb, C, ldc, gm, gn, om, on, plus | (h->gemm_stagger << 8) | (h->gemm_kshift << 16));
            }
        };
        // skinny updates (the next panel's column block in the look-ahead driver, block rows in the sharded
        // one): 128 x 128 tiles would leave most CUs idle, 64-row tiles double the workgroups
        if (tiles64) {
            if (m % 32 == 0 && (m / 32) * tn <= h->num_cu && !h->gemm_no_tiles32)   // even skinnier: one 32-row tile per CU
                hipLaunchKernelGGL((gemm_sub_kernel<T, 4, true, 32>), dim3((m / 32) * tn), dim3(256), 0, h->stream, m, n, k,
                                   A, lda, B, ldb, C, ldc, m / 32, tn, 0, 0, plus | (h->gemm_kshift << 16));
            else
                hipLaunchKernelGGL((gemm_sub_kernel<T, 4, true, 64>), dim3((m / 64) * tn), dim3(256), 0, h->stream, m, n, k,
                                   A, lda, B, ldb, C, ldc, m / 64, tn, 0, 0, plus | (h->gemm_kshift << 16));
            LSX_HIP(hipGetLastError());
            return LSX_OK;
        }
        if (h->gemm_queue && fm > 0 && fn > 0 && h->gemm_counters) {
            h->gemm_queue_used = 1;
            // the column-0 count covers that column only if no edge launch below touches it
            h->gemm_col0_complete = (h->gemm_col0 != nullptr) && fm == tm;
            h->gemm_col0_tiles = fm;
            // interior tiles through the work queue (look-ahead driver; counters zeroed by the driver)
            const dim3 grid(2 * h->num_cu);
            int *ctr = h->gemm_counters + 8 * (h->gemm_counter_set++ % h->gemm_counter_sets);
            if (waves == 8)
                hipLaunchKernelGGL((gemm_sub_queue_kernel<T, 4>), grid, dim3(512), 0, h->stream, m, n, k, A, lda, B, ldb, C, ldc, fm, fn, plus, ctr, h->gemm_avoid_word, h->gemm_pass_word, h->gemm_col0);
            else
                hipLaunchKernelGGL((gemm_sub_queue_kernel<T, 2>), grid, dim3(256), 0, h->stream, m, n, k, A, lda, B, ldb, C, ldc, fm, fn, plus, ctr, h->gemm_avoid_word, h->gemm_pass_word, h->gemm_col0);
        } else {
            go(true, fm, fn, 0, 0);                 // interior
        }
        go(false, tm - fm, tn, fm, 0);          // bottom strip (all columns)
        go(false, fm, tn - fn, 0, fn);          // right strip (complete tile rows only)
    }
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

template <typename T>
int launch_gemm_sub(lsx_handle_t h, int m, int n, int k, const T *A, int lda, const T *B, int ldb, T *C,
                    int ldc) {
    return launch_gemm_acc<T>(h, 0, m, n, k, A, lda, B, ldb, C, ldc);
}

template int launch_gemm_acc<double>(lsx_handle_t, int, int, int, int, const double *, int, const double *,
                                     int, double *, int);
template int launch_gemm_acc<float>(lsx_handle_t, int, int, int, int, const float *, int, const float *, int,
                                    float *, int);
template int launch_gemm_sub<double>(lsx_handle_t, int, int, int, const double *, int,
                                     const double *, int, double *, int);
template int launch_gemm_sub<float>(lsx_handle_t, int, int, int, const float *, int, const float *,
                                    int, float *, int);

}  // namespace lsx

LSX_TS_SETTER(gemm)
