// lsx_getrf_mg_f64: one process, P devices, one call.
//
// The same distribution and step order as the one-process-per-GPU driver (linalg_solver_amd/dist.py): 1-D
// block-cyclic by COLUMNS -- column block b (width nb) lives on device b % P, each device holding all n rows of
// its blocks as one row-major local matrix -- and per block step
//     owner     factors the panel (whole columns are local: the pivot search needs no exchange)
//     owner --> every peer: the factored panel + its gather list + pivots, as direct peer writes over xGMI
//               (hipMemcpyPeerAsync on one copy stream per link: a flat broadcast, the links are point-to-point)
//     everyone  interchanges on its other columns, U12 from L11, A22 -= L21 * U12 on the MFMA tile
// with look-ahead depth 1 (the owner of panel b+1 brings that block up to date, factors it and starts sending
// before it finishes its share of update b).  The panel travels in ROW CHUNKS: a receiver starts its interchanges
// and U12 behind the first chunk (header + top rows) and updates each row range as its chunk lands, so the
// transfer of a 16 MB panel overlaps the update instead of preceding it.
//
// Nothing here depends on torch.distributed; the factors and pivots are bit-identical to the single-GPU
// factorisation (same kernels per element; tests/test_dist_gpu.py rehearses it with several handles on one GPU).
#include <algorithm>
#include <vector>

#include "common.h"

namespace lsx {

namespace {

constexpr int MG_CHUNKS = 4;
constexpr size_t MG_HDR = 4096;   // bytes in front of the panel rows: gather list | info | pivots

struct Dev {
    lsx_handle_t h = nullptr;
    int dev = 0;
    hipStream_t comp = nullptr;                 // the handle's stream: all kernels of this device
    std::vector<hipStream_t> link;              // link[d]: copies this device -> device d
    char *buf[2] = {nullptr, nullptr};          // panel buffers (header + rows), alternating with the step
    hipEvent_t packed[2] = {nullptr, nullptr};  // buf[p] holds a packed panel (recorded on comp, owner side)
    hipEvent_t freed[2] = {nullptr, nullptr};   // this device is done reading buf[p] (recorded on comp)
    std::vector<hipEvent_t> landed;             // landed[(src * 2 + p) * MG_CHUNKS + c]: created on device src
    std::vector<int> blocks;                    // global block ids owned
    std::vector<int> offset;                    // local column offset per owned block (same order)
    int local_cols = 0;
};

struct Guard {
    int prev = -1;
    Guard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

// per-device building blocks (api.hip / kernels)
template <typename T>
int launch_panel(lsx_handle_t h, int m, int jb, T *P, int ldp, int row0, int32_t *d_ipiv, int *d_info);

static int mg_free(std::vector<Dev> &D, int rc) {
    for (auto &d : D) {
        (void)hipSetDevice(d.dev);
        if (d.comp) (void)hipStreamSynchronize(d.comp);
        for (auto s : d.link)
            if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        for (int p = 0; p < 2; ++p) {
            if (d.buf[p]) (void)hipFree(d.buf[p]);
            if (d.packed[p]) (void)hipEventDestroy(d.packed[p]);
            if (d.freed[p]) (void)hipEventDestroy(d.freed[p]);
        }
        for (auto e : d.landed)
            if (e) (void)hipEventDestroy(e);
    }
    return rc;
}

// Everything that can fail lives in this body; getrf_mg_f64 below runs it and then ALWAYS goes through mg_free, which
// waits for every compute and link stream of every device and releases the link streams, events and panel buffers:
// an error in the middle neither leaks them nor returns while kernels and peer copies are still queued.
static int getrf_mg_body(std::vector<Dev> &D, lsx_handle_t *hs, int P, int n, double *const *dA, const int *lda,
                         int32_t *const *d_ipiv, int *const *d_info) {
    const int nb = hs[0]->nb;
    const int nblocks = (n + nb - 1) / nb;
    const size_t buf_bytes = MG_HDR + sizeof(double) * (size_t)n * nb;
    for (int d = 0; d < P; ++d) {
        Dev &x = D[d];
        x.h = hs[d];
        x.dev = hs[d]->device;
        x.comp = hs[d]->stream;
        if (hs[d]->nb != nb) { set_error("getrf_mg: all handles must use the same nb"); return LSX_ERR_ARG; }
        for (int b = d; b < nblocks; b += P) {
            x.blocks.push_back(b);
            x.offset.push_back(x.local_cols);
            x.local_cols += std::min(nb, n - b * nb);
        }
        if (x.local_cols > 0 && (!dA[d] || lda[d] < x.local_cols)) {   // a device without a column block holds nothing
            set_error("getrf_mg: device %d holds %d columns: null matrix or lda too small", d, x.local_cols);
            return LSX_ERR_ARG;
        }
    }
    // resources
    for (int d = 0; d < P; ++d) {
        D[d].link.assign(P, nullptr);
        D[d].landed.assign((size_t)P * 2 * MG_CHUNKS, nullptr);
    }
    for (int d = 0; d < P; ++d) {
        Dev &x = D[d];
        LSX_HIP(hipSetDevice(x.dev));
        for (int e = 0; e < P; ++e) {
            if (e == d) continue;
            if (D[e].dev != x.dev) (void)hipDeviceEnablePeerAccess(D[e].dev, 0);   // already enabled is fine
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&x.link[e], hipStreamNonBlocking) != hipSuccess) { set_error("getrf_mg: stream creation failed"); return LSX_ERR_HIP; }
        }
        for (int p = 0; p < 2; ++p) {
            if (hipMalloc((void **)&x.buf[p], buf_bytes) != hipSuccess) { set_error("getrf_mg: hipMalloc"); return LSX_ERR_ALLOC; }
            if (hipEventCreateWithFlags(&x.packed[p], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&x.freed[p], hipEventDisableTiming) != hipSuccess) {
                set_error("getrf_mg: event creation failed");
                return LSX_ERR_HIP;
            }
        }
        // events of the copies THIS device issues (as source) towards each peer
        for (int e = 0; e < P; ++e)
            for (int p = 0; p < 2; ++p)
                for (int c = 0; c < MG_CHUNKS; ++c)
                    if (e != d && hipEventCreateWithFlags(&D[e].landed[((size_t)d * 2 + p) * MG_CHUNKS + c],
                                                          hipEventDisableTiming) != hipSuccess) {
                        set_error("getrf_mg: event creation failed");
                        return LSX_ERR_HIP;
                    }
        LSX_HIP(hipMemsetAsync(d_info[d], 0, sizeof(int), x.comp));
        LSX_HIP(hipEventRecord(x.freed[0], x.comp));
        LSX_HIP(hipEventRecord(x.freed[1], x.comp));
    }
    struct Modes {   // the look-ahead below shares a device's CUs between a panel and an update: device-scope panel
        std::vector<Dev> &D; std::vector<int> keep;
        explicit Modes(std::vector<Dev> &D_) : D(D_) { for (auto &d : D) { keep.push_back(d.h->panel_mode); if (d.h->panel_mode == 4) d.h->panel_mode = 3; } }
        ~Modes() { for (size_t i = 0; i < D.size(); ++i) D[i].h->panel_mode = keep[i]; }
    } modes(D);

    bool cur_list = false;   // the panel being applied came with a gather list (set per step)
    auto shape = [&](int b, int &k, int &jb, int &m) { k = b * nb; jb = std::min(nb, n - k); m = n - k; };
    // row range [r0, r1) of chunk c of a panel of m rows: multiples of 64 rows, chunk 0 holds at least the top jb rows
    auto chunk_rows = [&](int m, int jb, int c, int &r0, int &r1) {
        const int per = std::max(((m + MG_CHUNKS - 1) / MG_CHUNKS + 63) / 64 * 64, jb);
        r0 = std::min(m, c * per);
        r1 = std::min(m, (c + 1) * per);
        if (c == MG_CHUNKS - 1) r1 = m;
    };
    auto local_index = [&](const Dev &x, int b) { return (int)(std::lower_bound(x.blocks.begin(), x.blocks.end(), b) - x.blocks.begin()); };

    // owner: factor panel b in place, pack it, start the copies to every peer
    auto pack_and_send = [&](int b) -> int {
        int k, jb, m;
        shape(b, k, jb, m);
        const int o = b % P, p = b & 1;
        Dev &x = D[o];
        LSX_HIP(hipSetDevice(x.dev));
        const int li = local_index(x, b);
        double *Pn = dA[o] + (size_t)k * lda[o] + x.offset[li];
        // buf[p] must be free on the owner itself: its own update b-2 (comp, in order) and its copies of panel b-2
        for (int e = 0; e < P; ++e)
            if (e != o && D[e].landed[((size_t)o * 2 + p) * MG_CHUNKS + MG_CHUNKS - 1] && b >= 2 && (b - 2) % P == o)
                LSX_HIP(hipStreamWaitEvent(x.comp, D[e].landed[((size_t)o * 2 + p) * MG_CHUNKS + MG_CHUNKS - 1], 0));
        LSX_TRY(launch_panel<double>(x.h, m, jb, Pn, lda[o], k, d_ipiv[o] + k, d_info[o]));
        char *buf = x.buf[p];
        if (x.h->moves_valid) LSX_HIP(hipMemcpyAsync(buf, x.h->moves, 2048, hipMemcpyDeviceToDevice, x.comp));
        else LSX_HIP(hipMemsetAsync(buf, 0xff, 2048, x.comp));   // all (-1, -1): no gather list, peers use the pivots
        LSX_HIP(hipMemcpyAsync(buf + 2048, d_info[o], sizeof(int), hipMemcpyDeviceToDevice, x.comp));
        LSX_HIP(hipMemcpyAsync(buf + 2304, d_ipiv[o] + k, sizeof(int32_t) * jb, hipMemcpyDeviceToDevice, x.comp));
        LSX_TRY(launch_copy2d<double>(x.h, m, jb, Pn, lda[o], (double *)(buf + MG_HDR), jb));
        LSX_HIP(hipEventRecord(x.packed[p], x.comp));
        for (int e = 0; e < P; ++e) {
            if (e == o) continue;
            hipStream_t ls = x.link[e];
            LSX_HIP(hipStreamWaitEvent(ls, x.packed[p], 0));
            LSX_HIP(hipStreamWaitEvent(ls, D[e].freed[p], 0));   // the peer has finished with panel b-2 in this buffer
            for (int c = 0; c < MG_CHUNKS; ++c) {
                int r0, r1;
                chunk_rows(m, jb, c, r0, r1);
                const size_t lo = c == 0 ? 0 : MG_HDR + sizeof(double) * (size_t)r0 * jb;
                const size_t hi = MG_HDR + sizeof(double) * (size_t)r1 * jb;
                if (hi > lo) {
                    if (D[e].dev == x.dev)   // several handles on one device (rehearsal): a plain device copy
                        LSX_HIP(hipMemcpyAsync(D[e].buf[p] + lo, buf + lo, hi - lo, hipMemcpyDeviceToDevice, ls));
                    else
                        LSX_HIP(hipMemcpyPeerAsync(D[e].buf[p] + lo, D[e].dev, buf + lo, x.dev, hi - lo, ls));
                }
                LSX_HIP(hipEventRecord(D[e].landed[((size_t)o * 2 + p) * MG_CHUNKS + c], ls));
            }
        }
        return LSX_OK;
    };
    // device d: interchanges, U12 and trailing update of its local columns [c0, c1) with panel b
    auto apply_panel = [&](int d, int b, int c0, int c1, bool chunked) -> int {
        if (c1 <= c0) return LSX_OK;
        int k, jb, m;
        shape(b, k, jb, m);
        const int o = b % P, p = b & 1;
        Dev &x = D[d];
        LSX_HIP(hipSetDevice(x.dev));
        const char *buf = x.buf[p];
        const double *panel = (const double *)(buf + MG_HDR);
        double *Ac = dA[d] + c0;
        auto landed = [&](int c) { return D[d].landed[((size_t)o * 2 + p) * MG_CHUNKS + c]; };
        if (d != o) LSX_HIP(hipStreamWaitEvent(x.comp, landed(0), 0));
        // interchanges: the owner's gather list when there is one
        void *keep_moves = x.h->moves;
        const bool keep_valid = x.h->moves_valid;
        int rc = LSX_OK;
        if (cur_list) {
            x.h->moves = (void *)buf;
            x.h->moves_valid = true;
            rc = launch_laswp_moves<double>(x.h, c1 - c0, Ac, lda[d], k);
        } else {
            rc = launch_laswp<double>(x.h, c1 - c0, Ac, lda[d], k, jb, (const int32_t *)(buf + 2304));
        }
        x.h->moves = keep_moves;
        x.h->moves_valid = keep_valid;
        LSX_TRY(rc);
        double *U12 = dA[d] + (size_t)k * lda[d] + c0;
        const size_t tinv = (size_t)((jb + 63) / 64) * 4096 * sizeof(double);
        if (x.h->ws2_bytes < tinv) { set_error("getrf_mg: block-inverse workspace"); return LSX_ERR_INTERNAL; }
        LSX_TRY(launch_trtri<double>(x.h, 1, jb, panel, jb, (double *)x.h->ws2));
        LSX_TRY(launch_trsm_block<double>(x.h, 1, jb, c1 - c0, panel, jb, (const double *)x.h->ws2, U12, lda[d]));
        if (m > jb) {
            if (d == o || !chunked) {
                if (d != o) LSX_HIP(hipStreamWaitEvent(x.comp, landed(MG_CHUNKS - 1), 0));
                LSX_TRY(launch_gemm_sub<double>(x.h, m - jb, c1 - c0, jb, panel + (size_t)jb * jb, jb, U12, lda[d],
                                                dA[d] + (size_t)(k + jb) * lda[d] + c0, lda[d]));
            } else {
                for (int c = 0; c < MG_CHUNKS; ++c) {   // each row range as its chunk lands
                    int r0, r1;
                    chunk_rows(m, jb, c, r0, r1);
                    r0 = std::max(r0, jb);
                    if (r1 <= r0) continue;
                    LSX_HIP(hipStreamWaitEvent(x.comp, landed(c), 0));
                    LSX_TRY(launch_gemm_sub<double>(x.h, r1 - r0, c1 - c0, jb, panel + (size_t)r0 * jb, jb, U12, lda[d],
                                                    dA[d] + (size_t)(k + r0) * lda[d] + c0, lda[d]));
                }
            }
        }
        return LSX_OK;
    };
    auto swap_left = [&](int d, int b, int ncols) -> int {
        if (ncols <= 0) return LSX_OK;
        int k, jb, m;
        shape(b, k, jb, m);
        const int p = b & 1;
        Dev &x = D[d];
        LSX_HIP(hipSetDevice(x.dev));
        const char *buf = x.buf[p];
        void *keep_moves = x.h->moves;
        const bool keep_valid = x.h->moves_valid;
        int rc;
        if (cur_list) {
            x.h->moves = (void *)buf;
            x.h->moves_valid = true;
            rc = launch_laswp_moves<double>(x.h, ncols, dA[d], lda[d], k);
        } else {
            rc = launch_laswp<double>(x.h, ncols, dA[d], lda[d], k, jb, (const int32_t *)(buf + 2304));
        }
        x.h->moves = keep_moves;
        x.h->moves_valid = keep_valid;
        return rc;
    };
    // non-owner: pivots and info of panel b out of its buffer
    auto unpack = [&](int d, int b) -> int {
        int k, jb, m;
        shape(b, k, jb, m);
        const int p = b & 1;
        Dev &x = D[d];
        LSX_HIP(hipSetDevice(x.dev));
        LSX_HIP(hipMemcpyAsync(d_ipiv[d] + k, x.buf[p] + 2304, sizeof(int32_t) * jb, hipMemcpyDeviceToDevice, x.comp));
        LSX_HIP(hipMemcpyAsync(d_info[d], x.buf[p] + 2048, sizeof(int), hipMemcpyDeviceToDevice, x.comp));
        return LSX_OK;
    };

    // workspaces of the per-device kernels: the one computation getrf_dev uses (api.hip), under each device's guard
    for (int d = 0; d < P; ++d) {
        LSX_HIP(hipSetDevice(D[d].dev));
        if (ensure_getrf_workspace(hs[d], n, sizeof(double)) != LSX_OK) { set_error("getrf_mg: workspace allocation failed"); return LSX_ERR_ALLOC; }
        hs[d]->gemm_mfma_only = true;   // same summation order whatever the column split (as in getrf_dev)
    }
    struct MfmaOnly { std::vector<Dev> &D; ~MfmaOnly() { for (auto &d : D) d.h->gemm_mfma_only = false; } } mfma_only{D};

    bool has_list[2] = {false, false};   // whether the panel in buffer p came with a gather list
    int rc = pack_and_send(0);
    has_list[0] = D[0].h->moves_valid;
    for (int d = 1; d < P && rc == LSX_OK; ++d) {
        LSX_HIP(hipSetDevice(D[d].dev));
        LSX_HIP(hipStreamWaitEvent(D[d].comp, D[d].landed[((size_t)0 * 2 + 0) * MG_CHUNKS + 0], 0));
        rc = unpack(d, 0);
    }
    // first local block right of panel b on device d (index into blocks / offset)
    auto right_of = [&](const Dev &x, int b) { return (int)(std::upper_bound(x.blocks.begin(), x.blocks.end(), b) - x.blocks.begin()); };
    for (int b = 0; b < nblocks && rc == LSX_OK; ++b) {
        const int own = b % P;
        const bool has_next = b + 1 < nblocks;
        const int own_next = has_next ? (b + 1) % P : -1;
        cur_list = has_list[b & 1];
        // look-ahead FIRST in host order: the next owner brings block b+1 up to date, factors it and starts sending;
        // every event the other devices wait on below has then been recorded
        int done_cols = -1;   // on own_next: local columns already updated with panel b
        if (has_next) {
            Dev &x = D[own_next];
            const int li = right_of(x, b);
            const int w = std::min(nb, n - (b + 1) * nb);
            rc = apply_panel(own_next, b, x.offset[li], x.offset[li] + w, false);
            if (rc == LSX_OK) rc = pack_and_send(b + 1);
            has_list[(b + 1) & 1] = x.h->moves_valid;
            done_cols = x.offset[li] + w;
        }
        for (int d = 0; d < P && rc == LSX_OK; ++d) {
            Dev &x = D[d];
            const int li = right_of(x, b);
            int right0 = li < (int)x.blocks.size() ? x.offset[li] : x.local_cols;
            if (d == own_next) right0 = done_cols;
            rc = apply_panel(d, b, right0, x.local_cols, true);
            // interchanges on the columns left of the panel (the owner's own panel block is already in order)
            if (rc == LSX_OK) {
                const int left = (own == d) ? x.offset[local_index(x, b)]
                                            : (li < (int)x.blocks.size() ? x.offset[li] : x.local_cols);
                rc = swap_left(d, b, left);
            }
            if (rc != LSX_OK) break;
            LSX_HIP(hipSetDevice(x.dev));
            if (own == d)   // its copies of panel b read buf[b & 1]: the buffer is free only behind them
                for (int e = 0; e < P; ++e)
                    if (e != d) LSX_HIP(hipStreamWaitEvent(x.comp, D[e].landed[((size_t)d * 2 + (b & 1)) * MG_CHUNKS + MG_CHUNKS - 1], 0));
            LSX_HIP(hipEventRecord(x.freed[b & 1], x.comp));
            if (has_next && own_next != d) {
                LSX_HIP(hipStreamWaitEvent(x.comp, x.landed[((size_t)own_next * 2 + ((b + 1) & 1)) * MG_CHUNKS + 0], 0));
                rc = unpack(d, b + 1);
            }
        }
    }
    return rc;
}

int getrf_mg_f64(lsx_handle_t *hs, int P, int n, double *const *dA, const int *lda, int32_t *const *d_ipiv,
                 int *const *d_info) {
    Guard guard;
    std::vector<Dev> D(P);
    const int rc = getrf_mg_body(D, hs, P, n, dA, lda, d_ipiv, d_info);
    // everything queued (or abandoned half-way): wait for all devices, release
    return mg_free(D, rc);
}

}  // namespace lsx
