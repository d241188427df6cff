// Diagnostics: raw MFMA issue-rate microbenchmark.  Operands stay in registers,
// one workgroup of 4 waves per CU slot, 4 independent accumulators per wave; the
// measured rate is what the trailing-update kernel is priced against next to
// the datasheet peak (bench.py / DESIGN.md).
#include <algorithm>

#include "common.h"

namespace lsx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// stamps[block] = {shader-clock ticks, 100 MHz wall ticks} around the loop (wave 0 only): the
// in-kernel clock is ticks_shader / ticks_wall * 100 MHz (MICROARCH guide, DVFS give-back item 6).
__global__ __launch_bounds__(256) void mfma_peak_f64_kernel(int iters, double *out,
                                                            unsigned long long *stamps) {
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    // inline asm keeps the four accumulators in place: with the builtin hipcc copied all 32
    // registers VGPR<->AGPR around every 4 MFMAs and the loop measured the copies, not the pipe
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n\t"
                         "v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\t"
                         "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n\t"
                         "v_mfma_f64_16x16x4_f64 %3, %4, %5, %3"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                         : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    if (stamps && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = w1 - w0;
    }
    if (s == 12345.678) out[0] = s;  // keep the chain alive without a store in practice
}

__global__ __launch_bounds__(256) void mfma_peak_f32_kernel(int iters, double *out) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\t"
                         "v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n\t"
                         "v_mfma_f32_16x16x4_f32 %3, %4, %5, %3"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                         : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 12345.678f) out[0] = s;
}

int diag_mfma_peak(lsx_handle_t h, int is_f32, int iters, int blocks_per_cu, double *tflops, double *clock_mhz) {
    const int grid = h->num_cu * blocks_per_cu;
    unsigned long long *stamps = (unsigned long long *)((char *)h->scratch + 4096);
    hipEvent_t e0, e1;
    LSX_HIP(hipEventCreate(&e0));
    LSX_HIP(hipEventCreate(&e1));
    double *out = (double *)h->scratch;
    for (int rep = 0; rep < 2; ++rep) {  // first pass warms up clocks and code
        LSX_HIP(hipEventRecord(e0, h->stream));
        if (is_f32)
            hipLaunchKernelGGL(mfma_peak_f32_kernel, dim3(grid), dim3(256), 0, h->stream, iters, out);
        else
            hipLaunchKernelGGL(mfma_peak_f64_kernel, dim3(grid), dim3(256), 0, h->stream, iters, out, is_f32 ? nullptr : stamps);
        LSX_HIP(hipEventRecord(e1, h->stream));
        LSX_HIP(hipEventSynchronize(e1));
    }
    float ms = 0;
    LSX_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // per MFMA 16x16x4: 2*16*16*4 flops; per wave iters*4 MFMAs; 4 waves per block
    const double flops = 2.0 * 16 * 16 * 4 * 4.0 * iters * 4.0 * grid;
    *tflops = flops / (ms * 1e-3) / 1e12;
    if (clock_mhz) {
        *clock_mhz = 0;
        if (!is_f32) {
            std::vector<unsigned long long> hs(2 * (size_t)grid);
            LSX_HIP(hipMemcpy(hs.data(), stamps, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<double> mhz;
            for (int b = 0; b < grid; ++b)
                if (hs[2 * b + 1] > 0) mhz.push_back((double)hs[2 * b] / (double)hs[2 * b + 1] * 100.0);
            if (!mhz.empty()) {
                std::sort(mhz.begin(), mhz.end());
                *clock_mhz = mhz[mhz.size() / 2];
            }
        }
    }
    return LSX_OK;
}

}  // namespace lsx

// ---------------------------------------------------------------------------------------------
// Cross-CU exchange probe: what one "everybody publishes a 16-byte record, everybody reads all of
// them" round costs, by store policy and placement.  This is the primitive under the panel
// factorisation's per-column pivot exchange (kernels_panel_pipe.hip), measured on its own so the
// floor of the column chain is a number.
//   participants: blocks with blockIdx.x % stride == 0 (stride 8 = one XCD under round-robin dispatch)
//   mode 0: all-gather of headers only        (lane q of wave 0 polls participant q's header)
//   mode 1: header all-gather, then the 128-granule row of a rotating "winner" (two dependent reads)
//   mode 2: ping-pong between participants 0 and 1 (one-way latency = time / 2 / epochs)
// wt = 1: write-through (`sc1`) stores, the device-scope form; wt = 0: plain stores that stay in the
// XCD's L2 (valid only when every participant sits on one XCD).  Loads are always `sc1` (bypass L1).
// out[g] = {100 MHz ticks, xcc id, failures}.
#include "panel_xchg.h"

namespace lsx {

template <bool WT>
__device__ __forceinline__ void probe_store(__amdgpu_buffer_rsrc_t r, u4 v, int off) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, WT ? 16 : 0);
}

template <bool WT>
__global__ __launch_bounds__(256) void xchg_probe_kernel(int mode, int stride, int epochs, char *area,
                                                         unsigned long long *out) {
    if (blockIdx.x % stride) return;
    const int G = (gridDim.x + stride - 1) / stride, g = blockIdx.x / stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    __amdgpu_buffer_rsrc_t r_hdr = __builtin_amdgcn_make_buffer_rsrc(area, 0, G * HDR_STRIDE, 0x00020000);
    char *rows = area + (size_t)G * HDR_STRIDE;
    __amdgpu_buffer_rsrc_t r_row = __builtin_amdgcn_make_buffer_rsrc(rows, 0, G * 128 * 16, 0x00020000);
    int fails = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 2) {
        if (g < 2 && wave == 0) {
            for (int e = 1; e <= epochs; ++e) {
                // g = 0 sends, g = 1 answers
                if (g == 0) {
                    if (lane == 0) probe_store<WT>(r_hdr, u4{0u, 0u, 0u, (unsigned)e}, 0);
                    int spins = 0;
                    for (;;) {
                        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r_hdr, HDR_STRIDE, opaque_zero(), 16);
                        if (__builtin_amdgcn_readfirstlane((int)v.w) == e) break;
                        if (++spins > SPIN_LIMIT) { ++fails; break; }
                    }
                } else {
                    int spins = 0;
                    for (;;) {
                        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r_hdr, 0, opaque_zero(), 16);
                        if (__builtin_amdgcn_readfirstlane((int)v.w) == e) break;
                        if (++spins > SPIN_LIMIT) { ++fails; break; }
                    }
                    if (lane == 0) probe_store<WT>(r_hdr, u4{0u, 0u, 0u, (unsigned)e}, HDR_STRIDE);
                }
                if (fails) break;
            }
        }
    } else {
        for (int e = 1; e <= epochs && !fails; ++e) {
            if (wave == 0) {
                if (lane == 0) probe_store<WT>(r_hdr, u4{(unsigned)g, 0u, 0u, (unsigned)e}, g * HDR_STRIDE);
            }
            if (mode == 1 && wave < 2) {   // every participant publishes its candidate row: 128 granules
                const int c = tid;         // waves 0 and 1: one granule per lane
                probe_store<WT>(r_row, u4{(unsigned)c, (unsigned)g, (unsigned)e, 0u}, (g * 128 + c) * 16);
            }
            if (wave == 0) {
                int spins = 0;
                bool pend = lane < G;
                while (__any(pend)) {
                    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r_hdr, (lane < G ? lane : 0) * HDR_STRIDE,
                                                                       opaque_zero(), 16);
                    // headers of epoch e or later count (a fast participant may already be one ahead)
                    if ((int)v.w >= e) pend = false;
                    if (++spins > SPIN_LIMIT) { ++fails; break; }
                }
                if (mode == 1 && !fails) {
                    const int win = (e * 7) % G;
                    int spins2 = 0;
                    for (;;) {
                        const int oz = opaque_zero();
                        const u4 v0 = __builtin_amdgcn_raw_buffer_load_b128(r_row, (win * 128 + lane) * 16, oz, 16);
                        const u4 v1 = __builtin_amdgcn_raw_buffer_load_b128(r_row, (win * 128 + lane + 64) * 16, oz, 16);
                        if (!__any(((int)v0.z < e) | ((int)v1.z < e))) {
                            if (v0.x != (unsigned)lane || v1.x != (unsigned)(lane + 64) || v0.y != (unsigned)win) ++fails;
                            break;
                        }
                        if (++spins2 > SPIN_LIMIT) { ++fails; break; }
                    }
                }
            }
            // the other waves wait for wave 0 as the panel's bulk waves do (one barrier per column)
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        out[3 * g] = t1 - t0;
        out[3 * g + 1] = xcc;
        out[3 * g + 2] = (unsigned long long)fails;
    }
}

// us_per_epoch: mean over participants; xcc_ids[G] (may be null); failures through *nfail
int diag_xchg_probe(lsx_handle_t h, int mode, int G, int stride, int wt, int epochs, double *us_per_epoch,
                    int *xcc_ids, int *nfail) {
    const size_t body = ((size_t)G * HDR_STRIDE + (size_t)G * 128 * 16 + 255) & ~(size_t)255;
    const size_t need = body + (size_t)G * 24 + 256;
    if (need > h->scratch_bytes) { set_error("xchg_probe: scratch too small"); return LSX_ERR_INTERNAL; }
    char *area = (char *)h->scratch;
    unsigned long long *out = (unsigned long long *)(area + body);
    std::vector<unsigned long long> ho(3 * (size_t)G);
    for (int rep = 0; rep < 2; ++rep) {
        LSX_HIP(hipMemsetAsync(area, 0, need, h->stream));
        const int grid = (G - 1) * stride + 1;
        if (wt) hipLaunchKernelGGL(xchg_probe_kernel<true>, dim3(grid), dim3(256), 0, h->stream, mode, stride, epochs, area, out);
        else hipLaunchKernelGGL(xchg_probe_kernel<false>, dim3(grid), dim3(256), 0, h->stream, mode, stride, epochs, area, out);
        LSX_HIP(hipGetLastError());
        LSX_HIP(hipStreamSynchronize(h->stream));
    }
    LSX_HIP(hipMemcpy(ho.data(), out, ho.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double s = 0;
    int f = 0;
    const int cnt = mode == 2 ? 2 : G;
    for (int g = 0; g < cnt; ++g) s += (double)ho[3 * g] / 100.0 / epochs;
    for (int g = 0; g < G; ++g) {
        if (xcc_ids) xcc_ids[g] = (int)ho[3 * g + 1];
        f += (int)ho[3 * g + 2];
    }
    *us_per_epoch = s / cnt;
    *nfail = f;
    return LSX_OK;
}

}  // namespace lsx

// ---------------------------------------------------------------------------------------------
// Where do the workgroups of a stream created with hipExtStreamCreateWithCUMask land?  The look-ahead driver
// wants one stream whose kernels never need the XCD that the panel factorisation occupies; which mask bits
// belong to which XCD is not documented, so it is measured: every workgroup records its XCC id and HW_ID.
namespace lsx {

__global__ __launch_bounds__(256) void where_kernel(unsigned *out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);   // keep the workgroup resident so the grid spreads
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

// mask_words: the CU mask (ncu bits); out: per block {xcc, hw_id}
int diag_cu_mask_probe(lsx_handle_t h, const uint32_t *mask_words, int nwords, int nblocks, unsigned *out_host) {
    if ((size_t)nblocks * 8 > h->scratch_bytes) { set_error("cu_mask_probe: scratch too small"); return LSX_ERR_INTERNAL; }
    hipStream_t st = nullptr;
    if (nwords > 0) LSX_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask_words));
    else st = h->stream;
    LSX_HIP(hipStreamSynchronize(h->stream));
    hipLaunchKernelGGL(where_kernel, dim3(nblocks), dim3(256), 0, st, (unsigned *)h->scratch, 20);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(out_host, h->scratch, (size_t)nblocks * 8, hipMemcpyDeviceToHost);
    if (nwords > 0) (void)hipStreamDestroy(st);
    if (e != hipSuccess) { set_error("cu_mask_probe: %s", hipGetErrorString(e)); return LSX_ERR_HIP; }
    return LSX_OK;
}

}  // namespace lsx

// ---------------------------------------------------------------------------------------------
// A filler that holds `wgs` CUs of one XCD for `ms` milliseconds (one 1024-thread workgroup with 100 KB of LDS per
// CU; workgroups dealt to other XCDs leave at once).  Asynchronous, on the handle's side stream.  For the residency
// tests: the cooperative panel's workgroups then cannot all be resident and its bounded spins must end in the
// per-column fallback, not in a hang or in wrong factors (tests/test_gpu_parity.py).
namespace lsx {

__global__ __launch_bounds__(1024) void occupy_kernel(int xcc_want, unsigned long long ticks) {
    __shared__ char pad[100 * 1024];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if ((int)xcc != xcc_want) return;
    if (threadIdx.x == 0) pad[0] = 1;   // keep the allocation
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
    if (threadIdx.x == 2000) pad[1] = pad[0];
}

int diag_occupy(lsx_handle_t h, int xcc, int wgs, int ms) {
    // on the handle's look-ahead side stream: a stream created here may be mapped to the hardware queue of the
    // handle's main stream (the runtime multiplexes streams over a few queues), and kernels of one queue run in order
    hipLaunchKernelGGL(occupy_kernel, dim3(8 * wgs), dim3(1024), 0, h->side_stream, xcc, (unsigned long long)ms * 100000ull);
    LSX_HIP(hipGetLastError());
    return LSX_OK;
}

}  // namespace lsx
