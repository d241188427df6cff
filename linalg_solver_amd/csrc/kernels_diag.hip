// Diagnostics: raw MFMA issue-rate microbenchmark.  Operands stay in registers,
// one workgroup of 4 waves per CU slot, 4 independent accumulators per wave; the
// measured rate is what the trailing-update kernel is priced against next to
// the datasheet peak (bench.py / DESIGN.md).
#include "common.h"

namespace lsx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_peak_f64_kernel(int iters, double *out) {
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 12345.678) out[0] = s;  // keep the chain alive without a store in practice
}

__global__ __launch_bounds__(256) void mfma_peak_f32_kernel(int iters, double *out) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 12345.678f) out[0] = s;
}

int diag_mfma_peak(lsx_handle_t h, int is_f32, int iters, int blocks_per_cu, double *tflops) {
    const int grid = h->num_cu * blocks_per_cu;
    hipEvent_t e0, e1;
    LSX_HIP(hipEventCreate(&e0));
    LSX_HIP(hipEventCreate(&e1));
    double *out = (double *)h->scratch;
    for (int rep = 0; rep < 2; ++rep) {  // first pass warms up clocks and code
        LSX_HIP(hipEventRecord(e0, h->stream));
        if (is_f32)
            hipLaunchKernelGGL(mfma_peak_f32_kernel, dim3(grid), dim3(256), 0, h->stream, iters, out);
        else
            hipLaunchKernelGGL(mfma_peak_f64_kernel, dim3(grid), dim3(256), 0, h->stream, iters, out);
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
    return LSX_OK;
}

}  // namespace lsx
