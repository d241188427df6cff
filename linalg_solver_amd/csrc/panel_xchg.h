// Shared pieces of the one-launch panel kernels: the cross-CU exchange record formats and the
// cross-lane (DPP / readlane) reductions.  Included by kernels_panel_coop.hip,
// kernels_panel_pipe.hip and kernels_trsv.hip.
#pragma once
#include "common.h"

namespace lsx {

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int PC_COLS = 128;   // column capacity (16 thread columns x 8)
constexpr int HDR_STRIDE = 512;  // bytes between headers: one line / channel each, not 16 B apart
constexpr int SPIN_LIMIT = 1 << 20;   // ~1 s of polling before a workgroup gives up

struct __attribute__((aligned(16))) XGran {
    unsigned long long bits;  // value (fp64 bits, or fp32 bits in the low half)
    unsigned epoch;
    unsigned pad;
};

// Polling loads must be re-issued on every trip of a spin loop.  hipcc hoists a plain buffer load out
// of a loop without stores (also with the "volatile" cache-policy bit), and an
// `asm volatile("" ::: "memory")` fence keeps it in place only at the price of a full vmcnt(0) drain
// right behind the load.  An opaque zero as the scalar offset does it for free: the address looks
// different on every trip, and the waits stay exact.
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
}

// ---- DPP cross-lane moves: VALU-rate, no trip through the LDS crossbar (ds_bpermute costs ~100+
// cycles per dependent step).  Applied in the order quad xor 1, quad xor 2, row_half_mirror,
// row_mirror they leave every lane of a 16-lane row with the row's reduction (max with ties).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
#define LSX_DPP_STEP(CTRL, v, i)                                         \
    {                                                                    \
        const double ov_ = dpp_d<CTRL>(v);                               \
        const int oi_ = dpp_i<CTRL>(i);                                  \
        const bool b_ = (ov_ > v) | ((ov_ == v) & (oi_ < i));            \
        v = b_ ? ov_ : v;                                                \
        i = b_ ? oi_ : i;                                                \
    }
// arg-max (largest v, lowest i on ties) over each 16-lane row
__device__ __forceinline__ void row16_argmax(double &v, int &i) {
    LSX_DPP_STEP(0xB1, v, i)   // quad_perm [1,0,3,2]
    LSX_DPP_STEP(0x4E, v, i)   // quad_perm [2,3,0,1]
    LSX_DPP_STEP(0x141, v, i)  // row_half_mirror
    LSX_DPP_STEP(0x140, v, i)  // row_mirror
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}

__device__ __forceinline__ double readlane_t(double v, int l) { return readlane_d(v, l); }
__device__ __forceinline__ float readlane_t(float v, int l) {
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}

// ---- reductions with the DPP move fused into the VALU op (v_max_u32_dpp / v_min_i32_dpp): one
// instruction per step.  An arg-max over fp64 magnitudes is three such phases on the bit pattern
// (non-negative doubles order like unsigned 64-bit integers): high word, low word, then the lowest
// row index among the lanes that hold the maximum.
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
    v = max(v, (unsigned)dpp_i<0xB1>((int)v));
    v = max(v, (unsigned)dpp_i<0x4E>((int)v));
    v = max(v, (unsigned)dpp_i<0x141>((int)v));
    v = max(v, (unsigned)dpp_i<0x140>((int)v));
    return v;
}
__device__ __forceinline__ int row16_min_i32(int v) {
    v = min(v, dpp_i<0xB1>(v));
    v = min(v, dpp_i<0x4E>(v));
    v = min(v, dpp_i<0x141>(v));
    v = min(v, dpp_i<0x140>(v));
    return v;
}
// combine the 16-lane rows starting at lane0 (NROW of them) through scalar registers
template <int NROW>
__device__ __forceinline__ unsigned rows_max_u32(unsigned v, int lane0) {
    unsigned r = (unsigned)__builtin_amdgcn_readlane((int)v, lane0);
#pragma unroll
    for (int k = 1; k < NROW; ++k) r = max(r, (unsigned)__builtin_amdgcn_readlane((int)v, lane0 + 16 * k));
    return r;
}
template <int NROW>
__device__ __forceinline__ int rows_min_i32(int v, int lane0) {
    int r = __builtin_amdgcn_readlane(v, lane0);
#pragma unroll
    for (int k = 1; k < NROW; ++k) r = min(r, __builtin_amdgcn_readlane(v, lane0 + 16 * k));
    return r;
}
// arg-max over NROW 16-lane rows starting at lane0: key = (khi, klo) bit pattern of |a| (0 for "no
// candidate"), idx = row (INT_MAX for none).  Returns the winning row, wave-uniform; INT_MAX: none.
// All 64 lanes must be active.
template <int NROW>
__device__ __forceinline__ int argmax_rows(unsigned khi, unsigned klo, int idx, int lane0) {
    const unsigned mhi = rows_max_u32<NROW>(row16_max_u32(khi), lane0);
    const bool top = khi == mhi;
    const unsigned mlo = rows_max_u32<NROW>(row16_max_u32(top ? klo : 0u), lane0);
    return rows_min_i32<NROW>(row16_min_i32((top & (klo == mlo)) ? idx : 0x7fffffff), lane0);
}


}  // namespace lsx
