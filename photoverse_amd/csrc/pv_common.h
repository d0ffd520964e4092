// Shared device helpers for the gfx950 kernels of libphotoverse_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/photoverse_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

#define PV_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PV_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define PV_CHECK_LAUNCH() ((int)hipGetLastError())

__device__ __forceinline__ int pv_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int pv_lane_id() { return (int)(threadIdx.x & 63); }

// 16-byte async global -> LDS copy (LDS-DMA).  LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void pv_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(PV_GLOBAL_PTR(gsrc), PV_LDS_PTR(lds_wave_base), 16, 0, 0);
}

#ifdef PV_SLOW_EXP
#define PV_EXP2(x) exp2f(x)
#else
#define PV_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
__device__ __forceinline__ float pv_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float pv_quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, far below the fp16 output rounding): one rcp, one exp2,
// five FMAs - about a third of the instructions of the libm erff the GEGLU epilogue would otherwise spend per element.
__device__ __forceinline__ float pv_erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-poly * t, e, 1.0f);
    return copysignf(r, x);
}
#ifdef PV_GELU_IDENTITY   // timing experiment only (wrong results): what the erf costs the GEGLU epilogue
__device__ __forceinline__ float pv_gelu_erf(float x) { return x; }
#else
// Exact-erf GELU as it is used in the GEMM epilogues (diffusers GEGLU: hidden * F.gelu(gate), [EXT]): the GEGLU launches are bound by this
// function's VALU issue (84 M evaluations per 64x64-level launch), so it is written for instruction count:
//   gelu(x) = x * Phi(x) = max(x, 0) - |x| * Q(|x|),   Q(a) = 1 - Phi(a) = 0.5 * erfc(a / sqrt 2)
// (for x < 0: x * Phi(x) = -|x| * Q(|x|); for x > 0: x * (1 - Q) = x - |x| Q) - no sign transfer and no "1 -" on the erf side - with
// erfc by Abramowitz & Stegun 7.1.25: erfc(z) = (a1 t + a2 t^2 + a3 t^3) exp(-z^2), t = 1 / (1 + p z), |error| <= 2.5e-5, i.e. an absolute
// error <= 1.25e-5 |x| on gelu - a twentieth of the fp16 rounding of the output.  11 VALU ops, two of them transcendental (v_rcp, v_exp).
// inf / NaN contract: finite x of any size is exact in the limit (q underflows to 0: gelu(1e30) = 1e30, gelu(-1e30) = -0); x = +-inf gives NaN
// (inf * 0), where torch gives +inf / -0.  That input cannot occur: the argument is an fp32 accumulator of fp16 products, |x| <= K * 65504^2
// + |bias| ~ 2.2e13 at the largest K (5120) - thirteen orders below fp32 overflow - so no clamp (one more VALU op in a VALU-bound epilogue) is
// spent on it.  Values that overflow only at the fp16 STORE (|gelu| > 65504) become +-inf there, as in torch.
__device__ __forceinline__ float pv_gelu_erf(float x) {
    const float b = fabsf(x) * 0.84932180028801904272f;                 // z * sqrt(log2 e), z = |x| / sqrt 2: exp(-z^2) = exp2(-b^2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.39170375623211625f, b, 1.0f));   // p z = 0.47047 z = (0.47047 / sqrt(log2 e)) b
    float poly = fmaf(0.3739278f, t, -0.0479399f);                      // 0.5 * (a3, a2, a1) = 0.5 * (0.7478556, -0.0958798, 0.3480242)
    poly = fmaf(poly, t, 0.1740121f);
    const float e = __builtin_amdgcn_exp2f(-b * b);
    const float q = poly * t * e;                                        // Q(|x|)
    return fmaxf(x, 0.f) - fabsf(x) * q;
}
#endif
__device__ __forceinline__ float pv_apply_act(float x, int act) {
    switch (act) {
        case PV_ACT_SILU: return pv_silu(x);
        case PV_ACT_QUICK_GELU: return pv_quick_gelu(x);
        case PV_ACT_LEAKY_RELU: return x > 0.f ? x : 0.01f * x;
        case PV_ACT_GELU: return pv_gelu_erf(x);
        default: return x;
    }
}

__device__ __forceinline__ float pv_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float pv_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Reductions across the 4 lanes {l, l^16, l^32, l^48} that share (lane & 15), as pure VALU lane swaps
// (v_permlane16_swap / v_permlane32_swap) - no LDS round trip like ds_bpermute.
// NOTE: the two results are copied into scalars before the bit cast: __builtin_bit_cast applied directly to an
// element expression of the builtin's vector result (a[1]) reads element 0 (clang 22 / ROCm 7.2).
__device__ __forceinline__ void pv_swap16(float v, float& x, float& y) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    x = __builtin_bit_cast(float, a0);
    y = __builtin_bit_cast(float, a1);
}
__device__ __forceinline__ void pv_swap32(float v, float& x, float& y) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    x = __builtin_bit_cast(float, a0);
    y = __builtin_bit_cast(float, a1);
}
__device__ __forceinline__ float pv_quad_max(float v) {
    float x, y;
    pv_swap16(v, x, y);
    v = fmaxf(x, y);
    pv_swap32(v, x, y);
    return fmaxf(x, y);
}
__device__ __forceinline__ float pv_quad_sum(float v) {
    float x, y;
    pv_swap16(v, x, y);
    v = x + y;
    pv_swap32(v, x, y);
    return x + y;
}

// XCD-aware bijective block remap (8 XCDs, blocks dealt round-robin): consecutive remapped ids share an XCD/L2.
__device__ __forceinline__ int pv_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
