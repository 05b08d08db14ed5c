// common.h — shared device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/onedc_hip.h"

// ---- the 16-bit activation / weight format of this build ------------------------------------------------------------
// Every kernel stores activations and packed weights as raw 16-bit words (a16_t) and touches their meaning only through
// the four helpers below and ONEDC_MFMA16.  The library is built twice from the same sources:
//   libonedc_hip.so       bf16  (8-bit mantissa, f32's exponent range: nothing can overflow)
//   libonedc_hip_f16.so   IEEE half, -DONEDC_ACT_F16 (11-bit mantissa: every rounding of the path is 8x finer, which is
//                         what the 0.01 dB PSNR budget of the decode path needs; range +-65504, the
//                         host keeps streams inside it -- vae.py stream scale -- and the output is checked for non-finites)
// Same instruction rate (v_mfma_f32_16x16x32_f16 == ..._bf16), same bytes, f32 accumulation in both.
typedef unsigned short a16_t;
typedef __attribute__((ext_vector_type(8))) short a16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define ONEDC_LAUNCH_CHECK()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#ifdef ONEDC_ACT_F16
typedef __attribute__((ext_vector_type(2))) _Float16 a16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 mfma16x8_t;
#define ONEDC_ACT_NAME "f16"
#define ONEDC_A16_ONE 0x3c00u                     /* 1.0 */
__device__ __forceinline__ float a2f(a16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
#define ONEDC_MFMA16(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma16x8_t, (a)), __builtin_bit_cast(mfma16x8_t, (b)), (c), 0, 0, 0)
#else
typedef __attribute__((ext_vector_type(2))) __bf16 a16x2_t;
#define ONEDC_ACT_NAME "bf16"
#define ONEDC_A16_ONE 0x3f80u                     /* 1.0 */
__device__ __forceinline__ float a2f(a16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
#define ONEDC_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
#define ONEDC_A16_ONE2 ((ONEDC_A16_ONE << 16) | ONEDC_A16_ONE)
// the two halves of a packed pair
__device__ __forceinline__ float a2f_lo(uint32_t w) { return a2f((a16_t)(w & 0xffffu)); }
__device__ __forceinline__ float a2f_hi(uint32_t w) {
#ifdef ONEDC_ACT_F16
    return a2f((a16_t)(w >> 16));
#else
    return __uint_as_float(w & 0xffff0000u);
#endif
}

// f32 -> 16-bit, round to nearest even, NaN stays NaN (bf16: one v_cvt_pk_bf16_f32 per pair; f16: two v_cvt_f16_f32 + pack)
__device__ __forceinline__ uint32_t pack_a2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, a16x2_t));
}
// bf16 pair in either build (flash_attn64's P, and what its V^T producer stores: onedc_conv_params.out_bf16)
__device__ __forceinline__ uint32_t pack_bf16_2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_always_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_always_t));
}
__device__ __forceinline__ a16_t f2a(float f) { return (a16_t)(pack_a2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float leaky(float v, float slope) { return v >= 0.f ? v : v * slope; }
// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each; the result is stored as bf16 or feeds one)
__device__ __forceinline__ float silu_f(float v) {
#ifdef EXP_SILU_FREE                             // (ablation) what GroupNorm-on-load costs without the two transcendentals
    return v * 0.5f;
#else
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
#endif
}
// exact-erf GELU (the reference's F.gelu).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16
// rounding of the result): one v_rcp_f32, one v_exp_f32 and a degree-5 Horner chain instead of libm's ~40-instruction erff.
__device__ __forceinline__ float gelu_f(float v) {
#ifdef ONEDC_GELU_LIBM
    return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
#elif defined(EXP_GELU_FREE)                     // (ablation) what the GEGLU layers cost without the function
    return 0.5f * v;
#else
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.f));
    float q = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    q = __builtin_fmaf(q, t, 1.421413741f);
    q = __builtin_fmaf(q, t, -0.284496736f);
    q = __builtin_fmaf(q, t, 0.254829592f);
    const float e = 1.f - q * t * __builtin_amdgcn_exp2f(x * x * -1.4426950408889634f);      // erf(|v| / sqrt 2)
    return 0.5f * v * (1.f + (v < 0.f ? -e : e));
#endif
}

static inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel and launches come from several host
// threads (DecodePipeline): one bit per device ordinal in an atomic mask, set after the (idempotent) attribute calls.
#include <atomic>
#include <initializer_list>
#include <utility>
struct OnedcLdsAttrOnce {
    std::atomic<uint64_t> done[4] = {};
    void ensure(std::initializer_list<std::pair<const void *, int>> kernels) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        std::atomic<uint64_t> &word = done[(dev >> 6) & 3];
        if (word.load(std::memory_order_acquire) & bit) return;
        for (const auto &k : kernels) (void)hipFuncSetAttribute(k.first, hipFuncAttributeMaxDynamicSharedMemorySize, k.second);
        word.fetch_or(bit, std::memory_order_release);
    }
};
