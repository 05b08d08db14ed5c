// common.h — shared device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/onedc_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define ONEDC_LAUNCH_CHECK()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// f32 -> bf16, round to nearest even, NaN stays NaN: one v_cvt_pk_bf16_f32 per pair
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float leaky(float v, float slope) { return v >= 0.f ? v : v * slope; }
// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each; the result is stored as bf16 or feeds one)
__device__ __forceinline__ float silu_f(float v) {
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
// exact-erf GELU (the reference's F.gelu).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16
// rounding of the result): one v_rcp_f32, one v_exp_f32 and a degree-5 Horner chain instead of libm's ~40-instruction erff.
__device__ __forceinline__ float gelu_f(float v) {
#ifdef ONEDC_GELU_LIBM
    return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
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
