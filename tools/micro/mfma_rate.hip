// mfma_rate.hip — what the matrix cores of THIS chip deliver per format, registers only (no LDS, no memory): the ceiling
// an MXFP8 conv could reach against the 16-bit kernels.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_rate tools/micro/mfma_rate.hip && tools/micro/mfma_rate
// Every wave issues ITER x 16 independent MFMAs (16 accumulator tiles, so no instruction waits for the previous one's
// result); 8 waves per SIMD hide the issue gaps.  Prints TFLOP/s at the clock the chip holds under that load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int ITER = 2048, NACC = 16;

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int seed) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int t = threadIdx.x + seed;
    s16x8 a16 = {(short)(0x3c00 + (t & 7)), 0x3c00, 0x3800, 0x3c00, 0x3400, 0x3c00, 0x3c00, 0x3800};
    s16x8 b16 = {0x3c00, (short)(0x3800 + (t & 3)), 0x3c00, 0x3c00, 0x3800, 0x3400, 0x3c00, 0x3c00};
    i32x8 a8, b8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a8[j] = 0x38383838 + (t & 1);      // e4m3 1.0 = 0x38
        b8[j] = 0x38303838 + ((t >> 1) & 1);
    }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16, b16, acc[i], 0, 0, 0);
            if (MODE == 1)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a16), __builtin_bit_cast(h16x8, b16),
                                                                acc[i], 0, 0, 0);
            if (MODE == 2) {
                const long la = ((long)a8[1] << 32) | (unsigned)a8[0], lb = ((long)b8[1] << 32) | (unsigned)b8[0];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, acc[i], 0, 0, 0);
            }
            if (MODE == 4)   // the k = 16 forms of CDNA1-3 (half the K of the gfx950 forms): do they take half the cycles?
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, s16x4{a16[0], a16[1], a16[2], a16[3]}),
                                                               __builtin_bit_cast(h16x4, s16x4{b16[0], b16[1], b16[2], b16[3]}),
                                                               acc[i], 0, 0, 0);
            if (MODE == 5)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(s16x4{a16[0], a16[1], a16[2], a16[3]},
                                                                   s16x4{b16[0], b16[1], b16[2], b16[3]}, acc[i], 0, 0, 0);
            if (MODE == 3)   // block-scaled, both operands e4m3 (cbsz = blgp = 0), scales 2^0 (E8M0 127)
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;          // keep the loop alive
}

template <int MODE>
static double run(const char *name, double flops_per_mfma, float *out) {
    const int blocks = 256 * 8;                       // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, w);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)reps * blocks * 4 * ITER * NACC * flops_per_mfma;
    const double tf = fl / (ms * 1e-3) / 1e12;
    printf("%-44s %8.1f TFLOP/s   (%.3f ms per launch)\n", name, tf, ms / reps);
    return tf;
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
    const double b = run<0>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, out);
    const double h = run<1>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, out);
    const double f = run<2>("v_mfma_f32_16x16x32_fp8_fp8 (not scaled)", 2.0 * 16 * 16 * 32, out);
    const double m = run<3>("v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)", 2.0 * 16 * 16 * 128, out);
    const double l = run<4>("v_mfma_f32_16x16x16_f16 (k = 16 form)", 2.0 * 16 * 16 * 16, out);
    const double lb = run<5>("v_mfma_f32_16x16x16_bf16 (k = 16 form)", 2.0 * 16 * 16 * 16, out);
    printf("ratios to bf16: f16 %.3f, fp8 %.3f, MXFP8 %.3f, f16 k=16 %.3f, bf16 k=16 %.3f\n", h / b, f / b, m / b, l / b, lb / b);
    return 0;
}
