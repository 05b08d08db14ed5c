// mfma_power.hip — does the sustained matrix rate depend on the operands' BITS?  (the bf16 build of the window
// kernel runs 3 - 6 % faster than the f16 build on the same layers, while on constant operands the f16 instruction is the faster one.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_power tools/micro/mfma_power.hip && tools/micro/mfma_power
// Register-only loops of v_mfma_f32_16x16x32_f16 / _bf16, 8 waves per SIMD, ~0.5 s per case so that the power management settles;
// operands: random values (full mantissa), the same values with the low mantissa bits of A, or of A and B, cleared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;

constexpr int ITER = 2048, NACC = 16;

__device__ __forceinline__ uint32_t rng(uint32_t &s) {
    s = s * 1664525u + 1013904223u;
    return s;
}

template <bool BF16>
__global__ __launch_bounds__(256) void power_kernel(float *out, int seed, unsigned mask_a, unsigned mask_b) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t st = (threadIdx.x + 1) * 2654435761u + seed * 40503u + blockIdx.x;
    s16x8 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // values in [0.5, 2): exponent fixed near 1, sign and mantissa random (f16: 0x3800 .. 0x3fff; bf16: 0x3f00 .. 0x3fff)
            const uint32_t r1 = rng(st), r2 = rng(st);
            const unsigned va = BF16 ? (0x3f00u | (r1 & 0xffu) | ((r1 >> 9) & 0x8000u)) : (0x3800u | (r1 & 0x7ffu) | ((r1 >> 9) & 0x8000u));
            const unsigned vb = BF16 ? (0x3f00u | (r2 & 0xffu) | ((r2 >> 9) & 0x8000u)) : (0x3800u | (r2 & 0x7ffu) | ((r2 >> 9) & 0x8000u));
            a[k][j] = (short)(va & mask_a);
            b[k][j] = (short)(vb & mask_b);
        }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a[i & 3]), __builtin_bit_cast(h16x8, b[(i >> 2) & 3]),
                                                                acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// round 5: the same question for the 32x32x16 shape (half the operand registers read per multiply-add: 1024 operand elements
// for 16 384 products instead of 8 192) -- does the chip hold a higher clock on it?
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <bool BF16>
__global__ __launch_bounds__(256) void power32_kernel(float *out, int seed, unsigned mask_a, unsigned mask_b) {
    constexpr int NA = 8;
    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    uint32_t st = (threadIdx.x + 1) * 2654435761u + seed * 40503u + blockIdx.x;
    s16x8 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t r1 = rng(st), r2 = rng(st);
            const unsigned va = BF16 ? (0x3f00u | (r1 & 0xffu) | ((r1 >> 9) & 0x8000u)) : (0x3800u | (r1 & 0x7ffu) | ((r1 >> 9) & 0x8000u));
            const unsigned vb = BF16 ? (0x3f00u | (r2 & 0xffu) | ((r2 >> 9) & 0x8000u)) : (0x3800u | (r2 & 0x7ffu) | ((r2 >> 9) & 0x8000u));
            a[k][j] = (short)(va & mask_a);
            b[k][j] = (short)(vb & mask_b);
        }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a[i & 3]), __builtin_bit_cast(h16x8, b[(i >> 1) & 3]),
                                                                acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <bool BF16>
static double run32(const char *name, float *out, unsigned ma, unsigned mb, int waves_per_simd = 8) {
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int reps = 250;
    for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(power32_kernel<BF16>, dim3(blocks), dim3(256), 0, 0, out, w, ma, mb);
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(power32_kernel<BF16>, dim3(blocks), dim3(256), 0, 0, out, r, ma, mb);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)reps * blocks * 4 * ITER * 8 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("%-72s %8.1f TFLOP/s\n", name, tf);
    return tf;
}

template <bool BF16>
static double run(const char *name, float *out, unsigned ma, unsigned mb) {
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int reps = 250;                               // ~0.5 s
    for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(power_kernel<BF16>, dim3(blocks), dim3(256), 0, 0, out, w, ma, mb);
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(power_kernel<BF16>, dim3(blocks), dim3(256), 0, 0, out, r, ma, mb);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)reps * blocks * 4 * ITER * NACC * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
    printf("%-72s %8.1f TFLOP/s\n", name, tf);
    return tf;
}

int main() {
    float *out;
    (void)hipMalloc(&out, 4096);
    run<false>("f16, random sign + 11-bit mantissa in both operands", out, 0xffffu, 0xffffu);
    run<false>("f16, A (weights) cut to 8 significant bits (low 3 mantissa bits zero)", out, 0xfff8u, 0xffffu);
    run<false>("f16, A and B cut to 8 significant bits", out, 0xfff8u, 0xfff8u);
    run<false>("f16, A and B mantissa zero (powers of two)", out, 0xfc00u, 0xfc00u);
    run<true>("bf16, random sign + 8-bit mantissa in both operands", out, 0xffffu, 0xffffu);
    run<true>("bf16, A and B mantissa zero (powers of two)", out, 0xff80u, 0xff80u);
    run<false>("f16, random (again)", out, 0xffffu, 0xffffu);
    run32<false>("32x32x16 f16, random sign + 11-bit mantissa in both operands", out, 0xffffu, 0xffffu);
    run32<false>("32x32x16 f16, A and B mantissa zero (powers of two)", out, 0xfc00u, 0xfc00u);
    run32<true>("32x32x16 bf16, random sign + 8-bit mantissa in both operands", out, 0xffffu, 0xffffu);
    run<false>("f16 16x16x32, random (third time)", out, 0xffffu, 0xffffu);
    run32<false>("32x32x16 f16, random (again)", out, 0xffffu, 0xffffu);
    return 0;
}
