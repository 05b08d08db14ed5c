// issue_mix.hip — what the vector pipe and the matrix pipe of ONE SIMD cost each other, in shader cycles per instruction.
// The question behind it (flash_attn64): a wave issues 64 v_exp_f32, 32 packs and 56 MFMAs per tile and
// shares its SIMD with one other wave -- which of those streams add, which hide behind each other?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue_mix tools/micro/issue_mix.hip && tools/micro/issue_mix
// Every wave runs ITER iterations of one instruction mix on registers only and reads s_memtime around the loop; the host
// prints shader cycles per iteration and wave for one, two and three waves per SIMD (256, 512 or 768 workgroups of 4 waves on
// 256 CUs; their LDS request, 128 / 64 / 40 KB, decides how many share a CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 b16x8;

constexpr int ITER = 4096;

// MODE: 0  16 v_exp_f32            1  16 v_fma_f32            2  8 MFMA 16x16x32         3  4 MFMA 32x32x16
//       4  8 MFMA16 + 16 v_exp     5  8 MFMA16 + 16 v_fma     6  4 MFMA32 + 16 v_exp     7  4 MFMA32 + 16 v_fma
//       8  16 v_exp + 8 v_cvt_pk_bf16_f32 + 14 MFMA16 (flash_attn64's mix per 16 scores of a lane)
//       9  16 v_permlane16_swap    10 16 v_exp_f16             11 8 v_pk_fma_f32 (16 values)
template <int MODE>
__global__ __launch_bounds__(256) void mix_kernel(unsigned long long *cycles, float *out, int seed) {
    extern __shared__ char smem[];
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = -0.001f * (float)((threadIdx.x + seed + i) & 63);
    f32x4 a4[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 a16[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) a16[i][e] = 0.f;
    b16x8 fa, fb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        fa[j] = (__bf16)(1.f + 0.125f * (float)((threadIdx.x + j) & 3));
        fb[j] = (__bf16)(0.5f + 0.25f * (float)((threadIdx.x + seed + j) & 1));
    }
    unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr bool EXP = MODE == 0 || MODE == 4 || MODE == 6 || MODE == 8;
    constexpr bool FMA = MODE == 1 || MODE == 5 || MODE == 7;
    constexpr int N16 = (MODE == 2 || MODE == 4 || MODE == 5) ? 8 : MODE == 8 ? 14 : 0;
    constexpr int N32 = (MODE == 3 || MODE == 6 || MODE == 7) ? 4 : 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        // program order: one matrix instruction, then its share of the vector instructions
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (N16 && (i * N16) / 16 != ((i + 1) * N16) / 16) {
#pragma unroll
                for (int m = (i * N16) / 16; m < ((i + 1) * N16) / 16; ++m)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[m]) : "v"(fa), "v"(fb));
            }
            if (N32 && (i & 3) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a16[i >> 2]) : "v"(fa), "v"(fb));
            if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            if (FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
            if (MODE == 8 && (i & 1)) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i >> 1]) : "v"(x[i - 1]), "v"(x[i]));
            if (MODE == 9) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i ^ 1]));
            if (MODE == 10) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
            if (MODE == 11 && (i & 1))
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double *)&x[i - 1]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 14; ++i) s += a4[i][0] + a4[i][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += a16[i][0] + a16[i][15];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)pk[i];
    if (s == 123.456f) out[threadIdx.x] = s + smem[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char *name, unsigned long long *cyc, float *out) {
    hipFuncSetAttribute((const void *)mix_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const int lds[3] = {128 * 1024, 64 * 1024, 40 * 1024};      // 1, 2, 3 workgroups (= waves per SIMD) fit a CU's 160 KB
    double res[3], us[3];
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int n = 0; n < 3; ++n) {
        const int blocks = 256 * (n + 1);
        hipLaunchKernelGGL(mix_kernel<MODE>, dim3(blocks), dim3(256), lds[n], 0, cyc, out, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(mix_kernel<MODE>, dim3(blocks), dim3(256), lds[n], 0, cyc, out, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        us[n] = ms * 1e3;
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : h) sum += (double)v;
        res[n] = sum / h.size() / ITER;
    }
    // cycles: s_memtime ticks per iteration and wave; us: the whole launch (all workgroups resident at once)
    printf("%-58s 1 wave/SIMD %7.1f cyc %7.0f us | 2: %7.1f cyc %7.0f us | 3: %7.1f cyc %7.0f us\n", name, res[0], us[0], res[1], us[1],
           res[2], us[2]);
}

int main() {
    unsigned long long *cyc;
    float *out;
    hipMalloc(&cyc, 8 * 4 * 768);
    hipMalloc(&out, 4096);
    run<0>("16 v_exp_f32", cyc, out);
    run<10>("16 v_exp_f16", cyc, out);
    run<1>("16 v_fma_f32", cyc, out);
    run<11>("8 v_pk_fma_f32 (16 values)", cyc, out);
    run<9>("16 v_permlane16_swap_b32", cyc, out);
    run<2>("8 v_mfma_f32_16x16x32_bf16", cyc, out);
    run<3>("4 v_mfma_f32_32x32x16_bf16 (the same flops)", cyc, out);
    run<4>("8 MFMA 16x16x32 + 16 v_exp_f32", cyc, out);
    run<5>("8 MFMA 16x16x32 + 16 v_fma_f32", cyc, out);
    run<6>("4 MFMA 32x32x16 + 16 v_exp_f32", cyc, out);
    run<7>("4 MFMA 32x32x16 + 16 v_fma_f32", cyc, out);
    run<8>("flash_attn64's mix: 14 MFMA 16x16x32 + 16 v_exp + 8 packs", cyc, out);
    return 0;
}
