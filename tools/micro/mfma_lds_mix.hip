// mfma_lds_mix.hip -- round 6: does the clock the chip holds in an MFMA-bound loop rise when the loop reads
// fewer LDS bytes per MFMA?  The 4-wave window kernel (gemm_conv_win4.hip) reads 12 ds_read_b128 per 32 v_mfma_f32_16x16x32 of a
// wave (wave tile 64 channels x 128 pixels); a larger wave tile or B-fragment reuse across two taps would read 8 or fewer.  Before
// re-cutting that kernel: the same instruction mix without its DMA, barriers and epilogue -- two workgroups of four waves per CU
// (one wave of each per SIMD, as in the kernel), every operand register re-read from LDS by ds_read_b128 at a chosen rate R per
// 32 MFMAs, random f16 / bf16 data, ~1.5 s per case so that the power management settles; TFLOP/s from HIP events, the in-loop
// clock from s_memtime / s_memrealtime stamps of every wave (MI355X_MICROARCH.md, DVFS give-back item 6; the stamps go to a buffer
// nothing else reads).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_lds_mix tools/micro/mfma_lds_mix.hip && tools/micro/mfma_lds_mix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int TILES = 1024;                  // "K-tiles" of 32 MFMAs per launch and wave
constexpr int LDS_BYTES = 64 * 1024;         // per workgroup: two fit a CU

template <int R, bool BF16>
__global__ __launch_bounds__(256, 2) void mix_kernel(const uint32_t *seed_data, unsigned long long *stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // random operand data: values in [0.5, 2) with random sign and mantissa (as tools/micro/mfma_power.hip)
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += 256) {
        uint32_t s = seed_data[(blockIdx.x * 131 + i) & 0xffff];
        const uint32_t lo = BF16 ? (0x3f00u | (s & 0xffu) | ((s >> 9) & 0x8000u)) : (0x3800u | (s & 0x7ffu) | ((s >> 9) & 0x8000u));
        s = s * 1664525u + 1013904223u;
        const uint32_t hi = BF16 ? (0x3f00u | (s & 0xffu) | ((s >> 9) & 0x8000u)) : (0x3800u | (s & 0x7ffu) | ((s >> 9) & 0x8000u));
        ((uint32_t *)smem)[i] = lo | (hi << 16);
    }
    __syncthreads();
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(acc[i]));
    }
    s16x8 fa[2][4], fb[2][8];                    // double-buffered like the kernel's fragments: tile t multiplies buffer t & 1 while the reads
                                                 // of tile t + 1 land in the other one
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = wave * 4096 + lane * 16;          // conflict-free: granule == lane inside 1 KB blocks
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[b][i]) : "v"(base), "n"(b * 2048 + i * 1024));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[b][i]) : "v"(base), "n"(16384 + b * 1024 + i * 1024));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto tile = [&](int t, auto CUR) __attribute__((always_inline)) {
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        const int off = base + ((t & 3) << 13);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((r % 12) < 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fa[nxt][r % 12]) : "v"(off), "n"((r % 8) * 1024));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fb[nxt][(r % 12) - 4]) : "v"(off), "n"((r % 8) * 1024));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i * 8 + j]) : "v"(fa[cur][i]), "v"(fb[cur][j]));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i * 8 + j]) : "v"(fa[cur][i]), "v"(fb[cur][j]));
            }
        if (R) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[nxt][0]), "+v"(fa[nxt][1]), "+v"(fa[nxt][2]), "+v"(fa[nxt][3]), "+v"(fb[nxt][0]),
                             "+v"(fb[nxt][1]), "+v"(fb[nxt][2]), "+v"(fb[nxt][3]), "+v"(fb[nxt][4]), "+v"(fb[nxt][5]), "+v"(fb[nxt][6]),
                             "+v"(fb[nxt][7]));
    };
    for (int t = 0; t < TILES; t += 2) {
        tile(t, std::integral_constant<int, 0>());
        tile(t + 1, std::integral_constant<int, 1>());
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (lane == 0) {
        unsigned long long *d = stamps + ((size_t)blockIdx.x * 4 + wave) * 2;
        d[0] = t1 - t0;
        d[1] = (r1 - r0) | ((s == 123.456f) ? 1ull << 63 : 0ull);
    }
}

template <int R, bool BF16>
static void run(const uint32_t *seed, unsigned long long *stamps_dev, int blocks) {
    (void)hipFuncSetAttribute((const void *)mix_kernel<R, BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL((mix_kernel<R, BF16>), dim3(blocks), dim3(256), LDS_BYTES, 0, seed, stamps_dev); };
    for (int w = 0; w < 40; ++w) launch();
    (void)hipDeviceSynchronize();
    // ~1.5 s of back-to-back launches, timed
    float ms = 0.f;
    int reps = 0;
    (void)hipEventRecord(e0);
    for (int chunk = 0; chunk < 30; ++chunk) {
        for (int r = 0; r < 20; ++r) launch();
        reps += 20;
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st((size_t)blocks * 8);
    (void)hipMemcpy(st.data(), stamps_dev, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (size_t i = 0; i < st.size(); i += 2) ghz.push_back((double)st[i] / (double)(st[i + 1] & ~(1ull << 63)) * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double tf = (double)reps * blocks * 4 * TILES * 32 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
    printf("%-5s  %2d ds_read_b128 per 32 MFMAs (%4.0f B of LDS per MFMA)   %8.1f TFLOP/s   clock median %.3f GHz (p10 %.3f, p90 %.3f)   "
           "%.1f cycles per MFMA and SIMD\n",
           BF16 ? "bf16" : "f16", R, R * 1024.0 / 32, tf, ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10],
           ghz[ghz.size() / 2] * 1e9 / (tf * 1e12 / (2.0 * 16 * 16 * 32) / 1024.0));
}

int main() {
    const int blocks = 512 * 4;                // two resident workgroups per CU, four rounds
    uint32_t *seed;
    unsigned long long *stamps;
    (void)hipMalloc(&seed, 65536 * 4);
    (void)hipMalloc(&stamps, (size_t)blocks * 8 * 8);
    std::vector<uint32_t> h(65536);
    uint32_t s = 12345u;
    for (auto &v : h) v = (s = s * 1664525u + 1013904223u);
    (void)hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("two 4-wave workgroups per CU (one wave of each per SIMD), 1024 K-tiles of 32 v_mfma_f32_16x16x32 per wave and launch, random "
           "operands re-read from LDS\n");
    run<0, false>(seed, stamps, blocks);
    run<4, false>(seed, stamps, blocks);
    run<8, false>(seed, stamps, blocks);
    run<12, false>(seed, stamps, blocks);
    run<16, false>(seed, stamps, blocks);
    run<24, false>(seed, stamps, blocks);
    run<0, true>(seed, stamps, blocks);
    run<4, true>(seed, stamps, blocks);
    run<8, true>(seed, stamps, blocks);
    run<12, true>(seed, stamps, blocks);
    run<16, true>(seed, stamps, blocks);
    run<24, true>(seed, stamps, blocks);
    run<12, false>(seed, stamps, blocks);      // (again: drift of the box over the run)
    return 0;
}
