// conv_few_out.hip — 3x3 convolution (stride 1, pad 1) with at most 4 output channels: the VAE's conv_out
// (128 -> 3 at full picture resolution; reference: AutoencoderKL decoder.conv_out behind conv_norm_out + SiLU,
// src/models/sd15_onedc_codec_stage1/model_sd15_with_codec_stage1.py:184-188).
//
// With 3 output channels there is no matrix to speak of: 3456 multiply-adds per pixel against 256 input bytes.  The
// MFMA kernels pad the 3 channels to a 16-row tile and stream the pixel operand once per tap (nine L2 -> LDS passes over
// a 2.4 GB tensor: 1.4-2.4 ms).  Here a workgroup owns a 16 x 16-pixel patch, stages its 18 x 18 window once per
// 32-channel chunk in LDS (80-byte rows: conflict-free for 16 consecutive pixels), every thread owns one pixel and
// accumulates with v_dot2c_f32_bf16 (two multiply-adds per lane and instruction, f32 accumulation) against weights that
// are uniform over the workgroup and therefore come through scalar loads.  The GroupNorm (+SiLU) in front of the layer
// can ride on the window staging (gn_ab, same operations as onedc_scale_shift_act), so the normalised tensor is neither
// written nor re-read: the layer reads its input exactly once from HBM.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int FO_ROW = 80;                   // bytes per window pixel in LDS: 64 (32 channels) + 16 padding
constexpr int FO_WIN = 324 * FO_ROW;         // 25920 B


__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
#ifdef ONEDC_ACT_F16
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(a16x2_t, a), __builtin_bit_cast(a16x2_t, b), c, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(a16x2_t, a), __builtin_bit_cast(a16x2_t, b), c, false);
#endif
}

template <int COUT>
__global__ __launch_bounds__(256) void conv_few_out_kernel(const a16_t *__restrict__ x, const uint32_t *__restrict__ w,
                                                           const float *__restrict__ bias, const float2 *__restrict__ gn_ab,
                                                           void *__restrict__ y, int H, int W, int Cin, int x_cs, int x_co,
                                                           int y_cs, int y_co, int y_f32, int gn_silu, int Ktot) {
    __shared__ __attribute__((aligned(16))) char win[2][FO_WIN];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tiles_x = W >> 4;
    const int img = blockIdx.y;
    const int y0 = (blockIdx.x / tiles_x) * 16, x0 = (blockIdx.x % tiles_x) * 16;
    const int nc = Cin >> 5;                        // 32-channel chunks
    const a16_t *xi = x + (int64_t)img * H * W * x_cs + x_co;

    // window staging: unit u = tid + 256 i covers 16 bytes (8 channels) q = u & 3 = tid & 3 of window row u >> 2
    const int q = tid & 3;
    int goff[6];                                    // element offset of the unit's pixel, or -1 (halo / padding)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = (tid + 256 * i) >> 2;
        const int wy = row / 18, wx = row - wy * 18;
        const int iy = y0 - 1 + wy, ix = x0 - 1 + wx;
        goff[i] = (row < 324 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? (iy * W + ix) * x_cs + q * 8 : -1;
    }
    auto fetch = [&](int c, uint4 (&r)[6]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] = goff[i] >= 0 ? *(const uint4 *)(xi + goff[i] + c * 32) : make_uint4(0u, 0u, 0u, 0u);
    };
    auto store = [&](int c, const uint4 (&r)[6], int buf) {
        float a[8], b[8];
        if (gn_ab) {
            const float4 *ab = (const float4 *)(gn_ab + (int64_t)img * Cin + c * 32 + q * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 v = ab[e];
                a[2 * e] = v.x;
                b[2 * e] = v.y;
                a[2 * e + 1] = v.z;
                b[2 * e + 1] = v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int row = (tid + 256 * i) >> 2;
            if (row >= 324) continue;
            uint32_t o[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
            if (gn_ab) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float lo = __builtin_fmaf(a2f((a16_t)(o[e] & 0xffff)), a[2 * e], b[2 * e]);
                    float hi = __builtin_fmaf(a2f((a16_t)(o[e] >> 16)), a[2 * e + 1], b[2 * e + 1]);
                    if (gn_silu) {
                        lo = silu_f(lo);
                        hi = silu_f(hi);
                    }
                    o[e] = goff[i] >= 0 ? pack_a2(lo, hi) : 0u;   // the conv pads the NORMALISED tensor with zeros
                }
            }
            *(uint4 *)(win[buf] + row * FO_ROW + q * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    };

    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bias ? bias[o] : 0.f;

    uint4 r[6];
    fetch(0, r);
    store(0, r, 0);
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
        if (c + 1 < nc) fetch(c + 1, r);            // in flight during this chunk's arithmetic
        const char *wb = win[c & 1];
        // (not unrolled: with all nine taps in flight the compiler keeps 432 weight words in SGPRs, spills them into
        // VGPR lanes and pays a v_readlane per multiply-add)
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const uint4 *px = (const uint4 *)(wb + ((ty + ky) * 18 + tx + kx) * FO_ROW);
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 t = px[j];
                v[4 * j] = t.x;
                v[4 * j + 1] = t.y;
                v[4 * j + 2] = t.z;
                v[4 * j + 3] = t.w;
            }
            // weights: packed rows [o][tap * Cin + channel] as bf16 pairs, uniform over the workgroup -> scalar loads
            const uint32_t *wt = w + ((tap * Cin + c * 32) >> 1);
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const uint32_t *wo = wt + o * (Ktot >> 1);
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[o] = dot2(v[j], wo[j], acc[o]);
            }
        }
        if (c + 1 < nc) store(c + 1, r, (c + 1) & 1);
        __syncthreads();
    }
    const int64_t pix = ((int64_t)img * H + y0 + ty) * W + x0 + tx;
    if (y_f32) {
        float *yp = (float *)y + pix * y_cs + y_co;
#pragma unroll
        for (int o = 0; o < COUT; ++o) yp[o] = acc[o];
    } else {
        a16_t *yp = (a16_t *)y + pix * y_cs + y_co;
#pragma unroll
        for (int o = 0; o < COUT; ++o) yp[o] = f2a(acc[o]);
    }
}

// ---- round 4: the same layer with the multiply-adds on the matrix cores ---------------------------------------------------
// The kernel above spends 1 728 v_dot2c per pixel on top of the GroupNorm + SiLU of its window: vector-issue bound (1.25 ms per
// batch of 16 at 768x768 for 2.4 GB of input: 1.9 TB/s).  Here the weights are the A operand of v_mfma_f32_16x16x32 (rows =
// output channels: 3 real ones, the other 13 zero registers that no memory access feeds), 16 pixels of a patch row the B operand,
// so a lane of the first quarter ends up with the layer's channels of one pixel.  Window rows are 96 B in LDS (64 B of data: the
// stride at which the fragment reads of 16 consecutive pixels x 4 channel groups cover all banks once -- checked exhaustively for
// the three column offsets), the COUT real weight rows sit in LDS for the whole workgroup (9 KB), a wave owns four patch rows and
// reads each of its 18 (window row, column offset) fragments once per 32-channel chunk for up to three taps: 36 MFMAs per chunk
// and wave.  Staging, GroupNorm + SiLU on load (same operations as onedc_scale_shift_act) and the chunk pipeline are the kernel
// above's.  What is left of its time is the normalisation of the window and the read of the input.
constexpr int FM_ROW = 96;
constexpr int FM_WIN = 324 * FM_ROW;         // 31104 B

template <int COUT>
__global__ __launch_bounds__(256, 2) void conv_few_out_mfma_kernel(const a16_t *__restrict__ x, const a16_t *__restrict__ w,
                                                                   const float *__restrict__ bias, const float2 *__restrict__ gn_ab,
                                                                   void *__restrict__ y, int H, int W, int Cin, int x_cs, int x_co,
                                                                   int y_cs, int y_co, int y_f32, int gn_silu, int Ktot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const win0 = smem, *const win1 = smem + FM_WIN;
    a16_t *const wl = (a16_t *)(smem + 2 * FM_WIN);                  // [COUT][Ktot]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = W >> 4;
    const int img = blockIdx.y;
    const int y0 = (blockIdx.x / tiles_x) * 16, x0 = (blockIdx.x % tiles_x) * 16;
    const int nc = Cin >> 5;
    const a16_t *xi = x + (int64_t)img * H * W * x_cs + x_co;

    for (int i = tid * 8; i < COUT * Ktot; i += 256 * 8) *(uint4 *)(wl + i) = *(const uint4 *)(w + i);       // rows are Ktot apart in w too

    const int q = tid & 3;
    int goff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = (tid + 256 * i) >> 2;
        const int wy = row / 18, wx = row - wy * 18;
        const int iy = y0 - 1 + wy, ix = x0 - 1 + wx;
        goff[i] = (row < 324 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? (iy * W + ix) * x_cs + q * 8 : -1;
    }
    auto fetch = [&](int c, uint4 (&r)[6]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] = goff[i] >= 0 ? *(const uint4 *)(xi + goff[i] + c * 32) : make_uint4(0u, 0u, 0u, 0u);
    };
    auto store = [&](int c, const uint4 (&r)[6], char *buf) {
        float a[8], b[8];
        if (gn_ab) {
            const float4 *ab = (const float4 *)(gn_ab + (int64_t)img * Cin + c * 32 + q * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 v = ab[e];
                a[2 * e] = v.x;
                b[2 * e] = v.y;
                a[2 * e + 1] = v.z;
                b[2 * e + 1] = v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int row = (tid + 256 * i) >> 2;
            if (row >= 324) continue;
            uint32_t o[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
            if (gn_ab) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float lo = __builtin_fmaf(a2f((a16_t)(o[e] & 0xffff)), a[2 * e], b[2 * e]);
                    float hi = __builtin_fmaf(a2f((a16_t)(o[e] >> 16)), a[2 * e + 1], b[2 * e + 1]);
                    if (gn_silu) {
                        lo = silu_f(lo);
                        hi = silu_f(hi);
                    }
                    o[e] = goff[i] >= 0 ? pack_a2(lo, hi) : 0u;
                }
            }
            *(uint4 *)(buf + row * FM_ROW + q * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    };

    const int n = lane & 15, kq = lane >> 4;          // A: output channel n, k group kq; B: pixel n of a patch row, k group kq
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const a16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    uint4 r6[6];
    fetch(0, r6);
    store(0, r6, win0);
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
        if (c + 1 < nc) fetch(c + 1, r6);
        const char *wb = (c & 1) ? win1 : win0;
        a16x8 fa[9];                                // the nine taps' weight fragments of this chunk (zero for the padding rows)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            fa[tap] = n < COUT ? *(const a16x8 *)(wl + n * Ktot + tap * Cin + c * 32 + kq * 8) : zero8;
#pragma unroll
        for (int wr = 0; wr < 6; ++wr) {            // window rows 4 wave + wr of the wave's four patch rows (+ halo)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const a16x8 fb = *(const a16x8 *)(wb + ((4 * wave + wr) * 18 + n + kx) * FM_ROW + kq * 16);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = wr - ky;          // compile-time after unrolling
                    if (r >= 0 && r < 4) acc[r] = ONEDC_MFMA16(fa[ky * 3 + kx], fb, acc[r]);
                }
            }
        }
        if (c + 1 < nc) store(c + 1, r6, (c & 1) ? win0 : win1);
        __syncthreads();
    }
    if (kq == 0) {                                  // lanes 0 .. 15: channels 0 .. 3 of pixel n of the wave's rows
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t pix = ((int64_t)img * H + y0 + 4 * wave + r) * W + x0 + n;
            if (y_f32) {
                float *yp = (float *)y + pix * y_cs + y_co;
#pragma unroll
                for (int o = 0; o < COUT; ++o) yp[o] = acc[r][o] + (bias ? bias[o] : 0.f);
            } else {
                a16_t *yp = (a16_t *)y + pix * y_cs + y_co;
#pragma unroll
                for (int o = 0; o < COUT; ++o) yp[o] = f2a(acc[r][o] + (bias ? bias[o] : 0.f));
            }
        }
    }
}

}  // namespace

// x: bf16 NHWC view (x_cs, x_co); w: the layer's packed bf16 weight matrix [rows >= Cout][9 * Cin] (pack.conv_matrix
// order, k = (ky, kx, cin)); bias f32 [Cout] or NULL; gn_ab: NULL, or float2 (a, b) [B][Cin] of the GroupNorm in front
// (+SiLU when gn_silu); y: f32 (y_f32) or bf16 NHWC view.  H, W multiples of 16, Cin a multiple of 32, Cout <= 4.
extern "C" int onedc_conv3x3_few_out(const void *x, const void *w, const float *bias, const float *gn_ab, void *y, int B, int H,
                                     int W, int Cin, int x_cs, int x_co, int Cout, int y_cs, int y_co, int y_f32,
                                     int gn_silu, void *stream) {
    if (!x || !w || !y || B < 1 || Cout < 1 || Cout > 4 || (H & 15) || (W & 15) || (Cin & 31) || ((x_cs | x_co) & 7))
        return -1;
    if ((int64_t)H * W * x_cs >= 0x7fffffffll) return -2;        // pixel offsets inside a picture are 32-bit
    const dim3 grid((H >> 4) * (W >> 4), B);
    const int Ktot = 9 * Cin;
    // the matrix-core form (round 4) wherever its weights fit beside the two windows; ONEDC_FEW_OUT_VALU=1 keeps the
    // v_dot2c form (A/B)
    static const bool valu = getenv("ONEDC_FEW_OUT_VALU") && getenv("ONEDC_FEW_OUT_VALU")[0] == '1';
    const int lds = 2 * FM_WIN + Cout * Ktot * 2;
    if (!valu && lds <= 80 * 1024 && (Ktot & 7) == 0) {
        static OnedcLdsAttrOnce once;
        once.ensure({{(const void *)conv_few_out_mfma_kernel<1>, 80 * 1024}, {(const void *)conv_few_out_mfma_kernel<2>, 80 * 1024},
                     {(const void *)conv_few_out_mfma_kernel<3>, 80 * 1024}, {(const void *)conv_few_out_mfma_kernel<4>, 80 * 1024}});
#define FEWM_LAUNCH(N)                                                                                                    \
    hipLaunchKernelGGL(conv_few_out_mfma_kernel<N>, grid, dim3(256), lds, as_stream(stream), (const a16_t *)x, (const a16_t *)w, \
                       bias, (const float2 *)gn_ab, y, H, W, Cin, x_cs, x_co, y_cs, y_co, y_f32, gn_silu, Ktot)
        switch (Cout) {
            case 1: FEWM_LAUNCH(1); break;
            case 2: FEWM_LAUNCH(2); break;
            case 3: FEWM_LAUNCH(3); break;
            default: FEWM_LAUNCH(4); break;
        }
#undef FEWM_LAUNCH
        ONEDC_LAUNCH_CHECK();
        return 0;
    }
#define FEW_LAUNCH(N)                                                                                               \
    hipLaunchKernelGGL(conv_few_out_kernel<N>, grid, dim3(256), 0, as_stream(stream), (const a16_t *)x,              \
                       (const uint32_t *)w, bias, (const float2 *)gn_ab, y, H, W, Cin, x_cs, x_co, y_cs, y_co, y_f32, \
                       gn_silu, Ktot)
    switch (Cout) {
        case 1: FEW_LAUNCH(1); break;
        case 2: FEW_LAUNCH(2); break;
        case 3: FEW_LAUNCH(3); break;
        default: FEW_LAUNCH(4); break;
    }
#undef FEW_LAUNCH
    ONEDC_LAUNCH_CHECK();
    return 0;
}
