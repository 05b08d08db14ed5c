// elementwise.hip — the small HBM-bound heads of the decode path: depthwise 3x3, small-channel
// direct convolution, FSQ code book, the four-part-prior bookkeeping (masks, combine_for_writing,
// build_indexes, y_hat update), the epsilon -> x0 formula, output crop/convert, strided copies.
#include "common.h"

namespace {

// ---- four-part prior helpers (reference modules/entropy/compression_model.py:241-301) ------------
// step k assigns channel group g (C/4 channels each) the 2x2 phase PHASE[k][g];
// phases 0..3 = (y&1, x&1) = (0,0),(0,1),(1,0),(1,1)
__device__ __constant__ int PHASE[4][4] = {{0, 1, 2, 3}, {3, 2, 1, 0}, {2, 3, 0, 1}, {1, 0, 3, 2}};

__device__ __forceinline__ int active_group(int step, int y, int x) {
    const int ph = ((y & 1) << 1) | (x & 1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (PHASE[step][g] == ph) return g;
    return 0;
}

template <bool F32>
__device__ __forceinline__ float ldp(const void *p, int64_t i) {
    if (F32) return ((const float *)p)[i];
    return a2f(((const a16_t *)p)[i]);
}
template <bool F32>
__device__ __forceinline__ void stp(void *p, int64_t i, float v) {
    if (F32) ((float *)p)[i] = v;
    else ((a16_t *)p)[i] = f2a(v);
}

constexpr float LOG_SCALE_MIN = -2.2072749131897207f;           // ln 0.11
constexpr float LOG_SCALE_STEP = 0.02496532547666428f;           // (ln 64 - ln 0.11) / 255

// indexes[b][c4][y][x] (NCHW, the reference's flatten order) from the scale plane of `step`.
// combine_for_writing sums the four masked channel groups: exactly one is non-zero per position.
template <bool F32>
__global__ void prior_indexes_kernel(const void *params, int16_t *idx, int B, int H, int W, int C, int step) {
    const int q = C >> 2;
    const int64_t total = (int64_t)B * q * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        int64_t r = i / W;
        const int y = (int)(r % H);
        r /= H;
        const int c4 = (int)(r % q);
        const int b = (int)(r / q);
        const int g = active_group(step, y, x);
        const int64_t pix = ((int64_t)b * H + y) * W + x;
        float s = ldp<F32>(params, pix * (2 * C) + g * q + c4);     // scales = first C channels
        // the reference adds the three masked-out groups as +0.0f: a -0.0 scale becomes +0.0, harmless
        s = fmaxf(s, 1e-5f);
        float fi = (logf(s) - LOG_SCALE_MIN) / LOG_SCALE_STEP;
        fi = fminf(fmaxf(fi, 0.f), 255.f);
        idx[i] = (int16_t)(int)fi;
    }
}

// mode 0 decode: y_hat += (sym + mean)*mask ; mode 1 encode: sym = round((y-mean)*mask), same update;
// mode 2 z-only: y_hat += mean*mask.  step 0 initialises y_hat (masked-out positions become 0).
template <bool PF32, bool YF32, int MODE>
__global__ void prior_update_kernel(const void *params, const float *yin, int16_t *sym, void *y_hat, int y_cs,
                                    int B, int H, int W, int C, int step) {
    const int q = C >> 2;
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int b = (int)(pix / ((int64_t)W * H));
        const int g = c / q;
        const bool on = active_group(step, y, x) == g;
        float cur = 0.f;
        if (on) {
            const float mean = ldp<PF32>(params, pix * (2 * C) + C + c);
            const int64_t si = (((int64_t)b * q + (c - g * q)) * H + y) * W + x;
            if (MODE == 0) {
                cur = (float)sym[si] + mean;
            } else if (MODE == 1) {
                const float v = rintf(yin[pix * C + c] - mean);          // torch.round: half to even
                const float cl = fminf(fmaxf(v, -30000.f), 30000.f);      // entropy_models.py:67
                sym[si] = (int16_t)cl;
                cur = v + mean;
            } else {
                cur = mean;
            }
        }
        const int64_t yo = pix * y_cs + c;
        if (step == 0) {
            stp<YF32>(y_hat, yo, cur);
        } else if (on) {
            stp<YF32>(y_hat, yo, ldp<YF32>(y_hat, yo) + cur);
        }
    }
}

__global__ void fsq_codes_kernel(const int32_t *idx, float *codes, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * 8;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i & 7);
        const int v = idx[i >> 3];
        codes[i] = d < 7 ? (float)(((v >> (2 * d)) & 3) - 2) * 0.5f : 0.f;
    }
}

// FSQ quantisation, levels [4]*7 (vector-quantize-pytorch FSQ.forward; call site codec_module.py:379):
// q_d = round(tanh(z_d + atanh(0.5 / 1.5015)) * 1.5015 - 0.5) in {-2..1}; index = sum (q_d + 2) 4^d
template <bool F32>
__global__ void fsq_quantize_kernel(const void *z, int z_cs, int z_co, int32_t *idx, int64_t n) {
    const float half_l = 1.5015f;                    // (4 - 1)(1 + 1e-3) / 2
    const float shift = 0.34619963f;                 // atanh(0.5 / 1.5015)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int v = 0;
#pragma unroll
        for (int d = 0; d < 7; ++d) {
            const float q = rintf(tanhf(ldp<F32>(z, i * z_cs + z_co + d) + shift) * half_l - 0.5f);
            v |= (((int)q + 2) & 3) << (2 * d);
        }
        idx[i] = v;
    }
}

__global__ void x0_kernel(const float *reduced, const float *eps, float *x0, int64_t n, double sa, double sb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        x0[i] = (float)(((double)reduced[i] - sb * (double)eps[i]) / sa);
}

// nonfinite: NULL, or int32 [B] zero-filled by the caller: picture b's word becomes 1 when one of its values is NaN / Inf
// BEFORE the clamp (every writer stores the same constant: no atomic needed).  The clamp itself would hide it
// (fmaxf(NaN, -1) = -1): an activation that left the IEEE-half range upstream must not come out as a black pixel.
template <bool F32>
__global__ void image_out_kernel(const void *x, int x_cs, float *out, int B, int PH, int PW, int H, int W,
                                 int unit, int32_t *nonfinite) {
    const int64_t total = (int64_t)B * 3 * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        int64_t r = i / W;
        const int yy = (int)(r % H);
        r /= H;
        const int c = (int)(r % 3);
        const int b = (int)(r / 3);
        float v = ldp<F32>(x, (((int64_t)b * PH + yy) * PW + xx) * x_cs + c);
        if (nonfinite && !(fabsf(v) <= 3.0e38f)) nonfinite[b] = 1;
        if (unit) v = fminf(fmaxf(v, -1.f), 1.f) * 0.5f + 0.5f;
        out[i] = v;
    }
}

// The picture as the 8-bit samples a file or a display takes: clamp(-1,1)*0.5+0.5 (inference.py:106), then
// torchvision.utils.save_image's x255 + 0.5, clamp(0,255), truncation (inference.py:107,120) -- on the device, so that a
// batch leaves over PCIe as 1 byte per sample instead of 4.  One thread per output PIXEL (three samples); nhwc = 1: HWC
// order (what an image writer takes), 0: the reference's CHW planes.
template <bool F32>
__global__ void image_out_u8_kernel(const void *x, int x_cs, uint8_t *out, int B, int PH, int PW, int H, int W,
                                    int nhwc, int32_t *nonfinite) {
    const int64_t total = (int64_t)B * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        int64_t r = i / W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        const int64_t src = (((int64_t)b * PH + yy) * PW + xx) * x_cs;
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = ldp<F32>(x, src + c);
            bad |= !(fabsf(v) <= 3.0e38f);
            v = fminf(fmaxf(v, -1.f), 1.f) * 0.5f + 0.5f;
            v = fminf(fmaxf(__fadd_rn(__fmul_rn(v, 255.f), 0.5f), 0.f), 255.f);   // mul, then add: two roundings like torch's mul(255).add_(0.5)
            const uint8_t q = (uint8_t)(int)v;
            if (nhwc) out[i * 3 + c] = q;
            else out[(((int64_t)b * 3 + c) * H + yy) * W + xx] = q;
        }
        if (bad && nonfinite) nonfinite[b] = 1;
    }
}

template <bool XF, bool YF>
__global__ void copy_channels_kernel(const void *x, int x_cs, int x_co, void *y, int y_cs, int y_co,
                                     int64_t pixels, int C) {
    const int64_t total = pixels * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / C;
        const int c = (int)(i - pix * C);
        stp<YF>(y, pix * y_cs + y_co + c, ldp<XF>(x, pix * x_cs + x_co + c));
    }
}

// ---- encode side: picture input ---------------------------------------------------------------------------
// NCHW f32 picture -> NHWC (f32 or bf16) with channel stride y_cs (channels >= C are left untouched: the
// caller zero-fills once), optionally a replicate-padded (Hp, Wp) frame of an (H, W) picture
// (model_sd15_with_codec_stage1.py:281-285: F.pad(..., mode="replicate") up to a multiple of 64)
template <bool YF>
__global__ void nchw_to_nhwc_kernel(const float *x, void *y, int B, int C, int H, int W, int Hp, int Wp, int y_cs,
                                    int y_co, int reflect) {
    const int64_t total = (int64_t)B * Hp * Wp * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int xx = (int)(r % Wp);
        r /= Wp;
        const int yy = (int)(r % Hp);
        const int b = (int)(r / Hp);
        // replicate (stage-1 model) or reflect (z-only harness, models/sd15_onedc_codec_z_only/inference.py:106-109)
        const int sy = yy < H ? yy : (reflect ? 2 * (H - 1) - yy : H - 1);
        const int sx = xx < W ? xx : (reflect ? 2 * (W - 1) - xx : W - 1);
        stp<YF>(y, (((int64_t)b * Hp + yy) * Wp + xx) * y_cs + y_co + c, x[(((int64_t)b * C + c) * H + sy) * W + sx]);
    }
}

// 8x8 / stride 8 patch embedding input (Encoder.pix_emb, codec_module.py:60): NHWC f32 picture (3 channels of a
// buffer with stride x_cs) -> bf16 [B][H/8][W/8][192], k = (ky, kx, c): the conv is then a 1x1 GEMM
__global__ void patchify8_kernel(const float *x, int x_cs, a16_t *y, int B, int H, int W) {
    const int h8 = H >> 3, w8 = W >> 3;
    const int64_t total = (int64_t)B * h8 * w8 * 192;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % 192);
        int64_t r = i / 192;
        const int px = (int)(r % w8);
        r /= w8;
        const int py = (int)(r % h8);
        const int b = (int)(r / h8);
        const int c = k % 3, kx = (k / 3) & 7, ky = k / 24;
        y[i] = f2a(x[(((int64_t)b * H + py * 8 + ky) * W + px * 8 + kx) * x_cs + c]);
    }
}

// DiagonalGaussianDistribution.sample * scaling_factor (diffusers; call model_sd15_with_codec_stage1.py:193-194):
// moments f32 NHWC [n][8] = (mean 4 | logvar 4); out = (mean + exp(0.5 clamp(logvar, -30, 20)) * noise) * scale
__global__ void latent_sample_kernel(const float *moments, const float *noise, float *out, int64_t n, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * 4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i >> 2;
        const int c = (int)(i & 3);
        const float mean = moments[pix * 8 + c];
        const float lv = fminf(fmaxf(moments[pix * 8 + 4 + c], -30.f), 20.f);
        const float nz = noise ? noise[i] : 0.f;
        out[i] = (mean + expf(0.5f * lv) * nz) * scale;
    }
}

// region [y0,y0+RH) x [x0,x0+RW) of (B,H,W,C) <-> (B*nwin, wh*ww, C) windows, 8 channels per lane
__global__ void window_partition_kernel(const a16_t *x, a16_t *y, int B, int H, int W, int C, int y0, int x0,
                                        int RH, int RW, int wh, int ww, int reverse) {
    const int CC = C >> 3;
    const int nwx = RW / ww, nwy = RH / wh;
    const int64_t total = (int64_t)B * RH * RW * CC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % CC);
        const int64_t wp = i / CC;                       // index in window-major order
        const int pin = (int)(wp % (wh * ww));
        int64_t r = wp / (wh * ww);
        const int wx = (int)(r % nwx);
        r /= nwx;
        const int wy = (int)(r % nwy);
        const int b = (int)(r / nwy);
        const int64_t pix = ((int64_t)b * H + y0 + wy * wh + pin / ww) * W + x0 + wx * ww + pin % ww;
        if (!reverse)
            ((uint4 *)y)[wp * CC + cc] = ((const uint4 *)x)[pix * CC + cc];
        else
            ((uint4 *)y)[pix * CC + cc] = ((const uint4 *)x)[wp * CC + cc];
    }
}

// ---- MXFP8 experiment ------------------------------------------------------------------------------------
// What v_mfma_scale_f32_16x16x128_f8f6f4 would consume, made visible to the 16-bit kernels: every block of 32 consecutive
// channels of a pixel (= 32 consecutive k of a conv's GEMM: k = (ky, kx, cin)) is quantised to OCP e4m3 (4-bit significand,
// exponents -6 .. 8, subnormal step 2^-9, saturating at 448) with ONE shared power-of-two scale 2^(floor(log2 amax) - 8)
// (OCP MX v1.0), and written back dequantised.  q * 2^e is exact in either 16-bit format, an MFMA on these operands forms the
// same exact products the block-scaled instruction forms, and both accumulate in f32: the ACCURACY of an MXFP8 layer is
// measured on hardware without its kernel (tests/test_gpu_fullsize.py::test_mxfp8_*).  One thread per (pixel, block).
__device__ __forceinline__ float e4m3_round(float v) {
    float a = fminf(fabsf(v), 448.f);
    int ex = (int)(__float_as_uint(a) >> 23) - 127;
    ex = ex < -6 ? -6 : ex;
    const float q = __uint_as_float((uint32_t)(ex - 3 + 127) << 23);          // spacing of e4m3 at this magnitude
    a = fminf(rintf(a / q) * q, 448.f);                                        // exact division, round half to even
    return copysignf(a, v);
}

__global__ void mxfp8_roundtrip_kernel(const a16_t *x, a16_t *y, int64_t pixels, int C, int x_cs, int x_co, int y_cs,
                                       int y_co) {
    const int nb = C >> 5;
    const int64_t total = pixels * nb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / nb;
        const int b = (int)(i - pix * nb);
        const uint4 *src = (const uint4 *)(x + pix * x_cs + x_co + b * 32);
        uint4 *dst = (uint4 *)(y + pix * y_cs + y_co + b * 32);
        uint4 v[4];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = src[j];
            const uint32_t *w = (const uint32_t *)&v[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(a2f_lo(w[e])), fabsf(a2f_hi(w[e]))));
        }
        int es = (int)(__float_as_uint(amax) >> 23) - 127 - 8;                 // floor(log2 amax) - emax(e4m3)
        es = amax > 0.f ? (es < -126 ? -126 : es) : 0;
        const float scale = __uint_as_float((uint32_t)(es + 127) << 23), inv = __uint_as_float((uint32_t)(127 - es) << 23);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t *w = (uint32_t *)&v[j];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] = pack_a2(e4m3_round(a2f_lo(w[e]) * inv) * scale, e4m3_round(a2f_hi(w[e]) * inv) * scale);
            dst[j] = v[j];
        }
    }
}

__global__ void timestep_embedding_kernel(const float *t, float *emb, int n, int dim) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    const int row = i / half, k = i - row * half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);   // ln 10000
    const float a = t[row] * freq;
    emb[row * dim + k] = cosf(a);                 // flip_sin_to_cos: cos half first
    emb[row * dim + half + k] = sinf(a);
}

__global__ void add_kernel(const a16_t *a, const a16_t *b, a16_t *y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 va = ((const uint4 *)a)[i], vb = ((const uint4 *)b)[i];
        const uint32_t *pa = (const uint32_t *)&va, *pb = (const uint32_t *)&vb;
        uint4 o;
        uint32_t *po = (uint32_t *)&o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            po[e] = pack_a2(a2f((a16_t)(pa[e] & 0xffff)) + a2f((a16_t)(pb[e] & 0xffff)),
                             a2f((a16_t)(pa[e] >> 16)) + a2f((a16_t)(pb[e] >> 16)));
        ((uint4 *)y)[i] = o;
    }
}

// depthwise 3x3 pad 1: 8 channels per lane, weights f32 [9][C]
__global__ void dwconv3x3_kernel(const a16_t *__restrict__ x, const float *__restrict__ w,
                                 const float *__restrict__ bias, a16_t *__restrict__ y, int B, int H, int W,
                                 int C, int x_cs, int x_co, int y_cs, int y_co) {
    const int CC = C >> 3;
    const int64_t total = (int64_t)B * H * W * CC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % CC);
        const int64_t pix = i / CC;
        const int xx = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        const int64_t b = pix / ((int64_t)W * H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[cc * 8 + e] : 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = yy + ky - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = xx + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const uint4 v = *(const uint4 *)(x + ((b * H + iy) * W + ix) * x_cs + x_co + cc * 8);
                const uint32_t *pv = (const uint32_t *)&v;
                const float *wp = w + (ky * 3 + kx) * C + cc * 8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += a2f((a16_t)(pv[e] & 0xffff)) * wp[2 * e];
                    acc[2 * e + 1] += a2f((a16_t)(pv[e] >> 16)) * wp[2 * e + 1];
                }
            }
        }
        *(uint4 *)(y + pix * y_cs + y_co + cc * 8) =
            make_uint4(pack_a2(acc[0], acc[1]), pack_a2(acc[2], acc[3]), pack_a2(acc[4], acc[5]),
                       pack_a2(acc[6], acc[7]));
    }
}

// f32 variant (prior networks): 4 channels per lane, 16-byte accesses
__global__ void dwconv3x3_f32_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                     const float *__restrict__ bias, float *__restrict__ y, int B, int H, int W,
                                     int C, int x_cs, int x_co, int y_cs, int y_co) {
    const int CC = C >> 2;
    const int64_t total = (int64_t)B * H * W * CC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % CC);
        const int64_t pix = i / CC;
        const int xx = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        const int64_t b = pix / ((int64_t)W * H);
        float4 acc = bias ? *(const float4 *)(bias + cc * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = yy + ky - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = xx + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const float4 v = *(const float4 *)(x + ((b * H + iy) * W + ix) * x_cs + x_co + cc * 4);
                const float4 wv = *(const float4 *)(w + (ky * 3 + kx) * C + cc * 4);
                acc.x = __builtin_fmaf(v.x, wv.x, acc.x);
                acc.y = __builtin_fmaf(v.y, wv.y, acc.y);
                acc.z = __builtin_fmaf(v.z, wv.z, acc.z);
                acc.w = __builtin_fmaf(v.w, wv.w, acc.w);
            }
        }
        *(float4 *)(y + pix * y_cs + y_co + cc * 4) = acc;
    }
}

// direct conv for Cin < 64: one thread per (pixel, cout); w f32 [Cout][K][K][Cin]
template <bool XF>
__global__ void direct_conv_kernel(const onedc_direct_conv_params p) {
    const int64_t total = (int64_t)p.B * p.H * p.W * p.Cout;
    const int pad = p.K / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % p.Cout);
        const int64_t pix = i / p.Cout;
        const int xx = (int)(pix % p.W);
        const int yy = (int)((pix / p.W) % p.H);
        const int64_t b = pix / ((int64_t)p.W * p.H);
        float acc = 0.f;
        for (int ky = 0; ky < p.K; ++ky) {
            const int iy = yy + ky - pad;
            if (iy < 0 || iy >= p.H) continue;
            for (int kx = 0; kx < p.K; ++kx) {
                const int ix = xx + kx - pad;
                if (ix < 0 || ix >= p.W) continue;
                const int64_t xo = ((b * p.H + iy) * p.W + ix) * p.x_cs + p.x_co;
                const float *wp = p.w + ((int64_t)(n * p.K + ky) * p.K + kx) * p.Cin;
                for (int c = 0; c < p.Cin; ++c) acc += ldp<XF>(p.x, xo + c) * p.in_scale * wp[c];
            }
        }
        if (p.bias) acc += p.bias[n];
        if (p.epi == ONEDC_EPI_LEAKY) acc = leaky(acc, p.slope);
        const int64_t yo = pix * p.y_cs + p.y_co + n;
        if (p.out_f32) ((float *)p.y)[yo] = acc;
        else ((a16_t *)p.y)[yo] = f2a(acc);
    }
}

inline int grid_for(int64_t n, int per = 256) { return (int)min((int64_t)16384, (n + per - 1) / per); }

}  // namespace

#define LAUNCH(kern, n, ...)                                                                     \
    do {                                                                                         \
        if ((n) > 0) hipLaunchKernelGGL(kern, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), \
                                        __VA_ARGS__);                                            \
        ONEDC_LAUNCH_CHECK();                                                                    \
    } while (0)

extern "C" {

const char *onedc_hip_version(void) { return "onedc-hip 0.5 (gfx950, " ONEDC_ACT_NAME " activations)"; }
int onedc_hip_abi_version(void) { return ONEDC_HIP_ABI_VERSION; }
// identity of the sources this library was compiled from (onedc_amd/build.py: hip_source_id); the loader compares it with the
// sources next to it, so a stale .so is an error instead of a wrong answer
#ifndef ONEDC_BUILD_ID
#define ONEDC_BUILD_ID "unidentified"
#endif
static const char onedc_build_id_marker[] = "@(#)onedc-build-id:" ONEDC_BUILD_ID;      // (found in the file by build.py)
const char *onedc_hip_build_id(void) { return onedc_build_id_marker + 19; }
// the 16-bit activation format this library was built for: "bf16" or "f16" (exactly; the loader compares it)
const char *onedc_hip_act_format(void) { return ONEDC_ACT_NAME; }

int onedc_fsq_codes(const int32_t *idx, float *codes, int64_t n, void *stream) {
    if (!idx || !codes) return -1;
    LAUNCH(fsq_codes_kernel, n * 8, idx, codes, n);
    return 0;
}

int onedc_fsq_quantize(const void *z, int z_f32, int z_cs, int z_co, int32_t *idx, int64_t n, void *stream) {
    if (!z || !idx || z_cs < 7) return -1;
    if (z_f32) LAUNCH(fsq_quantize_kernel<true>, n, z, z_cs, z_co, idx, n);
    else LAUNCH(fsq_quantize_kernel<false>, n, z, z_cs, z_co, idx, n);
    return 0;
}

int onedc_prior_indexes(const void *params, int params_f32, int16_t *indexes, int B, int H, int W, int C,
                        int step, void *stream) {
    if (!params || !indexes || (C & 3) || step < 0 || step > 3) return -1;
    const int64_t n = (int64_t)B * (C / 4) * H * W;
    if (params_f32) LAUNCH(prior_indexes_kernel<true>, n, params, indexes, B, H, W, C, step);
    else LAUNCH(prior_indexes_kernel<false>, n, params, indexes, B, H, W, C, step);
    return 0;
}

#define PRIOR_DISPATCH(MODE, yin, sym)                                                                          \
    do {                                                                                                        \
        const int64_t n = (int64_t)B * H * W * C;                                                               \
        if (params_f32 && yhat_f32) LAUNCH((prior_update_kernel<true, true, MODE>), n, params, yin, sym, y_hat, y_cs, B, H, W, C, step); \
        else if (params_f32) LAUNCH((prior_update_kernel<true, false, MODE>), n, params, yin, sym, y_hat, y_cs, B, H, W, C, step);       \
        else if (yhat_f32) LAUNCH((prior_update_kernel<false, true, MODE>), n, params, yin, sym, y_hat, y_cs, B, H, W, C, step);        \
        else LAUNCH((prior_update_kernel<false, false, MODE>), n, params, yin, sym, y_hat, y_cs, B, H, W, C, step);                     \
    } while (0)

int onedc_prior_update_dec(const void *params, int params_f32, const int16_t *symbols, void *y_hat,
                           int yhat_f32, int y_cs, int B, int H, int W, int C, int step, void *stream) {
    if (!params || !symbols || !y_hat || (C & 3) || step < 0 || step > 3) return -1;
    PRIOR_DISPATCH(0, (const float *)nullptr, (int16_t *)symbols);
    return 0;
}
int onedc_prior_update_enc(const void *params, int params_f32, const float *y, int16_t *symbols, void *y_hat,
                           int yhat_f32, int y_cs, int B, int H, int W, int C, int step, void *stream) {
    if (!params || !y || !symbols || !y_hat || (C & 3) || step < 0 || step > 3) return -1;
    PRIOR_DISPATCH(1, y, symbols);
    return 0;
}
int onedc_prior_update_zonly(const void *params, int params_f32, void *y_hat, int yhat_f32, int y_cs, int B,
                             int H, int W, int C, int step, void *stream) {
    if (!params || !y_hat || (C & 3) || step < 0 || step > 3) return -1;
    PRIOR_DISPATCH(2, (const float *)nullptr, (int16_t *)nullptr);
    return 0;
}

int onedc_x0_from_noise(const float *reduced, const float *eps, float *x0, int64_t n, double alpha_bar,
                        void *stream) {
    if (!reduced || !eps || !x0 || !(alpha_bar > 0.0 && alpha_bar < 1.0)) return -1;
    LAUNCH(x0_kernel, n, reduced, eps, x0, n, sqrt(alpha_bar), sqrt(1.0 - alpha_bar));
    return 0;
}

int onedc_image_out(const void *x, int x_f32, int x_cs, float *out, int B, int PH, int PW, int H, int W, int unit_range,
                    void *stream) {                 // the 0.2 signature, kept: a host built against it must not break
    return onedc_image_out_checked(x, x_f32, x_cs, out, B, PH, PW, H, W, unit_range, nullptr, stream);
}

int onedc_image_out_checked(const void *x, int x_f32, int x_cs, float *out, int B, int PH, int PW, int H, int W,
                            int unit_range, int32_t *nonfinite, void *stream) {
    if (!x || !out || H > PH || W > PW) return -1;
    const int64_t n = (int64_t)B * 3 * H * W;
    if (x_f32) LAUNCH(image_out_kernel<true>, n, x, x_cs, out, B, PH, PW, H, W, unit_range, nonfinite);
    else LAUNCH(image_out_kernel<false>, n, x, x_cs, out, B, PH, PW, H, W, unit_range, nonfinite);
    return 0;
}

int onedc_image_out_u8(const void *x, int x_f32, int x_cs, uint8_t *out, int B, int PH, int PW, int H, int W,
                       int nhwc, int32_t *nonfinite, void *stream) {
    if (!x || !out || H > PH || W > PW) return -1;
    const int64_t n = (int64_t)B * H * W;
    if (x_f32) LAUNCH(image_out_u8_kernel<true>, n, x, x_cs, out, B, PH, PW, H, W, nhwc, nonfinite);
    else LAUNCH(image_out_u8_kernel<false>, n, x, x_cs, out, B, PH, PW, H, W, nhwc, nonfinite);
    return 0;
}

int onedc_copy_channels(const void *x, int x_f32, int x_cs, int x_co, void *y, int y_f32, int y_cs, int y_co,
                        int64_t pixels, int C, void *stream) {
    if (!x || !y) return -1;
    const int64_t n = pixels * C;
    if (x_f32 && y_f32) LAUNCH((copy_channels_kernel<true, true>), n, x, x_cs, x_co, y, y_cs, y_co, pixels, C);
    else if (x_f32) LAUNCH((copy_channels_kernel<true, false>), n, x, x_cs, x_co, y, y_cs, y_co, pixels, C);
    else if (y_f32) LAUNCH((copy_channels_kernel<false, true>), n, x, x_cs, x_co, y, y_cs, y_co, pixels, C);
    else LAUNCH((copy_channels_kernel<false, false>), n, x, x_cs, x_co, y, y_cs, y_co, pixels, C);
    return 0;
}

int onedc_nchw_to_nhwc(const float *x, void *y, int y_f32, int B, int C, int H, int W, int Hp, int Wp, int y_cs,
                       int y_co, int reflect, void *stream) {
    if (!x || !y || Hp < H || Wp < W || C < 1) return -1;
    if (reflect && (Hp - H >= H || Wp - W >= W)) return -2;
    const int64_t n = (int64_t)B * Hp * Wp * C;
    if (y_f32) LAUNCH(nchw_to_nhwc_kernel<true>, n, x, y, B, C, H, W, Hp, Wp, y_cs, y_co, reflect);
    else LAUNCH(nchw_to_nhwc_kernel<false>, n, x, y, B, C, H, W, Hp, Wp, y_cs, y_co, reflect);
    return 0;
}

int onedc_patchify8(const float *x, int x_cs, void *y, int B, int H, int W, void *stream) {
    if (!x || !y || (H & 7) || (W & 7) || x_cs < 3) return -1;
    const int64_t n = (int64_t)B * (H / 8) * (W / 8) * 192;
    LAUNCH(patchify8_kernel, n, x, x_cs, (a16_t *)y, B, H, W);
    return 0;
}

int onedc_latent_sample(const float *moments, const float *noise, float *out, int64_t n, float scale, void *stream) {
    if (!moments || !out) return -1;
    LAUNCH(latent_sample_kernel, n * 4, moments, noise, out, n, scale);
    return 0;
}

int onedc_window_partition(const void *x, void *y, int B, int H, int W, int C, int y0, int x0, int RH, int RW,
                           int wh, int ww, int reverse, void *stream) {
    if (!x || !y || wh < 1 || ww < 1 || RH % wh || RW % ww || (C & 7)) return -1;
    if (y0 < 0 || x0 < 0 || y0 + RH > H || x0 + RW > W) return -2;
    const int64_t n = (int64_t)B * RH * RW * (C / 8);
    LAUNCH(window_partition_kernel, n, (const a16_t *)x, (a16_t *)y, B, H, W, C, y0, x0, RH, RW, wh, ww, reverse);
    return 0;
}

int onedc_mxfp8_roundtrip(const void *x, void *y, int64_t pixels, int C, int x_cs, int x_co, int y_cs, int y_co,
                          void *stream) {
    if (!x || !y || (C & 31) || ((x_cs | x_co | y_cs | y_co) & 7)) return -1;
    const int64_t n = pixels * (C / 32);
    LAUNCH(mxfp8_roundtrip_kernel, n, (const a16_t *)x, (a16_t *)y, pixels, C, x_cs, x_co, y_cs, y_co);
    return 0;
}

int onedc_timestep_embedding(const float *t, float *emb, int n, int dim, void *stream) {
    if (!t || !emb || (dim & 1)) return -1;
    LAUNCH(timestep_embedding_kernel, (int64_t)n * (dim / 2), t, emb, n, dim);
    return 0;
}

int onedc_add(const void *a, const void *b, void *y, int64_t n, void *stream) {
    if (!a || !b || !y || (n & 7)) return -1;
    LAUNCH(add_kernel, n / 8, (const a16_t *)a, (const a16_t *)b, (a16_t *)y, n / 8);
    return 0;
}

int onedc_dwconv3x3_f32(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int C,
                        int x_cs, int x_co, int y_cs, int y_co, void *stream) {
    if (!x || !w || !y || (C & 3) || (x_cs & 3) || (x_co & 3) || (y_cs & 3) || (y_co & 3)) return -1;
    const int64_t n = (int64_t)B * H * W * (C / 4);
    LAUNCH(dwconv3x3_f32_kernel, n, x, w, bias, y, B, H, W, C, x_cs, x_co, y_cs, y_co);
    return 0;
}

int onedc_dwconv3x3(const void *x, const float *w, const float *bias, void *y, int B, int H, int W, int C,
                    int x_cs, int x_co, int y_cs, int y_co, void *stream) {
    if (!x || !w || !y || (C & 7) || (x_cs & 7) || (x_co & 7) || (y_cs & 7) || (y_co & 7)) return -1;
    const int64_t n = (int64_t)B * H * W * (C / 8);
    LAUNCH(dwconv3x3_kernel, n, (const a16_t *)x, w, bias, (a16_t *)y, B, H, W, C, x_cs, x_co, y_cs, y_co);
    return 0;
}

int onedc_direct_conv(const onedc_direct_conv_params *pp, void *stream) {
    if (!pp || !pp->x || !pp->w || !pp->y || pp->Cin < 1 || pp->Cin > 64 || (pp->K != 1 && pp->K != 3)) return -1;
    const int64_t n = (int64_t)pp->B * pp->H * pp->W * pp->Cout;
    if (pp->x_f32) LAUNCH(direct_conv_kernel<true>, n, *pp);
    else LAUNCH(direct_conv_kernel<false>, n, *pp);
    return 0;
}

}  // extern "C"
