// conv_common.h — pieces shared by the implicit-GEMM convolution kernels (gemm_conv.hip: 128-wide
// tiles, 4 waves; gemm_conv256.hip: 256 x 256 tiles, 8 waves, 8-phase schedule).
#pragma once
#include "common.h"

__device__ __forceinline__ void glds16(const char *src, char *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// XCD-aware bijective remap of the launch index: consecutive logical tiles land on one XCD (one L2).
__device__ __forceinline__ int xcd_logical_tile(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Output position of one pixel row m of the GEMM (decomposed once per pixel, used by every channel tile).
struct ConvOutPix {
    int m;            // GEMM row (output pixel on the conv grid)
    int ob, ooy, oox; // (image, y, x) on the conv grid; only filled for pixel_shuffle
};

__device__ __forceinline__ ConvOutPix conv_out_pix(const onedc_conv_params &p, int m) {
    ConvOutPix o;
    o.m = m;
    o.ob = o.ooy = o.oox = 0;
    if (p.pixel_shuffle) {
        const int ohw = p.OH * p.OW;
        o.ob = m / ohw;
        const int rem = m - o.ob * ohw;
        o.ooy = rem / p.OW;
        o.oox = rem - o.ooy * p.OW;
    }
    return o;
}

// Epilogue of one 16x16 MFMA accumulator tile as held by one lane: 4 consecutive packed weight rows
// (nrow16 + 4*fq + e, e = 0..3) of one pixel.  accg is the gate tile of a GLU pair (ignored otherwise).
//   v = act(alpha*acc + bias) [+ residual]  ->  bf16 / f32 store, optional pixel-shuffle scatter.
// ssum / ssq: running sum and sum of squares of the four stored values (GroupNorm statistics of the
// consumer, see conv_stats_store); untouched when the lane stores nothing.
template <bool F32>
__device__ __forceinline__ void conv_epilogue_tile(const onedc_conv_params &p, char *yb, const char *rb, bool glu,
                                                   int cout_store, const f32x4 &accv, const f32x4 &accg, int nrow16,
                                                   int fq, const ConvOutPix &op, float &ssum, float &ssq) {
    const int nrow = nrow16 + fq * 4;              // packed weight row of element 0
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = accv[e] * p.alpha;
        if (p.bias && nrow < p.Cout) t += p.bias[nrow + e];      // rows past Cout are tile padding
        v[e] = t;
    }
    int n_out = nrow;
    if (glu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float g = accg[e] * p.alpha;
            if (p.bias && nrow + 16 < p.Cout) g += p.bias[nrow + 16 + e];
            v[e] = p.epi == ONEDC_EPI_GLU_LEAKY ? leaky(v[e], 0.1f) + leaky(g, 0.01f) : v[e] * gelu_f(g);
        }
        n_out = nrow16 / 2 + fq * 4;
    } else if (p.epi == ONEDC_EPI_LEAKY) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = leaky(v[e], p.slope);
    } else if (p.epi == ONEDC_EPI_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
    }
    if (n_out >= cout_store) return;
    int64_t opix = op.m;                           // output pixel index on the y grid
    if (p.pixel_shuffle) {
        const int cq = p.Cout >> 2;                // channels after the shuffle
        const int sub = n_out / cq;
        opix = ((int64_t)op.ob * (2 * p.OH) + 2 * op.ooy + (sub >> 1)) * (2 * p.OW) + 2 * op.oox + (sub & 1);
        n_out -= sub * cq;
    }
    const bool full = p.pixel_shuffle || n_out + 3 < cout_store;
    if (rb && F32) {                               // f32 residual
        const float *rp = (const float *)rb + opix * p.r_cs + p.r_co + n_out;
        if (full) {
            const float4 rv = *(const float4 *)rp;
            v[0] += rv.x;
            v[1] += rv.y;
            v[2] += rv.z;
            v[3] += rv.w;
        } else {
            for (int e = 0; e < 4 && n_out + e < cout_store; ++e) v[e] += rp[e];
        }
    } else if (rb) {
        const bf16_t *rp = (const bf16_t *)rb + opix * p.r_cs + p.r_co + n_out;
        if (full) {
            const uint2 rv = *(const uint2 *)rp;
            v[0] += bf2f((bf16_t)(rv.x & 0xffff));
            v[1] += bf2f((bf16_t)(rv.x >> 16));
            v[2] += bf2f((bf16_t)(rv.y & 0xffff));
            v[3] += bf2f((bf16_t)(rv.y >> 16));
        } else {
            for (int e = 0; e < 4 && n_out + e < cout_store; ++e) v[e] += bf2f(rp[e]);
        }
    }
    if (p.stats) {
        ssum += (v[0] + v[1]) + (v[2] + v[3]);
        ssq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const int64_t yo = opix * p.y_cs + p.y_co + n_out;
    if (p.out_f32) {
        float *yp = (float *)yb + yo;
        if (full && ((p.y_cs | p.y_co) & 3) == 0) {
            *(float4 *)yp = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int e = 0; e < 4 && n_out + e < cout_store; ++e) yp[e] = v[e];
        }
    } else {
        bf16_t *yp = (bf16_t *)yb + yo;
        if (full && ((p.y_cs | p.y_co) & 3) == 0) {
            *(uint2 *)yp = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        } else {
            for (int e = 0; e < 4 && n_out + e < cout_store; ++e) yp[e] = f2bf(v[e]);
        }
    }
}

// GroupNorm statistics in the producer's epilogue.  A wave covers 64 consecutive output pixels and,
// per 16-row accumulator tile, 4 x 4 consecutive channels ("quads", one per lane quarter fq).  After
// the lane-local sums over the wave's pixel tiles, the 16 lanes of a quarter are reduced with four
// butterfly steps and lane (fq, 0) stores (sum, sum of squares) to
//     stats[(m_wave / 64) * (y_cs / 4) + (y_co + n_out) / 4]          (float2)
// Fixed order, no atomics: every (64-pixel block, quad) has exactly one writer.  onedc_gn_from_stats
// folds the quads of a group and the blocks of an image.  Requires OH*OW % 64 == 0 (a 64-pixel block
// never straddles two images), y_co % 4 == 0, stored channels % 4 == 0, no pixel shuffle.
__device__ __forceinline__ void conv_stats_store(const onedc_conv_params &p, float ssum, float ssq, int m_wave,
                                                 int n_out, int cout_store, int lane, int M) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        ssum += __shfl_xor(ssum, o);
        ssq += __shfl_xor(ssq, o);
    }
    if ((lane & 15) == 0 && n_out < cout_store && m_wave < M) {   // M % 64 == 0: a block is whole or absent
        float2 *dst = (float2 *)p.stats + (int64_t)(m_wave >> 6) * (p.y_cs >> 2) + ((p.y_co + n_out) >> 2);
        *dst = make_float2(ssum, ssq);
    }
}
