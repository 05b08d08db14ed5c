// conv_common.h — pieces shared by the implicit-GEMM convolution kernels (gemm_conv.hip: 128-wide
// tiles, 4 waves; gemm_conv256.hip: 256 x 256 tiles, 8 waves, 8-phase schedule).
#pragma once
#include "common.h"

__device__ __forceinline__ void glds16(const char *src, char *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: global address = descriptor base + voff (per lane) + soff (wave-uniform
// SGPR), range-checked against the descriptor's byte count.  A lane whose voff is >= that count transfers ZEROS
// to LDS: halo / tail rows need no zero page and no address select, and everything that is uniform over the
// wave (K-tile, channel step, row-block stride) rides in soff, so the per-instruction VALU work is nil.
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int onedc_rsrc_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void glds16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char *lds_dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_dst, 16, (int)voff,
                                             (int)soff, 0, 0);
}

// Division of a non-negative 31-bit integer by a launch constant as one multiply-high and a shift (the divisor's magic
// pair is computed on the host): tile decoding sits in front of a workgroup's first DMA, where a 25-instruction software
// division per quotient is dead time on every CU.
struct OnedcFastDiv {
    uint32_t mul, shift;
    __host__ __device__ explicit OnedcFastDiv(uint32_t d = 1) : mul(0), shift(0) {
        if (d > 1) {
            while ((1ull << shift) < d) ++shift;                          // ceil(log2 d)
            mul = (uint32_t)(((uint64_t)1 << (31 + shift)) / d + 1);     // Granlund-Montgomery, 31-bit numerators
        }
    }
    // exact for 0 <= x < 2^31
    __device__ __forceinline__ int div(int x) const {
        return mul ? (int)(((uint64_t)(uint32_t)x * mul) >> (31 + shift)) : x;
    }
};

// XCD-aware bijective remap of the launch index: consecutive logical tiles land on one XCD (one L2).
__device__ __forceinline__ int xcd_logical_tile(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- implicit-GEMM pixel sources ------------------------------------------------------------------------
// Each lane stages NP pixel rows of the B operand.  Output pixel m = (b, oy, ox) is decoded once; for a
// tap (ky, kx) the input position is (oy*stride - pad + ky, ox*stride - pad + kx) (on the x2 grid when
// the nearest-neighbour upsample is fused: then >> 1).  Switching taps costs ~10 VALU per pixel row, all
// 32-bit except one 64-bit multiply-add for the byte address.
template <int NP>
struct ConvPixSrc {
    int base[NP];      // pixel index of (b, 0, 0) in the input
    int iyx0[NP];      // (iy0 << 16) | (ix0 & 0xffff) for tap (0, 0)
    unsigned valid;    // bit i: row i is a real output pixel (m < M)
    const char *src[NP];
    unsigned ok;       // bit i: row i reads a real input pixel for the current tap (else: zeros)

    // img0: image that pixel indices are relative to (0, or the first image of the workgroup's tile when the
    // byte offsets must stay below 4 GB: buffer-descriptor DMA)
    __device__ __forceinline__ void init(const onedc_conv_params &p, const int (&m)[NP], int M, int img0 = 0) {
        valid = 0;
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int mm = m[i] < M ? m[i] : img0 * ohw;
            const int b = mm / ohw, rem = mm - b * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            base[i] = (b - img0) * p.IH * p.IW;
            iyx0[i] = (int)((unsigned)(oy * p.stride - p.pad) << 16) | ((ox * p.stride - p.pad) & 0xffff);
            valid |= (m[i] < M) ? (1u << i) : 0u;
        }
    }
    // byte offsets instead of pointers (buffer-descriptor DMA): off[i] relative to the input base, `oob` for rows
    // that read zeros.  lane_off = (x_co * elem + chunk * 16)
    unsigned off[NP];
    __device__ __forceinline__ void tap_off(const onedc_conv_params &p, unsigned lane_off, int pix_bytes, int ky, int kx,
                                            unsigned oob) {
        const int ly = p.up ? 2 * p.IH : p.IH, lx = p.up ? 2 * p.IW : p.IW;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int iy = (iyx0[i] >> 16) + ky;
            int ix = (int)(short)(iyx0[i] & 0xffff) + kx;
            const bool in = ((valid >> i) & 1) && (unsigned)iy < (unsigned)ly && (unsigned)ix < (unsigned)lx;
            if (p.up) {
                iy >>= 1;
                ix >>= 1;
            }
            const unsigned pix = (unsigned)(base[i] + iy * p.IW + ix);
            off[i] = in ? pix * (unsigned)pix_bytes + lane_off : oob;
        }
    }
    // 1x1 / stride 1 / pad 0 layers (two thirds of all launches: every Linear of the transformer blocks, every 1x1 of
    // the DepthConvBlocks): input pixel == output pixel, nothing to decode -- the general init + tap above cost some 200
    // instructions of integer division per lane in front of a workgroup's first DMA
    __device__ __forceinline__ void init_linear(const char *xb0, int pix_bytes, const int (&m)[NP], int M) {
        valid = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const bool in = m[i] < M;
            valid |= in ? (1u << i) : 0u;
            base[i] = 0;
            iyx0[i] = 0;
            src[i] = xb0 + (int64_t)(in ? m[i] : 0) * pix_bytes;
        }
        ok = valid;
    }
    // xb0 = input base + (x_co * elem + chunk * 16) bytes; pix_bytes = x_cs * elem
    __device__ __forceinline__ void tap(const onedc_conv_params &p, const char *xb0, int pix_bytes, int ky, int kx) {
        ok = 0;
        const int ly = p.up ? 2 * p.IH : p.IH, lx = p.up ? 2 * p.IW : p.IW;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int iy = (iyx0[i] >> 16) + ky;
            int ix = (int)(short)(iyx0[i] & 0xffff) + kx;
            const bool in = ((valid >> i) & 1) && (unsigned)iy < (unsigned)ly && (unsigned)ix < (unsigned)lx;
            if (p.up) {
                iy >>= 1;
                ix >>= 1;
            }
            const int pix = base[i] + iy * p.IW + ix;
            src[i] = xb0 + (int64_t)pix * pix_bytes;
            ok |= in ? (1u << i) : 0u;
        }
    }
};

// 2-D pixel tiles.  A GEMM row tile of 256 pixels is by default 256 consecutive pixels of the row-major
// (image, y, x) order: at 768 pixels per line that is a third of a line, and the 3x3 taps of a tile touch
// three such segments whose neighbours above and below belong to tiles that run much later (or on another
// XCD) -- measured HBM reads of the 256-tile kernel were 4.6x the input bytes.  With patch16 the tile is a
// 16 x 16 patch: its nine taps touch one 18 x 18 window (1.27x the patch), and tiles that follow each other
// on an XCD are neighbouring patches.  Needs OH, OW multiples of 16.
__host__ __device__ __forceinline__ bool conv_is_linear(const onedc_conv_params &p) {      // wave-uniform (kernel arguments only)
    return p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && !p.up && p.IH == p.OH && p.IW == p.OW;
}

struct ConvPatch {
    int img, y0, x0;      // patch origin of this workgroup's pixel tile (patch16 only)
    bool on;
};
__device__ __forceinline__ ConvPatch conv_patch(const onedc_conv_params &p, int m0) {
    ConvPatch c;
    c.on = p.patch16 != 0;
    c.img = c.y0 = c.x0 = 0;
    if (c.on) {
        const int px = p.OW >> 4, pi = (p.OH >> 4) * px;
        const int T = m0 >> 8;
        c.img = T / pi;
        const int r = T - c.img * pi;
        const int ty = r / px;
        c.y0 = ty << 4;
        c.x0 = (r - ty * px) << 4;
    }
    return c;
}
// linear output pixel index of tile position t (0..255) of the tile starting at GEMM row m0
__device__ __forceinline__ int conv_tile_pixel(const onedc_conv_params &p, const ConvPatch &c, int m0, int t) {
    if (c.on) return (c.img * p.OH + c.y0 + (t >> 4)) * p.OW + c.x0 + (t & 15);
    return m0 + t;
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

// ---- epilogue ------------------------------------------------------------------------------------------
// One 16x16 MFMA accumulator tile as held by one lane = 4 consecutive packed weight rows
// (nrow16 + 4*fq + e, e = 0..3) of one pixel.   v = act(alpha*acc + bias) [+ residual]  ->  bf16 / f32
// store, optional pixel-shuffle scatter.  The epilogue of a workgroup runs in two passes so that no
// load ever queues behind a store (vmcnt retires in order, a store takes microseconds to retire):
//   pass 1  conv_epi_pos + conv_epi_load_res for every tile: index math, residual loads in flight
//   pass 2  conv_epi_finish: activation, residual add, statistics, store
struct EpiPos {
    int64_t opix;     // output pixel index on the y grid
    int n_out;        // first of the four output channels
    bool store;       // the lane has something to store
    bool full;        // all four channels exist (vector access allowed if aligned)
};

// Compile-time description of the epilogue of a launch.  EpiGeneric decides everything at run time (f32 / bf16
// output, pixel shuffle, GLU pairs, partial or unaligned channel quads: ~300 instructions per 16x16 tile).
// EpiMode<true, ACT> is the common case -- bf16 output, no shuffle, whole 8-byte-aligned channel quads, activation
// ACT known -- where a tile is ~25 instructions.  Same arithmetic in the same order: results are bit-identical.
template <bool FAST, int ACT>
struct EpiMode {
    static constexpr bool fast = FAST;
    static constexpr int act = ACT;
    static constexpr bool glu = FAST && (ACT == ONEDC_EPI_GLU_LEAKY || ACT == ONEDC_EPI_GEGLU);
};
typedef EpiMode<false, 0> EpiGeneric;
__host__ __device__ __forceinline__ bool conv_epi_mode_fast(const onedc_conv_params &p) {
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    const int cout_store = glu ? p.Cout / 2 : p.Cout;
    // bf16 in / bf16 out, or the exact-f32 path with f32 output (16-byte stores: same 4-element alignment rule)
    return (p.out_f32 != 0) == (p.in_f32 != 0) && !p.out_bf16 && !p.pixel_shuffle && ((p.y_cs | p.y_co | cout_store) & 3) == 0 &&
           (!p.res || ((p.r_cs | p.r_co) & 3) == 0) && p.epi >= ONEDC_EPI_LINEAR && p.epi <= ONEDC_EPI_GEGLU;
}

template <class M = EpiGeneric>
__device__ __forceinline__ EpiPos conv_epi_pos(const onedc_conv_params &p, bool glu, int cout_store, int nrow16,
                                               int fq, const ConvOutPix &op, bool pixel_ok) {
    EpiPos e;
    if (M::fast) {
        e.n_out = (M::glu ? nrow16 / 2 : nrow16) + fq * 4;
        e.store = pixel_ok && e.n_out < cout_store;
        e.opix = op.m;
        e.full = true;
        return e;
    }
    int n_out = (glu ? nrow16 / 2 : nrow16) + fq * 4;
    e.store = pixel_ok && n_out < cout_store;
    e.opix = op.m;
    if (p.pixel_shuffle) {
        const int cq = p.Cout >> 2;                // channels after the shuffle
        const int sub = n_out / cq;
        e.opix = ((int64_t)op.ob * (2 * p.OH) + 2 * op.ooy + (sub >> 1)) * (2 * p.OW) + 2 * op.oox + (sub & 1);
        n_out -= sub * cq;
    }
    e.n_out = n_out;
    e.full = p.pixel_shuffle || n_out + 3 < cout_store;
    return e;
}

// residual of one tile: raw bits, 8 B (bf16 x 4) or 16 B (f32 x 4)
template <bool F32>
struct EpiRes {
    uint4 v;
};

template <bool F32, class M = EpiGeneric>
__device__ __forceinline__ void conv_epi_load_res(const onedc_conv_params &p, const char *rb, int cout_store,
                                                  const EpiPos &e, EpiRes<F32> &r) {
    r.v = make_uint4(0, 0, 0, 0);
    if (!rb || !e.store) return;
    if (M::fast) {
        if (F32) {
            r.v = *(const uint4 *)((const float *)rb + e.opix * p.r_cs + p.r_co + e.n_out);
        } else {
            const uint2 t = *(const uint2 *)((const a16_t *)rb + e.opix * p.r_cs + p.r_co + e.n_out);
            r.v.x = t.x;
            r.v.y = t.y;
        }
        return;
    }
    if (F32) {
        const float *rp = (const float *)rb + e.opix * p.r_cs + p.r_co + e.n_out;
        if (e.full) {
            r.v = *(const uint4 *)rp;
        } else {
            uint32_t t[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4 && e.n_out + k < cout_store; ++k) t[k] = __float_as_uint(rp[k]);
            r.v = make_uint4(t[0], t[1], t[2], t[3]);
        }
    } else {
        const a16_t *rp = (const a16_t *)rb + e.opix * p.r_cs + p.r_co + e.n_out;
        if (e.full) {
            const uint2 t = *(const uint2 *)rp;
            r.v.x = t.x;
            r.v.y = t.y;
        } else {
            uint32_t t[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4 && e.n_out + k < cout_store; ++k) t[k] = rp[k];
            r.v.x = t[0] | (t[1] << 16);
            r.v.y = t[2] | (t[3] << 16);
        }
    }
}

// bias of the four rows of a tile (and of its gate tile for GLU), loaded once per lane and tile row
struct EpiBias {
    float4 v, g;
};
__device__ __forceinline__ EpiBias conv_epi_bias(const onedc_conv_params &p, bool glu, int nrow16, int fq) {
    EpiBias b;
    b.v = b.g = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nrow = nrow16 + fq * 4;
    if (p.bias && nrow < p.Cout) b.v = *(const float4 *)(p.bias + nrow);          // rows past Cout are tile padding
    if (p.bias && glu && nrow + 16 < p.Cout) b.g = *(const float4 *)(p.bias + nrow + 16);
    return b;
}

// ssum / ssq: running sum and sum of squares of the four stored values (GroupNorm statistics of the
// consumer, see conv_stats_store); untouched when the lane stores nothing.
// to_line (fast 16-bit modes only): the four packed values go to (line0, line1) instead of to memory -- the caller collects a wave's
// tile in LDS and stores whole lines (conv_line_store below)
template <bool F32, class M = EpiGeneric>
__device__ __forceinline__ void conv_epi_finish(const onedc_conv_params &p, char *yb, bool glu, int cout_store,
                                                const f32x4 &accv, const f32x4 &accg, const EpiBias &bias,
                                                const EpiPos &e, const EpiRes<F32> &r, float &ssum, float &ssq,
                                                float &ssum_hi, float &ssq_hi, bool to_line, uint32_t &line0, uint32_t &line1) {
    if (!e.store) return;
    float v[4];
    const float bv[4] = {bias.v.x, bias.v.y, bias.v.z, bias.v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(accv[k], p.alpha, bv[k]);
    const int epi = M::fast ? M::act : p.epi;      // compile-time in the fast modes
    if (M::fast ? M::glu : glu) {
        const float bg[4] = {bias.g.x, bias.g.y, bias.g.z, bias.g.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = __builtin_fmaf(accg[k], p.alpha, bg[k]);
            v[k] = epi == ONEDC_EPI_GLU_LEAKY ? leaky(v[k], 0.1f) + leaky(g, 0.01f) : v[k] * gelu_f(g);
        }
    } else if (epi == ONEDC_EPI_LEAKY) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = leaky(v[k], p.slope);
    } else if (epi == ONEDC_EPI_SILU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = silu_f(v[k]);
    }
    if (F32) {
        v[0] += __uint_as_float(r.v.x);
        v[1] += __uint_as_float(r.v.y);
        v[2] += __uint_as_float(r.v.z);
        v[3] += __uint_as_float(r.v.w);
    } else {
        v[0] += a2f((a16_t)(r.v.x & 0xffff));
        v[1] += a2f((a16_t)(r.v.x >> 16));
        v[2] += a2f((a16_t)(r.v.y & 0xffff));
        v[3] += a2f((a16_t)(r.v.y >> 16));
    }
    if (p.stats) {       // explicit order and explicit fma: identical rounding in every kernel that includes this
        // quads:  ssum = (((ssum + v0) + v1) + v2) + v3, ssq likewise with fma
        // pairs:  (ssum, ssq) take channels 0, 1 and (ssum_hi, ssq_hi) channels 2, 3
        // written without branches on purpose: stores into different accumulators from two branches make the
        // compiler select between their ADDRESSES, which pins all of them in scratch memory
        const bool pairs = p.stats_gran == 2;
        const float s_lo = (ssum + v[0]) + v[1];
        const float q_lo = __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], ssq));
        const float s_in = pairs ? ssum_hi : s_lo, q_in = pairs ? ssq_hi : q_lo;
        const float s_up = (s_in + v[2]) + v[3];
        const float q_up = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], q_in));
        ssum = pairs ? s_lo : s_up;
        ssq = pairs ? q_lo : q_up;
        ssum_hi = pairs ? s_up : ssum_hi;
        ssq_hi = pairs ? q_up : ssq_hi;
    }
    const int64_t yo = e.opix * p.y_cs + p.y_co + e.n_out;
#ifdef EXP_NOSTORE
    if (p.alpha != 12345.f) return;
#endif
    if (M::fast) {
        if (F32) *(float4 *)((float *)yb + yo) = make_float4(v[0], v[1], v[2], v[3]);
        else if (to_line) {
            line0 = pack_a2(v[0], v[1]);
            line1 = pack_a2(v[2], v[3]);
        }
        else *(uint2 *)((a16_t *)yb + yo) = make_uint2(pack_a2(v[0], v[1]), pack_a2(v[2], v[3]));
        return;
    }
    if (p.out_f32) {
        float *yp = (float *)yb + yo;
        if (e.full && ((p.y_cs | p.y_co) & 3) == 0) {
            *(float4 *)yp = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int k = 0; k < 4 && e.n_out + k < cout_store; ++k) yp[k] = v[k];
        }
    } else {
        a16_t *yp = (a16_t *)yb + yo;
        if (p.out_bf16) {                          // bf16 whatever the build's format (flash_attn64's V^T)
            if (e.full && ((p.y_cs | p.y_co) & 3) == 0) {
                *(uint2 *)yp = make_uint2(pack_bf16_2(v[0], v[1]), pack_bf16_2(v[2], v[3]));
            } else {
                for (int k = 0; k < 4 && e.n_out + k < cout_store; ++k) yp[k] = (a16_t)(pack_bf16_2(v[k], 0.f) & 0xffffu);
            }
        } else if (e.full && ((p.y_cs | p.y_co) & 3) == 0) {
            *(uint2 *)yp = make_uint2(pack_a2(v[0], v[1]), pack_a2(v[2], v[3]));
        } else {
            for (int k = 0; k < 4 && e.n_out + k < cout_store; ++k) yp[k] = f2a(v[k]);
        }
    }
}

template <bool F32, class M = EpiGeneric>
__device__ __forceinline__ void conv_epi_finish(const onedc_conv_params &p, char *yb, bool glu, int cout_store,
                                                const f32x4 &accv, const f32x4 &accg, const EpiBias &bias,
                                                const EpiPos &e, const EpiRes<F32> &r, float &ssum, float &ssq,
                                                float &ssum_hi, float &ssq_hi) {
    uint32_t u0 = 0, u1 = 0;
    conv_epi_finish<F32, M>(p, yb, glu, cout_store, accv, accg, bias, e, r, ssum, ssq, ssum_hi, ssq_hi, false, u0, u1);
}

// ---- line stores of the 4-wave GEMM kernels ------------------------------------------------------------------------------
// In accumulator layout a lane owns 4 channels of one pixel: a wave's 16 x 16 tile is one 8-byte store per lane, 16 segments
// of 32 B, and a 64 x 64 wave tile is 16 such instructions -- the store path charges per instruction (DESIGN.md section 7:
// ~290 cycles each when the four waves of a workgroup finish together), which on the UNet's short-K 1x1 layers is a large
// part of a workgroup's life.  Here the wave's 16-bit outputs go through LDS once (wave-private region, written in
// accumulator layout, read back pixel-major) and leave as 16-byte stores: CW stored channels per pixel = CW / 8 lanes per
// pixel, 64 * CW / 512 instructions per wave (8 for 64 channels, 4 for 32) instead of 16 / 8.  Same values, same bits.
// slot8: index of the lane's 8-byte group inside the pixel's CW * 2 bytes.
template <int CW>
__device__ __forceinline__ void conv_line_put(char *my, int px, int slot8, uint32_t v0, uint32_t v1) {
    constexpr int LP = CW / 8;                     // 16-byte chunks (lanes) per pixel
    typedef unsigned line_u32x2 __attribute__((ext_vector_type(2)));
    *(__attribute__((address_space(3))) line_u32x2 *)((__attribute__((address_space(3))) char *)my + px * (CW * 2) +
                                                      ((((slot8 >> 1) ^ px) & (LP - 1)) << 4) + (slot8 & 1) * 8) = line_u32x2{v0, v1};
}
// the wave's 64 pixels m_wave .. m_wave + 63 (those < M) x CW channels starting at element offset y_off of a pixel
template <int CW>
__device__ __forceinline__ void conv_line_store(const char *my, a16_t *yb, int64_t y_cs, int64_t y_off, int m_wave, int M,
                                                int lane) {
    constexpr int LP = CW / 8, PPI = 64 / LP;      // pixels per store instruction
    typedef unsigned line_u32x4 __attribute__((ext_vector_type(4)));
    const int c = lane & (LP - 1);
#pragma unroll
    for (int it = 0; it < LP; ++it) {
        const int px = it * PPI + lane / LP;
        const line_u32x4 o = *(const __attribute__((address_space(3))) line_u32x4 *)((const __attribute__((address_space(3))) char *)my +
                                                                                      px * (CW * 2) + (((c ^ px) & (LP - 1)) << 4));
        const int m = m_wave + px;
        if (m < M) *(line_u32x4 *)(yb + (int64_t)m * y_cs + y_off + c * 8) = o;
    }
}

// ---- fast epilogue ------------------------------------------------------------------------------------------
// The general functions above decide per tile between f32 / bf16 output, pixel shuffle, GLU pairs, partial and
// unaligned channel quads: ~300 instructions per 16x16 tile, microseconds per workgroup.  Almost every launch is
// the plain case -- bf16 output, no shuffle, no GLU, channel quads whole and 8-byte aligned -- for which one tile
// is: 4 fma, activation, residual add, statistics, pack, one 8-byte store at a precomputed address.
// Same operations in the same order as conv_epi_finish: results are bit-identical to the general path.
__host__ __device__ __forceinline__ bool conv_epi_fast_ok(const onedc_conv_params &p, bool glu) {
    return !p.out_f32 && !p.out_bf16 && !p.pixel_shuffle && !glu && ((p.y_cs | p.y_co | p.Cout) & 3) == 0 &&
           (p.epi == ONEDC_EPI_LINEAR || p.epi == ONEDC_EPI_LEAKY || p.epi == ONEDC_EPI_SILU);
}
// STATS: -1 = decided at run time from p.stats / p.stats_gran (branch-free selects: 6 extra instructions per quad),
// 0 = none, 4 = one entry per channel quad, 2 = per channel pair (the compile-time forms do exactly the same operations
// in the same order: bit-identical sums)
template <int ACT, int STATS = -1>   // ONEDC_EPI_LINEAR / _LEAKY / _SILU: the four 16-bit values of one tile, packed
__device__ __forceinline__ uint2 conv_epi_fast_value(const onedc_conv_params &p, const f32x4 &accv, const float4 &bias,
                                                     const uint2 &res, float &ssum, float &ssq, float &ssum_hi,
                                                     float &ssq_hi) {
    float v[4];
    v[0] = __builtin_fmaf(accv[0], p.alpha, bias.x);
    v[1] = __builtin_fmaf(accv[1], p.alpha, bias.y);
    v[2] = __builtin_fmaf(accv[2], p.alpha, bias.z);
    v[3] = __builtin_fmaf(accv[3], p.alpha, bias.w);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ACT == ONEDC_EPI_LEAKY) v[k] = leaky(v[k], p.slope);
        if (ACT == ONEDC_EPI_SILU) v[k] = silu_f(v[k]);
    }
    v[0] += a2f((a16_t)(res.x & 0xffff));
    v[1] += a2f((a16_t)(res.x >> 16));
    v[2] += a2f((a16_t)(res.y & 0xffff));
    v[3] += a2f((a16_t)(res.y >> 16));
    if (STATS == 4) {
        ssum = (((ssum + v[0]) + v[1]) + v[2]) + v[3];
        ssq = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], ssq))));
    } else if (STATS == 2) {
        ssum = (ssum + v[0]) + v[1];
        ssq = __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], ssq));
        ssum_hi = (ssum_hi + v[2]) + v[3];
        ssq_hi = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], ssq_hi));
    } else if (STATS == -1 && p.stats) {
        const bool pairs = p.stats_gran == 2;
        const float s_lo = (ssum + v[0]) + v[1];
        const float q_lo = __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], ssq));
        const float s_in = pairs ? ssum_hi : s_lo, q_in = pairs ? ssq_hi : q_lo;
        const float s_up = (s_in + v[2]) + v[3];
        const float q_up = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], q_in));
        ssum = pairs ? s_lo : s_up;
        ssq = pairs ? q_lo : q_up;
        ssum_hi = pairs ? s_up : ssum_hi;
        ssq_hi = pairs ? q_up : ssq_hi;
    }
    return make_uint2(pack_a2(v[0], v[1]), pack_a2(v[2], v[3]));
}
template <int ACT>
__device__ __forceinline__ void conv_epi_fast(const onedc_conv_params &p, const f32x4 &accv, const float4 &bias,
                                              const uint2 &res, a16_t *yp, float &ssum, float &ssq, float &ssum_hi,
                                              float &ssq_hi) {
    const uint2 o = conv_epi_fast_value<ACT>(p, accv, bias, res, ssum, ssq, ssum_hi, ssq_hi);
#ifdef EXP_NOSTORE
    if (p.alpha != 12345.f) return;
#endif
    *(uint2 *)yp = o;
}

// Run `body(mode)` with the epilogue mode of this launch (body: generic lambda taking an EpiMode value).
#define ONEDC_EPI_DISPATCH(p, allow_fast, body)                                                   \
    do {                                                                                          \
        if ((allow_fast) && conv_epi_mode_fast(p)) {                                              \
            switch ((p).epi) {                                                                    \
                case ONEDC_EPI_LEAKY: body(EpiMode<true, ONEDC_EPI_LEAKY>()); break;              \
                case ONEDC_EPI_SILU: body(EpiMode<true, ONEDC_EPI_SILU>()); break;                \
                case ONEDC_EPI_GLU_LEAKY: body(EpiMode<true, ONEDC_EPI_GLU_LEAKY>()); break;      \
                case ONEDC_EPI_GEGLU: body(EpiMode<true, ONEDC_EPI_GEGLU>()); break;              \
                default: body(EpiMode<true, ONEDC_EPI_LINEAR>()); break;                          \
            }                                                                                     \
        } else {                                                                                  \
            body(EpiGeneric());                                                                   \
        }                                                                                         \
    } while (0)

// GroupNorm statistics in the producer's epilogue.  A wave covers 64 consecutive output pixels and,
// per 16-row accumulator tile, 4 x 4 consecutive channels ("quads", one per lane quarter fq).  After
// the lane-local sums over the wave's pixel tiles, the 16 lanes of a quarter are reduced with four
// butterfly steps and lane (fq, 0) stores (sum, sum of squares) to
//     stats[(m_wave / 64) * (y_cs / 4) + (y_co + n_out) / 4]          (float2)
// Fixed order, no atomics: every (64-pixel block, quad) has exactly one writer.  onedc_gn_from_stats
// folds the quads of a group and the blocks of an image.  Requires OH*OW % 64 == 0 (a 64-pixel block
// never straddles two images), y_co % 4 == 0, stored channels % 4 == 0, no pixel shuffle.
// row: index of the wave's 64-pixel block among all blocks of the batch (any numbering that keeps the
// blocks of one image contiguous: the finalize only sums over an image's rows)
// stats_gran = 2 (buffers whose group size is not a multiple of 4, e.g. 320 or 960 channels): one entry per channel
// PAIR, [rows][y_cs / 2]; the (_hi) sums are the lane's second pair.  Only the 4-wave and 128x256 kernels carry it.
__device__ __forceinline__ void conv_stats_store_row(const onedc_conv_params &p, float ssum, float ssq, int row,
                                                     int n_out, int cout_store, int lane, float ssum_hi = 0.f,
                                                     float ssq_hi = 0.f) {
    const bool pairs = p.stats_gran == 2;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        ssum += __shfl_xor(ssum, o);
        ssq += __shfl_xor(ssq, o);
    }
    if (pairs) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            ssum_hi += __shfl_xor(ssum_hi, o);
            ssq_hi += __shfl_xor(ssq_hi, o);
        }
    }
    if ((lane & 15) == 0 && n_out < cout_store) {
        if (pairs) {
            float2 *dst = (float2 *)p.stats + (int64_t)row * (p.y_cs >> 1) + ((p.y_co + n_out) >> 1);
            dst[0] = make_float2(ssum, ssq);
            dst[1] = make_float2(ssum_hi, ssq_hi);
        } else {
            float2 *dst = (float2 *)p.stats + (int64_t)row * (p.y_cs >> 2) + ((p.y_co + n_out) >> 2);
            *dst = make_float2(ssum, ssq);
        }
    }
}
__device__ __forceinline__ void conv_stats_store(const onedc_conv_params &p, float ssum, float ssq, int m_wave,
                                                 int n_out, int cout_store, int lane, int M, float ssum_hi = 0.f,
                                                 float ssq_hi = 0.f) {
    // M % 64 == 0: a block is whole or absent
    conv_stats_store_row(p, ssum, ssq, m_wave >> 6, m_wave < M ? n_out : cout_store, cout_store, lane, ssum_hi, ssq_hi);
}
