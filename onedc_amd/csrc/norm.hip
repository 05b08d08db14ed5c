// norm.hip — GroupNorm statistics / apply, LayerNorm, row softmax.  All HBM-bound streaming
// kernels: 16-byte (8 x bf16) accesses per lane, fp32 math, fixed-order reductions (no atomics) so
// results do not depend on batch size or scheduling.
#include <type_traits>

#include "common.h"

namespace {

constexpr int GN_GROUPS = 32;
// pixels per partial-sum chunk: a function of the map size ONLY (never of B: an image's sums are added in the same order alone
// and inside a batch).  Round 4: small maps get small chunks -- the UNet's 12x12 level was ONE block per image whose threads walked
// 144 pixels one dependent load after the other (51 us per launch, 27 launches per decode); now 9 blocks per image.
__host__ __device__ constexpr int gn_ppc(int HW) { return HW <= 1024 ? 16 : (HW <= 16384 ? 128 : 1024); }

__device__ __forceinline__ void unpack8(const uint4 &v, float *f) {
    f[0] = a2f((a16_t)(v.x & 0xffff)); f[1] = a2f((a16_t)(v.x >> 16));
    f[2] = a2f((a16_t)(v.y & 0xffff)); f[3] = a2f((a16_t)(v.y >> 16));
    f[4] = a2f((a16_t)(v.z & 0xffff)); f[5] = a2f((a16_t)(v.z >> 16));
    f[6] = a2f((a16_t)(v.w & 0xffff)); f[7] = a2f((a16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float *f) {
    return make_uint4(pack_a2(f[0], f[1]), pack_a2(f[2], f[3]), pack_a2(f[4], f[5]), pack_a2(f[6], f[7]));
}
__device__ __forceinline__ uint4 pack8_bf16(const float *f) {      // bf16 whatever the build's format
    return make_uint4(pack_bf16_2(f[0], f[1]), pack_bf16_2(f[2], f[3]), pack_bf16_2(f[4], f[5]), pack_bf16_2(f[6], f[7]));
}

// stage 1: block (chunk, b) sums x and x^2 per group over gn_ppc(HW) pixels.
// threads = CC * PL with CC = C/8 channel chunks and PL pixel lanes.
__global__ void gn_partial_kernel(const a16_t *__restrict__ x, float *__restrict__ part, int HW, int C,
                                  int cs, int co, int CC, int PL, int nchunk) {
    extern __shared__ float lds[];  // [PL][C][2]
    const int t = threadIdx.x;
    const int cc = t % CC, pl = t / CC;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int p0 = chunk * gn_ppc(HW);
    const int p1 = min(p0 + gn_ppc(HW), HW);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    const a16_t *base = x + ((int64_t)b * HW) * cs + co + cc * 8;
    for (int p = p0 + pl; p < p1; p += PL) {
        const uint4 v = *(const uint4 *)(base + (int64_t)p * cs);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[e] += f[e];
            q[e] += f[e] * f[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        lds[(pl * C + cc * 8 + e) * 2 + 0] = s[e];
        lds[(pl * C + cc * 8 + e) * 2 + 1] = q[e];
    }
    __syncthreads();
    if (t < GN_GROUPS) {
        const int cpg = C / GN_GROUPS;
        float ss = 0.f, qq = 0.f;
        for (int l = 0; l < PL; ++l)
            for (int c = t * cpg; c < (t + 1) * cpg; ++c) {
                ss += lds[(l * C + c) * 2 + 0];
                qq += lds[(l * C + c) * 2 + 1];
            }
        float *o = part + (((int64_t)b * nchunk + chunk) * GN_GROUPS + t) * 2;
        o[0] = ss;
        o[1] = qq;
    }
}

// stage 2: one block per image: reduce chunks (fp64), emit per-channel (scale, shift):
//   y = x * a + bb with a = rstd*gamma, bb = beta - mean*rstd*gamma
__global__ void gn_finalize_kernel(const float *__restrict__ part, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, float2 *__restrict__ ab, int HW, int C,
                                   int nchunk, float eps) {
    __shared__ float mean_s[GN_GROUPS], rstd_s[GN_GROUPS];
    const int b = blockIdx.x, t = threadIdx.x;
    const int cpg = C / GN_GROUPS;
    if (t < GN_GROUPS) {
        double ss = 0.0, qq = 0.0;
        for (int k = 0; k < nchunk; ++k) {
            const float *o = part + (((int64_t)b * nchunk + k) * GN_GROUPS + t) * 2;
            ss += (double)o[0];
            qq += (double)o[1];
        }
        const double n = (double)HW * cpg;
        const double mean = ss / n;
        double var = qq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[t] = (float)mean;
        rstd_s[t] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = t; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float a = rstd_s[g] * gamma[c];
        ab[(int64_t)b * C + c] = make_float2(a, beta[c] - mean_s[g] * a);
    }
}

// Small maps (HW <= 256: the 12 x 12 level of the UNet and of the codec, whose 144 pixels are not whole 64-pixel blocks, so no
// producer leaves partial sums for them): stages 1 and 2 in ONE launch.  Block (group, image); thread t takes the group's
// 16-byte units t, t + 256, ... (unit = 8 channels of one pixel) in fp64, then the fixed-shape tree of gn_from_stats_kernel.
// Round 4: 27 GroupNorms of a decode run here (tools/gn_pass_census.py); two dependent launches of ~15 + 7 us became one.
__global__ void gn_small_kernel(const a16_t *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                float2 *__restrict__ ab, int HW, int C, int cs, int co, float eps) {
    __shared__ double red[2][256];
    const int g = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int cpg = C / GN_GROUPS, upp = cpg >> 3;            // 16-byte units per pixel of this group
    const a16_t *base = x + ((int64_t)b * HW) * cs + co + g * cpg;
    double ss = 0.0, qq = 0.0;
    const int n = HW * upp;
    for (int u = t; u < n; u += 256) {
        const int p = u / upp, k = u - p * upp;
        float f[8];
        unpack8(*(const uint4 *)(base + (int64_t)p * cs + k * 8), f);
        float s8 = 0.f, q8 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s8 += f[e];
            q8 = __builtin_fmaf(f[e], f[e], q8);
        }
        ss += (double)s8;
        qq += (double)q8;
    }
    red[0][t] = ss;
    red[1][t] = qq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            red[0][t] += red[0][t + o];
            red[1][t] += red[1][t + o];
        }
        __syncthreads();
    }
    const double cnt = (double)HW * cpg;
    const double mean = red[0][0] / cnt;
    double var = red[1][0] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (t < cpg) {
        const int c = g * cpg + t;
        const float a = rstd * gamma[c];
        ab[(int64_t)b * C + c] = make_float2(a, beta[c] - (float)mean * a);
    }
}

// stage 2 when the producing conv left partial sums (onedc_conv_params.stats): block (group, image)
// folds the group's quads over the image's 64-pixel blocks.  Fixed order: thread t sums entries
// t, t+256, ... in fp64, then a fixed-shape tree over the 256 threads.
__global__ void gn_from_stats_kernel(const float2 *__restrict__ stats, const float *__restrict__ gamma,
                                     const float *__restrict__ beta, float2 *__restrict__ ab, int HW, int C,
                                     int cs, int co, float eps, int gran) {
    __shared__ double red[2][256];
    const int g = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int cpg = C / GN_GROUPS, qpg = cpg / gran;          // gran: channels per statistics entry (4 or 2)
    const int rows = HW >> 6, qs = cs / gran;
    const float2 *base = stats + (int64_t)b * rows * qs + (co + g * cpg) / gran;
    double ss = 0.0, qq = 0.0;
    const int n = rows * qpg;
    for (int e = t; e < n; e += 256) {
        const int r = e / qpg, q = e - r * qpg;
        const float2 v = base[(int64_t)r * qs + q];
        ss += (double)v.x;
        qq += (double)v.y;
    }
    red[0][t] = ss;
    red[1][t] = qq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            red[0][t] += red[0][t + o];
            red[1][t] += red[1][t + o];
        }
        __syncthreads();
    }
    const double cnt = (double)HW * cpg;
    const double mean = red[0][0] / cnt;
    double var = red[1][0] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (t < cpg) {
        const int c = g * cpg + t;
        const float a = rstd * gamma[c];
        ab[(int64_t)b * C + c] = make_float2(a, beta[c] - (float)mean * a);
    }
}

// y = act(x * a + bb).  Block (pixel chunk, image); thread = (channel chunk of 8, pixel lane): the
// thread's 8 (a, bb) pairs are loaded once and stay in registers while it walks its pixels, so the
// loop is one 16-byte load, 8 fma (+ 8 exp / rcp for SiLU) and one 16-byte store, no index division.
// ppc: pixels per block, 256 on the big maps; round 5: fewer on maps that would give less than two blocks per CU -- 24x24 and
// 12x12 maps of a batch of 16 were 48 / 16 blocks of serial 256-pixel walks: 56 us for 47 MB (0.8 TB/s), 28 us for 12 MB
constexpr int SSA_PPC = 256;
// OBF: y is stored as bf16 whatever the build's format (round 5: the consumer is a conv that multiplies bf16 operands,
// onedc_conv_params.x_bf16; the values before the rounding are the same)
template <bool OBF>
__global__ void scale_shift_act_kernel(const a16_t *__restrict__ x, const float2 *__restrict__ ab,
                                       a16_t *__restrict__ y, int HW, int C, int CC, int PL, int x_cs, int x_co,
                                       int y_cs, int y_co, int silu, int ppc) {
    const int t = threadIdx.x;
    const int cc = t % CC, pl = t / CC;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * ppc;
    const int p1 = min(p0 + ppc, HW);
    float a[8], bb[8];
    const float4 *abp = (const float4 *)(ab + (int64_t)b * C + cc * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float4 v = abp[e];
        a[2 * e] = v.x;
        bb[2 * e] = v.y;
        a[2 * e + 1] = v.z;
        bb[2 * e + 1] = v.w;
    }
    const a16_t *xp = x + ((int64_t)b * HW) * x_cs + x_co + cc * 8;
    a16_t *yp = y + ((int64_t)b * HW) * y_cs + y_co + cc * 8;
    // four pixels per iteration, all loads issued before the first use: the loop is bound by memory latency
    auto run = [&](auto silu_c) {
        constexpr bool SILU = decltype(silu_c)::value;
        int p = p0 + pl;
        for (; p + 3 * PL < p1; p += 4 * PL) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const uint4 *)(xp + (int64_t)(p + u * PL) * x_cs);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f[e] = __builtin_fmaf(f[e], a[e], bb[e]);
                    if (SILU) f[e] = silu_f(f[e]);
                }
                *(uint4 *)(yp + (int64_t)(p + u * PL) * y_cs) = OBF ? pack8_bf16(f) : pack8(f);
            }
        }
        for (; p < p1; p += PL) {
            float f[8];
            unpack8(*(const uint4 *)(xp + (int64_t)p * x_cs), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f[e] = __builtin_fmaf(f[e], a[e], bb[e]);
                if (SILU) f[e] = silu_f(f[e]);
            }
            *(uint4 *)(yp + (int64_t)p * y_cs) = OBF ? pack8_bf16(f) : pack8(f);
        }
    };
    if (silu) run(std::true_type());
    else run(std::false_type());
}

// LayerNorm: one wave per row, row cached in registers (C <= 8*64*NCH)
template <int NCH>
__global__ void layernorm_kernel(const a16_t *__restrict__ x, const float *__restrict__ gamma,
                                 const float *__restrict__ beta, a16_t *__restrict__ y, int64_t rows, int C,
                                 float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int CC = C >> 3;
    float f[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < CC) {
            unpack8(*(const uint4 *)(x + row * C + cc * 8), f[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[i][e];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < CC) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = f[i][e] - mean;
                q += d * d;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / C + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < CC) {
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o8[e] = (f[i][e] - mean) * rstd * gamma[cc * 8 + e] + beta[cc * 8 + e];
            *(uint4 *)(y + row * C + cc * 8) = pack8(o8);
        }
    }
}

// softmax over f32 rows (n <= 64*NV), bf16 out with zero padding up to n_pad
template <int NV>
__global__ void softmax_rows_kernel(const float *__restrict__ s, a16_t *__restrict__ p, int64_t rows, int n,
                                    int n_pad) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NV];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < n ? s[row * n + c] : -3.0e38f;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < n ? __expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < n_pad) p[row * n_pad + c] = f2a(v[i] * inv);
    }
}

// any row length: one wave per row, three sweeps over the row (max, sum of exponentials, write); the second and third
// sweep hit L2.  Same operations per element as the register-resident kernel above (max-subtracted __expf, one
// reciprocal), the wave reduction order differs only in how many elements a lane folds before the shuffles.
__global__ void softmax_rows_any_kernel(const float *__restrict__ s, a16_t *__restrict__ p, int64_t rows, int n,
                                        int n_pad) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *sr = s + row * n;
    float mx = -3.0e38f;
    for (int c = lane; c < n; c += 64) mx = fmaxf(mx, sr[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int c = lane; c < n; c += 64) sum += __expf(sr[c] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
    a16_t *pr = p + row * n_pad;
    for (int c = lane; c < n_pad; c += 64) pr[c] = c < n ? f2a(__expf(sr[c] - mx) * inv) : (a16_t)0;
}

}  // namespace

extern "C" {

int onedc_gn_chunks(int HW) { return (HW + gn_ppc(HW) - 1) / gn_ppc(HW); }

int onedc_gn_stats(const void *x, const float *gamma, const float *beta, float *partial, float *ab, int B,
                   int HW, int C, int x_cs, int x_co, float eps, void *stream) {
    if (!x || !gamma || !beta || !partial || !ab) return -1;
    if (C % 32 || C % 8 || (x_cs & 7) || (x_co & 7) || C > 4096) return -2;
    if (HW <= 256 && (C / GN_GROUPS) % 8 == 0 && C / GN_GROUPS <= 256) {       // small maps: one launch (a function of the shape only)
        hipLaunchKernelGGL(gn_small_kernel, dim3(GN_GROUPS, B), dim3(256), 0, as_stream(stream), (const a16_t *)x, gamma, beta,
                           (float2 *)ab, HW, C, x_cs, x_co, eps);
        ONEDC_LAUNCH_CHECK();
        return 0;
    }
    const int CC = C / 8;
    const int PL = CC >= 256 ? 1 : 256 / CC;
    const int threads = CC * PL;
    if (threads > 1024) return -3;
    const int nchunk = onedc_gn_chunks(HW);
    const size_t lds = (size_t)PL * C * 2 * sizeof(float);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(threads), lds, as_stream(stream),
                       (const a16_t *)x, partial, HW, C, x_cs, x_co, CC, PL, nchunk);
    ONEDC_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, as_stream(stream), partial, gamma, beta,
                       (float2 *)ab, HW, C, nchunk, eps);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

int onedc_gn_from_stats(const float *stats, const float *gamma, const float *beta, float *ab, int B, int HW,
                        int C, int cs, int co, float eps, int gran, void *stream) {
    if (!stats || !gamma || !beta || !ab || (gran != 2 && gran != 4)) return -1;
    if (C % 32 || ((C / 32) % gran) || (co % gran) || (cs % gran) || (HW & 63) || C / 32 > 256) return -2;
    hipLaunchKernelGGL(gn_from_stats_kernel, dim3(GN_GROUPS, B), dim3(256), 0, as_stream(stream),
                       (const float2 *)stats, gamma, beta, (float2 *)ab, HW, C, cs, co, eps, gran);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

int onedc_scale_shift_act_fmt(const void *x, const float *ab, void *y, int B, int HW, int C, int x_cs, int x_co,
                              int y_cs, int y_co, int silu, int out_bf16, void *stream) {
    if (!x || !ab || !y) return -1;
    if ((C & 7) || (x_cs & 7) || (x_co & 7) || (y_cs & 7) || (y_co & 7)) return -2;
    const int CC = C / 8;
    const int PL = CC >= 256 ? 1 : 256 / CC;
    const int threads = CC * PL;
    if (threads > 1024) return -3;
    int ppc = SSA_PPC;                // by map size and batch only; elementwise: the values do not depend on the split
#ifndef EXP_R5_BASELINE
    while (ppc > 8 && ppc > PL && (int64_t)((HW + ppc - 1) / ppc) * B < 512) ppc >>= 1;
#endif
    if (out_bf16)
        hipLaunchKernelGGL(scale_shift_act_kernel<true>, dim3((HW + ppc - 1) / ppc, B), dim3(threads), 0,
                           as_stream(stream), (const a16_t *)x, (const float2 *)ab, (a16_t *)y, HW, C, CC, PL, x_cs,
                           x_co, y_cs, y_co, silu, ppc);
    else
        hipLaunchKernelGGL(scale_shift_act_kernel<false>, dim3((HW + ppc - 1) / ppc, B), dim3(threads), 0,
                           as_stream(stream), (const a16_t *)x, (const float2 *)ab, (a16_t *)y, HW, C, CC, PL, x_cs,
                           x_co, y_cs, y_co, silu, ppc);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

int onedc_scale_shift_act(const void *x, const float *ab, void *y, int B, int HW, int C, int x_cs, int x_co,
                          int y_cs, int y_co, int silu, void *stream) {
    return onedc_scale_shift_act_fmt(x, ab, y, B, HW, C, x_cs, x_co, y_cs, y_co, silu, 0, stream);
}

int onedc_layernorm(const void *x, const float *gamma, const float *beta, void *y, int64_t rows, int C,
                    float eps, void *stream) {
    if (!x || !gamma || !beta || !y || (C & 7) || C > 1536) return -1;
    const int blocks = (int)((rows + 3) / 4);
    if (C <= 512)
        hipLaunchKernelGGL(layernorm_kernel<1>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           (const a16_t *)x, gamma, beta, (a16_t *)y, rows, C, eps);
    else if (C <= 1024)
        hipLaunchKernelGGL(layernorm_kernel<2>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           (const a16_t *)x, gamma, beta, (a16_t *)y, rows, C, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<3>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           (const a16_t *)x, gamma, beta, (a16_t *)y, rows, C, eps);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

int onedc_softmax_rows(const float *s, void *p, int64_t rows, int n, int n_pad, void *stream) {
    if (!s || !p || n < 1 || n_pad < n || rows < 0 || (rows + 3) / 4 > 0x7fffffff) return -1;
    const int blocks = (int)((rows + 3) / 4);
    if (n_pad > 2560)              // larger pictures (1024x1024 and up on the encode side): streaming kernel, any n
        hipLaunchKernelGGL(softmax_rows_any_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), s, (a16_t *)p, rows, n,
                           n_pad);
    else if (n_pad <= 256)
        hipLaunchKernelGGL(softmax_rows_kernel<4>, dim3(blocks), dim3(256), 0, as_stream(stream), s, (a16_t *)p,
                           rows, n, n_pad);
    else if (n_pad <= 1024)        // the analysis transform's single-head attention at 1/16 resolution (512x512 pictures)
        hipLaunchKernelGGL(softmax_rows_kernel<16>, dim3(blocks), dim3(256), 0, as_stream(stream), s, (a16_t *)p,
                           rows, n, n_pad);
    else                           // ... and 2304 tokens at 768x768
        hipLaunchKernelGGL(softmax_rows_kernel<40>, dim3(blocks), dim3(256), 0, as_stream(stream), s, (a16_t *)p,
                           rows, n, n_pad);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
