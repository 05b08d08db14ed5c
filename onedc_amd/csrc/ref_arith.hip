// ref_arith.hip — the prior networks in the REFERENCE'S CPU arithmetic, on the GPU.
//
// A y stream written by the reference (torch-CPU fp32: src/modules/entropy/compression_model.py:303-407 through
// entropy_models.py:355-362) can only be decoded by a prior that takes every table-index decision the writer took; the
// decisions truncate (ln sigma - ln 0.11)/step, so the decoder's sigma has to be the writer's sigma to the last bit.
// torch-CPU's convolutions are not "an f32 dot product in some order": measured (tests/test_ref_arith.py; DESIGN.md section
// 5), every one of them is a fixed arrangement of fused-multiply-add chains --
//   1x1, oneDNN jit_1x1:avx512_core (more than one torch thread and more than 20 480 input elements):
//        input channels in blocks of R; the first block is ONE fma chain that starts from the bias, every later block a chain
//        that starts from 0 and is added to the sum so far; R = 80, 128, 256 or 384 depending on the shape
//   1x1, native (one thread, or a small input: aten _slow_conv2d_forward -> MKL sgemm, beta = 1 on a bias-filled output):
//        fma chains from 0 over blocks of 256 input channels; the bias is added to the first block's sum, later blocks to that
//   depthwise 3x3, oneDNN jit_dw:avx512_core: one fma chain from the bias over the taps, row-major, padding taps skipped
//        (csrc/elementwise.hip: dwconv3x3_f32_kernel is exactly this already)
//   dense 3x3, oneDNN jit:avx512_core: per block of 16 input channels one fma chain from 0 over (ky, kx, channel in block),
//        padding taps skipped; the block sums are added one after the other to the bias
//   dense 3x3, oneDNN jit:avx2 (a writer without AVX-512, or DNNL_MAX_CPU_ISA=AVX2): ONE fma chain from the bias over
//        (8-channel block, ky, kx, channel in block); its 1x1 kernel is the jit_1x1 family above with blocks of 128
//   dense 3x3 on maps of <= 20 480 elements (pictures under ~0.16 MP), native: im2col + MKL sgemm, beta = 1 on a bias-filled
//        output; k = (channel, ky, kx).  K is cut into T equal parts (one per group of MKL threads), every part blocks of `blk`
//        (chains from 0, added in order); the bias joins part 0; parts [0, s) are added in order, parts [s, T) are added in
//        order, then the two sums.  Measured on the Intel build host: 1, 4, 6 threads T = 1, blk = 384; 2 and 8 threads T = 4,
//        s = 2; 3 threads T = 3, s = 1 -- small pictures are as thread-dependent as it gets
// -- and v_fma_f32 is the same IEEE operation as vfmadd231ps.  The kernels below evaluate those chains, one thread per
// output element (consecutive lanes = consecutive output channels: the transposed weights are read coalesced, the
// input value of a pixel is a broadcast), parameterised by (R, bias first / last); the host finds the parameters of each layer
// shape by trial against the local torch (onedc_amd/ref_arith.py: calibrate) and refuses shapes it cannot reproduce.
// This is a COMPATIBILITY path (IntraNoAR(prior_arith="torch_cpu")): ~10x slower than the exact-f32 MFMA path, used to decode
// streams a reference wrote; streams of this package's own encoder keep the MFMA prior on both sides.
#include "common.h"

namespace {

__device__ __forceinline__ float ref_leaky(float v, float s) { return v >= 0.f ? v : v * s; }

// ACT: 0 none, 1 LeakyReLU(slope), 2 ConvFFN3's gate: LeakyReLU(a, slope) + LeakyReLU(b, slope2), a = row n, b = row n + OC / 2
// One thread: one output channel of P consecutive pixels -- P independent chains share every weight load (the weights are the
// traffic: a [C][OC] f32 matrix read once per pixel GROUP from L2 instead of once per pixel), and the pixel group is a function of
// blockIdx alone, so the input values are wave-uniform (scalar loads, SGPR operands of v_fma_f32).  The order of the fma chain
// of every single output is what the header says; P changes nothing in it.
template <int ACT, int P>
__global__ void ref_conv1x1_kernel(const float *__restrict__ x, int x_cs, int x_co, const float *__restrict__ wT,
                                   const float *__restrict__ bias, const float *__restrict__ res, int r_cs, int r_co,
                                   float *__restrict__ y, int y_cs, int y_co, int64_t pixels, int C, int OC, int blk,
                                   int bias_first, float slope, float slope2, int shuffle, int H, int W) {
    const int NO = ACT == 2 ? OC >> 1 : OC;                       // channels this launch produces
    const int n = blockIdx.y * blockDim.x + threadIdx.x;
    if (n >= NO) return;
    const int64_t groups = (pixels + P - 1) / P;
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int64_t pix0 = grp * P;
        const float *xp[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int64_t pix = pix0 + p < pixels ? pix0 + p : pixels - 1;      // the tail repeats the last pixel, not stored
            xp[p] = x + pix * x_cs + x_co;
        }
        float v[ACT == 2 ? 2 : 1][P];
#pragma unroll
        for (int h = 0; h < (ACT == 2 ? 2 : 1); ++h) {
            const int oc = n + h * NO;
            const float b = bias ? bias[oc] : 0.f;
            float tot[P];
            for (int i0 = 0; i0 < C; i0 += blk) {
                float acc[P];
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p] = (bias_first && i0 == 0) ? b : 0.f;
                const int i1 = i0 + blk < C ? i0 + blk : C;
                const float *wp = wT + (int64_t)i0 * OC + oc;
                int ic = i0;
                for (; ic + 8 <= i1; ic += 8, wp += 8 * (int64_t)OC) {             // eight weight loads in flight
                    float w[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) w[k] = wp[(int64_t)k * OC];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
#pragma unroll
                        for (int p = 0; p < P; ++p) acc[p] = __builtin_fmaf(xp[p][ic + k], w[k], acc[p]);
                }
                for (; ic < i1; ++ic, wp += OC) {
                    const float w = wp[0];
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p] = __builtin_fmaf(xp[p][ic], w, acc[p]);
                }
                // native path (gemm with beta = 1 on a bias-filled output): the bias joins the FIRST block's sum, later
                // blocks are added to that -- (chain_0 + b) + chain_1 -- as measured for K = 512 on one thread
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (i0 == 0) tot[p] = bias_first ? acc[p] : acc[p] + b;
                    else tot[p] = tot[p] + acc[p];
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p) v[h][p] = tot[p];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int64_t pix = pix0 + p;
            if (pix >= pixels) break;
            float out = v[0][p];
            if (ACT == 1) out = ref_leaky(out, slope);
            if (ACT == 2) out = ref_leaky(v[0][p], slope) + ref_leaky(v[ACT == 2 ? 1 : 0][p], slope2);
            int64_t dst;
            if (shuffle) {                                         // F.pixel_shuffle(., 2): channel c * 4 + dy * 2 + dx -> (2 y + dy, 2 x + dx, c)
                const int xx = (int)(pix % W);
                const int yy = (int)((pix / W) % H);
                const int64_t bb = pix / ((int64_t)W * H);
                const int c = n >> 2, dy = (n >> 1) & 1, dx = n & 1;
                dst = ((bb * 2 * H + 2 * yy + dy) * (2 * W) + 2 * xx + dx) * y_cs + y_co + c;
            } else {
                dst = pix * y_cs + y_co + n;
                if (res) out = out + res[pix * r_cs + r_co + n];
            }
            y[dst] = out;
        }
    }
}

// dense 3x3, pad 1, oneDNN's direct kernels.  w3T [9][C][OC].
//   avx512_core (ICB = 16, !ONE): total = bias; per 16 input channels a chain from 0 over (ky, kx, channel); total += chain
//   avx2        (ICB = 8, ONE):   ONE chain that starts from the bias and runs over (8-channel block, ky, kx, channel)
template <int ICB, bool ONE>
__global__ void ref_conv3x3_kernel(const float *__restrict__ x, int x_cs, int x_co, const float *__restrict__ w3T,
                                   const float *__restrict__ bias, const float *__restrict__ res, int r_cs, int r_co,
                                   float *__restrict__ y, int y_cs, int y_co, int B, int H, int W, int C, int OC, int leaky_on,
                                   float slope) {
    const int64_t total = (int64_t)B * H * W * OC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / OC;
        const int oc = (int)(i - pix * OC);
        const int xx = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        const int64_t b = pix / ((int64_t)W * H);
        float tot = bias ? bias[oc] : 0.f;
        float acc = ONE ? tot : 0.f;
        for (int icb = 0; icb < C; icb += ICB) {
            if (!ONE) acc = 0.f;
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = yy + ky - 1;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = xx + kx - 1;
                    if (ix < 0 || ix >= W) continue;
                    const float *xp = x + ((b * H + iy) * W + ix) * x_cs + x_co + icb;
                    const float *wp = w3T + ((int64_t)(ky * 3 + kx) * C + icb) * OC + oc;
#pragma unroll
                    for (int ii = 0; ii < ICB; ++ii) acc = __builtin_fmaf(xp[ii], wp[(int64_t)ii * OC], acc);
                }
            }
            if (!ONE) tot = tot + acc;
        }
        float out = ONE ? acc : tot;
        if (leaky_on) out = ref_leaky(out, slope);
        if (res) out = out + res[pix * r_cs + r_co + oc];
        y[pix * y_cs + y_co + oc] = out;
    }
}

// dense 3x3, pad 1, in aten's im2col order k = c * 9 + ky * 3 + kx: wKT [9 C][OC]; K in T parts of K / T, blocks of blk inside
// a part, parts [0, s) and [s, T) summed in order, bias joined to part 0 (see the header).  Small maps only: one thread per output.
__global__ void ref_conv3x3_im2col_kernel(const float *__restrict__ x, int x_cs, int x_co, const float *__restrict__ wKT,
                                          const float *__restrict__ bias, const float *__restrict__ res, int r_cs, int r_co,
                                          float *__restrict__ y, int y_cs, int y_co, int B, int H, int W, int C, int OC, int T,
                                          int blk, int split, int leaky_on, float slope) {
    const int64_t total = (int64_t)B * H * W * OC;
    const int K = 9 * C, per = K / T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / OC;
        const int oc = (int)(i - pix * OC);
        const int xx = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        const int64_t b = pix / ((int64_t)W * H);
        float side[2] = {0.f, 0.f};
        for (int pi = 0; pi < T; ++pi) {
            float tot = 0.f;
            for (int k0 = pi * per; k0 < (pi + 1) * per; k0 += blk) {
                const int k1 = k0 + blk < (pi + 1) * per ? k0 + blk : (pi + 1) * per;
                float acc = 0.f;
                int c = k0 / 9, t = k0 - 9 * c;
                for (int k = k0; k < k1; ++k) {
                    const int ky = t / 3, kx = t - 3 * ky;
                    const int iy = yy + ky - 1, ix = xx + kx - 1;
                    const float xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((b * H + iy) * W + ix) * x_cs + x_co + c] : 0.f;
                    acc = __builtin_fmaf(xv, wKT[(int64_t)k * OC + oc], acc);       // the zero of a padding tap is multiplied, as im2col does
                    if (++t == 9) t = 0, ++c;
                }
                // the bias joins the first block's sum (the gemm's beta = 1 on the bias-filled output), as in the 1x1 native path
                tot = k0 == pi * per ? (pi == 0 ? acc + (bias ? bias[oc] : 0.f) : acc) : tot + acc;
            }
            const int sd = pi >= split;
            side[sd] = (pi == 0 || pi == split) ? tot : side[sd] + tot;
        }
        float out = split < T ? side[0] + side[1] : side[0];
        if (leaky_on) out = ref_leaky(out, slope);
        if (res) out = out + res[pix * r_cs + r_co + oc];
        y[pix * y_cs + y_co + oc] = out;
    }
}

__device__ __constant__ int REF_PHASE[4][4] = {{0, 1, 2, 3}, {3, 2, 1, 0}, {2, 3, 0, 1}, {1, 0, 3, 2}};

// build_indexes with the WRITER'S logarithm: index = number of thresholds <= sigma, thresholds[i] = the smallest f32 whose
// torch index is >= i + 1 (found on the host by bisection over torch.log itself, onedc_amd/ref_arith.py)
__global__ void ref_prior_indexes_kernel(const float *__restrict__ params, const float *__restrict__ thr, int n_thr,
                                         int16_t *__restrict__ idx, int B, int H, int W, int C, int step) {
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
        const int ph = ((y & 1) << 1) | (x & 1);
        int g = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (REF_PHASE[step][k] == ph) g = k;
        const float s = params[(((int64_t)b * H + y) * W + x) * (2 * C) + g * q + c4];
        int lo = 0, hi = n_thr;                         // first threshold > s
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (thr[mid] <= s) lo = mid + 1;
            else hi = mid;
        }
        idx[i] = (int16_t)lo;
    }
}

inline int ref_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 262144 ? 262144 : g));
}

}  // namespace

extern "C" {

int onedc_ref_conv1x1(const float *x, int x_cs, int x_co, const float *wT, const float *bias, const float *res, int r_cs,
                      int r_co, float *y, int y_cs, int y_co, int64_t pixels, int C, int OC, int blk, int bias_first,
                      int act, float slope, float slope2, int shuffle, int H, int W, void *stream) {
    if (!x || !wT || !y || C < 1 || OC < 1 || blk < 1 || pixels < 0 || act < 0 || act > 2) return -1;
    if ((act == 2 && (OC & 1)) || (shuffle && ((OC & 3) || act == 2 || res || H < 1 || W < 1))) return -2;
    if (pixels == 0) return 0;
    const int NO = act == 2 ? OC / 2 : OC;
    hipStream_t st = as_stream(stream);
    const int threads = NO >= 256 ? 256 : ((NO + 63) / 64) * 64;
    const int tiles = (NO + threads - 1) / threads;
    // 8 pixels per thread where that still leaves every CU several waves; small maps keep the chip busy with 2
    const bool wide = (pixels / 8) * ((NO + 63) / 64) >= 2048;
    const int64_t groups = wide ? (pixels + 7) / 8 : (pixels + 1) / 2;
    const dim3 grid((unsigned)(groups > 1048576 ? 1048576 : groups), (unsigned)tiles);
#define REF_L(A)                                                                                                          \
    do {                                                                                                                  \
        if (wide)                                                                                                         \
            hipLaunchKernelGGL((ref_conv1x1_kernel<A, 8>), grid, dim3(threads), 0, st, x, x_cs, x_co, wT, bias, res, r_cs, r_co, \
                               y, y_cs, y_co, pixels, C, OC, blk, bias_first, slope, slope2, shuffle, H, W);               \
        else                                                                                                              \
            hipLaunchKernelGGL((ref_conv1x1_kernel<A, 2>), grid, dim3(threads), 0, st, x, x_cs, x_co, wT, bias, res, r_cs, r_co, \
                               y, y_cs, y_co, pixels, C, OC, blk, bias_first, slope, slope2, shuffle, H, W);               \
    } while (0)
    if (act == 0) REF_L(0);
    else if (act == 1) REF_L(1);
    else REF_L(2);
#undef REF_L
    ONEDC_LAUNCH_CHECK();
    return 0;
}

int onedc_ref_conv3x3(const float *x, int x_cs, int x_co, const float *w3T, const float *bias, const float *res, int r_cs,
                      int r_co, float *y, int y_cs, int y_co, int B, int H, int W, int C, int OC, int leaky_on, float slope,
                      int icb, int one_chain, void *stream) {
    if (!x || !w3T || !y || OC < 1 || B < 1 || H < 1 || W < 1) return -1;
    if (!((icb == 16 && !one_chain) || (icb == 8 && one_chain)) || (C % icb)) return -2;
    const int64_t n = (int64_t)B * H * W * OC;
    if (icb == 16)
        hipLaunchKernelGGL((ref_conv3x3_kernel<16, false>), dim3(ref_grid(n)), dim3(256), 0, as_stream(stream), x, x_cs, x_co,
                           w3T, bias, res, r_cs, r_co, y, y_cs, y_co, B, H, W, C, OC, leaky_on, slope);
    else
        hipLaunchKernelGGL((ref_conv3x3_kernel<8, true>), dim3(ref_grid(n)), dim3(256), 0, as_stream(stream), x, x_cs, x_co,
                           w3T, bias, res, r_cs, r_co, y, y_cs, y_co, B, H, W, C, OC, leaky_on, slope);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

int onedc_ref_conv3x3_im2col(const float *x, int x_cs, int x_co, const float *wKT, const float *bias, const float *res, int r_cs,
                             int r_co, float *y, int y_cs, int y_co, int B, int H, int W, int C, int OC, int parts, int blk,
                             int split, int leaky_on, float slope, void *stream) {
    if (!x || !wKT || !y || C < 1 || OC < 1 || B < 1 || H < 1 || W < 1 || parts < 1 || blk < 1) return -1;
    if ((9 * C) % parts || split < 1 || split > parts) return -2;
    const int64_t n = (int64_t)B * H * W * OC;
    hipLaunchKernelGGL(ref_conv3x3_im2col_kernel, dim3(ref_grid(n)), dim3(256), 0, as_stream(stream), x, x_cs, x_co, wKT, bias,
                       res, r_cs, r_co, y, y_cs, y_co, B, H, W, C, OC, parts, blk, split, leaky_on, slope);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

int onedc_ref_prior_indexes(const float *params, const float *thresholds, int n_thresholds, int16_t *indexes, int B, int H,
                            int W, int C, int step, void *stream) {
    if (!params || !thresholds || !indexes || n_thresholds < 1 || (C & 3) || step < 0 || step > 3) return -1;
    const int64_t n = (int64_t)B * (C / 4) * H * W;
    hipLaunchKernelGGL(ref_prior_indexes_kernel, dim3(ref_grid(n)), dim3(256), 0, as_stream(stream), params, thresholds,
                       n_thresholds, indexes, B, H, W, C, step);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
