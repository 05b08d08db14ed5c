// gemm_conv_ups.hip — nearest x2 upsample followed by a 3x3 convolution (stride 1, pad 1), decomposed by output
// phase.
//
// Output pixel (2y + py, 2x + px) of conv3x3(upsample2(in)) reads input pixels ((2y + py + ky - 1) >> 1, ...):
// for py = 0 that is row y - 1 (ky = 0) and row y (ky = 1, 2); for py = 1 row y (ky = 0, 1) and row y + 1 (ky = 2).
// So each of the four phases (py, px) is a 2 x 2 convolution of the LOW-resolution input with weights that are sums
// of the 3x3 weights:
//     W'[py,px][ty,tx] = sum_{ky in S(py,ty)} sum_{kx in S(px,tx)} W[ky,kx],   S(0,0) = {0}, S(0,1) = {1,2},
//                                                                              S(1,0) = {0,1}, S(1,1) = {2}
// 4 taps instead of 9: 2.25x fewer multiply-adds than running the 3x3 kernel over the upsampled grid (which is what
// `up = 1` does, in every conv kernel of this library).  The host packs W' as [4 phases x C rows][4 taps x Cin]
// (onedc_amd/pack.py: upsample_phase_matrix); the sums are formed in f32 and rounded to bf16 once.
//
// Kernel structure = the 3x3 window kernel (gemm_conv_win.hip): a workgroup owns a 16 x 16 patch of LOW-resolution
// pixels x 128 rows of the phase-packed weights (all of one phase), keeps the patch's 18 x 18 input window of the
// current 64-channel chunk in LDS (double buffered), streams the weights (ring of 4 tiles = the 4 taps of a chunk),
// 8 waves 2 x 4, two 16-MFMA phases per K-tile, the wave rows one barrier apart, statically unrolled taps -- the
// body is instantiated per phase so that the window rows a tap reads are compile-time constants.  Outputs go through
// LDS and leave as whole 128-byte lines to pixel (2 y + py, 2 x + px).
#include <type_traits>

#include "conv_common.h"

namespace {

constexpr int UPS_A = 16384;                 // one weight tile: 128 rows x 128 B
constexpr int UPS_NA = 4;                    // weight ring: the slot of a tap is the tap
constexpr int UPS_WROWS = 328;               // window rows: 18 x 18 = 324, padded to 41 blocks of 8
constexpr int UPS_W = UPS_WROWS * 128;       // 41984 B
constexpr int UPS_LDS = UPS_NA * UPS_A + 2 * UPS_W;   // 149504 B

#define UPS_WAIT_LGKMCNT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void ups_wait_vmcnt(int n) {      // n is wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    }
}

template <int PY, int PX>
__device__ __forceinline__ void ups_body(const onedc_conv_params &p, char *smem, int n0, int pt, int img, int pin, int y0,
                                         int x0, int tiles_n, int tiles_pi, int Ktot) {
    constexpr int WW = 18;
    constexpr int NWB = 41;                         // 8-row DMA blocks per window
    constexpr int MAXWB = 6;                        // blocks per wave
    char *const lds_a = smem;
    char *const lds_w = smem + UPS_NA * UPS_A;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- DMA geometry: buffer descriptors, per-lane byte offsets fixed for the whole kernel ------------------
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ lrow;            // swizzle on the source side (row & 7 == lrow for every block)
    const int w_rows = tiles_n * 128;
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w, (unsigned)((int64_t)w_rows * Ktot * 2));
    const unsigned wvoff = (unsigned)((int64_t)(n0 + wave * 8 + lrow) * Ktot * 2) + chunk * 16;
    const unsigned wrow64 = (unsigned)(64 * Ktot * 2);
    // K-tile (chunk ci, tap k): weights are [row][tap][cin] -> byte offset (k * Cin + ci * 64) * 2 inside a row
    auto stage_A = [&](int k, int ci) {
        char *dst = lds_a + k * UPS_A + wave * 1024;
        const unsigned so = (unsigned)(k * p.Cin + ci * 64) * 2;
        glds16_buf(w_rsrc, wvoff, so, dst);
        glds16_buf(w_rsrc, wvoff, so + wrow64, dst + 8192);
    };
    // prologue, part 1 (same order as gemm_conv_win.hip): the first two weight tiles leave before the window offsets
    // are decoded, the window before the fragment addresses -- address arithmetic runs under DMA latency
    stage_A(0, 0);
    stage_A(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    const int64_t img_bytes = (int64_t)p.IH * p.IW * p.x_cs * 2;
    const unsigned x_bytes = img_bytes < 0xfffff000ll ? (unsigned)img_bytes : 0xfffff000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc((const char *)p.x + img * img_bytes, x_bytes);
    const int nwb = (NWB - wave + 7) >> 3;          // blocks this wave stages per window (wave-uniform): 6 or 5
    unsigned woff[MAXWB];
#pragma unroll
    for (int k = 0; k < MAXWB; ++k) {
        const int row = (wave + 8 * k) * 8 + lrow;
        const int wy = row / WW, wx = row - wy * WW;
        const int iy = y0 - 1 + wy, ix = x0 - 1 + wx;
        const bool ok = row < WW * WW && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        woff[k] = ok ? (unsigned)((iy * p.IW + ix) * p.x_cs + p.x_co) * 2u + chunk * 16 : x_bytes;
    }
    const int nc = p.Cin >> 6;                       // 64-channel chunks
    auto stage_W = [&](unsigned off, int k, int ci, int wbuf) {      // block k of this wave, window of chunk ci
        glds16_buf(x_rsrc, off, (unsigned)ci * 128, lds_w + wbuf * UPS_W + (wave + 8 * k) * 1024);
    };
#pragma unroll
    for (int k = 0; k < MAXWB; ++k)
        if (k < nwb) stage_W(woff[k], k, 0, 0);
    __builtin_amdgcn_sched_barrier(0);


    // ---- accumulators: acc[i][b][j]: rows wr*64 + i*16.., low-res row 4 wc + 2 b + j, column frow -----------
    f32x4 acc[4][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    int fao0 = wr * 64 * 128 + frow * 128 + ((fq ^ (lane & 7)) << 4);
    int fao1 = wr * 64 * 128 + frow * 128 + (((4 + fq) ^ (lane & 7)) << 4);
    // tap k = (ty, tx) of phase (PY, PX) reads window row (4 wc + bj + PY + ty) * 18 + frow + PX + tx:
    // 5 x 2 LDS addresses per lane (r = bj + ty), computed once.  k 32..63 is the address ^ 64.
    int wrd[5][2];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
            const int w = (4 * wc + r + PY) * WW + frow + PX + tx;
            wrd[r][tx] = UPS_NA * UPS_A + w * 128 + ((fq ^ (w & 7)) << 4);
        }
    a16x8 fa[4][2], fb[2][2];
    // addresses and zeroed accumulators exist before the wait (empty asm: the optimiser cannot sink them behind it)
    asm volatile("" : "+v"(fao0), "+v"(fao1));
#pragma unroll
    for (int r = 0; r < 5; ++r) asm volatile("" : "+v"(wrd[r][0]), "+v"(wrd[r][1]));
    const char *const fa0 = lds_a + fao0, *const fa1 = lds_a + fao1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][b][j]));

    // ---- prologue, part 2: window of chunk 0 and the weights of K-tiles 0 and 1 are on their way -----------------------
    __builtin_amdgcn_sched_barrier(0);
    ups_wait_vmcnt(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();       // the two wave rows run one barrier apart

    // ---- main loop: chunk ci, four statically unrolled taps; one K-tile = 2 phases of 16 MFMAs -----------------
    //   phase 0: read the 8 weight fragments and pixel rows 0, 1; DMA the weights of K-tile t + 2
    //   phase 1: read pixel rows 2, 3; DMA two blocks of the NEXT chunk's window (taps 0..2); wait until the weights
    //            of K-tile t + 1 (and at tap 3 the whole next window) have landed
    for (int ci = 0; ci < nc; ++ci) {
        const bool more = ci + 1 < nc;
        int c_prev = 0;                              // window blocks issued in the previous tap
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ty = k >> 1, tx = k & 1;
            const char *const a_slot0 = fa0 + k * UPS_A, *const a_slot1 = fa1 + k * UPS_A;
            // phase 0
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ro = wrd[j + ty][tx];
                fb[j][0] = *(const a16x8 *)(smem + ro);
                fb[j][1] = *(const a16x8 *)(smem + (ro ^ 64));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i][0] = *(const a16x8 *)(a_slot0 + i * 2048);
                fa[i][1] = *(const a16x8 *)(a_slot1 + i * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool a_next = k < 2 || more;       // weights of K-tile t + 2
            if (k < 2) stage_A(k + 2, ci);
            else if (more) stage_A(k - 2, ci + 1);
            __builtin_amdgcn_s_barrier();
            UPS_WAIT_LGKMCNT(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][0][j] = ONEDC_MFMA16(fa[i][ks], fb[j][ks], acc[i][0][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // phase 1
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ro = wrd[2 + j + ty][tx];
                fb[j][0] = *(const a16x8 *)(smem + ro);
                fb[j][1] = *(const a16x8 *)(smem + (ro ^ 64));
            }
            __builtin_amdgcn_sched_barrier(0);
            // window blocks 2 k, 2 k + 1 of the next chunk ride on taps 0..2
            int c_now = 0;
            if (k < 3 && more) {
                if (2 * k < nwb) {
                    stage_W(woff[2 * k < MAXWB ? 2 * k : 0], 2 * k, ci + 1, (ci & 1) ^ 1);
                    ++c_now;
                }
                if (2 * k + 1 < nwb) {
                    stage_W(woff[2 * k + 1 < MAXWB ? 2 * k + 1 : 0], 2 * k + 1, ci + 1, (ci & 1) ^ 1);
                    ++c_now;
                }
            }
            // DMA issued after the weights of K-tile t + 1: the previous tap's window blocks, the weights of K-tile
            // t + 2, this tap's window blocks
            // (at tap 3 the previous tap's blocks complete the next window: they must have landed too)
            ups_wait_vmcnt((a_next ? 2 : 0) + c_now + (k == 3 ? 0 : c_prev));
            c_prev = c_now;
            __builtin_amdgcn_s_barrier();
            UPS_WAIT_LGKMCNT(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][1][j] = ONEDC_MFMA16(fa[i][ks], fb[j][ks], acc[i][1][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        // the next chunk reads the other window buffer
        const int delta = (ci & 1) ? -UPS_W : UPS_W;
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) wrd[r][tx] += delta;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();

    // ---- epilogue: bias, activation, statistics; rows of 64 channels through LDS to pixel (2 y + PY, 2 x + PX) ----
    const int C = p.Cout;                            // real output channels; packed rows = 4 C
    const int phase = PY * 2 + PX;
    const int nw = n0 - phase * C + wr * 64;         // first of the wave's 64 output channels
    float ssum[4], ssq[4], ssum_hi[4], ssq_hi[4];
    float4 bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ssum[i] = ssq[i] = ssum_hi[i] = ssq_hi[i] = 0.f;
        bias[i] = p.bias ? *(const float4 *)(p.bias + nw + i * 16 + fq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    typedef __attribute__((address_space(3))) char lds_char;
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    lds_char *const my = (lds_char *)smem + wave * 8192;
    auto finish = [&](auto act, auto stats_mode) {
        constexpr int ST = decltype(stats_mode)::value;
        constexpr int ACT = decltype(act)::value;
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
            const int px = bj * 16 + frow;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint2 o = conv_epi_fast_value<ACT, ST>(p, acc[i][bj >> 1][bj & 1], bias[i], make_uint2(0u, 0u), ssum[i],
                                                         ssq[i], ssum_hi[i], ssq_hi[i]);
                const int s8 = i * 4 + fq;             // 8-byte slot of the pixel's 128-byte row
                *(__attribute__((address_space(3))) u32x2_t *)(my + px * 128 + (((s8 >> 1) ^ (px & 7)) << 4) + (s8 & 1) * 8) =
                    u32x2_t{o.x, o.y};
            }
        }
    };
    auto with_stats = [&](auto act) {              // compile-time statistics mode, as in gemm_conv_win.hip
        if (!p.stats) finish(act, std::integral_constant<int, 0>());
        else if (p.stats_gran == 2) finish(act, std::integral_constant<int, 2>());
        else finish(act, std::integral_constant<int, 4>());
    };
    if (p.epi == ONEDC_EPI_SILU) with_stats(std::integral_constant<int, ONEDC_EPI_SILU>());
    else if (p.epi == ONEDC_EPI_LEAKY) with_stats(std::integral_constant<int, ONEDC_EPI_LEAKY>());
    else with_stats(std::integral_constant<int, ONEDC_EPI_LINEAR>());
    {
        const int c = lane & 7;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int px = it * 8 + (lane >> 3);       // low-res pixel (row 4 wc + (px >> 4), column px & 15) of the patch
            const u32x4_t o = *(const __attribute__((address_space(3))) u32x4_t *)(my + px * 128 + ((c ^ (px & 7)) << 4));
            const int oy = 2 * (y0 + 4 * wc + (px >> 4)) + PY, ox = 2 * (x0 + (px & 15)) + PX;
            const int64_t pix = ((int64_t)img * p.OH + oy) * p.OW + ox;
            *(u32x4_t *)((a16_t *)p.y + pix * p.y_cs + p.y_co + nw + c * 8) = o;
        }
    }
    if (p.stats) {       // one statistics row per (image, phase, patch, wave column): 64 output pixels
        const int row = ((img * 4 + phase) * tiles_pi + pin) * 4 + wc;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            conv_stats_store_row(p, ssum[i], ssq[i], row, nw + i * 16 + fq * 4, C, lane, ssum_hi[i], ssq_hi[i]);
    }
    (void)pt;
}

__global__ __launch_bounds__(512, 2) void conv_ups_kernel(const onedc_conv_params p, int tiles_n, int tiles_x, int tiles_y,
                                                          int Ktot, OnedcFastDiv dv_n, OnedcFastDiv dv_pi, OnedcFastDiv dv_x,
                                                          OnedcFastDiv dv_c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // channel tile fastest, then x, y, image; a channel tile belongs to one phase (C % 128 == 0)
    const int tiles_pi = tiles_x * tiles_y;
    const int logical = xcd_logical_tile(blockIdx.x, tiles_n * tiles_pi * p.B);
    const int pt = dv_n.div(logical);
    const int n0 = (logical - pt * tiles_n) * 128;
    const int img = dv_pi.div(pt), pin = pt - img * tiles_pi;
    const int ty = dv_x.div(pin), tx = pin - ty * tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;            // low-resolution origin of the patch
    const int phase = dv_c.div(n0);
    switch (phase) {
        case 0: ups_body<0, 0>(p, smem, n0, pt, img, pin, y0, x0, tiles_n, tiles_pi, Ktot); break;
        case 1: ups_body<0, 1>(p, smem, n0, pt, img, pin, y0, x0, tiles_n, tiles_pi, Ktot); break;
        case 2: ups_body<1, 0>(p, smem, n0, pt, img, pin, y0, x0, tiles_n, tiles_pi, Ktot); break;
        default: ups_body<1, 1>(p, smem, n0, pt, img, pin, y0, x0, tiles_n, tiles_pi, Ktot); break;
    }
}

}  // namespace

// up == 2: x is the low-resolution input, w the phase-packed matrix [4 Cout rows][4 taps x Cin], y the x2 output.
bool onedc_conv_ups_ok(const onedc_conv_params &p) {
    // (out_bf16: only the 4-wave form stores it -- conv_choose refuses the 8-wave one -- so the fast-epilogue test runs without it)
    onedc_conv_params q = p;
    q.out_bf16 = 0;
    return p.up == 2 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && conv_epi_fast_ok(q, false) &&
           !p.res && p.batch <= 1 && !p.in_f32 && (p.Cout & 127) == 0 && (p.Cin & 63) == 0 && (p.IH & 15) == 0 &&
           (p.IW & 15) == 0 && p.OH == 2 * p.IH && p.OW == 2 * p.IW && ((p.y_cs | p.y_co) & 7) == 0 &&
           (int64_t)p.IH * p.IW * p.x_cs * 2 < 0xfffff000ll;
}

int onedc_conv_ups_launch(const onedc_conv_params &p, hipStream_t st) {
    static OnedcLdsAttrOnce once;
    once.ensure({{(const void *)conv_ups_kernel, UPS_LDS}});
    const int tiles_n = 4 * p.Cout / 128;
    const int tiles_x = p.IW / 16, tiles_y = p.IH / 16;
    const int Ktot = 4 * p.Cin;
    const int64_t blocks = (int64_t)tiles_n * tiles_x * tiles_y * p.B;
    if (blocks > 0x7fffffff) return -1;
    hipLaunchKernelGGL(conv_ups_kernel, dim3((unsigned)blocks), dim3(512), UPS_LDS, st, p, tiles_n, tiles_x, tiles_y, Ktot,
                       OnedcFastDiv((uint32_t)tiles_n), OnedcFastDiv((uint32_t)(tiles_x * tiles_y)), OnedcFastDiv((uint32_t)tiles_x),
                       OnedcFastDiv((uint32_t)p.Cout));
    ONEDC_LAUNCH_CHECK();
    return 0;
}
