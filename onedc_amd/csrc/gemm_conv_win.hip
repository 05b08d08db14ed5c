// gemm_conv_win.hip — 3x3 convolutions (stride 1, pad 1, optionally behind a fused nearest x2
// upsample) as an implicit GEMM that stages every input pixel ONCE per 64-channel chunk.
//
// The generic kernels (gemm_conv.hip, gemm_conv256.hip) DMA a fresh 256-pixel B tile for each of the nine
// taps although the taps of a 3x3 window overlap almost completely.  Here a workgroup owns a 16 x 16
// patch of output pixels (x 128 output channels) and keeps the patch's input WINDOW in LDS:
// 18 x 18 pixels (10 x 10 when the x2 upsample is fused: output (y, x) reads input ((y+ky-1)>>1, ...)),
// 128 bytes = 64 channels per pixel row, zero rows outside the picture.  The nine taps are nine
// different row offsets into that window, so per 64-channel chunk the B operand costs 41 (13) KB of
// LDS-DMA instead of 9 x 32 KB, and there is no per-tap address work at all.  Only the weights (16 KB per
// K-tile) stream per tap.  LDS-DMA bytes per MFMA drop 2.3x against the 128x256 kernel, which is what
// bounds these layers (VAE up-blocks at 768x768: 64 B/clk/CU of L1->LDS bandwidth).
//
//   waves        2 (channels) x 4 (pixel rows): wave (wr, wc) owns channels wr*64.. and output rows
//                4 wc .. 4 wc + 3 of the patch: 4 x 4 accumulator tiles, 16 MFMAs per phase, 2 phases per
//                K-tile (K-tile = one tap of one 64-channel chunk)
//   LDS          ring of 4 weight tiles (16 KB each, staged 3 K-tiles ahead) + 2 windows (next chunk's
//                window is streamed in 8-row pieces during taps 1..6 of the current chunk)
//   waits        one counted s_waitcnt vmcnt(N) per K-tile (N = DMA instructions this wave issued after the
//                next K-tile's weights), raw s_barrier, the two wave rows one barrier apart
//   fragments    ds_read_b128; 16-byte chunk c of LDS row r sits at c ^ (r & 7): conflict-free for ANY 16
//                consecutive rows, which is what a tap-shifted window read is
// STATUS: auto-dispatched by onedc_conv_gemm for every 3x3 / stride 1 / pad 1 layer it can take (conv_choose in
// gemm_conv.hip); tests/test_gpu_kernels.py::test_conv_win3x3_*.  Measured numbers: DESIGN.md section 3 / 4.
// Same operand layout, accumulation order per output (tap-major inside a chunk, chunks outer -- see note
// below), epilogues and statistics as the other kernels.
// Note: the k order is (chunk, tap) here against (tap, chunk) in the generic kernels, so results agree to
// fp32 summation-order rounding, not bit for bit; each kernel on its own is deterministic and
// batch-invariant.
#include <type_traits>

#include "conv_common.h"

namespace {

constexpr int WIN_A = 16384;                 // one weight tile: 128 rows x 128 B
constexpr int WIN_NA = 3;                    // weight ring depth (9 taps: the slot of a tap is tap % 3)
constexpr int WIN_ROWS = 328;                // window rows, padded to 41 blocks of 8
constexpr int WIN_W = WIN_ROWS * 128;        // 41984 B
constexpr int WIN_LDS = WIN_NA * WIN_A + 2 * WIN_W;   // 133120 B

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKMCNT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

template <bool UP, bool NORM>
__global__ __launch_bounds__(512, 2) void conv_win_kernel(const onedc_conv_params p, int tiles_n, int tiles_x,
                                                          int tiles_y, int Ktot, OnedcFastDiv dv_n, OnedcFastDiv dv_pi,
                                                          OnedcFastDiv dv_x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef EXP_WIN_TIMELINE
    unsigned long long tl[6];
    tl[0] = __builtin_amdgcn_s_memtime();
#define TL(i) tl[i] = __builtin_amdgcn_s_memtime()
#else
#define TL(i)
#endif
    char *const lds_a = smem;
    char *const lds_w = smem + WIN_NA * WIN_A;
    constexpr int WW = UP ? 10 : 18;               // window width (and height)
    constexpr int NWB = (WW * WW + 7) / 8;          // 8-row DMA blocks per window: 41 / 13
    constexpr int MAXWB = (NWB + 7) / 8;            // blocks per wave: 6 / 2

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- patch of this workgroup: channel tile fastest, then x, y, image --------------------------------
    const int tiles_pi = tiles_x * tiles_y;
    const int logical = xcd_logical_tile(blockIdx.x, tiles_n * tiles_pi * p.B);
    const int pt = dv_n.div(logical);               // patch index over all images
    const int n0 = (logical - pt * tiles_n) * 128;
    const int img = dv_pi.div(pt), pin = pt - img * tiles_pi;
    // Patch order inside an image: bands of four patch rows, inside a band column by column from top to bottom.  The
    // patch that follows on this XCD is then the one BELOW (its window shares two of eighteen rows with this one's and
    // finds them in L2), left / right neighbours are four patches apart; in row-major order the vertical neighbour came
    // a whole patch row later (48 patches x 82 KB at 768x768: the XCD's L2 had turned over).  `dv_x` divides by 4 * tiles_x.
    int ty, tx;
    {
        const int band = dv_x.div(pin), inb = pin - band * 4 * tiles_x;
        const int left = tiles_y - band * 4, rows = left < 4 ? left : 4;      // patch rows of this band
        int r;
        if (rows == 4) {
            tx = inb >> 2;
            r = inb & 3;
        } else if (rows == 2) {
            tx = inb >> 1;
            r = inb & 1;
        } else if (rows == 1) {
            tx = inb;
            r = 0;
        } else {                                                               // 3: exact for inb < 32768
            tx = (inb * 21846) >> 16;
            r = inb - 3 * tx;
        }
        ty = band * 4 + r;
    }
    const int y0 = ty * 16, x0 = tx * 16;           // output origin of the patch

    // ---- DMA geometry: buffer descriptors, per-lane byte offsets fixed for the whole kernel ------------------
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ lrow;            // swizzle on the source side (row & 7 == lrow for every block)
    // weights: 16 blocks of 8 rows per K-tile, two per wave: rows wave * 8 + lrow and + 64
    const int w_rows = tiles_n * 128;               // packed weight rows are padded to the channel tile
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w, (unsigned)((int64_t)w_rows * Ktot * 2));
    const unsigned wvoff = (unsigned)((int64_t)(n0 + wave * 8 + lrow) * Ktot * 2) + chunk * 16;
    const unsigned wrow64 = (unsigned)(64 * Ktot * 2);
    // K-tile (chunk ci, tap): weights are [row][tap][cin] -> byte offset (tap * Cin + ci * 64) * 2 inside a row
    auto stage_A = [&](int tap, int ci, int slot) {
        char *dst = lds_a + slot * WIN_A + wave * 1024;
        const unsigned so = (unsigned)(tap * p.Cin + ci * 64) * 2;
        glds16_buf(w_rsrc, wvoff, so, dst);
        glds16_buf(w_rsrc, wvoff, so + wrow64, dst + 8192);
    };
    // The first two weight tiles need nothing but the channel tile: they leave before the window offsets (six row
    // decodes per lane) are worked out, so that arithmetic runs under their latency instead of in front of it.
    stage_A(0, 0, 0);
    stage_A(1, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    // window: block k of this wave covers window rows (wave + 8 k) * 8 + lrow; rows outside the picture (and the
    // padding rows of the last block) carry an out-of-range offset: the descriptor's range check returns zeros
    const int64_t img_bytes = (int64_t)p.IH * p.IW * p.x_cs * 2;
    const unsigned x_bytes = img_bytes < 0xfffff000ll ? (unsigned)img_bytes : 0xfffff000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc((const char *)p.x + img * img_bytes, x_bytes);
    const int nwb = (NWB - wave + 7) >> 3;          // blocks this wave stages per window (wave-uniform)
    unsigned woff[MAXWB];
    {
        const int wy0 = UP ? (y0 >> 1) - 1 : y0 - 1, wx0 = UP ? (x0 >> 1) - 1 : x0 - 1;
#pragma unroll
        for (int k = 0; k < MAXWB; ++k) {
            const int row = (wave + 8 * k) * 8 + lrow;
            const int wy = row / WW, wx = row - wy * WW;
            const int iy = wy0 + wy, ix = wx0 + wx;
            const bool ok = row < WW * WW && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            woff[k] = ok ? (unsigned)((iy * p.IW + ix) * p.x_cs + p.x_co) * 2u + chunk * 16 : x_bytes;
        }
    }
    const int nc = p.Cin >> 6;                       // 64-channel chunks
    auto stage_W = [&](unsigned off, int k, int ci, int wbuf) {      // block k of this wave, window of chunk ci
        glds16_buf(x_rsrc, off, (unsigned)ci * 128, lds_w + wbuf * WIN_W + (wave + 8 * k) * 1024);
    };
    // prologue, part 1: the first window leaves as soon as its offsets exist (HBM latency is the long pole of the
    // prologue); the fragment addresses and the accumulator set-up below run while it is in flight
    const char *const xb_img = (const char *)p.x + img * img_bytes;
    auto load_raw = [&](unsigned off, int ci) {      // 8 channels of one window row (zeros for halo / padding rows)
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (off != x_bytes) r = *(const uint4 *)(xb_img + off + ci * 128);
        return r;
    };
    uint4 raw0[MAXWB];                               // NORM: the first window's blocks, raw, on their way to registers
    if (!NORM) {
#pragma unroll
        for (int k = 0; k < MAXWB; ++k)
            if (k < nwb) stage_W(woff[k], k, 0, 0);
    } else {
#pragma unroll
        for (int k = 0; k < MAXWB; ++k) raw0[k] = k < nwb ? load_raw(woff[k], 0) : make_uint4(0u, 0u, 0u, 0u);
    }
    __builtin_amdgcn_sched_barrier(0);

    // NORM: the input is the raw tensor in front of a GroupNorm (+SiLU); the window is staged through registers and
    // normalised on the way: v = silu(a[c] * x + b[c]) with (a, b) per (image, channel) from onedc_gn_from_stats, the
    // same operations in the same order as onedc_scale_shift_act, so the conv sees bit-identical operands -- and the
    // normalised tensor never exists in HBM (it was 10 % of the device time to write and re-read it).  (a, b) of the
    // image sit in LDS behind the window buffers; halo rows stay zero (the conv pads the NORMALISED tensor).
    float2 *const lds_ab = (float2 *)(smem + WIN_LDS);
    float na[8], nb[8];                              // (a, b) of this lane's 8 channels of the chunk being staged
    auto load_ab = [&](int ci) {
        const float4 *q = (const float4 *)(lds_ab + ci * 64 + chunk * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 v = q[e];
            na[2 * e] = v.x;
            nb[2 * e] = v.y;
            na[2 * e + 1] = v.z;
            nb[2 * e + 1] = v.w;
        }
    };
    // two of the four 32-bit words (4 of the 8 channels) of a block: the main loop spreads a block over two phases so
    // that the work of a phase fits under the other wave row's MFMA segment
    auto norm_half = [&](const uint4 &raw, unsigned off, int half, uint32_t &o0, uint32_t &o1) {
        const uint32_t w2[2] = {half ? raw.z : raw.x, half ? raw.w : raw.y};
        uint32_t o2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 4 * half + 2 * e;
            float lo = __builtin_fmaf(a2f((a16_t)(w2[e] & 0xffff)), na[c], nb[c]);
            float hi = __builtin_fmaf(a2f((a16_t)(w2[e] >> 16)), na[c + 1], nb[c + 1]);
#ifndef EXP_NORM_NOSILU
            if (p.gn_silu) {
                lo = silu_f(lo);
                hi = silu_f(hi);
            }
#endif
            o2[e] = off != x_bytes ? pack_a2(lo, hi) : 0u;
        }
        o0 = o2[0];
        o1 = o2[1];
    };
    auto norm_store = [&](const uint4 &raw, unsigned off, int k, int wbuf) {
        uint32_t o[4];
        norm_half(raw, off, 0, o[0], o[1]);
        norm_half(raw, off, 1, o[2], o[3]);
        *(uint4 *)(lds_w + wbuf * WIN_W + (wave + 8 * k) * 1024 + lane * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    };


    // ---- accumulators: acc[i][b][j]: channels wr*64 + i*16.., output row 4 wc + 2 b + j, column frow ------
    f32x4 acc[4][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    // weight fragments: row 16 i + frow of the wave's 64 channels, 16-byte chunk fq (k 0..31) / 4 + fq (k 32..63)
    int fao0 = wr * 64 * 128 + frow * 128 + ((fq ^ (lane & 7)) << 4);
    int fao1 = wr * 64 * 128 + frow * 128 + (((4 + fq) ^ (lane & 7)) << 4);
    // pixel fragments: output (row 4 wc + bj, column frow) under tap (ky, kx) reads window row
    //   (4 wc + bj + ky) * 18 + frow + kx        (fused x2 upsample: ((4 wc + bj + ky + 1) >> 1) * 10 + ((frow + kx + 1) >> 1))
    // which depends on bj + ky only: 6 x 3 LDS addresses per lane, computed once.  k 32..63 is the address ^ 64.
    int wrd[6][3];                                   // byte offsets into smem
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int w = UP ? ((4 * wc + r + 1) >> 1) * WW + ((frow + kx + 1) >> 1) : (4 * wc + r) * WW + frow + kx;
            wrd[r][kx] = WIN_NA * WIN_A + w * 128 + ((fq ^ (w & 7)) << 4);
        }
    a16x8 fa[4][2], fb[2][2];
    // these addresses are needed from the first K-tile on: make them exist HERE (an empty asm the optimiser cannot look
    // through), or the compiler sinks their arithmetic behind the prologue's wait, onto the critical path
#define PIN_V(x) asm volatile("" : "+v"(x))
    PIN_V(fao0);
    PIN_V(fao1);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) PIN_V(wrd[r][kx]);
    const char *const fa0 = lds_a + fao0, *const fa1 = lds_a + fao1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][b][j]));     // zeroed before the wait, not after

    // ---- prologue: window of chunk 0, weights of K-tiles 0 and 1 -------------------------------------------------
    if (NORM) {
        for (int c = tid; c < p.Cin; c += 512) lds_ab[c] = ((const float2 *)p.gn_ab)[(int64_t)img * p.Cin + c];
        __syncthreads();                             // (a, b) of the image are in LDS
        load_ab(0);
#pragma unroll
        for (int k = 0; k < MAXWB; ++k)
            if (k < nwb) norm_store(raw0[k], woff[k], k, 0);
    }
    // everything above (fragment addresses included) is address arithmetic that must not sink behind the wait: the
    // first bytes take a microsecond to arrive, which is when that arithmetic is free
    __builtin_amdgcn_sched_barrier(0);
    WAIT_VMCNT(0);
    WAIT_LGKMCNT(0);
    __builtin_amdgcn_s_barrier();
    TL(1);
    if (wr == 1) __builtin_amdgcn_s_barrier();       // the two wave rows run one barrier apart

    // ---- main loop: chunk ci, nine statically unrolled taps; one K-tile = 2 phases of 16 MFMAs ------------------
    //   phase 0: read the 8 weight fragments and pixel rows 0, 1; DMA the weights of K-tile t + 2
    //   phase 1: read pixel rows 2, 3; DMA one block of the NEXT chunk's window (taps 0 .. nwb - 1); wait until the
    //            weights of K-tile t + 1 (and at tap 8 the whole next window) have landed
    // NORM: window block k of the next chunk is fetched at tap k (phase 1), its first half is normalised at tap k + 2
    // (phase 1) and the second half at tap k + 3 (phase 0), where the 16 bytes go to LDS -- BEFORE that tap's weight
    // DMA is issued: the compiler drains every DMA in flight in front of an LDS write it can see, and at that point
    // only DMA issued a whole K-tile earlier is left.  Three taps between fetch and use cover the HBM latency (the raw
    // tensor streams from HBM; with one tap the load latency was exposed on every block: +0.65 us per block).
    uint4 nraw[MAXWB];
#pragma unroll
    for (int k = 0; k < MAXWB; ++k) nraw[k] = make_uint4(0u, 0u, 0u, 0u);
    uint32_t nout0 = 0, nout1 = 0;                   // first two normalised words of the block being converted
    for (int ci = 0; ci < nc; ++ci) {
        const bool more = ci + 1 < nc;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const char *const a_slot0 = fa0 + (tap % 3) * WIN_A, *const a_slot1 = fa1 + (tap % 3) * WIN_A;
            // phase 0
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ro = wrd[j + ky][kx];
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
            if (NORM && tap >= 3 && tap - 3 < MAXWB) {   // second half of block tap - 3, and its LDS write
                if (tap - 3 < nwb && more) {
                    const int kb = tap - 3 < MAXWB ? tap - 3 : 0;
                    uint32_t o2, o3;
                    norm_half(nraw[kb], woff[kb], 1, o2, o3);
                    // (inline asm: in front of an LDS write it can see, the compiler waits for vmcnt(0) -- it cannot
                    // prove that the LDS-DMA in flight does not alias it -- which here would also wait for the register
                    // loads issued one and two taps ago and undo their prefetch distance)
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    const u32x4_t ov = {nout0, nout1, o2, o3};
                    char *const dstp = lds_w + ((ci & 1) ^ 1) * WIN_W + (wave + 8 * kb) * 1024 + lane * 16;
                    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)dstp),
                                 "v"(ov)
                                 : "memory");
                }
            }
            if (tap < 7) stage_A(tap + 2, ci, (tap + 2) % 3);
            else if (more) stage_A(tap - 7, ci + 1, (tap + 2) % 3);
            __builtin_amdgcn_s_barrier();
            WAIT_LGKMCNT(0);
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
                const int ro = wrd[2 + j + ky][kx];
                fb[j][0] = *(const a16x8 *)(smem + ro);
                fb[j][1] = *(const a16x8 *)(smem + (ro ^ 64));
            }
            __builtin_amdgcn_sched_barrier(0);
            // DMA issued after the weights of K-tile t + 1: this tap's and the previous tap's window block, and
            // the weights of K-tile t + 2
#ifdef EXP_NORM_NOLOOP
            const bool w_now = false, w_prev = false;
#else
            const bool w_now = tap < MAXWB && tap < nwb && more;
            const bool w_prev = tap >= 1 && tap - 1 < MAXWB && tap - 1 < nwb && more;
#endif
            if (NORM) {
                if (tap == 0 && more) load_ab(ci + 1);
                if (tap >= 2 && tap - 2 < MAXWB) {   // first half of block tap - 2
                    if (tap - 2 < nwb && more) {
                        const int kb = tap - 2 < MAXWB ? tap - 2 : 0;
                        norm_half(nraw[kb], woff[kb], 0, nout0, nout1);
                    }
                }
                if (tap < MAXWB) {
                    if (w_now) nraw[tap < MAXWB ? tap : 0] = load_raw(woff[tap < MAXWB ? tap : 0], ci + 1);
                }
            } else if (tap < MAXWB) {
                if (w_now) stage_W(woff[tap < MAXWB ? tap : 0], tap, ci + 1, (ci & 1) ^ 1);
            }
            const bool a_next = tap < 7 || more;     // were weights of K-tile t + 2 issued in phase 0?
            const bool w_prev_dma = w_prev;          // (NORM: the register loads of this and the previous tap)
            if (!a_next) {
                WAIT_VMCNT(0);
            } else if (w_now && w_prev_dma) {
                WAIT_VMCNT(4);
            } else if (w_now || w_prev_dma) {
                WAIT_VMCNT(3);
            } else {
                WAIT_VMCNT(2);
            }
            __builtin_amdgcn_s_barrier();
            WAIT_LGKMCNT(0);
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
        const int delta = (ci & 1) ? -WIN_W : WIN_W;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wrd[r][kx] += delta;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    TL(2);

#ifdef EXP_WIN_NOEPI
    if (p.alpha != 12345.f) return;
#endif
    // ---- epilogue -----------------------------------------------------------------------------------------------
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    const int cout_store = glu ? p.Cout / 2 : p.Cout;
    char *yb = (char *)p.y;
    const char *rb = (const char *)p.res;
    const int nw = n0 + wr * 64;
    float ssum[4], ssq[4], ssum_hi[4], ssq_hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ssum[i] = ssq[i] = ssum_hi[i] = ssq_hi[i] = 0.f;
    if (conv_epi_fast_ok(p, glu)) {
        // channels nw + 16 i + 4 fq .. + 3 of pixel (row 4 wc + bj, column frow): one 8-byte store per tile
        float4 bias[4];
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nw + i * 16 + fq * 4;
            ok[i] = n < p.Cout;
            bias[i] = (p.bias && ok[i]) ? *(const float4 *)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int64_t pix0 = ((int64_t)img * p.OH + y0 + 4 * wc) * p.OW + x0 + frow;
        uint2 res[4][4];
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
            const a16_t *rp = (const a16_t *)rb + (pix0 + bj * p.OW) * p.r_cs + p.r_co + nw + fq * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) res[bj][i] = (rb && ok[i]) ? *(const uint2 *)(rp + i * 16) : make_uint2(0u, 0u);
        }
        // A wave whose 64 channels all exist (and whose channel chunks are 16-byte aligned in y) sends its 64 x 64
        // outputs through LDS: fragments are written as [pixel][64 channels] rows of 128 bytes (16-byte chunk c of
        // pixel row r at c ^ (r & 7)), read back 16 bytes per lane and stored as whole 128-byte lines -- 8 store
        // instructions of full lines instead of 16 of scattered 32-byte pieces, which is what the store path charges
        // for.  Every LDS read of the main loop is behind the last barrier, and a wave only touches its own 8 KB.
        const bool rows = nw + 64 <= p.Cout && ((p.y_cs | p.y_co) & 7) == 0;
        typedef __attribute__((address_space(3))) char lds_char;   // explicit LDS pointers: no generic-pointer merging
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        lds_char *const my = (lds_char *)smem + wave * 8192;
        auto finish = [&](auto act, auto stats_mode) {
            constexpr int ACT = decltype(act)::value;
            constexpr int ST = decltype(stats_mode)::value;
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) {
                a16_t *yp = (a16_t *)yb + (pix0 + bj * p.OW) * p.y_cs + p.y_co + nw + fq * 4;
                const int px = bj * 16 + frow;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (rows) {
                        const uint2 o = conv_epi_fast_value<ACT, ST>(p, acc[i][bj >> 1][bj & 1], bias[i], res[bj][i], ssum[i],
                                                                     ssq[i], ssum_hi[i], ssq_hi[i]);
                        const int s8 = i * 4 + fq;             // 8-byte slot of the pixel's 128-byte row
                        *(__attribute__((address_space(3))) u32x2_t *)(my + px * 128 + (((s8 >> 1) ^ (px & 7)) << 4) + (s8 & 1) * 8) = u32x2_t{o.x, o.y};
                    } else if (ok[i]) {
                        conv_epi_fast<ACT>(p, acc[i][bj >> 1][bj & 1], bias[i], res[bj][i], yp + i * 16, ssum[i], ssq[i],
                                           ssum_hi[i], ssq_hi[i]);
                    }
                }
            }
        };
        // (activation, statistics mode) are compile-time parameters of the 64-quad loop: the run-time form costs six
        // selects per quad, a fifth of the epilogue's instructions
        auto with_stats = [&](auto act) {
            if (!p.stats) finish(act, std::integral_constant<int, 0>());
            else if (p.stats_gran == 2) finish(act, std::integral_constant<int, 2>());
            else finish(act, std::integral_constant<int, 4>());
        };
        if (p.epi == ONEDC_EPI_SILU) with_stats(std::integral_constant<int, ONEDC_EPI_SILU>());
        else if (p.epi == ONEDC_EPI_LEAKY) with_stats(std::integral_constant<int, ONEDC_EPI_LEAKY>());
        else with_stats(std::integral_constant<int, ONEDC_EPI_LINEAR>());
        TL(4);
        if (rows) {
            const int c = lane & 7;
#ifdef EXP_NOSTORE
            if (p.alpha == 12345.f)
#endif
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int px = it * 8 + (lane >> 3);           // pixel (row 4 wc + (px >> 4), column px & 15) of the patch
                const u32x4_t o = *(const __attribute__((address_space(3))) u32x4_t *)(my + px * 128 + ((c ^ (px & 7)) << 4));
                const int64_t pix = ((int64_t)img * p.OH + y0 + 4 * wc + (px >> 4)) * p.OW + x0 + (px & 15);
                *(u32x4_t *)((a16_t *)yb + pix * p.y_cs + p.y_co + nw + c * 8) = o;
            }
        }
    } else {
    EpiBias bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = conv_epi_bias(p, glu, nw + i * 16, fq);
    EpiRes<false> res[4][4];                           // pass 1: positions + residual loads of all 16 tiles
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) {
        ConvOutPix op;
        op.ob = op.ooy = op.oox = 0;
        op.m = (img * p.OH + y0 + 4 * wc + bj) * p.OW + x0 + frow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (glu && (i & 1)) continue;
            const EpiPos e = conv_epi_pos(p, glu, cout_store, nw + i * 16, fq, op, true);
            conv_epi_load_res<false>(p, rb, cout_store, e, res[bj][i]);
        }
    }
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) {                   // pass 2: finish + store
        ConvOutPix op;
        op.ob = op.ooy = op.oox = 0;
        op.m = (img * p.OH + y0 + 4 * wc + bj) * p.OW + x0 + frow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (glu && (i & 1)) continue;
            const int i2 = (i + 1 < 4) ? i + 1 : i;
            const EpiPos e = conv_epi_pos(p, glu, cout_store, nw + i * 16, fq, op, true);
            conv_epi_finish<false>(p, yb, glu, cout_store, acc[i][bj >> 1][bj & 1], acc[i2][bj >> 1][bj & 1], bias[i], e,
                                   res[bj][i], ssum[i], ssq[i], ssum_hi[i], ssq_hi[i]);
        }
    }
    }
#ifdef EXP_WIN_TIMELINE
    TL(3);
    if (lane == 0) {     // diagnostic build: p.stats is a debug buffer [blocks][8 waves][6] of int64
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
        unsigned long long *dbg = (unsigned long long *)p.stats + ((int64_t)blockIdx.x * 8 + wave) * 6;
        dbg[0] = ((unsigned long long)xcc << 32) | hwid;
        dbg[1] = tl[0];
        dbg[2] = tl[1];
        dbg[3] = tl[2];
        dbg[4] = tl[3];
        dbg[5] = tl[4];
    }
    return;
#endif
    if (p.stats) {                                     // one statistics row per (patch, wave column): 64 pixels
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (glu && (i & 1)) continue;
            const int n16 = nw + i * 16;
            conv_stats_store_row(p, ssum[i], ssq[i], pt * 4 + wc, (glu ? n16 / 2 : n16) + fq * 4, cout_store, lane,
                                 ssum_hi[i], ssq_hi[i]);
        }
    }
}

}  // namespace

// Used by onedc_conv_gemm for 3x3 / stride 1 / pad 1 layers whose output map is whole 16x16 patches.
bool onedc_conv_win_ok(const onedc_conv_params &p) {
    return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && !p.pixel_shuffle && p.batch <= 1 && !p.in_f32 &&
           (p.OH & 15) == 0 && (p.OW & 15) == 0 && (p.Cin & 63) == 0 &&
           (p.up ? (p.OH == 2 * p.IH && p.OW == 2 * p.IW) : (p.OH == p.IH && p.OW == p.IW)) &&
           (int64_t)p.B * p.IH * p.IW * p.x_cs < (int64_t)1 << 40 &&
           // window offsets inside one image are 32-bit byte offsets against a descriptor of at most 0xfffff000 bytes (both
           // window kernels, gemm_conv_win4.hip too): a bigger image (128 channels at 4096x4096) goes to the generic GEMM,
           // whose addressing is 64-bit, instead of wrapping or reading the out-of-range zeros
           (int64_t)p.IH * p.IW * p.x_cs * 2 < 0xfffff000ll;
}

// With gn_ab set the launch needs the plain (not upsampled) variant and room for the image's (a, b) pairs behind the
// window buffers.  conv_choose / onedc_conv_win_takes ask this BEFORE a caller commits to the fused normalisation.
bool onedc_conv_win_gn_ok(const onedc_conv_params &p) {
    return !p.gn_ab || (!p.up && WIN_LDS + (int64_t)p.Cin * 8 <= 160 * 1024);
}

int onedc_conv_win_launch(const onedc_conv_params &p, hipStream_t st) {
    static OnedcLdsAttrOnce once;
    once.ensure({{(const void *)conv_win_kernel<false, false>, WIN_LDS},
                 {(const void *)conv_win_kernel<true, false>, WIN_LDS},
                 {(const void *)conv_win_kernel<false, true>, 160 * 1024}});
    const int tiles_n = (p.Cout + 127) / 128;
    const int tiles_x = p.OW / 16, tiles_y = p.OH / 16;
    const int Ktot = 9 * p.Cin;
    const int64_t blocks = (int64_t)tiles_n * tiles_x * tiles_y * p.B;
    if (blocks > 0x7fffffff) return -1;
    const OnedcFastDiv dv_n((uint32_t)tiles_n), dv_pi((uint32_t)(tiles_x * tiles_y)), dv_x((uint32_t)(4 * tiles_x));
    if (p.gn_ab) {
        if (!onedc_conv_win_gn_ok(p)) return -13;
        hipLaunchKernelGGL((conv_win_kernel<false, true>), dim3((unsigned)blocks), dim3(512), WIN_LDS + p.Cin * 8, st, p,
                           tiles_n, tiles_x, tiles_y, Ktot, dv_n, dv_pi, dv_x);
    } else if (p.up) {
        hipLaunchKernelGGL((conv_win_kernel<true, false>), dim3((unsigned)blocks), dim3(512), WIN_LDS, st, p, tiles_n, tiles_x,
                           tiles_y, Ktot, dv_n, dv_pi, dv_x);
    } else {
        hipLaunchKernelGGL((conv_win_kernel<false, false>), dim3((unsigned)blocks), dim3(512), WIN_LDS, st, p, tiles_n, tiles_x,
                           tiles_y, Ktot, dv_n, dv_pi, dv_x);
    }
    ONEDC_LAUNCH_CHECK();
    return 0;
}
