// gemm_conv_win4.hip — the 3x3 window convolution (gemm_conv_win.hip) re-cut so that TWO workgroups share a CU.
//
// The 8-wave window kernel owns a CU alone (133 KB of LDS): nothing runs while a workgroup fetches its first window
// (2.4 us), finishes its outputs (2.7 us) or hands the CU to its successor (1.3 us) -- 6-7 us of a 19 us tile on the
// 128-channel layers at 768x768.  Here a workgroup is 4 waves and 74 KB of LDS, two of them are resident per CU, and each
// SIMD holds one wave of either: while one workgroup is in its prologue or epilogue the other one's K loop has the
// matrix cores to itself.  What makes that fit:
//   K-tile       one tap of one 32-CHANNEL chunk (one MFMA k step): window 21 KB instead of 41 KB, weight tile 8 KB
//   wave tile    64 channels x 128 pixels (8 patch rows): 128 accumulator registers, 12 fragment reads per 32 MFMAs
//                (the 8-wave kernel: 16 per 32), weights staged once per 16x16 patch as before
//   LDS images   window: pixel-major, 64 B per pixel = four 16-byte granules (8-channel groups fq) stored at fq ^ s(x),
//                s(x) = 2 * ((x >> 2) & 1) with x the window column: a fragment read (16 consecutive pixels of a window row,
//                two fq values per hardware lane group) then covers all 16 bank groups, and because s depends on the column
//                only, every (row, tap) address is ONE per-lane base per kx plus an instruction immediate.  Weights: a DMA
//                block is 16 rows x 64 B and granule == lane on both sides (DMA lane l writes it, fragment lane l reads it)
//   pipeline     inside a wave: fragments of K-tile t + 1 are read while the second half of K-tile t multiplies (A
//                double-buffered, the two pixel halves alternate), ONE barrier and one counted vmcnt per K-tile; weights
//                ride a ring of four tiles, issued four K-tiles ahead
// Same arithmetic contract as the 8-wave kernel's fast path (16-bit output, bias / activation / residual / GroupNorm
// statistics in the epilogue, line stores through LDS); k order (32-channel chunk, tap): sums round differently from the
// other kernels, each kernel alone is deterministic and batch-invariant.
// STATUS: the automatic choice (conv_choose in gemm_conv.hip) for every window layer with the linear epilogue on whole
// 64-channel wave tiles, with or without GroupNorm on load, and for the upsample phases (MODE 2); tile 131 forces it, tile
// 130 the 8-wave kernels.  tests/test_gpu_kernels.py::test_conv_win3x3_4wave_*, test_conv_upsample_phases*.  Measured
// (DESIGN.md section 3 / 4): 3-19 % faster than the 8-wave kernel per layer; the chip holds 1.67 GHz in the K loop, where
// the two resident workgroups keep the matrix pipe 91-96 % busy; a workgroup ALONE on its CU (its partner in prologue or
// epilogue) reaches half of that, which is what the 128-channel layers at 768x768 still lose.
#include <type_traits>

#include "conv_common.h"

namespace {

constexpr int W4_A = 8192;                   // one weight tile: 128 rows x 32 channels
constexpr int W4_NA = 4;                     // weight ring
constexpr int W4_NWB = 21;                   // window blocks of 16 pixels (18 x 18 = 324 -> 336)
constexpr int W4_W = W4_NWB * 1024;          // 21504 B per window
constexpr int W4_WOFF = 0;                   // windows first, then the ring
constexpr int W4_AOFF = 2 * W4_W;            // 43008
constexpr int W4_LDS = W4_AOFF + W4_NA * W4_A;   // 75776 B: two workgroups per CU

#define W4_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define W4_WAIT_LGKMCNT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define W4_PIN(x) asm volatile("" : "+v"(x))
#define W4_WAIT_VMCNT_N(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int I, int N, class F>
__device__ __forceinline__ void w4_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        w4_static_for<I + 1, N>(f);
    }
}

// ST: GroupNorm statistics of the epilogue (0 none, 4 per channel quad, 2 per channel pair) -- a template parameter of the
// kernel, not a branch inside it: with the variants side by side in one kernel the compiler hoists their common head
// (128 fma + the residual conversions) above the branch and spills accumulators around it
// MODE: 0 plain 3x3; 1 GroupNorm (+SiLU) applied to the window on load; 2 one output phase of the x2-upsample + 3x3
// decomposition (gemm_conv_ups.hip: a 2x2 convolution of the low-resolution input with summed weights): tap (ty, tx) of
// phase (PY, PX) reads exactly what tap (PY + ty, PX + tx) of the 3x3 kernel reads, so the phase is two per-lane base
// addresses chosen at the start, four K-tiles per chunk instead of nine, and an output scatter to (2 y + PY, 2 x + PX).
// X2: a second source (onedc_conv_params.x2 / w2 / Cin2): the 1x1 convolution of a tensor on the output grid -- the resnet's
// conv_shortcut, whose input is the block's input -- accumulated into the same tile behind the 3x3 K loop, 32 channels per
// K-tile.  The separate shortcut launch, its output tensor and the residual read of it disappear (the VAE's 256 -> 128 block
// at 768x768: 1.7 ms + 2.4 GB written + 2.4 GB re-read per batch of 16).  A template parameter so that the kernels without a
// second source are compiled exactly as before.
// BF (round 5, IEEE-half build only): the 3x3 part multiplies bf16 OPERANDS -- x is a normalised tensor stored as bf16 by the
// scale/shift pass (MODE 0) or is normalised to bf16 on load (MODE 1), w is packed bf16 -- while everything that is STORED (output,
// residual, statistics, a second source and its weights) keeps the build's format.  The chip holds a 3 - 6 % higher clock on 8-bit
// mantissas, and tests/devtools/bf16_operands_sim.py: with f16 storage, bf16 operands in every resnet conv of the VAE and the UNet
// move the picture by 0.0003 dB of the 0.01 dB budget (the all-bf16 library fails it through its stored streams).
template <int MODE, int ST, bool X2 = false, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv_win4_kernel(const onedc_conv_params p, int tiles_n, int tiles_x, int tiles_y,
                                                           int Ktot, OnedcFastDiv dv_n, OnedcFastDiv dv_pi,
                                                           OnedcFastDiv dv_x, OnedcFastDiv dv_c) {
    constexpr bool NORM = MODE == 1, UPS = MODE == 2;
    constexpr int TAPS = UPS ? 4 : 9;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef EXP_W4_TIMELINE
    unsigned long long tl[14];
    tl[0] = __builtin_amdgcn_s_memtime();
#define W4_TL(i) tl[i] = __builtin_amdgcn_s_memtime()
#define W4_RT(i) tl[i] = __builtin_amdgcn_s_memrealtime()
#else
#define W4_TL(i)
#define W4_RT(i)
#endif
    // slots 10 .. 13 carry the prologue's detail stamps -- or, with -DEXP_W4_EPI_TL (round 6), the epilogue's: 10 staging rows
    // free (barrier passed), 12 residual rows 0 .. 3 landed (a forced wait: diagnostic only), 11 arithmetic + LDS transpose done,
    // 3 stores issued, 13 stores drained (tools/timeline_win4.py --epilogue)
#if defined(EXP_W4_TIMELINE) && defined(EXP_W4_EPI_TL)
#define W4_PTL(i)
#define W4_ETL(i) W4_TL(i)
#else
#define W4_PTL(i) W4_TL(i)
#define W4_ETL(i)
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;        // channel half, pixel half (patch rows 8 wc ..)
#ifdef EXP_W4_STAGGER
    // (experiment) The first round of workgroups starts within a few microseconds on every CU and all tiles of a layer
    // take the same time: the CUs run in lockstep and their epilogues (64 KB of stores each) hit the memory system as one
    // burst.  Delay the first-round workgroups of CU group q by q / EXP_W4_STAGGER of a tile time, once per launch.
    if (blockIdx.x < 2 * 256) {
        const int q = (blockIdx.x >> 3) % EXP_W4_STAGGER;
        const int cyc = q * ((Ktot >> 5) * 1100 + 15000) / EXP_W4_STAGGER;
        for (int i = 0; i < (cyc >> 12); ++i) __builtin_amdgcn_s_sleep(64);
    }
#endif

    // ---- patch of this workgroup (same order as the 8-wave kernel: channel tile fastest, bands of four patch rows) ----
    const int tiles_pi = tiles_x * tiles_y;
#ifdef EXP_W4_PERSIST2
    // (ablation) a workgroup runs TWO patches back to back; the second one without a prologue (no DMA of its first window and
    // weights: its K loop multiplies what is in LDS) -- the time a persistent form could reach at best, results of the second
    // patch are garbage.  Grid = half the tiles.
    const int total_tiles = tiles_n * tiles_pi * p.B;
    const int half_tiles = (total_tiles + 1) >> 1;
    const int logical = xcd_logical_tile(blockIdx.x, half_tiles);
#else
    const int logical = xcd_logical_tile(blockIdx.x, tiles_n * tiles_pi * p.B);
#endif
    int pt, n0, img, pin, y0, x0;
    auto decode_tile = [&](int lg) {
#ifdef EXP_W4_NSLOW
        // (experiment, round 5) channel tile SLOW inside groups of EXP_W4_NSLOW patches: the workgroups resident on an XCD
        // then stream ONE channel tile's weights at a time (1.2 MB at Cin = 512 instead of 4.7 MB next to a 4 MB L2);
        // the window of a patch is fetched again per channel tile by construction
        {
            const int P = tiles_pi * p.B, G = EXP_W4_NSLOW;
            const int g = lg / (G * tiles_n), rem = lg - g * G * tiles_n;
            const int gl = P - g * G < G ? P - g * G : G;
            const int nn = rem / gl;
            pt = g * G + rem - nn * gl;
            n0 = nn * 128;
        }
#else
        pt = dv_n.div(lg);
        n0 = (lg - pt * tiles_n) * 128;
#endif
        img = dv_pi.div(pt);
        pin = pt - img * tiles_pi;
        int ty, tx;
        const int band = dv_x.div(pin), inb = pin - band * 4 * tiles_x;
        const int left = tiles_y - band * 4, rows = left < 4 ? left : 4;
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
        } else {
            tx = (inb * 21846) >> 16;
            r = inb - 3 * tx;
        }
        ty = band * 4 + r;
        y0 = ty * 16;                               // patch origin (UPS: on the low-resolution grid)
        x0 = tx * 16;
    };
    decode_tile(logical);
    const int phase = UPS ? dv_c.div(n0) : 0;       // a 128-row tile of the phase-packed matrix belongs to one phase
    const int PY = phase >> 1, PX = phase & 1;

    // ---- DMA geometry ------------------------------------------------------------------------------------------------
    // lane l of a DMA instruction moves granule l of a 1 KB block: row (l & 15) of the block, channel group l >> 4
    const int drow = lane & 15, dfq = lane >> 4;    // (weights)
    const int w_rows = tiles_n * 128;
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w, (unsigned)((int64_t)w_rows * Ktot * 2));
    const unsigned wvoff = (unsigned)((int64_t)(n0 + wave * 16 + drow) * Ktot * 2) + dfq * 16;
    const unsigned wrow64 = (unsigned)(64 * Ktot * 2);
    char *const lds_a = smem + W4_AOFF;
    // weights of K-tile (32-channel chunk c32, tap) into ring slot `slot`: row blocks `wave` and `wave + 4`
    auto stage_A = [&](int tap, int c32, int slot) {
        char *dst = lds_a + slot * W4_A + wave * 1024;
        const unsigned so = (unsigned)(tap * p.Cin + c32 * 32) * 2;
        glds16_buf(w_rsrc, wvoff, so, dst);
        glds16_buf(w_rsrc, wvoff, so + wrow64, dst + 4096);
    };
    W4_PTL(10);                                      // (diagnostic) tile decoded, weight offsets ready
    stage_A(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    W4_PTL(11);                                      // (diagnostic) first weight tile issued

    const int64_t img_bytes = (int64_t)p.IH * p.IW * p.x_cs * 2;
    const unsigned x_bytes = img_bytes < 0xfffff000ll ? (unsigned)img_bytes : 0xfffff000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc((const char *)p.x + img * img_bytes, x_bytes);
    constexpr int MAXWB = 6;                          // window blocks wave + 4 k: wave 0 has six, the others five
    const int nwb = wave == 0 ? 6 : 5;
    // window block: 16 pixels x 64 B, lane l moves granule (l & 3) of pixel l >> 2: source channel group (l & 3) ^ s(x)
    unsigned woff[MAXWB];
    unsigned sbits = 0;                               // bit k: s(x) != 0 for this lane's pixel of block k (NORM: which channels)
#pragma unroll
    for (int k = 0; k < MAXWB; ++k) {
        const int w = (wave + 4 * k) * 16 + (lane >> 2);
        const int wy = (w * 3641) >> 16, wx = w - wy * 18;           // w / 18, exact for w < 1 << 12
        const int iy = y0 - 1 + wy, ix = x0 - 1 + wx;
        const bool ok = w < 324 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        const int sfq = (lane & 3) ^ (((wx >> 2) & 1) << 1);
        sbits |= (unsigned)((wx >> 2) & 1) << k;
        woff[k] = ok ? (unsigned)((iy * p.IW + ix) * p.x_cs + p.x_co) * 2u + sfq * 16 : x_bytes;
    }
    W4_PTL(12);                                      // (diagnostic) window offsets computed
    auto stage_W = [&](int k, int c32, int wbuf) {
        glds16_buf(x_rsrc, woff[k], (unsigned)c32 * 64, smem + W4_WOFF + wbuf * W4_W + (wave + 4 * k) * 1024);
    };
#pragma unroll
    for (int k = 0; k < MAXWB; ++k)
        if (k < nwb) stage_W(k, 0, 0);
    W4_PTL(13);                                      // (diagnostic) first window issued
    stage_A(1, 0, 1);
    stage_A(2, 0, 2);
    stage_A(3, 0, 3);
    __builtin_amdgcn_sched_barrier(0);
    W4_TL(6);                                        // (diagnostic) every prologue DMA is issued

    // NORM: x is the raw tensor in front of a GroupNorm (+SiLU); a window block is normalised IN PLACE two or more taps
    // after its DMA was issued: the lane that moved a granule reads it back, applies v = silu(a[c] * x + b[c]) -- the
    // operations of onedc_scale_shift_act in the same order, so the conv sees bit-identical operands -- and writes it
    // again; halo granules stay zero (the conv pads the NORMALISED tensor).  (a, b) of the image sit behind the ring.
    auto norm_block = [&](int k, int c32, int wbuf) {
        const int addr = W4_WOFF + wbuf * W4_W + (wave + 4 * k) * 1024 + lane * 16;
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        // (all LDS traffic of this lambda is asm: in front of an LDS access it can see, the compiler drains vmcnt --
        // it cannot prove that the LDS-DMA in flight does not alias it -- and with it the prefetch distance of the loop)
        u32x4_t raw;
        f32x4 ab[2];                                 // (a, b) pairs of four of the lane's 8 channels
        const int cbase = c32 * 32 + ((lane & 3) ^ (((sbits >> k) & 1) << 1)) * 8;
        const int abaddr = W4_LDS + cbase * 8;
#ifdef EXP_NORM_NONE                                 // (ablation) no normalisation work at all: what GroupNorm on load costs in total
        (void)addr;
        (void)abaddr;
        return;
#endif
#ifdef EXP_NORM_TILE0_ONLY                           // (ablation, round 6: results of the other channel tiles are garbage) only the FIRST
        if (n0 != 0) return;                         // channel tile of a window normalises it: the ceiling of a 256-channel tile that
#endif                                               // normalises once per window and shares the result
        asm volatile("ds_read_b128 %0, %1" : "=v"(raw) : "v"(addr));
#ifdef EXP_NORM_NOAB                                 // (ablation) (a, b) from registers instead of 4 LDS reads per block
        ab[0] = ab[1] = f32x4{1.f, 0.25f, 1.f, 0.25f};
#else
        asm volatile("ds_read_b128 %0, %1" : "=v"(ab[0]) : "v"(abaddr));
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(ab[1]) : "v"(abaddr));
#endif
        const bool live = woff[k] != x_bytes;
        u32x4_t o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#ifndef EXP_NORM_NOAB
            if (h == 1) {                            // second four channels into the same registers
                asm volatile("ds_read_b128 %0, %1 offset:32" : "+v"(ab[0]) : "v"(abaddr));
                asm volatile("ds_read_b128 %0, %1 offset:48" : "+v"(ab[1]) : "v"(abaddr));
            }
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw), "+v"(ab[0]), "+v"(ab[1]));
            const f32x4 ab0 = ab[0], ab1 = ab[1];     // (a, b) of channels 4 h .. 4 h + 3
            const uint32_t w0 = h ? raw[2] : raw[0], w1 = h ? raw[3] : raw[1];
            float v0 = __builtin_fmaf(a2f((a16_t)(w0 & 0xffff)), ab0[0], ab0[1]);
            float v1 = __builtin_fmaf(a2f((a16_t)(w0 >> 16)), ab0[2], ab0[3]);
            float v2 = __builtin_fmaf(a2f((a16_t)(w1 & 0xffff)), ab1[0], ab1[1]);
            float v3 = __builtin_fmaf(a2f((a16_t)(w1 >> 16)), ab1[2], ab1[3]);
            if (p.gn_silu) {
                v0 = silu_f(v0);
                v1 = silu_f(v1);
                v2 = silu_f(v2);
                v3 = silu_f(v3);
            }
            o[2 * h] = live ? (BF ? pack_bf16_2(v0, v1) : pack_a2(v0, v1)) : 0u;
            o[2 * h + 1] = live ? (BF ? pack_bf16_2(v2, v3) : pack_a2(v2, v3)) : 0u;
        }
        asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(o) : "memory");
    };
    if (NORM) {
        float2 *const lds_ab = (float2 *)(smem + W4_LDS);
        for (int c = tid; c < p.Cin; c += 256) lds_ab[c] = ((const float2 *)p.gn_ab)[(int64_t)img * p.Cin + c];
        W4_WAIT_VMCNT(6);                            // this wave's blocks of window 0 are in (weights 1 .. 3 may fly on)
        __syncthreads();                             // (a, b) are in LDS
#pragma unroll
        for (int k = 0; k < MAXWB; ++k)
            if (k < nwb) norm_block(k, 0, 0);
        W4_WAIT_LGKMCNT(0);
    }

    // ---- fragment addresses --------------------------------------------------------------------------------------------
    const int frow = lane & 15, fq = lane >> 4;
    // weights: row block wr * 4 + i of the slot, granule == lane
    int fao = W4_AOFF + wr * 4096 + lane * 16;
    // pixels: output (patch row 8 wc + r, column frow) under tap (ky, kx) reads window pixel (8 wc + r + ky) * 18 + frow + kx:
    // base of the lane for kx, immediate ((r + ky) * 18 + kx) * 64
    int wrd[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        wrd[kx] = W4_WOFF + (8 * wc * 18 + frow) * 64 + ((fq ^ ((((frow + kx) >> 2) & 1) << 1)) << 4);
    if (UPS) {                                       // tap (ty, tx) == 3x3 tap (PY + ty, PX + tx): bases of columns PX, PX + 1
        const int shift = (PY * 18 + PX) * 64;
        const int b0 = PX ? wrd[1] : wrd[0], b1 = PX ? wrd[2] : wrd[1];
        wrd[0] = b0 + shift;
        wrd[1] = b1 + shift;
    }
    W4_PIN(fao);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) W4_PIN(wrd[kx]);

#ifdef EXP_W4_PERSIST2
    for (int rep = 0; rep < 2; ++rep) {
    if (rep == 1) {
        if (logical + half_tiles >= total_tiles) break;
        decode_tile(logical + half_tiles);
    }
#endif
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc[i][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+v"(acc[i][r]));
        }

    // Fragment registers.  Every LDS read below is an asm statement whose destination is a READ-WRITE operand: old and new
    // value of a fragment variable then share one register for the whole kernel.  (With plain loads the 18 statically
    // unrolled K-tiles are 18 x 12 independent values for the register allocator, which lets them wander, renames the
    // accumulators around them and ends up spilling accumulators -- and a scratch reload in the loop drains vmcnt.)
    // The compiler does not track these reads: the lgkmcnt waits are explicit and carry the fragments as operands, so
    // no MFMA that uses them can be moved in front of the wait.
    a16x8 fa[2][4], fbA[4], fbB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[0][i] = fa[1][i] = fbA[i] = fbB[i] = a16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#define W4_RD(dst, addr, off)                                                                                 \
    do {                                                                                                      \
        a16x8 &d__ = (dst);                                                                                   \
        const int a__ = (addr);                                                                               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(d__) : "v"(a__), "n"(off));                        \
    } while (0)
    // The MFMAs are asm as well, accumulating in place: as builtins the allocator renamed accumulators from K-tile to
    // K-tile (dst != src C), which took 56 more registers and pushed the DMA offsets into scratch.  What the compiler no
    // longer sees: the accumulators are matrix-core results (wait states in front of the epilogue's first VALU read).
#ifdef ONEDC_ACT_F16
#define W4_MFMA_N(c, a, b)                                                                                    \
    do {                                                                                                      \
        f32x4 &c__ = (c);                                                                                     \
        const a16x8 &a__ = (a), &b__ = (b);                                                                   \
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c__) : "v"(a__), "v"(b__));               \
    } while (0)
#else
#define W4_MFMA_N(c, a, b)                                                                                    \
    do {                                                                                                      \
        f32x4 &c__ = (c);                                                                                     \
        const a16x8 &a__ = (a), &b__ = (b);                                                                   \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c__) : "v"(a__), "v"(b__));              \
    } while (0)
#endif
    // the 3x3 part: bf16 operands where the kernel is instantiated for them (BF), the build's format otherwise; the second
    // source (a stored tensor) always multiplies in the build's format (W4_MFMA_N)
#define W4_MFMA(c, a, b)                                                                                      \
    do {                                                                                                      \
        if constexpr (BF) {                                                                                   \
            f32x4 &c__ = (c);                                                                                 \
            const a16x8 &a__ = (a), &b__ = (b);                                                               \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c__) : "v"(a__), "v"(b__));          \
        } else {                                                                                              \
            W4_MFMA_N(c, a, b);                                                                               \
        }                                                                                                     \
    } while (0)
#define W4_WAIT_LGKM4(n, q)                                                                                   \
    do {                                                                                                      \
        a16x8 &q0__ = (q)[0], &q1__ = (q)[1], &q2__ = (q)[2], &q3__ = (q)[3];                                 \
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(q0__), "+v"(q1__), "+v"(q2__), "+v"(q3__) : "n"(n));       \
    } while (0)
    const int nc32 = p.Cin >> 5;                     // even: Cin is a multiple of 64
    __builtin_amdgcn_sched_barrier(0);
    W4_TL(7);                                        // (diagnostic) set-up arithmetic done
    // window of chunk 0 and the weights of K-tile 0 (issued first) are in; weights 1 .. 3 (six instructions) may fly on
    W4_WAIT_VMCNT(6);
    W4_TL(8);                                        // (diagnostic) own DMA landed
    __builtin_amdgcn_s_barrier();
    W4_TL(9);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) W4_RD(fa[0][i], fao, i * 1024);
#pragma unroll
    for (int r = 0; r < 4; ++r) W4_RD(fbA[r], wrd[0], r * 18 * 64);
    __builtin_amdgcn_sched_barrier(0);

    W4_TL(1);
    W4_RT(4);
    int slot = 0;                                    // ring slot of the K-tile being multiplied (wave-uniform)
    // two chunks (18 K-tiles) per pass: window buffer, tap and fragment buffer are compile-time.  LAST: no chunk behind the
    // odd one of the pair -- its own copy of the body, so that nothing in the loop is decided at run time
    auto chunk_pair = [&](int c2, auto last_tag) __attribute__((always_inline)) {
        constexpr bool last2 = decltype(last_tag)::value;
        w4_static_for<0, 2 * TAPS>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            constexpr int cp = j / TAPS, tap = j % TAPS;   // chunk parity (window buffer), tap
            constexpr int ky = UPS ? tap >> 1 : tap / 3, kx = UPS ? tap & 1 : tap % 3;
            constexpr int cur = j & 1, nxt = cur ^ 1;
            constexpr bool more = cp == 0 ? true : !last2;           // is there a chunk after this one?
            constexpr int wb_off = cp * W4_W;                        // this chunk's window buffer
            const int c32 = c2 + cp;
            a16x8(&FA)[2][4] = fa;
            f32x4(&ACC)[4][8] = acc;
            a16x8(&FBA)[4] = fbA;
            a16x8(&FBB)[4] = fbB;
            const int(&WRD)[3] = wrd;
            // (a) second pixel half of this K-tile
#pragma unroll
            for (int r = 0; r < 4; ++r) W4_RD(FBB[r], WRD[kx], ((4 + r + ky) * 18 + kx) * 64 + wb_off);
            // (b) first half: patch rows 0 .. 3 of the wave; its fragments were read during the previous K-tile
            W4_WAIT_LGKM4(4, FA[cur]);
            W4_WAIT_LGKM4(4, FBA);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) W4_MFMA(ACC[i][r], FA[cur][i], FBA[r]);
#ifndef EXP_W4_KPRIO                                  // (experiment, round 6: the whole K loop at raised priority, not only its MFMA groups)
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            // (c) the weights of K-tile t + 1 (and at tap 8 the whole next window) have landed -- everywhere.
            //     Issue order per K-tile: weights of K-tile t + 4 (two instructions), then the window blocks of the next
            //     chunk: blocks 0, 1 at tap 0, blocks 2, 3 at tap 1, block 4 (and 5: wave 0 only) at tap 2.
            //     Younger than the weights of t + 1: blocks of t - 3, weights t + 2, blocks of t - 2, weights t + 3, blocks
            //     of t - 1.  NORM normalises block k at tap k + 2 and needs IT landed: fewer may stay in flight.
            if constexpr (!UPS) {
                constexpr auto x2 = [](int k) { return (k == 0 || k == 1) ? 2 : (k == 2 ? 1 : 0); };   // + 1 at tap 2 for wave 0
                constexpr auto has5 = [](int k) { return k == 2; };
                constexpr int n_w = 4 + x2(tap - 3) + x2(tap - 2) + x2(tap - 1);
                constexpr bool w5 = has5(tap - 3) || has5(tap - 2) || has5(tap - 1);
                // NORM: block tap - 2 and what may follow it in the queue (derivation in DESIGN.md section 3)
                constexpr int n_n = tap == 2 ? 5 : tap == 3 ? 7 : tap == 4 ? 6 : tap == 5 ? 5 : 63;
                constexpr bool n5 = tap == 3 || tap == 4 || tap == 5;
                if constexpr (more) {
                    if constexpr (NORM && n_n < n_w) {
                        if (n5 && wave == 0) W4_WAIT_VMCNT_N(n_n + 1);
                        else W4_WAIT_VMCNT_N(n_n);
                    } else if constexpr (w5) {
                        if (wave == 0) W4_WAIT_VMCNT_N(n_w + 1);
                        else W4_WAIT_VMCNT_N(n_w);
                    } else {
                        W4_WAIT_VMCNT_N(n_w);
                    }
                } else {                             // last chunk: no window blocks; weights t + 2 / t + 3 exist up to tap 6 / 5
                    W4_WAIT_VMCNT_N(tap <= 5 ? 4 : tap == 6 ? 2 : 0);
                }
            } else {
                // four taps per chunk: the weights of K-tile t + 4 are the same tap of the next chunk, and a K-tile issues
                // its window blocks BEFORE its weights (at tap 3 the whole next window must be in; only the weights issued
                // at tap 2 may still fly).  Younger than the weights of t + 1 (issued at t - 3): blocks and weights of
                // t - 2 and t - 1.  At tap 0 everything up to the blocks of the previous tap 2 has landed already (the
                // tap-3 wait of the previous chunk), whichever chunk that was -- or the prologue.
                if constexpr (more) W4_WAIT_VMCNT_N(tap == 0 ? 4 : tap == 1 ? 6 : tap == 2 ? 8 : 2);
                else W4_WAIT_VMCNT_N(tap == 0 ? 4 : tap == 1 ? 2 : 0);
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // (d) DMA: weights of K-tile t + 4 into the slot of K-tile t (every wave is past its reads of it), window
            //     blocks of the next chunk
            {
                constexpr int tap4 = (tap + 4) % TAPS, over = (tap + 4) / TAPS;
                if constexpr (!UPS && (over == 0 || more)) stage_A(tap4, c32 + over, slot);
                if constexpr (tap < 3 && more) {
                    stage_W(2 * tap, c32 + 1, cp ^ 1);
                    if (tap < 2 || wave == 0) stage_W(2 * tap + 1, c32 + 1, cp ^ 1);
                }
                if constexpr (UPS && more) stage_A(tap4, c32 + over, slot);
                if constexpr (NORM && more && tap >= 2 && tap - 2 < MAXWB) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap - 2 < 5 || wave == 0) norm_block(tap - 2, c32 + 1, cp ^ 1);
                }
            }
            slot = (slot + 1) & 3;
            __builtin_amdgcn_sched_barrier(0);
            // (e) fragments of K-tile t + 1: weights and the first pixel half
            if constexpr (tap < TAPS - 1 || more) {
                constexpr int tap1 = (tap + 1) % TAPS;
                constexpr int ky1 = UPS ? tap1 >> 1 : tap1 / 3, kx1 = UPS ? tap1 & 1 : tap1 % 3;
                constexpr int wb1 = (tap == TAPS - 1 ? cp ^ 1 : cp) * W4_W;
                const int a_rd = fao + slot * W4_A;
#pragma unroll
                for (int i = 0; i < 4; ++i) W4_RD(FA[nxt][i], a_rd, i * 1024);
#pragma unroll
                for (int r = 0; r < 4; ++r) W4_RD(FBA[r], WRD[kx1], ((r + ky1) * 18 + kx1) * 64 + wb1);
                W4_WAIT_LGKM4(8, FBB);
            } else {
                W4_WAIT_LGKM4(0, FBB);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (f) second half: patch rows 4 .. 7
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) W4_MFMA(ACC[i][4 + r], FA[cur][i], FBB[r]);
#ifndef EXP_W4_KPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    int c2 = 0;
    for (; c2 + 2 < nc32; c2 += 2) chunk_pair(c2, std::false_type());
    chunk_pair(c2, std::true_type());

    if constexpr (X2) {
        // ---- second source: K-tiles of the 1x1 layer, one 32-channel chunk each ---------------------------------------
        // The 3x3 loop has drained its DMA (vmcnt 0 at its last tap) and every wave has waited for its last fragments; a
        // chunk of x2 is staged in the WINDOW layout (the same 21 blocks, offsets for x2's channel stride), so that its
        // fragments are the reads of the centre tap (ky = kx = 1), its weights go to ring slots 0 / 1.  Two chunks in
        // flight; per chunk: counted wait, barrier (landed everywhere), 12 fragment reads, barrier (buffers free), DMA of
        // chunk c + 2, 32 MFMAs.  The partner workgroup on the CU covers the waits.
        __builtin_amdgcn_s_barrier();                // every wave is past its last fragment read of the 3x3 part
        const int64_t img2_bytes = (int64_t)p.OH * p.OW * p.x2_cs * 2;
        const unsigned x2_bytes = img2_bytes < 0xfffff000ll ? (unsigned)img2_bytes : 0xfffff000u;
        const __amdgpu_buffer_rsrc_t x2_rsrc = make_rsrc((const char *)p.x2 + img * img2_bytes, x2_bytes);
        // (the lane index goes through an opaque asm: otherwise the compiler shares the window arithmetic below with the
        // prologue's and keeps two dozen registers alive -- and spilled -- across the 3x3 loop)
        int lane_t = lane;
        W4_PIN(lane_t);
        unsigned woff2[MAXWB];
#pragma unroll
        for (int k = 0; k < MAXWB; ++k) {
            const int w = (wave + 4 * k) * 16 + (lane_t >> 2);
            const int wy = (w * 3641) >> 16, wx = w - wy * 18;
            const int iy = y0 - 1 + wy, ix = x0 - 1 + wx;
            const bool ok = w < 324 && (unsigned)iy < (unsigned)p.OH && (unsigned)ix < (unsigned)p.OW;
            const int sfq = (lane_t & 3) ^ (((wx >> 2) & 1) << 1);
            woff2[k] = ok ? (unsigned)((iy * p.OW + ix) * p.x2_cs + p.x2_co) * 2u + sfq * 16 : x2_bytes;
        }
        const __amdgpu_buffer_rsrc_t w2_rsrc = make_rsrc(p.w2, (unsigned)((int64_t)w_rows * p.Cin2 * 2));
        const unsigned wvoff2 = (unsigned)((n0 + wave * 16 + (lane_t & 15)) * p.Cin2 * 2) + (lane_t >> 4) * 16;
        const unsigned w2row64 = (unsigned)(64 * p.Cin2 * 2);
        auto stage2 = [&](int c32, int par) {        // chunk c32 into window buffer / ring slot `par`
            char *dst = lds_a + par * W4_A + wave * 1024;
            glds16_buf(w2_rsrc, wvoff2, (unsigned)c32 * 64, dst);
            glds16_buf(w2_rsrc, wvoff2, (unsigned)c32 * 64 + w2row64, dst + 4096);
#pragma unroll
            for (int k = 0; k < MAXWB; ++k)
                if (k < nwb) glds16_buf(x2_rsrc, woff2[k], (unsigned)c32 * 64, smem + W4_WOFF + par * W4_W + (wave + 4 * k) * 1024);
        };
        const int n2 = p.Cin2 >> 5;
        stage2(0, 0);
        if (n2 > 1) stage2(1, 1);
        for (int c = 0; c < n2; ++c) {
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < n2) {                        // chunk c + 1 (2 weight + nwb window instructions) may fly on
                if (wave == 0) W4_WAIT_VMCNT(8);
                else W4_WAIT_VMCNT(7);
            } else {
                W4_WAIT_VMCNT(0);
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int par = c & 1;
            const int a_rd = fao + par * W4_A, b_rd = wrd[1] + par * W4_W;
#pragma unroll
            for (int i = 0; i < 4; ++i) W4_RD(fa[0][i], a_rd, i * 1024);
#pragma unroll
            for (int r = 0; r < 4; ++r) W4_RD(fbA[r], b_rd, ((r + 1) * 18 + 1) * 64);
#pragma unroll
            for (int r = 0; r < 4; ++r) W4_RD(fbB[r], b_rd, ((4 + r + 1) * 18 + 1) * 64);
            W4_WAIT_LGKM4(0, fa[0]);
            W4_WAIT_LGKM4(0, fbA);
            W4_WAIT_LGKM4(0, fbB);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();            // every wave holds its fragments: the buffers may be restaged
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < n2) stage2(c + 2, par);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) W4_MFMA_N(acc[i][r], fa[0][i], fbA[r]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) W4_MFMA_N(acc[i][4 + r], fa[0][i], fbB[r]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

#ifdef EXP_W4_KPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    W4_TL(2);
    W4_RT(5);
    // ---- epilogue --------------------------------------------------------------------------------------------------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (asm: the compiler does not count wait states)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(acc[i][r]));
    const int nw = n0 - phase * p.Cout + wr * 64;    // (UPS: packed rows are [phase][channel])
    const bool have = nw + 64 <= p.Cout;             // Cout is a multiple of 64: a wave's channels all exist or none does
    char *yb = (char *)p.y;
    const char *rb = (const char *)p.res;
    float4 bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        bias[i] = (p.bias && have) ? *(const float4 *)(p.bias + nw + i * 16 + fq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t pix0 = ((int64_t)img * p.OH + y0 + 8 * wc) * p.OW + x0 + frow;
    // residual: rows 0 .. 3 now; rows 4 .. 7 are fetched while the first rows are finished, into registers their
    // accumulators have left (all eight at once would be 64 registers on top of the 128 accumulators)
    uint2 res[8][4];
    auto load_res_row = [&](int r) {
        const a16_t *rp = (const a16_t *)rb + (pix0 + r * p.OW) * p.r_cs + p.r_co + nw + fq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) res[r][i] = (rb && have) ? *(const uint2 *)(rp + i * 16) : make_uint2(0u, 0u);
    };
#pragma unroll
    for (int r = 0; r < 4; ++r) load_res_row(r);
    // every wave is done with the windows and the ring: a wave's 64 x 128 outputs go through 16 KB of them
    W4_WAIT_LGKMCNT(0);
    __builtin_amdgcn_s_barrier();
    W4_ETL(10);
#if defined(EXP_W4_TIMELINE) && defined(EXP_W4_EPI_TL)
    W4_WAIT_VMCNT(0);                                // (diagnostic) residual rows 0 .. 3 are in
    W4_ETL(12);
#endif
    typedef __attribute__((address_space(3))) char lds_char;
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    lds_char *const my = (lds_char *)smem + wave * 16384;
    // Epilogue arithmetic on channel PAIRS (v_pk_fma_f32 / v_pk_add_f32: no MFMA of this wave is in flight any more, and
    // a wave finishes 128 outputs per lane): value = alpha * acc + bias + residual; statistics are kept per pair -- (sum,
    // sum of squares) of (v0, v1) and of (v2, v3) -- and folded to quads at the end where the buffer wants quads.  Fixed
    // order, no atomics; the order is this kernel's own (its k order already is).
    f32x2_t s_lo[2][4], q_lo[2][4], s_hi[2][4], q_hi[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) s_lo[b][i] = q_lo[b][i] = s_hi[b][i] = q_hi[b][i] = f32x2_t{0.f, 0.f};
    // (round 6 experiments, tools/exp_build.sh: -DEXP_W4_EPI_SCALAR does the pair arithmetic as two scalar instructions -- the
    // PARTNER workgroup's MFMAs share this SIMD's issue port, where packed f32 costs more than the scalar pair,
    // MI355X_MICROARCH.md "packed f32 VALU"; -DEXP_W4_EPI_OVERLAP issues the two line stores of patch row r - 1 behind the
    // arithmetic of row r instead of all sixteen behind row 7.  Same values, same order of every sum either way.)
#ifdef EXP_W4_EPI_SCALAR
#define W4_FMA2(x, y, z) f32x2_t{__builtin_fmaf((x)[0], (y)[0], (z)[0]), __builtin_fmaf((x)[1], (y)[1], (z)[1])}
#define W4_ADD2(x, y) f32x2_t{(x)[0] + (y)[0], (x)[1] + (y)[1]}
#else
#define W4_FMA2(x, y, z) __builtin_elementwise_fma((x), (y), (z))
#define W4_ADD2(x, y) ((x) + (y))
#endif
    if (have) {
        const f32x2_t alpha2 = {p.alpha, p.alpha};
        const int c = lane & 7;
        // the two 128-byte line groups (8 pixels x 64 channels each) of patch row r of this wave: back from the wave's own
        // staging rows -- written and read by this wave only, so no barrier in between -- and out as whole lines
        auto store_lines = [&](int r) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int px = (2 * r + h) * 8 + (lane >> 3);  // pixel (row 8 wc + (px >> 4), column px & 15) of the patch
                const u32x4_t o = *(const __attribute__((address_space(3))) u32x4_t *)(my + px * 128 + ((c ^ (px & 7)) << 4));
                const int py = y0 + 8 * wc + (px >> 4), pxx = x0 + (px & 15);
                const int64_t pix = UPS ? ((int64_t)img * p.OH + 2 * py + PY) * p.OW + 2 * pxx + PX
                                        : ((int64_t)img * p.OH + py) * p.OW + pxx;
                *(u32x4_t *)((a16_t *)yb + pix * p.y_cs + p.y_co + nw + c * 8) = o;
            }
        };
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int px = r * 16 + frow;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 a = acc[i][r];
                f32x2_t lo = W4_FMA2((f32x2_t{a[0], a[1]}), alpha2, (f32x2_t{bias[i].x, bias[i].y}));
                f32x2_t hi = W4_FMA2((f32x2_t{a[2], a[3]}), alpha2, (f32x2_t{bias[i].z, bias[i].w}));
                lo = W4_ADD2(lo, (f32x2_t{a2f_lo(res[r][i].x), a2f_hi(res[r][i].x)}));
                hi = W4_ADD2(hi, (f32x2_t{a2f_lo(res[r][i].y), a2f_hi(res[r][i].y)}));
                if (ST != 0) {
                    s_lo[r >> 2][i] = W4_ADD2(s_lo[r >> 2][i], lo);
                    s_hi[r >> 2][i] = W4_ADD2(s_hi[r >> 2][i], hi);
                    q_lo[r >> 2][i] = W4_FMA2(lo, lo, q_lo[r >> 2][i]);
                    q_hi[r >> 2][i] = W4_FMA2(hi, hi, q_hi[r >> 2][i]);
                }
                const int s8 = i * 4 + fq;
                // (upsample phases only: onedc_conv_params.out_bf16 stores bf16 whatever the build's format -- round 5: the VAE's
                // 768 x 768 block may run in bf16 inside an f16 model, and its input is this kernel's output)
                const bool as_bf16 = UPS && p.out_bf16;
                *(__attribute__((address_space(3))) u32x2_t *)(my + px * 128 + (((s8 >> 1) ^ (px & 7)) << 4) + (s8 & 1) * 8) =
                    as_bf16 ? u32x2_t{pack_bf16_2(lo[0], lo[1]), pack_bf16_2(hi[0], hi[1])}
                            : u32x2_t{pack_a2(lo[0], lo[1]), pack_a2(hi[0], hi[1])};
            }
            if (r == 1 || r == 3) {
                load_res_row(r + 3);
                load_res_row(r + 4);
            }
#ifdef EXP_W4_EPI_OVERLAP
            if (r >= 1) {
                __builtin_amdgcn_sched_barrier(0);
                store_lines(r - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        W4_ETL(11);
#ifdef EXP_W4_EPI_OVERLAP
        store_lines(7);
#else
#pragma unroll
        for (int r = 0; r < 8; ++r) store_lines(r);
#endif
#ifdef EXP_W4_TIMELINE
        W4_TL(3);
#ifdef EXP_W4_EPI_TL
        W4_WAIT_VMCNT(0);
        W4_ETL(13);
#endif
        if (lane == 0) {     // diagnostic build: p.stats is a debug buffer [blocks][4 waves][8] of int64
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
            unsigned long long *dbg = (unsigned long long *)p.stats + ((int64_t)blockIdx.x * 4 + wave) * 16;
            dbg[0] = ((unsigned long long)xcc << 32) | hwid;
#pragma unroll
            for (int i = 0; i < 14; ++i) dbg[1 + i] = tl[i];
        }
        return;
#endif
        if (ST != 0) {                                 // two 64-pixel statistics blocks per wave: patch rows 8 wc + 4 b ..
            const int srow = (UPS ? (img * 4 + phase) * tiles_pi + pin : pt) * 4 + wc * 2;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sl = s_lo[b][i][0] + s_lo[b][i][1], ql = q_lo[b][i][0] + q_lo[b][i][1];
                    const float sh = s_hi[b][i][0] + s_hi[b][i][1], qh = q_hi[b][i][0] + q_hi[b][i][1];
                    if (ST == 4)
                        conv_stats_store_row(p, sl + sh, ql + qh, srow + b, nw + i * 16 + fq * 4, p.Cout, lane);
                    else
                        conv_stats_store_row(p, sl, ql, srow + b, nw + i * 16 + fq * 4, p.Cout, lane, sh, qh);
                }
        }
    }
#ifdef EXP_W4_PERSIST2
    __builtin_amdgcn_s_barrier();               // every wave has read its staging rows back before the next K loop's DMA lands
    }
#endif
}

}  // namespace

// Layers of the window kernel (onedc_conv_win_ok) whose epilogue is the fast linear one on whole 64-channel wave tiles;
// with gn_ab, room for the image's (a, b) pairs behind the ring while two workgroups still fit a CU.
bool onedc_conv_win4_ok(const onedc_conv_params &p) {
    // (second source: 32-bit byte offsets inside one image of x2, like the window's own -- a bigger image, e.g. 256 channels
    // at 3072 x 3072, is refused here and the caller runs the shortcut as its own launch)
    if (p.x2 && (!p.w2 || p.Cin2 < 32 || (p.Cin2 & 31) || ((p.x2_cs | p.x2_co) & 7) || p.stride != 1 ||
                 (int64_t)((p.Cout + 127) / 128 * 128) * p.Cin2 * 2 > 0x7fffffffll ||
                 (int64_t)p.OH * p.OW * p.x2_cs * 2 >= 0xfffff000ll))
        return false;
    if (p.x_bf16 && (p.up || p.in_f32)) return false;
    return !p.up && conv_epi_fast_ok(p, false) && p.epi == ONEDC_EPI_LINEAR && (p.Cout & 63) == 0 &&
           ((p.y_cs | p.y_co) & 7) == 0 && (!p.res || ((p.r_cs | p.r_co) & 3) == 0) &&
           (!p.gn_ab || W4_LDS + (int64_t)p.Cin * 8 <= 80 * 1024);
}

template <int MODE>
static int win4_launch(const onedc_conv_params &p, hipStream_t st, int lds) {
    constexpr bool UPS = MODE == 2;
#ifdef EXP_W4_ALONE     // (diagnostic, round 6) ONE workgroup per CU: what a K-tile costs a workgroup that has the matrix pipe to itself
    lds = 100 * 1024;
#endif
    const int tiles_n = UPS ? 4 * p.Cout / 128 : (p.Cout + 127) / 128;
    const int tiles_x = (UPS ? p.IW : p.OW) / 16, tiles_y = (UPS ? p.IH : p.OH) / 16;
    const int Ktot = (UPS ? 4 : 9) * p.Cin;
    const int64_t blocks = (int64_t)tiles_n * tiles_x * tiles_y * p.B;
    if (blocks > 0x7fffffff) return -1;
    const OnedcFastDiv dv_n((uint32_t)tiles_n), dv_pi((uint32_t)(tiles_x * tiles_y)), dv_x((uint32_t)(4 * tiles_x)),
        dv_c((uint32_t)p.Cout);
#ifdef EXP_W4_PERSIST2
    const dim3 grid((unsigned)((blocks + 1) / 2)), block(256);
#else
    const dim3 grid((unsigned)blocks), block(256);
#endif
#define W4_LAUNCH_BF(ST_, X2_, BF_)                                                                                  \
    hipLaunchKernelGGL((conv_win4_kernel<MODE, ST_, X2_, BF_>), grid, block, lds, st, p, tiles_n, tiles_x, tiles_y, Ktot, dv_n, \
                       dv_pi, dv_x, dv_c)
#ifdef ONEDC_ACT_F16
#define W4_LAUNCH(ST_, X2_)                                                                                          \
    do {                                                                                                             \
        if constexpr (!UPS) {                                                                                        \
            if (p.x_bf16) {                                                                                          \
                W4_LAUNCH_BF(ST_, X2_, true);                                                                        \
                break;                                                                                               \
            }                                                                                                        \
        }                                                                                                            \
        W4_LAUNCH_BF(ST_, X2_, false);                                                                               \
    } while (0)
#else                                                     // (the bf16 build multiplies bf16 anyway)
#define W4_LAUNCH(ST_, X2_) W4_LAUNCH_BF(ST_, X2_, false)
#endif
    if constexpr (!UPS) {
        if (p.x2) {
            if (!p.stats) W4_LAUNCH(0, true);
            else if (p.stats_gran == 2) W4_LAUNCH(2, true);
            else W4_LAUNCH(4, true);
            ONEDC_LAUNCH_CHECK();
            return 0;
        }
    }
    if (!p.stats) W4_LAUNCH(0, false);
    else if (p.stats_gran == 2) W4_LAUNCH(2, false);
    else W4_LAUNCH(4, false);
#undef W4_LAUNCH
#undef W4_LAUNCH_BF
    ONEDC_LAUNCH_CHECK();
    return 0;
}

static void win4_attrs() {
    static OnedcLdsAttrOnce once;
#ifdef EXP_W4_ALONE
#define W4_LDS_ATTR(n) (100 * 1024)
#else
#define W4_LDS_ATTR(n) (n)
#endif
    once.ensure({{(const void *)conv_win4_kernel<0, 0>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<0, 2>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<0, 4>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<1, 0>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<1, 2>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<1, 4>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<0, 0, true>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<0, 2, true>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<0, 4, true>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<1, 0, true>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<1, 2, true>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<1, 4, true>, W4_LDS_ATTR(80 * 1024)},
#ifdef ONEDC_ACT_F16
                 {(const void *)conv_win4_kernel<0, 0, false, true>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<0, 2, false, true>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<0, 4, false, true>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<1, 0, false, true>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<1, 2, false, true>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<1, 4, false, true>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<0, 0, true, true>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<0, 2, true, true>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<0, 4, true, true>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<1, 0, true, true>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<1, 2, true, true>, W4_LDS_ATTR(80 * 1024)},
                 {(const void *)conv_win4_kernel<1, 4, true, true>, W4_LDS_ATTR(80 * 1024)},
#endif
                 {(const void *)conv_win4_kernel<2, 0>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<2, 2>, W4_LDS_ATTR(W4_LDS)},
                 {(const void *)conv_win4_kernel<2, 4>, W4_LDS_ATTR(W4_LDS)}});
}

int onedc_conv_win4_launch(const onedc_conv_params &p, hipStream_t st) {
    win4_attrs();
    if (p.gn_ab) return win4_launch<1>(p, st, W4_LDS + p.Cin * 8);
    return win4_launch<0>(p, st, W4_LDS);
}

// up == 2 (x2 upsample + 3x3 as four 2x2 phase convolutions, gemm_conv_ups.hip) with the linear epilogue
bool onedc_conv_ups4_ok(const onedc_conv_params &p) { return p.epi == ONEDC_EPI_LINEAR && !p.gn_ab; }

int onedc_conv_ups4_launch(const onedc_conv_params &p, hipStream_t st) {
    win4_attrs();
    return win4_launch<2>(p, st, W4_LDS);
}
