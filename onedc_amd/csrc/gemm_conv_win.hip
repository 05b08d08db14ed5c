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
// STATUS: experimental, correct (tests/test_gpu_kernels.py::test_conv_win3x3_*), not auto-dispatched.
// Measured on MI355X against the 128x256 kernel (random data, B = 16): 128->128 @768x768 758-818 vs 720-730
// TFLOP/s, but 875-1010 vs 1080-1290 on the 256 / 512-channel layers: cutting the LDS-DMA bytes did not move
// the bound of those layers ("what did not work").
// Same operand layout, accumulation order per output (tap-major inside a chunk, chunks outer -- see note
// below), epilogues and statistics as the other kernels.
// Note: the k order is (chunk, tap) here against (tap, chunk) in the generic kernels, so results agree to
// fp32 summation-order rounding, not bit for bit; each kernel on its own is deterministic and
// batch-invariant.
#include "conv_common.h"

namespace {

constexpr int WIN_A = 16384;                 // one weight tile: 128 rows x 128 B
constexpr int WIN_NA = 4;                    // weight ring depth
constexpr int WIN_ROWS = 328;                // window rows, padded to 41 blocks of 8
constexpr int WIN_W = WIN_ROWS * 128;        // 41984 B
constexpr int WIN_LDS = WIN_NA * WIN_A + 2 * WIN_W;   // 149504 B

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKMCNT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

template <bool UP>
__global__ __launch_bounds__(512, 2) void conv_win_kernel(const onedc_conv_params p, int tiles_n, int tiles_x,
                                                          int tiles_y, int Ktot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef EXP_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#define STAMP(i)                                                     \
    do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                           \
        unsigned long long t__;                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); \
        st_acc[i] += t__ - st_last;                                  \
        st_last = t__;                                               \
        __builtin_amdgcn_sched_barrier(0);                           \
    } while (0)
#else
#define STAMP(i)
#endif
    char *const lds_a = smem;
    char *const lds_w = smem + WIN_NA * WIN_A;
    constexpr int WW = UP ? 10 : 18;               // window width (and height)
    constexpr int NWB = (WW * WW + 7) / 8;          // 8-row DMA blocks per window: 41 / 13

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- patch of this workgroup: channel tile fastest, then x, y, image --------------------------------
    const int tiles_pi = tiles_x * tiles_y;
    const int logical = xcd_logical_tile(blockIdx.x, tiles_n * tiles_pi * p.B);
    const int n0 = (logical % tiles_n) * 128;
    const int pt = logical / tiles_n;               // patch index over all images
    const int img = pt / tiles_pi, pin = pt - img * tiles_pi;
    const int ty = pin / tiles_x, tx = pin - ty * tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;           // output origin of the patch

    const char *xb = (const char *)p.x;
    const char *wb = (const char *)p.w;
    const char *zeros = (const char *)p.zeros;

    // ---- DMA geometry ---------------------------------------------------------------------------------------
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ lrow;            // swizzle on the source side (row & 7 == lrow for every block)
    // weights: 16 blocks of 8 rows per K-tile, two per wave: rows (wave) * 8 + lrow and + 64
    const char *wsrc = wb + ((int64_t)(n0 + wave * 8 + lrow) * Ktot) * 2 + chunk * 16;
    const int64_t wrow64 = (int64_t)64 * Ktot * 2;
    // window: block k of this wave covers window rows (wave + 8 k) * 8 + lrow
    const int nwb = (NWB - wave + 7) >> 3;           // blocks this wave stages per window (wave-uniform)
    constexpr int MAXWB = (NWB + 7) / 8;              // 6 / 2
    const char *wsrcp[MAXWB];
    unsigned wok = 0;
    {
        const int wy0 = UP ? (y0 >> 1) - 1 : y0 - 1, wx0 = UP ? (x0 >> 1) - 1 : x0 - 1;
        const int64_t img_pix = (int64_t)img * p.IH * p.IW;
#pragma unroll
        for (int k = 0; k < MAXWB; ++k) {
            const int row = (wave + 8 * k) * 8 + lrow;
            const int wy = row / WW, wx = row - wy * WW;
            const int iy = wy0 + wy, ix = wx0 + wx;
            const bool ok = row < WW * WW && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            wsrcp[k] = xb + ((img_pix + (int64_t)iy * p.IW + ix) * p.x_cs + p.x_co) * 2 + chunk * 16;
            wok |= ok ? (1u << k) : 0u;
        }
    }
    const int nc = p.Cin >> 6;                       // 64-channel chunks
    const int T = 9 * nc;                            // K-tiles

    // K-tile t = (chunk ci, tap): weights are [row][tap][cin] -> byte offset (tap * Cin + ci * 64) * 2
    auto stage_A = [&](int tap, int ci, int slot) {
        char *dst = lds_a + slot * WIN_A + wave * 1024;
        const char *s = wsrc + ((int64_t)tap * p.Cin + ci * 64) * 2;
        glds16(s, dst);
        glds16(s + wrow64, dst + 8192);
    };
    auto stage_W = [&](int k, int ci, int wbuf) {      // block k of this wave, window of chunk ci
        char *dst = lds_w + wbuf * WIN_W + (wave + 8 * k) * 1024;
        const char *z = zeros + chunk * 16;
        const char *src = z;
#pragma unroll
        for (int kk = 0; kk < MAXWB; ++kk)             // static register indexing
            if (kk == k) src = ((wok >> kk) & 1) ? wsrcp[kk] + ci * 128 : z;
        glds16(src, dst);
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
    const int fo0 = frow * 128 + ((fq ^ (lane & 7)) << 4);           // weight fragments: rows 16 i + frow
    const int fo1 = frow * 128 + (((4 + fq) ^ (lane & 7)) << 4);
    const int a_off = wr * 64 * 128;

    // window row of output (row 4 wc + bj, column frow) under tap (ky, kx)
    auto win_row = [&](int bj, int ky, int kx) {
        const int ly = 4 * wc + bj;
        if (UP) return ((ly + ky + 1) >> 1) * WW + ((frow + kx + 1) >> 1);
        return (ly + ky) * WW + frow + kx;
    };
    // Fragments of one K-tile: 8 weight + 8 pixel ds_read_b128.  Two register sets: the reads of K-tile
    // t + 1 are issued BEFORE the 32 MFMAs of K-tile t, so LDS latency (hundreds of cycles under load) is
    // covered by the wave's own matrix work instead of being waited for between two barriers.
    struct Frags {
        bf16x8 a[4][2], b[4][2];
    };
    // part 0: the first 15 reads, part 1: the 16th (lgkmcnt is a 4-bit counter: at most 15 reads can be
    // left in flight by a counted wait, so the last read of K-tile t + 1 is issued after the wait for K-tile t)
    auto load_frags = [&](Frags &f, int slot, int wbuf, int tap, int part) {
        if (part == 1) {
            f.a[3][1] = *(const bf16x8 *)(lds_a + slot * WIN_A + a_off + 3 * 2048 + fo1);
            return;
        }
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
            const int w = win_row(bj, ky, kx);
            const char *rp = lds_w + wbuf * WIN_W + w * 128;
            f.b[bj][0] = *(const bf16x8 *)(rp + ((fq ^ (w & 7)) << 4));
            f.b[bj][1] = *(const bf16x8 *)(rp + (((4 + fq) ^ (w & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.a[i][0] = *(const bf16x8 *)(lds_a + slot * WIN_A + a_off + i * 2048 + fo0);
            if (i < 3) f.a[i][1] = *(const bf16x8 *)(lds_a + slot * WIN_A + a_off + i * 2048 + fo1);
        }
    };
    auto mfma_tile = [&](const Frags &f) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj)
                    acc[i][bj >> 1][bj & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[i][ks], f.b[bj][ks],
                                                                                    acc[i][bj >> 1][bj & 1], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: window of chunk 0, weights of K-tiles 0..3; fragments of K-tile 0 ---------------------
#pragma unroll
    for (int k = 0; k < MAXWB; ++k)
        if (k < nwb) stage_W(k, 0, 0);
    stage_A(0, 0, 0);
    stage_A(1, 0, 1);                                  // T >= 9 always
    stage_A(2, 0, 2);
    stage_A(3, 0, 3);
    WAIT_VMCNT(4);                                     // window 0, K-tiles 0 and 1 have landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    load_frags(f0, 0, 0, 0, 0);
    load_frags(f0, 0, 0, 0, 1);
    STAMP(0);

    // ---- main loop: one K-tile (= one tap of one chunk, 32 MFMAs per wave) per barrier ---------------------
    // iteration t:  issue the fragment reads of K-tile t + 1 | wait for the fragments of K-tile t | 32 MFMAs |
    //               wait until this wave's part of K-tile t + 2 has landed | barrier | DMA: one window block
    //               of the next chunk (taps 0..5) and the weights of K-tile t + 4 into the slot K-tile t just left
    int tap = 0, ci = 0;                               // K-tile t
    int tap4 = 4, ci4 = 0;                             // K-tile t + 4
    bool w_prev = false;                               // did iteration t - 1 issue a window block?
    auto iteration = [&](int t, Frags &cur, Frags &nxt) {
        int tap1 = tap + 1, ci1 = ci;
        if (tap1 == 9) {
            tap1 = 0;
            ++ci1;
        }
        if (t + 1 < T) load_frags(nxt, (t + 1) & 3, ci1 & 1, tap1, 0);
        STAMP(1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < T) WAIT_LGKMCNT(15);               // the 15 reads just issued may stay in flight
        else WAIT_LGKMCNT(0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < T) load_frags(nxt, (t + 1) & 3, ci1 & 1, tap1, 1);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
        mfma_tile(cur);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4);
        // K-tile t + 2 (weights, and at a chunk seam the window) must be in LDS before the next iteration reads it
        if (t + 4 < T) {
            if (w_prev) WAIT_VMCNT(3);
            else WAIT_VMCNT(2);
        } else {
            WAIT_VMCNT(0);
        }
        STAMP(6);
        __builtin_amdgcn_s_barrier();
        STAMP(5);
        const bool w_now = tap < nwb && ci + 1 < nc;   // window block `tap` of the next chunk
        if (w_now) stage_W(tap, ci + 1, (ci & 1) ^ 1);
        if (t + 4 < T) stage_A(tap4, ci4, t & 3);
        w_prev = w_now;
        if (++tap == 9) {
            tap = 0;
            ++ci;
        }
        if (++tap4 == 9) {
            tap4 = 0;
            ++ci4;
        }
    };
    for (int t = 0; t < T; t += 2) {                   // T = 9 nc may be odd: the second half is guarded
        iteration(t, f0, f1);
        if (t + 1 < T) iteration(t + 1, f1, f0);
    }

    // ---- epilogue -----------------------------------------------------------------------------------------------
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    const int cout_store = glu ? p.Cout / 2 : p.Cout;
    char *yb = (char *)p.y;
    const char *rb = (const char *)p.res;
    const int nw = n0 + wr * 64;
    float ssum[4], ssq[4], ssum_hi[4], ssq_hi[4];
    EpiBias bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ssum[i] = ssq[i] = ssum_hi[i] = ssq_hi[i] = 0.f;
        bias[i] = conv_epi_bias(p, glu, nw + i * 16, fq);
    }
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
#ifdef EXP_STAMP
    STAMP(7);
    if (lane == 0 && blockIdx.x < 2048) {              // diagnostic build: p.stats is a debug buffer [2048][8 waves][8]
        unsigned long long *dbg = (unsigned long long *)p.stats + ((int64_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 8; ++i) dbg[i] = st_acc[i];
    }
    if (p.alpha != 12345.f) return;
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
           (int64_t)p.B * p.IH * p.IW * p.x_cs < (int64_t)1 << 40;
}

int onedc_conv_win_launch(const onedc_conv_params &p, hipStream_t st) {
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute((const void *)conv_win_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS);
        (void)hipFuncSetAttribute((const void *)conv_win_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS);
        configured = true;
    }
    const int tiles_n = (p.Cout + 127) / 128;
    const int tiles_x = p.OW / 16, tiles_y = p.OH / 16;
    const int Ktot = 9 * p.Cin;
    const int64_t blocks = (int64_t)tiles_n * tiles_x * tiles_y * p.B;
    if (blocks > 0x7fffffff) return -1;
    if (p.up)
        hipLaunchKernelGGL(conv_win_kernel<true>, dim3((unsigned)blocks), dim3(512), WIN_LDS, st, p, tiles_n, tiles_x,
                           tiles_y, Ktot);
    else
        hipLaunchKernelGGL(conv_win_kernel<false>, dim3((unsigned)blocks), dim3(512), WIN_LDS, st, p, tiles_n, tiles_x,
                           tiles_y, Ktot);
    ONEDC_LAUNCH_CHECK();
    return 0;
}
