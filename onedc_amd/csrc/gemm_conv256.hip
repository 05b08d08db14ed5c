// gemm_conv256.hip — the large-tile implicit-GEMM convolution: 256 channels x 256 pixels per
// workgroup, 8 wave64s, 8-phase schedule with the LDS-DMA prefetch left in flight across barriers.
//
// Same math, operand layout and epilogues as gemm_conv.hip (y = act(alpha * x*w^T + bias) + res,
// A operand = packed weight rows, B operand = shifted / zero-padded NHWC pixel rows, bf16 in, f32
// accumulate, v_mfma_f32_16x16x32_bf16), different machine mapping:
//
//   * waves form a 2 (channels) x 4 (pixels) grid, each wave owns 128 channels x 64 pixels =
//     8 x 4 accumulator tiles (128 VGPRs): 64 MFMAs per 64-deep K-tile against 24 KB of LDS reads,
//     and every byte the workgroup DMAs from L2 feeds twice the MFMAs of the 128 x 128 kernel
//   * a K-tile is consumed in 4 phases of 16 MFMAs (one 64 x 32 quadrant of the wave's tile each);
//     LDS holds two K-tiles of four 16 KB "half-tiles" (A^0, A^1, B^0, B^1: the rows the waves read in
//     the same phase sit together), each restaged by two global_load_lds_dwordx4 per wave as soon as
//     its last reader is two phases (or, for B^0, one phase and a retiring lgkmcnt) behind
//   * prefetch distance is three half-tiles: the only vector-memory wait in the loop is one counted
//     s_waitcnt vmcnt(6) per K-tile, barriers are raw s_barrier (no drain), and the two wave rows run
//     one barrier apart so that on every SIMD one wave is in its MFMA segment while the other issues
//     LDS reads and DMA
//   * LDS rows are 128 B with the 16-byte chunk XOR (row & 7) swizzle applied on the DMA source
//     address and on the ds_read_b128 fragment reads (conflict-free)
//
// Used by onedc_conv_gemm for the layers with >= 256 output rows and enough pixels to fill the chip
// (all VAE up-block convolutions with 256/512 channels, the UNet's 1280-channel layers, the FF GEMMs).
#include <type_traits>

#include "conv_common.h"

namespace {

constexpr int HALF = 16384;           // bytes of one half-tile (128 rows x 128 B)
constexpr int STAGE = 4 * HALF;       // one K-tile: A^0 A^1 B^0 B^1
constexpr int OFF_A0 = 0, OFF_A1 = HALF, OFF_B0 = 2 * HALF, OFF_B1 = 3 * HALF;

// Ablation builds (tools/exp_build.sh): EXP_NOLDSREAD keeps the fragments of the first phase for the whole loop,
// EXP_NODMA issues no LDS-DMA inside the main loop.  Results are wrong; only the timing is read.
#ifdef EXP_NOLDSREAD
#define ABL_READS(x) if (t == 0) { x }
#else
#define ABL_READS(x) x
#endif
#ifdef EXP_NODMA
#define ABL_DMA(x)
#else
#define ABL_DMA(x) x
#endif
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKMCNT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

// gridDim.y > 1: split-K.  Part blockIdx.y owns the contiguous K-tile range [part * T / parts, (part + 1) * T / parts)
// and stores its raw f32 sums y_part elements behind the previous part's (the launcher points p.y at the scratch buffer and
// strips bias / activation / residual: onedc_conv_splitk_launch below).
__global__ __launch_bounds__(512, 2) void conv_gemm256_kernel(const onedc_conv_params p, int tiles_n, int tiles_m,
                                                              int M, int Ktot, int64_t y_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int logical = xcd_logical_tile(blockIdx.x, tiles_n * tiles_m);
    const int n0 = (logical % tiles_n) * 256;
    const int m0 = (logical / tiles_n) * 256;

    const char *xb = (const char *)p.x + (int64_t)blockIdx.z * p.x_bs * 2;
    const char *wb = (const char *)p.w + (int64_t)blockIdx.z * p.w_bs * 2;
    const int w_rows = tiles_n * 256;              // packed weight rows are padded to the channel tile

    // ---- DMA geometry: one wave instruction = 8 LDS rows; a wave stages blocks (wave) and (wave + 8)
    // of every half-tile.  LDS row rho of A^h is tile channel (rho / 64) * 128 + h * 64 + rho % 64,
    // LDS row rho of B^h is tile pixel (rho / 32) * 64 + h * 32 + rho % 32.
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ lrow;           // swizzle on the source side
    const int t64 = wave * 8 + lrow;               // 0..63: LDS row of block `wave`; block wave + 8 is row t64 + 64
    // weights: row(h, i) = n0 + i * 128 + h * 64 + t64
    // weights through a buffer descriptor: per-lane byte offset of row n0 + t64, everything else in soffset
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(wb, (unsigned)((int64_t)w_rows * Ktot * 2));
    const unsigned wvoff = (unsigned)((int64_t)(n0 + t64) * Ktot * 2) + chunk * 16;
    const unsigned wrow64 = (unsigned)(64 * Ktot * 2);     // byte stride of 64 weight rows
    // pixels: pixel(h, i) = m0 + ((t64 >> 5) + 2 i) * 64 + h * 32 + (t64 & 31)
    const ConvPatch patch = conv_patch(p, m0);
    const int img0 = patch.on ? patch.img : m0 / (p.OH * p.OW);
    ConvPixSrc<4> px;                              // [h * 2 + i]: tile position ((t64 >> 5) + 2 i) * 64 + h * 32 + (t64 & 31)
    {
        int mrow[4];
#pragma unroll
        for (int hi = 0; hi < 4; ++hi)
            mrow[hi] = conv_tile_pixel(p, patch, m0, ((t64 >> 5) + 2 * (hi & 1)) * 64 + (hi >> 1) * 32 + (t64 & 31));
        px.init(p, mrow, M, img0);
    }
    // input through a buffer descriptor that starts at the first image this tile touches: byte offsets stay far
    // below 4 GB whatever the batch (a 256-channel 768x768 batch of 16 is 4.8 GB)
    const int64_t img_bytes = (int64_t)p.IH * p.IW * p.x_cs * 2;
    const int64_t rest = (int64_t)(p.B - img0) * img_bytes;
    const unsigned x_bytes = rest < 0xfffff000ll ? (unsigned)rest : 0xfffff000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(xb + img0 * img_bytes, x_bytes);
    const unsigned lane_off = (unsigned)(p.x_co * 2 + chunk * 16);
    const int pix_bytes = p.x_cs * 2;
    const int nc = p.Cin >> 6;                     // 64-channel K-tiles per tap
    const int T_all = p.KH * p.KW * nc;            // K-tiles of the layer
    const int t_first = (int)((int64_t)blockIdx.y * T_all / gridDim.y);               // (wave-uniform: scalar registers)
    const int T = (int)((int64_t)(blockIdx.y + 1) * T_all / gridDim.y) - t_first;    // K-tiles of this workgroup

    auto tap_sources = [&](int tap) {
        const int ky = tap / p.KW;
        px.tap_off(p, lane_off, pix_bytes, ky, tap - ky * p.KW, x_bytes);
    };

    // stage half-tile A^h of K-tile t of this workgroup (weights are k-contiguous: K-tile t starts at byte t * 128 of a row)
    auto stage_A = [&](int h, int t, char *stage) {
        char *dst = stage + (h ? OFF_A1 : OFF_A0) + wave * 1024;
        const unsigned so = (unsigned)(t_first + t) * 128 + (h ? wrow64 : 0);
        glds16_buf(w_rsrc, wvoff, so, dst);
        glds16_buf(w_rsrc, wvoff, so + 2 * wrow64, dst + 8192);
    };
    // stage half-tile B^h of the K-tile whose tap is loaded in psrc/pok, channel step ci
    auto stage_B = [&](int h, int ci, char *stage) {
        char *dst = stage + (h ? OFF_B1 : OFF_B0) + wave * 1024;
        const unsigned cbyte = (unsigned)ci * 128;
        glds16_buf(x_rsrc, px.off[2 * h], cbyte, dst);
        glds16_buf(x_rsrc, px.off[2 * h + 1], cbyte, dst + 8192);
    };

    // ---- accumulators: acc[a][i][b][j]: channels wr*128 + a*64 + i*16.., pixels wc*64 + b*32 + j*16.. ----
    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    // fragment byte offsets inside a half-tile: row (base + 16 t + frow), chunk (4 ks + fq) ^ (row & 7)
    const int fo0 = frow * 128 + ((fq ^ (lane & 7)) << 4);
    const int fo1 = frow * 128 + (((4 + fq) ^ (lane & 7)) << 4);
    const int a_off = wr * 64 * 128;               // this wave's 64 rows inside A^h
    const int b_off = wc * 32 * 128;               // this wave's 32 rows inside B^h

    a16x8 fa[4][2], fb0[2][2], fb1[2][2];

#define LOAD_A(stage, h)                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
        fa[i][0] = *(const a16x8 *)((stage) + ((h) ? OFF_A1 : OFF_A0) + a_off + i * 2048 + fo0); \
        fa[i][1] = *(const a16x8 *)((stage) + ((h) ? OFF_A1 : OFF_A0) + a_off + i * 2048 + fo1); \
    }
#define LOAD_B(dstf, stage, h)                                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
        dstf[j][0] = *(const a16x8 *)((stage) + ((h) ? OFF_B1 : OFF_B0) + b_off + j * 2048 + fo0); \
        dstf[j][1] = *(const a16x8 *)((stage) + ((h) ? OFF_B1 : OFF_B0) + b_off + j * 2048 + fo1); \
    }
#define MFMA_QUAD(a, b, fbx)                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
                acc[a][i][b][j] = ONEDC_MFMA16(fa[i][ks], fbx[j][ks],  \
                                                                          acc[a][i][b][j]); \
    __builtin_amdgcn_s_setprio(0);

    // ---- prologue: K-tile 0 whole, K-tile 1 minus A^1 (which phase 0 of tile 0 stages) ---------------
    char *const st0 = smem, *const st1 = smem + STAGE;
    int tap2 = t_first / nc, ci2 = t_first - tap2 * nc;   // (tap, channel step) of the K-tile the B stager is at
    tap_sources(tap2);
    stage_A(0, 0, st0);
    stage_B(0, ci2, st0);
    stage_A(1, 0, st0);
    stage_B(1, ci2, st0);
    auto advance2 = [&]() {                        // move the B stager to the next K-tile
        if (++ci2 == nc) {
            ci2 = 0;
            ++tap2;
            tap_sources(tap2);
        }
    };
    if (T > 1) {
        advance2();
        stage_B(0, ci2, st1);
        stage_A(0, 1, st1);
        stage_B(1, ci2, st1);
        WAIT_VMCNT(6);
    } else {
        WAIT_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();     // the second wave row runs one barrier behind

    // ---- main loop: 4 phases per K-tile -----------------------------------------------------------------
    for (int t = 0; t < T; ++t) {
        char *cur = smem + (t & 1) * STAGE;        // K-tile t (and the target of K-tile t + 2)
        char *nxt = smem + ((t + 1) & 1) * STAGE;  // K-tile t + 1
        const bool s1 = t + 1 < T, s2 = t + 2 < T;

        // phase 0: B-sub0 + A-sub0 reads; stage A^1(t+1)
        ABL_READS(LOAD_B(fb0, cur, 0))
        __builtin_amdgcn_sched_barrier(0);
        ABL_READS(LOAD_A(cur, 0))
        __builtin_amdgcn_sched_barrier(0);
        ABL_DMA(if (s1) stage_A(1, t + 1, nxt);)
        WAIT_LGKMCNT(8);                           // the four B reads (issued first) have returned: B^0 may be restaged next phase
        __builtin_amdgcn_s_barrier();
        WAIT_LGKMCNT(0);
        __builtin_amdgcn_sched_barrier(0);
        MFMA_QUAD(0, 0, fb0)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();

        // phase 1: B-sub1 reads; stage B^0(t+2)
        ABL_READS(LOAD_B(fb1, cur, 1))
        __builtin_amdgcn_sched_barrier(0);
        ABL_DMA(if (s2) {
            advance2();
            stage_B(0, ci2, cur);
        })
        __builtin_amdgcn_s_barrier();
        WAIT_LGKMCNT(0);
        __builtin_amdgcn_sched_barrier(0);
        MFMA_QUAD(0, 1, fb1)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();

        // phase 2: A-sub1 reads; stage A^0(t+2)
        ABL_READS(LOAD_A(cur, 1))
        __builtin_amdgcn_sched_barrier(0);
        ABL_DMA(if (s2) stage_A(0, t + 2, cur);)
        __builtin_amdgcn_s_barrier();
        WAIT_LGKMCNT(0);
        __builtin_amdgcn_sched_barrier(0);
        MFMA_QUAD(1, 1, fb1)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();

        // phase 3: no reads (B-sub0 is still in registers); stage B^1(t+2); K-tile t+1 must have landed
#ifdef EXP_NODMA
        WAIT_VMCNT(0);
#else
        if (s2) {
            stage_B(1, ci2, cur);
            WAIT_VMCNT(6);
        } else {
            WAIT_VMCNT(0);
        }
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        MFMA_QUAD(1, 0, fb0)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();     // rejoin the wave rows

    // ---- epilogue --------------------------------------------------------------------------------------
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    const int cout_store = glu ? p.Cout / 2 : p.Cout;
    char *yb = (char *)p.y + ((int64_t)blockIdx.z * p.y_bs + (int64_t)blockIdx.y * y_part) * (p.out_f32 ? 4 : 2);
    const char *rb = p.res ? (const char *)p.res + (int64_t)blockIdx.z * p.r_bs * 2 : nullptr;
    const int nw = n0 + wr * 128;
    float ssum[2][4], ssq[2][4], stat_dummy = 0.f;  // quad statistics only (the dispatcher keeps pair buffers away)
    EpiBias bias[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ssum[a][i] = ssq[a][i] = 0.f;
            bias[a][i] = conv_epi_bias(p, glu, nw + a * 64 + i * 16, fq);
        }
    auto epilogue = [&](auto mode) {
        typedef decltype(mode) EM;
    // two pixel halves (b), each: pass 1 = positions + residual loads of its 16 tiles, pass 2 = finish + store
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        EpiRes<false> res[2][2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = conv_tile_pixel(p, patch, m0, wc * 64 + b * 32 + j * 16 + frow);
            const ConvOutPix op = conv_out_pix(p, m);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (glu && (i & 1)) continue;
                    const EpiPos e = conv_epi_pos<EM>(p, glu, cout_store, nw + a * 64 + i * 16, fq, op, m < M);
                    conv_epi_load_res<false, EM>(p, rb, cout_store, e, res[j][a][i]);
                }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = conv_tile_pixel(p, patch, m0, wc * 64 + b * 32 + j * 16 + frow);
            const ConvOutPix op = conv_out_pix(p, m);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (glu && (i & 1)) continue;
                    const int i2 = (i + 1 < 4) ? i + 1 : i;
                    const EpiPos e = conv_epi_pos<EM>(p, glu, cout_store, nw + a * 64 + i * 16, fq, op, m < M);
                    conv_epi_finish<false, EM>(p, yb, glu, cout_store, acc[a][i][b][j], acc[a][i2][b][j], bias[a][i], e,
                                           res[j][a][i], ssum[a][i], ssq[a][i], stat_dummy, stat_dummy);
                }
        }
    }
    };
    // (run-time epilogue only: this kernel sits at 256 VGPRs and a second instantiation of the epilogue spills; launches
    // with a plain epilogue are sent to the 128x256 kernel instead, see onedc_conv_gemm)
    epilogue(EpiGeneric());
    if (p.stats) {                                 // wave-uniform
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (glu && (i & 1)) continue;
                const int n16 = nw + a * 64 + i * 16;
                conv_stats_store(p, ssum[a][i], ssq[a][i], m0 + wc * 64, (glu ? n16 / 2 : n16) + fq * 4, cout_store,
                                 lane, M);
            }
    }
}


// ---------------------------------------------------------------------------------------------------
// 128 channels x 256 pixels: the same machinery for layers whose channel count is not a multiple of
// 256 (the VAE's 128-channel up block at full resolution, the UNet's 320/640-channel levels).
// Waves 2 (channels) x 4 (pixels), wave tile 64 x 64 = 4 x 4 accumulator tiles; a K-tile is two phases
// of 16 MFMAs (pixel halves); LDS holds a ring of THREE K-tiles of 48 KB (A, B^0, B^1), so a K-tile is
// restaged two whole K-tiles ahead (every half-tile's last reader is >= 2 phases behind) and the loop's
// only vector-memory wait is again one counted vmcnt(6) per K-tile.
constexpr int STAGE3 = 3 * HALF;
constexpr int O3_A = 0, O3_B0 = HALF, O3_B1 = 2 * HALF;

__global__ __launch_bounds__(512, 2) void conv_gemm128x256_kernel(const onedc_conv_params p, int tiles_n,
                                                                  int tiles_m, int M, int Ktot) {
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
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int logical = xcd_logical_tile(blockIdx.x, tiles_n * tiles_m);
    const int n0 = (logical % tiles_n) * 128;
    const int m0 = (logical / tiles_n) * 256;

    const char *xb = (const char *)p.x + (int64_t)blockIdx.z * p.x_bs * 2;
    const char *wb = (const char *)p.w + (int64_t)blockIdx.z * p.w_bs * 2;
    const int w_rows = tiles_n * 128;              // packed weight rows are padded to the channel tile

    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ lrow;
    const int t64 = wave * 8 + lrow;
    // weights through a buffer descriptor: per-lane byte offset of row n0 + t64, everything else in soffset
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(wb, (unsigned)((int64_t)w_rows * Ktot * 2));
    const unsigned wvoff = (unsigned)((int64_t)(n0 + t64) * Ktot * 2) + chunk * 16;
    const unsigned wrow64 = (unsigned)(64 * Ktot * 2);     // byte stride of 64 weight rows
    const ConvPatch patch = conv_patch(p, m0);
    const int img0 = patch.on ? patch.img : m0 / (p.OH * p.OW);
    ConvPixSrc<4> px;                              // [h * 2 + i]: tile position ((t64 >> 5) + 2 i) * 64 + h * 32 + (t64 & 31)
    {
        int mrow[4];
#pragma unroll
        for (int hi = 0; hi < 4; ++hi)
            mrow[hi] = conv_tile_pixel(p, patch, m0, ((t64 >> 5) + 2 * (hi & 1)) * 64 + (hi >> 1) * 32 + (t64 & 31));
        px.init(p, mrow, M, img0);
    }
    // input through a buffer descriptor that starts at the first image this tile touches: byte offsets stay far
    // below 4 GB whatever the batch (a 256-channel 768x768 batch of 16 is 4.8 GB)
    const int64_t img_bytes = (int64_t)p.IH * p.IW * p.x_cs * 2;
    const int64_t rest = (int64_t)(p.B - img0) * img_bytes;
    const unsigned x_bytes = rest < 0xfffff000ll ? (unsigned)rest : 0xfffff000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(xb + img0 * img_bytes, x_bytes);
    const unsigned lane_off = (unsigned)(p.x_co * 2 + chunk * 16);
    const int pix_bytes = p.x_cs * 2;
    const int nc = p.Cin >> 6;                     // 64-channel K-tiles per tap
    const int T = p.KH * p.KW * nc;                // K-tiles

    auto tap_sources = [&](int tap) {
        const int ky = tap / p.KW;
        px.tap_off(p, lane_off, pix_bytes, ky, tap - ky * p.KW, x_bytes);
    };

    auto stage_A = [&](int t, char *stage) {
        char *dst = stage + O3_A + wave * 1024;
        const unsigned so = (unsigned)t * 128;
        glds16_buf(w_rsrc, wvoff, so, dst);
        glds16_buf(w_rsrc, wvoff, so + wrow64, dst + 8192);
    };
    auto stage_B = [&](int h, int ci, char *stage) {
        char *dst = stage + (h ? O3_B1 : O3_B0) + wave * 1024;
        const unsigned cbyte = (unsigned)ci * 128;
        glds16_buf(x_rsrc, px.off[2 * h], cbyte, dst);
        glds16_buf(x_rsrc, px.off[2 * h + 1], cbyte, dst + 8192);
    };

    f32x4 acc[4][2][2];                            // [i][b][j]: channels wr*64 + i*16.., pixels wc*64 + b*32 + j*16..
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int fo0 = frow * 128 + ((fq ^ (lane & 7)) << 4);
    const int fo1 = frow * 128 + (((4 + fq) ^ (lane & 7)) << 4);
    const int a_off = wr * 64 * 128;
    const int b_off = wc * 32 * 128;
    a16x8 fa[4][2], fb[2][2];

#define LOAD3_A(stage)                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                               \
        fa[i][0] = *(const a16x8 *)((stage) + O3_A + a_off + i * 2048 + fo0);    \
        fa[i][1] = *(const a16x8 *)((stage) + O3_A + a_off + i * 2048 + fo1);    \
    }
#define LOAD3_B(stage, h)                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
        fb[j][0] = *(const a16x8 *)((stage) + ((h) ? O3_B1 : O3_B0) + b_off + j * 2048 + fo0); \
        fb[j][1] = *(const a16x8 *)((stage) + ((h) ? O3_B1 : O3_B0) + b_off + j * 2048 + fo1); \
    }
#define MFMA3_HALF(b)                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
                acc[i][b][j] = ONEDC_MFMA16(fa[i][ks], fb[j][ks], acc[i][b][j]); \
    __builtin_amdgcn_s_setprio(0);

    // ---- prologue: K-tiles 0 and 1 ------------------------------------------------------------------
    int tap2 = 0, ci2 = 0;
    auto advance2 = [&]() {
        if (++ci2 == nc) {
            ci2 = 0;
            ++tap2;
            tap_sources(tap2);
        }
    };
    tap_sources(0);
    stage_A(0, smem);
    stage_B(0, 0, smem);
    stage_B(1, 0, smem);
    if (T > 1) {
        advance2();
        stage_A(1, smem + STAGE3);
        stage_B(0, ci2, smem + STAGE3);
        stage_B(1, ci2, smem + STAGE3);
        WAIT_VMCNT(6);
    } else {
        WAIT_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    STAMP(0);

    int si = 0;                                    // ring slot of K-tile t
    for (int t = 0; t < T; ++t) {
        char *cur = smem + si * STAGE3;
        const int s2 = si >= 1 ? si - 1 : 2;       // (si + 2) % 3: slot of K-tile t + 2 (= slot of K-tile t - 1)
        char *tgt = smem + s2 * STAGE3;
        const bool st2 = t + 2 < T;

        // phase 0: pixels 0..31 of the wave; stage A and B^0 of K-tile t + 2
        LOAD3_B(cur, 0)
        __builtin_amdgcn_sched_barrier(0);
        LOAD3_A(cur)
        __builtin_amdgcn_sched_barrier(0);
        if (st2) {
            advance2();
            stage_A(t + 2, tgt);
            stage_B(0, ci2, tgt);
        }
        STAMP(1);
        __builtin_amdgcn_s_barrier();
        STAMP(2);
        WAIT_LGKMCNT(0);
        STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        MFMA3_HALF(0)
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4);
        __builtin_amdgcn_s_barrier();
        STAMP(5);

        // phase 1: pixels 32..63; stage B^1 of K-tile t + 2; K-tile t + 1 must have landed
        LOAD3_B(cur, 1)
        __builtin_amdgcn_sched_barrier(0);
        if (st2) {
            stage_B(1, ci2, tgt);
            STAMP(1);
            WAIT_VMCNT(6);
        } else {
            STAMP(1);
            WAIT_VMCNT(0);
        }
        STAMP(6);
        __builtin_amdgcn_s_barrier();
        STAMP(2);
        WAIT_LGKMCNT(0);
        STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        MFMA3_HALF(1)
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4);
        __builtin_amdgcn_s_barrier();
        STAMP(5);
        si = si == 2 ? 0 : si + 1;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    STAMP(5);

    // ---- epilogue --------------------------------------------------------------------------------------
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    const int cout_store = glu ? p.Cout / 2 : p.Cout;
    char *yb = (char *)p.y + (int64_t)blockIdx.z * p.y_bs * (p.out_f32 ? 4 : 2);
    const char *rb = p.res ? (const char *)p.res + (int64_t)blockIdx.z * p.r_bs * 2 : nullptr;
    const int nw = n0 + wr * 64;
    float ssum[4], ssq[4], ssum_hi[4], ssq_hi[4];
    EpiBias bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ssum[i] = ssq[i] = ssum_hi[i] = ssq_hi[i] = 0.f;
        bias[i] = conv_epi_bias(p, glu, nw + i * 16, fq);
    }
    auto epilogue = [&](auto mode) {
        typedef decltype(mode) EM;
    EpiRes<false> res[2][2][4];                    // pass 1: positions + residual loads of all 16 tiles
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = conv_tile_pixel(p, patch, m0, wc * 64 + b * 32 + j * 16 + frow);
            const ConvOutPix op = conv_out_pix(p, m);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (glu && (i & 1)) continue;
                const EpiPos e = conv_epi_pos<EM>(p, glu, cout_store, nw + i * 16, fq, op, m < M);
                conv_epi_load_res<false, EM>(p, rb, cout_store, e, res[b][j][i]);
            }
        }
#pragma unroll
    for (int b = 0; b < 2; ++b)                    // pass 2: finish + store
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = conv_tile_pixel(p, patch, m0, wc * 64 + b * 32 + j * 16 + frow);
            const ConvOutPix op = conv_out_pix(p, m);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (glu && (i & 1)) continue;
                const int i2 = (i + 1 < 4) ? i + 1 : i;
                const EpiPos e = conv_epi_pos<EM>(p, glu, cout_store, nw + i * 16, fq, op, m < M);
                conv_epi_finish<false, EM>(p, yb, glu, cout_store, acc[i][b][j], acc[i2][b][j], bias[i], e, res[b][j][i],
                                       ssum[i], ssq[i], ssum_hi[i], ssq_hi[i]);
            }
        }
    };
    ONEDC_EPI_DISPATCH(p, true, epilogue);
#ifdef EXP_STAMP
    STAMP(7);
    if (lane == 0 && blockIdx.x < 2048) {          // diagnostic build: p.stats is a debug buffer [2048][8 waves][8]
        unsigned long long *dbg = (unsigned long long *)p.stats + ((int64_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 8; ++i) dbg[i] = st_acc[i];
    }
#else
    if (p.stats) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (glu && (i & 1)) continue;
            const int n16 = nw + i * 16;
            conv_stats_store(p, ssum[i], ssq[i], m0 + wc * 64, (glu ? n16 / 2 : n16) + fq * 4, cout_store, lane, M,
                             ssum_hi[i], ssq_hi[i]);
        }
    }
#endif
}

}  // namespace

// called by onedc_conv_gemm (gemm_conv.hip) for the shapes this tile suits; same contract
int onedc_conv_gemm256_launch(const onedc_conv_params &p, hipStream_t st) {
    static OnedcLdsAttrOnce once;
    once.ensure({{(const void *)conv_gemm256_kernel, 2 * STAGE}});
    const int64_t M64 = (int64_t)p.B * p.OH * p.OW;
    if (M64 > 0x7fffffff) return -1;
    const int M = (int)M64;
    const int tiles_n = (p.Cout + 255) / 256;
    const int tiles_m = (M + 255) / 256;
    const int Ktot = p.KH * p.KW * p.Cin;
    dim3 grid(tiles_n * tiles_m, 1, p.batch > 1 ? p.batch : 1);
    hipLaunchKernelGGL(conv_gemm256_kernel, grid, dim3(512), 2 * STAGE, st, p, tiles_n, tiles_m, M, Ktot, (int64_t)0);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

// ---- split-K (onedc_conv_params.split_k) ---------------------------------------------------------------------------------
// The UNet's 12x12 level (decoder_unet.py:204-293: two down resnets, the mid block, three up resnets) is 2 304 pixels per
// batch of 16 against K = 11 520 / 23 040: 45 tiles of 256 x 256 for 256 CUs, each 180 - 360 K-tiles long (round 3: 64-row
// tiles of the 4-wave kernel, 360 workgroups at 300 - 480 TFLOP/s, 3.85 ms per batch).  Here every tile is cut into `parts`
// K ranges -- 225 workgroups of 36 / 72 K-tiles on the 8-phase kernel above -- and the ranges are added by
// splitk_reduce_kernel in range order: one writer per scratch element, no atomics, the same bits on every launch, and the
// number of parts depends on the layer's shape only (never on B), so a picture alone decodes like one inside a batch.
namespace {
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, int parts, int64_t y_part,
                                                            const onedc_conv_params p, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (pixel, channel quad)
    if (i >= total4) return;
    const int c4 = p.Cout >> 2;
    const int64_t m = i / c4;
    const int n = (int)(i - m * c4) * 4;
    const float *src = ws + m * p.Cout + n;
    float4 s = *(const float4 *)src;
    for (int z = 1; z < parts; ++z) {                                // fixed order
        const float4 t = *(const float4 *)(src + z * y_part);
        s.x += t.x;
        s.y += t.y;
        s.z += t.z;
        s.w += t.w;
    }
    float v[4] = {s.x, s.y, s.z, s.w};
    const float4 b = p.bias ? *(const float4 *)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = __builtin_fmaf(v[k], p.alpha, bv[k]);
        if (p.epi == ONEDC_EPI_LEAKY) v[k] = leaky(v[k], p.slope);
        else if (p.epi == ONEDC_EPI_SILU) v[k] = silu_f(v[k]);
    }
    if (p.res) {
        const uint2 r = *(const uint2 *)((const a16_t *)p.res + m * p.r_cs + p.r_co + n);
        v[0] += a2f_lo(r.x);
        v[1] += a2f_hi(r.x);
        v[2] += a2f_lo(r.y);
        v[3] += a2f_hi(r.y);
    }
    const int64_t yo = m * p.y_cs + p.y_co + n;
    if (p.out_f32) *(float4 *)((float *)p.y + yo) = make_float4(v[0], v[1], v[2], v[3]);
    else *(uint2 *)((a16_t *)p.y + yo) = make_uint2(pack_a2(v[0], v[1]), pack_a2(v[2], v[3]));
}
}  // namespace

int64_t onedc_conv_splitk_bytes(const onedc_conv_params &p, int parts) {
    return (int64_t)parts * p.B * p.OH * p.OW * p.Cout * 4;
}

int onedc_conv_splitk_launch(const onedc_conv_params &p0, hipStream_t st, int parts) {
    static OnedcLdsAttrOnce once;
    once.ensure({{(const void *)conv_gemm256_kernel, 2 * STAGE}});
    const int64_t M64 = (int64_t)p0.B * p0.OH * p0.OW;
    if (M64 > 0x7fffffff) return -1;
    if (!p0.ws || p0.ws_bytes < onedc_conv_splitk_bytes(p0, parts)) return -15;
    const int M = (int)M64;
    onedc_conv_params p = p0;                      // pass 1: raw sums of every K range, f32, [part][pixel][Cout]
    p.y = p0.ws;
    p.out_f32 = 1;
    p.y_cs = p0.Cout;
    p.y_co = 0;
    p.bias = nullptr;
    p.res = nullptr;
    p.stats = nullptr;
    p.alpha = 1.f;
    p.epi = ONEDC_EPI_LINEAR;
    const int tiles_n = (p.Cout + 255) / 256;
    const int tiles_m = (M + 255) / 256;
    const int Ktot = p.KH * p.KW * p.Cin;
    const int64_t y_part = M64 * p.Cout;
    hipLaunchKernelGGL(conv_gemm256_kernel, dim3(tiles_n * tiles_m, parts, 1), dim3(512), 2 * STAGE, st, p, tiles_n, tiles_m, M,
                       Ktot, y_part);
    ONEDC_LAUNCH_CHECK();
    const int64_t total4 = M64 * (p0.Cout >> 2);   // pass 2: add the parts in order, epilogue of the layer
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, (const float *)p0.ws, parts,
                       y_part, p0, total4);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

int onedc_conv_gemm128x256_launch(const onedc_conv_params &p, hipStream_t st) {
    static OnedcLdsAttrOnce once;
    once.ensure({{(const void *)conv_gemm128x256_kernel, 3 * STAGE3}});
    const int64_t M64 = (int64_t)p.B * p.OH * p.OW;
    if (M64 > 0x7fffffff) return -1;
    const int M = (int)M64;
    const int tiles_n = (p.Cout + 127) / 128;
    const int tiles_m = (M + 255) / 256;
    const int Ktot = p.KH * p.KW * p.Cin;
    dim3 grid(tiles_n * tiles_m, 1, p.batch > 1 ? p.batch : 1);
    hipLaunchKernelGGL(conv_gemm128x256_kernel, grid, dim3(512), 3 * STAGE3, st, p, tiles_n, tiles_m, M, Ktot);
    ONEDC_LAUNCH_CHECK();
    return 0;
}
