// gemm_conv.hip — implicit-GEMM convolution / linear layer on the CDNA4 matrix cores.
//
//   y[m, n] = act(alpha * sum_k x[m, k] * w[n, k] + bias[n]) + res[m, n]
//
// m = output pixel (b, oy, ox), n = output channel, k = (ky, kx, cin).  One workgroup of four
// wave64s owns a TN(channels) x TM(pixels) output tile.  Per 64-channel K-step the weight rows
// and the (shifted, zero-padded) pixel rows of the tile are streamed global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write traffic), double buffered so the
// next step's bytes are in flight while the matrix cores work on the current one.  LDS rows are
// 128 B (64 bf16); the 16-byte chunk c of row r is stored at position c ^ (r & 7), applied on the
// per-lane SOURCE address (the DMA destination is lane-linear) and again on the ds_read_b128
// fragment reads, which makes every fragment read bank-conflict free.
// MFMA orientation: A operand = weights (rows = channels), B operand = pixels, so each lane ends
// up with 4 consecutive channels of one pixel -> 8-byte NHWC stores and lane-local epilogues.
//
// Replaces nn.Conv2d / nn.Linear everywhere on the reference decode path (see onedc_hip.h).
#include <type_traits>

#include "conv_common.h"

namespace {

// Line stores of the 4-wave kernels' epilogue (conv_common.h: conv_line_put / conv_line_store), shared by conv_gemm_kernel
// and conv_lin_kernel.  Taken when the epilogue mode is a fast 16-bit one, the wave tile is 64 pixels wide and every channel
// of the wave exists and starts on a 16-byte boundary of the output pixel; wave-uniform.  The K loop's LDS is free behind a
// barrier (the other waves may still be reading fragments of the last step).
#define ONEDC_LINE_SETUP(F32_)                                                                                         \
    constexpr int CWL = EM::glu ? T::WTN / 2 : T::WTN;         /* stored channels of a wave */                         \
    const int n_store0 = EM::glu ? nw / 2 : nw;                                                                        \
    const bool line = EM::fast && !(F32_) && T::WTM == 64 && CWL >= 16 && (CWL & (CWL - 1)) == 0 && nw + T::WTN <= p.Cout && \
                      ((p.y_cs | (p.y_co + n_store0)) & 7) == 0 && p.tile != 132;                                      \
    char *const my_line = smem + wave * (64 * CWL * 2);                                                                \
    if (EM::fast && !(F32_)) __syncthreads();
#define ONEDC_LINE_PUT()                                                                                               \
    if constexpr (EM::fast && T::WTM == 64 && (EM::glu ? T::WTN / 2 : T::WTN) >= 16 && ((EM::glu ? T::WTN / 2 : T::WTN) & ((EM::glu ? T::WTN / 2 : T::WTN) - 1)) == 0) { \
        if (line) conv_line_put<CWL>(my_line, j * 16 + frow, (EM::glu ? (i >> 1) : i) * 4 + fq, lv0, lv1);                   \
    }
#define ONEDC_LINE_STORE()                                                                                             \
    if constexpr (EM::fast && T::WTM == 64 && (EM::glu ? T::WTN / 2 : T::WTN) >= 16 && ((EM::glu ? T::WTN / 2 : T::WTN) & ((EM::glu ? T::WTN / 2 : T::WTN) - 1)) == 0) { \
        if (line)                                                                                                      \
            conv_line_store<CWL>(my_line, (a16_t *)yb, p.y_cs, (int64_t)p.y_co + n_store0, m0 + wm * T::WTM, M, lane); \
    }

template <int TN_, int TM_, int WN_, int WM_>
struct Tile {
    static constexpr int TN = TN_, TM = TM_, WN = WN_, WM = WM_;
    static constexpr int ROWS = TN + TM;
    static constexpr int BUF = ROWS * 128;          // bytes per stage
    static constexpr int WTN = TN / WN, WTM = TM / WM;
    static constexpr int FI = WTN / 16, FJ = WTM / 16;
    static constexpr int WBLK = TN / 8;             // 8-row DMA blocks of weights
    static constexpr int NWI = (WBLK + 3) / 4;      // weight DMA instructions per wave per step
    static constexpr int NPI = TM / 32;             // pixel DMA instructions per wave per step
    static_assert(WN * WM == 4 && TM % 32 == 0 && TN % 8 == 0, "tile shape");
};

// F32 = true: the same tile machinery on f32 activations and weights (exact f32 products,
// v_mfma_f32_16x16x4_f32).  A 128-byte LDS row then holds 32 channels instead of 64; each lane's
// 16-byte fragment chunk feeds four k=4 MFMAs (element e of every lane's chunk is one k-slice, the
// same permutation on both operands).  Used for the prior networks, whose outputs feed integer
// decisions (SURVEY hard part 2).
// SPLIT (with F32; onedc_conv_params.in_f32 == 2): the same f32 activations, f32 accumulation and tiles at the 16-bit matrix
// rate.  Every operand is written as a PAIR of IEEE halves, v = h + l * 2^-11 (h = half(v), l = half((v - h) * 2^11): 22
// significant bits, relative representation error <= 2^-23), the weights once by the host (pack.pack_matrix_split16: per
// 32-channel K-step 64 B of h then 64 B of l, the same 128 bytes an f32 row has), the activations in registers behind the
// fragment read; the product is three v_mfma_f32_16x16x32_f16 per K-step and tile -- h.h into one accumulator, h.l + l.h
// into a second that joins scaled by 2^-11 at the end; l.l (2^-22 of the product) is dropped -- instead of eight
// v_mfma_f32_16x16x4_f32: 48 instead of 256 matrix-pipe cycles.  Each product is exact in f32 (11 x 11 bits), so a sum differs
// from the f32 fma chain by the operands' 2^-23, which is below the rounding that chain accumulates itself over K >= 128.
// Fixed k order, no atomics: as deterministic and batch-invariant as the exact form; operands must lie inside the half range
// (|v| < 65504: the prior networks stay below 600 with random-init weights, 11 with damped ones -- DESIGN.md section 3).
typedef __attribute__((ext_vector_type(8))) _Float16 split_h8;
typedef __attribute__((ext_vector_type(2))) _Float16 split_h2;
__device__ __forceinline__ void split16_pair(float v0, float v1, uint32_t &h, uint32_t &l) {
    const split_h2 hh = {(_Float16)v0, (_Float16)v1};                       // round to nearest even
    const float d0 = (v0 - (float)hh[0]) * 2048.f, d1 = (v1 - (float)hh[1]) * 2048.f;   // exact differences
    const split_h2 ll = {(_Float16)d0, (_Float16)d1};
    h = __builtin_bit_cast(uint32_t, hh);
    l = __builtin_bit_cast(uint32_t, ll);
}

template <class T, bool F32, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const onedc_conv_params p, int tiles_n,
                                                           int tiles_m, int M, int Ktot) {
    static_assert(!SPLIT || F32, "the split form is a form of the f32 path");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TN = T::TN, TM = T::TM;
    constexpr int ES = F32 ? 4 : 2;                // bytes per element
    constexpr int KSH = F32 ? 5 : 6;               // log2(channels per 128-byte row)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % T::WN, wm = wave / T::WN;

    // ---- tile of this workgroup; consecutive logical tiles share an XCD (and its L2) -------------
    const int logical = xcd_logical_tile(blockIdx.x, tiles_n * tiles_m);
    const int n0 = (logical % tiles_n) * TN;
    const int m0 = (logical / tiles_n) * TM;

    const char *xb = (const char *)p.x + (int64_t)blockIdx.z * p.x_bs * ES;
    const char *wb = (const char *)p.w + (int64_t)blockIdx.z * p.w_bs * ES;
    const char *zeros = (const char *)p.zeros;

    // ---- per-lane DMA sources -----------------------------------------------------------------
    const int lrow = lane >> 3;                    // row inside an 8-row DMA block
    const int chunk = (lane & 7) ^ lrow;           // swizzle on the source side
    const char *wsrc[T::NWI];
#pragma unroll
    for (int i = 0; i < T::NWI; ++i) {
        const int row = (wave + 4 * i) * 8 + lrow;
        wsrc[i] = wb + ((int64_t)(n0 + row) * Ktot) * ES + chunk * 16;
    }
    const int nc = p.Cin >> KSH;                   // 128-byte steps per tap
    const int steps = p.KH * p.KW * nc;
    // weights of step 0 leave first: their addresses need nothing but the channel tile, the pixel sources below
    // (a decode per row unless the layer is 1x1) are then worked out under that latency
    auto issue_w = [&](int tap, int ci, char *buf) {
        const int64_t kbyte = (int64_t)tap * p.Cin * ES + ci * 128;
#pragma unroll
        for (int i = 0; i < T::NWI; ++i) {
            const int blk = wave + 4 * i;          // wave-uniform
            if (T::WBLK % 4 == 0 || blk < T::WBLK) glds16(wsrc[i] + kbyte, buf + blk * 1024);
        }
    };
    issue_w(0, 0, smem);
    __builtin_amdgcn_sched_barrier(0);

    ConvPixSrc<T::NPI> px;
    const char *xb0 = xb + p.x_co * ES + chunk * 16;
    const int pix_bytes = p.x_cs * ES;
    const bool linear = conv_is_linear(p);
    {
        int mrow[T::NPI];
#pragma unroll
        for (int i = 0; i < T::NPI; ++i) mrow[i] = m0 + (wave + 4 * i) * 8 + lrow;
        if (linear) px.init_linear(xb0, pix_bytes, mrow, M);
        else px.init(p, mrow, M);
    }

    auto tap_sources = [&](int tap) {
        const int ky = tap / p.KW;
        px.tap(p, xb0, pix_bytes, ky, tap - ky * p.KW);
    };
    auto issue_px = [&](int ci, char *buf) {
        const int cbyte = ci * 128;
#pragma unroll
        for (int i = 0; i < T::NPI; ++i) {
            glds16(((px.ok >> i) & 1) ? px.src[i] + cbyte : zeros + chunk * 16,
                   buf + (T::WBLK + wave + 4 * i) * 1024);
        }
    };
    auto issue = [&](int step, int tap, int ci, char *buf) {
        issue_w(tap, ci, buf);
        issue_px(ci, buf);
        (void)step;
    };
    if (!linear) tap_sources(0);
    issue_px(0, smem);
    __builtin_amdgcn_sched_barrier(0);

    // ---- accumulators ---------------------------------------------------------------------------
    f32x4 acc[T::FI][T::FJ];
    f32x4 acc_x[SPLIT ? T::FI : 1][SPLIT ? T::FJ : 1];      // SPLIT: the cross terms h.l + l.h, scaled 2^11
#pragma unroll
    for (int i = 0; i < T::FI; ++i)
#pragma unroll
        for (int j = 0; j < T::FJ; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) acc_x[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    const int frow = lane & 15, fq = lane >> 4;
    int foff[2];                                    // fragment byte offset inside a 16-row group
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * 128 + (((ks * 4 + fq) ^ (lane & 7)) << 4);
    const int a_base = (wn * T::WTN) * 128;
    const int b_base = (TN + wm * T::WTM) * 128;

    // fragment addresses exist before the first wait (an empty asm the optimiser cannot look through keeps their
    // arithmetic from sinking behind it, onto the critical path of every workgroup)
    asm volatile("" : "+v"(foff[0]), "+v"(foff[1]));
    // ---- main loop: one barrier per K-step, next step's DMA in flight during the MFMAs -------------
    int tap = 0, ci = 0;
    for (int s = 0; s < steps; ++s) {
        char *cur = smem + (s & 1) * T::BUF;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < steps) {
            if (++ci == nc) {
                ci = 0;
                ++tap;
                tap_sources(tap);
            }
            issue(s + 1, tap, ci, smem + ((s + 1) & 1) * T::BUF);
        }
        if constexpr (SPLIT) {
            // one MFMA k step per 32-channel K-step.  Slot 8 fq + j of the k dimension is channel 4 fq + j (j < 4) or
            // 16 + 4 fq + j - 4: what a lane's two f32 chunks (fq, 4 + fq) hold -- the host packs the weight halves in
            // that order, so chunk fq of the row is the lane's h fragment and chunk 4 + fq its l fragment
            split_h8 ah[T::FI], al[T::FI], bh[T::FJ], bl[T::FJ];
#pragma unroll
            for (int i = 0; i < T::FI; ++i) {
                ah[i] = *(const split_h8 *)(cur + a_base + i * 2048 + foff[0]);
                al[i] = *(const split_h8 *)(cur + a_base + i * 2048 + foff[1]);
            }
#pragma unroll
            for (int j = 0; j < T::FJ; ++j) {
                const f32x4 v0 = *(const f32x4 *)(cur + b_base + j * 2048 + foff[0]);
                const f32x4 v1 = *(const f32x4 *)(cur + b_base + j * 2048 + foff[1]);
                uint32_t h[4], l[4];
                split16_pair(v0[0], v0[1], h[0], l[0]);
                split16_pair(v0[2], v0[3], h[1], l[1]);
                split16_pair(v1[0], v1[1], h[2], l[2]);
                split16_pair(v1[2], v1[3], h[3], l[3]);
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
                bh[j] = __builtin_bit_cast(split_h8, u32x4_{h[0], h[1], h[2], h[3]});
                bl[j] = __builtin_bit_cast(split_h8, u32x4_{l[0], l[1], l[2], l[3]});
            }
#pragma unroll
            for (int i = 0; i < T::FI; ++i)
#pragma unroll
                for (int j = 0; j < T::FJ; ++j) {
                    acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc_x[i][j], 0, 0, 0);
                    acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc_x[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        } else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (F32) {
                f32x4 a[T::FI], b[T::FJ];
#pragma unroll
                for (int i = 0; i < T::FI; ++i)
                    a[i] = *(const f32x4 *)(cur + a_base + i * 2048 + foff[ks]);
#pragma unroll
                for (int j = 0; j < T::FJ; ++j)
                    b[j] = *(const f32x4 *)(cur + b_base + j * 2048 + foff[ks]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < T::FI; ++i)
#pragma unroll
                        for (int j = 0; j < T::FJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            } else {
                a16x8 a[T::FI], b[T::FJ];
#pragma unroll
                for (int i = 0; i < T::FI; ++i)
                    a[i] = *(const a16x8 *)(cur + a_base + i * 2048 + foff[ks]);
#pragma unroll
                for (int j = 0; j < T::FJ; ++j)
                    b[j] = *(const a16x8 *)(cur + b_base + j * 2048 + foff[ks]);
#pragma unroll
                for (int i = 0; i < T::FI; ++i)
#pragma unroll
                    for (int j = 0; j < T::FJ; ++j)
                        acc[i][j] = ONEDC_MFMA16(a[i], b[j], acc[i][j]);
            }
        }
    }

    if constexpr (SPLIT) {
#pragma unroll
        for (int i = 0; i < T::FI; ++i)
#pragma unroll
            for (int j = 0; j < T::FJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = __builtin_fmaf(acc_x[i][j][e], 1.f / 2048.f, acc[i][j][e]);
    }
    // ---- epilogue: lane holds channels 4*fq..4*fq+3 of pixel (lane & 15) of every 16x16 tile --------
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    const int cout_store = glu ? p.Cout / 2 : p.Cout;
    char *yb = (char *)p.y + (int64_t)blockIdx.z * p.y_bs * (p.out_f32 ? 4 : 2);
    const char *rb = p.res ? (const char *)p.res + (int64_t)blockIdx.z * p.r_bs * ES : nullptr;
    const int nw = n0 + wn * T::WTN;               // first packed row of this wave

    float ssum[T::FI], ssq[T::FI], ssum_hi[T::FI], ssq_hi[T::FI];
    EpiBias bias[T::FI];
#pragma unroll
    for (int i = 0; i < T::FI; ++i) {
        ssum[i] = ssq[i] = ssum_hi[i] = ssq_hi[i] = 0.f;
        bias[i] = conv_epi_bias(p, glu, nw + i * 16, fq);
    }
    // pass 1: positions + residual loads of every tile; pass 2: finish + store
    auto epilogue = [&](auto mode) {
        typedef decltype(mode) EM;
        ONEDC_LINE_SETUP(F32)
        EpiRes<F32> res[T::FI][T::FJ];
#pragma unroll
        for (int j = 0; j < T::FJ; ++j) {
            const int m = m0 + wm * T::WTM + j * 16 + frow;
            const ConvOutPix op = conv_out_pix(p, m);
#pragma unroll
            for (int i = 0; i < T::FI; ++i) {
                if (glu && (i & 1)) continue;          // odd tiles are the gate halves, consumed with their value tile
                const EpiPos e = conv_epi_pos<EM>(p, glu, cout_store, nw + i * 16, fq, op, m < M);
                conv_epi_load_res<F32, EM>(p, rb, cout_store, e, res[i][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < T::FJ; ++j) {
            const int m = m0 + wm * T::WTM + j * 16 + frow;
            const ConvOutPix op = conv_out_pix(p, m);
#pragma unroll
            for (int i = 0; i < T::FI; ++i) {
                if (glu && (i & 1)) continue;
                const int i2 = (T::FI >= 2 && i + 1 < T::FI) ? i + 1 : i;
                const EpiPos e = conv_epi_pos<EM>(p, glu, cout_store, nw + i * 16, fq, op, m < M);
                uint32_t lv0 = 0, lv1 = 0;
                conv_epi_finish<F32, EM>(p, yb, glu, cout_store, acc[i][j], acc[i2][j], bias[i], e, res[i][j], ssum[i],
                                         ssq[i], ssum_hi[i], ssq_hi[i], line, lv0, lv1);
                ONEDC_LINE_PUT()
            }
        }
        ONEDC_LINE_STORE()
    };
    ONEDC_EPI_DISPATCH(p, true, epilogue);
    if (p.stats) {                                 // wave-uniform
        static_assert(T::WTM == 64, "a wave covers one 64-pixel statistics block");
#pragma unroll
        for (int i = 0; i < T::FI; ++i) {
            if (glu && (i & 1)) continue;
            const int n_out = (glu ? (nw + i * 16) / 2 : nw + i * 16) + fq * 4;
            conv_stats_store(p, ssum[i], ssq[i], m0 + wm * T::WTM, n_out, cout_store, lane, M, ssum_hi[i], ssq_hi[i]);
        }
    }
}

// ---- 1x1 layers: the same tiles on a deeper pipeline ------------------------------------------------------------------------
// The UNet's Linear / 1x1 layers (decoder_unet.py:204-293: to_q|k, to_out, GEGLU, ff.net.2, proj_in / proj_out; K = 320 ... 5120)
// are short-K GEMMs whose time is the L2 -> LDS stream: conv_gemm_kernel above keeps ONE 64-channel K-step in flight and
// drains it (vmcnt(0)) before every barrier, so a K-step costs a full L2 round trip however little there is to multiply
// (round 3: 128 x 128 tiles at 370 - 620 TFLOP/s, 9 - 10 TB/s of LDS-DMA against the 17 - 19 TB/s the L2 serves to gathers).
// Here a K-step is 32 channels -- one MFMA k step; an LDS row is 64 B, a fragment read covers 1 KB of consecutive bytes, no
// swizzle -- and the ring has FOUR stages: three K-steps are in flight while one multiplies, one counted vmcnt and one barrier
// per step.  Same LDS footprint (64 KB / 48 KB), same wave tiles, same k order as the 64-channel kernel: the accumulators
// see the same MFMA sequence, so outputs and epilogue statistics are bit-identical to conv_gemm_kernel's.
// Input through a buffer descriptor (tail rows read zeros): the dispatcher keeps tensors of 4 GB and more on the old kernel.
// NSTAGE 3 (tile 134): 48 KB for the 128 x 128 tile, THREE workgroups per CU (experiment: a third resident workgroup
// under the prologue and epilogue of the other two).
template <class T, int NSTAGE = 4>
__global__ __launch_bounds__(256, (NSTAGE == 3 && T::TN * T::TM <= 128 * 128) ? 3 : 2) void conv_lin_kernel(const onedc_conv_params p, int tiles_n, int tiles_m, int M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TN = T::TN, TM = T::TM;
    constexpr int STAGE = (TN + TM) * 64;          // bytes per K-step: 32 channels of every row
    constexpr int WB = TN / 16, PB = TM / 16;      // 16-row DMA blocks of weights / pixels
    constexpr int NBLK = WB + PB, NI = NBLK / 4;   // DMA instructions per wave and stage
    static_assert(NBLK % 4 == 0, "blocks divide over the four waves");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % T::WN, wm = wave / T::WN;

    const int logical = xcd_logical_tile(blockIdx.x, tiles_n * tiles_m);
    const int n0 = (logical % tiles_n) * TN;
    const int m0 = (logical / tiles_n) * TM;

    const char *xb = (const char *)p.x + (int64_t)blockIdx.z * p.x_bs * 2;
    const char *wb = (const char *)p.w + (int64_t)blockIdx.z * p.w_bs * 2;
    const int K = p.Cin;                            // 1x1: k = input channel
    const int steps = K >> 5;
    // lane l of a DMA instruction moves granule (l & 3) of row (l >> 2) of a 16-row block
    const int drow = lane >> 2, dg = lane & 3;
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(wb, (unsigned)((int64_t)tiles_n * TN * K * 2));
    const int64_t x_total = (int64_t)M * p.x_cs * 2;
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(xb, (unsigned)x_total);           // (< 4 GB: conv_lin_ok)
    unsigned voff[NI];
    bool is_w[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int blk = wave + 4 * i;               // wave-uniform
        is_w[i] = blk < WB;
        if (blk < WB) {
            voff[i] = (unsigned)((n0 + blk * 16 + drow) * K * 2) + dg * 16;
        } else {
            const int m = m0 + (blk - WB) * 16 + drow;
            voff[i] = m < M ? (unsigned)(m * p.x_cs + p.x_co) * 2u + dg * 16 : 0xfffffff0u;       // out of range: zeros
        }
    }
    auto issue = [&](int s) {
        char *dst = smem + (s % NSTAGE) * STAGE;
        const unsigned so = (unsigned)s * 64;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int blk = wave + 4 * i;
            if (is_w[i]) glds16_buf(w_rsrc, voff[i], so, dst + blk * 1024);
            else glds16_buf(x_rsrc, voff[i], so, dst + blk * 1024);
        }
    };
    issue(0);
    if (steps > 1) issue(1);
    if (NSTAGE > 3 && steps > 2) issue(2);
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[T::FI][T::FJ];
#pragma unroll
    for (int i = 0; i < T::FI; ++i)
#pragma unroll
        for (int j = 0; j < T::FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    int fa_off = (wn * T::WTN + frow) * 64 + fq * 16;
    int fb_off = (TN + wm * T::WTM + frow) * 64 + fq * 16;
    asm volatile("" : "+v"(fa_off), "+v"(fb_off));

    for (int s = 0; s < steps; ++s) {
        // K-step s has landed (this wave's part): what may still fly are the NI instructions of each younger step
        if (NSTAGE > 3 && s + 2 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else if (s + 1 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // ... everywhere; and every wave is past its reads of step s - 1
        __builtin_amdgcn_sched_barrier(0);
        if (s + NSTAGE - 1 < steps) issue(s + NSTAGE - 1);            // into the stage of step s - 1
        const char *cur = smem + (s % NSTAGE) * STAGE;
        a16x8 a[T::FI], b[T::FJ];
#pragma unroll
        for (int i = 0; i < T::FI; ++i) a[i] = *(const a16x8 *)(cur + fa_off + i * 1024);
#pragma unroll
        for (int j = 0; j < T::FJ; ++j) b[j] = *(const a16x8 *)(cur + fb_off + j * 1024);
#pragma unroll
        for (int i = 0; i < T::FI; ++i)
#pragma unroll
            for (int j = 0; j < T::FJ; ++j) acc[i][j] = ONEDC_MFMA16(a[i], b[j], acc[i][j]);
    }

    // ---- epilogue: as conv_gemm_kernel ------------------------------------------------------------------------------------
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    const int cout_store = glu ? p.Cout / 2 : p.Cout;
    char *yb = (char *)p.y + (int64_t)blockIdx.z * p.y_bs * (p.out_f32 ? 4 : 2);
    const char *rb = p.res ? (const char *)p.res + (int64_t)blockIdx.z * p.r_bs * 2 : nullptr;
    const int nw = n0 + wn * T::WTN;
    float ssum[T::FI], ssq[T::FI], ssum_hi[T::FI], ssq_hi[T::FI];
    EpiBias bias[T::FI];
#pragma unroll
    for (int i = 0; i < T::FI; ++i) {
        ssum[i] = ssq[i] = ssum_hi[i] = ssq_hi[i] = 0.f;
        bias[i] = conv_epi_bias(p, glu, nw + i * 16, fq);
    }
    auto epilogue = [&](auto mode) {
        typedef decltype(mode) EM;
        ONEDC_LINE_SETUP(false)
        EpiRes<false> res[T::FI][T::FJ];
#pragma unroll
        for (int j = 0; j < T::FJ; ++j) {
            const int m = m0 + wm * T::WTM + j * 16 + frow;
            const ConvOutPix op = conv_out_pix(p, m);
#pragma unroll
            for (int i = 0; i < T::FI; ++i) {
                if (glu && (i & 1)) continue;
                const EpiPos e = conv_epi_pos<EM>(p, glu, cout_store, nw + i * 16, fq, op, m < M);
                conv_epi_load_res<false, EM>(p, rb, cout_store, e, res[i][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < T::FJ; ++j) {
            const int m = m0 + wm * T::WTM + j * 16 + frow;
            const ConvOutPix op = conv_out_pix(p, m);
#pragma unroll
            for (int i = 0; i < T::FI; ++i) {
                if (glu && (i & 1)) continue;
                const int i2 = (T::FI >= 2 && i + 1 < T::FI) ? i + 1 : i;
                const EpiPos e = conv_epi_pos<EM>(p, glu, cout_store, nw + i * 16, fq, op, m < M);
                uint32_t lv0 = 0, lv1 = 0;
                conv_epi_finish<false, EM>(p, yb, glu, cout_store, acc[i][j], acc[i2][j], bias[i], e, res[i][j], ssum[i],
                                           ssq[i], ssum_hi[i], ssq_hi[i], line, lv0, lv1);
                ONEDC_LINE_PUT()
            }
        }
        ONEDC_LINE_STORE()
    };
    ONEDC_EPI_DISPATCH(p, true, epilogue);
    if (p.stats) {
        static_assert(T::WTM == 64, "a wave covers one 64-pixel statistics block");
#pragma unroll
        for (int i = 0; i < T::FI; ++i) {
            if (glu && (i & 1)) continue;
            const int n_out = (glu ? (nw + i * 16) / 2 : nw + i * 16) + fq * 4;
            conv_stats_store(p, ssum[i], ssq[i], m0 + wm * T::WTM, n_out, cout_store, lane, M, ssum_hi[i], ssq_hi[i]);
        }
    }
}

template <class T, int NSTAGE = 4>
int launch_lin(const onedc_conv_params &p, hipStream_t st) {
    constexpr int LDS = NSTAGE * (T::TN + T::TM) * 64 > 4 * 64 * 64 * 2 ? NSTAGE * (T::TN + T::TM) * 64 : 4 * 64 * 64 * 2;   // (>= the epilogue's line staging)
    static OnedcLdsAttrOnce once;
    once.ensure({{(const void *)conv_lin_kernel<T, NSTAGE>, LDS}});
    const int M = (int)((int64_t)p.B * p.OH * p.OW);
    const int tiles_n = (p.Cout + T::TN - 1) / T::TN;
    const int tiles_m = (M + T::TM - 1) / T::TM;
    dim3 grid(tiles_n * tiles_m, 1, p.batch > 1 ? p.batch : 1);
    hipLaunchKernelGGL((conv_lin_kernel<T, NSTAGE>), grid, dim3(256), LDS, st, p, tiles_n, tiles_m, M);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

template <class T, bool F32 = false, bool SPLIT = false>
int launch(const onedc_conv_params &p, hipStream_t st) {
    static OnedcLdsAttrOnce once;
    once.ensure({{(const void *)conv_gemm_kernel<T, F32, SPLIT>, 2 * T::BUF}});
    const int64_t M64 = (int64_t)p.B * p.OH * p.OW;
    if (M64 > 0x7fffffff) return -1;
    const int M = (int)M64;
    const int tiles_n = (p.Cout + T::TN - 1) / T::TN;
    const int tiles_m = (M + T::TM - 1) / T::TM;
    const int Ktot = p.KH * p.KW * p.Cin;
    dim3 grid(tiles_n * tiles_m, 1, p.batch > 1 ? p.batch : 1);
    hipLaunchKernelGGL((conv_gemm_kernel<T, F32, SPLIT>), grid, dim3(256), 2 * T::BUF, st, p, tiles_n, tiles_m, M, Ktot);
    ONEDC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int onedc_conv_gemm256_launch(const onedc_conv_params &p, hipStream_t st);      // gemm_conv256.hip
int onedc_conv_gemm128x256_launch(const onedc_conv_params &p, hipStream_t st);  // gemm_conv256.hip
int onedc_conv_splitk_launch(const onedc_conv_params &p, hipStream_t st, int parts);   // gemm_conv256.hip
int64_t onedc_conv_splitk_bytes(const onedc_conv_params &p, int parts);
bool onedc_conv_win_ok(const onedc_conv_params &p);                             // gemm_conv_win.hip
int onedc_conv_win_launch(const onedc_conv_params &p, hipStream_t st);
bool onedc_conv_win_gn_ok(const onedc_conv_params &p);                          // gn_ab (if set) can be honoured
bool onedc_conv_win4_ok(const onedc_conv_params &p);                            // gemm_conv_win4.hip
int onedc_conv_win4_launch(const onedc_conv_params &p, hipStream_t st);
bool onedc_conv_ups4_ok(const onedc_conv_params &p);                            // (given onedc_conv_ups_ok)
int onedc_conv_ups4_launch(const onedc_conv_params &p, hipStream_t st);
bool onedc_conv_ups_ok(const onedc_conv_params &p);                             // gemm_conv_ups.hip
int onedc_conv_ups_launch(const onedc_conv_params &p, hipStream_t st);

// Which kernel family a launch goes to.  ONE function decides (onedc_conv_gemm launches what it returns, and
// onedc_conv_gemm_kernel_id reports it to callers such as bench.py), so the report cannot drift from the dispatch.
//   < 0: the parameters are rejected with that status
static int conv_validate(onedc_conv_params &p) {
    // 2-D pixel tiles for the 8-wave kernels (conv_common.h: ConvPatch) are OFF: they were speed-neutral, and they
    // regroup the pixels of a 64-pixel statistics block, so the GroupNorm partial sums of an 8-wave launch would
    // round differently from a 4-wave launch of the same layer -- and which kernel runs depends on the batch size.
    // With linear tiles every kernel family sums the same 64 pixels in the same order: results do not depend on it.
#ifdef EXP_PATCH16
    p.patch16 = (p.KH > 1 && (p.OH & 15) == 0 && (p.OW & 15) == 0 && !p.pixel_shuffle && p.batch <= 1) ? 1 : 0;
#else
    p.patch16 = 0;
#endif
    if (!p.x || !p.w || !p.y || !p.zeros) return -1;
    if (p.Cin <= 0 || (p.Cin & (p.in_f32 ? 31 : 63)) || p.Cout <= 0) return -2;
    if ((p.x_cs | p.x_co) & (p.in_f32 ? 3 : 7)) return -3; // 16-byte aligned source chunks
    if (p.res && ((p.r_cs | p.r_co) & 3)) return -8;       // 8-byte (bf16) / 16-byte (f32) aligned residual reads

    if (p.KH < 1 || p.KW < 1 || p.stride < 1 || p.OH < 1 || p.OW < 1 || p.B < 1) return -4;
    if (p.OW > 0xffff || p.OH > 0x7fff) return -5;
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    if (glu && (p.Cout & 31)) return -6;
    if (p.pixel_shuffle && ((p.Cout & 15) || glu)) return -7;
    if (p.stats && p.stats_gran != 2 && p.stats_gran != 4) return -10;
    if (p.stats && (p.pixel_shuffle || ((p.OH * p.OW) & 63) || (p.y_co & 3) || (p.y_cs & 3) ||
                    ((glu ? p.Cout / 2 : p.Cout) & 3) || p.batch > 1))
        return -10;
    return 0;
}

// the window kernel takes the layer (shape only, never the batch size: its k order is (chunk, tap), so its sums round
// differently from the other kernels, and a picture must decode to the same bits alone and inside a batch)
static bool conv_win_wanted(const onedc_conv_params &p) {
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    const int rows128w = (p.Cout + 127) / 128 * 128;
    return (p.tile == 0 || p.tile == 130 || p.tile == 131) && p.up != 2 && !p.in_f32 && onedc_conv_win_ok(p) && !(p.Cout <= 16 && !glu) &&
           rows128w * 5 <= p.Cout * 6;
}

// Split-K (gemm_conv256.hip): 3x3 layers on maps of at most 256 pixels with whole 256-channel tiles and >= 96 K-tiles -- the
// UNet's 12x12 level and the codec's 768-channel layers at the same resolution.  Decided by the layer's SHAPE alone (not by B: determinism rule of the path), five K ranges
// (1280 -> 1280: 36 K-tiles each, 2560 -> 1280: 72; 45 x 5 = 225 workgroups per batch of 16).  0: not split.
static int conv_splitk_parts(const onedc_conv_params &p) {
    if (p.split_k == 1 || p.in_f32 || p.batch > 1 || p.stats || p.gn_ab || p.x2 || p.pixel_shuffle || p.out_bf16 || p.up ||
        p.tile != 0 || p.KH != 3 || p.KW != 3 || (p.Cout % 256) != 0 || p.OH * p.OW > 256 ||
        !(p.epi == ONEDC_EPI_LINEAR || p.epi == ONEDC_EPI_LEAKY || p.epi == ONEDC_EPI_SILU) ||
        ((p.y_cs | p.y_co) & 3) || (p.res && ((p.r_cs | p.r_co) & 3)))
        return 0;
    const int ktiles = p.KH * p.KW * (p.Cin >> 6);
    // from 96 K-tiles on in four ranges: the codec's 768 -> 768 layers at 1/64 resolution (SemanticAdaptor / synthesis, 108
    // K-tiles, 27 tiles of 256 x 256 per batch of 16: 130 us on 216 tiles of the 4-wave kernel)
    const int parts = p.split_k > 1 ? p.split_k : ktiles >= 160 ? 5 : 4;
    if (ktiles < 96 && p.split_k <= 1) return 0;
    if (ktiles < 4 * parts) return 0;
    return parts;
}

extern "C" int64_t onedc_conv_splitk_ws_bytes(const onedc_conv_params *pp) {
    if (!pp) return 0;
    const int parts = conv_splitk_parts(*pp);
    return parts ? onedc_conv_splitk_bytes(*pp, parts) : 0;
}

// conv_lin_kernel: 1x1 / stride 1 layers on 16-bit operands whose input and weights a 32-bit buffer offset reaches
static bool conv_lin_ok(const onedc_conv_params &p) {
    const int64_t M = (int64_t)p.B * p.OH * p.OW;
    return conv_is_linear(p) && !p.in_f32 && !p.gn_ab && !p.x2 && M * p.x_cs * 2 < 0xfff00000ll &&
           (int64_t)((p.Cout + 127) / 128 * 128) * p.Cin * 2 < 0xfff00000ll && M < 0x7fffffff;
}

static int conv_choose_any(const onedc_conv_params &p);
// x_bf16 (bf16 operands under f16 storage, gemm_conv_win4.hip) exists in the 4-wave window kernel only: refused elsewhere
static int conv_choose(const onedc_conv_params &p) {
    const int id = conv_choose_any(p);
#ifdef ONEDC_ACT_F16
    return (p.x_bf16 && id >= 0 && id != ONEDC_KERNEL_WIN3X3_4W) ? -17 : id;
#else
    return id;      // the bf16 build multiplies bf16 operands in every kernel: the flag asks for what it does anyway (header: "ignored")
#endif
}

static int conv_choose_any(const onedc_conv_params &p) {
    const bool glu = p.epi == ONEDC_EPI_GLU_LEAKY || p.epi == ONEDC_EPI_GEGLU;
    {
        // a layer whose SHAPE says split-K runs split or not at all: falling back to the unsplit kernel when the caller's
        // scratch is missing or too small for this batch would make the sums depend on B (the unsplit kernel adds in another
        // order) -- the one thing the path promises never to do.  split_k = 1 is the way to ask for the unsplit form.
        const int parts = conv_splitk_parts(p);
        if (parts) return (p.ws && p.ws_bytes >= onedc_conv_splitk_bytes(p, parts)) ? ONEDC_KERNEL_SPLITK_256 : -15;
    }
    // up == 2: x2 upsample + 3x3 decomposed into four 2x2 phase convolutions; w is the phase-packed matrix
    if (p.up == 2) {
        if (!onedc_conv_ups_ok(p)) return -12;
        if (p.tile != 130 && onedc_conv_ups4_ok(p)) return ONEDC_KERNEL_UPS2X2_4W;
        if (p.out_bf16) return -12;                // (the 8-wave phase kernel stores the build's own format only)
        return p.tile == 131 ? -11 : ONEDC_KERNEL_UPS2X2;
    }
    const int64_t M64 = (int64_t)p.B * p.OH * p.OW;
#ifdef EXP_TILES
    // (experiment) smaller LDS footprints = more workgroups per CU on the short-K layers
    if (!p.in_f32 && p.tile >= 132 && p.tile <= 135) return 1000 + p.tile - 132;
#endif
    if (p.in_f32) {
        // the f32 matrix cores run at 1/16 of the 16-bit rate: a 128x128x256 tile is 15 us of MFMA time, and the prior
        // networks' maps give 72 .. 576 such tiles for 512 workgroup slots.  Grids of less than two full rounds take the
        // 64-row tile (twice the workgroups, half the work each; same k order, same bits): 256->256 at 48x48, B = 16:
        // 64 -> 45 us (tools/bench_f32.py)
        const int64_t blocks = ((p.Cout + 127) / 128) * ((M64 + 127) / 128) * (p.batch > 1 ? p.batch : 1);
        const bool small = blocks < 1024 && (p.Cout % 64) == 0 && p.tile != 128;
        if (p.in_f32 == 2) return small ? ONEDC_KERNEL_F32S_64X128 : ONEDC_KERNEL_F32S_128X128;
        return small ? ONEDC_KERNEL_F32_64X128 : ONEDC_KERNEL_F32_128X128;
    }
    // 3x3 window kernel (16x16-pixel patches x 128 channels, every input pixel staged once per 64-channel chunk):
    // 5-27 % faster than the 128x256 / 256x256 kernels on every 3x3 layer it can take (tools/bench_conv.py)
    // The 4-wave cut of it (two workgroups per CU hide each other's prologue and epilogue: 3-18 % faster, most on the
    // 128-channel layers at 768x768) takes every such layer whose epilogue is the plain linear one on whole 64-channel
    // wave tiles; the rest (fused upsample, activations, GLU, f32 / bf16 output, ragged channel counts) stays on the
    // 8-wave kernel.  tile 130 / 131 force one or the other.
    if (conv_win_wanted(p)) {
        if (p.tile != 130 && onedc_conv_win4_ok(p)) return ONEDC_KERNEL_WIN3X3_4W;
        if (p.x2) return -16;                      // only the 4-wave window kernel accumulates a second source
        if (p.tile == 131) return -11;
        return onedc_conv_win_gn_ok(p) ? ONEDC_KERNEL_WIN3X3 : -13;
    }
    if (p.x2) return -16;
    if (p.tile == 130 || p.tile == 131) return -11;
    if (p.gn_ab) return -13;                       // only the window kernel normalises on load
    if (p.tile == 136) return (conv_lin_ok(p) && (p.Cout % 128) == 0) ? ONEDC_KERNEL_LIN_128X256 : -11;
    // 256 x 256 tiles, 8-phase schedule: full 256-row channel tiles and enough pixel tiles to fill the chip
    const int64_t blocks256 = (p.Cout / 256) * ((M64 + 255) / 256) * (p.batch > 1 ? p.batch : 1);
    // short K (< 32 K-tiles: 1x1 layers up to 1984 channels): the 4-wave kernels win, two workgroups per CU hide each
    // other's prologue and epilogue (tools/bench_1x1.py; round 4 moved the bound from 16 to 32 K-tiles: the K = 1280 layers
    // of the 24x24 level run 10 - 22 % faster on four waves -- qk 1280 -> 2560 100 -> 78 us, out-projection 52 -> 46, 1280 ->
    // 1280 51 -> 43 -- while K = 2560 is level and K = 5120 / the 3x3 layers are 13 - 20 % faster on eight); all three
    // families give identical bits
    const int ktiles = p.KH * p.KW * (p.Cin >> 6);
    constexpr int kLongK = 32;
    if (p.tile != 128 && p.tile != 132 && p.tile != 133 && p.tile != 134 && (p.Cout % 256) == 0 && ((blocks256 >= 160 && ktiles >= kLongK) || p.tile == 256) &&
        !(p.stats && p.stats_gran == 2))           // the 256x256 kernel only leaves quad statistics
        return ONEDC_KERNEL_256X256;
    // 128 x 256 tiles (8 waves, 3-deep LDS ring): channel counts that are whole 128-row tiles or waste < 20 %
    const int rows128 = (p.Cout + 127) / 128 * 128;
    const int64_t blocks128 = (rows128 / 128) * ((M64 + 255) / 256) * (p.batch > 1 ? p.batch : 1);
    if (p.tile != 128 && p.tile != 132 && p.tile != 133 && p.tile != 134 && p.tile != 256 && !(p.Cout <= 16 && !glu) && rows128 * 5 <= p.Cout * 6 &&
        ((blocks128 >= 160 && ktiles >= kLongK) || p.tile == 129))
        return ONEDC_KERNEL_128X256;
    if (p.Cout <= 16 && !glu) return ONEDC_KERNEL_16X256;
    // 64-row tiles: channel counts that leave <= 64 rows in the last 128-row tile, and grids of fewer than 256
    // 128x128 tiles (the 12x12 maps of the UNet: 180 tiles for 256 CUs)
    const int64_t blocks4w = (rows128 / 128) * ((M64 + 127) / 128) * (p.batch > 1 ? p.batch : 1);
    // 1x1 layers of both 4-wave tile shapes run on the 32-channel / four-stage pipeline (conv_lin_kernel: same tiles, same k
    // order, same bits); tile 128 keeps the 64-channel kernel (tests, A/B)
    // conv_lin_kernel (32-channel K-steps, four-stage ring) is a MEASURED NEGATIVE, kept selectable (tile 133) with its tests:
    // tools/ab_round4.py / bench_1x1.py, B = 16 -- 4 - 13 % slower than the 64-channel kernel on the short-K layers (their time is
    // prologue + epilogue, not the K loop), level with the 8-wave kernels where K is long (5120 -> 1280 at 24x24: 119 us
    // against 139 for the 64-channel 4-wave kernel and 118 for the 128 x 256 tile the dispatcher picks there anyway), slower at
    // 12x12 (5120 -> 1280: 73 against 55 us).  Never the automatic choice.
    // ... except its THREE-stage form on the 128 x 128 tile (48 KB: three workgroups per CU instead of two) for K <= 384, the
    // level-0 Linear layers with wide outputs, whose life is prologue + epilogue and wants one more resident workgroup:
    // to_q|k 320 -> 1024 at 96x96 211 -> 182 us, GEGLU 320 -> 2560 397 -> 368, to_q 320 -> 512 88 -> 80; level from K = 640 on.
    // Round 5: the gated Linear layers (GEGLU of the UNet's feed-forward blocks, the codec's ConvFFN3 pairs) on the 128 x 256 tile
    // of the same kernel (wave tile 128 channels x 64 pixels, three 24 KB stages, two workgroups per CU): their life is the L2 ->
    // LDS stream of a wide N against a short K plus an arithmetic epilogue, and this tile moves 25 % fewer bytes per flop than
    // 128 x 128.  tools/ab_gated.py (interleaved rounds in one process against -DEXP_R5_BASELINE, B = 16): 640 -> 5120 at 48x48
    // 331 -> 304 us, 1280 -> 10240 at 24x24 308 -> 278, at 12x12 83 -> 77; 320 -> 2560 at 96x96 level (365 / 364: there the
    // three-stage 128 x 128 ring with its third resident workgroup was as good).  The plain projections are level or slower on
    // it (tools/bench_1x1.py 136: to_q|k 320 -> 1024 174 -> 206 us).  Same k order, same bits.  By layer shape and epilogue only.
#ifndef EXP_R5_BASELINE                             // (A/B build: round 5's dispatch and launch changes off, tools/r5_ab_bench.sh)
    if (glu && p.tile == 0 && conv_lin_ok(p) && (p.Cout % 128) == 0 && ktiles < kLongK) return ONEDC_KERNEL_LIN_128X256;
#endif
    const bool lin = conv_lin_ok(p) && (p.tile == 133 || p.tile == 134);
    const bool lin3 = conv_lin_ok(p) && p.tile == 0 && p.Cin <= 384;
    if (((p.Cout % 128) != 0 && (p.Cout % 128) <= 64) || (blocks4w < 256 && (p.Cout % 64) == 0 && p.Cout >= 128))
        return lin ? ONEDC_KERNEL_LIN_64X128 : ONEDC_KERNEL_64X128;
    return (lin || lin3) ? ONEDC_KERNEL_LIN_128X128 : ONEDC_KERNEL_128X128;
}

extern "C" int onedc_conv_gemm_kernel_id(const onedc_conv_params *pp) {
    if (!pp) return -1;
    onedc_conv_params p = *pp;
    const int rc = conv_validate(p);
    return rc < 0 ? rc : conv_choose(p);
}

extern "C" const char *onedc_conv_kernel_name(int id) {
    switch (id) {
        case ONEDC_KERNEL_WIN3X3: return "win3x3";
        case ONEDC_KERNEL_UPS2X2: return "ups2x2";
        case ONEDC_KERNEL_256X256: return "256x256";
        case ONEDC_KERNEL_128X256: return "128x256";
        case ONEDC_KERNEL_16X256: return "16x256";
        case ONEDC_KERNEL_64X128: return "64x128";
        case ONEDC_KERNEL_128X128: return "128x128";
        case ONEDC_KERNEL_F32_128X128: return "f32_128x128";
        case ONEDC_KERNEL_F32_64X128: return "f32_64x128";
        case ONEDC_KERNEL_F32S_128X128: return "f32s_128x128";
        case ONEDC_KERNEL_F32S_64X128: return "f32s_64x128";
        case ONEDC_KERNEL_WIN3X3_4W: return "win3x3_4w";
        case ONEDC_KERNEL_UPS2X2_4W: return "ups2x2_4w";
        case ONEDC_KERNEL_SPLITK_256: return "splitk_256";
        case ONEDC_KERNEL_LIN_64X128: return "lin_64x128";
        case ONEDC_KERNEL_LIN_128X128: return "lin_128x128";
        case ONEDC_KERNEL_LIN_128X256: return "lin_128x256";
        default: return "?";
    }
}

extern "C" int onedc_conv_gemm(const onedc_conv_params *pp, void *stream) {
    if (!pp) return -1;
    onedc_conv_params p = *pp;
    const int rc = conv_validate(p);
    if (rc < 0) return rc;
    hipStream_t st = as_stream(stream);
    const int id = conv_choose(p);
    switch (id) {
        case ONEDC_KERNEL_WIN3X3: return onedc_conv_win_launch(p, st);
        case ONEDC_KERNEL_WIN3X3_4W: return onedc_conv_win4_launch(p, st);
        case ONEDC_KERNEL_UPS2X2: return onedc_conv_ups_launch(p, st);
        case ONEDC_KERNEL_UPS2X2_4W: return onedc_conv_ups4_launch(p, st);
        case ONEDC_KERNEL_SPLITK_256: return onedc_conv_splitk_launch(p, st, conv_splitk_parts(p));
        case ONEDC_KERNEL_256X256: return onedc_conv_gemm256_launch(p, st);
        case ONEDC_KERNEL_128X256: return onedc_conv_gemm128x256_launch(p, st);
        case ONEDC_KERNEL_16X256: return launch<Tile<16, 256, 1, 4>>(p, st);
        case ONEDC_KERNEL_64X128: return launch<Tile<64, 128, 2, 2>>(p, st);
        case ONEDC_KERNEL_128X128: return launch<Tile<128, 128, 2, 2>>(p, st);
        case ONEDC_KERNEL_LIN_64X128:
            return p.tile == 134 ? launch_lin<Tile<64, 128, 2, 2>, 3>(p, st) : launch_lin<Tile<64, 128, 2, 2>>(p, st);
        case ONEDC_KERNEL_LIN_128X128:
            return p.tile != 133 ? launch_lin<Tile<128, 128, 2, 2>, 3>(p, st) : launch_lin<Tile<128, 128, 2, 2>>(p, st);
        case ONEDC_KERNEL_LIN_128X256: return launch_lin<Tile<128, 256, 1, 4>, 3>(p, st);
        case ONEDC_KERNEL_F32_128X128: return launch<Tile<128, 128, 2, 2>, true>(p, st);
        case ONEDC_KERNEL_F32_64X128: return launch<Tile<64, 128, 2, 2>, true>(p, st);
        case ONEDC_KERNEL_F32S_128X128: return launch<Tile<128, 128, 2, 2>, true, true>(p, st);
        case ONEDC_KERNEL_F32S_64X128: return launch<Tile<64, 128, 2, 2>, true, true>(p, st);
#ifdef EXP_TILES
        case 1000: return launch<Tile<128, 64, 4, 1>>(p, st);      // 48 KB: three workgroups per CU
        case 1001: return launch<Tile<64, 64, 4, 1>>(p, st);       // 32 KB: five
        case 1002: return launch<Tile<64, 128, 2, 2>>(p, st);      // 48 KB (the existing 64-row tile, forced)
        case 1003: return launch<Tile<128, 128, 2, 2>>(p, st);     // 64 KB (the existing tile, forced)
#endif
        default: return id < 0 ? id : -14;
    }
}

// Would onedc_conv_gemm run p with its second source (x2: the resnet's conv_shortcut accumulated behind the 3x3 taps)?  Asked
// before any launch of the block; GroupNorm on load is an independent question (both variants take a second source).
extern "C" int onedc_conv_win_takes_x2(const onedc_conv_params *pp) {
    if (!pp || !pp->x2) return 0;
    onedc_conv_params p = *pp;
    if (conv_validate(p) < 0) return 0;
    return conv_choose(p) == ONEDC_KERNEL_WIN3X3_4W ? 1 : 0;
}

// Would the window kernel take this layer INCLUDING a GroupNorm-on-load request (p.gn_ab may still be null: the caller
// asks before it computes the (a, b) pairs)?  False = use the separate scale/shift pass.
extern "C" int onedc_conv_win_takes(const onedc_conv_params *pp) {
    if (!pp) return 0;
    onedc_conv_params p = *pp;
    if (conv_validate(p) < 0) return 0;
    if (!conv_win_wanted(p)) return 0;
    static const float2 dummy = {0.f, 0.f};
    if (!p.gn_ab) p.gn_ab = (const float *)&dummy; // the question is about the normalising variant
    if (p.tile != 130 && onedc_conv_win4_ok(p)) return 1;
    if (p.tile == 131) return 0;
    return onedc_conv_win_gn_ok(p) ? 1 : 0;
}
